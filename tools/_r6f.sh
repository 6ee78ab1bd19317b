cd $GRAFT_REPO_ROOT
python3 -m pytest tests/test_gpu_fgw.py -x -q 2>&1 | tail -8
python3 tools/sweep_fgw_complete.py 0 40 2>&1 | grep -v amdgpu | tail -6
PROBE_COMPLETE=1 python3 tools/ab.py tools/_old tools/probe_fgw_small.py 2 2>&1 | grep -v amdgpu
