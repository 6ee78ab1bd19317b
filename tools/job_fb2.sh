cd $GRAFT_REPO_ROOT
python3 tools/probe_filter_bwd2.py
python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "one_pass" 2>&1 | tail -n 2
