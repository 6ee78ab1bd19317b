#!/bin/bash
# graph-replay cfg2 step with the node-level slab kernels batched (in-tree default) and immediate (python-level switch), alternating
cd $GRAFT_REPO_ROOT
for r in 1 2; do
python bench.py --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('batched  ', d['ms_per_step'], d['eager']['ms_per_step'])"
python -c "
import sys, runpy
import conan_fgw_amd.ops as o
o._LATE_STAGE1_ROWS = 0
sys.argv = ['bench.py', '--no-cpu-baseline']
runpy.run_path('bench.py', run_name='__main__')" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('immediate', d['ms_per_step'], d['eager']['ms_per_step'])"
done
