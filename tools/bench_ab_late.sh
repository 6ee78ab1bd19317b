#!/bin/bash
# graph-replay cfg2 step with python-level settings of the postponed slab kernels, alternating in one job:  NAME=VALUE pairs of ops.* constants
cd $GRAFT_REPO_ROOT
for r in 1 2; do
for setting in "$@"; do
python -c "
import sys, runpy
import conan_fgw_amd.ops as o
for kv in '$setting'.split(','):
    k, v = kv.split('='); setattr(o, k, int(v))
sys.argv = ['bench.py', '--no-cpu-baseline']
runpy.run_path('bench.py', run_name='__main__')" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-40s' % '$setting', d['ms_per_step'], d['eager']['ms_per_step'])"
done
done
