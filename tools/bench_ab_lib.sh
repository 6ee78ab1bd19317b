#!/bin/bash
# graph-replay cfg2 step with one or more `*_supported` entry points of the library forced to 0 (the composed fallback runs), alternating with
# the in-tree default in one job.  usage: bash tools/bench_ab_lib.sh conan_mlp2_supported[,conan_filter_bwd_supported] [bench flags]
cd $GRAFT_REPO_ROOT
names=$1; shift
for r in 1 2; do
for off in "" "$names"; do
python -c "
import sys, runpy
from conan_fgw_amd import _lib
L = _lib.lib()
for n in [s for s in '$off'.split(',') if s]:
    setattr(L, n, lambda *a: 0)
sys.argv = ['bench.py', '--no-cpu-baseline'] + '$*'.split()
runpy.run_path('bench.py', run_name='__main__')" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-44s' % ('off: $off' if '$off' else 'in-tree'), d['ms_per_step'], d['eager']['ms_per_step'], d.get('forward_only', {}).get('ms_per_step'))"
done
done
