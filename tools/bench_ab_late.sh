#!/bin/bash
# graph-replay cfg2 step with python-level constants switched, alternating in one job:  [module.]NAME=VALUE pairs (module under conan_fgw_amd, default ops); extra bench.py flags in $BENCH_ARGS
cd $GRAFT_REPO_ROOT
for r in 1 2; do
for setting in "$@"; do
python -c "
import sys, runpy
import importlib
for kv in '$setting'.split(','):
    k, v = kv.split('=')
    mod, name = (k.rsplit('.', 1) if '.' in k else ('ops', k))
    setattr(importlib.import_module('conan_fgw_amd.' + mod), name, int(v))
sys.argv = ['bench.py', '--no-cpu-baseline'] + '${BENCH_ARGS:-}'.split()
runpy.run_path('bench.py', run_name='__main__')" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-40s' % '$setting', d['ms_per_step'], d['eager']['ms_per_step'])"
done
done
