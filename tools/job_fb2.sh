cd $GRAFT_REPO_ROOT
python3 tools/probe_filter_bwd2.py
for i in 1 2; do
python3 bench.py --steps 20 --warmup 3 --blocks 3 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('bench', d['value'], d['ms_per_step'], d['blocks']['ms_per_step'], 'eager', d['eager']['ms_per_step'])"
done
python -m pytest tests -m gpu -q -x 2>&1 | tail -n 3
