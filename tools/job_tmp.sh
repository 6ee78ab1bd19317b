cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q -x 2>&1 | tail -n 3
/usr/bin/time -v python3 bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; tail -n 3 gpurun_out/bench_default.err | head -2; grep "Elapsed" gpurun_out/bench_default.err
python3 -c "
import json; d=json.loads([l for l in open('gpurun_out/bench_default.json') if l.startswith('{')][-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['cold']['frac'], d['cpu_baseline']['value'], d['cpu_baseline']['sample'][:80], d['with_input_pipeline'])"
