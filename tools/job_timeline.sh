cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_visnet.py -m gpu -q -x 2>&1 | tail -n 15
rocprofv3 --kernel-trace -d gpurun_out/tl4 -o t --output-format csv -- python3 bench.py --steps 10 --warmup 3 --blocks 1 --no-cpu-baseline > gpurun_out/tl4_bench.json 2> gpurun_out/tl4_bench.err
python3 tools/trace_timeline.py gpurun_out/tl4 > gpurun_out/tl4_timeline.txt 2>&1
tail -n 3 gpurun_out/tl4_timeline.txt
python3 -c "
import json; d=json.loads([l for l in open('gpurun_out/tl4_bench.json') if l.startswith('{')][-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('cold'))"
