#!/bin/bash
# Per-kernel averages of the ViSNet BACE B=64 training step (eager, 6 steps) — the message / aggregate / update kernels and the edge-level GEMMs.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-visnet_kstats}; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/p -o v -- python3 $R/bench.py --model visnet --shape bace --batch 64 --steps 6 --warmup 2 --blocks 1 --no-cpu-baseline --eager > $O/bench.log 2>&1
f=$(find $O/p -name "*kernel_stats.csv" | head -1); python3 - "$f" <<'P' | tee $O/summary.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r['TotalDurationNs']) for r in rows)
for r in rows[:26]: print(f"{r['Name'][:70]:70s} {int(r['Calls']):6d} {float(r['AverageNs'])/1e3:8.1f} us {100*float(r['TotalDurationNs'])/tot:5.1f}%")
P
find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -size +8M -delete
