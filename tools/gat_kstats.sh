#!/bin/bash
# Per-kernel durations of the covalent branch by itself (tools/probe_gat.py): one rocprofv3 --kernel-trace --stats run of the in-tree library.
# (Round 3 compared the 16-lane-group backward kernels against the wavefront-per-node pair through -DCONAN_GAT_NO_GROUP16; that switch and the
# old pair were removed from gat.hip in round 4 — the A/B is profiles/r3_ab_gat.txt.)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-gat_kstats}; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/g16 -o f -- python3 $R/tools/probe_gat.py "" group16 > $O/g16.log 2>&1
f=$(find $O/g16 -name "*kernel_stats.csv" | head -1); python3 - "$f" <<'P' | tee $O/summary.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]: print(f"{r['Name'][:80]:80s} {int(r['Calls']):5d} {float(r['AverageNs'])/1e3:8.1f} us")
P
find $O -name "*.db" -delete
