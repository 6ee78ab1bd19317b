#!/bin/bash
# per-kernel durations of the batched FGW solve (tools/fgw_pmc.py) under rocprofv3 --kernel-trace --stats
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-fgw_kstats}; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O -o k -- python3 $R/tools/fgw_pmc.py > $O/log.txt 2>&1
python3 - <<PY
import csv,glob,re
f=glob.glob("$O/**/*kernel_stats.csv",recursive=True)[0]
for r in csv.DictReader(open(f)):
    n=r['Name']; m=re.search(r'(k_\w+(<[^>]*>)?)',n); nm=m.group(1) if m else n[:40]
    print(f"{nm[:50]:50s} calls={r['Calls']:>4s} avg_us={float(r['AverageNs'])/1e3:8.1f}")
PY
