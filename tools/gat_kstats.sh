#!/bin/bash
# Per-kernel durations of the covalent branch by itself (tools/probe_gat.py), 16-lane-group backward kernels vs the wavefront-per-node pair:
# builds the -DCONAN_GAT_NO_GROUP16 library on the box, then one rocprofv3 --kernel-trace --stats run per library.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-gat_kstats}; mkdir -p $O; T=$(mktemp -d); cd /tmp; export TMPDIR=/tmp
mkdir -p $T/include $T/pkg; cp $R/include/conan_fgw_hip.h $T/include/; cp -r $R/conan-fgw_amd/csrc $T/pkg/csrc; rm -f $T/pkg/csrc/*.o
make -C $T/pkg/csrc -s -j16 "CXXFLAGS=-O3 -fPIC -std=c++17 --offload-arch=gfx950 -Wno-unused-function -DCONAN_GAT_NO_GROUP16=1" 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $O/g16 -o f -- python3 $R/tools/probe_gat.py "" group16 > $O/g16.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/pair -o p -- python3 $R/tools/probe_gat.py $T/pkg/libconan_fgw_hip.so pair > $O/pair.log 2>&1
for v in g16 pair; do echo "== $v"; f=$(find $O/$v -name "*kernel_stats.csv" | head -1); python3 - "$f" <<'P'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]: print(f"{r['Name'][:80]:80s} {int(r['Calls']):5d} {float(r['AverageNs'])/1e3:8.1f} us")
P
done | tee $O/summary.txt
find $O -name "*.db" -delete; rm -rf $T
