#!/bin/bash
# tools/visnet_kstats.sh for a -D variant of the kernels: bash tools/visnet_kstats_variant.sh <out-name> "<-D flags>"
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-visnet_variant}; mkdir -p $O; T=$(mktemp -d); cd /tmp; export TMPDIR=/tmp
mkdir -p $T/include $T/pkg; cp $R/include/conan_fgw_hip.h $T/include/; cp -r $R/conan-fgw_amd/csrc $T/pkg/csrc; rm -f $T/pkg/csrc/*.o
make -C $T/pkg/csrc -s -j16 "CXXFLAGS=-O3 -fPIC -std=c++17 --offload-arch=gfx950 -Wno-unused-function $2" 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $O/p -o v -- python3 $R/tools/bench_with_lib.py $T/pkg/libconan_fgw_hip.so --model visnet --shape bace --batch 64 --steps 6 --warmup 2 --blocks 1 --no-cpu-baseline --eager > $O/bench.log 2>&1
f=$(find $O/p -name "*kernel_stats.csv" | head -1); python3 - "$f" <<'P' | tee $O/summary.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:40]:
    if any(k in r['Name'] for k in ("attn", "vec_agg", "edge_up", "edge_emb", "ne_scale")): print(f"{r['Name'][:70]:70s} {int(r['Calls']):6d} {float(r['AverageNs'])/1e3:8.1f} us")
P
find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -size +8M -delete; rm -rf $T
