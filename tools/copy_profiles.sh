#!/bin/bash
# gpurun_out/r6p (tools/collect_profiles.sh) -> profiles/r6_*
R=$(cd "$(dirname "$0")/.." && pwd); O=$R/gpurun_out/r6p; P=$R/profiles
last() { grep '^{' "$1" | tail -n 1; }
last $O/bench_train.json > $P/r6_bench_train.json
last $O/bench_fwd.json > $P/r6_bench_fwd.json
last $O/bench_lipo.json > $P/r6_bench_lipo_b128.json
last $O/bench_visnet_bace.json > $P/r6_bench_visnet_bace_b64.json
last $O/bench_freesolv_k20.json > $P/r6_bench_freesolv_k20_b64.json
cp $O/cfconv_cold.json $P/r6_cfconv_warm_cold.json
cp $O/stream_bw.txt $P/r6_stream_bandwidth.txt
cp $O/filter_bwd2_vs_pair.txt $P/r6_filter_bwd2_vs_pair.txt
cp $O/fb2_phases.txt $P/r6_filter_bwd2_phases.txt
cp $O/fb2_pmc/summary.txt $P/r6_filter_bwd2_pmc.txt
cp $O/step_timeline.txt $P/r6_step_timeline.txt
cp $O/visnet_step_timeline.txt $P/r6_visnet_step_timeline.txt
cp $O/visnet_edge_linears.txt $P/r6_visnet_edge_linears.txt
cp $O/r6_pmc_hbm.json $O/r6_pmc_mfma.json $P/
cp $O/fgw_pmc/summary.txt $P/r6_fgw_pmc_counters.txt
cp $O/fgw_pmc/fgw_pmc_sq.json $P/r6_fgw_pmc_sq.json
cp $O/fgw_pmc/fgw_pmc_hbm.json $P/r6_fgw_pmc_hbm.json
cp $O/pytest_gpu.log $P/r6_pytest_gpu.log
for n in train lipo visnet_bace freesolv_k20; do
  f=$(find $O/prof_$n -name "*kernel_stats.csv" | head -n 1)
  case $n in train) d=r6_train;; lipo) d=r6_lipo_b128;; visnet_bace) d=r6_visnet_bace_b64;; freesolv_k20) d=r6_freesolv_k20_b64;; esac
  [ -n "$f" ] && cp $f $P/${d}_kernel_stats.csv
done
cp $O/filter_cfconv.txt $P/r6_filter_cfconv_fused.txt
cp $O/fgw_large.txt $P/r6_fgw_large.txt; cp $O/fgw_small.txt $P/r6_fgw_small.txt; cp $O/cfconv_bwd.txt $P/r6_cfconv_bwd.txt
last $O/bench_lipo_fwd.json > $P/r6_bench_lipo_b128_fwd.json
ls -la $P | grep r6_
