#!/bin/bash
# Round-3 measurement set (run on the GPU box through gpurun): bench lines, rocprofv3 kernel stats, PMC passes (each in its own
# run: --pmc with --kernel-trace only).  Everything lands in gpurun_out/r3p/; the summaries are copied to profiles/ afterwards.
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3p; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
B="python3 $R/bench.py"
$B > $O/bench_train.json 2> $O/bench_train.err
$B --mode fwd --no-cpu-baseline > $O/bench_fwd.json 2>/dev/null
$B --shape lipo --batch 128 --no-cpu-baseline > $O/bench_lipo.json 2>/dev/null
$B --model visnet --shape bace --batch 64 --no-cpu-baseline > $O/bench_visnet_bace.json 2>/dev/null
$B --shape freesolv --conformers 20 --batch 64 --no-cpu-baseline > $O/bench_freesolv_k20.json 2>/dev/null
python3 $R/tools/cfconv_cold.py > $O/cfconv_cold.json 2>/dev/null
python3 $R/tools/probe_stream_bw.py > $O/stream_bw.txt 2>/dev/null
python3 $R/tools/probe_step.py "" fused conan_filter_bwd_supported 2>/dev/null | grep "step ms" > $O/ab_filter_bwd.txt
for cfg in "train:" "lipo:--shape lipo --batch 128" "visnet_bace:--model visnet --shape bace --batch 64" "freesolv_k20:--shape freesolv --conformers 20 --batch 64"; do
  name=${cfg%%:*}; fl=${cfg#*:}
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$name -o $name -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --eager $fl > /dev/null 2>&1
done
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o f -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --eager > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o w -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --eager > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAVES --output-format csv -d $O/pmc_sq -o s -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --eager > $O/pmc_sq.log 2>&1
python3 $R/tools/pmc_summary.py $O/pmc_fetch $O/pmc_write $O/r3_pmc_hbm.json "bench.py --steps 3 --warmup 1 --no-cpu-baseline --eager" > $O/pmc_hbm.txt 2>&1
python3 $R/tools/pmc_sq.py $O/pmc_sq $O/r3_pmc_mfma.json "bench.py --steps 3 --warmup 1 --no-cpu-baseline --eager" > $O/pmc_sq.txt 2>&1
# the batched FGW solve by itself (cfg2 shape, the models' byte-wide adjacency): per-kernel durations and PMC passes of the coupling kernel
bash $R/tools/fgw_pmc.sh r3p/fgw_pmc > $O/fgw_pmc.txt 2>&1
python3 $R/tools/ab.py CONAN_FGW_NO_FAST=1 $R/tools/probe_fgw_small.py 2 2>/dev/null | grep -v amdgpu > $O/ab_fgw_small.txt
python3 $R/tools/ab.py CONAN_FGW_NO_BIG=1,CONAN_FGW_NO_BLOCK22=1 $R/tools/probe_fgw_large.py 2 2>/dev/null | grep -v amdgpu > $O/ab_fgw_large.txt
python3 $R/tools/ab.py CONAN_FILTER_BF16X3=1 $R/tools/probe_filter_fwd.py 2 2>/dev/null | grep -v amdgpu > $O/ab_filter_fwd.txt
python3 $R/tools/probe_edge_bwd.py "" in-tree 2>/dev/null | grep -v amdgpu > $O/edge_bwd.txt
python3 $R/tools/ab.py CONAN_LINEAR_BF16X3=1 $R/tools/probe_linear.py 2 2>/dev/null | grep -v amdgpu > $O/ab_linear.txt
python3 $R/tools/ab.py CONAN_GAT_NO_GROUP16=1 $R/tools/probe_gat.py 2 2>/dev/null | grep -v amdgpu > $O/ab_gat.txt
bash $R/tools/linear_pmc.sh r3p/linear_pmc > $O/linear_pmc.txt 2>&1
bash $R/tools/linear_pmc.sh r3p/wgrad_pmc wgrad_pmc.py > $O/wgrad_pmc.txt 2>&1
bash $R/tools/gat_kstats.sh r3p/gat_kstats > $O/gat_kstats.txt 2>&1
find $O -name "*_kernel_stats.csv" | head; ls $O
# keep the merge small: drop raw traces
find $O -name "*kernel_trace.csv" -size +8M -delete; find $O -name "*.db" -delete
du -sh $O
