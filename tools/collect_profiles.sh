#!/bin/bash
# Round-6 measurement set (round 5's version of this script is in the history: 56400d1; round 4's: 2031e43)
# (run on the GPU box through gpurun): bench lines, rocprofv3 kernel stats, PMC passes (each in its own
# run: --pmc with --kernel-trace only).  Everything lands in gpurun_out/r6p/; the summaries are copied to profiles/ afterwards
# (tools/copy_profiles.sh).  Round 3's version of this script is in the history (d0d7844).
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6p; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
B="python3 $R/bench.py"
$B > $O/bench_train.json 2> $O/bench_train.err
$B --mode fwd --no-cpu-baseline > $O/bench_fwd.json 2>/dev/null
$B --config cfg3 --no-cpu-baseline > $O/bench_lipo.json 2>/dev/null
$B --config cfg3 --mode fwd --no-cpu-baseline > $O/bench_lipo_fwd.json 2>/dev/null
$B --config cfg4 --no-cpu-baseline > $O/bench_visnet_bace.json 2>/dev/null
$B --shape freesolv --conformers 20 --batch 64 --no-cpu-baseline > $O/bench_freesolv_k20.json 2>/dev/null
python3 $R/tools/cfconv_cold.py > $O/cfconv_cold.json 2>/dev/null
python3 $R/tools/probe_stream_bw.py > $O/stream_bw.txt 2>/dev/null
python3 $R/tools/probe_filter_bwd2.py 2>/dev/null | grep -v amdgpu > $O/filter_bwd2_vs_pair.txt
python3 $R/tools/probe_filter_cfconv.py "" cfg2 2>/dev/null | grep -v amdgpu > $O/filter_cfconv.txt
python3 $R/tools/probe_filter_cfconv.py "" lipo lipo 128 2>/dev/null | grep -v amdgpu >> $O/filter_cfconv.txt
python3 $R/tools/probe_fgw_large.py "" "dense random structures, no padding  " 2>/dev/null | grep -v amdgpu > $O/fgw_large.txt
PROBE_PADDED=1 python3 $R/tools/probe_fgw_large.py "" "sizes ~0.57 N padded to N (one at N)" 2>/dev/null | grep -v amdgpu >> $O/fgw_large.txt
python3 $R/tools/probe_edge_linears.py "" final 2>/dev/null | grep -v amdgpu > $O/visnet_edge_linears.txt
python3 $R/tools/probe_fgw_small.py "" "random structures (general path)" 2>/dev/null | grep -v amdgpu > $O/fgw_small.txt
PROBE_COMPLETE=1 python3 $R/tools/probe_fgw_small.py "" "complete graphs (row-sum form)   " 2>/dev/null | grep -v amdgpu >> $O/fgw_small.txt
python3 $R/tools/probe_cfconv_bwd.py "" final 2>/dev/null | grep -v amdgpu > $O/cfconv_bwd.txt
python3 $R/tools/probe_cfconv_bwd.py "" final lipo 128 2>/dev/null | grep -v amdgpu >> $O/cfconv_bwd.txt
for cfg in "train:" "lipo:--shape lipo --batch 128" "visnet_bace:--config cfg4" "freesolv_k20:--shape freesolv --conformers 20 --batch 64"; do
  name=${cfg%%:*}; fl=${cfg#*:}
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$name -o $name -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --eager $fl > /dev/null 2>&1
done
# graph-replay timeline of the cfg2 step (tools/trace_timeline.py)
rocprofv3 --kernel-trace -d $O/tl -o t --output-format csv -- python3 $R/bench.py --steps 10 --warmup 3 --blocks 1 --no-cpu-baseline > /dev/null 2>&1
python3 $R/tools/trace_timeline.py $O/tl > $O/step_timeline.txt 2>&1
rocprofv3 --kernel-trace -d $O/tlv -o t --output-format csv -- python3 $R/bench.py --steps 10 --warmup 3 --blocks 1 --no-cpu-baseline --config cfg4 > /dev/null 2>&1
python3 $R/tools/trace_timeline.py $O/tlv 0 > $O/visnet_step_timeline.txt 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o f -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --eager > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o w -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --eager > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAVES --output-format csv -d $O/pmc_sq -o s -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --eager > $O/pmc_sq.log 2>&1
python3 $R/tools/pmc_summary.py $O/pmc_fetch $O/pmc_write $O/r6_pmc_hbm.json "bench.py --steps 3 --warmup 1 --no-cpu-baseline --eager" > $O/pmc_hbm.txt 2>&1
python3 $R/tools/pmc_sq.py $O/pmc_sq $O/r6_pmc_mfma.json "bench.py --steps 3 --warmup 1 --no-cpu-baseline --eager" > $O/pmc_sq.txt 2>&1
# the batched FGW solve by itself (cfg2 shape, the models' byte-wide adjacency): per-kernel durations and PMC passes of the coupling kernel
bash $R/tools/fgw_pmc.sh r6p/fgw_pmc > $O/fgw_pmc.txt 2>&1
# the fused filter-network backward by itself
bash $R/tools/fb2_pmc.sh r6p/fb2_pmc > $O/fb2_pmc.txt 2>&1
python3 $R/tools/f2_phase_profile.py 2>/dev/null | grep -v amdgpu > $O/fb2_phases.txt
# the whole GPU suite of this tree
(cd $R && python3 -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1)
tail -n 3 $O/pytest_gpu.log
find $O -name "*kernel_trace.csv" -size +8M -delete; find $O -name "*.db" -delete
du -sh $O
