#!/bin/bash
# PMC passes over the fused filter-network backward (tools/probe_filter_bwd2.py <M> fused): each pass in its own run, --pmc with --kernel-trace only.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-fb2_pmc}; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
W="python3 $R/tools/probe_filter_bwd2.py 259048 fused"
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $O/p1 -o a -- $W > $O/p1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_BUSY_CYCLES --output-format csv -d $O/p2 -o b -- $W > $O/p2.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/p3 -o c -- $W > $O/p3.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/p4 -o d -- $W > $O/p4.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_SALU SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS --output-format csv -d $O/p5 -o e -- $W > $O/p5.log 2>&1
python3 $R/tools/pmc_dump.py "filter_bwd2" $O/p1 $O/p2 $O/p3 $O/p4 $O/p5 | tee $O/summary.txt
find $O -name "*.db" -delete
