#!/bin/bash
# PMC passes over the edge-level Linear kernel (tools/linear_pmc.py; or tools/wgrad_pmc.py as second argument): each pass in its own run, --pmc with --kernel-trace only.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-linear_pmc}; W=${2:-linear_pmc.py};  mkdir -p $O; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $O/p1 -o a -- python3 $R/tools/$W > $O/p1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_BUSY_CYCLES --output-format csv -d $O/p2 -o b -- python3 $R/tools/$W > $O/p2.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/p3 -o c -- python3 $R/tools/$W > $O/p3.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/p4 -o d -- python3 $R/tools/$W > $O/p4.log 2>&1
python3 $R/tools/pmc_dump.py "linear|wgrad" $O/p1 $O/p2 $O/p3 $O/p4 | tee $O/summary.txt
find $O -name "*.db" -delete
