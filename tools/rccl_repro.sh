#!/bin/bash
# Reproduce the 1-rank RCCL training step (tests/test_gpu_zz_rccl.py) with everything the child says kept.
# usage (GPU box): bash tools/rccl_repro.sh [tag] [extra bench.py args]
tag=${1:-a}; shift
out=gpurun_out/rccl_repro_$tag
mkdir -p $out
export TORCH_SHOW_CPP_STACKTRACES=1 NCCL_DEBUG=WARN PYTHONFAULTHANDLER=1
unset HSA_ENABLE_IPC_MODE_LEGACY
ulimit -c 0
for i in 1 2 3; do
  timeout 600 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node=1 --master-addr 127.0.0.1 --master-port $((29500 + i)) \
      bench.py --gpus 1 --steps 5 --warmup 3 --blocks 3 --no-cpu-baseline --force-collective "$@" > $out/run$i.out 2> $out/run$i.err
  echo "run $i rc=$?" | tee -a $out/summary.txt
done
tail -n 40 $out/run1.err
