"""The RCCL path of the training step, exercised on ONE GPU (the boxes of this pool have one): a fresh child process started through
torch.distributed.run — exactly how the driver starts the ranks of `bench.py --gpus N` — builds a 1-rank "nccl" (= RCCL) group and
runs the world > 1 step, graph A -> all_reduce of the flat gradient buffer -> graph B, because of --force-collective.
Reference: Lightning DDP of the training harness (conan_fgw/src/trainer.py:315-319; sampler data/datamodules.py:40-41)."""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _bench(args, distributed):
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.pop("HSA_ENABLE_IPC_MODE_LEGACY", None)                  # bench.py must set it by itself before its first GPU call
    cmd = [sys.executable]
    if distributed:
        cmd += ["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1", "--master-port", str(_free_port())]
    cmd += [os.path.join(ROOT, "bench.py")] + args
    out = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    return json.loads(line)


def test_one_rank_rccl_step_runs_the_collective_and_matches_the_plain_step():
    common = ["--gpus", "1", "--steps", "5", "--warmup", "3", "--blocks", "3", "--no-cpu-baseline"]
    d = _bench(common + ["--force-collective"], distributed=True)
    assert d["rccl_ranks"] == 1 and d["n_gpus"] == 1
    assert d["allreduce_us"] is not None and d["allreduce_us"] > 0
    assert d["allreduce"]["in_timed_step"] and d["allreduce"]["calls_per_step"] == 1 and d["allreduce"]["forced"]
    assert d["graph_capture_error"] is None and "HIP-graph" in d["config"]["execution"]
    p = _bench(common, distributed=False)
    assert p["rccl_ranks"] == 0 and not p["allreduce"]["in_timed_step"]
    # same model, same batch, same seed: the loss after the same number of optimiser steps agrees (a 1-rank sum is the identity) ...
    assert abs(d["loss"]["last"] - p["loss"]["last"]) <= 1e-4 * abs(p["loss"]["last"]) + 1e-6
    # ... and one more call per step costs a few tens of microseconds, not a different step (boxes of this pool differ by ~4 %)
    assert d["ms_per_step"] <= 1.10 * p["ms_per_step"] + 0.05, (d["ms_per_step"], p["ms_per_step"])
    assert d["value"] >= 0.90 * p["value"] - 1.0


def test_overlapped_buckets_with_deferred_weight_gradients():
    """2 ranks (gloo, both on cuda:0): FlatGradients.backward() with deferred weight gradients AND the overlapped early bucket.  The
    early-bucket hook must not mistake not-yet-accumulated siblings of a multi-output autograd node for copied gradients."""
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "_rank_probe_gpu.py")]
    out = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert d["error"] is None, d["error"]
    assert d["n_ranks"] == 2 and 0 < d["early"] < d["total"]
    assert all(n == 2 for n in d["launches"]), d["launches"]            # the early bucket really travelled on its own
