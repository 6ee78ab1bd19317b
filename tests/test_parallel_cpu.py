"""Data-parallel plumbing on CPU with the gloo backend, world_size 2 (the N>1 path of bench.py runs the same code over
RCCL): molecule sharding keeps conformers together, and ONE flat all-reduce reproduces the single-process gradient."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from conan_fgw_amd.parallel import FlatGradients, shard_range  # noqa: E402


def test_shard_range_partitions_molecules():
    for n, w in [(256, 8), (1024, 8), (10, 3), (7, 8), (1, 2)]:
        spans = [shard_range(n, r, w) for r in range(w)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
        sizes = [b - a for a, b in spans]
        assert max(sizes) - min(sizes) <= 1


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _toy_model():
    torch.manual_seed(0)
    return torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.Tanh(), torch.nn.Linear(5, 1))


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    model = _toy_model()
    flat = FlatGradients(model.parameters())
    g = torch.Generator().manual_seed(1)
    X, Y = torch.randn(16, 6, generator=g), torch.randn(16, 1, generator=g)
    lo, hi = shard_range(16, rank, world)
    for step in range(2):                      # two steps: zero() must not leave stale values behind
        flat.zero()
        loss = torch.nn.functional.mse_loss(model(X[lo:hi]), Y[lo:hi])
        loss.backward()
        flat.all_reduce_mean()
        lo_ptr, hi_ptr = flat.flat.data_ptr(), flat.flat.data_ptr() + 4 * flat.flat.numel()
        assert all(lo_ptr <= p.grad.data_ptr() < hi_ptr for p in flat.params)     # the optimizer sees the reduced buffer
    np.save(os.path.join(out_dir, f"grad{rank}.npy"), flat.flat.numpy())
    dist.destroy_process_group()


def test_flat_allreduce_matches_single_process(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    g0, g1 = np.load(tmp_path / "grad0.npy"), np.load(tmp_path / "grad1.npy")
    assert np.array_equal(g0, g1)                                   # every rank holds the same averaged gradient
    model = _toy_model()
    g = torch.Generator().manual_seed(1)
    X, Y = torch.randn(16, 6, generator=g), torch.randn(16, 1, generator=g)
    torch.nn.functional.mse_loss(model(X), Y).backward()            # equal shard sizes => mean of shard means == global mean
    ref = torch.cat([p.grad.reshape(-1) for p in model.parameters()]).numpy()
    np.testing.assert_allclose(g0, ref, rtol=1e-5, atol=1e-7)


def test_flat_gradients_deduplicates_shared_parameters():
    from conan_fgw_amd.schnet import SchNetNoSum
    m = SchNetNoSum(torch.device("cpu"), hidden_channels=32, num_filters=32, num_interactions=2)
    flat = FlatGradients(m.parameters())
    assert flat.flat.numel() == sum(p.numel() for p in m.parameters())      # mlp / conv.nn aliases counted once


def test_bench_spawn_path_forms_an_n_rank_group(tmp_path):
    """`python bench.py --gpus N` (no WORLD_SIZE in the environment) must start N ranks itself: bench.spawn_ranks drives
    torch.distributed.run exactly as the driver's launch line does.  The child here is tests/_rank_probe.py (gloo, CPU): the
    rank plumbing + the REAL parameter set of SchNetNoSum through FlatGradients with the overlapped early bucket."""
    import json
    import subprocess
    import bench
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    code = ("import sys, bench; sys.exit(bench.spawn_ranks(2, %r, ['--gpus', '2', '--steps', '3']))"
            % os.path.join(ROOT, "tests", "_rank_probe.py"))
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(line) == 1, r.stdout                                 # ONE JSON line, from rank 0
    out = json.loads(line[0])
    assert out["n_ranks"] == 2 and out["args"] == ["--gpus", "2", "--steps", "3"]
    assert out["ok"] and out["aliased"]
    assert 0 < out["early"] < out["total"] and out["launches"] == [2, 2]      # early bucket + late bucket, every step
    # and bench.main() takes that branch exactly when --gpus > 1 and no WORLD_SIZE is set
    a = bench.parse(["--gpus", "4"])
    assert a.gpus == 4 and "WORLD_SIZE" not in env


def test_overlap_falls_back_to_one_allreduce_when_single_process():
    m = _toy_model()
    flat = FlatGradients(m.parameters())
    flat.enable_overlap()
    flat.zero(); m(torch.randn(4, 6)).sum().backward()
    early, total = flat.calibrate()
    assert early == 0 and total == 4                                # world size 1: no early bucket
    flat.zero(); m(torch.randn(4, 6)).sum().backward()
    g = [p.grad.clone() for p in flat.params]
    flat.all_reduce_mean()
    for p, gg in zip(flat.params, g):
        assert torch.equal(p.grad, gg)
    assert flat.last_allreduce_launches == 0


def test_clip_grad_norm_on_a_cpu_flat_buffer_follows_torch():
    """FlatGradients.clip_grad_norm_ on a CPU buffer (the gloo tests' host logic; on the GPU it is conan_grad_clip_flat): the same rule as
    torch.nn.utils.clip_grad_norm_ — coefficient min(1, max_norm / (norm + 1e-6)) applied in place to the aliased .grad views."""
    torch.manual_seed(0)
    ps = [torch.nn.Parameter(torch.randn(7, 5)), torch.nn.Parameter(torch.randn(11))]
    qs = [torch.nn.Parameter(p.detach().clone()) for p in ps]
    for scale in (3.0, 1e-3):
        fg = FlatGradients(ps)
        for p, q in zip(ps, qs):
            g = torch.randn_like(p) * scale
            p.grad, q.grad = g.clone(), g.clone()
        fg.pack()
        n1 = fg.clip_grad_norm_(1.0)
        n2 = torch.nn.utils.clip_grad_norm_(qs, 1.0)
        assert abs(float(n1) - float(n2)) < 1e-5 * float(n2)
        for p, q in zip(ps, qs):
            assert torch.allclose(p.grad, q.grad, rtol=1e-5, atol=1e-8)
