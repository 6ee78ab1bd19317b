"""Probe for tools/ab.py: time of the batched FGW solve on the N <= 64 path (cfg2 shape and the K = 20 FreeSolv shape), with the
adjacency promise the models make (cs_small_int).  argv[1] = library to load ("" = in-tree), argv[2] = tag."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from conan_fgw_amd import _lib
if len(sys.argv) > 1 and sys.argv[1]:
    _lib._SO = sys.argv[1]
from conan_fgw_amd import ops
tag = sys.argv[2] if len(sys.argv) > 2 else ""
dev = torch.device("cuda:0")
out = []
for (B, K, N, d) in ((256, 5, 33, 64), (64, 20, 33, 64), (256, 5, 26, 64)):
    g = torch.Generator().manual_seed(0)
    Ys = (torch.rand(B, K, N, d, generator=g) * 1.9 + 0.1).to(dev)
    if os.environ.get("PROBE_COMPLETE"):      # complete graphs on n real nodes (n as in the ESOL-shaped batches), padded nodes isolated, the K graphs of a molecule alike
        n = torch.randint(6, N + 1, (B,), generator=g); n[0] = N
        idx = torch.arange(N)
        real = (idx[None, :] < n[:, None]).float()
        Cs = (real[:, :, None] * real[:, None, :] * (1.0 - torch.eye(N)))[:, None].expand(B, K, N, N).contiguous().to(dev)
        if os.environ.get("PROBE_COMPLETE") != "random_pad":      # padded nodes carry one common feature row, as the model's glue leaves them (round 6: they are solved as one node)
            Ys = torch.where(real.to(dev)[:, None, :, None] > 0, Ys, torch.full_like(Ys, 0.5))
    else:
        A = (torch.rand(B, K, N, N, generator=g) < 0.5).float(); Cs = torch.triu(A, 1); Cs = (Cs + Cs.transpose(-1, -2)).to(dev)
    for _ in range(3): r = ops.fgw_barycenter_batched(Ys, Cs, cs_small_int=True)
    torch.cuda.synchronize()
    ts = []
    for rep in range(3):
        t0 = time.perf_counter()
        for _ in range(20): ops.fgw_barycenter_batched(Ys, Cs, cs_small_int=True)
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / 20 * 1e3)
    out.append(f"B{B}K{K}N{N}: " + "/".join("%.3f" % t for t in ts) + f" ms (sk {float(r[3][:, 2].float().mean()):.1f})")
print(tag, " | ".join(out), flush=True)
