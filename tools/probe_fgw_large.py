"""Probe for tools/ab.py: the large-N FGW barycenter solve (BACE / lipo shapes), timed per solve."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from conan_fgw_amd import _lib
if len(sys.argv) > 1 and sys.argv[1]:
    _lib._SO = sys.argv[1]
tag = sys.argv[2] if len(sys.argv) > 2 else ""
from conan_fgw_amd import ops
dev = torch.device("cuda:0")
out = []
for name, B, K, N, d in (("bace B=64", 64, 5, 97, 64), ("lipo B=128", 128, 5, 85, 64), ("bace B=32", 32, 5, 97, 64)):
    g = torch.Generator().manual_seed(0)
    Ys = (torch.rand(B, K, N, d, generator=g) * 1.9 + 0.1).to(dev)
    A = (torch.rand(B, K, N, N, generator=g) < 0.1).float(); Cs = torch.triu(A, 1); Cs = (Cs + Cs.transpose(-1, -2))
    if os.environ.get("PROBE_PADDED"):       # molecules of different sizes padded to N as the model's glue leaves them: n real nodes (normal around 0.57 N, one molecule with
        # n = N), isolated padded nodes with one common feature row — round 6: the coupling kernels solve them as one node
        n = (torch.randn(B, generator=g) * 0.14 * N + 0.57 * N).round().clamp(8, N).long(); n[0] = N
        if os.environ.get("PROBE_ONLY_ABOVE"):      # diagnostic: molecules at or below that size shrink to 2 nodes (they end at once) — what the launch costs
            n = torch.where(n <= int(os.environ["PROBE_ONLY_ABOVE"]), torch.full_like(n, 2), n)      # without them = the bound of running them elsewhere
        real = (torch.arange(N)[None, :] < n[:, None]).float()
        Cs = Cs * (real[:, None, :, None] * real[:, None, None, :])
        for b_ in range(B): Cs[b_, :, int(n[b_]) - 1, 0] = 1.0; Cs[b_, :, 0, int(n[b_]) - 1] = 1.0
        Ys = torch.where(real.to(dev)[:, None, :, None] > 0, Ys, torch.full_like(Ys, 0.5))
    Cs = Cs.to(dev)
    ts = []
    for small_int in (True, False):          # the models' promise (adjacency bytes in LDS) / general fp32 structure matrices
        for _ in range(2): ops.fgw_barycenter_batched(Ys, Cs, cs_small_int=small_int)
        one = []                              # solve by solve, median: a host hiccup (allocator growth, 5-70 ms once per process on some boxes) is not the kernel's time
        for _ in range(7):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            ops.fgw_barycenter_batched(Ys, Cs, cs_small_int=small_int)
            torch.cuda.synchronize(); one.append((time.perf_counter() - t0) * 1e3)
        ts.append(sorted(one)[len(one) // 2])
    out.append("%s N=%d: %.3f ms (Cs as bytes) %.3f ms (Cs fp32)" % (name, N, ts[0], ts[1]))
print(tag, "  ".join(out))
