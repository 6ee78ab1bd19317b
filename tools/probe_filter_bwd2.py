"""Fused filter-network backward (conan_filter_bwd2) against the two kernels it replaces (conan_filter_bwd + conan_linear_wgrad_scaled) at the cfg2
pair-row count: per-call microseconds (HIP events around 20 back-to-back calls, caches as the loop leaves them) and the bytes each reads."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from conan_fgw_amd._lib import call, lib, ptr, stream_ptr
dev = torch.device("cuda:0")
M, F, Gs = int(sys.argv[1]) if len(sys.argv) > 1 else 259048, 128, 50
gen = torch.Generator().manual_seed(1)
g = torch.randn(M, F, generator=gen).to(dev); h1 = (torch.rand(M, F, generator=gen) * 3 - 0.6).to(dev)
dist = (torch.rand(M, generator=gen) * 10).to(dev); w2 = (torch.randn(F, F, generator=gen) / 11).to(dev)
off = torch.linspace(0, 10, Gs).to(dev); coeff = -0.5 / float(off[1] - off[0]) ** 2
md = torch.tensor([M], dtype=torch.int32, device=dev); gmax = g.abs().max().reshape(1).contiguous()
dW1, db1, dW2, db2 = torch.empty(F, Gs, device=dev), torch.empty(F, device=dev), torch.empty(F, F, device=dev), torch.empty(F, device=dev)
ws2 = torch.empty(int(lib().conan_filter_bwd2_ws(M, Gs, F)), device=dev)
wsa = torch.empty(int(lib().conan_filter_bwd_ws(M, Gs, F)), device=dev); wsb = torch.empty(int(lib().conan_linear_wgrad_ws(M, F, F)), device=dev)
def fused():
    call("conan_filter_bwd2", ptr(g), ptr(h1), ptr(dist), M, ptr(off), Gs, coeff, ptr(w2), F, ptr(md), ptr(gmax), None, None, None, None, ptr(ws2), stream_ptr())
def pair():
    call("conan_linear_wgrad_scaled", ptr(g), ptr(h1), M, F, F, ptr(md), None, None, ptr(wsb), ptr(gmax), stream_ptr())
    call("conan_filter_bwd", ptr(g), ptr(h1), ptr(dist), M, ptr(off), Gs, coeff, ptr(w2), F, ptr(md), None, None, ptr(wsa), ptr(gmax), stream_ptr())
def timed(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return 1e3 * s.elapsed_time(e) / n
if len(sys.argv) > 2 and sys.argv[2] == "fused":      # PMC workload: the fused kernel alone
    for _ in range(30): fused()
    torch.cuda.synchronize()
    sys.exit(0)
for rnd in range(3):
    tf, tp = timed(fused), timed(pair)
    print(f"M={M}: fused {tf:7.1f} us ({2 * M * F * 4 / tf / 1e6:6.2f} TB/s of g + h1)   pair {tp:7.1f} us", flush=True)
