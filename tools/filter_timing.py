"""Ablation timing of the fused filter kernel at the cfg2 edge count."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from conan_fgw_amd import ops
from conan_fgw_amd._lib import call, ptr, stream_ptr
dev = torch.device("cuda:0")
E, F, Gs = 518096, 128, 50
dist = (torch.rand(E, device=dev) * 9 + 0.7)
ne = torch.tensor([E], dtype=torch.int32, device=dev)
off = torch.linspace(0, 10, Gs, device=dev); coeff = -0.5 / (10 / 49) ** 2
w1 = torch.randn(F, Gs, device=dev) * 0.1; b1 = torch.zeros(F, device=dev); w2 = torch.randn(F, F, device=dev) * 0.1; b2 = torch.zeros(F, device=dev)
W = torch.empty(E, F, device=dev); h1 = torch.empty(E, F, device=dev)
def run(h):
    call("conan_filter_fwd", ptr(dist), ptr(ne), E, ptr(off), Gs, coeff, 10.0, F, ptr(w1), ptr(b1), ptr(w2), ptr(b2), ptr(W), ptr(h) if h is not None else None, stream_ptr())
for label, h in (("W only", None), ("W + h1", h1)):
    for _ in range(3): run(h)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): run(h)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    fl = E * 2 * (56 * F + F * F)
    print(f"{label:10s} dbg={os.environ.get('CONAN_FILTER_DEBUG','0')}: {dt*1e6:8.1f} us   {fl/dt/1e12:6.1f} TFLOP/s (padded K=56)")
