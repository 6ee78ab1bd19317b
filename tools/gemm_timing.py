"""Micro-benchmark of the edge-level GEMM kernels at the cfg2 edge count (E=518096, F=128)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from conan_fgw_amd._lib import call, ptr, stream_ptr, lib
dev = torch.device("cuda:0")
E, F = int(os.environ.get("ROWS", 518096)), 128
g = torch.randn(E, F, device=dev); h = torch.randn(E, F, device=dev); w = torch.randn(F, F, device=dev) * 0.1
rbf = torch.rand(E, 50, device=dev)
md = torch.tensor([E], dtype=torch.int32, device=dev)
out = torch.empty(E, F, device=dev); dw = torch.empty(F, F, device=dev); db = torch.empty(F, device=dev); dw1 = torch.empty(F, 50, device=dev)
ws = torch.empty(int(lib().conan_linear_wgrad_ws(E, F, F)), device=dev)
def timeit(label, fn, bytes_, flops):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): fn()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    print(f"{label:34s} {dt*1e6:8.1f} us  {bytes_/dt/1e12:5.2f} TB/s  {flops/dt/1e12:6.1f} TFLOP/s")
timeit("wgrad g^T h  [128x128]", lambda: call("conan_linear_wgrad", ptr(g), ptr(h), E, F, F, ptr(md), ptr(dw), ptr(db), ptr(ws), stream_ptr()), 2*E*F*4, 2*E*F*F)
timeit("wgrad g^T rbf [128x50]", lambda: call("conan_linear_wgrad", ptr(g), ptr(rbf), E, 50, F, ptr(md), ptr(dw1), ptr(db), ptr(ws), stream_ptr()), E*(F+50)*4, 2*E*F*50)
dist = torch.rand(E, device=dev) * 10; off = torch.linspace(0, 10, 50, device=dev)
timeit("rbf_wgrad g^T rbf(d) [128x50]", lambda: call("conan_rbf_wgrad", ptr(g), ptr(dist), E, ptr(off), 50, -12.0, F, ptr(md), ptr(dw1), ptr(db), ptr(ws), stream_ptr()), E*(F+1)*4, 2*E*F*50)
timeit("linear dx (act=2, w_kn)", lambda: call("conan_linear_fwd", ptr(g), ptr(w), None, ptr(h), E, F, F, 1, 2, ptr(md), ptr(out), stream_ptr()), 3*E*F*4, 2*E*F*F)
timeit("linear fwd (act=0)", lambda: call("conan_linear_fwd", ptr(g), ptr(w), None, None, E, F, F, 0, 0, ptr(md), ptr(out), stream_ptr()), 2*E*F*4, 2*E*F*F)
