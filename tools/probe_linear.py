"""Probe for tools/ab.py: ops.linear (k_linear_t16: edge level 482 k rows, node level 25 k rows; plain and transposed weight) and the fused
pair ops.mlp2 / its backward at node level, 128-wide; timings and errors against fp64."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from conan_fgw_amd import _lib
if len(sys.argv) > 1 and sys.argv[1]:
    _lib._SO = sys.argv[1]
tag = sys.argv[2] if len(sys.argv) > 2 else ""
from conan_fgw_amd.ops import call, ptr, stream_ptr
dev = torch.device("cuda:0")
torch.manual_seed(0)
F = 128
w = torch.randn(F, F, device=dev) / 11; b = torch.randn(F, device=dev) / 10; w2 = torch.randn(F, F, device=dev) / 11
def timed(fn, reps=30):
    for _ in range(3): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3
out = []
for M in (482_110, 25_275):
    x = torch.randn(M, F, device=dev); y = torch.empty(M, F, device=dev)
    for w_kn, act in ((0, 0), (1, 0), (0, 3)):
        fn = lambda: call("conan_linear_fwd", ptr(x), ptr(w), ptr(b) if not w_kn else None, None, M, F, F, w_kn, act, None, ptr(y), stream_ptr())
        t = timed(fn)
        n = min(M, 20000)
        r = x[:n].double() @ (w.double() if w_kn else w.double().t()) + (0 if w_kn else b.double())
        if act == 3: r = r * torch.sigmoid(r)
        out.append(f"M={M} kn={w_kn} act={act}: {t:6.1f} us err {float((y[:n].double() - r).norm() / r.norm()):.1e}")
M = 25_275
x = torch.randn(M, F, device=dev); mid = torch.empty(M, F, device=dev); y = torch.empty(M, F, device=dev)
t = timed(lambda: call("conan_mlp2_fwd", ptr(x), ptr(w), ptr(b), ptr(w2), ptr(b), ptr(x), M, F, F, F, ptr(mid), ptr(y), stream_ptr()))
h = torch.nn.functional.softplus(x.double() @ w.double().t() + b.double()) - 0.6931471805599453
r = h @ w2.double().t() + b.double() + x.double()
out.append(f"mlp2 fwd {t:5.1f} us err {float((y.double() - r).norm() / r.norm()):.1e}")
dmid = torch.empty(M, F, device=dev); dx = torch.empty(M, F, device=dev)
t = timed(lambda: call("conan_mlp2_bwd", ptr(x), ptr(w2), ptr(w), ptr(mid), M, F, F, F, ptr(dmid), ptr(dx), stream_ptr()))
dh = (x.double() @ w2.double()) * (1.0 - 0.5 * torch.exp(-mid.double()))
r = dh @ w.double()
out.append(f"mlp2 bwd {t:5.1f} us err {float((dx.double() - r).norm() / r.norm()):.1e}")
print(tag, " | ".join(out))
