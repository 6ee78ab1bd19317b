"""Probe for tools/ab.py: the edge-level backward kernels of one SchNet interaction at cfg2 size (P = 259k pairs, F = 128, 50
Gaussians), each timed with HIP events around 20 back-to-back launches, and checked against an fp64 torch product."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from conan_fgw_amd import _lib
if len(sys.argv) > 1 and sys.argv[1]:
    _lib._SO = sys.argv[1]
tag = sys.argv[2] if len(sys.argv) > 2 else ""
from conan_fgw_amd.ops import call, ptr, stream_ptr
dev = torch.device("cuda:0")
P, F, Gs = 259_000, 128, 50
torch.manual_seed(0)
g = torch.randn(P, F, device=dev); h1 = torch.rand(P, F, device=dev) * 2
dist = torch.rand(P, device=dev) * 10; offset = torch.linspace(0, 10, Gs, device=dev); coeff = -0.5 / float(offset[1] - offset[0]) ** 2
w2 = torch.randn(F, F, device=dev) / 11
md = torch.tensor([P], dtype=torch.int32, device=dev)
dw = torch.empty(F, F, device=dev); db = torch.empty(F, device=dev); dw1 = torch.empty(F, Gs, device=dev)
ws = torch.empty(_lib.lib().conan_linear_wgrad_ws(P, F, F), device=dev)
dx = torch.empty(P, F, device=dev)
f32 = torch.float32


def timed(fn, reps=20):
    for _ in range(3): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3


def k2(): call("conan_linear_wgrad", ptr(g), ptr(h1), P, F, F, ptr(md), ptr(dw), ptr(db), ptr(ws), stream_ptr())
def k4(): call("conan_rbf_wgrad", ptr(g), ptr(dist), P, ptr(offset, f32), Gs, coeff, F, ptr(md), ptr(dw1), ptr(db), ptr(ws), stream_ptr())
def k3(): call("conan_linear_fwd", ptr(g), ptr(w2), None, ptr(h1), P, F, F, 1, 2, ptr(md), ptr(dx), stream_ptr())


t2, t4, t3 = timed(k2), timed(k4), timed(k3)
k2(); ref = g.double().t() @ h1.double(); e2 = float((dw.double() - ref).abs().max() / ref.abs().max())
rb = torch.exp(coeff * (dist[:, None].double() - offset[None].double()) ** 2)
k4(); ref = g.double().t() @ rb; e4 = float((dw1.double() - ref).abs().max() / ref.abs().max())
eb = float((db.double() - g.double().sum(0)).abs().max() / g.double().sum(0).abs().max())
# node-level sizes (25 k atoms): a 128 -> 128 Linear forward, its dx, and its weight gradient
Mn = 25275
xn = torch.randn(Mn, F, device=dev); yn = torch.empty(Mn, F, device=dev); bn = torch.randn(F, device=dev)
wsn = torch.empty(_lib.lib().conan_linear_wgrad_ws(Mn, F, F), device=dev)
def n1(): call("conan_linear_fwd", ptr(xn), ptr(w2), ptr(bn), None, Mn, F, F, 0, 0, None, ptr(yn), stream_ptr())
def n2(): call("conan_linear_fwd", ptr(xn), ptr(w2), ptr(bn), None, Mn, F, F, 0, 1, None, ptr(yn), stream_ptr())
def n3(): call("conan_linear_wgrad", ptr(xn), ptr(yn), Mn, F, F, None, ptr(dw), ptr(db), ptr(wsn), stream_ptr())
tn1, tn2, tn3 = timed(n1, 50), timed(n2, 50), timed(n3, 50)
n1(); en = float((yn.double() - (xn.double() @ w2.double().t() + bn.double())).abs().max())
print(f"{tag} node-level (M={Mn}): linear {tn1:6.1f} us (abs err {en:.1e})  linear+ssp {tn2:6.1f} us  wgrad+reduce {tn3:6.1f} us")
extra = ""
if hasattr(_lib.lib(), "conan_filter_bwd"):
    ws2 = torch.empty(_lib.lib().conan_filter_bwd_ws(P, Gs, F), device=dev); dwf = torch.empty(F, Gs, device=dev); dbf = torch.empty(F, device=dev)
    def kf(): call("conan_filter_bwd", ptr(g), ptr(h1), ptr(dist), P, ptr(offset, f32), Gs, coeff, ptr(w2), F, ptr(md), ptr(dwf), ptr(dbf), ptr(ws2), None, stream_ptr())
    tf = timed(kf)
    dh = (g.double() @ w2.double()) * (1 - 0.5 * torch.exp(-h1.double()))
    rw, rbias = dh.t() @ rb, dh.sum(0)
    extra = f"   fused dx+dw1 {tf:7.1f} us (err {float((dwf.double() - rw).abs().max() / rw.abs().max()):.1e}, bias {float((dbf.double() - rbias).abs().max() / rbias.abs().max()):.1e})"
# the two-plane fp16 forms (round 3): gradient scaled from its device-side maximum
gmax = g.abs().max().reshape(1).contiguous()
if hasattr(_lib.lib(), "conan_linear_wgrad_scaled"):
    def k2h(): call("conan_linear_wgrad_scaled", ptr(g), ptr(h1), P, F, F, ptr(md), ptr(dw), ptr(db), ptr(ws), ptr(gmax), stream_ptr())
    t2h = timed(k2h); k2h(); ref = g.double().t() @ h1.double(); e2h = float((dw.double() - ref).abs().max() / ref.abs().max())
    def kfh(): call("conan_filter_bwd", ptr(g), ptr(h1), ptr(dist), P, ptr(offset, f32), Gs, coeff, ptr(w2), F, ptr(md), ptr(dwf), ptr(dbf), ptr(ws2), ptr(gmax), stream_ptr())
    tfh = timed(kfh); kfh()
    extra += f"\n{tag} f16x2: wgrad[128x128] {t2h:7.1f} us (err {e2h:.1e})   fused dx+dw1 {tfh:7.1f} us (err {float((dwf.double() - rw).abs().max() / rw.abs().max()):.1e}, bias {float((dbf.double() - rbias).abs().max() / rbias.abs().max()):.1e})"
print(f"{tag} wgrad[128x128] {t2:7.1f} us (err {e2:.1e})   rbf wgrad[128x50] {t4:7.1f} us (err {e4:.1e}, bias {eb:.1e})   dx+ssp' {t3:7.1f} us{extra}")
