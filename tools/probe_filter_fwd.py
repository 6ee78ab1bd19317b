"""Probe for tools/ab.py: conan_filter_fwd at cfg2 size (259 k pair rows, F = 128, 50 Gaussians), with and without the h1 output."""
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
dist = torch.rand(P, device=dev) * 10; offset = torch.linspace(0, 10, Gs, device=dev); coeff = -0.5 / float(offset[1] - offset[0]) ** 2
w1 = torch.randn(F, Gs, device=dev) / 7; b1 = torch.randn(F, device=dev) / 10; w2 = torch.randn(F, F, device=dev) / 11; b2 = torch.randn(F, device=dev) / 10
cnt = torch.tensor([P], dtype=torch.int32, device=dev)
W = torch.empty(P, F, device=dev); h1 = torch.empty(P, F, device=dev)
def timed(fn, reps=20):
    for _ in range(3): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3
def f_train(): call("conan_filter_fwd", ptr(dist), ptr(cnt), P, ptr(offset), Gs, coeff, 10.0, F, ptr(w1), ptr(b1), ptr(w2), ptr(b2), ptr(W), ptr(h1), stream_ptr())
def f_infer(): call("conan_filter_fwd", ptr(dist), ptr(cnt), P, ptr(offset), Gs, coeff, 10.0, F, ptr(w1), ptr(b1), ptr(w2), ptr(b2), ptr(W), None, stream_ptr())
t1, t2 = timed(f_train), timed(f_infer)
f_train()
n = 20000
rb = torch.exp(coeff * (dist[:n, None].double() - offset[None].double()) ** 2)
hh = torch.nn.functional.softplus(rb @ w1.double().t() + b1.double()) - 0.6931471805599453
ref = (hh @ w2.double().t() + b2.double()) * (0.5 * (torch.cos(dist[:n].double() * 3.141592653589793 / 10.0) + 1))[:, None]
print(f"{tag} filter fwd: with h1 {t1:6.1f} us   W only {t2:6.1f} us   rel err W {float((W[:n].double() - ref).norm() / ref.norm()):.1e}  h1 {float((h1[:n].double() - hh).norm() / hh.norm()):.1e}")
