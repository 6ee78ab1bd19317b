"""Probe for tools/ab.py: the fused pair of node-level linears (conan_mlp2_fwd / _bwd) against the launches it replaces, M = 25 275."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from conan_fgw_amd import _lib
if len(sys.argv) > 1 and sys.argv[1]:
    _lib._SO = sys.argv[1]
tag = sys.argv[2] if len(sys.argv) > 2 else ""
from conan_fgw_amd.ops import call, ptr, stream_ptr
dev = torch.device("cuda:0")
M, F = 25275, 128
g = torch.Generator().manual_seed(0)
x = torch.randn(M, F, generator=g).to(dev); res = torch.randn(M, F, generator=g).to(dev)
w1 = (torch.randn(F, F, generator=g) / 11).to(dev); b1 = torch.randn(F, generator=g).to(dev); w2 = (torch.randn(F, F, generator=g) / 11).to(dev); b2 = torch.randn(F, generator=g).to(dev)
mid = torch.empty(M, F, device=dev); y = torch.empty(M, F, device=dev); dm = torch.empty(M, F, device=dev); dx = torch.empty(M, F, device=dev); tmp = torch.empty(M, F, device=dev)
def timed(fn, reps=100):
    for _ in range(5): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3
def fused_f(): call("conan_mlp2_fwd", ptr(x), ptr(w1), ptr(b1), ptr(w2), ptr(b2), ptr(res), M, F, F, F, ptr(mid), ptr(y), stream_ptr())
def comp_f():
    call("conan_linear_fwd", ptr(x), ptr(w1), ptr(b1), None, M, F, F, 0, 1, None, ptr(mid), stream_ptr())
    call("conan_linear_fwd", ptr(mid), ptr(w2), ptr(b2), ptr(res), M, F, F, 0, 0, None, ptr(y), stream_ptr())
def fused_b(): call("conan_mlp2_bwd", ptr(y), ptr(w2), ptr(w1), ptr(mid), M, F, F, F, ptr(dm), ptr(dx), stream_ptr())
def comp_b():
    call("conan_linear_fwd", ptr(y), ptr(w2), None, None, M, F, F, 1, 0, None, ptr(tmp), stream_ptr())
    call("conan_ssp_bwd", ptr(tmp), ptr(mid), M, F, None, ptr(dm), stream_ptr())
    call("conan_linear_fwd", ptr(dm), ptr(w1), None, None, M, F, F, 1, 0, None, ptr(dx), stream_ptr())
print(f"{tag} forward: fused {timed(fused_f):5.1f} us, two launches {timed(comp_f):5.1f} us   backward: fused {timed(fused_b):5.1f} us, three launches {timed(comp_b):5.1f} us")
