"""Probe for tools/ab.py: edge-level weight gradients that share their x (482 k rows, K = 128): three N = 128 jobs as separate launches and as one
conan_linear_wgrad_slabs_batch launch, and one N = 256 job (two n tiles)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from conan_fgw_amd import _lib
if len(sys.argv) > 1 and sys.argv[1]:
    _lib._SO = sys.argv[1]
tag = sys.argv[2] if len(sys.argv) > 2 else ""
from conan_fgw_amd._lib import lib, WgradSlabJob
from conan_fgw_amd.ops import call, ptr, stream_ptr
dev = torch.device("cuda:0")
torch.manual_seed(0)
M, K = 482_110, 128
x = torch.randn(M, K, device=dev); gs = [torch.randn(M, 128, device=dev) for _ in range(3)]; g2 = torch.randn(M, 256, device=dev)
ws = [torch.empty(int(lib().conan_linear_wgrad_ws(M, K, 128)), device=dev) for _ in range(3)]
ws2 = torch.empty(int(lib().conan_linear_wgrad_ws(M, K, 256)), device=dev)
dW = torch.empty(256, K, device=dev); db = torch.empty(256, device=dev)
def timed(fn, reps=10):
    for _ in range(2): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
def separate():
    for q in range(3): call("conan_linear_wgrad_slabs", ptr(gs[q]), ptr(x), M, K, 128, None, ptr(ws[q]), stream_ptr())
sj = (WgradSlabJob * 3)()
for q in range(3):
    sj[q].g, sj[q].x, sj[q].m_dev, sj[q].ws = ptr(gs[q]), ptr(x), None, ptr(ws[q]); sj[q].M, sj[q].K, sj[q].N, sj[q].slices = M, K, 128, 0
def batch(): call("conan_linear_wgrad_slabs_batch", sj, 3, stream_ptr())
def wide(): call("conan_linear_wgrad", ptr(g2), ptr(x), M, K, 256, None, ptr(dW), ptr(db), ptr(ws2), stream_ptr())
t1, t2, t3 = timed(separate), timed(batch), timed(wide)
print(f"{tag} 3 x N=128 separate {t1:6.1f} us   one batched launch {t2:6.1f} us   N=256 (+ reduce) {t3:6.1f} us   checksum {float(dW.double().sum()):.6e}")
