"""Probe for tools/ab.py: the 128 -> 128 Linear forward as a function of the row count (fixed cost vs streaming cost)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from conan_fgw_amd import _lib
if len(sys.argv) > 1 and sys.argv[1]:
    _lib._SO = sys.argv[1]
tag = sys.argv[2] if len(sys.argv) > 2 else ""
from conan_fgw_amd.ops import call, ptr, stream_ptr
dev = torch.device("cuda:0")
F = 128
w = torch.randn(F, F, device=dev) / 11; b = torch.randn(F, device=dev)
def timed(fn, reps=100):
    for _ in range(5): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3
out = []
for M in (32, 1024, 8192, 25275, 65536, 131072):
    x = torch.randn(M, F, device=dev); y = torch.empty(M, F, device=dev)
    out.append("M=%d: %.1f us" % (M, timed(lambda: call("conan_linear_fwd", ptr(x), ptr(w), ptr(b), None, M, F, F, 0, 0, None, ptr(y), stream_ptr()))))
e = torch.cuda.Event(enable_timing=True); s = torch.cuda.Event(enable_timing=True)
z = torch.empty(64, device=dev)
out.append("torch fill(64 floats): %.1f us" % timed(lambda: z.fill_(1.0)))
print(tag, "  ".join(out))
