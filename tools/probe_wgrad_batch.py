"""In-process sweep of the row-slice count of the batched node-level weight gradients (ops.LATE_SLICES): the 22 jobs of a cfg2 backward pass
(14 of 128 x 128, 8 of 64-wide layers, 25 275 rows each) through conan_linear_wgrad_slabs_batch + conan_wgrad_reduce_batch, alternating settings
inside ONE process (per-process A/Bs carry a first-process bias: profiles/r5_ab_filter_fwd_768_threads.txt)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from conan_fgw_amd import ops
dev = torch.device("cuda:0")
M = 25275
torch.manual_seed(0)
shapes = [(128, 128)] * 14 + [(64, 128)] * 2 + [(64, 64)] * 6            # (N, K)
gs = [torch.randn(M, n, device=dev) for n, k in shapes]
xs = [torch.randn(M, k, device=dev) for n, k in shapes]
ws = [torch.randn(n, k, device=dev) for n, k in shapes]


def run(ev=None):
    with ops.deferred_weight_gradients():
        outs = [ops._wgrad(g, x, M, k, n, None, w, True) for g, x, w, (n, k) in zip(gs, xs, ws, shapes)]
        if ev: ev[0].record()
    if ev: ev[1].record()                                   # the context's exit ran the two batched launches (slab kernels, reduction)
    return outs


def timed(reps=20):
    for _ in range(3): run()
    tot = 0.0
    for _ in range(reps):
        ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        torch.cuda.synchronize()
        run(ev); torch.cuda.synchronize()
        tot += ev[0].elapsed_time(ev[1])
    return tot / reps * 1e3


ref = [o[0].clone() for o in run()]
settings = [int(s) for s in (sys.argv[1].split(",") if len(sys.argv) > 1 else "0,96,48,24".split(","))]
for rnd in range(3):
    line = []
    for sl in (settings if rnd % 2 == 0 else settings[::-1]):
        ops.LATE_SLICES = sl
        t = timed()
        err = max(float((o[0] - r).abs().max() / r.abs().max()) for o, r in zip(run(), ref))
        line.append(f"slices {sl or 'default(198)'}: {t:6.1f} us (max rel diff to default {err:.1e})")
    print(" | ".join(line), flush=True)
ops.LATE_SLICES = 0
