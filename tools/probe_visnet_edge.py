"""Probe for tools/ab.py: the ViSNet edge-level streaming kernels by themselves (BACE B=64 size: 20 k atoms, 24 edges per atom, H = 128) —
conan_visnet_edge_update and conan_visnet_vec_aggregate_bwd on a synthetic CSR with local neighbourhoods."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from conan_fgw_amd import _lib
if len(sys.argv) > 1 and sys.argv[1]:
    _lib._SO = sys.argv[1]
tag = sys.argv[2] if len(sys.argv) > 2 else ""
from conan_fgw_amd.ops import call, ptr, stream_ptr
dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
n, deg, H = 20330, 24, 128
E = n * deg
tgt = np.repeat(np.arange(n), deg)
col = np.clip(tgt + rng.integers(-40, 41, E), 0, n - 1)
order = np.lexsort((col, tgt)); col = col[order]
rowptr = np.arange(0, E + 1, deg)
srt = np.argsort(col, kind="stable"); t_eid = srt; t_rowptr = np.concatenate([[0], np.cumsum(np.bincount(col, minlength=n))])
I = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.int32)).to(dev)
rowptr, col, tgt, t_rowptr, t_eid = I(rowptr), I(col), I(tgt), I(t_rowptr), I(t_eid)
ne = torch.tensor([E], dtype=torch.int32, device=dev)
torch.manual_seed(0)
R = lambda *s: torch.randn(*s, device=dev)
wt, ws, vec, dvagg = R(n, 3, H), R(n, 3, H), R(n, 3, H), R(n, 3, H)
t, f, fo = R(E, H), R(E, H), torch.empty(E, H, device=dev)
d3 = torch.nn.functional.normalize(R(E, 3), dim=1)
s_pre, ds, dvec = R(E, 2 * H), torch.empty(E, 2 * H, device=dev), torch.empty(n, 3, H, device=dev)
def timed(fn, reps=10):
    for _ in range(2): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
t1 = timed(lambda: call("conan_visnet_edge_update", ptr(wt), ptr(ws), ptr(t), ptr(d3), ptr(col), ptr(tgt), ptr(ne), E, H, 1, ptr(f), ptr(fo), stream_ptr()))
t2 = timed(lambda: call("conan_visnet_vec_aggregate_bwd", ptr(vec), ptr(s_pre), ptr(d3), ptr(dvagg), ptr(col), ptr(tgt), ptr(t_rowptr), ptr(t_eid), ptr(ne), E, n, H, 1,
                        ptr(ds), ptr(dvec), stream_ptr()))
print(f"{tag} edge_update {t1:6.1f} us   vec_aggregate_bwd (s + v passes) {t2:6.1f} us   checksums {float(fo.double().sum()):.6e} {float(ds.double().sum()):.6e} {float(dvec.double().sum()):.6e}")
