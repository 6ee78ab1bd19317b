"""Probe: the edge-level Linear work of ONE ViS_MP layer at BACE B = 64 (E = 482 k directed edges, H = 128), launch by launch —
forward projections of f (dk / dv / f_proj in one launch), s_proj, their input-gradient GEMMs (today: a chain of launches that carries the
running sum through HBM) and their weight-gradient slab launches — each against the bytes it has to move (U = one [E,128] tensor).

    python tools/probe_edge_linears.py [library ("" = in-tree) [tag [E]]]        (tools/ab.py's probe convention)"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from conan_fgw_amd import _lib
if len(sys.argv) > 1 and sys.argv[1]:
    _lib._SO = sys.argv[1]
tag = sys.argv[2] if len(sys.argv) > 2 else ""
from conan_fgw_amd._lib import call, lib, ptr, stream_ptr, WgradSlabJob
E = int(sys.argv[3]) if len(sys.argv) > 3 else 482_000
H = 128
dev = torch.device("cuda:0")
torch.manual_seed(0)
f = torch.randn(E, H, device=dev)
ws = [torch.randn(H, H, device=dev) / 11 for _ in range(3)]
gs = [torch.randn(E, H, device=dev) for _ in range(3)]
ys = [torch.empty(E, H, device=dev) for _ in range(3)]
seed = torch.randn(E, H, device=dev)
dx = torch.empty(E, H, device=dev)
w_s = torch.randn(2 * H, H, device=dev) / 11
sact = torch.empty(E, 2 * H, device=dev)
gsact = torch.randn(E, 2 * H, device=dev)
U = E * H * 4 / 1e6        # MB
s = stream_ptr()


def timed(fn, reps=20):
    for _ in range(3): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


arr = lambda ts: (ctypes.c_void_p * len(ts))(*[t.data_ptr() for t in ts])


def multi_fwd():
    call("conan_linear_multi_fwd", ptr(f), arr(ws), None, E, H, H, 3, 0, None, arr(ys), None, s)


def dx_chain():
    call("conan_linear_fwd", ptr(gs[0]), ptr(ws[0]), None, ptr(seed), E, H, H, 1, 0, None, ptr(dx), s)
    call("conan_linear_fwd", ptr(gs[1]), ptr(ws[1]), None, ptr(dx), E, H, H, 1, 0, None, ptr(dx), s)
    call("conan_linear_fwd", ptr(gs[2]), ptr(ws[2]), None, ptr(dx), E, H, H, 1, 0, None, ptr(dx), s)


def dx_sum():
    call("conan_linear_sum_fwd", arr(gs), (ctypes.c_int * 3)(H, H, H), arr(ws), 3, 1, None, ptr(seed), E, H, None, ptr(dx), s)


def dx_sum2():
    call("conan_linear_sum_fwd", arr(gs[:2]), (ctypes.c_int * 2)(H, H), arr(ws[:2]), 2, 1, None, None, E, H, None, ptr(dx), s)


def sproj_fwd():
    call("conan_linear_fwd", ptr(f), ptr(w_s), None, None, E, H, 2 * H, 0, 0, None, ptr(sact), s)


def sproj_dx():
    call("conan_linear_fwd", ptr(gsact), ptr(w_s), None, None, E, 2 * H, H, 1, 0, None, ptr(dx), s)


wsz = int(lib().conan_linear_wgrad_ws(E, H, H))
slabs = [torch.empty(wsz, device=dev) for _ in range(3)]
sj = (WgradSlabJob * 3)()
for q in range(3):
    sj[q].g, sj[q].x, sj[q].m_dev, sj[q].ws = ptr(gs[q]), ptr(f), None, ptr(slabs[q])
    sj[q].M, sj[q].K, sj[q].N, sj[q].slices = E, H, H, 0


def wgrad_shared():
    call("conan_linear_wgrad_slabs_batch", sj, 3, s)


wsz2 = int(lib().conan_linear_wgrad_ws(E, H, 2 * H))
slab2 = torch.empty(wsz2, device=dev)


def wgrad_sproj():
    call("conan_linear_wgrad_slabs", ptr(gsact), ptr(f), E, H, 2 * H, None, ptr(slab2), s)


def wgrad_one():
    call("conan_linear_wgrad_slabs", ptr(gs[0]), ptr(f), E, H, H, None, ptr(slabs[0]), s)


rows = [("forward dk / dv / f_proj of f (1 launch)", multi_fwd, 4), ("input gradient of the three (3 chained launches)", dx_chain, 9),
        ("   ... in one launch (conan_linear_sum_fwd)", dx_sum, 5), ("   ... two layers, no seed (last ViS_MP layer)", dx_sum2, 3),
        ("forward s_proj 128 -> 256", sproj_fwd, 3), ("input gradient of s_proj 256 -> 128 (one launch since round 5; two chained before: 280 us)", sproj_dx, 3),
        ("weight-gradient slabs, three layers of one x (1 launch)", wgrad_shared, 4), ("weight-gradient slabs, one layer", wgrad_one, 2),
        ("weight-gradient slabs, s_proj (N = 256)", wgrad_sproj, 3)]
print(f"{tag} E = {E}, H = {H}: U = one [E,H] fp32 tensor = {U:.0f} MB")
for name, fn, u in rows:
    if fn is None:
        print(f"{name:62s} {'':8s}   {u} U = {u * U:6.0f} MB")
        continue
    t = timed(fn)
    print(f"{name:62s} {t:8.1f} us   {u} U = {u * U:6.0f} MB   {u * U / t:5.2f} TB/s")
