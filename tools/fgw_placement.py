"""Where the workgroups of a coupling launch run and when they end: builds a PRIVATE copy of the library with -DCONAN_FGW_PROFILE, runs the
barycenter solve of a synthetic batch through the model path (ragged neighbour lists: the size-ordered dealing is on) and reads the per-workgroup
records (block, size the problem ran at, HW_ID / XCC_ID, first and last tick).  Prints, for the last coupling launch of the solve: how the
dispatcher spread the blocks over the CUs, per-CU load and end time, and which workgroups end the launch.
    python tools/fgw_placement.py [shape batch conformers]          (PROF_DEFS="-DNAME=VALUE ..." for a variant)"""
import ctypes, os, shutil, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
shape = sys.argv[1] if len(sys.argv) > 1 else "esol"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
K = int(sys.argv[3]) if len(sys.argv) > 3 else 5
tmp = tempfile.mkdtemp(prefix="conan_place_")
os.makedirs(os.path.join(tmp, "include")); shutil.copy(os.path.join(ROOT, "include", "conan_fgw_hip.h"), os.path.join(tmp, "include"))
src = os.path.join(tmp, "pkg", "csrc"); shutil.copytree(os.path.join(ROOT, "conan-fgw_amd", "csrc"), src, ignore=shutil.ignore_patterns("*.o"))
subprocess.check_call(["make", "-C", src, "-s", "-j16", "CXXFLAGS=-O3 -fPIC -std=c++17 --offload-arch=gfx950 -Wno-unused-function -DCONAN_FGW_PROFILE " + os.environ.get("PROF_DEFS", "")])
import numpy as np, torch
from conan_fgw_amd import _lib
_lib._SO = os.path.join(tmp, "pkg", "libconan_fgw_hip.so")
from conan_fgw_amd.collate import DeviceCollator, molecules_from_synthetic
from conan_fgw_amd.head import EmbeddingsWithGATAggregationBaryCenter
from conan_fgw_amd.synthetic import make_batch, make_bond_graph
L = _lib.lib()
dev = torch.device("cuda:0")
b = make_batch(shape, B, K, seed=1236); bg = make_bond_graph(b, seed=2236)
items = molecules_from_synthetic(b, bg)
sizes = np.array([len(it.z) for it in items])
torch.manual_seed(5)
model = EmbeddingsWithGATAggregationBaryCenter(K, dev).to(dev)
data = DeviceCollator(dev, K, depth=2, static=True)(items).wait()
cidx = model.create_aggregation_index(data.num_graphs, dev)
N = int(data.max_nodes)
TRACE = L.conan_debug_fgw_trace if N <= 64 else L.conan_debug_fgw_trace_large
TRACE.restype = ctypes.c_int; TRACE.argtypes = [ctypes.POINTER(ctypes.c_longlong), ctypes.c_int]
buf = (ctypes.c_longlong * (2 + 6 * 8192))()
PROF = L.conan_debug_fgw_prof if N <= 64 else L.conan_debug_fgw_prof_large
PROF.restype = ctypes.c_int; PROF.argtypes = [ctypes.POINTER(ctypes.c_longlong), ctypes.c_int]
pbuf = (ctypes.c_longlong * 32)()
with torch.no_grad():
    for _ in range(3): model(data, cidx, data.batch, num_graphs=data.num_graphs, max_nodes=data.max_nodes)
    torch.cuda.synchronize(); TRACE(buf, 1); PROF(pbuf, 1)
    model(data, cidx, data.batch, num_graphs=data.num_graphs, max_nodes=data.max_nodes)
    torch.cuda.synchronize(); TRACE(buf, 0); PROF(pbuf, 0)
n = min(int(buf[0]), 8192)
rec = np.array(buf[2:2 + 6 * n], dtype=np.int64).reshape(n, 6)
rec = rec[np.argsort(rec[:, 4], kind="stable")]
its, sks = (rec[:, 1] >> 8) & 0xfff, rec[:, 1] >> 20          # projected-gradient and Sinkhorn iterations of the coupling in this launch
rec[:, 1] &= 0xff
# launches: a record that starts after every earlier record has ended opens a new one
launches, cur, end = [], [0], rec[0, 5]
for i in range(1, n):
    if rec[i, 4] >= end: launches.append(cur); cur = []
    cur.append(i); end = max(end, rec[i, 5])
launches.append(cur)
names = (["staging", "dot(Y,Z) + base", "T0", "A = C1 @ T", "G -> K = exp(Mr - ref)", "K -> registers", "Sinkhorn iterations", "T store + err", "T -> global", "Ypart = T @ Z", "Cpart = T C2 T^T"] if N <= 64 else
         ["staging", "dot(Y,Z)", "base", "A = C1 @ T (product)", "G, K", "max |A| + digits + barrier", "Sinkhorn iterations", "T store + err", "T -> global", "Ypart = T @ Z", "Cpart = T C2 T^T"])
tot = sum(pbuf[:11])
print("phases of the coupling kernel in this forward pass, us per workgroup (mean over all workgroups of the solve): " +
      "; ".join(f"{nm} {pbuf[k] / 100.0 / max(1, int(buf[0])):.1f}" for k, nm in enumerate(names) if pbuf[k]) + f"; total {tot / 100.0 / max(1, int(buf[0])):.1f}")
print(f"{shape} B={B} K={K} N={N}: {n} records in {len(launches)} launches of {[len(l) for l in launches]} workgroups; sizes min/mean/max {sizes.min()}/{sizes.mean():.1f}/{sizes.max()}")
for li in sorted({0, len(launches) - 1}):          # the first launch (every coupling active, cold start) and the last (some molecules have converged)
    r = rec[launches[li]]; it_l, sk_l = its[launches[li]], sks[launches[li]]
    t0 = r[:, 4].min()
    start, endt = (r[:, 4] - t0) / 100.0, (r[:, 5] - t0) / 100.0
    hw, xcc = r[:, 2], r[:, 3] & 0xf
    cu = (xcc << 12) | (((hw >> 13) & 7) << 8) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 0xf)      # XCC | SE | SH | CU
    cus = np.unique(cu)
    print(f"launch {li}: {len(r)} workgroups on {len(cus)} CUs of {len(np.unique(xcc))} XCDs; launch length {endt.max():.1f} us; workgroup time mean {np.mean(endt - start):.1f} max {np.max(endt - start):.1f} us; "
          f"late starters (> 5 us): {int((start > 5).sum())}")
    per = []
    for c in cus:
        m = cu == c
        per.append((endt[m].max(), int(m.sum()), float(np.sum(r[m, 1].astype(float) ** 2.5)), sorted(r[m, 1].tolist(), reverse=True), sorted((r[m, 0] >> 3).tolist())))
    per.sort(reverse=True)
    w = np.array([p[2] for p in per]); e = np.array([p[0] for p in per]); cnt = np.array([p[1] for p in per])
    print(f"  workgroups per CU: min {cnt.min()} max {cnt.max()};  per-CU end: mean {e.mean():.1f} min {e.min():.1f} max {e.max():.1f} us;  "
          f"per-CU sum of size^2.5: max / mean {w.max() / w.mean():.2f};  corr(end, load) {np.corrcoef(e, w)[0, 1]:.2f}")
    for p in per[:6]: print(f"    ends {p[0]:6.1f} us  {p[1]} workgroups, sizes {p[3]}  block>>3 {p[4]}")
    for c in [c for c in cus if endt[cu == c].max() == per[0][0]][:1] + [cus[len(cus) // 2]]:      # the records of the CU that ends last and of one other
        m = np.where(cu == c)[0]
        print("    records of one CU: " + "; ".join(f"block>>3 {r[i, 0] >> 3} size {r[i, 1]} its {it_l[i]}/{sk_l[i]} {start[i]:.1f} -> {endt[i]:.1f} us" for i in m[np.argsort(start[m])]))
    print("    ...")
    for p in per[-3:]: print(f"    ends {p[0]:6.1f} us  {p[1]} workgroups, sizes {p[3]}  block>>3 {p[4]}")
    big = np.argsort(-(endt - start))[:5]
    for i in big: print(f"  longest workgroups: size {r[i, 1]} ran {endt[i] - start[i]:.1f} us (start {start[i]:.1f}), with {int((cu == cu[i]).sum()) - 1} others on its CU")
    for q in (10, 30, 50, 70, 90):
        m = np.abs(r[:, 1] - np.percentile(r[:, 1], q)) < 1
        if m.any(): print(f"  size ~{np.percentile(r[:, 1], q):.0f}: mean run {np.mean((endt - start)[m]):.1f} us, iterations {np.mean(it_l[m]):.1f} / {np.mean(sk_l[m]):.1f}")
    print(f"  iterations per coupling: mean {it_l.mean():.1f} min {it_l.min()} max {it_l.max()} (Sinkhorn {sk_l.mean():.1f}); corr(run time, iterations) {np.corrcoef(endt - start, it_l)[0, 1]:.2f}, corr(run time, size) {np.corrcoef(endt - start, r[:, 1])[0, 1]:.2f}")
shutil.rmtree(tmp, ignore_errors=True)
