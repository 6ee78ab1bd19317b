"""Where the time of k_fgw_coupling_small goes: builds a PRIVATE copy of the library with -DCONAN_FGW_PROFILE (wall-clock marks at
the phase boundaries, thread 0 of every workgroup), runs the cfg2-shaped solve and prints microseconds per phase and launch.
The product library is untouched (it is compiled without the macro).  Run on the GPU box: python tools/fgw_phase_profile.py"""
import ctypes, os, shutil, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
tmp = tempfile.mkdtemp(prefix="conan_prof_")
src = os.path.join(tmp, "csrc"); shutil.copytree(os.path.join(ROOT, "conan-fgw_amd", "csrc"), src, ignore=shutil.ignore_patterns("*.o"))
os.makedirs(os.path.join(tmp, "include")); shutil.copy(os.path.join(ROOT, "include", "conan_fgw_hip.h"), os.path.join(tmp, "include"))
# csrc/common.h includes ../../include/...: recreate that relative layout
os.makedirs(os.path.join(tmp, "pkg")); shutil.move(src, os.path.join(tmp, "pkg", "csrc")); src = os.path.join(tmp, "pkg", "csrc")
subprocess.check_call(["make", "-C", src, "-s", "-j16", "CXXFLAGS=-O3 -fPIC -std=c++17 --offload-arch=gfx950 -Wno-unused-function -DCONAN_FGW_PROFILE " + os.environ.get("PROF_DEFS", "")])      # PROF_DEFS: extra -D switches (A/B of a phase)
import torch
from conan_fgw_amd import _lib
_lib._SO = os.path.join(tmp, "pkg", "libconan_fgw_hip.so")
from conan_fgw_amd import ops
L = _lib.lib()
for _n in ("conan_debug_fgw_prof", "conan_debug_fgw_prof_large"):
    getattr(L, _n).restype = ctypes.c_int
    getattr(L, _n).argtypes = [ctypes.POINTER(ctypes.c_longlong), ctypes.c_int]
dev = torch.device("cuda:0")
B, K, N, d = int(os.environ.get("PROF_B", 256)), 5, int(os.environ.get("PROF_N", 33)), 64
g = torch.Generator().manual_seed(0)
Ys = (torch.rand(B, K, N, d, generator=g) * 1.9 + 0.1).to(dev)
print(f"B={B} K={K} N={N} d={d}  ({'register-resident kernel, N <= 64' if N <= 64 else 'large-N kernel'})")
if os.environ.get("PROF_COMPLETE"):      # complete graphs on n real nodes (the cfg2 batch's shape: the row-sum form of k_fgw_coupling_fast)
    n = torch.randint(6, N + 1, (B,), generator=g); n[0] = N
    real = (torch.arange(N)[None, :] < n[:, None]).float()
    Cs = (real[:, :, None] * real[:, None, :] * (1.0 - torch.eye(N)))[:, None].expand(B, K, N, N).contiguous().to(dev)
else:
    A = (torch.rand(B, K, N, N, generator=g) < 0.5).float(); Cs = torch.triu(A, 1); Cs = (Cs + Cs.transpose(-1, -2)).to(dev)
SMALL_INT = os.environ.get("PROF_FLOAT_CS", "") == ""
for _ in range(2): ops.fgw_barycenter_batched(Ys, Cs, cs_small_int=SMALL_INT)
torch.cuda.synchronize()
buf = (ctypes.c_longlong * 32)()
PROF = L.conan_debug_fgw_prof if N <= 64 else L.conan_debug_fgw_prof_large
PROF(buf, 1)
reps = 5
for _ in range(reps): out = ops.fgw_barycenter_batched(Ys, Cs, cs_small_int=SMALL_INT)
torch.cuda.synchronize()
PROF(buf, 0)
names = ["staging", "T0 + dot(Y,Z)", "base registers", "A = C1 @ T (product)", "G = A @ 2C2^T", "max |A| + digits of A + barrier (large-N kernel)", "Sinkhorn iterations",
         "T store + err", "T -> global", "Ypart = T @ Z", "Cpart = T C2 T^T"]
if N <= 64 and not os.environ.get("PROF_OLD"):      # round-3 kernel (k_fgw_coupling_fast): its marks
    names = ["staging", "dot(Y,Z) + base", "T0", "A = C1 @ T", "G = A @ 2C2^T -> K = exp(Mr - ref)", "K -> registers", "Sinkhorn iterations",
             "T store + err", "T -> global", "Ypart = T @ Z", "Cpart = T C2 T^T"]
launches = reps * 5
wgs = B * K
tot = sum(buf[:11])
print(f"info mean (outer, pgd, sinkhorn): {out[3].float().mean(0).tolist()[:3]}")
print(f"{'phase':40s} {'us / workgroup / launch':>24s} {'share':>7s}")
for k, n in enumerate(names):
    print(f"{n:40s} {buf[k] / 100.0 / (launches * wgs):24.2f} {100.0 * buf[k] / tot:6.1f}%")
print(f"{'total':40s} {tot / 100.0 / (launches * wgs):24.2f}")
if buf[21]:
    print(f"shader clock during the kernel: {100.0 * buf[20] / buf[21]:.0f} MHz (clock64 ticks per 100 MHz wall-clock tick)")
shutil.rmtree(tmp, ignore_errors=True)
