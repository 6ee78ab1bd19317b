"""Where the time of k_filter_bwd2 goes: builds a PRIVATE copy of the library with -DCONAN_F2_PROFILE (shader-clock marks at the phase boundaries,
lane 0 of every wavefront), runs the cfg2-sized launch and prints cycles per tile and wavefront role.  The product library is untouched."""
import ctypes, os, shutil, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
tmp = tempfile.mkdtemp(prefix="conan_prof_")
os.makedirs(os.path.join(tmp, "include")); shutil.copy(os.path.join(ROOT, "include", "conan_fgw_hip.h"), os.path.join(tmp, "include"))
src = os.path.join(tmp, "pkg", "csrc"); shutil.copytree(os.path.join(ROOT, "conan-fgw_amd", "csrc"), src, ignore=shutil.ignore_patterns("*.o"))
subprocess.check_call(["make", "-C", src, "-s", "-j16", "CXXFLAGS=-O3 -fPIC -std=c++17 --offload-arch=gfx950 -Wno-unused-function -DCONAN_F2_PROFILE" + (" " + os.environ["F2_EXTRA"] if os.environ.get("F2_EXTRA") else "")])
import torch
from conan_fgw_amd import _lib
_lib._SO = os.path.join(tmp, "pkg", "libconan_fgw_hip.so")
from conan_fgw_amd._lib import call, lib, ptr, stream_ptr
L = lib()
L.conan_debug_f2_prof.restype = ctypes.c_int
L.conan_debug_f2_prof.argtypes = [ctypes.POINTER(ctypes.c_longlong), ctypes.c_int]
dev = torch.device("cuda:0")
M, F, Gs = int(sys.argv[1]) if len(sys.argv) > 1 else 259048, 128, 50
gen = torch.Generator().manual_seed(1)
g = torch.randn(M, F, generator=gen).to(dev); h1 = (torch.rand(M, F, generator=gen) * 3 - 0.6).to(dev)
dist = (torch.rand(M, generator=gen) * 10).to(dev); w2 = (torch.randn(F, F, generator=gen) / 11).to(dev)
off = torch.linspace(0, 10, Gs).to(dev); coeff = -0.5 / float(off[1] - off[0]) ** 2
md = torch.tensor([M], dtype=torch.int32, device=dev); gmax = g.abs().max().reshape(1).contiguous()
ws2 = torch.empty(int(L.conan_filter_bwd2_ws(M, Gs, F)), device=dev)
def fused():
    call("conan_filter_bwd2", ptr(g), ptr(h1), ptr(dist), M, ptr(off), Gs, coeff, ptr(w2), F, ptr(md), ptr(gmax), None, None, None, None, ptr(ws2), stream_ptr())
for _ in range(3): fused()
torch.cuda.synchronize()
buf = (ctypes.c_longlong * 16)()
L.conan_debug_f2_prof(buf, 1)
reps = 10
for _ in range(reps): fused()
torch.cuda.synchronize()
s_, e_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s_.record()
for _ in range(reps): fused()
e_.record(); torch.cuda.synchronize()
print(f"[{os.environ.get('F2_EXTRA', 'as built')}] {1e3 * s_.elapsed_time(e_) / reps:.1f} us per launch")
L.conan_debug_f2_prof(buf, 0)
tiles = (M + 31) // 32
names = ["loop", "B staging", "A dx strip", "A h1 / ssp'", "A dw1 | B dw2 products", "A staging", "barrier", "-"]
for role, base in (("A (dx, dh1, dw1)", 0), ("B (dw2, rbf)", 8)):
    tot = sum(buf[base:base + 8])
    print(f"{role}: shader-clock cycles per tile and wavefront (4 wavefronts per role and tile)")
    for k, n in enumerate(names[:7]):
        print(f"    {n:28s} {buf[base + k] / (2 * reps * tiles * 4):10.0f}  {100.0 * buf[base + k] / max(tot, 1):5.1f}%")
    print(f"    {'total':28s} {tot / (2 * reps * tiles * 4):10.0f}")
shutil.rmtree(tmp, ignore_errors=True)
