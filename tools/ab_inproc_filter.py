"""In-PROCESS A/B of conan_filter_fwd between the in-tree library and a variant built here (an override directory or a -DNAME=VALUE switch, as
tools/ab.py): both libraries are loaded into one process and timed alternately on the same buffers, so neither sits first after an idle
period (tools/ab.py runs one process per library and the first one measured 5-10 % slow on the W-only timing whichever library it was).
    python tools/ab_inproc_filter.py <override_dir | NAME=VALUE> [rounds]"""
import ctypes, os, shutil, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
o = sys.argv[1]; rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 4
tmp = tempfile.mkdtemp(prefix="conan_ab_")
os.makedirs(os.path.join(tmp, "include")); shutil.copy(os.path.join(ROOT, "include", "conan_fgw_hip.h"), os.path.join(tmp, "include"))
src = os.path.join(tmp, "pkg", "csrc"); shutil.copytree(os.path.join(ROOT, "conan-fgw_amd", "csrc"), src, ignore=shutil.ignore_patterns("*.o"))
extra = []
if "=" in o and not os.path.isdir(o):
    extra = ["CXXFLAGS=-O3 -fPIC -std=c++17 --offload-arch=gfx950 -Wno-unused-function -D" + o]
else:
    for f in os.listdir(o):
        shutil.copy(os.path.join(o, f), os.path.join(src, f))
subprocess.check_call(["make", "-C", src, "-s", "-j16"] + extra, stderr=subprocess.DEVNULL)
libs = {"in-tree": ctypes.CDLL(os.path.join(ROOT, "conan-fgw_amd", "libconan_fgw_hip.so")), os.path.basename(o.rstrip("/")): ctypes.CDLL(os.path.join(tmp, "pkg", "libconan_fgw_hip.so"))}
P_, I, Fl = ctypes.c_void_p, ctypes.c_int, ctypes.c_float
for L in libs.values():
    L.conan_filter_fwd.argtypes = [P_, P_, I, P_, I, Fl, Fl, I, P_, P_, P_, P_, P_, P_, P_]; L.conan_filter_fwd.restype = I
dev = torch.device("cuda:0")
Pn, F, Gs = 259_000, 128, 50
torch.manual_seed(0)
dist = torch.rand(Pn, device=dev) * 10; offset = torch.linspace(0, 10, Gs, device=dev); coeff = -0.5 / float(offset[1] - offset[0]) ** 2
w1 = torch.randn(F, Gs, device=dev) / 7; b1 = torch.randn(F, device=dev) / 10; w2 = torch.randn(F, F, device=dev) / 11; b2 = torch.randn(F, device=dev) / 10
cnt = torch.tensor([Pn], dtype=torch.int32, device=dev)
W = torch.empty(Pn, F, device=dev); h1 = torch.empty(Pn, F, device=dev)
s = torch.cuda.current_stream().cuda_stream
def run(L, with_h1):
    rc = L.conan_filter_fwd(dist.data_ptr(), cnt.data_ptr(), Pn, offset.data_ptr(), Gs, coeff, 10.0, F, w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), b2.data_ptr(),
                            W.data_ptr(), h1.data_ptr() if with_h1 else None, s)
    assert rc == 0
def timed(fn, reps=20):
    for _ in range(3): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
for L in libs.values():
    for _ in range(5): run(L, True); run(L, False)
for r in range(rounds):
    order = list(libs.items()) if r % 2 == 0 else list(libs.items())[::-1]
    print("  ".join(f"{name}: with h1 {timed(lambda: run(L, True)):6.1f} us  W only {timed(lambda: run(L, False)):6.1f} us" for name, L in order), flush=True)
shutil.rmtree(tmp, ignore_errors=True)
