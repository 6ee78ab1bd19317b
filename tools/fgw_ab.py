"""A/B of the FGW solve: the in-tree library vs a library built from a directory of source overrides (default tools/_ab_csrc, scratch: put the variant .hip/.h files there)."""
import os, shutil, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OVR = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "tools", "_ab_csrc")
sys.path.insert(0, ROOT)
tmp = tempfile.mkdtemp(prefix="conan_ab_")
os.makedirs(os.path.join(tmp, "include")); shutil.copy(os.path.join(ROOT, "include", "conan_fgw_hip.h"), os.path.join(tmp, "include"))
src = os.path.join(tmp, "pkg", "csrc"); shutil.copytree(os.path.join(ROOT, "conan-fgw_amd", "csrc"), src, ignore=shutil.ignore_patterns("*.o"))
for f in os.listdir(OVR):
    shutil.copy(os.path.join(OVR, f), os.path.join(src, f))
subprocess.check_call(["make", "-C", src, "-s", "-j16"])
code = r'''
import sys, time, torch
sys.path.insert(0, %r)
from conan_fgw_amd import _lib
if %r: _lib._SO = %r
from conan_fgw_amd import ops
dev = torch.device("cuda:0")
B, K, N, d = 256, 5, 33, 64
g = torch.Generator().manual_seed(0)
Ys = (torch.rand(B, K, N, d, generator=g) * 1.9 + 0.1).to(dev)
A = (torch.rand(B, K, N, N, generator=g) < 0.5).float(); Cs = torch.triu(A, 1); Cs = (Cs + Cs.transpose(-1, -2)).to(dev)
for _ in range(5): ops.fgw_barycenter_batched(Ys, Cs)
torch.cuda.synchronize()
ts = []
for rep in range(5):
    t0 = time.perf_counter()
    for _ in range(20): ops.fgw_barycenter_batched(Ys, Cs)
    torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / 20 * 1e3)
print(%r, " ".join("%%.4f" %% t for t in ts), "ms")
'''
for rnd in range(2):
    for tag, so in (("in-tree ", ""), ("override", os.path.join(tmp, "pkg", "libconan_fgw_hip.so"))):
        subprocess.check_call([sys.executable, "-c", code % (ROOT, bool(so), so, tag)])
shutil.rmtree(tmp, ignore_errors=True)
