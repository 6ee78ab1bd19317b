"""In-job A/B of a kernel change on the WHOLE step: like tools/ab.py, but what runs alternately against the in-tree library and the variant
libraries is `bench.py` itself (graph replay), not a probe.

    python tools/ab_bench.py <override_dir | NAME=VALUE>[,...] [rounds] [-- extra bench.py flags]"""
import json, os, shutil, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = sys.argv[1:]
extra = []
if "--" in args:
    extra = args[args.index("--") + 1:]; args = args[:args.index("--")]
ovrs = args[0].split(","); rounds = int(args[1]) if len(args) > 1 else 2
libs, tmps = [("in-tree", "")], []
for o in ovrs:
    tmp = tempfile.mkdtemp(prefix="conan_ab_"); tmps.append(tmp)
    os.makedirs(os.path.join(tmp, "include")); shutil.copy(os.path.join(ROOT, "include", "conan_fgw_hip.h"), os.path.join(tmp, "include"))
    src = os.path.join(tmp, "pkg", "csrc"); shutil.copytree(os.path.join(ROOT, "conan-fgw_amd", "csrc"), src, ignore=shutil.ignore_patterns("*.o"))
    mk = []
    if "=" in o and not os.path.isdir(o):
        mk = ["CXXFLAGS=-O3 -fPIC -std=c++17 --offload-arch=gfx950 -Wno-unused-function -D" + o]
    else:
        for f in os.listdir(o):
            shutil.copy(os.path.join(o, f), os.path.join(src, f))
    subprocess.check_call(["make", "-C", src, "-s", "-j16"] + mk, stderr=subprocess.DEVNULL)
    libs.append((os.path.basename(o.rstrip("/")), os.path.join(tmp, "pkg", "libconan_fgw_hip.so")))
runner = ("import sys, runpy; sys.path.insert(0, %r)\nfrom conan_fgw_amd import _lib\nif %%r: _lib._SO = %%r\n"
          "sys.argv = ['bench.py', '--no-cpu-baseline'] + %r\nrunpy.run_path(%r, run_name='__main__')" % (ROOT, extra, os.path.join(ROOT, "bench.py")))
w = max(len(t) for t, _ in libs)
for rnd in range(rounds):
    for tag, so in libs:
        out = subprocess.run([sys.executable, "-c", runner % (bool(so), so)], capture_output=True, text=True).stdout.strip().splitlines()
        d = json.loads(out[-1])
        print(tag.ljust(w), "step %.4f ms  eager %.4f ms  forward %s ms" % (d["ms_per_step"], d["eager"]["ms_per_step"], (d.get("forward_only") or {}).get("ms_per_step")), flush=True)
for t in tmps:
    shutil.rmtree(t, ignore_errors=True)
