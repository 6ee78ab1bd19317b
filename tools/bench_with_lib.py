"""bench.py against another build of the library: python tools/bench_with_lib.py <libconan_fgw_hip.so> <bench.py arguments...>
(for profiling -D variants of the kernels inside the whole step; the product never loads anything but its in-tree library)."""
import os, runpy, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from conan_fgw_amd import _lib
_lib._SO = sys.argv[1]
sys.argv = [os.path.join(ROOT, "bench.py")] + sys.argv[2:]
runpy.run_path(sys.argv[0], run_name="__main__")
