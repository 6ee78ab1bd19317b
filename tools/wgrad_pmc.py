"""Workload for PMC passes over the edge-level weight gradient (tools/linear_pmc.sh wgrad): 8 launches of conan_linear_wgrad at 482 k x 128 x 128."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from conan_fgw_amd._lib import lib
from conan_fgw_amd.ops import call, ptr, stream_ptr
dev = torch.device("cuda:0")
torch.manual_seed(0)
M, F = 482_110, 128
g = torch.randn(M, F, device=dev); x = torch.randn(M, F, device=dev); dW = torch.empty(F, F, device=dev); db = torch.empty(F, device=dev)
ws = torch.empty(int(lib().conan_linear_wgrad_ws(M, F, F)), device=dev)
for _ in range(8):
    call("conan_linear_wgrad", ptr(g), ptr(x), M, F, F, None, ptr(dW), ptr(db), ptr(ws), stream_ptr())
torch.cuda.synchronize()
