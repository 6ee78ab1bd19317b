"""Workload for the PMC passes over the edge-level Linear kernel (tools/linear_pmc.sh): 10 launches of conan_linear_fwd at 482 k x 128 x 128."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from conan_fgw_amd.ops import call, ptr, stream_ptr
dev = torch.device("cuda:0")
torch.manual_seed(0)
M, F = 482_110, 128
x = torch.randn(M, F, device=dev); y = torch.empty(M, F, device=dev); w = torch.randn(F, F, device=dev) / 11; b = torch.randn(F, device=dev) / 10
for act in (0, 3):
    for _ in range(5):
        call("conan_linear_fwd", ptr(x), ptr(w), ptr(b), None, M, F, F, 0, act, None, ptr(y), stream_ptr())
torch.cuda.synchronize()
