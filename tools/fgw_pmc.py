"""Runs the production FGW solve a few times (for rocprofv3 --pmc passes)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from conan_fgw_amd import ops
dev = torch.device("cuda:0")
B, K, N, d = 256, 5, 33, 64
g = torch.Generator().manual_seed(0)
Ys = (torch.rand(B, K, N, d, generator=g) * 1.9 + 0.1).to(dev)
A = (torch.rand(B, K, N, N, generator=g) < 0.5).float(); Cs = torch.triu(A, 1); Cs = (Cs + Cs.transpose(-1, -2)).to(dev)
for _ in range(3):
    ops.fgw_barycenter_batched(Ys, Cs, cs_small_int=True)
torch.cuda.synchronize()
