"""Runs the production FGW solve a few times (for rocprofv3 --pmc passes)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from conan_fgw_amd import ops
dev = torch.device("cuda:0")
B, K, N, d = 256, 5, 33, 64
g = torch.Generator().manual_seed(0)
Ys = (torch.rand(B, K, N, d, generator=g) * 1.9 + 0.1).to(dev)
if os.environ.get("PROBE_RANDOM"):      # random structures: the general path of k_fgw_coupling_fast
    A = (torch.rand(B, K, N, N, generator=g) < 0.5).float(); Cs = torch.triu(A, 1); Cs = (Cs + Cs.transpose(-1, -2)).to(dev)
else:                                   # complete graphs on n real nodes, padded nodes isolated: what the ESOL-shaped batch of cfg2 is (round 6: the row-sum form)
    n = torch.randint(6, N + 1, (B,), generator=g); n[0] = N
    real = (torch.arange(N)[None, :] < n[:, None]).float()
    Cs = (real[:, :, None] * real[:, None, :] * (1.0 - torch.eye(N)))[:, None].expand(B, K, N, N).contiguous().to(dev)
    Ys = torch.where(real.to(dev)[:, None, :, None] > 0, Ys, torch.full_like(Ys, 0.5))      # padded nodes: one common feature row, as the model's glue leaves them (solved as one node)
for _ in range(3):
    ops.fgw_barycenter_batched(Ys, Cs, cs_small_int=True)
torch.cuda.synchronize()
