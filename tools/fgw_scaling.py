"""FGW solve time versus the number of molecules (K=5, N=33, d=64): is the kernel throughput-bound (time ~ work) or bound by
rounds of resident workgroups (768 slots = 3 per CU)?"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from conan_fgw_amd import ops
dev = torch.device("cuda:0")
K, N, d = 5, 33, 64
for B in (51, 102, 153, 154, 205, 256, 307, 384, 512):
    g = torch.Generator().manual_seed(0)
    Ys = (torch.rand(B, K, N, d, generator=g) * 1.9 + 0.1).to(dev)
    A = (torch.rand(B, K, N, N, generator=g) < 0.5).float(); Cs = torch.triu(A, 1); Cs = (Cs + Cs.transpose(-1, -2)).to(dev)
    for _ in range(2): ops.fgw_barycenter_batched(Ys, Cs)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): ops.fgw_barycenter_batched(Ys, Cs)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
    print(f"B={B:4d}  workgroups={B*K:5d}  rounds={B*K/768:5.2f}  {dt*1e3:7.3f} ms   {dt*1e6/B:6.2f} us/molecule")
