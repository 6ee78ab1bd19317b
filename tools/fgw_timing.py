"""Ablation timing of the batched FGW kernel at the cfg2 shape (B=256, K=5, N=33, d=64) on the GPU."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from conan_fgw_amd import ops
dev = torch.device("cuda:0")
B, K, N, d = 256, 5, 33, 64
g = torch.Generator().manual_seed(0)
Ys = (torch.rand(B, K, N, d, generator=g) * 1.9 + 0.1).to(dev)
A = (torch.rand(B, K, N, N, generator=g) < 0.5).float(); Cs = torch.triu(A, 1); Cs = (Cs + Cs.transpose(-1, -2)).to(dev)

def t(label, **kw):
    for _ in range(2): out = ops.fgw_barycenter_batched(Ys, Cs, **kw)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): out = ops.fgw_barycenter_batched(Ys, Cs, **kw)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
    info = out[3].float().mean(0).tolist()
    print(f"{label:40s} {dt*1e3:8.3f} ms   mean(outer,pgd,sk)={info[:3]}")

t("production")
t("outer=1", max_iter=1)
t("sinkhorn iters=1", num_iter_max=1)
t("pgd tol huge (1 pgd iter)", inner_tol=1e9)
t("sinkhorn=1,pgd=1", num_iter_max=1, inner_tol=1e9)
t("tol huge (1 outer)", tol=1e9)
