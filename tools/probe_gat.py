"""Probe for tools/ab.py: the covalent branch (GATBased: two GATConv layers + sum pooling) at cfg2 size, forward and forward + backward."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from conan_fgw_amd import _lib
if len(sys.argv) > 1 and sys.argv[1]:
    _lib._SO = sys.argv[1]
tag = sys.argv[2] if len(sys.argv) > 2 else ""
from conan_fgw_amd.gat import GATBased
from conan_fgw_amd.synthetic import make_batch, make_bond_graph
dev = torch.device("cuda:0")
b = make_batch("esol", 256, 5, seed=1236); g = make_bond_graph(b, seed=2236)
torch.manual_seed(3)
m = GATBased(out_channels=64).to(dev)
x = torch.from_numpy(g.x).to(dev) if hasattr(g, "x") else torch.randn(len(b.z), 9, device=dev)
ei = torch.from_numpy(g.edge_index).to(dev); ea = torch.from_numpy(g.edge_attr).to(dev); batch = torch.from_numpy(b.batch).to(dev)
def fwd(): return m(x, ei, ea, batch, num_graphs=b.num_graphs)
def fb():
    for p in m.parameters(): p.grad = None
    fwd().square().sum().backward()
def timed(fn, reps=20):
    for _ in range(3): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3
gr = torch.cuda.CUDAGraph()
fb(); torch.cuda.synchronize()
side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    fb()
    with torch.cuda.graph(gr): fb()
torch.cuda.current_stream().wait_stream(side)
t = timed(gr.replay)
grads = torch.cat([p.grad.reshape(-1) for p in m.parameters()])
print(f"{tag} GAT branch fwd+bwd (graph replay, {len(b.z)} atoms, {ei.shape[1]} bonds): {t:6.1f} us   grad checksum {float(grads.double().abs().sum()):.9e}")
