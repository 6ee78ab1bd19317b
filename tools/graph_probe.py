"""Capture one whole training step (forward, backward, gradient pack, fused Adam) of the stage-2 model into a HIP graph and
compare replay time with eager execution.

Note: every eager step before the capture runs on a side stream (the PyTorch recipe).  Capturing after steps that ran on the
legacy default stream leaves AccumulateGrad nodes bound to that stream and capture_end crashes — which is why bench.py does not
do this in-process after its eager measurement."""
import os, sys, time, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from conan_fgw_amd.head import EmbeddingsWithGATAggregationBaryCenter
from conan_fgw_amd.parallel import FlatGradients
from conan_fgw_amd.synthetic import make_batch, make_bond_graph
dev = torch.device("cuda:0")
K = 5
b = make_batch("esol", 256, K, seed=1236); bg = make_bond_graph(b, seed=2236)
t = lambda a: torch.from_numpy(a).to(dev)
data = types.SimpleNamespace(z=t(b.z), pos=t(b.pos), batch=t(b.batch), x=t(bg.x), edge_index=t(bg.edge_index), edge_attr=t(bg.edge_attr))
y = t(b.y)[:, None]
torch.manual_seed(5)
model = EmbeddingsWithGATAggregationBaryCenter(K, dev).to(dev)
cidx = model.create_aggregation_index(b.num_graphs, dev)
flat = FlatGradients(model.parameters()); opt = torch.optim.Adam(flat.params, lr=1e-4, fused=True, capturable=True)
loss_out = torch.zeros((), device=dev)
def step():
    flat.zero()
    pred = model(data, cidx, data.batch, num_graphs=b.num_graphs, max_nodes=b.max_nodes)
    loss = torch.nn.functional.mse_loss(pred, y)
    loss.backward()
    flat.all_reduce_mean(); opt.step()
    loss_out.copy_(loss.detach())
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(5): step()
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
n = 20
t0 = time.perf_counter()
for _ in range(n): step()
torch.cuda.synchronize()
print(f"eager  {1e3*(time.perf_counter()-t0)/n:.3f} ms/step  loss {float(loss_out):.6f}")
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    step()
torch.cuda.synchronize()
for _ in range(3): g.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(n): g.replay()
torch.cuda.synchronize()
print(f"graph  {1e3*(time.perf_counter()-t0)/n:.3f} ms/step  loss {float(loss_out):.6f}")
