"""How long does the host take to ENQUEUE one training step (no synchronisation) versus the GPU time of the step?"""
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
flat = FlatGradients(model.parameters()); opt = torch.optim.Adam(flat.params, lr=1e-4, fused=True)
def step():
    flat.zero()
    pred = model(data, cidx, data.batch, num_graphs=b.num_graphs, max_nodes=b.max_nodes)
    torch.nn.functional.mse_loss(pred, y).backward()
    flat.all_reduce_mean(); opt.step()
for _ in range(5): step()
torch.cuda.synchronize()
n = 20
t0 = time.perf_counter()
for _ in range(n): step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"enqueue {1e3*(t1-t0)/n:.3f} ms/step, total {1e3*(t2-t0)/n:.3f} ms/step")
# single isolated step: enqueue time when the queue is empty
torch.cuda.synchronize(); t0 = time.perf_counter(); step(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"isolated step: enqueue {1e3*(t1-t0):.3f} ms, complete {1e3*(t2-t0):.3f} ms")
if os.environ.get("PROFILE"):
    import cProfile, pstats
    pr = cProfile.Profile()
    torch.cuda.synchronize()
    pr.enable()
    for _ in range(10): step()
    pr.disable()
    torch.cuda.synchronize()
    st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(28)
