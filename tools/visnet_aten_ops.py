"""Which aten ops (adds of gradient accumulation, copies, clones) a ViSNet training step still launches, by shape — torch.profiler over 2 eager steps."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from conan_fgw_amd.head import EmbeddingsWithGATAggregationBaryCenter
from conan_fgw_amd.parallel import FlatGradients
from conan_fgw_amd.synthetic import make_batch, make_bond_graph
from conan_fgw_amd.collate import DeviceCollator, molecules_from_synthetic
dev = torch.device("cuda:0")
shape, B, K, model_name = os.environ.get("PROBE_SHAPE", "bace"), int(os.environ.get("PROBE_BATCH", 64)), 5, os.environ.get("PROBE_MODEL", "visnet")
b = make_batch(shape, B, K, seed=1236); bg = make_bond_graph(b, seed=2236)
data = DeviceCollator(dev, K, depth=2, static=True)(molecules_from_synthetic(b, bg)).wait()
y = torch.from_numpy(b.y).to(dev)[:, None]
torch.manual_seed(5)
model = EmbeddingsWithGATAggregationBaryCenter(K, dev, model_name=model_name).to(dev)
cidx = model.create_aggregation_index(b.num_graphs, dev)
flat = FlatGradients(model.parameters())
def step():
    flat.zero()
    loss = torch.nn.functional.mse_loss(model(data, cidx, data.batch, num_graphs=b.num_graphs, max_nodes=b.max_nodes), y)
    flat.backward(loss)
for _ in range(2): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=False) as prof:
    step(); torch.cuda.synchronize()
rows = [e for e in prof.key_averages(group_by_input_shape=True) if e.key.startswith("aten::") and e.device_time_total > 0]
rows.sort(key=lambda e: -e.device_time_total)
for e in rows[:30]:
    print(f"{e.key:28s} n={e.count:4d} dev {e.device_time_total / 1e3:8.2f} ms  shapes {str(e.input_shapes)[:110]}")
