"""Probe (tools/ab.py): the captured BACE B = 64 ViSNet training step (forward + backward + gradient pack | Adam) against the library in
argv[1] ("" = in-tree): one line, ms per step over 3 blocks of 20 replays.

    python tools/ab.py NAME=VALUE tools/probe_visnet_step.py 3"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from conan_fgw_amd import _lib
if len(sys.argv) > 1 and sys.argv[1]:
    _lib._SO = sys.argv[1]
tag = sys.argv[2] if len(sys.argv) > 2 else ""
shape, B, K, model_name = (sys.argv[3], int(sys.argv[4]), int(sys.argv[5]), sys.argv[6]) if len(sys.argv) > 6 else ("bace", 64, 5, "visnet")
from conan_fgw_amd import ops
from conan_fgw_amd.collate import DeviceCollator, molecules_from_synthetic
from conan_fgw_amd.head import EmbeddingsWithGATAggregationBaryCenter
from conan_fgw_amd.parallel import FlatAdam, FlatGradients
from conan_fgw_amd.synthetic import make_batch, make_bond_graph
dev = torch.device("cuda:0")
b = make_batch(shape, B, K, seed=1236); bg = make_bond_graph(b, seed=2236)
data = DeviceCollator(dev, K, depth=2, static=True)(molecules_from_synthetic(b, bg)).wait()
y = torch.from_numpy(b.y).to(dev)[:, None]
torch.manual_seed(5)
model = EmbeddingsWithGATAggregationBaryCenter(K, dev, model_name=model_name).to(dev)
cidx = model.create_aggregation_index(b.num_graphs, dev)
flat = FlatGradients(model.parameters())
opt = FlatAdam(flat, lr=1e-4)
seed = torch.ones((), device=dev)


def fwd_bwd():
    flat.zero()
    loss = ops.mse_loss(model(data, cidx, data.batch, num_graphs=b.num_graphs, max_nodes=b.max_nodes), y)
    flat.backward(loss, grad_scale=seed)


side = torch.cuda.Stream(device=dev); side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(3):
        fwd_bwd(); flat.pack(); opt.step()
    torch.cuda.synchronize()
    gA = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gA, stream=side, capture_error_mode="thread_local"):
        fwd_bwd(); flat.pack()
    gB = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gB, stream=side, pool=gA.pool(), capture_error_mode="thread_local"):
        opt.step()
    ts = []
    for _ in range(3):
        for _ in range(3): gA.replay(); gB.replay()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): gA.replay(); gB.replay()
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / 20 * 1e3)
print(f"{tag} {model_name} {shape} B={B} K={K}: " + "  ".join(f"{t:.3f}" for t in ts) + " ms per step", flush=True)
