"""In-process A/B of the ORDER of the molecules inside a batch on the whole captured training step (same molecules, same model): the order make_batch
drew them in against sizes dealt over the XCDs / CUs (sorted by atoms, rank 8 j + x at position 32 x + j of an eighth of the batch — under the coupling
kernel's XCD-contiguous numbering every CU then receives one coupling from each size quintile).  python tools/probe_batch_order.py [shape batch conformers rounds]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
shape = sys.argv[1] if len(sys.argv) > 1 else "esol"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
K = int(sys.argv[3]) if len(sys.argv) > 3 else 5
rounds = int(sys.argv[4]) if len(sys.argv) > 4 else 4
from conan_fgw_amd import ops
from conan_fgw_amd.collate import DeviceCollator, molecules_from_synthetic
from conan_fgw_amd.head import EmbeddingsWithGATAggregationBaryCenter
from conan_fgw_amd.parallel import FlatAdam, FlatGradients
from conan_fgw_amd.synthetic import make_batch, make_bond_graph
dev = torch.device("cuda:0")
b = make_batch(shape, B, K, seed=1236); bg = make_bond_graph(b, seed=2236)
items = molecules_from_synthetic(b, bg)
sizes = np.array([len(it.z) for it in items])
rank = np.argsort(-sizes, kind="stable")
per = B // 8
dealt = [items[rank[8 * j + x]] for x in range(8) for j in range(per)] if B % 8 == 0 else items
orders = {"as drawn": items, "dealt by size": dealt, "sorted descending": [items[r] for r in rank]}
torch.manual_seed(5)
model = EmbeddingsWithGATAggregationBaryCenter(K, dev).to(dev)
flat = FlatGradients(model.parameters()); opt = FlatAdam(flat, lr=1e-4)
seed = torch.ones((), device=dev)
side = torch.cuda.Stream(device=dev); side.wait_stream(torch.cuda.current_stream())
graphs = {}
with torch.cuda.stream(side):
    for tag, its in orders.items():
        data = DeviceCollator(dev, K, depth=2, static=True)(its).wait()
        y = data.y[::K][:, None].contiguous()
        cidx = model.create_aggregation_index(data.num_graphs, dev)
        def fwd_bwd(data=data, y=y, cidx=cidx):
            flat.zero()
            loss = ops.mse_loss(model(data, cidx, data.batch, num_graphs=data.num_graphs, max_nodes=data.max_nodes), y)
            flat.backward(loss, grad_scale=seed)
        for _ in range(3): fwd_bwd(); flat.pack(); opt.step()
        torch.cuda.synchronize()
        gA = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gA, stream=side, capture_error_mode="thread_local"):
            fwd_bwd(); flat.pack()
        gB = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gB, stream=side, pool=gA.pool(), capture_error_mode="thread_local"):
            opt.step()
        graphs[tag] = (gA, gB, data)
    def block(tag, n=20):
        gA, gB, _ = graphs[tag]
        for _ in range(3): gA.replay(); gB.replay()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): gA.replay(); gB.replay()
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
    tags = list(orders)
    for r in range(rounds):
        order = tags if r % 2 == 0 else tags[::-1]
        print("  ".join(f"{t}: {block(t):.4f} ms" for t in order), flush=True)
