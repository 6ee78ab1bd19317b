"""In-PROCESS A/B of a Python-level switch on the whole captured training step: the step (forward + backward + gradient pack | Adam) is captured
into HIP graphs once per setting of `module.ATTRIBUTE`, and the two sets of graphs are replayed alternately in the same process (blocks of
20 steps, order reversed every round) — no first-process bias, no box-to-box spread.

    python tools/ab_step_switch.py schnet.HEADS_ON_TWO_STREAMS=True,False [shape batch conformers [rounds [schnet|visnet]]]"""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
spec = sys.argv[1]
target, vals = spec.split("=")
modname, attr = target.rsplit(".", 1)
mod = importlib.import_module("conan_fgw_amd." + modname)
values = [eval(v) for v in vals.split(",")]
shape = sys.argv[2] if len(sys.argv) > 2 else "esol"
B = int(sys.argv[3]) if len(sys.argv) > 3 else 256
K = int(sys.argv[4]) if len(sys.argv) > 4 else 5
rounds = int(sys.argv[5]) if len(sys.argv) > 5 else 4
model_name = sys.argv[6] if len(sys.argv) > 6 else "schnet"
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
losses = {}


def fwd_bwd(tag):
    flat.zero()
    loss = ops.mse_loss(model(data, cidx, data.batch, num_graphs=b.num_graphs, max_nodes=b.max_nodes), y)
    flat.backward(loss, grad_scale=seed)
    losses[tag] = loss.detach()


side = torch.cuda.Stream(device=dev); side.wait_stream(torch.cuda.current_stream())
graphs = {}
with torch.cuda.stream(side):
    for v in values:
        setattr(mod, attr, v)
        for _ in range(3):
            fwd_bwd(v); flat.pack(); opt.step()
        torch.cuda.synchronize()
        gA = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gA, stream=side, capture_error_mode="thread_local"):
            fwd_bwd(v); flat.pack()
        gB = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gB, stream=side, pool=gA.pool(), capture_error_mode="thread_local"):
            opt.step()
        graphs[v] = (gA, gB)

    def block(v, n=20):
        gA, gB = graphs[v]
        for _ in range(3): gA.replay(); gB.replay()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): gA.replay(); gB.replay()
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3

    for r in range(rounds):
        order = values if r % 2 == 0 else values[::-1]
        print("  ".join(f"{attr}={v}: {block(v):.4f} ms" for v in order) + f"   (losses {', '.join('%.6f' % float(losses[v]) for v in values)})", flush=True)
