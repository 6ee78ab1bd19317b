"""Probe for tools/ab.py (or stand-alone): the cfg2 training step (stage-2 model, eager, side stream) timed in blocks of 20 steps.
argv[1] = library override ("" = in-tree), argv[2] = tag, argv[3] (optional) = comma list of C entry points whose
`*_supported` query is forced to 0 in alternating blocks (in-process A/B of a fused path against its composed fallback)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from conan_fgw_amd import _lib
if len(sys.argv) > 1 and sys.argv[1]:
    _lib._SO = sys.argv[1]
tag = sys.argv[2] if len(sys.argv) > 2 else ""
off = [s for s in (sys.argv[3].split(",") if len(sys.argv) > 3 else []) if s]
from conan_fgw_amd import ops
from conan_fgw_amd.head import EmbeddingsWithGATAggregationBaryCenter
from conan_fgw_amd.parallel import FlatGradients
from conan_fgw_amd.synthetic import make_batch, make_bond_graph
from conan_fgw_amd.collate import DeviceCollator, molecules_from_synthetic
dev = torch.device("cuda:0")
shape, B, K, model_name = os.environ.get("PROBE_SHAPE", "esol"), int(os.environ.get("PROBE_BATCH", 256)), int(os.environ.get("PROBE_K", 5)), os.environ.get("PROBE_MODEL", "schnet")
b = make_batch(shape, B, K, seed=1236); bg = make_bond_graph(b, seed=2236)
data = DeviceCollator(dev, K, depth=2, static=True)(molecules_from_synthetic(b, bg)).wait()
y = torch.from_numpy(b.y).to(dev)[:, None]
torch.manual_seed(5)
model = EmbeddingsWithGATAggregationBaryCenter(K, dev, model_name=model_name).to(dev)
cidx = model.create_aggregation_index(b.num_graphs, dev)
flat = FlatGradients(model.parameters())
opt = torch.optim.Adam(flat.params, lr=1e-4, fused=True, capturable=True)


def step():
    flat.zero()
    loss = torch.nn.functional.mse_loss(model(data, cidx, data.batch, num_graphs=b.num_graphs, max_nodes=b.max_nodes), y)
    flat.backward(loss)
    flat.all_reduce_mean()
    opt.step()
    return loss


def block(n=20):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): step()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3


L = _lib.lib()
py_off = [o for o in off if o == "late_stage1"]            # python-level switch: node-level slab kernels immediate instead of batched
off = [o for o in off if o not in py_off]
saved = {name: getattr(L, name) for name in off}
late_rows = ops._LATE_STAGE1_ROWS
side = torch.cuda.Stream(device=dev); side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(5): step()
    res = {"on": [], "off": []}
    for rep in range(4):
        res["on"].append(block())
        if off or py_off:
            for name in off: setattr(L, name, lambda *a: 0)
            if py_off: ops._LATE_STAGE1_ROWS = 0
            block(3); res["off"].append(block())
            for name in off: setattr(L, name, saved[name])
            ops._LATE_STAGE1_ROWS = late_rows
            block(3)
print(tag, "step ms:", " ".join("%.3f" % t for t in res["on"]), ("| with %s off: " % ",".join(off + py_off) + " ".join("%.3f" % t for t in res["off"])) if (off or py_off) else "")
