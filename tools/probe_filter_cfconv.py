"""Probe (tools/ab.py or stand-alone): CFConv's edge half at inference on the cfg2 batch (25 k atoms, 518 k directed edges, 259 k pairs) —
the two kernels of the training path (conan_filter_fwd without the h1 output + conan_cfconv_fwd on pair rows) against the fused
conan_filter_cfconv_fwd (filter rows generated per directed edge and consumed from the accumulators), and the whole forward of the
stage-2 model with the fusion on and off.  One line per library."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from conan_fgw_amd import _lib
if len(sys.argv) > 1 and sys.argv[1]:
    _lib._SO = sys.argv[1]
tag = sys.argv[2] if len(sys.argv) > 2 else ""
from conan_fgw_amd import ops, schnet
from conan_fgw_amd.synthetic import make_batch, make_bond_graph
shape, B = (sys.argv[3], int(sys.argv[4])) if len(sys.argv) > 4 else ("esol", 256)
dev = torch.device("cuda:0")
b = make_batch(shape, B, 5, seed=1236)
pos = torch.from_numpy(b.pos).to(dev); batch = torch.from_numpy(b.batch).to(dev)
g = ops.RadiusGraph(pos, ops.graph_ptr_from_batch(batch, b.num_graphs), b.num_graphs, 10.0, 32)
g.pairs()
n, F, Gs = g.num_atoms, 128, 50
torch.manual_seed(0)
offset = torch.linspace(0, 10, Gs, device=dev); coeff = -0.5 / float(offset[1] - offset[0]) ** 2
w1 = torch.randn(F, Gs, device=dev) / 7; b1 = torch.randn(F, device=dev) / 10; w2 = torch.randn(F, F, device=dev) / 11; b2 = torch.randn(F, device=dev) / 10
x = torch.randn(n, F, device=dev)


def timed(fn, reps=30):
    for _ in range(5): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3


with torch.no_grad():
    def two():
        W = ops.filter_generate(g, offset, coeff, w1, b1, w2, b2)
        return ops.cfconv(x, W, g, pre_cutoff_grad=True, use_pairs=True)
    def fused():
        return ops.filter_cfconv(x, g, offset, coeff, w1, b1, w2, b2)
    t_two, t_fused = timed(two), timed(fused)
    err = float((fused().double() - two().double()).norm() / two().double().norm())
    # whole forward of the stage-2 model, fusion on / off
    from conan_fgw_amd.head import EmbeddingsWithGATAggregationBaryCenter
    import types
    bg = make_bond_graph(b, seed=2236)
    torch.manual_seed(5)
    model = EmbeddingsWithGATAggregationBaryCenter(5, dev).to(dev)
    data = types.SimpleNamespace(z=torch.from_numpy(b.z).to(dev), pos=pos, batch=batch, x=torch.from_numpy(bg.x).to(dev),
                                 edge_index=torch.from_numpy(bg.edge_index).to(dev), edge_attr=torch.from_numpy(bg.edge_attr).to(dev))
    cidx = model.create_aggregation_index(b.num_graphs, dev)
    fwd = lambda: model(data, cidx, batch, num_graphs=b.num_graphs, max_nodes=b.max_nodes)
    res = {}
    for rnd in range(2):
        for on in (True, False):
            schnet.FUSE_FILTER_INTO_GATHER = on; schnet.FUSE_MIN_FILTER_BYTES = 0          # (on = fused whatever the size: the A/B decides the threshold)
            res.setdefault(on, []).append(timed(fwd, 20))
    schnet.FUSE_FILTER_INTO_GATHER = True; y1 = fwd()
    schnet.FUSE_FILTER_INTO_GATHER = False; y0 = fwd()
    yerr = float((y1 - y0).abs().max() / y0.abs().max())
print(f"{tag} {shape} B={B} E={g.num_edges}: filter + gather {t_two:6.1f} us   fused {t_fused:6.1f} us   rel diff {err:.1e} | stage-2 forward (eager): "
      f"fused {min(res[True]) / 1e3:.3f} ms  two kernels {min(res[False]) / 1e3:.3f} ms  |y diff| {yerr:.1e}")
