"""Probe for tools/ab.py: the three CFConv kernels at cfg2 shape, HIP-event timed over 20 launches each (filter tensor warm)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from conan_fgw_amd import _lib
if len(sys.argv) > 1 and sys.argv[1]:
    _lib._SO = sys.argv[1]
tag = sys.argv[2] if len(sys.argv) > 2 else ""
from conan_fgw_amd import ops
from conan_fgw_amd.synthetic import make_batch
dev = torch.device("cuda:0")
b = make_batch("esol", 256, 5, seed=1236)
pos = torch.from_numpy(b.pos).to(dev); batch = torch.from_numpy(b.batch).to(dev)
gp = ops.graph_ptr_from_batch(batch, b.num_graphs)
g = ops.RadiusGraph(pos, gp, b.num_graphs, 10.0, 32).pairs()
n, F = len(b.z), 128
x = torch.randn(n, F, device=dev, requires_grad=True); W = torch.randn(g.max_edges, F, device=dev, requires_grad=True)
gy = torch.randn(n, F, device=dev)
def timed(fn, reps=20):
    for _ in range(3): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3
out = torch.empty(n, F, device=dev)
def fwd(): ops.call("conan_cfconv_fwd", ops.ptr(x), ops.ptr(W), ops.ptr(g.rowptr), ops.ptr(g.col), ops.ptr(g.pid), n, F, ops.ptr(out), None, ops.stream_ptr())
tr, te = g.transpose(); dx = torch.empty(n, F, device=dev); dW = torch.empty(g.max_edges, F, device=dev)
def bx(): ops.call("conan_cfconv_bwd_x", ops.ptr(W), ops.ptr(gy), ops.ptr(tr), ops.ptr(te), ops.ptr(g.tgt), ops.ptr(g.pid), n, F, ops.ptr(dx), None, ops.stream_ptr())
def bw(): ops.call("conan_cfconv_bwd_w_pairs", ops.ptr(x), ops.ptr(gy), ops.ptr(g.num_pairs_dev), g.max_edges, ops.ptr(g.pair_e0), ops.ptr(g.pair_e1), ops.ptr(g.col), ops.ptr(g.tgt), F, ops.ptr(g.pair_dist), 10.0, ops.ptr(dW), None, ops.stream_ptr())
tf, tx, tw = timed(fwd), timed(bx), timed(bw)
# in-model cache state: the filter kernel writes W (and h1) right before the gather reads W
P = int(g.num_pairs_dev.item()); Gs = 50
offset = torch.linspace(0, 10, Gs, device=dev); coeff = -0.5 / float(offset[1] - offset[0]) ** 2
w1 = torch.randn(F, Gs, device=dev) / 7; b1 = torch.randn(F, device=dev) / 10; w2 = torch.randn(F, F, device=dev) / 11; b2 = torch.randn(F, device=dev) / 10
Wm = torch.empty(g.max_edges, F, device=dev); h1 = torch.empty(g.max_edges, F, device=dev)
def produce(): ops.call("conan_filter_fwd", ops.ptr(g.pair_dist), ops.ptr(g.num_pairs_dev), g.max_edges, ops.ptr(offset), Gs, coeff, 10.0, F, ops.ptr(w1), ops.ptr(b1), ops.ptr(w2), ops.ptr(b2), ops.ptr(Wm), (None if os.environ.get("NO_H1") else ops.ptr(h1)), ops.stream_ptr())
def fwd_m(): ops.call("conan_cfconv_fwd", ops.ptr(x), ops.ptr(Wm), ops.ptr(g.rowptr), ops.ptr(g.col), ops.ptr(g.pid), n, F, ops.ptr(out), None, ops.stream_ptr())
ts = []
for _ in range(12):
    produce()
    s_, e_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s_.record(); fwd_m(); e_.record(); torch.cuda.synchronize(); ts.append(s_.elapsed_time(e_) * 1e3)
tm = sorted(ts)[len(ts) // 2]
fwd(); ref = torch.zeros(n, F, device=dev, dtype=torch.float64)
E = g.num_edges
src, tgt, pid = g.col[:E].long(), g.tgt[:E].long(), g.pid[:E].long()
ref.index_add_(0, tgt, x.detach().double()[src] * W.detach().double()[pid])
print(f"{tag} cfconv fwd {tf:6.1f} us (err {float((out.double() - ref).abs().max() / ref.abs().max()):.1e})   bwd_x {tx:6.1f} us   bwd_w_pairs {tw:6.1f} us   | fwd right after the filter kernel (event bracket incl. ~6 us): {tm:6.1f} us")
