"""Probe: the backward kernels of the CFConv gather at cfg2 / Lipophilicity shape — conan_cfconv_bwd_x + conan_cfconv_bwd_w_pairs against the one-launch
conan_cfconv_bwd_xw_pairs — HIP-event timed, warm (20 launches back to back) and with L2 / Infinity Cache evicted in front of every launch (a 1 GiB fill: in
the step the filter tensor W was written a forward pass earlier).  argv: [library ("" = in-tree)] [tag] [shape batch]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from conan_fgw_amd import _lib
if len(sys.argv) > 1 and sys.argv[1]:
    _lib._SO = sys.argv[1]
tag = sys.argv[2] if len(sys.argv) > 2 else ""
shape = sys.argv[3] if len(sys.argv) > 3 else "esol"
B = int(sys.argv[4]) if len(sys.argv) > 4 else 256
from conan_fgw_amd import ops
from conan_fgw_amd.synthetic import make_batch
dev = torch.device("cuda:0")
b = make_batch(shape, B, 5, seed=1236)
pos = torch.from_numpy(b.pos).to(dev); batch = torch.from_numpy(b.batch).to(dev)
gp = ops.graph_ptr_from_batch(batch, b.num_graphs)
g = ops.RadiusGraph(pos, gp, b.num_graphs, 10.0, 32).pairs()
n, F = len(b.z), 128
x = torch.randn(n, F, device=dev); W = torch.randn(g.max_edges, F, device=dev); gy = torch.randn(n, F, device=dev)
tr, te = g.transpose(); dx = torch.empty(n, F, device=dev); dW = torch.empty(g.max_edges, F, device=dev); gm = torch.zeros(1, device=dev)
def bx(): ops.call("conan_cfconv_bwd_x", ops.ptr(W), ops.ptr(gy), ops.ptr(tr), ops.ptr(te), ops.ptr(g.tgt), ops.ptr(g.pid), n, F, ops.ptr(dx), ops.ptr(gm), ops.stream_ptr())
def bw(): ops.call("conan_cfconv_bwd_w_pairs", ops.ptr(x), ops.ptr(gy), ops.ptr(g.num_pairs_dev), g.max_edges, ops.ptr(g.pair_e0), ops.ptr(g.pair_e1), ops.ptr(g.col), ops.ptr(g.tgt), F, ops.ptr(g.pair_dist), 10.0, ops.ptr(dW), ops.ptr(gm), ops.stream_ptr())
def two(): bx(); bw()
def xw(): ops.call("conan_cfconv_bwd_xw_pairs", ops.ptr(W), ops.ptr(x), ops.ptr(gy), ops.ptr(tr), ops.ptr(te), ops.ptr(g.tgt), ops.ptr(g.pid), ops.ptr(g.pair_e0), ops.ptr(g.pair_e1), ops.ptr(g.pair_dist), 10.0, n, F, ops.ptr(dx), ops.ptr(dW), ops.ptr(gm), ops.stream_ptr())
def warm(fn, reps=20):
    for _ in range(3): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3
big = torch.empty(1 << 28, dtype=torch.float32, device=dev)
def cold(fn, reps=9):
    ts = []
    for _ in range(reps):
        big.fill_(1.0)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record(); torch.cuda.synchronize(); ts.append(s.elapsed_time(e) * 1e3)
    return sorted(ts)[len(ts) // 2]
print(f"{tag} {shape} B={B}: warm bwd_x {warm(bx):6.1f} + bwd_w_pairs {warm(bw):6.1f} = {warm(two):6.1f} us | one launch {warm(xw):6.1f} us || evicted: two {cold(two):6.1f} us, one launch {cold(xw):6.1f} us", flush=True)
