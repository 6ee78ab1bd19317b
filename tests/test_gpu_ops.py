"""GPU parity of the SchNet trunk kernels against the oracle (oracle/pyg_semantics.py, run on the CPU in fp64)."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from helpers import rel
from conan_fgw_amd import ops
from conan_fgw_amd.synthetic import make_batch
from oracle import pyg_semantics as ps

pytestmark = pytest.mark.gpu
dev = torch.device("cuda:0")
TOL = 2e-6        # fp32 kernels vs fp64 oracle, relative Frobenius


def _ssp(x):
    return F.softplus(x) - math.log(2.0)


@pytest.mark.parametrize("M,K,N", [(1, 50, 128), (63, 128, 128), (64, 128, 64), (1000, 64, 64), (257, 50, 32), (3210, 128, 128), (130, 10, 256), (5000, 32, 128)])
@pytest.mark.parametrize("act", [False, True])
def test_linear_fwd_bwd(M, K, N, act):
    g = torch.Generator().manual_seed(M + K + N)
    x = torch.randn(M, K, generator=g); w = torch.randn(N, K, generator=g) * 0.2; b = torch.randn(N, generator=g)
    gy = torch.randn(M, N, generator=g)
    xd, wd, bd = (t.to(dev).requires_grad_(True) for t in (x, w, b))
    y = ops.linear(xd, wd, bd, act=act)
    y.backward(gy.to(dev))
    x64, w64, b64 = (t.double().requires_grad_(True) for t in (x, w, b))
    r = F.linear(x64, w64, b64)
    r = _ssp(r) if act else r
    r.backward(gy.double())
    assert rel(y.detach().cpu(), r.detach()) < TOL
    assert rel(xd.grad.cpu(), x64.grad) < TOL
    assert rel(wd.grad.cpu(), w64.grad) < 5e-6
    assert rel(bd.grad.cpu(), b64.grad) < 5e-6


@pytest.mark.parametrize("wmag", [1e-4, 0.2, 300.0])
def test_linear_rows_of_very_different_magnitude_keep_their_own_precision(wmag):
    """The two-plane fp16 GEMM (gemm_t.hip, mlp2.hip) scales every x row by its own power of two: a batch whose rows span ten decades
    (and a zero row, and a row with one non-zero element) must come out with fp32-class error in EVERY row, forward and input gradient."""
    g = torch.Generator().manual_seed(7)
    M, K, N = 2000, 128, 128
    x = torch.randn(M, K, generator=g) * (10.0 ** (torch.rand(M, 1, generator=g) * 10.0 - 6.0))
    x[5] = 0.0; x[6] = 0.0; x[6, 17] = 3.0e-3
    gy = torch.randn(M, N, generator=g) * (10.0 ** (torch.rand(M, 1, generator=g) * 10.0 - 6.0))
    w = torch.randn(N, K, generator=g) * wmag
    xd = x.to(dev).requires_grad_(True); wd = w.to(dev)
    y = ops.linear(xd, wd, None)
    y.backward(gy.to(dev))
    r = x.double() @ w.double().T
    rx = gy.double() @ w.double()

    def rows(a, b):
        err = (a.double().cpu() - b).norm(dim=1); nrm = b.norm(dim=1)
        return float((err / nrm.clamp_min(1e-300))[nrm > 0].max()), float(err[nrm == 0].max()) if bool((nrm == 0).any()) else 0.0
    e, z = rows(y.detach(), r)
    assert e < 2e-6 and z == 0.0
    e, z = rows(xd.grad, rx)
    assert e < 2e-6 and z == 0.0
    # the fused pair of layers: the intermediate is scaled per row as well (taken from the accumulators).  Rows from 1 to 1e3 here:
    # below that the fp32 shifted softplus itself (softplus(r) - ln 2 for r -> 0) is the larger error, in any GEMM form
    x2 = torch.randn(M, K, generator=g) * (10.0 ** (torch.rand(M, 1, generator=g) * 3.0))
    w1 = torch.randn(N, K, generator=g) * 0.2
    w2 = torch.randn(N, N, generator=g) * wmag
    zb = torch.zeros(N, device=dev)
    y2 = ops.mlp2(x2.to(dev), w1.to(dev), zb, w2.to(dev), zb)
    r2 = _ssp(x2.double() @ w1.double().T) @ w2.double().T
    e, _ = rows(y2, r2)
    assert e < 3e-6


def test_linear_residual_nobias_and_device_row_count():
    g = torch.Generator().manual_seed(1)
    x = torch.randn(500, 128, generator=g); w = torch.randn(128, 128, generator=g) * 0.1; res = torch.randn(500, 128, generator=g)
    y = ops.linear(x.to(dev), w.to(dev), None, residual=res.to(dev))
    assert rel(y.cpu(), F.linear(x.double(), w.double()) + res.double()) < TOL
    m_dev = torch.tensor([321], dtype=torch.int32, device=dev)
    xd = x.to(dev).requires_grad_(True); wd = w.to(dev).requires_grad_(True)
    y = ops.linear(xd, wd, None, act=True, m_dev=m_dev)
    assert rel(y[:321].detach().cpu(), _ssp(F.linear(x[:321].double(), w.double()))) < TOL
    gy = torch.randn(500, 128, generator=g)
    y.backward(gy.to(dev))
    x64 = x[:321].double().requires_grad_(True); w64 = w.double().requires_grad_(True)
    _ssp(F.linear(x64, w64)).backward(gy[:321].double())
    assert rel(wd.grad.cpu(), w64.grad) < 5e-6                       # rows >= m_dev must not contribute
    assert rel(xd.grad[:321].cpu(), x64.grad) < TOL and float(xd.grad[321:].abs().max()) == 0.0


def _edges(b, cutoff=10.0, cap=32):
    pos = torch.from_numpy(b.pos).to(dev); batch = torch.from_numpy(b.batch).to(dev)
    gp = ops.graph_ptr_from_batch(batch, b.num_graphs)
    return ops.RadiusGraph(pos, gp, b.num_graphs, cutoff, cap)


@pytest.mark.parametrize("F_", [128, 64, 256, 32])
def test_cfconv_fwd_bwd(F_):
    b = make_batch("esol", 6, 5, seed=5, box=14.0)
    g = _edges(b)
    E = g.num_edges
    gen = torch.Generator().manual_seed(F_)
    x = torch.randn(len(b.z), F_, generator=gen); W = torch.randn(E, F_, generator=gen); gy = torch.randn(len(b.z), F_, generator=gen)
    Wfull = torch.zeros(g.max_edges, F_); Wfull[:E] = W
    xd = x.to(dev).requires_grad_(True); Wd = Wfull.to(dev).requires_grad_(True)
    out = ops.cfconv(xd, Wd, g)
    out.backward(gy.to(dev))
    ei = g.edge_index().cpu()
    x64 = x.double().requires_grad_(True); W64 = W.double().requires_grad_(True)
    ref = ps.scatter(x64[ei[0]] * W64, ei[1], dim=0, dim_size=len(b.z))
    ref.backward(gy.double())
    assert rel(out.detach().cpu(), ref.detach()) < TOL
    assert rel(xd.grad.cpu(), x64.grad) < TOL
    assert rel(Wd.grad[:E].cpu(), W64.grad) < TOL


def test_rbf_cutoff_embedding_segment_sum():
    b = make_batch("esol", 4, 5, seed=9, box=14.0)
    g = _edges(b)
    E = g.num_edges
    gs = ps.GaussianSmearing(0.0, 10.0, 50)
    rbf = ops.rbf_expand(g, gs.offset.to(dev), gs.coeff)[:E].cpu()
    d = g.edge_weight().cpu()
    assert rel(rbf, gs(d.double())) < TOL
    Wr = torch.randn(g.max_edges, 128)
    Wd = Wr.to(dev).requires_grad_(True)
    out = ops.cutoff_scale(Wd, g)
    C = 0.5 * (torch.cos(d.double() * math.pi / 10.0) + 1.0)
    assert rel(out[:E].detach().cpu(), Wr[:E].double() * C[:, None]) < TOL
    out.backward(torch.ones_like(out))
    assert rel(Wd.grad[:E].cpu(), C[:, None].expand(-1, 128)) < TOL
    # embedding with padding row
    emb = torch.nn.Embedding(100, 64, padding_idx=0)
    z = torch.from_numpy(b.z); z[0] = 0
    wd = emb.weight.detach().to(dev).requires_grad_(True)
    o = ops.embedding(z.to(dev), wd, 0)
    assert torch.equal(o.detach().cpu(), emb(z).detach())
    gy = torch.randn(len(z), 64)
    o.backward(gy.to(dev)); emb(z).backward(gy)
    assert rel(wd.grad.cpu(), emb.weight.grad) < TOL and float(wd.grad[0].abs().max()) == 0.0
    # sum readout
    gp = g.graph_ptr
    xs = torch.randn(len(z), 64)
    xd = xs.to(dev).requires_grad_(True)
    o = ops.segment_sum(xd, gp, b.num_graphs)
    assert rel(o.detach().cpu(), ps.scatter(xs.double(), torch.from_numpy(b.batch), 0, b.num_graphs)) < TOL
    gg = torch.randn(b.num_graphs, 64)
    o.backward(gg.to(dev))
    assert torch.equal(xd.grad.cpu(), gg[torch.from_numpy(b.batch)])


@pytest.mark.parametrize("wmag", [1.0, 1e-4, 300.0], ids=["w1", "w1e-4", "w300"])
@pytest.mark.parametrize("F_,Gs", [(128, 50), (64, 50), (32, 50), (128, 10)])
def test_fused_filter_matches_oracle_fwd_bwd(F_, Gs, wmag):
    """conan_filter_fwd (rbf -> mlp -> cosine cutoff in registers) vs the oracle's GaussianSmearing + mlp + C(d).  wmag: the second
    layer's weights scaled by 1e-4 / 300 — the fp16 planes of the weights take their scale from the matrix's maximum, so tiny and huge
    weights must come out as exact as ordinary ones."""
    assert ops.filter_fused_supported(Gs, F_)
    b = make_batch("esol", 5, 5, seed=13, box=14.0)
    g = _edges(b)
    E = g.num_edges
    torch.manual_seed(F_ + Gs)
    gs = ps.GaussianSmearing(0.0, 10.0, Gs)
    mlp = torch.nn.Sequential(torch.nn.Linear(Gs, F_), ps.ShiftedSoftplus(), torch.nn.Linear(F_, F_))
    with torch.no_grad():
        for p in mlp.parameters():
            p.add_(0.1 * torch.randn_like(p))
        mlp[2].weight.mul_(wmag); mlp[0].weight.mul_(min(wmag, 3.0))
    prm = [p.detach().to(dev).requires_grad_(True) for p in (mlp[0].weight, mlp[0].bias, mlp[2].weight, mlp[2].bias)]
    W = ops.filter_generate(g, gs.offset.to(dev), gs.coeff, *prm, use_pairs=False)  # NB: backward expects a pre-cutoff gradient
    d = g.edge_weight().cpu().double()
    m64 = mlp.double()
    C = 0.5 * (torch.cos(d * math.pi / 10.0) + 1.0)
    ref = m64(gs(d)) * C[:, None]
    assert rel(W[:E].detach().cpu(), ref.detach()) < TOL
    gy = torch.randn(E, F_)
    gfull = torch.zeros(g.max_edges, F_); gfull[:E] = gy * C[:, None].float()      # the op consumes the PRE-cutoff gradient
    W.backward(gfull.to(dev))                                                      # (what cfconv(..., pre_cutoff_grad=True) hands it)
    ref.backward(gy.double())
    for got, want in zip(prm, (m64[0].weight, m64[0].bias, m64[2].weight, m64[2].bias)):
        assert rel(got.grad.cpu(), want.grad) < 1e-5
    # and the composed (generic) path gives the same filters
    rbf = ops.rbf_expand(g, gs.offset.to(dev), gs.coeff)
    h1 = ops.linear(rbf, prm[0].detach(), prm[1].detach(), act=True, m_dev=g.num_edges_dev)
    W2 = ops.cutoff_scale(ops.linear(h1, prm[2].detach(), prm[3].detach(), m_dev=g.num_edges_dev), g)
    assert rel(W[:E].detach().cpu(), W2[:E].cpu()) < 1e-6


@pytest.mark.parametrize("shape,B,K,Gs", [("esol", 6, 5, 50), ("lipo", 3, 3, 50), ("esol", 3, 2, 10)])
def test_filter_fused_into_the_gather_matches_the_two_kernels_and_fp64(shape, B, K, Gs):
    """conan_filter_cfconv_fwd (forward only): the filter rows are generated per directed edge and consumed from the accumulators — against
    (i) filter_generate (one row per pair) + cfconv, the training path's two kernels: same arithmetic, other summation order (1e-6), and
    (ii) the fp64 formula out[i] = sum_j x_j * mlp(rbf(d_ij)) * C(d_ij).  Lipophilicity-sized conformers have truncated rows (33 edges: two
    tiles per target).  Bitwise repeatable; refuses to run where a gradient is required."""
    F_ = 128
    assert ops.filter_cfconv_supported(Gs, F_) and not ops.filter_cfconv_supported(Gs, 64)
    b = make_batch(shape, B, K, seed=33)
    g = _edges(b)
    E, n = g.num_edges, g.num_atoms
    torch.manual_seed(Gs)
    gs = ps.GaussianSmearing(0.0, 10.0, Gs)
    mlp = torch.nn.Sequential(torch.nn.Linear(Gs, F_), ps.ShiftedSoftplus(), torch.nn.Linear(F_, F_))
    with torch.no_grad():
        for p in mlp.parameters():
            p.add_(0.1 * torch.randn_like(p))
    prm = [p.detach().to(dev) for p in (mlp[0].weight, mlp[0].bias, mlp[2].weight, mlp[2].bias)]
    x = torch.randn(n, F_, device=dev)
    off = gs.offset.to(dev)
    with torch.no_grad():
        out = ops.filter_cfconv(x, g, off, gs.coeff, *prm)
        out2 = ops.filter_cfconv(x, g, off, gs.coeff, *prm)
        W = ops.filter_generate(g, off, gs.coeff, *prm)
        two = ops.cfconv(x, W, g, pre_cutoff_grad=True, use_pairs=True)
    assert torch.equal(out, out2)                                              # a target's row is 0 + a (+ b): order-independent
    assert rel(out.cpu(), two.cpu()) < 1e-6
    ei = g.edge_index().cpu()
    d = g.edge_weight().cpu().double()
    Wd = mlp.double()(gs(d)) * (0.5 * (torch.cos(d * math.pi / 10.0) + 1.0))[:, None]
    ref = torch.zeros(n, F_, dtype=torch.float64).index_add_(0, ei[1], x.cpu().double()[ei[0]] * Wd.detach())
    assert rel(out.cpu(), ref) < TOL
    deg = torch.bincount(ei[1], minlength=n)
    assert bool((out[(deg == 0).to(dev)] == 0).all())                          # targets without edges: cleared rows
    if shape == "lipo":
        assert int(deg.max()) == 33                                             # the two-tile case is in the test
    xr = x.clone().requires_grad_(True)
    with pytest.raises(RuntimeError, match="forward-only"):
        ops.filter_cfconv(xr, g, off, gs.coeff, *prm)


@pytest.mark.parametrize("shape,B,K", [("esol", 6, 5), ("lipo", 3, 2), ("bace", 2, 2)])
def test_cfconv_backward_in_one_launch_equals_the_two_kernels(shape, B, K):
    """conan_cfconv_bwd_xw_pairs (dx and the pair gradient from ONE walk of the by-source CSR: a pair row is written by the edge that is its e0,
    from the rows the dx gather holds plus x[target]) against conan_cfconv_bwd_x + conan_cfconv_bwd_w_pairs: the same bits in dx, in every pair
    row and in max |g| — also where the neighbour cap leaves one-directional pairs (lipo / bace: e1 = -1) — and repeated passes stay equal
    (the forward gather clears the maximum's word every time)."""
    b = make_batch(shape, B, K, seed=17)
    g = _edges(b)
    n, F_ = g.num_atoms, 128
    P = int(g.pairs().num_pairs_dev.item())
    if shape != "esol":
        assert int((g.pair_e1[:P] < 0).sum()) > 0                               # one-directional pairs are in the test
    gen = torch.Generator().manual_seed(B)
    x0, W0, gy = torch.randn(n, F_, generator=gen), torch.randn(g.max_edges, F_, generator=gen), torch.randn(n, F_, generator=gen)

    def run(fused):
        ops.FUSED_CFCONV_BACKWARD = fused
        try:
            outs = []
            for rep in range(2):
                x = x0.to(dev).requires_grad_(True); W = W0.to(dev).requires_grad_(True)
                out = ops.cfconv(x, W, g, pre_cutoff_grad=True, use_pairs=True)
                (dx, dW) = torch.autograd.grad(out, (x, W), (gy * (1.0 if rep == 0 else 1e-3)).to(dev))
                tag = getattr(dW, "_conan_gmax", None)
                torch.cuda.synchronize()
                outs.append((out.detach().clone(), dx.clone(), dW[:P].clone(), None if tag is None else tag[0].clone()))
            return outs
        finally:
            ops.FUSED_CFCONV_BACKWARD = True

    a, c = run(True), run(False)
    for (o1, dx1, dw1, m1), (o2, dx2, dw2, m2) in zip(a, c):
        assert torch.equal(o1, o2) and torch.equal(dx1, dx2) and torch.equal(dw1, dw2)
        assert m1 is not None and m2 is not None and torch.equal(m1, m2)
        assert float(m1) == float(dw1.abs().max())
    assert float(a[1][3]) < 0.01 * float(a[0][3])                                # the second pass's (1000x smaller) maximum is not the first's


def test_filter_gradient_maximum_travels_with_the_tensor_and_is_dropped_when_the_gradient_was_touched():
    """The fp16-plane filter backward scales g from max|g|, which the pair-gradient kernel tracks (ops._tag_gmax / _take_gmax).  (i) One
    consumer per filter (every model): the tag survives the hop through the autograd engine, the fast kernels run.  (ii) ONE filter feeding
    TWO CFConvs: autograd sums the two pair gradients (the first buffer may be reused in place) — the maximum of one addend is not the
    maximum of the sum, so the backward must fall back to the scale-free kernels, and the result must still match fp64."""
    b = make_batch("esol", 6, 5, seed=21)
    g = _edges(b)
    n, F_, Gs = g.num_atoms, 128, 50
    torch.manual_seed(3)
    gs = ps.GaussianSmearing(0.0, 10.0, Gs)
    mlp = torch.nn.Sequential(torch.nn.Linear(Gs, F_), ps.ShiftedSoftplus(), torch.nn.Linear(F_, F_))
    x1, x2 = torch.randn(n, F_), 30.0 * torch.randn(n, F_)          # the second consumer's gradient is 30x larger than the first's
    gy = torch.randn(n, F_)

    def run(two):
        prm = [p.detach().to(dev).requires_grad_(True) for p in (mlp[0].weight, mlp[0].bias, mlp[2].weight, mlp[2].bias)]
        W = ops.filter_generate(g, gs.offset.to(dev), gs.coeff, *prm, use_pairs=True)
        out = ops.cfconv(x1.to(dev), W, g, pre_cutoff_grad=True, use_pairs=True)
        if two:
            out = out + ops.cfconv(x2.to(dev), W, g, pre_cutoff_grad=True, use_pairs=True)
        before = dict(ops.gmax_stats)
        out.backward(gy.to(dev))
        torch.cuda.synchronize()
        return prm, {k: ops.gmax_stats[k] - before[k] for k in before}

    def ref(two):
        m64 = torch.nn.Sequential(torch.nn.Linear(Gs, F_), ps.ShiftedSoftplus(), torch.nn.Linear(F_, F_)).double()
        m64.load_state_dict({k: v.double() for k, v in mlp.state_dict().items()})
        ei = torch.stack([g.col[: g.num_edges].cpu().long(), g.tgt[: g.num_edges].cpu().long()])
        d = g.edge_weight().cpu().double()
        Wr = m64(gs(d)) * (0.5 * (torch.cos(d * math.pi / 10.0) + 1.0))[:, None]
        xs = x1.double() + (x2.double() if two else 0.0)
        out = torch.zeros(n, F_, dtype=torch.float64).index_add_(0, ei[1], xs[ei[0]] * Wr)
        out.backward(gy.double())
        return [m64[0].weight.grad, m64[0].bias.grad, m64[2].weight.grad, m64[2].bias.grad]

    prm, st = run(False)
    assert st == {"tracked": 1, "used": 1}, st
    for got, want in zip(prm, ref(False)):
        assert rel(got.grad.cpu(), want) < 1e-5
    prm, st = run(True)
    assert st["tracked"] == 2 and st["used"] == 0, st
    for got, want in zip(prm, ref(True)):
        assert rel(got.grad.cpu(), want) < 1e-5


@pytest.mark.parametrize("gscale", [1.0, 2e-6, 3e3])
@pytest.mark.parametrize("M,N,K", [(1000, 128, 128), (40000, 128, 128), (33, 128, 128), (5000, 256, 128)])
def test_wgrad_scaled_two_plane_fp16_matches_fp64(M, N, K, gscale):
    """conan_linear_wgrad_scaled: dW = g^T x on two fp16 planes with g scaled from its device-side maximum (the edge-level dw2 of the
    filter network); 1e-5 of an fp64 matmul whatever the gradient's magnitude, bias sums exact to fp32, bitwise reproducible."""
    from conan_fgw_amd._lib import call, lib, ptr, stream_ptr
    gen = torch.Generator().manual_seed(M + N + K)
    g = (torch.randn(M + 40, N, generator=gen) * gscale).to(dev)
    x = (torch.rand(M + 40, K, generator=gen) * 4 - 0.69).to(dev)            # shifted-softplus outputs
    md = torch.tensor([M], dtype=torch.int32, device=dev)
    gmax = g[:M].abs().max().reshape(1).contiguous()
    dW, db = torch.empty(N, K, device=dev), torch.empty(N, device=dev)
    ws = torch.empty(int(lib().conan_linear_wgrad_ws(M + 40, K, N)), device=dev)
    call("conan_linear_wgrad_scaled", ptr(g), ptr(x), M + 40, K, N, ptr(md), ptr(dW), ptr(db), ptr(ws), ptr(gmax), stream_ptr())
    ref = g[:M].double().T @ x[:M].double()
    assert rel(dW.double().cpu().numpy(), ref.cpu().numpy()) < 1e-5
    assert rel(db.double().cpu().numpy(), g[:M].double().sum(0).cpu().numpy()) < 1e-5
    dW2 = torch.empty_like(dW)
    call("conan_linear_wgrad_scaled", ptr(g), ptr(x), M + 40, K, N, ptr(md), ptr(dW2), ptr(db), ptr(ws), ptr(gmax), stream_ptr())
    assert torch.equal(dW, dW2)
    assert lib().conan_linear_wgrad_scaled(ptr(g), ptr(x), M, 64, N, None, ptr(dW), None, ptr(ws), ptr(gmax), stream_ptr()) < 0      # K <= 64: not offered


def test_pair_gradient_kernel_reports_its_maximum():
    """conan_cfconv_bwd_w_pairs(gmax): the device float ends up as max |dWp| over the pair rows it wrote (zeroed by the caller)."""
    from conan_fgw_amd._lib import call, ptr, stream_ptr
    from conan_fgw_amd.synthetic import make_batch
    b = make_batch("esol", 6, 3, seed=11)
    pos = torch.from_numpy(b.pos).to(dev); batch = torch.from_numpy(b.batch).to(dev)
    gp = ops.graph_ptr_from_batch(batch, b.num_graphs)
    g = ops.RadiusGraph(pos, gp, b.num_graphs, 10.0, 32).pairs()
    n, F = len(b.z), 128
    x = torch.randn(n, F, device=dev); dout = torch.randn(n, F, device=dev) * 1e-3
    dWp = torch.zeros(g.max_edges, F, device=dev)
    gmax = torch.zeros(1, device=dev)
    call("conan_cfconv_bwd_w_pairs", ptr(x), ptr(dout), ptr(g.num_pairs_dev), g.max_edges, ptr(g.pair_e0), ptr(g.pair_e1), ptr(g.col), ptr(g.tgt), F,
         ptr(g.pair_dist), 10.0, ptr(dWp), ptr(gmax), stream_ptr())
    P = int(g.num_pairs_dev.item())
    assert P > 0 and float(gmax) == float(dWp[:P].abs().max()) and float(gmax) > 0


@pytest.mark.parametrize("M,N,K", [(1000, 128, 128), (4133, 128, 64), (777, 64, 128), (2500, 32, 32), (300, 128, 52), (33, 64, 64), (5000, 256, 128)])
def test_wgrad_lds_staged_matches_fp64(M, N, K):
    """conan_linear_wgrad (LDS-staged bf16-split path: N % 4 == 0, K % 4 == 0) against an fp64 matmul; a device-side row
    count below the buffer size must mask the tail rows; tolerance 1e-5 relative (fp32-class, well inside the 1e-4 bar)."""
    from conan_fgw_amd._lib import call, lib, ptr, stream_ptr
    dev = torch.device("cuda:0")
    gen = torch.Generator().manual_seed(M + N + K)
    g = torch.randn(M + 40, N, generator=gen).to(dev)
    x = torch.randn(M + 40, K, generator=gen).to(dev)
    md = torch.tensor([M], dtype=torch.int32, device=dev)
    dW, db = torch.empty(N, K, device=dev), torch.empty(N, device=dev)
    ws = torch.empty(int(lib().conan_linear_wgrad_ws(M + 40, K, N)), device=dev)
    call("conan_linear_wgrad", ptr(g), ptr(x), M + 40, K, N, ptr(md), ptr(dW), ptr(db), ptr(ws), stream_ptr())
    ref = g[:M].double().T @ x[:M].double()
    assert rel(dW.double().cpu().numpy(), ref.cpu().numpy()) < 1e-5
    assert rel(db.double().cpu().numpy(), g[:M].double().sum(0).cpu().numpy()) < 1e-5
    dW2 = torch.empty_like(dW)
    call("conan_linear_wgrad", ptr(g), ptr(x), M + 40, K, N, ptr(md), ptr(dW2), ptr(db), ptr(ws), stream_ptr())
    assert torch.equal(dW, dW2)                                     # fixed reduction order => bitwise reproducible


@pytest.mark.parametrize("M,N,Gs", [(3000, 128, 50), (517, 64, 50), (100, 32, 20), (2000, 128, 128)])
def test_rbf_wgrad_matches_materialised_rbf(M, N, Gs):
    """conan_rbf_wgrad == g^T GaussianSmearing(dist) with the expansion generated inside the GEMM."""
    from conan_fgw_amd._lib import call, lib, ptr, stream_ptr
    dev = torch.device("cuda:0")
    gen = torch.Generator().manual_seed(M + N)
    g = torch.randn(M + 7, N, generator=gen).to(dev)
    dist = (torch.rand(M + 7, generator=gen) * 10).to(dev)
    off = torch.linspace(0, 10, Gs).to(dev)
    coeff = -0.5 / float(off[1] - off[0]) ** 2
    md = torch.tensor([M], dtype=torch.int32, device=dev)
    dW, db = torch.empty(N, Gs, device=dev), torch.empty(N, device=dev)
    ws = torch.empty(int(lib().conan_linear_wgrad_ws(M + 7, Gs, N)), device=dev)
    call("conan_rbf_wgrad", ptr(g), ptr(dist), M + 7, ptr(off), Gs, coeff, N, ptr(md), ptr(dW), ptr(db), ptr(ws), stream_ptr())
    rbf = torch.exp(coeff * (dist[:M, None].double() - off[None].double()) ** 2)
    ref = g[:M].double().T @ rbf
    assert rel(dW.double().cpu().numpy(), ref.cpu().numpy()) < 1e-5
    assert rel(db.double().cpu().numpy(), g[:M].double().sum(0).cpu().numpy()) < 1e-5


@pytest.mark.parametrize("M", [1, 33, 1000, 25275])
@pytest.mark.parametrize("with_res", [False, True])
def test_mlp2_fused_pair_of_linears_fwd_bwd(M, with_res):
    """ops.mlp2 (conan_mlp2_fwd / conan_mlp2_bwd: lin2 -> ssp -> lin (+ x) of an InteractionBlock in one launch each way) against the fp64
    formula with torch autograd, and against the two-kernel composition it replaces."""
    gen = torch.Generator().manual_seed(M)
    Fh = 128
    x = torch.randn(M, Fh, generator=gen).to(dev).requires_grad_(True)
    res = torch.randn(M, Fh, generator=gen).to(dev).requires_grad_(True) if with_res else None
    w1 = (torch.randn(Fh, Fh, generator=gen) / 11).to(dev).requires_grad_(True); b1 = (torch.randn(Fh, generator=gen) / 5).to(dev).requires_grad_(True)
    w2 = (torch.randn(Fh, Fh, generator=gen) / 11).to(dev).requires_grad_(True); b2 = (torch.randn(Fh, generator=gen) / 5).to(dev).requires_grad_(True)
    gy = torch.randn(M, Fh, generator=gen).to(dev)
    leaves = [x, w1, b1, w2, b2] + ([res] if with_res else [])
    from conan_fgw_amd._lib import lib
    assert lib().conan_mlp2_supported(M, Fh, Fh, Fh) == 1 and lib().conan_mlp2_supported(70000, Fh, Fh, Fh) == 0 and lib().conan_mlp2_supported(M, 64, Fh, Fh) == 0

    def grads(fn):
        for t in leaves:
            t.grad = None
        y = fn()
        (y * gy).sum().backward()
        return [y.detach()] + [t.grad.detach().clone() for t in leaves]

    fused = grads(lambda: ops.mlp2(x, w1, b1, w2, b2, residual=res))
    comp = grads(lambda: ops.linear(ops.linear(x, w1, b1, act=True), w2, b2, residual=res))
    d = [t.detach().double() for t in leaves]
    for t in d:
        t.requires_grad_(True)
    yd = _ssp(d[0] @ d[1].T + d[2]) @ d[3].T + d[4] + (d[5] if with_res else 0.0)
    (yd * gy.double()).sum().backward()
    ref = [yd.detach()] + [t.grad for t in d]
    for a, c, r in zip(fused, comp, ref):
        assert rel(a.double().cpu().numpy(), r.cpu().numpy()) < 2e-6
        assert rel(a.cpu().numpy(), c.cpu().numpy()) < 2e-6


@pytest.mark.parametrize("M", [1, 33, 1000, 25275])
def test_mlp2_outact_fused_head_fwd_bwd(M):
    """ops.mlp2_outact (lin1 -> lin2 -> ssp of the per-atom heads in one launch each way, 128 -> 64 -> 64) against the fp64 formula with torch
    autograd and against the two-kernel composition."""
    gen = torch.Generator().manual_seed(M + 5)
    x = torch.randn(M, 128, generator=gen).to(dev).requires_grad_(True)
    w1 = (torch.randn(64, 128, generator=gen) / 11).to(dev).requires_grad_(True); b1 = (torch.randn(64, generator=gen) / 5).to(dev).requires_grad_(True)
    w2 = (torch.randn(64, 64, generator=gen) / 8).to(dev).requires_grad_(True); b2 = (torch.randn(64, generator=gen) / 5).to(dev).requires_grad_(True)
    gy = torch.randn(M, 64, generator=gen).to(dev)
    leaves = [x, w1, b1, w2, b2]
    from conan_fgw_amd._lib import lib
    assert lib().conan_mlp2_outact_supported(M, 128, 64, 64) == 1 and lib().conan_mlp2_outact_supported(M, 128, 128, 128) == 0

    def grads(fn):
        for t in leaves:
            t.grad = None
        y = fn()
        (y * gy).sum().backward()
        return [y.detach()] + [t.grad.detach().clone() for t in leaves]

    fused = grads(lambda: ops.mlp2_outact(x, w1, b1, w2, b2))
    comp = grads(lambda: ops.linear(ops.linear(x, w1, b1), w2, b2, act=True))
    d = [t.detach().double().requires_grad_(True) for t in leaves]
    yd = _ssp((d[0] @ d[1].T + d[2]) @ d[3].T + d[4])
    (yd * gy.double()).sum().backward()
    ref = [yd.detach()] + [t.grad for t in d]
    for a, c, r in zip(fused, comp, ref):
        assert rel(a.double().cpu().numpy(), r.cpu().numpy()) < 2e-6
        assert rel(a.cpu().numpy(), c.cpu().numpy()) < 2e-6


def test_wgrad_slabs_batch_equals_the_per_job_launches():
    """conan_linear_wgrad_slabs_batch: many weight gradients' stage 1 in one launch per k-tile width.  With default slice counts the slabs
    (hence the reduced dW / db) are bit-identical to conan_linear_wgrad_slabs job by job; with fewer, longer slices the result agrees with
    the fp64 product to fp32 rounding; a device-side row count masks the tail; a job whose device-side count is 0 yields zeros."""
    from conan_fgw_amd._lib import WgradJob, WgradSlabJob, call, lib, ptr, stream_ptr
    shapes = [(3000, 128, 128), (2500, 64, 128), (777, 64, 64), (5000, 128, 64), (1, 128, 128), (300, 32, 128), (4000, 256, 128), (0, 128, 128)]
    gen = torch.Generator().manual_seed(7)
    gs = [torch.randn(M + 9, N, generator=gen).to(dev) for M, K, N in shapes]
    xs = [torch.randn(M + 9, K, generator=gen).to(dev) for M, K, N in shapes]
    mds = [torch.tensor([M], dtype=torch.int32, device=dev) if i % 2 else None for i, (M, K, N) in enumerate(shapes)]
    rows = [M if mds[i] is not None else M + 9 for i, (M, K, N) in enumerate(shapes)]

    def reduce(wss, slices):
        jobs = (WgradJob * len(shapes))()
        out = []
        for q, (M, K, N) in enumerate(shapes):
            dW, db = torch.empty(N, K, device=dev), torch.empty(N, device=dev)
            jobs[q].ws, jobs[q].dW, jobs[q].dbias = wss[q].data_ptr(), dW.data_ptr(), db.data_ptr()
            jobs[q].M, jobs[q].K, jobs[q].N, jobs[q].slices = M + 9, K, N, slices[q]
            out.append((dW, db))
        call("conan_wgrad_reduce_batch", jobs, len(shapes), stream_ptr())
        return out

    def batch(slices):
        wss = [torch.empty(int(lib().conan_linear_wgrad_ws(M + 9, K, N)), device=dev) for M, K, N in shapes]
        sj = (WgradSlabJob * len(shapes))()
        for q, (M, K, N) in enumerate(shapes):
            sj[q].g, sj[q].x, sj[q].m_dev, sj[q].ws = gs[q].data_ptr(), xs[q].data_ptr(), mds[q].data_ptr() if mds[q] is not None else None, wss[q].data_ptr()
            sj[q].M, sj[q].K, sj[q].N, sj[q].slices = M + 9, K, N, slices[q]
        call("conan_linear_wgrad_slabs_batch", sj, len(shapes), stream_ptr())
        return reduce(wss, slices)

    wss = [torch.empty(int(lib().conan_linear_wgrad_ws(M + 9, K, N)), device=dev) for M, K, N in shapes]
    for q, (M, K, N) in enumerate(shapes):
        call("conan_linear_wgrad_slabs", ptr(gs[q]), ptr(xs[q]), M + 9, K, N, ptr(mds[q]), ptr(wss[q]), stream_ptr())
    one_by_one = reduce(wss, [0] * len(shapes))
    batched = batch([0] * len(shapes))
    long_slices = batch([max(1, (M + 9 + 255) // 256) for M, K, N in shapes])
    for q, (M, K, N) in enumerate(shapes):
        ref = gs[q][:rows[q]].double().T @ xs[q][:rows[q]].double()
        assert torch.equal(batched[q][0], one_by_one[q][0]) and torch.equal(batched[q][1], one_by_one[q][1]), shapes[q]
        assert rel(long_slices[q][0].double().cpu().numpy(), ref.cpu().numpy()) < 1e-5, shapes[q]
        assert rel(long_slices[q][1].double().cpu().numpy(), gs[q][:rows[q]].double().sum(0).cpu().numpy()) < 1e-5, shapes[q]


@pytest.mark.parametrize("M,run", [(65536, 3), (70001, 3), (131075, 2)])
def test_edge_level_weight_gradients_over_one_x_share_its_staging_bit_for_bit(M, run):
    """From 65 536 rows on (a job then has all 512 slices) a run of two or three 128 x 128 weight gradients over the same x — ViS_MP's dk / dv /
    f_proj of one f_ij, torch_geometric_visnet.py:600-604,637-640 — and the two n tiles of a [M,256] gradient (s_proj, :615) run with ONE
    workgroup per row slice that stages x once for all of them (k_wgrad_lds_shared).  Slices, stage order and the order of the partial
    products are those of the single-job kernel: dW and db are bit for bit what separate launches give (for N = 256: what two launches on the
    two halves of g give), a device-side row count masks the tail, and both agree with the fp64 product."""
    from conan_fgw_amd._lib import WgradJob, WgradSlabJob, call, lib, ptr, stream_ptr
    gen = torch.Generator().manual_seed(M + run)
    x = torch.randn(M + 9, 128, generator=gen).to(dev)
    gs = [torch.randn(M + 9, 128, generator=gen).to(dev) * (10.0 ** q) for q in range(run)]
    md = torch.tensor([M], dtype=torch.int32, device=dev)

    def reduce(wss):
        jobs = (WgradJob * run)()
        out = []
        for q in range(run):
            dW, db = torch.empty(128, 128, device=dev), torch.empty(128, device=dev)
            jobs[q].ws, jobs[q].dW, jobs[q].dbias = wss[q].data_ptr(), dW.data_ptr(), db.data_ptr()
            jobs[q].M, jobs[q].K, jobs[q].N, jobs[q].slices = M + 9, 128, 128, 0
            out.append((dW, db))
        call("conan_wgrad_reduce_batch", jobs, run, stream_ptr())
        return out

    wsz = int(lib().conan_linear_wgrad_ws(M + 9, 128, 128))
    wss = [torch.empty(wsz, device=dev) for _ in range(run)]
    for q in range(run):
        call("conan_linear_wgrad_slabs", ptr(gs[q]), ptr(x), M + 9, 128, 128, ptr(md), ptr(wss[q]), stream_ptr())
    one_by_one = reduce(wss)
    wss2 = [torch.full((wsz,), float("nan"), device=dev) for _ in range(run)]
    sj = (WgradSlabJob * run)()
    for q in range(run):
        sj[q].g, sj[q].x, sj[q].m_dev, sj[q].ws = gs[q].data_ptr(), x.data_ptr(), md.data_ptr(), wss2[q].data_ptr()
        sj[q].M, sj[q].K, sj[q].N, sj[q].slices = M + 9, 128, 128, 0
    call("conan_linear_wgrad_slabs_batch", sj, run, stream_ptr())
    shared = reduce(wss2)
    for q in range(run):
        assert torch.equal(shared[q][0], one_by_one[q][0]) and torch.equal(shared[q][1], one_by_one[q][1]), q
        ref = gs[q][:M].double().T @ x[:M].double()
        assert rel(shared[q][0].double().cpu().numpy(), ref.cpu().numpy()) < 2e-6
        assert rel(shared[q][1].double().cpu().numpy(), gs[q][:M].double().sum(0).cpu().numpy()) < 2e-6
    if run == 2:                                                         # the [M,256] gradient of one layer: its two n tiles
        gw = torch.cat(gs, dim=1).contiguous()
        dW, db = torch.empty(256, 128, device=dev), torch.empty(256, device=dev)
        ws = torch.empty(int(lib().conan_linear_wgrad_ws(M + 9, 128, 256)), device=dev)
        call("conan_linear_wgrad", ptr(gw), ptr(x), M + 9, 128, 256, ptr(md), ptr(dW), ptr(db), ptr(ws), stream_ptr())
        for q in range(2):
            assert torch.equal(dW[128 * q: 128 * (q + 1)], one_by_one[q][0]) and torch.equal(db[128 * q: 128 * (q + 1)], one_by_one[q][1])


@pytest.mark.parametrize("gscale", [None, 1.0, 3e-7, 4e4], ids=["bf16x3", "f16x2", "f16x2_tiny_grad", "f16x2_huge_grad"])
@pytest.mark.parametrize("M,Gs", [(1, 50), (31, 50), (4133, 50), (40000, 50), (3000, 20), (2500, 63)])
def test_filter_bwd_fused_matches_fp64_and_the_composed_kernels(M, Gs, gscale):
    """conan_filter_bwd (dh1 = (g w2) * ssp'(h1) kept in registers, dW1 = dh1^T rbf, db1 = colsum dh1) against the fp64 formula and
    against the two kernels it replaces (conan_linear_fwd act=2 + conan_rbf_wgrad); rows beyond the device-side count are
    masked; the slab form reduced through conan_wgrad_reduce_batch gives the same bits; repeat runs are bitwise equal."""
    import ctypes
    from conan_fgw_amd._lib import WgradJob, call, lib, ptr, stream_ptr
    Fh, pad = 128, 37
    assert lib().conan_filter_bwd_supported(Gs, Fh) == 1 and lib().conan_filter_bwd_supported(64, Fh) == 0 and lib().conan_filter_bwd_supported(50, 64) == 0
    gen = torch.Generator().manual_seed(M + Gs)
    g = (torch.randn(M + pad, Fh, generator=gen) * (gscale or 1.0)).to(dev)
    # gscale given: the two-plane fp16 path, which takes max |g| from the device (here: over the VALID rows, as the producer kernel
    # would have tracked it) and scales the gradient into fp16's range — tiny (3e-7) and huge (4e4) gradients must come out as exact
    gmax = g[:M].abs().max().reshape(1).contiguous() if gscale is not None else None
    gm = ptr(gmax) if gmax is not None else None
    h1 = (torch.rand(M + pad, Fh, generator=gen) * 3 - 0.6).to(dev)                    # ssp output range (> -ln 2)
    dist = (torch.rand(M + pad, generator=gen) * 10).to(dev)
    w2 = (torch.randn(Fh, Fh, generator=gen) / 11).to(dev)
    off = torch.linspace(0, 10, Gs).to(dev)
    coeff = -0.5 / float(off[1] - off[0]) ** 2
    md = torch.tensor([M], dtype=torch.int32, device=dev)
    ws = torch.empty(int(lib().conan_filter_bwd_ws(M + pad, Gs, Fh)), device=dev)
    dW, db = torch.empty(Fh, Gs, device=dev), torch.empty(Fh, device=dev)
    call("conan_filter_bwd", ptr(g), ptr(h1), ptr(dist), M + pad, ptr(off), Gs, coeff, ptr(w2), Fh, ptr(md), ptr(dW), ptr(db), ptr(ws), gm, stream_ptr())
    dh = (g[:M].double() @ w2.double()) * (1 - 0.5 * torch.exp(-h1[:M].double()))
    rbf = torch.exp(coeff * (dist[:M, None].double() - off[None].double()) ** 2)
    assert rel(dW.double().cpu().numpy(), (dh.T @ rbf).cpu().numpy()) < 1e-5
    assert rel(db.double().cpu().numpy(), dh.sum(0).cpu().numpy()) < 1e-5
    # the composed pair
    dh1 = torch.empty(M + pad, Fh, device=dev)
    call("conan_linear_fwd", ptr(g), ptr(w2), None, ptr(h1), M + pad, Fh, Fh, 1, 2, ptr(md), ptr(dh1), stream_ptr())
    dWc, dbc = torch.empty_like(dW), torch.empty_like(db)
    wsc = torch.empty(int(lib().conan_linear_wgrad_ws(M + pad, Gs, Fh)), device=dev)
    call("conan_rbf_wgrad", ptr(dh1), ptr(dist), M + pad, ptr(off), Gs, coeff, Fh, ptr(md), ptr(dWc), ptr(dbc), ptr(wsc), stream_ptr())
    assert rel(dW.cpu().numpy(), dWc.cpu().numpy()) < 1e-5 and rel(db.cpu().numpy(), dbc.cpu().numpy()) < 1e-5
    # slab form + batched reduction; bitwise reproducibility
    dW2, db2 = torch.empty_like(dW), torch.empty_like(db)
    call("conan_filter_bwd", ptr(g), ptr(h1), ptr(dist), M + pad, ptr(off), Gs, coeff, ptr(w2), Fh, ptr(md), None, None, ptr(ws), gm, stream_ptr())
    job = (WgradJob * 1)()
    job[0].ws, job[0].dW, job[0].dbias = ws.data_ptr(), dW2.data_ptr(), db2.data_ptr()
    job[0].M, job[0].K, job[0].N, job[0].slices = M + pad, Gs, Fh, int(lib().conan_filter_bwd_slices(M + pad))
    call("conan_wgrad_reduce_batch", job, 1, stream_ptr())
    assert torch.equal(dW, dW2) and torch.equal(db, db2)


@pytest.mark.parametrize("gscale", [1.0, 3e-7, 4e4], ids=["g1", "tiny_grad", "huge_grad"])
@pytest.mark.parametrize("M,Gs", [(1, 50), (31, 50), (33, 50), (4133, 50), (40000, 50), (300000, 50), (3000, 20), (2500, 63)])
def test_filter_network_backward_in_one_pass(M, Gs, gscale):
    """conan_filter_bwd2: both layers' weight / bias gradients of the filter network from ONE pass over g and h1 (dw2 = g^T h1, db2 = colsum g,
    dh1 = (g w2) * ssp'(h1) on chip, dw1 = dh1^T rbf, db1 = colsum dh1), against the fp64 formulas and against the two kernels it replaces
    (conan_filter_bwd + conan_linear_wgrad_scaled: same arithmetic, summed in another order); rows beyond the device-side count are masked;
    the slab form reduced by two conan_wgrad_reduce_batch jobs gives the same bits; repeat runs are bitwise equal.  M = 300000 makes every
    workgroup walk several tiles (double-buffered images, 256 workgroups)."""
    from conan_fgw_amd._lib import WgradJob, call, lib, ptr, stream_ptr
    Fh, pad = 128, 37
    assert lib().conan_filter_bwd2_supported(Gs, Fh) == 1 and lib().conan_filter_bwd2_supported(64, Fh) == 0 and lib().conan_filter_bwd2_supported(50, 64) == 0
    gen = torch.Generator().manual_seed(M + Gs)
    g = (torch.randn(M + pad, Fh, generator=gen) * gscale).to(dev)
    g[M:] = float("nan")                                                                    # rows beyond the count must never be touched
    gmax = g[:M].abs().max().reshape(1).contiguous()
    h1 = (torch.rand(M + pad, Fh, generator=gen) * 3 - 0.6).to(dev)                        # ssp output range (> -ln 2)
    h1[M:] = float("nan")
    dist = (torch.rand(M + pad, generator=gen) * 10).to(dev)
    w2 = (torch.randn(Fh, Fh, generator=gen) / 11).to(dev)
    off = torch.linspace(0, 10, Gs).to(dev)
    coeff = -0.5 / float(off[1] - off[0]) ** 2
    md = torch.tensor([M], dtype=torch.int32, device=dev)
    ws = torch.empty(int(lib().conan_filter_bwd2_ws(M + pad, Gs, Fh)), device=dev)
    out = lambda: (torch.empty(Fh, Gs, device=dev), torch.empty(Fh, device=dev), torch.empty(Fh, Fh, device=dev), torch.empty(Fh, device=dev))
    dW1, db1, dW2, db2 = out()
    args = (ptr(g), ptr(h1), ptr(dist), M + pad, ptr(off), Gs, coeff, ptr(w2), Fh, ptr(md), ptr(gmax))
    call("conan_filter_bwd2", *args, ptr(dW1), ptr(db1), ptr(dW2), ptr(db2), ptr(ws), stream_ptr())
    g64, h64 = g[:M].double(), h1[:M].double()
    dh = (g64 @ w2.double()) * (1 - 0.5 * torch.exp(-h64))
    rbf = torch.exp(coeff * (dist[:M, None].double() - off[None].double()) ** 2)
    for got, want in ((dW1, dh.T @ rbf), (db1, dh.sum(0)), (dW2, g64.T @ h64), (db2, g64.sum(0))):
        assert torch.isfinite(got).all()
        assert rel(got.double().cpu().numpy(), want.cpu().numpy()) < 1e-5
    # the pair of kernels it replaces
    wsa = torch.empty(int(lib().conan_filter_bwd_ws(M + pad, Gs, Fh)), device=dev)
    dWa, dba = torch.empty_like(dW1), torch.empty_like(db1)
    g0, h0 = g.clone(), h1.clone(); g0[M:] = 0; h0[M:] = 0
    call("conan_filter_bwd", ptr(g0), ptr(h0), ptr(dist), M + pad, ptr(off), Gs, coeff, ptr(w2), Fh, ptr(md), ptr(dWa), ptr(dba), ptr(wsa), ptr(gmax), stream_ptr())
    assert rel(dW1.cpu().numpy(), dWa.cpu().numpy()) < 1e-5 and rel(db1.cpu().numpy(), dba.cpu().numpy()) < 1e-5
    wsb = torch.empty(int(lib().conan_linear_wgrad_ws(M + pad, Fh, Fh)), device=dev)
    dWb, dbb = torch.empty_like(dW2), torch.empty_like(db2)
    call("conan_linear_wgrad_scaled", ptr(g0), ptr(h0), M + pad, Fh, Fh, ptr(md), ptr(dWb), ptr(dbb), ptr(wsb), ptr(gmax), stream_ptr())
    assert rel(dW2.cpu().numpy(), dWb.cpu().numpy()) < 1e-5 and rel(db2.cpu().numpy(), dbb.cpu().numpy()) < 1e-5
    # slab form + batched reduction; bitwise reproducibility
    e1, eb1, e2, eb2 = out()
    call("conan_filter_bwd2", *args, None, None, None, None, ptr(ws), stream_ptr())
    slices = int(lib().conan_filter_bwd2_slices(M + pad))
    job = (WgradJob * 2)()
    job[0].ws, job[0].dW, job[0].dbias = ws.data_ptr(), e1.data_ptr(), eb1.data_ptr()
    job[0].M, job[0].K, job[0].N, job[0].slices = M + pad, Gs, Fh, slices
    job[1].ws, job[1].dW, job[1].dbias = ws.data_ptr() + 4 * slices * (Fh * Gs + Fh), e2.data_ptr(), eb2.data_ptr()
    job[1].M, job[1].K, job[1].N, job[1].slices = M + pad, Fh, Fh, slices
    call("conan_wgrad_reduce_batch", job, 2, stream_ptr())
    assert torch.equal(dW1, e1) and torch.equal(db1, eb1) and torch.equal(dW2, e2) and torch.equal(db2, eb2)


def test_global_gradient_scale_has_the_documented_floor_under_an_outlier():
    """The fp16-plane kernels put ALL of g on one power-of-two scale taken from max |g| (include/conan_fgw_hip.h, conan_filter_bwd2).  One
    element a million times larger than the rest pushes the rest towards fp16's subnormal spacing: every element of g then carries an absolute
    error of up to ~2^-29 max|g| (no longer 2^-22 relative), and the results inherit it — a bounded, documented floor, not fp32-class.
    This test pins that floor (so that a change of the scale policy that makes it worse is noticed) and that WITHOUT the outlier the same
    data are fp32-class.  Gradients of a trained model span a few decades, not six."""
    from conan_fgw_amd._lib import call, lib, ptr, stream_ptr
    M, Fh, Gs = 5000, 128, 50
    gen = torch.Generator().manual_seed(9)
    g = torch.randn(M, Fh, generator=gen).to(dev)
    h1 = (torch.rand(M, Fh, generator=gen) * 3 - 0.6).to(dev)
    dist = (torch.rand(M, generator=gen) * 10).to(dev)
    w2 = (torch.randn(Fh, Fh, generator=gen) / 11).to(dev)
    off = torch.linspace(0, 10, Gs).to(dev); coeff = -0.5 / float(off[1] - off[0]) ** 2
    md = torch.tensor([M], dtype=torch.int32, device=dev)
    ws = torch.empty(int(lib().conan_filter_bwd2_ws(M, Gs, Fh)), device=dev)

    def run(gt):
        gmax = gt.abs().max().reshape(1).contiguous()
        dW1, db1, dW2, db2 = torch.empty(Fh, Gs, device=dev), torch.empty(Fh, device=dev), torch.empty(Fh, Fh, device=dev), torch.empty(Fh, device=dev)
        call("conan_filter_bwd2", ptr(gt), ptr(h1), ptr(dist), M, ptr(off), Gs, coeff, ptr(w2), Fh, ptr(md), ptr(gmax), ptr(dW1), ptr(db1), ptr(dW2),
             ptr(db2), ptr(ws), stream_ptr())
        return dW2.double().cpu(), float(gmax)
    got, _ = run(g)
    want = (g.double().T @ h1.double()).cpu()
    assert rel(got.numpy(), want.numpy()) < 1e-5                                             # ordinary data: fp32-class
    go = g.clone(); go[17, 5] = 1.0e6 * float(g.abs().max())
    got, gmax = run(go)
    want = (go.double().T @ h1.double()).cpu()
    floor = gmax * 2.0 ** -28 * h1.abs().double().sum(0).cpu()                                # per column k: sum_e |h1[e, k]| errors of <= 2^-28 max|g| each
    assert bool(((got - want).abs() <= floor[None, :] + 1e-6 * want.abs()).all())
    assert rel(got[5].numpy(), want[5].numpy()) < 1e-6                                       # the outlier's own row of dW2 is dominated by it: exact to fp32
    others = torch.ones(Fh, dtype=torch.bool); others[5] = False
    assert rel(got[others].numpy(), want[others].numpy()) > 1e-5                             # ... and the other rows ARE degraded (this is the floor, not a bug)


@pytest.mark.parametrize("M,K,N,act,w_kn", [(3000, 128, 384, 0, 0), (3000, 384, 128, 0, 1), (1777, 128, 256, 3, 0), (2048, 256, 128, 1, 1),
                                            (999, 512, 256, 1, 0), (640, 192, 320, 0, 0), (1500, 256, 256, 2, 1),
                                            (1500, 256, 128, 0, 0), (901, 384, 128, 0, 0), (70001, 256, 128, 0, 1), (1, 384, 128, 0, 1)])      # (the contractions k_linear_sum16 takes)
def test_linear_wide_layers_are_tiled_into_strided_chunks(M, K, N, act, w_kn):
    """Layers wider than 128 (ViSNet's 128->256/384 projections and their transposes, the 512/256 classification SchNet) run as
    (n chunk, k chunk) launches of the register-streamed kernel: bias with the first k chunk, activation / residual with the last,
    partial sums accumulated in place.  Checked against fp64 for every activation code and both weight orientations."""
    from conan_fgw_amd._lib import call, ptr, stream_ptr
    dev = torch.device("cuda:0")
    gen = torch.Generator().manual_seed(M + K + N + act)
    x = torch.randn(M + 9, K, generator=gen).to(dev)
    w = (torch.randn(K, N, generator=gen) if w_kn else torch.randn(N, K, generator=gen)).mul_(1.0 / math.sqrt(K)).to(dev)
    b = torch.randn(N, generator=gen).to(dev)
    res = torch.randn(M + 9, N, generator=gen).to(dev)
    md = torch.tensor([M], dtype=torch.int32, device=dev)
    y = torch.full((M + 9, N), float("nan"), device=dev)
    call("conan_linear_fwd", ptr(x), ptr(w), ptr(b), ptr(res), M + 9, K, N, w_kn, act, ptr(md), ptr(y), stream_ptr())
    xd, wd = x[:M].double(), w.double()
    pre = xd @ (wd if w_kn else wd.T) + b.double()
    r = res[:M].double()
    if act == 0:
        ref = pre + r
    elif act == 1:
        ref = F.softplus(pre) - math.log(2.0) + r
    elif act == 3:
        ref = pre * torch.sigmoid(pre) + r
    else:
        ref = pre * (1.0 - 0.5 * torch.exp(-r))
    assert rel(y[:M].double().cpu().numpy(), ref.cpu().numpy()) < 2e-6
    assert torch.isnan(y[M:]).all()                                 # rows beyond the device-side count are not touched


@pytest.mark.parametrize("M,nsrc,w_kn,mags", [(1, 2, 1, (1.0, 1.0)), (33, 3, 1, (1.0, 1.0, 1.0)), (5000, 3, 1, (1.0e-6, 3.0, 1.0e-3)), (2049, 2, 0, (1.0e4, 1.0e-4)),
                                              (70000, 3, 1, (1.0, 0.0, 1.0e-9)), (300, 3, 0, (0.0, 0.0, 0.0))])
def test_sum_of_contractions_in_one_launch_matches_fp64_chunk_by_chunk(M, nsrc, w_kn, mags):
    """conan_linear_sum_fwd: y = sum_c x_c W_c^T + b + residual with the sum kept in the accumulators (the input gradient of ViS_MP's dk / dv /
    f_proj projections of one f_ij, torch_geometric_visnet.py:600-604,637-640).  The chunks of a row share one power-of-two unit, so the bound
    is the one a single scale for the concatenated row gives: every chunk's own product to 2e-6 of the LARGEST chunk's, and the sum to 2e-6 of
    itself for ordinary data.  Inputs with their own row pitch (columns of a wider tensor), rows beyond the device-side count untouched,
    residual aliasing y, all-zero chunks."""
    import ctypes
    from conan_fgw_amd._lib import call, ptr, stream_ptr
    gen = torch.Generator().manual_seed(M + nsrc)
    wide = torch.randn(M + 5, 128 * nsrc + 64, generator=gen).to(dev)                   # chunk c = columns [32 + 128 c, 32 + 128 (c + 1)) — 128-byte aligned, pitch != 128
    rowmag = torch.exp(4.0 * torch.randn(M + 5, 1, generator=gen)).to(dev)              # rows of very different size
    xs = []
    for c in range(nsrc):
        v = wide[:, 32 + 128 * c: 32 + 128 * (c + 1)]
        v.mul_(mags[c] * rowmag)
        xs.append(v)
    ws = [(torch.randn(128, 128, generator=gen) * (0.1 * 3.0 ** c)).to(dev) for c in range(nsrc)]
    b = torch.randn(128, generator=gen).to(dev)
    md = torch.tensor([M], dtype=torch.int32, device=dev)
    for with_extras in (False, True):
        y = torch.full((M + 5, 128), float("nan"), device=dev)
        res0 = torch.randn(M + 5, 128, generator=gen).to(dev)
        if with_extras:
            y.copy_(res0)                                                                   # residual == y
        call("conan_linear_sum_fwd", (ctypes.c_void_p * nsrc)(*[v.data_ptr() for v in xs]), (ctypes.c_int * nsrc)(*[wide.stride(0)] * nsrc),
             (ctypes.c_void_p * nsrc)(*[w.data_ptr() for w in ws]), nsrc, w_kn, ptr(b) if with_extras else None, ptr(y) if with_extras else None,
             M + 5, 128, ptr(md), ptr(y), stream_ptr())
        parts = [xs[c][:M].double() @ (ws[c].double() if w_kn else ws[c].double().T) for c in range(nsrc)]
        ref = sum(parts)
        if with_extras:
            ref = ref + b.double() + res0[:M].double()
            assert torch.equal(y[M:], res0[M:])                                             # rows beyond the device-side count are not touched
        else:
            assert torch.isnan(y[M:]).all()
        got = y[:M].double()
        scale = sum(p.abs() for p in parts).max(dim=1, keepdim=True).values + ((res0[:M].abs().double() + b.abs().double()) if with_extras else 0.0) + 1e-300
        assert float(((got - ref).abs() / scale).max()) < 2e-6                              # row by row: relative to the row's largest term
        if all(m == 1.0 for m in mags):
            assert rel(got.cpu().numpy(), ref.cpu().numpy()) < 2e-6


@pytest.mark.parametrize("n", [1, 256, 1000])
def test_mse_loss_and_its_gradient_in_one_launch(n):
    g = torch.Generator().manual_seed(n)
    pred, y = torch.randn(n, 1, generator=g), torch.randn(n, 1, generator=g)
    pd = pred.to(dev).requires_grad_(True)
    loss = ops.mse_loss(pd, y.to(dev))
    (3.0 * loss).backward()
    p64 = pred.double().requires_grad_(True)
    ref = F.mse_loss(p64, y.double())
    (3.0 * ref).backward()
    assert loss.shape == () and abs(float(loss) - float(ref)) <= 1e-6 * abs(float(ref))
    assert rel(pd.grad.cpu(), p64.grad) < 1e-6
    l2 = ops.mse_loss(pd.detach(), y.to(dev))
    assert float(l2) == float(loss)                                   # fixed summation order
    with pytest.raises(RuntimeError):
        ops.mse_loss(pd, y.to(dev)[:, 0])


def test_embedding_out_of_range_index_is_nan_not_out_of_bounds():
    """torch.nn.Embedding device-asserts on an index outside [0, num_embeddings); here the row is NaN and the backward skips it."""
    w = torch.randn(100, 64, device=dev, requires_grad=True)
    z = torch.tensor([1, 6, 100, 8, -3, 99], device=dev)
    out = ops.embedding(z, w, 0)
    assert torch.isnan(out[2]).all() and torch.isnan(out[4]).all()
    ok = [0, 1, 3, 5]
    assert torch.equal(out[ok], w.detach()[z[ok]])
    out.nan_to_num().sum().backward()
    ref = torch.zeros(100, 64, device=dev)
    ref[[1, 6, 8, 99]] = 1.0
    assert torch.equal(w.grad, ref)


def test_shifted_softplus_module_runs_on_the_hip_kernel():
    from conan_fgw_amd.schnet import ShiftedSoftplus
    x = torch.linspace(-30, 30, 1001, device=dev).requires_grad_(True)
    y = ShiftedSoftplus()(x)
    r = torch.nn.functional.softplus(x.detach().double()) - np.log(2.0)
    assert float((y.detach().double() - r).abs().max()) < 3e-6
    y.sum().backward()
    assert float((x.grad.double() - torch.sigmoid(x.detach().double())).abs().max()) < 1e-6
