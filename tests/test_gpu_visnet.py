"""GPU parity of the ViSNet forward path against the reference's own ViSNet classes (tests/golden/visnet_ref_*.npz) and,
at the production width H=128 on BACE-shaped conformers, against the fp64 oracle.  Tolerance 1e-4 relative (north_star)."""
import os

import numpy as np
import pytest
import torch

from helpers import golden_files, rel
from conan_fgw_amd.synthetic import make_batch
from conan_fgw_amd.visnet import ViSNet
from oracle.visnet import ViSNetOracle

pytestmark = pytest.mark.gpu
dev = torch.device("cuda:0")
CASES = golden_files("visnet_ref_")


@pytest.mark.parametrize("path", CASES, ids=[os.path.basename(p)[11:-4] for p in CASES])
def test_against_reference_visnet(path):
    g = np.load(path)
    m = ViSNet(dev, hidden_channels=int(g["hidden"]))
    m.load_state_dict({k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sd:")}, strict=True)
    m = m.to(dev)
    K = int(g["K"])
    z, pos, batch = (torch.from_numpy(g[k]).to(dev) for k in ("z", "pos", "batch"))
    gp = torch.from_numpy(np.concatenate([[0], np.cumsum(np.bincount(g["batch"]))]).astype(np.int32)).to(dev)
    xs, vs = m.representation_model(z, pos, gp, int(g["batch"].max()) + 1)
    assert rel(xs.detach().cpu().numpy(), g["r64_x"]) < 2e-5 and rel(vs.detach().cpu().numpy(), g["r64_vec"]) < 2e-5
    assert rel(m(z, pos, batch).detach().cpu().numpy(), g["r64_forward"]) < 2e-5
    h, hb = m.forward_3d_bary(z, pos, batch)
    assert rel(h.detach().cpu().numpy(), g["r64_h"]) < 2e-5 and rel(hb.detach().cpu().numpy(), g["r64_h_bary_nodes"]) < 2e-5
    ei, _ = m.interaction_graph(pos, batch)
    assert np.array_equal(ei.detach().cpu().numpy(), g["edge_index"])
    h3d, hbary = m.forward_w_barycenter(z, pos, K, batch)
    assert rel(h3d.detach().cpu().numpy(), g["r64_h_3d"]) < 2e-5
    for tag in ("r32", "r64"):
        assert rel(hbary.detach().cpu().numpy(), g[tag + "_h_bary"]) < 1e-4


def test_production_width_vs_oracle_and_invariance():
    b = make_batch("bace", 2, 5, seed=77)                       # n ~ 65 atoms: cap-32 truncation with self loops is active
    torch.manual_seed(5)
    m = ViSNet(dev, hidden_channels=128).to(dev)
    with torch.no_grad():
        for p in m.parameters():
            p.add_(0.02 * torch.randn_like(p))
    ref = ViSNetOracle(128)
    ref.load_state_dict({k: v.detach().cpu() for k, v in m.state_dict().items()}, strict=True)
    ref = ref.double()
    z, pos, batch = torch.from_numpy(b.z), torch.from_numpy(b.pos), torch.from_numpy(b.batch)
    h3d, hbary = m.forward_w_barycenter(z.to(dev), pos.to(dev), 5, batch.to(dev), num_graphs=b.num_graphs, max_nodes=b.max_nodes)
    with torch.no_grad():
        r3, rb = ref.forward_w_barycenter(z, pos.double(), 5, batch)
    assert rel(h3d.detach().cpu().numpy(), r3.numpy()) < 2e-5
    assert rel(hbary.detach().cpu().numpy(), rb.numpy()) < 1e-4
    # E(3) invariance of the scalar head under a random rotation + translation
    Q, _ = np.linalg.qr(np.random.RandomState(1).normal(size=(3, 3)))
    pos2 = (b.pos @ Q.T.astype(np.float32) + np.float32([1.5, -2.0, 0.7])).astype(np.float32)
    out1 = m(z.to(dev), pos.to(dev), batch.to(dev))
    out2 = m(z.to(dev), torch.from_numpy(pos2).to(dev), batch.to(dev))
    assert rel(out2.detach().cpu().numpy(), out1.detach().cpu().numpy()) < 1e-4


@pytest.mark.parametrize("H,shape,B,K", [(32, "esol", 2, 3), (64, "bace", 2, 3), (128, "esol", 2, 5), (512, "esol", 2, 3), (256, "bace", 1, 2)])
def test_backward_matches_oracle_autograd(H, shape, B, K):
    """Gradients of every trainable ViSNet parameter (trunk, both heads, atomref priors) through forward_w_barycenter,
    against torch autograd of the fp64 oracle (the oracle's FGW backward is the reference's: T held constant).
    H = 512 is the width the reference's classification head instantiates its backbone with (common.py:444-446, feat_dim = 512; not
    reachable from the reference's CLI with ViSNet, reachable from the class API): attention runs in blocks of 128 channels."""
    b = make_batch(shape, B, K, seed=55, box=7.0 if shape == "esol" else None)
    torch.manual_seed(H)
    m = ViSNet(dev, hidden_channels=H).to(dev)
    with torch.no_grad():
        for p in m.parameters():
            p.add_(0.02 * torch.randn_like(p))
    ref = ViSNetOracle(H)
    ref.load_state_dict({k: v.detach().cpu() for k, v in m.state_dict().items()}, strict=True)
    ref = ref.double()
    z, pos, batch = torch.from_numpy(b.z), torch.from_numpy(b.pos), torch.from_numpy(b.batch)
    g1 = torch.randn(b.num_graphs, H // 2, generator=torch.Generator().manual_seed(1))
    g2 = torch.randn(b.num_graphs, H // 2, generator=torch.Generator().manual_seed(2))
    h3, hb = m.forward_w_barycenter(z.to(dev), pos.to(dev), K, batch.to(dev))
    ((h3 * g1.to(dev)).sum() + (hb * g2.to(dev)).sum()).backward()
    r3, rb = ref.forward_w_barycenter(z, pos.double(), K, batch)
    ((r3 * g1.double()).sum() + (rb * g2.double()).sum()).backward()
    assert rel(h3.detach().cpu().numpy(), r3.detach().numpy()) < 2e-5
    refp = dict(ref.named_parameters())
    checked = 0
    for name, p in m.named_parameters():
        q = refp[name]
        if q.grad is None:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, name
            continue
        assert p.grad is not None, name
        gn = float(q.grad.norm())
        if gn == 0.0:
            assert float(p.grad.abs().max()) < 1e-6, name
            continue
        assert rel(p.grad.detach().cpu().numpy(), q.grad.numpy()) < 1e-4, name      # north_star's bar (measured: 3e-6 .. 8e-6 over the five widths)
        checked += 1
    assert checked > 100


def test_edge_buffer_tails_are_never_read():
    """The worst-case-sized edge buffers are handed out uninitialised (visnet_ops._tail0_shape): outputs and every parameter gradient
    must be bitwise independent of what the allocator's recycled blocks contain — the same forward + backward is run after the
    caching allocator's free blocks were filled with zeros and after they were filled with NaNs."""
    b = make_batch("bace", 2, 3, seed=31)                        # cap-32 truncation active: max_edges is well above the edge count
    z, pos, batch = torch.from_numpy(b.z).to(dev), torch.from_numpy(b.pos).to(dev), torch.from_numpy(b.batch).to(dev)
    torch.manual_seed(9)
    m = ViSNet(dev, hidden_channels=64).to(dev)

    def run(fill):
        junk = [torch.full((1 << 22,), fill, device=dev) for _ in range(24)]           # 384 MiB of recycled blocks in many sizes' pools
        junk += [torch.full((n,), fill, device=dev) for n in (1 << 12, 1 << 14, 1 << 16, 1 << 18, 1 << 20) for _ in range(8)]
        del junk
        for p in m.parameters():
            p.grad = None
        h3, hb = m.forward_w_barycenter(z, pos, 3, batch)
        (h3.sum() + hb.sum()).backward()
        torch.cuda.synchronize()
        return [h3.detach().clone(), hb.detach().clone()] + [p.grad.detach().clone() for p in m.parameters() if p.grad is not None]

    a, c = run(0.0), run(float("nan"))
    assert len(a) == len(c) and len(a) > 50
    for u, v in zip(a, c):
        assert torch.isfinite(v).all()
        assert torch.equal(u, v)


def test_unread_tails_poisoned_with_nan_change_nothing():
    """The deterministic form of the test above: ops.POISON_UNREAD_TAILS makes every buffer handed out under the "rows beyond the device-side
    count are never read" contract (ops.unread_rows: the edge-level activations of visnet_ops and the input gradient of s_proj, the one
    `lin()` call site that opts in) start as NaN.  A whole stage-2 training step on the ViSNet backbone — forward, loss, backward, every
    parameter gradient — must give the same bits with and without the poison."""
    import types
    from conan_fgw_amd import ops
    from conan_fgw_amd.head import EmbeddingsWithGATAggregationBaryCenter
    from conan_fgw_amd.synthetic import make_bond_graph
    b = make_batch("bace", 3, 3, seed=33)
    bg = make_bond_graph(b, seed=34)
    torch.manual_seed(10)
    m = EmbeddingsWithGATAggregationBaryCenter(3, dev, model_name="visnet").to(dev)
    data = types.SimpleNamespace(z=torch.from_numpy(b.z).to(dev), pos=torch.from_numpy(b.pos).to(dev), batch=torch.from_numpy(b.batch).to(dev),
                                 x=torch.from_numpy(bg.x).to(dev), edge_index=torch.from_numpy(bg.edge_index).to(dev), edge_attr=torch.from_numpy(bg.edge_attr).to(dev))
    cidx = m.create_aggregation_index(b.num_graphs, dev)

    def run(poison):
        ops.POISON_UNREAD_TAILS = poison
        try:
            for p in m.parameters():
                p.grad = None
            y = m(data, cidx, data.batch)
            y.square().mean().backward()
            torch.cuda.synchronize()
        finally:
            ops.POISON_UNREAD_TAILS = False
        return [y.detach().clone()] + [p.grad.detach().clone() for p in m.parameters() if p.grad is not None]

    a, c = run(False), run(True)
    assert len(a) == len(c) and len(a) > 100
    for u, v in zip(a, c):
        assert torch.isfinite(v).all() and torch.equal(u, v)


@pytest.mark.parametrize("M,n,tap,bias", [(3000, 3, True, True), (70001, 3, True, False), (66000, 2, False, True), (70001, 3, False, True)])
def test_layers_of_one_input_forward_and_backward_match_fp64(M, n, tap, bias):
    """visnet_ops.multi_lin — ViS_MP's dk / dv / f_proj of one f_ij (torch_geometric_visnet.py:600-604,637-640) and q / k / v of one x (:596-598):
    outputs, the summed input gradient (with the gradient of the handed-through x as its seed), weight and bias gradients against fp64 autograd.
    From 65 536 rows on the three of them run through the round-5 kernels — all layers of a tile in one workgroup (k_linear_fan16), the input
    gradients summed in the accumulators (k_linear_sum16), the weight gradients over one staging of x (k_wgrad_lds_shared); below that through
    the side-by-side and chained forms.  A device-side row count masks the tail of worst-case-sized edge buffers."""
    from conan_fgw_amd import visnet_ops as vo
    gen = torch.Generator().manual_seed(M + n)
    x = (torch.randn(M + 7, 128, generator=gen) * torch.exp(2.0 * torch.randn(M + 7, 1, generator=gen))).to(dev).requires_grad_(True)
    mods = [torch.nn.Linear(128, 128, bias=bias).to(dev) for _ in range(n)]
    md = torch.tensor([M], dtype=torch.int32, device=dev)
    gys = [torch.randn(M + 7, 128, generator=gen).to(dev) * (3.0 ** q) for q in range(n)]
    gtap = torch.randn(M + 7, 128, generator=gen).to(dev)
    outs = vo.multi_lin(x, mods, False, md, tap=tap)
    loss = sum((o[:M] * g[:M]).sum() for o, g in zip(outs[:n], gys))
    if tap:
        loss = loss + (outs[n][:M] * gtap[:M]).sum()
    loss.backward()
    xd = x.detach()[:M].double().requires_grad_(True)
    wd = [m.weight.detach().double().requires_grad_(True) for m in mods]
    bd = [m.bias.detach().double().requires_grad_(True) if bias else None for m in mods]
    refs = [xd @ w.T + (b if b is not None else 0.0) for w, b in zip(wd, bd)]
    lref = sum((o * g[:M].double()).sum() for o, g in zip(refs, gys))
    if tap:
        lref = lref + (xd * gtap[:M].double()).sum()
    lref.backward()
    for o, r in zip(outs[:n], refs):
        assert rel(o[:M].detach().double().cpu().numpy(), r.detach().cpu().numpy()) < 2e-6
    assert rel(x.grad[:M].double().cpu().numpy(), xd.grad.cpu().numpy()) < 2e-6
    for m, w, b in zip(mods, wd, bd):
        assert rel(m.weight.grad.double().cpu().numpy(), w.grad.cpu().numpy()) < 2e-6
        if bias:
            assert rel(m.bias.grad.double().cpu().numpy(), b.grad.cpu().numpy()) < 2e-6
