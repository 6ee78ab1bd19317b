"""GPU parity of the drop-in SchNetNoSum against outputs of the reference's own class (tests/golden/schnet_ref_*.npz)
and against the fp64 oracle at ragged shapes.  Tolerance: 1e-4 relative (north_star), neighbour indices exact."""
import os

import numpy as np
import pytest
import torch

from helpers import golden_files, golden_state_dict, rel
from conan_fgw_amd.schnet import SchNetNoSum
from conan_fgw_amd.synthetic import make_batch
from oracle.schnet import SchNetNoSumOracle

pytestmark = pytest.mark.gpu
dev = torch.device("cuda:0")
CASES = golden_files("schnet_ref_")


def _model(g):
    H = int(g["hidden"])
    m = SchNetNoSum(dev, hidden_channels=H, num_filters=H, num_interactions=3)
    m.load_state_dict(golden_state_dict(g), strict=True)
    return m.to(dev)


@pytest.mark.parametrize("path", CASES, ids=[os.path.basename(p)[11:-4] for p in CASES])
def test_against_reference_class(path):
    g = np.load(path)
    m = _model(g)
    K = int(g["K"])
    z, pos, batch = (torch.from_numpy(g[k]).to(dev) for k in ("z", "pos", "batch"))
    ei, ew = m.interaction_graph(pos, batch)
    assert np.array_equal(ei.cpu().numpy(), g["edge_index"])                       # bit-exact neighbour indices
    assert rel(ew.cpu().numpy(), g["r64_edge_weight"]) < 1e-6
    out = m(z, pos, batch)
    assert rel(out.detach().cpu().numpy(), g["r64_forward"]) < 1e-5
    h, hb = m.forward_3d_bary(z, pos, batch)
    assert rel(h.detach().cpu().numpy(), g["r64_h"]) < 1e-5 and rel(hb.detach().cpu().numpy(), g["r64_h_bary_nodes"]) < 1e-5
    h3d, hbary = m.forward_w_barycenter(z, pos, K, batch)
    assert h3d.shape == g["r32_h_3d"].shape and hbary.shape == g["r32_h_bary"].shape
    assert rel(h3d.detach().cpu().numpy(), g["r64_h_3d"]) < 1e-5
    for tag in ("r32", "r64"):
        assert rel(hbary.detach().cpu().numpy(), g[tag + "_h_bary"]) < 1e-4         # FGW readout within 1e-4 of the reference
    hb_np = hbary.detach().cpu().numpy().reshape(-1, K, hbary.shape[1])
    assert np.all(hb_np == hb_np[:, :1])                                            # rows identical within a molecule
    # gradients of the reference's autograd through trunk + FGW
    ((h3d * torch.from_numpy(g["gw_h3d"]).float().to(dev)).sum() + (hbary * torch.from_numpy(g["gw_hbary"]).float().to(dev)).sum()).backward()
    for name, p in m.named_parameters():
        ref = g["r32_grad:" + name]
        if ref.size == 0:
            continue
        assert rel(p.grad.cpu().numpy(), ref) < 2e-3, name
    if "r64_grad:lin1_bary.weight" in g.files:
        for name, p in m.named_parameters():
            assert rel(p.grad.cpu().numpy(), g["r64_grad:" + name]) < 1e-4, name
    else:
        # the wider goldens carry the reference's fp32 gradients only (its own fp32 FGW loop sits 1e-3 from fp64: hence 2e-3 above); the 1e-4 bar
        # is held against the fp64 oracle with the same weights, run here (the oracle's wiring is pinned by the r64 forward fields of this file)
        H = int(g["hidden"])
        ref = SchNetNoSumOracle(H, H, 3)
        ref.load_state_dict(golden_state_dict(g), strict=True)
        ref = ref.double()
        r3, rb = ref.forward_w_barycenter(torch.from_numpy(g["z"]), torch.from_numpy(g["pos"]).double(), K, torch.from_numpy(g["batch"]))
        ((r3 * torch.from_numpy(g["gw_h3d"]).double()).sum() + (rb * torch.from_numpy(g["gw_hbary"]).double()).sum()).backward()
        refg = dict(ref.named_parameters())
        checked = 0
        for name, p in m.named_parameters():
            if p.grad is None or refg[name].grad is None:
                continue
            assert rel(p.grad.cpu().numpy(), refg[name].grad.numpy()) < 1e-4, name
            checked += 1
        assert checked >= 20


def test_hints_avoid_host_sync_and_match():
    b = make_batch("esol", 6, 5, seed=33)
    torch.manual_seed(5)
    m = SchNetNoSum(dev, hidden_channels=128, num_filters=128, num_interactions=3).to(dev)
    z, pos, batch = (torch.from_numpy(a).to(dev) for a in (b.z, b.pos, b.batch))
    a3, ab = m.forward_w_barycenter(z, pos, 5, batch)
    b3, bb = m.forward_w_barycenter(z, pos, 5, batch, num_graphs=b.num_graphs, max_nodes=b.max_nodes)
    assert torch.equal(a3, b3) and torch.equal(ab, bb)
    # ragged batch vs the fp64 oracle with the same weights
    ref = SchNetNoSumOracle(128, 128, 3)
    ref.load_state_dict({k: v.detach().cpu() for k, v in m.state_dict().items()})
    ref = ref.double()
    r3, rb = ref.forward_w_barycenter(torch.from_numpy(b.z), torch.from_numpy(b.pos).double(), 5, torch.from_numpy(b.batch))
    assert rel(a3.detach().cpu().numpy(), r3.detach().numpy()) < 1e-5
    assert rel(ab.detach().cpu().numpy(), rb.detach().numpy()) < 1e-4


def test_compute_barycenter_accepts_reference_edge_index():
    b = make_batch("esol", 3, 5, seed=34)
    m = SchNetNoSum(dev, hidden_channels=64, num_filters=64, num_interactions=1).to(dev)
    z, pos, batch = (torch.from_numpy(a).to(dev) for a in (b.z, b.pos, b.batch))
    _, hb = m.forward_3d_bary(z, pos, batch)
    ei, _ = m.interaction_graph(pos, batch)
    n1, f1 = m._compute_barycenter(hb, ei, batch, 3, 5)
    h3, f2 = m.forward_w_barycenter(z, pos, 5, batch)
    assert torch.equal(f1, f2) and n1.shape == (15, 32)
