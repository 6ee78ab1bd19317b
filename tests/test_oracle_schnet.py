"""The oracle's restatement of the reference's SchNet wiring (oracle/schnet.py) against outputs of the reference's own
`SchNetNoSum` class (tests/golden/schnet_ref_*.npz, made by tests/golden/make_model_golden.py)."""
import os

import numpy as np
import pytest
import torch

from helpers import golden_files, golden_state_dict, rel
from oracle.schnet import SchNetNoSumOracle

CASES = golden_files("schnet_ref_")


def _load(path, dtype):
    g = np.load(path)
    H = int(g["hidden"])
    m = SchNetNoSumOracle(hidden_channels=H, num_filters=H, num_interactions=3)
    missing = m.load_state_dict(golden_state_dict(g), strict=True)
    m = m.to(dtype)
    return g, m, torch.from_numpy(g["z"]), torch.from_numpy(g["pos"]).to(dtype), torch.from_numpy(g["batch"])


@pytest.mark.parametrize("path", CASES, ids=[os.path.basename(p)[11:-4] for p in CASES])
def test_state_dict_surface(path):
    g = np.load(path)
    H = int(g["hidden"])
    m = SchNetNoSumOracle(hidden_channels=H, num_filters=H, num_interactions=3)
    keys = set(m.state_dict().keys())
    assert keys == set(golden_state_dict(g).keys())
    assert len(keys) == 49                                  # 49 state_dict keys


@pytest.mark.parametrize("path", CASES, ids=[os.path.basename(p)[11:-4] for p in CASES])
@pytest.mark.parametrize("tag,dtype,tol", [("r64", torch.float64, 1e-9), ("r32", torch.float32, 2e-5)])
def test_oracle_matches_reference_class(path, tag, dtype, tol):
    g, m, z, pos, batch = _load(path, dtype)
    K = int(g["K"])
    with torch.no_grad():
        ei, ew = m.interaction_graph(pos, batch)
        assert np.array_equal(ei.numpy(), g["edge_index"])              # integer output: exact
        assert rel(ew.numpy(), g[tag + "_edge_weight"]) < tol
        assert rel(m(z, pos, batch).numpy(), g[tag + "_forward"]) < tol
        h, hb = m.forward_3d_bary(z, pos, batch)
        assert rel(h.numpy(), g[tag + "_h"]) < tol
        assert rel(hb.numpy(), g[tag + "_h_bary_nodes"]) < tol
    h3d, hbary = m.forward_w_barycenter(z, pos, K, batch)
    assert rel(h3d.detach().numpy(), g[tag + "_h_3d"]) < tol
    # FGW readout: fp64 must agree tightly; fp32 within 1e-4 (north_star tolerance)
    assert rel(hbary.detach().numpy(), g[tag + "_h_bary"]) < (2e-7 if tag == "r64" else 1e-4)   # F_bary_batch is a float32 buffer even in the fp64 run (schnet_no_sum.py:254-256)
    if (tag + "_grad:lin1_bary.weight") in g.files:
        gw1 = torch.from_numpy(g["gw_h3d"]).to(dtype); gw2 = torch.from_numpy(g["gw_hbary"]).to(dtype)
        ((h3d * gw1).sum() + (hbary * gw2).sum()).backward()
        for name, p in m.named_parameters():
            ref = g[f"{tag}_grad:{name}"]
            assert rel(p.grad.numpy(), ref) < (1e-7 if tag == "r64" else 2e-3), name
