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


def _radius_scan(pos, lo, hi, r, cap, loop):
    """torch-cluster 1.6.1 written out as its CUDA kernel's loops (radius_cuda.cu: one query, linear scan over the example's
    points in index order, stop after max_num_neighbors hits) + radius_graph's post-filter of the self pairs."""
    limit = cap if loop else cap + 1
    rows, cols = [], []
    r2 = np.float32(r) * np.float32(r)
    for i in range(lo, hi):
        cnt = 0
        for j in range(lo, hi):
            d = pos[j] - pos[i]
            sq = d * d
            if np.float32(np.float32(sq[0] + sq[1]) + sq[2]) < r2:
                if loop or j != i:
                    rows.append(j); cols.append(i)
                cnt += 1
            if cnt >= limit:
                break
    return np.array([rows, cols], dtype=np.int64)


@pytest.mark.parametrize("n,r,cap,loop", [(40, 10.0, 32, False), (40, 10.0, 32, True), (47, 3.0, 8, False), (20, 10.0, 32, False)])
def test_radius_graph_truncation_rule(n, r, cap, loop):
    """radius(x, x, r, ..., cap+1) INCLUDING self, then self pairs dropped: a target with >= cap+1 lower-index
    in-range atoms keeps cap+1 edges."""
    from oracle import pyg_semantics as ps
    rng = np.random.RandomState(n)
    pos = rng.uniform(0, 4.0, size=(2 * n, 3)).astype(np.float32)
    batch = torch.arange(2).repeat_interleave(n)
    ei = ps.radius_graph(torch.from_numpy(pos), r, batch, loop=loop, max_num_neighbors=cap).numpy()
    ref = np.concatenate([_radius_scan(pos, 0, n, r, cap, loop), _radius_scan(pos, n, 2 * n, r, cap, loop)], axis=1)
    assert np.array_equal(ei, ref)
    if n == 40 and not loop:
        deg = np.bincount(ei[1], minlength=2 * n).reshape(2, n)
        assert np.all(deg[:, :33] == 32) and np.all(deg[:, 33:] == 33)
