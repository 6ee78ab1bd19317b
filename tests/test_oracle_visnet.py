"""The oracle's ViSNet restatement (oracle/visnet.py) against outputs of the reference's own ViSNet classes
(torch_geometric_visnet.py + visnet.py, run over the PyG stand-in: tests/golden/visnet_ref_*.npz)."""
import os

import numpy as np
import pytest
import torch

from helpers import golden_files, rel
from oracle.visnet import ViSNetOracle

CASES = golden_files("visnet_ref_")


def load_oracle(g, dtype):
    m = ViSNetOracle(hidden_channels=int(g["hidden"]))
    sd = {k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sd:")}
    m.load_state_dict(sd, strict=True)                       # same module / parameter / buffer names as the reference
    return m.to(dtype)


@pytest.mark.parametrize("path", CASES, ids=[os.path.basename(p)[11:-4] for p in CASES])
@pytest.mark.parametrize("tag,dtype,tol", [("r64", torch.float64, 1e-9), ("r32", torch.float32, 5e-5)])
def test_oracle_matches_reference_visnet(path, tag, dtype, tol):
    g = np.load(path)
    m = load_oracle(g, dtype)
    z, pos, batch = torch.from_numpy(g["z"]), torch.from_numpy(g["pos"]).to(dtype), torch.from_numpy(g["batch"])
    K = int(g["K"])
    with torch.no_grad():
        xs, vs = m.representation_model(z, pos, batch)
        assert rel(xs.numpy(), g[tag + "_x"]) < tol and rel(vs.numpy(), g[tag + "_vec"]) < tol
        assert rel(m(z, pos, batch).numpy(), g[tag + "_forward"]) < tol
        h, hb = m.forward_3d_bary(z, pos, batch)
        assert rel(h.numpy(), g[tag + "_h"]) < tol and rel(hb.numpy(), g[tag + "_h_bary_nodes"]) < tol
        ei, _ = m.interaction_graph(pos, batch)
        assert np.array_equal(ei.numpy(), g["edge_index"])
        h3d, hbary = m.forward_w_barycenter(z, pos, K, batch)
    assert rel(h3d.numpy(), g[tag + "_h_3d"]) < tol
    assert rel(hbary.numpy(), g[tag + "_h_bary"]) < (2e-7 if tag == "r64" else 2e-4)
