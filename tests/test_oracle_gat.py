"""Oracle of the covalent branch (oracle/gat.py) and host-side checks of the drop-in module that need no GPU."""
import numpy as np
import torch

from conan_fgw_amd.synthetic import make_batch, make_bond_graph
from oracle.gat import GATBasedOracle, GATConvOracle


def test_oracle_gat_softmax_rows_and_self_loops():
    """Attention coefficients of a target sum to 1 (self loop included); an isolated node reproduces lin(x) + bias."""
    torch.manual_seed(0)
    conv = GATConvOracle(9, 16).double()
    x = torch.randn(5, 9, dtype=torch.float64)
    ei = torch.tensor([[0, 1, 1, 2], [1, 0, 2, 1]])
    ea = torch.randn(4, 3, dtype=torch.float64)
    out = conv(x, ei, ea)
    h = conv.lin_src(x)
    assert torch.allclose(out[3], h[3] + conv.bias) and torch.allclose(out[4], h[4] + conv.bias)     # nodes 3, 4 have no bonds
    # node 0 has one neighbour: out_0 is a convex combination of h_0 and h_1
    w = torch.linalg.lstsq(torch.stack([h[0], h[1]], 1), (out[0] - conv.bias)[:, None]).solution.flatten()
    w = w.detach()
    assert abs(float(w.sum()) - 1.0) < 1e-9 and (w > 0).all()


def test_oracle_gat_edge_order_invariance_and_graph_sum():
    b = make_batch("esol", 3, 2, seed=4)
    g = make_bond_graph(b, seed=5)
    torch.manual_seed(1)
    m = GATBasedOracle().double()
    x, ei, ea, bt = (torch.from_numpy(a) for a in (g.x, g.edge_index, g.edge_attr, b.batch))
    o1 = m(x, ei, ea, bt)
    perm = torch.randperm(ei.shape[1], generator=torch.Generator().manual_seed(2))
    o2 = m(x, ei[:, perm], ea[perm], bt)
    assert o1.shape == (b.num_graphs, 64) and torch.allclose(o1, o2, atol=1e-12)
    # the K conformers of a molecule share the 2-D graph => identical covalent embeddings
    o = o1.view(b.num_molecules, b.num_conformers, -1)
    assert torch.allclose(o[:, 0], o[:, 1], atol=1e-12)


def test_drop_in_gat_has_pyg_parameter_names_and_no_cpu_path():
    from conan_fgw_amd.gat import GATBased
    m = GATBased(out_channels=64, edge_dim=3)
    keys = set(m.state_dict().keys())
    assert keys == set(GATBasedOracle().state_dict().keys())
    assert m.gat_conv1.lin_src.weight.shape == (64, 9) and m.gat_conv2.lin_src.weight.shape == (64, 64)
    assert m.gat_conv1.lin_dst is m.gat_conv1.lin_src and m.gat_conv1.att_src.shape == (1, 1, 64)
    b = make_batch("esol", 2, 2, seed=1)
    g = make_bond_graph(b, seed=1)
    try:
        m(torch.from_numpy(g.x), torch.from_numpy(g.edge_index), torch.from_numpy(g.edge_attr), torch.from_numpy(b.batch))
    except RuntimeError as e:
        assert "no CPU path" in str(e)
    else:
        raise AssertionError("CPU tensors must be rejected")
