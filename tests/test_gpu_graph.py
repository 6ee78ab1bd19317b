"""GPU parity: graph construction kernels vs the oracle (integer outputs must be bit-exact)."""
import numpy as np
import pytest
import torch

from helpers import golden_files, rel
from conan_fgw_amd import ops
from conan_fgw_amd.synthetic import make_batch
from oracle import pyg_semantics as ps

pytestmark = pytest.mark.gpu
dev = torch.device("cuda:0")


def _graph(b, cutoff, cap, loop=False):
    pos = torch.from_numpy(b.pos).to(dev)
    batch = torch.from_numpy(b.batch).to(dev)
    gp = ops.graph_ptr_from_batch(batch, b.num_graphs)
    return gp, ops.RadiusGraph(pos, gp, b.num_graphs, cutoff, cap, loop)


@pytest.mark.parametrize("shape,B,K,box,cutoff,cap,loop", [
    ("esol", 16, 5, None, 10.0, 32, False),        # complete graphs, production SchNet
    ("esol", 16, 5, 16.0, 10.0, 32, False),        # stretched: cutoff actually prunes
    ("lipo", 6, 5, None, 10.0, 32, False),         # n > 33: truncation active (first cap+1 candidates incl. self, then drop self)
    ("bace", 4, 5, None, 5.0, 32, False),          # ViSNet FGW adjacency (visnet.py:90)
    ("bace", 4, 5, None, 5.0, 32, True),           # ViSNet Distance(): loop=True, cap includes self
    ("freesolv", 8, 20, 12.0, 3.0, 4, False),      # tiny cap
])
def test_radius_graph_bit_exact(shape, B, K, box, cutoff, cap, loop):
    b = make_batch(shape, B, K, seed=11, box=box)
    gp, g = _graph(b, cutoff, cap, loop)
    assert np.array_equal(gp.cpu().numpy(), b.graph_ptr.astype(np.int32))
    ref = ps.radius_graph(torch.from_numpy(b.pos), cutoff, torch.from_numpy(b.batch), loop=loop, max_num_neighbors=cap)
    ei = g.edge_index().cpu()
    assert ei.dtype == torch.int64
    assert torch.equal(ei, ref), "neighbour lists differ from the oracle"
    row, col = ref
    ew = (torch.from_numpy(b.pos)[row] - torch.from_numpy(b.pos)[col]).norm(dim=-1).numpy()
    assert np.abs(g.edge_weight().cpu().numpy() - ew).max() <= 1e-6 * max(1.0, ew.max())
    # CSR invariants + transpose
    rowptr = g.rowptr.cpu().numpy()
    assert rowptr[0] == 0 and rowptr[-1] == ref.shape[1] and np.all(np.diff(rowptr) <= (cap if loop else cap + 1))
    t_rowptr, t_eid = g.transpose()
    t_rowptr, t_eid = t_rowptr.cpu().numpy(), t_eid.cpu().numpy()[: ref.shape[1]]
    assert sorted(t_eid.tolist()) == list(range(ref.shape[1]))
    src = ref[0].numpy()
    for j in np.random.RandomState(0).choice(len(b.z), size=min(50, len(b.z)), replace=False):
        ids = t_eid[t_rowptr[j]:t_rowptr[j + 1]]
        assert np.all(src[ids] == j) and np.all(np.diff(ids) > 0)
        assert len(ids) == int((src == j).sum())


def test_golden_edge_index():
    for path in golden_files("schnet_ref_"):
        gd = np.load(path)
        pos = torch.from_numpy(gd["pos"]).to(dev); batch = torch.from_numpy(gd["batch"]).to(dev)
        G = int(gd["batch"].max()) + 1
        g = ops.RadiusGraph(pos, ops.graph_ptr_from_batch(batch, G), G, 10.0, 32)
        assert np.array_equal(g.edge_index().cpu().numpy(), gd["edge_index"])
        assert rel(g.edge_weight().cpu().numpy(), gd["r64_edge_weight"]) < 1e-6


def test_truncation_window_includes_self():
    """torch-cluster 1.6.1: radius_graph(loop=False) = radius(..., cap + 1) in ascending source order INCLUDING the target, then
    the self pair is removed.  40 atoms, all within the cutoff: targets with local index < 33 find themselves inside the
    33-candidate window and keep 32 edges (sources 0..32 without self); targets >= 33 never reach themselves and keep 33
    (sources 0..32).  With loop=True the window is cap = 32 candidates and nothing is removed."""
    rng = np.random.RandomState(5)
    n, G = 40, 3
    pos = torch.from_numpy(rng.uniform(0, 4.0, size=(n * G, 3)).astype(np.float32)).to(dev)
    batch = torch.arange(G, device=dev).repeat_interleave(n)
    gp = ops.graph_ptr_from_batch(batch, G)
    g = ops.RadiusGraph(pos, gp, G, 10.0, 32)
    deg = np.diff(g.rowptr.cpu().numpy()).reshape(G, n)
    assert np.all(deg[:, :33] == 32) and np.all(deg[:, 33:] == 33)
    ei = g.edge_index().cpu()
    assert torch.equal(ei, ps.radius_graph(pos.cpu(), 10.0, batch.cpu(), max_num_neighbors=32))
    src = ei[0].numpy().reshape(-1) - (ei[1].numpy() // n) * n
    assert src.max() == 32                                             # no source beyond the window
    gl = ops.RadiusGraph(pos, gp, G, 10.0, 32, loop=True)
    assert np.all(np.diff(gl.rowptr.cpu().numpy()) == 32)
    assert torch.equal(gl.edge_index().cpu(), ps.radius_graph(pos.cpu(), 10.0, batch.cpu(), loop=True, max_num_neighbors=32))
    # every downstream consumer handles 33-edge rows: transpose is a permutation, pairs cover every edge
    E = g.num_edges
    t_rowptr, t_eid = g.transpose()
    assert sorted(t_eid[:E].cpu().tolist()) == list(range(E))
    pid = g.pairs().pid[:E].cpu().numpy()
    assert pid.min() == 0 and pid.max() == int(g.num_pairs_dev.item()) - 1


def test_empty_and_single_atom_graphs():
    pos = torch.tensor([[0., 0, 0], [1, 0, 0], [0, 1, 0], [5, 5, 5]], device=dev)
    batch = torch.tensor([0, 0, 0, 3], device=dev)                     # graphs 1 and 2 are empty, graph 3 has one atom
    gp = ops.graph_ptr_from_batch(batch, 5)
    assert gp.cpu().tolist() == [0, 3, 3, 3, 4, 4]
    g = ops.RadiusGraph(pos, gp, 5, 10.0, 32)
    assert g.num_edges == 6 and g.rowptr.cpu().tolist() == [0, 2, 4, 6, 6]


def test_full_size_properties():
    """cfg2 size: sortedness, symmetry of the un-truncated graph, degree bound — size-independent checks."""
    b = make_batch("esol", 256, 5, seed=1236)
    gp, g = _graph(b, 10.0, 32)
    E = g.num_edges
    col, tgt = g.col[:E].cpu().numpy(), g.tgt[:E].cpu().numpy()
    assert np.all(np.diff(tgt) >= 0)
    key = tgt.astype(np.int64) * len(b.z) + col
    assert np.all(np.diff(key) > 0)                                    # sorted by (target, source), no duplicates
    assert np.array_equal(b.batch[col], b.batch[tgt]) and np.all(col != tgt)
    rev = set(zip(col.tolist(), tgt.tolist()))
    assert all((t, c) in rev for c, t in list(rev)[:5000])             # n <= 33 => no truncation => symmetric
