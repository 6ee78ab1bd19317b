"""Host half of the batch assembly (conan_collate_layout / conan_collate_pack: plain C on host memory, no GPU involved) against
the oracle's restatement of the reference's collate_fn (oracle/collate.py)."""
import ctypes

import numpy as np
import pytest

from conan_fgw_amd._lib import BatchLayout, lib
from conan_fgw_amd.collate import ConformerMolecule, molecules_from_synthetic
from conan_fgw_amd.synthetic import make_batch, make_bond_graph
from oracle import collate as ocoll


def _pack(items, K):
    B = len(items)
    n_atoms = np.array([len(it.z) for it in items], np.int32); n_bonds = np.array([it.edge_index.shape[1] for it in items], np.int32)
    L = BatchLayout()
    assert lib().conan_collate_layout(B, K, n_atoms.ctypes.data, n_bonds.ctypes.data, 9, 3, ctypes.byref(L)) == 0
    keep = []
    def P(arrs):
        keep.extend(arrs)
        return (ctypes.c_void_p * B)(*[a.ctypes.data for a in arrs])
    buf = np.zeros(L.bytes, np.uint8)
    ys = np.array([it.y for it in items], np.float32)
    rc = lib().conan_collate_pack(ctypes.byref(L), n_atoms.ctypes.data, n_bonds.ctypes.data,
                                  P([np.ascontiguousarray(it.z, np.int64) for it in items]), P([np.ascontiguousarray(it.pos, np.float32) for it in items]),
                                  P([np.ascontiguousarray(it.x, np.float32) for it in items]), P([np.ascontiguousarray(it.edge_index, np.int64) for it in items]),
                                  P([np.ascontiguousarray(it.edge_attr, np.float32) for it in items]), ys.ctypes.data, buf.ctypes.data)
    return rc, L, buf


def test_pack_layout_and_contents():
    K = 3
    cb = make_batch("esol", 5, K, seed=3); bg = make_bond_graph(cb, seed=4)
    items = molecules_from_synthetic(cb, bg)
    rc, L, buf = _pack(items, K)
    assert rc == 0
    ref = ocoll.collate(items, K)
    assert (L.B, L.K, L.num_graphs) == (5, K, 15)
    assert L.num_atoms == len(ref["z"]) == len(cb.z) and L.num_bond_edges == ref["edge_index"].shape[1] == bg.edge_index.shape[1]
    assert L.max_nodes == cb.max_nodes
    for off in (L.off_z, L.off_pos, L.off_x, L.off_bsrc, L.off_bdst, L.off_battr, L.off_y):
        assert off % 16 == 0 and off < L.bytes
    aoff = buf[L.off_atom_off: L.off_atom_off + 4 * 6].view(np.int32)
    assert aoff.tolist() == np.concatenate([[0], np.cumsum([len(it.z) for it in items])]).tolist()
    zs = buf[L.off_z: L.off_z + 4 * aoff[-1]].view(np.int32)
    assert np.array_equal(zs, np.concatenate([it.z for it in items]))
    pos = buf[L.off_pos: L.off_pos + 12 * K * aoff[-1]].view(np.float32)
    assert np.array_equal(pos, np.concatenate([it.pos.reshape(-1) for it in items]))
    # the synthetic flat batch and the oracle's collate of its items agree (edge order inside a graph is the item's own)
    assert np.array_equal(ref["z"], cb.z) and np.array_equal(ref["pos"], cb.pos) and np.array_equal(ref["batch"], cb.batch)
    assert np.array_equal(ocoll.aggregation_index(ref["smiles"], K), np.repeat(np.arange(5), K))
    # conf_node_batch written out as the reference's torch expression (datasets.py:177,189-192,197)
    import torch
    cnb, node_count = [], 0
    for it in items:
        cnb.extend((torch.arange(len(it.z)) + node_count).repeat(K))
        node_count += len(it.z)
    assert np.array_equal(ref["conf_node_batch"], torch.LongTensor(cnb).numpy())


def test_pack_rejects_a_bond_that_leaves_its_molecule():
    cb = make_batch("esol", 2, 2, seed=5); bg = make_bond_graph(cb, seed=6)
    items = molecules_from_synthetic(cb, bg)
    items[1].edge_index = items[1].edge_index.copy(); items[1].edge_index[0, 0] = len(items[1].z)      # out of range
    rc, _, _ = _pack(items, 2)
    assert rc == -1


def test_layout_rejects_bad_sizes():
    L = BatchLayout()
    n = np.array([3, -1], np.int32); e = np.array([2, 2], np.int32)
    assert lib().conan_collate_layout(2, 2, n.ctypes.data, e.ctypes.data, 9, 3, ctypes.byref(L)) == -1
    assert lib().conan_collate_layout(0, 2, n.ctypes.data, e.ctypes.data, 9, 3, ctypes.byref(L)) == -1
