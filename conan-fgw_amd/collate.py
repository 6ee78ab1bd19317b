"""Host-side batch assembly for the MI355X hot path.

Counterpart of the reference's `LargeConformerBasedDataset.collate_fn` (conan_fgw/src/data/datasets.py:170-199) and
`EquivAggregation.create_aggregation_index` (conan_fgw/src/model/common.py:414-423): a list of dataset items — one molecule
each, K conformers that share `z`, the 2-D bond graph and its features and differ only in `pos` (datasets.py:133-148,
conformers/features.py:196-205) — becomes the flat molecule-major batch the model API consumes, `(data_batch,
batch_node_index)`, plus `conformers_index`.

Division of labour (include/conan_fgw_hip.h, "batch assembly"):
  host    conan_collate_layout / conan_collate_pack : every molecule packed ONCE into one pinned buffer (plain memcpy in C,
          no per-atom Python), ~1.6 KB per ESOL-sized molecule;
  copy    ONE asynchronous H2D transfer on a copy stream (double-buffered: batch i+1 travels while batch i computes);
  device  conan_collate_unpack : one kernel expands the packed bytes into z / pos / batch / x / edge_index (node offsets
          applied) / edge_attr / y / graph_ptr / conformers_index.
`num_graphs` and `max_nodes` are known on the host from the item sizes, so the model runs without the two host
synchronisations the reference performs per step (`len(batch.unique())`, schnet_no_sum.py:345; `to_dense_batch`, :242).
"""
from __future__ import annotations

import ctypes
import dataclasses
import types
from typing import List, Optional, Sequence

import numpy as np
import torch

from ._lib import BatchLayout, call, lib, ptr, stream_ptr


@dataclasses.dataclass
class ConformerMolecule:
    """One dataset item: what `LargeConformerBasedDataset.get` returns for a molecule (datasets.py:133-148) without the PyG
    containers — K conformer `Data` objects with identical z / x / edge_index / edge_attr / y and their own pos."""
    z: np.ndarray            # [n] int64 atomic numbers
    pos: np.ndarray          # [K, n, 3] float32
    x: np.ndarray            # [n, x_dim] float32 atom features (PyG from_smiles: 9 integer-valued columns)
    edge_index: np.ndarray   # [2, e] int64 directed covalent bonds, molecule-local atom indices
    edge_attr: np.ndarray    # [e, ea_dim] float32
    y: float
    smiles: str = ""

    @staticmethod
    def from_reference_item(item) -> "ConformerMolecule":
        """Adapter for the reference's own item `(Batch of K conformer Data, [n_atoms] * K)`: anything whose first element has
        `to_data_list()` yielding objects with z / pos / x / edge_index / edge_attr / y (PyG Data)."""
        data, _n_atoms = item
        confs = data.to_data_list()
        d0 = confs[0]
        npy = lambda t: t.detach().cpu().numpy() if torch.is_tensor(t) else np.asarray(t)
        return ConformerMolecule(
            z=npy(d0.z).astype(np.int64), pos=np.stack([npy(c.pos) for c in confs]).astype(np.float32),
            x=npy(d0.x).astype(np.float32), edge_index=npy(d0.edge_index).astype(np.int64),
            edge_attr=npy(d0.edge_attr).astype(np.float32).reshape(npy(d0.edge_index).shape[1], -1),
            y=float(np.asarray(npy(d0.y)).reshape(-1)[0]), smiles=getattr(d0, "smiles", ""))


class DeviceBatch(types.SimpleNamespace):
    """The collated batch on the device: `z, pos, batch, x, edge_index, edge_attr, y, smiles` (the PyG `Batch` fields the models
    read, schnet_based_models.py:135-173) + `conf_node_batch` (datasets.py:197) + `batch_node_index` (== `batch`), `conformers_index`, `graph_ptr` and the host-known
    `num_graphs`, `max_nodes`, `num_molecules`.  `ready` is recorded on the copy stream behind the unpack kernel."""

    def wait(self, stream: Optional[torch.cuda.Stream] = None):
        (stream or torch.cuda.current_stream()).wait_event(self.ready)
        return self

    def as_model_input(self):
        """`(data_batch, batch_node_index)`: exactly what the reference's collate_fn returns (datasets.py:199)."""
        return self, self.batch_node_index


def _as_items(batch_items: Sequence) -> List[ConformerMolecule]:
    return [it if isinstance(it, ConformerMolecule) else ConformerMolecule.from_reference_item(it) for it in batch_items]


class DeviceCollator:
    """collate_fn for the HIP path.  Holds `depth` pinned host buffers and device staging buffers (grown on demand) and a copy
    stream; `__call__(items)` packs on the host, enqueues the H2D copy and the expansion kernel on the copy stream and returns
    immediately.  With `static=True` the expanded tensors are allocated once (worst case over the batches seen so far must not
    grow) and re-used, which is what a captured HIP graph needs: fixed input addresses."""

    def __init__(self, device, num_conformers: int, depth: int = 2, static: bool = False):
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("DeviceCollator assembles batches for the GPU path; there is no CPU fallback")
        self.K = int(num_conformers)
        self.depth = depth
        self.static = static
        self._pinned = [None] * depth
        self._staged = [None] * depth
        self._events = [None] * depth
        self._slot = 0
        self._out = None
        self.copy_stream = torch.cuda.Stream(device=self.device)
        self.last_packed_bytes = 0

    # ---------------------------------------------------------------------------------------------- host half
    def pack(self, batch_items: Sequence):
        """Host half only: returns (layout, pinned uint8 tensor holding the packed batch, smiles)."""
        items = _as_items(batch_items)
        B, K = len(items), self.K
        n_atoms = np.fromiter((len(it.z) for it in items), dtype=np.int32, count=B)
        n_bonds = np.fromiter((it.edge_index.shape[1] for it in items), dtype=np.int32, count=B)
        x_dim = items[0].x.shape[1] if items[0].x.ndim == 2 else 0
        ea_dim = items[0].edge_attr.shape[1] if items[0].edge_attr.ndim == 2 else 0
        keep = []                                            # keeps converted arrays alive until the C call returns

        def arr(a, dtype, shape=None):
            a = np.ascontiguousarray(a, dtype=dtype)
            if shape is not None and tuple(a.shape) != tuple(shape):
                raise ValueError(f"collate: expected shape {tuple(shape)}, got {tuple(a.shape)}")
            keep.append(a)
            return a.ctypes.data

        PP = ctypes.c_void_p * B
        zs = PP(*[arr(it.z, np.int64, (n_atoms[m],)) for m, it in enumerate(items)])
        ps = PP(*[arr(it.pos, np.float32, (K, n_atoms[m], 3)) for m, it in enumerate(items)])
        xs = PP(*[arr(it.x, np.float32, (n_atoms[m], x_dim)) for m, it in enumerate(items)])
        es = PP(*[arr(it.edge_index, np.int64, (2, n_bonds[m])) for m, it in enumerate(items)])
        eas = PP(*[arr(it.edge_attr, np.float32, (n_bonds[m], ea_dim)) for m, it in enumerate(items)])
        ys = np.fromiter((it.y for it in items), dtype=np.float32, count=B)
        L = BatchLayout()
        rc = lib().conan_collate_layout(B, K, n_atoms.ctypes.data, n_bonds.ctypes.data, x_dim, ea_dim, ctypes.byref(L))
        if rc != 0:
            raise RuntimeError(f"conan_collate_layout failed ({rc})")
        slot = self._slot
        if self._events[slot] is not None:
            self._events[slot].synchronize()                 # the copy that last used this pinned buffer has finished
        if self._pinned[slot] is None or self._pinned[slot].numel() < L.bytes:
            self._pinned[slot] = torch.empty(int(L.bytes * 1.25) + 256, dtype=torch.uint8, pin_memory=True)
        pinned = self._pinned[slot]
        rc = lib().conan_collate_pack(ctypes.byref(L), n_atoms.ctypes.data, n_bonds.ctypes.data, zs, ps, xs, es, eas, ys.ctypes.data,
                                      pinned.data_ptr())
        if rc != 0:
            raise ValueError("collate: bad item (a bond leaves its molecule, or an array is missing)")
        self.last_packed_bytes = int(L.bytes)
        return L, pinned, [it.smiles for it in items]

    # ---------------------------------------------------------------------------------------------- device half
    def _outputs(self, L: BatchLayout):
        dev = self.device
        A, E, G = L.num_atoms, L.num_bond_edges, L.num_graphs
        if self.static and self._out is not None:
            o = self._out
            if (o["z"].shape[0], o["edge_index"].shape[1], o["y"].shape[0]) != (A, E, G):
                raise RuntimeError("static DeviceCollator: the batch shape changed (atoms / bond edges / graphs); "
                                   "captured-graph replay needs shape-stable batches")
            return o
        o = dict(z=torch.empty(A, dtype=torch.int64, device=dev), pos=torch.empty(A, 3, dtype=torch.float32, device=dev),
                 batch=torch.empty(A, dtype=torch.int64, device=dev), x=torch.empty(A, L.x_dim, dtype=torch.float32, device=dev),
                 edge_index=torch.empty(2, E, dtype=torch.int64, device=dev), edge_attr=torch.empty(E, L.ea_dim, dtype=torch.float32, device=dev),
                 y=torch.empty(G, dtype=torch.float32, device=dev), graph_ptr=torch.empty(G + 1, dtype=torch.int32, device=dev),
                 conformers_index=torch.empty(G, dtype=torch.int64, device=dev), conf_node_batch=torch.empty(A, dtype=torch.int64, device=dev))
        if self.static:
            self._out = o
        return o

    def __call__(self, batch_items: Sequence) -> DeviceBatch:
        L, pinned, smiles = self.pack(batch_items)
        slot = self._slot
        self._slot = (slot + 1) % self.depth
        cs = self.copy_stream
        main = torch.cuda.current_stream(self.device)
        if self._staged[slot] is None or self._staged[slot].numel() < L.bytes:
            with torch.cuda.stream(cs):                      # allocated on the stream that writes it: the caching allocator orders a
                self._staged[slot] = torch.empty(pinned.numel(), dtype=torch.uint8, device=self.device)      # reused block behind its last use there
        staged = self._staged[slot]
        with torch.cuda.stream(cs):
            staged[: L.bytes].copy_(pinned[: L.bytes], non_blocking=True)      # overlaps whatever the caller's stream is running
        if self.static and self._out is not None:
            cs.wait_stream(main)                             # the shared output tensors may still be read by work already queued
        with torch.cuda.stream(cs):
            o = self._outputs(L)
            call("conan_collate_unpack", ptr(staged), ctypes.byref(L), ptr(o["z"]), ptr(o["pos"]), ptr(o["batch"]), ptr(o["x"]) if L.x_dim else None,
                 ptr(o["edge_index"]) if L.num_bond_edges else None, ptr(o["edge_attr"]) if (L.num_bond_edges and L.ea_dim) else None,
                 ptr(o["y"]), ptr(o["graph_ptr"]), ptr(o["conformers_index"]), ptr(o["conf_node_batch"]), stream_ptr())
            ev = torch.cuda.Event()
            ev.record(cs)
        self._events[slot] = ev
        for t in o.values():
            t.record_stream(main)                            # allocated on the copy stream, consumed on the caller's
        staged.record_stream(cs)
        return DeviceBatch(**o, batch_node_index=o["batch"], smiles=[s for s in smiles for _ in range(L.K)], num_graphs=int(L.num_graphs), max_nodes=int(L.max_nodes),
                           num_molecules=int(L.B), num_conformers=int(L.K), ready=ev)


def collate_fn(batch_items: Sequence, device=None, num_conformers: Optional[int] = None):
    """Functional form with the reference's return value `(data_batch, batch_node_index)` (datasets.py:199); the batch is ready
    on the current stream when it returns (no host synchronisation).  For a training loop build ONE `DeviceCollator` and call it
    per batch instead: it re-uses its pinned buffers and overlaps the copy with the previous step."""
    items = _as_items(batch_items)
    K = int(num_conformers) if num_conformers is not None else int(items[0].pos.shape[0])
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    b = DeviceCollator(dev, K, depth=1)(items).wait()
    return b.as_model_input()


def molecules_from_synthetic(cb, bg) -> List[ConformerMolecule]:
    """Dataset items out of a synthetic flat batch (`synthetic.make_batch` + `make_bond_graph`): test / benchmark helper."""
    K = cb.num_conformers
    gp = cb.graph_ptr
    items = []
    e_src, e_dst = bg.edge_index
    for m in range(cb.num_molecules):
        lo, hi = int(gp[m * K]), int(gp[m * K + 1])
        n = hi - lo
        pos = np.stack([cb.pos[int(gp[m * K + k]): int(gp[m * K + k + 1])] for k in range(K)])
        sel = (e_dst >= lo) & (e_dst < hi)                  # the first conformer's copy of the bond graph, in batch order
        items.append(ConformerMolecule(z=cb.z[lo:hi].copy(), pos=pos, x=bg.x[lo:hi].copy(), edge_index=np.stack([e_src[sel] - lo, e_dst[sel] - lo]),
                                       edge_attr=bg.edge_attr[sel].copy(), y=float(cb.y[m]), smiles=f"mol{m}"))
    return items
