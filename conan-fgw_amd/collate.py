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
        stream = stream or torch.cuda.current_stream()
        stream.wait_event(self.ready)
        pair = getattr(self, "_static_pair", None)
        if pair is not None:
            # static collator: the expansion kernel wrote a landing copy; ONE device-to-device copy moves it to the fixed addresses the
            # model (a captured HIP graph) reads.  The next batch may be expanded as soon as this copy is done — not only when the step
            # that reads these tensors is — so expansion and step overlap.
            collator, fixed, landing = pair
            with torch.cuda.stream(stream):
                fixed.copy_(landing, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(stream)
            collator._consumed = ev
            collator._landing_busy = False
            self._static_pair = None
            # The fixed views are ONE set of tensor objects for every batch: the host-known sizes are attached here, on the consumer's thread,
            # at the moment this batch becomes their content — not at enqueue time, where a worker thread assembling batch i + 1 would
            # overwrite what a size-less call on batch i is about to read (a smaller max_nodes under-sizes the dense barycenter padding).
            self.batch._conan_hints = (int(self.num_graphs), int(self.max_nodes))
            if collator._landing_released is not None:
                collator._landing_released.release()          # CollatePipeline's worker may assemble the next batch
        return self

    def as_model_input(self):
        """`(data_batch, batch_node_index)`: exactly what the reference's collate_fn returns (datasets.py:199)."""
        return self, self.batch_node_index


def _item_record(it: ConformerMolecule, K: int, x_dim: int, ea_dim: int):
    """(n_atoms, n_bonds, five array addresses, the arrays themselves) of one dataset item, validated and made contiguous ONCE and kept on
    the item: a dataset hands the same objects out every epoch, and five `np.ascontiguousarray` calls per molecule per batch were the
    whole cost of the host half (1.3 of 1.4 ms per 256-molecule batch, round 4)."""
    rec = getattr(it, "_conan_record", None)
    if rec is not None and rec[0] == (K, x_dim, ea_dim, id(it.z), id(it.pos), id(it.x), id(it.edge_index), id(it.edge_attr)):
        return rec[1]
    n, nb = len(it.z), it.edge_index.shape[1]

    def arr(a, dtype, shape):
        a = np.ascontiguousarray(a, dtype=dtype)
        if tuple(a.shape) != tuple(shape):
            raise ValueError(f"collate: expected shape {tuple(shape)}, got {tuple(a.shape)}")
        return a
    arrays = (arr(it.z, np.int64, (n,)), arr(it.pos, np.float32, (K, n, 3)), arr(it.x, np.float32, (n, x_dim)),
              arr(it.edge_index, np.int64, (2, nb)), arr(it.edge_attr, np.float32, (nb, ea_dim)))
    out = (n, nb, tuple(a.ctypes.data for a in arrays), arrays)
    # Cached only when every array IS the item's own (right dtype, contiguous: no copy was made): a converted copy would keep serving the old
    # values after an in-place change of the source, and `id()` of a replaced array can be reused.  Items that need a conversion pay it per batch.
    if all(a is src for a, src in zip(arrays, (it.z, it.pos, it.x, it.edge_index, it.edge_attr))):
        it._conan_record = ((K, x_dim, ea_dim, id(it.z), id(it.pos), id(it.x), id(it.edge_index), id(it.edge_attr)), out)
    return out


def _as_items(batch_items: Sequence) -> List[ConformerMolecule]:
    return [it if isinstance(it, ConformerMolecule) else ConformerMolecule.from_reference_item(it) for it in batch_items]


class DeviceCollator:
    """collate_fn for the HIP path.  Holds `depth` pinned host buffers and device staging buffers (grown on demand) and a copy
    stream; `__call__(items)` packs on the host, enqueues the H2D copy and the expansion kernel on the copy stream and returns
    immediately.  With `static=True` the expanded tensors are allocated once (the batch shape must not change) and re-used, which is
    what a captured HIP graph needs: fixed input addresses.  The kernel then expands into a landing copy and `DeviceBatch.wait()` moves
    it to the fixed addresses with one device-to-device transfer, so the expansion of batch i+1 overlaps the step on batch i."""

    def __init__(self, device, num_conformers: int, depth: int = 2, static: bool = False, strict_max_nodes: bool = True):
        """`strict_max_nodes` (static collators): reject a batch whose largest conformer differs from the first batch's — the padded size N of
        the dense FGW problem is a launch parameter of the step, so a captured graph replayed on such a batch would run with the wrong N
        (and N is part of the result: SURVEY.md Appendix D.1).  False: only the addresses are fixed (eager consumers); the sizes of every
        batch ride on its index tensor from `wait()` on."""
        self.device = torch.device(device)
        self.strict_max_nodes = bool(strict_max_nodes)
        self._static_max_nodes = None
        if self.device.type != "cuda":
            raise RuntimeError("DeviceCollator assembles batches for the GPU path; there is no CPU fallback")
        self.K = int(num_conformers)
        self.depth = depth
        self.static = static
        self._pinned = [None] * depth
        self._staged = [None] * depth
        self._events = [None] * depth
        self._pack_slot = 0
        self._landing_released = None                        # set by CollatePipeline (static collators): a semaphore the consumer's wait() releases
        self._out = None
        self._consumed = None                                # static mode: event behind the last landing -> fixed copy
        self._landing_busy = False
        self.copy_stream = torch.cuda.Stream(device=self.device)
        self.last_packed_bytes = 0

    # ---------------------------------------------------------------------------------------------- host half
    def pack(self, batch_items: Sequence):
        """Host half only: returns (layout, pinned uint8 tensor holding the packed batch, smiles, ring slot).  Safe to call from a worker
        thread while the calling thread runs `enqueue` for earlier batches (CollatePipeline): the slots are handed out round-robin and a slot
        is reused only after the copy that last read it has finished."""
        items = _as_items(batch_items)
        B, K = len(items), self.K
        x_dim = items[0].x.shape[1] if items[0].x.ndim == 2 else 0
        ea_dim = items[0].edge_attr.shape[1] if items[0].edge_attr.ndim == 2 else 0
        recs = [_item_record(it, K, x_dim, ea_dim) for it in items]      # (the arrays stay alive on the items until the C call returns)
        n_atoms = np.fromiter((r[0] for r in recs), dtype=np.int32, count=B)
        n_bonds = np.fromiter((r[1] for r in recs), dtype=np.int32, count=B)
        PP = ctypes.c_void_p * B
        zs, ps, xs, es, eas = (PP(*[r[2][q] for r in recs]) for q in range(5))
        ys = np.fromiter((it.y for it in items), dtype=np.float32, count=B)
        L = BatchLayout()
        rc = lib().conan_collate_layout(B, K, n_atoms.ctypes.data, n_bonds.ctypes.data, x_dim, ea_dim, ctypes.byref(L))
        if rc != 0:
            raise RuntimeError(f"conan_collate_layout failed ({rc})")
        slot = self._pack_slot
        self._pack_slot = (slot + 1) % self.depth
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
        return L, pinned, [it.smiles for it in items], slot

    # ---------------------------------------------------------------------------------------------- device half
    def _outputs(self, L: BatchLayout):
        """The expanded tensors of a batch.  static=True: allocated once as views of ONE buffer — in fact two of them: `fixed`, the addresses
        the model reads, and `landing`, where the expansion kernel writes (DeviceBatch.wait copies landing -> fixed in one transfer)."""
        dev = self.device
        A, E, G = L.num_atoms, L.num_bond_edges, L.num_graphs
        if self.static and self._out is not None:
            o = self._out[0]
            if (o["z"].shape[0], o["edge_index"].shape[1], o["y"].shape[0]) != (A, E, G):
                raise RuntimeError("static DeviceCollator: the batch shape changed (atoms / bond edges / graphs); "
                                   "captured-graph replay needs shape-stable batches")
            return self._out
        spec = [("z", (A,), torch.int64), ("pos", (A, 3), torch.float32), ("batch", (A,), torch.int64), ("x", (A, L.x_dim), torch.float32),
                ("edge_index", (2, E), torch.int64), ("edge_attr", (E, L.ea_dim), torch.float32), ("y", (G,), torch.float32),
                ("graph_ptr", (G + 1,), torch.int32), ("conformers_index", (G,), torch.int64), ("conf_node_batch", (A,), torch.int64)]
        if not self.static:
            return {k: torch.empty(*shp, dtype=dt, device=dev) for k, shp, dt in spec}, None, None, None
        offs, total = [], 0
        for _k, shp, dt in spec:
            nbytes = int(np.prod(shp)) * torch.empty((), dtype=dt).element_size()
            offs.append(total)
            total += (nbytes + 255) & ~255
        flats = [torch.empty(max(total, 256), dtype=torch.uint8, device=dev) for _ in range(2)]

        def views(flat):
            out = {}
            for (k, shp, dt), off in zip(spec, offs):
                n = int(np.prod(shp))
                out[k] = flat[off: off + n * torch.empty((), dtype=dt).element_size()].view(dt).view(*shp)
            return out
        for f in flats:
            f.record_stream(self._main_stream)               # (allocated under the copy stream by enqueue; both are also used on the caller's)
        self._out = (views(flats[0]), views(flats[1]), flats[0], flats[1])      # (fixed views, landing views, fixed, landing)
        return self._out

    def __call__(self, batch_items: Sequence) -> DeviceBatch:
        return self.enqueue(*self.pack(batch_items))

    def enqueue(self, L: BatchLayout, pinned, smiles, slot: int, main: Optional[torch.cuda.Stream] = None, host_wait: bool = False) -> DeviceBatch:
        """Device half: the H2D copy and the expansion kernel of a packed batch on the copy stream; returns immediately.  `main`: the stream
        that will consume the batch (default: the calling thread's current stream).  `host_wait` (static collators): wait on the HOST until the
        previous batch has left the landing copy instead of queueing a device-side wait for it — for a caller that runs ahead of the GPU, see
        CollatePipeline."""
        cs = self.copy_stream
        main = self._main_stream = main if main is not None else torch.cuda.current_stream(self.device)
        if self._staged[slot] is None or self._staged[slot].numel() < L.bytes:
            with torch.cuda.stream(cs):                      # allocated on the stream that writes it: the caching allocator orders a
                self._staged[slot] = torch.empty(pinned.numel(), dtype=torch.uint8, device=self.device)      # reused block behind its last use there
        staged = self._staged[slot]
        with torch.cuda.stream(cs):
            staged[: L.bytes].copy_(pinned[: L.bytes], non_blocking=True)      # overlaps whatever the caller's stream is running
        with torch.cuda.stream(cs):
            o, landing_views, fixed, landing = self._outputs(L)
            w = landing_views if self.static else o           # static: the kernel writes the landing copy (see DeviceBatch.wait)
            if self.static:
                if self._landing_busy:
                    raise RuntimeError("static DeviceCollator: call .wait() on the previous batch before assembling the next one "
                                       "(there is one landing copy)")
                if self._static_max_nodes is None:
                    self._static_max_nodes = int(L.max_nodes)
                elif self.strict_max_nodes and int(L.max_nodes) != self._static_max_nodes:
                    raise RuntimeError(f"static DeviceCollator: the largest conformer of this batch has {int(L.max_nodes)} atoms, the first batch's "
                                       f"had {self._static_max_nodes}; a captured step was sized for the latter (strict_max_nodes=False for eager consumers)")
                if self._consumed is not None:
                    if host_wait:
                        self._consumed.synchronize()
                    else:
                        cs.wait_event(self._consumed)         # the previous batch has left the landing copy
                self._landing_busy = True
            call("conan_collate_unpack", ptr(staged), ctypes.byref(L), ptr(w["z"]), ptr(w["pos"]), ptr(w["batch"]), ptr(w["x"]) if L.x_dim else None,
                 ptr(w["edge_index"]) if L.num_bond_edges else None, ptr(w["edge_attr"]) if (L.num_bond_edges and L.ea_dim) else None,
                 ptr(w["y"]), ptr(w["graph_ptr"]), ptr(w["conformers_index"]), ptr(w["conf_node_batch"]), stream_ptr())
            ev = torch.cuda.Event()
            ev.record(cs)
        self._events[slot] = ev
        if not self.static:
            for t in o.values():
                t.record_stream(main)                        # allocated on the copy stream, consumed on the caller's
        staged.record_stream(cs)
        b = DeviceBatch(**o, batch_node_index=o["batch"], smiles=[s for s in smiles for _ in range(L.K)], num_graphs=int(L.num_graphs), max_nodes=int(L.max_nodes),
                        num_molecules=int(L.B), num_conformers=int(L.K), ready=ev)
        # The reference's call shape is forward(batch, conformers_index, node_index) with no size arguments (schnet_based_models.py:135-173):
        # the host-known sizes ride on the index tensors themselves, so that an unchanged harness reaches the sync-free path (ops.batch_hints).
        if self.static:
            b._static_pair = (self, fixed, landing)               # (shared fixed views: DeviceBatch.wait attaches the sizes)
        else:
            o["batch"]._conan_hints = (int(L.num_graphs), int(L.max_nodes))
        return b


class CollatePipeline:
    """Batches ready on the device, assembled ahead of the consumer: a worker thread runs the host half (`DeviceCollator.pack`: C memcpy into a
    pinned ring, the GIL released) AND enqueues the copy + expansion kernel on the copy stream, `prefetch` batches ahead (non-static collators)
    or one batch ahead (static: there is one landing copy), while the calling thread only launches steps.  Iterating yields `DeviceBatch`
    objects in source order (call `.wait()` on each, as with the collator itself).  The reference gets the same overlap from DataLoader worker
    processes (datamodules.py); here one thread suffices: the host half of a 256-molecule batch is ~0.2 ms.

        for batch in CollatePipeline(collator, loader):      # loader yields lists of dataset items
            step(batch.wait())

    Static collators: the worker waits ON THE HOST for the previous batch to have left the landing copy before it enqueues the next expansion.
    A consumer that replays captured steps runs several steps ahead of the GPU; a device-side wait queued that far ahead sits at the head of
    the copy stream's hardware queue for milliseconds, and every variant of that measured slower (round 4: fed step 2.69 ms against 2.50 resident
    with the copy stream sharing a hardware queue, 3.7 ms with a queue of its own) than never queueing a wait that is not already satisfied."""

    def __init__(self, collator: DeviceCollator, source, prefetch: int = 2):
        import queue
        import threading
        if collator.depth < prefetch + 2:
            raise ValueError(f"CollatePipeline(prefetch={prefetch}) needs a DeviceCollator with depth >= {prefetch + 2} pinned buffers")
        self.collator, self._q, self._src, self._stop = collator, queue.Queue(maxsize=prefetch), iter(source), False
        self._consumer_stream = torch.cuda.current_stream(collator.device)
        self._free = None
        if collator.static:
            if collator._landing_busy:
                raise RuntimeError("CollatePipeline: call .wait() on the collator's outstanding batch first")
            self._free = collator._landing_released = threading.Semaphore(1)
        self._thread = threading.Thread(target=self._run, name="conan-collate", daemon=True)
        self._thread.start()

    def _run(self):
        try:
            torch.cuda.set_device(self.collator.device)
            for items in self._src:
                if self._stop:
                    return
                p = self.collator.pack(items)
                if self._free is not None:
                    self._free.acquire()                       # the consumer has called wait() on the previous batch ...
                    if self._stop:
                        return
                self._q.put(self.collator.enqueue(*p, main=self._consumer_stream, host_wait=True))      # ... and the GPU has executed its landing copy
            self._q.put(None)
        except BaseException as e:                           # noqa: BLE001 - re-raised in the consumer
            self._q.put(e)

    def __iter__(self):
        return self

    def __next__(self) -> DeviceBatch:
        b = self._q.get()
        if b is None:
            raise StopIteration
        if isinstance(b, BaseException):
            raise b
        return b

    def close(self):
        """Stop the worker.  A static collator's batch that was already assembled is handed back by marking the landing copy free."""
        self._stop = True
        if self._free is not None:
            self._free.release()
        try:
            while True:
                self._q.get_nowait()
        except Exception:                                     # noqa: BLE001 - queue.Empty
            pass
        self._thread.join(timeout=5.0)
        try:
            while True:
                self._q.get_nowait()
        except Exception:                                     # noqa: BLE001
            pass
        if self.collator.static:
            self.collator._landing_released = None
            self.collator._landing_busy = False                # (an assembled, never consumed batch: the landing copy is simply overwritten next time)


def host_pack_benchmark(batch_items: Sequence, num_conformers: int, reps: int = 20) -> float:
    """Milliseconds per host-side pack of one batch (item records cached, conan_collate_layout + conan_collate_pack into plain host memory):
    the host half of the collator without a GPU — bench.py runs it in eight processes at once to see what eight ranks of one node cost each
    other."""
    import time
    items = _as_items(batch_items)
    B, K = len(items), int(num_conformers)
    x_dim, ea_dim = items[0].x.shape[1], items[0].edge_attr.shape[1]
    buf = None
    t0 = 0.0
    for r in range(reps + 2):
        if r == 2:
            t0 = time.perf_counter()
        recs = [_item_record(it, K, x_dim, ea_dim) for it in items]
        n_atoms = np.fromiter((q[0] for q in recs), dtype=np.int32, count=B)
        n_bonds = np.fromiter((q[1] for q in recs), dtype=np.int32, count=B)
        PP = ctypes.c_void_p * B
        zs, ps, xs, es, eas = (PP(*[q[2][c] for q in recs]) for c in range(5))
        ys = np.fromiter((it.y for it in items), dtype=np.float32, count=B)
        L = BatchLayout()
        if lib().conan_collate_layout(B, K, n_atoms.ctypes.data, n_bonds.ctypes.data, x_dim, ea_dim, ctypes.byref(L)) != 0:
            raise RuntimeError("conan_collate_layout failed")
        if buf is None or buf.size < L.bytes:
            buf = np.empty(int(L.bytes) + 256, dtype=np.uint8)
        if lib().conan_collate_pack(ctypes.byref(L), n_atoms.ctypes.data, n_bonds.ctypes.data, zs, ps, xs, es, eas, ys.ctypes.data, buf.ctypes.data) != 0:
            raise ValueError("collate: bad item")
    return 1e3 * (time.perf_counter() - t0) / reps


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
