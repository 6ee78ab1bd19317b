"""Operators of the ConAN hot path on MI355X: thin torch.autograd wrappers over the C-ABI (include/conan_fgw_hip.h).

torch is used for device memory, streams and autograd bookkeeping only; every computation below is a HIP kernel of
libconan_fgw_hip.so.  All ops require CUDA(ROCm) tensors and raise otherwise.
"""
from __future__ import annotations

from typing import Optional

import torch
from torch import Tensor

from ._lib import FgwParams, call, lib, ptr, stream_ptr

f32, i32, i64 = torch.float32, torch.int32, torch.int64


def _c(t: Tensor) -> Tensor:
    return t if t.is_contiguous() else t.contiguous()


# ------------------------------------------------------------------------------------------------ graphs
class RadiusGraph:
    """Device-resident neighbour lists of a batch of conformer graphs: CSR by target (+ lazily its by-source transpose).

    Counterpart of what `RadiusInteractionGraph.forward` returns in the reference (schnet_no_sum.py:160,208,342), kept in
    the layout the kernels consume.  `edge_index()` / `edge_weight()` export the reference's tensors (one host sync for E).
    """

    def __init__(self, pos: Tensor, graph_ptr: Tensor, num_graphs: int, cutoff: float, max_num_neighbors: int, loop: bool = False):
        pos = _c(pos)
        if pos.dtype != f32:
            raise RuntimeError("pos must be float32")
        n = pos.shape[0]
        dev = pos.device
        self.num_atoms, self.num_graphs, self.graph_ptr = n, num_graphs, graph_ptr
        self.cutoff, self.cap, self.loop = float(cutoff), int(max_num_neighbors), bool(loop)
        # worst case per target: cap + 1 edges without self loops (the self hit may fall outside torch-cluster's cap + 1 window)
        self.max_edges = max(1, n * (self.cap if self.loop else self.cap + 1))
        self.rowptr = torch.empty(n + 1, dtype=i32, device=dev)
        self.col = torch.empty(self.max_edges, dtype=i32, device=dev)
        self.tgt = torch.empty(self.max_edges, dtype=i32, device=dev)
        self.dist = torch.empty(self.max_edges, dtype=f32, device=dev)
        self._deg = torch.empty(n + 1, dtype=i32, device=dev)
        call("conan_radius_graph_csr", ptr(pos), ptr(graph_ptr, i32), n, num_graphs, self.cutoff, self.cap, int(self.loop),
             ptr(self._deg), ptr(self.rowptr), ptr(self.col), ptr(self.tgt), ptr(self.dist), stream_ptr())
        self.num_edges_dev = self.rowptr[n:]            # device-side edge count (1-element view)
        self._num_edges: Optional[int] = None
        self._t_rowptr = self._t_eid = None
        self.pid = None

    @property
    def num_edges(self) -> int:
        if self._num_edges is None:
            self._num_edges = int(self.num_edges_dev.item())     # host sync
        return self._num_edges

    def transpose(self):
        if self._t_rowptr is None:
            dev = self.rowptr.device
            self._t_rowptr = torch.empty(self.num_atoms + 1, dtype=i32, device=dev)
            self._t_eid = torch.empty(self.max_edges, dtype=i32, device=dev)
            call("conan_csr_transpose", ptr(self.graph_ptr), self.num_graphs, self.num_atoms, ptr(self.rowptr), ptr(self.col),
                 ptr(self._deg), ptr(self._t_rowptr), ptr(self._t_eid), stream_ptr())
        return self._t_rowptr, self._t_eid

    def pairs(self):
        """Undirected pairs of the edge set (conan_edge_pairs): the continuous filter is evaluated once per pair.
        Returns self; fills pid [max_edges], pair_e0/pair_e1/pair_dist [max_edges] and num_pairs_dev (device int)."""
        if getattr(self, "pid", None) is None:
            dev, ME = self.rowptr.device, self.max_edges
            flag = torch.empty(ME + 1, dtype=i32, device=dev)
            pidx = torch.empty(ME + 1, dtype=i32, device=dev)
            scan_ws = torch.empty(2 * (ME // 4096 + 2), dtype=i32, device=dev)
            self.pid = torch.empty(ME, dtype=i32, device=dev)
            self.pair_e0 = torch.empty(ME, dtype=i32, device=dev)
            self.pair_e1 = torch.empty(ME, dtype=i32, device=dev)
            self.pair_dist = torch.empty(ME, dtype=f32, device=dev)
            call("conan_edge_pairs", ptr(self.rowptr), ptr(self.col), ptr(self.tgt), ptr(self.dist), ptr(self.num_edges_dev), ME, ptr(flag),
                 ptr(pidx), ptr(scan_ws), ptr(self.pid), ptr(self.pair_e0), ptr(self.pair_e1), ptr(self.pair_dist), stream_ptr())
            self.num_pairs_dev = pidx[ME:]
        return self

    def edge_index(self) -> Tensor:
        E = self.num_edges
        ei = torch.empty(2, E, dtype=i64, device=self.rowptr.device)
        call("conan_edge_index_i64", ptr(self.col), ptr(self.tgt), E, ptr(ei), stream_ptr())
        return ei

    def edge_weight(self) -> Tensor:
        return self.dist[: self.num_edges]


def empty_rows(rows: int, width: int, device, m_dev: Optional[Tensor]) -> Tensor:
    """[rows, width] fp32 buffer whose rows beyond the device-side count `m_dev` are zero (the rows below it are for the caller's
    kernel to write): a tail-only clear instead of a full memset of a worst-case-sized edge buffer."""
    t = torch.empty(rows, width, dtype=f32, device=device)
    if m_dev is not None:
        call("conan_zero_tail", ptr(t), ptr(m_dev), rows, width, stream_ptr())
    return t


# Debug switch for the "rows beyond the device-side count are never read" contract of the worst-case-sized edge buffers (unread_rows): when set,
# those buffers start as NaN instead of whatever the allocator hands out, so that a consumer that does read the tail shows up as non-finite
# results (tests/test_gpu_visnet.py runs a whole training step both ways and demands identical bits).
POISON_UNREAD_TAILS = False


def unread_rows(rows: int, width: int, device) -> Tensor:
    """[rows, width] fp32 buffer of which the caller's kernel writes the first m_dev rows and NOBODY reads the rest (every consumer walks the CSR
    or takes the same device-side count): no clear at all."""
    t = torch.empty(rows, width, dtype=f32, device=device)
    if POISON_UNREAD_TAILS:
        t.fill_(float("nan"))
    return t


def batch_hints(batch: Optional[Tensor]):
    """(num_graphs, max_nodes) that DeviceCollator attached to the node -> graph index tensor it produced (host-known from the item sizes), or
    (None, None): a caller that passes no size arguments — the reference's call shape — then needs no device -> host read to size anything."""
    h = getattr(batch, "_conan_hints", None) if batch is not None else None
    return h if h is not None else (None, None)


def graph_ptr_from_batch(batch: Tensor, num_graphs: int) -> Tensor:
    batch = _c(batch)
    out = torch.empty(num_graphs + 1, dtype=i32, device=batch.device)
    call("conan_graph_ptr_from_batch", ptr(batch, i64), batch.shape[0], num_graphs, ptr(out), stream_ptr())
    return out


# ------------------------------------------------------------------------------------------------ deferred weight gradients
# A weight gradient is produced in two stages: per-slice slabs (k_wgrad_lds) and their fixed-order sum.  Inside
# `deferred_weight_gradients()` the second stage of every Linear layer of a backward pass is postponed and run as ONE launch
# (conan_wgrad_reduce_batch) by `flush_weight_gradients()` — FlatGradients.pack() calls it — instead of 24 launches of ~6 us each.
# The returned dW / db tensors are only valid after the flush; a weight that appears twice in one backward (autograd would add
# the two results right away) flushes on the spot and takes the immediate path.  Node-level layers postpone their slab kernel as well
# (one batched launch).  With the library's slice count (LATE_SLICES_AUTO = False) results do not depend on the mode, bit for bit; by default the batch
# cuts every job into fewer, longer slices (_late_slices): the same sums in another fixed order (1e-7 relative), 15-18 % less time for the pair of launches.
_pending = None            # None: immediate mode; list of pending jobs (dicts) while deferring


class deferred_weight_gradients:
    def __enter__(self):
        global _pending
        self._outer = _pending
        if _pending is None:
            _pending = []
        return self

    def __exit__(self, *exc):
        global _pending
        if self._outer is None:
            flush_weight_gradients()
            _pending = None
        return False


def flush_weight_gradients():
    """Reduce every pending slab set on the current stream (no-op when nothing is pending).  Returns {weight data_ptr: (dW data_ptr,
    db data_ptr | None)} of what was flushed, so that the owner of the parameters can check that autograd adopted those very tensors
    (FlatGradients._flush_deferred does, for dW and db).  The deferred mode is only sound behind that check: use it through
    FlatGradients.backward(), not as a bare `with deferred_weight_gradients(): loss.backward()`."""
    global _pending
    if not _pending:
        return {}
    import ctypes
    from ._lib import WgradJob, WgradSlabJob
    jobs = (WgradJob * len(_pending))()
    cur = torch.cuda.current_stream()
    for st in {j["stream"] for j in _pending}:
        if st != cur:
            cur.wait_stream(st)                                   # slabs written on another stream (the covalent branch runs on one)
    late = [j for j in _pending if "operands" in j]               # node-level layers: stage 1 was postponed as well (see _wgrad)
    if late:
        sj = (WgradSlabJob * len(late))()
        for q, j in enumerate(late):
            g, x, md = j["operands"]
            sj[q].g, sj[q].x, sj[q].m_dev, sj[q].ws = ptr(g), ptr(x), ptr(md), ptr(j["ws"])
            sl = _late_slices(len(late), j["M"])
            sj[q].M, sj[q].K, sj[q].N, sj[q].slices = j["M"], j["K"], j["N"], sl      # 0: the library's default slice count
            if sl:
                j["slices"] = sl
        call("conan_linear_wgrad_slabs_batch", sj, len(late), stream_ptr())
        for j in late:
            for t in j.pop("operands"):
                if t is not None:
                    t.record_stream(cur)
    for q, j in enumerate(_pending):
        jobs[q].ws, jobs[q].dW, jobs[q].dbias = ptr(j["ws"]), j["dw_ptr"], j["db_ptr"]
        jobs[q].M, jobs[q].K, jobs[q].N, jobs[q].slices = j["M"], j["K"], j["N"], j.get("slices", 0)
    call("conan_wgrad_reduce_batch", jobs, len(_pending), stream_ptr())
    done = {j["weight_ptr"]: (j["dw_ptr"], j["db_ptr"]) for j in _pending}
    for j in _pending:
        j["ws"].record_stream(cur)
    _pending.clear()
    _flushed.update(done)
    return done


_LATE_STAGE1_ROWS = 65536   # below this row count a weight gradient's slab kernel is postponed to the batched launch
LATE_SLICES = 0             # row slices per postponed job, forced (0 = automatic, below): tools/probe_wgrad_batch.py sweeps it
LATE_SLICES_AUTO = True     # False: the library's default (one slice per 128 rows), i.e. the same slabs — and bits — as the immediate form


def _late_slices(n_jobs: int, M: int) -> int:
    """Row slices per postponed node-level job.  The library's default (one per 128 rows: 198 at cfg2) gives a 22-job batch 4 356 workgroups, each
    writing a 64 KB slab for 8 stages of work; measured in one process (tools/probe_wgrad_batch.py, profiles/r5_wgrad_batch_slices.txt): 198 slices
    249-261 us, 128: 218-227, 96: 212-219, 80: 217, 64: 207-209, 48: 238, 24: 270 — best where a launch (<= 24 jobs) holds ~1 400 workgroups, a
    multiple of 8 per job (the XCD grouping of jobs that share x).  Fixed order of summation either way; not the same order as the default's."""
    dflt = max(1, (M + 127) // 128)
    if LATE_SLICES:
        return LATE_SLICES if LATE_SLICES < dflt else 0           # (never more than the default: the workspace and the reducer are sized by it)
    if not LATE_SLICES_AUTO:
        return 0
    s = 8 * max(1, round(1400 / max(1, min(n_jobs, 24)) / 8))
    return s if s < dflt else 0
_flushed = {}              # weight data_ptr -> (dW data_ptr, db data_ptr | None) of the last flushes (cleared by whoever verifies them)


# max |g| of pair-level filter gradients: produced by conan_cfconv_bwd_w_pairs (one device float per backward of a CFConv), consumed by
# the filter network's backward, whose two MFMA kernels then run on two fp16 planes (half the matrix-pipe work).  It travels ON the
# gradient tensor (attribute _conan_gmax = (device float, tensor version at production)): a gradient that autograd copied or summed into
# a new tensor has no attribute, one it accumulated into IN PLACE has another version — both take the bf16 path, which needs no scale.
# (Round 3 kept a module-level dict keyed by data_ptr: a pointer is not an identity — stale entries, in-place sums.)
gmax_stats = {"tracked": 0, "used": 0}


def _tag_gmax(t: Tensor, gmax: Tensor):
    t._conan_gmax = (gmax, t._version)
    gmax_stats["tracked"] += 1


def _take_gmax(t: Tensor, g: Tensor):
    tag = getattr(t, "_conan_gmax", None)
    if tag is None or tag[1] != t._version or g.data_ptr() != t.data_ptr():
        return None
    gmax_stats["used"] += 1
    return tag[0]


def _wgrad(g, x, M, K, N, md, weight, has_bias, rbf=None, gmax=None):
    """dW [N,K] (+ db [N]) = g^T x, or g^T rbf(dist) with rbf = (dist, offset, coeff).  Immediate, or slabs now + batched sum later.

    Deferred mode hands autograd tensors whose values arrive at the flush.  That is only sound if autograd ADOPTS them (it does when it
    holds the only reference and the parameter has no .grad yet; otherwise it copies on the spot), so the pending list keeps the raw
    pointers and the storages — never the tensors — and FlatGradients.pack() verifies the adoption."""
    dev = g.device
    ws = torch.empty(int(lib().conan_linear_wgrad_ws(M, K, N)), dtype=f32, device=dev)
    dw = torch.empty(N, K, dtype=f32, device=dev)
    db = torch.empty(N, dtype=f32, device=dev) if has_bias else None
    wptr = weight.data_ptr()
    defer = _pending is not None and bool(lib().conan_wgrad_batchable(K, N))
    if defer and any(j["weight_ptr"] == wptr for j in _pending):
        flush_weight_gradients()                                  # second use of the same weight in this backward: autograd adds the two at once
        defer = False
    # Node-level layers (a few ten thousand rows) are latency chains that leave most of the chip idle: in deferred mode their slab
    # kernels are postponed too and all of them run as ONE launch at the flush (conan_linear_wgrad_slabs_batch).  g and x stay alive
    # until then (a node-level pair is 26 MB); edge-level layers already fill the chip and keep their immediate stage 1.
    late = defer and rbf is None and M <= _LATE_STAGE1_ROWS and not (gmax is not None and K > 64)      # (the fp16-plane form has no batched launch)
    if rbf is None:
        if late:
            pass
        elif gmax is not None and K > 64:
            call("conan_linear_wgrad_scaled", ptr(g), ptr(x), M, K, N, ptr(md), None if defer else ptr(dw), None if defer else ptr(db), ptr(ws),
                 ptr(gmax), stream_ptr())
        elif defer:
            call("conan_linear_wgrad_slabs", ptr(g), ptr(x), M, K, N, ptr(md), ptr(ws), stream_ptr())
        else:
            call("conan_linear_wgrad", ptr(g), ptr(x), M, K, N, ptr(md), ptr(dw), ptr(db), ptr(ws), stream_ptr())
    else:
        dist, offset, coeff = rbf
        if defer:
            call("conan_rbf_wgrad_slabs", ptr(g), ptr(dist), M, ptr(offset, f32), K, coeff, N, ptr(md), ptr(ws), stream_ptr())
        else:
            call("conan_rbf_wgrad", ptr(g), ptr(dist), M, ptr(offset, f32), K, coeff, N, ptr(md), ptr(dw), ptr(db), ptr(ws), stream_ptr())
    if defer:
        _pending.append(dict(ws=ws, dw_ptr=dw.data_ptr(), db_ptr=db.data_ptr() if db is not None else None,
                             keep=(dw.untyped_storage(), db.untyped_storage() if db is not None else None),
                             M=M, K=K, N=N, weight_ptr=wptr, stream=torch.cuda.current_stream()))
        if late:
            _pending[-1]["operands"] = (g, x, md)
    return dw, db


def _wgrad_shared_x(gs, x, M, K, N, md, weights, has_bias):
    """[_wgrad(g_i, x, ...) for g_i in gs] for several Linear layers of the SAME input and the same width (dk / dv / f_proj of f; q / k / v): the
    slab kernels of the run are ONE launch whose workgroups for one row slice sit next to each other, so x is streamed from HBM once per run
    instead of once per layer (conan_linear_wgrad_slabs_batch; the slabs, and with them the results, are bit for bit those of the separate
    launches).  Falls back to the separate launches where the batched kernel does not apply."""
    n = len(gs)
    usable = n > 1 and bool(lib().conan_wgrad_batchable(K, N)) and K > 64 and N <= 128 and M > _LATE_STAGE1_ROWS      # (node level: the late batch groups such runs itself)
    wptrs = [w.data_ptr() for w in weights]
    if usable and _pending is not None and (len(set(wptrs)) < n or any(j["weight_ptr"] in wptrs for j in _pending)):
        usable = False                                            # a weight used twice in one backward: the immediate path of _wgrad handles it
    if not usable:
        return [_wgrad(g, x, M, K, N, md, w, hb) for g, w, hb in zip(gs, weights, has_bias)]
    from ._lib import WgradJob, WgradSlabJob
    dev = x.device
    wsz = int(lib().conan_linear_wgrad_ws(M, K, N))
    wss = [torch.empty(wsz, dtype=f32, device=dev) for _ in range(n)]
    dws = [torch.empty(N, K, dtype=f32, device=dev) for _ in range(n)]
    dbs = [torch.empty(N, dtype=f32, device=dev) if hb else None for hb in has_bias]
    sj = (WgradSlabJob * n)()
    for q in range(n):
        sj[q].g, sj[q].x, sj[q].m_dev, sj[q].ws = ptr(gs[q]), ptr(x), ptr(md), ptr(wss[q])
        sj[q].M, sj[q].K, sj[q].N, sj[q].slices = M, K, N, 0
    call("conan_linear_wgrad_slabs_batch", sj, n, stream_ptr())
    if _pending is not None:
        for q in range(n):
            _pending.append(dict(ws=wss[q], dw_ptr=dws[q].data_ptr(), db_ptr=dbs[q].data_ptr() if dbs[q] is not None else None,
                                 keep=(dws[q].untyped_storage(), dbs[q].untyped_storage() if dbs[q] is not None else None),
                                 M=M, K=K, N=N, weight_ptr=wptrs[q], stream=torch.cuda.current_stream()))
    else:
        jobs = (WgradJob * n)()
        for q in range(n):
            jobs[q].ws, jobs[q].dW, jobs[q].dbias = ptr(wss[q]), dws[q].data_ptr(), dbs[q].data_ptr() if dbs[q] is not None else None
            jobs[q].M, jobs[q].K, jobs[q].N, jobs[q].slices = M, K, N, 0
        call("conan_wgrad_reduce_batch", jobs, n, stream_ptr())
    return list(zip(dws, dbs))


def _filter_bwd(g, h1, dist, offset, coeff, w1, w2, M, md, gmax=None):
    """dW1 [F,Gs], db1 [F] of the filter network's first Linear from the gradient g of its output, fused (conan_filter_bwd): the
    input gradient of the second Linear times ssp'(h1) is formed tile by tile in registers and contracted with the regenerated
    rbf(dist) on the spot.  Immediate, or slabs now + batched sum later (see _wgrad)."""
    F, Gs = w1.shape
    dev = g.device
    ws = torch.empty(int(lib().conan_filter_bwd_ws(M, Gs, F)), dtype=f32, device=dev)
    dw = torch.empty(F, Gs, dtype=f32, device=dev)
    db = torch.empty(F, dtype=f32, device=dev)
    wptr = w1.data_ptr()
    defer = _pending is not None
    if defer and any(j["weight_ptr"] == wptr for j in _pending):
        flush_weight_gradients()
        defer = False
    call("conan_filter_bwd", ptr(g), ptr(h1), ptr(dist), M, ptr(offset, f32), Gs, coeff, ptr(w2), F, ptr(md),
         None if defer else ptr(dw), None if defer else ptr(db), ptr(ws), ptr(gmax), stream_ptr())
    if defer:
        _pending.append(dict(ws=ws, dw_ptr=dw.data_ptr(), db_ptr=db.data_ptr(), keep=(dw.untyped_storage(), db.untyped_storage()),
                             M=M, K=Gs, N=F, slices=int(lib().conan_filter_bwd_slices(M)), weight_ptr=wptr, stream=torch.cuda.current_stream()))
    return dw, db


def _filter_bwd2(g, h1, dist, offset, coeff, w1, w2, M, md, gmax):
    """(dW1 [F,Gs], db1 [F]), (dW2 [F,F], db2 [F]) of the filter network from the gradient g of its output in ONE pass over g and h1
    (conan_filter_bwd2: _filter_bwd and the second layer's _wgrad fused).  Immediate, or slabs now + batched sum later (see _wgrad)."""
    F, Gs = w1.shape
    dev = g.device
    ws = torch.empty(int(lib().conan_filter_bwd2_ws(M, Gs, F)), dtype=f32, device=dev)
    slices = int(lib().conan_filter_bwd2_slices(M))
    dw1, db1 = torch.empty(F, Gs, dtype=f32, device=dev), torch.empty(F, dtype=f32, device=dev)
    dw2, db2 = torch.empty(F, F, dtype=f32, device=dev), torch.empty(F, dtype=f32, device=dev)
    p1, p2 = w1.data_ptr(), w2.data_ptr()
    defer = _pending is not None
    if defer and any(j["weight_ptr"] in (p1, p2) for j in _pending):
        flush_weight_gradients()
        defer = False
    call("conan_filter_bwd2", ptr(g), ptr(h1), ptr(dist), M, ptr(offset, f32), Gs, coeff, ptr(w2), F, ptr(md), ptr(gmax),
         None if defer else ptr(dw1), None if defer else ptr(db1), None if defer else ptr(dw2), None if defer else ptr(db2), ptr(ws), stream_ptr())
    if defer:
        cut = slices * (F * Gs + F)
        for wsv, dw, db, K, wp in ((ws[:cut], dw1, db1, Gs, p1), (ws[cut:], dw2, db2, F, p2)):
            _pending.append(dict(ws=wsv, dw_ptr=dw.data_ptr(), db_ptr=db.data_ptr(), keep=(dw.untyped_storage(), db.untyped_storage()),
                                 M=M, K=K, N=F, slices=slices, weight_ptr=wp, stream=torch.cuda.current_stream()))
    return (dw1, db1), (dw2, db2)


# ------------------------------------------------------------------------------------------------ linear / activation
class _LinearFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b, residual, act, m_dev, grad_tail_unread=False):
        x, w = _c(x), _c(w)
        M, K = x.shape
        N = w.shape[0]
        y = torch.empty(M, N, dtype=f32, device=x.device)
        call("conan_linear_fwd", ptr(x, f32), ptr(w, f32), ptr(b), ptr(_c(residual)) if residual is not None else None,
             M, K, N, 0, act, ptr(m_dev), ptr(y), stream_ptr())
        ctx.act, ctx.m_dev, ctx.has_b, ctx.has_res, ctx.grad_tail_unread = act, m_dev, b is not None, residual is not None, grad_tail_unread
        ctx.save_for_backward(x, w, y if act else None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, y = ctx.saved_tensors
        dy = _c(dy)
        M, K = x.shape
        N = w.shape[0]
        md = ctx.m_dev
        if ctx.act:
            if ctx.has_res:
                raise RuntimeError("act + residual backward is not defined for this op")
            g = torch.empty_like(dy)
            call("conan_ssp_bwd", ptr(dy), ptr(y), M, N, ptr(md), ptr(g), stream_ptr())
        else:
            g = dy
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            if ctx.grad_tail_unread:       # the caller's promise: whatever consumes dx takes md or walks the CSR (ViS_MP's s_proj: 22 us per layer for a tail nobody reads)
                dx = unread_rows(x.shape[0], x.shape[1], x.device)
            else:
                dx = empty_rows(x.shape[0], x.shape[1], x.device, md)
            call("conan_linear_fwd", ptr(g), ptr(w), None, None, M, N, K, 1, 0, ptr(md), ptr(dx), stream_ptr())
        if ctx.needs_input_grad[1] or (ctx.has_b and ctx.needs_input_grad[2]):
            dw, db = _wgrad(g, x, M, K, N, md, w, ctx.has_b)
        return dx, dw, db, (dy if ctx.has_res else None), None, None, None


class _LinearTapFn(torch.autograd.Function):
    """(x W^T, x): a bias-free Linear that also hands its input through.  The second output is for a residual connection taken from the
    same x further down: its gradient then arrives HERE, together with the gradient of the product, and the backward forms
    dx = dy W + d_tap in the epilogue of the one input-gradient GEMM instead of leaving a separate add of two [M,K] tensors to autograd."""

    @staticmethod
    def forward(ctx, x, w):
        x, w = _c(x), _c(w)
        M, K = x.shape
        N = w.shape[0]
        y = torch.empty(M, N, dtype=f32, device=x.device)
        call("conan_linear_fwd", ptr(x, f32), ptr(w, f32), None, None, M, K, N, 0, 0, None, ptr(y), stream_ptr())
        ctx.save_for_backward(x, w)
        return y, x.view_as(x)

    @staticmethod
    def backward(ctx, dy, dtap):
        x, w = ctx.saved_tensors
        M, K = x.shape
        N = w.shape[0]
        dx = dw = None
        if dy is None:                                             # only the tap was used downstream
            return dtap, None
        dy = _c(dy)
        if ctx.needs_input_grad[0]:
            dx = torch.empty(M, K, dtype=f32, device=x.device)
            call("conan_linear_fwd", ptr(dy), ptr(w), None, ptr(_c(dtap)) if dtap is not None else None, M, N, K, 1, 0, None, ptr(dx), stream_ptr())
        elif dtap is not None:
            dx = dtap
        if ctx.needs_input_grad[1]:
            dw, _ = _wgrad(dy, x, M, K, N, None, w, False)
        return dx, dw


def linear_tap(x: Tensor, weight: Tensor):
    """Returns (x @ weight.T, x'): x' is x, to be used for a residual connection downstream (see _LinearTapFn)."""
    return _LinearTapFn.apply(x, weight)


def linear(x: Tensor, weight: Tensor, bias: Optional[Tensor] = None, act: bool = False, residual: Optional[Tensor] = None,
           m_dev: Optional[Tensor] = None, grad_tail_unread: bool = False) -> Tensor:
    """act(x @ weight.T + bias) (+ residual) with act = shifted softplus.  `m_dev`: device int32 row count (edge-level); the rows of the input
    gradient beyond it are zero unless `grad_tail_unread` (then undefined: for callers whose consumers never read them)."""
    return _LinearFn.apply(x, weight, bias, residual, 1 if act else 0, m_dev, grad_tail_unread)


class _Mlp2Fn(torch.autograd.Function):
    """y = ssp(x w1^T + b1) w2^T + b2 (+ residual) in one launch; backward: one launch for both input-gradient GEMMs and the
    activation derivative between them, then the two weight gradients (conan_mlp2_fwd / conan_mlp2_bwd)."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, residual):
        x, w1, w2 = _c(x), _c(w1), _c(w2)
        M, K = x.shape
        N1, N2 = w1.shape[0], w2.shape[0]
        need = any(ctx.needs_input_grad[:5])
        mid = torch.empty(M, N1, dtype=f32, device=x.device) if need else None
        y = torch.empty(M, N2, dtype=f32, device=x.device)
        call("conan_mlp2_fwd", ptr(x, f32), ptr(w1, f32), ptr(_c(b1), f32), ptr(w2, f32), ptr(_c(b2), f32),
             ptr(_c(residual)) if residual is not None else None, M, K, N1, N2, ptr(mid), ptr(y), stream_ptr())
        ctx.has_res = residual is not None
        ctx.save_for_backward(x, w1, w2, mid)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w1, w2, mid = ctx.saved_tensors
        dy = _c(dy)
        M, K = x.shape
        N1, N2 = w1.shape[0], w2.shape[0]
        dmid = torch.empty(M, N1, dtype=f32, device=x.device)
        dx = torch.empty(M, K, dtype=f32, device=x.device)
        call("conan_mlp2_bwd", ptr(dy), ptr(w2), ptr(w1), ptr(mid), M, K, N1, N2, ptr(dmid), ptr(dx), stream_ptr())
        dw2, db2 = _wgrad(dy, mid, M, N1, N2, None, w2, True)
        dw1, db1 = _wgrad(dmid, x, M, K, N1, None, w1, True)
        return dx, dw1, db1, dw2, db2, (dy if ctx.has_res else None)


def mlp2(x: Tensor, w1: Tensor, b1: Tensor, w2: Tensor, b2: Tensor, residual: Optional[Tensor] = None) -> Tensor:
    """ssp(x w1^T + b1) w2^T + b2 (+ residual).  One launch where conan_mlp2_supported (node-level rows, 128-wide layers), else the two
    linear kernels."""
    if x.is_cuda and b1 is not None and b2 is not None and lib().conan_mlp2_supported(x.shape[0], x.shape[1], w1.shape[0], w2.shape[0]):
        return _Mlp2Fn.apply(x, w1, b1, w2, b2, residual)
    return linear(linear(x, w1, b1, act=True), w2, b2, residual=residual)


class _Mlp2OutActFn(torch.autograd.Function):
    """y = ssp((x w1^T + b1) w2^T + b2) in one launch; backward: dy * ssp'(y), both input-gradient GEMMs in one launch, then the two weight
    gradients (conan_mlp2_outact_fwd / _bwd)."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2):
        x, w1, w2 = _c(x), _c(w1), _c(w2)
        M, K = x.shape
        N1, N2 = w1.shape[0], w2.shape[0]
        need = any(ctx.needs_input_grad)
        mid = torch.empty(M, N1, dtype=f32, device=x.device) if need else None
        y = torch.empty(M, N2, dtype=f32, device=x.device)
        call("conan_mlp2_outact_fwd", ptr(x, f32), ptr(w1, f32), ptr(_c(b1), f32), ptr(w2, f32), ptr(_c(b2), f32), M, K, N1, N2, ptr(mid), ptr(y),
             stream_ptr())
        ctx.save_for_backward(x, w1, w2, mid, y)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w1, w2, mid, y = ctx.saved_tensors
        dy = _c(dy)
        M, K = x.shape
        N1, N2 = w1.shape[0], w2.shape[0]
        dev = x.device
        g = torch.empty(M, N2, dtype=f32, device=dev)
        dmid = torch.empty(M, N1, dtype=f32, device=dev)
        dx = torch.empty(M, K, dtype=f32, device=dev)
        call("conan_mlp2_outact_bwd", ptr(dy), ptr(y), ptr(w2), ptr(w1), M, K, N1, N2, ptr(g), ptr(dmid), ptr(dx), stream_ptr())
        dw2, db2 = _wgrad(g, mid, M, N1, N2, None, w2, True)
        dw1, db1 = _wgrad(dmid, x, M, K, N1, None, w1, True)
        return dx, dw1, db1, dw2, db2


def mlp2_outact(x: Tensor, w1: Tensor, b1: Tensor, w2: Tensor, b2: Tensor) -> Tensor:
    """ssp((x w1^T + b1) w2^T + b2).  One launch where conan_mlp2_outact_supported (node-level rows, 128 -> 64 -> 64), else two linear kernels."""
    if x.is_cuda and b1 is not None and b2 is not None and lib().conan_mlp2_outact_supported(x.shape[0], x.shape[1], w1.shape[0], w2.shape[0]):
        return _Mlp2OutActFn.apply(x, w1, b1, w2, b2)
    return linear(linear(x, w1, b1), w2, b2, act=True)


class _Stage2HeadFn(torch.autograd.Function):
    """out = Linreg(mean_K(Lin3d(x3) + xc + aw * Linbary(xb))): one launch each way (conan_stage2_head_fwd / _bwd)."""

    @staticmethod
    def forward(ctx, x3, xc, xb, W3, b3, Wb, bb, wreg, breg, aw, K):
        x3, xc, xb = _c(x3), _c(xc), _c(xb)
        G, D = x3.shape
        if G % K != 0:
            raise ValueError(f"stage2_head: {G} conformer graphs are not a multiple of num_conformers={K}")
        B = G // K
        dev = x3.device
        out = torch.empty(B, 1, dtype=f32, device=dev)
        m3, mb, t = (torch.empty(B, D, dtype=f32, device=dev) for _ in range(3))
        call("conan_stage2_head_fwd", ptr(x3, f32), ptr(xc, f32), ptr(xb, f32), ptr(_c(W3), f32), ptr(_c(b3), f32), ptr(_c(Wb), f32), ptr(_c(bb), f32),
             ptr(_c(wreg), f32), ptr(_c(breg), f32), float(aw), B, K, D, ptr(out), ptr(m3), ptr(mb), ptr(t), stream_ptr())
        ctx.save_for_backward(W3, Wb, wreg, m3, mb, t)
        ctx.dims, ctx.aw = (B, K, D), float(aw)
        return out

    @staticmethod
    def backward(ctx, dout):
        W3, Wb, wreg, m3, mb, t = ctx.saved_tensors
        B, K, D = ctx.dims
        dev = dout.device
        dx3, dxc, dxb = (torch.empty(B * K, D, dtype=f32, device=dev) for _ in range(3))
        dW3, dWb = torch.empty(D, D, dtype=f32, device=dev), torch.empty(D, D, dtype=f32, device=dev)
        db3, dbb = torch.empty(D, dtype=f32, device=dev), torch.empty(D, dtype=f32, device=dev)
        dwreg, dbreg = torch.empty(1, D, dtype=f32, device=dev), torch.empty(1, dtype=f32, device=dev)
        call("conan_stage2_head_bwd", ptr(_c(dout)), ptr(_c(W3)), ptr(_c(Wb)), ptr(_c(wreg)), ptr(m3), ptr(mb), ptr(t), ctx.aw, B, K, D,
             ptr(dx3), ptr(dxc), ptr(dxb), ptr(dW3), ptr(db3), ptr(dWb), ptr(dbb), ptr(dwreg), ptr(dbreg), stream_ptr())
        return dx3, dxc, dxb, dW3, db3, dWb, dbb, dwreg, dbreg, None, None


def stage2_head(x3: Tensor, xc: Tensor, xb: Tensor, lin3d, linbary, linreg, agg_weight: float, K: int) -> Tensor:
    """[G,D] x 3 -> [G/K, 1]: Linreg(mean over the K conformers of (Lin3d(x3) + xc + agg_weight * Linbary(xb)))."""
    return _Stage2HeadFn.apply(x3, xc, xb, lin3d.weight, lin3d.bias, linbary.weight, linbary.bias, linreg.weight, linreg.bias, agg_weight, K)


def stage2_head_supported(D: int) -> bool:
    return bool(lib().conan_stage2_head_supported(int(D)))


class _UnaryFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, op):
        x = _c(x)
        y = torch.empty_like(x)
        call("conan_unary_fwd", ptr(x, f32), x.numel(), op, ptr(y), stream_ptr())
        ctx.op = op
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        dx = torch.empty_like(y)
        call("conan_unary_bwd", ptr(y), ptr(_c(dy)), y.numel(), ctx.op, ptr(dx), stream_ptr())
        return dx, None


def relu(x: Tensor) -> Tensor:
    return _UnaryFn.apply(x, 0)


def sigmoid(x: Tensor) -> Tensor:
    return _UnaryFn.apply(x, 1)


def shifted_softplus(x: Tensor) -> Tensor:
    return _UnaryFn.apply(x, 2)


class _EmbeddingFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, z, weight, padding_idx):
        z, weight = _c(z), _c(weight)
        out = torch.empty(z.shape[0], weight.shape[1], dtype=f32, device=weight.device)
        call("conan_embedding_fwd", ptr(z, i64), ptr(weight, f32), z.shape[0], weight.shape[1], weight.shape[0], ptr(out), stream_ptr())
        ctx.save_for_backward(z, weight)
        ctx.shape, ctx.padding_idx = weight.shape, padding_idx
        return out

    @staticmethod
    def backward(ctx, dout):
        z, weight = ctx.saved_tensors
        rows, H = ctx.shape
        if _pending is not None and rows <= 128 and lib().conan_wgrad_batchable(H, rows) and not any(j["weight_ptr"] == weight.data_ptr() for j in _pending):
            # inside a deferred backward pass (FlatGradients.backward): dW = onehot(z)^T dout is one more job of the batched node-level weight-gradient
            # launch (≈ 16 us at cfg2 for the one-hot build and its share of the batch, against 34 us for the two kernels of conan_embedding_bwd)
            dout = _c(dout)
            onehot = torch.empty(z.shape[0], rows, dtype=f32, device=dout.device)
            call("conan_onehot_rows", ptr(z), z.shape[0], rows, -1 if ctx.padding_idx is None else ctx.padding_idx, ptr(onehot), stream_ptr())
            dw, _ = _wgrad(onehot, dout, z.shape[0], H, rows, None, weight, False)
            return None, dw, None
        dw = torch.empty(ctx.shape, dtype=f32, device=dout.device)
        ws = torch.empty(int(lib().conan_embedding_bwd_ws(z.shape[0], ctx.shape[1], ctx.shape[0])), dtype=f32, device=dout.device)
        call("conan_embedding_bwd", ptr(z), ptr(_c(dout)), z.shape[0], ctx.shape[1], ctx.shape[0],
             -1 if ctx.padding_idx is None else ctx.padding_idx, ptr(dw), ptr(ws), stream_ptr())
        return None, dw, None


def embedding(z: Tensor, weight: Tensor, padding_idx: Optional[int] = 0) -> Tensor:
    return _EmbeddingFn.apply(z, weight, padding_idx)


# ------------------------------------------------------------------------------------------------ continuous filter pieces
def rbf_expand(graph: RadiusGraph, offset: Tensor, coeff: float) -> Tensor:
    """GaussianSmearing of every edge distance -> [max_edges, Gs] (rows >= E untouched).  No gradient (pos is an input)."""
    Gs = offset.shape[0]
    out = torch.empty(graph.max_edges, Gs, dtype=f32, device=offset.device)
    call("conan_rbf_fwd", ptr(graph.dist), ptr(graph.num_edges_dev), graph.max_edges, ptr(_c(offset), f32), Gs, float(coeff),
         ptr(out), stream_ptr())
    return out


class _CutoffScaleFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, w_raw, graph):
        out = torch.empty_like(w_raw)
        call("conan_cutoff_scale", ptr(graph.dist), ptr(graph.num_edges_dev), graph.max_edges, w_raw.shape[1], graph.cutoff,
             ptr(_c(w_raw)), ptr(out), stream_ptr())
        ctx.graph = graph
        return out

    @staticmethod
    def backward(ctx, dout):
        g = ctx.graph
        dout = _c(dout)
        din = torch.empty_like(dout)
        call("conan_cutoff_scale", ptr(g.dist), ptr(g.num_edges_dev), g.max_edges, dout.shape[1], g.cutoff, ptr(dout), ptr(din),
             stream_ptr())
        return din, None


def cutoff_scale(w_raw: Tensor, graph: RadiusGraph) -> Tensor:
    return _CutoffScaleFn.apply(w_raw, graph)


FUSED_FILTER_BACKWARD = True      # tools / tests switch the round-3 pair of kernels back on to compare (never read from the environment)


class _FilterFn(torch.autograd.Function):
    """Fused filter generator (conan_filter_fwd).  Its output must be consumed by `cfconv(..., pre_cutoff_grad=True)`:
    the incoming gradient g is then w.r.t. the un-scaled filter.  Backward: dw2 = g^T h1 (conan_linear_wgrad), then ONE pass
    (conan_filter_bwd) for gpre = (g w2) * ssp'(h1) and dw1 = gpre^T rbf, with gpre kept in registers and rbf regenerated."""

    @staticmethod
    def forward(ctx, graph, offset, coeff, w1, b1, w2, b2, use_pairs):
        F, Gs = w1.shape
        dev = w1.device
        need_grad = any(ctx.needs_input_grad[3:7])
        dist, cnt = (graph.pairs().pair_dist, graph.num_pairs_dev) if use_pairs else (graph.dist, graph.num_edges_dev)
        W = torch.empty(graph.max_edges, F, dtype=f32, device=dev)
        h1 = torch.empty(graph.max_edges, F, dtype=f32, device=dev) if need_grad else None
        call("conan_filter_fwd", ptr(dist), ptr(cnt), graph.max_edges, ptr(_c(offset), f32), Gs, float(coeff),
             graph.cutoff, F, ptr(_c(w1), f32), ptr(_c(b1), f32), ptr(_c(w2), f32), ptr(_c(b2), f32), ptr(W), ptr(h1), stream_ptr())
        ctx.graph, ctx.coeff, ctx.rows = graph, float(coeff), (dist, cnt)
        ctx.save_for_backward(offset, w1, w2, h1)
        return W

    @staticmethod
    def backward(ctx, dW):
        offset, w1, w2, h1 = ctx.saved_tensors
        g_ = ctx.graph
        dist, md = ctx.rows                          # per-edge or per-pair distances and their device-side count
        F, Gs = w1.shape
        ME = g_.max_edges
        dev = dW.device
        g = _c(dW)                                   # already multiplied by C(d): cfconv(..., pre_cutoff_grad=True)
        gmax = _take_gmax(dW, g) if F == 128 else None      # max |g|, when the producer tracked it and nothing touched g since: the two kernels below run on fp16 planes
        if gmax is not None and FUSED_FILTER_BACKWARD and lib().conan_filter_bwd2_supported(Gs, F):      # both layers' gradients in one pass over g and h1
            (dw1, db1), (dw2, db2) = _filter_bwd2(g, h1, dist, _c(offset), ctx.coeff, w1, _c(w2), ME, md, gmax)
            return None, None, None, dw1, db1, dw2, db2, None
        dw2, db2 = _wgrad(g, h1, ME, F, F, md, w2, True, gmax=gmax)
        if lib().conan_filter_bwd_supported(Gs, F):              # (g @ w2) * ssp'(h1) and its contraction with rbf(dist) in one pass
            dw1, db1 = _filter_bwd(g, h1, dist, _c(offset), ctx.coeff, w1, _c(w2), ME, md, gmax=gmax)
        else:
            dh1 = torch.empty_like(g)
            call("conan_linear_fwd", ptr(g), ptr(_c(w2)), None, ptr(h1), ME, F, F, 1, 2, ptr(md), ptr(dh1), stream_ptr())   # (g @ w2) * ssp'(h1)
            dw1, db1 = _wgrad(dh1, None, ME, Gs, F, md, w1, True, rbf=(dist, _c(offset), ctx.coeff))      # rbf(dist) regenerated inside the GEMM
        return None, None, None, dw1, db1, dw2, db2, None


def filter_fused_supported(num_gaussians: int, num_filters: int) -> bool:
    return bool(lib().conan_filter_fused_supported(int(num_gaussians), int(num_filters)))


def filter_generate(graph: "RadiusGraph", offset: Tensor, coeff: float, w1: Tensor, b1: Tensor, w2: Tensor, b2: Tensor,
                    use_pairs: bool = True) -> Tensor:
    """Filter rows mlp(rbf(d)) * C(d) -> [max_edges, F].  use_pairs=True: ONE row per undirected pair (d_ij = d_ji), to be
    consumed by `cfconv(..., pre_cutoff_grad=True, use_pairs=True)`; False: one row per directed edge."""
    return _FilterFn.apply(graph, offset, coeff, w1, b1, w2, b2, use_pairs)


def filter_cfconv_supported(num_gaussians: int, num_filters: int) -> bool:
    return bool(lib().conan_filter_cfconv_fwd_supported(int(num_gaussians), int(num_filters)))


def filter_cfconv(x: Tensor, graph: "RadiusGraph", offset: Tensor, coeff: float, w1: Tensor, b1: Tensor, w2: Tensor, b2: Tensor) -> Tensor:
    """CFConv's edge half in ONE launch, forward only (conan_filter_cfconv_fwd): out[i] = sum_{j in N(i)} x[j] * (mlp(rbf(d_ij)) * C(d_ij)) with the
    filter rows generated per directed edge and consumed on the spot — no [E, F] tensor exists.  Inference only: raises when a gradient is
    required (the training step's backward reads the filter tensor that `filter_generate` + `cfconv` save)."""
    if torch.is_grad_enabled() and any(t.requires_grad for t in (x, w1, b1, w2, b2)):
        raise RuntimeError("filter_cfconv is forward-only: use filter_generate + cfconv when gradients are needed")
    x = _c(x)
    out = torch.empty_like(x)
    call("conan_filter_cfconv_fwd", ptr(x, f32), ptr(graph.dist), ptr(graph.col), ptr(graph.tgt), ptr(graph.num_edges_dev), graph.max_edges,
         ptr(_c(offset), f32), offset.shape[0], float(coeff), graph.cutoff, w1.shape[0], ptr(_c(w1), f32), ptr(_c(b1), f32), ptr(_c(w2), f32),
         ptr(_c(b2), f32), graph.num_atoms, ptr(out), stream_ptr())
    return out


FUSED_CFCONV_BACKWARD = True      # dx and the pair gradient in one launch (conan_cfconv_bwd_xw_pairs); tools / tests switch it off to compare


class _CFConvFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, W, graph, pre_cutoff_grad=False, use_pairs=False):
        x, W = _c(x), _c(W)
        out = torch.empty_like(x)
        pid = graph.pairs().pid if use_pairs else None
        F = x.shape[1]
        # max |dW| of the pair gradient (F = 128): one device float per CFConv, raised with atomicMax by the backward kernel that writes the pair
        # gradient.  Something has to clear it before every backward pass without a fill launch of its own: the fused backward (dx + pair gradient in
        # one launch) has no kernel in front of it, so this forward launch does (zero_slot) — fresh on every forward / backward pair, captured or not.
        ctx.gmax = None
        if use_pairs and F == 128 and ctx.needs_input_grad[0] and ctx.needs_input_grad[1] and FUSED_CFCONV_BACKWARD and pre_cutoff_grad:
            ctx.gmax = torch.empty(1, dtype=f32, device=x.device)
        call("conan_cfconv_fwd", ptr(x, f32), ptr(W, f32), ptr(graph.rowptr), ptr(graph.col), ptr(pid), graph.num_atoms, F,
             ptr(out), ptr(ctx.gmax), stream_ptr())
        ctx.graph, ctx.pre, ctx.pairs = graph, pre_cutoff_grad, use_pairs
        ctx.save_for_backward(x, W)
        return out

    @staticmethod
    def backward(ctx, dout):
        x, W = ctx.saved_tensors
        g = ctx.graph
        dout = _c(dout)
        dx = dW = None
        F = x.shape[1]
        if ctx.gmax is not None and ctx.needs_input_grad[0] and ctx.needs_input_grad[1]:
            # dx and the pair gradient from ONE walk of the by-source CSR (the expression and the bits of conan_cfconv_bwd_w_pairs)
            t_rowptr, t_eid = g.transpose()
            dx = torch.empty_like(x)
            dW = torch.empty_like(W)
            call("conan_cfconv_bwd_xw_pairs", ptr(W), ptr(x), ptr(dout), ptr(t_rowptr), ptr(t_eid), ptr(g.tgt), ptr(g.pid), ptr(g.pair_e0), ptr(g.pair_e1),
                 ptr(g.pair_dist), float(g.cutoff), g.num_atoms, F, ptr(dx), ptr(dW), ptr(ctx.gmax), stream_ptr())
            _tag_gmax(dW, ctx.gmax)
            return dx, dW, None, None, None
        # max |dW| of the pair gradient (F = 128): one device float per CFConv backward, raised by conan_cfconv_bwd_w_pairs with atomicMax.  It is
        # cleared by the dx kernel that runs right before it on this stream (zero_slot) — fresh on every backward pass, no fill launch, and
        # nothing lives on the graph object (round 4 kept a zeroed pool there: stale once a graph object outlived one backward).
        want_gmax = ctx.needs_input_grad[1] and ctx.pairs and F == 128
        gmax = torch.empty(1, dtype=f32, device=x.device) if want_gmax else None
        if ctx.needs_input_grad[0]:
            t_rowptr, t_eid = g.transpose()
            dx = torch.empty_like(x)
            call("conan_cfconv_bwd_x", ptr(W), ptr(dout), ptr(t_rowptr), ptr(t_eid), ptr(g.tgt), ptr(g.pid) if ctx.pairs else None,
                 g.num_atoms, F, ptr(dx), ptr(gmax), stream_ptr())
        elif gmax is not None:
            gmax.zero_()
        if ctx.needs_input_grad[1]:
            dW = torch.empty_like(W)
            if ctx.pairs:
                if not ctx.pre:
                    raise RuntimeError("use_pairs requires pre_cutoff_grad=True (the pair gradient includes the cosine cutoff)")
                call("conan_cfconv_bwd_w_pairs", ptr(x), ptr(dout), ptr(g.num_pairs_dev), g.max_edges, ptr(g.pair_e0), ptr(g.pair_e1), ptr(g.col),
                     ptr(g.tgt), F, ptr(g.pair_dist), float(g.cutoff), ptr(dW), ptr(gmax), stream_ptr())
                if gmax is not None:
                    _tag_gmax(dW, gmax)
            else:
                call("conan_cfconv_bwd_w", ptr(x), ptr(dout), ptr(g.num_edges_dev), g.max_edges, ptr(g.col), ptr(g.tgt), F,
                     ptr(g.dist) if ctx.pre else None, float(g.cutoff or 0.0), ptr(dW), stream_ptr())
        return dx, dW, None, None, None


def cfconv(x: Tensor, W: Tensor, graph: RadiusGraph, pre_cutoff_grad: bool = False, use_pairs: bool = False) -> Tensor:
    """out[i] = sum_{j in N(i)} x[j] * W[(j->i)]   (CFConv.propagate).
    pre_cutoff_grad=True: the gradient returned for W is already multiplied by the cosine cutoff C(d_e), i.e. it is the
    gradient w.r.t. the un-scaled filter (used with `filter_generate`, whose backward then skips its own scaling pass)."""
    return _CFConvFn.apply(x, W, graph, pre_cutoff_grad, use_pairs)


class _SegmentSumFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, graph_ptr, num_graphs):
        x = _c(x)
        out = torch.empty(num_graphs, x.shape[1], dtype=f32, device=x.device)
        call("conan_segment_sum_fwd", ptr(x, f32), ptr(graph_ptr, i32), num_graphs, x.shape[1], ptr(out), stream_ptr())
        ctx.save_for_backward(graph_ptr)
        ctx.shape, ctx.G = x.shape, num_graphs
        return out

    @staticmethod
    def backward(ctx, dout):
        (gp,) = ctx.saved_tensors
        dx = torch.empty(ctx.shape, dtype=f32, device=dout.device)
        call("conan_segment_sum_bwd", ptr(_c(dout)), ptr(gp), ctx.G, ctx.shape[1], ptr(dx), stream_ptr())
        return dx, None, None


def segment_sum(x: Tensor, graph_ptr: Tensor, num_graphs: int) -> Tensor:
    """Sum readout per conformer graph (SumAggregation)."""
    return _SegmentSumFn.apply(x, graph_ptr, num_graphs)


# ------------------------------------------------------------------------------------------------ FGW barycenter
class _DensifyFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, feat, graph, N, shift, a, b, adjacency=True):
        feat = _c(feat)
        G, d = graph.num_graphs, feat.shape[1]
        dev = feat.device
        Ys = torch.empty(G, N, d, dtype=f32, device=dev)
        # adjacency=False: features only — the solver reads the structure from the graph's ragged lists (fgw_barycenter_batched(adjacency=graph))
        Cs = torch.empty(G, N, N, dtype=f32, device=dev) if adjacency else torch.empty(0, dtype=f32, device=dev)
        minmax = torch.empty(G, 2, dtype=f32, device=dev)
        call("conan_fgw_densify", ptr(feat, f32), ptr(graph.graph_ptr), ptr(graph.rowptr), ptr(graph.col), G, N, d, shift, a, b,
             ptr(Ys), ptr(Cs) if adjacency else None, ptr(minmax), stream_ptr())
        ctx.save_for_backward(feat, minmax)
        ctx.graph, ctx.args = graph, (N, shift, a, b)
        ctx.mark_non_differentiable(Cs)
        ctx.set_materialize_grads(False)          # no [G,N,N] zero tensor for the adjacency's "gradient" (a 5 us fill per step)
        return Ys, Cs

    @staticmethod
    def backward(ctx, dYs, _dCs):
        if dYs is None:
            return None, None, None, None, None, None, None
        feat, minmax = ctx.saved_tensors
        N, shift, a, b = ctx.args
        g = ctx.graph
        dfeat = torch.empty_like(feat)
        call("conan_fgw_densify_bwd", ptr(feat), ptr(_c(dYs)), ptr(g.graph_ptr), ptr(minmax), g.num_graphs, N, feat.shape[1],
             shift, a, b, ptr(dfeat), stream_ptr())
        return dfeat, None, None, None, None, None, None


def fgw_densify(feat: Tensor, graph: RadiusGraph, max_nodes: int, shift: float, a: float = 0.1, b: float = 2.0, adjacency: bool = True):
    """to_dense_batch + shift + normalize_tensor per conformer slab, and to_dense_adj (schnet_no_sum.py:242-252).  adjacency=False: the [G,N,N]
    adjacency tensors are not built (the second result is empty) — pass the graph itself to `fgw_barycenter_batched(adjacency=graph)`."""
    return _DensifyFn.apply(feat, graph, max_nodes, float(shift), float(a), float(b), bool(adjacency))


PROD_FGW = dict(alpha=0.1, epsilon=0.1, max_iter=5, tol=1e-2, inner_tol=1e-4, num_iter_max=5, stop_thr=1e-2,
                fixed_structure=False, fixed_features=False, warmstart=True)       # schnet_no_sum.py:281-306


class _FgwBarycenterFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, Ys, Cs, ps, p, lambdas, init_C, init_Y, params):
        adj = params.get("adjacency")             # a RadiusGraph: the input graphs' structure straight from its ragged lists, Cs unused
        Ys = _c(Ys)
        Cs = _c(Cs) if adj is None else None
        B, K, N, d = Ys.shape
        if adj is not None and adj.num_graphs != B * K:
            raise RuntimeError(f"adjacency graph holds {adj.num_graphs} conformer graphs, Ys holds {B} x {K}")
        dev = Ys.device
        prm = FgwParams(float(params["alpha"]), float(params["epsilon"]), int(params["max_iter"]), float(params["tol"]),
                        float(params["inner_tol"]), int(params["num_iter_max"]), float(params["stop_thr"]),
                        int(bool(params["fixed_structure"])), int(bool(params["fixed_features"])), int(bool(params["warmstart"])),
                        {"square_loss": 0, "kl_loss": 1}[params.get("loss_fun", "square_loss")], int(bool(params.get("cs_small_int", False))))
        Y = torch.empty(B, N, d, dtype=f32, device=dev)
        C = torch.empty(B, N, N, dtype=f32, device=dev)
        T = torch.empty(B, K, N, N, dtype=f32, device=dev)
        T_iter = torch.empty(prm.max_iter, B, K, N, N, dtype=f32, device=dev) if params.get("keep_iterates") else None
        info = torch.empty(B, 4, dtype=i32, device=dev)
        errs = torch.empty(B, 2, prm.max_iter, dtype=f32, device=dev)
        import ctypes
        if adj is None:
            ws = torch.empty(int(lib().conan_fgw_workspace_bytes(B, K, N, d)), dtype=torch.uint8, device=dev)
            call("conan_fgw_barycenter_fwd", ptr(Ys, f32), ptr(Cs, f32), ptr(ps), ptr(p), ptr(lambdas), ptr(init_C), ptr(init_Y),
                 B, K, N, d, ctypes.byref(prm), ptr(Y), ptr(C), ptr(T), ptr(T_iter), ptr(info), ptr(errs), ptr(ws), stream_ptr())
        else:
            ws = torch.empty(int(lib().conan_fgw_workspace_bytes_ragged(B, K, N, d)), dtype=torch.uint8, device=dev)
            call("conan_fgw_barycenter_fwd_ragged", ptr(Ys, f32), ptr(adj.graph_ptr, i32), ptr(adj.rowptr, i32), ptr(adj.col, i32), ptr(adj.tgt, i32),
                 ptr(ps), ptr(p), ptr(lambdas), ptr(init_C), ptr(init_Y), B, K, N, d, ctypes.byref(prm), ptr(Y), ptr(C), ptr(T), ptr(T_iter),
                 ptr(info), ptr(errs), ptr(ws), stream_ptr())
        ctx.save_for_backward(T, p, lambdas)
        ctx.dims = (B, K, N, d)
        ctx.set_materialize_grads(False)          # C, T, info, errs carry no gradient: without this autograd fills four zero tensors per backward
        if T_iter is None:
            ctx.mark_non_differentiable(C, T, info, errs)
            return Y, C, T, info, errs
        ctx.mark_non_differentiable(C, T, info, errs, T_iter)
        return Y, C, T, info, errs, T_iter

    @staticmethod
    def backward(ctx, dY, *_):
        if dY is None:
            return (None,) * 8
        T, p, lambdas = ctx.saved_tensors
        B, K, N, d = ctx.dims
        dYs = torch.empty(B, K, N, d, dtype=f32, device=dY.device)
        call("conan_fgw_barycenter_bwd", ptr(T), ptr(_c(dY)), ptr(p), ptr(lambdas), B, K, N, d, ptr(dYs), stream_ptr())
        return dYs, None, None, None, None, None, None, None


def fgw_barycenter_batched(Ys: Tensor, Cs: Tensor, ps: Optional[Tensor] = None, p: Optional[Tensor] = None,
                           lambdas: Optional[Tensor] = None, init_C: Optional[Tensor] = None, init_Y: Optional[Tensor] = None,
                           **params):
    """B independent FGW barycenters.  Ys [B,K,N,d], Cs [B,K,N,N] -> Y [B,N,d], C [B,N,N], T [B,K,N,N], info [B,4], errs [B,2,max_iter]
    (+ T_iter [max_iter,B,K,N,N] with keep_iterates=True: the couplings after every outer iteration, barycenter.py:196).
    Gradient flows to Ys only (through the final couplings as constants), like the reference.
    `adjacency=graph` (a RadiusGraph with B * K conformer graphs, Cs=None): the input structures are to_dense_adj of those graphs, read by the
    coupling kernels from the ragged neighbour lists — no [B,K,N,N] tensor exists (what the models do)."""
    prm = dict(PROD_FGW)
    prm.update(params)
    opt = lambda t: _c(t) if t is not None else None
    return _FgwBarycenterFn.apply(Ys, Cs, opt(ps), opt(p), opt(lambdas), opt(init_C), opt(init_Y), prm)


class _MseLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, target):
        pred, target = _c(pred), _c(target)
        if pred.shape != target.shape:
            raise RuntimeError(f"mse_loss: pred {tuple(pred.shape)} and target {tuple(target.shape)} must have the same shape")
        loss = torch.empty((), dtype=f32, device=pred.device)
        dpred = torch.empty_like(pred)
        call("conan_mse_loss_fwd", ptr(pred, f32), ptr(target, f32), pred.numel(), ptr(loss), ptr(dpred), stream_ptr())
        ctx.save_for_backward(dpred)
        return loss

    @staticmethod
    def backward(ctx, g):
        (dpred,) = ctx.saved_tensors
        if g.is_cuda and g.dtype == f32 and g.numel() == 1:        # the seed of loss.backward() (1, or 1 / world_size): one HIP launch, no aten kernel in the step
            out = torch.empty_like(dpred)
            call("conan_scale_scalar", ptr(dpred), ptr(_c(g)), dpred.numel(), ptr(out), stream_ptr())
            return out, None
        return dpred * g, None


def mse_loss(pred: Tensor, target: Tensor) -> Tensor:
    """mean((pred - target)^2) with its gradient formed in the same launch (the reference's nn.MSELoss criterion, common.py)."""
    return _MseLossFn.apply(pred, target)


class _ReadoutFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, Y, K, mode):
        Y = _c(Y)
        B, N, d = Y.shape
        out = torch.empty(B * K, d, dtype=f32, device=Y.device)
        call("conan_fgw_readout_fwd", ptr(Y, f32), B, K, N, d, mode, ptr(out), stream_ptr())
        ctx.save_for_backward(Y)
        ctx.K, ctx.mode = K, mode
        return out

    @staticmethod
    def backward(ctx, dout):
        (Y,) = ctx.saved_tensors
        B, N, d = Y.shape
        dY = torch.empty_like(Y)
        call("conan_fgw_readout_bwd", ptr(Y), ptr(_c(dout)), B, ctx.K, N, d, ctx.mode, ptr(dY), stream_ptr())
        return dY, None, None


def fgw_readout(Y: Tensor, num_conformers: int, mode: int = 0) -> Tensor:
    """[B,N,d] -> [B*K,d]: sum over barycenter nodes, repeated K times (schnet_no_sum.py:308-312); mode 1 = ViSNet variant."""
    return _ReadoutFn.apply(Y, num_conformers, mode)
