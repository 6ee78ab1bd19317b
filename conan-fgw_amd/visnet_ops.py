"""torch.autograd wrappers of the ViSNet kernels (csrc/visnet.hip, csrc/visnet_bwd.hip).  Plumbing only: every forward and
backward below is a HIP kernel of libconan_fgw_hip.so."""
from __future__ import annotations

import torch
from torch import Tensor

from . import ops
from ._lib import call, lib, ptr, stream_ptr

f32 = torch.float32
_c = ops._c


SUM_INPUT_GRADIENTS = True       # _MultiLinear.backward: the input gradients of the layers summed in one launch (False: a chain of launches)


def _new(like: Tensor, *shape):
    return torch.empty(*shape, dtype=f32, device=like.device)


def _tail0_shape(rows: int, width: int, device, m_dev):
    """Worst-case-sized edge buffer.  The rows beyond the device-side edge count are never read: every HIP consumer walks the CSR rows or
    takes the same device-side count, and the torch element-wise ops in between only carry the tail along (round 1 cleared it with a
    kernel per buffer: 93 launches = 1.6 ms of a BACE step)."""
    return ops.unread_rows(rows, width, device)


def _tail0(like: Tensor, m_dev):
    return _tail0_shape(like.shape[0], like[0].numel(), like.device, m_dev).view(like.shape)


class _Silu(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, m_dev):
        x = _c(x)
        y = _tail0(x, m_dev)
        call("conan_silu_fwd", ptr(x, f32), x.shape[0], x.shape[1], ptr(m_dev), ptr(y), stream_ptr())
        ctx.save_for_backward(x)
        ctx.m_dev = m_dev
        return y

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        dx = _tail0(x, ctx.m_dev)
        call("conan_silu_bwd", ptr(x), ptr(_c(dy)), x.shape[0], x.shape[1], ptr(ctx.m_dev), ptr(dx), stream_ptr())
        return dx, None


def silu(x, m_dev=None):
    return _Silu.apply(x, m_dev)


class _LinearSilu(torch.autograd.Function):
    """silu(x W^T + b) in one launch.  With gradients the kernel also stores the pre-activation (conan_linear_act_fwd); without,
    it is the plain fused act = 3 epilogue.  Backward: g = dy * silu'(pre) (conan_silu_bwd), then the usual dx / weight-gradient GEMMs."""

    @staticmethod
    def forward(ctx, x, w, b, m_dev):
        x, w = _c(x), _c(w)
        M, K = x.shape
        N = w.shape[0]
        need = any(ctx.needs_input_grad[:3])
        y = _tail0_shape(M, N, x.device, m_dev)
        if need:
            pre = torch.empty(M, N, dtype=f32, device=x.device)
            call("conan_linear_act_fwd", ptr(x, f32), ptr(w, f32), ptr(b), M, K, N, 3, ptr(m_dev), ptr(y), ptr(pre), stream_ptr())
            ctx.save_for_backward(x, w, pre)
        else:
            call("conan_linear_fwd", ptr(x, f32), ptr(w, f32), ptr(b), None, M, K, N, 0, 3, ptr(m_dev), ptr(y), stream_ptr())
        ctx.m_dev, ctx.has_b = m_dev, b is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, pre = ctx.saved_tensors
        md = ctx.m_dev
        M, K = x.shape
        N = w.shape[0]
        g = _tail0(pre, md)
        call("conan_silu_bwd", ptr(pre), ptr(_c(dy)), M, N, ptr(md), ptr(g), stream_ptr())
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = _tail0_shape(M, K, x.device, md)
            call("conan_linear_fwd", ptr(g), ptr(w), None, None, M, N, K, 1, 0, ptr(md), ptr(dx), stream_ptr())
        if ctx.needs_input_grad[1] or (ctx.has_b and ctx.needs_input_grad[2]):
            dw, db = ops._wgrad(g, x, M, K, N, md, w, ctx.has_b)         # immediate, or slabs now + one batched reduction per backward pass
        return dx, dw, db, None


class _MultiLinear(torch.autograd.Function):
    """Several Linear layers (optionally + SiLU) of the SAME input: y_i = act(x W_i^T + b_i).  Forward: one launch per layer as before.
    Backward: the input gradient sum_i g_i W_i is accumulated by the GEMMs themselves (each launch adds the running sum in its epilogue)
    instead of leaving len - 1 element-wise adds of [M,K] tensors to autograd — at edge level (dk / dv / f_proj of `f`) an add costs as much
    traffic as a GEMM."""

    @staticmethod
    def forward(ctx, x, m_dev, act, tap, *wb):
        x = _c(x)
        M, K = x.shape
        n = len(wb) // 2
        ws, bs = [_c(w) for w in wb[:n]], list(wb[n:])
        need = x.requires_grad or any(w.requires_grad for w in ws)
        ys, pres = [], []
        widths = {w.shape[0] for w in ws}
        if 2 <= n <= 4 and K == 128 and widths == {128} and (all(b is not None for b in bs) or all(b is None for b in bs)):
            # the layers' workgroups for the same rows run side by side in ONE launch: x is streamed once (conan_linear_multi_fwd)
            import ctypes
            ys = [_tail0_shape(M, 128, x.device, m_dev) for _ in range(n)]
            if act and need:
                pres = [torch.empty(M, 128, dtype=f32, device=x.device) for _ in range(n)]
            arr = lambda ts: (ctypes.c_void_p * n)(*[t.data_ptr() for t in ts])
            call("conan_linear_multi_fwd", ptr(x, f32), arr(ws), arr(bs) if bs[0] is not None else None, M, K, 128, n, 3 if act else 0, ptr(m_dev),
                 arr(ys), arr(pres) if pres else None, stream_ptr())
        else:
            for w, b in zip(ws, bs):
                N = w.shape[0]
                y = _tail0_shape(M, N, x.device, m_dev)
                if act and need:
                    pre = torch.empty(M, N, dtype=f32, device=x.device)
                    call("conan_linear_act_fwd", ptr(x, f32), ptr(w, f32), ptr(b), M, K, N, 3, ptr(m_dev), ptr(y), ptr(pre), stream_ptr())
                    pres.append(pre)
                else:
                    call("conan_linear_fwd", ptr(x, f32), ptr(w, f32), ptr(b), None, M, K, N, 0, 3 if act else 0, ptr(m_dev), ptr(y), stream_ptr())
                ys.append(y)
        ctx.save_for_backward(x, *ws, *pres)
        ctx.n, ctx.act, ctx.m_dev, ctx.has_b, ctx.tap = n, act, m_dev, [b is not None for b in bs], bool(tap)
        # tap: x handed through as one more output, for a consumer that uses the same x as a residual (edge_update's f): the gradient of
        # that use then arrives here and seeds the running sum of the input-gradient GEMMs instead of being added by autograd afterwards
        return tuple(ys) + ((x.view_as(x),) if tap else ())

    @staticmethod
    def backward(ctx, *dys):
        n, md = ctx.n, ctx.m_dev
        saved = ctx.saved_tensors
        x, ws, pres = saved[0], saved[1:1 + n], saved[1 + n:]
        M, K = x.shape
        seed = _c(dys[n]) if (ctx.tap and dys[n] is not None) else None      # gradient of the handed-through x: seeds the running sum
        dx = None
        dws, dbs, gs = [None] * n, [None] * n, [None] * n
        for i in range(n):
            if dys[i] is None:
                continue
            N = ws[i].shape[0]
            if ctx.act:
                gs[i] = _tail0(pres[i], md)
                call("conan_silu_bwd", ptr(pres[i]), ptr(_c(dys[i])), M, N, ptr(md), ptr(gs[i]), stream_ptr())
            else:
                gs[i] = _c(dys[i])
        live = [i for i in range(n) if gs[i] is not None]
        # dx = sum_i g_i W_i (+ seed) in ONE launch with the sum in the accumulators (conan_linear_sum_fwd) where the shapes allow; otherwise
        # a chain of GEMMs that carries the running sum through dx (each launch adds it in its epilogue)
        one_launch = (SUM_INPUT_GRADIENTS and ctx.needs_input_grad[0] and K == 128 and 2 <= len(live) <= 3
                      and all(ws[i].shape[0] == 128 for i in live))
        if one_launch:
            import ctypes
            m = len(live)
            dx = _tail0_shape(M, K, x.device, md)
            call("conan_linear_sum_fwd", (ctypes.c_void_p * m)(*[gs[i].data_ptr() for i in live]), (ctypes.c_int * m)(*([128] * m)),
                 (ctypes.c_void_p * m)(*[ws[i].data_ptr() for i in live]), m, 1, None, ptr(seed), M, K, ptr(md), ptr(dx), stream_ptr())
            seed = None
        for i in ([] if one_launch else live):
            w, N, g = ws[i], ws[i].shape[0], gs[i]
            if ctx.needs_input_grad[0]:
                if dx is None:
                    dx, res = _tail0_shape(M, K, x.device, md), seed
                    seed = None
                else:
                    res = dx
                if res is None or N <= 128 or (K == 128 and N in (256, 384)):      # (256 / 384-wide contractions into 128 outputs are ONE launch that takes a residual: k_linear_sum16)
                    call("conan_linear_fwd", ptr(g), ptr(w), None, ptr(res), M, N, K, 1, 0, ptr(md), ptr(dx), stream_ptr())      # dx = g W (+ running sum)
                else:           # a contraction wider than one 128-chunk accumulates in place over several launches: it cannot also read dx as its residual
                    tmp = _tail0_shape(M, K, x.device, md)
                    call("conan_linear_fwd", ptr(g), ptr(w), None, None, M, N, K, 1, 0, ptr(md), ptr(tmp), stream_ptr())
                    dx = res + tmp
        if len(live) > 1 and len({ws[i].shape[0] for i in live}) == 1:      # same input, same width: one batched slab launch (x streamed once)
            for i, (dw, db) in zip(live, ops._wgrad_shared_x([gs[i] for i in live], x, M, K, ws[live[0]].shape[0], md, [ws[i] for i in live],
                                                               [ctx.has_b[i] for i in live])):
                dws[i], dbs[i] = dw, db
        else:
            for i in live:
                dws[i], dbs[i] = ops._wgrad(gs[i], x, M, K, ws[i].shape[0], md, ws[i], ctx.has_b[i])
        if dx is None and seed is not None:
            dx = seed
        return (dx, None, None, None) + tuple(dws) + tuple(dbs)


def multi_lin(x: Tensor, mods, act_silu: bool = False, m_dev=None, tap: bool = False):
    """[act(Linear_i(x)) for Linear_i in mods] (+ [x] with tap=True) with the input gradients accumulated inside the backward GEMMs (see
    _MultiLinear)."""
    if all(x.shape[1] % 64 == 0 and m.weight.shape[0] % 64 == 0 for m in mods):
        return _MultiLinear.apply(x, m_dev, act_silu, tap, *[m.weight for m in mods], *[m.bias for m in mods])
    return tuple(lin(x, m, act_silu, m_dev) for m in mods) + ((x,) if tap else ())


def lin(x: Tensor, m: torch.nn.Linear, act_silu: bool = False, m_dev=None, grad_tail_unread: bool = False) -> Tensor:
    """`grad_tail_unread`: the caller's promise, per call site, that whatever consumes the INPUT gradient of this edge-level layer walks the CSR or
    takes the same m_dev (then its rows beyond m_dev are left as allocated instead of cleared: ViS_MP's s_proj, 22 us per layer); the default
    keeps them zero, which is what an arbitrary consumer (an element-wise torch op, a second use summed by autograd) needs."""
    if act_silu and x.shape[1] % 64 == 0 and m.weight.shape[0] % 64 == 0:
        return _LinearSilu.apply(x, m.weight, m.bias, m_dev)
    y = ops.linear(x, m.weight, m.bias, m_dev=m_dev, grad_tail_unread=grad_tail_unread and m_dev is not None)
    return silu(y, m_dev) if act_silu else y


class _NeighborScale(torch.autograd.Function):
    @staticmethod
    def forward(ctx, W, graph, cutoff):
        W = _c(W)
        out = torch.empty_like(W)                                   # rows beyond the device-side edge count are never read (see _tail0_shape)
        call("conan_visnet_neighbor_scale_to", ptr(W), ptr(graph.dist), ptr(graph.col), ptr(graph.tgt), ptr(graph.num_edges_dev),
             graph.max_edges, W.shape[1], float(cutoff), ptr(out), stream_ptr())
        ctx.graph, ctx.cutoff = graph, float(cutoff)
        return out

    @staticmethod
    def backward(ctx, dout):
        g = ctx.graph
        dout = _c(dout)
        d = torch.empty_like(dout)
        call("conan_visnet_neighbor_scale_to", ptr(dout), ptr(g.dist), ptr(g.col), ptr(g.tgt), ptr(g.num_edges_dev), g.max_edges, dout.shape[1],
             ctx.cutoff, ptr(d), stream_ptr())
        return d, None, None


def neighbor_scale(W, graph, cutoff):
    return _NeighborScale.apply(W, graph, cutoff)


class _Concat2(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        a, b = _c(a), _c(b)
        out = _new(a, a.shape[0], a.shape[1] + b.shape[1])
        call("conan_concat2", ptr(a, f32), a.shape[1], ptr(b, f32), b.shape[1], a.shape[0], ptr(out), stream_ptr())
        ctx.dims = (a.shape[0], a.shape[1], b.shape[1])
        return out

    @staticmethod
    def backward(ctx, dout):
        n, Ha, Hb = ctx.dims
        da, db = _new(dout, n, Ha), _new(dout, n, Hb)
        call("conan_split2", ptr(_c(dout)), Ha, Hb, n, ptr(da), ptr(db), stream_ptr())
        return da, db


def concat2(a, b):
    return _Concat2.apply(a, b)


class _EdgeEmbed(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, p, graph):
        x, p = _c(x), _c(p)
        H = x.shape[1]
        f = _tail0_shape(graph.max_edges, H, x.device, graph.num_edges_dev)
        call("conan_visnet_edge_embed", ptr(x, f32), ptr(p, f32), ptr(graph.col), ptr(graph.tgt), ptr(graph.num_edges_dev), graph.max_edges, H,
             ptr(f), stream_ptr())
        ctx.save_for_backward(x, p)
        ctx.graph = graph
        return f

    @staticmethod
    def backward(ctx, df):
        x, p = ctx.saved_tensors
        g = ctx.graph
        tr, te = g.transpose()
        dp, dx = _tail0(p, g.num_edges_dev), torch.empty_like(x)
        call("conan_visnet_edge_embed_bwd", ptr(x), ptr(p), ptr(_c(df)), ptr(g.rowptr), ptr(g.col), ptr(g.tgt), ptr(tr), ptr(te),
             ptr(g.num_edges_dev), g.max_edges, x.shape[0], x.shape[1], ptr(dp), ptr(dx), stream_ptr())
        return dx, dp, None


def edge_embed(x, p, graph):
    return _EdgeEmbed.apply(x, p, graph)


class _LayerNorm(torch.autograd.Function):
    """tap=True also hands x through (second output) for a residual use of the same tensor: its gradient is added to dx by the backward kernel."""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps, tap):
        x = _c(x)
        out = torch.empty_like(x)
        call("conan_layernorm_fwd", ptr(x, f32), ptr(_c(gamma)), ptr(_c(beta)), x.shape[0], x.shape[1], float(eps), ptr(out), stream_ptr())
        ctx.save_for_backward(x, gamma)
        ctx.eps = float(eps)
        return (out, x.view_as(x)) if tap else out

    @staticmethod
    def backward(ctx, dy, dtap=None):
        x, gamma = ctx.saved_tensors
        if dy is None:
            return dtap, None, None, None, None
        n, H = x.shape
        dx, dg, db = torch.empty_like(x), torch.empty_like(gamma), torch.empty_like(gamma)
        ws = torch.empty(int(lib().conan_layernorm_bwd_ws(n, H)), dtype=f32, device=x.device)
        if dtap is not None:
            call("conan_layernorm_bwd_res", ptr(x), ptr(_c(gamma)), ptr(_c(dy)), ptr(_c(dtap)), n, H, ctx.eps, ptr(dx), ptr(dg), ptr(db), ptr(ws), stream_ptr())
        else:
            call("conan_layernorm_bwd", ptr(x), ptr(_c(gamma)), ptr(_c(dy)), n, H, ctx.eps, ptr(dx), ptr(dg), ptr(db), ptr(ws), stream_ptr())
        return dx, dg, db, None, None


def layernorm(x, m: torch.nn.LayerNorm, tap: bool = False):
    return _LayerNorm.apply(x, m.weight, m.bias, m.eps, bool(tap))


class _ScaleChannels(torch.autograd.Function):
    """v * w[channel].  tap=True also hands v through (a second output) for a residual use of the same tensor: that use's gradient then arrives here and
    is added by the backward kernel itself (conan_scale_channels_add) instead of by autograd in a launch of its own."""

    @staticmethod
    def forward(ctx, v, w, tap):
        v = _c(v)
        out = torch.empty_like(v)
        H = v.shape[-1]
        call("conan_scale_channels", ptr(v, f32), ptr(w, f32), v.numel() // H, H, ptr(out), stream_ptr())
        ctx.save_for_backward(w)
        return (out, v.view_as(v)) if tap else out

    @staticmethod
    def backward(ctx, dout, dtap=None):
        (w,) = ctx.saved_tensors
        if dout is None:
            return dtap, None, None
        dout = _c(dout)
        dv = torch.empty_like(dout)
        H = dout.shape[-1]
        if dtap is not None:
            call("conan_scale_channels_add", ptr(dout), ptr(w), ptr(_c(dtap)), dout.numel() // H, H, ptr(dv), stream_ptr())
        else:
            call("conan_scale_channels", ptr(dout), ptr(w), dout.numel() // H, H, ptr(dv), stream_ptr())
        return dv, None, None


def scale_channels(v, w, tap=False):
    return _ScaleChannels.apply(v, w, bool(tap))


class _VecDot(torch.autograd.Function):
    @staticmethod
    def forward(ctx, vp, n, H):
        vp = _c(vp)
        out = _new(vp, n, H)
        call("conan_visnet_vecdot", ptr(vp, f32), n, H, ptr(out), stream_ptr())
        ctx.save_for_backward(vp)
        ctx.dims = (n, H)
        return out

    @staticmethod
    def backward(ctx, dout):
        (vp,) = ctx.saved_tensors
        n, H = ctx.dims
        dvp = torch.empty_like(vp)
        call("conan_visnet_vecdot_bwd", ptr(vp), ptr(_c(dout)), n, H, ptr(dvp), stream_ptr())
        return dvp, None, None


def vecdot(vp, n, H):
    return _VecDot.apply(vp, n, H)


class _AttnMessage(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q, k, v, dk, dv, graph, cutoff, heads, pre_act):
        q, k, v, dk, dv = (_c(t) for t in (q, k, v, dk, dv))
        n, H = q.shape
        vmsg = _tail0_shape(graph.max_edges, H, q.device, graph.num_edges_dev)
        xagg = _new(q, n, H)
        call("conan_visnet_attn_message", ptr(q, f32), ptr(k), ptr(v), ptr(dk), ptr(dv), ptr(graph.rowptr), ptr(graph.col), ptr(graph.dist),
             float(cutoff), n, H, heads, int(pre_act), ptr(vmsg), ptr(xagg), stream_ptr())
        ctx.save_for_backward(q, k, v, dk, dv)
        ctx.graph, ctx.cutoff, ctx.heads, ctx.pre = graph, float(cutoff), heads, int(pre_act)
        return vmsg, xagg

    @staticmethod
    def backward(ctx, dvmsg, dxagg):
        q, k, v, dk, dv = ctx.saved_tensors
        g = ctx.graph
        tr, te = g.transpose()
        n, H = q.shape
        dq, dkn, dvn = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        ddk, ddv = _tail0(dk, g.num_edges_dev), _tail0(dv, g.num_edges_dev)
        call("conan_visnet_attn_message_bwd", ptr(q), ptr(k), ptr(v), ptr(dk), ptr(dv), ptr(_c(dvmsg)), ptr(_c(dxagg)), ptr(g.rowptr), ptr(g.col),
             ptr(g.tgt), ptr(tr), ptr(te), ptr(g.dist), ctx.cutoff, n, H, ctx.heads, ctx.pre, ptr(dq), ptr(dkn), ptr(dvn), ptr(ddk), ptr(ddv),
             stream_ptr())
        return dq, dkn, dvn, ddk, ddv, None, None, None, None


def attn_message(q, k, v, dk, dv, graph, cutoff, heads, pre_act=False):
    """pre_act=True: dk / dv are the projections' pre-activations; SiLU is applied inside the kernels as they are loaded (and the gradients
    returned for them are w.r.t. the pre-activations), so the activated [E,H] tensors and the stand-alone SiLU backward passes never exist."""
    return _AttnMessage.apply(q, k, v, dk, dv, graph, cutoff, heads, pre_act)


class _VecAggregate(torch.autograd.Function):
    @staticmethod
    def forward(ctx, vec, s, dvec3, graph, pre_act):
        vec, s = _c(vec), _c(s)
        n, _, H = vec.shape
        vagg = torch.empty_like(vec)
        call("conan_visnet_vec_aggregate", ptr(vec, f32), ptr(s, f32), ptr(dvec3), ptr(graph.rowptr), ptr(graph.col), n, H, int(pre_act), ptr(vagg),
             stream_ptr())
        ctx.save_for_backward(vec, s, dvec3)
        ctx.graph, ctx.pre = graph, int(pre_act)
        return vagg

    @staticmethod
    def backward(ctx, dvagg):
        vec, s, dvec3 = ctx.saved_tensors
        g = ctx.graph
        tr, te = g.transpose()
        n, _, H = vec.shape
        ds, dvec = _tail0(s, g.num_edges_dev), torch.empty_like(vec)
        call("conan_visnet_vec_aggregate_bwd", ptr(vec), ptr(s), ptr(dvec3), ptr(_c(dvagg)), ptr(g.col), ptr(g.tgt), ptr(tr), ptr(te),
             ptr(g.num_edges_dev), g.max_edges, n, H, ctx.pre, ptr(ds), ptr(dvec), stream_ptr())
        return dvec, ds, None, None, None


def vec_aggregate(vec, s, dvec3, graph, pre_act=False):
    """pre_act=True: s is s_proj's pre-activation (SiLU applied on load, gradient returned w.r.t. it; see attn_message)."""
    return _VecAggregate.apply(vec, s, dvec3, graph, pre_act)


class _NodeUpdate(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, vec, vdot, o, vp, vagg, vdot_of_vp):
        x, vec, vdot, o, vp, vagg = (_c(t) for t in (x, vec, vdot, o, vp, vagg))
        n, H = x.shape
        xo, veco = torch.empty_like(x), torch.empty_like(vec)
        call("conan_visnet_node_update", ptr(x, f32), ptr(vec), ptr(vdot), ptr(o), ptr(vp), ptr(vagg), n, H, ptr(xo), ptr(veco), stream_ptr())
        ctx.save_for_backward(vdot, o, vp)
        ctx.vdot_of_vp = bool(vdot_of_vp)
        return xo, veco

    @staticmethod
    def backward(ctx, dxo, dveco):
        vdot, o, vp = ctx.saved_tensors
        n, H = vdot.shape
        dxo, dveco = _c(dxo), _c(dveco)
        do, dvp = torch.empty_like(o), torch.empty_like(vp)
        dvdot = None if ctx.vdot_of_vp else torch.empty_like(vdot)     # vdot = vecdot_detached(vp): its gradient is folded into dvp by the kernel
        call("conan_visnet_node_update_bwd", ptr(dxo), ptr(dveco), ptr(vdot), ptr(o), ptr(vp), n, H, ptr(dvdot), ptr(do), ptr(dvp), stream_ptr())
        return dxo, dveco, dvdot, do, dvp, dveco, None


def vecdot_detached(vp, n, H):
    """vecdot(vp) outside autograd — for node_update(..., vdot_of_vp=True), whose backward carries the gradient through it."""
    out = _new(vp, n, H)
    call("conan_visnet_vecdot", ptr(_c(vp).detach(), f32), n, H, ptr(out), stream_ptr())
    return out


def node_update(x, vec, vdot, o, vp, vagg, vdot_of_vp=False):
    """vdot_of_vp=True: vdot is vecdot_detached(vp) — the kernel's backward then also writes the vec1 / vec2 columns of dvp (the gradient through vdot)
    instead of leaving a second [3n,3H] tensor to a vecdot backward and their sum to autograd."""
    return _NodeUpdate.apply(x, vec, vdot, o, vp, vagg, vdot_of_vp)


class _EdgeUpdate(torch.autograd.Function):
    @staticmethod
    def forward(ctx, wt, ws, t, dvec3, f, graph, pre_act):
        wt, ws, t, f = (_c(a) for a in (wt, ws, t, f))
        H = f.shape[1]
        fo = _tail0(f, graph.num_edges_dev)
        call("conan_visnet_edge_update", ptr(wt, f32), ptr(ws), ptr(t), ptr(dvec3), ptr(graph.col), ptr(graph.tgt), ptr(graph.num_edges_dev),
             graph.max_edges, H, int(pre_act), ptr(f), ptr(fo), stream_ptr())
        ctx.save_for_backward(wt, ws, t, dvec3)
        ctx.graph, ctx.pre = graph, int(pre_act)
        return fo

    @staticmethod
    def backward(ctx, dfo):
        wt, ws, t, dvec3 = ctx.saved_tensors
        g = ctx.graph
        tr, te = g.transpose()
        n, H = wt.shape[0] // 3, t.shape[1]
        dfo = _c(dfo)
        dwt, dws, dt = torch.empty_like(wt), torch.empty_like(ws), _tail0(t, g.num_edges_dev)
        call("conan_visnet_edge_update_bwd", ptr(wt), ptr(ws), ptr(t), ptr(dvec3), ptr(dfo), ptr(g.rowptr), ptr(g.col), ptr(g.tgt), ptr(tr), ptr(te),
             n, H, ctx.pre, ptr(dwt), ptr(dws), ptr(dt), stream_ptr())
        return dwt, dws, dt, None, dfo, None, None


def edge_update(wt, ws, t, dvec3, f, graph, pre_act=False):
    """pre_act=True: t is f_proj's pre-activation (SiLU applied on load, gradient returned w.r.t. it; see attn_message)."""
    return _EdgeUpdate.apply(wt, ws, t, dvec3, f, graph, pre_act)


class _SpatialNorm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, v, n, H):
        v = _c(v)
        out = _new(v, n, H)
        call("conan_visnet_spatial_norm", ptr(v, f32), n, H, ptr(out), stream_ptr())
        ctx.save_for_backward(v)
        ctx.dims = (n, H)
        return out

    @staticmethod
    def backward(ctx, dout):
        (v,) = ctx.saved_tensors
        n, H = ctx.dims
        dv = torch.empty_like(v)
        call("conan_visnet_spatial_norm_bwd", ptr(v), ptr(_c(dout)), n, H, ptr(dv), stream_ptr())
        return dv, None, None


def spatial_norm(v, n, H):
    return _SpatialNorm.apply(v, n, H)


class _Gate(torch.autograd.Function):
    @staticmethod
    def forward(ctx, u, v2, n, O, act):
        u, v2 = _c(u), _c(v2)
        xo, vo = _new(u, n, O), _new(u, n, 3, O)
        call("conan_visnet_gate", ptr(u, f32), ptr(v2, f32), n, O, act, ptr(xo), ptr(vo), stream_ptr())
        ctx.save_for_backward(u, v2)
        ctx.dims = (n, O, act)
        return xo, vo

    @staticmethod
    def backward(ctx, dxo, dvo):
        u, v2 = ctx.saved_tensors
        n, O, act = ctx.dims
        du, dv2 = torch.empty_like(u), torch.empty_like(v2)
        call("conan_visnet_gate_bwd", ptr(u), ptr(v2), ptr(_c(dxo)), ptr(_c(dvo)), n, O, act, ptr(du), ptr(dv2), stream_ptr())
        return du, dv2, None, None, None


def gate(u, v2, n, O, act):
    return _Gate.apply(u, v2, n, O, act)


class _Prior(torch.autograd.Function):
    """out = x * std + atomref[z]   (std is a buffer: no gradient)."""

    @staticmethod
    def forward(ctx, x, z, atomref_w, std):
        x = _c(x)
        out = torch.empty_like(x)
        call("conan_visnet_prior", ptr(x, f32), ptr(_c(z), torch.int64), ptr(_c(atomref_w)), ptr(std), x.shape[0], x.shape[1], ptr(out), stream_ptr())
        ctx.save_for_backward(z, std)
        ctx.rows = atomref_w.shape[0]
        return out

    @staticmethod
    def backward(ctx, dout):
        z, std = ctx.saved_tensors
        dout = _c(dout)
        n, O = dout.shape
        dx = torch.empty_like(dout)
        call("conan_scale_scalar", ptr(dout), ptr(std), dout.numel(), ptr(dx), stream_ptr())
        rs = _new(dout, n, 1)
        call("conan_rowsum", ptr(dout), n, O, ptr(rs), stream_ptr())
        dw = _new(dout, ctx.rows, 1)
        ws = torch.empty(int(lib().conan_embedding_bwd_ws(n, 1, ctx.rows)), dtype=f32, device=dout.device)
        call("conan_embedding_bwd", ptr(z), ptr(rs), n, 1, ctx.rows, -1, ptr(dw), ptr(ws), stream_ptr())
        return dx, None, dw, None


def prior(x, z, atomref_w, std):
    return _Prior.apply(x, z, atomref_w, std)
