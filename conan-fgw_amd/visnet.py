"""Drop-in ConAN `ViSNet` backbone for MI355X.

Mirrors the reference's wrapper class (conan_fgw/src/model/graph_embeddings/visnet.py:82-288) and the vendored ViSNet it
subclasses (torch_geometric_visnet.py:1061-1229) with the only configuration ConAN instantiates (common.py:542-546:
lmax=1, 8 heads, 6 layers, 32 non-trainable RBFs, cutoff 5 A, vertex=False, vecnorm_type=None): same constructor
`ViSNet(device, hidden_channels, cutoff=5.0)`, same methods, same parameter / buffer names, so the reference's
`state_dict` loads with strict=True.  The nn modules are parameter containers; compute runs on libconan_fgw_hip.so.

Forward and backward run on HIP kernels (csrc/visnet.hip, csrc/visnet_bwd.hip) through the autograd wrappers of
visnet_ops.py; gradients reach every trainable parameter exactly as in the reference (the RBF means/betas and the
VecLayerNorm weights are non-trainable buffers there too).  No CPU fallback.
"""
from __future__ import annotations

import math
from typing import Optional

import torch
from torch import Tensor
from torch.nn import Embedding, LayerNorm, Linear

from . import ops
from . import visnet_ops as vo
from ._lib import call, ptr, stream_ptr

f32 = torch.float32


class ExpNormalSmearing(torch.nn.Module):
    def __init__(self, cutoff: float = 5.0, num_rbf: int = 32):
        super().__init__()
        self.cutoff, self.num_rbf, self.alpha = cutoff, num_rbf, 5.0 / cutoff
        start = torch.exp(torch.tensor(-cutoff))
        self.register_buffer("means", torch.linspace(start, 1, num_rbf))
        self.register_buffer("betas", torch.tensor([(2 / num_rbf * (1 - start)) ** -2] * num_rbf))


class VecLayerNorm(torch.nn.Module):
    def __init__(self, hidden: int):
        super().__init__()
        self.register_buffer("weight", torch.ones(hidden))


class NeighborEmbedding(torch.nn.Module):
    def __init__(self, hidden: int, num_rbf: int, cutoff: float, max_z: int = 100):
        super().__init__()
        self.embedding = Embedding(max_z, hidden)
        self.distance_proj = Linear(num_rbf, hidden)
        self.combine = Linear(hidden * 2, hidden)
        torch.nn.init.xavier_uniform_(self.distance_proj.weight); self.distance_proj.bias.data.zero_()
        torch.nn.init.xavier_uniform_(self.combine.weight); self.combine.bias.data.zero_()


class EdgeEmbedding(torch.nn.Module):
    def __init__(self, num_rbf: int, hidden: int):
        super().__init__()
        self.edge_proj = Linear(num_rbf, hidden)
        torch.nn.init.xavier_uniform_(self.edge_proj.weight); self.edge_proj.bias.data.zero_()


NODE_LEVEL_ON_SIDE_STREAM = 4         # 3 = also the 3-D output head beside the barycenter head; ViS_MP: 1 = the atom-level head of a layer beside the edge-level projections of f; 2 = also o_proj / node_update beside
                                      # s_proj + vec_aggregate / edge_update; 4 = a layer's atom-level head no longer waits for the previous layer's edge update (its inputs come
                                      # from node_update on the second stream itself); 0 = one stream (tools/ab_step_switch.py compares)


TAP_X_RESIDUAL = True                 # x handed through the LayerNorm's autograd node to node_update's residual (conan_layernorm_bwd_res adds the residual's gradient)
TAP_VEC_RESIDUAL = True               # vec handed through VecLayerNorm's autograd node to node_update's residual: the two gradients of vec are summed by the scaling's backward kernel
TAP_VEC_INTO_PROJECTIONS = True       # vl handed through vec_proj / w_trg / w_src's autograd node to the vector aggregation (its gradient seeds their input-gradient sum); False: autograd adds
FOLD_VECDOT_BACKWARD = True           # vec_dot's backward inside node_update's (one [3n,3H] gradient instead of two and their autograd sum); False: separate nodes (tools/ab_step_switch.py)


class ViS_MP(torch.nn.Module):
    def __init__(self, num_heads: int, hidden: int, cutoff: float, last_layer: bool = False):
        super().__init__()
        self.num_heads, self.hidden_channels, self.last_layer, self.cutoff = num_heads, hidden, last_layer, cutoff
        self.layernorm = LayerNorm(hidden)
        self.vec_layernorm = VecLayerNorm(hidden)
        self.vec_proj = Linear(hidden, hidden * 3, False)
        self.q_proj, self.k_proj, self.v_proj = Linear(hidden, hidden), Linear(hidden, hidden), Linear(hidden, hidden)
        self.dk_proj, self.dv_proj = Linear(hidden, hidden), Linear(hidden, hidden)
        self.s_proj = Linear(hidden, hidden * 2)
        if not last_layer:
            self.f_proj = Linear(hidden, hidden)
            self.w_src_proj = Linear(hidden, hidden, False)
            self.w_trg_proj = Linear(hidden, hidden, False)
        self.o_proj = Linear(hidden, hidden * 3)
        for m in self.modules():
            if isinstance(m, Linear):
                torch.nn.init.xavier_uniform_(m.weight)
                if m.bias is not None:
                    m.bias.data.zero_()


class ViSNetBlock(torch.nn.Module):
    def __init__(self, num_heads=8, num_layers=6, hidden_channels=128, num_rbf=32, max_z=100, cutoff=5.0, max_num_neighbors=32):
        super().__init__()
        self.num_heads, self.hidden_channels, self.cutoff, self.max_num_neighbors = num_heads, hidden_channels, cutoff, max_num_neighbors
        self.embedding = Embedding(max_z, hidden_channels)
        self.distance_expansion = ExpNormalSmearing(cutoff, num_rbf)
        self.neighbor_embedding = NeighborEmbedding(hidden_channels, num_rbf, cutoff, max_z)
        self.edge_embedding = EdgeEmbedding(num_rbf, hidden_channels)
        self.vis_mp_layers = torch.nn.ModuleList([ViS_MP(num_heads, hidden_channels, cutoff, last_layer=(i == num_layers - 1))
                                                  for i in range(num_layers)])
        self.out_norm = LayerNorm(hidden_channels)
        self.vec_out_norm = VecLayerNorm(hidden_channels)

    def forward(self, z: Tensor, pos: Tensor, graph_ptr: Tensor, num_graphs: int):
        """torch_geometric_visnet.py:843-886."""
        H, dev, n, s = self.hidden_channels, z.device, z.shape[0], stream_ptr()
        g = ops.RadiusGraph(pos, graph_ptr, num_graphs, self.cutoff, self.max_num_neighbors, loop=True)      # Distance, :331-347
        md, ME = g.num_edges_dev, g.max_edges
        dvec = torch.empty(ME, 3, dtype=f32, device=dev)                                                     # geometry: no gradient (pos is an input)
        call("conan_visnet_edge_unit", ptr(pos.contiguous(), f32), ptr(g.col), ptr(g.tgt), ptr(md), ME, ptr(dvec), s)
        de = self.distance_expansion
        rbf = ops.empty_rows(ME, de.num_rbf, dev, md)
        call("conan_visnet_expnormal", ptr(g.dist), ptr(md), ME, ptr(de.means), ptr(de.betas), de.num_rbf, de.alpha, de.cutoff, ptr(rbf), s)
        x = ops.embedding(z, self.embedding.weight, None)
        # NeighborEmbedding, :387-420
        ne = self.neighbor_embedding
        W = vo.neighbor_scale(vo.lin(rbf, ne.distance_proj, False, md), g, self.cutoff)
        xn = ops.cfconv(ops.embedding(z, ne.embedding.weight, None), W, g)
        x = vo.lin(vo.concat2(x, xn), ne.combine)
        vec = torch.zeros(n, 3, H, dtype=f32, device=dev)                                                    # :868-870
        f = vo.edge_embed(x, vo.lin(rbf, self.edge_embedding.edge_proj, False, md), g)                       # EdgeEmbedding, :463-465
        on_side = False                      # x, vec were produced on the second stream (by the previous layer's node_update)
        for layer in self.vis_mp_layers:
            x, vec, f, on_side = self._vis_mp(layer, x, vec, f, g, dvec, on_side)
        return vo.layernorm(x, self.out_norm), vo.scale_channels(vec, self.vec_out_norm.weight)

    def _vis_mp(self, L: ViS_MP, x, vec, f, g, dvec, inputs_on_side: bool = False):
        """ViS_MP.forward / message / aggregate / edge_update, torch_geometric_visnet.py:579-673."""
        H, n = self.hidden_channels, x.shape[0]
        md = g.num_edges_dev
        wt = ws = None
        x_res = x
        vec_res = vec                        # the tensors node_update takes as its residuals (handed through VecLayerNorm's node when TAP_VEC_RESIDUAL)

        def node_level():
            """Everything of the layer's head that lives on the atoms (a chain of small launches: LayerNorm, the q / k / v and vec / w_trg / w_src
            projections, vec_dot) — independent of the edge-level projections of f below."""
            nonlocal wt, ws, vec_res, x_res
            if TAP_X_RESIDUAL:                                                     # x is also node_update's residual: its gradient joins the LayerNorm's in one pass
                xl_, x_res = vo.layernorm(x, L.layernorm, tap=True)
            else:
                xl_ = vo.layernorm(x, L.layernorm)
            if TAP_VEC_RESIDUAL:                                                   # vec is also node_update's residual: its gradient joins VecLayerNorm's in one pass
                vl_, vec_res = vo.scale_channels(vec, L.vec_layernorm.weight, tap=True)
            else:
                vl_ = vo.scale_channels(vec, L.vec_layernorm.weight)
            q_, k_, v_ = vo.multi_lin(xl_, [L.q_proj, L.k_proj, L.v_proj])
            mods = [L.vec_proj] if L.last_layer else [L.vec_proj, L.w_trg_proj, L.w_src_proj]      # they read the same vl: one autograd node
            if TAP_VEC_INTO_PROJECTIONS:
                # vl is also what the vector aggregation gathers: handed through the projections' node, so that the aggregation's gradient seeds the
                # running sum of their input-gradient GEMMs instead of being added to it by autograd afterwards (a [3n,H] element-wise add per layer)
                outs = vo.multi_lin(vl_.view(3 * n, H), mods, tap=True)
                vl_ = outs[-1].view(n, 3, H)
                outs = outs[:-1]
            else:
                outs = vo.multi_lin(vl_.view(3 * n, H), mods) if not L.last_layer else (vo.lin(vl_.view(3 * n, H), L.vec_proj),)
            vp_ = outs[0]                                                          # [3n, 3H] = [vec1|vec2|vec3]
            if not L.last_layer:
                wt, ws = outs[1], outs[2]
            return vl_, q_, k_, v_, vp_, (vo.vecdot_detached(vp_, n, H) if FOLD_VECDOT_BACKWARD else vo.vecdot(vp_, n, H))

        side = None
        if NODE_LEVEL_ON_SIDE_STREAM:
            # The atom-level chain (~100 us of small launches at BACE B = 64) runs on a second HIP stream under the edge-level projection of f
            # (218 us, fills the chip); autograd replays each op's backward on the stream of its forward, so the backward overlaps the same way.
            main = torch.cuda.current_stream()
            side = getattr(self, "_node_stream", None)
            if side is None or side.device != main.device:
                side = torch.cuda.Stream(device=main.device)
                object.__setattr__(self, "_node_stream", side)                     # (not module state)
            # (level 4) x and vec come from node_update on the second stream: the chain then starts under the previous layer's edge update (a
            # gather kernel of small workgroups that shares the CUs) instead of queueing behind it — and then behind this layer's projection of
            # f, a persistent kernel with one 137 KB workgroup per CU beside which nothing else becomes resident: the timeline had the chain
            # run AFTER it (layernorm "228 us"), on the critical path of the forward pass
            if not (inputs_on_side and NODE_LEVEL_ON_SIDE_STREAM >= 4):
                side.wait_stream(main)
            with torch.cuda.stream(side):
                vl, q, k, v, vp, vdot = node_level()
        else:
            vl, q, k, v, vp, vdot = node_level()
        # dk / dv / f_proj read the same f: one autograd node, so that the three input gradients are summed inside the backward GEMMs
        # dk / dv stay PRE-activations: act (SiLU) is applied inside the attention kernels as they load them
        if L.last_layer:
            dk, dv = vo.multi_lin(f, [L.dk_proj, L.dv_proj], False, md)
        else:
            # (t too: act applied inside edge_update; f itself is handed through for edge_update's residual, see _MultiLinear)
            dk, dv, t, f = vo.multi_lin(f, [L.dk_proj, L.dv_proj, L.f_proj], False, md, tap=True)
        if side is not None:
            main.wait_stream(side)
            for tt in (vl, q, k, v, vp, vdot, wt, ws):
                if tt is not None:
                    tt.record_stream(main)
        vmsg, xagg = vo.attn_message(q, k, v, dk, dv, g, L.cutoff, L.num_heads, pre_act=True)
        tail_on_side = side is not None and NODE_LEVEL_ON_SIDE_STREAM >= 2
        if tail_on_side:                                                           # o_proj (atoms) under s_proj + the vector aggregation (edges)
            side.wait_stream(main)
            with torch.cuda.stream(side):
                o = vo.lin(xagg, L.o_proj)
        # [E, 2H] = pre-activation of [s1|s2] (act applied inside vec_aggregate); vmsg's gradient is consumed by the attention backward's CSR walk
        sact = vo.lin(vmsg, L.s_proj, False, md, grad_tail_unread=True)
        vagg = vo.vec_aggregate(vl, sact, dvec, g, pre_act=True)
        if tail_on_side:                                                           # the residual node update (atoms) under the edge update (edges)
            side.wait_stream(main)
            with torch.cuda.stream(side):
                xo, veco = vo.node_update(x_res, vec_res, vdot, o, vp, vagg, FOLD_VECDOT_BACKWARD)
            fo = f if L.last_layer else vo.edge_update(wt, ws, t, dvec, f, g, pre_act=True)
            main.wait_stream(side)
            for tt in (o, xo, veco):
                tt.record_stream(main)
            return xo, veco, fo, True
        o = vo.lin(xagg, L.o_proj)
        xo, veco = vo.node_update(x_res, vec_res, vdot, o, vp, vagg, FOLD_VECDOT_BACKWARD)
        if L.last_layer:
            return xo, veco, f, False
        # (wt, ws: node-level — Linear commutes with the gather)
        return xo, veco, vo.edge_update(wt, ws, t, dvec, f, g, pre_act=True), False


def ne_cutoff(block: ViSNetBlock) -> float:
    return float(block.cutoff)


class GatedEquivariantBlock(torch.nn.Module):
    def __init__(self, hidden: int, out: int, scalar_activation: bool = False):
        super().__init__()
        self.hidden, self.out_channels, self.scalar_activation = hidden, out, scalar_activation
        self.vec1_proj = Linear(hidden, hidden, bias=False)
        self.vec2_proj = Linear(hidden, out, bias=False)
        self.update_net = torch.nn.Sequential(Linear(hidden * 2, hidden), torch.nn.SiLU(), Linear(hidden, out * 2))
        for m in (self.vec1_proj, self.vec2_proj, self.update_net[0], self.update_net[2]):
            torch.nn.init.xavier_uniform_(m.weight)
            if m.bias is not None:
                m.bias.data.zero_()

    def forward(self, x: Tensor, v: Tensor):
        """torch_geometric_visnet.py:942-960."""
        n, Hh, O = x.shape[0], self.hidden, self.out_channels
        v1n = vo.spatial_norm(vo.lin(v.reshape(3 * n, Hh), self.vec1_proj), n, Hh)
        v2 = vo.lin(v.reshape(3 * n, Hh), self.vec2_proj)                            # [3n, O]
        u = vo.lin(vo.lin(vo.concat2(x, v1n), self.update_net[0], True), self.update_net[2])     # [n, 2*O]
        return vo.gate(u, v2, n, O, int(self.scalar_activation))


class EquivariantScalar(torch.nn.Module):
    def __init__(self, hidden: int, out: int):
        super().__init__()
        self.output_network = torch.nn.ModuleList([GatedEquivariantBlock(hidden, hidden // 2, True), GatedEquivariantBlock(hidden // 2, out, False)])

    def pre_reduce(self, x: Tensor, v: Tensor) -> Tensor:
        for layer in self.output_network:
            x, v = layer(x, v)
        return x                                                                    # "+ v.sum() * 0" (:1014) is an autograd anchor only


class Atomref(torch.nn.Module):
    def __init__(self, max_z: int = 100):
        super().__init__()
        self.register_buffer("initial_atomref", torch.zeros(max_z, 1))
        self.atomref = Embedding(max_z, 1)
        self.atomref.weight.data.copy_(self.initial_atomref)


class ViSNet(torch.nn.Module):
    FEATURE_SHIFT = 1.0          # visnet.py:50
    READOUT_MODE = 1             # NaN guard + column L2 normalisation, visnet.py:233-242

    def __init__(self, device, hidden_channels: int, cutoff: float = 5.0):
        super().__init__()
        self.device = device
        self.hidden_channels = hidden_channels
        self.cutoff = cutoff
        self.derivative = False
        self.reduce_op = "sum"
        self.representation_model = ViSNetBlock(hidden_channels=hidden_channels)    # keeps ViSNet's default cutoff 5.0 (visnet.py:84-86)
        self.output_model = EquivariantScalar(hidden_channels, hidden_channels // 2)
        self.prior_model = Atomref()
        self.output_model_bary = EquivariantScalar(hidden_channels, hidden_channels // 2)
        self.prior_model_bary = Atomref()
        self.register_buffer("mean", torch.tensor(0.0))
        self.register_buffer("std", torch.tensor(1.0))
        from .schnet import RadiusInteractionGraph
        self.interaction_graph = RadiusInteractionGraph(cutoff, max_num_neighbors=32)   # FGW adjacency, no self loops (visnet.py:90)

    # ---------------------------------------------------------------------------------------------- helpers
    def _prep(self, z: Tensor, batch: Optional[Tensor], num_graphs: Optional[int]):
        if not z.is_cuda:
            raise RuntimeError("ViSNet (MI355X) runs on the GPU only: move the inputs to the device; there is no CPU fallback")
        batch = torch.zeros_like(z) if batch is None else batch
        if num_graphs is None:
            hint_g, _ = ops.batch_hints(batch)                             # DeviceCollator's host-known count (no sync), else one read
            num_graphs = hint_g if hint_g is not None else int(batch[-1].item()) + 1
        return batch, ops.graph_ptr_from_batch(batch, num_graphs), num_graphs

    def _head(self, xs, vs, z, output_model, prior):
        return vo.prior(output_model.pre_reduce(xs, vs), z, prior.atomref.weight, self.std)

    # ---------------------------------------------------------------------------------------------- reference API
    def forward(self, z: Tensor, pos: Tensor, batch: Tensor, num_graphs: Optional[int] = None) -> Tensor:
        """visnet.py:93-122: per-conformer sum of the scalar head."""
        batch, gp, G = self._prep(z, batch, num_graphs)
        xs, vs = self.representation_model(z, pos, gp, G)
        return ops.segment_sum(self._head(xs, vs, z, self.output_model, self.prior_model), gp, G)

    def forward_3d_bary(self, z: Tensor, pos: Tensor, batch: Tensor, num_graphs: Optional[int] = None):
        """visnet.py:124-158: two per-atom heads from the shared representation."""
        batch, gp, G = self._prep(z, batch, num_graphs)
        xs, vs = self.representation_model(z, pos, gp, G)
        if NODE_LEVEL_ON_SIDE_STREAM >= 3:
            # the two heads are independent chains of ~10 atom-level launches each: one of them on the second stream (forward and, through
            # autograd's stream rule, backward)
            main = torch.cuda.current_stream()
            side = getattr(self.representation_model, "_node_stream", None) or torch.cuda.Stream(device=main.device)
            object.__setattr__(self.representation_model, "_node_stream", side)
            side.wait_stream(main)
            with torch.cuda.stream(side):
                h = self._head(xs, vs, z, self.output_model, self.prior_model)
            hb = self._head(xs, vs, z, self.output_model_bary, self.prior_model_bary)
            main.wait_stream(side)
            h.record_stream(main)
            return h, hb
        return self._head(xs, vs, z, self.output_model, self.prior_model), self._head(xs, vs, z, self.output_model_bary, self.prior_model_bary)

    def _compute_barycenter(self, node_feature: Tensor, edge_index, batch: Tensor, batch_size: int, num_conformers: int,
                            max_nodes: Optional[int] = None):
        """visnet.py:160-249 (shift +1.0, NaN guard, column normalisation)."""
        from .schnet import _graph_from_edge_index
        K, G = num_conformers, batch_size * num_conformers
        graph = edge_index if isinstance(edge_index, ops.RadiusGraph) else _graph_from_edge_index(edge_index, batch, G)
        if max_nodes is None:
            hint_g, hint_n = ops.batch_hints(batch)
            max_nodes = hint_n if hint_g == G else None
        if max_nodes is None:
            gp = graph.graph_ptr
            max_nodes = int((gp[1:] - gp[:-1]).max().item())
        Ys, _ = ops.fgw_densify(node_feature, graph, max_nodes, self.FEATURE_SHIFT, adjacency=False)      # (to_dense_adj, :249-252, is read by the solver from the graph's ragged lists)
        N, d = max_nodes, node_feature.shape[1]
        Y, C, T, info, errs = ops.fgw_barycenter_batched(Ys.view(batch_size, K, N, d), None, adjacency=graph)
        self.last_fgw = dict(Y=Y, C=C, T=T, info=info, errs=errs)
        return ops.segment_sum(node_feature, graph.graph_ptr, G), ops.fgw_readout(Y, K, self.READOUT_MODE)

    def forward_w_barycenter(self, z: Tensor, pos: Tensor, num_conformers: int, batch: Optional[Tensor] = None, data_batch=None,
                             max_iter: int = 100, epsilon: float = 0.1, num_graphs: Optional[int] = None, max_nodes: Optional[int] = None):
        """visnet.py:251-288."""
        batch, gp, G = self._prep(z, batch, num_graphs)
        h_3d, h_bary = self.forward_3d_bary(z, pos, batch, num_graphs=G)
        graph = self.interaction_graph.csr(pos, gp, G)                              # visnet.py:276
        _, h_bary = self._compute_barycenter(h_bary, graph, batch, G // num_conformers, num_conformers, max_nodes=max_nodes)
        return ops.segment_sum(h_3d, gp, G), h_bary
