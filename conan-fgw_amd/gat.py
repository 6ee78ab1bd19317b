"""Covalent (2-D bond graph) branch on the HIP path: drop-in for `GATBased`
(conan_fgw/src/model/graph_embeddings/gat.py:5-25, built by EquivModelsHolder.get_model("gat", ..., feat_dim=128),
common.py:534-537, and called from EmbeddingsWithGATAggregationBaryCenter.forward, schnet_based_models.py:166-168).

Two PyG-2.3.0 GATConv layers (heads 1, edge_dim 3, self loops with the mean incoming edge attribute, LeakyReLU 0.2, softmax
over the incoming edges) + SumAggregation per conformer graph.  Parameter names follow PyG (`gat_convN.lin_src.weight`,
`lin_dst.weight` = the same tensor, `att_src/att_dst/att_edge` [1,1,C], `lin_edge.weight`, `bias`) so a reference checkpoint
loads strictly.  No CPU path: every op is a conan_* HIP entry point.
"""
from __future__ import annotations

import math
from typing import Optional

import torch
from torch import Tensor, nn

from . import ops
from ._lib import call, lib, ptr, stream_ptr

f32 = torch.float32
i32 = torch.int32


class BondGraph:
    """CSR by target and by source of PyG's `edge_index` [2,E] (int64, any order; self loops dropped), built on the device."""

    def __init__(self, edge_index: Tensor, num_nodes: int):
        if not edge_index.is_cuda:
            raise RuntimeError("conan_fgw_amd has no CPU path: edge_index must live on the GPU")
        dev = edge_index.device
        ei = edge_index.to(torch.int64).contiguous()
        self.num_nodes, self.num_edges = int(num_nodes), int(ei.shape[1])
        n, E = self.num_nodes, max(self.num_edges, 1)
        ws = torch.empty(2 * (n + 1), dtype=i32, device=dev)
        self.rowptr, self.t_rowptr = torch.empty(n + 1, dtype=i32, device=dev), torch.empty(n + 1, dtype=i32, device=dev)
        self.col, self.eid, self.t_pos, self.t_tgt = (torch.empty(E, dtype=i32, device=dev) for _ in range(4))
        call("conan_bond_graph_csr", ptr(ei), self.num_edges, n, ptr(ws), ptr(self.rowptr), ptr(self.col), ptr(self.eid),
             ptr(self.t_rowptr), ptr(self.t_pos), ptr(self.t_tgt), stream_ptr())


class _GATAggregateFn(torch.autograd.Function):
    """h [n,C] -> softmax-attention aggregation over the bond graph (everything of GATConv after lin_src)."""

    @staticmethod
    def forward(ctx, h, att_src, att_dst, w_edge, att_edge, bias, graph, edge_attr, slope):
        h = h.contiguous()
        n, C = h.shape
        D = w_edge.shape[1]
        dev = h.device
        a_s, a_d = att_src.reshape(-1).contiguous(), att_dst.reshape(-1).contiguous()
        a_e, w_e = att_edge.reshape(-1).contiguous(), w_edge.contiguous()
        ea = edge_attr.to(f32).contiguous()
        v = torch.empty(D, dtype=f32, device=dev)
        call("conan_gat_edge_vec", ptr(w_e), ptr(a_e), C, D, ptr(v), stream_ptr())
        al_s, al_d = torch.empty(n, dtype=f32, device=dev), torch.empty(n, dtype=f32, device=dev)
        call("conan_gat_node_alpha", ptr(h), ptr(a_s), ptr(a_d), n, C, ptr(al_s), ptr(al_d), stream_ptr())
        out = torch.empty(n, C, dtype=f32, device=dev)
        alpha = torch.empty(max(graph.num_edges, 1), dtype=f32, device=dev)
        alpha_self = torch.empty(n, dtype=f32, device=dev)
        call("conan_gat_aggregate_fwd", ptr(h), ptr(al_s), ptr(al_d), ptr(graph.rowptr), ptr(graph.col), ptr(graph.eid), ptr(ea), D, ptr(v),
             ptr(bias), float(slope), n, C, ptr(out), ptr(alpha), ptr(alpha_self), stream_ptr())
        ctx.graph, ctx.slope, ctx.shapes = graph, float(slope), (att_src.shape, att_dst.shape, att_edge.shape)
        ctx.save_for_backward(h, a_s, a_d, w_e, a_e, ea, v, al_s, al_d, alpha, alpha_self)
        return out

    @staticmethod
    def backward(ctx, dout):
        h, a_s, a_d, w_e, a_e, ea, v, al_s, al_d, alpha, alpha_self = ctx.saved_tensors
        g = ctx.graph
        dout = dout.contiguous()
        n, C = h.shape
        D = w_e.shape[1]
        dev = h.device
        ws = torch.empty(int(lib().conan_gat_bwd_ws(n, g.num_edges, C, D)), dtype=f32, device=dev)
        dh = torch.empty_like(h)
        dpar = torch.empty(3 * C + D, dtype=f32, device=dev)            # d att_src | d att_dst | d bias | dv
        call("conan_gat_aggregate_bwd", ptr(h), ptr(dout), ptr(alpha), ptr(alpha_self), ptr(al_s), ptr(al_d), ptr(a_s), ptr(a_d), ptr(g.rowptr),
             ptr(g.col), ptr(g.eid), ptr(g.t_rowptr), ptr(g.t_pos), ptr(g.t_tgt), ptr(ea), D, ptr(v), ctx.slope, n, g.num_edges, C, ptr(ws), ptr(dh),
             ptr(dpar), stream_ptr())
        dw_e, datt_e = torch.empty_like(w_e), torch.empty(C, dtype=f32, device=dev)
        call("conan_gat_edge_vec_bwd", ptr(w_e), ptr(a_e), ptr(dpar[3 * C:]), C, D, ptr(dw_e), ptr(datt_e), stream_ptr())
        s_src, s_dst, s_edge = ctx.shapes
        return (dh, dpar[:C].view(s_src), dpar[C:2 * C].view(s_dst), dw_e, datt_e.view(s_edge),
                dpar[2 * C:3 * C] if ctx.needs_input_grad[5] else None, None, None, None)


def _glorot(t: Tensor):
    a = math.sqrt(6.0 / (t.size(-2) + t.size(-1)))
    with torch.no_grad():
        t.uniform_(-a, a)


class GATConv(nn.Module):
    """torch_geometric.nn.GATConv(in_channels, out_channels, edge_dim=edge_dim) with its defaults (heads=1, concat, slope 0.2,
    add_self_loops, fill_value='mean', bias)."""

    def __init__(self, in_channels: int, out_channels: int, edge_dim: int = 3, negative_slope: float = 0.2):
        super().__init__()
        self.in_channels, self.out_channels, self.edge_dim, self.negative_slope = in_channels, out_channels, edge_dim, negative_slope
        self.heads = 1
        self.lin_src = nn.Linear(in_channels, out_channels, bias=False)
        self.lin_dst = self.lin_src                                   # PyG: one shared module registered under both names
        self.att_src = nn.Parameter(torch.empty(1, 1, out_channels))
        self.att_dst = nn.Parameter(torch.empty(1, 1, out_channels))
        self.lin_edge = nn.Linear(edge_dim, out_channels, bias=False)
        self.att_edge = nn.Parameter(torch.empty(1, 1, out_channels))
        self.bias = nn.Parameter(torch.zeros(out_channels))
        for t in (self.lin_src.weight, self.lin_edge.weight, self.att_src, self.att_dst, self.att_edge):
            _glorot(t)

    def forward(self, x: Tensor, graph: BondGraph, edge_attr: Tensor) -> Tensor:
        h = ops.linear(x, self.lin_src.weight, None)
        return _GATAggregateFn.apply(h, self.att_src, self.att_dst, self.lin_edge.weight, self.att_edge, self.bias, graph, edge_attr,
                                     self.negative_slope)


class GATBased(nn.Module):
    """Drop-in for gat.py:5-25.  `in_channels` replaces PyG's lazy `in_channels=-1` (the width of `batch.x`, 9 atom features
    in the reference's featurisation); everything else has the reference's signature."""

    def __init__(self, out_channels: int = 64, edge_dim: int = 3, in_channels: int = 9):
        super().__init__()
        self.gat_conv1 = GATConv(in_channels, out_channels, edge_dim)
        self.gat_conv2 = GATConv(out_channels, out_channels, edge_dim)

    def forward(self, x: Tensor, edge_index: Tensor, edge_attr: Tensor, batch: Tensor, num_graphs: Optional[int] = None) -> Tensor:
        if not x.is_cuda:
            raise RuntimeError("conan_fgw_amd has no CPU path: inputs must live on the GPU")
        x = x.float().contiguous()                                     # gat.py:20
        edge_attr = edge_attr.float()                                  # gat.py:21
        graph = BondGraph(edge_index, x.shape[0])
        h = self.gat_conv1(x, graph, edge_attr)
        h = self.gat_conv2(h, graph, edge_attr)
        if num_graphs is None:
            num_graphs = ops.batch_hints(batch)[0]                     # DeviceCollator's host-known count
        if num_graphs is None:
            num_graphs = int(batch.max().item()) + 1                   # host sync, as inside PyG's SumAggregation
        gp = ops.graph_ptr_from_batch(batch, num_graphs)
        return ops.segment_sum(h, gp, num_graphs)                      # self.node_aggregation(h, batch)
