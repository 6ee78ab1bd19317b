"""CPU restatement of the covalent branch — TEST INFRASTRUCTURE ONLY (never imported by the product).

`GATBasedOracle` follows conan_fgw/src/model/graph_embeddings/gat.py:5-25: two `torch_geometric.nn.GATConv` layers
(in_channels=-1 -> lazily the width of batch.x, out_channels, edge_dim=3, everything else default) and a SumAggregation over
`batch`.  torch_geometric==2.3.0 (environment.yml:163) is not in /root/reference and not installable here, so `GATConvOracle`
restates its published forward (torch_geometric/nn/conv/gat_conv.py, v2.3.0) from the algorithm:

    heads=1, concat=True, negative_slope=0.2, dropout=0, add_self_loops=True, fill_value='mean', bias=True
    lin_src = lin_dst = Linear(in, H*C, bias=False)  (one shared module when in_channels is an int, registered under both names)
    x_src = x_dst = lin_src(x).view(-1, H, C);  alpha_src = (x_src*att_src).sum(-1);  alpha_dst = (x_dst*att_dst).sum(-1)
    remove_self_loops; add_self_loops(edge_attr fill = mean of the attributes of each node's incoming edges; 0 without any)
    edge_update: alpha = alpha_j + alpha_i + (lin_edge(edge_attr).view(-1,H,C)*att_edge).sum(-1); leaky_relu;
                 softmax over the edges of a target (exp(a - max) / (sum + 1e-16))
    out_i = sum_j alpha_ji * x_src[j];  out = out.view(-1, H*C) + bias

PARITY UNPINNED: the reference holds no test or stored vector for this branch; the rules above are from the package's
published source as remembered (SURVEY.md Appendix B conventions: edge_index[0] = source j, edge_index[1] = target i).
"""
from __future__ import annotations

import math

import torch
from torch import Tensor, nn


def _glorot(t: Tensor):
    a = math.sqrt(6.0 / (t.size(-2) + t.size(-1)))
    with torch.no_grad():
        t.uniform_(-a, a)


class GATConvOracle(nn.Module):
    def __init__(self, in_channels: int, out_channels: int, edge_dim: int = 3, negative_slope: float = 0.2):
        super().__init__()
        self.in_channels, self.out_channels, self.edge_dim, self.negative_slope = in_channels, out_channels, edge_dim, negative_slope
        self.lin_src = nn.Linear(in_channels, out_channels, bias=False)
        self.lin_dst = self.lin_src                                   # shared module, both names appear in the state_dict
        self.att_src = nn.Parameter(torch.empty(1, 1, out_channels))
        self.att_dst = nn.Parameter(torch.empty(1, 1, out_channels))
        self.lin_edge = nn.Linear(edge_dim, out_channels, bias=False)
        self.att_edge = nn.Parameter(torch.empty(1, 1, out_channels))
        self.bias = nn.Parameter(torch.zeros(out_channels))
        for t in (self.lin_src.weight, self.lin_edge.weight, self.att_src, self.att_dst, self.att_edge):
            _glorot(t)

    def forward(self, x: Tensor, edge_index: Tensor, edge_attr: Tensor) -> Tensor:
        n, C = x.shape[0], self.out_channels
        h = self.lin_src(x)                                           # [n, C]  (H = 1)
        a_src = (h * self.att_src.view(1, C)).sum(-1)
        a_dst = (h * self.att_dst.view(1, C)).sum(-1)
        src, dst = edge_index[0], edge_index[1]
        keep = src != dst                                             # remove_self_loops
        src, dst, ea = src[keep], dst[keep], edge_attr[keep]
        # add_self_loops(fill_value='mean'): scatter-mean of the incoming attributes per target
        cnt = torch.zeros(n, dtype=ea.dtype).index_add_(0, dst, torch.ones(dst.numel(), dtype=ea.dtype))
        loop_attr = torch.zeros(n, ea.shape[1], dtype=ea.dtype).index_add_(0, dst, ea) / cnt.clamp(min=1.0)[:, None]
        loop = torch.arange(n)
        src, dst, ea = torch.cat([src, loop]), torch.cat([dst, loop]), torch.cat([ea, loop_attr])
        a_edge = (self.lin_edge(ea) * self.att_edge.view(1, C)).sum(-1)
        alpha = torch.nn.functional.leaky_relu(a_src[src] + a_dst[dst] + a_edge, self.negative_slope)
        amax = torch.full((n,), -float("inf"), dtype=alpha.dtype).scatter_reduce(0, dst, alpha.detach(), "amax", include_self=True)
        ex = (alpha - amax[dst]).exp()
        den = torch.zeros(n, dtype=alpha.dtype).index_add_(0, dst, ex) + 1e-16
        alpha = ex / den[dst]
        out = torch.zeros(n, C, dtype=h.dtype).index_add_(0, dst, alpha[:, None] * h[src])
        return out + self.bias


class GATBasedOracle(nn.Module):
    """gat.py:5-25 (in_channels = width of batch.x, fixed at construction instead of lazily)."""

    def __init__(self, out_channels: int = 64, edge_dim: int = 3, in_channels: int = 9):
        super().__init__()
        self.gat_conv1 = GATConvOracle(in_channels, out_channels, edge_dim)
        self.gat_conv2 = GATConvOracle(out_channels, out_channels, edge_dim)

    def forward(self, x: Tensor, edge_index: Tensor, edge_attr: Tensor, batch: Tensor) -> Tensor:
        dt = self.gat_conv1.lin_src.weight.dtype
        x, edge_attr = x.to(dt), edge_attr.to(dt)                      # gat.py:20-21 (.float())
        h = self.gat_conv1(x, edge_index, edge_attr)
        h = self.gat_conv2(h, edge_index, edge_attr)
        G = int(batch.max()) + 1 if batch.numel() else 0
        return torch.zeros(G, h.shape[1], dtype=h.dtype).index_add_(0, batch, h)       # SumAggregation
