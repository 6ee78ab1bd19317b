"""TEST INFRASTRUCTURE — CPU restatement of the third-party arithmetic the reference's SchNet path runs on.

The reference's SchNet trunk is not in /root/reference: `SchNetNoSum` subclasses `torch_geometric.nn.SchNet`
(conan_fgw/src/model/graph_embeddings/schnet_no_sum.py:6-9,90,109-122).  The pinned third-party versions are
torch-geometric==2.3.0, torch-cluster==1.6.1, torch-scatter==2.1.2 (environment.yml:161-164); none of them is
installed here and there is no network.  This module restates their PUBLISHED algorithms for exactly the names the
reference uses in plain torch, dtype-agnostic so the same code serves as "ref32" and "ref64".

PARITY UNPINNED for this layer: the reference holds no test or golden vector at the PyG boundary.  What IS pinned:
the reference's own glue (`forward_3d_bary`, `_compute_barycenter`, `forward_w_barycenter`) and ViSNet arithmetic,
by importing the reference files over these names (tests/golden/make_model_golden.py).

Conventions fixed here (and used identically by the HIP kernels):
 * radius_graph: candidate (j, i) iff same graph and d2 < r*r STRICTLY, with d2 = fl(fl(dx*dx + dy*dy) + dz*dz)
   evaluated in the dtype of `pos` without FMA contraction; the target itself is a candidate.  Per target the first
   `limit` candidates in ascending source index are kept, limit = cap if loop else cap + 1 (torch-cluster 1.6.1:
   radius_graph -> radius(x, x, r, batch, batch, cap if loop else cap + 1); linear scan in index order = its CUDA
   kernel radius_cuda.cu, the CPU KD-tree order is unspecified when truncating), and only THEN are self pairs removed
   (unless loop).  So a target with >= cap + 1 lower-index candidates keeps cap + 1 = 33 edges, any other truncated
   target keeps cap.  Edges are emitted grouped by target i ascending, sources ascending; edge_index[0] = source j,
   edge_index[1] = target i (flow source_to_target).
"""
from __future__ import annotations

import math
from typing import Optional

import torch
import torch.nn.functional as F
from torch import Tensor
from torch.nn import Embedding, Linear, ModuleList, Sequential

OptTensor = Optional[Tensor]


# ----------------------------------------------------------------------------- torch-cluster 1.6.1
def radius_graph(x: Tensor, r: float, batch: OptTensor = None, loop: bool = False, max_num_neighbors: int = 32,
                 flow: str = "source_to_target", num_workers: int = 1) -> Tensor:
    assert flow == "source_to_target"
    n_total = x.shape[0]
    if batch is None:
        batch = torch.zeros(n_total, dtype=torch.long)
    rows, cols = [], []
    r2 = torch.tensor(r, dtype=x.dtype) * torch.tensor(r, dtype=x.dtype)
    counts = torch.bincount(batch) if n_total else torch.zeros(0, dtype=torch.long)
    ptr = torch.cat([torch.zeros(1, dtype=torch.long), counts.cumsum(0)])
    for g in range(len(counts)):
        lo, hi = int(ptr[g]), int(ptr[g + 1])
        if hi == lo:
            continue
        p = x[lo:hi]
        diff = p[:, None, :] - p[None, :, :]               # [i, j, 3] = pos_i - pos_j
        sq = diff * diff
        d2 = (sq[..., 0] + sq[..., 1]) + sq[..., 2]
        ok = d2 < r2                                        # the target itself is a candidate (d2 = 0)
        # torch-cluster 1.6.1 radius_graph: radius(x, x, r, batch, batch, cap if loop else cap + 1) keeps the first `limit`
        # candidates in ascending source index (the CUDA kernel's linear scan; the CPU KD-tree order is unspecified when it
        # truncates), THEN removes the self pairs: a target with >= cap + 1 lower-index candidates keeps cap + 1 edges.
        limit = max_num_neighbors if loop else max_num_neighbors + 1
        rank = ok.cumsum(1)
        ok = ok & (rank <= limit)
        if not loop:
            ok = ok & ~torch.eye(hi - lo, dtype=torch.bool)
        i_idx, j_idx = ok.nonzero(as_tuple=True)            # sorted by i then j
        rows.append(j_idx + lo)
        cols.append(i_idx + lo)
    if not rows:
        return torch.zeros(2, 0, dtype=torch.long)
    return torch.stack([torch.cat(rows), torch.cat(cols)])


# ----------------------------------------------------------------------------- torch_geometric.utils 2.3.0
def scatter(src: Tensor, index: Tensor, dim: int = 0, dim_size: Optional[int] = None, reduce: str = "sum") -> Tensor:
    assert reduce in ("sum", "add", "mean")
    if dim < 0:
        dim += src.dim()
    if dim_size is None:
        dim_size = int(index.max()) + 1 if index.numel() else 0
    shape = list(src.shape)
    shape[dim] = dim_size
    out = torch.zeros(shape, dtype=src.dtype, device=src.device)
    view = [1] * src.dim()
    view[dim] = -1
    idx = index.view(view).expand_as(src)
    out.scatter_add_(dim, idx, src)
    if reduce == "mean":
        cnt = torch.zeros(dim_size, dtype=src.dtype).scatter_add_(0, index, torch.ones_like(index, dtype=src.dtype))
        out = out / cnt.clamp(min=1).view(view)
    return out


def to_dense_batch(x: Tensor, batch: OptTensor = None, fill_value: float = 0.0, max_num_nodes: Optional[int] = None):
    if batch is None:
        batch = torch.zeros(x.shape[0], dtype=torch.long)
    G = int(batch.max()) + 1
    counts = torch.bincount(batch, minlength=G)
    ptr = torch.cat([torch.zeros(1, dtype=torch.long), counts.cumsum(0)])
    N = int(counts.max()) if max_num_nodes is None else max_num_nodes
    slot = torch.arange(x.shape[0]) - ptr[batch]
    out = torch.full((G, N) + tuple(x.shape[1:]), fill_value, dtype=x.dtype)
    mask = torch.zeros(G, N, dtype=torch.bool)
    out[batch, slot] = x
    mask[batch, slot] = True
    return out, mask


def to_dense_adj(edge_index: Tensor, batch: OptTensor = None, edge_attr: OptTensor = None,
                 max_num_nodes: Optional[int] = None) -> Tensor:
    if batch is None:
        n = int(edge_index.max()) + 1 if edge_index.numel() else 0
        batch = torch.zeros(n, dtype=torch.long)
    G = int(batch.max()) + 1
    counts = torch.bincount(batch, minlength=G)
    ptr = torch.cat([torch.zeros(1, dtype=torch.long), counts.cumsum(0)])
    N = int(counts.max()) if max_num_nodes is None else max_num_nodes
    g = batch[edge_index[0]]
    r = edge_index[0] - ptr[g]
    c = edge_index[1] - ptr[g]
    adj = torch.zeros(G, N, N, dtype=torch.float32)
    adj.index_put_((g, r, c), torch.ones(edge_index.shape[1], dtype=torch.float32), accumulate=True)
    return adj


# ----------------------------------------------------------------------------- torch_geometric.nn.aggr 2.3.0
class SumAggregation(torch.nn.Module):
    """`Aggregation.__call__(x, index=None, dim=...)`: index None => every row reduces into ONE output row."""

    def forward(self, x: Tensor, index: OptTensor = None, ptr=None, dim_size=None, dim: int = -2) -> Tensor:
        if index is None:
            index = torch.zeros(x.shape[dim], dtype=torch.long)
        return scatter(x, index, dim=dim, dim_size=dim_size, reduce="sum")


class MeanAggregation(torch.nn.Module):
    def forward(self, x: Tensor, index: OptTensor = None, ptr=None, dim_size=None, dim: int = -2) -> Tensor:
        if index is None:
            index = torch.zeros(x.shape[dim], dtype=torch.long)
        return scatter(x, index, dim=dim, dim_size=dim_size, reduce="mean")


def aggregation_resolver(name: str):
    return {"add": SumAggregation, "sum": SumAggregation, "mean": MeanAggregation}[name]()


# ----------------------------------------------------------------------------- torch_geometric.nn.MessagePassing
class MessagePassing(torch.nn.Module):
    """The subset the reference's vendored ViSNet and PyG's CFConv rely on: flow source_to_target, node_dim=0,
    `x_j = x[edge_index[0]]`, `x_i = x[edge_index[1]]`, aggregation index `edge_index[1]`."""

    def __init__(self, aggr: str = "add", node_dim: int = 0, **kwargs):
        super().__init__()
        self.aggr = aggr
        self.node_dim = node_dim

    def _collect(self, fn, edge_index, size, kwargs):
        import inspect
        out = {}
        for name in inspect.signature(fn).parameters:
            if name in ("index", "ptr", "dim_size"):
                continue
            if name.endswith("_j") or name.endswith("_i"):
                src = kwargs[name[:-2]]
                out[name] = src.index_select(self.node_dim, edge_index[0] if name.endswith("_j") else edge_index[1])
            elif name in kwargs:
                out[name] = kwargs[name]
        return out

    def propagate(self, edge_index: Tensor, size=None, **kwargs):
        dim_size = size if isinstance(size, int) else None
        if dim_size is None:
            for v in kwargs.values():
                if isinstance(v, Tensor) and v.dim() > 0:
                    dim_size = v.shape[self.node_dim]
                    break
        msg = self.message(**self._collect(self.message, edge_index, size, kwargs))
        import inspect
        agg_params = inspect.signature(self.aggregate).parameters
        extra = {}
        if "ptr" in agg_params:
            extra["ptr"] = None
        out = self.aggregate(msg, edge_index[1], dim_size=dim_size, **extra)
        return self.update(out)

    def aggregate(self, inputs, index, ptr=None, dim_size=None):
        return scatter(inputs, index, dim=self.node_dim, dim_size=dim_size, reduce="sum" if self.aggr == "add" else self.aggr)

    def update(self, inputs):
        return inputs

    def edge_updater(self, edge_index: Tensor, **kwargs):
        return self.edge_update(**self._collect(self.edge_update, edge_index, None, kwargs))


# ----------------------------------------------------------------------------- torch_geometric.nn.models.schnet 2.3.0
class ShiftedSoftplus(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.shift = math.log(2.0)

    def forward(self, x: Tensor) -> Tensor:
        return F.softplus(x) - self.shift


class GaussianSmearing(torch.nn.Module):
    def __init__(self, start: float = 0.0, stop: float = 5.0, num_gaussians: int = 50):
        super().__init__()
        offset = torch.linspace(start, stop, num_gaussians)
        self.coeff = -0.5 / (offset[1] - offset[0]).item() ** 2
        self.register_buffer("offset", offset)

    def forward(self, dist: Tensor) -> Tensor:
        dist = dist.view(-1, 1) - self.offset.to(dist.dtype).view(1, -1)
        return torch.exp(self.coeff * torch.pow(dist, 2))


class RadiusInteractionGraph(torch.nn.Module):
    def __init__(self, cutoff: float = 10.0, max_num_neighbors: int = 32):
        super().__init__()
        self.cutoff = cutoff
        self.max_num_neighbors = max_num_neighbors

    def forward(self, pos: Tensor, batch: Tensor):
        edge_index = radius_graph(pos, r=self.cutoff, batch=batch, max_num_neighbors=self.max_num_neighbors)
        row, col = edge_index
        edge_weight = (pos[row] - pos[col]).norm(dim=-1)
        return edge_index, edge_weight


class CFConv(MessagePassing):
    def __init__(self, in_channels: int, out_channels: int, num_filters: int, nn: Sequential, cutoff: float):
        super().__init__(aggr="add")
        self.lin1 = Linear(in_channels, num_filters, bias=False)
        self.lin2 = Linear(num_filters, out_channels)
        self.nn = nn
        self.cutoff = cutoff
        self.reset_parameters()

    def reset_parameters(self):
        torch.nn.init.xavier_uniform_(self.lin1.weight)
        torch.nn.init.xavier_uniform_(self.lin2.weight)
        self.lin2.bias.data.fill_(0)

    def forward(self, x: Tensor, edge_index: Tensor, edge_weight: Tensor, edge_attr: Tensor) -> Tensor:
        C = 0.5 * (torch.cos(edge_weight * math.pi / self.cutoff) + 1.0)
        W = self.nn(edge_attr) * C.view(-1, 1)
        x = self.lin1(x)
        x = self.propagate(edge_index, x=x, W=W)
        x = self.lin2(x)
        return x

    def message(self, x_j: Tensor, W: Tensor) -> Tensor:
        return x_j * W


class InteractionBlock(torch.nn.Module):
    def __init__(self, hidden_channels: int, num_gaussians: int, num_filters: int, cutoff: float):
        super().__init__()
        self.mlp = Sequential(Linear(num_gaussians, num_filters), ShiftedSoftplus(), Linear(num_filters, num_filters))
        self.conv = CFConv(hidden_channels, hidden_channels, num_filters, self.mlp, cutoff)
        self.act = ShiftedSoftplus()
        self.lin = Linear(hidden_channels, hidden_channels)
        self.reset_parameters()

    def reset_parameters(self):
        torch.nn.init.xavier_uniform_(self.mlp[0].weight)
        self.mlp[0].bias.data.fill_(0)
        torch.nn.init.xavier_uniform_(self.mlp[2].weight)
        self.mlp[2].bias.data.fill_(0)
        self.conv.reset_parameters()
        torch.nn.init.xavier_uniform_(self.lin.weight)
        self.lin.bias.data.fill_(0)

    def forward(self, x: Tensor, edge_index: Tensor, edge_weight: Tensor, edge_attr: Tensor) -> Tensor:
        x = self.conv(x, edge_index, edge_weight, edge_attr)
        x = self.act(x)
        x = self.lin(x)
        return x


class SchNet(torch.nn.Module):
    """Constructor surface and attribute names of torch_geometric.nn.SchNet 2.3.0 (only what the reference's
    subclass touches: schnet_no_sum.py:109-122 and the attributes used in :159-186)."""

    def __init__(self, hidden_channels: int = 128, num_filters: int = 128, num_interactions: int = 6,
                 num_gaussians: int = 50, cutoff: float = 10.0, interaction_graph=None, max_num_neighbors: int = 32,
                 readout: str = "add", dipole: bool = False, mean=None, std=None, atomref: OptTensor = None):
        super().__init__()
        self.hidden_channels = hidden_channels
        self.num_filters = num_filters
        self.num_interactions = num_interactions
        self.num_gaussians = num_gaussians
        self.cutoff = cutoff
        self.dipole = dipole
        self.sum_aggr = SumAggregation()
        self.readout = aggregation_resolver("sum" if dipole else readout)
        self.mean, self.std, self.scale = mean, std, None
        self.embedding = Embedding(100, hidden_channels, padding_idx=0)
        self.interaction_graph = interaction_graph if interaction_graph is not None else RadiusInteractionGraph(cutoff, max_num_neighbors)
        self.distance_expansion = GaussianSmearing(0.0, cutoff, num_gaussians)
        self.interactions = ModuleList([InteractionBlock(hidden_channels, num_gaussians, num_filters, cutoff)
                                        for _ in range(num_interactions)])
        self.lin1 = Linear(hidden_channels, hidden_channels // 2)
        self.act = ShiftedSoftplus()
        self.lin2 = Linear(hidden_channels // 2, 1)
        self.register_buffer("initial_atomref", atomref)
        self.atomref = None
        self.reset_parameters()

    def reset_parameters(self):
        self.embedding.reset_parameters()
        for interaction in self.interactions:
            interaction.reset_parameters()
        torch.nn.init.xavier_uniform_(self.lin1.weight)
        self.lin1.bias.data.fill_(0)
        torch.nn.init.xavier_uniform_(self.lin2.weight)
        self.lin2.bias.data.fill_(0)
