"""Drop-in `SchNetNoSum` for MI355X.

Mirrors the reference's backbone class (conan_fgw/src/model/graph_embeddings/schnet_no_sum.py:90-354): same constructor
arguments, same four methods (`forward`, `forward_3d_bary`, `_compute_barycenter`, `forward_w_barycenter`), same attribute
names read by the Lightning heads (`hidden_channels`, schnet_based_models.py:95) and the same 49 `state_dict` keys
(`embedding.weight`, `distance_expansion.offset`, `interactions.{i}.mlp/conv/lin.*`, `lin1`, `lin2`, `lin1_bary`,
`lin2_bary`) so that stage-1 checkpoints load with `strict=True` (train_val.py:175-183).

The torch.nn modules below are parameter containers only; every forward runs on the HIP kernels of
libconan_fgw_hip.so through conan_fgw_amd.ops.  No CPU fallback: CPU tensors raise.

Deviations from the reference, all documented in DESIGN.md:
 * the radius graph is built once per `forward_w_barycenter` instead of twice (schnet_no_sum.py:208 and :342 give the
   same result);
 * `h_bary` is returned on the input's device (the reference allocates it on the device captured at construction,
   usually the CPU, and the caller moves it back: schnet_no_sum.py:254-256, schnet_based_models.py:162-163);
 * optional `num_graphs` / `max_nodes` hints avoid the host synchronisations the reference performs
   (`len(batch.unique())`, :345; `to_dense_batch`, :242).
"""
from __future__ import annotations

import math
from typing import Callable, Optional

import torch
from torch import Tensor
from torch.nn import Embedding, Linear, ModuleList, Sequential

from . import ops

OptTensor = Optional[Tensor]


class ShiftedSoftplus(torch.nn.Module):
    """softplus(x) - ln 2 (PyG `ShiftedSoftplus`).  On the model path the activation is fused into the epilogue of
    `conan_linear_fwd`; the module keeps `interactions.{i}.mlp`'s Sequential indices (0, 2) and, called directly (the
    reference's `self.act(h)`, schnet_no_sum.py:178,227,231), runs the stand-alone HIP kernel."""

    def __init__(self):
        super().__init__()
        self.shift = math.log(2.0)

    def forward(self, x: Tensor) -> Tensor:
        return ops.shifted_softplus(x)


class GaussianSmearing(torch.nn.Module):
    def __init__(self, start: float = 0.0, stop: float = 5.0, num_gaussians: int = 50):
        super().__init__()
        offset = torch.linspace(start, stop, num_gaussians)
        self.coeff = -0.5 / (offset[1] - offset[0]).item() ** 2
        self.register_buffer("offset", offset)

    def forward(self, graph: ops.RadiusGraph) -> Tensor:
        return ops.rbf_expand(graph, self.offset, self.coeff)


class RadiusInteractionGraph(torch.nn.Module):
    """`(pos, batch) -> (edge_index, edge_weight)` like the reference's interaction graph; `.csr()` is the fast path."""

    def __init__(self, cutoff: float = 10.0, max_num_neighbors: int = 32):
        super().__init__()
        self.cutoff = cutoff
        self.max_num_neighbors = max_num_neighbors

    def csr(self, pos: Tensor, graph_ptr: Tensor, num_graphs: int) -> ops.RadiusGraph:
        return ops.RadiusGraph(pos, graph_ptr, num_graphs, self.cutoff, self.max_num_neighbors, loop=False)

    def forward(self, pos: Tensor, batch: Tensor):
        num_graphs = int(batch[-1].item()) + 1 if batch.numel() else 0
        g = self.csr(pos, ops.graph_ptr_from_batch(batch, num_graphs), num_graphs)
        return g.edge_index(), g.edge_weight()


FUSE_FILTER_INTO_GATHER = True      # inference path of CFConv: tools / tests switch it off to compare (never read from the environment)
FUSE_MIN_FILTER_BYTES = 192 << 20   # fuse when the pair-shared filter tensor would not stay in the 256 MiB Infinity Cache between its two kernels


def _filter_tensor_outgrows_cache(graph: "ops.RadiusGraph", num_filters: int) -> bool:
    """Measured (profiles/r5_filter_cfconv_fused_v3.txt): the fused forward generates one filter row per DIRECTED edge (twice the matrix work of the
    pair-shared generator), which pays once the [pairs, F] tensor the two-kernel form writes and re-reads no longer fits the Infinity Cache —
    Lipophilicity-sized batches: stage-2 forward 4.03 -> 3.78 ms — and does not below that (cfg2: 132 MB, 1.31-1.32 ms either way).  The edge count is
    device-side; the host-known estimate is atoms x min(cap, atoms per conformer - 1), no sync."""
    n, G = graph.num_atoms, max(1, graph.num_graphs)
    est_pairs = 0.5 * n * min(float(graph.cap), max(0.0, n / G - 1.0))
    return est_pairs * num_filters * 4 >= FUSE_MIN_FILTER_BYTES

class CFConv(torch.nn.Module):
    def __init__(self, in_channels: int, out_channels: int, num_filters: int, nn: Sequential, cutoff: float):
        super().__init__()
        self.lin1 = Linear(in_channels, num_filters, bias=False)
        self.lin2 = Linear(num_filters, out_channels)
        self.nn = nn
        self.cutoff = cutoff
        self.reset_parameters()

    def reset_parameters(self):
        torch.nn.init.xavier_uniform_(self.lin1.weight)
        torch.nn.init.xavier_uniform_(self.lin2.weight)
        self.lin2.bias.data.fill_(0)

    def forward(self, x: Tensor, graph: ops.RadiusGraph, rbf, tap: bool = False):
        if isinstance(rbf, Tensor):                                                       # generic path: any (Gs, F)
            md = graph.num_edges_dev
            h1 = ops.linear(rbf, self.nn[0].weight, self.nn[0].bias, act=True, m_dev=md)   # mlp[0] + ssp
            w_raw = ops.linear(h1, self.nn[2].weight, self.nn[2].bias, m_dev=md)          # mlp[2]
            W = ops.cutoff_scale(w_raw, graph)                                            # * C(d)
        else:                                                                             # fused: rbf -> mlp -> * C(d) in one kernel
            offset, coeff = rbf
            if (FUSE_FILTER_INTO_GATHER and not torch.is_grad_enabled() and _filter_tensor_outgrows_cache(graph, self.nn[2].weight.shape[0])
                    and ops.filter_cfconv_supported(offset.shape[0], self.nn[2].weight.shape[0])):
                # inference: the filter rows are consumed where they are generated — no [E, F] tensor, one launch for the whole edge half
                x_in = x if tap else None
                x = ops.linear(x, self.lin1.weight)
                x = ops.filter_cfconv(x, graph, offset, coeff, self.nn[0].weight, self.nn[0].bias, self.nn[2].weight, self.nn[2].bias)
                return (x, x_in) if tap else x
            W = ops.filter_generate(graph, offset, coeff, self.nn[0].weight, self.nn[0].bias, self.nn[2].weight, self.nn[2].bias)
        x_in = None
        if tap:                                                                           # lin1 (no bias), handing x through for the residual
            x, x_in = ops.linear_tap(x, self.lin1.weight)
        else:
            x = ops.linear(x, self.lin1.weight)
        fused = not isinstance(rbf, Tensor)            # fused path: one filter row per undirected pair (W_ij = W_ji)
        x = ops.cfconv(x, W, graph, pre_cutoff_grad=fused, use_pairs=fused)               # propagate: gather * W, scatter-add
        return (x, x_in) if tap else x                                                    # lin2 applied by the caller (fused with ssp)


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

    def forward(self, x: Tensor, graph: ops.RadiusGraph, rbf: Tensor) -> Tensor:
        """Returns x + lin(ssp(conv(x)))  — the residual of schnet_no_sum.py:164 is fused into the last linear."""
        # the residual's x is taken from conv.lin1's second output: its gradient then meets lin1's input gradient inside one GEMM
        m, x = self.conv(x, graph, rbf, tap=True)
        # conv.lin2 + InteractionBlock.act, then lin + x: one launch at node-level sizes (ops.mlp2), two linear kernels otherwise
        return ops.mlp2(m, self.conv.lin2.weight, self.conv.lin2.bias, self.lin.weight, self.lin.bias, residual=x)


class SchNetNoSum(torch.nn.Module):
    FEATURE_SHIFT = 0.5          # schnet_no_sum.py:59
    READOUT_MODE = 0

    def __init__(self, device, hidden_channels: int = 128, num_filters: int = 128, num_interactions: int = 6,
                 num_gaussians: int = 50, cutoff: float = 10.0, interaction_graph: Optional[Callable] = None,
                 max_num_neighbors: int = 32, readout: str = "add", dipole: bool = False, mean: Optional[float] = None,
                 std: Optional[float] = None, atomref: OptTensor = None, use_covalent: bool = False, use_readout: bool = True):
        super().__init__()
        if interaction_graph is not None:
            raise NotImplementedError("custom interaction_graph callables are not supported; the radius graph is a HIP kernel")
        if use_covalent:
            raise NotImplementedError("use_covalent=True (ESAN-only branch, schnet_no_sum.py:132-142) is out of scope")
        if dipole or atomref is not None:
            raise NotImplementedError("dipole / atomref are not used by ConAN (common.py:524-529)")
        if readout not in ("add", "sum"):
            raise NotImplementedError("only the sum readout is used by ConAN")
        self.device = device
        self.hidden_channels = hidden_channels
        self.num_filters = num_filters
        self.num_interactions = num_interactions
        self.num_gaussians = num_gaussians
        self.cutoff = cutoff
        self.use_readout = use_readout
        self.use_covalent = use_covalent
        self.fused_filter = True            # False: compose the filter from rbf / linear / cutoff kernels (any shape)
        self.mean, self.std, self.scale = mean, std, None
        self.embedding = Embedding(100, hidden_channels, padding_idx=0)
        self.interaction_graph = RadiusInteractionGraph(cutoff, max_num_neighbors)
        self.distance_expansion = GaussianSmearing(0.0, cutoff, num_gaussians)
        self.interactions = ModuleList([InteractionBlock(hidden_channels, num_gaussians, num_filters, cutoff)
                                        for _ in range(num_interactions)])
        self.lin1 = Linear(hidden_channels, hidden_channels // 2)
        self.act = ShiftedSoftplus()
        self.register_buffer("initial_atomref", None)
        self.atomref = None
        # PyG SchNet.reset_parameters for the inherited layers ...
        torch.nn.init.xavier_uniform_(self.lin1.weight)
        self.lin1.bias.data.fill_(0)
        # ... ConAN's own layers keep torch's default Linear init (schnet_no_sum.py:126-130)
        self.lin1_bary = Linear(hidden_channels, hidden_channels // 2)
        self.lin2_bary = Linear(hidden_channels // 2, hidden_channels // 2)
        self.lin2 = Linear(hidden_channels // 2, hidden_channels // 2)

    # ---------------------------------------------------------------------------------------------- helpers
    def _graphs(self, z: Tensor, pos: Tensor, batch: OptTensor, num_graphs: Optional[int]):
        if not z.is_cuda:
            raise RuntimeError("SchNetNoSum (MI355X) runs on the GPU only: move the inputs to the device; there is no CPU fallback")
        batch = torch.zeros_like(z) if batch is None else batch            # schnet_no_sum.py:157
        hint_g, hint_n = ops.batch_hints(batch)                            # sizes DeviceCollator left on the tensor (reference-shaped calls carry no size arguments)
        if num_graphs is None:
            num_graphs = hint_g if hint_g is not None else int(batch[-1].item()) + 1      # host sync only for foreign tensors; pass num_graphs= to avoid it
        gptr = ops.graph_ptr_from_batch(batch, num_graphs)
        graph = self.interaction_graph.csr(pos, gptr, num_graphs)
        graph.max_nodes_hint = hint_n if hint_g == num_graphs else None
        return batch, gptr, graph, num_graphs

    def _trunk(self, z: Tensor, graph: ops.RadiusGraph) -> Tensor:
        """embedding -> interactions with residual (schnet_no_sum.py:159-164 == :207-212)."""
        h = ops.embedding(z, self.embedding.weight, self.embedding.padding_idx)
        if self.fused_filter and ops.filter_fused_supported(self.num_gaussians, self.num_filters):
            rbf = (self.distance_expansion.offset, self.distance_expansion.coeff)          # expanded inside the fused kernel
        else:
            rbf = self.distance_expansion(graph)
        for interaction in self.interactions:
            h = interaction(h, graph, rbf)
        return h

    def _head(self, h: Tensor, lin1: Linear, lin2: Linear) -> Tensor:
        return ops.mlp2_outact(h, lin1.weight, lin1.bias, lin2.weight, lin2.bias)      # lin1 -> lin2 -> act: one launch at node level

    # ---------------------------------------------------------------------------------------------- reference API
    def forward(self, z: Tensor, pos: Tensor, batch: OptTensor = None, data_batch=None, num_graphs: Optional[int] = None) -> Tensor:
        """Stage-1 path (schnet_no_sum.py:144-188): per-conformer embedding [G, hidden/2] (or per-atom if use_readout=False)."""
        batch, gptr, graph, G = self._graphs(z, pos, batch, num_graphs)
        h = self._head(self._trunk(z, graph), self.lin1, self.lin2)        # :176-178
        return ops.segment_sum(h, gptr, G) if self.use_readout else h      # :182-186

    def forward_3d_bary(self, z: Tensor, pos: Tensor, batch: OptTensor = None, data_batch=None, num_graphs: Optional[int] = None,
                        _graph=None):
        """Two per-atom embeddings from the shared trunk (schnet_no_sum.py:190-232)."""
        if _graph is None:
            _, _, graph, _ = self._graphs(z, pos, batch, num_graphs)
        else:
            graph = _graph
        h_shared = self._trunk(z, graph)
        h = self._head(h_shared, self.lin1, self.lin2)                     # :225-227
        h_bary = self._head(h_shared, self.lin1_bary, self.lin2_bary)      # :229-231
        return h, h_bary

    def _compute_barycenter(self, node_feature: Tensor, edge_index, batch: Tensor, batch_size: int, num_conformers: int,
                            max_nodes: Optional[int] = None, _want_node_out: bool = True):
        """schnet_no_sum.py:234-315.  `edge_index` may be the reference's int64 [2,E] tensor or an ops.RadiusGraph.
        `_want_node_out=False` (forward_w_barycenter, which discards it like the reference does at :346) skips the readout of the
        node features: the first return value is then None."""
        K = num_conformers
        G = batch_size * K
        if isinstance(edge_index, ops.RadiusGraph):
            graph = edge_index
        else:
            graph = _graph_from_edge_index(edge_index, batch, G)
        if max_nodes is None:
            max_nodes = getattr(graph, "max_nodes_hint", None)
        if max_nodes is None:
            gp = graph.graph_ptr
            max_nodes = int((gp[1:] - gp[:-1]).max().item())                # host sync (to_dense_batch does the same, :242)
        Ys, _ = ops.fgw_densify(node_feature, graph, max_nodes, self.FEATURE_SHIFT, adjacency=False)      # (to_dense_adj, :249-252, is read by the solver from the graph's ragged lists)       # :242-252 with :41-87
        N, d = max_nodes, node_feature.shape[1]
        Y, C, T, info, errs = ops.fgw_barycenter_batched(Ys.view(batch_size, K, N, d), None, adjacency=graph)   # :259-306
        self.last_fgw = dict(Y=Y, C=C, T=T, info=info, errs=errs, Ys=Ys)
        F_bary_batch = ops.fgw_readout(Y, K, self.READOUT_MODE)                            # :308-312
        node_out = ops.segment_sum(node_feature, graph.graph_ptr, G) if _want_node_out else None    # :314
        return node_out, F_bary_batch

    def forward_w_barycenter(self, z: Tensor, pos: Tensor, num_conformers: int, batch: OptTensor = None, data_batch=None,
                             max_iter: int = 100, epsilon: float = 0.1, num_graphs: Optional[int] = None,
                             max_nodes: Optional[int] = None):
        """schnet_no_sum.py:317-354.  `max_iter` / `epsilon` are accepted and ignored exactly like the reference
        (the FGW hyper-parameters are the literals of :281-306)."""
        batch, gptr, graph, G = self._graphs(z, pos, batch, num_graphs)
        h_3d, h_bary = self.forward_3d_bary(z, pos, batch, _graph=graph)                   # :341
        batch_size = G // num_conformers                                                  # :345
        _, h_bary = self._compute_barycenter(h_bary, graph, batch, batch_size, num_conformers, max_nodes=max_nodes,
                                             _want_node_out=False)                            # :346-352 (its first result is dropped there too)
        h_3d = ops.segment_sum(h_3d, gptr, G)                                              # :353
        return h_3d, h_bary


def _graph_from_edge_index(edge_index: Tensor, batch: Tensor, num_graphs: int) -> ops.RadiusGraph:
    """Compatibility path for callers that hand `_compute_barycenter` the reference's int64 edge_index: rebuild the
    CSR-by-target view with torch index ops (host-side plumbing, not the hot path)."""
    g = object.__new__(ops.RadiusGraph)
    n = batch.shape[0]
    src, tgt = edge_index[0], edge_index[1]
    order = torch.argsort(tgt * n + src)
    src, tgt = src[order], tgt[order]
    deg = torch.bincount(tgt, minlength=n)
    rowptr = torch.zeros(n + 1, dtype=torch.int32, device=batch.device)
    rowptr[1:] = deg.cumsum(0).to(torch.int32)
    g.num_atoms, g.num_graphs = n, num_graphs
    g.graph_ptr = ops.graph_ptr_from_batch(batch, num_graphs)
    g.rowptr, g.col, g.tgt = rowptr, src.to(torch.int32).contiguous(), tgt.to(torch.int32).contiguous()
    g.dist = None
    g.max_edges = max(1, int(src.shape[0]))
    g.num_edges_dev = rowptr[n:]
    g._num_edges = int(src.shape[0])
    g._t_rowptr = g._t_eid = None
    g.pid = None
    g._deg = torch.empty(n + 1, dtype=torch.int32, device=batch.device)
    g.cutoff = g.cap = g.loop = None
    return g
