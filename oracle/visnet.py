"""TEST INFRASTRUCTURE — CPU restatement of the reference's ViSNet backbone and its ConAN wrapper.

Restates, in plain dtype-agnostic torch, the arithmetic of the reference's vendored ViSNet
(conan_fgw/src/model/graph_embeddings/torch_geometric_visnet.py, cited per class) with `vertex=False`, `lmax=1`,
`vecnorm_type=None`, `derivative=False` (the only configuration ConAN instantiates: visnet.py:83-91, common.py:542-546), and
the wrapper conan_fgw/src/model/graph_embeddings/visnet.py:82-288.  Module / parameter names equal the reference's so
that its state_dict loads unchanged.  Pinned by tests/golden/visnet_ref_*.npz, produced by running the reference's own
classes over the PyG stand-in (tests/golden/make_model_golden.py).
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F
from torch import Tensor
from torch.nn import Embedding, LayerNorm, Linear

from . import fgw as ofgw
from .pyg_semantics import RadiusInteractionGraph, radius_graph, scatter, to_dense_adj, to_dense_batch
from .schnet import _FGWBarycenterFn, normalize_tensor


def cosine_cutoff(d: Tensor, cutoff: float) -> Tensor:
    """torch_geometric_visnet.py:33-46."""
    return 0.5 * ((d * math.pi / cutoff).cos() + 1.0) * (d < cutoff).to(d.dtype)


class ExpNormalSmearing(torch.nn.Module):
    """torch_geometric_visnet.py:49-111 (trainable=False => buffers)."""

    def __init__(self, cutoff=5.0, num_rbf=32):
        super().__init__()
        self.cutoff, self.alpha = cutoff, 5.0 / cutoff
        start = torch.exp(torch.tensor(-cutoff))
        self.register_buffer("means", torch.linspace(start, 1, num_rbf))
        self.register_buffer("betas", torch.tensor([(2 / num_rbf * (1 - start)) ** -2] * num_rbf))

    def forward(self, dist):
        dist = dist.unsqueeze(-1)
        return cosine_cutoff(dist, self.cutoff) * (-self.betas.to(dist.dtype) * ((self.alpha * (-dist)).exp() - self.means.to(dist.dtype)) ** 2).exp()


class VecLayerNorm(torch.nn.Module):
    """torch_geometric_visnet.py:192-282 with norm_type=None: vec * weight (weight is a buffer of ones)."""

    def __init__(self, hidden):
        super().__init__()
        self.register_buffer("weight", torch.ones(hidden))

    def forward(self, vec):
        return vec * self.weight.to(vec.dtype).unsqueeze(0).unsqueeze(0)


class NeighborEmbedding(torch.nn.Module):
    """torch_geometric_visnet.py:350-423."""

    def __init__(self, hidden, num_rbf, cutoff, max_z=100):
        super().__init__()
        self.embedding = Embedding(max_z, hidden)
        self.distance_proj = Linear(num_rbf, hidden)
        self.combine = Linear(hidden * 2, hidden)
        self.cutoff = cutoff

    def forward(self, z, x, edge_index, edge_weight, edge_attr):
        mask = edge_index[0] != edge_index[1]                               # :408-412 self loops removed here only
        ei, ew, ea = edge_index[:, mask], edge_weight[mask], edge_attr[mask]
        W = self.distance_proj(ea) * cosine_cutoff(ew, self.cutoff).view(-1, 1)
        msg = self.embedding(z)[ei[0]] * W
        x_nb = scatter(msg, ei[1], dim=0, dim_size=x.shape[0])
        return self.combine(torch.cat([x, x_nb], dim=1))


class EdgeEmbedding(torch.nn.Module):
    """torch_geometric_visnet.py:426-465."""

    def __init__(self, num_rbf, hidden):
        super().__init__()
        self.edge_proj = Linear(num_rbf, hidden)

    def forward(self, edge_index, edge_attr, x):
        return (x[edge_index[1]] + x[edge_index[0]]) * self.edge_proj(edge_attr)


class ViS_MP(torch.nn.Module):
    """torch_geometric_visnet.py:468-673."""

    def __init__(self, num_heads, hidden, cutoff, last_layer=False):
        super().__init__()
        self.num_heads, self.hidden, self.head_dim, self.last_layer, self.cutoff = num_heads, hidden, hidden // num_heads, last_layer, cutoff
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

    @staticmethod
    def vector_rejection(vec, d_ij):
        return vec - (vec * d_ij.unsqueeze(2)).sum(dim=1, keepdim=True) * d_ij.unsqueeze(2)      # :552-555

    def forward(self, x, vec, edge_index, r_ij, f_ij, d_ij):
        H, nh, hd = self.hidden, self.num_heads, self.head_dim
        x = self.layernorm(x)
        vec = self.vec_layernorm(vec)
        q = self.q_proj(x).reshape(-1, nh, hd); k = self.k_proj(x).reshape(-1, nh, hd); v = self.v_proj(x).reshape(-1, nh, hd)
        dk = F.silu(self.dk_proj(f_ij)).reshape(-1, nh, hd)
        dv = F.silu(self.dv_proj(f_ij)).reshape(-1, nh, hd)
        vec1, vec2, vec3 = torch.split(self.vec_proj(vec), H, dim=-1)
        vec_dot = (vec1 * vec2).sum(dim=1)
        src, tgt = edge_index[0], edge_index[1]
        # message :632-653
        attn = (q[tgt] * k[src] * dk).sum(dim=-1)
        attn = F.silu(attn) * cosine_cutoff(r_ij, self.cutoff).unsqueeze(1)
        v_j = (v[src] * dv * attn.unsqueeze(2)).view(-1, H)
        s1, s2 = torch.split(F.silu(self.s_proj(v_j)), H, dim=1)
        vec_j = vec[src] * s1.unsqueeze(1) + s2.unsqueeze(1) * d_ij.unsqueeze(2)
        # aggregate :663-673
        x_agg = scatter(v_j, tgt, dim=0, dim_size=x.shape[0])
        vec_agg = scatter(vec_j, tgt, dim=0, dim_size=x.shape[0])
        o1, o2, o3 = torch.split(self.o_proj(x_agg), H, dim=1)
        dx = vec_dot * o2 + o3
        dvec = vec3 * o1.unsqueeze(1) + vec_agg
        if self.last_layer:
            return dx, dvec, None
        # edge_update :655-661
        w1 = self.vector_rejection(self.w_trg_proj(vec[tgt]), d_ij)
        w2 = self.vector_rejection(self.w_src_proj(vec[src]), -d_ij)
        df = F.silu(self.f_proj(f_ij)) * (w1 * w2).sum(dim=1)
        return dx, dvec, df


class ViSNetBlock(torch.nn.Module):
    """torch_geometric_visnet.py:741-886."""

    def __init__(self, num_heads=8, num_layers=6, hidden=128, num_rbf=32, max_z=100, cutoff=5.0, max_num_neighbors=32):
        super().__init__()
        self.cutoff, self.max_num_neighbors = cutoff, max_num_neighbors
        self.embedding = Embedding(max_z, hidden)
        self.distance_expansion = ExpNormalSmearing(cutoff, num_rbf)
        self.neighbor_embedding = NeighborEmbedding(hidden, num_rbf, cutoff, max_z)
        self.edge_embedding = EdgeEmbedding(num_rbf, hidden)
        self.vis_mp_layers = torch.nn.ModuleList([ViS_MP(num_heads, hidden, cutoff, last_layer=(i == num_layers - 1)) for i in range(num_layers)])
        self.out_norm = LayerNorm(hidden)
        self.vec_out_norm = VecLayerNorm(hidden)

    def forward(self, z, pos, batch):
        x = self.embedding(z)
        # Distance :313-347 (self loops kept, cap includes the atom itself)
        edge_index = radius_graph(pos, r=self.cutoff, batch=batch, loop=True, max_num_neighbors=self.max_num_neighbors)
        edge_vec = pos[edge_index[0]] - pos[edge_index[1]]
        mask = edge_index[0] != edge_index[1]
        edge_weight = torch.zeros(edge_vec.shape[0], dtype=pos.dtype)
        edge_weight[mask] = torch.norm(edge_vec[mask], dim=-1)
        edge_attr = self.distance_expansion(edge_weight)
        edge_vec = edge_vec.clone()
        edge_vec[mask] = edge_vec[mask] / torch.norm(edge_vec[mask], dim=1).unsqueeze(1)      # :864-866 ; Sphere(lmax=1) = identity
        x = self.neighbor_embedding(z, x, edge_index, edge_weight, edge_attr)
        vec = torch.zeros(x.shape[0], 3, x.shape[1], dtype=x.dtype)
        edge_attr = self.edge_embedding(edge_index, edge_attr, x)
        for attn in self.vis_mp_layers[:-1]:
            dx, dvec, df = attn(x, vec, edge_index, edge_weight, edge_attr, edge_vec)
            x, vec, edge_attr = x + dx, vec + dvec, edge_attr + df
        dx, dvec, _ = self.vis_mp_layers[-1](x, vec, edge_index, edge_weight, edge_attr, edge_vec)
        x, vec = x + dx, vec + dvec
        return self.out_norm(x), self.vec_out_norm(vec)


class GatedEquivariantBlock(torch.nn.Module):
    """torch_geometric_visnet.py:889-960."""

    def __init__(self, hidden, out, scalar_activation=False):
        super().__init__()
        self.out = out
        self.vec1_proj = Linear(hidden, hidden, bias=False)
        self.vec2_proj = Linear(hidden, out, bias=False)
        self.update_net = torch.nn.Sequential(Linear(hidden * 2, hidden), torch.nn.SiLU(), Linear(hidden, out * 2))
        self.scalar_activation = scalar_activation

    def forward(self, x, v):
        vec1 = torch.norm(self.vec1_proj(v), dim=-2)
        vec2 = self.vec2_proj(v)
        x, g = torch.split(self.update_net(torch.cat([x, vec1], dim=-1)), self.out, dim=-1)
        v = g.unsqueeze(1) * vec2
        return (F.silu(x) if self.scalar_activation else x), v


class EquivariantScalar(torch.nn.Module):
    """torch_geometric_visnet.py:963-1014."""

    def __init__(self, hidden, out):
        super().__init__()
        self.output_network = torch.nn.ModuleList([GatedEquivariantBlock(hidden, hidden // 2, True), GatedEquivariantBlock(hidden // 2, out, False)])

    def pre_reduce(self, x, v):
        for layer in self.output_network:
            x, v = layer(x, v)
        return x + v.sum() * 0


class Atomref(torch.nn.Module):
    """torch_geometric_visnet.py:1017-1058."""

    def __init__(self, max_z=100):
        super().__init__()
        self.register_buffer("initial_atomref", torch.zeros(max_z, 1))
        self.atomref = Embedding(max_z, 1)
        self.atomref.weight.data.copy_(self.initial_atomref)

    def forward(self, x, z):
        return x + self.atomref(z)


class ViSNetOracle(torch.nn.Module):
    """visnet.py:82-288 on top of torch_geometric_visnet.py:1061-1229 (ViSNet.__init__)."""

    FEATURE_SHIFT = 1.0          # visnet.py:50

    def __init__(self, hidden_channels=128, cutoff=5.0):
        super().__init__()
        self.hidden_channels = hidden_channels
        self.representation_model = ViSNetBlock(hidden=hidden_channels, cutoff=5.0)          # the block keeps ViSNet's default cutoff
        self.output_model = EquivariantScalar(hidden_channels, hidden_channels // 2)
        self.prior_model = Atomref()
        self.output_model_bary = EquivariantScalar(hidden_channels, hidden_channels // 2)
        self.prior_model_bary = Atomref()
        self.register_buffer("mean", torch.tensor(0.0))
        self.register_buffer("std", torch.tensor(1.0))
        self.interaction_graph = RadiusInteractionGraph(cutoff, max_num_neighbors=32)          # visnet.py:90
        self.last_fgw = None

    def forward_3d_bary(self, z, pos, batch):
        """visnet.py:124-158."""
        xs, vs = self.representation_model(z, pos, batch)
        x = self.prior_model(self.output_model.pre_reduce(xs, vs) * self.std.to(xs.dtype), z)
        xb = self.prior_model_bary(self.output_model_bary.pre_reduce(xs, vs) * self.std.to(xs.dtype), z)
        return x, xb

    def forward(self, z, pos, batch):
        """visnet.py:93-122."""
        xs, vs = self.representation_model(z, pos, batch)
        x = self.prior_model(self.output_model.pre_reduce(xs, vs) * self.std.to(xs.dtype), z)
        return scatter(x, batch, dim=0, dim_size=int(batch.max()) + 1)

    def _compute_barycenter(self, node_feature, edge_index, batch, batch_size, K):
        """visnet.py:160-249."""
        dense, _ = to_dense_batch(node_feature, batch)
        adj = to_dense_adj(edge_index, batch).to(node_feature.dtype)
        rows, infos = [], []
        for b in range(batch_size):
            slab = dense[b * K:(b + 1) * K] + self.FEATURE_SHIFT
            Ys = torch.stack([normalize_tensor(s, 0.1, 2.0) for s in slab])
            Y, C = _FGWBarycenterFn.apply(Ys, adj[b * K:(b + 1) * K])
            infos.append((Y.detach(), C.detach()))
            if torch.isnan(Y).any():                                       # :233-238
                Y = torch.zeros_like(Y)
            Y = Y / torch.linalg.norm(Y, dim=0)                            # :240-241
            rows.append(Y.sum(0, keepdim=True).repeat(K, 1))               # :242-246
        self.last_fgw = infos
        G = batch_size * K
        return scatter(node_feature, batch, dim=0, dim_size=G), torch.cat(rows, 0)

    def forward_w_barycenter(self, z, pos, num_conformers, batch):
        """visnet.py:251-288."""
        h_3d, h_bary = self.forward_3d_bary(z, pos, batch)
        edge_index, _ = self.interaction_graph(pos, batch)
        batch_size = int(len(batch.unique()) / num_conformers)
        _, h_bary = self._compute_barycenter(h_bary, edge_index, batch, batch_size, num_conformers)
        G = batch_size * num_conformers
        return scatter(h_3d, batch, dim=0, dim_size=G), h_bary
