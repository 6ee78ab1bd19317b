"""TEST INFRASTRUCTURE — CPU restatement of the reference's ConAN SchNet backbone and barycenter glue.

Follows conan_fgw/src/model/graph_embeddings/schnet_no_sum.py (cited per method) on top of oracle/pyg_semantics.py
(the PyG-2.3.0 trunk, parity unpinned) and oracle/fgw.py (the C restatement of the FGW solver, pinned).
dtype-agnostic: `.double()` gives the "ref64" yard-stick.  The wiring of this class is pinned by
tests/golden/schnet_ref_*.npz, produced by running the reference's own class (tests/golden/make_model_golden.py).
"""
from __future__ import annotations

import numpy as np
import torch
from torch import Tensor

from . import fgw as ofgw
from .pyg_semantics import SchNet, to_dense_adj, to_dense_batch


class _FGWBarycenterFn(torch.autograd.Function):
    """Forward: C oracle.  Backward: dYs = lam_s T_s^T diag(1/p) dY with the final couplings as constants
    (barycenter.py:120 solves the couplings under no_grad)."""

    @staticmethod
    def forward(ctx, Ys: Tensor, Cs: Tensor, shift_unused: float = 0.0):
        dt = np.float64 if Ys.dtype == torch.float64 else np.float32
        r = ofgw.fgw_barycenter(Ys.detach().numpy(), Cs.detach().numpy(), dtype=dt)
        ctx.T = r["T"]
        ctx.dt = dt
        ctx.info = r
        return torch.from_numpy(r["Y"]), torch.from_numpy(r["C"])

    @staticmethod
    def backward(ctx, dY, dC):
        dYs = ofgw.fgw_barycenter_bwd(ctx.T, dY.contiguous().numpy(), dtype=ctx.dt)
        return torch.from_numpy(dYs), None, None


def normalize_tensor(t: Tensor, a: float, b: float) -> Tensor:
    """barycenter.py:393-399."""
    mn, mx = t.min(), t.max()
    return a + (t - mn) * (b - a) / (mx - mn)


class SchNetNoSumOracle(SchNet):
    """schnet_no_sum.py:90-354 with use_covalent=False."""

    FEATURE_SHIFT = 0.5     # schnet_no_sum.py:59

    def __init__(self, hidden_channels=128, num_filters=128, num_interactions=6, num_gaussians=50, cutoff=10.0,
                 max_num_neighbors=32, readout="add"):
        super().__init__(hidden_channels, num_filters, num_interactions, num_gaussians, cutoff, None,
                         max_num_neighbors, readout)
        # schnet_no_sum.py:126-130
        self.lin1_bary = torch.nn.Linear(hidden_channels, hidden_channels // 2)
        self.lin2_bary = torch.nn.Linear(hidden_channels // 2, hidden_channels // 2)
        self.lin2 = torch.nn.Linear(hidden_channels // 2, hidden_channels // 2)
        self.last_fgw_info = None

    def trunk(self, z, pos, batch):
        """schnet_no_sum.py:159-164 == :207-212."""
        h = self.embedding(z)
        edge_index, edge_weight = self.interaction_graph(pos, batch)
        edge_attr = self.distance_expansion(edge_weight)
        for interaction in self.interactions:
            h = h + interaction(h, edge_index, edge_weight, edge_attr)
        return h, edge_index, edge_weight

    def forward(self, z, pos, batch=None):
        """schnet_no_sum.py:144-188 (use_readout=True)."""
        batch = torch.zeros_like(z) if batch is None else batch
        h, _, _ = self.trunk(z, pos, batch)
        h = self.act(self.lin2(self.lin1(h)))                      # :176-178
        return self.readout(h, batch, dim=0)                       # :183

    def forward_3d_bary(self, z, pos, batch=None):
        """schnet_no_sum.py:190-232."""
        batch = torch.zeros_like(z) if batch is None else batch
        h_shared, _, _ = self.trunk(z, pos, batch)
        h = self.act(self.lin2(self.lin1(h_shared)))               # :225-227
        h_bary = self.act(self.lin2_bary(self.lin1_bary(h_shared)))   # :229-231
        return h, h_bary

    def _compute_barycenter(self, node_feature, edge_index, batch, batch_size, num_conformers):
        """schnet_no_sum.py:234-315."""
        K = num_conformers
        dense, _mask = to_dense_batch(node_feature, batch)         # :242-244 (mask ignored afterwards)
        adj = to_dense_adj(edge_index, batch).to(node_feature.dtype)   # :249-252
        out = torch.zeros(batch_size * K, node_feature.shape[1], dtype=node_feature.dtype)
        infos = []
        rows = []
        for b in range(batch_size):                                # :259
            slab = dense[b * K:(b + 1) * K] + self.FEATURE_SHIFT   # :59
            Ys = torch.stack([normalize_tensor(s, 0.1, 2.0) for s in slab])   # :66
            Cs = adj[b * K:(b + 1) * K]
            Y, _C = _FGWBarycenterFn.apply(Ys, Cs)                 # :281-306 (production literals = ofgw.PROD)
            infos.append((Y.detach(), _C.detach()))
            F_bary = self.post_barycenter(Y)
            rows.append(F_bary.sum(0, keepdim=True).repeat(K, 1))  # :308-312
        out = torch.cat(rows, 0)
        self.last_fgw_info = infos
        return self.readout(node_feature, batch, dim=0), out       # :314-315

    def post_barycenter(self, F_bary):
        return F_bary

    def forward_w_barycenter(self, z, pos, num_conformers, batch=None):
        """schnet_no_sum.py:317-354."""
        batch = torch.zeros_like(z) if batch is None else batch
        h_3d, h_bary = self.forward_3d_bary(z, pos, batch)         # :341
        edge_index, _ = self.interaction_graph(pos, batch)         # :342
        batch_size = int(len(batch.unique()) / num_conformers)     # :345
        _, h_bary = self._compute_barycenter(h_bary, edge_index, batch, batch_size, num_conformers)
        h_3d = self.readout(h_3d, batch, dim=0)                    # :353
        return h_3d, h_bary
