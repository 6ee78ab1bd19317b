"""Regression head of the reference's stage-2 model without the covalent (GAT) branch.

`EmbeddingsWithGATAggregationBaryCenter.forward` (conan_fgw/src/model/schnet_based_models.py:135-173) combines
x = Lin3d(h_3d) + Lin_cov(GAT(...)) + agg_weight * Lin_bary(h_bary), averages over the K conformers and applies the final
linear.  The GAT branch is outside this build's scope; this head reproduces the rest so that the
benchmark and the training tests have a loss to differentiate.  All linears run on conan_linear_fwd.
"""
from __future__ import annotations

import torch
from torch import Tensor
from torch.nn import Linear

from . import ops


class ConformerAggregationHead(torch.nn.Module):
    def __init__(self, feat_dim: int = 64, agg_weight: float = 0.2):
        super().__init__()
        self.agg_weight = agg_weight                         # config `agg-weight`, config_parser.py default 0.2
        self.lin_3d = Linear(feat_dim, feat_dim)             # schnet_based_models.py:94-110
        self.lin_bary = Linear(feat_dim, feat_dim)
        self.out = Linear(feat_dim, 1)

    def forward(self, h_3d: Tensor, h_bary: Tensor, num_conformers: int) -> Tensor:
        G, d = h_3d.shape
        B = G // num_conformers
        x3 = ops.linear(h_3d, self.lin_3d.weight, self.lin_3d.bias)
        xb = ops.linear(h_bary, self.lin_bary.weight, self.lin_bary.bias)
        x = x3 + self.agg_weight * xb                        # :170 (x_cov omitted)
        x = x.view(B, num_conformers, d).mean(dim=1)         # :171
        return ops.linear(x.contiguous(), self.out.weight, self.out.bias)    # :172  -> [B,1]
