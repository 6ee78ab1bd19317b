"""CPU restatement of the stage-2 regression model — TEST INFRASTRUCTURE ONLY.

`Stage2Oracle.forward` follows EmbeddingsWithGATAggregationBaryCenter.forward
(conan_fgw/src/model/schnet_based_models.py:135-173): x = Lin3d(h_3d) + Lin_cov(GAT(...)) + agg_weight * Lin_bary(h_bary),
MeanAggregation over the K conformers (common.py:404), `molecular_regression_lin` = Linear(d, 1) (build_mlp, :17-28).
The Lightning shell is not imported (pytorch_lightning / torchmetrics are absent); the pin is compositional: the backbone
wiring is pinned by the goldens of tests/golden/make_model_golden.py, the GAT branch is unpinned (oracle/gat.py header).
Sub-module names equal the reference's, so the product's state_dict loads strictly.
"""
from __future__ import annotations

import torch
from torch import nn

from .gat import GATBasedOracle
from .schnet import SchNetNoSumOracle


class Stage2Oracle(nn.Module):
    def __init__(self, num_conformers: int, agg_weight: float = 0.2, gat_in_channels: int = 9, model_name: str = "schnet"):
        super().__init__()
        self.num_conformers, self.agg_weight = num_conformers, agg_weight
        if model_name == "visnet":                                  # EquivModelsHolder.get_model("visnet", feat_dim=128), common.py:542-546
            from .visnet import ViSNetOracle
            self.node_embeddings_model = ViSNetOracle(128)
        else:
            self.node_embeddings_model = SchNetNoSumOracle(128, 128, 3)
        self.gat_embeddings_model = GATBasedOracle(64, 3, gat_in_channels)
        self.transformation_matrix_3d = nn.Linear(64, 64)
        self.transformation_matrix_bary = nn.Linear(64, 64)
        self.transformation_matrix_cov = nn.Linear(64, 64)
        self.molecular_regression_lin = nn.Linear(64, 1)

    def forward(self, z, pos, node_index, x, edge_index, edge_attr):
        K = self.num_conformers
        x_3d, x_bary = self.node_embeddings_model.forward_w_barycenter(z, pos, K, node_index)
        x_bary = self.transformation_matrix_bary(x_bary)
        x_3d = self.transformation_matrix_3d(x_3d)
        x_cov = self.transformation_matrix_cov(self.gat_embeddings_model(x, edge_index, edge_attr, node_index))
        h = x_3d + x_cov + self.agg_weight * x_bary
        h = h.view(h.shape[0] // K, K, -1).mean(1)
        return self.molecular_regression_lin(h)


class Stage1Oracle(Stage2Oracle):
    """EmbeddingsWithGATAggregation.forward (schnet_based_models.py:231-244): no barycenter branch."""

    def forward(self, z, pos, node_index, x, edge_index, edge_attr):
        K = self.num_conformers
        x_3d = self.transformation_matrix_3d(self.node_embeddings_model(z, pos, node_index))
        x_cov = self.transformation_matrix_cov(self.gat_embeddings_model(x, edge_index, edge_attr, node_index))
        h = x_3d + x_cov
        h = h.view(h.shape[0] // K, K, -1).mean(1)
        return self.molecular_regression_lin(h)


class Stage2ClassificationOracle(nn.Module):
    """EmbeddingsWithGATAggregationClassificationBaryCenter.forward (schnet_based_models.py:350-369): SchNet 512 / 256 filters /
    10 gaussians (common.py:513-522), 256-wide branches, build_mlp_class(is_complex=True) (:31-45), sigmoid."""

    def __init__(self, num_conformers: int, agg_weight: float = 0.2, gat_in_channels: int = 9, model_name: str = "schnet", feat_dim: int = 512):
        super().__init__()
        self.num_conformers, self.agg_weight = num_conformers, agg_weight
        if model_name == "visnet":                                  # get_model("visnet", feat_dim=...), common.py:444-446 -> :542-546
            from .visnet import ViSNetOracle
            self.node_embeddings_model = ViSNetOracle(feat_dim)
        else:
            self.node_embeddings_model = SchNetNoSumOracle(feat_dim, 256, 3, num_gaussians=10, cutoff=10.0)
        c = feat_dim // 2
        self.gat_embeddings_model = GATBasedOracle(c, 3, gat_in_channels)
        self.transformation_matrix_3d = nn.Linear(c, c)
        self.transformation_matrix_cov = nn.Linear(c, c)
        self.transformation_matrix_bary = nn.Linear(c, c)
        self.molecular_regression_lin = nn.Sequential(nn.Linear(c, c), nn.ReLU(), nn.Linear(c, c // 2), nn.ReLU(), nn.Linear(c // 2, 1))
        self.self_attention = nn.ModuleDict({"query": nn.Linear(c, c), "key": nn.Linear(c, c), "value": nn.Linear(c, c)})

    def forward(self, z, pos, node_index, x, edge_index, edge_attr):
        K = self.num_conformers
        x_3d, x_bary = self.node_embeddings_model.forward_w_barycenter(z, pos, K, node_index)
        x_3d = self.transformation_matrix_3d(x_3d)
        x_bary = self.transformation_matrix_bary(x_bary)
        x_cov = self.transformation_matrix_cov(self.gat_embeddings_model(x, edge_index, edge_attr, node_index))
        h = x_3d + x_cov + self.agg_weight * x_bary
        h = h.view(h.shape[0] // K, K, -1).mean(1)
        return torch.sigmoid(self.molecular_regression_lin(h))
