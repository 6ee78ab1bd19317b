"""The reference's conformer-aggregation models on top of the MI355X backbones (without their Lightning shell).

`EmbeddingsWithGATAggregationBaryCenter.forward` (conan_fgw/src/model/schnet_based_models.py:135-173) combines
x = Lin3d(h_3d) + Lin_cov(GAT(...)) + agg_weight * Lin_bary(h_bary), averages over the K conformers and applies the final
linear; the stage-1 and classification twins (:176-244, :308-369) are the same assembly.  Backbones come from
`EquivModelsHolder.get_model` exactly like the reference (common.py:400-402,444-446).  All linears run on conan_linear_fwd.
"""
from __future__ import annotations

import torch
from torch import Tensor
from torch.nn import Linear

from . import ops


class EquivModelsHolder:
    """`EquivModelsHolder.get_model(name, device, **kwargs)` of the reference (conan_fgw/src/model/common.py:469-546) for the
    backbones this backend implements: the object the Lightning models store as `node_embeddings_model` /
    `gat_embeddings_model`.  Same names, same keyword meaning (`feat_dim`, optional `cutoff` selecting the classification
    SchNet), same hyper-parameter literals."""

    @staticmethod
    def get_model(name: str, device, **kwargs):
        if name == "schnet":
            from .schnet import SchNetNoSum
            if "cutoff" in kwargs:                                  # common.py:513-522 (classification: 256 filters, 10 gaussians)
                return SchNetNoSum(device, hidden_channels=kwargs.get("feat_dim"), use_covalent=False, cutoff=kwargs.get("cutoff"),
                                   num_gaussians=10, num_filters=256, num_interactions=3)
            return SchNetNoSum(device, hidden_channels=kwargs.get("feat_dim"), use_covalent=False, num_interactions=3)   # :524-529
        if name == "gat":
            from .gat import GATBased
            return GATBased(out_channels=kwargs.get("feat_dim") // 2)                                                   # :534-537
        if name == "visnet":
            from .visnet import ViSNet
            return ViSNet(device, hidden_channels=kwargs.get("feat_dim"))                                               # :542-546
        raise ValueError(f"get_model({name!r}): not a backbone of the MI355X hot path (schnet, visnet, gat); DimeNet / ESAN / "
                         "schnet_covalent variants are not reachable from train_val.py and are not built here")


def get_model(name: str, device, **kwargs):
    """Module-level alias of `EquivModelsHolder.get_model` (`conan_fgw_amd.get_model`)."""
    return EquivModelsHolder.get_model(name, device, **kwargs)


GAT_ON_SIDE_STREAM = True      # stage-2 regression model: the covalent branch beside the backbone (tools/ab_step_switch.py compares)


class EmbeddingsWithGATAggregationBaryCenter(torch.nn.Module):
    """The stage-2 regression model of the reference without its Lightning shell
    (`EmbeddingsWithGATAggregationBaryCenter`, conan_fgw/src/model/schnet_based_models.py:83-173 on top of `EquivAggregation`,
    common.py:388-423): 3-D backbone with the FGW barycenter branch + covalent GAT branch + conformer mean + regression.

    Sub-module names are the reference's (`node_embeddings_model`, `gat_embeddings_model`, `transformation_matrix_3d`,
    `transformation_matrix_bary`, `transformation_matrix_cov`, `molecular_regression_lin`), so the weights of a reference
    checkpoint load by name.  `forward(batch, conformers_index, node_index)` keeps the reference's argument meaning: `batch`
    carries `z, pos, x, edge_index, edge_attr, batch`; `node_index` = conformer-graph id per atom of the 3-D graphs;
    `conformers_index` = molecule id per conformer graph (`create_aggregation_index`, common.py:414-423).
    """

    def __init__(self, num_conformers: int, device=None, model_name: str = "schnet", agg_weight: float = 0.2,
                 max_iter: int = 100, epsilon: float = 0.1, gat_in_channels: int = 9):
        super().__init__()
        from .gat import GATBased
        self.num_conformers = num_conformers
        device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.node_embeddings_model = EquivModelsHolder.get_model(model_name, device, feat_dim=128)    # common.py:400-402
        out_channels = self.node_embeddings_model.hidden_channels // 2
        self.gat_embeddings_model = (EquivModelsHolder.get_model("gat", device, feat_dim=128) if gat_in_channels == 9
                                     else GATBased(out_channels=128 // 2, in_channels=gat_in_channels))      # schnet_based_models.py:96
        self.transformation_matrix_3d = Linear(out_channels, out_channels)
        self.transformation_matrix_bary = Linear(out_channels, out_channels)
        self.transformation_matrix_cov = Linear(out_channels, out_channels)
        self.molecular_regression_lin = Linear(out_channels, 1)          # build_mlp(out_channels), is_complex=False
        self.numItermax, self.epsilon, self.agg_weight = max_iter, epsilon, agg_weight

    def _side_stream(self, device) -> "torch.cuda.Stream":
        s = getattr(self, "_gat_stream", None)
        if s is None or s.device != device:
            s = torch.cuda.Stream(device=device)
            object.__setattr__(self, "_gat_stream", s)       # not a module attribute: streams are not part of the state
        return s

    def forward_dummy(self, batch, conformers_index, node_index):
        """`load_dummy` (conan_fgw/src/model/utils.py:23-33) calls this once on a CPU mini-batch before the DDP wrap, to
        materialise PyG's lazily-shaped GATConv parameters (schnet_based_models.py:112-133).  Every parameter of this backend
        has its final shape at construction, so there is nothing to materialise: a no-op returning None like the reference's
        method (which also discards its activations).  It must not raise on CPU inputs — it is the one call the harness makes
        before the model is moved to the GPU."""
        return None

    def create_aggregation_index(self, num_graphs, device=None) -> Tensor:
        """Molecule id of every conformer graph.  Accepts the reference's argument — the collated batch, from which the
        reference counts conformer graphs as len(batch.smiles) (common.py:414-423) — or the number of graphs directly."""
        if not isinstance(num_graphs, int):
            batch = num_graphs
            num_graphs = len(batch.smiles) if hasattr(batch, "smiles") else int(batch.num_graphs)
            if device is None:
                device = batch.z.device
        return self._aggregation_index(num_graphs, device)

    def _aggregation_index(self, num_graphs: int, device) -> Tensor:
        """Molecule id of every conformer graph: [0]*K + [1]*K + ... (common.py:414-423, built there from len(batch.smiles))."""
        return torch.arange(num_graphs // self.num_conformers, device=device).repeat_interleave(self.num_conformers)

    def forward(self, batch, conformers_index: Tensor, node_index: Tensor, num_graphs: int = None, max_nodes: int = None) -> Tensor:
        K = self.num_conformers
        # The covalent branch depends on nothing the 3-D branch produces: it runs on a second HIP stream, next to the backbone
        # and the FGW solve (whose last wave of workgroups leaves two thirds of the CUs idle).  autograd replays each op's
        # backward on the stream of its forward, so the overlap carries over to the backward pass.
        main = torch.cuda.current_stream()
        side = self._side_stream(main.device) if GAT_ON_SIDE_STREAM else main
        side.wait_stream(main)
        with torch.cuda.stream(side):
            x_cov = self.gat_embeddings_model(batch.x, batch.edge_index, batch.edge_attr, batch.batch,
                                              **({"num_graphs": num_graphs} if num_graphs is not None else {}))     # :165-167
            x_cov = ops.linear(x_cov, self.transformation_matrix_cov.weight, self.transformation_matrix_cov.bias)   # :168
        x_3d, x_bary = self.node_embeddings_model.forward_w_barycenter(
            z=batch.z, pos=batch.pos, num_conformers=K, batch=node_index, max_iter=self.numItermax, epsilon=self.epsilon,
            **({"num_graphs": num_graphs, "max_nodes": max_nodes} if num_graphs is not None else {}))             # :153-160
        G, d = x_3d.shape
        # conformers_mean_aggr(x, conformers_index): the index is K consecutive copies of every molecule id, so the mean is a
        # reshape (checked, because the reference's aggregation accepts any sorted index)
        if conformers_index is not None and conformers_index.numel() != G:
            raise ValueError("conformers_index must have one entry per conformer graph")
        main.wait_stream(side)
        x_cov.record_stream(main)
        if isinstance(self.molecular_regression_lin, Linear) and ops.stage2_head_supported(d):
            # :163-171 in one launch (all of it is linear: the conformer mean is taken first); 17 launches of 5-15 us otherwise
            return ops.stage2_head(x_3d, x_cov, x_bary, self.transformation_matrix_3d, self.transformation_matrix_bary,
                                   self.molecular_regression_lin, self.agg_weight, K)
        x_bary = ops.linear(x_bary, self.transformation_matrix_bary.weight, self.transformation_matrix_bary.bias)  # :163
        x_3d = ops.linear(x_3d, self.transformation_matrix_3d.weight, self.transformation_matrix_3d.bias)          # :164
        x = x_3d + x_cov + self.agg_weight * x_bary                                                                   # :169
        x = x.view(G // K, K, d).mean(dim=1)                                                                          # :170
        return ops.linear(x.contiguous(), self.molecular_regression_lin.weight, self.molecular_regression_lin.bias)   # :171


class EmbeddingsWithGATAggregation(EmbeddingsWithGATAggregationBaryCenter):
    """Stage-1 ("conan_fgw_pre") model: `EmbeddingsWithGATAggregation` (schnet_based_models.py:176-244).  Same sub-modules as
    stage 2 — `transformation_matrix_bary` exists but is unused — so its `state_dict` loads strictly into the stage-2 model,
    which is how the reference starts stage 2 (train_val.py:175-183).  forward: x = Lin3d(backbone(z, pos)) + Lin_cov(GAT(...)),
    conformer mean, regression (:231-244)."""

    def forward(self, batch, conformers_index: Tensor, node_index: Tensor, num_graphs: int = None, max_nodes: int = None) -> Tensor:
        K = self.num_conformers
        x_3d = self.node_embeddings_model(batch.z, batch.pos, node_index)                                            # :231
        x_3d = ops.linear(x_3d, self.transformation_matrix_3d.weight, self.transformation_matrix_3d.bias)            # :232
        x_cov = self.gat_embeddings_model(batch.x, batch.edge_index, batch.edge_attr, batch.batch,
                                          **({"num_graphs": num_graphs} if num_graphs is not None else {}))           # :233-235
        x_cov = ops.linear(x_cov, self.transformation_matrix_cov.weight, self.transformation_matrix_cov.bias)        # :236
        x = x_3d + x_cov                                                                                               # :237
        G, d = x.shape
        if conformers_index is not None and conformers_index.numel() != G:
            raise ValueError("conformers_index must have one entry per conformer graph")
        x = x.view(G // K, K, d).mean(dim=1)                                                                           # :238
        return ops.linear(x.contiguous(), self.molecular_regression_lin.weight, self.molecular_regression_lin.bias)    # :239


class _SelfAttentionParams(torch.nn.Module):
    """`SelfAttention(out_channels)` (attention_layer.py:17-33): the classification model constructs it (:336) but never calls
    it in forward; it exists here so that the reference's state_dict loads strictly."""

    def __init__(self, dim: int):
        super().__init__()
        self.query, self.key, self.value = Linear(dim, dim), Linear(dim, dim), Linear(dim, dim)


class EmbeddingsWithGATAggregationClassificationBaryCenter(EmbeddingsWithGATAggregationBaryCenter):
    """Classification twin (schnet_based_models.py:308-369 on EquivAggregationClassification, common.py:426-466): SchNet with
    hidden 512 / 256 filters / 10 gaussians / cutoff 10 (common.py:513-522), 256-wide GAT and transformations, the three-layer
    ReLU MLP of `build_mlp_class(is_complex=True)` (:31-45) and a sigmoid (:367).  Sub-module names as in the reference
    (`molecular_regression_lin.{0,2,4}`, `self_attention.*`)."""

    def __init__(self, num_conformers: int, device=None, agg_weight: float = 0.2, gat_in_channels: int = 9, model_name: str = "schnet",
                 feat_dim: int = 512):
        """`model_name` as in the reference's constructor (schnet_based_models.py:313 -> common.py:444-446): "schnet" (512 / 256 / 10
        gaussians) or "visnet" (`get_model("visnet", feat_dim=...)` ignores the cutoff keyword and keeps 5 A, common.py:542-546).
        `feat_dim` is the reference's literal 512; SURVEY.md 8(d) cfg4 (BACE + ViSNet-128 + sigmoid head) passes 128."""
        torch.nn.Module.__init__(self)
        from .gat import GATBased
        self.num_conformers = num_conformers
        device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.node_embeddings_model = EquivModelsHolder.get_model(model_name, device, feat_dim=feat_dim, cutoff=10.0)   # common.py:444-446
        out_channels = self.node_embeddings_model.hidden_channels // 2                      # 256
        self.gat_embeddings_model = (EquivModelsHolder.get_model("gat", device, feat_dim=feat_dim) if gat_in_channels == 9      # :330 (512)
                                     else GATBased(out_channels=feat_dim // 2, in_channels=gat_in_channels))
        self.transformation_matrix_3d = Linear(out_channels, out_channels)
        self.transformation_matrix_cov = Linear(out_channels, out_channels)
        self.transformation_matrix_bary = Linear(out_channels, out_channels)
        self.molecular_regression_lin = torch.nn.Sequential(                                # build_mlp_class(out_channels, is_complex=True)
            Linear(out_channels, out_channels), torch.nn.ReLU(), Linear(out_channels, out_channels // 2), torch.nn.ReLU(),
            Linear(out_channels // 2, 1))
        self.self_attention = _SelfAttentionParams(out_channels)
        self.numItermax, self.epsilon, self.agg_weight = 100, 0.1, agg_weight

    def forward(self, batch, conformers_index: Tensor, node_index: Tensor, num_graphs: int = None, max_nodes: int = None) -> Tensor:
        K = self.num_conformers
        main = torch.cuda.current_stream()
        side = self._side_stream(main.device)
        side.wait_stream(main)
        with torch.cuda.stream(side):                                                                                # covalent branch next to the 3-D branch
            x_cov = self.gat_embeddings_model(batch.x, batch.edge_index, batch.edge_attr, batch.batch,
                                              **({"num_graphs": num_graphs} if num_graphs is not None else {}))     # :359-361
            x_cov = ops.linear(x_cov, self.transformation_matrix_cov.weight, self.transformation_matrix_cov.bias)   # :362
        x_3d, x_bary = self.node_embeddings_model.forward_w_barycenter(
            z=batch.z, pos=batch.pos, num_conformers=K, batch=node_index,
            **({"num_graphs": num_graphs, "max_nodes": max_nodes} if num_graphs is not None else {}))             # :351-353
        x_3d = ops.linear(x_3d, self.transformation_matrix_3d.weight, self.transformation_matrix_3d.bias)          # :356
        x_bary = ops.linear(x_bary, self.transformation_matrix_bary.weight, self.transformation_matrix_bary.bias)  # :357
        main.wait_stream(side)
        x_cov.record_stream(main)
        x = x_3d + x_cov + self.agg_weight * x_bary                                                                   # :364
        G, d = x.shape
        if conformers_index is not None and conformers_index.numel() != G:
            raise ValueError("conformers_index must have one entry per conformer graph")
        x = x.view(G // K, K, d).mean(dim=1).contiguous()                                                             # :365
        mlp = self.molecular_regression_lin
        x = ops.relu(ops.linear(x, mlp[0].weight, mlp[0].bias))                                                       # :366
        x = ops.relu(ops.linear(x, mlp[2].weight, mlp[2].bias))
        x = ops.linear(x, mlp[4].weight, mlp[4].bias)
        return ops.sigmoid(x)                                                                                         # :367
