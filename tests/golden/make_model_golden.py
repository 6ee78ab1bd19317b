"""Generates tests/golden/schnet_ref_*.npz.

RUNS ONLY IN THE BUILD CONTAINER (needs /root/reference).  PyG / torch-cluster are not installed here, so the
reference's `SchNetNoSum` (conan_fgw/src/model/graph_embeddings/schnet_no_sum.py) is imported UNCHANGED over a
stand-in `torch_geometric` package assembled in sys.modules from oracle/pyg_semantics.py (our restatement of the
PyG-2.3.0 names it uses).  What this pins: the reference's own wiring and FGW code (forward, forward_3d_bary,
_compute_barycenter, forward_w_barycenter, fgw_barycenters).  What it does NOT pin: the PyG trunk arithmetic, which
comes from the stand-in (parity unpinned).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_model_golden.py
"""
import os
import sys
import types
import warnings

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")
warnings.filterwarnings("ignore")

from oracle import pyg_semantics as ps  # noqa: E402


def install_standin():
    def mod(name):
        m = types.ModuleType(name)
        sys.modules[name] = m
        return m

    tg = mod("torch_geometric")
    nn = mod("torch_geometric.nn")
    aggr = mod("torch_geometric.nn.aggr")
    models = mod("torch_geometric.nn.models")
    schnet = mod("torch_geometric.nn.models.schnet")
    resolver = mod("torch_geometric.nn.resolver")
    typing_ = mod("torch_geometric.typing")
    utils = mod("torch_geometric.utils")
    tg.nn, tg.typing, tg.utils = nn, typing_, utils
    nn.aggr, nn.models, nn.resolver = aggr, models, resolver
    models.schnet = schnet
    nn.SchNet = ps.SchNet
    nn.MessagePassing = ps.MessagePassing
    nn.radius_graph = ps.radius_graph
    aggr.SumAggregation, aggr.MeanAggregation = ps.SumAggregation, ps.MeanAggregation
    for n in ("InteractionBlock", "CFConv", "GaussianSmearing", "ShiftedSoftplus", "RadiusInteractionGraph", "SchNet"):
        setattr(schnet, n, getattr(ps, n))
    resolver.aggregation_resolver = ps.aggregation_resolver
    typing_.OptTensor = ps.OptTensor
    utils.to_dense_adj, utils.to_dense_batch, utils.scatter = ps.to_dense_adj, ps.to_dense_batch, ps.scatter


def main():
    install_standin()
    from conan_fgw.src.model.graph_embeddings.schnet_no_sum import SchNetNoSum  # the reference's class
    import conan_fgw.src.model.graph_embeddings.schnet_no_sum as ref_mod
    from conan_fgw_amd.synthetic import make_batch

    # The reference's glue builds ps/lambdas/adjacency as float32 literals (schnet_no_sum.py:264-279), so its fp64
    # ("ref64") run needs those three cast to the feature dtype at the call boundary; nothing else is touched.
    orig_fgw = ref_mod.fgw_barycenters

    def fgw_cast(**kw):
        dt = kw["Ys"][0].dtype
        kw["Cs"] = [c.to(dt) for c in kw["Cs"]]
        kw["ps"] = [q.to(dt) for q in kw["ps"]]
        kw["lambdas"] = kw["lambdas"].to(dt)
        kw["init_C"] = kw["init_C"].to(dt)
        return orig_fgw(**kw)

    ref_mod.fgw_barycenters = fgw_cast

    cases = [
        # name, shape, B, K, seed, hidden, box
        ("b2_k3_h32", "freesolv", 2, 3, 101, 32, None),
        ("b4_k5_h128", "esol", 4, 5, 102, 128, None),
        ("b3_k5_h64_stretched", "esol", 3, 5, 103, 64, 16.0),
        # Lipophilicity-shaped conformers (n > 33 at r = 10 A): the neighbour cap truncates (33-candidate window incl. self)
        ("b2_k3_h32_lipo", "lipo", 2, 3, 104, 32, None),
    ]
    for name, shape, B, K, seed, H, box in cases:
        torch.manual_seed(5)                         # train_val.py:223
        model = SchNetNoSum(torch.device("cpu"), hidden_channels=H, num_filters=H, num_interactions=3,
                            use_covalent=False)      # common.py:524-529
        # default init leaves biases at zero; perturb every parameter so that biases and the padding row matter
        g = torch.Generator().manual_seed(seed)
        with torch.no_grad():
            for p in model.parameters():
                p.add_(0.05 * torch.randn(p.shape, generator=g))
        b = make_batch(shape, B, K, seed=seed, box=box)
        z, pos, batch = torch.from_numpy(b.z), torch.from_numpy(b.pos), torch.from_numpy(b.batch)
        rec = dict(z=b.z, pos=b.pos, batch=b.batch, K=np.int64(K), B=np.int64(B), hidden=np.int64(H))
        sd = model.state_dict()
        for k, v in sd.items():
            if ".conv.nn." in k:      # alias of interactions.i.mlp.* (PyG registers the filter MLP twice); re-added on load
                continue
            rec["sd:" + k] = v.numpy()
        for tag, dt in (("r32", torch.float32), ("r64", torch.float64)):
            m = model.double() if dt == torch.float64 else model.float()
            p = pos.to(dt)
            with torch.no_grad():
                out = m(z, p, batch)                                       # stage-1 path, schnet_no_sum.py:144-188
                h, hb = m.forward_3d_bary(z, p, batch)
                ei, ew = m.interaction_graph(p, batch)
            for prm in m.parameters():
                prm.grad = None
            h3d, hbary = m.forward_w_barycenter(z=z, pos=p, num_conformers=K, batch=batch)
            gw1 = torch.from_numpy(np.random.RandomState(3).normal(size=tuple(h3d.shape))).to(dt)
            gw2 = torch.from_numpy(np.random.RandomState(4).normal(size=tuple(hbary.shape))).to(dt)
            ((h3d * gw1).sum() + (hbary * gw2).sum()).backward()
            f = (lambda a: a.detach().numpy().astype(np.float32)) if tag == "r32" else (lambda a: a.detach().numpy())
            rec.update({f"{tag}_forward": f(out), f"{tag}_h": f(h), f"{tag}_h_bary_nodes": f(hb), f"{tag}_edge_weight": f(ew),
                        f"{tag}_h_3d": f(h3d), f"{tag}_h_bary": f(hbary)})
            if tag == "r32" or H <= 32:
                for k, prm in m.named_parameters():
                    rec[f"{tag}_grad:{k}"] = f(prm.grad) if prm.grad is not None else np.zeros(0, np.float32)
            if tag == "r32":
                rec["edge_index"] = ei.numpy()
                rec["gw_h3d"], rec["gw_hbary"] = gw1.numpy(), gw2.numpy()
            else:
                assert np.array_equal(rec["edge_index"], ei.numpy()), "neighbour lists differ between fp32 and fp64 positions"
        model.float()
        np.savez_compressed(os.path.join(HERE, f"schnet_ref_{name}.npz"), **rec)
        rel = lambda a, b: float(np.linalg.norm(a - b) / np.linalg.norm(b))
        print(name, "atoms", len(b.z), "E", rec["edge_index"].shape[1], "Nmax", b.max_nodes,
              "rel(h3d32,64)=%.2e rel(hbary32,64)=%.2e" % (rel(rec["r32_h_3d"], rec["r64_h_3d"]), rel(rec["r32_h_bary"], rec["r64_h_bary"])))


def main_visnet():
    """Same procedure for the ViSNet wrapper (visnet.py) over the vendored torch_geometric_visnet.py."""
    install_standin()
    import conan_fgw.src.model.graph_embeddings.visnet as ref_mod
    from conan_fgw_amd.synthetic import make_batch
    orig_fgw = ref_mod.fgw_barycenters

    def fgw_cast(**kw):
        dt = kw["Ys"][0].dtype
        kw["Cs"] = [c.to(dt) for c in kw["Cs"]]
        kw["ps"] = [q.to(dt) for q in kw["ps"]]
        kw["lambdas"] = kw["lambdas"].to(dt)
        kw["init_C"] = kw["init_C"].to(dt)
        return orig_fgw(**kw)

    ref_mod.fgw_barycenters = fgw_cast
    cases = [("b2_k3_h32", "bace", 2, 3, 201, 32, None), ("b3_k5_h64", "esol", 3, 5, 202, 64, 6.0)]
    for name, shape, B, K, seed, H, box in cases:
        torch.manual_seed(5)
        model = ref_mod.ViSNet(torch.device("cpu"), hidden_channels=H)            # common.py:542-546
        g = torch.Generator().manual_seed(seed)
        with torch.no_grad():
            for p in model.parameters():
                p.add_(0.05 * torch.randn(p.shape, generator=g))
        b = make_batch(shape, B, K, seed=seed, box=box)
        z, pos, batch = torch.from_numpy(b.z), torch.from_numpy(b.pos), torch.from_numpy(b.batch)
        rec = dict(z=b.z, pos=b.pos, batch=b.batch, K=np.int64(K), B=np.int64(B), hidden=np.int64(H))
        for k, v in model.state_dict().items():
            rec["sd:" + k] = v.numpy()
        for tag, dt in (("r32", torch.float32), ("r64", torch.float64)):
            m = model.double() if dt == torch.float64 else model.float()
            p = pos.to(dt)
            # the vendored ViSNet allocates with torch.zeros(...) (torch_geometric_visnet.py:342,868): its fp64 run needs the
            # process-wide default dtype switched; no reference code is modified
            torch.set_default_dtype(dt)
            with torch.no_grad():
                xs, vs = m.representation_model(z, p, batch)
                out = m(z, p, batch)
                h, hb = m.forward_3d_bary(z, p, batch)
                h3d, hbary = m.forward_w_barycenter(z=z, pos=p, num_conformers=K, batch=batch)
                ei, ew = m.interaction_graph(p, batch)
                ei_loop, ew_loop, _ = m.representation_model.distance(p, batch)
            f = (lambda a: a.detach().numpy().astype(np.float32)) if tag == "r32" else (lambda a: a.detach().numpy())
            rec.update({f"{tag}_x": f(xs), f"{tag}_vec": f(vs), f"{tag}_forward": f(out), f"{tag}_h": f(h), f"{tag}_h_bary_nodes": f(hb),
                        f"{tag}_h_3d": f(h3d), f"{tag}_h_bary": f(hbary)})
            if tag == "r32":
                rec["edge_index"], rec["edge_index_loop"] = ei.numpy(), ei_loop.numpy()
        torch.set_default_dtype(torch.float32)
        model.float()
        np.savez_compressed(os.path.join(HERE, f"visnet_ref_{name}.npz"), **rec)
        rel = lambda a, c: float(np.linalg.norm(a - c) / np.linalg.norm(c))
        print("visnet", name, "atoms", len(b.z), "E_loop", rec["edge_index_loop"].shape[1], "params", sum(p.numel() for p in model.parameters()),
              "rel(x32,64)=%.2e rel(h3d)=%.2e rel(hbary)=%.2e" % (rel(rec["r32_x"], rec["r64_x"]), rel(rec["r32_h_3d"], rec["r64_h_3d"]), rel(rec["r32_h_bary"], rec["r64_h_bary"])))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "visnet":
        main_visnet()
    else:
        main()
        main_visnet()
