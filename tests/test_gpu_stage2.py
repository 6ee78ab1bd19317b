"""The whole stage-2 regression model (3-D backbone + FGW barycenter + covalent GAT branch + conformer mean + regression,
schnet_based_models.py:135-173) through the HIP path vs the fp64 CPU oracle: predictions within 1e-4 relative (the bar of
BASELINE.json for energies), gradients of representative parameters of every branch within 1e-4."""
import types

import numpy as np
import pytest
import torch

from helpers import rel
from conan_fgw_amd.synthetic import make_batch, make_bond_graph

pytestmark = pytest.mark.gpu


def _build(B=6, K=5, seed=21):
    from conan_fgw_amd.head import EmbeddingsWithGATAggregationBaryCenter
    from oracle.head import Stage2Oracle
    dev = torch.device("cuda:0")
    b = make_batch("esol", B, K, seed=seed)
    g = make_bond_graph(b, seed=seed + 1)
    torch.manual_seed(5)
    m = EmbeddingsWithGATAggregationBaryCenter(K, dev).to(dev)
    ref = Stage2Oracle(K).double()
    missing = ref.load_state_dict({k: v.detach().cpu().double() for k, v in m.state_dict().items()}, strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    return dev, b, g, m, ref


def test_stage2_prediction_and_gradients_match_oracle():
    dev, b, g, m, ref = _build()
    t = lambda a: torch.from_numpy(a)
    batch = types.SimpleNamespace(z=t(b.z).to(dev), pos=t(b.pos).to(dev), x=t(g.x).to(dev), edge_index=t(g.edge_index).to(dev),
                                  edge_attr=t(g.edge_attr).to(dev), batch=t(b.batch).to(dev))
    cidx = m.create_aggregation_index(b.num_graphs, dev)
    assert cidx.tolist() == [i for i in range(b.num_molecules) for _ in range(b.num_conformers)]       # common.py:414-423
    y = m(batch, cidx, batch.batch)
    r = ref(t(b.z), t(b.pos).double(), t(b.batch), t(g.x), t(g.edge_index), t(g.edge_attr))
    assert y.shape == (b.num_molecules, 1)
    assert rel(y.detach().cpu().double().numpy(), r.detach().numpy()) < 1e-4
    tgt = t(b.y)[:, None]
    torch.nn.functional.mse_loss(y, tgt.to(dev)).backward()
    torch.nn.functional.mse_loss(r, tgt.double()).backward()
    gp, rp = dict(m.named_parameters()), dict(ref.named_parameters())
    assert set(gp) == set(rp)
    gmax = max(float(p.grad.norm()) for p in rp.values() if p.grad is not None)
    for k in ["molecular_regression_lin.weight", "transformation_matrix_cov.weight", "transformation_matrix_bary.bias",
              "gat_embeddings_model.gat_conv1.lin_src.weight", "gat_embeddings_model.gat_conv2.att_src",
              "gat_embeddings_model.gat_conv1.lin_edge.weight", "node_embeddings_model.lin1_bary.weight",
              "node_embeddings_model.interactions.0.mlp.0.weight", "node_embeddings_model.embedding.weight"]:
        err = float((gp[k].grad.cpu().double() - rp[k].grad).norm())
        assert err <= 1e-4 * float(rp[k].grad.norm()) + 1e-6 * gmax, (k, err, float(rp[k].grad.norm()))


def test_stage2_hints_do_not_change_the_result():
    dev, b, g, m, _ = _build(B=3, K=3, seed=8)
    t = lambda a: torch.from_numpy(a).to(dev)
    batch = types.SimpleNamespace(z=t(b.z), pos=t(b.pos), x=t(g.x), edge_index=t(g.edge_index), edge_attr=t(g.edge_attr), batch=t(b.batch))
    cidx = m.create_aggregation_index(b.num_graphs, dev)
    with torch.no_grad():
        y1 = m(batch, cidx, batch.batch)
        y2 = m(batch, cidx, batch.batch, num_graphs=b.num_graphs, max_nodes=b.max_nodes)
    assert torch.equal(y1, y2)


def test_stage1_model_and_checkpoint_round_trip_into_stage2(tmp_path):
    """Stage 1 (`EmbeddingsWithGATAggregation`, no FGW) vs its oracle, then the two-stage recipe of the reference: the stage-1
    `state_dict` is saved as a Lightning-style checkpoint and loaded strictly into the stage-2 model (train_val.py:175-183)."""
    from conan_fgw_amd.head import EmbeddingsWithGATAggregation, EmbeddingsWithGATAggregationBaryCenter
    from oracle.head import Stage1Oracle
    dev = torch.device("cuda:0")
    K = 3
    b = make_batch("esol", 4, K, seed=31)
    g = make_bond_graph(b, seed=32)
    torch.manual_seed(7)
    m1 = EmbeddingsWithGATAggregation(K, dev).to(dev)
    ref = Stage1Oracle(K).double()
    ref.load_state_dict({k: v.detach().cpu().double() for k, v in m1.state_dict().items()}, strict=True)
    t = lambda a: torch.from_numpy(a)
    batch = types.SimpleNamespace(z=t(b.z).to(dev), pos=t(b.pos).to(dev), x=t(g.x).to(dev), edge_index=t(g.edge_index).to(dev),
                                  edge_attr=t(g.edge_attr).to(dev), batch=t(b.batch).to(dev))
    cidx = m1.create_aggregation_index(b.num_graphs, dev)
    y = m1(batch, cidx, batch.batch)
    r = ref(t(b.z), t(b.pos).double(), t(b.batch), t(g.x), t(g.edge_index), t(g.edge_attr))
    assert rel(y.detach().cpu().double().numpy(), r.detach().numpy()) < 1e-5
    y.sum().backward()
    assert m1.transformation_matrix_bary.weight.grad is None                 # unused in stage 1, exactly like the reference
    assert m1.node_embeddings_model.lin2.weight.grad is not None
    path = tmp_path / "stage1.ckpt"
    torch.save({"state_dict": m1.state_dict()}, path)
    torch.manual_seed(99)
    m2 = EmbeddingsWithGATAggregationBaryCenter(K, dev).to(dev)
    res = m2.load_state_dict(torch.load(path)["state_dict"])                 # strict
    assert not res.missing_keys and not res.unexpected_keys
    for (k1, v1), (k2, v2) in zip(m1.state_dict().items(), m2.state_dict().items()):
        assert k1 == k2 and torch.equal(v1, v2)
    with torch.no_grad():
        y2 = m2(batch, cidx, batch.batch)
    assert y2.shape == y.shape and torch.isfinite(y2).all()


def test_classification_stage2_matches_oracle():
    """Classification twin (SchNet 512 / 256 filters / 10 gaussians, 256-wide GAT, ReLU MLP, sigmoid; schnet_based_models.py:308-369):
    probabilities within 1e-4 of the fp64 oracle, BCE gradients of parameters of every branch within 1e-4, strict state_dict."""
    from conan_fgw_amd.head import EmbeddingsWithGATAggregationClassificationBaryCenter
    from oracle.head import Stage2ClassificationOracle
    dev = torch.device("cuda:0")
    K = 3
    b = make_batch("esol", 4, K, seed=41)
    g = make_bond_graph(b, seed=42)
    torch.manual_seed(9)
    m = EmbeddingsWithGATAggregationClassificationBaryCenter(K, dev).to(dev)
    ref = Stage2ClassificationOracle(K).double()
    res = ref.load_state_dict({k: v.detach().cpu().double() for k, v in m.state_dict().items()}, strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    assert m.node_embeddings_model.hidden_channels == 512 and m.molecular_regression_lin[2].weight.shape == (128, 256)
    t = lambda a: torch.from_numpy(a)
    batch = types.SimpleNamespace(z=t(b.z).to(dev), pos=t(b.pos).to(dev), x=t(g.x).to(dev), edge_index=t(g.edge_index).to(dev),
                                  edge_attr=t(g.edge_attr).to(dev), batch=t(b.batch).to(dev))
    p = m(batch, m.create_aggregation_index(b.num_graphs, dev), batch.batch)
    r = ref(t(b.z), t(b.pos).double(), t(b.batch), t(g.x), t(g.edge_index), t(g.edge_attr))
    assert p.shape == (b.num_molecules, 1) and float(p.detach().min()) > 0.0 and float(p.detach().max()) < 1.0
    assert rel(p.detach().cpu().double().numpy(), r.detach().numpy()) < 1e-4
    lab = torch.tensor([[1.0], [0.0], [1.0], [0.0]])
    torch.nn.functional.binary_cross_entropy(p, lab.to(dev)).backward()
    torch.nn.functional.binary_cross_entropy(r, lab.double()).backward()
    gp, rp = dict(m.named_parameters()), dict(ref.named_parameters())
    gmax = max(float(q.grad.norm()) for q in rp.values() if q.grad is not None)
    for k in ["molecular_regression_lin.0.weight", "molecular_regression_lin.4.bias", "transformation_matrix_bary.weight",
              "gat_embeddings_model.gat_conv2.lin_src.weight", "node_embeddings_model.lin2_bary.weight",
              "node_embeddings_model.interactions.2.mlp.2.weight", "node_embeddings_model.interactions.0.conv.lin1.weight"]:
        err = float((gp[k].grad.cpu().double() - rp[k].grad).norm())
        assert err <= 1e-4 * float(rp[k].grad.norm()) + 1e-6 * gmax, (k, err, float(rp[k].grad.norm()))
    assert m.self_attention.query.weight.grad is None                       # constructed, never used (like the reference)


def test_stage2_with_visnet_backbone_matches_oracle():
    """model_name="visnet" (common.py:542-546): the same stage-2 assembly on the ViSNet backbone; BACE-sized conformers."""
    from conan_fgw_amd.head import EmbeddingsWithGATAggregationBaryCenter
    from oracle.head import Stage2Oracle
    dev = torch.device("cuda:0")
    K = 2
    b = make_batch("bace", 2, K, seed=61)
    g = make_bond_graph(b, seed=62)
    torch.manual_seed(4)
    m = EmbeddingsWithGATAggregationBaryCenter(K, dev, model_name="visnet").to(dev)
    ref = Stage2Oracle(K, model_name="visnet").double()
    res = ref.load_state_dict({k: v.detach().cpu().double() for k, v in m.state_dict().items()}, strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    t = lambda a: torch.from_numpy(a)
    batch = types.SimpleNamespace(z=t(b.z).to(dev), pos=t(b.pos).to(dev), x=t(g.x).to(dev), edge_index=t(g.edge_index).to(dev),
                                  edge_attr=t(g.edge_attr).to(dev), batch=t(b.batch).to(dev))
    with torch.no_grad():
        y = m(batch, m.create_aggregation_index(b.num_graphs, dev), batch.batch)
        r = ref(t(b.z), t(b.pos).double(), t(b.batch), t(g.x), t(g.edge_index), t(g.edge_attr))
    assert y.shape == (b.num_molecules, 1)
    assert rel(y.cpu().double().numpy(), r.numpy()) < 1e-4


def test_classification_head_on_visnet_backbone_cfg4():
    """BACE-shaped conformers + ViSNet-128 + the classification (sigmoid) head — `model_name="visnet"` of
    EmbeddingsWithGATAggregationClassificationBaryCenter (schnet_based_models.py:308-369, common.py:444-446 -> :542-546)."""
    from conan_fgw_amd.head import EmbeddingsWithGATAggregationClassificationBaryCenter
    from oracle.head import Stage2ClassificationOracle
    dev = torch.device("cuda:0")
    K = 2
    b = make_batch("bace", 2, K, seed=71)
    g = make_bond_graph(b, seed=72)
    torch.manual_seed(6)
    m = EmbeddingsWithGATAggregationClassificationBaryCenter(K, dev, model_name="visnet", feat_dim=128).to(dev)
    ref = Stage2ClassificationOracle(K, model_name="visnet", feat_dim=128).double()
    res = ref.load_state_dict({k: v.detach().cpu().double() for k, v in m.state_dict().items()}, strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    t = lambda a: torch.from_numpy(a)
    batch = types.SimpleNamespace(z=t(b.z).to(dev), pos=t(b.pos).to(dev), x=t(g.x).to(dev), edge_index=t(g.edge_index).to(dev),
                                  edge_attr=t(g.edge_attr).to(dev), batch=t(b.batch).to(dev))
    p = m(batch, m.create_aggregation_index(b.num_graphs, dev), batch.batch)
    r = ref(t(b.z), t(b.pos).double(), t(b.batch), t(g.x), t(g.edge_index), t(g.edge_attr))
    assert p.shape == (b.num_molecules, 1) and 0.0 < float(p.detach().min()) and float(p.detach().max()) < 1.0
    assert rel(p.detach().cpu().double().numpy(), r.detach().numpy()) < 1e-4
    lab = torch.tensor([[1.0], [0.0]])
    torch.nn.functional.binary_cross_entropy(p, lab.to(dev)).backward()
    torch.nn.functional.binary_cross_entropy(r, lab.double()).backward()
    gp, rp = dict(m.named_parameters()), dict(ref.named_parameters())
    gmax = max(float(q.grad.norm()) for q in rp.values() if q.grad is not None)
    for k in ["molecular_regression_lin.2.weight", "transformation_matrix_3d.weight", "transformation_matrix_bary.weight"]:
        err = float((gp[k].grad.cpu().double() - rp[k].grad).norm())
        assert err <= 1e-4 * float(rp[k].grad.norm()) + 1e-6 * gmax, (k, err, float(rp[k].grad.norm()))


def test_deferred_weight_gradients_are_bitwise_equal_to_immediate_ones():
    """ops.deferred_weight_gradients / FlatGradients.backward: the slab reductions of all Linear layers in ONE launch and the slab kernels
    of the node-level layers in one launch too — same slices, same sums in the same order => every gradient bit-identical to the
    immediate path; a weight used twice in one backward is handled (flush + immediate)."""
    from conan_fgw_amd import ops
    from conan_fgw_amd.parallel import FlatGradients
    dev, b, g, m, _ = _build(B=4, K=3, seed=31)
    t = lambda a: torch.from_numpy(a).to(dev)
    batch = types.SimpleNamespace(z=t(b.z), pos=t(b.pos), x=t(g.x), edge_index=t(g.edge_index), edge_attr=t(g.edge_attr), batch=t(b.batch))
    cidx = m.create_aggregation_index(b.num_graphs, dev)
    tgt = t(b.y)[:, None]

    def grads(deferred):
        for p in m.parameters():
            p.grad = None
        loss = torch.nn.functional.mse_loss(m(batch, cidx, batch.batch), tgt)
        if deferred:
            flat = FlatGradients(m.parameters())
            flat.backward(loss)
            assert ops._pending is None                                   # context closed => everything flushed
        else:
            loss.backward()
        torch.cuda.synchronize()
        return {k: p.grad.clone() for k, p in m.named_parameters()}

    a, c = grads(False), grads(True)
    for k in a:                                                       # default: the batch picks its own slice count (fewer, longer slices: another order of the same sums)
        assert rel(c[k].cpu(), a[k].cpu()) < 2e-6, k
    ops.LATE_SLICES_AUTO = False                                      # with the library's slices the deferred path reproduces the immediate one bit for bit
    try:
        c = grads(True)
    finally:
        ops.LATE_SLICES_AUTO = True
    for k in a:
        if k.endswith("embedding.weight"):
            # round 5: inside a deferred pass the embedding gradient is the weight gradient onehot(z)^T dout of the batched launch (MFMA, exact
            # planes), outside it the two-kernel LDS-table form: the same sums in another order
            assert rel(c[k].cpu(), a[k].cpu()) < 1e-6, k
            continue
        assert torch.equal(a[k], c[k]), k
    # the same weight twice in one graph: y = lin(lin(x))
    w = torch.randn(64, 64, device=dev, requires_grad=True); x = torch.randn(300, 64, device=dev)
    ops.linear(ops.linear(x, w), w).sum().backward(); ref = w.grad.clone(); w.grad = None
    with ops.deferred_weight_gradients():
        ops.linear(ops.linear(x, w), w).sum().backward()
    torch.cuda.synchronize()
    assert torch.equal(w.grad, ref)


@pytest.mark.parametrize("B,K,D", [(1, 1, 64), (7, 5, 64), (256, 5, 64), (33, 20, 32)])
def test_stage2_head_fused_matches_the_composed_modules(B, K, D):
    """ops.stage2_head (conan_stage2_head_fwd / _bwd: transformation_matrix_3d / _bary, weighted sum, conformer mean and the regression
    layer in one launch each way) against the same expression in fp64 torch with autograd: output, the three input gradients and all six
    parameter gradients; repeat runs are bitwise equal."""
    from conan_fgw_amd import ops
    dev = torch.device("cuda:0")
    gen = torch.Generator().manual_seed(B * 100 + K)
    G = B * K
    mk = lambda *s: torch.randn(*s, generator=gen).to(dev).requires_grad_(True)
    x3, xc, xb = mk(G, D), mk(G, D), mk(G, D)
    l3, lb, lr = torch.nn.Linear(D, D).to(dev), torch.nn.Linear(D, D).to(dev), torch.nn.Linear(D, 1).to(dev)
    gy = torch.randn(B, 1, generator=gen).to(dev)
    leaves = [x3, xc, xb, l3.weight, l3.bias, lb.weight, lb.bias, lr.weight, lr.bias]

    def run():
        for t in leaves:
            t.grad = None
        out = ops.stage2_head(x3, xc, xb, l3, lb, lr, 0.2, K)
        (out * gy).sum().backward()
        return [out.detach().clone()] + [t.grad.detach().clone() for t in leaves]

    a, a2 = run(), run()
    d = [t.detach().double().requires_grad_(True) for t in leaves]
    x = (d[0] @ d[3].T + d[4]) + d[1] + 0.2 * (d[2] @ d[5].T + d[6])
    out = x.view(B, K, D).mean(1) @ d[7].T + d[8]
    (out * gy.double()).sum().backward()
    ref = [out.detach()] + [t.grad for t in d]
    assert a[0].shape == (B, 1)
    for u, v, r in zip(a, a2, ref):
        assert torch.equal(u, v)
        assert rel(u.double().cpu().numpy(), r.cpu().numpy()) < 2e-6


def test_training_steps_on_ragged_batches_follow_the_oracle():
    """Four optimiser steps on four DIFFERENT ragged batches, the product path in its production form — worker-thread collate pipeline,
    host-known hints, one-launch MSE, deferred + batched weight gradients, flat gradient buffer (the all-reduce is the identity at world
    size 1) — against the fp64 oracle model trained on the same batches with plain autograd.  SGD with momentum, not Adam: its update is
    linear in the gradient, so the parameter trajectories are comparable (Adam divides by |g| and turns the rounding noise of a vanishing
    gradient into a full-size step of random sign in either implementation).  trainer: train_val.py:223-241, common.py:246-262."""
    from conan_fgw_amd import ops
    from conan_fgw_amd.collate import CollatePipeline, DeviceCollator, molecules_from_synthetic
    from conan_fgw_amd.parallel import FlatGradients
    dev, _b, _g, m, ref = _build(B=2, K=3, seed=4)
    K = 3
    sets, raw = [], []
    for i, (B, seed) in enumerate([(5, 61), (3, 62), (6, 63), (4, 64)]):
        cb = make_batch("esol", B, K, seed=seed); bg = make_bond_graph(cb, seed=seed + 100)
        sets.append(molecules_from_synthetic(cb, bg)); raw.append((cb, bg))
    flat = FlatGradients(m.parameters())
    opt = torch.optim.SGD(flat.params, lr=1e-6, momentum=0.9)        # (the untrained model predicts O(10): gradients are O(1e3), the steps O(1e-3))
    opt_ref = torch.optim.SGD(ref.parameters(), lr=1e-6, momentum=0.9)
    p0 = {k: v.detach().clone() for k, v in ref.named_parameters()}
    losses, losses_ref = [], []
    feed = CollatePipeline(DeviceCollator(dev, K, depth=4), sets, prefetch=2)
    t = lambda a: torch.from_numpy(a)
    for (cb, bg), db in zip(raw, feed):
        db.wait()
        data, node_index = db.as_model_input()
        flat.zero()
        pred = m(data, db.conformers_index, node_index, num_graphs=db.num_graphs, max_nodes=db.max_nodes)
        loss = ops.mse_loss(pred, db.y[::K][:, None].contiguous())             # one label per molecule (the collated y is per conformer graph)
        flat.backward(loss)
        flat.all_reduce_mean()
        opt.step()
        losses.append(float(loss.detach()))
        opt_ref.zero_grad()
        r = ref(t(cb.z), t(cb.pos).double(), t(cb.batch), t(bg.x), t(bg.edge_index), t(bg.edge_attr))
        lr_ = torch.nn.functional.mse_loss(r, t(cb.y)[:, None].double())
        lr_.backward()
        opt_ref.step()
        losses_ref.append(float(lr_.detach()))
    np.testing.assert_allclose(losses, losses_ref, rtol=2e-4)
    gp, rp = dict(m.named_parameters()), dict(ref.named_parameters())
    num = sum(float((gp[k].detach().cpu().double() - rp[k].detach()).pow(2).sum()) for k in rp)
    den = sum(float((rp[k].detach() - p0[k]).pow(2).sum()) for k in rp)
    assert den > 0 and (num / den) ** 0.5 < 2e-3, (num, den)                  # the whole parameter displacement after four steps
    for k in ["molecular_regression_lin.weight", "node_embeddings_model.interactions.1.mlp.2.weight", "gat_embeddings_model.gat_conv1.lin_src.weight",
              "node_embeddings_model.lin1_bary.weight"]:
        d_ref = rp[k].detach() - p0[k]
        err = float((gp[k].detach().cpu().double() - rp[k].detach()).norm())
        assert err <= 5e-3 * float(d_ref.norm()) + 1e-9, (k, err, float(d_ref.norm()))


def test_flat_adam_follows_torch_adam_step_by_step():
    """parallel.FlatAdam (conan_adam_flat_step: one launch over flat parameter / gradient / moment buffers, device-side step counter) against
    torch.optim.Adam on a copy of the same stage-2 model: the same gradients go into both, parameters must agree after every one of six steps
    (1e-6 relative: the two differ only in how b^t is evaluated), with and without weight decay; the re-pointed parameters still drive the model
    and round-trip through state_dict."""
    import copy
    from conan_fgw_amd.head import EmbeddingsWithGATAggregationBaryCenter
    from conan_fgw_amd.parallel import FlatAdam, FlatGradients
    dev = torch.device("cuda:0")
    for wd in (0.0, 1e-2):
        torch.manual_seed(11)
        ma = EmbeddingsWithGATAggregationBaryCenter(3, dev).to(dev)
        mb = copy.deepcopy(ma)
        fa = FlatGradients(ma.parameters())
        oa = FlatAdam(fa, lr=3e-3, weight_decay=wd)
        pb = [p for p in mb.parameters() if p.requires_grad]
        ob = torch.optim.Adam(pb, lr=3e-3, weight_decay=wd)
        assert all(p.data_ptr() >= oa.params.data_ptr() and p.data_ptr() < oa.params.data_ptr() + 4 * oa.params.numel() for p in fa.params)
        g = torch.Generator(device="cpu").manual_seed(5)
        for step in range(6):
            grads = [torch.randn(p.shape, generator=g).to(dev) * (10.0 ** (step - 3)) for p in fa.params]      # magnitudes over six decades
            fa.zero()
            for p, q, gr in zip(fa.params, pb, grads):
                p.grad = gr.clone(); q.grad = gr.clone()
            fa.pack()
            oa.step(); ob.step()
            torch.cuda.synchronize()
            for p, q in zip(fa.params, pb):
                assert rel(p.detach().cpu(), q.detach().cpu()) < 1e-6, (wd, step)
        assert float(oa.step_dev) == 6.0
        sd = ma.state_dict()
        mc = EmbeddingsWithGATAggregationBaryCenter(3, dev).to(dev)
        mc.load_state_dict(sd, strict=True)
        assert all(torch.equal(a, c) for a, c in zip(ma.parameters(), mc.parameters()))


def test_flat_adam_is_an_optimizer_the_reference_loop_can_use():
    """What the reference's training loop does with its optimiser (common.py:253-262: Adam + ReduceLROnPlateau; Lightning checkpoints
    `optimizer.state_dict()`), on parallel.FlatAdam: the learning rate lives on the device (a scheduler's change reaches a CAPTURED launch),
    the state round-trips in torch.optim.Adam's format in both directions, a Parameter whose .data was replaced is taken over again instead of
    training a buffer the model no longer reads, and replaced Parameter OBJECTS are reported."""
    import copy
    from conan_fgw_amd.head import EmbeddingsWithGATAggregationBaryCenter
    from conan_fgw_amd.parallel import FlatAdam, FlatGradients
    dev = torch.device("cuda:0")
    torch.manual_seed(12)
    ma = EmbeddingsWithGATAggregationBaryCenter(3, dev).to(dev)
    mb = copy.deepcopy(ma)
    fa = FlatGradients(ma.parameters())
    oa = FlatAdam(fa, lr=3e-3, module=ma)
    pb = [p for p in mb.parameters() if p.requires_grad]
    ob = torch.optim.Adam(pb, lr=3e-3)
    assert isinstance(oa, torch.optim.Optimizer) and len(oa.param_groups) == 1 and oa.param_groups[0]["lr"].is_cuda
    sa = torch.optim.lr_scheduler.ReduceLROnPlateau(oa, mode="min", patience=0, factor=0.8)          # common.py:258-262
    sb = torch.optim.lr_scheduler.ReduceLROnPlateau(ob, mode="min", patience=0, factor=0.8)
    g = torch.Generator(device="cpu").manual_seed(6)

    def feed(n):
        for _ in range(n):
            grads = [torch.randn(p.shape, generator=g).to(dev) for p in fa.params]
            fa.zero()
            for p, q, gr in zip(fa.params, pb, grads):
                p.grad = gr.clone(); q.grad = gr.clone()
            fa.pack()
            oa.step(); ob.step()

    def same(tol=1e-6):
        torch.cuda.synchronize()
        for p, q in zip(fa.params, pb):
            assert rel(p.detach().cpu(), q.detach().cpu()) < tol

    feed(2); same()
    for metric in (1.0, 2.0, 3.0):                           # no improvement twice: both schedulers cut the rate twice
        sa.step(metric); sb.step(metric)
    assert abs(oa.lr - ob.param_groups[0]["lr"]) < 1e-12 and abs(oa.lr - 3e-3 * 0.64) < 1e-12
    feed(2); same()
    oa.param_groups[0]["lr"] = 1e-3; ob.param_groups[0]["lr"] = 1e-3                                  # plain assignment (user code, other schedulers)
    assert oa.param_groups[0]["lr"].is_cuda and abs(oa.lr - 1e-3) < 1e-15
    feed(1); same()

    # a captured launch follows the device-side rate between replays
    grads = [torch.randn(p.shape, generator=g).to(dev) for p in fa.params]
    fa.zero()
    for p, q, gr in zip(fa.params, pb, grads):
        p.grad = gr.clone(); q.grad = gr.clone()
    fa.pack()
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        with torch.cuda.graph(graph, stream=side):
            oa.step()
    torch.cuda.current_stream().wait_stream(side)
    for lr in (1e-3, 5e-4):
        oa.param_groups[0]["lr"] = lr; ob.param_groups[0]["lr"] = lr
        graph.replay(); ob.step()
        same()

    # checkpoint: torch's format, both directions
    sd_a, sd_b = oa.state_dict(), ob.state_dict()
    assert set(sd_a) == {"state", "param_groups"} and sorted(sd_a["state"]) == sorted(sd_b["state"]) and sd_a["param_groups"][0]["params"] == sd_b["param_groups"][0]["params"]
    for k in sd_b["state"]:
        assert float(sd_a["state"][k]["step"]) == float(sd_b["state"][k]["step"]) == 7.0
        assert rel(sd_a["state"][k]["exp_avg"].cpu(), sd_b["state"][k]["exp_avg"].cpu()) < 1e-5
        assert rel(sd_a["state"][k]["exp_avg_sq"].cpu(), sd_b["state"][k]["exp_avg_sq"].cpu()) < 1e-5
    mc, md = copy.deepcopy(mb), copy.deepcopy(mb)
    fc = FlatGradients(mc.parameters()); oc = FlatAdam(fc, lr=1.0)
    oc.load_state_dict(copy.deepcopy(sd_b))                                                      # resume FlatAdam from torch.optim.Adam's checkpoint
    od = torch.optim.Adam([p for p in md.parameters() if p.requires_grad], lr=1.0)
    od.load_state_dict(copy.deepcopy(sd_a))                                                      # ... and torch.optim.Adam from FlatAdam's
    assert abs(oc.lr - 5e-4) < 1e-15 and od.param_groups[0]["lr"] == 5e-4 and float(oc.step_dev) == 7.0
    grads = [torch.randn(p.shape, generator=g).to(dev) for p in fa.params]
    fa.zero(); fc.zero()
    pd = [p for p in md.parameters() if p.requires_grad]
    for o, p, q, r_, gr in zip(fa.params, fc.params, pb, pd, grads):
        o.grad = gr.clone(); p.grad = gr.clone(); q.grad = gr.clone(); r_.grad = gr.clone()
    fa.pack(); fc.pack()
    oa.step(); oc.step(); ob.step(); od.step()                                                   # (the two originals step too: they stay in step for the checks below)
    same()
    for p, q, r_ in zip(fc.params, pb, pd):
        assert rel(p.detach().cpu(), q.detach().cpu()) < 1e-6 and rel(r_.detach().cpu(), q.detach().cpu()) < 1e-6

    # a replaced .data is taken over (the model's tensors are the truth), not silently left behind
    w = fa.params[0]
    with torch.no_grad():
        w.data = (w.detach() * 0.5).clone()
        pb[0].mul_(0.5)
    with pytest.raises(RuntimeError, match="alias"):
        oa.check_aliasing(repair=False)
    feed(1); same()
    assert oa.check_aliasing() and w.data_ptr() >= oa.params.data_ptr() and w.data_ptr() < oa.params.data_ptr() + 4 * oa.params.numel()
    # replaced Parameter objects (load_state_dict(assign=True)) cannot be re-adopted: reported
    ma.load_state_dict({k: v.clone() for k, v in ma.state_dict().items()}, assign=True)
    with pytest.raises(RuntimeError, match="Parameter objects"):
        oa.step()


def test_clip_grad_norm_on_the_flat_buffer_matches_torch():
    """FlatGradients.clip_grad_norm_ (conan_grad_clip_flat: fp64 sum of squares in a fixed order, coefficient and scaling on the device) against
    torch.nn.utils.clip_grad_norm_ — Lightning's gradient_clip_val=1.0 of the reference's Trainer (trainer.py:177) — for a norm above and below
    the threshold; repeated calls give the same bits."""
    from conan_fgw_amd.head import EmbeddingsWithGATAggregationBaryCenter
    from conan_fgw_amd.parallel import FlatGradients
    dev = torch.device("cuda:0")
    torch.manual_seed(13)
    m = EmbeddingsWithGATAggregationBaryCenter(3, dev).to(dev)
    fa = FlatGradients(m.parameters())
    g = torch.Generator(device="cpu").manual_seed(8)
    for scale in (10.0, 1e-4):
        grads = [torch.randn(p.shape, generator=g).to(dev) * scale for p in fa.params]
        outs = []
        for _ in range(2):
            fa.zero()
            for p, gr in zip(fa.params, grads):
                p.grad = gr.clone()
            fa.pack()
            nrm = fa.clip_grad_norm_(1.0)
            outs.append((nrm.clone(), fa.flat.clone()))
        assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
        ref = [gr.clone().requires_grad_(False) for gr in grads]
        holders = [torch.nn.Parameter(torch.zeros_like(r)) for r in ref]
        for h, r in zip(holders, ref):
            h.grad = r
        ref_norm = torch.nn.utils.clip_grad_norm_(holders, 1.0)
        assert abs(float(nrm) - float(ref_norm)) <= 2e-6 * float(ref_norm)
        for p, h in zip(fa.params, holders):
            assert p.grad.data_ptr() >= fa.flat.data_ptr()                                       # still the aliased view: clipped in place
            assert rel(p.grad.cpu(), h.grad.cpu()) < 2e-6
        if scale < 1:
            assert all(torch.equal(p.grad, gr) for p, gr in zip(fa.params, grads))                    # below the threshold nothing is touched


def test_checkpoint_resume_continues_the_run_bit_for_bit():
    """The reference's optimisation loop on the product path — collate pipeline, stage-2 model, MSE, flat gradients, global-norm clipping (trainer.py:177),
    FlatAdam under ReduceLROnPlateau (common.py:253-262) — interrupted by a checkpoint: seven steps straight through against four steps, a checkpoint of
    model / optimiser / scheduler state (what Lightning saves), a FRESH model + optimiser + scheduler loaded from it, and the remaining three steps.
    Every kernel of the step sums in a fixed order, so the resumed run must land on the same parameters, moments and losses bit for bit."""
    import copy
    import io
    from conan_fgw_amd import ops
    from conan_fgw_amd.collate import DeviceCollator, molecules_from_synthetic
    from conan_fgw_amd.head import EmbeddingsWithGATAggregationBaryCenter
    from conan_fgw_amd.parallel import FlatAdam, FlatGradients
    dev = torch.device("cuda:0")
    K = 3
    sets = []
    for B, seed in [(5, 71), (3, 72), (6, 73), (4, 74), (5, 75), (2, 76), (6, 77)]:
        cb = make_batch("esol", B, K, seed=seed); bg = make_bond_graph(cb, seed=seed + 100)
        sets.append(molecules_from_synthetic(cb, bg))
    coll = DeviceCollator(dev, K, depth=2)

    def build():
        torch.manual_seed(21)
        m = EmbeddingsWithGATAggregationBaryCenter(K, dev).to(dev)
        flat = FlatGradients(m.parameters())
        opt = FlatAdam(flat, lr=2e-3, module=m)
        sched = torch.optim.lr_scheduler.ReduceLROnPlateau(opt, mode="min", patience=0, factor=0.5)
        return m, flat, opt, sched

    def run(m, flat, opt, sched, items_list, losses):
        for items in items_list:
            db = coll(items).wait()
            data, node_index = db.as_model_input()
            flat.zero()
            loss = ops.mse_loss(m(data, db.conformers_index, node_index), db.y[::K][:, None].contiguous())
            flat.backward(loss)
            flat.all_reduce_mean()
            flat.clip_grad_norm_(1.0)
            opt.step()
            losses.append(float(loss.detach()))
            sched.step(losses[-1] if len(losses) % 2 else 1e9)          # (every second "validation" is bad: the rate is cut during the run)

    # straight through
    ma, fa, oa, sa = build()
    la = []
    run(ma, fa, oa, sa, sets, la)
    # interrupted
    mb, fb, ob, sb = build()
    lb = []
    run(mb, fb, ob, sb, sets[:4], lb)
    buf = io.BytesIO()
    torch.save({"model": mb.state_dict(), "optimizer": ob.state_dict(), "scheduler": sb.state_dict()}, buf)
    del mb, fb, ob, sb
    buf.seek(0)
    ck = torch.load(buf, map_location=dev, weights_only=False)
    torch.manual_seed(999)                                                   # (a different initialisation: everything must come from the checkpoint)
    mc = EmbeddingsWithGATAggregationBaryCenter(K, dev).to(dev)
    fc = FlatGradients(mc.parameters())
    oc = FlatAdam(fc, lr=1.0, module=mc)
    sc = torch.optim.lr_scheduler.ReduceLROnPlateau(oc, mode="min", patience=0, factor=0.5)
    mc.load_state_dict(ck["model"], strict=True)
    oc.load_state_dict(ck["optimizer"])
    sc.load_state_dict(ck["scheduler"])
    assert oc.check_aliasing()                                               # load_state_dict copied INTO the aliased tensors
    run(mc, fc, oc, sc, sets[4:], lb)
    torch.cuda.synchronize()
    assert la == lb, (la, lb)
    assert oa.lr == oc.lr and oa.lr < 2e-3 and float(oa.step_dev) == float(oc.step_dev) == 7.0
    for (k, p), (_, q) in zip(ma.named_parameters(), mc.named_parameters()):
        assert torch.equal(p, q), k
    sda, sdc = oa.state_dict(), oc.state_dict()
    for i in sda["state"]:
        assert torch.equal(sda["state"][i]["exp_avg"], sdc["state"][i]["exp_avg"]) and torch.equal(sda["state"][i]["exp_avg_sq"], sdc["state"][i]["exp_avg_sq"])
