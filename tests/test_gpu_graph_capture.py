"""The whole training step (forward, backward, gradient pack, fused Adam, the second stream of the covalent branch) is
HIP-graph capturable: no host synchronisation, no allocation outside the capture pool, edge counts stay on the device.
Replaying the captured step must reproduce the eager run bit for bit (every reduction has a fixed order)."""
import types

import pytest
import torch

from conan_fgw_amd.synthetic import make_batch, make_bond_graph

pytestmark = pytest.mark.gpu


def _setup(seed):
    from conan_fgw_amd.head import EmbeddingsWithGATAggregationBaryCenter
    from conan_fgw_amd.parallel import FlatGradients
    dev = torch.device("cuda:0")
    K = 3
    b = make_batch("esol", 8, K, seed=51)
    g = make_bond_graph(b, seed=52)
    t = lambda a: torch.from_numpy(a).to(dev)
    data = types.SimpleNamespace(z=t(b.z), pos=t(b.pos), batch=t(b.batch), x=t(g.x), edge_index=t(g.edge_index), edge_attr=t(g.edge_attr))
    y = t(b.y)[:, None]
    torch.manual_seed(seed)
    model = EmbeddingsWithGATAggregationBaryCenter(K, dev).to(dev)
    cidx = model.create_aggregation_index(b.num_graphs, dev)
    flat = FlatGradients(model.parameters())
    opt = torch.optim.Adam(flat.params, lr=1e-3, fused=True, capturable=True)
    loss_out = torch.zeros((), device=dev)

    def step():
        flat.zero()
        pred = model(data, cidx, data.batch, num_graphs=b.num_graphs, max_nodes=b.max_nodes)   # hints: no host sync in the step
        loss = torch.nn.functional.mse_loss(pred, y)
        loss.backward()
        flat.all_reduce_mean()
        opt.step()
        loss_out.copy_(loss.detach())
    return model, step, loss_out


def test_training_step_replays_from_a_hip_graph_bit_for_bit():
    warm, replays = 3, 3
    m_e, step_e, loss_e = _setup(7)
    for _ in range(warm + replays):
        step_e()
    torch.cuda.synchronize()

    m_g, step_g, loss_g = _setup(7)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(warm):
            step_g()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        step_g()
    for _ in range(replays):
        graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(loss_e, loss_g), (float(loss_e), float(loss_g))
    for (k, a), (_, c) in zip(m_e.state_dict().items(), m_g.state_dict().items()):
        assert torch.equal(a, c), k
    assert float(loss_g) == float(loss_g) and float(loss_g) > 0.0


def test_captured_train_step_follows_the_eager_loop_bit_for_bit():
    """conan_fgw_amd.capture.CapturedTrainStep — the stage-2 training step as two HIP graphs (zero / forward / backward / pack | clip + FlatAdam), fed
    with shape-stable batches through a static DeviceCollator — against the same loop run eagerly: six steps over three batches of one shape (the same
    molecules, coordinates jittered: the edge counts differ and stay on the device), the learning rate cut by the scheduler half way.  Same losses,
    same parameters, bit for bit: the captured launches follow the new batches (fixed addresses) and the new rate (a device scalar)."""
    import copy
    import numpy as np
    from conan_fgw_amd import ops
    from conan_fgw_amd.capture import CapturedTrainStep
    from conan_fgw_amd.collate import DeviceCollator, molecules_from_synthetic
    from conan_fgw_amd.head import EmbeddingsWithGATAggregationBaryCenter
    from conan_fgw_amd.parallel import FlatAdam, FlatGradients
    dev = torch.device("cuda:0")
    K = 3
    cb = make_batch("esol", 8, K, seed=81); bg = make_bond_graph(cb, seed=82)
    base = molecules_from_synthetic(cb, bg)
    rng = np.random.default_rng(5)
    batches = []
    for _ in range(3):
        its = [copy.copy(it) for it in base]
        for it in its:
            it.__dict__.pop("_conan_record", None)
            it.pos = (it.pos + rng.normal(0.0, 0.05, size=it.pos.shape)).astype(np.float32)
        batches.append(its)
    order = [0, 1, 2, 1, 0, 2]

    def build():
        torch.manual_seed(31)
        m = EmbeddingsWithGATAggregationBaryCenter(K, dev).to(dev)
        flat = FlatGradients(m.parameters())
        opt = FlatAdam(flat, lr=2e-3, module=m)
        return m, flat, opt

    # eager: three warm-up steps on batch 0 (what the captured object does before it captures), then the six steps
    me, fe, oe = build()
    coll_e = DeviceCollator(dev, K, depth=2)
    losses_e = []

    def eager(items):
        db = coll_e(items).wait()
        fe.zero()
        loss = ops.mse_loss(me(db, db.conformers_index, db.batch), db.y[::K][:, None].contiguous())
        fe.backward(loss)
        fe.all_reduce_mean()
        fe.clip_grad_norm_(1.0)
        oe.step()
        return float(loss.detach())
    for _ in range(3):
        eager(batches[0])
    for t, q in enumerate(order):
        if t == 3:
            oe.param_groups[0]["lr"] = 1e-3
        losses_e.append(eager(batches[q]))
    torch.cuda.synchronize()

    # captured
    mg, fg, og = build()
    coll = DeviceCollator(dev, K, depth=2, static=True)
    db = coll(batches[0]).wait()
    y_view = db.y                                                       # fixed views of the static collator
    step = CapturedTrainStep(lambda: ops.mse_loss(mg(db, db.conformers_index, db.batch), y_view[::K][:, None].contiguous()), fg, og, clip_norm=1.0, warmup=3)
    losses_g = []
    for t, q in enumerate(order):
        if t == 3:
            og.param_groups[0]["lr"] = 1e-3
        coll(batches[q]).wait()
        losses_g.append(float(step()))
    torch.cuda.synchronize()
    assert losses_e == losses_g, (losses_e, losses_g)
    for (k, a), (_, c) in zip(me.state_dict().items(), mg.state_dict().items()):
        assert torch.equal(a, c), k
    assert float(oe.step_dev) == float(og.step_dev) == 9.0
