"""Batch assembly through the HIP path (pinned pack -> one H2D copy -> conan_collate_unpack) against the oracle's restatement of the
reference's collate_fn + create_aggregation_index, and the collated batch driven through the stage-2 model."""
import numpy as np
import pytest
import torch

from conan_fgw_amd.collate import DeviceCollator, collate_fn, molecules_from_synthetic
from conan_fgw_amd.synthetic import make_batch, make_bond_graph
from oracle import collate as ocoll

pytestmark = pytest.mark.gpu
dev = torch.device("cuda:0")


@pytest.mark.parametrize("shape,B,K", [("esol", 7, 5), ("lipo", 3, 3), ("freesolv", 4, 20)])
def test_collate_matches_reference_semantics(shape, B, K):
    cb = make_batch(shape, B, K, seed=9); bg = make_bond_graph(cb, seed=10)
    items = molecules_from_synthetic(cb, bg)
    ref = ocoll.collate(items, K)
    data, node_index = collate_fn(items, dev)
    torch.cuda.synchronize()
    for k in ("z", "pos", "x", "batch", "edge_index", "edge_attr", "y", "conf_node_batch"):
        got = getattr(data, k).cpu().numpy()
        assert got.dtype == ref[k].dtype and np.array_equal(got, ref[k]), k                 # bit-exact: integer and copied float data
    assert np.array_equal(node_index.cpu().numpy(), ref["batch_node_index"])
    assert np.array_equal(data.conformers_index.cpu().numpy(), ocoll.aggregation_index(ref["smiles"], K))
    assert data.smiles == ref["smiles"] and len(data.smiles) == B * K                          # common.py:418 counts graphs as len(batch.smiles)
    assert np.array_equal(data.graph_ptr.cpu().numpy(), cb.graph_ptr.astype(np.int32))
    assert (data.num_graphs, data.max_nodes, data.num_molecules) == (B * K, cb.max_nodes, B)


def test_double_buffered_collator_and_model_run():
    """Two batches in flight on the copy stream; the model consumes the collated batch with the host-known hints (no device sync)."""
    import types
    from conan_fgw_amd.head import EmbeddingsWithGATAggregationBaryCenter
    K = 3
    coll = DeviceCollator(dev, K, depth=2)
    batches = []
    for seed in (1, 2, 3):
        cb = make_batch("esol", 4, K, seed=seed); bg = make_bond_graph(cb, seed=seed + 10)
        batches.append((cb, bg, coll(molecules_from_synthetic(cb, bg))))
    torch.manual_seed(0)
    model = EmbeddingsWithGATAggregationBaryCenter(K, dev).to(dev)
    for cb, bg, db in batches:
        db.wait()
        assert torch.equal(db.pos.cpu(), torch.from_numpy(cb.pos))                            # the slot re-use did not clobber an earlier batch
        data, node_index = db.as_model_input()
        cidx = model.create_aggregation_index(data)                                           # the reference's call: the batch itself
        assert torch.equal(cidx, db.conformers_index)
        y1 = model(data, cidx, node_index, num_graphs=db.num_graphs, max_nodes=db.max_nodes)
        t = lambda a: torch.from_numpy(a).to(dev)
        flat = types.SimpleNamespace(z=t(cb.z), pos=t(cb.pos), batch=t(cb.batch), x=t(bg.x), edge_index=t(bg.edge_index), edge_attr=t(bg.edge_attr))
        y2 = model(flat, cidx, flat.batch)
        assert torch.allclose(y1, y2, rtol=1e-5, atol=1e-6)       # same molecules; only the order of the bond edges inside a graph differs


@pytest.mark.parametrize("model_name", ["schnet", "visnet"])
def test_reference_shaped_call_on_a_collated_batch_never_syncs(model_name):
    """The reference calls `model(batch, conformers_index, node_index)` with no size arguments (schnet_based_models.py:135-173).  On a batch from
    DeviceCollator the sizes ride on the index tensors (ops.batch_hints), so forward AND backward run without one device -> host read:
    torch's sync debug mode turns any such read into an error.  Same prediction as the call with explicit hints."""
    from conan_fgw_amd.head import EmbeddingsWithGATAggregationBaryCenter
    K = 3
    cb = make_batch("esol", 5, K, seed=21); bg = make_bond_graph(cb, seed=22)
    torch.manual_seed(1)
    model = EmbeddingsWithGATAggregationBaryCenter(K, dev, model_name=model_name).to(dev)
    data, node_index = DeviceCollator(dev, K, depth=2)(molecules_from_synthetic(cb, bg)).wait().as_model_input()
    cidx = data.conformers_index
    y_hint = model(data, cidx, node_index, num_graphs=data.num_graphs, max_nodes=data.max_nodes).detach().clone()       # (warm-up: lazy allocations, side stream)
    torch.cuda.synchronize()
    prev = torch.cuda.get_sync_debug_mode()
    torch.cuda.set_sync_debug_mode("error")
    try:
        y = model(data, cidx, node_index)                          # the reference's call, unchanged
        y.square().mean().backward()
    finally:
        torch.cuda.set_sync_debug_mode(prev)
    torch.cuda.synchronize()
    assert torch.equal(y.detach(), y_hint)
    grads = [p.grad for p in model.parameters() if p.grad is not None]
    assert len(grads) > 10 and all(bool(torch.isfinite(g).all()) for g in grads)
    # a foreign index tensor (no hints) still works: it pays the reference's two host reads instead
    y_foreign = model(data, cidx, node_index.clone())
    assert torch.equal(y_foreign.detach(), y_hint)


def test_static_collator_keeps_addresses_and_rejects_shape_changes():
    K = 2
    coll = DeviceCollator(dev, K, depth=2, static=True)
    cb = make_batch("esol", 3, K, seed=4); bg = make_bond_graph(cb, seed=5)
    items = molecules_from_synthetic(cb, bg)
    a = coll(items).wait(); p0 = a.pos.data_ptr()
    ref = {k: getattr(a, k).clone() for k in ("z", "pos", "batch", "x", "edge_index", "edge_attr", "y", "graph_ptr")}
    b = coll(items)
    with pytest.raises(RuntimeError, match="wait"):          # one landing copy: the previous batch must have been taken over first
        coll(items)
    b.wait()
    assert b.pos.data_ptr() == p0
    torch.cuda.synchronize()
    assert all(torch.equal(getattr(b, k), ref[k]) for k in ref)
    # a different batch of the SAME shape lands at the same addresses with its own values
    items2 = molecules_from_synthetic(cb, make_bond_graph(cb, seed=5))
    items2[0].pos = items2[0].pos + 1.0
    c = coll(items2).wait()
    torch.cuda.synchronize()
    assert c.pos.data_ptr() == p0 and not torch.equal(c.pos, ref["pos"]) and torch.equal(c.z, ref["z"])
    cb2 = make_batch("esol", 4, K, seed=6); bg2 = make_bond_graph(cb2, seed=7)
    with pytest.raises(RuntimeError, match="shape"):
        coll(molecules_from_synthetic(cb2, bg2))


def _plain_molecule(n, K, seed):
    from conan_fgw_amd.collate import ConformerMolecule
    rng = np.random.default_rng(seed)
    return ConformerMolecule(z=rng.integers(1, 9, n).astype(np.int64), pos=rng.normal(size=(K, n, 3)).astype(np.float32),
                             x=rng.normal(size=(n, 4)).astype(np.float32), edge_index=np.array([[0, 1], [1, 0]], dtype=np.int64),
                             edge_attr=rng.normal(size=(2, 3)).astype(np.float32), y=float(seed))


def test_static_collator_sizes_follow_the_batch_being_consumed():
    """Two batches of EQUAL shape (atoms, bond edges, graphs) and different largest conformer through one static collator: the fixed views
    are one set of tensor objects, so the host-known sizes a size-less call reads (ops.batch_hints) must change when a batch is taken over by
    wait(), not when the next one is enqueued (a worker thread does that while the consumer still runs the previous batch: a smaller
    max_nodes would under-size the dense barycenter padding).  By default a changed largest conformer is rejected (a captured step was sized
    for the first)."""
    from conan_fgw_amd import ops
    K = 2
    big, even = [_plain_molecule(3, K, 1), _plain_molecule(5, K, 2)], [_plain_molecule(4, K, 3), _plain_molecule(4, K, 4)]
    coll = DeviceCollator(dev, K, depth=2, static=True, strict_max_nodes=False)
    a = coll(big).wait()
    assert ops.batch_hints(a.batch) == (4, 5) and a.max_nodes == 5
    b = coll(even)                                          # assembled (here: by the same thread), not yet taken over
    assert b.batch is a.batch                               # the shared fixed view
    assert ops.batch_hints(a.batch) == (4, 5)               # ... still describes the batch being consumed
    b.wait()
    assert ops.batch_hints(b.batch) == (4, 4) and b.max_nodes == 4
    torch.cuda.synchronize()
    assert b.graph_ptr.tolist() == [0, 4, 8, 12, 16]
    strict = DeviceCollator(dev, K, depth=2, static=True)
    strict(big).wait()
    with pytest.raises(RuntimeError, match="largest conformer"):
        strict(even)
    strict(big).wait()                                      # the same size again is fine


def test_pipeline_on_a_worker_thread_yields_the_same_batches_in_order():
    """CollatePipeline: the host half runs on a worker thread `prefetch` batches ahead; the batches arrive in source order and equal what the
    collator produces when called directly (item records are cached on the items: the second epoch takes the cached path)."""
    from conan_fgw_amd.collate import CollatePipeline
    K = 3
    sets = []
    for seed in (3, 4, 5, 6, 7):
        cb = make_batch("esol", 5, K, seed=seed); bg = make_bond_graph(cb, seed=seed + 50)
        sets.append(molecules_from_synthetic(cb, bg))
    direct = DeviceCollator(dev, K, depth=2)
    want = []
    for items in sets:
        b = direct(items).wait()
        torch.cuda.synchronize()
        want.append({k: getattr(b, k).clone() for k in ("z", "pos", "batch", "x", "edge_index", "edge_attr", "y", "graph_ptr")})
    coll = DeviceCollator(dev, K, depth=4)
    with pytest.raises(ValueError):
        CollatePipeline(DeviceCollator(dev, K, depth=2), [], prefetch=2)
    for epoch in range(2):
        got = []
        for b in CollatePipeline(coll, sets, prefetch=2):
            b.wait()
            torch.cuda.synchronize()
            got.append({k: getattr(b, k).clone() for k in want[0]})
        assert len(got) == len(want)
        for g, w in zip(got, want):
            assert all(torch.equal(g[k], w[k]) for k in w)


def test_pipeline_with_a_static_collator_hands_over_one_landing_copy_at_a_time():
    """static=True (captured-step replay): the worker enqueues the expansion of batch i + 1 only after the consumer's wait() on batch i has
    been executed by the GPU (host-side wait, no device-side wait queued ahead); the fixed tensors keep their addresses, every batch arrives
    intact even when the consumer is slow or fast, and close() in mid-stream leaves the collator usable."""
    import copy
    import time
    from conan_fgw_amd.collate import CollatePipeline
    K = 3
    cb = make_batch("esol", 6, K, seed=11); bg = make_bond_graph(cb, seed=61)
    base = molecules_from_synthetic(cb, bg)
    sets = []
    for v in range(6):                                       # same shapes, different coordinates and targets
        items = copy.deepcopy(base)
        for it in items:
            it.pos = (it.pos + np.float32(0.25 * v)).astype(np.float32)
            it.y = float(it.y) + v
        sets.append(items)
    direct = DeviceCollator(dev, K, depth=2)
    want = []
    for items in sets:
        b = direct(items).wait()
        torch.cuda.synchronize()
        want.append({k: getattr(b, k).clone() for k in ("z", "pos", "batch", "x", "edge_index", "edge_attr", "y", "graph_ptr")})
    coll = DeviceCollator(dev, K, depth=4, static=True)
    addr = None
    for delay in (0.0, 0.01):
        feed = CollatePipeline(coll, sets, prefetch=2)
        for i, b in enumerate(feed):
            b.wait()
            if addr is None:
                addr = b.pos.data_ptr()
            assert b.pos.data_ptr() == addr
            got = {k: getattr(b, k).clone() for k in want[0]}       # (on the consumer's stream, behind the landing copy)
            torch.cuda.synchronize()
            assert all(torch.equal(got[k], want[i][k]) for k in got), (delay, i)
            time.sleep(delay)
        assert i == len(sets) - 1
    feed = CollatePipeline(coll, sets, prefetch=2)
    next(feed).wait()
    feed.close()                                              # the worker may hold an assembled batch: it is dropped
    b = coll(sets[3]).wait()
    torch.cuda.synchronize()
    assert torch.equal(b.pos, want[3]["pos"]) and torch.equal(b.y, want[3]["y"])
