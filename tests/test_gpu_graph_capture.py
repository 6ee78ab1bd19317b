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
