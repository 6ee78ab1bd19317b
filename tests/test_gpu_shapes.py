"""GPU parity at the other shapes the reference instantiates: classification SchNet (H=512, F=256, Gs=10, d=256;
common.py:444-446,513-522) takes the generic kernels (tiled linear, composed filter, F=256 CFConv, d=256 FGW); K=20
conformers (BASELINE.json configs[4]); cap-32 truncation (Lipophilicity-sized conformers, configs[2])."""
import numpy as np
import pytest
import torch

from helpers import rel
from conan_fgw_amd.schnet import SchNetNoSum
from conan_fgw_amd.synthetic import make_batch
from oracle.schnet import SchNetNoSumOracle

pytestmark = pytest.mark.gpu
dev = torch.device("cuda:0")


def _pair(H, F, Gs, seed):
    torch.manual_seed(seed)
    m = SchNetNoSum(dev, hidden_channels=H, num_filters=F, num_interactions=3, num_gaussians=Gs).to(dev)
    with torch.no_grad():
        for p in m.parameters():
            p.add_(0.03 * torch.randn_like(p))
    ref = SchNetNoSumOracle(H, F, 3, Gs)
    ref.load_state_dict({k: v.detach().cpu() for k, v in m.state_dict().items()}, strict=True)
    return m, ref.double()


@pytest.mark.parametrize("shape,B,K,H,F,Gs", [("esol", 2, 5, 512, 256, 10), ("freesolv", 2, 20, 128, 128, 50), ("lipo", 2, 5, 128, 128, 50)])
def test_other_reference_shapes(shape, B, K, H, F, Gs):
    b = make_batch(shape, B, K, seed=91)
    m, ref = _pair(H, F, Gs, 3)
    z, pos, batch = torch.from_numpy(b.z), torch.from_numpy(b.pos), torch.from_numpy(b.batch)
    h3, hb = m.forward_w_barycenter(z.to(dev), pos.to(dev), K, batch.to(dev), num_graphs=b.num_graphs, max_nodes=b.max_nodes)
    loss = h3.square().mean() + hb.square().mean()
    loss.backward()
    r3, rb = ref.forward_w_barycenter(z, pos.double(), K, batch)
    (r3.square().mean() + rb.square().mean()).backward()
    assert rel(h3.detach().cpu().numpy(), r3.detach().numpy()) < 1e-5
    assert rel(hb.detach().cpu().numpy(), rb.detach().numpy()) < 1e-4
    refp = dict(ref.named_parameters())
    for name, p in m.named_parameters():
        q = refp[name]
        if q.grad is None:
            continue
        assert rel(p.grad.cpu().numpy(), q.grad.numpy()) < 1e-4, name
