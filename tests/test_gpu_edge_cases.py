"""Edge cases through the full drop-in path: tiny conformers (1-3 atoms: rows without neighbours, N_max tiny), K=1 and K=2,
a single molecule, and widely ragged batches (padding rows dominate some slabs).  Checked against the fp64 oracle."""
import numpy as np
import pytest
import torch

from helpers import rel
from conan_fgw_amd.schnet import SchNetNoSum
from conan_fgw_amd.synthetic import ConformerBatch
from oracle.schnet import SchNetNoSumOracle

pytestmark = pytest.mark.gpu
dev = torch.device("cuda:0")


def _batch(atom_counts, K, seed):
    rng = np.random.RandomState(seed)
    zs, poss, bs = [], [], []
    g = 0
    for n in atom_counts:
        z = rng.choice([1, 6, 7, 8], size=n)
        for _ in range(K):
            zs.append(z); poss.append(rng.uniform(0, 3.0 + n ** (1 / 3), size=(n, 3)).astype(np.float32)); bs.append(np.full(n, g)); g += 1
    return ConformerBatch(z=np.concatenate(zs).astype(np.int64), pos=np.concatenate(poss), batch=np.concatenate(bs).astype(np.int64),
                          y=np.zeros(len(atom_counts), np.float32), num_molecules=len(atom_counts), num_conformers=K,
                          atoms_per_molecule=np.asarray(atom_counts, np.int64))


@pytest.mark.parametrize("atoms,K", [([2, 3, 5], 2), ([7], 1), ([4, 30, 2, 17], 3), ([3, 3], 5)])
def test_tiny_and_ragged_batches(atoms, K):
    b = _batch(atoms, K, seed=sum(atoms) + K)
    torch.manual_seed(1)
    m = SchNetNoSum(dev, hidden_channels=64, num_filters=64, num_interactions=2).to(dev)
    ref = SchNetNoSumOracle(64, 64, 2)
    ref.load_state_dict({k: v.detach().cpu() for k, v in m.state_dict().items()}, strict=True)
    ref = ref.double()
    z, pos, batch = torch.from_numpy(b.z), torch.from_numpy(b.pos), torch.from_numpy(b.batch)
    h3, hb = m.forward_w_barycenter(z.to(dev), pos.to(dev), K, batch.to(dev))
    (h3.sum() + hb.sum()).backward()
    r3, rb = ref.forward_w_barycenter(z, pos.double(), K, batch)
    assert h3.shape == (len(atoms) * K, 32) and hb.shape == h3.shape
    assert torch.isfinite(h3).all() and torch.isfinite(hb).all()
    assert rel(h3.detach().cpu().numpy(), r3.detach().numpy()) < 1e-5
    assert rel(hb.detach().cpu().numpy(), rb.detach().numpy()) < 1e-4
    assert all(torch.isfinite(p.grad).all() for p in m.parameters() if p.grad is not None)


def test_isolated_atoms_have_no_edges_and_constant_slab_is_nan_like_the_reference():
    """One-atom conformers: the radius graph is empty and the [1,d] slab is normalised over its own min/max.  A slab that
    is constant divides by zero exactly like the reference (Appendix D-13: normalize_tensor has no epsilon)."""
    from conan_fgw_amd import ops
    pos = torch.tensor([[0., 0, 0], [9, 9, 9]], device=dev); batch = torch.tensor([0, 1], device=dev)
    gp = ops.graph_ptr_from_batch(batch, 2)
    g = ops.RadiusGraph(pos, gp, 2, 10.0, 32)
    assert g.num_edges == 0
    feat = torch.full((2, 8), 0.25, device=dev)
    Ys, Cs = ops.fgw_densify(feat, g, 1, 0.5)
    assert torch.isnan(Ys).all() and float(Cs.abs().max()) == 0.0
