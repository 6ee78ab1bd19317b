"""Host-side logic that needs no GPU: module surface, error behaviour, synthetic data, no-CPU-fallback guarantees."""
import numpy as np
import pytest
import torch

from helpers import golden_files, golden_state_dict
from conan_fgw_amd import fgw as pfgw
from conan_fgw_amd.schnet import SchNetNoSum
from conan_fgw_amd.synthetic import CONFIGS, make_batch, make_config


def test_state_dict_matches_reference_class():
    g = np.load(golden_files("schnet_ref_b4_k5_h128")[0])
    m = SchNetNoSum(torch.device("cpu"), hidden_channels=128, num_filters=128, num_interactions=3)
    sd = golden_state_dict(g)
    assert set(m.state_dict().keys()) == set(sd.keys())
    m.load_state_dict(sd, strict=True)                     # train_val.py:182-183 loads stage-1 checkpoints strictly
    for k, v in m.state_dict().items():
        assert v.shape == sd[k].shape
    assert sum(p.numel() for p in m.parameters()) == 254976
    assert m.interactions[0].mlp[0].weight is m.interactions[0].conv.nn[0].weight     # PyG registers the filter MLP twice
    assert m.hidden_channels == 128                        # read by schnet_based_models.py:95


def test_no_cpu_fallback():
    m = SchNetNoSum(torch.device("cpu"), hidden_channels=32, num_filters=32, num_interactions=1)
    z = torch.ones(4, dtype=torch.long); pos = torch.rand(4, 3)
    for fn in (lambda: m(z, pos), lambda: m.forward_3d_bary(z, pos), lambda: m.forward_w_barycenter(z, pos, 1)):
        with pytest.raises(RuntimeError, match="GPU only"):
            fn()


def test_unsupported_constructor_options():
    with pytest.raises(NotImplementedError):
        SchNetNoSum(torch.device("cpu"), use_covalent=True)
    with pytest.raises(NotImplementedError):
        SchNetNoSum(torch.device("cpu"), interaction_graph=lambda p, b: None)


def test_fgw_barycenters_error_behaviour():
    Ys = [torch.zeros(3, 2)]; Cs = [torch.zeros(3, 3)]
    with pytest.raises(ValueError, match="loss_fun"):
        pfgw.fgw_barycenters(3, Ys, Cs, loss_fun="nope")
    with pytest.raises(ValueError, match="stop_criterion"):
        pfgw.fgw_barycenters(3, Ys, Cs, stop_criterion="nope")
    with pytest.raises(ValueError, match="solver"):
        pfgw.fgw_barycenters(3, Ys, Cs, solver="nope")
    with pytest.raises(ValueError, match="fixed"):
        pfgw.fgw_barycenters(3, Ys, Cs, fixed_structure=True)
    with pytest.raises(NotImplementedError):
        pfgw.fgw_barycenters(3, Ys, Cs, solver="BAPG", init_C=Cs[0])
    with pytest.raises(RuntimeError, match="GPU only"):          # kl_loss is implemented; CPU tensors are refused (no CPU path)
        pfgw.fgw_barycenters(3, Ys, Cs, loss_fun="kl_loss", init_C=Cs[0])


def test_synthetic_batches_are_deterministic_and_shaped():
    a, b = make_batch("esol", 8, 5, seed=3), make_batch("esol", 8, 5, seed=3)
    assert np.array_equal(a.pos, b.pos) and np.array_equal(a.z, b.z)
    assert a.num_graphs == 40 and a.batch.max() == 39 and np.all(np.diff(a.batch) >= 0)
    assert a.z.min() >= 1
    gp = a.graph_ptr
    for m in range(8):                                     # K conformers share n and z, differ in pos
        n = a.atoms_per_molecule[m]
        zs = [a.z[gp[m * 5 + k]:gp[m * 5 + k + 1]] for k in range(5)]
        assert all(len(q) == n and np.array_equal(q, zs[0]) for q in zs)
    assert 6 <= a.atoms_per_molecule.min() and a.atoms_per_molecule.max() <= 33
    for name in CONFIGS:
        c = make_config(name, num_molecules=2)
        assert c.num_conformers == CONFIGS[name][2]


def test_visnet_state_dict_matches_reference_class():
    from conan_fgw_amd.visnet import ViSNet
    g = np.load(golden_files("visnet_ref_b2_k3_h32")[0])
    m = ViSNet(torch.device("cpu"), hidden_channels=32)
    sd = {k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sd:")}
    assert set(m.state_dict().keys()) == set(sd.keys())
    m.load_state_dict(sd, strict=True)
    assert sum(p.numel() for p in ViSNet(torch.device("cpu"), hidden_channels=128).parameters()) == 1798472
    with pytest.raises(RuntimeError, match="GPU only"):
        m(torch.ones(3, dtype=torch.long), torch.rand(3, 3), torch.zeros(3, dtype=torch.long))


def test_get_model_mirrors_the_reference_factory():
    """EquivModelsHolder.get_model(name, device, **kwargs) (common.py:469-546): same names, keyword meaning and literals."""
    import types
    import conan_fgw_amd
    from conan_fgw_amd.gat import GATBased
    from conan_fgw_amd.head import EquivModelsHolder, EmbeddingsWithGATAggregationBaryCenter
    from conan_fgw_amd.visnet import ViSNet
    cpu = torch.device("cpu")
    m = EquivModelsHolder.get_model("schnet", cpu, feat_dim=128)                      # common.py:524-529 (EquivAggregation :400-402)
    assert isinstance(m, SchNetNoSum) and (m.hidden_channels, m.num_filters, m.num_gaussians, m.num_interactions, m.cutoff) == (128, 128, 50, 3, 10.0)
    m = conan_fgw_amd.get_model("schnet", cpu, feat_dim=512, cutoff=10.0)             # :513-522 (EquivAggregationClassification :444-446)
    assert (m.hidden_channels, m.num_filters, m.num_gaussians, m.num_interactions, m.cutoff) == (512, 256, 10, 3, 10.0)
    assert isinstance(conan_fgw_amd.get_model("visnet", cpu, feat_dim=128), ViSNet)
    g = conan_fgw_amd.get_model("gat", cpu, feat_dim=128)
    assert isinstance(g, GATBased) and g.gat_conv2.out_channels == 64
    with pytest.raises(ValueError):
        conan_fgw_amd.get_model("dimenet", cpu, feat_dim=128)
    # load_dummy (model/utils.py:23-33) calls forward_dummy on a CPU mini-batch before the DDP wrap: a harmless no-op here
    model = EmbeddingsWithGATAggregationBaryCenter(5, cpu)
    batch = types.SimpleNamespace(z=torch.ones(4, dtype=torch.long), pos=torch.rand(4, 3), x=torch.zeros(4, 9), edge_index=torch.zeros(2, 0, dtype=torch.long),
                                  edge_attr=torch.zeros(0, 3), batch=torch.zeros(4, dtype=torch.long), smiles=["C"] * 10)
    idx = model.create_aggregation_index(batch)                                      # the reference's argument: the collated batch
    assert idx.tolist() == [0] * 5 + [1] * 5 and idx.dtype == torch.long
    assert model.forward_dummy(batch, idx, batch.batch) is None


def test_bench_config_shorthand_and_explicit_flags():
    """bench.py --config names BASELINE.json's configurations as the per-GPU workload; a flag given beside it wins (the 8-rank dry run of cfg3 uses
    --batch 16)."""
    import importlib.util, os
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"))
    bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
    a = bench.parse(["--config", "cfg3"])
    assert (a.shape, a.batch, a.conformers, a.model) == ("lipo", 128, 5, "schnet")
    a = bench.parse(["--config", "cfg3", "--batch", "16"])
    assert (a.shape, a.batch) == ("lipo", 16)
    a = bench.parse(["--config", "cfg4"])
    assert (a.shape, a.batch, a.model, a.head) == ("bace", 64, "visnet", "classification")          # BASELINE configs[3]: BACE classification + ViSNet
    a = bench.parse(["--config", "cfg5"])
    assert (a.shape, a.batch, a.conformers) == ("freesolv", 64, 20)
    a = bench.parse(["--config", "cfg3", "--batch", "256"])                                             # an explicit flag that equals the parser's default still wins
    assert (a.shape, a.batch) == ("lipo", 256)
    a = bench.parse(["--config", "cfg4", "--model=schnet", "--conformers", "5"])
    assert (a.shape, a.batch, a.model, a.conformers) == ("bace", 64, "schnet", 5)
    a = bench.parse([])
    assert (a.shape, a.batch, a.conformers, a.model, a.gpus) == ("esol", 256, 5, "schnet", 1)          # BASELINE configs[1] on one GPU


def test_slice_rule_of_the_batched_weight_gradients():
    """ops._late_slices: ~1 400 workgroups per launch of <= 24 jobs, a multiple of 8 per job, never more than the library's default (one per 128
    rows: the workspace and the reducer are sized by it); the forced knob obeys the same cap."""
    from conan_fgw_amd import ops
    assert ops._late_slices(22, 25275) == 64                     # cfg2's backward pass
    assert ops._late_slices(40, 15000) == 56                     # more than 24 jobs: the launch is cut at 24
    assert ops._late_slices(22, 1280) == 0                       # graph-level layers (default 10 slices): the default stays
    assert ops._late_slices(1, 25275) == 0                       # a lone job: 1 400 > its default 198
    keep = ops.LATE_SLICES
    try:
        ops.LATE_SLICES = 96
        assert ops._late_slices(22, 25275) == 96 and ops._late_slices(22, 1280) == 0
    finally:
        ops.LATE_SLICES = keep
    ops.LATE_SLICES_AUTO = False
    try:
        assert ops._late_slices(22, 25275) == 0
    finally:
        ops.LATE_SLICES_AUTO = True


def test_fused_gather_is_chosen_by_the_estimated_filter_tensor_size():
    """schnet._filter_tensor_outgrows_cache: host-side estimate atoms x min(cap, atoms per conformer - 1) / 2 pair rows of 4F bytes against 192 MiB —
    cfg2 (25 k atoms in 1 280 conformers: 123 MB) keeps the two kernels, a Lipophilicity shard (29.6 k atoms in 640: 242 MB) takes the fused one."""
    import types
    from conan_fgw_amd import schnet
    cfg2 = types.SimpleNamespace(num_atoms=25275, num_graphs=1280, cap=32)
    lipo = types.SimpleNamespace(num_atoms=29600, num_graphs=640, cap=32)
    assert not schnet._filter_tensor_outgrows_cache(cfg2, 128)
    assert schnet._filter_tensor_outgrows_cache(lipo, 128)
    assert not schnet._filter_tensor_outgrows_cache(types.SimpleNamespace(num_atoms=0, num_graphs=0, cap=32), 128)


def test_batch_hints_are_absent_on_foreign_tensors():
    from conan_fgw_amd import ops
    t = torch.zeros(5, dtype=torch.int64)
    assert ops.batch_hints(t) == (None, None) and ops.batch_hints(None) == (None, None)
    t._conan_hints = (3, 7)
    assert ops.batch_hints(t) == (3, 7) and ops.batch_hints(t.clone()) == (None, None)      # the tag does not travel with copies


def test_captured_step_and_flat_adam_refuse_cpu_buffers():
    """The graph-captured step and the one-launch Adam exist on the GPU only; on CPU buffers they fail loudly instead of falling back."""
    import pytest
    import torch
    from conan_fgw_amd.capture import CapturedTrainStep
    from conan_fgw_amd.parallel import FlatAdam, FlatGradients
    lin = torch.nn.Linear(4, 3)
    flat = FlatGradients(lin.parameters())
    with pytest.raises(RuntimeError, match="GPU only"):
        FlatAdam(flat)
    with pytest.raises(RuntimeError, match="GPU only"):
        CapturedTrainStep(lambda: lin(torch.ones(2, 4)).sum(), flat, torch.optim.SGD(lin.parameters(), lr=0.1))
