"""Pins the CPU oracle (oracle/fgw_oracle_impl.h) against the reference's own outputs:
 - cfm_log.npz: the reference's only known-answer fixture (notebooks/data/cfm_log.pt);
 - fgw_ref_*.npz: the reference's fgw_barycenters run in fp32/fp64 in the build container
   (tests/golden/make_fgw_golden.py).
The f64 oracle must match ref64 tightly (same algorithm, same control flow, same iteration counts); the f32
oracle must match ref32 within the reference's own fp32 noise floor."""
import glob
import os

import numpy as np
import pytest

from oracle import fgw

CASES = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "fgw_ref_*.npz")))


def rel(a, b):
    return float(np.linalg.norm(np.asarray(a, np.float64) - np.asarray(b, np.float64)) / np.linalg.norm(np.asarray(b, np.float64)))


def test_cfm_log_known_answer(golden_dir):
    g = np.load(os.path.join(golden_dir, "cfm_log.npz"))
    # hyper-parameters that reproduce the stored answer: dimenet.py:235-260 (alpha=0.5, fixed_structure=True)
    for dt, tol in ((np.float32, 1e-4), (np.float64, 1e-4)):
        r = fgw.fgw_barycenter(g["Ys"], g["Cs"], g["ps"], g["lambdas"], g["Cs"][0], alpha=0.5, fixed_structure=True, dtype=dt)
        assert r["outer"] == len(g["err_feature"]) == 5
        assert np.abs(r["Y"] - g["F_bary"]).max() < tol          # stored vs re-run of the reference itself: 2.1e-5
        assert np.array_equal(r["C"].astype(np.float32), g["C_bary"])      # structure fixed: C_bary == Cs[0]
        np.testing.assert_allclose(r["err_feature"], g["err_feature"], rtol=2e-4)
        assert rel(r["T"], g["T"]) < 2e-3


def test_cfm_log_pins_glue(golden_dir):
    """Ys[k] == normalize_tensor(node_feature[k] + 0.5, 0.1, 2.0) exactly (schnet_no_sum.py:59,66)."""
    g = np.load(os.path.join(golden_dir, "cfm_log.npz"))
    nf = g["node_feature"].reshape(10, 22, 3)
    for k in range(10):
        out = fgw.normalize_tensor(nf[k] + np.float32(0.5), 0.1, 2.0)
        assert np.abs(out - g["Ys"][k]).max() <= 2.4e-7


@pytest.mark.parametrize("N", [40, 70, 96])
def test_oracle_takes_an_initial_feature_matrix_like_the_reference(N, golden_dir):
    """init_Y given (barycenter.py:78-80), with a barycenter node ~95 e-folds above every column's best cost: the reference's log-domain
    Sinkhorn gives that node its full mass (row sum 1/N).  Until round 4 the oracle silently started from Y = 0 whatever init_Y said."""
    g = np.load(os.path.join(golden_dir, f"fgw_inity_n{N}.npz"))
    i0 = int(g["i0"])
    assert abs(g["r64_T"][0, i0].sum() * N - 1.0) < 1e-9
    r = fgw.fgw_barycenter(g["Ys"], g["Cs"], dtype=np.float64, init_Y=g["init_Y"], max_iter=int(g["max_iter"]))
    assert np.array_equal(r["pgd"], g["r64_pgd"]) and np.array_equal(r["sinkhorn"][..., : g["r64_sinkhorn"].shape[-1]], g["r64_sinkhorn"])
    assert rel(r["T"], g["r64_T"]) < 1e-9 and rel(r["T"][:, i0], g["r64_T"][:, i0]) < 1e-9
    assert rel(r["Y"], g["r64_Y"]) < 1e-9 and rel(r["C"], g["r64_C"]) < 1e-9
    r0 = fgw.fgw_barycenter(g["Ys"], g["Cs"], dtype=np.float64, max_iter=int(g["max_iter"]))          # Y = 0 start: a different problem
    assert rel(r0["T"][:, i0], g["r64_T"][:, i0]) > 0.1
    r32 = fgw.fgw_barycenter(g["Ys"], g["Cs"], dtype=np.float32, init_Y=g["init_Y"], max_iter=int(g["max_iter"]))
    assert rel(r32["T"], g["r32_T"]) < 1e-4


@pytest.mark.parametrize("path", CASES, ids=[os.path.basename(p)[8:-4] for p in CASES])
def test_oracle_f64_matches_ref64(path):
    g = np.load(path)
    r = fgw.fgw_barycenter(g["Ys"], g["Cs"], dtype=np.float64)
    assert r["outer"] == len(g["r64_err_feature"])
    assert np.array_equal(r["pgd"], g["r64_pgd"])
    assert np.array_equal(r["sinkhorn"][..., : g["r64_sinkhorn"].shape[-1]], g["r64_sinkhorn"])
    assert rel(r["Y"], g["r64_Y"]) < 1e-9
    assert rel(r["C"], g["r64_C"]) < 1e-9
    assert rel(r["T"], g["r64_T"]) < 1e-8
    np.testing.assert_allclose(r["err_feature"], g["r64_err_feature"], rtol=1e-8, atol=1e-12)
    np.testing.assert_allclose(r["err_structure"], g["r64_err_structure"], rtol=1e-8, atol=1e-12)
    # backward identity against the reference's autograd
    dYs = fgw.fgw_barycenter_bwd(r["T"], g["r32_grad_w"].astype(np.float64), dtype=np.float64)
    assert rel(dYs, g["r64_dYs"]) < 1e-6
    # fgw_dist at the final barycenter (bregman.py:163-164)
    import numpy as _np
    K, N, d = g["Ys"].shape
    for s in range(K):
        Y = r["Y"]; Ys = g["Ys"][s].astype(_np.float64)
        M = _np.maximum((Y * Y).sum(1)[:, None] + (Ys * Ys).sum(1)[None, :] - 2 * Y @ Ys.T, 0)
        fd = fgw.fgw_dist(M, r["C"], g["Cs"][s], r["T"][s], alpha=0.1, dtype=_np.float64)
        assert abs(fd - g["r64_fgw_dist"][s]) <= 1e-6 * abs(g["r64_fgw_dist"][s])


@pytest.mark.parametrize("path", CASES, ids=[os.path.basename(p)[8:-4] for p in CASES])
def test_oracle_f32_within_reference_noise_floor(path):
    """Appendix F protocol: err(oracle32, ref32) <= 1e-4 or err(oracle32, ref64) <= 3 * err(ref32, ref64)."""
    g = np.load(path)
    r = fgw.fgw_barycenter(g["Ys"], g["Cs"], dtype=np.float32)
    assert r["outer"] == len(g["r32_err_feature"])
    for key in ("Y", "C"):
        yard = rel(g["r32_" + key], g["r64_" + key])
        e32 = rel(r[key], g["r32_" + key]); e64 = rel(r[key], g["r64_" + key])
        assert e32 <= 1e-4 or e64 <= 3 * yard + 1e-6, (key, e32, e64, yard)
    ro, rr = r["Y"].sum(0), g["r64_Y"].sum(0)      # the readout the model consumes (schnet_no_sum.py:308)
    assert rel(ro, rr) < 1e-4


def test_oracle_kl_loss_matches_reference_goldens():
    """loss_fun="kl_loss" (utils.py:20-32,76-87): the f64 restatement reproduces the reference's fp64 run (same iteration counts,
    1e-9 on Y / C / T); the f32 build stays inside the reference's own fp32-vs-fp64 spread (Appendix-F protocol)."""
    import glob
    files = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "fgw_kl_*.npz")))
    assert len(files) == 4
    for f in files:
        g = np.load(f)
        Ys, Cs = g["Ys"], g["Cs"].astype(np.float32)
        o64 = fgw.fgw_barycenter(Ys, Cs, dtype=np.float64, loss_fun="kl_loss")
        assert o64["outer"] == len(g["r64_err_feature"])
        assert np.array_equal(o64["pgd"], g["r64_pgd"])
        assert np.array_equal(o64["sinkhorn"][:, :, : g["r64_sinkhorn"].shape[2]], g["r64_sinkhorn"])
        for k in ("Y", "C", "T"):
            assert rel(o64[k], g["r64_" + k]) < 1e-9, (f, k)
        o32 = fgw.fgw_barycenter(Ys, Cs, dtype=np.float32, loss_fun="kl_loss")
        for k in ("Y", "C"):
            yard = rel(g["r32_" + k].astype(np.float64), g["r64_" + k])
            assert rel(o32[k].astype(np.float64), g["r32_" + k].astype(np.float64)) <= 1e-4 or \
                rel(o32[k].astype(np.float64), g["r64_" + k]) <= 2.0 * yard + 1e-6, (f, k)


RECT = [p for p in sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "fgw_rect_*.npz"))) if "ragged" not in p]


@pytest.mark.parametrize("path", RECT, ids=[os.path.basename(p)[9:-4] for p in RECT])
def test_oracle_rectangular_problems_match_the_reference(path):
    """Input graphs with n != N nodes (barycenter.py:50-67): the restatement is written for N x n couplings throughout; the reference run on
    the same lists (make_fgw_golden.py rect) pins it — same iteration counts, fp64 results to 1e-9.  The start is the reference's own seeded
    draw (barycenter.py:61-65), reproduced here with torch's generator."""
    import torch
    g = np.load(path)
    N, n = int(g["N"]), int(g["sizes"][0])
    torch.manual_seed(int(g["seed"]))
    xa = torch.randn(N, 2).double().numpy()
    a2 = (xa * xa).sum(1)
    c0 = np.maximum(a2[:, None] + a2[None, :] - 2 * xa @ xa.T, 0) * (1 - np.eye(N))
    r = fgw.fgw_barycenter(g["Ys"][:, :n], g["Cs"][:, :n, :n], N=N, init_C=c0, dtype=np.float64)
    assert r["outer"] == len(g["r64_err_feature"]) and np.array_equal(r["pgd"], g["r64_pgd"])
    assert np.array_equal(r["sinkhorn"][..., : g["r64_sinkhorn"].shape[-1]], g["r64_sinkhorn"])
    assert rel(r["Y"], g["r64_Y"]) < 1e-7 and rel(r["C"], g["r64_C"]) < 1e-7 and rel(r["T"], g["r64_T"][:, :, :n]) < 1e-7
