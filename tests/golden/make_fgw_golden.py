"""Generates tests/golden/fgw_ref_*.npz and tests/golden/cfm_log.npz.

RUNS ONLY IN THE BUILD CONTAINER (needs /root/reference).  It imports the reference's own FGW solver
(conan_fgw/src/model/fgw/*.py: pure torch, importable as-is) and records, for seeded synthetic inputs shaped
like the production call (schnet_no_sum.py:234-306), the reference's outputs in fp32 ("ref32") and in fp64
("ref64", same code, double inputs: SURVEY.md Appendix F).  The fixtures are data (inputs + expected outputs);
no reference source is copied.

    PYTHONPATH=/root/reference PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_fgw_golden.py
"""
import os
import sys
import warnings

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")
warnings.filterwarnings("ignore")

from conan_fgw.src.model.fgw import barycenter as ref_bary  # noqa: E402
from conan_fgw.src.model.fgw import bregman as ref_breg  # noqa: E402
from conan_fgw.src.model.fgw import sinkhorn as ref_sk  # noqa: E402
from conan_fgw_amd.synthetic import make_batch  # noqa: E402

PROD = dict(warmstartT=True, symmetric=True, method="sinkhorn_log", alpha=0.1, solver="PGD", fixed_structure=False,
            fixed_features=False, epsilon=0.1, p=None, loss_fun="square_loss", max_iter=5, tol=1e-2, numItermax=5,
            stopThr=1e-2, verbose=False, log=True, init_X=None, random_state=None)   # schnet_no_sum.py:281-306


class Counter:
    """Counts PGD and Sinkhorn iterations by wrapping the reference's own call sites."""

    def __init__(self):
        self.calls = []          # one entry per fgw() call: list of sinkhorn iteration counts
        self._lse = 0
        self._orig_lse = torch.logsumexp
        self._orig_sinkhorn = ref_breg.sinkhorn
        self._orig_fgw = ref_bary.fgw

    def __enter__(self):
        c = self

        def lse(*a, **k):
            c._lse += 1
            return c._orig_lse(*a, **k)

        def sinkhorn(*a, **k):
            c._lse = 0
            out = c._orig_sinkhorn(*a, **k)
            c.calls[-1].append(c._lse // 2)
            return out

        def fgw(*a, **k):
            c.calls.append([])
            return c._orig_fgw(*a, **k)

        torch.logsumexp = lse
        ref_breg.sinkhorn = sinkhorn
        ref_bary.fgw = fgw
        return self

    def __exit__(self, *exc):
        torch.logsumexp = self._orig_lse
        ref_breg.sinkhorn = self._orig_sinkhorn
        ref_bary.fgw = self._orig_fgw


def ssp(x):
    return np.logaddexp(x, 0.0) - np.log(2.0)


def make_inputs(seed, K, n_real, n_pad, d, r, shift=0.5):
    """Production-shaped inputs: K conformers of one molecule with n_real atoms padded to N=n_real+n_pad rows."""
    rng = np.random.RandomState(seed)
    b = make_batch("esol", 1, K, seed=seed, fixed_atoms=n_real)
    pos = b.pos.reshape(K, n_real, 3)
    N = n_real + n_pad
    Cs = np.zeros((K, N, N), np.float32)
    for k in range(K):
        diff = pos[k][:, None] - pos[k][None]
        d2 = (diff * diff).sum(-1)
        adj = (d2 < np.float32(r * r)) & ~np.eye(n_real, dtype=bool)
        Cs[k, :n_real, :n_real] = adj
    h = ssp(rng.normal(0, 0.7, size=(K, n_real, d))).astype(np.float32)   # post-activation node features
    dense = np.zeros((K, N, d), np.float32)
    dense[:, :n_real] = h
    Ys = np.stack([ref_bary.normalize_tensor(torch.from_numpy(x + np.float32(shift)), 0.1, 2.0).numpy() for x in dense])
    return Ys.astype(np.float32), Cs


def run_ref(Ys, Cs, dtype, _init_first=True, **over):
    K, N, d = Ys.shape
    t = lambda a: torch.from_numpy(np.asarray(a)).to(dtype)
    args = dict(PROD); args.update(over)
    if args.get("init_Y") is not None:
        args["init_Y"] = t(args["init_Y"])
    Ysl = [t(y).requires_grad_(True) for y in Ys]
    Csl = [t(c) for c in Cs]
    ps = [torch.ones(N, dtype=dtype) / N for _ in range(K)]
    lambdas = torch.ones(K, dtype=dtype) / K
    with Counter() as cnt:
        Y, C, log = ref_bary.fgw_barycenters(N=N, Ys=Ysl, Cs=Csl, ps=ps, lambdas=lambdas, init_C=Csl[0] if _init_first else None, **args)
    outer = len(log["err_feature"])
    mi = args["max_iter"]
    pgd = np.zeros((outer, K), np.int32); sk = np.zeros((outer, K, mi), np.int32)
    for o in range(outer):
        for s in range(K):
            c = cnt.calls[o * K + s]
            pgd[o, s] = len(c); sk[o, s, : len(c)] = c
    # fgw_dist of every input graph to the final barycenter, evaluated AT the final coupling T[s]
    # (bregman.py:163-164; max_iter=0 => no further PGD step is taken, the log is computed for G0 = T[s])
    fd = []
    with torch.no_grad():
        for s in range(K):
            _, lg = ref_breg.fgw(log["Ms"][s], C, Csl[s], log["p"], ps[s], args["loss_fun"], args["epsilon"], True, args["alpha"],
                                 log["T"][s], 0, 1e-4, solver="PGD", method="sinkhorn_log", log=True,
                                 numItermax=args["numItermax"], stopThr=args["stopThr"])
            fd.append(float(lg["fgw_dist"]))
    # autograd check vector: d(sum(w*Y))/dYs for a fixed w
    gw = torch.from_numpy(np.random.RandomState(7).normal(size=tuple(Y.shape))).to(dtype)
    if Y.requires_grad:
        (Y * gw).sum().backward()
        dYs = np.stack([y.grad.numpy() for y in Ysl])
    else:
        dYs = np.zeros_like(Ys)
    return dict(Y=Y.detach().numpy(), C=C.detach().numpy(), T=np.stack([x.detach().numpy() for x in log["T"]]),
                err_feature=np.array([float(e) for e in log["err_feature"]]),
                err_structure=np.array([float(e) for e in log["err_structure"]]),
                pgd=pgd, sinkhorn=sk, fgw_dist=np.array(fd), dYs=dYs, grad_w=gw.numpy())


CASES = [
    # name, seed, K, n_real, n_pad, d, r
    ("k5_n9_d3", 11, 5, 9, 0, 3, 10.0),
    ("k5_n20_d64", 12, 5, 20, 0, 64, 10.0),
    ("k5_n20p4_d64", 13, 5, 20, 4, 64, 10.0),
    ("k5_n26_d64", 14, 5, 22, 4, 64, 10.0),
    ("k5_n33_d64", 15, 5, 33, 0, 64, 10.0),
    ("k3_n15p5_d64", 16, 3, 15, 5, 64, 10.0),
    ("k10_n18p2_d64", 17, 10, 18, 2, 64, 10.0),
    ("k20_n16p4_d64", 18, 20, 16, 4, 64, 10.0),
    ("k5_n24_d64_r5", 19, 5, 24, 0, 64, 5.0),
    ("k5_n30p3_d64_r5", 20, 5, 30, 3, 64, 5.0),
    ("k5_n20p6_d64_visnet", 21, 5, 20, 6, 64, 5.0),   # shift +1.0 (visnet.py:50)
    ("k5_n12_d256", 22, 5, 12, 0, 256, 10.0),
]


def main():
    out = {}
    for name, seed, K, n_real, n_pad, d, r in CASES:
        shift = 1.0 if "visnet" in name else 0.5
        Ys, Cs = make_inputs(seed, K, n_real, n_pad, d, r, shift)
        r32 = run_ref(Ys, Cs, torch.float32)
        r64 = run_ref(Ys, Cs, torch.float64)
        rec = dict(Ys=Ys, Cs=Cs.astype(np.uint8))
        for tag, rr in (("r32", r32), ("r64", r64)):
            for k, v in rr.items():
                if tag == "r64" and k == "grad_w":
                    continue
                rec[f"{tag}_{k}"] = v.astype(np.float32) if (tag == "r32" and v.dtype.kind == "f") else v
        np.savez_compressed(os.path.join(HERE, f"fgw_ref_{name}.npz"), **rec)
        rel = lambda a, b: float(np.linalg.norm(a - b) / np.linalg.norm(b))
        print(f"{name}: outer32={len(r32['err_feature'])} outer64={len(r64['err_feature'])} "
              f"relY={rel(r32['Y'], r64['Y']):.2e} relC={rel(r32['C'], r64['C']):.2e} relT={rel(r32['T'], r64['T']):.2e} "
              f"pgd32={r32['pgd'].sum()} pgd64={r64['pgd'].sum()} sk32={r32['sinkhorn'].sum()} sk64={r64['sinkhorn'].sum()}")

    # the reference's only known-answer fixture, re-exported as plain arrays (data, not source)
    dd = torch.load("/root/reference/notebooks/data/cfm_log.pt", weights_only=True, map_location="cpu")
    np.savez_compressed(
        os.path.join(HERE, "cfm_log.npz"),
        N=np.int64(dd["N"]), Ys=torch.stack(list(dd["Ys"])).numpy(), Cs=torch.stack(list(dd["Cs"])).numpy(),
        ps=torch.stack(list(dd["ps"])).numpy(), lambdas=dd["lambdas"].numpy(), F_bary=dd["F_bary"].numpy(),
        C_bary=dd["C_bary"].numpy(), err_feature=np.array([float(e) for e in dd["log"]["err_feature"]]),
        err_structure=np.array([float(e) for e in dd["log"]["err_structure"]]),
        T=torch.stack(list(dd["log"]["T"])).numpy(), Ms=torch.stack(list(dd["log"]["Ms"])).numpy(),
        Ts_iter=np.stack([torch.stack(list(t)).numpy() for t in dd["log"]["Ts_iter"]]),
        batch=dd["batch"].numpy(), edge_index=dd["edge_index"].numpy(), node_feature=dd["node_feature"].numpy())
    print("cfm_log.npz written")


KL_CASES = [c for c in CASES if c[0] in ("k5_n9_d3", "k5_n20p4_d64", "k3_n15p5_d64", "k5_n30p3_d64_r5")]


def main_kl():
    """loss_fun="kl_loss" (utils.py:20-32,76-87): same inputs as the square-loss cases, written to fgw_kl_*.npz.
    Run as `python make_fgw_golden.py kl`; the square-loss fixtures are not touched."""
    for name, seed, K, n_real, n_pad, d, r in KL_CASES:
        Ys, Cs = make_inputs(seed, K, n_real, n_pad, d, r, 0.5)
        r32 = run_ref(Ys, Cs, torch.float32, loss_fun="kl_loss")
        r64 = run_ref(Ys, Cs, torch.float64, loss_fun="kl_loss")
        rec = dict(Ys=Ys, Cs=Cs.astype(np.uint8))
        for tag, rr in (("r32", r32), ("r64", r64)):
            for k, v in rr.items():
                if tag == "r64" and k == "grad_w":
                    continue
                rec[f"{tag}_{k}"] = v.astype(np.float32) if (tag == "r32" and v.dtype.kind == "f") else v
        np.savez_compressed(os.path.join(HERE, f"fgw_kl_{name}.npz"), **rec)
        rel = lambda a, b: float(np.linalg.norm(a - b) / np.linalg.norm(b))
        print(f"kl {name}: outer32={len(r32['err_feature'])} outer64={len(r64['err_feature'])} relY={rel(r32['Y'], r64['Y']):.2e} "
              f"relC={rel(r32['C'], r64['C']):.2e} relT={rel(r32['T'], r64['T']):.2e} finite={np.isfinite(r64['C']).all()}")


def starved_row_inputs(N, K=2, d=16, i0=5, seed=11):
    """Inputs whose barycenter node i0 starts ~95 e-folds (at eps = 0.1, alpha = 0.1) above every column's best cost: near-identical features
    everywhere, init_Y[i0] displaced by |.|^2 = 10.5.  In an fp32 kernel matrix that row is all denormals (tests/test_gpu_fgw.py)."""
    rng = np.random.default_rng(seed)
    Ys = (0.6 + 0.01 * rng.standard_normal((K, N, d))).astype(np.float32)
    Y0 = (0.6 + 0.01 * rng.standard_normal((N, d))).astype(np.float32)
    Y0[i0] += np.float32(np.sqrt(10.5 / d))
    A = (rng.random((K, N, N)) < 0.3).astype(np.float32); Cs = np.triu(A, 1); Cs = Cs + Cs.transpose(0, 2, 1)
    return Ys, Cs, Y0


def main_inity():
    """init_Y given (barycenter.py:78-80) with a starved row, one outer iteration: run as `python make_fgw_golden.py inity`; writes fgw_inity_*.npz."""
    for N in (40, 70, 96):
        Ys, Cs, Y0 = starved_row_inputs(N)
        rec = dict(Ys=Ys, Cs=Cs.astype(np.uint8), init_Y=Y0, i0=np.int64(5), max_iter=np.int64(1))
        for tag, dt in (("r32", torch.float32), ("r64", torch.float64)):
            rr = run_ref(Ys, Cs, dt, init_Y=Y0, max_iter=1)
            for k in ("Y", "C", "T", "err_feature", "err_structure", "pgd", "sinkhorn"):
                rec[f"{tag}_{k}"] = rr[k].astype(np.float32) if (tag == "r32" and rr[k].dtype.kind == "f") else rr[k]
        np.savez_compressed(os.path.join(HERE, f"fgw_inity_n{N}.npz"), **rec)
        rel = lambda a, b: float(np.linalg.norm(a - b) / np.linalg.norm(b))
        print(f"inity n{N}: relY={rel(rec['r32_Y'], rec['r64_Y']):.2e} relT={rel(rec['r32_T'], rec['r64_T']):.2e} row i0 * N = {rec['r64_T'][0, 5].sum() * N:.6f} "
              f"pgd={rec['r64_pgd'].ravel()} sk={rec['r64_sinkhorn'].ravel()}")


def main_randinit():
    """init_C=None: the reference draws its own initial structure (barycenter.py:61-65: torch.manual_seed(seed); randn(N, 2); dist).
    Run as `python make_fgw_golden.py randinit`; writes fgw_randinit_*.npz only."""
    for name, seed, K, n_real, n_pad, d, r in [c for c in CASES if c[0] in ("k5_n9_d3", "k3_n15p5_d64")]:
        Ys, Cs = make_inputs(seed, K, n_real, n_pad, d, r, 0.5)
        rec = dict(Ys=Ys, Cs=Cs.astype(np.uint8), seed=np.int64(3))
        for tag, dt in (("r32", torch.float32), ("r64", torch.float64)):
            rr = run_ref(Ys, Cs, dt, _init_first=False, seed=3)
            for k in ("Y", "C", "T", "err_feature", "err_structure", "pgd", "sinkhorn"):
                rec[f"{tag}_{k}"] = rr[k].astype(np.float32) if (tag == "r32" and rr[k].dtype.kind == "f") else rr[k]
        np.savez_compressed(os.path.join(HERE, f"fgw_randinit_{name}.npz"), **rec)
        rel = lambda a, b: float(np.linalg.norm(a - b) / np.linalg.norm(b))
        print(f"randinit {name}: outer={len(rec['r64_err_feature'])} relY={rel(rec['r32_Y'], rec['r64_Y']):.2e} relC={rel(rec['r32_C'], rec['r64_C']):.2e}")


RECT_CASES = [
    # name, seed, N (barycenter nodes), sizes of the input graphs, d
    ("N6_n9", 31, 6, (9, 9, 9), 8),
    ("N10_n7", 32, 10, (7, 7, 7, 7), 8),
    ("N6_ragged", 33, 6, (5, 8, 7), 8),
    ("N40_n33", 34, 40, (33, 33), 16),
    ("N70_n80", 35, 70, (80, 80), 16),
]


def main_rect():
    """Input graphs whose node counts differ from N and from each other (barycenter.py:50-67 takes any): `python make_fgw_golden.py rect`
    writes fgw_rect_*.npz — inputs zero-padded to the largest graph with their sizes beside them, the reference's outputs as they are.
    init_C=None: the reference draws its own N x N start (seed 3)."""
    for name, seed, N, sizes, d in RECT_CASES:
        K, nmax = len(sizes), max(sizes)
        Ysp = np.zeros((K, nmax, d), np.float32); Csp = np.zeros((K, nmax, nmax), np.float32)
        for s, n in enumerate(sizes):
            y, c = make_inputs(seed + 100 * s, 1, n, 0, d, 10.0)
            Ysp[s, :n] = y[0]; Csp[s, :n, :n] = c[0]
        rec = dict(Ys=Ysp, Cs=Csp.astype(np.uint8), sizes=np.array(sizes, np.int64), N=np.int64(N), seed=np.int64(3))
        for tag, dt in (("r32", torch.float32), ("r64", torch.float64)):
            t = lambda a: torch.from_numpy(np.asarray(a)).to(dt)
            Ysl = [t(Ysp[s, :n]) for s, n in enumerate(sizes)]
            Csl = [t(Csp[s, :n, :n]) for s, n in enumerate(sizes)]
            ps = [torch.ones(n, dtype=dt) / n for n in sizes]
            lambdas = torch.ones(K, dtype=dt) / K
            args = dict(PROD); args["seed"] = 3
            with Counter() as cnt:
                Y, C, log = ref_bary.fgw_barycenters(N=N, Ys=Ysl, Cs=Csl, ps=ps, lambdas=lambdas, init_C=None, **args)
            outer = len(log["err_feature"])
            mi = args["max_iter"]
            pgd = np.zeros((outer, K), np.int32); sk = np.zeros((outer, K, mi), np.int32)
            for o in range(outer):
                for s in range(K):
                    c = cnt.calls[o * K + s]
                    pgd[o, s] = len(c); sk[o, s, : len(c)] = c
            T = np.zeros((K, N, nmax)); 
            for s, n in enumerate(sizes):
                T[s, :, :n] = log["T"][s].detach().numpy()
            rr = dict(Y=Y.detach().numpy(), C=C.detach().numpy(), T=T, err_feature=np.array([float(e) for e in log["err_feature"]]),
                      err_structure=np.array([float(e) for e in log["err_structure"]]), pgd=pgd, sinkhorn=sk)
            for k, v in rr.items():
                rec[f"{tag}_{k}"] = v.astype(np.float32) if (tag == "r32" and v.dtype.kind == "f") else v
        np.savez_compressed(os.path.join(HERE, f"fgw_rect_{name}.npz"), **rec)
        rel = lambda a, b: float(np.linalg.norm(a - b) / np.linalg.norm(b))
        print(f"rect {name}: outer={len(rec['r64_err_feature'])} relY={rel(rec['r32_Y'], rec['r64_Y']):.2e} relC={rel(rec['r32_C'], rec['r64_C']):.2e} "
              f"pgd={rec['r64_pgd'].sum()} sk={rec['r64_sinkhorn'].sum()}")


if __name__ == "__main__":
    mode = sys.argv[1] if len(sys.argv) > 1 else ""
    {"kl": main_kl, "randinit": main_randinit, "inity": main_inity, "rect": main_rect}.get(mode, main)()
