"""Functional boundary of the FGW solver, mirroring the reference's signature
(conan_fgw/src/model/fgw/barycenter.py:7-31 `fgw_barycenters`, :393-399 `normalize_tensor`).

Same argument names, defaults and error behaviour (`ValueError` for unknown `loss_fun` / `stop_criterion` / `solver`,
barycenter.py:33-44).  `loss_fun` = "square_loss" (every model) or "kl_loss" (utils.py:20-32,76-87).  Option values that exist in
the reference but are not reached by any model (`BAPG`, `PPA`, `stop_criterion="loss"` — the latter is broken in the reference
itself, SURVEY.md 8c — and input graphs whose node count differs from N) raise `NotImplementedError`.  Runs on the GPU only.
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Union

import torch
from torch import Tensor

from . import ops


def fgw_barycenters(N, Ys: Sequence[Tensor], Cs: Sequence[Tensor], ps=None, p=None, lambdas=None, loss_fun="square_loss",
                    epsilon=0.1, symmetric=True, alpha=0.5, max_iter=100, tol=1e-9, solver="PGD", stop_criterion="barycenter",
                    warmstartT=False, verbose=False, log=False, init_C=None, init_Y=None, fixed_structure=False,
                    fixed_features=False, seed=0, **kwargs):
    if loss_fun not in ("square_loss", "kl_loss"):
        raise ValueError(f"Unknown `loss_fun='{loss_fun}'`. Use one of: {'square_loss', 'kl_loss'}.")
    if stop_criterion not in ["barycenter", "loss"]:
        raise ValueError(f"Unknown `stop_criterion='{stop_criterion}'`. Use one of: {'barycenter', 'loss'}.")
    if solver not in ["PGD", "PPA", "BAPG"]:
        raise ValueError("Unknown solver '%s'. Pick one in ['PGD', 'PPA', 'BAPG']." % solver)
    if solver != "PGD" or stop_criterion != "barycenter":
        raise NotImplementedError("only solver='PGD', stop_criterion='barycenter' (the path every ConAN model takes, "
                                  "schnet_no_sum.py:281-306) is implemented on this backend")
    if not symmetric:
        raise NotImplementedError("symmetric=False is not reached by any ConAN model")
    method = kwargs.pop("method", "sinkhorn_log")
    if str(method).lower() != "sinkhorn_log":
        raise NotImplementedError("only method='sinkhorn_log' is implemented")
    num_iter_max = int(kwargs.pop("numItermax", 100))          # sinkhorn.py:12
    stop_thr = float(kwargs.pop("stopThr", 1e-5))              # sinkhorn.py:13
    if fixed_structure and init_C is None:
        raise ValueError("If C is fixed it must be initialized")
    if fixed_features and init_Y is None:
        raise ValueError("If Y is fixed it must be initialized")

    Ys_t = torch.stack([y.to(torch.float32) for y in Ys]) if not torch.is_tensor(Ys) else Ys
    Cs_t = torch.stack([c.to(torch.float32) for c in Cs]) if not torch.is_tensor(Cs) else Cs
    K, n, d = Ys_t.shape
    if n != N or Cs_t.shape[1] != N:
        raise NotImplementedError("input graphs must all have N nodes (the ConAN glue pads them, schnet_no_sum.py:242-252)")
    if init_C is None:
        # barycenter.py:61-65: torch.manual_seed(seed); xalea = torch.randn(N, 2); C = dist(xalea, xalea) — a host-side random
        # squared-distance matrix (utils.py:154-171 with X is Y: clamped at 0, zero diagonal).  Reproduced draw for draw,
        # including the reference's re-seeding of the global generator; N x 2 numbers of initialisation, not the solver.
        torch.manual_seed(seed)
        xalea = torch.randn(N, 2)
        a2 = torch.einsum("ij,ij->i", xalea, xalea)
        c0 = -2 * (xalea @ xalea.T)
        c0 += a2[:, None]
        c0 += a2[None, :]
        init_C = (torch.clamp(c0, min=0) * (1 - torch.eye(N))).to(Ys_t.device)
    ps_t = None
    if ps is not None:
        ps_t = (torch.stack(list(ps)) if not torch.is_tensor(ps) else ps).to(torch.float32).view(1, K, N)
    p_t = p.to(torch.float32).view(1, N) if p is not None else None
    lam = None
    if lambdas is not None:
        lam = torch.as_tensor(lambdas, dtype=torch.float32, device=Ys_t.device)
    # adjacency-like inputs (integers in [0, 255]: what to_dense_adj produces) take the byte-wide LDS layout of the N <= 64 kernel
    small_int = bool(((Cs_t == Cs_t.round()) & (Cs_t >= 0) & (Cs_t <= 255)).all())
    res = ops.fgw_barycenter_batched(
        Ys_t.view(1, K, N, d), Cs_t.view(1, K, N, N), ps=ps_t, p=p_t, lambdas=lam,
        init_C=init_C.to(torch.float32).view(1, N, N), init_Y=None if init_Y is None else init_Y.to(torch.float32).view(1, N, d),
        alpha=alpha, epsilon=epsilon, max_iter=max_iter, tol=tol, inner_tol=1e-4, num_iter_max=num_iter_max, stop_thr=stop_thr,
        fixed_structure=fixed_structure, fixed_features=fixed_features, warmstart=warmstartT, loss_fun=loss_fun, keep_iterates=bool(log),
        cs_small_int=small_int)
    Y, C, T, info, errs = res[:5]
    if not log:
        return Y[0], C[0]
    outer = int(info[0, 0].item())
    T_iter = res[5]
    # log["Ms"] (barycenter.py:82,177,220): the feature costs dist(Y, Ys[s]) of the returned barycenter — clamped squared
    # euclidean distances (utils.py:154-171); a by-product for the caller's inspection, formed here from the outputs
    Yd = Y[0]
    y2 = (Yd * Yd).sum(1)
    Ms = [torch.clamp(y2[:, None] + (Ys_t[s] * Ys_t[s]).sum(1)[None, :] - 2.0 * (Yd @ Ys_t[s].T), min=0) for s in range(K)]
    log_ = {"err_feature": [errs[0, 0, i] for i in range(outer)], "err_structure": [errs[0, 1, i] for i in range(outer)],
            "Ts_iter": [[T_iter[i, 0, s] for s in range(K)] for i in range(outer)],           # barycenter.py:196
            "T": [T[0, s] for s in range(K)],
            "p": p if p is not None else torch.ones(N, device=Y.device) / N,
            "Ms": Ms,
            "n_outer": outer, "n_pgd": int(info[0, 1].item()), "n_sinkhorn": int(info[0, 2].item())}
    return Y[0], C[0], log_


def normalize_tensor(tensor: Tensor, a: float, b: float) -> Tensor:
    """a + (t - min) * (b - a) / (max - min) over the whole tensor (barycenter.py:393-399), on the densify kernel."""
    flat = tensor.reshape(1, -1).to(torch.float32).contiguous()
    n = flat.shape[1]
    dev = flat.device
    g = object.__new__(ops.RadiusGraph)
    g.num_graphs, g.num_atoms = 1, 1
    g.graph_ptr = torch.tensor([0, 1], dtype=torch.int32, device=dev)
    g.rowptr = torch.zeros(2, dtype=torch.int32, device=dev)
    g.col = torch.zeros(1, dtype=torch.int32, device=dev)
    Ys, _ = ops.fgw_densify(flat.view(1, n), g, 1, 0.0, a, b)
    return Ys.view(tensor.shape)
