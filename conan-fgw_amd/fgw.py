"""Functional boundary of the FGW solver, mirroring the reference's signature
(conan_fgw/src/model/fgw/barycenter.py:7-31 `fgw_barycenters`, :393-399 `normalize_tensor`).

Same argument names, defaults and error behaviour (`ValueError` for unknown `loss_fun` / `stop_criterion` / `solver`,
barycenter.py:33-44).  `loss_fun` = "square_loss" (every model) or "kl_loss" (utils.py:20-32,76-87).  Option values that exist in
the reference but are not reached by any model (`BAPG`, `PPA`, `stop_criterion="loss"` — the latter is broken in the reference
itself) raise `NotImplementedError`.  Input graphs of any size (n_s != N, ragged lists) are solved by embedding them in a
square problem with massless nodes (below).  Runs on the GPU only.
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

    N = int(N)
    Ys_l = [y.to(torch.float32) for y in (Ys.unbind(0) if torch.is_tensor(Ys) else Ys)]
    Cs_l = [c.to(torch.float32) for c in (Cs.unbind(0) if torch.is_tensor(Cs) else Cs)]
    K, d = len(Ys_l), Ys_l[0].shape[1]
    sizes = [int(y.shape[0]) for y in Ys_l]
    if len(Cs_l) != K or any(tuple(c.shape) != (n, n) for c, n in zip(Cs_l, sizes)):
        raise ValueError("Cs[s] must be a square matrix over the nodes of Ys[s]")
    # Input graphs whose node counts differ from N or from each other (barycenter.py:50-67 takes any; no ConAN model does this: the glue pads
    # every conformer to N, schnet_no_sum.py:242-252).  The kernels solve square problems, so the call is embedded in one of size
    # Np = max(N, max n_s): the extra nodes carry NO mass (p_i = 0 / ps[s]_j = 0), zero features and no edges.  A massless node's row / column of
    # every coupling is exactly zero in the Sinkhorn scaling (u_i = p_i / (K v)_i), it adds nothing to any product, and the barycenter update
    # keeps its row of Y and its row / column of C at zero (fgw_small.hip: the divisions by p are guarded) — the leading N x N / N x n_s blocks
    # are the reference's rectangular problem, term for term.
    Np = max([N] + sizes)
    embedded = Np != N or any(n != N for n in sizes)
    dev = Ys_l[0].device
    if embedded:
        def pad(t, *shape):
            out = torch.zeros(*shape, dtype=torch.float32, device=dev)
            out[tuple(slice(0, k) for k in t.shape)] = t
            return out
        ps_l = [torch.ones(n, device=dev) / n for n in sizes] if ps is None else [q.to(torch.float32).to(dev) for q in (ps.unbind(0) if torch.is_tensor(ps) else ps)]
        p_full = (torch.ones(N, device=dev) / N) if p is None else p.to(torch.float32).to(dev)
        Ys_l = [pad(y, Np, d) for y in Ys_l]
        Cs_l = [pad(c, Np, Np) for c in Cs_l]
        ps = [pad(q, Np) for q in ps_l]
        p_embedded = pad(p_full, Np)
    Ys_t, Cs_t = torch.stack(Ys_l), torch.stack(Cs_l)
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
    N_user = N
    if embedded:
        ic = torch.zeros(Np, Np, dtype=torch.float32, device=dev); ic[:N, :N] = init_C.to(torch.float32)
        init_C = ic
        if init_Y is not None:
            iy = torch.zeros(Np, d, dtype=torch.float32, device=dev); iy[:N] = init_Y.to(torch.float32)
            init_Y = iy
        N = Np
    ps_t = None
    if ps is not None:
        ps_t = (torch.stack(list(ps)) if not torch.is_tensor(ps) else ps).to(torch.float32).view(1, K, N)
    p_t = p_embedded.view(1, N) if embedded else (p.to(torch.float32).view(1, N) if p is not None else None)
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
    if embedded:
        Y, C = Y[:, :N_user], C[:, :N_user, :N_user]
    if not log:
        return Y[0], C[0]
    outer = int(info[0, 0].item())
    T_iter = res[5]
    # log["Ms"] (barycenter.py:82,177,220): the feature costs dist(Y, Ys[s]) of the returned barycenter — clamped squared
    # euclidean distances (utils.py:154-171); a by-product for the caller's inspection, formed here from the outputs
    Yd = Y[0]
    y2 = (Yd * Yd).sum(1)
    Yin = [Ys_t[s, :sizes[s]] for s in range(K)]              # (an embedded call: the caller's own nodes)
    Ms = [torch.clamp(y2[:, None] + (Yin[s] * Yin[s]).sum(1)[None, :] - 2.0 * (Yd @ Yin[s].T), min=0) for s in range(K)]
    log_ = {"err_feature": [errs[0, 0, i] for i in range(outer)], "err_structure": [errs[0, 1, i] for i in range(outer)],
            "Ts_iter": [[T_iter[i, 0, s, :N_user, :sizes[s]] for s in range(K)] for i in range(outer)],           # barycenter.py:196
            "T": [T[0, s, :N_user, :sizes[s]] for s in range(K)],
            "p": p if p is not None else torch.ones(N_user, device=Y.device) / N_user,
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
