"""TEST INFRASTRUCTURE — ctypes front-end of oracle/libconan_oracle.so (the C restatement of the reference's
FGW barycenter, oracle/fgw_oracle_impl.h).  Mirrors the call the reference's glue makes
(conan_fgw/src/model/graph_embeddings/schnet_no_sum.py:281-306)."""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

# literals of schnet_no_sum.py:281-306 == visnet.py:205-230
PROD = dict(alpha=0.1, epsilon=0.1, max_iter=5, tol=1e-2, inner_tol=1e-4, numItermax=5, stopThr=1e-2,
            fixed_structure=False, fixed_features=False)


def build(force: bool = False) -> str:
    so = os.path.join(_HERE, "libconan_oracle.so")
    if force or not os.path.exists(so):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = ctypes.CDLL(build())
        _LIB.conan_oracle_fgw_dist_f32.restype = ctypes.c_float
        _LIB.conan_oracle_fgw_dist_f64.restype = ctypes.c_double
    return _LIB


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p) if a is not None else None


def fgw_barycenter(Ys, Cs, ps=None, lambdas=None, init_C=None, N=None, p=None, dtype=np.float32, **kw):
    """Ys [K,n,d], Cs [K,n,n] -> dict(Y, C, T, err_feature, err_structure, outer, pgd[outer][K], sinkhorn[...])."""
    o = dict(PROD); o.update(kw)
    dt = np.dtype(dtype)
    suf = "_f32" if dt == np.float32 else "_f64"
    cr = ctypes.c_float if dt == np.float32 else ctypes.c_double
    Ys = np.ascontiguousarray(Ys, dt); Cs = np.ascontiguousarray(Cs, dt)
    K, n, d = Ys.shape
    N = int(N or n)
    ps = np.ascontiguousarray(ps, dt) if ps is not None else np.full((K, n), 1.0 / n, dt)
    lambdas = np.ascontiguousarray(lambdas, dt) if lambdas is not None else np.full((K,), 1.0 / K, dt)
    init_C = np.ascontiguousarray(init_C if init_C is not None else Cs[0], dt)
    p_arr = np.ascontiguousarray(p, dt) if p is not None else None
    mi = int(o["max_iter"])
    Y = np.zeros((N, d), dt) if o.get("init_Y") is None else np.ascontiguousarray(o["init_Y"], dt).copy()
    C = np.zeros((N, N), dt); T = np.zeros((K, N, n), dt)
    ef = np.full((mi,), np.nan, dt); es = np.full((mi,), np.nan, dt)
    iters = np.zeros((1 + mi * K * (1 + mi),), np.int32)
    loss = {"square_loss": 0, "kl_loss": 1}[o.get("loss_fun", "square_loss")]
    fn = getattr(lib(), "conan_oracle_fgw_barycenter_loss" + suf)
    rc = fn(ctypes.c_int(N), ctypes.c_int(K), ctypes.c_int(n), ctypes.c_int(d), _p(Ys), _p(Cs), _p(ps), _p(p_arr),
            _p(lambdas), _p(init_C), cr(o["alpha"]), cr(o["epsilon"]), ctypes.c_int(mi), cr(o["tol"]),
            cr(o["inner_tol"]), ctypes.c_int(int(o["numItermax"])), cr(o["stopThr"]),
            ctypes.c_int(int(bool(o["fixed_structure"]))), ctypes.c_int(int(bool(o["fixed_features"])) | (2 if o.get("init_Y") is not None else 0)),
            _p(Y), _p(C), _p(T), _p(ef), _p(es), _p(iters), ctypes.c_int(loss))
    if rc != 0:
        raise RuntimeError("oracle fgw_barycenter failed rc=%d" % rc)
    outer = int(iters[0])
    body = iters[1:].reshape(mi, K, 1 + mi)
    return dict(Y=Y, C=C, T=T, err_feature=ef[:outer], err_structure=es[:outer], outer=outer,
                pgd=body[:outer, :, 0].copy(), sinkhorn=body[:outer, :, 1:].copy(), lambdas=lambdas, ps=ps)


def fgw_barycenter_bwd(T, dY, lambdas=None, p=None, dtype=np.float32):
    dt = np.dtype(dtype); suf = "_f32" if dt == np.float32 else "_f64"
    T = np.ascontiguousarray(T, dt); dY = np.ascontiguousarray(dY, dt)
    K, N, n = T.shape; d = dY.shape[1]
    lambdas = np.ascontiguousarray(lambdas, dt) if lambdas is not None else np.full((K,), 1.0 / K, dt)
    p_arr = np.ascontiguousarray(p, dt) if p is not None else None
    dYs = np.zeros((K, n, d), dt)
    getattr(lib(), "conan_oracle_fgw_barycenter_bwd" + suf)(
        ctypes.c_int(N), ctypes.c_int(K), ctypes.c_int(n), ctypes.c_int(d), _p(T), _p(p_arr), _p(lambdas), _p(dY), _p(dYs))
    return dYs


def fgw_dist(M, C1, C2, T, alpha=0.1, p=None, q=None, dtype=np.float32):
    dt = np.dtype(dtype); suf = "_f32" if dt == np.float32 else "_f64"
    cr = ctypes.c_float if dt == np.float32 else ctypes.c_double
    M = np.ascontiguousarray(M, dt); C1 = np.ascontiguousarray(C1, dt); C2 = np.ascontiguousarray(C2, dt)
    T = np.ascontiguousarray(T, dt)
    n1, n2 = M.shape
    p = np.ascontiguousarray(p, dt) if p is not None else np.full((n1,), 1.0 / n1, dt)
    q = np.ascontiguousarray(q, dt) if q is not None else np.full((n2,), 1.0 / n2, dt)
    return float(getattr(lib(), "conan_oracle_fgw_dist" + suf)(
        ctypes.c_int(n1), ctypes.c_int(n2), _p(M), _p(C1), _p(C2), _p(p), _p(q), _p(T), cr(alpha)))


def normalize_tensor(x, a, b, dtype=np.float32):
    dt = np.dtype(dtype); suf = "_f32" if dt == np.float32 else "_f64"
    cr = ctypes.c_float if dt == np.float32 else ctypes.c_double
    x = np.ascontiguousarray(x, dt); out = np.empty_like(x)
    getattr(lib(), "conan_oracle_normalize_tensor" + suf)(ctypes.c_long(x.size), _p(x), cr(a), cr(b), _p(out))
    return out
