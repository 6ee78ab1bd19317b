"""Design study (not product, not oracle): which stages of the FGW barycenter need more than fp32 so that the
result lands on the ref64 side of the reference's own fp32 noise floor?
Emulates per-stage precision with torch dtypes on the golden inputs."""
import glob, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
f32, f64 = torch.float32, torch.float64

def lse(x, dim):
    return torch.logsumexp(x, dim)

def sinkhorn(a, b, M, reg, n_it, thr, dt_pot, split_exp=False):
    # Mr, u, v kept in dt_pot
    Mr = (-M.to(dt_pot) / reg)
    u = torch.zeros_like(a, dtype=dt_pot); v = torch.zeros_like(b, dtype=dt_pot)
    la, lb = torch.log(a.to(dt_pot)), torch.log(b.to(dt_pot))
    for ii in range(n_it):
        v = lb - lse(Mr + u[:, None], 0)
        u = la - lse(Mr + v[None, :], 1)
        if ii % 10 == 0:
            tmp = torch.exp(Mr + u[:, None] + v[None, :]).sum(0)
            if torch.norm(tmp - b.to(dt_pot)) < thr: break
    return torch.exp(Mr + u[:, None] + v[None, :])

def fgw_bary(Ys, Cs, dtM, dtMM, dtSK, dtUP, dtT=f32, alpha=0.1, eps=0.1):
    K, N, d = Ys.shape
    Ys = torch.from_numpy(Ys); Cs = torch.from_numpy(Cs.astype(np.float32))
    p = torch.full((N,), 1.0 / N, dtype=f64)
    C = Cs[0].clone().to(dtUP); Y = torch.zeros(N, d, dtype=dtUP)
    def dist(Y, Z, dt):
        Y = Y.to(dt); Z = Z.to(dt)
        c = -2 * (Y @ Z.T); c = c + (Y * Y).sum(1)[:, None]; c = c + (Z * Z).sum(1)[None, :]
        return c.clamp(min=0)
    Ms = [dist(Y, Ys[s], dtM) for s in range(K)]
    T = [None] * K
    ef = es = 1e15; cpt = 0
    while (ef > 1e-2 or es > 1e-2) and cpt < 5:
        Cp, Yp = C, Y
        for s in range(K):
            C1 = C.to(dtMM); C2 = Cs[s].to(dtMM); pp = p.to(dtMM)
            constC = ((C1 * C1) @ pp)[:, None] + ((C2 * C2) @ pp)[None, :]
            Ts = T[s] if T[s] is not None else torch.outer(p, p).to(dtT)
            err = 1; it = 0
            while err > 1e-4 and it < 5:
                Tprev = Ts
                A = -(C1 @ Ts.to(dtMM)) @ (2 * C2).T
                tens = alpha * 2 * (constC + A) + (1 - alpha) * Ms[s].to(dtMM)
                Ts = sinkhorn(p, p, tens.to(dtSK), eps, 5, 1e-2, dtSK).to(dtT)
                if it % 10 == 0: err = torch.norm(Ts.double() - Tprev.double())
                it += 1
            T[s] = Ts
        Y = sum([(1.0 / K) * ((T[s].to(dtUP) @ Ys[s].to(dtUP)) / p.to(dtUP)[:, None]) for s in range(K)])
        Ms = [dist(Y, Ys[s], dtM) for s in range(K)]
        C = sum([(1.0 / K) * (T[s].to(dtUP) @ Cs[s].to(dtUP) @ T[s].to(dtUP).T) for s in range(K)]) / torch.outer(p, p).to(dtUP)
        ef = float(torch.norm(Y.double() - Yp.double())); es = float(torch.norm(C.double() - Cp.double())); cpt += 1
    return Y.double().numpy(), C.double().numpy(), torch.stack(T).double().numpy()

rel = lambda a, b: float(np.linalg.norm(a - b) / np.linalg.norm(b))
configs = {
    "all32": (f32, f32, f32, f32),
    "sk64": (f32, f32, f64, f32),
    "M64+sk64": (f64, f32, f64, f32),
    "M64+mm64+sk64": (f64, f64, f64, f32),
    "M64+sk64+up64": (f64, f32, f64, f64),
    "all64(T32)": (f64, f64, f64, f64),
}
names = sys.argv[1:] or ["k5_n26_d64", "k5_n24_d64_r5", "k5_n30p3_d64_r5", "k5_n20p6_d64_visnet", "k5_n20p4_d64", "k10_n18p2_d64"]
for nm in names:
    g = np.load(os.path.join(ROOT, "tests/golden/fgw_ref_%s.npz" % nm))
    print(f"== {nm}: yardstick ref32-vs-ref64  Y={rel(g['r32_Y'], g['r64_Y']):.1e} C={rel(g['r32_C'], g['r64_C']):.1e} T={rel(g['r32_T'], g['r64_T']):.1e} ro={rel(g['r32_Y'].sum(0), g['r64_Y'].sum(0)):.1e}")
    for cn, cfg in configs.items():
        Y, C, T = fgw_bary(g["Ys"], g["Cs"], *cfg)
        print(f"   {cn:16s} vs ref64: Y={rel(Y, g['r64_Y']):.1e} C={rel(C, g['r64_C']):.1e} T={rel(T, g['r64_T']):.1e} ro={rel(Y.sum(0), g['r64_Y'].sum(0)):.1e}")
