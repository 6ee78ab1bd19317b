"""Random-shape sweep of the batched FGW solve on COMPLETE input graphs (round 6: the row-sum form of k_fgw_coupling_fast) against the fp64 C oracle:
N 3..64, K 1..7, d 1..100, real-node counts 2..N, uniform random features.  Prints iteration-count mismatches and the worst Y / C / T distances."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from conan_fgw_amd import ops
from oracle import fgw as ofgw
dev = torch.device("cuda:0")
rng = np.random.RandomState(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
cases = int(sys.argv[2]) if len(sys.argv) > 2 else 60
rel = lambda a, b: float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))
bad, worst, mols, flagged = 0, {"Y": 0.0, "C": 0.0, "T": 0.0}, 0, 0
for c in range(cases):
    N, K, d, B = int(rng.randint(3, 65)), int(rng.randint(1, 8)), int(rng.randint(1, 101)), int(rng.randint(1, 4))
    Ys = np.full((B, K, N, d), 0.5, np.float32); Cs = np.zeros((B, K, N, N), np.float32)
    for b in range(B):
        n = int(rng.randint(2, N + 1))
        Ys[b, :, :n] = rng.uniform(0.1, 2.0, size=(K, n, d)); Cs[b, :, :n, :n] = 1.0 - np.eye(n, dtype=np.float32)
    for small in (True, False):
        Y, C, T, info, _ = ops.fgw_barycenter_batched(torch.from_numpy(Ys).to(dev), torch.from_numpy(Cs).to(dev), cs_small_int=small)
        for b in range(B):
            ref = ofgw.fgw_barycenter(Ys[b], Cs[b], dtype=np.float64)
            mols += 1; flagged += int(info[b, 3]) & 1
            ok = int(info[b, 0]) == ref["outer"] and int(info[b, 1]) == int(ref["pgd"].sum()) and int(info[b, 2]) == int(ref["sinkhorn"].sum())
            if not ok:
                bad += 1; print(f"count mismatch: N={N} K={K} d={d} b={b} small={small}: {info[b].tolist()} vs {ref['outer']}, {int(ref['pgd'].sum())}, {int(ref['sinkhorn'].sum())}")
            for k, v in (("Y", Y), ("C", C), ("T", T)):
                e = rel(v[b].cpu().numpy().astype(np.float64), ref[k]); worst[k] = max(worst[k], e)
                if not np.isfinite(e): bad += 1; print("non-finite", N, K, d, b, k)
print(f"{cases} shapes, {mols} molecule solves (both layouts): {bad} mismatches; couplings on the exact second pass in {flagged} molecules; worst distance to the fp64 oracle Y {worst['Y']:.2e} C {worst['C']:.2e} T {worst['T']:.2e}")
