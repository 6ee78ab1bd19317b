// FGW coupling solve for N <= 64 barycenter nodes (every ESOL/FreeSolv-shaped batch): register-resident Sinkhorn.
//
// Same algorithm and citations as fgw.hip (bregman.py:70-167, sinkhorn.py:318-450, utils.py:39-95).  What changes is the
// mapping to the CU.  A 256-thread workgroup (4 wavefronts) owns one (molecule, input graph) problem:
//
//   lane  <-> column j (layout A)  /  lane <-> row i (layout B),      wavefront w <-> index residue (w + 4r), r < R
//
// Every thread keeps its R entries of the N x N cost matrix Mr in registers in BOTH layouts, so that the column
// log-sum-exp (reduce over rows) and the row log-sum-exp (reduce over columns) are each a serial loop over registers
// followed by a 4-way combine through LDS: no cross-lane shuffles of fp64 values, two barriers per half-iteration.
// The two N^3 products of the gradient use the same mapping with T, C1 (fp64), C2 resident in LDS, one operand
// broadcast per FMA.  At the end the workgroup also forms its contribution to the barycenter update
// (T_s @ Ys_s and T_s @ Cs_s @ T_s^T) while T is still in LDS; the update kernel is then a K-term elementwise sum.
#include "fgw_common.h"

namespace {

// KL = true: loss_fun = "kl_loss" (utils.py:20-32,76-87): f1(a) = a log(a + 1e-15) - a, f2(b) = b, h2(b) = log(b + 1e-15) in the
// gradient, log(clamp(C_s, 1e-15)) in the structure update (exp applied by the update kernel).  The logarithms are evaluated in
// fp64 where the operand is fetched (the kl path is a capability of the signature, not a tuned path); KL = false compiles to
// exactly the square-loss kernel.
template <int R, bool KL>
__global__ void __launch_bounds__(FGW_THREADS, R <= 9 ? 3 : 1) k_fgw_coupling_small(
    const float *__restrict__ Ys, const float *__restrict__ Cs, const float *__restrict__ ps, const float *__restrict__ pb,
    FgwDims D, conan_fgw_params prm, int outer, int y_zero, const double *__restrict__ Cw, const double *__restrict__ Yw,
    const int *__restrict__ active, float *__restrict__ Tw, int *__restrict__ info, double *__restrict__ Ypart,
    double *__restrict__ Cpart) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int b = blockIdx.x / D.K, s = blockIdx.x % D.K;
    if (!active[b]) return;
    const int N = D.N, P = D.P, d = D.d;
    const int NN = N * N, NP = N * P;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const bool lane_ok = lane < N;

    // ---- LDS carve (doubles first)
    double *Al = reinterpret_cast<double *>(smem);            // [N,P]  A = C1 @ T ; dot(Y,Z) ; T @ C2
    double *Gl = Al + NP;                                      // [N,P]  G = A @ (2 C2)^T
    double *C1l = Gl + NP;                                     // [N,P]
    // Sinkhorn scratch (6 x [4][64] doubles) aliases the two product buffers, which are dead while the Sinkhorn loop runs;
    // for tiny problems (N*P < 768) it gets its own region behind the matrices.
    double *pq = C1l + NP;                                     // [2][64] p, q
    double *red = pq + 128;                                    // [8]: block reductions use 5; the last word is a write sink
    float *t_dummy = reinterpret_cast<float *>(red + 7);
    const bool alias = NP >= 768;
    double *own = red + 8;
    double *pm = alias ? Gl : own;                             // [4][64] partial max
    double *psum = pm + 256;                                   // [4][64] partial sums
    double *pm2 = psum + 256;                                  // [4][64] second partial buffers (row half-iteration)
    double *psum2 = alias ? Al : pm2 + 256;                    // [4][64]
    double *us = psum2 + 256;                                  // [4][64] per-wave copy of u (indexed by i)
    double *vs = us + 256;                                     // [4][64] per-wave copy of v (indexed by j)
    float *Tl = reinterpret_cast<float *>(own + (alias ? 0 : 6 * 256));     // [N,P]
    float *C2l = Tl + NP;                                      // [N,P]

    const float *Z = Ys + ((size_t)b * D.K + s) * N * d;
    const float *C2 = Cs + ((size_t)b * D.K + s) * NN;
    const double *C1 = Cw + (size_t)b * NN;
    const double *Y = Yw + (size_t)b * N * d;
    float *Tg = Tw + ((size_t)b * D.K + s) * NN;
    const double alpha = (double)prm.alpha, inv_eps = 1.0 / (double)prm.epsilon;

    // ---- stage p, q, C1, C2, Y, Z (coalesced)
    if (tid < 64) {
        pq[tid] = tid < N ? (pb ? (double)pb[(size_t)b * N + tid] : 1.0 / (double)N) : 1.0;
        pq[64 + tid] = tid < N ? (ps ? (double)ps[((size_t)b * D.K + s) * N + tid] : 1.0 / (double)N) : 1.0;
    }
    for (int t = tid; t < NN; t += FGW_THREADS) {
        const int i = t / N, j = t - i * N;
        C1l[i * P + j] = C1[t];
        C2l[i * P + j] = C2[t];
    }
    __syncthreads();
    const double loga = log(pq[lane]), logb = log(pq[64 + lane]);      // lane <-> i for loga, lane <-> j for logb
    const double qj = pq[64 + lane];
    for (int t = tid; t < NN; t += FGW_THREADS) {
        const int i = t / N, j = t - i * N;
        Tl[i * P + j] = (outer > 0 && prm.warmstart) ? Tg[t] : (float)(pq[i] * pq[64 + j]);     // bregman.py:98-101
    }
    // ---- per-index vectors: r1_i = sum_k C1[i,k]^2 p_k, r2_j = sum_k q_k C2[j,k]^2, |y_i|^2, |z_j|^2  -> pm / psum scratch
    if (tid < N) {
        double r1 = 0.0, r2 = 0.0, y2 = 0.0, z2 = 0.0;
        for (int k = 0; k < N; ++k) {
            const double c1 = C1l[tid * P + k], c2 = (double)C2l[tid * P + k];
            r1 += (KL ? c1 * log(c1 + 1e-15) - c1 : c1 * c1) * pq[k]; r2 += pq[64 + k] * (KL ? c2 : c2 * c2);
        }
        for (int c = 0; c < d; ++c) { const double yy = y_zero ? 0.0 : Y[(size_t)tid * d + c], zz = (double)Z[(size_t)tid * d + c]; y2 += yy * yy; z2 += zz * zz; }
        pm[tid] = r1; pm[64 + tid] = r2; psum[tid] = y2; psum[64 + tid] = z2;
    }
    // ---- dot(Y_i, Z_j) on MFMA -> Al
    if (!y_zero)
        mm_f64(N, N, d, [&](int i, int k) { return Y[(size_t)i * d + k]; }, [&](int k, int j) { return (double)Z[(size_t)j * d + k]; },
               [&](int i, int j, double v) { Al[i * P + j] = v; });
    __syncthreads();
    // ---- base = 2*alpha*constC + (1-alpha)*M in both register layouts            (utils.py:39-43,154-171, bregman.py:124-125)
    double baseA[R], baseB[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int q = w + 4 * r;                               // row index in layout A, column index in layout B
        double va = 0.0, vb = 0.0;
        if (q < N && lane_ok) {
            {   // layout A: (i, j) = (q, lane)
                double m = -2.0 * (y_zero ? 0.0 : Al[q * P + lane]); m += psum[q]; m += psum[64 + lane];
                m = m > 0.0 ? m : 0.0;
                va = 2.0 * alpha * (pm[q] + pm[64 + lane]) + (1.0 - alpha) * m;
            }
            {   // layout B: (i, j) = (lane, q)
                double m = -2.0 * (y_zero ? 0.0 : Al[lane * P + q]); m += psum[lane]; m += psum[64 + q];
                m = m > 0.0 ? m : 0.0;
                vb = 2.0 * alpha * (pm[lane] + pm[64 + q]) + (1.0 - alpha) * m;
            }
        }
        baseA[r] = va; baseB[r] = vb;
    }
    __syncthreads();

    const BorderIdx bnn = border_prepare(N, N);                          // border ownership of the N x N products below
    int cpt = 0, sk_total = 0;
    double err = 1.0;
    while (err > (double)prm.inner_tol && cpt < prm.max_iter) {          // bregman.py:119
        // ---- A = C1 @ T ; G = A @ (2 C2)^T                                (utils.py:48-64)
        mm_f64(N, N, N, [&](int i, int k) { return C1l[i * P + k]; }, [&](int k, int j) { return (double)Tl[k * P + j]; },
               [&](int i, int j, double v) { Al[i * P + j] = v; }, bnn);
        __syncthreads();
        mm_f64(N, N, N, [&](int i, int k) { return Al[i * P + k]; },
               [&](int k, int j) { const double cv = (double)C2l[j * P + k]; return KL ? log(cv + 1e-15) : cv; },
               [&](int i, int j, double v) { Gl[i * P + j] = KL ? v : 2.0 * v; }, bnn);
        __syncthreads();
        // ---- Mr = -(base - 2 alpha G)/eps in both layouts (sinkhorn.py:388)
        double mA[R], mB[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int q = w + 4 * r;
            const bool ok = q < N && lane_ok;
            // padding entries (row/column >= N) are masked BY VALUE: -1e300 plus any u or v stays -1e300, its exp is 0, so
            // the log-sum-exp loops below carry no per-element bounds tests (they cost an exec-mask branch each)
            mA[r] = ok ? -(baseA[r] - 2.0 * alpha * Gl[q * P + lane]) * inv_eps : -1.0e300;
            mB[r] = ok ? -(baseB[r] - 2.0 * alpha * Gl[lane * P + q]) * inv_eps : -1.0e300;
        }

        __syncthreads();                                               // G fully consumed: its storage now holds the Sinkhorn scratch
        // ---- log-domain Sinkhorn (sinkhorn.py:393-433); u, v in registers (u_l: i = lane, v_l: j = lane)
        double u_l = 0.0, v_l = 0.0;
        us[w * 64 + lane] = 0.0;
        int ii = 0;
        for (; ii < prm.num_iter_max; ++ii) {
            // v_j = logb_j - logsumexp_i(Mr_ij + u_i)          layout A: serial over my rows, then ONE 4-way combine of
            // (partial max, partial sum) pairs: sum = sum_w s_w * exp(m_w - M)  => one barrier per half-iteration
            double z[R];
            double mx = -1.0e300;
#pragma unroll
            for (int r = 0; r < R; ++r) { z[r] = mA[r] + us[w * 64 + w + 4 * r]; mx = fmax(z[r], mx); }
            double sm = 0.0;
#pragma unroll
            for (int r = 0; r < R; ++r) sm += exp_lse(z[r] - mx);
            pm[w * 64 + lane] = mx; psum[w * 64 + lane] = sm;
            __syncthreads();
            {
                const double m0 = pm[lane], m1 = pm[64 + lane], m2 = pm[128 + lane], m3 = pm[192 + lane];
                const double M = fmax(fmax(m0, m1), fmax(m2, m3));
                sm = ((psum[lane] * exp_lse(m0 - M) + psum[64 + lane] * exp_lse(m1 - M)) + psum[128 + lane] * exp_lse(m2 - M)) +
                     psum[192 + lane] * exp_lse(m3 - M);
                v_l = lane_ok ? logb - (log_acc(sm) + M) : 0.0;          // padding lanes keep a finite (zero) potential
            }
            vs[w * 64 + lane] = v_l;                                   // private per-wave copy: read back by this wave only
            // u_i = loga_i - logsumexp_j(Mr_ij + v_j)          layout B (pm2/psum2: second buffer => no extra barrier)
            mx = -1.0e300;
#pragma unroll
            for (int r = 0; r < R; ++r) { z[r] = mB[r] + vs[w * 64 + w + 4 * r]; mx = fmax(z[r], mx); }
            sm = 0.0;
#pragma unroll
            for (int r = 0; r < R; ++r) sm += exp_lse(z[r] - mx);
            pm2[w * 64 + lane] = mx; psum2[w * 64 + lane] = sm;
            __syncthreads();
            {
                const double m0 = pm2[lane], m1 = pm2[64 + lane], m2 = pm2[128 + lane], m3 = pm2[192 + lane];
                const double M = fmax(fmax(m0, m1), fmax(m2, m3));
                sm = ((psum2[lane] * exp_lse(m0 - M) + psum2[64 + lane] * exp_lse(m1 - M)) + psum2[128 + lane] * exp_lse(m2 - M)) +
                     psum2[192 + lane] * exp_lse(m3 - M);
                u_l = lane_ok ? loga - (log_acc(sm) + M) : 0.0;
            }
            us[w * 64 + lane] = u_l;
            if (ii % 10 == 0) {                                        // marginal violation (sinkhorn.py:418-433)
                double cs = 0.0;
#pragma unroll
                for (int r = 0; r < R; ++r) cs += exp_acc(mA[r] + us[w * 64 + w + 4 * r] + v_l);
                __syncthreads();                                       // previous psum fully consumed
                psum[w * 64 + lane] = cs;
                __syncthreads();
                cs = ((psum[lane] + psum[64 + lane]) + psum[128 + lane]) + psum[192 + lane];
                double df = lane_ok ? cs - qj : 0.0;
                df = wave_sum_d(df * df);
                if (sqrt(df) < (double)prm.stop_thr) { ++ii; break; }
            }
        }
        sk_total += ii;
        // ---- T = exp(Mr + u + v) (sinkhorn.py:450); err = ||T - Tprev||_F when cpt % 10 == 0 (bregman.py:144-147)
        // Branch-free: padding entries have Mr = -1e300 => exp = 0; their LDS traffic is redirected to a scratch word.
        double e2 = 0.0;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int i = w + 4 * r;
            const bool ok = lane_ok && i < N;
            float *tp = ok ? &Tl[i * P + lane] : t_dummy;
            const float tn = (float)exp_acc(mA[r] + us[w * 64 + i] + v_l);
            const double df = ok ? (double)tn - (double)*tp : 0.0;
            e2 += df * df;
            *tp = tn;
        }
        if (cpt % 10 == 0) err = sqrt(block_sum_d(e2, red));
        else __syncthreads();
        ++cpt;
    }
    __syncthreads();
    for (int t = tid; t < NN; t += FGW_THREADS) { const int i = t / N, j = t - i * N; Tg[t] = Tl[i * P + j]; }
    if (tid == 0) { atomicAdd(&info[b * 4 + 1], cpt); atomicAdd(&info[b * 4 + 2], sk_total); }

    // ---- contributions to the barycenter update while T is resident
    if (!prm.fixed_features) {                                          // Ypart = T @ Z                      (utils.py:90-95)
        double *Yp = Ypart + ((size_t)b * D.K + s) * N * d;
        mm_f64(N, d, N, [&](int i, int k) { return (double)Tl[i * P + k]; }, [&](int k, int c) { return (double)Z[(size_t)k * d + c]; },
               [&](int i, int c, double v) { Yp[(size_t)i * d + c] = v; });
    }
    if (!prm.fixed_structure) {                                         // Cpart = T @ C2 @ T^T               (utils.py:67-73)
        double *Cp = Cpart + ((size_t)b * D.K + s) * NN;
        mm_f64(N, N, N, [&](int i, int k) { return (double)Tl[i * P + k]; },
               [&](int k, int j) { const double cv = (double)C2l[k * P + j]; return KL ? log(cv > 1e-15 ? cv : 1e-15) : cv; },
               [&](int i, int j, double v) { Al[i * P + j] = v; }, bnn);
        __syncthreads();
        mm_f64(N, N, N, [&](int i, int k) { return Al[i * P + k]; }, [&](int k, int j) { return (double)Tl[j * P + k]; },
               [&](int i, int j, double v) { Cp[i * N + j] = v; }, bnn);
    }
}

// Barycenter update from the per-graph contributions: elementwise, one workgroup per molecule.
constexpr int UPD_THREADS = 1024;       // the kernel is a few dependent L2 round trips per molecule: more threads, fewer trips each
__global__ void __launch_bounds__(UPD_THREADS) k_fgw_update_parts(
    const float *__restrict__ pb, const float *__restrict__ lambdas, FgwDims D, conan_fgw_params prm, int outer,
    const double *__restrict__ Ypart, const double *__restrict__ Cpart, double *__restrict__ Cw, double *__restrict__ Yw,
    int *__restrict__ active, int *__restrict__ info, float *__restrict__ errs, float *__restrict__ Yout, float *__restrict__ Cout) {
    __shared__ double red[UPD_THREADS / 64 + 1];
    const int b = blockIdx.x;
    if (!active[b]) return;
    const int N = D.N, d = D.d, K = D.K, NN = N * N, Nd = N * d;
    const int tid = threadIdx.x;
    double ef2 = 0.0, es2 = 0.0;
    // The K contributions of U consecutive elements per thread are requested together (K * U loads in flight): the kernel is a
    // handful of L2 round trips per molecule, so its time is the number of dependent trips, not the byte count.
    constexpr int U = 4;
    if (!prm.fixed_features) {
        double *Yb = Yw + (size_t)b * Nd;
        for (int t0 = tid; t0 < Nd; t0 += U * UPD_THREADS) {
            double acc[U], old[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int t = t0 + u * UPD_THREADS;
                acc[u] = 0.0; old[u] = 0.0;
                if (t < Nd) {
                    old[u] = Yb[t];
                    for (int s = 0; s < K; ++s) {
                        const double lam = lambdas ? (double)lambdas[s] : 1.0 / (double)K;
                        const int i = t / d;
                        const double pinv = 1.0 / (pb ? (double)pb[(size_t)b * N + i] : 1.0 / (double)N);
                        acc[u] += lam * Ypart[((size_t)b * K + s) * Nd + t] * pinv;          // utils.py:94
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int t = t0 + u * UPD_THREADS;
                if (t < Nd) {
                    const double df = acc[u] - old[u];
                    ef2 += df * df;
                    Yb[t] = acc[u];
                    Yout[(size_t)b * Nd + t] = (float)acc[u];
                }
            }
        }
    }
    if (!prm.fixed_structure) {
        double *Cb = Cw + (size_t)b * NN;
        for (int t0 = tid; t0 < NN; t0 += U * UPD_THREADS) {
            double acc[U], old[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int t = t0 + u * UPD_THREADS;
                acc[u] = 0.0; old[u] = 0.0;
                if (t < NN) {
                    old[u] = Cb[t];
                    for (int s = 0; s < K; ++s) {
                        const double lam = lambdas ? (double)lambdas[s] : 1.0 / (double)K;
                        acc[u] += lam * Cpart[((size_t)b * K + s) * NN + t];                 // utils.py:70
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int t = t0 + u * UPD_THREADS;
                if (t < NN) {
                    const int i = t / N, j = t - i * N;
                    const double pi = pb ? (double)pb[(size_t)b * N + i] : 1.0 / (double)N;
                    const double pj = pb ? (double)pb[(size_t)b * N + j] : 1.0 / (double)N;
                    const double cn = prm.loss_fun ? exp(acc[u] / (pi * pj)) : acc[u] / (pi * pj);     // :72-73 / :86-87
                    const double df = cn - old[u];
                    es2 += df * df;
                    Cb[t] = cn;
                    Cout[(size_t)b * NN + t] = (float)cn;
                }
            }
        }
    }
    const double ef = sqrt(block_sum_d<UPD_THREADS / 64>(ef2, red));
    const double es = sqrt(block_sum_d<UPD_THREADS / 64>(es2, red));
    if (tid == 0) {
        errs[((size_t)b * 2 + 0) * prm.max_iter + outer] = (float)ef;
        errs[((size_t)b * 2 + 1) * prm.max_iter + outer] = (float)es;
        info[b * 4 + 0] = outer + 1;
        active[b] = (ef > (double)prm.tol || es > (double)prm.tol) ? 1 : 0;           // barycenter.py:112
    }
}

inline size_t small_lds(int N, int d) {
    const size_t NP = (size_t)N * (N | 1);
    (void)d; return NP * 8 * 3 + ((NP >= 768 ? 0 : 256 * 6) + 128 + 8) * 8 + NP * 4 * 2;
}

}  // namespace

bool conan_fgw_small_supported(int N, int d) { return N <= 64 && small_lds(N, d) <= 160 * 1024; }

size_t conan_fgw_small_part_bytes(int B, int K, int N, int d) {
    return ((size_t)B * K * N * d + (size_t)B * K * N * N) * 8 + 512;
}

void conan_fgw_small_coupling(const float *Ys, const float *Cs, const float *ps, const float *pb, FgwDims D,
                              conan_fgw_params prm, int outer, int y_zero, const double *Cw, const double *Yw,
                              const int *active, float *Tw, int *info, double *Ypart, double *Cpart, hipStream_t s) {
    const size_t lds = small_lds(D.N, D.d);
    const int R = (D.N + 3) / 4;
    const int grid = D.B * D.K;
#define LAUNCH(RR)                                                                                                              \
    do {                                                                                                                        \
        if (lds > 64 * 1024)                                                                                                    \
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_fgw_coupling_small<RR, KLV>),                          \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                                    \
        k_fgw_coupling_small<RR, KLV><<<grid, FGW_THREADS, lds, s>>>(Ys, Cs, ps, pb, D, prm, outer, y_zero, Cw, Yw, active, Tw, \
                                                                info, Ypart, Cpart);                                            \
    } while (0)
    if (prm.loss_fun) {
        constexpr bool KLV = true;
        if (R <= 6) LAUNCH(6);
        else if (R <= 9) LAUNCH(9);
        else if (R <= 12) LAUNCH(12);
        else LAUNCH(16);
    } else {
        constexpr bool KLV = false;
        if (R <= 6) LAUNCH(6);
        else if (R <= 9) LAUNCH(9);
        else if (R <= 12) LAUNCH(12);
        else LAUNCH(16);
    }
#undef LAUNCH
}

void conan_fgw_small_update(const float *pb, const float *lambdas, FgwDims D, conan_fgw_params prm, int outer,
                            const double *Ypart, const double *Cpart, double *Cw, double *Yw, int *active, int *info,
                            float *errs, float *Yout, float *Cout, hipStream_t s) {
    k_fgw_update_parts<<<D.B, UPD_THREADS, 0, s>>>(pb, lambdas, D, prm, outer, Ypart, Cpart, Cw, Yw, active, info, errs, Yout, Cout);
}
