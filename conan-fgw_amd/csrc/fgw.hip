// Batched Fused-Gromov-Wasserstein barycenter for gfx950.
//
// Reference algorithm: fgw_barycenters (conan_fgw/src/model/fgw/barycenter.py:7-225) -> fgw_projected
// (bregman.py:70-167) -> sinkhorn_log (sinkhorn.py:318-450) with utils.py:39-95,154-171, called once per molecule from a
// Python loop (schnet_no_sum.py:259-312).  Here every molecule of the batch is solved concurrently:
//
//   k_fgw_init                          C <- init_C (or Cs[b,0]),  Y <- init_Y (or 0)
//   repeat max_iter times (fixed launch count, data-dependent early exit through a per-molecule `active` flag,
//   no host synchronisation, hipGraph-capturable):
//     k_fgw_coupling  grid B*K          one workgroup per (molecule, input graph): projected-gradient loop with
//                                       log-domain Sinkhorn, matrices resident in LDS
//     k_fgw_update    grid B            barycenter feature/structure update + stopping test
//
// Numerics ("FGW numerics" in DESIGN.md): I/O is fp32, the iteration state (C, Y, Sinkhorn potentials, cost matrices,
// matmul accumulators) is fp64.  The reference's own fp32 run sits 1e-4..1e-3 away from its fp64 run because the
// 5-iteration scheme amplifies rounding; computing on-chip in fp64 puts this kernel on the fp64
// side of that yard-stick at negligible cost (the per-molecule matrices are tiny and MI355X runs fp64 FMA at half the
// fp32 rate).  exp() is evaluated as 2^n * v_exp_f32(frac) with the range reduction done in fp64 (1 ulp of fp32).
#include "fgw_common.h"
#ifdef CONAN_FGW_PROFILE
FGW_PROF_ACCESSOR(conan_debug_fgw_prof_large)
FGW_PROF_TRACE_ACCESSOR(conan_debug_fgw_trace_large)
#endif

namespace {
// bytes of global scratch per coupling workgroup (Mr, A, base fp64 + T fp32 = 28 B per matrix entry), rounded to 16 B so that every
// workgroup's fp64 region is aligned whatever the parity of N*P
__host__ __device__ inline size_t coupling_scratch_stride(size_t NP) { return (NP * 28 + 15) / 16 * 16; }
}
namespace {

// ------------------------------------------------------------------------------------------------ init
__global__ void k_fgw_init(const float *__restrict__ Cs, const float *__restrict__ init_C, const float *__restrict__ init_Y,
                           FgwDims D, int max_iter, double *__restrict__ Cw, double *__restrict__ Yw, int *__restrict__ active,
                           int *__restrict__ info, float *__restrict__ errs, float *__restrict__ Yout, float *__restrict__ Cout, FgwAdj adj) {
    const int b = blockIdx.x;
    const int NN = D.N * D.N, Nd = D.N * D.d;
    if (!init_C && adj.rowptr) {                                          // init_C = adjacency of the molecule's first graph, from its ragged lists
        const int N = D.N, g = b * D.K;
        for (int t = threadIdx.x; t < NN; t += blockDim.x) Cout[(size_t)b * NN + t] = 0.f;
        __syncthreads();
        const int lo = adj.gptr[g], n = min(adj.gptr[g + 1] - lo, N);
        for (int e = adj.rowptr[lo] + (int)threadIdx.x; e < adj.rowptr[lo + n]; e += blockDim.x) {
            const int i = adj.tgt[e] - lo, j = adj.col[e] - lo;
            if (i >= 0 && i < N && j >= 0 && j < N) atomicAdd(&Cout[(size_t)b * NN + j * N + i], 1.0f);
        }
        __threadfence_block();
        __syncthreads();
        for (int t = threadIdx.x; t < NN; t += blockDim.x) Cw[(size_t)b * NN + t] = (double)Cout[(size_t)b * NN + t];
    } else {
    const float *c0 = init_C ? init_C + (size_t)b * NN : Cs + (size_t)b * D.K * NN;   // init_C = Cs[0] (schnet_no_sum.py:303)
    for (int t = threadIdx.x; t < NN; t += blockDim.x) { Cw[(size_t)b * NN + t] = (double)c0[t]; Cout[(size_t)b * NN + t] = c0[t]; }
    }
    for (int t = threadIdx.x; t < Nd; t += blockDim.x) {
        float y = init_Y ? init_Y[(size_t)b * Nd + t] : 0.f;                            // barycenter.py:76-77
        Yw[(size_t)b * Nd + t] = (double)y; Yout[(size_t)b * Nd + t] = y;
    }
    if (threadIdx.x == 0) { fgw_active_init(active, D.B, b); info[b * 4 + 0] = 0; info[b * 4 + 1] = 0; info[b * 4 + 2] = 0; info[b * 4 + 3] = 0; }
    for (int t = threadIdx.x; t < 2 * max_iter; t += blockDim.x) errs[(size_t)b * 2 * max_iter + t] = __builtin_nanf("");
    if (adj.order && blockIdx.x == gridDim.x - 1) fgw_order_by_size<256>(adj, D.B, D.K, (int)threadIdx.x);      // (launched with 256 threads)
}

// ------------------------------------------------------------------------------------------------ coupling solve
// One workgroup per (molecule b, input graph s).  Matrices (pitch P): Tl fp32, Mr/Al/base fp64.
// LDS_MODE: the four matrices live in LDS; otherwise in a per-workgroup global scratch (large N).
// NW wavefronts per workgroup (8 = 512 threads): a workgroup's time is its dependent chain (tools/fgw_scaling.py), and at N ~ 80-110
// the chain is made of serial loops over N/NW rows and of N^2/256/NW product tiles per wavefront, so twice the wavefronts shorten
// it; residency is bound by LDS (the Sinkhorn cost, 2 workgroups per CU) either way.
// MODE 2: the four matrices in LDS; 1: only the coupling (Mr / K) in LDS; 0: everything in the global scratch.  A template parameter,
// not a runtime flag: a pointer chosen at run time between LDS and global memory compiles to flat_* accesses for every
// element of the Sinkhorn passes.
template <int MODE, bool KL, int NW, bool SECOND = false>      // KL: loss_fun = "kl_loss"; SECOND: the pass behind k_fgw_coupling_big (see fgw_small.hip)
__global__ void __launch_bounds__(64 * NW) k_fgw_coupling(
    const float *__restrict__ Ys, const float *__restrict__ Cs, const float *__restrict__ ps, const float *__restrict__ pb,
    FgwDims D, conan_fgw_params prm, int outer, int y_zero, const double *__restrict__ Cw, const double *__restrict__ Yw,
    const int *__restrict__ active, float *__restrict__ Tw, int *__restrict__ info, char *__restrict__ scratch,
    fgw_part_t *__restrict__ Ypart, fgw_part_t *__restrict__ Cpart, const int *__restrict__ only, FgwAdj adj) {
    constexpr bool LDS_MODE = MODE == 2, MR_LDS = MODE >= 1;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NT = 64 * NW;
    auto solve = [&](const int cid) {
    const int b = cid / D.K, s = cid % D.K;
    if (!fgw_active(active, D.B, b, outer)) return;
    const int N = D.N, P = D.P, d = D.d;
    const int NN = N * N, NP = N * P;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    FGW_PROF_DECL;

    // ---- carve
    double *vec = reinterpret_cast<double *>(smem);          // [(6 + 2 NW)*N + 16] : u, v, loga, logb, r1/y2, r2/z2, red, pm[NW][N], psm[NW][N]
    double *u = vec, *v = vec + N, *loga = vec + 2 * N, *logb = vec + 3 * N, *ra = vec + 4 * N, *rb = vec + 5 * N;
    double *red = vec + 6 * N;
    double *pm = red + 16, *psm = pm + NW * N;                // per-wavefront partial (max, sum) of the log-sum-exp loops
    // LDS_MODE: all four matrices in LDS.  Otherwise only the Sinkhorn cost Mr (read 2x per Sinkhorn iteration, once by
    // rows and once by columns) stays in LDS when it fits (mr_lds); A, base and T live in an L2-resident global scratch.
    char *gs = scratch + (size_t)cid * coupling_scratch_stride(NP);     // 16-byte aligned per coupling
    char *ls = smem + (size_t)((6 + 2 * NW) * N + 16) * 8;
    double *Mr = MR_LDS ? reinterpret_cast<double *>(ls) : reinterpret_cast<double *>(gs);
    double *Al = LDS_MODE ? Mr + NP : reinterpret_cast<double *>(gs) + NP;
    double *base = Al + NP;
    float *Tl = reinterpret_cast<float *>(base + NP);

    const float *Z = Ys + ((size_t)b * D.K + s) * N * d;        // features of input graph s      [N,d]
    // structure of input graph s [N,N]; with the ragged structure (FgwAdj) this kernel is only the exact pass behind k_fgw_coupling_big, and a
    // flagged coupling expands its graph into its own slice of the dense scratch first
    const float *C2 = adj.rowptr ? adj_dense_slice<NT>(adj, cid, N, tid) : Cs + ((size_t)b * D.K + s) * NN;
    const double *C1 = Cw + (size_t)b * NN;                     // current barycenter structure   [N,N]
    const double *Y = Yw + (size_t)b * N * d;                   // current barycenter features    [N,d]
    float *Tg = Tw + ((size_t)b * D.K + s) * NN;
    const double alpha = (double)prm.alpha, eps = (double)prm.epsilon;

    // ---- marginals: p (barycenter), q = ps[s]; uniform when not given (barycenter.py:50-51, schnet_no_sum.py:264-279)
    for (int i = tid; i < N; i += NT) {
        const double pi = pb ? (double)pb[(size_t)b * N + i] : 1.0 / (double)N;
        const double qi = ps ? (double)ps[((size_t)b * D.K + s) * N + i] : 1.0 / (double)N;
        loga[i] = log(pi); logb[i] = log(qi);
        u[i] = pi; v[i] = qi;                                  // temporarily hold p, q
    }
    __syncthreads();
    // ---- T0: warm start from the previous outer iteration, else outer(p, q)      (bregman.py:98-101)
    for (int t = tid; t < NN; t += NT) {
        const int i = t / N, j = t - i * N;
        const float t0 = (outer > 0 && prm.warmstart) ? Tg[t] : (float)(u[i] * v[j]);
        Tl[i * P + j] = t0;
        Mr[i * P + j] = (double)t0;     // the coupling also lives (fp64) where the Sinkhorn state K will: the products read it from there
    }
    // ---- init_matrix (utils.py:39-43): constC[i][j] = sum_k C1[i,k]^2 p_k + sum_k q_k C2[j,k]^2 ; squared feature norms
    double *y2a = Al, *z2a = Al + N;                            // Al is not live yet
    {   // 8 lanes per index, strided partial sums combined by xor-shuffles (fixed order): one thread per index walked N + d
        // dependent L2 round trips
        constexpr int LPI = 8;
        for (int i0 = 0; i0 < N; i0 += NT / LPI) {
            const int i = i0 + tid / LPI, sub = tid % LPI;
            double r1 = 0.0, r2 = 0.0, y2 = 0.0, z2 = 0.0;
            if (i < N) {
                for (int k = sub; k < N; k += LPI) {
                    const double c1 = C1[i * N + k], c2 = (double)C2[i * N + k];
                    r1 += (KL ? c1 * log(c1 + 1e-15) - c1 : c1 * c1) * u[k];
                    r2 += v[k] * (KL ? c2 : c2 * c2);
                }
                for (int c = sub; c < d; c += LPI) {
                    const double yy = Y[i * d + c], zz = (double)Z[i * d + c];
                    y2 += yy * yy; z2 += zz * zz;
                }
            }
#pragma unroll
            for (int o = 1; o < LPI; o <<= 1) {
                r1 += __shfl_xor(r1, o, 64); r2 += __shfl_xor(r2, o, 64); y2 += __shfl_xor(y2, o, 64); z2 += __shfl_xor(z2, o, 64);
            }
            if (i < N && sub == 0) { ra[i] = r1; rb[i] = r2; y2a[i] = y2; z2a[i] = z2; }
        }
    }
    __syncthreads();
    FGW_PROF(0);      // staging: T0, per-index vectors
    // ---- base = alpha*2*constC + (1-alpha)*M,  M = clamp(|y_i|^2 + |z_j|^2 - 2 y_i.z_j, 0)   (utils.py:154-171, bregman.py:124-125)
    // dot(Y_i, Z_j) on fp64 MFMA straight from global memory (L2-resident), then the elementwise assembly
    if (!y_zero)
        mm_f64_glb<NW, true>(N, N, d, Y, d, Z, d, [&](int i, int j, double v) { base[i * P + j] = v; });
    __syncthreads();
    FGW_PROF(1);      // dot(Y, Z)
    for (int t = tid; t < NN; t += NT) {
        const int i = t / N, j = t - i * N;
        double m = -2.0 * (y_zero ? 0.0 : base[i * P + j]);    // utils.py:159-161
        m += y2a[i]; m += z2a[j];
        m = m > 0.0 ? m : 0.0;                                  // :163
        base[i * P + j] = 2.0 * alpha * (ra[i] + rb[j]) + (1.0 - alpha) * m;
    }
    __syncthreads();
    int zero_mass = 0;
    for (int i = tid; i < N; i += NT) { ra[i] = exp(loga[i]); rb[i] = exp(logb[i]); zero_mass |= (loga[i] < -1.0e300 || logb[i] < -1.0e300) ? 1 : 0; }      // p_i, q_j for the scaling form (r1 / r2 are consumed)
    double *pa = ra, *qb = rb;
    // nodes without mass (fgw.py embeds n != N problems with such nodes): the scaling form's first half-step would count them (g = 1 on every
    // row), so such couplings take the log-domain path, whose potentials start at -inf there
    const bool massless = __syncthreads_or(zero_mass) != 0;
    FGW_PROF(2);      // base

    // ---- projected gradient loop (bregman.py:119-157)
    int cpt = 0, sk_total = 0;
    double err = 1.0;
    while (err > (double)prm.inner_tol && cpt < prm.max_iter) {
        // A = C1 @ T ; G = A @ (2 C2)^T on fp64 MFMA ; tens = base - 2*alpha*G ; Mr = -tens/eps
        // (utils.py:48-64, bregman.py:124-125, sinkhorn.py:388)
        // W operand: the coupling as the Sinkhorn state left it in Mr's storage (LDS in modes 1 / 2; the fp32-rounded T in an
        // fp64 container) instead of the fp32 copy in the global scratch: half of this product's memory accesses
        mm_f64_glb<NW, false>(N, N, N, C1, N, Mr, P, [&](int i, int j, double v) { Al[i * P + j] = v; });
        __syncthreads();
        FGW_PROF(3);  // A = C1 @ T
        auto form_mr = [&]() {
            if constexpr (KL)
                mm_f64<NW>(N, N, N, [&](int i, int k) { return Al[i * P + k]; }, [&](int k, int j) { return log((double)C2[j * N + k] + 1e-15); },
                       [&](int i, int j, double g) { Mr[i * P + j] = -(base[i * P + j] - 2.0 * alpha * g) / eps; });
            else
                mm_f64_glb<NW, true>(N, N, N, Al, P, C2, N,
                                     [&](int i, int j, double g) { Mr[i * P + j] = -(base[i * P + j] - 4.0 * alpha * g) / eps; });      // hC2 = 2 C2
        };
        form_mr();
        __syncthreads();
        FGW_PROF(4);  // G, Mr
        // ---- Sinkhorn in its matrix-scaling form, in place on Mr (see fgw_small.hip): K = exp(Mr - Mr_jj) with the first column
        // step folded in, then alternating passes   K <- K diag(f), row sums -> g = a / rowsum   (lane <-> row)   and
        // K <- diag(g) K, column sums -> f = b / colsum   (lane <-> column): one multiply-add per entry and half-iteration instead
        // of two exp.  After a row-scaled pass K is exp(Mr + u + v) of the reference iteration (sinkhorn.py:415-416), its column
        // sums are the marginal check (:418-433) and feed the next v update.  A sum outside [1e-150, 1e150] (or not finite) sends
        // the call to the exact log-domain path below.  u[] holds the row factors g, v[] the column factors f.
        int ii = 0;
        bool exact = false;
        auto bad = [](double x) { return !(x > 1e-150 && x < 1e150); };
        static_assert(NW < 15, "red[15] is the range flag: block_sum_d<NW> must not reach it");
        double *bad_flag = red + 15;                                      // set by whoever sees a sum out of range; read after the next barrier
        for (int j = tid; j < N; j += NT) v[j] = Mr[j * P + j];          // column references (the diagonal), before K overwrites them
        if (tid == 0) *bad_flag = massless ? 1.0 : 0.0;
        __syncthreads();
        for (int j = lane; j < N; j += 64) {                              // K = exp(Mr - ref_j), partial column sums
            const double ref = v[j];
            double cs = 0.0;
            for (int i = wave; i < N; i += NW) { const double k = exp_fast(Mr[i * P + j] - ref); Mr[i * P + j] = k; cs += k; }
            psm[wave * N + j] = cs;
        }
        __syncthreads();
        for (int j = tid; j < N; j += NT) {
            double cs = 0.0;
#pragma unroll
            for (int w = 0; w < NW; ++w) cs += psm[w * N + j];
            if (bad(cs)) *bad_flag = 1.0;
            v[j] = qb[j] / cs;                                            // f_j of the first v update (u = 0)
        }
        __syncthreads();
        exact = *bad_flag != 0.0;                                         // workgroup-uniform: read after the barrier that published it
        FGW_PROF(5);  // K = exp(Mr - ref), first column step
        for (; !exact && ii < prm.num_iter_max; ++ii) {
            // K <- K diag(f); row sums                                                         (sinkhorn.py:415 applied, :416 prepared)
            for (int i = lane; i < N; i += 64) {
                double rs = 0.0;
                for (int j = wave; j < N; j += NW) { const double k = Mr[i * P + j] * v[j]; Mr[i * P + j] = k; rs += k; }
                pm[wave * N + i] = rs;
            }
            __syncthreads();
            for (int i = tid; i < N; i += NT) {
                double rs = 0.0;
#pragma unroll
                for (int w = 0; w < NW; ++w) rs += pm[w * N + i];
                if (bad(rs)) *bad_flag = 1.0;
                u[i] = pa[i] / rs;
            }
            __syncthreads();
            if (*bad_flag != 0.0) { exact = true; break; }
            // K <- diag(g) K; column sums = marginals of the iterate                           (sinkhorn.py:416 applied)
            for (int j = lane; j < N; j += 64) {
                double cs = 0.0;
                for (int i = wave; i < N; i += NW) { const double k = Mr[i * P + j] * u[i]; Mr[i * P + j] = k; cs += k; }
                psm[wave * N + j] = cs;
            }
            __syncthreads();
            double e2 = 0.0;
            for (int j = tid; j < N; j += NT) {
                double cs = 0.0;
#pragma unroll
                for (int w = 0; w < NW; ++w) cs += psm[w * N + j];
                if (bad(cs)) *bad_flag = 1.0;
                const double df = cs - qb[j];
                e2 += df * df;
                v[j] = qb[j] / cs;                                        // f_j of the next v update
            }
            if (ii % 10 == 0) {                                           // marginal violation, sinkhorn.py:418-433
                const double tot = block_sum_d<NW>(e2, red);              // (its barriers publish v[] and the flag)
                if (*bad_flag != 0.0) { exact = true; break; }
                if (sqrt(tot) < (double)prm.stop_thr) { ++ii; break; }
            } else {
                __syncthreads();
                if (*bad_flag != 0.0) { exact = true; break; }
            }
        }
        if (exact) {
            // ---- exact log-domain Sinkhorn (sinkhorn.py:393-433), restarted from u = v = 0 on a re-formed Mr
            __syncthreads();
            form_mr();
            // (a node without mass has log-weight -inf and its potential is -inf after its first update; it starts there, so that it never
            // enters the other side's first log-sum-exp: the rectangular problem fgw.py embeds has no such node at all)
            for (int i = tid; i < N; i += NT) { u[i] = loga[i] < -1.0e300 ? loga[i] : 0.0; v[i] = logb[i] < -1.0e300 ? logb[i] : 0.0; }     // sinkhorn.py:393-394
            __syncthreads();
            for (ii = 0; ii < prm.num_iter_max; ++ii) {
                // v_j = logb_j - logsumexp_i(Mr_ij + u_i).  lane <-> column (consecutive lanes read consecutive LDS words), the
                // wavefronts split the rows: serial (max, sum) per thread, no cross-lane fp64 reductions; the partials per column
                // are combined through LDS.
                for (int j = lane; j < N; j += 64) {
                    double mx = -1.0e300;
                    for (int i = wave; i < N; i += NW) { const double z = Mr[i * P + j] + u[i]; mx = fmax(z, mx); }
                    double sm = 0.0;
                    for (int i = wave; i < N; i += NW) sm += exp_lse(Mr[i * P + j] + u[i] - mx);      // argument <= 0: fp32 exponent unit
                    pm[wave * N + j] = mx; psm[wave * N + j] = sm;
                }
                __syncthreads();
                for (int j = tid; j < N; j += NT) {
                    double M = pm[j];
#pragma unroll
                    for (int w = 1; w < NW; ++w) M = fmax(M, pm[w * N + j]);
                    double sm = 0.0;
#pragma unroll
                    for (int w = 0; w < NW; ++w) sm += psm[w * N + j] * exp_lse(pm[w * N + j] - M);
                    v[j] = logb[j] - (log_acc(sm) + M);
                }
                __syncthreads();
                // u_i = loga_i - logsumexp_j(Mr_ij + v_j): lane <-> row (odd pitch: conflict-free), wavefronts split the columns
                for (int i = lane; i < N; i += 64) {
                    double mx = -1.0e300;
                    for (int j = wave; j < N; j += NW) { const double z = Mr[i * P + j] + v[j]; mx = fmax(z, mx); }
                    double sm = 0.0;
                    for (int j = wave; j < N; j += NW) sm += exp_lse(Mr[i * P + j] + v[j] - mx);
                    pm[wave * N + i] = mx; psm[wave * N + i] = sm;
                }
                __syncthreads();
                for (int i = tid; i < N; i += NT) {
                    double M = pm[i];
#pragma unroll
                    for (int w = 1; w < NW; ++w) M = fmax(M, pm[w * N + i]);
                    double sm = 0.0;
#pragma unroll
                    for (int w = 0; w < NW; ++w) sm += psm[w * N + i] * exp_lse(pm[w * N + i] - M);
                    u[i] = loga[i] - (log_acc(sm) + M);
                }
                __syncthreads();
                if (ii % 10 == 0) {                                 // marginal violation, sinkhorn.py:418-433
                    for (int j = lane; j < N; j += 64) {
                        double sm = 0.0;
                        for (int i = wave; i < N; i += NW) sm += exp_acc(Mr[i * P + j] + u[i] + v[j]);
                        psm[wave * N + j] = sm;
                    }
                    __syncthreads();
                    double e2 = 0.0;
                    for (int j = tid; j < N; j += NT) {
                        double cs = 0.0;
#pragma unroll
                        for (int w = 0; w < NW; ++w) cs += psm[w * N + j];
                        const double df = cs - qb[j];
                        e2 += df * df;
                    }
                    const double tot = block_sum_d<NW>(e2, red);
                    if (sqrt(tot) < (double)prm.stop_thr) { ++ii; break; }
                }
            }
            for (int t = tid; t < NN; t += NT) {                    // K = exp(Mr + u + v) in place: both paths hand the same state on
                const int i = t / N, j = t - i * N;
                Mr[i * P + j] = exp_acc(Mr[i * P + j] + u[i] + v[j]);
            }
            __syncthreads();
        }
        sk_total += ii;
        FGW_PROF(6);  // Sinkhorn iterations
        // ---- T = exp(Mr + u + v) (sinkhorn.py:450); err = ||T - Tprev||_F evaluated when cpt % 10 == 0 (bregman.py:144-147)
        double e2 = 0.0;
        for (int t = tid; t < NN; t += NT) {
            const int i = t / N, j = t - i * N;
            const float tn = (float)Mr[i * P + j];                  // the scaled coupling = exp(Mr + u + v), sinkhorn.py:450
            const double df = (double)tn - (double)Tl[i * P + j];
            e2 += df * df;
            Tl[i * P + j] = tn;
            Mr[i * P + j] = (double)tn;                             // the products read T from here: the same fp32-rounded values that are returned
        }
        if (cpt % 10 == 0) err = sqrt(block_sum_d<NW>(e2, red));
        else __syncthreads();
        ++cpt;
        FGW_PROF(7);  // T store + err
    }
    __syncthreads();
    for (int t = tid; t < NN; t += NT) { const int i = t / N, j = t - i * N; Tg[t] = Tl[i * P + j]; }
    if (tid == 0) { atomicAdd(&info[b * 4 + 1], cpt); atomicAdd(&info[b * 4 + 2], sk_total); }
    FGW_PROF(8);      // T -> global
    // ---- contributions to the barycenter update (summed over s by k_fgw_update_parts)
    if (!prm.fixed_features) {                                          // Ypart = T @ Z                      (utils.py:90-95)
        fgw_part_t *Yp = Ypart + ((size_t)b * D.K + s) * N * d;
        mm_f64_glb<NW, false>(N, d, N, Mr, P, Z, d, [&](int i, int c, double v) { Yp[(size_t)i * d + c] = (fgw_part_t)v; });
    }
    FGW_PROF(9);      // Ypart = T @ Z
    if (!prm.fixed_structure) {                                         // Cpart = T @ C2 @ T^T               (utils.py:67-73)
        fgw_part_t *Cp = Cpart + ((size_t)b * D.K + s) * NN;
        if constexpr (KL)
            mm_f64<NW>(N, N, N, [&](int i, int k) { return (double)Tl[i * P + k]; },
                   [&](int k, int j) { const double cv = (double)C2[k * N + j]; return log(cv > 1e-15 ? cv : 1e-15); },
                   [&](int i, int j, double v) { Al[i * P + j] = v; });
        else
            mm_f64_glb<NW, false>(N, N, N, Mr, P, C2, N, [&](int i, int j, double v) { Al[i * P + j] = v; });
        __syncthreads();
        mm_f64_glb<NW, true>(N, N, N, Al, P, Mr, P, [&](int i, int j, double v) { Cp[i * N + j] = (fgw_part_t)v; });
    }
    FGW_PROF(10);     // Cpart = T @ C2 @ T^T
    FGW_PROF_FLUSH;
    };
    if constexpr (!SECOND) solve(blockIdx.x);
    else {      // one workgroup per 64 couplings: their flags are fetched by ONE load per lane (a ballot every wavefront forms for itself)
        const int total = D.B * D.K, base = (int)blockIdx.x * 64, l = (int)threadIdx.x & 63;
        unsigned long long m = __ballot(base + l < total && only[base + l < total ? base + l : 0] != 0);
        while (m) {
            const int k = __ffsll((long long)m) - 1;
            m &= m - 1;
            solve(base + k);
            __syncthreads();                                            // LDS is re-staged by the next trip
        }
    }
}

// ------------------------------------------------------------------------------------------------ round-3 coupling kernel, N > 64
// Same algorithm as k_fgw_coupling (square loss), re-cut like the N <= 64 kernel of fgw_small.hip (k_fgw_coupling_fast):
//   * Sinkhorn iterates on the scaling VECTORS against a FIXED kernel matrix K = exp(Mr - ref_j), formed once per projected-gradient
//     iteration where the product G = A (2 C2)^T leaves its result; every half-iteration is a matrix-vector product streamed from
//     LDS (one read + one FMA per entry, no write-back) instead of an in-place rescaling of the whole matrix.
//   * K is kept in LDS as fp32 — its entries carry the ~1e-7 relative error of the hardware exp2 anyway, every sum over them is
//     accumulated in fp64 — with ref_j = max_i(-base_ij / eps), so that the exponent of every entry is <= 2 G_ij (a few units: no
//     overflow in fp32; entries below e^-87 of their column's best vanish, as they do in the fp32 coupling that is returned).
//     Half the LDS (27.5 KB + vectors at N = 83) puts THREE 8-wavefront workgroups on a CU instead of two: the 640 couplings
//     of a Lipophilicity-shaped shard of 128 molecules are resident at once (768 slots) instead of running 1.25 rounds.
//   * The coupling itself lives in K's storage between iterations (fp32, the values that are returned).
// A row / column sum outside [1e-150, 1e150] — or a first-update row sum below 1e-28, i.e. a K row in or near the fp32 denormals — makes the
// workgroup give up without writing anything and raise redo[b, s]; the launcher then runs k_fgw_coupling — which carries the exact
// log-domain path — on the flagged couplings only.
// 2 x 2 register-blocked products (one operand load per MFMA instead of two: -9.5 % of the solve).  Round 5 built the exact-tile
// form — whole tiles + v_mfma_f64_4x4x4_4b_f64 strips for the ragged edge (N = 85: 25 tiles + strips instead of 36 padded tiles) — twice and
// measured it slower both times (profiles/r5_ab_fgw_large_strips_v1.txt, _v2.txt; DESIGN 3.3 has the numbers and why): removed again.
#define FGW_MMG mm_f64_glb22
#ifndef CONAN_FGW_ADJ_I8
#define CONAN_FGW_ADJ_I8 1      // G = A C2^T against the byte adjacency on the integer matrix pipe, exact digits (mm_adj_i8); 0: the fp64 product
#endif
// C2U8: the adjacency of the input graph is staged ONCE into LDS as bytes (caller's promise cs_small_int: integers in [0, 255]) and both
// products that contract with it read it there instead of fetching fp32 from L2 in every projected-gradient iteration.
// WPC: workgroups per CU the register budget is cut for — 3 (80 registers: a 640-coupling shard is resident in one round, with ~200 B / lane of
// scratch) or 2 (128 registers, 12 B of scratch): the launcher takes 2 when the batch has at most 512 couplings anyway (BACE B = 64: 320) and, since
// round 6, at ANY number of couplings when they are dealt by size (FgwAdj.order: the model's ragged layout) — the largest couplings start first and the
// smallest wait for the first free slots: Lipophilicity B = 128 (640 couplings on 512 slots) 4.54 against 4.89 ms per step for the three-per-CU build,
// B = 256 / 512 per GPU -1.2 / -1.6 % (profiles/r6_ab_fgw_placement.txt).
// ORD: the launch deals its workgroups by molecule size (FgwAdj.order).  A template parameter, not a run-time test of the pointer: launches without an
// order (dense structure matrices: the reference-shaped call) keep exactly the code and the register allocation they had.
template <int NW, bool C2U8, int WPC = 3, bool ORD = false>
__global__ void __launch_bounds__(64 * NW, (NW * WPC + 3) / 4) k_fgw_coupling_big(
    const float *__restrict__ Ys, const float *__restrict__ Cs, const float *__restrict__ ps, const float *__restrict__ pb,
    FgwDims D, conan_fgw_params prm, FastConst fc, int outer, int y_zero, const double *__restrict__ Cw, const double *__restrict__ Yw,
    const int *__restrict__ active, float *__restrict__ Tw, int *__restrict__ info, char *__restrict__ scratch,
    fgw_part_t *__restrict__ Ypart, fgw_part_t *__restrict__ Cpart, int *__restrict__ redo, FgwAdj adj) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NT = 64 * NW;
    // XCD k owns the k-th contiguous eighth of the couplings (see k_fgw_coupling_fast): the K workgroups of a molecule share C in one L2
    int cid = (gridDim.x & 7) == 0 ? xcd_contiguous_block((int)blockIdx.x, (int)gridDim.x) : (int)blockIdx.x;
    if constexpr (ORD) {
        // Molecules by descending size (FgwAdj.order, filled by k_fgw_init), ranks dealt round the XCDs, the blocks of an XCD in rank order: the largest
        // couplings are dispatched first, and when the grid exceeds the resident slots (two workgroups per CU) the ones that wait are the smallest —
        // they go to whichever CU frees a slot first, which balances the launch better than any fixed assignment tried.
        const int x = (int)blockIdx.x & 7, pblk = (int)blockIdx.x >> 3;
        cid = __builtin_amdgcn_readfirstlane(adj.order[x + 8 * (pblk / D.K)]) * D.K + pblk % D.K;
        asm volatile("" : "+s"(cid));      // pinned to a scalar register: without it everything derived from the LOADED index is kept per lane (128 VGPRs + scratch instead of 106)
    }
    const int b = cid / D.K, s = cid % D.K;
    if (!fgw_active(active, D.B, b, outer)) return;
    const int N = D.N, P = D.P, d = D.d;
    const int NN = N * N, NP = N * P;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    FGW_PROF_DECL;

    float *Kf = reinterpret_cast<float *>(smem);                          // [N,P]  T (between iterations) / K (inside one)
    double *vec = reinterpret_cast<double *>(smem + (((size_t)NP * 4 + 15) & ~(size_t)15));
    double *fv = vec, *gv = vec + N, *pa = vec + 2 * N, *qb = vec + 3 * N, *refb = vec + 4 * N;      // f_j, g_i, p_i, q_j, min_i base_ij
    double *ra = vec + 5 * N, *rb = vec + 6 * N, *y2a = vec + 7 * N, *z2a = vec + 8 * N;             // prologue: r1_i, r2_j, |y_i|^2, |z_j|^2
    double *red = vec + 9 * N;                                            // [16]; red[15] = range flag
    double *part = red + 16;                                              // [NW][N] partial sums
    unsigned char *C2b = reinterpret_cast<unsigned char *>(part + NW * N);      // C2U8: [N,P] adjacency bytes
    double *bad_flag = red + 15;
    static_assert(NW < 15, "red[15] is the range flag");
    char *gs = scratch + (size_t)cid * coupling_scratch_stride(NP);
    double *Al = reinterpret_cast<double *>(gs) + NP;                     // [N,P] fp64, L2-resident scratch (same carve as k_fgw_coupling)
    double *base = Al + NP;

    const float *Z = Ys + ((size_t)b * D.K + s) * N * d;
    const float *C2 = Cs + ((size_t)b * D.K + s) * NN;
    const double *C1 = Cw + (size_t)b * NN;
    const double *Y = Yw + (size_t)b * N * d;
    float *Tg = Tw + ((size_t)b * D.K + s) * NN;
    const bool warm = outer > 0 && prm.warmstart;

    int zero_mass = 0;
    for (int i = tid; i < N; i += NT) {
        pa[i] = pb ? (double)pb[(size_t)b * N + i] : fc.inv_n;
        qb[i] = ps ? (double)ps[((size_t)b * D.K + s) * N + i] : fc.inv_n;
        zero_mass |= (pa[i] <= 0.0 || qb[i] <= 0.0) ? 1 : 0;
    }
    if (tid == 0) *bad_flag = 0.0;
    // A node without mass (fgw.py embeds n != N problems with such nodes) must not enter the first Sinkhorn half-step, which this kernel takes
    // with g = 1 on every row: such couplings go to the exact pass, whose potentials start at -inf on those nodes.
    const bool massless = __syncthreads_or(zero_mass) != 0;
    const bool ragged = C2U8 && adj.rowptr != nullptr;
    constexpr bool ADJ_I8 = C2U8 && CONAN_FGW_ADJ_I8 != 0;
    int c2_wide = 0;                                                      // an adjacency byte above 127: the signed-byte product below does not apply
    if constexpr (C2U8) {
        for (int t = tid; t < NN; t += NT) {
            const int i = t / N, j = t - i * N;
            const unsigned char cb = ragged ? (unsigned char)0 : (unsigned char)C2[t];
            C2b[i * P + j] = cb;
            c2_wide |= cb > 127 ? 1 : 0;
        }
    }
    // (ragged: the bytes are neighbour-list counts, at most the graph's cap)
    const bool adj_i8 = ADJ_I8 && __syncthreads_or(c2_wide) == 0;
    uint4 *Adig = reinterpret_cast<uint4 *>(gs);                         // digit records of A: the first N P doubles of the coupling's scratch (the exact kernel's Mr: unused here)
    if constexpr (C2U8) {
        if (ragged) {                                                     // the graph's adjacency counts straight from its neighbour lists (FgwAdj)
            __syncthreads();
            adj_scatter_lds_bytes<NT>(adj, cid, N, P, C2b, tid);
            __syncthreads();
        }
    }
    // ---- PADDED NODES AS ONE NODE (k_fgw_coupling_fast has the argument): the m = N - n padded nodes of the input graph and of the barycenter are
    // exchangeable, the solve runs on the (n + 1)-node problem whose last node carries their mass — Lipophilicity- / BACE-shaped batches: n = 49 / 62 of
    // N = 85 / 97 on average, the products shrink by (Nx / N)^3.  Byte layout only (the layout of the model path); checked against the data per coupling.
    int nb = N, m_pad = 0;
    if constexpr (C2U8) {
        if (ragged) nb = min(adj.gptr[cid + 1] - adj.gptr[cid], N);
        else {                                                            // dense structure: one past the last node that has an edge
            int mx = 1;
            for (int t = tid; t < NN; t += NT) {
                const int i = t / N, j = t - i * N;
                if (C2b[i * P + j]) mx = max(mx, max(i, j) + 1);
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) mx = max(mx, __shfl_xor(mx, o, 64));
            int *ired = reinterpret_cast<int *>(red);
            if (lane == 0) ired[wave] = mx;
            __syncthreads();
            nb = ired[0];
#pragma unroll
            for (int w = 1; w < NW; ++w) nb = max(nb, ired[w]);
            __syncthreads();
        }
        if (N - nb >= 2 && nb >= 1 && !pb && !ps && !massless && !prm.fixed_structure && !prm.fixed_features) {      // (workgroup-uniform)
            auto near = [](double a, double b_) { return fabs(a - b_) <= 1e-9 * (fabs(b_) + 1e-30) || a == b_; };
            bool ok = true;
            const double cbb = C1[(size_t)nb * N + nb];
            for (int r = nb + 1 + wave; r < N; r += NW) {                 // wavefront <-> row, lane <-> column
                for (int c = lane; c < d; c += 64) {
                    ok = ok && near((double)Z[(size_t)r * d + c], (double)Z[(size_t)nb * d + c]);
                    if (!y_zero) ok = ok && near(Y[(size_t)r * d + c], Y[(size_t)nb * d + c]);
                }
                for (int k = lane; k < N; k += 64) ok = ok && near(C1[(size_t)r * N + k], k < nb ? C1[(size_t)nb * N + k] : cbb);
            }
            for (int k = nb + tid; k < N; k += NT) ok = ok && near(C1[(size_t)nb * N + k], cbb);
            for (int k = wave; k < nb; k += NW)                           // columns of the block
                for (int c = nb + 1 + lane; c < N; c += 64) ok = ok && near(C1[(size_t)k * N + c], C1[(size_t)k * N + nb]);
            if (__syncthreads_and(ok) != 0) m_pad = N - nb;
        }
    }
    const int Nx = m_pad ? nb + 1 : N;                                    // logical size of the problem from here on
    const double mult = m_pad ? (double)m_pad : 1.0;
    if (m_pad && tid == 0) { pa[nb] *= mult; qb[nb] *= mult; }          // the merged node carries its block's mass
    __syncthreads();
    // ---- T0: warm start from the previous outer iteration, else outer(p, q)      (bregman.py:98-101); the merged row / column holds its block's SUM
    for (int t = tid; t < Nx * Nx; t += NT) {
        const int i = t / Nx, j = t - i * Nx;
        Kf[i * P + j] = warm ? (float)((double)Tg[i * N + j] * ((m_pad && i == nb ? mult : 1.0) * (m_pad && j == nb ? mult : 1.0))) : (float)(pa[i] * qb[j]);
    }
    {   // init_matrix vectors (utils.py:39-43) and squared feature norms: 8 lanes per index (see k_fgw_coupling)
        constexpr int LPI = 8;
        for (int i0 = 0; i0 < Nx; i0 += NT / LPI) {
            const int i = i0 + tid / LPI, sub = tid % LPI;
            double r1 = 0.0, r2 = 0.0, y2 = 0.0, z2 = 0.0;
            if (i < Nx) {
                for (int k = sub; k < Nx; k += LPI) {
                    double c2;
                    if constexpr (C2U8) c2 = ragged ? (double)C2b[i * P + k] : (double)C2[i * N + k];
                    else c2 = (double)C2[i * N + k];
                    const double c1 = C1[i * N + k];
                    r1 += c1 * c1 * pa[k];
                    r2 += qb[k] * c2 * c2;
                }
                for (int c = sub; c < d; c += LPI) {
                    const double yy = Y[i * d + c], zz = (double)Z[i * d + c];
                    y2 += yy * yy; z2 += zz * zz;
                }
            }
#pragma unroll
            for (int o = 1; o < LPI; o <<= 1) {
                r1 += __shfl_xor(r1, o, 64); r2 += __shfl_xor(r2, o, 64); y2 += __shfl_xor(y2, o, 64); z2 += __shfl_xor(z2, o, 64);
            }
            if (i < Nx && sub == 0) { ra[i] = r1; rb[i] = r2; y2a[i] = y2; z2a[i] = z2; }
        }
    }
    __syncthreads();
    FGW_PROF(0);      // staging: T0, per-index vectors
    if (!y_zero)
        FGW_MMG<NW, true>(Nx, Nx, d, Y, d, Z, d, [&](int i, int j, double v) { base[i * P + j] = v; });
    __syncthreads();
    FGW_PROF(1);      // dot(Y, Z)
    // ---- base = 2 alpha constC + (1 - alpha) M (utils.py:154-171, bregman.py:124-125), lane <-> column, wavefronts split the rows;
    // the column minimum of base rides along: ref_j = -min_i base_ij / eps is the stabiliser of K's column j
    for (int j = lane; j < Nx; j += 64) {
        double mn = 1.0e300;
        for (int i = wave; i < Nx; i += NW) {
            double m = -2.0 * (y_zero ? 0.0 : base[i * P + j]);
            m += y2a[i]; m += z2a[j];
            m = m > 0.0 ? m : 0.0;
            const double bv = fc.two_alpha * (ra[i] + rb[j]) + fc.one_m_alpha * m;
            base[i * P + j] = bv;
            mn = bv < mn ? bv : mn;
        }
        part[wave * N + j] = mn;
    }
    __syncthreads();
    for (int j = tid; j < Nx; j += NT) {
        double mn = part[j];
#pragma unroll
        for (int w = 1; w < NW; ++w) { const double o = part[w * N + j]; mn = o < mn ? o : mn; }
        refb[j] = mn;
    }
    __syncthreads();
    FGW_PROF(2);      // base

    int cpt = 0, sk_total = 0;
    double err = 1.0;
    bool bail = massless;
    while (!bail && err > fc.inner_tol && cpt < prm.max_iter) {         // bregman.py:119
        // per-lane and uniform offsets of this iteration are derived from laundered copies: hoisted out of the loop they spill
        int tq = tid, N = Nx, P = D.P;                                  // (N: the logical size; D.N stays the pitch of the global matrices)
        asm volatile("" : "+v"(tq), "+s"(N), "+s"(P));
        const int lane = tq & 63, wave = tq >> 6;
        const int tid = tq;
        int nbq = m_pad ? nb : -1;                                      // index of the merged node (-1: none)
        asm volatile("" : "+s"(nbq));
        // ---- A = C1 @ T                                                        (utils.py:48-53)
        // max |A| as a bit pattern (integer form of the next product: its fixed-point unit); compared as integers so that a NaN or an infinity in A
        // ends up as the maximum instead of being dropped by a floating-point comparison (block_max_bits)
        unsigned long long amax = 0ull;
        FGW_MMG<NW, false>(N, N, N, C1, D.N, Kf, P, [&](int i, int j, double v) {
            Al[i * P + j] = v;
            if constexpr (ADJ_I8) { const unsigned long long ab = (unsigned long long)__double_as_longlong(v) & 0x7fffffffffffffffull; amax = ab > amax ? ab : amax; }
        }, tq);
        FGW_PROF(3);  // A = C1 @ T (the product as wavefront 0 sees it; its stores are still in flight)
        int aexp = 0;
        if constexpr (ADJ_I8) {
            if (adj_i8) {
                amax = block_max_bits<NW>(amax, red);                   // (its barriers also publish A)
                const int e = (int)(amax >> 52);                        // biased exponent of max |A|; 0x7ff = an infinity or a NaN somewhere in A
                // max |A| >= 2^899 (e - 1023 >= 899), non-finite included, must reach the range guard: the integer digits would turn a NaN into a
                // finite G (__double2int_rn(NaN) = 0) and saturate beyond the unit's clamp below
                if (e >= 1922) { if (tid == 0) *bad_flag = 1.0; }
                aexp = e == 0 ? -1022 : e - 1021;                       // amax < 2^(aexp - 1): the 32-bit fixed-point image stays below 2^30, no digit overflows
                aexp = aexp < -900 ? -900 : (aexp > 900 ? 900 : aexp);
                fgw_digits_from_f64<NT>(N, N, Al, P, aexp, Adig, tid);
            }
        }
        __syncthreads();
        FGW_PROF(5);  // max |A|, digits of A (integer path), barrier
        // ---- G = A @ (2 C2)^T ; K_ij = exp(Mr_ij - ref_j), Mr = -(base - 2 alpha G) / eps   (utils.py:62-64, sinkhorn.py:388)
        auto k_entry = [&](int i, int j, double v) {
            Kf[i * P + j] = (float)exp_fast(fma(v, fc.four_alpha_inv_eps, (refb[j] - base[i * P + j]) * fc.inv_eps));
        };
        if constexpr (ADJ_I8) {
            if (adj_i8) mm_adj_i8<NW>(N, N, N, Adig, C2b, P, aexp, k_entry, tq);
            else FGW_MMG<NW, true>(N, N, N, Al, P, C2b, P, k_entry, tq);
        } else if constexpr (C2U8) FGW_MMG<NW, true>(N, N, N, Al, P, C2b, P, k_entry, tq);
        else FGW_MMG<NW, true>(N, N, N, Al, P, C2, D.N, k_entry, tq);
        for (int i = tid; i < N; i += NT) gv[i] = i == nbq ? (double)m_pad : 1.0;      // u = 0 (the merged row: its m rows enter a column sum)
        __syncthreads();
        FGW_PROF(4);  // G, K
        // ---- Sinkhorn on the scaling vectors (sinkhorn.py:413-433): f_j = b_j / sum_i K_ij g_i ; g_i = a_i / sum_j K_ij f_j
        auto bad = [](double x) { return !(x > 1e-150 && x < 1e150); };
        auto col_products = [&]() {                                     // part[w][j] = sum over this wavefront's rows of K_ij g_i
            for (int j = lane; j < N; j += 64) {
                double c0 = 0.0, c1 = 0.0;
                int i = wave;
                for (; i + NW < N; i += 2 * NW) { c0 = fma((double)Kf[i * P + j], gv[i], c0); c1 = fma((double)Kf[(i + NW) * P + j], gv[i + NW], c1); }
                if (i < N) c0 = fma((double)Kf[i * P + j], gv[i], c0);
                part[wave * N + j] = c0 + c1;
            }
        };
        int ii = 0;
        bool have_f = false;                                            // fv already holds the next v update (from the marginal check)
        for (; ii < prm.num_iter_max; ++ii) {
            if (!have_f) {                                              // v update (:415)
                col_products();
                __syncthreads();
                for (int j = tid; j < N; j += NT) {
                    double c = 0.0;
#pragma unroll
                    for (int w = 0; w < NW; ++w) c += part[w * N + j];
                    if (bad(c)) *bad_flag = 1.0;
                    fv[j] = qb[j] / c;
                }
                __syncthreads();
            }
            have_f = false;
            if (*bad_flag != 0.0) { bail = true; break; }
            for (int i = lane; i < N; i += 64) {                        // u update (:416): lane <-> row (odd pitch: conflict-free)
                double r0 = 0.0, r1 = 0.0;
                int j = wave;
                for (; j + NW < N; j += 2 * NW) { r0 = fma((double)Kf[i * P + j], fv[j], r0); r1 = fma((double)Kf[i * P + j + NW], fv[j + NW], r1); }
                if (j < N) r0 = fma((double)Kf[i * P + j], fv[j], r0);
                part[wave * N + i] = r0 + r1;
            }
            __syncthreads();
            for (int i = tid; i < N; i += NT) {
                double r = 0.0;
#pragma unroll
                for (int w = 0; w < NW; ++w) r += part[w * N + i];
                // K is fp32: at the first u update (g = 1, f_j in [q_j / N, q_j]) the row sum brackets the row's largest K entry within N^2.
                // Below 1e-28 that entry is within a few digits of the fp32 denormals (a row 87..103 e-folds under every column's best):
                // the row would be returned with two or three significant bits.  Such couplings go to the exact pass like a zero sum.
                if (bad(r) || (ii == 0 && r < 1e-28)) *bad_flag = 1.0;
                gv[i] = pa[i] / r;
            }
            __syncthreads();
            if (*bad_flag != 0.0) { bail = true; break; }
            if (ii % 10 == 0) {                                         // marginal violation (:418-433): || f * (K^T g) - b ||_2
                col_products();
                __syncthreads();
                double e2 = 0.0;
                for (int j = tid; j < N; j += NT) {
                    double c = 0.0;
#pragma unroll
                    for (int w = 0; w < NW; ++w) c += part[w * N + j];
                    if (bad(c)) *bad_flag = 1.0;
                    const double df = fv[j] * c - qb[j];
                    e2 += j == nbq ? df * df / (double)m_pad : df * df;      // the block's m columns, each with 1 / m of the merged residual
                    part[j] = qb[j] / c;                                // the next v update, parked in this thread's own slot of row 0 until the test below
                }
                const double tot = block_sum_d<NW>(e2, red);            // (its barriers publish the flag)
                if (*bad_flag != 0.0) { bail = true; break; }
                if (sqrt(tot) < fc.stop_thr) { ++ii; break; }           // stop: f stays the factor the marginals were measured with
                for (int j = tid; j < N; j += NT) fv[j] = part[j];
                have_f = true;
                __syncthreads();
            }
        }
        if (bail) break;
        sk_total += ii;
        FGW_PROF(6);  // Sinkhorn iterations
        // ---- T = diag(g) K diag(f) (sinkhorn.py:450), in place; err = ||T - Tprev||_F when cpt % 10 == 0 (bregman.py:144-147).  Tprev at
        // cpt = 0 is T0, re-read / re-formed here (K has taken its place).
        double e2 = 0.0;
        const bool want_err = cpt % 10 == 0;
        for (int t = tid; t < N * N; t += NT) {
            const int i = t / N, j = t - i * N;
            const float tn = (float)((gv[i] * (double)Kf[i * P + j]) * fv[j]);
            if (want_err) {
                const double mi = i == nbq ? (double)m_pad : 1.0, mj = j == nbq ? (double)m_pad : 1.0;
                const double tp = cpt == 0 ? (warm ? (double)(float)((double)Tg[i * D.N + j] * (mi * mj)) : (double)(float)(pa[i] * qb[j])) : 0.0;
                const double df = (double)tn - tp;
                e2 += df * df / (mi * mj);                              // a merged entry stands for m (m^2) entries of 1 / m (1 / m^2) of its value
            }
            Kf[i * P + j] = tn;
        }
        if (want_err) err = sqrt(block_sum_d<NW>(e2, red));
        else __syncthreads();
        ++cpt;
        FGW_PROF(7);  // T store + err
    }
    if (bail) {
        if (tid == 0) { redo[cid] = 1; atomicOr(&info[b * 4 + 3], 1); }      // info flag bit 0: a coupling of this molecule took the second pass
        return;
    }
    __syncthreads();
    if (m_pad) {                                                        // every entry of the block gets its share of the merged entry
        const float im = (float)(1.0 / mult);
        for (int t = tid; t < NN; t += NT) {
            const int i = t / N, j = t - i * N;
            Tg[t] = Kf[min(i, nb) * P + min(j, nb)] * ((i >= nb ? im : 1.0f) * (j >= nb ? im : 1.0f));
        }
    } else {
        for (int t = tid; t < NN; t += NT) { const int i = t / N, j = t - i * N; Tg[t] = Kf[i * P + j]; }
    }
    if (tid == 0) { atomicAdd(&info[b * 4 + 1], cpt); atomicAdd(&info[b * 4 + 2], sk_total); redo[cid] = 0; if (m_pad) atomicOr(&info[b * 4 + 3], 2); }      // (flag bit 1: solved with its padded nodes merged)
    FGW_PROF(8);      // T -> global
    if (!prm.fixed_features) {                                          // Ypart = T @ Z                      (utils.py:90-95)
        fgw_part_t *Yp = Ypart + ((size_t)b * D.K + s) * N * d;
        FGW_MMG<NW, false>(Nx, d, Nx, Kf, P, Z, d, [&](int i, int c, double v) { Yp[(size_t)i * d + c] = (fgw_part_t)v; });
        if (m_pad) {                                                    // rows of the block: 1 / m of the merged row (this workgroup wrote it: fence + barrier)
            __threadfence_block();
            __syncthreads();
            const double im = 1.0 / mult;
            for (int t = tid; t < (N - nb) * d; t += NT) {
                const int r = t / d, c = t - r * d;
                const double v = (double)Yp[(size_t)nb * d + c] * im;
                if (r > 0) Yp[(size_t)(nb + r) * d + c] = (fgw_part_t)v;
            }
            __syncthreads();
            for (int c = tid; c < d; c += NT) Yp[(size_t)nb * d + c] = (fgw_part_t)((double)Yp[(size_t)nb * d + c] * im);
        }
    }
    FGW_PROF(9);      // Ypart = T @ Z
    if (!prm.fixed_structure) {                                         // Cpart = T @ C2 @ T^T               (utils.py:67-73)
        fgw_part_t *Cp = Cpart + ((size_t)b * D.K + s) * NN;
        if constexpr (C2U8) FGW_MMG<NW, false>(Nx, Nx, Nx, Kf, P, C2b, P, [&](int i, int j, double v) { Al[i * P + j] = v; });
        else FGW_MMG<NW, false>(N, N, N, Kf, P, C2, N, [&](int i, int j, double v) { Al[i * P + j] = v; });
        __syncthreads();
        if (m_pad) {                                                    // at the logical size into `base` (free since the loop ended), expanded on the way out
            FGW_MMG<NW, true>(Nx, Nx, Nx, Al, P, Kf, P, [&](int i, int j, double v) { base[i * P + j] = v; });
            __threadfence_block();
            __syncthreads();
            const double im = 1.0 / mult;
            for (int t = tid; t < NN; t += NT) {
                const int i = t / N, j = t - i * N;
                Cp[t] = (fgw_part_t)(base[min(i, nb) * P + min(j, nb)] * ((i >= nb ? im : 1.0) * (j >= nb ? im : 1.0)));
            }
        } else {
            FGW_MMG<NW, true>(N, N, N, Al, P, Kf, P, [&](int i, int j, double v) { Cp[i * N + j] = (fgw_part_t)v; });
        }
    }
    FGW_PROF(10);     // Cpart = T @ C2 @ T^T
    FGW_PROF_FLUSH;
    FGW_PROF_TRACE(Nx | (cpt << 8) | (sk_total << 20));
}

// ------------------------------------------------------------------------------------------------ backward
// dYs[b,s,j,c] = lam_s * sum_i T[b,s,i,j] * (1/p_i) * dY[b,i,c]
template <bool LDS>
__global__ void __launch_bounds__(256) k_fgw_bwd(const float *__restrict__ T, const float *__restrict__ dY, const float *__restrict__ pb,
                                                 const float *__restrict__ lambdas, int K, int N, int d, float *__restrict__ dYs) {
    extern __shared__ float bw_smem[];
    const int b = blockIdx.x / K, s = blockIdx.x % K;
    const float *Ts = T + ((size_t)b * K + s) * N * N;
    const float *g = dY + (size_t)b * N * d;
    const float lam = lambdas ? lambdas[s] : 1.0f / (float)K;
    if (LDS) {                                                // T_s and diag(1/p) dY staged once (coalesced), products from LDS
        float *Tl = bw_smem, *gl = bw_smem + N * N;
        for (int t = threadIdx.x; t < N * N; t += 256) Tl[t] = Ts[t];
        for (int t = threadIdx.x; t < N * d; t += 256) {
            const int i = t / d;
            const float pinv = pb ? (pb[(size_t)b * N + i] > 0.f ? 1.0f / pb[(size_t)b * N + i] : 0.f) : (float)N;      // (massless node: no gradient)
            gl[t] = pinv * g[t];
        }
        __syncthreads();
        for (int t = threadIdx.x; t < N * d; t += 256) {
            const int j = t / d, c = t - j * d;
            float a = 0.f;
            for (int i = 0; i < N; ++i) a += Tl[i * N + j] * gl[i * d + c];
            dYs[((size_t)b * K + s) * N * d + t] = lam * a;
        }
        return;
    }
    for (int t = threadIdx.x; t < N * d; t += 256) {
        const int j = t / d, c = t - j * d;
        float a = 0.f;
        for (int i = 0; i < N; ++i) {
            const float pinv = pb ? (pb[(size_t)b * N + i] > 0.f ? 1.0f / pb[(size_t)b * N + i] : 0.f) : (float)N;
            a += Ts[i * N + j] * pinv * g[i * d + c];
        }
        dYs[((size_t)b * K + s) * N * d + t] = lam * a;
    }
}

// ------------------------------------------------------------------------------------------------ glue: densify / normalise
// One workgroup per conformer graph g.  feat[sumN,d] -> Ys[g,N,d] = a + (x + shift - min)*(b-a)/(max-min) with padded rows
// x = 0 included in min/max (schnet_no_sum.py:59,66 / barycenter.py:393-399); Cs[g,src,tgt] += 1 per edge (to_dense_adj).
__global__ void __launch_bounds__(256) k_densify(const float *__restrict__ feat, const int *__restrict__ gptr, const int *__restrict__ rowptr,
                                                 const int *__restrict__ col, int N, int d, float shift, float a, float b,
                                                 float *__restrict__ Ys, float *__restrict__ Cs, float *__restrict__ minmax) {
    __shared__ float smn[4], smx[4];
    const int g = blockIdx.x;
    const int lo = gptr[g], n_raw = gptr[g + 1] - lo;
    // N is a caller-supplied hint (max_nodes).  A conformer with more atoms than N cannot be represented: nothing is
    // read or written outside this graph's [N,d] / [N,N] slabs, and the slab's min/max are poisoned with NaN so that the
    // whole molecule's output is NaN (visible) instead of silently wrong.
    const bool overflow = n_raw > N;
    const int n = overflow ? N : n_raw;
    const int tid = threadIdx.x;
    float mn = 3.0e38f, mx = -3.0e38f;
    for (int t = tid; t < n * d; t += 256) { const float v = feat[(size_t)lo * d + t] + shift; mn = fminf(mn, v); mx = fmaxf(mx, v); }
    if (n < N) { const float v = 0.f + shift; mn = fminf(mn, v); mx = fmaxf(mx, v); }     // zero padding rows of to_dense_batch
    mn = wave_min(mn); mx = wave_max(mx);
    if ((tid & 63) == 0) { smn[tid >> 6] = mn; smx[tid >> 6] = mx; }
    __syncthreads();
    mn = fminf(fminf(smn[0], smn[1]), fminf(smn[2], smn[3]));
    mx = fmaxf(fmaxf(smx[0], smx[1]), fmaxf(smx[2], smx[3]));
    if (overflow) mn = mx = __builtin_nanf("");
    if (tid == 0) { minmax[g * 2] = mn; minmax[g * 2 + 1] = mx; }
    const float scale = b - a, range = mx - mn;
    float *Yg = Ys + (size_t)g * N * d;
    for (int t = tid; t < N * d; t += 256) {
        const float x = (t < n * d ? feat[(size_t)lo * d + t] : 0.f) + shift;
        Yg[t] = a + __fdiv_rn((x - mn) * scale, range);                 // a + (t - min) * (b - a) / (max - min)
    }
    if (!Cs) return;                                                     // features only: the solver reads the structure from the ragged lists (conan_fgw_barycenter_fwd_ragged)
    float *Cg = Cs + (size_t)g * N * N;
    for (int t = tid; t < N * N; t += 256) Cg[t] = 0.f;
    __syncthreads();
    // edges of this graph: targets lo..lo+n-1 ; adj[src_local, tgt_local] += 1   (to_dense_adj accumulates).  One thread per EDGE
    // (a thread per target walked its ~20 edges through dependent loads); the target of an edge is found by bisection in the graph's
    // row pointers, staged in LDS.  Counts are added atomically: a radius graph has no duplicate pairs, an arbitrary edge_index may,
    // and sums of 1.0f are exact in any order.
    __shared__ int rp[257];
    const int e0 = rowptr[lo], e1 = rowptr[lo + n];
    for (int i0 = 0; i0 < n; i0 += 256) {                                // graphs above 256 atoms: row pointers in chunks
        const int m = min(256, n - i0);
        __syncthreads();
        for (int t = tid; t <= m; t += 256) rp[t] = rowptr[lo + i0 + t];
        __syncthreads();
        for (int e = rp[0] + tid; e < rp[m]; e += 256) {
            int a = 0, b = m;                                            // rp[a] <= e < rp[b]
            while (b - a > 1) { const int c = (a + b) >> 1; if (rp[c] <= e) a = c; else b = c; }
            const int i = i0 + a, j = col[e] - lo;
            if (j >= 0 && j < N) atomicAdd(&Cg[j * N + i], 1.0f);
        }
    }
    (void)e0; (void)e1;
}

// to_dense_adj of every graph into the dense scratch (shapes / losses whose kernels have no ragged load stage): one workgroup per graph
__global__ void __launch_bounds__(256) k_adj_dense(FgwAdj adj, int N) { (void)adj_dense_slice<256>(adj, (int)blockIdx.x, N, (int)threadIdx.x); }

// Backward of the feature half: y = a + (x + shift - mn) * s / r, r = mx - mn, through min() and max() like autograd
// (the gradient of a full-tensor min/max is split evenly among ties; padded entries absorb their share).
__global__ void __launch_bounds__(256) k_densify_bwd(const float *__restrict__ feat, const float *__restrict__ dYs, const int *__restrict__ gptr,
                                                     const float *__restrict__ minmax, int N, int d, float shift, float a, float b,
                                                     float *__restrict__ dfeat) {
    __shared__ float red[3][4];
    const int g = blockIdx.x;
    const int lo = gptr[g], n_raw = gptr[g + 1] - lo;
    const int n = n_raw > N ? N : n_raw;          // overflowing conformer (see k_densify): minmax is NaN, so is every gradient
    const int tid = threadIdx.x;
    const float mn = minmax[g * 2], mx = minmax[g * 2 + 1];
    const float s = b - a, r = mx - mn;
    const float *G = dYs + (size_t)g * N * d;
    // S0 = sum g ; S1 = sum g*(x - mn) over the whole padded slab ; count ties of min and max
    float s0 = 0.f, s1 = 0.f, cmin = 0.f, cmax = 0.f;
    for (int t = tid; t < N * d; t += 256) {
        const float x = (t < n * d ? feat[(size_t)lo * d + t] : 0.f) + shift;
        const float gg = G[t];
        s0 += gg; s1 += gg * (x - mn);
        cmin += (x == mn) ? 1.f : 0.f; cmax += (x == mx) ? 1.f : 0.f;
    }
    float vals[4] = {s0, s1, cmin, cmax};
    __shared__ float tot[4];
    for (int q = 0; q < 4; ++q) {
        float v = wave_sum(vals[q]);
        if ((tid & 63) == 0) red[0][tid >> 6] = v;
        __syncthreads();
        if (tid == 0) tot[q] = red[0][0] + red[0][1] + red[0][2] + red[0][3];
        __syncthreads();
    }
    // dL/dmn = sum g * ( -s/r + (x-mn)*s/r^2 ) ; dL/dmx = sum g * ( -(x-mn)*s/r^2 )
    const float dmn = -tot[0] * s / r + tot[1] * s / (r * r);
    const float dmx = -tot[1] * s / (r * r);
    for (int t = tid; t < n * d; t += 256) {
        const float x = feat[(size_t)lo * d + t] + shift;
        float gx = G[t] * s / r;
        if (x == mn) gx += dmn / tot[2];
        if (x == mx) gx += dmx / tot[3];
        dfeat[(size_t)lo * d + t] = gx;
    }
    for (int t = n * d + tid; t < n_raw * d; t += 256) dfeat[(size_t)lo * d + t] = __builtin_nanf("");
}

// F_bary readout (schnet_no_sum.py:308-312 ; visnet.py:233-248)
__global__ void __launch_bounds__(64) k_readout_fwd(const float *__restrict__ Y, int K, int N, int d, int mode, float *__restrict__ out) {
    const int b = blockIdx.x;
    const float *Yb = Y + (size_t)b * N * d;
    __shared__ int has_nan;
    if (threadIdx.x == 0) has_nan = 0;
    __syncthreads();
    if (mode == 1) {
        int bad = 0;
        for (int t = threadIdx.x; t < N * d; t += 64) bad |= isnan(Yb[t]) ? 1 : 0;
        if (bad) has_nan = 1;
        __syncthreads();
    }
    for (int c = threadIdx.x; c < d; c += 64) {
        float sum = 0.f, sq = 0.f;
        for (int i0 = 0; i0 < N; i0 += 8) {                     // 8 loads in flight (clamped), summed in node order
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = Yb[min(i0 + u, N - 1) * d + c];
#pragma unroll
            for (int u = 0; u < 8; ++u) { const float y = (has_nan || i0 + u >= N) ? 0.f : v[u]; sum += y; sq += y * y; }
        }
        float r = mode == 1 ? sum / sqrtf(sq) : sum;       // sum_i Y_ic / ||Y_:c||_2   (division by zero -> NaN, as the reference)
        for (int k = 0; k < K; ++k) out[((size_t)b * K + k) * d + c] = r;
    }
}
__global__ void __launch_bounds__(64) k_readout_bwd(const float *__restrict__ Y, const float *__restrict__ dout, int K, int N, int d, int mode,
                                                    float *__restrict__ dY) {
    const int b = blockIdx.x;
    const float *Yb = Y + (size_t)b * N * d;
    for (int c = threadIdx.x; c < d; c += 64) {
        float g = 0.f;
        for (int k = 0; k < K; ++k) g += dout[((size_t)b * K + k) * d + c];
        if (mode == 0) {
            for (int i = 0; i < N; ++i) dY[((size_t)b * N + i) * d + c] = g;
        } else {
            float sum = 0.f, sq = 0.f;
            for (int i = 0; i < N; ++i) { const float y = Yb[i * d + c]; sum += y; sq += y * y; }
            const float nrm = sqrtf(sq);
            // d/dy_i [ sum / nrm ] = 1/nrm - sum * y_i / nrm^3
            for (int i = 0; i < N; ++i) dY[((size_t)b * N + i) * d + c] = g * (1.0f / nrm - sum * Yb[i * d + c] / (nrm * nrm * nrm));
        }
    }
}

inline int pitch_of(int N) { return fgw_pitch(N); }
inline size_t al256(size_t v) { return (v + 255) / 256 * 256; }
#ifndef CONAN_FGW_WPC2_MAX
#define CONAN_FGW_WPC2_MAX 512      // (A/B switch) couplings up to which the 128-register build is taken
#endif
#ifndef CONAN_FGW_WPC2_ORDERED
#define CONAN_FGW_WPC2_ORDERED 1    // (A/B switch) the 128-register build at any number of couplings when they are dealt by size
#endif
#ifndef CONAN_FGW_BIG_ORDER
#define CONAN_FGW_BIG_ORDER 1       // (A/B switch) size-ordered dealing on the N > 64 path
#endif
#ifndef CONAN_FGW_BIG_WPC2
#define CONAN_FGW_BIG_WPC2 1
#endif
constexpr bool BIG_WPC2 = CONAN_FGW_BIG_WPC2 != 0;      // (A/B switch: 0 = always the three-per-CU build)
constexpr int GEN_NW = 8;                   // wavefronts per workgroup of the large-N coupling kernel (16 measured no faster: 453 vs 443 us per workgroup and launch)
inline size_t coupling_lds(int N) { return (size_t)((6 + 2 * GEN_NW) * N + 16) * 8 + (size_t)N * pitch_of(N) * 28; }
inline size_t big_lds(int N, bool c2_bytes) {
    return (((size_t)N * pitch_of(N) * 4 + 15) & ~(size_t)15) + (size_t)((9 + GEN_NW) * N + 16) * 8 + (c2_bytes ? (size_t)N * pitch_of(N) : 0);
}
constexpr size_t LDS_LIMIT = 160 * 1024;

}  // namespace

extern "C" {

long long conan_fgw_workspace_bytes(int B, int K, int N, int d) {
    if (B <= 0 || K <= 0 || N <= 0 || d <= 0) return 0;
    const size_t NN = (size_t)N * N, NP = (size_t)N * pitch_of(N);
    size_t bytes = 0;
    bytes += al256((size_t)B * NN * 8);               // Cw          (every region starts 256-byte aligned)
    bytes += al256((size_t)B * N * d * 8);            // Yw
    bytes += al256((size_t)B * 16);                   // active: [parity][features | structure][B] (fgw_active)
    bytes += al256((size_t)B * K * coupling_scratch_stride(NP));    // coupling scratch (global mode)
    bytes += al256((size_t)B * NP * 16);              // update scratch (global mode)
    bytes += conan_fgw_small_part_bytes(B, K, N, d);   // per-graph update contributions (register-resident path)
    return (long long)bytes;
}

static int fgw_fwd_impl(const float *Ys, const float *Cs, const float *ps, const float *p, const float *lambdas,
                        const float *init_C, const float *init_Y, int B, int K, int N, int d,
                        const conan_fgw_params *params_in, float *Y, float *C, float *T, float *T_iter, int *info,
                        float *errs, void *workspace, void *stream, FgwAdj adj) {
    if (!Ys || (!Cs && !adj.rowptr) || !params_in || !Y || !C || !T || !info || !errs || !workspace || B <= 0 || K <= 0 || N <= 0 || d <= 0)
        return CONAN_E_BADARG;
    conan_fgw_params params_v = *params_in;
    if (adj.rowptr) params_v.cs_small_int = 1;                          // adjacency counts are small integers by construction
    const conan_fgw_params *params = &params_v;
    if (params->max_iter <= 0 || params->num_iter_max <= 0) return CONAN_E_BADARG;
    if (params->fixed_features && !init_Y) return CONAN_E_BADARG;      // barycenter.py:70-72
    hipStream_t s = as_stream(stream);
    FgwDims D{B, K, N, d, pitch_of(N)};
    const size_t NN = (size_t)N * N, NP = (size_t)N * D.P;
    char *w = static_cast<char *>(workspace);
    double *Cw = reinterpret_cast<double *>(w); w += al256((size_t)B * NN * 8);
    double *Yw = reinterpret_cast<double *>(w); w += al256((size_t)B * N * d * 8);
    int *active = reinterpret_cast<int *>(w); w += al256((size_t)B * 16);
    char *sc_c = w; w += al256((size_t)B * K * coupling_scratch_stride(NP));
    int *order_ws = reinterpret_cast<int *>(w);      // [B] ints at the head of the reserved region: FgwAdj.order
    w += al256((size_t)B * NP * 16);        // (reserved)
    fgw_part_t *Ypart = reinterpret_cast<fgw_part_t *>(w);
    fgw_part_t *Cpart = Ypart + (size_t)B * K * N * d;
    double *zvec = reinterpret_cast<double *>(w + conan_fgw_part_offset(B, K, N, d));      // [B,K,2N]  |z_j|^2, r2_j      (register-resident path)
    double *yvec = zvec + (size_t)B * K * 2 * N;            // [B,2N]    |y_i|^2, r1_i
    int *redo = reinterpret_cast<int *>(yvec + (size_t)B * 2 * N);      // [B,K]  couplings the round-3 kernel hands back to the exact path
    const bool small = conan_fgw_small_supported(N, d);
    const bool kl = params->loss_fun != 0;
    if (params->loss_fun != 0 && params->loss_fun != 1) return CONAN_E_BADARG;
    if (adj.rowptr) {
        // ragged structure: the dense scratch sits behind the regular workspace (conan_fgw_workspace_bytes_ragged).  The kernels with a ragged
        // load stage are the square-loss ones of the model path (k_fgw_coupling_fast for N <= 64, k_fgw_coupling_big above); any other shape /
        // loss expands the graphs into the scratch once and continues on the dense path.
        adj.dense = reinterpret_cast<float *>(static_cast<char *>(workspace) + conan_fgw_workspace_bytes(B, K, N, d));
        const bool ragged_ok = !kl && (small ? conan_fgw_fast_supported(N, d, 1) : big_lds(N, true) <= LDS_LIMIT);
        if (!ragged_ok) {
            k_adj_dense<<<B * K, 256, 0, s>>>(adj, N);
            Cs = adj.dense;
            adj.rowptr = nullptr;
        }
    }

    // size-ordered dealing of the coupling workgroups (speed only): k_fgw_coupling_fast and k_fgw_coupling_big
    if (adj.rowptr && !kl && (B & 7) == 0 && B <= 4096 && (small || (CONAN_FGW_BIG_ORDER && big_lds(N, true) <= LDS_LIMIT))) adj.order = order_ws;
    if (small) conan_fgw_small_prepare(Ys, Cs, ps, p, D, *params, Cw, Yw, zvec, yvec, init_C, init_Y, active, info, errs, Y, C, adj, s);
    else k_fgw_init<<<B, 256, 0, s>>>(Cs, init_C, init_Y, D, params->max_iter, Cw, Yw, active, info, errs, Y, C, adj);
    const size_t lc = coupling_lds(N);
    const bool c_lds = lc <= LDS_LIMIT;
    const size_t vec_c = (size_t)((6 + 2 * GEN_NW) * N + 16) * 8;
    const size_t mr_bytes = NP * 8;
    const int mode = c_lds ? 2 : (vec_c + mr_bytes <= LDS_LIMIT ? 1 : 0);
    const size_t lds_bytes = mode == 2 ? lc : vec_c + (mode == 1 ? mr_bytes : 0);
#define CONAN_CPL_(M, KLV, SEC, GRID)                                                                                               \
    do {                                                                                                                            \
        if (lds_bytes > 64 * 1024)                                                                                                  \
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_fgw_coupling<M, KLV, GEN_NW, SEC>),                        \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);                                  \
        k_fgw_coupling<M, KLV, GEN_NW, SEC><<<GRID, 64 * GEN_NW, lds_bytes, s>>>(Ys, Cs, ps, p, D, *params, outer, y_zero, Cw, Yw, active, T, \
                                                                              info, sc_c, Ypart, Cpart, only, adj);                 \
    } while (0)
#define CONAN_CPL(M, KLV)                                                                                                           \
    do {                                                                                                                            \
        if (!KLV && only) CONAN_CPL_(M, false, true, (B * K + 63) / 64);                                                            \
        else CONAN_CPL_(M, KLV, false, B * K);                                                                                      \
    } while (0)
    // N > 64, square loss: the round-3 kernel first (fp32 kernel matrix in LDS: three workgroups per CU), then k_fgw_coupling over the
    // same grid for whatever it handed back (redo[b, s]; an early-exit launch otherwise)
    const bool c2b = params->cs_small_int != 0;
    const size_t lb = big_lds(N, c2b);
    const bool big = !small && !kl && lb <= LDS_LIMIT;
    const FastConst fc = fast_const(*params, N);
    // N <= 64: the update kernel forms the feature contributions T_s Z_s itself (fgw_small.hip); the coupling kernels then skip that product
    const bool y_from_t = small && !params->fixed_features && conan_fgw_update_chunk(K, N, d, B) > 0;
    fgw_part_t *Ypart_c = y_from_t ? nullptr : Ypart;
    for (int outer = 0; outer < params->max_iter; ++outer) {
        const int y_zero = (outer == 0 && !init_Y) ? 1 : 0;
        const int *only = nullptr;
        if (big) {
#define CONAN_BIG(U8, WPC, ORD)                                                                                                     \
    do {                                                                                                                            \
        if (lb > 64 * 1024)                                                                                                         \
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_fgw_coupling_big<GEN_NW, U8, WPC, ORD>),                    \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lb);                                         \
        k_fgw_coupling_big<GEN_NW, U8, WPC, ORD><<<B * K, 64 * GEN_NW, lb, s>>>(Ys, Cs, ps, p, D, *params, fc, outer, y_zero, Cw, Yw, active, T, info, sc_c, \
                                                                                 Ypart, Cpart, redo, adj);                           \
    } while (0)
            const bool two_per_cu = BIG_WPC2 && (B * K <= CONAN_FGW_WPC2_MAX || (CONAN_FGW_WPC2_ORDERED && adj.order));      // the 128-register build (see WPC above)
            const bool ordered = adj.order != nullptr && (B * K & 7) == 0 && two_per_cu && c2b;      // (an order implies the byte layout and the two-per-CU build)
            if (ordered) CONAN_BIG(true, 2, true);
            else if (c2b) { if (two_per_cu) CONAN_BIG(true, 2, false); else CONAN_BIG(true, 3, false); }
            else { if (two_per_cu) CONAN_BIG(false, 2, false); else CONAN_BIG(false, 3, false); }
#undef CONAN_BIG
            only = redo;
        }
        if (small)
            conan_fgw_small_coupling(Ys, Cs, ps, p, D, *params, outer, y_zero, Cw, Yw, active, T, info, Ypart_c, Cpart, zvec, yvec, redo, adj, s);
        else if (mode == 2) { if (kl) CONAN_CPL(2, true); else CONAN_CPL(2, false); }
        else if (mode == 1) { if (kl) CONAN_CPL(1, true); else CONAN_CPL(1, false); }
        else { if (kl) CONAN_CPL(0, true); else CONAN_CPL(0, false); }
        if (T_iter)      // log["Ts_iter"] (barycenter.py:196): a snapshot per outer iteration, only when the caller asks for the log
            (void)hipMemcpyAsync(T_iter + (size_t)outer * B * K * NN, T, (size_t)B * K * NN * sizeof(float), hipMemcpyDeviceToDevice, s);
        conan_fgw_small_update(p, lambdas, D, *params, outer, Ypart, Cpart, Cw, Yw, active, info, errs, Y, C, small ? yvec : nullptr, y_from_t ? T : nullptr, y_from_t ? Ys : nullptr, s);
    }
#undef CONAN_CPL
#undef CONAN_CPL_
    CONAN_LAUNCH_CHECK();
    return CONAN_OK;
}

int conan_fgw_barycenter_fwd(const float *Ys, const float *Cs, const float *ps, const float *p, const float *lambdas,
                             const float *init_C, const float *init_Y, int B, int K, int N, int d,
                             const conan_fgw_params *params, float *Y, float *C, float *T, float *T_iter, int *info,
                             float *errs, void *workspace, void *stream) {
    if (!Cs) return CONAN_E_BADARG;
    return fgw_fwd_impl(Ys, Cs, ps, p, lambdas, init_C, init_Y, B, K, N, d, params, Y, C, T, T_iter, info, errs, workspace, stream,
                        FgwAdj{nullptr, nullptr, nullptr, nullptr, nullptr});
}

long long conan_fgw_workspace_bytes_ragged(int B, int K, int N, int d) {
    const long long w = conan_fgw_workspace_bytes(B, K, N, d);
    return w <= 0 ? 0 : w + (long long)al256((size_t)B * K * N * N * sizeof(float));
}

int conan_fgw_barycenter_fwd_ragged(const float *Ys, const int *graph_ptr, const int *rowptr, const int *col, const int *tgt, const float *ps,
                                    const float *p, const float *lambdas, const float *init_C, const float *init_Y, int B, int K, int N, int d,
                                    const conan_fgw_params *params, float *Y, float *C, float *T, float *T_iter, int *info, float *errs,
                                    void *workspace, void *stream) {
    if (!graph_ptr || !rowptr || !col || !tgt) return CONAN_E_BADARG;
    return fgw_fwd_impl(Ys, nullptr, ps, p, lambdas, init_C, init_Y, B, K, N, d, params, Y, C, T, T_iter, info, errs, workspace, stream,
                        FgwAdj{graph_ptr, rowptr, col, tgt, nullptr});
}

int conan_fgw_barycenter_bwd(const float *T, const float *dY, const float *p, const float *lambdas, int B, int K,
                             int N, int d, float *dYs, void *stream) {
    if (!T || !dY || !dYs || B <= 0 || K <= 0 || N <= 0 || d <= 0) return CONAN_E_BADARG;
    const size_t bw_lds = ((size_t)N * N + (size_t)N * d) * sizeof(float);
    if (bw_lds <= 64 * 1024) k_fgw_bwd<true><<<B * K, 256, bw_lds, as_stream(stream)>>>(T, dY, p, lambdas, K, N, d, dYs);
    else k_fgw_bwd<false><<<B * K, 256, 0, as_stream(stream)>>>(T, dY, p, lambdas, K, N, d, dYs);
    CONAN_LAUNCH_CHECK();
    return CONAN_OK;
}

int conan_fgw_densify(const float *feat, const int *graph_ptr, const int *rowptr, const int *col, int num_graphs,
                      int N, int d, float shift, float a, float b, float *Ys, float *Cs, float *minmax, void *stream) {
    if (!feat || !graph_ptr || !Ys || !minmax || num_graphs <= 0 || N <= 0 || d <= 0 || (Cs && (!rowptr || !col))) return CONAN_E_BADARG;
    k_densify<<<num_graphs, 256, 0, as_stream(stream)>>>(feat, graph_ptr, rowptr, col, N, d, shift, a, b, Ys, Cs, minmax);
    CONAN_LAUNCH_CHECK();
    return CONAN_OK;
}

int conan_fgw_densify_bwd(const float *feat, const float *dYs, const int *graph_ptr, const float *minmax,
                          int num_graphs, int N, int d, float shift, float a, float b, float *dfeat, void *stream) {
    if (!feat || !dYs || !graph_ptr || !minmax || !dfeat || num_graphs <= 0 || N <= 0 || d <= 0) return CONAN_E_BADARG;
    k_densify_bwd<<<num_graphs, 256, 0, as_stream(stream)>>>(feat, dYs, graph_ptr, minmax, N, d, shift, a, b, dfeat);
    CONAN_LAUNCH_CHECK();
    return CONAN_OK;
}

int conan_fgw_readout_fwd(const float *Y, int B, int K, int N, int d, int mode, float *out, void *stream) {
    if (!Y || !out || B <= 0 || K <= 0 || N <= 0 || d <= 0 || mode < 0 || mode > 1) return CONAN_E_BADARG;
    k_readout_fwd<<<B, 64, 0, as_stream(stream)>>>(Y, K, N, d, mode, out);
    CONAN_LAUNCH_CHECK();
    return CONAN_OK;
}

int conan_fgw_readout_bwd(const float *Y, const float *dout, int B, int K, int N, int d, int mode, float *dY,
                          void *stream) {
    if (!Y || !dout || !dY || B <= 0 || K <= 0 || N <= 0 || d <= 0 || mode < 0 || mode > 1) return CONAN_E_BADARG;
    k_readout_bwd<<<B, 64, 0, as_stream(stream)>>>(Y, dout, K, N, d, mode, dY);
    CONAN_LAUNCH_CHECK();
    return CONAN_OK;
}

}  // extern "C"
