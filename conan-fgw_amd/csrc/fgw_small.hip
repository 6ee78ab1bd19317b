// FGW coupling solve for N <= 64 barycenter nodes (every ESOL/FreeSolv-shaped batch): register-resident Sinkhorn.
//
// Same algorithm and citations as fgw.hip (bregman.py:70-167, sinkhorn.py:318-450, utils.py:39-95).  What changes is the
// mapping to the CU.  A 256-thread workgroup (4 wavefronts) owns one (molecule, input graph) problem:
//
//   lane  <-> column j (layout A)  /  lane <-> row i (layout B),      wavefront w <-> index residue (w + 4r), r < R
//
// Every thread keeps its R entries of the N x N matrix in registers in BOTH layouts, so that a column reduction (over rows)
// and a row reduction (over columns) are each a serial loop over registers followed by a 4-way combine through LDS: no
// cross-lane shuffles of fp64 values, one barrier per half-iteration.
//
// Sinkhorn runs in its MATRIX-SCALING form on the coupling itself: with T = exp(Mr + u 1^T + 1 v^T) the log-domain update
// v <- logb - logsumexp_i(Mr + u) (sinkhorn.py:415) is T <- T diag(b / colsum(T)) and the u update (:416) is
// T <- diag(a / rowsum(T)) T — the same iteration, algebraically, with one division per row/column instead of N exp + 1 log,
// and the marginal check (:418-433) and the returned coupling (:450) are the state itself.  The first column step
// (u = v = 0) is formed in the log domain against a per-column reference (the diagonal entry), K = exp(Mr - Mr_jj), which
// keeps K inside the fp64 range for any cost spread below ~600 eps; should a row or column sum ever leave [1e-150, 1e150]
// (never seen on conformer features) the workgroup restarts that Sinkhorn call on the exact log-sum-exp path below.
// The two N^3 products of the gradient use the same mapping with T, C1 (fp64), C2 resident in LDS, one operand
// broadcast per FMA.  At the end the workgroup also forms its contribution to the barycenter update
// (T_s @ Ys_s and T_s @ Cs_s @ T_s^T) while T is still in LDS; the update kernel is then a K-term elementwise sum.
#include "fgw_common.h"

#ifdef CONAN_FGW_PROFILE
FGW_PROF_ACCESSOR(conan_debug_fgw_prof)
FGW_PROF_TRACE_ACCESSOR(conan_debug_fgw_trace)
#endif

namespace {

constexpr int SK_SCRATCH_DOUBLES = 3 * 256 + 2 * 64;      // Sinkhorn scratch: three [4][64] partial buffers + two [64] factor vectors

// KL = true: loss_fun = "kl_loss" (utils.py:20-32,76-87): f1(a) = a log(a + 1e-15) - a, f2(b) = b, h2(b) = log(b + 1e-15) in the
// gradient, log(clamp(C_s, 1e-15)) in the structure update (exp applied by the update kernel).  The logarithms are evaluated in
// fp64 where the operand is fetched (the kl path is a capability of the signature, not a tuned path); KL = false compiles to
// exactly the square-loss kernel.
template <int R, bool KL, bool SECOND = false>      // SECOND: the pass behind k_fgw_coupling_fast (walks the couplings, solves the flagged ones)
__global__ void __launch_bounds__(FGW_THREADS, (SECOND || R > 9) ? 1 : 3) k_fgw_coupling_small(      // (SECOND: full register budget — inside its loop the body's invariants are hoisted; a kernel with scratch costs every launch)
    const float *__restrict__ Ys, const float *__restrict__ Cs, const float *__restrict__ ps, const float *__restrict__ pb,
    FgwDims D, conan_fgw_params prm, int outer, int y_zero, const double *__restrict__ Cw, const double *__restrict__ Yw,
    const int *__restrict__ active, float *__restrict__ Tw, int *__restrict__ info, fgw_part_t *__restrict__ Ypart,
    fgw_part_t *__restrict__ Cpart, const double *__restrict__ zvec, const double *__restrict__ yvec, const int *__restrict__ only, FgwAdj adj) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // SECOND (the pass behind k_fgw_coupling_fast): a SMALL grid walks all couplings and solves the ones that were handed back (only[]) —
    // an empty pass then costs a few dozen workgroups instead of B * K.  Otherwise: one workgroup per coupling, no loop (inside a loop
    // the optimiser hoists the body's invariants and the kernel spills).
    auto solve = [&](const int cid) {
    const int b = cid / D.K, s = cid % D.K;
    if (!fgw_active(active, D.B, b, outer)) return;
    const int N = D.N, P = D.P, d = D.d;
    const int NN = N * N, NP = N * P;
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool lane_ok = lane < N;
    FGW_PROF_DECL;

    // ---- LDS carve (doubles first).  The constant part of the Sinkhorn cost ("base") lives in LDS, not in registers: holding it
    // in both layouts costs 4R VGPRs, which pushed the kernel over the 168-VGPR budget of three resident workgroups per CU, and
    // a spilled register here is a scratch (memory) round trip inside every phase of the loop.
    double *Al = reinterpret_cast<double *>(smem);            // [N,P]  A = C1 @ T ; dot(Y,Z) ; T @ C2
    double *Gl = Al + NP;                                      // [N,P]  G = A @ (2 C2)^T ; later the Sinkhorn scratch
    double *C1l = Gl + NP;                                     // [N,P]
    double *Bl = C1l + NP;                                     // [N,P]  base = 2 alpha constC + (1 - alpha) M
    double *pq = Bl + NP;                                      // [2][64] p, q
    double *red = pq + 128;                                    // [8]: block reductions use 5; the last word is a write sink
    float *t_dummy = reinterpret_cast<float *>(red + 7);
    double *vec4 = red + 8;                                    // [4][64]: r1_i, r2_j, |y_i|^2, |z_j|^2 (k_fgw_small_vectors / update kernel)
    // Sinkhorn scratch, 7 KB: three [4][64] partial buffers and two [64] factor vectors.  It overlays A — dead once G = A (2 C2)^T
    // is complete, which the barrier behind that product already guarantees — when A is large enough, else it has its own
    // region behind the vectors.
    const bool alias = (size_t)NP >= SK_SCRATCH_DOUBLES;
    double *own = vec4 + 256;
    double *bufC = alias ? Al : own;                           // [4][64] column partials
    double *bufK = bufC + 256;                                 // [4][64] column partials of the marginal check
    double *bufR = bufK + 256;                                 // [4][64] row partials
    double *fcp = bufR + 256;                                  // [64] column factors f_j (every wave writes the same values)
    double *gcp = fcp + 64;                                    // [64] row factors g_i
    float *Tl = reinterpret_cast<float *>(own + (alias ? 0 : SK_SCRATCH_DOUBLES));     // [N,P]
    float *C2l = Tl + NP;                                      // [N,P]
    // Y and Z are staged in LDS for the prologue's dot(Y, Z) when they fit (d <= 2P): Y (fp64 [N,d]) over A|G, Z (fp32 [N,d])
    // over base, the product lands in C1's storage, and C1 itself waits in registers until base has consumed it.  Every
    // global load of the prologue is issued before the first LDS store: one memory round trip instead of one per loop trip.
    // Epilogue: Z again over G for T @ Z.  Otherwise the products read Y / Z from global memory (L2-resident).
    double *Yl = Al;
    float *Zl = reinterpret_cast<float *>(Bl);
    float *Zl2 = reinterpret_cast<float *>(Gl);
    const bool yz_lds = d <= 2 * P;
    const bool z_lds = (size_t)N * d * sizeof(float) <= (size_t)NP * sizeof(double);

    const float *Z = Ys + ((size_t)b * D.K + s) * N * d;
    // (ragged structure: this kernel is then only the exact pass behind k_fgw_coupling_fast, and a flagged coupling expands its graph into its
    // own slice of the dense scratch first)
    const float *C2 = adj.rowptr ? adj_dense_slice<FGW_THREADS>(adj, cid, N, tid) : Cs + ((size_t)b * D.K + s) * NN;
    const double *C1 = Cw + (size_t)b * NN;
    const double *Y = Yw + (size_t)b * N * d;
    float *Tg = Tw + ((size_t)b * D.K + s) * NN;
    const double alpha = (double)prm.alpha, inv_eps = 1.0 / (double)prm.epsilon;

    // ---- loads first ...
    constexpr int EPT = (16 * R * R + FGW_THREADS - 1) / FGW_THREADS;      // matrix entries per thread (N <= 4R)
    double c1v[EPT];
    float c2v[EPT], tv[EPT];
    const bool warm = outer > 0 && prm.warmstart;
#pragma unroll
    for (int u = 0; u < EPT; ++u) {
        const int t = tid + u * FGW_THREADS, tc = t < NN ? t : NN - 1;
        c1v[u] = C1[tc]; c2v[u] = C2[tc]; tv[u] = warm ? Tg[tc] : 0.f;
    }
    const double p_own = tid < N ? (pb ? (double)pb[(size_t)b * N + tid] : 1.0 / (double)N) : 1.0;
    const double q_own = tid < N ? (ps ? (double)ps[((size_t)b * D.K + s) * N + tid] : 1.0 / (double)N) : 1.0;
    double vec_own;
    {   // r1_i = sum_k f1(C1[i,k]) p_k, |y_i|^2 (per molecule, refreshed by the update kernel); r2_j = sum_k q_k f2(C2[j,k]), |z_j|^2 (static)
        const int v = tid >> 6, i = tid & 63;
        const double *src = (v == 0 || v == 2) ? yvec + (size_t)b * 2 * N + (v == 0 ? N : 0) : zvec + ((size_t)b * D.K + s) * 2 * N + (v == 1 ? N : 0);
        vec_own = src[i < N ? i : N - 1];
        vec_own = i < N ? vec_own : 0.0;
    }
    const bool stage_yz = yz_lds && !y_zero;
    if (stage_yz) {
        const int Nd = N * d;
        for (int t0 = tid; t0 < Nd; t0 += 4 * FGW_THREADS) {
            double yv[4]; float zv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { const int t = t0 + u * FGW_THREADS, tc = t < Nd ? t : Nd - 1; yv[u] = Y[tc]; zv[u] = Z[tc]; }
#pragma unroll
            for (int u = 0; u < 4; ++u) { const int t = t0 + u * FGW_THREADS; if (t < Nd) { Yl[t] = yv[u]; Zl[t] = zv[u]; } }
        }
    }
    // ---- ... then the LDS stores
    if (tid < 64) { pq[tid] = p_own; pq[64 + tid] = q_own; }
    // a node without mass (fgw.py embeds n != N problems with such nodes) must not enter the scaling form's first half-step (g = 1 on every
    // row): such couplings take the log-domain path below, whose potentials start at -inf on those nodes
    const bool massless = __syncthreads_or(tid < N && (p_own <= 0.0 || q_own <= 0.0)) != 0;
    vec4[tid] = vec_own;
#pragma unroll
    for (int u = 0; u < EPT; ++u) {
        const int t = tid + u * FGW_THREADS;
        if (t < NN) { const int i = t / N, j = t - i * N; C2l[i * P + j] = c2v[u]; if (!stage_yz) C1l[i * P + j] = c1v[u]; }
    }
    __syncthreads();
    FGW_PROF(0);      // staging
    const double qj = pq[64 + lane];                                    // b_j with j = lane (layout A)
    const double pi_l = pq[lane];                                       // a_i with i = lane (layout B)
#pragma unroll
    for (int u = 0; u < EPT; ++u) {
        const int t = tid + u * FGW_THREADS;
        if (t < NN) { const int i = t / N, j = t - i * N; Tl[i * P + j] = warm ? tv[u] : (float)(pq[i] * pq[64 + j]); }     // bregman.py:98-101
    }
    const double *r1v = vec4, *r2v = vec4 + 64, *y2v = vec4 + 128, *z2v = vec4 + 192;
    // ---- dot(Y_i, Z_j) on MFMA -> Dl (C1's storage when Y, Z were staged; A otherwise)
    double *Dl = stage_yz ? C1l : Al;
    if (!y_zero) {
        if (stage_yz)
            mm_lds<FGW_WAVES, true>(N, N, d, Yl, d, Zl, d, [&](int i, int j, double v) { Dl[i * P + j] = v; }, border_prepare(N, N));
        else
            mm_f64(N, N, d, [&](int i, int k) { return Y[(size_t)i * d + k]; }, [&](int k, int j) { return (double)Z[(size_t)j * d + k]; },
                   [&](int i, int j, double v) { Dl[i * P + j] = v; });
    }
    __syncthreads();
    FGW_PROF(1);      // T0 + dot(Y, Z)
    // ---- base = 2*alpha*constC + (1-alpha)*M  -> Bl                              (utils.py:39-43,154-171, bregman.py:124-125)
#pragma unroll
    for (int u = 0; u < EPT; ++u) {
        const int t = tid + u * FGW_THREADS;
        if (t < NN) {
            const int i = t / N, j = t - i * N;
            double m = -2.0 * (y_zero ? 0.0 : Dl[i * P + j]); m += y2v[i]; m += z2v[j];
            m = m > 0.0 ? m : 0.0;
            Bl[i * P + j] = 2.0 * alpha * (r1v[i] + r2v[j]) + (1.0 - alpha) * m;      // (Z, staged here for the product, is dead since the barrier)
        }
    }
    __syncthreads();
    if (stage_yz) {                                                     // the product is consumed: C1 comes home from the registers
#pragma unroll
        for (int u = 0; u < EPT; ++u) {
            const int t = tid + u * FGW_THREADS;
            if (t < NN) { const int i = t / N, j = t - i * N; C1l[i * P + j] = c1v[u]; }
        }
        __syncthreads();
    }
    FGW_PROF(2);      // base

    const BorderIdx bnn = border_prepare(N, N);                          // border ownership of the N x N products below
    int cpt = 0, sk_total = 0;
    double err = 1.0;
    while (err > (double)prm.inner_tol && cpt < prm.max_iter) {          // bregman.py:119
        // ---- A = C1 @ T ; G = A @ (2 C2)^T                                (utils.py:48-64)
        mm_lds<FGW_WAVES, false>(N, N, N, C1l, P, Tl, P, [&](int i, int j, double v) { Al[i * P + j] = v; }, bnn);
        __syncthreads();
        FGW_PROF(3);  // A = C1 @ T
        auto product_G = [&]() {
            if constexpr (KL)
                mm_f64(N, N, N, [&](int i, int k) { return Al[i * P + k]; }, [&](int k, int j) { return log((double)C2l[j * P + k] + 1e-15); },
                       [&](int i, int j, double v) { Gl[i * P + j] = v; }, bnn);
            else
                mm_lds<FGW_WAVES, true>(N, N, N, Al, P, C2l, P, [&](int i, int j, double v) { Gl[i * P + j] = 2.0 * v; }, bnn);
        };
        product_G();
        __syncthreads();
        FGW_PROF(4);  // G = A @ (2 C2)^T
        // ---- Mr = -(base - 2 alpha G)/eps (sinkhorn.py:388).  Padding entries (row/column >= N) are masked BY VALUE: Mr = -1e300, K = 0.
        // The per-entry LDS offsets are derived from values the optimiser cannot see through, once per iteration: left alone it
        // hoists all 4R of them (x 3 matrices) out of the loop and keeps them in registers across the products, which spills.
        int lq = lane, wq = w;
        asm volatile("" : "+v"(lq), "+v"(wq));
        // Reads are clamped into the matrix and the value selected afterwards: no exec-mask branch per entry.
        const int lc = lq < N ? lq : N - 1;
        auto mrA = [&](int r) { const int q = wq + 4 * r, qc = q < N ? q : N - 1; const double m = -(Bl[qc * P + lc] - 2.0 * alpha * Gl[qc * P + lc]) * inv_eps; return (q < N && lane_ok) ? m : -1.0e300; };
        auto mrB = [&](int r) { const int q = wq + 4 * r, qc = q < N ? q : N - 1; const double m = -(Bl[lc * P + qc] - 2.0 * alpha * Gl[lc * P + qc]) * inv_eps; return (q < N && lane_ok) ? m : -1.0e300; };
        double kA[R], kB[R];                                            // the coupling in both layouts
        int ii = 0;
        bool exact = massless;                                          // workgroup-uniform
        {
            // ---- first column step (u = v = 0): K = exp(Mr - ref_j), f_j = b_j / sum_i K.  The stabiliser is the column's DIAGONAL
            // entry Mr_jj, which every wave reads for itself (no combine, no barrier); any reference within ~ +-600 of the column
            // maximum keeps K inside the fp64 range, and the range check on the sums covers the rest.
            const double ref_own = -(Bl[lc * P + lc] - 2.0 * alpha * Gl[lc * P + lc]) * inv_eps;      // Mr_jj, j = lane
            fcp[lane] = ref_own;                                        // identical in every wave; a wave reads back what it wrote
            double cs = 0.0;
#pragma unroll
            for (int r = 0; r < R; ++r) { kA[r] = exp_fast(mrA(r) - ref_own); cs += kA[r]; }
            bufK[w * 64 + lane] = cs;
#pragma unroll
            for (int r = 0; r < R; ++r) { const int q = wq + 4 * r; kB[r] = exp_fast(mrB(r) - fcp[q < N ? q : N - 1]); }
            __syncthreads();
            cs = ((bufK[lane] + bufK[64 + lane]) + bufK[128 + lane]) + bufK[192 + lane];
            if (__any(lane_ok && !(cs > 1e-150 && cs < 1e150))) exact = true;
            const double f = lane_ok ? qj * rcp_pos(cs) : 0.0;
            gcp[lane] = f;                                              // (gcp: free until the first row step)
#pragma unroll
            for (int r = 0; r < R; ++r) { kA[r] *= f; kB[r] *= gcp[w + 4 * r]; }
        }
        FGW_PROF(5);  // K = exp(Mr - ref), first column step
        double colsum = 0.0;                                            // column sums of the current state, when known
        bool have_colsum = false;
        for (; !exact && ii < prm.num_iter_max; ++ii) {
            if (ii > 0) {
                // ---- v update: T <- T diag(b / colsum(T))                                  (sinkhorn.py:415)
                if (!have_colsum) {
                    double cs = 0.0;
#pragma unroll
                    for (int r = 0; r < R; ++r) cs += kA[r];
                    bufC[w * 64 + lane] = cs;
                    __syncthreads();
                    colsum = ((bufC[lane] + bufC[64 + lane]) + bufC[128 + lane]) + bufC[192 + lane];
                }
                have_colsum = false;
                if (__any(lane_ok && !(colsum > 1e-150 && colsum < 1e150))) { exact = true; break; }
                const double f = lane_ok ? qj * rcp_pos(colsum) : 0.0;
                fcp[lane] = f;
#pragma unroll
                for (int r = 0; r < R; ++r) { kA[r] *= f; kB[r] *= fcp[w + 4 * r]; }
            }
            // ---- u update: T <- diag(a / rowsum(T)) T                                       (sinkhorn.py:416)
            double rs = 0.0;
#pragma unroll
            for (int r = 0; r < R; ++r) rs += kB[r];
            bufR[w * 64 + lane] = rs;
            __syncthreads();
            rs = ((bufR[lane] + bufR[64 + lane]) + bufR[128 + lane]) + bufR[192 + lane];
            if (__any(lane_ok && !(rs > 1e-150 && rs < 1e150))) { exact = true; break; }
            const double g = lane_ok ? pi_l * rcp_pos(rs) : 0.0;
            gcp[lane] = g;
#pragma unroll
            for (int r = 0; r < R; ++r) { kB[r] *= g; kA[r] *= gcp[w + 4 * r]; }
            if (ii % 10 == 0) {                                        // marginal violation (sinkhorn.py:418-433): ||colsum(T) - b||_2
                double cs = 0.0;
#pragma unroll
                for (int r = 0; r < R; ++r) cs += kA[r];
                bufK[w * 64 + lane] = cs;
                __syncthreads();
                colsum = ((bufK[lane] + bufK[64 + lane]) + bufK[128 + lane]) + bufK[192 + lane];
                have_colsum = true;                                    // the next v update starts from these sums
                double df = lane_ok ? colsum - qj : 0.0;
                df = wave_sum_d(df * df);                              // identical in every wave: the break is workgroup-uniform
                if (sqrt(df) < (double)prm.stop_thr) { ++ii; break; }
            }
        }
        if (exact) {
            // ---- exact log-domain Sinkhorn (sinkhorn.py:393-433), restarted from u = v = 0.  Only reached when the scaling form
            // would lose entries to underflow / overflow (never observed on conformer features).  Mr is re-formed from base and G at
            // every use, so this path costs the scaling path no registers; its partial buffers share the scaling path's scratch.
            __syncthreads();
            double *xm = bufC, *xs = bufK, *uc = fcp, *vc = gcp;        // [4][64] max, [4][64] sum, [64] u, [64] v
            const double loga = log(pi_l), logb = log(qj);
            double u_l = 0.0, v_l = 0.0;
            uc[lane] = (lane_ok && !(pi_l > 0.0)) ? loga : 0.0;        // (massless row: -inf from the start, see `massless`)
            for (ii = 0; ii < prm.num_iter_max; ++ii) {
                // v_j = logb_j - logsumexp_i(Mr_ij + u_i): serial over my rows, then ONE 4-way combine of (partial max, partial
                // sum) pairs: sum = sum_w s_w * exp(m_w - M)
                double z[R];
                double mx = -1.0e300;
#pragma unroll
                for (int r = 0; r < R; ++r) { z[r] = mrA(r) + uc[w + 4 * r]; mx = fmax(z[r], mx); }
                double sm = 0.0;
#pragma unroll
                for (int r = 0; r < R; ++r) sm += exp_lse(z[r] - mx);
                __syncthreads();                                       // previous readers of xm / xs are done
                xm[w * 64 + lane] = mx; xs[w * 64 + lane] = sm;
                __syncthreads();
                {
                    const double m0 = xm[lane], m1 = xm[64 + lane], m2 = xm[128 + lane], m3 = xm[192 + lane];
                    const double M = fmax(fmax(m0, m1), fmax(m2, m3));
                    sm = ((xs[lane] * exp_lse(m0 - M) + xs[64 + lane] * exp_lse(m1 - M)) + xs[128 + lane] * exp_lse(m2 - M)) +
                         xs[192 + lane] * exp_lse(m3 - M);
                    v_l = lane_ok ? logb - (log_acc(sm) + M) : 0.0;      // padding lanes keep a finite (zero) potential
                }
                vc[lane] = v_l;
                // u_i = loga_i - logsumexp_j(Mr_ij + v_j)
                mx = -1.0e300;
#pragma unroll
                for (int r = 0; r < R; ++r) { z[r] = mrB(r) + vc[w + 4 * r]; mx = fmax(z[r], mx); }
                sm = 0.0;
#pragma unroll
                for (int r = 0; r < R; ++r) sm += exp_lse(z[r] - mx);
                __syncthreads();
                xm[w * 64 + lane] = mx; xs[w * 64 + lane] = sm;
                __syncthreads();
                {
                    const double m0 = xm[lane], m1 = xm[64 + lane], m2 = xm[128 + lane], m3 = xm[192 + lane];
                    const double M = fmax(fmax(m0, m1), fmax(m2, m3));
                    sm = ((xs[lane] * exp_lse(m0 - M) + xs[64 + lane] * exp_lse(m1 - M)) + xs[128 + lane] * exp_lse(m2 - M)) +
                         xs[192 + lane] * exp_lse(m3 - M);
                    u_l = lane_ok ? loga - (log_acc(sm) + M) : 0.0;
                }
                uc[lane] = u_l;
                if (ii % 10 == 0) {                                    // marginal violation (sinkhorn.py:418-433)
                    double cs = 0.0;
#pragma unroll
                    for (int r = 0; r < R; ++r) cs += exp_acc(mrA(r) + uc[w + 4 * r] + v_l);
                    __syncthreads();
                    xs[w * 64 + lane] = cs;
                    __syncthreads();
                    cs = ((xs[lane] + xs[64 + lane]) + xs[128 + lane]) + xs[192 + lane];
                    double df = lane_ok ? cs - qj : 0.0;
                    df = wave_sum_d(df * df);
                    if (sqrt(df) < (double)prm.stop_thr) { ++ii; break; }
                }
            }
#pragma unroll
            for (int r = 0; r < R; ++r) kA[r] = exp_acc(mrA(r) + uc[w + 4 * r] + v_l);      // sinkhorn.py:450
        }
        sk_total += ii;
        FGW_PROF(6);  // Sinkhorn iterations
        // ---- T = the scaled coupling (= exp(Mr + u + v), sinkhorn.py:450); err = ||T - Tprev||_F when cpt % 10 == 0 (bregman.py:144-147)
        // Branch-free: padding entries are 0; their LDS traffic is redirected to a scratch word.
        double e2 = 0.0;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int i = wq + 4 * r;
            const bool ok = lane_ok && i < N;
            float *tp = ok ? &Tl[i * P + lq] : t_dummy;
            const float tn = (float)kA[r];
            const double df = ok ? (double)tn - (double)*tp : 0.0;
            e2 += df * df;
            *tp = tn;
        }
        if (cpt % 10 == 0) err = sqrt(block_sum_d(e2, red));
        else __syncthreads();
        ++cpt;
        FGW_PROF(7);  // T store + err
    }
    __syncthreads();
    for (int t = tid; t < NN; t += FGW_THREADS) { const int i = t / N, j = t - i * N; Tg[t] = Tl[i * P + j]; }
    if (tid == 0) { atomicAdd(&info[b * 4 + 1], cpt); atomicAdd(&info[b * 4 + 2], sk_total); }
    FGW_PROF(8);      // T -> global

    // ---- contributions to the barycenter update while T is resident
    if (!prm.fixed_features && Ypart) {                                 // Ypart = T @ Z (utils.py:90-95); nullptr: the update kernel forms it from T itself
        fgw_part_t *Yp = Ypart + ((size_t)b * D.K + s) * N * d;
        if (z_lds) {                                                    // G's storage is idle again: Z through LDS, one coalesced pass
            const int Nd = N * d;
            for (int t0 = tid; t0 < Nd; t0 += 4 * FGW_THREADS) {
                float zv[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) { const int t = t0 + u * FGW_THREADS; zv[u] = Z[t < Nd ? t : Nd - 1]; }
#pragma unroll
                for (int u = 0; u < 4; ++u) { const int t = t0 + u * FGW_THREADS; if (t < Nd) Zl2[t] = zv[u]; }
            }
            __syncthreads();
            mm_lds<FGW_WAVES, false>(N, d, N, Tl, P, Zl2, d, [&](int i, int c, double v) { Yp[(size_t)i * d + c] = (fgw_part_t)v; }, border_prepare(N, d));
        } else {
            mm_f64(N, d, N, [&](int i, int k) { return (double)Tl[i * P + k]; }, [&](int k, int c) { return (double)Z[(size_t)k * d + c]; },
                   [&](int i, int c, double v) { Yp[(size_t)i * d + c] = (fgw_part_t)v; });
        }
    }
    FGW_PROF(9);      // Ypart = T @ Z
    if (!prm.fixed_structure) {                                         // Cpart = T @ C2 @ T^T               (utils.py:67-73)
        fgw_part_t *Cp = Cpart + ((size_t)b * D.K + s) * NN;
        if constexpr (KL)
            mm_f64(N, N, N, [&](int i, int k) { return (double)Tl[i * P + k]; },
                   [&](int k, int j) { const double cv = (double)C2l[k * P + j]; return log(cv > 1e-15 ? cv : 1e-15); },
                   [&](int i, int j, double v) { Al[i * P + j] = v; }, bnn);
        else
            mm_lds<FGW_WAVES, false>(N, N, N, Tl, P, C2l, P, [&](int i, int j, double v) { Al[i * P + j] = v; }, bnn);
        __syncthreads();
        mm_lds<FGW_WAVES, true>(N, N, N, Al, P, Tl, P, [&](int i, int j, double v) { Cp[i * N + j] = (fgw_part_t)v; }, bnn);
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

// ================================================================================================================================
// Round-3 coupling kernel (square loss): same algorithm and mapping as k_fgw_coupling_small above, re-cut for RESIDENCY and fewer
// vector instructions.  At cfg2 (N = 33) the kernel above holds 46.7 KB of LDS and 151 VGPRs: three workgroups per CU, 1 280
// couplings on 768 slots = 1.67 rounds of a ~100-phase dependent chain with the vector ALU issuing in 39 % of the cycles.  Here:
//   * Sinkhorn iterates on the SCALING VECTORS with the kernel matrix fixed: K = exp(Mr - ref_j) is formed ONCE per projected-
//     gradient iteration, in the accumulator layout of the product that produces G (one exp per entry instead of one per entry and
//     register layout), and every half-iteration is a matrix-vector product against it, f = b / (K^T g), g = a / (K f)
//     (sinkhorn.py:415-416 with f = e^v, g = e^u): 9 FMAs per thread instead of 9 adds + 18 multiplies; the marginal check
//     (:418-433) is f * (K^T g), whose K^T g is the next iteration's column product.  The returned coupling (:450) is g_i K_ij f_j.
//   * LDS: A and K share one matrix (the product that turns A into K holds its results in registers until every wavefront has
//     read its operands, mm_lds_hold), the adjacency C2 is kept as bytes when its entries are small integers (the model's
//     to_dense_adj output always is; conan_fgw_params.cs_small_int), the per-index vectors and the Sinkhorn partial sums overlay
//     matrices that are dead while they live: 31.0 KB at N = 33 => FIVE workgroups per CU (160 KB / 5) and <= 96 VGPRs,
//     i.e. all 1 280 couplings of cfg2 resident at once.
//   * Integer work: (row, column) of the staging loops advance incrementally (one division per thread instead of one per entry),
//     the column reference of K is base's diagonal (read where the entry is formed; no combine, no barrier).
// What it does NOT contain is the exact log-domain fallback: a row / column sum outside [1e-150, 1e150] (never observed on conformer
// features) makes the workgroup give up WITHOUT writing anything and raise redo[b, s]; the launcher then runs the kernel above —
// which carries the exact path — on the flagged couplings only (an early-exit launch otherwise).
// ================================================================================================================================
template <int R> struct FastCfg {
    static constexpr int OCC = R <= 9 ? 5 : (R <= 12 ? 2 : 1);       // workgroups per CU the register budget is cut for
};
// tiles per wavefront of an N x N product (the dispatch rule of mm_lds): a template parameter of the kernel, because the products that
// hold their results in registers unroll over it — N = 32..34 (one 16 x 16 tile per wavefront + a thin border) must not pay the
// registers of N = 36 (nine padded tiles: three per wavefront)
inline int fast_tiles_per_wave(int N) {
    const int Mc = (N >> 4) << 4;
    const bool border = (N - Mc) * N + Mc * (N - Mc) <= FGW_WAVES * 32 && Mc > 0;
    const int q = border ? N >> 4 : (N + 15) >> 4;
    return (q * q + FGW_WAVES - 1) / FGW_WAVES;
}
constexpr int FAST_VEC_DOUBLES = 128 + 256;                          // p, q [2][64] + r1, r2, |y|^2, |z|^2 [4][64]
constexpr int FAST_SK_DOUBLES = 2 * 256 + 2 * 64;                    // column / row partial sums [4][64] each + f, g [64] each

struct FastLds {
    int npa;                 // doubles per fp64 matrix slot (N * P rounded up to even: 16-byte aligned slots)
    size_t off_t, off_c2, off_red, off_vec, off_sk, off_rs, bytes;
    bool vec_alias, sk_alias;
};
template <typename C2T>
__host__ __device__ inline FastLds fast_lds(int N) {
    FastLds L;
    const int NP = N * fgw_pitch(N);
    L.npa = (NP + 1) & ~1;
    size_t o = (size_t)3 * L.npa * 8;                                // C1 | AK | base
    L.off_t = o; o += ((size_t)NP * 4 + 15) & ~(size_t)15;           // T (fp32)
    L.off_c2 = o; o += ((size_t)NP * sizeof(C2T) + 15) & ~(size_t)15;
    L.off_red = o; o += 64;
    L.vec_alias = (size_t)NP * 4 >= FAST_VEC_DOUBLES * 8;            // vectors overlay T until T0 is written
    L.sk_alias = L.npa >= FAST_SK_DOUBLES;                           // Sinkhorn scratch overlays AK once K sits in registers
    L.off_vec = L.vec_alias ? L.off_t : o; if (!L.vec_alias) o += FAST_VEC_DOUBLES * 8;
    L.off_sk = L.sk_alias ? (size_t)L.npa * 8 : o; if (!L.sk_alias) o += FAST_SK_DOUBLES * 8;
    // [4][N] doubles: per-wavefront row sums of the complete-graph form (the input graph's structure matrix is not read on that path: they overlay it)
    const bool rs_alias = (size_t)NP * sizeof(C2T) >= (size_t)4 * N * 8;
    L.off_rs = rs_alias ? L.off_c2 : o; if (!rs_alias) o += (size_t)4 * N * 8;
    L.bytes = o;
    return L;
}

template <int R, int MAXT, typename C2T>
__global__ void __launch_bounds__(FGW_THREADS, FastCfg<R>::OCC) k_fgw_coupling_fast(
    const float *__restrict__ Ys, const float *__restrict__ Cs, const float *__restrict__ ps, const float *__restrict__ pb,
    FgwDims D, conan_fgw_params prm, FastConst fc, int outer, int y_zero, const double *__restrict__ Cw, const double *__restrict__ Yw,
    const int *__restrict__ active, float *__restrict__ Tw, int *__restrict__ info, fgw_part_t *__restrict__ Ypart,
    fgw_part_t *__restrict__ Cpart, const double *__restrict__ zvec, const double *__restrict__ yvec, int *__restrict__ redo, FgwAdj adj) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // The K workgroups of a molecule read the same C (28 KB), Y and vectors: with the plain numbering they sit on K different XCDs (workgroups are
    // dealt round-robin) and each pulls its own copy through its own L2.  XCD k takes the k-th contiguous eighth of the couplings instead.
    int cid = (gridDim.x & 7) == 0 ? xcd_contiguous_block((int)blockIdx.x, (int)gridDim.x) : (int)blockIdx.x;
    if (adj.order && (gridDim.x & 7) == 0) {
        // Molecules by descending size (k_fgw_small_vectors), dealt like cards: rank r goes to XCD r % 8 (blocks are dealt round-robin over the XCDs,
        // so block b and b + 8 share one), and the blocks of an XCD walk its molecules rank by rank — the K couplings of a molecule stay on one XCD and
        // next to each other, and the five couplings a CU receives (every 32nd block of its XCD) come from five different parts of the size range.  A
        // coupling's work grows with its real nodes (padded nodes are solved as one node): in the order the batch was drawn the busiest CU of an
        // ESOL-shaped batch carries 1.3 x the mean, dealt 1.1 x.  Placement is for speed only; every molecule's result is the same bits anywhere.
        const int x = (int)blockIdx.x & 7, pblk = (int)blockIdx.x >> 3;
        cid = adj.order[x + 8 * (pblk / D.K)] * D.K + pblk % D.K;
    }
    const int b = cid / D.K, s = cid % D.K;
    if (!fgw_active(active, D.B, b, outer)) return;
    const int N = D.N, P = D.P, d = D.d;
    const int NN = N * N;
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    FGW_PROF_DECL;

    const FastLds L = fast_lds<C2T>(N);
    double *C1l = reinterpret_cast<double *>(smem);                  // [N,P]  barycenter structure C
    double *AKl = C1l + L.npa;                                       // [N,P]  A = C1 @ T, then K = exp(Mr - ref); Sinkhorn scratch; T @ C2
    double *Bl = AKl + L.npa;                                        // [N,P]  base = 2 alpha constC + (1 - alpha) M
    float *Tl = reinterpret_cast<float *>(smem + L.off_t);           // [N,P]
    C2T *C2l = reinterpret_cast<C2T *>(smem + L.off_c2);             // [N,P]
    double *red = reinterpret_cast<double *>(smem + L.off_red);      // [8]
    float *t_dummy = reinterpret_cast<float *>(red + 7);
    double *pq = reinterpret_cast<double *>(smem + L.off_vec);       // [2][64] p, q   (prologue only)
    double *vec4 = pq + 128;                                         // [4][64] r1_i, r2_j, |y_i|^2, |z_j|^2   (prologue only)
    double *bufC = reinterpret_cast<double *>(smem + L.off_sk);      // [4][64] column partial sums
    double *bufR = bufC + 256;                                       // [4][64] row partial sums
    double *fvP = bufR + 256;                                        // [4][16] column factors f, index (j & 3) * 16 + (j >> 2)
    double *gvP = fvP + 64;                                          // [4][16] row factors g, same permutation
    double *Yl = AKl;                                                // prologue: Y (fp64 [N,d]) over AK | base
    float *Zl = reinterpret_cast<float *>(C1l);                      // prologue / epilogue: Z (fp32 [N,d]) over C1
    const bool yz_lds = d <= 2 * P;

    const float *Z = Ys + ((size_t)b * D.K + s) * N * d;
    const float *C2 = Cs + ((size_t)b * D.K + s) * NN;
    const double *C1 = Cw + (size_t)b * NN;
    const double *Y = Yw + (size_t)b * N * d;
    float *Tg = Tw + ((size_t)b * D.K + s) * NN;

    // ---- loads first: every global load of the prologue is in flight before the first LDS store.  (row, column) of a thread's
    // matrix entries: ONE division, then steps of 256 entries (the staging loops below walk them incrementally).
    constexpr int EPT = (16 * R * R + FGW_THREADS - 1) / FGW_THREADS;      // matrix entries per thread (N <= 4R)
    const int e_dq = FGW_THREADS / N, e_dr = FGW_THREADS - e_dq * N;
    const int e_i0 = tid / N, e_j0 = tid - e_i0 * N;
    auto for_entries = [&](auto fn) {                                     // fn(u, i * P + j) for this thread's entries t = tid + 256 u < N * N
        int i = e_i0, j = e_j0;
#pragma unroll
        for (int u = 0; u < EPT; ++u) {
            if (tid + u * FGW_THREADS < NN) fn(u, i * P + j);
            j += e_dr; i += e_dq;
            if (j >= N) { j -= N; ++i; }
        }
    };
    const bool warm = outer > 0 && prm.warmstart;
    const bool stage_yz = d <= 2 * P && !y_zero;                          // Y / Z through LDS for the prologue's dot(Y, Z)
    // the input graph's structure: dense [N,N] floats, or (C2T = bytes) straight from the ragged neighbour lists (FgwAdj)
    const bool ragged = sizeof(C2T) == 1 && adj.rowptr != nullptr;
    float c2v[EPT];
#pragma unroll
    for (int u = 0; u < EPT; ++u) { const int t = tid + u * FGW_THREADS; c2v[u] = ragged ? 0.f : C2[t < NN ? t : NN - 1]; }
    // C1 (and the warm-start coupling) cannot go to LDS before the dot product has consumed Z, which is staged over C1's storage: they
    // are requested when the product's MFMAs are done (below) instead of being parked in registers across it.
    double c1v[EPT];
    float tv[EPT];
    auto load_c1_t = [&]() {
#pragma unroll
        for (int u = 0; u < EPT; ++u) {
            const int t = tid + u * FGW_THREADS, tc = t < NN ? t : NN - 1;
            c1v[u] = C1[tc]; tv[u] = warm ? Tg[tc] : 0.f;
        }
    };
    if (!stage_yz) load_c1_t();
    const double p_own = tid < N ? (pb ? (double)pb[(size_t)b * N + tid] : fc.inv_n) : 1.0;
    const double q_own = tid < N ? (ps ? (double)ps[((size_t)b * D.K + s) * N + tid] : fc.inv_n) : 1.0;
    double vec_own;
    {
        const int v = tid >> 6, i = tid & 63;
        const double *src = (v == 0 || v == 2) ? yvec + (size_t)b * 2 * N + (v == 0 ? N : 0) : zvec + ((size_t)b * D.K + s) * 2 * N + (v == 1 ? N : 0);
        vec_own = src[i < N ? i : N - 1];
        vec_own = i < N ? vec_own : 0.0;
    }
    // (padded nodes as one node — see the detection below: the checks of the barycenter's structure matrix read global memory; with the ragged lists the
    // number of real nodes is known here, so those loads fly with the staging loads)
    int nb = N;                                                         // first padded node = number of real nodes (dense structure: found below)
    if (ragged) nb = min(adj.gptr[cid + 1] - adj.gptr[cid], N);
    const bool pad_try = !pb && !ps && !Ypart && !prm.fixed_structure && !prm.fixed_features;
    auto c1_block_ok = [&](int nb_) {                                   // rows nb_+1 .. N-1 of C equal to row nb_, its columns likewise, the block constant (1e-9)
        auto near = [](double a, double b_) { return fabs(a - b_) <= 1e-9 * (fabs(b_) + 1e-30) || a == b_; };
        bool ok = true;
        const int lc = lane < N ? lane : N - 1;
        const double c_nb = C1[(size_t)nb_ * N + lc];                   // row nb_ (lane <-> column)
        const double cbb = __shfl(c_nb, nb_, 64);
        const double want = lane < nb_ ? c_nb : cbb;
        for (int r = nb_ + 1 + w; r < N; r += FGW_WAVES) ok = ok && (lane >= N || near(C1[(size_t)r * N + lc], want));
        ok = ok && (lane < nb_ || lane >= N || near(c_nb, cbb));
        const int mcol = N - nb_;                                       // columns of the block, lane <-> column nb_ + lane
        for (int k = w; k < nb_; k += FGW_WAVES) {
            const double v = C1[(size_t)k * N + nb_ + (lane < mcol ? lane : 0)];
            ok = ok && near(v, __shfl(v, 0, 64));
        }
        return ok;
    };
    bool pad_ok = pad_try && ragged && N - nb >= 2 && nb >= 1;
    if (pad_ok) pad_ok = c1_block_ok(nb);
    if (stage_yz) {
        const int Nd = N * d;
        for (int t0 = tid; t0 < Nd; t0 += 4 * FGW_THREADS) {
            double yv[4]; float zv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { const int t = t0 + u * FGW_THREADS, tc = t < Nd ? t : Nd - 1; yv[u] = Y[tc]; zv[u] = Z[tc]; }
#pragma unroll
            for (int u = 0; u < 4; ++u) { const int t = t0 + u * FGW_THREADS; if (t < Nd) { Yl[t] = yv[u]; Zl[t] = zv[u]; } }
        }
    }
    if (tid < 64) { pq[tid] = p_own; pq[64 + tid] = q_own; }
    vec4[tid] = vec_own;
    for_entries([&](int u, int o) { C2l[o] = (C2T)c2v[u]; if (!stage_yz) C1l[o] = c1v[u]; });
    if constexpr (sizeof(C2T) == 1) {
        if (ragged) {                                                   // (workgroup-uniform) zeros are in place: one thread per edge adds its count
            __syncthreads();
            adj_scatter_lds_bytes<FGW_THREADS>(adj, cid, N, P, reinterpret_cast<unsigned char *>(C2l), tid);
        }
    }
    __syncthreads();
    // ---- A COMPLETE input graph (every pair of its n real nodes adjacent, weight 1; padded nodes isolated — what a conformer of an ESOL- /
    // FreeSolv-sized molecule is under the 10 A cutoff, schnet_no_sum.py:94-100) has C2 = 1 1^T - I on the real block, so the two products against it are
    //     (A C2^T)_ij = [j < n] (sum_{k<n} A_ik - A_ij)          T C2 T^T = t t^T - T_r T_r^T,  t = T 1_r
    // row sums and one product over the n real columns instead of three N^3 products (utils.py:62-64, :67-73).  Detected from the staged matrix itself
    // (dense and ragged inputs alike): n = 1 + the non-zeros of row 0, then every entry is compared with the pattern.  Workgroup-uniform.
    int n_real;
    {
        const bool nz = lane < N && C2l[lane] != (C2T)0;
        n_real = 1 + __popcll(__ballot(nz));
    }
    bool complete;
    {
        bool ok = true;
        int i = e_i0, j = e_j0;
#pragma unroll
        for (int u = 0; u < EPT; ++u) {
            if (tid + u * FGW_THREADS < NN) ok = ok && (C2l[i * P + j] == (C2T)((i != j && i < n_real && j < n_real) ? 1 : 0));
            j += e_dr; i += e_dq;
            if (j >= N) { j -= N; ++i; }
        }
        complete = __syncthreads_and(ok) != 0 && n_real >= 2;
    }
    double *rsum = reinterpret_cast<double *>(smem + L.off_rs);      // [4][N] (complete graphs only; may overlay C2, which that path never reads again)
    // ---- PADDED NODES AS ONE NODE.  The reference pads every conformer of a batch to N = N_max nodes (schnet_no_sum.py:242-252, 282: SURVEY.md Appendix
    // D.1): the m = N - n padded nodes of an input graph are isolated, carry the same feature row and the same mass 1 / N — they are EXCHANGEABLE, and so
    // are the barycenter's nodes n .. N - 1 (init_C = Cs[0] has them isolated, Y starts at zero, every update is permutation-equivariant): all iterates
    // have identical rows / columns there, with C constant on the whole block (its diagonal included: T's rows are identical).  The solve on the
    // (n + 1)-node problem whose last node carries the block's mass m / N is then the SAME iteration (sinkhorn.py:413-416 with the potentials of the merged
    // row / column shifted by log m: the scaling vector g starts at m there instead of 1) and gives T'_iP = sum of the block's entries — provided the
    // norms of the stopping rules count a merged entry m (or m^2) times at 1 / m (1 / m^2) of its value, as the full matrices do (bregman.py:144-147,
    // sinkhorn.py:418-433).  The reduced matrices are the top-left (n + 1)^2 corners of the staged ones (node n stands for its block), so nothing moves:
    // the loops below run to Nx = n + 1, masses and the warm-start coupling take the multiplicity, the results are expanded when they are written.
    // Products shrink by (Nx / N)^3 — ESOL-shaped batches: n = 20 of N = 33 on average.  Checked per coupling (workgroup-uniform), against the data:
    // uniform masses, rows n + 1 .. N - 1 of Z, Y and C equal to row n (1e-9: the update kernel sums border rows in another order), the block of C
    // constant; anything else runs at full size.
    if (!ragged) {                                                      // dense structure: one past the last node that has an edge
        int mx = 1;
        int i = e_i0, j = e_j0;
#pragma unroll
        for (int u = 0; u < EPT; ++u) {
            if (tid + u * FGW_THREADS < NN && C2l[i * P + j] != (C2T)0) mx = max(mx, max(i, j) + 1);
            j += e_dr; i += e_dq;
            if (j >= N) { j -= N; ++i; }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mx = max(mx, __shfl_xor(mx, o, 64));
        int *ired = reinterpret_cast<int *>(red);
        if (lane == 0) ired[w] = mx;
        __syncthreads();
        nb = max(max(ired[0], ired[1]), max(ired[2], ired[3]));
        __syncthreads();
        pad_ok = pad_try && N - nb >= 2 && nb >= 1;
        if (pad_ok) pad_ok = c1_block_ok(nb);
    }
    int m_pad = 0;                                                      // multiplicity of the merged node; 0 = the problem runs at full size
    if (pad_try && N - nb >= 2 && nb >= 1) {                            // (workgroup-uniform condition: the barrier below is taken by all or by none)
        auto nearf = [](double a, double b_) { return fabs(a - b_) <= 1e-9 * (fabs(b_) + 1e-30) || a == b_; };
        bool ok = pad_ok;
        for (int r = nb + 1 + w; r < N; r += FGW_WAVES)                 // feature rows of the input graph and of the barycenter: wavefront <-> row, lane <-> column
            for (int c = lane; c < d; c += 64) {
                if (stage_yz) {
                    ok = ok && nearf((double)Zl[r * d + c], (double)Zl[nb * d + c]) && nearf(Yl[r * d + c], Yl[nb * d + c]);
                } else {
                    ok = ok && nearf((double)Z[(size_t)r * d + c], (double)Z[(size_t)nb * d + c]);
                    if (!y_zero) ok = ok && nearf(Y[(size_t)r * d + c], Y[(size_t)nb * d + c]);
                }
            }
        if (__syncthreads_and(ok) != 0) m_pad = N - nb;
    }
    const int Nx = m_pad ? nb + 1 : N;                                  // logical size of the problem from here on
    const double mult = m_pad ? (double)m_pad : 1.0;
    FGW_PROF(0);      // staging
    const double qj = pq[64 + lane] * ((m_pad && lane == nb) ? mult : 1.0);      // b_j with j = lane (layout A); the merged node carries its block's mass
    const double pi_l = pq[lane] * ((m_pad && lane == nb) ? mult : 1.0);         // a_i with i = lane (layout B)
    const double *r1v = vec4, *r2v = vec4 + 64, *y2v = vec4 + 128, *z2v = vec4 + 192;
    auto base_of = [&](int i, int j, double dot) {                      // utils.py:39-43,154-171, bregman.py:124-125
        double m = -2.0 * dot; m += y2v[i]; m += z2v[j];
        m = m > 0.0 ? m : 0.0;
        return fc.two_alpha * (r1v[i] + r2v[j]) + fc.one_m_alpha * m;
    };
    // ---- base = 2 alpha constC + (1 - alpha) M, M from dot(Y_i, Z_j) on MFMA.  With Y / Z staged over AK | base and C1, the products are
    // held in registers until every wavefront has read its operands (base lands on Y's second half).
    if (y_zero) {
        int i = e_i0, j = e_j0;
#pragma unroll
        for (int u = 0; u < EPT; ++u) {
            if (tid + u * FGW_THREADS < NN) Bl[i * P + j] = base_of(i, j, 0.0);
            j += e_dr; i += e_dq;
            if (j >= N) { j -= N; ++i; }
        }
    } else if (stage_yz) {
        mm_lds2<FGW_WAVES, MAXT, 0, true, true>(Nx, Nx, d, Yl, d, Zl, d, [&]() { load_c1_t(); __syncthreads(); },
                                                [&](int i, int j, double v) { Bl[i * P + j] = base_of(i, j, v); });
        for_entries([&](int u, int o) { C1l[o] = c1v[u]; });            // Z is consumed (the barrier inside the product): C1 takes its place
    } else {
        mm_f64(Nx, Nx, d, [&](int i, int k) { return Y[(size_t)i * d + k]; }, [&](int k, int j) { return (double)Z[(size_t)j * d + k]; },
               [&](int i, int j, double v) { Bl[i * P + j] = base_of(i, j, v); });
    }
    // ---- T0 = G0 (warm start) or p q^T (bregman.py:98-101), written over the vectors once nobody reads them any more
    __syncthreads();
    FGW_PROF(1);      // dot(Y, Z) + base
    if (warm) {
        if (m_pad) {                                                    // the merged row / column holds the SUM of its block's entries
            int i = e_i0, j = e_j0;
#pragma unroll
            for (int u = 0; u < EPT; ++u) {
                if (tid + u * FGW_THREADS < NN) Tl[i * P + j] = (float)((double)tv[u] * ((i == nb ? mult : 1.0) * (j == nb ? mult : 1.0)));
                j += e_dr; i += e_dq;
                if (j >= N) { j -= N; ++i; }
            }
        } else {
            for_entries([&](int u, int o) { Tl[o] = tv[u]; });
        }
    } else {
#pragma unroll
        for (int r = 0; r < R; ++r) {                                   // (first outer iteration only: p comes straight from global memory)
            const int i = w + 4 * r;
            const double pi_r = (pb ? (double)pb[(size_t)b * N + (i < N ? i : N - 1)] : fc.inv_n) * ((m_pad && i == nb) ? mult : 1.0);
            if (i < Nx && lane < Nx) Tl[i * P + lane] = (float)(pi_r * qj);
        }
    }
    __syncthreads();
    FGW_PROF(2);      // T0

    int cpt = 0, sk_total = 0;
    double err = 1.0;
    // A node without mass (fgw.py embeds n != N problems with such nodes) must not enter the first Sinkhorn half-step, which the scaling form
    // takes with g = 1 on every row: such couplings are handed to the exact pass, whose potentials start at -inf on those nodes.
    bool bail = __syncthreads_or(tid < N && (p_own <= 0.0 || q_own <= 0.0)) != 0;      // workgroup-uniform
    while (!bail && err > fc.inner_tol && cpt < prm.max_iter) {         // bregman.py:119
        // Everything per-lane below (LDS offsets, tile indices, fragment pointers) is derived from THIS copy of the thread index, which the
        // optimiser cannot see through: left alone it hoists some sixty loop-invariant offsets out of the loop and spills them.
        int tq = tid, N = Nx, P = D.P;                                  // (shadow the kernel-wide N, P on purpose; N = the logical size)
        asm volatile("" : "+v"(tq), "+s"(N), "+s"(P));                  // uniform offsets (k * P, tile counts ...) are re-derived too: they spill SGPRs
        int nbq = m_pad ? nb : -1;                                      // index of the merged node (-1: none)
        asm volatile("" : "+s"(nbq));
        const int lq = tq & 63, wq = tq >> 6;
        const bool lq_ok = lq < N;
        // ---- A = C1 @ T                                                        (utils.py:48-53)
        mm_lds2<FGW_WAVES, 1, 0, false, false>(N, N, N, C1l, P, Tl, P, [] {}, [&](int i, int j, double v) { AKl[i * P + j] = v; }, tq);
        __syncthreads();
        FGW_PROF(3);  // A = C1 @ T
        // ---- G = A @ (2 C2)^T, Mr = -(base - 2 alpha G) / eps (utils.py:62-64, sinkhorn.py:388), K = exp(Mr - ref_j) with the column
        // reference ref_j = -base_jj / eps: formed where the product leaves its result, written over A once every wavefront has read A.
        int nr = n_real;
        asm volatile("" : "+s"(nr));
        if (complete) {
            // (A C2^T)_ij = [j < n] (a_i - A_ij), a_i = sum_{k<n} A_ik: lane <-> row, wavefront <-> columns k = w, w + 4, ...; the four partial sums
            // of a row meet through `rsum`, K is formed in place (an entry is read and written by the same thread)
            {
                const int lc = lq_ok ? lq : N - 1;
                const double *ar = AKl + lc * P + wq;
                double p0 = 0.0, p1 = 0.0;
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const int q = wq + 4 * r;
                    if (q < nr) { if (r & 1) p1 += ar[4 * r]; else p0 += ar[4 * r]; }      // (wavefront-uniform)
                }
                if (lq_ok) rsum[wq * N + lq] = p0 + p1;
            }
            __syncthreads();
            {
                int i = tq / N, j = tq - i * N;
                const int dq = FGW_THREADS / N, dr = FGW_THREADS - dq * N;
                for (int t = tq; t < N * N; t += FGW_THREADS) {
                    const int o = i * P + j;
                    const double ai = ((rsum[i] + rsum[N + i]) + rsum[2 * N + i]) + rsum[3 * N + i];
                    const double v = j < nr ? ai - AKl[o] : 0.0;
                    const double x = fma(v, fc.four_alpha_inv_eps, (Bl[j * P + j] - Bl[o]) * fc.inv_eps);
                    AKl[o] = exp_fast(x);
                    j += dr; i += dq;
                    if (j >= N) { j -= N; ++i; }
                }
            }
        } else {
            mm_lds2<FGW_WAVES, MAXT, 0, true, true>(N, N, N, AKl, P, C2l, P, [&]() { __syncthreads(); }, [&](int i, int j, double v) {
                const double x = fma(v, fc.four_alpha_inv_eps, (Bl[j * P + j] - Bl[i * P + j]) * fc.inv_eps);      // Mr_ij - ref_j, G = 2 v
                AKl[i * P + j] = exp_fast(x);
            }, tq);
        }
        __syncthreads();
        FGW_PROF(4);  // G, K
        // ---- K into registers in both layouts: kA[r] = K[w + 4r][lane] (lane <-> column), kB[r] = K[lane][w + 4r] (lane <-> row)
        double kA[R], kB[R];
        {
            const int lc = lq_ok ? lq : N - 1;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int q = wq + 4 * r, qc = q < N ? q : N - 1;
                const double a = AKl[qc * P + lc], bb = AKl[lc * P + qc];
                const bool ok = q < N && lq_ok;
                kA[r] = ok ? a : 0.0; kB[r] = ok ? bb : 0.0;
            }
        }
        __syncthreads();                                                // K is in registers: its storage becomes the Sinkhorn scratch
        FGW_PROF(5);  // K -> registers
        // ---- Sinkhorn on the scaling vectors (sinkhorn.py:413-433): f_j = b_j / sum_i K_ij g_i ; g_i = a_i / sum_j K_ij f_j
        const int permL = (lq & 3) * 16 + (lq >> 2);                    // slot of this lane's factor in fvP / gvP
        double *bufCw = bufC + wq * 64 + lq, *bufRw = bufR + wq * 64 + lq;
        const double *fw = fvP + wq * 16, *gw = gvP + wq * 16;
        // this thread's share of a matrix-vector product: sum_r k[r] * v[r], three partial sums (the dependent-FMA chain is what a phase waits for)
        auto dotR = [&](const double (&k)[R], const double *v) {
            double p0 = 0.0, p1 = 0.0, p2 = 0.0;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                if (r % 3 == 0) p0 = fma(k[r], v[r], p0);
                else if (r % 3 == 1) p1 = fma(k[r], v[r], p1);
                else p2 = fma(k[r], v[r], p2);
            }
            return (p0 + p1) + p2;
        };
        int ii = 0;
        double f = 0.0, g = 0.0, colp = 0.0;
        bool have_colp = false;
        for (; ii < prm.num_iter_max; ++ii) {
            if (!have_colp) {                                           // v update (:415): column products against the current g (1 at ii = 0)
                double pc = 0.0;
                if (ii == 0) {
                    // g starts at 1 (u = 0, sinkhorn.py:396) — at the merged node's multiplicity for that row (its m rows enter a column sum)
                    gvP[permL] = lq_ok ? (lq == nbq ? (double)m_pad : 1.0) : 0.0;
                }
                pc = dotR(kA, gw);
                *bufCw = pc;
                __syncthreads();
                colp = ((bufC[lq] + bufC[64 + lq]) + bufC[128 + lq]) + bufC[192 + lq];
            }
            have_colp = false;
            if (__any(lq_ok && !(colp > 1e-150 && colp < 1e150))) { bail = true; break; }
            f = lq_ok ? qj * rcp_pos(colp) : 0.0;
            fvP[permL] = f;                                             // every wavefront writes the same 64 values and reads back its own
            *bufRw = dotR(kB, fw);                                      // u update (:416)
            __syncthreads();
            const double rs = ((bufR[lq] + bufR[64 + lq]) + bufR[128 + lq]) + bufR[192 + lq];
            if (__any(lq_ok && !(rs > 1e-150 && rs < 1e150))) { bail = true; break; }
            g = lq_ok ? pi_l * rcp_pos(rs) : 0.0;
            gvP[permL] = g;
            if (ii % 10 == 0) {                                         // marginal violation (:418-433): || f * (K^T g) - b ||_2
                *bufCw = dotR(kA, gw);
                __syncthreads();
                colp = ((bufC[lq] + bufC[64 + lq]) + bufC[128 + lq]) + bufC[192 + lq];
                have_colp = true;                                       // the next v update starts from these products
                double df = lq_ok ? f * colp - qj : 0.0;
                df = df * df;
                if (lq == nbq) df /= (double)m_pad;                     // the block's m columns, each with 1 / m of the merged residual
                df = wave_sum_d(df);                                   // identical in every wavefront: the break is workgroup-uniform
                if (sqrt(df) < fc.stop_thr) { ++ii; break; }
            }
        }
        if (bail) break;
        sk_total += ii;
        FGW_PROF(6);  // Sinkhorn iterations
        // ---- T = diag(g) K diag(f) (= exp(Mr + u + v), sinkhorn.py:450); err = ||T - Tprev||_F when cpt % 10 == 0 (bregman.py:144-147)
        double e2 = 0.0;
        {
            float *trow = Tl + wq * P + lq;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const bool ok = lq_ok && wq + 4 * r < N;
                float *tp = ok ? trow + 4 * r * P : t_dummy;
                const float tn = (float)((gw[r] * kA[r]) * f);
                double df = ok ? (double)tn - (double)*tp : 0.0;
                df *= df;
                if (nbq >= 0) {                                         // a merged entry stands for m (m^2) entries of 1 / m (1 / m^2) of its value
                    const double im = 1.0 / (double)m_pad;
                    df *= (lq == nbq ? im : 1.0) * (wq + 4 * r == nbq ? im : 1.0);
                }
                e2 += df;
                *tp = tn;
            }
        }
        if (cpt % 10 == 0) err = sqrt(block_sum_d(e2, red));
        else __syncthreads();
        ++cpt;
        FGW_PROF(7);  // T store + err
    }
    if (bail) {                                                         // nothing has been written: the launcher re-runs this coupling on the exact path
        if (tid == 0) { redo[cid] = 1; atomicOr(&info[b * 4 + 3], 1); }      // info flag bit 0: a coupling of this molecule took the second pass
        return;
    }
    __syncthreads();
    {   // per-thread global addresses are re-derived here instead of being kept (spilled) from the prologue
        int te = tid;
        asm volatile("" : "+v"(te));
        int i = te / N, j = te - i * N;
        const int dq = FGW_THREADS / N, dr = FGW_THREADS - dq * N;
        if (m_pad) {                                                    // every entry of the block gets its share of the merged entry
            const float im = (float)(1.0 / mult);
            for (int t = te; t < NN; t += FGW_THREADS) {
                Tg[t] = Tl[min(i, nb) * P + min(j, nb)] * ((i >= nb ? im : 1.0f) * (j >= nb ? im : 1.0f));
                j += dr; i += dq;
                if (j >= N) { j -= N; ++i; }
            }
        } else {
            for (int t = te; t < NN; t += FGW_THREADS) {
                Tg[t] = Tl[i * P + j];
                j += dr; i += dq;
                if (j >= N) { j -= N; ++i; }
            }
        }
    }
    if (tid == 0) { atomicAdd(&info[b * 4 + 1], cpt); atomicAdd(&info[b * 4 + 2], sk_total); redo[cid] = 0; if (m_pad) atomicOr(&info[b * 4 + 3], 2); }      // (flag bit 1: solved with its padded nodes merged)
    FGW_PROF(8);      // T -> global

    // ---- contributions to the barycenter update while T is resident
    if (!prm.fixed_features && Ypart) {                                 // Ypart = T @ Z (utils.py:90-95); nullptr: the update kernel forms it from T itself
        fgw_part_t *Yp = Ypart + ((size_t)b * D.K + s) * N * d;
        if (yz_lds) {                                                   // C1 is dead: Z through its storage, one coalesced pass
            const int Nd = N * d;
            for (int t0 = tid; t0 < Nd; t0 += 4 * FGW_THREADS) {
                float zv[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) { const int t = t0 + u * FGW_THREADS; zv[u] = Z[t < Nd ? t : Nd - 1]; }
#pragma unroll
                for (int u = 0; u < 4; ++u) { const int t = t0 + u * FGW_THREADS; if (t < Nd) Zl[t] = zv[u]; }
            }
            __syncthreads();
            mm_lds2<FGW_WAVES, 1, 0, false, false>(N, d, N, Tl, P, Zl, d, [] {}, [&](int i, int c, double v) { Yp[(size_t)i * d + c] = (fgw_part_t)v; });
        } else {
            mm_f64(N, d, N, [&](int i, int k) { return (double)Tl[i * P + k]; }, [&](int k, int c) { return (double)Z[(size_t)k * d + c]; },
                   [&](int i, int c, double v) { Yp[(size_t)i * d + c] = (fgw_part_t)v; });
        }
    }
    FGW_PROF(9);      // Ypart = T @ Z
    if (!prm.fixed_structure) {                                         // Cpart = T @ C2 @ T^T               (utils.py:67-73)
        fgw_part_t *Cp = Cpart + ((size_t)b * D.K + s) * NN;
        // (merged problem: the product is formed at the logical size into `base`'s storage — free since the loop ended — and expanded on the way out:
        // entry (i, j) of the block rows / columns is 1 / m (1 / m^2) of the merged entry, both indices being barycenter nodes)
        if (complete) {                                                 // t t^T - T_r T_r^T (see the detection above)
            {
                const int lc = lane < Nx ? lane : Nx - 1;
                const float *tr = Tl + lc * P + w;
                double p0 = 0.0, p1 = 0.0;
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const int q = w + 4 * r;
                    if (q < n_real) { if (r & 1) p1 += (double)tr[4 * r]; else p0 += (double)tr[4 * r]; }
                }
                if (lane < Nx) rsum[w * N + lane] = p0 + p1;
            }
            __syncthreads();
            mm_lds2<FGW_WAVES, 1, 0, true, false>(Nx, Nx, n_real, Tl, P, Tl, P, [] {}, [&](int i, int j, double v) {
                const double ti = ((rsum[i] + rsum[N + i]) + rsum[2 * N + i]) + rsum[3 * N + i];
                const double tj = ((rsum[j] + rsum[N + j]) + rsum[2 * N + j]) + rsum[3 * N + j];
                if (m_pad) Bl[i * P + j] = ti * tj - v;
                else Cp[i * N + j] = (fgw_part_t)(ti * tj - v);
            });
        } else {
            mm_lds2<FGW_WAVES, 1, 0, false, false>(Nx, Nx, Nx, Tl, P, C2l, P, [] {}, [&](int i, int j, double v) { AKl[i * P + j] = v; });
            __syncthreads();
            mm_lds2<FGW_WAVES, 1, 0, true, false>(Nx, Nx, Nx, AKl, P, Tl, P, [] {}, [&](int i, int j, double v) {
                if (m_pad) Bl[i * P + j] = v;
                else Cp[i * N + j] = (fgw_part_t)v;
            });
        }
        if (m_pad) {
            __syncthreads();
            const double im = 1.0 / mult;
            int i = tid / N, j = tid - i * N;
            const int dq = FGW_THREADS / N, dr = FGW_THREADS - dq * N;
            for (int t = tid; t < NN; t += FGW_THREADS) {
                Cp[t] = (fgw_part_t)(Bl[min(i, nb) * P + min(j, nb)] * ((i >= nb ? im : 1.0) * (j >= nb ? im : 1.0)));
                j += dr; i += dq;
                if (j >= N) { j -= N; ++i; }
            }
        }
    }
    FGW_PROF(10);     // Cpart = T @ C2 @ T^T
    FGW_PROF_FLUSH;
    FGW_PROF_TRACE(Nx | (cpt << 8) | (sk_total << 20));
}

// Per-index vectors of the gradient's constant part (init_matrix, utils.py:39-43) and of the feature cost (utils.py:154-171).
// out[0..N) = |y_i|^2, out[N..2N) = r1_i = sum_k f1(C[i,k]) p_k with f1(a) = a^2 (square loss) or a log(a + 1e-15) - a (kl).
// LPI = NT / 64 lanes per index (N <= 64), strided partial sums combined by xor-shuffles: a fixed order, bitwise reproducible.
template <int NT>
__device__ __forceinline__ void molecule_vectors(const double *__restrict__ Y, const double *__restrict__ C, const float *__restrict__ p, int N,
                                                 int d, bool kl, double *__restrict__ out, int which = 2) {      // which: 0 = |y|^2 only, 1 = r1 only, 2 = both
    constexpr int LPI = NT / 64;
    const int i = (int)threadIdx.x / LPI, sub = (int)threadIdx.x % LPI;
    double y2 = 0.0, r1 = 0.0;
    if (i < N) {
        if (which != 1)
            for (int c = sub; c < d; c += LPI) { const double v = Y[(size_t)i * d + c]; y2 += v * v; }
        if (which != 0)
            for (int k = sub; k < N; k += LPI) {
                const double c1 = C[i * N + k], pk = p ? (double)p[k] : 1.0 / (double)N;
                r1 += (kl ? c1 * log(c1 + 1e-15) - c1 : c1 * c1) * pk;
            }
    }
#pragma unroll
    for (int o = 1; o < LPI; o <<= 1) { y2 += __shfl_xor(y2, o, 64); r1 += __shfl_xor(r1, o, 64); }
    if (i < N && sub == 0) { if (which != 1) out[i] = y2; if (which != 0) out[N + i] = r1; }
}

// One workgroup per (molecule, input graph): the static vectors |z_j|^2 and r2_j = sum_k q_k f2(C2[j,k]) (f2(b) = b^2, or b for kl),
// and (s == 0) the molecule's initial |y_i|^2, r1_i.  Launched once per solve, after k_fgw_init.
// The workgroup of input graph 0 also initialises its molecule (what k_fgw_init does on the N > 64 path: C <- init_C or Cs[b, 0], Y <- init_Y
// or 0, barycenter.py:57-77; flags, counters, error log): one launch instead of two at the head of every solve.
__global__ void __launch_bounds__(256) k_fgw_small_vectors(const float *__restrict__ Ys, const float *__restrict__ Cs, const float *__restrict__ ps,
                                                           const float *__restrict__ pb, FgwDims D, int kl, double *__restrict__ Cw,
                                                           double *__restrict__ Yw, double *__restrict__ zvec, double *__restrict__ yvec,
                                                           const float *__restrict__ init_C, const float *__restrict__ init_Y, int max_iter,
                                                           int *__restrict__ active, int *__restrict__ info, float *__restrict__ errs,
                                                           float *__restrict__ Yout, float *__restrict__ Cout, FgwAdj adj) {
    const int b = blockIdx.x / D.K, s = blockIdx.x % D.K;
    const int N = D.N, d = D.d;
    // ragged structure (FgwAdj): this graph's adjacency counts as bytes in LDS (N <= 64), built once here; the dense Cs is never read
    __shared__ __attribute__((aligned(16))) unsigned char cm[64 * 65 + 16];
    const int Pm = D.P;
    if (adj.rowptr) {
        for (int t = threadIdx.x; t < (64 * 65 + 16) / 4; t += 256) reinterpret_cast<unsigned *>(cm)[t] = 0u;
        __syncthreads();
        adj_scatter_lds_bytes<256>(adj, (int)blockIdx.x, N, Pm, cm, (int)threadIdx.x);
        __syncthreads();
    }
    if (s == 0) {                                                       // (workgroup-uniform)
        const int NN = N * N, Nd = N * d;
        const float *c0 = init_C ? init_C + (size_t)b * NN : Cs + (size_t)b * D.K * NN;      // init_C = Cs[0] (schnet_no_sum.py:303)
        for (int t = threadIdx.x; t < NN; t += 256) {
            const float cv = (!init_C && adj.rowptr) ? (float)cm[(t / N) * Pm + (t % N)] : c0[t];
            Cw[(size_t)b * NN + t] = (double)cv; Cout[(size_t)b * NN + t] = cv;
        }
        for (int t = threadIdx.x; t < Nd; t += 256) {
            const float y = init_Y ? init_Y[(size_t)b * Nd + t] : 0.f;                          // barycenter.py:76-77
            Yw[(size_t)b * Nd + t] = (double)y; Yout[(size_t)b * Nd + t] = y;
        }
        if (threadIdx.x == 0) { fgw_active_init(active, D.B, b); info[b * 4 + 0] = 0; info[b * 4 + 1] = 0; info[b * 4 + 2] = 0; info[b * 4 + 3] = 0; }
        for (int t = threadIdx.x; t < 2 * max_iter; t += 256) errs[(size_t)b * 2 * max_iter + t] = __builtin_nanf("");
        __syncthreads();                                                // Cw / Yw of this molecule are read back below (same workgroup)
    }
    const float *Z = Ys + ((size_t)b * D.K + s) * N * d;
    const float *C2 = Cs + ((size_t)b * D.K + s) * N * N;
    const float *q = ps ? ps + ((size_t)b * D.K + s) * N : nullptr;
    const int j = (int)threadIdx.x >> 2, sub = (int)threadIdx.x & 3;
    double z2 = 0.0, r2 = 0.0;
    if (j < N) {
        for (int c = sub; c < d; c += 4) { const double v = (double)Z[(size_t)j * d + c]; z2 += v * v; }
        for (int k = sub; k < N; k += 4) {
            const double c2 = adj.rowptr ? (double)cm[j * Pm + k] : (double)C2[j * N + k], qk = q ? (double)q[k] : 1.0 / (double)N;
            r2 += qk * (kl ? c2 : c2 * c2);
        }
    }
    z2 += __shfl_xor(z2, 1, 64); z2 += __shfl_xor(z2, 2, 64);
    r2 += __shfl_xor(r2, 1, 64); r2 += __shfl_xor(r2, 2, 64);
    double *zo = zvec + ((size_t)b * D.K + s) * 2 * N;
    if (j < N && sub == 0) { zo[j] = z2; zo[N + j] = r2; }
    if (s == 0)
        molecule_vectors<256>(Yw + (size_t)b * N * d, Cw + (size_t)b * N * N, pb ? pb + (size_t)b * N : nullptr, N, d, kl != 0, yvec + (size_t)b * 2 * N);
    // ---- FgwAdj.order: the molecules by descending number of real nodes, for the coupling kernel's workgroup dealing.  The LAST workgroup does it
    // (its own work above is the shortest wait for the launch).
    if (adj.order && blockIdx.x == gridDim.x - 1) fgw_order_by_size<256>(adj, D.B, D.K, (int)threadIdx.x);
}

// Barycenter update from the per-graph contributions (utils.py:67-95, barycenter.py:112).  TWO workgroups per molecule (round 4), because the two
// halves share nothing: block b updates the FEATURES Y (and |y_i|^2), block B + b the STRUCTURE C (and r1_i); each logs its own error and raises
// its own "still moving" flag.  A molecule is active in outer iteration o + 1 if either flag of iteration o is set (fgw_active): the flags are
// double-buffered by the parity of the outer iteration, so that a workgroup of iteration o never reads a flag its sibling is writing.
//
// Y_FROM_T (the N <= 64 path): the feature contributions T_s Z_s are formed HERE, from the couplings the coupling kernels wrote and the input
// features, instead of being written as fp64 [N,d] matrices by every coupling workgroup and read back: that product was the largest phase of
// k_fgw_coupling_fast (23 of 124 us per workgroup and launch: Z fetched a second time at the kernel's tail, behind its own stores, 17 KB of fp64
// results per coupling) and 2 x 21.6 MB of traffic per launch at cfg2.  `chunk` graphs at a time are staged in LDS (T_s, Z_s as fp32: 12.8 KB per
// graph at N = 33, d = 64); a wavefront owns one padded 16 x 16 output tile per round and sums the graphs' products (fp64 MFMA, the same
// converted operands as in the coupling kernel) in registers in the order s = 0 .. K - 1.
constexpr int UPD_THREADS = 512;        // eight wavefronts: with <= 102 VGPRs the two halves of a molecule are resident on one CU together
template <bool Y_FROM_T>
__global__ void __launch_bounds__(UPD_THREADS) k_fgw_update_parts(
    const float *__restrict__ pb, const float *__restrict__ lambdas, FgwDims D, conan_fgw_params prm, int outer,
    const fgw_part_t *__restrict__ Ypart, const fgw_part_t *__restrict__ Cpart, double *__restrict__ Cw, double *__restrict__ Yw,
    int *__restrict__ active, int *__restrict__ info, float *__restrict__ errs, float *__restrict__ Yout, float *__restrict__ Cout,
    double *__restrict__ yvec, const float *__restrict__ Tw, const float *__restrict__ Ys, int chunk) {
    __shared__ double red[UPD_THREADS / 64 + 1];
    extern __shared__ __attribute__((aligned(16))) char usmem[];
    const int part = (int)blockIdx.x >= D.B ? 1 : 0;                    // 0: features, 1: structure
    const int b = (int)blockIdx.x - part * D.B;
    const int N = D.N, d = D.d, K = D.K, NN = N * N, Nd = N * d;
    const int tid = threadIdx.x;
    int *flag_out = active + (outer & 1) * 2 * D.B + part * D.B + b;
    if (!fgw_active(active, D.B, b, outer)) {                           // converged earlier: stays converged
        if (tid == 0) *flag_out = 0;
        return;
    }
    constexpr int U = 4;
    double e2 = 0.0;
    if (part == 0) {
        if (!prm.fixed_features) {
            double *Yb = Yw + (size_t)b * Nd;
            if constexpr (Y_FROM_T) {
                float *Tl = reinterpret_cast<float *>(usmem), *Zl = Tl + (size_t)chunk * NN;
                const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), li = lane & 15, lk = lane >> 4;
                const int Mq = (N + 15) >> 4, Nq = (d + 15) >> 4;
                constexpr int NWU = UPD_THREADS / 64, TW = 2;                          // two padded 16 x 16 output tiles per wavefront and round: one round up to N = 64, d = 64
                for (int t0 = 0; t0 < Mq * Nq; t0 += NWU * TW) {
                    int ti0[TW], tj0[TW], tic[TW], tjc[TW];
                    bool mine[TW];                                                  // (wave-uniform)
                    f64x4 sum[TW];
#pragma unroll
                    for (int w = 0; w < TW; ++w) {
                        const int t = t0 + wave + w * NWU;
                        mine[w] = t < Mq * Nq;
                        ti0[w] = mine[w] ? (t / Nq) << 4 : 0; tj0[w] = mine[w] ? (t % Nq) << 4 : 0;
                        tic[w] = min(ti0[w] + li, N - 1); tjc[w] = min(tj0[w] + li, d - 1);      // clamped: outputs beyond N / d are not stored
                        sum[w] = f64x4{0.0, 0.0, 0.0, 0.0};
                    }
                    for (int s0 = 0; s0 < K; s0 += chunk) {
                        const int cs = min(chunk, K - s0);
                        __syncthreads();                                            // the previous stage is consumed
                        const float *tg = Tw + ((size_t)b * K + s0) * NN, *zg = Ys + ((size_t)b * K + s0) * Nd;
                        // (every load of a batch is requested before the first LDS store: one memory round trip per batch, not one per element)
                        auto stage = [&](const float *__restrict__ src, float *dst, int n) {
                            constexpr int SB = 12;
                            for (int base = 0; base < n; base += SB * UPD_THREADS) {
                                float v[SB];
#pragma unroll
                                for (int u = 0; u < SB; ++u) { const int q = base + u * UPD_THREADS + tid; v[u] = src[q < n ? q : n - 1]; }
#pragma unroll
                                for (int u = 0; u < SB; ++u) { const int q = base + u * UPD_THREADS + tid; if (q < n) dst[q] = v[u]; }
                            }
                        };
                        stage(tg, Tl, cs * NN);
                        stage(zg, Zl, cs * Nd);
                        __syncthreads();
#pragma unroll
                        for (int w = 0; w < TW; ++w)
                            if (mine[w])
                                for (int q = 0; q < cs; ++q) {
                                    const double lam = lambdas ? (double)lambdas[s0 + q] : 1.0 / (double)K;
                                    const f64x4 r = mm2_tile<0, false>(Tl + (size_t)q * NN + tic[w] * N + lk, Zl + (size_t)q * Nd + lk * d + tjc[w], d, N, lk);
                                    sum[w] = sum[w] + lam * r;                      // utils.py:94, s = 0 .. K - 1
                                }
                    }
#pragma unroll
                    for (int w = 0; w < TW; ++w) {
                        const int jb = tj0[w] + li;
                        if (mine[w] && jb < d) {
#pragma unroll
                            for (int qq = 0; qq < 4; ++qq) {
                                const int i = ti0[w] + lk + 4 * qq;
                                if (i < N) {
                                    const double pw = pb ? (double)pb[(size_t)b * N + i] : 1.0 / (double)N;
                                    const double yn = sum[w][qq] * (pw > 0.0 ? 1.0 / pw : 0.0);      // a node without mass (fgw.py embeds n != N problems with such nodes) keeps a zero row
                                    const double df = yn - Yb[i * d + jb];
                                    e2 += df * df;
                                    Yb[i * d + jb] = yn;
                                    Yout[(size_t)b * Nd + i * d + jb] = (float)yn;
                                }
                            }
                        }
                    }
                }
            } else {
                // The K contributions of U consecutive elements per thread are requested together (K * U loads in flight): the kernel is a
                // handful of L2 round trips per molecule, so its time is the number of dependent trips, not the byte count.
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
                                const double pw = pb ? (double)pb[(size_t)b * N + i] : 1.0 / (double)N;
                                const double pinv = pw > 0.0 ? 1.0 / pw : 0.0;       // a node without mass keeps a zero row
                                acc[u] += lam * Ypart[((size_t)b * K + s) * Nd + t] * pinv;          // utils.py:94
                            }
                        }
                    }
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        const int t = t0 + u * UPD_THREADS;
                        if (t < Nd) {
                            const double df = acc[u] - old[u];
                            e2 += df * df;
                            Yb[t] = acc[u];
                            Yout[(size_t)b * Nd + t] = (float)acc[u];
                        }
                    }
                }
            }
        }
    } else if (!prm.fixed_structure) {
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
                    const double cn = pi * pj > 0.0 ? (prm.loss_fun ? exp(acc[u] / (pi * pj)) : acc[u] / (pi * pj)) : 0.0;     // :72-73 / :86-87 (massless nodes: zero)
                    const double df = cn - old[u];
                    e2 += df * df;
                    Cb[t] = cn;
                    Cout[(size_t)b * NN + t] = (float)cn;
                }
            }
        }
    }
    const double err = sqrt(block_sum_d<UPD_THREADS / 64>(e2, red));       // (its barriers also publish Yb / Cb to the whole workgroup)
    if (yvec) molecule_vectors<UPD_THREADS>(Yw + (size_t)b * Nd, Cw + (size_t)b * NN, pb ? pb + (size_t)b * N : nullptr, N, d, prm.loss_fun != 0,
                                            yvec + (size_t)b * 2 * N, part);
    if (tid == 0) {
        errs[((size_t)b * 2 + part) * prm.max_iter + outer] = (float)err;  // err_feature / err_structure
        info[b * 4 + 0] = outer + 1;                                       // (both halves write the same value)
        *flag_out = err > (double)prm.tol ? 1 : 0;                         // barycenter.py:112: the loop continues while either error exceeds tol
    }
}

inline size_t small_lds(int N, int d) {
    const size_t NP = (size_t)N * fgw_pitch(N);
    (void)d; return NP * 8 * 4 + (256 + 128 + 8 + (NP >= (size_t)SK_SCRATCH_DOUBLES ? 0 : SK_SCRATCH_DOUBLES)) * 8 + NP * 4 * 2;
}

}  // namespace

bool conan_fgw_small_supported(int N, int d) { return N <= 64 && small_lds(N, d) <= 160 * 1024; }

size_t conan_fgw_part_offset(int B, int K, int N, int d) {
    return (((size_t)B * K * N * d + (size_t)B * K * N * N) * sizeof(fgw_part_t) + 15) & ~(size_t)15;
}
size_t conan_fgw_small_part_bytes(int B, int K, int N, int d) {
    // Ypart [B,K,N,d] + Cpart [B,K,N,N] (fgw_part_t); zvec [B,K,2N] + yvec [B,2N], fp64; redo [B,K] int32
    return conan_fgw_part_offset(B, K, N, d) + ((size_t)B * K * 2 * N + (size_t)B * 2 * N) * 8 + (size_t)B * K * 4 + 512;
}

void conan_fgw_small_prepare(const float *Ys, const float *Cs, const float *ps, const float *pb, FgwDims D, conan_fgw_params prm,
                             double *Cw, double *Yw, double *zvec, double *yvec, const float *init_C, const float *init_Y, int *active, int *info,
                             float *errs, float *Yout, float *Cout, FgwAdj adj, hipStream_t s) {
    k_fgw_small_vectors<<<D.B * D.K, 256, 0, s>>>(Ys, Cs, ps, pb, D, prm.loss_fun, Cw, Yw, zvec, yvec, init_C, init_Y, prm.max_iter, active, info,
                                                 errs, Yout, Cout, adj);
}

template <int R, int MAXT, typename C2T>
static void launch_fast_t(const float *Ys, const float *Cs, const float *ps, const float *pb, FgwDims D, conan_fgw_params prm, int outer, int y_zero,
                        const double *Cw, const double *Yw, const int *active, float *Tw, int *info, fgw_part_t *Ypart, fgw_part_t *Cpart,
                        const double *zvec, const double *yvec, int *redo, FgwAdj adj, hipStream_t s) {
    const size_t lds = fast_lds<C2T>(D.N).bytes;
    if (lds > 64 * 1024)
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_fgw_coupling_fast<R, MAXT, C2T>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)lds);
    const FastConst fc = fast_const(prm, D.N);
    k_fgw_coupling_fast<R, MAXT, C2T><<<D.B * D.K, FGW_THREADS, lds, s>>>(Ys, Cs, ps, pb, D, prm, fc, outer, y_zero, Cw, Yw, active, Tw, info, Ypart,
                                                                          Cpart, zvec, yvec, redo, adj);
}
template <int R, typename C2T>
static void launch_fast(const float *Ys, const float *Cs, const float *ps, const float *pb, FgwDims D, conan_fgw_params prm, int outer, int y_zero,
                        const double *Cw, const double *Yw, const int *active, float *Tw, int *info, fgw_part_t *Ypart, fgw_part_t *Cpart,
                        const double *zvec, const double *yvec, int *redo, FgwAdj adj, hipStream_t s) {
    const int tpw = fast_tiles_per_wave(D.N);                          // <= ceil(ceil(4R / 16)^2 / 4)
#define ARGS Ys, Cs, ps, pb, D, prm, outer, y_zero, Cw, Yw, active, Tw, info, Ypart, Cpart, zvec, yvec, redo, adj, s
    if constexpr (R <= 6) launch_fast_t<R, 1, C2T>(ARGS);
    else if constexpr (R <= 12) { if (tpw <= 1) launch_fast_t<R, 1, C2T>(ARGS); else launch_fast_t<R, 3, C2T>(ARGS); }
    else { if (tpw <= 1) launch_fast_t<R, 1, C2T>(ARGS); else if (tpw <= 3) launch_fast_t<R, 3, C2T>(ARGS); else launch_fast_t<R, 4, C2T>(ARGS); }
#undef ARGS
}

bool conan_fgw_fast_supported(int N, int d, int small_int) {
    if (N > 64 || N < 4) return false;
    const size_t lds = small_int ? fast_lds<unsigned char>(N).bytes : fast_lds<float>(N).bytes;
    return lds <= 160 * 1024 && d >= 1;
}

void conan_fgw_small_coupling(const float *Ys, const float *Cs, const float *ps, const float *pb, FgwDims D,
                              conan_fgw_params prm, int outer, int y_zero, const double *Cw, const double *Yw,
                              const int *active, float *Tw, int *info, fgw_part_t *Ypart, fgw_part_t *Cpart, const double *zvec,
                              const double *yvec, int *redo, FgwAdj adj, hipStream_t s) {
    const size_t lds = small_lds(D.N, D.d);
    const int R = (D.N + 3) / 4;
    const int grid = D.B * D.K;
    // Square loss: the round-3 kernel first; whatever it hands back (redo[b, s] = 1: a Sinkhorn sum left the fp64-safe range) is solved
    // by the kernel with the exact log-domain path, launched over the same grid with an early exit for everything else.
    const int *only = nullptr;
    if (!prm.loss_fun && redo && conan_fgw_fast_supported(D.N, D.d, prm.cs_small_int)) {
#define FAST(RR)                                                                                                                     \
    do {                                                                                                                             \
        if (prm.cs_small_int) launch_fast<RR, unsigned char>(Ys, Cs, ps, pb, D, prm, outer, y_zero, Cw, Yw, active, Tw, info, Ypart, Cpart, zvec, yvec, redo, adj, s); \
        else launch_fast<RR, float>(Ys, Cs, ps, pb, D, prm, outer, y_zero, Cw, Yw, active, Tw, info, Ypart, Cpart, zvec, yvec, redo, adj, s);     \
    } while (0)
        if (R <= 6) FAST(6);
        else if (R <= 9) FAST(9);
        else if (R <= 12) FAST(12);
        else FAST(16);
#undef FAST
        only = redo;
    }
#define LAUNCH_(RR, SEC, GRID)                                                                                                  \
    do {                                                                                                                        \
        if (lds > 64 * 1024)                                                                                                    \
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_fgw_coupling_small<RR, KLV, SEC>),                     \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                                    \
        k_fgw_coupling_small<RR, KLV, SEC><<<GRID, FGW_THREADS, lds, s>>>(Ys, Cs, ps, pb, D, prm, outer, y_zero, Cw, Yw, active, Tw, \
                                                                info, Ypart, Cpart, zvec, yvec, only, adj);                     \
    } while (0)
#define LAUNCH(RR)                                                                                                              \
    do {                                                                                                                        \
        if constexpr (KLV) LAUNCH_(RR, false, grid);                                                                            \
        else if (only) LAUNCH_(RR, true, (grid + 63) / 64);                                                                     \
        else LAUNCH_(RR, false, grid);                                                                                          \
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
#undef LAUNCH_
}

void conan_fgw_small_update(const float *pb, const float *lambdas, FgwDims D, conan_fgw_params prm, int outer,
                            const fgw_part_t *Ypart, const fgw_part_t *Cpart, double *Cw, double *Yw, int *active, int *info,
                            float *errs, float *Yout, float *Cout, double *yvec, const float *Tw, const float *Ys, hipStream_t s) {
    // Tw / Ys given: the feature contributions are formed in this kernel from the couplings (conan_fgw_update_chunk > 0; the coupling kernels were
    // then launched with Ypart = nullptr)
    const int chunk = Tw ? conan_fgw_update_chunk(D.K, D.N, D.d, D.B) : 0;
    if (chunk > 0) {
        const size_t lds = (size_t)chunk * ((size_t)D.N * D.N + (size_t)D.N * D.d) * 4;
        if (lds > 64 * 1024)
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_fgw_update_parts<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        k_fgw_update_parts<true><<<2 * D.B, UPD_THREADS, lds, s>>>(pb, lambdas, D, prm, outer, nullptr, Cpart, Cw, Yw, active, info, errs, Yout, Cout, yvec, Tw, Ys, chunk);
    } else {
        k_fgw_update_parts<false><<<2 * D.B, UPD_THREADS, 0, s>>>(pb, lambdas, D, prm, outer, Ypart, Cpart, Cw, Yw, active, info, errs, Yout, Cout, yvec, nullptr, nullptr, 0);
    }
}

// graphs per LDS stage of the update kernel's own T_s Z_s products (0: not on that path — N > 64, or one graph does not fit)
int conan_fgw_update_chunk(int K, int N, int d, int B) {
    if (N > 64) return 0;
    // two workgroups per CU (the molecule's two halves run side by side) — unless the launch has no more workgroups than the chip has CUs
    const size_t per = ((size_t)N * N + (size_t)N * d) * 4, budget = (2 * B <= 256 ? 144 : 72) * 1024;
    if (per > budget) return 0;
    const int c = (int)(budget / per);
    return c < K ? c : K;
}
