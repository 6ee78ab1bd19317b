// Shared device helpers of the FGW barycenter kernels (fgw.hip: generic path, fgw_small.hip: register-resident path).
#pragma once
#include "common.h"

// Phase timing of the coupling kernels (tools/fgw_phase_profile.py builds a private library with -DCONAN_FGW_PROFILE; the product
// library is compiled without it and contains none of this).  Thread 0 of every workgroup accumulates the 100 MHz wall-clock ticks
// between marks in registers and adds them to the global slots once, at the end.
#ifdef CONAN_FGW_PROFILE
static __device__ long long g_fgw_prof[32];            // one copy per translation unit (no relocatable device code)
#define FGW_PROF_ACCESSOR(name)                                                          \
    extern "C" int name(long long *out, int reset) {                                     \
        if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(g_fgw_prof), sizeof(long long) * 32) != hipSuccess) return -2; \
        if (reset) { long long z[32] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_fgw_prof), z, sizeof(z)) != hipSuccess) return -2; } \
        return 0;                                                                        \
    }
// Placement trace (tools/fgw_placement.py): one record per workgroup — block, a tag (the size its problem ran at), the HW_ID and XCC_ID registers
// (which CU of which XCD it ran on), first and last wall-clock tick.
static __device__ long long g_fgw_trace[2 + 6 * 8192];
#define FGW_PROF_TRACE_ACCESSOR(name)                                                    \
    extern "C" int name(long long *out, int reset) {                                     \
        if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(g_fgw_trace), sizeof(long long) * (2 + 6 * 8192)) != hipSuccess) return -2; \
        if (reset) { long long z[2] = {0, 0}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_fgw_trace), z, sizeof(z)) != hipSuccess) return -2; } \
        return 0;                                                                        \
    }
#define FGW_PROF_TRACE(tag)                                                              \
    do {                                                                                 \
        if (threadIdx.x == 0) {                                                          \
            const unsigned long long i_ = atomicAdd(reinterpret_cast<unsigned long long *>(&g_fgw_trace[0]), 1ull); \
            if (i_ < 8192) {                                                             \
                long long *r_ = &g_fgw_trace[2 + 6 * i_];                                \
                r_[0] = (long long)blockIdx.x; r_[1] = (long long)(tag);                 \
                r_[2] = (long long)__builtin_amdgcn_s_getreg(4 | (31 << 11));            \
                r_[3] = (long long)__builtin_amdgcn_s_getreg(20 | (31 << 11));           \
                r_[4] = prof_w0; r_[5] = wall_clock64();                                 \
            }                                                                            \
        }                                                                                \
    } while (0)
#define FGW_PROF_DECL long long prof_t = wall_clock64(); const long long prof_w0 = prof_t, prof_c0 = clock64(); long long prof_acc[11] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}
#define FGW_PROF(k)                                                                      \
    do {                                                                                 \
        const long long t_ = wall_clock64();                                             \
        prof_acc[k] += t_ - prof_t;                                                      \
        prof_t = t_;                                                                     \
    } while (0)
#define FGW_PROF_FLUSH                                                                   \
    do {                                                                                 \
        if (threadIdx.x == 0) {                                                          \
            for (int k_ = 0; k_ < 11; ++k_) atomicAdd(reinterpret_cast<unsigned long long *>(&g_fgw_prof[k_]), (unsigned long long)prof_acc[k_]); \
            atomicAdd(reinterpret_cast<unsigned long long *>(&g_fgw_prof[20]), (unsigned long long)(clock64() - prof_c0));      \
            atomicAdd(reinterpret_cast<unsigned long long *>(&g_fgw_prof[21]), (unsigned long long)(wall_clock64() - prof_w0)); \
        }                                                                                \
    } while (0)
#elif defined(CONAN_FGW_MARK)      /* (static instruction counts per phase: comment markers in the assembly, tools only) */
#define FGW_PROF_DECL
#define FGW_PROF(k) asm volatile("; FGWMARK " #k)
#define FGW_PROF_FLUSH
#define FGW_PROF_TRACE(tag)
#else
#define FGW_PROF_DECL
#define FGW_PROF(k)
#define FGW_PROF_FLUSH
#define FGW_PROF_TRACE(tag)
#endif

constexpr int FGW_THREADS = 256;
constexpr int FGW_WAVES = FGW_THREADS / 64;

// Per-graph contributions to the barycenter update (T_s Z_s — N > 64 only: the N <= 64 update kernel forms it from T itself — and T_s C_s T_s^T,
// summed over s by k_fgw_update_parts).  fp64: they are the
// largest HBM streams of a solve (41 MB written + 41 MB read per outer iteration at cfg2), but as fp32 (measured: coupling launch
// 132.3 -> 130.8 us, update 16.3 -> 15.3 us) their rounding — 6e-8 relative per outer iteration — is amplified by the scheme like every
// other perturbation: tests/test_gpu_fgw.py::test_large_graphs_vs_oracle[64-2-16] moved from < 1e-4 to 1.6e-4 of the fp64 reference on
// T.  Not worth 1.5 % of a solve.
typedef double fgw_part_t;

// Row pitch (elements) of the N x N matrices in LDS / scratch: odd, so that walks down a column of fp64 rows are conflict-free.  (The smallest
// P >= N with P = 2 (mod 4) — conflict-free fp64 MFMA A-operand reads instead — measured 0.806 ms against 0.719 ms per cfg2 solve.)
__host__ __device__ inline int fgw_pitch(int N) { return N | 1; }
struct FgwDims {
    int B, K, N, d, P;      // P = row pitch of the LDS/scratch matrices (odd => conflict-free column access)
};

// "Molecule b still moves" flags of the outer loop (barycenter.py:112): [parity of the outer iteration][features | structure][B] ints.  The update
// kernel of iteration o writes the flags of parity o & 1 — one from each of the molecule's two workgroups — and everything that runs in iteration
// o (coupling kernels, the update itself) reads the flags of iteration o - 1, i.e. parity (o + 1) & 1; the init kernels raise parity 1.
__device__ __forceinline__ bool fgw_active(const int *__restrict__ active, int B, int b, int outer) {
    const int *a = active + ((outer + 1) & 1) * 2 * B;
    return (a[b] | a[B + b]) != 0;
}
__device__ __forceinline__ void fgw_active_init(int *__restrict__ active, int B, int b) { active[2 * B + b] = 1; active[3 * B + b] = 1; }

// The input graphs' structure straight from the ragged neighbour lists ("never materialise [G, N_max, N_max]"): CSR by target
// over the whole batch + the target of every edge, as the radius-graph kernel leaves them (graph.hip).  Graph g owns the nodes gptr[g] ..
// gptr[g + 1] - 1 and the edges rowptr[lo] .. rowptr[lo + n] - 1 (n clamped to N like k_densify).  rowptr == nullptr: dense Cs [B,K,N,N].
// `dense`: [B,K,N,N] fp32 scratch that only the exact second pass (and kernels without a ragged load stage) fill, each workgroup its own slice.
struct FgwAdj {
    const int *gptr, *rowptr, *col, *tgt;
    float *dense;
    int *order;              // [B] (nullable): molecules by descending number of real nodes — k_fgw_small_vectors / k_fgw_init fill it (fgw_order_by_size), k_fgw_coupling_fast and
                             // k_fgw_coupling_big deal their workgroups over the XCDs / CUs in that order (a coupling's work grows with its real nodes: padded nodes are solved as one)
};
// to_dense_adj (schnet_no_sum.py:249; orientation of k_densify: entry [source][target] += 1) of graph g into a ZEROED LDS byte matrix with row
// pitch P (4-byte aligned): one thread per edge, counts packed four to a word (an entry above 255 would carry: a radius graph has no repeated
// pair at all).  The caller's barriers bracket it.
template <int NT>
__device__ __forceinline__ void adj_scatter_lds_bytes(const FgwAdj &A, int g, int N, int P, unsigned char *m, int tid) {
    const int lo = A.gptr[g], n = min(A.gptr[g + 1] - lo, N);
    const int e0 = A.rowptr[lo], e1 = A.rowptr[lo + n];
    for (int e = e0 + tid; e < e1; e += NT) {
        const int i = A.tgt[e] - lo, j = A.col[e] - lo;
        if (i >= 0 && i < N && j >= 0 && j < N) {
            const unsigned o = (unsigned)(j * P + i);
            atomicAdd(reinterpret_cast<unsigned *>(m + (o & ~3u)), 1u << (8 * (o & 3u)));
        }
    }
}
// the same into this workgroup's slice of the dense scratch (global memory; exact second pass only): returns the slice.  Contains barriers.
template <int NT>
__device__ __forceinline__ const float *adj_dense_slice(const FgwAdj &A, int g, int N, int tid) {
    float *dst = A.dense + (size_t)g * N * N;
    for (int t = tid; t < N * N; t += NT) dst[t] = 0.f;
    __syncthreads();
    const int lo = A.gptr[g], n = min(A.gptr[g + 1] - lo, N);
    const int e0 = A.rowptr[lo], e1 = A.rowptr[lo + n];
    for (int e = e0 + tid; e < e1; e += NT) {
        const int i = A.tgt[e] - lo, j = A.col[e] - lo;
        if (i >= 0 && i < N && j >= 0 && j < N) atomicAdd(&dst[j * N + i], 1.0f);
    }
    __threadfence_block();
    __syncthreads();
    return dst;
}

// Uniform fp64 constants of the round-3 coupling kernels, formed on the host: the scalar unit has no fp64 conversions, so the same
// values derived in the kernel from the fp32 parameters live in VECTOR registers for the whole kernel (the register budgets of
// those kernels have no room for them).
struct FastConst {
    double two_alpha, one_m_alpha, four_alpha_inv_eps, inv_eps, inner_tol, stop_thr, inv_n;
};
inline FastConst fast_const(const conan_fgw_params &prm, int N) {
    const double alpha = (double)prm.alpha, inv_eps = 1.0 / (double)prm.epsilon;
    return FastConst{2.0 * alpha, 1.0 - alpha, 4.0 * alpha * inv_eps, inv_eps, (double)prm.inner_tol, (double)prm.stop_thr, 1.0 / (double)N};
}

__device__ __forceinline__ double exp_acc(double x) {
    // exp(x) = 2^(x*log2e); integer part applied with ldexp, fractional part on the fp32 transcendental unit.
    if (x < -745.0) return 0.0;
    const double t = x * 1.4426950408889634074;
    const double n = rint(t);
    const float f = (float)(t - n);
    const float e = __builtin_amdgcn_exp2f(f);
    return ldexp((double)e, (int)n);
}

// Branch-free exp(x) for |x| <= 700 (the scaling form's K = exp(Mr - ref)): 2^n from integer arithmetic on the exponent field,
// the fraction on v_exp_f32; n and the rounded t come from the 1.5 * 2^52 shift trick (two fp64 adds instead of v_rndne_f64 +
// v_cvt_i32_f64 + v_ldexp_f64).  Arguments below -700 give 0 (also covers the -1e300 padding mask), above +700 give +inf
// (the caller's range check on the sums then takes the exact path).  Relative error ~1e-7 (the fp32 exp2 of the fraction),
// the same as exp_acc.
__device__ __forceinline__ double exp_fast(double x) {
    const double xc = fmin(fmax(x, -700.0), 700.0);
    const double t = xc * 1.4426950408889634074;
    const double sh = t + 6755399441055744.0;                        // 1.5 * 2^52: the low word now holds rint(t) as an int
    const int n = __double2loint(sh);
    const double f = t - (sh - 6755399441055744.0);                  // in [-0.5, 0.5]
    const double e = (double)__builtin_amdgcn_exp2f((float)f);
    const double scale = __hiloint2double((n + 1023) << 20, 0);      // 2^n, |n| <= 1010
    const double r = e * scale;
    return x > -700.0 ? (x < 700.0 ? r : __builtin_inf()) : 0.0;
}

// 1 / s for a positive normal s: v_rcp_f64 seed (~2^-27 relative... refined by two Newton steps to ~1 ulp); no division fix-up
// sequence (the operands here are sums in [1e-150, 1e150], checked by the caller).
__device__ __forceinline__ double rcp_pos(double s) {
    double r = __builtin_amdgcn_rcp(s);
    r = fma(fma(-s, r, 1.0), r, r);
    r = fma(fma(-s, r, 1.0), r, r);
    return r;
}

// log(s) for s in [1, 2^20]: v_log_f32 seed (1 ulp of fp32) refined by one Newton step y <- y + (s*exp(-y) - 1) - r^2/2,
// which squares the relative error: ~1e-14.  Replaces the ~100-instruction ocml fp64 log in the Sinkhorn loops.
__device__ __forceinline__ double log_acc(double s) {
    const double y0 = (double)(__builtin_amdgcn_logf((float)s) * 0.693147180559945309f);
    const double r = s * exp_acc(-y0) - 1.0;
    return y0 + (r - 0.5 * r * r);
}

__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_max_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { double w = __shfl_xor(v, o, 64); v = w > v ? w : v; }
    return v;
}

// block-wide sum of one double per thread; result broadcast to every thread. red[] has FGW_WAVES+1 doubles.
template <int NW = FGW_WAVES>
__device__ __forceinline__ double block_sum_d(double v, double *red) {
    v = wave_sum_d(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    double s = 0.0;
#pragma unroll
    for (int w = 0; w < NW; ++w) s += red[w];
    return s;
}


typedef double f64x4 __attribute__((ext_vector_type(4)));

// exp(x) for x <= 0 in the log-sum-exp loops: x is a difference formed in fp64, so converting it to fp32 keeps a relative
// precision of 2^-24 in the exponent; terms below e^-80 cannot change an fp64 sum whose largest term is 1.
__device__ __forceinline__ double exp_lse(double x) {
    const float f = (float)x;
    return (double)__builtin_amdgcn_exp2f(f * 1.44269504088896340736f);
}

// D[M x Nn] = X[M x Kd] @ W[Kd x Nn] with fp64 MFMA (v_mfma_f64_16x16x4_f64).  The problem is covered by ceil(M/16) x
// ceil(Nn/16) tiles; out-of-range operand elements are read as 0 and out-of-range results are not stored, so ragged
// sizes (N = 33) need no separate border code (a VALU border loop serialises on the one wavefront that owns it).
// X(i,k), W(k,j) are element readers, st(i,j,v) the writer.  Fragment layout: A lane l -> X[i0 + (l&15)][k0 + (l>>4)],
// B lane l -> W[k0 + (l>>4)][j0 + (l&15)], D reg q lane l -> row i0 + (l>>4) + 4q, col j0 + (l&15).
// Workgroup-collective (tiles are dealt round-robin to the wavefronts); no barrier inside.
template <int NW = FGW_WAVES, class FX, class FW, class FS>
__device__ __forceinline__ void mm_f64_pad(int M, int Nn, int Kd, FX X, FW W, FS st) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int Mq = (M + 15) >> 4, Nq = (Nn + 15) >> 4;
    const int li = lane & 15, lk = lane >> 4;
    for (int t = wave; t < Mq * Nq; t += NW) {
        const int i0 = (t / Nq) << 4, j0 = (t % Nq) << 4;
        const int ia = i0 + li, jb = j0 + li;
        const bool ra = ia < M, cb = jb < Nn;
        const int ic = ra ? ia : M - 1, jc = cb ? jb : Nn - 1;          // clamped: loads stay in range, values masked
        f64x4 acc = {0.0, 0.0, 0.0, 0.0};
        int k0 = 0;
        for (; k0 + 16 <= Kd; k0 += 16) {
            double a[4], b[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { a[u] = X(ic, k0 + 4 * u + lk); b[u] = W(k0 + 4 * u + lk, jc); }
#pragma unroll
            for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(ra ? a[u] : 0.0, cb ? b[u] : 0.0, acc, 0, 0, 0);
        }
        if (k0 < Kd) {                                    // up to 4 remaining k-steps in ONE trip, clamped loads with zeroed operands
            double a[4], b[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int k = k0 + 4 * u + lk, kc = k < Kd ? k : Kd - 1;
                a[u] = X(ic, kc); b[u] = W(kc, jc);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const bool kin = k0 + 4 * u + lk < Kd;
                if (k0 + 4 * u < Kd)                       // workgroup-uniform
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64((ra && kin) ? a[u] : 0.0, (cb && kin) ? b[u] : 0.0, acc, 0, 0, 0);
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int i = i0 + lk + 4 * q;
            if (i < M && cb) st(i, jb, acc[q]);
        }
    }
}

// Padded-tile product for row-major operands in GLOBAL memory (the large-N kernel: L2-resident matrices).  X(i,k) = X[i*pX + k];
// W(k,j) = W[k*pW + j] (WT = false) or W[j*pW + k] (WT = true).  Inside a 16-wide k-trip the MFMA step u takes k = k0 + 4*lk + u from
// lane group lk (instead of k0 + 4*u + lk): every lane then owns FOUR CONSECUTIVE k of its row, i.e. one contiguous 32-byte
// (fp64) / 16-byte (fp32) read per operand and trip instead of four scattered 8-byte ones.  The reader-lambda form issued
// 8 wave-loads per trip that each touched 16 rows, and the products were bound by that.  Both operands use
// the same permutation, so each k is still multiplied exactly once; the summation order inside a trip differs from mm_f64_pad.
template <int NW, bool WT, typename TX, typename TW, class FS>
__device__ __forceinline__ void mm_f64_glb(int M, int Nn, int Kd, const TX *__restrict__ X, int pX, const TW *__restrict__ W, int pW, FS st,
                                           const int tid = threadIdx.x) {
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int Mq = (M + 15) >> 4, Nq = (Nn + 15) >> 4;
    const int li = lane & 15, lk = lane >> 4;
    for (int t = wave; t < Mq * Nq; t += NW) {
        const int i0 = (t / Nq) << 4, j0 = (t % Nq) << 4;
        const int ia = i0 + li, jb = j0 + li;
        const bool ra = ia < M, cb = jb < Nn;
        const TX *xr = X + (size_t)(ra ? ia : M - 1) * pX;             // clamped rows: loads stay in range, values masked
        const int jc = cb ? jb : Nn - 1;
        f64x4 acc = {0.0, 0.0, 0.0, 0.0}, acc2 = {0.0, 0.0, 0.0, 0.0};
        for (int k0 = 0; k0 < Kd; k0 += 16) {
            const int kb = k0 + 4 * lk;                                 // this lane's four consecutive k
            double a[4], b[4];
            if (k0 + 16 <= Kd) {
#pragma unroll
                for (int u = 0; u < 4; ++u) a[u] = (double)xr[kb + u];
                if (WT) {
                    const TW *wr = W + (size_t)jc * pW + kb;
#pragma unroll
                    for (int u = 0; u < 4; ++u) b[u] = (double)wr[u];
                } else {
#pragma unroll
                    for (int u = 0; u < 4; ++u) b[u] = (double)W[(size_t)(kb + u) * pW + jc];
                }
            } else {                                                    // ragged last trip: clamp, then zero
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int k = kb + u, kc = k < Kd ? k : Kd - 1;
                    const double av = (double)xr[kc], bv = (double)(WT ? W[(size_t)jc * pW + kc] : W[(size_t)kc * pW + jc]);
                    a[u] = k < Kd ? av : 0.0; b[u] = k < Kd ? bv : 0.0;
                }
            }
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(ra ? a[0] : 0.0, cb ? b[0] : 0.0, acc, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(ra ? a[1] : 0.0, cb ? b[1] : 0.0, acc2, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(ra ? a[2] : 0.0, cb ? b[2] : 0.0, acc, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(ra ? a[3] : 0.0, cb ? b[3] : 0.0, acc2, 0, 0, 0);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int i = i0 + lk + 4 * q;
            if (i < M && cb) st(i, jb, acc[q] + acc2[q]);
        }
    }
}

// FgwAdj.order: the molecules by descending number of real nodes (ties in index order: a stable counting sort, the same permutation on every
// run).  One workgroup of NT >= 256 threads, B <= 4096 (the launcher's bound); sizes above 255 share the last bin.
template <int NT>
__device__ __forceinline__ void fgw_order_by_size(const FgwAdj &adj, int B, int K, int tid) {
    static_assert(NT >= 256, "one thread per size bin");
    __shared__ unsigned char nsz[4096];
    __shared__ int hist[256], base[256];
    if (B > 4096) return;                                                   // (the launcher does not hand out an order buffer beyond that; workgroup-uniform)
    for (int t = tid; t < 256; t += NT) hist[t] = 0;
    __syncthreads();
    for (int m = tid; m < B; m += NT) {
        const int n = min(max(adj.gptr[m * K + 1] - adj.gptr[m * K], 0), 255);
        nsz[m] = (unsigned char)n;
        atomicAdd(&hist[n], 1);
    }
    __syncthreads();
    if (tid == 0) {
        int run = 0;
        for (int v = 255; v >= 0; --v) { base[v] = run; run += hist[v]; }
    }
    __syncthreads();
    if (tid < 256 && hist[tid] > 0) {
        int pos = base[tid];
        for (int m = 0; m < B; ++m)
            if (nsz[m] == (unsigned char)tid) adj.order[pos++] = m;
    }
}

// Register-blocked form of mm_f64_glb: a wavefront owns a 2 x 2 block of 16 x 16 tiles and feeds its four MFMAs per k-step from
// TWO A fragments and TWO B fragments — one operand load per MFMA instead of two.  The large-N coupling kernel is bound by the
// operand traffic of its products (every fragment comes from L2 or LDS; three workgroups per CU share the path), not by the matrix
// pipe alone.  Work items: the 2 x 2 blocks that fill whole rounds of the NW wavefronts first, then the remaining tiles one by
// one (so that no wavefront ends up with a whole block more than the others).  Same lane <-> k permutation as mm_f64_glb
// (four consecutive k per lane and trip); one accumulator per tile (four independent chains per wavefront).
#ifndef CONAN_FGW_KT
#define CONAN_FGW_KT 2      // (A/B switch: 1 and 4 measured in round 5, profiles/r5_ab_fgw_large_kt.txt)
#endif
template <int NW, bool WT, typename TX, typename TW, class FS, int KT = CONAN_FGW_KT>      // KT consecutive k per lane and trip (a trip = 4 KT k)
__device__ __forceinline__ void mm_f64_glb22(int M, int Nn, int Kd, const TX *__restrict__ X, int pX, const TW *__restrict__ W, int pW, FS st,
                                             const int tid = threadIdx.x) {
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int Mq = (M + 15) >> 4, Nq = (Nn + 15) >> 4;
    const int li = lane & 15, lk = lane >> 4;
    const int Mb = Mq >> 1, Nb = Nq >> 1;                             // 2 x 2 blocks over the even part
    const int nblk = Mb * Nb, nblk_used = (nblk / NW) * NW;          // blocks that fill whole rounds
    auto load_a = [&](const TX *xr, int kb, int k0, double (&a)[KT]) {
        if (k0 + 4 * KT <= Kd) {
#pragma unroll
            for (int u = 0; u < KT; ++u) a[u] = (double)xr[kb + u];
        } else {
#pragma unroll
            for (int u = 0; u < KT; ++u) { const int k = kb + u; const double v = (double)xr[k < Kd ? k : Kd - 1]; a[u] = k < Kd ? v : 0.0; }
        }
    };
    auto load_b = [&](int jc, int kb, int k0, double (&b)[KT]) {
        if (k0 + 4 * KT <= Kd) {
#pragma unroll
            for (int u = 0; u < KT; ++u) b[u] = (double)(WT ? W[(size_t)jc * pW + kb + u] : W[(size_t)(kb + u) * pW + jc]);
        } else {
#pragma unroll
            for (int u = 0; u < KT; ++u) {
                const int k = kb + u, kc = k < Kd ? k : Kd - 1;
                const double v = (double)(WT ? W[(size_t)jc * pW + kc] : W[(size_t)kc * pW + jc]);
                b[u] = k < Kd ? v : 0.0;
            }
        }
    };
    auto store_tile = [&](int i0, int j0, const f64x4 &r) {
        const int jb = j0 + li;
        if (jb < Nn) {
#pragma unroll
            for (int q = 0; q < 4; ++q) { const int i = i0 + lk + 4 * q; if (i < M) st(i, jb, r[q]); }
        }
    };
    for (int t = wave; t < nblk_used; t += NW) {
        const int i0 = (t / Nb) << 5, j0 = (t % Nb) << 5;
        const TX *x0 = X + (size_t)min(i0 + li, M - 1) * pX, *x1 = X + (size_t)min(i0 + 16 + li, M - 1) * pX;     // clamped rows: outputs beyond M are not stored
        const int jc0 = min(j0 + li, Nn - 1), jc1 = min(j0 + 16 + li, Nn - 1);
        f64x4 c00 = {0.0, 0.0, 0.0, 0.0}, c01 = c00, c10 = c00, c11 = c00;
        for (int k0 = 0; k0 < Kd; k0 += 4 * KT) {
            const int kb = k0 + KT * lk;
            double a0[KT], a1[KT], b0[KT], b1[KT];
            load_a(x0, kb, k0, a0); load_a(x1, kb, k0, a1); load_b(jc0, kb, k0, b0); load_b(jc1, kb, k0, b1);
#pragma unroll
            for (int u = 0; u < KT; ++u) {
                c00 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[u], b0[u], c00, 0, 0, 0);
                c01 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[u], b1[u], c01, 0, 0, 0);
                c10 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[u], b0[u], c10, 0, 0, 0);
                c11 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[u], b1[u], c11, 0, 0, 0);
            }
        }
        store_tile(i0, j0, c00); store_tile(i0, j0 + 16, c01); store_tile(i0 + 16, j0, c10); store_tile(i0 + 16, j0 + 16, c11);
    }
    // the remaining tiles, one per item: tiles of the unused blocks first, then the odd last row / column of tiles
    const int rest_blk = nblk - nblk_used;
    const int n_single = rest_blk * 4 + (Mq * Nq - 4 * nblk);
    for (int t = wave; t < n_single; t += NW) {
        int ti, tj;
        if (t < rest_blk * 4) { const int bq = nblk_used + (t >> 2); ti = 2 * (bq / Nb) + ((t >> 1) & 1); tj = 2 * (bq % Nb) + (t & 1); }
        else {
            int q = t - rest_blk * 4;                                  // strips: odd last column of tiles (all rows), then odd last row (even columns part)
            const int ncol = (Nq & 1) ? Mq : 0;
            if (q < ncol) { ti = q; tj = Nq - 1; }
            else { q -= ncol; ti = Mq - 1; tj = q; }
        }
        const int i0 = ti << 4, j0 = tj << 4;
        const TX *xr = X + (size_t)min(i0 + li, M - 1) * pX;
        const int jc = min(j0 + li, Nn - 1);
        f64x4 acc = {0.0, 0.0, 0.0, 0.0}, acc2 = {0.0, 0.0, 0.0, 0.0};
        for (int k0 = 0; k0 < Kd; k0 += 4 * KT) {
            const int kb = k0 + KT * lk;
            double a[KT], bq[KT];
            load_a(xr, kb, k0, a); load_b(jc, kb, k0, bq);
#pragma unroll
            for (int u = 0; u < KT; ++u) {
                if (u & 1) acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u], bq[u], acc2, 0, 0, 0);
                else acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u], bq[u], acc, 0, 0, 0);
            }
        }
        store_tile(i0, j0, acc + acc2);
    }
}

// ---- G = X Wb^T on the INTEGER matrix pipe, exact (round 5) ---------------------------------------------------------------------------------
// X fp64 (the large-N coupling kernel's A = C1 T), Wb small integers <= 127 (the input graph's adjacency bytes, in LDS).  The consumer of G needs an
// ABSOLUTE accuracy — G enters exp(4 alpha G / eps + ...), stored as fp32 — so X is cut ONCE per element (fgw_digits_from_f64: a pass of N^2 / 512
// elements per thread after the product that forms X, whose epilogue collects max |X|) into a 32-bit fixed-point number q against a block-wide
// power of two, and q into four balanced base-256 digits in [-128, 127]: (q + 0x00808080) ^ 0x00808080 holds them as its four signed bytes, two 4 x 4
// byte transposes turn 8 consecutive elements of a row into the four 8-byte digit planes = the A operands of v_mfma_i32_16x16x32_i8, stored as one
// 32-byte record (plane p at byte 8 p).  The product then runs on int32 accumulators — no rounding anywhere — with ONE 32-byte load per lane and
// k-step for X (half the bytes of the fp64 operand) and no vector arithmetic on it:  G = 2^(xexp - 31) (2^24 S3 + 2^16 S2 + 2^8 S1 + S0), off the exact
// product by the fixed-point cut alone (<= 2^-30 of max |X| per unit of Wb).  A first form that cut the digits inside the product loop (every element
// three times, ~40 or ~15 vector instructions each on bf16 / int8 digits) measured 248 / 136 us per launch against the fp64 product's 131
// (profiles/r5_ab_fgw_adj_i8.txt): the phase is bound by vector issue and trips, not by MFMA cycles on either pipe.
// Lane layout of the 16x16x32 form: A lane (row = lane & 15, k = 8 (lane >> 4) .. + 7 = the 8 bytes of the operand), B lane (column = lane & 15, same
// k), D lane (column = lane & 15, rows 4 (lane >> 4) + r).  Digit records: row i, block kb (k = 8 kb .. + 7) at D + (i * KB + kb) * 32 bytes, KB =
// 4 ceil(Kd / 32) blocks per row (blocks beyond Kd hold zeros).
typedef int fgw_i32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void fgw_digit_planes4(const double (&v)[4], double scale, unsigned (&pl)[4]) {
    unsigned r[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) r[u] = ((unsigned)__double2int_rn(v[u] * scale) + 0x00808080u) ^ 0x00808080u;      // bytes = balanced digits, least significant first
    // 4 x 4 byte transpose: pl[p] = {r0.byte p, r1.byte p, r2.byte p, r3.byte p}   (v_perm_b32: selector 0-3 = bytes of the SECOND source, 4-7 = of the first)
    const unsigned t0 = __builtin_amdgcn_perm(r[1], r[0], 0x05010400u), t1 = __builtin_amdgcn_perm(r[1], r[0], 0x07030602u);
    const unsigned t2 = __builtin_amdgcn_perm(r[3], r[2], 0x05010400u), t3 = __builtin_amdgcn_perm(r[3], r[2], 0x07030602u);
    pl[0] = __builtin_amdgcn_perm(t2, t0, 0x05040100u); pl[1] = __builtin_amdgcn_perm(t2, t0, 0x07060302u);
    pl[2] = __builtin_amdgcn_perm(t3, t1, 0x05040100u); pl[3] = __builtin_amdgcn_perm(t3, t1, 0x07060302u);
}
// X [M x Kd] (pitch pX) -> digit records; |X| < 2^(xexp - 1) everywhere.  All NT threads; the caller's barriers order it against producer and consumer.
template <int NT>
__device__ __forceinline__ void fgw_digits_from_f64(int M, int Kd, const double *__restrict__ X, int pX, int xexp, uint4 *__restrict__ D, const int tid = threadIdx.x) {
    const int KB = ((Kd + 31) >> 5) << 2;
    const double scale = __longlong_as_double((long long)(1023 + 31 - xexp) << 52);       // 2^(31 - xexp)
    constexpr int R = 2;                                                  // records per thread and round: their 16 loads are in flight together (X was just written: L2)
    for (int base = 0; base < M * KB; base += R * NT) {
        double v[R][8];
#pragma unroll
        for (int q = 0; q < R; ++q) {
            const int idx = base + q * NT + tid;
            const int ic = idx < M * KB ? idx : M * KB - 1;
            const int i = ic / KB, kb = ic - i * KB;
            const double *xr = X + (size_t)i * pX;
#pragma unroll
            for (int u = 0; u < 8; ++u) { const int k = 8 * kb + u; v[q][u] = xr[k < Kd ? k : Kd - 1]; }
        }
#pragma unroll
        for (int q = 0; q < R; ++q) {
            const int idx = base + q * NT + tid;
            if (idx >= M * KB) continue;
            const int kb = idx % KB;
            double v0[4], v1[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { v0[u] = 8 * kb + u < Kd ? v[q][u] : 0.0; v1[u] = 8 * kb + 4 + u < Kd ? v[q][4 + u] : 0.0; }
            unsigned lo[4], hi[4];
            fgw_digit_planes4(v0, scale, lo); fgw_digit_planes4(v1, scale, hi);
            D[2 * idx] = make_uint4(lo[0], hi[0], lo[1], hi[1]);
            D[2 * idx + 1] = make_uint4(lo[2], hi[2], lo[3], hi[3]);
        }
    }
}
__device__ __forceinline__ long fgw_bytes8(const unsigned char *__restrict__ w, int kb, int Kd) {
    unsigned lo = 0, hi = 0;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int k0 = kb + u, k1 = kb + 4 + u;
        lo |= (unsigned)w[k0 < Kd ? k0 : Kd - 1] << (8 * u);                   // (k >= Kd: the digits there are 0, any byte will do)
        hi |= (unsigned)w[k1 < Kd ? k1 : Kd - 1] << (8 * u);
    }
    return (long)(((unsigned long long)hi << 32) | lo);
}
template <int NW, class FS>
__device__ __forceinline__ void mm_adj_i8(int M, int Nn, int Kd, const uint4 *__restrict__ D, const unsigned char *__restrict__ Wb, int pW, int xexp, FS st,
                                          const int tid = threadIdx.x) {
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lg = lane >> 4;
    const int Mq = (M + 15) >> 4, Nh = (Nn + 31) >> 5, KB = ((Kd + 31) >> 5) << 2;
    const double unscale = __longlong_as_double((long long)(1023 + xexp - 31) << 52);      // 2^(xexp - 31)
    for (int t = wave; t < Mq * Nh; t += NW) {
        const int i0 = (t / Nh) << 4, j0 = (t % Nh) << 5;
        const uint4 *dr = D + (size_t)min(i0 + li, M - 1) * KB * 2;
        const unsigned char *w0 = Wb + (size_t)min(j0 + li, Nn - 1) * pW, *w1 = Wb + (size_t)min(j0 + 16 + li, Nn - 1) * pW;
        fgw_i32x4 acc[2][4];
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int d = 0; d < 4; ++d) acc[c][d] = fgw_i32x4{0, 0, 0, 0};
        for (int k0 = 0; k0 < Kd; k0 += 128) {                          // chunks of four k-steps: their records are in flight together (one L2 round trip per chunk)
            uint4 da[4], db[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int kb8 = min(((k0 + 32 * q) >> 3) + lg, KB - 1);
                da[q] = dr[2 * kb8]; db[q] = dr[2 * kb8 + 1];
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (k0 + 32 * q >= Kd) break;                           // (uniform)
                const int kb8 = ((k0 + 32 * q) >> 3) + lg;
                const long b0 = fgw_bytes8(w0, 8 * kb8, Kd), b1 = fgw_bytes8(w1, 8 * kb8, Kd);
                const long a[4] = {(long)(((unsigned long long)da[q].y << 32) | da[q].x), (long)(((unsigned long long)da[q].w << 32) | da[q].z),
                                   (long)(((unsigned long long)db[q].y << 32) | db[q].x), (long)(((unsigned long long)db[q].w << 32) | db[q].z)};
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    acc[0][d] = __builtin_amdgcn_mfma_i32_16x16x32_i8(a[d], b0, acc[0][d], 0, 0, 0);
                    acc[1][d] = __builtin_amdgcn_mfma_i32_16x16x32_i8(a[d], b1, acc[1][d], 0, 0, 0);
                }
            }
        }
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int jb = j0 + 16 * c + li;
            if (jb >= Nn) continue;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = i0 + 4 * lg + r;
                if (i >= M) continue;
                const double g = (((double)acc[c][3][r] * 256.0 + (double)acc[c][2][r]) * 256.0 + (double)acc[c][1][r]) * 256.0 + (double)acc[c][0][r];
                st(i, jb, g * unscale);
            }
        }
    }
}
// block-wide maximum of one non-negative double per thread; red[] as block_sum_d
template <int NW = FGW_WAVES>
__device__ __forceinline__ double block_max_d(double v, double *red) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const double w = __shfl_xor(v, o, 64); v = w > v ? w : v; }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    double m = red[0];
#pragma unroll
    for (int w = 1; w < NW; ++w) m = red[w] > m ? red[w] : m;
    return m;
}
// The same on the BIT PATTERNS of non-negative doubles (sign bit cleared by the caller): integer order is numeric order there, and an infinity or
// a NaN sorts above every finite value — a floating-point maximum drops NaN (every comparison with it is false), this one carries it to the caller.
template <int NW>
__device__ __forceinline__ unsigned long long block_max_bits(unsigned long long v, double *red) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const unsigned long long w = __shfl_xor(v, o, 64); v = w > v ? w : v; }
    unsigned long long *r = reinterpret_cast<unsigned long long *>(red);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) r[threadIdx.x >> 6] = v;
    __syncthreads();
    unsigned long long m = r[0];
#pragma unroll
    for (int w = 1; w < NW; ++w) m = r[w] > m ? r[w] : m;
    return m;
}

// Variant for thin ragged borders: MFMA on the 16-aligned core, plain FMA loops for the few border rows/columns.
// The border output owned by a thread (two passes of 64 outputs at most) depends only on (M, Nn): it is computed once per
// kernel (two integer divisions per pass) and reused by every product of that shape.
struct BorderIdx {
    int i[2], j[2], count;
    bool on[2];
};
template <int NW = FGW_WAVES>
__device__ __forceinline__ BorderIdx border_prepare(int M, int Nn, const int tid = threadIdx.x) {
    BorderIdx bi;
    const int Mc = (M >> 4) << 4, Nc = (Nn >> 4) << 4;
    const int nb1 = (M - Mc) * Nn, nb2 = Mc * (Nn - Nc);
    bi.count = nb1 + nb2;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const int t = p * (NW * 16) + (tid >> 2);
        bi.on[p] = t < bi.count;
        bi.i[p] = 0; bi.j[p] = 0;
        if (bi.on[p]) {
            if (t < nb1) { bi.i[p] = Mc + t / Nn; bi.j[p] = t % Nn; }
            else { const int q = t - nb1; bi.i[p] = q / (Nn - Nc); bi.j[p] = Nc + q % (Nn - Nc); }
        }
    }
    return bi;
}
template <int NW = FGW_WAVES>
__device__ __forceinline__ bool border_path(int M, int Nn) {       // dispatch rule of mm_f64 (workgroup-uniform)
    const int Mc = (M >> 4) << 4, Nc = (Nn >> 4) << 4;
    return (M - Mc) * Nn + Mc * (Nn - Nc) <= NW * 32 && Mc > 0 && Nc > 0;
}

template <int NW = FGW_WAVES, class FX, class FW, class FS>
__device__ __forceinline__ void mm_f64_border(int M, int Nn, int Kd, FX X, FW W, FS st, const BorderIdx &bi) {
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int Mq = M >> 4, Nq = Nn >> 4;
    const int li = lane & 15, lk = lane >> 4;
    for (int t = wave; t < Mq * Nq; t += NW) {
        const int i0 = (t / Nq) << 4, j0 = (t % Nq) << 4;
        f64x4 acc = {0.0, 0.0, 0.0, 0.0};
        int k0 = 0;
        for (; k0 + 16 <= Kd; k0 += 16) {
            double a[4], b[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { a[u] = X(i0 + li, k0 + 4 * u + lk); b[u] = W(k0 + 4 * u + lk, j0 + li); }
#pragma unroll
            for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u], b[u], acc, 0, 0, 0);
        }
        for (; k0 < Kd; k0 += 4) {
            const int k = k0 + lk;
            const double a = k < Kd ? X(i0 + li, k) : 0.0;
            const double b = k < Kd ? W(k, j0 + li) : 0.0;
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) st(i0 + lk + 4 * q, j0 + li, acc[q]);
    }
    // Border outputs: 4 lanes per output, each summing every 4th k, combined with two xor-shuffles.  This spreads the
    // (few) border dot products over all wavefronts: with one thread per output the whole border lands on wavefront 0,
    // which then holds every barrier of the caller (PMC: 58 % of the wave cycles were spent waiting).
    const int sub = tid & 3;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        if (p * (NW * 16) >= bi.count) break;                 // workgroup-uniform
        double a = 0.0;
        if (bi.on[p]) {
            // operands of 8 k-steps are requested together, then multiplied: the serial form of this loop is one LDS (or L2)
            // round trip per step, and the border then costs more than the MFMA core it completes
            int k = sub;
            for (; k + 28 < Kd; k += 32) {
                double xa[8], wb[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) { xa[u] = X(bi.i[p], k + 4 * u); wb[u] = W(k + 4 * u, bi.j[p]); }
#pragma unroll
                for (int u = 0; u < 8; ++u) a += xa[u] * wb[u];
            }
            if (k < Kd) {                                   // tail: up to 8 steps, clamped loads with zeroed products
                double xa[8], wb[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) { const int kk = k + 4 * u < Kd ? k + 4 * u : Kd - 1; xa[u] = X(bi.i[p], kk); wb[u] = W(kk, bi.j[p]); }
#pragma unroll
                for (int u = 0; u < 8; ++u) a += (k + 4 * u < Kd) ? xa[u] * wb[u] : 0.0;
            }
        }
        a += __shfl_xor(a, 1, 64);
        a += __shfl_xor(a, 2, 64);
        if (bi.on[p] && sub == 0) st(bi.i[p], bi.j[p], a);
    }
}

// Dispatch (workgroup-uniform): a border of at most one pass of the 256 threads (e.g. N = 33: 65 outputs) is cheaper on
// the FMA path than the extra mostly-empty tiles; anything thicker goes to the padded-tile path.
template <int NW = FGW_WAVES, class FX, class FW, class FS>
__device__ __forceinline__ void mm_f64(int M, int Nn, int Kd, FX X, FW W, FS st) {
    if (border_path<NW>(M, Nn)) mm_f64_border<NW>(M, Nn, Kd, X, W, st, border_prepare<NW>(M, Nn));
    else mm_f64_pad<NW>(M, Nn, Kd, X, W, st);
}
// same with the border ownership prepared by the caller (products repeated inside a loop)
template <class FX, class FW, class FS>
__device__ __forceinline__ void mm_f64(int M, int Nn, int Kd, FX X, FW W, FS st, const BorderIdx &bi) {
    if (border_path(M, Nn)) mm_f64_border(M, Nn, Kd, X, W, st, bi);
    else mm_f64_pad(M, Nn, Kd, X, W, st);
}

// ---- Products whose operands live in LDS with a fixed pitch (the register-resident coupling kernel) -----------------------------
// D[M x Nn] = X[M x Kd] @ W[Kd x Nn]; X(i,k) = X[i*pX + k]; W(k,j) = W[k*pW + j] (WT = false) or W[j*pW + k] (WT = true); element types
// float or double.  Same tiling, fragment layout, summation order and border rule as mm_f64 above (results are bitwise equal); what
// changes is the address arithmetic: with element-reader lambdas the compiler re-derives i*pitch + k for every operand (a matrix
// product of 33^3 spent ~700 VALU instructions per wave around 9 MFMAs); here each lane owns two running pointers and the k loop
// is pointer + immediate offset.
template <bool WT, typename TW>
__device__ __forceinline__ double mm_w_at(const TW *W, int pW, int k, int j) { return (double)(WT ? W[j * pW + k] : W[k * pW + j]); }

template <int NW = FGW_WAVES, bool WT, typename TX, typename TW, class FS>
__device__ __forceinline__ void mm_lds(int M, int Nn, int Kd, const TX *__restrict__ X, int pX, const TW *__restrict__ W, int pW, FS st,
                                       const BorderIdx &bi, const int tid = threadIdx.x) {
    // `tid`: callers inside a long loop pass a copy of threadIdx.x laundered through an empty asm, so that the per-lane pointers and
    // tile indices below are re-derived per call instead of being hoisted out of the loop and kept (spilled) across it
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lk = lane >> 4;
    const bool border = border_path<NW>(M, Nn);
    const int Mq = border ? M >> 4 : (M + 15) >> 4, Nq = border ? Nn >> 4 : (Nn + 15) >> 4;
    const int kfull = Kd & ~3;                                  // k-steps with all four lk rows valid
    const int wstep = WT ? 4 : 4 * pW;                          // W pointer advance per k-step
    for (int t = wave; t < Mq * Nq; t += NW) {
        const int i0 = (t / Nq) << 4, j0 = (t % Nq) << 4;
        const int ia = i0 + li, jb = j0 + li;
        const bool ra = ia < M, cb = jb < Nn;                   // always true on the border path
        const TX *xp = X + (ra ? ia : M - 1) * pX + lk;
        const TW *wp = WT ? W + (cb ? jb : Nn - 1) * pW + lk : W + lk * pW + (cb ? jb : Nn - 1);
        // two accumulators (even / odd k-steps): the products of a tile form one dependent MFMA chain, and that chain — not the
        // matrix pipe — is what a 33-wide product waits for
        f64x4 acc = {0.0, 0.0, 0.0, 0.0}, acc2 = {0.0, 0.0, 0.0, 0.0};
        int k0 = 0;
        for (; k0 + 16 <= kfull; k0 += 16) {                    // 4 k-steps per trip: 8 operand reads in flight before the MFMAs
            double a[4], b[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { a[u] = (double)xp[4 * u]; b[u] = (double)wp[u * wstep]; }
            xp += 16; wp += 4 * wstep;
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(ra ? a[0] : 0.0, cb ? b[0] : 0.0, acc, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(ra ? a[1] : 0.0, cb ? b[1] : 0.0, acc2, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(ra ? a[2] : 0.0, cb ? b[2] : 0.0, acc, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(ra ? a[3] : 0.0, cb ? b[3] : 0.0, acc2, 0, 0, 0);
        }
        for (; k0 < kfull; k0 += 4) {
            const double a = (double)xp[0], b = (double)wp[0];
            xp += 4; wp += wstep;
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(ra ? a : 0.0, cb ? b : 0.0, acc, 0, 0, 0);
        }
        if (k0 < Kd) {                                          // ragged last step: rows lk >= Kd - k0 contribute zero
            const bool kin = k0 + lk < Kd;
            const int back = kin ? 0 : lk;                      // stay inside the matrix for the masked lanes
            const double a = (double)xp[-back], b = (double)(WT ? wp[-back] : wp[-back * pW]);
            acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64((ra && kin) ? a : 0.0, (cb && kin) ? b : 0.0, acc2, 0, 0, 0);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int i = i0 + lk + 4 * q;
            if (i < M && cb) st(i, jb, acc[q] + acc2[q]);
        }
    }
    if (!border) return;
    if (bi.count <= NW * 21) {
        // thin border (N = 33: 65 outputs): THREE lanes per output and a single pass — with four, the 65th output costs a whole second
        // pass; lane `sub3` sums k = sub3, sub3 + 3, ...  Triples are formed inside a wavefront (lanes 0..62: 21 outputs per wavefront).
        const int o = wave * 21 + lane / 3, sub3 = lane % 3;
        const bool on = lane < 63 && o < bi.count;
        const int Mc = (M >> 4) << 4, Nc = (Nn >> 4) << 4, nb1 = (M - Mc) * Nn;
        int bi_i = 0, bi_j = 0;
        if (on) {
            if (o < nb1) { bi_i = Mc + o / Nn; bi_j = o % Nn; }
            else { const int q = o - nb1; bi_i = q / (Nn - Nc); bi_j = Nc + q % (Nn - Nc); }
        }
        double acc = 0.0;
        if (on) {
            const int ws3 = WT ? 3 : 3 * pW;
            const TX *xp = X + bi_i * pX + sub3;
            const TW *wp = WT ? W + bi_j * pW + sub3 : W + sub3 * pW + bi_j;
            int k = sub3;
            for (; k + 9 < Kd; k += 12) {                       // 4 steps per trip (8 reads in flight), two partial sums
                double xa[4], wb[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) { xa[u] = (double)xp[3 * u]; wb[u] = (double)wp[u * ws3]; }
                xp += 12; wp += 4 * ws3;
                acc += (xa[0] * wb[0] + xa[2] * wb[2]) + (xa[1] * wb[1] + xa[3] * wb[3]);
            }
            if (k < Kd) {                                       // tail: up to 4 steps, clamped reads with zeroed products
                double xa[4], wb[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int uu = k + 3 * u < Kd ? u : 0;
                    xa[u] = (double)xp[3 * uu]; wb[u] = (double)wp[uu * ws3];
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) acc += (k + 3 * u < Kd) ? xa[u] * wb[u] : 0.0;
            }
        }
        const double a1 = __shfl(acc, lane + 1, 64), a2 = __shfl(acc, lane + 2, 64);      // the triple's other two partial sums
        if (on && sub3 == 0) st(bi_i, bi_j, acc + a1 + a2);
        return;
    }
    // border outputs: 4 lanes per output, lane `sub` sums k = sub, sub + 4, ... (same order as mm_f64_border)
    const int sub = tid & 3;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        if (p * (NW * 16) >= bi.count) break;                   // workgroup-uniform
        double acc = 0.0;
        if (bi.on[p]) {
            const TX *xp = X + bi.i[p] * pX + sub;
            const TW *wp = WT ? W + bi.j[p] * pW + sub : W + sub * pW + bi.j[p];
            int k = sub;
            for (; k + 28 < Kd; k += 32) {
                double xa[8], wb[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) { xa[u] = (double)xp[4 * u]; wb[u] = (double)wp[u * wstep]; }
                xp += 32; wp += 8 * wstep;
#pragma unroll
                for (int u = 0; u < 8; ++u) acc += xa[u] * wb[u];
            }
            if (k < Kd) {                                       // tail: up to 8 steps, clamped reads with zeroed products
                double xa[8], wb[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int uu = k + 4 * u < Kd ? u : 0;
                    xa[u] = (double)xp[4 * uu]; wb[u] = (double)wp[uu * wstep];
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) acc += (k + 4 * u < Kd) ? xa[u] * wb[u] : 0.0;
            }
        }
        acc += __shfl_xor(acc, 1, 64);
        acc += __shfl_xor(acc, 2, 64);
        if (bi.on[p] && sub == 0) st(bi.i[p], bi.j[p], acc);
    }
}

// ---- Second generation of the LDS-operand products (the round-3 coupling kernel) ------------------------------------------------
// Same tiling, fragment layout and border rule as mm_lds.  What changes:
//   KMAX > 0  : compile-time bound on the k-steps (Kd <= 4 KMAX): all operand reads of a tile (and of a border output) are issued
//               before the first MFMA / FMA — ONE LDS round trip per tile instead of one per 16 k.  KMAX = 0: chunked loops (any Kd).
//               Measured on the round-3 coupling kernel (same-job A/B, cfg2 shape): KMAX = 9 0.779 ms per solve, KMAX = 0 0.726 ms —
//               with five workgroups per CU the kernel is bound by vector-issue and register pressure, not by the length of a
//               tile's LDS chain, so the kernel uses KMAX = 0; the batched form is kept for low-occupancy callers.
//   HOLD      : the results stay in registers until `mid()` has run (every wavefront finishes READING its operands, then mid() — a
//               workgroup barrier —, then the stores), so that the output may overwrite an operand (A -> K in place) or a staging
//               area that shares storage with the output.  MAXT = compile-time bound on the tiles per wavefront (HOLD only).
//   tid       : callers inside a long loop pass a copy of threadIdx.x laundered through an empty asm (see mm_lds).
// Partial sums alternate between two accumulators per tile; the summation order differs from mm_lds in the last bits.
// Rows / columns of a tile that lie outside the matrix are read from a clamped (valid) row / column and produce accumulator rows /
// columns that are never stored — an MFMA output (i, j) depends on A row i and B column j only — so no operand is masked for them;
// only the ragged LAST k-step zeroes its out-of-range A elements (B is read from a clamped, finite location).
template <int KMAX, bool WT, typename TX, typename TW>
__device__ __forceinline__ f64x4 mm2_tile(const TX *__restrict__ xp, const TW *__restrict__ wp, int pW, int Kd, int lk) {
    f64x4 acc = {0.0, 0.0, 0.0, 0.0}, acc2 = {0.0, 0.0, 0.0, 0.0};
    const int nfull = Kd >> 2, rem = Kd & 3;                     // workgroup-uniform
    if constexpr (KMAX > 0) {
        // full k-steps: immediate-offset reads behind uniform guards, all in flight before the first MFMA
        double a[KMAX], b[KMAX];
#pragma unroll
        for (int u = 0; u < KMAX; ++u)
            if (u < nfull) { a[u] = (double)xp[4 * u]; b[u] = (double)(WT ? wp[4 * u] : wp[4 * u * pW]); }
        double at = 0.0, bt = 0.0;
        if (rem) {                                                // ragged last step: lanes lk >= rem read element 0 of the step and are zeroed
            const int o = 4 * nfull - (lk < rem ? 0 : lk);
            at = (double)xp[o]; bt = (double)(WT ? wp[o] : wp[o * pW]);
            at = lk < rem ? at : 0.0;
        }
#pragma unroll
        for (int u = 0; u < KMAX; ++u) {
            if (u < nfull) {
                if (u & 1) acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u], b[u], acc2, 0, 0, 0);
                else acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u], b[u], acc, 0, 0, 0);
            }
        }
        if (rem) acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(at, bt, acc2, 0, 0, 0);
    } else {
        const int wstep = WT ? 4 : 4 * pW;
        int k0 = 0;
        for (; k0 + 4 <= nfull; k0 += 4) {
            double a[4], b[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { a[u] = (double)xp[4 * u]; b[u] = (double)wp[u * wstep]; }
            xp += 16; wp += 4 * wstep;
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[0], b[0], acc, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[1], b[1], acc2, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[2], b[2], acc, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[3], b[3], acc2, 0, 0, 0);
        }
        for (; k0 < nfull; ++k0) {
            const double a = (double)xp[0], b = (double)wp[0];
            xp += 4; wp += wstep;
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
        }
        if (rem) {
            const int back = lk < rem ? 0 : lk;
            double a = (double)xp[-back];
            const double b = (double)(WT ? wp[-back] : wp[-back * pW]);
            a = lk < rem ? a : 0.0;
            acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc2, 0, 0, 0);
        }
    }
    return acc + acc2;
}

// one lane's share of a border output: k = sub, sub + S, ... (S lanes per output; xp / wp already point at k = sub)
template <int KMAX, int S, bool WT, typename TX, typename TW>
__device__ __forceinline__ double mm2_border_part(const TX *__restrict__ xp, const TW *__restrict__ wp, int pW, int Kd, int sub) {
    const int nfull = Kd / S, rem = Kd - nfull * S;              // workgroup-uniform: steps every lane of the group takes; lanes sub < rem take one more
    double acc = 0.0;
    if constexpr (KMAX > 0) {
        constexpr int KB = (4 * KMAX) / S;
        double xa[KB], wb[KB];
#pragma unroll
        for (int u = 0; u < KB; ++u)
            if (u < nfull) { xa[u] = (double)xp[S * u]; wb[u] = (double)(WT ? wp[S * u] : wp[S * u * pW]); }
        double xt = 0.0, wt = 0.0;
        if (rem) {
            const int o = S * nfull - (sub < rem ? 0 : sub);
            xt = (double)xp[o]; wt = (double)(WT ? wp[o] : wp[o * pW]);
            xt = sub < rem ? xt : 0.0;
        }
        double p0 = 0.0, p1 = 0.0, p2 = 0.0;
#pragma unroll
        for (int u = 0; u < KB; ++u) {
            if (u < nfull) {
                if (u % 3 == 0) p0 = fma(xa[u], wb[u], p0);
                else if (u % 3 == 1) p1 = fma(xa[u], wb[u], p1);
                else p2 = fma(xa[u], wb[u], p2);
            }
        }
        if (rem) p0 = fma(xt, wt, p0);
        acc = (p0 + p1) + p2;
    } else {
        const int ws = WT ? S : S * pW;
        int k0 = 0;
        double p0 = 0.0, p1 = 0.0;
        for (; k0 + 8 <= nfull; k0 += 8) {                        // 8 steps per trip: 16 reads in flight
            double xa[8], wb[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { xa[u] = (double)xp[S * u]; wb[u] = (double)wp[u * ws]; }
            xp += 8 * S; wp += 8 * ws;
#pragma unroll
            for (int u = 0; u < 8; u += 2) { p0 = fma(xa[u], wb[u], p0); p1 = fma(xa[u + 1], wb[u + 1], p1); }
        }
        for (; k0 < nfull; ++k0) { p0 = fma((double)xp[0], (double)wp[0], p0); xp += S; wp += ws; }
        if (rem) {
            const int back = sub < rem ? 0 : sub;
            double xt = (double)xp[-back];
            const double wt = (double)(WT ? wp[-back] : wp[-back * pW]);
            xt = sub < rem ? xt : 0.0;
            p1 = fma(xt, wt, p1);
        }
        acc = p0 + p1;
    }
    return acc;
}

template <int NW, int MAXT, int KMAX, bool WT, bool HOLD, typename TX, typename TW, class FM, class FS>
__device__ __forceinline__ void mm_lds2(int M, int Nn, int Kd, const TX *__restrict__ X, int pX, const TW *__restrict__ W, int pW, FM mid, FS st,
                                        const int tid = threadIdx.x) {
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lk = lane >> 4;
    const bool border = border_path<NW>(M, Nn);
    const int Mq = border ? M >> 4 : (M + 15) >> 4, Nq = border ? Nn >> 4 : (Nn + 15) >> 4;
    auto tile = [&](int t) {
        const int i0 = (t / Nq) << 4, j0 = (t % Nq) << 4;
        const int ia = i0 + li, jb = j0 + li;
        const bool ra = ia < M, cb = jb < Nn;
        const TX *xp = X + (ra ? ia : M - 1) * pX + lk;
        const TW *wp = WT ? W + (cb ? jb : Nn - 1) * pW + lk : W + lk * pW + (cb ? jb : Nn - 1);
        return mm2_tile<KMAX, WT>(xp, wp, pW, Kd, lk);
    };
    auto store_tile = [&](int t, const f64x4 &r) {
        const int i0 = (t / Nq) << 4, j0 = (t % Nq) << 4, jb = j0 + li;
        if (jb < Nn) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int i = i0 + lk + 4 * q;
                if (i < M) st(i, jb, r[q]);
            }
        }
    };
    f64x4 res[HOLD ? MAXT : 1];
    if constexpr (HOLD) {
#pragma unroll
        for (int tt = 0; tt < MAXT; ++tt) {
            const int t = wave + tt * NW;
            res[tt] = f64x4{0.0, 0.0, 0.0, 0.0};
            if (t < Mq * Nq) res[tt] = tile(t);                   // wave-uniform
        }
    } else {
        for (int t = wave; t < Mq * Nq; t += NW) store_tile(t, tile(t));
    }
    // border outputs: thin border (<= 21 per wavefront) = three lanes per output in one pass, else four lanes per output in up to two
    const int Mc = (M >> 4) << 4, Nc = (Nn >> 4) << 4, nb1 = (M - Mc) * Nn;
    const int bcount = border ? nb1 + Mc * (Nn - Nc) : 0;
    const bool thin = bcount <= NW * 21;
    double bval[2] = {0.0, 0.0};
    int b_i[2] = {0, 0}, b_j[2] = {0, 0};
    bool b_on[2] = {false, false};
    auto border_ij = [&](int o, int &bi, int &bj) {
        if (o < nb1) { bi = Mc + o / Nn; bj = o % Nn; }
        else { const int q = o - nb1; bi = q / (Nn - Nc); bj = Nc + q % (Nn - Nc); }
    };
    if (border && thin) {
        const int o = wave * 21 + lane / 3, sub3 = lane % 3;
        const bool on = lane < 63 && o < bcount;
        if (on) border_ij(o, b_i[0], b_j[0]);
        double acc = 0.0;
        if (on) {
            const TX *xp = X + b_i[0] * pX + sub3;
            const TW *wp = WT ? W + b_j[0] * pW + sub3 : W + sub3 * pW + b_j[0];
            acc = mm2_border_part<KMAX, 3, WT>(xp, wp, pW, Kd, sub3);
        }
        const double a1 = __shfl(acc, lane + 1, 64), a2 = __shfl(acc, lane + 2, 64);
        bval[0] = acc + a1 + a2;
        b_on[0] = on && sub3 == 0;
    } else if (border) {
        const int sub = tid & 3;
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            if (p * (NW * 16) >= bcount) break;
            const int o = p * (NW * 16) + (tid >> 2);
            const bool on = o < bcount;
            if (on) border_ij(o, b_i[p], b_j[p]);
            double acc = 0.0;
            if (on) {
                const TX *xp = X + b_i[p] * pX + sub;
                const TW *wp = WT ? W + b_j[p] * pW + sub : W + sub * pW + b_j[p];
                acc = mm2_border_part<KMAX, 4, WT>(xp, wp, pW, Kd, sub);
            }
            acc += __shfl_xor(acc, 1, 64);
            acc += __shfl_xor(acc, 2, 64);
            bval[p] = acc;
            b_on[p] = on && sub == 0;
        }
    }
    mid();
    if constexpr (HOLD) {
#pragma unroll
        for (int tt = 0; tt < MAXT; ++tt) {
            const int t = wave + tt * NW;
            if (t < Mq * Nq) store_tile(t, res[tt]);
        }
    }
#pragma unroll
    for (int p = 0; p < 2; ++p)
        if (b_on[p]) st(b_i[p], b_j[p], bval[p]);
}

// Launchers of the register-resident path (fgw_small.hip), N <= 64.
bool conan_fgw_small_supported(int N, int d);
bool conan_fgw_fast_supported(int N, int d, int small_int);
size_t conan_fgw_small_part_bytes(int B, int K, int N, int d);
size_t conan_fgw_part_offset(int B, int K, int N, int d);      // bytes of Ypart + Cpart (16-byte aligned): where the fp64 vectors start
// (also initialises the molecules: the N <= 64 path launches no k_fgw_init)
void conan_fgw_small_prepare(const float *Ys, const float *Cs, const float *ps, const float *pb, FgwDims D, conan_fgw_params prm,
                             double *Cw, double *Yw, double *zvec, double *yvec, const float *init_C, const float *init_Y, int *active, int *info,
                             float *errs, float *Yout, float *Cout, FgwAdj adj, hipStream_t s);
void conan_fgw_small_coupling(const float *Ys, const float *Cs, const float *ps, const float *pb, FgwDims D,
                              conan_fgw_params prm, int outer, int y_zero, const double *Cw, const double *Yw,
                              const int *active, float *Tw, int *info, fgw_part_t *Ypart, fgw_part_t *Cpart, const double *zvec,
                              const double *yvec, int *redo, FgwAdj adj, hipStream_t s);
// yvec (nullable): the register-resident path's per-molecule vectors, refreshed after every update
void conan_fgw_small_update(const float *pb, const float *lambdas, FgwDims D, conan_fgw_params prm, int outer,
                            const fgw_part_t *Ypart, const fgw_part_t *Cpart, double *Cw, double *Yw, int *active, int *info,
                            float *errs, float *Yout, float *Cout, double *yvec, const float *Tw, const float *Ys, hipStream_t s);
int conan_fgw_update_chunk(int K, int N, int d, int B);
