// fp32 MFMA linear layers for gfx950:  y = act(x @ W^T + b) (+ residual), the weight-gradient reduction, and the
// generic (any-shape) continuous-filter generator built from them.
//
// v_mfma_f32_32x32x2_f32 is exact fp32 (bitwise a k-ordered fmaf chain), so the result equals a plain fp32 GEMM up to
// summation order: no TF32-style truncation (the reference's CPU path is true fp32).
//
// Tiling: 256 threads = 4 wavefronts as 2(M) x 2(N); block tile 64 x (64*NB), K step 32.  Operands are staged in LDS
// with odd pitches (33 / BN+1 floats) so that both the transposed staging writes and the MFMA fragment reads
// (lane -> row for A, lane -> column for B) are bank-conflict free with ds_read_b32.
#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BM = 64;
constexpr int BK = 32;
constexpr int XP = BK + 1;   // pitch of the x tile  [BM][XP]

template <int NB>            // NB = 32-column blocks per wave (1 or 2); block tile N = 64*NB
__global__ void __launch_bounds__(256) k_linear(const float *__restrict__ x, const float *__restrict__ w,
                                                const float *__restrict__ bias, const float *__restrict__ residual,
                                                int M, int K, int N, int w_kn, int act, float *__restrict__ y,
                                                const int *__restrict__ m_dev) {
    constexpr int BN = 64 * NB;
    if (m_dev) M = min(M, *m_dev);                 // row count known only on the device (edge-level calls)
    if ((int)blockIdx.x * BM >= M) return;
    constexpr int WP = BN + 1;
    __shared__ float xs[BM * XP];
    __shared__ float ws[BK * WP];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int row0 = blockIdx.x * BM, col0 = blockIdx.y * BN;
    f32x16 acc[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[b][r] = 0.f;

    for (int k0 = 0; k0 < K; k0 += BK) {
        // stage x tile: 64 rows x 32 k, coalesced along k
        for (int t = tid; t < BM * BK; t += 256) {
            int r = t / BK, k = t % BK;
            int gr = row0 + r, gk = k0 + k;
            xs[r * XP + k] = (gr < M && gk < K) ? x[(size_t)gr * K + gk] : 0.f;
        }
        // stage W^T tile: ws[k][n]
        if (w_kn) {
            for (int t = tid; t < BK * BN; t += 256) {
                int k = t / BN, n = t % BN;
                int gk = k0 + k, gn = col0 + n;
                ws[k * WP + n] = (gk < K && gn < N) ? w[(size_t)gk * N + gn] : 0.f;
            }
        } else {
            for (int t = tid; t < BK * BN; t += 256) {
                int n = t / BK, k = t % BK;
                int gk = k0 + k, gn = col0 + n;
                ws[k * WP + n] = (gk < K && gn < N) ? w[(size_t)gn * K + gk] : 0.f;
            }
        }
        __syncthreads();
        const int arow = wm * 32 + (lane & 31), kh = lane >> 5;
#pragma unroll
        for (int kk = 0; kk < BK; kk += 2) {
            float a = xs[arow * XP + kk + kh];
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                float bv = ws[(kk + kh) * WP + wn * 32 * NB + b * 32 + (lane & 31)];
                acc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bv, acc[b], 0, 0, 0);
            }
        }
        __syncthreads();
    }
    // epilogue: D layout col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const int gc = col0 + wn * 32 * NB + b * 32 + (lane & 31);
        if (gc >= N) continue;
        const float bv = bias ? bias[gc] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int gr = row0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            if (gr >= M) continue;
            float v = acc[b][r] + bv;
            if (act == 1) v = ssp_f(v);
            if (act == 3) v = v / (1.0f + __expf(-v));          // SiLU
            if (act == 2) v *= 1.0f - 0.5f * __expf(-residual[(size_t)gr * N + gc]);      // * ssp'(pre) from the saved output
            else if (residual) v += residual[(size_t)gr * N + gc];
            y[(size_t)gr * N + gc] = v;
        }
    }
}

// ---- weight-resident variant for K <= 128, N <= 128 (every layer of SchNet-128) ------------------------------------
// Persistent workgroups: W^T is staged into LDS once ([K][N+1], <= 66 KB) and reused for every 128-row tile the workgroup
// processes; only x streams.  x chunks (128 rows x 32 k) are prefetched global->registers while the previous chunk is
// multiplied, then written to the other LDS buffer (one barrier per chunk).  4 wavefronts as 2x2, 64 rows x 32*NBW cols
// each => 2*NBW accumulators of 32x32.
constexpr int RM = 128;            // rows per tile
template <int NBW>
__global__ void __launch_bounds__(256) k_linear_res(const float *__restrict__ x, const float *__restrict__ w,
                                                    const float *__restrict__ bias, const float *__restrict__ residual,
                                                    int M, int K, int N, int w_kn, int act, float *__restrict__ y,
                                                    const int *__restrict__ m_dev) {
    constexpr int BN = 64 * NBW;
    constexpr int WP = BN + 1;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    if (m_dev) M = min(M, *m_dev);
    const int tiles = (M + RM - 1) / RM;
    if ((int)blockIdx.x >= tiles) return;
    const int Kp = (K + BK - 1) / BK * BK;                // K rounded up to the chunk size (zero filled)
    float *ws = lds;                                      // [Kp][WP]
    float *xs = lds + Kp * WP;                            // [2][RM][XP]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, kh = lane >> 5;
    // stage W^T once: 8 independent global loads in flight per thread before the LDS writes (the loop is latency-bound
    // otherwise: one L2 round trip per element, which dominates node-level calls that process a single tile)
    {
        const int total = Kp * BN;
        for (int t0 = tid; t0 < total; t0 += 256 * 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int t = t0 + u * 256;
                float val = 0.f;
                if (t < total) {
                    int k, n;
                    if (w_kn) { k = t / BN; n = t - k * BN; if (k < K && n < N) val = w[(size_t)k * N + n]; }
                    else { n = t / Kp; k = t - n * Kp; if (k < K && n < N) val = w[(size_t)n * K + k]; }
                }
                v[u] = val;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int t = t0 + u * 256;
                if (t < total) {
                    int k, n;
                    if (w_kn) { k = t / BN; n = t - k * BN; } else { n = t / Kp; k = t - n * Kp; }
                    ws[k * WP + n] = v[u];
                }
            }
        }
    }
    const int chunks = Kp / BK;
    // each thread prefetches 4 float4 per chunk: rows (tid>>3) + 32*q, columns (tid&7)*4 .. +3.  Branch-free: K is a
    // multiple of 32 on this path and rows past M are clamped to M-1 (their results are masked at the store).
    const int pr = tid >> 3, pc = (tid & 7) * 4;
    float4 pre[4];
    auto fetch = [&](int row0, int kc) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int gr = min(row0 + pr + 32 * q, M - 1);
            pre[q] = *reinterpret_cast<const float4 *>(x + (size_t)gr * K + kc * BK + pc);
        }
    };
    auto stash = [&](int buf) {
        float *dst = xs + buf * (RM * XP);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float *p = dst + (pr + 32 * q) * XP + pc;
            p[0] = pre[q].x; p[1] = pre[q].y; p[2] = pre[q].z; p[3] = pre[q].w;
        }
    };
    for (int tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        const int row0 = tile * RM;
        f32x16 acc[2][NBW];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < NBW; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
        fetch(row0, 0);
        __syncthreads();                                   // previous tile's readers of xs are done (and ws is staged)
        stash(0);
        __syncthreads();
        for (int kc = 0; kc < chunks; ++kc) {
            const int buf = kc & 1;
            if (kc + 1 < chunks) fetch(row0, kc + 1);
            const float *xb = xs + buf * (RM * XP);
            const float *wb = ws + (kc * BK) * WP;
#pragma unroll
            for (int kk = 0; kk < BK; kk += 2) {
                const float a0 = xb[(wm * 64 + l31) * XP + kk + kh];
                const float a1 = xb[(wm * 64 + 32 + l31) * XP + kk + kh];
#pragma unroll
                for (int b = 0; b < NBW; ++b) {
                    const float bv = wb[(kk + kh) * WP + wn * 32 * NBW + b * 32 + l31];
                    acc[0][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, bv, acc[0][b], 0, 0, 0);
                    acc[1][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, bv, acc[1][b], 0, 0, 0);
                }
            }
            if (kc + 1 < chunks) {
                stash(buf ^ 1);                            // the other buffer was last read one barrier ago
                __syncthreads();
            }
        }
        // epilogue.  The residual / saved-output operand is fetched for the whole fragment FIRST (independent loads in
        // flight together); interleaving each load with its store serialises 32*NBW HBM round trips per thread.
        f32x16 rv[2][NBW];
        if (residual) {
#pragma unroll
            for (int b = 0; b < NBW; ++b) {
                const int gc = wn * 32 * NBW + b * 32 + l31;
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int gr = row0 + wm * 64 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
                        rv[a][b][r] = (gc < N && gr < M) ? residual[(size_t)gr * N + gc] : 0.f;
                    }
            }
        }
#pragma unroll
        for (int b = 0; b < NBW; ++b) {
            const int gc = wn * 32 * NBW + b * 32 + l31;
            if (gc >= N) continue;
            const float bv = bias ? bias[gc] : 0.f;
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int gr = row0 + wm * 64 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
                    if (gr >= M) continue;
                    float v = acc[a][b][r] + bv;
                    if (act == 1) v = ssp_f(v);
                    if (act == 3) v = v / (1.0f + __expf(-v));      // SiLU
                    if (act == 2) v *= 1.0f - 0.5f * __expf(-rv[a][b][r]);   // * ssp'(pre) from the saved output
                    else if (residual) v += rv[a][b][r];
                    y[(size_t)gr * N + gc] = v;
                }
        }
    }
}

__global__ void k_ssp_bwd(const float *__restrict__ dy, const float *__restrict__ y, long long n, int width,
                          const int *__restrict__ m_dev, float *__restrict__ g) {
    if (m_dev) n = min(n, (long long)(*m_dev) * width);
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (; i < n; i += stride) g[i] = dy[i] * (1.0f - 0.5f * expf(-y[i]));   // ssp'(v) = sigmoid(v) = 1 - exp(-(y + ln2))
}

// ---- weight gradient: dW[N,K] = g^T @ x, reduction over the M rows ---------------------------------------------
// Stage 1: the rows are cut into slices, one workgroup per (slice, 128x128 tile of dW), partial slabs written per slice.
// Stage 2: the per-slice slabs are summed in a fixed order => bitwise reproducible, no float atomics.
constexpr int WG_SLICES_MAX = 768;      // 3 workgroups per CU


// bf16-split arithmetic of the weight gradient: 16 rows per MFMA step; an operand fragment (8 consecutive rows of one
// column per lane) is split exactly into three bf16 parts, six partial products per 32x32 block on
// v_mfma_f32_32x32x16_bf16 (fp32-class accuracy, 2.7x the fp32 MFMA rate).
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ void wg_split3(const float *v, bf16x8 &p1, bf16x8 &p2, bf16x8 &p3) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const __bf16 h1 = (__bf16)v[j];
        const float r1 = v[j] - (float)h1;
        const __bf16 h2 = (__bf16)r1;
        const float r2 = r1 - (float)h2;
        p1[j] = h1; p2[j] = h2; p3[j] = (__bf16)r2;
    }
}
__device__ __forceinline__ f32x16 wg_mma6(const bf16x8 &a1, const bf16x8 &a2, const bf16x8 &a3, const bf16x8 &b1, const bf16x8 &b2,
                                         const bf16x8 &b3, f32x16 acc) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3, b1, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b2, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b3, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b1, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b2, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc, 0, 0, 0);
    return acc;
}


// LDS-staged weight gradient.  Each value is loaded and split ONCE by a producer thread, in exactly the MFMA fragment
// unit: 8 consecutive rows of one column (lane <-> column, so the dword loads are coalesced 128-B segments), split into
// three bf16x8 planes and parked in LDS as three 16-byte words [plane][row group][column].  Consumers fetch a fragment
// with three contiguous ds_read_b128.  16 rows (one MFMA k-step) per stage, double-buffered, loads issued one stage ahead.
// KT = 128: 2x2 waves of 64x64;  KT = 64: 4 waves of 32(n) x 64(k).
// RBF = true: the x operand is the Gaussian expansion of dist[] (GaussianSmearing) generated by the producers instead of
// loaded — the backward of the first filter layer then reads no [P, 50] buffer at all.
// `slice` of `num_slices` row slices, 128-row tile `tile_n` of N, KT-wide tile `tile_k` of K: blockIdx / gridDim of the plain launch,
// decoded from a job table by the batched one.
// H16 (round 3, the edge-level dw2 = g^T h1 of the filter network): two fp16 planes per operand and three MFMAs per product instead of
// three bf16 planes and six (filter_fused.hip); g is scaled by s = 2^k with s * gmax in [16, 32) (gmax = max |g|, tracked by the kernel
// that produced g), x (shifted-softplus outputs, O(1)) goes in unscaled, the slab is unscaled as it is written.
typedef _Float16 wg_f16x8 __attribute__((ext_vector_type(8)));
template <int KT, bool RBF, bool H16 = false>
__device__ __forceinline__ void wgrad_lds_body(const float *__restrict__ g, const float *__restrict__ x, int M, int K, int N,
                                               float *__restrict__ slabs, float *__restrict__ bias_slabs,
                                               const int *__restrict__ m_dev, const float *__restrict__ dist,
                                               const float *__restrict__ offset, float coeff, int slice, int num_slices, int tile_n, int tile_k,
                                               const float *__restrict__ gmax = nullptr) {
    constexpr int ROWS = 16, RG = ROWS / 8, W = 128 + KT, FRAGS = RG * W, NF = (FRAGS + 255) / 256;
    constexpr int TNB = KT == 128 ? 2 : 1;
    constexpr int NPL = H16 ? 2 : 3;
    static_assert(W % 64 == 0, "a wave's 64 fragments share one row group and one operand");
    __shared__ uint4 frag[2][NPL][RG][W];
    if (m_dev) M = min(M, *m_dev);
    float gsc = 1.0f, gun = 1.0f;
    if constexpr (H16) {
        const float gm = *gmax;
        if (gm > 0.f && gm < 3.0e38f) { int e; (void)frexpf(gm, &e); gsc = ldexpf(1.0f, 5 - e); gun = ldexpf(1.0f, e - 5); }
    }
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, h = lane >> 5;
    const int n0 = KT == 128 ? (wave >> 1) * 64 : wave * 32, k0 = KT == 128 ? (wave & 1) * 64 : 0;     // inside the tile
    const int nb = tile_n * 128, kb = tile_k * KT;
    // stages are dealt round-robin: slice s owns rows [16 t, 16 t + 16) for t = s, s + slices, ...  At any moment the
    // resident workgroups then stream one contiguous window of g and x (all HBM channels busy) instead of `slices`
    // streams a power-of-two stride apart.
    const int r_begin = slice * ROWS, r_end = M, STEP = num_slices * ROWS;
    f32x16 acc[TNB][2];
#pragma unroll
    for (int a = 0; a < TNB; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    // Producer roles.  Fragment f = 64 wave + 256 i + lane  ->  (row group, column); the 64 fragments of one wave share the
    // row group and the operand (g or x), so row index, operand base and leading dimension are scalar registers and a load
    // costs no vector ALU work: scalar base + per-lane column offset.  Dead columns read column 0 and are masked.
    int f_rg[NF], f_col[NF];                         // wave-uniform row group, per-lane column inside [0, W)
    unsigned f_ldc[NF];                              // per-lane column offset of the load
    bool f_act[NF], f_isg[NF], f_all[NF], f_on[NF];
    float f_off[NF], bsum[NF];
#pragma unroll
    for (int i = 0; i < NF; ++i) {
        const int fb = 64 * wave + 256 * i;          // uniform
        f_act[i] = fb < FRAGS;
        f_rg[i] = fb / W;
        const int cb = fb - f_rg[i] * W;             // uniform, multiple of 64
        f_col[i] = cb + lane;
        f_isg[i] = cb < 128;
        const int cc = f_isg[i] ? nb + f_col[i] : kb + f_col[i] - 128;
        f_on[i] = cc < (f_isg[i] ? N : K);
        f_all[i] = __builtin_amdgcn_ballot_w64(f_on[i]) == ~0ull;
        f_ldc[i] = f_on[i] ? (unsigned)cc : 0u;
        f_off[i] = (RBF && !f_isg[i] && f_on[i]) ? offset[cc] : 0.f;
        bsum[i] = 0.f;
    }
    float stA[NF][8], stB[NF][8];                // two stages of loads in flight
    auto fetch = [&](float (&st)[NF][8], int m0) {
        const bool full = m0 + ROWS <= r_end;
#pragma unroll
        for (int i = 0; i < NF; ++i) {
            if (!f_act[i]) continue;
            const int mr = m0 + 8 * f_rg[i];
            if (RBF && !f_isg[i]) {
#pragma unroll
                for (int j = 0; j < 8; ++j) st[i][j] = dist[full ? mr + j : min(mr + j, r_end - 1)];      // scalar loads
            } else {
                const float *src = f_isg[i] ? g : x;
                const int ld = f_isg[i] ? N : K;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int m = full ? mr + j : min(mr + j, r_end - 1);
                    st[i][j] = (src + (size_t)m * ld)[f_ldc[i]];
                }
            }
        }
    };
    auto stash = [&](int buf, float (&st)[NF][8], int m0) {
        const bool full = m0 + ROWS <= r_end;
#pragma unroll
        for (int i = 0; i < NF; ++i) {
            if (!f_act[i]) continue;
            if (RBF && !f_isg[i]) {
#pragma unroll
                for (int j = 0; j < 8; ++j) { const float t = st[i][j] - f_off[i]; st[i][j] = exp_neg_f(coeff * (t * t)); }
            }
            if (!full || !f_all[i]) {
#pragma unroll
                for (int j = 0; j < 8; ++j) st[i][j] = (f_on[i] && (m0 + 8 * f_rg[i] + j < r_end)) ? st[i][j] : 0.f;
            }
            if (f_isg[i]) bsum[i] += ((st[i][0] + st[i][1]) + (st[i][2] + st[i][3])) + ((st[i][4] + st[i][5]) + (st[i][6] + st[i][7]));
            if constexpr (H16) {
                const float sc = f_isg[i] ? gsc : 1.0f;
                wg_f16x8 p1, p2;
#pragma unroll
                for (int j = 0; j < 8; ++j) { const float v = st[i][j] * sc; p1[j] = (_Float16)v; p2[j] = (_Float16)(v - (float)p1[j]); }
                frag[buf][0][f_rg[i]][f_col[i]] = __builtin_bit_cast(uint4, p1);
                frag[buf][1][f_rg[i]][f_col[i]] = __builtin_bit_cast(uint4, p2);
                continue;
            }
            bf16x8 p1, p2, p3;
            wg_split3(st[i], p1, p2, p3);
            frag[buf][0][f_rg[i]][f_col[i]] = __builtin_bit_cast(uint4, p1);
            frag[buf][1][f_rg[i]][f_col[i]] = __builtin_bit_cast(uint4, p2);
            frag[buf][2][f_rg[i]][f_col[i]] = __builtin_bit_cast(uint4, p3);
        }
    };
    auto compute = [&](int buf) {
        if constexpr (H16) {
            wg_f16x8 p[TNB][2], q[2][2];
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) {
#pragma unroll
                for (int a = 0; a < TNB; ++a) p[a][pl] = __builtin_bit_cast(wg_f16x8, frag[buf][pl][h][n0 + 32 * a + l31]);
#pragma unroll
                for (int b = 0; b < 2; ++b) q[b][pl] = __builtin_bit_cast(wg_f16x8, frag[buf][pl][h][128 + k0 + 32 * b + l31]);
            }
            constexpr int HA[3] = {1, 0, 0}, HB[3] = {0, 1, 0};          // (a2,b1) (a1,b2) (a1,b1): smallest terms first
#pragma unroll
            for (int t = 0; t < 3; ++t)
#pragma unroll
                for (int a = 0; a < TNB; ++a)
#pragma unroll
                    for (int b = 0; b < 2; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(p[a][HA[t]], q[b][HB[t]], acc[a][b], 0, 0, 0);
            return;
        }
        bf16x8 p[TNB][3], q[2][3];
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
#pragma unroll
            for (int a = 0; a < TNB; ++a) p[a][pl] = __builtin_bit_cast(bf16x8, frag[buf][pl][h][n0 + 32 * a + l31]);
#pragma unroll
            for (int b = 0; b < 2; ++b) q[b][pl] = __builtin_bit_cast(bf16x8, frag[buf][pl][h][128 + k0 + 32 * b + l31]);
        }
        // the six partial products of a block are issued round-robin over the 2 * TNB accumulators: back-to-back MFMAs on
        // ONE accumulator wait for each other's 16 passes, independent ones pipeline
        constexpr int PA[6] = {2, 1, 0, 1, 0, 0}, PB[6] = {0, 1, 2, 0, 1, 0};
#pragma unroll
        for (int t = 0; t < 6; ++t)
#pragma unroll
            for (int a = 0; a < TNB; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(p[a][PA[t]], q[b][PB[t]], acc[a][b], 0, 0, 0);
    };

#define WG_HALF(BUF, RA)                                                                                     \
    {                                                                                                        \
        compute(BUF);                                                                                        \
        if (m + STEP < r_end) stash((BUF) ^ 1, RA, m + STEP);                                                \
        if (m + 3 * STEP < r_end) fetch(RA, m + 3 * STEP);                                                   \
        __syncthreads();                                                                                     \
        m += STEP;                                                                                           \
        if (m >= r_end) break;                                                                               \
    }
    if (r_begin < r_end) {
        int m = r_begin;
        fetch(stA, m);
        if (m + STEP < r_end) fetch(stB, m + STEP);  // requested before the first wait: a slice of a node-level layer is only 8 stages
        stash(0, stA, m);
        if (m + 2 * STEP < r_end) fetch(stA, m + 2 * STEP);
        __syncthreads();
        for (;;) {                                   // stage m is in buffer 0; stB holds the next stage, stA the one after
            WG_HALF(0, stB) WG_HALF(1, stA)
        }
    }
#undef WG_HALF

    float *slab = slabs + (size_t)slice * N * K;
#pragma unroll
    for (int a = 0; a < TNB; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = nb + n0 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                const int k = kb + k0 + b * 32 + l31;
                if (n < N && k < K) slab[(size_t)n * K + k] = acc[a][b][r] * gun;      // (H16: undo the gradient's scale; 1 otherwise)
            }
    if (tile_k == 0) {                           // bias gradient = column sums of g, combined over the row groups in LDS
        float *bp = reinterpret_cast<float *>(&frag[0][0][0][0]);            // [RG][128]; the stages are idle by now
#pragma unroll
        for (int i = 0; i < NF; ++i)
            if (f_act[i] && f_isg[i]) bp[f_rg[i] * 128 + f_col[i]] = bsum[i];
        __syncthreads();
        if (tid < 128 && nb + tid < N) {
            float t = 0.f;
#pragma unroll
            for (int r = 0; r < RG; ++r) t += bp[r * 128 + tid];
            bias_slabs[(size_t)slice * N + nb + tid] = t;
        }
    }
}

template <int KT, bool RBF>
__global__ void __launch_bounds__(256, KT == 128 ? 2 : 3) k_wgrad_lds(const float *__restrict__ g, const float *__restrict__ x, int M, int K, int N,
                                                   float *__restrict__ slabs, float *__restrict__ bias_slabs,
                                                   const int *__restrict__ m_dev, const float *__restrict__ dist,
                                                   const float *__restrict__ offset, float coeff, int tn_remap, int slices) {
    // tn_remap > 1 (1-D grid, N > 128, slices % 8 == 0): the n tiles of one row slice sit on consecutive block ids of the SAME XCD
    // (id = lo + 8 (tile_n + tn * hi), slice = lo + 8 hi), so that they stream the slice's x rows together and all but the first find
    // them in that XCD's L2 — with the 3-D grid (slice fastest) every n tile re-read x from HBM a whole sweep later
    if (tn_remap > 1) {
        const int b = blockIdx.x, lo = b & 7, q = b >> 3;
        wgrad_lds_body<KT, RBF>(g, x, M, K, N, slabs, bias_slabs, m_dev, dist, offset, coeff, lo + 8 * (q / tn_remap), slices, q % tn_remap, 0);
        return;
    }
    wgrad_lds_body<KT, RBF>(g, x, M, K, N, slabs, bias_slabs, m_dev, dist, offset, coeff, blockIdx.x, gridDim.x, blockIdx.y, blockIdx.z);
}
__global__ void __launch_bounds__(256, 2) k_wgrad_lds_h16(const float *__restrict__ g, const float *__restrict__ x, int M, int K, int N,
                                                          float *__restrict__ slabs, float *__restrict__ bias_slabs, const int *__restrict__ m_dev,
                                                          const float *__restrict__ gmax) {
    wgrad_lds_body<128, false, true>(g, x, M, K, N, slabs, bias_slabs, m_dev, nullptr, nullptr, 0.f, blockIdx.x, gridDim.x, blockIdx.y, blockIdx.z, gmax);
}

// Many weight gradients in ONE launch.  A node-level layer (25 k rows) is a latency chain of 8 stages in 198 workgroups — 16-21 us
// with most of the chip idle — and a backward pass of the stage-2 model has 22 of them, all independent of each other once their
// g and x exist.  Workgroup b belongs to the job whose [first, first + count) range contains it and is decoded into (slice, n tile,
// k tile) there; per-job arithmetic and slab layout are exactly those of the plain launch (bitwise-equal results).
constexpr int WGS_BATCH = 24;
struct WgradSlabJobs {
    const float *g[WGS_BATCH], *x[WGS_BATCH];
    const int *m_dev[WGS_BATCH];
    float *slabs[WGS_BATCH], *bias_slabs[WGS_BATCH];
    int M[WGS_BATCH], K[WGS_BATCH], N[WGS_BATCH], slices[WGS_BATCH], tiles_n[WGS_BATCH], first[WGS_BATCH + 1];
    int gsz[WGS_BATCH];      // > 1 at the LAST job of a run of jobs that read the same x (same M, K, N <= 128, slices % 8 == 0): interleaved, see the kernel
    int count;
};
template <int KT>
__global__ void __launch_bounds__(256, KT == 128 ? 2 : 3) k_wgrad_lds_batch(const WgradSlabJobs J) {
    const int b = blockIdx.x;
    int j = 0;
    while (j + 1 < J.count && b >= J.first[j + 1]) ++j;            // uniform scan over <= 24 entries held in scalar registers
    const int local = b - J.first[j];
    if (J.gsz[j] > 1) {
        // a run of jobs over the SAME x (q / k / v, dk / dv / f_proj of one input): their workgroups for one row slice sit on consecutive
        // block ids of one XCD (id = lo + 8 (member + gsz * hi), slice = lo + 8 hi) and stream the slice's x rows together — x comes
        // from HBM once per run instead of once per job.  The members' ranges are zero-width except the last one's, which the scan finds.
        const int gs = J.gsz[j], lo = local & 7, q = local >> 3, jm = j - (gs - 1) + q % gs;
        wgrad_lds_body<KT, false>(J.g[jm], J.x[jm], J.M[jm], J.K[jm], J.N[jm], J.slabs[jm], J.bias_slabs[jm], J.m_dev[jm], nullptr, nullptr, 0.f,
                                  lo + 8 * (q / gs), J.slices[jm], 0, 0);
        return;
    }
    const int slice = local % J.slices[j], rest = local / J.slices[j];
    const int tile_n = rest % J.tiles_n[j], tile_k = rest / J.tiles_n[j];
    wgrad_lds_body<KT, false>(J.g[j], J.x[j], J.M[j], J.K[j], J.N[j], J.slabs[j], J.bias_slabs[j], J.m_dev[j], nullptr, nullptr, 0.f, slice,
                              J.slices[j], tile_n, tile_k);
}

// Several 128 x 128 weight gradients over the SAME x in one workgroup (round 5): dW_j = g_j^T x for j < NJ, K = 128, N_j = 128 — ViS_MP's
// dk / dv / f_proj of one f_ij (three [E,128] gradients, one x) and the two n tiles of s_proj (g [E,256]).  k_wgrad_lds_batch ran them as NJ
// workgroups side by side on one XCD, each staging x again (from L2) and each paying the full round trip per 16-row stage: 391 us for the 4
// tensors of a run of three against 135 us for the 2 of a single job — the kernel is bound by the bytes a CU keeps in flight (two stages of
// loads per workgroup), not by the bytes it moves, so a stage that feeds NJ products costs what a stage that feeds one does.  Here a stage
// is 16 rows of x and of all NJ gradients, x is split into its planes once, and every wave holds NJ sets of accumulators (NJ = 3: one
// workgroup per CU, ~350 registers).  Slices, stage order, the six partial products per block and the slab layout are those of
// wgrad_lds_body: the slabs are bit for bit what NJ separate launches write.
template <int NJ>
struct WgShared {
    const float *g[NJ];          // gradient of job j, row pitch ldg
    float *slab[NJ], *bslab[NJ]; // slab / bias slab of slice 0 for job j (n offset included), slice pitches slab_pitch / bias_pitch
    const float *x;
    const int *m_dev;
    int M, ldg, slices;
    long long slab_pitch;
    int bias_pitch;
};
// NW = 4: 2 x 2 waves of 64 x 64 per job (NJ = 2: two workgroups per CU); NW = 8: 4 x 2 waves of 32(n) x 64(k) per job — NJ = 3 then needs ~200
// registers instead of 344, two waves per SIMD cover each other's split / MFMA / wait phases (with one, a stage was the SUM of its phases).
template <int NJ, int D, int NW>
__global__ void __launch_bounds__(64 * NW, (NJ == 2 && NW == 4) ? 2 : 1) k_wgrad_lds_shared(const WgShared<NJ> P) {
    constexpr int K = 128, ROWS = 16, RG = 2, W = 128 * (NJ + 1), FRAGS = RG * W, NT = 64 * NW, NF = FRAGS / NT, TNB = NW == 8 ? 1 : 2;
    static_assert(FRAGS % NT == 0, "whole fragments per thread");
    extern __shared__ __attribute__((aligned(16))) uint4 frag_sh[];          // [2][3][RG][W]
    auto frag = [&](int buf, int pl, int rg, int col) -> uint4 & { return frag_sh[((buf * 3 + pl) * RG + rg) * W + col]; };
    int M = P.M;
    if (P.m_dev) M = min(M, *P.m_dev);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, h = lane >> 5;
    const int n0 = (wave >> 1) * (32 * TNB), k0 = (wave & 1) * 64;
    const int slice = blockIdx.x, num_slices = P.slices;
    const int r_begin = slice * ROWS, r_end = M, STEP = num_slices * ROWS;
    f32x16 acc[NJ][TNB][2];
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int a = 0; a < TNB; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[j][a][b][r] = 0.f;
    // producer roles: fragment f = 64 wave + NT i + lane -> (row group, operand, column); wave-uniform row group and operand
    int f_rg[NF], f_col[NF], f_op[NF];
    float bsum[NF];
#pragma unroll
    for (int i = 0; i < NF; ++i) {
        const int fb = 64 * wave + NT * i;
        f_rg[i] = fb / W;
        const int cb = fb - f_rg[i] * W;
        f_op[i] = cb >> 7;                            // 0 .. NJ - 1: gradient of job op; NJ: x
        f_col[i] = cb + lane;                         // column inside [0, W)
        bsum[i] = 0.f;
    }
    float st[D][NF][8];
    auto fetch = [&](float (&sr)[NF][8], int m0) {
        const bool full = m0 + ROWS <= r_end;
#pragma unroll
        for (int i = 0; i < NF; ++i) {
            const int mr = m0 + 8 * f_rg[i];
            const bool isg = f_op[i] < NJ;
            const float *src = P.x;
#pragma unroll
            for (int j = 0; j < NJ; ++j) src = f_op[i] == j ? P.g[j] : src;
            const int ld = isg ? P.ldg : K;
            const unsigned cc = (unsigned)(f_col[i] & 127);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int m = full ? mr + j : min(mr + j, r_end - 1);
                sr[i][j] = (src + (size_t)m * ld)[cc];
            }
        }
    };
    auto stash = [&](int buf, float (&sr)[NF][8], int m0) {
        const bool full = m0 + ROWS <= r_end;
#pragma unroll
        for (int i = 0; i < NF; ++i) {
            if (!full) {
#pragma unroll
                for (int j = 0; j < 8; ++j) sr[i][j] = (m0 + 8 * f_rg[i] + j < r_end) ? sr[i][j] : 0.f;
            }
            if (f_op[i] < NJ) bsum[i] += ((sr[i][0] + sr[i][1]) + (sr[i][2] + sr[i][3])) + ((sr[i][4] + sr[i][5]) + (sr[i][6] + sr[i][7]));
            bf16x8 p1, p2, p3;
            wg_split3(sr[i], p1, p2, p3);
            frag(buf, 0, f_rg[i], f_col[i]) = __builtin_bit_cast(uint4, p1);
            frag(buf, 1, f_rg[i], f_col[i]) = __builtin_bit_cast(uint4, p2);
            frag(buf, 2, f_rg[i], f_col[i]) = __builtin_bit_cast(uint4, p3);
        }
    };
    auto compute = [&](int buf) {
        bf16x8 q[2][3];
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
#pragma unroll
            for (int b = 0; b < 2; ++b) q[b][pl] = __builtin_bit_cast(bf16x8, frag(buf, pl, h, 128 * NJ + k0 + 32 * b + l31));
        constexpr int PA[6] = {2, 1, 0, 1, 0, 0}, PB[6] = {0, 1, 2, 0, 1, 0};
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            bf16x8 p[TNB][3];
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
#pragma unroll
                for (int a = 0; a < TNB; ++a) p[a][pl] = __builtin_bit_cast(bf16x8, frag(buf, pl, h, 128 * j + n0 + 32 * a + l31));
#pragma unroll
            for (int t = 0; t < 6; ++t)
#pragma unroll
                for (int a = 0; a < TNB; ++a)
#pragma unroll
                    for (int b = 0; b < 2; ++b)
                        acc[j][a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(p[a][PA[t]], q[b][PB[t]], acc[j][a][b], 0, 0, 0);
        }
    };
    if (r_begin < r_end) {
        int m = r_begin;
#pragma unroll
        for (int d = 0; d < D; ++d)
            if (m + d * STEP < r_end) fetch(st[d], m + d * STEP);
        stash(0, st[0], m);
        if (m + D * STEP < r_end) fetch(st[0], m + D * STEP);
        __syncthreads();
        constexpr int U = D == 2 ? 2 : 6;             // stage i: LDS buffer i & 1, register set (i + 1) % D for the stage after it
        bool more = true;
        while (more) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (!more) break;
                const int buf = u & 1, rs = (u + 1) % D;
                compute(buf);
                if (m + STEP < r_end) stash(buf ^ 1, st[rs], m + STEP);
                if (m + (D + 1) * STEP < r_end) fetch(st[rs], m + (D + 1) * STEP);
                __syncthreads();
                m += STEP;
                more = m < r_end;
            }
        }
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        float *slab = P.slab[j] + (size_t)slice * P.slab_pitch;
#pragma unroll
        for (int a = 0; a < TNB; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int n = n0 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                    const int k = k0 + b * 32 + l31;
                    slab[(size_t)n * K + k] = acc[j][a][b][r];
                }
    }
    float *bp = reinterpret_cast<float *>(frag_sh);    // [RG][128 NJ]; the stages are idle by now
#pragma unroll
    for (int i = 0; i < NF; ++i)
        if (f_op[i] < NJ) bp[f_rg[i] * (128 * NJ) + f_col[i]] = bsum[i];
    __syncthreads();
    if (tid < 128) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            float t = 0.f;
#pragma unroll
            for (int r = 0; r < RG; ++r) t += bp[r * (128 * NJ) + 128 * j + tid];
            P.bslab[j][(size_t)slice * P.bias_pitch + tid] = t;
        }
    }
}
#ifndef CONAN_WGRAD_SHARED_CFG
#define CONAN_WGRAD_SHARED_CFG 2248                  // digits: stages of loads in flight (three jobs, two jobs), waves per workgroup (two jobs, three jobs)
#endif
#ifndef CONAN_WGRAD_SHARED_MIN_ROWS
#define CONAN_WGRAD_SHARED_MIN_ROWS 65536            // (= the row count from which a job has all 512 slices; 0x7fffffff switches the kernel off)
#endif
template <int NJ>
static void launch_wgrad_shared(const WgShared<NJ> &P, hipStream_t s) {
    constexpr int CFG = CONAN_WGRAD_SHARED_CFG;
    constexpr int D = NJ == 3 ? CFG / 1000 : (CFG / 100) % 10, NW = NJ == 3 ? CFG % 10 : (CFG / 10) % 10;
    const size_t lds = (size_t)2 * 3 * 2 * 128 * (NJ + 1) * sizeof(uint4);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_wgrad_lds_shared<NJ, D, NW>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    k_wgrad_lds_shared<NJ, D, NW><<<P.slices, 64 * NW, lds, s>>>(P);
}

// slabs [slices][NK] (+ bias_slabs [slices][N]) -> out[g][NK] (+ bout[g][N]) for slice group g = blockIdx.y; fixed order.
__global__ void k_wgrad_reduce(const float *__restrict__ slabs, const float *__restrict__ bias_slabs, int slices, int per_group,
                               int NK, int N, float *__restrict__ out, float *__restrict__ bout) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int g = blockIdx.y;
    const int s0 = g * per_group, s1 = min(slices, s0 + per_group);
    if (i < NK) {
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        int sl = s0;
        for (; sl + 3 < s1; sl += 4) {
            a0 += slabs[(size_t)sl * NK + i]; a1 += slabs[(size_t)(sl + 1) * NK + i];
            a2 += slabs[(size_t)(sl + 2) * NK + i]; a3 += slabs[(size_t)(sl + 3) * NK + i];
        }
        for (; sl < s1; ++sl) a0 += slabs[(size_t)sl * NK + i];
        out[(size_t)g * NK + i] = (a0 + a1) + (a2 + a3);
    }
    if (bout && i < N) {
        float a = 0.f;
        for (int sl = s0; sl < s1; ++sl) a += bias_slabs[(size_t)sl * N + i];
        bout[(size_t)g * N + i] = a;
    }
}

// One-launch reduction for NK % 4 == 0 && N % 4 == 0: a workgroup owns 32 float4 outputs and splits the slices 8 ways
// (thread = (slice residue g, column c)); the 8 partial sums are combined through LDS in the fixed order g = 0..7, so the
// result is still bitwise reproducible.  The bias sums ride along as N/4 extra float4 columns.  (The two-launch version
// above spent 7.5 us per launch, 41 launches per training step.)
__global__ void __launch_bounds__(256) k_wgrad_reduce4(const float4 *__restrict__ slabs, const float4 *__restrict__ bias_slabs, int slices,
                                                       int NK4, int N4, float4 *__restrict__ out, float4 *__restrict__ bout) {
    __shared__ float4 part[8][32];
    const int g = threadIdx.x >> 5, c = threadIdx.x & 31;
    const int idx = blockIdx.x * 32 + c;
    const bool is_w = idx < NK4, is_b = !is_w && bout && idx - NK4 < N4;
    const float4 *src = is_w ? slabs + idx : bias_slabs + (idx - NK4);
    const int stride = is_w ? NK4 : N4;
    float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0, a2 = a0, a3 = a0;
    if (is_w || is_b) {
        int s = g;
        for (; s + 56 < slices; s += 64) {                    // 8 independent loads in flight per trip
            float4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = src[(size_t)(s + 8 * u) * stride];
#pragma unroll
            for (int u = 0; u < 8; u += 4) {
                a0.x += v[u].x; a0.y += v[u].y; a0.z += v[u].z; a0.w += v[u].w;
                a1.x += v[u + 1].x; a1.y += v[u + 1].y; a1.z += v[u + 1].z; a1.w += v[u + 1].w;
                a2.x += v[u + 2].x; a2.y += v[u + 2].y; a2.z += v[u + 2].z; a2.w += v[u + 2].w;
                a3.x += v[u + 3].x; a3.y += v[u + 3].y; a3.z += v[u + 3].z; a3.w += v[u + 3].w;
            }
        }
        for (; s < slices; s += 8) { const float4 v0 = src[(size_t)s * stride]; a0.x += v0.x; a0.y += v0.y; a0.z += v0.z; a0.w += v0.w; }
    }
    a0.x += a2.x; a0.y += a2.y; a0.z += a2.z; a0.w += a2.w;
    a1.x += a3.x; a1.y += a3.y; a1.z += a3.z; a1.w += a3.w;
    part[g][c] = make_float4(a0.x + a1.x, a0.y + a1.y, a0.z + a1.z, a0.w + a1.w);
    __syncthreads();
    if (g == 0 && (is_w || is_b)) {
        float4 r = part[0][c];
#pragma unroll
        for (int q = 1; q < 8; ++q) { const float4 v = part[q][c]; r.x += v.x; r.y += v.y; r.z += v.z; r.w += v.w; }
        if (is_w) out[idx] = r; else bout[idx - NK4] = r;
    }
}

// Batched form: blockIdx.y selects one of up to WG_BATCH weight gradients (same per-job arithmetic and order as k_wgrad_reduce4), so
// the slab reductions of a whole backward pass are ONE launch instead of one per Linear layer (24 per SchNet training step).
constexpr int WG_BATCH = 32;
struct WgradJobs {
    const float4 *slabs[WG_BATCH];
    const float4 *bias_slabs[WG_BATCH];
    float4 *out[WG_BATCH];
    float4 *bout[WG_BATCH];
    int slices[WG_BATCH], NK4[WG_BATCH], N4[WG_BATCH];
};
__global__ void __launch_bounds__(256) k_wgrad_reduce4_batch(WgradJobs J) {
    __shared__ float4 part[8][32];
    const int job = blockIdx.y;
    const int NK4 = J.NK4[job], N4 = J.N4[job], slices = J.slices[job];
    float4 *bout = J.bout[job];
    const int g = threadIdx.x >> 5, c = threadIdx.x & 31;
    const int idx = blockIdx.x * 32 + c;
    if ((int)blockIdx.x * 32 >= NK4 + (bout ? N4 : 0)) return;           // workgroup-uniform: this job has fewer columns than the widest one
    const bool is_w = idx < NK4, is_b = !is_w && bout && idx - NK4 < N4;
    const float4 *src = is_w ? J.slabs[job] + idx : J.bias_slabs[job] + (idx - NK4);
    const int stride = is_w ? NK4 : N4;
    float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0, a2 = a0, a3 = a0;
    if (is_w || is_b) {
        int s = g;
        for (; s + 56 < slices; s += 64) {                    // 8 independent loads in flight per trip
            float4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = src[(size_t)(s + 8 * u) * stride];
#pragma unroll
            for (int u = 0; u < 8; u += 4) {
                a0.x += v[u].x; a0.y += v[u].y; a0.z += v[u].z; a0.w += v[u].w;
                a1.x += v[u + 1].x; a1.y += v[u + 1].y; a1.z += v[u + 1].z; a1.w += v[u + 1].w;
                a2.x += v[u + 2].x; a2.y += v[u + 2].y; a2.z += v[u + 2].z; a2.w += v[u + 2].w;
                a3.x += v[u + 3].x; a3.y += v[u + 3].y; a3.z += v[u + 3].z; a3.w += v[u + 3].w;
            }
        }
        for (; s < slices; s += 8) { const float4 v0 = src[(size_t)s * stride]; a0.x += v0.x; a0.y += v0.y; a0.z += v0.z; a0.w += v0.w; }
    }
    a0.x += a2.x; a0.y += a2.y; a0.z += a2.z; a0.w += a2.w;
    a1.x += a3.x; a1.y += a3.y; a1.z += a3.z; a1.w += a3.w;
    part[g][c] = make_float4(a0.x + a1.x, a0.y + a1.y, a0.z + a1.z, a0.w + a1.w);
    __syncthreads();
    if (g == 0 && (is_w || is_b)) {
        float4 r = part[0][c];
#pragma unroll
        for (int q = 1; q < 8; ++q) { const float4 v = part[q][c]; r.x += v.x; r.y += v.y; r.z += v.z; r.w += v.w; }
        if (is_w) J.out[job][idx] = r; else bout[idx - NK4] = r;
    }
}

constexpr int WG_GROUPS = 16;

static int wgrad_slices(int M, int K) {
    int s = (M + 127) / 128;           // >= 128 rows per slice
    if (s < 1) s = 1;
    const int cap = K > 64 ? 512 : WG_SLICES_MAX;      // resident workgroups per CU: 2 (128-wide k tile) or 3
    if (s > cap) s = cap;
    return s;
}

}  // namespace

int conan_linear_t_try(const float *x, const float *w, const float *bias, const float *residual, int M, int K, int N, int w_kn,
                       int act, float *y, const int *m_dev, hipStream_t s, int *rc, float *pre_out);      // gemm_t.hip
int conan_linear_t_multi(const float *x, const float *const *w, const float *const *bias, int M, int K, int N, int njobs, int act, float *const *y,
                         float *const *pre, const int *m_dev, hipStream_t s, int *rc);                    // gemm_t.hip
int conan_linear_t_sum(const float *const *x, const int *ldx, const float *const *w, int nsrc, int w_kn, int ldw, const float *bias,
                       const float *residual, int M, int N, float *y, const int *m_dev, hipStream_t s, int *rc);   // gemm_t.hip

extern "C" {

int conan_linear_fwd(const float *x, const float *w, const float *bias, const float *residual, int M, int K, int N,
                     int w_kn, int act, const int *m_dev, float *y, void *stream) {
    if (!x || !w || !y || M < 0 || K <= 0 || N <= 0 || act < 0 || act > 3 || (act == 2 && !residual)) return CONAN_E_BADARG;
    if (M == 0) return CONAN_OK;
    hipStream_t s = as_stream(stream);
    {
        int rc = CONAN_OK;
        if (conan_linear_t_try(x, w, bias, residual, M, K, N, w_kn, act, y, m_dev, s, &rc, nullptr)) return rc;   // K, N multiples of 64
    }
    if (K <= 128 && N <= 128 && (K % BK) == 0) {
        const int Kp = (K + BK - 1) / BK * BK;
        const int tiles = (M + RM - 1) / RM;
        const int grid = tiles < 512 ? tiles : 512;        // persistent: <= 2 workgroups per CU
        if (N > 64) {
            const size_t lds = ((size_t)Kp * 129 + 2 * RM * XP) * 4;
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_linear_res<2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            k_linear_res<2><<<grid, 256, lds, s>>>(x, w, bias, residual, M, K, N, w_kn, act, y, m_dev);
        } else {
            const size_t lds = ((size_t)Kp * 65 + 2 * RM * XP) * 4;
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_linear_res<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            k_linear_res<1><<<grid, 256, lds, s>>>(x, w, bias, residual, M, K, N, w_kn, act, y, m_dev);
        }
        CONAN_LAUNCH_CHECK();
        return CONAN_OK;
    }
    if (N > 64) {
        dim3 grid((M + BM - 1) / BM, (N + 127) / 128);
        k_linear<2><<<grid, 256, 0, s>>>(x, w, bias, residual, M, K, N, w_kn, act, y, m_dev);
    } else {
        dim3 grid((M + BM - 1) / BM, (N + 63) / 64);
        k_linear<1><<<grid, 256, 0, s>>>(x, w, bias, residual, M, K, N, w_kn, act, y, m_dev);
    }
    CONAN_LAUNCH_CHECK();
    return CONAN_OK;
}

int conan_linear_act_fwd(const float *x, const float *w, const float *bias, int M, int K, int N, int act, const int *m_dev, float *y,
                         float *pre, void *stream) {
    if (!x || !w || !y || !pre || M < 0 || K <= 0 || N <= 0 || (act != 1 && act != 3)) return CONAN_E_BADARG;
    if (M == 0) return CONAN_OK;
    int rc = CONAN_OK;
    if (conan_linear_t_try(x, w, bias, nullptr, M, K, N, 0, act, y, m_dev, as_stream(stream), &rc, pre)) return rc;
    return CONAN_E_UNSUPPORTED;
}

int conan_linear_multi_fwd(const float *x, const float *const *w, const float *const *bias, int M, int K, int N, int num_layers, int act,
                           const int *m_dev, float *const *y, float *const *pre, void *stream) {
    if (!x || !w || !y || M < 0 || K <= 0 || N <= 0 || num_layers < 1 || (act != 0 && act != 1 && act != 3)) return CONAN_E_BADARG;
    for (int q = 0; q < num_layers; ++q)
        if (!w[q] || !y[q]) return CONAN_E_BADARG;
    if (M == 0) return CONAN_OK;
    int rc = CONAN_OK;
    if (conan_linear_t_multi(x, w, bias, M, K, N, num_layers, act, y, pre, m_dev, as_stream(stream), &rc)) return rc;
    for (int q = 0; q < num_layers; ++q) {                        // shapes outside the one-launch form: one launch per layer
        if (!conan_linear_t_try(x, w[q], bias ? bias[q] : nullptr, nullptr, M, K, N, 0, act, y[q], m_dev, as_stream(stream), &rc, pre ? pre[q] : nullptr))
            return CONAN_E_UNSUPPORTED;
        if (rc != CONAN_OK) return rc;
    }
    return CONAN_OK;
}

int conan_linear_sum_fwd(const float *const *x, const int *ldx, const float *const *w, int num_inputs, int w_kn, const float *bias,
                         const float *residual, int M, int N, const int *m_dev, float *y, void *stream) {
    if (!x || !ldx || !w || !y || M < 0 || N <= 0 || num_inputs < 1) return CONAN_E_BADARG;
    for (int c = 0; c < num_inputs; ++c)
        if (!x[c] || !w[c] || ldx[c] < 128 || (ldx[c] & 3)) return CONAN_E_BADARG;
    if (M == 0) return CONAN_OK;
    int rc = CONAN_OK;
    if (conan_linear_t_sum(x, ldx, w, num_inputs, w_kn, w_kn ? N : 128, bias, residual, M, N, y, m_dev, as_stream(stream), &rc)) return rc;
    return CONAN_E_UNSUPPORTED;
}

int conan_ssp_bwd(const float *dy, const float *y, int rows, int width, const int *m_dev, float *g, void *stream) {
    const long long count = (long long)rows * width;
    if (rows < 0 || width <= 0 || (count && (!dy || !y || !g))) return CONAN_E_BADARG;
    if (!count) return CONAN_OK;
    int blocks = (int)((count + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    k_ssp_bwd<<<blocks, 256, 0, as_stream(stream)>>>(dy, y, count, width, m_dev, g);
    CONAN_LAUNCH_CHECK();
    return CONAN_OK;
}

long long conan_linear_wgrad_ws(int M, int K, int N) {
    return (long long)(wgrad_slices(M, K) + WG_GROUPS) * ((long long)N * K + N);
}

static int wgrad_launch(const float *g, const float *x, int M, int K, int N, const int *m_dev, float *dW, float *dbias, float *ws,
                        hipStream_t s, const float *dist, const float *offset, float coeff, const float *gmax = nullptr) {
    const bool rbf = dist != nullptr;
    const int slices = wgrad_slices(M, K);
    float *slabs = ws, *bias_slabs = ws + (size_t)slices * N * K;
    {
        const int KT = K > 64 ? 128 : 64;
        dim3 grid(slices, (N + 127) / 128, (K + KT - 1) / KT);
        int tn_remap = 0;
        if (grid.y > 1 && grid.z == 1 && (slices & 7) == 0 && !gmax) { tn_remap = (int)grid.y; grid = dim3(slices * tn_remap, 1, 1); }
        if (!rbf && !gmax && K == 128 && N == 256 && M >= CONAN_WGRAD_SHARED_MIN_ROWS) {
            // the two n tiles of one [M,256] gradient over one x: one workgroup per slice stages x once (k_wgrad_lds_shared)
            WgShared<2> P;
            for (int j = 0; j < 2; ++j) { P.g[j] = g + 128 * j; P.slab[j] = slabs + (size_t)128 * j * K; P.bslab[j] = bias_slabs + 128 * j; }
            P.x = x; P.m_dev = m_dev; P.M = M; P.ldg = N; P.slices = slices; P.slab_pitch = (long long)N * K; P.bias_pitch = N;
            launch_wgrad_shared<2>(P, s);
        } else
        if (gmax) {
            if (rbf || KT != 128) return CONAN_E_UNSUPPORTED;
            k_wgrad_lds_h16<<<grid, 256, 0, s>>>(g, x, M, K, N, slabs, bias_slabs, m_dev, gmax);
        } else if (KT == 128) {
            if (rbf) k_wgrad_lds<128, true><<<grid, 256, 0, s>>>(g, x, M, K, N, slabs, bias_slabs, m_dev, dist, offset, coeff, tn_remap, slices);
            else k_wgrad_lds<128, false><<<grid, 256, 0, s>>>(g, x, M, K, N, slabs, bias_slabs, m_dev, dist, offset, coeff, tn_remap, slices);
        } else {
            if (rbf) k_wgrad_lds<64, true><<<grid, 256, 0, s>>>(g, x, M, K, N, slabs, bias_slabs, m_dev, dist, offset, coeff, tn_remap, slices);
            else k_wgrad_lds<64, false><<<grid, 256, 0, s>>>(g, x, M, K, N, slabs, bias_slabs, m_dev, dist, offset, coeff, tn_remap, slices);
        }
    }
    const int NK = N * K;
    if (!dW) {                                        // slabs only: the caller reduces them later (conan_wgrad_reduce_batch)
        CONAN_LAUNCH_CHECK();
        return CONAN_OK;
    }
    if ((NK & 3) == 0 && (N & 3) == 0) {
        const int cols4 = NK / 4 + (dbias ? N / 4 : 0);
        k_wgrad_reduce4<<<(cols4 + 31) / 32, 256, 0, s>>>(reinterpret_cast<const float4 *>(slabs), reinterpret_cast<const float4 *>(bias_slabs), slices,
                                                         NK / 4, N / 4, reinterpret_cast<float4 *>(dW), reinterpret_cast<float4 *>(dbias));
    } else if (slices > 2 * WG_GROUPS) {             // two-level, both levels in a fixed order
        float *mid = bias_slabs + (size_t)slices * N, *bmid = mid + (size_t)WG_GROUPS * NK;
        const int per = (slices + WG_GROUPS - 1) / WG_GROUPS;
        k_wgrad_reduce<<<dim3((NK + 255) / 256, WG_GROUPS), 256, 0, s>>>(slabs, bias_slabs, slices, per, NK, N, mid, dbias ? bmid : nullptr);
        k_wgrad_reduce<<<dim3((NK + 255) / 256, 1), 256, 0, s>>>(mid, bmid, WG_GROUPS, WG_GROUPS, NK, N, dW, dbias);
    } else {
        k_wgrad_reduce<<<dim3((NK + 255) / 256, 1), 256, 0, s>>>(slabs, bias_slabs, slices, slices, NK, N, dW, dbias);
    }
    CONAN_LAUNCH_CHECK();
    return CONAN_OK;
}

int conan_linear_wgrad(const float *g, const float *x, int M, int K, int N, const int *m_dev, float *dW, float *dbias,
                       float *ws, void *stream) {
    if (!g || !x || !dW || !ws || M < 0 || K <= 0 || N <= 0) return CONAN_E_BADARG;
    return wgrad_launch(g, x, M, K, N, m_dev, dW, dbias, ws, as_stream(stream), nullptr, nullptr, 0.f);
}

int conan_linear_wgrad_scaled(const float *g, const float *x, int M, int K, int N, const int *m_dev, float *dW, float *dbias, float *ws,
                              const float *gmax, void *stream) {
    if (!g || !x || !ws || !gmax || M < 0 || K <= 64 || N <= 0 || (!dW && !conan_wgrad_batchable(K, N))) return CONAN_E_BADARG;
    return wgrad_launch(g, x, M, K, N, m_dev, dW, dbias, ws, as_stream(stream), nullptr, nullptr, 0.f, gmax);
}

int conan_rbf_wgrad(const float *g, const float *dist, int M, const float *offset, int num_gaussians, float coeff, int N,
                    const int *m_dev, float *dW, float *dbias, float *ws, void *stream) {
    if (!g || !dist || !offset || !dW || !ws || M < 0 || num_gaussians <= 0 || N <= 0) return CONAN_E_BADARG;
    return wgrad_launch(g, nullptr, M, num_gaussians, N, m_dev, dW, dbias, ws, as_stream(stream), dist, offset, coeff);
}

}  // extern "C"

// slabs [slices][NK] + bias_slabs [slices][N] produced by another translation unit (filter_bwd.hip) -> dW, dbias; NK, N multiples of 4
int conan_wgrad_reduce_now(const float *slabs, const float *bias_slabs, int slices, int NK, int N, float *dW, float *dbias, hipStream_t s) {
    if ((NK & 3) || (N & 3)) return CONAN_E_UNSUPPORTED;
    const int cols4 = NK / 4 + (dbias ? N / 4 : 0);
    k_wgrad_reduce4<<<(cols4 + 31) / 32, 256, 0, s>>>(reinterpret_cast<const float4 *>(slabs), reinterpret_cast<const float4 *>(bias_slabs), slices,
                                                     NK / 4, N / 4, reinterpret_cast<float4 *>(dW), reinterpret_cast<float4 *>(dbias));
    CONAN_LAUNCH_CHECK();
    return CONAN_OK;
}

extern "C" {

int conan_wgrad_batchable(int K, int N) { return (((long long)N * K) & 3) == 0 && (N & 3) == 0 ? 1 : 0; }

int conan_linear_wgrad_slabs(const float *g, const float *x, int M, int K, int N, const int *m_dev, float *ws, void *stream) {
    if (!g || !x || !ws || M < 0 || K <= 0 || N <= 0 || !conan_wgrad_batchable(K, N)) return CONAN_E_BADARG;
    return wgrad_launch(g, x, M, K, N, m_dev, nullptr, nullptr, ws, as_stream(stream), nullptr, nullptr, 0.f);
}

int conan_rbf_wgrad_slabs(const float *g, const float *dist, int M, const float *offset, int num_gaussians, float coeff, int N,
                          const int *m_dev, float *ws, void *stream) {
    if (!g || !dist || !offset || !ws || M < 0 || num_gaussians <= 0 || N <= 0 || !conan_wgrad_batchable(num_gaussians, N)) return CONAN_E_BADARG;
    return wgrad_launch(g, nullptr, M, num_gaussians, N, m_dev, nullptr, nullptr, ws, as_stream(stream), dist, offset, coeff);
}

int conan_linear_wgrad_slabs_batch(const conan_wgrad_slab_job *jobs, int num_jobs, void *stream) {
    if (num_jobs < 0 || (num_jobs && !jobs)) return CONAN_E_BADARG;
    hipStream_t s = as_stream(stream);
    for (int j = 0; j < num_jobs; ++j) {
        const conan_wgrad_slab_job &b = jobs[j];
        if (!b.g || !b.x || !b.ws || b.M < 0 || b.K <= 0 || b.N <= 0 || b.slices < 0 || !conan_wgrad_batchable(b.K, b.N)) return CONAN_E_BADARG;
    }
    for (int wide = 0; wide < 2; ++wide) {                  // one launch sequence per k-tile width (two template instantiations)
        WgradSlabJobs J;
        J.count = 0; J.first[0] = 0;
        auto flush = [&]() {
            if (!J.count) return;
            if (wide) k_wgrad_lds_batch<128><<<J.first[J.count], 256, 0, s>>>(J);
            else k_wgrad_lds_batch<64><<<J.first[J.count], 256, 0, s>>>(J);
            J.count = 0; J.first[0] = 0;
        };
        for (int j = 0; j < num_jobs; ++j) {
            const conan_wgrad_slab_job &b = jobs[j];
            if ((b.K > 64) != (wide != 0)) continue;              // (a job with M = 0 still runs: its one slice writes zero slabs for the reducer)
            const int KT = wide ? 128 : 64;
            const int dflt = wgrad_slices(b.M, b.K), slices = (b.slices > 0 && b.slices <= dflt) ? b.slices : dflt;
            const int tn = (b.N + 127) / 128, tk = (b.K + KT - 1) / KT;
            // how many of the following jobs read the same x with the same shape (a run is kept inside one launch)
            int run = 1;
            if (tn == 1 && tk == 1 && (slices & 7) == 0 && b.M > 0)
                while (j + run < num_jobs && run < 4 && jobs[j + run].x == b.x && jobs[j + run].M == b.M && jobs[j + run].K == b.K && jobs[j + run].N == b.N &&
                       jobs[j + run].m_dev == b.m_dev && jobs[j + run].slices == b.slices)
                    ++run;
            if ((run == 2 || run == 3) && wide && b.K == 128 && b.N == 128 && b.M >= CONAN_WGRAD_SHARED_MIN_ROWS) {
                // an edge-level run over one x: one workgroup per slice for all of its jobs (k_wgrad_lds_shared; same slabs bit for bit)
                auto fill = [&](auto &P) {
                    for (int r = 0; r < run; ++r) { P.g[r] = jobs[j + r].g; P.slab[r] = jobs[j + r].ws; P.bslab[r] = jobs[j + r].ws + (size_t)slices * b.N * b.K; }
                    P.x = b.x; P.m_dev = b.m_dev; P.M = b.M; P.ldg = b.N; P.slices = slices; P.slab_pitch = (long long)b.N * b.K; P.bias_pitch = b.N;
                };
                if (run == 2) { WgShared<2> P; fill(P); launch_wgrad_shared<2>(P, s); }
                else { WgShared<3> P; fill(P); launch_wgrad_shared<3>(P, s); }
                j += run - 1;
                continue;
            }
            if (J.count + run > WGS_BATCH) flush();
            for (int r = 0; r < run; ++r) {
                const conan_wgrad_slab_job &c = jobs[j + r];
                const int q = J.count;
                J.g[q] = c.g; J.x[q] = c.x; J.m_dev[q] = c.m_dev;
                J.slabs[q] = c.ws; J.bias_slabs[q] = c.ws + (size_t)slices * c.N * c.K;
                J.M[q] = c.M; J.K[q] = c.K; J.N[q] = c.N; J.slices[q] = slices; J.tiles_n[q] = tn;
                J.gsz[q] = (run > 1 && r == run - 1) ? run : 1;
                J.first[q + 1] = J.first[q] + ((run > 1 && r < run - 1) ? 0 : run * slices * tn * tk);
                ++J.count;
            }
            j += run - 1;
            if (J.count == WGS_BATCH) flush();
        }
        flush();
    }
    CONAN_LAUNCH_CHECK();
    return CONAN_OK;
}

int conan_wgrad_reduce_batch(const conan_wgrad_job *jobs, int num_jobs, void *stream) {
    if (num_jobs < 0 || (num_jobs && !jobs)) return CONAN_E_BADARG;
    hipStream_t s = as_stream(stream);
    for (int j0 = 0; j0 < num_jobs; j0 += WG_BATCH) {
        WgradJobs J;
        const int nb = num_jobs - j0 < WG_BATCH ? num_jobs - j0 : WG_BATCH;
        int max_cols4 = 0;
        for (int q = 0; q < nb; ++q) {
            const conan_wgrad_job &b = jobs[j0 + q];
            if (!b.ws || !b.dW || b.M < 0 || b.K <= 0 || b.N <= 0 || !conan_wgrad_batchable(b.K, b.N)) return CONAN_E_BADARG;
            const int slices = b.slices > 0 ? b.slices : wgrad_slices(b.M, b.K);
            const int NK = b.N * b.K;
            J.slabs[q] = reinterpret_cast<const float4 *>(b.ws);
            J.bias_slabs[q] = reinterpret_cast<const float4 *>(b.ws + (size_t)slices * NK);
            J.out[q] = reinterpret_cast<float4 *>(b.dW);
            J.bout[q] = reinterpret_cast<float4 *>(b.dbias);
            J.slices[q] = slices; J.NK4[q] = NK / 4; J.N4[q] = b.N / 4;
            const int cols4 = NK / 4 + (b.dbias ? b.N / 4 : 0);
            if (cols4 > max_cols4) max_cols4 = cols4;
        }
        for (int q = nb; q < WG_BATCH; ++q) { J.slabs[q] = J.bias_slabs[q] = nullptr; J.out[q] = J.bout[q] = nullptr; J.slices[q] = J.NK4[q] = J.N4[q] = 0; }
        k_wgrad_reduce4_batch<<<dim3((max_cols4 + 31) / 32, nb), 256, 0, s>>>(J);
    }
    CONAN_LAUNCH_CHECK();
    return CONAN_OK;
}

}  // extern "C"

