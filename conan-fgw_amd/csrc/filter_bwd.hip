// Backward of the filter network  W = mlp2(ssp(mlp0(rbf(d))))  below its second Linear, in ONE pass over the edge rows:
//
//     dh1 = (g @ w2) * ssp'(h1)           (input gradient of mlp.2 times the activation derivative, from the saved h1)
//     dw1 = dh1^T @ rbf(d),  db1 = colsum(dh1)      (weight / bias gradient of mlp.0; rbf regenerated from d)
//
// The composed form wrote dh1 [P,128] to HBM (k_linear_t16<128,128,2>) and read it back in the weight-gradient kernel
// (k_wgrad_lds<64,true>): 2 x 132 MB per interaction at cfg2 that nobody else consumes.  Here dh1 never leaves registers.
//
// Mapping.  A wavefront owns 32 edge rows at a time.  The dx GEMM is oriented D[e][k] = sum_n g[e][n] w2[n][k] (A = g rows
// straight from global memory, split into three bf16 planes in registers; B = the three bf16 images of w2 in LDS), so a
// lane of the 32x32 accumulator block holds ONE channel k and 16 edge rows: e = (r&3) + 8(r>>2) + 4h for register r.  Eight
// consecutive registers are then exactly an A fragment of the next product  dw1[k][j] += sum_e dh1[e][k] rbf[e][j]  (the
// contraction index e may be permuted freely as long as both operands use the same order) — no transposition through LDS.
// The rbf B fragments are generated for that edge order from the wave's 32 distances.  Column 63 of the rbf operand is the
// constant 1, so dw1[:, 63] accumulates db1 for free (needs num_gaussians <= 63).
//
// Registers: the per-wave partial dw1 [128 x 64] is 128 accumulator registers, the dx block another 64; the kernel runs one
// workgroup of 4 waves per CU (LDS: 102 KB of w2 images) = one wave per SIMD with the whole 512-entry register file, and
// hides memory latency by loading the next tile's g rows and this tile's h1 rows before the MFMA chain of the current tile.
// At the end the four waves add their partial sums in LDS (fixed order) and the workgroup writes one slab; the slabs of all
// workgroups are reduced by the batched reducer of gemm.hip (bitwise reproducible, no float atomics).
//
// Measured at cfg2 (259 k rows): 94 us against 106 + 71 us for the two kernels it replaces (264 MB instead of 704 MB of HBM
// traffic).  rocprofv3 counters (2.17 GHz under this load): 2.33 M MFMAs = 35 % of the SIMD cycles, 18.2 M VALU instructions
// = 34 % — more than half of them the bf16 splitting of g, dh1 and rbf — and only 14 % of the MFMA cycles have VALU work
// running beside them: the kernel is bound by the SUM of the two pipes, not by HBM (floor 61 us).  A variant with two waves per
// tile (half the channel blocks each, 96 accumulator registers, 2 waves per SIMD) overlapped more (31 %) but issued 24 M VALU
// instructions for the duplicated splitting and measured the same 95-98 us.
#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
constexpr int FB_THREADS = 256, FB_WAVES = 4;
constexpr int FB_GRID_MAX = 256;

__device__ __forceinline__ void fb_split3(const float *v, bf16x8 &p1, bf16x8 &p2, bf16x8 &p3) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const __bf16 h1 = (__bf16)v[j];
        const float r1 = v[j] - (float)h1;
        const __bf16 h2 = (__bf16)r1;
        const float r2 = r1 - (float)h2;
        p1[j] = h1; p2[j] = h2; p3[j] = (__bf16)r2;
    }
}

// H16 (round 3): both products on TWO fp16 planes per operand (p1q1 + p1q2 + p2q1: 3 MFMAs instead of the 6 of the three-plane bf16
// form, ~2^-22 relative) — see filter_fused.hip.  fp16's range needs the gradient operand scaled: `gmax` is max |g| over the whole
// tensor (the kernel that produced g tracked it, conan_cfconv_bwd_w_pairs), s = 2^k with s * gmax in [16, 32); A = s * g, the w2
// planes carry their own power-of-two scale (from max |w2|), dh1 is formed as s * dh1 (<= 32 * column abs-sum of w2: far inside 65504) and the accumulated s * dW1 is
// unscaled once, when the slab is written.  Entries below 2e-6 * gmax fall into fp16's subnormal spacing (3e-8 / s absolute): they
// cannot matter to a sum dominated by entries 1e6 times larger.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void fb_split2h(const float *v, float sc, f16x8 &p1, f16x8 &p2) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float x = v[j] * sc;
        const _Float16 h1 = (_Float16)x;
        p1[j] = h1; p2[j] = (_Float16)(x - (float)h1);
    }
}
// plane scale of w2 (see filter_fused.hip: 2^k with max |w| * 2^k in [256, 512)); its inverse is applied where dh1 is formed
__device__ __forceinline__ void fb_plane_scale(float amax, float &sc, float &un) {
    sc = 1.0f; un = 1.0f;
    if (amax > 0.f && amax < 3.0e38f) { int e; (void)frexpf(amax, &e); sc = ldexpf(1.0f, 9 - e); un = ldexpf(1.0f, e - 9); }
}

template <int F, bool H16>
__global__ void __launch_bounds__(FB_THREADS) k_filter_bwd(const float *__restrict__ g, const float *__restrict__ h1,
                                                           const float *__restrict__ dist, const float *__restrict__ offset, int Gs,
                                                           float coeff, const float *__restrict__ w2, int M,
                                                           const int *__restrict__ m_dev, float *__restrict__ slabs,
                                                           float *__restrict__ bias_slabs, const float *__restrict__ gmax) {
    constexpr int NB = F / 32;            // 32-wide blocks of the channel dimension
    constexpr int S = F / 16;             // MFMA k-steps of the dx GEMM
    constexpr int WS = F + 8;             // LDS pitch of a w2 image row (bf16 elements)
    constexpr int JP = 64;                // Gaussians padded to two 32-wide blocks; column 63 = bias
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int NPL = H16 ? 2 : 3;                            // operand planes
    __bf16 *WB = reinterpret_cast<__bf16 *>(lds);               // [NPL][k][WS]: image row k holds w2[n][k] for n = 0..F-1
    _Float16 *WH = reinterpret_cast<_Float16 *>(lds);
    float *DL = lds + (NPL * F * WS) / 2;                       // [FB_WAVES][32] distances of the wave's tile
    __shared__ float wred[FB_WAVES];
    float gsc = 1.0f, gun = 1.0f, wsc = 1.0f, wun = 1.0f;       // H16: scales of the gradient operand and of the w2 planes, and their inverses
    if (m_dev) M = min(M, *m_dev);
    const int tiles = (M + 31) >> 5;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, h = lane >> 5;

    // ---- stage the three bf16 images of w2, transposed: a thread owns a 4(n) x 4(k) block (see gemm_t.hip) -------------
    {
        constexpr int PATCHES = (F / 16) * (F / 64), PERW = (PATCHES + FB_WAVES - 1) / FB_WAVES;
        float4 wv[PERW][4];
        const int n4l = (lane & 3) | ((lane >> 4) << 2), k4l = (lane >> 2) & 3;
#pragma unroll
        for (int u = 0; u < PERW; ++u) {
            const int pt = wave + u * FB_WAVES;
            const int r0 = (pt / (F / 64)) * 16 + 4 * k4l, c0 = (pt % (F / 64)) * 64 + 4 * n4l;       // w2 rows r0.., columns c0..
#pragma unroll
            for (int j = 0; j < 4; ++j)
                wv[u][j] = pt < PATCHES ? *reinterpret_cast<const float4 *>(w2 + (size_t)(r0 + j) * F + c0) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        if constexpr (H16) {
            float am = 0.f;
#pragma unroll
            for (int u = 0; u < PERW; ++u)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    am = fmaxf(fmaxf(am, fmaxf(fabsf(wv[u][j].x), fabsf(wv[u][j].y))), fmaxf(fabsf(wv[u][j].z), fabsf(wv[u][j].w)));
            am = wave_max(am);
            if (lane == 0) wred[wave] = am;
            __syncthreads();
            const float wmax = fmaxf(fmaxf(wred[0], wred[1]), fmaxf(wred[2], wred[3]));
            fb_plane_scale(wmax, wsc, wun);
            // gradient scale: s * gmax in [16, 32), lowered when the weights are so large that s * dh1 (<= 32 * F * max |w2|) could leave fp16
            const float gm = *gmax;
            if (gm > 0.f && gm < 3.0e38f) {
                int e; (void)frexpf(gm, &e);
                int sh = 5 - e;
                const float bound = 32.0f * F * wmax;                   // upper bound of |s * dh1| at the nominal scale
                if (bound > 16384.0f && bound < 3.0e38f) { int eb; (void)frexpf(bound * (1.0f / 16384.0f), &eb); sh -= eb; }
                gsc = ldexpf(1.0f, sh); gun = ldexpf(1.0f, -sh);
            }
        }
#pragma unroll
        for (int u = 0; u < PERW; ++u) {
            const int pt = wave + u * FB_WAVES;
            if (pt >= PATCHES) continue;
            const int r0 = (pt / (F / 64)) * 16 + 4 * k4l, c0 = (pt % (F / 64)) * 64 + 4 * n4l;
#pragma unroll
            for (int e = 0; e < 4; ++e) {                     // image row c0 + e: elements r0 .. r0 + 3
                const float v4[4] = {e == 0 ? wv[u][0].x : e == 1 ? wv[u][0].y : e == 2 ? wv[u][0].z : wv[u][0].w,
                                     e == 0 ? wv[u][1].x : e == 1 ? wv[u][1].y : e == 2 ? wv[u][1].z : wv[u][1].w,
                                     e == 0 ? wv[u][2].x : e == 1 ? wv[u][2].y : e == 2 ? wv[u][2].z : wv[u][2].w,
                                     e == 0 ? wv[u][3].x : e == 1 ? wv[u][3].y : e == 2 ? wv[u][3].z : wv[u][3].w};
                if constexpr (H16) {
                    f16x4 q1, q2;
#pragma unroll
                    for (int j = 0; j < 4; ++j) { const float x = v4[j] * wsc; q1[j] = (_Float16)x; q2[j] = (_Float16)(x - (float)q1[j]); }
                    *reinterpret_cast<f16x4 *>(&WH[(0 * F + c0 + e) * WS + r0]) = q1;
                    *reinterpret_cast<f16x4 *>(&WH[(1 * F + c0 + e) * WS + r0]) = q2;
                    continue;
                }
                bf16x4 q1, q2, q3;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    q1[j] = (__bf16)v4[j]; const float r1 = v4[j] - (float)q1[j];
                    q2[j] = (__bf16)r1; q3[j] = (__bf16)(r1 - (float)q2[j]);
                }
                *reinterpret_cast<bf16x4 *>(&WB[(0 * F + c0 + e) * WS + r0]) = q1;
                *reinterpret_cast<bf16x4 *>(&WB[(1 * F + c0 + e) * WS + r0]) = q2;
                *reinterpret_cast<bf16x4 *>(&WB[(2 * F + c0 + e) * WS + r0]) = q3;
            }
        }
    }
    __syncthreads();

    // Gaussian centres of this lane's two rbf columns; column 63 is the bias column, columns Gs..62 are dead.
    float mu[2];
    int jkind[2];                                              // 0 = Gaussian, 1 = zero, 2 = one
#pragma unroll
    for (int jb = 0; jb < 2; ++jb) {
        const int j = 32 * jb + l31;
        jkind[jb] = j < Gs ? 0 : (j == JP - 1 ? 2 : 1);
        mu[jb] = j < Gs ? offset[j] : 0.f;
    }

    f32x16 dwacc[NB][2];
#pragma unroll
    for (int kb = 0; kb < NB; ++kb)
#pragma unroll
        for (int jb = 0; jb < 2; ++jb)
#pragma unroll
            for (int r = 0; r < 16; ++r) dwacc[kb][jb][r] = 0.f;

    float *dl = DL + 32 * wave;
    const int wave_stride = gridDim.x * FB_WAVES;
    // Rows are walked from the END: the weight-gradient kernel that ran just before streamed g and h1 (264 MB) front to back through the
    // 256 MiB Infinity Cache, so their tails are what it still holds — read in the same direction every line would be evicted just before it
    // is needed (step -0.06 ms in a same-job A/B).  Logical tile t is physical tile tiles - 1 - t.
#define FB_PHYS(t) max(tiles - 1 - (t), 0)
    int tile = blockIdx.x * FB_WAVES + wave;
    float4 xa[S], xb[S];
    auto load_x = [&](int t) {
        const int m = min((FB_PHYS(t) << 5) + l31, M - 1);
        const float *xr = g + (size_t)m * F + 8 * h;
#pragma unroll
        for (int s = 0; s < S; ++s) {                          // lane-half h owns n = 16s + 8h .. +7 of its row
            xa[s] = *reinterpret_cast<const float4 *>(xr + 16 * s);
            xb[s] = *reinterpret_cast<const float4 *>(xr + 16 * s + 4);
        }
    };
    if (tile < tiles) load_x(tile);
    for (; tile < tiles; tile += wave_stride) {
        const int e0 = FB_PHYS(tile) << 5;
        // this tile's distances -> LDS (wave-private; LDS operations of one wave execute in order)
        if (lane < 32) dl[lane] = dist[min(e0 + lane, M - 1)];
        // this tile's h1 rows in accumulator layout: register r of block kb <-> edge row e0 + (r&3) + 8(r>>2) + 4h, channel 32kb + l31
        float hv[NB][16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int e = min(e0 + (r & 3) + 8 * (r >> 2) + 4 * h, M - 1);
            const float *hr = h1 + (size_t)e * F + l31;
#pragma unroll
            for (int kb = 0; kb < NB; ++kb) hv[kb][r] = hr[32 * kb];
        }
        // ---- dx GEMM: acc[kb][r] = sum_n g[e][n] w2[n][32kb + l31] ---------------------------------------------------
        f32x16 acc[NB];
#pragma unroll
        for (int kb = 0; kb < NB; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[kb][r] = 0.f;
        const bool has_next = tile + wave_stride < tiles;
        const float *xn = g + (size_t)min((FB_PHYS(tile + wave_stride) << 5) + l31, M - 1) * F + 8 * h;
        if constexpr (H16) {
            // ---- both products on two fp16 planes (three MFMAs per product, issued over the four channel blocks: independent chains)
            f16x8 qa[2][2];                                        // [double buffer][plane] of the A fragment (s * g)
            auto split_xh = [&](f16x8 (&dst)[2], int s) {
                const float xv[8] = {xa[s].x, xa[s].y, xa[s].z, xa[s].w, xb[s].x, xb[s].y, xb[s].z, xb[s].w};
                fb_split2h(xv, gsc, dst[0], dst[1]);
                if (has_next) {                                    // this k-step's slice of g is consumed: reload it for the next tile
                    xa[s] = *reinterpret_cast<const float4 *>(xn + 16 * s);
                    xb[s] = *reinterpret_cast<const float4 *>(xn + 16 * s + 4);
                }
            };
            split_xh(qa[0], 0);
#pragma unroll
            for (int s = 0; s < S; ++s) {
                f16x8 p[NB][2];
#pragma unroll
                for (int kb = 0; kb < NB; ++kb)
#pragma unroll
                    for (int pl = 0; pl < 2; ++pl)
                        p[kb][pl] = *reinterpret_cast<const f16x8 *>(&WH[(pl * F + 32 * kb + l31) * WS + 16 * s + 8 * h]);
                if (s + 1 < S) split_xh(qa[(s + 1) & 1], s + 1);
#pragma unroll
                for (int kb = 0; kb < NB; ++kb) acc[kb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(qa[s & 1][1], p[kb][0], acc[kb], 0, 0, 0);
#pragma unroll
                for (int kb = 0; kb < NB; ++kb) acc[kb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(qa[s & 1][0], p[kb][1], acc[kb], 0, 0, 0);
#pragma unroll
                for (int kb = 0; kb < NB; ++kb) acc[kb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(qa[s & 1][0], p[kb][0], acc[kb], 0, 0, 0);
            }
            // rbf B fragments: [e-step s2][column block jb][plane]
            f16x8 rbh[2][2][2];
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const float4 da = *reinterpret_cast<const float4 *>(dl + 16 * s2 + 4 * h);
                const float4 db = *reinterpret_cast<const float4 *>(dl + 16 * s2 + 4 * h + 8);
                const float dv[8] = {da.x, da.y, da.z, da.w, db.x, db.y, db.z, db.w};
#pragma unroll
                for (int jb = 0; jb < 2; ++jb) {
                    float rv[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const float t = dv[j] - mu[jb];
                        const float ex = exp_neg_f(coeff * (t * t));
                        rv[j] = jkind[jb] == 0 ? ex : (jkind[jb] == 2 ? 1.0f : 0.0f);
                    }
                    fb_split2h(rv, 1.0f, rbh[s2][jb][0], rbh[s2][jb][1]);
                }
            }
            // epilogue per channel block: s * dh1 = acc / 2^6 * ssp'(h1), then s * dw1[32kb.., :] += (s * dh1)^T rbf
#pragma unroll
            for (int kb = 0; kb < NB; ++kb) {
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    float v[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const int r = 8 * s2 + j;
                        const bool ok = e0 + (r & 3) + 8 * (r >> 2) + 4 * h < M;
                        const float d = acc[kb][r] * wun * (1.0f - 0.5f * __expf(-hv[kb][r]));      // ssp'(pre) from the saved output
                        v[j] = ok ? d : 0.f;
                    }
                    f16x8 a1, a2;
                    fb_split2h(v, 1.0f, a1, a2);
#pragma unroll
                    for (int jb = 0; jb < 2; ++jb) dwacc[kb][jb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a2, rbh[s2][jb][0], dwacc[kb][jb], 0, 0, 0);
#pragma unroll
                    for (int jb = 0; jb < 2; ++jb) dwacc[kb][jb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, rbh[s2][jb][1], dwacc[kb][jb], 0, 0, 0);
#pragma unroll
                    for (int jb = 0; jb < 2; ++jb) dwacc[kb][jb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, rbh[s2][jb][0], dwacc[kb][jb], 0, 0, 0);
                }
            }
            continue;
        }
        // MFMAs on ONE accumulator wait for each other (16 passes each): with a single wave per SIMD nothing else fills the
        // gaps, so the six partial products are issued round-robin over two channel blocks (two independent chains).
        constexpr int PQ[6] = {0, 1, 2, 0, 1, 0}, PP[6] = {2, 1, 0, 1, 0, 0};
        // Software pipeline over the 2 * S groups (k-step s, channel-block pair): the w2 fragments of group i + 1 are read from
        // LDS and (at the first group of a k-step) the next k-step's slice of g is split while the 12 MFMAs of group i run —
        // there is one wave per SIMD, nobody else covers an LDS round trip or a VALU burst.
        bf16x8 q[2][3], p[2][2][3];
        auto read_w = [&](bf16x8 (&dst)[2][3], int grp) {
            const int s = grp >> 1, kb = (grp & 1) * 2;
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl)
                    dst[u][pl] = *reinterpret_cast<const bf16x8 *>(&WB[(pl * F + 32 * (kb + u) + l31) * WS + 16 * s + 8 * h]);
        };
        auto split_x = [&](bf16x8 (&dst)[3], int s) {
            const float xv[8] = {xa[s].x, xa[s].y, xa[s].z, xa[s].w, xb[s].x, xb[s].y, xb[s].z, xb[s].w};
            fb_split3(xv, dst[0], dst[1], dst[2]);
            if (has_next) {                                    // this k-step's slice of g is consumed: reload it for the next tile
                xa[s] = *reinterpret_cast<const float4 *>(xn + 16 * s);
                xb[s] = *reinterpret_cast<const float4 *>(xn + 16 * s + 4);
            }
        };
        split_x(q[0], 0);
        read_w(p[0], 0);
#pragma unroll
        for (int grp = 0; grp < 2 * S; ++grp) {
            const int s = grp >> 1, kb = (grp & 1) * 2;
            if (grp + 1 < 2 * S) read_w(p[(grp + 1) & 1], grp + 1);
            if ((grp & 1) == 0 && s + 1 < S) split_x(q[(s + 1) & 1], s + 1);
#pragma unroll
            for (int t = 0; t < 6; ++t)
#pragma unroll
                for (int u = 0; u < 2; ++u)
                    acc[kb + u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(q[s & 1][PQ[t]], p[grp & 1][u][PP[t]], acc[kb + u], 0, 0, 0);
            // per group: 12 MFMAs, 6 LDS reads, ~25 VALU (half of a split)  ->  { 1 DS read, 4 VALU, 2 MFMA } x 6
#pragma unroll
            for (int z = 0; z < 6; ++z) {
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 8, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
            }
        }
        // ---- rbf B fragments of the tile: [e-step s2][column block jb], lane (column 32jb + l31, edge group h) ---------------
        bf16x8 rb[2][2][3];
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            const float4 da = *reinterpret_cast<const float4 *>(dl + 16 * s2 + 4 * h);
            const float4 db = *reinterpret_cast<const float4 *>(dl + 16 * s2 + 4 * h + 8);
            const float dv[8] = {da.x, da.y, da.z, da.w, db.x, db.y, db.z, db.w};
#pragma unroll
            for (int jb = 0; jb < 2; ++jb) {
                float rv[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float t = dv[j] - mu[jb];
                    const float ex = exp_neg_f(coeff * (t * t));
                    rv[j] = jkind[jb] == 0 ? ex : (jkind[jb] == 2 ? 1.0f : 0.0f);
                }
                fb_split3(rv, rb[s2][jb][0], rb[s2][jb][1], rb[s2][jb][2]);
            }
        }
        constexpr int PP2[6] = {0, 1, 2, 0, 1, 0};             // (a3,b1) (a2,b2) (a1,b3) (a2,b1) (a1,b2) (a1,b1)
        // ---- epilogue per channel block: dh1 = acc * ssp'(h1), then dw1[32kb.., :] += dh1^T rbf ----------------------------
#pragma unroll
        for (int kb = 0; kb < NB; ++kb) {
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                float v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int r = 8 * s2 + j;
                    const bool ok = e0 + (r & 3) + 8 * (r >> 2) + 4 * h < M;
                    const float d = acc[kb][r] * (1.0f - 0.5f * __expf(-hv[kb][r]));      // ssp'(pre) from the saved output
                    v[j] = ok ? d : 0.f;
                }
                bf16x8 a1, a2, a3;
                fb_split3(v, a1, a2, a3);
#pragma unroll
                for (int t = 0; t < 6; ++t)
#pragma unroll
                    for (int jb = 0; jb < 2; ++jb)
                        dwacc[kb][jb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(t < 3 ? (t == 0 ? a3 : t == 1 ? a2 : a1) : (t == 3 ? a2 : a1),
                                                                                rb[s2][jb][PP2[t]], dwacc[kb][jb], 0, 0, 0);
            }
        }
    }

    // ---- the four waves' partial sums -> one slab per workgroup (waves added in the fixed order 0,1,2,3) -------------------
    __syncthreads();                                           // every wave is done with the w2 images
    float *RED = lds;                                          // [F][JP]
    for (int w = 0; w < FB_WAVES; ++w) {
        if (wave == w) {
#pragma unroll
            for (int kb = 0; kb < NB; ++kb)
#pragma unroll
                for (int jb = 0; jb < 2; ++jb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int k = 32 * kb + (r & 3) + 8 * (r >> 2) + 4 * h;
                        float *p = &RED[k * JP + 32 * jb + l31];
                        const float val = dwacc[kb][jb][r] * gun;          // (H16: the accumulators carry the gradient's scale; 1 otherwise)
                        *p = w == 0 ? val : *p + val;
                    }
        }
        __syncthreads();
    }
    float *slab = slabs + (size_t)blockIdx.x * F * Gs;
    for (int idx = tid; idx < F * Gs; idx += FB_THREADS) {
        const int k = idx / Gs, j = idx - k * Gs;
        slab[idx] = RED[k * JP + j];
    }
    if (tid < F) bias_slabs[(size_t)blockIdx.x * F + tid] = RED[tid * JP + JP - 1];
}


}  // namespace

int conan_wgrad_reduce_now(const float *slabs, const float *bias_slabs, int slices, int NK, int N, float *dW, float *dbias, hipStream_t s);   // gemm.hip

extern "C" {

int conan_filter_bwd_supported(int num_gaussians, int num_filters) { return num_filters == 128 && num_gaussians >= 1 && num_gaussians <= 63; }

int conan_filter_bwd_slices(int M) {
    const int tiles = (M + 31) / 32;
    int grid = (tiles + FB_WAVES - 1) / FB_WAVES;
    if (grid < 1) grid = 1;
    return grid > FB_GRID_MAX ? FB_GRID_MAX : grid;
}

long long conan_filter_bwd_ws(int M, int num_gaussians, int num_filters) {
    return (long long)conan_filter_bwd_slices(M) * ((long long)num_filters * num_gaussians + num_filters);
}

int conan_filter_bwd(const float *g, const float *h1, const float *dist, int M, const float *offset, int num_gaussians, float coeff,
                     const float *w2, int num_filters, const int *m_dev, float *dW1, float *db1, float *ws, const float *gmax, void *stream) {
    if (!g || !h1 || !dist || !offset || !w2 || !ws || M < 1) return CONAN_E_BADARG;
    if (!conan_filter_bwd_supported(num_gaussians, num_filters)) return CONAN_E_UNSUPPORTED;
    const int F = num_filters, Gs = num_gaussians, slices = conan_filter_bwd_slices(M);
    hipStream_t s = as_stream(stream);
    float *slabs = ws, *bias_slabs = ws + (size_t)slices * F * Gs;
    if (gmax) {       // two fp16 planes, gradient scaled from its maximum
        const size_t lds = ((size_t)(2 * F * (F + 8)) / 2 + FB_WAVES * 32) * 4;
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_filter_bwd<128, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        k_filter_bwd<128, true><<<slices, FB_THREADS, lds, s>>>(g, h1, dist, offset, Gs, coeff, w2, M, m_dev, slabs, bias_slabs, gmax);
    } else {
        const size_t lds = ((size_t)(3 * F * (F + 8)) / 2 + FB_WAVES * 32) * 4;
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_filter_bwd<128, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        k_filter_bwd<128, false><<<slices, FB_THREADS, lds, s>>>(g, h1, dist, offset, Gs, coeff, w2, M, m_dev, slabs, bias_slabs, nullptr);
    }
    CONAN_LAUNCH_CHECK();
    if (!dW1) return CONAN_OK;                        // slabs only: reduced later by conan_wgrad_reduce_batch (job.slices = conan_filter_bwd_slices(M))
    return conan_wgrad_reduce_now(slabs, bias_slabs, slices, F * Gs, F, dW1, db1, s);
}

}  // extern "C"
