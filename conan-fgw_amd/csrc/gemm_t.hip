// Register-streamed linear layer for gfx950:  y = act(x @ W^T + b) (+ residual)  for K, N in {64, 128}.
//
// Same transposed mapping as the fused filter kernel (filter_fused.hip): the ROW index of x lives on the MFMA column
// (= lane), the output feature on the MFMA row.  A wavefront owns 32 rows of x; lane (m, h) reads its own row straight
// from global memory as K/8 float4 (the two lane-halves split each group of 8 k's), which are exactly the B operands of
// v_mfma_f32_32x32x2_f32 — x never touches LDS, there is no barrier after the weights are staged, and every wavefront
// keeps K/8 independent 16-B loads in flight.  W ([N][K+4] in LDS, <= 68 KB) is read as conflict-free ds_read_b128 A
// fragments, software-pipelined one group ahead.  The accumulators (D layout: row = feature, column = x row) are
// stored / combined with `residual` as float4 per lane.
#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
constexpr int LT_THREADS = 512;
#ifndef CONAN_LINEAR_MAX_WGS
#define CONAN_LINEAR_MAX_WGS 256                    // persistent workgroups of an edge-level launch (one per CU: 137-141 KB of LDS each); a multiple of 16
#endif


// Two-plane fp16 form (default; see filter_fused.hip): v = h1 + h2 with the three products p1q1, p1q2, p2q1 on v_mfma_f32_32x32x16_f16 —
// half the matrix-pipe time and 2/3 of the splitting work of the three-plane bf16 form.  fp16's narrow exponent range is met by exact
// power-of-two scales: the weight by one factor per matrix (block maximum at staging time -> max |w| in [256, 512)), every x ROW by its
// own factor (the row lives on one lane pair, so its maximum costs one cross-half exchange) — both undone by one multiply per output in
// the epilogue (the D column is the x row, so the row factor is a per-lane scalar).  Inside a row, elements down to 2^-12 of the row
// maximum keep 22 bits; below that the absolute error is 3e-8 / 256 of the row maximum.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
constexpr int LT_NPL = 2;                                      // operand planes
constexpr int LT_OP = 68;                                      // pitch of the per-wave output slab (floats): 64 channels + 4
__device__ __forceinline__ void split2h(const float *v, float sc, f16x8 &p1, f16x8 &p2) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float t = v[j] * sc;
        const _Float16 h1 = (_Float16)t;
        p1[j] = h1; p2[j] = (_Float16)(t - (float)h1);
    }
}
// 2^k with amax * 2^k in [256, 512) and its inverse, from the exponent field (zero / subnormal / non-finite maxima: 1)
__device__ __forceinline__ void pow2_scale(float amax, float &sc, float &un) {
    const int e = (int)((__float_as_uint(amax) >> 23) & 0xffu);
    const bool ok = e >= 9 && e <= 254;
    sc = ok ? __uint_as_float((unsigned)(262 - e) << 23) : 1.0f;
    un = ok ? __uint_as_float((unsigned)(e - 8) << 23) : 1.0f;
}
template <int NT>
__device__ __forceinline__ float lt_block_absmax(float v, float *red) {      // red: NT / 64 floats of LDS; every thread gets the maximum
    v = wave_max(v);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float m = red[0];
#pragma unroll
    for (int w = 1; w < NT / 64; ++w) m = fmaxf(m, red[w]);
    return m;
}
// four consecutive k of image row n: 8-byte stores, one per plane
__device__ __forceinline__ void lt_store4(void *WBv, int N, int WS, int n, int k, const float *v4, float sc) {
    _Float16 *WB = reinterpret_cast<_Float16 *>(WBv);
    f16x4 h1, h2;
#pragma unroll
    for (int e = 0; e < 4; ++e) { const float t = v4[e] * sc; h1[e] = (_Float16)t; h2[e] = (_Float16)(t - (float)h1[e]); }
    *reinterpret_cast<f16x4 *>(&WB[(0 * N + n) * WS + k]) = h1;
    *reinterpret_cast<f16x4 *>(&WB[(1 * N + n) * WS + k]) = h2;
}

// SPLIT variant: both operands are split into two fp16 planes (above) and the product is formed from the three significant
// partial products on v_mfma_f32_32x32x16_f16 (fp32-class accuracy).  W's two planes are
// staged once per workgroup; x is split in registers after the global load.
// ACT is a template parameter: the epilogue is straight-line code over 16*NB outputs per lane, and with a run-time
// activation switch the kernel was 31 KB of code executed exactly once per wavefront at node-level sizes — rocprofv3
// showed 45 % of the wave cycles waiting for instruction fetch (SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES).
// Up to four Linear layers of the SAME input in one launch (q / k / v; dk / dv / f_proj of one f): chunk c of the nc > 1 mechanism below takes its
// weight, bias and outputs from slot c here instead of from an offset into one wide layer.  w[0] == nullptr: not used.
struct LtJobs {
    const float *w[4], *bias[4];
    float *y[4], *pre[4];
};

template <int K, int N, int ACT, int NT>
__global__ void __launch_bounds__(NT) k_linear_t16(const float *__restrict__ x, const float *__restrict__ w,
                                                           const float *__restrict__ bias, const float *__restrict__ residual,
                                                           int M, int w_kn, float *y,
                                                           const int *__restrict__ m_dev, int ldx, int ldw, int ldy,
                                                           const float *accum, float *pre_out, int nc, const LtJobs J) {
    // ldx / ldw / ldy: row pitches of x, w and of y / residual / accum — the launcher tiles wider layers into 64/128-wide
    // (K, N) chunks of one strided problem; `accum` (may alias y) carries the partial sum of the previous K chunks and is
    // added BEFORE the activation.  pre_out (nullable): also store the pre-activation (bias and accum included), which the
    // backward of a SiLU layer needs — one extra store instead of a separate activation kernel re-reading it.
    constexpr int act = ACT;
    constexpr int NB = N / 32;
    constexpr int S = K / 16;             // MFMA k-steps
    constexpr int WS = K + 8;             // LDS pitch (16-bit elements)
    extern __shared__ __attribute__((aligned(16))) float lds[];
    __bf16 *WB = reinterpret_cast<__bf16 *>(lds);               // [planes][N][WS] (fp16 planes; the pointer type is only a 16-bit carrier)
    float *BL = lds + (LT_NPL * N * WS) / 2;                    // [N]
    float *OT = BL + N + (threadIdx.x >> 6) * (32 * LT_OP);    // fp16 form: this wave's [32][LT_OP] output slab
    __shared__ float wred[NT / 64];
    float wun = 1.0f;                                           // inverse scale of the weight planes (fp16 form)
    if (m_dev) M = min(M, *m_dev);
    const int tiles = (M + 31) >> 5;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // nc > 1: the nc N-wide output chunks of ONE wider layer in one launch.  Block b serves chunk b % nc for tile slot b / nc, so the
    // workgroups that read the same x tiles are dispatched side by side and all but the first find the rows in L2 / the Infinity Cache
    // (as separate launches every chunk streamed x from HBM again, and a node-level layer paid launch + staging per chunk).
    const int chunk = nc > 1 ? (int)blockIdx.x % nc : 0, slot = nc > 1 ? (int)blockIdx.x / nc : (int)blockIdx.x;
    const int nslots = nc > 1 ? (int)gridDim.x / nc : (int)gridDim.x;
    if (nc > 1 && J.w[0]) {
        w = J.w[chunk]; bias = J.bias[chunk]; y = J.y[chunk]; pre_out = J.pre[chunk];
    } else if (nc > 1) {
        const int n0 = chunk * N;
        w += w_kn ? (size_t)n0 : (size_t)n0 * ldw;
        if (bias) bias += n0;
        if (residual) residual += n0;
        if (accum) accum += n0;
        if (pre_out) pre_out += n0;
        y += n0;
    }
    if (slot * (NT / 64) >= tiles) return;
    // The wave's first tile of x is requested BEFORE the weights are staged (the two are independent): at node-level sizes a
    // wave owns exactly one tile and the kernel is two serial memory round trips otherwise.  Later tiles are requested as soon
    // as the MFMA loop has consumed the current one, i.e. they fly during the epilogue.
    const int l31 = lane & 31, h = lane >> 5;
    const int wave_stride = nslots * (NT / 64);
    float4 xa[S], xb[S];
    auto load_x = [&](int t) {
        const int mr = min((t << 5) + l31, M - 1);
        const float *xr = x + (size_t)mr * ldx + 8 * h;
#pragma unroll
        for (int s = 0; s < S; ++s) {                          // lane-half h owns k = 16s + 8h .. +7 of its row
            xa[s] = *reinterpret_cast<const float4 *>(xr + 16 * s);
            xb[s] = *reinterpret_cast<const float4 *>(xr + 16 * s + 4);
        }
    };
    if (slot * (NT / 64) + wave < tiles) load_x(slot * (NT / 64) + wave);
    // Staging of the two fp16 planes of W as [n][k]; all of a thread's loads are in flight before the first use.
    if (!w_kn) {                                              // w is [N][K]: float4 = 4 consecutive k -> one 8-byte store per image
        constexpr int V4 = N * K / 4, PER = (V4 + NT - 1) / NT;
        float4 wv[PER];
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int q = tid + u * NT;
            const int n = (4 * q) / K, k = 4 * q - n * K;
            wv[u] = q < V4 ? *reinterpret_cast<const float4 *>(w + (size_t)n * ldw + k) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        float wsc = 1.0f;
        float am = 0.f;
#pragma unroll
        for (int u = 0; u < PER; ++u) am = fmaxf(am, fmaxf(fmaxf(fabsf(wv[u].x), fabsf(wv[u].y)), fmaxf(fabsf(wv[u].z), fabsf(wv[u].w))));
        pow2_scale(lt_block_absmax<NT>(am, wred), wsc, wun);
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int q = tid + u * NT;
            if (q >= V4) continue;
            const float v4[4] = {wv[u].x, wv[u].y, wv[u].z, wv[u].w};
            const int n = (4 * q) / K, k = 4 * q - n * K;
            lt_store4(WB, N, WS, n, k, v4, wsc);
        }
    } else {
        // w is [K][N] (the dx GEMM of the backward reads the forward weight transposed).  A thread owns a 4(k) x 4(n) block:
        // four float4 loads along n, transposed in registers, 8-byte stores along k.  Inside a wavefront the blocks form a
        // 4(k4) x 16(n4) patch with lane = (n4 & 3) | (k4 << 2) | ((n4 >> 2) << 4): every 16-lane group then covers 4 rows x
        // 4 k-blocks = 16 distinct 8-byte bank slots (rows 4 apart sit 64 B apart modulo the 256-B bank cycle).
        constexpr int PATCHES = (K / 16) * (N / 64), PERW = (PATCHES + NT / 64 - 1) / (NT / 64);
        float4 wv[PERW][4];
        const int n4l = (lane & 3) | ((lane >> 4) << 2), k4l = (lane >> 2) & 3;
#pragma unroll
        for (int u = 0; u < PERW; ++u) {
            const int pt = wave + u * (NT / 64);
            const int k0 = (pt / (N / 64)) * 16 + 4 * k4l, n0 = (pt % (N / 64)) * 64 + 4 * n4l;
#pragma unroll
            for (int j = 0; j < 4; ++j)
                wv[u][j] = pt < PATCHES ? *reinterpret_cast<const float4 *>(w + (size_t)(k0 + j) * ldw + n0) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        float wsc = 1.0f;
        float am = 0.f;
#pragma unroll
        for (int u = 0; u < PERW; ++u)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                am = fmaxf(am, fmaxf(fmaxf(fabsf(wv[u][j].x), fabsf(wv[u][j].y)), fmaxf(fabsf(wv[u][j].z), fabsf(wv[u][j].w))));
        pow2_scale(lt_block_absmax<NT>(am, wred), wsc, wun);
#pragma unroll
        for (int u = 0; u < PERW; ++u) {
            const int pt = wave + u * (NT / 64);
            if (pt >= PATCHES) continue;
            const int k0 = (pt / (N / 64)) * 16 + 4 * k4l, n0 = (pt % (N / 64)) * 64 + 4 * n4l;
#pragma unroll
            for (int e = 0; e < 4; ++e) {                     // row n0 + e of the image: k0 .. k0 + 3
                const float v4[4] = {e == 0 ? wv[u][0].x : e == 1 ? wv[u][0].y : e == 2 ? wv[u][0].z : wv[u][0].w,
                                     e == 0 ? wv[u][1].x : e == 1 ? wv[u][1].y : e == 2 ? wv[u][1].z : wv[u][1].w,
                                     e == 0 ? wv[u][2].x : e == 1 ? wv[u][2].y : e == 2 ? wv[u][2].z : wv[u][2].w,
                                     e == 0 ? wv[u][3].x : e == 1 ? wv[u][3].y : e == 2 ? wv[u][3].z : wv[u][3].w};
                lt_store4(WB, N, WS, n0 + e, k0, v4, wsc);
            }
        }
    }
    for (int t = tid; t < N; t += NT) BL[t] = bias ? bias[t] : 0.f;
    __syncthreads();

    for (int tile = slot * (NT / 64) + wave; tile < tiles; tile += wave_stride) {
        const int m = (tile << 5) + l31;
        const bool valid = m < M;
        f32x16 acc[NB];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[nb][r] = 0.f;
        float xun = 1.0f;                                      // inverse of this lane's row scale x inverse weight scale
        float am = 0.f;
#pragma unroll
        for (int s = 0; s < S; ++s) {
            am = fmaxf(am, fmaxf(fmaxf(fabsf(xa[s].x), fabsf(xa[s].y)), fmaxf(fabsf(xa[s].z), fabsf(xa[s].w))));
            am = fmaxf(am, fmaxf(fmaxf(fabsf(xb[s].x), fabsf(xb[s].y)), fmaxf(fabsf(xb[s].z), fabsf(xb[s].w))));
        }
        am = fmaxf(am, __shfl_xor(am, 32));                // the other half of the row sits on lane ^ 32
        float xsc, xu;
        pow2_scale(am, xsc, xu);
        xun = xu * wun;
        const _Float16 *WH = reinterpret_cast<const _Float16 *>(WB);
        // the whole tile becomes fp16 planes first (same register count as the fp32 rows): the fp32 registers are then free for the
        // NEXT tile's rows, whose loads fly during this tile's MFMA loop and epilogue instead of during the epilogue alone
        f16x8 q1[S], q2[S];
#pragma unroll
        for (int s = 0; s < S; ++s) {
            const float xv[8] = {xa[s].x, xa[s].y, xa[s].z, xa[s].w, xb[s].x, xb[s].y, xb[s].z, xb[s].w};
            split2h(xv, xsc, q1[s], q2[s]);
        }
        if (tile + wave_stride < tiles) load_x(tile + wave_stride);
#pragma unroll
        for (int s = 0; s < S; ++s) {
            const int colp = 16 * s + 8 * h;
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                const int row = 32 * nb + l31;
                const f16x8 p1 = *reinterpret_cast<const f16x8 *>(&WH[(0 * N + row) * WS + colp]);
                const f16x8 p2 = *reinterpret_cast<const f16x8 *>(&WH[(1 * N + row) * WS + colp]);
                acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(p2, q1[s], acc[nb], 0, 0, 0);
                acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(p1, q2[s], acc[nb], 0, 0, 0);
                acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(p1, q1[s], acc[nb], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        const int mc = valid ? m : M - 1;                      // rows beyond M compute on a clamped row and are not stored
        const float *rr = residual ? residual + (size_t)mc * ldy + 4 * h : nullptr;
        const float *ar = accum ? accum + (size_t)mc * ldy + 4 * h : nullptr;
        float4 rv[NB][4];
        if (rr) {
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                for (int q = 0; q < 4; ++q) rv[nb][q] = *reinterpret_cast<const float4 *>(rr + 32 * nb + 8 * q);
        }
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 bb = *reinterpret_cast<const float4 *>(&BL[32 * nb + 8 * q + 4 * h]);
                float v[4] = {fmaf(acc[nb][4 * q], xun, bb.x), fmaf(acc[nb][4 * q + 1], xun, bb.y), fmaf(acc[nb][4 * q + 2], xun, bb.z),
                              fmaf(acc[nb][4 * q + 3], xun, bb.w)};
                if (ar) {
                    const float4 av = *reinterpret_cast<const float4 *>(ar + 32 * nb + 8 * q);
                    v[0] += av.x; v[1] += av.y; v[2] += av.z; v[3] += av.w;
                }
                if (pre_out && valid) *reinterpret_cast<float4 *>(pre_out + (size_t)m * ldy + 4 * h + 32 * nb + 8 * q) = make_float4(v[0], v[1], v[2], v[3]);
                const float r4[4] = {rr ? rv[nb][q].x : 0.f, rr ? rv[nb][q].y : 0.f, rr ? rv[nb][q].z : 0.f, rr ? rv[nb][q].w : 0.f};
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if (act == 1) v[u] = ssp_f(v[u]);
                    else if (act == 3) v[u] = v[u] * __builtin_amdgcn_rcpf(1.0f + __expf(-v[u]));      // hardware reciprocal (1 ulp), as visnet.hip's silu_f
                    if (act == 2) v[u] *= 1.0f - 0.5f * __expf(-r4[u]);
                    else if (rr) v[u] += r4[u];
                }
                *reinterpret_cast<float4 *>(&OT[l31 * LT_OP + 32 * (nb & 1) + 8 * q + 4 * h]) = make_float4(v[0], v[1], v[2], v[3]);
            }
            // fp16 form: the planes leave room in LDS for a [32][64] slab per wave — the outputs cross it so that a store instruction
            // writes 4 rows x 256 contiguous bytes instead of 32 rows x 32 bytes (partial lines that L2 has to merge; with the loads
            // switched off the kernel took 86 us of its 128 for the stores alone: 128 -> 116 us).  The slab is private to the wave: no barrier.
            if (nb & 1) {
                const int rbase = tile << 5;
                wave_lds_fence();                              // the slab rows below were written by other lanes of this wavefront
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int r = 4 * i + (lane >> 4), c = 4 * (lane & 15);
                    const float4 o = *reinterpret_cast<const float4 *>(&OT[r * LT_OP + c]);
                    if (rbase + r < M) *reinterpret_cast<float4 *>(y + (size_t)(rbase + r) * ldy + 32 * (nb - 1) + c) = o;
                }
                wave_lds_fence();                              // ... and are overwritten by the next pair of blocks
            }
        }
    }
}

template <int K, int N>
int launch_t(const float *x, const float *w, const float *bias, const float *residual, int M, int w_kn, int act, float *y,
             const int *m_dev, hipStream_t s, int ldx = K, int ldw = 0, int ldy = N, const float *accum = nullptr, float *pre_out = nullptr,
             int nc = 1, const LtJobs *jobs = nullptr) {
    if (ldw == 0) ldw = w_kn ? N : K;
    LtJobs J{};
    if (jobs) J = *jobs;
    const size_t lds_w = ((size_t)(LT_NPL * N * (K + 8)) / 2 + N) * 4;
    const int tiles16 = (M + 31) / 32;
    // 8 waves per workgroup (2 per SIMD cover each other's latencies) once every CU gets a full workgroup; below that 4-wave
    // workgroups spread the tiles over twice as many CUs (node-level layers: 790 tiles -> 198 instead of 99 CUs).
    const bool narrow = tiles16 < 8 * 256;
    const size_t lds16 = lds_w + (size_t)(narrow ? 4 : 8) * 32 * LT_OP * 4;
    const int per = narrow ? 4 : 8;
    int grid16 = (tiles16 + per - 1) / per;
    if (grid16 > CONAN_LINEAR_MAX_WGS / nc) grid16 = CONAN_LINEAR_MAX_WGS / nc;          // (nc chunks: MAX_WGS / nc tile slots x nc)
    grid16 *= nc;
#define LAUNCH16(A)                                                                                                              \
    do {                                                                                                                         \
        if (narrow) {                                                                                                            \
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_linear_t16<K, N, A, 256>),                               \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds16);                                   \
            k_linear_t16<K, N, A, 256><<<grid16, 256, lds16, s>>>(x, w, bias, residual, M, w_kn, y, m_dev, ldx, ldw, ldy, accum, pre_out, nc, J); \
        } else {                                                                                                                 \
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_linear_t16<K, N, A, LT_THREADS>),                        \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds16);                                   \
            k_linear_t16<K, N, A, LT_THREADS><<<grid16, LT_THREADS, lds16, s>>>(x, w, bias, residual, M, w_kn, y, m_dev, ldx, ldw, ldy, accum, \
                                                                                pre_out, nc, J);                                 \
        }                                                                                                                        \
    } while (0)
    switch (act) {
        case 0: LAUNCH16(0); break;
        case 1: LAUNCH16(1); break;
        case 2: LAUNCH16(2); break;
        default: LAUNCH16(3); break;
    }
#undef LAUNCH16
    return hipGetLastError() == hipSuccess ? CONAN_OK : CONAN_E_LAUNCH;
}


// ---- y = sum_c x_c W_c^T (+ bias) (+ residual): up to three 128-wide contractions of DIFFERENT inputs in one launch -------------------------
// The input gradient of several Linear layers of one input (ViS_MP's dk / dv / f_proj of f: dx = sum_c g_c W_c) and of a layer wider than one
// 128-chunk (s_proj: g [E,256]) was a chain of launches that carried the running sum through HBM: read g_c, read the sum, write the sum, per
// chunk — 9 and 5 [E,128] tensors where 5 and 3 have to move.  Here the sum stays in the accumulators: a wavefront streams its 32 rows of
// chunk 0, 1, 2 one after the other (the next chunk's rows fly during this chunk's MFMA loop).  All chunks' weight planes have to stay in
// LDS for the whole persistent loop, which is why a workgroup owns 64 of the 128 outputs (3 x 34 KB instead of 3 x 68 KB) and the two halves of
// a tile slot run as two workgroups of the same XCD that find each other's rows in that XCD's L2.
// Row scales: chunk c of a row has its own power-of-two scale; the accumulators live in units of 2^r with r the largest unit so far — when a
// chunk raises r the accumulators are multiplied by the (exact) ratio, a chunk below it is scaled by 2^-r' >= its own unit instead, which is
// what one scale for the concatenated row would do to it.
struct LtSrcs {
    const float *x[3], *w[3];
    int ldx[3];
};
__device__ __forceinline__ float pow2i(int e) { return e < -126 ? 0.f : __uint_as_float((unsigned)(min(e, 127) + 127) << 23); }
constexpr int LT_NOEXP = -100000;                              // "no unit": a zero / subnormal / non-finite maximum
__device__ __forceinline__ int pow2_exp(float amax) {          // e with amax * 2^-e in [256, 512) (pow2_scale's un = 2^e), or LT_NOEXP
    const int e = (int)((__float_as_uint(amax) >> 23) & 0xffu);
    return (e >= 9 && e <= 254) ? e - 135 : LT_NOEXP;
}
// the two fp16 planes of one [N][K] weight image (k_linear_t16's staging as a function); returns the image's unit exponent
template <int K, int N, int NT>
__device__ __forceinline__ int lt_stage_planes(const float *__restrict__ w, int w_kn, int ldw, _Float16 *WB, float *wred) {
    constexpr int WS = K + 8;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int ew;
    if (!w_kn) {
        constexpr int V4 = N * K / 4, PER = (V4 + NT - 1) / NT;
        float4 wv[PER];
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int q = tid + u * NT;
            const int n = (4 * q) / K, k = 4 * q - n * K;
            wv[u] = q < V4 ? *reinterpret_cast<const float4 *>(w + (size_t)n * ldw + k) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        float am = 0.f;
#pragma unroll
        for (int u = 0; u < PER; ++u) am = fmaxf(am, fmaxf(fmaxf(fabsf(wv[u].x), fabsf(wv[u].y)), fmaxf(fabsf(wv[u].z), fabsf(wv[u].w))));
        ew = pow2_exp(lt_block_absmax<NT>(am, wred));
        const float wsc = ew == LT_NOEXP ? 1.0f : pow2i(-ew);
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int q = tid + u * NT;
            if (q >= V4) continue;
            const float v4[4] = {wv[u].x, wv[u].y, wv[u].z, wv[u].w};
            const int n = (4 * q) / K, k = 4 * q - n * K;
            lt_store4(WB, N, WS, n, k, v4, wsc);
        }
    } else {
        constexpr int PATCHES = (K / 16) * (N / 64), PERW = (PATCHES + NT / 64 - 1) / (NT / 64);
        float4 wv[PERW][4];
        const int n4l = (lane & 3) | ((lane >> 4) << 2), k4l = (lane >> 2) & 3;
#pragma unroll
        for (int u = 0; u < PERW; ++u) {
            const int pt = wave + u * (NT / 64);
            const int k0 = (pt / (N / 64)) * 16 + 4 * k4l, n0 = (pt % (N / 64)) * 64 + 4 * n4l;
#pragma unroll
            for (int j = 0; j < 4; ++j)
                wv[u][j] = pt < PATCHES ? *reinterpret_cast<const float4 *>(w + (size_t)(k0 + j) * ldw + n0) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        float am = 0.f;
#pragma unroll
        for (int u = 0; u < PERW; ++u)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                am = fmaxf(am, fmaxf(fmaxf(fabsf(wv[u][j].x), fabsf(wv[u][j].y)), fmaxf(fabsf(wv[u][j].z), fabsf(wv[u][j].w))));
        ew = pow2_exp(lt_block_absmax<NT>(am, wred));
        const float wsc = ew == LT_NOEXP ? 1.0f : pow2i(-ew);
#pragma unroll
        for (int u = 0; u < PERW; ++u) {
            const int pt = wave + u * (NT / 64);
            if (pt >= PATCHES) continue;
            const int k0 = (pt / (N / 64)) * 16 + 4 * k4l, n0 = (pt % (N / 64)) * 64 + 4 * n4l;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float v4[4] = {e == 0 ? wv[u][0].x : e == 1 ? wv[u][0].y : e == 2 ? wv[u][0].z : wv[u][0].w,
                                     e == 0 ? wv[u][1].x : e == 1 ? wv[u][1].y : e == 2 ? wv[u][1].z : wv[u][1].w,
                                     e == 0 ? wv[u][2].x : e == 1 ? wv[u][2].y : e == 2 ? wv[u][2].z : wv[u][2].w,
                                     e == 0 ? wv[u][3].x : e == 1 ? wv[u][3].y : e == 2 ? wv[u][3].z : wv[u][3].w};
                lt_store4(WB, N, WS, n0 + e, k0, v4, wsc);
            }
        }
    }
    __syncthreads();                                           // wred is reused by the next image
    return ew == LT_NOEXP ? 0 : ew;                            // (an all-zero image: unit 1, planes 0)
}

constexpr int LS_OP = 36;                                      // the per-wave output slab is 32 x LS_OP floats, used as [16][LT_OP]
// The 64 outputs x 32 rows a wave holds after its epilogue (vv[nb][q] = features 32 nb + 8 q + 4 h .. + 3 of row l31) to y: through the wave's
// slab as two half tiles of [16 rows][64 + 4], so that a store instruction writes 4 rows x 256 contiguous bytes (as [32][32 + 4], 8 rows x 128 bytes per instruction, it measured the same).
__device__ __forceinline__ void ls_store64(float *OT, const float4 (&vv)[2][4], float *yb, int ldy, int rbase, int M, int lane) {
    const int l31 = lane & 31, h = lane >> 5;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        if ((l31 >> 4) == half) {
#pragma unroll
            for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                for (int q = 0; q < 4; ++q) *reinterpret_cast<float4 *>(&OT[(l31 & 15) * LT_OP + 32 * nb + 8 * q + 4 * h]) = vv[nb][q];
        }
        wave_lds_fence();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int rw = 4 * i + (lane >> 4), cc = 4 * (lane & 15);
            const float4 o = *reinterpret_cast<const float4 *>(&OT[rw * LT_OP + cc]);
            const int r = rbase + 16 * half + rw;
            if (r < M) *reinterpret_cast<float4 *>(yb + (size_t)r * ldy + cc) = o;
        }
        wave_lds_fence();
    }
}
#ifndef CONAN_LSUM_PAIR_XCD
#define CONAN_LSUM_PAIR_XCD 1                                  // the two output halves of a tile slot on ONE XCD (blocks b and b + 8); 0: blocks b, b + 1
#endif
template <int NCH, int NT>
__global__ void __launch_bounds__(NT) k_linear_sum16(const LtSrcs Sx, const float *__restrict__ bias, const float *__restrict__ residual, int M,
                                                     int w_kn, float *y, const int *__restrict__ m_dev, int ldw, int ldy) {
    constexpr int K = 128, N = 64, NB = N / 32, S = K / 16, WS = K + 8, PL = LT_NPL * N * WS;      // PL: 16-bit elements of one chunk's planes
    extern __shared__ __attribute__((aligned(16))) float lds[];
    _Float16 *WH = reinterpret_cast<_Float16 *>(lds);          // [NCH][planes][N][WS]
    float *BL = lds + (NCH * PL) / 2;                          // [N]
    float *OT = BL + N + (threadIdx.x >> 6) * (32 * LS_OP);    // this wave's [32][LS_OP] output slab
    __shared__ float wred[NT / 64];
    if (m_dev) M = min(M, *m_dev);
    const int tiles = (M + 31) >> 5;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = (int)blockIdx.x;
#if CONAN_LSUM_PAIR_XCD
    const int half = (b >> 3) & 1, slot = ((b >> 4) << 3) | (b & 7);        // gridDim.x is a multiple of 16
#else
    const int half = b & 1, slot = b >> 1;
#endif
    const int nslots = (int)gridDim.x >> 1;
    const int n0 = half * N;
    if (slot * (NT / 64) >= tiles) return;
    const int l31 = lane & 31, h = lane >> 5;
    const int wave_stride = nslots * (NT / 64);
    float4 xa[S], xb[S];
    auto load_x = [&](int t, int c) {
        const int mr = min((t << 5) + l31, M - 1);
        const float *xr = Sx.x[c] + (size_t)mr * Sx.ldx[c] + 8 * h;
#pragma unroll
        for (int s = 0; s < S; ++s) {
            xa[s] = *reinterpret_cast<const float4 *>(xr + 16 * s);
            xb[s] = *reinterpret_cast<const float4 *>(xr + 16 * s + 4);
        }
    };
    if (slot * (NT / 64) + wave < tiles) load_x(slot * (NT / 64) + wave, 0);
    int ew[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c)
        ew[c] = lt_stage_planes<K, N, NT>(Sx.w[c] + (w_kn ? (size_t)n0 : (size_t)n0 * ldw), w_kn, ldw, WH + c * PL, wred);
    for (int t = tid; t < N; t += NT) BL[t] = bias ? bias[n0 + t] : 0.f;
    __syncthreads();

    for (int tile = slot * (NT / 64) + wave; tile < tiles; tile += wave_stride) {
        const int m = (tile << 5) + l31;
        const bool valid = m < M;
        f32x16 acc[NB];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[nb][r] = 0.f;
        int r = LT_NOEXP;                                      // unit exponent of the accumulators
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            float am = 0.f;
#pragma unroll
            for (int s = 0; s < S; ++s) {
                am = fmaxf(am, fmaxf(fmaxf(fabsf(xa[s].x), fabsf(xa[s].y)), fmaxf(fabsf(xa[s].z), fabsf(xa[s].w))));
                am = fmaxf(am, fmaxf(fmaxf(fabsf(xb[s].x), fabsf(xb[s].y)), fmaxf(fabsf(xb[s].z), fabsf(xb[s].w))));
            }
            am = fmaxf(am, __shfl_xor(am, 32));
            const int ex = pow2_exp(am);
            const int u = ex == LT_NOEXP ? LT_NOEXP : ex + ew[c];
            const int rn = max(r, u);
            if (c > 0) {
                const float fac = pow2i(r - rn);               // 1 unless this chunk raises the unit (0 when r was "no unit": nothing summed yet)
#pragma unroll
                for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                    for (int q = 0; q < 16; ++q) acc[nb][q] *= fac;
            }
            r = rn;
            const float xsc = r == LT_NOEXP ? 1.0f : pow2i(ew[c] - r);
            f16x8 q1[S], q2[S];
#pragma unroll
            for (int s = 0; s < S; ++s) {
                const float xv[8] = {xa[s].x, xa[s].y, xa[s].z, xa[s].w, xb[s].x, xb[s].y, xb[s].z, xb[s].w};
                split2h(xv, xsc, q1[s], q2[s]);
            }
            if (c + 1 < NCH) load_x(tile, c + 1);
            else if (tile + wave_stride < tiles) load_x(tile + wave_stride, 0);
            const _Float16 *Wc = WH + c * PL;
#pragma unroll
            for (int s = 0; s < S; ++s) {
                const int colp = 16 * s + 8 * h;
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) {
                    const int row = 32 * nb + l31;
                    const f16x8 p1 = *reinterpret_cast<const f16x8 *>(&Wc[(0 * N + row) * WS + colp]);
                    const f16x8 p2 = *reinterpret_cast<const f16x8 *>(&Wc[(1 * N + row) * WS + colp]);
                    acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(p2, q1[s], acc[nb], 0, 0, 0);
                    acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(p1, q2[s], acc[nb], 0, 0, 0);
                    acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(p1, q1[s], acc[nb], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        const float xun = r == LT_NOEXP ? 0.f : pow2i(r / 2) * pow2i(r - r / 2);
        const int mc = valid ? m : M - 1;
        const float *rr = residual ? residual + (size_t)mc * ldy + n0 + 4 * h : nullptr;
        float4 rv[NB][4];
        if (rr) {
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                for (int q = 0; q < 4; ++q) rv[nb][q] = *reinterpret_cast<const float4 *>(rr + 32 * nb + 8 * q);
        }
        float4 vv[2][4];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 bb = *reinterpret_cast<const float4 *>(&BL[32 * nb + 8 * q + 4 * h]);
                float4 v = make_float4(fmaf(acc[nb][4 * q], xun, bb.x), fmaf(acc[nb][4 * q + 1], xun, bb.y), fmaf(acc[nb][4 * q + 2], xun, bb.z),
                                       fmaf(acc[nb][4 * q + 3], xun, bb.w));
                if (rr) { v.x += rv[nb][q].x; v.y += rv[nb][q].y; v.z += rv[nb][q].z; v.w += rv[nb][q].w; }
                vv[nb][q] = v;
            }
        ls_store64(OT, vv, y + n0, ldy, tile << 5, M, lane);
    }
}

// ---- y_l = x W_l^T + b_l for up to three 128 -> 128 layers of ONE input in one workgroup per tile (round 5) -------------------------------
// The side-by-side form above (nc = 3) gives every layer its own workgroup: x is fetched and split into its planes three times and a launch
// is 3 x tiles / 8 workgroup iterations of one tile per wave — 290 us for ViS_MP's dk / dv / f_proj at BACE B = 64, three times the single
// layer.  Here a workgroup owns 64 of the 128 outputs of ALL layers (the planes of 3 x 64 x 128 weights: 104 KB, as k_linear_sum16), a wave
// splits its 32 rows once and runs the layers one after the other on the same planes; the two output halves of a tile slot are the
// workgroups b and b + 8 of one XCD.
template <int NL, int NT>
__global__ void __launch_bounds__(NT) k_linear_fan16(const float *__restrict__ x, const LtJobs J, int M, const int *__restrict__ m_dev, int ldx, int ldw,
                                                     int ldy) {
    constexpr int K = 128, N = 64, NB = N / 32, S = K / 16, WS = K + 8, PL = LT_NPL * N * WS;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    _Float16 *WH = reinterpret_cast<_Float16 *>(lds);          // [NL][planes][N][WS]
    float *BL = lds + (NL * PL) / 2;                           // [NL][N]
    float *OT = BL + NL * N + (threadIdx.x >> 6) * (32 * LS_OP);
    __shared__ float wred[NT / 64];
    if (m_dev) M = min(M, *m_dev);
    const int tiles = (M + 31) >> 5;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = (int)blockIdx.x;
    const int half = (b >> 3) & 1, slot = ((b >> 4) << 3) | (b & 7);        // gridDim.x is a multiple of 16
    const int nslots = (int)gridDim.x >> 1;
    const int n0 = half * N;
    if (slot * (NT / 64) >= tiles) return;
    const int l31 = lane & 31, h = lane >> 5;
    const int wave_stride = nslots * (NT / 64);
    float4 xa[S], xb[S];
    auto load_x = [&](int t) {
        const int mr = min((t << 5) + l31, M - 1);
        const float *xr = x + (size_t)mr * ldx + 8 * h;
#pragma unroll
        for (int s = 0; s < S; ++s) {
            xa[s] = *reinterpret_cast<const float4 *>(xr + 16 * s);
            xb[s] = *reinterpret_cast<const float4 *>(xr + 16 * s + 4);
        }
    };
    if (slot * (NT / 64) + wave < tiles) load_x(slot * (NT / 64) + wave);
    int ew[NL];
#pragma unroll
    for (int l = 0; l < NL; ++l) ew[l] = lt_stage_planes<K, N, NT>(J.w[l] + (size_t)n0 * ldw, 0, ldw, WH + l * PL, wred);
    for (int t = tid; t < NL * N; t += NT) BL[t] = J.bias[t / N] ? J.bias[t / N][n0 + (t % N)] : 0.f;
    __syncthreads();

    for (int tile = slot * (NT / 64) + wave; tile < tiles; tile += wave_stride) {
        float am = 0.f;
#pragma unroll
        for (int s = 0; s < S; ++s) {
            am = fmaxf(am, fmaxf(fmaxf(fabsf(xa[s].x), fabsf(xa[s].y)), fmaxf(fabsf(xa[s].z), fabsf(xa[s].w))));
            am = fmaxf(am, fmaxf(fmaxf(fabsf(xb[s].x), fabsf(xb[s].y)), fmaxf(fabsf(xb[s].z), fabsf(xb[s].w))));
        }
        am = fmaxf(am, __shfl_xor(am, 32));
        float xsc, xu;
        pow2_scale(am, xsc, xu);
        f16x8 q1[S], q2[S];
#pragma unroll
        for (int s = 0; s < S; ++s) {
            const float xv[8] = {xa[s].x, xa[s].y, xa[s].z, xa[s].w, xb[s].x, xb[s].y, xb[s].z, xb[s].w};
            split2h(xv, xsc, q1[s], q2[s]);
        }
        if (tile + wave_stride < tiles) load_x(tile + wave_stride);
        const int rbase = tile << 5;
#pragma unroll
        for (int l = 0; l < NL; ++l) {
            f32x16 acc[NB];
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[nb][r] = 0.f;
            const _Float16 *Wl = WH + l * PL;
#pragma unroll
            for (int s = 0; s < S; ++s) {
                const int colp = 16 * s + 8 * h;
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) {
                    const int row = 32 * nb + l31;
                    const f16x8 p1 = *reinterpret_cast<const f16x8 *>(&Wl[(0 * N + row) * WS + colp]);
                    const f16x8 p2 = *reinterpret_cast<const f16x8 *>(&Wl[(1 * N + row) * WS + colp]);
                    acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(p2, q1[s], acc[nb], 0, 0, 0);
                    acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(p1, q2[s], acc[nb], 0, 0, 0);
                    acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(p1, q1[s], acc[nb], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            const float xun = xu * pow2i(ew[l]);               // (pow2_scale's inverse row scale x the layer's weight unit)
            float4 vv[2][4];
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 bb = *reinterpret_cast<const float4 *>(&BL[l * N + 32 * nb + 8 * q + 4 * h]);
                    vv[nb][q] = make_float4(fmaf(acc[nb][4 * q], xun, bb.x), fmaf(acc[nb][4 * q + 1], xun, bb.y), fmaf(acc[nb][4 * q + 2], xun, bb.z),
                                            fmaf(acc[nb][4 * q + 3], xun, bb.w));
                }
            ls_store64(OT, vv, J.y[l] + n0, ldy, rbase, M, lane);
        }
    }
}

template <int NL>
int launch_fan(const float *x, const LtJobs &J, int M, const int *m_dev, hipStream_t s) {
    constexpr int NT = LT_THREADS;
    const size_t lds = ((size_t)NL * LT_NPL * 64 * (128 + 8) / 2 + NL * 64 + (size_t)(NT / 64) * 32 * LS_OP) * 4;
    const int tiles = (M + 31) / 32;
    int slots = (tiles + NT / 64 - 1) / (NT / 64);
    if (slots > CONAN_LINEAR_MAX_WGS / 2) slots = CONAN_LINEAR_MAX_WGS / 2;
    slots = (slots + 7) & ~7;
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_linear_fan16<NL, NT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    k_linear_fan16<NL, NT><<<2 * slots, NT, lds, s>>>(x, J, M, m_dev, 128, 128, 128);
    return hipGetLastError() == hipSuccess ? CONAN_OK : CONAN_E_LAUNCH;
}

template <int NCH>
int launch_sum(const LtSrcs &Sx, const float *bias, const float *residual, int M, int w_kn, float *y, const int *m_dev, int ldw, int ldy,
               hipStream_t s) {
    constexpr int NT = LT_THREADS;
    const size_t lds = ((size_t)NCH * LT_NPL * 64 * (128 + 8) / 2 + 64 + (size_t)(NT / 64) * 32 * LS_OP) * 4;
    const int tiles = (M + 31) / 32;
    int slots = (tiles + NT / 64 - 1) / (NT / 64);
    if (slots > CONAN_LINEAR_MAX_WGS / 2) slots = CONAN_LINEAR_MAX_WGS / 2;
    slots = (slots + 7) & ~7;                                  // (the XCD pairing wants 8 slots per group of 16 blocks; surplus workgroups return at once)
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_linear_sum16<NCH, NT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    k_linear_sum16<NCH, NT><<<2 * slots, NT, lds, s>>>(Sx, bias, residual, M, w_kn, y, m_dev, ldw, ldy);
    return hipGetLastError() == hipSuccess ? CONAN_OK : CONAN_E_LAUNCH;
}

}  // namespace

// Returns 1 and launches when the layer can be tiled into register-streamed (K, N) chunks of 64 / 128, 0 otherwise (caller
// falls back).  Wider layers (ViSNet's 128 -> 256 / 384 projections and their transposes, the 512 / 256-wide classification
// SchNet) become one strided problem per (n chunk, k chunk): the k chunks of an n chunk accumulate in place through `accum`,
// bias enters with the first k chunk, activation and residual with the last.
template <int KC, int NC>
static int chunk_launch(const float *x, const float *w, const float *bias, const float *residual, int M, int K, int N, int w_kn, int act,
                        float *y, const int *m_dev, hipStream_t s, float *pre_out) {
    if (K == KC && N / NC > 1 && N / NC <= 4)              // one contraction chunk: the N / NC output chunks share their x tiles in ONE launch
        return launch_t<KC, NC>(x, w, bias, residual, M, w_kn, act, y, m_dev, s, K, w_kn ? N : K, N, nullptr, pre_out, N / NC);
    for (int n0 = 0; n0 < N; n0 += NC)
        for (int k0 = 0; k0 < K; k0 += KC) {
            const bool first = k0 == 0, last = k0 + KC >= K;
            const float *wc = w_kn ? w + (size_t)k0 * N + n0 : w + (size_t)n0 * K + k0;
            const int rc = launch_t<KC, NC>(x + k0, wc, (first && bias) ? bias + n0 : nullptr, (last && residual) ? residual + n0 : nullptr, M, w_kn,
                                            last ? act : 0, y + n0, m_dev, s, K, w_kn ? N : K, N, first ? nullptr : y + n0,
                                            (last && pre_out) ? pre_out + n0 : nullptr);
            if (rc != CONAN_OK) return rc;
        }
    return CONAN_OK;
}

#ifndef CONAN_LINEAR_FAN
#define CONAN_LINEAR_FAN 1
#endif
#ifndef CONAN_LINEAR_FAN_MIN_ROWS
#define CONAN_LINEAR_FAN_MIN_ROWS 65536              // edge level; a node-level run (q / k / v of 17 k atoms) keeps one workgroup per layer and tile: more workgroups in flight
#endif
// njobs (2..4) Linear layers K = 128 -> N = 128 of the same x in one launch; returns 0 when the shape is not covered (caller: one call per layer)
int conan_linear_t_multi(const float *x, const float *const *w, const float *const *bias, int M, int K, int N, int njobs, int act, float *const *y,
                         float *const *pre, const int *m_dev, hipStream_t s, int *rc) {
    if (K != 128 || N != 128 || njobs < 2 || njobs > 4 || M < 1) return 0;
    LtJobs J{};
    for (int q = 0; q < njobs; ++q) { J.w[q] = w[q]; J.bias[q] = bias ? bias[q] : nullptr; J.y[q] = y[q]; J.pre[q] = pre ? pre[q] : nullptr; }
    if (CONAN_LINEAR_FAN && act == 0 && !pre && njobs <= 3 && M >= CONAN_LINEAR_FAN_MIN_ROWS) {
        *rc = njobs == 2 ? launch_fan<2>(x, J, M, m_dev, s) : launch_fan<3>(x, J, M, m_dev, s);
        return 1;
    }
    *rc = launch_t<128, 128>(x, w[0], nullptr, nullptr, M, 0, act, y[0], m_dev, s, 128, 0, 128, nullptr, nullptr, njobs, &J);
    return 1;
}

// y [M,128] = sum_c x_c [M,128 | ldx_c] W_c^T (+ bias) (+ residual), nsrc = 2 or 3, in one launch; returns 0 when the shape is not covered
int conan_linear_t_sum(const float *const *x, const int *ldx, const float *const *w, int nsrc, int w_kn, int ldw, const float *bias,
                       const float *residual, int M, int N, float *y, const int *m_dev, hipStream_t s, int *rc) {
    if (N != 128 || nsrc < 2 || nsrc > 3 || M < 1) return 0;
    LtSrcs Sx{};
    for (int c = 0; c < nsrc; ++c) { Sx.x[c] = x[c]; Sx.w[c] = w[c]; Sx.ldx[c] = ldx[c]; }
    *rc = nsrc == 2 ? launch_sum<2>(Sx, bias, residual, M, w_kn, y, m_dev, ldw, N, s) : launch_sum<3>(Sx, bias, residual, M, w_kn, y, m_dev, ldw, N, s);
    return 1;
}

#ifndef CONAN_LINEAR_NO_KSUM
#define CONAN_LINEAR_KSUM 1
#else
#define CONAN_LINEAR_KSUM 0
#endif
int conan_linear_t_try(const float *x, const float *w, const float *bias, const float *residual, int M, int K, int N, int w_kn,
                       int act, float *y, const int *m_dev, hipStream_t s, int *rc, float *pre_out) {
    if (M < 1) return 0;
    if (CONAN_LINEAR_KSUM && N == 128 && (K == 256 || K == 384) && act == 0 && !pre_out) {
        // a contraction of two or three 128-chunks: the chunks of one x, summed in the accumulators instead of through y (k_linear_sum16)
        const float *xs[3], *wc[3];
        int ldx[3];
        for (int c = 0; c < K / 128; ++c) { xs[c] = x + 128 * c; ldx[c] = K; wc[c] = w_kn ? w + (size_t)128 * c * N : w + 128 * c; }
        if (conan_linear_t_sum(xs, ldx, wc, K / 128, w_kn, w_kn ? N : K, bias, residual, M, N, y, m_dev, s, rc)) return 1;
    }
    if (K == 128 && N == 128) { *rc = launch_t<128, 128>(x, w, bias, residual, M, w_kn, act, y, m_dev, s, 128, 0, 128, nullptr, pre_out); return 1; }
    if (K == 128 && N == 64) { *rc = launch_t<128, 64>(x, w, bias, residual, M, w_kn, act, y, m_dev, s, 128, 0, 64, nullptr, pre_out); return 1; }
    if (K == 64 && N == 64) { *rc = launch_t<64, 64>(x, w, bias, residual, M, w_kn, act, y, m_dev, s, 64, 0, 64, nullptr, pre_out); return 1; }
    if (K == 64 && N == 128) { *rc = launch_t<64, 128>(x, w, bias, residual, M, w_kn, act, y, m_dev, s, 64, 0, 128, nullptr, pre_out); return 1; }
    if (K == 32 && N == 128 && !w_kn) { *rc = launch_t<32, 128>(x, w, bias, residual, M, w_kn, act, y, m_dev, s, 32, 0, 128, nullptr, pre_out); return 1; }   // ViSNet's rbf projections (32 -> 128, edge level)
    if ((K % 64) || (N % 64) || K > 1024 || N > 1024) return 0;
    const bool k128 = (K % 128) == 0, n128 = (N % 128) == 0;
    if (k128 && n128) *rc = chunk_launch<128, 128>(x, w, bias, residual, M, K, N, w_kn, act, y, m_dev, s, pre_out);
    else if (k128) *rc = chunk_launch<128, 64>(x, w, bias, residual, M, K, N, w_kn, act, y, m_dev, s, pre_out);
    else if (n128) *rc = chunk_launch<64, 128>(x, w, bias, residual, M, K, N, w_kn, act, y, m_dev, s, pre_out);
    else *rc = chunk_launch<64, 64>(x, w, bias, residual, M, K, N, w_kn, act, y, m_dev, s, pre_out);
    return 1;
}
