// Fused continuous-filter generator of SchNet's CFConv for gfx950:
//
//     W[e,:] = ( mlp2( ssp( mlp0( rbf(d_e) ) ) ) ) * 0.5*(cos(d_e*pi/cutoff)+1)
//
// (GaussianSmearing + InteractionBlock.mlp + CFConv's cosine cutoff of PyG; reached from
// conan_fgw/src/model/graph_embeddings/schnet_no_sum.py:161-164,209-212).  One kernel, no [E,Gs] / [E,F] intermediates
// in HBM: per edge the only traffic is 4 B in (d_e) and 4F B out (W[e,:]).
//
// MI355X mapping.  Everything is computed TRANSPOSED so that the edge index lives on the MFMA column (= lane) and the
// filter channel on the MFMA row: a wavefront owns a tile of 32 edges and keeps H1^T[F][32] and W^T[F][32] in
// accumulators (2 * F/32 * 16 registers).
//   GEMM1^T  H1^T = W1 . rbf^T : the B operand (rbf_k(d_e), lane = e) is evaluated in registers, never stored;
//   GEMM2^T  W^T  = W2 . H1    : the B operand IS the accumulator of GEMM1 (after bias + ssp), register r of lane-half h
//                                being row 32mb + (r&3) + 8(r>>2) + 4h, so the k order of the A fragments is permuted to
//                                match: no LDS round trip, no cross-lane traffic between the two GEMMs.
// The A operands (the two weight matrices as two fp16 planes each, pre-scaled by an exact power of two: 36 + 68 KB at F = 128, of the CU's
// 160 KB) are staged once per workgroup in LDS and read with conflict-free ds_read_b128 (one 16-wide k-step of one plane per read).
// Wavefronts never synchronise after staging: 8 independent waves per CU issue MFMAs back to back, both GEMMs as two-plane fp16 splits on
// v_mfma_f32_32x32x16_f16 (three partial products per fp32 product, fp32-class accuracy; there is no other arithmetic mode and no
// environment switch.  Round 2 used three bf16 planes and six partial products: same accuracy, 28 % slower, removed in round 4).
#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
// Two-plane fp16 split of 8 fp32 values: v = h1 + h2 up to 2^-22 |v| while the remainder v - h1 is a normal fp16 number (|v| >~ 0.06),
// and to 3e-8 absolute below that (fp16 subnormal spacing).  With the three products p1q1, p1q2, p2q1 (the dropped p2q2 is 2^-22
// relative) an fp16 MFMA chain reproduces the fp32 product to ~2.4e-7 at HALF the matrix-pipe time and ~2/3 of the splitting work
// of a three-plane bf16 form (6 products).  Range: the operands here are O(1e-2..1e2) — Gaussians in [0, 1], shifted-softplus
// outputs, weights pre-scaled by a power of two chosen from their maximum (exact, undone in the epilogue's FMA) — far inside fp16's 6e-5..65504.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ void split2h(const float *v, f16x8 &p1, f16x8 &p2) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const _Float16 h1 = (_Float16)v[j];
        p1[j] = h1; p2[j] = (_Float16)(v[j] - (float)h1);
    }
}
// Weight pre-scale of the fp16 planes: 2^k with max |w| * 2^k in [256, 512) — whatever the magnitude of the weights, their planes sit in the
// middle of fp16's range (remainders normal, nothing near 65504); the exact inverse goes into the epilogue's FMA.  One block-wide
// maximum per matrix at staging time.
__device__ __forceinline__ void plane_scale(float amax, float &sc, float &un) {
    sc = 1.0f; un = 1.0f;
    if (amax > 0.f && amax < 3.0e38f) { int e; (void)frexpf(amax, &e); sc = ldexpf(1.0f, 9 - e); un = ldexpf(1.0f, e - 9); }
}
template <int NT>
__device__ __forceinline__ float block_absmax(float v, float *red) {      // red: NT / 64 floats of LDS; every thread gets the maximum
    v = wave_max(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float m = red[0];
#pragma unroll
    for (int w = 1; w < NT / 64; ++w) m = fmaxf(m, red[w]);
    return m;
}

constexpr int GP = 64;        // gaussians padded to four MFMA k-steps of 16 (zero weights beyond num_gaussians)
constexpr int W1S = GP + 8;   // LDS pitch of a W1 row in the split images (16-bit elements: 144 B, 16-B slots of 8 consecutive rows stay distinct)
constexpr int FF_THREADS = 512;      // eight wavefronts per workgroup.  (Twelve — three per SIMD, 154 KB of LDS, 160 registers — measured the same in an in-process A/B:
                                     // with h1 67-68 us both, W only 58-61 us both; tools/ab_inproc_filter.py, profiles/r5_ab_filter_fwd_768_threads.txt)

constexpr int FF_NPL = 2;                                    // operand planes: two fp16 planes (the round-2 three-plane bf16 form measured 28 % slower: DESIGN 3.1)
constexpr int FF_OP = 36;                                    // pitch of the per-wave output slab (floats): 32 channels + 4

// CF = true (round 5, forward-only): the filter rows never reach HBM — the tile's rows are the DIRECTED edges e0 .. e0 + 31 in CSR order (grouped
// by target), each row is multiplied by its source's features x[col[e], :] as it leaves the accumulators and the products are summed per
// target (CFConv.propagate: out[i] = sum_{e in row i} x[col[e]] * W[e]) through the per-wave slab: lane <-> channel walks the 32 rows in edge
// order and adds each finished target segment to `out` (pre-zeroed by the entry point).  A target with <= 33 edges spans at most two tiles, so
// its row of `out` is 0 + a (+ b): the result does not depend on which tile's add lands first (bitwise reproducible at cap 32).
template <int F, bool CF = false, int NT = FF_THREADS>
__global__ void __launch_bounds__(NT) k_filter_fused(
    const float *__restrict__ dist, const int *__restrict__ num_edges_dev, int max_edges, const float *__restrict__ offset,
    int Gs, float coeff, float cutoff, const float *__restrict__ w1, const float *__restrict__ b1,
    const float *__restrict__ w2, const float *__restrict__ b2, float *__restrict__ Wout, float *__restrict__ h1_out,
    const float *__restrict__ xin = nullptr, const int *__restrict__ col = nullptr, const int *__restrict__ tgt = nullptr) {
    constexpr int MB = F / 32;            // 32-row blocks of the channel dimension
    constexpr int W2S = F + 8;            // LDS pitch of a W2 row in the split images (16-bit elements; 16-B slots stay distinct)
    constexpr int W2WORDS = (FF_NPL * F * W2S) / 2;      // floats occupied by the 16-bit W2 images
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int W1WORDS = (FF_NPL * F * W1S) / 2;      // floats occupied by the 16-bit W1 images
    float *W1L = lds;                     // 2 x fp16 [F][W1S]
    float *W2L = W1L + W1WORDS;           // 2 x fp16 [F][W2S], columns permuted per 16-group
    float *B1L = W2L + W2WORDS;           // [F]
    float *B2L = B1L + F;                 // [F]
    float *OFL = B2L + F;                 // [GP]
    // fp16 form: a [32][FF_OP] output slab per wave — the outputs of a 32-channel block cross it so that a store instruction writes 8 rows
    // x 128 contiguous bytes (whole lines) instead of 32 rows x 32 bytes that L2 has to merge (gemm_t.hip measured the effect)
    float *OT = OFL + GP + (threadIdx.x >> 6) * (32 * FF_OP);
    __shared__ float wred[NT / 64];
    float us1 = 1.0f, us2 = 1.0f;                                 // inverse plane scales of W1 / W2 (fp16 form)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int E = num_edges_dev ? min(*num_edges_dev, max_edges) : max_edges;
    const int tiles = (E + 31) >> 5;
    if (!CF && (int)blockIdx.x * (NT / 64) >= tiles) return;
    if (CF && tiles == 0) return;

    {
        constexpr int PER1 = (F * GP + NT - 1) / NT;
        float wv[PER1];
#pragma unroll
        for (int u = 0; u < PER1; ++u) {
            const int t = tid + u * NT, f = t / GP, k = t - f * GP;
            wv[u] = (t < F * GP && k < Gs) ? w1[(size_t)f * Gs + k] : 0.f;
        }
        float sc1 = 1.0f;
        float am = 0.f;
#pragma unroll
        for (int u = 0; u < PER1; ++u) am = fmaxf(am, fabsf(wv[u]));
        plane_scale(block_absmax<NT>(am, wred), sc1, us1);
#pragma unroll
        for (int u = 0; u < PER1; ++u) {
            const int t = tid + u * NT, f = t / GP, k = t - f * GP;
            if (t >= F * GP) continue;
            _Float16 *W1H = reinterpret_cast<_Float16 *>(W1L);
            const float v = wv[u] * sc1;
            const _Float16 h1 = (_Float16)v;
            W1H[(0 * F + f) * W1S + k] = h1;
            W1H[(1 * F + f) * W1S + k] = (_Float16)(v - (float)h1);
        }
    }
    {
        // two fp16 planes of W2; inside every group of 16 input channels the columns are stored in the order the B
        // fragment (GEMM1's accumulator registers 8s..8s+7 of lane-half h) enumerates them: position 8h + j holds
        // channel (j&3) + 8(j>>2) + 4h, so a lane reads its 8 k-values as one 16-byte access.
        // all of a thread's loads are issued before the first use: a load-convert-store loop serialises F*F/512 L2 round
        // trips (a fixed ~15 us per launch)
        constexpr int PER2 = (F * F + NT - 1) / NT;
        float wv[PER2];
#pragma unroll
        for (int u = 0; u < PER2; ++u) { const int t = tid + u * NT; wv[u] = t < F * F ? w2[t] : 0.f; }
        float sc2 = 1.0f;
        float am = 0.f;
#pragma unroll
        for (int u = 0; u < PER2; ++u) am = fmaxf(am, fabsf(wv[u]));
        plane_scale(block_absmax<NT>(am, wred), sc2, us2);
#pragma unroll
        for (int u = 0; u < PER2; ++u) {
            const int t = tid + u * NT;
            if (t >= F * F) continue;
            const int f2 = t / F, f = t - f2 * F;
            const int kk = f & 15, hh = (kk >> 2) & 1, jj = (kk & 3) + 4 * (kk >> 3);
            const int colp = (f & ~15) + 8 * hh + jj;
            _Float16 *W2H = reinterpret_cast<_Float16 *>(W2L);
            const float v = wv[u] * sc2;
            const _Float16 h1 = (_Float16)v;
            W2H[(0 * F + f2) * W2S + colp] = h1;
            W2H[(1 * F + f2) * W2S + colp] = (_Float16)(v - (float)h1);
        }
    }
    for (int t = tid; t < F; t += NT) { B1L[t] = b1[t]; B2L[t] = b2[t]; }
    for (int t = tid; t < GP; t += NT) OFL[t] = t < Gs ? offset[t] : 0.f;
    __syncthreads();

    const int l31 = lane & 31, h = lane >> 5;
    // CF: every workgroup owns a CONTIGUOUS range of tiles, XCD k the k-th eighth of the edges (the x rows of a conformer are gathered by ~20
    // neighbouring targets: they meet in one L2); the plain generator deals tiles round-robin as before
    const int per_wg = CF ? (tiles + (int)gridDim.x - 1) / (int)gridDim.x : 0;
    const int lb = CF ? ((gridDim.x & 7) == 0 ? xcd_contiguous_block((int)blockIdx.x, (int)gridDim.x) : (int)blockIdx.x) : 0;
    const int t_begin = CF ? lb * per_wg + wave : blockIdx.x * (NT / 64) + wave;
    const int t_end = CF ? min(tiles, (lb + 1) * per_wg) : tiles;
    const int wave_stride = CF ? NT / 64 : gridDim.x * (NT / 64);
    for (int tile = t_begin; tile < t_end; tile += wave_stride) {
        const int e = (tile << 5) + l31;
        const bool valid = e < E;
        const float d = valid ? dist[e] : 0.f;
        int src = 0, trg = -1;
        if constexpr (CF) { if (valid) { src = col[e]; trg = tgt[e]; } }

        // ---------------- GEMM1^T: acc1[mb] = W1[32mb.., :] . rbf^T
        f32x16 acc1[MB];
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc1[mb][r] = 0.f;
        {
            // both operands as two-plane fp16 splits (as GEMM2): the B fragment of k-step s is rbf_k(d_e) for k = 16s + 8h .. +7,
            // evaluated and split in registers; the A fragments are the two W1 planes.  3 x 32 cycles per 32x32x16 block instead
            // of 8 x 64 on the fp32 MFMA.
#pragma unroll
            for (int s = 0; s < GP / 16; ++s) {
                const int kb = 16 * s + 8 * h;
                const float4 o0 = *reinterpret_cast<const float4 *>(&OFL[kb]), o1 = *reinterpret_cast<const float4 *>(&OFL[kb + 4]);
                const float of[8] = {o0.x, o0.y, o0.z, o0.w, o1.x, o1.y, o1.z, o1.w};
                float rb[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) { const float t0 = d - of[j]; rb[j] = exp_neg_f(coeff * (t0 * t0)); }
                const _Float16 *W1H = reinterpret_cast<const _Float16 *>(W1L);
                f16x8 q1, q2;
                split2h(rb, q1, q2);
#pragma unroll
                for (int mb = 0; mb < MB; ++mb) {
                    const int row = 32 * mb + l31;
                    const f16x8 p1 = *reinterpret_cast<const f16x8 *>(&W1H[(0 * F + row) * W1S + kb]);
                    const f16x8 p2 = *reinterpret_cast<const f16x8 *>(&W1H[(1 * F + row) * W1S + kb]);
                    acc1[mb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(p2, q1, acc1[mb], 0, 0, 0);      // smallest terms first
                    acc1[mb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(p1, q2, acc1[mb], 0, 0, 0);
                    acc1[mb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(p1, q1, acc1[mb], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // bias + shifted softplus on the accumulators: register r of half h is channel 32mb + (r&3) + 8(r>>2) + 4h
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 bb = *reinterpret_cast<const float4 *>(&B1L[32 * mb + 8 * q + 4 * h]);
                const float us = us1;                                // the fp16 planes of W1 carry its plane scale
                acc1[mb][4 * q + 0] = ssp_f(fmaf(acc1[mb][4 * q + 0], us, bb.x));
                acc1[mb][4 * q + 1] = ssp_f(fmaf(acc1[mb][4 * q + 1], us, bb.y));
                acc1[mb][4 * q + 2] = ssp_f(fmaf(acc1[mb][4 * q + 2], us, bb.z));
                acc1[mb][4 * q + 3] = ssp_f(fmaf(acc1[mb][4 * q + 3], us, bb.w));
            }
        if (!CF && h1_out) {
            // streaming stores: h1 is not read again before the backward pass, W is read by the very next kernel — without the hint
            // the two 132 MB streams together overflow the 256 MiB Infinity Cache and the gather finds none of W there
            typedef float f4 __attribute__((ext_vector_type(4)));
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) {
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    *reinterpret_cast<float4 *>(&OT[l31 * FF_OP + 8 * q + 4 * h]) =
                        make_float4(acc1[mb][4 * q], acc1[mb][4 * q + 1], acc1[mb][4 * q + 2], acc1[mb][4 * q + 3]);
                wave_lds_fence();                              // slab rows are read by other lanes of this wavefront
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int r = 8 * i + (lane >> 3), c = 4 * (lane & 7);
                    const float4 o = *reinterpret_cast<const float4 *>(&OT[r * FF_OP + c]);
                    const f4 hv4 = {o.x, o.y, o.z, o.w};
                    if ((tile << 5) + r < E) __builtin_nontemporal_store(hv4, reinterpret_cast<f4 *>(h1_out + (size_t)((tile << 5) + r) * F + 32 * mb + c));
                }
                wave_lds_fence();                              // ... and rewritten for the next block
            }
        }

        // ---------------- GEMM2^T in groups of NG output row-blocks (bounds the live accumulators: 16*(MB + NG) registers)
        const float C = 0.5f * (cosf(__fdiv_rn(d * 3.14159265358979323846f, cutoff)) + 1.0f);
        constexpr int NG = MB >= 4 ? 2 : MB;
#pragma unroll
        for (int n0 = 0; n0 < MB; n0 += NG) {
            f32x16 acc2[NG];
#pragma unroll
            for (int nb = 0; nb < NG; ++nb)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc2[nb][r] = 0.f;
            float4 xg[CF ? NG : 1][4];
            if constexpr (CF) {                                        // the source's features for this lane's channels: requested BEFORE the group's MFMAs
#pragma unroll
                for (int nb = 0; nb < NG; ++nb)
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        xg[nb][q] = valid ? *reinterpret_cast<const float4 *>(xin + (size_t)src * F + 32 * (n0 + nb) + 8 * q + 4 * h) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
            {
#pragma unroll
                for (int ms = 0; ms < 2 * MB; ++ms) {
                    const int mb = ms >> 1, sgrp = ms & 1;
                    float hv[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) hv[j] = acc1[mb][8 * sgrp + j];
                    const int colp = 32 * mb + 16 * sgrp + 8 * h;
                    const _Float16 *W2H = reinterpret_cast<const _Float16 *>(W2L);
                    f16x8 q1, q2;
                    split2h(hv, q1, q2);
#pragma unroll
                    for (int nb = 0; nb < NG; ++nb) {
                        const int row = 32 * (n0 + nb) + l31;
                        const f16x8 p1 = *reinterpret_cast<const f16x8 *>(&W2H[(0 * F + row) * W2S + colp]);
                        const f16x8 p2 = *reinterpret_cast<const f16x8 *>(&W2H[(1 * F + row) * W2S + colp]);
                        acc2[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(p2, q1, acc2[nb], 0, 0, 0);
                        acc2[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(p1, q2, acc2[nb], 0, 0, 0);
                        acc2[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(p1, q1, acc2[nb], 0, 0, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            // epilogue: + b2, * C(d), store W[e, 32nb + 8q + 4h .. +3] (fp16 form: through the slab, whole lines)
#pragma unroll
            for (int nb = 0; nb < NG; ++nb) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 bb = *reinterpret_cast<const float4 *>(&B2L[32 * (n0 + nb) + 8 * q + 4 * h]);
                    float4 o;
                    const float us = us2;
                    o.x = fmaf(acc2[nb][4 * q + 0], us, bb.x) * C;
                    o.y = fmaf(acc2[nb][4 * q + 1], us, bb.y) * C;
                    o.z = fmaf(acc2[nb][4 * q + 2], us, bb.z) * C;
                    o.w = fmaf(acc2[nb][4 * q + 3], us, bb.w) * C;
                    if constexpr (CF) { o.x *= xg[nb][q].x; o.y *= xg[nb][q].y; o.z *= xg[nb][q].z; o.w *= xg[nb][q].w; }
                    *reinterpret_cast<float4 *>(&OT[l31 * FF_OP + 8 * q + 4 * h]) = o;
                }
                wave_lds_fence();
                if constexpr (CF) {
                    // segment sums: lane <-> channel (the lower half-wavefront), rows in edge order; a row's target is wave-uniform (read from
                    // the lane that owns the edge), so the segment test is a scalar branch.  One add to `out` per (target, tile).
                    // Both half-wavefronts walk: lanes 0..31 rows 0..15, lanes 32..63 rows 16..31, lane & 31 = channel.  When rows 15 and 16 belong to
                    // the same target the upper half does not flush its first segment: it hands the sum down (carry) and the lower half adds it to
                    // its last segment — one add per (target, tile), partial sums combined in a fixed order.
                    {
                        const int c = lane & 31, r0 = 16 * h;
                        float *ocol = Wout + 32 * (n0 + nb) + c;           // (Wout = out [n, F] in this mode)
                        // (every cross-lane read below is executed by ALL lanes and selected afterwards: a shuffle inside an arm of `h ? ... : ...`
                        // runs under half an EXEC mask and reads an inactive lane — undefined; it returned 0 and every upper half began on target 0)
                        const int t0 = __shfl(trg, 0, 64), t15 = __shfl(trg, 15, 64), t16 = __shfl(trg, 16, 64);
                        const bool joined = t15 == t16;                                      // wave-uniform
                        float accs = 0.f, carry = 0.f;
                        int cur = h ? t16 : t0;                                              // uniform per half: the branches below diverge by half only
                        bool first = true;                                                   // still inside the half's first segment
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int tl = __shfl(trg, r, 64), th = __shfl(trg, 16 + r, 64);
                            const int tr = h ? th : tl;
                            if (tr != cur) {
                                if (h && first && joined) carry = accs;
                                else if (cur >= 0) unsafeAtomicAdd(ocol + (size_t)cur * F, accs);
                                accs = 0.f; cur = tr; first = false;
                            }
                            accs += OT[(r0 + r) * FF_OP + c];
                        }
                        if (h && first && joined) { carry = accs; cur = -1; }               // the whole upper half continues the lower half's last target
                        const float down = __shfl(carry, (lane & 31) + 32, 64);              // upper half's carried sum of this channel
                        if (!h && joined) accs += down;
                        if (cur >= 0) unsafeAtomicAdd(ocol + (size_t)cur * F, accs);
                    }
                    wave_lds_fence();
                    continue;
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int r = 8 * i + (lane >> 3), c = 4 * (lane & 7);
                    const float4 o = *reinterpret_cast<const float4 *>(&OT[r * FF_OP + c]);
                    if ((tile << 5) + r < E) *reinterpret_cast<float4 *>(Wout + (size_t)((tile << 5) + r) * F + 32 * (n0 + nb) + c) = o;
                }
                wave_lds_fence();
            }
        }
    }
}

template <int F, int NT>
int launch_nt(const float *dist, const int *num_edges_dev, int max_edges, const float *offset, int Gs, float coeff,
              float cutoff, const float *w1, const float *b1, const float *w2, const float *b2, float *W, float *h1, hipStream_t s) {
    const size_t lds = ((size_t)(FF_NPL * F * W1S) / 2 + (size_t)(FF_NPL * F * (F + 8)) / 2 + 2 * F + GP + (NT / 64) * 32 * FF_OP) * 4;
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_filter_fused<F, false, NT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const int tiles = (max_edges + 31) / 32;
    int grid = (tiles + NT / 64 - 1) / (NT / 64);
    if (grid > 256) grid = 256;                           // one persistent workgroup per CU
    k_filter_fused<F, false, NT><<<grid, NT, lds, s>>>(dist, num_edges_dev, max_edges, offset, Gs, coeff, cutoff, w1, b1, w2, b2, W, h1);
    return hipGetLastError() == hipSuccess ? CONAN_OK : CONAN_E_LAUNCH;
}
template <int F>
int launch(const float *dist, const int *num_edges_dev, int max_edges, const float *offset, int Gs, float coeff,
           float cutoff, const float *w1, const float *b1, const float *w2, const float *b2, float *W, float *h1,
           hipStream_t s) {
    return launch_nt<F, FF_THREADS>(dist, num_edges_dev, max_edges, offset, Gs, coeff, cutoff, w1, b1, w2, b2, W, h1, s);
}

template <int F>
int launch_cf(const float *x, const float *dist, const int *col, const int *tgt, const int *num_edges_dev, int max_edges, const float *offset, int Gs,
              float coeff, float cutoff, const float *w1, const float *b1, const float *w2, const float *b2, int num_atoms, float *out, hipStream_t s) {
    const size_t lds = ((size_t)(FF_NPL * F * W1S) / 2 + (size_t)(FF_NPL * F * (F + 8)) / 2 + 2 * F + GP + (FF_THREADS / 64) * 32 * FF_OP) * 4;
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_filter_fused<F, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (hipMemsetAsync(out, 0, (size_t)num_atoms * F * sizeof(float), s) != hipSuccess) return CONAN_E_LAUNCH;      // every target's row starts at 0 (targets without edges stay there)
    k_filter_fused<F, true><<<256, FF_THREADS, lds, s>>>(dist, num_edges_dev, max_edges, offset, Gs, coeff, cutoff, w1, b1, w2, b2, out, nullptr, x, col, tgt);
    return hipGetLastError() == hipSuccess ? CONAN_OK : CONAN_E_LAUNCH;
}

}  // namespace

extern "C" {

int conan_filter_cfconv_fwd_supported(int num_gaussians, int num_filters) { return (num_gaussians >= 2 && num_gaussians <= GP && num_filters == 128) ? 1 : 0; }

int conan_filter_cfconv_fwd(const float *x, const float *dist, const int *col, const int *tgt, const int *num_edges_dev, int max_edges,
                            const float *offset, int num_gaussians, float coeff, float cutoff, int num_filters, const float *w1,
                            const float *b1, const float *w2, const float *b2, int num_atoms, float *out, void *stream) {
    if (!x || !dist || !col || !tgt || !offset || !w1 || !b1 || !w2 || !b2 || !out || max_edges < 0 || num_atoms < 0) return CONAN_E_BADARG;
    if (!conan_filter_cfconv_fwd_supported(num_gaussians, num_filters)) return CONAN_E_UNSUPPORTED;
    if (num_atoms == 0) return CONAN_OK;
    hipStream_t s = as_stream(stream);
    if (max_edges == 0) return hipMemsetAsync(out, 0, (size_t)num_atoms * num_filters * sizeof(float), s) == hipSuccess ? CONAN_OK : CONAN_E_LAUNCH;
    return launch_cf<128>(x, dist, col, tgt, num_edges_dev, max_edges, offset, num_gaussians, coeff, cutoff, w1, b1, w2, b2, num_atoms, out, s);
}

int conan_filter_fused_supported(int num_gaussians, int num_filters) {
    return (num_gaussians >= 2 && num_gaussians <= GP && (num_filters == 32 || num_filters == 64 || num_filters == 128)) ? 1 : 0;
}

int conan_filter_fwd(const float *dist, const int *num_edges_dev, int max_edges, const float *offset, int num_gaussians,
                     float coeff, float cutoff, int num_filters, const float *w1, const float *b1, const float *w2,
                     const float *b2, float *W, float *h1_out, void *stream) {
    if (!dist || !offset || !w1 || !b1 || !w2 || !b2 || !W || max_edges < 0) return CONAN_E_BADARG;
    if (!conan_filter_fused_supported(num_gaussians, num_filters)) return CONAN_E_UNSUPPORTED;
    if (max_edges == 0) return CONAN_OK;
    hipStream_t s = as_stream(stream);
#define CONAN_FF(FV) return launch<FV>(dist, num_edges_dev, max_edges, offset, num_gaussians, coeff, cutoff, w1, b1, w2, b2, W, h1_out, s)
    switch (num_filters) {
        case 32: CONAN_FF(32);
        case 64: CONAN_FF(64);
        default: CONAN_FF(128);
    }
#undef CONAN_FF
}

}  // extern "C"
