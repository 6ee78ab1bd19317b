// ViSNet backward kernels for gfx950 (gradients of the kernels in visnet.hip; same layouts and citations).
// Node-level gradients that receive contributions from many edges are accumulated WITHOUT atomics: one wavefront per
// node walks the node's by-target CSR row (gradients flowing to the target side) or its by-source list
// (t_rowptr / t_eid: gradients flowing to the source side) in a fixed order => bitwise reproducible.
#include "common.h"

namespace {

__device__ __forceinline__ float sigmoid_f(float v) { return __builtin_amdgcn_rcpf(1.0f + __expf(-v)); }      // hardware reciprocal, 1 ulp
__device__ __forceinline__ float silu_f(float v) { return v * sigmoid_f(v); }
__device__ __forceinline__ float dsilu_f(float v) { const float s = sigmoid_f(v); return s * (1.0f + v * (1.0f - s)); }
__device__ __forceinline__ float cos_cutoff(float d, float cutoff) {
    return d < cutoff ? 0.5f * (cosf(__fdiv_rn(d * 3.14159265358979323846f, cutoff)) + 1.0f) : 0.0f;
}
inline int nblk(long long n) { long long b = (n + 255) / 256; return (int)(b < 1 ? 1 : (b > 4096 ? 4096 : b)); }

// ---------------------------------------------------------------------------------------------- SiLU
__global__ void k_silu_fwd(const float *__restrict__ x, int rows, int width, const int *__restrict__ m_dev, float *__restrict__ y) {
    long long n = (long long)(m_dev ? min(rows, *m_dev) : rows) * width;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += stride) y[t] = silu_f(x[t]);
}
__global__ void k_silu_bwd(const float *__restrict__ x, const float *__restrict__ dy, int rows, int width, const int *__restrict__ m_dev,
                           float *__restrict__ dx) {
    long long n = (long long)(m_dev ? min(rows, *m_dev) : rows) * width;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += stride) dx[t] = dy[t] * dsilu_f(x[t]);
}

__global__ void k_split2(const float *__restrict__ in, int Ha, int Hb, long long rows, float *__restrict__ a, float *__restrict__ b) {
    const int Ho = Ha + Hb;
    const long long n = rows * Ho, stride = (long long)gridDim.x * blockDim.x;
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += stride) {
        const long long r = t / Ho; const int c = (int)(t - r * Ho);
        if (c < Ha) a[r * Ha + c] = in[t]; else b[r * Hb + (c - Ha)] = in[t];
    }
}

__global__ void k_rowsum(const float *__restrict__ x, int rows, int width, float *__restrict__ out) {
    for (int r = blockIdx.x * blockDim.x + threadIdx.x; r < rows; r += gridDim.x * blockDim.x) {
        float s = 0.f;
        for (int c = 0; c < width; ++c) s += x[(size_t)r * width + c];
        out[r] = s;
    }
}

// ---------------------------------------------------------------------------------------------- EdgeEmbedding backward
// CPL consecutive channels of one row as ONE load / store (float2 for CPL = 2: c0 is even and every row starts at a multiple of H floats from a
// 256-byte aligned allocation; the compiler cannot prove that and would issue two dword instructions)
template <int CPL>
__device__ __forceinline__ void vld(const float *__restrict__ p, float (&r)[CPL]) {
    if constexpr (CPL == 4) { const float4 t = *reinterpret_cast<const float4 *>(p); r[0] = t.x; r[1] = t.y; r[2] = t.z; r[3] = t.w; }
    else if constexpr (CPL == 2) { const float2 t = *reinterpret_cast<const float2 *>(p); r[0] = t.x; r[1] = t.y; }
    else {
#pragma unroll
        for (int u = 0; u < CPL; ++u) r[u] = p[u];
    }
}
template <int CPL, bool HALF>
__device__ __forceinline__ void vfold(float (&a)[CPL]) {      // HALF: even entries (lanes 0-31) + odd entries (lanes 32-63), fixed order
    if constexpr (HALF) {
#pragma unroll
        for (int u = 0; u < CPL; ++u) a[u] += __shfl_xor(a[u], 32, 64);
    }
}
template <int CPL>
__device__ __forceinline__ void vst(float *__restrict__ p, const float (&r)[CPL]) {
    if constexpr (CPL == 4) *reinterpret_cast<float4 *>(p) = make_float4(r[0], r[1], r[2], r[3]);
    else if constexpr (CPL == 2) *reinterpret_cast<float2 *>(p) = make_float2(r[0], r[1]);
    else {
#pragma unroll
        for (int u = 0; u < CPL; ++u) p[u] = r[u];
    }
}

#ifndef CONAN_V_EB
#define CONAN_V_EB 4
#endif
constexpr bool V_HALF = true;     // H = 128: a half-wavefront per edge (visnet.hip)
constexpr int VB_EB = CONAN_V_EB;
#ifndef CONAN_VB_RUN
#define CONAN_VB_RUN 16
#endif
constexpr int VB_RUN = CONAN_VB_RUN;     // edges per wavefront in the kernels that walk runs of consecutive edges
// dp[e] = (x_i + x_j) * df[e]      (runs of VB_RUN consecutive edges per wavefront, as k_edge_embed)
template <int CPL, bool HALF = false>
__global__ void __launch_bounds__(256) k_edge_embed_bwd_p(const float *__restrict__ x, const float *__restrict__ df, const int *__restrict__ col,
                                                          const int *__restrict__ tgt, const int *__restrict__ ne_dev, int max_edges, int H,
                                                          float *__restrict__ dp) {
    const int E = min(*ne_dev, max_edges);
    const int lane = threadIdx.x & 63;
    const int hf = HALF ? lane >> 5 : 0, ll = HALF ? (lane & 31) : lane;      // HALF: a half-wavefront per edge (H = 32 CPL), two edges per step
    constexpr int ES = HALF ? 2 : 1;
    const int wave = __builtin_amdgcn_readfirstlane((int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6)), nw = (gridDim.x * blockDim.x) >> 6;
    for (int base = wave * VB_RUN; base < E; base += nw * VB_RUN) {
        const int cnt = min(VB_RUN, E - base);
        const int my_j = lane < cnt ? col[base + lane] : 0, my_i = lane < cnt ? tgt[base + lane] : 0;
        for (int cp = 0; cp < H; cp += (HALF ? 32 : 64) * CPL) {
            const int c0 = cp + ll * CPL; const bool on = c0 < H; const int cl = on ? c0 : 0;
            for (int tq = 0; tq < cnt; tq += ES * VB_EB) {
                float xi[VB_EB][CPL], xj[VB_EB][CPL], gv[VB_EB][CPL];
#pragma unroll
                for (int b = 0; b < VB_EB; ++b) {
                    const int tt = min(tq + ES * b + hf, cnt - 1);
                    const size_t j = (size_t)__shfl(my_j, tt, 64), i = (size_t)__shfl(my_i, tt, 64);
                    vld<CPL>(x + i * H + cl, xi[b]); vld<CPL>(x + j * H + cl, xj[b]); vld<CPL>(df + (size_t)(base + tt) * H + cl, gv[b]);
                }
#pragma unroll
                for (int b = 0; b < VB_EB; ++b) {
                    if (tq + ES * b + hf >= cnt) continue;
                    float o[CPL];
#pragma unroll
                    for (int u = 0; u < CPL; ++u) o[u] = (xi[b][u] + xj[b][u]) * gv[b][u];
                    if (on) vst<CPL>(dp + (size_t)(base + tq + ES * b + hf) * H + c0, o);
                }
            }
        }
    }
}
// dx[i] = sum_{e in row(i)} df[e]*p[e] + sum_{e in srclist(i)} df[e]*p[e]
template <int CPL, bool HALF = false>
__global__ void __launch_bounds__(256) k_edge_embed_bwd_x(const float *__restrict__ p, const float *__restrict__ df, const int *__restrict__ rowptr,
                                                          const int *__restrict__ t_rowptr, const int *__restrict__ t_eid, int n, int H,
                                                          float *__restrict__ dx) {
    const int lane = threadIdx.x & 63;
    const int hf = HALF ? lane >> 5 : 0, ll = HALF ? (lane & 31) : lane;      // HALF: a half-wavefront per edge (H = 32 CPL), two edges per step
    constexpr int ES = HALF ? 2 : 1;
    const int wave = __builtin_amdgcn_readfirstlane((int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6)), nw = (gridDim.x * blockDim.x) >> 6;
    for (int i = wave; i < n; i += nw)
        for (int cp = 0; cp < H; cp += (HALF ? 32 : 64) * CPL) {
            const int c0 = cp + ll * CPL; const bool on = c0 < H; const int cl = on ? c0 : 0;
            float a[CPL];
#pragma unroll
            for (int u = 0; u < CPL; ++u) a[u] = 0.f;
            const int e0 = rowptr[i], e1 = rowptr[i + 1];
            for (int e = e0; e < e1; e += ES * VB_EB) {                             // the row itself: consecutive edges, no indices
                float g[VB_EB][CPL], q[VB_EB][CPL];
#pragma unroll
                for (int b = 0; b < VB_EB; ++b) { const size_t ee = (size_t)min(e + ES * b + hf, e1 - 1); vld<CPL>(df + ee * H + cl, g[b]); vld<CPL>(p + ee * H + cl, q[b]); }
#pragma unroll
                for (int b = 0; b < VB_EB; ++b) {
                    if (e + ES * b + hf >= e1) continue;
#pragma unroll
                    for (int u = 0; u < CPL; ++u) a[u] += g[b][u] * q[b][u];
                }
            }
            const int s0 = t_rowptr[i], s1 = t_rowptr[i + 1];
            for (int base = s0; base < s1; base += 64) {                            // the by-source list: edge ids handed out per lane
                const int cnt = min(64, s1 - base);
                const int my_e = lane < cnt ? t_eid[base + lane] : 0;
                for (int tq = 0; tq < cnt; tq += ES * VB_EB) {
                    float g[VB_EB][CPL], q[VB_EB][CPL];
#pragma unroll
                    for (int b = 0; b < VB_EB; ++b) {
                        const size_t ee = (size_t)__shfl(my_e, min(tq + ES * b + hf, cnt - 1), 64);
                        vld<CPL>(df + ee * H + cl, g[b]); vld<CPL>(p + ee * H + cl, q[b]);
                    }
#pragma unroll
                    for (int b = 0; b < VB_EB; ++b) {
                        if (tq + ES * b + hf >= cnt) continue;
#pragma unroll
                        for (int u = 0; u < CPL; ++u) a[u] += g[b][u] * q[b][u];
                    }
                }
            }
            vfold<CPL, HALF>(a);
            if (on && hf == 0) vst<CPL>(dx + (size_t)i * H + c0, a);
        }
}

// ---------------------------------------------------------------------------------------------- LayerNorm backward
// dx per row (one wavefront per row); stats[r] = (mean, rstd) saved for the parameter-gradient pass
// dres (nullable): a second gradient of x (its residual use next to the LayerNorm, :583,:659) added here instead of by autograd
__global__ void __launch_bounds__(256) k_layernorm_bwd_x(const float *__restrict__ x, const float *__restrict__ gamma, const float *__restrict__ dy,
                                                         const float *__restrict__ dres, int rows, int H, float eps, float *__restrict__ dx,
                                                         float *__restrict__ stats) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6)), nw = (gridDim.x * blockDim.x) >> 6;
    for (int r = wave; r < rows; r += nw) {
        const float *xr = x + (size_t)r * H, *gr = dy + (size_t)r * H;
        float s = 0.f;
        for (int c = lane; c < H; c += 64) s += xr[c];
        const float mean = wave_sum(s) / (float)H;
        float v = 0.f;
        for (int c = lane; c < H; c += 64) { const float d = xr[c] - mean; v += d * d; }
        const float rstd = 1.0f / sqrtf(wave_sum(v) / (float)H + eps);
        float s1 = 0.f, s2 = 0.f;                        // mean(dxhat), mean(dxhat * xhat)
        for (int c = lane; c < H; c += 64) { const float dxh = gr[c] * gamma[c], xh = (xr[c] - mean) * rstd; s1 += dxh; s2 += dxh * xh; }
        s1 = wave_sum(s1) / (float)H; s2 = wave_sum(s2) / (float)H;
        for (int c = lane; c < H; c += 64) {
            const float dxh = gr[c] * gamma[c], xh = (xr[c] - mean) * rstd;
            dx[(size_t)r * H + c] = rstd * (dxh - s1 - xh * s2) + (dres ? dres[(size_t)r * H + c] : 0.f);
        }
        if (lane == 0) { stats[2 * r] = mean; stats[2 * r + 1] = rstd; }
    }
}
// per-chunk partial dgamma / dbeta: thread <-> (column, one of LN_RL row lanes); a row lane takes rows r0 + lane, r0 + lane + LN_RL, ... of
// the chunk in order, the LN_RL partial sums are combined through LDS in lane order, the chunks in chunk order (bitwise reproducible).
// (Round 3: was one thread per column walking all 256 rows of its chunk — 80 workgroups of two wavefronts for 20 k rows, 72 us.)
constexpr int LN_CHUNK = 256, LN_RL = 8;
__global__ void __launch_bounds__(128 * LN_RL) k_layernorm_bwd_p(const float *__restrict__ x, const float *__restrict__ dy, const float *__restrict__ stats,
                                                                 int rows, int H, float *__restrict__ slabs) {
    __shared__ float sg[LN_RL][128], sb[LN_RL][128];
    const int chunk = blockIdx.x, cl = threadIdx.x & 127, rl = threadIdx.x >> 7, c = blockIdx.y * 128 + cl;
    const int r0 = chunk * LN_CHUNK, r1 = min(rows, r0 + LN_CHUNK);
    float dg = 0.f, db = 0.f;
    if (c < H)
        for (int r = r0 + rl; r < r1; r += LN_RL) {
            const float g = dy[(size_t)r * H + c];
            dg += g * (x[(size_t)r * H + c] - stats[2 * r]) * stats[2 * r + 1];
            db += g;
        }
    sg[rl][cl] = dg; sb[rl][cl] = db;
    __syncthreads();
    if (rl == 0 && c < H) {
#pragma unroll
        for (int q = 1; q < LN_RL; ++q) { dg += sg[q][cl]; db += sb[q][cl]; }
        slabs[((size_t)chunk * 2) * H + c] = dg;
        slabs[((size_t)chunk * 2 + 1) * H + c] = db;
    }
}
__global__ void k_layernorm_bwd_reduce(const float *__restrict__ slabs, int chunks, int H, float *__restrict__ dgamma, float *__restrict__ dbeta) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= H) return;
    float a = 0.f, b = 0.f;
    for (int ch = 0; ch < chunks; ++ch) { a += slabs[((size_t)ch * 2) * H + c]; b += slabs[((size_t)ch * 2 + 1) * H + c]; }
    dgamma[c] = a; dbeta[c] = b;
}

// ---------------------------------------------------------------------------------------------- vec_dot backward
// dvp[a,sp,0:H] = dout * vp[a,sp,H:2H]; dvp[a,sp,H:2H] = dout * vp[a,sp,0:H]; dvp[a,sp,2H:3H] = 0
__global__ void k_vecdot_bwd(const float *__restrict__ vp, const float *__restrict__ dout, int n, int H, float *__restrict__ dvp) {
    const long long tot = (long long)n * 3 * H, stride = (long long)gridDim.x * blockDim.x;
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < tot; t += stride) {
        const long long row = t / H; const int c = (int)(t - row * H);          // row = a*3 + sp
        const int a = (int)(row / 3);
        const float g = dout[(size_t)a * H + c];
        const float *vr = vp + (size_t)row * 3 * H;
        float *dr = dvp + (size_t)row * 3 * H;
        dr[c] = g * vr[H + c]; dr[H + c] = g * vr[c]; dr[2 * H + c] = 0.f;
    }
}

// ---------------------------------------------------------------------------------------------- attention message backward
// Shared per-edge recomputation: lane owns CPL channels of head (lane / lph)
template <int CPL>
__device__ __forceinline__ void attn_edge(const float *qi, const float *kj, const float *vj, const float *dke, const float *dve, const float *dm,
                                          float cut, int lph, float &attn, float &da) {
    float part = 0.f, t1 = 0.f;
#pragma unroll
    for (int u = 0; u < CPL; ++u) { part += qi[u] * kj[u] * dke[u]; t1 += dm[u] * vj[u] * dve[u]; }
    for (int o = 1; o < lph; o <<= 1) { part += __shfl_xor(part, o, 64); t1 += __shfl_xor(t1, o, 64); }
    const float sg = sigmoid_f(part);
    attn = part * sg * cut;                                    // SiLU(a) * C
    da = t1 * cut * (sg * (1.0f + part * (1.0f - sg)));        // d attn * C * SiLU'(a)
}

// Round 3 (all row-walking kernels below): a row's indices (and what hangs off them per edge: target, cutoff, unit vector) are fetched once,
// one edge per lane, and handed out with cross-lane reads; the rows of VB_EB edges are requested before the first is used.  The loops were
// chains of two or three dependent round trips per edge (2.3-3 TB/s); the sums still run in list order (bitwise-equal results).

// target side: dq[i] (sum over row i), and the edge gradients d dk[e], d dv[e]
template <int CPL, bool HALF = false>
__global__ void __launch_bounds__(256) k_attn_bwd_target(const float *__restrict__ q, const float *__restrict__ k, const float *__restrict__ v,
                                                         const float *__restrict__ dk, const float *__restrict__ dv, const float *__restrict__ dvmsg,
                                                         const float *__restrict__ dxagg, const int *__restrict__ rowptr, const int *__restrict__ col,
                                                         const float *__restrict__ dist, float cutoff, int n, int H, int lph, int pre,
                                                         float *__restrict__ dq, float *__restrict__ ddk, float *__restrict__ ddv) {
    const int lane = threadIdx.x & 63;
    const int hf = HALF ? lane >> 5 : 0, ll = HALF ? (lane & 31) : lane;      // HALF: a half-wavefront per edge (H = 32 CPL), two edges per step
    constexpr int ES = HALF ? 2 : 1;
    const int wave = __builtin_amdgcn_readfirstlane((int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6)), nw = (gridDim.x * blockDim.x) >> 6;
    const int c0 = (int)blockIdx.y * (HALF ? 32 : 64) * CPL + ll * CPL;      // blockIdx.y: channel block of whole heads (H > 128, visnet.hip)
    const bool on = c0 < H;
    const int cl = on ? c0 : 0;                                       // idle lanes read column 0 and store nothing
    for (int i = wave; i < n; i += nw) {
        float qi[CPL], gx[CPL], acc[CPL];
#pragma unroll
        for (int u = 0; u < CPL; ++u) { qi[u] = on ? q[(size_t)i * H + cl + u] : 0.f; gx[u] = dxagg[(size_t)i * H + cl + u]; acc[u] = 0.f; }
        const int e0 = rowptr[i], e1 = rowptr[i + 1];
        for (int base = e0; base < e1; base += 64) {
            const int cnt = min(64, e1 - base);
            const int my_j = lane < cnt ? col[base + lane] : 0;
            const float my_c = lane < cnt ? cos_cutoff(dist[base + lane], cutoff) : 0.f;
            for (int t = 0; t < cnt; t += ES * VB_EB) {
                float kj[VB_EB][CPL], vj[VB_EB][CPL], dke[VB_EB][CPL], dve[VB_EB][CPL], dm[VB_EB][CPL];
#pragma unroll
                for (int b = 0; b < VB_EB; ++b) {
                    const int tt = min(t + ES * b + hf, cnt - 1);
                    const int j = __shfl(my_j, tt, 64);
                    const size_t e = (size_t)(base + tt);
                    vld<CPL>(k + (size_t)j * H + cl, kj[b]); vld<CPL>(v + (size_t)j * H + cl, vj[b]);
                    vld<CPL>(dk + e * H + cl, dke[b]); vld<CPL>(dv + e * H + cl, dve[b]); vld<CPL>(dvmsg + e * H + cl, dm[b]);
                }
#pragma unroll
                for (int b = 0; b < VB_EB; ++b) {
                    const float cutb = __shfl(my_c, min(t + ES * b + hf, cnt - 1), 64);      // before the halves diverge
                    if (t + ES * b + hf >= cnt) continue;
                    const size_t e = (size_t)(base + t + ES * b + hf);
                    float sk[CPL], sv[CPL];
#pragma unroll
                    for (int u = 0; u < CPL; ++u) {
                        sk[u] = 1.f; sv[u] = 1.f;
                        if (pre) { sk[u] = dsilu_f(dke[b][u]); sv[u] = dsilu_f(dve[b][u]); dke[b][u] = silu_f(dke[b][u]); dve[b][u] = silu_f(dve[b][u]); }   // pre-activations in, gradients w.r.t. them out
                        dm[b][u] = on ? dm[b][u] + gx[u] : 0.f;
                        if (!on) { kj[b][u] = 0.f; vj[b][u] = 0.f; dke[b][u] = 0.f; dve[b][u] = 0.f; }
                    }
                    float attn, da;
                    attn_edge<CPL>(qi, kj[b], vj[b], dke[b], dve[b], dm[b], cutb, lph, attn, da);
                    float o1[CPL], o2[CPL];
#pragma unroll
                    for (int u = 0; u < CPL; ++u) {
                        o1[u] = dm[b][u] * vj[b][u] * attn * sv[u]; o2[u] = da * qi[u] * kj[b][u] * sk[u];
                        acc[u] += da * kj[b][u] * dke[b][u];
                    }
                    if (on) { vst<CPL>(ddv + e * H + c0, o1); vst<CPL>(ddk + e * H + c0, o2); }
                }
            }
        }
        vfold<CPL, HALF>(acc);
        if (on && hf == 0) vst<CPL>(dq + (size_t)i * H + c0, acc);
    }
}
// source side: dk[j], dv[j] (sum over the by-source list of j)
template <int CPL, bool HALF = false>
__global__ void __launch_bounds__(256) k_attn_bwd_source(const float *__restrict__ q, const float *__restrict__ k, const float *__restrict__ v,
                                                         const float *__restrict__ dk, const float *__restrict__ dv, const float *__restrict__ dvmsg,
                                                         const float *__restrict__ dxagg, const int *__restrict__ t_rowptr, const int *__restrict__ t_eid,
                                                         const int *__restrict__ tgt, const float *__restrict__ dist, float cutoff, int n, int H, int lph,
                                                         int pre, float *__restrict__ dkn, float *__restrict__ dvn) {
    const int lane = threadIdx.x & 63;
    const int hf = HALF ? lane >> 5 : 0, ll = HALF ? (lane & 31) : lane;      // HALF: a half-wavefront per edge (H = 32 CPL), two edges per step
    constexpr int ES = HALF ? 2 : 1;
    const int wave = __builtin_amdgcn_readfirstlane((int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6)), nw = (gridDim.x * blockDim.x) >> 6;
    const int c0 = (int)blockIdx.y * (HALF ? 32 : 64) * CPL + ll * CPL;
    const bool on = c0 < H;
    const int cl = on ? c0 : 0;
    for (int j = wave; j < n; j += nw) {
        float kj[CPL], vj[CPL], ak[CPL], av[CPL];
#pragma unroll
        for (int u = 0; u < CPL; ++u) { kj[u] = on ? k[(size_t)j * H + cl + u] : 0.f; vj[u] = on ? v[(size_t)j * H + cl + u] : 0.f; ak[u] = 0.f; av[u] = 0.f; }
        const int s0 = t_rowptr[j], s1 = t_rowptr[j + 1];
        for (int base = s0; base < s1; base += 64) {
            const int cnt = min(64, s1 - base);
            const int my_e = lane < cnt ? t_eid[base + lane] : 0;
            const int my_i = lane < cnt ? tgt[my_e] : 0;
            const float my_c = lane < cnt ? cos_cutoff(dist[my_e], cutoff) : 0.f;
            for (int t = 0; t < cnt; t += ES * VB_EB) {
                float qi[VB_EB][CPL], dke[VB_EB][CPL], dve[VB_EB][CPL], dm[VB_EB][CPL], gx[VB_EB][CPL];
#pragma unroll
                for (int b = 0; b < VB_EB; ++b) {
                    const int tt = min(t + ES * b + hf, cnt - 1);
                    const size_t e = (size_t)__shfl(my_e, tt, 64), i = (size_t)__shfl(my_i, tt, 64);
                    vld<CPL>(q + i * H + cl, qi[b]); vld<CPL>(dxagg + i * H + cl, gx[b]);
                    vld<CPL>(dk + e * H + cl, dke[b]); vld<CPL>(dv + e * H + cl, dve[b]); vld<CPL>(dvmsg + e * H + cl, dm[b]);
                }
#pragma unroll
                for (int b = 0; b < VB_EB; ++b) {
                    const float cutb = __shfl(my_c, min(t + ES * b + hf, cnt - 1), 64);      // before the halves diverge
                    if (t + ES * b + hf >= cnt) continue;
#pragma unroll
                    for (int u = 0; u < CPL; ++u) {
                        if (pre) { dke[b][u] = silu_f(dke[b][u]); dve[b][u] = silu_f(dve[b][u]); }
                        dm[b][u] = on ? dm[b][u] + gx[b][u] : 0.f;
                        if (!on) { qi[b][u] = 0.f; dke[b][u] = 0.f; dve[b][u] = 0.f; }
                    }
                    float attn, da;
                    attn_edge<CPL>(qi[b], kj, vj, dke[b], dve[b], dm[b], cutb, lph, attn, da);
#pragma unroll
                    for (int u = 0; u < CPL; ++u) { ak[u] += da * qi[b][u] * dke[b][u]; av[u] += dm[b][u] * dve[b][u] * attn; }
                }
            }
        }
        vfold<CPL, HALF>(ak); vfold<CPL, HALF>(av);
        if (on && hf == 0) { vst<CPL>(dkn + (size_t)j * H + c0, ak); vst<CPL>(dvn + (size_t)j * H + c0, av); }
    }
}

// ---------------------------------------------------------------------------------------------- vector aggregate backward
// ds[e] = [ sum_sp dvagg[tgt,sp]*vec[src,sp] | sum_sp dvagg[tgt,sp]*d_e[sp] ]
template <int CPL, bool HALF = false>
__global__ void __launch_bounds__(256) k_vec_aggregate_bwd_s(const float *__restrict__ vec, const float *__restrict__ dvagg, const float *__restrict__ dvec3,
                                                             const int *__restrict__ col, const int *__restrict__ tgt, const int *__restrict__ ne_dev,
                                                             int max_edges, int H, const float *__restrict__ s_pre, float *__restrict__ ds) {
    // one wavefront per run of VB_RUN consecutive edges, indices and unit vectors handed out per lane, VB_EB edges in flight (see k_edge_update)
    const int E = min(*ne_dev, max_edges);
    const int lane = threadIdx.x & 63;
    const int hf = HALF ? lane >> 5 : 0, ll = HALF ? (lane & 31) : lane;      // HALF: a half-wavefront per edge (H = 32 CPL), two edges per step
    constexpr int ES = HALF ? 2 : 1;
    const int wave = __builtin_amdgcn_readfirstlane((int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6)), nw = (gridDim.x * blockDim.x) >> 6;
    for (int base = wave * VB_RUN; base < E; base += nw * VB_RUN) {
        const int cnt = min(VB_RUN, E - base);
        const int my_j = lane < cnt ? col[base + lane] : 0, my_i = lane < cnt ? tgt[base + lane] : 0;
        float my_d[3];
#pragma unroll
        for (int sp = 0; sp < 3; ++sp) my_d[sp] = lane < cnt ? dvec3[(size_t)(base + lane) * 3 + sp] : 0.f;
        for (int cp = 0; cp < H; cp += (HALF ? 32 : 64) * CPL) {
            const int c0 = cp + ll * CPL; const bool on = c0 < H; const int cl = on ? c0 : 0;
            for (int tq = 0; tq < cnt; tq += ES * VB_EB) {
                float g[VB_EB][3][CPL], vj[VB_EB][3][CPL], p1[VB_EB][CPL], p2[VB_EB][CPL];
#pragma unroll
                for (int b = 0; b < VB_EB; ++b) {
                    const int tt = min(tq + ES * b + hf, cnt - 1);
                    const size_t j = (size_t)__shfl(my_j, tt, 64), i = (size_t)__shfl(my_i, tt, 64), e = (size_t)(base + tt);
#pragma unroll
                    for (int sp = 0; sp < 3; ++sp) { vld<CPL>(dvagg + (i * 3 + sp) * H + cl, g[b][sp]); vld<CPL>(vec + (j * 3 + sp) * H + cl, vj[b][sp]); }
                    if (s_pre) { vld<CPL>(s_pre + e * 2 * H + cl, p1[b]); vld<CPL>(s_pre + e * 2 * H + H + cl, p2[b]); }
                    else {
#pragma unroll
                        for (int u = 0; u < CPL; ++u) { p1[b][u] = 0.f; p2[b][u] = 0.f; }
                    }
                }
#pragma unroll
                for (int b = 0; b < VB_EB; ++b) {
                    const float d0 = __shfl(my_d[0], min(tq + ES * b + hf, cnt - 1), 64), d1 = __shfl(my_d[1], min(tq + ES * b + hf, cnt - 1), 64), d2 = __shfl(my_d[2], min(tq + ES * b + hf, cnt - 1), 64);      // before the halves diverge
                    if (tq + ES * b + hf >= cnt) continue;
                    const size_t e = (size_t)(base + tq + ES * b + hf);
                    float r1[CPL], r2[CPL];
#pragma unroll
                    for (int u = 0; u < CPL; ++u) {
                        const float g0 = g[b][0][u], g1 = g[b][1][u], g2 = g[b][2][u];
                        r1[u] = g0 * vj[b][0][u] + g1 * vj[b][1][u] + g2 * vj[b][2][u];
                        r2[u] = g0 * d0 + g1 * d1 + g2 * d2;
                        if (s_pre) { r1[u] *= dsilu_f(p1[b][u]); r2[u] *= dsilu_f(p2[b][u]); }   // gradient w.r.t. the pre-activation
                    }
                    if (on) { vst<CPL>(ds + e * 2 * H + c0, r1); vst<CPL>(ds + e * 2 * H + H + c0, r2); }
                }
            }
        }
    }
}
// dvec[j,sp] = sum_{e in srclist(j)} dvagg[tgt_e,sp] * s1_e
template <int CPL, bool HALF = false>
__global__ void __launch_bounds__(256) k_vec_aggregate_bwd_v(const float *__restrict__ s, const float *__restrict__ dvagg, const int *__restrict__ t_rowptr,
                                                             const int *__restrict__ t_eid, const int *__restrict__ tgt, int n, int H, int pre,
                                                             float *__restrict__ dvec) {
    const int lane = threadIdx.x & 63;
    const int hf = HALF ? lane >> 5 : 0, ll = HALF ? (lane & 31) : lane;      // HALF: a half-wavefront per edge (H = 32 CPL), two edges per step
    constexpr int ES = HALF ? 2 : 1;
    const int wave = __builtin_amdgcn_readfirstlane((int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6)), nw = (gridDim.x * blockDim.x) >> 6;
    for (int j = wave; j < n; j += nw)
        for (int cp = 0; cp < H; cp += (HALF ? 32 : 64) * CPL) {
            const int c0 = cp + ll * CPL; const bool on = c0 < H; const int cl = on ? c0 : 0;                        // H = 64 CPL: one pass; every lane walks the row (idle ones: column 0, no stores)
            float a0[CPL], a1[CPL], a2[CPL];
#pragma unroll
            for (int u = 0; u < CPL; ++u) { a0[u] = 0.f; a1[u] = 0.f; a2[u] = 0.f; }
            const int q0 = t_rowptr[j], q1 = t_rowptr[j + 1];
            for (int base = q0; base < q1; base += 64) {
                const int cnt = min(64, q1 - base);
                const int my_e = lane < cnt ? t_eid[base + lane] : 0;
                const int my_i = lane < cnt ? tgt[my_e] : 0;
                for (int t = 0; t < cnt; t += ES * VB_EB) {
                    float s1[VB_EB][CPL], g[VB_EB][3][CPL];
#pragma unroll
                    for (int b = 0; b < VB_EB; ++b) {
                        const int tt = min(t + ES * b + hf, cnt - 1);
                        const size_t e = (size_t)__shfl(my_e, tt, 64), i = (size_t)__shfl(my_i, tt, 64);
                        vld<CPL>(s + e * 2 * H + cl, s1[b]);
#pragma unroll
                        for (int sp = 0; sp < 3; ++sp) vld<CPL>(dvagg + (i * 3 + sp) * H + cl, g[b][sp]);
                    }
#pragma unroll
                    for (int b = 0; b < VB_EB; ++b) {
                        if (t + ES * b + hf >= cnt) continue;
#pragma unroll
                        for (int u = 0; u < CPL; ++u) {
                            const float x1 = pre ? silu_f(s1[b][u]) : s1[b][u];
                            a0[u] += g[b][0][u] * x1; a1[u] += g[b][1][u] * x1; a2[u] += g[b][2][u] * x1;
                        }
                    }
                }
            }
            float *o = dvec + (size_t)j * 3 * H;
            vfold<CPL, HALF>(a0); vfold<CPL, HALF>(a1); vfold<CPL, HALF>(a2);
            if (on && hf == 0) { vst<CPL>(o + c0, a0); vst<CPL>(o + H + c0, a1); vst<CPL>(o + 2 * H + c0, a2); }
        }
}

// ---------------------------------------------------------------------------------------------- node update backward
// forward: xo = x + vdot*o2 + o3 ; veco[sp] = vec[sp] + vec3[sp]*o1 + vagg[sp]      (dx = dxo, dvec = dveco, dvagg = dveco: aliases)
// outputs: dvdot[n,H], do[n,3H] = [sum_sp dveco*vec3 | dxo*vdot | dxo], dvp[3n,3H] = [0 | 0 | dveco*o1]
// dvdot == nullptr (round 5): vdot = sum_sp vec1 * vec2 is a function of the same vp, and its gradient g = dx_out * o2 is formed right here — the vec1 / vec2
// columns of dvp take g * vec2 / g * vec1 instead of zeros, so that no separate vecdot backward writes a second [3n,3H] tensor (zero in the vec3 columns) for
// autograd to add to this one (zero in the other two): per layer one launch and a 240 MB element-wise add less.
__global__ void k_node_update_bwd(const float *__restrict__ dxo, const float *__restrict__ dveco, const float *__restrict__ vdot, const float *__restrict__ o,
                                  const float *__restrict__ vp, int n, int H, float *__restrict__ dvdot, float *__restrict__ dout_o, float *__restrict__ dvp) {
    const long long tot = (long long)n * H, stride = (long long)gridDim.x * blockDim.x;
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < tot; t += stride) {
        const int a = (int)(t / H), c = (int)(t - (long long)a * H);
        const float gx = dxo[t];
        const float o1 = o[(size_t)a * 3 * H + c], o2 = o[(size_t)a * 3 * H + H + c];
        const float gd = gx * o2;
        if (dvdot) dvdot[t] = gd;
        float g1 = 0.f;
        for (int sp = 0; sp < 3; ++sp) {
            const size_t vi = ((size_t)a * 3 + sp) * H + c, pi = ((size_t)a * 3 + sp) * 3 * H;
            const float gv = dveco[vi];
            g1 += gv * vp[pi + 2 * H + c];
            dvp[pi + c] = dvdot ? 0.f : gd * vp[pi + H + c];
            dvp[pi + H + c] = dvdot ? 0.f : gd * vp[pi + c];
            dvp[pi + 2 * H + c] = gv * o1;
        }
        dout_o[(size_t)a * 3 * H + c] = g1;
        dout_o[(size_t)a * 3 * H + H + c] = gx * vdot[t];
        dout_o[(size_t)a * 3 * H + 2 * H + c] = gx;
    }
}

// ---------------------------------------------------------------------------------------------- edge update backward
// forward: fo = f + t * sum_sp w1*w2, w1 = a - (a.d)d, w2 = b - (b.d)d  (a = wt[tgt], b = ws[src]; the sign of d cancels)
// target pass: dwt[i] = sum_{e in row(i)} P_d (g * w2), and dt[e] = dfo * (w1.w2)   with g = dfo * t, P_d u = u - (u.d)d
template <int CPL, bool HALF = false>
__global__ void __launch_bounds__(256) k_edge_update_bwd_t(const float *__restrict__ wt, const float *__restrict__ ws, const float *__restrict__ t,
                                                           const float *__restrict__ dvec3, const float *__restrict__ dfo, const int *__restrict__ rowptr,
                                                           const int *__restrict__ col, int n, int H, int pre, float *__restrict__ dwt, float *__restrict__ dt) {
    const int lane = threadIdx.x & 63;
    const int hf = HALF ? lane >> 5 : 0, ll = HALF ? (lane & 31) : lane;      // HALF: a half-wavefront per edge (H = 32 CPL), two edges per step
    constexpr int ES = HALF ? 2 : 1;
    const int wave = __builtin_amdgcn_readfirstlane((int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6)), nw = (gridDim.x * blockDim.x) >> 6;
    for (int i = wave; i < n; i += nw)
        for (int cp = 0; cp < H; cp += (HALF ? 32 : 64) * CPL) {
            const int c0 = cp + ll * CPL; const bool on = c0 < H; const int cl = on ? c0 : 0;
            const float *a = wt + (size_t)i * 3 * H;
            float a0[CPL], a1[CPL], a2[CPL], s0[CPL], s1[CPL], s2[CPL];
#pragma unroll
            for (int u = 0; u < CPL; ++u) { a0[u] = a[cl + u]; a1[u] = a[H + cl + u]; a2[u] = a[2 * H + cl + u]; s0[u] = 0.f; s1[u] = 0.f; s2[u] = 0.f; }
            const int e0 = rowptr[i], e1 = rowptr[i + 1];
            for (int base = e0; base < e1; base += 64) {
                const int cnt = min(64, e1 - base);
                const int my_j = lane < cnt ? col[base + lane] : 0;
                float my_d[3];
#pragma unroll
                for (int sp = 0; sp < 3; ++sp) my_d[sp] = lane < cnt ? dvec3[(size_t)(base + lane) * 3 + sp] : 0.f;
                for (int tq = 0; tq < cnt; tq += ES * VB_EB) {
                    float bb[VB_EB][3][CPL], gf[VB_EB][CPL], tr[VB_EB][CPL];
#pragma unroll
                    for (int b = 0; b < VB_EB; ++b) {
                        const int tt = min(tq + ES * b + hf, cnt - 1);
                        const size_t j = (size_t)__shfl(my_j, tt, 64), e = (size_t)(base + tt);
#pragma unroll
                        for (int sp = 0; sp < 3; ++sp) vld<CPL>(ws + (j * 3 + sp) * H + cl, bb[b][sp]);
                        vld<CPL>(dfo + e * H + cl, gf[b]); vld<CPL>(t + e * H + cl, tr[b]);
                    }
#pragma unroll
                    for (int b = 0; b < VB_EB; ++b) {
                        const float d0 = __shfl(my_d[0], min(tq + ES * b + hf, cnt - 1), 64), d1 = __shfl(my_d[1], min(tq + ES * b + hf, cnt - 1), 64), d2 = __shfl(my_d[2], min(tq + ES * b + hf, cnt - 1), 64);      // before the halves diverge
                        if (tq + ES * b + hf >= cnt) continue;
                        const size_t e = (size_t)(base + tq + ES * b + hf);
                        float dtv[CPL];
#pragma unroll
                        for (int u = 0; u < CPL; ++u) {
                            const float b0 = bb[b][0][u], b1 = bb[b][1][u], b2 = bb[b][2][u];
                            const float pa = a0[u] * d0 + a1[u] * d1 + a2[u] * d2, pb = b0 * d0 + b1 * d1 + b2 * d2;
                            const float w10 = a0[u] - pa * d0, w11 = a1[u] - pa * d1, w12 = a2[u] - pa * d2;
                            const float w20 = b0 - pb * d0, w21 = b1 - pb * d1, w22 = b2 - pb * d2;
                            dtv[u] = gf[b][u] * (w10 * w20 + w11 * w21 + w12 * w22) * (pre ? dsilu_f(tr[b][u]) : 1.0f);
                            const float g = gf[b][u] * (pre ? silu_f(tr[b][u]) : tr[b][u]);
                            const float u0 = g * w20, u1 = g * w21, u2 = g * w22;
                            const float pu = u0 * d0 + u1 * d1 + u2 * d2;
                            s0[u] += u0 - pu * d0; s1[u] += u1 - pu * d1; s2[u] += u2 - pu * d2;
                        }
                        if (on) vst<CPL>(dt + e * H + c0, dtv);
                    }
                }
            }
            float *o = dwt + (size_t)i * 3 * H;
            vfold<CPL, HALF>(s0); vfold<CPL, HALF>(s1); vfold<CPL, HALF>(s2);
            if (on && hf == 0) { vst<CPL>(o + c0, s0); vst<CPL>(o + H + c0, s1); vst<CPL>(o + 2 * H + c0, s2); }
        }
}
// source pass: dws[j] = sum_{e in srclist(j)} P_d (g * w1)
template <int CPL, bool HALF = false>
__global__ void __launch_bounds__(256) k_edge_update_bwd_s(const float *__restrict__ wt, const float *__restrict__ t, const float *__restrict__ dvec3,
                                                           const float *__restrict__ dfo, const int *__restrict__ t_rowptr, const int *__restrict__ t_eid,
                                                           const int *__restrict__ tgt, int n, int H, int pre, float *__restrict__ dws) {
    const int lane = threadIdx.x & 63;
    const int hf = HALF ? lane >> 5 : 0, ll = HALF ? (lane & 31) : lane;      // HALF: a half-wavefront per edge (H = 32 CPL), two edges per step
    constexpr int ES = HALF ? 2 : 1;
    const int wave = __builtin_amdgcn_readfirstlane((int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6)), nw = (gridDim.x * blockDim.x) >> 6;
    for (int j = wave; j < n; j += nw)
        for (int cp = 0; cp < H; cp += (HALF ? 32 : 64) * CPL) {
            const int c0 = cp + ll * CPL; const bool on = c0 < H; const int cl = on ? c0 : 0;
            float s0[CPL], s1[CPL], s2[CPL];
#pragma unroll
            for (int u = 0; u < CPL; ++u) { s0[u] = 0.f; s1[u] = 0.f; s2[u] = 0.f; }
            const int q0 = t_rowptr[j], q1 = t_rowptr[j + 1];
            for (int base = q0; base < q1; base += 64) {
                const int cnt = min(64, q1 - base);
                const int my_e = lane < cnt ? t_eid[base + lane] : 0;
                const int my_i = lane < cnt ? tgt[my_e] : 0;
                float my_d[3];
#pragma unroll
                for (int sp = 0; sp < 3; ++sp) my_d[sp] = lane < cnt ? dvec3[(size_t)my_e * 3 + sp] : 0.f;
                for (int tq = 0; tq < cnt; tq += ES * VB_EB) {
                    float aa[VB_EB][3][CPL], gf[VB_EB][CPL], tr[VB_EB][CPL];
#pragma unroll
                    for (int b = 0; b < VB_EB; ++b) {
                        const int tt = min(tq + ES * b + hf, cnt - 1);
                        const size_t e = (size_t)__shfl(my_e, tt, 64), i = (size_t)__shfl(my_i, tt, 64);
#pragma unroll
                        for (int sp = 0; sp < 3; ++sp) vld<CPL>(wt + (i * 3 + sp) * H + cl, aa[b][sp]);
                        vld<CPL>(dfo + e * H + cl, gf[b]); vld<CPL>(t + e * H + cl, tr[b]);
                    }
#pragma unroll
                    for (int b = 0; b < VB_EB; ++b) {
                        const float d0 = __shfl(my_d[0], min(tq + ES * b + hf, cnt - 1), 64), d1 = __shfl(my_d[1], min(tq + ES * b + hf, cnt - 1), 64), d2 = __shfl(my_d[2], min(tq + ES * b + hf, cnt - 1), 64);      // before the halves diverge
                        if (tq + ES * b + hf >= cnt) continue;
#pragma unroll
                        for (int u = 0; u < CPL; ++u) {
                            const float a0 = aa[b][0][u], a1 = aa[b][1][u], a2 = aa[b][2][u];
                            const float pa = a0 * d0 + a1 * d1 + a2 * d2;
                            const float w10 = a0 - pa * d0, w11 = a1 - pa * d1, w12 = a2 - pa * d2;
                            const float g = gf[b][u] * (pre ? silu_f(tr[b][u]) : tr[b][u]);
                            const float u0 = g * w10, u1 = g * w11, u2 = g * w12;
                            const float pu = u0 * d0 + u1 * d1 + u2 * d2;
                            s0[u] += u0 - pu * d0; s1[u] += u1 - pu * d1; s2[u] += u2 - pu * d2;
                        }
                    }
                }
            }
            float *o = dws + (size_t)j * 3 * H;
            vfold<CPL, HALF>(s0); vfold<CPL, HALF>(s1); vfold<CPL, HALF>(s2);
            if (on && hf == 0) { vst<CPL>(o + c0, s0); vst<CPL>(o + H + c0, s1); vst<CPL>(o + 2 * H + c0, s2); }
        }
}

// ---------------------------------------------------------------------------------------------- head pieces backward
__global__ void k_spatial_norm_bwd(const float *__restrict__ v, const float *__restrict__ dout, int n, int H, float *__restrict__ dv) {
    const long long tot = (long long)n * H, stride = (long long)gridDim.x * blockDim.x;
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < tot; t += stride) {
        const int a = (int)(t / H), c = (int)(t - (long long)a * H);
        const size_t i0 = ((size_t)a * 3) * H + c;
        const float v0 = v[i0], v1 = v[i0 + H], v2 = v[i0 + 2 * H];
        const float nrm = sqrtf(v0 * v0 + v1 * v1 + v2 * v2);
        const float s = nrm > 0.f ? dout[t] / nrm : 0.f;           // torch.norm backward: zero sub-gradient at the origin
        dv[i0] = s * v0; dv[i0 + H] = s * v1; dv[i0 + 2 * H] = s * v2;
    }
}
// du[n,2O] = [dxo * act'(xv) | sum_sp dvo*v2],  dv2[sp] = dvo[sp] * gate
__global__ void k_gate_bwd(const float *__restrict__ u, const float *__restrict__ v2, const float *__restrict__ dxo, const float *__restrict__ dvo, int n, int O,
                           int act, float *__restrict__ du, float *__restrict__ dv2) {
    const long long tot = (long long)n * O, stride = (long long)gridDim.x * blockDim.x;
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < tot; t += stride) {
        const int a = (int)(t / O), c = (int)(t - (long long)a * O);
        const float xv = u[(size_t)a * 2 * O + c], g = u[(size_t)a * 2 * O + O + c];
        du[(size_t)a * 2 * O + c] = dxo[t] * (act ? dsilu_f(xv) : 1.0f);
        float dg = 0.f;
        for (int sp = 0; sp < 3; ++sp) {
            const size_t i = ((size_t)a * 3 + sp) * O + c;
            dg += dvo[i] * v2[i];
            dv2[i] = dvo[i] * g;
        }
        du[(size_t)a * 2 * O + O + c] = dg;
    }
}
__global__ void k_scale_scalar(const float *__restrict__ x, const float *__restrict__ sdev, long long n, float *__restrict__ out) {
    const float s = *sdev;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += stride) out[t] = x[t] * s;
}

}  // namespace

#define VB_CHECK(cond) if (!(cond)) return CONAN_E_BADARG
extern "C" {

int conan_silu_fwd(const float *x, int rows, int width, const int *m_dev, float *y, void *stream) {
    VB_CHECK(x && y && rows >= 0 && width > 0);
    k_silu_fwd<<<nblk((long long)rows * width), 256, 0, as_stream(stream)>>>(x, rows, width, m_dev, y);
    CONAN_LAUNCH_CHECK(); return CONAN_OK;
}
int conan_silu_bwd(const float *x, const float *dy, int rows, int width, const int *m_dev, float *dx, void *stream) {
    VB_CHECK(x && dy && dx && rows >= 0 && width > 0);
    k_silu_bwd<<<nblk((long long)rows * width), 256, 0, as_stream(stream)>>>(x, dy, rows, width, m_dev, dx);
    CONAN_LAUNCH_CHECK(); return CONAN_OK;
}
int conan_split2(const float *in, int Ha, int Hb, long long rows, float *a, float *b, void *stream) {
    VB_CHECK(in && a && b && Ha > 0 && Hb > 0 && rows >= 0);
    k_split2<<<nblk(rows * (Ha + Hb)), 256, 0, as_stream(stream)>>>(in, Ha, Hb, rows, a, b);
    CONAN_LAUNCH_CHECK(); return CONAN_OK;
}
int conan_rowsum(const float *x, int rows, int width, float *out, void *stream) {
    VB_CHECK(x && out && rows >= 0 && width > 0);
    k_rowsum<<<nblk(rows), 256, 0, as_stream(stream)>>>(x, rows, width, out);
    CONAN_LAUNCH_CHECK(); return CONAN_OK;
}
int conan_scale_scalar(const float *x, const float *scale_dev, long long count, float *out, void *stream) {
    VB_CHECK(x && scale_dev && out && count >= 0);
    k_scale_scalar<<<nblk(count), 256, 0, as_stream(stream)>>>(x, scale_dev, count, out);
    CONAN_LAUNCH_CHECK(); return CONAN_OK;
}
int conan_visnet_edge_embed_bwd(const float *x, const float *p, const float *df, const int *rowptr, const int *col, const int *tgt,
                                const int *t_rowptr, const int *t_eid, const int *num_edges_dev, int max_edges, int n, int H, float *dp,
                                float *dx, void *stream) {
    VB_CHECK(x && p && df && rowptr && col && tgt && t_rowptr && t_eid && num_edges_dev && dp && dx && H > 0);
    hipStream_t s = as_stream(stream);
    if (H % 128 == 0) {
        if (H == 128 && V_HALF) k_edge_embed_bwd_p<4, true><<<nblk((long long)max_edges * (64 / VB_RUN)), 256, 0, s>>>(x, df, col, tgt, num_edges_dev, max_edges, H, dp);
        else k_edge_embed_bwd_p<2><<<nblk((long long)max_edges * (64 / VB_RUN)), 256, 0, s>>>(x, df, col, tgt, num_edges_dev, max_edges, H, dp);
        if (H == 128 && V_HALF) k_edge_embed_bwd_x<4, true><<<nblk((long long)n * 64), 256, 0, s>>>(p, df, rowptr, t_rowptr, t_eid, n, H, dx);
        else k_edge_embed_bwd_x<2><<<nblk((long long)n * 64), 256, 0, s>>>(p, df, rowptr, t_rowptr, t_eid, n, H, dx);
    } else {
        k_edge_embed_bwd_p<1><<<nblk((long long)max_edges * (64 / VB_RUN)), 256, 0, s>>>(x, df, col, tgt, num_edges_dev, max_edges, H, dp);
        k_edge_embed_bwd_x<1><<<nblk((long long)n * 64), 256, 0, s>>>(p, df, rowptr, t_rowptr, t_eid, n, H, dx);
    }
    CONAN_LAUNCH_CHECK(); return CONAN_OK;
}
long long conan_layernorm_bwd_ws(int rows, int H) { return 2LL * rows + 2LL * ((rows + LN_CHUNK - 1) / LN_CHUNK) * H; }
static int layernorm_bwd_launch(const float *x, const float *gamma, const float *dy, const float *dres, int rows, int H, float eps, float *dx, float *dgamma,
                                float *dbeta, float *ws, void *stream);
int conan_layernorm_bwd(const float *x, const float *gamma, const float *dy, int rows, int H, float eps, float *dx, float *dgamma,
                        float *dbeta, float *ws, void *stream) {
    VB_CHECK(x && gamma && dy && dx && dgamma && dbeta && ws && rows >= 0 && H > 0);
    return layernorm_bwd_launch(x, gamma, dy, nullptr, rows, H, eps, dx, dgamma, dbeta, ws, stream);
}
int conan_layernorm_bwd_res(const float *x, const float *gamma, const float *dy, const float *dres, int rows, int H, float eps, float *dx, float *dgamma,
                            float *dbeta, float *ws, void *stream) {
    VB_CHECK(x && gamma && dy && dres && dx && dgamma && dbeta && ws && rows >= 0 && H > 0);
    return layernorm_bwd_launch(x, gamma, dy, dres, rows, H, eps, dx, dgamma, dbeta, ws, stream);
}
static int layernorm_bwd_launch(const float *x, const float *gamma, const float *dy, const float *dres, int rows, int H, float eps, float *dx, float *dgamma,
                                float *dbeta, float *ws, void *stream) {
    hipStream_t s = as_stream(stream);
    float *stats = ws, *slabs = ws + 2 * (size_t)rows;
    const int chunks = (rows + LN_CHUNK - 1) / LN_CHUNK;
    if (rows > 0) {
        k_layernorm_bwd_x<<<nblk((long long)rows * 64), 256, 0, s>>>(x, gamma, dy, dres, rows, H, eps, dx, stats);
        k_layernorm_bwd_p<<<dim3(chunks, (H + 127) / 128), 128 * LN_RL, 0, s>>>(x, dy, stats, rows, H, slabs);
    }
    k_layernorm_bwd_reduce<<<(H + 255) / 256, 256, 0, s>>>(slabs, chunks, H, dgamma, dbeta);
    CONAN_LAUNCH_CHECK(); return CONAN_OK;
}
int conan_visnet_vecdot_bwd(const float *vp, const float *dout, int n, int H, float *dvp, void *stream) {
    VB_CHECK(vp && dout && dvp && n >= 0 && H > 0);
    k_vecdot_bwd<<<nblk((long long)n * 3 * H), 256, 0, as_stream(stream)>>>(vp, dout, n, H, dvp);
    CONAN_LAUNCH_CHECK(); return CONAN_OK;
}
int conan_visnet_attn_message_bwd(const float *q, const float *k, const float *v, const float *dk, const float *dv, const float *dvmsg,
                                  const float *dxagg, const int *rowptr, const int *col, const int *tgt, const int *t_rowptr,
                                  const int *t_eid, const float *dist, float cutoff, int n, int H, int num_heads, int pre_act, float *dq,
                                  float *dkn, float *dvn, float *ddk, float *ddv, void *stream) {
    VB_CHECK(q && k && v && dk && dv && dvmsg && dxagg && rowptr && col && tgt && t_rowptr && t_eid && dist && dq && dkn && dvn && ddk && ddv);
    VB_CHECK(n >= 0 && H > 0 && num_heads > 0 && H % num_heads == 0);
    const int hd = H / num_heads, cpl = H > 64 ? (H + 63) / 64 : 1;
    const bool blocks128 = H % 128 == 0 && V_HALF && hd % 4 == 0 && (((hd / 4) & (hd / 4 - 1)) == 0) && 128 % hd == 0;      // as conan_visnet_attn_message
    if (!blocks128 && (H > 128 || (H > 64 && H != 128) || hd % cpl != 0)) return CONAN_E_UNSUPPORTED;
    const int lph = blocks128 ? hd / 4 : hd / cpl;
    if (lph & (lph - 1)) return CONAN_E_UNSUPPORTED;
    if (n == 0) return CONAN_OK;
    hipStream_t s = as_stream(stream);
    const int g = nblk((long long)n * 64);
    if (blocks128) {
        const dim3 gb(g, H / 128);
        k_attn_bwd_target<4, true><<<gb, 256, 0, s>>>(q, k, v, dk, dv, dvmsg, dxagg, rowptr, col, dist, cutoff, n, H, hd / 4, pre_act, dq, ddk, ddv);
        k_attn_bwd_source<4, true><<<gb, 256, 0, s>>>(q, k, v, dk, dv, dvmsg, dxagg, t_rowptr, t_eid, tgt, dist, cutoff, n, H, hd / 4, pre_act, dkn, dvn);
    } else if (cpl == 2) {
        k_attn_bwd_target<2><<<g, 256, 0, s>>>(q, k, v, dk, dv, dvmsg, dxagg, rowptr, col, dist, cutoff, n, H, lph, pre_act, dq, ddk, ddv);
        k_attn_bwd_source<2><<<g, 256, 0, s>>>(q, k, v, dk, dv, dvmsg, dxagg, t_rowptr, t_eid, tgt, dist, cutoff, n, H, lph, pre_act, dkn, dvn);
    } else {
        k_attn_bwd_target<1><<<g, 256, 0, s>>>(q, k, v, dk, dv, dvmsg, dxagg, rowptr, col, dist, cutoff, n, H, lph, pre_act, dq, ddk, ddv);
        k_attn_bwd_source<1><<<g, 256, 0, s>>>(q, k, v, dk, dv, dvmsg, dxagg, t_rowptr, t_eid, tgt, dist, cutoff, n, H, lph, pre_act, dkn, dvn);
    }
    CONAN_LAUNCH_CHECK(); return CONAN_OK;
}
int conan_visnet_vec_aggregate_bwd(const float *vec, const float *s, const float *dvec3, const float *dvagg, const int *col, const int *tgt,
                                   const int *t_rowptr, const int *t_eid, const int *num_edges_dev, int max_edges, int n, int H, int pre_act,
                                   float *ds, float *dvec, void *stream) {
    VB_CHECK(vec && s && dvec3 && dvagg && col && tgt && t_rowptr && t_eid && num_edges_dev && ds && dvec && H > 0);
    hipStream_t st = as_stream(stream);
    if (H == 128 && V_HALF) k_vec_aggregate_bwd_s<4, true><<<nblk((long long)max_edges * (64 / VB_RUN)), 256, 0, st>>>(vec, dvagg, dvec3, col, tgt, num_edges_dev, max_edges, H, pre_act ? s : nullptr, ds);
    else if (H % 128 == 0) k_vec_aggregate_bwd_s<2><<<nblk((long long)max_edges * (64 / VB_RUN)), 256, 0, st>>>(vec, dvagg, dvec3, col, tgt, num_edges_dev, max_edges, H, pre_act ? s : nullptr, ds);
    else k_vec_aggregate_bwd_s<1><<<nblk((long long)max_edges * (64 / VB_RUN)), 256, 0, st>>>(vec, dvagg, dvec3, col, tgt, num_edges_dev, max_edges, H, pre_act ? s : nullptr, ds);
    if (n > 0) {
        if (H == 128 && V_HALF) k_vec_aggregate_bwd_v<4, true><<<nblk((long long)n * 64), 256, 0, st>>>(s, dvagg, t_rowptr, t_eid, tgt, n, H, pre_act, dvec);
        else if (H % 128 == 0) k_vec_aggregate_bwd_v<2><<<nblk((long long)n * 64), 256, 0, st>>>(s, dvagg, t_rowptr, t_eid, tgt, n, H, pre_act, dvec);
        else k_vec_aggregate_bwd_v<1><<<nblk((long long)n * 64), 256, 0, st>>>(s, dvagg, t_rowptr, t_eid, tgt, n, H, pre_act, dvec);
    }
    CONAN_LAUNCH_CHECK(); return CONAN_OK;
}
int conan_visnet_node_update_bwd(const float *dxo, const float *dveco, const float *vdot, const float *o, const float *vp, int n, int H,
                                 float *dvdot, float *dout_o, float *dvp, void *stream) {
    VB_CHECK(dxo && dveco && vdot && o && vp && dout_o && dvp && n >= 0 && H > 0);
    k_node_update_bwd<<<nblk((long long)n * H), 256, 0, as_stream(stream)>>>(dxo, dveco, vdot, o, vp, n, H, dvdot, dout_o, dvp);
    CONAN_LAUNCH_CHECK(); return CONAN_OK;
}
int conan_visnet_edge_update_bwd(const float *wt, const float *ws, const float *t, const float *dvec3, const float *dfo, const int *rowptr,
                                 const int *col, const int *tgt, const int *t_rowptr, const int *t_eid, int n, int H, int pre_act, float *dwt,
                                 float *dws, float *dt, void *stream) {
    VB_CHECK(wt && ws && t && dvec3 && dfo && rowptr && col && tgt && t_rowptr && t_eid && dwt && dws && dt && n >= 0 && H > 0);
    if (n == 0) return CONAN_OK;
    hipStream_t s = as_stream(stream);
    if (H % 128 == 0) {
        if (H == 128 && V_HALF) k_edge_update_bwd_t<4, true><<<nblk((long long)n * 64), 256, 0, s>>>(wt, ws, t, dvec3, dfo, rowptr, col, n, H, pre_act, dwt, dt);
        else k_edge_update_bwd_t<2><<<nblk((long long)n * 64), 256, 0, s>>>(wt, ws, t, dvec3, dfo, rowptr, col, n, H, pre_act, dwt, dt);
        if (H == 128 && V_HALF) k_edge_update_bwd_s<4, true><<<nblk((long long)n * 64), 256, 0, s>>>(wt, t, dvec3, dfo, t_rowptr, t_eid, tgt, n, H, pre_act, dws);
        else k_edge_update_bwd_s<2><<<nblk((long long)n * 64), 256, 0, s>>>(wt, t, dvec3, dfo, t_rowptr, t_eid, tgt, n, H, pre_act, dws);
    } else {
        k_edge_update_bwd_t<1><<<nblk((long long)n * 64), 256, 0, s>>>(wt, ws, t, dvec3, dfo, rowptr, col, n, H, pre_act, dwt, dt);
        k_edge_update_bwd_s<1><<<nblk((long long)n * 64), 256, 0, s>>>(wt, t, dvec3, dfo, t_rowptr, t_eid, tgt, n, H, pre_act, dws);
    }
    CONAN_LAUNCH_CHECK(); return CONAN_OK;
}
int conan_visnet_spatial_norm_bwd(const float *v, const float *dout, int n, int H, float *dv, void *stream) {
    VB_CHECK(v && dout && dv && n >= 0 && H > 0);
    k_spatial_norm_bwd<<<nblk((long long)n * H), 256, 0, as_stream(stream)>>>(v, dout, n, H, dv);
    CONAN_LAUNCH_CHECK(); return CONAN_OK;
}
int conan_visnet_gate_bwd(const float *u, const float *v2, const float *dxo, const float *dvo, int n, int out_channels, int scalar_activation,
                          float *du, float *dv2, void *stream) {
    VB_CHECK(u && v2 && dxo && dvo && du && dv2 && n >= 0 && out_channels > 0);
    k_gate_bwd<<<nblk((long long)n * out_channels), 256, 0, as_stream(stream)>>>(u, v2, dxo, dvo, n, out_channels, scalar_activation, du, dv2);
    CONAN_LAUNCH_CHECK(); return CONAN_OK;
}

}  // extern "C"
