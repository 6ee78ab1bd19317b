// ViSNet (vector-scalar interactive message passing) forward kernels for gfx950: everything of
// conan_fgw/src/model/graph_embeddings/torch_geometric_visnet.py that is not a plain Linear layer
// (those run on conan_linear_fwd).  Layouts: x [n,H]; vec [n,3,H]; edges = CSR by target INCLUDING self loops
// (Distance(add_self_loops=True), :331-347); f_ij [E,H]; d_ij [E,3] unit vectors (0 for loops); r_ij [E] (0 for loops).
// All kernels are HBM/L2-streaming gathers; node-side projections were hoisted out of the edge loops
// (w_trg_proj / w_src_proj commute with the gather, :657-658), so the only edge-level GEMMs left are dk/dv/s/f_proj.
#include "common.h"

namespace {

// v * sigmoid(v) with the hardware reciprocal (1 ulp) instead of an IEEE division: the message kernels apply it per loaded element now
__device__ __forceinline__ float silu_f(float v) { return v * __builtin_amdgcn_rcpf(1.0f + __expf(-v)); }
__device__ __forceinline__ float cos_cutoff(float d, float cutoff) {      // CosineCutoff, :33-46
    return d < cutoff ? 0.5f * (cosf(__fdiv_rn(d * 3.14159265358979323846f, cutoff)) + 1.0f) : 0.0f;
}

// d_ij = (pos[src]-pos[tgt]) / |.| for src != tgt, 0 for self loops   (:340-347, :864-866; Sphere(lmax=1) is the identity)
__global__ void k_edge_unit(const float *__restrict__ pos, const int *__restrict__ col, const int *__restrict__ tgt,
                            const int *__restrict__ ne_dev, int max_edges, float *__restrict__ dvec) {
    const int E = min(*ne_dev, max_edges);
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < E; e += gridDim.x * blockDim.x) {
        const int j = col[e], i = tgt[e];
        float dx = pos[j * 3] - pos[i * 3], dy = pos[j * 3 + 1] - pos[i * 3 + 1], dz = pos[j * 3 + 2] - pos[i * 3 + 2];
        if (i != j) { const float nrm = sqrtf(dx * dx + dy * dy + dz * dz); dx /= nrm; dy /= nrm; dz /= nrm; }
        else { dx = dy = dz = 0.f; }
        dvec[e * 3] = dx; dvec[e * 3 + 1] = dy; dvec[e * 3 + 2] = dz;
    }
}

// ExpNormalSmearing (:100-111): cutoff(d) * exp(-beta_k * (exp(-alpha d) - mu_k)^2)
__global__ void k_expnormal(const float *__restrict__ dist, const int *__restrict__ ne_dev, int max_edges, const float *__restrict__ means,
                            const float *__restrict__ betas, int R, float alpha, float cutoff, float *__restrict__ out) {
    const int E = min(*ne_dev, max_edges);
    const long long n = (long long)E * R, stride = (long long)gridDim.x * blockDim.x;
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += stride) {
        const int e = (int)(t / R), k = (int)(t - (long long)e * R);
        const float d = dist[e];
        const float u = expf(alpha * (-d)) - means[k];
        out[t] = cos_cutoff(d, cutoff) * expf(-betas[k] * (u * u));
    }
}

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
constexpr bool V_HALF = true;     // H = 128: a half-wavefront per edge (32 lanes x float4 = one 512-byte row), two edges per instruction
constexpr int VN_EB = CONAN_V_EB;      // edges in flight per wavefront
#ifndef CONAN_VB_RUN
#define CONAN_VB_RUN 16
#endif
constexpr int VN_RUN = CONAN_VB_RUN;    // edges per wavefront in the kernels that walk runs of consecutive edges (64 left too few wavefronts in flight)

// The three element-wise edge kernels below walk runs of VN_RUN consecutive edges per wavefront (round 3; were one thread per element with a
// 64-bit division, the index loads and — in k_ne_scale — a full-precision cosine per ELEMENT): per-edge quantities are formed once, one edge
// per lane, and handed out; lane <-> CPL channels; VN_EB edges in flight.
// W[e,:] *= C(r_e) * [src != tgt]      (NeighborEmbedding, :408-415: loops removed, cosine cutoff)
template <int CPL, bool HALF = false>
__global__ void __launch_bounds__(256) k_ne_scale(const float *Win, float *W, const float *__restrict__ dist, const int *__restrict__ col,
                                                  const int *__restrict__ tgt, const int *__restrict__ ne_dev, int max_edges, int H, float cutoff) {
    const int E = min(*ne_dev, max_edges);
    const int lane = threadIdx.x & 63;
    const int hf = HALF ? lane >> 5 : 0, ll = HALF ? (lane & 31) : lane;      // HALF: a half-wavefront per edge (H = 32 CPL), two edges per step
    constexpr int ES = HALF ? 2 : 1;
    const int wave = __builtin_amdgcn_readfirstlane((int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6)), nw = (gridDim.x * blockDim.x) >> 6;
    for (int base = wave * VN_RUN; base < E; base += nw * VN_RUN) {
        const int cnt = min(VN_RUN, E - base);
        const float my_s = (lane < cnt && col[base + lane] != tgt[base + lane]) ? cos_cutoff(dist[base + lane], cutoff) : 0.0f;
        for (int cp = 0; cp < H; cp += (HALF ? 32 : 64) * CPL) {
            const int c0 = cp + ll * CPL; const bool on = c0 < H; const int cl = on ? c0 : 0;
            for (int tq = 0; tq < cnt; tq += ES * VN_EB) {
                float w[VN_EB][CPL];
#pragma unroll
                for (int b = 0; b < VN_EB; ++b) vld<CPL>(Win + (size_t)(base + min(tq + ES * b + hf, cnt - 1)) * H + cl, w[b]);
#pragma unroll
                for (int b = 0; b < VN_EB; ++b) {
                    const float sc = __shfl(my_s, min(tq + ES * b + hf, cnt - 1), 64);      // before the halves diverge
                    if (tq + ES * b + hf >= cnt) continue;
#pragma unroll
                    for (int u = 0; u < CPL; ++u) w[b][u] *= sc;
                    if (on) vst<CPL>(W + (size_t)(base + tq + ES * b + hf) * H + c0, w[b]);
                }
            }
        }
    }
}

__global__ void k_concat2(const float *__restrict__ a, int Ha, const float *__restrict__ b, int Hb, long long rows, float *__restrict__ out) {
    const int Ho = Ha + Hb;
    const long long n = rows * Ho, stride = (long long)gridDim.x * blockDim.x;
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += stride) {
        const long long r = t / Ho; const int c = (int)(t - r * Ho);
        out[t] = c < Ha ? a[r * Ha + c] : b[r * Hb + (c - Ha)];
    }
}

// f_ij = (x_i + x_j) * p_e    (EdgeEmbedding, :463-465)
template <int CPL, bool HALF = false>
__global__ void __launch_bounds__(256) k_edge_embed(const float *__restrict__ x, const float *__restrict__ p, const int *__restrict__ col,
                                                    const int *__restrict__ tgt, const int *__restrict__ ne_dev, int max_edges, int H, float *__restrict__ f) {
    const int E = min(*ne_dev, max_edges);
    const int lane = threadIdx.x & 63;
    const int hf = HALF ? lane >> 5 : 0, ll = HALF ? (lane & 31) : lane;      // HALF: a half-wavefront per edge (H = 32 CPL), two edges per step
    constexpr int ES = HALF ? 2 : 1;
    const int wave = __builtin_amdgcn_readfirstlane((int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6)), nw = (gridDim.x * blockDim.x) >> 6;
    for (int base = wave * VN_RUN; base < E; base += nw * VN_RUN) {
        const int cnt = min(VN_RUN, E - base);
        const int my_j = lane < cnt ? col[base + lane] : 0, my_i = lane < cnt ? tgt[base + lane] : 0;
        for (int cp = 0; cp < H; cp += (HALF ? 32 : 64) * CPL) {
            const int c0 = cp + ll * CPL; const bool on = c0 < H; const int cl = on ? c0 : 0;
            for (int tq = 0; tq < cnt; tq += ES * VN_EB) {
                float xi[VN_EB][CPL], xj[VN_EB][CPL], pv[VN_EB][CPL];
#pragma unroll
                for (int b = 0; b < VN_EB; ++b) {
                    const int tt = min(tq + ES * b + hf, cnt - 1);
                    const size_t j = (size_t)__shfl(my_j, tt, 64), i = (size_t)__shfl(my_i, tt, 64);
                    vld<CPL>(x + i * H + cl, xi[b]); vld<CPL>(x + j * H + cl, xj[b]); vld<CPL>(p + (size_t)(base + tt) * H + cl, pv[b]);
                }
#pragma unroll
                for (int b = 0; b < VN_EB; ++b) {
                    if (tq + ES * b + hf >= cnt) continue;
                    float o[CPL];
#pragma unroll
                    for (int u = 0; u < CPL; ++u) o[u] = (xi[b][u] + xj[b][u]) * pv[b][u];
                    if (on) vst<CPL>(f + (size_t)(base + tq + ES * b + hf) * H + c0, o);
                }
            }
        }
    }
}

// torch.nn.LayerNorm over the last dim (biased variance), one wavefront per row
__global__ void __launch_bounds__(256) k_layernorm(const float *__restrict__ x, const float *__restrict__ gamma, const float *__restrict__ beta,
                                                   int rows, int H, float eps, float *__restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6)), nw = (gridDim.x * blockDim.x) >> 6;
    for (int r = wave; r < rows; r += nw) {
        float s = 0.f;
        for (int c = lane; c < H; c += 64) s += x[(size_t)r * H + c];
        const float mean = wave_sum(s) / (float)H;
        float v = 0.f;
        for (int c = lane; c < H; c += 64) { const float d = x[(size_t)r * H + c] - mean; v += d * d; }
        const float rstd = 1.0f / sqrtf(wave_sum(v) / (float)H + eps);
        for (int c = lane; c < H; c += 64) out[(size_t)r * H + c] = (x[(size_t)r * H + c] - mean) * rstd * gamma[c] + beta[c];
    }
}

// out = v * w[channel] (+ add): `add` (nullable) is a second gradient of the same tensor (the residual use of vec next to VecLayerNorm, :587,:660) that autograd
// would otherwise sum in a launch of its own.  H % 4 == 0: float4 per thread, the channel index from 32-bit arithmetic (the scalar form paid a 64-bit modulo per element).
__global__ void k_scale_channels(const float *__restrict__ v, const float *__restrict__ w, const float *__restrict__ add, long long rows, int H, float *__restrict__ out) {
    const long long n = rows * H, stride = (long long)gridDim.x * blockDim.x;
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += stride) out[t] = v[t] * w[t % H] + (add ? add[t] : 0.f);
}
__global__ void k_scale_channels4(const float4 *__restrict__ v, const float4 *__restrict__ w, const float4 *__restrict__ add, long long rows, int H4,
                                  float4 *__restrict__ out) {
    const int stride = (int)(gridDim.x * blockDim.x);                   // (a multiple of H4 is not needed: the channel is recomputed per element)
    for (long long r0 = 0; r0 < rows * H4; r0 += stride) {
        const long long t = r0 + (long long)blockIdx.x * blockDim.x + threadIdx.x;
        if (t >= rows * H4) break;
        const int c = (int)(t % H4);
        const float4 a = v[t], b = w[c];
        float4 o = make_float4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w);
        if (add) { const float4 d = add[t]; o.x += d.x; o.y += d.y; o.z += d.z; o.w += d.w; }
        out[t] = o;
    }
}

// vec_dot[a,c] = sum_sp vp[a,sp,c] * vp[a,sp,H+c]   with vp = vec_proj(vec) [n,3,3H]    (:606-607)
__global__ void k_vecdot(const float *__restrict__ vp, int n, int H, float *__restrict__ out) {
    const long long tot = (long long)n * H, stride = (long long)gridDim.x * blockDim.x;
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < tot; t += stride) {
        const int a = (int)(t / H), c = (int)(t - (long long)a * H);
        float s = 0.f;
        for (int sp = 0; sp < 3; ++sp) { const float *row = vp + ((size_t)a * 3 + sp) * 3 * H; s += row[c] * row[H + c]; }
        out[t] = s;
    }
}

// Attention message + scalar aggregation (ViS_MP.message first half + aggregate, :632-645, :671):
// attn_h = SiLU(sum_{c in head h} q_i k_j dk_e) * C(r_e);  vmsg_e = v_j * dv_e * attn_h;  xagg_i = sum_e vmsg_e.
// One wavefront per target; lane l owns CPL consecutive channels; a head spans LPH lanes (xor-shuffle reduction).
// Round 3: a row's col[] / dist[] entries are fetched once, one edge per lane, and handed out with cross-lane reads (a wave-uniform
// index becomes a scalar register: row base in SGPRs, lane offset in a VGPR), and the rows of EB edges are requested before the first
// of them is used — the loop was a chain of two dependent round trips per edge (index, then the k_j / v_j rows) and ran at 2.3 TB/s.
// The sums run in edge order as before (bitwise-equal results).
template <int CPL, bool HALF = false>
__global__ void __launch_bounds__(256) k_attn_msg(const float *__restrict__ q, const float *__restrict__ k, const float *__restrict__ v,
                                                  const float *__restrict__ dk, const float *__restrict__ dv, const int *__restrict__ rowptr,
                                                  const int *__restrict__ col, const float *__restrict__ dist, float cutoff, int n, int H,
                                                  int lph, int pre, float *__restrict__ vmsg, float *__restrict__ xagg) {
    const int lane = threadIdx.x & 63;
    const int hf = HALF ? lane >> 5 : 0, ll = HALF ? (lane & 31) : lane;      // HALF: a half-wavefront per edge (H = 32 CPL), two edges per step
    constexpr int ES = HALF ? 2 : 1;
    const int wave = __builtin_amdgcn_readfirstlane((int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6)), nw = (gridDim.x * blockDim.x) >> 6;
    const int c0 = (int)blockIdx.y * (HALF ? 32 : 64) * CPL + ll * CPL;      // blockIdx.y: block of 64 CPL (HALF: 32 CPL) channels — whole heads (H > 128)
    const bool on = c0 < H;
    const int cl = on ? c0 : 0;                                       // idle lanes (H < 64 CPL) read column 0 and store nothing
    for (int i = wave; i < n; i += nw) {
        float qi[CPL], acc[CPL];
#pragma unroll
        for (int u = 0; u < CPL; ++u) { qi[u] = q[(size_t)i * H + cl + u]; acc[u] = 0.f; }
        const int e0 = rowptr[i], e1 = rowptr[i + 1];
        for (int base = e0; base < e1; base += 64) {
            const int cnt = min(64, e1 - base);
            const int my_j = lane < cnt ? col[base + lane] : 0;
            const float my_c = lane < cnt ? cos_cutoff(dist[base + lane], cutoff) : 0.f;
            for (int t = 0; t < cnt; t += ES * VN_EB) {
                float kj[VN_EB][CPL], vj[VN_EB][CPL], dke[VN_EB][CPL], dve[VN_EB][CPL];
#pragma unroll
                for (int b = 0; b < VN_EB; ++b) {
                    const int tt = min(t + ES * b + hf, cnt - 1);                  // slots past the row repeat its last edge (not used)
                    const int j = __shfl(my_j, tt, 64);
                    const size_t e = (size_t)(base + tt);
                    vld<CPL>(k + (size_t)j * H + cl, kj[b]); vld<CPL>(v + (size_t)j * H + cl, vj[b]);
                    vld<CPL>(dk + e * H + cl, dke[b]); vld<CPL>(dv + e * H + cl, dve[b]);
                }
#pragma unroll
                for (int b = 0; b < VN_EB; ++b) {
                    const float cutb = __shfl(my_c, min(t + ES * b + hf, cnt - 1), 64);      // before the halves diverge
                    if (t + ES * b + hf >= cnt) continue;
                    float part = 0.f;
#pragma unroll
                    for (int u = 0; u < CPL; ++u) {
                        if (pre) { dke[b][u] = silu_f(dke[b][u]); dve[b][u] = silu_f(dve[b][u]); }      // dk / dv arrive as the projections' pre-activations
                        part += qi[u] * kj[b][u] * dke[b][u];
                    }
                    if (!on) part = 0.f;
                    for (int o = 1; o < lph; o <<= 1) part += __shfl_xor(part, o, 64);
                    const float attn = silu_f(part) * cutb;
                    float m[CPL];
#pragma unroll
                    for (int u = 0; u < CPL; ++u) { m[u] = vj[b][u] * dve[b][u] * attn; acc[u] += m[u]; }
                    if (on) vst<CPL>(vmsg + (size_t)(base + t + ES * b + hf) * H + c0, m);
                }
            }
        }
        vfold<CPL, HALF>(acc);
        if (on && hf == 0) vst<CPL>(xagg + (size_t)i * H + c0, acc);
    }
}

// Vector message + aggregation (:646-653, :672): vagg_i[sp] = sum_e vec_j[sp]*s1_e + s2_e*d_e[sp],  s = [s1|s2] in [E,2H]
template <int CPL, bool HALF = false>
__global__ void __launch_bounds__(256) k_vec_aggregate(const float *__restrict__ vec, const float *__restrict__ s, const float *__restrict__ dvec,
                                                       const int *__restrict__ rowptr, const int *__restrict__ col, int n, int H, int pre,
                                                       float *__restrict__ vagg) {
    // lane <-> CPL consecutive channels (H = 64 CPL: one pass over the row); indices and unit vectors handed out per lane, VN_EB edges in flight
    const int lane = threadIdx.x & 63;
    const int hf = HALF ? lane >> 5 : 0, ll = HALF ? (lane & 31) : lane;      // HALF: a half-wavefront per edge (H = 32 CPL), two edges per step
    constexpr int ES = HALF ? 2 : 1;
    const int wave = __builtin_amdgcn_readfirstlane((int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6)), nw = (gridDim.x * blockDim.x) >> 6;
    const int c0 = ll * CPL;
    for (int i = wave; i < n; i += nw) {
        float a0[CPL], a1[CPL], a2[CPL];
#pragma unroll
        for (int u = 0; u < CPL; ++u) { a0[u] = 0.f; a1[u] = 0.f; a2[u] = 0.f; }
        const int e0 = rowptr[i], e1 = rowptr[i + 1];
        for (int base = e0; base < e1; base += 64) {
            const int cnt = min(64, e1 - base);
            const int my_j = lane < cnt ? col[base + lane] : 0;
            float my_d[3];
#pragma unroll
            for (int sp = 0; sp < 3; ++sp) my_d[sp] = lane < cnt ? dvec[(size_t)(base + lane) * 3 + sp] : 0.f;
            for (int t = 0; t < cnt; t += ES * VN_EB) {
                float s1[VN_EB][CPL], s2[VN_EB][CPL], vj[VN_EB][3][CPL];
#pragma unroll
                for (int b = 0; b < VN_EB; ++b) {
                    const int tt = min(t + ES * b + hf, cnt - 1);
                    const int j = __shfl(my_j, tt, 64);
                    const size_t e = (size_t)(base + tt);
                    vld<CPL>(s + e * 2 * H + c0, s1[b]); vld<CPL>(s + e * 2 * H + H + c0, s2[b]);
#pragma unroll
                    for (int sp = 0; sp < 3; ++sp) vld<CPL>(vec + ((size_t)j * 3 + sp) * H + c0, vj[b][sp]);
                }
#pragma unroll
                for (int b = 0; b < VN_EB; ++b) {
                    const float d0 = __shfl(my_d[0], min(t + ES * b + hf, cnt - 1), 64), d1 = __shfl(my_d[1], min(t + ES * b + hf, cnt - 1), 64), d2 = __shfl(my_d[2], min(t + ES * b + hf, cnt - 1), 64);      // before the halves diverge
                    if (t + ES * b + hf >= cnt) continue;
#pragma unroll
                    for (int u = 0; u < CPL; ++u) {
                        float x1 = s1[b][u], x2 = s2[b][u];
                        if (pre) { x1 = silu_f(x1); x2 = silu_f(x2); }             // s arrives as s_proj's pre-activation
                        a0[u] += vj[b][0][u] * x1 + x2 * d0;
                        a1[u] += vj[b][1][u] * x1 + x2 * d1;
                        a2[u] += vj[b][2][u] * x1 + x2 * d2;
                    }
                }
            }
        }
        float *o = vagg + (size_t)i * 3 * H;
        vfold<CPL, HALF>(a0); vfold<CPL, HALF>(a1); vfold<CPL, HALF>(a2);
        if (hf == 0) { vst<CPL>(o + c0, a0); vst<CPL>(o + H + c0, a1); vst<CPL>(o + 2 * H + c0, a2); }
    }
}
// any H: lane <-> channel, strided passes (the round-1 form)
__global__ void __launch_bounds__(256) k_vec_aggregate_any(const float *__restrict__ vec, const float *__restrict__ s, const float *__restrict__ dvec,
                                                           const int *__restrict__ rowptr, const int *__restrict__ col, int n, int H, int pre,
                                                           float *__restrict__ vagg) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6)), nw = (gridDim.x * blockDim.x) >> 6;
    for (int i = wave; i < n; i += nw) {
        for (int c = lane; c < H; c += 64) {
            float a0 = 0.f, a1 = 0.f, a2 = 0.f;
            for (int e = rowptr[i]; e < rowptr[i + 1]; ++e) {
                const int j = col[e];
                float s1 = s[(size_t)e * 2 * H + c], s2 = s[(size_t)e * 2 * H + H + c];
                if (pre) { s1 = silu_f(s1); s2 = silu_f(s2); }
                const float *vj = vec + (size_t)j * 3 * H;
                a0 += vj[c] * s1 + s2 * dvec[e * 3];
                a1 += vj[H + c] * s1 + s2 * dvec[e * 3 + 1];
                a2 += vj[2 * H + c] * s1 + s2 * dvec[e * 3 + 2];
            }
            float *o = vagg + (size_t)i * 3 * H;
            o[c] = a0; o[H + c] = a1; o[2 * H + c] = a2;
        }
    }
}

// Node update (:623-625, :873-881): x' = x + vec_dot*o2 + o3 ;  vec'[sp] = vec[sp] + vec3[sp]*o1 + vagg[sp]
// o = o_proj(xagg) [n,3H] = [o1|o2|o3];  vec3 = vp[:, :, 2H:3H]
__global__ void k_node_update(const float *__restrict__ x, const float *__restrict__ vec, const float *__restrict__ vdot, const float *__restrict__ o,
                              const float *__restrict__ vp, const float *__restrict__ vagg, int n, int H, float *__restrict__ xo, float *__restrict__ veco) {
    const long long tot = (long long)n * H, stride = (long long)gridDim.x * blockDim.x;
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < tot; t += stride) {
        const int a = (int)(t / H), c = (int)(t - (long long)a * H);
        const float o1 = o[(size_t)a * 3 * H + c], o2 = o[(size_t)a * 3 * H + H + c], o3 = o[(size_t)a * 3 * H + 2 * H + c];
        xo[t] = x[t] + (vdot[t] * o2 + o3);
        for (int sp = 0; sp < 3; ++sp) {
            const size_t idx = ((size_t)a * 3 + sp) * H + c;
            veco[idx] = vec[idx] + (vp[((size_t)a * 3 + sp) * 3 * H + 2 * H + c] * o1 + vagg[idx]);
        }
    }
}

// Edge update (:655-661): w1 = rej(wt[tgt], d), w2 = rej(ws[src], -d), f' = f + SiLU(f_proj(f)) * sum_sp w1*w2
// wt = w_trg_proj(vec), ws = w_src_proj(vec) are node-level [n,3,H]; t = SiLU(f_proj(f_ij)) [E,H]
// One wavefront per run of VN_RUN consecutive edges (round 3; was one thread per element: a 64-bit division, two index loads and six gathered
// rows per ELEMENT, each waiting for its index): sources, targets and unit vectors of the run are fetched once, one edge per lane, and
// handed out; lane <-> CPL channels; VN_EB edges in flight (consecutive edges share their target: its rows are L1 hits).
template <int CPL, bool HALF = false>
__global__ void __launch_bounds__(256) k_edge_update(const float *__restrict__ wt, const float *__restrict__ ws, const float *__restrict__ t,
                                                     const float *__restrict__ dvec, const int *__restrict__ col, const int *__restrict__ tgt,
                                                     const int *__restrict__ ne_dev, int max_edges, int H, int pre, const float *__restrict__ f,
                                                     float *__restrict__ fo) {
    const int E = min(*ne_dev, max_edges);
    const int lane = threadIdx.x & 63;
    const int hf = HALF ? lane >> 5 : 0, ll = HALF ? (lane & 31) : lane;      // HALF: a half-wavefront per edge (H = 32 CPL), two edges per step
    constexpr int ES = HALF ? 2 : 1;
    const int wave = __builtin_amdgcn_readfirstlane((int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6)), nw = (gridDim.x * blockDim.x) >> 6;
    for (int base = wave * VN_RUN; base < E; base += nw * VN_RUN) {
        const int cnt = min(VN_RUN, E - base);
        const int my_j = lane < cnt ? col[base + lane] : 0, my_i = lane < cnt ? tgt[base + lane] : 0;
        float my_d[3];
#pragma unroll
        for (int sp = 0; sp < 3; ++sp) my_d[sp] = lane < cnt ? dvec[(size_t)(base + lane) * 3 + sp] : 0.f;
        for (int cp = 0; cp < H; cp += (HALF ? 32 : 64) * CPL) {
            const int c0 = cp + ll * CPL; const bool on = c0 < H; const int cl = on ? c0 : 0;
            for (int tq = 0; tq < cnt; tq += ES * VN_EB) {
                float aa[VN_EB][3][CPL], bb[VN_EB][3][CPL], tv[VN_EB][CPL], fv[VN_EB][CPL];
#pragma unroll
                for (int b = 0; b < VN_EB; ++b) {
                    const int tt = min(tq + ES * b + hf, cnt - 1);
                    const size_t j = (size_t)__shfl(my_j, tt, 64), i = (size_t)__shfl(my_i, tt, 64), e = (size_t)(base + tt);
#pragma unroll
                    for (int sp = 0; sp < 3; ++sp) { vld<CPL>(wt + (i * 3 + sp) * H + cl, aa[b][sp]); vld<CPL>(ws + (j * 3 + sp) * H + cl, bb[b][sp]); }
                    vld<CPL>(t + e * H + cl, tv[b]); vld<CPL>(f + e * H + cl, fv[b]);
                }
#pragma unroll
                for (int b = 0; b < VN_EB; ++b) {
                    const float d0 = __shfl(my_d[0], min(tq + ES * b + hf, cnt - 1), 64), d1 = __shfl(my_d[1], min(tq + ES * b + hf, cnt - 1), 64), d2 = __shfl(my_d[2], min(tq + ES * b + hf, cnt - 1), 64);      // before the halves diverge
                    if (tq + ES * b + hf >= cnt) continue;
                    const size_t e = (size_t)(base + tq + ES * b + hf);
                    float ov[CPL];
#pragma unroll
                    for (int u = 0; u < CPL; ++u) {
                        const float a0 = aa[b][0][u], a1 = aa[b][1][u], a2 = aa[b][2][u], b0 = bb[b][0][u], b1 = bb[b][1][u], b2 = bb[b][2][u];
                        const float pa = a0 * d0 + a1 * d1 + a2 * d2;                 // vec . d
                        const float pb = b0 * (-d0) + b1 * (-d1) + b2 * (-d2);        // vec . (-d)
                        const float w10 = a0 - pa * d0, w11 = a1 - pa * d1, w12 = a2 - pa * d2;
                        const float w20 = b0 - pb * (-d0), w21 = b1 - pb * (-d1), w22 = b2 - pb * (-d2);
                        const float tvv = pre ? silu_f(tv[b][u]) : tv[b][u];          // t arrives as f_proj's pre-activation
                        ov[u] = fv[b][u] + tvv * (w10 * w20 + w11 * w21 + w12 * w22);
                    }
                    if (on) vst<CPL>(fo + e * H + c0, ov);
                }
            }
        }
    }
}

// |v|_2 over the spatial axis: [n,3,H] -> [n,H]      (GatedEquivariantBlock, :943)
__global__ void k_spatial_norm(const float *__restrict__ v, int n, int H, float *__restrict__ out) {
    const long long tot = (long long)n * H, stride = (long long)gridDim.x * blockDim.x;
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < tot; t += stride) {
        const int a = (int)(t / H), c = (int)(t - (long long)a * H);
        const float v0 = v[((size_t)a * 3) * H + c], v1 = v[((size_t)a * 3 + 1) * H + c], v2 = v[((size_t)a * 3 + 2) * H + c];
        out[t] = sqrtf(v0 * v0 + v1 * v1 + v2 * v2);
    }
}

// u [n,2*O] = [x|gate] -> x_out = act ? SiLU(x) : x ;  v_out[sp] = gate * v2[sp]        (:950-959)
__global__ void k_gate(const float *__restrict__ u, const float *__restrict__ v2, int n, int O, int act, float *__restrict__ xo, float *__restrict__ vo) {
    const long long tot = (long long)n * O, stride = (long long)gridDim.x * blockDim.x;
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < tot; t += stride) {
        const int a = (int)(t / O), c = (int)(t - (long long)a * O);
        const float xv = u[(size_t)a * 2 * O + c], g = u[(size_t)a * 2 * O + O + c];
        xo[t] = act ? silu_f(xv) : xv;
        for (int sp = 0; sp < 3; ++sp) vo[((size_t)a * 3 + sp) * O + c] = g * v2[((size_t)a * 3 + sp) * O + c];
    }
}

// out = x * std + atomref[z]     (visnet.py:148-156; Atomref :1051-1058)
__global__ void k_prior(const float *__restrict__ x, const int64_t *__restrict__ z, const float *__restrict__ atomref, const float *__restrict__ stdp,
                        int n, int O, float *__restrict__ out) {
    const long long tot = (long long)n * O, stride = (long long)gridDim.x * blockDim.x;
    const float sd = *stdp;
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < tot; t += stride) out[t] = x[t] * sd + atomref[z[t / O]];
}

inline int nblk(long long n) { long long b = (n + 255) / 256; return (int)(b < 1 ? 1 : (b > 4096 ? 4096 : b)); }

}  // namespace

#define VN_CHECK(cond) if (!(cond)) return CONAN_E_BADARG
extern "C" {

int conan_visnet_edge_unit(const float *pos, const int *col, const int *tgt, const int *num_edges_dev, int max_edges, float *dvec, void *stream) {
    VN_CHECK(pos && col && tgt && num_edges_dev && dvec && max_edges >= 0);
    k_edge_unit<<<nblk(max_edges), 256, 0, as_stream(stream)>>>(pos, col, tgt, num_edges_dev, max_edges, dvec);
    CONAN_LAUNCH_CHECK(); return CONAN_OK;
}
int conan_visnet_expnormal(const float *dist, const int *num_edges_dev, int max_edges, const float *means, const float *betas, int num_rbf,
                           float alpha, float cutoff, float *out, void *stream) {
    VN_CHECK(dist && num_edges_dev && means && betas && out && num_rbf > 0);
    k_expnormal<<<nblk((long long)max_edges * num_rbf), 256, 0, as_stream(stream)>>>(dist, num_edges_dev, max_edges, means, betas, num_rbf, alpha, cutoff, out);
    CONAN_LAUNCH_CHECK(); return CONAN_OK;
}
int conan_visnet_neighbor_scale_to(const float *W, const float *dist, const int *col, const int *tgt, const int *num_edges_dev, int max_edges, int H,
                                   float cutoff, float *out, void *stream) {
    VN_CHECK(W && out && dist && col && tgt && num_edges_dev && H > 0);
    if (max_edges <= 0) return CONAN_OK;
    if (H == 128 && V_HALF) k_ne_scale<4, true><<<nblk((long long)max_edges * (64 / VN_RUN)), 256, 0, as_stream(stream)>>>(W, out, dist, col, tgt, num_edges_dev, max_edges, H, cutoff);
    else if (H % 128 == 0) k_ne_scale<2><<<nblk((long long)max_edges * (64 / VN_RUN)), 256, 0, as_stream(stream)>>>(W, out, dist, col, tgt, num_edges_dev, max_edges, H, cutoff);
    else k_ne_scale<1><<<nblk((long long)max_edges * (64 / VN_RUN)), 256, 0, as_stream(stream)>>>(W, out, dist, col, tgt, num_edges_dev, max_edges, H, cutoff);
    CONAN_LAUNCH_CHECK(); return CONAN_OK;
}
int conan_visnet_neighbor_scale(float *W, const float *dist, const int *col, const int *tgt, const int *num_edges_dev, int max_edges, int H,
                                float cutoff, void *stream) {
    return conan_visnet_neighbor_scale_to(W, dist, col, tgt, num_edges_dev, max_edges, H, cutoff, W, stream);
}
int conan_concat2(const float *a, int Ha, const float *b, int Hb, long long rows, float *out, void *stream) {
    VN_CHECK(a && b && out && Ha > 0 && Hb > 0 && rows >= 0);
    k_concat2<<<nblk(rows * (Ha + Hb)), 256, 0, as_stream(stream)>>>(a, Ha, b, Hb, rows, out);
    CONAN_LAUNCH_CHECK(); return CONAN_OK;
}
int conan_visnet_edge_embed(const float *x, const float *p, const int *col, const int *tgt, const int *num_edges_dev, int max_edges, int H,
                            float *f, void *stream) {
    VN_CHECK(x && p && col && tgt && num_edges_dev && f && H > 0);
    if (H == 128 && V_HALF) k_edge_embed<4, true><<<nblk((long long)max_edges * (64 / VN_RUN)), 256, 0, as_stream(stream)>>>(x, p, col, tgt, num_edges_dev, max_edges, H, f);
    else if (H % 128 == 0) k_edge_embed<2><<<nblk((long long)max_edges * (64 / VN_RUN)), 256, 0, as_stream(stream)>>>(x, p, col, tgt, num_edges_dev, max_edges, H, f);
    else k_edge_embed<1><<<nblk((long long)max_edges * (64 / VN_RUN)), 256, 0, as_stream(stream)>>>(x, p, col, tgt, num_edges_dev, max_edges, H, f);
    CONAN_LAUNCH_CHECK(); return CONAN_OK;
}
int conan_layernorm_fwd(const float *x, const float *gamma, const float *beta, int rows, int H, float eps, float *out, void *stream) {
    VN_CHECK(x && gamma && beta && out && rows >= 0 && H > 0);
    if (rows == 0) return CONAN_OK;
    k_layernorm<<<nblk((long long)rows * 64), 256, 0, as_stream(stream)>>>(x, gamma, beta, rows, H, eps, out);
    CONAN_LAUNCH_CHECK(); return CONAN_OK;
}
static int scale_channels_launch(const float *v, const float *w, const float *add, long long rows, int H, float *out, void *stream) {
    if (rows == 0) return CONAN_OK;
    const bool al = ((reinterpret_cast<uintptr_t>(v) | reinterpret_cast<uintptr_t>(w) | reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(add)) & 15) == 0;
    if ((H & 3) == 0 && al)
        k_scale_channels4<<<nblk(rows * (H / 4)), 256, 0, as_stream(stream)>>>(reinterpret_cast<const float4 *>(v), reinterpret_cast<const float4 *>(w),
                                                                                 reinterpret_cast<const float4 *>(add), rows, H / 4, reinterpret_cast<float4 *>(out));
    else k_scale_channels<<<nblk(rows * H), 256, 0, as_stream(stream)>>>(v, w, add, rows, H, out);
    CONAN_LAUNCH_CHECK(); return CONAN_OK;
}
int conan_scale_channels(const float *v, const float *w, long long rows, int H, float *out, void *stream) {
    VN_CHECK(v && w && out && rows >= 0 && H > 0);
    return scale_channels_launch(v, w, nullptr, rows, H, out, stream);
}
int conan_scale_channels_add(const float *v, const float *w, const float *add, long long rows, int H, float *out, void *stream) {
    VN_CHECK(v && w && add && out && rows >= 0 && H > 0);
    return scale_channels_launch(v, w, add, rows, H, out, stream);
}
int conan_visnet_vecdot(const float *vp, int n, int H, float *out, void *stream) {
    VN_CHECK(vp && out && n >= 0 && H > 0);
    k_vecdot<<<nblk((long long)n * H), 256, 0, as_stream(stream)>>>(vp, n, H, out);
    CONAN_LAUNCH_CHECK(); return CONAN_OK;
}
int conan_visnet_attn_message(const float *q, const float *k, const float *v, const float *dk, const float *dv, const int *rowptr,
                              const int *col, const float *dist, float cutoff, int n, int H, int num_heads, int pre_act, float *vmsg,
                              float *xagg, void *stream) {
    VN_CHECK(q && k && v && dk && dv && rowptr && col && dist && vmsg && xagg && n >= 0 && H > 0 && num_heads > 0 && H % num_heads == 0);
    const int hd = H / num_heads;
    // H a multiple of 128 (the classification backbone's 512, common.py:444-446): blocks of 128 channels on blockIdx.y, each a half-wavefront
    // per edge — heads must not straddle a block (hd divides 128) and span a power-of-two number of 4-channel lanes
    const bool blocks128 = H % 128 == 0 && V_HALF && hd % 4 == 0 && (((hd / 4) & (hd / 4 - 1)) == 0) && 128 % hd == 0;
    const int cpl = H > 64 ? (H + 63) / 64 : 1;
    if (!blocks128 && (H > 128 || (H > 64 && H != 128) || hd % cpl != 0)) return CONAN_E_UNSUPPORTED;
    const int lph = blocks128 ? hd / 4 : hd / cpl;
    if (lph & (lph - 1)) return CONAN_E_UNSUPPORTED;
    if (n == 0) return CONAN_OK;
    if (blocks128)
        k_attn_msg<4, true><<<dim3(nblk((long long)n * 64), H / 128), 256, 0, as_stream(stream)>>>(q, k, v, dk, dv, rowptr, col, dist, cutoff, n, H, hd / 4, pre_act, vmsg, xagg);
    else if (cpl == 2) k_attn_msg<2><<<nblk((long long)n * 64), 256, 0, as_stream(stream)>>>(q, k, v, dk, dv, rowptr, col, dist, cutoff, n, H, lph, pre_act, vmsg, xagg);
    else k_attn_msg<1><<<nblk((long long)n * 64), 256, 0, as_stream(stream)>>>(q, k, v, dk, dv, rowptr, col, dist, cutoff, n, H, lph, pre_act, vmsg, xagg);
    CONAN_LAUNCH_CHECK(); return CONAN_OK;
}
int conan_visnet_vec_aggregate(const float *vec, const float *s, const float *dvec, const int *rowptr, const int *col, int n, int H,
                               int pre_act, float *vagg, void *stream) {
    VN_CHECK(vec && s && dvec && rowptr && col && vagg && n >= 0 && H > 0);
    if (n == 0) return CONAN_OK;
    if (H == 128 && V_HALF) k_vec_aggregate<4, true><<<nblk((long long)n * 64), 256, 0, as_stream(stream)>>>(vec, s, dvec, rowptr, col, n, H, pre_act, vagg);
    else if (H == 128) k_vec_aggregate<2><<<nblk((long long)n * 64), 256, 0, as_stream(stream)>>>(vec, s, dvec, rowptr, col, n, H, pre_act, vagg);
    else if (H == 64) k_vec_aggregate<1><<<nblk((long long)n * 64), 256, 0, as_stream(stream)>>>(vec, s, dvec, rowptr, col, n, H, pre_act, vagg);
    else k_vec_aggregate_any<<<nblk((long long)n * 64), 256, 0, as_stream(stream)>>>(vec, s, dvec, rowptr, col, n, H, pre_act, vagg);
    CONAN_LAUNCH_CHECK(); return CONAN_OK;
}
int conan_visnet_node_update(const float *x, const float *vec, const float *vdot, const float *o, const float *vp, const float *vagg, int n,
                             int H, float *x_out, float *vec_out, void *stream) {
    VN_CHECK(x && vec && vdot && o && vp && vagg && x_out && vec_out && n >= 0 && H > 0);
    k_node_update<<<nblk((long long)n * H), 256, 0, as_stream(stream)>>>(x, vec, vdot, o, vp, vagg, n, H, x_out, vec_out);
    CONAN_LAUNCH_CHECK(); return CONAN_OK;
}
int conan_visnet_edge_update(const float *wt, const float *ws, const float *t, const float *dvec, const int *col, const int *tgt,
                             const int *num_edges_dev, int max_edges, int H, int pre_act, const float *f, float *f_out, void *stream) {
    VN_CHECK(wt && ws && t && dvec && col && tgt && num_edges_dev && f && f_out && H > 0);
    if (H == 128 && V_HALF) k_edge_update<4, true><<<nblk((long long)max_edges * (64 / VN_RUN)), 256, 0, as_stream(stream)>>>(wt, ws, t, dvec, col, tgt, num_edges_dev, max_edges, H, pre_act, f, f_out);
    else if (H % 128 == 0) k_edge_update<2><<<nblk((long long)max_edges * (64 / VN_RUN)), 256, 0, as_stream(stream)>>>(wt, ws, t, dvec, col, tgt, num_edges_dev, max_edges, H, pre_act, f, f_out);
    else k_edge_update<1><<<nblk((long long)max_edges * (64 / VN_RUN)), 256, 0, as_stream(stream)>>>(wt, ws, t, dvec, col, tgt, num_edges_dev, max_edges, H, pre_act, f, f_out);
    CONAN_LAUNCH_CHECK(); return CONAN_OK;
}
int conan_visnet_spatial_norm(const float *v, int n, int H, float *out, void *stream) {
    VN_CHECK(v && out && n >= 0 && H > 0);
    k_spatial_norm<<<nblk((long long)n * H), 256, 0, as_stream(stream)>>>(v, n, H, out);
    CONAN_LAUNCH_CHECK(); return CONAN_OK;
}
int conan_visnet_gate(const float *u, const float *v2, int n, int out_channels, int scalar_activation, float *x_out, float *v_out, void *stream) {
    VN_CHECK(u && v2 && x_out && v_out && n >= 0 && out_channels > 0);
    k_gate<<<nblk((long long)n * out_channels), 256, 0, as_stream(stream)>>>(u, v2, n, out_channels, scalar_activation, x_out, v_out);
    CONAN_LAUNCH_CHECK(); return CONAN_OK;
}
int conan_visnet_prior(const float *x, const int64_t *z, const float *atomref, const float *std_dev, int n, int out_channels, float *out,
                       void *stream) {
    VN_CHECK(x && z && atomref && std_dev && out && n >= 0 && out_channels > 0);
    k_prior<<<nblk((long long)n * out_channels), 256, 0, as_stream(stream)>>>(x, z, atomref, std_dev, n, out_channels, out);
    CONAN_LAUNCH_CHECK(); return CONAN_OK;
}

}  // extern "C"
