// Covalent (2-D bond graph) branch: the two GATConv layers of GATBased (conan_fgw/src/model/graph_embeddings/gat.py:5-25),
// PyG 2.3.0 GATConv semantics with heads = 1, concat, negative_slope 0.2, add_self_loops (fill_value = "mean"), edge_dim = 3:
//     h = lin_src(x)                       a_src[j] = <h_j, att_src>   a_dst[i] = <h_i, att_dst>
//     pre(j->i) = a_src[j] + a_dst[i] + <lin_edge(ea_ji), att_edge>    (self loop: ea = mean of the incoming edge attributes)
//     alpha = softmax_{j in N(i) + {i}} leaky_relu(pre)                 out_i = sum_j alpha_ji h_j + bias
// <lin_edge(ea), att_edge> = <ea, v>, v = W_edge^T att_edge (edge_dim numbers per layer), so no [E, C] edge tensor exists.
// Mapping: one wavefront per node, lane <-> channel (C = 64 is exactly one wavefront; wider C strides the lanes); rows are
// bonds (a handful of edges), so the softmax is a serial loop.  Backward: gradients that collect over many edges are formed
// per target row and per source list (by-source CSR) — no float atomics, bitwise reproducible.
#include "common.h"

namespace {

constexpr int GAT_MAXD = 8;          // edge_dim supported by the register arrays

__device__ __forceinline__ float leaky(float v, float slope) { return v > 0.f ? v : slope * v; }

__global__ void k_gat_edge_vec(const float *__restrict__ w_edge, const float *__restrict__ att_edge, int C, int D, float *__restrict__ v) {
    const int d = threadIdx.x;
    if (d >= D) return;
    float s = 0.f;
    for (int c = 0; c < C; ++c) s += w_edge[(size_t)c * D + d] * att_edge[c];
    v[d] = s;
}
__global__ void k_gat_edge_vec_bwd(const float *__restrict__ w_edge, const float *__restrict__ att_edge, const float *__restrict__ dv, int C, int D,
                                   float *__restrict__ dw_edge, float *__restrict__ datt_edge) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    float s = 0.f;
    for (int d = 0; d < D; ++d) { dw_edge[(size_t)c * D + d] = att_edge[c] * dv[d]; s += w_edge[(size_t)c * D + d] * dv[d]; }
    datt_edge[c] = s;
}

__global__ void __launch_bounds__(256) k_gat_node_alpha(const float *__restrict__ h, const float *__restrict__ att_src, const float *__restrict__ att_dst,
                                                        int n, int C, float *__restrict__ a_src, float *__restrict__ a_dst) {
    const int lane = threadIdx.x & 63;
    const int i = __builtin_amdgcn_readfirstlane((int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6));
    if (i >= n) return;
    float s = 0.f, t = 0.f;
    for (int c = lane; c < C; c += 64) { const float v = h[(size_t)i * C + c]; s += v * att_src[c]; t += v * att_dst[c]; }
    s = wave_sum(s); t = wave_sum(t);
    if (lane == 0) { a_src[i] = s; a_dst[i] = t; }
}

// forward aggregation; also stores the attention coefficients (alpha[p] for by-target position p, alpha_self[i]).
// The row's edges are staged one per lane (source, <ea, v>, a_src[source]) — bonds: a handful per atom — so logits, max and
// sum are wavefront reductions and the gather of the h_j rows is fed by cross-lane reads; rows longer than 64 edges take
// the serial path below.
template <int DT = GAT_MAXD>
__device__ __forceinline__ float gat_dot(const float *__restrict__ ea, const float *vd, int D) {
    float dot = 0.f;
#pragma unroll
    for (int d = 0; d < DT; ++d) if (d < D) dot += ea[d] * vd[d];
    return dot;
}

__global__ void __launch_bounds__(256) k_gat_aggregate_fwd(const float *__restrict__ h, const float *__restrict__ a_src, const float *__restrict__ a_dst,
                                                           const int *__restrict__ rowptr, const int *__restrict__ col, const int *__restrict__ eid,
                                                           const float *__restrict__ edge_attr, int D, const float *__restrict__ v,
                                                           const float *__restrict__ bias, float slope, int n, int C, float *__restrict__ out,
                                                           float *__restrict__ alpha, float *__restrict__ alpha_self) {
    const int lane = threadIdx.x & 63;
    const int i = __builtin_amdgcn_readfirstlane((int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6));
    if (i >= n) return;
    const int e0 = rowptr[i], e1 = rowptr[i + 1], deg = e1 - e0;
    float vd[GAT_MAXD];
#pragma unroll
    for (int d = 0; d < GAT_MAXD; ++d) vd[d] = d < D ? v[d] : 0.f;
    const float ad = a_dst[i], asi = a_src[i];
    if (deg <= 64) {
        const bool on = lane < deg;
        const int j = on ? col[e0 + lane] : i;
        const float dot = on ? gat_dot(edge_attr + (size_t)eid[e0 + lane] * D, vd, D) : 0.f;
        const float logit = on ? leaky(a_src[j] + ad + dot, slope) : -3.0e38f;
        const float mean_dot = deg > 0 ? wave_sum(dot) / (float)deg : 0.f;
        const float l_self = leaky(asi + ad + mean_dot, slope);
        const float mx = fmaxf(wave_max(logit), l_self);
        const float ex = on ? expf(logit - mx) : 0.f, ex_self = expf(l_self - mx);
        const float inv = 1.0f / (wave_sum(ex) + ex_self + 1e-16f);          // torch_geometric.utils.softmax: exp / (sum + 1e-16)
        const float a = ex * inv, as = ex_self * inv;
        if (on) alpha[e0 + lane] = a;
        if (lane == 0) alpha_self[i] = as;
        for (int c = lane; c < C; c += 64) {
            float acc = as * h[(size_t)i * C + c];
            for (int k = 0; k < deg; ++k) acc += __shfl(a, k, 64) * h[(size_t)__shfl(j, k, 64) * C + c];
            out[(size_t)i * C + c] = acc + (bias ? bias[c] : 0.f);
        }
        return;
    }
    // ---- general path (deg > 64): serial passes over the row
    float acc_dot = 0.f;
    for (int p = e0; p < e1; ++p) acc_dot += gat_dot(edge_attr + (size_t)eid[p] * D, vd, D);
    const float mean_dot = acc_dot / (float)deg;
    const float l_self = leaky(asi + ad + mean_dot, slope);
    float mx = l_self;
    for (int p = e0; p < e1; ++p) mx = fmaxf(mx, leaky(a_src[col[p]] + ad + gat_dot(edge_attr + (size_t)eid[p] * D, vd, D), slope));
    float sum = expf(l_self - mx);
    for (int p = e0; p < e1; ++p) sum += expf(leaky(a_src[col[p]] + ad + gat_dot(edge_attr + (size_t)eid[p] * D, vd, D), slope) - mx);
    const float inv = 1.0f / (sum + 1e-16f);
    const float as = expf(l_self - mx) * inv;
    if (lane == 0) alpha_self[i] = as;
    for (int c = lane; c < C; c += 64) {
        float acc = as * h[(size_t)i * C + c];
        for (int p = e0; p < e1; ++p) {
            const int j = col[p];
            const float a = expf(leaky(a_src[j] + ad + gat_dot(edge_attr + (size_t)eid[p] * D, vd, D), slope) - mx) * inv;
            if (c == lane && lane == 0) alpha[p] = a;
            acc += a * h[(size_t)j * C + c];
        }
        out[(size_t)i * C + c] = acc + (bias ? bias[c] : 0.f);
    }
}

// Backward.  Both kernels run on a fixed grid of GAT_BW_WGS workgroups x 4 wavefronts (node i -> wavefront i mod
// GAT_BW_WAVES: ~3 nodes per wavefront at 25 k atoms — the per-node work is a chain of dependent loads and reductions, so it
// needs many wavefronts in flight).  Every wavefront accumulates its share of the parameter gradients in registers — d att_src,
// d att_dst, d bias (source kernel, lane <-> channel) and dv (target kernel); the four wavefronts of a workgroup are combined
// through LDS into one partial row part[workgroup][3C + D], and k_gat_param_reduce sums the rows in a fixed order.  No float
// atomics anywhere: bitwise reproducible.
constexpr int GAT_BW_WGS = 2048;
constexpr int GAT_BW_WAVES = 4 * GAT_BW_WGS;

// target side: dpre per edge / self loop, da_dst, partial dv
// DT: size of the per-lane edge-attribute register arrays (4 for edge_dim <= 4, else DT): every entry costs wave-wide
// reductions, and the generic 8-wide body was 34 % instruction-fetch stalls (SQ_WAIT_INST_ANY) on top of them.
template <int DT>
__global__ void __launch_bounds__(256) k_gat_bwd_target(const float *__restrict__ h, const float *__restrict__ dout, const float *__restrict__ alpha,
                                                        const float *__restrict__ alpha_self, const float *__restrict__ a_src,
                                                        const float *__restrict__ a_dst, const int *__restrict__ rowptr, const int *__restrict__ col,
                                                        const int *__restrict__ eid, const float *__restrict__ edge_attr, int D,
                                                        const float *__restrict__ v, float slope, int n, int C, float *__restrict__ dpre,
                                                        float *__restrict__ dpre_self, float *__restrict__ da_dst, float *__restrict__ part, int PW) {
    const int lane = threadIdx.x & 63;
    const int wg = __builtin_amdgcn_readfirstlane((int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6));
    float vd[DT], dvacc[DT];
#pragma unroll
    for (int d = 0; d < DT; ++d) { vd[d] = d < D ? v[d] : 0.f; dvacc[d] = 0.f; }
    for (int i = wg; i < n; i += GAT_BW_WAVES) {
        const int e0 = rowptr[i], e1 = rowptr[i + 1], deg = e1 - e0;
        const float ad = a_dst[i], as = alpha_self[i];
        float dal_self = 0.f;
        for (int c = lane; c < C; c += 64) dal_self += dout[(size_t)i * C + c] * h[(size_t)i * C + c];
        dal_self = wave_sum(dal_self);
        float dad = 0.f;
        float mean_ea[DT];
#pragma unroll
        for (int d = 0; d < DT; ++d) mean_ea[d] = 0.f;
        float mean_dot = 0.f, S = as * dal_self;
        if (deg <= 64) {
            const bool on = lane < deg;
            const int j = on ? col[e0 + lane] : i;
            float ea[DT];
#pragma unroll
            for (int d = 0; d < DT; ++d) ea[d] = (on && d < D) ? edge_attr[(size_t)eid[e0 + lane] * D + d] : 0.f;
            float dot = 0.f;
#pragma unroll
            for (int d = 0; d < DT; ++d) dot += ea[d] * vd[d];
            const float al = on ? alpha[e0 + lane] : 0.f;
            const float pre = on ? a_src[j] + ad + dot : 0.f;
            float dal = 0.f;                                   // <dout_i, h_j> of the edge owned by this lane
            for (int k = 0; k < deg; ++k) {
                const int jk = __shfl(j, k, 64);
                float t = 0.f;
                for (int c = lane; c < C; c += 64) t += dout[(size_t)i * C + c] * h[(size_t)jk * C + c];
                t = wave_sum(t);
                if (lane == k) dal = t;
            }
            S += wave_sum(al * dal);
            if (deg > 0) {
                mean_dot = wave_sum(dot) / (float)deg;
#pragma unroll
                for (int d = 0; d < DT; ++d) mean_ea[d] = wave_sum(ea[d]) / (float)deg;
            }
            const float g = on ? al * (dal - S) * (pre > 0.f ? 1.f : slope) : 0.f;
            if (on) dpre[e0 + lane] = g;
            dad += wave_sum(g);
#pragma unroll
            for (int d = 0; d < DT; ++d) dvacc[d] += wave_sum(g * ea[d]);
        } else {
            for (int p = e0; p < e1; ++p) {
                const int j = col[p];
                float da = 0.f;
                for (int c = lane; c < C; c += 64) da += dout[(size_t)i * C + c] * h[(size_t)j * C + c];
                S += alpha[p] * wave_sum(da);
                const float *ea = edge_attr + (size_t)eid[p] * D;
#pragma unroll
                for (int d = 0; d < DT; ++d) if (d < D) { mean_ea[d] += ea[d]; mean_dot += ea[d] * vd[d]; }
            }
            mean_dot /= (float)deg;
#pragma unroll
            for (int d = 0; d < DT; ++d) mean_ea[d] /= (float)deg;
            for (int p = e0; p < e1; ++p) {
                const int j = col[p];
                float da = 0.f;
                for (int c = lane; c < C; c += 64) da += dout[(size_t)i * C + c] * h[(size_t)j * C + c];
                da = wave_sum(da);
                const float *ea = edge_attr + (size_t)eid[p] * D;
                const float pre = a_src[j] + ad + gat_dot<DT>(ea, vd, D);
                const float g = alpha[p] * (da - S) * (pre > 0.f ? 1.f : slope);
                if (lane == 0) dpre[p] = g;
                dad += g;
#pragma unroll
                for (int d = 0; d < DT; ++d) if (d < D) dvacc[d] += g * ea[d];
            }
        }
        {
            const float pre = a_src[i] + ad + mean_dot;
            const float g = as * (dal_self - S) * (pre > 0.f ? 1.f : slope);
            if (lane == 0) dpre_self[i] = g;
            dad += g;
#pragma unroll
            for (int d = 0; d < DT; ++d) dvacc[d] += g * mean_ea[d];
        }
        if (lane == 0) da_dst[i] = dad;
    }
    __shared__ float sm_dv[4][DT];
    if (lane == 0)
        for (int d = 0; d < DT; ++d) sm_dv[threadIdx.x >> 6][d] = dvacc[d];
    __syncthreads();
    if (threadIdx.x < D)
        part[(size_t)blockIdx.x * PW + 3 * C + threadIdx.x] =
            ((sm_dv[0][threadIdx.x] + sm_dv[1][threadIdx.x]) + sm_dv[2][threadIdx.x]) + sm_dv[3][threadIdx.x];
}

// source side: da_src, dh (messages sent by node j, its self loop, the two attention projections) and the partial sums of
// d att_src, d att_dst, d bias
__global__ void __launch_bounds__(256) k_gat_bwd_source(const float *__restrict__ h, const float *__restrict__ dout, const float *__restrict__ alpha,
                                                        const float *__restrict__ alpha_self, const float *__restrict__ dpre,
                                                        const float *__restrict__ dpre_self, const float *__restrict__ da_dst,
                                                        const float *__restrict__ att_src, const float *__restrict__ att_dst,
                                                        const int *__restrict__ t_rowptr, const int *__restrict__ t_pos, const int *__restrict__ t_tgt,
                                                        int n, int C, float *__restrict__ dh, float *__restrict__ part, int PW) {
    const int lane = threadIdx.x & 63;
    const int wg = __builtin_amdgcn_readfirstlane((int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6));
    constexpr int MAXV = 4;                                   // C <= 256
    float p_as[MAXV], p_ad[MAXV], p_b[MAXV];
#pragma unroll
    for (int u = 0; u < MAXV; ++u) { p_as[u] = 0.f; p_ad[u] = 0.f; p_b[u] = 0.f; }
    for (int j = wg; j < n; j += GAT_BW_WAVES) {
        const int q0 = t_rowptr[j], q1 = t_rowptr[j + 1], deg = q1 - q0;
        const float dad = da_dst[j], as = alpha_self[j];
        float das = dpre_self[j];
        if (deg <= 64) {
            const bool on = lane < deg;
            const int pos = on ? t_pos[q0 + lane] : 0;
            const int tg = on ? t_tgt[q0 + lane] : j;
            const float al = on ? alpha[pos] : 0.f;
            das += wave_sum(on ? dpre[pos] : 0.f);
#pragma unroll
            for (int u = 0; u < MAXV; ++u) {
                const int c = lane + 64 * u;
                if (c >= C) break;
                const float dj = dout[(size_t)j * C + c], hj = h[(size_t)j * C + c];
                float acc = as * dj;
                for (int k = 0; k < deg; ++k) acc += __shfl(al, k, 64) * dout[(size_t)__shfl(tg, k, 64) * C + c];
                dh[(size_t)j * C + c] = acc + das * att_src[c] + dad * att_dst[c];
                p_as[u] += das * hj; p_ad[u] += dad * hj; p_b[u] += dj;
            }
        } else {
            for (int q = q0; q < q1; ++q) das += dpre[t_pos[q]];
#pragma unroll
            for (int u = 0; u < MAXV; ++u) {
                const int c = lane + 64 * u;
                if (c >= C) break;
                const float dj = dout[(size_t)j * C + c], hj = h[(size_t)j * C + c];
                float acc = as * dj;
                for (int q = q0; q < q1; ++q) acc += alpha[t_pos[q]] * dout[(size_t)t_tgt[q] * C + c];
                dh[(size_t)j * C + c] = acc + das * att_src[c] + dad * att_dst[c];
                p_as[u] += das * hj; p_ad[u] += dad * hj; p_b[u] += dj;
            }
        }
    }
    __shared__ float sm_p[4][3][64 * MAXV];
    const int wv = threadIdx.x >> 6;
#pragma unroll
    for (int u = 0; u < MAXV; ++u) {
        const int c = lane + 64 * u;
        sm_p[wv][0][c] = p_as[u]; sm_p[wv][1][c] = p_ad[u]; sm_p[wv][2][c] = p_b[u];
    }
    __syncthreads();
    for (int t = threadIdx.x; t < 3 * C; t += 256) {
        const int k = t / C, c = t - k * C;
        part[(size_t)blockIdx.x * PW + t] = ((sm_p[0][k][c] + sm_p[1][k][c]) + sm_p[2][k][c]) + sm_p[3][k][c];
    }
}

// ---- round 3: the same two passes on 16-lane groups -----------------------------------------------------------------------
// The kernels above give a node a whole wavefront and walk its row as a chain of dependent loads and 6-step reductions: by itself
// the pair takes 36 + 14.5 us at 25 k atoms (3 nodes per wavefront, one L2 / Infinity-Cache round trip per link of the chain).  Here a
// node is owned by a group of 16 lanes (lane <-> CH = C/16 consecutive channels, float4 loads, 4-step reductions, 4 nodes per
// wavefront, one node per group at cfg2 size) and every level of indirection is ONE trip for the whole row: indices of all edges,
// then all rows (rows up to GAT_MD edges; longer ones walk serially).  Scalars of the row live in every lane of the group — no
// wave-wide reductions for the means and the parameter sums.  Parameter gradients: registers -> LDS (16 groups, fixed order) -> one
// partial row per workgroup -> k_gat_param_reduce.  No float atomics.  (A one-kernel form that recomputes the target sums two hops
// away was measured and rejected: 70 us — five levels of indirection in one chain, 190 VGPRs.  The forward aggregation in this
// form measured 24.8 us against 21.2 us for k_gat_aggregate_fwd, which already stages its row one edge per lane: not kept.)
// grid: one node per group (16 groups per workgroup), rounded up to a multiple of 64 rows for k_gat_param_reduce, at most GAT_BW_WGS
static inline int gat_g_wgs(int n) { const int w = ((n + 15) / 16 + 63) / 64 * 64; return w > GAT_BW_WGS ? GAT_BW_WGS : w; }
#ifndef CONAN_GAT_MD
#define CONAN_GAT_MD 6
#endif
constexpr int GAT_MD = CONAN_GAT_MD;
template <int CH>
__device__ __forceinline__ void gat_row(const float *__restrict__ base, int node, int sub, float (&r)[CH]) {
    const float4 *p = reinterpret_cast<const float4 *>(base + (size_t)node * (16 * CH) + sub * CH);
#pragma unroll
    for (int u = 0; u < CH / 4; ++u) { const float4 t = p[u]; r[4 * u] = t.x; r[4 * u + 1] = t.y; r[4 * u + 2] = t.z; r[4 * u + 3] = t.w; }
}
template <int CH>
__device__ __forceinline__ float gat_dot16(const float (&a)[CH], const float (&b)[CH]) {
    float t = 0.f;
#pragma unroll
    for (int u = 0; u < CH; ++u) t += a[u] * b[u];
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) t += __shfl_xor(t, o, 16);
    return t;
}
// the 16 groups of a workgroup -> columns [c0, c0 + cols) of its partial row, fixed order; sm: [16][cols]
__device__ __forceinline__ void gat_part_row(const float *sm, int cols, float *__restrict__ dst, int cols_out) {
    for (int t = threadIdx.x; t < cols_out; t += 256) {
        float r = sm[t];
#pragma unroll
        for (int g = 1; g < 16; ++g) r += sm[g * cols + t];
        dst[t] = r;
    }
}

// target side: dpre per edge / self loop, da_dst, partial dv
template <int CH, int DT>
__global__ void __launch_bounds__(256) k_gat_bwd_target16(const float *__restrict__ h, const float *__restrict__ dout, const float *__restrict__ alpha,
                                                          const float *__restrict__ alpha_self, const float *__restrict__ a_src,
                                                          const float *__restrict__ a_dst, const int *__restrict__ rowptr, const int *__restrict__ col,
                                                          const int *__restrict__ eid, const float *__restrict__ edge_attr, int D,
                                                          const float *__restrict__ v, float slope, int n, float *__restrict__ dpre,
                                                          float *__restrict__ dpre_self, float *__restrict__ da_dst, float *__restrict__ part, int PW) {
    constexpr int C = 16 * CH, M = GAT_MD;
    const int sub = threadIdx.x & 15, gl = threadIdx.x >> 4;
    const int ngroups = gridDim.x * 16;
    float vd[DT], dvacc[DT];
#pragma unroll
    for (int d = 0; d < DT; ++d) { vd[d] = d < D ? v[d] : 0.f; dvacc[d] = 0.f; }
    for (int j = blockIdx.x * 16 + gl; j < n; j += ngroups) {
        const int e0 = rowptr[j], e1 = rowptr[j + 1], deg = e1 - e0;
        const float ad_j = a_dst[j], as_j = alpha_self[j], asrc_j = a_src[j];
        float dj[CH], hj[CH];
        gat_row<CH>(dout, j, sub, dj);
        gat_row<CH>(h, j, sub, hj);
        const float dal_self = gat_dot16<CH>(dj, hj);
        float S = as_j * dal_self, sum_dot = 0.f, dad = 0.f, mean_ea[DT];
#pragma unroll
        for (int d = 0; d < DT; ++d) mean_ea[d] = 0.f;
        const float inv_deg = deg > 0 ? 1.0f / (float)deg : 0.f;
        if (deg <= M) {
            int kk[M], ee[M];
            float al[M];
#pragma unroll
            for (int t = 0; t < M; ++t) { kk[t] = t < deg ? col[e0 + t] : j; ee[t] = t < deg ? eid[e0 + t] : 0; al[t] = t < deg ? alpha[e0 + t] : 0.f; }
            float hk[M][CH], ask[M], eav[M][DT], dal[M];
#pragma unroll
            for (int t = 0; t < M; ++t) {
                gat_row<CH>(h, kk[t], sub, hk[t]);
                ask[t] = a_src[kk[t]];
#pragma unroll
                for (int d = 0; d < DT; ++d) eav[t][d] = (d < D && t < deg) ? edge_attr[(size_t)ee[t] * D + d] : 0.f;
            }
#pragma unroll
            for (int t = 0; t < M; ++t) {
                dal[t] = gat_dot16<CH>(dj, hk[t]);
                if (t < deg) {
                    S += al[t] * dal[t];
#pragma unroll
                    for (int d = 0; d < DT; ++d) if (d < D) { mean_ea[d] += eav[t][d]; sum_dot += eav[t][d] * vd[d]; }
                }
            }
#pragma unroll
            for (int t = 0; t < M; ++t)
                if (t < deg) {
                    float dot = 0.f;
#pragma unroll
                    for (int d = 0; d < DT; ++d) if (d < D) dot += eav[t][d] * vd[d];
                    const float pre = ask[t] + ad_j + dot;
                    const float g = al[t] * (dal[t] - S) * (pre > 0.f ? 1.f : slope);
                    if (sub == 0) dpre[e0 + t] = g;
                    dad += g;
#pragma unroll
                    for (int d = 0; d < DT; ++d) if (d < D) dvacc[d] += g * eav[t][d];
                }
        } else {
            for (int p = e0; p < e1; ++p) {
                float hk[CH];
                gat_row<CH>(h, col[p], sub, hk);
                S += alpha[p] * gat_dot16<CH>(dj, hk);
                const float *ea = edge_attr + (size_t)eid[p] * D;
#pragma unroll
                for (int d = 0; d < DT; ++d) if (d < D) { mean_ea[d] += ea[d]; sum_dot += ea[d] * vd[d]; }
            }
            for (int p = e0; p < e1; ++p) {
                const int k = col[p];
                float hk[CH];
                gat_row<CH>(h, k, sub, hk);
                const float dal = gat_dot16<CH>(dj, hk);
                const float *ea = edge_attr + (size_t)eid[p] * D;
                const float pre = a_src[k] + ad_j + gat_dot<DT>(ea, vd, D);
                const float g = alpha[p] * (dal - S) * (pre > 0.f ? 1.f : slope);
                if (sub == 0) dpre[p] = g;
                dad += g;
#pragma unroll
                for (int d = 0; d < DT; ++d) if (d < D) dvacc[d] += g * ea[d];
            }
        }
        const float pre_self = asrc_j + ad_j + sum_dot * inv_deg;
        const float g_self = as_j * (dal_self - S) * (pre_self > 0.f ? 1.f : slope);
        dad += g_self;
#pragma unroll
        for (int d = 0; d < DT; ++d) dvacc[d] += g_self * (mean_ea[d] * inv_deg);
        if (sub == 0) { dpre_self[j] = g_self; da_dst[j] = dad; }
    }
    __shared__ float sm_dv[16 * DT];
    if (sub == 0)
#pragma unroll
        for (int d = 0; d < DT; ++d) sm_dv[gl * DT + d] = dvacc[d];
    __syncthreads();
    gat_part_row(sm_dv, DT, part + (size_t)blockIdx.x * PW + 3 * C, D);
}

// source side: da_src, dh (messages sent by node j, its self loop, the two attention projections) and the partial sums of d att_src, d att_dst, d bias
template <int CH>
__global__ void __launch_bounds__(256) k_gat_bwd_source16(const float *__restrict__ h, const float *__restrict__ dout, const float *__restrict__ alpha,
                                                          const float *__restrict__ alpha_self, const float *__restrict__ dpre,
                                                          const float *__restrict__ dpre_self, const float *__restrict__ da_dst,
                                                          const float *__restrict__ att_src, const float *__restrict__ att_dst,
                                                          const int *__restrict__ t_rowptr, const int *__restrict__ t_pos, const int *__restrict__ t_tgt,
                                                          int n, float *__restrict__ dh, float *__restrict__ part, int PW) {
    constexpr int C = 16 * CH, M = GAT_MD;
    const int sub = threadIdx.x & 15, gl = threadIdx.x >> 4;
    const int ngroups = gridDim.x * 16;
    float ats[CH], atd[CH], p_as[CH], p_ad[CH], p_b[CH];
#pragma unroll
    for (int u = 0; u < CH; ++u) { ats[u] = att_src[sub * CH + u]; atd[u] = att_dst[sub * CH + u]; p_as[u] = 0.f; p_ad[u] = 0.f; p_b[u] = 0.f; }
    for (int j = blockIdx.x * 16 + gl; j < n; j += ngroups) {
        const int q0 = t_rowptr[j], q1 = t_rowptr[j + 1], deg = q1 - q0;
        const float dad = da_dst[j], as_j = alpha_self[j];
        float das = dpre_self[j];
        float dj[CH], hj[CH], acc[CH];
        gat_row<CH>(dout, j, sub, dj);
        gat_row<CH>(h, j, sub, hj);
#pragma unroll
        for (int u = 0; u < CH; ++u) acc[u] = as_j * dj[u];
        if (deg <= M) {
            int op[M], oi[M];
#pragma unroll
            for (int t = 0; t < M; ++t) { op[t] = t < deg ? t_pos[q0 + t] : 0; oi[t] = t < deg ? t_tgt[q0 + t] : j; }
            float di[M][CH], oa[M], og[M];
#pragma unroll
            for (int t = 0; t < M; ++t) {
                gat_row<CH>(dout, oi[t], sub, di[t]);
                oa[t] = t < deg ? alpha[op[t]] : 0.f; og[t] = t < deg ? dpre[op[t]] : 0.f;
            }
#pragma unroll
            for (int t = 0; t < M; ++t)
                if (t < deg) {
                    das += og[t];
#pragma unroll
                    for (int u = 0; u < CH; ++u) acc[u] += oa[t] * di[t][u];
                }
        } else {
            for (int q = q0; q < q1; ++q) {
                const int pos = t_pos[q];
                float di[CH];
                gat_row<CH>(dout, t_tgt[q], sub, di);
                const float a = alpha[pos];
                das += dpre[pos];
#pragma unroll
                for (int u = 0; u < CH; ++u) acc[u] += a * di[u];
            }
        }
        float4 *o = reinterpret_cast<float4 *>(dh + (size_t)j * C + sub * CH);
#pragma unroll
        for (int u = 0; u < CH / 4; ++u)
            o[u] = make_float4(acc[4 * u] + das * ats[4 * u] + dad * atd[4 * u], acc[4 * u + 1] + das * ats[4 * u + 1] + dad * atd[4 * u + 1],
                               acc[4 * u + 2] + das * ats[4 * u + 2] + dad * atd[4 * u + 2], acc[4 * u + 3] + das * ats[4 * u + 3] + dad * atd[4 * u + 3]);
#pragma unroll
        for (int u = 0; u < CH; ++u) { p_as[u] += das * hj[u]; p_ad[u] += dad * hj[u]; p_b[u] += dj[u]; }
    }
    extern __shared__ float sm_f[];                                                   // [16][3 C]
#pragma unroll
    for (int u = 0; u < CH; ++u) {
        sm_f[gl * 3 * C + sub * CH + u] = p_as[u]; sm_f[gl * 3 * C + C + sub * CH + u] = p_ad[u]; sm_f[gl * 3 * C + 2 * C + sub * CH + u] = p_b[u];
    }
    __syncthreads();
    gat_part_row(sm_f, 3 * C, part + (size_t)blockIdx.x * PW, 3 * C);
}

// out[w] = sum over the GAT_BW_WGS partial rows, fixed order: a workgroup owns 32 columns and splits the rows 8 ways
__global__ void __launch_bounds__(256) k_gat_param_reduce(const float *__restrict__ part, int PW, float *__restrict__ out, int rows) {
    __shared__ float sm[8][32];
    const int g = threadIdx.x >> 5, c = threadIdx.x & 31, w = blockIdx.x * 32 + c;
    float a0 = 0.f, a1 = 0.f;
    if (w < PW) {
        for (int r = g; r < rows; r += 64) {                    // 8 loads in flight (rows: a multiple of 64)
            float t[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) t[u] = part[(size_t)(r + 8 * u) * PW + w];
#pragma unroll
            for (int u = 0; u < 8; u += 2) { a0 += t[u]; a1 += t[u + 1]; }
        }
    }
    sm[g][c] = a0 + a1;
    __syncthreads();
    if (g == 0 && w < PW) {
        float r = sm[0][c];
#pragma unroll
        for (int q = 1; q < 8; ++q) r += sm[q][c];
        out[w] = r;
    }
}

inline int wave_blocks(int n) { return (n + 3) / 4; }

}  // namespace

extern "C" {

int conan_gat_edge_vec(const float *w_edge, const float *att_edge, int channels, int edge_dim, float *v, void *stream) {
    if (!w_edge || !att_edge || !v || channels <= 0 || edge_dim <= 0 || edge_dim > GAT_MAXD) return CONAN_E_BADARG;
    k_gat_edge_vec<<<1, 64, 0, as_stream(stream)>>>(w_edge, att_edge, channels, edge_dim, v);
    CONAN_LAUNCH_CHECK();
    return CONAN_OK;
}

int conan_gat_edge_vec_bwd(const float *w_edge, const float *att_edge, const float *dv, int channels, int edge_dim, float *dw_edge,
                           float *datt_edge, void *stream) {
    if (!w_edge || !att_edge || !dv || !dw_edge || !datt_edge || channels <= 0 || edge_dim <= 0 || edge_dim > GAT_MAXD) return CONAN_E_BADARG;
    k_gat_edge_vec_bwd<<<(channels + 63) / 64, 64, 0, as_stream(stream)>>>(w_edge, att_edge, dv, channels, edge_dim, dw_edge, datt_edge);
    CONAN_LAUNCH_CHECK();
    return CONAN_OK;
}

int conan_gat_node_alpha(const float *h, const float *att_src, const float *att_dst, int n, int channels, float *a_src, float *a_dst,
                         void *stream) {
    if (!h || !att_src || !att_dst || !a_src || !a_dst || n < 0 || channels <= 0) return CONAN_E_BADARG;
    if (n == 0) return CONAN_OK;
    k_gat_node_alpha<<<wave_blocks(n), 256, 0, as_stream(stream)>>>(h, att_src, att_dst, n, channels, a_src, a_dst);
    CONAN_LAUNCH_CHECK();
    return CONAN_OK;
}

int conan_gat_aggregate_fwd(const float *h, const float *a_src, const float *a_dst, const int *rowptr, const int *col, const int *eid,
                            const float *edge_attr, int edge_dim, const float *v, const float *bias, float negative_slope, int n,
                            int channels, float *out, float *alpha, float *alpha_self, void *stream) {
    if (!h || !a_src || !a_dst || !rowptr || !v || !out || !alpha_self || n < 0 || channels <= 0 || edge_dim <= 0 || edge_dim > GAT_MAXD)
        return CONAN_E_BADARG;
    if (n == 0) return CONAN_OK;
    k_gat_aggregate_fwd<<<wave_blocks(n), 256, 0, as_stream(stream)>>>(h, a_src, a_dst, rowptr, col, eid, edge_attr, edge_dim, v, bias, negative_slope,
                                                                         n, channels, out, alpha, alpha_self);
    CONAN_LAUNCH_CHECK();
    return CONAN_OK;
}

long long conan_gat_bwd_ws(int n, int num_edges, int channels, int edge_dim) {
    return 2LL * n + (num_edges > 0 ? num_edges : 1) + (long long)GAT_BW_WGS * (3 * channels + edge_dim) + (3 * channels + edge_dim);
}

int conan_gat_aggregate_bwd(const float *h, const float *dout, const float *alpha, const float *alpha_self, const float *a_src,
                            const float *a_dst, const float *att_src, const float *att_dst, const int *rowptr, const int *col, const int *eid,
                            const int *t_rowptr, const int *t_pos, const int *t_tgt, const float *edge_attr, int edge_dim, const float *v,
                            float negative_slope, int n, int num_edges, int channels, float *ws, float *dh, float *dparams, void *stream) {
    if (!h || !dout || !alpha_self || !a_src || !a_dst || !att_src || !att_dst || !rowptr || !t_rowptr || !v || !ws || !dh || !dparams || n <= 0 ||
        channels <= 0 || channels > 256 || edge_dim <= 0 || edge_dim > GAT_MAXD || num_edges < 0)
        return CONAN_E_BADARG;
    hipStream_t s = as_stream(stream);
    const int PW = 3 * channels + edge_dim;
    float *dpre_self = ws, *da_dst = ws + n, *dpre = ws + 2 * (size_t)n, *part = dpre + (num_edges > 0 ? num_edges : 1);
    if (edge_dim <= 4 && (channels == 64 || channels == 128 || channels == 256)) {
        const int wgs16 = gat_g_wgs(n);
#define GAT_G16(CHN)                                                                                                                          \
        do {                                                                                                                                  \
            const size_t lds = (size_t)16 * 3 * 16 * CHN * sizeof(float);                                                                     \
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_gat_bwd_source16<CHN>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
            k_gat_bwd_target16<CHN, 4><<<wgs16, 256, 0, s>>>(h, dout, alpha, alpha_self, a_src, a_dst, rowptr, col, eid, edge_attr, edge_dim, v,  \
                                                                negative_slope, n, dpre, dpre_self, da_dst, part, PW);                        \
            k_gat_bwd_source16<CHN><<<wgs16, 256, lds, s>>>(h, dout, alpha, alpha_self, dpre, dpre_self, da_dst, att_src, att_dst, t_rowptr,    \
                                                               t_pos, t_tgt, n, dh, part, PW);                                                \
        } while (0)
        if (channels == 64) GAT_G16(4);
        else if (channels == 128) GAT_G16(8);
        else GAT_G16(16);
#undef GAT_G16
        k_gat_param_reduce<<<(PW + 31) / 32, 256, 0, s>>>(part, PW, dparams, wgs16);
        CONAN_LAUNCH_CHECK();
        return CONAN_OK;
    }
    if (edge_dim <= 4)
        k_gat_bwd_target<4><<<GAT_BW_WGS, 256, 0, s>>>(h, dout, alpha, alpha_self, a_src, a_dst, rowptr, col, eid, edge_attr, edge_dim, v, negative_slope, n,
                                                          channels, dpre, dpre_self, da_dst, part, PW);
    else
        k_gat_bwd_target<GAT_MAXD><<<GAT_BW_WGS, 256, 0, s>>>(h, dout, alpha, alpha_self, a_src, a_dst, rowptr, col, eid, edge_attr, edge_dim, v,
                                                                 negative_slope, n, channels, dpre, dpre_self, da_dst, part, PW);
    k_gat_bwd_source<<<GAT_BW_WGS, 256, 0, s>>>(h, dout, alpha, alpha_self, dpre, dpre_self, da_dst, att_src, att_dst, t_rowptr, t_pos, t_tgt, n,
                                                       channels, dh, part, PW);
    k_gat_param_reduce<<<(PW + 31) / 32, 256, 0, s>>>(part, PW, dparams, GAT_BW_WGS);
    CONAN_LAUNCH_CHECK();
    return CONAN_OK;
}

}  // extern "C"
