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
    const int i = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (i >= n) return;
    float s = 0.f, t = 0.f;
    for (int c = lane; c < C; c += 64) { const float v = h[(size_t)i * C + c]; s += v * att_src[c]; t += v * att_dst[c]; }
    s = wave_sum(s); t = wave_sum(t);
    if (lane == 0) { a_src[i] = s; a_dst[i] = t; }
}

// forward aggregation; also stores the attention coefficients (alpha[p] for by-target position p, alpha_self[i])
__global__ void __launch_bounds__(256) k_gat_aggregate_fwd(const float *__restrict__ h, const float *__restrict__ a_src, const float *__restrict__ a_dst,
                                                           const int *__restrict__ rowptr, const int *__restrict__ col, const int *__restrict__ eid,
                                                           const float *__restrict__ edge_attr, int D, const float *__restrict__ v,
                                                           const float *__restrict__ bias, float slope, int n, int C, float *__restrict__ out,
                                                           float *__restrict__ alpha, float *__restrict__ alpha_self) {
    const int lane = threadIdx.x & 63;
    const int i = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (i >= n) return;
    const int e0 = rowptr[i], e1 = rowptr[i + 1];
    float vd[GAT_MAXD];
#pragma unroll
    for (int d = 0; d < GAT_MAXD; ++d) vd[d] = d < D ? v[d] : 0.f;
    const float ad = a_dst[i];
    // pass 1: logits, running max, mean edge attribute (self-loop fill value)
    float mean_dot = 0.f, mx;
    {
        float acc = 0.f;
        for (int p = e0; p < e1; ++p) {
            const float *ea = edge_attr + (size_t)eid[p] * D;
            float dot = 0.f;
#pragma unroll
            for (int d = 0; d < GAT_MAXD; ++d) if (d < D) dot += ea[d] * vd[d];
            acc += dot;
        }
        mean_dot = e1 > e0 ? acc / (float)(e1 - e0) : 0.f;      // <mean(ea), v> == mean(<ea, v>) up to rounding; see gat.py oracle
    }
    const float l_self = leaky(a_src[i] + ad + mean_dot, slope);
    mx = l_self;
    for (int p = e0; p < e1; ++p) {
        const float *ea = edge_attr + (size_t)eid[p] * D;
        float dot = 0.f;
#pragma unroll
        for (int d = 0; d < GAT_MAXD; ++d) if (d < D) dot += ea[d] * vd[d];
        mx = fmaxf(mx, leaky(a_src[col[p]] + ad + dot, slope));
    }
    float sum = expf(l_self - mx);
    for (int p = e0; p < e1; ++p) {
        const float *ea = edge_attr + (size_t)eid[p] * D;
        float dot = 0.f;
#pragma unroll
        for (int d = 0; d < GAT_MAXD; ++d) if (d < D) dot += ea[d] * vd[d];
        sum += expf(leaky(a_src[col[p]] + ad + dot, slope) - mx);
    }
    const float inv = 1.0f / (sum + 1e-16f);                    // torch_geometric.utils.softmax: exp / (sum + 1e-16)
    const float as = expf(l_self - mx) * inv;
    if (lane == 0) alpha_self[i] = as;
    for (int c = lane; c < C; c += 64) {
        float acc = as * h[(size_t)i * C + c];
        for (int p = e0; p < e1; ++p) {
            const float *ea = edge_attr + (size_t)eid[p] * D;
            float dot = 0.f;
#pragma unroll
            for (int d = 0; d < GAT_MAXD; ++d) if (d < D) dot += ea[d] * vd[d];
            const int j = col[p];
            const float a = expf(leaky(a_src[j] + ad + dot, slope) - mx) * inv;
            if (c == lane && lane == 0) alpha[p] = a;
            acc += a * h[(size_t)j * C + c];
        }
        out[(size_t)i * C + c] = acc + (bias ? bias[c] : 0.f);
    }
}

// backward, target side: dpre per edge / self loop, da_dst, per-node contribution to dv
__global__ void __launch_bounds__(256) k_gat_bwd_target(const float *__restrict__ h, const float *__restrict__ dout, const float *__restrict__ alpha,
                                                        const float *__restrict__ alpha_self, const float *__restrict__ a_src,
                                                        const float *__restrict__ a_dst, const int *__restrict__ rowptr, const int *__restrict__ col,
                                                        const int *__restrict__ eid, const float *__restrict__ edge_attr, int D,
                                                        const float *__restrict__ v, float slope, int n, int C, float *__restrict__ dpre,
                                                        float *__restrict__ dpre_self, float *__restrict__ da_dst, float *__restrict__ dv_part) {
    const int lane = threadIdx.x & 63;
    const int i = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (i >= n) return;
    const int e0 = rowptr[i], e1 = rowptr[i + 1];
    float vd[GAT_MAXD];
#pragma unroll
    for (int d = 0; d < GAT_MAXD; ++d) vd[d] = d < D ? v[d] : 0.f;
    const float ad = a_dst[i];
    // d alpha = <dout_i, h_j>;  S = sum alpha * dalpha
    float dal_self = 0.f;
    for (int c = lane; c < C; c += 64) dal_self += dout[(size_t)i * C + c] * h[(size_t)i * C + c];
    dal_self = wave_sum(dal_self);
    const float as = alpha_self[i];
    float S = as * dal_self;
    for (int p = e0; p < e1; ++p) {
        const int j = col[p];
        float da = 0.f;
        for (int c = lane; c < C; c += 64) da += dout[(size_t)i * C + c] * h[(size_t)j * C + c];
        da = wave_sum(da);
        S += alpha[p] * da;
    }
    float mean_ea[GAT_MAXD], dvp[GAT_MAXD];
#pragma unroll
    for (int d = 0; d < GAT_MAXD; ++d) { mean_ea[d] = 0.f; dvp[d] = 0.f; }
    float mean_dot = 0.f;
    for (int p = e0; p < e1; ++p) {
        const float *ea = edge_attr + (size_t)eid[p] * D;
        float dot = 0.f;
#pragma unroll
        for (int d = 0; d < GAT_MAXD; ++d) if (d < D) { mean_ea[d] += ea[d]; dot += ea[d] * vd[d]; }
        mean_dot += dot;
    }
    if (e1 > e0) {
        const float r = 1.0f / (float)(e1 - e0);
        mean_dot = mean_dot / (float)(e1 - e0);
#pragma unroll
        for (int d = 0; d < GAT_MAXD; ++d) mean_ea[d] *= r;
    }
    float dad = 0.f;
    {
        const float pre = a_src[i] + ad + mean_dot;
        const float g = as * (dal_self - S) * (pre > 0.f ? 1.f : slope);
        if (lane == 0) dpre_self[i] = g;
        dad += g;
#pragma unroll
        for (int d = 0; d < GAT_MAXD; ++d) dvp[d] += g * mean_ea[d];
    }
    for (int p = e0; p < e1; ++p) {
        const int j = col[p];
        float da = 0.f;
        for (int c = lane; c < C; c += 64) da += dout[(size_t)i * C + c] * h[(size_t)j * C + c];
        da = wave_sum(da);
        const float *ea = edge_attr + (size_t)eid[p] * D;
        float dot = 0.f;
#pragma unroll
        for (int d = 0; d < GAT_MAXD; ++d) if (d < D) dot += ea[d] * vd[d];
        const float pre = a_src[j] + ad + dot;
        const float g = alpha[p] * (da - S) * (pre > 0.f ? 1.f : slope);
        if (lane == 0) dpre[p] = g;
        dad += g;
#pragma unroll
        for (int d = 0; d < GAT_MAXD; ++d) if (d < D) dvp[d] += g * ea[d];
    }
    if (lane == 0) {
        da_dst[i] = dad;
        for (int d = 0; d < D; ++d) dv_part[(size_t)i * D + d] = dvp[d];
    }
}

// backward, source side: da_src and dh (messages sent by node j, its self loop, and the two attention projections)
__global__ void __launch_bounds__(256) k_gat_bwd_source(const float *__restrict__ dout, const float *__restrict__ alpha, const float *__restrict__ alpha_self,
                                                        const float *__restrict__ dpre, const float *__restrict__ dpre_self,
                                                        const float *__restrict__ da_dst, const float *__restrict__ att_src,
                                                        const float *__restrict__ att_dst, const int *__restrict__ t_rowptr,
                                                        const int *__restrict__ t_pos, const int *__restrict__ t_tgt, int n, int C,
                                                        float *__restrict__ dh, float *__restrict__ da_src) {
    const int lane = threadIdx.x & 63;
    const int j = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (j >= n) return;
    const int q0 = t_rowptr[j], q1 = t_rowptr[j + 1];
    float das = dpre_self[j];
    for (int q = q0; q < q1; ++q) das += dpre[t_pos[q]];
    if (lane == 0) da_src[j] = das;
    const float dad = da_dst[j], as = alpha_self[j];
    for (int c = lane; c < C; c += 64) {
        float acc = as * dout[(size_t)j * C + c];
        for (int q = q0; q < q1; ++q) acc += alpha[t_pos[q]] * dout[(size_t)t_tgt[q] * C + c];
        dh[(size_t)j * C + c] = acc + das * att_src[c] + dad * att_dst[c];
    }
}

// deterministic column sums: out[c] = sum_r x[r, c]; stage 1 per 256-row chunk, stage 2 over the chunks
constexpr int CS_CHUNK = 256;
__global__ void __launch_bounds__(256) k_colsum_partial(const float *__restrict__ x, int rows, int width, float *__restrict__ part) {
    const int r0 = blockIdx.x * CS_CHUNK, r1 = min(rows, r0 + CS_CHUNK);
    for (int c = threadIdx.x; c < width; c += 256) {
        float s = 0.f;
        for (int r = r0; r < r1; ++r) s += x[(size_t)r * width + c];
        part[(size_t)blockIdx.x * width + c] = s;
    }
}
__global__ void k_colsum_final(const float *__restrict__ part, int chunks, int width, float *__restrict__ out) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= width) return;
    float s = 0.f;
    for (int k = 0; k < chunks; ++k) s += part[(size_t)k * width + c];
    out[c] = s;
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

int conan_gat_aggregate_bwd(const float *h, const float *dout, const float *alpha, const float *alpha_self, const float *a_src,
                            const float *a_dst, const float *att_src, const float *att_dst, const int *rowptr, const int *col, const int *eid,
                            const int *t_rowptr, const int *t_pos, const int *t_tgt, const float *edge_attr, int edge_dim, const float *v,
                            float negative_slope, int n, int channels, float *dpre_ws, float *dh, float *da_src, float *da_dst, float *dv_part,
                            void *stream) {
    if (!h || !dout || !alpha_self || !a_src || !a_dst || !att_src || !att_dst || !rowptr || !t_rowptr || !v || !dpre_ws || !dh || !da_src || !da_dst ||
        !dv_part || n < 0 || channels <= 0 || edge_dim <= 0 || edge_dim > GAT_MAXD)
        return CONAN_E_BADARG;
    if (n == 0) return CONAN_OK;
    hipStream_t s = as_stream(stream);
    float *dpre_self = dpre_ws, *dpre = dpre_ws + n;             // [n] + [E]
    k_gat_bwd_target<<<wave_blocks(n), 256, 0, s>>>(h, dout, alpha, alpha_self, a_src, a_dst, rowptr, col, eid, edge_attr, edge_dim, v, negative_slope, n,
                                                     channels, dpre, dpre_self, da_dst, dv_part);
    k_gat_bwd_source<<<wave_blocks(n), 256, 0, s>>>(dout, alpha, alpha_self, dpre, dpre_self, da_dst, att_src, att_dst, t_rowptr, t_pos, t_tgt, n, channels,
                                                     dh, da_src);
    CONAN_LAUNCH_CHECK();
    return CONAN_OK;
}

long long conan_colsum_ws(int rows, int width) { return (long long)((rows + CS_CHUNK - 1) / CS_CHUNK) * width; }

int conan_colsum(const float *x, int rows, int width, float *out, float *ws, void *stream) {
    if (!out || width <= 0 || rows < 0 || (rows && (!x || !ws))) return CONAN_E_BADARG;
    hipStream_t s = as_stream(stream);
    const int chunks = (rows + CS_CHUNK - 1) / CS_CHUNK;
    if (chunks) k_colsum_partial<<<chunks, 256, 0, s>>>(x, rows, width, ws);
    k_colsum_final<<<(width + 255) / 256, 256, 0, s>>>(ws, chunks, width, out);
    CONAN_LAUNCH_CHECK();
    return CONAN_OK;
}

}  // extern "C"
