// Node-level glue kernels: atom-type embedding lookup and the per-conformer sum readout, with their backward.
#include "common.h"

namespace {

// An index outside [0, rows) (torch.nn.Embedding device-asserts there) never reads out of bounds: its output row is NaN,
// which poisons the loss visibly instead of corrupting silently.
__global__ void k_embedding_fwd(const int64_t *__restrict__ z, const float *__restrict__ weight, int n, int H4, int rows,
                                float *__restrict__ out) {
    const long long total = (long long)n * H4;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        int a = (int)(i / H4), c = (int)(i - (long long)a * H4);
        const long long r = z[a];
        const float qn = __builtin_nanf("");
        reinterpret_cast<float4 *>(out)[i] = (r >= 0 && r < rows) ? reinterpret_cast<const float4 *>(weight)[(size_t)r * H4 + c]
                                                                   : make_float4(qn, qn, qn, qn);
    }
}

// Deterministic embedding gradient, two stages.  Stage 1: one workgroup per (chunk of EMB_CHUNK atoms, 128-column tile) keeps a
// [num_embeddings x 128] partial table in LDS; thread c owns column c, walks the chunk's atoms in order and adds
// dout[a][c] into row z[a] (no atomics, fixed order).  Stage 2 sums the per-chunk tables in chunk order.
constexpr int EMB_CHUNK = 128;      // atoms per workgroup: the chunk is walked serially (4 batches of 32 loads), 198 workgroups at 25 k atoms
constexpr int EMB_ROWS_MAX = 100;     // torch.nn.Embedding(100, H): atomic numbers
__global__ void __launch_bounds__(128) k_embedding_bwd_partial(const int64_t *__restrict__ z, const float *__restrict__ dout, int n, int H,
                                                              int rows, float *__restrict__ slabs) {
    __shared__ float acc[EMB_ROWS_MAX * 128];
    __shared__ int zs[EMB_CHUNK];
    const int chunk = blockIdx.x, c = blockIdx.y * 128 + threadIdx.x;
    const int a0 = chunk * EMB_CHUNK, a1 = min(n, a0 + EMB_CHUNK);
    for (int t = threadIdx.x; t < rows * 128; t += 128) acc[t] = 0.f;
    for (int t = threadIdx.x; t < a1 - a0; t += 128) {          // out-of-range indices (NaN rows in the forward) add nothing
        const long long r = z[a0 + t];
        zs[t] = (r >= 0 && r < rows) ? (int)r : -1;
    }
    __syncthreads();
    if (c < H) {                            // loads issued 32 deep (a batch costs one memory round trip), then the order-preserving LDS accumulation
        constexpr int U = 32;
        int a = a0;
        for (; a + U <= a1; a += U) {
            float v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) v[u] = dout[(size_t)(a + u) * H + c];
#pragma unroll
            for (int u = 0; u < U; ++u) { const int r = zs[a + u - a0]; if (r >= 0) acc[r * 128 + threadIdx.x] += v[u]; }
        }
        for (; a < a1; ++a) { const int r = zs[a - a0]; if (r >= 0) acc[r * 128 + threadIdx.x] += dout[(size_t)a * H + c]; }
    }
    __syncthreads();
    if (c < H)
        for (int r = 0; r < rows; ++r) slabs[((size_t)chunk * rows + r) * H + c] = acc[r * 128 + threadIdx.x];
}
// out[a, r] = 1 if z[a] == r (and r is not the padding row), else 0: the embedding gradient dW = onehot(z)^T dout then is an ordinary weight
// gradient and joins the batched node-level launch of a backward pass (ops._EmbeddingFn; exact: 0 / 1 split into bf16 planes without error)
__global__ void __launch_bounds__(256) k_onehot_rows(const int64_t *__restrict__ z, int n, int rows, int padding_idx, float *__restrict__ out) {
    const long long total = (long long)n * rows, stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int a = (int)(i / rows), r = (int)(i - (long long)a * rows);
        out[i] = (z[a] == (long long)r && r != padding_idx) ? 1.0f : 0.0f;
    }
}

__global__ void k_embedding_bwd_reduce(const float *__restrict__ slabs, int chunks, int rows, int H, int padding_idx,
                                       float *__restrict__ dweight) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * H) return;
    float s = 0.f;
    int ch = 0;
    for (; ch + 8 <= chunks; ch += 8) {                       // 8 loads in flight, summed in chunk order
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = slabs[(size_t)(ch + u) * rows * H + i];
#pragma unroll
        for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; ch < chunks; ++ch) s += slabs[(size_t)ch * rows * H + i];
    dweight[i] = (i / H == padding_idx) ? 0.f : s;
}

__global__ void __launch_bounds__(64) k_segment_sum(const float *__restrict__ x, const int *__restrict__ gptr, int W,
                                                    float *__restrict__ out) {
    const int g = blockIdx.x;
    const int lo = gptr[g], hi = gptr[g + 1];
    for (int c = threadIdx.x; c < W; c += 64) {
        float s = 0.f;
        int a = lo;
        for (; a + 8 <= hi; a += 8) {                            // 8 loads in flight, summed in atom order (a serial loop is one round trip per atom)
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = x[(size_t)(a + u) * W + c];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
        if (a < hi) {                                            // tail: clamped loads, masked adds
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = x[(size_t)min(a + u, hi - 1) * W + c];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += a + u < hi ? v[u] : 0.f;
        }
        out[(size_t)g * W + c] = s;
    }
}

__global__ void __launch_bounds__(64) k_segment_bcast(const float *__restrict__ dout, const int *__restrict__ gptr, int W,
                                                      float *__restrict__ dx) {
    const int g = blockIdx.x;
    const int lo = gptr[g], hi = gptr[g + 1];
    for (int c = threadIdx.x; c < W; c += 64) {
        const float v = dout[(size_t)g * W + c];
        for (int a = lo; a < hi; ++a) dx[(size_t)a * W + c] = v;
    }
}

// ReLU / sigmoid of the classification head (schnet_based_models.py:31-45,367): tiny [B, width] tensors.  Backward from the OUTPUT:
// relu' = (y > 0), sigmoid' = y (1 - y).
template <int OP>      // 0 = relu, 1 = sigmoid
__global__ void k_unary_fwd(const float *__restrict__ x, long long n, float *__restrict__ y) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const float v = x[i];
        y[i] = OP == 0 ? fmaxf(v, 0.f) : OP == 1 ? 1.0f / (1.0f + expf(-v)) : ssp_f(v);
    }
}
template <int OP>
__global__ void k_unary_bwd(const float *__restrict__ y, const float *__restrict__ dy, long long n, float *__restrict__ dx) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const float o = y[i];
        dx[i] = OP == 0 ? (o > 0.f ? dy[i] : 0.f) : OP == 1 ? dy[i] * o * (1.0f - o) : dy[i] * (1.0f - 0.5f * __expf(-o));
    }
}


}  // namespace

extern "C" {

int conan_embedding_fwd(const int64_t *z, const float *weight, int num_atoms, int hidden, int num_embeddings, float *out, void *stream) {
    if (!z || !weight || !out || num_atoms < 0 || hidden <= 0 || (hidden & 3) || num_embeddings <= 0) return CONAN_E_BADARG;
    if (num_atoms == 0) return CONAN_OK;
    long long total = (long long)num_atoms * (hidden >> 2);
    int blocks = (int)((total + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    k_embedding_fwd<<<blocks, 256, 0, as_stream(stream)>>>(z, weight, num_atoms, hidden >> 2, num_embeddings, out);
    CONAN_LAUNCH_CHECK();
    return CONAN_OK;
}

int conan_onehot_rows(const int64_t *z, int num_atoms, int num_embeddings, int padding_idx, float *out, void *stream) {
    if (!z || !out || num_atoms < 0 || num_embeddings <= 0) return CONAN_E_BADARG;
    if (num_atoms == 0) return CONAN_OK;
    const long long total = (long long)num_atoms * num_embeddings;
    const int blocks = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
    k_onehot_rows<<<blocks, 256, 0, as_stream(stream)>>>(z, num_atoms, num_embeddings, padding_idx, out);
    CONAN_LAUNCH_CHECK();
    return CONAN_OK;
}

long long conan_embedding_bwd_ws(int num_atoms, int hidden, int num_embeddings) {
    return (long long)((num_atoms + EMB_CHUNK - 1) / EMB_CHUNK) * num_embeddings * hidden;
}

int conan_embedding_bwd(const int64_t *z, const float *dout, int num_atoms, int hidden, int num_embeddings,
                        int padding_idx, float *dweight, float *ws, void *stream) {
    if (!z || !dout || !dweight || !ws || num_atoms < 0 || hidden <= 0 || num_embeddings <= 0) return CONAN_E_BADARG;
    if (num_embeddings > EMB_ROWS_MAX) return CONAN_E_UNSUPPORTED;
    const int chunks = (num_atoms + EMB_CHUNK - 1) / EMB_CHUNK;
    hipStream_t s = as_stream(stream);
    if (chunks > 0) {
        dim3 grid(chunks, (hidden + 127) / 128);
        k_embedding_bwd_partial<<<grid, 128, 0, s>>>(z, dout, num_atoms, hidden, num_embeddings, ws);
    }
    const int total = num_embeddings * hidden;
    k_embedding_bwd_reduce<<<(total + 255) / 256, 256, 0, s>>>(ws, chunks, num_embeddings, hidden, padding_idx, dweight);
    CONAN_LAUNCH_CHECK();
    return CONAN_OK;
}

int conan_segment_sum_fwd(const float *x, const int *graph_ptr, int num_graphs, int width, float *out, void *stream) {
    if (!x || !graph_ptr || !out || num_graphs < 0 || width <= 0) return CONAN_E_BADARG;
    if (num_graphs == 0) return CONAN_OK;
    k_segment_sum<<<num_graphs, 64, 0, as_stream(stream)>>>(x, graph_ptr, width, out);
    CONAN_LAUNCH_CHECK();
    return CONAN_OK;
}

int conan_segment_sum_bwd(const float *dout, const int *graph_ptr, int num_graphs, int width, float *dx, void *stream) {
    if (!dout || !graph_ptr || !dx || num_graphs < 0 || width <= 0) return CONAN_E_BADARG;
    if (num_graphs == 0) return CONAN_OK;
    k_segment_bcast<<<num_graphs, 64, 0, as_stream(stream)>>>(dout, graph_ptr, width, dx);
    CONAN_LAUNCH_CHECK();
    return CONAN_OK;
}

int conan_unary_fwd(const float *x, long long count, int op, float *y, void *stream) {
    if (count < 0 || op < 0 || op > 2 || (count && (!x || !y))) return CONAN_E_BADARG;
    if (!count) return CONAN_OK;
    const int blocks = (int)((count + 255) / 256 > 2048 ? 2048 : (count + 255) / 256);
    if (op == 0) k_unary_fwd<0><<<blocks, 256, 0, as_stream(stream)>>>(x, count, y);
    else if (op == 1) k_unary_fwd<1><<<blocks, 256, 0, as_stream(stream)>>>(x, count, y);
    else k_unary_fwd<2><<<blocks, 256, 0, as_stream(stream)>>>(x, count, y);
    CONAN_LAUNCH_CHECK();
    return CONAN_OK;
}

int conan_unary_bwd(const float *y, const float *dy, long long count, int op, float *dx, void *stream) {
    if (count < 0 || op < 0 || op > 2 || (count && (!y || !dy || !dx))) return CONAN_E_BADARG;
    if (!count) return CONAN_OK;
    const int blocks = (int)((count + 255) / 256 > 2048 ? 2048 : (count + 255) / 256);
    if (op == 0) k_unary_bwd<0><<<blocks, 256, 0, as_stream(stream)>>>(y, dy, count, dx);
    else if (op == 1) k_unary_bwd<1><<<blocks, 256, 0, as_stream(stream)>>>(y, dy, count, dx);
    else k_unary_bwd<2><<<blocks, 256, 0, as_stream(stream)>>>(y, dy, count, dx);
    CONAN_LAUNCH_CHECK();
    return CONAN_OK;
}

}  // extern "C"
