// Node-level glue kernels: atom-type embedding lookup and the per-conformer sum readout, with their backward.
#include "common.h"

namespace {

__global__ void k_embedding_fwd(const int64_t *__restrict__ z, const float *__restrict__ weight, int n, int H4,
                                float *__restrict__ out) {
    const long long total = (long long)n * H4;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        int a = (int)(i / H4), c = (int)(i - (long long)a * H4);
        reinterpret_cast<float4 *>(out)[i] = reinterpret_cast<const float4 *>(weight)[(size_t)z[a] * H4 + c];
    }
}

// Deterministic embedding gradient: one workgroup per embedding row r, fixed-order sum over the atoms with z == r.
// (num_embeddings = 100 rows, a handful of them populated; the scan over z is L2-resident.)
__global__ void __launch_bounds__(256) k_embedding_bwd(const int64_t *__restrict__ z, const float *__restrict__ dout, int n, int H,
                                                       int padding_idx, float *__restrict__ dweight) {
    const int r = blockIdx.x;
    if (r == padding_idx) return;
    for (int c = threadIdx.x; c < H; c += blockDim.x) {
        float s = 0.f;
        for (int a = 0; a < n; ++a)
            if (z[a] == r) s += dout[(size_t)a * H + c];
        dweight[(size_t)r * H + c] = s;
    }
}

__global__ void __launch_bounds__(64) k_segment_sum(const float *__restrict__ x, const int *__restrict__ gptr, int W,
                                                    float *__restrict__ out) {
    const int g = blockIdx.x;
    const int lo = gptr[g], hi = gptr[g + 1];
    for (int c = threadIdx.x; c < W; c += 64) {
        float s = 0.f;
        for (int a = lo; a < hi; ++a) s += x[(size_t)a * W + c];
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

}  // namespace

extern "C" {

int conan_embedding_fwd(const int64_t *z, const float *weight, int num_atoms, int hidden, float *out, void *stream) {
    if (!z || !weight || !out || num_atoms < 0 || hidden <= 0 || (hidden & 3)) return CONAN_E_BADARG;
    if (num_atoms == 0) return CONAN_OK;
    long long total = (long long)num_atoms * (hidden >> 2);
    int blocks = (int)((total + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    k_embedding_fwd<<<blocks, 256, 0, as_stream(stream)>>>(z, weight, num_atoms, hidden >> 2, out);
    CONAN_LAUNCH_CHECK();
    return CONAN_OK;
}

int conan_embedding_bwd(const int64_t *z, const float *dout, int num_atoms, int hidden, int num_embeddings,
                        int padding_idx, float *dweight, void *stream) {
    if (!z || !dout || !dweight || num_atoms < 0 || hidden <= 0 || num_embeddings <= 0) return CONAN_E_BADARG;
    k_embedding_bwd<<<num_embeddings, 256, 0, as_stream(stream)>>>(z, dout, num_atoms, hidden, padding_idx, dweight);
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

}  // extern "C"
