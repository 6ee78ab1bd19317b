// Edge-level elementwise kernels of the SchNet continuous-filter generator (generic, any-shape path):
// Gaussian radial basis expansion and the cosine-cutoff scaling.  Both are HBM-streaming kernels; the edge count lives
// on the device (rowptr[num_atoms]) so launches are grid-stride over a fixed grid and need no host sync.
#include "common.h"

namespace {

// rbf[e,k] = exp(coeff * (dist[e] - offset[k])^2)      (PyG GaussianSmearing.forward)
__global__ void k_rbf(const float *__restrict__ dist, const int *__restrict__ num_edges_dev, int max_edges,
                      const float *__restrict__ offset, int Gs, float coeff, float *__restrict__ rbf) {
    const int E = num_edges_dev ? min(*num_edges_dev, max_edges) : max_edges;
    const long long n = (long long)E * Gs;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        int e = (int)(i / Gs), k = (int)(i - (long long)e * Gs);
        float t = dist[e] - offset[k];
        rbf[i] = expf(coeff * (t * t));
    }
}

// out[e,:] = in[e,:] * C(dist[e]),  C(d) = 0.5*(cos(d*pi/cutoff)+1)       (PyG CFConv.forward)
__global__ void k_cutoff_scale(const float *__restrict__ dist, const int *__restrict__ num_edges_dev, int max_edges, int F,
                               float cutoff, const float *__restrict__ in, float *__restrict__ out) {
    const int E = num_edges_dev ? min(*num_edges_dev, max_edges) : max_edges;
    const int F4 = F >> 2;
    const long long n4 = (long long)E * F4;
    const long long stride = (long long)gridDim.x * blockDim.x;
    const float4 *in4 = reinterpret_cast<const float4 *>(in);
    float4 *out4 = reinterpret_cast<float4 *>(out);
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        int e = (int)(i / F4);
        float c = 0.5f * (cosf(__fdiv_rn(dist[e] * 3.14159265358979323846f, cutoff)) + 1.0f);
        float4 v = in4[i];
        v.x *= c; v.y *= c; v.z *= c; v.w *= c;
        out4[i] = v;
    }
}

// zero the rows [*m_dev, rows) of a [rows, width] buffer: edge-level buffers are sized for the worst case (cap * atoms) and
// only the tail beyond the device-side edge count has to be defined, not the whole buffer
__global__ void k_zero_tail(float *__restrict__ buf, const int *__restrict__ m_dev, int rows, int width) {
    const long long lo = (long long)min(*m_dev, rows) * width, hi = (long long)rows * width;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = lo + (long long)blockIdx.x * blockDim.x + threadIdx.x; i < hi; i += stride) buf[i] = 0.f;
}


}  // namespace

extern "C" {

int conan_rbf_fwd(const float *dist, const int *num_edges_dev, int max_edges, const float *offset, int num_gaussians,
                  float coeff, float *rbf, void *stream) {
    if (!dist || !offset || !rbf || max_edges < 0 || num_gaussians <= 0) return CONAN_E_BADARG;
    if (max_edges == 0) return CONAN_OK;
    k_rbf<<<2048, 256, 0, as_stream(stream)>>>(dist, num_edges_dev, max_edges, offset, num_gaussians, coeff, rbf);
    CONAN_LAUNCH_CHECK();
    return CONAN_OK;
}

int conan_cutoff_scale(const float *dist, const int *num_edges_dev, int max_edges, int width, float cutoff,
                       const float *in, float *out, void *stream) {
    if (!dist || !in || !out || max_edges < 0 || width <= 0 || (width & 3)) return CONAN_E_BADARG;
    if (max_edges == 0) return CONAN_OK;
    k_cutoff_scale<<<2048, 256, 0, as_stream(stream)>>>(dist, num_edges_dev, max_edges, width, cutoff, in, out);
    CONAN_LAUNCH_CHECK();
    return CONAN_OK;
}

int conan_zero_tail(float *buf, const int *m_dev, int rows, int width, void *stream) {
    if (!buf || !m_dev || rows < 0 || width <= 0) return CONAN_E_BADARG;
    if (rows == 0) return CONAN_OK;
    k_zero_tail<<<512, 256, 0, as_stream(stream)>>>(buf, m_dev, rows, width);
    CONAN_LAUNCH_CHECK();
    return CONAN_OK;
}

}  // extern "C"
