// CFConv message + aggregation: out[i,:] = sum_{e in CSR row i} x[col[e],:] * W[e,:]   — the HBM-bound kernel of the
// SchNet path (it streams the [E,F] filter tensor once) — and its backward.
//
// Mapping (wave64): one half-wavefront (32 lanes x float4 = 128 channels = one 512-B row) per edge, so a wavefront
// consumes two edges of the same target per step with fully coalesced 16-B/lane loads of the W row and of the gathered
// x row; partial sums stay in registers, the two halves are combined with one cross-lane add, one 512-B store per
// target.  No atomics: the CSR is sorted by target, results are bitwise reproducible.
// Several targets per workgroup keep >= 8 wavefronts per SIMD resident to cover HBM latency.
#include "common.h"

namespace {

template <int VEC>   // VEC = F / 32 / 4  (number of float4 per lane per row): F=128 -> 1, F=256 -> 2
__global__ void __launch_bounds__(256) k_cfconv_fwd(const float *__restrict__ x, const float *__restrict__ W,
                                                    const int *__restrict__ rowptr, const int *__restrict__ col, int num_atoms,
                                                    float *__restrict__ out, const int *__restrict__ pid) {
    constexpr int F = 128 * VEC;
    const int lane = threadIdx.x & 63, half = lane >> 5, l32 = lane & 31;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int nwaves = (gridDim.x * blockDim.x) >> 6;
    for (int i = wave; i < num_atoms; i += nwaves) {
        const int e0 = rowptr[i], e1 = rowptr[i + 1];
        float4 acc[VEC];
#pragma unroll
        for (int v = 0; v < VEC; ++v) acc[v] = make_float4(0.f, 0.f, 0.f, 0.f);
        // 2 edges per step per wave (one per half), unrolled x2 for more loads in flight
        int e = e0 + half;
        for (; e + 2 < e1; e += 4) {
            const int j0 = col[e], j1 = col[e + 2];
            const int r0 = pid ? pid[e] : e, r1 = pid ? pid[e + 2] : e + 2;      // filter row (shared by both directions of a pair)
#pragma unroll
            for (int v = 0; v < VEC; ++v) {
                const float4 w0 = reinterpret_cast<const float4 *>(W + (size_t)r0 * F)[l32 + 32 * v];
                const float4 w1 = reinterpret_cast<const float4 *>(W + (size_t)r1 * F)[l32 + 32 * v];
                const float4 x0 = reinterpret_cast<const float4 *>(x + (size_t)j0 * F)[l32 + 32 * v];
                const float4 x1 = reinterpret_cast<const float4 *>(x + (size_t)j1 * F)[l32 + 32 * v];
                acc[v].x += x0.x * w0.x; acc[v].y += x0.y * w0.y; acc[v].z += x0.z * w0.z; acc[v].w += x0.w * w0.w;
                acc[v].x += x1.x * w1.x; acc[v].y += x1.y * w1.y; acc[v].z += x1.z * w1.z; acc[v].w += x1.w * w1.w;
            }
        }
        for (; e < e1; e += 2) {
            const int j0 = col[e];
            const int r0 = pid ? pid[e] : e;
#pragma unroll
            for (int v = 0; v < VEC; ++v) {
                const float4 w0 = reinterpret_cast<const float4 *>(W + (size_t)r0 * F)[l32 + 32 * v];
                const float4 x0 = reinterpret_cast<const float4 *>(x + (size_t)j0 * F)[l32 + 32 * v];
                acc[v].x += x0.x * w0.x; acc[v].y += x0.y * w0.y; acc[v].z += x0.z * w0.z; acc[v].w += x0.w * w0.w;
            }
        }
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
            acc[v].x += __shfl_xor(acc[v].x, 32, 64); acc[v].y += __shfl_xor(acc[v].y, 32, 64);
            acc[v].z += __shfl_xor(acc[v].z, 32, 64); acc[v].w += __shfl_xor(acc[v].w, 32, 64);
            if (half == 0) reinterpret_cast<float4 *>(out + (size_t)i * F)[l32 + 32 * v] = acc[v];
        }
    }
}

// generic width (F % 4 == 0, any size): one wavefront per target, lanes stride over float4 columns
__global__ void __launch_bounds__(256) k_cfconv_fwd_generic(const float *__restrict__ x, const float *__restrict__ W,
                                                            const int *__restrict__ rowptr, const int *__restrict__ col, int num_atoms,
                                                            int F, float *__restrict__ out, const int *__restrict__ pid) {
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int nwaves = (gridDim.x * blockDim.x) >> 6;
    const int F4 = F >> 2;
    for (int i = wave; i < num_atoms; i += nwaves) {
        const int e0 = rowptr[i], e1 = rowptr[i + 1];
        for (int c = lane; c < F4; c += 64) {
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int e = e0; e < e1; ++e) {
                const float4 w = reinterpret_cast<const float4 *>(W + (size_t)(pid ? pid[e] : e) * F)[c];
                const float4 xv = reinterpret_cast<const float4 *>(x + (size_t)col[e] * F)[c];
                acc.x += xv.x * w.x; acc.y += xv.y * w.y; acc.z += xv.z * w.z; acc.w += xv.w * w.w;
            }
            reinterpret_cast<float4 *>(out + (size_t)i * F)[c] = acc;
        }
    }
}

// dx[j,:] = sum over edges e with source j of W[row(e),:] * dout[tgt[e],:]   (by-source CSR: t_rowptr / t_eid).
// F = 128: same mapping as the forward — one half-wavefront (32 lanes x float4) per edge, two edges per step.
__global__ void __launch_bounds__(256) k_cfconv_bwd_x128(const float *__restrict__ W, const float *__restrict__ dout,
                                                         const int *__restrict__ t_rowptr, const int *__restrict__ t_eid,
                                                         const int *__restrict__ tgt, int num_atoms, float *__restrict__ dx,
                                                         const int *__restrict__ pid) {
    constexpr int F = 128;
    const int lane = threadIdx.x & 63, half = lane >> 5, l32 = lane & 31;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int nwaves = (gridDim.x * blockDim.x) >> 6;
    for (int j = wave; j < num_atoms; j += nwaves) {
        const int s0 = t_rowptr[j], s1 = t_rowptr[j + 1];
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        int s = s0 + half;
        for (; s + 2 < s1; s += 4) {
            const int ea = t_eid[s], eb = t_eid[s + 2];
            const float4 wa = reinterpret_cast<const float4 *>(W + (size_t)(pid ? pid[ea] : ea) * F)[l32];
            const float4 wb = reinterpret_cast<const float4 *>(W + (size_t)(pid ? pid[eb] : eb) * F)[l32];
            const float4 ga = reinterpret_cast<const float4 *>(dout + (size_t)tgt[ea] * F)[l32];
            const float4 gb = reinterpret_cast<const float4 *>(dout + (size_t)tgt[eb] * F)[l32];
            acc.x += ga.x * wa.x; acc.y += ga.y * wa.y; acc.z += ga.z * wa.z; acc.w += ga.w * wa.w;
            acc.x += gb.x * wb.x; acc.y += gb.y * wb.y; acc.z += gb.z * wb.z; acc.w += gb.w * wb.w;
        }
        for (; s < s1; s += 2) {
            const int ea = t_eid[s];
            const float4 wa = reinterpret_cast<const float4 *>(W + (size_t)(pid ? pid[ea] : ea) * F)[l32];
            const float4 ga = reinterpret_cast<const float4 *>(dout + (size_t)tgt[ea] * F)[l32];
            acc.x += ga.x * wa.x; acc.y += ga.y * wa.y; acc.z += ga.z * wa.z; acc.w += ga.w * wa.w;
        }
        acc.x += __shfl_xor(acc.x, 32, 64); acc.y += __shfl_xor(acc.y, 32, 64);
        acc.z += __shfl_xor(acc.z, 32, 64); acc.w += __shfl_xor(acc.w, 32, 64);
        if (half == 0) reinterpret_cast<float4 *>(dx + (size_t)j * F)[l32] = acc;
    }
}

// generic width
__global__ void __launch_bounds__(256) k_cfconv_bwd_x(const float *__restrict__ W, const float *__restrict__ dout,
                                                      const int *__restrict__ t_rowptr, const int *__restrict__ t_eid,
                                                      const int *__restrict__ tgt, int num_atoms, int F, float *__restrict__ dx,
                                                      const int *__restrict__ pid) {
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int nwaves = (gridDim.x * blockDim.x) >> 6;
    const int F4 = F >> 2;
    for (int j = wave; j < num_atoms; j += nwaves) {
        const int s0 = t_rowptr[j], s1 = t_rowptr[j + 1];
        for (int c = lane; c < F4; c += 64) {
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int s = s0; s < s1; ++s) {
                const int e = t_eid[s];
                const float4 w = reinterpret_cast<const float4 *>(W + (size_t)(pid ? pid[e] : e) * F)[c];
                const float4 g = reinterpret_cast<const float4 *>(dout + (size_t)tgt[e] * F)[c];
                acc.x += g.x * w.x; acc.y += g.y * w.y; acc.z += g.z * w.z; acc.w += g.w * w.w;
            }
            reinterpret_cast<float4 *>(dx + (size_t)j * F)[c] = acc;
        }
    }
}

// dW[e,:] = x[col[e],:] * dout[tgt[e],:] * scale(e)      (scale = optional per-edge factor, e.g. the cosine cutoff)
__global__ void __launch_bounds__(256) k_cfconv_bwd_w(const float *__restrict__ x, const float *__restrict__ dout,
                                                      const int *__restrict__ num_edges_dev, int max_edges,
                                                      const int *__restrict__ col, const int *__restrict__ tgt, int F,
                                                      const float *__restrict__ dist, float cutoff, float *__restrict__ dW) {
    const int E = num_edges_dev ? min(*num_edges_dev, max_edges) : max_edges;
    const int F4 = F >> 2;
    const long long n4 = (long long)E * F4;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        const int e = (int)(i / F4), c = (int)(i - (long long)e * F4);
        const float4 xv = reinterpret_cast<const float4 *>(x + (size_t)col[e] * F)[c];
        const float4 g = reinterpret_cast<const float4 *>(dout + (size_t)tgt[e] * F)[c];
        const float cc = dist ? 0.5f * (cosf(__fdiv_rn(dist[e] * 3.14159265358979323846f, cutoff)) + 1.0f) : 1.0f;
        reinterpret_cast<float4 *>(dW)[i] = make_float4(xv.x * g.x * cc, xv.y * g.y * cc, xv.z * g.z * cc, xv.w * g.w * cc);
    }
}

// pair-level filter gradient: dWp[p,:] = C(d_p) * ( x[src(e0)]*dout[tgt(e0)] + x[src(e1)]*dout[tgt(e1)] )   (e1 = -1: one direction only)
__global__ void __launch_bounds__(256) k_cfconv_bwd_wp(const float *__restrict__ x, const float *__restrict__ dout,
                                                       const int *__restrict__ num_pairs_dev, int max_pairs, const int *__restrict__ pe0,
                                                       const int *__restrict__ pe1, const int *__restrict__ col, const int *__restrict__ tgt,
                                                       int F, const float *__restrict__ pdist, float cutoff, float *__restrict__ dWp) {
    const int P = min(*num_pairs_dev, max_pairs);
    const int F4 = F >> 2;
    const long long n4 = (long long)P * F4;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        const int p = (int)(i / F4), c = (int)(i - (long long)p * F4);
        const int e0 = pe0[p], e1 = pe1[p];
        const float4 xa = reinterpret_cast<const float4 *>(x + (size_t)col[e0] * F)[c];
        const float4 ga = reinterpret_cast<const float4 *>(dout + (size_t)tgt[e0] * F)[c];
        float4 r = make_float4(xa.x * ga.x, xa.y * ga.y, xa.z * ga.z, xa.w * ga.w);
        if (e1 >= 0) {
            const float4 xb = reinterpret_cast<const float4 *>(x + (size_t)col[e1] * F)[c];
            const float4 gb = reinterpret_cast<const float4 *>(dout + (size_t)tgt[e1] * F)[c];
            r.x += xb.x * gb.x; r.y += xb.y * gb.y; r.z += xb.z * gb.z; r.w += xb.w * gb.w;
        }
        const float cc = 0.5f * (cosf(__fdiv_rn(pdist[p] * 3.14159265358979323846f, cutoff)) + 1.0f);
        reinterpret_cast<float4 *>(dWp)[i] = make_float4(r.x * cc, r.y * cc, r.z * cc, r.w * cc);
    }
}

}  // namespace

extern "C" {

int conan_cfconv_fwd(const float *x, const float *W, const int *rowptr, const int *col, const int *pid, int num_atoms,
                     int num_filters, float *out, void *stream) {
    if (!x || !W || !rowptr || !col || !out || num_atoms < 0 || num_filters <= 0 || (num_filters & 3)) return CONAN_E_BADARG;
    if (num_atoms == 0) return CONAN_OK;
    hipStream_t s = as_stream(stream);
    int blocks = (num_atoms + 3) / 4;                 // 4 wavefronts (targets) per 256-thread workgroup
    if (blocks > 256 * 16) blocks = 256 * 16;
    if (num_filters == 128) k_cfconv_fwd<1><<<blocks, 256, 0, s>>>(x, W, rowptr, col, num_atoms, out, pid);
    else if (num_filters == 256) k_cfconv_fwd<2><<<blocks, 256, 0, s>>>(x, W, rowptr, col, num_atoms, out, pid);
    else k_cfconv_fwd_generic<<<blocks, 256, 0, s>>>(x, W, rowptr, col, num_atoms, num_filters, out, pid);
    CONAN_LAUNCH_CHECK();
    return CONAN_OK;
}

int conan_cfconv_bwd_x(const float *W, const float *dout, const int *t_rowptr, const int *t_eid, const int *tgt,
                       const int *pid, int num_atoms, int num_filters, float *dx, void *stream) {
    if (!W || !dout || !t_rowptr || !t_eid || !tgt || !dx || num_atoms < 0 || num_filters <= 0 || (num_filters & 3)) return CONAN_E_BADARG;
    if (num_atoms == 0) return CONAN_OK;
    int blocks = (num_atoms + 3) / 4;
    if (blocks > 256 * 16) blocks = 256 * 16;
    if (num_filters == 128) k_cfconv_bwd_x128<<<blocks, 256, 0, as_stream(stream)>>>(W, dout, t_rowptr, t_eid, tgt, num_atoms, dx, pid);
    else k_cfconv_bwd_x<<<blocks, 256, 0, as_stream(stream)>>>(W, dout, t_rowptr, t_eid, tgt, num_atoms, num_filters, dx, pid);
    CONAN_LAUNCH_CHECK();
    return CONAN_OK;
}

int conan_cfconv_bwd_w(const float *x, const float *dout, const int *num_edges_dev, int max_edges, const int *col,
                       const int *tgt, int num_filters, const float *dist, float cutoff, float *dW, void *stream) {
    if (!x || !dout || !col || !tgt || !dW || max_edges < 0 || num_filters <= 0 || (num_filters & 3)) return CONAN_E_BADARG;
    if (max_edges == 0) return CONAN_OK;
    k_cfconv_bwd_w<<<4096, 256, 0, as_stream(stream)>>>(x, dout, num_edges_dev, max_edges, col, tgt, num_filters, dist, cutoff, dW);
    CONAN_LAUNCH_CHECK();
    return CONAN_OK;
}

int conan_cfconv_bwd_w_pairs(const float *x, const float *dout, const int *num_pairs_dev, int max_pairs, const int *pair_e0,
                             const int *pair_e1, const int *col, const int *tgt, int num_filters, const float *pair_dist,
                             float cutoff, float *dWp, void *stream) {
    if (!x || !dout || !num_pairs_dev || !pair_e0 || !pair_e1 || !col || !tgt || !pair_dist || !dWp || max_pairs < 0 || num_filters <= 0 ||
        (num_filters & 3))
        return CONAN_E_BADARG;
    if (max_pairs == 0) return CONAN_OK;
    k_cfconv_bwd_wp<<<4096, 256, 0, as_stream(stream)>>>(x, dout, num_pairs_dev, max_pairs, pair_e0, pair_e1, col, tgt, num_filters, pair_dist,
                                                          cutoff, dWp);
    CONAN_LAUNCH_CHECK();
    return CONAN_OK;
}

}  // extern "C"
