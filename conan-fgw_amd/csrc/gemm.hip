// fp32 MFMA linear layers for gfx950:  y = act(x @ W^T + b) (+ residual), the weight-gradient reduction, and the
// generic (any-shape) continuous-filter generator built from them.
//
// v_mfma_f32_32x32x2_f32 is exact fp32 (bitwise a k-ordered fmaf chain), so the result equals a plain fp32 GEMM up to
// summation order: no TF32-style truncation (the reference's CPU path is true fp32).
//
// Tiling: 256 threads = 4 wavefronts as 2(M) x 2(N); block tile 64 x (64*NB), K step 32.  Operands are staged in LDS
// with odd pitches (33 / BN+1 floats) so that both the transposed staging writes and the MFMA fragment reads
// (lane -> row for A, lane -> column for B) are bank-conflict free with ds_read_b32.
#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BM = 64;
constexpr int BK = 32;
constexpr int XP = BK + 1;   // pitch of the x tile  [BM][XP]

template <int NB>            // NB = 32-column blocks per wave (1 or 2); block tile N = 64*NB
__global__ void __launch_bounds__(256) k_linear(const float *__restrict__ x, const float *__restrict__ w,
                                                const float *__restrict__ bias, const float *__restrict__ residual,
                                                int M, int K, int N, int w_kn, int act, float *__restrict__ y,
                                                const int *__restrict__ m_dev) {
    constexpr int BN = 64 * NB;
    if (m_dev) M = min(M, *m_dev);                 // row count known only on the device (edge-level calls)
    if ((int)blockIdx.x * BM >= M) return;
    constexpr int WP = BN + 1;
    __shared__ float xs[BM * XP];
    __shared__ float ws[BK * WP];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int row0 = blockIdx.x * BM, col0 = blockIdx.y * BN;
    f32x16 acc[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[b][r] = 0.f;

    for (int k0 = 0; k0 < K; k0 += BK) {
        // stage x tile: 64 rows x 32 k, coalesced along k
        for (int t = tid; t < BM * BK; t += 256) {
            int r = t / BK, k = t % BK;
            int gr = row0 + r, gk = k0 + k;
            xs[r * XP + k] = (gr < M && gk < K) ? x[(size_t)gr * K + gk] : 0.f;
        }
        // stage W^T tile: ws[k][n]
        if (w_kn) {
            for (int t = tid; t < BK * BN; t += 256) {
                int k = t / BN, n = t % BN;
                int gk = k0 + k, gn = col0 + n;
                ws[k * WP + n] = (gk < K && gn < N) ? w[(size_t)gk * N + gn] : 0.f;
            }
        } else {
            for (int t = tid; t < BK * BN; t += 256) {
                int n = t / BK, k = t % BK;
                int gk = k0 + k, gn = col0 + n;
                ws[k * WP + n] = (gk < K && gn < N) ? w[(size_t)gn * K + gk] : 0.f;
            }
        }
        __syncthreads();
        const int arow = wm * 32 + (lane & 31), kh = lane >> 5;
#pragma unroll
        for (int kk = 0; kk < BK; kk += 2) {
            float a = xs[arow * XP + kk + kh];
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                float bv = ws[(kk + kh) * WP + wn * 32 * NB + b * 32 + (lane & 31)];
                acc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bv, acc[b], 0, 0, 0);
            }
        }
        __syncthreads();
    }
    // epilogue: D layout col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const int gc = col0 + wn * 32 * NB + b * 32 + (lane & 31);
        if (gc >= N) continue;
        const float bv = bias ? bias[gc] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int gr = row0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            if (gr >= M) continue;
            float v = acc[b][r] + bv;
            if (act == 1) v = ssp_f(v);
            if (residual) v += residual[(size_t)gr * N + gc];
            y[(size_t)gr * N + gc] = v;
        }
    }
}

__global__ void k_ssp_bwd(const float *__restrict__ dy, const float *__restrict__ y, long long n, int width,
                          const int *__restrict__ m_dev, float *__restrict__ g) {
    if (m_dev) n = min(n, (long long)(*m_dev) * width);
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (; i < n; i += stride) g[i] = dy[i] * (1.0f - 0.5f * expf(-y[i]));   // ssp'(v) = sigmoid(v) = 1 - exp(-(y + ln2))
}

// ---- weight gradient: dW[N,K] = g^T @ x, reduction over the M rows ---------------------------------------------
// Stage 1: each workgroup owns a contiguous slice of rows and one 64x64 tile of dW, accumulates it on MFMA
// (A = g^T: lane -> n, B = x: lane -> k, both read row-wise from the staged tiles) and writes a partial slab.
// Stage 2: slabs are summed in a fixed order => bitwise reproducible (no float atomics).
constexpr int WG_ROWS = 64;    // rows staged per step
constexpr int WG_SLICES_MAX = 64;

__global__ void __launch_bounds__(256) k_wgrad_partial(const float *__restrict__ g, const float *__restrict__ x, int M, int K, int N,
                                                       int rows_per_slice, float *__restrict__ slabs, float *__restrict__ bias_slabs,
                                                       const int *__restrict__ m_dev) {
    if (m_dev) M = min(M, *m_dev);
    __shared__ float gs[WG_ROWS * 65];
    __shared__ float xs[WG_ROWS * 65];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;          // 2x2 waves over the 64(n) x 64(k) tile
    const int n0 = blockIdx.y * 64, k0 = blockIdx.z * 64;
    const int slice = blockIdx.x;
    const int r_begin = slice * rows_per_slice, r_end = min(M, r_begin + rows_per_slice);
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    float bsum = 0.f;       // thread t < 64 accumulates the bias gradient of column n0 + t (only for blockIdx.z == 0)
    for (int r0 = r_begin; r0 < r_end; r0 += WG_ROWS) {
        for (int t = tid; t < WG_ROWS * 64; t += 256) {
            int r = t >> 6, c = t & 63;
            int gr = r0 + r;
            bool ok = gr < r_end;
            gs[r * 65 + c] = (ok && n0 + c < N) ? g[(size_t)gr * N + n0 + c] : 0.f;
            xs[r * 65 + c] = (ok && k0 + c < K) ? x[(size_t)gr * K + k0 + c] : 0.f;
        }
        __syncthreads();
        if (blockIdx.z == 0 && tid < 64) {
            float s = 0.f;
            for (int r = 0; r < WG_ROWS; ++r) s += gs[r * 65 + tid];
            bsum += s;
        }
        const int kh = lane >> 5;
#pragma unroll 8
        for (int m = 0; m < WG_ROWS; m += 2) {
            float a = gs[(m + kh) * 65 + wm * 32 + (lane & 31)];     // A[n][m] = g[m][n]
            float b = xs[(m + kh) * 65 + wn * 32 + (lane & 31)];     // B[m][k] = x[m][k]
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        }
        __syncthreads();
    }
    float *slab = slabs + (size_t)slice * N * K;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        int n = n0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        int k = k0 + wn * 32 + (lane & 31);
        if (n < N && k < K) slab[(size_t)n * K + k] = acc[r];
    }
    if (blockIdx.z == 0 && tid < 64 && n0 + tid < N) bias_slabs[(size_t)slice * N + n0 + tid] = bsum;
}

__global__ void k_wgrad_reduce(const float *__restrict__ slabs, const float *__restrict__ bias_slabs, int slices, int NK, int N,
                               float *__restrict__ dW, float *__restrict__ dbias) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < NK) {
        float s = 0.f;
        for (int sl = 0; sl < slices; ++sl) s += slabs[(size_t)sl * NK + i];
        dW[i] = s;
    }
    if (dbias && i < N) {
        float s = 0.f;
        for (int sl = 0; sl < slices; ++sl) s += bias_slabs[(size_t)sl * N + i];
        dbias[i] = s;
    }
}

static int wgrad_slices(int M) {
    int s = (M + 4095) / 4096;
    if (s < 1) s = 1;
    if (s > WG_SLICES_MAX) s = WG_SLICES_MAX;
    return s;
}

}  // namespace

extern "C" {

int conan_linear_fwd(const float *x, const float *w, const float *bias, const float *residual, int M, int K, int N,
                     int w_kn, int act, const int *m_dev, float *y, void *stream) {
    if (!x || !w || !y || M < 0 || K <= 0 || N <= 0 || act < 0 || act > 1) return CONAN_E_BADARG;
    if (M == 0) return CONAN_OK;
    hipStream_t s = as_stream(stream);
    if (N > 64) {
        dim3 grid((M + BM - 1) / BM, (N + 127) / 128);
        k_linear<2><<<grid, 256, 0, s>>>(x, w, bias, residual, M, K, N, w_kn, act, y, m_dev);
    } else {
        dim3 grid((M + BM - 1) / BM, (N + 63) / 64);
        k_linear<1><<<grid, 256, 0, s>>>(x, w, bias, residual, M, K, N, w_kn, act, y, m_dev);
    }
    CONAN_LAUNCH_CHECK();
    return CONAN_OK;
}

int conan_ssp_bwd(const float *dy, const float *y, int rows, int width, const int *m_dev, float *g, void *stream) {
    const long long count = (long long)rows * width;
    if (rows < 0 || width <= 0 || (count && (!dy || !y || !g))) return CONAN_E_BADARG;
    if (!count) return CONAN_OK;
    int blocks = (int)((count + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    k_ssp_bwd<<<blocks, 256, 0, as_stream(stream)>>>(dy, y, count, width, m_dev, g);
    CONAN_LAUNCH_CHECK();
    return CONAN_OK;
}

long long conan_linear_wgrad_ws(int M, int K, int N) {
    return (long long)wgrad_slices(M) * ((long long)N * K + N);
}

int conan_linear_wgrad(const float *g, const float *x, int M, int K, int N, const int *m_dev, float *dW, float *dbias,
                       float *ws, void *stream) {
    if (!g || !x || !dW || !ws || M < 0 || K <= 0 || N <= 0) return CONAN_E_BADARG;
    hipStream_t s = as_stream(stream);
    const int slices = wgrad_slices(M);
    int rows = (M + slices - 1) / slices;
    rows = ((rows + WG_ROWS - 1) / WG_ROWS) * WG_ROWS;
    float *slabs = ws, *bias_slabs = ws + (size_t)slices * N * K;
    dim3 grid(slices, (N + 63) / 64, (K + 63) / 64);
    k_wgrad_partial<<<grid, 256, 0, s>>>(g, x, M, K, N, rows, slabs, bias_slabs, m_dev);
    const int NK = N * K;
    k_wgrad_reduce<<<(NK + 255) / 256, 256, 0, s>>>(slabs, bias_slabs, slices, NK, N, dW, dbias);
    CONAN_LAUNCH_CHECK();
    return CONAN_OK;
}

}  // extern "C"

