// Shared device helpers of the FGW barycenter kernels (fgw.hip: generic path, fgw_small.hip: register-resident path).
#pragma once
#include "common.h"

constexpr int FGW_THREADS = 256;
constexpr int FGW_WAVES = FGW_THREADS / 64;

struct FgwDims {
    int B, K, N, d, P;      // P = row pitch of the LDS/scratch matrices (odd => conflict-free column access)
};

__device__ __forceinline__ double exp_acc(double x) {
    // exp(x) = 2^(x*log2e); integer part applied with ldexp, fractional part on the fp32 transcendental unit.
    if (x < -745.0) return 0.0;
    const double t = x * 1.4426950408889634074;
    const double n = rint(t);
    const float f = (float)(t - n);
    const float e = __builtin_amdgcn_exp2f(f);
    return ldexp((double)e, (int)n);
}

// log(s) for s in [1, 2^20]: v_log_f32 seed (1 ulp of fp32) refined by one Newton step y <- y + (s*exp(-y) - 1) - r^2/2,
// which squares the relative error: ~1e-14.  Replaces the ~100-instruction ocml fp64 log in the Sinkhorn loops.
__device__ __forceinline__ double log_acc(double s) {
    const double y0 = (double)(__builtin_amdgcn_logf((float)s) * 0.693147180559945309f);
    const double r = s * exp_acc(-y0) - 1.0;
    return y0 + (r - 0.5 * r * r);
}

__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_max_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { double w = __shfl_xor(v, o, 64); v = w > v ? w : v; }
    return v;
}

// block-wide sum of one double per thread; result broadcast to every thread. red[] has FGW_WAVES+1 doubles.
__device__ __forceinline__ double block_sum_d(double v, double *red) {
    v = wave_sum_d(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    double s = 0.0;
#pragma unroll
    for (int w = 0; w < FGW_WAVES; ++w) s += red[w];
    return s;
}


// Launchers of the register-resident path (fgw_small.hip), N <= 64.
bool conan_fgw_small_supported(int N, int d);
size_t conan_fgw_small_part_bytes(int B, int K, int N, int d);
void conan_fgw_small_coupling(const float *Ys, const float *Cs, const float *ps, const float *pb, FgwDims D,
                              conan_fgw_params prm, int outer, int y_zero, const double *Cw, const double *Yw,
                              const int *active, float *Tw, int *info, double *Ypart, double *Cpart, hipStream_t s);
void conan_fgw_small_update(const float *pb, const float *lambdas, FgwDims D, conan_fgw_params prm, int outer,
                            const double *Ypart, const double *Cpart, double *Cw, double *Yw, int *active, int *info,
                            float *errs, float *Yout, float *Cout, hipStream_t s);
