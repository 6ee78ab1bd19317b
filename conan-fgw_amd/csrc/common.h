// Shared device/host helpers for the gfx950 kernels of libconan_fgw_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/conan_fgw_hip.h"

#define CONAN_WAVE 64

#define CONAN_LAUNCH_CHECK()                                     \
    do {                                                         \
        hipError_t e_ = hipGetLastError();                       \
        if (e_ != hipSuccess) return CONAN_E_LAUNCH;             \
    } while (0)

static inline hipStream_t as_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }

// ---- wavefront (64-lane) reductions -------------------------------------------------------------
template <typename T>
__device__ __forceinline__ T wave_sum(T v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
template <typename T>
__device__ __forceinline__ T wave_max(T v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { T w = __shfl_xor(v, o, 64); v = w > v ? w : v; }
    return v;
}
template <typename T>
__device__ __forceinline__ T wave_min(T v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { T w = __shfl_xor(v, o, 64); v = w < v ? w : v; }
    return v;
}

// Shifted softplus, softplus(v) - ln 2, on the hardware transcendentals (v_exp_f32 / v_log_f32, 1 ulp each):
// softplus(v) = max(v,0) + log1p(exp(-|v|)).  Absolute error <= ~1.2e-7 over the whole range (the fp32 reference itself
// rounds softplus(v) and the subtraction to 6e-8 each); identical to torch's threshold-20 branch for v > 20.
__device__ __forceinline__ float ssp_f(float v) {
    const float t = __builtin_amdgcn_exp2f(-fabsf(v) * 1.44269504088896340736f);
    const float l = __builtin_amdgcn_logf(1.0f + t) * 0.693147180559945309f;
    return (fmaxf(v, 0.0f) + l) - 0.693147180559945309f;
}

// exp(x) for x <= 0 with |x| up to ~1e3 on v_exp_f32: absolute error <= ~4e-8 (|x| e^x <= 0.37 scales the argument rounding).
// Wave-private LDS hand-off (a lane stores, ANOTHER lane of the same wavefront loads): one wavefront's LDS operations execute in order in
// hardware, but nothing told the compiler not to move the loads above the stores (or the next round's stores above these loads).  A
// wavefront-scope release / acquire pair around a wave barrier does; it costs no instruction beyond the s_waitcnt the data dependence needs.
__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ __forceinline__ float exp_neg_f(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896340736f); }

// Workgroups are dispatched round-robin over the 8 XCDs (blockIdx % 8) and every XCD has a private 4 MB L2.  The gather
// kernels remap the block index so that XCD k works on the k-th contiguous eighth of the index space: rows that are
// re-read by neighbouring items (a conformer's atoms, the two directions of a pair) then meet in ONE L2 instead of being
// fetched through the fabric by up to 8.  gridDim.x must be a multiple of 8.
__device__ __forceinline__ int xcd_contiguous_block(int b, int nb) { return (b & 7) * (nb >> 3) + (b >> 3); }
static inline int round_up8(int v) { return (v + 7) & ~7; }

