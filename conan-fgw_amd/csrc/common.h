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

__device__ __forceinline__ float ssp_f(float v) {
    // shifted softplus, torch semantics: softplus(v) (threshold 20) - ln 2
    float sp = v > 20.0f ? v : log1pf(expf(v));
    return sp - 0.693147180559945309f;
}
