// fp64 issue rates on gfx950, measured with s_memtime around unrolled instruction streams (one workgroup per CU, W waves per SIMD):
//   v_mfma_f64_16x16x4_f64, v_mfma_f64_4x4x4_4b_f64, v_fma_f64, v_add_f64, v_cvt_f64_f32 — cycles per wave-instruction per SIMD —
// and whether an fp64 MFMA stream and an fp64 VALU stream on the SAME SIMD (two waves) overlap or serialise.
// Decides how the large-N FGW coupling kernel treats the ragged border of its 16 x 16 tiles.
// build: hipcc --offload-arch=gfx950 -O2 tools/probes/f64_rates.hip -o gpurun_out/f64_rates ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef double f64x4 __attribute__((ext_vector_type(4)));
constexpr int REP = 256;

template <int MODE>
__global__ void __launch_bounds__(1024) k(long long *out, double *sink, int mixed) {
    const int wave = threadIdx.x >> 6;
    double a = 1.0 + threadIdx.x * 1e-9, b = 0.5, c0 = 0.0, c1 = 0.0, c2 = 0.0, c3 = 0.0;
    float fa = (float)a;
    f64x4 acc0 = {0, 0, 0, 0}, acc1 = acc0, acc2 = acc0, acc3 = acc0;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    int mode = MODE;
    if (mixed && (wave & 4)) mode = 2;                      // waves 4..7 (the second wave of every SIMD) run v_fma_f64 beside the MFMA waves
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime();
    if (mode == 0) {
#pragma unroll 1
        for (int r = 0; r < REP / 4; ++r) {
            acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc1, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc2, 0, 0, 0);
            acc3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc3, 0, 0, 0);
        }
    } else if (mode == 1) {
#pragma unroll 1
        for (int r = 0; r < REP / 4; ++r) {
            s0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, s0, 0, 0, 0);
            s1 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, s1, 0, 0, 0);
            s2 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, s2, 0, 0, 0);
            s3 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, s3, 0, 0, 0);
        }
    } else if (mode == 2) {
#pragma unroll 1
        for (int r = 0; r < REP / 4; ++r) {
            asm volatile("v_fma_f64 %0, %4, %5, %0\n v_fma_f64 %1, %4, %5, %1\n v_fma_f64 %2, %4, %5, %2\n v_fma_f64 %3, %4, %5, %3"
                         : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(a), "v"(b));
        }
    } else if (mode == 3) {
#pragma unroll 1
        for (int r = 0; r < REP / 4; ++r) {
            asm volatile("v_add_f64 %0, %4, %0\n v_add_f64 %1, %4, %1\n v_add_f64 %2, %4, %2\n v_add_f64 %3, %4, %3"
                         : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(a));
        }
    } else if (mode == 4) {
#pragma unroll 1
        for (int r = 0; r < REP / 4; ++r) {
            asm volatile("v_cvt_f64_f32 %0, %4\n v_cvt_f64_f32 %1, %4\n v_cvt_f64_f32 %2, %4\n v_cvt_f64_f32 %3, %4"
                         : "=v"(c0), "=v"(c1), "=v"(c2), "=v"(c3) : "v"(fa));
        }
    }
    asm volatile("s_nop 0" ::: "memory");
    const long long t1 = __builtin_amdgcn_s_memtime();
    // keep everything alive
    double keep = c0 + c1 + c2 + c3 + s0 + s1 + s2 + s3 + acc0[0] + acc1[1] + acc2[2] + acc3[3];
    if (keep == 12345.678) sink[threadIdx.x] = keep;
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + wave] = t1 - t0;
}

template <int MODE>
static void run(const char *name, int waves_per_simd, int mixed = 0) {
    long long *d; double *sink;
    const int blocks = 256;
    hipMalloc(&d, blocks * 16 * sizeof(long long)); hipMalloc(&sink, 1024 * sizeof(double));
    hipMemset(d, 0, blocks * 16 * sizeof(long long));
    const int threads = 256 * waves_per_simd;
    for (int it = 0; it < 3; ++it) hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, d, sink, mixed);
    hipDeviceSynchronize();
    std::vector<long long> h(blocks * 16);
    hipMemcpy(h.data(), d, h.size() * sizeof(long long), hipMemcpyDeviceToHost);
    std::vector<double> first, second;
    for (int b = 0; b < blocks; ++b)
        for (int w = 0; w < 4 * waves_per_simd; ++w) (mixed && (w & 4) ? second : first).push_back((double)h[b * 16 + w] / REP);
    auto med = [](std::vector<double> &v) { std::sort(v.begin(), v.end()); return v.empty() ? 0.0 : v[v.size() / 2]; };
    if (mixed) printf("%-46s waves/SIMD %d : MFMA wave %.1f cyc/instr, v_fma_f64 wave beside it %.1f cyc/instr\n", name, waves_per_simd, med(first), med(second));
    else printf("%-46s waves/SIMD %d : %.1f cyc per wave-instruction (one wave's own stream)\n", name, waves_per_simd, med(first));
    hipFree(d); hipFree(sink);
}

int main() {
    for (int w : {1, 2, 4}) {
        run<0>("v_mfma_f64_16x16x4_f64 (1024 FMA)", w);
        run<1>("v_mfma_f64_4x4x4_4b_f64 (256 FMA)", w);
        run<2>("v_fma_f64 (64 FMA)", w);
        run<3>("v_add_f64", w);
        run<4>("v_cvt_f64_f32", w);
    }
    run<0>("16x16x4 MFMA waves 0-3 | v_fma_f64 waves 4-7", 2, 1);
    run<1>("4x4x4 MFMA waves 0-3 | v_fma_f64 waves 4-7", 2, 1);
    return 0;
}
