// Lane layout of v_mfma_f64_4x4x4_4b_f64 on gfx950 by brute force: A = one-hot at lane la, B = one-hot at lane lb, C = 0; which D lanes
// receive the product?  Prints, for every D lane, the (A lane, B lane) pairs that feed it, and checks the closed form the large-N FGW
// kernel's border strips assume:  block = lane / 16;  A: row i = lane % 4, k = (lane / 4) % 4;  B: column j = lane % 4, k = (lane / 4) % 4;
// D: row i = (lane / 4) % 4 ... (the probe prints what the hardware does; the closed form is whatever it shows).
// build: hipcc --offload-arch=gfx950 -O2 tools/probes/mfma_f64_4x4_probe.hip -o /tmp/mfma44 ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned char *out) {          // out[la][lb][ld] = 1 if D[ld] != 0
    const int lane = threadIdx.x;
    for (int la = 0; la < 64; ++la)
        for (int lb = 0; lb < 64; ++lb) {
            const double a = lane == la ? 1.0 : 0.0, b = lane == lb ? 1.0 : 0.0;
            const double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
            out[(la * 64 + lb) * 64 + lane] = d != 0.0 ? 1 : 0;
        }
}
int main() {
    unsigned char *d; static unsigned char h[64 * 64 * 64];
    hipMalloc(&d, sizeof(h));
    k<<<1, 64>>>(d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int ld = 0; ld < 64; ++ld) {
        printf("D lane %2d <-", ld);
        int n = 0;
        for (int la = 0; la < 64; ++la)
            for (int lb = 0; lb < 64; ++lb)
                if (h[(la * 64 + lb) * 64 + ld]) { printf(" (A%d,B%d)", la, lb); ++n; }
        printf("\n");
        // hypothesis: block blk = ld / 16; D row i = ld % 4 ... test both orientations below
        (void)n;
    }
    // closed-form check, hypothesis H1: D lane ld = 16 blk + 4 j + i  <-  sum_k A[16 blk + 4 k + i] * B[16 blk + 4 k + j]
    for (int hyp = 0; hyp < 2; ++hyp) {
        bad = 0;
        for (int ld = 0; ld < 64; ++ld)
            for (int la = 0; la < 64; ++la)
                for (int lb = 0; lb < 64; ++lb) {
                    const int blk = ld / 16, x = ld % 4, y = (ld / 4) % 4;
                    const int i = hyp == 0 ? x : y, j = hyp == 0 ? y : x;
                    const bool want = la / 16 == blk && lb / 16 == blk && (la / 4) % 4 == (lb / 4) % 4 && la % 4 == i && lb % 4 == j;
                    if (want != (h[(la * 64 + lb) * 64 + ld] != 0)) ++bad;
                }
        printf("hypothesis %d (A lane = 16 blk + 4 k + i, B lane = 16 blk + 4 k + j, D lane = 16 blk + %s): %d mismatches\n", hyp,
               hyp == 0 ? "4 j + i" : "4 i + j", bad);
    }
    return 0;
}
