// Probe of ds_read_b64_tr_b16 (gfx950): which LDS element lands in which lane / vector slot.  Image: 32 rows x 128 columns of 16-bit
// elements, row pitch 136; element (r, c) holds the value 128 r + c.  Every 16-lane group reads the 4 x 16 block (r0, c0) the host asks for.
// build: hipcc --offload-arch=gfx950 -O2 tools/probes/tr_read_probe.hip -o /tmp/tr_probe ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s16x4 __attribute__((__vector_size__(4 * sizeof(short))));
__global__ void k(short* out) {
    __shared__ __attribute__((aligned(16))) short img[32 * 136];
    for (int i = threadIdx.x; i < 32 * 128; i += 64) img[(i / 128) * 136 + (i % 128)] = (short)i;
    __syncthreads();
    const int lane = threadIdx.x, g = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
    const int r0 = 4 * g, c0 = 16 * g;                    // group g: rows 4g..4g+3, columns 16g..16g+15
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)&img[(r0 + q) * 136 + c0 + 4 * p]);
    for (int j = 0; j < 4; ++j) out[lane * 4 + j] = v[j];
}
int main() {
    short* d; short h[256];
    hipMalloc(&d, sizeof(h));
    k<<<1, 64>>>(d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int lane = 0; lane < 64; ++lane) {
        const int g = lane >> 4, i = lane & 15;
        printf("lane %2d:", lane);
        for (int j = 0; j < 4; ++j) {
            const int r = h[lane * 4 + j] / 128, c = h[lane * 4 + j] % 128;
            printf(" (%d,%d)", r, c);
            if (r != 4 * g + j || c != 16 * g + i) ++bad;      // expected: element j = row r0 + j of column c0 + i
        }
        printf("\n");
    }
    printf("mismatches against 'lane i gets column c0 + i, element j = row r0 + j': %d\n", bad);
    return 0;
}
