// Backward of the WHOLE filter network  W = mlp2(ssp(mlp0(rbf(d))))  in ONE pass over the pair rows (round 4):
//
//     dw2 = g^T h1,  db2 = colsum(g)                       (weight / bias gradient of mlp.2; was k_wgrad_lds_h16, gemm.hip)
//     dh1 = (g @ w2) * ssp'(h1)                             (never leaves the chip)
//     dw1 = dh1^T rbf(d),  db1 = colsum(dh1)                (weight / bias gradient of mlp.0; was k_filter_bwd, filter_bwd.hip)
//
// (schnet_no_sum.py:161-164,209-212 with PyG's CFConv: the filter network of an interaction block.)  The two kernels this replaces
// each streamed g and h1 — 2 x 132 MB per interaction at cfg2 — and both sat between their memory floor and their matrix work with one
// wavefront per SIMD (k_filter_bwd) or a deep LDS staging pipeline (k_wgrad_lds_h16): 88 + 70 us.  Here g and h1 cross HBM once.
//
// Mapping.  A workgroup of EIGHT wavefronts (two per SIMD) owns a tile of 32 pair rows at a time:
//   * staging (all 512 threads): a thread loads 8 consecutive channels of one row of g and of h1 (two float4 each), splits them into two
//     fp16 planes (g scaled by the power of two that puts max |g| into [16, 32), as in filter_bwd.hip) and stores 16-byte chunks into the
//     tile images G and H in LDS: [plane][32 rows][128 x fp16], 256-byte rows, chunk index XOR-swizzled so that BOTH the row reads
//     (ds_read_b128) and the transposed reads (ds_read_b64_tr_b16) below are bank-conflict free.  Each value is split ONCE.  The column
//     sums of g (db2) ride along in eight registers per thread.
//   * wavefronts 0-3 ("A", channel block kb = wave): dx strip D[e][k] = sum_n g[e][n] w2[n][32kb + k'] — A fragments = row reads of G,
//     B fragments = the fp16 planes of w2 (LDS, staged once per workgroup) — then dh1 = D * ssp'(h1) with h1 fetched in the accumulator's
//     own layout by transposed reads of H, then dw1[32kb.., :] += dh1^T rbf with the rbf fragments of the tile read from LDS.
//   * wavefronts 4-7 ("B", channel block kb = wave - 4): dw2[:, 32kb..] += g^T h1 — both operands by transposed reads of the SAME images (the
//     contraction runs over the tile's rows; the two operands enumerate them identically by construction).  They also generate the
//     tile's rbf fragments (Gaussians of the 32 distances, column 63 = 1 for db1) once for the four A wavefronts: their matrix share is
//     24 MFMAs per tile against 36.
// Double-buffered images, one barrier per tile, the next tile's rows requested a full tile ahead.  Every strip of the two gradients is
// owned by exactly one wavefront: the workgroup's slabs are written straight from the accumulators and reduced over the workgroups by the
// batched reducer of gemm.hip (fixed order: bitwise reproducible, no float atomics).
//
// Arithmetic: identical to the two kernels it replaces (two fp16 planes per operand, three partial products per fp32 product, fp32
// accumulation; g scaled from its device-side maximum, w2 planes carrying their own power-of-two scale) — the same tests, same tolerances.
#include "common.h"

// Phase timing (tools/f2_phase_profile.py builds a private library with -DCONAN_F2_PROFILE; the product library contains none of this): lane 0
// of every wavefront accumulates shader-clock ticks between marks and adds them to 16 global slots at the end (slots 0-7: A wavefronts, 8-15: B).
#ifdef CONAN_F2_PROFILE
static __device__ long long g_f2_prof[16];
extern "C" int conan_debug_f2_prof(long long *out, int reset) {
    if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(g_f2_prof), sizeof(long long) * 16) != hipSuccess) return -2;
    if (reset) { long long z[16] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_f2_prof), z, sizeof(z)) != hipSuccess) return -2; }
    return 0;
}
#define F2_PROF_DECL long long prof_t = clock64(); long long prof_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define F2_PROF(k) do { const long long t_ = clock64(); prof_acc[k] += t_ - prof_t; prof_t = t_; } while (0)
#define F2_PROF_FLUSH do { if ((threadIdx.x & 63) == 0) for (int k_ = 0; k_ < 8; ++k_) atomicAdd(reinterpret_cast<unsigned long long *>(&g_f2_prof[(threadIdx.x >= 256 ? 8 : 0) + k_]), (unsigned long long)prof_acc[k_]); } while (0)
#else
#define F2_PROF_DECL
#define F2_PROF(k)
#define F2_PROF_FLUSH
#endif

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((__vector_size__(4 * sizeof(short))));

constexpr int F2_RING = 1;                                     // tiles in flight per workgroup beyond the one being stored (staging registers: 16 + 8 per tile and thread; 2 measured no faster)
constexpr int F2_THREADS = 512, F2_WAVES = 8, F2_F = 128, F2_WS = F2_F + 8, F2_JP = 64, F2_GRID_MAX = 256;
constexpr int F2_WH_BYTES = 2 * F2_F * F2_WS * 2;            // w2 planes [2][k][WS] fp16
constexpr int F2_IMG_PLANE = 32 * 256;                         // one plane of a tile image
constexpr int F2_IMG_BYTES = 2 * F2_IMG_PLANE;                 // both planes
constexpr int F2_RB_FRAG = 64 * 16;                            // one plane of one rbf fragment (a 16-byte word per lane)
constexpr int F2_RB_BYTES = 4 * 2 * F2_RB_FRAG;                // (s2, jb) x plane
constexpr int F2_OFF_G = F2_WH_BYTES, F2_OFF_H = F2_OFF_G + 2 * F2_IMG_BYTES, F2_OFF_RB = F2_OFF_H + 2 * F2_IMG_BYTES;
constexpr int F2_OFF_RED = F2_OFF_RB + 2 * F2_RB_BYTES, F2_LDS_BYTES = F2_OFF_RED + 64;
static_assert(F2_LDS_BYTES <= 160 * 1024, "k_filter_bwd2: LDS budget");
static_assert(32 * F2_F * 4 <= 2 * F2_IMG_BYTES, "the column-sum reduction reuses the G images");

// byte offset of 16-byte chunk ch (0..15) of row `row` in a [32][128 x 16-bit] image with 256-byte rows: the chunk index is XORed with a
// row-dependent pattern so that ds_read_b128 row reads (32 lanes, one row each, same chunk) and ds_read_b64_tr_b16 blocks (4 rows x 32 bytes
// per 16 lanes) both touch every bank once
__device__ __forceinline__ int f2_off(int row, int ch) { return 256 * row + 16 * (ch ^ (((row & 3) << 2) | ((row >> 2) & 3))); }

// ds_read_b64_tr_b16: the 16 lanes of a group read a 4-row x 16-column block; lane 4q+p supplies the address of row q, columns 4p..4p+3;
// lane i receives column i of the four rows (element j = row j).  EXEC must be all ones (tools/probes/tr_read_probe.hip pins the mapping).
__device__ __forceinline__ f16x4 f2_tr(const char *p) {
    const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)p);
    return __builtin_bit_cast(f16x4, v);
}
// 8 consecutive tile rows r0 .. r0 + 7 of the column this lane stands for (16-lane group `gidx` of the half <-> columns 16 gidx .. + 15 of the
// 32-column block that starts at chunk c0): two transposed reads
__device__ __forceinline__ f16x8 f2_tr8(const char *plane, int r0, int c0, int li) {
    const int q = li >> 2, p = li & 3;
    const f16x4 a = f2_tr(plane + f2_off(r0 + q, c0 + (p >> 1)) + 8 * (p & 1));
    const f16x4 b = f2_tr(plane + f2_off(r0 + 4 + q, c0 + (p >> 1)) + 8 * (p & 1));
    f16x8 r;
    r[0] = a[0]; r[1] = a[1]; r[2] = a[2]; r[3] = a[3]; r[4] = b[0]; r[5] = b[1]; r[6] = b[2]; r[7] = b[3];
    return r;
}
__device__ __forceinline__ void f2_split2h(const float *v, float sc, f16x8 &p1, f16x8 &p2) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float x = v[j] * sc;
        const _Float16 h1 = (_Float16)x;
        p1[j] = h1; p2[j] = (_Float16)(x - (float)h1);
    }
}
__device__ __forceinline__ void f2_plane_scale(float amax, float &sc, float &un) {      // 2^k with amax * 2^k in [256, 512) (filter_fused.hip)
    sc = 1.0f; un = 1.0f;
    if (amax > 0.f && amax < 3.0e38f) { int e; (void)frexpf(amax, &e); sc = ldexpf(1.0f, 9 - e); un = ldexpf(1.0f, e - 9); }
}

__global__ void __launch_bounds__(F2_THREADS, 1) k_filter_bwd2(const float *__restrict__ g, const float *__restrict__ h1, const float *__restrict__ dist,
                                                               const float *__restrict__ offset, int Gs, float coeff, const float *__restrict__ w2, int M,
                                                               const int *__restrict__ m_dev, float *__restrict__ slabs1, float *__restrict__ bias1,
                                                               float *__restrict__ slabs2, float *__restrict__ bias2, const float *__restrict__ gmax) {
    constexpr int F = F2_F, WS = F2_WS;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    _Float16 *WH = reinterpret_cast<_Float16 *>(smem);          // [2][k][WS]: plane row k holds w2[n][k], n = 0..F-1
    char *GI = smem + F2_OFF_G, *HI = smem + F2_OFF_H, *RB = smem + F2_OFF_RB;
    float *wred = reinterpret_cast<float *>(smem + F2_OFF_RED);
    if (m_dev) M = min(M, *m_dev);
    const int tiles = (M + 31) >> 5;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, h = lane >> 5, li = lane & 15, gidx = (lane >> 4) & 1;
    float gsc = 1.0f, gun = 1.0f, wsc = 1.0f, wun = 1.0f;

    // ---- stage the two fp16 planes of w2, transposed: a thread owns a 4(n) x 4(k) block (filter_bwd.hip / gemm_t.hip) --------------
    {
        constexpr int PATCHES = (F / 16) * (F / 64), PERW = (PATCHES + F2_WAVES - 1) / F2_WAVES;
        float4 wv[PERW][4];
        const int n4l = (lane & 3) | ((lane >> 4) << 2), k4l = (lane >> 2) & 3;
#pragma unroll
        for (int u = 0; u < PERW; ++u) {
            const int pt = wave + u * F2_WAVES;
            const int r0 = (pt / (F / 64)) * 16 + 4 * k4l, c0 = (pt % (F / 64)) * 64 + 4 * n4l;       // w2 rows r0.., columns c0..
#pragma unroll
            for (int j = 0; j < 4; ++j)
                wv[u][j] = pt < PATCHES ? *reinterpret_cast<const float4 *>(w2 + (size_t)(r0 + j) * F + c0) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        float am = 0.f;
#pragma unroll
        for (int u = 0; u < PERW; ++u)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                am = fmaxf(fmaxf(am, fmaxf(fabsf(wv[u][j].x), fabsf(wv[u][j].y))), fmaxf(fabsf(wv[u][j].z), fabsf(wv[u][j].w)));
        am = wave_max(am);
        if (lane == 0) wred[wave] = am;
        __syncthreads();
        float wmax = wred[0];
#pragma unroll
        for (int w = 1; w < F2_WAVES; ++w) wmax = fmaxf(wmax, wred[w]);
        f2_plane_scale(wmax, wsc, wun);
        // gradient scale: s * gmax in [16, 32), lowered when the weights are so large that s * dh1 (<= 32 * F * max |w2|) could leave fp16
        const float gm = *gmax;
        if (gm > 0.f && gm < 3.0e38f) {
            int e; (void)frexpf(gm, &e);
            int sh = 5 - e;
            const float bound = 32.0f * F * wmax;
            if (bound > 16384.0f && bound < 3.0e38f) { int eb; (void)frexpf(bound * (1.0f / 16384.0f), &eb); sh -= eb; }
            gsc = ldexpf(1.0f, sh); gun = ldexpf(1.0f, -sh);
        }
#pragma unroll
        for (int u = 0; u < PERW; ++u) {
            const int pt = wave + u * F2_WAVES;
            if (pt >= PATCHES) continue;
            const int r0 = (pt / (F / 64)) * 16 + 4 * k4l, c0 = (pt % (F / 64)) * 64 + 4 * n4l;
#pragma unroll
            for (int e = 0; e < 4; ++e) {                     // plane row c0 + e: elements r0 .. r0 + 3
                const float v4[4] = {e == 0 ? wv[u][0].x : e == 1 ? wv[u][0].y : e == 2 ? wv[u][0].z : wv[u][0].w,
                                     e == 0 ? wv[u][1].x : e == 1 ? wv[u][1].y : e == 2 ? wv[u][1].z : wv[u][1].w,
                                     e == 0 ? wv[u][2].x : e == 1 ? wv[u][2].y : e == 2 ? wv[u][2].z : wv[u][2].w,
                                     e == 0 ? wv[u][3].x : e == 1 ? wv[u][3].y : e == 2 ? wv[u][3].z : wv[u][3].w};
                f16x4 q1, q2;
#pragma unroll
                for (int j = 0; j < 4; ++j) { const float x = v4[j] * wsc; q1[j] = (_Float16)x; q2[j] = (_Float16)(x - (float)q1[j]); }
                *reinterpret_cast<f16x4 *>(&WH[(0 * F + c0 + e) * WS + r0]) = q1;
                *reinterpret_cast<f16x4 *>(&WH[(1 * F + c0 + e) * WS + r0]) = q2;
            }
        }
    }

    // Rows are walked from the END (filter_bwd.hip: the kernel that produced g streamed it front to back through the Infinity Cache, so
    // its tail is what the cache still holds).  Logical tile t of this workgroup is physical tile tiles - 1 - (blockIdx.x + t * gridDim.x).
    const int G = gridDim.x;
    auto phys = [&](int t) { return tiles - 1 - (int)(blockIdx.x + t * G); };      // < 0: no such tile
    const int srow = tid >> 4, sch = tid & 15;                                        // staging role: row of the tile, 16-byte chunk (8 channels)
    // Staging registers: a ring of F2_RING tiles per thread — the rows of tile t + 1 + F2_RING are requested when tile t + 1 is stored, so
    // F2_RING tiles (32 KB each per workgroup) are in flight at any time.
    struct Rows { float4 ga, gb, ha, hb; float dv[8]; bool ok; };      // (dv: B wavefronts, the 8 distances of their rbf fragment; ok: the row exists)
    Rows R[F2_RING];
    float cs[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) cs[j] = 0.f;
    // B wavefront u = wave - 4 generates the rbf fragment (s2, jb) = (u >> 1, u & 1) of a tile: lane (column 32 jb + l31, row group h) holds the
    // 8 rows 16 s2 + 4h + {0..3, 8..11} — the order in which a lane-half enumerates the dx accumulator's registers 8 s2 .. 8 s2 + 7
    const int bu = wave - 4, b_s2 = bu >> 1, b_jb = bu & 1;
    float mu = 0.f;
    int jkind = 1;                                             // 0 = Gaussian, 1 = zero, 2 = one (bias column 63)
    if (wave >= 4) {
        const int j = 32 * b_jb + l31;
        jkind = j < Gs ? 0 : (j == F2_JP - 1 ? 2 : 1);
        mu = j < Gs ? offset[j] : 0.f;
    }
    auto load_rows = [&](int t, Rows &r) {
        const int pt = phys(t);
        const int m = (pt << 5) + srow;
        const bool ok = pt >= 0 && m < M;
        const size_t o = (size_t)(ok ? m : 0) * F + 8 * sch;
        // (rows that do not exist read row 0 and are zeroed when they are STORED: a select here would make the loads synchronous)
        r.ga = *reinterpret_cast<const float4 *>(g + o); r.gb = *reinterpret_cast<const float4 *>(g + o + 4);
        r.ha = *reinterpret_cast<const float4 *>(h1 + o); r.hb = *reinterpret_cast<const float4 *>(h1 + o + 4);
        r.ok = ok;
        if (wave >= 4) {
            const int e0 = (max(pt, 0) << 5) + 16 * b_s2 + 4 * h;
#pragma unroll
            for (int j = 0; j < 8; ++j) r.dv[j] = dist[max(min(e0 + (j & 3) + 8 * (j >> 2), M - 1), 0)];      // (M == 0: a batch without pairs reads dist[0], unused)
        }
    };
    auto store_rows = [&](int buf, const Rows &r) {
        float gv[8] = {r.ga.x, r.ga.y, r.ga.z, r.ga.w, r.gb.x, r.gb.y, r.gb.z, r.gb.w};
        float hv[8] = {r.ha.x, r.ha.y, r.ha.z, r.ha.w, r.hb.x, r.hb.y, r.hb.z, r.hb.w};
#pragma unroll
        for (int j = 0; j < 8; ++j) { gv[j] = r.ok ? gv[j] : 0.f; hv[j] = r.ok ? hv[j] : 0.f; }
#pragma unroll
        for (int j = 0; j < 8; ++j) cs[j] += gv[j];
        f16x8 p1, p2;
        const int o = buf * F2_IMG_BYTES + f2_off(srow, sch);
        f2_split2h(gv, gsc, p1, p2);
        *reinterpret_cast<f16x8 *>(GI + o) = p1;
        *reinterpret_cast<f16x8 *>(GI + o + F2_IMG_PLANE) = p2;
        f2_split2h(hv, 1.0f, p1, p2);
        *reinterpret_cast<f16x8 *>(HI + o) = p1;
        *reinterpret_cast<f16x8 *>(HI + o + F2_IMG_PLANE) = p2;
        if (wave >= 4) {
            float rv[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float t = r.dv[j] - mu;
                const float ex = exp_neg_f(coeff * (t * t));
                rv[j] = jkind == 0 ? ex : (jkind == 2 ? 1.0f : 0.0f);
            }
            f2_split2h(rv, 1.0f, p1, p2);
            char *ro = RB + buf * F2_RB_BYTES + (bu * 2) * F2_RB_FRAG + lane * 16;
            *reinterpret_cast<f16x8 *>(ro) = p1;
            *reinterpret_cast<f16x8 *>(ro + F2_RB_FRAG) = p2;
        }
    };

    f32x16 accs[4];                                            // A: [jb] = dw1[32kb.., 32jb..] (two of them)      B: [nb] = dw2[32nb.., 32(wave-4)..]
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) accs[a][r] = 0.f;

    const int my_tiles = tiles > (int)blockIdx.x ? (tiles - 1 - (int)blockIdx.x) / G + 1 : 0;
#pragma unroll
    for (int i = 0; i < F2_RING; ++i) load_rows(i, R[i]);
    store_rows(0, R[0]);
    load_rows(F2_RING, R[0]);
    __syncthreads();                                           // w2 planes, tile 0

    F2_PROF_DECL;
    for (int t0 = 0; t0 < my_tiles; t0 += F2_RING) {
#pragma unroll
    for (int u = 0; u < F2_RING; ++u) {
        const int t = t0 + u;
        if (t >= my_tiles) break;
        const int buf = t & 1;
        const char *Gp = GI + buf * F2_IMG_BYTES, *Hp = HI + buf * F2_IMG_BYTES;
        // The next tile goes into the other buffer (free since the barrier that ended iteration t - 1).  The B wavefronts stage it BEFORE their
        // matrix work, the A wavefronts AFTER theirs: the two wavefronts of a SIMD are an A and a B, so one splits and stores while the other
        // feeds the matrix pipe (with both staging first, the barrier lined the vector phase and the matrix phase of all eight up one behind
        // the other: 41 % vector issue, 26 % matrix pipe, 42 % of the wave cycles waiting).
        auto stage_next = [&]() {
            if (t + 1 < my_tiles) {
                Rows &r = R[(u + 1) % F2_RING];                // ring slot of tile t + 1 (t0 is a multiple of F2_RING)
                store_rows(buf ^ 1, r);
                load_rows(t + 1 + F2_RING, r);                 // unconditionally (a tile past the end reads row 0 and is never stored): the same number
                                                               // of loads on every path lets the compiler wait for exactly the ring slot it stores
            }
        };
        F2_PROF(0);            // (loop overhead)
        if (wave >= 4) stage_next();
        F2_PROF(1);            // B: staging
        if (wave < 4) {
            const int kb = wave;
            // ---- dx strip: acc[r] = sum_n g[e][n] w2[n][32kb + l31], e = (r&3) + 8(r>>2) + 4h ---------------------------------
            // the three partial products of the fp16 planes accumulate in THREE accumulators (accs[2], accs[3] are free in an A wavefront): back-to-back
            // MFMAs on one accumulator wait for each other's 16 passes, and this wavefront has a single channel block to work on
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc[r] = 0.f; accs[2][r] = 0.f; accs[3][r] = 0.f; }
#pragma unroll
            for (int s = 0; s < F / 16; ++s) {
                const int o = f2_off(l31, 2 * s + h);
                const f16x8 a1 = *reinterpret_cast<const f16x8 *>(Gp + o), a2 = *reinterpret_cast<const f16x8 *>(Gp + o + F2_IMG_PLANE);
                const f16x8 b1 = *reinterpret_cast<const f16x8 *>(&WH[(0 * F + 32 * kb + l31) * WS + 16 * s + 8 * h]);
                const f16x8 b2 = *reinterpret_cast<const f16x8 *>(&WH[(1 * F + 32 * kb + l31) * WS + 16 * s + 8 * h]);
                accs[2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a2, b1, accs[2], 0, 0, 0);
                accs[3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b2, accs[3], 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b1, acc, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = (accs[2][r] + accs[3][r]) + acc[r];      // the two small cross terms first
            F2_PROF(2);        // A: dx strip
            // ---- dh1 = acc / (w2 scale) * ssp'(h1): h1 of (row e(r), channel 32kb + l31) by transposed reads of H -----------------
            float dh[16];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int qq = li >> 2, pp = li & 3;
                const int o = f2_off(8 * q + 4 * h + qq, 4 * kb + 2 * gidx + (pp >> 1)) + 8 * (pp & 1);
                const f16x4 t1 = f2_tr(Hp + o), t2 = f2_tr(Hp + o + F2_IMG_PLANE);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float hvv = (float)t1[j] + (float)t2[j];
                    dh[4 * q + j] = acc[4 * q + j] * wun * (1.0f - 0.5f * __expf(-hvv));      // ssp'(pre) from the saved output
                }
            }
            F2_PROF(3);        // A: h1 by transposed reads, ssp'
            // ---- dw1[32kb.., :] += dh1^T rbf ---------------------------------------------------------------------------------
            const char *Rp = RB + buf * F2_RB_BYTES + lane * 16;
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                f16x8 a1, a2;
                f2_split2h(dh + 8 * s2, 1.0f, a1, a2);
#pragma unroll
                for (int jb = 0; jb < 2; ++jb) {
                    const f16x8 r1 = *reinterpret_cast<const f16x8 *>(Rp + ((s2 * 2 + jb) * 2) * F2_RB_FRAG);
                    const f16x8 r2 = *reinterpret_cast<const f16x8 *>(Rp + ((s2 * 2 + jb) * 2 + 1) * F2_RB_FRAG);
                    accs[jb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a2, r1, accs[jb], 0, 0, 0);
                    accs[jb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, r2, accs[jb], 0, 0, 0);
                    accs[jb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, r1, accs[jb], 0, 0, 0);
                }
            }
        }
        else {
            // ---- dw2[:, 32kb..] += g^T h1: both operands column-wise out of the row-major images -----------------------------------
            const int kb = wave - 4;
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const int r0 = 16 * s2 + 8 * h;
                const f16x8 hb1 = f2_tr8(Hp, r0, 4 * kb + 2 * gidx, li), hb2 = f2_tr8(Hp + F2_IMG_PLANE, r0, 4 * kb + 2 * gidx, li);
#pragma unroll
                for (int nb = 0; nb < 4; ++nb) {
                    const f16x8 g1 = f2_tr8(Gp, r0, 4 * nb + 2 * gidx, li), g2 = f2_tr8(Gp + F2_IMG_PLANE, r0, 4 * nb + 2 * gidx, li);
                    accs[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(g2, hb1, accs[nb], 0, 0, 0);
                    accs[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(g1, hb2, accs[nb], 0, 0, 0);
                    accs[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(g1, hb1, accs[nb], 0, 0, 0);
                }
            }
        }
        F2_PROF(4);            // A: dw1 products / B: dw2 products
        if (wave < 4) stage_next();
        F2_PROF(5);            // A: staging
        __syncthreads();                                       // the other buffer is complete; this one may be overwritten
        F2_PROF(6);            // barrier
    }
    }

    F2_PROF_FLUSH;
    // ---- slabs of this workgroup: every strip comes from exactly one wavefront ---------------------------------------------------------
    if (wave < 4) {
        float *slab = slabs1 + (size_t)blockIdx.x * F * Gs;
#pragma unroll
        for (int jb = 0; jb < 2; ++jb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int k = 32 * wave + (r & 3) + 8 * (r >> 2) + 4 * h, j = 32 * jb + l31;
                const float val = accs[jb][r] * gun;           // the accumulators carry the gradient's scale
                if (j < Gs) slab[k * Gs + j] = val;
                if (j == F2_JP - 1) bias1[(size_t)blockIdx.x * F + k] = val;
            }
    } else {
        float *slab = slabs2 + (size_t)blockIdx.x * F * F;
#pragma unroll
        for (int nb = 0; nb < 4; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = 32 * nb + (r & 3) + 8 * (r >> 2) + 4 * h, k = 32 * (wave - 4) + l31;
                slab[n * F + k] = accs[nb][r] * gun;
            }
    }
    // db2 = column sums of g: the 32 staging rows of a chunk are added in row order
    float *PS = reinterpret_cast<float *>(GI);
    *reinterpret_cast<float4 *>(&PS[srow * F + 8 * sch]) = make_float4(cs[0], cs[1], cs[2], cs[3]);
    *reinterpret_cast<float4 *>(&PS[srow * F + 8 * sch + 4]) = make_float4(cs[4], cs[5], cs[6], cs[7]);
    __syncthreads();
    if (tid < F) {
        float t = 0.f;
#pragma unroll 8
        for (int r = 0; r < 32; ++r) t += PS[r * F + tid];
        bias2[(size_t)blockIdx.x * F + tid] = t;
    }
}

}  // namespace

int conan_wgrad_reduce_now(const float *slabs, const float *bias_slabs, int slices, int NK, int N, float *dW, float *dbias, hipStream_t s);   // gemm.hip

extern "C" {

int conan_filter_bwd2_supported(int num_gaussians, int num_filters) { return num_filters == F2_F && num_gaussians >= 1 && num_gaussians <= F2_JP - 1; }

int conan_filter_bwd2_slices(int M) {
    const int tiles = (M + 31) / 32;
    return tiles < 1 ? 1 : (tiles > F2_GRID_MAX ? F2_GRID_MAX : tiles);
}

long long conan_filter_bwd2_ws(int M, int num_gaussians, int num_filters) {
    const long long s = conan_filter_bwd2_slices(M), F = num_filters;
    return s * (F * num_gaussians + F) + s * (F * F + F);       // [slabs1 | bias1] then [slabs2 | bias2]: two conan_wgrad_reduce jobs
}

int conan_filter_bwd2(const float *g, const float *h1, const float *dist, int M, const float *offset, int num_gaussians, float coeff,
                      const float *w2, int num_filters, const int *m_dev, const float *gmax, float *dW1, float *db1, float *dW2, float *db2,
                      float *ws, void *stream) {
    if (!g || !h1 || !dist || !offset || !w2 || !ws || !gmax || M < 1 || ((dW1 == nullptr) != (dW2 == nullptr))) return CONAN_E_BADARG;
    if (!conan_filter_bwd2_supported(num_gaussians, num_filters)) return CONAN_E_UNSUPPORTED;
    const int F = num_filters, Gs = num_gaussians, slices = conan_filter_bwd2_slices(M);
    hipStream_t s = as_stream(stream);
    float *slabs1 = ws, *bias1 = slabs1 + (size_t)slices * F * Gs, *slabs2 = bias1 + (size_t)slices * F, *bias2 = slabs2 + (size_t)slices * F * F;
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_filter_bwd2), hipFuncAttributeMaxDynamicSharedMemorySize, F2_LDS_BYTES);
    k_filter_bwd2<<<slices, F2_THREADS, F2_LDS_BYTES, s>>>(g, h1, dist, offset, Gs, coeff, w2, M, m_dev, slabs1, bias1, slabs2, bias2, gmax);
    CONAN_LAUNCH_CHECK();
    if (!dW1) return CONAN_OK;      // slabs only: two conan_wgrad_reduce_batch jobs (ws, K = Gs) and (ws + slices * (F * Gs + F), K = F), slices = conan_filter_bwd2_slices(M)
    const int rc = conan_wgrad_reduce_now(slabs1, bias1, slices, F * Gs, F, dW1, db1, s);
    if (rc != CONAN_OK) return rc;
    return conan_wgrad_reduce_now(slabs2, bias2, slices, F * F, F, dW2, db2, s);
}

}  // extern "C"
