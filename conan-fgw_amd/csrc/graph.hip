// Graph construction on the device: graph_ptr from `batch`, fixed-radius neighbour lists as CSR by target,
// the by-source transpose, and export to the reference's int64 edge_index.
//
// Layout choice (MI355X): a conformer graph has <= ~120 atoms, so one workgroup owns one graph, stages its positions
// in LDS once and every thread scans sources in ascending index order for one target.  The output is a compact CSR
// (no padding to `cap`) so that every edge-level kernel downstream streams exactly E rows.
#include "common.h"

namespace {

__global__ void k_graph_ptr(const int64_t *__restrict__ batch, int n, int G, int *__restrict__ ptr) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    long long cur = batch[i];
    long long prev = i == 0 ? -1 : batch[i - 1];
    for (long long g = prev + 1; g <= cur && g < G; ++g) ptr[g] = i;     // first atom of g (and of empty graphs before it)
    if (i == n - 1)
        for (long long g = cur + 1; g <= G; ++g) ptr[g] = n;
}
__global__ void k_graph_ptr_empty(int G, int *ptr) {
    int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g <= G) ptr[g] = 0;
}

// d2 with the rounding sequence fixed: fl(fl(dx*dx + dy*dy) + dz*dz), no FMA contraction.
__device__ __forceinline__ float dist2_rn(float ax, float ay, float az, float bx, float by, float bz) {
    float dx = __fsub_rn(ax, bx), dy = __fsub_rn(ay, by), dz = __fsub_rn(az, bz);
    return __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
}

constexpr int RG_THREADS = 128;
constexpr int RG_MAX_LDS_ATOMS = 2048;

// PASS 0: count neighbours per target.  PASS 1: fill col/tgt/dist at rowptr offsets.
template <int PASS>
__global__ void __launch_bounds__(RG_THREADS) k_radius(const float *__restrict__ pos, const int *__restrict__ gptr, float r2,
                                                       int cap, int loop, int *__restrict__ deg,
                                                       const int *__restrict__ rowptr, int *__restrict__ col,
                                                       int *__restrict__ tgt, float *__restrict__ dist) {
    __shared__ float sp[RG_MAX_LDS_ATOMS * 3];
    const int g = blockIdx.x;
    const int lo = gptr[g], hi = gptr[g + 1];
    const int n = hi - lo;
    const bool in_lds = n <= RG_MAX_LDS_ATOMS;
    if (in_lds)
        for (int t = threadIdx.x; t < n * 3; t += RG_THREADS) sp[t] = pos[(size_t)lo * 3 + t];
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += RG_THREADS) {
        float ax, ay, az;
        if (in_lds) { ax = sp[i * 3]; ay = sp[i * 3 + 1]; az = sp[i * 3 + 2]; }
        else { ax = pos[(size_t)(lo + i) * 3]; ay = pos[(size_t)(lo + i) * 3 + 1]; az = pos[(size_t)(lo + i) * 3 + 2]; }
        // torch-cluster 1.6.1: radius(x, x, r, batch, batch, loop ? cap : cap + 1) scans the sources of the graph in ascending
        // index order INCLUDING the target itself and stops after `limit` hits; radius_graph drops the self pair afterwards.
        // A target with >= cap + 1 lower-index in-range atoms therefore keeps cap + 1 edges (its own hit never enters the
        // window), every other truncated target keeps cap.
        const int limit = loop ? cap : cap + 1;
        int cnt = 0, out = 0;
        int base = PASS ? rowptr[lo + i] : 0;
        for (int j = 0; j < n && cnt < limit; ++j) {
            float bx, by, bz;
            if (in_lds) { bx = sp[j * 3]; by = sp[j * 3 + 1]; bz = sp[j * 3 + 2]; }
            else { bx = pos[(size_t)(lo + j) * 3]; by = pos[(size_t)(lo + j) * 3 + 1]; bz = pos[(size_t)(lo + j) * 3 + 2]; }
            // the reference's edge_weight is ||pos[row]-pos[col]|| with row = source j, col = target i: same d2 by symmetry
            float d2 = dist2_rn(bx, by, bz, ax, ay, az);
            if (d2 < r2) {
                ++cnt;
                if (loop || j != i) {
                    if (PASS) { col[base + out] = lo + j; tgt[base + out] = lo + i; dist[base + out] = __fsqrt_rn(d2); }
                    ++out;
                }
            }
        }
        if (!PASS) deg[lo + i] = out;
    }
}

// Single-workgroup exclusive scan through LDS for n <= SCAN_LDS_MAX (every per-atom scan of a batch): the input is read with
// coalesced loads into LDS, each thread scans its contiguous chunk there (odd chunk pitch: conflict-free), the thread totals are
// scanned by shuffles, and the result leaves with coalesced stores.  The register variant below has every lane read its own
// contiguous chunk from global memory — 25 wave-loads that each touch 64 cache lines for n = 25 k (16 us against ~6 us here).
constexpr int SCAN_LDS_MAX = 36 * 1024;          // 144 KB of LDS
__global__ void __launch_bounds__(1024) k_exclusive_scan_lds(const int *__restrict__ in, int n, int *__restrict__ out) {
    extern __shared__ int sdat[];                 // [n] + [16] wavefront totals
    int *wsum = sdat + n;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    for (int i0 = t; i0 < n; i0 += 4 * 1024) {
        int v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { const int i = i0 + u * 1024; v[u] = in[i < n ? i : n - 1]; }
#pragma unroll
        for (int u = 0; u < 4; ++u) { const int i = i0 + u * 1024; if (i < n) sdat[i] = v[u]; }
    }
    __syncthreads();
    const int chunk = ((n + 1023) / 1024) | 1;    // odd pitch between the threads' chunks: conflict-free LDS walks
    const int b = min(t * chunk, n), e = min(b + chunk, n);
    int s = 0;
    for (int i = b; i < e; ++i) s += sdat[i];
    int inc = s;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int u = __shfl_up(inc, o, 64); if (lane >= o) inc += u; }
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    int woff = 0, total = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) { const int x = wsum[w]; if (w < wave) woff += x; total += x; }
    int run = woff + inc - s;
    for (int i = b; i < e; ++i) { const int x = sdat[i]; sdat[i] = run; run += x; }
    __syncthreads();
    for (int i = t; i < n; i += 1024) out[i] = sdat[i];
    if (t == 0) out[n] = total;
}

// Single-workgroup exclusive scan, n up to a few million: out[0..n], out[n] = total.
constexpr int SCAN_THREADS = 1024;
__global__ void __launch_bounds__(SCAN_THREADS) k_exclusive_scan(const int *__restrict__ in, int n, int *__restrict__ out) {
    __shared__ int wsum[SCAN_THREADS / 64];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int chunk = (n + SCAN_THREADS - 1) / SCAN_THREADS;
    const int b = t * chunk, e = min(b + chunk, n);
    constexpr int MAXC = 32;                 // chunks of up to 32 entries (n <= 32768) live in registers: one pass over memory,
    int v[MAXC];                             // all loads of a thread in flight at once
    int s = 0;
    if (chunk <= MAXC) {
#pragma unroll
        for (int k = 0; k < MAXC; ++k) { v[k] = (b + k < e) ? in[b + k] : 0; }
#pragma unroll
        for (int k = 0; k < MAXC; ++k) s += v[k];
    } else {
        for (int i = b; i < e; ++i) s += in[i];
    }
    // exclusive scan of the 1024 thread sums: wavefront scan by shuffles, 16 wavefront totals through LDS
    int inc = s;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int u = __shfl_up(inc, o, 64); if (lane >= o) inc += u; }
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    int woff = 0, total = 0;
#pragma unroll
    for (int w = 0; w < SCAN_THREADS / 64; ++w) { const int x = wsum[w]; if (w < wave) woff += x; total += x; }
    int run = woff + inc - s;
    if (chunk <= MAXC) {
#pragma unroll
        for (int k = 0; k < MAXC; ++k) if (b + k < e) { out[b + k] = run; run += v[k]; }
    } else {
        for (int i = b; i < e; ++i) { const int x = in[i]; out[i] = run; run += x; }
    }
    if (t == 0) out[n] = total;
}

static void launch_exclusive_scan(const int *in, int n, int *out, hipStream_t s) {
    if (n > 0 && n <= SCAN_LDS_MAX) {
        const size_t lds = (size_t)(n + 16) * sizeof(int);
        if (lds > 64 * 1024)
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_exclusive_scan_lds), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        k_exclusive_scan_lds<<<1, 1024, lds, s>>>(in, n, out);
    } else {
        k_exclusive_scan<<<1, SCAN_THREADS, 0, s>>>(in, n, out);
    }
}

// Multi-workgroup exclusive scan for long inputs (the pair flags: up to cap * num_atoms entries): per-block sums, a
// single-workgroup scan of the (few hundred) block sums, then a per-block local scan with the block offset.
constexpr int SB_THREADS = 256;
constexpr int SB_ITEMS = 16;                 // 4096 elements per workgroup
__global__ void __launch_bounds__(SB_THREADS) k_scan_block_sums(const int *__restrict__ in, int n, int *__restrict__ bsum) {
    __shared__ int red[SB_THREADS / 64];
    const int base = blockIdx.x * SB_THREADS * SB_ITEMS + threadIdx.x * SB_ITEMS;
    int s = 0;
#pragma unroll
    for (int u = 0; u < SB_ITEMS; ++u) s += (base + u < n) ? in[base + u] : 0;
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) bsum[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}
__global__ void __launch_bounds__(SB_THREADS) k_scan_block_apply(const int *__restrict__ in, int n, const int *__restrict__ boff,
                                                                 int nblocks, int *__restrict__ out) {
    __shared__ int part[SB_THREADS];
    const int t = threadIdx.x;
    const int base = blockIdx.x * SB_THREADS * SB_ITEMS + t * SB_ITEMS;
    int v[SB_ITEMS], s = 0;
#pragma unroll
    for (int u = 0; u < SB_ITEMS; ++u) { v[u] = (base + u < n) ? in[base + u] : 0; s += v[u]; }
    part[t] = s;
    __syncthreads();
    for (int o = 1; o < SB_THREADS; o <<= 1) {
        const int a = t >= o ? part[t - o] : 0;
        __syncthreads();
        part[t] += a;
        __syncthreads();
    }
    int run = boff[blockIdx.x] + (t == 0 ? 0 : part[t - 1]);
#pragma unroll
    for (int u = 0; u < SB_ITEMS; ++u) { if (base + u < n) out[base + u] = run; run += v[u]; }
    if (blockIdx.x == nblocks - 1 && t == SB_THREADS - 1) out[n] = boff[nblocks];      // grand total
}
// out[0..n], out[n] = total; ws holds 2 * (nblocks + 1) ints
static void scan_large(const int *in, int n, int *out, int *ws, hipStream_t s) {
    const int per = SB_THREADS * SB_ITEMS;
    const int nblocks = (n + per - 1) / per;
    int *bsum = ws, *boff = ws + nblocks + 1;
    k_scan_block_sums<<<nblocks, SB_THREADS, 0, s>>>(in, n, bsum);
    launch_exclusive_scan(bsum, nblocks, boff, s);
    k_scan_block_apply<<<nblocks, SB_THREADS, 0, s>>>(in, n, boff, nblocks, out);
}

// by-source transpose, one workgroup per graph; deterministic (ascending edge id inside each source row).
// The by-source lists of a graph fill exactly the edge range [e0, e1) of that graph, so the row pointers need no global
// scan: t_rowptr[j] = e0 + (number of edges of the graph whose source is < j).  One wavefront per source atom scans the
// graph's edges 64 at a time: ballot + popcount give the row start (scan 1) and the in-order write slots (scan 2).
constexpr int TR_THREADS = 1024;     // 16 wavefronts per graph: a BACE / Lipophilicity-sized graph has 64-97 source atoms to place (4 wavefronts: 107 us per step at Lipophilicity B = 128)
__global__ void __launch_bounds__(TR_THREADS) k_transpose_graph(const int *__restrict__ gptr, const int *__restrict__ rowptr,
                                                                const int *__restrict__ col, int num_atoms,
                                                                int *__restrict__ t_rowptr, int *__restrict__ t_eid) {
    const int g = blockIdx.x;
    const int lo = gptr[g], hi = gptr[g + 1];
    if (g == gridDim.x - 1 && threadIdx.x == 0) t_rowptr[num_atoms] = rowptr[num_atoms];
    if (hi <= lo) return;
    const int e0 = rowptr[lo], e1 = rowptr[hi];
    const int steps = (e1 - e0 + 63) >> 6;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned long long below = (1ull << lane) - 1ull;
    constexpr int TR_REG = 18;             // 18 x 64 = 1 152 edges: every conformer of <= 36 atoms at cap 32
    if (steps <= TR_REG) {
        // the usual case: the graph's sources are read ONCE into registers, every source atom
        // is then placed with ballots alone (no memory traffic inside the per-atom loop)
        int cv[TR_REG];
#pragma unroll
        for (int s = 0; s < TR_REG; ++s) {
            const int e = e0 + s * 64 + lane;
            cv[s] = (s < steps && e < e1) ? col[e] : 0x7fffffff;
        }
        for (int j = lo + wave; j < hi; j += TR_THREADS / 64) {
            int less = 0;
#pragma unroll
            for (int s = 0; s < TR_REG; ++s) less += __popcll(__ballot(cv[s] < j));
            int base = e0 + less;
            if (lane == 0) t_rowptr[j] = base;
#pragma unroll
            for (int s = 0; s < TR_REG; ++s) {
                const bool hit = cv[s] == j;
                const unsigned long long m = __ballot(hit);
                if (hit) t_eid[base + __popcll(m & below)] = e0 + s * 64 + lane;
                base += __popcll(m);
            }
        }
        return;
    }
    // larger graphs (BACE at cap 32: ~1 500-2 800 edges): the sources are staged in LDS once — the general loop below re-reads them from
    // global memory twice per source atom, 254 us per step at BACE B = 64
    constexpr int TR_LDS = 4096;
    __shared__ int sc[TR_LDS];
    if (steps * 64 <= TR_LDS) {
        for (int q = threadIdx.x; q < steps * 64; q += TR_THREADS) sc[q] = (e0 + q < e1) ? col[e0 + q] : 0x7fffffff;
        __syncthreads();
        for (int j = lo + wave; j < hi; j += TR_THREADS / 64) {
            int less = 0;
            for (int s = 0; s < steps; ++s) less += __popcll(__ballot(sc[s * 64 + lane] < j));
            int base = e0 + less;
            if (lane == 0) t_rowptr[j] = base;
            for (int s = 0; s < steps; ++s) {
                const bool hit = sc[s * 64 + lane] == j;
                const unsigned long long m = __ballot(hit);
                if (hit) t_eid[base + __popcll(m & below)] = e0 + s * 64 + lane;
                base += __popcll(m);
            }
        }
        return;
    }
    for (int j = lo + wave; j < hi; j += TR_THREADS / 64) {
        int less = 0;
        for (int s = 0; s < steps; ++s) {
            const int e = e0 + s * 64 + lane;
            less += __popcll(__ballot(e < e1 && col[e] < j));
        }
        int base = e0 + less;
        if (lane == 0) t_rowptr[j] = base;
        for (int s = 0; s < steps; ++s) {
            const int e = e0 + s * 64 + lane;
            const bool hit = e < e1 && col[e] == j;
            const unsigned long long m = __ballot(hit);
            if (hit) t_eid[base + __popcll(m & below)] = e;
            base += __popcll(m);
        }
    }
}

// ---- undirected pairs ----------------------------------------------------------------------------------------------
// The continuous filter depends on the edge only through d_ij = d_ji, so both directions of a pair can share one filter
// row.  An edge (s -> t) is the REPRESENTATIVE of its pair when s <= t, or when s > t and the reverse edge (t -> s) is not
// in the graph (possible once the neighbour cap truncates rows).  Rows are sorted by source => binary search.
__device__ __forceinline__ int find_edge(const int *__restrict__ rowptr, const int *__restrict__ col, int target, int source) {
    int lo = rowptr[target], hi = rowptr[target + 1] - 1;
    while (lo <= hi) {
        const int mid = (lo + hi) >> 1;
        const int c = col[mid];
        if (c == source) return mid;
        if (c < source) lo = mid + 1; else hi = mid - 1;
    }
    return -1;
}
__global__ void k_pair_flag(const int *__restrict__ rowptr, const int *__restrict__ col, const int *__restrict__ tgt,
                            const int *__restrict__ ne_dev, int max_edges, int *__restrict__ flag) {
    const int E = min(*ne_dev, max_edges);
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < max_edges; e += gridDim.x * blockDim.x) {
        int f = 0;
        if (e < E) {
            const int s = col[e], t = tgt[e];
            f = (s <= t) ? 1 : (find_edge(rowptr, col, s, t) < 0 ? 1 : 0);
        }
        flag[e] = f;
    }
}
__global__ void k_pair_fill(const int *__restrict__ rowptr, const int *__restrict__ col, const int *__restrict__ tgt, const float *__restrict__ dist,
                            const int *__restrict__ ne_dev, int max_edges, const int *__restrict__ flag, const int *__restrict__ pidx,
                            int *__restrict__ pid, int *__restrict__ pe0, int *__restrict__ pe1, float *__restrict__ pdist) {
    const int E = min(*ne_dev, max_edges);
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < E; e += gridDim.x * blockDim.x) {
        const int s = col[e], t = tgt[e];
        const int rev = (s == t) ? -1 : find_edge(rowptr, col, s, t);      // the edge (t -> s) lives in row s
        if (flag[e]) {
            const int p = pidx[e];
            pid[e] = p; pe0[p] = e; pe1[p] = rev; pdist[p] = dist[e];
        } else {
            pid[e] = pidx[rev];
        }
    }
}

__global__ void k_edge_index(const int *__restrict__ col, const int *__restrict__ tgt, int E, int64_t *__restrict__ ei) {
    int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e < E) { ei[e] = col[e]; ei[(size_t)E + e] = tgt[e]; }
}


// ---- covalent-bond graph of the GAT branch ------------------------------------------------------------------------------
// edge_index[2,E] (int64, arbitrary order, PyG convention row 0 = source, row 1 = target) -> CSR by target (col = source,
// eid = original edge id) and CSR by source (t_pos = position of the edge in the by-target arrays, t_tgt = its target).
// Self loops are dropped (GATConv removes them before adding its own, gat.py:9-12 -> PyG GATConv.forward).  The fill uses
// an atomic cursor per row; every row is then sorted (insertion sort, rows are bonds: a handful of entries), so the layout —
// and with it every floating-point summation order downstream — is deterministic.
__global__ void k_bond_count(const int64_t *__restrict__ ei, int E, int n, int *__restrict__ deg_t, int *__restrict__ deg_s) {
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < E; e += gridDim.x * blockDim.x) {
        const int s = (int)ei[e], t = (int)ei[(size_t)E + e];
        if (s == t || s < 0 || t < 0 || s >= n || t >= n) continue;
        atomicAdd(&deg_t[t], 1);
        atomicAdd(&deg_s[s], 1);
    }
}
__global__ void k_bond_fill_t(const int64_t *__restrict__ ei, int E, int n, const int *__restrict__ rowptr, int *__restrict__ cursor,
                              int *__restrict__ col, int *__restrict__ eid) {
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < E; e += gridDim.x * blockDim.x) {
        const int s = (int)ei[e], t = (int)ei[(size_t)E + e];
        if (s == t || s < 0 || t < 0 || s >= n || t >= n) continue;
        const int p = rowptr[t] + atomicAdd(&cursor[t], 1);
        col[p] = s; eid[p] = e;
    }
}
// sort every row by (key, val) ascending; one thread per row
__global__ void k_rows_sort2(const int *__restrict__ rowptr, int n, int *__restrict__ key, int *__restrict__ val) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int a = rowptr[i], b = rowptr[i + 1];
        for (int p = a + 1; p < b; ++p) {
            const int k = key[p], v = val[p];
            int q = p - 1;
            while (q >= a && (key[q] > k || (key[q] == k && val[q] > v))) { key[q + 1] = key[q]; val[q + 1] = val[q]; --q; }
            key[q + 1] = k; val[q + 1] = v;
        }
    }
}
__global__ void k_bond_fill_s(const int *__restrict__ rowptr, const int *__restrict__ col, int n, const int *__restrict__ t_rowptr,
                              int *__restrict__ cursor, int *__restrict__ t_pos, int *__restrict__ t_tgt) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
        for (int p = rowptr[i]; p < rowptr[i + 1]; ++p) {
            const int s = col[p];
            const int q = t_rowptr[s] + atomicAdd(&cursor[s], 1);
            t_pos[q] = p; t_tgt[q] = i;
        }
}

}  // namespace

extern "C" {

int conan_abi_version(void) { return CONAN_FGW_ABI_VERSION; }

int conan_graph_ptr_from_batch(const int64_t *batch, int num_atoms, int num_graphs, int *graph_ptr, void *stream) {
    if (num_atoms < 0 || num_graphs < 0 || !graph_ptr) return CONAN_E_BADARG;
    if (num_atoms == 0) {
        k_graph_ptr_empty<<<(num_graphs + 256) / 256, 256, 0, as_stream(stream)>>>(num_graphs, graph_ptr);
    } else {
        if (!batch) return CONAN_E_BADARG;
        k_graph_ptr<<<(num_atoms + 255) / 256, 256, 0, as_stream(stream)>>>(batch, num_atoms, num_graphs, graph_ptr);
    }
    CONAN_LAUNCH_CHECK();
    return CONAN_OK;
}

int conan_radius_graph_csr(const float *pos, const int *graph_ptr, int num_atoms, int num_graphs, float r, int cap,
                           int loop, int *deg_ws, int *rowptr, int *col, int *tgt, float *dist, void *stream) {
    if (!pos || !graph_ptr || !deg_ws || !rowptr || !col || !tgt || !dist || num_atoms < 0 || num_graphs <= 0 || cap <= 0)
        return CONAN_E_BADARG;
    hipStream_t s = as_stream(stream);
    const float r2 = r * r;
    k_radius<0><<<num_graphs, RG_THREADS, 0, s>>>(pos, graph_ptr, r2, cap, loop, deg_ws, nullptr, nullptr, nullptr, nullptr);
    launch_exclusive_scan(deg_ws, num_atoms, rowptr, s);
    k_radius<1><<<num_graphs, RG_THREADS, 0, s>>>(pos, graph_ptr, r2, cap, loop, nullptr, rowptr, col, tgt, dist);
    CONAN_LAUNCH_CHECK();
    return CONAN_OK;
}

int conan_csr_transpose(const int *graph_ptr, int num_graphs, int num_atoms, const int *rowptr, const int *col,
                        int *deg_ws, int *t_rowptr, int *t_eid, void *stream) {
    if (!graph_ptr || !rowptr || !col || !t_rowptr || !t_eid || num_graphs <= 0) return CONAN_E_BADARG;
    hipStream_t s = as_stream(stream);
    (void)deg_ws;
    k_transpose_graph<<<num_graphs, TR_THREADS, 0, s>>>(graph_ptr, rowptr, col, num_atoms, t_rowptr, t_eid);
    CONAN_LAUNCH_CHECK();
    return CONAN_OK;
}

int conan_edge_pairs(const int *rowptr, const int *col, const int *tgt, const float *dist, const int *num_edges_dev, int max_edges,
                     int *flag_ws, int *pidx_ws, int *scan_ws, int *pid, int *pair_e0, int *pair_e1, float *pair_dist, void *stream) {
    if (!rowptr || !col || !tgt || !dist || !num_edges_dev || !flag_ws || !pidx_ws || !scan_ws || !pid || !pair_e0 || !pair_e1 || !pair_dist ||
        max_edges <= 0)
        return CONAN_E_BADARG;
    hipStream_t s = as_stream(stream);
    int blocks = (max_edges + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    k_pair_flag<<<blocks, 256, 0, s>>>(rowptr, col, tgt, num_edges_dev, max_edges, flag_ws);
    // pidx_ws[max_edges] = number of pairs; the block-sum scratch lives in the tail of pair_e1 (overwritten afterwards by
    // k_pair_fill only for pair ids < number of pairs <= number of edges, and the scan is complete by then)
    scan_large(flag_ws, max_edges, pidx_ws, scan_ws, s);
    k_pair_fill<<<blocks, 256, 0, s>>>(rowptr, col, tgt, dist, num_edges_dev, max_edges, flag_ws, pidx_ws, pid, pair_e0, pair_e1, pair_dist);
    CONAN_LAUNCH_CHECK();
    return CONAN_OK;
}

int conan_bond_graph_csr(const int64_t *edge_index, int num_edges, int num_nodes, int *ws, int *rowptr, int *col, int *eid,
                         int *t_rowptr, int *t_pos, int *t_tgt, void *stream) {
    if (num_edges < 0 || num_nodes <= 0 || !ws || !rowptr || !t_rowptr || (num_edges && (!edge_index || !col || !eid || !t_pos || !t_tgt)))
        return CONAN_E_BADARG;
    hipStream_t s = as_stream(stream);
    int *deg_t = ws, *deg_s = ws + (num_nodes + 1);
    if (hipMemsetAsync(ws, 0, sizeof(int) * 2 * (size_t)(num_nodes + 1), s) != hipSuccess) return CONAN_E_LAUNCH;
    const int eb = num_edges ? (num_edges + 255) / 256 : 1, nb = (num_nodes + 255) / 256;
    if (num_edges) k_bond_count<<<eb, 256, 0, s>>>(edge_index, num_edges, num_nodes, deg_t, deg_s);
    launch_exclusive_scan(deg_t, num_nodes, rowptr, s);
    launch_exclusive_scan(deg_s, num_nodes, t_rowptr, s);
    if (num_edges) {
        if (hipMemsetAsync(ws, 0, sizeof(int) * 2 * (size_t)(num_nodes + 1), s) != hipSuccess) return CONAN_E_LAUNCH;
        k_bond_fill_t<<<eb, 256, 0, s>>>(edge_index, num_edges, num_nodes, rowptr, deg_t, col, eid);
        k_rows_sort2<<<nb, 256, 0, s>>>(rowptr, num_nodes, col, eid);
        k_bond_fill_s<<<nb, 256, 0, s>>>(rowptr, col, num_nodes, t_rowptr, deg_s, t_pos, t_tgt);
        k_rows_sort2<<<nb, 256, 0, s>>>(t_rowptr, num_nodes, t_pos, t_tgt);
    }
    CONAN_LAUNCH_CHECK();
    return CONAN_OK;
}

int conan_edge_index_i64(const int *col, const int *tgt, int num_edges, int64_t *edge_index, void *stream) {
    if (num_edges < 0 || (num_edges && (!col || !tgt || !edge_index))) return CONAN_E_BADARG;
    if (num_edges == 0) return CONAN_OK;
    k_edge_index<<<(num_edges + 255) / 256, 256, 0, as_stream(stream)>>>(col, tgt, num_edges, edge_index);
    CONAN_LAUNCH_CHECK();
    return CONAN_OK;
}

}  // extern "C"
