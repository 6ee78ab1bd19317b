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
        int cnt = 0;
        int base = PASS ? rowptr[lo + i] : 0;
        for (int j = 0; j < n && cnt < cap; ++j) {
            if (!loop && j == i) continue;
            float bx, by, bz;
            if (in_lds) { bx = sp[j * 3]; by = sp[j * 3 + 1]; bz = sp[j * 3 + 2]; }
            else { bx = pos[(size_t)(lo + j) * 3]; by = pos[(size_t)(lo + j) * 3 + 1]; bz = pos[(size_t)(lo + j) * 3 + 2]; }
            // the reference's edge_weight is ||pos[row]-pos[col]|| with row = source j, col = target i: same d2 by symmetry
            float d2 = dist2_rn(bx, by, bz, ax, ay, az);
            if (d2 < r2) {
                if (PASS) { col[base + cnt] = lo + j; tgt[base + cnt] = lo + i; dist[base + cnt] = __fsqrt_rn(d2); }
                ++cnt;
            }
        }
        if (!PASS) deg[lo + i] = cnt;
    }
}

// Single-workgroup exclusive scan, n up to a few million: out[0..n], out[n] = total.
constexpr int SCAN_THREADS = 1024;
__global__ void __launch_bounds__(SCAN_THREADS) k_exclusive_scan(const int *__restrict__ in, int n, int *__restrict__ out) {
    __shared__ int part[SCAN_THREADS];
    const int t = threadIdx.x;
    const int chunk = (n + SCAN_THREADS - 1) / SCAN_THREADS;
    const int b = t * chunk, e = min(b + chunk, n);
    int s = 0;
    for (int i = b; i < e; ++i) s += in[i];
    part[t] = s;
    __syncthreads();
    // Hillis-Steele inclusive scan over the 1024 partials
    for (int o = 1; o < SCAN_THREADS; o <<= 1) {
        int v = t >= o ? part[t - o] : 0;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    int run = t == 0 ? 0 : part[t - 1];
    for (int i = b; i < e; ++i) { int v = in[i]; out[i] = run; run += v; }
    if (t == SCAN_THREADS - 1) out[n] = part[SCAN_THREADS - 1];
}

// by-source transpose, one workgroup per graph; deterministic (ascending edge id inside each source row)
template <int PASS>
__global__ void __launch_bounds__(RG_THREADS) k_transpose(const int *__restrict__ gptr, const int *__restrict__ rowptr,
                                                          const int *__restrict__ col, int *__restrict__ deg,
                                                          const int *__restrict__ t_rowptr, int *__restrict__ t_eid) {
    const int g = blockIdx.x;
    const int lo = gptr[g], hi = gptr[g + 1];
    if (hi <= lo) return;
    const int e0 = rowptr[lo], e1 = rowptr[hi];
    for (int j = lo + threadIdx.x; j < hi; j += RG_THREADS) {
        int cnt = 0;
        int base = PASS ? t_rowptr[j] : 0;
        for (int e = e0; e < e1; ++e)
            if (col[e] == j) { if (PASS) t_eid[base + cnt] = e; ++cnt; }
        if (!PASS) deg[j] = cnt;
    }
}

__global__ void k_edge_index(const int *__restrict__ col, const int *__restrict__ tgt, int E, int64_t *__restrict__ ei) {
    int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e < E) { ei[e] = col[e]; ei[(size_t)E + e] = tgt[e]; }
}

}  // namespace

extern "C" {

int conan_abi_version(void) { return 1; }

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
    k_exclusive_scan<<<1, SCAN_THREADS, 0, s>>>(deg_ws, num_atoms, rowptr);
    k_radius<1><<<num_graphs, RG_THREADS, 0, s>>>(pos, graph_ptr, r2, cap, loop, nullptr, rowptr, col, tgt, dist);
    CONAN_LAUNCH_CHECK();
    return CONAN_OK;
}

int conan_csr_transpose(const int *graph_ptr, int num_graphs, int num_atoms, const int *rowptr, const int *col,
                        int *deg_ws, int *t_rowptr, int *t_eid, void *stream) {
    if (!graph_ptr || !rowptr || !col || !deg_ws || !t_rowptr || !t_eid || num_graphs <= 0) return CONAN_E_BADARG;
    hipStream_t s = as_stream(stream);
    k_transpose<0><<<num_graphs, RG_THREADS, 0, s>>>(graph_ptr, rowptr, col, deg_ws, nullptr, nullptr);
    k_exclusive_scan<<<1, SCAN_THREADS, 0, s>>>(deg_ws, num_atoms, t_rowptr);
    k_transpose<1><<<num_graphs, RG_THREADS, 0, s>>>(graph_ptr, rowptr, col, nullptr, t_rowptr, t_eid);
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
