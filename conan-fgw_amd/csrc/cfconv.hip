// CFConv message + aggregation: out[i,:] = sum_{e in CSR row i} x[col[e],:] * W[e,:]   — the HBM-bound kernel of the
// SchNet path (it streams the [E,F] filter tensor once) — and its backward.
//
// Mapping (wave64): one half-wavefront (32 lanes x float4 = 128 channels = one 512-B row) per edge, so a wavefront
// consumes two edges of the same target per step with fully coalesced 16-B/lane loads of the W row and of the gathered
// x row; partial sums stay in registers, the two halves are combined with one cross-lane add, one 512-B store per
// target.  No atomics: the CSR is sorted by target, results are bitwise reproducible.
// Several targets per workgroup keep >= 8 wavefronts per SIMD resident to cover HBM latency.
#include "common.h"

namespace {
// The wavefront's index inside its workgroup, as a value the compiler KNOWS to be wave-uniform: everything derived from it (the target row, its CSR bounds,
// the loop over its edges) then lives in scalar registers and is fetched by scalar loads — threadIdx.x >> 6 alone is a per-lane value to the compiler.
#ifndef CONAN_CFCONV_UNIFORM
#define CONAN_CFCONV_UNIFORM 1      // (A/B switch)
#endif
__device__ __forceinline__ int wave_u() {
#if CONAN_CFCONV_UNIFORM
    return __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
#else
    return (int)(threadIdx.x >> 6);
#endif
}

template <int VEC>   // VEC = F / 32 / 4  (number of float4 per lane per row): F=128 -> 1, F=256 -> 2
__global__ void __launch_bounds__(256) k_cfconv_fwd(const float *__restrict__ x, const float *__restrict__ W,
                                                    const int *__restrict__ rowptr, const int *__restrict__ col, int num_atoms,
                                                    float *__restrict__ out, const int *__restrict__ pid, float *__restrict__ zero_slot) {
    constexpr int F = 128 * VEC;
    // zero_slot: the max |g| word that the backward of THIS gather raises with atomicMax (k_cfconv_bwd_xw128: dx and the pair gradient in one
    // launch, so no kernel of the backward pass runs in front of it) — cleared here, by a kernel of the forward pass that is launched anyway
    if (zero_slot && blockIdx.x == 0 && threadIdx.x == 0) *zero_slot = 0.f;
    const int lane = threadIdx.x & 63, half = lane >> 5, l32 = lane & 31;
    const int lb = xcd_contiguous_block(blockIdx.x, gridDim.x);
    const int per = (num_atoms + (int)gridDim.x - 1) / (int)gridDim.x;          // contiguous targets per workgroup
    const int a_lo = lb * per + wave_u(), a_hi = min(num_atoms, (lb + 1) * per);
    for (int i = a_lo; i < a_hi; i += 4) {
        const int e0 = rowptr[i], e1 = rowptr[i + 1];
        float4 acc[VEC];
#pragma unroll
        for (int v = 0; v < VEC; ++v) acc[v] = make_float4(0.f, 0.f, 0.f, 0.f);
        // The row's source and filter-row indices are fetched once, one edge per lane (a capped row has <= 32 edges), and
        // handed out by cross-lane reads: the W / x row loads of all edges then issue back to back instead of each
        // waiting for its own index load.  Half-wavefront `half` takes the edges of its parity.
        for (int base = e0; base < e1; base += 64) {
            const int cnt = min(64, e1 - base);
            const int my_j = lane < cnt ? col[base + lane] : 0;
            const int my_r = lane < cnt ? (pid ? pid[base + lane] : base + lane) : 0;
            for (int k2 = 0; k2 < cnt; k2 += 8) {             // 4 steps x 2 edges; the tail repeats the last edge with weight 0
                int j[4], r[4];
                float m[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int k = k2 + 2 * u + half, kk = min(k, cnt - 1);
                    j[u] = __shfl(my_j, kk, 64); r[u] = __shfl(my_r, kk, 64);
                    m[u] = k < cnt ? 1.f : 0.f;
                }
#pragma unroll
                for (int v = 0; v < VEC; ++v) {
                    float4 w4[4], x4[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        w4[u] = reinterpret_cast<const float4 *>(W + (size_t)r[u] * F)[l32 + 32 * v];
                        x4[u] = reinterpret_cast<const float4 *>(x + (size_t)j[u] * F)[l32 + 32 * v];
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        acc[v].x += m[u] * x4[u].x * w4[u].x; acc[v].y += m[u] * x4[u].y * w4[u].y;
                        acc[v].z += m[u] * x4[u].z * w4[u].z; acc[v].w += m[u] * x4[u].w * w4[u].w;
                    }
                }
            }
        }
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
            acc[v].x += __shfl_xor(acc[v].x, 32, 64); acc[v].y += __shfl_xor(acc[v].y, 32, 64);
            acc[v].z += __shfl_xor(acc[v].z, 32, 64); acc[v].w += __shfl_xor(acc[v].w, 32, 64);
            if (half == 0) reinterpret_cast<float4 *>(out + (size_t)i * F)[l32 + 32 * v] = acc[v];
        }
    }
}

// generic width (F % 4 == 0, any size): one wavefront per target, lanes stride over float4 columns
__global__ void __launch_bounds__(256) k_cfconv_fwd_generic(const float *__restrict__ x, const float *__restrict__ W,
                                                            const int *__restrict__ rowptr, const int *__restrict__ col, int num_atoms,
                                                            int F, float *__restrict__ out, const int *__restrict__ pid, float *__restrict__ zero_slot) {
    if (zero_slot && blockIdx.x == 0 && threadIdx.x == 0) *zero_slot = 0.f;      // see k_cfconv_fwd
    const int lane = threadIdx.x & 63;
    const int lb = xcd_contiguous_block(blockIdx.x, gridDim.x);
    const int per = (num_atoms + (int)gridDim.x - 1) / (int)gridDim.x;          // contiguous targets per workgroup
    const int a_lo = lb * per + wave_u(), a_hi = min(num_atoms, (lb + 1) * per);
    const int F4 = F >> 2;
    for (int i = a_lo; i < a_hi; i += 4) {
        const int e0 = rowptr[i], e1 = rowptr[i + 1];
        for (int c = lane; c < F4; c += 64) {
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int e = e0; e < e1; ++e) {
                const float4 w = reinterpret_cast<const float4 *>(W + (size_t)(pid ? pid[e] : e) * F)[c];
                const float4 xv = reinterpret_cast<const float4 *>(x + (size_t)col[e] * F)[c];
                acc.x += xv.x * w.x; acc.y += xv.y * w.y; acc.z += xv.z * w.z; acc.w += xv.w * w.w;
            }
            reinterpret_cast<float4 *>(out + (size_t)i * F)[c] = acc;
        }
    }
}

// dx[j,:] = sum over edges e with source j of W[row(e),:] * dout[tgt[e],:]   (by-source CSR: t_rowptr / t_eid).
// F = 128: same mapping as the forward — one half-wavefront (32 lanes x float4) per edge, two edges per step.
__global__ void __launch_bounds__(256) k_cfconv_bwd_x128(const float *__restrict__ W, const float *__restrict__ dout,
                                                         const int *__restrict__ t_rowptr, const int *__restrict__ t_eid,
                                                         const int *__restrict__ tgt, int num_atoms, float *__restrict__ dx,
                                                         const int *__restrict__ pid, float *__restrict__ zero_slot) {
    constexpr int F = 128;
    // zero_slot: the max |g| word that the NEXT kernel of this backward (k_cfconv_bwd_wp128, same stream) raises with atomicMax — cleared here,
    // by a kernel that is launched anyway, instead of by a fill launch of its own (and fresh on every backward pass, captured or not)
    if (zero_slot && blockIdx.x == 0 && threadIdx.x == 0) *zero_slot = 0.f;
    const int lane = threadIdx.x & 63, half = lane >> 5, l32 = lane & 31;
    const int lb = xcd_contiguous_block(blockIdx.x, gridDim.x);
    const int per = (num_atoms + (int)gridDim.x - 1) / (int)gridDim.x;          // contiguous targets per workgroup
    const int a_lo = lb * per + wave_u(), a_hi = min(num_atoms, (lb + 1) * per);
    for (int j = a_lo; j < a_hi; j += 4) {
        const int s0 = t_rowptr[j], s1 = t_rowptr[j + 1];
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int base = s0; base < s1; base += 64) {          // indices one edge per lane, then cross-lane hand-out (see forward)
            const int cnt = min(64, s1 - base);
            const int my_e = lane < cnt ? t_eid[base + lane] : 0;
            const int my_t = lane < cnt ? tgt[my_e] : 0;
            const int my_r = lane < cnt ? (pid ? pid[my_e] : my_e) : 0;
            for (int k2 = 0; k2 < cnt; k2 += 8) {
                float4 w4[4], g4[4];
                float m[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int k = k2 + 2 * u + half, kk = min(k, cnt - 1);
                    const int t = __shfl(my_t, kk, 64), r = __shfl(my_r, kk, 64);
                    m[u] = k < cnt ? 1.f : 0.f;
                    w4[u] = reinterpret_cast<const float4 *>(W + (size_t)r * F)[l32];
                    g4[u] = reinterpret_cast<const float4 *>(dout + (size_t)t * F)[l32];
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    acc.x += m[u] * g4[u].x * w4[u].x; acc.y += m[u] * g4[u].y * w4[u].y;
                    acc.z += m[u] * g4[u].z * w4[u].z; acc.w += m[u] * g4[u].w * w4[u].w;
                }
            }
        }
        acc.x += __shfl_xor(acc.x, 32, 64); acc.y += __shfl_xor(acc.y, 32, 64);
        acc.z += __shfl_xor(acc.z, 32, 64); acc.w += __shfl_xor(acc.w, 32, 64);
        if (half == 0) reinterpret_cast<float4 *>(dx + (size_t)j * F)[l32] = acc;
    }
}

// generic width
__global__ void __launch_bounds__(256) k_cfconv_bwd_x(const float *__restrict__ W, const float *__restrict__ dout,
                                                      const int *__restrict__ t_rowptr, const int *__restrict__ t_eid,
                                                      const int *__restrict__ tgt, int num_atoms, int F, float *__restrict__ dx,
                                                      const int *__restrict__ pid, float *__restrict__ zero_slot) {
    if (zero_slot && blockIdx.x == 0 && threadIdx.x == 0) *zero_slot = 0.f;      // see k_cfconv_bwd_x128
    const int lane = threadIdx.x & 63;
    const int lb = xcd_contiguous_block(blockIdx.x, gridDim.x);
    const int per = (num_atoms + (int)gridDim.x - 1) / (int)gridDim.x;          // contiguous targets per workgroup
    const int a_lo = lb * per + wave_u(), a_hi = min(num_atoms, (lb + 1) * per);
    const int F4 = F >> 2;
    for (int j = a_lo; j < a_hi; j += 4) {
        const int s0 = t_rowptr[j], s1 = t_rowptr[j + 1];
        for (int c = lane; c < F4; c += 64) {
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int s = s0; s < s1; ++s) {
                const int e = t_eid[s];
                const float4 w = reinterpret_cast<const float4 *>(W + (size_t)(pid ? pid[e] : e) * F)[c];
                const float4 g = reinterpret_cast<const float4 *>(dout + (size_t)tgt[e] * F)[c];
                acc.x += g.x * w.x; acc.y += g.y * w.y; acc.z += g.z * w.z; acc.w += g.w * w.w;
            }
            reinterpret_cast<float4 *>(dx + (size_t)j * F)[c] = acc;
        }
    }
}

// dW[e,:] = x[col[e],:] * dout[tgt[e],:] * scale(e)      (scale = optional per-edge factor, e.g. the cosine cutoff)
__global__ void __launch_bounds__(256) k_cfconv_bwd_w(const float *__restrict__ x, const float *__restrict__ dout,
                                                      const int *__restrict__ num_edges_dev, int max_edges,
                                                      const int *__restrict__ col, const int *__restrict__ tgt, int F,
                                                      const float *__restrict__ dist, float cutoff, float *__restrict__ dW) {
    const int E = num_edges_dev ? min(*num_edges_dev, max_edges) : max_edges;
    const int F4 = F >> 2;
    const long long n4 = (long long)E * F4;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        const int e = (int)(i / F4), c = (int)(i - (long long)e * F4);
        const float4 xv = reinterpret_cast<const float4 *>(x + (size_t)col[e] * F)[c];
        const float4 g = reinterpret_cast<const float4 *>(dout + (size_t)tgt[e] * F)[c];
        const float cc = dist ? 0.5f * (cosf(__fdiv_rn(dist[e] * 3.14159265358979323846f, cutoff)) + 1.0f) : 1.0f;
        reinterpret_cast<float4 *>(dW)[i] = make_float4(xv.x * g.x * cc, xv.y * g.y * cc, xv.z * g.z * cc, xv.w * g.w * cc);
    }
}

// pair-level filter gradient: dWp[p,:] = C(d_p) * ( x[src(e0)]*dout[tgt(e0)] + x[src(e1)]*dout[tgt(e1)] )   (e1 = -1: one direction only)
__global__ void __launch_bounds__(256) k_cfconv_bwd_wp(const float *__restrict__ x, const float *__restrict__ dout,
                                                       const int *__restrict__ num_pairs_dev, int max_pairs, const int *__restrict__ pe0,
                                                       const int *__restrict__ pe1, const int *__restrict__ col, const int *__restrict__ tgt,
                                                       int F, const float *__restrict__ pdist, float cutoff, float *__restrict__ dWp) {
    const int P = min(*num_pairs_dev, max_pairs);
    const int F4 = F >> 2;
    const int lb = xcd_contiguous_block(blockIdx.x, gridDim.x);
    const int per = (P + (int)gridDim.x - 1) / (int)gridDim.x;                  // contiguous pairs per workgroup
    const long long i_lo = (long long)lb * per * F4, i_hi = (long long)min(P, (lb + 1) * per) * F4;
    for (long long i = i_lo + threadIdx.x; i < i_hi; i += blockDim.x) {
        const int p = (int)(i / F4), c = (int)(i - (long long)p * F4);
        const int e0 = pe0[p], e1 = pe1[p];
        const float4 xa = reinterpret_cast<const float4 *>(x + (size_t)col[e0] * F)[c];
        const float4 ga = reinterpret_cast<const float4 *>(dout + (size_t)tgt[e0] * F)[c];
        float4 r = make_float4(xa.x * ga.x, xa.y * ga.y, xa.z * ga.z, xa.w * ga.w);
        if (e1 >= 0) {
            const float4 xb = reinterpret_cast<const float4 *>(x + (size_t)col[e1] * F)[c];
            const float4 gb = reinterpret_cast<const float4 *>(dout + (size_t)tgt[e1] * F)[c];
            r.x += xb.x * gb.x; r.y += xb.y * gb.y; r.z += xb.z * gb.z; r.w += xb.w * gb.w;
        }
        const float cc = 0.5f * (cosf(__fdiv_rn(pdist[p] * 3.14159265358979323846f, cutoff)) + 1.0f);
        reinterpret_cast<float4 *>(dWp)[i] = make_float4(r.x * cc, r.y * cc, r.z * cc, r.w * cc);
    }
}

// F = 128: one wavefront per 64 pairs.  Edge ids, endpoints and the cutoff factor are computed one pair per lane (one
// cosine per pair, not per float4), then handed out by cross-lane reads; one half-wavefront (32 lanes x float4) per pair.
__global__ void __launch_bounds__(256) k_cfconv_bwd_wp128(const float *__restrict__ x, const float *__restrict__ dout,
                                                          const int *__restrict__ num_pairs_dev, int max_pairs, const int *__restrict__ pe0,
                                                          const int *__restrict__ pe1, const int *__restrict__ col, const int *__restrict__ tgt,
                                                          const float *__restrict__ pdist, float cutoff, float *__restrict__ dWp,
                                                          unsigned *__restrict__ gmax_bits) {
    constexpr int F = 128;
    const int P = min(*num_pairs_dev, max_pairs);
    const int lane = threadIdx.x & 63, half = lane >> 5, l32 = lane & 31;
    float amax = 0.f;                                             // max |dWp| seen by this thread (gmax_bits: see conan_cfconv_bwd_w_pairs)
    const int lb = xcd_contiguous_block(blockIdx.x, gridDim.x);
    const int per = (P + (int)gridDim.x - 1) / (int)gridDim.x;                  // contiguous pairs per workgroup
    const int p_lo = lb * per, p_hi = min(P, (lb + 1) * per);
    for (int base = p_lo + 64 * wave_u(); base < p_hi; base += 256) {
        const int cnt = min(64, p_hi - base);
        int s0 = 0, t0 = 0, s1 = 0, t1 = 0;
        float cc = 0.f, m1 = 0.f;
        if (lane < cnt) {
            const int e0 = pe0[base + lane], e1 = pe1[base + lane];
            s0 = col[e0]; t0 = tgt[e0];
            if (e1 >= 0) { s1 = col[e1]; t1 = tgt[e1]; m1 = 1.f; }
            cc = 0.5f * (cosf(__fdiv_rn(pdist[base + lane] * 3.14159265358979323846f, cutoff)) + 1.0f);
        }
        for (int k2 = 0; k2 < cnt; k2 += 4) {                 // 2 steps x 2 pairs
            float4 xa[2], ga[2], xb[2], gb[2];
            float c2[2], mm[2];
            int kk[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                kk[u] = min(k2 + 2 * u + half, cnt - 1);
                const int a0 = __shfl(s0, kk[u], 64), b0 = __shfl(t0, kk[u], 64), a1 = __shfl(s1, kk[u], 64), b1 = __shfl(t1, kk[u], 64);
                c2[u] = __shfl(cc, kk[u], 64); mm[u] = __shfl(m1, kk[u], 64);
                xa[u] = reinterpret_cast<const float4 *>(x + (size_t)a0 * F)[l32];
                ga[u] = reinterpret_cast<const float4 *>(dout + (size_t)b0 * F)[l32];
                xb[u] = reinterpret_cast<const float4 *>(x + (size_t)a1 * F)[l32];
                gb[u] = reinterpret_cast<const float4 *>(dout + (size_t)b1 * F)[l32];
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                if (k2 + 2 * u + half >= cnt) continue;
                float4 r;
                r.x = (xa[u].x * ga[u].x + mm[u] * xb[u].x * gb[u].x) * c2[u];
                r.y = (xa[u].y * ga[u].y + mm[u] * xb[u].y * gb[u].y) * c2[u];
                r.z = (xa[u].z * ga[u].z + mm[u] * xb[u].z * gb[u].z) * c2[u];
                r.w = (xa[u].w * ga[u].w + mm[u] * xb[u].w * gb[u].w) * c2[u];
                reinterpret_cast<float4 *>(dWp + (size_t)(base + kk[u]) * F)[l32] = r;
                amax = fmaxf(fmaxf(amax, fmaxf(fabsf(r.x), fabsf(r.y))), fmaxf(fabsf(r.z), fabsf(r.w)));
            }
        }
    }
    if (gmax_bits) {
        // non-negative floats order like their bit patterns: one atomicMax per wavefront, and only when it can raise the value (a plain
        // read first: after the first few workgroups almost nobody has to touch the word)
        amax = wave_max(amax);
        const unsigned bits = __float_as_uint(amax);
        if (lane == 0 && bits > *reinterpret_cast<volatile unsigned *>(gmax_bits) && bits < 0x7f800000u) atomicMax(gmax_bits, bits);
    }
}

// dx AND the pair gradient in one pass over the by-source CSR (round 6).  The two kernels above walk the same neighbourhoods: k_cfconv_bwd_x128 gathers
// W[pair] and dout[target] for every edge of a source j, k_cfconv_bwd_wp128 gathers x and dout of both endpoints of every pair.  A pair belongs to
// the edge that is its e0 (conan_pair_list: the direction with source <= target, or the only direction the neighbour cap left): while source j walks
// its edges, an edge that owns its pair has x[j], dout[j] (this wavefront's own rows) and dout[target] (gathered for dx anyway) at hand — the pair
// row g = C(d) (x[j] dout[t] + [reverse edge exists] x[t] dout[j]) needs ONE more row, x[t].  Per pair that is 1 extra row gather instead of 4 and one
// launch less on the backward chain; the expression and its order of operations are k_cfconv_bwd_wp128's.  max |g| is raised in gmax_bits, which
// the forward gather of the same CFConv cleared (conan_cfconv_fwd's zero_slot).
template <int U, bool ANYX>
__global__ void __launch_bounds__(256) k_cfconv_bwd_xw128(const float *__restrict__ W, const float *__restrict__ x, const float *__restrict__ dout,
                                                          const int *__restrict__ t_rowptr, const int *__restrict__ t_eid,
                                                          const int *__restrict__ tgt, const int *__restrict__ pid, const int *__restrict__ pe0,
                                                          const int *__restrict__ pe1, const float *__restrict__ pdist, float cutoff,
                                                          int num_atoms, float *__restrict__ dx, float *__restrict__ dWp,
                                                          unsigned *__restrict__ gmax_bits) {
    constexpr int F = 128;
    const int lane = threadIdx.x & 63, half = lane >> 5, l32 = lane & 31;
    const int lb = xcd_contiguous_block(blockIdx.x, gridDim.x);
    const int per = (num_atoms + (int)gridDim.x - 1) / (int)gridDim.x;          // contiguous sources per workgroup
    const int a_lo = lb * per + wave_u(), a_hi = min(num_atoms, (lb + 1) * per);
    float amax = 0.f;
    for (int j = a_lo; j < a_hi; j += 4) {
        const int s0 = t_rowptr[j], s1 = t_rowptr[j + 1];
        const float4 xs = reinterpret_cast<const float4 *>(x + (size_t)j * F)[l32];          // this source's own rows
        const float4 ds = reinterpret_cast<const float4 *>(dout + (size_t)j * F)[l32];
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int base = s0; base < s1; base += 64) {          // indices one edge per lane, then cross-lane hand-out (see the forward)
            const int cnt = min(64, s1 - base);
            int my_t = 0, my_r = 0, my_own = 0;
            float my_m1 = 0.f, my_cc = 0.f;
            if (lane < cnt) {
                const int e = t_eid[base + lane];
                my_t = tgt[e];
                my_r = pid[e];
                my_own = pe0[my_r] == e ? 1 : 0;
                if (my_own) {
                    my_m1 = pe1[my_r] >= 0 ? 1.f : 0.f;
                    my_cc = 0.5f * (cosf(__fdiv_rn(pdist[my_r] * 3.14159265358979323846f, cutoff)) + 1.0f);
                }
            }
            for (int k2 = 0; k2 < cnt; k2 += 2 * U) {
                float4 w4[U], g4[U], x4[U];
                float m[U], mm[U], cc[U];
                int rr[U], own[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int k = k2 + 2 * u + half, kk = min(k, cnt - 1);
                    const int t = __shfl(my_t, kk, 64);
                    rr[u] = __shfl(my_r, kk, 64);
                    w4[u] = reinterpret_cast<const float4 *>(W + (size_t)rr[u] * F)[l32];
                    g4[u] = reinterpret_cast<const float4 *>(dout + (size_t)t * F)[l32];
                    const int ow = __shfl(my_own, kk, 64);            // (never inside a conditional arm: a cross-lane read under half an EXEC mask sees a switched-off lane as 0)
                    own[u] = k < cnt ? ow : 0;
                    mm[u] = __shfl(my_m1, kk, 64); cc[u] = __shfl(my_cc, kk, 64);
                    m[u] = k < cnt ? 1.f : 0.f;
                    // ANYX: x[t] for every edge (its address does not wait for the pair table; an edge that does not own its pair wastes an L2 row read);
                    // else: such an edge re-reads the row this wavefront holds
                    x4[u] = reinterpret_cast<const float4 *>(x + (size_t)((ANYX || own[u]) ? t : j) * F)[l32];
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    acc.x += m[u] * g4[u].x * w4[u].x; acc.y += m[u] * g4[u].y * w4[u].y;
                    acc.z += m[u] * g4[u].z * w4[u].z; acc.w += m[u] * g4[u].w * w4[u].w;
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    if (!own[u]) continue;
                    float4 r;
                    r.x = (xs.x * g4[u].x + mm[u] * x4[u].x * ds.x) * cc[u];
                    r.y = (xs.y * g4[u].y + mm[u] * x4[u].y * ds.y) * cc[u];
                    r.z = (xs.z * g4[u].z + mm[u] * x4[u].z * ds.z) * cc[u];
                    r.w = (xs.w * g4[u].w + mm[u] * x4[u].w * ds.w) * cc[u];
                    reinterpret_cast<float4 *>(dWp + (size_t)rr[u] * F)[l32] = r;
                    amax = fmaxf(fmaxf(amax, fmaxf(fabsf(r.x), fabsf(r.y))), fmaxf(fabsf(r.z), fabsf(r.w)));
                }
            }
        }
        acc.x += __shfl_xor(acc.x, 32, 64); acc.y += __shfl_xor(acc.y, 32, 64);
        acc.z += __shfl_xor(acc.z, 32, 64); acc.w += __shfl_xor(acc.w, 32, 64);
        if (half == 0) reinterpret_cast<float4 *>(dx + (size_t)j * F)[l32] = acc;
    }
    if (gmax_bits) {
        amax = wave_max(amax);
        const unsigned bits = __float_as_uint(amax);
        if (lane == 0 && bits > *reinterpret_cast<volatile unsigned *>(gmax_bits) && bits < 0x7f800000u) atomicMax(gmax_bits, bits);
    }
}

}  // namespace

extern "C" {

int conan_cfconv_fwd(const float *x, const float *W, const int *rowptr, const int *col, const int *pid, int num_atoms,
                     int num_filters, float *out, float *zero_slot, void *stream) {
    if (!x || !W || !rowptr || !col || !out || num_atoms < 0 || num_filters <= 0 || (num_filters & 3)) return CONAN_E_BADARG;
    if (num_atoms == 0) {
        if (zero_slot && hipMemsetAsync(zero_slot, 0, sizeof(float), as_stream(stream)) != hipSuccess) return CONAN_E_LAUNCH;
        return CONAN_OK;
    }
    hipStream_t s = as_stream(stream);
    int blocks = (num_atoms + 3) / 4;                 // 4 wavefronts (targets) per 256-thread workgroup
    if (blocks > 65536) blocks = 65536;
    blocks = round_up8(blocks);
    if (num_filters == 128) k_cfconv_fwd<1><<<blocks, 256, 0, s>>>(x, W, rowptr, col, num_atoms, out, pid, zero_slot);
    else if (num_filters == 256) k_cfconv_fwd<2><<<blocks, 256, 0, s>>>(x, W, rowptr, col, num_atoms, out, pid, zero_slot);
    else k_cfconv_fwd_generic<<<blocks, 256, 0, s>>>(x, W, rowptr, col, num_atoms, num_filters, out, pid, zero_slot);
    CONAN_LAUNCH_CHECK();
    return CONAN_OK;
}

int conan_cfconv_bwd_x(const float *W, const float *dout, const int *t_rowptr, const int *t_eid, const int *tgt,
                       const int *pid, int num_atoms, int num_filters, float *dx, float *zero_slot, void *stream) {
    if (!W || !dout || !t_rowptr || !t_eid || !tgt || !dx || num_atoms < 0 || num_filters <= 0 || (num_filters & 3)) return CONAN_E_BADARG;
    if (num_atoms == 0) {
        if (zero_slot && hipMemsetAsync(zero_slot, 0, sizeof(float), as_stream(stream)) != hipSuccess) return CONAN_E_LAUNCH;
        return CONAN_OK;
    }
    int blocks = (num_atoms + 3) / 4;
    if (blocks > 65536) blocks = 65536;
    blocks = round_up8(blocks);
    if (num_filters == 128) k_cfconv_bwd_x128<<<blocks, 256, 0, as_stream(stream)>>>(W, dout, t_rowptr, t_eid, tgt, num_atoms, dx, pid, zero_slot);
    else k_cfconv_bwd_x<<<blocks, 256, 0, as_stream(stream)>>>(W, dout, t_rowptr, t_eid, tgt, num_atoms, num_filters, dx, pid, zero_slot);
    CONAN_LAUNCH_CHECK();
    return CONAN_OK;
}

int conan_cfconv_bwd_w(const float *x, const float *dout, const int *num_edges_dev, int max_edges, const int *col,
                       const int *tgt, int num_filters, const float *dist, float cutoff, float *dW, void *stream) {
    if (!x || !dout || !col || !tgt || !dW || max_edges < 0 || num_filters <= 0 || (num_filters & 3)) return CONAN_E_BADARG;
    if (max_edges == 0) return CONAN_OK;
    k_cfconv_bwd_w<<<4096, 256, 0, as_stream(stream)>>>(x, dout, num_edges_dev, max_edges, col, tgt, num_filters, dist, cutoff, dW);
    CONAN_LAUNCH_CHECK();
    return CONAN_OK;
}

int conan_cfconv_bwd_w_pairs(const float *x, const float *dout, const int *num_pairs_dev, int max_pairs, const int *pair_e0,
                             const int *pair_e1, const int *col, const int *tgt, int num_filters, const float *pair_dist,
                             float cutoff, float *dWp, float *gmax, void *stream) {
    if (!x || !dout || !num_pairs_dev || !pair_e0 || !pair_e1 || !col || !tgt || !pair_dist || !dWp || max_pairs < 0 || num_filters <= 0 ||
        (num_filters & 3))
        return CONAN_E_BADARG;
    if (max_pairs == 0) return CONAN_OK;
    if (gmax && num_filters != 128) return CONAN_E_UNSUPPORTED;
    if (num_filters == 128)
        k_cfconv_bwd_wp128<<<4096, 256, 0, as_stream(stream)>>>(x, dout, num_pairs_dev, max_pairs, pair_e0, pair_e1, col, tgt, pair_dist, cutoff, dWp,
                                                                reinterpret_cast<unsigned *>(gmax));
    else
        k_cfconv_bwd_wp<<<4096, 256, 0, as_stream(stream)>>>(x, dout, num_pairs_dev, max_pairs, pair_e0, pair_e1, col, tgt, num_filters, pair_dist,
                                                              cutoff, dWp);
    CONAN_LAUNCH_CHECK();
    return CONAN_OK;
}

int conan_cfconv_bwd_xw_pairs_supported(int num_filters) { return num_filters == 128; }

int conan_cfconv_bwd_xw_pairs(const float *W, const float *x, const float *dout, const int *t_rowptr, const int *t_eid, const int *tgt, const int *pid,
                              const int *pair_e0, const int *pair_e1, const float *pair_dist, float cutoff, int num_atoms, int num_filters, float *dx,
                              float *dWp, float *gmax, void *stream) {
    if (!W || !x || !dout || !t_rowptr || !t_eid || !tgt || !pid || !pair_e0 || !pair_e1 || !pair_dist || !dx || !dWp || num_atoms < 0) return CONAN_E_BADARG;
    if (!conan_cfconv_bwd_xw_pairs_supported(num_filters)) return CONAN_E_UNSUPPORTED;
    if (num_atoms == 0) return CONAN_OK;
    int blocks = (num_atoms + 3) / 4;
    if (blocks > 65536) blocks = 65536;
    blocks = round_up8(blocks);
    // U = 2 edge steps per trip, x[target] only for edges that own their pair: the forms U = 1 / 2 / 4 with and without the unconditional x gather measured
    // within 4 % of each other (profiles/r6_ab_cfconv_bwd_one_launch.txt: the launch sits on its W read and g write streams, not on its occupancy)
    k_cfconv_bwd_xw128<2, false><<<blocks, 256, 0, as_stream(stream)>>>(W, x, dout, t_rowptr, t_eid, tgt, pid, pair_e0, pair_e1, pair_dist, cutoff, num_atoms, dx, dWp,
                                                                        reinterpret_cast<unsigned *>(gmax));
    CONAN_LAUNCH_CHECK();
    return CONAN_OK;
}

}  // extern "C"
