// Stage-2 aggregation head in one launch each way (schnet_based_models.py:163-171 of the reference, `EmbeddingsWithGATAggregationBaryCenter`):
//
//     x   = Lin3d(x3) + xc + agg_weight * Linbary(xb)          [G, D]   (xc = Lin_cov(GAT) arrives transformed from the side stream)
//     out = Linreg( mean over the K conformers of a molecule of x )    [G/K, 1]
//
// Everything is linear, so the conformer mean is taken FIRST (three D-vectors per molecule) and the two D x D layers act on the means:
// out_b = wreg . (W3 m3_b + b3 + mc_b + aw (Wb mb_b + bb)) + breg.  As separate launches this was 8 kernels forward and 9 backward of
// 5-15 us each on 1 280 x 64 operands — pure launch cost in a 3.2 ms step.  D <= 64 (one lane per channel), one wavefront per molecule.
// Backward: blocks [0, nmol/4) write the input gradients (broadcast over K), the remaining blocks form the parameter gradients with a
// fixed-order loop over the molecules (bitwise reproducible, no atomics): 2 x D/4 blocks of 4 rows of dW3 / dWb, one block for the vectors.
#include "common.h"

namespace {

constexpr int HD_THREADS = 256, HD_WAVES = 4, HD_MAXD = 64, HD_PITCH = HD_MAXD + 1, HD_U = 16;

__global__ void __launch_bounds__(HD_THREADS) k_stage2_head_fwd(const float *__restrict__ x3, const float *__restrict__ xc, const float *__restrict__ xb,
                                                                const float *__restrict__ W3, const float *__restrict__ b3,
                                                                const float *__restrict__ Wb, const float *__restrict__ bb,
                                                                const float *__restrict__ wreg, const float *__restrict__ breg, float aw, int nmol, int K,
                                                                int D, float *__restrict__ out, float *__restrict__ m3s, float *__restrict__ mbs,
                                                                float *__restrict__ ts) {
    __shared__ float W3L[HD_MAXD * HD_PITCH], WbL[HD_MAXD * HD_PITCH], vec[HD_WAVES][2][HD_MAXD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int t = tid; t < D * D; t += HD_THREADS) { const int c = t / D, j = t - c * D; W3L[c * HD_PITCH + j] = W3[t]; WbL[c * HD_PITCH + j] = Wb[t]; }
    const int b = blockIdx.x * HD_WAVES + wave;
    const bool on = b < nmol && lane < D;
    float m3 = 0.f, mc = 0.f, mb = 0.f;
    if (on) {
        const size_t base = (size_t)b * K * D + lane;
        for (int k = 0; k < K; ++k) { m3 += x3[base + (size_t)k * D]; mc += xc[base + (size_t)k * D]; mb += xb[base + (size_t)k * D]; }
        const float inv = 1.0f / (float)K;
        m3 *= inv; mc *= inv; mb *= inv;
    }
    vec[wave][0][lane] = m3; vec[wave][1][lane] = mb;
    __syncthreads();
    float t = 0.f, o = 0.f;
    if (on) {
        float a = b3[lane], c = bb[lane];
        for (int j = 0; j < D; ++j) { a += W3L[lane * HD_PITCH + j] * vec[wave][0][j]; c += WbL[lane * HD_PITCH + j] * vec[wave][1][j]; }
        t = a + mc + aw * c;
        o = wreg[lane] * t;
        m3s[(size_t)b * D + lane] = m3; mbs[(size_t)b * D + lane] = mb; ts[(size_t)b * D + lane] = t;
    }
    o = wave_sum(o);
    if (b < nmol && lane == 0) out[b] = o + breg[0];
}

__global__ void __launch_bounds__(HD_THREADS) k_stage2_head_bwd(const float *__restrict__ dout, const float *__restrict__ W3, const float *__restrict__ Wb,
                                                                const float *__restrict__ wreg, const float *__restrict__ m3s,
                                                                const float *__restrict__ mbs, const float *__restrict__ ts, float aw, int nmol, int K, int D,
                                                                float *__restrict__ dx3, float *__restrict__ dxc, float *__restrict__ dxb,
                                                                float *__restrict__ dW3, float *__restrict__ db3, float *__restrict__ dWb,
                                                                float *__restrict__ dbb, float *__restrict__ dwreg, float *__restrict__ dbreg) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nblk_x = (nmol + HD_WAVES - 1) / HD_WAVES, rows4 = (D + 3) / 4;
    if ((int)blockIdx.x < nblk_x) {
        // ---- input gradients of the molecules of this block: u = dout_b * wreg, then u W3 / u / aw u Wb, all divided by K
        __shared__ float W3L[HD_MAXD * HD_PITCH], WbL[HD_MAXD * HD_PITCH], u[HD_WAVES][HD_MAXD];
        for (int t = tid; t < D * D; t += HD_THREADS) { const int c = t / D, j = t - c * D; W3L[c * HD_PITCH + j] = W3[t]; WbL[c * HD_PITCH + j] = Wb[t]; }
        const int b = blockIdx.x * HD_WAVES + wave;
        const bool on = b < nmol && lane < D;
        const float uo = on ? dout[b] * wreg[lane] : 0.f;
        u[wave][lane] = uo;
        __syncthreads();
        if (on) {
            float g3 = 0.f, gb = 0.f;
            for (int c = 0; c < D; ++c) { g3 += u[wave][c] * W3L[c * HD_PITCH + lane]; gb += u[wave][c] * WbL[c * HD_PITCH + lane]; }
            const float inv = 1.0f / (float)K;
            g3 *= inv; gb *= aw * inv;
            const float gc = uo * inv;
            const size_t base = (size_t)b * K * D + lane;
            for (int k = 0; k < K; ++k) { dx3[base + (size_t)k * D] = g3; dxc[base + (size_t)k * D] = gc; dxb[base + (size_t)k * D] = gb; }
        }
        return;
    }
    const int pb = blockIdx.x - nblk_x;
    if (pb < 2 * rows4) {
        // ---- 4 rows of dW3 (pb < rows4) or dWb: dW[c][j] = sum_b dout_b wreg[c] m_b[j] (x aw for the barycenter layer), molecules in order
        const bool bary = pb >= rows4;
        const int c = 4 * (bary ? pb - rows4 : pb) + wave, j = lane;
        if (c >= D || j >= D) return;
        const float *ms = bary ? mbs : m3s;
        const float wc = wreg[c];
        float acc = 0.f;
        int b = 0;
        for (; b + HD_U <= nmol; b += HD_U) {                       // HD_U loads in flight, summed in molecule order (a serial loop is one L2 round trip per molecule)
            float dv[HD_U], mv[HD_U];
#pragma unroll
            for (int q = 0; q < HD_U; ++q) { dv[q] = dout[b + q]; mv[q] = ms[(size_t)(b + q) * D + j]; }
#pragma unroll
            for (int q = 0; q < HD_U; ++q) acc += (dv[q] * wc) * mv[q];
        }
        for (; b < nmol; ++b) acc += (dout[b] * wc) * ms[(size_t)b * D + j];
        (bary ? dWb : dW3)[c * D + j] = bary ? aw * acc : acc;
        return;
    }
    // ---- vectors: db3 = sum_b u_b, dbb = aw db3, dwreg = sum_b dout_b t_b, dbreg = sum_b dout_b   (wave 0: lane = channel)
    if (wave == 0) {
        float su = 0.f, sw = 0.f, sd = 0.f;
        if (lane < D) {
            const float wc = wreg[lane];
            int b = 0;
            for (; b + HD_U <= nmol; b += HD_U) {
                float dv[HD_U], tv[HD_U];
#pragma unroll
                for (int q = 0; q < HD_U; ++q) { dv[q] = dout[b + q]; tv[q] = ts[(size_t)(b + q) * D + lane]; }
#pragma unroll
                for (int q = 0; q < HD_U; ++q) { su += dv[q] * wc; sw += dv[q] * tv[q]; sd += dv[q]; }
            }
            for (; b < nmol; ++b) { const float d = dout[b]; su += d * wc; sw += d * ts[(size_t)b * D + lane]; sd += d; }
            db3[lane] = su; dbb[lane] = aw * su; dwreg[lane] = sw;
        }
        if (lane == 0) dbreg[0] = sd;
    }
}

// Mean-squared-error loss of the stage-2 step (the reference's criterion for regression, common.py: nn.MSELoss) and its gradient in ONE launch:
// loss = mean((pred - target)^2), dpred = 2 (pred - target) / n.  One workgroup; the sum runs in a fixed order (bitwise reproducible).
// torch's mse_loss + backward are five launches of ~4.7 us each on the critical path between the forward and the backward of the step.
__global__ void __launch_bounds__(256) k_mse_loss(const float *__restrict__ pred, const float *__restrict__ target, int n, float *__restrict__ loss,
                                                  float *__restrict__ dpred) {
    __shared__ float red[4];
    const int tid = threadIdx.x;
    const float inv = 1.0f / (float)n;
    float acc = 0.f;
    for (int i = tid; i < n; i += 256) {
        const float d = pred[i] - target[i];
        acc += d * d;
        dpred[i] = 2.0f * d * inv;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if ((tid & 63) == 0) red[tid >> 6] = acc;
    __syncthreads();
    if (tid == 0) loss[0] = (((red[0] + red[1]) + red[2]) + red[3]) * inv;
}


// Adam (torch.optim.Adam, no amsgrad / maximize) over ONE flat parameter buffer and its flat gradient, moment and step buffers (round 5): the step of
// the reference's optimiser (train_val.py) for the whole model in one launch instead of torch's multi-tensor kernels (a step-counter foreach add
// + two fused-Adam launches over 100 parameter tensors: 5.7 + 15.8 + 7.0 us at cfg2 for 1.09 MB of parameters).  `step` lives on the device
// (captured graphs replay the launch): every workgroup reads it first, the workgroup that finishes last advances it.
//   m = b1 m + (1 - b1) g;  v = b2 v + (1 - b2) g^2;  p -= lr / (1 - b1^t) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)      (+ weight_decay * p added to g first)
__global__ void __launch_bounds__(256) k_adam_flat(float *__restrict__ p, const float *__restrict__ g, float *__restrict__ m, float *__restrict__ v,
                                                   float *__restrict__ step, unsigned *__restrict__ ticket, long long n, double lr_d, double b1_d, double b2_d,
                                                   float eps, float wd, const double *__restrict__ lr_dev) {
    // the bias corrections 1 - b^t cancel badly in fp32 for the first steps (1 - 0.999^2 keeps three digits), and 0.999 itself is 1.3e-5 of
    // (1 - b2) away from its fp32 rounding: they are formed in fp64 from the fp64 hyper-parameters, like torch's host-side (non-capturable) path
    __shared__ float sh[4];
    const float t = *step + 1.0f;
    if (threadIdx.x == 0) {
        // the learning rate of a captured launch is read where a scheduler can change it between replays (a by-value argument is baked into the graph)
        if (lr_dev) lr_d = *lr_dev;
        const double bc1d = 1.0 - pow(b1_d, (double)t), bc2d = 1.0 - pow(b2_d, (double)t);
        sh[0] = (float)(lr_d / bc1d); sh[1] = (float)sqrt(bc2d); sh[2] = (float)(1.0 - b1_d); sh[3] = (float)(1.0 - b2_d);
    }
    __syncthreads();
    const float step_size = sh[0], sqrt_bc2 = sh[1], omb1 = sh[2], omb2 = sh[3];
    const float b2 = (float)b2_d;
    const long long n4 = n >> 2;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        float4 pv = reinterpret_cast<float4 *>(p)[i];
        const float4 gv = reinterpret_cast<const float4 *>(g)[i];
        float4 mv = reinterpret_cast<float4 *>(m)[i], vv = reinterpret_cast<float4 *>(v)[i];
        float *pp = &pv.x, *mp = &mv.x, *vp = &vv.x;
        const float *gp = &gv.x;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float gg = gp[e] + wd * pp[e];
            mp[e] = mp[e] + omb1 * (gg - mp[e]);                         // lerp(m, g, 1 - b1), as torch's kernels form it
            vp[e] = b2 * vp[e] + omb2 * gg * gg;
            pp[e] -= step_size * mp[e] / (sqrtf(vp[e]) / sqrt_bc2 + eps);
        }
        reinterpret_cast<float4 *>(p)[i] = pv; reinterpret_cast<float4 *>(m)[i] = mv; reinterpret_cast<float4 *>(v)[i] = vv;
    }
    if (blockIdx.x == 0)
        for (long long i = (n4 << 2) + threadIdx.x; i < n; i += blockDim.x) {      // tail (n not a multiple of 4)
            const float gg = g[i] + wd * p[i];
            const float mm = m[i] + omb1 * (gg - m[i]), vv = b2 * v[i] + omb2 * gg * gg;
            m[i] = mm; v[i] = vv; p[i] -= step_size * mm / (sqrtf(vv) / sqrt_bc2 + eps);
        }
    __syncthreads();                                                    // every thread of this workgroup has read *step
    if (threadIdx.x == 0) {
        const unsigned done = atomicAdd(ticket, 1u);
        if (done == gridDim.x - 1) { *step = t; *ticket = 0u; }         // the last workgroup to finish: everyone has read the old value
    }
}
// Global-norm gradient clipping over the flat gradient buffer (torch.nn.utils.clip_grad_norm_, which Lightning runs for the reference's
// Trainer(gradient_clip_val=1.0), trainer.py:177): squared sums per workgroup in fp64 into fixed slots, the last workgroup to arrive adds the slots in
// index order (same result on every run), writes total_norm and the coefficient min(1, max_norm / (total_norm + 1e-6)); a second launch scales in place.
__global__ void __launch_bounds__(256) k_grad_sqnorm(const float *__restrict__ g, long long n, double max_norm, double *__restrict__ part,
                                                     unsigned *__restrict__ ticket, float *__restrict__ out2) {
    __shared__ double red[4];
    __shared__ bool last;
    const long long n4 = n >> 2;
    const long long stride = (long long)gridDim.x * blockDim.x;
    double acc = 0.0;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        const float4 v = reinterpret_cast<const float4 *>(g)[i];
        acc += (double)v.x * v.x + (double)v.y * v.y + (double)v.z * v.z + (double)v.w * v.w;
    }
    if (blockIdx.x == 0)
        for (long long i = (n4 << 2) + threadIdx.x; i < n; i += blockDim.x) acc += (double)g[i] * g[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        // hand-off of ONE 8-byte partial per workgroup to the last arriver: a write-through (sc1) store drained before the ticket add, read back with
        // sc1 loads after the add has returned — no agent-scope release fence (an L2 write-back per workgroup: the first form of this kernel took 15 us)
        __hip_atomic_store(&part[blockIdx.x], ((red[0] + red[1]) + red[2]) + red[3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        last = atomicAdd(ticket, 1u) == gridDim.x - 1;
    }
    __syncthreads();
    if (last) {                                     // (workgroup-uniform) the partials: one per thread, summed in a fixed tree — the same bits on every run
        double v = threadIdx.x < gridDim.x ? __hip_atomic_load(&part[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        __syncthreads();
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
        __syncthreads();
        if (threadIdx.x == 0) {
            const double nrm = sqrt(((red[0] + red[1]) + red[2]) + red[3]);
            double c = max_norm / (nrm + 1e-6);
            if (!(c < 1.0)) c = 1.0;                    // clamp(max=1)
            if (nrm != nrm) c = nrm;                    // torch with error_if_nonfinite=False: a NaN norm gives a NaN coefficient (an infinite one gives 0)
            out2[0] = (float)nrm; out2[1] = (float)c;
            *ticket = 0u;
        }
    }
}
__global__ void __launch_bounds__(256) k_grad_scale_dev(float *__restrict__ g, long long n, const float *__restrict__ out2) {
    const float c = out2[1];
    if (c == 1.0f) return;
    const long long n4 = n >> 2;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        float4 v = reinterpret_cast<float4 *>(g)[i];
        v.x *= c; v.y *= c; v.z *= c; v.w *= c;
        reinterpret_cast<float4 *>(g)[i] = v;
    }
    if (blockIdx.x == 0)
        for (long long i = (n4 << 2) + threadIdx.x; i < n; i += blockDim.x) g[i] *= c;
}
}  // namespace

extern "C" {

int conan_grad_clip_flat(float *grads, long long n, double max_norm, float *norm_coef_dev, double *partials_dev, unsigned *ticket_dev, void *stream) {
    if (!grads || !norm_coef_dev || !partials_dev || !ticket_dev || n < 0 || !(max_norm > 0.0)) return CONAN_E_BADARG;
    if ((uintptr_t)grads & 15) return CONAN_E_BADARG;
    long long blocks = ((n >> 2) + 255) / 256;      // one float4 per thread up to 256 workgroups (= one partial per thread of the last arriver)
    if (blocks < 1) blocks = 1;
    if (blocks > CONAN_GRAD_CLIP_MAX_BLOCKS) blocks = CONAN_GRAD_CLIP_MAX_BLOCKS;
    k_grad_sqnorm<<<(int)blocks, 256, 0, as_stream(stream)>>>(grads, n, max_norm, partials_dev, ticket_dev, norm_coef_dev);
    CONAN_LAUNCH_CHECK();
    k_grad_scale_dev<<<(int)blocks, 256, 0, as_stream(stream)>>>(grads, n, norm_coef_dev);
    CONAN_LAUNCH_CHECK();
    return CONAN_OK;
}

int conan_adam_flat_step(float *params, const float *grads, float *exp_avg, float *exp_avg_sq, float *step_dev, unsigned *ticket_dev, long long n,
                         double lr, double beta1, double beta2, double eps, double weight_decay, const double *lr_dev, void *stream) {
    if (!params || !grads || !exp_avg || !exp_avg_sq || !step_dev || !ticket_dev || n < 0) return CONAN_E_BADARG;
    if (((uintptr_t)params | (uintptr_t)grads | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) & 15) return CONAN_E_BADARG;      // float4 access
    if (n == 0) return CONAN_OK;
    long long blocks = ((n >> 2) + 255) / 256;
    if (blocks < 1) blocks = 1;
    if (blocks > 1024) blocks = 1024;
    k_adam_flat<<<(int)blocks, 256, 0, as_stream(stream)>>>(params, grads, exp_avg, exp_avg_sq, step_dev, ticket_dev, n, lr, beta1, beta2, (float)eps, (float)weight_decay, lr_dev);
    CONAN_LAUNCH_CHECK();
    return CONAN_OK;
}

int conan_mse_loss_fwd(const float *pred, const float *target, int n, float *loss, float *dpred, void *stream) {
    if (!pred || !target || !loss || !dpred || n <= 0) return CONAN_E_BADARG;
    k_mse_loss<<<1, 256, 0, as_stream(stream)>>>(pred, target, n, loss, dpred);
    CONAN_LAUNCH_CHECK();
    return CONAN_OK;
}

int conan_stage2_head_supported(int D) { return (D >= 1 && D <= HD_MAXD) ? 1 : 0; }

int conan_stage2_head_fwd(const float *x3, const float *xc, const float *xb, const float *W3, const float *b3, const float *Wb, const float *bb,
                          const float *wreg, const float *breg, float agg_weight, int num_molecules, int K, int D, float *out, float *m3, float *mb,
                          float *t, void *stream) {
    if (!x3 || !xc || !xb || !W3 || !b3 || !Wb || !bb || !wreg || !breg || !out || !m3 || !mb || !t || num_molecules < 0 || K <= 0) return CONAN_E_BADARG;
    if (!conan_stage2_head_supported(D)) return CONAN_E_UNSUPPORTED;
    if (num_molecules == 0) return CONAN_OK;
    k_stage2_head_fwd<<<(num_molecules + HD_WAVES - 1) / HD_WAVES, HD_THREADS, 0, as_stream(stream)>>>(x3, xc, xb, W3, b3, Wb, bb, wreg, breg, agg_weight,
                                                                                                       num_molecules, K, D, out, m3, mb, t);
    CONAN_LAUNCH_CHECK();
    return CONAN_OK;
}

int conan_stage2_head_bwd(const float *dout, const float *W3, const float *Wb, const float *wreg, const float *m3, const float *mb, const float *t,
                          float agg_weight, int num_molecules, int K, int D, float *dx3, float *dxc, float *dxb, float *dW3, float *db3, float *dWb,
                          float *dbb, float *dwreg, float *dbreg, void *stream) {
    if (!dout || !W3 || !Wb || !wreg || !m3 || !mb || !t || !dx3 || !dxc || !dxb || !dW3 || !db3 || !dWb || !dbb || !dwreg || !dbreg ||
        num_molecules <= 0 || K <= 0)
        return CONAN_E_BADARG;
    if (!conan_stage2_head_supported(D)) return CONAN_E_UNSUPPORTED;
    const int blocks = (num_molecules + HD_WAVES - 1) / HD_WAVES + 2 * ((D + 3) / 4) + 1;
    k_stage2_head_bwd<<<blocks, HD_THREADS, 0, as_stream(stream)>>>(dout, W3, Wb, wreg, m3, mb, t, agg_weight, num_molecules, K, D, dx3, dxc, dxb, dW3,
                                                                    db3, dWb, dbb, dwreg, dbreg);
    CONAN_LAUNCH_CHECK();
    return CONAN_OK;
}

}  // extern "C"
