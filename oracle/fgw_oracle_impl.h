/*
 * TEST INFRASTRUCTURE — NOT PRODUCT CODE.
 *
 * CPU restatement (plain C, scalar, single thread) of the reference's Fused-Gromov-Wasserstein
 * barycenter for the production path.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this.  The HIP product path never calls into it.
 *
 * This file is a "template": fgw_oracle.c includes it twice, once with REAL=float (suffix _f32)
 * and once with REAL=double (suffix _f64).  The f64 build is the "ref64" yard-stick of
 * SURVEY.md Appendix F (the reference code itself is dtype-agnostic).
 *
 * Each function cites the reference file:line it restates (paths relative to the reference
 * repository root, conan_fgw/src/model/fgw/...).
 *
 * Parity pin: tests/test_oracle_fgw.py checks this against
 *   - notebooks/data/cfm_log.pt re-exported as tests/golden/cfm_log.npz (the reference's only
 *     known-answer fixture: stored F_bary/C_bary/err_feature),
 *   - tests/golden/fgw_ref_*.npz, produced by tests/golden/make_fgw_golden.py importing the
 *     reference's own fgw_barycenters in fp32 and fp64 in the build container.
 */

#ifndef REAL
#error "define REAL and SUF before including"
#endif

#define CAT_(a, b) a##b
#define CAT(a, b) CAT_(a, b)
#define FN(name) CAT(name, SUF)

/* sinkhorn.py:388-450 (sinkhorn_log, single histogram, no warmstart, log=False).
 * a[n1], b[n2], M[n1*n2] row-major; T_out[n1*n2]; returns the number of iterations executed.
 * scratch: Mr[n1*n2], u[n1], v[n2]. */
static int FN(sinkhorn_log)(int n1, int n2, const REAL *a, const REAL *b, const REAL *M, REAL reg,
                            int numItermax, REAL stopThr, REAL *T_out, REAL *Mr, REAL *u, REAL *v)
{
    int i, j, ii, done = 0;
    for (i = 0; i < n1 * n2; ++i) Mr[i] = -M[i] / reg;   /* sinkhorn.py:388 */
    for (i = 0; i < n1; ++i) u[i] = 0;                    /* :393-394 */
    for (j = 0; j < n2; ++j) v[j] = 0;
    for (ii = 0; ii < numItermax; ++ii) {                 /* :413 */
        /* v = logb - logsumexp(Mr + u[:,None], 0)           :415 */
        for (j = 0; j < n2; ++j) {
            REAL mx = -(REAL)INFINITY, s = 0;
            for (i = 0; i < n1; ++i) { REAL z = Mr[i * n2 + j] + u[i]; if (z > mx) mx = z; }
            for (i = 0; i < n1; ++i) s += FN(r_exp)(Mr[i * n2 + j] + u[i] - mx);
            v[j] = FN(r_log)(b[j]) - (FN(r_log)(s) + mx);
        }
        /* u = loga - logsumexp(Mr + v[None,:], 1)           :416 */
        for (i = 0; i < n1; ++i) {
            REAL mx = -(REAL)INFINITY, s = 0;
            for (j = 0; j < n2; ++j) { REAL z = Mr[i * n2 + j] + v[j]; if (z > mx) mx = z; }
            for (j = 0; j < n2; ++j) s += FN(r_exp)(Mr[i * n2 + j] + v[j] - mx);
            u[i] = FN(r_log)(a[i]) - (FN(r_log)(s) + mx);
        }
        done = ii + 1;
        if (ii % 10 == 0) {                               /* :418-433 */
            REAL e2 = 0;
            for (j = 0; j < n2; ++j) {
                REAL s = 0;
                for (i = 0; i < n1; ++i) s += FN(r_exp)(Mr[i * n2 + j] + u[i] + v[j]);
                e2 += (s - b[j]) * (s - b[j]);
            }
            if (FN(r_sqrt)(e2) < stopThr) break;
        }
    }
    for (i = 0; i < n1; ++i)                              /* :450 */
        for (j = 0; j < n2; ++j) T_out[i * n2 + j] = FN(r_exp)(Mr[i * n2 + j] + u[i] + v[j]);
    return done;
}

/* utils.py:154-171 (euclidean_distances, squared=True, X is not Y): M[n1,n2]. */
static void FN(sqdist)(int n1, int n2, int d, const REAL *X, const REAL *Y, REAL *M)
{
    int i, j, c;
    for (i = 0; i < n1; ++i) {
        REAL a2 = 0;
        for (c = 0; c < d; ++c) a2 += X[i * d + c] * X[i * d + c];
        for (j = 0; j < n2; ++j) {
            REAL b2 = 0, xy = 0, m;
            for (c = 0; c < d; ++c) { b2 += Y[j * d + c] * Y[j * d + c]; xy += X[i * d + c] * Y[j * d + c]; }
            m = -2 * xy; m += a2; m += b2;                /* utils.py:159-161 */
            M[i * n2 + j] = m > 0 ? m : 0;                /* :163 clamp(min=0) */
        }
    }
}

/* bregman.py:70-167 (fgw_projected: solver PGD, symmetric=True, warmstart=False, log=False)
 * together with utils.py:39-64 (init_matrix square_loss, tensor_product, gwggrad).
 * M[n1,n2], C1[n1,n1], C2[n2,n2], p[n1], q[n2]; T[n1,n2] holds G0 on entry when have_G0 else is
 * initialised to outer(p,q) (bregman.py:98-101).  Returns PGD iterations executed;
 * sk_iters[cpt] receives the Sinkhorn iteration count of each PGD step (may be NULL).
 * work: 6*n1*n2 + n1 + n2 REALs. */
/* loss_fun: 0 = "square_loss" (f1 = a^2, f2 = b^2, h1 = a, h2 = 2b), 1 = "kl_loss" (f1 = a log(a + 1e-15) - a, f2 = b,
 * h1 = a, h2 = log(b + 1e-15)) — utils.py:6-32. */
static REAL FN(lf_f1)(REAL a, int kl) { return kl ? a * FN(r_log)(a + (REAL)1e-15) - a : a * a; }
static REAL FN(lf_f2)(REAL b, int kl) { return kl ? b : b * b; }
static REAL FN(lf_h2)(REAL b, int kl) { return kl ? FN(r_log)(b + (REAL)1e-15) : 2 * b; }

static int FN(fgw_projected)(int n1, int n2, const REAL *M, const REAL *C1, const REAL *C2, const REAL *p,
                             const REAL *q, REAL alpha, REAL epsilon, int have_G0, int max_iter, REAL tol,
                             int numItermax, REAL stopThr, REAL *T, int *sk_iters, REAL *work, int loss_fun)
{
    REAL *constC = work, *A = work + n1 * n2, *tens = A + n1 * n2, *Tprev = tens + n1 * n2;
    REAL *Mr = Tprev + n1 * n2, *Tn = Mr + n1 * n2, *u = Tn + n1 * n2, *v = u + n1;
    int i, j, k, cpt = 0;
    REAL err = 1;
    if (!have_G0)
        for (i = 0; i < n1; ++i) for (j = 0; j < n2; ++j) T[i * n2 + j] = p[i] * q[j];
    /* init_matrix, utils.py:39-43: constC = f1(C1) p 1^T + 1 q^T f2(C2)^T ; hC1 = C1 ; hC2 = 2 C2 */
    for (i = 0; i < n1; ++i) {
        REAL r1 = 0;
        for (k = 0; k < n1; ++k) r1 += FN(lf_f1)(C1[i * n1 + k], loss_fun) * p[k];
        for (j = 0; j < n2; ++j) constC[i * n2 + j] = r1;
    }
    for (j = 0; j < n2; ++j) {
        REAL r2 = 0;
        for (k = 0; k < n2; ++k) r2 += q[k] * FN(lf_f2)(C2[j * n2 + k], loss_fun);
        for (i = 0; i < n1; ++i) constC[i * n2 + j] += r2;
    }
    while (err > tol && cpt < max_iter) {                 /* bregman.py:119 */
        for (i = 0; i < n1 * n2; ++i) Tprev[i] = T[i];
        /* tensor_product utils.py:48-53: A = -hC1 @ T @ hC2^T */
        for (i = 0; i < n1; ++i)
            for (j = 0; j < n2; ++j) {
                REAL s = 0;
                for (k = 0; k < n1; ++k) s += (-C1[i * n1 + k]) * T[k * n2 + j];
                A[i * n2 + j] = s;
            }
        for (i = 0; i < n1; ++i)
            for (j = 0; j < n2; ++j) {
                REAL s = 0;
                for (k = 0; k < n2; ++k) s += A[i * n2 + k] * FN(lf_h2)(C2[j * n2 + k], loss_fun);
                /* gwggrad = 2*(constC + A) ; tens = alpha*gw + (1-alpha)*M   bregman.py:124-125 */
                tens[i * n2 + j] = alpha * (2 * (constC[i * n2 + j] + s)) + (1 - alpha) * M[i * n2 + j];
            }
        {
            int it = FN(sinkhorn_log)(n1, n2, p, q, tens, epsilon, numItermax, stopThr, Tn, Mr, u, v); /* :142 */
            if (sk_iters) sk_iters[cpt] = it;
        }
        for (i = 0; i < n1 * n2; ++i) T[i] = Tn[i];
        if (cpt % 10 == 0) {                              /* :144-147 */
            REAL e2 = 0;
            for (i = 0; i < n1 * n2; ++i) e2 += (T[i] - Tprev[i]) * (T[i] - Tprev[i]);
            err = FN(r_sqrt)(e2);
        }
        ++cpt;
    }
    return cpt;
}

/* bregman.py:163-164 with utils.py:56-59: fgw_dist = (1-alpha) sum(M*T) + alpha * sum((constC - hC1 T hC2^T) * T). */
REAL FN(conan_oracle_fgw_dist)(int n1, int n2, const REAL *M, const REAL *C1, const REAL *C2, const REAL *p,
                               const REAL *q, const REAL *T, REAL alpha)
{
    int i, j, k;
    REAL lin = 0, gw = 0;
    REAL *A = (REAL *)malloc(sizeof(REAL) * (size_t)n1 * n2);
    for (i = 0; i < n1; ++i)
        for (j = 0; j < n2; ++j) {
            REAL s = 0;
            for (k = 0; k < n1; ++k) s += (-C1[i * n1 + k]) * T[k * n2 + j];
            A[i * n2 + j] = s;
        }
    for (i = 0; i < n1; ++i) {
        REAL r1 = 0;
        for (k = 0; k < n1; ++k) r1 += C1[i * n1 + k] * C1[i * n1 + k] * p[k];
        for (j = 0; j < n2; ++j) {
            REAL r2 = 0, s = 0;
            for (k = 0; k < n2; ++k) { r2 += q[k] * (C2[j * n2 + k] * C2[j * n2 + k]); s += A[i * n2 + k] * (2 * C2[j * n2 + k]); }
            gw += (r1 + r2 + s) * T[i * n2 + j];
            lin += M[i * n2 + j] * T[i * n2 + j];
        }
    }
    free(A);
    return (1 - alpha) * lin + alpha * gw;
}

/* barycenter.py:7-225 (fgw_barycenters) restricted to: loss_fun="square_loss" or "kl_loss", solver="PGD",
 * stop_criterion="barycenter", warmstartT=True, symmetric=True, init_C given, init_Y=None or given (bit 1 of
 * fixed_features: Y holds init_Y on entry, barycenter.py:69-80; bit 0 = fixed_features itself), p given or
 * uniform, every input graph of the same size n (the production glue always pads to N_max,
 * schnet_no_sum.py:242-252) — n may differ from N.
 *
 * Ys[K,n,d], Cs[K,n,n], ps[K,n], p[N] (NULL => uniform, barycenter.py:50-51), lambdas[K], init_C[N,N].
 * Outputs: Y[N,d], C[N,N], T[K,N,n], err_feature[max_iter], err_structure[max_iter] (entries past the
 * executed count are left untouched), iters[1 + max_iter*K*(1+max_iter)]:
 *   iters[0] = outer iterations executed; then for outer o, graph s at base 1 + (o*K+s)*(1+max_iter):
 *   [PGD iterations, Sinkhorn iterations of PGD step 0, 1, ...].
 * Returns 0, or -1 on allocation failure / bad arguments. */
int FN(conan_oracle_fgw_barycenter_loss)(int N, int K, int n, int d, const REAL *Ys, const REAL *Cs, const REAL *ps,
                                         const REAL *p_in, const REAL *lambdas, const REAL *init_C, REAL alpha,
                                         REAL epsilon, int max_iter, REAL tol, REAL inner_tol, int numItermax,
                                         REAL stopThr, int fixed_structure, int fixed_features, REAL *Y, REAL *C, REAL *T,
                                         REAL *err_feature, REAL *err_structure, int *iters, int loss_fun)
{
    int s, i, j, k, c, cpt = 0;
    REAL ef = (REAL)1e15, es = (REAL)1e15;                /* barycenter.py:89-91 */
    size_t nn = (size_t)N * n;
    REAL *p = (REAL *)malloc(sizeof(REAL) * N);
    REAL *Ms = (REAL *)malloc(sizeof(REAL) * K * nn);
    REAL *Yprev = (REAL *)malloc(sizeof(REAL) * (size_t)N * d);
    REAL *Cprev = (REAL *)malloc(sizeof(REAL) * (size_t)N * N);
    REAL *TC = (REAL *)malloc(sizeof(REAL) * nn);
    REAL *work = (REAL *)malloc(sizeof(REAL) * (6 * nn + N + n));
    int *have_T = (int *)calloc(K, sizeof(int));
    if (!p || !Ms || !Yprev || !Cprev || !TC || !work || !have_T || !init_C) return -1;
    for (i = 0; i < N; ++i) p[i] = p_in ? p_in[i] : (REAL)1 / (REAL)N;
    for (i = 0; i < N * N; ++i) C[i] = init_C[i];        /* :54-67 */
    const int have_init_Y = (fixed_features & 2) != 0;   /* :78-80: Y = init_Y */
    fixed_features &= 1;
    if (!fixed_features && !have_init_Y) for (i = 0; i < N * d; ++i) Y[i] = 0;   /* :76-77 ; otherwise Y holds init_Y on entry */
    for (s = 0; s < K; ++s) FN(sqdist)(N, n, d, Y, Ys + (size_t)s * n * d, Ms + s * nn);   /* :82 */
    if (iters) iters[0] = 0;
    while ((ef > tol || es > tol) && cpt < max_iter) {   /* :112 (err_rel_loss == 0) */
        for (i = 0; i < N * N; ++i) Cprev[i] = C[i];
        for (i = 0; i < N * d; ++i) Yprev[i] = Y[i];
        for (s = 0; s < K; ++s) {                         /* :122-142: fgw(Ms[s], C, Cs[s], p, ps[s], ..., T[s], max_iter, 1e-4) */
            int *it = iters ? iters + 1 + ((size_t)cpt * K + s) * (1 + max_iter) : 0;
            int npgd = FN(fgw_projected)(N, n, Ms + s * nn, C, Cs + (size_t)s * n * n, p, ps + (size_t)s * n, alpha,
                                         epsilon, have_T[s], max_iter, inner_tol, numItermax, stopThr,
                                         T + s * nn, it ? it + 1 : 0, work, loss_fun);
            if (it) it[0] = npgd;
            have_T[s] = 1;
        }
        if (!fixed_features) {
            /* update_feature_matrix utils.py:90-95: Y[i,c] = sum_s lam_s (1/p_i) sum_j T_s[i,j] Ys_s[j,c] */
            for (i = 0; i < N; ++i)
                for (c = 0; c < d; ++c) {
                    REAL acc = 0;
                    for (s = 0; s < K; ++s) {
                        REAL t = 0;
                        for (j = 0; j < n; ++j) t += Ys[((size_t)s * n + j) * d + c] * T[s * nn + (size_t)i * n + j];
                        acc += lambdas[s] * t * ((REAL)1 / p[i]);
                    }
                    Y[i * d + c] = acc;
                }
            for (s = 0; s < K; ++s) FN(sqdist)(N, n, d, Y, Ys + (size_t)s * n * d, Ms + s * nn);   /* :177 */
        }
        if (!fixed_structure) {
            /* update_square_loss utils.py:67-73: C = sum_s lam_s T_s Cs_s T_s^T / (p p^T) */
            for (i = 0; i < N * N; ++i) C[i] = 0;
            for (s = 0; s < K; ++s) {
                const REAL *Ts = T + s * nn, *Csp = Cs + (size_t)s * n * n;
                for (i = 0; i < N; ++i)
                    for (j = 0; j < n; ++j) {
                        REAL a = 0;
                        for (k = 0; k < n; ++k) {         /* kl_loss: log(clamp(Cs, min=1e-15)), utils.py:80-81 */
                            REAL cv = Csp[k * n + j];
                            if (loss_fun) cv = FN(r_log)(cv > (REAL)1e-15 ? cv : (REAL)1e-15);
                            a += Ts[(size_t)i * n + k] * cv;
                        }
                        TC[(size_t)i * n + j] = a;
                    }
                for (i = 0; i < N; ++i)
                    for (j = 0; j < N; ++j) {
                        REAL a = 0;
                        for (k = 0; k < n; ++k) a += TC[(size_t)i * n + k] * Ts[(size_t)j * n + k];
                        C[i * N + j] += lambdas[s] * a;
                    }
            }
            for (i = 0; i < N; ++i) for (j = 0; j < N; ++j) {
                C[i * N + j] /= (p[i] * p[j]);
                if (loss_fun) C[i * N + j] = FN(r_exp)(C[i * N + j]);          /* update_kl_loss utils.py:86-87 */
            }
        }
        ef = 0; es = 0;                                   /* :186-192 */
        if (!fixed_features) {
            REAL e2 = 0;
            for (i = 0; i < N * d; ++i) e2 += (Y[i] - Yprev[i]) * (Y[i] - Yprev[i]);
            ef = FN(r_sqrt)(e2);
        }
        if (!fixed_structure) {
            REAL e2 = 0;
            for (i = 0; i < N * N; ++i) e2 += (C[i] - Cprev[i]) * (C[i] - Cprev[i]);
            es = FN(r_sqrt)(e2);
        }
        if (err_feature) err_feature[cpt] = ef;
        if (err_structure) err_structure[cpt] = es;
        ++cpt;
        if (iters) iters[0] = cpt;
    }
    free(p); free(Ms); free(Yprev); free(Cprev); free(TC); free(work); free(have_T);
    return 0;
}

int FN(conan_oracle_fgw_barycenter)(int N, int K, int n, int d, const REAL *Ys, const REAL *Cs, const REAL *ps,
                                    const REAL *p_in, const REAL *lambdas, const REAL *init_C, REAL alpha,
                                    REAL epsilon, int max_iter, REAL tol, REAL inner_tol, int numItermax,
                                    REAL stopThr, int fixed_structure, int fixed_features, REAL *Y, REAL *C, REAL *T,
                                    REAL *err_feature, REAL *err_structure, int *iters)
{
    return FN(conan_oracle_fgw_barycenter_loss)(N, K, n, d, Ys, Cs, ps, p_in, lambdas, init_C, alpha, epsilon, max_iter, tol, inner_tol,
                                                numItermax, stopThr, fixed_structure, fixed_features, Y, C, T, err_feature,
                                                err_structure, iters, 0);
}

/* Backward of the block given the saved couplings (Y is the only output carrying
 * autograd, barycenter.py:120 wraps the coupling solves in no_grad):
 *   dYs[s][j,c] = lam_s * sum_i T_s[i,j] * (1/p_i) * dY[i,c]   (adjoint of utils.py:90-95). */
void FN(conan_oracle_fgw_barycenter_bwd)(int N, int K, int n, int d, const REAL *T, const REAL *p_in,
                                         const REAL *lambdas, const REAL *dY, REAL *dYs)
{
    int s, i, j, c;
    for (s = 0; s < K; ++s)
        for (j = 0; j < n; ++j)
            for (c = 0; c < d; ++c) {
                REAL a = 0;
                for (i = 0; i < N; ++i) {
                    REAL pi = p_in ? p_in[i] : (REAL)1 / (REAL)N;
                    a += T[((size_t)s * N + i) * n + j] * ((REAL)1 / pi) * dY[i * d + c];
                }
                dYs[((size_t)s * n + j) * d + c] = lambdas[s] * a;
            }
}

/* barycenter.py:393-399 (normalize_tensor): a + (x - min) * (b - a) / (max - min) over the whole slab. */
void FN(conan_oracle_normalize_tensor)(long count, const REAL *x, REAL a, REAL b, REAL *out)
{
    long i;
    REAL mn = x[0], mx = x[0];
    for (i = 1; i < count; ++i) { if (x[i] < mn) mn = x[i]; if (x[i] > mx) mx = x[i]; }
    for (i = 0; i < count; ++i) out[i] = a + (x[i] - mn) * (b - a) / (mx - mn);
}

#undef FN
#undef CAT
#undef CAT_
