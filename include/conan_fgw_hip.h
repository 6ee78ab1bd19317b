/*
 * conan_fgw_hip.h — C ABI of libconan_fgw_hip.so: the MI355X (gfx950) hot path of ConAN-FGW.
 *
 * The reference (duyhominhnguyen/conan-fgw) is 100 % Python and has no FFI of its own; its boundary for this path is
 * the Python model API (EquivModelsHolder.get_model -> SchNetNoSum / ViSNet, conan_fgw/src/model/common.py:469-546).
 * Each entry point below replaces one piece of third-party native code or one Python hot loop that the reference
 * reaches from that API; the replaced call site is cited as file:line relative to the reference root.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer on the current HIP device unless marked (host);
 *  - no entry point allocates, frees, synchronises the host or reads results back: all are stream-ordered and
 *    hipGraph-capturable; workspaces are caller-provided;
 *  - `stream` is a hipStream_t passed as void* (NULL = default stream);
 *  - return value: 0 = launched, <0 = error (CONAN_E_*), nothing launched on a bad argument;
 *  - float = IEEE fp32; indices are int32 on the device (edge counts < 2^31), the Python host converts to/from the
 *    reference's int64 tensors at the API edge.
 *  - graphs: `graph_ptr[G+1]` = first atom of every conformer graph (atoms of a graph are contiguous, as produced by
 *    the reference's collate_fn, conan_fgw/src/data/datasets.py:170-199).
 *  - edges: CSR by TARGET atom: edges of target i are rowptr[i] .. rowptr[i+1]-1, `col[e]` = source atom (global id),
 *    sources ascending inside a row, `tgt[e]` = i.  Same orientation as PyG flow source_to_target
 *    (edge_index[0] = col, edge_index[1] = tgt).
 */
#ifndef CONAN_FGW_HIP_H
#define CONAN_FGW_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CONAN_OK 0
#define CONAN_E_BADARG (-1)
#define CONAN_E_LAUNCH (-2)
#define CONAN_E_UNSUPPORTED (-3)

/* Library / device probe.  Returns CONAN_FGW_ABI_VERSION of the build.  The version changes whenever an existing export changes its
 * signature or meaning (v2: num_embeddings / pre_act arguments of round 2; v3: round-3 signatures), so a consumer compiled against
 * this header can detect a stale library: compare the return value with the macro. */
#define CONAN_FGW_ABI_VERSION 5
int conan_abi_version(void);

/* ---------------------------------------------------------------------------------------------- batch assembly */

/* Byte layout of one packed batch: what the host writes (pinned) and ONE H2D copy moves.  The K conformers of a molecule
 * share z, the 2-D bond graph and its features, so each molecule is packed once; sections are 16-byte aligned. */
typedef struct conan_batch_layout {
    int B, K;                /* molecules, conformers per molecule */
    int num_graphs;          /* B*K conformer graphs, molecule-major (datasets.py:170-199) */
    int num_atoms;           /* atoms over all conformer graphs */
    int num_bond_edges;      /* directed 2-D bond edges over all conformer graphs */
    int max_nodes;           /* atoms of the largest conformer = N_max of to_dense_batch (schnet_no_sum.py:242) */
    int x_dim, ea_dim;       /* atom / bond feature widths (9 / 3 for PyG from_smiles) */
    long long off_atom_off, off_bond_off;   /* int32 [B+1] prefix sums of atoms / bond edges per molecule */
    long long off_z;         /* int32 [sum n] */
    long long off_pos;       /* float [sum K*n*3]: per molecule [K][n][3] */
    long long off_x;         /* float [sum n*x_dim] */
    long long off_bsrc, off_bdst;           /* int32 [sum e] molecule-local atom indices */
    long long off_battr;     /* float [sum e*ea_dim] */
    long long off_y;         /* float [B] */
    long long bytes;         /* size of the packed buffer */
} conan_batch_layout;

/* (host) Layout for B molecules of n_atoms[m] atoms and n_bonds[m] directed bond edges, K conformers each. */
int conan_collate_layout(int B, int K, const int *n_atoms, const int *n_bonds, int x_dim, int ea_dim, conan_batch_layout *layout);
/* (host) Pack B dataset items — what LargeConformerBasedDataset.get returns per molecule (datasets.py:133-148): z[m][n] int64,
 * pos[m][K][n][3], x[m][n][x_dim], edge_index[m][2][e] (molecule-local), edge_attr[m][e][ea_dim], y[m] — into `packed`
 * (layout->bytes bytes, pinned host memory for an asynchronous copy).  All pointers are HOST pointers.  Replaces the
 * Python loops + Batch.from_data_list of collate_fn (datasets.py:170-199). */
int conan_collate_pack(const conan_batch_layout *layout, const int *n_atoms, const int *n_bonds, const int64_t *const *z,
                       const float *const *pos, const float *const *x, const int64_t *const *edge_index,
                       const float *const *edge_attr, const float *y, void *packed);
/* (device) Expand a packed batch (already copied to the device) into the flat tensors of the reference's model API:
 * z[A] int64, pos[A,3], batch[A] int64 (= batch_node_index = data_batch.batch), x[A,x_dim], edge_index[2,E] int64 with node
 * offsets (Batch.from_data_list), edge_attr[E,ea_dim], y[G] (every conformer Data carries its molecule's target),
 * graph_ptr[G+1] int32, conformers_index[G] int64 (create_aggregation_index, common.py:414-423) and, when not NULL,
 * conf_node_batch[A] int64 (data_batch.conf_node_batch, datasets.py:177,191,197: the molecule-level atom number of every
 * conformer atom; nothing in the reference reads it).  `layout` is a HOST pointer (passed to the kernel by value). */
int conan_collate_unpack(const void *packed_dev, const conan_batch_layout *layout, int64_t *z, float *pos, int64_t *batch,
                         float *x, int64_t *edge_index, float *edge_attr, float *y, int *graph_ptr, int64_t *conformers_index,
                         int64_t *conf_node_batch, void *stream);

/* ---------------------------------------------------------------------------------------------- graph construction */

/* graph_ptr[G+1] from the sorted per-atom graph id vector the reference passes around (`batch`,
 * schnet_no_sum.py:157,205,338).  Empty graphs are allowed. */
int conan_graph_ptr_from_batch(const int64_t *batch, int num_atoms, int num_graphs, int *graph_ptr, void *stream);

/* Fixed-radius neighbour search: replaces torch_cluster.radius_graph as reached through
 * RadiusInteractionGraph.forward (schnet_no_sum.py:160,208,342; visnet.py:276; torch_geometric_visnet.py:331-337).
 * Candidate (j, i) iff same graph and d2 < r*r strictly with d2 = fl(fl(dx*dx + dy*dy) + dz*dz) (no FMA); the target itself is a
 * candidate.  Per target the first `limit` candidates in ascending source index are kept, limit = loop ? cap : cap + 1
 * (torch-cluster 1.6.1: radius_graph calls radius(x, x, r, batch, batch, cap if loop else cap + 1)), then the self pair is
 * dropped unless `loop`: a target with >= cap + 1 lower-index candidates keeps cap + 1 edges, any other truncated one cap.
 * WHICH neighbours survive a truncated row (more than `limit` atoms in range: conformers of > 33 atoms) follows torch-cluster's
 * CUDA kernel (radius_cuda.cu: linear scan in ascending index); its CPU path (nanoflann KD-tree, unsorted result) keeps an
 * implementation-defined subset of the same size, so on truncated rows "bit-exact neighbour indices" means equal to the CUDA
 * scan order, not to a CPU run.  Untruncated rows (every ESOL / FreeSolv-sized conformer) are the same set either way.
 * Outputs: rowptr[num_atoms+1]; col/tgt/dist sized for (loop ? cap : cap + 1)*num_atoms entries (rowptr[num_atoms] are written);
 * dist[e] = sqrt(d2) = the reference's edge_weight.  `deg_ws[num_atoms+1]` is scratch. */
int conan_radius_graph_csr(const float *pos, const int *graph_ptr, int num_atoms, int num_graphs, float r, int cap,
                           int loop, int *deg_ws, int *rowptr, int *col, int *tgt, float *dist, void *stream);

/* CSR by SOURCE of the same edge set (needed by the backward of the message passing): t_rowptr[num_atoms+1],
 * t_eid[e'] = edge id (position in the by-target CSR), ascending inside each source row.  deg_ws is unused (may be NULL). */
int conan_csr_transpose(const int *graph_ptr, int num_graphs, int num_atoms, const int *rowptr, const int *col,
                        int *deg_ws, int *t_rowptr, int *t_eid, void *stream);

/* Undirected pairs of the edge set.  The continuous filter depends on an edge only through d_ij = d_ji, so both
 * directions of a pair share one filter row: the filter network then runs on ~E/2 rows.  pid[e] = pair of edge e;
 * pair p is represented by edge pair_e0[p] (source <= target, or a one-directional edge when the cap truncated its
 * reverse) and pair_e1[p] = the reverse edge or -1; pair_dist[p] = its distance; the pair count is written to
 * pidx_ws[max_edges] (device).  flag_ws, pidx_ws: int scratch of max_edges + 1 entries; scan_ws: 2*(max_edges/4096 + 2) ints. */
int conan_edge_pairs(const int *rowptr, const int *col, const int *tgt, const float *dist, const int *num_edges_dev,
                     int max_edges, int *flag_ws, int *pidx_ws, int *scan_ws, int *pid, int *pair_e0, int *pair_e1,
                     float *pair_dist, void *stream);

/* edge_index[2, E] int64 in the reference's layout from the CSR (E = capacity of the output rows = host-known edge
 * count).  Row 0 = source, row 1 = target. */
int conan_edge_index_i64(const int *col, const int *tgt, int num_edges, int64_t *edge_index, void *stream);

/* ---------------------------------------------------------------------------------------------- SchNet trunk */

/* out[a, :] = weight[z[a], :]  (torch.nn.Embedding(100, H, padding_idx=0); schnet_no_sum.py:159,207).  weight is
 * [num_embeddings, hidden]; an index outside [0, num_embeddings) (a device assert in torch) yields a NaN row and
 * contributes nothing to the backward — never an out-of-bounds access. */
int conan_embedding_fwd(const int64_t *z, const float *weight, int num_atoms, int hidden, int num_embeddings, float *out,
                        void *stream);
/* dweight[r, :] = sum_{a: z[a]==r} dout[a, :] (row padding_idx = 0), deterministic two-stage reduction;
 * ws holds conan_embedding_bwd_ws(...) floats; num_embeddings <= 100. */
/* out[a, r] = 1 if z[a] == r and r != padding_idx, else 0 (fp32 [num_atoms, num_embeddings]): with it the embedding gradient is the weight gradient
 * onehot(z)^T dout and can join the batched slab launch of a backward pass (conan_linear_wgrad_slabs_batch) instead of running its own two kernels. */
int conan_onehot_rows(const int64_t *z, int num_atoms, int num_embeddings, int padding_idx, float *out, void *stream);
long long conan_embedding_bwd_ws(int num_atoms, int hidden, int num_embeddings);
int conan_embedding_bwd(const int64_t *z, const float *dout, int num_atoms, int hidden, int num_embeddings,
                        int padding_idx, float *dweight, float *ws, void *stream);

/* y[M,N] = act(x[M,K] @ W^T + bias) (+ residual[M,N]).  Arithmetic: K, N multiples of 64 run as a two-plane fp16 split of both operands
 * (v = h1 + h2, three partial products per block on v_mfma_f32_32x32x16_f16, fp32 accumulation) with exact power-of-two scales taken from
 * the data — one per weight matrix, one per x row, undone by one multiply per output — so that operands of any magnitude sit in the middle
 * of fp16's range: fp32-CLASS accuracy (measured 1.4e-7 relative to an fp64 reference, rows spanning ten decades each to 2e-6 of their own
 * norm; a plain fp32 GEMM measures the same), not bit-identical to an fp32 FMA chain; other shapes run on the fp32 MFMA
 * (v_mfma_f32_32x32x2_f32).  No reduced-precision mode exists.
 * W is torch.nn.Linear's [N,K] when w_kn == 0, or a [K,N] matrix when w_kn == 1 (used by the backward: dx = g @ W).
 * act: 0 = identity, 1 = shifted softplus (softplus(v) - ln 2), 3 = SiLU, 2 = multiply by ssp'(.) evaluated from `residual`, which
 * then holds the saved OUTPUT o of an ssp layer (ssp' = 1 - 0.5*exp(-o)) instead of being added: the fused backward
 * dpre = (g @ W) * ssp'(pre).  m_dev (nullable): device int holding the real row count
 * (<= M) for edge-level calls whose row count is only known on the device; rows >= *m_dev are not touched.
 * Replaces the aten addmm + softplus of CFConv.lin1/lin2, InteractionBlock.mlp/.lin and the heads
 * (PyG SchNet, reached from schnet_no_sum.py:163-164,176-178,211-212,225-231). */
int conan_linear_fwd(const float *x, const float *w, const float *bias, const float *residual, int M, int K, int N,
                     int w_kn, int act, const int *m_dev, float *y, void *stream);

/* Same Linear with an activation (1 = shifted softplus, 3 = SiLU) that ALSO stores the pre-activation x W^T + b in `pre` — what the
 * backward of a SiLU layer needs (silu' is not a function of the output) — with one extra store instead of a separate
 * activation kernel re-reading the GEMM result.  K and N multiples of 64 (else CONAN_E_UNSUPPORTED: compose it). */
int conan_linear_act_fwd(const float *x, const float *w, const float *bias, int M, int K, int N, int act, const int *m_dev,
                         float *y, float *pre, void *stream);

/* num_layers Linear layers of the SAME input x [M,K] -> y[q] [M,N] (weights w[q] [N,K], biases bias[q] or bias == NULL, all of one width):
 * y[q] = act(x w[q]^T + bias[q]), pre[q] (pre == NULL or entries NULL: not wanted) = the pre-activation.  K = N = 128 and 2..4 layers run as ONE
 * launch whose workgroups for the same rows sit side by side (x is streamed from HBM once instead of once per layer); other shapes are
 * one conan_linear_fwd per layer.  Replaces the q / k / v and dk / dv / f_proj projections of ViS_MP (torch_geometric_visnet.py:596-604,
 * 623-640), which the reference evaluates as separate nn.Linear calls on one tensor.  The pointer arrays are host arrays. */
int conan_linear_multi_fwd(const float *x, const float *const *w, const float *const *bias, int M, int K, int N, int num_layers, int act,
                           const int *m_dev, float *const *y, float *const *pre, void *stream);
/* y [M,N] = sum_c x[c] W[c]^T (+ bias) (+ residual): num_inputs (2 or 3) contractions of 128 columns each — x[c] [M,128] with row pitch ldx[c],
 * W[c] torch.nn.Linear's [N,128] (w_kn == 0) or a [128,N] matrix (w_kn == 1: the backward's dx = sum_c g_c W_c) — in ONE launch, the sum kept in
 * the accumulators.  N = 128 (else CONAN_E_UNSUPPORTED: chain conan_linear_fwd through `residual`).  residual may be y.  Same two-plane fp16
 * arithmetic as conan_linear_fwd; a row's chunks share one power-of-two unit (the largest of their own), i.e. what one scale for the
 * concatenated row [x_0 | x_1 | x_2] gives.  Replaces the autograd sum of the input gradients of ViS_MP's dk / dv / f_proj projections of one
 * f_ij (torch_geometric_visnet.py:600-604,637-640 through aten addmm + add).  The pointer arrays are host arrays. */
int conan_linear_sum_fwd(const float *const *x, const int *ldx, const float *const *w, int num_inputs, int w_kn, const float *bias,
                         const float *residual, int M, int N, const int *m_dev, float *y, void *stream);
/* g[rows,width] = dy * ssp'(v) computed from the layer OUTPUT y (ssp'(v) = sigmoid(v) = 1 - 0.5*exp(-y)). In place allowed. */
int conan_ssp_bwd(const float *dy, const float *y, int rows, int width, const int *m_dev, float *g, void *stream);
/* dW[N,K] = g^T @ x and dbias[N] = column sums of g (dbias nullable), deterministic two-stage reduction (no float
 * atomics); ws holds conan_linear_wgrad_ws(M,K,N) floats. */
long long conan_linear_wgrad_ws(int M, int K, int N);
int conan_linear_wgrad(const float *g, const float *x, int M, int K, int N, const int *m_dev, float *dW, float *dbias,
                       float *ws, void *stream);
/* conan_linear_wgrad (dW != NULL) / conan_linear_wgrad_slabs (dW == NULL: slabs only, reduced later by conan_wgrad_reduce_batch) for a
 * gradient g whose maximum magnitude is known on the device (`gmax`: one float, e.g. from conan_cfconv_bwd_w_pairs): the product runs
 * on two fp16 planes per operand with g scaled into fp16's range — 3 MFMAs per product instead of 6.  x must be O(1..1e3)
 * (activations); K > 64. */
int conan_linear_wgrad_scaled(const float *g, const float *x, int M, int K, int N, const int *m_dev, float *dW, float *dbias, float *ws,
                              const float *gmax, void *stream);
/* Same with x = GaussianSmearing(dist) generated on the fly: dW[N,Gs] = g^T rbf(dist), rbf[m,k] = exp(coeff (dist[m]-offset[k])^2)
 * (weight gradient of the first filter-network layer, schnet_no_sum.py InteractionBlock.mlp[0]; no [M,Gs] buffer is read).
 * ws as conan_linear_wgrad_ws(M, Gs, N). */
int conan_rbf_wgrad(const float *g, const float *dist, int M, const float *offset, int num_gaussians, float coeff, int N,
                    const int *m_dev, float *dW, float *dbias, float *ws, void *stream);

/* Deferred form of the two weight gradients above: stage 1 only (the per-slice partial sums stay in ws, which must live until the
 * reduction), and ONE launch that reduces the slabs of many weight gradients — a backward pass of the stage-2 model has 24
 * Linear layers, i.e. 24 slab reductions of ~6 us each when done one by one.  Same arithmetic and summation order as the
 * immediate form (bitwise-equal results).  Only for shapes with conan_wgrad_batchable(K, N) != 0 (N*K and N multiples of 4).
 * `jobs` is a HOST array. */
typedef struct conan_wgrad_job {
    const float *ws;         /* the workspace the slabs were written to (conan_linear_wgrad_ws(M, K, N) floats) */
    float *dW, *dbias;       /* outputs [N,K], [N] (dbias nullable) */
    int M, K, N;             /* the shape the slabs were produced for */
    int slices;              /* 0: slabs of conan_linear_wgrad_slabs / conan_rbf_wgrad_slabs (count derived from M, K);
                                > 0: explicit slab count (conan_filter_bwd_slices(M) for conan_filter_bwd) */
} conan_wgrad_job;
int conan_wgrad_batchable(int K, int N);
int conan_linear_wgrad_slabs(const float *g, const float *x, int M, int K, int N, const int *m_dev, float *ws, void *stream);
int conan_rbf_wgrad_slabs(const float *g, const float *dist, int M, const float *offset, int num_gaussians, float coeff, int N,
                          const int *m_dev, float *ws, void *stream);
int conan_wgrad_reduce_batch(const conan_wgrad_job *jobs, int num_jobs, void *stream);
/* Stage 1 of MANY weight gradients in one launch (per k-tile width): job j is exactly conan_linear_wgrad_slabs(g, x, M, K, N, m_dev, ws)
 * — same slabs, same bits — but the jobs' workgroups share one grid.  A node-level layer alone (25 k rows) is 198 latency-bound
 * workgroups on a 256-CU chip; the 22 of a stage-2 backward pass, postponed until their g and x all exist, fill it.  The caller keeps
 * g, x and ws alive until the launch has run; the slabs are then reduced with conan_wgrad_reduce_batch as usual. */
typedef struct {
    const float *g, *x;      /* [M,N], [M,K] */
    const int *m_dev;        /* nullable device-side row count */
    float *ws;               /* conan_linear_wgrad_ws(M, K, N) floats */
    int M, K, N;
    int slices;              /* 0: as conan_linear_wgrad_slabs; > 0: this many row slices (<= the default count; the same value then goes
                                into conan_wgrad_job.slices).  With many jobs in flight fewer, longer slices keep the chip just as
                                busy and shrink the slab volume the reducer has to read. */
} conan_wgrad_slab_job;
int conan_linear_wgrad_slabs_batch(const conan_wgrad_slab_job *jobs, int num_jobs, void *stream);

/* Backward of the filter network below its second Linear, fused (filter_bwd.hip):
 *     dh1 = (g @ w2) * ssp'(h1) ;  dW1[F,Gs] = dh1^T rbf(dist) ;  db1[F] = colsum(dh1)
 * i.e. conan_linear_fwd(g, w2, residual=h1, w_kn=1, act=2) followed by conan_rbf_wgrad, without dh1 [M,F] ever reaching HBM
 * (schnet_no_sum.py InteractionBlock.mlp = Linear(Gs,F) - ShiftedSoftplus - Linear(F,F); autograd of mlp[2] input, the
 * activation and mlp[0] weight).  g [M,F] is the gradient w.r.t. the mlp output, h1 [M,F] the saved ssp output
 * (conan_filter_fwd h1_out), w2 [F,F] the forward weight of mlp[2].  ws holds conan_filter_bwd_ws(M, Gs, F) floats.
 * dW1 == NULL: slabs only, to be reduced by conan_wgrad_reduce_batch with job.slices = conan_filter_bwd_slices(M), K = Gs, N = F.
 * Supported: conan_filter_bwd_supported(Gs, F) (F = 128, Gs <= 63); otherwise CONAN_E_UNSUPPORTED (compose the two calls). */
int conan_filter_bwd_supported(int num_gaussians, int num_filters);
int conan_filter_bwd_slices(int M);
long long conan_filter_bwd_ws(int M, int num_gaussians, int num_filters);
int conan_filter_bwd(const float *g, const float *h1, const float *dist, int M, const float *offset, int num_gaussians, float coeff,
                     const float *w2, int num_filters, const int *m_dev, float *dW1, float *db1, float *ws, const float *gmax, void *stream);

/* Backward of the WHOLE filter network in one pass over the pair rows (filter_bwd2.hip, round 4): conan_filter_bwd plus the weight / bias
 * gradient of the second Linear, dW2 [F,F] = g^T h1, db2 [F] = column sums of g — g and h1 cross HBM once instead of twice.  Requires the
 * device-side maximum of |g| (`gmax`, see conan_cfconv_bwd_w_pairs): both products run on two fp16 planes.  ws holds
 * conan_filter_bwd2_ws(M, Gs, F) floats = [slabs1 | bias1 | slabs2 | bias2] with conan_filter_bwd2_slices(M) slabs each.
 * dW1 == dW2 == NULL: slabs only, to be reduced by two conan_wgrad_reduce_batch jobs — (ws, K = Gs, N = F) and
 * (ws + slices * (F * Gs + F), K = F, N = F), both with job.slices = conan_filter_bwd2_slices(M).
 * Supported: conan_filter_bwd2_supported(Gs, F) (F = 128, Gs <= 63); reference: schnet_no_sum.py:161-164,209-212 (the filter network's backward).
 * Precision floor of the single global scale (also conan_filter_bwd / conan_linear_wgrad_scaled with gmax): an element of g carries an absolute
 * error of up to ~2^-29 max|g| once it is smaller than ~2^-19 max|g| — fp32-class while |g| spans fewer than ~five decades, a bounded floor
 * beyond that (tests/test_gpu_ops.py pins it with a 1e6 outlier).  Callers whose gradients can span more pass gmax = NULL to
 * conan_filter_bwd + conan_linear_wgrad (three bf16 planes, no scale, full fp32 exponent range, twice the matrix work). */
int conan_filter_bwd2_supported(int num_gaussians, int num_filters);
int conan_filter_bwd2_slices(int M);
long long conan_filter_bwd2_ws(int M, int num_gaussians, int num_filters);
int conan_filter_bwd2(const float *g, const float *h1, const float *dist, int M, const float *offset, int num_gaussians, float coeff,
                      const float *w2, int num_filters, const int *m_dev, const float *gmax, float *dW1, float *db1, float *dW2, float *db2,
                      float *ws, void *stream);

/* rbf[e,k] = exp(coeff * (dist[e] - offset[k])^2): GaussianSmearing (PyG; schnet_no_sum.py:161,209).  `offset` is the
 * module's buffer (distance_expansion.offset), coeff = -0.5/(offset[1]-offset[0])^2.  num_edges_dev (nullable) = device
 * int with the edge count (rowptr[num_atoms]); at most max_edges rows are written. */
int conan_rbf_fwd(const float *dist, const int *num_edges_dev, int max_edges, const float *offset, int num_gaussians,
                  float coeff, float *rbf, void *stream);
/* out[e,:] = in[e,:] * 0.5*(cos(dist[e]*pi/cutoff)+1): CFConv's cosine cutoff (PyG; schnet_no_sum.py:163-164).
 * Also its own backward (the factor does not depend on trainable parameters).  In place allowed. */
int conan_cutoff_scale(const float *dist, const int *num_edges_dev, int max_edges, int width, float cutoff,
                       const float *in, float *out, void *stream);

/* Stage-2 aggregation head (head.hip; schnet_based_models.py:163-171): for conformer-graph rows x3, xc, xb [G = num_molecules*K, D]
 *   out[b] = wreg . mean_k( x3 W3^T + b3 + xc + agg_weight (xb Wb^T + bb) ) + breg            [num_molecules]
 * = transformation_matrix_3d / _bary (xc is transformation_matrix_cov's output), the weighted sum, conformers_mean_aggr and
 * molecular_regression_lin (Linear(D, 1)) in ONE launch; the conformer mean is taken first (all of it is linear).  The forward also
 * returns the per-molecule means m3, mb and the pre-regression vector t [num_molecules, D] for the backward, which writes the three input
 * gradients [G, D] and all six parameter gradients (fixed summation order over the molecules: bitwise reproducible).  D <= 64. */
int conan_stage2_head_supported(int D);
int conan_stage2_head_fwd(const float *x3, const float *xc, const float *xb, const float *W3, const float *b3, const float *Wb, const float *bb,
                          const float *wreg, const float *breg, float agg_weight, int num_molecules, int K, int D, float *out, float *m3, float *mb,
                          float *t, void *stream);
int conan_stage2_head_bwd(const float *dout, const float *W3, const float *Wb, const float *wreg, const float *m3, const float *mb, const float *t,
                          float agg_weight, int num_molecules, int K, int D, float *dx3, float *dxc, float *dxb, float *dW3, float *db3, float *dWb,
                          float *dbb, float *dwreg, float *dbreg, void *stream);

/* Regression criterion of the training step and its gradient in one launch: loss[0] = mean((pred - target)^2), dpred[i] = 2 (pred[i] -
 * target[i]) / n.  Replaces torch.nn.functional.mse_loss + its backward (nn.MSELoss of the reference's Lightning module, common.py) — five
 * small launches on the critical path between the forward and the backward of a step.  Fixed summation order. */
int conan_mse_loss_fwd(const float *pred, const float *target, int n, float *loss, float *dpred, void *stream);

/* torch.optim.Adam's step (the reference's optimiser, train_val.py; no amsgrad, no maximize) for a WHOLE model in one launch (round 5): parameters,
 * gradients and both moments are flat fp32 buffers of n elements in one common order (16-byte aligned; conan_fgw_amd.parallel.FlatAdam lays the
 * parameters out in FlatGradients' order and re-points every Parameter at its slice).  step_dev: one device float holding the number of steps
 * taken so far (0 at the start), advanced by the call — device-side so that a captured graph replays it; ticket_dev: one zeroed device word of
 * workspace.  Hyper-parameters are doubles: the bias corrections 1 - b^t are formed in fp64 (they cancel in fp32).   m = b1 m + (1 - b1) g;  v = b2 v + (1 - b2) g^2;  p -= lr / (1 - b1^t) * m / (sqrt(v) / sqrt(1 - b2^t) + eps), g += weight_decay * p first.
 * lr_dev (nullable, ABI 5): one device double that overrides `lr` — a launch captured in a HIP graph then follows a learning-rate scheduler
 * (the reference's ReduceLROnPlateau, common.py:253-262) between replays; the by-value `lr` would be baked into the graph. */
int conan_adam_flat_step(float *params, const float *grads, float *exp_avg, float *exp_avg_sq, float *step_dev, unsigned *ticket_dev, long long n,
                         double lr, double beta1, double beta2, double eps, double weight_decay, const double *lr_dev, void *stream);

/* torch.nn.utils.clip_grad_norm_(parameters, max_norm) (norm type 2; what Lightning runs for the reference's Trainer(gradient_clip_val=1.0),
 * trainer.py:177) on the flat gradient buffer, without a host round trip: norm_coef_dev[0] = total L2 norm, norm_coef_dev[1] = min(1, max_norm /
 * (norm + 1e-6)), then grads *= coefficient in place (skipped when it is 1).  Squares are summed in fp64, per workgroup into partials_dev
 * (CONAN_GRAD_CLIP_MAX_BLOCKS doubles of workspace) and from there in a fixed tree by the last workgroup to arrive: the same bits on every run.  ticket_dev: one zeroed device word.
 * Two launches on `stream`; graph-capturable. */
#define CONAN_GRAD_CLIP_MAX_BLOCKS 256
int conan_grad_clip_flat(float *grads, long long n, double max_norm, float *norm_coef_dev, double *partials_dev, unsigned *ticket_dev, void *stream);

/* Two chained node-level Linear layers in one launch (mlp2.hip):
 *   forward : mid = ssp(x w1^T + b1) [M,N1];  y = mid w2^T + b2 (+ residual) [M,N2]
 *             = InteractionBlock's  conv.lin2 -> act -> lin (+ x)  (schnet_no_sum.py:164 with PyG's InteractionBlock.forward)
 *   backward: dmid = (dy w2) * ssp'(mid) [M,N1];  dx = dmid w1 [M,K]   (mid = the forward's saved output; the weight gradients are
 *             conan_linear_wgrad(dy, mid) and conan_linear_wgrad(dmid, x))
 * Same arithmetic as two conan_linear_fwd calls (two fp16 planes with per-matrix / per-row scales, fp32-class; the intermediate's row scale
 * is taken from the accumulators).  mid_out / dmid_out nullable.
 * Supported: conan_mlp2_supported(M, K, N1, N2) (K = N1 = N2 = 128, M <= 65536: one 32-row tile per wavefront, weights staged once per
 * workgroup — a node-level kernel); otherwise CONAN_E_UNSUPPORTED (compose the two calls). */
int conan_mlp2_supported(int M, int K, int N1, int N2);
int conan_mlp2_fwd(const float *x, const float *w1, const float *b1, const float *w2, const float *b2, const float *residual, int M, int K, int N1,
                   int N2, float *mid_out, float *y, void *stream);
int conan_mlp2_bwd(const float *dy, const float *w2, const float *w1, const float *mid, int M, int K, int N1, int N2, float *dmid_out, float *dx,
                   void *stream);
/* Same kernel with the activation behind the second layer — the per-atom heads lin1 -> lin2 -> act (schnet_no_sum.py:176-178, 225-231):
 *   forward : mid = x w1^T + b1 [M,N1];  y = ssp(mid w2^T + b2) [M,N2]
 *   backward: g = dy * ssp'(y) [M,N2] (written to g_out);  dmid = g w2 [M,N1];  dx = dmid w1 [M,K]
 *             (weight gradients: conan_linear_wgrad(g, mid) and conan_linear_wgrad(dmid, x))
 * Supported: conan_mlp2_outact_supported(M, K, N1, N2) (K = 128, N1 = N2 = 64, M <= 65536). */
int conan_mlp2_outact_supported(int M, int K, int N1, int N2);
int conan_mlp2_outact_fwd(const float *x, const float *w1, const float *b1, const float *w2, const float *b2, int M, int K, int N1, int N2,
                          float *mid_out, float *y, void *stream);
int conan_mlp2_outact_bwd(const float *dy, const float *y, const float *w2, const float *w1, int M, int K, int N1, int N2, float *g_out,
                          float *dmid_out, float *dx, void *stream);

/* Fused continuous-filter generator: for every edge e
 *   W[e,:] = ( mlp2( ssp( mlp0( rbf(dist[e]) ) ) ) ) * 0.5*(cos(dist[e]*pi/cutoff)+1)
 * = GaussianSmearing + InteractionBlock.mlp + CFConv's cosine cutoff in ONE kernel, both GEMMs on two fp16 planes per operand
 * (v_mfma_f32_32x32x16_f16; weights scaled by a power of two from their block maximum; fp32-class accuracy, see conan_linear_fwd), every intermediate in registers (PyG; reached from schnet_no_sum.py:161-164,209-212).  w1[F,Gs], b1[F], w2[F,F], b2[F]
 * are the torch Linear parameters of interactions.{i}.mlp.{0,2}.  h1_out (nullable) receives ssp(mlp0(rbf)) [E,F] for the
 * backward.  Supported shapes: conan_filter_fused_supported(Gs, F) (Gs <= 64, F in {32,64,128}); otherwise
 * CONAN_E_UNSUPPORTED and the caller composes conan_rbf_fwd / conan_linear_fwd / conan_cutoff_scale. */
int conan_filter_fused_supported(int num_gaussians, int num_filters);
int conan_filter_fwd(const float *dist, const int *num_edges_dev, int max_edges, const float *offset, int num_gaussians,
                     float coeff, float cutoff, int num_filters, const float *w1, const float *b1, const float *w2,
                     const float *b2, float *W, float *h1_out, void *stream);

/* Forward-only fusion of conan_filter_fwd and conan_cfconv_fwd (round 5): out[i,:] = sum_{e in row i} x[col[e],:] * W[e,:] with W[e,:] generated
 * from dist[e] inside the kernel and consumed from the accumulators — the [E,F] filter tensor (and h1) never reaches HBM.  Rows are the DIRECTED
 * edges in CSR order (tgt[e] ascending); one filter row per directed edge, i.e. twice the matrix work of the pair-shared generator for
 * 4 B + 2 indices in and nothing out per edge.  No backward exists for it: the training step keeps conan_filter_fwd + conan_cfconv_fwd, whose
 * saved W / h1 the backward kernels read.  `out` [num_atoms, F] is cleared by the call.  num_filters == 128, num_gaussians <= 64
 * (conan_filter_cfconv_fwd_supported), else CONAN_E_UNSUPPORTED.  Same arithmetic as the two kernels it replaces (two fp16 planes per operand);
 * a target's sum is formed in edge order inside a 32-edge tile and the (at most two, at cap 32) tile partials are added to the cleared row, so
 * results are bitwise reproducible.  Replaces CFConv.forward's edge half (PyG; reached from schnet_no_sum.py:161-164,209-212) at inference. */
int conan_filter_cfconv_fwd_supported(int num_gaussians, int num_filters);
int conan_filter_cfconv_fwd(const float *x, const float *dist, const int *col, const int *tgt, const int *num_edges_dev, int max_edges,
                            const float *offset, int num_gaussians, float coeff, float cutoff, int num_filters, const float *w1,
                            const float *b1, const float *w2, const float *b2, int num_atoms, float *out, void *stream);

/* CFConv message + aggregation (the HBM-bound kernel of the path): out[i,:] = sum_{e in row i} x[col[e],:] * W[e,:].
 * Replaces index_select + mul + scatter-add inside CFConv.propagate (PyG; schnet_no_sum.py:163-164,211-212).
 * CSR segment sum, one wavefront per target, no atomics.  pid (nullable): row of W used by edge e (conan_edge_pairs);
 * NULL = row e.  zero_slot (nullable, ABI 5): one device float that this launch clears — the `gmax` word that conan_cfconv_bwd_xw_pairs raises in
 * the backward pass of the same gather (no kernel of the backward pass runs in front of that launch, so the forward clears it). */
int conan_cfconv_fwd(const float *x, const float *W, const int *rowptr, const int *col, const int *pid, int num_atoms,
                     int num_filters, float *out, float *zero_slot, void *stream);
/* Backward: dx[j,:] = sum_{e: col[e]==j} W[e,:]*dout[tgt[e],:] (via the by-source CSR), dW[e,:] = x[col[e],:]*dout[tgt[e],:].
 * zero_slot (nullable, ABI 4): one device float that this launch clears — the `gmax` word of the conan_cfconv_bwd_w_pairs call that follows on
 * the same stream, so that no fill launch of its own is needed and the word is fresh on every backward pass (captured graphs included). */
int conan_cfconv_bwd_x(const float *W, const float *dout, const int *t_rowptr, const int *t_eid, const int *tgt,
                       const int *pid, int num_atoms, int num_filters, float *dx, float *zero_slot, void *stream);
/* Pair-level filter gradient (before the cosine cutoff): dWp[p,:] = C(d_p) * sum over the (1 or 2) edges of pair p of
 * x[src(e),:] * dout[tgt(e),:].  gmax (nullable, num_filters = 128 only): one device float, ZEROED by the caller before the call; the
 * kernel raises it to max |dWp| as it writes.  conan_filter_bwd / conan_linear_wgrad take that word to run their products on two
 * fp16 planes of the gradient scaled into fp16's range (half the matrix-pipe work of the three-plane bf16 form). */
int conan_cfconv_bwd_w_pairs(const float *x, const float *dout, const int *num_pairs_dev, int max_pairs, const int *pair_e0,
                             const int *pair_e1, const int *col, const int *tgt, int num_filters, const float *pair_dist,
                             float cutoff, float *dWp, float *gmax, void *stream);
/* conan_cfconv_bwd_x and conan_cfconv_bwd_w_pairs in ONE pass over the by-source CSR (round 6, ABI 5; num_filters = 128): while source j walks its
 * edges for dx[j], an edge that is its pair's e0 has x[j], dout[j] and dout[target] at hand and gathers x[target] to write the pair row — one extra
 * row gather per pair instead of four, one launch less on the backward chain; the expression is conan_cfconv_bwd_w_pairs' (same bits).  gmax (nullable):
 * raised to max |dWp|; the caller has cleared it — conan_cfconv_fwd's zero_slot in the forward pass. */
int conan_cfconv_bwd_xw_pairs_supported(int num_filters);
int conan_cfconv_bwd_xw_pairs(const float *W, const float *x, const float *dout, const int *t_rowptr, const int *t_eid, const int *tgt, const int *pid,
                              const int *pair_e0, const int *pair_e1, const float *pair_dist, float cutoff, int num_atoms, int num_filters, float *dx,
                              float *dWp, float *gmax, void *stream);
/* dist (nullable) + cutoff: additionally multiply row e by 0.5*(cos(dist[e]*pi/cutoff)+1), i.e. return the gradient with
 * respect to the filter BEFORE the cosine cutoff. */
int conan_cfconv_bwd_w(const float *x, const float *dout, const int *num_edges_dev, int max_edges, const int *col,
                       const int *tgt, int num_filters, const float *dist, float cutoff, float *dW, void *stream);

/* Sum readout per conformer graph: out[g,:] = sum_{a in graph g} x[a,:]  (SumAggregation; schnet_no_sum.py:183,353). */
int conan_segment_sum_fwd(const float *x, const int *graph_ptr, int num_graphs, int width, float *out, void *stream);
/* dx[a,:] = dout[graph(a),:] */
int conan_segment_sum_bwd(const float *dout, const int *graph_ptr, int num_graphs, int width, float *dx, void *stream);

/* ---------------------------------------------------------------------------------------------- ViSNet (forward)
 * Everything of the vendored ViSNet (conan_fgw/src/model/graph_embeddings/torch_geometric_visnet.py) that is not a plain
 * Linear layer; Linear layers run on conan_linear_fwd (act 3 = SiLU).  Edges: CSR by target INCLUDING self loops
 * (Distance(add_self_loops=True), :331-347 => conan_radius_graph_csr(..., loop=1)); vec tensors are [n,3,H]. */

/* d_ij[e,3] = (pos[src]-pos[tgt])/|.|, 0 for self loops (:340-347, :864-866; Sphere(lmax=1) = identity, :132-189). */
int conan_visnet_edge_unit(const float *pos, const int *col, const int *tgt, const int *num_edges_dev, int max_edges,
                           float *dvec, void *stream);
/* ExpNormalSmearing (:100-111): out[e,k] = C(d_e) * exp(-betas[k] * (exp(-alpha d_e) - means[k])^2), C = 0 for d >= cutoff. */
int conan_visnet_expnormal(const float *dist, const int *num_edges_dev, int max_edges, const float *means,
                           const float *betas, int num_rbf, float alpha, float cutoff, float *out, void *stream);
/* NeighborEmbedding (:408-415): W[e,:] *= C(d_e) for src != tgt, 0 for self loops (in place). */
int conan_visnet_neighbor_scale(float *W, const float *dist, const int *col, const int *tgt, const int *num_edges_dev,
                                int max_edges, int H, float cutoff, void *stream);
/* Same, out of place: out[e,:] = W[e,:] * C(d_e) (out may be W).  The autograd wrapper uses this form: no clone of the worst-case-sized buffer. */
int conan_visnet_neighbor_scale_to(const float *W, const float *dist, const int *col, const int *tgt, const int *num_edges_dev,
                                   int max_edges, int H, float cutoff, float *out, void *stream);
/* out[r,:] = [a[r,:Ha] | b[r,:Hb]]   (torch.cat(dim=1), :419, :945). */
int conan_concat2(const float *a, int Ha, const float *b, int Hb, long long rows, float *out, void *stream);
/* EdgeEmbedding (:463-465): f[e,:] = (x[tgt] + x[src]) * p[e,:]. */
int conan_visnet_edge_embed(const float *x, const float *p, const int *col, const int *tgt, const int *num_edges_dev,
                            int max_edges, int H, float *f, void *stream);
/* torch.nn.LayerNorm over the last dimension (:583, :883). */
int conan_layernorm_fwd(const float *x, const float *gamma, const float *beta, int rows, int H, float eps, float *out,
                        void *stream);
/* out[r,c] = v[r,c] * w[c]   (VecLayerNorm with norm_type=None, :262-268). */
int conan_scale_channels(const float *v, const float *w, long long rows, int H, float *out, void *stream);
/* out = v * w[channel] + add (all [rows,H]): the backward of a channel scaling whose input is also used as a residual — the residual's gradient `add`
 * joins in the same pass instead of an autograd sum (VecLayerNorm's weight next to ViS_MP's vec + dvec, torch_geometric_visnet.py:587,660). */
int conan_scale_channels_add(const float *v, const float *w, const float *add, long long rows, int H, float *out, void *stream);
/* vec_dot[a,c] = sum_sp vp[a,sp,c] * vp[a,sp,H+c], vp = vec_proj(vec) [n,3,3H] (:605-607). */
int conan_visnet_vecdot(const float *vp, int n, int H, float *out, void *stream);
/* ViS_MP.message (scalar half) + aggregate (:632-645, :671): attn_h = SiLU(sum_{c in head} q_i k_j dk_e) * C(r_e);
 * vmsg[e,:] = v_j * dv_e * attn_h;  xagg[i,:] = sum_{e in row i} vmsg[e,:].  H <= 64 or H == 128.
 * pre_act != 0: dk / dv are the PRE-activations of dk_proj / dv_proj (:623-624) and act (SiLU) is applied as they are loaded — the
 * activated [E,H] tensors then never exist in HBM (the backward returns the gradients w.r.t. the pre-activations, likewise). */
int conan_visnet_attn_message(const float *q, const float *k, const float *v, const float *dk, const float *dv,
                              const int *rowptr, const int *col, const float *dist, float cutoff, int n, int H,
                              int num_heads, int pre_act, float *vmsg, float *xagg, void *stream);
/* ViS_MP.message (vector half) + aggregate (:646-653, :672): vagg[i,sp,:] = sum_e vec[src,sp,:]*s1_e + s2_e*d_e[sp], s=[s1|s2]. */
/* pre_act != 0 (here and in conan_visnet_edge_update and the two backward entry points): s / t are the PRE-activations of s_proj /
 * f_proj; SiLU is applied as they are loaded and the gradients returned for them are w.r.t. the pre-activations (see conan_visnet_attn_message). */
int conan_visnet_vec_aggregate(const float *vec, const float *s, const float *dvec, const int *rowptr, const int *col,
                               int n, int H, int pre_act, float *vagg, void *stream);
/* Residual node update (:621-625, :873-881): x' = x + vec_dot*o2 + o3; vec' = vec + vec3*o1 + vagg; o=[o1|o2|o3], vec3=vp[:,:,2H:]. */
int conan_visnet_node_update(const float *x, const float *vec, const float *vdot, const float *o, const float *vp,
                             const float *vagg, int n, int H, float *x_out, float *vec_out, void *stream);
/* ViS_MP.edge_update (:655-661) with the node-side projections hoisted: wt = w_trg_proj(vec), ws = w_src_proj(vec) [n,3,H],
 * t = SiLU(f_proj(f)) [E,H]:  f'[e] = f[e] + t[e] * sum_sp rej(wt[tgt],d)[sp] * rej(ws[src],-d)[sp]. */
int conan_visnet_edge_update(const float *wt, const float *ws, const float *t, const float *dvec, const int *col,
                             const int *tgt, const int *num_edges_dev, int max_edges, int H, int pre_act, const float *f,
                             float *f_out, void *stream);
/* GatedEquivariantBlock pieces (:942-960): |v|_2 over the spatial axis; gated split of update_net's output. */
int conan_visnet_spatial_norm(const float *v, int n, int H, float *out, void *stream);
int conan_visnet_gate(const float *u, const float *v2, int n, int out_channels, int scalar_activation, float *x_out,
                      float *v_out, void *stream);
/* x * std + atomref[z]   (visnet.py:147-156; Atomref :1051-1058; std is a device scalar). */
int conan_visnet_prior(const float *x, const int64_t *z, const float *atomref, const float *std_dev, int n,
                       int out_channels, float *out, void *stream);

/* ---------------------------------------------------------------------------------------------- ViSNet (backward)
 * Gradients of the kernels above (autograd of torch_geometric_visnet.py).  Node-level gradients fed by many edges are
 * accumulated without atomics: one wavefront per node walks its CSR row (target side) or its by-source list
 * (t_rowptr / t_eid from conan_csr_transpose) in a fixed order. */
int conan_silu_fwd(const float *x, int rows, int width, const int *m_dev, float *y, void *stream);
int conan_silu_bwd(const float *x, const float *dy, int rows, int width, const int *m_dev, float *dx, void *stream);
/* inverse of conan_concat2 (backward of torch.cat). */
int conan_split2(const float *in, int Ha, int Hb, long long rows, float *a, float *b, void *stream);
int conan_rowsum(const float *x, int rows, int width, float *out, void *stream);
int conan_scale_scalar(const float *x, const float *scale_dev, long long count, float *out, void *stream);
/* EdgeEmbedding: dp[e] = (x_i+x_j)*df[e]; dx[i] = sum over the edges incident to i (either side) of df*p. */
int conan_visnet_edge_embed_bwd(const float *x, const float *p, const float *df, const int *rowptr, const int *col,
                                const int *tgt, const int *t_rowptr, const int *t_eid, const int *num_edges_dev,
                                int max_edges, int n, int H, float *dp, float *dx, void *stream);
/* LayerNorm: dx, dgamma, dbeta (deterministic chunked reduction); ws holds conan_layernorm_bwd_ws(rows, H) floats. */
long long conan_layernorm_bwd_ws(int rows, int H);
int conan_layernorm_bwd(const float *x, const float *gamma, const float *dy, int rows, int H, float eps, float *dx,
                        float *dgamma, float *dbeta, float *ws, void *stream);
/* ... with a second gradient of x (`dres` [rows,H]: x's residual use next to the LayerNorm, torch_geometric_visnet.py:583,659) added to dx in the same pass. */
int conan_layernorm_bwd_res(const float *x, const float *gamma, const float *dy, const float *dres, int rows, int H, float eps, float *dx,
                            float *dgamma, float *dbeta, float *ws, void *stream);
/* dvp [n,3,3H] = [dout*vec2 | dout*vec1 | 0]. */
int conan_visnet_vecdot_bwd(const float *vp, const float *dout, int n, int H, float *dvp, void *stream);
/* Attention message: given dvmsg [E,H] and dxagg [n,H] returns dq, dk, dv [n,H] and d(dk), d(dv) [E,H]. */
int conan_visnet_attn_message_bwd(const float *q, const float *k, const float *v, const float *dk, const float *dv,
                                  const float *dvmsg, const float *dxagg, const int *rowptr, const int *col,
                                  const int *tgt, const int *t_rowptr, const int *t_eid, const float *dist, float cutoff,
                                  int n, int H, int num_heads, int pre_act, float *dq, float *dkn, float *dvn, float *ddk,
                                  float *ddv, void *stream);
/* Vector aggregate: ds [E,2H], dvec [n,3,H] from dvagg [n,3,H]. */
int conan_visnet_vec_aggregate_bwd(const float *vec, const float *s, const float *dvec3, const float *dvagg,
                                   const int *col, const int *tgt, const int *t_rowptr, const int *t_eid,
                                   const int *num_edges_dev, int max_edges, int n, int H, int pre_act, float *ds,
                                   float *dvec, void *stream);
/* Node update: dvdot [n,H], do [n,3H], dvp [n,3,3H] = [0|0|dvec_out*o1] (dx = dx_out, dvec = dvagg = dvec_out).  dvdot == NULL: vdot was formed from this vp
 * (conan_visnet_vecdot) and its backward is folded in: dvp = [g*vec2 | g*vec1 | dvec_out*o1] with g = dx_out*o2, nothing else to add to it. */
int conan_visnet_node_update_bwd(const float *dxo, const float *dveco, const float *vdot, const float *o, const float *vp,
                                 int n, int H, float *dvdot, float *dout_o, float *dvp, void *stream);
/* Edge update: dwt, dws [n,3,H] and dt [E,H] from df_out (df = df_out). */
int conan_visnet_edge_update_bwd(const float *wt, const float *ws, const float *t, const float *dvec3, const float *dfo,
                                 const int *rowptr, const int *col, const int *tgt, const int *t_rowptr, const int *t_eid,
                                 int n, int H, int pre_act, float *dwt, float *dws, float *dt, void *stream);
int conan_visnet_spatial_norm_bwd(const float *v, const float *dout, int n, int H, float *dv, void *stream);
int conan_visnet_gate_bwd(const float *u, const float *v2, const float *dxo, const float *dvo, int n, int out_channels,
                          int scalar_activation, float *du, float *dv2, void *stream);

/* ---------------------------------------------------------------------------------------------- FGW barycenter */

/* Glue of _compute_barycenter (schnet_no_sum.py:242-252 with :41-87; visnet.py:168-176 with :32-79):
 * to_dense_batch + shift + normalize_tensor(.,a,b) per conformer slab (min/max over the WHOLE padded [N,d] slab,
 * barycenter.py:393-399) and to_dense_adj.  Ys[G,N,d], Cs[G,N,N] (Cs[g, src, tgt] = multiplicity of edge src->tgt),
 * minmax[G,2] saved for the backward.  Cs == NULL: features only (rowptr / col are then unused) — the structure goes to
 * conan_fgw_barycenter_fwd_ragged as neighbour lists. */
int conan_fgw_densify(const float *feat, const int *graph_ptr, const int *rowptr, const int *col, int num_graphs,
                      int N, int d, float shift, float a, float b, float *Ys, float *Cs, float *minmax, void *stream);
/* Backward of the feature half of conan_fgw_densify (autograd through +shift, min(), max() and the affine map). */
int conan_fgw_densify_bwd(const float *feat, const float *dYs, const int *graph_ptr, const float *minmax,
                          int num_graphs, int N, int d, float shift, float a, float b, float *dfeat, void *stream);

typedef struct conan_fgw_params {
    float alpha;            /* trade-off structure/features          (schnet_no_sum.py:289: 0.1) */
    float epsilon;          /* entropic regularisation                (:294: 0.1) */
    int max_iter;           /* outer AND inner PGD iteration cap       (:297: 5; barycenter.py:134 passes it on) */
    float tol;              /* outer stop on ||Y-Yprev||, ||C-Cprev||  (:298: 1e-2) */
    float inner_tol;        /* PGD stop on ||T-Tprev||                 (barycenter.py:135: 1e-4) */
    int num_iter_max;       /* Sinkhorn iteration cap                  (:299: 5) */
    float stop_thr;         /* Sinkhorn marginal-violation threshold   (:300: 1e-2) */
    int fixed_structure;    /* keep C = init_C                         (:291) */
    int fixed_features;     /* keep Y = init_Y                         (:292) */
    int warmstart;          /* warmstartT: start each coupling solve from the previous outer iteration's T (:285) */
    int loss_fun;           /* 0 = "square_loss" (:295, every model), 1 = "kl_loss" (utils.py:20-32,76-87) */
    int cs_small_int;       /* promise of the caller: every entry of Cs is an integer in [0, 255] (to_dense_adj output: 0/1, or the
                             * multiplicity of a repeated edge) — lets the N <= 64 kernel keep Cs as bytes in LDS (5 instead of 4
                             * workgroups per CU at N = 33).  0 = arbitrary floats (fgw_barycenters called with general matrices). */
} conan_fgw_params;

/* Workspace size in BYTES for conan_fgw_barycenter_fwd. */
long long conan_fgw_workspace_bytes(int B, int K, int N, int d);

/* Batched fgw_barycenters (barycenter.py:7-225 -> bregman.py:70-167 -> sinkhorn.py:318-450 -> utils.py), one
 * independent problem per molecule, all K input graphs of a molecule padded to the same N nodes (the production glue
 * always does; schnet_no_sum.py:281-306).  Replaces the Python loop over molecules (schnet_no_sum.py:259-312).
 * Ys[B,K,N,d], Cs[B,K,N,N], ps[B,K,N] or NULL (uniform), p[B,N] or NULL (uniform), lambdas[K] or NULL (1/K),
 * init_C[B,N,N] or NULL (= Cs[b,0], schnet_no_sum.py:303), init_Y[B,N,d] or NULL (zeros).
 * Entries of ps / p may be ZERO: a node without mass takes no part in the problem (its row / column of every coupling, its row of Y and
 * its row / column of C are exactly zero).  That is how input graphs with fewer than N nodes, or a barycenter with fewer nodes than its
 * inputs (barycenter.py:50-67 takes any sizes), are passed: zero rows in Ys / Cs / init_C and zero weights (conan-fgw_amd/fgw.py does
 * this); couplings with massless nodes are solved on the log-domain path (flags bit 0).
 * Outputs: Y[B,N,d], C[B,N,N], T[B,K,N,N] (final couplings, saved for the backward),
 * T_iter[max_iter,B,K,N,N] or NULL: the couplings after every outer iteration (the reference's log["Ts_iter"], barycenter.py:196;
 * a molecule that stopped early keeps its last couplings in the later slots),
 * info[B,4] int32 = {outer iterations, total PGD iterations, total Sinkhorn iterations, flags}; flags bit 0: at least one coupling
 * solve of the molecule left the range of the scaling-form Sinkhorn and was redone on the exact log-domain path (same result
 * contract, slower); bit 1 (round 6): at least one coupling solve ran with the molecule's padded nodes merged into one node (the N - n padded nodes
 * of a conformer graph and of the barycenter are exchangeable: same iteration, same result contract, (n + 1)^3 instead of N^3 — DESIGN.md 3.3),
 * errs[B,2,max_iter] fp32 = err_feature / err_structure per outer iteration (NaN where not executed).
 * Internal arithmetic is fp64 (DESIGN.md section "FGW numerics"); I/O is fp32. */
int conan_fgw_barycenter_fwd(const float *Ys, const float *Cs, const float *ps, const float *p, const float *lambdas,
                             const float *init_C, const float *init_Y, int B, int K, int N, int d,
                             const conan_fgw_params *params /* (host) */, float *Y, float *C, float *T, float *T_iter,
                             int *info, float *errs, void *workspace, void *stream);
/* The same solve with the input graphs' structure read straight from the ragged neighbour lists instead of Cs[B,K,N,N] (SURVEY.md 2.2 / 7:
 * the reference materialises to_dense_adj per conformer, schnet_no_sum.py:249-252; here no [G,N,N] tensor exists): graph g = b * K + s owns the
 * nodes graph_ptr[g] .. graph_ptr[g+1]-1 and the edges rowptr[lo] .. rowptr[lo+n]-1 of the by-target CSR that conan_radius_graph_csr leaves
 * (col = source, tgt = target of every edge; n clamped to N as in conan_fgw_densify); structure entry [source][target] = multiplicity of the edge.
 * The coupling kernels of the model path (square loss; N <= 64, or N > 64 within the LDS budget) build the adjacency counts as bytes in LDS in
 * their load stage; any other shape / loss, and the exact second pass of a flagged coupling, expand graphs into a scratch behind the regular
 * workspace (conan_fgw_workspace_bytes_ragged; untouched otherwise).  cs_small_int is implied.  Same outputs, same arithmetic: results equal
 * conan_fgw_barycenter_fwd on the densified inputs bit for bit. */
long long conan_fgw_workspace_bytes_ragged(int B, int K, int N, int d);
int conan_fgw_barycenter_fwd_ragged(const float *Ys, const int *graph_ptr, const int *rowptr, const int *col, const int *tgt, const float *ps,
                                    const float *p, const float *lambdas, const float *init_C, const float *init_Y, int B, int K, int N, int d,
                                    const conan_fgw_params *params /* (host) */, float *Y, float *C, float *T, float *T_iter, int *info,
                                    float *errs, void *workspace, void *stream);

/* dYs[b,s,j,:] = lambdas[s] * sum_i T[b,s,i,j] * (1/p[b,i]) * dY[b,i,:]  — the whole backward of the block given the
 * saved couplings (the reference solves them under torch.no_grad(), barycenter.py:120). */
int conan_fgw_barycenter_bwd(const float *T, const float *dY, const float *p, const float *lambdas, int B, int K,
                             int N, int d, float *dYs, void *stream);

/* F_bary readout of _compute_barycenter: out[b*K + k, :] = sum_i post(Y[b])[i, :] for k < K
 * (schnet_no_sum.py:308-312).  mode 0 = SchNet (post = identity); mode 1 = ViSNet (NaN guard -> zeros, then column
 * L2 normalisation over the N rows, visnet.py:233-242). */
int conan_fgw_readout_fwd(const float *Y, int B, int K, int N, int d, int mode, float *out, void *stream);
int conan_fgw_readout_bwd(const float *Y, const float *dout, int B, int K, int N, int d, int mode, float *dY,
                          void *stream);

/* Elementwise activations: op 0 = ReLU, 1 = sigmoid (classification head: build_mlp_class + torch.sigmoid,
 * schnet_based_models.py:31-45,367), 2 = shifted softplus (PyG ShiftedSoftplus as a stand-alone module, `SchNetNoSum.act`,
 * schnet_no_sum.py:178,227,231 — on the model path it is fused into conan_linear_fwd).  The backward takes the forward
 * OUTPUT y (relu' = [y > 0], sigmoid' = y (1 - y), ssp' = 1 - 0.5 exp(-y)). */
int conan_unary_fwd(const float *x, long long count, int op, float *y, void *stream);
int conan_unary_bwd(const float *y, const float *dy, long long count, int op, float *dx, void *stream);

/* buf[r,:] = 0 for r in [*m_dev, rows): defines the tail of a worst-case-sized edge buffer without clearing all of it. */
int conan_zero_tail(float *buf, const int *m_dev, int rows, int width, void *stream);

/* ---------------------------------------------------------------------------------------------- covalent (GAT) branch
 * GATBased (conan_fgw/src/model/graph_embeddings/gat.py:5-25): two PyG-2.3.0 GATConv layers (heads = 1, edge_dim = 3,
 * add_self_loops with fill_value "mean", negative_slope 0.2) on the 2-D bond graph + sum readout; called from
 * EmbeddingsWithGATAggregationBaryCenter.forward (schnet_based_models.py:166-168).  SURVEY.md 8(f) rank 1. */

/* Bond graph from PyG's edge_index[2,E] (int64; row 0 = source, row 1 = target; any order; self loops dropped):
 * CSR by target (rowptr[n+1], col[E] = source, eid[E] = original edge id) and by source (t_rowptr[n+1], t_pos[E] = position of
 * the edge in the by-target arrays, t_tgt[E] = its target); rows sorted => deterministic layout.  ws: 2*(n+1) ints. */
int conan_bond_graph_csr(const int64_t *edge_index, int num_edges, int num_nodes, int *ws, int *rowptr, int *col, int *eid,
                         int *t_rowptr, int *t_pos, int *t_tgt, void *stream);
/* v[edge_dim] = lin_edge.weight^T att_edge, so that <lin_edge(ea), att_edge> = <ea, v> (GATConv.edge_update). */
int conan_gat_edge_vec(const float *w_edge, const float *att_edge, int channels, int edge_dim, float *v, void *stream);
int conan_gat_edge_vec_bwd(const float *w_edge, const float *att_edge, const float *dv, int channels, int edge_dim,
                           float *dw_edge, float *datt_edge, void *stream);
/* a_src[i] = <h_i, att_src>, a_dst[i] = <h_i, att_dst>   (h = lin_src(x), shared with lin_dst). */
int conan_gat_node_alpha(const float *h, const float *att_src, const float *att_dst, int n, int channels, float *a_src,
                         float *a_dst, void *stream);
/* out_i = sum_{j in N(i) + {i}} alpha_ji h_j + bias; alpha = softmax_i(leaky_relu(a_src[j] + a_dst[i] + <ea_ji, v>)), the self
 * loop's ea is the mean of the incoming edge attributes.  alpha[E] (by-target order) and alpha_self[n] are saved for backward. */
int conan_gat_aggregate_fwd(const float *h, const float *a_src, const float *a_dst, const int *rowptr, const int *col,
                            const int *eid, const float *edge_attr, int edge_dim, const float *v, const float *bias,
                            float negative_slope, int n, int channels, float *out, float *alpha, float *alpha_self, void *stream);
/* Backward of the aggregation: dh[n,C] (messages + attention projections) and dparams[3C + edge_dim] = d att_src | d att_dst |
 * d bias | dv (v of conan_gat_edge_vec).  ws: conan_gat_bwd_ws(n, E, C, edge_dim) floats.  Per-edge gradients are formed per
 * target row, scattered sums per by-source list, parameter gradients as per-wavefront partials summed in a fixed order:
 * no float atomics, bitwise reproducible.  channels <= 256. */
long long conan_gat_bwd_ws(int n, int num_edges, int channels, int edge_dim);
int conan_gat_aggregate_bwd(const float *h, const float *dout, const float *alpha, const float *alpha_self, const float *a_src,
                            const float *a_dst, const float *att_src, const float *att_dst, const int *rowptr, const int *col,
                            const int *eid, const int *t_rowptr, const int *t_pos, const int *t_tgt, const float *edge_attr,
                            int edge_dim, const float *v, float negative_slope, int n, int num_edges, int channels, float *ws,
                            float *dh, float *dparams, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* CONAN_FGW_HIP_H */
