/*
 * simspread_hip.h -- C ABI of libsimspread_hip.so (MI355X / gfx950).
 *
 * Drop-in boundary for the SimSpread.jl hot path
 *     featurize -> construct -> spread -> predict -> clean!
 * The reference (cvigilv/SimSpread.jl, pure Julia) has no FFI of its own; its
 * boundary is the method table exported at src/SimSpread.jl:21-56.  Each entry
 * point below names the reference method it stands behind (file:line under the
 * reference tree).  Julia binds these with `ccall` (see INTEGRATION.md and
 * julia/SimSpreadHIP.jl), this repository's Python mirror binds them with ctypes.
 *
 * Conventions
 *   - plain C types only; no C++/torch types cross this boundary;
 *   - every function returns SS_OK (0) or a negative SS_E* code and never throws or
 *     aborts; ss_last_error() returns a thread-local message for the last failure;
 *   - one process drives one GPU (ss_init(device)); multi-GPU = one process per GPU,
 *     rows/folds sharded by the host layer, RCCL only for the final score gather
 *     (ss_comm_init / ss_gather_rows_*);
 *   - thread safety: state is handle-scoped.  Calls on DIFFERENT handles may come from several host threads /
 *     Julia tasks at once and overlap; calls on the same handle queue up on that handle's lock; ss_init /
 *     ss_shutdown / ss_set_stream / ss_reset_stream exclude everything else while they run.  ss_last_error,
 *     ss_timing_last and ss_path_last are per host thread;
 *   - `mem` says where caller buffers live: SS_MEM_HOST (copied during the call) or
 *     SS_MEM_DEVICE (used in place, e.g. a torch tensor's data_ptr()); the caller
 *     keeps ownership of every buffer it passes; the library owns what is behind
 *     the opaque handles;
 *   - CSR inputs: int64 row pointers, int32 column indices, `index_base` 0 or 1
 *     (Julia's SparseMatrixCSC is the 1-based CSR of the transpose), column
 *     indices sorted within each row; a NULL value pointer means "all ones";
 *     explicitly stored zeros are dropped (degree = number of NON-ZEROS,
 *     src/graphs.jl:9-11);
 *   - dense inputs are column-major with a leading dimension, like Julia arrays;
 *   - score blocks are written in the layout the caller asks for:
 *       SS_LAYOUT_ROWMAJOR  (r,t) at out[r*ld + t]   (numpy C order; native, no extra pass)
 *       SS_LAYOUT_COLMAJOR  (r,t) at out[r + t*ld]   (Julia Matrix; one device transpose).
 */
#ifndef SIMSPREAD_HIP_H
#define SIMSPREAD_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SS_VERSION 100 /* 0.1.0 */

enum {
  SS_OK = 0,
  SS_EINVAL = -1,       /* bad argument / shape / unsorted or out-of-range indices */
  SS_ENOMEM = -2,       /* device or host allocation failed */
  SS_EHIP = -3,         /* a HIP runtime call failed (message has the HIP error string) */
  SS_ENODEV = -4,       /* no usable gfx950 device / ss_init not called */
  SS_EUNSUPPORTED = -5  /* valid request this build cannot serve (e.g. nnz >= 2^31) */
};

enum { SS_MEM_HOST = 0, SS_MEM_DEVICE = 1 };
enum { SS_ROWS_QUERY = 0, SS_ROWS_SOURCE = 1 };
enum { SS_LAYOUT_ROWMAJOR = 0, SS_LAYOUT_COLMAJOR = 1 };

typedef struct ss_graph ss_graph; /* tri-partite query/source/feature/target graph, device resident */
typedef struct ss_spmat ss_spmat; /* one sparse operand W of F = W*R, device resident            */

/* ---------------------------------------------------------------- runtime ---- */
int ss_version(void);
/* Hash of the kernel sources (the .hip and .hpp files of csrc/) this library was built from -- what profiles/ records next to its
 * counters; a loader that sees the sources can tell a stale library from a current one. */
const char* ss_source_hash(void);
const char* ss_last_error(void);
int ss_device_count(void);
/* Select the GPU this process drives and create the library's stream.  Replaces the
 * reference's `GPU::Bool` switch (src/core.jl:402,404,446,448). */
int ss_init(int device);
int ss_shutdown(void);
/* Run on the caller's HIP stream (e.g. torch's current stream) so that kernels that produced
 * SS_MEM_DEVICE inputs are ordered before the library's.  NULL is HIP's null (default) stream.
 * The caller keeps ownership of the stream.  ss_reset_stream() returns to the library's own
 * (non-blocking) stream. */
int ss_set_stream(void* hip_stream);
int ss_reset_stream(void);
int ss_synchronize(void);
/* Timings of the last predict/spmm call, milliseconds, measured with hipEvents on the
 * library stream: ms[0] whole call on device, [1] transfer stage (stage 1), [2] W*R SpMM
 * (stage 2), [3] epilogues/transposes, [4] host->device, [5] device->host,
 * [6] number of SpMM launches, [7] number of stage-1 launches.  Writes min(n,8) values. */
int ss_timing_last(double* ms, int n);
/* Which kernels the last predict / spmm call of this host thread went through: a comma-separated list of tags
 * ("transfer", "transfer_loo", "transfer_dense_bf16_ring", "transfer_dense_bf16_128", "transfer_dense_f32_mfma",
 * "spmm_sell", "spmm_sell_sorted", "spmm_csell", "spmm_colgroup", "spmm_chunked_narrow", ...), NUL-terminated, truncated
 * to n - 1 characters.  Lets a caller (and the parity tests) assert that a size-dependent routing decision was the one
 * expected.  Has no counterpart in the reference (its only switch is GPU::Bool, src/core.jl:402,404). */
int ss_path_last(char* buf, int n);
/* enable != 0: from now on the timings of successive calls add up (ss_timing_last returns the sums and launch
 * counts since the hold began) instead of replacing one another, so that a benchmark loop need not stop after
 * every call to read them; enable == 0: back to per-call timings. */
int ss_timing_hold(int enable);

/* ------------------------------------------------------------ final score gather ---- */
/* The one exchange of the multi-GPU path (north star: "RCCL over xGMI only for the final score gather"; the reference has
 * no counterpart -- no NCCL/MPI anywhere).  One process per GPU.  Rank 0 fills `id` (128 bytes) with ss_comm_unique_id,
 * the host framework hands those bytes to the other processes (MPI, Distributed.jl, torch.distributed: any channel), every
 * process calls ss_comm_init(id, rank, nranks) after ss_init(device).  RCCL is dlopen'ed at that point; a process that
 * never calls these functions never touches it.
 * ss_gather_rows_*: rank r holds counts[r] finished rows of the score matrix (`local`, counts[r] x ncols, row-major,
 * DEVICE memory); the ranks exchange them directly -- one receive per peer straight into its slice of `full`
 * (sum(counts) x ncols, row-major, DEVICE memory), one send per peer, one ncclGroup on the library stream -- so all
 * point-to-point xGMI links carry traffic at once.  root < 0: every rank receives the full matrix; root = r: only rank r
 * does (`full` may be NULL elsewhere).  Stream-ordered like a kernel launch: ss_synchronize() waits for it. */
int ss_comm_unique_id(char id[128]);
int ss_comm_init(const char id[128], int rank, int nranks);
int ss_comm_destroy(void);
int ss_comm_info(int* rank, int* nranks); /* nranks = 0 when no communicator exists */
int ss_gather_rows_f32(const float* local, int64_t ncols, const int64_t* counts, float* full, int root);
int ss_gather_rows_f64(const double* local, int64_t ncols, const int64_t* counts, double* full, int root);

/* ------------------------------------------------------ similarity producer -- */
/* The step before featurize in the reference's tutorial (docs/src/tutorial/fishers-flowers.jl:66,
 * `1 .- pairwise(Jaccard(), X, dims=1)`): S[i][j] = sum_k min(F[i,k],F[j,k]) / sum_k max(F[i,k],F[j,k]) between the
 * rows of the n x d feature matrix F (column-major, ld >= n); S is n x n column-major (lds >= n), symmetric with
 * unit diagonal; two all-zero rows have similarity 1 (Distances.jl: distance 0). */
int ss_similarity_jaccard_f32(const float* F, int64_t n, int64_t d, int64_t ld, float* S, int64_t lds, int mem);
int ss_similarity_jaccard_f64(const double* F, int64_t n, int64_t d, int64_t ld, double* S, int64_t lds, int mem);

/* ------------------------------------------------------- cutoff / k / spread -- */
/* cutoff(X, alpha, weighted): out = x >= alpha ? (weighted ? x : 1) : 0, element-wise
 * (src/core.jl:37-43,55-60; `>=` inclusive per test/runtests.jl:46-47).  Also the value
 * part of featurize (src/core.jl:106-112).  rows x cols, column-major, ld >= rows. */
int ss_cutoff_f32(const float* X, int64_t rows, int64_t cols, int64_t ld, float alpha,
                  int weighted, float* out, int64_t ldo, int mem);
int ss_cutoff_f64(const double* X, int64_t rows, int64_t cols, int64_t ld, double alpha,
                  int weighted, double* out, int64_t ldo, int mem);
/* k(G): number of non-zeros in every row (src/graphs.jl:9-11). */
int ss_row_degree_f32(const float* G, int64_t rows, int64_t cols, int64_t ld, int64_t* deg, int mem);
int ss_row_degree_f64(const double* G, int64_t rows, int64_t cols, int64_t ld, int64_t* deg, int mem);
/* spread(G): W[i,j] = G[i,j] / k(i), rows of degree 0 give 0 (src/core.jl:365-371). */
int ss_spread_f32(const float* G, int64_t rows, int64_t cols, int64_t ld, float* W, int64_t ldw, int mem);
int ss_spread_f64(const double* G, int64_t rows, int64_t cols, int64_t ld, double* W, int64_t ldw, int mem);

/* ------------------------------------------------------------ graph handles --- */
/* construct(...) (src/core.jl:148-201,217-276,294-296,308-337): instead of the dense
 * N x N block matrices A and B the handle keeps the three non-zero blocks
 *     Xq = A[queries, features]  (nq x nf)    Xs = A[sources, features]  (ns x nf)
 *     Ys = A[sources, targets]   (ns x nt)
 * as CSR on the device together with their transposes and the count degrees kf, ks, kt
 * of B (spread, src/core.jl:365-371).  nq may be 0 (3-layer graph of src/core.jl:308-337). */
int ss_graph_create_csr_f32(int64_t nq, int64_t ns, int64_t nf, int64_t nt,
                            const int64_t* xq_ptr, const int32_t* xq_idx, const float* xq_val,
                            const int64_t* xs_ptr, const int32_t* xs_idx, const float* xs_val,
                            const int64_t* ys_ptr, const int32_t* ys_idx, const float* ys_val,
                            int index_base, int mem, ss_graph** out);
int ss_graph_create_csr_f64(int64_t nq, int64_t ns, int64_t nf, int64_t nt,
                            const int64_t* xq_ptr, const int32_t* xq_idx, const double* xq_val,
                            const int64_t* xs_ptr, const int32_t* xs_idx, const double* xs_val,
                            const int64_t* ys_ptr, const int32_t* ys_idx, const double* ys_val,
                            int index_base, int mem, ss_graph** out);
/* Same graph from dense column-major blocks; the similarity cutoff of featurize
 * (src/core.jl:106-112) is applied on the device while the CSR is assembled when
 * apply_cutoff != 0 (Sq, Ss raw similarities), otherwise non-zeros are kept as they are.
 * Y (ns x nt) keeps its non-zero values. */
int ss_graph_create_dense_f32(int64_t nq, int64_t ns, int64_t nf, int64_t nt,
                              const float* Sq, int64_t ldq, const float* Ss, int64_t lds,
                              const float* Y, int64_t ldy, int apply_cutoff, float alpha,
                              int weighted, int mem, ss_graph** out);
int ss_graph_create_dense_f64(int64_t nq, int64_t ns, int64_t nf, int64_t nt,
                              const double* Sq, int64_t ldq, const double* Ss, int64_t lds,
                              const double* Y, int64_t ldy, int apply_cutoff, double alpha,
                              int weighted, int mem, ss_graph** out);
/* Dense-similarity regime (thresholded similarity too full for CSR, e.g. 90 % of 50k x 50k): the raw
 * similarities stay dense on the device and featurize's cutoff (src/core.jl:106-112) is applied inside
 * the stage-1 product, which runs on the matrix cores (bf16 MFMA over exact bf16 planes of the fp32 operands;
 * SS_DENSE_BF16=0: fp32-input MFMA); the labels Y stay sparse (CSR, ns x nt).
 * Sq (nq x ns) and Ss (ns x ns) are column-major raw similarities whose columns are the features named
 * after the sources; nq may be 0.  Serves ss_predict_f32 (query and source rows), ss_predict_loo_f32 and
 * ss_predict_kfold_f32.
 * _f32: bf16 matrix cores on exact bf16 planes (the reference's GPU=true precision, src/core.jl:404); _f64: the fp64
 * matrix instruction (the reference's default precision, src/core.jl:402 GPU=false). */
int ss_graph_create_similarity_f32(int64_t nq, int64_t ns, int64_t nt,
                                   const float* Sq, int64_t ldq, const float* Ss, int64_t lds,
                                   const int64_t* y_ptr, const int32_t* y_idx, const float* y_val,
                                   int index_base, float alpha, int weighted, int mem, ss_graph** out);
int ss_graph_create_similarity_f64(int64_t nq, int64_t ns, int64_t nt,
                                   const double* Sq, int64_t ldq, const double* Ss, int64_t lds,
                                   const int64_t* y_ptr, const int32_t* y_idx, const double* y_val,
                                   int index_base, double alpha, int weighted, int mem, ss_graph** out);
/* General form for caller-built adjacency matrices: predict accepts ANY named A, B
 * (src/core.jl:402-425; the reference's own test passes hand-written 9 x 9 matrices,
 * test/runtests.jl:120-158).  With n nodes, the caller passes
 *     L  = A[rows of y, :]        (nr x n)   the rows of A that are asked for
 *     Bm = B                      (n  x n)   the graph spread() normalises
 *     Wt = (B[:, cols of y])'     (nc x n)   the columns of B that are asked for, transposed
 * and ss_predict_*(g, SS_ROWS_QUERY, 0, nr, ...) returns the nr x nc block of A * spread(B)^2.
 * Degrees are the row non-zero counts of B.  SS_ROWS_SOURCE / leave-one-out do not apply. */
int ss_graph_create_general_f32(int64_t n, int64_t nr, int64_t nc,
                                const int64_t* l_ptr, const int32_t* l_idx, const float* l_val,
                                const int64_t* b_ptr, const int32_t* b_idx, const float* b_val,
                                const int64_t* w_ptr, const int32_t* w_idx, const float* w_val,
                                int index_base, int mem, ss_graph** out);
int ss_graph_create_general_f64(int64_t n, int64_t nr, int64_t nc,
                                const int64_t* l_ptr, const int32_t* l_idx, const double* l_val,
                                const int64_t* b_ptr, const int32_t* b_idx, const double* b_val,
                                const int64_t* w_ptr, const int32_t* w_idx, const double* w_val,
                                int index_base, int mem, ss_graph** out);
int ss_graph_destroy(ss_graph* g);
/* sizes[0..6] = nq, ns, nf, nt, nnz(Xq), nnz(Xs), nnz(Ys) after dropping stored zeros. */
int ss_graph_info(const ss_graph* g, int64_t sizes[7]);
/* Count degrees of the query-free graph B: kf[nf], ks[ns], kt[nt] (host buffers; any may be NULL). */
int ss_graph_degrees(const ss_graph* g, int64_t* kf, int64_t* ks, int64_t* kt);

/* ------------------------------------------------------------------ predict --- */
/* predict((A,B), y) / predict(A,B,y) / predict(A, ytrain) (src/core.jl:402-425,446-466):
 * the block of F = A * spread(B)^2 for rows [row_begin,row_end) of the query nodes
 * (SS_ROWS_QUERY) or of the source nodes (SS_ROWS_SOURCE; feature path + target path,
 * which is also what the 3-layer predict(A, ytrain) returns) and all nt targets.
 * clean != 0 fuses clean! (src/core.jl:478-484): column t becomes -99 when target t
 * has no edge in A.  out holds (row_end-row_begin) x nt scores in `layout`.
 * With mem == SS_MEM_DEVICE and SS_LAYOUT_ROWMAJOR the kernels write straight into `out` and the call returns
 * once they are enqueued (stream order, like a kernel launch): work queued later on the same stream sees the
 * scores, anything else waits with ss_synchronize().  Every other combination returns with `out` complete. */
int ss_predict_f32(ss_graph* g, int rows_kind, int64_t row_begin, int64_t row_end, int clean,
                   float* out, int64_t ld, int layout, int mem);
int ss_predict_f64(ss_graph* g, int rows_kind, int64_t row_begin, int64_t row_end, int clean,
                   double* out, int64_t ld, int layout, int mem);
/* Leave-one-out cross-validation: row i of the result is
 *   predict(construct(y, X, [source_i]), y[[source_i], :])   (+ clean! when clean != 0)
 * for i in [i_begin, i_end), i.e. construct's fold form (src/core.jl:148-201) with
 * one query per fold, computed from the resident graph by rank-1 degree corrections
 * instead of rebuilding a graph per fold.  Needs a graph with nq == 0 and ns == nf whose
 * feature column j is the one named after source j (the column construct drops,
 * src/core.jl:152).  Folds are independent: shard [i_begin,i_end) across processes. */
int ss_predict_loo_f32(ss_graph* g, int64_t i_begin, int64_t i_end, int clean,
                       float* out, int64_t ld, int layout, int mem);
int ss_predict_loo_f64(ss_graph* g, int64_t i_begin, int64_t i_end, int clean,
                       double* out, int64_t ld, int layout, int mem);

/* k-fold cross-validation in one call: fold_of_source[i] in [0, nfolds) for every source i; row i of the
 * result is what the reference's fold loop computes for source i when its fold is the query set,
 *   predict(construct(y, X, fold_members), y[fold_members, :])  (+ clean!)   (src/core.jl:148-201,402-423),
 * i.e. the members of a fold are removed from the sources AND their feature columns are dropped
 * (src/core.jl:152-153), degrees recounted.  Needs a graph with nq == 0 and ns == nf whose feature column j
 * is the one named after source j.  out is ns x nt.  With one source per fold this equals leave-one-out. */
int ss_predict_kfold_f32(ss_graph* g, const int32_t* fold_of_source, int nfolds, int clean,
                         float* out, int64_t ld, int layout, int mem);
int ss_predict_kfold_f64(ss_graph* g, const int32_t* fold_of_source, int nfolds, int clean,
                         double* out, int64_t ld, int layout, int mem);

/* Ranked evaluation without moving the scores ("next" row of the scope table: recallatL / precisionatL,
 * src/performance.jl:308-385): for every row of a row-major score block the L best columns in the order
 * sortperm(yhat, rev=true) gives (score descending, ties by ascending column; as Julia's isless orders floats, +0.0 ranks
 * before -0.0 -- SimSpread scores are sums of non-negative products and clean!'s -99, so -0.0 does not occur in them).
 * idx and val are nrows x L,
 * row-major; L <= 1024 and L <= ncols.  With the labels of those columns recall@L and precision@L follow on
 * the host from nrows*L numbers instead of nrows*ncols scores. */
int ss_topl_f32(const float* scores, int64_t nrows, int64_t ncols, int64_t ld, int L,
                int32_t* idx, float* val, int mem);

/* Threshold-free evaluation of one score vector on the device ("next" row of the scope table):
 * out[0] = AuROC, out[1] = AuPRC (src/performance.jl:49-89: confusion matrix at every unique score, a sample is
 * predicted positive when score >= threshold, trapezoidal rule over exactly those points), out[2] =
 * BEDROC(alpha) (src/performance.jl:22-38; ranks in sortperm(yhat, rev=true) order), out[3] = validity ratio
 * (share of non-zero scores, src/performance.jl:558-560).  y: n labels (0 = negative, anything else =
 * positive), yhat: n scores; both host or both device (mem); out is always host memory.  n < 2^31.
 * AuROC/AuPRC are NaN when a class is missing, as in the reference. */
int ss_rank_metrics_f32(const uint8_t* y, const float* yhat, int64_t n, double alpha, double out[4], int mem);

/* -------------------------------------------------------------- raw W*R SpMM --- */
/* The resource-spreading product F = W * R on its own (kernel unit tests and the
 * roofline benchmark; inside predict W = Ys' and R = the transfer block, src/core.jl:413).
 * W: rows x cols CSR.  R: cols x B dense, F: rows x B dense.
 *   *_layout == SS_LAYOUT_ROWMAJOR : R(k,b) at R[k*ldr + b], F(m,b) at F[m*ldf + b]
 *   *_layout == SS_LAYOUT_COLMAJOR : R(k,b) at R[k + b*ldr], F(m,b) at F[m + b*ldf]
 * B <= 64 with row-major operands takes the HBM-bound CSR kernel; wider B takes the
 * LDS-tiled kernel (natively column-major; other layouts pay one transpose). */
int ss_spmat_create_csr_f32(int64_t rows, int64_t cols, const int64_t* ptr, const int32_t* idx,
                            const float* val, int index_base, int mem, ss_spmat** out);
int ss_spmat_create_csr_f64(int64_t rows, int64_t cols, const int64_t* ptr, const int32_t* idx,
                            const double* val, int index_base, int mem, ss_spmat** out);
int ss_spmat_destroy(ss_spmat* w);
int ss_spmm_f32(ss_spmat* w, const float* R, int64_t B, int64_t ldr, int r_layout,
                float* F, int64_t ldf, int f_layout, int mem);
int ss_spmm_f64(ss_spmat* w, const double* R, int64_t B, int64_t ldr, int r_layout,
                double* F, int64_t ldf, int f_layout, int mem);
/* Algorithmic bytes of one ss_spmm call (SURVEY.md section 8d:
 * nnz*(vb+4) + (rows+1)*4 + cols*B*vb + rows*B*vb) and its flops 2*nnz*B. */
int ss_spmat_cost(const ss_spmat* w, int64_t B, double* bytes, double* flops);

#ifdef __cplusplus
}
#endif
#endif /* SIMSPREAD_HIP_H */
