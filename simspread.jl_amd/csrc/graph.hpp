// Device-resident sparse operands of the resource-spreading pass.
//
//   DevCsr   plain CSR, 4-byte row pointers and column indices (SURVEY.md 8d byte model)
//   DevSell  the same matrix re-cut for the wide W*R kernel: rows in slices of 64 (one
//            lane per row), columns in chunks of KC (one LDS tile of R per chunk),
//            chunk-local 16-bit column indices, four entries per lane per load
//   Graph    the blocks Xq, Xs, Ys of construct's adjacency (src/core.jl:165-187), their
//            transposes, and the count degrees of B (src/core.jl:365-371)
#pragma once
#include "common.hpp"

namespace ss {

template <class T>
struct DevCsr {
  int64_t rows = 0, cols = 0, nnz = 0;
  DevBuf<int> ptr;   // rows+1
  DevBuf<int> idx;   // nnz, sorted within a row
  DevBuf<T> val;     // nnz
  bool binary = false;  // every stored value == 1
};

template <class T>
struct DevSell {
  int64_t rows = 0, cols = 0;
  int KC = 0;          // columns per chunk; local indices KC .. KC+15 are zero rows (padding targets)
  int qt = 0;          // columns per tile row the entry order was scheduled for (slot classes = 256 / (qt * sizeof(T)))
  int nchunks = 0;
  int nslices = 0;     // ceil(rows / 64)
  int64_t nquads = 0;  // total storage in units of 64 lanes x 4 entries
  bool binary = false;
  DevBuf<int> off;              // [nchunks*nslices + 1] quad offsets, chunk-major
  DevBuf<unsigned short> idx;   // [nquads][64][4]
  DevBuf<T> val;                // [nquads][64][4] unless binary
  // skewed row lengths (power-law graphs): rows are sorted by length before they are cut into slices so
  // that the 64 rows of a slice are equally long; results then come out in sorted order and are put back
  // by unpermute_kernel.  perm[sorted position] = row, inv[row] = sorted position.
  // Rows longer than a wave's fair share are first split into "virtual rows" of bounded length (their
  // partial scores are summed, in order, by the same epilogue), so a hot target cannot serialise a slice.
  bool sorted = false;
  int64_t vrows = 0;         // virtual rows (= slices * 64 rounded down); rows when not sorted
  DevBuf<int> vs, ve;        // [vrows] CSR entry range of the virtual row at each sorted position
  DevBuf<int> vfirst;        // [rows + 1] virtual ids of each real row
  DevBuf<int> inv;           // [vrows] sorted position of each virtual id
};

// CSR cut into column chunks of SC columns, stored chunk-major with chunk-local 16-bit indices:
// sub-row (c, r) = entries off[c*rows + r] .. off[c*rows + r + 1].  Stage 1 gives chunk c of every
// transfer row to workgroups with blockIdx % nchunks == c, so one XCD's L2 only ever sees the
// sub-rows of "its" chunks (nchunks is a multiple of 8).
template <class T>
struct DevChunked {
  int64_t rows = 0, cols = 0, nnz = 0;
  int64_t stored = 0;          // entries incl. padding
  bool binary = false;         // every stored value == 1 (unweighted features): kernels may skip the value stream
  int SC = 0, nchunks = 0;
  int align = 1;               // sub-rows padded to whole units of `align` entries; off[] counts units.
                               // Padding entries carry local index SC (a zero operand row) and value 0.
  DevBuf<int> off;             // [nchunks*rows + 1]
  DevBuf<unsigned short> idx;  // [stored + 64]
  DevBuf<T> val;               // [stored + 64]
  DevBuf<unsigned short> len;  // align 32 only: exact entry count of every sub-row [nchunks*rows]
  bool len_ok = false;         //   ... all of them < 65536
  float vmax = 0.f, vmin = 0.f;  // largest / smallest non-zero |value|
};

// Mid-width W*R operand of round 3 (spmm_csell.hip): "compact sliced ELL".  Rows in slices of 64 (one lane per row),
// columns in chunks of KC (one LDS tile of R); block (chunk c, slice s) = c*nslices + s stores its entries pair by pair:
// step u holds entries 2u, 2u+1 of every lane whose sub-row has them, lanes in ascending order, nothing for the others,
// so a wave step is one coalesced 4-byte (two 16-bit local indices) and one 8-byte (two values) load per lane and memory
// holds no padding except the odd last entry of a sub-row (index KC = the zero row of the tile, value 0).  The order of
// the entries inside a sub-row is chosen when the operand is built so that lanes which read the same 16-byte slot of an
// LDS line in the same cycle hold tile rows of different bank classes (tile rows narrower than 256 bytes).
template <class T>
struct DevCsell {
  int64_t rows = 0, cols = 0, nnz = 0;
  int KC = 0, nchunks = 0, nslices = 0;
  int QT = 0;                     // tile width (columns of R per tile row) the entry order was scheduled for: QT * sizeof(T) = 64, 128 or 256 bytes
  bool binary = false;            // every value 1: no value stream
  bool ok = false;                // built (false: not yet, or the matrix does not fit the format -> the 2-D kernel serves it)
  int64_t npairs = 0;
  DevBuf<int> desc;               // [nblocks + 1][2] {first pair, steps}; the last block is empty
  DevBuf<unsigned short> np;      // [(nblocks + 1) * 64] pairs per lane
  DevBuf<unsigned> pidx;          // [npairs + 64]
  DevBuf<T> pval;                 // [2 * (npairs + 64)] unless binary
};

// Dense-similarity regime: the raw similarities stay dense on the device (column-major), the cutoff is
// applied inside the stage-1 GEMM (dense.hip).
template <class T>
struct DenseSim {
  bool on = false;
  int64_t nq = 0, ns = 0, nf = 0;
  T alpha = T(0);
  bool weighted = true;
  DevBuf<T> Sq, Ss;       // nq x nf and ns x nf, column-major, ld = rows
  DevBuf<T> inv_kf_m1;    // 1/(kf-1): leave-one-out coefficient
  // bf16 planes of the thresholded operands (dense_bf16.hip): source side once per graph, query side per row block
  DevBuf<unsigned short> Bpl, Apl;
  int Bpl_np = 0;
  int64_t Bpl_Np = 0, Bpl_Kp = 0;
};

template <class T>
struct Graph {
  int64_t nq = 0, ns = 0, nf = 0, nt = 0;
  bool general = false;
  DenseSim<T> dense;  // built by ss_graph_create_general: only Xq, XsT, YsT and kf == ks are set
  DevCsr<T> Xq;    // nq x nf
  DevCsr<T> Xs;    // ns x nf
  DevCsr<T> XsT;   // nf x ns
  DevCsr<T> Ys;    // ns x nt
  DevCsr<T> YsT;   // nt x ns   (the W of F = W*R)
  DevBuf<int> kf, ks, kt;           // count degrees in B
  DevBuf<T> inv_kf, inv_ks, inv_kt; // 1/k, 0 where k == 0   (the Inf/NaN -> 0 rule)
  // stage-2 operand, built lazily per tile width
  DevSell<T> W;
  int W_qt = 0;
  DevChunked<T> XsTc, YsTc;  // stage-1 operands (YsTc only for source rows), built lazily
  DevChunked<T> XsTb;        // stage-1 operand of the query-block kernel (sub-rows padded to L2 sectors), built lazily
  DevBuf<T> Cq;              // workspace of that kernel: Xq[r,a] * inv_kf[a], entry by entry
  DevBuf<T> Tws;  // workspace: rows of the transfer block T between stage 1 and stage 2
  DevBuf<T> Sws;  // workspace: sorted-order scores of a skew-sorted stage-2 operand
};

template <class T>
struct SpMat {
  DevCsr<T> csr;
  DevSell<T> sell;
  int sell_qt = 0;
  DevChunked<T> narrow[3];  // quad-aligned chunked operands of the narrow kernel for B <= 1, 2, 4 (built lazily)
  DevChunked<T> col[3];     // operands of the 2-D kernel: 64-, 128- and 256-byte tile rows (built lazily)
  DevCsell<T> csell[6];     // operands of the lane-per-row kernel for 64-, 128-, 256- and (fp32, B <= 8) 32-byte tile rows (built lazily)
  bool csell_tried[6] = {false, false, false, false, false, false};
  DevBuf<T> partial;        // partial sums of the narrow / 2-D kernels (and the padded copy of R, spmm_colgroup.hip)
};

// ---- assemble.hip
template <class T>
int csr_from_user(int64_t rows, int64_t cols, const int64_t* ptr, const int32_t* idx, const T* val,
                  int index_base, int mem, DevCsr<T>& out);
template <class T>
int csr_from_dense(const T* S, int64_t rows, int64_t cols, int64_t ld, bool apply_cutoff, T alpha,
                   bool weighted, int mem, DevCsr<T>& out);
template <class T>
int csr_transpose(const DevCsr<T>& in, DevCsr<T>& out);
template <class T>
int sell_build(const DevCsr<T>& in, int KCmax, DevSell<T>& out, int qt = 0);   // qt = 0: sell_tile_width<T>()
template <class T>
int chunked_build(const DevCsr<T>& in, int SC, int align, DevChunked<T>& out);
template <class T>
int csell_build(const DevCsr<T>& in, int KC, int QT, DevCsell<T>& out);   // out.ok == false: not representable (no error)
template <class T>
int graph_finalize(Graph<T>& g);  // transposes + degrees
template <class T>
int graph_finalize_general(Graph<T>& g);  // degrees = row counts of B (held in XsT)
template <class T>
int graph_finalize_general_targets(Graph<T>& g);  // kt / inv_kt from YsT only

// ---- kernels.hip (launch wrappers; everything is enqueued on ctx().stream)
template <class T>
struct CsrView {
  const int* ptr;
  const int* idx;
  const T* val;
};
template <class T>
inline CsrView<T> view(const DevCsr<T>& m) { return CsrView<T>{m.ptr.p, m.idx.p, m.val.p}; }

template <class T>
int launch_cutoff(const T* X, int64_t rows, int64_t cols, int64_t ld, T alpha, bool weighted, T* out, int64_t ldo);
template <class T>
int launch_row_degree(const T* G, int64_t rows, int64_t cols, int64_t ld, int* deg);
template <class T>
int launch_spread_dense(const T* G, int64_t rows, int64_t cols, int64_t ld, const int* deg, T* W, int64_t ldw);

// stage 1: T[r][j] = inv2[j] * sum_terms sum_a L[r,a] * inv1[a] * Mt[a][j]   (rows row_begin..+nrows)
// all Mt operands must share SC / nchunks
template <class T>
int launch_transfer(int nterms, const DevCsr<T>* L[2], const T* inv1[2], const DevChunked<T>* Mt[2],
                    const T* inv2, int64_t row_begin, int64_t nrows, int64_t nj, T* out, int64_t ld,
                    const int* row_ids = nullptr, bool accumulate = false);
// stage 1 for plain query rows, query-block workgroups: sub-row offsets in LDS, coefficients precomputed into `coef`
// (L.nnz values of workspace).  transfer_block_fits: whether offsets + >= 8 waves of accumulators fit in LDS.
template <class T>
int launch_transfer_block(const DevCsr<T>& L, const T* inv1, const DevChunked<T>& Mt, const T* inv2, int64_t row_begin,
                          int64_t nrows, int64_t nj, T* out, int64_t ld, T* coef, float xmax, bool fixed);
template <class T>
bool transfer_block_fits(int64_t mrows, int SC);
// k-fold: degrees / reciprocal degrees of the graph without the members of one fold
template <class T>
int launch_fold_degrees(const DevCsr<T>& X, const DevCsr<T>& XT, const DevCsr<T>& Y, const int* members,
                        int64_t nmembers, int* kf, int* ks, int* kt);
template <class T>
int launch_fold_inverse(const int* kf, const int* ks, const int* fold, int phi, int64_t nf, int64_t ns, T* inv_kf,
                        T* inv_ks);
// leave-one-out flavour: row i of X against X' with the rank-1 degree corrections
template <class T>
int launch_transfer_loo(const DevCsr<T>& X, const DevChunked<T>& XT, const int* kf, const int* ks,
                        int64_t i_begin, int64_t nrows, T* out, int64_t ld);

// stage 2, wide: F[b][m] = sum_k W[m][k] * R[b][k]   (R, F "column-major": one row of length K / M per b)
template <class T>
int sell_tile_width();  // QT for this T
template <class T>
int sell_max_chunk(int qt);
template <class T>
int launch_spmm_sell(const DevSell<T>& W, const T* R, int64_t ldr, int64_t B, T* F, int64_t ldf,
                     const int* clean_deg, const int* out_rows = nullptr);
// stage 2, narrow (serves B <= 7 by default, built for B <= 16): R chunk in LDS, W streamed once in chunk-major order
template <class T>
int narrow_chunk_cols(int bv);  // KC for a padded width bv
template <class T>
int launch_spmm_chunked_narrow(const DevChunked<T>& W, int bv, const T* R, int64_t ldr, int B, T* F, int64_t ldf,
                               DevBuf<T>& partial);
// stage 2, mid width (5 <= B, B*sizeof(T) <= 256 bytes): 2-D cut (row blocks x chunk groups) with conflict-free gathers
// (spmm_colgroup.hip): tile rows of 64, 128 or 256 bytes (fp32: B <= 16, 32, 64; fp64: B <= 8, 16, 32); same chunked
// operand format as the narrow kernel
// ---- spmm_csell.hip (5 <= B, B * sizeof(T) <= 256 bytes, row-major R and F)
int csell_chunk_cols(int rowb);   // tile rows (of rowb bytes) that fit in LDS next to the zero row
template <class T>
int launch_spmm_csell(const DevCsell<T>& W, const T* R, int64_t ldr, int B, T* F, int64_t ldf, DevBuf<T>& partial);
template <class T>
int colgroup_chunk_cols(int bv);
template <class T>
int launch_spmm_colgroup(const DevChunked<T>& W, int bv, const T* R, int64_t ldr, int B, T* F, int64_t ldf,
                         DevBuf<T>& partial);

// similarity producer of the reference's tutorial: weighted Jaccard between the rows of a feature matrix
template <class T>
int launch_jaccard(const T* F, int64_t n, int64_t d, int64_t ld, T* S, int64_t lds);

int launch_topl(const float* scores, int64_t nrows, int64_t ncols, int64_t ld, int L, int* oidx, float* oval);
// metrics.hip: AuROC, AuPRC, BEDROC(alpha), validity ratio of one score vector (device inputs, host outputs)
int launch_rank_metrics(const unsigned char* y, const float* yhat, int64_t n, double alpha, double* out4);

// ---- dense.hip (fp32 only: fp32-input MFMA)
int launch_transfer_dense(const DenseSim<float>& d, bool loo, const float* inv_k, const float* inv_n, const int* ks,
                          int64_t row_begin, int64_t nrows, float* out, int64_t ldo, bool source_rows = false);
template <class T>
int dense_degrees(Graph<T>& g);
// dense_bf16.hip: the same product on the bf16 matrix cores with the operands split into exact bf16 planes
int launch_transfer_dense_bf16(DenseSim<float>& d, bool loo, const float* inv_k, const float* inv_n, const int* ks,
                               int64_t row_begin, int64_t nrows, float* out, int64_t ldo, bool source_rows = false,
                               const int* row_ids = nullptr);
template <class T>
int dense_fold_degrees(const Graph<T>& g, const int* members, int64_t nm, int* kf, int* ks, int* kt);
// dense_f64.hip: the dense-similarity stage 1 in fp64 (fp64 MFMA; the reference's default precision)
int launch_transfer_dense_f64(const DenseSim<double>& d, bool loo, const double* inv_k, const double* inv_n, const int* ks,
                              int64_t row_begin, int64_t nrows, double* out, int64_t ldo, bool source_rows = false,
                              const int* row_ids = nullptr);

template <class T>
int launch_transpose(const T* in, int64_t rows, int64_t cols, int64_t ldin, T* out, int64_t ldout);
// out[r][t] = clean(sum_{v in vfirst[t]..vfirst[t+1]} in[r][inv[v]]): undo the row split + sort of a
// skew-sorted SELL operand (+ fused clean!)
template <class T>
int launch_unpermute(const T* in, int64_t ldin, int64_t nrows, int64_t nt, const int* vfirst, const int* inv,
                     const int* clean_deg, T* out, int64_t ldout);
// clean! for leave-one-out rows: target t whose only edge belongs to query i
template <class T>
int launch_loo_clean_fix(const DevCsr<T>& YsT, const int* kt, int64_t i_begin, int64_t nrows, T* out, int64_t ld);

// ---- comm.hip: in-library score gather over RCCL (dlopen'ed), one process per GPU
int comm_unique_id(char* id128);
int comm_init(const char* id128, int rank, int nranks);
int comm_destroy();
int comm_info(int* rank, int* nranks);
int gather_rows(const void* local, int64_t ncols, const int64_t* counts, void* full, int root, int elem);

}  // namespace ss
