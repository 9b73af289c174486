// Dense-similarity regime in fp64 (the reference's default precision: predict(...; GPU=false) is Float64,
// src/core.jl:402): stage 1 as a GEMM on the fp64 matrix instruction,
//     T[q][s] = inv_ks[s] * sum_f cut(Sq[q,f]) * inv_kf[f] * cut(Ss[s,f]),
// with featurize's cutoff (src/core.jl:37-43,106-112) applied while the operand tiles are staged into LDS -- the
// thresholded matrices are never written.  v_mfma_f64_16x16x4_f64: exact fp64 products and sums (78.6 TF peak on
// MI355X, the fp64 vector rate).  Same structure as the fp32 kernel of dense.hip: 128 x 128 workgroup tile, 4 waves
// as 2 x 2, a wave owns 4 x 4 MFMA tiles of 16 x 16, K-steps of 16, the raw tile of step k+1 is loaded into registers
// before the MFMA loop of step k and thresholded / scaled / parked in the second LDS buffer after it.  Serves query
// rows, source rows (feature path), leave-one-out (rank-1 degree corrections in the epilogue, own feature dropped
// while staging) and k-fold (member rows gathered through row_ids, this fold's reciprocal degrees).
// Inputs are the raw similarities, column-major (Julia layout): element (row, f) at S[row + f*ld].
#include "graph.hpp"

namespace ss {

#define SS_LAUNCH_CHECK()                                                             \
  do {                                                                                \
    hipError_t _e = hipGetLastError();                                                \
    if (_e != hipSuccess)                                                             \
      return fail(SS_EHIP, "%s:%d kernel launch: %s", __FILE__, __LINE__, hipGetErrorString(_e)); \
  } while (0)

__device__ __forceinline__ double cut64(double x, double alpha, int weighted) {
  return (x >= alpha) ? (weighted ? x : 1.0) : 0.0;
}

struct Dense64Args {
  const double* A;  // query-side similarity, column-major, ld = lda; row m of the product is row arow(m) of A
  int64_t lda;
  const double* B;  // source-side similarity, column-major (N x K)
  int64_t ldb;
  int64_t M, N, K;
  int64_t row_begin;     // product row m <-> A row row_begin + m (or row_ids[row_begin + m])
  const int* row_ids;    // k-fold: member rows
  const double* inv_k;   // [K] 1/kf (LOO: 1/(kf-1); k-fold: this fold's, 0 on the members' own feature columns)
  const double* inv_n;   // [N] 1/ks
  const int* ks;         // LOO: integer source degrees
  double alpha;
  int weighted;
  double* out;           // T, row-major M x N
  int64_t ldo;
  int gx, gy;
};

constexpr int D64_BK = 16;
constexpr int D64_T = 128;        // workgroup tile (rows and columns)
constexpr int D64_LD = D64_T + 2; // 16-byte aligned LDS rows

using f64x4 = __attribute__((ext_vector_type(4))) double;

template <bool LOO>
__global__ void __launch_bounds__(256) transfer_dense_f64_kernel(Dense64Args a) {
  __shared__ __align__(16) double As[2][D64_BK][D64_LD];
  __shared__ __align__(16) double Bs[2][D64_BK][D64_LD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  // tile order: all row blocks of a group of 16 column blocks before the next group (operand tiles shared in L2)
  int bm, bn;
  {
    const int gy = a.gy, GW = 16;
    const int id = (int)blockIdx.x;
    const int grp = id / (gy * GW);
    const int local = id - grp * gy * GW;
    bm = local % gy;
    bn = grp * GW + local / gy;
  }
  if (bn >= a.gx) return;
  const int64_t m0 = (int64_t)bm * D64_T, n0 = (int64_t)bn * D64_T;
  f64x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.0;

  // staging: thread t takes elements (kk, mp .. mp+1) with kk = e / 64, mp = (e % 64) * 2 for e = t, t + 256, ...
  constexpr int IT = D64_BK * D64_T / 2 / 256;  // 4 pairs per thread, operand and K-step
  double ra[IT][2], rb[IT][2], rw[IT];
  // A rows: contiguous (row_begin + m) or gathered (row_ids)
  auto arow = [&](int64_t m) __attribute__((always_inline)) {
    return a.row_ids ? (int64_t)a.row_ids[a.row_begin + m] : a.row_begin + m;
  };
  auto load_raw = [&](int64_t k0) __attribute__((always_inline)) {
#pragma unroll
    for (int it = 0; it < IT; ++it) {
      const int e = tid + it * 256;
      const int kk = e >> 6, mp = (e & 63) * 2;
      const int64_t k = k0 + kk;
      ra[it][0] = ra[it][1] = rb[it][0] = rb[it][1] = 0.0;
      rw[it] = 0.0;
      if (k < a.K) {
        rw[it] = a.inv_k[k];
#pragma unroll
        for (int x = 0; x < 2; ++x) {
          if (m0 + mp + x < a.M) ra[it][x] = a.A[arow(m0 + mp + x) + k * a.lda];
          if (n0 + mp + x < a.N) rb[it][x] = a.B[n0 + mp + x + k * a.ldb];
        }
      }
    }
  };
  auto store_tiles = [&](int64_t k0, int buf) __attribute__((always_inline)) {
#pragma unroll
    for (int it = 0; it < IT; ++it) {
      const int e = tid + it * 256;
      const int kk = e >> 6, mp = (e & 63) * 2;
#pragma unroll
      for (int x = 0; x < 2; ++x) {
        double va = (m0 + mp + x < a.M) ? cut64(ra[it][x], a.alpha, a.weighted) * rw[it] : 0.0;
        if (LOO && (k0 + kk) == arow(m0 + mp + x < a.M ? m0 + mp + x : 0)) va = 0.0;  // the query's own feature is not in the fold
        As[buf][kk][mp + x] = va;
        Bs[buf][kk][mp + x] = (n0 + mp + x < a.N) ? cut64(rb[it][x], a.alpha, a.weighted) : 0.0;
      }
    }
  };

  load_raw(0);
  store_tiles(0, 0);
  __syncthreads();
  if (D64_BK < a.K) load_raw(D64_BK);
  int cur = 0;
  for (int64_t k0 = 0; k0 < a.K; k0 += D64_BK) {
#pragma unroll
    for (int kk = 0; kk < D64_BK; kk += 4) {
      // A operand of v_mfma_f64_16x16x4_f64: lane l holds A[l % 16][l / 16]; B operand: B[l / 16][l % 16]
      const int kr = kk + (lane >> 4);
      const int c = lane & 15;
      double av[4], bv[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) av[i] = As[cur][kr][wm * 64 + i * 16 + c];
#pragma unroll
      for (int j = 0; j < 4; ++j) bv[j] = Bs[cur][kr][wn * 64 + j * 16 + c];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[i], bv[j], acc[i][j], 0, 0, 0);
    }
    if (k0 + D64_BK < a.K) store_tiles(k0 + D64_BK, cur ^ 1);
    __syncthreads();
    if (k0 + 2 * D64_BK < a.K) load_raw(k0 + 2 * D64_BK);  // in flight during the next MFMA loop
    cur ^= 1;
  }
  // C/D layout of v_mfma_f64_16x16x4_f64 (probed on MI355X with one-hot operands): col = lane % 16,
  // row = 4 * reg + lane / 16  (unlike the f32 16x16x4 form, whose row is 4 * (lane / 16) + reg)
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int64_t n = n0 + wn * 64 + j * 16 + (lane & 15);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int64_t m = m0 + wm * 64 + i * 16 + 4 * r + (lane >> 4);
        if (m < a.M && n < a.N) {
          double z;
          if (LOO) {
            const int64_t q = arow(m);  // query = source q; S is the square source similarity
            const int has = cut64(a.B[n + q * a.ldb], a.alpha, a.weighted) != 0.0 ? 1 : 0;  // X[s][f_q]
            const int d = a.ks[n] - has;
            z = (d > 0 && n != q) ? acc[i][j][r] * (1.0 / (double)d) : 0.0;
          } else {
            z = acc[i][j][r] * a.inv_n[n];
          }
          a.out[m * a.ldo + n] = z;
        }
      }
    }
}

int launch_transfer_dense_f64(const DenseSim<double>& d, bool loo, const double* inv_k, const double* inv_n, const int* ks,
                              int64_t row_begin, int64_t nrows, double* out, int64_t ldo, bool source_rows,
                              const int* row_ids) {
  if (nrows <= 0 || d.ns <= 0) return SS_OK;
  path_add("transfer_dense_f64_mfma");
  Dense64Args a{};
  const bool from_ss = loo || source_rows || row_ids != nullptr;  // rows of the source similarity itself
  a.A = from_ss ? d.Ss.p : d.Sq.p;
  a.lda = from_ss ? d.ns : d.nq;
  a.B = d.Ss.p;
  a.ldb = d.ns;
  a.M = nrows;
  a.N = d.ns;
  a.K = d.nf;
  a.row_begin = row_begin;
  a.row_ids = row_ids;
  a.inv_k = inv_k;
  a.inv_n = inv_n;
  a.ks = ks;
  a.alpha = d.alpha;
  a.weighted = d.weighted ? 1 : 0;
  a.out = out;
  a.ldo = ldo;
  a.gx = (int)ceil_div(d.ns, D64_T);
  a.gy = (int)ceil_div(nrows, D64_T);
  dim3 grid((unsigned)(a.gx * a.gy));
  if (loo) hipLaunchKernelGGL(transfer_dense_f64_kernel<true>, grid, dim3(256), 0, ctx().stream, a);
  else hipLaunchKernelGGL(transfer_dense_f64_kernel<false>, grid, dim3(256), 0, ctx().stream, a);
  SS_LAUNCH_CHECK();
  return SS_OK;
}

}  // namespace ss
