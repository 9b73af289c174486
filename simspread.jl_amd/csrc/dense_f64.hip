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
constexpr int D64_LD = D64_T + 2; // 16-byte aligned LDS rows (1040 bytes)

using f64x4 = __attribute__((ext_vector_type(4))) double;

// WEIGHTED: featurize(..., weighted): the value itself or 1 above the cutoff; GATHER: A rows through row_ids (k-fold).
// Staging is written for few vector instructions per element -- the first version spent 7.7 VALU instructions per MFMA
// on bounds checks, 64-bit address arithmetic and selects, and the matrix pipe was busy 46 % of the time
// (profiles/r02_c4_f64_pmc.txt): everything that does not change along K is computed once per thread (its two A rows
// and two B rows, their validity folded into the cutoff they are compared with: +inf for a row outside the matrix),
// loads are unconditional from clamped addresses, and the two adjacent rows of a lane move as one 16-byte access.
template <bool LOO, bool WEIGHTED, bool GATHER>
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

  // staging: thread t owns tile rows mp, mp + 1 (mp = 2 * lane) of both operands and the k values wave + 4 * it of a step
  constexpr int IT = D64_BK / 4;
  const int mp = lane * 2;
  auto arow = [&](int64_t m) __attribute__((always_inline)) {
    return GATHER ? (int64_t)a.row_ids[a.row_begin + m] : a.row_begin + m;
  };
  const double inf = __builtin_huge_val();
  int64_t rowa[2];        // A row (clamped) of tile rows mp, mp + 1
  double cuta[2], cutb[2];  // the cutoff they are compared with: +inf outside the matrix -> staged as 0
  const double* pb;       // B rows are contiguous: n0 + mp (+1)
  {
#pragma unroll
    for (int x = 0; x < 2; ++x) {
      const int64_t m = m0 + mp + x;
      rowa[x] = arow(m < a.M ? m : a.M - 1);
      cuta[x] = m < a.M ? a.alpha : inf;
      cutb[x] = (n0 + mp + x < a.N) ? a.alpha : inf;
    }
  }
  // two adjacent rows in one 16-byte access when they are adjacent in memory and inside the matrix (else two 8-byte ones)
  // (both conditions are the same for every lane of the workgroup: scalar branches)
  const bool pair_a = !GATHER && (m0 + D64_T <= a.M) && (((a.row_begin + m0) & 1) == 0) && ((a.lda & 1) == 0) &&
                      ((reinterpret_cast<uintptr_t>(a.A) & 15) == 0);
  const bool pair_b = (n0 + D64_T <= a.N) && ((a.ldb & 1) == 0) && ((reinterpret_cast<uintptr_t>(a.B) & 15) == 0);
  const int64_t nb0 = (n0 + mp < a.N) ? n0 + mp : a.N - 1, nb1 = (n0 + mp + 1 < a.N) ? n0 + mp + 1 : a.N - 1;
  pb = a.B;
  using d2 = __attribute__((ext_vector_type(2))) double;
  double ra[IT][2], rb[IT][2], rw[IT];
  auto load_raw = [&](int64_t k0) __attribute__((always_inline)) {
#pragma unroll
    for (int it = 0; it < IT; ++it) {
      const int64_t k = k0 + wave + 4 * it;            // wave-uniform
      const int64_t kc = k < a.K ? k : a.K - 1;        // clamped: a step past K reads valid data and weighs it with 0
      rw[it] = k < a.K ? a.inv_k[kc] : 0.0;
      if (pair_a) {
        const d2 v = *reinterpret_cast<const d2*>(a.A + rowa[0] + kc * a.lda);
        ra[it][0] = v[0]; ra[it][1] = v[1];
      } else {
        ra[it][0] = a.A[rowa[0] + kc * a.lda];
        ra[it][1] = a.A[rowa[1] + kc * a.lda];
      }
      if (pair_b) {
        const d2 v = *reinterpret_cast<const d2*>(pb + nb0 + kc * a.ldb);
        rb[it][0] = v[0]; rb[it][1] = v[1];
      } else {
        rb[it][0] = pb[nb0 + kc * a.ldb];
        rb[it][1] = pb[nb1 + kc * a.ldb];
      }
    }
  };
  // one quarter of the staging work of a K-step (the k value wave + 4 * it of both operands)
  auto store_piece = [&](int64_t k0, int buf, int it) __attribute__((always_inline)) {
    const int kk = wave + 4 * it;
    d2 va, vb;
#pragma unroll
    for (int x = 0; x < 2; ++x) {
      bool on = ra[it][x] >= cuta[x];
      if (LOO) on = on && (k0 + kk != rowa[x]);   // the query's own feature is not in the fold
      va[x] = on ? (WEIGHTED ? ra[it][x] * rw[it] : rw[it]) : 0.0;
      vb[x] = (rb[it][x] >= cutb[x]) ? (WEIGHTED ? rb[it][x] : 1.0) : 0.0;
    }
    *reinterpret_cast<d2*>(&As[buf][kk][mp]) = va;
    *reinterpret_cast<d2*>(&Bs[buf][kk][mp]) = vb;
  };

  load_raw(0);
#pragma unroll
  for (int it = 0; it < IT; ++it) store_piece(0, 0, it);
  __syncthreads();
  load_raw(D64_BK);   // (clamped past K: valid addresses, weight 0)
  int cur = 0;
  for (int64_t k0 = 0; k0 < a.K; k0 += D64_BK) {
    // The tile of the next K-step is thresholded and parked in the other LDS buffer piece by piece BETWEEN the four MFMA
    // groups of this step (same basic block, no branches: a step past K stages zero weights into a buffer nobody reads),
    // so the vector and LDS-store instructions issue while the matrix pipe works instead of after it.
#pragma unroll
    for (int st = 0; st < D64_BK / 4; ++st) {
      const int kk = 4 * st;
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
      store_piece(k0 + D64_BK, cur ^ 1, st);
    }
    __syncthreads();
    load_raw(k0 + 2 * D64_BK);  // in flight during the next MFMA loop
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
  const bool gather = row_ids != nullptr;
#define SS_D64(L, W, G) hipLaunchKernelGGL((transfer_dense_f64_kernel<L, W, G>), grid, dim3(256), 0, ctx().stream, a)
  if (loo) {
    if (d.weighted) SS_D64(true, true, false); else SS_D64(true, false, false);
  } else if (gather) {
    if (d.weighted) SS_D64(false, true, true); else SS_D64(false, false, true);
  } else {
    if (d.weighted) SS_D64(false, true, false); else SS_D64(false, false, false);
  }
#undef SS_D64
  SS_LAUNCH_CHECK();
  return SS_OK;
}

}  // namespace ss
