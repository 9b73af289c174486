// Dense-similarity regime (BASELINE config 4): when the thresholded similarity is too full for CSR
// (90 % fill of a 50k x 50k matrix is 2.25e9 non-zeros) stage 1 runs as a GEMM on the matrix cores,
//     T[q][s] = inv_ks[s] * sum_f cut(Sq[q,f]) * inv_kf[f] * cut(Ss[s,f]),
// with featurize's cutoff (src/core.jl:37-43,106-112) applied while the operand tiles are staged into
// LDS -- the thresholded matrices are never written.  fp32-input MFMA (v_mfma_f32_32x32x2_f32): exact
// fp32 products and sums, the only matrix instruction that keeps weighted features at full precision.
// Inputs are the raw similarities, column-major (Julia layout): element (row, f) at S[row + f*ld], which
// is exactly the [k][m] order the LDS tiles want.  Stage 2 (W*R over Y) is the sparse SELL kernel.
#include "graph.hpp"

namespace ss {

#define SS_LAUNCH_CHECK()                                                             \
  do {                                                                                \
    hipError_t _e = hipGetLastError();                                                \
    if (_e != hipSuccess)                                                             \
      return fail(SS_EHIP, "%s:%d kernel launch: %s", __FILE__, __LINE__, hipGetErrorString(_e)); \
  } while (0)

__device__ __forceinline__ float cut_val(float x, float alpha, int weighted) {
  // cutoff(x, alpha, weighted); a kept weight of 0 is no edge
  return (x >= alpha) ? (weighted ? x : 1.0f) : 0.0f;
}

// ------------------------------------------------------------------ degrees of the thresholded similarity
// kf[f] = #rows with cut(S[row,f]) != 0: one wave per column (contiguous in column-major)
__global__ void dense_col_degree_kernel(const float* __restrict__ S, int64_t rows, int64_t cols, int64_t ld,
                                        float alpha, int weighted, int* __restrict__ deg) {
  const int lane = threadIdx.x & 63;
  const int64_t wave0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t f = wave0; f < cols; f += nwaves) {
    int n = 0;
    for (int64_t r = lane; r < rows; r += 64) n += cut_val(S[r + f * ld], alpha, weighted) != 0.0f ? 1 : 0;
    for (int o = 32; o > 0; o >>= 1) n += __shfl_xor(n, o);
    if (lane == 0) deg[f] = n;
  }
}

// row counts: thread per row, lanes walk a column together; column splits add with integer atomics
__global__ void dense_row_degree_kernel(const float* __restrict__ S, int64_t rows, int64_t cols, int64_t ld,
                                        float alpha, int weighted, int64_t cols_per_split, int* __restrict__ deg) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= rows) return;
  const int64_t c0 = (int64_t)blockIdx.y * cols_per_split;
  const int64_t c1 = (c0 + cols_per_split < cols) ? c0 + cols_per_split : cols;
  int n = 0;
  for (int64_t c = c0; c < c1; ++c) n += cut_val(S[r + c * ld], alpha, weighted) != 0.0f ? 1 : 0;
  atomicAdd(&deg[r], n);
}

template <class T>
__global__ void dense_finish_degrees_kernel(const int* __restrict__ kx, const int* __restrict__ yptr, int64_t n,
                                            int* __restrict__ k, T* __restrict__ inv, T* __restrict__ inv_m1) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    int d = kx[i];
    if (yptr) d += yptr[i + 1] - yptr[i];
    k[i] = d;
    inv[i] = d > 0 ? T(1) / T(d) : T(0);
    if (inv_m1) inv_m1[i] = d > 1 ? T(1) / T(d - 1) : T(0);  // leave-one-out: the query left this column
  }
}

// ------------------------------------------------------------------ the fused cutoff + GEMM
struct DenseArgs {
  const float* A;  // query-side similarity, column-major (M x K), ld = lda
  int64_t lda;
  const float* B;  // source-side similarity, column-major (N x K)
  int64_t ldb;
  int64_t M, N, K;
  int64_t row_begin;   // global index of A's row 0 (LOO: query i = row_begin + m)
  const float* inv_k;  // [K] 1/kf (LOO: 1/(kf-1))
  const float* inv_n;  // [N] 1/ks
  const int* ks;       // LOO: integer source degrees
  float alpha;
  int weighted;
  float* out;          // T, row-major M x N
  int64_t ldo;
};

constexpr int DBM = 128, DBN = 128, DBK = 32, DLD = 132;  // DLD: 16-byte aligned rows for ds_write_b128

using f32x16 = __attribute__((ext_vector_type(16))) float;

// block 256 threads = 4 waves as 2 x 2, each wave 64 x 64 = 2 x 2 MFMA tiles of 32 x 32
template <bool LOO>
__global__ void __launch_bounds__(256) transfer_dense_kernel(DenseArgs a) {
  __shared__ __align__(16) float As[DBK][DLD];
  __shared__ __align__(16) float Bs[DBK][DLD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int64_t m0 = (int64_t)blockIdx.y * DBM, n0 = (int64_t)blockIdx.x * DBN;
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  // staging: 128 x 32 tile = 1024 float4 along m; thread t takes quads t, t+256, t+512, t+768
  const bool a_vec = (a.lda % 4 == 0) && ((reinterpret_cast<uintptr_t>(a.A) & 15) == 0);
  const bool b_vec = (a.ldb % 4 == 0) && ((reinterpret_cast<uintptr_t>(a.B) & 15) == 0);
  for (int64_t k0 = 0; k0 < a.K; k0 += DBK) {
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int qd = tid + it * 256;
      const int kk = qd >> 5;          // 0..31
      const int mq = (qd & 31) * 4;    // 0..124
      const int64_t k = k0 + kk;
      float4 va = make_float4(0.f, 0.f, 0.f, 0.f), vb = make_float4(0.f, 0.f, 0.f, 0.f);
      if (k < a.K) {
        const float* pa = a.A + k * a.lda + m0 + mq;
        const float* pb = a.B + k * a.ldb + n0 + mq;
        if (a_vec && m0 + mq + 3 < a.M) va = *reinterpret_cast<const float4*>(pa);
        else {
          if (m0 + mq + 0 < a.M) va.x = pa[0];
          if (m0 + mq + 1 < a.M) va.y = pa[1];
          if (m0 + mq + 2 < a.M) va.z = pa[2];
          if (m0 + mq + 3 < a.M) va.w = pa[3];
        }
        if (b_vec && n0 + mq + 3 < a.N) vb = *reinterpret_cast<const float4*>(pb);
        else {
          if (n0 + mq + 0 < a.N) vb.x = pb[0];
          if (n0 + mq + 1 < a.N) vb.y = pb[1];
          if (n0 + mq + 2 < a.N) vb.z = pb[2];
          if (n0 + mq + 3 < a.N) vb.w = pb[3];
        }
        const float w = a.inv_k[k];
        // out-of-range rows were loaded as 0: with alpha <= 0 they must still stay zero
        va.x = (m0 + mq + 0 < a.M) ? cut_val(va.x, a.alpha, a.weighted) * w : 0.f;
        va.y = (m0 + mq + 1 < a.M) ? cut_val(va.y, a.alpha, a.weighted) * w : 0.f;
        va.z = (m0 + mq + 2 < a.M) ? cut_val(va.z, a.alpha, a.weighted) * w : 0.f;
        va.w = (m0 + mq + 3 < a.M) ? cut_val(va.w, a.alpha, a.weighted) * w : 0.f;
        vb.x = (n0 + mq + 0 < a.N) ? cut_val(vb.x, a.alpha, a.weighted) : 0.f;
        vb.y = (n0 + mq + 1 < a.N) ? cut_val(vb.y, a.alpha, a.weighted) : 0.f;
        vb.z = (n0 + mq + 2 < a.N) ? cut_val(vb.z, a.alpha, a.weighted) : 0.f;
        vb.w = (n0 + mq + 3 < a.N) ? cut_val(vb.w, a.alpha, a.weighted) : 0.f;
        if (LOO) {  // the feature named after the query itself is not in the fold's graph
          const int64_t d = k - (a.row_begin + m0 + mq);
          if (d == 0) va.x = 0.f;
          if (d == 1) va.y = 0.f;
          if (d == 2) va.z = 0.f;
          if (d == 3) va.w = 0.f;
        }
      }
      *reinterpret_cast<float4*>(&As[kk][mq]) = va;
      *reinterpret_cast<float4*>(&Bs[kk][mq]) = vb;
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < DBK; kk += 2) {
      const int kr = kk + (lane >> 5);
      const int c = lane & 31;
      const float a0 = As[kr][wm * 64 + c], a1 = As[kr][wm * 64 + 32 + c];
      const float b0 = Bs[kr][wn * 64 + c], b1 = Bs[kr][wn * 64 + 32 + c];
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
    }
    __syncthreads();
  }
  // C/D layout of 32x32 f32 MFMA: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int64_t n = n0 + wn * 64 + j * 32 + (lane & 31);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t m = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (m < a.M && n < a.N) {
          float z;
          if (LOO) {
            const int64_t q = a.row_begin + m;  // query = source q; S is the square source similarity
            const int has = cut_val(a.B[n + q * a.ldb], a.alpha, a.weighted) != 0.0f ? 1 : 0;  // X[s][f_q]
            const int d = a.ks[n] - has;
            z = (d > 0 && n != q) ? acc[i][j][r] * (1.0f / (float)d) : 0.0f;
          } else {
            z = acc[i][j][r] * a.inv_n[n];
          }
          a.out[m * a.ldo + n] = z;
        }
      }
    }
}

int launch_transfer_dense(const DenseSim<float>& d, bool loo, const float* inv_k, const float* inv_n, const int* ks,
                          int64_t row_begin, int64_t nrows, float* out, int64_t ldo) {
  if (nrows <= 0 || d.ns <= 0) return SS_OK;
  DenseArgs a{};
  if (loo) { a.A = d.Ss.p + row_begin; a.lda = d.ns; }
  else { a.A = d.Sq.p + row_begin; a.lda = d.nq; }
  a.B = d.Ss.p;
  a.ldb = d.ns;
  a.M = nrows;
  a.N = d.ns;
  a.K = d.nf;
  a.row_begin = row_begin;
  a.inv_k = inv_k;
  a.inv_n = inv_n;
  a.ks = ks;
  a.alpha = d.alpha;
  a.weighted = d.weighted ? 1 : 0;
  a.out = out;
  a.ldo = ldo;
  dim3 grid((unsigned)ceil_div(d.ns, DBN), (unsigned)ceil_div(nrows, DBM));
  if (loo) hipLaunchKernelGGL(transfer_dense_kernel<true>, grid, dim3(256), 0, ctx().stream, a);
  else hipLaunchKernelGGL(transfer_dense_kernel<false>, grid, dim3(256), 0, ctx().stream, a);
  SS_LAUNCH_CHECK();
  return SS_OK;
}

// degrees of the thresholded similarity + labels: kf (columns of cut(Ss)), ks (rows of cut(Ss) + rows of Y)
int dense_degrees(Graph<float>& g) {
  hipStream_t st = ctx().stream;
  DenseSim<float>& d = g.dense;
  DevBuf<int> kx;
  SS_TRY(kx.alloc(d.ns));
  SS_TRY(g.kf.alloc(d.nf));
  SS_TRY(g.inv_kf.alloc(d.nf));
  SS_TRY(d.inv_kf_m1.alloc(d.nf));
  SS_TRY(g.ks.alloc(d.ns));
  SS_TRY(g.inv_ks.alloc(d.ns));
  if (d.ns > 0 && d.nf > 0) {
    DevBuf<int> kcol;
    SS_TRY(kcol.alloc(d.nf));
    hipLaunchKernelGGL(dense_col_degree_kernel, dim3((unsigned)(ceil_div(d.nf * 64, 256) < 4096 ? ceil_div(d.nf * 64, 256) : 4096)),
                       dim3(256), 0, st, d.Ss.p, d.ns, d.nf, d.ns, d.alpha, d.weighted ? 1 : 0, kcol.p);
    SS_LAUNCH_CHECK();
    SS_HIP(hipMemsetAsync(kx.p, 0, d.ns * sizeof(int), st));
    int nsplit = (int)ceil_div(65536, d.ns);
    if (nsplit < 1) nsplit = 1;
    if (nsplit > d.nf) nsplit = (int)d.nf;
    if (nsplit > 1024) nsplit = 1024;
    const int64_t cps = ceil_div(d.nf, nsplit);
    nsplit = (int)ceil_div(d.nf, cps);
    hipLaunchKernelGGL(dense_row_degree_kernel, dim3((unsigned)ceil_div(d.ns, 64), (unsigned)nsplit), dim3(64), 0, st,
                       d.Ss.p, d.ns, d.nf, d.ns, d.alpha, d.weighted ? 1 : 0, cps, kx.p);
    SS_LAUNCH_CHECK();
    hipLaunchKernelGGL(dense_finish_degrees_kernel<float>, dim3((unsigned)ceil_div(d.nf, 256)), dim3(256), 0, st, kcol.p,
                       (const int*)nullptr, d.nf, g.kf.p, g.inv_kf.p, d.inv_kf_m1.p);
    SS_LAUNCH_CHECK();
    hipLaunchKernelGGL(dense_finish_degrees_kernel<float>, dim3((unsigned)ceil_div(d.ns, 256)), dim3(256), 0, st, kx.p,
                       g.Ys.ptr.p, d.ns, g.ks.p, g.inv_ks.p, (float*)nullptr);
    SS_LAUNCH_CHECK();
  }
  SS_HIP(hipStreamSynchronize(st));
  return SS_OK;
}

}  // namespace ss
