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

template <class T>
__device__ __forceinline__ T cut_val(T x, T alpha, int weighted) {
  // cutoff(x, alpha, weighted); a kept weight of 0 is no edge
  return (x >= alpha) ? (weighted ? x : T(1)) : T(0);
}

// ------------------------------------------------------------------ degrees of the thresholded similarity
// kf[f] = #rows with cut(S[row,f]) != 0: one wave per column (contiguous in column-major)
template <class T>
__global__ void dense_col_degree_kernel(const T* __restrict__ S, int64_t rows, int64_t cols, int64_t ld,
                                        T alpha, int weighted, int* __restrict__ deg) {
  const int lane = threadIdx.x & 63;
  const int64_t wave0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t f = wave0; f < cols; f += nwaves) {
    int n = 0;
    for (int64_t r = lane; r < rows; r += 64) n += cut_val(S[r + f * ld], alpha, weighted) != T(0) ? 1 : 0;
    for (int o = 32; o > 0; o >>= 1) n += __shfl_xor(n, o);
    if (lane == 0) deg[f] = n;
  }
}

// row counts: thread per row, lanes walk a column together; column splits add with integer atomics
template <class T>
__global__ void dense_row_degree_kernel(const T* __restrict__ S, int64_t rows, int64_t cols, int64_t ld,
                                        T alpha, int weighted, int64_t cols_per_split, int* __restrict__ deg) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= rows) return;
  const int64_t c0 = (int64_t)blockIdx.y * cols_per_split;
  const int64_t c1 = (c0 + cols_per_split < cols) ? c0 + cols_per_split : cols;
  int n = 0;
  for (int64_t c = c0; c < c1; ++c) n += cut_val(S[r + c * ld], alpha, weighted) != T(0) ? 1 : 0;
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
  int gx, gy;          // column / row blocks of the launch
};

constexpr int DBK = 32;

using f32x16 = __attribute__((ext_vector_type(16))) float;

// Workgroup = 256 threads = 4 waves as 2 x 2; a wave owns WM x WN MFMA tiles of 32 x 32, so the workgroup tile
// is (64*WM) x (64*WN): 128 x 128 (WM = WN = 2, the default) or 256 x 128 (WM = 4).  The raw similarities of
// K-step k+1 are loaded into registers before the MFMA loop of step k and thresholded / scaled / written to the
// other LDS buffer after it, so the global latency hides behind the matrix work (80 -> 99 TFLOP/s at 50k).
template <bool LOO, int WM, int WN>
__global__ void __launch_bounds__(256) transfer_dense_kernel(DenseArgs a) {
  constexpr int TM = 64 * WM, TN = 64 * WN;        // workgroup tile
  constexpr int LDA = TM + 4, LDB = TN + 4;        // 16-byte aligned LDS rows for ds_write_b128
  constexpr int ITA = TM / 32, ITB = TN / 32;      // float4 per thread and K-step
  __shared__ __align__(16) float As[2][DBK][LDA];  // two buffers: step k+1 is parked while step k is still being read
  __shared__ __align__(16) float Bs[2][DBK][LDB];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  // tile order: all row blocks of a group of 16 column blocks before the next group (the workgroups in flight then
  // share gy + 16 operand tiles per K-step instead of 1 + gx: the source operand is not re-streamed per row block)
  int bm, bn;
  {
    const int gy = a.gy, GW = 16;
    const int id = (int)blockIdx.x;
    const int grp = id / (gy * GW);
    const int local = id - grp * gy * GW;
    bm = local % gy;
    bn = grp * GW + local / gy;
  }
  const int64_t m0 = (int64_t)bm * TM, n0 = (int64_t)bn * TN;
  f32x16 acc[WM][WN];
#pragma unroll
  for (int i = 0; i < WM; ++i)
#pragma unroll
    for (int j = 0; j < WN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  const bool a_vec = (a.lda % 4 == 0) && ((reinterpret_cast<uintptr_t>(a.A) & 15) == 0);
  const bool b_vec = (a.ldb % 4 == 0) && ((reinterpret_cast<uintptr_t>(a.B) & 15) == 0);

  float4 ra[ITA], rb[ITB];
  float rw[ITA];
  // raw tile of one operand: thread t takes quads t, t+256, ... of the [DBK][T/4] float4 grid
  auto load_quad = [&](const float* S, int64_t ld, bool vec, int64_t r0, int64_t rows, int64_t k, int mq)
                       __attribute__((always_inline)) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (k < a.K) {
      const float* p = S + k * ld + r0 + mq;
      if (vec && r0 + mq + 3 < rows) v = *reinterpret_cast<const float4*>(p);
      else {
        if (r0 + mq + 0 < rows) v.x = p[0];
        if (r0 + mq + 1 < rows) v.y = p[1];
        if (r0 + mq + 2 < rows) v.z = p[2];
        if (r0 + mq + 3 < rows) v.w = p[3];
      }
    }
    return v;
  };
  auto load_raw = [&](int64_t k0) __attribute__((always_inline)) {
#pragma unroll
    for (int it = 0; it < ITA; ++it) {
      const int qd = tid + it * 256;
      const int kk = qd / (TM / 4), mq = (qd % (TM / 4)) * 4;
      ra[it] = load_quad(a.A, a.lda, a_vec, m0, a.M, k0 + kk, mq);
      rw[it] = (k0 + kk < a.K) ? a.inv_k[k0 + kk] : 0.f;
    }
#pragma unroll
    for (int it = 0; it < ITB; ++it) {
      const int qd = tid + it * 256;
      const int kk = qd / (TN / 4), mq = (qd % (TN / 4)) * 4;
      rb[it] = load_quad(a.B, a.ldb, b_vec, n0, a.N, k0 + kk, mq);
    }
  };
  // threshold, scale and park the raw registers in LDS (out-of-range rows stay zero also for alpha <= 0)
  auto store_tiles = [&](int64_t k0, int buf) __attribute__((always_inline)) {
#pragma unroll
    for (int it = 0; it < ITA; ++it) {
      const int qd = tid + it * 256;
      const int kk = qd / (TM / 4), mq = (qd % (TM / 4)) * 4;
      const float w = rw[it];
      float4 v = ra[it];
      v.x = (m0 + mq + 0 < a.M) ? cut_val(v.x, a.alpha, a.weighted) * w : 0.f;
      v.y = (m0 + mq + 1 < a.M) ? cut_val(v.y, a.alpha, a.weighted) * w : 0.f;
      v.z = (m0 + mq + 2 < a.M) ? cut_val(v.z, a.alpha, a.weighted) * w : 0.f;
      v.w = (m0 + mq + 3 < a.M) ? cut_val(v.w, a.alpha, a.weighted) * w : 0.f;
      if (LOO) {  // the feature named after the query itself is not in the fold's graph
        const int64_t d = (k0 + kk) - (a.row_begin + m0 + mq);
        if (d == 0) v.x = 0.f;
        if (d == 1) v.y = 0.f;
        if (d == 2) v.z = 0.f;
        if (d == 3) v.w = 0.f;
      }
      *reinterpret_cast<float4*>(&As[buf][kk][mq]) = v;
    }
#pragma unroll
    for (int it = 0; it < ITB; ++it) {
      const int qd = tid + it * 256;
      const int kk = qd / (TN / 4), mq = (qd % (TN / 4)) * 4;
      float4 v = rb[it];
      v.x = (n0 + mq + 0 < a.N) ? cut_val(v.x, a.alpha, a.weighted) : 0.f;
      v.y = (n0 + mq + 1 < a.N) ? cut_val(v.y, a.alpha, a.weighted) : 0.f;
      v.z = (n0 + mq + 2 < a.N) ? cut_val(v.z, a.alpha, a.weighted) : 0.f;
      v.w = (n0 + mq + 3 < a.N) ? cut_val(v.w, a.alpha, a.weighted) : 0.f;
      *reinterpret_cast<float4*>(&Bs[buf][kk][mq]) = v;
    }
  };

  // one barrier per K-step: while a wave multiplies out of buffer `cur`, the raw registers of the next step
  // (loaded during the previous MFMA loop) are thresholded and parked in the other buffer, which nobody reads
  load_raw(0);
  store_tiles(0, 0);
  __syncthreads();
  if (DBK < a.K) load_raw(DBK);
  int cur = 0;
  for (int64_t k0 = 0; k0 < a.K; k0 += DBK) {
#pragma unroll
    for (int kk = 0; kk < DBK; kk += 2) {
      const int kr = kk + (lane >> 5);
      const int c = lane & 31;
      float av[WM], bv[WN];
#pragma unroll
      for (int i = 0; i < WM; ++i) av[i] = As[cur][kr][wm * (32 * WM) + i * 32 + c];
#pragma unroll
      for (int j = 0; j < WN; ++j) bv[j] = Bs[cur][kr][wn * (32 * WN) + j * 32 + c];
#pragma unroll
      for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[j], acc[i][j], 0, 0, 0);
    }
    if (k0 + DBK < a.K) store_tiles(k0 + DBK, cur ^ 1);
    __syncthreads();
    if (k0 + 2 * DBK < a.K) load_raw(k0 + 2 * DBK);  // in flight during the next MFMA loop
    cur ^= 1;
  }
  // C/D layout of 32x32 f32 MFMA: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
#pragma unroll
  for (int i = 0; i < WM; ++i)
#pragma unroll
    for (int j = 0; j < WN; ++j) {
      const int64_t n = n0 + wn * (32 * WN) + j * 32 + (lane & 31);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t m = m0 + wm * (32 * WM) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
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
                          int64_t row_begin, int64_t nrows, float* out, int64_t ldo, bool source_rows) {
  if (nrows <= 0 || d.ns <= 0) return SS_OK;
  path_add("transfer_dense_f32_mfma");
  DenseArgs a{};
  if (loo || source_rows) { a.A = d.Ss.p + row_begin; a.lda = d.ns; }  // rows of the source similarity itself
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
  // 128 x 128 tiles (2 waves per SIMD).  SS_DENSE_TILE=256 selects the 256 x 128 variant: less L2 traffic per flop
  // but 128 accumulator + 200 other registers leave one wave per SIMD -- measured 87 vs 99 TFLOP/s at 50k
  int tm = 128;
  if (const char* e = getenv("SS_DENSE_TILE")) tm = atoi(e) == 256 ? 256 : 128;
  a.gx = (int)ceil_div(d.ns, 128);
  a.gy = (int)ceil_div(nrows, tm);
  dim3 grid((unsigned)(a.gx * a.gy));
  if (tm == 256) {
    if (loo) hipLaunchKernelGGL((transfer_dense_kernel<true, 4, 2>), grid, dim3(256), 0, ctx().stream, a);
    else hipLaunchKernelGGL((transfer_dense_kernel<false, 4, 2>), grid, dim3(256), 0, ctx().stream, a);
  } else {
    if (loo) hipLaunchKernelGGL((transfer_dense_kernel<true, 2, 2>), grid, dim3(256), 0, ctx().stream, a);
    else hipLaunchKernelGGL((transfer_dense_kernel<false, 2, 2>), grid, dim3(256), 0, ctx().stream, a);
  }
  SS_LAUNCH_CHECK();
  return SS_OK;
}

// ------------------------------------------------------------------ k-fold: degrees of the graph without one fold
// kf[f] -= #{members m : cut(S[m,f]) != 0} (one wave per feature column, lanes walk the members),
// ks[s] -= #{member features m : cut(S[s,m]) != 0} (thread per source row, coalesced along the column),
// kt[t] -= #{members m : Y[m,t] != 0} (CSR rows of the members).
template <class T>
__global__ void dense_fold_kf_kernel(const T* __restrict__ S, int64_t ld, int64_t nf, T alpha, int weighted,
                                     const int* __restrict__ members, int64_t nm, int* __restrict__ kf) {
  const int lane = threadIdx.x & 63;
  const int64_t wave0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t f = wave0; f < nf; f += nwaves) {
    int n = 0;
    for (int64_t i = lane; i < nm; i += 64) n += cut_val(S[members[i] + f * ld], alpha, weighted) != T(0) ? 1 : 0;
    for (int o = 32; o > 0; o >>= 1) n += __shfl_xor(n, o);
    if (lane == 0) kf[f] -= n;
  }
}
template <class T>
__global__ void dense_fold_ks_kernel(const T* __restrict__ S, int64_t ld, int64_t ns, T alpha, int weighted,
                                     const int* __restrict__ members, int64_t nm, int* __restrict__ ks) {
  for (int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; s < ns; s += (int64_t)gridDim.x * blockDim.x) {
    int n = 0;
    for (int64_t i = 0; i < nm; ++i) n += cut_val(S[s + (int64_t)members[i] * ld], alpha, weighted) != T(0) ? 1 : 0;
    ks[s] -= n;
  }
}
__global__ void dense_fold_kt_kernel(const int* __restrict__ yptr, const int* __restrict__ yidx,
                                     const int* __restrict__ members, int64_t nm, int* __restrict__ kt) {
  const int lane = threadIdx.x & 63;
  const int64_t wave0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t m = wave0; m < nm; m += nwaves) {
    const int g = members[m];
    for (int x = yptr[g] + lane; x < yptr[g + 1]; x += 64) atomicSub(&kt[yidx[x]], 1);
  }
}

template <class T>
int dense_fold_degrees(const Graph<T>& g, const int* members, int64_t nm, int* kf, int* ks, int* kt) {
  if (nm <= 0) return SS_OK;
  const DenseSim<T>& d = g.dense;
  hipStream_t st = ctx().stream;
  const auto cap = [](int64_t x) { return (unsigned)(x < 1 ? 1 : (x > 4096 ? 4096 : x)); };
  hipLaunchKernelGGL(dense_fold_kf_kernel<T>, dim3(cap(ceil_div(d.nf * 64, 256))), dim3(256), 0, st, d.Ss.p, d.ns, d.nf,
                     d.alpha, d.weighted ? 1 : 0, members, nm, kf);
  SS_LAUNCH_CHECK();
  hipLaunchKernelGGL(dense_fold_ks_kernel<T>, dim3(cap(ceil_div(d.ns, 256))), dim3(256), 0, st, d.Ss.p, d.ns, d.ns, d.alpha,
                     d.weighted ? 1 : 0, members, nm, ks);
  SS_LAUNCH_CHECK();
  hipLaunchKernelGGL(dense_fold_kt_kernel, dim3(cap(ceil_div(nm * 64, 256))), dim3(256), 0, st, g.Ys.ptr.p, g.Ys.idx.p,
                     members, nm, kt);
  SS_LAUNCH_CHECK();
  return SS_OK;
}

// degrees of the thresholded similarity + labels: kf (columns of cut(Ss)), ks (rows of cut(Ss) + rows of Y)
template <class T>
int dense_degrees(Graph<T>& g) {
  hipStream_t st = ctx().stream;
  DenseSim<T>& d = g.dense;
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
    hipLaunchKernelGGL(dense_col_degree_kernel<T>, dim3((unsigned)(ceil_div(d.nf * 64, 256) < 4096 ? ceil_div(d.nf * 64, 256) : 4096)),
                       dim3(256), 0, st, d.Ss.p, d.ns, d.nf, d.ns, d.alpha, d.weighted ? 1 : 0, kcol.p);
    SS_LAUNCH_CHECK();
    SS_HIP(hipMemsetAsync(kx.p, 0, d.ns * sizeof(int), st));
    int nsplit = (int)ceil_div(65536, d.ns);
    if (nsplit < 1) nsplit = 1;
    if (nsplit > d.nf) nsplit = (int)d.nf;
    if (nsplit > 1024) nsplit = 1024;
    const int64_t cps = ceil_div(d.nf, nsplit);
    nsplit = (int)ceil_div(d.nf, cps);
    hipLaunchKernelGGL(dense_row_degree_kernel<T>, dim3((unsigned)ceil_div(d.ns, 64), (unsigned)nsplit), dim3(64), 0, st,
                       d.Ss.p, d.ns, d.nf, d.ns, d.alpha, d.weighted ? 1 : 0, cps, kx.p);
    SS_LAUNCH_CHECK();
    hipLaunchKernelGGL(dense_finish_degrees_kernel<T>, dim3((unsigned)ceil_div(d.nf, 256)), dim3(256), 0, st, kcol.p,
                       (const int*)nullptr, d.nf, g.kf.p, g.inv_kf.p, d.inv_kf_m1.p);
    SS_LAUNCH_CHECK();
    hipLaunchKernelGGL(dense_finish_degrees_kernel<T>, dim3((unsigned)ceil_div(d.ns, 256)), dim3(256), 0, st, kx.p,
                       g.Ys.ptr.p, d.ns, g.ks.p, g.inv_ks.p, (T*)nullptr);
    SS_LAUNCH_CHECK();
  }
  SS_HIP(hipStreamSynchronize(st));
  return SS_OK;
}

template int dense_degrees<float>(Graph<float>&);
template int dense_degrees<double>(Graph<double>&);
template int dense_fold_degrees<float>(const Graph<float>&, const int*, int64_t, int*, int*, int*);
template int dense_fold_degrees<double>(const Graph<double>&, const int*, int64_t, int*, int*, int*);

}  // namespace ss
