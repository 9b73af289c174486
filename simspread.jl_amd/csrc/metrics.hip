// Threshold-free evaluation of one score vector on the device: AuROC, AuPRC, BEDROC and the validity ratio
// (src/performance.jl:22-89,558-560), so that a score block can be judged where it was produced.
//
// The reference evaluates a confusion matrix at every unique score (`roc(y, yhat, sort(unique(yhat)))`, a sample
// counts as predicted positive when score >= threshold) and integrates with the trapezoidal rule over exactly
// those points -- there is no (0,0) point, so the area left of the highest threshold is not counted; that is kept.
// Here: one stable descending radix sort of (score, position), a prefix sum of the labels in that order, and one
// pass over the ends of the tie groups (= the unique thresholds).  BEDROC uses the 1-based rank in
// sortperm(yhat, rev=true) order (ties by position: the sort is stable).
// rocPRIM supplies the sort and the scans; the reductions are two-pass with a fixed order (bitwise repeatable).
#include <hip/hip_runtime.h>

#include <cstring>

#include <rocprim/rocprim.hpp>

#include "graph.hpp"

namespace ss {

#define SS_LAUNCH_CHECK()                                                             \
  do {                                                                                \
    hipError_t _e = hipGetLastError();                                                \
    if (_e != hipSuccess)                                                             \
      return fail(SS_EHIP, "%s:%d kernel launch: %s", __FILE__, __LINE__, hipGetErrorString(_e)); \
  } while (0)

__global__ void iota_kernel(unsigned* __restrict__ v, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    v[i] = (unsigned)i;
}

// labels in sorted order (0/1) and "last element of its tie group" marks (position or -1)
__global__ void rank_prepare_kernel(const unsigned char* __restrict__ y, const float* __restrict__ key,
                                    const unsigned* __restrict__ pos, int64_t n, int* __restrict__ lab,
                                    int* __restrict__ endmark) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    lab[i] = y[pos[i]] != 0 ? 1 : 0;
    const bool end = (i == n - 1) || (key[i] != key[i + 1]);
    endmark[i] = end ? (int)i : -1;
  }
}

constexpr int RM_BLOCK = 256;
constexpr int RM_NOUT = 4;  // auroc, auprc, bedroc sum, non-zero scores

// ctp: inclusive prefix sum of the sorted labels; prev: exclusive running maximum of endmark (previous group end)
__global__ void __launch_bounds__(RM_BLOCK) rank_terms_kernel(const float* __restrict__ key, const int* __restrict__ lab,
                                                              const int* __restrict__ ctp, const int* __restrict__ endmark,
                                                              const int* __restrict__ prev, int64_t n, double P,
                                                              double Nn, double alpha, double* __restrict__ partial) {
  double s[RM_NOUT] = {0.0, 0.0, 0.0, 0.0};
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    if (endmark[i] >= 0 && prev[i] >= 0) {
      const int j = prev[i];
      const double tp1 = ctp[i], fp1 = (double)(i + 1) - tp1;
      const double tp0 = ctp[j], fp0 = (double)(j + 1) - tp0;
      s[0] += (fp1 / Nn - fp0 / Nn) * (tp1 / P + tp0 / P) * 0.5;
      s[1] += (tp1 / P - tp0 / P) * (tp1 / (tp1 + fp1) + tp0 / (tp0 + fp0)) * 0.5;
    }
    if (lab[i]) s[2] += exp(-alpha * (double)(i + 1) / (double)n);
    if (key[i] != 0.0f) s[3] += 1.0;
  }
  __shared__ double sh[RM_NOUT][RM_BLOCK];
#pragma unroll
  for (int q = 0; q < RM_NOUT; ++q) sh[q][threadIdx.x] = s[q];
  __syncthreads();
  for (int w = RM_BLOCK / 2; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) {
#pragma unroll
      for (int q = 0; q < RM_NOUT; ++q) sh[q][threadIdx.x] += sh[q][threadIdx.x + w];
    }
    __syncthreads();
  }
  if (threadIdx.x < RM_NOUT) partial[(size_t)blockIdx.x * RM_NOUT + threadIdx.x] = sh[threadIdx.x][0];
}

__global__ void rank_final_kernel(const double* __restrict__ partial, int nblocks, double* __restrict__ out) {
  if (threadIdx.x < RM_NOUT) {
    double s = 0.0;
    for (int b = 0; b < nblocks; ++b) s += partial[(size_t)b * RM_NOUT + threadIdx.x];
    out[threadIdx.x] = s;
  }
}

// y, yhat on the device; out4 on the host: AuROC, AuPRC, BEDROC, validity ratio
int launch_rank_metrics(const unsigned char* y, const float* yhat, int64_t n, double alpha, double* out4) {
  hipStream_t st = ctx().stream;
  if (n >= (1LL << 31) - 1) return fail(SS_EUNSUPPORTED, "rank metrics: n >= 2^31");
  DevBuf<float> key;
  DevBuf<unsigned> pos_in, pos;
  DevBuf<int> lab, ctp, endmark, prev;
  SS_TRY(key.alloc(n));
  SS_TRY(pos_in.alloc(n));
  SS_TRY(pos.alloc(n));
  const int grid = (int)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256);
  hipLaunchKernelGGL(iota_kernel, dim3(grid), dim3(256), 0, st, pos_in.p, n);
  SS_LAUNCH_CHECK();
  {
    size_t bytes = 0;
    SS_HIP(rocprim::radix_sort_pairs_desc(nullptr, bytes, yhat, key.p, pos_in.p, pos.p, (size_t)n, 0, 32, st));
    DevBuf<unsigned char> tmp;
    SS_TRY(tmp.alloc(bytes));
    SS_HIP(rocprim::radix_sort_pairs_desc(tmp.p, bytes, yhat, key.p, pos_in.p, pos.p, (size_t)n, 0, 32, st));
    SS_HIP(hipStreamSynchronize(st));
  }
  pos_in.release();
  SS_TRY(lab.alloc(n));
  SS_TRY(ctp.alloc(n));
  SS_TRY(endmark.alloc(n));
  SS_TRY(prev.alloc(n));
  hipLaunchKernelGGL(rank_prepare_kernel, dim3(grid), dim3(256), 0, st, y, key.p, pos.p, n, lab.p, endmark.p);
  SS_LAUNCH_CHECK();
  {
    size_t b1 = 0, b2 = 0;
    SS_HIP(rocprim::inclusive_scan(nullptr, b1, lab.p, ctp.p, (size_t)n, rocprim::plus<int>(), st));
    SS_HIP(rocprim::exclusive_scan(nullptr, b2, endmark.p, prev.p, -1, (size_t)n, rocprim::maximum<int>(), st));
    DevBuf<unsigned char> tmp;
    SS_TRY(tmp.alloc(b1 > b2 ? b1 : b2));
    SS_HIP(rocprim::inclusive_scan(tmp.p, b1, lab.p, ctp.p, (size_t)n, rocprim::plus<int>(), st));
    SS_HIP(rocprim::exclusive_scan(tmp.p, b2, endmark.p, prev.p, -1, (size_t)n, rocprim::maximum<int>(), st));
    SS_HIP(hipStreamSynchronize(st));
  }
  int npos = 0;
  SS_HIP(hipMemcpyAsync(&npos, ctp.p + (n - 1), sizeof(int), hipMemcpyDeviceToHost, st));
  SS_HIP(hipStreamSynchronize(st));
  const double P = (double)npos, Nn = (double)(n - npos);
  const int nblocks = grid > 1024 ? 1024 : grid;
  DevBuf<double> partial, dout;
  SS_TRY(partial.alloc((size_t)nblocks * RM_NOUT));
  SS_TRY(dout.alloc(RM_NOUT));
  hipLaunchKernelGGL(rank_terms_kernel, dim3(nblocks), dim3(RM_BLOCK), 0, st, key.p, lab.p, ctp.p, endmark.p, prev.p, n,
                     P, Nn, alpha, partial.p);
  SS_LAUNCH_CHECK();
  hipLaunchKernelGGL(rank_final_kernel, dim3(1), dim3(64), 0, st, partial.p, nblocks, dout.p);
  SS_LAUNCH_CHECK();
  double h[RM_NOUT];
  SS_HIP(hipMemcpyAsync(h, dout.p, sizeof(h), hipMemcpyDeviceToHost, st));
  SS_HIP(hipStreamSynchronize(st));
  // BEDROC (src/performance.jl:22-38)
  const double N = (double)n, Ra = P / N;
  const double rand_sum = Ra * (1.0 - exp(-alpha)) / (exp(alpha / N) - 1.0);
  const double fac = Ra * sinh(alpha / 2.0) / (cosh(alpha / 2.0) - cosh(alpha / 2.0 - alpha * Ra));
  const double cte = 1.0 / (1.0 - exp(alpha * (1.0 - Ra)));
  out4[0] = fabs(h[0]);  // 0/0 -> NaN when one class is missing, as the reference's trapz over NaN rates
  out4[1] = fabs(h[1]);
  out4[2] = h[2] * fac / rand_sum + cte;
  out4[3] = h[3] / N;
  return SS_OK;
}

}  // namespace ss
