// Device-side assembly of the tri-partite graph: the blocks Xq, Xs, Ys of construct's block
// adjacency (src/core.jl:165-187) become CSR operands on the GPU, with the featurize cutoff
// (src/core.jl:106-112) fused into the dense -> CSR compaction; the N x N matrices A and B of
// the reference are never formed.  Degrees are non-zero COUNTS (src/graphs.jl:9-11).
//
// Set-up path only (runs once per graph): rocPRIM supplies the scan and the stable radix sort
// used for the transposes; the per-prediction kernels live in kernels.hip.
#include <cstdlib>
#include <cstring>
#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>

#include "graph.hpp"

namespace ss {

#define SS_LAUNCH_CHECK()                                                             \
  do {                                                                                \
    hipError_t _e = hipGetLastError();                                                \
    if (_e != hipSuccess)                                                             \
      return fail(SS_EHIP, "%s:%d kernel launch: %s", __FILE__, __LINE__, hipGetErrorString(_e)); \
  } while (0)

static inline int grid_for(int64_t work, int block, int cap = 256 * 16) {
  int64_t g = ceil_div(work, block);
  if (g < 1) g = 1;
  if (g > cap) g = cap;
  return (int)g;
}

__global__ void scan_tail_kernel(const int* in, int* out, int64_t n) { out[n] = out[n - 1] + in[n - 1]; }

// exclusive scan of n ints, out[n] = total (out has n+1 entries)
static int exclusive_scan_int(const int* in, int* out, int64_t n) {
  hipStream_t st = ctx().stream;
  if (n == 0) {
    SS_HIP(hipMemsetAsync(out, 0, sizeof(int), st));
    return SS_OK;
  }
  size_t bytes = 0;
  SS_HIP(rocprim::exclusive_scan(nullptr, bytes, in, out, 0, (size_t)n, rocprim::plus<int>(), st));
  DevBuf<unsigned char> tmp;
  SS_TRY(tmp.alloc(bytes));
  SS_HIP(rocprim::exclusive_scan(tmp.p, bytes, in, out, 0, (size_t)n, rocprim::plus<int>(), st));
  // total = out[n-1] + in[n-1]
  hipLaunchKernelGGL(scan_tail_kernel, dim3(1), dim3(1), 0, st, in, out, n);
  SS_LAUNCH_CHECK();
  SS_HIP(hipStreamSynchronize(st));  // tmp is freed on return
  return SS_OK;
}

static int read_int(const int* dev, int* host) {
  SS_HIP(hipMemcpyAsync(host, dev, sizeof(int), hipMemcpyDeviceToHost, ctx().stream));
  SS_HIP(hipStreamSynchronize(ctx().stream));
  return SS_OK;
}

// ------------------------------------------------------------------ user CSR -> canonical CSR
// wave per row: validate (monotone pointers, indices in range and strictly increasing) and count
// the non-zero values; status bits: 1 bad pointer, 2 index out of range, 4 unsorted/duplicate
template <class T>
__global__ void csr_check_count_kernel(const int64_t* __restrict__ ptr, const int32_t* __restrict__ idx,
                                       const T* __restrict__ val, int64_t rows, int64_t cols, int base,
                                       int64_t nnz_in, int* __restrict__ cnt, int* __restrict__ status) {
  const int lane = threadIdx.x & 63;
  const int64_t wave0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t r = wave0; r < rows; r += nwaves) {
    const int64_t b = ptr[r] - base, e = ptr[r + 1] - base;
    if (b < 0 || e < b || e > nnz_in) {
      if (lane == 0) atomicOr(status, 1);
      if (lane == 0) cnt[r] = 0;
      continue;
    }
    int n = 0;
    for (int64_t x = b + lane; x < e; x += 64) {
      const int64_t c = (int64_t)idx[x] - base;
      if (c < 0 || c >= cols) atomicOr(status, 2);
      if (x > b && (int64_t)idx[x - 1] - base >= c) atomicOr(status, 4);
      const bool nz = val ? (val[x] != T(0)) : true;
      n += nz ? 1 : 0;
    }
    for (int o = 32; o > 0; o >>= 1) n += __shfl_xor(n, o);
    if (lane == 0) cnt[r] = n;
  }
}

template <class T>
__global__ void csr_compact_kernel(const int64_t* __restrict__ ptr, const int32_t* __restrict__ idx,
                                   const T* __restrict__ val, int64_t rows, int base, const int* __restrict__ optr,
                                   int* __restrict__ oidx, T* __restrict__ oval, int* __restrict__ not_binary) {
  const int lane = threadIdx.x & 63;
  const int64_t wave0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t r = wave0; r < rows; r += nwaves) {
    const int64_t b = ptr[r] - base, e = ptr[r + 1] - base;
    int o = optr[r];
    for (int64_t x0 = b; x0 < e; x0 += 64) {
      const int64_t x = x0 + lane;
      const bool in = x < e;
      const T v = in ? (val ? val[x] : T(1)) : T(0);
      const bool keep = in && v != T(0);
      const unsigned long long mask = __ballot(keep);
      const int pos = __popcll(mask & ((1ull << lane) - 1ull));
      if (keep) {
        oidx[o + pos] = idx[x] - base;
        oval[o + pos] = v;
        if (v != T(1)) *not_binary = 1;
      }
      o += __popcll(mask);
    }
  }
}

template <class T>
int csr_from_user(int64_t rows, int64_t cols, const int64_t* ptr, const int32_t* idx, const T* val,
                  int index_base, int mem, DevCsr<T>& out) {
  hipStream_t st = ctx().stream;
  if (rows < 0 || cols < 0) return fail(SS_EINVAL, "negative matrix size");
  if (index_base != 0 && index_base != 1) return fail(SS_EINVAL, "index_base must be 0 or 1");
  out.rows = rows;
  out.cols = cols;
  if (rows >= (1LL << 31) - 64 || cols >= (1LL << 31) - 64) return fail(SS_EUNSUPPORTED, "dimension >= 2^31");
  if (rows == 0) {
    out.nnz = 0;
    out.binary = true;
    SS_TRY(out.ptr.alloc(1));
    SS_HIP(hipMemsetAsync(out.ptr.p, 0, sizeof(int), st));
    SS_TRY(out.idx.alloc(0));
    SS_TRY(out.val.alloc(0));
    return SS_OK;
  }
  if (!ptr) return fail(SS_EINVAL, "row pointer array is NULL");
  // the last row pointer gives the stored entry count
  int64_t last = 0, first = 0;
  if (mem == SS_MEM_HOST) {
    last = ptr[rows];
    first = ptr[0];
  } else {
    SS_HIP(hipMemcpyAsync(&last, ptr + rows, sizeof(int64_t), hipMemcpyDeviceToHost, st));
    SS_HIP(hipMemcpyAsync(&first, ptr, sizeof(int64_t), hipMemcpyDeviceToHost, st));
    SS_HIP(hipStreamSynchronize(st));
  }
  if (first != index_base) return fail(SS_EINVAL, "row pointer does not start at index_base");
  const int64_t nnz_in = last - index_base;
  if (nnz_in < 0) return fail(SS_EINVAL, "negative entry count in row pointers");
  if (nnz_in >= (1LL << 31) - 64) return fail(SS_EUNSUPPORTED, "nnz >= 2^31 needs 64-bit row pointers");
  if (nnz_in > 0 && !idx) return fail(SS_EINVAL, "column index array is NULL");

  DevBuf<int64_t> dptr;
  DevBuf<int32_t> didx;
  DevBuf<T> dval;
  const int64_t* uptr = ptr;
  const int32_t* uidx = idx;
  const T* uval = val;
  if (mem == SS_MEM_HOST) {
    SS_TRY(dptr.alloc(rows + 1));
    SS_TRY(upload(dptr.p, ptr, rows + 1, mem));
    SS_TRY(didx.alloc(nnz_in));
    SS_TRY(upload(didx.p, idx, nnz_in, mem));
    if (val) {
      SS_TRY(dval.alloc(nnz_in));
      SS_TRY(upload(dval.p, val, nnz_in, mem));
      uval = dval.p;
    }
    uptr = dptr.p;
    uidx = didx.p;
  }
  DevBuf<int> cnt, flags;
  SS_TRY(cnt.alloc(rows));
  SS_TRY(flags.alloc(2));
  SS_HIP(hipMemsetAsync(flags.p, 0, 2 * sizeof(int), st));
  hipLaunchKernelGGL(csr_check_count_kernel<T>, dim3(grid_for(rows * 64, 256)), dim3(256), 0, st, uptr, uidx, uval,
                     rows, cols, index_base, nnz_in, cnt.p, flags.p);
  SS_LAUNCH_CHECK();
  int status = 0;
  SS_TRY(read_int(flags.p, &status));
  if (status & 1) return fail(SS_EINVAL, "row pointers are not monotone / exceed the entry count");
  if (status & 2) return fail(SS_EINVAL, "column index out of range");
  if (status & 4) return fail(SS_EINVAL, "column indices must be strictly increasing within a row");
  SS_TRY(out.ptr.alloc(rows + 1));
  SS_TRY(exclusive_scan_int(cnt.p, out.ptr.p, rows));
  int nnz = 0;
  SS_TRY(read_int(out.ptr.p + rows, &nnz));
  out.nnz = nnz;
  SS_TRY(out.idx.alloc(nnz));
  SS_TRY(out.val.alloc(nnz));
  hipLaunchKernelGGL(csr_compact_kernel<T>, dim3(grid_for(rows * 64, 256)), dim3(256), 0, st, uptr, uidx, uval,
                     rows, index_base, out.ptr.p, out.idx.p, out.val.p, flags.p + 1);
  SS_LAUNCH_CHECK();
  int notbin = 0;
  SS_TRY(read_int(flags.p + 1, &notbin));
  out.binary = (notbin == 0);
  return SS_OK;
}

// ------------------------------------------------------------------ dense (column-major) -> CSR with the cutoff fused
template <class T>
__device__ __forceinline__ bool keep_entry(T x, bool cut, T alpha, bool weighted, T& v) {
  if (cut) {
    // cutoff(x, alpha, weighted): x >= alpha ? (weighted ? x : 1) : 0; a kept weight of 0 is no edge
    const bool k = x >= alpha;
    v = weighted ? x : T(1);
    return k && v != T(0);
  }
  v = x;
  return x != T(0);
}

// grid: (row blocks, column splits); one thread per (row, split); counts[split*rows + row]
template <class T>
__global__ void dense_count_kernel(const T* __restrict__ S, int64_t rows, int64_t cols, int64_t ld, int cut,
                                   T alpha, int weighted, int64_t cols_per_split, int* __restrict__ counts) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= rows) return;
  const int64_t c0 = (int64_t)blockIdx.y * cols_per_split;
  const int64_t c1 = (c0 + cols_per_split < cols) ? c0 + cols_per_split : cols;
  int n = 0;
  for (int64_t c = c0; c < c1; ++c) {
    T v;
    n += keep_entry<T>(S[r + c * ld], cut != 0, alpha, weighted != 0, v) ? 1 : 0;
  }
  counts[(int64_t)blockIdx.y * rows + r] = n;
}

// per row: turn the per-split counts into per-split start offsets within the row, emit the row total
__global__ void dense_row_offsets_kernel(int* __restrict__ counts, int64_t rows, int nsplit, int* __restrict__ rowcnt) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= rows) return;
  int run = 0;
  for (int s = 0; s < nsplit; ++s) {
    const int c = counts[(int64_t)s * rows + r];
    counts[(int64_t)s * rows + r] = run;
    run += c;
  }
  rowcnt[r] = run;
}

template <class T>
__global__ void dense_fill_kernel(const T* __restrict__ S, int64_t rows, int64_t cols, int64_t ld, int cut, T alpha,
                                  int weighted, int64_t cols_per_split, const int* __restrict__ offs,
                                  const int* __restrict__ ptr, int* __restrict__ oidx, T* __restrict__ oval,
                                  int* __restrict__ not_binary) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= rows) return;
  const int64_t c0 = (int64_t)blockIdx.y * cols_per_split;
  const int64_t c1 = (c0 + cols_per_split < cols) ? c0 + cols_per_split : cols;
  int o = ptr[r] + offs[(int64_t)blockIdx.y * rows + r];
  bool nb = false;
  for (int64_t c = c0; c < c1; ++c) {
    T v;
    if (keep_entry<T>(S[r + c * ld], cut != 0, alpha, weighted != 0, v)) {
      oidx[o] = (int)c;
      oval[o] = v;
      nb |= (v != T(1));
      ++o;
    }
  }
  if (nb) *not_binary = 1;
}

template <class T>
int csr_from_dense(const T* S, int64_t rows, int64_t cols, int64_t ld, bool apply_cutoff, T alpha,
                   bool weighted, int mem, DevCsr<T>& out) {
  hipStream_t st = ctx().stream;
  if (rows < 0 || cols < 0) return fail(SS_EINVAL, "negative matrix size");
  if (rows > 0 && cols > 0 && (!S || ld < rows)) return fail(SS_EINVAL, "dense block: NULL pointer or ld < rows");
  if (rows >= (1LL << 31) - 64 || cols >= (1LL << 31) - 64) return fail(SS_EUNSUPPORTED, "dimension >= 2^31");
  out.rows = rows;
  out.cols = cols;
  SS_TRY(out.ptr.alloc(rows + 1));
  if (rows == 0 || cols == 0) {
    out.nnz = 0;
    out.binary = true;
    SS_HIP(hipMemsetAsync(out.ptr.p, 0, (rows + 1) * sizeof(int), st));
    SS_TRY(out.idx.alloc(0));
    SS_TRY(out.val.alloc(0));
    return SS_OK;
  }
  DevBuf<T> dS;
  const T* src = S;
  int64_t sld = ld;
  if (mem == SS_MEM_HOST) {
    SS_TRY(dS.alloc((size_t)rows * cols));
    SS_HIP(hipMemcpy2DAsync(dS.p, rows * sizeof(T), S, ld * sizeof(T), rows * sizeof(T), cols,
                            hipMemcpyHostToDevice, st));
    src = dS.p;
    sld = rows;
  }
  // enough (row, split) threads to fill the chip: ~64k threads
  int nsplit = (int)ceil_div(65536, rows);
  if (nsplit < 1) nsplit = 1;
  if (nsplit > cols) nsplit = (int)cols;
  if (nsplit > 1024) nsplit = 1024;
  const int64_t cps = ceil_div(cols, nsplit);
  nsplit = (int)ceil_div(cols, cps);
  DevBuf<int> counts, rowcnt, flag;
  SS_TRY(counts.alloc((size_t)nsplit * rows));
  SS_TRY(rowcnt.alloc(rows));
  SS_TRY(flag.alloc(1));
  SS_HIP(hipMemsetAsync(flag.p, 0, sizeof(int), st));
  dim3 grid((unsigned)ceil_div(rows, 64), (unsigned)nsplit);
  hipLaunchKernelGGL(dense_count_kernel<T>, grid, dim3(64), 0, st, src, rows, cols, sld, apply_cutoff ? 1 : 0, alpha,
                     weighted ? 1 : 0, cps, counts.p);
  SS_LAUNCH_CHECK();
  hipLaunchKernelGGL(dense_row_offsets_kernel, dim3((unsigned)ceil_div(rows, 64)), dim3(64), 0, st, counts.p, rows,
                     nsplit, rowcnt.p);
  SS_LAUNCH_CHECK();
  SS_TRY(exclusive_scan_int(rowcnt.p, out.ptr.p, rows));
  int nnz = 0;
  SS_TRY(read_int(out.ptr.p + rows, &nnz));
  if (nnz < 0) return fail(SS_EUNSUPPORTED, "nnz >= 2^31 needs 64-bit row pointers");
  out.nnz = nnz;
  SS_TRY(out.idx.alloc(nnz));
  SS_TRY(out.val.alloc(nnz));
  hipLaunchKernelGGL(dense_fill_kernel<T>, grid, dim3(64), 0, st, src, rows, cols, sld, apply_cutoff ? 1 : 0, alpha,
                     weighted ? 1 : 0, cps, counts.p, out.ptr.p, out.idx.p, out.val.p, flag.p);
  SS_LAUNCH_CHECK();
  int notbin = 0;
  SS_TRY(read_int(flag.p, &notbin));
  out.binary = (notbin == 0);
  return SS_OK;
}

// ------------------------------------------------------------------ transpose (stable sort by column)
__global__ void expand_rows_kernel(const int* __restrict__ ptr, int64_t rows, int* __restrict__ rowid,
                                   int* __restrict__ iota) {
  const int lane = threadIdx.x & 63;
  const int64_t wave0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t r = wave0; r < rows; r += nwaves) {
    const int b = ptr[r], e = ptr[r + 1];
    for (int x = b + lane; x < e; x += 64) {
      rowid[x] = (int)r;
      iota[x] = x;
    }
  }
}

__global__ void histogram_kernel(const int* __restrict__ keys, int64_t n, int* __restrict__ hist) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    atomicAdd(&hist[keys[i]], 1);
}

template <class T>
__global__ void gather_transpose_kernel(const int* __restrict__ perm, const int* __restrict__ rowid,
                                        const T* __restrict__ val, int64_t n, int* __restrict__ oidx,
                                        T* __restrict__ oval) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int p = perm[i];
    oidx[i] = rowid[p];
    oval[i] = val[p];
  }
}

template <class T>
int csr_transpose(const DevCsr<T>& in, DevCsr<T>& out) {
  hipStream_t st = ctx().stream;
  out.rows = in.cols;
  out.cols = in.rows;
  out.nnz = in.nnz;
  out.binary = in.binary;
  SS_TRY(out.ptr.alloc(out.rows + 1));
  SS_TRY(out.idx.alloc(in.nnz));
  SS_TRY(out.val.alloc(in.nnz));
  const int64_t n = in.nnz;
  DevBuf<int> hist;
  SS_TRY(hist.alloc(out.rows + 1));
  SS_HIP(hipMemsetAsync(hist.p, 0, (out.rows + 1) * sizeof(int), st));
  if (n == 0) {
    SS_HIP(hipMemsetAsync(out.ptr.p, 0, (out.rows + 1) * sizeof(int), st));
    return SS_OK;
  }
  DevBuf<int> rowid, iota, keys_out, perm;
  SS_TRY(rowid.alloc(n));
  SS_TRY(iota.alloc(n));
  SS_TRY(keys_out.alloc(n));
  SS_TRY(perm.alloc(n));
  hipLaunchKernelGGL(expand_rows_kernel, dim3(grid_for(in.rows * 64, 256)), dim3(256), 0, st, in.ptr.p, in.rows,
                     rowid.p, iota.p);
  SS_LAUNCH_CHECK();
  int bits = 1;
  while ((1LL << bits) < in.cols && bits < 32) ++bits;
  size_t bytes = 0;
  SS_HIP(rocprim::radix_sort_pairs(nullptr, bytes, in.idx.p, keys_out.p, iota.p, perm.p, (size_t)n, 0u,
                                   (unsigned)bits, st));
  DevBuf<unsigned char> tmp;
  SS_TRY(tmp.alloc(bytes));
  SS_HIP(rocprim::radix_sort_pairs(tmp.p, bytes, in.idx.p, keys_out.p, iota.p, perm.p, (size_t)n, 0u,
                                   (unsigned)bits, st));
  hipLaunchKernelGGL(histogram_kernel, dim3(grid_for(n, 256)), dim3(256), 0, st, in.idx.p, n, hist.p);
  SS_LAUNCH_CHECK();
  SS_TRY(exclusive_scan_int(hist.p, out.ptr.p, out.rows));
  hipLaunchKernelGGL(gather_transpose_kernel<T>, dim3(grid_for(n, 256)), dim3(256), 0, st, perm.p, rowid.p,
                     in.val.p, n, out.idx.p, out.val.p);
  SS_LAUNCH_CHECK();
  SS_HIP(hipStreamSynchronize(st));
  return SS_OK;
}

// ------------------------------------------------------------------ CSR -> chunked SELL-64 with 16-bit local indices
// one wave per (chunk, slice): lane = row; width = max entries of the 64 rows inside the chunk, in quads
__global__ void sell_width_kernel(const int* __restrict__ ptr, const int* __restrict__ idx, int64_t rows, int KC,
                                  int nslices, int nchunks, const int* __restrict__ vs, const int* __restrict__ ve,
                                  int* __restrict__ widthq) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (wave >= (int64_t)nslices * nchunks) return;
  const int c = (int)(wave / nslices), s = (int)(wave % nslices);
  const int64_t pos = (int64_t)s * 64 + lane;  // `rows` counts virtual rows when vs/ve are given
  int n = 0;
  if (pos < rows) {
    int lo = vs ? vs[pos] : ptr[pos], hi = ve ? ve[pos] : ptr[pos + 1];
    const int64_t k0 = (int64_t)c * KC, k1 = k0 + KC;
    int a = lo, b = hi;
    while (a < b) { const int m = (a + b) >> 1; if (idx[m] < k0) a = m + 1; else b = m; }
    const int first = a;
    b = hi;
    while (a < b) { const int m = (a + b) >> 1; if (idx[m] < k1) a = m + 1; else b = m; }
    n = a - first;
  }
  for (int o = 32; o > 0; o >>= 1) { const int t = __shfl_xor(n, o); n = t > n ? t : n; }
  // in quads.  (Rounds 1-2 rounded up to whole groups of four quads because the SpMM kernel consumed four at a time;
  // since round 3 its last group is consumed quad by quad: ~6 entries fewer per slice and chunk at C2 / C3, and the short
  // rows of a power-law operand no longer cost 16 slots per chunk.)
  if (lane == 0) widthq[wave] = (n + 3) >> 2;
}

template <class T>
__global__ void sell_fill_kernel(const int* __restrict__ ptr, const int* __restrict__ idx, const T* __restrict__ val,
                                 int64_t rows, int KC, int nslices, int nchunks, const int* __restrict__ off,
                                 const int* __restrict__ vs, const int* __restrict__ ve,
                                 unsigned short* __restrict__ sidx, T* __restrict__ sval) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (wave >= (int64_t)nslices * nchunks) return;
  const int c = (int)(wave / nslices), s = (int)(wave % nslices);
  const int64_t pos = (int64_t)s * 64 + lane;
  int first = 0, n = 0;
  const int64_t k0 = (int64_t)c * KC, k1 = k0 + KC;
  if (pos < rows) {
    int lo = vs ? vs[pos] : ptr[pos], hi = ve ? ve[pos] : ptr[pos + 1];
    int a = lo, b = hi;
    while (a < b) { const int m = (a + b) >> 1; if (idx[m] < k0) a = m + 1; else b = m; }
    first = a;
    b = hi;
    while (a < b) { const int m = (a + b) >> 1; if (idx[m] < k1) a = m + 1; else b = m; }
    n = a - first;
  }
  const int o = off[wave], oe = off[wave + 1];
  for (int u = o; u < oe; ++u) {
    const int64_t base = ((int64_t)u * 64 + lane) * 4;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int p = (u - o) * 4 + e;
      const bool in = p < n;
      sidx[base + e] = in ? (unsigned short)(idx[first + p] - k0) : (unsigned short)KC;
      if (sval) sval[base + e] = in ? val[first + p] : T(0);
    }
  }
}

// ds_read_b128 serves a wave in four fixed groups of 16 lanes (MI355X: {0-3,12-15,20-27}, {4-11,16-19,28-31}
// and the same +32); the 16 lanes of a group proceed in one LDS cycle only when their 16-byte slots
// (k mod 16 for a [k][QT] tile) are distinct.  The order of the non-zeros inside a row is free, so the
// fill below schedules, position by position, a different slot for each of the 16 rows of a group
// (most constrained row first, its largest remaining bucket), falling back to a conflicting entry only when a
// row has nothing else left.
__device__ __forceinline__ void b128_group_of_lane(int lane, int& grp, int& gi) {
  const int l = lane & 31;
  int g, i;
  if (l < 4) { g = 0; i = l; }
  else if (l < 12) { g = 1; i = l - 4; }
  else if (l < 16) { g = 0; i = l - 8; }
  else if (l < 20) { g = 1; i = l - 8; }
  else if (l < 28) { g = 0; i = l - 12; }
  else { g = 1; i = l - 16; }
  grp = g + ((lane >> 5) << 1);
  gi = i;
}
__device__ __forceinline__ int b128_lane_of_group(int grp, int gi) {
  const int g = grp & 1;
  int l;
  if (g == 0) l = gi < 4 ? gi : (gi < 8 ? gi + 8 : gi + 12);
  else l = gi < 8 ? gi + 4 : (gi < 12 ? gi + 8 : gi + 16);
  return l + ((grp >> 1) << 5);
}

// one wave (= one 64-thread block) per (chunk, slice)
template <class T>
__global__ void __launch_bounds__(64) sell_fill_sched_kernel(const int* __restrict__ ptr, const int* __restrict__ idx,
                                                             const T* __restrict__ val, int64_t rows, int KC,
                                                             int nslices, int nchunks, const int* __restrict__ off,
                                                             const int* __restrict__ vs, const int* __restrict__ ve,
                                                             int* __restrict__ perm, unsigned short* __restrict__ sidx,
                                                             T* __restrict__ sval, int ncls, int npid) {
  // ncls slot classes (tile rows per 256-byte LDS line: 16 for 16-byte rows, 8 for 32-byte rows), npid pieces per tile
  // row: a lane reads piece r ^ (lane & (npid - 1)) in read r, so only lanes with the same lane & (npid - 1) compete for
  // a class (claim bit = pid * ncls + class)
  __shared__ unsigned short cnt[64][17];
  __shared__ int cur[64][17];
  const int lane = threadIdx.x;
  const int64_t wave = blockIdx.x;
  const int c = (int)(wave / nslices), s = (int)(wave % nslices);
  const int64_t pos = (int64_t)s * 64 + lane;
  int first = 0, n = 0;
  const int64_t k0 = (int64_t)c * KC, k1 = k0 + KC;
  if (pos < rows) {
    int lo = vs ? vs[pos] : ptr[pos], hi = ve ? ve[pos] : ptr[pos + 1];
    int a = lo, b = hi;
    while (a < b) { const int m = (a + b) >> 1; if (idx[m] < k0) a = m + 1; else b = m; }
    first = a;
    b = hi;
    while (a < b) { const int m = (a + b) >> 1; if (idx[m] < k1) a = m + 1; else b = m; }
    n = a - first;
  }
  // bucket this row's entries by 16-byte slot
  for (int q = 0; q < ncls; ++q) cnt[lane][q] = 0;
  for (int x = 0; x < n; ++x) cnt[lane][(idx[first + x] - k0) & (ncls - 1)]++;
  int run = first;
  for (int q = 0; q < ncls; ++q) { cur[lane][q] = run; run += cnt[lane][q]; }
  for (int x = 0; x < n; ++x) {
    const int q = (int)((idx[first + x] - k0) & (ncls - 1));
    perm[cur[lane][q]++] = first + x;
  }
  for (int q = 0; q < ncls; ++q) cur[lane][q] -= cnt[lane][q];

  int grp, gi;
  b128_group_of_lane(lane, grp, gi);
  const int o = off[wave], oe = off[wave + 1];
  int left = n;
  unsigned have = 0;   // slots this row still has entries in
  for (int q = 0; q < ncls; ++q) have |= (cnt[lane][q] > 0 ? 1u : 0u) << q;
  const int psh = (lane & (npid - 1)) * ncls;   // where this lane's claim bits start
  // Per position the 16 rows of a group pick their slots MOST CONSTRAINED FIRST (round 3; before: rotating priority): the
  // row with the fewest slots that are still free picks next (ties: the longer rest), and takes the free slot it has the
  // most entries in.  On Poisson(100) rows this leaves 1.10 LDS cycles per group and position instead of 1.23 (simulated;
  // the kernel counters agree).  A row whose slots are all taken adds a conflict wherever it goes.
  for (int p = 0; p < (oe - o) * 4; ++p) {
    unsigned claimed = 0;
    int mine = -1;
    bool done = left == 0;
    for (int it = 0; it < 16; ++it) {
      if (__ballot(!done) == 0ull) break;
      const int rest = left > 4095 ? 4095 : left;
      int key = done ? 0x7fffffff : ((__popc(have & ~(claimed >> psh)) << 20) | ((4095 - rest) << 4) | gi);
#pragma unroll
      for (int m2 = 1; m2 < 16; m2 <<= 1) {
        const int other = __shfl(key, b128_lane_of_group(grp, gi ^ m2));
        key = other < key ? other : key;
      }
      const bool any = key != 0x7fffffff;
      const int wgi = key & 15;
      int choice = -1;
      if (any && !done && gi == wgi) {
        int best = -1, bestc = 0;
        for (int q = 0; q < ncls; ++q) {
          const int cq = cnt[lane][q];
          if (cq > bestc && !((claimed >> (psh + q)) & 1u)) { best = q; bestc = cq; }
        }
        if (best < 0)
          for (int q = 0; q < ncls; ++q) {
            const int cq = cnt[lane][q];
            if (cq > bestc) { best = q; bestc = cq; }
          }
        choice = best;
        mine = best;
        done = true;
      }
      const int wl = b128_lane_of_group(grp, wgi);
      const int ch = __shfl(choice, wl);
      if (any && ch >= 0) claimed |= 1u << ((wl & (npid - 1)) * ncls + ch);
    }
    const int64_t base = ((int64_t)(o + (p >> 2)) * 64 + lane) * 4 + (p & 3);
    if (mine >= 0) {
      const int e = perm[cur[lane][mine]];
      cur[lane][mine]++;
      cnt[lane][mine]--;
      if (cnt[lane][mine] == 0) have &= ~(1u << mine);
      --left;
      sidx[base] = (unsigned short)(idx[e] - k0);
      if (sval) sval[base] = val[e];
    } else {
      // padding: one of the 16 zero rows behind the tile (KC .. KC+15, spmm_sell_kernel), the one whose 16-byte slot no
      // active lane of this LDS cycle reads (the lowest free one; padding lanes of a group share it: same address)
      int q = 0;
      while (q < ncls - 1 && ((claimed >> (psh + q)) & 1u)) ++q;
      sidx[base] = (unsigned short)(KC + ((q - KC) & (ncls - 1)));
      if (sval) sval[base] = T(0);
    }
  }
}

__global__ void vrow_parts_kernel(const int* __restrict__ ptr, int64_t rows, int lmax, int* __restrict__ nparts) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < rows; i += (int64_t)gridDim.x * blockDim.x) {
    const int len = ptr[i + 1] - ptr[i];
    nparts[i] = len > lmax ? (len + lmax - 1) / lmax : 1;
  }
}
// virtual row v = part p of real row r: entries [ptr[r] + p*lmax, min(ptr[r+1], ...+lmax))
__global__ void vrow_fill_kernel(const int* __restrict__ ptr, const int* __restrict__ vfirst, int64_t rows, int lmax,
                                 int* __restrict__ vstart, int* __restrict__ vlen, int* __restrict__ iota) {
  for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < rows; r += (int64_t)gridDim.x * blockDim.x) {
    const int lo = ptr[r], hi = ptr[r + 1];
    for (int v = vfirst[r], p = 0; v < vfirst[r + 1]; ++v, ++p) {
      const int b = lo + p * lmax;
      const int e = (b + lmax < hi) ? b + lmax : hi;
      vstart[v] = b;
      vlen[v] = e > b ? e - b : 0;
      iota[v] = v;
    }
  }
}
// perm[pos] = virtual id (sorted by length, descending): ranges by position and the inverse map
__global__ void vrow_place_kernel(const int* __restrict__ perm, const int* __restrict__ vstart,
                                  const int* __restrict__ vlen, int64_t n, int* __restrict__ vs, int* __restrict__ ve,
                                  int* __restrict__ inv) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int v = perm[i];
    vs[i] = vstart[v];
    ve[i] = vstart[v] + vlen[v];
    inv[v] = (int)i;
  }
}

template <class T>
int sell_build(const DevCsr<T>& in, int KCmax, DevSell<T>& out, int qt) {
  hipStream_t st = ctx().stream;
  if (qt <= 0) qt = sell_tile_width<T>();
  const int rowb = qt * (int)sizeof(T);
  if (rowb != 16 && rowb != 32) return fail(SS_EINVAL, "SELL tile rows must be 16 or 32 bytes");
  out.qt = qt;
  const int ncls = 256 / rowb, npid = rowb / 16;
  if (KCmax < 1 || KCmax > 65520) return fail(SS_EINVAL, "SELL chunk size out of range");   // local indices KC .. KC+15 are the zero rows
  out.rows = in.rows;
  out.cols = in.cols;
  out.binary = in.binary;
  out.KC = (int)((in.cols <= KCmax) ? (in.cols > 0 ? in.cols : 1) : KCmax);
  out.nchunks = (int)(in.cols > 0 ? ceil_div(in.cols, out.KC) : 1);
  out.nslices = (int)ceil_div(in.rows, 64);
  const int64_t nws = (int64_t)out.nslices * out.nchunks;
  SS_TRY(out.off.alloc(nws + 1));
  if (nws == 0) {
    SS_HIP(hipMemsetAsync(out.off.p, 0, sizeof(int), st));
    out.nquads = 0;
    SS_TRY(out.idx.alloc(0));
    SS_TRY(out.val.alloc(0));
    return SS_OK;
  }
  DevBuf<int> widthq;
  SS_TRY(widthq.alloc(nws));
  out.sorted = false;
  hipLaunchKernelGGL(sell_width_kernel, dim3((unsigned)ceil_div(nws * 64, 256)), dim3(256), 0, st, in.ptr.p, in.idx.p,
                     in.rows, out.KC, out.nslices, out.nchunks, (const int*)nullptr, (const int*)nullptr, widthq.p);
  SS_LAUNCH_CHECK();
  SS_TRY(exclusive_scan_int(widthq.p, out.off.p, nws));
  int nq = 0;
  SS_TRY(read_int(out.off.p + nws, &nq));
  out.vrows = in.rows;
  // padded storage well above nnz means skewed row lengths inside slices: split the longest rows into
  // virtual rows of bounded length and sort all (virtual) rows by length before cutting slices
  const char* force = getenv("SS_SELL_SORT");
  // (the rounding of every slice to whole quads adds ~0.4 quads per slice and is not skew)
  const bool want_sort = force ? atoi(force) != 0 : (((double)nq - 0.4 * (double)nws) * 256.0 > 1.3 * (double)in.nnz + 65536.0);
  if (want_sort && in.rows > 64) {
    int64_t lmax = in.nnz / 4096;  // a slice of full-length virtual rows is ~1/4 of one wave's share of a workgroup
    if (const char* e = getenv("SS_SELL_LMAX")) lmax = atoll(e);
    if (lmax < 256) lmax = 256;
    if (lmax > 65536) lmax = 65536;
    lmax &= ~3LL;
    DevBuf<int> nparts, vstart, vlen, vlen_sorted, iota, perm;
    SS_TRY(nparts.alloc(in.rows));
    SS_TRY(out.vfirst.alloc(in.rows + 1));
    hipLaunchKernelGGL(vrow_parts_kernel, dim3(grid_for(in.rows, 256)), dim3(256), 0, st, in.ptr.p, in.rows, (int)lmax,
                       nparts.p);
    SS_LAUNCH_CHECK();
    SS_TRY(exclusive_scan_int(nparts.p, out.vfirst.p, in.rows));
    int nv = 0;
    SS_TRY(read_int(out.vfirst.p + in.rows, &nv));
    out.vrows = nv;
    SS_TRY(vstart.alloc(nv));
    SS_TRY(vlen.alloc(nv));
    SS_TRY(vlen_sorted.alloc(nv));
    SS_TRY(iota.alloc(nv));
    SS_TRY(perm.alloc(nv));
    SS_TRY(out.vs.alloc(nv));
    SS_TRY(out.ve.alloc(nv));
    SS_TRY(out.inv.alloc(nv));
    hipLaunchKernelGGL(vrow_fill_kernel, dim3(grid_for(in.rows, 256)), dim3(256), 0, st, in.ptr.p, out.vfirst.p, in.rows,
                       (int)lmax, vstart.p, vlen.p, iota.p);
    SS_LAUNCH_CHECK();
    size_t bytes = 0;
    SS_HIP(rocprim::radix_sort_pairs_desc(nullptr, bytes, vlen.p, vlen_sorted.p, iota.p, perm.p, (size_t)nv, 0u, 32u, st));
    DevBuf<unsigned char> tmp;
    SS_TRY(tmp.alloc(bytes));
    SS_HIP(rocprim::radix_sort_pairs_desc(tmp.p, bytes, vlen.p, vlen_sorted.p, iota.p, perm.p, (size_t)nv, 0u, 32u, st));
    hipLaunchKernelGGL(vrow_place_kernel, dim3(grid_for(nv, 256)), dim3(256), 0, st, perm.p, vstart.p, vlen.p,
                       (int64_t)nv, out.vs.p, out.ve.p, out.inv.p);
    SS_LAUNCH_CHECK();
    out.sorted = true;
    out.nslices = (int)ceil_div((int64_t)nv, 64);
    const int64_t nws2 = (int64_t)out.nslices * out.nchunks;
    SS_TRY(out.off.alloc(nws2 + 1));
    SS_TRY(widthq.alloc(nws2));
    hipLaunchKernelGGL(sell_width_kernel, dim3((unsigned)ceil_div(nws2 * 64, 256)), dim3(256), 0, st, in.ptr.p,
                       in.idx.p, (int64_t)nv, out.KC, out.nslices, out.nchunks, (const int*)out.vs.p,
                       (const int*)out.ve.p, widthq.p);
    SS_LAUNCH_CHECK();
    SS_TRY(exclusive_scan_int(widthq.p, out.off.p, nws2));
    SS_TRY(read_int(out.off.p + nws2, &nq));
    SS_HIP(hipStreamSynchronize(st));
  }
  const int64_t nws_f = (int64_t)out.nslices * out.nchunks;
  if (nq < 0 || (int64_t)nq * 256 >= (1LL << 40)) return fail(SS_EUNSUPPORTED, "SELL storage too large");
  out.nquads = nq;
  // two groups of four quads of slack: the SpMM kernel loads (and ignores) up to that much past the last slice
  const size_t slack = 8 * 256;
  SS_TRY(out.idx.alloc((size_t)nq * 256 + slack));
  SS_HIP(hipMemsetAsync(out.idx.p + (size_t)nq * 256, 0, slack * sizeof(unsigned short), st));
  if (!out.binary) {
    SS_TRY(out.val.alloc((size_t)nq * 256 + slack));
    SS_HIP(hipMemsetAsync(out.val.p + (size_t)nq * 256, 0, slack * sizeof(T), st));
  } else {
    out.val.release();
  }
  if (getenv("SS_SELL_PLAIN")) {
    hipLaunchKernelGGL(sell_fill_kernel<T>, dim3((unsigned)ceil_div(nws_f * 64, 256)), dim3(256), 0, st, in.ptr.p,
                       in.idx.p, in.val.p, out.vrows, out.KC, out.nslices, out.nchunks, out.off.p,
                       out.sorted ? (const int*)out.vs.p : (const int*)nullptr,
                       out.sorted ? (const int*)out.ve.p : (const int*)nullptr, out.idx.p,
                       out.binary ? (T*)nullptr : out.val.p);
    SS_LAUNCH_CHECK();
  } else {
    DevBuf<int> perm;
    SS_TRY(perm.alloc(in.nnz));
    hipLaunchKernelGGL(sell_fill_sched_kernel<T>, dim3((unsigned)nws_f), dim3(64), 0, st, in.ptr.p, in.idx.p, in.val.p,
                       out.vrows, out.KC, out.nslices, out.nchunks, out.off.p,
                       out.sorted ? (const int*)out.vs.p : (const int*)nullptr,
                       out.sorted ? (const int*)out.ve.p : (const int*)nullptr, perm.p, out.idx.p,
                       out.binary ? (T*)nullptr : out.val.p, ncls, npid);
    SS_LAUNCH_CHECK();
    SS_HIP(hipStreamSynchronize(st));
  }
  SS_HIP(hipStreamSynchronize(st));
  return SS_OK;
}

// ------------------------------------------------------------------ CSR -> compact sliced ELL (spmm_csell.hip)
// entries of row `row` inside chunk [k0, k0 + KC): first CSR position and count
__device__ __forceinline__ void csell_subrow(const int* __restrict__ ptr, const int* __restrict__ idx, int64_t row,
                                             int64_t rows, int64_t k0, int KC, int& first, int& n) {
  first = 0; n = 0;
  if (row < rows) {
    const int lo = ptr[row], hi = ptr[row + 1];
    const int64_t k1 = k0 + KC;
    int a = lo, b = hi;
    while (a < b) { const int m = (a + b) >> 1; if (idx[m] < k0) a = m + 1; else b = m; }
    first = a;
    b = hi;
    while (a < b) { const int m = (a + b) >> 1; if (idx[m] < k1) a = m + 1; else b = m; }
    n = a - first;
  }
}

// one wave per block (chunk, slice): pairs per lane, steps of the block (its longest lane) and its pairs
__global__ void __launch_bounds__(64) csell_count_kernel(const int* __restrict__ ptr, const int* __restrict__ idx,
                                                         int64_t rows, int KC, int nslices, int64_t nblocks,
                                                         unsigned short* __restrict__ np, int* __restrict__ steps,
                                                         int* __restrict__ pairs) {
  const int lane = threadIdx.x;
  for (int64_t b = blockIdx.x; b < nblocks; b += gridDim.x) {
    const int c = (int)(b / nslices), s = (int)(b % nslices);
    int first, n;
    csell_subrow(ptr, idx, (int64_t)s * 64 + lane, rows, (int64_t)c * KC, KC, first, n);
    const int p = (n + 1) >> 1;
    np[b * 64 + lane] = (unsigned short)p;
    int mx = p, sm = p;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const int m2 = __shfl_xor(mx, o);
      mx = m2 > mx ? m2 : mx;
      sm += __shfl_xor(sm, o);
    }
    if (lane == 0) { steps[b] = mx; pairs[b] = sm; }
  }
}

__global__ void csell_desc_kernel(const int* __restrict__ base, const int* __restrict__ steps, int64_t nblocks,
                                  int* __restrict__ desc) {
  for (int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; b <= nblocks; b += (int64_t)gridDim.x * blockDim.x) {
    desc[2 * b] = base[b];                       // base[nblocks] = all pairs: the empty block past the end
    desc[2 * b + 1] = b < nblocks ? steps[b] : 0;
  }
}

// One wave per block: writes the block's pairs in step order.  The order of a lane's entries is free; for tile rows
// narrower than an LDS line (64 / 128 bytes: 4 / 2 tile rows per 256 bytes) it is scheduled position by position: the lanes
// that read the same 16-byte slot number in the same LDS cycle (same lane & (pieces - 1) inside one of the four 16-lane
// groups ds_read_b128 is served in) take tile rows of different classes (k mod 4 / k mod 2): most constrained lane first,
// its largest remaining class; a lane with nothing else left takes a conflicting entry.
template <class T, int ROWB>
__global__ void __launch_bounds__(64) csell_fill_kernel(const int* __restrict__ ptr, const int* __restrict__ idx,
                                                        const T* __restrict__ val, int64_t rows, int KC, int nslices,
                                                        int64_t nblocks, const int* __restrict__ desc, int* __restrict__ perm,
                                                        unsigned short* __restrict__ pidx16, T* __restrict__ pval) {
  constexpr int NPC = ROWB / 16, NCL = 256 / ROWB;   // 16-byte pieces per tile row, tile rows per LDS line
  __shared__ unsigned short cnt[64][NCL + 1];
  __shared__ int cur[64][NCL + 1];
  const int lane = threadIdx.x;
  int grp, gi;
  b128_group_of_lane(lane, grp, gi);
  for (int64_t b = blockIdx.x; b < nblocks; b += gridDim.x) {
    const int c = (int)(b / nslices), s = (int)(b % nslices);
    const int64_t k0 = (int64_t)c * KC;
    int first, n;
    csell_subrow(ptr, idx, (int64_t)s * 64 + lane, rows, k0, KC, first, n);
    const int64_t base = desc[2 * b];
    const int steps = desc[2 * b + 1];
    const int npl = (n + 1) >> 1;
    if (NCL > 1) {
      for (int q = 0; q < NCL; ++q) cnt[lane][q] = 0;
      for (int x = 0; x < n; ++x) cnt[lane][(idx[first + x] - k0) & (NCL - 1)]++;
      int run = first;
      for (int q = 0; q < NCL; ++q) { cur[lane][q] = run; run += cnt[lane][q]; }
      for (int x = 0; x < n; ++x) {
        const int q = (int)((idx[first + x] - k0) & (NCL - 1));
        perm[cur[lane][q]++] = first + x;
      }
      for (int q = 0; q < NCL; ++q) cur[lane][q] -= cnt[lane][q];
    }
    int left = n;
    unsigned have = 0;   // classes this lane still has entries in
    if (NCL > 1)
      for (int q = 0; q < NCL; ++q) have |= (cnt[lane][q] > 0 ? 1u : 0u) << q;
    int64_t runp = 0;
    unsigned long long mask = 0ull;
    int rank = 0;
    for (int t = 0; t < 2 * steps; ++t) {
      int e = -1;   // CSR position of the entry this lane puts at position t
      if (NCL == 1) {
        if (t < n) e = first + t;
      } else {
        // most constrained lane first (the lane with the fewest classes that are still free for its slot number; ties: the
        // longer rest), as in the SELL builder
        unsigned claimed = 0;
        int mine = -1;
        bool done = left == 0;
        const int psh = (lane & (NPC - 1)) * NCL;
        for (int it = 0; it < 16; ++it) {
          if (__ballot(!done) == 0ull) break;
          const int rest = left > 4095 ? 4095 : left;
          int key = done ? 0x7fffffff : ((__popc(have & ~(claimed >> psh)) << 20) | ((4095 - rest) << 4) | gi);
#pragma unroll
          for (int m2 = 1; m2 < 16; m2 <<= 1) {
            const int other = __shfl(key, b128_lane_of_group(grp, gi ^ m2));
            key = other < key ? other : key;
          }
          const bool any = key != 0x7fffffff;
          const int wl = b128_lane_of_group(grp, key & 15);
          int choice = -1;
          if (any && !done && gi == (key & 15)) {
            int best = -1, bestc = 0;
            for (int q = 0; q < NCL; ++q) {
              const int cq = cnt[lane][q];
              if (cq > bestc && !((claimed >> (psh + q)) & 1u)) { best = q; bestc = cq; }
            }
            if (best < 0)
              for (int q = 0; q < NCL; ++q) {
                const int cq = cnt[lane][q];
                if (cq > bestc) { best = q; bestc = cq; }
              }
            choice = best;
            mine = best;
            done = true;
          }
          const int ch = __shfl(choice, wl);
          if (any && ch >= 0) claimed |= 1u << ((wl & (NPC - 1)) * NCL + ch);
        }
        if (mine >= 0) {
          e = perm[cur[lane][mine]];
          cur[lane][mine]++;
          cnt[lane][mine]--;
          if (cnt[lane][mine] == 0) have &= ~(1u << mine);
          --left;
        }
      }
      const int u = t >> 1, half = t & 1;
      const bool act = npl > u;
      if (half == 0) {
        mask = __ballot(act);
        rank = __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
      }
      if (act) {
        const int64_t at = (base + runp + rank) * 2 + half;
        pidx16[at] = e >= 0 ? (unsigned short)(idx[e] - k0) : (unsigned short)KC;
        if (pval) pval[at] = e >= 0 ? val[e] : T(0);
      }
      if (half == 1) runp += __builtin_popcountll(mask);
    }
  }
}

template <class T>
int csell_build(const DevCsr<T>& in, int KC, int QT, DevCsell<T>& out) {
  hipStream_t st = ctx().stream;
  if (KC < 1 || KC > 32767) return fail(SS_EINVAL, "compact sliced ELL: chunk size out of range");
  const int rowb = QT * (int)sizeof(T);
  if (rowb != 16 && rowb != 32 && rowb != 64 && rowb != 128 && rowb != 256)
    return fail(SS_EINVAL, "compact sliced ELL: tile rows must be 16, 32, 64, 128 or 256 bytes");
  out.ok = false;
  out.rows = in.rows; out.cols = in.cols; out.nnz = in.nnz; out.binary = in.binary;
  out.KC = KC; out.QT = QT;
  out.nchunks = (int)(in.cols > 0 ? ceil_div(in.cols, KC) : 1);
  out.nslices = (int)ceil_div(in.rows, 64);
  const int64_t nblocks = (int64_t)out.nchunks * out.nslices;
  if (nblocks <= 0 || nblocks >= (1LL << 24)) return SS_OK;   // (the 2-D kernel serves what does not fit here)
  DevBuf<int> steps, pairs, base;
  SS_TRY(out.np.alloc((size_t)(nblocks + 1) * 64));
  SS_TRY(steps.alloc(nblocks + 1));
  SS_TRY(pairs.alloc(nblocks + 1));
  SS_TRY(base.alloc(nblocks + 2));
  SS_HIP(hipMemsetAsync(out.np.p + (size_t)nblocks * 64, 0, 64 * sizeof(unsigned short), st));
  const unsigned grid = (unsigned)(nblocks < (1 << 20) ? nblocks : (1 << 20));
  hipLaunchKernelGGL(csell_count_kernel, dim3(grid), dim3(64), 0, st, in.ptr.p, in.idx.p, in.rows, KC, out.nslices,
                     nblocks, out.np.p, steps.p, pairs.p);
  SS_LAUNCH_CHECK();
  SS_TRY(exclusive_scan_int(pairs.p, base.p, nblocks));
  int total = 0;
  SS_TRY(read_int(base.p + nblocks, &total));
  if (total < 0) return SS_OK;
  out.npairs = total;
  SS_TRY(out.desc.alloc((size_t)(nblocks + 1) * 2));
  hipLaunchKernelGGL(csell_desc_kernel, dim3(grid_for(nblocks + 1, 256)), dim3(256), 0, st, base.p, steps.p, nblocks, out.desc.p);
  SS_LAUNCH_CHECK();
  SS_TRY(out.pidx.alloc((size_t)total + 64));
  SS_HIP(hipMemsetAsync(out.pidx.p + total, 0, 64 * sizeof(unsigned), st));
  if (!in.binary) {
    SS_TRY(out.pval.alloc(2 * ((size_t)total + 64)));
    SS_HIP(hipMemsetAsync(out.pval.p + 2 * (size_t)total, 0, 128 * sizeof(T), st));
  } else {
    out.pval.release();
  }
  DevBuf<int> perm;
  if (rowb < 256) SS_TRY(perm.alloc(in.nnz));
  unsigned short* p16 = reinterpret_cast<unsigned short*>(out.pidx.p);
  T* pv = in.binary ? (T*)nullptr : out.pval.p;
  if (rowb == 16)
    hipLaunchKernelGGL((csell_fill_kernel<T, 16>), dim3(grid), dim3(64), 0, st, in.ptr.p, in.idx.p, in.val.p, in.rows, KC,
                       out.nslices, nblocks, out.desc.p, perm.p, p16, pv);
  else if (rowb == 32)
    hipLaunchKernelGGL((csell_fill_kernel<T, 32>), dim3(grid), dim3(64), 0, st, in.ptr.p, in.idx.p, in.val.p, in.rows, KC,
                       out.nslices, nblocks, out.desc.p, perm.p, p16, pv);
  else if (rowb == 64)
    hipLaunchKernelGGL((csell_fill_kernel<T, 64>), dim3(grid), dim3(64), 0, st, in.ptr.p, in.idx.p, in.val.p, in.rows, KC,
                       out.nslices, nblocks, out.desc.p, perm.p, p16, pv);
  else if (rowb == 128)
    hipLaunchKernelGGL((csell_fill_kernel<T, 128>), dim3(grid), dim3(64), 0, st, in.ptr.p, in.idx.p, in.val.p, in.rows, KC,
                       out.nslices, nblocks, out.desc.p, perm.p, p16, pv);
  else
    hipLaunchKernelGGL((csell_fill_kernel<T, 256>), dim3(grid), dim3(64), 0, st, in.ptr.p, in.idx.p, in.val.p, in.rows, KC,
                       out.nslices, nblocks, out.desc.p, perm.p, p16, pv);
  SS_LAUNCH_CHECK();
  SS_HIP(hipStreamSynchronize(st));   // perm and the counters are freed on return
  out.ok = true;
  return SS_OK;
}
template int csell_build<float>(const DevCsr<float>&, int, int, DevCsell<float>&);
template int csell_build<double>(const DevCsr<double>&, int, int, DevCsell<double>&);

// largest and smallest non-zero |value| of an array, as fp32 bit patterns (non-negative floats order like integers)
template <class T>
__global__ void abs_range_kernel(const T* __restrict__ v, int64_t n, int* __restrict__ mm) {
  int mx = 0, mn = 0x7f7fffff;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float a = fabsf((float)v[i]);
    const int b = __float_as_int(a);
    if (a > 0.f) { mx = b > mx ? b : mx; mn = b < mn ? b : mn; }
  }
  atomicMax(&mm[0], mx);
  atomicMin(&mm[1], mn);
}

// ------------------------------------------------------------------ CSR -> column-chunked CSR (stage-1 operand)
// one thread per (chunk, row): entries of the row inside the chunk
__global__ void chunk_count_kernel(const int* __restrict__ ptr, const int* __restrict__ idx, int64_t rows, int SC,
                                   int nchunks, int align, int* __restrict__ cnt, unsigned short* __restrict__ len16,
                                   int* __restrict__ toolong) {
  const int64_t total = rows * nchunks;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i / rows);
    const int64_t r = i - (int64_t)c * rows;
    const int lo = ptr[r], hi = ptr[r + 1];
    const int64_t k0 = (int64_t)c * SC, k1 = k0 + SC;
    int a = lo, b = hi;
    while (a < b) { const int m = (a + b) >> 1; if (idx[m] < k0) a = m + 1; else b = m; }
    const int first = a;
    b = hi;
    while (a < b) { const int m = (a + b) >> 1; if (idx[m] < k1) a = m + 1; else b = m; }
    cnt[i] = (a - first + align - 1) / align;  // in units of `align` entries
    if (len16) {                               // exact entry count of the sub-row (the units only give the padded length)
      len16[i] = (unsigned short)(a - first);
      if (a - first > 65535) *toolong = 1;
    }
  }
}

// wave per (chunk, row): copy the sub-row with chunk-local indices, padded to whole units
constexpr int CHUNK_SCHED_ITERS = 4;  // sub-rows of up to 256 entries are bank-scheduled, longer ones copied in order
template <class T>
__global__ void chunk_fill_kernel(const int* __restrict__ ptr, const int* __restrict__ idx, const T* __restrict__ val,
                                  int64_t rows, int SC, int nchunks, int align, int sched,
                                  const int* __restrict__ off, unsigned short* __restrict__ oidx, T* __restrict__ oval) {
  const int lane = threadIdx.x & 63;
  const int64_t wave0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  const int64_t total = rows * nchunks;
  for (int64_t i = wave0; i < total; i += nwaves) {
    const int c = (int)(i / rows);
    const int64_t r = i - (int64_t)c * rows;
    const int64_t o = (int64_t)off[i] * align;
    const int npad = (off[i + 1] - off[i]) * align;
    if (npad == 0) continue;
    const int lo = ptr[r], hi = ptr[r + 1];
    const int64_t k0 = (int64_t)c * SC, k1 = k0 + SC;
    int a = lo, b = hi;
    while (a < b) { const int m = (a + b) >> 1; if (idx[m] < k0) a = m + 1; else b = m; }
    const int first = a;
    b = hi;
    while (a < b) { const int m = (a + b) >> 1; if (idx[m] < k1) a = m + 1; else b = m; }
    const int n = a - first;
    if (sched && n > 32 && n <= 64 * CHUNK_SCHED_ITERS) {
      // Bank schedule (stage-1 operand only).  The transfer kernel folds a sub-row into its LDS accumulators 64
      // entries per read-add-write, which the LDS serves in two groups of 32 lanes, one cycle per group when the
      // 32 addresses fall into different banks (bank = column mod 32) and one more for every extra address on a
      // busy bank.  The columns of a sub-row are distinct, so their order is free: sort the entries by bank and
      // deal them round-robin over the G = ceil(n/32) lane groups -- a bank then meets a group twice only when
      // more than G of the sub-row's columns share it.  The last group holds the remainder L = n - 32(G-1): the
      // first L*G entries go round all G groups, the rest round the first G-1.
      const int G = (n + 31) >> 5;
      const int L = n - 32 * (G - 1);
      int cnt = 0;  // lane b < 32: entries of the sub-row in bank b
      int bank[CHUNK_SCHED_ITERS], col[CHUNK_SCHED_ITERS];
#pragma unroll
      for (int j = 0; j < CHUNK_SCHED_ITERS; ++j) {
        const int x = lane + 64 * j;
        col[j] = x < n ? (int)(idx[first + x] - k0) : -1;
        bank[j] = x < n ? (col[j] & 31) : -1;
      }
      for (int b = 0; b < 32; ++b) {
        int c = 0;
#pragma unroll
        for (int j = 0; j < CHUNK_SCHED_ITERS; ++j) c += __popcll(__ballot(bank[j] == b));
        if (lane == b) cnt = c;
      }
      int base = 0;  // exclusive prefix of cnt over the banks
      for (int b = 0; b < 31; ++b) {
        const int c = __builtin_amdgcn_readlane(cnt, b);
        if (lane > b) base += c;
      }
      int run = 0;  // lane b: entries of bank b placed so far
#pragma unroll
      for (int j = 0; j < CHUNK_SCHED_ITERS; ++j) {
        int p = -1;
        for (int b = 0; b < 32; ++b) {
          const unsigned long long m = __ballot(bank[j] == b);
          const int bb = __builtin_amdgcn_readlane(base, b) + __builtin_amdgcn_readlane(run, b);
          if (bank[j] == b) p = bb + __popcll(m & ((1ull << lane) - 1ull));
          if (lane == b) run += __popcll(m);
        }
        if (p >= 0) {
          int grp, slot;
          if (p < L * G) { grp = p % G; slot = p / G; }
          else { const int q = p - L * G; grp = q % (G - 1); slot = L + q / (G - 1); }
          const int pos = grp * 32 + slot;
          oidx[o + pos] = (unsigned short)col[j];
          oval[o + pos] = val[first + lane + 64 * j];
        }
      }
      // the scheduled entries occupy positions [0, n); padded sub-rows (align 32) end in zero entries
      for (int x = n + lane; x < npad; x += 64) {
        oidx[o + x] = (unsigned short)SC;
        oval[o + x] = T(0);
      }
      continue;
    }
    for (int x = lane; x < npad; x += 64) {
      oidx[o + x] = x < n ? (unsigned short)(idx[first + x] - k0) : (unsigned short)SC;  // SC = zero sentinel
      oval[o + x] = x < n ? val[first + x] : T(0);
    }
  }
}

template <class T>
int chunked_build(const DevCsr<T>& in, int SC, int align, DevChunked<T>& out) {
  hipStream_t st = ctx().stream;
  if (SC < 1 || SC > 65535) return fail(SS_EINVAL, "chunk size out of range");
  if (align != 1 && align != 4 && align != 32) return fail(SS_EINVAL, "chunk alignment must be 1, 4 or 32");
  out.rows = in.rows;
  out.cols = in.cols;
  out.nnz = in.nnz;
  out.SC = SC;
  out.align = align;
  out.binary = in.binary;
  out.nchunks = (int)(in.cols > 0 ? ceil_div(in.cols, SC) : 1);
  const int64_t total = in.rows * out.nchunks;
  SS_TRY(out.off.alloc(total + 1));
  if (total == 0) {
    SS_HIP(hipMemsetAsync(out.off.p, 0, sizeof(int), st));
    SS_TRY(out.idx.alloc(64));
    SS_TRY(out.val.alloc(64));
    SS_HIP(hipMemsetAsync(out.idx.p, 0, 64 * sizeof(unsigned short), st));
    SS_HIP(hipMemsetAsync(out.val.p, 0, 64 * sizeof(T), st));
    return SS_OK;
  }
  DevBuf<int> cnt;
  SS_TRY(cnt.alloc(total));
  DevBuf<int> toolong;
  if (align == 32) {   // the flat-stream transfer kernel walks exact sub-row lengths (starts stay sector-aligned)
    SS_TRY(out.len.alloc(total));
    SS_TRY(toolong.alloc(1));
    SS_HIP(hipMemsetAsync(toolong.p, 0, sizeof(int), st));
  }
  hipLaunchKernelGGL(chunk_count_kernel, dim3(grid_for(total, 256)), dim3(256), 0, st, in.ptr.p, in.idx.p, in.rows, SC,
                     out.nchunks, align, cnt.p, align == 32 ? out.len.p : (unsigned short*)nullptr,
                     align == 32 ? toolong.p : (int*)nullptr);
  SS_LAUNCH_CHECK();
  if (align == 32) {
    int tl = 0;
    SS_TRY(read_int(toolong.p, &tl));
    out.len_ok = (tl == 0);
  }
  // range of the |values| (the fixed-point transfer kernel scales its sums by it)
  {
    DevBuf<int> mm;
    SS_TRY(mm.alloc(2));
    const int init[2] = {0, 0x7f7fffff};
    SS_HIP(hipMemcpyAsync(mm.p, init, sizeof(init), hipMemcpyHostToDevice, st));
    if (in.nnz > 0) {
      hipLaunchKernelGGL(abs_range_kernel<T>, dim3(grid_for(in.nnz, 256)), dim3(256), 0, st, in.val.p, in.nnz, mm.p);
      SS_LAUNCH_CHECK();
    }
    int got[2];
    SS_HIP(hipMemcpyAsync(got, mm.p, sizeof(got), hipMemcpyDeviceToHost, st));
    SS_HIP(hipStreamSynchronize(st));
    memcpy(&out.vmax, &got[0], 4);
    memcpy(&out.vmin, &got[1], 4);
    if (in.nnz == 0) out.vmax = out.vmin = 0.f;
  }
  SS_TRY(exclusive_scan_int(cnt.p, out.off.p, total));
  int units = 0;
  SS_TRY(read_int(out.off.p + total, &units));
  if (units < 0 || (int64_t)units * align >= (1LL << 31) - 64) return fail(SS_EUNSUPPORTED, "chunked operand too large");
  out.stored = (int64_t)units * align;
  // 256 entries of slack: the transfer kernels load whole waves (up to 16 bytes per lane) past the end of a sub-row
  SS_TRY(out.idx.alloc(out.stored + 256));
  SS_TRY(out.val.alloc(out.stored + 256));
  SS_HIP(hipMemsetAsync(out.idx.p + out.stored, 0, 256 * sizeof(unsigned short), st));
  SS_HIP(hipMemsetAsync(out.val.p + out.stored, 0, 256 * sizeof(T), st));
  // entry order inside a sub-row: bank-scheduled for the stage-1 operands (align 1 or 32) unless SS_CHUNK_SCHED=0
  const int sched = ((align == 1 || align == 32) && !(getenv("SS_CHUNK_SCHED") && atoi(getenv("SS_CHUNK_SCHED")) == 0)) ? 1 : 0;
  hipLaunchKernelGGL(chunk_fill_kernel<T>, dim3(grid_for(total * 64, 256)), dim3(256), 0, st, in.ptr.p, in.idx.p,
                     in.val.p, in.rows, SC, out.nchunks, align, sched, out.off.p, out.idx.p, out.val.p);
  SS_LAUNCH_CHECK();
  SS_HIP(hipStreamSynchronize(st));
  return SS_OK;
}

// ------------------------------------------------------------------ degrees + transposes of a graph
template <class T>
__global__ void degree_kernel(const int* __restrict__ pa, const int* __restrict__ pb, int64_t n, int* __restrict__ k,
                              T* __restrict__ inv) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    int d = pa[i + 1] - pa[i];
    if (pb) d += pb[i + 1] - pb[i];
    k[i] = d;
    inv[i] = d > 0 ? T(1) / T(d) : T(0);  // spread's Inf/NaN -> 0 rule (src/core.jl:367-368)
  }
}

template <class T>
int graph_finalize(Graph<T>& g) {
  hipStream_t st = ctx().stream;
  SS_TRY(csr_transpose(g.Xs, g.XsT));
  SS_TRY(csr_transpose(g.Ys, g.YsT));
  SS_TRY(g.kf.alloc(g.nf));
  SS_TRY(g.ks.alloc(g.ns));
  SS_TRY(g.kt.alloc(g.nt));
  SS_TRY(g.inv_kf.alloc(g.nf));
  SS_TRY(g.inv_ks.alloc(g.ns));
  SS_TRY(g.inv_kt.alloc(g.nt));
  // kf: sources per feature; ks: features + targets per source; kt: sources per target
  if (g.nf > 0) {
    hipLaunchKernelGGL(degree_kernel<T>, dim3(grid_for(g.nf, 256)), dim3(256), 0, st, g.XsT.ptr.p, (const int*)nullptr,
                       g.nf, g.kf.p, g.inv_kf.p);
    SS_LAUNCH_CHECK();
  }
  if (g.ns > 0) {
    hipLaunchKernelGGL(degree_kernel<T>, dim3(grid_for(g.ns, 256)), dim3(256), 0, st, g.Xs.ptr.p, g.Ys.ptr.p, g.ns,
                       g.ks.p, g.inv_ks.p);
    SS_LAUNCH_CHECK();
  }
  if (g.nt > 0) {
    hipLaunchKernelGGL(degree_kernel<T>, dim3(grid_for(g.nt, 256)), dim3(256), 0, st, g.YsT.ptr.p, (const int*)nullptr,
                       g.nt, g.kt.p, g.inv_kt.p);
    SS_LAUNCH_CHECK();
  }
  SS_HIP(hipStreamSynchronize(st));
  return SS_OK;
}

template <class T>
int graph_finalize_general(Graph<T>& g) {
  hipStream_t st = ctx().stream;
  const int64_t n = g.ns;
  SS_TRY(g.kf.alloc(n));
  SS_TRY(g.inv_kf.alloc(n));
  SS_TRY(g.ks.alloc(n));
  SS_TRY(g.inv_ks.alloc(n));
  SS_TRY(g.kt.alloc(g.nt));
  SS_TRY(g.inv_kt.alloc(g.nt));
  if (n > 0) {
    hipLaunchKernelGGL(degree_kernel<T>, dim3(grid_for(n, 256)), dim3(256), 0, st, g.XsT.ptr.p, (const int*)nullptr, n,
                       g.kf.p, g.inv_kf.p);
    SS_LAUNCH_CHECK();
    hipLaunchKernelGGL(degree_kernel<T>, dim3(grid_for(n, 256)), dim3(256), 0, st, g.XsT.ptr.p, (const int*)nullptr, n,
                       g.ks.p, g.inv_ks.p);
    SS_LAUNCH_CHECK();
  }
  if (g.nt > 0) {
    hipLaunchKernelGGL(degree_kernel<T>, dim3(grid_for(g.nt, 256)), dim3(256), 0, st, g.YsT.ptr.p, (const int*)nullptr,
                       g.nt, g.kt.p, g.inv_kt.p);
    SS_LAUNCH_CHECK();
  }
  SS_HIP(hipStreamSynchronize(st));
  return SS_OK;
}

template <class T>
int graph_finalize_general_targets(Graph<T>& g) {
  hipStream_t st = ctx().stream;
  SS_TRY(g.kt.alloc(g.nt));
  SS_TRY(g.inv_kt.alloc(g.nt));
  if (g.nt > 0) {
    hipLaunchKernelGGL(degree_kernel<T>, dim3(grid_for(g.nt, 256)), dim3(256), 0, st, g.YsT.ptr.p,
                       (const int*)nullptr, g.nt, g.kt.p, g.inv_kt.p);
    SS_LAUNCH_CHECK();
  }
  SS_HIP(hipStreamSynchronize(st));
  return SS_OK;
}

#define SS_INSTANTIATE(T)                                                                                      \
  template int graph_finalize_general_targets<T>(Graph<T>&);                                                    \
  template int csr_from_user<T>(int64_t, int64_t, const int64_t*, const int32_t*, const T*, int, int, DevCsr<T>&); \
  template int csr_from_dense<T>(const T*, int64_t, int64_t, int64_t, bool, T, bool, int, DevCsr<T>&);         \
  template int csr_transpose<T>(const DevCsr<T>&, DevCsr<T>&);                                                 \
  template int chunked_build<T>(const DevCsr<T>&, int, int, DevChunked<T>&);                                     \
  template int sell_build<T>(const DevCsr<T>&, int, DevSell<T>&, int);                                            \
  template int graph_finalize<T>(Graph<T>&);                                                                   \
  template int graph_finalize_general<T>(Graph<T>&);
SS_INSTANTIATE(float)
SS_INSTANTIATE(double)

}  // namespace ss
