// Hand-written gfx950 kernels of the SimSpread resource-spreading pass.
//
//   transfer_kernel      stage 1: one row of the transfer block T = (L D1^-1) Mt D2^-1 per
//                        workgroup, accumulated in LDS (the sparse x sparse -> dense product
//                        hidden in the reference's W*W, src/core.jl:413)
//   spmm_sell_kernel     stage 2, wide: F = W*R with an LDS-resident tile of QT columns of R,
//                        one lane per row of W, chunk-local 16-bit indices (the A*W^2 row
//                        gather of src/core.jl:413,421 restricted to the Nq x Nt corner)
//   spmm_csr_narrow      stage 2, narrow (B <= 64): CSR streamed once from HBM, wave per row,
//                        __shfl broadcast of (index,value) and __shfl_xor reduction
//   cutoff / row_degree / spread_dense   the element-wise pieces (src/core.jl:37-43,365-371,
//                        src/graphs.jl:9-11)
//
// Wavefront = 64 lanes everywhere.  No float atomics anywhere: every sum has a fixed order, so
// results are bitwise reproducible from run to run.
#include <type_traits>

#include "graph.hpp"

namespace ss {

static inline int grid_1d(int64_t work, int block, int cap = 256 * 16) {
  int64_t g = ceil_div(work, block);
  if (g < 1) g = 1;
  if (g > cap) g = cap;
  return (int)g;
}

#define SS_LAUNCH_CHECK()                                                             \
  do {                                                                                \
    hipError_t _e = hipGetLastError();                                                \
    if (_e != hipSuccess)                                                             \
      return fail(SS_EHIP, "%s:%d kernel launch: %s", __FILE__, __LINE__, hipGetErrorString(_e)); \
  } while (0)

// ============================================================== element-wise
template <class T>
__global__ void cutoff_kernel(const T* __restrict__ X, int64_t rows, int64_t cols, int64_t ld, T alpha,
                              int weighted, T* __restrict__ out, int64_t ldo) {
  const int64_t total = rows * cols;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t c = i / rows, r = i - c * rows;
    const T x = X[r + c * ld];
    out[r + c * ldo] = (x >= alpha) ? (weighted ? x : T(1)) : T(0);
  }
}

template <class T>
int launch_cutoff(const T* X, int64_t rows, int64_t cols, int64_t ld, T alpha, bool weighted, T* out, int64_t ldo) {
  if (rows * cols == 0) return SS_OK;
  hipLaunchKernelGGL(cutoff_kernel<T>, dim3(grid_1d(rows * cols, 256)), dim3(256), 0, ctx().stream, X, rows,
                     cols, ld, alpha, weighted ? 1 : 0, out, ldo);
  SS_LAUNCH_CHECK();
  return SS_OK;
}

// k(G): one thread per row, lanes walk the same column together (coalesced in column-major)
template <class T>
__global__ void row_degree_kernel(const T* __restrict__ G, int64_t rows, int64_t cols, int64_t ld,
                                  int* __restrict__ deg) {
  for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < rows;
       r += (int64_t)gridDim.x * blockDim.x) {
    int n = 0;
    for (int64_t c = 0; c < cols; ++c) n += (G[r + c * ld] != T(0)) ? 1 : 0;
    deg[r] = n;
  }
}

template <class T>
int launch_row_degree(const T* G, int64_t rows, int64_t cols, int64_t ld, int* deg) {
  if (rows == 0) return SS_OK;
  hipLaunchKernelGGL(row_degree_kernel<T>, dim3(grid_1d(rows, 64)), dim3(64), 0, ctx().stream, G, rows, cols,
                     ld, deg);
  SS_LAUNCH_CHECK();
  return SS_OK;
}

template <class T>
__global__ void spread_dense_kernel(const T* __restrict__ G, int64_t rows, int64_t cols, int64_t ld,
                                    const int* __restrict__ deg, T* __restrict__ W, int64_t ldw) {
  const int64_t total = rows * cols;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t c = i / rows, r = i - c * rows;
    const int d = deg[r];
    W[r + c * ldw] = d > 0 ? G[r + c * ld] / T(d) : T(0);
  }
}

template <class T>
int launch_spread_dense(const T* G, int64_t rows, int64_t cols, int64_t ld, const int* deg, T* W, int64_t ldw) {
  if (rows * cols == 0) return SS_OK;
  hipLaunchKernelGGL(spread_dense_kernel<T>, dim3(grid_1d(rows * cols, 256)), dim3(256), 0, ctx().stream, G,
                     rows, cols, ld, deg, W, ldw);
  SS_LAUNCH_CHECK();
  return SS_OK;
}

// ============================================================== stage 1: transfer rows
template <class T>
struct ChunkedView {
  const int* off;            // [nchunks*rows + 1]
  const unsigned short* idx; // chunk-local column
  const T* val;
  int64_t rows;
};
template <class T>
static inline ChunkedView<T> view(const DevChunked<T>& m) {
  return ChunkedView<T>{m.off.p, m.idx.p, m.val.p, m.rows};
}

template <class T>
struct TransferArgs {
  int nterms;
  CsrView<T> L[2];
  const T* inv1[2];
  ChunkedView<T> M[2];
  const T* inv2;
  const int* kf;  // LOO: integer degrees
  const int* ks;
  int64_t row_begin;
  const int* row_ids;  // optional: row of L handled by workgroup-row r is row_ids[row_begin + r] (k-fold members)
  int64_t nj;
  int SC;       // columns of T per workgroup (= chunk of the Mt operands)
  int nchunks;
  T* out;
  int64_t ld;
  int accumulate;  // add to what `out` already holds (dense regime: the feature path came from the GEMM)
  float xmax;      // FIX: largest |value| of the Mt operand
  int small_weighted;  // host only: weighted operand of at most 32 MiB (picks the default batch size)
  int chunk_interleave;  // SS_TRANSFER_ORDER=0: rows take all their chunks in turn (the order of rounds 1-2)
};
template <class T>
__device__ __forceinline__ bool getenv_chunk_interleave(const TransferArgs<T>& p) { return p.chunk_interleave != 0; }

// One single-wave workgroup per (row r of L, column chunk c of T); c = blockIdx % nchunks so that,
// with the dispatcher dealing workgroups round-robin over the 8 XCDs, chunk c is always served by
// the same XCD and the sub-rows of Mt it touches stay in that XCD's L2 (speed only -- any placement
// is correct).  acc[j] (LDS, owned by this one wave) collects sum_a L[r,a]*inv1[a]*Mt[a][c*SC + j].
// The wave takes U neighbours a of r at a time, issues all their global loads, then folds the
// sub-rows in one after the other with plain LDS read-add-write: the columns inside one sub-row are
// distinct and LDS operations of one wave execute in order, so no atomics are needed and the sum
// order is fixed (bitwise reproducible).  Measured on MI355X: ds_add_f32 costs ~193 clk per
// wave-instruction (lanes serialised), ds_add_f64 ~27, a plain read-add-write pair ~17
// (tools/lds_atomic_bench.hip) -- which is why this is not an LDS-atomic kernel.
constexpr int TRANSFER_THREADS = 64;

template <class T> struct BitsOf;
template <> struct BitsOf<float> {
  static __device__ __forceinline__ float bcast(float v, int src) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src));
  }
};
template <> struct BitsOf<double> {
  static __device__ __forceinline__ double bcast(double v, int src) {
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffll), src);
    const int hi = __builtin_amdgcn_readlane((int)(b >> 32), src);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
  }
};

// U sub-rows: scalar bounds + coefficient, and the (unconditional) loads of their first 64 entries.
// The operand arrays carry 64 entries of slack, so lanes past the end of a sub-row read valid memory;
// such lanes are steered to a per-lane dummy accumulator behind the chunk.
template <class T, int U>
struct SubRows {
  int n[U];
  unsigned b[U];
  T cf[U];
  unsigned short j[U];
  T v[U];
  unsigned short j2[U];  // entries 64..127 of the sub-row, fetched only when it is that long (wave-uniform)
  T v2[U];
};

// The same loads as buffer instructions: resource descriptor (SGPRs) + scalar byte offset of the sub-row + a 32-bit lane
// offset.  A global_load with per-lane 64-bit addresses hands the texture addresser 512 bytes of address per
// wave-instruction, the buffer form 256; stage 1 is bound by that unit (TA busy 92 % of the kernel at ~7 clk per load,
// profiles/r03_stage1_mem_pmc.txt).
template <int U, bool BINM>
__device__ __forceinline__ void subrows_load_buf(SubRows<float, U>& s, int first, int b_l, int n_l, float cf_l,
                                                 __amdgpu_buffer_rsrc_t ridx, __amdgpu_buffer_rsrc_t rval, int lane) {
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int src = first + u;
    s.b[u] = (unsigned)__builtin_amdgcn_readlane(b_l, src);
    s.n[u] = __builtin_amdgcn_readlane(n_l, src);
    s.cf[u] = BitsOf<float>::bcast(cf_l, src);
    s.j[u] = __builtin_amdgcn_raw_buffer_load_b16(ridx, lane * 2, (int)(s.b[u] * 2u), 0);
    s.v[u] = BINM ? 1.f : __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rval, lane * 4, (int)(s.b[u] * 4u), 0));
    if (s.n[u] > 64) {
      s.j2[u] = __builtin_amdgcn_raw_buffer_load_b16(ridx, lane * 2, (int)(s.b[u] * 2u + 128u), 0);
      s.v2[u] = BINM ? 1.f : __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rval, lane * 4, (int)(s.b[u] * 4u + 256u), 0));
    }
  }
}

template <class T, int U, bool BINM>
__device__ __forceinline__ void subrows_load(SubRows<T, U>& s, int first, int b_l, int n_l, T cf_l,
                                             const unsigned short* __restrict__ midx, const T* __restrict__ mval,
                                             int lane) {
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int src = first + u;  // < 64 by construction
    s.b[u] = (unsigned)__builtin_amdgcn_readlane(b_l, src);
    s.n[u] = __builtin_amdgcn_readlane(n_l, src);
    s.cf[u] = BitsOf<T>::bcast(cf_l, src);
    const unsigned short* pi = midx + s.b[u];
    const T* pv = mval + s.b[u];
    s.j[u] = pi[lane];
    s.v[u] = BINM ? T(1) : pv[lane];  // unweighted features: every stored value is 1, two thirds of the bytes vanish
    // With the chunk sized for a mean sub-row of ~60 entries, ~40 % of the sub-rows are longer than one
    // wave: their second half must be prefetched like the first (a dependent load inside the fold
    // stalls the whole batch), so it is issued here under a wave-uniform branch.
    if (s.n[u] > 64) {
      s.j2[u] = pi[64 + lane];
      s.v2[u] = BINM ? T(1) : pv[64 + lane];
    }
  }
}

// DUAL: two copies of the accumulators (acc and acc + astride); sub-rows 2i and 2i+1 of a batch go to different
// copies, so their read-add-write pairs are independent and both LDS reads are in flight together (with one copy
// every pair waits for the previous pair's LDS round trip).  The copies are added when T is written out: the sum
// order is still fixed.
template <class T, int U, bool DUAL>
__device__ __forceinline__ void subrows_fold(const SubRows<T, U>& s, int first, int b_l, int n_l, T cf_l,
                                             T* __restrict__ acc, int astride, int dummy,
                                             const unsigned short* __restrict__ midx, const T* __restrict__ mval,
                                             int lane) {
  int nmax = 0;
  if constexpr (DUAL) {
    T* __restrict__ acc1 = acc + astride;
#pragma unroll
    for (int u = 0; u < U; u += 2) {
      const int ja = lane < s.n[u] ? (int)s.j[u] : dummy;
      const int jb = lane < s.n[u + 1] ? (int)s.j[u + 1] : dummy;
      const T ra = acc[ja], rb = acc1[jb];
      acc[ja] = fma(s.cf[u], s.v[u], ra);
      acc1[jb] = fma(s.cf[u + 1], s.v[u + 1], rb);
      if (s.n[u] > 64 || s.n[u + 1] > 64) {
        const int ka = (s.n[u] > 64 && 64 + lane < s.n[u]) ? (int)s.j2[u] : dummy;
        const int kb = (s.n[u + 1] > 64 && 64 + lane < s.n[u + 1]) ? (int)s.j2[u + 1] : dummy;
        const T va = s.n[u] > 64 ? s.v2[u] : T(0), vb = s.n[u + 1] > 64 ? s.v2[u + 1] : T(0);
        const T qa = acc[ka], qb = acc1[kb];
        acc[ka] = fma(s.cf[u], va, qa);
        acc1[kb] = fma(s.cf[u + 1], vb, qb);
      }
      nmax = s.n[u] > nmax ? s.n[u] : nmax;
      nmax = s.n[u + 1] > nmax ? s.n[u + 1] : nmax;
    }
  } else {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int j = lane < s.n[u] ? (int)s.j[u] : dummy;
      acc[j] = fma(s.cf[u], s.v[u], acc[j]);
      if (s.n[u] > 64) {
        const int k = 64 + lane < s.n[u] ? (int)s.j2[u] : dummy;
        acc[k] = fma(s.cf[u], s.v2[u], acc[k]);
      }
      nmax = s.n[u] > nmax ? s.n[u] : nmax;
    }
  }
  if (nmax > 128) {  // rare: a sub-row longer than two waves (bounds re-read by lane index, no register arrays)
#pragma unroll 1
    for (int u = 0; u < U; ++u) {
      const int n = __builtin_amdgcn_readlane(n_l, first + u);
      const unsigned b = (unsigned)__builtin_amdgcn_readlane(b_l, first + u);
      const T cf = BitsOf<T>::bcast(cf_l, first + u);
      for (int x = 128 + lane; x < n; x += 64) {
        const int j = midx[b + x];
        acc[j] = fma(cf, mval[b + x], acc[j]);
      }
    }
  }
}

// Fixed-point fold (fp32 graphs): the product cf * v is scaled by 2^k (k per row of L, see transfer_block_kernel),
// rounded to an integer and added with ds_add_u32.  An LDS integer atomic costs ~8 clk per wave-instruction on random
// addresses where the plain read-add-write pair costs ~15 (tools/lds_atomic_bench.hip) -- and stage 1 runs at exactly
// the LDS rate of its scatter -- it returns nothing (no wait, no dependent chain through the LDS round trip) and integer
// addition commutes: the sums do not depend on any order, so the result is bitwise reproducible by construction.
// Lanes past the end of a sub-row add 0 to whatever valid column they happened to read.
__device__ __forceinline__ int cvt_rpi(float x) {
  int r;
  asm("v_cvt_rpi_i32_f32 %0, %1" : "=v"(r) : "v"(x));   // floor(x + 0.5)
  return r;
}
template <int U>
__device__ __forceinline__ void subrows_fold_fixed(const SubRows<float, U>& s, int first, int b_l, int n_l, float cf_l,
                                                   unsigned* __restrict__ acc, const unsigned short* __restrict__ midx,
                                                   const float* __restrict__ mval, int lane) {
  int nmax = 0;
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const float v = lane < s.n[u] ? s.v[u] : 0.f;
    atomicAdd(&acc[s.j[u]], (unsigned)cvt_rpi(s.cf[u] * v));
    if (s.n[u] > 64) {
      const float v2 = 64 + lane < s.n[u] ? s.v2[u] : 0.f;
      atomicAdd(&acc[s.j2[u]], (unsigned)cvt_rpi(s.cf[u] * v2));
    }
    nmax = s.n[u] > nmax ? s.n[u] : nmax;
  }
  if (nmax > 128) {
#pragma unroll 1
    for (int u = 0; u < U; ++u) {
      const int n = __builtin_amdgcn_readlane(n_l, first + u);
      const unsigned b = (unsigned)__builtin_amdgcn_readlane(b_l, first + u);
      const float cf = BitsOf<float>::bcast(cf_l, first + u);
      for (int x = 128 + lane; x < n; x += 64) atomicAdd(&acc[midx[b + x]], (unsigned)cvt_rpi(cf * mval[b + x]));
    }
  }
}

template <class T, bool LOO, int U, bool BINM, bool DUAL, bool FIX = false, bool BUF = false>
__global__ void __launch_bounds__(TRANSFER_THREADS) transfer_kernel(TransferArgs<T> p) {
  static_assert(64 % (2 * U) == 0, "U must divide 32");
  extern __shared__ __align__(16) unsigned char smem_raw[];
  T* acc = reinterpret_cast<T*>(smem_raw);                        // [SC] sums + [64] per-lane dummies (x2 when DUAL)
  const int astride = p.SC + 64;
  constexpr int NCOPY = DUAL ? 2 : 1;
  unsigned* bits = reinterpret_cast<unsigned*>(acc + NCOPY * astride);  // LOO only: source owns the dropped feature
  const int lane = threadIdx.x;
  // Workgroup i runs on XCD i % 8 and chunk c on XCD c % 8 (the chunk count is a multiple of 8).  With more than 8
  // chunks the groups of 8 chunks are walked ONE AFTER THE OTHER over all rows (round 3; before: every row took all its
  // chunks in turn): what the 8 XCDs re-read from row to row is then 8 chunk slices of X' instead of all of it -- at
  // C3 200 MB instead of 600 MB, i.e. inside the 256 MB Infinity Cache instead of in HBM.  8 chunks (C2): same order as before.
  int c;
  int64_t r;
  if (p.nchunks > 8 && p.nchunks % 8 == 0 && !getenv_chunk_interleave(p)) {
    const int64_t nrows = (int64_t)gridDim.x / p.nchunks, per = nrows * 8;
    const int64_t grp = blockIdx.x / per, rem = blockIdx.x - grp * per;
    r = rem >> 3;
    c = (int)(grp * 8 + (rem & 7));
  } else {
    c = blockIdx.x % p.nchunks;
    r = blockIdx.x / p.nchunks;
  }
  const int64_t gr = p.row_ids ? (int64_t)p.row_ids[p.row_begin + r] : p.row_begin + r;
  const int64_t j0 = (int64_t)c * p.SC;
  const int jn = (int)((p.nj - j0 < p.SC) ? (p.nj - j0) : p.SC);
  const int dummy = p.SC + lane;

  for (int j = lane; j < NCOPY * astride; j += 64) acc[j] = T(0);
  if (LOO)
    for (int j = lane; j < (p.SC + 31) / 32; j += 64) bits[j] = 0u;
  __syncthreads();

  float fix_unscale = 1.f;
  for (int t = 0; t < p.nterms; ++t) {
    const CsrView<T> L = p.L[t];
    const ChunkedView<T> M = p.M[t];
    const int* __restrict__ off = M.off + (int64_t)c * M.rows;
    const unsigned short* __restrict__ midx = M.idx;
    const T* __restrict__ mval = M.val;
    const int lb = L.ptr[gr], le = L.ptr[gr + 1];
    float scale = 1.f;
    if constexpr (FIX) {   // (experiment: fixed-point sums in the single-wave kernel; one term, not LOO)
      float sabs = 0.f;
      for (int q = lb + lane; q < le; q += 64) sabs += fabsf((float)(L.val[q] * p.inv1[t][L.idx[q]]));
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) sabs += __shfl_xor(sabs, o);
      int e = 0;
      (void)frexpf(sabs * p.xmax, &e);
      if (!(sabs * p.xmax > 0.f)) e = 0;
      e = e < -90 ? -90 : (e > 120 ? 120 : e);
      scale = ldexpf(1.f, 30 - e);
      fix_unscale = ldexpf(1.f, e - 30);
    }
    for (int g0 = lb; g0 < le; g0 += 64) {
      // ---- the next 64 neighbours a of r, one per lane: coefficient and sub-row bounds
      const int q = g0 + lane;
      int b_l = 0, n_l = 0;
      T cf_l = T(0);
      if (q < le) {
        const int a = L.idx[q];
        const T lv = L.val[q];
        if (LOO) {
          const int d = p.kf[a] - 1;  // the query leaves every feature column it touched
          cf_l = (a != (int)gr && d > 0) ? lv * (T(1) / T(d)) : T(0);  // a == gr: the dropped feature
        } else {
          cf_l = lv * p.inv1[t][a];
          if constexpr (FIX) cf_l = (T)((float)cf_l * scale);
        }
        if (cf_l != T(0)) { b_l = off[a]; n_l = off[a + 1] - b_l; }
      }
      const int cnt = __builtin_amdgcn_readfirstlane((le - g0 < 64) ? (le - g0) : 64);
      // ---- fold the sub-rows in, U at a time; the loads of the next U are in flight meanwhile
      SubRows<T, U> A, B;
      auto load = [&](SubRows<T, U>& sr, int first) __attribute__((always_inline)) {
        if constexpr (BUF) {
          const __amdgpu_buffer_rsrc_t ridx = __builtin_amdgcn_make_buffer_rsrc((void*)midx, 0, -1, 0x00020000);
          const __amdgpu_buffer_rsrc_t rval = __builtin_amdgcn_make_buffer_rsrc((void*)mval, 0, -1, 0x00020000);
          subrows_load_buf<U, BINM>(sr, first, b_l, n_l, cf_l, ridx, rval, lane);
        } else {
          subrows_load<T, U, BINM>(sr, first, b_l, n_l, cf_l, midx, mval, lane);
        }
      };
      load(A, 0);
      for (int u0 = 0; u0 < cnt; u0 += 2 * U) {
        if (u0 + U < cnt) load(B, u0 + U);
        if constexpr (FIX) subrows_fold_fixed<U>(A, u0, b_l, n_l, cf_l, reinterpret_cast<unsigned*>(acc), midx, mval, lane);
        else subrows_fold<T, U, DUAL>(A, u0, b_l, n_l, cf_l, acc, astride, dummy, midx, mval, lane);
        if (u0 + U < cnt) {
          if (u0 + 2 * U < cnt) load(A, u0 + 2 * U);
          if constexpr (FIX) subrows_fold_fixed<U>(B, u0 + U, b_l, n_l, cf_l, reinterpret_cast<unsigned*>(acc), midx, mval, lane);
          else subrows_fold<T, U, DUAL>(B, u0 + U, b_l, n_l, cf_l, acc, astride, dummy, midx, mval, lane);
        }
      }
    }
  }
  if (LOO) {
    // sources that own the dropped feature column f_i: their degree is one lower in this fold
    const ChunkedView<T> M = p.M[0];
    const int* off = M.off + (int64_t)c * M.rows;
    for (int x = off[gr] + lane; x < off[gr + 1]; x += 64) {
      const int j = M.idx[x];
      atomicOr(&bits[j >> 5], 1u << (j & 31));
    }
  }
  __syncthreads();

  T* orow = p.out + r * p.ld + j0;
  for (int j = lane; j < jn; j += 64) {
    T z;
    T sum = DUAL ? acc[j] + acc[astride + j] : acc[j];
    if constexpr (FIX) sum = (T)((float)reinterpret_cast<const int*>(acc)[j] * fix_unscale);
    if (LOO) {
      const int d = p.ks[j0 + j] - (int)((bits[j >> 5] >> (j & 31)) & 1u);
      z = (d > 0 && (j0 + j) != gr) ? sum * (T(1) / T(d)) : T(0);
    } else {
      z = sum * p.inv2[j0 + j];
      if (p.accumulate) z += orow[j];
    }
    orow[j] = z;
  }
}

constexpr int TRANSFER_U = 8;

// SS_TRANSFER_DUAL=0/1: one or two copies of the LDS accumulators (see subrows_fold)
static bool transfer_dual() {
  const char* e = getenv("SS_TRANSFER_DUAL");
  return e ? atoi(e) != 0 : false;
}

// sub-rows in flight per wave.  Default 8; 4 (55 instead of 90 registers: 30 instead of 20 single-wave workgroups per CU)
// where it measured faster: weighted operands that stay in L2 (C2: 1.46 vs 1.51 ms; pattern-only operands 1.24 vs
// 1.19 ms and the 600 MB operand of C3 5.1 vs 4.6 ms go the other way).  SS_TRANSFER_U overrides.
static int transfer_u(bool small_weighted = false) {
  const char* e = getenv("SS_TRANSFER_U");
  const int u = e ? atoi(e) : (small_weighted ? 4 : TRANSFER_U);
  return (u == 4 || u == 16) ? u : 8;
}

// ---------------------------------------------------------------- stage 1, query-block workgroups (round 3)
// Same product, same per-wave loop (U sub-rows in flight, plain LDS read-add-write), different workgroup: NW waves,
// one row of L per wave, all on chunk c.  What the single-wave kernel fetched per (row, feature) from L2 besides the
// sub-row itself -- off[a], off[a+1] (one 64-byte sector for 8 bytes) and inv1[a] (another sector for 4 bytes), i.e.
// ~130 of the ~600 bytes of L2 -> L1 traffic per sub-row visit -- is gone: the chunk's sub-row offsets ((rows+1) ints)
// are staged once per workgroup in LDS and read from there, and the coefficients L[r,a] * inv1[a] come from a
// coalesced array written by transfer_coef_kernel (one pass over L per launch).  The operand may be cut with
// sub-rows padded to 32 entries (64-byte index runs, 128-byte value runs: every sub-row starts on an L2 sector).
template <class T>
__global__ void transfer_coef_kernel(const int* __restrict__ ptr, const int* __restrict__ idx, const T* __restrict__ val,
                                     const T* __restrict__ inv1, int64_t row_begin, int64_t nrows, T* __restrict__ coef) {
  const int lo = ptr[row_begin], hi = ptr[row_begin + nrows];
  for (int64_t e = (int64_t)lo + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < hi; e += (int64_t)gridDim.x * blockDim.x)
    coef[e] = val[e] * inv1[idx[e]];
}

template <class T, int U, bool BINM, bool FIX>
__global__ void __launch_bounds__(1024) transfer_block_kernel(TransferArgs<T> p, const T* __restrict__ coef, int64_t nrows,
                                                              int align, int offs_bytes, float xmax) {
  extern __shared__ __align__(16) unsigned char smem_raw[];
  int* offs = reinterpret_cast<int*>(smem_raw);                               // [rows + 1] sub-row offsets of chunk c
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const int astride = p.SC + 64;
  T* acc = reinterpret_cast<T*>(smem_raw + offs_bytes) + (size_t)wave * astride;  // this wave's sums + per-lane dummies
  const int c = blockIdx.x % p.nchunks;
  const int64_t r = (int64_t)(blockIdx.x / p.nchunks) * nw + wave;
  const ChunkedView<T> M = p.M[0];
  const CsrView<T> L = p.L[0];
  {
    const int* __restrict__ off = M.off + (int64_t)c * M.rows;
    for (int64_t i = threadIdx.x; i <= M.rows; i += blockDim.x) offs[i] = off[i];
  }
  for (int j = lane; j < astride; j += 64) acc[j] = T(0);
  __syncthreads();
  if (r >= nrows) return;   // (after the only barrier)
  const int64_t gr = p.row_begin + r;
  const int64_t j0 = (int64_t)c * p.SC;
  const int jn = (int)((p.nj - j0 < p.SC) ? (p.nj - j0) : p.SC);
  const int dummy = p.SC + lane;
  const unsigned short* __restrict__ midx = M.idx;
  const T* __restrict__ mval = M.val;
  const int lb = L.ptr[gr], le = L.ptr[gr + 1];
  // FIX: scale 2^k of this row's fixed-point sums.  Every sum is bounded by bound = (sum_a |coef[a]|) * xmax
  // (xmax = largest |value| of the operand); bound < 2^e, so with k = 30 - e all sums stay below 2^30, the roundings
  // (one unit per term at most) below 2^24: no overflow in 32 bits, resolution bound * 2^-30.
  float scale = 1.f, unscale = 1.f;
  if constexpr (FIX) {
    float sabs = 0.f;
    for (int q = lb + lane; q < le; q += 64) sabs += fabsf((float)coef[q]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sabs += __shfl_xor(sabs, o);
    int e = 0;
    (void)frexpf(sabs * xmax, &e);
    if (!(sabs * xmax > 0.f)) e = 0;
    e = e < -90 ? -90 : (e > 120 ? 120 : e);
    scale = ldexpf(1.f, 30 - e);
    unscale = ldexpf(1.f, e - 30);
  }
  for (int g0 = lb; g0 < le; g0 += 64) {
    const int q = g0 + lane;
    int b_l = 0, n_l = 0;
    T cf_l = T(0);
    if (q < le) {
      const int a = L.idx[q];
      cf_l = coef[q];
      if constexpr (FIX) cf_l = (T)((float)cf_l * scale);   // exact: a power of two
      if (cf_l != T(0)) {
        const int o0 = offs[a], o1 = offs[a + 1];
        b_l = o0 * align;
        n_l = (o1 - o0) * align;
      }
    }
    const int cnt = __builtin_amdgcn_readfirstlane((le - g0 < 64) ? (le - g0) : 64);
    SubRows<T, U> A, B;
    subrows_load<T, U, BINM>(A, 0, b_l, n_l, cf_l, midx, mval, lane);
    for (int u0 = 0; u0 < cnt; u0 += 2 * U) {
      if (u0 + U < cnt) subrows_load<T, U, BINM>(B, u0 + U, b_l, n_l, cf_l, midx, mval, lane);
      if constexpr (FIX) subrows_fold_fixed<U>(A, u0, b_l, n_l, cf_l, reinterpret_cast<unsigned*>(acc), midx, mval, lane);
      else subrows_fold<T, U, false>(A, u0, b_l, n_l, cf_l, acc, astride, dummy, midx, mval, lane);
      if (u0 + U < cnt) {
        if (u0 + 2 * U < cnt) subrows_load<T, U, BINM>(A, u0 + 2 * U, b_l, n_l, cf_l, midx, mval, lane);
        if constexpr (FIX) subrows_fold_fixed<U>(B, u0 + U, b_l, n_l, cf_l, reinterpret_cast<unsigned*>(acc), midx, mval, lane);
        else subrows_fold<T, U, false>(B, u0 + U, b_l, n_l, cf_l, acc, astride, dummy, midx, mval, lane);
      }
    }
  }
  // the wave's own LDS operations execute in order: no barrier needed before it reads its sums back
  T* orow = p.out + r * p.ld + j0;
  if constexpr (FIX) {
    const int* iacc = reinterpret_cast<const int*>(acc);
    for (int j = lane; j < jn; j += 64) orow[j] = (T)((float)iacc[j] * unscale) * p.inv2[j0 + j];
  } else {
    for (int j = lane; j < jn; j += 64) orow[j] = acc[j] * p.inv2[j0 + j];
  }
}

// ---------------------------------------------------------------- stage 1, flat stream of quads (round 3, measured variant)
// Workgroup as in transfer_block_kernel (NW waves = NW rows of L on chunk c; the chunk's sub-row offsets AND exact
// sub-row lengths staged in LDS; coefficients precomputed).  The entries of the 64 sub-rows of a feature group are walked
// as ONE stream of QUADS (4 consecutive entries of a sub-row, the last quad of a sub-row padded with zero entries) in
// steps of 64 quads = 256 entries ~ four 62-entry sub-rows: lane l of step t takes stream position 64 t + l, whichever
// sub-row it falls into -- no half-empty second halves.  One global_load_dwordx4 (values) + one global_load_dwordx2
// (indices) per step; a step holds entries of several features, whose columns can coincide: only safe because the sums
// are integer atomics (ds_add_u32 on 2^k-scaled fixed-point products, see subrows_fold_fixed).  Per step: the owner of
// the first lane (popcount of a ballot), then one select per sub-row boundary inside the step, all from wave-uniform
// broadcasts.  Steps run through a ring of D slots, every load unconditional.  Measured (C2): 2.04 ms -- the boundary
// loop's readlane -> scalar -> select chains make a step ~1900 clk per wave.
template <class T, int D, bool BINM>
__global__ void __launch_bounds__(1024) transfer_qflat_kernel(TransferArgs<T> p, const T* __restrict__ coef, int64_t nrows,
                                                             int align, int offs_bytes, int lens_bytes, float xmax,
                                                             const unsigned short* __restrict__ glen) {
  static_assert(std::is_same<T, float>::value, "fixed-point sums: fp32 graphs only");
  extern __shared__ __align__(16) unsigned char smem_raw[];
  int* offs = reinterpret_cast<int*>(smem_raw);                                        // [rows + 1], units of `align`
  unsigned short* lens = reinterpret_cast<unsigned short*>(smem_raw + offs_bytes);     // [rows] exact entries
  const int lane = threadIdx.x & 63, nw = blockDim.x >> 6;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int astride = p.SC + 64;
  unsigned* acc = reinterpret_cast<unsigned*>(smem_raw + offs_bytes + lens_bytes) + (size_t)wave * astride;
  const int c = blockIdx.x % p.nchunks;
  const int64_t r = (int64_t)(blockIdx.x / p.nchunks) * nw + wave;
  const ChunkedView<T> M = p.M[0];
  const CsrView<T> L = p.L[0];
  {
    const int* __restrict__ off = M.off + (int64_t)c * M.rows;
    const unsigned short* __restrict__ gl = glen + (int64_t)c * M.rows;
    for (int64_t i = threadIdx.x; i <= M.rows; i += blockDim.x) offs[i] = off[i];
    for (int64_t i = threadIdx.x; i < M.rows; i += blockDim.x) lens[i] = gl[i];
  }
  for (int j = lane; j < astride; j += 64) acc[j] = 0u;
  __syncthreads();
  if (r >= nrows) return;
  const int64_t gr = p.row_begin + r;
  const int64_t j0 = (int64_t)c * p.SC;
  const int jn = (int)((p.nj - j0 < p.SC) ? (p.nj - j0) : p.SC);
  const unsigned short* __restrict__ midx = M.idx;
  const T* __restrict__ mval = M.val;
  const int lb = __builtin_amdgcn_readfirstlane(L.ptr[gr]), le = __builtin_amdgcn_readfirstlane(L.ptr[gr + 1]);
  // scale 2^k of the row's sums: bound = (sum |coef|) * xmax < 2^e, k = 30 - e (see transfer_block_kernel)
  float scale, unscale;
  {
    float sabs = 0.f;
    for (int q = lb + lane; q < le; q += 64) sabs += fabsf(coef[q]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sabs += __shfl_xor(sabs, o);
    int e = 0;
    (void)frexpf(sabs * xmax, &e);
    if (!(sabs * xmax > 0.f)) e = 0;
    e = e < -90 ? -90 : (e > 120 ? 120 : e);
    scale = ldexpf(1.f, 30 - e);
    unscale = ldexpf(1.f, e - 30);
  }
  const int ngroups = (le - lb + 63) >> 6;
  auto raw_load = [&](int g, int& a, float& cf) __attribute__((always_inline)) {
    const int q = lb + g * 64 + lane;
    a = 0;
    cf = 0.f;
    if (q < le) { a = L.idx[q]; cf = coef[q]; }
  };
  // sub-row of feature a in this chunk: start (entries) and exact length, from LDS; nothing for a zero coefficient
  auto meta = [&](int a, float cf, int& b_l, int& n_l, float& cf_l) __attribute__((always_inline)) {
    cf_l = cf * scale;   // exact: a power of two
    b_l = 0;
    n_l = 0;
    if (cf_l != 0.f) {
      b_l = (offs[a] * align) >> 2;        // start in quads (sub-rows start on 32-entry boundaries)
      n_l = ((int)lens[a] + 3) >> 2;       // length in quads; the last quad is padded with zero entries
    }
  };
  // current group: per lane i the stream start P_i of its sub-row, delta_i = start in memory - P_i, coefficient
  int Pc = 0, dc = 0, ra, bn, nn;
  float cc = 0.f, cn, rc;
  int total = 0, nst = 0, t = 0, gcur = -1;
  raw_load(0, ra, rc); meta(ra, rc, bn, nn, cn);
  raw_load(1, ra, rc);
  bool done = false;
  auto switch_group = [&]() __attribute__((always_inline)) {
    // the prepared next group becomes current: exclusive prefix sum of the lengths over the 64 lanes
    ++gcur;
    int incl = nn;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int up = __shfl_up(incl, o);
      if (lane >= o) incl += up;
    }
    Pc = incl - nn;
    dc = bn - Pc;
    cc = cn;
    total = __builtin_amdgcn_readlane(incl, 63);
    nst = (total + 63) >> 6;
    t = 0;
    meta(ra, rc, bn, nn, cn);            // group gcur + 1 (its raw pairs were requested a group ago)
    raw_load(gcur + 2, ra, rc);
  };
  typedef float f4 __attribute__((ext_vector_type(4)));
  typedef unsigned u2 __attribute__((ext_vector_type(2)));
  u2 sj[D];
  f4 sv[D];
  float sc[D];
  const f4* __restrict__ mval4 = reinterpret_cast<const f4*>(mval);
  const u2* __restrict__ midx4 = reinterpret_cast<const u2*>(midx);
  auto next_step = [&](u2& oj, f4& ov, float& ocf) __attribute__((always_inline)) {
    if (t >= nst) {
      if (gcur + 1 < ngroups) switch_group();   // (may land on a group without entries: a null step, next call moves on)
      else done = true;
    }
    float cfv = 0.f;
    int addr = lane;   // lanes without work add 0 to different columns (same-address atomics serialise)
    if (t < nst) {
      const int lo = t << 6;
      const int pos = lo + lane;
      const unsigned long long le_mask = __ballot(Pc <= lo);
      const int own0 = __builtin_popcountll(le_mask) - 1;                 // last sub-row that starts at or before lo
      unsigned long long m = __ballot(Pc > lo && Pc <= lo + 63);          // sub-rows that start inside the step
      cfv = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(cc), own0));
      int dlv = __builtin_amdgcn_readlane(dc, own0);
      while (m != 0ull) {
        const int i = __builtin_ctzll(m);
        m &= m - 1ull;
        const int Pi = __builtin_amdgcn_readlane(Pc, i);
        const float cfi = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(cc), i));
        const int dli = __builtin_amdgcn_readlane(dc, i);
        const bool in = pos >= Pi;
        cfv = in ? cfi : cfv;
        dlv = in ? dli : dlv;
      }
      const bool valid = pos < total;
      cfv = valid ? cfv : 0.f;
      addr = valid ? pos + dlv : lane;
      ++t;
    }
    ocf = cfv;
    oj = midx4[addr];
    if constexpr (BINM) ov = f4{1.f, 1.f, 1.f, 1.f};
    else ov = mval4[addr];
  };
  auto fold = [&](u2 j, f4 v, float cf) __attribute__((always_inline)) {
    atomicAdd(acc + (j.x & 0xffffu), (unsigned)cvt_rpi(cf * v.x));
    atomicAdd(acc + (j.x >> 16), (unsigned)cvt_rpi(cf * v.y));
    atomicAdd(acc + (j.y & 0xffffu), (unsigned)cvt_rpi(cf * v.z));
    atomicAdd(acc + (j.y >> 16), (unsigned)cvt_rpi(cf * v.w));
  };
#pragma unroll
  for (int i = 0; i < D; ++i) next_step(sj[i], sv[i], sc[i]);
  while (!done) {
#pragma unroll
    for (int i = 0; i < D; ++i) {
      fold(sj[i], sv[i], sc[i]);
      next_step(sj[i], sv[i], sc[i]);
    }
  }
#pragma unroll
  for (int i = 0; i < D; ++i) fold(sj[i], sv[i], sc[i]);
  T* orow = p.out + r * p.ld + j0;
  const int* iacc = reinterpret_cast<const int*>(acc);
  for (int j = lane; j < jn; j += 64) orow[j] = ((float)iacc[j] * unscale) * p.inv2[j0 + j];
}

// ---------------------------------------------------------------- stage 1, wide loads + fixed-point sums (round 3)
// What the counters of the single-wave kernel say (profiles/r03_stage1_mem_pmc.txt): its 2- and 4-byte-per-lane loads
// cost the vector L1 ~6 accesses each (TCP_TOTAL_CACHE_ACCESSES ~0.8 per cycle and CU, TA busy 92 %), its LDS
// read-add-write pairs keep the LDS 64 % busy, and a wave needs ~400 clk per sub-row (readlane -> scalar -> address
// chains, one LDS round trip per fold).  This kernel removes all three:
//  * 16 bytes per lane: 16 lanes share a sub-row, lane e takes its entries 4e .. 4e+3 -- ONE global_load_dwordx4
//    (values) + ONE global_load_dwordx2 (indices) fetch the first 64 entries of FOUR sub-rows;
//  * sums are 2^k-scaled integers added with ds_add_u32 (half the LDS cycles of the pair, no return value, no wait;
//    columns of different sub-rows may coincide inside one instruction, which only atomics get right; integer sums do
//    not depend on any order);
//  * no per-sub-row scalar work: a wave writes (start, length, coefficient) of its 64 current features to a small LDS
//    table once per group and every lane reads its sub-row's triple from there (one ds_read per step).
// Entries past the first 64 of a sub-row (40 % of the sub-rows have some, ~8 on average) are taken afterwards, 4 lanes
// (16 entries) per sub-row and step, through a list of the features that still have entries left.
// Workgroup = NW waves = NW rows of L on chunk c; sub-row offsets and exact lengths of the chunk staged in LDS;
// coefficients precomputed (transfer_coef_kernel); steps run through a ring of D slots, every load unconditional.
constexpr int WIDE_META = 64 * 12;   // bytes per wave: (start quad, length, coefficient) of 64 features
constexpr int WIDE_LIST = 64;        // bytes per wave: features with entries left (one byte each)
template <int D, bool BINM>
__global__ void __launch_bounds__(1024) transfer_wide_kernel(TransferArgs<float> p, const float* __restrict__ coef,
                                                             int64_t nrows, int align, int offs_bytes, int lens_bytes,
                                                             float xmax, const unsigned short* __restrict__ glen) {
  extern __shared__ __align__(16) unsigned char smem_raw[];
  int* offs = reinterpret_cast<int*>(smem_raw);
  unsigned short* lens = reinterpret_cast<unsigned short*>(smem_raw + offs_bytes);
  unsigned char* ident = smem_raw + offs_bytes + lens_bytes;                 // [64] identity list (first 64 entries: all features)
  const int lane = threadIdx.x & 63, nw = blockDim.x >> 6;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int accb = (((p.SC + 4) * 4 + 15) / 16) * 16;   // + the padding column SC (zero entries point at it)
  unsigned char* mine = smem_raw + offs_bytes + lens_bytes + 64 + (size_t)wave * (size_t)(accb + WIDE_META + WIDE_LIST);
  unsigned* acc = reinterpret_cast<unsigned*>(mine);
  unsigned char* metab = mine + accb;                                        // [64][12]
  unsigned char* list = metab + WIDE_META;                                   // [64]
  const int c = blockIdx.x % p.nchunks;
  const int64_t r = (int64_t)(blockIdx.x / p.nchunks) * nw + wave;
  const ChunkedView<float> M = p.M[0];
  const CsrView<float> L = p.L[0];
  {
    const int* __restrict__ off = M.off + (int64_t)c * M.rows;
    const unsigned short* __restrict__ gl = glen + (int64_t)c * M.rows;
    for (int64_t i = threadIdx.x; i <= M.rows; i += blockDim.x) offs[i] = off[i];
    for (int64_t i = threadIdx.x; i < M.rows; i += blockDim.x) lens[i] = gl[i];
    if (threadIdx.x < 64) ident[threadIdx.x] = (unsigned char)threadIdx.x;
  }
  for (int j = lane; j < p.SC; j += 64) acc[j] = 0u;
  __syncthreads();
  if (r >= nrows) return;
  const int64_t gr = p.row_begin + r;
  const int64_t j0 = (int64_t)c * p.SC;
  const int jn = (int)((p.nj - j0 < p.SC) ? (p.nj - j0) : p.SC);
  const int lb = __builtin_amdgcn_readfirstlane(L.ptr[gr]), le = __builtin_amdgcn_readfirstlane(L.ptr[gr + 1]);
  float scale, unscale;
  {
    float sabs = 0.f;
    for (int q = lb + lane; q < le; q += 64) sabs += fabsf(coef[q]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sabs += __shfl_xor(sabs, o);
    int e = 0;
    (void)frexpf(sabs * xmax, &e);
    if (!(sabs * xmax > 0.f)) e = 0;
    e = e < -90 ? -90 : (e > 120 ? 120 : e);
    scale = ldexpf(1.f, 30 - e);
    unscale = ldexpf(1.f, e - 30);
  }
  const int ngroups = (le - lb + 63) >> 6;
  typedef float f4 __attribute__((ext_vector_type(4)));
  typedef unsigned u2 __attribute__((ext_vector_type(2)));
  const f4* __restrict__ mval4 = reinterpret_cast<const f4*>(M.val);
  const u2* __restrict__ midx4 = reinterpret_cast<const u2*>(M.idx);
  auto raw_load = [&](int g, int& a, float& cf) __attribute__((always_inline)) {
    const int q = lb + g * 64 + lane;
    a = 0;
    cf = 0.f;
    if (q < le) { a = L.idx[q]; cf = coef[q]; }
  };
  // ---- generator state (wave-uniform): group, level (0: entries 0..63, 16 lanes per sub-row; h >= 1: entries
  // 64 + 16 (h-1) .. +15, 4 lanes per sub-row), step inside the level
  int gcur = -1, level = 0, st = 0, nst = 0, nlist = 0, qoff = 0;
  int ncur = 0;            // lane i: length of the sub-row of feature i of the current group
  int ra;
  float rc;
  raw_load(0, ra, rc);
  bool done = false;
  auto open_group = [&]() __attribute__((always_inline)) {
    // the raw (feature, coefficient) pairs of this group are here; those of the next group are requested now
    ++gcur;
    const float cf = rc * scale;   // exact: a power of two
    int b = 0, n = 0;
    if (cf != 0.f) { b = (offs[ra] * align) >> 2; n = (int)lens[ra]; }
    raw_load(gcur + 1, ra, rc);
    ncur = n;
    unsigned* m = reinterpret_cast<unsigned*>(metab + lane * 12);
    m[0] = (unsigned)b; m[1] = (unsigned)n; m[2] = __float_as_uint(cf);
    level = 0;
    st = 0;
    qoff = 0;
    const int cnt = (le - lb - gcur * 64 < 64) ? (le - lb - gcur * 64) : 64;
    nlist = cnt;
    nst = (cnt + 3) >> 2;          // four sub-rows per step
  };
  auto next_level = [&]() __attribute__((always_inline)) {
    // features that still have entries: level 1 starts at entry 64, each further level 16 entries on
    const int covered = 64 + 16 * level;
    ++level;
    qoff = covered >> 2;
    const bool more = ncur > covered;
    const unsigned long long mk = __ballot(more);
    nlist = __builtin_popcountll(mk);
    if (more) list[__builtin_amdgcn_mbcnt_hi((unsigned)(mk >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mk, 0u))] = (unsigned char)lane;
    st = 0;
    nst = (nlist + 15) >> 4;       // sixteen sub-rows per step
  };
  u2 sj[D];
  f4 sv[D];
  float sc[D];
  auto next_step = [&](u2& oj, f4& ov, float& ocf) __attribute__((always_inline)) {
    if (st >= nst) {
      if (gcur >= 0 && nlist > 0) next_level();          // (an empty level: nst = 0, handled by the next call)
      else if (gcur + 1 < ngroups) open_group();
      else done = true;
    }
    float cfv = 0.f;
    int quad = lane;   // lanes without work add 0 to 64 DIFFERENT columns (64 lanes adding to one LDS word serialise)
    int rem = 0;       // entries of the sub-row in this lane's quad (the last quad of a sub-row ends in padding)
    if (st < nst) {
      const int sh = level == 0 ? 4 : 2;                 // log2(lanes per sub-row)
      const int pos = (st << (6 - sh)) + (lane >> sh);   // position in the list of features of this level
      const int e = lane & ((1 << sh) - 1);
      const unsigned char* lst = level == 0 ? ident : list;
      const int fi = pos < nlist ? (int)lst[pos] : 0;
      const unsigned* m = reinterpret_cast<const unsigned*>(metab + fi * 12);
      const int b = (int)m[0], n = (int)m[1];
      const float cf = __uint_as_float(m[2]);
      const int qe = qoff + e;                           // quad of the sub-row this lane takes
      const bool valid = pos < nlist && 4 * qe < n;
      cfv = valid ? cf : 0.f;
      quad = valid ? b + qe : lane;
      rem = n - 4 * qe;
      ++st;
    }
    ocf = cfv;
    oj = midx4[quad];
    // pattern-only operand: the values are not read, so the padding of a sub-row's last quad must be masked here (a
    // stored padding value is 0, an implied one would be 1 -- added to the padding's column SC)
    if constexpr (BINM) ov = f4{rem > 0 ? 1.f : 0.f, rem > 1 ? 1.f : 0.f, rem > 2 ? 1.f : 0.f, rem > 3 ? 1.f : 0.f};
    else ov = mval4[quad];
  };
  auto fold = [&](u2 j, f4 v, float cf) __attribute__((always_inline)) {
    atomicAdd(acc + (j.x & 0xffffu), (unsigned)cvt_rpi(cf * v.x));
    atomicAdd(acc + (j.x >> 16), (unsigned)cvt_rpi(cf * v.y));
    atomicAdd(acc + (j.y & 0xffffu), (unsigned)cvt_rpi(cf * v.z));
    atomicAdd(acc + (j.y >> 16), (unsigned)cvt_rpi(cf * v.w));
  };
#pragma unroll
  for (int i = 0; i < D; ++i) next_step(sj[i], sv[i], sc[i]);
  while (!done) {
#pragma unroll
    for (int i = 0; i < D; ++i) {
      fold(sj[i], sv[i], sc[i]);
      next_step(sj[i], sv[i], sc[i]);
    }
  }
#pragma unroll
  for (int i = 0; i < D; ++i) fold(sj[i], sv[i], sc[i]);
  float* orow = p.out + r * p.ld + j0;
  const int* iacc = reinterpret_cast<const int*>(acc);
  for (int j = lane; j < jn; j += 64) orow[j] = ((float)iacc[j] * unscale) * p.inv2[j0 + j];
}

// ---------------------------------------------------------------- stage 1, wide loads, static schedule (round 3)
// transfer_wide_kernel without its step generator (whose scalar control flow cost ~45 instructions and two dependent LDS
// reads per step: 1.2 ms of the 1.8 ms with loads and atomics removed).  A group of 64 features is ALWAYS 20 steps:
//   steps  0..15  entries  0..63 of features 4s .. 4s+3        (16 lanes x 4 entries per sub-row)
//   steps 16..18  entries 64..79 of the features that have them (4 lanes per sub-row, 16 features per step, from list 1)
//   step  19      entries 80..95 of the features that have them (list 2)
// so the whole row is one straight-line software pipeline: the loop body is one group, completely unrolled, LDS table
// reads use immediate offsets, a ring of 4 slots carries the loads across steps and across groups (the table of the
// next group is written -- into the other of two buffers -- four steps before the current group ends), every load is
// unconditional and the waits are exact counts.  What the 20 steps cannot hold (a 49th feature longer than 64 entries,
// a 17th longer than 80, entries past 96; ~1e-4 of the sub-rows at a mean of 62) is added by the owning lane itself,
// entry by entry, when the table is written.
constexpr int W2_TBL = 64 * 12;
constexpr int W2_STEPS = 20;
constexpr int W2_D = 4;
template <bool BINM>
__global__ void __launch_bounds__(1024) transfer_wide2_kernel(TransferArgs<float> p, const float* __restrict__ coef,
                                                              int64_t nrows, int align, int offs_bytes, int lens_bytes,
                                                              float xmax, const unsigned short* __restrict__ glen) {
  extern __shared__ __align__(16) unsigned char smem_raw[];
  int* offs = reinterpret_cast<int*>(smem_raw);
  unsigned short* lens = reinterpret_cast<unsigned short*>(smem_raw + offs_bytes);
  const int lane = threadIdx.x & 63, nw = blockDim.x >> 6;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int accb = (((p.SC + 4) * 4 + 15) / 16) * 16;   // + the padding column SC (zero entries point at it)
  const int per_wave = accb + 2 * W2_TBL + 4 * 64;
  unsigned char* mine = smem_raw + offs_bytes + lens_bytes + (size_t)wave * (size_t)per_wave;
  unsigned* acc = reinterpret_cast<unsigned*>(mine);
  unsigned char* tbl0 = mine + accb;                      // two tables [64][12]: (start quad, length, coefficient)
  unsigned char* lst0 = tbl0 + 2 * W2_TBL;                // two x (list 1 [64], list 2 [64])
  const int c = blockIdx.x % p.nchunks;
  const int64_t r = (int64_t)(blockIdx.x / p.nchunks) * nw + wave;
  const ChunkedView<float> M = p.M[0];
  const CsrView<float> L = p.L[0];
  {
    const int* __restrict__ off = M.off + (int64_t)c * M.rows;
    const unsigned short* __restrict__ gl = glen + (int64_t)c * M.rows;
    for (int64_t i = threadIdx.x; i <= M.rows; i += blockDim.x) offs[i] = off[i];
    for (int64_t i = threadIdx.x; i < M.rows; i += blockDim.x) lens[i] = gl[i];
  }
  for (int j = lane; j < p.SC; j += 64) acc[j] = 0u;
  __syncthreads();
  if (r >= nrows) return;
  const int64_t gr = p.row_begin + r;
  const int64_t j0 = (int64_t)c * p.SC;
  const int jn = (int)((p.nj - j0 < p.SC) ? (p.nj - j0) : p.SC);
  const int lb = __builtin_amdgcn_readfirstlane(L.ptr[gr]), le = __builtin_amdgcn_readfirstlane(L.ptr[gr + 1]);
  float scale, unscale;
  {
    float sabs = 0.f;
    for (int q = lb + lane; q < le; q += 64) sabs += fabsf(coef[q]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sabs += __shfl_xor(sabs, o);
    int e = 0;
    (void)frexpf(sabs * xmax, &e);
    if (!(sabs * xmax > 0.f)) e = 0;
    e = e < -90 ? -90 : (e > 120 ? 120 : e);
    scale = ldexpf(1.f, 30 - e);
    unscale = ldexpf(1.f, e - 30);
  }
  const int ngroups = (le - lb + 63) >> 6;
  typedef float f4 __attribute__((ext_vector_type(4)));
  typedef unsigned u2 __attribute__((ext_vector_type(2)));
  const f4* __restrict__ mval4 = reinterpret_cast<const f4*>(M.val);
  const u2* __restrict__ midx4 = reinterpret_cast<const u2*>(M.idx);
  const unsigned short* __restrict__ midx = M.idx;
  const float* __restrict__ mval = M.val;
  const int lastq = le > lb ? le - 1 : lb;   // (an empty row never dereferences: ngroups == 0)
  // raw (feature, coefficient) pair of this lane in group g; lanes past the end of the row get coefficient 0.
  // Unconditional loads from a clamped position: no branch, exact wait counts.
  auto raw_load = [&](int g, int& a, float& cf) __attribute__((always_inline)) {
    const int q = lb + g * 64 + lane;
    const int qc = q < le ? q : lastq;
    a = L.idx[qc];
    const float x = coef[qc];
    cf = q < le ? x : 0.f;
  };
  // write the table + lists of group g (its raw pair is in (ra, rc)) into buffer g & 1; request the raw pair of g + 1
  int ra = 0;
  float rc = 0.f;
  auto open_group = [&](int g) __attribute__((always_inline)) {
    unsigned char* tb = tbl0 + (g & 1) * W2_TBL;
    unsigned char* l1 = lst0 + (g & 1) * 128;
    unsigned char* l2 = l1 + 64;
    const float cf = (g < ngroups) ? rc * scale : 0.f;   // exact: a power of two
    int b = 0, n = 0;
    if (cf != 0.f) { b = (offs[ra] * align) >> 2; n = (int)lens[ra]; }
    raw_load(g + 1, ra, rc);                            // (clamped past the end of the row: always safe, never a branch)
    unsigned* m = reinterpret_cast<unsigned*>(tb + lane * 12);
    m[0] = (unsigned)b; m[1] = (unsigned)n; m[2] = __float_as_uint(cf);
    // list 1 holds the first 48 features longer than 64 entries (steps 16..18), list 2 the first 16 of THOSE that are
    // longer than 80 (step 19): a feature that found no place in list 1 must not appear in list 2 -- its owner adds
    // everything past entry 64 itself, and step 19 would add entries 80..95 a second time
    const unsigned long long m1 = __ballot(n > 64);
    const int r1 = __builtin_amdgcn_mbcnt_hi((unsigned)(m1 >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m1, 0u));
    const bool in1 = n > 64 && r1 < 48;
    const unsigned long long m2 = __ballot(in1 && n > 80);
    const int r2 = __builtin_amdgcn_mbcnt_hi((unsigned)(m2 >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m2, 0u));
    const bool in2 = in1 && n > 80 && r2 < 16;
    l1[lane] = 255; l2[lane] = 255;                      // 255 = no feature at this list position
    if (in1) l1[r1] = (unsigned char)lane;
    if (in2) l2[r2] = (unsigned char)lane;
    // the rare rest: entries the 20 steps do not reach, added by the owning lane
    const int cov = in2 ? 96 : (in1 ? 80 : 64);
    if (__ballot(n > cov) != 0ull) {
      for (int x = cov; x < n; ++x) {
        const int at = 4 * b + x;
        atomicAdd(acc + midx[at], (unsigned)cvt_rpi(cf * (BINM ? 1.f : mval[at])));
      }
    }
  };
  // ---- the ring
  u2 sj[W2_D];
  f4 sv[W2_D];
  float sc[W2_D];
  auto fetch = [&](int quad, float cfv, int rem, u2& oj, f4& ov, float& ocf) __attribute__((always_inline)) {
    ocf = cfv;
    oj = midx4[quad];
    // pattern-only operand: the values are not read, so the padding of a sub-row's last quad is masked here
    if constexpr (BINM) ov = f4{rem > 0 ? 1.f : 0.f, rem > 1 ? 1.f : 0.f, rem > 2 ? 1.f : 0.f, rem > 3 ? 1.f : 0.f};
    else ov = mval4[quad];
  };
  // step s (compile-time after unrolling) of the group whose table sits in buffer `par`
  auto gen = [&](int s, int par, u2& oj, f4& ov, float& ocf) __attribute__((always_inline)) {
    const unsigned char* tb = tbl0 + par * W2_TBL;
    int fi, e, qoff;
    bool have = true;
    if (s < 16) {
      fi = 4 * s + (lane >> 4);
      e = lane & 15;
      qoff = 0;
    } else {
      const unsigned char* lst = lst0 + par * 128 + (s < 19 ? 0 : 64);
      const int pos = (s < 19 ? 16 * (s - 16) : 0) + (lane >> 2);
      fi = (int)lst[pos];
      have = fi != 255;
      fi = have ? fi : 0;
      e = lane & 3;
      qoff = s < 19 ? 16 : 20;
    }
    const unsigned* m = reinterpret_cast<const unsigned*>(tb + fi * 12);
    const int b = (int)m[0], n = (int)m[1];
    const float cf = __uint_as_float(m[2]);
    const int qe = qoff + e;
    const bool valid = have && 4 * qe < n;
    // lanes without work add 0 to 64 DIFFERENT columns (many lanes adding to one LDS word serialise)
    fetch(valid ? b + qe : lane, valid ? cf : 0.f, n - 4 * qe, oj, ov, ocf);
  };
  auto fold = [&](u2 j, f4 v, float cf) __attribute__((always_inline)) {
    atomicAdd(acc + (j.x & 0xffffu), (unsigned)cvt_rpi(cf * v.x));
    atomicAdd(acc + (j.x >> 16), (unsigned)cvt_rpi(cf * v.y));
    atomicAdd(acc + (j.y & 0xffffu), (unsigned)cvt_rpi(cf * v.z));
    atomicAdd(acc + (j.y >> 16), (unsigned)cvt_rpi(cf * v.w));
  };
  if (ngroups > 0) {
    raw_load(0, ra, rc);
    open_group(0);
#pragma unroll
    for (int i = 0; i < W2_D; ++i) gen(i, 0, sj[i], sv[i], sc[i]);
    for (int g = 0; g < ngroups; ++g) {
      const int par = g & 1;
#pragma unroll
      for (int s = 0; s < W2_STEPS; ++s) {
        if (s == W2_STEPS - W2_D) open_group(g + 1);     // (past the last group: an empty table, all steps idle)
        fold(sj[s % W2_D], sv[s % W2_D], sc[s % W2_D]);
        if (s + W2_D < W2_STEPS) gen(s + W2_D, par, sj[s % W2_D], sv[s % W2_D], sc[s % W2_D]);
        else gen(s + W2_D - W2_STEPS, par ^ 1, sj[s % W2_D], sv[s % W2_D], sc[s % W2_D]);
      }
    }
    // the ring now holds the first steps of the (empty) group past the end: nothing left to add
  }
  float* orow = p.out + r * p.ld + j0;
  const int* iacc = reinterpret_cast<const int*>(acc);
  for (int j = lane; j < jn; j += 64) orow[j] = ((float)iacc[j] * unscale) * p.inv2[j0 + j];
}

// waves per workgroup of the block kernel that fit next to the offsets: 0 = does not fit (fall back)
template <class T>
static int transfer_block_waves(int64_t mrows, int SC, int* offs_bytes) {
  const int64_t ob = (((mrows + 1) * 4 + 15) / 16) * 16;
  const int64_t per_wave = (int64_t)(SC + 64) * (int64_t)sizeof(T);
  int64_t nw = ((int64_t)ctx().lds_per_block - ob) / per_wave;
  if (ob >= (int64_t)ctx().lds_per_block) nw = 0;
  if (nw > 16) nw = 16;
  if (const char* e = getenv("SS_TRANSFER_NW")) {
    const int v = atoi(e);
    if (v >= 1 && v < nw) nw = v;
  }
  *offs_bytes = (int)ob;
  return nw >= 8 ? (int)nw : 0;
}

template <class T>
int launch_transfer_block(const DevCsr<T>& L, const T* inv1, const DevChunked<T>& Mt, const T* inv2, int64_t row_begin,
                          int64_t nrows, int64_t nj, T* out, int64_t ld, T* coef, float xmax, bool fixed) {
  if (nrows <= 0 || nj <= 0) return SS_OK;
  int offs_bytes = 0;
  const int nw = transfer_block_waves<T>(Mt.rows, Mt.SC, &offs_bytes);
  if (nw == 0) return fail(SS_EUNSUPPORTED, "transfer_block: offsets + accumulators do not fit in LDS");
  TransferArgs<T> p{};
  p.nterms = 1;
  p.L[0] = view(L);
  p.M[0] = view(Mt);
  p.inv2 = inv2;
  p.row_begin = row_begin;
  p.nj = nj;
  p.SC = Mt.SC;
  p.nchunks = Mt.nchunks;
  p.out = out;
  p.ld = ld;
  const int64_t groups = ceil_div(nrows, (int64_t)nw);
  const int64_t grid = groups * p.nchunks;
  if (grid >= (1LL << 31)) return fail(SS_EUNSUPPORTED, "transfer grid too large; lower SS_TRANSFER_BYTES");
  path_add("transfer_block");
  hipLaunchKernelGGL(transfer_coef_kernel<T>, dim3(grid_1d(L.nnz > 0 ? L.nnz : 1, 256, 256 * 8)), dim3(256), 0, ctx().stream,
                     L.ptr.p, L.idx.p, L.val.p, inv1, row_begin, nrows, coef);
  SS_LAUNCH_CHECK();
  const size_t lds = (size_t)offs_bytes + (size_t)nw * (size_t)(p.SC + 64) * sizeof(T);
  constexpr bool CANFIX = std::is_same<T, float>::value;
  const bool fx = CANFIX && fixed;
  if (fx) path_add("fixed_point");
  if constexpr (std::is_same<T, float>::value) {
    // the fixed-point kernels with 16 bytes per lane: they walk exact sub-row lengths (operand cut with align 32)
    auto envi = [](const char* k) { const char* e = getenv(k); return e ? atoi(e) : 0; };
    const int wide2 = envi("SS_TRANSFER_WIDE2"), wide = envi("SS_TRANSFER_WIDE"), qflat = envi("SS_TRANSFER_QFLAT");
    if ((wide2 > 0 || wide > 0 || qflat > 0) && fixed && Mt.align == 32 && Mt.len_ok) {
      const int64_t lens_bytes = ((Mt.rows * 2 + 15) / 16) * 16;
      const int64_t accb = ((((int64_t)p.SC + 4) * 4 + 15) / 16) * 16;
      const int64_t per_wave = wide2 > 0 ? accb + 2 * W2_TBL + 4 * 64
                               : wide > 0 ? accb + WIDE_META + WIDE_LIST : (int64_t)(p.SC + 64) * 4;
      const int64_t fixed_bytes = (int64_t)offs_bytes + lens_bytes + (wide > 0 && wide2 == 0 ? 64 : 0);
      int64_t nwf = ((int64_t)ctx().lds_per_block - fixed_bytes) / per_wave;
      if (nwf > 16) nwf = 16;
      if (const int v = envi("SS_TRANSFER_NW"); v >= 1 && v < nwf) nwf = v;
      if (nwf >= 8) {
        const int64_t fgrid = ceil_div(nrows, nwf) * p.nchunks;
        const size_t flds = (size_t)fixed_bytes + (size_t)nwf * (size_t)per_wave;
        const unsigned short* glen = Mt.len.p;
#define SS_TX_LAUNCH(KERNEL)                                                                                          \
  do {                                                                                                                \
    static std::atomic<bool> attr_done{false};                                                                        \
    if (!attr_done) {                                                                                                 \
      SS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&KERNEL), hipFuncAttributeMaxDynamicSharedMemorySize,  \
                                 160 * 1024));                                                                        \
      attr_done = true;                                                                                               \
    }                                                                                                                 \
    hipLaunchKernelGGL(KERNEL, dim3((unsigned)fgrid), dim3(64 * (int)nwf), flds, ctx().stream, p, (const float*)coef,  \
                       nrows, Mt.align, offs_bytes, (int)lens_bytes, xmax, glen);                                     \
  } while (0)
        if (wide2 > 0) {
          path_add("wide2");
          if (Mt.binary) SS_TX_LAUNCH((transfer_wide2_kernel<true>)); else SS_TX_LAUNCH((transfer_wide2_kernel<false>));
        } else if (wide > 0) {
          path_add("wide");
          if (wide == 2) { if (Mt.binary) SS_TX_LAUNCH((transfer_wide_kernel<2, true>)); else SS_TX_LAUNCH((transfer_wide_kernel<2, false>)); }
          else { if (Mt.binary) SS_TX_LAUNCH((transfer_wide_kernel<4, true>)); else SS_TX_LAUNCH((transfer_wide_kernel<4, false>)); }
        } else {
          path_add("qflat");
          if (Mt.binary) SS_TX_LAUNCH((transfer_qflat_kernel<float, 4, true>)); else SS_TX_LAUNCH((transfer_qflat_kernel<float, 4, false>));
        }
#undef SS_TX_LAUNCH
        SS_LAUNCH_CHECK();
        return SS_OK;
      }
    }
  }
  const int pu = transfer_u() == 4 ? 4 : 8;
#define SS_TB_LAUNCH(U, BINM, FIX)                                                                                    \
  do {                                                                                                                \
    static std::atomic<bool> attr_done{false};                                                                        \
    if (!attr_done) {                                                                                                 \
      SS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&transfer_block_kernel<T, U, BINM, FIX>),              \
                                 hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));                            \
      attr_done = true;                                                                                               \
    }                                                                                                                 \
    hipLaunchKernelGGL((transfer_block_kernel<T, U, BINM, FIX>), dim3((unsigned)grid), dim3(64 * nw), lds,            \
                       ctx().stream, p, (const T*)coef, nrows, Mt.align, offs_bytes, xmax);                           \
  } while (0)
#define SS_TB_BIN(U, FIX) do { if (Mt.binary) SS_TB_LAUNCH(U, true, FIX); else SS_TB_LAUNCH(U, false, FIX); } while (0)
#define SS_TB_FIX(U) do { if (fx) SS_TB_BIN(U, CANFIX); else SS_TB_BIN(U, false); } while (0)
  if (pu == 4) SS_TB_FIX(4);
  else SS_TB_FIX(8);
#undef SS_TB_FIX
#undef SS_TB_BIN
#undef SS_TB_LAUNCH
  SS_LAUNCH_CHECK();
  return SS_OK;
}

template <class T>
bool transfer_block_fits(int64_t mrows, int SC) {
  int ob = 0;
  return transfer_block_waves<T>(mrows, SC, &ob) > 0;
}

template <class T, bool LOO>
static int launch_transfer_variant(const TransferArgs<T>& p, unsigned grid, size_t lds, bool binm, bool dual) {
#define SS_TRANSFER_LAUNCH(U, BINM, DUAL)                                                               \
  hipLaunchKernelGGL((transfer_kernel<T, LOO, U, BINM, DUAL>), dim3(grid), dim3(TRANSFER_THREADS), lds, \
                     ctx().stream, p)
#define SS_TRANSFER_U(U)                                                                      \
  do {                                                                                        \
    if (binm) { if (dual) SS_TRANSFER_LAUNCH(U, true, true); else SS_TRANSFER_LAUNCH(U, true, false); } \
    else { if (dual) SS_TRANSFER_LAUNCH(U, false, true); else SS_TRANSFER_LAUNCH(U, false, false); }    \
  } while (0)
  if constexpr (std::is_same<T, float>::value && !LOO) {
    const char* eb = getenv("SS_TRANSFER_LD");
    if (eb && atoi(eb) == 1 && !dual) {
      path_add("buffer_loads");
      const bool fx = getenv("SS_TRANSFER_FIX1") && atoi(getenv("SS_TRANSFER_FIX1")) == 1 && p.nterms == 1 && !p.accumulate;
      if (fx) path_add("fixed_point");
#define SS_TL(U, BINM, FX) hipLaunchKernelGGL((transfer_kernel<T, LOO, U, BINM, false, FX, true>), dim3(grid), dim3(TRANSFER_THREADS), lds, ctx().stream, p)
      if (transfer_u(p.small_weighted != 0 && !LOO) == 4) {
        if (binm) { if (fx) SS_TL(4, true, true); else SS_TL(4, true, false); }
        else { if (fx) SS_TL(4, false, true); else SS_TL(4, false, false); }
      } else {
        if (binm) { if (fx) SS_TL(8, true, true); else SS_TL(8, true, false); }
        else { if (fx) SS_TL(8, false, true); else SS_TL(8, false, false); }
      }
#undef SS_TL
      SS_LAUNCH_CHECK();
      return SS_OK;
    }
    const char* e = getenv("SS_TRANSFER_FIX1");
    if (e && atoi(e) == 1 && p.nterms == 1 && !dual && !p.accumulate) {
      path_add("fixed_point");
      if (transfer_u(p.small_weighted != 0 && !LOO) == 4) {
        if (binm) hipLaunchKernelGGL((transfer_kernel<T, LOO, 4, true, false, true>), dim3(grid), dim3(TRANSFER_THREADS), lds, ctx().stream, p);
        else hipLaunchKernelGGL((transfer_kernel<T, LOO, 4, false, false, true>), dim3(grid), dim3(TRANSFER_THREADS), lds, ctx().stream, p);
      } else {
        if (binm) hipLaunchKernelGGL((transfer_kernel<T, LOO, 8, true, false, true>), dim3(grid), dim3(TRANSFER_THREADS), lds, ctx().stream, p);
        else hipLaunchKernelGGL((transfer_kernel<T, LOO, 8, false, false, true>), dim3(grid), dim3(TRANSFER_THREADS), lds, ctx().stream, p);
      }
      SS_LAUNCH_CHECK();
      return SS_OK;
    }
  }
  switch (transfer_u(p.small_weighted != 0 && !LOO)) {
    case 4: SS_TRANSFER_U(4); break;
    case 16: SS_TRANSFER_U(16); break;
    default: SS_TRANSFER_U(8); break;
  }
#undef SS_TRANSFER_U
#undef SS_TRANSFER_LAUNCH
  SS_LAUNCH_CHECK();
  return SS_OK;
}

template <class T>
int launch_transfer(int nterms, const DevCsr<T>* L[2], const T* inv1[2], const DevChunked<T>* Mt[2],
                    const T* inv2, int64_t row_begin, int64_t nrows, int64_t nj, T* out, int64_t ld,
                    const int* row_ids, bool accumulate) {
  if (nrows <= 0 || nj <= 0) return SS_OK;
  TransferArgs<T> p{};
  p.nterms = nterms;
  for (int t = 0; t < nterms; ++t) {
    p.L[t] = view(*L[t]);
    p.M[t] = view(*Mt[t]);
    p.inv1[t] = inv1[t];
    if (Mt[t]->SC != Mt[0]->SC || Mt[t]->nchunks != Mt[0]->nchunks)
      return fail(SS_EINVAL, "transfer operands were cut with different chunk sizes");
  }
  p.inv2 = inv2;
  p.row_begin = row_begin;
  p.row_ids = row_ids;
  p.nj = nj;
  p.SC = Mt[0]->SC;
  p.nchunks = Mt[0]->nchunks;
  p.out = out;
  p.ld = ld;
  p.accumulate = accumulate ? 1 : 0;
  p.xmax = 1.0f;
  {
    int64_t bytes = 0;
    bool weighted = false;
    for (int t = 0; t < nterms; ++t) {
      bytes += Mt[t]->stored * (int64_t)(2 + sizeof(T));
      weighted = weighted || !Mt[t]->binary;
    }
    p.small_weighted = (weighted && bytes <= (32LL << 20)) ? 1 : 0;
  }
  p.chunk_interleave = (getenv("SS_TRANSFER_ORDER") && atoi(getenv("SS_TRANSFER_ORDER")) == 0) ? 1 : 0;
  const int64_t grid = nrows * p.nchunks;
  if (grid >= (1LL << 31)) return fail(SS_EUNSUPPORTED, "transfer grid too large; lower SS_TRANSFER_BYTES");
  bool binm = true;
  for (int t = 0; t < nterms; ++t) binm = binm && Mt[t]->binary;
  const bool dual = transfer_dual();
  path_add("transfer");
  const size_t lds = (size_t)(p.SC + 64) * sizeof(T) * (dual ? 2 : 1);
  SS_TRY((launch_transfer_variant<T, false>(p, (unsigned)grid, lds, binm, dual)));
  return SS_OK;
}

template <class T>
int launch_transfer_loo(const DevCsr<T>& X, const DevChunked<T>& XT, const int* kf, const int* ks,
                        int64_t i_begin, int64_t nrows, T* out, int64_t ld) {
  if (nrows <= 0) return SS_OK;
  TransferArgs<T> p{};
  p.nterms = 1;
  p.L[0] = view(X);
  p.M[0] = view(XT);
  p.kf = kf;
  p.ks = ks;
  p.row_begin = i_begin;
  p.nj = X.rows;
  p.SC = XT.SC;
  p.nchunks = XT.nchunks;
  p.out = out;
  p.ld = ld;
  p.chunk_interleave = (getenv("SS_TRANSFER_ORDER") && atoi(getenv("SS_TRANSFER_ORDER")) == 0) ? 1 : 0;
  const int64_t grid = nrows * p.nchunks;
  if (grid >= (1LL << 31)) return fail(SS_EUNSUPPORTED, "transfer grid too large; lower SS_TRANSFER_BYTES");
  const bool dual = transfer_dual();
  path_add("transfer_loo");
  const size_t lds = (size_t)(p.SC + 64) * sizeof(T) * (dual ? 2 : 1) + (size_t)((p.SC + 31) / 32) * 4;
  SS_TRY((launch_transfer_variant<T, true>(p, (unsigned)grid, lds, XT.binary, dual)));
  return SS_OK;
}

// ============================================================== stage 2, wide: SELL x LDS tile
template <class T, int QT>
struct alignas(sizeof(T) * QT) Vec {
  T v[QT];
};

template <class T>
struct SellArgs {
  const int* off;
  const unsigned short* idx;
  const T* val;
  int nslices, nchunks, KC;
  int64_t K, M, B;
  const T* R;
  int64_t ldr;
  T* F;
  int64_t ldf;
  const int* clean_deg;
  const int* out_rows;  // optional: column b of R goes to row out_rows[b] of F (k-fold members -> source rows)
};

constexpr int SELL_THREADS = 1024;
constexpr int SELL_NB = 4;  // quads of W in flight per lane ahead of the gathers
constexpr int SELL_QT_F32_DEFAULT = 4;
constexpr int SELL_ZERO_ROWS = 16;  // zero rows behind the tile: one per 16-byte slot class of the 256-byte LDS line
constexpr int SELL_STAGE_IT = 5;  // passes of SELL_THREADS per staging round (the largest tile, 10240 rows, takes two rounds)

// Workgroup = QT columns of R (QT queries).  Per chunk of KC columns of W: the tile
// R[b0..b0+QT)[k0..k0+KC) sits in LDS as [k][QT] so that one ds_read_b128 fetches the QT
// operands of a non-zero; waves walk the slices, lane = row of W, no cross-lane reduction.
// (16-bit half H of x) << sh in one VALU instruction (SDWA source select)
template <int H>
__device__ __forceinline__ unsigned half_shl(unsigned x, unsigned sh) {
  unsigned r;
  if (H == 0)
    asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0"
        : "=v"(r) : "v"(sh), "v"(x));
  else
    asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1"
        : "=v"(r) : "v"(sh), "v"(x));
  return r;
}

template <class T, int QT, bool BIN>
__global__ void __launch_bounds__(SELL_THREADS) spmm_sell_kernel(SellArgs<T> a) {
  // Tile rows of 16 bytes ((float,4), (double,2)), or of 32 bytes = two 16-byte pieces ((float,8), round 3): the index
  // stream of W, which every workgroup re-reads for its own columns, is then shared by twice as many columns.  It is what
  // bounds this kernel (measured by ablation: without it stage 2 takes 0.36 instead of 0.59 ms at C2, 7.4 instead of
  // 12.7 ms at C3; with half the LDS reads 0.55 / 12.1).  With two pieces a lane reads piece r ^ (lane & 1) in read r, so
  // that the 16 lanes of an LDS cycle are spread over 16 (slot class, piece) cells; the accumulators hold the columns in
  // that rotated order and are put back when they are stored.
  constexpr int RB = (int)sizeof(T) * QT, NP = RB / 16;
  // registers: two-piece rows stage fewer passes per round and (weighted) keep fewer quads in flight
  constexpr int STAGE_IT = NP == 2 ? 2 : SELL_STAGE_IT;
  constexpr int NBQ = NP == 2 ? 2 : SELL_NB;   // divides SELL_NB: the builder pads slices to groups of SELL_NB quads
  constexpr unsigned TSH = NP == 2 ? 5 : 4;  // log2(RB)
  static_assert(RB == 16 || (RB == 32 && sizeof(T) == 4), "tile row must be 16 bytes, or 32 bytes of fp32");
  extern __shared__ __align__(16) unsigned char smem_raw[];
  using V = Vec<T, QT>;
  // [KC + 16]: rows KC .. KC+15 stay zero -- one padding target per 16-byte slot class, so that the builder can point a
  // padding entry at a zero row whose slot no active lane of its LDS cycle uses (sell_fill_sched_kernel, assemble.hip)
  V* tile = reinterpret_cast<V*>(smem_raw);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nwaves = blockDim.x >> 6;
  const int64_t b0 = (int64_t)blockIdx.x * QT;
  const unsigned pid = NP == 2 ? (unsigned)(lane & 1) : 0u;   // which piece this lane reads first
  // (the gathers below use tile-relative LDS addresses: the dynamic LDS segment must start at LDS address 0, i.e. the
  // kernel must have no static __shared__ object -- launch_spmm_sell checks that on the host before the first launch)

  for (int c = 0; c < a.nchunks; ++c) {
    const int64_t k0 = (int64_t)c * a.KC;
    const int kn = (int)((a.K - k0 < a.KC) ? (a.K - k0) : a.KC);
    if (c) __syncthreads();
    // Stage the tile in rounds of SELL_STAGE_IT passes of 1024 threads: all loads of a round are requested before its
    // first LDS write (clamped, branch-free addresses) -- one memory latency per round (two for the largest tile)
    // instead of one per pass.  Nothing else runs on the CU meanwhile (one workgroup per CU): this latency is not hidden.
    for (int kb = 0; kb < a.KC + SELL_ZERO_ROWS; kb += STAGE_IT * SELL_THREADS) {
      T stg[STAGE_IT][QT];
      const int klast = kn > 0 ? kn - 1 : 0;
#pragma unroll
      for (int it = 0; it < STAGE_IT; ++it) {
        const int k = kb + tid + it * SELL_THREADS;
        const int kk = k < kn ? k : klast;
#pragma unroll
        for (int q = 0; q < QT; ++q) {
          const int64_t b = (b0 + q < a.B) ? b0 + q : a.B - 1;
          stg[it][q] = kn > 0 ? a.R[b * a.ldr + k0 + kk] : T(0);
        }
      }
#pragma unroll
      for (int it = 0; it < STAGE_IT; ++it) {
        const int k = kb + tid + it * SELL_THREADS;
        if (k < a.KC + SELL_ZERO_ROWS) {
          V v;
#pragma unroll
          for (int q = 0; q < QT; ++q) v.v[q] = (k < kn && b0 + q < a.B) ? stg[it][q] : T(0);
          tile[k] = v;
        }
      }
    }
    __syncthreads();

    const int* off = a.off + (int64_t)c * a.nslices;
    const bool last = (c == a.nchunks - 1);
    // few tiles (narrow R): the slices are also split over gridDim.y workgroups per tile
    for (int s = wave + nwaves * (int)blockIdx.y; s < a.nslices; s += nwaves * (int)gridDim.y) {
      const int o = __builtin_amdgcn_readfirstlane(off[s]);
      const int oe = __builtin_amdgcn_readfirstlane(off[s + 1]);
      T acc[QT];
#pragma unroll
      for (int q = 0; q < QT; ++q) acc[q] = T(0);
      // chunks after the first add to what is already in F: request those values now, they are needed only
      // after the gathers of this slice (otherwise every slice ends on an exposed global-load latency)
      const int64_t m = (int64_t)s * 64 + lane;
      T fprev[QT];
#pragma unroll
      for (int q = 0; q < QT; ++q) {
        fprev[q] = T(0);
        if (NP == 1 && c && m < a.M && b0 + q < a.B)   // (two-piece rows: read when the slice is done -- registers)
          fprev[q] = a.F[(a.out_rows ? (int64_t)a.out_rows[b0 + q] : b0 + q) * a.ldf + m];
      }
      const ushort4* ip = reinterpret_cast<const ushort4*>(a.idx) + (int64_t)o * 64 + lane;
      const Vec<T, 4>* vp = reinterpret_cast<const Vec<T, 4>*>(a.val) + (int64_t)o * 64 + lane;
      // NB quads of indices (and values) are fetched ahead of the LDS gathers they feed: the index stream comes
      // from L2 at several hundred cycles per access, the loop must not wait per quad.  The builder pads every
      // slice to whole groups of NB quads and leaves two groups of slack behind the last slice, so the loads need
      // neither clamps nor pad selects nor branches: with branches around them hipcc waits vmcnt(0) right after
      // issuing the next group, i.e. nothing would be in flight while the current group is gathered.  The group
      // past the end of a slice is loaded and never used.  Two register sets (A/B) alternate, nothing is copied.
      const uint2* ipw = reinterpret_cast<const uint2*>(ip);
      const int nq = oe - o;
      uint2 ia[NBQ], ib[NBQ];
      Vec<T, 4> wa[NBQ], wb[NBQ];
      auto fetch = [&](uint2 (&iq)[NBQ], Vec<T, 4> (&wq)[NBQ], int base) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < NBQ; ++j) {
          iq[j] = ipw[(int64_t)(base + j) * 64];
          if (!BIN) wq[j] = vp[(int64_t)(base + j) * 64];
        }
      };
      auto gather = [&](const uint2 (&iq)[NBQ], const Vec<T, 4> (&wq)[NBQ], int cnt) __attribute__((always_inline)) {
        // LDS address of tile[k] = k * 16: one SDWA shift per 16-bit index, used as the address itself (the tile
        // is the only LDS object of this kernel and starts at LDS address 0 -- checked once at kernel entry;
        // going through the generic tile pointer costs one more VALU add per non-zero)
        constexpr int PV = 16 / (int)sizeof(T);   // values per 16-byte piece
        typedef T NV __attribute__((ext_vector_type(PV)));
        using LV = const __attribute__((address_space(3))) NV*;
        const unsigned lb = pid << 4;
        auto row = [&](unsigned addr) __attribute__((always_inline)) {
          V r;
          if (NP == 1) {
            const NV n = *(LV)(uintptr_t)addr;
#pragma unroll
            for (int q = 0; q < PV; ++q) r.v[q] = n[q];
          } else {
            const unsigned a0 = addr | lb;
            const NV n0 = *(LV)(uintptr_t)a0;
            const NV n1 = *(LV)(uintptr_t)(a0 ^ 16u);
#pragma unroll
            for (int q = 0; q < PV; ++q) { r.v[q] = n0[q]; r.v[PV + q] = n1[q]; }
          }
          return r;
        };
#pragma unroll
        for (int j = 0; j < NBQ; ++j) {
          if (j >= cnt) break;   // wave-uniform; cnt == NBQ inside the loop (folds away), smaller only for a slice's tail
          const V t0 = row(half_shl<0>(iq[j].x, TSH));
          const V t1 = row(half_shl<1>(iq[j].x, TSH));
          const V t2 = row(half_shl<0>(iq[j].y, TSH));
          const V t3 = row(half_shl<1>(iq[j].y, TSH));
          if (BIN) {
#pragma unroll
            for (int q = 0; q < QT; ++q) acc[q] += (t0.v[q] + t1.v[q]) + (t2.v[q] + t3.v[q]);
          } else {
#pragma unroll
            for (int q = 0; q < QT; ++q) {
              acc[q] = fma(wq[j].v[0], t0.v[q], acc[q]);
              acc[q] = fma(wq[j].v[1], t1.v[q], acc[q]);
              acc[q] = fma(wq[j].v[2], t2.v[q], acc[q]);
              acc[q] = fma(wq[j].v[3], t3.v[q], acc[q]);
            }
          }
          // two-piece rows: one quad's eight reads at a time (all of a group's at once would need 128 registers)
          if (NP == 2) __builtin_amdgcn_sched_barrier(0);
        }
      };
      // (both gathers of the loop body are unconditional on purpose: a gather under `if` lets the compiler sink
      // its loads into the branch, right in front of their use)
      // Measured and not kept (round 3): a third register set (two groups of quads in flight while one is gathered; the
      // gathers split in two by a compiler barrier so that it fits in 123 registers): 0.605 vs 0.594 ms at C2, 12.8 vs
      // 12.6 ms at C3 -- the index stream is bound by the bytes through the CU's vector-memory path, not by its latency.
      //
      // Two register sets, both requested before the first gather; a slice is a whole number of QUADS (not of groups, since
      // round 3): the last one or two groups are consumed quad by quad under wave-uniform conditions, outside the loop.
      // Requests run up to two groups past the slice (the next slice, or the slack behind the last one).
      fetch(ia, wa, 0);
      fetch(ib, wb, NBQ);
      int u = 0;
      for (; u + 2 * NBQ <= nq; u += 2 * NBQ) {
        gather(ia, wa, NBQ);
        fetch(ia, wa, u + 2 * NBQ);
        gather(ib, wb, NBQ);
        fetch(ib, wb, u + 3 * NBQ);
      }
      {
        const int rest = nq - u;   // 0 .. 2 * NBQ - 1 quads; ia holds quads u .., ib quads u + NBQ ..
        if (rest > 0) gather(ia, wa, rest < NBQ ? rest : NBQ);
        if (rest > NBQ) gather(ib, wb, rest - NBQ);
      }
      if (m < a.M) {
        const bool flag = last && a.clean_deg != nullptr && a.clean_deg[m] == 0;
#pragma unroll
        for (int q = 0; q < QT; ++q) {
          if (b0 + q < a.B) {
            T* f = a.F + (a.out_rows ? (int64_t)a.out_rows[b0 + q] : b0 + q) * a.ldf + m;
            // two-piece rows: odd lanes read the pieces in the order 1, 0 -- their accumulators q and q ^ 4 are swapped
            const T mine = NP == 2 ? (pid ? acc[q ^ (QT / 2)] : acc[q]) : acc[q];
            const T r = mine + ((NP == 2 && c) ? *f : fprev[q]);
            *f = flag ? T(-99) : r;
          }
        }
      }
    }
  }
}

// fp32: 8 columns per tile row since round 3 (SS_SELL_QT=4: the 16-byte rows of rounds 1-2)
template <> int sell_tile_width<float>() {
  if (const char* e = getenv("SS_SELL_QT")) {
    const int v = atoi(e);
    if (v == 4 || v == 8) return v;
  }
  return SELL_QT_F32_DEFAULT;
}
template <> int sell_tile_width<double>() { return 2; }

template <class T>
int sell_max_chunk(int qt) {
  // (KC + 16) * qt * sizeof(T) <= 160 KiB and KC + 15 <= 65535 (16-bit local indices; KC .. KC+15 are the zero rows)
  int64_t kc = (int64_t)(160 * 1024) / ((int64_t)qt * (int64_t)sizeof(T)) - SELL_ZERO_ROWS;
  if (kc > 65535 - (SELL_ZERO_ROWS - 1)) kc = 65535 - (SELL_ZERO_ROWS - 1);
  return (int)kc;
}

template <class T>
int launch_spmm_sell(const DevSell<T>& W, const T* R, int64_t ldr, int64_t B, T* F, int64_t ldf,
                     const int* clean_deg, const int* out_rows) {
  if (B <= 0 || W.rows <= 0) return SS_OK;
  path_add(W.sorted ? "spmm_sell_sorted" : "spmm_sell");
  const int QT = W.qt > 0 ? W.qt : (sizeof(T) == 4 ? 4 : 2);
  SellArgs<T> a{};
  a.off = W.off.p;
  a.idx = W.idx.p;
  a.val = W.val.p;
  a.nslices = W.nslices;
  a.nchunks = W.nchunks;
  a.KC = W.KC;
  a.K = W.cols;
  a.M = W.sorted ? W.vrows : W.rows;
  a.B = B;
  a.R = R;
  a.ldr = ldr;
  a.F = F;
  a.ldf = ldf;
  a.clean_deg = clean_deg;
  a.out_rows = out_rows;
  const size_t lds = (size_t)(W.KC + SELL_ZERO_ROWS) * QT * sizeof(T);
  if (lds > (size_t)160 * 1024) return fail(SS_EINVAL, "SELL chunk does not fit the LDS tile of %d columns", QT);
  const unsigned gx = (unsigned)ceil_div(B, QT);
  unsigned gy = 1;
  if ((int)gx < 2 * ctx().num_cu) {
    gy = (unsigned)ceil_div(2 * ctx().num_cu, (int64_t)gx);
    const unsigned maxy = (unsigned)ceil_div(W.nslices, SELL_THREADS / 64);
    if (gy > maxy) gy = maxy > 0 ? maxy : 1;
  }
  const dim3 grid(gx, gy);
  // first launch of an instantiation: raise its dynamic-LDS limit and check, on the host, what the kernel's LDS
  // addressing relies on -- no static LDS in the code object's kernel, so that the dynamic segment starts at 0.
  // (A violated assumption is an error code for the caller, never a device-side abort.)
  auto prepare = [](const void* fn) -> int {
    hipFuncAttributes fa{};
    SS_HIP(hipFuncGetAttributes(&fa, fn));
    if (fa.sharedSizeBytes != 0)
      return fail(SS_EUNSUPPORTED, "spmm_sell_kernel was built with %zu bytes of static LDS: its tile-relative LDS "
                  "addresses need the dynamic segment at LDS address 0", (size_t)fa.sharedSizeBytes);
    SS_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    return SS_OK;
  };
#define SS_SELL_LAUNCH(QTV, BINV, SLOT)                                                                      \
  do {                                                                                                       \
    static std::atomic<bool> attr_done{false};                                                               \
    if (!attr_done) {                                                                                        \
      SS_TRY(prepare(reinterpret_cast<const void*>(&spmm_sell_kernel<T, QTV, BINV>)));                       \
      attr_done = true;                                                                                      \
    }                                                                                                        \
    hipLaunchKernelGGL((spmm_sell_kernel<T, QTV, BINV>), grid, dim3(SELL_THREADS), lds, ctx().stream, a);    \
  } while (0)
  if constexpr (sizeof(T) == 4) {
    if (QT == 8) { if (W.binary) SS_SELL_LAUNCH(8, true, 0); else SS_SELL_LAUNCH(8, false, 1); }
    else if (QT == 4) { if (W.binary) SS_SELL_LAUNCH(4, true, 2); else SS_SELL_LAUNCH(4, false, 3); }
    else return fail(SS_EINVAL, "SELL tile width must be 4 or 8 (fp32)");
  } else {
    if (QT != 2) return fail(SS_EINVAL, "SELL tile width must be 2 (fp64)");
    if (W.binary) SS_SELL_LAUNCH(2, true, 0); else SS_SELL_LAUNCH(2, false, 1);
  }
#undef SS_SELL_LAUNCH
  SS_LAUNCH_CHECK();
  return SS_OK;
}

// ============================================================== stage 2, narrow, HBM-bound: R chunk in LDS
// F = W*R for B <= 4 columns.  W is cut into column chunks of KC (all of R's rows k0..k0+KC for the B
// columns fit in LDS), stored chunk-major with 16-bit local indices, sub-rows padded to 4 entries so
// that a lane streams 8 B of indices + 16 B of values per step; workgroup (c, slot) keeps chunk c of R
// in LDS and strides over the rows, GL lanes per sub-row, __shfl_xor to fold the lanes.
// Every non-zero of W is read exactly once from HBM (6 B instead of CSR's 8 B); partial sums per chunk
// are combined in fixed order by narrow_reduce_kernel.
template <class T>
struct NarrowArgs {
  const int* off;
  const unsigned short* idx;
  const T* val;
  int64_t M, K;
  int KC, nchunks, B;
  const T* R;
  int64_t ldr;
  T* F;       // direct output when nchunks == 1
  int64_t ldf;
  T* P;       // [nchunks][M][BV] otherwise
};

constexpr int NARROW_THREADS = 1024;

// Sum over each group of GL consecutive lanes (GL = 8, 16, 32, 64) with DPP moves only -- no LDS traffic, where
// __shfl_xor compiles to one ds_bpermute_b32 per step (at B = 4 the 5 x 4 permutes per row group were 0.07 ms of LDS
// time in a 0.18 ms launch).  quad_perm swaps inside quads, row_half_mirror / row_mirror fold 8 and 16 lanes (after the
// quad steps every lane of a quad holds the quad's sum, so the reversed order does not matter), row_bcast15 adds lane
// 15 of rows 0 and 2 to rows 1 and 3, row_bcast31 lane 31 to rows 2 and 3.  The order of the additions is fixed.
// The sum of a group is valid in its LAST lane (for GL <= 16 in all its lanes).
template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ float dpp_add(float v) {
  return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xF, false));
}
template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ double dpp_add(double v) {
  const long long b = __double_as_longlong(v);
  const int lo = __builtin_amdgcn_update_dpp(0, (int)(b & 0xffffffffll), CTRL, ROW_MASK, 0xF, false);
  const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, ROW_MASK, 0xF, false);
  return v + __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
template <int GL, class T>
__device__ __forceinline__ T lane_group_sum(T v) {
  static_assert(GL == 8 || GL == 16 || GL == 32 || GL == 64, "group of 8, 16, 32 or 64 lanes");
  v = dpp_add<0xB1>(v);                       // quad_perm [1,0,3,2]
  v = dpp_add<0x4E>(v);                       // quad_perm [2,3,0,1]
  v = dpp_add<0x141>(v);                      // row_half_mirror
  if constexpr (GL >= 16) v = dpp_add<0x140>(v);        // row_mirror
  if constexpr (GL >= 32) v = dpp_add<0x142, 0xA>(v);   // row_bcast15 into rows 1 and 3
  if constexpr (GL >= 64) v = dpp_add<0x143, 0xC>(v);   // row_bcast31 into rows 2 and 3
  return v;
}

// VEC = columns (1, 2 or 4); GL = lanes that share one sub-row (64, 32, 16 or 8), so a wave streams 64/GL rows at once;
// UR row groups per step; NBQ quads requested per lane and row.
template <class T, int VEC, int GL, int UR, int NBQ>
__global__ void __launch_bounds__(NARROW_THREADS) spmm_chunked_narrow_kernel(NarrowArgs<T> a) {
  constexpr int BV = VEC;
  extern __shared__ __align__(16) unsigned char smem_raw[];
  T* tile = reinterpret_cast<T*>(smem_raw);  // [KC + 1][BV]; row KC stays zero (padding target)
  using V = Vec<T, VEC>;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nwaves = blockDim.x >> 6;
  const int c = blockIdx.x % a.nchunks;
  const int slot = blockIdx.x / a.nchunks, nslots = gridDim.x / a.nchunks;
  const int64_t k0 = (int64_t)c * a.KC;
  const int kn = (int)((a.K - k0 < a.KC) ? (a.K - k0) : a.KC);

  // 32-byte tile rows (two 16-byte halves per non-zero): all lanes fetch the same half at the same time, so with
  // a plain layout only every other 16-byte slot of the 256-byte LDS line is ever hit (a fixed 2-way conflict);
  // half h of row k is stored at h ^ ((k >> 3) & 1), which spreads random k over all 16 slots
  constexpr bool SWZ = (BV * sizeof(T) == 32);
  constexpr int HB = 16 / (int)sizeof(T);  // values per 16-byte half
  // rows of R packed without gaps (ldr == B == BV) and a 16-byte aligned chunk: the tile is a plain copy, moved in
  // 16-byte pieces (KC is a multiple of 4 by construction); otherwise element by element
  constexpr int PW16 = 16 / (int)sizeof(T);
  const bool packed = a.ldr == BV && a.B == BV &&
                      ((reinterpret_cast<uintptr_t>(a.R + k0 * a.ldr) & 15) == 0) && ((a.KC * BV) % PW16 == 0);
  if (packed) {
    using P16 = Vec<T, PW16>;
    const P16* __restrict__ src = reinterpret_cast<const P16*>(a.R + k0 * a.ldr);
    P16* dst = reinterpret_cast<P16*>(tile);
    const int npieces = ((a.KC + 1) * BV + PW16 - 1) / PW16, nfull = (kn * BV) / PW16;
    for (int i = tid; i < npieces; i += blockDim.x) {
      P16 v;
      if (i < nfull) {
        v = src[i];
      } else {
#pragma unroll
        for (int e = 0; e < PW16; ++e) {
          const int x = i * PW16 + e;   // element of the tile; row x / BV
          v.v[e] = (x < kn * BV) ? a.R[k0 * a.ldr + x] : T(0);
        }
      }
      if ((i + 1) * PW16 <= (a.KC + 1) * BV) {
        // 32-byte tile rows: half h of row k sits at h ^ ((k >> 3) & 1) (see below), i.e. piece i at i ^ ((i >> 4) & 1)
        dst[SWZ ? (i ^ ((i >> 4) & 1)) : i] = v;
      } else {
#pragma unroll
        for (int e = 0; e < PW16; ++e)
          if (i * PW16 + e < (a.KC + 1) * BV) tile[i * PW16 + e] = v.v[e];
      }
    }
  } else {
    for (int e = tid; e < (a.KC + 1) * BV; e += blockDim.x) {
      const int k = e / BV, b = e - k * BV;
      const T v = (k < kn && b < a.B) ? a.R[(k0 + k) * a.ldr + b] : T(0);
      if constexpr (SWZ) tile[k * BV + ((((b / HB) ^ (k >> 3)) & 1) * HB) + (b % HB)] = v;
      else tile[e] = v;
    }
  }
  __syncthreads();
  auto tile_row = [&](int k) __attribute__((always_inline)) {
    V r;
    if constexpr (SWZ) {
      using H = Vec<T, HB>;
      const int s = (k >> 3) & 1;
      const H lo = *reinterpret_cast<const H*>(&tile[k * BV + s * HB]);
      const H hi = *reinterpret_cast<const H*>(&tile[k * BV + (s ^ 1) * HB]);
#pragma unroll
      for (int i = 0; i < HB; ++i) { r.v[i] = lo.v[i]; r.v[HB + i] = hi.v[i]; }
    } else {
      r = *reinterpret_cast<const V*>(&tile[k * BV]);
    }
    return r;
  };

  const int* __restrict__ off = a.off + (int64_t)c * a.M;
  {
    constexpr int RPS = 64 / GL;  // rows per wave-step
    const int sub = lane / GL, gl = lane % GL;
    const ushort4* __restrict__ ip = reinterpret_cast<const ushort4*>(a.idx);
    const Vec<T, 4>* __restrict__ vp = reinterpret_cast<const Vec<T, 4>*>(a.val);
    const int64_t stride = (int64_t)nslots * nwaves * RPS * UR;
    // Software pipeline over the row groups of this wave: while group i is gathered, the quads of group i+1 are in
    // flight and the sub-row bounds of group i+2 are being read, so the W stream never waits for a dependent load.
    // Every lane requests NBQ quads per row unconditionally (q, q + GL, ...; clamped to a valid quad, masked below):
    // loads stay out of branches and their waits stay counted.  Rows longer than NBQ * GL quads take the rolled tail.
    struct Bounds { int lo[UR], hi[UR]; };   // raw off[m], off[m + 1] (row clamped; validity is applied at use)
    // quads in flight for one row group; mask bit u*NBQ + nb: this lane's quad nb of row u is inside the sub-row,
    // bit 31: some row of the group is longer than NBQ * GL quads
    struct Quads { ushort4 iv[UR][NBQ]; Vec<T, 4> w[UR][NBQ]; unsigned mask; };
    auto bounds = [&](int64_t mb, Bounds& b) __attribute__((always_inline)) {
#pragma unroll
      for (int u = 0; u < UR; ++u) {
        const int64_t m = mb + u * RPS + sub;
        const int64_t mm = m < a.M ? m : a.M - 1;
        b.lo[u] = off[mm];
        b.hi[u] = off[mm + 1];
      }
    };
    auto issue = [&](int64_t mb, const Bounds& b, Quads& x) __attribute__((always_inline)) {
      unsigned mask = 0;
#pragma unroll
      for (int u = 0; u < UR; ++u) {
        const int oe = (mb + u * RPS + sub < a.M) ? b.hi[u] : b.lo[u];
#pragma unroll
        for (int nb = 0; nb < NBQ; ++nb) {
          const int q = b.lo[u] + gl + nb * GL;
          const bool in = q < oe;
          const int qq = in ? q : b.lo[u];
          {
            // W is read exactly once per launch: non-temporal loads keep the stream from displacing the tile of R and the
            // partial sums in L2 (100k x 100k, 1 %: B = 1 0.124 -> 0.115 ms, B = 2 0.136 -> 0.120 ms, B = 4 0.185 -> 0.178 ms)
            typedef unsigned nt_u2 __attribute__((ext_vector_type(2)));
            typedef T nt_t4 __attribute__((ext_vector_type(4)));
            const nt_u2 ii = __builtin_nontemporal_load(reinterpret_cast<const nt_u2*>(&ip[qq]));
            const nt_t4 ww = __builtin_nontemporal_load(reinterpret_cast<const nt_t4*>(&vp[qq]));
            x.iv[u][nb] = ushort4((unsigned short)(ii.x & 0xffffu), (unsigned short)(ii.x >> 16), (unsigned short)(ii.y & 0xffffu),
                                  (unsigned short)(ii.y >> 16));
            x.w[u][nb].v[0] = ww.x; x.w[u][nb].v[1] = ww.y; x.w[u][nb].v[2] = ww.z; x.w[u][nb].v[3] = ww.w;
          }
          mask |= in ? (1u << (u * NBQ + nb)) : 0u;
        }
        mask |= (b.lo[u] + NBQ * GL < oe) ? 0x80000000u : 0u;
      }
      x.mask = mask;
    };
    auto quad = [&](T (&acc)[VEC], const ushort4 jv, const Vec<T, 4> x) __attribute__((always_inline)) {
      const V r0 = tile_row((int)jv.x);
      const V r1 = tile_row((int)jv.y);
      const V r2 = tile_row((int)jv.z);
      const V r3 = tile_row((int)jv.w);
#pragma unroll
      for (int i = 0; i < VEC; ++i) {
        acc[i] = fma(x.v[0], r0.v[i], acc[i]);
        acc[i] = fma(x.v[1], r1.v[i], acc[i]);
        acc[i] = fma(x.v[2], r2.v[i], acc[i]);
        acc[i] = fma(x.v[3], r3.v[i], acc[i]);
      }
    };
    int64_t mb = ((int64_t)slot * nwaves + wave) * RPS * UR;
    Bounds bn;
    Quads xa, xb;
    if (mb < a.M) {
      Bounds b0;
      bounds(mb, b0);
      bounds(mb + stride < a.M ? mb + stride : mb, bn);
      issue(mb, b0, xa);
    }
    // one row group: request the quads of the next group (its bounds arrived a step ago) and the bounds of the one
    // after, then gather the current group.  Two copies of the step alternate the quad registers (a copy of a
    // register with a load in flight would wait for it); only the bounds, long arrived, are moved.
    auto step = [&](const int64_t mb, const Quads& xc, Quads& xn) __attribute__((always_inline)) {
      // unconditional (the last group re-requests itself): no load sits in a branch
      const int64_t mb1 = mb + stride < a.M ? mb + stride : mb;
      const int64_t mb2 = mb + 2 * stride < a.M ? mb + 2 * stride : mb;
      Bounds bnn;
      bounds(mb2, bnn);
      issue(mb1, bn, xn);
      __builtin_amdgcn_sched_barrier(0);  // the requests go out BEFORE the gathers of the current group (hipcc sinks them otherwise)
      T acc[UR][VEC];
#pragma unroll
      for (int u = 0; u < UR; ++u) {
#pragma unroll
        for (int i = 0; i < VEC; ++i) acc[u][i] = T(0);
#pragma unroll
        for (int nb = 0; nb < NBQ; ++nb) {
          // lanes past the end of the sub-row hold a clamped (valid) quad: gather it with zero weights
          const bool has = (xc.mask >> (u * NBQ + nb)) & 1u;
          Vec<T, 4> w = xc.w[u][nb];
#pragma unroll
          for (int e = 0; e < 4; ++e) w.v[e] = has ? w.v[e] : T(0);
          quad(acc[u], xc.iv[u][nb], w);
        }
      }
      if (__any((int)(xc.mask >> 31))) {
        // rows longer than the NBQ * GL quads requested ahead (rare by the choice of GL): bounds re-read, rolled loop
#pragma unroll
        for (int u = 0; u < UR; ++u) {
          const int64_t m = mb + u * RPS + sub;
          if (m < a.M) {
            const int lo = off[m], hi = off[m + 1];
            for (int q = lo + gl + NBQ * GL; q < hi; q += GL) quad(acc[u], ip[q], vp[q]);
          }
        }
      }
#pragma unroll
      for (int u = 0; u < UR; ++u) {
#pragma unroll
        for (int i = 0; i < VEC; ++i) acc[u][i] = lane_group_sum<GL>(acc[u][i]);
        const int64_t m = mb + u * RPS + sub;
        if (gl == GL - 1 && m < a.M) {   // the group's sum sits in its last lane
          if (a.P) {
            T* p = a.P + ((int64_t)c * a.M + m) * BV;
#pragma unroll
            for (int i = 0; i < VEC; ++i) p[i] = acc[u][i];
          } else {
#pragma unroll
            for (int i = 0; i < VEC; ++i)
              if (i < a.B) a.F[m * a.ldf + i] = acc[u][i];
          }
        }
      }
      bn = bnn;
    };
    while (mb < a.M) {
      step(mb, xa, xb);
      mb += stride;
      if (mb >= a.M) break;
      step(mb, xb, xa);
      mb += stride;
    }
  }
}

// F[m][0..B) = sum of the per-chunk partial sums, in chunk order (one thread per row: BV <= 4 values)
template <class T, int BV>
__global__ void narrow_reduce_kernel(const T* __restrict__ P, int nchunks, int64_t M, int B, T* __restrict__ F,
                                     int64_t ldf) {
  using PV = Vec<T, BV>;
  for (int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; m < M; m += (int64_t)gridDim.x * blockDim.x) {
    PV s = *reinterpret_cast<const PV*>(P + m * BV);
    for (int c = 1; c < nchunks; ++c) {
      const PV t = *reinterpret_cast<const PV*>(P + ((int64_t)c * M + m) * BV);
#pragma unroll
      for (int e = 0; e < BV; ++e) s.v[e] += t.v[e];
    }
#pragma unroll
    for (int e = 0; e < BV; ++e)
      if (e < B) F[m * ldf + e] = s.v[e];
  }
}

template <class T>
int narrow_chunk_cols(int bv) {
  int64_t kc = (int64_t)(160 * 1024) / ((int64_t)bv * (int64_t)sizeof(T)) - 1;
  if (kc > 65535) kc = 65535;
  kc &= ~3LL;  // chunks of R start on 16-byte boundaries (vector staging of the tile)
  return (int)kc;
}

template <class T, int VEC, int GL, int UR, int NBQ = 1>
static int launch_narrow_variant(const NarrowArgs<T>& a, unsigned grid, size_t lds) {
  static std::atomic<bool> attr_set{false};
  if (!attr_set) {
    SS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&spmm_chunked_narrow_kernel<T, VEC, GL, UR, NBQ>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_set = true;
  }
  hipLaunchKernelGGL((spmm_chunked_narrow_kernel<T, VEC, GL, UR, NBQ>), dim3(grid), dim3(NARROW_THREADS), lds,
                     ctx().stream, a);
  SS_LAUNCH_CHECK();
  return SS_OK;
}

// Row groups per step (UR) such that two sets of quads in flight (UR * NBQ quads of 8 + 4*sizeof(T) bytes per lane and
// set), the UR * VEC accumulators, four gathered tile rows and the bounds stay inside the 128 registers a lane has at 16
// waves per CU.
constexpr int narrow_ur(int elem, int vec, int nbq) {
  const int per_quad = 2 + elem;  // registers
  int ur = 4;
  while (ur > 1 && 2 * ur * nbq * per_quad + (ur + 4) * vec * (elem / 4) + 4 * ur + 24 > 124) ur >>= 1;
  return ur;
}

template <class T, int VEC>
static int launch_narrow_width(const NarrowArgs<T>& a, unsigned grid, size_t lds, double mean_quads) {
  constexpr int E = (int)sizeof(T);
  // GL lanes per row, one request per lane (two above 64 quads): the smallest capacity about 10 % above the mean
  // sub-row length in quads.  Measured (100k x 100k, 1 %): lanes past the end of a sub-row are not free (a step is
  // bound by the latency of its requests, so rows per step count: B = 4, mean 25 quads, 32 vs 64 lanes per row
  // 0.20 vs 0.35 ms), while the rolled tail for the few longer rows costs little
  const double need = 1.1 * mean_quads + 1.0;
  if (need > 64.0) return launch_narrow_variant<T, VEC, 64, narrow_ur(E, VEC, 2), 2>(a, grid, lds);
  if (need > 32.0) return launch_narrow_variant<T, VEC, 64, narrow_ur(E, VEC, 1)>(a, grid, lds);
  if (need > 16.0) return launch_narrow_variant<T, VEC, 32, narrow_ur(E, VEC, 1)>(a, grid, lds);
  if (need > 8.0) return launch_narrow_variant<T, VEC, 16, narrow_ur(E, VEC, 1)>(a, grid, lds);
  return launch_narrow_variant<T, VEC, 8, narrow_ur(E, VEC, 1)>(a, grid, lds);
}

template <class T>
int launch_spmm_chunked_narrow(const DevChunked<T>& W, int bv, const T* R, int64_t ldr, int B, T* F, int64_t ldf,
                               DevBuf<T>& partial) {
  if (W.rows <= 0 || B <= 0) return SS_OK;
  if (W.align != 4) return fail(SS_EINVAL, "narrow operand must be quad-aligned");
  path_add("spmm_chunked_narrow");
  NarrowArgs<T> a{};
  a.off = W.off.p; a.idx = W.idx.p; a.val = W.val.p;
  a.M = W.rows; a.K = W.cols; a.KC = W.SC; a.nchunks = W.nchunks; a.B = B;
  a.R = R; a.ldr = ldr; a.F = F; a.ldf = ldf; a.P = nullptr;
  if (W.nchunks > 1) {
    const size_t need = (size_t)W.nchunks * (size_t)W.rows * (size_t)bv;
    if (partial.n < need) SS_TRY(partial.alloc(need));
    a.P = partial.p;
  }
  // one resident round: the 160 KB tile allows one workgroup per CU, and every workgroup strides over
  // all rows of its chunk, so a partial second round would only add idle CUs
  int nslots = ctx().num_cu / W.nchunks;
  const int64_t max_slots = ceil_div(W.rows, NARROW_THREADS / 64);
  if (nslots > max_slots) nslots = (int)max_slots;
  if (nslots < 1) nslots = 1;
  const unsigned grid = (unsigned)(W.nchunks * nslots);
  const size_t lds = (size_t)(W.SC + 1) * bv * sizeof(T);
  const double mean_quads = (double)W.stored / 4.0 / ((double)W.rows * (double)W.nchunks);
  int rc;
  switch (bv) {
    case 1: rc = launch_narrow_width<T, 1>(a, grid, lds, mean_quads); break;
    case 2: rc = launch_narrow_width<T, 2>(a, grid, lds, mean_quads); break;
    case 4: rc = launch_narrow_width<T, 4>(a, grid, lds, mean_quads); break;
    default: return fail(SS_EINVAL, "narrow width must be 1, 2 or 4");
  }
  SS_TRY(rc);
  if (W.nchunks > 1) {
    const dim3 rg(grid_1d(W.rows, 256)), rb(256);
    switch (bv) {
      case 1: hipLaunchKernelGGL((narrow_reduce_kernel<T, 1>), rg, rb, 0, ctx().stream, partial.p, W.nchunks, W.rows, B, F, ldf); break;
      case 2: hipLaunchKernelGGL((narrow_reduce_kernel<T, 2>), rg, rb, 0, ctx().stream, partial.p, W.nchunks, W.rows, B, F, ldf); break;
      default: hipLaunchKernelGGL((narrow_reduce_kernel<T, 4>), rg, rb, 0, ctx().stream, partial.p, W.nchunks, W.rows, B, F, ldf); break;
    }
    SS_LAUNCH_CHECK();
  }
  return SS_OK;
}

// ============================================================== transpose (layout conversion)
// in: rows x cols, element (r,c) at in[r*ldin + c]; out: (c,r) at out[c*ldout + r]
template <class T>
__global__ void __launch_bounds__(256) transpose_kernel(const T* __restrict__ in, int64_t rows, int64_t cols,
                                                        int64_t ldin, T* __restrict__ out, int64_t ldout) {
  __shared__ T tile[64][65];
  const int64_t c0 = (int64_t)blockIdx.x * 64, r0 = (int64_t)blockIdx.y * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int i = ty; i < 64; i += 4) {
    const int64_t r = r0 + i, c = c0 + tx;
    tile[i][tx] = (r < rows && c < cols) ? in[r * ldin + c] : T(0);
  }
  __syncthreads();
  for (int i = ty; i < 64; i += 4) {
    const int64_t c = c0 + i, r = r0 + tx;
    if (r < rows && c < cols) out[c * ldout + r] = tile[tx][i];
  }
}

template <class T>
int launch_transpose(const T* in, int64_t rows, int64_t cols, int64_t ldin, T* out, int64_t ldout) {
  if (rows <= 0 || cols <= 0) return SS_OK;
  dim3 grid((unsigned)ceil_div(cols, 64), (unsigned)ceil_div(rows, 64));
  hipLaunchKernelGGL(transpose_kernel<T>, grid, dim3(256), 0, ctx().stream, in, rows, cols, ldin, out, ldout);
  SS_LAUNCH_CHECK();
  return SS_OK;
}

// ============================================================== unpermute (skew-sorted SELL results)
template <class T>
__global__ void unpermute_kernel(const T* __restrict__ in, int64_t ldin, int64_t nrows, int64_t nt,
                                 const int* __restrict__ vfirst, const int* __restrict__ inv,
                                 const int* __restrict__ clean_deg, T* __restrict__ out, int64_t ldout) {
  const int64_t r = blockIdx.y;
  const T* src = in + r * ldin;
  T* dst = out + r * ldout;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < nt; t += (int64_t)gridDim.x * blockDim.x) {
    T v = T(0);
    for (int x = vfirst[t]; x < vfirst[t + 1]; ++x) v += src[inv[x]];  // the row of `in` sits in L2 meanwhile
    dst[t] = (clean_deg != nullptr && clean_deg[t] == 0) ? T(-99) : v;
  }
}

template <class T>
int launch_unpermute(const T* in, int64_t ldin, int64_t nrows, int64_t nt, const int* vfirst, const int* inv,
                     const int* clean_deg, T* out, int64_t ldout) {
  if (nrows <= 0 || nt <= 0) return SS_OK;
  for (int64_t r0 = 0; r0 < nrows; r0 += 65535) {
    const int64_t nb = nrows - r0 < 65535 ? nrows - r0 : 65535;
    dim3 grid((unsigned)grid_1d(nt, 256, 64), (unsigned)nb);
    hipLaunchKernelGGL(unpermute_kernel<T>, grid, dim3(256), 0, ctx().stream, in + r0 * ldin, ldin, nb, nt, vfirst,
                       inv, clean_deg, out + r0 * ldout, ldout);
    SS_LAUNCH_CHECK();
  }
  return SS_OK;
}

// ============================================================== k-fold: degrees of the graph without a fold's members
// construct(y, X, queries) (src/core.jl:148-201) drops the query rows from the sources and the feature
// columns named after them: kf[f] loses one per member row with X[g,f] != 0, ks[s] one per member feature
// column with X[s,g] != 0, kt[t] one per member with Y[g,t] != 0.  One wave per member; integer atomics.
__global__ void fold_degrees_kernel(const int* __restrict__ xptr, const int* __restrict__ xidx,
                                    const int* __restrict__ tptr, const int* __restrict__ tidx,
                                    const int* __restrict__ yptr, const int* __restrict__ yidx,
                                    const int* __restrict__ members, int64_t nmembers, int* __restrict__ kf,
                                    int* __restrict__ ks, int* __restrict__ kt) {
  const int lane = threadIdx.x & 63;
  const int64_t wave0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t m = wave0; m < nmembers; m += nwaves) {
    const int g = members[m];
    for (int x = xptr[g] + lane; x < xptr[g + 1]; x += 64) atomicSub(&kf[xidx[x]], 1);
    for (int x = tptr[g] + lane; x < tptr[g + 1]; x += 64) atomicSub(&ks[tidx[x]], 1);
    for (int x = yptr[g] + lane; x < yptr[g + 1]; x += 64) atomicSub(&kt[yidx[x]], 1);
  }
}

template <class T>
int launch_fold_degrees(const DevCsr<T>& X, const DevCsr<T>& XT, const DevCsr<T>& Y, const int* members,
                        int64_t nmembers, int* kf, int* ks, int* kt) {
  if (nmembers <= 0) return SS_OK;
  hipLaunchKernelGGL(fold_degrees_kernel, dim3(grid_1d(nmembers * 64, 256)), dim3(256), 0, ctx().stream, X.ptr.p,
                     X.idx.p, XT.ptr.p, XT.idx.p, Y.ptr.p, Y.idx.p, members, nmembers, kf, ks, kt);
  SS_LAUNCH_CHECK();
  return SS_OK;
}

// reciprocal degrees of the fold's graph; members are neither features nor sources in it (-> 0)
template <class T>
__global__ void fold_inverse_kernel(const int* __restrict__ kf, const int* __restrict__ ks,
                                    const int* __restrict__ fold, int phi, int64_t nf, int64_t ns,
                                    T* __restrict__ inv_kf, T* __restrict__ inv_ks) {
  const int64_t n = nf > ns ? nf : ns;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const bool member = fold[i] == phi;
    if (i < nf) inv_kf[i] = (!member && kf[i] > 0) ? T(1) / T(kf[i]) : T(0);
    if (i < ns) inv_ks[i] = (!member && ks[i] > 0) ? T(1) / T(ks[i]) : T(0);
  }
}

template <class T>
int launch_fold_inverse(const int* kf, const int* ks, const int* fold, int phi, int64_t nf, int64_t ns, T* inv_kf,
                        T* inv_ks) {
  const int64_t n = nf > ns ? nf : ns;
  if (n <= 0) return SS_OK;
  hipLaunchKernelGGL(fold_inverse_kernel<T>, dim3(grid_1d(n, 256)), dim3(256), 0, ctx().stream, kf, ks, fold, phi, nf,
                     ns, inv_kf, inv_ks);
  SS_LAUNCH_CHECK();
  return SS_OK;
}

// ============================================================== top-L per row (ranked evaluation on the device)
// One workgroup per row.  Radix select (4 passes of 8 bits over an order-preserving integer image of the
// float) finds the L-th largest key; elements above it are collected, ties at the threshold are taken in
// ascending column order (what a stable descending sortperm does), then the L pairs are bitonic-sorted in LDS.
__device__ __forceinline__ unsigned float_key(float f) {
  const unsigned u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);  // larger float <=> larger key
}

constexpr int TOPL_THREADS = 256;
constexpr int TOPL_MAX = 1024;

// `only_flagged`: second launch after topl_bound_kernel -- serve only the rows it marked (oidx[row*L] == -1)
__global__ void __launch_bounds__(TOPL_THREADS) topl_kernel(const float* __restrict__ scores, int64_t ncols, int64_t ld,
                                                            int L, int* __restrict__ oidx, float* __restrict__ oval,
                                                            int only_flagged) {
  if (only_flagged && oidx[(int64_t)blockIdx.x * L] != -1) return;
  __shared__ unsigned hist[256];
  __shared__ unsigned long long sel[TOPL_MAX];  // (key << 32) | ~column : descending sort = score desc, column asc
  __shared__ unsigned s_prefix, s_need, s_count, s_base;
  const int tid = threadIdx.x;
  const float* row = scores + (int64_t)blockIdx.x * ld;
  if (tid == 0) { s_prefix = 0; s_need = (unsigned)L; }
  __syncthreads();
  // ---- radix select: after the loop s_prefix is the key of the L-th largest element,
  //      s_need how many elements equal to it are wanted
  for (int pass = 0; pass < 4; ++pass) {
    const int shift = 24 - 8 * pass;
    hist[tid] = 0;
    __syncthreads();
    const unsigned prefix = s_prefix;
    const unsigned himask = pass == 0 ? 0u : (0xFFFFFFFFu << (shift + 8));
    for (int64_t c = tid; c < ncols; c += TOPL_THREADS) {
      const unsigned k = float_key(row[c]);
      if ((k & himask) == prefix) atomicAdd(&hist[(k >> shift) & 255u], 1u);
    }
    __syncthreads();
    if (tid == 0) {
      unsigned need = s_need, b = 255;
      for (;; --b) {  // buckets from the largest digit down
        if (hist[b] >= need) break;
        need -= hist[b];
        if (b == 0) break;
      }
      s_prefix = prefix | (b << shift);
      s_need = need;
    }
    __syncthreads();
  }
  const unsigned kth = s_prefix;
  const unsigned need_eq = s_need;
  if (tid == 0) { s_count = 0; s_base = 0; }
  __syncthreads();
  // ---- strictly above the threshold: any order (sorted afterwards)
  for (int64_t c = tid; c < ncols; c += TOPL_THREADS) {
    const unsigned k = float_key(row[c]);
    if (k > kth) {
      const unsigned p = atomicAdd(&s_count, 1u);
      sel[p] = ((unsigned long long)k << 32) | (unsigned)(~(unsigned)c);
    }
  }
  __syncthreads();
  const unsigned above = s_count;
  // ---- ties at the threshold: the first need_eq in column order (block-wide ordered compaction)
  for (int64_t c0 = 0; c0 < ncols && s_base < need_eq; c0 += TOPL_THREADS) {
    const int64_t c = c0 + tid;
    const bool eq = c < ncols && float_key(row[c]) == kth;
    // ordered rank of this thread among the equal ones of the chunk: wave ballots + per-wave offsets in hist[]
    const unsigned long long m = __ballot(eq);
    const int lane = tid & 63, w = tid >> 6;
    if (lane == 0) hist[w] = (unsigned)__popcll(m);
    __syncthreads();
    unsigned before = s_base;
    for (int i = 0; i < w; ++i) before += hist[i];
    const unsigned r = before + (unsigned)__popcll(m & ((1ull << lane) - 1ull));
    if (eq && r < need_eq) sel[above + r] = ((unsigned long long)kth << 32) | (unsigned)(~(unsigned)c);
    __syncthreads();
    if (tid == 0) s_base += hist[0] + hist[1] + hist[2] + hist[3];
    __syncthreads();
  }
  // ---- sort the L pairs descending (bitonic on the next power of two, padding with 0 = smallest)
  int P = 1;
  while (P < L) P <<= 1;
  for (int i = L + tid; i < P; i += TOPL_THREADS) sel[i] = 0ull;
  __syncthreads();
  for (int k = 2; k <= P; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = tid; i < P; i += TOPL_THREADS) {
        const int ixj = i ^ j;
        if (ixj > i) {
          const unsigned long long a = sel[i], b = sel[ixj];
          const bool desc = (i & k) == 0;
          if (desc ? (a < b) : (a > b)) { sel[i] = b; sel[ixj] = a; }
        }
      }
      __syncthreads();
    }
  for (int i = tid; i < L; i += TOPL_THREADS) {
    const unsigned long long e = sel[i];
    const unsigned c = ~(unsigned)(e & 0xFFFFFFFFull);
    oidx[(int64_t)blockIdx.x * L + i] = (int)c;
    oval[(int64_t)blockIdx.x * L + i] = row[c];
  }
}

// Fast path (two reads of the row instead of six, no per-element histogram): every thread takes the maximum key of its
// strided share; the L-th largest of those 1024 maxima is a lower bound of the L-th largest element (they are L
// distinct elements), so only elements at or above it can be among the top L.  They are collected as (key, ~column)
// pairs -- unique, and their descending order IS score descending / column ascending, so ties need no special case --
// and bitonic-sorted in LDS.  A row with more than TOPL2_CAP candidates (long runs of equal scores at the bound:
// e.g. mostly zeros) is marked with oidx = -1 and left to the radix-select kernel above.
constexpr int TOPL2_THREADS = 1024;
constexpr int TOPL2_CAP = 4096;

__global__ void __launch_bounds__(TOPL2_THREADS) topl_bound_kernel(const float* __restrict__ scores, int64_t ncols,
                                                                   int64_t ld, int L, int* __restrict__ oidx,
                                                                   float* __restrict__ oval) {
  __shared__ unsigned long long cand[TOPL2_CAP];
  __shared__ unsigned hist[256];
  __shared__ unsigned s_prefix, s_need, s_count;
  const int tid = threadIdx.x, lane = tid & 63;
  const float* row = scores + (int64_t)blockIdx.x * ld;
  unsigned mx = 0;  // smaller than the key of any float
  for (int64_t c = tid; c < ncols; c += TOPL2_THREADS) {
    const unsigned k = float_key(row[c]);
    mx = k > mx ? k : mx;
  }
  if (tid == 0) { s_prefix = 0; s_need = (unsigned)L; s_count = 0; }
  __syncthreads();
  // radix select of the L-th largest thread maximum (L <= 1024 = number of maxima; threads without elements hold 0)
  for (int pass = 0; pass < 4; ++pass) {
    const int shift = 24 - 8 * pass;
    if (tid < 256) hist[tid] = 0;
    __syncthreads();
    const unsigned prefix = s_prefix;
    const unsigned himask = pass == 0 ? 0u : (0xFFFFFFFFu << (shift + 8));
    if ((mx & himask) == prefix) atomicAdd(&hist[(mx >> shift) & 255u], 1u);
    __syncthreads();
    if (tid == 0) {
      unsigned need = s_need, b = 255;
      for (;; --b) {
        if (hist[b] >= need) break;
        need -= hist[b];
        if (b == 0) break;
      }
      s_prefix = prefix | (b << shift);
      s_need = need;
    }
    __syncthreads();
  }
  const unsigned bound = s_prefix;
  // candidates: one counter update per wave (ballot), order irrelevant (sorted below)
  for (int64_t c0 = 0; c0 < ncols; c0 += TOPL2_THREADS) {
    const int64_t c = c0 + tid;
    const unsigned k = c < ncols ? float_key(row[c]) : 0u;
    const bool in = c < ncols && k >= bound;
    const unsigned long long m = __ballot(in);
    if (m) {
      unsigned base = 0;
      if (lane == 0) base = atomicAdd(&s_count, (unsigned)__popcll(m));
      base = __shfl(base, 0);
      const unsigned p = base + (unsigned)__popcll(m & ((1ull << lane) - 1ull));
      if (in && p < (unsigned)TOPL2_CAP) cand[p] = ((unsigned long long)k << 32) | (unsigned)(~(unsigned)c);
    }
  }
  __syncthreads();
  const unsigned n = s_count;
  if (n > (unsigned)TOPL2_CAP) {
    if (tid == 0) oidx[(int64_t)blockIdx.x * L] = -1;
    return;
  }
  int P = 1;
  while (P < (int)n) P <<= 1;   // n >= L
  for (int i = (int)n + tid; i < P; i += TOPL2_THREADS) cand[i] = 0ull;
  __syncthreads();
  for (int k = 2; k <= P; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = tid; i < P; i += TOPL2_THREADS) {
        const int ixj = i ^ j;
        if (ixj > i) {
          const unsigned long long a = cand[i], b = cand[ixj];
          const bool desc = (i & k) == 0;
          if (desc ? (a < b) : (a > b)) { cand[i] = b; cand[ixj] = a; }
        }
      }
      __syncthreads();
    }
  for (int i = tid; i < L; i += TOPL2_THREADS) {
    const unsigned long long e = cand[i];
    const unsigned c = ~(unsigned)(e & 0xFFFFFFFFull);
    oidx[(int64_t)blockIdx.x * L + i] = (int)c;
    oval[(int64_t)blockIdx.x * L + i] = row[c];
  }
}

int launch_topl(const float* scores, int64_t nrows, int64_t ncols, int64_t ld, int L, int* oidx, float* oval) {
  if (nrows <= 0) return SS_OK;
  if (L < 1 || L > TOPL_MAX || L > ncols) return fail(SS_EINVAL, "top-L needs 1 <= L <= min(%d, ncols)", TOPL_MAX);
  const bool fast = !(getenv("SS_TOPL_BOUND") && atoi(getenv("SS_TOPL_BOUND")) == 0);
  for (int64_t r0 = 0; r0 < nrows; r0 += (1 << 30)) {
    const int64_t nb = nrows - r0 < (1 << 30) ? nrows - r0 : (1 << 30);
    if (fast) {
      hipLaunchKernelGGL(topl_bound_kernel, dim3((unsigned)nb), dim3(TOPL2_THREADS), 0, ctx().stream, scores + r0 * ld,
                         ncols, ld, L, oidx + r0 * L, oval + r0 * L);
      SS_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(topl_kernel, dim3((unsigned)nb), dim3(TOPL_THREADS), 0, ctx().stream, scores + r0 * ld, ncols, ld,
                       L, oidx + r0 * L, oval + r0 * L, fast ? 1 : 0);
    SS_LAUNCH_CHECK();
  }
  return SS_OK;
}

// ============================================================== LOO clean! fix-up
// In fold i target t has degree kt[t] - [Y[i,t] != 0] (src/core.jl:479): the kt == 0 columns are
// flagged by the SpMM epilogue, here the columns whose single edge belongs to the query itself.
template <class T>
__global__ void loo_clean_fix_kernel(const int* __restrict__ tptr, const int* __restrict__ tidx,
                                     const int* __restrict__ kt, int64_t nt, int64_t i_begin, int64_t nrows,
                                     T* __restrict__ out, int64_t ld) {
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < nt; t += (int64_t)gridDim.x * blockDim.x) {
    if (kt[t] == 1) {
      const int64_t s = tidx[tptr[t]];
      if (s >= i_begin && s < i_begin + nrows) out[(s - i_begin) * ld + t] = T(-99);
    }
  }
}

template <class T>
int launch_loo_clean_fix(const DevCsr<T>& YsT, const int* kt, int64_t i_begin, int64_t nrows, T* out, int64_t ld) {
  if (YsT.rows <= 0 || nrows <= 0) return SS_OK;
  hipLaunchKernelGGL(loo_clean_fix_kernel<T>, dim3(grid_1d(YsT.rows, 256)), dim3(256), 0, ctx().stream,
                     YsT.ptr.p, YsT.idx.p, kt, YsT.rows, i_begin, nrows, out, ld);
  SS_LAUNCH_CHECK();
  return SS_OK;
}

// ============================================================== explicit instantiations
// ============================================================== similarity producer (the step before featurize)
// S[i][j] = sum_k min(F[i,k], F[j,k]) / sum_k max(F[i,k], F[j,k]): the weighted Jaccard (Ruzicka) similarity of
// the reference's tutorial, `1 .- pairwise(Jaccard(), X, dims=1)` (docs/src/tutorial/fishers-flowers.jl:66;
// Distances.jl: distance 0 when both rows are all zero).  F: n x d column-major; S: n x n column-major.
// 64 x 64 tile per workgroup, 4 x 4 pairs per thread, feature columns staged 16 at a time through LDS; only the
// tiles on and above the diagonal are computed, the mirror image is written with them.
template <class T>
__global__ void __launch_bounds__(256) jaccard_kernel(const T* __restrict__ F, int64_t n, int64_t d, int64_t ld,
                                                      T* __restrict__ S, int64_t lds_) {
  constexpr int TS = 64, BK = 16;
  if (blockIdx.y < blockIdx.x) return;  // lower triangle comes from the mirror write
  __shared__ T A[BK][TS + 1];
  __shared__ T B[BK][TS + 1];
  const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
  const int64_t i0 = (int64_t)blockIdx.x * TS, j0 = (int64_t)blockIdx.y * TS;
  T smin[4][4], smax[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) { smin[a][b] = T(0); smax[a][b] = T(0); }
  for (int64_t k0 = 0; k0 < d; k0 += BK) {
    for (int e = tid; e < BK * TS; e += 256) {
      const int kk = e / TS, r = e % TS;
      const int64_t k = k0 + kk;
      A[kk][r] = (k < d && i0 + r < n) ? F[i0 + r + k * ld] : T(0);
      B[kk][r] = (k < d && j0 + r < n) ? F[j0 + r + k * ld] : T(0);
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < BK; ++kk) {
      T av[4], bv[4];
#pragma unroll
      for (int a = 0; a < 4; ++a) av[a] = A[kk][tx + 16 * a];
#pragma unroll
      for (int b = 0; b < 4; ++b) bv[b] = B[kk][ty + 16 * b];
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          smin[a][b] += av[a] < bv[b] ? av[a] : bv[b];
          smax[a][b] += av[a] < bv[b] ? bv[b] : av[a];
        }
    }
    __syncthreads();
  }
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const int64_t i = i0 + tx + 16 * a, j = j0 + ty + 16 * b;
      if (i < n && j < n) {
        const T v = smax[a][b] == T(0) ? T(1) : smin[a][b] / smax[a][b];
        S[i + j * lds_] = v;
        S[j + i * lds_] = v;
      }
    }
}

template <class T>
int launch_jaccard(const T* F, int64_t n, int64_t d, int64_t ld, T* S, int64_t lds_) {
  if (n <= 0) return SS_OK;
  const unsigned g = (unsigned)ceil_div(n, 64);
  hipLaunchKernelGGL(jaccard_kernel<T>, dim3(g, g), dim3(256), 0, ctx().stream, F, n, d, ld, S, lds_);
  SS_LAUNCH_CHECK();
  return SS_OK;
}

#define SS_INSTANTIATE(T)                                                                                   \
  template int launch_cutoff<T>(const T*, int64_t, int64_t, int64_t, T, bool, T*, int64_t);                 \
  template int launch_jaccard<T>(const T*, int64_t, int64_t, int64_t, T*, int64_t);                        \
  template int launch_row_degree<T>(const T*, int64_t, int64_t, int64_t, int*);                             \
  template int launch_spread_dense<T>(const T*, int64_t, int64_t, int64_t, const int*, T*, int64_t);        \
  template int launch_transfer<T>(int, const DevCsr<T>*[2], const T*[2], const DevChunked<T>*[2], const T*, \
                                  int64_t, int64_t, int64_t, T*, int64_t, const int*, bool);                      \
  template int launch_fold_degrees<T>(const DevCsr<T>&, const DevCsr<T>&, const DevCsr<T>&, const int*,     \
                                      int64_t, int*, int*, int*);                                           \
  template int launch_fold_inverse<T>(const int*, const int*, const int*, int, int64_t, int64_t, T*, T*);   \
  template int launch_transfer_block<T>(const DevCsr<T>&, const T*, const DevChunked<T>&, const T*, int64_t, int64_t, \
                                        int64_t, T*, int64_t, T*, float, bool);                               \
  template bool transfer_block_fits<T>(int64_t, int);                                                        \
  template int launch_transfer_loo<T>(const DevCsr<T>&, const DevChunked<T>&, const int*, const int*,       \
                                      int64_t, int64_t, T*, int64_t);                                                \
  template int sell_max_chunk<T>(int);                                                                      \
  template int launch_spmm_sell<T>(const DevSell<T>&, const T*, int64_t, int64_t, T*, int64_t, const int*,  \
                                   const int*);                                                             \
  template int narrow_chunk_cols<T>(int);                                                                   \
  template int launch_spmm_chunked_narrow<T>(const DevChunked<T>&, int, const T*, int64_t, int, T*, int64_t, \
                                             DevBuf<T>&);                                                   \
  template int launch_transpose<T>(const T*, int64_t, int64_t, int64_t, T*, int64_t);                       \
  template int launch_unpermute<T>(const T*, int64_t, int64_t, int64_t, const int*, const int*, const int*, T*, \
                                   int64_t);                                                                \
  template int launch_loo_clean_fix<T>(const DevCsr<T>&, const int*, int64_t, int64_t, T*, int64_t);
SS_INSTANTIATE(float)
SS_INSTANTIATE(double)

}  // namespace ss
