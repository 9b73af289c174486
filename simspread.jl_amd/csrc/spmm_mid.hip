// Stage 2, mid width (8 < B <= 64 columns of R): F = W*R with the accumulators of a block of rows held in
// registers while the row chunks of R cycle through LDS.
//
// The narrow kernel (kernels.hip, spmm_chunked_narrow_kernel) keeps one chunk of R per workgroup and writes
// per-chunk partial sums; at B = 16 and K = 100k that is 40 chunks, i.e. 0.5 GB of partial-sum traffic next to
// 0.6 GB of W.  Here the workgroup owns rows instead: workgroup i scores rows [i*rows_per_wg, ...), walks ALL
// chunks (R is small in this regime and comes from L2: nWG * |R| bytes), and every group of GL lanes keeps
// UR rows x BV accumulators in registers from the first chunk to the last -- no partial sum leaves the chip and
// W is streamed exactly once per pass of BV columns.
//
//  * Operand: row-block ELL (DevEll).  Sub-row (chunk c, row m) owns the fixed slot of GL quads
//    ((c*M + m)*GL + q), so a lane's loads depend on nothing but its own ids and are issued a whole phase
//    ahead; entries beyond 4*GL of a sub-row (under 1 % for Poisson rows) sit in an overflow CSR and are
//    applied at the end straight from global R.
//  * LDS: two tiles of [KC+1][BV/2] (32-byte rows): the left and the right half of the chunk's BV columns.
//    A phase computes one half of one chunk from one tile while the LDS-DMA (global_load_lds, no registers)
//    fills the other tile for the next phase; the quads of W stay in registers for both halves.  One barrier
//    per phase.  With B <= BV/2 there is one phase per chunk and the tiles simply alternate.
//  * A non-zero fetches its half row as two ds_read_b128.  All lanes ask for the same 16-byte slot of their row
//    at the same time, so with a plain layout only every other slot of the 256-byte LDS line would be hit;
//    slot s of row k is therefore stored at s ^ ((k >> 3) & 1), which spreads random k over all 16 slots.
//  * Every sum has a fixed order (chunks ascending, slot order, then the overflow list, then a butterfly over
//    the GL lanes): results are bitwise reproducible.
#include <type_traits>

#include "graph.hpp"

namespace ss {

#define SS_LAUNCH_CHECK()                                                             \
  do {                                                                                \
    hipError_t _e = hipGetLastError();                                                \
    if (_e != hipSuccess)                                                             \
      return fail(SS_EHIP, "%s:%d kernel launch: %s", __FILE__, __LINE__, hipGetErrorString(_e)); \
  } while (0)

template <class T>
struct MidArgs {
  const unsigned short* idx;  // [nchunks][M][GL] quads of chunk-local indices (pad = KC -> zero tile row)
  const T* val;               // same shape; unused when the operand is binary
  const int* optr;            // overflow CSR (global column indices)
  const int* oidx;
  const T* oval;
  int64_t M, K;
  int KC, nchunks, B;
  const T* R;  // row-major [K][ldr]
  int64_t ldr;
  T* F;        // row-major [M][ldf]
  int64_t ldf;
  int rows_per_wg;
  int vec_ok;  // rows of R can be moved in 16-byte pieces
  int dbg;     // ablation switches (SS_MID_DBG): 1 no restaging, 2 no compute, 4 no fetch after the first
};

__device__ __attribute__((aligned(16))) unsigned int mid_zero[4] = {0u, 0u, 0u, 0u};

constexpr int MID_THREADS = 512;  // 8 waves: 256 VGPRs per lane for accumulators + two sets of quads

// LDS-DMA piece: every active lane moves SIZE bytes from its own global address to LDS byte address
// lds_base + lane*SIZE.  Written as inline assembly on purpose: through the builtin the compiler knows that LDS
// is being written asynchronously and puts s_waitcnt vmcnt(0) in front of every later ds_read, which serialises
// the staging of the next tile with the gathers from the current one (measured: the two times simply added up).
// Completion is awaited explicitly (s_waitcnt vmcnt(0) + barrier) before the tile is read.
__device__ __forceinline__ void lds_dma16(const void* src, unsigned lds_base) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(src), "s"(lds_base) : "memory");
}
__device__ __forceinline__ void lds_dma4(const void* src, unsigned lds_base) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %0, off" ::"v"(src), "s"(lds_base) : "memory");
}

template <class T, int N>
struct alignas(16) Pack {
  T v[N];
};

template <class T, int GL, int UR, bool BIN>
__global__ void __launch_bounds__(MID_THREADS) spmm_ell_kernel(MidArgs<T> a) {
  constexpr int PW = 16 / (int)sizeof(T);  // values per 16-byte slot
  constexpr int BV = 4 * PW;               // columns per pass: 64-byte rows of R
  constexpr int HV = BV / 2;               // columns per tile
  constexpr int ROWB = 32;                 // bytes per tile row
  using P = Pack<T, PW>;
  using Q = Pack<T, 4>;
  extern __shared__ __align__(16) unsigned char smem_raw[];
  const unsigned tileb = (unsigned)(a.KC + 1) * ROWB;
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem_raw;

  constexpr int G = MID_THREADS / GL;  // lane groups per workgroup
  const int tid = threadIdx.x;
  const int g = tid / GL, gl = tid % GL;
  const int64_t r0 = (int64_t)blockIdx.x * a.rows_per_wg;
  const int64_t rend = (r0 + a.rows_per_wg < a.M) ? r0 + a.rows_per_wg : a.M;
  const int nu = (a.rows_per_wg + G - 1) / G;  // row sets in use (uniform over the workgroup)
  const int nh = a.B > HV ? 2 : 1;             // tiles per chunk

  T acc[UR][BV];
#pragma unroll
  for (int u = 0; u < UR; ++u)
#pragma unroll
    for (int i = 0; i < BV; ++i) acc[u][i] = T(0);

  const ushort4* __restrict__ ip = reinterpret_cast<const ushort4*>(a.idx);
  const Q* __restrict__ vp = reinterpret_cast<const Q*>(a.val);

  struct Head {  // the quad of each of the lane's rows in one chunk
    ushort4 iv[UR];
    Q w[UR];
  };

  auto fetch = [&](int c, Head& h) __attribute__((always_inline)) {
#pragma unroll
    for (int u = 0; u < UR; ++u) {
      if (u < nu) {
        int64_t m = r0 + (int64_t)u * G + g;
        const bool ok = m < rend;
        m = ok ? m : rend - 1;  // a valid slot; masked below
        const int64_t q = ((int64_t)c * a.M + m) * GL + gl;
        const ushort4 iv = ip[q];
        const unsigned short pad = (unsigned short)a.KC;
        h.iv[u] = ok ? iv : make_ushort4(pad, pad, pad, pad);
        if (!BIN) h.w[u] = vp[q];  // a padded index reads the zero tile row: any weight is fine
      }
    }
  };

  // phase p = (chunk c, half hf) -> tile p & 1
  auto stage = [&](int c, int hf, unsigned tb) __attribute__((always_inline)) {  // tb: LDS byte address of the tile
    const int64_t k0 = (int64_t)c * a.KC;
    const int kn = (int)((a.K - k0 < a.KC) ? (a.K - k0) : a.KC);
    const unsigned char* rbase = reinterpret_cast<const unsigned char*>(a.R + k0 * a.ldr);
    const int64_t rowstride = a.ldr * (int64_t)sizeof(T);
    if (a.vec_ok) {
      const int pieces = (a.KC + 1) * 2;  // 16-byte pieces
      const int bslots = a.B / PW;
      for (int base = (tid >> 6) * 64; base < pieces; base += MID_THREADS) {
        const int p = base + (tid & 63);
        if (p < pieces) {
          const int k = p >> 1;
          const int ls = hf * 2 + ((p & 1) ^ ((k >> 3) & 1));  // logical 16-byte slot of the 64-byte row of R
          const void* src = (k < kn && ls < bslots) ? (const void*)(rbase + k * rowstride + ls * 16) : (const void*)mid_zero;
          lds_dma16(src, __builtin_amdgcn_readfirstlane(tb + (unsigned)base * 16u));
        }
      }
    } else {
      const int words = (a.KC + 1) * 8;  // 4-byte words
      const int bwords = a.B * ((int)sizeof(T) / 4);
      for (int base = (tid >> 6) * 64; base < words; base += MID_THREADS) {
        const int p = base + (tid & 63);
        if (p < words) {
          const int k = p >> 3, wc = p & 7;
          const int ls = hf * 2 + ((wc >> 2) ^ ((k >> 3) & 1));
          const int lw = ls * 4 + (wc & 3);  // logical word column
          const void* src = (k < kn && lw < bwords) ? (const void*)(rbase + k * rowstride + lw * 4) : (const void*)mid_zero;
          lds_dma4(src, __builtin_amdgcn_readfirstlane(tb + (unsigned)base * 4u));
        }
      }
    }
  };

  // HF (which half of the BV accumulators) is a compile-time tag: accumulator indices stay static
  auto compute_half = [&](const unsigned char* tb, const Head& h, auto HF) __attribute__((always_inline)) {
    constexpr int O = decltype(HF)::value * HV;
#pragma unroll
    for (int u = 0; u < UR; ++u) {
      if (u < nu) {
        const unsigned ks[4] = {h.iv[u].x, h.iv[u].y, h.iv[u].z, h.iv[u].w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const unsigned k = ks[e];
          const unsigned base = k * ROWB;
          const unsigned sw = ((k >> 3) & 1u) << 4;
          const P ra = *reinterpret_cast<const P*>(tb + base + sw);
          const P rb = *reinterpret_cast<const P*>(tb + base + (16u ^ sw));
#pragma unroll
          for (int i = 0; i < PW; ++i) {
            acc[u][O + i] = BIN ? acc[u][O + i] + ra.v[i] : fma(h.w[u].v[e], ra.v[i], acc[u][O + i]);
            acc[u][O + PW + i] = BIN ? acc[u][O + PW + i] + rb.v[i] : fma(h.w[u].v[e], rb.v[i], acc[u][O + PW + i]);
          }
        }
      }
    }
  };
  // one chunk: its nh phases.  H holds this chunk's quads, Hn receives the next chunk's during the last phase.
  // The barrier at the head of a phase says: my pieces of this phase's tile have landed (vmcnt) and I am done
  // with the other tile, which the DMA issued right after the barrier may overwrite.  (The two halves are
  // written out one after the other: a loop over the half would turn the accumulator index into a runtime value.)
  auto chunk = [&](int c, Head& H, Head& Hn) __attribute__((always_inline)) {
    const int p = c * nh;
    unsigned char* cur = smem_raw + (p & 1) * tileb;
    unsigned char* nxt = smem_raw + ((p + 1) & 1) * tileb;
    const unsigned cur_l = lds0 + (p & 1) * tileb, nxt_l = lds0 + ((p + 1) & 1) * tileb;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (nh == 2) {
      if (!(a.dbg & 1)) stage(c, 1, nxt_l);
    } else if (c + 1 < a.nchunks) {
      if (!(a.dbg & 4)) fetch(c + 1, Hn);  // older than the DMA below: both are complete at the next barrier
      if (!(a.dbg & 1)) stage(c + 1, 0, nxt_l);
    }
    if (!(a.dbg & 2)) compute_half(cur, H, std::integral_constant<int, 0>{});
    if (nh == 2) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (c + 1 < a.nchunks) {
        if (!(a.dbg & 4)) fetch(c + 1, Hn);
        if (!(a.dbg & 1)) stage(c + 1, 0, cur_l);
      }
      if (!(a.dbg & 2)) compute_half(nxt, H, std::integral_constant<int, 1>{});
    }
  };

  Head ha, hb;
  fetch(0, ha);
  stage(0, 0, lds0);
  for (int c = 0; c < a.nchunks; c += 2) {
    chunk(c, ha, hb);
    if (c + 1 < a.nchunks) chunk(c + 1, hb, ha);
  }

  // overflow entries: operands straight from global R (rare)
#pragma unroll
  for (int u = 0; u < UR; ++u) {
    if (u < nu) {
      const int64_t m = r0 + (int64_t)u * G + g;
      if (m < rend) {
        const int xb = a.optr[m], xe = a.optr[m + 1];
        for (int x = xb + gl; x < xe; x += GL) {
          const int64_t k = a.oidx[x];
          const T v = a.oval[x];
          const T* rr = a.R + k * a.ldr;
#pragma unroll
          for (int i = 0; i < BV; ++i)
            if (i < a.B) acc[u][i] = fma(v, rr[i], acc[u][i]);
        }
      }
    }
  }

  // fold the GL lanes of a group (fixed butterfly order), then lane gl stores columns gl, gl+GL, ...
#pragma unroll
  for (int u = 0; u < UR; ++u) {
    if (u < nu) {
#pragma unroll
      for (int sft = 1; sft < GL; sft <<= 1)
#pragma unroll
        for (int i = 0; i < BV; ++i) acc[u][i] += __shfl_xor(acc[u][i], sft);
      const int64_t m = r0 + (int64_t)u * G + g;
      if (m < rend) {
#pragma unroll
        for (int i = 0; i < BV; ++i)
          if ((i % GL) == gl && i < a.B) a.F[m * a.ldf + i] = acc[u][i];
      }
    }
  }
}

template <class T>
int mid_tile_cols() {
  return 64 / (int)sizeof(T);
}

template <class T>
int mid_chunk_rows() {
  return (80 * 1024) / 32 - 1;  // two tiles of (KC + 1) 32-byte rows
}

template <class T, int GL, int UR>
static int launch_ell_variant(MidArgs<T>& a, bool binary) {
  constexpr int G = MID_THREADS / GL;
  // one workgroup per CU when the rows allow it; a workgroup can hold G*UR rows
  int64_t rpw = ceil_div(a.M, (int64_t)ctx().num_cu);
  if (rpw > (int64_t)G * UR) rpw = (int64_t)G * UR;
  if (rpw < 1) rpw = 1;
  a.rows_per_wg = (int)rpw;
  const unsigned grid = (unsigned)ceil_div(a.M, rpw);
  const size_t lds = 2 * (size_t)(a.KC + 1) * 32;
  static bool attr_set[2] = {false, false};
  if (binary) {
    if (!attr_set[0]) {
      SS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&spmm_ell_kernel<T, GL, UR, true>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      attr_set[0] = true;
    }
    hipLaunchKernelGGL((spmm_ell_kernel<T, GL, UR, true>), dim3(grid), dim3(MID_THREADS), lds, ctx().stream, a);
  } else {
    if (!attr_set[1]) {
      SS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&spmm_ell_kernel<T, GL, UR, false>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      attr_set[1] = true;
    }
    hipLaunchKernelGGL((spmm_ell_kernel<T, GL, UR, false>), dim3(grid), dim3(MID_THREADS), lds, ctx().stream, a);
  }
  SS_LAUNCH_CHECK();
  return SS_OK;
}

// F[:, 0..B) = W * R[:, 0..B) for B <= mid_tile_cols<T>() (one pass over W)
template <class T>
int launch_spmm_ell(const DevEll<T>& W, const T* R, int64_t ldr, int B, T* F, int64_t ldf) {
  if (W.rows <= 0 || B <= 0) return SS_OK;
  if (B > mid_tile_cols<T>()) return fail(SS_EINVAL, "B exceeds the tile width of the mid-width kernel");
  if (W.KC > mid_chunk_rows<T>() || W.KC < 1) return fail(SS_EINVAL, "chunk does not fit the LDS tiles");
  MidArgs<T> a{};
  a.idx = W.idx.p; a.val = W.val.p; a.optr = W.optr.p; a.oidx = W.oidx.p; a.oval = W.oval.p;
  a.M = W.rows; a.K = W.cols; a.KC = W.KC; a.nchunks = W.nchunks; a.B = B;
  a.R = R; a.ldr = ldr; a.F = F; a.ldf = ldf;
  a.dbg = getenv("SS_MID_DBG") ? atoi(getenv("SS_MID_DBG")) : 0;
  constexpr int PW = 16 / (int)sizeof(T);
  a.vec_ok = (ldr % PW == 0 && B % PW == 0 && (reinterpret_cast<uintptr_t>(R) & 15) == 0) ? 1 : 0;
  // rows a lane group must hold for one workgroup per CU: 2 at GL=2 ... 7 at GL=8 (M = 100k on 256 CUs)
  switch (W.GL) {
    case 2: return launch_ell_variant<T, 2, 2>(a, W.binary);
    case 4: return launch_ell_variant<T, 4, 4>(a, W.binary);
    case 8: return launch_ell_variant<T, 8, 7>(a, W.binary);
    case 16: return launch_ell_variant<T, 16, 7>(a, W.binary);
  }
  return fail(SS_EINVAL, "ELL slot width must be 2, 4, 8 or 16 quads");
}

template int mid_tile_cols<float>();
template int mid_tile_cols<double>();
template int mid_chunk_rows<float>();
template int mid_chunk_rows<double>();
template int launch_spmm_ell<float>(const DevEll<float>&, const float*, int64_t, int, float*, int64_t);
template int launch_spmm_ell<double>(const DevEll<double>&, const double*, int64_t, int, double*, int64_t);

}  // namespace ss
