// Stage 2, mid width (8 <= B <= 32 columns of R; fp64: 8 < B <= 16): F = W*R with the accumulators of a block of rows
// held in registers while the row chunks of R cycle through LDS.
//
// The narrow kernel (kernels.hip, spmm_chunked_narrow_kernel) keeps one chunk of R per workgroup and writes
// per-chunk partial sums; at B = 16 and K = 100k that is 40 chunks, i.e. 0.5 GB of partial-sum traffic next to
// 0.6 GB of W.  Here the workgroup owns rows instead: workgroup i scores rows [i*rows_per_wg, ...), walks ALL
// chunks (R is small in this regime and comes from L2: nWG * |R| bytes), and every group of GL lanes keeps
// UR rows x BV accumulators in registers from the first chunk to the last -- no partial sum leaves the chip and
// W is still streamed exactly once (6 B per non-zero, chunk-major, 16-bit local indices: the same DevChunked
// operand as the narrow kernel).
//
//  * LDS tile [KC+1][BV] (160 KB), restaged per chunk by LDS-DMA (global_load_lds: no registers, every wave keeps
//    all its pieces in flight).  Measured on MI355X: L2 -> LDS runs at about 20 B/clk per CU when all 256 CUs
//    pull at once (13 TB/s aggregate), and the issuing wave is held while its pieces are accepted, so staging and
//    gathering do not overlap inside a workgroup -- two half-size tiles (double buffering), a dedicated DMA wave
//    and a fixed-slot ELL operand were all measured slower or equal and are not kept.
//  * A non-zero fetches its BV operands as BV*sizeof(T)/16 ds_read_b128.  All lanes ask for the same 16-byte
//    column group at the same time, so with a plain layout only 256/rowbytes of the 16 slots of a 256-byte LDS
//    line would ever be hit; the tile is XOR-swizzled: column group cg of row k lives at slot
//    cg ^ ((k / rows_per_line) % groups_per_row), which spreads random k over all 16 slots.
//  * The first quads of chunk c are requested before the tile is restaged, so their latency overlaps the staging
//    (loading the offsets a chunk ahead costs 14 more registers per lane and spilled: no gain).
//  * Every sum has a fixed order (chunks ascending, entries in order, then a butterfly over the GL lanes):
//    results are bitwise reproducible.
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
  const int* off;             // [nchunks][M] (+1) in quads, chunk-major and contiguous
  const unsigned short* idx;  // quads of chunk-local indices (pad = KC -> zero tile row)
  const T* val;
  int64_t M, K;
  int KC, nchunks, B;
  const T* R;  // row-major [K][ldr]
  int64_t ldr;
  T* F;        // row-major [M][ldf]
  int64_t ldf;
  int rows_per_wg;
  int vec_ok;  // rows of R can be moved in 16-byte pieces
};

__device__ __attribute__((aligned(16))) unsigned int mid_zero[4] = {0u, 0u, 0u, 0u};

constexpr int MID_THREADS = 512;  // 8 waves: 256 VGPRs per lane for the accumulators (one workgroup per CU anyway)

template <class T, int N>
struct alignas(16) Pack {
  T v[N];
};

template <class T, int BV, int GL, int UR>
__global__ void __launch_bounds__(MID_THREADS) spmm_rowblock_kernel(MidArgs<T> a) {
  constexpr int PW = 16 / (int)sizeof(T);          // values per 16-byte slot
  constexpr int NCG = BV / PW;                     // slots per tile row (2, 4 or 8)
  constexpr int RPL = 16 / NCG;                    // tile rows per 256-byte LDS line
  constexpr int RSH = RPL == 8 ? 3 : (RPL == 4 ? 2 : (RPL == 2 ? 1 : 0));
  constexpr int ROWB = BV * (int)sizeof(T);        // bytes per tile row
  static_assert(NCG == 2 || NCG == 4 || NCG == 8 || NCG == 16, "tile row must be 32, 64, 128 or 256 bytes");
  using P = Pack<T, PW>;
  using Q = Pack<T, 4>;
  extern __shared__ __align__(16) unsigned char smem_raw[];
  unsigned char* const tb = smem_raw;

  constexpr int G = MID_THREADS / GL;  // lane groups per workgroup
  const int tid = threadIdx.x;
  const int g = tid / GL, gl = tid % GL;
  const int64_t r0 = (int64_t)blockIdx.x * a.rows_per_wg;
  const int64_t rend = (r0 + a.rows_per_wg < a.M) ? r0 + a.rows_per_wg : a.M;
  const int nu = (a.rows_per_wg + G - 1) / G;  // row sets in use (uniform over the workgroup)

  T acc[UR][BV];
#pragma unroll
  for (int u = 0; u < UR; ++u)
#pragma unroll
    for (int i = 0; i < BV; ++i) acc[u][i] = T(0);

  const ushort4* __restrict__ ip = reinterpret_cast<const ushort4*>(a.idx);
  const Q* __restrict__ vp = reinterpret_cast<const Q*>(a.val);

  auto consume = [&](int u, const ushort4 iv, const Q w) __attribute__((always_inline)) {
    const unsigned ks[4] = {iv.x, iv.y, iv.z, iv.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const unsigned k = ks[e];
      const unsigned base = k * ROWB;
      const unsigned sw = ((k >> RSH) & (NCG - 1)) << 4;
#pragma unroll
      for (int cg = 0; cg < NCG; ++cg) {
        const P r = *reinterpret_cast<const P*>(tb + base + (((unsigned)cg << 4) ^ sw));
#pragma unroll
        for (int i = 0; i < PW; ++i) acc[u][cg * PW + i] = fma(w.v[e], r.v[i], acc[u][cg * PW + i]);
      }
    }
  };

  for (int c = 0; c < a.nchunks; ++c) {
    const int64_t k0 = (int64_t)c * a.KC;
    const int kn = (int)((a.K - k0 < a.KC) ? (a.K - k0) : a.KC);
    const int* __restrict__ off = a.off + (int64_t)c * a.M;
    // first quad of every row of this group: requested before the tile is (re)staged so that the latency
    // overlaps the staging; loads are unconditional (clamped) to keep them out of branches
    int o[UR], oe[UR];
    ushort4 iv[UR];
    Q w[UR];
#pragma unroll
    for (int u = 0; u < UR; ++u) {
      const int64_t m = r0 + (int64_t)u * G + g;
      const bool ok = u < nu && m < rend;
      o[u] = ok ? off[m] : 0;
      oe[u] = ok ? off[m + 1] : 0;
    }
#pragma unroll
    for (int u = 0; u < UR; ++u) {
      const int q = o[u] + gl;
      const int qq = q < oe[u] ? q : o[u];
      iv[u] = ip[qq];
      w[u] = vp[qq];
    }

    if (c) __syncthreads();  // everybody is done with the previous tile
    // LDS-DMA: a wave instruction fills 64 consecutive pieces of the tile; the lane picks the global element
    // that belongs there (the swizzle is applied on the source side), pieces outside R read a zero word
    const unsigned char* rbase = reinterpret_cast<const unsigned char*>(a.R + k0 * a.ldr);
    const int64_t rowstride = a.ldr * (int64_t)sizeof(T);
    if (a.vec_ok) {
      const int pieces = (a.KC + 1) * NCG;  // 16-byte pieces
      const int bslots = a.B / PW;
      for (int base = (tid >> 6) * 64; base < pieces; base += MID_THREADS) {
        const int p = base + (tid & 63);
        if (p < pieces) {
          const int k = p / NCG, slot = p % NCG;
          const int cg = slot ^ ((k >> RSH) & (NCG - 1));
          const void* src = (k < kn && cg < bslots) ? (const void*)(rbase + k * rowstride + cg * 16) : (const void*)mid_zero;
          __builtin_amdgcn_global_load_lds(src, (__attribute__((address_space(3))) void*)(tb + (size_t)base * 16), 16, 0, 0);
        }
      }
    } else {
      constexpr int WPR = ROWB / 4;         // 4-byte words per tile row
      const int words = (a.KC + 1) * WPR;
      const int bwords = a.B * ((int)sizeof(T) / 4);
      for (int base = (tid >> 6) * 64; base < words; base += MID_THREADS) {
        const int p = base + (tid & 63);
        if (p < words) {
          const int k = p / WPR, wc = p % WPR;
          const int lw = ((((wc >> 2) ^ ((k >> RSH) & (NCG - 1)))) << 2) | (wc & 3);  // logical word column
          const void* src = (k < kn && lw < bwords) ? (const void*)(rbase + k * rowstride + lw * 4) : (const void*)mid_zero;
          __builtin_amdgcn_global_load_lds(src, (__attribute__((address_space(3))) void*)(tb + (size_t)base * 4), 4, 0, 0);
        }
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

#pragma unroll
    for (int u = 0; u < UR; ++u) {
      if (u < nu) {
        if (o[u] + gl < oe[u]) consume(u, iv[u], w[u]);
        for (int q = o[u] + gl + GL; q < oe[u]; q += GL) consume(u, ip[q], vp[q]);
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
int mid_chunk_cols(int bv) {
  int64_t kc = (int64_t)(160 * 1024) / ((int64_t)bv * (int64_t)sizeof(T)) - 1;
  if (kc > 65535) kc = 65535;
  return (int)kc;
}

template <class T, int BV, int GL, int UR>
static int launch_mid_variant(MidArgs<T>& a, size_t lds) {
  constexpr int G = MID_THREADS / GL;
  // one workgroup per CU when the rows allow it; a workgroup can hold G*UR rows
  int64_t rpw = ceil_div(a.M, (int64_t)ctx().num_cu);
  if (rpw > (int64_t)G * UR) rpw = (int64_t)G * UR;
  if (rpw < 1) rpw = 1;
  a.rows_per_wg = (int)rpw;
  const unsigned grid = (unsigned)ceil_div(a.M, rpw);
  static std::atomic<bool> attr_set{false};
  if (!attr_set) {
    SS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&spmm_rowblock_kernel<T, BV, GL, UR>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_set = true;
  }
  hipLaunchKernelGGL((spmm_rowblock_kernel<T, BV, GL, UR>), dim3(grid), dim3(MID_THREADS), lds, ctx().stream, a);
  SS_LAUNCH_CHECK();
  return SS_OK;
}

template <class T, int BV>
static int launch_mid_bv(MidArgs<T>& a, size_t lds, double mean_quads) {
  // accumulators per lane = UR*BV values (112..128 registers) next to the prefetched first quads; with one
  // workgroup per CU a group of GL lanes must hold ceil(M / (256 * 512/GL)) rows
  constexpr int UR = (BV * sizeof(T) == 32) ? 13 : ((BV * sizeof(T) == 64) ? 7 : 4);
  if (mean_quads > 24.0) return launch_mid_variant<T, BV, 64, UR>(a, lds);
  if (mean_quads > 12.0) return launch_mid_variant<T, BV, 16, UR>(a, lds);
  if (mean_quads > 5.0) return launch_mid_variant<T, BV, 8, UR>(a, lds);
  if (mean_quads > 2.4) return launch_mid_variant<T, BV, 4, UR>(a, lds);
  return launch_mid_variant<T, BV, 2, UR>(a, lds);
}

template <class T>
int launch_spmm_rowblock(const DevChunked<T>& W, int bv, const T* R, int64_t ldr, int B, T* F, int64_t ldf) {
  if (W.rows <= 0 || B <= 0) return SS_OK;
  if (W.align != 4) return fail(SS_EINVAL, "mid-width operand must be quad-aligned");
  path_add("spmm_rowblock");
  if (B > bv) return fail(SS_EINVAL, "B exceeds the tile width");
  if (W.SC > mid_chunk_cols<T>(bv)) return fail(SS_EINVAL, "chunk does not fit the LDS tile");
  MidArgs<T> a{};
  a.off = W.off.p; a.idx = W.idx.p; a.val = W.val.p;
  a.M = W.rows; a.K = W.cols; a.KC = W.SC; a.nchunks = W.nchunks; a.B = B;
  a.R = R; a.ldr = ldr; a.F = F; a.ldf = ldf;
  constexpr int PW = 16 / (int)sizeof(T);
  a.vec_ok = (ldr % PW == 0 && B % PW == 0 && (reinterpret_cast<uintptr_t>(R) & 15) == 0) ? 1 : 0;
  const size_t lds = (size_t)(W.SC + 1) * bv * sizeof(T);
  const double mean_quads = (double)W.stored / 4.0 / ((double)W.rows * (double)W.nchunks);
  if constexpr (sizeof(T) == 4) {
    switch (bv) {
      case 8: return launch_mid_bv<T, 8>(a, lds, mean_quads);
      case 16: return launch_mid_bv<T, 16>(a, lds, mean_quads);
      case 32: return launch_mid_bv<T, 32>(a, lds, mean_quads);
    }
  } else {
    if (bv == 16) return launch_mid_bv<T, 16>(a, lds, mean_quads);
  }
  return fail(SS_EINVAL, "mid-width tile must be 16 or 32 columns (fp64: 16)");
}

template int mid_chunk_cols<float>(int);
template int mid_chunk_cols<double>(int);
template int launch_spmm_rowblock<float>(const DevChunked<float>&, int, const float*, int64_t, int, float*, int64_t);
template int launch_spmm_rowblock<double>(const DevChunked<double>&, int, const double*, int64_t, int, double*, int64_t);

}  // namespace ss
