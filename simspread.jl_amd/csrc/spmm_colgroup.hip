// Stage 2, mid width (5 <= B columns of R, B * sizeof(T) <= 256 bytes: fp32 B <= 64, fp64 B <= 32): F = W*R with a 2-D
// cut of the work and bank-conflict-free gathers.
//
// A kernel that gives every CU its own rows and makes it walk ALL chunks of R (the row-block kernel this one replaced,
// DESIGN.md 4.3) restages R 256 times (B = 64, K = 100k: 6.5 GB of L2 -> LDS traffic for a 0.85 GB problem), and lanes
// that gather rows of the tile at random 16-byte slots pay 2.5-3 LDS cycles per ds_read_b128 instead of 1.  Here
//
//  * the grid is RB row blocks x CG chunk groups: workgroup (rb, cg) owns rows_per_wg rows (accumulators in
//    registers for as many rows as the register file holds: 8 waves x NP x 16 rows) and walks only the chunks
//    c = cg, cg + CG, ...; R is restaged RB times instead of 256, and the CG partial sums per row are combined
//    in fixed order by colgroup_reduce_kernel (CG = 1: written directly).  RB and CG are chosen per launch so that
//    RB*CG fills the CUs and RB*|R| + 2*CG*|F| is smallest;
//  * four lanes share a non-zero and a wave works on 16 rows at once (lane group j = lane / 4, row = base + j).
//    The tile row of a non-zero (ROWB = 64, 128 or 256 bytes) is read as NR = ROWB/64 ds_read_b128 per lane.
//    A ds_read_b128 is served in four groups of 16 lanes, {0-3,12-15,20-27}, {4-11,16-19,28-31} and the same
//    +32 (MI355X_MICROARCH.md, LDS), i.e. lane groups {j0,j3,j5,j6} and {j1,j2,j4,j7}: in read r lane group j
//    fetches the 64-byte quarter (r ^ rot_j) of its tile row with rot_j = (j & 7) >> 1, so the four non-zeros of
//    a hardware group always sit in four different quarters of the 256-byte LDS line -- with 256-byte tile rows
//    (B = 64) every gather is conflict-free whatever the column indices are; with 128- and 64-byte rows two or four
//    tile rows share a line and the position also depends on the index (1.75 / 2.1 cycles on random indices);
//  * same chunked operand as the narrow and row-block kernels (16-bit chunk-local indices, quads, W streamed
//    once), tile restaged per chunk by LDS-DMA; every sum has a fixed order.
#include "graph.hpp"

namespace ss {

#define SS_LAUNCH_CHECK()                                                             \
  do {                                                                                \
    hipError_t _e = hipGetLastError();                                                \
    if (_e != hipSuccess)                                                             \
      return fail(SS_EHIP, "%s:%d kernel launch: %s", __FILE__, __LINE__, hipGetErrorString(_e)); \
  } while (0)

template <class T>
struct ColArgs {
  const int* off;             // [nchunks][M] (+1) in quads
  const unsigned short* idx;  // quads of chunk-local indices (pad = KC -> zero tile row)
  const T* val;
  int64_t M, K;
  int KC, nchunks, B;
  int BR;      // columns present in the R buffer (B, or the tile width when R was padded for 16-byte staging)
  const T* R;  // row-major [K][ldr]
  int64_t ldr;
  T* F;        // row-major [M][ldf]; used when CG == 1
  int64_t ldf;
  T* P;        // [CG][M][BV] partial sums when CG > 1
  int rows_per_wg, RBn, CG, np_used;
};

__device__ __attribute__((aligned(16))) unsigned int col_zero[4] = {0u, 0u, 0u, 0u};

constexpr int COL_ROWS = 16;      // rows per wave step (four lanes per row)
#ifndef COL_AHEAD_OVERRIDE
#define COL_AHEAD_OVERRIDE 0
#endif

template <class T, int N>
struct alignas(16) ColPack {
  T v[N];
};

// DPP quad_perm inside each group of four lanes: CTRL = 0x55 * e makes every lane take lane e's value (broadcast)
template <class T, int CTRL>
__device__ __forceinline__ T dpp_quad(T v) {
  if constexpr (sizeof(T) == 4) {
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), CTRL, 0xF, 0xF, true));
  } else {
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_mov_dpp((int)(b & 0xffffffffll), CTRL, 0xF, 0xF, true);
    const int hi = __builtin_amdgcn_mov_dpp((int)(b >> 32), CTRL, 0xF, 0xF, true);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
  }
}

template <class T, int BV, int NP, bool BIN, int WAVES>
__global__ void __launch_bounds__(WAVES * 64) spmm_colgroup_kernel(ColArgs<T> a) {
  constexpr int COL_THREADS = WAVES * 64;    // 16 waves: 128 VGPRs per lane, 8 waves: 256
  constexpr int PW = 16 / (int)sizeof(T);    // values per 16-byte slot
  constexpr int ROWB = BV * (int)sizeof(T);  // bytes per tile row: 64, 128 or 256
  constexpr int NR = ROWB / 64;              // ds_read_b128 per non-zero and lane
  constexpr int RSH = ROWB == 256 ? 8 : (ROWB == 128 ? 7 : 6);
  constexpr int NCG = ROWB / 16;             // 16-byte slots per tile row
  constexpr int CPL = NR * PW;               // columns per lane
  constexpr int NBQ = ROWB == 256 ? 1 : 2;   // batches of four quads (16 entries) requested ahead per row: mean sub-row 23 / 11 / 6 entries, so that the slow path below stays rare
  // row sets whose first batches are in flight (measured at 100k x 100k, 1 %: 1, 2 and 3 differ by < 5 % -- the kernel is
  // bound by instruction issue, not by the latency of the W stream)
  constexpr int AHEAD = COL_AHEAD_OVERRIDE ? COL_AHEAD_OVERRIDE : (BIN ? 2 : 1);
  static_assert(ROWB == 64 || ROWB == 128 || ROWB == 256, "tile row must be 64, 128 or 256 bytes");
  using P = ColPack<T, PW>;
  using Q = ColPack<T, 4>;
  extern __shared__ __align__(16) unsigned char smem_raw[];
  unsigned char* const tb = smem_raw;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane >> 2, cl = lane & 3;
  const int rot = ((j & 7) >> 1) & (NR - 1);
  unsigned sl[NR];  // byte offset inside the tile row of read r
#pragma unroll
  for (int r = 0; r < NR; ++r) sl[r] = (unsigned)((cl + 4 * (r ^ rot)) * 16);

  const int rb = blockIdx.x % a.RBn, cg = blockIdx.x / a.RBn;
  const int64_t r0 = (int64_t)rb * a.rows_per_wg;
  const int64_t rend = (r0 + a.rows_per_wg < a.M) ? r0 + a.rows_per_wg : a.M;
  const int64_t wrow0 = r0 + (int64_t)wave * (NP * COL_ROWS);

  T acc[NP][CPL];
#pragma unroll
  for (int p = 0; p < NP; ++p)
#pragma unroll
    for (int i = 0; i < CPL; ++i) acc[p][i] = T(0);

  const uint2* __restrict__ ip = reinterpret_cast<const uint2*>(a.idx);
  const Q* __restrict__ vp = reinterpret_cast<const Q*>(a.val);

  // one non-zero: tile row k, weight wv, into row set p
  auto entry = [&](int p, unsigned k, T wv) __attribute__((always_inline)) {
    const unsigned base = k << RSH;
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      const P t = *reinterpret_cast<const P*>(tb + base + sl[r]);
#pragma unroll
      for (int i = 0; i < PW; ++i) {
        if (BIN) acc[p][r * PW + i] += t.v[i];
        else acc[p][r * PW + i] = fma(wv, t.v[i], acc[p][r * PW + i]);
      }
    }
  };

  // LDS: the tile [KC + 1][ROWB], then two buffers for the sub-row bounds of this workgroup's rows.  The bounds
  // off[r0 .. r0 + rows_per_wg] of a chunk are read once (coalesced LDS-DMA) instead of by four lanes each, and one
  // chunk AHEAD, in the same staging phase as the tile of the chunk before: no LDS-DMA is ever in flight during a
  // gather phase (the compiler cannot tell LDS regions apart and would drain vmcnt(0), i.e. every prefetched quad, in
  // front of each LDS read while one is).
  const int obuf = (a.rows_per_wg + 1 + 3) & ~3;
  int* const offs0 = reinterpret_cast<int*>(tb + (size_t)(a.KC + 1) * ROWB);
  const int lrow0 = wave * (NP * COL_ROWS) + j;  // workgroup-local row of row set 0
  auto stage_bounds = [&](int c, int* dst) __attribute__((always_inline)) {
    const int* __restrict__ off = a.off + (int64_t)c * a.M;
    for (int base = (tid >> 6) * 64; base <= a.rows_per_wg; base += COL_THREADS) {
      const int i = base + (tid & 63);
      if (i <= a.rows_per_wg) {
        const int64_t m = (r0 + i < rend) ? r0 + i : rend;   // rows past the block: empty sub-rows
        __builtin_amdgcn_global_load_lds((const void*)(off + m), (__attribute__((address_space(3))) void*)(dst + base), 4, 0, 0);
      }
    }
  };
  if (cg < a.nchunks) stage_bounds(cg, offs0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  int it = 0;
  for (int c = cg; c < a.nchunks; c += a.CG, ++it) {
    const int64_t k0 = (int64_t)c * a.KC;
    const int kn = (int)((a.K - k0 < a.KC) ? (a.K - k0) : a.KC);
    const int* const offs = offs0 + (it & 1) * obuf;

    __syncthreads();  // everybody is done with the previous tile

    // The four lanes of a row load four CONSECUTIVE quads of its sub-row (lane cl: quad q + cl), NBQ such batches per
    // row set, AHEAD row sets before the gathers that use them: one load instruction per batch brings 16 entries per
    // row.  A quad is then broadcast inside its group of four lanes by DPP quad_perm (no LDS traffic).  Loads are
    // unconditional (clamped to a valid quad: the operand carries 64 entries of slack) so that they stay out of
    // branches.  The first AHEAD row sets are requested BEFORE the tile is restaged: their latency overlaps the staging.
    int bq[NP], be[NP];
    uint2 fi[NP][NBQ];
    Q fw[NP][NBQ];
    auto issue = [&](int p) __attribute__((always_inline)) {
      const int lr = lrow0 + p * COL_ROWS;
      bq[p] = offs[lr];
      be[p] = offs[lr + 1];
#pragma unroll
      for (int nb = 0; nb < NBQ; ++nb) {
        const int x = bq[p] + nb * 4 + cl;
        const int xx = x < be[p] ? x : bq[p];
        fi[p][nb] = ip[xx];
        if (!BIN) fw[p][nb] = vp[xx];
      }
    };
#pragma unroll
    for (int p = 0; p < AHEAD && p < NP; ++p) issue(p);

    // LDS-DMA: a wave instruction fills 64 consecutive pieces of the tile; pieces outside R read a zero word
    const unsigned char* rbase = reinterpret_cast<const unsigned char*>(a.R + k0 * a.ldr);
    const int64_t rowstride = a.ldr * (int64_t)sizeof(T);
    {
      // rows of R arrive in 16-byte pieces (the launcher pads R when B, the leading dimension or the base address do not allow it)
      const int pieces = (a.KC + 1) * NCG;
      const int bslots = a.BR / PW;
      for (int base = (tid >> 6) * 64; base < pieces; base += COL_THREADS) {
        const int pc = base + (tid & 63);
        if (pc < pieces) {
          const int k = pc / NCG, slot = pc % NCG;
          const void* src = (k < kn && slot < bslots) ? (const void*)(rbase + k * rowstride + slot * 16) : (const void*)col_zero;
          __builtin_amdgcn_global_load_lds(src, (__attribute__((address_space(3))) void*)(tb + (size_t)base * 16), 16, 0, 0);
        }
      }
    }
    if (c + a.CG < a.nchunks) stage_bounds(c + a.CG, offs0 + ((it + 1) & 1) * obuf);  // next chunk's bounds
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    // gathers of one batch of (up to four) quads held by lanes cl = 0 .. nq-1 of each group: quad e is broadcast from
    // lane e of the group by DPP quad_perm (no LDS traffic; the four copies below differ only in the lane they read).
    // No memory loads in here: the waits for the prefetched quads stay counted (a load inside would turn them into
    // vmcnt(0)).
    auto batch = [&](int p, int nq, const uint2 iv, const Q w) __attribute__((always_inline)) {
#define SS_COL_QUAD(E, CTRL)                                                                      \
      if (E < nq) {                                                                               \
        const unsigned x = (unsigned)__builtin_amdgcn_mov_dpp((int)iv.x, CTRL, 0xF, 0xF, true);   \
        const unsigned y = (unsigned)__builtin_amdgcn_mov_dpp((int)iv.y, CTRL, 0xF, 0xF, true);   \
        T w0 = T(0), w1 = T(0), w2 = T(0), w3 = T(0);                                             \
        if (!BIN) {                                                                               \
          w0 = dpp_quad<T, CTRL>(w.v[0]); w1 = dpp_quad<T, CTRL>(w.v[1]);                         \
          w2 = dpp_quad<T, CTRL>(w.v[2]); w3 = dpp_quad<T, CTRL>(w.v[3]);                         \
        }                                                                                         \
        entry(p, x & 0xffffu, w0);                                                                \
        entry(p, x >> 16, w1);                                                                    \
        /* weighted variants: keep the scheduler from gathering all four tile rows (64 registers) at once */ \
        if (!BIN) __builtin_amdgcn_sched_barrier(0);                                              \
        entry(p, y & 0xffffu, w2);                                                                \
        entry(p, y >> 16, w3);                                                                    \
      }
      SS_COL_QUAD(0, 0x00) SS_COL_QUAD(1, 0x55) SS_COL_QUAD(2, 0xAA) SS_COL_QUAD(3, 0xFF)
#undef SS_COL_QUAD
    };
    bool longer = false;  // some row of this lane group has more quads in this chunk than were prefetched
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      if (p + AHEAD < NP) issue(p + AHEAD);
      const int nquads = be[p] - bq[p];
      longer = longer || nquads > NBQ * 4;
#pragma unroll
      for (int nb = 0; nb < NBQ; ++nb) {
        int nq = nquads - nb * 4;
        nq = nq > 4 ? 4 : nq;
        batch(p, nq, fi[p][nb], fw[p][nb]);
      }
    }
    // Sub-rows longer than the prefetched batches (a few per cent of the rows on Poisson lengths): one rolled pass over
    // the row sets with its own loads, into a scratch accumulator that is then added to the row set it belongs to.
    if (__any(longer)) {
      for (int pp = 0; pp < NP; ++pp) {
        const int q = offs[lrow0 + pp * COL_ROWS], qend = offs[lrow0 + pp * COL_ROWS + 1];
        if (!__any(q + NBQ * 4 < qend)) continue;
        T keep[CPL];
#pragma unroll
        for (int i = 0; i < CPL; ++i) { keep[i] = acc[0][i]; acc[0][i] = T(0); }
        for (int x = q + NBQ * 4; x < qend; x += 4) {
          const int xx = x + cl < qend ? x + cl : x;
          const uint2 iv = ip[xx];
          Q w;
          if (!BIN) w = vp[xx];
          int nq = qend - x;
          nq = nq > 4 ? 4 : nq;
          batch(0, nq, iv, w);
        }
#pragma unroll
        for (int i = 0; i < CPL; ++i) {
          const T extra = acc[0][i];
          acc[0][i] = keep[i];
#pragma unroll
          for (int p = 0; p < NP; ++p)
            if (p == pp) acc[p][i] += extra;
        }
      }
    }
  }

  // store: read r of lane (j, cl) holds columns (r ^ rot) * 4*PW + cl*PW .. + PW of row wrow0 + p*16 + j
#pragma unroll
  for (int p = 0; p < NP; ++p) {
    {
      const int64_t m = wrow0 + p * COL_ROWS + j;
      if (m < rend) {
#pragma unroll
        for (int r = 0; r < NR; ++r) {
          const int col = (r ^ rot) * (4 * PW) + cl * PW;
          if (a.CG > 1) {
            P t;
#pragma unroll
            for (int i = 0; i < PW; ++i) t.v[i] = acc[p][r * PW + i];
            *reinterpret_cast<P*>(a.P + (((int64_t)cg * a.M + m) * BV + col)) = t;
          } else {
#pragma unroll
            for (int i = 0; i < PW; ++i)
              if (col + i < a.B) a.F[m * a.ldf + col + i] = acc[p][r * PW + i];
          }
        }
      }
    }
  }
}

// F[m][0..B) = sum over the CG partial sums, in chunk-group order.  One thread per 16-byte piece of a row (BV is a
// multiple of the piece), 32-bit index arithmetic (M * BV < 2^31 is checked by the launcher; else the 64-bit variant).
template <class T, class I>
__global__ void colgroup_reduce_kernel(const T* __restrict__ P, int CG, int64_t M, int BV, int B, T* __restrict__ F,
                                       int64_t ldf) {
  constexpr int PW = 16 / (int)sizeof(T);
  using P4 = ColPack<T, PW>;
  const I ppr = (I)(BV / PW);                 // pieces per row
  const I total = (I)M * ppr;
  const I stride = (I)gridDim.x * blockDim.x;
  const int64_t plane = M * (int64_t)BV;
  for (I i = (I)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const I m = i / ppr;
    const int b = (int)(i - m * ppr) * PW;
    if (b >= B) continue;
    P4 s = *reinterpret_cast<const P4*>(P + (int64_t)i * PW);
    for (int c = 1; c < CG; ++c) {
      const P4 t = *reinterpret_cast<const P4*>(P + (int64_t)c * plane + (int64_t)i * PW);
#pragma unroll
      for (int e = 0; e < PW; ++e) s.v[e] += t.v[e];
    }
    T* f = F + (int64_t)m * ldf + b;
#pragma unroll
    for (int e = 0; e < PW; ++e)
      if (b + e < B) f[e] = s.v[e];
  }
}

// R rows that cannot be moved in 16-byte pieces (B or the leading dimension no multiple of 16 bytes, or a misaligned
// base): one pass copies them into [K][BV] rows padded with zeros -- K*BV values, against the 4x as many 4-byte LDS-DMA
// instructions of every restaging otherwise (measured, fp32 B = 9: 0.27 -> 0.23 ms)
template <class T>
__global__ void colgroup_pad_rows_kernel(const T* __restrict__ R, int64_t ldr, int64_t K, int B, int BV, T* __restrict__ out) {
  const int64_t total = K * BV;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t k = i / BV;
    const int b = (int)(i - k * BV);
    out[i] = b < B ? R[k * ldr + b] : T(0);
  }
}

template <class T, int BV, int NP, bool BIN, int WAVES>
static int launch_col_variant(const ColArgs<T>& a, unsigned grid, size_t lds) {
  static std::atomic<bool> attr_set{false};
  if (!attr_set) {
    SS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&spmm_colgroup_kernel<T, BV, NP, BIN, WAVES>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_set = true;
  }
  hipLaunchKernelGGL((spmm_colgroup_kernel<T, BV, NP, BIN, WAVES>), dim3(grid), dim3(WAVES * 64), lds, ctx().stream, a);
  SS_LAUNCH_CHECK();
  return SS_OK;
}

// Waves per workgroup and row sets per wave (16 rows each).  Rows per workgroup = WAVES * NP * 16 = 2048 (64-byte tile
// rows: 16 waves x 8), 2112 (128-byte rows: 12 waves x 11) or 1280 (256-byte rows: 8 waves x 10): NP * ROWB/16
// accumulator registers per lane next to the quads in flight, the gathered tile rows and addresses, inside the
// 128 / 168 / 256 registers a lane may use at 16 / 12 / 8 waves per CU.  More waves hide the latencies better
// (measured, B = 16 pattern-only: 0.22 ms with 16 waves vs 0.29 ms with 8) where the accumulators fit.
constexpr int col_waves(int rowb) { return rowb == 256 ? 8 : (rowb == 128 ? 12 : 16); }
constexpr int col_np(int rowb, bool bin = true, int elem = 4) {
  return rowb == 256 ? ((!bin && elem == 8) ? 8 : 10) : (rowb == 128 ? 11 : 8);
}
constexpr int col_max_rows(int rowb) { return col_waves(rowb) * col_np(rowb) * COL_ROWS; }

// LDS = tile (KC + 1 rows) + two buffers of sub-row bounds of the workgroup's rows (col_max_rows + 1 ints each)
template <class T>
int colgroup_chunk_cols(int bv) {
  const int rowb = bv * (int)sizeof(T);
  const int64_t bounds = 2 * (((int64_t)(col_max_rows(rowb) + 1) * 4 + 15) / 16 * 16);
  int64_t kc = ((int64_t)(160 * 1024) - bounds) / rowb - 1;
  if (kc > 65535) kc = 65535;
  return (int)kc;
}

template <class T, int BV, bool BIN>
struct ColNP {
  static constexpr int value = col_np(BV * (int)sizeof(T), BIN, (int)sizeof(T));
};

template <class T>
int launch_spmm_colgroup(const DevChunked<T>& W, int bv, const T* R, int64_t ldr, int B, T* F, int64_t ldf,
                         DevBuf<T>& partial) {
  if (W.rows <= 0 || B <= 0) return SS_OK;
  if (W.align != 4) return fail(SS_EINVAL, "mid-width operand must be quad-aligned");
  path_add("spmm_colgroup");
  if (B > bv) return fail(SS_EINVAL, "B exceeds the tile width");
  if (W.SC > colgroup_chunk_cols<T>(bv)) return fail(SS_EINVAL, "chunk does not fit the LDS tile");
  const int rowb = bv * (int)sizeof(T);
  if (rowb != 64 && rowb != 128 && rowb != 256) return fail(SS_EINVAL, "tile row must be 64, 128 or 256 bytes");
  ColArgs<T> a{};
  a.off = W.off.p; a.idx = W.idx.p; a.val = W.val.p;
  a.M = W.rows; a.K = W.cols; a.KC = W.SC; a.nchunks = W.nchunks; a.B = B;
  a.R = R; a.ldr = ldr; a.F = F; a.ldf = ldf; a.BR = B;
  constexpr int PW = 16 / (int)sizeof(T);
  const bool vec_ok = (ldr % PW == 0 && B % PW == 0 && (reinterpret_cast<uintptr_t>(R) & 15) == 0);

  // the cut: RB row blocks x CG chunk groups.  A workgroup holds 8 waves * NP * 16 rows (accumulators in registers);
  // RB follows from the row count, CG fills the CUs with one resident round (more chunk groups = more partial sums,
  // fewer = idle CUs; SS_COL_CG overrides)
  const int np = col_np(rowb, W.binary, (int)sizeof(T));
  a.np_used = np;
  a.rows_per_wg = np * col_waves(rowb) * COL_ROWS;
  a.RBn = (int)ceil_div(W.rows, (int64_t)a.rows_per_wg);
  int cgn = ctx().num_cu / a.RBn;
  if (cgn < 1) cgn = 1;
  if (cgn > W.nchunks) cgn = W.nchunks;
  if (cgn > 16) cgn = 16;
  if (const char* e = getenv("SS_COL_CG")) {
    const int v = atoi(e);
    if (v >= 1 && v <= W.nchunks) cgn = v;
  }
  a.CG = cgn;
  a.P = nullptr;
  // scratch: the CG partial sums, then (rows of R not movable in 16-byte pieces) the padded copy of R
  const size_t need_p = a.CG > 1 ? (size_t)a.CG * (size_t)W.rows * (size_t)bv : 0;
  const size_t need_r = vec_ok ? 0 : (size_t)W.cols * (size_t)bv;
  if (partial.n < need_p + need_r) SS_TRY(partial.alloc(need_p + need_r));
  if (a.CG > 1) a.P = partial.p;
  if (!vec_ok) {
    T* rp = partial.p + need_p;   // 16-byte aligned: need_p is a multiple of bv values
    if (W.cols > 0) {
      int64_t g = ceil_div(W.cols * (int64_t)bv, 256);
      if (g > 256 * 16) g = 256 * 16;
      hipLaunchKernelGGL(colgroup_pad_rows_kernel<T>, dim3((unsigned)g), dim3(256), 0, ctx().stream, R, ldr, W.cols, B, bv, rp);
      SS_LAUNCH_CHECK();
    }
    a.R = rp; a.ldr = bv; a.BR = bv;
  }
  const unsigned grid = (unsigned)(a.RBn * a.CG);
  if (getenv("SS_COL_DEBUG"))
    fprintf(stderr, "colgroup: rowb %d KC %d chunks %d RB %d CG %d np %d rows/wg %d grid %u\n", rowb, a.KC, a.nchunks, a.RBn,
            a.CG, a.np_used, a.rows_per_wg, grid);
  const size_t lds = (size_t)(W.SC + 1) * rowb + 2 * (((size_t)(a.rows_per_wg + 1) * 4 + 15) / 16 * 16);
  int rc;
#define SS_COL(BVV)                                                                                      \
  (W.binary ? launch_col_variant<T, BVV, ColNP<T, BVV, true>::value, true, col_waves(BVV * (int)sizeof(T))>(a, grid, lds)   \
            : launch_col_variant<T, BVV, ColNP<T, BVV, false>::value, false, col_waves(BVV * (int)sizeof(T))>(a, grid, lds))
  if constexpr (sizeof(T) == 4) {
    switch (bv) {
      case 16: rc = SS_COL(16); break;
      case 32: rc = SS_COL(32); break;
      case 64: rc = SS_COL(64); break;
      default: return fail(SS_EINVAL, "mid-width tile must be 16, 32 or 64 columns");
    }
  } else {
    switch (bv) {
      case 8: rc = SS_COL(8); break;
      case 16: rc = SS_COL(16); break;
      case 32: rc = SS_COL(32); break;
      default: return fail(SS_EINVAL, "mid-width tile must be 8, 16 or 32 columns (fp64)");
    }
  }
#undef SS_COL
  SS_TRY(rc);
  if (a.CG > 1) {
    const int64_t pieces = W.rows * (int64_t)(bv / PW);
    int64_t g = ceil_div(pieces, 256);
    if (g > 256 * 16) g = 256 * 16;
    if (pieces < (1LL << 31))
      hipLaunchKernelGGL((colgroup_reduce_kernel<T, unsigned>), dim3((unsigned)g), dim3(256), 0, ctx().stream, a.P, a.CG,
                         W.rows, bv, B, F, ldf);
    else
      hipLaunchKernelGGL((colgroup_reduce_kernel<T, int64_t>), dim3((unsigned)g), dim3(256), 0, ctx().stream, a.P, a.CG,
                         W.rows, bv, B, F, ldf);
    SS_LAUNCH_CHECK();
  }
  return SS_OK;
}

template int colgroup_chunk_cols<float>(int);
template int colgroup_chunk_cols<double>(int);
template int launch_spmm_colgroup<float>(const DevChunked<float>&, int, const float*, int64_t, int, float*, int64_t,
                                         DevBuf<float>&);
template int launch_spmm_colgroup<double>(const DevChunked<double>&, int, const double*, int64_t, int, double*, int64_t,
                                          DevBuf<double>&);

}  // namespace ss
