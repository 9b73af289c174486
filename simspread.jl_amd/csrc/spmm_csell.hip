// Mid-width F = W*R (5 <= B columns, B * sizeof(T) <= 256 bytes: fp32 B <= 64, fp64 B <= 32), row-major R and F: ONE LANE PER
// ROW on the compact sliced-ELL operand (DevCsell, graph.hpp; built by csell_build, assemble.hip).  Round 3; the 2-D kernel
// (spmm_colgroup.hip) stays for matrices the operand cannot hold and under SS_CSELL=0.  Below, QT = columns per tile row,
// ROWB = QT * sizeof(T) = 64, 128 or 256 bytes (fp32 also 32: B <= 8), a "piece" = 16 bytes of a tile row.
//
// Why another kernel: in the 2-D kernel four lanes share a non-zero, so every entry costs a DPP broadcast, an index
// extraction and one address per ds_read_b128 next to its FMAs, 16 rows advance in lockstep quad by quad, and each
// (row set, chunk) starts with a dependent chain bounds -> quads -> gathers that two to four waves per SIMD do not hide.
// Here
//  * a lane owns a row and keeps its QT accumulators; it reads the whole tile row of an entry as QT/4 ds_read_b128,
//    piece r ^ (lane & (QT/4 - 1)) in read r.  With 256-byte tile rows (QT = 64) the 16 lanes the hardware serves together
//    (MI355X_MICROARCH.md, LDS) then always hit 16 different 16-byte slots of the line: no bank conflicts whatever the
//    indices are.  With 128- / 64-byte rows two / four tile rows share a line and the operand's entry order (chosen at
//    build time) keeps lanes that read the same slot number on tile rows of different classes;
//  * the stream of a wave is ONE software pipeline across its blocks and across the restaging of the tile: two rings of D
//    steps (a step = one pair of entries per lane: one 4-byte and one 8-byte coalesced load), one being multiplied while
//    the other is in flight; the last group of a block requests the first group of the next block, whose descriptors
//    (pairs per lane, first pair, steps) were loaded two blocks earlier.  No load sits in a branch;
//  * the grid is RB row blocks x CG chunk groups as in the 2-D kernel (accumulators in registers for 16 waves x NPS x 64
//    rows; R restaged RB times; CG partial sums per row combined in fixed order by csell_reduce_kernel).
// Every sum has a fixed order (entry order of the operand, then chunk order, then chunk-group order).
#include "graph.hpp"

namespace ss {

#define SS_LAUNCH_CHECK()                                                             \
  do {                                                                                \
    hipError_t _e = hipGetLastError();                                                \
    if (_e != hipSuccess)                                                             \
      return fail(SS_EHIP, "%s:%d kernel launch: %s", __FILE__, __LINE__, hipGetErrorString(_e)); \
  } while (0)

template <class T>
struct CsArgs {
  const int2* desc;              // [nblocks + 1] {first pair, steps}; the last entry is an empty block
  const unsigned short* np;      // [(nblocks + 1) * 64] pairs per lane
  const unsigned* pidx;          // per pair: two 16-bit chunk-local indices
  const T* pval;                 // per pair: two values (nullptr: pattern-only)
  int64_t M, K;
  int KC, nchunks, S;
  const T* R;                    // row-major [K][ldr], rows movable in 16-byte pieces
  int64_t ldr;
  T* F;                          // row-major [M][ldf], used when CG == 1
  int64_t ldf;
  T* P;                          // [CG][M][QT] partial sums when CG > 1
  int B;                         // columns of F
  int fvec;                      // F rows can be stored in 16-byte pieces
  int spw, RBn, CG;              // slices per workgroup, row blocks, chunk groups
};

__device__ __attribute__((aligned(16))) unsigned int csell_zero[4] = {0u, 0u, 0u, 0u};


constexpr int CSELL_WAVES = 16;
constexpr int csell_nps(int rowb) { return 256 / rowb; }   // row sets (64 rows each) per wave: 64 accumulator registers per lane

template <class T, int QT, bool BIN, int D>
__global__ void __launch_bounds__(CSELL_WAVES * 64) spmm_csell_kernel(CsArgs<T> a) {
  constexpr int ROWB = QT * (int)sizeof(T), NPC = ROWB / 16;
  constexpr int NPS = csell_nps(ROWB);
  constexpr int PW = 16 / (int)sizeof(T);    // values per piece
  typedef T f4 __attribute__((ext_vector_type(PW)));   // one piece
  typedef T t2 __attribute__((ext_vector_type(2)));    // the two values of a pair
  constexpr int NB = (ROWB == 128 || ROWB <= 32) ? 2 : 4;   // LDS reads per batch: one batch in flight while the one before is multiplied
  constexpr int RSH = ROWB == 256 ? 8 : (ROWB == 128 ? 7 : (ROWB == 64 ? 6 : (ROWB == 32 ? 5 : 4)));
  extern __shared__ __align__(16) unsigned char tb[];   // the tile [KC + 1][ROWB] at LDS offset 0 (no static LDS here)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int rb = blockIdx.x % a.RBn, cg = blockIdx.x / a.RBn;
  const unsigned lanebits = (unsigned)(lane & (NPC - 1)) << 4;
  const int nblk = a.nchunks * a.S;

  f4 acc[NPS][NPC];
#pragma unroll
  for (int p = 0; p < NPS; ++p)
#pragma unroll
    for (int r = 0; r < NPC; ++r) acc[p][r] = f4(T(0));

  // the block k places after (chunk c, row set p) in this wave's order; past the end: the empty block
  auto bid = [&](int c, int p, int k) __attribute__((always_inline)) -> int {
    const int pp = p + k;
    const int cc = c + a.CG * (pp / NPS), q = pp % NPS;
    const int sloc = wave + CSELL_WAVES * q, sl = rb * a.spw + sloc;
    return (cc < a.nchunks && sloc < a.spw && sl < a.S) ? cc * a.S + sl : nblk;
  };
  int np_cur, np_n1, np_n2;
  int cur_cur, cur_n1, cur_n2, nst_cur, nst_n1, nst_n2;
  auto load_desc = [&](int b, int& np, int& cur, int& nst) __attribute__((always_inline)) {
    const int2 d = a.desc[b];
    cur = __builtin_amdgcn_readfirstlane(d.x);
    nst = __builtin_amdgcn_readfirstlane(d.y);
    np = (int)a.np[(int64_t)b * 64 + lane];
  };
  load_desc(bid(cg, 0, 0), np_cur, cur_cur, nst_cur);
  load_desc(bid(cg, 0, 1), np_n1, cur_n1, nst_n1);
  load_desc(bid(cg, 0, 2), np_n2, cur_n2, nst_n2);

  unsigned piA[D], piB[D];
  t2 pvA[D], pvB[D];
  // request steps v0 .. v0 + D - 1 of the block with npx pairs per lane whose next pair is curx.  Lanes without a pair in
  // a step request the step's first pair (always a valid address: the arrays carry 64 pairs of slack) and ignore it.
  auto issue_group = [&](unsigned (&pi)[D], t2 (&pv)[D], int npx, int& curx, int v0) __attribute__((always_inline)) {
#pragma unroll
    for (int d = 0; d < D; ++d) {
      const bool act = npx > v0 + d;
      const unsigned long long mask = __ballot(act);
      const int rank = __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
      const int pos = curx + (act ? rank : 0);
      pi[d] = a.pidx[pos];
      if (!BIN) pv[d] = reinterpret_cast<const t2*>(a.pval)[pos];
      curx += __builtin_popcountll(mask);
    }
  };
  issue_group(piA, pvA, np_cur, cur_cur, 0);

  for (int c = cg; c < a.nchunks; c += a.CG) {
    const int64_t k0 = (int64_t)c * a.KC;
    const int kn = (int)((a.K - k0 < a.KC) ? (a.K - k0) : a.KC);
    __syncthreads();  // everybody is done with the previous tile
    {
      // LDS-DMA: a wave instruction fills 64 consecutive 16-byte pieces of the tile; pieces outside R read a zero word
      const unsigned char* rbase = reinterpret_cast<const unsigned char*>(a.R + k0 * a.ldr);
      const int64_t rowstride = a.ldr * (int64_t)sizeof(T);
      const int pieces = (a.KC + 1) * NPC;
      for (int base = (tid >> 6) * 64; base < pieces; base += CSELL_WAVES * 64) {
        const int pc = base + (tid & 63);
        if (pc < pieces) {
          const int k = pc / NPC, slot = pc % NPC;
          const void* src = (k < kn) ? (const void*)(rbase + k * rowstride + slot * 16) : (const void*)csell_zero;
          __builtin_amdgcn_global_load_lds(src, (__attribute__((address_space(3))) void*)(tb + (size_t)base * 16), 16, 0, 0);
        }
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

#pragma unroll
    for (int p = 0; p < NPS; ++p) {
      // multiply steps u0 .. u0 + D - 1 of the current block (those it has); no memory loads in here
      auto consume_group = [&](const unsigned (&pi)[D], const t2 (&pv)[D], int u0) __attribute__((always_inline)) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
          if (u0 + d < nst_cur) {   // wave-uniform
            // lanes without this pair sit the step out (EXEC): reading the zero row instead would cost them nothing, but
            // the zero row has a bank class too and they would collide with the active lanes of their slot number
            if (np_cur <= u0 + d) continue;
            const unsigned x = pi[d];
            const unsigned ka = x & 0xffffu, kb = x >> 16;
            const unsigned basea = (ka << RSH) | lanebits, baseb = (kb << RSH) | lanebits;
            const T wa = BIN ? T(1) : pv[d].x, wb = BIN ? T(1) : pv[d].y;
            constexpr int NIT = 2 * NPC, NBAT = NIT / NB;
            f4 t[2][NB];
            auto loads = [&](int bt, int buf) __attribute__((always_inline)) {
#pragma unroll
              for (int i = 0; i < NB; ++i) {
                const int it = bt * NB + i, r = it % NPC;
                const unsigned base = it < NPC ? basea : baseb;
                // integer LDS address: the tile starts at offset 0, nothing to add
                t[buf][i] = *(const __attribute__((address_space(3))) f4*)(uintptr_t)(base ^ (unsigned)(r << 4));
              }
            };
            auto fmas = [&](int bt, int buf) __attribute__((always_inline)) {
#pragma unroll
              for (int i = 0; i < NB; ++i) {
                const int it = bt * NB + i, r = it % NPC;
                const T w = it < NPC ? wa : wb;
                if (BIN) acc[p][r] += t[buf][i];
                else acc[p][r] = __builtin_elementwise_fma(f4(w), t[buf][i], acc[p][r]);
              }
            };
            loads(0, 0);
#pragma unroll
            for (int bt = 0; bt < NBAT; ++bt) {
              __builtin_amdgcn_sched_barrier(0);
              if (bt + 1 < NBAT) loads(bt + 1, (bt + 1) & 1);
              fmas(bt, bt & 1);
            }
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      };
      int ng = (nst_cur + 2 * D - 1) / (2 * D) * 2;   // groups of this block: even, at least two (ring A holds group 0)
      ng = ng < 2 ? 2 : ng;
      for (int g = 0; g < ng; g += 2) {
        issue_group(piB, pvB, np_cur, cur_cur, (g + 1) * D);
        consume_group(piA, pvA, g * D);
        const bool tail = g + 2 >= ng;                 // the next group 0 belongs to the next block
        int curx = tail ? cur_n1 : cur_cur;
        issue_group(piA, pvA, tail ? np_n1 : np_cur, curx, tail ? 0 : (g + 2) * D);
        cur_n1 = tail ? curx : cur_n1;
        cur_cur = tail ? cur_cur : curx;
        consume_group(piB, pvB, (g + 1) * D);
      }
      np_cur = np_n1; cur_cur = cur_n1; nst_cur = nst_n1;
      np_n1 = np_n2; cur_n1 = cur_n2; nst_n1 = nst_n2;
      load_desc(bid(c, p, 3), np_n2, cur_n2, nst_n2);
    }
  }

  // store: accumulator r of lane l holds piece r ^ (l & (NPC - 1)) of row slice*64 + l
#pragma unroll
  for (int p = 0; p < NPS; ++p) {
    const int sloc = wave + p * CSELL_WAVES;
    const int sl = rb * a.spw + sloc;
    if (sloc >= a.spw || sl >= a.S) continue;
    const int64_t m = (int64_t)sl * 64 + lane;
    if (m < a.M) {
#pragma unroll
      for (int r = 0; r < NPC; ++r) {
        const int col = (r ^ (lane & (NPC - 1))) * PW;
        if (a.CG > 1) {
          *reinterpret_cast<f4*>(a.P + (((int64_t)cg * a.M + m) * QT + col)) = acc[p][r];
        } else if (a.fvec) {
          if (col < a.B) *reinterpret_cast<f4*>(a.F + m * a.ldf + col) = acc[p][r];
        } else {
#pragma unroll
          for (int i = 0; i < PW; ++i)
            if (col + i < a.B) a.F[m * a.ldf + col + i] = acc[p][r][i];
        }
      }
    }
  }
}

// F[m][0..B) = sum of the CG partial sums in chunk-group order; one thread per 16-byte piece of a row
template <class T>
__global__ void csell_reduce_kernel(const T* __restrict__ P, int CG, int64_t M, int QT, int B, int fvec,
                                    T* __restrict__ F, int64_t ldf) {
  constexpr int PW = 16 / (int)sizeof(T);
  typedef T f4 __attribute__((ext_vector_type(PW)));
  const int ppr = QT / PW;
  const int64_t total = M * ppr;
  const int64_t plane = M * (int64_t)QT;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t m = i / ppr;
    const int b = (int)(i - m * ppr) * PW;
    if (b >= B) continue;
    f4 s = *reinterpret_cast<const f4*>(P + i * PW);
    for (int c = 1; c < CG; ++c) s += *reinterpret_cast<const f4*>(P + c * plane + i * PW);
    T* f = F + m * ldf + b;
    if (fvec) {
      *reinterpret_cast<f4*>(f) = s;
    } else {
#pragma unroll
      for (int e = 0; e < PW; ++e)
        if (b + e < B) f[e] = s[e];
    }
  }
}

// R rows that cannot be moved in 16-byte pieces, or narrower than the tile: one pass copies them into [K][QT] rows padded
// with zeros
template <class T>
__global__ void csell_pad_rows_kernel(const T* __restrict__ R, int64_t ldr, int64_t K, int B, int QT, T* __restrict__ out) {
  const int64_t total = K * QT;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t k = i / QT;
    const int b = (int)(i - k * QT);
    out[i] = b < B ? R[k * ldr + b] : T(0);
  }
}

int csell_chunk_cols(int rowb) {
  int64_t kc = (int64_t)(160 * 1024) / rowb - 1;   // + the zero row
  if (kc > 32767) kc = 32767;
  return (int)kc;
}

template <class T, int QT, bool BIN>
static int launch_csell_variant(const CsArgs<T>& a, unsigned grid, size_t lds) {
  static std::atomic<bool> attr_set{false};
  if (!attr_set) {
    SS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&spmm_csell_kernel<T, QT, BIN, 2>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    // the integer LDS addresses of the kernel assume the dynamic tile starts at offset 0: no static LDS may appear in it
    hipFuncAttributes fa;
    SS_HIP(hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(&spmm_csell_kernel<T, QT, BIN, 2>)));
    if (fa.sharedSizeBytes != 0) return fail(SS_EUNSUPPORTED, "spmm_csell: kernel image carries static LDS");
    attr_set = true;
  }
  hipLaunchKernelGGL((spmm_csell_kernel<T, QT, BIN, 2>), dim3(grid), dim3(CSELL_WAVES * 64), lds, ctx().stream, a);
  SS_LAUNCH_CHECK();
  return SS_OK;
}

template <class T>
int launch_spmm_csell(const DevCsell<T>& W, const T* R, int64_t ldr, int B, T* F, int64_t ldf, DevBuf<T>& partial) {
  if (W.rows <= 0 || B <= 0) return SS_OK;
  if (!W.ok) return fail(SS_EINVAL, "compact sliced ELL operand was not built");
  const int QT = W.QT;
  const int rowb = QT * (int)sizeof(T);
  constexpr int PW = 16 / (int)sizeof(T);
  if (B > QT) return fail(SS_EINVAL, "B exceeds the tile width");
  if (W.KC > csell_chunk_cols(rowb)) return fail(SS_EINVAL, "chunk does not fit the LDS tile");
  path_add("spmm_csell");
  CsArgs<T> a{};
  a.desc = reinterpret_cast<const int2*>(W.desc.p);
  a.np = W.np.p;
  a.pidx = W.pidx.p;
  a.pval = W.binary ? nullptr : W.pval.p;
  a.M = W.rows; a.K = W.cols; a.KC = W.KC; a.nchunks = W.nchunks; a.S = W.nslices;
  a.R = R; a.ldr = ldr; a.F = F; a.ldf = ldf; a.B = B;
  a.fvec = (B % PW == 0 && ldf % PW == 0 && (reinterpret_cast<uintptr_t>(F) & 15) == 0) ? 1 : 0;
  const bool rvec = (B == QT && ldr % PW == 0 && (reinterpret_cast<uintptr_t>(R) & 15) == 0);

  // the cut: every workgroup holds up to 16 waves x NPS slices; RB x CG workgroups in whole rounds of the CUs:
  // cost = rounds x (blocks + restagings of the busiest CU) + the partial sums
  const int nps = csell_nps(rowb), max_spw = CSELL_WAVES * nps;
  const int ncu = ctx().num_cu > 0 ? ctx().num_cu : 256;
  int64_t best = -1;
  int spw = max_spw, rbn = (int)ceil_div(W.nslices, (int64_t)max_spw), cgn = 1;
  for (int s = 1; s <= max_spw; ++s) {
    const int64_t rb = ceil_div(W.nslices, (int64_t)s);
    for (int cg = 1; cg <= 16 && cg <= W.nchunks; ++cg) {
      const int64_t rounds = ceil_div(rb * cg, (int64_t)ncu);
      const int64_t per = ceil_div(W.nchunks, (int64_t)cg);
      const int64_t cost = rounds * per * (s + 40) + 2 * cg;
      if (best < 0 || cost < best) { best = cost; spw = s; rbn = (int)rb; cgn = cg; }
    }
  }
  if (const char* e = getenv("SS_CSELL_CUT")) {   // "slices per workgroup,chunk groups" (comparisons)
    int s = 0, cg = 0;
    if (sscanf(e, "%d,%d", &s, &cg) == 2 && s >= 1 && s <= max_spw && cg >= 1 && cg <= W.nchunks) {
      spw = s; cgn = cg; rbn = (int)ceil_div(W.nslices, (int64_t)s);
    }
  }
  a.spw = spw; a.RBn = rbn; a.CG = cgn;
  const size_t need_p = a.CG > 1 ? (size_t)a.CG * (size_t)W.rows * (size_t)QT : 0;
  const size_t need_r = rvec ? 0 : (size_t)W.cols * (size_t)QT;
  if (partial.n < need_p + need_r) SS_TRY(partial.alloc(need_p + need_r));
  a.P = a.CG > 1 ? partial.p : nullptr;
  if (!rvec) {
    T* rp = partial.p + need_p;   // 16-byte aligned: need_p is a multiple of QT values
    if (W.cols > 0) {
      int64_t g = ceil_div(W.cols * (int64_t)QT, 256);
      if (g > 256 * 16) g = 256 * 16;
      hipLaunchKernelGGL(csell_pad_rows_kernel<T>, dim3((unsigned)g), dim3(256), 0, ctx().stream, R, ldr, W.cols, B, QT, rp);
      SS_LAUNCH_CHECK();
    }
    a.R = rp; a.ldr = QT;
  }
  const int64_t grid = (int64_t)a.RBn * a.CG;
  if (grid >= (1LL << 31)) return fail(SS_EUNSUPPORTED, "spmm_csell: grid too large");
  if (getenv("SS_COL_DEBUG"))
    fprintf(stderr, "csell: rowb %d QT %d KC %d chunks %d slices %d spw %d RB %d CG %d grid %lld\n", rowb, QT, a.KC, a.nchunks,
            a.S, a.spw, a.RBn, a.CG, (long long)grid);
  const size_t lds = (size_t)(W.KC + 1) * rowb;
  int rc;
#define SS_CS(QTV) (W.binary ? launch_csell_variant<T, QTV, true>(a, (unsigned)grid, lds) : launch_csell_variant<T, QTV, false>(a, (unsigned)grid, lds))
  if constexpr (sizeof(T) == 4) {
    switch (QT) {
      case 4: rc = SS_CS(4); break;
      case 8: rc = SS_CS(8); break;
      case 16: rc = SS_CS(16); break;
      case 32: rc = SS_CS(32); break;
      case 64: rc = SS_CS(64); break;
      default: return fail(SS_EINVAL, "spmm_csell: tile width must be 4, 8, 16, 32 or 64 (fp32)");
    }
  } else {
    switch (QT) {
      case 2: rc = SS_CS(2); break;
      case 4: rc = SS_CS(4); break;
      case 8: rc = SS_CS(8); break;
      case 16: rc = SS_CS(16); break;
      case 32: rc = SS_CS(32); break;
      default: return fail(SS_EINVAL, "spmm_csell: tile width must be 2, 4, 8, 16 or 32 (fp64)");
    }
  }
#undef SS_CS
  SS_TRY(rc);
  if (a.CG > 1) {
    int64_t g = ceil_div(W.rows * (int64_t)(QT / PW), 256);
    if (g > 256 * 16) g = 256 * 16;
    hipLaunchKernelGGL(csell_reduce_kernel<T>, dim3((unsigned)g), dim3(256), 0, ctx().stream, (const T*)a.P, a.CG, W.rows, QT, B,
                       a.fvec, F, ldf);
    SS_LAUNCH_CHECK();
  }
  return SS_OK;
}

template int launch_spmm_csell<float>(const DevCsell<float>&, const float*, int64_t, int, float*, int64_t, DevBuf<float>&);
template int launch_spmm_csell<double>(const DevCsell<double>&, const double*, int64_t, int, double*, int64_t, DevBuf<double>&);

}  // namespace ss
