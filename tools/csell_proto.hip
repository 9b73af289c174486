// Prototype (not part of the library): mid-width W*R with ONE LANE PER ROW and a compact sliced-ELL operand per chunk.
//
// The 2-D kernel of the library (spmm_colgroup.hip) spends most of its vector instructions next to the FMAs: four lanes
// share a non-zero, so every entry costs a DPP broadcast, an index extraction and one address per ds_read_b128, and 16
// rows advance in lockstep quad by quad (slots executed / non-zeros = 1.4 at B = 16, 2.0 at B = 64).  Here a lane owns a
// row: it reads the whole tile row of its entry (QT/4 ds_read_b128, piece r ^ (lane & (QT/4 - 1)) in read r, so that the
// 16 lanes the hardware serves together always hit 16 different 16-byte slots of a 256-byte line when QT = 64) and keeps
// QT accumulators.  Per entry: QT/2 v_pk_fma + QT/4 address xors + 2.
//
// Operand ("compact sliced ELL", built by csell_proto.py): rows in slices of 64, columns in chunks of KC; a (chunk, slice)
// block stores its entries pair by pair -- step u holds entries 2u, 2u+1 of every lane whose sub-row has them, lanes in
// ascending order, nothing for the others -- so a wave step is one coalesced 4-byte and one 8-byte load per lane, no
// padding in memory except the odd last entry of a sub-row (index KC = the zero row of the tile).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

struct CsArgs {
  const int* blk_base;            // [nchunks * S]  first pair of the block
  const unsigned char* blk_np;    // [nchunks * S * 64]  pairs per lane
  const unsigned char* blk_max;   // [nchunks * S]  max over the lanes
  const unsigned* pidx;           // per pair: two 16-bit chunk-local indices
  const float2* pval;
  int64_t M, K;
  int KC, nchunks, S;
  const float* R;
  int64_t ldr;
  float* F;
  int64_t ldf;
  float* P;                        // [CG][M][QT] partial sums when CG > 1
  int spw, RBn, CG, xcd_map;
};

__device__ __attribute__((aligned(16))) unsigned int cs_zero[4] = {0u, 0u, 0u, 0u};

typedef float f4 __attribute__((ext_vector_type(4)));

template <int QT, int NPS, bool BIN, int D, int ABL = 0>
__global__ void __launch_bounds__(1024) csell_kernel(CsArgs a) {
  constexpr int ROWB = QT * 4, NPC = QT / 4;
  constexpr int NB = QT >= 32 ? 4 : 8;   // reads per batch
  constexpr int RSH = ROWB == 256 ? 8 : (ROWB == 128 ? 7 : (ROWB == 64 ? 6 : 5));
  extern __shared__ __align__(16) unsigned char tb[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // XCD-aware cut: workgroup i runs on XCD i % 8 (each has its own L2).  All workgroups of an XCD walk the SAME chunks
  // (cg = xcd % CG, CG in {1, 2, 4, 8}), so a tile of R is fetched from the Infinity Cache once per XCD and round and
  // re-read from L2 by the other 31 CUs; with cg = i / RB every CU staged a different tile (12.5 TB/s asked of the
  // Infinity Cache: 0.2 of the 0.9 ms at B = 64, measured by staging once)
  int rb, cg;
  if (a.xcd_map) {
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3, per = 8 / a.CG;
    cg = xcd % a.CG;
    rb = j * per + xcd / a.CG;
    if (rb >= a.RBn) return;
  } else {
    rb = blockIdx.x % a.RBn;
    cg = blockIdx.x / a.RBn;
  }
  const unsigned lanebits = (unsigned)(lane & (NPC - 1)) << 4;

  f4 acc[NPS][NPC];
#pragma unroll
  for (int p = 0; p < NPS; ++p)
#pragma unroll
    for (int r = 0; r < NPC; ++r) acc[p][r] = f4{0.f, 0.f, 0.f, 0.f};

  for (int c = cg; c < a.nchunks; c += a.CG) {
    const int64_t k0 = (int64_t)c * a.KC;
    const int kn = (int)((a.K - k0 < a.KC) ? (a.K - k0) : a.KC);
    __syncthreads();
    if (ABL != 3 || c == cg) {   // ablation 3: the tile is staged once
      const unsigned char* rbase = reinterpret_cast<const unsigned char*>(a.R + k0 * a.ldr);
      const int64_t rowstride = a.ldr * 4;
      const int pieces = (a.KC + 1) * NPC;
      for (int base = (tid >> 6) * 64; base < pieces; base += 1024) {
        const int pc = base + (tid & 63);
        if (pc < pieces) {
          const int k = pc / NPC, slot = pc % NPC;
          const void* src = (k < kn) ? (const void*)(rbase + k * rowstride + slot * 16) : (const void*)cs_zero;
          __builtin_amdgcn_global_load_lds(src, (__attribute__((address_space(3))) void*)(tb + (size_t)base * 16), 16, 0, 0);
        }
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

#pragma unroll
    for (int p = 0; p < NPS; ++p) {
      const int sloc = wave + p * 16;
      const int sl = rb * a.spw + sloc;
      if (sloc >= a.spw || sl >= a.S) continue;   // wave-uniform
      const int b = c * a.S + sl;
      const int np = (int)a.blk_np[(int64_t)b * 64 + lane];
      const int nst = __builtin_amdgcn_readfirstlane((int)a.blk_max[b]);
      int cur = __builtin_amdgcn_readfirstlane(a.blk_base[b]);
      unsigned pi[D];
      float2 pv[D];
      auto issue = [&](int u, int d) __attribute__((always_inline)) {
        const bool act = np > u;
        const unsigned long long mask = __ballot(act);
        const int rank = __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
        const int pos = cur + (act ? rank : 0);
        pi[d] = a.pidx[pos];
        if (!BIN) pv[d] = a.pval[pos];
        cur += __builtin_popcountll(mask);
      };
      // one pair = 2 * NPC (tile row, piece) items, read in batches of NB with the batch after it in flight while a
      // batch is multiplied (the compiler alone keeps two reads in flight and waits for each: LDS 41 % busy, measured)
      auto consume = [&](int u, int d) __attribute__((always_inline)) {
        const bool act = np > u;
        const unsigned x = pi[d];
        const unsigned ka = act ? (x & 0xffffu) : (unsigned)a.KC;
        const unsigned kb = act ? (x >> 16) : (unsigned)a.KC;
        const unsigned basea = (ka << RSH) | lanebits, baseb = (kb << RSH) | lanebits;
        const float wa = BIN ? 1.f : pv[d].x, wb = BIN ? 1.f : pv[d].y;
        constexpr int NIT = 2 * NPC, NBAT = NIT / NB;
        f4 t[2][NB];
        auto loads = [&](int bt, int buf) __attribute__((always_inline)) {
#pragma unroll
          for (int i = 0; i < NB; ++i) {
            const int it = bt * NB + i, r = it % NPC;
            const unsigned base = it < NPC ? basea : baseb;
            // the tile starts at LDS offset 0 (the kernel has no static LDS): an integer address, no base to add
            if (ABL == 2 && (i & 1)) { t[buf][i] = t[buf][i - 1]; continue; }   // ablation: half the LDS reads
            t[buf][i] = *(const __attribute__((address_space(3))) f4*)(uintptr_t)(base ^ (unsigned)(r << 4));
          }
        };
        auto fmas = [&](int bt, int buf) __attribute__((always_inline)) {
#pragma unroll
          for (int i = 0; i < NB; ++i) {
            const int it = bt * NB + i, r = it % NPC;
            const float w = it < NPC ? wa : wb;
            if (ABL == 1 && (i & 1)) { acc[p][r].x += t[buf][i].x + t[buf][i].w; continue; }   // ablation: half the FMAs (the read stays live)
            if (BIN) acc[p][r] += t[buf][i];
            else acc[p][r] = __builtin_elementwise_fma(f4{w, w, w, w}, t[buf][i], acc[p][r]);
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
      };
#pragma unroll
      for (int d = 0; d < D; ++d) issue(d, d);
      for (int u = 0; u < nst; u += D) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
          if (u + d < nst) consume(u + d, d);
          issue(u + d + D, d);
        }
      }
    }
  }

#pragma unroll
  for (int p = 0; p < NPS; ++p) {
    const int sloc = wave + p * 16;
    const int sl = rb * a.spw + sloc;
    if (sloc >= a.spw || sl >= a.S) continue;
    const int64_t m = (int64_t)sl * 64 + lane;
    if (m < a.M) {
#pragma unroll
      for (int r = 0; r < NPC; ++r) {
        const int piece = r ^ (lane & (NPC - 1));
        if (a.CG > 1) *reinterpret_cast<f4*>(a.P + (((int64_t)cg * a.M + m) * QT + piece * 4)) = acc[p][r];
        else *reinterpret_cast<f4*>(a.F + m * a.ldf + piece * 4) = acc[p][r];
      }
    }
  }
}


// ---- version 2: the stream of a wave is one software pipeline across blocks and chunks.  Two rings of D steps: while
// ring A is multiplied, ring B is in flight (and the other way round); the last group of a block requests the FIRST group
// of the next block (its descriptors -- pairs per lane, first pair -- are loaded two blocks ahead), also across the
// restaging of the tile.  Every block runs an even number of groups (>= 2): padded steps are requests of a valid
// address that nobody consumes.  No load sits in a branch; the waits are the compiler's (all loads of a group are
// requested at the top of the half-iteration before the one that uses them).
struct Cs2Args {
  const int2* blk_desc;           // [nblk + 1] {first pair, steps}; the last entry is an empty block
  const unsigned char* blk_np;    // [(nblk + 1) * 64]
  const unsigned* pidx;
  const float2* pval;
  int64_t M, K;
  int KC, nchunks, S;
  const float* R;
  int64_t ldr;
  float* F;
  int64_t ldf;
  float* P;
  int spw, RBn, CG, abl;
};

template <int QT, int NPS, bool BIN, int D>
__global__ void __launch_bounds__(1024) csell2_kernel(Cs2Args a) {
  constexpr int ROWB = QT * 4, NPC = QT / 4;
  constexpr int NB = 4;
  constexpr int RSH = ROWB == 256 ? 8 : (ROWB == 128 ? 7 : (ROWB == 64 ? 6 : 5));
  extern __shared__ __align__(16) unsigned char tb[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int rb = blockIdx.x % a.RBn, cg = blockIdx.x / a.RBn;
  const unsigned lanebits = (unsigned)(lane & (NPC - 1)) << 4;
  const int nblk = a.nchunks * a.S;

  f4 acc[NPS][NPC];
#pragma unroll
  for (int p = 0; p < NPS; ++p)
#pragma unroll
    for (int r = 0; r < NPC; ++r) acc[p][r] = f4{0.f, 0.f, 0.f, 0.f};

  // block k steps after (c, p) in this wave's order
  auto bid = [&](int c, int p, int k) __attribute__((always_inline)) -> int {
    const int pp = p + k;
    const int cc = c + a.CG * (pp / NPS), q = pp % NPS;
    const int sloc = wave + 16 * q, sl = rb * a.spw + sloc;
    return (cc < a.nchunks && sloc < a.spw && sl < a.S) ? cc * a.S + sl : nblk;
  };
  int np_cur, np_n1, np_n2;
  int cur_cur, cur_n1, cur_n2, nst_cur, nst_n1, nst_n2;
  auto load_desc = [&](int b, int& np, int& cur, int& nst) __attribute__((always_inline)) {
    const int2 d = a.blk_desc[b];
    cur = __builtin_amdgcn_readfirstlane(d.x);
    nst = __builtin_amdgcn_readfirstlane(d.y);
    np = (int)a.blk_np[(int64_t)b * 64 + lane];
  };
  load_desc(bid(cg, 0, 0), np_cur, cur_cur, nst_cur);
  load_desc(bid(cg, 0, 1), np_n1, cur_n1, nst_n1);
  load_desc(bid(cg, 0, 2), np_n2, cur_n2, nst_n2);

  unsigned piA[D], piB[D];
  float2 pvA[D], pvB[D];
  // request steps v0 .. v0 + D - 1 of the block with (npx pairs per lane, next pair curx)
  auto issue_group = [&](unsigned (&pi)[D], float2 (&pv)[D], int npx, int& curx, int v0) __attribute__((always_inline)) {
#pragma unroll
    for (int d = 0; d < D; ++d) {
      const bool act = npx > v0 + d;
      const unsigned long long mask = __ballot(act);
      const int rank = __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
      const int pos = curx + (act ? rank : 0);
      pi[d] = a.pidx[pos];
      if (!BIN) pv[d] = a.pval[pos];
      curx += __builtin_popcountll(mask);
    }
  };
  issue_group(piA, pvA, np_cur, cur_cur, 0);

  for (int c = cg; c < a.nchunks; c += a.CG) {
    const int64_t k0 = (int64_t)c * a.KC;
    const int kn = (int)((a.K - k0 < a.KC) ? (a.K - k0) : a.KC);
    __syncthreads();
    {
      const unsigned char* rbase = reinterpret_cast<const unsigned char*>(a.R + k0 * a.ldr);
      const int64_t rowstride = a.ldr * 4;
      const int pieces = (a.KC + 1) * NPC;
      for (int base = (tid >> 6) * 64; base < pieces; base += 1024) {
        const int pc = base + (tid & 63);
        if (pc < pieces) {
          const int k = pc / NPC, slot = pc % NPC;
          const void* src = (k < kn) ? (const void*)(rbase + k * rowstride + slot * 16) : (const void*)cs_zero;
          __builtin_amdgcn_global_load_lds(src, (__attribute__((address_space(3))) void*)(tb + (size_t)base * 16), 16, 0, 0);
        }
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

#pragma unroll
    for (int p = 0; p < NPS; ++p) {
      auto consume_group = [&](const unsigned (&pi)[D], const float2 (&pv)[D], int u0) __attribute__((always_inline)) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
          if (u0 + d < nst_cur) {   // wave-uniform; no memory loads inside
            const bool act = np_cur > u0 + d;
            const unsigned x = pi[d];
            const unsigned ka = act ? (x & 0xffffu) : (unsigned)a.KC;
            const unsigned kb = act ? (x >> 16) : (unsigned)a.KC;
            unsigned kaa = ka, kbb = kb;
            if (a.abl == 4) {   // ablation: tile rows forced into the class of the lane -> no bank conflicts (wrong sums, right time)
              constexpr unsigned NCL = 64 / QT > 1 ? 64 / QT : 1;
              const unsigned q = (unsigned)(lane / NPC) % NCL;
              kaa = (ka & ~(NCL - 1)) | q; kbb = (kb & ~(NCL - 1)) | q;
              if (kaa > (unsigned)a.KC) kaa = a.KC; if (kbb > (unsigned)a.KC) kbb = a.KC;
            }
            const unsigned basea = (kaa << RSH) | lanebits, baseb = (kbb << RSH) | lanebits;
            const float wa = BIN ? 1.f : pv[d].x, wb = BIN ? 1.f : pv[d].y;
            constexpr int NIT = 2 * NPC, NBAT = NIT / NB;
            f4 t[2][NB];
            auto loads = [&](int bt, int buf) __attribute__((always_inline)) {
#pragma unroll
              for (int i = 0; i < NB; ++i) {
                const int it = bt * NB + i, r = it % NPC;
                const unsigned base = it < NPC ? basea : baseb;
                t[buf][i] = *(const __attribute__((address_space(3))) f4*)(uintptr_t)(base ^ (unsigned)(r << 4));
              }
            };
            auto fmas = [&](int bt, int buf) __attribute__((always_inline)) {
#pragma unroll
              for (int i = 0; i < NB; ++i) {
                const int it = bt * NB + i, r = it % NPC;
                const float w = it < NPC ? wa : wb;
                if (BIN) acc[p][r] += t[buf][i];
                else acc[p][r] = __builtin_elementwise_fma(f4{w, w, w, w}, t[buf][i], acc[p][r]);
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
      int ng = (nst_cur + 2 * D - 1) / (2 * D) * 2;   // groups of this block: even, at least two
      ng = ng < 2 ? 2 : ng;
      for (int g = 0; g < ng; g += 2) {
        issue_group(piB, pvB, np_cur, cur_cur, (g + 1) * D);
        consume_group(piA, pvA, g * D);
        const bool tail = g + 2 >= ng;
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

#pragma unroll
  for (int p = 0; p < NPS; ++p) {
    const int sloc = wave + p * 16;
    const int sl = rb * a.spw + sloc;
    if (sloc >= a.spw || sl >= a.S) continue;
    const int64_t m = (int64_t)sl * 64 + lane;
    if (m < a.M) {
#pragma unroll
      for (int r = 0; r < NPC; ++r) {
        const int piece = r ^ (lane & (NPC - 1));
        if (a.CG > 1) *reinterpret_cast<f4*>(a.P + (((int64_t)cg * a.M + m) * QT + piece * 4)) = acc[p][r];
        else *reinterpret_cast<f4*>(a.F + m * a.ldf + piece * 4) = acc[p][r];
      }
    }
  }
}

__global__ void csell_reduce_kernel(const float* __restrict__ P, int CG, int64_t M, int QT, float* __restrict__ F, int64_t ldf) {
  const int64_t total = M * (QT / 4);
  const int64_t plane = M * QT;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    f4 s = *reinterpret_cast<const f4*>(P + i * 4);
    for (int c = 1; c < CG; ++c) s += *reinterpret_cast<const f4*>(P + c * plane + i * 4);
    const int64_t m = i / (QT / 4);
    const int b = (int)(i % (QT / 4)) * 4;
    *reinterpret_cast<f4*>(F + m * ldf + b) = s;
  }
}

template <int QT, int NPS, bool BIN, int D, int ABL = 0>
static int run_variant(const CsArgs& a, size_t lds, int iters, float* ms) {
  auto kern = csell_kernel<QT, NPS, BIN, D, ABL>;
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return 2;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const unsigned grid = a.xcd_map ? (unsigned)(8 * ((a.RBn + 8 / a.CG - 1) / (8 / a.CG))) : (unsigned)(a.RBn * a.CG);
  auto once = [&]() {
    hipLaunchKernelGGL(kern, dim3(grid), dim3(1024), lds, 0, a);
    if (a.CG > 1) {
      int64_t g = (a.M * (QT / 4) + 255) / 256;
      if (g > 4096) g = 4096;
      hipLaunchKernelGGL(csell_reduce_kernel, dim3((unsigned)g), dim3(256), 0, 0, a.P, a.CG, a.M, QT, a.F, a.ldf);
    }
  };
  once();
  if (hipDeviceSynchronize() != hipSuccess) { fprintf(stderr, "csell: %s\n", hipGetErrorString(hipGetLastError())); return 3; }
  hipEventRecord(e0, 0);
  for (int i = 0; i < iters; ++i) once();
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float t = 0.f;
  hipEventElapsedTime(&t, e0, e1);
  *ms = t / (iters > 0 ? iters : 1);
  hipEventDestroy(e0);
  hipEventDestroy(e1);
  return hipGetLastError() == hipSuccess ? 0 : 4;
}

extern "C" int csell_run(const int* blk_base, const unsigned char* blk_np, const unsigned char* blk_max, const unsigned* pidx,
                         const float* pval, int64_t M, int64_t K, int KC, int nchunks, int S, const float* R, int64_t ldr,
                         float* F, int64_t ldf, float* P, int spw, int RBn, int CG, int QT, int binary, int depth, int iters,
                         float* ms) {
  CsArgs a{blk_base, blk_np, blk_max, pidx, reinterpret_cast<const float2*>(pval), M, K, KC, nchunks, S, R, ldr, F, ldf, P, spw, RBn, CG, 0};
  if (getenv("CSELL_XCD") && atoi(getenv("CSELL_XCD")) && (CG == 1 || CG == 2 || CG == 4 || CG == 8)) a.xcd_map = 1;
  const size_t lds = (size_t)(KC + 1) * QT * 4;
  const int abl = getenv("CSELL_ABL") ? atoi(getenv("CSELL_ABL")) : 0;
  if (lds > 160 * 1024) return 1;
#define CS(QTV, NPSV)                                                                             \
  if (QT == QTV) {                                                                                \
    if (spw > 16 * NPSV) return 5;                                                                \
    if (abl == 1) return run_variant<QTV, NPSV, false, 4, 1>(a, lds, iters, ms);                  \
    if (abl == 2) return run_variant<QTV, NPSV, false, 4, 2>(a, lds, iters, ms);                  \
    if (abl == 3) return run_variant<QTV, NPSV, false, 4, 3>(a, lds, iters, ms);                  \
    if (binary) return depth == 2 ? run_variant<QTV, NPSV, true, 2>(a, lds, iters, ms) : run_variant<QTV, NPSV, true, 4>(a, lds, iters, ms); \
    return depth == 2 ? run_variant<QTV, NPSV, false, 2>(a, lds, iters, ms) : run_variant<QTV, NPSV, false, 4>(a, lds, iters, ms); \
  }
  CS(16, 4) CS(32, 2) CS(64, 1)
#undef CS
  return 6;
}

template <int QT, int NPS, bool BIN, int D>
static int run_variant2(const Cs2Args& a, size_t lds, int iters, float* ms) {
  auto kern = csell2_kernel<QT, NPS, BIN, D>;
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return 2;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const unsigned grid = (unsigned)(a.RBn * a.CG);
  auto once = [&]() {
    hipLaunchKernelGGL(kern, dim3(grid), dim3(1024), lds, 0, a);
    if (a.CG > 1) {
      int64_t g = (a.M * (QT / 4) + 255) / 256;
      if (g > 4096) g = 4096;
      hipLaunchKernelGGL(csell_reduce_kernel, dim3((unsigned)g), dim3(256), 0, 0, a.P, a.CG, a.M, QT, a.F, a.ldf);
    }
  };
  once();
  if (hipDeviceSynchronize() != hipSuccess) { fprintf(stderr, "csell2: %s\n", hipGetErrorString(hipGetLastError())); return 3; }
  hipEventRecord(e0, 0);
  for (int i = 0; i < iters; ++i) once();
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float t = 0.f;
  hipEventElapsedTime(&t, e0, e1);
  *ms = t / (iters > 0 ? iters : 1);
  hipEventDestroy(e0);
  hipEventDestroy(e1);
  return hipGetLastError() == hipSuccess ? 0 : 4;
}

extern "C" int csell2_run(const int* blk_desc, const unsigned char* blk_np, const unsigned* pidx, const float* pval, int64_t M,
                          int64_t K, int KC, int nchunks, int S, const float* R, int64_t ldr, float* F, int64_t ldf, float* P,
                          int spw, int RBn, int CG, int QT, int binary, int depth, int iters, float* ms) {
  Cs2Args a{reinterpret_cast<const int2*>(blk_desc), blk_np, pidx, reinterpret_cast<const float2*>(pval), M, K, KC, nchunks, S,
            R, ldr, F, ldf, P, spw, RBn, CG, getenv("CSELL_ABL") ? atoi(getenv("CSELL_ABL")) : 0};
  const size_t lds = (size_t)(KC + 1) * QT * 4;
  if (lds > 160 * 1024) return 1;
#define CS2(QTV, NPSV)                                                                           \
  if (QT == QTV) {                                                                                \
    if (spw > 16 * NPSV) return 5;                                                                \
    if (binary) return depth == 2 ? run_variant2<QTV, NPSV, true, 2>(a, lds, iters, ms) : run_variant2<QTV, NPSV, true, 4>(a, lds, iters, ms); \
    return depth == 2 ? run_variant2<QTV, NPSV, false, 2>(a, lds, iters, ms) : run_variant2<QTV, NPSV, false, 4>(a, lds, iters, ms); \
  }
  CS2(16, 4) CS2(32, 2) CS2(64, 1)
#undef CS2
  return 6;
}
