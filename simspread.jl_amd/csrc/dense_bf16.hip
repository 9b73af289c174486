// Dense-similarity regime on the bf16 matrix cores with fp32-accurate operands.
//
// dense.hip multiplies the thresholded similarities with the fp32-input MFMA (157 TFLOP/s peak).  The bf16 MFMA of
// gfx950 is 16x faster per instruction, and an fp32 number is EXACTLY the sum of three bf16 numbers (8 + 8 + 8
// significant bits: hi = bf16(x), mid = bf16(x - hi), lo = bf16(x - hi - mid)), so
//   * unweighted features (featurize(.., weighted = false), src/core.jl:106-112): cut(S) is 0/1, exact in one bf16
//     plane; the query side cut(S)/kf needs three planes -> 3 bf16 products, every one exact, summed in fp32 by
//     the MFMA.  The result differs from the fp32 path only by the order of the fp32 additions;
//   * weighted features: three planes on both sides; the six largest of the nine plane products are kept
//     (hi*hi, hi*mid, mid*hi, mid*mid, hi*lo, lo*hi); the dropped ones are below 2^-24 of the product, i.e.
//     under fp32 rounding.
// A pre-pass writes the planes K-contiguous ([row][Kp] bf16, rows padded to 128, K to 64, zero filled) once per
// graph for the source side and once per block of rows for the query side (threshold, 1/kf and the leave-one-out
// diagonal are applied there), so the GEMM itself is a plain bf16 GEMM: 128 x 128 x 64 tiles, 4 waves x (2 x 2)
// v_mfma_f32_32x32x16_bf16, both operand tiles staged by LDS-DMA (global_load_lds, 16 bytes per lane) into two
// LDS buffers, one barrier per K-step.  The LDS image is lane-linear, so the bank swizzle is applied on the source
// side: the 16-byte slot c of tile row r lives at slot c ^ ((r >> 1) & 7), which makes every ds_read_b128 lane
// group hit 16 different slots of the 256-byte LDS line.
#include "graph.hpp"

namespace ss {

#define SS_LAUNCH_CHECK()                                                             \
  do {                                                                                \
    hipError_t _e = hipGetLastError();                                                \
    if (_e != hipSuccess)                                                             \
      return fail(SS_EHIP, "%s:%d kernel launch: %s", __FILE__, __LINE__, hipGetErrorString(_e)); \
  } while (0)

__device__ __forceinline__ unsigned short bf16_rne(float x) {
  unsigned u = __float_as_uint(x);
  u += 0x7FFFu + ((u >> 16) & 1u);  // round to nearest even (inputs are finite)
  return (unsigned short)(u >> 16);
}
__device__ __forceinline__ float bf16_f(unsigned short h) { return __uint_as_float((unsigned)h << 16); }

// ------------------------------------------------------------------ planes of one operand
// src: column-major similarities S[row + k*ld], rows [row0, row0 + rows).  dst: NP planes [Rp][Kp] bf16, zeroed.
// value = cut(S) (* scale[k]) and 0 on the leave-one-out diagonal (k == loo_first + local row).
template <int NP>
__global__ void __launch_bounds__(256) dense_planes_kernel(const float* __restrict__ S, int64_t ld, int64_t row0,
                                                            int64_t rows, int64_t K, float alpha, int weighted,
                                                            const float* __restrict__ scale, int64_t loo_first,
                                                            const int* __restrict__ row_ids,
                                                            unsigned short* __restrict__ dst, int64_t Rp, int64_t Kp) {
  __shared__ float tile[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  const int64_t r0 = (int64_t)blockIdx.x * 32, k0 = (int64_t)blockIdx.y * 32;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int64_t k = k0 + ty + 8 * i, r = r0 + tx;
    float v = 0.f;
    if (k < K && r < rows) {
      const float x = S[(row_ids ? (int64_t)row_ids[row0 + r] : row0 + r) + k * ld];  // k-fold: rows of a fold's members
      v = (x >= alpha) ? (weighted ? x : 1.0f) : 0.0f;
      if (scale) v *= scale[k];
      if (loo_first >= 0 && k == loo_first + r) v = 0.f;
    }
    tile[ty + 8 * i][tx] = v;  // [k][r]
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int64_t r = r0 + ty + 8 * i, k = k0 + tx;
    if (r < rows && k < K) {
      const float v = tile[tx][ty + 8 * i];
      const unsigned short hi = bf16_rne(v);
      dst[r * Kp + k] = hi;
      if (NP > 1) {
        const float r1 = v - bf16_f(hi);
        const unsigned short mid = bf16_rne(r1);
        dst[Rp * Kp + r * Kp + k] = mid;
        dst[2 * Rp * Kp + r * Kp + k] = bf16_rne(r1 - bf16_f(mid));
      }
    }
  }
}

// ------------------------------------------------------------------ the GEMM
struct DenseBf16Args {
  const unsigned short* A;   // planes [npa][Mp][Kp]
  const unsigned short* B;   // planes [npb][Np][Kp]
  int64_t a_plane, b_plane;  // elements per plane
  int npairs;
  int pa[6], pb[6];
  int bnew[6], bidx[6], nbs;  // ring kernel: source-tile staging plan of the pairs
  int64_t Kp, M, N;
  int64_t row_begin;         // LOO: query i = row_begin + m
  const float* inv_n;        // [N] 1/ks
  const int* ks;             // LOO: integer source degrees
  const float* Braw;         // LOO: raw source similarities (column-major, ld = ldb) for the X[s][f_q] test
  int64_t ldb;
  float alpha;
  int weighted;
  float* out;
  int64_t ldo;
  int gx, gy;                // column / row blocks
};

using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using f32x16b = __attribute__((ext_vector_type(16))) float;

constexpr int BT = 128, BKB = 64;            // tile rows / K per step (bf16 elements)
constexpr int TILE_BYTES = BT * BKB * 2;     // 16 KiB per operand tile

template <bool LOO>
__global__ void __launch_bounds__(256) transfer_dense_bf16_kernel(DenseBf16Args a) {
  __shared__ __align__(16) unsigned char lds[2][2][TILE_BYTES];  // [buffer][A|B]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  // tile order: all row blocks of a group of GW column blocks before the next group, so that the ~512 workgroups
  // in flight share gy + GW operand tiles per K-step instead of 1 + 391 (B would be re-streamed from HBM per row block)
  int bm, bn;
  {
    const int gy = a.gy, gx = a.gx, GW = 16;
    const int id = (int)blockIdx.x;
    const int grp = id / (gy * GW);
    const int w = (gx - grp * GW < GW) ? (gx - grp * GW) : GW;  // width of this (possibly last, narrower) group
    const int local = id - grp * gy * GW;
    (void)w;
    bm = local % gy;
    bn = grp * GW + local / gy;
  }
  const int64_t m0 = (int64_t)bm * BT, n0 = (int64_t)bn * BT;
  f32x16b acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  const int ksteps = (int)(a.Kp / BKB);
  const int total = a.npairs * ksteps;

  // one operand tile = 1024 pieces of 16 bytes; wave w moves pieces [256 w, 256 w + 256) with four instructions
  auto stage = [&](int step, int buf) __attribute__((always_inline)) {
    const int pair = step / ksteps, kt = step - pair * ksteps;
    const unsigned short* Ap = a.A + (int64_t)a.pa[pair] * a.a_plane + m0 * a.Kp + (int64_t)kt * BKB;
    const unsigned short* Bp = a.B + (int64_t)a.pb[pair] * a.b_plane + n0 * a.Kp + (int64_t)kt * BKB;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int p0 = wave * 256 + i * 64;
      const int p = p0 + lane;
      const int row = p >> 3, c = (p & 7) ^ ((row >> 1) & 7);
      __builtin_amdgcn_global_load_lds((const void*)(Ap + (int64_t)row * a.Kp + 8 * c),
                                       (__attribute__((address_space(3))) void*)(&lds[buf][0][p0 * 16]), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((const void*)(Bp + (int64_t)row * a.Kp + 8 * c),
                                       (__attribute__((address_space(3))) void*)(&lds[buf][1][p0 * 16]), 16, 0, 0);
    }
  };

  const int r = lane & 31, h = lane >> 5;
  auto frag = [&](const unsigned char* tile, int row, int c) __attribute__((always_inline)) {
    return *reinterpret_cast<const bf16x8*>(tile + row * 128 + ((c ^ ((row >> 1) & 7)) << 4));
  };

  stage(0, 0);
  for (int step = 0; step < total; ++step) {
    const int buf = step & 1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();  // tile `step` has landed for everybody; everybody is done with the other buffer
    if (step + 1 < total) stage(step + 1, buf ^ 1);
    const unsigned char* At = lds[buf][0];
    const unsigned char* Bt = lds[buf][1];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int c = 2 * s + h;
      bf16x8 af[2], bfr[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) af[i] = frag(At, wm * 64 + i * 32 + r, c);
#pragma unroll
      for (int j = 0; j < 2; ++j) bfr[j] = frag(Bt, wn * 64 + j * 32 + r, c);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
    }
  }

  // C/D layout: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int64_t n = n0 + wn * 64 + j * 32 + (lane & 31);
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int64_t m = m0 + wm * 64 + i * 32 + (q & 3) + 8 * (q >> 2) + 4 * (lane >> 5);
        if (m < a.M && n < a.N) {
          float z;
          if (LOO) {
            const int64_t qi = a.row_begin + m;
            const float x = a.Braw[n + qi * a.ldb];
            const int has = ((x >= a.alpha) ? (a.weighted ? x : 1.0f) : 0.0f) != 0.0f ? 1 : 0;  // X[s][f_q]
            const int d = a.ks[n] - has;
            z = (d > 0 && n != qi) ? acc[i][j][q] * (1.0f / (float)d) : 0.0f;
          } else {
            z = acc[i][j][q] * a.inv_n[n];
          }
          a.out[m * a.ldo + n] = z;
        }
      }
    }
}

// ------------------------------------------------------------------ the GEMM, three-stage ring
// The kernel above requests a tile while the previous one is multiplied (512 cycles per wave), which is less than
// the L2 latency, so it lives off the second workgroup of the CU.  Here one workgroup per CU (8 waves as 4 x 2,
// 256 x TN tile, wave tile 64 x TN/2) keeps two-slot rings in LDS with the next step in flight; with the default
// 256 x 256 x 64 a step is 1024 MFMA cycles per wave, two waves per SIMD.  The LDS-DMA is written as inline
// assembly so that the compiler does not drain it (s_waitcnt vmcnt(0)) in front of every ds_read; completion is
// awaited with an explicit s_waitcnt + one barrier per step.
__device__ __forceinline__ void lds_dma16(const void* src, unsigned lds_base) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(src), "s"(lds_base) : "memory");
}

constexpr int RING_TM = 256;

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// TN: tile columns (128 or 256), BK: K per step (32 or 64).  Steps run K-tile outer, plane pair inner, and the
// operand tiles live in two separate two-slot rings: the query-side tile changes every step, the source-side tile
// only when the pair names another source plane (unweighted: once per K-tile for the three query planes), so a
// third of the DMA bytes go away -- and L2 -> LDS DMA at ~21 B/clk per CU is what a 256 x 256 x 64 step waits for.
template <bool LOO, int TN, int BK>
__global__ void __launch_bounds__(512) transfer_dense_bf16_ring_kernel(DenseBf16Args a) {
  constexpr int SLOTS = BK / 8, ROWB = BK * 2;          // 16-byte slots / bytes per tile row
  constexpr int A_BYTES = RING_TM * ROWB, B_BYTES = TN * ROWB;
  constexpr int NA = RING_TM * SLOTS / 512, NB = TN * SLOTS / 512;  // DMA instructions per thread and tile
  constexpr int WN = TN / 64;                           // 32-column MFMA tiles per wave (wave tile 64 x TN/2)
  extern __shared__ __align__(16) unsigned char ring[];  // [A slot 0 | A slot 1 | B slot 0 | B slot 1]
  const unsigned ring0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)ring;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;  // 4 x 2 waves
  int bm, bn;
  {
    const int gy = a.gy, GW = (TN == 128) ? 16 : 8;
    const int id = (int)blockIdx.x;
    const int grp = id / (gy * GW);
    const int local = id - grp * gy * GW;
    bm = local % gy;
    bn = grp * GW + local / gy;
  }
  const int64_t m0 = (int64_t)bm * RING_TM, n0 = (int64_t)bn * TN;
  f32x16b acc[2][WN];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < WN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  const int ksteps = (int)(a.Kp / BK);
  const int np = a.npairs;
  const int total = np * ksteps;
  auto swz = [](int row) __attribute__((always_inline)) { return SLOTS == 8 ? ((row >> 1) & 7) : ((row >> 2) & 3); };

  // step t = (K-tile kt, pair p).  a.bnew[p]: pair p stages a source tile; a.bidx[p]: which of the K-tile's
  // source stagings pair p reads; a.nbs: source stagings per K-tile.  Source slot = (kt * nbs + bidx[p]) & 1.
  auto stage = [&](int t) __attribute__((always_inline)) {
    const int kt = t / np, p = t - kt * np;
    const unsigned short* Ap = a.A + (int64_t)a.pa[p] * a.a_plane + m0 * a.Kp + (int64_t)kt * BK;
    const unsigned abase = ring0 + (unsigned)(t & 1) * A_BYTES;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int p0 = wave * (64 * NA) + i * 64;
      const int q = p0 + lane;
      const int row = q / SLOTS, c = (q % SLOTS) ^ swz(row);
      lds_dma16(Ap + (int64_t)row * a.Kp + 8 * c, __builtin_amdgcn_readfirstlane(abase + (unsigned)p0 * 16u));
    }
    if (a.bnew[p]) {
      const unsigned short* Bp = a.B + (int64_t)a.pb[p] * a.b_plane + n0 * a.Kp + (int64_t)kt * BK;
      const unsigned bbase = ring0 + 2u * A_BYTES + (unsigned)((kt * a.nbs + a.bidx[p]) & 1) * B_BYTES;
#pragma unroll
      for (int i = 0; i < NB; ++i) {
        const int p0 = wave * (64 * NB) + i * 64;
        const int q = p0 + lane;
        const int row = q / SLOTS, c = (q % SLOTS) ^ swz(row);
        lds_dma16(Bp + (int64_t)row * a.Kp + 8 * c, __builtin_amdgcn_readfirstlane(bbase + (unsigned)p0 * 16u));
      }
    }
  };

  const int r = lane & 31, h = lane >> 5;
  auto frag = [&](const unsigned char* tile, int row, int c) __attribute__((always_inline)) {
    return *reinterpret_cast<const bf16x8*>(tile + row * ROWB + ((c ^ swz(row)) << 4));
  };

  stage(0);
  for (int t = 0; t < total; ++t) {
    wait_vmcnt<0>();   // my pieces of step t (requested one step ago) have landed ...
    __syncthreads();   // ... and everybody's; everybody is also done with step t-1, whose slots are refilled now
    if (t + 1 < total) stage(t + 1);
    const int kt = t / np, p = t - kt * np;
    const unsigned char* At = ring + (t & 1) * A_BYTES;
    const unsigned char* Bt = ring + 2 * A_BYTES + ((kt * a.nbs + a.bidx[p]) & 1) * B_BYTES;
#pragma unroll
    for (int s = 0; s < BK / 16; ++s) {
      const int c = 2 * s + h;
      bf16x8 af[2], bfr[WN];
#pragma unroll
      for (int i = 0; i < 2; ++i) af[i] = frag(At, wm * 64 + i * 32 + r, c);
#pragma unroll
      for (int j = 0; j < WN; ++j) bfr[j] = frag(Bt, wn * (TN / 2) + j * 32 + r, c);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
    }
  }

#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < WN; ++j) {
      const int64_t n = n0 + wn * (TN / 2) + j * 32 + (lane & 31);
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int64_t m = m0 + wm * 64 + i * 32 + (q & 3) + 8 * (q >> 2) + 4 * (lane >> 5);
        if (m < a.M && n < a.N) {
          float z;
          if (LOO) {
            const int64_t qi = a.row_begin + m;
            const float x = a.Braw[n + qi * a.ldb];
            const int has = ((x >= a.alpha) ? (a.weighted ? x : 1.0f) : 0.0f) != 0.0f ? 1 : 0;  // X[s][f_q]
            const int d = a.ks[n] - has;
            z = (d > 0 && n != qi) ? acc[i][j][q] * (1.0f / (float)d) : 0.0f;
          } else {
            z = acc[i][j][q] * a.inv_n[n];
          }
          a.out[m * a.ldo + n] = z;
        }
      }
    }
}

template <bool LOO, int TN, int BK>
static int launch_ring(DenseBf16Args& a, int64_t Mp, int64_t Np) {
  constexpr size_t lds = 2 * (size_t)(RING_TM + TN) * BK * 2;
  static_assert(lds <= 160 * 1024, "rings do not fit the LDS");
  a.gx = (int)(Np / TN);
  a.gy = (int)(Mp / RING_TM);
  // pair order of the ring kernel: pairs that read the same source plane next to one another
  int nbs = 0;
  for (int p = 0; p < a.npairs; ++p) {
    a.bnew[p] = (p == 0 || a.pb[p] != a.pb[p - 1]) ? 1 : 0;
    if (a.bnew[p]) ++nbs;
    a.bidx[p] = nbs - 1;
  }
  a.nbs = nbs;
  static std::atomic<bool> attr_set{false};
  if (!attr_set) {
    SS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&transfer_dense_bf16_ring_kernel<LOO, TN, BK>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_set = true;
  }
  hipLaunchKernelGGL((transfer_dense_bf16_ring_kernel<LOO, TN, BK>), dim3((unsigned)(a.gx * a.gy)), dim3(512), lds,
                     ctx().stream, a);
  SS_LAUNCH_CHECK();
  return SS_OK;
}

static inline int64_t round_up(int64_t x, int64_t m) { return (x + m - 1) / m * m; }

// planes of rows [row0, row0 + rows) of a column-major similarity block into `buf` (grown as needed)
static int make_planes(const float* S, int64_t ld, int64_t row0, int64_t rows, int64_t K, float alpha, bool weighted,
                       const float* scale, int64_t loo_first, const int* row_ids, int np, DevBuf<unsigned short>& buf,
                       int64_t* Rp_out, int64_t* Kp_out) {
  const int64_t Rp = round_up(rows > 0 ? rows : 1, RING_TM), Kp = round_up(K > 0 ? K : 1, BKB);
  const size_t need = (size_t)np * Rp * Kp;
  if (buf.n < need) SS_TRY(buf.alloc(need));
  SS_HIP(hipMemsetAsync(buf.p, 0, need * sizeof(unsigned short), ctx().stream));
  dim3 grid((unsigned)ceil_div(rows, 32), (unsigned)ceil_div(K, 32));
  if (np == 1)
    hipLaunchKernelGGL(dense_planes_kernel<1>, grid, dim3(256), 0, ctx().stream, S, ld, row0, rows, K, alpha,
                       weighted ? 1 : 0, scale, loo_first, row_ids, buf.p, Rp, Kp);
  else
    hipLaunchKernelGGL(dense_planes_kernel<3>, grid, dim3(256), 0, ctx().stream, S, ld, row0, rows, K, alpha,
                       weighted ? 1 : 0, scale, loo_first, row_ids, buf.p, Rp, Kp);
  SS_LAUNCH_CHECK();
  *Rp_out = Rp;
  *Kp_out = Kp;
  return SS_OK;
}

int launch_transfer_dense_bf16(DenseSim<float>& d, bool loo, const float* inv_k, const float* inv_n, const int* ks,
                               int64_t row_begin, int64_t nrows, float* out, int64_t ldo, bool source_rows,
                               const int* row_ids) {
  if (nrows <= 0 || d.ns <= 0) return SS_OK;
  // source side: once per graph (alpha and the weighting are fixed in the handle)
  const int npb = d.weighted ? 3 : 1;
  int64_t Np = 0, Kp = 0;
  if (d.Bpl_np != npb) {
    SS_TRY(make_planes(d.Ss.p, d.ns, 0, d.ns, d.nf, d.alpha, d.weighted, nullptr, -1, nullptr, npb, d.Bpl, &Np, &Kp));
    d.Bpl_np = npb;
    d.Bpl_Np = Np;
    d.Bpl_Kp = Kp;
  }
  Np = d.Bpl_Np;
  Kp = d.Bpl_Kp;
  // query side: this block of rows, threshold * 1/kf (LOO: 1/(kf-1), own feature dropped)
  int64_t Mp = 0, Kp2 = 0;
  // rows of the source similarity itself (LOO: own feature dropped; k-fold: the members row_ids[row_begin ...])
  const bool from_ss = loo || source_rows || row_ids != nullptr;
  SS_TRY(make_planes(from_ss ? d.Ss.p : d.Sq.p, from_ss ? d.ns : d.nq, row_begin, nrows, d.nf, d.alpha, d.weighted,
                     inv_k, loo ? row_begin : -1, row_ids, 3, d.Apl, &Mp, &Kp2));
  DenseBf16Args a{};
  a.A = d.Apl.p;
  a.B = d.Bpl.p;
  a.a_plane = Mp * Kp;
  a.b_plane = Np * Kp;
  if (d.weighted) {
    // the six largest plane products, grouped by source plane (so that the ring kernel stages a source tile once
    // per group), smaller terms first inside a group
    const int pa[6] = {0, 1, 0, 2, 1, 0}, pb[6] = {2, 1, 1, 0, 0, 0};
    a.npairs = 6;
    for (int i = 0; i < 6; ++i) { a.pa[i] = pa[i]; a.pb[i] = pb[i]; }
  } else {
    a.npairs = 3;
    a.pa[0] = 2; a.pa[1] = 1; a.pa[2] = 0;
    a.pb[0] = a.pb[1] = a.pb[2] = 0;
  }
  a.Kp = Kp;
  a.M = nrows;
  a.N = d.ns;
  a.row_begin = row_begin;
  a.inv_n = inv_n;
  a.ks = ks;
  a.Braw = d.Ss.p;
  a.ldb = d.ns;
  a.alpha = d.alpha;
  a.weighted = d.weighted ? 1 : 0;
  a.out = out;
  a.ldo = ldo;
  a.gx = (int)(Np / BT);
  // enough 256 x 256 tiles to fill the chip: one workgroup per CU (measured at 50k, unweighted: 44.6 ms; the
  // 128 x 128 kernel with two workgroups per CU 65).  SS_DENSE_RING=0 / 1 forces the choice.
  const bool ring_ok = (Mp / RING_TM) * (Np / 256) >= ctx().num_cu;
  const char* ring_env = getenv("SS_DENSE_RING");  // 0: never, 1: always, unset: by size
  if (ring_env ? atoi(ring_env) != 0 : ring_ok) {
    path_add("transfer_dense_bf16_ring");
    return loo ? launch_ring<true, 256, 64>(a, Mp, Np) : launch_ring<false, 256, 64>(a, Mp, Np);
  }
  path_add("transfer_dense_bf16_128");
  a.gy = (int)(Mp / BT);
  dim3 grid((unsigned)(a.gx * a.gy));
  if (loo) hipLaunchKernelGGL(transfer_dense_bf16_kernel<true>, grid, dim3(256), 0, ctx().stream, a);
  else hipLaunchKernelGGL(transfer_dense_bf16_kernel<false>, grid, dim3(256), 0, ctx().stream, a);
  SS_LAUNCH_CHECK();
  return SS_OK;
}

}  // namespace ss
