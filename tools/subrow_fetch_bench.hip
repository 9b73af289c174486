// How fast can a CU fetch many short runs ("sub-rows": ~62 x (u16 index + f32 value)) from an L2-resident table?
//   A: what stage 1 does today -- one global_load_ushort + one global_load_dword per sub-row and wave (2-4 B/lane)
//   B: LDS-DMA -- global_load_lds_dwordx4, one 16-byte piece per lane (24 pieces per sub-row), data lands in LDS
// Same sub-row list for both; each wave handles its own list; results are consumed so nothing is optimised away.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

constexpr int SUBROW = 64;           // entries per sub-row (fixed here), 16-byte aligned starts
constexpr int PER_WAVE = 512;        // sub-rows per wave

__global__ void __launch_bounds__(64) variant_a(const unsigned short* __restrict__ idx, const float* __restrict__ val,
                                                const int* __restrict__ starts, float* out) {
  const int lane = threadIdx.x;
  const int* st = starts + (size_t)blockIdx.x * PER_WAVE;
  float acc = 0.f;
  for (int i = 0; i < PER_WAVE; i += 8) {
    unsigned short j[8]; float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int b = __builtin_amdgcn_readfirstlane(st[i + u]);
      j[u] = idx[b + lane]; v[u] = val[b + lane];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) acc += v[u] * (float)j[u];
  }
  if (acc == 12345.678f) out[0] = acc;
}

__global__ void __launch_bounds__(64) variant_b(const unsigned short* __restrict__ idx, const float* __restrict__ val,
                                                const int* __restrict__ starts, float* out) {
  // per wave: 8 sub-rows per batch = 8 * (128 B idx + 256 B val) = 3 KB = 3 LDS-DMA instructions of 1 KB
  __shared__ __align__(16) unsigned char stage[2][3072];
  const int lane = threadIdx.x;
  const int* st = starts + (size_t)blockIdx.x * PER_WAVE;
  float acc = 0.f;
  auto issue = [&](int buf, int i) {
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int piece = k * 64 + lane;            // 0..191 : sub-row u = piece / 24, p = piece % 24
      const int u = piece / 24, p = piece % 24;
      const int b = st[i + u];
      const char* src = p < 8 ? (const char*)(idx + b) + p * 16 : (const char*)(val + b) + (p - 8) * 16;
      __builtin_amdgcn_global_load_lds((const void*)src, (__attribute__((address_space(3))) void*)(stage[buf] + k * 1024),
                                       16, 0, 0);
    }
  };
  issue(0, 0);
  for (int i = 0; i < PER_WAVE; i += 8) {
    const int buf = (i >> 3) & 1;
    if (i + 8 < PER_WAVE) issue(buf ^ 1, i + 8);
    if (i + 8 < PER_WAVE) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const unsigned short* si = (const unsigned short*)(stage[buf] + u * 384);
      const float* sv = (const float*)(stage[buf] + u * 384 + 128);
      acc += sv[lane] * (float)si[lane];
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  if (acc == 12345.678f) out[0] = acc;
}

// C: the same sub-rows stored as ONE contiguous block each (128 B of indices directly followed by 256 B of values):
// one run of 384 bytes instead of two runs in two arrays -- fewer 64-byte sectors per sub-row
__global__ void __launch_bounds__(64) variant_c(const unsigned char* __restrict__ blk, const int* __restrict__ starts,
                                                float* out) {
  const int lane = threadIdx.x;
  const int* st = starts + (size_t)blockIdx.x * PER_WAVE;
  float acc = 0.f;
  for (int i = 0; i < PER_WAVE; i += 8) {
    unsigned short j[8]; float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int b = __builtin_amdgcn_readfirstlane(st[i + u]);       // block start in units of 6 bytes * 8 = 48 B steps
      const unsigned char* p = blk + (size_t)b * 6;
      j[u] = reinterpret_cast<const unsigned short*>(p)[lane];
      v[u] = reinterpret_cast<const float*>(p + 128)[lane];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) acc += v[u] * (float)j[u];
  }
  if (acc == 12345.678f) out[0] = acc;
}

int main() {
  const size_t table = 640000;   // entries: 1.28 MB idx + 2.56 MB val = one XCD's chunk at C2
  unsigned short* idx; float* val; int* starts; float* out;
  const int waves = 256 * 24 * 4;
  CK(hipMalloc(&idx, (table + 128) * 2)); CK(hipMalloc(&val, (table + 128) * 4));
  CK(hipMalloc(&starts, (size_t)waves * PER_WAVE * 4)); CK(hipMalloc(&out, 4));
  CK(hipMemset(idx, 1, (table + 128) * 2)); CK(hipMemset(val, 0, (table + 128) * 4));
  std::vector<int> h((size_t)waves * PER_WAVE);
  srand(3);
  for (auto& x : h) x = (rand() % (int)(table / 8 - 8)) * 8;   // 16-byte aligned for both arrays
  CK(hipMemcpy(starts, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  unsigned char* blk; CK(hipMalloc(&blk, (table + 128) * 6 + 512)); CK(hipMemset(blk, 0, (table + 128) * 6 + 512));
  for (int var = 0; var < 3; ++var) {
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipEventRecord(a));
      if (var == 0) hipLaunchKernelGGL(variant_a, dim3(waves), dim3(64), 0, 0, idx, val, starts, out);
      else if (var == 1) hipLaunchKernelGGL(variant_b, dim3(waves), dim3(64), 0, 0, idx, val, starts, out);
      else hipLaunchKernelGGL(variant_c, dim3(waves), dim3(64), 0, 0, blk, starts, out);
      CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); CK(hipEventElapsedTime(&ms, a, b));
    }
    const double subrows = (double)waves * PER_WAVE;
    printf("%s: %.3f ms  %.1f G sub-rows/s  %.2f TB/s  (%.1f clk per sub-row per CU at 2.4 GHz)\n",
           var == 0 ? "A narrow register loads" : (var == 1 ? "B LDS-DMA 16-byte pieces" : "C one 384-byte block per sub-row"), ms, subrows / ms / 1e6,
           subrows * 384 / ms / 1e9, ms * 1e-3 * 2.4e9 * 256 / subrows);
  }
  return 0;
}
