// Microbenchmark: cost of scattered accumulation into an LDS array (stage-1 inner operation).
//   mode 0: atomicAdd(float) random   1: atomicAdd sequential   2: plain read-add-write random
//   3: plain read-add-write sequential  4: ds_read only random  5: atomicAdd(int) random
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

template <int MODE>
__global__ void __launch_bounds__(256) k(const unsigned short* __restrict__ idx, int n_per_wave, int SC, float* out) {
  extern __shared__ float acc[];
  int* iacc = (int*)acc;
  for (int j = threadIdx.x; j < SC; j += blockDim.x) acc[j] = 0.f;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const unsigned short* p = idx + ((size_t)(blockIdx.x * (blockDim.x >> 6) + wave) % 4096) * 64;
  float s = 0.f;
  for (int i = 0; i < n_per_wave; i += 4) {
    int j[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      unsigned short v = p[((i + u) % 64) * 4096 * 64 / 64 * 0 + lane + ((i + u) & 63) * 64 * 0];
      j[u] = (MODE == 1 || MODE == 3) ? ((lane + (i + u) * 64) % SC) : (int)((v * 2654435761u + (unsigned)(i + u) * 40503u) >> 8) % SC;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (MODE == 0 || MODE == 1) atomicAdd(&acc[j[u]], 1.0f);
      else if (MODE == 2 || MODE == 3) acc[j[u]] += 1.0f;
      else if (MODE == 4) s += acc[j[u]];
      else if (MODE == 5) atomicAdd(&iacc[j[u]], 1);
      else if (MODE == 6) atomicAdd(&((double*)acc)[j[u] >> 1], 1.0);
      else if (MODE == 7) atomicAdd(&((unsigned long long*)acc)[j[u] >> 1], 1ull);
      else if (MODE == 8) { double* d = (double*)acc; d[j[u] >> 1] += 1.0; }
    }
  }
  __syncthreads();
  float t = s;
  for (int j = threadIdx.x; j < SC; j += blockDim.x) t += acc[j];
  if (t == 12345.678f) out[0] = t;
}

template <int MODE>
void run(const char* name, const unsigned short* idx, float* out, int SC, int threads) {
  const int blocks = 256 * 8, n = 4096;
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int rep = 0; rep < 2; ++rep) {
    CK(hipEventRecord(a));
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), SC * 4, 0, idx, n, SC, out);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  }
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  double waveops = (double)blocks * (threads / 64) * n;
  // cycles per wave-instruction per CU at 2.4 GHz, 256 CUs
  printf("%-34s SC=%5d thr=%4d  %8.3f ms  %7.1f Gops/s  %6.1f clk/wave-op/CU\n", name, SC, threads, ms,
         waveops * 64 / ms / 1e6, ms * 1e-3 * 2.4e9 * 256 / waveops);
}

int main() {
  unsigned short* idx; float* out;
  CK(hipMalloc(&idx, 4096 * 64 * 2)); CK(hipMalloc(&out, 4));
  unsigned short* h = (unsigned short*)malloc(4096 * 64 * 2);
  srand(1); for (int i = 0; i < 4096 * 64; ++i) h[i] = rand() & 0xffff;
  CK(hipMemcpy(idx, h, 4096 * 64 * 2, hipMemcpyHostToDevice));
  for (int SC : {1250, 10000}) {
    run<0>("atomicAdd f32 random", idx, out, SC, 256);
    run<1>("atomicAdd f32 sequential", idx, out, SC, 256);
    run<2>("read-add-write random (racy)", idx, out, SC, 256);
    run<3>("read-add-write sequential", idx, out, SC, 256);
    run<4>("read only random", idx, out, SC, 256);
    run<5>("atomicAdd i32 random", idx, out, SC, 256);
    run<6>("atomicAdd f64 random", idx, out, SC, 256);
    run<7>("atomicAdd u64 random", idx, out, SC, 256);
    run<8>("read-add-write f64 random", idx, out, SC, 256);
  }
  run<0>("atomicAdd f32 random, 1 wave WG", idx, out, 1250, 64);
  run<2>("read-add-write random, 1 wave WG", idx, out, 1250, 64);
  return 0;
}
