// LDS scatter cost with indices held in registers (no index arithmetic in the loop).
// Reports LDS-pipeline-bound cycles per 64-lane operation per CU for each accumulate flavour.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

template <int MODE>
__global__ void __launch_bounds__(256) k(const int* __restrict__ idx, int iters, int SC, float* out) {
  extern __shared__ __align__(16) unsigned char smem[];
  float* acc = (float*)smem;
  for (int j = threadIdx.x; j < SC * 2; j += blockDim.x) acc[j] = 0.f;
  __syncthreads();
  int j[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) j[u] = idx[(blockIdx.x * 8 + u) * 256 + threadIdx.x] % SC;
  float s = 0.f;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (MODE == 0) atomicAdd(&acc[j[u]], 1.0f);
      else if (MODE == 1) acc[j[u]] += 1.0f;                                       // dependent chain per op
      else if (MODE == 2) s += acc[j[u]];
      else if (MODE == 3) atomicAdd(&((int*)acc)[j[u]], 1);
      else if (MODE == 4) atomicAdd(&((double*)acc)[j[u]], 1.0);
      else if (MODE == 5) atomicAdd(&((unsigned long long*)acc)[j[u]], 1ull);
      else if (MODE == 6) acc[j[u]] = 1.0f;                                       // write only
    }
  }
  __syncthreads();
  float t = s;
  for (int q = threadIdx.x; q < SC * 2; q += blockDim.x) t += acc[q];
  if (t == 12345.678f) out[0] = t;
}

template <int MODE>
void run(const char* name, const int* idx, float* out, int SC) {
  const int blocks = 256 * 8, iters = 512, threads = 256;
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  float ms = 0;
  for (int rep = 0; rep < 2; ++rep) {
    CK(hipEventRecord(a));
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), SC * 8, 0, idx, iters, SC, out);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    CK(hipEventElapsedTime(&ms, a, b));
  }
  double waveops = (double)blocks * (threads / 64) * iters * 8;
  printf("%-34s SC=%5d  %8.3f ms  %6.2f clk/wave-op/CU (at 2.4 GHz)\n", name, SC, ms, ms * 1e-3 * 2.4e9 * 256 / waveops);
}

int main() {
  int* idx; float* out;
  const int n = 256 * 8 * 8 * 256;
  CK(hipMalloc(&idx, n * 4)); CK(hipMalloc(&out, 4));
  int* h = (int*)malloc(n * 4);
  srand(1); for (int i = 0; i < n; ++i) h[i] = rand();
  CK(hipMemcpy(idx, h, n * 4, hipMemcpyHostToDevice));
  for (int SC : {1250, 10000}) {
    run<0>("ds_add_f32 (atomic)", idx, out, SC);
    run<1>("read-add-write chain f32", idx, out, SC);
    run<2>("ds_read_b32 only", idx, out, SC);
    run<6>("ds_write_b32 only", idx, out, SC);
    run<3>("ds_add_u32 (atomic)", idx, out, SC);
    run<4>("ds_add_f64 (atomic)", idx, out, SC);
    run<5>("ds_add_u64 (atomic)", idx, out, SC);
  }
  return 0;
}
