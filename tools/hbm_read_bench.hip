// Microbenchmark: what a hand-written read-only stream reaches on this GPU (the scale for the narrow W*R kernel, whose
// operand is read exactly once).  16 bytes per lane (global_load_dwordx4), U loads in flight per lane, grid-stride over a
// buffer far larger than the 256 MiB Infinity Cache; a read + write copy for comparison.
//   hipcc -O3 --offload-arch=gfx950 tools/hbm_read_bench.hip -o tools/hbm_read_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));

template <int U>
__global__ void __launch_bounds__(256) read_kernel(const f4* __restrict__ p, size_t n4, float* out) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  f4 s = {0.f, 0.f, 0.f, 0.f};
  for (; i + (U - 1) * stride < n4; i += U * stride) {
    f4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = __builtin_nontemporal_load(p + i + u * stride);
#pragma unroll
    for (int u = 0; u < U; ++u) s += v[u];
  }
  for (; i < n4; i += stride) s += p[i];
  const float t = s.x + s.y + s.z + s.w;
  if (t == 12345.678f) out[0] = t;   // never true: keeps the loads alive
}

__global__ void __launch_bounds__(256) copy_kernel(const f4* __restrict__ p, f4* __restrict__ q, size_t n4) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) q[i] = p[i];
}

template <int U>
static void run_read(const f4* p, size_t n4, float* out, int blocks) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  float best = 1e30f;
  for (int rep = 0; rep < 5; ++rep) {
    CK(hipEventRecord(a));
    hipLaunchKernelGGL(read_kernel<U>, dim3(blocks), dim3(256), 0, 0, p, n4, out);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    best = ms < best ? ms : best;
  }
  printf("read  16 B/lane, %d loads in flight, %5d blocks: %8.3f ms  %6.2f TB/s\n", U, blocks, best, n4 * 16.0 / best / 1e9);
}

int main(int argc, char** argv) {
  const size_t bytes = (argc > 1 ? atoll(argv[1]) : 2048ll) << 20;   // MiB
  const size_t n4 = bytes / 16;
  f4 *p, *q; float* out;
  CK(hipMalloc(&p, bytes)); CK(hipMalloc(&q, bytes)); CK(hipMalloc(&out, 4));
  CK(hipMemset(p, 1, bytes)); CK(hipMemset(q, 0, bytes));
  printf("buffer %zu MiB\n", bytes >> 20);
  for (int blocks : {1024, 2048, 4096, 8192, 16384}) {
    run_read<1>(p, n4, out, blocks);
    run_read<4>(p, n4, out, blocks);
    run_read<8>(p, n4, out, blocks);
  }
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int blocks : {2048, 8192}) {
    float best = 1e30f;
    for (int rep = 0; rep < 5; ++rep) {
      CK(hipEventRecord(a));
      hipLaunchKernelGGL(copy_kernel, dim3(blocks), dim3(256), 0, 0, p, q, n4);
      CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
      float ms; CK(hipEventElapsedTime(&ms, a, b));
      best = ms < best ? ms : best;
    }
    printf("copy  16 B/lane, %5d blocks: %8.3f ms  %6.2f TB/s read + %6.2f TB/s write\n", blocks, best, n4 * 16.0 / best / 1e9, n4 * 16.0 / best / 1e9);
  }
  return 0;
}
