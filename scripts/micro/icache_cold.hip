// What does straight-line code cost a wave that runs it ONCE?  Kernels of N dependent / independent FMAs with distinct
// literal constants (8 bytes each: nothing loops, every instruction is fetched), one wave per workgroup, 64 workgroups:
// time from a workgroup's entry to its exit (100 MHz wall clock) -> instructions per microsecond and the cold-fetch
// penalty (the same kernel launched twice in a row, and again after a different kernel ran in between).
// hipcc --offload-arch=gfx950 -O3 icache_cold.hip -o icache_cold
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int N, int ILP>
__global__ __launch_bounds__(64) void chain(float* out, unsigned long long* t, float seed) {
  unsigned long long t0 = wall_clock64();
  float a[ILP];
#pragma unroll
  for (int j = 0; j < ILP; ++j) a[j] = seed + threadIdx.x + j;
#pragma unroll
  for (int i = 0; i < N / ILP; ++i) {
#pragma unroll
    for (int j = 0; j < ILP; ++j) a[j] = __builtin_fmaf(a[j], 1.0001f + 1e-6f * (i * ILP + j), 0.5f + 1e-5f * (i * ILP + j));
  }
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < ILP; ++j) s += a[j];
  unsigned long long t1 = wall_clock64();
  if (s == 12345.678f) out[0] = s;
  if (threadIdx.x == 0) { t[2 * blockIdx.x] = t0; t[2 * blockIdx.x + 1] = t1; }
}
__global__ void other(float* out) { out[threadIdx.x + 1] = 1.f; }
template <int N, int ILP>
void run(float* out, unsigned long long* t_d) {
  const int G = 64;
  std::vector<unsigned long long> t(2 * G);
  auto once = [&]() {
    hipLaunchKernelGGL((chain<N, ILP>), dim3(G), dim3(64), 0, 0, out, t_d, 1.0f);
    hipDeviceSynchronize();
    hipMemcpy(t.data(), t_d, 16 * G, hipMemcpyDeviceToHost);
    double mx = 0, av = 0;
    for (int i = 0; i < G; ++i) { double d = (double)(t[2 * i + 1] - t[2 * i]) * 0.01; av += d / G; if (d > mx) mx = d; }
    return std::make_pair(av, mx);
  };
  once(); once();
  auto warm = once();
  hipLaunchKernelGGL(other, dim3(1), dim3(64), 0, 0, out); hipDeviceSynchronize();
  auto after_other = once();
  printf("N = %6d FMAs, %d independent chains (~%3d KB of code): entry->exit avg %.2f us max %.2f us (%.0f instr/us) | after another "
         "kernel: avg %.2f max %.2f us\n", N, ILP, N * 8 / 1024, warm.first, warm.second, N / warm.first, after_other.first, after_other.second);
}
int main() {
  float* out; unsigned long long* t;
  hipMalloc(&out, 4096); hipMalloc(&t, 16 * 64);
  run<256, 1>(out, t); run<1024, 1>(out, t); run<4096, 1>(out, t); run<16384, 1>(out, t);
  run<1024, 4>(out, t); run<4096, 4>(out, t); run<16384, 4>(out, t); run<65536, 4>(out, t);
  return 0;
}
