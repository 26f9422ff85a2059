// Practical fp32 MFMA ceiling on gfx950: register-only v_mfma_f32_16x16x4_f32 chains (no memory traffic).
// Build + run on the GPU box: hipcc --offload-arch=gfx950 -O3 scripts/micro/mfma_peak.hip -o /tmp/mfma_peak && /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int CH>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
  f32x4 acc[CH];
  for (int c = 0; c < CH; ++c) acc[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float a = threadIdx.x * 1e-3f, b = blockIdx.x * 1e-3f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int c = 0; c < CH; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[c], 0, 0, 0);
  }
  float s = 0.f;
  for (int c = 0; c < CH; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int CH>
void run(int wgs, int iters) {
  float* out;
  hipMalloc(&out, sizeof(float) * wgs * 256);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  k<CH><<<wgs, 256>>>(out, iters);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<CH><<<wgs, 256>>>(out, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double fl = (double)wgs * 4 * iters * 4 * CH * 2048.0;
  printf("chains=%d wgs=%d (%.1f waves/SIMD): %.3f ms  %.1f TFLOP/s\n", CH, wgs, wgs * 4 / 1024.0, ms, fl / ms / 1e9);
  hipFree(out);
}
int main() {
  run<1>(256, 20000); run<1>(512, 20000); run<1>(1024, 10000);
  run<2>(256, 20000); run<4>(256, 10000); run<4>(512, 10000); run<4>(1024, 5000);
  return 0;
}
