// How long does a launch of N workgroups take when almost all of them exit after one scalar load?  (Stage B of the
// two-stage scatter launches waves for the worst case -- every example a unique row in every field -- and most of them
// find nothing to do: 45 k workgroups at batch 65 536.)   hipcc --offload-arch=gfx950 -O3 empty_wg.hip -o empty_wg
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(256) void probe(const int* n, float* out, int live) {
  if ((int)blockIdx.x >= n[0] + live) return;          // n[0] == 0: only the first `live` workgroups do anything
  float s = 0.f;
  for (int i = 0; i < 64; ++i) s += out[(blockIdx.x * 256 + threadIdx.x + i * 4096) & 0xFFFFF];
  if (s == 12345.f) out[0] = s;
}
int main() {
  int* n; float* out;
  hipMalloc(&n, 4); hipMemset(n, 0, 4); hipMalloc(&out, 4 << 20); hipMemset(out, 0, 4 << 20);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int grids[] = {1, 1024, 2808, 11000, 45000, 160000};
  for (int live : {0, 2500}) for (int g : grids) {
    if (g < live) continue;
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(probe, dim3(g), dim3(256), 0, 0, n, out, live);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    for (int r = 0; r < 50; ++r) hipLaunchKernelGGL(probe, dim3(g), dim3(256), 0, 0, n, out, live);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("workgroups %6d (working: %4d): %.2f us per launch\n", g, live, ms * 1000.f / 50);
  }
  return 0;
}
