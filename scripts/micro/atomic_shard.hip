// Micro-benchmark (round 6): what do non-returning 64-bit global atomics cost when 256 workgroups add their 100-column partial
// sums (4 limbs each) into SHARED accumulator rows?  shards = 1 (one row), 8 (row = XCC_ID), 32 (row = blockIdx & 31).
// build: hipcc --offload-arch=gfx950 -O3 scripts/micro/atomic_shard.hip -o /tmp/atomic_shard && /tmp/atomic_shard
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ __launch_bounds__(256) void k(unsigned long long* acc, int ncol, int limbs, int shards, int mode, float* sink) {
  // some work first so that the workgroups do not arrive in lock step: a short dependent chain
  float v = threadIdx.x * 1e-3f;
  for (int i = 0; i < 200 + (blockIdx.x % 7) * 20; ++i) v = v * 1.0001f + 1e-4f;
  int row = 0;
  if (mode == 1) row = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 7;
  if (mode == 2) row = blockIdx.x % shards;
  const int n = ncol * limbs;
  unsigned long long* a = acc + (size_t)row * n;
  for (int e = threadIdx.x; e < n; e += 256) atomicAdd(a + e, (unsigned long long)(e + 1));
  if (v == 12345.f) sink[0] = v;
}
int main() {
  unsigned long long* acc; float* sink;
  hipMalloc(&acc, 64 * 4096 * 8); hipMalloc(&sink, 4);
  hipMemset(acc, 0, 64 * 4096 * 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  struct C { const char* name; int ncol, limbs, shards, mode; } cs[] = {
    {"no atomics (ncol 0)", 0, 4, 1, 0}, {"1 row, 100 cols x 4 limbs", 100, 4, 1, 0}, {"8 rows by XCC_ID, 100 x 4", 100, 4, 8, 1},
    {"32 rows by block, 100 x 4", 100, 4, 32, 2}, {"8 rows by XCC_ID, 100 x 2", 100, 2, 8, 1}, {"32 rows by block, 100 x 2", 100, 2, 32, 2},
    {"1 row, 100 x 2", 100, 2, 1, 0}, {"64 rows by block, 100 x 2", 100, 2, 64, 2}};
  for (auto& c : cs) {
    for (int w = 0; w < 3; ++w) k<<<256, 256>>>(acc, c.ncol, c.limbs, c.shards, c.mode, sink);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    const int reps = 200;
    for (int r = 0; r < reps; ++r) k<<<256, 256>>>(acc, c.ncol, c.limbs, c.shards, c.mode, sink);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-34s %.2f us per launch (back-to-back launches, 256 WGs x 256 threads)\n", c.name, ms * 1e3f / reps);
  }
  return 0;
}
