// What does a kernel pay at its END for the data it wrote?  G workgroups write `mb` MB in total (plain or nontemporal
// stores) and exit; HIP events around back-to-back launches give the duration per launch, the in-kernel stamps (100 MHz
// wall clock) the time from the first workgroup's entry to the last one's exit: the difference grows with the dirty bytes the
// end-of-kernel release has to write back from the per-XCD L2s.   hipcc --offload-arch=gfx950 -O3 flush_tail.hip -o flush_tail
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
template <bool NT>
__global__ __launch_bounds__(256) void wr(float4* __restrict__ out, size_t n4, unsigned long long* __restrict__ t) {
  const unsigned long long t0 = wall_clock64();
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    typedef float nt4 __attribute__((ext_vector_type(4)));
    const nt4 v = {(float)i, 1.f, 2.f, 3.f};
    if (NT) __builtin_nontemporal_store(v, reinterpret_cast<nt4*>(&out[i])); else *reinterpret_cast<nt4*>(&out[i]) = v;
  }
  __syncthreads();
  if (threadIdx.x == 0) { t[2 * blockIdx.x] = t0; t[2 * blockIdx.x + 1] = wall_clock64(); }
}
template <bool NT>
void run(float4* out, unsigned long long* t_d, double mb, int G) {
  const size_t n4 = (size_t)(mb * 1e6 / 16);
  std::vector<unsigned long long> t(2 * G);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(wr<NT>, dim3(G), dim3(256), 0, 0, out, n4, t_d);
  hipDeviceSynchronize();
  const int R = 50;
  hipEventRecord(e0, 0);
  for (int r = 0; r < R; ++r) hipLaunchKernelGGL(wr<NT>, dim3(G), dim3(256), 0, 0, out, n4, t_d);
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  hipMemcpy(t.data(), t_d, 16 * G, hipMemcpyDeviceToHost);
  unsigned long long lo = ~0ull, hi = 0;
  for (int i = 0; i < G; ++i) { lo = std::min(lo, t[2 * i]); hi = std::max(hi, t[2 * i + 1]); }
  printf("%-12s %6.2f MB by %4d workgroups: %.2f us per launch (events, back to back), %.2f us first entry -> last exit\n",
         NT ? "nontemporal" : "plain", mb, G, ms * 1000 / R, (double)(hi - lo) * 0.01);
}
int main() {
  float4* out; unsigned long long* t;
  hipMalloc(&out, 64 << 20); hipMalloc(&t, 16 * 4096);
  for (double mb : {0.0, 0.5, 2.0, 8.0, 32.0}) { run<false>(out, t, mb, 512); run<true>(out, t, mb, 512); }
  return 0;
}
