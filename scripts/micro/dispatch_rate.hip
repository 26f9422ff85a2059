// When does workgroup i of a launch START?  Every workgroup records the 100 MHz wall clock at entry (and at exit, after a
// chain of `hops` dependent loads): the entry times give the dispatcher's rate for workgroups of 64 / 256 / 1024 threads,
// with and without registers / LDS that limit the residency.  hipcc --offload-arch=gfx950 -O3 dispatch_rate.hip -o dispatch_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
template <int T, int LDSB>
__global__ __launch_bounds__(T) void probe(const int* __restrict__ chain, unsigned long long* __restrict__ t, int hops) {
  __shared__ int lds[LDSB / 4 + 1];
  unsigned long long t0 = wall_clock64();
  int p = (blockIdx.x * 97 + threadIdx.x) & 0xFFFF;
  for (int i = 0; i < hops; ++i) p = chain[p];
  if (LDSB > 4) { lds[threadIdx.x % (LDSB / 4)] = p; __syncthreads(); p += lds[0]; }
  unsigned long long t1 = wall_clock64();
  if (threadIdx.x == 0) { t[2 * blockIdx.x] = t0; t[2 * blockIdx.x + 1] = t1 + (p == 123456789 ? 1 : 0); }
}
template <int T, int LDSB>
void run(const char* what, int grid, const int* chain, unsigned long long* t_d, int hops) {
  std::vector<unsigned long long> t(2 * grid);
  std::vector<double> ent(grid), ext(grid);
  double acc_e[8] = {0}, acc_x = 0, acc_l = 0;
  const int reps = 20;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float ms_sum = 0;
  for (int r = 0; r < reps + 3; ++r) {
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((probe<T, LDSB>), dim3(grid), dim3(T), 0, 0, chain, t_d, hops);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    if (r < 3) continue;
    float ms; hipEventElapsedTime(&ms, e0, e1); ms_sum += ms;
    hipMemcpy(t.data(), t_d, 16 * grid, hipMemcpyDeviceToHost);
    unsigned long long t00 = t[0];
    for (int i = 0; i < grid; ++i) t00 = std::min(t00, t[2 * i]);
    double last_exit = 0, last_entry = 0;
    for (int i = 0; i < grid; ++i) { last_entry = std::max(last_entry, (double)(t[2 * i] - t00)); last_exit = std::max(last_exit, (double)(t[2 * i + 1] - t00)); }
    const int idx[8] = {0, grid / 8, grid / 4, grid / 2, 3 * grid / 4, grid - 1, 0, 0};
    for (int k = 0; k < 6; ++k) acc_e[k] += (double)(t[2 * idx[k]] - t00) * 0.01;
    acc_x += last_exit * 0.01; acc_l += last_entry * 0.01;
  }
  printf("%-28s grid %5d x %4d thr, %d hops: entry of wg 0 / N/8 / N/4 / N/2 / 3N/4 / N-1 = %5.2f %5.2f %5.2f %5.2f %5.2f %5.2f us | "
         "last entry %5.2f, last exit %5.2f us | launch (events) %.2f us\n", what, grid, T, hops, acc_e[0] / reps, acc_e[1] / reps,
         acc_e[2] / reps, acc_e[3] / reps, acc_e[4] / reps, acc_e[5] / reps, acc_l / reps, acc_x / reps, ms_sum * 1000 / reps);
}
int main() {
  const int N = 1 << 16;
  std::vector<int> c(N);
  for (int i = 0; i < N; ++i) c[i] = (int)(((long long)i * 40503 + 12345) & (N - 1));
  int* chain; unsigned long long* t;
  hipMalloc(&chain, N * 4); hipMemcpy(chain, c.data(), N * 4, hipMemcpyHostToDevice);
  hipMalloc(&t, 16 * 8192);
  for (int hops : {0, 6}) {
    for (int g : {128, 256, 512, 1024, 2048}) run<256, 4>("256 threads", g, chain, t, hops);
    for (int g : {512, 2048, 8192}) run<64, 4>("64 threads", g, chain, t, hops);
    for (int g : {32, 64, 128, 256, 512}) run<1024, 4>("1024 threads", g, chain, t, hops);
    for (int g : {256, 1024}) run<256, 65536>("256 threads + 64 KB LDS", g, chain, t, hops);
  }
  return 0;
}
