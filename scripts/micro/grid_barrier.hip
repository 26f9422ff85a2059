// What does a phase boundary cost INSIDE a launch against a kernel boundary, at the geometry of the batch-256 tower?
// G workgroups of T threads run P phases.  A phase: every workgroup publishes `words` floats (write-through sc1 stores), the
// grid synchronises on ONE monotonic counter (drain -> __syncthreads -> lane 0: relaxed agent atomic add, then a relaxed
// sc1-load poll with s_sleep; no fences: the payload is read back with sc1 loads, which bypass the CU's L1), then every
// workgroup reads 16 B from each of the other workgroups' payloads and checks the phase tag.  Against it: the same phase body
// as P dependent launches inside one HIP graph.  Per-phase microseconds, host events over graph replays.
//   hipcc --offload-arch=gfx950 -O3 grid_barrier.hip -o grid_barrier
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32;
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc_of(const void* p) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, 0x7fffffff, 0x00020000);
}
__device__ __forceinline__ void st_sc1(float* base, int off_bytes, f4 v) {
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((ext_vector_type(4))) u32, v), rsrc_of(base), off_bytes, 0, 16);
}
__device__ __forceinline__ f4 ld_sc1(const float* base, int off_bytes) {
  return __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_of(base), off_bytes, 0, 16));
}

__device__ __forceinline__ void phase_body_publish(float* pay, int words, int p) {
  // payload of this workgroup: `words` floats, value = phase tag
  for (int i = threadIdx.x * 4; i < words; i += blockDim.x * 4) {
    const f4 v = {(float)p, (float)blockIdx.x, (float)i, 1.f};
    st_sc1(pay + (size_t)blockIdx.x * words, i * 4, v);
  }
}
__device__ __forceinline__ int phase_body_consume(const float* pay, int words, int p, int G) {
  int bad = 0;
  for (int w = threadIdx.x; w < G; w += blockDim.x) {
    const int src = (w + blockIdx.x) % G;
    const f4 v = ld_sc1(pay + (size_t)src * words, 0);
    bad += (v.x != (float)p) || (v.y != (float)src);
  }
  return bad;
}

// V = 0: ONE counter, arrivals and polls on the same word.  V = 1: counter + a separate flag word 4 KB away that the last
// arriver (returned atomic) stores the epoch to: the polls do not queue in front of the arrivals.  V = 2: two-level -- groups
// of 16 consecutive workgroups arrive on their group's counter (256 B apart), each group's last arriver arrives on the top
// counter, the last of those stores the flag.  ctr layout (u32 words): [0] top, [64 * (1 + g)] group g, [1024 * 2] flag.
template <int V>
__device__ __forceinline__ void grid_sync(u32* ctr, u32 phase /* 1.. */) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    const u32 G = gridDim.x;
    u32* flag = ctr + 2048;
    if (V == 0) {
      __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      flag = ctr;
    } else if (V == 1) {
      if (__hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == G * phase - 1)
        __hip_atomic_store(flag, G * phase, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      const u32 g = blockIdx.x >> 4, ng = (G + 15) >> 4, gsz = min(16u, G - g * 16);
      if (__hip_atomic_fetch_add(ctr + 64 * (1 + g), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gsz * phase - 1)
        if (__hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == ng * phase - 1)
          __hip_atomic_store(flag, G * phase, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    unsigned spins = 0;
    while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < G * phase) {
      __builtin_amdgcn_s_sleep(1);
      if (++spins > (1u << 22)) break;      // bounded
    }
  }
  __syncthreads();
}

template <int V, bool BODY>
__global__ void persistent_k(float* pay0, float* pay1, u32* ctr, int* err, int words, int P, unsigned long long* t) {
  const unsigned long long t0 = wall_clock64();
  int bad = 0;
  for (int p = 0; p < P; ++p) {
    float* pay = (p & 1) ? pay1 : pay0;
    if (BODY) phase_body_publish(pay, words, p + 1);
    grid_sync<V>(ctr, (u32)(p + 1));
    if (BODY) bad += phase_body_consume(pay, words, p + 1, gridDim.x);
  }
  if (bad) atomicAdd(err, bad);
  if (threadIdx.x == 0 && blockIdx.x == 0) { t[0] = t0; t[1] = wall_clock64(); }
}
__global__ void publish_k(float* pay, int words, int p) { phase_body_publish(pay, words, p); }
__global__ void consume_publish_k(const float* payin, float* payout, int* err, int words, int p) {
  const int bad = phase_body_consume(payin, words, p, gridDim.x);
  if (bad) atomicAdd(err, bad);
  phase_body_publish(payout, words, p + 1);
}

int main() {
  const int P = 16, R = 200;
  float *pay0, *pay1; u32* ctr; int* err; unsigned long long* t;
  CHECK(hipMalloc(&pay0, 1024 * 4096 * 4)); CHECK(hipMalloc(&pay1, 1024 * 4096 * 4));
  CHECK(hipMalloc(&ctr, 16384)); CHECK(hipMalloc(&err, 4)); CHECK(hipMalloc(&t, 64));
  CHECK(hipMemset(err, 0, 4));
  hipStream_t s; CHECK(hipStreamCreate(&s));
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int V = 0; V < 6; ++V) for (int T : {256}) for (int G : {16, 56, 112, 128, 256}) for (int words : {256}) {
    // persistent: memset node + one launch per graph
    hipGraph_t g; hipGraphExec_t ge;
    CHECK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    CHECK(hipMemsetAsync(ctr, 0, 16384, s));
    switch (V) {
      case 0: hipLaunchKernelGGL((persistent_k<0, true>), dim3(G), dim3(T), 0, s, pay0, pay1, ctr, err, words, P, t); break;
      case 1: hipLaunchKernelGGL((persistent_k<1, true>), dim3(G), dim3(T), 0, s, pay0, pay1, ctr, err, words, P, t); break;
      case 2: hipLaunchKernelGGL((persistent_k<2, true>), dim3(G), dim3(T), 0, s, pay0, pay1, ctr, err, words, P, t); break;
      case 3: hipLaunchKernelGGL((persistent_k<0, false>), dim3(G), dim3(T), 0, s, pay0, pay1, ctr, err, words, P, t); break;
      case 4: hipLaunchKernelGGL((persistent_k<1, false>), dim3(G), dim3(T), 0, s, pay0, pay1, ctr, err, words, P, t); break;
      case 5: hipLaunchKernelGGL((persistent_k<2, false>), dim3(G), dim3(T), 0, s, pay0, pay1, ctr, err, words, P, t); break;
    }
    CHECK(hipStreamEndCapture(s, &g)); CHECK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int w = 0; w < 5; ++w) CHECK(hipGraphLaunch(ge, s));
    CHECK(hipStreamSynchronize(s));
    hipEventRecord(e0, s);
    for (int r = 0; r < R; ++r) CHECK(hipGraphLaunch(ge, s));
    hipEventRecord(e1, s); CHECK(hipEventSynchronize(e1));
    float ms_p; hipEventElapsedTime(&ms_p, e0, e1);
    unsigned long long th[2]; CHECK(hipMemcpy(th, t, 16, hipMemcpyDeviceToHost));
    // launches: P dependent kernels per graph
    hipGraph_t g2; hipGraphExec_t ge2;
    CHECK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    hipLaunchKernelGGL(publish_k, dim3(G), dim3(T), 0, s, pay0, words, 1);
    for (int p = 1; p <= P; ++p)
      hipLaunchKernelGGL(consume_publish_k, dim3(G), dim3(T), 0, s, (p & 1) ? pay0 : pay1, (p & 1) ? pay1 : pay0, err, words, p);
    CHECK(hipStreamEndCapture(s, &g2)); CHECK(hipGraphInstantiate(&ge2, g2, nullptr, nullptr, 0));
    for (int w = 0; w < 5; ++w) CHECK(hipGraphLaunch(ge2, s));
    CHECK(hipStreamSynchronize(s));
    hipEventRecord(e0, s);
    for (int r = 0; r < R; ++r) CHECK(hipGraphLaunch(ge2, s));
    hipEventRecord(e1, s); CHECK(hipEventSynchronize(e1));
    float ms_l; hipEventElapsedTime(&ms_l, e0, e1);
    int bad; CHECK(hipMemcpy(&bad, err, 4, hipMemcpyDeviceToHost));
    printf("barrier V%d%s T %4d G %3d payload %5d B/wg: persistent %.2f us per graph, in-kernel %.2f us per phase | launches %.2f us per graph = %.2f per launch | mismatches %d\n",
           V % 3, V >= 3 ? " (barrier only)" : "", T, G, words * 4, ms_p * 1000 / R, (double)(th[1] - th[0]) * 0.01 / P, ms_l * 1000 / R, ms_l * 1000 / R / (P + 1), bad);
    CHECK(hipMemset(err, 0, 4));
    hipGraphExecDestroy(ge); hipGraphDestroy(g); hipGraphExecDestroy(ge2); hipGraphDestroy(g2);
  }
  return 0;
}
