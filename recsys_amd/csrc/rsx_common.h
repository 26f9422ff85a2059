// Shared helpers for the gfx950 kernels of librsx.so (wave = 64 lanes, hard-coded).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rsx.h"

#define RSX_WAVE 64

#define RSX_CHECK_LAUNCH()                                  \
  do {                                                      \
    if (hipGetLastError() != hipSuccess) return RSX_ELAUNCH; \
  } while (0)

static inline hipStream_t rsx_s(rsx_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

// Every kernel launch of the library is counted (a relaxed process-wide counter, read by rsx_dbg_launch_count): bench.py
// reports launches per step from it -- at batch 256 the step IS the sum of its dependent launches.
#include "rsx_launch_count.h"
#define RSX_COUNT_LAUNCH() rsx_launches_g.fetch_add(1ull, std::memory_order_relaxed)
#define RSX_LAUNCH(...)                \
  do {                                 \
    RSX_COUNT_LAUNCH();                \
    hipLaunchKernelGGL(__VA_ARGS__);   \
  } while (0)

// A zero float4 is always this literal.  A NAMED zero (`const float4 z = make_float4(0, 0, 0, 0)`) that is selected against
// (`ok ? load : z`) is placed in scratch memory by this toolchain; every use reloads it and waits with `s_waitcnt
// vmcnt(0)` -- i.e. for every load AND store in flight (found with phase stamps in segsum_tiles_k: 5.6 us for 16
// position sums; the literal is rematerialised in registers).
#define F4Z make_float4(0.f, 0.f, 0.f, 0.f)

__device__ __forceinline__ float4 f4_add(float4 a, float4 b) {
  return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
}
__device__ __forceinline__ float4 f4_sub(float4 a, float4 b) {
  return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w);
}
__device__ __forceinline__ float4 f4_mul(float4 a, float4 b) {
  return make_float4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w);
}
__device__ __forceinline__ float4 f4_scale(float s, float4 a) {
  return make_float4(s * a.x, s * a.y, s * a.z, s * a.w);
}
__device__ __forceinline__ float4 f4_shfl_xor(float4 a, int m) {
  return make_float4(__shfl_xor(a.x, m), __shfl_xor(a.y, m), __shfl_xor(a.z, m), __shfl_xor(a.w, m));
}

// Profiling aid (NOT in the product build): -DRSX_STAMPS makes selected workgroups record the 100 MHz wall clock at phase
// boundaries into a per-translation-unit device array, read back by rsx_dbg_stamps_<unit>(); scripts/stamp_probe.py
// builds that variant into scripts/_build/ and prints where a latency-bound kernel spends its microseconds.
#ifdef RSX_STAMPS
static __device__ unsigned long long rsx_stamps_d[64];     // one array per translation unit
#define RSX_STAMP_DECL
#define RSX_STAMP(slot, cond)                                                        \
  do {                                                                               \
    if ((cond) && threadIdx.x == 0) rsx_stamps_d[slot] = wall_clock64();             \
  } while (0)
// latest time any workgroup passed this point (the probe zeroes the slot before the launch)
#define RSX_STAMP_MAX(slot, cond)                                                                 \
  do {                                                                                            \
    if ((cond) && threadIdx.x == 0) atomicMax(&rsx_stamps_d[slot], (unsigned long long)wall_clock64()); \
  } while (0)
#else
#define RSX_STAMP_DECL
#define RSX_STAMP(slot, cond) \
  do {                        \
  } while (0)
#define RSX_STAMP_MAX(slot, cond) \
  do {                            \
  } while (0)
#endif
