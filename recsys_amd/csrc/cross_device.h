// The forward of the DCN cross layers for ONE example by one wave (dcn/dcn.py:132-142: x_{l+1} = (x_l . w_l) * x0 + x_l + b_l), shared by
// cross_fwd_k (cross.hip) and gather_cross_fwd_k (embedding.hip, round 4: the input_layer lookup and the cross layers in one launch --
// the gather's lane layout IS this one: lane l holds float4 #(l + 64 v) of the example's row).  One body: the same bits either way.
#pragma once
#include "rsx_common.h"

constexpr int CROSS_MAX_L = 8;
constexpr int CROSS_NV = 4;   // float4 per lane -> dim <= 1024

// float4 #e of a row of n4 float4s; lanes past the row read the last element (in range) and get zero by multiplication:
// a guarded load is compiled into a branch of its own and the vectors of a lane then load one after the other
__device__ __forceinline__ float4 cross_ld(const float4* __restrict__ row, int e, int n4) {
  const float4 v = row[e < n4 ? e : n4 - 1];
  const float f = e < n4 ? 1.f : 0.f;
  return make_float4(v.x * f, v.y * f, v.z * f, v.w * f);
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int m = 1; m < 64; m <<= 1) v += __shfl_xor(v, m);
  return v;
}
__device__ __forceinline__ float dot4(float4 a, float4 b) { return (a.x * b.x + a.y * b.y) + (a.z * b.z + a.w * b.w); }

// x0[v] = float4 #(lane + 64 v) of the example's input row (zero past the row).  s_row: [L] of this example; xL_row (nullable):
// [dim]; cz (nullable, with wout): the logit contribution <x_L, wout>.
__device__ __forceinline__ void cross_fwd_wave(const float4 (&x0)[CROSS_NV], const float* __restrict__ W, const float* __restrict__ Bc,
                                               const float* __restrict__ wout, float* __restrict__ s_row, float* __restrict__ xL_row,
                                               float* __restrict__ cz, const int dim, const int L, const int lane) {
  const int n4 = dim >> 2;
  float4 x[CROSS_NV];
#pragma unroll
  for (int v = 0; v < CROSS_NV; ++v) x[v] = x0[v];
  for (int l = 0; l < L; ++l) {
    float4 w[CROSS_NV], bb[CROSS_NV];
    float part = 0.f;
#pragma unroll
    for (int v = 0; v < CROSS_NV; ++v) {
      const int e = lane + 64 * v;
      w[v] = cross_ld(reinterpret_cast<const float4*>(W) + (size_t)l * n4, e, n4);
      bb[v] = cross_ld(reinterpret_cast<const float4*>(Bc) + (size_t)l * n4, e, n4);
      part += dot4(x[v], w[v]);
    }
    const float s = wave_sum(part);
    if (lane == 0) s_row[l] = s;
#pragma unroll
    for (int v = 0; v < CROSS_NV; ++v) x[v] = f4_add(f4_add(f4_scale(s, x0[v]), x[v]), bb[v]);
  }
  float part = 0.f;
#pragma unroll
  for (int v = 0; v < CROSS_NV; ++v) {
    const int e = lane + 64 * v;
    if (e < n4) {
      if (xL_row != nullptr) reinterpret_cast<float4*>(xL_row)[e] = x[v];
      if (wout != nullptr) part += dot4(x[v], reinterpret_cast<const float4*>(wout)[e]);
    }
  }
  if (cz != nullptr) {
    const float c = wave_sum(part);
    if (lane == 0) cz[0] = c;
  }
}
