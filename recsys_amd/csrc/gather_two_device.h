// xDeepFM's two input_layer lookups + the linear_net pre-activation of ONE example by one wave (the body of gather_two_fwd_k,
// csrc/xdeepfm_input.hip; xdeepfm/xdeepfm.py:125-131,185) -- shared with csrc/cin_split.hip, whose filter-preparation launch can
// carry the lookup as extra workgroups (rsx_cin_split_prep_gather): both are short launches at the head of the step's chain.
#pragma once
#include "rsx_common.h"

struct GatherTwoArgs {
  const float* tables1;
  const float* w1;
  const float* tables2;
  const int32_t* row_off;
  const int32_t* ids;
  const float* num_x;
  const float* num_w;
  float* E1;
  float* E2;
  float* y1;
  uint64_t w1_mask;
  int B, F, ND;
};

// (scalar arguments, not the block: handed the struct -- by reference or by value -- the compiler kept a copy of it in scratch
// memory and re-read its fields after every store; the stand-alone launch went from 5 to 27 us)
template <int D>
__device__ __forceinline__ void gather_two_example(const float* __restrict__ tables1, const float* __restrict__ w1,
                                                   const float* __restrict__ tables2, const int32_t* __restrict__ row_off,
                                                   const int32_t* __restrict__ ids, const float* __restrict__ num_x,
                                                   const float* __restrict__ num_w, float* __restrict__ E1, float* __restrict__ E2,
                                                   float* __restrict__ y1, const uint64_t w1_mask, const int F, const int ND,
                                                   const int b, const int lane) {
  constexpr int LPR = D / 4;
  constexpr int PPP = RSX_WAVE / LPR;
  const int q = lane % LPR, j = lane / LPR;
  const float4* __restrict__ T1 = reinterpret_cast<const float4*>(tables1);
  const float4* __restrict__ T2 = reinterpret_cast<const float4*>(tables2);
  float4* __restrict__ O1 = reinterpret_cast<float4*>(E1);
  float4* __restrict__ O2 = reinterpret_cast<float4*>(E2);
  const int32_t* __restrict__ idb = ids + (size_t)b * F;
  // numeric part of the linear net: lanes 0 .. ND-1 hold one product each (ND <= 64), added in the final butterfly
  float a1 = 0.f;
  {
    const int c = lane < ND ? lane : 0;
    const float x = num_x[(size_t)b * ND + c], w = num_w[c];
    a1 = lane < ND ? x * w : 0.f;
  }
  for (int f0 = j; f0 < F; f0 += 4 * PPP) {
    int row[4];
    bool ok[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int f = f0 + k * PPP;
      ok[k] = f < F;
      const int fc = ok[k] ? f : F - 1;
      row[k] = row_off[fc] + idb[fc];
    }
    float4 e1[4], e2[4];
    float wv[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      e1[k] = T1[(size_t)row[k] * LPR + q];
      e2[k] = T2[(size_t)row[k] * LPR + q];
      wv[k] = w1[row[k]];
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int f = f0 + k * PPP;
      if (ok[k]) {
        O1[((size_t)b * F + f) * LPR + q] = e1[k];
        O2[((size_t)b * F + f) * LPR + q] = e2[k];
        if (q == 0 && ((w1_mask >> f) & 1ull)) a1 += wv[k];
      }
    }
  }
#pragma unroll
  for (int m = 1; m < RSX_WAVE; m <<= 1) a1 += __shfl_xor(a1, m);
  if (lane == 0) y1[b] = a1;
}

// (a macro: the fields are named on the kernel's own by-value parameter)
#define RSX_GATHER_TWO_EXAMPLE(D_, G, B_, LANE_)                                                                                    \
  gather_two_example<D_>((G).tables1, (G).w1, (G).tables2, (G).row_off, (G).ids, (G).num_x, (G).num_w, (G).E1, (G).E2, (G).y1,     \
                         (G).w1_mask, (G).F, (G).ND, (B_), (LANE_))
