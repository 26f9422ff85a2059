// The input_layer lookup of ONE example by one wave (fm/fm.py:117-129: embedding rows, first-order sum, FM term), shared
// by every launch that performs it: gather_fm_fwd_k / gather_fm_sort_k (csrc/embedding.hip) and the gather workgroups of
// tower_gather_fwd_k (csrc/tower.hip).  One body = one lane mapping and one summation order: E, S, y1, y2 are the same bits
// whichever launch produced them.
#pragma once
#include "rsx_common.h"

// lane = (j, q): q = float4 quarter of the row, j = pair slot; the wave walks fields f = j, j + PPP, ...  The field
// reductions (S, sum of squares, first-order sum) are xor-butterflies over the j bits.
// y1v (valid in the lanes of quarter q == 0, e.g. lane 0) / y2v (valid in every lane): the first-order sum and the FM term, for
// a caller that goes on with them (gather_fm_head_k); E may be NULL (fm.py's TRAIN step never reads it).
template <int D>
__device__ __forceinline__ void gather_fm_example(const float* __restrict__ tables, const float* __restrict__ w1,
                                                  const int32_t* __restrict__ row_off, const int32_t* __restrict__ ids,
                                                  float* __restrict__ E, float* __restrict__ S, float* __restrict__ y1,
                                                  float* __restrict__ y2, const uint64_t w1_mask, const int b, const int F,
                                                  const int lane, float* y1v = nullptr, float* y2v = nullptr) {
  constexpr int LPR = D / 4;
  constexpr int PPP = RSX_WAVE / LPR;
  const int q = lane % LPR, j = lane / LPR;
  const float4* __restrict__ T4 = reinterpret_cast<const float4*>(tables);
  float4* __restrict__ E4 = reinterpret_cast<float4*>(E);
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f), qq = s;
  float a1 = 0.f;
  const int32_t* idb = ids + (size_t)b * F;
  // The lane's fields f = j, j + PPP, ... in batches of 4: ids and offsets of the batch first, then its row loads, all
  // unconditional on clamped field indices (one dependent chain per batch instead of one per field; a load behind a
  // per-lane guard is compiled into a branch of its own).  Sums run in ascending f.
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
    float4 e[4];
    float wv[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      e[k] = T4[(size_t)row[k] * LPR + q];
      wv[k] = w1 != nullptr ? w1[row[k]] : 0.f;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int f = f0 + k * PPP;
      if (ok[k]) {
        if (E != nullptr) E4[((size_t)b * F + f) * LPR + q] = e[k];
        s = f4_add(s, e[k]);
        qq = f4_add(qq, f4_mul(e[k], e[k]));
        if (q == 0 && ((w1_mask >> f) & 1ull)) a1 += wv[k];
      }
    }
  }
#pragma unroll
  for (int m = LPR; m < RSX_WAVE; m <<= 1) {
    s = f4_add(s, f4_shfl_xor(s, m));
    qq = f4_add(qq, f4_shfl_xor(qq, m));
    a1 += __shfl_xor(a1, m);
  }
  if (S != nullptr && j == 0) reinterpret_cast<float4*>(S)[(size_t)b * LPR + q] = s;
  if (y2 != nullptr || y2v != nullptr) {
    float t = ((s.x * s.x - qq.x) + (s.y * s.y - qq.y)) + ((s.z * s.z - qq.z) + (s.w * s.w - qq.w));
#pragma unroll
    for (int m = 1; m < LPR; m <<= 1) t += __shfl_xor(t, m);
    if (y2 != nullptr && lane == 0) y2[b] = 0.5f * t;
    if (y2v != nullptr) *y2v = 0.5f * t;
  }
  if (y1 != nullptr && lane == 0) y1[b] = a1;
  if (y1v != nullptr) *y1v = a1;
}
