// Counter-based dropout shared by the fused tower (tower.hip) and the fused DIN attention MLP (din_attn.hip).
#pragma once
#include "rsx_common.h"

// counter-based dropout keep-mask: lowbias32-style hash of (seed, step, layer, element).  The same
// function is evaluated wherever the mask is needed (next layer's A-load, head, backward), so no mask
// buffer exists unless the caller injects one (parity tests).
__device__ __forceinline__ uint32_t rsx_hash32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}
struct DropRng {
  uint32_t key;        // mixes seed, step and layer
  uint32_t thresh;     // drop when hash < thresh  (thresh = rate * 2^32)
  float inv_keep;
  int mode;            // 0: no dropout, 1: explicit mask buffer, 2: RNG
};
__device__ __forceinline__ DropRng drop_make(float rate, const float* mask, const uint32_t* step, uint32_t seed,
                                             uint32_t layer) {
  DropRng d;
  d.inv_keep = 1.0f / (1.0f - rate);
  d.mode = rate == 0.f ? 0 : (mask != nullptr ? 1 : 2);
  d.thresh = (uint32_t)((double)rate * 4294967296.0);
  const uint32_t st = step != nullptr ? step[0] : 0u;
  d.key = rsx_hash32(seed ^ (st * 0x9E3779B9u) ^ (layer * 0x85EBCA6Bu + 0x27220A95u));
  return d;
}
// multiplier (0 or inv_keep, 1 when dropout is off) for element idx = b*N + c of the layer's output
__device__ __forceinline__ float drop_mul(const DropRng& d, const float* mask, size_t idx) {
  if (d.mode == 0) return 1.f;
  if (d.mode == 1) return mask[idx] * d.inv_keep;
  return rsx_hash32((uint32_t)idx ^ d.key) < d.thresh ? 0.f : d.inv_keep;
}
