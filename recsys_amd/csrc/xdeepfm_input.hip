// xDeepFM's two input_layer calls and the pre-activation of its linear_net in ONE launch (xdeepfm/xdeepfm.py:125-131,185):
// the same ids index two independent embedding table sets (the CIN's and the DNN's), and the linear net is
// dense([13 log-values | one-hot indicator blocks], 1) = sum of the indicator weights of the example's rows + <log-values, w_num>.
// One wave per example, D/4 lanes per field row (the layout of gather_fm_fwd_k, embedding.hip): ids and offsets of a batch of 4
// fields first, then the rows of BOTH table sets and the first-order weights -- all unconditional on clamped field indices.
// HBM-bound: 4 B of id + 2 x D*4 B of rows read, 2 x D*4 B written per (example, field).
#include "rsx_common.h"

template <int D>
__global__ void gather_two_fwd_k(const float* __restrict__ tables1, const float* __restrict__ w1,
                                 const float* __restrict__ tables2, const int32_t* __restrict__ row_off,
                                 const int32_t* __restrict__ ids, const float* __restrict__ num_x,
                                 const float* __restrict__ num_w, float* __restrict__ E1, float* __restrict__ E2,
                                 float* __restrict__ y1, uint64_t w1_mask, int B, int F, int ND) {
  constexpr int LPR = D / 4;
  constexpr int PPP = RSX_WAVE / LPR;
  const int lane = threadIdx.x & 63;
  const int b = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (b >= B) return;
  const int q = lane % LPR, j = lane / LPR;
  const float4* __restrict__ T1 = reinterpret_cast<const float4*>(tables1);
  const float4* __restrict__ T2 = reinterpret_cast<const float4*>(tables2);
  float4* __restrict__ O1 = reinterpret_cast<float4*>(E1);
  float4* __restrict__ O2 = reinterpret_cast<float4*>(E2);
  const int32_t* idb = ids + (size_t)b * F;
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

template <int D>
static void launch_two(dim3 grid, dim3 block, hipStream_t st, const float* t1, const float* w1, const float* t2,
                       const int32_t* row_off, const int32_t* ids, const float* nx, const float* nw, float* E1, float* E2,
                       float* y1, uint64_t mask, int B, int F, int ND) {
  RSX_COUNT_LAUNCH(); gather_two_fwd_k<D><<<grid, block, 0, st>>>(t1, w1, t2, row_off, ids, nx, nw, E1, E2, y1, mask, B, F, ND);
}

extern "C" int rsx_gather_two_fwd(const float* tables1, const float* w1, const float* tables2, const int32_t* row_off,
                                  const int32_t* ids, const float* num_x, const float* num_w, float* E1, float* E2, float* y1,
                                  uint64_t w1_field_mask, int B, int F, int D, int ND, rsx_stream_t stream) {
  if (B < 0 || F <= 0 || F > 64 || ND <= 0 || ND > 64) return RSX_EINVAL;
  if (D != 4 && D != 8 && D != 16 && D != 32 && D != 64) return RSX_EINVAL;
  if (B == 0) return RSX_OK;
  if (!tables1 || !w1 || !tables2 || !row_off || !ids || !num_x || !num_w || !E1 || !E2 || !y1) return RSX_EINVAL;
  const int waves = B >= 2048 ? 4 : 1;  // small batches: one wave per workgroup spreads over all 256 CUs
  const dim3 grid((B + waves - 1) / waves), block(64 * waves);
  switch (D) {
    case 4: launch_two<4>(grid, block, rsx_s(stream), tables1, w1, tables2, row_off, ids, num_x, num_w, E1, E2, y1, w1_field_mask, B, F, ND); break;
    case 8: launch_two<8>(grid, block, rsx_s(stream), tables1, w1, tables2, row_off, ids, num_x, num_w, E1, E2, y1, w1_field_mask, B, F, ND); break;
    case 16: launch_two<16>(grid, block, rsx_s(stream), tables1, w1, tables2, row_off, ids, num_x, num_w, E1, E2, y1, w1_field_mask, B, F, ND); break;
    case 32: launch_two<32>(grid, block, rsx_s(stream), tables1, w1, tables2, row_off, ids, num_x, num_w, E1, E2, y1, w1_field_mask, B, F, ND); break;
    default: launch_two<64>(grid, block, rsx_s(stream), tables1, w1, tables2, row_off, ids, num_x, num_w, E1, E2, y1, w1_field_mask, B, F, ND); break;
  }
  RSX_CHECK_LAUNCH();
  return RSX_OK;
}
