// xDeepFM's two input_layer calls and the pre-activation of its linear_net in ONE launch (xdeepfm/xdeepfm.py:125-131,185):
// the same ids index two independent embedding table sets (the CIN's and the DNN's), and the linear net is
// dense([13 log-values | one-hot indicator blocks], 1) = sum of the indicator weights of the example's rows + <log-values, w_num>.
// One wave per example, D/4 lanes per field row (the layout of gather_fm_fwd_k, embedding.hip): ids and offsets of a batch of 4
// fields first, then the rows of BOTH table sets and the first-order weights -- all unconditional on clamped field indices.
// HBM-bound: 4 B of id + 2 x D*4 B of rows read, 2 x D*4 B written per (example, field).
#include "rsx_common.h"
#include "gather_two_device.h"

template <int D>
__global__ void gather_two_fwd_k(const GatherTwoArgs g) {
  const int lane = threadIdx.x & 63;
  const int b = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (b >= g.B) return;
  RSX_GATHER_TWO_EXAMPLE(D, g, b, lane);
}

template <int D>
static void launch_two(dim3 grid, dim3 block, hipStream_t st, const GatherTwoArgs& g) {
  RSX_COUNT_LAUNCH(); gather_two_fwd_k<D><<<grid, block, 0, st>>>(g);
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
  const GatherTwoArgs g{tables1, w1, tables2, row_off, ids, num_x, num_w, E1, E2, y1, w1_field_mask, B, F, ND};
  switch (D) {
    case 4: launch_two<4>(grid, block, rsx_s(stream), g); break;
    case 8: launch_two<8>(grid, block, rsx_s(stream), g); break;
    case 16: launch_two<16>(grid, block, rsx_s(stream), g); break;
    case 32: launch_two<32>(grid, block, rsx_s(stream), g); break;
    default: launch_two<64>(grid, block, rsx_s(stream), g); break;
  }
  RSX_CHECK_LAUNCH();
  return RSX_OK;
}
