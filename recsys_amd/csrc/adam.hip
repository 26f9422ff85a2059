// TF-1.x AdamOptimizer on gfx950: one streaming launch over every variable segment of the model.
// Reference call site: tf.train.AdamOptimizer(lr).minimize(loss) fm/fm.py:162-163 (all five scripts).
// Semantics restated from TF 1.13/1.14: dense variables use the ApplyAdam
// kernel formula; embedding tables use the NON-lazy sparse path (_apply_sparse_shared): every row
// decays and moves each step, touched rows add their summed gradient first.  That makes this sweep
// the dominant HBM stream of a training step: 6 * 4 B per table element (read+write of var, m, v).
//
// Each lane owns one float4; workgroups own 1024 consecutive float4.  The touched-row gradient is
// *pulled* through the row->slot map written by rsx_field_sort, so no dense gradient buffer exists.
// alpha = lr*sqrt(1-b2^t)/(1-b1^t) is derived from device-resident beta powers that the last
// workgroup to finish advances ("epsilon-hat" form) -- no per-step host argument, graph-replayable.
#include "adam_device.h"

__global__ __launch_bounds__(ADAM_T) void adam_multi_k(const AdamArgs a) {
  const float b1p = a.state[0], b2p = a.state[1];
  adam_block(a, blockIdx.x);
  // ticket: the last workgroup to finish advances the beta powers for the next step
  __syncthreads();
  if (threadIdx.x == 0 && adam_arrive_last(a.state, a.total_blocks)) {
    a.state[0] = b1p * a.b1;
    a.state[1] = b2p * a.b2;
    reinterpret_cast<uint32_t*>(a.state)[3] += 1u;
  }
}

extern "C" int rsx_adam_state_init_h(float* state_h, float beta1, float beta2) {
  if (!state_h) return RSX_EINVAL;
  state_h[0] = beta1;
  state_h[1] = beta2;
  reinterpret_cast<uint32_t*>(state_h)[2] = 0u;
  reinterpret_cast<uint32_t*>(state_h)[3] = 1u;
  return RSX_OK;
}

extern "C" int rsx_adam_tf1_multi(const rsx_adam_seg* segs_h, int nseg, float* state, float lr, float beta1,
                                  float beta2, float eps, rsx_stream_t stream) {
  AdamArgs a;
  uint32_t blocks = 0;
  const int rc = adam_build_args(segs_h, nseg, state, lr, beta1, beta2, eps, a, &blocks);
  if (rc != RSX_OK) return rc;
  if (blocks == 0) return RSX_OK;
  hipLaunchKernelGGL(adam_multi_k, dim3(blocks), dim3(ADAM_T), 0, rsx_s(stream), a);
  RSX_CHECK_LAUNCH();
  return RSX_OK;
}

// Number of workgroups the given segment list occupies (to cut a COLD sweep into slices).
extern "C" int64_t rsx_adam_num_blocks(const rsx_adam_seg* segs_h, int nseg) {
  AdamArgs a;
  uint32_t blocks = 0;
  float dummy;
  const int rc = adam_build_args(segs_h, nseg, &dummy, 0.f, 0.f, 0.f, 0.f, a, &blocks);
  return rc != RSX_OK ? (int64_t)rc : (int64_t)blocks;
}

// The COLD kinds do not advance the beta powers (no ticket): a sweep cut into slices by the caller, run stand-alone.
__global__ __launch_bounds__(ADAM_T) void adam_slice_k(const AdamSlice s) { adam_block(s.args, s.blk_lo + blockIdx.x); }

// The sweep of an optimizer window (rsx_adam_seg.slot_w): 1 + nw updates per untouched row in one pass.
template <int NW>
__global__ __launch_bounds__(ADAM_T, 4) void adam_window_k(const AdamSlice s) {
  adam_window_block<NW>(s.args, s.blk_lo + blockIdx.x);
}

extern "C" int rsx_adam_slice_run(const rsx_adam_slice* slice_h, rsx_stream_t stream) {
  AdamSlice s;
  const int rc = adam_build_slice(slice_h, s);
  if (rc != RSX_OK) return rc;
  if (s.n_blk == 0) return RSX_OK;
  const dim3 grid(s.n_blk), block(ADAM_T);
  switch (s.args.nw) {
    case 0: hipLaunchKernelGGL(adam_slice_k, grid, block, 0, rsx_s(stream), s); break;
    case 1: hipLaunchKernelGGL(adam_window_k<1>, grid, block, 0, rsx_s(stream), s); break;
    case 2: hipLaunchKernelGGL(adam_window_k<2>, grid, block, 0, rsx_s(stream), s); break;
    case 3: hipLaunchKernelGGL(adam_window_k<3>, grid, block, 0, rsx_s(stream), s); break;
    case 4: hipLaunchKernelGGL(adam_window_k<4>, grid, block, 0, rsx_s(stream), s); break;
    case 5: hipLaunchKernelGGL(adam_window_k<5>, grid, block, 0, rsx_s(stream), s); break;
    case 6: hipLaunchKernelGGL(adam_window_k<6>, grid, block, 0, rsx_s(stream), s); break;
    default: hipLaunchKernelGGL(adam_window_k<7>, grid, block, 0, rsx_s(stream), s); break;
  }
  RSX_CHECK_LAUNCH();
  return RSX_OK;
}
