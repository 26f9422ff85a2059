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
  RSX_LAUNCH(adam_multi_k, dim3(blocks), dim3(ADAM_T), 0, rsx_s(stream), a);
  RSX_CHECK_LAUNCH();
  return RSX_OK;
}

// Number of workgroups the given segment list occupies (to cut a COLD sweep into slices).
extern "C" int64_t rsx_adam_num_blocks(const rsx_adam_seg* segs_h, int nseg) {
  return rsx_adam_num_blocks_u(segs_h, nseg, 0);
}
extern "C" int64_t rsx_adam_num_blocks_u(const rsx_adam_seg* segs_h, int nseg, int window_block_u) {
  AdamArgs a;
  uint32_t blocks = 0;
  float dummy;
  const int rc = adam_build_args(segs_h, nseg, &dummy, 0.f, 0.f, 0.f, 0.f, a, &blocks, window_block_u);
  return rc != RSX_OK ? (int64_t)rc : (int64_t)blocks;
}

// The COLD kinds do not advance the beta powers (no ticket): a sweep cut into slices by the caller, run stand-alone.
__global__ __launch_bounds__(ADAM_T) void adam_slice_k(const AdamSlice s) { adam_block(s.args, s.blk_lo + blockIdx.x); }

// The sweep of an optimizer window (rsx_adam_seg.slot_w): 1 + nw updates per untouched row in one pass.
#ifndef RSX_ADAM_WIN_OCC
#define RSX_ADAM_WIN_OCC 3      // waves per SIMD the window sweep is compiled for.  3 = 168 registers, no spills at NW = 7; 4 (128 registers, spills at NW = 7) measured equal up to 6-step windows and 3 us slower per 8-step sweep (76.6 vs 73.5 us; DeepFM step 0.0668 -> 0.0659 ms; scripts/ab_window_occ.sh, profiles/r03_q_ab_window_occ.txt)
#endif
template <int NW>
__global__ __launch_bounds__(ADAM_T, RSX_ADAM_WIN_OCC) void adam_window_k(const AdamSlice s) {
  // (the window's step sizes for the lazy window pass: state words 8.., adam_device.h)
  if (blockIdx.x == 0 && threadIdx.x == 0 && s.blk_lo == 0 && !s.args.alpha_src) adam_publish_step_sizes<NW>(s.args);
  adam_window_block<NW>(s.args, s.blk_lo + blockIdx.x);
}
template <int NW>
static void launch_window(const AdamSlice& s, hipStream_t st) {
  RSX_LAUNCH(adam_window_k<NW>, dim3(s.n_blk), dim3(ADAM_T), 0, st, s);
}

extern "C" int rsx_adam_slice_run(const rsx_adam_slice* slice_h, rsx_stream_t stream) {
  AdamSlice s;
  const int rc = adam_build_slice(slice_h, s);
  if (rc != RSX_OK) return rc;
  if (s.n_blk == 0) return RSX_OK;
  if (s.args.win_u != 0) return RSX_EUNSUPPORTED;      // small window blocks exist as riders only (rsx_tower_head)
  const dim3 grid(s.n_blk), block(ADAM_T);
  switch (s.args.nw) {
    case 0: RSX_LAUNCH(adam_slice_k, grid, block, 0, rsx_s(stream), s); break;
    case 1: launch_window<1>(s, rsx_s(stream)); break;
    case 2: launch_window<2>(s, rsx_s(stream)); break;
    case 3: launch_window<3>(s, rsx_s(stream)); break;
    case 4: launch_window<4>(s, rsx_s(stream)); break;
    case 5: launch_window<5>(s, rsx_s(stream)); break;
    case 6: launch_window<6>(s, rsx_s(stream)); break;
    default: launch_window<7>(s, rsx_s(stream)); break;
  }
  RSX_CHECK_LAUNCH();
  return RSX_OK;
}

// ---- self-test of adam_fast.h (tests/test_gpu_fast_math.py): the packed forms against the compiler's sqrtf and '/' ---------
namespace {
__device__ __forceinline__ uint32_t st_mix(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}
// every float with bits in [lo, hi): one mismatch counted per differing value
__global__ void st_sqrt_k(const uint32_t lo, const uint32_t hi, unsigned long long* bad) {
  const uint32_t n = hi - lo;
  unsigned long long mine = 0;
  for (uint64_t i = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * 2u; i < n; i += (uint64_t)gridDim.x * blockDim.x * 2u) {
    const uint32_t b0 = lo + (uint32_t)i, b1 = i + 1 < n ? b0 + 1 : b0;
    const float x0 = __uint_as_float(b0), x1 = __uint_as_float(b1);
    const rsx_f2 s = rsx_sqrt2_fast((rsx_f2){x0, x1});
    mine += __float_as_uint(s.x) != __float_as_uint(sqrtf(x0));
    mine += b1 != b0 && __float_as_uint(s.y) != __float_as_uint(sqrtf(x1));
  }
  if (mine) atomicAdd(bad, mine);
}
// pseudo-random pairs over the whole domain of rsx_div2_fast, exponents uniform, mantissas random / all-zero / all-one
__global__ void st_div_k(const uint32_t seed, const int iters, unsigned long long* bad) {
  uint32_t s = st_mix(seed ^ (blockIdx.x * blockDim.x + threadIdx.x) * 0x9e3779b9u);
  unsigned long long mine = 0;
  for (int it = 0; it < iters; ++it) {
    float n[2], d[2];
    for (int k = 0; k < 2; ++k) {
      s = st_mix(s + 0x632be5abu);
      const uint32_t en = 127 - 94 + s % (94 + 34 + 1);
      s = st_mix(s + 1);
      const uint32_t ed = 127 - 30 + s % (30 + 21 + 1);
      s = st_mix(s + 2);
      uint32_t mn = s & 0x7fffffu;
      const uint32_t sg = s >> 31;
      if ((s >> 23 & 31u) == 0u) mn = (s >> 28) & 1u ? 0x7fffffu : 0u;
      if (en == 127u + 34u) mn = 0u;                                   // |n| <= 2^34
      s = st_mix(s + 3);
      uint32_t md = s & 0x7fffffu;
      if ((s >> 23 & 15u) == 0u) md = (s >> 27) & 1u ? 0x7fffffu : 0u;
      if (ed == 127u + 21u) md = 0u;                                   // d <= 2^21
      n[k] = __uint_as_float((sg << 31) | (en << 23) | mn);
      if ((s >> 28) == 0u) n[k] = 0.f;
      d[k] = __uint_as_float((ed << 23) | md);
    }
    const rsx_f2 q = rsx_div2_fast((rsx_f2){n[0], n[1]}, (rsx_f2){d[0], d[1]});
    mine += __float_as_uint(q.x) != __float_as_uint(n[0] / d[0]);
    mine += __float_as_uint(q.y) != __float_as_uint(n[1] / d[1]);
  }
  if (mine) atomicAdd(bad, mine);
}

// every (numerator mantissa, denominator mantissa) pair at exponents 0 / 0: workgroup <- one denominator
__global__ void st_div_all_k(const uint32_t md_lo, unsigned long long* bad) {
  unsigned long long mine = 0;
  const float d = __uint_as_float(0x3f800000u | (md_lo + blockIdx.x));
  for (uint32_t mn = threadIdx.x * 2u; mn < (1u << 23); mn += blockDim.x * 2u) {
    const float n0 = __uint_as_float(0x3f800000u | mn), n1 = __uint_as_float(0x3f800000u | (mn + 1u));
    const rsx_f2 q = rsx_div2_fast((rsx_f2){n0, n1}, (rsx_f2){d, d});
    mine += __float_as_uint(q.x) != __float_as_uint(n0 / d);
    mine += __float_as_uint(q.y) != __float_as_uint(n1 / d);
  }
  if (mine) atomicAdd(bad, mine);
}
}  // namespace

extern "C" int rsx_adam_fast_math_selftest(unsigned long long* counts, uint32_t seed, int div_iters, int exhaustive_div,
                                           rsx_stream_t stream) {
  if (!counts || div_iters < 0) return RSX_EINVAL;
  hipStream_t st = rsx_s(stream);
  if (hipMemsetAsync(counts, 0, 4 * sizeof(unsigned long long), st) != hipSuccess) return RSX_ELAUNCH;
  RSX_LAUNCH(st_sqrt_k, dim3(4096), dim3(256), 0, st, (127u - 96u) << 23, (127u + 41u) << 23, counts);
  const int grid = 4096, block = 256;
  if (div_iters > 0) RSX_LAUNCH(st_div_k, dim3(grid), dim3(block), 0, st, seed, div_iters, counts + 1);
  const unsigned long long pairs = 2ull * grid * block * (unsigned long long)div_iters;
  if (hipMemcpyAsync(counts + 2, &pairs, sizeof(pairs), hipMemcpyHostToDevice, st) != hipSuccess) return RSX_ELAUNCH;
  if (exhaustive_div) {
    for (uint32_t lo = 0; lo < (1u << 23); lo += (1u << 18)) {      // 32 launches of 2^18 denominators (~1 s each)
      RSX_LAUNCH(st_div_all_k, dim3(1u << 18), dim3(256), 0, st, lo, counts + 3);
      if (hipStreamSynchronize(st) != hipSuccess) return RSX_ELAUNCH;
    }
  }
  if (hipStreamSynchronize(st) != hipSuccess) return RSX_ELAUNCH;      // (`pairs` is a stack variable)
  RSX_CHECK_LAUNCH();
  return RSX_OK;
}
