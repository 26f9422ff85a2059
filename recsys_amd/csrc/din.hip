// DIN attention-weighted history pooling on gfx950 + the generic sparse-row machinery DIN needs.
// Reference call sites: din/din.py:103-125 `_attention` -- wgt_emb = sum_p dense_emb[b,p,:] * att_wgt[b,p] *
// (ids[b,p] > 0) (no softmax, no scaling; :118-124), called twice (:127-128); lookups din/din.py:96-105.
// the gather and the weighted masked sum are the hand-kernel part, the attention
// MLP GEMMs (M = B*P rows) are library GEMMs.  HBM-bound: per history P*(4 + K*4 + 4) B read, K*4 written.
#include "rsx_common.h"
#include "mlp_reduce_device.h"
#include "gather_rows_device.h"

// out[b,:] = sum_p H[b,p,:] * w[b,p] * (ids[b,p] > 0).   One wave per example; LPR = K/4 lanes per row.
template <int K>
__device__ __forceinline__ void pool_fwd_wave(const float* __restrict__ H, const float* __restrict__ w,
                                              const int32_t* __restrict__ ids, float* __restrict__ out, int B, int P,
                                              int ldo) {
  constexpr int LPR = K / 4, RPW = 64 / LPR;
  const int lane = threadIdx.x & 63;
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= B) return;
  const int q = lane % LPR, j = lane / LPR;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  // positions p = j, j + RPW, ... in batches of 4: ids, weights and rows of the batch loaded unconditionally on clamped
  // positions, padding (id 0) and the tail masked by a zero weight -- a load behind `if (ids > 0)` is a branch of its own
  // and the positions would then load one after the other.  Adding w*h with w = 0 is exact, the order stays ascending p.
  for (int p0 = j; p0 < P; p0 += 4 * RPW) {
    float wv[4];
    float4 hv[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int pp = p0 + k * RPW;
      const size_t e = (size_t)b * P + (pp < P ? pp : P - 1);
      const float keep = (pp < P && ids[e] > 0) ? 1.f : 0.f;
      wv[k] = w[e] * keep;
      hv[k] = reinterpret_cast<const float4*>(H)[e * LPR + q];
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) acc = f4_add(acc, f4_scale(wv[k], hv[k]));
  }
#pragma unroll
  for (int m = LPR; m < 64; m <<= 1) acc = f4_add(acc, f4_shfl_xor(acc, m));
  if (j == 0) *reinterpret_cast<float4*>(out + (size_t)b * ldo + 4 * q) = acc;   // (ldo: row stride of `out`, floats)
}
template <int K>
__global__ __launch_bounds__(256) void din_pool_fwd_k(const float* __restrict__ H, const float* __restrict__ w,
                                                      const int32_t* __restrict__ ids, float* __restrict__ out, int B,
                                                      int P, int ldo) {
  pool_fwd_wave<K>(H, w, ids, out, B, P, ldo);
}
// both histories of din.py in one launch (grid.y = history): the two launches were 6.7 us each for 13 MB of traffic
struct PoolFwdSet { const float* H; const float* w; const int32_t* ids; float* out; };
template <int K>
__global__ __launch_bounds__(256) void din_pool_fwd_pair_k(const PoolFwdSet s0, const PoolFwdSet s1, int B, int P, int ldo) {
  if (blockIdx.y == 0) pool_fwd_wave<K>(s0.H, s0.w, s0.ids, s0.out, B, P, ldo);
  else pool_fwd_wave<K>(s1.H, s1.w, s1.ids, s1.out, B, P, ldo);
}

// dH[b,p,:] (+)= dout[b,:] * w[b,p] * mask ;  dw[b,p] = <H[b,p,:], dout[b,:]> * mask.
template <int K>
__device__ __forceinline__ void pool_bwd_wave(const float* __restrict__ H, const float* __restrict__ w,
                                              const int32_t* __restrict__ ids, const float* __restrict__ dout,
                                              float* __restrict__ dH, float* __restrict__ dw, int accumulate, int B, int P,
                                              int ldg, int ldh) {
  constexpr int LPR = K / 4, RPW = 64 / LPR;
  const int lane = threadIdx.x & 63;
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= B) return;
  const int q = lane % LPR, j = lane / LPR;
  const float4 g = *reinterpret_cast<const float4*>(dout + (size_t)b * ldg + 4 * q);     // (ldg / ldh: row strides, floats)
  // four positions per lane group and trip, their loads all issued before the first use (one position per trip was a chain
  // of 13 dependent round trips per example at P = 100: 14.5 us per launch for 26 MB of traffic)
  for (int p0 = 0; p0 < P; p0 += 4 * RPW) {   // wave-uniform trip count: the shuffles below need every lane
    bool in[4];
    size_t e[4];
    float keep[4], wv[4];
    float4 h[4], prev[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int p = p0 + k * RPW + j;
      in[k] = p < P;
      e[k] = (size_t)b * P + (in[k] ? p : 0);
      // unconditional loads, padding masked by multiplication (a load behind `if (ids > 0)` is a branch of its own)
      keep[k] = (in[k] && ids[e[k]] > 0) ? 1.f : 0.f;
      h[k] = reinterpret_cast<const float4*>(H)[e[k] * LPR + q];
      wv[k] = w[e[k]];
      prev[k] = accumulate ? *reinterpret_cast<const float4*>(dH + e[k] * ldh + 4 * q) : F4Z;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float d = ((h[k].x * g.x + h[k].y * g.y) + (h[k].z * g.z + h[k].w * g.w)) * keep[k];
      const float4 o = f4_scale(wv[k] * keep[k], g);
#pragma unroll
      for (int m = 1; m < LPR; m <<= 1) d += __shfl_xor(d, m);
      if (in[k]) {
        *reinterpret_cast<float4*>(dH + e[k] * ldh + 4 * q) = accumulate ? f4_add(prev[k], o) : o;
        if (q == 0) dw[e[k]] = d;
      }
    }
  }
}
template <int K>
__global__ __launch_bounds__(256) void din_pool_bwd_k(const float* __restrict__ H, const float* __restrict__ w,
                                                      const int32_t* __restrict__ ids, const float* __restrict__ dout,
                                                      float* __restrict__ dH, float* __restrict__ dw, int accumulate,
                                                      int B, int P, int ldg, int ldh) {
  pool_bwd_wave<K>(H, w, ids, dout, dH, dw, accumulate, B, P, ldg, ldh);
}
struct PoolBwdSet { const float* H; const float* w; const int32_t* ids; const float* dout; float* dH; float* dw; };
template <int K>
__global__ __launch_bounds__(256) void din_pool_bwd_pair_k(const PoolBwdSet s0, const PoolBwdSet s1, int accumulate, int B,
                                                           int P, int ldg, int ldh) {
  if (blockIdx.y == 0) pool_bwd_wave<K>(s0.H, s0.w, s0.ids, s0.dout, s0.dH, s0.dw, accumulate, B, P, ldg, ldh);
  else pool_bwd_wave<K>(s1.H, s1.w, s1.ids, s1.dout, s1.dH, s1.dw, accumulate, B, P, ldg, ldh);
}

// -------------------------------------------------------------------------------------------------
// Sorted keys -> segments.  Input: keys sorted ascending (any stable sort; ties keep entry order).  Outputs: uniq_row[j],
// seg_off[j] (seg_off[U] = N), nuniq[0] = U and the row -> j slot map (the previous call's entries cleared first) -- the
// rsx_field_sort contract with F = 1 -- plus, when segid != NULL, the two-stage segment-sum workspace (segment index of
// every position, long / huge segment lists).  Three small launches over 1024-key blocks: head counts (+ slot clearing),
// emit (each block adds the counts of the blocks before it: <= N/1024 values), long lists.
// -------------------------------------------------------------------------------------------------
constexpr int SS_T = 256, SS_IPT = 4, SS_BLK = SS_T * SS_IPT;

__global__ __launch_bounds__(SS_T) void sorted_heads_k(const int32_t* __restrict__ keys, int N,
                                                       const int32_t* __restrict__ uniq_row,
                                                       const int32_t* __restrict__ nuniq, int32_t* __restrict__ slot,
                                                       int32_t* __restrict__ blk_cnt, int32_t* __restrict__ long_cnt) {
  __shared__ int wsum[SS_T / 64];
  const int tid = threadIdx.x;
  const int prev = nuniq[0];
  for (int jj = blockIdx.x * SS_T + tid; jj < prev; jj += gridDim.x * SS_T) slot[uniq_row[jj]] = -1;
  if (long_cnt != nullptr && blockIdx.x == 0 && tid < 2) long_cnt[tid] = 0;
  const int i0 = blockIdx.x * SS_BLK + tid * SS_IPT;
  int cnt = 0;
#pragma unroll
  for (int k = 0; k < SS_IPT; ++k) {
    const int i = i0 + k;
    if (i < N && (i == 0 || keys[i] != keys[i - 1])) ++cnt;
  }
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) cnt += __shfl_xor(cnt, d);
  if ((tid & 63) == 0) wsum[tid >> 6] = cnt;
  __syncthreads();
  if (tid == 0) blk_cnt[blockIdx.x] = (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]);
}

__global__ __launch_bounds__(SS_T) void sorted_emit_k(const int32_t* __restrict__ keys, int N,
                                                      int32_t* __restrict__ uniq_row, int32_t* __restrict__ seg_off,
                                                      int32_t* __restrict__ nuniq, int32_t* __restrict__ slot,
                                                      int32_t* __restrict__ segid, const int32_t* __restrict__ blk_cnt) {
  __shared__ int wsum[SS_T / 64];
  __shared__ int red[SS_T / 64];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  // heads in the blocks before this one
  int before = 0;
  for (int b = tid; b < (int)blockIdx.x; b += SS_T) before += blk_cnt[b];
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) before += __shfl_xor(before, d);
  if (lane == 0) red[w] = before;
  const int i0 = blockIdx.x * SS_BLK + tid * SS_IPT;
  int key[SS_IPT], pk = 0;
  bool head[SS_IPT];
  int cnt = 0;
  if (i0 > 0 && i0 < N) pk = keys[i0 - 1];
#pragma unroll
  for (int k = 0; k < SS_IPT; ++k) {
    const int i = i0 + k;
    key[k] = i < N ? keys[i] : 0;
    head[k] = i < N && (i == 0 || key[k] != (k == 0 ? pk : key[k - 1]));
    cnt += head[k];
  }
  int incl = cnt;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int o = __shfl_up(incl, d);
    if (lane >= d) incl += o;
  }
  if (lane == 63) wsum[w] = incl;
  __syncthreads();
  int jn = (red[0] + red[1]) + (red[2] + red[3]) + incl - cnt;
  for (int ww = 0; ww < w; ++ww) jn += wsum[ww];
#pragma unroll
  for (int k = 0; k < SS_IPT; ++k) {
    const int i = i0 + k;
    if (i >= N) break;
    if (head[k]) {
      uniq_row[jn] = key[k];
      seg_off[jn] = i;
      slot[key[k]] = jn;
      ++jn;
    }
    if (segid != nullptr) segid[i] = jn - 1;
  }
  if (i0 + SS_IPT >= N && i0 < N) {   // the thread holding the last key: jn is now the total
    seg_off[jn] = N;
    nuniq[0] = jn;
  }
  if (N == 0 && blockIdx.x == 0 && tid == 0) {
    seg_off[0] = 0;
    nuniq[0] = 0;
  }
}

// long (> 16 entries) segments from the front of the list, huge (> 256) from the back; one returning atomic per wave
__global__ __launch_bounds__(256) void sorted_long_lists_k(const int32_t* __restrict__ seg_off,
                                                           const int32_t* __restrict__ nuniq, int32_t* __restrict__ cnt,
                                                           int32_t* __restrict__ ll, int nch) {
  const int j = blockIdx.x * 256 + threadIdx.x, lane = threadIdx.x & 63;
  const int U = nuniq[0];
  const int L = j < U ? seg_off[j + 1] - seg_off[j] : 0;
  const bool lg = L > 16 && L <= 256, hg = L > 256;
  const uint64_t ml = __ballot(lg), mh = __ballot(hg);
  int bl = 0, bh = 0;
  if (lane == 0 && ml) bl = atomicAdd(cnt, __popcll(ml));
  if (lane == 0 && mh) bh = atomicAdd(cnt + 1, __popcll(mh));
  bl = __shfl(bl, 0);
  bh = __shfl(bh, 0);
  const uint64_t lt = (1ull << lane) - 1ull;
  if (lg) ll[bl + __popcll(ml & lt)] = j;
  if (hg) ll[nch - 1 - (bh + __popcll(mh & lt))] = j;
}

template <int K>
static void launch_pool_fwd(hipStream_t st, const float* H, const float* w, const int32_t* ids, float* out, int B, int P,
                            int ldo) {
  RSX_COUNT_LAUNCH(); din_pool_fwd_k<K><<<dim3((B + 3) / 4), dim3(256), 0, st>>>(H, w, ids, out, B, P, ldo);
}
template <int K>
static void launch_pool_bwd(hipStream_t st, const float* H, const float* w, const int32_t* ids, const float* dout,
                            float* dH, float* dw, int acc, int B, int P, int ldg, int ldh) {
  RSX_COUNT_LAUNCH(); din_pool_bwd_k<K><<<dim3((B + 3) / 4), dim3(256), 0, st>>>(H, w, ids, dout, dH, dw, acc, B, P, ldg, ldh);
}

extern "C" int rsx_din_pool_fwd_ld(const float* H, const float* w, const int32_t* ids, float* out, int B, int P, int K,
                                   int ld_out, rsx_stream_t stream) {
  if (B < 0 || P <= 0) return RSX_EINVAL;
  if (B == 0) return RSX_OK;
  if (!H || !w || !ids || !out || ld_out < K || (ld_out & 3)) return RSX_EINVAL;
  switch (K) {
    case 4: launch_pool_fwd<4>(rsx_s(stream), H, w, ids, out, B, P, ld_out); break;
    case 8: launch_pool_fwd<8>(rsx_s(stream), H, w, ids, out, B, P, ld_out); break;
    case 16: launch_pool_fwd<16>(rsx_s(stream), H, w, ids, out, B, P, ld_out); break;
    case 32: launch_pool_fwd<32>(rsx_s(stream), H, w, ids, out, B, P, ld_out); break;
    case 64: launch_pool_fwd<64>(rsx_s(stream), H, w, ids, out, B, P, ld_out); break;
    default: return RSX_EUNSUPPORTED;
  }
  RSX_CHECK_LAUNCH();
  return RSX_OK;
}
extern "C" int rsx_din_pool_fwd(const float* H, const float* w, const int32_t* ids, float* out, int B, int P, int K,
                                rsx_stream_t stream) {
  return rsx_din_pool_fwd_ld(H, w, ids, out, B, P, K, K, stream);
}

template <int K>
static void launch_pool_fwd_pair(hipStream_t st, const PoolFwdSet& a, const PoolFwdSet& b, int B, int P, int ldo) {
  RSX_COUNT_LAUNCH(); din_pool_fwd_pair_k<K><<<dim3((B + 3) / 4, 2), dim3(256), 0, st>>>(a, b, B, P, ldo);
}
template <int K>
static void launch_pool_bwd_pair(hipStream_t st, const PoolBwdSet& a, const PoolBwdSet& b, int acc, int B, int P, int ldg,
                                 int ldh) {
  RSX_COUNT_LAUNCH(); din_pool_bwd_pair_k<K><<<dim3((B + 3) / 4, 2), dim3(256), 0, st>>>(a, b, acc, B, P, ldg, ldh);
}
// the same launch with a rider (round 4): the weight-gradient reduce of din.py's one-launch 'mlp_layer' (mlp_reduce_device.h) as
// extra workgroups behind history 0's -- first needed by the optimizer, so its 9 us launch leaves the step's chain
template <int K>
__global__ __launch_bounds__(256) void din_pool_bwd_pair_ride_k(const PoolBwdSet s0, const PoolBwdSet s1, int accumulate, int B,
                                                                int P, int ldg, int ldh, const MlpRed red, int n_own) {
  if ((int)blockIdx.x >= n_own) {
    if (blockIdx.y == 0) mlp_reduce_body<16>(red, (blockIdx.x - n_own) * 256 + threadIdx.x);
    return;
  }
  if (blockIdx.y == 0) pool_bwd_wave<K>(s0.H, s0.w, s0.ids, s0.dout, s0.dH, s0.dw, accumulate, B, P, ldg, ldh);
  else pool_bwd_wave<K>(s1.H, s1.w, s1.ids, s1.dout, s1.dH, s1.dw, accumulate, B, P, ldg, ldh);
}
template <int K>
static void launch_pool_bwd_pair_ride(hipStream_t st, const PoolBwdSet& a, const PoolBwdSet& b, int acc, int B, int P, int ldg,
                                      int ldh, const MlpRed& red) {
  const int n_own = (B + 3) / 4;
  RSX_COUNT_LAUNCH();
  din_pool_bwd_pair_ride_k<K><<<dim3(n_own + (red.e4_last + 255) / 256, 2), dim3(256), 0, st>>>(a, b, acc, B, P, ldg, ldh, red, n_own);
}
extern "C" int rsx_din_pool_fwd_pair(const float* H0, const float* w0, const int32_t* ids0, float* out0, const float* H1,
                                     const float* w1, const int32_t* ids1, float* out1, int B, int P, int K, int ld_out,
                                     rsx_stream_t stream) {
  if (B < 0 || P <= 0) return RSX_EINVAL;
  if (B == 0) return RSX_OK;
  if (!H0 || !w0 || !ids0 || !out0 || !H1 || !w1 || !ids1 || !out1 || ld_out < K || (ld_out & 3)) return RSX_EINVAL;
  const PoolFwdSet a{H0, w0, ids0, out0}, b{H1, w1, ids1, out1};
  switch (K) {
    case 4: launch_pool_fwd_pair<4>(rsx_s(stream), a, b, B, P, ld_out); break;
    case 8: launch_pool_fwd_pair<8>(rsx_s(stream), a, b, B, P, ld_out); break;
    case 16: launch_pool_fwd_pair<16>(rsx_s(stream), a, b, B, P, ld_out); break;
    case 32: launch_pool_fwd_pair<32>(rsx_s(stream), a, b, B, P, ld_out); break;
    case 64: launch_pool_fwd_pair<64>(rsx_s(stream), a, b, B, P, ld_out); break;
    default: return RSX_EUNSUPPORTED;
  }
  RSX_CHECK_LAUNCH();
  return RSX_OK;
}
extern "C" int rsx_din_pool_bwd_pair(const float* H0, const float* w0, const int32_t* ids0, const float* dout0, float* dH0,
                                     float* dw0, const float* H1, const float* w1, const int32_t* ids1, const float* dout1,
                                     float* dH1, float* dw1, int accumulate, int B, int P, int K, int ld_dout, int ld_dH,
                                     rsx_stream_t stream) {
  if (B < 0 || P <= 0) return RSX_EINVAL;
  if (B == 0) return RSX_OK;
  if (!H0 || !w0 || !ids0 || !dout0 || !dH0 || !dw0 || !H1 || !w1 || !ids1 || !dout1 || !dH1 || !dw1 || ld_dout < K ||
      (ld_dout & 3) || ld_dH < K || (ld_dH & 3))
    return RSX_EINVAL;
  const PoolBwdSet a{H0, w0, ids0, dout0, dH0, dw0}, b{H1, w1, ids1, dout1, dH1, dw1};
  switch (K) {
    case 4: launch_pool_bwd_pair<4>(rsx_s(stream), a, b, accumulate, B, P, ld_dout, ld_dH); break;
    case 8: launch_pool_bwd_pair<8>(rsx_s(stream), a, b, accumulate, B, P, ld_dout, ld_dH); break;
    case 16: launch_pool_bwd_pair<16>(rsx_s(stream), a, b, accumulate, B, P, ld_dout, ld_dH); break;
    case 32: launch_pool_bwd_pair<32>(rsx_s(stream), a, b, accumulate, B, P, ld_dout, ld_dH); break;
    case 64: launch_pool_bwd_pair<64>(rsx_s(stream), a, b, accumulate, B, P, ld_dout, ld_dH); break;
    default: return RSX_EUNSUPPORTED;
  }
  RSX_CHECK_LAUNCH();
  return RSX_OK;
}

extern "C" int rsx_din_pool_bwd_pair_ride(const float* H0, const float* w0, const int32_t* ids0, const float* dout0, float* dH0,
                                          float* dw0, const float* H1, const float* w1, const int32_t* ids1, const float* dout1,
                                          float* dH1, float* dw1, int accumulate, int B, int P, int K, int ld_dout, int ld_dH,
                                          const rsx_mlp_reduce_job* rider, rsx_stream_t stream) {
  if (rider == nullptr || rider->e4_last == 0)
    return rsx_din_pool_bwd_pair(H0, w0, ids0, dout0, dH0, dw0, H1, w1, ids1, dout1, dH1, dw1, accumulate, B, P, K, ld_dout, ld_dH,
                                 stream);
  if (B <= 0 || P <= 0) return RSX_EINVAL;            // (a rider needs a launch to ride in)
  if (!H0 || !w0 || !ids0 || !dout0 || !dH0 || !dw0 || !H1 || !w1 || !ids1 || !dout1 || !dH1 || !dw1 || ld_dout < K ||
      (ld_dout & 3) || ld_dH < K || (ld_dH & 3) || !rider->part || rider->nwg <= 0)
    return RSX_EINVAL;
  const PoolBwdSet a{H0, w0, ids0, dout0, dH0, dw0}, b{H1, w1, ids1, dout1, dH1, dw1};
  switch (K) {
    case 4: launch_pool_bwd_pair_ride<4>(rsx_s(stream), a, b, accumulate, B, P, ld_dout, ld_dH, *rider); break;
    case 8: launch_pool_bwd_pair_ride<8>(rsx_s(stream), a, b, accumulate, B, P, ld_dout, ld_dH, *rider); break;
    case 16: launch_pool_bwd_pair_ride<16>(rsx_s(stream), a, b, accumulate, B, P, ld_dout, ld_dH, *rider); break;
    case 32: launch_pool_bwd_pair_ride<32>(rsx_s(stream), a, b, accumulate, B, P, ld_dout, ld_dH, *rider); break;
    case 64: launch_pool_bwd_pair_ride<64>(rsx_s(stream), a, b, accumulate, B, P, ld_dout, ld_dH, *rider); break;
    default: return RSX_EUNSUPPORTED;
  }
  RSX_CHECK_LAUNCH();
  return RSX_OK;
}

extern "C" int rsx_din_pool_bwd_ld(const float* H, const float* w, const int32_t* ids, const float* dout, float* dH,
                                   float* dw, int accumulate, int B, int P, int K, int ld_dout, int ld_dH,
                                   rsx_stream_t stream) {
  if (B < 0 || P <= 0) return RSX_EINVAL;
  if (B == 0) return RSX_OK;
  if (!H || !w || !ids || !dout || !dH || !dw || ld_dout < K || (ld_dout & 3) || ld_dH < K || (ld_dH & 3)) return RSX_EINVAL;
#define RSX_POOL_BWD(KK) launch_pool_bwd<KK>(rsx_s(stream), H, w, ids, dout, dH, dw, accumulate, B, P, ld_dout, ld_dH)
  switch (K) {
    case 4: RSX_POOL_BWD(4); break;
    case 8: RSX_POOL_BWD(8); break;
    case 16: RSX_POOL_BWD(16); break;
    case 32: RSX_POOL_BWD(32); break;
    case 64: RSX_POOL_BWD(64); break;
    default: return RSX_EUNSUPPORTED;
  }
#undef RSX_POOL_BWD
  RSX_CHECK_LAUNCH();
  return RSX_OK;
}
extern "C" int rsx_din_pool_bwd(const float* H, const float* w, const int32_t* ids, const float* dout, float* dH,
                                float* dw, int accumulate, int B, int P, int K, rsx_stream_t stream) {
  return rsx_din_pool_bwd_ld(H, w, ids, dout, dH, dw, accumulate, B, P, K, K, K, stream);
}

// ---- several row gathers in ONE launch (din/din.py:96-105: i_item / i_id / i_cate looked up by the target ids and by the
// two id histories: five tf.gather / embedding_lookup calls per step) ----------------------------------------------------------
__global__ __launch_bounds__(256) void gather_rows_multi_k(const GatherJobs g) { RSX_GATHER_ROWS_BLOCK(g, blockIdx.x); }

extern "C" int rsx_gather_rows_multi(const rsx_gather_job* jobs_h, int njobs, rsx_stream_t stream) {
  if (!jobs_h || njobs <= 0 || njobs > RSX_GATHER_MAX_JOBS) return RSX_EINVAL;
  GatherJobs g;
  uint32_t end = 0;
  const int rc = gather_jobs_pack(jobs_h, 0, njobs, g, &end);
  if (rc != RSX_OK) return rc;
  if (end == 0) return RSX_OK;
  RSX_LAUNCH(gather_rows_multi_k, dim3(end), dim3(256), 0, rsx_s(stream), g);
  RSX_CHECK_LAUNCH();
  return RSX_OK;
}

// ---- the sort keys of DIN's two id tables in one launch: entry e < B = the target lookup of example e, entry B + b*P + p =
// history position (b, p); a padding position (id <= 0, din/din.py:107) gets the field's DUMMY row (the table's last row:
// never looked up, zero gradient), so that a real row 0 stays an ordinary row.  keys2 [B + B*P, 2] (item, category). ---------
__global__ __launch_bounds__(256) void din_keys_k(const int32_t* __restrict__ i_id, const int32_t* __restrict__ i_cate,
                                                  const int32_t* __restrict__ hist_i, const int32_t* __restrict__ hist_c, int B,
                                                  long long BP, int dummy_i, int dummy_c, int32_t* __restrict__ keys2) {
  const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
  if (e >= B + BP) return;
  int a, c;
  if (e < B) {
    a = i_id[e];
    c = i_cate[e];
  } else {
    a = hist_i[e - B];
    c = hist_c[e - B];
    a = a > 0 ? a : dummy_i;
    c = c > 0 ? c : dummy_c;
  }
  reinterpret_cast<int2*>(keys2)[e] = make_int2(a, c);
}

extern "C" int rsx_din_keys(const int32_t* i_id, const int32_t* i_cate, const int32_t* hist_i, const int32_t* hist_c, int B,
                            int P, int dummy_item_row, int dummy_cate_row, int32_t* keys2, rsx_stream_t stream) {
  if (B < 0 || P < 0) return RSX_EINVAL;
  if (B == 0) return RSX_OK;
  if (!i_id || !i_cate || !hist_i || !hist_c || !keys2) return RSX_EINVAL;
  const long long n = (long long)B * (P + 1);
  RSX_LAUNCH(din_keys_k, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, rsx_s(stream), i_id, i_cate, hist_i, hist_c, B,
                     (long long)B * P, dummy_item_row, dummy_cate_row, keys2);
  RSX_CHECK_LAUNCH();
  return RSX_OK;
}

extern "C" int rsx_sorted_segments(const int32_t* sorted_keys, int N, int32_t* uniq_row, int32_t* seg_off,
                                   int32_t* nuniq, int32_t* slot, int32_t* segid, int stride, int32_t* scratch,
                                   rsx_stream_t stream) {
  if (N < 0) return RSX_EINVAL;
  if (!uniq_row || !seg_off || !nuniq || !slot || !scratch || (N > 0 && !sorted_keys)) return RSX_EINVAL;
  if (segid != nullptr && stride < N) return RSX_EINVAL;
  const int nblk = N > 0 ? (N + SS_BLK - 1) / SS_BLK : 1;
  int32_t* cnt = segid != nullptr ? segid + stride : nullptr;       // [2] counts, then the list [ceil(N/16)]
  RSX_LAUNCH(sorted_heads_k, dim3(nblk), dim3(SS_T), 0, rsx_s(stream), sorted_keys, N, uniq_row, nuniq, slot,
                     scratch, cnt);
  RSX_CHECK_LAUNCH();
  RSX_LAUNCH(sorted_emit_k, dim3(nblk), dim3(SS_T), 0, rsx_s(stream), sorted_keys, N, uniq_row, seg_off, nuniq,
                     slot, segid, scratch);
  RSX_CHECK_LAUNCH();
  if (segid != nullptr && N > 0) {
    RSX_LAUNCH(sorted_long_lists_k, dim3((N + 255) / 256), dim3(256), 0, rsx_s(stream), seg_off, nuniq, cnt,
                       cnt + 2, (N + 15) / 16);
    RSX_CHECK_LAUNCH();
  }
  return RSX_OK;
}
