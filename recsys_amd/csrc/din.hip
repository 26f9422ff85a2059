// DIN attention-weighted history pooling on gfx950 + the generic sparse-row machinery DIN needs.
// Reference call sites: din/din.py:103-125 `_attention` -- wgt_emb = sum_p dense_emb[b,p,:] * att_wgt[b,p] *
// (ids[b,p] > 0) (no softmax, no scaling; :118-124), called twice (:127-128); lookups din/din.py:96-105.
// the gather and the weighted masked sum are the hand-kernel part, the attention
// MLP GEMMs (M = B*P rows) are library GEMMs.  HBM-bound: per history P*(4 + K*4 + 4) B read, K*4 written.
#include "rsx_common.h"

// out[b,:] = sum_p H[b,p,:] * w[b,p] * (ids[b,p] > 0).   One wave per example; LPR = K/4 lanes per row.
template <int K>
__global__ __launch_bounds__(256) void din_pool_fwd_k(const float* __restrict__ H, const float* __restrict__ w,
                                                      const int32_t* __restrict__ ids, float* __restrict__ out, int B,
                                                      int P) {
  constexpr int LPR = K / 4, RPW = 64 / LPR;
  const int lane = threadIdx.x & 63;
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= B) return;
  const int q = lane % LPR, j = lane / LPR;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int p = j; p < P; p += RPW) {
    const size_t e = (size_t)b * P + p;
    if (ids[e] > 0) {
      const float4 h = reinterpret_cast<const float4*>(H)[e * LPR + q];
      acc = f4_add(acc, f4_scale(w[e], h));
    }
  }
#pragma unroll
  for (int m = LPR; m < 64; m <<= 1) acc = f4_add(acc, f4_shfl_xor(acc, m));
  if (j == 0) reinterpret_cast<float4*>(out)[(size_t)b * LPR + q] = acc;
}

// dH[b,p,:] (+)= dout[b,:] * w[b,p] * mask ;  dw[b,p] = <H[b,p,:], dout[b,:]> * mask.
template <int K>
__global__ __launch_bounds__(256) void din_pool_bwd_k(const float* __restrict__ H, const float* __restrict__ w,
                                                      const int32_t* __restrict__ ids, const float* __restrict__ dout,
                                                      float* __restrict__ dH, float* __restrict__ dw, int accumulate,
                                                      int B, int P) {
  constexpr int LPR = K / 4, RPW = 64 / LPR;
  const int lane = threadIdx.x & 63;
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= B) return;
  const int q = lane % LPR, j = lane / LPR;
  const float4 g = reinterpret_cast<const float4*>(dout)[(size_t)b * LPR + q];
  const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int p0 = 0; p0 < P; p0 += RPW) {   // wave-uniform trip count: the shuffles below need every lane
    const int p = p0 + j;
    const bool in = p < P;
    const size_t e = (size_t)b * P + (in ? p : 0);
    const bool on = in && ids[e] > 0;
    float d = 0.f;
    float4 o = z;
    if (on) {
      const float4 h = reinterpret_cast<const float4*>(H)[e * LPR + q];
      d = (h.x * g.x + h.y * g.y) + (h.z * g.z + h.w * g.w);
      o = f4_scale(w[e], g);
    }
#pragma unroll
    for (int m = 1; m < LPR; m <<= 1) d += __shfl_xor(d, m);
    if (in) {
      float4* dst = reinterpret_cast<float4*>(dH) + e * LPR + q;
      *dst = accumulate ? f4_add(*dst, o) : o;
      if (q == 0) dw[e] = d;
    }
  }
}

// -------------------------------------------------------------------------------------------------
// Sorted keys -> segments.  Input: keys sorted ascending (any stable sort; ties keep entry order).  One
// workgroup: head flags + block scan ->  uniq_row[j], seg_off[j] (seg_off[U] = N), nuniq[0] = U, and the
// row -> j slot map (clearing the previous call's entries first).  Same outputs as rsx_field_sort with F = 1.
// -------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void sorted_segments_k(const int32_t* __restrict__ keys, int N,
                                                          int32_t* __restrict__ uniq_row, int32_t* __restrict__ seg_off,
                                                          int32_t* __restrict__ nuniq, int32_t* __restrict__ slot) {
  __shared__ int wsum[16];
  __shared__ int carry;
  const int tid = threadIdx.x;
  const int prev = nuniq[0];
  for (int jj = tid; jj < prev; jj += 1024) slot[uniq_row[jj]] = -1;
  if (tid == 0) carry = 0;
  __syncthreads();
  const int per = (N + 1023) / 1024;
  const int i0 = tid * per, i1 = i0 + per < N ? i0 + per : N;
  int cnt = 0;
  for (int i = i0; i < i1; ++i)
    if (i == 0 || keys[i] != keys[i - 1]) ++cnt;
  int incl = cnt;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int o = __shfl_up(incl, d);
    if ((tid & 63) >= d) incl += o;
  }
  if ((tid & 63) == 63) wsum[tid >> 6] = incl;
  __syncthreads();
  int base = incl - cnt, total = 0;
  for (int w = 0; w < 16; ++w) {
    if (w < (tid >> 6)) base += wsum[w];
    total += wsum[w];
  }
  int jn = base;
  for (int i = i0; i < i1; ++i) {
    if (i == 0 || keys[i] != keys[i - 1]) {
      uniq_row[jn] = keys[i];
      seg_off[jn] = i;
      slot[keys[i]] = jn;
      ++jn;
    }
  }
  if (tid == 0) {
    seg_off[total] = N;
    nuniq[0] = total;
  }
}

template <int K>
static void launch_pool_fwd(hipStream_t st, const float* H, const float* w, const int32_t* ids, float* out, int B, int P) {
  din_pool_fwd_k<K><<<dim3((B + 3) / 4), dim3(256), 0, st>>>(H, w, ids, out, B, P);
}
template <int K>
static void launch_pool_bwd(hipStream_t st, const float* H, const float* w, const int32_t* ids, const float* dout,
                            float* dH, float* dw, int acc, int B, int P) {
  din_pool_bwd_k<K><<<dim3((B + 3) / 4), dim3(256), 0, st>>>(H, w, ids, dout, dH, dw, acc, B, P);
}

extern "C" int rsx_din_pool_fwd(const float* H, const float* w, const int32_t* ids, float* out, int B, int P, int K,
                                rsx_stream_t stream) {
  if (B < 0 || P <= 0) return RSX_EINVAL;
  if (B == 0) return RSX_OK;
  if (!H || !w || !ids || !out) return RSX_EINVAL;
  switch (K) {
    case 4: launch_pool_fwd<4>(rsx_s(stream), H, w, ids, out, B, P); break;
    case 8: launch_pool_fwd<8>(rsx_s(stream), H, w, ids, out, B, P); break;
    case 16: launch_pool_fwd<16>(rsx_s(stream), H, w, ids, out, B, P); break;
    case 32: launch_pool_fwd<32>(rsx_s(stream), H, w, ids, out, B, P); break;
    case 64: launch_pool_fwd<64>(rsx_s(stream), H, w, ids, out, B, P); break;
    default: return RSX_EUNSUPPORTED;
  }
  RSX_CHECK_LAUNCH();
  return RSX_OK;
}

extern "C" int rsx_din_pool_bwd(const float* H, const float* w, const int32_t* ids, const float* dout, float* dH,
                                float* dw, int accumulate, int B, int P, int K, rsx_stream_t stream) {
  if (B < 0 || P <= 0) return RSX_EINVAL;
  if (B == 0) return RSX_OK;
  if (!H || !w || !ids || !dout || !dH || !dw) return RSX_EINVAL;
  switch (K) {
    case 4: launch_pool_bwd<4>(rsx_s(stream), H, w, ids, dout, dH, dw, accumulate, B, P); break;
    case 8: launch_pool_bwd<8>(rsx_s(stream), H, w, ids, dout, dH, dw, accumulate, B, P); break;
    case 16: launch_pool_bwd<16>(rsx_s(stream), H, w, ids, dout, dH, dw, accumulate, B, P); break;
    case 32: launch_pool_bwd<32>(rsx_s(stream), H, w, ids, dout, dH, dw, accumulate, B, P); break;
    case 64: launch_pool_bwd<64>(rsx_s(stream), H, w, ids, dout, dH, dw, accumulate, B, P); break;
    default: return RSX_EUNSUPPORTED;
  }
  RSX_CHECK_LAUNCH();
  return RSX_OK;
}

extern "C" int rsx_sorted_segments(const int32_t* sorted_keys, int N, int32_t* uniq_row, int32_t* seg_off,
                                   int32_t* nuniq, int32_t* slot, rsx_stream_t stream) {
  if (N < 0) return RSX_EINVAL;
  if (!uniq_row || !seg_off || !nuniq || !slot || (N > 0 && !sorted_keys)) return RSX_EINVAL;
  hipLaunchKernelGGL(sorted_segments_k, dim3(1), dim3(1024), 0, rsx_s(stream), sorted_keys, N, uniq_row, seg_off, nuniq,
                     slot);
  RSX_CHECK_LAUNCH();
  return RSX_OK;
}
