// Work of other launches of the TRAIN step that only the optimizer needs -- the tower's dW partial-tile reductions
// (tower_reduce_dw_jobs_k) and the cross layers' gradient reduce (cross_reduce_k) -- as bodies that either run in their own
// launch or ride as extra 256-thread workgroups of the scatter's stage-A launch (embedding.hip segsum_tiles_ride_k, round 4).
#pragma once
#include "rsx_common.h"

struct DwReduceJobs {
  rsx_dw_reduce_job j[RSX_DW_REDUCE_MAX_JOBS];
  uint32_t blk_end[RSX_DW_REDUCE_MAX_JOBS];
};

// job of workgroup BLK of the list G -> JB (a copy) and its workgroup index inside the job -> LOCAL.  A macro: the selection must
// index the kernel's own by-value parameter with compile-time indices (gather_rows_device.h has the story)
#define RSX_DW_REDUCE_SELECT(G, BLK, JB, LOCAL)                                  \
  rsx_dw_reduce_job JB = (G).j[0];                                               \
  uint32_t LOCAL = (BLK);                                                        \
  _Pragma("unroll") for (int k_ = 1; k_ < RSX_DW_REDUCE_MAX_JOBS; ++k_) {        \
    if ((BLK) >= (G).blk_end[k_ - 1]) {                                          \
      JB = (G).j[k_];                                                            \
      LOCAL = (BLK) - (G).blk_end[k_ - 1];                                       \
    }                                                                            \
  }

// one 256-thread workgroup of ONE job (a plain struct of scalars: safe to pass by value)
__device__ __forceinline__ void dw_reduce_job_block(const rsx_dw_reduce_job jb, const uint32_t blk) {
  const int tid = threadIdx.x;
  if (jb.layout == 0) {            // tile-major partials [tiles][sb][256] (tower_bwd_k<true>): one workgroup per tile
    const int tile = (int)blk;
    const float* all = jb.partials + (size_t)tile * jb.sb * 256;
    float s = 0.f;
    for (int q = 0; q < jb.sb; q += 8) {
      float t[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) t[u] = q + u < jb.sb ? all[(size_t)(q + u) * 256 + tid] : 0.f;
#pragma unroll
      for (int u = 0; u < 8; ++u) s += t[u];
    }
    const int ct_n = (jb.N + 15) / 16;
    const int nt = tile % ct_n, kf = tile / ct_n;
    const int orow = kf * 16 + (tid >> 4), ocol = nt * 16 + (tid & 15);
    if (ocol < jb.N) {
      if (orow < jb.K) jb.dW[(size_t)orow * jb.N + ocol] = s;
      else if (orow == jb.K) jb.db[ocol] = s;
    }
    return;
  }
  // row-block partials [sb][KR][NP] (tower_bwd_big_k)
  const int KR = (jb.K + 1 + 15) / 16 * 16, NP = (jb.N + 15) / 16 * 16;
  const size_t e4 = (size_t)blk * 256 + tid, tot4 = (size_t)KR * NP / 4;
  if (e4 >= tot4) return;
  const float4* src = reinterpret_cast<const float4*>(jb.partials);
  float4 s = F4Z;
  for (int q = 0; q < jb.sb; q += 8) {
    float4 t[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) t[u] = src[(size_t)(q + u < jb.sb ? q + u : jb.sb - 1) * tot4 + e4];
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (q + u < jb.sb) s = f4_add(s, t[u]);
  }
  const int kk = (int)((e4 * 4) / NP), n = (int)((e4 * 4) - (size_t)kk * NP);
  const float v[4] = {s.x, s.y, s.z, s.w};
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    if (n + t < jb.N) {
      if (kk < jb.K) jb.dW[(size_t)kk * jb.N + n + t] = v[t];
      else if (kk == jb.K) jb.db[n + t] = v[t];
    }
  }
}

// host: packs njobs jobs; *blocks = workgroups they need.  rsx_status.
static inline int dw_reduce_pack(const rsx_dw_reduce_job* jobs_h, int njobs, DwReduceJobs& g, uint32_t* blocks) {
  if (njobs < 0 || njobs > RSX_DW_REDUCE_MAX_JOBS || (njobs > 0 && !jobs_h)) return RSX_EINVAL;
  uint32_t end = 0;
  for (int k = 0; k < RSX_DW_REDUCE_MAX_JOBS; ++k) {
    if (k < njobs) {
      const rsx_dw_reduce_job& j = jobs_h[k];
      if (!j.partials || !j.dW || !j.db || j.sb <= 0 || j.K <= 0 || j.N <= 0 || (j.layout != 0 && j.layout != 1)) return RSX_EINVAL;
      if (j.layout == 0) end += (uint32_t)(((j.K + 1 + 15) / 16) * ((j.N + 15) / 16));
      else end += (uint32_t)((((size_t)(j.K + 1 + 15) / 16 * 16) * ((size_t)(j.N + 15) / 16 * 16) / 4 + 255) / 256);
      g.j[k] = j;
    } else {
      g.j[k] = njobs > 0 ? jobs_h[njobs - 1] : rsx_dw_reduce_job{};
    }
    g.blk_end[k] = end;
  }
  *blocks = end;
  return RSX_OK;
}

// ---- the cross layers' gradient reduce (cross.hip cross_reduce_k): out[j] = sum over the RT per-wave partials.  16 threads per
// output: thread q sums partials q, q + 16, ... in order, then the 16 totals are added in ascending q through LDS (fixed order).
// Workgroup blk of ceil(n / 16); red: 256 floats of LDS.
__device__ __forceinline__ void cross_reduce_block(const float* __restrict__ part, const int RT, const int n, float* __restrict__ dW,
                                                   float* __restrict__ dB, float* __restrict__ dwout, const int L, const int dim,
                                                   const uint32_t blk, float* red /* [16][16] */) {
  const int jl = threadIdx.x & 15, q = threadIdx.x >> 4;
  const int j = (int)blk * 16 + jl;
  float s = 0.f;
  if (j < n) {   // row group q takes rows q, q+16, ...: 8 loads in flight, fixed order
    int r = q;
    for (; r + 7 * 16 < RT; r += 8 * 16) {
      float t[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) t[u] = part[(size_t)(r + 16 * u) * n + j];
#pragma unroll
      for (int u = 0; u < 8; ++u) s += t[u];
    }
    for (; r < RT; r += 16) s += part[(size_t)r * n + j];
  }
  red[q * 16 + jl] = s;
  __syncthreads();
  if (q == 0 && j < n) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) t += red[k * 16 + jl];
    const int vec = j / dim, e = j - vec * dim;
    if (vec < L) dW[(size_t)vec * dim + e] = t;
    else if (vec < 2 * L) dB[(size_t)(vec - L) * dim + e] = t;
    else if (dwout != nullptr) dwout[e] = t;
  }
}

// ---- out[j] = sum of G partial vectors (rsx_vec_reduce_job; din_attn.hip attn_finish_block's weight-gradient reduce with the SAME
// association: sub-sum w = partials [w * per, (w + 1) * per), per = ceil(G / 16), 8 loads in flight; then the 16 sub-sums in
// ascending order).  A 256-thread workgroup: 4 waves x 4 sub-sums each, lane = column blk * 64 + lane.  sub: 1024 floats of LDS.
__device__ __forceinline__ void vec_reduce_block(const float* __restrict__ part, const int G, const int n, float* __restrict__ out,
                                                 const uint32_t blk, float* sub /* [16][64] */) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int j = (int)blk * 64 + lane;
  const int per = (G + 15) / 16;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int w = 4 * wv + k;
    const int g0 = w * per, g1 = g0 + per < G ? g0 + per : G;
    float s = 0.f;
    if (j < n) {
      int g = g0;
      for (; g + 8 <= g1; g += 8) {
        float t[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) t[u] = part[(size_t)(g + u) * n + j];
#pragma unroll
        for (int u = 0; u < 8; ++u) s += t[u];
      }
      for (; g < g1; ++g) s += part[(size_t)g * n + j];
    }
    sub[w * 64 + lane] = s;
  }
  __syncthreads();
  if (wv == 0 && j < n) {
    float t = sub[lane];
#pragma unroll
    for (int k = 1; k < 16; ++k) t += sub[k * 64 + lane];
    out[j] = t;
  }
}
