// The weight-gradient reduce of the one-launch batch-norm-free tower (mlp_fused.hip) as a device function: its own launch
// (mlp_reduce_k) or extra workgroups of another launch of the step (din.py: the pooling backward, rsx_din_pool_bwd_pair_ride --
// the dense gradients are first needed by the optimizer, so the 9 us launch leaves the step's chain).
#pragma once
#include "rsx_common.h"

constexpr int MLP_MAX_L = RSX_MLP_MAX_LAYERS;          // hidden layers
typedef rsx_mlp_reduce_job MlpRed;                     // (include/rsx.h: the job is a plain struct the host hands around)

// one thread per float4 `e` of the concatenated regions: the nwg partials in ascending workgroup order, INFL loads in flight
template <int INFL>
__device__ __forceinline__ void mlp_reduce_body(const MlpRed& r, const unsigned e) {
  int q = 0;
  unsigned base = 0;
#pragma unroll
  for (int k = 0; k < MLP_MAX_L; ++k) {
    if (k < r.L && e >= r.e4_end[k]) {
      q = k + 1;
      base = r.e4_end[k];
    }
  }
  if (e >= r.e4_last) return;
  const unsigned e4 = e - base;
  const int K = q == 0 ? r.K[0] : (q == 1 ? r.K[1] : r.K[2]), N = q == 0 ? r.N[0] : (q == 1 ? r.N[1] : r.N[2]);
  const int KR = (K + 1 + 15) & ~15, NP = (N + 15) & ~15;
  const size_t reg4 = q < r.L ? (size_t)KR * NP / 4 : (size_t)r.NPo / 4;
  const long long pq = q == r.L ? r.poff[RSX_MLP_MAX_LAYERS] : (q == 0 ? r.poff[0] : (q == 1 ? r.poff[1] : r.poff[2]));
  const float4* src = reinterpret_cast<const float4*>(r.part + pq) + e4;
  float4 s = F4Z;
  double sl = 0.0;
  for (int g = 0; g < r.nwg; g += INFL) {
    float4 t[INFL];
#pragma unroll
    for (int u = 0; u < INFL; ++u) t[u] = src[(size_t)(g + u < r.nwg ? g + u : r.nwg - 1) * reg4];
#pragma unroll
    for (int u = 0; u < INFL; ++u) {
      if (g + u < r.nwg) {
        s = f4_add(s, t[u]);
        // (the loss term is summed in fp64: 1 024 terms of ~0.7 in fp32 would cost the reported loss its last digits)
        if (q == r.L) {
          const int c0 = (int)e4 * 4, cl = r.NL + 1 - c0;
          if (cl >= 0 && cl < 4) sl += (double)(cl == 0 ? t[u].x : cl == 1 ? t[u].y : cl == 2 ? t[u].z : t[u].w);
        }
      }
    }
  }
  const float v[4] = {s.x, s.y, s.z, s.w};
  if (q < r.L) {
    const int kk = (int)(((size_t)e4 * 4) / NP), n = (int)(((size_t)e4 * 4) - (size_t)kk * NP);
    float* dW = q == 0 ? r.dW[0] : (q == 1 ? r.dW[1] : r.dW[2]);
    float* db = q == 0 ? r.db[0] : (q == 1 ? r.db[1] : r.db[2]);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      if (n + t < N) {
        if (kk < K) dW[(size_t)kk * N + n + t] = v[t];
        else if (kk == K) db[n + t] = v[t];
      }
    }
  } else {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int c = (int)e4 * 4 + t;
      if (c < r.NL) r.dwout[c] = v[t];
      else if (c == r.NL) r.dbout[0] = v[t];
      else if (c == r.NL + 1) r.loss[0] = (float)(sl * r.inv_B);
    }
  }
}
