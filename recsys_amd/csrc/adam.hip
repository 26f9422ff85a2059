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
#include "rsx_common.h"

struct SegDev {
  int32_t kind, d;
  long long n;
  float *var, *m, *v, *g;
  const int32_t *slot, *uniq_row, *nuniq;
  int32_t B, stride, zero_grad;
  uint32_t blk_begin;
};
struct AdamArgs {
  SegDev seg[RSX_ADAM_MAX_SEGS];
  int32_t nseg;
  float lr, b1, b2, eps;
  float* state;
  uint32_t total_blocks;
};

struct Hp {
  float alpha, b1, b2, omb1, omb2, eps;
};

__device__ __forceinline__ void adam_sparse1(float& var, float& m, float& v, float g, bool has, const Hp& h) {
  float m1 = m * h.b1;
  float v1 = v * h.b2;
  if (has) {
    m1 = m1 + g * h.omb1;
    v1 = v1 + (g * g) * h.omb2;
  }
  var = var - (h.alpha * m1) / (sqrtf(v1) + h.eps);
  m = m1;
  v = v1;
}
__device__ __forceinline__ void adam_dense1(float& var, float& m, float& v, float g, const Hp& h) {
  const float m1 = m + (g - m) * h.omb1;
  const float v1 = v + (g * g - v) * h.omb2;
  var = var - (m1 * h.alpha) / (sqrtf(v1) + h.eps);
  m = m1;
  v = v1;
}

#define F4_APPLY(FN, VAR, M, V, G, ...)   \
  FN(VAR.x, M.x, V.x, G.x, __VA_ARGS__);  \
  FN(VAR.y, M.y, V.y, G.y, __VA_ARGS__);  \
  FN(VAR.z, M.z, V.z, G.z, __VA_ARGS__);  \
  FN(VAR.w, M.w, V.w, G.w, __VA_ARGS__)

constexpr int ADAM_T = 256;    // threads per workgroup
constexpr int ADAM_U = 4;      // float4 per lane
constexpr long long ADAM_Q = (long long)ADAM_T * ADAM_U;  // float4 per workgroup

__global__ __launch_bounds__(ADAM_T) void adam_multi_k(const AdamArgs a) {
  const float b1p = a.state[0], b2p = a.state[1];
  Hp h;
  h.b1 = a.b1;
  h.b2 = a.b2;
  h.omb1 = 1.0f - a.b1;
  h.omb2 = 1.0f - a.b2;
  h.eps = a.eps;
  h.alpha = a.lr * sqrtf(1.0f - b2p) / (1.0f - b1p);
  int si = 0;
#pragma unroll 1
  for (int k = 1; k < a.nseg; ++k)
    if (blockIdx.x >= a.seg[k].blk_begin) si = k;
  const SegDev& s = a.seg[si];
  const long long base = (long long)(blockIdx.x - s.blk_begin) * ADAM_Q;
  float4* __restrict__ var4 = reinterpret_cast<float4*>(s.var);
  float4* __restrict__ m4 = reinterpret_cast<float4*>(s.m);
  float4* __restrict__ v4 = reinterpret_cast<float4*>(s.v);
  const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);

  if (s.kind == RSX_ADAM_TABLE_TF1) {
    const int lpr = s.d >> 2;
    const long long n4 = s.n * lpr;
    const float4* __restrict__ G4 = reinterpret_cast<const float4*>(s.g);
#pragma unroll
    for (int u = 0; u < ADAM_U; ++u) {
      const long long e = base + (long long)u * ADAM_T + threadIdx.x;
      if (e < n4) {
        const long long row = e / lpr;
        const int q = (int)(e - row * lpr);
        const int sl = s.slot[row];
        float4 var = var4[e], m = m4[e], v = v4[e];
        const bool has = sl >= 0;
        const float4 g = has ? G4[(long long)sl * lpr + q] : z4;
        F4_APPLY(adam_sparse1, var, m, v, g, has, h);
        var4[e] = var;
        m4[e] = m;
        v4[e] = v;
      }
    }
  } else if (s.kind == RSX_ADAM_DENSE) {
    float4* __restrict__ g4 = reinterpret_cast<float4*>(s.g);
    const long long n4 = s.n >> 2;
#pragma unroll
    for (int u = 0; u < ADAM_U; ++u) {
      const long long e = base + (long long)u * ADAM_T + threadIdx.x;
      if (e < n4) {
        float4 var = var4[e], m = m4[e], v = v4[e];
        const float4 g = g4[e];
        F4_APPLY(adam_dense1, var, m, v, g, h);
        var4[e] = var;
        m4[e] = m;
        v4[e] = v;
        if (s.zero_grad) g4[e] = z4;
      } else if (e == n4) {  // scalar tail (n not a multiple of 4)
        for (long long i = n4 * 4; i < s.n; ++i) {
          adam_dense1(s.var[i], s.m[i], s.v[i], s.g[i], h);
          if (s.zero_grad) s.g[i] = 0.f;
        }
      }
    }
  } else if (s.kind == RSX_ADAM_VEC_SLOT) {
    const long long n4 = s.n >> 2;
#pragma unroll
    for (int u = 0; u < ADAM_U; ++u) {
      const long long e = base + (long long)u * ADAM_T + threadIdx.x;
      if (e < n4) {
        const int4 sl = reinterpret_cast<const int4*>(s.slot)[e];
        float4 g;
        g.x = sl.x >= 0 ? s.g[sl.x] : 0.f;
        g.y = sl.y >= 0 ? s.g[sl.y] : 0.f;
        g.z = sl.z >= 0 ? s.g[sl.z] : 0.f;
        g.w = sl.w >= 0 ? s.g[sl.w] : 0.f;
        float4 var = var4[e], m = m4[e], v = v4[e];
        F4_APPLY(adam_dense1, var, m, v, g, h);
        var4[e] = var;
        m4[e] = m;
        v4[e] = v;
      } else if (e == n4) {
        for (long long i = n4 * 4; i < s.n; ++i) {
          const int sl = s.slot[i];
          adam_dense1(s.var[i], s.m[i], s.v[i], sl >= 0 ? s.g[sl] : 0.f, h);
        }
      }
    }
  } else if (s.kind == RSX_ADAM_TABLE_ROWS) {
    // lazy_rows mode: n = F*B slots, lpr lanes per slot; only listed rows move (NOT TF semantics)
    const int lpr = s.d >> 2;
    const long long n4 = s.n * lpr;
    const float4* __restrict__ G4 = reinterpret_cast<const float4*>(s.g);
#pragma unroll
    for (int u = 0; u < ADAM_U; ++u) {
      const long long e = base + (long long)u * ADAM_T + threadIdx.x;
      if (e < n4) {
        const long long sidx = e / lpr;
        const int q = (int)(e - sidx * lpr);
        const int f = (int)(sidx / s.B), j = (int)(sidx - (long long)f * s.B);
        if (j < s.nuniq[f]) {
          const long long sl = (long long)f * s.stride + j;
          const long long r = (long long)s.uniq_row[sl] * lpr + q;
          float4 var = var4[r], m = m4[r], v = v4[r];
          const float4 g = G4[sl * lpr + q];
          F4_APPLY(adam_sparse1, var, m, v, g, true, h);
          var4[r] = var;
          m4[r] = m;
          v4[r] = v;
        }
      }
    }
  } else {  // RSX_ADAM_VEC_ROWS
#pragma unroll
    for (int u = 0; u < ADAM_U; ++u) {
      const long long sidx = base + (long long)u * ADAM_T + threadIdx.x;
      if (sidx < s.n) {
        const int f = (int)(sidx / s.B), j = (int)(sidx - (long long)f * s.B);
        if (j < s.nuniq[f]) {
          const long long sl = (long long)f * s.stride + j;
          const int r = s.uniq_row[sl];
          adam_sparse1(s.var[r], s.m[r], s.v[r], s.g[sl], true, h);
        }
      }
    }
  }

  // ticket: the last workgroup to finish advances the beta powers for the next step
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t* ticket = reinterpret_cast<uint32_t*>(a.state + 2);
    const uint32_t t = atomicAdd(ticket, 1u);
    if (t == a.total_blocks - 1u) {
      a.state[0] = b1p * a.b1;
      a.state[1] = b2p * a.b2;
      *ticket = 0u;
      reinterpret_cast<uint32_t*>(a.state)[3] += 1u;
    }
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
  if (!segs_h || !state || nseg <= 0 || nseg > RSX_ADAM_MAX_SEGS) return RSX_EINVAL;
  AdamArgs a;
  uint32_t blocks = 0;
  int k = 0;
  for (int i = 0; i < nseg; ++i) {
    const rsx_adam_seg& s = segs_h[i];
    if (s.n < 0 || !s.var || !s.m || !s.v || !s.g) return RSX_EINVAL;
    long long work;  // float4 (or slot) units
    switch (s.kind) {
      case RSX_ADAM_DENSE: work = (s.n >> 2) + 1; break;
      case RSX_ADAM_TABLE_TF1:
        if (!s.slot || s.d < 4 || (s.d & 3)) return RSX_EINVAL;
        work = s.n * (s.d >> 2);
        break;
      case RSX_ADAM_VEC_SLOT:
        if (!s.slot) return RSX_EINVAL;
        work = (s.n >> 2) + 1;
        break;
      case RSX_ADAM_TABLE_ROWS:
        if (!s.uniq_row || !s.nuniq || s.B <= 0 || s.d < 4 || (s.d & 3)) return RSX_EINVAL;
        work = s.n * (s.d >> 2);
        break;
      case RSX_ADAM_VEC_ROWS:
        if (!s.uniq_row || !s.nuniq || s.B <= 0) return RSX_EINVAL;
        work = s.n;
        break;
      default: return RSX_EINVAL;
    }
    if (s.n == 0) continue;
    SegDev& d = a.seg[k++];
    d.kind = s.kind;
    d.d = s.d;
    d.n = s.n;
    d.var = s.var;
    d.m = s.m;
    d.v = s.v;
    d.g = s.g;
    d.slot = s.slot;
    d.uniq_row = s.uniq_row;
    d.nuniq = s.nuniq;
    d.B = s.B;
    d.stride = s.stride;
    d.zero_grad = s.zero_grad;
    d.blk_begin = blocks;
    blocks += (uint32_t)((work + ADAM_Q - 1) / ADAM_Q);
  }
  if (k == 0) return RSX_OK;
  a.nseg = k;
  a.lr = lr;
  a.b1 = beta1;
  a.b2 = beta2;
  a.eps = eps;
  a.state = state;
  a.total_blocks = blocks;
  hipLaunchKernelGGL(adam_multi_k, dim3(blocks), dim3(ADAM_T), 0, rsx_s(stream), a);
  RSX_CHECK_LAUNCH();
  return RSX_OK;
}
