// DCN cross layers on gfx950, all L layers fused, forward and backward.
// Reference call site: dcn/dcn.py:132-142 -- per layer  x_{l+1} = (x_l . w_l) * x0 + x_l + b_l  with
// w_l, b_l in R^dim (dim = 39*16 = 624).  SURVEY.md 8a row a-9: element-wise/reduce work, HBM-bound:
// fused, one example costs 2 496 B read (+ 2 496 B written in backward), not 2*L passes.
//
// One wave per example; a lane keeps NV float4 of x0 and of the running x_l in registers (dim <= 1024).
// The per-layer dot product is a fixed xor-butterfly.  Backward recomputes x_1..x_L from the saved
// scalars s_l instead of storing L activations.  Batch reductions (dW, dB, d wout) are deterministic:
// each wave accumulates its examples in order into a lane-private LDS slab, the per-wave partials are
// summed in order by cross_reduce_k.  No atomics.
#include <cstdlib>
#include <type_traits>

#include "rsx_common.h"
#include "step_riders_device.h"
#include "cross_device.h"


struct CrossFwdArgs {
  const float* x0;     // [B, dim]
  const float* W;      // [L, dim]
  const float* Bc;     // [L, dim]
  const float* wout;   // [dim] or null
  float* s;            // [B, L]
  float* xL;           // [B, dim] or null
  float* cz;           // [B] <x_L, wout> or null
  int B, dim, L;
};

__global__ __launch_bounds__(256) void cross_fwd_k(const CrossFwdArgs p) {
  const int lane = threadIdx.x & 63;
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= p.B) return;
  const int n4 = p.dim >> 2;
  float4 x0[CROSS_NV];
#pragma unroll
  for (int v = 0; v < CROSS_NV; ++v) x0[v] = cross_ld(reinterpret_cast<const float4*>(p.x0) + (size_t)b * n4, lane + 64 * v, n4);
  cross_fwd_wave(x0, p.W, p.Bc, p.wout, p.s + (size_t)b * p.L, p.xL != nullptr ? p.xL + (size_t)b * p.dim : nullptr,
                 p.cz != nullptr ? p.cz + b : nullptr, p.dim, p.L, lane);
}


// grid = ceil(B/epw), block = 64: ONE wave walks `epw` examples and accumulates dW / dB / dwout in its own
// LDS slab (lane-private float4 slots: no conflicts, no barriers), then writes the slab as one partial.
// dyn LDS: (2L+1)*dim floats.
__global__ __launch_bounds__(64) void cross_bwd_k(const CrossBwdArgs p, int epw) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int lane = threadIdx.x;
  const int n4 = p.dim >> 2;
  const int nvec = 2 * p.L + 1;
  float4* acc = reinterpret_cast<float4*>(lds);
  for (int e = lane; e < nvec * n4; e += 64) acc[e] = F4Z;
  __syncthreads();   // single wave: orders the zero fill before the per-lane read-modify-writes below
  float4 wv[CROSS_MAX_L][CROSS_NV];
#pragma unroll
  for (int l = 0; l < CROSS_MAX_L; ++l)
#pragma unroll
    for (int v = 0; v < CROSS_NV; ++v) {
      const int e = lane + 64 * v;
      wv[l][v] = l < p.L ? cross_ld(reinterpret_cast<const float4*>(p.W) + (size_t)l * n4, e, n4) : F4Z;   // l: uniform
    }
  for (int rr = 0; rr < epw; ++rr) {
    const int b = blockIdx.x * epw + rr;
    if (b >= p.B) break;
    float4 x0[CROSS_NV], xs[CROSS_MAX_L][CROSS_NV], x[CROSS_NV], dx[CROSS_NV], dx0[CROSS_NV];
    float sl[CROSS_MAX_L];
#pragma unroll
    for (int v = 0; v < CROSS_NV; ++v) {
      const int e = lane + 64 * v;
      x0[v] = cross_ld(reinterpret_cast<const float4*>(p.x0) + (size_t)b * n4, e, n4);
      x[v] = x0[v];
      dx0[v] = F4Z;
    }
#pragma unroll
    for (int l = 0; l < CROSS_MAX_L; ++l) {   // recompute x_0 .. x_{L-1} (and x_L in `x`)
      if (l < p.L) {
        sl[l] = p.s[(size_t)b * p.L + l];
#pragma unroll
        for (int v = 0; v < CROSS_NV; ++v) {
          const int e = lane + 64 * v;
          xs[l][v] = x[v];
          const float4 bb = cross_ld(reinterpret_cast<const float4*>(p.Bc) + (size_t)l * n4, e, n4);
          x[v] = f4_add(f4_add(f4_scale(sl[l], x0[v]), x[v]), bb);
        }
      }
    }
    const float g = p.gz != nullptr ? p.gz[b] : 0.f;
#pragma unroll
    for (int v = 0; v < CROSS_NV; ++v) {
      const int e = lane + 64 * v;
      float4 d = F4Z;
      if (p.dxL != nullptr) d = cross_ld(reinterpret_cast<const float4*>(p.dxL) + (size_t)b * n4, e, n4);   // uniform
      if (p.gz != nullptr) {
        d = f4_add(d, f4_scale(g, cross_ld(reinterpret_cast<const float4*>(p.wout), e, n4)));
        if (e < n4) {
          float4* o = acc + (size_t)(2 * p.L) * n4 + e;
          *o = f4_add(*o, f4_scale(g, x[v]));                       // d wout += gz * x_L
        }
      }
      dx[v] = d;
    }
#pragma unroll
    for (int l = CROSS_MAX_L - 1; l >= 0; --l) {
      if (l < p.L) {
        float part = 0.f;
#pragma unroll
        for (int v = 0; v < CROSS_NV; ++v) part += dot4(dx[v], x0[v]);
        const float ds = wave_sum(part);
#pragma unroll
        for (int v = 0; v < CROSS_NV; ++v) {
          const int e = lane + 64 * v;
          if (e < n4) {
            float4* db = acc + (size_t)(p.L + l) * n4 + e;
            float4* dw = acc + (size_t)l * n4 + e;
            *db = f4_add(*db, dx[v]);                                  // dB_l += dx_{l+1}
            *dw = f4_add(*dw, f4_scale(ds, xs[l][v]));                 // dW_l += ds * x_l
          }
          dx0[v] = f4_add(dx0[v], f4_scale(sl[l], dx[v]));             // dx0 += s_l * dx_{l+1}
          dx[v] = f4_add(dx[v], f4_scale(ds, wv[l][v]));               // dx_l = dx_{l+1} + ds * w_l
        }
      }
    }
#pragma unroll
    for (int v = 0; v < CROSS_NV; ++v) {
      const int e = lane + 64 * v;
      if (e < n4) {
        float4 o = f4_add(dx0[v], dx[v]);
        float4* dst = reinterpret_cast<float4*>(p.dX) + (size_t)b * n4 + e;
        if (p.accumulate) o = f4_add(*dst, o);
        *dst = o;
      }
    }
  }
  __syncthreads();
  float4* dst = reinterpret_cast<float4*>(p.part) + (size_t)blockIdx.x * nvec * n4;
  for (int e = lane; e < nvec * n4; e += 64) dst[e] = acc[e];
}

// (wave_sum_dpp and the body of the L <= 3 backward live in cross_device.h since round 6: tower.hip carries the same body as
// extra workgroups of a tower-backward launch)
template <int L>
__global__ __launch_bounds__(256, 2) void cross_bwd4_k(const CrossBwdArgs p, int epw) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  cross_bwd4_body<L>(p, epw, lds, (int)blockIdx.x);
}

// out[j] = sum over the RT per-wave partials (step_riders_device.h cross_reduce_block).  grid = ceil(n/16), block = 256.
__global__ __launch_bounds__(256) void cross_reduce_k(const float* __restrict__ part, int RT, int n, float* __restrict__ dW,
                                                      float* __restrict__ dB, float* __restrict__ dwout, int L, int dim) {
  __shared__ float red[256];
  cross_reduce_block(part, RT, n, dW, dB, dwout, L, dim, blockIdx.x, red);
}

extern "C" int rsx_cross_fwd(const float* x0, const float* W, const float* Bc, const float* wout, float* s, float* xL,
                             float* cz, int B, int dim, int L, rsx_stream_t stream) {
  if (B < 0 || dim <= 0 || L <= 0) return RSX_EINVAL;
  if (B == 0) return RSX_OK;
  if (!x0 || !W || !Bc || !s || (cz != nullptr && wout == nullptr)) return RSX_EINVAL;
  if (dim % 4 != 0 || dim > 256 * CROSS_NV || L > CROSS_MAX_L) return RSX_EUNSUPPORTED;
  CrossFwdArgs p{x0, W, Bc, wout, s, xL, cz, B, dim, L};
  RSX_LAUNCH(cross_fwd_k, dim3((B + 3) / 4), dim3(256), 0, rsx_s(stream), p);
  RSX_CHECK_LAUNCH();
  return RSX_OK;
}

static inline int cross_epw(int B) {   // examples per wave
  static const int forced = getenv("RSX_CROSS_EPW") ? atoi(getenv("RSX_CROSS_EPW")) : 0;   // tuning aid
  if (forced > 0) return forced;
  return B <= 512 ? 1 : (B <= 2048 ? 2 : 4);   // ~4 us of dependent latency per example: parallelism first (measured)
}

static inline int cross_epw4(int B) {   // examples per wave of cross_bwd4_k (a workgroup takes 4x that)
  static const int forced = getenv("RSX_CROSS_EPW4") ? atoi(getenv("RSX_CROSS_EPW4")) : 0;   // tuning aid
  if (forced > 0) return forced;
  return B <= 1024 ? 1 : (B <= 8192 ? 2 : 4);
}

extern "C" size_t rsx_cross_bwd_workspace_floats(int B, int dim, int L) {   // (the one-wave form's: never less than bwd4's)
  const int epw = cross_epw(B);
  return (size_t)((B + epw - 1) / epw) * (size_t)(2 * L + 1) * (size_t)dim;
}

extern "C" int rsx_cross_bwd_defer(const float*, const float*, const float*, const float*, const float*, const float*, const float*,
                                   float*, int, float*, float*, float*, float*, int, int, int, rsx_cross_reduce_job*, rsx_stream_t);
extern "C" int rsx_cross_bwd(const float* x0, const float* W, const float* Bc, const float* s, const float* dxL,
                             const float* gz, const float* wout, float* dX, int accumulate, float* dW, float* dB,
                             float* dwout, float* workspace, int B, int dim, int L, rsx_stream_t stream) {
  return rsx_cross_bwd_defer(x0, W, Bc, s, dxL, gz, wout, dX, accumulate, dW, dB, dwout, workspace, B, dim, L, nullptr, stream);
}
// reduce_out (nullable): the second launch (the sum of the per-wave gradient partials) is handed back as a job instead of
// launched -- rsx_cross_reduce_run runs it, or another launch of the step carries it (rsx_segsum_partials_ride)
extern "C" int rsx_cross_bwd_defer(const float* x0, const float* W, const float* Bc, const float* s, const float* dxL,
                                   const float* gz, const float* wout, float* dX, int accumulate, float* dW, float* dB,
                                   float* dwout, float* workspace, int B, int dim, int L, rsx_cross_reduce_job* reduce_out,
                                   rsx_stream_t stream) {
  if (reduce_out != nullptr) reduce_out->n = 0;
  if (B < 0 || dim <= 0 || L <= 0) return RSX_EINVAL;
  if (B == 0) return RSX_OK;
  if (!x0 || !W || !Bc || !s || !dX || !dW || !dB || !workspace) return RSX_EINVAL;
  if (gz != nullptr && (!wout || !dwout)) return RSX_EINVAL;
  if (dxL == nullptr && gz == nullptr) return RSX_EINVAL;
  if (dim % 4 != 0 || dim > 256 * CROSS_NV || L > CROSS_MAX_L) return RSX_EUNSUPPORTED;
  CrossBwdArgs p{x0, W, Bc, s, dxL, gz, wout, dX, workspace, accumulate, B, dim, L};
  int RT;
  static const int bwd4 = getenv("RSX_CROSS_BWD4") ? atoi(getenv("RSX_CROSS_BWD4")) : 1;   // 0: the one-wave form for every L
  const size_t lds4 = (size_t)4 * (2 * L + 1) * dim * sizeof(float);
  if (bwd4 && L <= 3 && lds4 <= 80 * 1024) {
    const int epw = cross_epw4(B);
    RT = (B + 4 * epw - 1) / (4 * epw);
    auto launch = [&](auto kern) {
      static bool attr_set = false;      // (> 64 KB of dynamic LDS per workgroup)
      if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
        attr_set = true;
      }
      RSX_LAUNCH(kern, dim3(RT), dim3(256), lds4, rsx_s(stream), p, epw);
    };
    switch (L) {
      case 1: launch(cross_bwd4_k<1>); break;
      case 2: launch(cross_bwd4_k<2>); break;
      default: launch(cross_bwd4_k<3>); break;
    }
  } else {
    const int epw = cross_epw(B);
    RT = (B + epw - 1) / epw;
    const size_t lds = (size_t)(2 * L + 1) * dim * sizeof(float);
    if (lds > 64 * 1024) return RSX_EUNSUPPORTED;
    RSX_LAUNCH(cross_bwd_k, dim3(RT), dim3(64), lds, rsx_s(stream), p, epw);
  }
  RSX_CHECK_LAUNCH();
  const int n = (2 * L + 1) * dim;
  if (reduce_out != nullptr) {
    *reduce_out = rsx_cross_reduce_job{workspace, dW, dB, dwout, RT, n, L, dim};
    return RSX_OK;
  }
  RSX_LAUNCH(cross_reduce_k, dim3((n + 15) / 16), dim3(256), 0, rsx_s(stream), workspace, RT, n, dW, dB, dwout,
                     L, dim);
  RSX_CHECK_LAUNCH();
  return RSX_OK;
}

extern "C" int rsx_cross_reduce_run(const rsx_cross_reduce_job* j, rsx_stream_t stream) {
  if (!j) return RSX_EINVAL;
  if (j->n == 0) return RSX_OK;
  if (!j->part || !j->dW || !j->dB || j->RT <= 0 || j->n < 0 || j->L <= 0 || j->dim <= 0) return RSX_EINVAL;
  RSX_LAUNCH(cross_reduce_k, dim3((j->n + 15) / 16), dim3(256), 0, rsx_s(stream), j->part, j->RT, j->n, j->dW, j->dB, j->dwout,
                     j->L, j->dim);
  RSX_CHECK_LAUNCH();
  return RSX_OK;
}
