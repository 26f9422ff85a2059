// The forward of the DCN cross layers for ONE example by one wave (dcn/dcn.py:132-142: x_{l+1} = (x_l . w_l) * x0 + x_l + b_l), shared by
// cross_fwd_k (cross.hip) and gather_cross_fwd_k (embedding.hip, round 4: the input_layer lookup and the cross layers in one launch --
// the gather's lane layout IS this one: lane l holds float4 #(l + 64 v) of the example's row).  One body: the same bits either way.
#pragma once
#include <type_traits>

#include "rsx_common.h"

constexpr int CROSS_MAX_L = 8;
constexpr int CROSS_NV = 4;   // float4 per lane -> dim <= 1024

// float4 #e of a row of n4 float4s; lanes past the row read the last element (in range) and get zero by multiplication:
// a guarded load is compiled into a branch of its own and the vectors of a lane then load one after the other
__device__ __forceinline__ float4 cross_ld(const float4* __restrict__ row, int e, int n4) {
  const float4 v = row[e < n4 ? e : n4 - 1];
  const float f = e < n4 ? 1.f : 0.f;
  return make_float4(v.x * f, v.y * f, v.z * f, v.w * f);
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int m = 1; m < 64; m <<= 1) v += __shfl_xor(v, m);
  return v;
}
__device__ __forceinline__ float dot4(float4 a, float4 b) { return (a.x * b.x + a.y * b.y) + (a.z * b.z + a.w * b.w); }

// x0[v] = float4 #(lane + 64 v) of the example's input row (zero past the row).  s_row: [L] of this example; xL_row (nullable):
// [dim]; cz (nullable, with wout): the logit contribution <x_L, wout>.
__device__ __forceinline__ void cross_fwd_wave(const float4 (&x0)[CROSS_NV], const float* __restrict__ W, const float* __restrict__ Bc,
                                               const float* __restrict__ wout, float* __restrict__ s_row, float* __restrict__ xL_row,
                                               float* __restrict__ cz, const int dim, const int L, const int lane) {
  const int n4 = dim >> 2;
  float4 x[CROSS_NV];
#pragma unroll
  for (int v = 0; v < CROSS_NV; ++v) x[v] = x0[v];
  for (int l = 0; l < L; ++l) {
    float4 w[CROSS_NV], bb[CROSS_NV];
    float part = 0.f;
#pragma unroll
    for (int v = 0; v < CROSS_NV; ++v) {
      const int e = lane + 64 * v;
      w[v] = cross_ld(reinterpret_cast<const float4*>(W) + (size_t)l * n4, e, n4);
      bb[v] = cross_ld(reinterpret_cast<const float4*>(Bc) + (size_t)l * n4, e, n4);
      part += dot4(x[v], w[v]);
    }
    const float s = wave_sum(part);
    if (lane == 0) s_row[l] = s;
#pragma unroll
    for (int v = 0; v < CROSS_NV; ++v) x[v] = f4_add(f4_add(f4_scale(s, x0[v]), x[v]), bb[v]);
  }
  float part = 0.f;
#pragma unroll
  for (int v = 0; v < CROSS_NV; ++v) {
    const int e = lane + 64 * v;
    if (e < n4) {
      if (xL_row != nullptr) reinterpret_cast<float4*>(xL_row)[e] = x[v];
      if (wout != nullptr) part += dot4(x[v], reinterpret_cast<const float4*>(wout)[e]);
    }
  }
  if (cz != nullptr) {
    const float c = wave_sum(part);
    if (lane == 0) cz[0] = c;
  }
}

// ---- backward (dcn/dcn.py:132-142 differentiated), L <= 3: the body of cross_bwd4_k ------------------------------------------
struct CrossBwdArgs {
  const float* x0;     // [B, dim]
  const float* W;      // [L, dim]
  const float* Bc;     // [L, dim]
  const float* s;      // [B, L]
  const float* dxL;    // [B, dim] or null
  const float* gz;     // [B] or null (d loss / d cz)
  const float* wout;   // [dim] (required with gz)
  float* dX;           // [B, dim]
  float* part;         // [RT, 2L+1, dim]: dW (L), dB (L), dwout
  int accumulate;      // dX += (1) or dX = (0)
  int B, dim, L;
};

// Sum over the wave, the same value in every lane: two quad permutes and two row mirrors (DPP, no LDS crossbar: the six
// ds_bpermute of the xor butterfly cost ~0.3 us per call in the backward's dependent chain), then the four row totals in
// row order.
__device__ __forceinline__ float wave_sum_dpp(float v) {
  auto dpp = [](float x, auto ctrl) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), decltype(ctrl)::value, 0xF, 0xF, false));
  };
  v += dpp(v, std::integral_constant<int, 0xB1>{});     // quad_perm [1,0,3,2]
  v += dpp(v, std::integral_constant<int, 0x4E>{});     // quad_perm [2,3,0,1]
  v += dpp(v, std::integral_constant<int, 0x141>{});    // row_half_mirror
  v += dpp(v, std::integral_constant<int, 0x140>{});    // row_mirror
  const int b = __builtin_bit_cast(int, v);
  const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 0));
  const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 16));
  const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 32));
  const float r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 48));
  return ((r0 + r1) + r2) + r3;
}

// The same backward for L <= 3 known at compile time (dcn.py: 3; L = 4 would spill): 4 waves per workgroup, each walking `epw` examples into
// its OWN slab; the four slabs are added in wave order into one partial per workgroup.  Against cross_bwd_k (one wave per
// workgroup, loops unrolled to CROSS_MAX_L: 442 registers, one wave per SIMD, one partial per wave): ~190 registers and
// 70 KB of LDS per workgroup -> two workgroups = 8 waves per CU, a quarter of the partials.
// grid = ceil(B / (4 epw)), block = 256, dyn LDS: 4 (2L+1) dim floats.
// vblock: the workgroup's index among the launch's cross workgroups (blockIdx.x when the launch is cross_bwd4_k's own)
template <int L>
__device__ __forceinline__ void cross_bwd4_body(const CrossBwdArgs& p, const int epw, float* lds, const int vblock) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int n4 = p.dim >> 2;
  constexpr int nvec = 2 * L + 1;
  float4* acc = reinterpret_cast<float4*>(lds) + (size_t)w * nvec * n4;     // this wave's slab (LDS operations of one wave are ordered)
  for (int e = lane; e < nvec * n4; e += 64) acc[e] = F4Z;
  float4 wv[L][CROSS_NV];
#pragma unroll
  for (int l = 0; l < L; ++l)
#pragma unroll
    for (int v = 0; v < CROSS_NV; ++v)
      wv[l][v] = cross_ld(reinterpret_cast<const float4*>(p.W) + (size_t)l * n4, lane + 64 * v, n4);
  for (int rr = 0; rr < epw; ++rr) {
    const int b = (vblock * 4 + w) * epw + rr;
    if (b >= p.B) break;
    float4 x0[CROSS_NV], xs[L][CROSS_NV], x[CROSS_NV], dx[CROSS_NV], dx0[CROSS_NV];
    float sl[L];
#pragma unroll
    for (int v = 0; v < CROSS_NV; ++v) {
      x0[v] = cross_ld(reinterpret_cast<const float4*>(p.x0) + (size_t)b * n4, lane + 64 * v, n4);
      x[v] = x0[v];
      dx0[v] = F4Z;
    }
    const float g = p.gz != nullptr ? p.gz[b] : 0.f;
#pragma unroll
    for (int v = 0; v < CROSS_NV; ++v) {          // (issued before the recomputation below needs anything)
      float4 d = F4Z;
      if (p.dxL != nullptr) d = cross_ld(reinterpret_cast<const float4*>(p.dxL) + (size_t)b * n4, lane + 64 * v, n4);   // uniform
      dx[v] = d;
    }
#pragma unroll
    for (int l = 0; l < L; ++l) sl[l] = p.s[(size_t)b * L + l];
#pragma unroll
    for (int l = 0; l < L; ++l) {                 // recompute x_0 .. x_{L-1} (and x_L in `x`)
#pragma unroll
      for (int v = 0; v < CROSS_NV; ++v) {
        xs[l][v] = x[v];
        const float4 bb = cross_ld(reinterpret_cast<const float4*>(p.Bc) + (size_t)l * n4, lane + 64 * v, n4);
        x[v] = f4_add(f4_add(f4_scale(sl[l], x0[v]), x[v]), bb);
      }
    }
    if (p.gz != nullptr) {
#pragma unroll
      for (int v = 0; v < CROSS_NV; ++v) {
        const int e = lane + 64 * v;
        dx[v] = f4_add(dx[v], f4_scale(g, cross_ld(reinterpret_cast<const float4*>(p.wout), e, n4)));
        if (e < n4) {
          float4* o = acc + (size_t)(2 * L) * n4 + e;
          *o = f4_add(*o, f4_scale(g, x[v]));                       // d wout += gz * x_L
        }
      }
    }
#pragma unroll
    for (int l = L - 1; l >= 0; --l) {
      float part = 0.f;
#pragma unroll
      for (int v = 0; v < CROSS_NV; ++v) part += dot4(dx[v], x0[v]);
      const float ds = wave_sum_dpp(part);
#pragma unroll
      for (int v = 0; v < CROSS_NV; ++v) {
        const int e = lane + 64 * v;
        if (e < n4) {
          float4* db = acc + (size_t)(L + l) * n4 + e;
          float4* dw = acc + (size_t)l * n4 + e;
          *db = f4_add(*db, dx[v]);                                  // dB_l += dx_{l+1}
          *dw = f4_add(*dw, f4_scale(ds, xs[l][v]));                 // dW_l += ds * x_l
        }
        dx0[v] = f4_add(dx0[v], f4_scale(sl[l], dx[v]));             // dx0 += s_l * dx_{l+1}
        dx[v] = f4_add(dx[v], f4_scale(ds, wv[l][v]));               // dx_l = dx_{l+1} + ds * w_l
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
  const float4* sl4 = reinterpret_cast<const float4*>(lds);
  const int per = nvec * n4;
  float4* dst = reinterpret_cast<float4*>(p.part) + (size_t)vblock * per;
  for (int e = threadIdx.x; e < per; e += 256)
    dst[e] = f4_add(f4_add(f4_add(sl4[e], sl4[per + e]), sl4[2 * per + e]), sl4[3 * per + e]);
}

