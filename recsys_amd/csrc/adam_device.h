// TF-1.x Adam sweep as device/host building blocks shared by adam.hip (the stand-alone launch) and tower.hip (where
// slices of the untouched-row sweep ride along in the tower launches).  See adam.hip for the semantics.
#pragma once
#include "adam_fast.h"
#include "rsx_common.h"

struct SegDev {
  int32_t kind, d;
  long long n;
  float *var, *m, *v, *g;
  const int32_t *slot, *uniq_row, *nuniq;
  int32_t B, stride, zero_grad;
  uint32_t blk_begin;
  int32_t g_rep, g_rep_stride4;      // *_ROWS kinds: replicas of g summed in order, float4 units apart (rsx_adam_seg.g_replicas)
};
constexpr int ADAM_WMAX = RSX_ADAM_WINDOW_MAX - 1;     // extra slot maps of an optimizer window
struct AdamArgs {
  SegDev seg[RSX_ADAM_MAX_SEGS];
  // *_COLD kinds of an optimizer WINDOW (rsx_adam_seg.slot_w): the slot maps of the window's later steps, shared by every COLD
  // segment of the launch (kernel arguments are limited to 4 KB and a launch may embed two AdamArgs)
  // -- equally spaced (one allocation; base + stride keeps them out of the scalar registers)
  const int32_t* slot_w0;
  long long slot_w_stride;      // in int32 elements
  int32_t nw;
  int32_t nseg;
  float lr, b1, b2, eps;
  float* state;
  uint32_t total_blocks;
  // Window sweeps that RIDE in the steps' launches, a slice per step (round 4, rsx_adam_slice.window_block_u / alphas_from_state):
  // win_u = float4 per lane of a TABLE_TF1_COLD block (0: ADAM_U) -- smaller blocks spread a slice over the idle CUs of a
  // latency-bound launch --; alpha_src != 0: the window's step sizes are READ from state words 8.. (the slice of the window's
  // first step computed them from the beta powers, which later steps have advanced since).
  int32_t win_u, alpha_src;
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

#ifndef RSX_ADAM_U
#define RSX_ADAM_U 8
#endif
#ifndef RSX_ADAM_NT
#define RSX_ADAM_NT 0
#endif
#ifndef RSX_ADAM_WIN_HB
#define RSX_ADAM_WIN_HB 2      // float4 per lane and pipeline stage of the window sweep (adam_window_block)
#endif
constexpr int ADAM_T = 256;         // threads per workgroup
constexpr int ADAM_U = RSX_ADAM_U;  // float4 per lane
constexpr long long ADAM_Q = (long long)ADAM_T * ADAM_U;  // float4 per workgroup
// DENSE segments take 2 float4 per lane: a dense arena is small (0.3 MB for DeepFM) and rides in the latency-bound scatter
// launch, where 8 sequential IEEE div + sqrt updates per lane (~1 900 instructions of one wave) made its 9 workgroups the
// LAST ones to finish (stamps: 8.6 us after the launch's first workgroup, against 4.8 for the window pass).
constexpr int ADAM_U_DENSE = 2;
constexpr long long ADAM_Q_DENSE = (long long)ADAM_T * ADAM_U_DENSE;

// "Am I the last workgroup of this launch?" for thread 0 of every workgroup, after its block's work (and a __syncthreads):
// arrival counters by workgroup index modulo 32 (one 128-byte line each: state words 4 + 32 r), then the shared ticket
// (state word 2) for the last arrival of each residue.  All counters are left at zero.  total = workgroups of the launch.
__device__ __forceinline__ bool adam_arrive_last(float* state, const uint32_t total) {
  uint32_t* w = reinterpret_cast<uint32_t*>(state);
  const uint32_t r = blockIdx.x & 31u;
  const uint32_t n_r = (total + 31u - r) >> 5;                  // workgroups with this residue
  uint32_t* sub = w + 4 + 32 * r;
  if (atomicAdd(sub, 1u) != n_r - 1u) return false;
  *sub = 0u;
  const uint32_t n_top = total < 32u ? total : 32u;             // residues in use
  if (atomicAdd(w + 2, 1u) != n_top - 1u) return false;
  w[2] = 0u;
  return true;
}

// alpha of the window's later steps: step j of the window runs with the beta powers advanced j times -- the same fp32
// products the per-step advance of the powers makes.  (Named scalars, no array: a dynamically indexed local array is
// promoted to LDS / scratch by this toolchain.)
struct AlphaW {
  float a0, a1, a2, a3, a4, a5, a6;
  __device__ __forceinline__ float get(int j) const {
    return j == 0 ? a0 : j == 1 ? a1 : j == 2 ? a2 : j == 3 ? a3 : j == 4 ? a4 : j == 5 ? a5 : a6;
  }
};
static_assert(ADAM_WMAX == 7, "AlphaW holds 7 steps");
__device__ __forceinline__ AlphaW alpha_window(const AdamArgs& a, int nw, float b1p, float b2p) {
  AlphaW aw = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (nw > 0) {
    float p1 = b1p, p2 = b2p;
#define RSX_ALPHA_NEXT(dst) p1 *= a.b1; p2 *= a.b2; dst = a.lr * sqrtf(1.0f - p2) / (1.0f - p1)
    RSX_ALPHA_NEXT(aw.a0); RSX_ALPHA_NEXT(aw.a1); RSX_ALPHA_NEXT(aw.a2); RSX_ALPHA_NEXT(aw.a3);
    RSX_ALPHA_NEXT(aw.a4); RSX_ALPHA_NEXT(aw.a5); RSX_ALPHA_NEXT(aw.a6);
#undef RSX_ALPHA_NEXT
  }
  return aw;
}

// The step sizes of a sweep over 1 + nw steps: from the beta powers (state words 0, 1), or -- a slice of a window sweep that runs
// at a LATER step of its window -- the values the window's first slice left in state words 8.. (the same floats).
__device__ __forceinline__ void adam_step_sizes(const AdamArgs& a, int nw, float& alpha0, AlphaW& aw) {
  if (a.alpha_src) {
    alpha0 = a.state[8];
    aw = AlphaW{a.state[9], a.state[10], a.state[11], a.state[12], a.state[13], a.state[14], a.state[15]};
    return;
  }
  const float b1p = a.state[0], b2p = a.state[1];
  alpha0 = a.lr * sqrtf(1.0f - b2p) / (1.0f - b1p);
  aw = alpha_window(a, nw, b1p, b2p);
}
// The step sizes of the window's 1 + NW steps, for the lazy window pass of segsum_adam_k (csrc/embedding.hip), which applies
// a row's zero-gradient updates of several steps -- also FUTURE ones -- in one go: state word 8 + j = the step size of
// window step j, from the products the per-step advance of the beta powers will make.  ONE thread of the window's first sweep
// launch (or first slice) calls this.
template <int NW>
__device__ __forceinline__ void adam_publish_step_sizes(const AdamArgs& a) {
  const float b1p = a.state[0], b2p = a.state[1];
  const AlphaW aw = alpha_window(a, NW, b1p, b2p);
  a.state[8] = a.lr * sqrtf(1.0f - b2p) / (1.0f - b1p);
#pragma unroll
  for (int j = 0; j < NW; ++j) a.state[9 + j] = aw.get(j);
}

// One workgroup (ADAM_T = 256 threads) of the sweep: block `blk` of the launch-wide block index space of `a`.
__device__ __forceinline__ void adam_block(const AdamArgs& a, const uint32_t blk, const int tid = threadIdx.x) {
  Hp h;
  h.b1 = a.b1;
  h.b2 = a.b2;
  h.omb1 = 1.0f - a.b1;
  h.omb2 = 1.0f - a.b2;
  h.eps = a.eps;
  int si = 0;
#pragma unroll 1
  for (int k = 1; k < a.nseg; ++k)
    if (blk >= a.seg[k].blk_begin) si = k;
  const SegDev& s = a.seg[si];
  // window of 1 + nw steps (COLD kinds only): step j of the window runs with the beta powers advanced j times -- the same
  // fp32 products the per-step advance of the powers makes
  const bool is_cold = s.kind == RSX_ADAM_TABLE_TF1_COLD || s.kind == RSX_ADAM_VEC_COLD;
  const int nw = is_cold ? a.nw : 0;
  AlphaW aw;
  adam_step_sizes(a, nw, h.alpha, aw);
  // the window's extra slot maps are equally spaced (one allocation): base + stride instead of 7 pointers in scalar registers
  const int32_t* __restrict__ swb = a.slot_w0;
  const long long sws = a.slot_w_stride;
  const long long base = (long long)(blk - s.blk_begin) * (s.kind == RSX_ADAM_DENSE ? ADAM_Q_DENSE : ADAM_Q);
  float4* __restrict__ var4 = reinterpret_cast<float4*>(s.var);
  float4* __restrict__ m4 = reinterpret_cast<float4*>(s.m);
  float4* __restrict__ v4 = reinterpret_cast<float4*>(s.v);

  if (s.kind == RSX_ADAM_TABLE_TF1 || s.kind == RSX_ADAM_TABLE_TF1_COLD) {
    const bool cold_only = s.kind == RSX_ADAM_TABLE_TF1_COLD;   // touched rows are left to the TABLE_ROWS launch
    const int lpr = s.d >> 2;
    const long long n4 = s.n * lpr;
    const float4* __restrict__ G4 = reinterpret_cast<const float4*>(s.g);
#pragma unroll
    for (int u = 0; u < ADAM_U; ++u) {
      const long long e = base + (long long)u * ADAM_T + tid;
      if (e < n4) {
        const long long row = e / lpr;
        const int q = (int)(e - row * lpr);
        int sl = s.slot[row];
        for (int l = 0; l < nw; ++l) sl &= swb[(long long)l * sws + row];      // window (see adam_window_block for the fast form)
        if (cold_only && sl >= 0) continue;
#if RSX_ADAM_NT
        float4 var = __builtin_nontemporal_load(&var4[e]), m = __builtin_nontemporal_load(&m4[e]),
               v = __builtin_nontemporal_load(&v4[e]);
#else
        float4 var = var4[e], m = m4[e], v = v4[e];
#endif
        const bool has = sl >= 0;
        const float4 g = has ? G4[(long long)sl * lpr + q] : F4Z;
        F4_APPLY(adam_sparse1, var, m, v, g, has, h);
#pragma unroll 1
        for (int j = 0; j < nw; ++j) {
          Hp hj = h;
          hj.alpha = aw.get(j);
          F4_APPLY(adam_sparse1, var, m, v, F4Z, false, hj);
        }
#if RSX_ADAM_NT
        __builtin_nontemporal_store(var, &var4[e]);
        __builtin_nontemporal_store(m, &m4[e]);
        __builtin_nontemporal_store(v, &v4[e]);
#else
        var4[e] = var;
        m4[e] = m;
        v4[e] = v;
#endif
      }
    }
  } else if (s.kind == RSX_ADAM_DENSE) {
    float4* __restrict__ g4 = reinterpret_cast<float4*>(s.g);
    const long long n4 = s.n >> 2;
    // B > 1: the gradient is the sum of B replicas' arenas, `stride` floats apart inside an all-gathered buffer, added in
    // rank order (every rank computes the same bits) -- the data-parallel dense all-reduce folded into the update
    const int nrep = s.B > 1 ? s.B : 1;
    const long long rs4 = (long long)s.stride >> 2;
#pragma unroll
    for (int u = 0; u < ADAM_U_DENSE; ++u) {
      const long long e = base + (long long)u * ADAM_T + tid;
      if (e < n4) {
        float4 var = var4[e], m = m4[e], v = v4[e];
        float4 g = g4[e];
        int r = 1;
        // (many replicas -- fm.py's per-example head gradients, round 4: 16 loads in flight per trip; the same ascending order)
        for (; r + 15 < nrep; r += 16) {
          float4 t[16];
#pragma unroll
          for (int q = 0; q < 16; ++q) t[q] = g4[e + (long long)(r + q) * rs4];
#pragma unroll
          for (int q = 0; q < 16; ++q) { g.x += t[q].x; g.y += t[q].y; g.z += t[q].z; g.w += t[q].w; }
        }
        for (; r < nrep; ++r) {
          const float4 gr = g4[e + r * rs4];
          g.x += gr.x; g.y += gr.y; g.z += gr.z; g.w += gr.w;
        }
        F4_APPLY(adam_dense1, var, m, v, g, h);
        var4[e] = var;
        m4[e] = m;
        v4[e] = v;
        if (s.zero_grad) g4[e] = F4Z;
      } else if (e == n4) {  // scalar tail (n not a multiple of 4)
        for (long long i = n4 * 4; i < s.n; ++i) {
          float g = s.g[i];
          for (int r = 1; r < nrep; ++r) g += s.g[i + (long long)r * s.stride];
          adam_dense1(s.var[i], s.m[i], s.v[i], g, h);
          if (s.zero_grad) s.g[i] = 0.f;
        }
      }
    }
  } else if (s.kind == RSX_ADAM_VEC_SLOT || s.kind == RSX_ADAM_VEC_COLD) {
    const bool cold_only = s.kind == RSX_ADAM_VEC_COLD;         // touched elements keep their state for VEC_ROWS_DENSE
    const long long n4 = s.n >> 2;
#pragma unroll
    for (int u = 0; u < ADAM_U; ++u) {
      const long long e = base + (long long)u * ADAM_T + tid;
      if (e < n4) {
        int4 sl = reinterpret_cast<const int4*>(s.slot)[e];
        for (int l = 0; l < nw; ++l) {
          const int4 t = reinterpret_cast<const int4*>(swb + (long long)l * sws)[e];
          sl.x &= t.x; sl.y &= t.y; sl.z &= t.z; sl.w &= t.w;
        }
        float4 g;
        g.x = (!cold_only && sl.x >= 0) ? s.g[sl.x] : 0.f;     // COLD: g is not provided (touched elements are restored below)
        g.y = (!cold_only && sl.y >= 0) ? s.g[sl.y] : 0.f;
        g.z = (!cold_only && sl.z >= 0) ? s.g[sl.z] : 0.f;
        g.w = (!cold_only && sl.w >= 0) ? s.g[sl.w] : 0.f;
        float4 var = var4[e], m = m4[e], v = v4[e];
        const float4 var0 = var, m0 = m, v0 = v;
        F4_APPLY(adam_dense1, var, m, v, g, h);
        for (int j = 0; j < nw; ++j) {
          Hp hj = h;
          hj.alpha = aw.get(j);
          F4_APPLY(adam_dense1, var, m, v, F4Z, hj);
        }
        if (cold_only) {   // element-wise: leave the touched elements exactly as they were
          if (sl.x >= 0) { var.x = var0.x; m.x = m0.x; v.x = v0.x; }
          if (sl.y >= 0) { var.y = var0.y; m.y = m0.y; v.y = v0.y; }
          if (sl.z >= 0) { var.z = var0.z; m.z = m0.z; v.z = v0.z; }
          if (sl.w >= 0) { var.w = var0.w; m.w = m0.w; v.w = v0.w; }
        }
        var4[e] = var;
        m4[e] = m;
        v4[e] = v;
      } else if (e == n4) {
        for (long long i = n4 * 4; i < s.n; ++i) {
          int sl = s.slot[i];
          for (int l = 0; l < nw; ++l) sl &= swb[(long long)l * sws + i];
          if (cold_only && sl >= 0) continue;
          adam_dense1(s.var[i], s.m[i], s.v[i], sl >= 0 ? s.g[sl] : 0.f, h);
          for (int j = 0; j < nw; ++j) {
            Hp hj = h;
            hj.alpha = aw.get(j);
            adam_dense1(s.var[i], s.m[i], s.v[i], 0.f, hj);
          }
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
      const long long e = base + (long long)u * ADAM_T + tid;
      if (e < n4) {
        const long long sidx = e / lpr;
        const int q = (int)(e - sidx * lpr);
        const int f = (int)(sidx / s.B), j = (int)(sidx - (long long)f * s.B);
        if (j < s.nuniq[f]) {
          const long long sl = (long long)f * s.stride + j;
          const long long r = (long long)s.uniq_row[sl] * lpr + q;
          float4 var = var4[r], m = m4[r], v = v4[r];
          float4 g = G4[sl * lpr + q];
          for (int rr = 1; rr < s.g_rep; ++rr) g = f4_add(g, G4[(long long)rr * s.g_rep_stride4 + sl * lpr + q]);   // replicas, in rank order
          F4_APPLY(adam_sparse1, var, m, v, g, true, h);
          var4[r] = var;
          m4[r] = m;
          v4[r] = v;
        }
      }
    }
  } else {  // RSX_ADAM_VEC_ROWS (sparse formula) / RSX_ADAM_VEC_ROWS_DENSE (ApplyAdam formula)
#pragma unroll
    for (int u = 0; u < ADAM_U; ++u) {
      const long long sidx = base + (long long)u * ADAM_T + tid;
      if (sidx < s.n) {
        const int f = (int)(sidx / s.B), j = (int)(sidx - (long long)f * s.B);
        if (j < s.nuniq[f]) {
          const long long sl = (long long)f * s.stride + j;
          const int r = s.uniq_row[sl];
          float g = s.g[sl];
          for (int rr = 1; rr < s.g_rep; ++rr) g += s.g[(long long)rr * 4 * s.g_rep_stride4 + sl];
          if (s.kind == RSX_ADAM_VEC_ROWS_DENSE) adam_dense1(s.var[r], s.m[r], s.v[r], g, h);
          else adam_sparse1(s.var[r], s.m[r], s.v[r], g, true, h);
        }
      }
    }
  }
}

#ifndef RSX_ADAM_WIN_FAST
#define RSX_ADAM_WIN_FAST 1
#endif

// ---- scalar zero-gradient update (the first-order vector's rows in the lazy window pass of segsum_adam_k) ----------------
// adam_fast.h's sequences on ONE element, behind a guard on the operands themselves (checked after they have been computed):
// v1 in the square root's domain -- or +0 together with a +0 numerator, where both forms give var - (+0) --, numerator and
// denominator in the division's.  A wave with an active element outside takes the IEEE form for this update.
__device__ __forceinline__ float rsx_sqrt1_fast(const float x) {
  const float r = __builtin_amdgcn_rsqf(x), g = x * r, hh = r * 0.5f;
  return __builtin_fmaf(__builtin_fmaf(-g, g, x), hh, g);
}
__device__ __forceinline__ float rsx_div1_fast(const float n, const float d) {
  float r = __builtin_amdgcn_rcpf(d);
  r = __builtin_fmaf(__builtin_fmaf(-d, r, 1.f), r, r);
  const float q = n * r;
  return __builtin_fmaf(__builtin_fmaf(-d, q, n), r, q);
}
template <bool DENSE>
__device__ __forceinline__ void adam_zero_grad1(float& var, float& m, float& v, const bool active, const Hp& h) {
  float m1, v1, n;
  if constexpr (DENSE) {                      // adam_dense1 with g = 0, operation by operation
    m1 = m + (0.f - m) * h.omb1;
    v1 = v + (0.f * 0.f - v) * h.omb2;
    n = m1 * h.alpha;
  } else {                                    // adam_sparse1 with has = false
    m1 = m * h.b1;
    v1 = v * h.b2;
    n = h.alpha * m1;
  }
  const bool vz = __float_as_uint(v1) == 0u && __float_as_uint(n) == 0u;
  const float d = (vz ? 0.f : rsx_sqrt1_fast(vz ? 1.f : v1)) + h.eps;
  const uint32_t vb = __float_as_uint(v1), nb = __float_as_uint(n), na = nb & 0x7fffffffu, db = __float_as_uint(d);
  constexpr uint32_t V_LO = (127u - 96u) << 23, V_HI = (127u + 41u) << 23;      // rsx_sqrt2_fast: 2^-96 <= v1 < 2^41
  constexpr uint32_t D_LO = (127u - 30u) << 23, D_HI = (127u + 21u) << 23;      // rsx_div2_fast:  2^-30 <= d <= 2^21
  constexpr uint32_t N_LO = (127u - 94u) << 23, N_HI = (127u + 34u) << 23;      //                 n == +0 or 2^-94 <= |n| <= 2^34
  const bool ok = (vz || (vb - V_LO) < (V_HI - V_LO)) && (db - D_LO) <= (D_HI - D_LO) &&
                  (nb == 0u || (na - N_LO) <= (N_HI - N_LO));
  if (RSX_ADAM_WIN_FAST && __builtin_amdgcn_ballot_w64(active && !ok) == 0ull) {
    if (active) {
      var = var - rsx_div1_fast(n, d);
      m = m1;
      v = v1;
    }
  } else if (active) {
    if constexpr (DENSE) adam_dense1(var, m, v, 0.f, h);
    else adam_sparse1(var, m, v, 0.f, false, h);
  }
}

// ---- window sweep: packed fast form of the zero-gradient update --------------------------------------------------------

// One zero-gradient TF-1 update of two elements -- adam_sparse1(has = false) -- with the packed square root / division.
__device__ __forceinline__ void adam_zero_grad2(rsx_f2& var, rsx_f2& m, rsx_f2& v, const float alpha, const Hp& h) {
  m = m * h.b1;
  v = v * h.b2;
  const rsx_f2 n = m * alpha;
  const rsx_f2 d = rsx_sqrt2_fast(v) + h.eps;
  var = var - rsx_div2_fast(n, d);
}
// Does this element leave the domain on which the packed form returns the bits of the IEEE form, at any of the window's
// 1 + NW updates?  (Evaluated once per window on the loaded state; m and v only decay inside a window.)
//   v:  2^-86 <= v <= 2^40                            -> every v_j in [2^-94, 2^40], d_j = sqrt(v_j) + eps in [2^-30, 2^21]
//       v == +0 with m == +0 (a row no gradient has reached yet): the sweep computes with 1 in place of v -- n_j = +0, so
//       the quotient is +0 whatever the denominator -- and stores the zero back (adam_window_block)
//   m:  m_lo <= |m| <= 2^30 (m_lo = 2^-90 / min alpha) -> every n_j = alpha_j m_j has 2^-92 <= |n_j| <= 2^34: inside both domains
//       m == +0                                        -> n_j = +0, both forms return +0
//       |m| < m_lo (denormals, -0: where the moments of a row end up ~900 steps after its last gradient) and |var| >= 2^-24
//                                                      -> |n_j| < 2^-88, so |q_j| < 2^-56 in EITHER form (d_j >= 2^-30; the
//                                                         fast chain's result is bounded by 3 |n_j| / d_j): var - q_j == var
// Anything else (NaN / inf moments, a tiny moment under a tiny weight, v below 2^-86) sends the wave to the IEEE form.
__device__ __forceinline__ bool adam_win_guard1(const float var, const float m, const float v, const uint32_t m_lo_bits) {
  const uint32_t vb = __float_as_uint(v), mb = __float_as_uint(m), ma = mb & 0x7fffffffu;
  const uint32_t va = __float_as_uint(var) & 0x7fffffffu;
  constexpr uint32_t V_LO = (127u - 86u) << 23, V_HI = (127u + 40u) << 23, M_HI = (127u + 30u) << 23;
  constexpr uint32_t VAR_LO = (127u - 24u) << 23;
  const bool v_ok = (vb - V_LO) <= (V_HI - V_LO);
  const bool m_mid = (ma - m_lo_bits) <= (M_HI - m_lo_bits);
  const bool m_tiny = ma < m_lo_bits && va >= VAR_LO;
  return !(mb == 0u ? (v_ok || vb == 0u) : (v_ok && (m_mid || m_tiny)));
}
__device__ __forceinline__ bool adam_win_guard4(const float4& var, const float4& m, const float4& v, const uint32_t m_lo_bits) {
  return adam_win_guard1(var.x, m.x, v.x, m_lo_bits) | adam_win_guard1(var.y, m.y, v.y, m_lo_bits) |
         adam_win_guard1(var.z, m.z, v.z, m_lo_bits) | adam_win_guard1(var.w, m.w, v.w, m_lo_bits);
}

// The untouched rows of a window of 1 + NW steps (its own kernel, adam_window_k: the packed forms' temporaries would
// otherwise count against every carrier kernel and the one-step sweep, which inherit adam_block's 61 registers).
//   - TABLE_TF1_COLD blocks: batches of HB float4 per lane in phases, so that every load of a batch is in flight before
//     the first use: slot maps (clamped index), then var / m / v unconditionally (a skipped row costs its read, ~1 % of the
//     traffic; a guarded load would serialise the batch), then the 1 + NW updates, then the stores of the rows that moved.
//     Software pipeline over the block's NB batches: the loads of batch b + 1 are issued before batch b is updated.
//     (Measured against it at 8 / 4 steps per window, us per sweep of the DeepFM state: this form 86.5 / 61.0; persistent
//     workgroups walking batches with a stride of the grid 85.2 / 64.7 at 4 per CU, 81.0 / 59.9 at 8 per CU; one batch per
//     workgroup without a pipeline 97.5 / 65.6.  With the state L2-resident the updates alone cost ~9 us per step of the
//     window: from ~4 steps per window on the kernel is VALU-bound, not HBM-bound.)
//   - blocks of the other kinds (the first-order vector, ...): adam_block.
// U = float4 per lane of a block (== a.win_u, or ADAM_U when that is 0: the host picks the instantiation)
template <int NW, int U = ADAM_U>
__device__ __forceinline__ void adam_window_block(const AdamArgs& a, const uint32_t blk, const int tid = threadIdx.x) {
  constexpr int nw = NW;        // == a.nw (the host picks the instantiation)
  constexpr int HB = RSX_ADAM_WIN_HB;
  constexpr int NB = U / HB;
  static_assert(U % HB == 0 && U <= ADAM_U, "U");
  int si = 0;
#pragma unroll 1
  for (int k = 1; k < a.nseg; ++k)
    if (blk >= a.seg[k].blk_begin) si = k;
  const SegDev& s = a.seg[si];
  if (s.kind != RSX_ADAM_TABLE_TF1_COLD) {
    adam_block(a, blk, tid);
    return;
  }
  Hp h;
  h.b1 = a.b1;
  h.b2 = a.b2;
  h.omb1 = 1.0f - a.b1;
  h.omb2 = 1.0f - a.b2;
  h.eps = a.eps;
  AlphaW aw;
  adam_step_sizes(a, nw, h.alpha, aw);
  // Packed fast form of the zero-gradient updates (adam_fast.h): usable when the hyper-parameters keep every operand of a
  // guarded element inside the fast forms' domains for all 1 + NW updates (b1^8 >= 1/4, b2^8 >= 2^-8, alphas within 4x).
  float amin = h.alpha, amax = h.alpha;
#pragma unroll
  for (int j = 0; j < NW; ++j) {
    amin = fminf(amin, aw.get(j));
    amax = fmaxf(amax, aw.get(j));
  }
  const bool fast_ok = RSX_ADAM_WIN_FAST && a.b1 >= 0.85f && a.b1 < 1.f && a.b2 >= 0.5f && a.b2 < 1.f && a.eps >= 0x1p-30f &&
                       a.eps <= 1.f && amin >= 0x1p-24f && amax <= 16.f && amax <= 4.f * amin;
  const uint32_t m_lo_bits = __float_as_uint(fast_ok ? 0x1p-90f / amin : 1.f);
  const int32_t* __restrict__ swb = a.slot_w0;
  const int sws = (int)a.slot_w_stride;           // (< 2^28: adam_build_args) 32-bit element offsets: scalar base + vector offset
  float4* __restrict__ var4 = reinterpret_cast<float4*>(s.var);
  float4* __restrict__ m4 = reinterpret_cast<float4*>(s.m);
  float4* __restrict__ v4 = reinterpret_cast<float4*>(s.v);
  const long long base = (long long)(blk - s.blk_begin) * ((long long)ADAM_T * U);
  const int lpr = s.d >> 2;
  const long long n4 = s.n * lpr;
  const bool pow2 = (lpr & (lpr - 1)) == 0;
  const int lsh = 31 - __clz(lpr);
  struct Batch {
    float4 var[HB], m[HB], v[HB];
    long long ec[HB];
    int t[HB];
    bool live[HB];
  };
  auto issue = [&](const int b, Batch& B) {
#pragma unroll
    for (int u = 0; u < HB; ++u) {
      const long long e = base + (long long)(b * HB + u) * ADAM_T + tid;
      B.ec[u] = e < n4 ? e : n4 - 1;
      B.live[u] = e < n4;
      const long long row = pow2 ? (B.ec[u] >> lsh) : (B.ec[u] / lpr);
      B.t[u] = s.slot[row];
      // (untouched = -1 = all ones: one value >= 0 clears the sign of the AND.)  All of them in flight at once.
#pragma unroll
      for (int l = 0; l < NW; ++l) B.t[u] &= swb[l * sws + (int)row];
    }
#pragma unroll
    for (int u = 0; u < HB; ++u) {
#if RSX_ADAM_NT
      B.var[u] = __builtin_nontemporal_load(&var4[B.ec[u]]);
      B.m[u] = __builtin_nontemporal_load(&m4[B.ec[u]]);
      B.v[u] = __builtin_nontemporal_load(&v4[B.ec[u]]);
#else
      B.var[u] = var4[B.ec[u]];
      B.m[u] = m4[B.ec[u]];
      B.v[u] = v4[B.ec[u]];
#endif
    }
  };
  auto store = [&](const Batch& B) {
#pragma unroll
    for (int u = 0; u < HB; ++u) {
      if (B.live[u] && B.t[u] < 0) {
#if RSX_ADAM_NT
        __builtin_nontemporal_store(B.var[u], &var4[B.ec[u]]);
        __builtin_nontemporal_store(B.m[u], &m4[B.ec[u]]);
        __builtin_nontemporal_store(B.v[u], &v4[B.ec[u]]);
#else
        var4[B.ec[u]] = B.var[u];
        m4[B.ec[u]] = B.m[u];
        v4[B.ec[u]] = B.v[u];
#endif
      }
    }
  };
  // Batches in which some element of the wave leaves the packed forms' domain are left untouched by the pipeline and redone
  // in the IEEE form after it, from memory (a wave-uniform decision; the hot loop then holds the packed form only).
  uint32_t redo = fast_ok ? 0u : (1u << NB) - 1u;
  auto finish = [&](const int b, Batch& B) {
    if (!fast_ok) return;
    bool bad = false;
#pragma unroll
    for (int u = 0; u < HB; ++u) bad |= adam_win_guard4(B.var[u], B.m[u], B.v[u], m_lo_bits);
    if (__builtin_amdgcn_ballot_w64(bad) != 0ull) {
      redo |= 1u << b;
      return;
    }
    rsx_f2 var2[HB][2], m2[HB][2], v2[HB][2];
#pragma unroll
    for (int u = 0; u < HB; ++u) {
      var2[u][0] = (rsx_f2){B.var[u].x, B.var[u].y}; var2[u][1] = (rsx_f2){B.var[u].z, B.var[u].w};
      m2[u][0] = (rsx_f2){B.m[u].x, B.m[u].y}; m2[u][1] = (rsx_f2){B.m[u].z, B.m[u].w};
      // (v == +0, which the guard admits under m == +0 only: compute with 1, store the zero back)
      v2[u][0] = (rsx_f2){B.v[u].x == 0.f ? 1.f : B.v[u].x, B.v[u].y == 0.f ? 1.f : B.v[u].y};
      v2[u][1] = (rsx_f2){B.v[u].z == 0.f ? 1.f : B.v[u].z, B.v[u].w == 0.f ? 1.f : B.v[u].w};
    }
    auto step = [&](const float alpha) {
#pragma unroll
      for (int u = 0; u < HB; ++u) {
        adam_zero_grad2(var2[u][0], m2[u][0], v2[u][0], alpha, h);
        adam_zero_grad2(var2[u][1], m2[u][1], v2[u][1], alpha, h);
      }
    };
    step(h.alpha);
#pragma unroll 1
    for (int j = 0; j < NW; ++j) step(aw.get(j));
#pragma unroll
    for (int u = 0; u < HB; ++u) {
      B.var[u] = make_float4(var2[u][0].x, var2[u][0].y, var2[u][1].x, var2[u][1].y);
      B.m[u] = make_float4(m2[u][0].x, m2[u][0].y, m2[u][1].x, m2[u][1].y);
      B.v[u] = make_float4(B.v[u].x == 0.f ? 0.f : v2[u][0].x, B.v[u].y == 0.f ? 0.f : v2[u][0].y,
                           B.v[u].z == 0.f ? 0.f : v2[u][1].x, B.v[u].w == 0.f ? 0.f : v2[u][1].y);
    }
    store(B);
  };
  {
    Batch bt[2];
    issue(0, bt[0]);
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      if (b + 1 < NB) issue(b + 1, bt[(b + 1) & 1]);
      finish(b, bt[b & 1]);
    }
  }
  redo = __builtin_amdgcn_readfirstlane(redo);
  if (redo != 0u) {
#pragma unroll 1
    for (int b = 0; b < NB; ++b) {
      if (!(redo >> b & 1u)) continue;
      Batch B;
      issue(b, B);
#pragma unroll
      for (int u = 0; u < HB; ++u) {     // (unrolled: a dynamically indexed register array is demoted to scratch)
        F4_APPLY(adam_sparse1, B.var[u], B.m[u], B.v[u], F4Z, false, h);
#pragma unroll 1
        for (int j = 0; j < NW; ++j) {          // the later steps of the window, back to back in registers
          Hp hj = h;
          hj.alpha = aw.get(j);
          F4_APPLY(adam_sparse1, B.var[u], B.m[u], B.v[u], F4Z, false, hj);
        }
      }
      store(B);
    }
  }
}

// Host: validates the segment list and lays the segments out over the launch-wide block index space.
// Returns an rsx_status; *blocks_out = number of workgroups (0 when there is nothing to do).
static inline int adam_build_args(const rsx_adam_seg* segs_h, int nseg, float* state, float lr, float beta1, float beta2,
                                  float eps, AdamArgs& a, uint32_t* blocks_out, int win_u = 0, int alpha_src = 0) {
  if (!segs_h || !state || nseg <= 0 || nseg > RSX_ADAM_MAX_SEGS) return RSX_EINVAL;
  if (win_u != 0 && win_u != 2 && win_u != 4 && win_u != ADAM_U) return RSX_EINVAL;
  a.win_u = win_u == ADAM_U ? 0 : win_u;
  a.alpha_src = alpha_src ? 1 : 0;
  uint32_t blocks = 0;
  int k = 0, n_cold = 0;
  a.nw = 0;
  a.slot_w0 = nullptr;
  a.slot_w_stride = 0;
  for (int i = 0; i < nseg; ++i) {
    const rsx_adam_seg& s = segs_h[i];
    if (s.n < 0 || !s.var || !s.m || !s.v) return RSX_EINVAL;
    long long work;  // float4 (or slot) units
    switch (s.kind) {
      case RSX_ADAM_DENSE:
        if (!s.g) return RSX_EINVAL;
        if (s.B > 1 && (s.stride < s.n || (s.stride & 3))) return RSX_EINVAL;     // replica sum: 16-byte aligned blocks
        work = (s.n >> 2) + 1;
        break;
      case RSX_ADAM_TABLE_TF1:
      case RSX_ADAM_TABLE_TF1_COLD:
        if (!s.slot || (!s.g && s.kind == RSX_ADAM_TABLE_TF1) || s.d < 4 || (s.d & 3)) return RSX_EINVAL;
        if (s.kind == RSX_ADAM_TABLE_TF1 && s.slot_w[0]) return RSX_EINVAL;          // windows: COLD kinds only
        work = s.n * (s.d >> 2);
        break;
      case RSX_ADAM_VEC_SLOT:
      case RSX_ADAM_VEC_COLD:
        if (!s.slot || (!s.g && s.kind == RSX_ADAM_VEC_SLOT)) return RSX_EINVAL;
        if (s.kind == RSX_ADAM_VEC_SLOT && s.slot_w[0]) return RSX_EINVAL;
        work = (s.n >> 2) + 1;
        break;
      case RSX_ADAM_TABLE_ROWS:
        if (!s.g || !s.uniq_row || !s.nuniq || s.B <= 0 || s.d < 4 || (s.d & 3)) return RSX_EINVAL;
        if (s.g_replicas > 1 && (s.g_replica_stride <= 0 || (s.g_replica_stride & 3) || s.g_replica_stride >= (1ll << 33))) return RSX_EINVAL;
        work = s.n * (s.d >> 2);
        break;
      case RSX_ADAM_VEC_ROWS:
      case RSX_ADAM_VEC_ROWS_DENSE:
        if (!s.g || !s.uniq_row || !s.nuniq || s.B <= 0) return RSX_EINVAL;
        if (s.g_replicas > 1 && (s.g_replica_stride <= 0 || (s.g_replica_stride & 3) || s.g_replica_stride >= (1ll << 33))) return RSX_EINVAL;
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
    const bool rows_kind = s.kind == RSX_ADAM_TABLE_ROWS || s.kind == RSX_ADAM_VEC_ROWS || s.kind == RSX_ADAM_VEC_ROWS_DENSE;
    d.g_rep = rows_kind && s.g_replicas > 1 ? s.g_replicas : 1;
    d.g_rep_stride4 = d.g_rep > 1 ? (int32_t)(s.g_replica_stride >> 2) : 0;
    if (s.kind == RSX_ADAM_TABLE_TF1_COLD || s.kind == RSX_ADAM_VEC_COLD) {     // the window's extra slot maps (AdamArgs)
      int nw = 0;
      while (nw < ADAM_WMAX && s.slot_w[nw] != nullptr) ++nw;
      for (int j = nw; j < ADAM_WMAX; ++j)
        if (s.slot_w[j] != nullptr) return RSX_EINVAL;                          // a prefix, no holes
      const long long str = nw > 1 ? (long long)(s.slot_w[1] - s.slot_w[0]) : 0;
      for (int j = 1; j < nw; ++j)
        if (s.slot_w[j] != s.slot_w[0] + j * str) return RSX_EINVAL;           // equally spaced (see rsx.h)
      if (nw > 0 && (reinterpret_cast<uintptr_t>(s.slot_w[0]) & 15u || (str & 3) || str < 0 || str >= (1ll << 28)))
        return RSX_EINVAL;                                                      // int4 reads (VEC_COLD), 32-bit offsets
      if (n_cold++ == 0) {
        a.nw = nw;
        a.slot_w0 = s.slot_w[0];
        a.slot_w_stride = str;
      } else if (nw != a.nw || (nw > 0 && (a.slot_w0 != s.slot_w[0] || a.slot_w_stride != str))) {
        return RSX_EINVAL;                                                      // one window per launch
      }
    }
    d.blk_begin = blocks;
    const long long quantum = s.kind == RSX_ADAM_DENSE ? ADAM_Q_DENSE
                              : (s.kind == RSX_ADAM_TABLE_TF1_COLD && a.win_u > 0 ? (long long)ADAM_T * a.win_u : ADAM_Q);
    blocks += (uint32_t)((work + quantum - 1) / quantum);
  }
  a.nseg = k;
  a.lr = lr;
  a.b1 = beta1;
  a.b2 = beta2;
  a.eps = eps;
  a.state = state;
  a.total_blocks = blocks;
  *blocks_out = k == 0 ? 0u : blocks;
  return RSX_OK;
}

// A slice [blk_lo, blk_hi) of a COLD sweep carried by another kernel's launch as extra workgroups.
struct AdamSlice {
  AdamArgs args;
  uint32_t blk_lo, n_blk;     // n_blk == 0: no slice
};
// Workgroup order of a carrier launch with T workgroups of its own and R riders.  Appending the riders after the carrier's
// workgroups serialises the two when T alone fills the chip (every slot is taken by a latency/MFMA-bound tile until the
// first ones retire; the HBM-bound riders then run alone: measured, cin_bwd_dw_bf16_k 35 us + 27 us = 62 us).  Dealing them
// in alternating runs of 8 (consecutive workgroup ids go to the 8 XCDs round-robin, so both kinds land on every XCD) makes
// both resident from the start.  -> rider? and the index within its kind.
struct RiderSplit {
  bool rider;
  uint32_t idx;
};
__device__ __forceinline__ RiderSplit rider_split(const uint32_t lin, const uint32_t T, const uint32_t R) {
  const uint32_t m8 = (T < R ? T : R) & ~7u;          // alternating part: m8 of each kind
  if (lin < 2u * m8) return {(lin & 8u) != 0u, ((lin >> 4) << 3) | (lin & 7u)};
  const uint32_t r = lin - 2u * m8;                   // the rest: the carrier's own first
  if (r < T - m8) return {false, m8 + r};
  return {true, m8 + (r - (T - m8))};
}

static inline int adam_build_slice(const rsx_adam_slice* sl_h, AdamSlice& out) {
  out.n_blk = 0;
  out.blk_lo = 0;
  if (sl_h == nullptr) return RSX_OK;
  uint32_t blocks = 0;
  const int rc = adam_build_args(sl_h->segs, sl_h->nseg, sl_h->state, sl_h->lr, sl_h->beta1, sl_h->beta2, sl_h->eps,
                                 out.args, &blocks, sl_h->window_block_u, sl_h->alphas_from_state);
  if (rc != RSX_OK) return rc;
  if (sl_h->blk_hi < sl_h->blk_lo || sl_h->blk_hi > blocks) return RSX_EINVAL;
  out.blk_lo = sl_h->blk_lo;
  out.n_blk = sl_h->blk_hi - sl_h->blk_lo;
  return RSX_OK;
}
