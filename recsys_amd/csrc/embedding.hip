// Embedding path of the CTR hot path on gfx950: multi-field gather + first-order + FM second order
// (forward), per-field LDS sort -> unique rows (dedup), sorted segment-sum (row-wise gradient scatter).
// Reference call sites: fm/fm.py:117-129, deepfm/deepfm.py:85-98, xdeepfm/xdeepfm.py:128,185,
// dcn/dcn.py:123 (see include/rsx.h for the per-entry-point citations).
//
// HBM layout: tables[R, D] fp32, rows of D*4 bytes (64 B at D=16) -> one row = LPR = D/4 lanes of
// float4.  A wave handles 64/LPR (b, f) pairs per load instruction; E[B, F*D] is written fully
// coalesced (consecutive lanes -> consecutive 16 B).  These kernels are HBM/latency-bound integer +
// fp32 streaming work: no MFMA here by design.
#include "rsx_common.h"
#include "sort_device.h"
#include "adam_device.h"
#include "hot_adam_device.h"
#include "gather_device.h"
#include "step_riders_device.h"
#include "cross_device.h"

RSX_STAMP_DECL
// (profiling build only) which workgroup of a launch stamps slots [16 g, 16 g + 16): g = 0 -> workgroup 0, g = 1 -> workgroup 24
#define RSX_STAMP2(slot) do { RSX_STAMP(slot, blockIdx.x == 0); RSX_STAMP(16 + (slot), blockIdx.x == 24); } while (0)
// segsum_adam_k: workgroup 0 (field 0's helpers of huge segments at B = 4096) -> slots 32.., workgroup 432 (row owners of a
// 100 000-row field at B = 4096) -> slots 48..
#ifndef RSX_STAMP_BLK2
#define RSX_STAMP_BLK2 432      // (-DRSX_STAMP_BLK2=100: a row-owner workgroup of a hashed field at B = 256)
#endif
#define RSX_STAMP3(slot) do { RSX_STAMP(32 + (slot), blockIdx.x == 0); RSX_STAMP(48 + (slot), blockIdx.x == RSX_STAMP_BLK2); } while (0)

// ------------------------------------------------------------------ forward --------------------
// One wave per example b (gather_device.h gather_fm_example: lane = (pair slot j, float4 quarter q), fields j, j + PPP, ...;
// the field reductions are xor-butterflies over the j bits, a fixed tree, so results are deterministic run to run).
template <int D>
__global__ void gather_fm_fwd_k(const float* __restrict__ tables, const float* __restrict__ w1,
                                const int32_t* __restrict__ row_off, const int32_t* __restrict__ ids,
                                float* __restrict__ E, float* __restrict__ S, float* __restrict__ y1,
                                float* __restrict__ y2, uint64_t w1_mask, int B, int F) {
  const int b = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (b >= B) return;
  gather_fm_example<D>(tables, w1, row_off, ids, E, S, y1, y2, w1_mask, b, F, threadIdx.x & 63);
}

// The same gather with the step's dedup sort riding along: workgroups [0, n_gather) gather 4 examples each (one wave per
// example), the next F workgroups sort one field each (field_sort_block, ids only).  The sort is the first consumer-free
// work of a training step -- its slot map tells which table rows the step touches -- so with it in the FIRST launch every
// later launch (both tower-forward layers included) may carry a slice of the untouched-row optimizer sweep.
template <int D>
__global__ __launch_bounds__(256) void gather_fm_sort_k(const float* __restrict__ tables, const float* __restrict__ w1,
                                                        const int32_t* __restrict__ row_off, const int32_t* __restrict__ ids,
                                                        float* __restrict__ E, float* __restrict__ S, float* __restrict__ y1,
                                                        float* __restrict__ y2, uint64_t w1_mask, int B, int F, int n_gather,
                                                        const SortArgs sort) {
  extern __shared__ __attribute__((aligned(16))) uint32_t gs_lds[];
  if ((int)blockIdx.x >= n_gather) {
    field_sort_block(sort, blockIdx.x - n_gather, gs_lds);
    return;
  }
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= B) return;
  gather_fm_example<D>(tables, w1, row_off, ids, E, S, y1, y2, w1_mask, b, F, threadIdx.x & 63);
}

// fm.py's whole forward + head in ONE launch (round 4; fm/fm.py:117-133,146-149): one wave per example gathers the rows (no
// E: fm.py's backward recomputes the FM term from the table row), forms the first-order sum and the FM term, and goes on to
//   z = wo[0] relu(y1 + c0) + wo[1] y2 + bo,  prob = sigmoid(z),  ce,  dz = (prob - label) loss_scale
// and the head's backward: gy1 = d loss / d y1, gy2 = d loss / d y2 for the scatter, and this EXAMPLE's contribution to the
// four dense gradients, written as one row of `terms` in the dense arena's own layout (terms[b, off_c0] = dc0, [off_wo + 0 / 1]
// = dwo, [off_bo] = dbo, every other element of the row's first `n_dense` floats 0) + its cross-entropy term at [b, n_dense].
// Rows of 16 consecutive examples are pre-added in example order (terms [ceil(B/16), stride]); the optimizer launch sums the
// group rows in order (an RSX_ADAM_DENSE segment with B = ceil(batch/16) "replicas" `stride` floats apart), the host reads the
// loss from column n_dense when somebody asks for it: no separate head launch, no cross-workgroup reduction here.
struct FmHeadFused {
  const float* c0; const float* wo; const float* bo; const float* labels;
  float* prob; float* gy1; float* gy2; float* terms;
  int stride, n_dense, off_c0, off_wo, off_bo;
  float loss_scale;
};
// One workgroup = 16 waves = 16 consecutive examples; their rows of dense-gradient terms are added in example order by the
// lanes of wave 0 (one per column) and leave as ONE row of `terms` per workgroup: the optimizer launch then adds B / 16 rows
// instead of B (its 3-thread dense segment walked 256 rows in 16 dependent trips: 4 us at the tail of the scatter launch).
template <int D>
__global__ __launch_bounds__(1024) void gather_fm_head_k(const float* __restrict__ tables, const float* __restrict__ w1,
                                                         const int32_t* __restrict__ row_off, const int32_t* __restrict__ ids,
                                                         float* __restrict__ S, uint64_t w1_mask, int B, int F,
                                                         const FmHeadFused h) {
  __shared__ float rows[16][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int b = blockIdx.x * 16 + w;
  const bool live = b < B;
  const int bc = live ? b : B - 1;                  // (a wave past the batch repeats the last example and contributes nothing)
  const float c0 = h.c0[0], w0 = h.wo[0], w1o = h.wo[1], bo = h.bo[0], y = h.labels[bc];     // (before the gather's chain)
  float y1v, y2v;
  gather_fm_example<D>(tables, w1, row_off, ids, nullptr, live ? S : nullptr, nullptr, nullptr, w1_mask, bc, F, lane, &y1v, &y2v);
  y1v = __shfl(y1v, 0);          // (the first-order sum is complete in the lanes of quarter 0 only)
  // the arithmetic of fm_head_k, one example per wave (every lane computes it)
  const float v0 = y1v + c0, v1 = y2v;
  const float t0 = v0 > 0.f ? v0 : 0.f;
  const float z = w0 * t0 + w1o * v1 + bo;
  const float pr = 1.f / (1.f + expf(-z));
  const float ce = fmaxf(z, 0.f) - z * y + log1pf(expf(-fabsf(z)));
  const float dz = (pr - y) * h.loss_scale;
  const float g0 = v0 > 0.f ? dz * w0 : 0.f;
  if (lane == 0 && live) {
    h.prob[b] = pr;
    h.gy1[b] = g0;
    h.gy2[b] = dz * w1o;
  }
  {
    float t = 0.f;
    t = lane == h.off_c0 ? g0 : t;
    t = lane == h.off_wo ? dz * t0 : t;
    t = lane == h.off_wo + 1 ? dz * v1 : t;
    t = lane == h.off_bo ? dz : t;
    t = lane == h.n_dense ? ce : t;
    rows[w][lane] = t;
  }
  __syncthreads();
  if (w == 0 && lane < h.stride) {
    const int n = B - blockIdx.x * 16 < 16 ? B - blockIdx.x * 16 : 16;     // examples of this group (>= 1)
    float t = rows[0][lane];
    for (int i = 1; i < n; ++i) t += rows[i][lane];                       // ascending example order: fm_head_k's terms mode adds the same way
    h.terms[(size_t)blockIdx.x * h.stride + lane] = t;
  }
}

// F = 1 (tf.nn.embedding_lookup of one table: DIN's item / category / history lookups, din/din.py:96-105): a plain row
// gather.  The multi-field kernel would keep one wave per id with D/4 of its 64 lanes busy; here a wave serves 64/(D/4)
// ids at once, one float4 per lane, fully coalesced on the output side.
// dcn.py's input_layer lookup AND its cross layers' forward in ONE launch (round 4; dcn/dcn.py:125-142): a wave gathers the 39 rows
// of one example -- lane (j, q) holds quarter q of fields j, j + 16, j + 32: float4 #(lane + 64 v) of the example's row, which is
// exactly the lane layout of cross_fwd_wave (cross_device.h) -- writes E as the gather does and runs the cross layers on the
// registers it already holds.  Replaces gather_fm_fwd_k + cross_fwd_k (the second re-read the 10 MB the first had just written and
// was a launch of its own on the step's chain); same bits as the two launches.  D = 16, F <= 64.
__global__ __launch_bounds__(256) void gather_cross_fwd_k(const float* __restrict__ tables, const int32_t* __restrict__ row_off,
                                                          const int32_t* __restrict__ ids, float* __restrict__ E,
                                                          const float* __restrict__ W, const float* __restrict__ Bc,
                                                          const float* __restrict__ wout, float* __restrict__ s,
                                                          float* __restrict__ cz, int B, int F, int L) {
  constexpr int LPR = 4, PPP = 16;
  const int lane = threadIdx.x & 63;
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= B) return;
  const int q = lane % LPR, j = lane / LPR;
  const float4* __restrict__ T4 = reinterpret_cast<const float4*>(tables);
  float4* __restrict__ E4 = reinterpret_cast<float4*>(E);
  const int32_t* idb = ids + (size_t)b * F;
  int row[CROSS_NV];
  bool ok[CROSS_NV];
#pragma unroll
  for (int k = 0; k < CROSS_NV; ++k) {     // fields j, j + 16, j + 32, j + 48: ids and offsets first, then the row loads, all unconditional
    const int f = j + k * PPP;
    ok[k] = f < F;
    const int fc = ok[k] ? f : F - 1;
    row[k] = row_off[fc] + idb[fc];
  }
  float4 x0[CROSS_NV];
#pragma unroll
  for (int k = 0; k < CROSS_NV; ++k) x0[k] = T4[(size_t)row[k] * LPR + q];
#pragma unroll
  for (int k = 0; k < CROSS_NV; ++k) {
    if (ok[k]) E4[((size_t)b * F + j + k * PPP) * LPR + q] = x0[k];
    else x0[k] = F4Z;
  }
  cross_fwd_wave(x0, W, Bc, wout, s + (size_t)b * L, nullptr, cz != nullptr ? cz + b : nullptr, F * 16, L, lane);
}

template <int D>
__global__ __launch_bounds__(256) void gather_rows_k(const float* __restrict__ table, const int32_t* __restrict__ row_off,
                                                     const int32_t* __restrict__ ids, float* __restrict__ out,
                                                     long long n) {
  constexpr int LPR = D / 4;
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long e = t / LPR;
  if (e >= n) return;
  const int q = (int)(t - e * LPR);
  const long long row = (long long)row_off[0] + ids[e];
  reinterpret_cast<float4*>(out)[e * LPR + q] = reinterpret_cast<const float4*>(table)[row * LPR + q];
}

// ------------------------------------------------------------------ dedup: per-field LDS sort ---
// One workgroup per field (sort_device.h).  n = padded power of two (>= 128), T = min(1024, n/2) threads.
__global__ __launch_bounds__(1024) void field_sort_k(const SortArgs a) {
  extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
  field_sort_block(a, blockIdx.x, lds);
}

struct SortMulti {
  SortArgs a[RSX_ADAM_WINDOW_MAX];
};
__global__ __launch_bounds__(1024) void field_sort_multi_k(const SortMulti m) {
  extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
  field_sort_block(m.a[blockIdx.y], blockIdx.x, lds);
}

// Several workgroups per field (sort_device.h field_sort_split_block): grid (F * G, jobs), workgroup x = f * G + g.
static __device__ SortSplitScratch g_sort_split_scr;     // zero between launches (the last range of a field resets its words)
struct SortSplit {
  SortArgs a[RSX_ADAM_WINDOW_MAX];
  int G, W;
};
__global__ __launch_bounds__(1024) void field_sort_split_k(const SortSplit m) {
  extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
  const int f = blockIdx.x / m.G, g = blockIdx.x - f * m.G;
  field_sort_split_block(m.a[blockIdx.y], f, g, m.G, m.W, lds, &g_sort_split_scr.w[blockIdx.y][f][0]);
}
// launches the split form when it applies (all jobs share B / F / stride); returns RSX_OK, an error, or 1 = not applicable
static int launch_sort_split(const SortArgs* jobs, int njobs, int max_rows_per_field, hipStream_t st) {
  static const int on = getenv("RSX_SORT_SPLIT") ? atoi(getenv("RSX_SORT_SPLIT")) : 1;
  // One sort per launch only: a window's k sorts already fill the chip with one workgroup per (field, batch), and cut in two
  // they measured slower (dcn.py, 4 batches of 4 096: 0.2087 vs 0.2040 ms per step).
  const int G = (on && njobs == 1) ? rsx_sort_split_parts(jobs[0], max_rows_per_field) : 0;
  if (G < 2) return 1;
  SortSplit m;
  for (int k = 0; k < RSX_ADAM_WINDOW_MAX; ++k) m.a[k] = jobs[k < njobs ? k : 0];
  m.G = G;
  m.W = rsx_sort_split_words(max_rows_per_field);
  const int T = jobs[0].n >= 2048 ? 1024 : 512;
  const size_t lds = rsx_sort_split_lds_bytes(jobs[0], T, m.W);
  if (lds > 64 * 1024) {
    static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(field_sort_split_k),
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (attr != hipSuccess) return RSX_EUNSUPPORTED;
  }
  RSX_LAUNCH(field_sort_split_k, dim3((unsigned)(jobs[0].F * G), (unsigned)njobs), dim3(T), lds, st, m);
  return RSX_OK;
}

// ------------------------------------------------------------------ backward: sorted segment-sum -
// A wave owns GPW = 64/LPR consecutive unique rows (f, j0..j0+GPW-1) of one field; LPR lanes (one float4
// each) form the group of one row.  Per entry of a segment the contribution is
//   (gy2[b]*S[b,:] - gy2[b]*T[row,:]) + dX[b, f*D:(f+1)*D]   (term order = TF autodiff of fm/fm.py:127-129)
// and E[b,f,:] == T[row,:] for the whole segment, so the row is read once instead of once per example.
//  * short segments (<= SEG_SHORT entries): the group sums sequentially in ascending b -- exactly the
//    order of a CPU unsorted_segment_sum -- with the loads of 4 entries in flight at a time.
//  * long segments (tiny-vocabulary fields, Zipf heads): the whole wave cooperates: the segment is cut
//    into GPW contiguous sub-ranges, each group sums its sub-range in ascending b, and the GPW partials
//    are added in ascending sub-range order.  A fixed order -> deterministic; no atomics anywhere.
constexpr int SEG_SHORT = 16;

// Per-example inputs (dX, S, gy1, gy2) either contiguous over the batch (b == 0) or RANK-BLOCKED as an all-gather leaves
// them: example e lives in block r = e / b at local index e % b, blocks `stride` floats apart (each pointer names its
// array inside block 0).  Lets the data-parallel scatter read the gathered buffer in place.
struct ExBlocks {
  int b;
  long long stride;
};
__device__ __forceinline__ void ex_locate(const ExBlocks& xb, int e, int& i, size_t& off) {
  if (xb.b > 0) {
    const int r = (int)((unsigned)e / (unsigned)xb.b);
    i = e - r * xb.b;
    off = (size_t)r * (size_t)xb.stride;
  } else {
    i = e;
    off = 0;
  }
}

template <int LPR>
struct SegCtx {
  ExBlocks xb;
  const float4* __restrict__ S4;
  const float4* __restrict__ X4;
  const float* __restrict__ gy1;
  const float* __restrict__ gy2;
  const int32_t* __restrict__ pf;
  int F, f, q;
  bool do1;
};

// sequential ascending sum over sorted positions [i0, i1) of one field; 4 entries' loads in flight.
// Every load is UNCONDITIONAL (positions past the end re-read the last one, absent inputs are compiled out by FM / XG, the
// first-order scalar comes through a pointer that always names a valid [B] array) and the tail is masked on the VALUES: a
// load behind a condition costs a branch and ends the compiler's counting of the loads in flight.
template <int LPR, bool FM, bool XG, bool BLK>
__device__ __forceinline__ void seg_range_sum_t(const SegCtx<LPR>& c, const float4 e, int i0, int i1, float4& acc,
                                                float& a1) {
  const float* __restrict__ h1 = c.gy1 != nullptr ? c.gy1 : (FM ? c.gy2 : reinterpret_cast<const float*>(c.X4));
  for (int i = i0; i < i1; i += 4) {
    int bb[4];
    float g[4], h[4];
    float4 s[4], x[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) bb[k] = c.pf[i + k < i1 ? i + k : i1 - 1];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      int bi = bb[k];
      size_t bo = 0;
      if constexpr (BLK) ex_locate(c.xb, bb[k], bi, bo);      // (the uniform "rank-blocked?" branch stays outside the loop)
      if constexpr (FM) {
        g[k] = c.gy2[bo + bi];
        s[k] = c.S4[bo / 4 + (size_t)bi * LPR + c.q];
      }
      if constexpr (XG) x[k] = c.X4[bo / 4 + ((size_t)bi * c.F + c.f) * LPR + c.q];
      h[k] = h1[bo + bi];
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (i + k < i1) {
        float4 t;
        if constexpr (FM) {
          t = f4_sub(f4_scale(g[k], s[k]), f4_scale(g[k], e));
          if constexpr (XG) t = f4_add(t, x[k]);
        } else {
          t = x[k];
        }
        acc = f4_add(acc, t);
        if (c.do1) a1 += h[k];
      }
    }
  }
}
template <int LPR>
__device__ __forceinline__ void seg_range_sum(const SegCtx<LPR>& c, const float4 e, int i0, int i1, float4& acc,
                                              float& a1) {
  if (c.xb.b > 0) {                       // all kernel-uniform
    if (c.gy2 != nullptr) {
      if (c.X4 != nullptr) seg_range_sum_t<LPR, true, true, true>(c, e, i0, i1, acc, a1);
      else seg_range_sum_t<LPR, true, false, true>(c, e, i0, i1, acc, a1);
    } else {
      seg_range_sum_t<LPR, false, true, true>(c, e, i0, i1, acc, a1);
    }
  } else if (c.gy2 != nullptr) {
    if (c.X4 != nullptr) seg_range_sum_t<LPR, true, true, false>(c, e, i0, i1, acc, a1);
    else seg_range_sum_t<LPR, true, false, false>(c, e, i0, i1, acc, a1);
  } else {
    seg_range_sum_t<LPR, false, true, false>(c, e, i0, i1, acc, a1);
  }
}

// Two-stage mode (B > 512).  Stage A (segsum_partials_k): every SEG_CHUNK-position chunk of the sorted order sums its
// overlap with the LONG segments (> SEG_SHORT entries) -- uniform work units however skewed the ids are -- and the
// chunk where a long segment starts appends it to the field's long list.  Stage B: row-owner waves sum the segments of
// <= SEG_SHORT entries entry by entry (the oracle's order); every long segment gets a helper wave of its own (the
// waves past the row owners) that adds its chunk partials in ascending chunk order.  SEG_CHUNK == SEG_SHORT, so a
// chunk overlaps at most two long segments: partial slot 0 = the segment holding the chunk's first position, slot 1 =
// the one holding its last.
constexpr int SEG_CHUNK = SEG_SHORT;
constexpr int SEG_HUGE = 256;   // above: a helper WAVE per segment; SEG_SHORT+1..SEG_HUGE: a helper GROUP per segment
// stage-B waves per field: row owners + long-segment groups + huge-segment waves (+ rounding slack)
__host__ __device__ inline int seg_waves_per_field(int B, int gpw, bool two_stage) {
  const int own = (B + gpw - 1) / gpw;
  return two_stage ? own + (B / (SEG_SHORT + 1) + gpw - 1) / gpw + B / (SEG_HUGE + 1) + 2 : own;
}

// sequential ascending sum of chunk partials t in [t0, t1) of one long segment whose first chunk is `base`
template <int LPR>
__device__ __forceinline__ void partial_range_sum(const float4* __restrict__ P4, const float* __restrict__ P1, size_t base,
                                                  bool first_slot1, int q, bool do1, int t0, int t1, float4& acc,
                                                  float& a1) {
  for (int t = t0; t < t1; t += 8) {
    float4 v[8];
    float h[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const bool ok = t + k < t1;
      const size_t idx = (base + (size_t)(ok ? t + k : t0)) * 2 + ((t + k) == 0 && first_slot1 ? 1 : 0);
      v[k] = ok ? P4[idx * LPR + q] : F4Z;
      h[k] = (ok && do1) ? P1[idx] : 0.f;
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      if (t + k < t1) {
        acc = f4_add(acc, v[k]);
        if (do1) a1 += h[k];
      }
    }
  }
}

// Optional prefetch for the fused touched-row Adam (segsum_adam_k): the row's optimizer state is
// requested as soon as the row is known, so that its latency hides behind the segment walk instead of following it.
// (PRE template flag of segsum_wave / segsum_wave2; whenever they return `valid`, the state is loaded.  A nullable pointer to
// this struct with a `loaded` flag kept the whole struct in SCRATCH memory on this toolchain -- 57 scratch instructions in
// segsum_adam_k, each store sharing the in-order vmcnt counter with the loads that matter.)
struct AdamRowPrefetch {
  const float* tables;
  const float* m_t;
  const float* v_t;
  // first-order vector riding in the same launch (or any valid arrays with stride 0: the loads are unconditional)
  const float* w1;
  const float* m_w;
  const float* v_w;
  int w1_stride;
  float4 var, m, v;
  float w, mw, vw;
};

// Stage B's work as a COMPACT list of wave-sized units: field f contributes ceil(nu_f / GPW) row-owner units, ceil(nlong_f
// / GPW) long-segment units and nhuge_f huge-segment units, in that order, fields in order.  (A grid sized for the worst
// case -- B unique rows in every field -- launched 2 808 workgroups at 4 096 examples x 39 fields of which ~690 had work;
// each of the others still cost a dispatch slot, a barrier and a returning atomic on the arrival counter: the launch took
// 15 us while no single workgroup lived longer than 7.)  Every wave derives the list from the F-entry count arrays with
// one load per lane and a wave scan; waves walk it with a grid stride.
struct SegUnits {
  int incl;          // lane f: units of fields 0 .. f
  int nu, nl, nh;    // lane f: unique rows / long / huge segments of field f
  int total;
};
template <int GPW>
__device__ __forceinline__ SegUnits seg_units(const int32_t* __restrict__ nuniq, const SegPartials& part, int F, int stride) {
  const int lane = threadIdx.x & 63;
  const int lc = lane < F ? lane : F - 1;
  SegUnits u;
  u.nu = nuniq[lc];
  u.nl = part.counts(F, stride)[lc];
  u.nh = part.counts(F, stride)[F + lc];
  int incl = lane < F ? (u.nu + GPW - 1) / GPW + (u.nl + GPW - 1) / GPW + u.nh : 0;
#pragma unroll
  for (int d = 1; d < RSX_WAVE; d <<= 1) {
    const int o = __shfl_up(incl, d);
    if (lane >= d) incl += o;
  }
  u.incl = incl;
  u.total = __shfl(incl, RSX_WAVE - 1);
  return u;
}
// unit (wave-uniform, < u.total) -> field, unit within the field, the field's counts
__device__ __forceinline__ void seg_unit_locate(const SegUnits& u, int unit, int& f, int& wf, int& nu, int& nl, int& nh) {
  f = __popcll(__ballot(u.incl <= unit));
  const int below = __shfl(u.incl, f > 0 ? f - 1 : 0);
  wf = unit - (f > 0 ? below : 0);
  nu = __shfl(u.nu, f);
  nl = __shfl(u.nl, f);
  nh = __shfl(u.nh, f);
}
constexpr int SEG_STAGE_B_MAX_WG = 1024;   // stage-B workgroups of a launch at most (4 waves each, grid stride over the units)

// Stage B wave body (same contract as segsum_wave) for unit wf of field f: units [0, nact) own GPW unique rows each (the
// short ones were summed by stage A: their sums are picked up); the next ceil(nlong / GPW) units are the helpers of the
// field's long segments (a group each), the last nhuge units those of its huge segments (a wave each).
template <int D, bool PRE>
__device__ __forceinline__ bool segsum_wave2(int f, int wf, int nu, int nlong, int nhuge, const float* __restrict__ tables,
                                             const float* __restrict__ S,
                                             const float* __restrict__ dX, const float* __restrict__ gy1,
                                             const float* __restrict__ gy2, const int32_t* __restrict__ perm,
                                             const int32_t* __restrict__ seg_off, const int32_t* __restrict__ uniq_row,
                                             const int32_t* __restrict__ nuniq, uint64_t w1_mask, int B, int F, int stride,
                                             int null_row, const SegPartials& part, const ExBlocks& xb, bool& valid, size_t& sl,
                                             float4& acc, float& a1, float4& e, int& row, bool& do1, bool& staged,
                                             const bool load_staged, AdamRowPrefetch& pre) {
  constexpr int LPR = D / 4;
  constexpr int GPW = RSX_WAVE / LPR;
  const int lane = threadIdx.x & 63;
  const int q = lane % LPR, g = lane / LPR;
  staged = false;
  // the row's optimizer state, requested as soon as the row is known (segsum_adam_k): one round trip less per row
  auto prefetch = [&](const int r) {
    if constexpr (PRE) {
      const size_t o = (size_t)r * LPR + q;
      pre.var = reinterpret_cast<const float4*>(pre.tables)[o];
      pre.m = reinterpret_cast<const float4*>(pre.m_t)[o];
      pre.v = reinterpret_cast<const float4*>(pre.v_t)[o];
      const size_t wi = (size_t)r * pre.w1_stride;
      pre.w = pre.w1[wi];
      pre.mw = pre.m_w[wi];
      pre.vw = pre.v_w[wi];
    }
  };
  auto table_row = [&](const int r) -> float4 {     // the FM term's row: the prefetched variable where there is one
    if constexpr (PRE) return pre.var;
    else return reinterpret_cast<const float4*>(tables)[(size_t)r * LPR + q];
  };
  valid = false;
  const int nrow = part.null_of(f, null_row);                       // this field's padding row, or -1
  const int nact = (nu + GPW - 1) / GPW;
  const int32_t* so = seg_off + (size_t)f * (stride + 1);
  do1 = gy1 != nullptr && q == 0 && ((w1_mask >> f) & 1ull);
  acc = F4Z;
  a1 = 0.f;
  e = F4Z;
  if (wf < nact) {
    const int j = wf * GPW + g;
    const bool own = j < nu;
    sl = (size_t)f * stride + (own ? j : wf * GPW);
    const int beg = own ? so[j] : 0;
    int end = own ? so[j + 1] : 0;
    row = own ? uniq_row[sl] : 0;
    // every segment of <= SEG_SHORT entries was finished by stage A (segsum_tiles_k): pick its sum up.  The loads do not
    // depend on the row, so they share the round trip of the segment bounds.
    float4 gs = F4Z;
    float g1s = 0.f;
    if (load_staged) {
      gs = reinterpret_cast<const float4*>(part.G)[sl * LPR + q];
      g1s = (do1 && part.gw1 != nullptr) ? part.gw1[sl] : 0.f;
    }
    if (own && end - beg <= SEG_SHORT) {
      valid = true;
      staged = true;
      if (load_staged) {
        acc = gs;
        a1 = g1s;
        prefetch(row);
        if (gy2 != nullptr) e = table_row(row);
      }
      return true;
    }
    const bool is_null = own && nrow >= 0 && row == nrow;
    if (is_null) end = beg;
    valid = own && end - beg <= SEG_SHORT;        // long rows belong to their helper wave
    if (!valid) return true;
    SegCtx<LPR> c;
    c.xb = xb;
    c.S4 = reinterpret_cast<const float4*>(S);
    c.X4 = reinterpret_cast<const float4*>(dX);
    c.gy1 = gy1;
    c.gy2 = gy2;
    c.pf = perm + (size_t)f * stride;
    c.F = F;
    c.f = f;
    c.q = q;
    c.do1 = do1;
    prefetch(row);
    if (gy2 != nullptr) e = table_row(row);
    seg_range_sum<LPR>(c, e, beg, end, acc, a1);
    return true;
  }
  const size_t nch = (size_t)(B + SEG_CHUNK - 1) / SEG_CHUNK;
  const int32_t* ll = part.long_list(F, stride) + (size_t)f * nch;
  const int nlw = (nlong + GPW - 1) / GPW;
  const float4* P4 = reinterpret_cast<const float4*>(part.P);
  if (wf < nact + nlw) {
    // long segments (SEG_SHORT < entries <= SEG_HUGE): one GROUP each, <= 17 chunk partials in ascending order
    const int k = (wf - nact) * GPW + g;
    valid = k < nlong;
    if (!valid) return true;
    const int j = ll[k];
    const int sb = so[j], se = so[j + 1];
    sl = (size_t)f * stride + j;
    row = uniq_row[sl];
    if (nrow >= 0 && row == nrow) {   // the padding row is written (as zero) by its row-owner group
      valid = false;
      return true;
    }
    prefetch(row);
    if (gy2 != nullptr) e = table_row(row);
    const int c0 = sb / SEG_CHUNK, nt = (se - 1) / SEG_CHUNK - c0 + 1;
    partial_range_sum<LPR>(P4, part.P1, (size_t)f * nch + c0, sb % SEG_CHUNK != 0, q, do1, 0, nt, acc, a1);
    return true;
  }
  // huge segments: a whole wave each; the groups sum contiguous sub-ranges of the partials, combined in ascending order
  const int k = wf - nact - nlw;
  if (k >= nhuge) return false;
  const int j = ll[nch - 1 - k];
  const int sb = so[j], se = so[j + 1];
  sl = (size_t)f * stride + j;
  row = uniq_row[sl];
  if (nrow >= 0 && row == nrow) return false;   // the padding row is written (as zero) by its row-owner group
  valid = g == 0;
  if (valid) prefetch(row);
  if (gy2 != nullptr && valid) e = table_row(row);
  const int c0 = sb / SEG_CHUNK, nt = (se - 1) / SEG_CHUNK - c0 + 1;
  const int per = (nt + GPW - 1) / GPW;
  const int t0 = g * per;
  const int t1 = t0 + per < nt ? t0 + per : nt;
  float4 p = F4Z;
  float p1 = 0.f;
  if (t0 < t1) partial_range_sum<LPR>(P4, part.P1, (size_t)f * nch + c0, sb % SEG_CHUNK != 0, q, do1, t0, t1, p, p1);
  float4 tot = make_float4(__shfl(p.x, q), __shfl(p.y, q), __shfl(p.z, q), __shfl(p.w, q));
  float t1s = __shfl(p1, 0);
#pragma unroll
  for (int kk = 1; kk < GPW; ++kk) {  // ascending sub-range order
    const int ln = kk * LPR + q;
    tot = f4_add(tot, make_float4(__shfl(p.x, ln), __shfl(p.y, ln), __shfl(p.z, ln), __shfl(p.w, ln)));
    t1s += __shfl(p1, kk * LPR);
  }
  acc = tot;
  a1 = t1s;
  return true;
}

// The per-wave body of the segment-sum: returns false when the wave owns no unique row.  On return, for lanes with
// `valid`: sl = slot index of the row, acc = summed gradient quarter, a1 = summed first-order gradient (q == 0 lanes),
// e = the table row quarter (loaded only when the FM term is active), row = global row.
template <int D, bool PRE>
__device__ __forceinline__ bool segsum_wave(int wave, const float* __restrict__ tables, const float* __restrict__ S,
                                            const float* __restrict__ dX, const float* __restrict__ gy1,
                                            const float* __restrict__ gy2, const int32_t* __restrict__ perm,
                                            const int32_t* __restrict__ seg_off, const int32_t* __restrict__ uniq_row,
                                            const int32_t* __restrict__ nuniq, uint64_t w1_mask, int B, int F, int stride,
                                            int null_row, const SegPartials& part, const ExBlocks& xb, bool& valid, size_t& sl,
                                            float4& acc, float& a1, float4& e, int& row, bool& do1, bool& staged,
                                            const bool load_staged, AdamRowPrefetch& pre) {
  staged = false;
  constexpr int LPR = D / 4;
  constexpr int GPW = RSX_WAVE / LPR;
  const int lane = threadIdx.x & 63;
  const int q = lane % LPR, g = lane / LPR;
  const int wpf = (B + GPW - 1) / GPW;  // waves per field: a wave never straddles two fields
  const int f = wave / wpf;
  valid = false;
  if (f >= F) return false;
  const int wf = wave - f * wpf;
  const int nu = nuniq[f];
  // The field's unique rows are dealt EVENLY to its wpf waves: rpw = ceil(nu / wpf) rows per wave (groups 0 .. rpw-1 own
  // them), so a field with a mid-sized vocabulary (say 50 rows, each a long Zipf-skewed segment) keeps 13 waves busy with 4
  // rows each instead of 4 waves with 16 -- the wave-cooperative walk of a long segment is sequential per wave.  rpw = 1
  // (tiny vocabularies: every segment is long): group 0 owns the row and anything above 2 entries is walked by the whole wave.
  const int rpw = nu <= wpf ? 1 : (nu + wpf - 1) / wpf;
  const bool spread = rpw == 1;
  const int j0 = wf * rpw;
  if (j0 >= nu) return false;  // wave-uniform
  const int j = j0 + g;
  valid = g < rpw && j < nu;
  sl = (size_t)f * stride + (valid ? j : j0);
  const int beg = valid ? seg_off[(size_t)f * (stride + 1) + j] : 0;
  int end = valid ? seg_off[(size_t)f * (stride + 1) + j + 1] : 0;
  row = valid ? uniq_row[sl] : 0;
  // null_row: a padding row whose per-entry gradients are exactly zero by construction (DIN history padding id 0,
  // din/din.py:107): its (possibly huge) segment is not walked, G = 0 is written -- the same value the sum would give
  if (valid && null_row >= 0 && row == null_row) end = beg;
  const int L = end - beg;
  SegCtx<LPR> c;
  c.xb = xb;
  c.S4 = reinterpret_cast<const float4*>(S);
  c.X4 = reinterpret_cast<const float4*>(dX);
  c.gy1 = gy1;
  c.gy2 = gy2;
  c.pf = perm + (size_t)f * stride;
  c.F = F;
  c.f = f;
  c.q = q;
  c.do1 = gy1 != nullptr && q == 0 && ((w1_mask >> f) & 1ull);
  do1 = c.do1;
  e = F4Z;
  if (gy2 != nullptr && valid) e = reinterpret_cast<const float4*>(tables)[(size_t)row * LPR + q];
  if constexpr (PRE) {
    if (valid) {
      const size_t o = (size_t)row * LPR + q;
      pre.var = gy2 != nullptr ? e : reinterpret_cast<const float4*>(pre.tables)[o];
      pre.m = reinterpret_cast<const float4*>(pre.m_t)[o];
      pre.v = reinterpret_cast<const float4*>(pre.v_t)[o];
      const size_t wi = (size_t)row * pre.w1_stride;
      pre.w = pre.w1[wi];
      pre.mw = pre.m_w[wi];
      pre.vw = pre.v_w[wi];
    }
  }
  acc = F4Z;
  a1 = 0.f;
  const int short_len = spread ? 2 : SEG_SHORT;   // a spread wave has 15 idle groups: cooperate on anything > 2
  if (valid && L <= short_len) seg_range_sum<LPR>(c, e, beg, end, acc, a1);
  unsigned long long todo = __ballot(valid && L > short_len && q == 0);
  while (todo) {  // wave-uniform loop over the long segments owned by this wave
    const int src = __ffsll((long long)todo) - 1;  // lane (owner group, q = 0)
    todo &= todo - 1;
    const int sb = __shfl(beg, src), se = __shfl(end, src);
    const float4 es = make_float4(__shfl(e.x, src + q), __shfl(e.y, src + q), __shfl(e.z, src + q), __shfl(e.w, src + q));
    const int per = (se - sb + GPW - 1) / GPW;
    const int r0 = sb + g * per;
    const int r1 = r0 + per < se ? r0 + per : se;
    float4 p = F4Z;
    float p1 = 0.f;
    if (r0 < r1) seg_range_sum<LPR>(c, es, r0, r1, p, p1);
    float4 tot = make_float4(__shfl(p.x, q), __shfl(p.y, q), __shfl(p.z, q), __shfl(p.w, q));
    float t1 = __shfl(p1, 0);
#pragma unroll
    for (int k = 1; k < GPW; ++k) {  // ascending sub-range order
      const int ln = k * LPR + q;
      tot = f4_add(tot, make_float4(__shfl(p.x, ln), __shfl(p.y, ln), __shfl(p.z, ln), __shfl(p.w, ln)));
      t1 += __shfl(p1, k * LPR);
    }
    if (g == src / LPR) {
      acc = tot;
      a1 = t1;
    }
  }
  return true;
}

// Stage A of the two-stage scatter (B > 1024): TILES of the sorted order.  A 256-thread workgroup takes 256/LPR * 16
// consecutive sorted positions of one field (1024 at D = 16), its LPR-lane group g the 16-position chunk g.  The index
// side -- segment index and example index of every position of the tile, plus 16 positions of halo on either side --
// arrives with ONE coalesced round trip into LDS; from it a group knows, without touching seg_off, where the segments
// of its chunk begin and end (a count of equal neighbours, capped at 16 = "long"), so the 16 gradient rows of the chunk
// are requested at once, one round trip later.  Then, in ascending position (the oracle's order):
//   * every segment of <= SEG_SHORT entries is FINISHED here by the group of the chunk it starts in -- a segment that
//     crosses into the next chunk is followed there (at most 15 more entries) -- and its sum written to ws.G / ws.gw1;
//     stage B never walks the permutation or a gradient row again;
//   * the chunk's overlap with a LONG segment goes to the partial slots (slot 0: the segment of the chunk's first
//     position, slot 1: of its last); stage B adds them in ascending chunk order.
// The previous form (one group = one chunk with its own 34 scalar index loads, then seg_off / uniq_row, then the rows in
// two batches: 5-6 dependent round trips, short segments crossing a chunk boundary left to stage B) measured 12 us at
// 4 096 examples x 39 fields on 156 workgroups -- a pure latency chain.
template <int LPR>
struct SegTile {
  static constexpr int GPB = 256 / LPR;          // groups (= chunks) per workgroup
  static constexpr int POS = GPB * SEG_CHUNK;    // sorted positions per workgroup
  static constexpr int SID = POS + 2 * SEG_CHUNK;
  static constexpr int PRM = POS + SEG_CHUNK;
  static constexpr size_t lds_bytes = (size_t)(SID + PRM) * sizeof(int32_t);
};

// W1: the first-order sums ride along (compile-time, so that the gy1 loads are as unconditional as the row loads: behind a
// per-lane condition the first of them is waited for alone, ahead of the whole batch)
template <int D, bool FM, bool W1>
__device__ __forceinline__ void segsum_tiles_body(const float* __restrict__ tables, const float* __restrict__ S,
                                                  const float* __restrict__ dX, const float* __restrict__ gy1,
                                                  const float* __restrict__ gy2, const int32_t* __restrict__ perm,
                                                  const int32_t* __restrict__ uniq_row, const SegPartials& ws,
                                                  uint64_t w1_mask, int B, int F, int stride, int null_row,
                                                  const ExBlocks& xb) {
  constexpr int LPR = D / 4;
  using T = SegTile<LPR>;
  extern __shared__ __attribute__((aligned(16))) int32_t tile_lds[];
  int32_t* sid_l = tile_lds;                       // [SID]: segment index of positions base - 16 ..; -1 before 0, -2 from B on
  int32_t* prm_l = tile_lds + T::SID;              // [PRM]: example index of positions base ..
  const int tid = threadIdx.x;
  const int tiles = (B + T::POS - 1) / T::POS;
  const int f = blockIdx.x / tiles;
  if ((ws.skip >> f) & 1ull) return;               // (workgroup-uniform) a field the sort skipped (rsx_sort_job.skip_mask)
  const int base = (blockIdx.x - f * tiles) * T::POS;
  const int32_t* pf = perm + (size_t)f * stride;
  const int32_t* sid = ws.segid + (size_t)f * stride;
  RSX_STAMP2(0);
  // the padding row of this field (exactly-zero gradients by construction: never walked), or -1
  const int nrow = ws.null_of(f, null_row);
  {
    constexpr int NI = (T::SID + 255) / 256;
    int32_t a[NI], b[NI];
#pragma unroll
    for (int u = 0; u < NI; ++u) {                 // every load of the tile before the first LDS store
      const int i = tid + 256 * u, pos = base - SEG_CHUNK + i;
      a[u] = sid[pos < 0 ? 0 : (pos < B ? pos : B - 1)];
      const int pp = base + i;
      b[u] = pf[pp < B ? pp : B - 1];
    }
#pragma unroll
    for (int u = 0; u < NI; ++u) {
      const int i = tid + 256 * u, pos = base - SEG_CHUNK + i;
      if (i < T::SID) sid_l[i] = pos < 0 ? -1 : (pos < B ? a[u] : -2);
      if (i < T::PRM) prm_l[i] = b[u];
    }
  }
  __syncthreads();
  RSX_STAMP2(1);
  const int g = tid / LPR, q = tid % LPR;
  const int p0 = base + g * SEG_CHUNK;
  if (p0 >= B) return;
  const int n_in = B - p0 < SEG_CHUNK ? B - p0 : SEG_CHUNK;
  // the chunk's window of segment indices: 16 before, the chunk, 16 after
  int sw[3 * SEG_CHUNK];
  {
    const int4* w4 = reinterpret_cast<const int4*>(sid_l + g * SEG_CHUNK);
#pragma unroll
    for (int u = 0; u < 3 * SEG_CHUNK / 4; ++u) {
      const int4 v = w4[u];
      sw[4 * u] = v.x; sw[4 * u + 1] = v.y; sw[4 * u + 2] = v.z; sw[4 * u + 3] = v.w;
    }
  }
  int pb[SEG_CHUNK];
  {
    const int4* p4 = reinterpret_cast<const int4*>(prm_l + g * SEG_CHUNK);
#pragma unroll
    for (int u = 0; u < SEG_CHUNK / 4; ++u) {
      const int4 v = p4[u];
      pb[4 * u] = v.x; pb[4 * u + 1] = v.y; pb[4 * u + 2] = v.z; pb[4 * u + 3] = v.w;
    }
  }
  const int jf = sw[SEG_CHUNK], jl = sid_l[(g + 1) * SEG_CHUNK + n_in - 1];
  uint32_t mb = 0, ma = 0, mf = 0, ml = 0;
#pragma unroll
  for (int k = 0; k < SEG_CHUNK; ++k) {
    mb |= (uint32_t)(sw[SEG_CHUNK - 1 - k] == jf) << k;
    ma |= (uint32_t)(sw[2 * SEG_CHUNK + k] == jl) << k;
    mf |= (uint32_t)(sw[SEG_CHUNK + k] == jf) << k;
    ml |= (uint32_t)(sw[SEG_CHUNK + k] == jl) << k;
  }
  const int cb = __builtin_ctz(~mb);                               // equal neighbours before the chunk (16: "16 or more")
  const int ca = n_in == SEG_CHUNK ? __builtin_ctz(~ma) : 0;       // ... after it
  const int cf = __builtin_ctz(~mf);                               // entries of the first segment inside the chunk
  const bool one = jf == jl;
  const int cl = one ? 0 : n_in - __builtin_ctz(ml);               // ... of the last one
  // (the rows of the chunk's first / last segment: only looked at inside the sum loop, after the wait for the row loads)
  int rowf = -1, rowl = -1;
  if (nrow >= 0) {
    rowf = uniq_row[(size_t)f * stride + jf];
    rowl = uniq_row[(size_t)f * stride + jl];
  }
  bool long0, own0, long1 = false, own1 = false;
  int next;                                                        // entries of the last owned segment beyond the chunk
  if (one) {
    long0 = cb == SEG_CHUNK || ca == SEG_CHUNK || cb + n_in + ca > SEG_SHORT;
    own0 = !long0 && cb == 0;
    next = own0 ? ca : 0;
  } else {
    long0 = cb == SEG_CHUNK || cb + cf > SEG_SHORT;
    own0 = !long0 && cb == 0;
    long1 = ca == SEG_CHUNK || cl + ca > SEG_SHORT;
    own1 = !long1;
    next = own1 ? ca : 0;
  }
  // dX may be absent with the FM term alone (fm.py): the row loads then read S instead and are multiplied by 0 -- a load
  // behind even a uniform condition is waited for on its own, one position after the other
  const bool has_x = dX != nullptr;
  const float xm = has_x ? 1.f : 0.f;
  const size_t xrow = has_x ? (size_t)F * LPR : (size_t)LPR, xoff = has_x ? (size_t)f * LPR + q : (size_t)q;
  const bool do1 = W1 && q == 0 && ((w1_mask >> f) & 1ull);
  const float4* T4 = reinterpret_cast<const float4*>(tables);
  const float4* S4 = reinterpret_cast<const float4*>(S);
  const float4* X4 = reinterpret_cast<const float4*>(has_x ? dX : S);
  float4* P4 = reinterpret_cast<float4*>(ws.P);
  float4* G4 = reinterpret_cast<float4*>(ws.G);
  const size_t fs = (size_t)f * stride;
  const size_t po = ((size_t)f * ((B + SEG_CHUNK - 1) / SEG_CHUNK) + (size_t)(p0 / SEG_CHUNK)) * 2;
  constexpr int NB = FM ? 8 : 16;                                  // entries whose loads are in flight together
  RSX_STAMP2(2);
  float4 acc = F4Z;
  float a1 = 0.f;
  // The first XE entries of the owned short segment that runs on into the next chunk are requested NOW, with the chunk's
  // own rows: a load issued after the sums below have been stored could only be waited for together with those stores
  // (loads and stores share one in-order counter), and most such tails are 1-3 entries long.
  constexpr int XE = 4;
  float xg[XE], xh[XE];
  float4 xs[XE], xx0[XE];
  const int rowx = FM ? uniq_row[fs + jl] : 0;                     // (unconditional: a guarded load is waited for alone)
#pragma unroll
  for (int k = 0; k < XE; ++k) {
    int bi;
    size_t bo;
    ex_locate(xb, prm_l[(g + 1) * SEG_CHUNK + (k < next ? k : 0)], bi, bo);
    xg[k] = FM ? gy2[bo + bi] : 0.f;
    xs[k] = FM ? S4[bo / 4 + (size_t)bi * LPR + q] : F4Z;
    xx0[k] = X4[bo / 4 + (size_t)bi * xrow + xoff];
    xh[k] = W1 ? gy1[bo + bi] : 0.f;
  }
  auto to_G = [&](const int j) {
    G4[(fs + j) * LPR + q] = acc;
    if (ws.gw1 != nullptr && q == 0) ws.gw1[fs + j] = do1 ? a1 : 0.f;
  };
  auto to_P = [&](const int slot) {
    P4[(po + slot) * LPR + q] = acc;
    if (do1) ws.P1[po + slot] = a1;
  };
#pragma unroll
  for (int k0 = 0; k0 < SEG_CHUNK; k0 += NB) {
    float gg[NB], hh[NB];
    float4 ss[NB], xx[NB], ee[NB];
    int rw[NB];
#pragma unroll
    for (int k = 0; k < NB; ++k) {
      int bi;
      size_t bo;
      ex_locate(xb, pb[k0 + k], bi, bo);                           // (positions past B hold the last example: valid, unused)
      rw[k] = FM ? uniq_row[fs + (sw[SEG_CHUNK + k0 + k] >= 0 ? sw[SEG_CHUNK + k0 + k] : 0)] : 0;
      gg[k] = FM ? gy2[bo + bi] : 0.f;
      ss[k] = FM ? S4[bo / 4 + (size_t)bi * LPR + q] : F4Z;
      xx[k] = X4[bo / 4 + (size_t)bi * xrow + xoff];
      hh[k] = W1 ? gy1[bo + bi] : 0.f;
    }
#pragma unroll
    for (int k = 0; k < NB; ++k) ee[k] = FM ? T4[(size_t)rw[k] * LPR + q] : F4Z;
    // every load above has landed before the first sum is stored: after a store, the compiler's per-block waits for a
    // loaded value become "wait for everything", i.e. for the stores (measured: 5.6 us for the 16 positions of a chunk)
    __builtin_amdgcn_s_waitcnt(0x0F70);                            // vmcnt(0)
    RSX_STAMP2(3 + (k0 ? 3 : 0));
    const bool nullf = nrow >= 0 && rowf == nrow, nulll = nrow >= 0 && rowl == nrow;   // a padding segment: never walked,
#pragma unroll
    for (int k = 0; k < NB; ++k) {
      const int kp = k0 + k;
      if (kp < n_in) {                                              // (group-uniform over the LPR lanes of a row)
        const int j = sw[SEG_CHUNK + kp];
        const bool isf = j == jf, isl = !one && j == jl;
        const bool use = isf ? ((long0 || own0) && !nullf) : (isl ? !nulll : true);
        const bool head = kp == 0 || j != sw[SEG_CHUNK + kp - 1];
        const bool tail = kp == n_in - 1 || j != sw[SEG_CHUNK + kp + 1];
        if (head) {
          acc = F4Z;
          a1 = 0.f;
        }
        if (use) {
          float4 t = F4Z;
          if (FM) t = f4_sub(f4_scale(gg[k], ss[k]), f4_scale(gg[k], ee[k]));
          t = FM ? f4_add(t, f4_scale(xm, xx[k])) : xx[k];
          acc = f4_add(acc, t);
          if (W1) a1 += hh[k];
        }
        if (tail) {
          if (isf) {                                               // no partials (a short one that starts here: its zero)
            if (long0) { if (!nullf) to_P(0); }
            else if (own0 && !(one && next > 0)) to_G(j);
          } else if (isl) {
            if (long1) { if (!nulll) to_P(1); }
            else if (own1 && next == 0) to_G(j);
          } else {
            to_G(j);
          }
        }
      }
    }
  }
  RSX_STAMP2(4);
  // the owned short segment that runs on into the next chunk: its remaining <= 15 entries, in order
  const float4 ex = FM ? T4[(size_t)rowx * LPR + q] : F4Z;
  const bool nullx = nrow >= 0 && (one ? rowf : rowl) == nrow;
#pragma unroll
  for (int k = 0; k < XE; ++k) {
    if (k < next && !nullx) {
      float4 t = F4Z;
      if (FM) t = f4_sub(f4_scale(xg[k], xs[k]), f4_scale(xg[k], ex));
      t = FM ? f4_add(t, f4_scale(xm, xx0[k])) : xx0[k];
      acc = f4_add(acc, t);
      if (W1) a1 += xh[k];
    }
  }
  for (int k0 = XE; __ballot(k0 < next) != 0ull; k0 += 4) {
    float gg[4], hh[4];
    float4 ss[4], xx[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int kk = k0 + k < next ? k0 + k : 0;
      int bi;
      size_t bo;
      ex_locate(xb, prm_l[(g + 1) * SEG_CHUNK + kk], bi, bo);
      gg[k] = FM ? gy2[bo + bi] : 0.f;
      ss[k] = FM ? S4[bo / 4 + (size_t)bi * LPR + q] : F4Z;
      xx[k] = X4[bo / 4 + (size_t)bi * xrow + xoff];
      hh[k] = W1 ? gy1[bo + bi] : 0.f;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (k0 + k < next && !nullx) {
        float4 t = F4Z;
        if (FM) t = f4_sub(f4_scale(gg[k], ss[k]), f4_scale(gg[k], ex));
        t = FM ? f4_add(t, f4_scale(xm, xx[k])) : xx[k];
        acc = f4_add(acc, t);
        if (W1) a1 += hh[k];
      }
    }
  }
  if (next > 0) to_G(jl);
  RSX_STAMP2(5);
}

template <int D, bool FM, bool W1>
__global__ __launch_bounds__(256) void segsum_tiles_k(const float* __restrict__ tables, const float* __restrict__ S,
                                                      const float* __restrict__ dX, const float* __restrict__ gy1,
                                                      const float* __restrict__ gy2, const int32_t* __restrict__ perm,
                                                      const int32_t* __restrict__ uniq_row, const SegPartials ws,
                                                      uint64_t w1_mask, int B, int F, int stride, int null_row,
                                                      const ExBlocks xb) {
  segsum_tiles_body<D, FM, W1>(tables, S, dX, gy1, gy2, perm, uniq_row, ws, w1_mask, B, F, stride, null_row, xb);
}
// the same launch carrying reductions of the step's other launches that only the optimizer reads (step_riders_device.h): its
// own n_own position-tile workgroups, then the tower's dW partial-tile reductions, then the cross layers' gradient reduce
template <int D, bool FM, bool W1>
__global__ __launch_bounds__(256) void segsum_tiles_ride_k(const float* __restrict__ tables, const float* __restrict__ S,
                                                           const float* __restrict__ dX, const float* __restrict__ gy1,
                                                           const float* __restrict__ gy2, const int32_t* __restrict__ perm,
                                                           const int32_t* __restrict__ uniq_row, const SegPartials ws,
                                                           uint64_t w1_mask, int B, int F, int stride, int null_row,
                                                           const ExBlocks xb, const DwReduceJobs dwj, const rsx_cross_reduce_job cr,
                                                           const rsx_vec_reduce_job v0, const rsx_vec_reduce_job v1,
                                                           const uint32_t n_own, const uint32_t n_dw, const uint32_t n_cross,
                                                           const uint32_t n_v0) {
  if (blockIdx.x >= n_own) {
    extern __shared__ __attribute__((aligned(16))) int32_t tile_lds[];      // (>= 4 KB: the host sizes the launch for the riders)
    const uint32_t r = blockIdx.x - n_own;
    if (r < n_dw) {
      RSX_DW_REDUCE_SELECT(dwj, r, jb, blk)
      dw_reduce_job_block(jb, blk);
    } else if (r < n_dw + n_cross) {
      cross_reduce_block(cr.part, cr.RT, cr.n, cr.dW, cr.dB, cr.dwout, cr.L, cr.dim, r - n_dw, reinterpret_cast<float*>(tile_lds));
    } else if (r < n_dw + n_cross + n_v0) {
      vec_reduce_block(v0.part, v0.G, v0.n, v0.out, r - n_dw - n_cross, reinterpret_cast<float*>(tile_lds));
    } else {
      vec_reduce_block(v1.part, v1.G, v1.n, v1.out, r - n_dw - n_cross - n_v0, reinterpret_cast<float*>(tile_lds));
    }
    return;
  }
  segsum_tiles_body<D, FM, W1>(tables, S, dX, gy1, gy2, perm, uniq_row, ws, w1_mask, B, F, stride, null_row, xb);
}

// ---- single stage through LDS (B <= SEG_LDS_MAX_B entries per field, 4 | waves per field) ---------------------------------
// segsum_wave is a chain of dependent global accesses: field count -> segment bounds + row -> permutation -> the entries'
// rows (-> further batches of 4 for longer segments): 4-5 round trips of 0.3-1 us each in a launch whose other work is
// nothing.  Here the WORKGROUP (4 consecutive waves of one field) does it in two: (1) the field's index arrays -- segment
// offsets, unique rows, permutation: 3 B ints -- go to LDS whole, needing no count; (2) with them, every address is known:
// the rows' own loads (table row, Adam state) and ALL entries of the workgroup's position range [plo, phi) are requested at
// once, the entries staged in LDS; the sums then walk LDS.  Same association as segsum_wave, entry for entry (short segments
// sequentially in ascending position, long ones in 16 sub-ranges whose partials are added in ascending order): bit-identical.
constexpr int SEG_LDS_MAX_B = 512;
__host__ __device__ inline size_t seg_lds_idx4(int B) { return ((size_t)3 * B + 4 + 3) / 4; }      // index arrays, float4 units
__host__ __device__ inline size_t seg_lds_bytes(int B, int D) {
  return (seg_lds_idx4(B) * 4 + (size_t)2 * B * D + (size_t)2 * B) * 4;
}
static inline bool seg_lds_ok(int B, int D, bool two_stage) {
  const int gpw = 64 / (D / 4), wpf = (B + gpw - 1) / gpw;
  static const int env = getenv("RSX_SEG_LDS") ? atoi(getenv("RSX_SEG_LDS")) : 1;      // (0: the per-wave form, A/B runs)
  return env != 0 && !two_stage && B <= SEG_LDS_MAX_B && wpf % 4 == 0 && seg_lds_bytes(B, D) <= 64 * 1024;
}

template <int LPR, bool FM, bool XG>
__device__ __forceinline__ void lds_range_sum(const float4* __restrict__ Xs, const float4* __restrict__ Ss,
                                              const float* __restrict__ g2s, const float* __restrict__ g1s, int plo, int q,
                                              bool do1, const float4 e, int i0, int i1, float4& acc, float& a1) {
  for (int i = i0; i < i1; ++i) {
    const int p = i - plo;
    float4 t;
    if constexpr (FM) {
      const float g = g2s[p];
      t = f4_sub(f4_scale(g, Ss[(size_t)p * LPR + q]), f4_scale(g, e));
      if constexpr (XG) t = f4_add(t, Xs[(size_t)p * LPR + q]);
    } else {
      t = Xs[(size_t)p * LPR + q];
    }
    acc = f4_add(acc, t);
    if (do1) a1 += g1s[p];
  }
}

template <int D, bool PRE>
__device__ __forceinline__ bool segsum_wave_lds(int wave, const float* __restrict__ tables, const float* __restrict__ S,
                                                const float* __restrict__ dX, const float* __restrict__ gy1,
                                                const float* __restrict__ gy2, const int32_t* __restrict__ perm,
                                                const int32_t* __restrict__ seg_off, const int32_t* __restrict__ uniq_row,
                                                const int32_t* __restrict__ nuniq, uint64_t w1_mask, int B, int F, int stride,
                                                int null_row, const ExBlocks& xb, bool& valid, size_t& sl, float4& acc,
                                                float& a1, float4& e, int& row, bool& do1, bool& staged, AdamRowPrefetch& pre,
                                                float4* __restrict__ lds) {
  staged = false;
  valid = false;
  constexpr int LPR = D / 4;
  constexpr int GPW = RSX_WAVE / LPR;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int q = lane % LPR, g = lane / LPR;
  const int wpf = (B + GPW - 1) / GPW;          // a multiple of 4: the workgroup's 4 waves belong to ONE field
  const int f = wave / wpf;
  if (f >= F) return false;                     // workgroup-uniform
  const int wf = wave - f * wpf;
  const int wf0 = wf - w;
  int32_t* __restrict__ so_l = reinterpret_cast<int32_t*>(lds);
  int32_t* __restrict__ ur_l = so_l + (B + 1);
  int32_t* __restrict__ pf_l = ur_l + B;
  float4* __restrict__ Xs = lds + seg_lds_idx4(B);
  float4* __restrict__ Ss = Xs + (size_t)B * LPR;
  float* __restrict__ g2s = reinterpret_cast<float*>(Ss + (size_t)B * LPR);
  float* __restrict__ g1s = g2s + B;
  // (1) the field's index arrays, whole (entries past the field's counts are stale and never used)
  {
    const int32_t* __restrict__ so = seg_off + (size_t)f * (stride + 1);
    const int32_t* __restrict__ ur = uniq_row + (size_t)f * stride;
    const int32_t* __restrict__ pf = perm + (size_t)f * stride;
    int a[SEG_LDS_MAX_B / 256 + 1], b[SEG_LDS_MAX_B / 256], c[SEG_LDS_MAX_B / 256];
#pragma unroll
    for (int k = 0; k <= SEG_LDS_MAX_B / 256; ++k) a[k] = so[min(k * 256 + tid, B)];
#pragma unroll
    for (int k = 0; k < SEG_LDS_MAX_B / 256; ++k) {
      b[k] = ur[min(k * 256 + tid, B - 1)];
      c[k] = pf[min(k * 256 + tid, B - 1)];
    }
#pragma unroll
    for (int k = 0; k <= SEG_LDS_MAX_B / 256; ++k)
      if (k * 256 + tid <= B) so_l[k * 256 + tid] = a[k];
#pragma unroll
    for (int k = 0; k < SEG_LDS_MAX_B / 256; ++k)
      if (k * 256 + tid < B) {
        ur_l[k * 256 + tid] = b[k];
        pf_l[k * 256 + tid] = c[k];
      }
  }
  const int nu = nuniq[f];
  __syncthreads();
  const int rpw = nu <= wpf ? 1 : (nu + wpf - 1) / wpf;      // (the dealing of segsum_wave)
  const bool spread = rpw == 1;
  const int jlo = wf0 * rpw;
  if (jlo >= nu) return false;                  // workgroup-uniform: none of its waves owns a row
  const int jhi = (wf0 + 4) * rpw < nu ? (wf0 + 4) * rpw : nu;
  const int plo = so_l[jlo], phi = so_l[jhi];
  const int j0 = wf * rpw;
  const int j = j0 + g;
  valid = g < rpw && j < nu;
  sl = (size_t)f * stride + (valid ? j : jlo);
  const int beg = valid ? so_l[j] : 0;
  int end = valid ? so_l[j + 1] : 0;
  row = valid ? ur_l[j] : 0;
  if (valid && null_row >= 0 && row == null_row) end = beg;
  const int L = end - beg;
  do1 = gy1 != nullptr && q == 0 && ((w1_mask >> f) & 1ull);
  // (2) the rows' own loads ...
  e = F4Z;
  if (gy2 != nullptr) e = reinterpret_cast<const float4*>(tables)[(size_t)row * LPR + q];
  if constexpr (PRE) {
    const size_t o = (size_t)row * LPR + q;
    pre.var = gy2 != nullptr ? e : reinterpret_cast<const float4*>(pre.tables)[o];
    pre.m = reinterpret_cast<const float4*>(pre.m_t)[o];
    pre.v = reinterpret_cast<const float4*>(pre.v_t)[o];
    const size_t wi = (size_t)row * pre.w1_stride;
    pre.w = pre.w1[wi];
    pre.mw = pre.m_w[wi];
    pre.vw = pre.v_w[wi];
  }
  // ... and every entry of the position range, 4 per thread in flight.  Absent inputs are read through a pointer to a
  // present one (loads stay unconditional; the values are never used).
  {
    const float4* __restrict__ X4 = reinterpret_cast<const float4*>(dX != nullptr ? dX : S);
    const float4* __restrict__ S4 = reinterpret_cast<const float4*>(gy2 != nullptr ? S : dX);
    const float* __restrict__ h2 = gy2 != nullptr ? gy2 : dX;
    const float* __restrict__ h1 = gy1 != nullptr ? gy1 : h2;
    const int xF = dX != nullptr ? F : 1, xf = dX != nullptr ? f : 0;     // (S rows have no field index)
    const int ntask = (phi - plo) * LPR;
    for (int base = 0; base < ntask; base += 1024) {
      float4 xv[4], sv[4];
      float gv[4], hv[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int t = base + k * 256 + tid;
        const int tc = t < ntask ? t : ntask - 1;
        const int p = tc / LPR, qq = tc % LPR;
        int bi;
        size_t bo;
        ex_locate(xb, pf_l[plo + p], bi, bo);
        xv[k] = X4[bo / 4 + ((size_t)bi * xF + xf) * LPR + qq];
        sv[k] = S4[bo / 4 + (gy2 != nullptr ? (size_t)bi * LPR : ((size_t)bi * xF + xf) * LPR) + qq];
        gv[k] = h2[bo + bi];
        hv[k] = h1[bo + bi];
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int t = base + k * 256 + tid;
        if (t < ntask) {
          Xs[t] = xv[k];
          Ss[t] = sv[k];
          if (t % LPR == 0) {
            g2s[t / LPR] = gv[k];
            g1s[t / LPR] = hv[k];
          }
        }
      }
    }
  }
  __syncthreads();
  if (j0 >= nu) return false;                   // (a wave without rows leaves after the barriers)
  acc = F4Z;
  a1 = 0.f;
  const bool fm = gy2 != nullptr, xg = dX != nullptr;
  auto range = [&](const float4 es, int i0, int i1, float4& ac, float& a) {
    if (fm) {
      if (xg) lds_range_sum<LPR, true, true>(Xs, Ss, g2s, g1s, plo, q, do1, es, i0, i1, ac, a);
      else lds_range_sum<LPR, true, false>(Xs, Ss, g2s, g1s, plo, q, do1, es, i0, i1, ac, a);
    } else {
      lds_range_sum<LPR, false, true>(Xs, Ss, g2s, g1s, plo, q, do1, es, i0, i1, ac, a);
    }
  };
  const int short_len = spread ? 2 : SEG_SHORT;
  if (valid && L <= short_len) range(e, beg, end, acc, a1);
  unsigned long long todo = __ballot(valid && L > short_len && q == 0);
  while (todo) {  // wave-uniform loop over the long segments owned by this wave (as in segsum_wave)
    const int src = __ffsll((long long)todo) - 1;
    todo &= todo - 1;
    const int sb = __shfl(beg, src), se = __shfl(end, src);
    const float4 es = make_float4(__shfl(e.x, src + q), __shfl(e.y, src + q), __shfl(e.z, src + q), __shfl(e.w, src + q));
    const int per = (se - sb + GPW - 1) / GPW;
    const int r0 = sb + g * per;
    const int r1 = r0 + per < se ? r0 + per : se;
    float4 p = F4Z;
    float p1 = 0.f;
    if (r0 < r1) range(es, r0, r1, p, p1);
    float4 tot = make_float4(__shfl(p.x, q), __shfl(p.y, q), __shfl(p.z, q), __shfl(p.w, q));
    float t1 = __shfl(p1, 0);
#pragma unroll
    for (int k = 1; k < GPW; ++k) {  // ascending sub-range order
      const int ln = k * LPR + q;
      tot = f4_add(tot, make_float4(__shfl(p.x, ln), __shfl(p.y, ln), __shfl(p.z, ln), __shfl(p.w, ln)));
      t1 += __shfl(p1, k * LPR);
    }
    if (g == src / LPR) {
      acc = tot;
      a1 = t1;
    }
  }
  return true;
}

template <int D>
__global__ __launch_bounds__(256) void segsum_bwd_k(const float* __restrict__ tables, const float* __restrict__ S,
                                                    const float* __restrict__ dX, const float* __restrict__ gy1,
                                                    const float* __restrict__ gy2, const int32_t* __restrict__ perm,
                                                    const int32_t* __restrict__ seg_off,
                                                    const int32_t* __restrict__ uniq_row,
                                                    const int32_t* __restrict__ nuniq, float* __restrict__ G,
                                                    float* __restrict__ gw1, uint64_t w1_mask, int B, int F,
                                                    int stride, int null_row, const SegPartials part,
                                                    const ExBlocks xb, const int32_t* __restrict__ goff) {
  constexpr int LPR = D / 4;
  const int q = (threadIdx.x & 63) % LPR;
  bool valid, do1;
  size_t sl;
  float4 acc, e;
  float a1;
  int row;
  bool staged;
  // rows finished by stage A already sit in G / gw1 when stage A was given these buffers: nothing to load or store
  const bool same = part.G == G;
  // goff (nullable, [F + 1]; rsx_segsum_bwd_packed): unique row j of field f is written at row goff[f] + j of G / gw1 instead
  // of f * stride + j -- the per-rank block of the data-parallel unique-list exchange, packed to cap_f = goff[f+1] - goff[f]
  // rows per field (min(batch, rows of the field): nuniq[f] can never exceed it)
  auto out_of = [&](const size_t s_l) -> size_t {
    if (goff == nullptr) return s_l;
    const int f = (int)(s_l / (size_t)stride);
    return (size_t)goff[f] + (s_l - (size_t)f * stride);
  };
  AdamRowPrefetch nopre;
  if (part.P != nullptr) {     // two-stage: the compact unit list, grid stride
    constexpr int GPW = RSX_WAVE / LPR;
    const SegUnits su = seg_units<GPW>(nuniq, part, F, stride);
    const int nwaves = (gridDim.x * blockDim.x) >> 6;
    for (int unit = (blockIdx.x * blockDim.x + threadIdx.x) >> 6; unit < su.total; unit += nwaves) {
      int f, wf, nu, nl, nh;
      seg_unit_locate(su, unit, f, wf, nu, nl, nh);
      if (segsum_wave2<D, false>(f, wf, nu, nl, nh, tables, S, dX, gy1, gy2, perm, seg_off, uniq_row, nuniq, w1_mask, B, F,
                                 stride, null_row, part, xb, valid, sl, acc, a1, e, row, do1, staged, !same, nopre) &&
          valid && !(staged && same)) {
        const size_t so = out_of(sl);
        reinterpret_cast<float4*>(G)[so * LPR + q] = acc;
        if (gw1 != nullptr && q == 0) gw1[so] = do1 ? a1 : 0.f;
      }
    }
    return;
  }
  if (part.lds_mode) {
    extern __shared__ float4 seg_dyn_lds[];
    if (!segsum_wave_lds<D, false>((blockIdx.x * blockDim.x + threadIdx.x) >> 6, tables, S, dX, gy1, gy2, perm, seg_off,
                                   uniq_row, nuniq, w1_mask, B, F, stride, null_row, xb, valid, sl, acc, a1, e, row, do1, staged,
                                   nopre, seg_dyn_lds))
      return;
  } else if (!segsum_wave<D, false>((blockIdx.x * blockDim.x + threadIdx.x) >> 6, tables, S, dX, gy1, gy2, perm, seg_off,
                                    uniq_row, nuniq, w1_mask, B, F, stride, null_row, part, xb, valid, sl, acc, a1, e, row, do1,
                                    staged, !same, nopre))
    return;
  if (valid && !(staged && same)) {
    const size_t so = out_of(sl);
    reinterpret_cast<float4*>(G)[so * LPR + q] = acc;
    if (gw1 != nullptr && q == 0) gw1[so] = do1 ? a1 : 0.f;
  }
}

template <int D, int NR>
__global__ __launch_bounds__(256) void segsum_adam_k(const float* __restrict__ S, const float* __restrict__ dX,
                                                     const float* __restrict__ gy1, const float* __restrict__ gy2,
                                                     const int32_t* __restrict__ perm, const int32_t* __restrict__ seg_off,
                                                     const int32_t* __restrict__ uniq_row,
                                                     const int32_t* __restrict__ nuniq, uint64_t w1_mask, int B, int F,
                                                     int stride, const HotAdam h, const SegPartials part,
                                                     const ExBlocks xb) {
  constexpr int LPR = D / 4;
  RSX_STAMP3(0);
  RSX_STAMP(58, blockIdx.x == gridDim.x - 1);
  const float b1p = h.state[0], b2p = h.state[1];
  const uint32_t n_rows = h.tables2 != nullptr ? 2u * h.n_own : h.n_own;
  if (blockIdx.x >= n_rows + h.win_blk + h.extra.n_blk) {
    adam_block(h.cold.args, h.cold.blk_lo + (blockIdx.x - n_rows - h.win_blk - h.extra.n_blk));
  } else if (blockIdx.x >= n_rows + h.win_blk) {
    adam_block(h.extra.args, h.extra.blk_lo + (blockIdx.x - n_rows - h.win_blk));
  } else if (blockIdx.x >= n_rows) {      // window pass (window_pass above)
    const uint32_t wb = blockIdx.x - n_rows;
    RSX_STAMP(40, wb == 0);
    RSX_STAMP_MAX(62, true);                                                           // latest ENTRY of a window-pass workgroup
    window_pass<D, NR>(h, wb, b1p, b2p, F, stride);
  } else if (blockIdx.x >= h.n_own) {     // second table set
    const int q = (threadIdx.x & 63) % LPR;
    bool valid, do1;
    size_t sl;
    float4 acc, e;
    float a1;
    int row;
    bool staged;
    AdamRowPrefetch pre{h.tables2, h.m_t2, h.v_t2, h.tables2, h.m_t2, h.v_t2, 0};
    Hp hp;
    hp.b1 = h.b1; hp.b2 = h.b2; hp.omb1 = 1.0f - h.b1; hp.omb2 = 1.0f - h.b2; hp.eps = h.eps;
    hp.alpha = h.lr * sqrtf(1.0f - b2p) / (1.0f - b1p);
    auto update2 = [&]() {
      const size_t o = (size_t)row * LPR + q;
      float4 var = pre.var, m = pre.m, v = pre.v;
      F4_APPLY(adam_sparse1, var, m, v, acc, true, hp);
      reinterpret_cast<float4*>(h.tables2)[o] = var;
      reinterpret_cast<float4*>(h.m_t2)[o] = m;
      reinterpret_cast<float4*>(h.v_t2)[o] = v;
    };
    if (h.part2.P != nullptr) {           // two-stage: the compact unit list, grid stride over this set's workgroups
      constexpr int GPW = RSX_WAVE / LPR;
      const SegUnits su = seg_units<GPW>(nuniq, h.part2, F, stride);
      for (int unit = ((blockIdx.x - h.n_own) * blockDim.x + threadIdx.x) >> 6; unit < su.total; unit += (int)h.n_own * 4) {
        int f, wf, nu, nl, nh;
        seg_unit_locate(su, unit, f, wf, nu, nl, nh);
        if (segsum_wave2<D, true>(f, wf, nu, nl, nh, h.tables2, nullptr, h.dX2, nullptr, nullptr, perm, seg_off, uniq_row, nuniq,
                                  0, B, F, stride, -1, h.part2, xb, valid, sl, acc, a1, e, row, do1, staged, true, pre) &&
            valid)
          update2();
      }
    } else if (part.lds_mode) {
      extern __shared__ float4 seg_dyn_lds[];
      if (segsum_wave_lds<D, true>(((blockIdx.x - h.n_own) * blockDim.x + threadIdx.x) >> 6, h.tables2, nullptr, h.dX2, nullptr,
                                   nullptr, perm, seg_off, uniq_row, nuniq, 0, B, F, stride, -1, xb, valid, sl, acc, a1, e, row,
                                   do1, staged, pre, seg_dyn_lds) &&
          valid)
        update2();
    } else if (segsum_wave<D, true>(((blockIdx.x - h.n_own) * blockDim.x + threadIdx.x) >> 6, h.tables2, nullptr, h.dX2,
                                    nullptr, nullptr, perm, seg_off, uniq_row, nuniq, 0, B, F, stride, -1, h.part2, xb, valid, sl,
                                    acc, a1, e, row, do1, staged, true, pre) &&
               valid) {
      update2();
    }
  } else {
    const int q = (threadIdx.x & 63) % LPR;
    bool valid, do1;
    size_t sl;
    float4 acc, e;
    float a1;
    int row;
    bool staged;
    const bool hw1 = h.w1 != nullptr;
    AdamRowPrefetch pre{h.tables, h.m_t, h.v_t, hw1 ? h.w1 : h.tables, hw1 ? h.m_w : h.m_t, hw1 ? h.v_w : h.v_t,
                        hw1 ? h.w1_stride : 0};
    Hp hp;
    hp.b1 = h.b1; hp.b2 = h.b2; hp.omb1 = 1.0f - h.b1; hp.omb2 = 1.0f - h.b2; hp.eps = h.eps;
    hp.alpha = h.lr * sqrtf(1.0f - b2p) / (1.0f - b1p);
    auto update1 = [&]() {
      const size_t o = (size_t)row * LPR + q;
      float4 var = pre.var, m = pre.m, v = pre.v;
      F4_APPLY(adam_sparse1, var, m, v, acc, true, hp);
      reinterpret_cast<float4*>(h.tables)[o] = var;
      reinterpret_cast<float4*>(h.m_t)[o] = m;
      reinterpret_cast<float4*>(h.v_t)[o] = v;
      if (h.w1 != nullptr && q == 0) {       // (state prefetched with the row's: no round trip of its own)
        const size_t wi = (size_t)row * h.w1_stride;
        float w = pre.w, mw = pre.mw, vw = pre.vw;
        bool st = true;
        if (h.w1_sparse) {
          st = do1;                             // (fields outside the mask: not its rows)
          adam_sparse1(w, mw, vw, a1, true, hp);
        } else {
          adam_dense1(w, mw, vw, do1 ? a1 : 0.f, hp);
        }
        if (st) {
          h.w1[wi] = w;
          h.m_w[wi] = mw;
          h.v_w[wi] = vw;
        }
      }
    };
    if (part.P != nullptr) {              // two-stage: the compact unit list, grid stride over the row workgroups
      constexpr int GPW = RSX_WAVE / LPR;
      const SegUnits su = seg_units<GPW>(nuniq, part, F, stride);
      for (int unit = (blockIdx.x * blockDim.x + threadIdx.x) >> 6; unit < su.total; unit += (int)h.n_own * 4) {
        int f, wf, nu, nl, nh;
        seg_unit_locate(su, unit, f, wf, nu, nl, nh);
        if (segsum_wave2<D, true>(f, wf, nu, nl, nh, h.tables, S, dX, gy1, gy2, perm, seg_off, uniq_row, nuniq, w1_mask, B, F,
                                  stride, -1, part, xb, valid, sl, acc, a1, e, row, do1, staged, true, pre) &&
            valid)
          update1();
      }
      RSX_STAMP3(1);
    } else if (part.lds_mode) {
      extern __shared__ float4 seg_dyn_lds[];
      const bool any = segsum_wave_lds<D, true>((blockIdx.x * blockDim.x + threadIdx.x) >> 6, h.tables, S, dX, gy1, gy2, perm,
                                                seg_off, uniq_row, nuniq, w1_mask, B, F, stride, -1, xb, valid, sl, acc, a1, e,
                                                row, do1, staged, pre, seg_dyn_lds);
      RSX_STAMP3(1);
      if (any && valid) update1();
    } else {
      const bool any = segsum_wave<D, true>((blockIdx.x * blockDim.x + threadIdx.x) >> 6, h.tables, S, dX, gy1, gy2, perm,
                                            seg_off, uniq_row, nuniq, w1_mask, B, F, stride, -1, part, xb, valid, sl, acc, a1,
                                            e, row, do1, staged, true, pre);
      RSX_STAMP3(1);
      if (any && valid) update1();
    }
  }
  RSX_STAMP3(2);
  RSX_STAMP(43, blockIdx.x == n_rows);
  RSX_STAMP(56, blockIdx.x == gridDim.x - 1);
  __syncthreads();
  RSX_STAMP3(3);
  if (threadIdx.x == 0 && adam_arrive_last(h.state, h.total_blocks)) {
    // (the lazy window pass of the window's later steps reads this step's step size from here)
    if (h.win_k > 1) h.state[8 + h.win_cur] = h.lr * sqrtf(1.0f - b2p) / (1.0f - b1p);
    if (h.advance) {
      h.state[0] = b1p * h.b1;
      h.state[1] = b2p * h.b2;
      reinterpret_cast<uint32_t*>(h.state)[3] += 1u;
    }
  }
  RSX_STAMP3(4);
  RSX_STAMP(44, blockIdx.x == n_rows);
  RSX_STAMP_MAX(59, blockIdx.x < n_rows);                                            // last exit: row owners
  RSX_STAMP_MAX(60, blockIdx.x >= n_rows && blockIdx.x < n_rows + h.win_blk);        //            window pass
  RSX_STAMP_MAX(61, blockIdx.x >= n_rows + h.win_blk);                               //            dense / cold riders
  RSX_STAMP(57, blockIdx.x == gridDim.x - 1);
}

// ------------------------------------------------------------------ C ABI ----------------------
static inline bool d_ok(int D) { return D == 4 || D == 8 || D == 16 || D == 32 || D == 64; }

#define RSX_DISPATCH_D(D, FN, ...)            \
  switch (D) {                                \
    case 4: FN<4>(__VA_ARGS__); break;        \
    case 8: FN<8>(__VA_ARGS__); break;        \
    case 16: FN<16>(__VA_ARGS__); break;      \
    case 32: FN<32>(__VA_ARGS__); break;      \
    default: FN<64>(__VA_ARGS__); break;      \
  }

template <int D>
static void launch_gather(dim3 grid, dim3 block, hipStream_t st, const float* tables, const float* w1,
                          const int32_t* row_off, const int32_t* ids, float* E, float* S, float* y1, float* y2,
                          uint64_t mask, int B, int F) {
  RSX_COUNT_LAUNCH(); gather_fm_fwd_k<D><<<grid, block, 0, st>>>(tables, w1, row_off, ids, E, S, y1, y2, mask, B, F);
}
template <int D>
static void launch_segsum(dim3 grid, dim3 block, hipStream_t st, const float* tables, const float* S, const float* dX,
                          const float* gy1, const float* gy2, const int32_t* perm, const int32_t* seg_off,
                          const int32_t* uniq_row, const int32_t* nuniq, float* G, float* gw1, uint64_t mask, int B,
                          int F, int stride, int null_row, const SegPartials& part, const ExBlocks& xb,
                          const int32_t* goff) {
  const size_t lds = part.lds_mode ? seg_lds_bytes(B, D) : 0;
  RSX_COUNT_LAUNCH(); segsum_bwd_k<D><<<grid, block, lds, st>>>(tables, S, dX, gy1, gy2, perm, seg_off, uniq_row, nuniq, G, gw1, mask, B, F,
                                            stride, null_row, part, xb, goff);
}
template <int D>
static void launch_tiles(dim3 grid, hipStream_t st, const float* tables, const float* S, const float* dX,
                         const float* gy1, const float* gy2, const int32_t* perm, const int32_t* uniq_row,
                         const SegPartials& ws, uint64_t mask, int B, int F, int stride, int null_row, const ExBlocks& xb) {
  constexpr size_t lds = SegTile<D / 4>::lds_bytes;
#define RSX_TILES(FM, W1) \
  segsum_tiles_k<D, FM, W1><<<grid, dim3(256), lds, st>>>(tables, S, dX, gy1, gy2, perm, uniq_row, ws, mask, B, F, stride, null_row, xb)
  RSX_COUNT_LAUNCH();
  if (gy2 != nullptr) {
    if (gy1 != nullptr) RSX_TILES(true, true); else RSX_TILES(true, false);
  } else {
    if (gy1 != nullptr) RSX_TILES(false, true); else RSX_TILES(false, false);
  }
#undef RSX_TILES
}
// host view of the rank-blocked input layout; nullptr -> contiguous
static inline int ex_blocks(const rsx_example_blocks* h, int B, ExBlocks& out) {
  out = ExBlocks{0, 0};
  if (h == nullptr) return RSX_OK;
  if (h->examples <= 0 || h->stride_floats < 0 || (h->stride_floats & 3) != 0) return RSX_EINVAL;
  if (h->examples >= B) return RSX_OK;                 // one block: contiguous
  out = ExBlocks{h->examples, (long long)h->stride_floats};
  return RSX_OK;
}
// host view of the two-stage workspace; nullptr -> single-stage
static inline int seg_partials(const rsx_seg_partials* h, bool need_p1, SegPartials& out) {
  out = SegPartials{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, RSX_NULL_NONE, 0, 0ull};
  if (h == nullptr) return RSX_OK;
  if (!h->segid || !h->P || (need_p1 && !h->P1)) return RSX_EINVAL;
  if (h->null_row == RSX_NULL_LAST_ROW && !h->row_off) return RSX_EINVAL;
  if (h->null_row < RSX_NULL_LAST_ROW) return RSX_EINVAL;
  out = SegPartials{h->segid, h->P, h->P1, h->G, h->G ? h->gw1 : nullptr, h->row_off, h->null_row, 0, h->skip_mask};
  return RSX_OK;
}

template <int D>
static void launch_gather_rows(dim3 grid, dim3 block, hipStream_t st, const float* tables, const int32_t* row_off,
                               const int32_t* ids, float* E, long long n) {
  RSX_COUNT_LAUNCH(); gather_rows_k<D><<<grid, block, 0, st>>>(tables, row_off, ids, E, n);
}

extern "C" int rsx_gather_fm_fwd(const float* tables, const float* w1, const int32_t* row_off, const int32_t* ids,
                                 float* E, float* S, float* y1, float* y2, uint64_t w1_field_mask, int B, int F,
                                 int D, rsx_stream_t stream) {
  if (B < 0 || F <= 0 || F > 64 || !d_ok(D)) return RSX_EINVAL;
  if (B == 0) return RSX_OK;  // empty batch: nothing to read or write (buffers may be null)
  if (!tables || !row_off || !ids || !E) return RSX_EINVAL;
  if ((y1 != nullptr) != (w1 != nullptr)) return RSX_EINVAL;
  if (y2 != nullptr && S == nullptr) return RSX_EINVAL;
  if (F == 1 && w1 == nullptr && S == nullptr && y2 == nullptr) {   // one table, rows only
    const long long lanes = (long long)B * (D / 4);
    RSX_DISPATCH_D(D, launch_gather_rows, dim3((unsigned)((lanes + 255) / 256)), dim3(256), rsx_s(stream), tables, row_off,
                   ids, E, (long long)B);
    RSX_CHECK_LAUNCH();
    return RSX_OK;
  }
  const int waves = B >= 2048 ? 4 : 1;  // small batches: one wave per workgroup spreads over all 256 CUs
  const dim3 grid((B + waves - 1) / waves), block(64 * waves);
  RSX_DISPATCH_D(D, launch_gather, grid, block, rsx_s(stream), tables, w1, row_off, ids, E, S, y1, y2,
                 w1_field_mask, B, F);
  RSX_CHECK_LAUNCH();
  return RSX_OK;
}

template <int D>
static void launch_gather_sort(dim3 grid, size_t lds, hipStream_t st, const float* tables, const float* w1,
                               const int32_t* row_off, const int32_t* ids, float* E, float* S, float* y1, float* y2,
                               uint64_t mask, int B, int F, int n_gather, const SortArgs& sort) {
  RSX_COUNT_LAUNCH(); gather_fm_sort_k<D><<<grid, dim3(256), lds, st>>>(tables, w1, row_off, ids, E, S, y1, y2, mask, B, F, n_gather, sort);
}

extern "C" int rsx_gather_cross_fwd(const float* tables, const int32_t* row_off, const int32_t* ids, float* E, const float* W,
                                    const float* Bc, const float* wout, float* s, float* cz, int B, int F, int D, int L,
                                    rsx_stream_t stream) {
  if (!tables || !row_off || !ids || !E || !W || !Bc || !s || B < 0 || F <= 0 || L <= 0) return RSX_EINVAL;
  if ((cz != nullptr) != (wout != nullptr)) return RSX_EINVAL;
  if (D != 16 || F > 64 || L > CROSS_MAX_L) return RSX_EUNSUPPORTED;
  if (B == 0) return RSX_OK;
  RSX_LAUNCH(gather_cross_fwd_k, dim3((B + 3) / 4), dim3(256), 0, rsx_s(stream), tables, row_off, ids, E, W, Bc, wout, s, cz, B, F, L);
  RSX_CHECK_LAUNCH();
  return RSX_OK;
}

extern "C" int rsx_gather_fm_fwd_sort(const float* tables, const float* w1, const int32_t* row_off, const int32_t* ids,
                                      float* E, float* S, float* y1, float* y2, uint64_t w1_field_mask, int B, int F,
                                      int D, const rsx_sort_job* sort_h, rsx_stream_t stream) {
  if (sort_h == nullptr) return rsx_gather_fm_fwd(tables, w1, row_off, ids, E, S, y1, y2, w1_field_mask, B, F, D, stream);
  if (B <= 0 || F <= 0 || F > 64 || !d_ok(D)) return RSX_EINVAL;
  if (!tables || !row_off || !ids || !E) return RSX_EINVAL;
  if ((y1 != nullptr) != (w1 != nullptr)) return RSX_EINVAL;
  if (y2 != nullptr && S == nullptr) return RSX_EINVAL;
  SortArgs sa;
  size_t lds = 0;
  const int rc = sort_job_args(*sort_h, sa, &lds);
  if (rc != RSX_OK) return rc;
  const int n_gather = (B + 3) / 4;
  RSX_DISPATCH_D(D, launch_gather_sort, dim3((unsigned)(n_gather + sort_h->F)), lds, rsx_s(stream), tables, w1, row_off, ids, E,
                 S, y1, y2, w1_field_mask, B, F, n_gather, sa);
  RSX_CHECK_LAUNCH();
  return RSX_OK;
}

template <int D>
static void launch_gather_fm_head(dim3 grid, hipStream_t st, const float* tables, const float* w1, const int32_t* row_off,
                                  const int32_t* ids, float* S, uint64_t mask, int B, int F, const FmHeadFused& h) {
  RSX_COUNT_LAUNCH(); gather_fm_head_k<D><<<grid, dim3(1024), 0, st>>>(tables, w1, row_off, ids, S, mask, B, F, h);
}

extern "C" int rsx_gather_fm_head(const float* tables, const float* w1, const int32_t* row_off, const int32_t* ids, float* S,
                                  uint64_t w1_field_mask, const float* c0, const float* wo, const float* bo,
                                  const float* labels, float* prob, float* gy1, float* gy2, float* terms, int term_stride,
                                  int n_dense, int off_c0, int off_wo, int off_bo, float loss_scale, int B, int F, int D,
                                  rsx_stream_t stream) {
  if (B < 0 || F <= 0 || !d_ok(D)) return RSX_EINVAL;
  if (B == 0) return RSX_OK;
  if (!tables || !w1 || !row_off || !ids || !S || !c0 || !wo || !bo || !labels || !prob || !gy1 || !gy2 || !terms)
    return RSX_EINVAL;
  if (term_stride < n_dense + 1 || term_stride > 64 || (term_stride & 3) || n_dense <= 0 || off_c0 < 0 || off_wo < 0 ||
      off_bo < 0 || off_c0 >= n_dense || off_wo + 1 >= n_dense || off_bo >= n_dense)
    return RSX_EINVAL;
  const FmHeadFused h{c0, wo, bo, labels, prob, gy1, gy2, terms, term_stride, n_dense, off_c0, off_wo, off_bo, loss_scale};
  RSX_DISPATCH_D(D, launch_gather_fm_head, dim3((unsigned)((B + 15) / 16)), rsx_s(stream), tables, w1, row_off, ids, S,
                 w1_field_mask, B, F, h);
  RSX_CHECK_LAUNCH();
  return RSX_OK;
}


extern "C" int rsx_field_sort(const int32_t* ids, const int32_t* row_off, int32_t* perm, int32_t* seg_off,
                              int32_t* uniq_row, int32_t* nuniq, int32_t* slot, int32_t* segid, int max_rows_per_field,
                              int B, int F, int stride, rsx_stream_t stream) {
  if (!ids || !row_off || !perm || !seg_off || !uniq_row || !nuniq || !slot || B < 0 || F <= 0 || stride < B ||
      max_rows_per_field <= 0)
    return RSX_EINVAL;
  SortArgs a{ids, row_off, perm, seg_off, uniq_row, nuniq, slot, segid, B, F, stride, 0, 0};
  int n = 128;
  while (n < B) n <<= 1;
  const int T = n <= 512 ? n : ((n >> 1) < 1024 ? (n >> 1) : 1024);   // n <= 512: one thread per key (rank sort)
  const int rc = rsx_sort_args(a, max_rows_per_field, T);
  if (rc != RSX_OK) return rc;
  {
    const int rs = launch_sort_split(&a, 1, max_rows_per_field, rsx_s(stream));
    if (rs != 1) {
      if (rs != RSX_OK) return rs;
      RSX_CHECK_LAUNCH();
      return RSX_OK;
    }
  }
  const size_t lds = rsx_sort_lds_bytes(a, T);
  if (lds > 64 * 1024) {     // opt in to the CU's full 160 KB LDS (B > 4096)
    static const hipError_t attr =
        hipFuncSetAttribute(reinterpret_cast<const void*>(field_sort_k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (attr != hipSuccess) return RSX_EUNSUPPORTED;
  }
  RSX_LAUNCH(field_sort_k, dim3(F), dim3(T), lds, rsx_s(stream), a);
  RSX_CHECK_LAUNCH();
  return RSX_OK;
}

extern "C" int rsx_field_sort_multi(const rsx_sort_job* jobs_h, int njobs, rsx_stream_t stream) {
  if (!jobs_h || njobs <= 0 || njobs > RSX_ADAM_WINDOW_MAX) return RSX_EINVAL;
  SortMulti m;
  int T = 0;
  size_t lds = 0;
  for (int k = 0; k < njobs; ++k) {
    const rsx_sort_job& j = jobs_h[k];
    if (!j.ids || !j.row_off || !j.perm || !j.seg_off || !j.uniq_row || !j.nuniq || !j.slot || j.B < 0 || j.F <= 0 ||
        j.stride < j.B || j.max_rows_per_field <= 0)
      return RSX_EINVAL;
    if (j.B != jobs_h[0].B || j.F != jobs_h[0].F || j.stride != jobs_h[0].stride) return RSX_EINVAL;
    for (int i = 0; i < k; ++i)
      if (jobs_h[i].slot == j.slot || jobs_h[i].perm == j.perm) return RSX_EINVAL;       // one workspace per job
    m.a[k] = SortArgs{j.ids, j.row_off, j.perm, j.seg_off, j.uniq_row, j.nuniq, j.slot, j.segid, j.B, j.F, j.stride, 0, 0, j.skip_mask};
    int n = 128;
    while (n < j.B) n <<= 1;
    T = n <= 512 ? n : ((n >> 1) < 1024 ? (n >> 1) : 1024);
    const int rc = rsx_sort_args(m.a[k], j.max_rows_per_field, T);
    if (rc != RSX_OK) return rc;
    const size_t need = rsx_sort_lds_bytes(m.a[k], T);
    if (need > lds) lds = need;
  }
  if (jobs_h[0].B == 0) return RSX_OK;
  {
    int mr = 0;
    for (int k = 0; k < njobs; ++k) mr = jobs_h[k].max_rows_per_field > mr ? jobs_h[k].max_rows_per_field : mr;
    const int rs = launch_sort_split(m.a, njobs, mr, rsx_s(stream));
    if (rs != 1) {
      if (rs != RSX_OK) return rs;
      RSX_CHECK_LAUNCH();
      return RSX_OK;
    }
  }
  for (int k = njobs; k < RSX_ADAM_WINDOW_MAX; ++k) m.a[k] = m.a[0];
  if (lds > 64 * 1024) {
    static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(field_sort_multi_k),
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (attr != hipSuccess) return RSX_EUNSUPPORTED;
  }
  RSX_LAUNCH(field_sort_multi_k, dim3((unsigned)jobs_h[0].F, (unsigned)njobs), dim3(T), lds, rsx_s(stream), m);
  RSX_CHECK_LAUNCH();
  return RSX_OK;
}

static int segsum_impl(const float* tables, const float* S, const float* dX, const float* gy1, const float* gy2,
                       const int32_t* perm, const int32_t* seg_off, const int32_t* uniq_row, const int32_t* nuniq, float* G,
                       float* gw1, uint64_t w1_field_mask, int B, int F, int D, int stride, int null_row,
                       const rsx_seg_partials* partials_h, const rsx_example_blocks* blocks_h, rsx_stream_t stream,
                       const int32_t* goff = nullptr) {
  if (!perm || !seg_off || !uniq_row || !nuniq || !G || B < 0 || F <= 0 || F > 64 || stride < B || !d_ok(D))
    return RSX_EINVAL;
  if (gy2 != nullptr && (S == nullptr || tables == nullptr)) return RSX_EINVAL;
  if ((gy1 != nullptr) != (gw1 != nullptr)) return RSX_EINVAL;
  if (B == 0) return RSX_OK;
  SegPartials part;
  const int rcp = seg_partials(partials_h, gy1 != nullptr, part);
  if (rcp != RSX_OK) return rcp;
  part.lds_mode = seg_lds_ok(B, D, part.P != nullptr) ? 1 : 0;
  ExBlocks xb;
  const int rcb = ex_blocks(blocks_h, B, xb);
  if (rcb != RSX_OK) return rcb;
  const int gpw = 64 / (D / 4);                                   // unique rows per wave
  const long long waves = (long long)F * seg_waves_per_field(B, gpw, part.P != nullptr);   // two-stage: + helpers
  long long wgs = (waves + 3) / 4;
  if (part.P != nullptr && wgs > SEG_STAGE_B_MAX_WG) wgs = SEG_STAGE_B_MAX_WG;   // (grid stride over the compact unit list)
  const dim3 grid((unsigned)wgs), block(256);
  if (goff != nullptr && part.G == G) return RSX_EINVAL;     // packed output: stage A's G is the full-stride scratch, not the output
  RSX_DISPATCH_D(D, launch_segsum, grid, block, rsx_s(stream), tables, S, dX, gy1, gy2, perm, seg_off, uniq_row,
                 nuniq, G, gw1, w1_field_mask, B, F, stride, null_row, part, xb, goff);
  RSX_CHECK_LAUNCH();
  return RSX_OK;
}

extern "C" int rsx_segsum_bwd(const float* tables, const float* S, const float* dX, const float* gy1,
                              const float* gy2, const int32_t* perm, const int32_t* seg_off,
                              const int32_t* uniq_row, const int32_t* nuniq, float* G, float* gw1,
                              uint64_t w1_field_mask, int B, int F, int D, int stride,
                              const rsx_seg_partials* partials_h, const rsx_example_blocks* blocks_h,
                              rsx_stream_t stream) {
  return segsum_impl(tables, S, dX, gy1, gy2, perm, seg_off, uniq_row, nuniq, G, gw1, w1_field_mask, B, F, D, stride, -1,
                     partials_h, blocks_h, stream);
}

extern "C" int rsx_segsum_bwd_packed(const float* tables, const float* S, const float* dX, const float* gy1,
                                     const float* gy2, const int32_t* perm, const int32_t* seg_off,
                                     const int32_t* uniq_row, const int32_t* nuniq, float* G, float* gw1,
                                     uint64_t w1_field_mask, int B, int F, int D, int stride, int null_row,
                                     const rsx_seg_partials* partials_h, const int32_t* goff, rsx_stream_t stream) {
  if (!goff) return RSX_EINVAL;
  return segsum_impl(tables, S, dX, gy1, gy2, perm, seg_off, uniq_row, nuniq, G, gw1, w1_field_mask, B, F, D, stride, null_row,
                     partials_h, nullptr, stream, goff);
}

extern "C" int rsx_segsum_partials(const float* tables, const float* S, const float* dX, const float* gy1,
                                   const float* gy2, const int32_t* perm, const int32_t* seg_off,
                                   const int32_t* uniq_row, const rsx_seg_partials* ws_h, uint64_t w1_field_mask, int B,
                                   int F, int D, int stride, int null_row, const rsx_example_blocks* blocks_h,
                                   rsx_stream_t stream) {
  if (!perm || !seg_off || !uniq_row || !ws_h || B < 0 || F <= 0 || F > 64 || stride < B || !d_ok(D)) return RSX_EINVAL;
  if (gy2 != nullptr && (S == nullptr || tables == nullptr)) return RSX_EINVAL;
  if (B == 0) return RSX_OK;
  SegPartials ws;
  const int rc = seg_partials(ws_h, gy1 != nullptr, ws);
  if (rc != RSX_OK) return rc;
  ExBlocks xb;
  const int rcb = ex_blocks(blocks_h, B, xb);
  if (rcb != RSX_OK) return rcb;
  if (ws.G == nullptr) return RSX_EINVAL;                          // stage A finishes the short segments into G (/ gw1)
  if (gy2 == nullptr && dX == nullptr) return RSX_EINVAL;          // nothing to sum
  const int pos = (256 / (D / 4)) * SEG_CHUNK;                     // sorted positions per workgroup
  const dim3 grid((unsigned)((size_t)F * ((B + pos - 1) / pos)));
  RSX_DISPATCH_D(D, launch_tiles, grid, rsx_s(stream), tables, S, dX, gy1, gy2, perm, uniq_row, ws, w1_field_mask, B, F,
                 stride, null_row, xb);
  RSX_CHECK_LAUNCH();
  return RSX_OK;
}

extern "C" int rsx_segsum_partials_ride(const float* tables, const float* S, const float* dX, const float* gy1,
                                        const float* gy2, const int32_t* perm, const int32_t* seg_off,
                                        const int32_t* uniq_row, const rsx_seg_partials* ws_h, uint64_t w1_field_mask, int B,
                                        int F, int D, int stride, int null_row, const rsx_example_blocks* blocks_h,
                                        const rsx_scatter_riders* riders, rsx_stream_t stream) {
  const bool any = riders != nullptr && (riders->n_dw > 0 || riders->cross.n > 0 || riders->n_vec > 0);
  if (riders != nullptr && (riders->n_vec < 0 || riders->n_vec > RSX_VEC_REDUCE_MAX_JOBS)) return RSX_EINVAL;
  // the rider forms that exist: dcn.py's scatter (D = 16, dX only) and din.py's (D = 32, dX + first order)
  const bool form = (D == 16 && gy2 == nullptr && gy1 == nullptr) || (D == 32 && gy2 == nullptr && gy1 != nullptr);
  if (any && (!form || B == 0)) {        // no rider form for this launch: the riders' own launches, then the plain stage A
    if (riders->n_dw > 0) {
      const int rc = rsx_tower_reduce_dw_jobs(riders->dw, riders->n_dw, stream);
      if (rc != RSX_OK) return rc;
    }
    const int rc = rsx_cross_reduce_run(&riders->cross, stream);
    if (rc != RSX_OK) return rc;
    const int rcv = rsx_vec_reduce_run(riders->vec, riders->n_vec, stream);
    if (rcv != RSX_OK) return rcv;
  }
  if (!any || !form || B == 0)
    return rsx_segsum_partials(tables, S, dX, gy1, gy2, perm, seg_off, uniq_row, ws_h, w1_field_mask, B, F, D, stride, null_row,
                               blocks_h, stream);
  if (!perm || !seg_off || !uniq_row || !ws_h || B < 0 || F <= 0 || F > 64 || stride < B || !dX) return RSX_EINVAL;
  SegPartials ws;
  const int rc = seg_partials(ws_h, gy1 != nullptr, ws);
  if (rc != RSX_OK) return rc;
  ExBlocks xb;
  const int rcb = ex_blocks(blocks_h, B, xb);
  if (rcb != RSX_OK) return rcb;
  if (ws.G == nullptr) return RSX_EINVAL;
  DwReduceJobs dwj;
  uint32_t n_dw = 0;
  const int rcd = dw_reduce_pack(riders->dw, riders->n_dw, dwj, &n_dw);
  if (rcd != RSX_OK) return rcd;
  const rsx_cross_reduce_job cr = riders->cross;
  if (cr.n > 0 && (!cr.part || !cr.dW || !cr.dB || cr.RT <= 0 || cr.L <= 0 || cr.dim <= 0)) return RSX_EINVAL;
  const uint32_t n_cross = cr.n > 0 ? (uint32_t)((cr.n + 15) / 16) : 0u;
  rsx_vec_reduce_job v0{nullptr, nullptr, 0, 0}, v1{nullptr, nullptr, 0, 0};
  if (riders->n_vec > 0) v0 = riders->vec[0];
  if (riders->n_vec > 1) v1 = riders->vec[1];
  if ((v0.n > 0 && (!v0.part || !v0.out || v0.G <= 0)) || (v1.n > 0 && (!v1.part || !v1.out || v1.G <= 0))) return RSX_EINVAL;
  const uint32_t n_v0 = v0.n > 0 ? (uint32_t)((v0.n + 63) / 64) : 0u, n_v1 = v1.n > 0 ? (uint32_t)((v1.n + 63) / 64) : 0u;
  const int pos = (256 / (D / 4)) * SEG_CHUNK;
  const uint32_t n_own = (uint32_t)((size_t)F * ((B + pos - 1) / pos));
  const dim3 grid(n_own + n_dw + n_cross + n_v0 + n_v1);
  const size_t lds_own = D == 16 ? SegTile<4>::lds_bytes : SegTile<8>::lds_bytes;
  const size_t lds = lds_own > 4096 ? lds_own : 4096;                  // (the riders' LDS: 256 / 1 024 floats)
  RSX_COUNT_LAUNCH();
  if (D == 16)
    segsum_tiles_ride_k<16, false, false><<<grid, dim3(256), lds, rsx_s(stream)>>>(
        tables, S, dX, gy1, gy2, perm, uniq_row, ws, w1_field_mask, B, F, stride, null_row, xb, dwj, cr, v0, v1, n_own, n_dw, n_cross, n_v0);
  else
    segsum_tiles_ride_k<32, false, true><<<grid, dim3(256), lds, rsx_s(stream)>>>(
        tables, S, dX, gy1, gy2, perm, uniq_row, ws, w1_field_mask, B, F, stride, null_row, xb, dwj, cr, v0, v1, n_own, n_dw, n_cross, n_v0);
  RSX_CHECK_LAUNCH();
  return RSX_OK;
}

template <int D>
static void launch_segsum_adam(dim3 grid, dim3 block, hipStream_t st, const float* S, const float* dX, const float* gy1,
                               const float* gy2, const int32_t* perm, const int32_t* seg_off, const int32_t* uniq_row,
                               const int32_t* nuniq, uint64_t mask, int B, int F, int stride, const HotAdam& h,
                               const SegPartials& part, const ExBlocks& xb) {
  const size_t lds = part.lds_mode ? seg_lds_bytes(B, D) : 0;
  // (the window pass's rows per lane group as a template parameter: one instantiation holding both forms ran at 199
  // registers instead of 159 and DeepFM's step 3 % slower)
  RSX_COUNT_LAUNCH();
  if (h.win_nr == 4)
    segsum_adam_k<D, 4><<<grid, block, lds, st>>>(S, dX, gy1, gy2, perm, seg_off, uniq_row, nuniq, mask, B, F, stride, h, part,
                                                  xb);
  else
    segsum_adam_k<D, 1><<<grid, block, lds, st>>>(S, dX, gy1, gy2, perm, seg_off, uniq_row, nuniq, mask, B, F, stride, h, part,
                                                  xb);
}

extern "C" int rsx_segsum_adam_rows(float* tables, float* m_t, float* v_t, float* w1, float* m_w, float* v_w,
                                    const float* S, const float* dX, const float* gy1, const float* gy2,
                                    const int32_t* perm, const int32_t* seg_off, const int32_t* uniq_row,
                                    const int32_t* nuniq, uint64_t w1_field_mask, int B, int F, int D, int stride,
                                    const rsx_adam_seg* extra_segs_h, int n_extra, const rsx_adam_slice* sweep_h,
                                    const rsx_seg_partials* partials_h, const rsx_example_blocks* blocks_h,
                                    const rsx_table_set* second_h, const rsx_adam_window* win_h, float* state,
                                    int advance_step, float lr, float beta1, float beta2, float eps, rsx_stream_t stream) {
  return rsx_segsum_adam_rows2(tables, m_t, v_t, w1, m_w, v_w, S, dX, gy1, gy2, perm, seg_off, uniq_row, nuniq, w1_field_mask, B,
                               F, D, stride, extra_segs_h, n_extra, sweep_h, partials_h, blocks_h, second_h, win_h, state,
                               advance_step, lr, beta1, beta2, eps, 1, 0, stream);
}

extern "C" int rsx_segsum_adam_rows2(float* tables, float* m_t, float* v_t, float* w1, float* m_w, float* v_w,
                                     const float* S, const float* dX, const float* gy1, const float* gy2,
                                     const int32_t* perm, const int32_t* seg_off, const int32_t* uniq_row,
                                     const int32_t* nuniq, uint64_t w1_field_mask, int B, int F, int D, int stride,
                                     const rsx_adam_seg* extra_segs_h, int n_extra, const rsx_adam_slice* sweep_h,
                                     const rsx_seg_partials* partials_h, const rsx_example_blocks* blocks_h,
                                     const rsx_table_set* second_h, const rsx_adam_window* win_h, float* state,
                                     int advance_step, float lr, float beta1, float beta2, float eps, int w1_stride,
                                     int w1_sparse_formula, rsx_stream_t stream) {
  if (w1_stride < 1) return RSX_EINVAL;
  if (!tables || !m_t || !v_t || !perm || !seg_off || !uniq_row || !nuniq || !state || B <= 0 || F <= 0 || F > 64 ||
      stride < B || !d_ok(D))
    return RSX_EINVAL;
  if (gy2 != nullptr && S == nullptr) return RSX_EINVAL;
  if ((gy1 != nullptr) != (w1 != nullptr)) return RSX_EINVAL;
  if (w1 != nullptr && (!m_w || !v_w)) return RSX_EINVAL;
  SegPartials part;
  const int rcp = seg_partials(partials_h, gy1 != nullptr, part);
  if (rcp != RSX_OK) return rcp;
  part.lds_mode = seg_lds_ok(B, D, part.P != nullptr) ? 1 : 0;
  ExBlocks xb;
  const int rcb = ex_blocks(blocks_h, B, xb);
  if (rcb != RSX_OK) return rcb;
  HotAdam h;
  const int rch = hot_adam_init(h, tables, m_t, v_t, w1, m_w, v_w, w1_stride, w1_sparse_formula, extra_segs_h, n_extra, sweep_h,
                                win_h, uniq_row, state, advance_step, lr, beta1, beta2, eps, F, D);
  if (rch != RSX_OK) return rch;
  if (second_h != nullptr) {
    if (!second_h->tables || !second_h->m || !second_h->v || !second_h->dX) return RSX_EINVAL;
    h.tables2 = second_h->tables; h.m_t2 = second_h->m; h.v_t2 = second_h->v; h.dX2 = second_h->dX;
    const int rc2 = seg_partials(second_h->partials, false, h.part2);
    if (rc2 != RSX_OK) return rc2;
    if ((h.part2.P != nullptr) != (part.P != nullptr)) return RSX_EINVAL;   // both sets one- or two-stage
  }
  const int gpw = 64 / (D / 4);
  const long long waves = (long long)F * seg_waves_per_field(B, gpw, part.P != nullptr);
  h.n_own = (uint32_t)((waves + 3) / 4);
  if (part.P != nullptr && h.n_own > (uint32_t)SEG_STAGE_B_MAX_WG) h.n_own = SEG_STAGE_B_MAX_WG;   // (grid stride, compact units)
  h.total_blocks = (second_h != nullptr ? 2u : 1u) * h.n_own + h.win_blk + h.extra.n_blk + h.cold.n_blk;
  const dim3 grid(h.total_blocks), block(256);
  RSX_DISPATCH_D(D, launch_segsum_adam, grid, block, rsx_s(stream), S, dX, gy1, gy2, perm, seg_off, uniq_row, nuniq,
                 w1_field_mask, B, F, stride, h, part, xb);
  RSX_CHECK_LAUNCH();
  return RSX_OK;
}

extern "C" int rsx_segsum_rows(const float* vals, const int32_t* perm, const int32_t* seg_off, const int32_t* uniq_row,
                               const int32_t* nuniq, float* G, int N, int K, int stride, int null_row,
                               const rsx_seg_partials* partials_h, rsx_stream_t stream) {
  return segsum_impl(nullptr, nullptr, vals, nullptr, nullptr, perm, seg_off, uniq_row, nuniq, G, nullptr, 0, N, 1, K,
                     stride, null_row, partials_h, nullptr, stream);
}

// ---- staged host batches -> the step's static input buffers, as a KERNEL -----------------------------------------------------
// The window graphs of the streaming TRAIN path (estimator.py _train_window_packed) read their 8 packed host batches (432 KB)
// straight from the pinned staging buffer with this launch, captured as the graph's first node: a window is then ONE
// hipGraphLaunch -- no hipMemcpyAsync, copy-engine start-up or cross-stream event in front of it (bench.py --host_input:
// 0.0680 ms per DeepFM step against 0.0696 with one in-line copy per window and 0.0749-0.0786 with copies on a copy stream).
// Pinned host memory is mapped into the device's address space; 16-byte loads.
__global__ __launch_bounds__(256) void copy_bytes_k(uint4* __restrict__ dst, const uint4* __restrict__ src, const size_t n16) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) dst[i] = src[i];
}

extern "C" int rsx_copy_bytes(void* dst, const void* src, size_t nbytes, rsx_stream_t stream) {
  if (nbytes == 0) return RSX_OK;
  if (!dst || !src || (nbytes & 15) || ((uintptr_t)dst & 15) || ((uintptr_t)src & 15)) return RSX_EINVAL;
  const size_t n16 = nbytes >> 4;
  const size_t blocks = (n16 + 255) / 256;
  RSX_LAUNCH(copy_bytes_k, dim3((unsigned)(blocks < 1024 ? blocks : 1024)), dim3(256), 0, rsx_s(stream),
                     static_cast<uint4*>(dst), static_cast<const uint4*>(src), n16);
  RSX_CHECK_LAUNCH();
  return RSX_OK;
}

#ifdef RSX_STAMPS
// profiling build only: copy this unit's phase stamps (100 MHz wall clock ticks) to the host
extern "C" int rsx_dbg_stamps_embedding_zero() {
  static const unsigned long long z[64] = {0};
  return hipMemcpyToSymbol(HIP_SYMBOL(rsx_stamps_d), z, sizeof(z)) == hipSuccess ? RSX_OK : RSX_ELAUNCH;
}
extern "C" int rsx_dbg_stamps_embedding(unsigned long long* out_h) {
  return hipMemcpyFromSymbol(out_h, HIP_SYMBOL(rsx_stamps_d), sizeof(unsigned long long) * 64) == hipSuccess ? RSX_OK : RSX_ELAUNCH;
}
#endif
