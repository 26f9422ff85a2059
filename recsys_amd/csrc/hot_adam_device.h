// The touched-row half of the exact TF-1 Adam update as launch state shared by the optimizer launches that own unique rows:
// segsum_adam_k (embedding.hip: segment-sum of per-example gradients, single replica and the pre-dedup data-parallel exchange)
// and merged_adam_k (uniq_exchange.hip: the data-parallel exchange of per-rank unique-row lists).  Both carry the same riders
// -- dense-variable segments, an optional slice of the untouched-row sweep, the LAZY window pass -- so the struct and the
// window pass live here, one body for both (same bits).
#pragma once
#include "rsx_common.h"
#include "adam_device.h"
#include <stdlib.h>

struct SegPartials {
  const int32_t* segid;   // [F, stride] then [F] long- and [F] huge-segment counters (reset by the sort)
  float* P;               // [F, nch, 2, D]
  float* P1;              // [F, nch, 2]
  float* G;               // [F*stride, D]: stage A FINISHES every segment of <= SEG_SHORT entries and writes its sum here
  float* gw1;             // (gw1: nullable [F*stride]); stage B picks those rows up
  const int32_t* row_off; // [F + 1], needed for null_row == RSX_NULL_LAST_ROW only
  int null_row;           // padding row: >= 0 a global row, RSX_NULL_LAST_ROW the last row of every field, RSX_NULL_NONE
  int lds_mode;           // single stage only: the workgroup-cooperative form through LDS (segsum_wave_lds; set by the host)
  uint64_t skip;          // fields the sort skipped (rsx_sort_job.skip_mask): stage A's position tiles have nothing to walk
  // segid + F*stride: [F] long- and [F] huge-segment counts, then the long list [F, nch] (unique index j of the
  // field's long segments from the front, huge ones from the back) -- all written by the sort
  __host__ __device__ const int32_t* counts(int F, int stride) const { return segid + (size_t)F * stride; }
  __host__ __device__ const int32_t* long_list(int F, int stride) const { return segid + (size_t)F * stride + 2 * F; }
  // the padding row of field f (or -1): wave-uniform
  __device__ __forceinline__ int null_of(int f, int explicit_row) const {
    const int nr = explicit_row != RSX_NULL_NONE ? explicit_row : null_row;
    return nr == RSX_NULL_LAST_ROW ? row_off[f + 1] - 1 : nr;
  }
};

// Segment-sum fused with the touched-row half of the exact TF-1 Adam update (the untouched rows are swept by COLD
// slices): the group that owns unique row (f, j) has its summed gradient in registers and is the only reader/writer
// of that row, so it applies   m = m*b1 + g(1-b1); v = v*b2 + g*g(1-b2); var -= (alpha*m)/(sqrt(v)+eps)   at once --
// no G round trip, no separate launch.  The first-order vector uses the ApplyAdam (dense) formula.  Extra
// workgroups carry the dense-variable segment; the last workgroup to finish advances the beta powers.
struct HotAdam {
  float* tables; float* m_t; float* v_t;
  float* w1; float* m_w; float* v_w;     // nullable
  int w1_stride;                         // floats between the elements of w1 / m_w / v_w (1; 4: column 0 of a 4-wide table)
  int w1_sparse;                         // != 0: the sparse (IndexedSlices) formula for w1 -- a 1-D variable read through
                                         // tf.gather (din/din.py:96) -- instead of the dense-kernel formula (fm/fm.py:121)
  float lr, b1, b2, eps;
  float* state;
  uint32_t n_own, total_blocks;
  int advance;                           // the last workgroup advances the beta powers / step counter (0: a later launch of the
                                         // same step does -- e.g. the first of xDeepFM's two table sets)
  // optional second table set looked up with the same ids (same sort outputs): its row-owner workgroups follow the first
  // set's in the same grid (xDeepFM's two input_layer calls); no first-order vector, no FM term
  float* tables2; float* m_t2; float* v_t2; const float* dX2;
  SegPartials part2;
  AdamSlice extra;                       // dense variables (any non-COLD kinds), n_blk may be 0
  AdamSlice cold;                        // optional slice of the untouched-row sweep (rows disjoint from the touched ones)
  // optimizer window (rsx_adam_window): rows that ANOTHER step of the window touches and this one does not are skipped by
  // the window's sweep and get this step's untouched-row update here, from the other steps' unique-row lists
  int win_k, win_cur;
  const int32_t* win_uniq[RSX_ADAM_WINDOW_MAX];
  const int32_t* win_nuniq[RSX_ADAM_WINDOW_MAX];
  const int32_t* win_slot[RSX_ADAM_WINDOW_MAX];
  uint32_t win_blk, win_per_f;           // workgroups of the pass; per (list, field)
  int win_nr;                            // rows per lane group in the pass: 1 (batches up to 1024) or 4
  int win_compact;                       // window_pass_compact (grid stride over the compact unit list) instead of the dense grid
};

#ifndef RSX_WIN_PASS_NT
#define RSX_WIN_PASS_NT 0      // (A/B knob: streaming stores in the window pass)
#endif

// LAZY window pass of segsum_adam_k (round 3).  A row that some step of the window touches is skipped by the window's sweep;
// its zero-gradient updates of the steps that do NOT touch it are applied here, by extra workgroups of the scatter launch,
// when they are needed instead of step by step:
//   * step cur < k - 1 walks the NEXT step's unique-row list: a row that step cur does not touch itself receives the updates
//     of the steps (its last touch, cur] back to back in registers -- so the next step's gather reads, and its scatter
//     updates, a row that is current;
//   * the window's last step walks every earlier list: a row whose LAST touch was step o receives the updates of the steps
//     (o, k - 1].
// Every row thus gets the same updates in the same order as step by step -- with 2 (k - 1) list walks per window.  (Round 2
// walked ALL other lists in EVERY step, one update per visit: k (k - 1) walks, 273 workgroups and 11 MB of traffic per DeepFM
// step.  Measured and dropped: also finishing, at every step, the rows whose last touch was the step before -- no heavy last
// step, but two lists per step: 0.4 us per step slower on DeepFM; keeping round 2's form for large batches inside this
// code: dcn.py bs 4096 0.210 ms against 0.202 lazy.)  The step size of window step s is state[8 + s]: the window's sweep
// (adam_window_k) writes all of them at the window's start with the products the per-step advance of the beta powers
// makes, and every step's launch leaves its own there too.  Inside a window the tables are not a state any step-by-step
// run passes through (that was already so).
// NR rows per LPR-lane group (1: batches up to 1024, where the launch is latency-bound -- more workgroups, shorter chains; 4
// above), phase by phase (unique rows; all slot maps + the rows' state; the pending updates; stores).
// window_pass_rows: the workgroup's rows are jbase + thread / LPR (+ i * 256 / LPR, i < NR) of field f in list li (tail step)
// or in the next step's list.  Two ways of dealing the (list, field, row block) units to workgroups:
//   window_pass          a dense grid sized for `max_unique` rows in EVERY field of every walked list (segsum_adam_k: lists of a
//                        few hundred rows, the workgroups past a list's end leave at once);
//   window_pass_compact  a fixed number of workgroups walk the COMPACT unit list with a grid stride (merged_adam_k, round 5:
//                        the GLOBAL lists of a data-parallel step are bounded by min(N b, rows of the field) -- 2 048 at
//                        8 x 256 -- while 25 of Criteo's 39 fields hold fewer than 1 500 rows: the dense grid of a window's
//                        last step was 7 lists x 39 fields x 32 = 8 736 workgroups, most of them empty).
template <int D, int NR>
__device__ __forceinline__ void window_pass_rows(const HotAdam& h, const int li, const int f, const int jbase, const float b1p,
                                                 const float b2p, const int stride);

template <int D, int NR>
__device__ __forceinline__ void window_pass(const HotAdam& h, const uint32_t wb, const float b1p, const float b2p,
                                            const int F, const int stride) {
  constexpr int RPW = 256 / (D / 4);
  const uint32_t per_l = (uint32_t)F * h.win_per_f;
  const int li = (int)(wb / per_l);
  const uint32_t rem = wb - (uint32_t)li * per_l;
  const int f = (int)(rem / h.win_per_f);
  window_pass_rows<D, NR>(h, li, f, (int)(rem - (uint32_t)f * h.win_per_f) * (RPW * NR), b1p, b2p, stride);
}

template <int D, int NR>
__device__ __forceinline__ void window_pass_compact(const HotAdam& h, const uint32_t wb, const float b1p, const float b2p,
                                                    const int F, const int stride) {
  constexpr int RPB = (256 / (D / 4)) * NR;            // rows per workgroup unit
  const int lane = threadIdx.x & 63;
  const int cur = h.win_cur, wk = h.win_k;
  const bool tail = cur == wk - 1;
  const int nlists = tail ? wk - 1 : 1;
  // lane f: inclusive unit count of fields 0 .. f of list l (all lists' counts in ONE round trip), and the lists' totals
  int incl[RSX_ADAM_WINDOW_MAX - 1];
  int ltot[RSX_ADAM_WINDOW_MAX - 1];
  const int lc = lane < F ? lane : F - 1;
#pragma unroll
  for (int l = 0; l < RSX_ADAM_WINDOW_MAX - 1; ++l) {
    const int o = tail ? (l < nlists ? l : 0) : cur + 1;
    const int nu = h.win_nuniq[o][lc];
    incl[l] = (lane < F && l < nlists) ? (nu + RPB - 1) / RPB : 0;
  }
#pragma unroll
  for (int l = 0; l < RSX_ADAM_WINDOW_MAX - 1; ++l) {
#pragma unroll
    for (int d = 1; d < RSX_WAVE; d <<= 1) {
      const int t = __shfl_up(incl[l], d);
      if (lane >= d) incl[l] += t;
    }
    ltot[l] = __shfl(incl[l], RSX_WAVE - 1);
  }
  int total = 0;
#pragma unroll
  for (int l = 0; l < RSX_ADAM_WINDOW_MAX - 1; ++l) total += ltot[l];
  for (int u = (int)wb; u < total; u += (int)h.win_blk) {          // (block-uniform)
    int li = 0, base = 0, inc_l = incl[0];
#pragma unroll
    for (int l = 1; l < RSX_ADAM_WINDOW_MAX - 1; ++l) {
      int below = 0;
#pragma unroll
      for (int m = 0; m < l; ++m) below += ltot[m];
      if (u >= below && l < nlists) {
        li = l;
        base = below;
        inc_l = incl[l];
      }
    }
    const int v = u - base;
    const int f = __popcll(__ballot(inc_l <= v));      // (lanes >= F hold the list's total, which v never reaches)
    const int bf = __shfl(inc_l, f > 0 ? f - 1 : 0);
    window_pass_rows<D, NR>(h, li, f, (v - (f > 0 ? bf : 0)) * RPB, b1p, b2p, stride);
  }
}

template <int D, int NR>
__device__ __forceinline__ void window_pass_rows(const HotAdam& h, const int li, const int f, const int jbase, const float b1p,
                                                 const float b2p, const int stride) {
  constexpr int LPR = D / 4;
  constexpr int WIN_NR = NR;
  constexpr int RPW = 256 / LPR;
  const uint32_t wb = (uint32_t)jbase;                 // (profiling stamps only)
  const int j0 = jbase + (int)threadIdx.x / LPR;
  const int q = (int)threadIdx.x % LPR;
  const int cur = h.win_cur, wk = h.win_k;
  const bool tail = cur == wk - 1;
  const int o = tail ? li : cur + 1;                   // the list this workgroup walks
  const int last = cur;                                // the last step whose update is applied here
  const int nu = h.win_nuniq[o][f];
  if (j0 < nu) {
    const int32_t* __restrict__ ur = h.win_uniq[o] + (size_t)f * stride;
    int row[WIN_NR];
#pragma unroll
    for (int i = 0; i < WIN_NR; ++i) {
      const int j = j0 + i * RPW;
      row[i] = ur[j < nu ? j : nu - 1];
    }
    // which steps of the window touch the row: all slot maps in ONE round trip (maps past the window re-read this step's)
    uint32_t touch[WIN_NR];
#pragma unroll
    for (int i = 0; i < WIN_NR; ++i) touch[i] = 0u;
    const int32_t* __restrict__ scur = h.win_slot[cur];
#pragma unroll
    for (int l = 0; l < RSX_ADAM_WINDOW_MAX; ++l) {
      const int32_t* __restrict__ sp = l < wk ? h.win_slot[l] : scur;
      const uint32_t bit = l < wk ? (1u << l) : 0u;
#pragma unroll
      for (int i = 0; i < WIN_NR; ++i) touch[i] |= sp[row[i]] >= 0 ? bit : 0u;
    }
    const bool hw1 = h.w1 != nullptr;
    const float* __restrict__ w1p = hw1 ? h.w1 : h.tables;
    const float* __restrict__ mwp = hw1 ? h.m_w : h.m_t;
    const float* __restrict__ vwp = hw1 ? h.v_w : h.v_t;
    const size_t wst = hw1 ? (size_t)h.w1_stride : 0;
    float w[WIN_NR], mw[WIN_NR], vw[WIN_NR];
#pragma unroll
    for (int i = 0; i < WIN_NR; ++i) {
      w[i] = w1p[(size_t)row[i] * wst];
      mw[i] = mwp[(size_t)row[i] * wst];
      vw[i] = vwp[(size_t)row[i] * wst];
    }
    // owner + first pending step of every row
    bool own[WIN_NR];
    int p0[WIN_NR];
#pragma unroll
    for (int i = 0; i < WIN_NR; ++i) {
      const uint32_t m = touch[i];
      if (tail) {
        own[i] = (m >> (o + 1)) == 0u;                 // no later step touches it: step o was its last
        p0[i] = o + 1;
      } else {
        own[i] = ((m >> cur) & 1u) == 0u;              // (touched now: the row owners bring it up to date)
        const uint32_t below = m & ((1u << cur) - 1u);
        p0[i] = below != 0u ? 32 - __clz(below) : 0;   // the step after its last touch
      }
      own[i] = own[i] && j0 + i * RPW < nu;
    }
    // (wave-uniform, by ballot: groups past the list's end are inactive here) does any row of the wave wait for step st?
    auto any_pending = [&](const int st) -> bool {
      bool need = false;
#pragma unroll
      for (int i = 0; i < WIN_NR; ++i) need |= own[i] && st >= p0[i];
      return __builtin_amdgcn_ballot_w64(need) != 0ull;
    };
    Hp hp;
    hp.b1 = h.b1; hp.b2 = h.b2; hp.omb1 = 1.0f - h.b1; hp.omb2 = 1.0f - h.b2; hp.eps = h.eps;
    const float alpha_now = h.lr * sqrtf(1.0f - b2p) / (1.0f - b1p);
    // the step size of window step st (uniform): state[8 + st] for st < cur, this step's own for st == cur.  (Read in
    // place: a register array of them indexed by the loop counter goes to scratch memory on this toolchain.)
    // ORDER DEPENDENCE: words 8 + j are written by the window's sweep (adam_window_k, rsx_adam_slice_run with slot_w) and by
    // every step's own launch (word 8 + cur, end of this kernel); a window pass therefore needs the sweep of ITS window to
    // have run at position 0 -- the only order the host issues (deepfm._train_fused: window_sweep before the first step).
    // Without it the words hold the previous window's values or zeros: the step sizes of steps < cur would be WRONG, not
    // merely slow, so rsx_segsum_adam_rows2 documents the sweep as a precondition of window != NULL (include/rsx.h).
    auto alpha_of = [&](const int st) -> float { return st == cur ? alpha_now : h.state[8 + st]; };
    float amin = alpha_now, amax = alpha_now;
#pragma unroll
    for (int l = 0; l < RSX_ADAM_WINDOW_MAX; ++l) {
      const float a = (l < wk && l != cur) ? h.state[8 + l] : alpha_now;
      amin = fminf(amin, a);
      amax = fmaxf(amax, a);
    }
    const bool fast_ok = RSX_ADAM_WIN_FAST && h.b1 >= 0.85f && h.b1 < 1.f && h.b2 >= 0.5f && h.b2 < 1.f && h.eps >= 0x1p-30f &&
                         h.eps <= 1.f && amin >= 0x1p-24f && amax <= 16.f && amax <= 4.f * amin;
    const uint32_t m_lo_bits = __float_as_uint(fast_ok ? 0x1p-90f / amin : 1.f);
    const int nset = h.tables2 != nullptr ? 2 : 1;
    for (int set = 0; set < nset; ++set) {
      float4* __restrict__ T4 = reinterpret_cast<float4*>(set ? h.tables2 : h.tables);
      float4* __restrict__ M4 = reinterpret_cast<float4*>(set ? h.m_t2 : h.m_t);
      float4* __restrict__ V4 = reinterpret_cast<float4*>(set ? h.v_t2 : h.v_t);
      float4 var[WIN_NR], m[WIN_NR], v[WIN_NR];
#pragma unroll
      for (int i = 0; i < WIN_NR; ++i) {
        const size_t o4 = (size_t)row[i] * LPR + q;
        var[i] = T4[o4]; m[i] = M4[o4]; v[i] = V4[o4];
      }
      RSX_STAMP(41, wb == 0 && set == 0 && touch[0] != 12345u);
      bool bad = !fast_ok;
#pragma unroll
      for (int i = 0; i < WIN_NR; ++i) bad |= own[i] && adam_win_guard4(var[i], m[i], v[i], m_lo_bits);
      if (__builtin_amdgcn_ballot_w64(bad) == 0ull) {
        // packed fast form (adam_fast.h; the guard keeps every operand of <= 8 updates inside its domains)
        rsx_f2 var2[WIN_NR][2], m2[WIN_NR][2], v2[WIN_NR][2];
#pragma unroll
        for (int i = 0; i < WIN_NR; ++i) {
          var2[i][0] = (rsx_f2){var[i].x, var[i].y}; var2[i][1] = (rsx_f2){var[i].z, var[i].w};
          m2[i][0] = (rsx_f2){m[i].x, m[i].y}; m2[i][1] = (rsx_f2){m[i].z, m[i].w};
          // (v == +0, which the guard admits under m == +0 only: compute with 1, store the zero back)
          v2[i][0] = (rsx_f2){v[i].x == 0.f ? 1.f : v[i].x, v[i].y == 0.f ? 1.f : v[i].y};
          v2[i][1] = (rsx_f2){v[i].z == 0.f ? 1.f : v[i].z, v[i].w == 0.f ? 1.f : v[i].w};
        }
#pragma unroll 1
        for (int st = 0; st <= last; ++st) {
          if (!any_pending(st)) continue;
          const float a = alpha_of(st);
#pragma unroll
          for (int i = 0; i < WIN_NR; ++i) {
            if (st >= p0[i]) {
              adam_zero_grad2(var2[i][0], m2[i][0], v2[i][0], a, hp);
              adam_zero_grad2(var2[i][1], m2[i][1], v2[i][1], a, hp);
            }
          }
        }
#pragma unroll
        for (int i = 0; i < WIN_NR; ++i) {
          var[i] = make_float4(var2[i][0].x, var2[i][0].y, var2[i][1].x, var2[i][1].y);
          m[i] = make_float4(m2[i][0].x, m2[i][0].y, m2[i][1].x, m2[i][1].y);
          v[i] = make_float4(v[i].x == 0.f ? 0.f : v2[i][0].x, v[i].y == 0.f ? 0.f : v2[i][0].y,
                             v[i].z == 0.f ? 0.f : v2[i][1].x, v[i].w == 0.f ? 0.f : v2[i][1].y);
        }
      } else {
#pragma unroll 1
        for (int st = 0; st <= last; ++st) {
          if (!any_pending(st)) continue;
          hp.alpha = alpha_of(st);
#pragma unroll
          for (int i = 0; i < WIN_NR; ++i) {
            if (st >= p0[i]) { F4_APPLY(adam_sparse1, var[i], m[i], v[i], F4Z, false, hp); }
          }
        }
      }
#pragma unroll
      for (int i = 0; i < WIN_NR; ++i) {
        if (own[i]) {
          const size_t o4 = (size_t)row[i] * LPR + q;
          T4[o4] = var[i]; M4[o4] = m[i]; V4[o4] = v[i];
        }
      }
    }
    RSX_STAMP(42, wb == 0);
    if (hw1 && q == 0) {
#pragma unroll 1
      for (int st = 0; st <= last; ++st) {
        if (!any_pending(st)) continue;
        hp.alpha = alpha_of(st);
#pragma unroll
        for (int i = 0; i < WIN_NR; ++i) {
          if (h.w1_sparse) adam_zero_grad1<false>(w[i], mw[i], vw[i], st >= p0[i], hp);
          else adam_zero_grad1<true>(w[i], mw[i], vw[i], st >= p0[i], hp);
        }
      }
#pragma unroll
      for (int i = 0; i < WIN_NR; ++i) {
        if (own[i]) {
          const size_t wi = (size_t)row[i] * wst;
          h.w1[wi] = w[i]; h.m_w[wi] = mw[i]; h.v_w[wi] = vw[i];
        }
      }
    }
  }
}


// Host side: everything of a HotAdam that does not depend on how the launch obtains its row gradients -- the variables, the
// hyper-parameters, the riders (dense-variable segments, an optional slice of the untouched-row sweep) and the lazy window
// pass.  The caller sets n_own (+ the second table set) and total_blocks.  uniq_row: this step's unique-row list (must be
// entry `cur` of the window).
static inline int hot_adam_init(HotAdam& h, float* tables, float* m_t, float* v_t, float* w1, float* m_w, float* v_w,
                                int w1_stride, int w1_sparse_formula, const rsx_adam_seg* extra_segs_h, int n_extra,
                                const rsx_adam_slice* sweep_h, const rsx_adam_window* win_h, const int32_t* uniq_row,
                                float* state, int advance_step, float lr, float beta1, float beta2, float eps, int F, int D) {
  h.tables = tables; h.m_t = m_t; h.v_t = v_t; h.w1 = w1; h.m_w = m_w; h.v_w = v_w;
  h.w1_stride = w1_stride; h.w1_sparse = w1_sparse_formula != 0;
  h.lr = lr; h.b1 = beta1; h.b2 = beta2; h.eps = eps; h.state = state; h.advance = advance_step != 0;
  h.tables2 = nullptr; h.m_t2 = nullptr; h.v_t2 = nullptr; h.dX2 = nullptr;
  h.part2 = SegPartials{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, RSX_NULL_NONE, 0};
  h.n_own = 0; h.total_blocks = 0;
  h.extra.n_blk = 0; h.extra.blk_lo = 0;
  if (n_extra > 0) {
    uint32_t blocks = 0;
    const int rc = adam_build_args(extra_segs_h, n_extra, state, lr, beta1, beta2, eps, h.extra.args, &blocks);
    if (rc != RSX_OK) return rc;
    h.extra.n_blk = blocks;
  }
  const int rcs = adam_build_slice(sweep_h, h.cold);
  if (rcs != RSX_OK) return rcs;
  // A VEC_COLD slice rewrites (restores) the touched elements of its float4s: racing with this launch's own update of
  // those elements.  Only table slices (whole touched rows are skipped, never written) may ride here.
  for (int k = 0; h.cold.n_blk != 0 && k < h.cold.args.nseg; ++k) {
    const uint32_t b0 = h.cold.args.seg[k].blk_begin;
    const uint32_t b1 = k + 1 < h.cold.args.nseg ? h.cold.args.seg[k + 1].blk_begin : h.cold.args.total_blocks;
    const bool overlaps = b0 < h.cold.blk_lo + h.cold.n_blk && h.cold.blk_lo < b1;     // segment k has blocks in the slice
    if (overlaps && h.cold.args.seg[k].kind != RSX_ADAM_TABLE_TF1_COLD) return RSX_EINVAL;
  }
  h.win_k = 0; h.win_cur = 0; h.win_blk = 0; h.win_per_f = 0; h.win_nr = 1; h.win_compact = 0;
  for (int i = 0; i < RSX_ADAM_WINDOW_MAX; ++i) h.win_uniq[i] = h.win_nuniq[i] = h.win_slot[i] = nullptr;
  if (win_h != nullptr && win_h->k > 1) {
    if (win_h->k > RSX_ADAM_WINDOW_MAX || win_h->cur < 0 || win_h->cur >= win_h->k || win_h->max_unique <= 0) return RSX_EINVAL;
    for (int i = 0; i < win_h->k; ++i) {
      if (!win_h->uniq_row[i] || !win_h->nuniq[i] || !win_h->slot[i]) return RSX_EINVAL;
      h.win_uniq[i] = win_h->uniq_row[i]; h.win_nuniq[i] = win_h->nuniq[i]; h.win_slot[i] = win_h->slot[i];
    }
    if (win_h->uniq_row[win_h->cur] != uniq_row) return RSX_EINVAL;              // entry `cur` is this step's own sort
    h.win_k = win_h->k; h.win_cur = win_h->cur;
    // rows per lane group: the lazy pass applies up to 8 updates per row back to back -- one row per group (more workgroups,
    // shorter chains) where the launch is latency-bound, four at large batches
    {
      // (measured, round 5: deepfm.py as 8 emulated ranks -- lists of up to 2 048 rows -- 0.0977 ms per step with four rows
      // per group, 0.0911 with one; dcn.py at 8 x 4 096 -- up to 32 768 -- 0.311 with four, 0.337 with one.  RSX_WIN_NR4_MIN: A/B knob)
      static const int nr4_min = getenv("RSX_WIN_NR4_MIN") ? atoi(getenv("RSX_WIN_NR4_MIN")) : 2048;
      h.win_nr = win_h->max_unique > nr4_min ? 4 : 1;
    }
    const int rpw = h.win_nr * 256 / (D / 4);
    h.win_per_f = (uint32_t)((win_h->max_unique + rpw - 1) / rpw);
    // lists walked: the next step's, or -- the window's last step -- every earlier one
    h.win_blk = (uint32_t)(win_h->cur == win_h->k - 1 ? win_h->k - 1 : 1) * (uint32_t)F * h.win_per_f;
  }
  return RSX_OK;
}

// The tail every launch that carries a HotAdam ends with: the last workgroup to arrive publishes this step's step size (read
// by the lazy window pass of the window's later steps) and advances the beta powers / step counter.
__device__ __forceinline__ void hot_adam_finish(const HotAdam& h, const float b1p, const float b2p) {
  __syncthreads();
  if (threadIdx.x == 0 && adam_arrive_last(h.state, h.total_blocks)) {
    if (h.win_k > 1) h.state[8 + h.win_cur] = h.lr * sqrtf(1.0f - b2p) / (1.0f - b1p);
    if (h.advance) {
      h.state[0] = b1p * h.b1;
      h.state[1] = b2p * h.b2;
      reinterpret_cast<uint32_t*>(h.state)[3] += 1u;
    }
  }
}
