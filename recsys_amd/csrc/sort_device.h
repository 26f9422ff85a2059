// Per-field dedup sort as a workgroup-level device function, shared by field_sort_k (embedding.hip) and by
// tower_bwd_k (tower.hip), where the sort of the SAME step rides along as extra workgroups of the backward launch
// (it depends on ids only and is first needed by the segment-sum, so its ~10 us hide behind the tower backward).
#pragma once
#include "rsx_common.h"

struct SortArgs {
  const int32_t* ids;
  const int32_t* row_off;
  int32_t* perm;
  int32_t* seg_off;
  int32_t* uniq_row;
  int32_t* nuniq;
  int32_t* slot;
  // nullable (two-stage segment-sum workspace, B > 512 only): [F, stride] segment index j of every sorted position,
  // then [F] long- and [F] huge-segment counts, then [F, ceil(B/16)] the long list (long from the front, huge from the back)
  int32_t* segid;
  int B, F, stride, n, bbits;
  // fields whose bit is set are NOT sorted (their workspace rows keep nuniq = 0: every consumer then finds nothing to do).
  // The data-parallel step sets it for the small-vocabulary fields, whose gradients travel as dense per-row buckets
  // (rsx_bucket_scatter) instead of through the global sort + scatter.  A field must be skipped always or never.
  uint64_t skip;
};

// lanes of this wave whose `digit` equals mine (all 64 lanes must call it): nbits ballots
__device__ __forceinline__ uint64_t rsx_match_digit(uint32_t digit, int nbits) {
  uint64_t m = ~0ull;
  for (int b = 0; b < nbits; ++b) {
    const bool bit = (digit >> b) & 1u;
    const uint64_t bal = __ballot(bit);
    m &= bit ? bal : ~bal;
  }
  return m;
}

// One workgroup (blockDim a power of two >= 64, <= n) sorts field f by key = (id << bbits) | b.  The key is unique, so
// every path below yields entries ordered by id, then by ascending example index -- the order TF's CPU
// unsorted_segment_sum accumulates in.  n = padded power of two (>= blockDim).
//   n <= 512 (and n <= blockDim): barrier-free rank sort.                          lds: n + 32 words
//   otherwise: stable LSD radix sort in LDS over the id bits only (8-bit digits, ceil(log2(rows_f)/8) <= 3 passes; the
//   initial order IS ascending b, and stability keeps it).  Ranking inside a wave uses ballots (stable, no atomics), waves are
//   ordered by a (digit-major, wave-minor) scan of per-wave counters: ~4 barriers per pass instead of the
//   log2(n)*(log2(n)+1)/2 = 66..105 barrier-separated stages of a bitonic network.  lds: 2n + 256*waves + 32 words
__device__ __forceinline__ void field_sort_block(const SortArgs& a, int f, uint32_t* lds) {
  if ((a.skip >> f) & 1ull) return;                     // (workgroup-uniform)
  const int tid = threadIdx.x, T = blockDim.x;
  const int B = a.B, n = a.n, bbits = a.bbits, stride = a.stride;
  const bool small = n <= 512 && n <= T;
  uint32_t* key = lds;                                  // [n]
  uint32_t* wsum = small ? lds + n : lds + 2 * n;       // [32]
  const int roff = a.row_off[f];
  // forget the previous step's rows of this field (they may differ from this step's); loads batched 8 deep --
  // one dependent global round trip per iteration would otherwise dominate large batches
  const int prev = a.nuniq[f];
  {
    const int32_t* ur = a.uniq_row + (size_t)f * stride;
    int jj = tid;
    for (; jj + 7 * T < prev; jj += 8 * T) {
      int r[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) r[u] = ur[jj + u * T];
#pragma unroll
      for (int u = 0; u < 8; ++u) a.slot[r[u]] = -1;
    }
    for (; jj < prev; jj += T) a.slot[ur[jj]] = -1;
  }
  {
    const int32_t* col = a.ids + f;
    int i = tid;
    for (; i + 7 * T < B; i += 8 * T) {
      uint32_t v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = (uint32_t)col[(size_t)(i + u * T) * a.F];
#pragma unroll
      for (int u = 0; u < 8; ++u) key[i + u * T] = (v[u] << bbits) | (uint32_t)(i + u * T);
    }
    for (; i < n; i += T)
      key[i] = i < B ? (((uint32_t)col[(size_t)i * a.F] << bbits) | (uint32_t)i) : 0xFFFFFFFFu;
  }
  __syncthreads();
  if (small) {
    // small batches: rank sort.  Keys are unique, so rank = #{keys smaller than mine}; every thread scans the n keys
    // with broadcast LDS reads (4 per ds_read_b128) -- n/4 iterations, NO barriers.  n <= 512 keeps the O(n^2)
    // compare count below the radix passes' latency.
    const uint32_t mine = tid < n ? key[tid] : 0xFFFFFFFFu;
    int rank = 0;
    const uint4* k4 = reinterpret_cast<const uint4*>(key);
    for (int i = 0; i < (n >> 2); ++i) {
      const uint4 q = k4[i];
      rank += (q.x < mine) + (q.y < mine) + (q.z < mine) + (q.w < mine);
    }
    __syncthreads();
    if (tid < n) key[rank] = mine;     // real keys land on 0..B-1; the equal padding keys all write slot B (never read)
    __syncthreads();
  } else {
    uint32_t* src = lds;
    uint32_t* dst = lds + n;
    uint32_t* cnt = lds + 2 * n + 32;                   // [waves][256]
    const int nw = T >> 6, w = tid >> 6, lane = tid & 63;
    const int ipw = n / nw;                              // keys per wave: a contiguous range, in order
    const int rows = a.row_off[f + 1] - roff;
    int idbits = 0;
    while ((1 << idbits) < rows) ++idbits;
    const uint64_t lt = (1ull << lane) - 1ull;
    for (int sh = 0; sh < idbits; sh += 8) {
      const int nb = idbits - sh < 8 ? idbits - sh : 8;
      const uint32_t dmask = (1u << nb) - 1u;            // padding keys (all ones) take the top digit: they stay last
      for (int i = tid; i < nw * 256; i += T) cnt[i] = 0;
      __syncthreads();
      // (1) per-wave digit histogram (integer LDS atomics: order-free, so no ballots needed here)
      for (int j = 0; j < ipw; j += 64)
        atomicAdd(&cnt[w * 256 + ((src[w * ipw + j + lane] >> (bbits + sh)) & dmask)], 1u);
      __syncthreads();
      // (2) exclusive scan in (digit-major, wave-minor) order -> first output position of every (wave, digit)
      int tot = 0;
      if (tid < 256) {
        for (int ww = 0; ww < nw; ++ww) {
          const int c = (int)cnt[ww * 256 + tid];
          cnt[ww * 256 + tid] = (uint32_t)tot;
          tot += c;
        }
      }
      int incl = tot;
#pragma unroll
      for (int dd = 1; dd < RSX_WAVE; dd <<= 1) {
        const int o = __shfl_up(incl, dd);
        if (lane >= dd) incl += o;
      }
      if (lane == 63 && tid < 256) wsum[w] = (uint32_t)incl;
      __syncthreads();
      if (tid < 256) {
        int base = incl - tot;
        for (int ww = 0; ww < w; ++ww) base += (int)wsum[ww];
        for (int ww = 0; ww < nw; ++ww) cnt[ww * 256 + tid] += (uint32_t)base;
      }
      __syncthreads();
      // (3) stable scatter: position = running (wave, digit) offset + rank among the equal-digit lanes below me
      for (int j = 0; j < ipw; j += 64) {
        const uint32_t k = src[w * ipw + j + lane];
        const uint32_t d = (k >> (bbits + sh)) & dmask;
        const uint64_t m = rsx_match_digit(d, nb);
        const uint32_t old = cnt[w * 256 + d];
        const int r = __popcll(m & lt);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        if (r == 0) cnt[w * 256 + d] = old + (uint32_t)__popcll(m);
        dst[old + r] = k;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      }
      __syncthreads();
      uint32_t* t = src;
      src = dst;
      dst = t;
    }
    key = src;
  }
  // head flags + exclusive scan -> unique index j of every sorted position
  const int ipt = n / T;
  const int i0 = tid * ipt;
  const uint32_t bmask = (1u << bbits) - 1u;
  int cnt = 0;
  for (int i = i0; i < i0 + ipt; ++i)
    if (i < B && (i == 0 || (key[i] >> bbits) != (key[i - 1] >> bbits))) ++cnt;
  int incl = cnt;
#pragma unroll
  for (int d = 1; d < RSX_WAVE; d <<= 1) {
    const int o = __shfl_up(incl, d);
    if ((tid & 63) >= d) incl += o;
  }
  if ((tid & 63) == 63) wsum[tid >> 6] = (uint32_t)incl;
  __syncthreads();
  int base = incl - cnt, total = 0;
  const int nw = (T + 63) >> 6;
  for (int w = 0; w < nw; ++w) {
    const int v = (int)wsum[w];
    if (w < (tid >> 6)) base += v;
    total += v;
  }
  int jn = base;
  if (small) {
    for (int i = i0; i < i0 + ipt; ++i) {
      if (i >= B) break;
      const uint32_t kk = key[i];
      a.perm[(size_t)f * stride + i] = (int32_t)(kk & bmask);
      if (i == 0 || (kk >> bbits) != (key[i - 1] >> bbits)) {
        const int row = roff + (int)(kk >> bbits);
        a.uniq_row[(size_t)f * stride + jn] = row;
        a.seg_off[(size_t)f * (stride + 1) + jn] = i;
        a.slot[row] = f * stride + jn;
        ++jn;
      }
      if (a.segid != nullptr) a.segid[(size_t)f * stride + i] = jn - 1;
    }
  } else {
    // unique index of every head position into the idle LDS buffer, then ONE coalesced, store-only pass
    uint32_t* jix = key == lds ? lds + n : lds;
    for (int i = i0; i < i0 + ipt; ++i) {
      const bool head = i < B && (i == 0 || (key[i] >> bbits) != (key[i - 1] >> bbits));
      jn += head;
      jix[i] = (uint32_t)(jn - 1) | (head ? 0x80000000u : 0u);     // segment index of position i; top bit = head
    }
    __syncthreads();
    for (int i = tid; i < B; i += T) {
      const uint32_t kk = key[i], jh = jix[i], j = jh & 0x7FFFFFFFu;
      a.perm[(size_t)f * stride + i] = (int32_t)(kk & bmask);
      if (a.segid != nullptr) a.segid[(size_t)f * stride + i] = (int32_t)j;
      if (jh & 0x80000000u) {
        const int row = roff + (int)(kk >> bbits);
        a.uniq_row[(size_t)f * stride + j] = row;
        a.seg_off[(size_t)f * (stride + 1) + j] = i;
        a.slot[row] = f * stride + (int)j;
      }
    }
  }
  if (tid == 0) {
    a.seg_off[(size_t)f * (stride + 1) + total] = B;
    a.nuniq[f] = total;
  }
  if (a.segid != nullptr && !small) {
    // long-segment lists of the two-stage scatter (> 16 entries: a helper group, > 256: a helper wave each), from the
    // segment starts gathered in the now idle key buffer.  List order is arrival order: it only assigns work.
    uint32_t* jix = key == lds ? lds + n : lds;
    uint32_t* hs = key;
    __syncthreads();
    for (int i = tid; i < B; i += T)
      if (jix[i] & 0x80000000u) hs[jix[i] & 0x7FFFFFFFu] = (uint32_t)i;
    if (tid < 2) wsum[tid] = 0;
    __syncthreads();
    const int nch = (B + 15) >> 4;
    int32_t* cnt = a.segid + (size_t)a.F * stride;
    int32_t* ll = cnt + 2 * a.F + (size_t)f * nch;
    for (int j = tid; j < total; j += T) {
      const int L = (j + 1 < total ? (int)hs[j + 1] : B) - (int)hs[j];
      if (L > 256) ll[nch - 1 - (int)atomicAdd(&wsum[1], 1u)] = j;
      else if (L > 16) ll[atomicAdd(&wsum[0], 1u)] = j;
    }
    __syncthreads();
    if (tid < 2) cnt[tid * a.F + f] = (int32_t)wsum[tid];
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Round 4: SEVERAL workgroups per field (stand-alone sort launches, 1 024 <= B <= 8 192).  The one-workgroup form above leaves
// 217 of 256 CUs idle and its critical path is the 17-bit fields' three 8-bit passes over all B keys (26 us at 4 096 x 39).
// Here field f is cut into G id ranges [lo, hi) of equal width; workgroup (f, g)
//   * reads the WHOLE id column (cheap: B ints), marks every id in an LDS presence bitmap (rows <= 131 072 -> 16 KB), counts
//     the keys below its range and compacts the keys inside it in ascending example order (stable);
//   * sorts ITS keys only (~B / G of them, id bits of the range only: two 8-bit passes at most for 2^16-row ranges);
//   * knows every output index without talking to the other workgroups: sorted position = keys below + local rank, unique
//     index j of id r = popcount of the bitmap below r (a prefix popcount per word + one masked popcount) -- heads, segment
//     ids and the slot map follow; a segment never crosses a range, so the long / huge classification is local too.
// No workgroup waits for another one (no spin, no co-residency requirement).  What the old form did by being alone:
//   * forgetting the previous step's rows: a workgroup owns the slot-map entries of its row range and nobody else reads or
//     writes them during the sort, so it sweeps that range of the map itself (entries >= 0 -> -1) while its ids are in flight
//     (the previous unique-row list cannot be used: other workgroups overwrite it at their own pace);
//   * the long / huge lists of the two-stage segment-sum: a workgroup reserves list space with ONE atomic per class on a
//     per-(job, field) running counter; the last workgroup of the field to arrive (a ticket) publishes the totals and zeroes
//     the three words for the next launch.  List order is arrival order, as before (it only assigns work).
// The result arrays are bit-identical to field_sort_block's except for the order of those two lists.
struct SortSplitScratch {
  int32_t w[RSX_ADAM_WINDOW_MAX][64][4];     // [job][field]{running long, running huge, ticket, -}: zero between launches
};

__device__ __forceinline__ void field_sort_split_block(const SortArgs& a, int f, int g, int G, int W, uint32_t* lds,
                                                        int32_t* scr /* [4] of (job, f) */) {
  if ((a.skip >> f) & 1ull) return;                     // (workgroup-uniform)
  const int tid = threadIdx.x, T = blockDim.x, lane = tid & 63, w = tid >> 6, nw = T >> 6;
  const int B = a.B, n = a.n, bbits = a.bbits, stride = a.stride;
  const int roff = a.row_off[f];
  const int rows = a.row_off[f + 1] - roff;
  const int Gw = rows < G ? rows : G;
  const int span = (rows + Gw - 1) / Gw;
  const int lo = g * span;
  if (lo >= rows) return;
  const int hi = lo + span < rows ? lo + span : rows;
  const bool last_range = hi == rows;
  const int Gf = (rows + span - 1) / span;              // ranges that hold rows (the others returned)
  uint32_t* keyA = lds;                                 // [n]
  uint32_t* keyB = lds + n;                             // [n]
  uint32_t* cnt = lds + 2 * n;                          // [nw * 256]
  uint32_t* bm = cnt + nw * 256;                        // [W]   presence bitmap of the field's ids
  uint32_t* pre = bm + W;                               // [W]   distinct ids in the words before this one
  uint32_t* wsum = pre + W;                             // [64]
  const int Wf = (rows + 31) >> 5;
  RSX_STAMP(0, blockIdx.x == 0 && blockIdx.y == 0);
  const int wpt = (Wf + T - 1) / T;                     // bitmap words per thread (contiguous)
  for (int i = tid; i < wpt * T; i += T)
    if (i < W) bm[i] = 0;
  // the column: thread t holds examples t*ipt .. t*ipt + ipt - 1 (ascending: the compaction below is stable)
  const int ipt = n / T;                                // 2, 4 or 8
  uint32_t v[8];
  {
    const int32_t* col = a.ids + f;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int b = tid * ipt + u;
      v[u] = (u < ipt) ? (uint32_t)col[(size_t)(b < B ? b : B - 1) * a.F] : 0u;
    }
  }
  // forget the previous sort's rows of MY range (see above): 16-byte loads, 8 in flight -- a 25 000-row range is ONE round trip
  {
    int32_t* sl = a.slot + roff;                          // (hipMalloc'ed base: 16-byte aligned where roff + r is a multiple of 4)
    const int mis = (int)((reinterpret_cast<uintptr_t>(sl + lo) >> 2) & 3u);
    const int a0 = lo + ((4 - mis) & 3) < hi ? lo + ((4 - mis) & 3) : hi;      // first 16-byte aligned row of the range
    const int a1 = a0 + ((hi - a0) & ~3);                                     // end of the aligned body
    if (tid < a0 - lo && sl[lo + tid] >= 0) sl[lo + tid] = -1;
    if (tid < hi - a1 && sl[a1 + tid] >= 0) sl[a1 + tid] = -1;
    int4* s4 = reinterpret_cast<int4*>(sl + a0);
    const int n4 = (a1 - a0) >> 2;
    for (int q = tid; q < n4; q += 8 * T) {
      int4 sv[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) sv[u] = s4[q + u * T < n4 ? q + u * T : n4 - 1];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        if (q + u * T < n4 && (sv[u].x >= 0 || sv[u].y >= 0 || sv[u].z >= 0 || sv[u].w >= 0))
          s4[q + u * T] = make_int4(-1, -1, -1, -1);
      }
    }
  }
  __syncthreads();                                      // bitmap zeroed
  RSX_STAMP_MAX(1, true);
  int mine = 0, below = 0;
  int fbits = 0;
  while ((1 << fbits) < rows) ++fbits;
  // Marking: an LDS atomic serialises over the lanes (and waves) that hit the same word.  A field of <= 256 rows has <= 8
  // words and every wave instruction would queue 64 lanes on them (measured: rows = 3 -> 27 us for the 4 096 marks of one
  // workgroup); there ONE lane per distinct id and wave instruction marks (ballot match over the id bits).
  const bool few = fbits <= 8;                           // (workgroup-uniform)
  const uint64_t ltm = (1ull << lane) - 1ull;
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    if (u < ipt) {                                       // (uniform)
      const bool ok = tid * ipt + u < B;
      bool mark = ok;
      if (few) {
        uint64_t m = __ballot(ok);
        m = ok ? m : ~m;
        m &= rsx_match_digit(v[u], fbits);
        mark = ok && (m & ltm) == 0ull;
      }
      if (mark) atomicOr(&bm[v[u] >> 5], 1u << (v[u] & 31u));
      if (ok) {
        mine += ((int)v[u] >= lo && (int)v[u] < hi) ? 1 : 0;
        below += (int)v[u] < lo ? 1 : 0;
      }
    }
  }
  __syncthreads();                                      // bitmap complete
  RSX_STAMP_MAX(2, true);
  // three block scans at once: my keys (exclusive -> compaction offset), keys below (total), distinct ids per word group
  int pc = 0;
  for (int k = 0; k < wpt; ++k) {
    const int i = tid * wpt + k;
    pc += i < Wf ? __popc(bm[i]) : 0;
  }
  int i_m = mine, i_b = below, i_p = pc;
#pragma unroll
  for (int d = 1; d < RSX_WAVE; d <<= 1) {
    const int om = __shfl_up(i_m, d), ob = __shfl_up(i_b, d), op = __shfl_up(i_p, d);
    if (lane >= d) { i_m += om; i_b += ob; i_p += op; }
  }
  if (lane == 63) { wsum[w] = (uint32_t)i_m; wsum[16 + w] = (uint32_t)i_b; wsum[32 + w] = (uint32_t)i_p; }
  __syncthreads();
  int off = i_m - mine, n_own = 0, n_below = 0, pbase = i_p - pc, total_u = 0;
  for (int ww = 0; ww < nw; ++ww) {
    const int vm = (int)wsum[ww], vb = (int)wsum[16 + ww], vp = (int)wsum[32 + ww];
    if (ww < w) { off += vm; pbase += vp; }
    n_own += vm; n_below += vb; total_u += vp;
  }
  {
    int run = pbase;
    for (int k = 0; k < wpt; ++k) {
      const int i = tid * wpt + k;
      if (i < Wf) { pre[i] = (uint32_t)run; run += __popc(bm[i]); }
    }
  }
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    if (u < ipt && tid * ipt + u < B && (int)v[u] >= lo && (int)v[u] < hi)
      keyA[off++] = ((v[u] - (uint32_t)lo) << bbits) | (uint32_t)(tid * ipt + u);
  }
  // keys per wave of the radix passes: a contiguous range, a multiple of 64; all-ones padding sorts last
  int ipw = ((n_own + nw - 1) / nw + 63) & ~63;
  if (ipw == 0) ipw = 64;
  for (int i = n_own + tid; i < nw * ipw; i += T) keyA[i] = 0xFFFFFFFFu;
  __syncthreads();
  RSX_STAMP_MAX(3, true);
  uint32_t* src = keyA;
  uint32_t* dst = keyB;
  {
    int idbits = 0;
    while ((1 << idbits) < hi - lo) ++idbits;
    const int npass = (idbits + 7) >> 3;
    const int width = npass ? (idbits + npass - 1) / npass : 0;
    const uint64_t lt = (1ull << lane) - 1ull;
    for (int sh = 0; sh < idbits; sh += width) {
      const int nb = idbits - sh < width ? idbits - sh : width;
      const uint32_t dmask = (1u << nb) - 1u;
      for (int i = tid; i < nw * 256; i += T) cnt[i] = 0;
      __syncthreads();
      for (int j = 0; j < ipw; j += 64) atomicAdd(&cnt[w * 256 + ((src[w * ipw + j + lane] >> (bbits + sh)) & dmask)], 1u);
      __syncthreads();
      // exclusive scan in (digit-major, wave-minor) order over the 256 * nw counters as ONE linear array: thread t owns the four
      // consecutive entries 4t .. 4t + 3 = digit 4t / nw, waves 4t % nw .. + 3 (nw is a multiple of 4; T = 64 nw threads cover it)
      {
        const int d0 = (4 * tid) / nw, w0 = (4 * tid) - d0 * nw;
        int c[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) c[q] = (int)cnt[(w0 + q) * 256 + d0];
        const int tot = (c[0] + c[1]) + (c[2] + c[3]);
        int incl = tot;
#pragma unroll
        for (int dd = 1; dd < RSX_WAVE; dd <<= 1) {
          const int o = __shfl_up(incl, dd);
          if (lane >= dd) incl += o;
        }
        if (lane == 63) wsum[w] = (uint32_t)incl;
        __syncthreads();
        int run = incl - tot;
        for (int ww = 0; ww < w; ++ww) run += (int)wsum[ww];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          cnt[(w0 + q) * 256 + d0] = (uint32_t)run;
          run += c[q];
        }
      }
      __syncthreads();
      for (int j = 0; j < ipw; j += 64) {
        const uint32_t k = src[w * ipw + j + lane];
        const uint32_t d = (k >> (bbits + sh)) & dmask;
        const uint64_t m = rsx_match_digit(d, nb);
        const uint32_t old = cnt[w * 256 + d];
        const int r = __popcll(m & lt);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        if (r == 0) cnt[w * 256 + d] = old + (uint32_t)__popcll(m);
        dst[old + r] = k;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      }
      __syncthreads();
      uint32_t* t = src;
      src = dst;
      dst = t;
    }
  }
  // emit: sorted position = n_below + i; unique index from the bitmap
  const uint32_t bmask = (1u << bbits) - 1u;
  auto rank_of = [&](const int id) -> int {             // distinct ids of the field below `id` (id < rows)
    return (int)pre[id >> 5] + __popc(bm[id >> 5] & ((1u << (id & 31)) - 1u));
  };
  const int jlo = rank_of(lo);
  const int jhi = last_range ? total_u : rank_of(hi);
  const int U = jhi - jlo;                              // my distinct ids
  // Segment starts, the long / huge classification and the list reservations first (LDS only + two atomics whose round trip
  // hides behind the stores below); then every global store; then the ticket.  No fence: the only cross-workgroup traffic
  // of this launch are the atomics on the three scratch words, and the lists are read by LATER launches.
  uint32_t* hs = dst;                                   // [U] local position of every segment start (the idle key buffer)
  uint32_t* lst = cnt;                                  // [<= n/17] long from the front, huge from the back of cnt's nw*256 words
  const int lcap = nw * 256;
  const bool lists = a.segid != nullptr;
  if (lists) {
    for (int i = tid; i < n_own; i += T) {
      const uint32_t kk = src[i];
      if (i == 0 || (src[i - 1] >> bbits) != (kk >> bbits)) hs[rank_of(lo + (int)(kk >> bbits)) - jlo] = (uint32_t)i;
    }
    if (tid == 0) { wsum[0] = 0; wsum[1] = 0; }
    __syncthreads();
    for (int jl = tid; jl < U; jl += T) {
      const int L = (jl + 1 < U ? (int)hs[jl + 1] : n_own) - (int)hs[jl];
      if (L > 256) lst[lcap - 1 - (int)atomicAdd(&wsum[1], 1u)] = (uint32_t)(jlo + jl);
      else if (L > 16) lst[atomicAdd(&wsum[0], 1u)] = (uint32_t)(jlo + jl);
    }
    __syncthreads();
    if (tid < 2) {                                      // lane 0: long, lane 1: huge -- both reservations in flight together
      const int cnum = (int)wsum[tid];
      wsum[2 + tid] = cnum ? (uint32_t)atomicAdd(&scr[tid], cnum) : 0u;
    }
  }
  RSX_STAMP_MAX(4, true);
  for (int i = tid; i < n_own; i += T) {
    const uint32_t kk = src[i];
    const int id = lo + (int)(kk >> bbits);
    const int pos = n_below + i;
    const int j = rank_of(id);
    a.perm[(size_t)f * stride + pos] = (int32_t)(kk & bmask);
    if (lists) a.segid[(size_t)f * stride + pos] = j;
    if (i == 0 || (src[i - 1] >> bbits) != (kk >> bbits)) {
      const int row = roff + id;
      a.uniq_row[(size_t)f * stride + j] = row;
      a.seg_off[(size_t)f * (stride + 1) + j] = pos;
      a.slot[row] = f * stride + j;
    }
  }
  if (last_range && tid == 0) {
    a.seg_off[(size_t)f * (stride + 1) + total_u] = B;
    a.nuniq[f] = total_u;
  }
  if (!lists) return;
  RSX_STAMP_MAX(5, true);
  __syncthreads();                                      // the reservations' results (and every wave's stores issued)
  const int nl = (int)wsum[0], nh = (int)wsum[1];
  const int nch = (B + 15) >> 4;
  int32_t* cn = a.segid + (size_t)a.F * stride;
  int32_t* ll = cn + 2 * a.F + (size_t)f * nch;
  const int bl = (int)wsum[2], bh = (int)wsum[3];
  RSX_STAMP_MAX(6, true);
  for (int k = tid; k < nl; k += T) ll[bl + k] = (int32_t)lst[k];
  for (int k = tid; k < nh; k += T) ll[nch - 1 - (bh + k)] = (int32_t)lst[lcap - 1 - k];
  if (tid == 0) {
    const int t = atomicAdd(&scr[2], 1);
    if (t == Gf - 1) {                                  // every range of the field has reserved: the totals are final
      cn[f] = atomicExch(&scr[0], 0);
      cn[a.F + f] = atomicExch(&scr[1], 0);
      atomicExch(&scr[2], 0);
    }
  }
  RSX_STAMP_MAX(7, true);
}

static inline int rsx_ceil_log2(int x) {
  int b = 0;
  while ((1 << b) < x) ++b;
  return b;
}

// Fills n / bbits for a launch with `threads` threads per workgroup; returns an rsx_status.
static inline int rsx_sort_args(SortArgs& a, int max_rows_per_field, int threads) {
  if (a.B > 16384) return RSX_EUNSUPPORTED;              // 2n keys must fit the 160 KB LDS of one workgroup
  if (a.segid != nullptr && a.B <= 512) return RSX_EINVAL;   // the two-stage workspace is for B > 512
  a.bbits = rsx_ceil_log2(a.B < 2 ? 2 : a.B);
  if (((uint64_t)(max_rows_per_field - 1) << a.bbits) >= (1ull << 32) - 1ull) return RSX_EUNSUPPORTED;
  int n = 128;
  while (n < a.B) n <<= 1;
  if (n < threads) n = threads;
  a.n = n;
  return RSX_OK;
}

// dynamic LDS bytes field_sort_block needs for a launch with `threads` threads per workgroup
static inline size_t rsx_sort_lds_bytes(const SortArgs& a, int threads) {
  if (a.n <= 512 && a.n <= threads) return ((size_t)a.n + 32) * sizeof(uint32_t);
  return ((size_t)2 * a.n + 32 + (size_t)(threads / 64) * 256) * sizeof(uint32_t);
}

// the split form: applicability and launch geometry (stand-alone launches only; RSX_SORT_SPLIT=0 switches it off for A/B runs)
static inline int rsx_sort_split_parts(const SortArgs& a, int max_rows_per_field) {
  // (measured, 39 Criteo fields: 1 024 keys 16.6 vs 12.7 us, 2 048: 23.8 vs 18.8 -- the one-workgroup form wins; 4 096: 24.6 vs 26.7,
  // 8 192: 41.3 vs 43.6.  Every workgroup pulls the whole [B, F] id matrix through its L1 to read one column -- ~4-7 us that no
  // split shortens; profiles/r04_g_sort_split_stamps.txt)
  if (a.n < 4096 || a.n > 8192 || max_rows_per_field > 131072 || a.F > 64) return 0;
  return a.n <= 4096 ? 4 : 8;
}
static inline int rsx_sort_split_words(int max_rows_per_field) { return (((max_rows_per_field + 31) >> 5) + 3) & ~3; }
static inline size_t rsx_sort_split_lds_bytes(const SortArgs& a, int threads, int W) {
  return ((size_t)2 * a.n + (size_t)(threads / 64) * 256 + 2 * (size_t)W + 64) * sizeof(uint32_t);
}

// host: a rsx_sort_job (the step's dedup sort carried by another launch as extra 256-thread workgroups) -> SortArgs;
// raises *lds to what field_sort_block needs
static inline int sort_job_args(const rsx_sort_job& j, SortArgs& out, size_t* lds) {
  if (!j.ids || !j.row_off || !j.perm || !j.seg_off || !j.uniq_row || !j.nuniq || !j.slot || j.B < 0 || j.F <= 0 ||
      j.stride < j.B || j.max_rows_per_field <= 0)
    return RSX_EINVAL;
  out = SortArgs{j.ids, j.row_off, j.perm, j.seg_off, j.uniq_row, j.nuniq, j.slot, j.segid, j.B, j.F, j.stride, 0, 0, j.skip_mask};
  const int rc = rsx_sort_args(out, j.max_rows_per_field, 256);
  if (rc != RSX_OK) return rc;
  const size_t need = rsx_sort_lds_bytes(out, 256);
  if (need > 64 * 1024) return RSX_EUNSUPPORTED;         // carrier launches keep the default 64 KB window (B <= 4096)
  if (need > *lds) *lds = need;
  return RSX_OK;
}

