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

