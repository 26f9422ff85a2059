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
  int B, F, stride, n, bbits;
};

// One workgroup (any blockDim that is a multiple of 64 and <= n/2 ... n) sorts field f.  key = (id << bbits) | b is
// unique, so the (unstable) bitonic network yields entries ordered by id, then by ascending example index -- the order
// TF's CPU unsorted_segment_sum accumulates in.  n = padded power of two (>= blockDim).  lds: n + 32 words.
__device__ __forceinline__ void field_sort_block(const SortArgs& a, int f, uint32_t* lds) {
  uint32_t* key = lds;            // [n]
  uint32_t* wsum = lds + a.n;     // [32]
  const int tid = threadIdx.x, T = blockDim.x;
  const int B = a.B, n = a.n, bbits = a.bbits, stride = a.stride;
  const int roff = a.row_off[f];
  // forget the previous step's rows of this field (they may differ from this step's)
  const int prev = a.nuniq[f];
  for (int jj = tid; jj < prev; jj += T) a.slot[a.uniq_row[(size_t)f * stride + jj]] = -1;
  for (int i = tid; i < n; i += T)
    key[i] = i < B ? (((uint32_t)a.ids[(size_t)i * a.F + f] << bbits) | (uint32_t)i) : 0xFFFFFFFFu;
  __syncthreads();
  if (n <= 512 && n <= T) {
    // small batches: rank sort.  Keys are unique, so rank = #{keys smaller than mine}; every thread scans the n keys
    // with broadcast LDS reads (4 per ds_read_b128) -- n/4 iterations, NO barriers, vs 36+ barrier-separated bitonic
    // stages.  n <= 512 keeps the O(n^2) compare count below the bitonic latency.
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
    for (int k = 2; k <= n; k <<= 1) {
      for (int jst = k >> 1; jst > 0; jst >>= 1) {
        for (int t = tid; t < (n >> 1); t += T) {
          const int i = ((t & ~(jst - 1)) << 1) | (t & (jst - 1));
          const int l = i | jst;
          const uint32_t x = key[i], c = key[l];
          const bool up = (i & k) == 0;
          if ((x > c) == up) {
            key[i] = c;
            key[l] = x;
          }
        }
        __syncthreads();
      }
    }
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
  }
  if (tid == 0) {
    a.seg_off[(size_t)f * (stride + 1) + total] = B;
    a.nuniq[f] = total;
  }
}

static inline int rsx_ceil_log2(int x) {
  int b = 0;
  while ((1 << b) < x) ++b;
  return b;
}

// Fills n / bbits for a launch with `threads` threads per workgroup; returns an rsx_status.
static inline int rsx_sort_args(SortArgs& a, int max_rows_per_field, int threads) {
  if (a.B > 16384) return RSX_EUNSUPPORTED;
  a.bbits = rsx_ceil_log2(a.B < 2 ? 2 : a.B);
  if (((uint64_t)(max_rows_per_field - 1) << a.bbits) >= (1ull << 32) - 1ull) return RSX_EUNSUPPORTED;
  int n = 128;
  while (n < a.B) n <<= 1;
  if (n < threads) n = threads;
  a.n = n;
  return RSX_OK;
}
