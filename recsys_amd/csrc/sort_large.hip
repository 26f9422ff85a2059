// Per-field dedup sort for batches beyond one workgroup's LDS (B > 16 384; data-parallel steps see N*b examples):
// a stable LSD radix sort over the id bits through global memory, all F fields per launch.
//   transpose      ids[B,F] -> idsT[F,B]                                  (coalesced both ways through an LDS tile)
//   2 passes x { histogram per (field, 4096-key tile) ; scan per field (digit-major, tile-minor) ; stable scatter }
//                  digit widths ceil(idbits/2), floor(idbits/2) <= 9 bits; keys and example indices travel as two int32
//                  arrays (B up to 2^20 would not fit beside an 18-bit id in one word); the initial order IS ascending
//                  example index, stability keeps it -> entries end ordered by (id, b), TF's accumulation order
//   segments       head flags + scan per field -> the rsx_field_sort workspace (unique rows, offsets, slot map, and the
//                  two-stage segment-sum lists), multi-workgroup
// Ranking inside a wave uses ballots (stable, no atomics); every cross-wave / cross-tile order is a scan: deterministic.
// Replaces the same TF ops as rsx_field_sort (unique() in safe_embedding_lookup_sparse).
#include "rsx_common.h"
#include "sort_device.h"
RSX_STAMP_DECL

namespace {
constexpr int LS_TILE = 4096;   // keys per workgroup
constexpr int LS_T = 1024;      // threads (16 waves x 4 items of 64 keys each, wave-major order)
constexpr int LS_BINS = 512;

struct LargeSort {
  const int32_t* ids;       // [B, F]
  const int32_t* row_off;   // [F+1]
  int32_t* idsT;            // [F, stride]  transposed ids, then key buffer B of the ping-pong
  int32_t* keyA;            // [F, stride]
  int32_t* valA;            // [F, stride]
  int32_t* perm;            // [F, stride]  final example indices
  int32_t* hist;            // [F, LS_BINS, nT]
  int32_t* dtot;            // [2 passes, F, LS_BINS] digit totals (integer atomics: order-independent); NULL: not used (many
                            // fields).  All zero between calls: each pass's scatter launch clears what its scan consumed
  const int32_t* src0;      // pass-0 keys: idsT, or ids itself when F == 1 (the transpose is then the identity)
  int B, F, stride, nT;
  int fuse_scan;            // few tiles: every scatter workgroup derives its own offsets from the raw counts (no scan launch)
};

__device__ __forceinline__ void digit_split(const int32_t* row_off, int f, int pass, int& sh, int& nb) {
  const int rows = row_off[f + 1] - row_off[f];
  int idbits = 0;
  while ((1 << idbits) < rows) ++idbits;
  const int w0 = (idbits + 1) >> 1;
  sh = pass == 0 ? 0 : w0;
  nb = pass == 0 ? w0 : idbits - w0;
}

__global__ __launch_bounds__(1024) void ls_transpose_k(const LargeSort a) {
  __shared__ int32_t tile[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int b0 = blockIdx.x * 32, f0 = blockIdx.y * 32;
  if (b0 + ty < a.B && f0 + tx < a.F) tile[ty][tx] = a.ids[(size_t)(b0 + ty) * a.F + f0 + tx];
  __syncthreads();
  if (f0 + ty < a.F && b0 + tx < a.B) a.idsT[(size_t)(f0 + ty) * a.stride + b0 + tx] = tile[tx][ty];
}

// grid (nT, F).  pass 0 reads idsT; pass 1 reads keyA.
__global__ __launch_bounds__(LS_T) void ls_hist_k(const LargeSort a, int pass) {
  __shared__ uint32_t h[LS_BINS];
  const int f = blockIdx.y, t = blockIdx.x, tid = threadIdx.x;
  int sh, nb;
  digit_split(a.row_off, f, pass, sh, nb);
  const uint32_t mask = (1u << nb) - 1u;
  for (int i = tid; i < LS_BINS; i += LS_T) h[i] = 0;
  __syncthreads();
  const int32_t* src = (pass == 0 ? a.src0 : a.keyA) + (size_t)f * a.stride;
#pragma unroll
  for (int k = 0; k < LS_TILE / LS_T; ++k) {
    const int i = t * LS_TILE + k * LS_T + tid;
    if (i < a.B) atomicAdd(&h[((uint32_t)src[i] >> sh) & mask], 1u);
  }
  __syncthreads();
  for (int d = tid; d < LS_BINS; d += LS_T) {
    a.hist[((size_t)f * LS_BINS + d) * a.nT + t] = (int32_t)h[d];
    if (a.dtot != nullptr && h[d]) atomicAdd(&a.dtot[((size_t)pass * a.F + f) * LS_BINS + d], (int32_t)h[d]);
  }
}

// grid F * LS_BINS / 4, block 256: exclusive scan of hist[f] in (digit-major, tile-minor) order, in place.  One wave per
// (field, digit): its base is the sum of the smaller digits' totals (accumulated by ls_hist_k), its tiles are scanned 64 at a
// time with lane shuffles.  (One workgroup per field walking the tiles serially cost 16 us at F = 1.)
__global__ __launch_bounds__(256) void ls_scan_k(const LargeSort a, int pass) {
  const int wid = (blockIdx.x * 256 + threadIdx.x) >> 6, lane = threadIdx.x & 63;
  const int f = wid / LS_BINS, d = wid - f * LS_BINS;
  if (f >= a.F) return;
  const int32_t* tot = a.dtot + ((size_t)pass * a.F + f) * LS_BINS;
  int base = 0;
#pragma unroll
  for (int k = 0; k < LS_BINS / 64; ++k) {
    const int dd = lane + 64 * k;
    const int v = tot[dd];
    base += dd < d ? v : 0;
  }
#pragma unroll
  for (int m = 1; m < 64; m <<= 1) base += __shfl_xor(base, m);
  int32_t* h = a.hist + ((size_t)f * LS_BINS + d) * a.nT;
  for (int t0 = 0; t0 < a.nT; t0 += 64) {
    const int t = t0 + lane;
    const int c = t < a.nT ? h[t] : 0;
    int incl = c;
#pragma unroll
    for (int s_ = 1; s_ < 64; s_ <<= 1) {
      const int o = __shfl_up(incl, s_);
      if (lane >= s_) incl += o;
    }
    if (t < a.nT) h[t] = base + incl - c;
    base += __shfl(incl, 63);
  }
}

// grid F, block 512: the same scan with one workgroup per field, each thread walking its digit's tiles (many fields: the
// fields themselves fill the chip and the per-digit version's re-reads of the totals cost more than they save).
__global__ __launch_bounds__(LS_BINS) void ls_scan_field_k(const LargeSort a) {
  __shared__ int wsum[LS_BINS / 64];
  const int f = blockIdx.x, d = threadIdx.x, lane = d & 63, w = d >> 6;
  int32_t* h = a.hist + ((size_t)f * LS_BINS + d) * a.nT;
  int tot = 0;
  {
    int t = 0;
    for (; t + 8 <= a.nT; t += 8) {     // 8 loads in flight (one dependent L2 round trip per tile would dominate)
      int c[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) c[u] = h[t + u];
#pragma unroll
      for (int u = 0; u < 8; ++u) tot += c[u];
    }
    for (; t < a.nT; ++t) tot += h[t];
  }
  int incl = tot;
#pragma unroll
  for (int s = 1; s < 64; s <<= 1) {
    const int o = __shfl_up(incl, s);
    if (lane >= s) incl += o;
  }
  if (lane == 63) wsum[w] = incl;
  __syncthreads();
  int run = incl - tot;
  for (int ww = 0; ww < w; ++ww) run += wsum[ww];
  {
    int t = 0;
    for (; t + 8 <= a.nT; t += 8) {
      int c[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) c[u] = h[t + u];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        h[t + u] = run;
        run += c[u];
      }
    }
    for (; t < a.nT; ++t) {
      const int c = h[t];
      h[t] = run;
      run += c;
    }
  }
}

// grid (nT, F): stable scatter of one tile.  Position = global offset of (digit, tile) + keys of the same digit in
// earlier waves of the tile + rank among the equal-digit lanes below me.
__global__ __launch_bounds__(LS_T) void ls_scatter_k(const LargeSort a, int pass) {
  __shared__ uint32_t cnt[LS_T / 64][LS_BINS];      // 32 KB
  const int f = blockIdx.y, t = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  int sh, nb;
  digit_split(a.row_off, f, pass, sh, nb);
  const uint32_t mask = (1u << nb) - 1u;
  for (int i = tid; i < (LS_T / 64) * LS_BINS; i += LS_T) (&cnt[0][0])[i] = 0;
  __syncthreads();
  // Leave the digit totals clean for the next call once nobody reads them any more: with a scan launch that is now (this
  // pass's scan is done); with the scan fused into this kernel the pass-0 totals are cleared by the pass-1 launch and the
  // pass-1 totals by the segment kernels.
  if (a.dtot != nullptr && t == 0 && (!a.fuse_scan || pass == 1))
    for (int i = tid; i < LS_BINS; i += LS_T) a.dtot[((size_t)(a.fuse_scan ? 0 : pass) * a.F + f) * LS_BINS + i] = 0;
  const size_t fo = (size_t)f * a.stride;
  const int32_t* skey = (pass == 0 ? a.src0 : a.keyA) + fo;
  const int32_t* sval = a.valA + fo;                 // pass 1 only; pass 0: value = position
  int32_t* dkey = (pass == 0 ? a.keyA : a.idsT) + fo;
  int32_t* dval = (pass == 0 ? a.valA : a.perm) + fo;
  constexpr int IPW = LS_TILE / (LS_T / 64) / 64;    // items per wave-lane: 4
  uint32_t key[IPW];
  int val[IPW];
  bool ok[IPW];
#pragma unroll
  for (int k = 0; k < IPW; ++k) {
    const int i = t * LS_TILE + w * (IPW * 64) + k * 64 + lane;
    ok[k] = i < a.B;
    key[k] = ok[k] ? (uint32_t)skey[i] : 0xFFFFFFFFu;
    val[k] = pass == 0 ? i : (ok[k] ? sval[i] : 0);
    if (ok[k]) atomicAdd(&cnt[w][(key[k] >> sh) & mask], 1u);
  }
  __syncthreads();
  __shared__ int wtot[LS_BINS / 64];
  uint32_t run0 = 0;
  if (a.fuse_scan) {
    // offset of (digit, this tile) = keys of smaller digits (exclusive scan of the digit totals) + keys of this digit in
    // earlier tiles (raw counts: the scan launch is skipped)
    int tot = 0, before = 0;
    if (tid < LS_BINS) {
      tot = a.dtot[((size_t)pass * a.F + f) * LS_BINS + tid];
      const int32_t* h = a.hist + ((size_t)f * LS_BINS + tid) * a.nT;
      int tt = 0;
      for (; tt + 8 <= t; tt += 8) {
        int c[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) c[u] = h[tt + u];
#pragma unroll
        for (int u = 0; u < 8; ++u) before += c[u];
      }
      for (; tt < t; ++tt) before += h[tt];
    }
    int incl = tot;
#pragma unroll
    for (int s_ = 1; s_ < 64; s_ <<= 1) {
      const int o = __shfl_up(incl, s_);
      if (lane >= s_) incl += o;
    }
    if (tid < LS_BINS && lane == 63) wtot[w] = incl;
    __syncthreads();
    if (tid < LS_BINS) {
      int base = incl - tot;
      for (int ww = 0; ww < w; ++ww) base += wtot[ww];
      run0 = (uint32_t)(base + before);
    }
  } else if (tid < LS_BINS) {
    run0 = (uint32_t)a.hist[((size_t)f * LS_BINS + tid) * a.nT + t];
  }
  if (tid < LS_BINS) {
    uint32_t run = run0;
#pragma unroll
    for (int ww = 0; ww < LS_T / 64; ++ww) {
      const uint32_t c = cnt[ww][tid];
      cnt[ww][tid] = run;
      run += c;
    }
  }
  __syncthreads();
  const uint64_t lt = (1ull << lane) - 1ull;
#pragma unroll
  for (int k = 0; k < IPW; ++k) {
    // padding lanes (past B) take digit LS_BINS-1 with a ballot of their own: they never disturb real ranks
    const uint32_t d = (key[k] >> sh) & mask;
    uint64_t m = __ballot(ok[k]);
    m = ok[k] ? m : ~m;
    m &= rsx_match_digit(d, nb);
    const uint32_t old = cnt[w][d];
    const int r = __popcll(m & lt);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    if (ok[k] && r == 0) cnt[w][d] = old + (uint32_t)__popcll(m);
    if (ok[k]) {
      dkey[old + r] = (int32_t)key[k];
      dval[old + r] = val[k];
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  }
}

// ---- segments of the sorted keys (keys = idsT after pass 1), per field, 1024-key blocks ----------------------------
constexpr int SG_T = 256, SG_IPT = 4, SG_BLK = SG_T * SG_IPT;

struct LargeSeg {
  const int32_t* keys;      // [F, stride] sorted ids
  const int32_t* row_off;
  int32_t* seg_off;         // [F, stride+1]
  int32_t* uniq_row;        // [F, stride]
  int32_t* nuniq;           // [F]
  int32_t* slot;            // [R]
  int32_t* segid;           // nullable: [F*stride | 2F counts | F*nch lists]
  int32_t* blk_cnt;         // [F, nblk]
  int32_t* dtot1;           // nullable: the sort's pass-1 digit totals [F, LS_BINS], cleared here when its scan was fused
  int B, F, stride, nblk;
};

// grid (nblk, F)
__global__ __launch_bounds__(SG_T) void ls_heads_k(const LargeSeg a) {
  __shared__ int wsum[SG_T / 64];
  const int f = blockIdx.y, tid = threadIdx.x;
  const int prev = a.nuniq[f];
  for (int jj = blockIdx.x * SG_T + tid; jj < prev; jj += gridDim.x * SG_T) a.slot[a.uniq_row[(size_t)f * a.stride + jj]] = -1;
  if (a.segid != nullptr && blockIdx.x == 0 && tid < 2) a.segid[(size_t)a.F * a.stride + tid * a.F + f] = 0;
  if (a.dtot1 != nullptr && blockIdx.x == 0)
    for (int i = tid; i < LS_BINS; i += SG_T) a.dtot1[(size_t)f * LS_BINS + i] = 0;
  const int32_t* keys = a.keys + (size_t)f * a.stride;
  const int i0 = blockIdx.x * SG_BLK + tid * SG_IPT;
  int cnt = 0;
#pragma unroll
  for (int k = 0; k < SG_IPT; ++k) {
    const int i = i0 + k;
    if (i < a.B && (i == 0 || keys[i] != keys[i - 1])) ++cnt;
  }
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) cnt += __shfl_xor(cnt, d);
  if ((tid & 63) == 0) wsum[tid >> 6] = cnt;
  __syncthreads();
  if (tid == 0) a.blk_cnt[(size_t)f * a.nblk + blockIdx.x] = (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]);
}

__global__ __launch_bounds__(SG_T) void ls_emit_k(const LargeSeg a) {
  __shared__ int wsum[SG_T / 64];
  __shared__ int red[SG_T / 64];
  const int f = blockIdx.y, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int32_t* keys = a.keys + (size_t)f * a.stride;
  const int roff = a.row_off[f];
  int before = 0;
  for (int b = tid; b < (int)blockIdx.x; b += SG_T) before += a.blk_cnt[(size_t)f * a.nblk + b];
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) before += __shfl_xor(before, d);
  if (lane == 0) red[w] = before;
  const int i0 = blockIdx.x * SG_BLK + tid * SG_IPT;
  int key[SG_IPT], pk = 0;
  bool head[SG_IPT];
  int cnt = 0;
  if (i0 > 0 && i0 < a.B) pk = keys[i0 - 1];
#pragma unroll
  for (int k = 0; k < SG_IPT; ++k) {
    const int i = i0 + k;
    key[k] = i < a.B ? keys[i] : 0;
    head[k] = i < a.B && (i == 0 || key[k] != (k == 0 ? pk : key[k - 1]));
    cnt += head[k];
  }
  int incl = cnt;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int o = __shfl_up(incl, d);
    if (lane >= d) incl += o;
  }
  if (lane == 63) wsum[w] = incl;
  __syncthreads();
  int jn = (red[0] + red[1]) + (red[2] + red[3]) + incl - cnt;
  for (int ww = 0; ww < w; ++ww) jn += wsum[ww];
#pragma unroll
  for (int k = 0; k < SG_IPT; ++k) {
    const int i = i0 + k;
    if (i >= a.B) break;
    if (head[k]) {
      const int row = roff + key[k];
      a.uniq_row[(size_t)f * a.stride + jn] = row;
      a.seg_off[(size_t)f * (a.stride + 1) + jn] = i;
      a.slot[row] = f * a.stride + jn;
      ++jn;
    }
    if (a.segid != nullptr) a.segid[(size_t)f * a.stride + i] = jn - 1;
  }
  if (i0 + SG_IPT >= a.B && i0 < a.B) {   // the thread holding the last key: jn is now the field's total
    a.seg_off[(size_t)f * (a.stride + 1) + jn] = a.B;
    a.nuniq[f] = jn;
  }
}

// grid (ceil(B/256), F): long (> 16 entries) segments from the front of the field's list, huge (> 256) from the back
__global__ __launch_bounds__(256) void ls_long_lists_k(const LargeSeg a) {
  const int f = blockIdx.y, j = blockIdx.x * 256 + threadIdx.x, lane = threadIdx.x & 63;
  const int U = a.nuniq[f];
  const int32_t* so = a.seg_off + (size_t)f * (a.stride + 1);
  const int L = j < U ? so[j + 1] - so[j] : 0;
  const bool lg = L > 16 && L <= 256, hg = L > 256;
  const uint64_t ml = __ballot(lg), mh = __ballot(hg);
  const int nch = (a.B + 15) >> 4;
  int32_t* cnt = a.segid + (size_t)a.F * a.stride;
  int32_t* ll = cnt + 2 * a.F + (size_t)f * nch;
  int bl = 0, bh = 0;
  if (lane == 0 && ml) bl = atomicAdd(cnt + f, __popcll(ml));
  if (lane == 0 && mh) bh = atomicAdd(cnt + a.F + f, __popcll(mh));
  bl = __shfl(bl, 0);
  bh = __shfl(bh, 0);
  const uint64_t lt = (1ull << lane) - 1ull;
  if (lg) ll[bl + __popcll(ml & lt)] = j;
  if (hg) ll[nch - 1 - (bh + __popcll(mh & lt))] = j;
}
}  // namespace

extern "C" size_t rsx_field_sort_large_workspace_ints(int B, int F, int stride) {
  (void)B;
  const size_t nT = ((size_t)stride + LS_TILE - 1) / LS_TILE, nblk = ((size_t)stride + SG_BLK - 1) / SG_BLK;
  return (size_t)3 * F * stride + (size_t)F * LS_BINS * nT + (size_t)F * nblk + (size_t)2 * F * LS_BINS;
}

static int field_sort_large_impl(const int32_t* ids, int32_t* idsT_caller, const int32_t* row_off, int32_t* perm,
                                 int32_t* seg_off, int32_t* uniq_row, int32_t* nuniq, int32_t* slot, int32_t* segid,
                                 int32_t* workspace, int max_rows_per_field, int B, int F, int stride, rsx_stream_t stream) {
  if (idsT_caller != nullptr) ids = idsT_caller;
  if (!ids || !row_off || !perm || !seg_off || !uniq_row || !nuniq || !slot || !workspace || B <= 0 || F <= 0 ||
      stride < B || max_rows_per_field <= 0)
    return RSX_EINVAL;
  if (max_rows_per_field > (1 << 18) || B > (1 << 24)) return RSX_EUNSUPPORTED;   // two 9-bit digits; 24-bit lists
  hipStream_t st = rsx_s(stream);
  LargeSort a;
  a.ids = ids; a.row_off = row_off; a.perm = perm;
  a.B = B; a.F = F; a.stride = stride; a.nT = (B + LS_TILE - 1) / LS_TILE;
  // keys already field-major ([F, stride], rsx_field_sort_large_t): the caller's buffer takes the place of the transposed
  // copy -- no transpose launch -- and, like it, ends up holding the sorted keys
  a.idsT = idsT_caller != nullptr ? idsT_caller : workspace;
  a.keyA = workspace + (size_t)F * stride;
  a.valA = workspace + (size_t)2 * F * stride;
  a.hist = workspace + (size_t)3 * F * stride;
  const size_t nT_cap = ((size_t)stride + LS_TILE - 1) / LS_TILE;
  a.dtot = F >= 16 ? nullptr : a.hist + (size_t)F * LS_BINS * nT_cap + (size_t)F * (((size_t)stride + SG_BLK - 1) / SG_BLK);
  a.src0 = F == 1 ? ids : a.idsT;
  a.fuse_scan = a.dtot != nullptr && a.nT <= 64;
  if (idsT_caller != nullptr) a.src0 = idsT_caller;
  else if (F > 1) RSX_LAUNCH(ls_transpose_k, dim3((B + 31) / 32, (F + 31) / 32), dim3(1024), 0, st, a);
  for (int pass = 0; pass < 2; ++pass) {
    RSX_LAUNCH(ls_hist_k, dim3(a.nT, F), dim3(LS_T), 0, st, a, pass);
    if (a.fuse_scan) {}      // the scatter workgroups derive their offsets themselves
    else if (F >= 16) RSX_LAUNCH(ls_scan_field_k, dim3(F), dim3(LS_BINS), 0, st, a);
    else RSX_LAUNCH(ls_scan_k, dim3((unsigned)((F * LS_BINS + 3) / 4)), dim3(256), 0, st, a, pass);
    RSX_LAUNCH(ls_scatter_k, dim3(a.nT, F), dim3(LS_T), 0, st, a, pass);
  }
  RSX_CHECK_LAUNCH();
  LargeSeg g;
  g.keys = a.idsT; g.row_off = row_off; g.seg_off = seg_off; g.uniq_row = uniq_row; g.nuniq = nuniq; g.slot = slot;
  g.segid = segid;
  g.dtot1 = a.fuse_scan ? a.dtot + (size_t)F * LS_BINS : nullptr;
  g.blk_cnt = a.hist + (size_t)F * LS_BINS * nT_cap;
  // (a.dtot sits after blk_cnt's F * nblk ints)
  g.B = B; g.F = F; g.stride = stride; g.nblk = (B + SG_BLK - 1) / SG_BLK;
  RSX_LAUNCH(ls_heads_k, dim3(g.nblk, F), dim3(SG_T), 0, st, g);
  RSX_LAUNCH(ls_emit_k, dim3(g.nblk, F), dim3(SG_T), 0, st, g);
  if (segid != nullptr) RSX_LAUNCH(ls_long_lists_k, dim3((B + 255) / 256, F), dim3(256), 0, st, g);
  RSX_CHECK_LAUNCH();
  return RSX_OK;
}

extern "C" int rsx_field_sort_large(const int32_t* ids, const int32_t* row_off, int32_t* perm, int32_t* seg_off,
                                    int32_t* uniq_row, int32_t* nuniq, int32_t* slot, int32_t* segid, int32_t* workspace,
                                    int max_rows_per_field, int B, int F, int stride, rsx_stream_t stream) {
  return field_sort_large_impl(ids, nullptr, row_off, perm, seg_off, uniq_row, nuniq, slot, segid, workspace,
                               max_rows_per_field, B, F, stride, stream);
}

extern "C" int rsx_field_sort_large_t(int32_t* ids_t, const int32_t* row_off, int32_t* perm, int32_t* seg_off,
                                      int32_t* uniq_row, int32_t* nuniq, int32_t* slot, int32_t* segid, int32_t* workspace,
                                      int max_rows_per_field, int B, int F, int stride, rsx_stream_t stream) {
  if (!ids_t) return RSX_EINVAL;
  return field_sort_large_impl(nullptr, ids_t, row_off, perm, seg_off, uniq_row, nuniq, slot, segid, workspace,
                               max_rows_per_field, B, F, stride, stream);
}
