// Data-parallel exchange of per-rank UNIQUE-ROW lists (round 5).
//
// tf.distribute.MirroredStrategy (fm/fm.py:184-194, deepfm/readme.md:24) hands the optimizer the replicas' IndexedSlices of
// every embedding variable -- each replica's gradient is already one (row, sum) pair per row it touched -- and the optimizer
// de-duplicates the concatenation.  Rounds 1-4 exchanged the PRE-dedup per-example block and
// every rank re-sorted and re-scattered the global batch of N b examples: per-rank work and bytes grew with N.  Here every
// rank keeps what it would do alone -- the dedup sort and the sorted segment-sum of ITS batch -- and what crosses xGMI is
//   ids phase (depends on the batch ids only; once per optimizer window):   keys = [nuniq[F] | unique rows packed at goff]
//   after backward:                                                         [dense gradients | G[capT, D] | (G2) | gw1[capT]]
// with cap_f = goff[f+1] - goff[f] = min(batch, rows of field f) rows per field -- a STATIC bound on the unique rows of a
// field (4 918 rows per rank at Criteo-39 batch 256 against 9 984 per-example entries; 57 809 against 159 744 at 4 096), so
// every collective has a fixed size and the step stays graph-capturable without a host round trip for counts.
//   uniq_pack_k    local sort outputs -> the key block (one launch for the k batches of an optimizer window)
//   uniq_merge_k   the N ranks' key blocks -> the GLOBAL unique-row list per field, its slot map, and for every rank the
//                  position src[r][f, j] of global unique row j in rank r's list (or -1): each rank contributes at most ONE
//                  entry per row, so the "segment" of a row is <= N looked-up rows -- no global sort, no long segments.
//                  Presence bitmap of the field's rows in LDS + prefix popcounts: unique index of row x = set bits below x.
//   merged_adam_k  the optimizer launch: gradient of global unique row j = sum over ranks r = 0 .. N-1 IN RANK ORDER of
//                  G_r[goff[f] + src[r][f, j]] (deterministic, bit-identical on every replica), then the touched-row half of
//                  TF-1's Adam exactly as segsum_adam_k applies it; same riders (dense segments with the replica sum, the lazy
//                  window pass, a cold slice) through the shared HotAdam state (hot_adam_device.h).
// Per-rank work after the exchange: N lookups per GLOBAL unique row.  Against the single-process sum over the global batch the
// per-rank partial sums are another association of the same terms (fp32 rounding, 2e-6 in the tests).
#include "rsx_common.h"
#include "adam_device.h"
#include "hot_adam_device.h"

RSX_STAMP_DECL

constexpr int UX_MAX_JOBS = RSX_ADAM_WINDOW_MAX;
constexpr int UX_MAX_RANKS = RSX_UNIQ_MAX_RANKS;

// ---------------------------------------------------------------- pack -------------------------------------------------
constexpr int UX_PACK_CHUNK = 2048;
struct UxPack {
  const int32_t* uniq_row[UX_MAX_JOBS];
  const int32_t* nuniq[UX_MAX_JOBS];
  int32_t* keys[UX_MAX_JOBS];
  const int32_t* goff;
  const int32_t* row_off;
  int F, stride, P;
};

// rows of field f owned by one of the P row-range PARTS of the merge (a multiple of 32: parts own whole bitmap words)
__host__ __device__ __forceinline__ int ux_rows_per_part(int rows, int P) { return ((rows + P - 1) / P + 31) & ~31; }

// key block = [nuniq[F] | rstart[F][P + 1] | rows packed at goff]: rstart[f][p] = first position of the (ascending) list whose
// row lies in part p or above (rstart[f][P] = nuniq[f]) -- the rank that owns the list finds these boundaries for free while it
// copies it, so that the merge's part workgroups go straight to their sub-ranges of every rank's list.
__global__ __launch_bounds__(256) void uniq_pack_k(const UxPack a) {
  const int f = blockIdx.x, job = blockIdx.y, tid = threadIdx.x;
  const int32_t* ur = a.uniq_row[0];
  const int32_t* nq = a.nuniq[0];
  int32_t* keys = a.keys[0];
  constexpr int CH = UX_PACK_CHUNK;               // list positions per workgroup (grid z): din.py's item list holds ~45 000
#pragma unroll
  for (int k = 1; k < UX_MAX_JOBS; ++k) {        // (a dynamically indexed by-value table goes to scratch: unrolled selection)
    if (k == job) {
      ur = a.uniq_row[k];
      nq = a.nuniq[k];
      keys = a.keys[k];
    }
  }
  const int g0 = a.goff[f], cap = a.goff[f + 1] - g0;
  const int nu0 = nq[f];
  const int nu = nu0 < cap ? nu0 : cap;
  if (tid == 0 && blockIdx.z == 0) keys[f] = nu;
  ur += (size_t)f * a.stride;
  const int row0 = a.row_off[f];
  const int rpp = ux_rows_per_part(a.row_off[f + 1] - row0, a.P);
  int32_t* rs = keys + a.F + f * (a.P + 1);
  int32_t* out = keys + a.F + a.F * (a.P + 1) + g0;
  const int jend = (int)(blockIdx.z + 1) * CH < cap + 1 ? (int)(blockIdx.z + 1) * CH : cap + 1;
  for (int j = (int)blockIdx.z * CH + tid; j < jend; j += 256) {
    const int32_t r = ur[j < nu ? j : 0];
    if (j < cap) out[j] = j < nu ? r : -1;
    if (j <= nu) {                                // boundaries crossed between entries j - 1 and j (j == nu: the list's end)
      const int pj = j < nu ? (r - row0) / rpp : a.P;
      const int pp = j > 0 ? (ur[j - 1] - row0) / rpp : -1;
      for (int q = pp + 1; q <= pj; ++q) rs[q] = j;
    }
  }
}

extern "C" int rsx_uniq_pack(const rsx_uniq_pack_job* jobs_h, int njobs, const int32_t* goff, const int32_t* row_off, int F,
                             int stride, int parts, int max_cap, rsx_stream_t stream) {
  if (max_cap < 1) return RSX_EINVAL;
  if (!jobs_h || njobs < 1 || njobs > UX_MAX_JOBS || !goff || !row_off || F <= 0 || F > 64 || stride <= 0 || parts < 1 ||
      parts > RSX_UNIQ_MAX_PARTS)
    return RSX_EINVAL;
  UxPack a;
  for (int k = 0; k < UX_MAX_JOBS; ++k) {
    const rsx_uniq_pack_job& j = jobs_h[k < njobs ? k : 0];
    if (!j.uniq_row || !j.nuniq || !j.keys) return RSX_EINVAL;
    a.uniq_row[k] = j.uniq_row; a.nuniq[k] = j.nuniq; a.keys[k] = j.keys;
  }
  a.goff = goff; a.row_off = row_off; a.F = F; a.stride = stride; a.P = parts;
  RSX_LAUNCH(uniq_pack_k, dim3((unsigned)F, (unsigned)njobs, (unsigned)((max_cap + 1 + UX_PACK_CHUNK - 1) / UX_PACK_CHUNK)),
             dim3(256), 0, rsx_s(stream), a);
  RSX_CHECK_LAUNCH();
  return RSX_OK;
}

// ---------------------------------------------------------------- merge ------------------------------------------------
struct UxMerge {
  const int32_t* keys;          // rank r, job k: keys + r * rank_stride + k * job_stride = [F counts | capT rows at goff]
  long long rank_stride;
  int job_stride;
  int32_t* uniq_row[UX_MAX_JOBS];
  int32_t* nuniq[UX_MAX_JOBS];
  int32_t* slot[UX_MAX_JOBS];
  int32_t* src[UX_MAX_JOBS];    // [N][F * stride]
  const int32_t* goff;
  const int32_t* row_off;
  int32_t* cnt;                 // [jobs][F * P]: distinct rows per (field, part) -- written by the count pass (P > 1)
  int F, N, stride, wmax;       // wmax: bitmap words of the largest field (or part: the LDS layout)
  int P;                        // row-range parts per field (1: uniq_merge_k alone; > 1: uniq_merge_count_k + uniq_merge_part_k)
};

// Workgroup (f, r, job): builds the union bitmap of field f over all N lists (every workgroup of the field does -- N x the
// marking work, no cross-workgroup hand-off), then writes ITS rank's column of src; the rank-0 workgroup also publishes the
// field's global list, clears the previous step's slot entries of this workspace and writes the new ones.
template <int T>
__global__ __launch_bounds__(T) void uniq_merge_k(const UxMerge a) {
  extern __shared__ uint32_t ux_lds[];
  uint32_t* bm = ux_lds;                 // [wmax]  presence bits of the field's rows
  uint32_t* pre = ux_lds + a.wmax;       // [wmax]  set bits below word w
  uint32_t* wsum = pre + a.wmax;         // [T / 64] wave totals of the scan
  const int f = blockIdx.x, r = blockIdx.y, job = blockIdx.z, tid = threadIdx.x;
  int32_t* ur = a.uniq_row[0];
  int32_t* nq = a.nuniq[0];
  int32_t* slot = a.slot[0];
  int32_t* src = a.src[0];
#pragma unroll
  for (int k = 1; k < UX_MAX_JOBS; ++k) {
    if (k == job) {
      ur = a.uniq_row[k];
      nq = a.nuniq[k];
      slot = a.slot[k];
      src = a.src[k];
    }
  }
  const int row0 = a.row_off[f];
  const int rows = a.row_off[f + 1] - row0;
  const int W = (rows + 31) >> 5;
  const int g0 = a.goff[f];
  const int32_t* kb = a.keys + (size_t)job * a.job_stride;
  for (int w = tid; w < W; w += T) bm[w] = 0u;
  // the N lists as ONE index space (entry e of the concatenation): every load of the marking pass is independent of the others
  // -- a loop over the ranks around it made 8 dependent round trips of it
  int cum[UX_MAX_RANKS + 1];
  cum[0] = 0;
#pragma unroll
  for (int rr = 0; rr < UX_MAX_RANKS; ++rr) cum[rr + 1] = cum[rr] + (rr < a.N ? kb[(size_t)rr * a.rank_stride + f] : 0);
  __syncthreads();
  const int roff = a.F + a.F * (a.P + 1) + g0;     // this field's rows inside a key block
  for (int e = tid; e < cum[UX_MAX_RANKS]; e += T) {
    int rr = 0;
#pragma unroll
    for (int k = 1; k < UX_MAX_RANKS; ++k) rr += e >= cum[k] ? 1 : 0;
    int base = 0;
#pragma unroll
    for (int k = 1; k < UX_MAX_RANKS; ++k) base = k == rr ? cum[k] : base;
    const uint32_t x = (uint32_t)(kb[(size_t)rr * a.rank_stride + roff + (e - base)] - row0);
    atomicOr(&bm[x >> 5], 1u << (x & 31u));
  }
  __syncthreads();
  // exclusive prefix of the words' popcounts: a contiguous run of words per thread, then a block scan of the run totals
  const int per = (W + T - 1) / T;
  const int w0 = tid * per, w1 = w0 + per < W ? w0 + per : W;
  uint32_t run = 0;
  for (int w = w0; w < w1; ++w) run += (uint32_t)__popc(bm[w]);
  uint32_t incl = run;
  const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
  for (int d = 1; d < RSX_WAVE; d <<= 1) {
    const uint32_t o = __shfl_up(incl, d);
    if (lane >= d) incl += o;
  }
  if (lane == RSX_WAVE - 1) wsum[wave] = incl;
  __syncthreads();
  uint32_t base = 0, total = 0;
#pragma unroll
  for (int k = 0; k < T / 64; ++k) {
    const uint32_t s = wsum[k];
    base += k < wave ? s : 0u;
    total += s;
  }
  uint32_t p = base + incl - run;
  for (int w = w0; w < w1; ++w) {
    pre[w] = p;
    p += (uint32_t)__popc(bm[w]);
  }
  __syncthreads();
  // this rank's column: -1 everywhere, then the position of each of its rows
  int32_t* sr = src + ((size_t)r * a.F + f) * a.stride;
  for (uint32_t j = tid; j < total; j += T) sr[j] = -1;
  __syncthreads();
  {
    const int32_t* kr = kb + (size_t)r * a.rank_stride;
    const int nu = kr[f];
    const int32_t* xr = kr + roff;
    for (int i = tid; i < nu; i += T) {
      const uint32_t x = (uint32_t)(xr[i] - row0);
      const uint32_t g = pre[x >> 5] + (uint32_t)__popc(bm[x >> 5] & ((1u << (x & 31u)) - 1u));
      sr[g] = i;
    }
  }
  if (r != 0) return;          // (block-uniform)
  // the field's global list + slot map (this workspace's previous entries first: the contract of rsx_field_sort)
  int32_t* uf = ur + (size_t)f * a.stride;
  const int prev = nq[f];
  for (int j = tid; j < prev; j += T) slot[uf[j]] = -1;
  __syncthreads();
  for (int w = tid; w < W; w += T) {
    uint32_t bits = bm[w];
    uint32_t g = pre[w];
    while (bits) {
      const int b = __ffs((int)bits) - 1;
      bits &= bits - 1u;
      const int row = row0 + (w << 5) + b;
      uf[g] = row;
      slot[row] = f * a.stride + (int)g;
      ++g;
    }
  }
  if (tid == 0) nq[f] = (int)total;
}

// ---- the merge with P row-range parts per field (long lists: dcn.py at 8 x 4 096 -- 32 768 entries per hashed field -- and
// din.py's item table: 8 x ~45 000 entries in ONE field, which a single workgroup per (field, rank) marks in ~100 us) ---------
// Part p of field f owns rows [p rpp, (p + 1) rpp) and finds its entries of every rank's list through rstart (uniq_pack_k).
//   uniq_merge_count_k   workgroup (f, p, job): marks its range from all N lists, counts the distinct rows -> cnt[job][f P + p];
//                        also resets this workspace's PREVIOUS slot entries (its share of the old list): the emit pass of another
//                        part may write a new entry for the same row, so the reset cannot live in the emit launch
//   uniq_merge_part_k    workgroup (f P + p, r, job): marks again, prefix of its range + the counts of the parts below = global
//                        unique index; writes rank r's column of src for its range; r == 0 publishes the range's rows + slots
template <int T>
__device__ __forceinline__ uint32_t ux_mark_part(const UxMerge& a, const int32_t* kb, const int f, const int p, const int row0,
                                                 const int rlo, uint32_t* bm, const int W) {
  const int tid = threadIdx.x;
  for (int w = tid; w < W; w += T) bm[w] = 0u;
  int cum[UX_MAX_RANKS + 1], lo[UX_MAX_RANKS];
  cum[0] = 0;
#pragma unroll
  for (int rr = 0; rr < UX_MAX_RANKS; ++rr) {
    const int32_t* rs = kb + (size_t)(rr < a.N ? rr : 0) * a.rank_stride + a.F + f * (a.P + 1) + p;
    const int b0 = rs[0], b1 = rs[1];
    lo[rr] = b0;
    cum[rr + 1] = cum[rr] + (rr < a.N ? b1 - b0 : 0);
  }
  __syncthreads();
  const int roff = a.F + a.F * (a.P + 1) + a.goff[f];
  for (int e = tid; e < cum[UX_MAX_RANKS]; e += T) {
    int rr = 0;
#pragma unroll
    for (int k = 1; k < UX_MAX_RANKS; ++k) rr += e >= cum[k] ? 1 : 0;
    int base = 0, l0 = lo[0];
#pragma unroll
    for (int k = 1; k < UX_MAX_RANKS; ++k) {
      base = k == rr ? cum[k] : base;
      l0 = k == rr ? lo[k] : l0;
    }
    const uint32_t x = (uint32_t)(kb[(size_t)rr * a.rank_stride + roff + l0 + (e - base)] - row0 - rlo);
    atomicOr(&bm[x >> 5], 1u << (x & 31u));
  }
  __syncthreads();
  return (uint32_t)cum[UX_MAX_RANKS];
}

template <int T>
__global__ __launch_bounds__(T) void uniq_merge_count_k(const UxMerge a) {
  extern __shared__ uint32_t ux_lds[];
  uint32_t* bm = ux_lds;
  uint32_t* wsum = ux_lds + a.wmax;
  const int f = blockIdx.x, p = blockIdx.y, job = blockIdx.z, tid = threadIdx.x;
  int32_t* ur = a.uniq_row[0];
  int32_t* nq = a.nuniq[0];
  int32_t* slot = a.slot[0];
#pragma unroll
  for (int k = 1; k < UX_MAX_JOBS; ++k) {
    if (k == job) {
      ur = a.uniq_row[k];
      nq = a.nuniq[k];
      slot = a.slot[k];
    }
  }
  const int row0 = a.row_off[f], rows = a.row_off[f + 1] - row0;
  const int rpp = ux_rows_per_part(rows, a.P);
  const int rlo = p * rpp;
  const int rn = rows - rlo < rpp ? (rows - rlo > 0 ? rows - rlo : 0) : rpp;
  const int W = (rn + 31) >> 5;
  // the previous list's slot entries, this part's share of it
  {
    const int prev = nq[f];
    const int32_t* uf = ur + (size_t)f * a.stride;
    const int j0 = (int)((long long)prev * p / a.P), j1 = (int)((long long)prev * (p + 1) / a.P);
    for (int j = j0 + tid; j < j1; j += T) slot[uf[j]] = -1;
  }
  const int32_t* kb = a.keys + (size_t)job * a.job_stride;
  ux_mark_part<T>(a, kb, f, p, row0, rlo, bm, W);
  uint32_t c = 0;
  for (int w = tid; w < W; w += T) c += (uint32_t)__popc(bm[w]);
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) c += __shfl_xor(c, d);
  if ((tid & 63) == 0) wsum[tid >> 6] = c;
  __syncthreads();
  if (tid == 0) {
    uint32_t t = 0;
    for (int k = 0; k < T / 64; ++k) t += wsum[k];
    a.cnt[(size_t)job * a.F * a.P + f * a.P + p] = (int32_t)t;
  }
}

template <int T>
__global__ __launch_bounds__(T) void uniq_merge_part_k(const UxMerge a) {
  extern __shared__ uint32_t ux_lds[];
  uint32_t* bm = ux_lds;
  uint32_t* pre = ux_lds + a.wmax;
  uint32_t* wsum = pre + a.wmax;
  const int f = blockIdx.x / a.P, p = blockIdx.x - f * a.P, r = blockIdx.y, job = blockIdx.z, tid = threadIdx.x;
  int32_t* ur = a.uniq_row[0];
  int32_t* nq = a.nuniq[0];
  int32_t* slot = a.slot[0];
  int32_t* src = a.src[0];
#pragma unroll
  for (int k = 1; k < UX_MAX_JOBS; ++k) {
    if (k == job) {
      ur = a.uniq_row[k];
      nq = a.nuniq[k];
      slot = a.slot[k];
      src = a.src[k];
    }
  }
  const int row0 = a.row_off[f], rows = a.row_off[f + 1] - row0;
  const int rpp = ux_rows_per_part(rows, a.P);
  const int rlo = p * rpp;
  const int rn = rows - rlo < rpp ? (rows - rlo > 0 ? rows - rlo : 0) : rpp;
  const int W = (rn + 31) >> 5;
  const int32_t* kb = a.keys + (size_t)job * a.job_stride;
  // distinct rows of the parts below (and of the whole field): one load per lane of the first wave's worth, every wave alike
  const int lane = tid & 63, wave = tid >> 6;
  const int32_t* cf = a.cnt + (size_t)job * a.F * a.P + f * a.P;
  const int cv = lane < a.P ? cf[lane] : 0;
  int below = lane < p ? cv : 0, all = cv;
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    below += __shfl_xor(below, d);
    all += __shfl_xor(all, d);
  }
  ux_mark_part<T>(a, kb, f, p, row0, rlo, bm, W);
  const int per = (W + T - 1) / T;
  const int w0 = tid * per, w1 = w0 + per < W ? w0 + per : W;
  uint32_t run = 0;
  for (int w = w0; w < w1; ++w) run += (uint32_t)__popc(bm[w]);
  uint32_t incl = run;
#pragma unroll
  for (int d = 1; d < RSX_WAVE; d <<= 1) {
    const uint32_t o = __shfl_up(incl, d);
    if (lane >= d) incl += o;
  }
  if (lane == RSX_WAVE - 1) wsum[wave] = incl;
  __syncthreads();
  uint32_t base = 0, total = 0;
#pragma unroll
  for (int k = 0; k < T / 64; ++k) {
    const uint32_t sv = wsum[k];
    base += k < wave ? sv : 0u;
    total += sv;
  }
  uint32_t q = base + incl - run;
  for (int w = w0; w < w1; ++w) {
    pre[w] = q;
    q += (uint32_t)__popc(bm[w]);
  }
  __syncthreads();
  int32_t* sr = src + ((size_t)r * a.F + f) * a.stride + below;
  for (uint32_t j = tid; j < total; j += T) sr[j] = -1;
  __syncthreads();
  {
    const int32_t* kr = kb + (size_t)r * a.rank_stride;
    const int32_t* rs = kr + a.F + f * (a.P + 1) + p;
    const int i0 = rs[0], i1 = rs[1];
    const int32_t* xr = kr + a.F + a.F * (a.P + 1) + a.goff[f];
    for (int i = i0 + tid; i < i1; i += T) {
      const uint32_t x = (uint32_t)(xr[i] - row0 - rlo);
      const uint32_t g = pre[x >> 5] + (uint32_t)__popc(bm[x >> 5] & ((1u << (x & 31u)) - 1u));
      sr[g] = i;
    }
  }
  if (r != 0) return;
  int32_t* uf = ur + (size_t)f * a.stride + below;
  for (int w = tid; w < W; w += T) {
    uint32_t bits = bm[w];
    uint32_t g = pre[w];
    while (bits) {
      const int b = __ffs((int)bits) - 1;
      bits &= bits - 1u;
      const int row = row0 + rlo + (w << 5) + b;
      uf[g] = row;
      slot[row] = f * a.stride + below + (int)g;
      ++g;
    }
  }
  if (p == 0 && tid == 0) nq[f] = all;
}

extern "C" int rsx_uniq_merge(const int32_t* keys, long long rank_stride, int job_stride, const rsx_uniq_merge_job* jobs_h,
                              int njobs, const int32_t* goff, const int32_t* row_off, int32_t* part_counts, int parts,
                              int max_rows_per_field, int max_entries, int F, int N, int stride, rsx_stream_t stream) {
  if (!keys || !jobs_h || njobs < 1 || njobs > UX_MAX_JOBS || !goff || !row_off || F <= 0 || F > 64 || N < 1 ||
      N > UX_MAX_RANKS || stride <= 0 || max_rows_per_field <= 0 || rank_stride < (long long)njobs * job_stride || parts < 1 ||
      parts > RSX_UNIQ_MAX_PARTS || (parts > 1 && !part_counts))
    return RSX_EINVAL;
  UxMerge a;
  a.keys = keys; a.rank_stride = rank_stride; a.job_stride = job_stride;
  for (int k = 0; k < UX_MAX_JOBS; ++k) {
    const rsx_uniq_merge_job& j = jobs_h[k < njobs ? k : 0];
    if (!j.uniq_row || !j.nuniq || !j.slot || !j.src) return RSX_EINVAL;
    a.uniq_row[k] = j.uniq_row; a.nuniq[k] = j.nuniq; a.slot[k] = j.slot; a.src[k] = j.src;
  }
  a.goff = goff; a.row_off = row_off; a.cnt = part_counts; a.F = F; a.N = N; a.stride = stride; a.P = parts;
  a.wmax = (ux_rows_per_part(max_rows_per_field, parts) + 31) / 32;
  // few entries per workgroup (small batches): 256 threads; else 1024
  // (1024-thread workgroups are two per CU: a grid of F x parts x N x jobs of them -- 4 992 at dcn.py's 8 x 4 096 -- ran in ten
  // rounds; 256 threads walk up to 16 384 entries of a part just as well)
  // (measured, dcn.py as 8 emulated ranks: 0.2926 ms per step against 0.2995 with 1024 threads from 2 048 entries on)
  const bool big = max_entries / parts > 16384;
  const int T = big ? 1024 : 256;
  const size_t lds = ((size_t)2 * a.wmax + T / 64) * sizeof(uint32_t);
  if (lds > 160 * 1024) return RSX_EUNSUPPORTED;          // a field of more than ~650 000 rows per part: the caller keeps the example exchange
#define UX_BIG_LDS(K)                                                                                                   \
  if (lds > 64 * 1024) {                                                                                                \
    static const hipError_t attr =                                                                                      \
        hipFuncSetAttribute(reinterpret_cast<const void*>(K), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);  \
    if (attr != hipSuccess) return RSX_EUNSUPPORTED;                                                                    \
  }
  if (parts == 1) {
    const dim3 grid((unsigned)F, (unsigned)N, (unsigned)njobs);
    if (big) {
      UX_BIG_LDS(uniq_merge_k<1024>)
      RSX_LAUNCH(uniq_merge_k<1024>, grid, dim3(1024), lds, rsx_s(stream), a);
    } else {
      UX_BIG_LDS(uniq_merge_k<256>)
      RSX_LAUNCH(uniq_merge_k<256>, grid, dim3(256), lds, rsx_s(stream), a);
    }
    RSX_CHECK_LAUNCH();
    return RSX_OK;
  }
  const dim3 gc((unsigned)F, (unsigned)parts, (unsigned)njobs), gp((unsigned)(F * parts), (unsigned)N, (unsigned)njobs);
  if (big) {
    UX_BIG_LDS(uniq_merge_count_k<1024>)
    UX_BIG_LDS(uniq_merge_part_k<1024>)
    RSX_LAUNCH(uniq_merge_count_k<1024>, gc, dim3(1024), lds, rsx_s(stream), a);
    RSX_LAUNCH(uniq_merge_part_k<1024>, gp, dim3(1024), lds, rsx_s(stream), a);
  } else {
    UX_BIG_LDS(uniq_merge_count_k<256>)
    UX_BIG_LDS(uniq_merge_part_k<256>)
    RSX_LAUNCH(uniq_merge_count_k<256>, gc, dim3(256), lds, rsx_s(stream), a);
    RSX_LAUNCH(uniq_merge_part_k<256>, gp, dim3(256), lds, rsx_s(stream), a);
  }
#undef UX_BIG_LDS
  RSX_CHECK_LAUNCH();
  return RSX_OK;
}

// ---------------------------------------------------------------- merged Adam ------------------------------------------
struct UxSrc {
  const float* G;        // rank 0's [capT, D] block inside the gathered buffer; rank r's = G + r * rank_stride
  const float* G2;       // second table set's (nullable)
  const float* gw1;      // [capT] first-order sums (nullable)
  long long rank_stride; // floats between rank blocks (a multiple of 4)
  const int32_t* src;    // [N][F * stride]
  const int32_t* goff;   // [F + 1]
  int N;
};

// Row-owner body: LPR-lane group g of the wave owns global unique row j = wf * GPW + g of field f.
//   round trip 1: the row id + the N positions;   round trip 2: the row's optimizer state + up to N gradient rows (all requested
//   at once, from clamped indices: a load behind a condition is waited for alone), summed in RANK order.
template <int D, bool SECOND>
__device__ __forceinline__ void merged_rows(const UxSrc& ms, const HotAdam& h, const int32_t* __restrict__ uniq_row,
                                            const int32_t* __restrict__ nuniq, const uint64_t w1_mask, const int F,
                                            const int stride, const uint32_t wg, const float b1p, const float b2p) {
  constexpr int LPR = D / 4;
  constexpr int GPW = RSX_WAVE / LPR;
  const int lane = threadIdx.x & 63;
  const int q = lane % LPR, g = lane / LPR;
  // the compact unit list: field f contributes ceil(nuniq[f] / GPW) wave-sized units (one load per lane + a wave scan)
  const int lc = lane < F ? lane : F - 1;
  const int nu_l = nuniq[lc];
  int incl = lane < F ? (nu_l + GPW - 1) / GPW : 0;
#pragma unroll
  for (int d = 1; d < RSX_WAVE; d <<= 1) {
    const int o = __shfl_up(incl, d);
    if (lane >= d) incl += o;
  }
  const int total = __shfl(incl, RSX_WAVE - 1);
  const int nwaves = (int)h.n_own * 4;
  float* __restrict__ T_ = SECOND ? h.tables2 : h.tables;
  float* __restrict__ M_ = SECOND ? h.m_t2 : h.m_t;
  float* __restrict__ V_ = SECOND ? h.v_t2 : h.v_t;
  const float4* __restrict__ G4 = reinterpret_cast<const float4*>(SECOND ? ms.G2 : ms.G);
  const bool hw1 = !SECOND && h.w1 != nullptr;
  const float* __restrict__ w1p = hw1 ? h.w1 : T_;
  const float* __restrict__ mwp = hw1 ? h.m_w : M_;
  const float* __restrict__ vwp = hw1 ? h.v_w : V_;
  const size_t wst = hw1 ? (size_t)h.w1_stride : 0;
  const float* __restrict__ gwp = (hw1 && ms.gw1 != nullptr) ? ms.gw1 : reinterpret_cast<const float*>(G4);
  Hp hp;
  hp.b1 = h.b1; hp.b2 = h.b2; hp.omb1 = 1.0f - h.b1; hp.omb2 = 1.0f - h.b2; hp.eps = h.eps;
  hp.alpha = h.lr * sqrtf(1.0f - b2p) / (1.0f - b1p);
  const size_t rs4 = (size_t)ms.rank_stride / 4;
  for (int unit = (int)((wg * 256u + threadIdx.x) >> 6); unit < total; unit += nwaves) {
    const int f = __popcll(__ballot(incl <= unit));
    const int below = __shfl(incl, f > 0 ? f - 1 : 0);
    const int wf = unit - (f > 0 ? below : 0);
    const int nu = __shfl(nu_l, f);
    const int j = wf * GPW + g;
    const bool own = j < nu;
    const size_t sl = (size_t)f * stride + (own ? j : nu - 1);
    const int row = uniq_row[sl];
    int idx[UX_MAX_RANKS];
#pragma unroll
    for (int r = 0; r < UX_MAX_RANKS; ++r) idx[r] = ms.src[(size_t)(r < ms.N ? r : ms.N - 1) * F * stride + sl];
    const int g0 = ms.goff[f];
    const bool do1 = hw1 && q == 0 && ((w1_mask >> f) & 1ull);
    // the row's state and the ranks' gradient rows: ONE round trip
    const size_t o = (size_t)row * LPR + q;
    float4 var = reinterpret_cast<const float4*>(T_)[o];
    float4 m = reinterpret_cast<const float4*>(M_)[o];
    float4 v = reinterpret_cast<const float4*>(V_)[o];
    const size_t wi = (size_t)row * wst;
    float w = w1p[wi], mw = mwp[wi], vw = vwp[wi];
    float4 gr[UX_MAX_RANKS];
    float g1[UX_MAX_RANKS];
#pragma unroll
    for (int r = 0; r < UX_MAX_RANKS; ++r) {
      const size_t rr = (size_t)(r < ms.N ? r : ms.N - 1);
      const size_t e = (size_t)(g0 + (idx[r] > 0 ? idx[r] : 0));
      gr[r] = G4[rr * rs4 + e * LPR + q];
      g1[r] = gwp[rr * (size_t)ms.rank_stride + (ms.gw1 != nullptr && hw1 ? e : 0)];
    }
    float4 acc = F4Z;
    float a1 = 0.f;
#pragma unroll
    for (int r = 0; r < UX_MAX_RANKS; ++r) {
      if (r < ms.N && idx[r] >= 0) {           // rank order; a rank that did not touch the row adds nothing (not even a zero)
        acc = f4_add(acc, gr[r]);
        a1 += g1[r];
      }
    }
    if (own) {
      F4_APPLY(adam_sparse1, var, m, v, acc, true, hp);
      reinterpret_cast<float4*>(T_)[o] = var;
      reinterpret_cast<float4*>(M_)[o] = m;
      reinterpret_cast<float4*>(V_)[o] = v;
      if (hw1 && q == 0) {
        bool st = true;
        if (h.w1_sparse) {
          st = do1;
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
    }
  }
}

template <int D, int NR>
__global__ __launch_bounds__(256) void merged_adam_k(const UxSrc ms, const int32_t* __restrict__ uniq_row,
                                                     const int32_t* __restrict__ nuniq, const uint64_t w1_mask, const int F,
                                                     const int stride, const HotAdam h) {
  const float b1p = h.state[0], b2p = h.state[1];
  const uint32_t n_rows = h.tables2 != nullptr ? 2u * h.n_own : h.n_own;
  if (blockIdx.x >= n_rows + h.win_blk + h.extra.n_blk) {
    adam_block(h.cold.args, h.cold.blk_lo + (blockIdx.x - n_rows - h.win_blk - h.extra.n_blk));
  } else if (blockIdx.x >= n_rows + h.win_blk) {
    adam_block(h.extra.args, h.extra.blk_lo + (blockIdx.x - n_rows - h.win_blk));
  } else if (blockIdx.x >= n_rows) {
    window_pass_compact<D, NR>(h, blockIdx.x - n_rows, b1p, b2p, F, stride);
  } else if (blockIdx.x >= h.n_own) {
    merged_rows<D, true>(ms, h, uniq_row, nuniq, 0ull, F, stride, blockIdx.x - h.n_own, b1p, b2p);
  } else {
    merged_rows<D, false>(ms, h, uniq_row, nuniq, w1_mask, F, stride, blockIdx.x, b1p, b2p);
  }
  hot_adam_finish(h, b1p, b2p);
}

template <int D>
static void launch_merged_adam(dim3 grid, hipStream_t st, const UxSrc& ms, const int32_t* uniq_row, const int32_t* nuniq,
                               uint64_t mask, int F, int stride, const HotAdam& h) {
  RSX_COUNT_LAUNCH();
  if (h.win_nr == 4) merged_adam_k<D, 4><<<grid, dim3(256), 0, st>>>(ms, uniq_row, nuniq, mask, F, stride, h);
  else merged_adam_k<D, 1><<<grid, dim3(256), 0, st>>>(ms, uniq_row, nuniq, mask, F, stride, h);
}

extern "C" int rsx_merged_adam_rows(float* tables, float* m_t, float* v_t, float* w1, float* m_w, float* v_w,
                                    const float* G, const float* gw1, long long rank_stride, int N, const int32_t* src,
                                    const int32_t* goff, const int32_t* uniq_row, const int32_t* nuniq,
                                    uint64_t w1_field_mask, int max_units, int F, int D, int stride,
                                    const rsx_adam_seg* extra_segs_h, int n_extra, const rsx_adam_slice* sweep_h,
                                    const rsx_table_set* second_h, const rsx_adam_window* win_h, float* state,
                                    int advance_step, float lr, float beta1, float beta2, float eps, int w1_stride,
                                    int w1_sparse_formula, rsx_stream_t stream) {
  if (w1_stride < 1) return RSX_EINVAL;
  if (!tables || !m_t || !v_t || !G || !src || !goff || !uniq_row || !nuniq || !state || N < 1 || N > UX_MAX_RANKS || F <= 0 ||
      F > 64 || stride <= 0 || max_units <= 0 || (rank_stride & 3) || (N > 1 && rank_stride <= 0))
    return RSX_EINVAL;
  if (!(D == 4 || D == 8 || D == 16 || D == 32 || D == 64)) return RSX_EINVAL;
  if ((gw1 != nullptr) != (w1 != nullptr)) return RSX_EINVAL;
  if (w1 != nullptr && (!m_w || !v_w)) return RSX_EINVAL;
  if (((uintptr_t)G & 15) != 0) return RSX_EINVAL;
  HotAdam h;
  const int rch = hot_adam_init(h, tables, m_t, v_t, w1, m_w, v_w, w1_stride, w1_sparse_formula, extra_segs_h, n_extra, sweep_h,
                                win_h, uniq_row, state, advance_step, lr, beta1, beta2, eps, F, D);
  if (rch != RSX_OK) return rch;
  UxSrc ms;
  ms.G = G; ms.G2 = nullptr; ms.gw1 = gw1; ms.rank_stride = rank_stride; ms.src = src; ms.goff = goff; ms.N = N;
  if (second_h != nullptr) {
    // (rsx_table_set.dX: rank 0's [capT, D] block of the second set's per-rank sums inside the same gathered buffer)
    if (!second_h->tables || !second_h->m || !second_h->v || !second_h->dX || ((uintptr_t)second_h->dX & 15)) return RSX_EINVAL;
    h.tables2 = second_h->tables; h.m_t2 = second_h->m; h.v_t2 = second_h->v; h.dX2 = second_h->dX;
    ms.G2 = second_h->dX;
  }
  // the window pass walks the COMPACT unit list of the lists it visits with a grid stride (window_pass_compact): at most as
  // many workgroups as the dense grid would have, and never more than a launch-full
  if (h.win_blk > 768u) h.win_blk = 768u;
  h.win_compact = 1;
  const long long wgs = ((long long)max_units + 3) / 4;
  h.n_own = (uint32_t)(wgs < 1024 ? wgs : 1024);                 // (grid stride over the compact unit list; 2 048 / 4 096 measured no better)
  h.total_blocks = (second_h != nullptr ? 2u : 1u) * h.n_own + h.win_blk + h.extra.n_blk + h.cold.n_blk;
  const dim3 grid(h.total_blocks);
  switch (D) {
    case 4: launch_merged_adam<4>(grid, rsx_s(stream), ms, uniq_row, nuniq, w1_field_mask, F, stride, h); break;
    case 8: launch_merged_adam<8>(grid, rsx_s(stream), ms, uniq_row, nuniq, w1_field_mask, F, stride, h); break;
    case 16: launch_merged_adam<16>(grid, rsx_s(stream), ms, uniq_row, nuniq, w1_field_mask, F, stride, h); break;
    case 32: launch_merged_adam<32>(grid, rsx_s(stream), ms, uniq_row, nuniq, w1_field_mask, F, stride, h); break;
    default: launch_merged_adam<64>(grid, rsx_s(stream), ms, uniq_row, nuniq, w1_field_mask, F, stride, h); break;
  }
  RSX_CHECK_LAUNCH();
  return RSX_OK;
}
