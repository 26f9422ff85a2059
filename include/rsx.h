/* rsx.h -- C ABI of librsx.so: the MI355X (gfx950) CTR training hot path.
 *
 * The reference (wangruichens/recsys) has NO native/FFI/plugin interface: every op below is a
 * TensorFlow-1.x graph op invoked from the model_fn bodies.  Each entry point cites the reference
 * call site (file:line under /root/reference) whose arithmetic it replaces; SURVEY.md section 8b is
 * the contract.  INTEGRATION.md shows the ctypes binding a maintainer adds on the Python side.
 *
 * Conventions
 *   - extern "C", plain pointers + sizes, no torch / C++ types.
 *   - every pointer is a DEVICE pointer unless the name ends in _h (host).
 *   - the caller owns all buffers incl. workspaces; the library never allocates or frees device
 *     memory and never synchronises the stream (safe under hipGraph stream capture).
 *   - return 0 (RSX_OK) or a negative rsx_status; no exceptions cross the ABI.
 *   - re-entrant: no global mutable state; one stream per call (rsx_stream_t == hipStream_t).
 *   - fp32 everywhere; compiled with -ffp-contract=off and correctly rounded div/sqrt so the
 *     element-wise math is reproducible against an IEEE fp32 restatement.
 */
#ifndef RSX_H_
#define RSX_H_
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* rsx_stream_t; /* hipStream_t */

typedef enum {
  RSX_OK = 0,
  RSX_EINVAL = -1,       /* bad argument (null pointer, unsupported D, ...) */
  RSX_ELAUNCH = -2,      /* hipGetLastError() != hipSuccess after a launch */
  RSX_EUNSUPPORTED = -3, /* valid request outside the implemented envelope */
  RSX_EDATA = -4,        /* corrupt input data (TFRecord crc, malformed Example) */
  RSX_ECOMM = -5         /* a collective library call failed (rsx_comm_last_error_h() has RCCL's message) */
} rsx_status;

int rsx_version(void);
/* Kernel launches issued by this library in this process so far (a relaxed counter; bench.py derives launches per step). */
unsigned long long rsx_dbg_launch_count(void);
const char* rsx_strerror(int status);

/* ---------------------------------------------------------------------------------------------
 * Embedding path
 * Layout: all F per-field tables concatenated row-wise into tables[R, D] in TF's name-sorted slot
 * order; row_off[F+1] (int32, device) are the slot offsets; ids[B, F] int32 are table-local ids.
 * D in {4, 8, 16, 32, 64}.
 * ------------------------------------------------------------------------------------------- */

/* Replaces tf.feature_column.input_layer(embedding cols) fm/fm.py:118 (deepfm/deepfm.py:85,
 * xdeepfm/xdeepfm.py:128,185, dcn/dcn.py:123), the first-order one-hot matmul fm/fm.py:117,121
 * (without bias/relu) and the FM second-order term fm/fm.py:124-129.
 *   E[B, F*D]   gathered rows (always written)
 *   S[B, D]     sum over fields (nullable; required when y2 != NULL -- saved for backward)
 *   y1[B]       sum_f w1[row] over fields whose bit is set in w1_field_mask (nullable with w1)
 *   y2[B]       0.5 * sum_d((sum_f E)^2 - sum_f E^2)   (nullable)                               */
int rsx_gather_fm_fwd(const float* tables, const float* w1, const int32_t* row_off, const int32_t* ids,
                      float* E, float* S, float* y1, float* y2, uint64_t w1_field_mask,
                      int B, int F, int D, rsx_stream_t stream);

/* Device workspace written by rsx_field_sort, all caller-owned.  `stride` (>= B) is the per-field
 * capacity the buffers were allocated with, so partial batches reuse the same workspace:
 *   perm     int32 [F, stride]    example index b of the i-th entry of field f sorted by (id, b)
 *   seg_off  int32 [F, stride+1]  start position of the j-th unique id of field f; seg_off[f][nuniq]=B
 *   uniq_row int32 [F, stride]    global row (row_off[f]+id) of the j-th unique id
 *   nuniq    int32 [F]            number of unique ids of field f (zero-initialise once)
 *   slot     int32 [R]            row -> f*stride+j of this step, -1 elsewhere (initialise to -1 once;
 *                                 the kernel clears the previous step's entries itself)
 *   segid    int32 [F*stride + 2F + F*ceil(stride/16)]  (optional, B > 512 only) two-stage segment-sum workspace:
 *                                 unique index j of the segment holding sorted position i, then the per-field counts
 *                                 and lists of the long (> 16 entries) and huge (> 256) segments
 * The sparse gradient lives at the same slot index: G[F*stride, D], gw1[F*stride].                 */
/* Dedup stage of the sparse gradient (TF: unique() inside safe_embedding_lookup_sparse and
 * _apply_sparse_duplicate_indices): one workgroup per field sorts the
 * composite key (id << log2B | b) in LDS -- a barrier-free rank sort for B <= 512, a stable LSD radix
 * sort over the id bits above -- then flags segment heads and scans them.
 * Depends on ids only, so it may ride in another launch (rsx_sort_job).  segid is nullable.
 * Envelope: B <= 16384 and (max rows per field) << ceil(log2 B) < 2^32.                         */
int rsx_field_sort(const int32_t* ids, const int32_t* row_off, int32_t* perm, int32_t* seg_off,
                   int32_t* uniq_row, int32_t* nuniq, int32_t* slot, int32_t* segid,
                   int max_rows_per_field, int B, int F, int stride, rsx_stream_t stream);
/* The same contract for batches beyond one workgroup's LDS (B > 16384: data-parallel steps sort N*b examples): a stable
 * LSD radix sort over the id bits through global memory (transpose, 2 x {histogram, scan, stable scatter}, multi-workgroup
 * segment detection), all fields per launch.  workspace: rsx_field_sort_large_workspace_ints(B, F, stride) int32.
 * Envelope: max rows per field <= 2^18, B <= 2^24.  Works for any B >= 1; rsx_field_sort is faster below 16384.        */
/* The workspace must be ZERO before its first use with a given (F, stride); every call leaves it reusable (the digit
 * totals it accumulates with integer atomics are cleared again by the call itself).                                  */
size_t rsx_field_sort_large_workspace_ints(int B, int F, int stride);
int rsx_field_sort_large(const int32_t* ids, const int32_t* row_off, int32_t* perm, int32_t* seg_off,
                         int32_t* uniq_row, int32_t* nuniq, int32_t* slot, int32_t* segid, int32_t* workspace,
                         int max_rows_per_field, int B, int F, int stride, rsx_stream_t stream);
/* The same for keys that are already field-major: ids_t [F, stride] int32 (field f's B keys at ids_t + f * stride).  No
 * transpose launch; ids_t is CLOBBERED (it ends up holding the sorted keys).                                            */
int rsx_field_sort_large_t(int32_t* ids_t, const int32_t* row_off, int32_t* perm, int32_t* seg_off, int32_t* uniq_row,
                           int32_t* nuniq, int32_t* slot, int32_t* segid, int32_t* workspace, int max_rows_per_field, int B,
                           int F, int stride, rsx_stream_t stream);
/* fm.py's forward AND head in one launch (round 4; fm/fm.py:117-133,146-149): rsx_gather_fm_fwd (without E: fm.py's backward
 * recomputes the FM term from the table row) followed, in the same wave, by rsx_fm_head's arithmetic for that example --
 * prob, gy1 = d loss / d y1, gy2 = d loss / d y2 -- and the head's dense gradients as rows of `terms` [ceil(B/16), term_stride]
 * laid out like the caller's dense gradient arena: row g = the sum, in example order, over examples 16 g .. 16 g + 15 of the
 * example's d/d c0 at [off_c0], d/d wo at [off_wo], [off_wo + 1], d/d bo at [off_bo], zeros elsewhere below n_dense, and of
 * its cross-entropy term at [n_dense].  The optimizer sums the rows in order (an RSX_ADAM_DENSE segment with g = terms,
 * B = ceil(batch/16), stride = term_stride); the mean loss is sum(terms[:, n_dense]) / batch.  term_stride: a multiple of 4,
 * n_dense < term_stride <= 64.                                                                                           */
int rsx_gather_fm_head(const float* tables, const float* w1, const int32_t* row_off, const int32_t* ids, float* S,
                       uint64_t w1_field_mask, const float* c0, const float* wo, const float* bo, const float* labels,
                       float* prob, float* gy1, float* gy2, float* terms, int term_stride, int n_dense, int off_c0,
                       int off_wo, int off_bo, float loss_scale, int B, int F, int D, rsx_stream_t stream);
/* A per-field dedup sort job (the arguments of rsx_field_sort) that may ride along in another launch. */
typedef struct {
  const int32_t* ids;
  const int32_t* row_off;
  int32_t* perm;
  int32_t* seg_off;
  int32_t* uniq_row;
  int32_t* nuniq;
  int32_t* slot;
  int32_t* segid;             /* nullable */
  int32_t max_rows_per_field, B, F, stride;
  uint64_t skip_mask;         /* bit f set: field f is NOT sorted -- its workspace rows must be zero-initialised and then keep
                                 nuniq[f] = 0, so the scatter / optimizer entry points find nothing to do for it (a
                                 caller that handles some fields' gradients elsewhere).  Skip a field always or never; 0 =
                                 sort every field. */
} rsx_sort_job;
/* njobs (<= 8) independent sorts of the same shape (B, F, stride) in ONE launch, F workgroups each: the k batches of an
 * optimizer window (rsx_adam_window), each into its own workspace.  B <= 16384 (the rsx_field_sort range).              */
int rsx_field_sort_multi(const rsx_sort_job* jobs_h, int njobs, rsx_stream_t stream);
/* rsx_gather_fm_fwd with the step's dedup sort (`sort_h`, see rsx_sort_job above) riding along as F extra workgroups of
 * the same launch: the sort only needs the ids, and once it sits in the step's first launch every later launch may carry
 * a slice of the untouched-row optimizer sweep (which reads the sort's slot map).  sort_h == NULL: plain gather.       */
int rsx_gather_fm_fwd_sort(const float* tables, const float* w1, const int32_t* row_off, const int32_t* ids,
                           float* E, float* S, float* y1, float* y2, uint64_t w1_field_mask, int B, int F, int D,
                           const rsx_sort_job* sort_h, rsx_stream_t stream);

/* Workspace of the two-stage segment-sum (large / skewed batches), caller-owned, host struct; nch = ceil(stride/16):
 *   segid  written by rsx_field_sort (see above);  P float [F, nch, 2, D];  P1 float [F, nch, 2].
 * Stage A, rsx_segsum_partials: every 16-position chunk of the sorted order sums its overlap with the LONG segments
 * (> 16 entries; at most two per chunk) -- uniform work however skewed the ids are (Zipf heads, 3-row vocabularies).
 * Stage B, rsx_segsum_bwd / _rows / _adam_rows with partials_h != NULL: every long segment gets a helper group (a
 * whole wave above 256 entries) that adds its chunk partials in ascending chunk order; segments of <= 16 entries are
 * summed entry by entry exactly as in the single-stage form.  Both forms are deterministic; they differ in the
 * rounding of long segments only.  B passed to stage A and stage B must be the B of the sort.                   */
typedef struct {
  const int32_t* segid;
  float* P;
  float* P1;                  /* nullable when gy1 is NULL */
  float* G;                   /* nullable [F*stride, D] (+ gw1 [F*stride], nullable): rsx_segsum_partials then also FINISHES
                                 every segment that lies inside one 16-position chunk of the sorted order and writes its sum
                                 here; the stage-B entry points pick those rows up (rsx_segsum_bwd with the same G: nothing
                                 left to do for them).  NULL: stage A only produces the long segments' chunk partials.   */
  float* gw1;
  /* Padding rows (DIN's history padding, din/din.py:56-57,107: their entries carry exactly-zero gradients by construction, so
   * their -- huge -- segments need not be walked).  null_row >= 0: that global row; RSX_NULL_LAST_ROW: the LAST row of every
   * field, row_off[f + 1] - 1 (row_off required then: a dummy row appended to each table, to which the padding entries' keys
   * are mapped -- this keeps a real row 0 usable by ordinary lookups); RSX_NULL_NONE: none.  Read by the stage-A / stage-B
   * entry points that take this struct (the explicit null_row arguments of rsx_segsum_partials / _rows still work).    */
  const int32_t* row_off;
  int32_t null_row;
  uint64_t skip_mask;         /* the sort's rsx_sort_job.skip_mask (fields without sort results): stage A passes over them */
} rsx_seg_partials;
#define RSX_NULL_NONE (-1)
#define RSX_NULL_LAST_ROW (-2)
/* Layout of the per-example inputs (dX, S, gy1, gy2) when they are read in place from an all-gathered buffer (data
 * parallel): example e lives in rank block e / examples at local index e % examples; blocks are stride_floats apart
 * (a multiple of 4) and each pointer names its array inside block 0.  NULL = contiguous over the batch.            */
typedef struct {
  int32_t examples;
  int64_t stride_floats;
} rsx_example_blocks;
int rsx_segsum_partials(const float* tables, const float* S, const float* dX, const float* gy1, const float* gy2,
                        const int32_t* perm, const int32_t* seg_off, const int32_t* uniq_row,
                        const rsx_seg_partials* ws_h, uint64_t w1_field_mask, int B, int F, int D, int stride,
                        int null_row, const rsx_example_blocks* blocks_h, rsx_stream_t stream);


/* Row-wise gradient "scatter" as a sorted segment-sum (replaces the IndexedSlices gradient of the
 * gather + tf.unsorted_segment_sum, Appendix A-4): for unique row (f, j)
 *   G[f*B+j, :]  = sum over its examples b, ascending:  (gy2[b]*S[b,:] - gy2[b]*T[row,:]) + dX[b, f*D:(f+1)*D]
 *   gw1[f*B+j]   = sum over its examples b, ascending:  gy1[b]       (fields in w1_field_mask)
 * dX (grad of the flattened embedding from the DNN/CIN/cross consumers), gy2/S (FM term) and
 * gy1/gw1 (first-order term) are each nullable.  Sums run in ascending b like TF's CPU kernels.   */
int rsx_segsum_bwd(const float* tables, const float* S, const float* dX, const float* gy1, const float* gy2,
                   const int32_t* perm, const int32_t* seg_off, const int32_t* uniq_row, const int32_t* nuniq,
                   float* G, float* gw1, uint64_t w1_field_mask, int B, int F, int D, int stride,
                   const rsx_seg_partials* partials_h, const rsx_example_blocks* blocks_h, rsx_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Optimizer: tf.train.AdamOptimizer(lr).minimize(...) fm/fm.py:162-163.
 * One launch sweeps any number of variable segments.  state (device, RSX_ADAM_STATE_WORDS 32-bit words) =
 * {beta1^t, beta2^t, <uint32 blocks-done ticket>, <uint32 step t>, then 32 arrival counters one 128-byte line apart;
 *  words 8 .. 15: the step sizes of the current optimizer window's steps (rsx_adam_window)};
 * initialise words 0..3 with rsx_adam_state_init_h and the rest with zeros.  The last workgroup to finish advances the
 * powers, so the sweep is replayable from a hipGraph with no per-step host arguments.  (Every workgroup announces its
 * arrival on the counter of its index modulo 32 and only the last of each residue touches the shared ticket: one
 * counter for all of them serialised ~45 ns per workgroup -- 30 us of a 4096-example scatter launch.)
 * ------------------------------------------------------------------------------------------- */
typedef enum {
  RSX_ADAM_DENSE = 0,      /* ApplyAdam: m += (g-m)(1-b1); v += (g*g-v)(1-b2); var -= (m*a)/(sqrt(v)+eps).
                              g dense [n]; if zero_grad != 0 g is zeroed after use.                 */
  RSX_ADAM_TABLE_TF1 = 1,  /* non-lazy sparse (_apply_sparse_shared): whole table decays and moves;
                              rows with slot[row]>=0 add G[slot[row], :] first.  n = rows, d = D.   */
  RSX_ADAM_VEC_SLOT = 2,   /* dense formula on a vector whose dense gradient is g[slot[row]] where
                              slot[row]>=0 and 0 elsewhere (one-hot matmul kernels: w1). n = rows.  */
  RSX_ADAM_TABLE_ROWS = 3, /* only the rows listed in uniq_row/nuniq, sparse formula with their gradient.  Alone it is the
                              lazy_rows mode (NOT TF semantics); together with TABLE_TF1_COLD it is the exact TF-1 update. */
  RSX_ADAM_VEC_ROWS = 4,   /* lazy_rows counterpart for vectors (sparse formula).                                        */
  RSX_ADAM_TABLE_TF1_COLD = 5, /* the untouched part of TABLE_TF1: rows with slot[row] < 0 decay and move, touched rows
                              are skipped (they are updated by a TABLE_ROWS segment once their gradient exists).  g unused. */
  RSX_ADAM_VEC_COLD = 6,   /* the untouched part of VEC_SLOT (dense formula, zero gradient); touched elements skipped.   */
  RSX_ADAM_VEC_ROWS_DENSE = 7 /* touched elements of a VEC_SLOT variable, ApplyAdam formula.                            */
} rsx_adam_kind;

typedef struct {
  int32_t kind;            /* rsx_adam_kind */
  int32_t d;               /* row width for table kinds */
  int64_t n;               /* elements (DENSE) or rows (others); for *_ROWS kinds: F*B slots        */
  float* var;
  float* m;
  float* v;
  float* g;                /* DENSE: grad [n]; TABLE_*: G [F*B, d]; VEC_*: gw1 [F*B]                */
  const int32_t* slot;     /* TABLE_TF1 / VEC_SLOT: row -> slot map                                  */
  const int32_t* uniq_row; /* *_ROWS */
  const int32_t* nuniq;    /* *_ROWS */
  int32_t B;               /* *_ROWS: examples per field this step (n = F*B).  DENSE: replicas whose arenas are summed */
  int32_t stride;          /* *_ROWS: per-field capacity of the rsx_field_sort workspace.  DENSE with B > 1: floats
                              between the replicas' arenas inside g (an all-gathered buffer), summed in order r = 0.. */
  int32_t zero_grad;       /* DENSE */
  /* *_COLD kinds, optimizer WINDOW of 1 + w steps (w = leading non-NULL entries, 0..7): the slot maps of the window's LATER
   * steps (their batches' ids are known ahead: rsx_field_sort_multi).  A row / element that NO step of the window touches
   * (slot and every slot_w < 0) receives the window's 1 + w untouched-row updates back to back in registers -- the same fp32
   * operations, in the same order, as 1 + w single-step sweeps, for 1/(1 + w) of their HBM traffic; every other row is
   * skipped and is brought up to date step by step by rsx_segsum_adam_rows (`win_h`).  All NULL: a one-step sweep.  Every
   * COLD segment of one call / slice carries the same maps (tables that share a dedup sort), and the maps are equally spaced
   * in memory (slot_w[j] = slot_w[0] + j * stride, 16-byte aligned: slices of one allocation), else RSX_EINVAL.             */
  const int32_t* slot_w[7];
  /* TABLE_ROWS / VEC_ROWS_DENSE: g is the FIRST of g_replicas (> 1) arrays g_replica_stride floats apart (a multiple of 4) --
   * per-replica gradient rows inside an all-gathered buffer -- added in order r = 0.. before the update: every replica applies
   * the same sum.  0 / 1: g alone.                                                        */
  int32_t g_replicas;
  int64_t g_replica_stride;
} rsx_adam_seg;

#define RSX_ADAM_MAX_SEGS 12
#define RSX_ADAM_STATE_WORDS (4 + 32 * 32)
int rsx_adam_state_init_h(float* state_h /* host float[4]: words 0..3 of the state */, float beta1, float beta2);
int rsx_adam_tf1_multi(const rsx_adam_seg* segs_h, int nseg, float* state, float lr, float beta1,
                       float beta2, float eps, rsx_stream_t stream);

/* Split form of the same TF-1 update, for overlapping the HBM-bound sweep with latency-bound kernels:
 * the untouched rows (*_COLD kinds) depend only on the PREVIOUS optimizer state and on which rows this step touches
 * (the slot map of rsx_field_sort), so their sweep may run -- cut into slices of workgroups -- as extra workgroups of
 * the tower launches (rsx_tower_fwd_layer / rsx_tower_head / rsx_tower_bwd_layer `sweep_h`) or stand-alone
 * (rsx_adam_slice_run); the touched rows + dense variables follow in one rsx_adam_tf1_multi launch
 * (TABLE_ROWS / VEC_ROWS_DENSE / DENSE kinds), which also advances the beta powers.  COLD slices never advance them.
 * Element for element the arithmetic is that of TABLE_TF1 / VEC_SLOT.                                               */
typedef struct {
  const rsx_adam_seg* segs;   /* host array, *_COLD kinds                                  */
  int32_t nseg;
  float lr, beta1, beta2, eps;
  float* state;               /* device state (see above), read only                        */
  uint32_t blk_lo, blk_hi;    /* workgroup range [lo, hi) out of rsx_adam_num_blocks(segs)  */
  /* A WINDOW sweep (segments with slot_w) cut into one slice per step of its window, each riding in that step's head launch
   * (rsx_tower_head, round 4): a row that no step of the window touches is read by none of them, so its 1 + w updates may be
   * applied at any time inside the window.
   *   window_block_u      float4 per lane of a TABLE_TF1_COLD block: 0 (= 8, the stand-alone sweep's) or 2 / 4 -- smaller
   *                       blocks spread a slice over the CUs a latency-bound launch leaves idle; blk_lo / blk_hi then count
   *                       blocks of rsx_adam_num_blocks_u(segs, nseg, window_block_u).
   *   alphas_from_state   0: the step sizes come from the beta powers (state words 0, 1) -- the slice of the window's FIRST
   *                       step, whose block 0 also leaves them in state words 8.. --; 1: they are read from there (later steps:
   *                       the powers have been advanced since).                                                              */
  int32_t window_block_u;
  int32_t alphas_from_state;
} rsx_adam_slice;
int64_t rsx_adam_num_blocks(const rsx_adam_seg* segs_h, int nseg);
int64_t rsx_adam_num_blocks_u(const rsx_adam_seg* segs_h, int nseg, int window_block_u);
typedef struct rsx_table_set {
  float* tables; float* m; float* v;        /* [R, D] each */
  const float* dX;                           /* [B, F*D] */
  const rsx_seg_partials* partials;          /* two-stage workspace of this set (P filled by rsx_segsum_partials), or NULL */
} rsx_table_set;
/* rsx_segsum_bwd fused with the touched-row half of the split update: the group that sums the gradient of unique row
 * (f, j) applies the TABLE_ROWS (and VEC_ROWS_DENSE for w1) update to it at once; `extra_segs_h` (e.g. the DENSE segment)
 * ride along as extra workgroups and the last workgroup advances the beta powers.  Replaces
 * rsx_segsum_bwd + rsx_adam_tf1_multi(TABLE_ROWS, VEC_ROWS_DENSE, DENSE) with identical arithmetic.  sweep_h
 * (nullable) may carry TABLE_TF1_COLD segments only: a VEC_COLD slice restores touched elements of its float4s and
 * would race with this launch's own update of them (RSX_EINVAL).                                                */
/* Optimizer window: k <= RSX_ADAM_WINDOW_MAX consecutive steps whose batches' ids are known when the first one starts (the
 * input pipeline runs ahead of the device).  All k dedup sorts run first (rsx_field_sort_multi, one workspace per step);
 * the rows NO step of the window touches then need ONE pass over the optimizer state for the whole window
 * (rsx_adam_seg.slot_w) instead of one per step -- TF-1's non-lazy Adam moves every row every step, and
 * for an untouched row step t+1's update only needs step t's result, so k of them are applied back to back in registers.
 * A row that some step of the window touches receives exactly the updates of k single steps, in order, but its
 * zero-gradient updates are applied LAZILY (round 3): step `cur`'s scatter launch updates the rows it touches with their
 * gradient (as always); for cur < k - 1 it also walks the NEXT step's unique-row list and brings the rows that it does not
 * touch itself up to date -- the updates of the steps (last touch, cur] back to back in registers -- so the next step's
 * gather and scatter see current rows; the window's last step walks every earlier list and finishes the rows whose last touch
 * was that list's step.  (Round 2 walked all other lists in every step: k (k - 1) walks per window instead of 2 (k - 1).)
 * The step sizes of the window's steps are kept in state words 8 .. 8 + k - 1 (written by the window's sweep, rsx_adam_slice_run
 * over COLD segments with slot_w, which therefore has to run before the window's first step; each step's launch also
 * leaves its own there).  Between two windows the state equals
 * that of k single steps, bit for bit; INSIDE a window neither the untouched rows (already k steps ahead) nor the touched
 * ones (possibly behind) are a state a step-by-step run passes through: the caller must not read, evaluate or checkpoint
 * the variables there, and must run the window's steps 0 .. k-1 in order. */
#define RSX_ADAM_WINDOW_MAX 8
typedef struct {
  int32_t k;                 /* steps of the window (1: no window) */
  int32_t cur;               /* this step's position, 0 .. k-1 */
  int32_t max_unique;        /* upper bound of nuniq[i][f] (the batch size) */
  const int32_t* uniq_row[RSX_ADAM_WINDOW_MAX];   /* the k sort workspaces (rsx_field_sort outputs), entry `cur` = this step's */
  const int32_t* nuniq[RSX_ADAM_WINDOW_MAX];
  const int32_t* slot[RSX_ADAM_WINDOW_MAX];
} rsx_adam_window;
int rsx_segsum_adam_rows(float* tables, float* m_t, float* v_t, float* w1, float* m_w, float* v_w, const float* S,
                         const float* dX, const float* gy1, const float* gy2, const int32_t* perm, const int32_t* seg_off,
                         const int32_t* uniq_row, const int32_t* nuniq, uint64_t w1_field_mask, int B, int F, int D,
                         int stride, const rsx_adam_seg* extra_segs_h, int n_extra, const rsx_adam_slice* sweep_h,
                         const rsx_seg_partials* partials_h, const rsx_example_blocks* blocks_h,
                         const rsx_table_set* second_h, const rsx_adam_window* win_h, float* state, int advance_step,
                         float lr, float beta1, float beta2, float eps, rsx_stream_t stream);
/* The same with two options for the first-order vector: its elements w1_stride floats apart (4: column 0 of a 4-wide table,
 * as DIN's item bias is stored), and w1_sparse_formula != 0: the sparse (IndexedSlices) Adam formula -- the vector is a 1-D
 * variable read through tf.gather (din/din.py:96,139), not the kernel of a dense layer on one-hot input (fm/fm.py:121) --
 * applied to the rows of the fields in w1_field_mask only.                                                            */
int rsx_segsum_adam_rows2(float* tables, float* m_t, float* v_t, float* w1, float* m_w, float* v_w, const float* S,
                          const float* dX, const float* gy1, const float* gy2, const int32_t* perm, const int32_t* seg_off,
                          const int32_t* uniq_row, const int32_t* nuniq, uint64_t w1_field_mask, int B, int F, int D,
                          int stride, const rsx_adam_seg* extra_segs_h, int n_extra, const rsx_adam_slice* sweep_h,
                          const rsx_seg_partials* partials_h, const rsx_example_blocks* blocks_h,
                          const rsx_table_set* second_h, const rsx_adam_window* win_h, float* state, int advance_step,
                          float lr, float beta1, float beta2, float eps, int w1_stride, int w1_sparse_formula,
                          rsx_stream_t stream);

/* second_h (nullable): a second table set looked up with the SAME ids (one sort serves both: xDeepFM's two input_layer
 * calls, xdeepfm/xdeepfm.py:125,185); its row-owner workgroups run in the same launch.  It has no first-order vector and
 * no FM term; its gradient rows dX use the same example blocks.                                                      */
/* advance_step = 0: this launch leaves the beta powers / step counter alone because a later launch of the SAME step advances
 * them (models with two table sets: xdeepfm.py); 1 otherwise.                                                      */
/* ---- data parallel: the exchange of per-rank UNIQUE-ROW lists (round 5) -------------------------------------------------
 * tf.distribute.MirroredStrategy (fm/fm.py:184-194, deepfm/deepfm.py:202-204, xdeepfm/xdeepfm.py:249-251, dcn/dcn.py:235-237,
 * din/din.py:204-206; deepfm/readme.md:24 "each step runs two batches") hands AdamOptimizer the replicas' IndexedSlices of
 * every embedding variable, one (row, gradient) pair per looked-up id, concatenated in replica order; _apply_sparse
 * de-duplicates them.  Each rank here reduces ITS slices first (its own dedup sort + sorted
 * segment-sum: what a single replica runs) and the ranks exchange unique (row, sum) lists in two fixed-size collectives:
 *   ids phase (once per optimizer window):  keys = [nuniq[F] | unique rows, field f's at goff[f] .. goff[f] + nuniq[f])
 *   after backward:                         G[capT, D] (+ a second table set's, + gw1[capT]) laid out the same way
 * goff [F + 1] (device): cap_f = goff[f+1] - goff[f] >= the unique rows ONE rank can have in field f -- min(batch, rows of the
 * field) -- and capT = goff[F].                                                                                          */
#define RSX_UNIQ_MAX_RANKS 8
#define RSX_UNIQ_MAX_PARTS 64     /* row-range parts per field of the merge (one bitmap word range each; one lane each) */
/* The same sorted segment-sum as rsx_segsum_bwd, unique row j of field f written at row goff[f] + j of G / gw1 (the rank's
 * block of the exchange) instead of f * stride + j.  Two-stage (partials_h): stage A must have been given its own full-stride
 * G / gw1 scratch (RSX_EINVAL when partials_h->G == G).  null_row as in rsx_segsum_partials.                               */
int rsx_segsum_bwd_packed(const float* tables, const float* S, const float* dX, const float* gy1, const float* gy2,
                          const int32_t* perm, const int32_t* seg_off, const int32_t* uniq_row, const int32_t* nuniq,
                          float* G, float* gw1, uint64_t w1_field_mask, int B, int F, int D, int stride, int null_row,
                          const rsx_seg_partials* partials_h, const int32_t* goff, rsx_stream_t stream);
/* The key block of a rank's batch, [F + F (parts + 1) + capT] int32:  keys[0 .. F) = nuniq;  keys[F + f (parts + 1) + p] = the
 * first position of field f's (ascending) list whose row lies in row-range part p or above, p = 0 .. parts (part p = rows
 * [p rpp, (p + 1) rpp) of the field, rpp = ceil(rows_f / parts) rounded up to 32; entry `parts` = nuniq[f]) -- the merge's part
 * workgroups go straight to their sub-range of every rank's list;  then keys[F + F (parts + 1) + goff[f] + j] = uniq_row[f, j]
 * (j < nuniq[f]; -1 up to cap_f).  For the LOCAL sort outputs of up to RSX_ADAM_WINDOW_MAX batches (an optimizer window) in one
 * launch.  parts: 1 .. RSX_UNIQ_MAX_PARTS, the same value rsx_uniq_merge is given.                                        */
typedef struct {
  const int32_t* uniq_row;    /* [F, stride]: rsx_field_sort's output for the rank's own batch */
  const int32_t* nuniq;       /* [F] */
  int32_t* keys;              /* out: [F + capT] */
} rsx_uniq_pack_job;
int rsx_uniq_pack(const rsx_uniq_pack_job* jobs_h, int njobs, const int32_t* goff, const int32_t* row_off, int F, int stride,
                  int parts, int max_cap /* the largest cap_f: launch shape */, rsx_stream_t stream);
/* The N ranks' key blocks -> what rsx_field_sort leaves for the GLOBAL batch, without a global sort: per job (a batch of
 * the window) the global unique rows of every field in ascending order (uniq_row [F, stride], nuniq [F]), the slot map (row
 * -> f * stride + j for touched rows; the entries of the workspace's PREVIOUS contents are reset to -1 first, so uniq_row /
 * nuniq must hold what the last call -- or zero-initialisation -- left), and src [N][F, stride]: src[r][f, j] = position of
 * global unique row j in rank r's list, or -1.  keys: rank r's block of job k at keys + r * rank_stride + k * job_stride
 * (int32 units), as an all-gather of the ranks' packed blocks leaves them.  stride >= the global unique rows of any field
 * (min(N * batch, rows of the field)).  max_rows_per_field: the largest field's row count (a presence bitmap of it lives
 * in LDS: RSX_EUNSUPPORTED beyond ~650 000 rows per part); max_entries: N * the largest cap_f (launch shape only).
 * parts > 1 (long lists: dcn.py at 8 x 4 096, din.py's item table): every field's row range is cut into `parts` ranges merged by
 * their own workgroups -- a count pass (distinct rows per part -> part_counts [njobs][F * parts] int32, caller-owned; it also
 * resets the previous slot entries) and an emit pass, two launches; parts == 1: one launch, part_counts unused.
 * N <= RSX_UNIQ_MAX_RANKS.  Deterministic.                                                                              */
typedef struct {
  int32_t* uniq_row;          /* in/out [F, stride] */
  int32_t* nuniq;             /* in/out [F] */
  int32_t* slot;              /* in/out [R + 4] */
  int32_t* src;               /* out [N][F, stride] */
} rsx_uniq_merge_job;
int rsx_uniq_merge(const int32_t* keys, long long rank_stride, int job_stride, const rsx_uniq_merge_job* jobs_h, int njobs,
                   const int32_t* goff, const int32_t* row_off, int32_t* part_counts, int parts, int max_rows_per_field,
                   int max_entries, int F, int N, int stride, rsx_stream_t stream);
/* The optimizer launch of the exchange: rsx_segsum_adam_rows2 with the gradient of global unique row (f, j) taken from the
 * ranks' lists -- sum over r = 0 .. N-1, IN RANK ORDER (every replica adds the same numbers in the same order: replicas stay
 * bit-identical), of G_r[goff[f] + src[r][f, j]] over the ranks with src >= 0; G_r = G + r * rank_stride floats (rank 0's
 * block inside the gathered buffer; rank_stride % 4 == 0, G 16-byte aligned), gw1 likewise (nullable with w1).  The per-rank
 * sums already carry the FM term (rsx_segsum_bwd_packed).  uniq_row / nuniq / src: rsx_uniq_merge's outputs for this step;
 * max_units: sum over fields of ceil(min(N * batch, rows_f) / (256 / D)) (the launch's row-owner grid, capped inside).
 * second_h: rsx_table_set with dX = rank 0's [capT, D] block of the second table set's sums in the same buffer (partials
 * ignored).  extra segments, sweep slice, window, state, advance_step, w1_stride / w1_sparse_formula: as rsx_segsum_adam_rows2. */
int rsx_merged_adam_rows(float* tables, float* m_t, float* v_t, float* w1, float* m_w, float* v_w, const float* G,
                         const float* gw1, long long rank_stride, int N, const int32_t* src, const int32_t* goff,
                         const int32_t* uniq_row, const int32_t* nuniq, uint64_t w1_field_mask, int max_units, int F, int D,
                         int stride, const rsx_adam_seg* extra_segs_h, int n_extra, const rsx_adam_slice* sweep_h,
                         const rsx_table_set* second_h, const rsx_adam_window* win_h, float* state, int advance_step,
                         float lr, float beta1, float beta2, float eps, int w1_stride, int w1_sparse_formula,
                         rsx_stream_t stream);
/* ---------------------------------------------------------------------------------------------
 * Collectives of the data-parallel step (csrc/comm.cpp), on the step's own stream.
 * Replace what tf.distribute.MirroredStrategy() puts INTO the training graph (fm/fm.py:184-194, twins deepfm/deepfm.py:
 * 177-187, xdeepfm/xdeepfm.py:258-268, dcn/dcn.py:215-225, din/din.py:187-190): the cross-replica sum of the dense gradients
 * (all-reduce) and the aggregation of the replicas' IndexedSlices (concatenation = all-gather).
 * One process per GPU; the communicator is RCCL's (xGMI inside a node), bound at run time (no link-time dependency: on a
 * host without RCCL every entry returns RSX_EUNSUPPORTED and the rest of the library works).
 *   - a communicator is a caller-owned opaque handle; rsx_comm_init_h is COLLECTIVE (every rank calls it with the same
 *     unique id, rank r in 0 .. world-1) and binds to the calling thread's current HIP device;
 *   - the unique id (RSX_COMM_UNIQUE_ID_BYTES) is made by ONE rank and handed to the others by the caller (the launcher's
 *     key-value store);
 *   - rsx_all_gather / rsx_all_reduce_* only ENQUEUE on `stream` (no host synchronisation, no helper thread): legal under
 *     hipGraph stream capture, so a training step with its collectives is one graph;
 *   - buffers are device pointers; recv of rsx_all_gather holds world * bytes_per_rank bytes in rank order; send may not
 *     overlap recv unless send == recv + rank * bytes_per_rank (in place).
 * Errors: RSX_ECOMM, message by rsx_comm_last_error_h() (thread-local).                                                   */
#define RSX_COMM_UNIQUE_ID_BYTES 128
typedef void* rsx_comm_t;
int rsx_comm_available_h(int* version_out /* nullable: RCCL's NCCL_VERSION_CODE */);
const char* rsx_comm_last_error_h(void);
int rsx_comm_unique_id_h(void* id_out /* [RSX_COMM_UNIQUE_ID_BYTES] host */);
int rsx_comm_init_h(const void* unique_id, int rank, int world, rsx_comm_t* comm_out);
int rsx_comm_destroy_h(rsx_comm_t comm);
int rsx_comm_rank_world_h(rsx_comm_t comm, int* rank, int* world);
/* IndexedSlices aggregation / batch ids: rank r's `bytes_per_rank` bytes land at recv + r * bytes_per_rank on every rank. */
int rsx_all_gather(rsx_comm_t comm, const void* send, void* recv, size_t bytes_per_rank, rsx_stream_t stream);
/* Dense gradients: recv[i] = sum over ranks of send[i] (send == recv allowed: in place). */
int rsx_all_reduce_sum_f32(rsx_comm_t comm, const float* send, float* recv, size_t n, rsx_stream_t stream);
/* xdeepfm.py's step: the 3.3 MB of CIN filters take a true all-reduce (in place over grad[n]) BESIDE the all-gather of the
 * sparse block -- both in one RCCL group (one fused launch).  bytes_per_rank % 4 == 0.                                    */
int rsx_all_reduce_all_gather(rsx_comm_t comm, float* grad, size_t n, const void* send, void* recv, size_t bytes_per_rank,
                              rsx_stream_t stream);

/* dst[0 .. nbytes) = src[0 .. nbytes) by a kernel (16-byte aligned, nbytes % 16 == 0); src may be pinned HOST memory (it is
 * mapped into the device's address space): the captured windows of the streaming TRAIN path fetch their staged batches with
 * this launch as their first graph node, so that a window is one graph launch with no hipMemcpyAsync / copy-engine start-up /
 * cross-stream event in front of it.                                                                                      */
int rsx_copy_bytes(void* dst, const void* src, size_t nbytes, rsx_stream_t stream);
int rsx_adam_slice_run(const rsx_adam_slice* slice_h, rsx_stream_t stream);
/* The window sweep (nw > 0) applies its 1 + nw zero-gradient updates with packed square roots / divisions that are correctly
 * rounded on a restricted operand domain (csrc/adam_fast.h; elements outside it take the IEEE form).  Self-test of those forms
 * against the compiler's sqrtf and '/': EVERY float of the square root's domain; 2 * 4096 * 256 * div_iters pseudo-random
 * pairs over the division's domain; with exhaustive_div != 0 also all 2^46 pairs of mantissas at exponents 0 / 0 (~30 s).
 * counts (device, 4 x uint64): sqrt mismatches, random-division mismatches, random pairs tried, exhaustive-division
 * mismatches.  Synchronises the stream.                                                                                   */
int rsx_adam_fast_math_selftest(unsigned long long* counts, uint32_t seed, int div_iters, int exhaustive_div,
                                rsx_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Fused DNN tower + head, TRAIN step: per layer
 * tf.layers.dense(relu) -> tf.layers.batch_normalization(training=True) -> tf.layers.dropout
 * (deepfm/deepfm.py:103-107, xdeepfm/xdeepfm.py:188-191, dcn/dcn.py:146-149), the 1-unit layer and
 * logits of deepfm/deepfm.py:90-91,108-112 and the mean sigmoid-CE of fm/fm.py:146-149.
 * Workspaces (caller-owned), RT = ceil(B/16):
 *   a_l [B,N_l] relu outputs; fstat_l double[RT,2,N_l] partial (sum a, sum a^2) (B > 512: see below); bn_l [2,N_l] mean,rstd;
 *   mask_l [B,N_l] 1 keep / 0 drop, or NULL: the keep mask of layer l is then the counter-based hash
 *   hash32[element ^ key(seed, *rng_step, l)] >= rate*2^32, re-evaluated wherever it is needed (rng_step is a
 *   DEVICE uint32 that changes every step, e.g. word 3 of the Adam state); dy_l [B,N_l] grad wrt BN_l output;
 *   bstat_l double[RT,2,N_l] partial (sum dy, sum dy*xhat).
 * K_l % 4 == 0; head width N <= 256.
 * Layers WITHOUT batch-norm (din/din.py:132-137: dense(relu) -> dropout): pass gamma = beta = NULL for that layer on every
 * entry point (and bn_prev_out / dgamma / dbeta NULL); the statistics workspaces are still required (fstat_prev / bn_prev
 * non-NULL is what marks "there is a previous layer") but their contents are not used.
 * ------------------------------------------------------------------------------------------- */
/* Batches > 512 (round 6; rounds 1-5 folded the row partials with a launch per statistics buffer): fstat_l / bstat_l are then
 * FIXED-POINT accumulators, int64 [8][4][Npad_l] (Npad = N rounded up to 16; RSX_TOWER_FIXED_STATS_DOUBLES(N) doubles): 8 rows of
 * the planes hi(sum) | lo(sum) | hi(sum sq) | lo(sum sq) with value = (hi * 2^32 + lo) * 2^-52 -- producer workgroup b adds its partial sums
 * to row b & 7 with non-returning integer atomics (order-independent, hence deterministic; resolution 2.2e-16, |sum| < 8.8e12;
 * 8 rows because same-address atomics serialise), every consumer adds the 8 rows: no launch between producer and consumer.
 * The rows must be ZERO before the step's first producer adds to them.  No consumer can clear it (other workgroups of
 * the same launch still read it), so later launches of the same stream do, through `zero_stats` / `zero_n` (doubles; NULL / 0:
 * nothing) of rsx_tower_fwd_layer / rsx_gather_tower_fwd0 / rsx_tower_bwd_layer_defer: the FIRST layer's backward launch (the
 * step's last tower launch) clears every row of the tower except bstat_0, which it consumes itself; the first layer's forward
 * launch of the next step clears bstat_0.  Rows start zeroed (caller allocates them so).                                    */
#define RSX_TOWER_FIXED_STATS_MIN_B 513
#define RSX_TOWER_FIXED_STATS_DOUBLES(N) (8 * 4 * (((N) + 15) / 16 * 16))
/* a_out = relu(in' . W + bias); in' = in for the first layer (fstat_prev == NULL), else
 * dropout(BN(in)) with the previous layer's statistics reduced from fstat_prev (bn_prev_out receives them). */
int rsx_tower_fwd_layer(const float* in, const float* W, const float* bias, float* a_out, double* fstat_out,
                        const double* fstat_prev, const float* gamma_prev, const float* beta_prev,
                        const float* mask_prev, float* bn_prev_out, const uint32_t* rng_step, uint32_t seed,
                        int layer, float dropout_rate, int B, int K, int N, const rsx_sort_job* sort_h,
                        const rsx_adam_slice* sweep_h, double* zero_stats, int zero_n, rsx_stream_t stream);
/* rsx_gather_fm_fwd + the FIRST rsx_tower_fwd_layer in ONE launch (round 4): the input_layer lookup, first-order sum and FM
 * term of deepfm/deepfm.py:85-98 (= fm/fm.py:117-129) and a_0 = relu(E . W_0 + b_0) of deepfm/deepfm.py:103-104, K = F * D.
 * Every tile workgroup gathers its 16 examples' rows into LDS and feeds the MFMA A operand from there; E / S / y1 / y2 are
 * still written (backward and the head read them) and are bit-identical to rsx_gather_fm_fwd's, a_out / fstat_out to
 * rsx_tower_fwd_layer's.  Envelope (rsx_gather_tower_fwd0_supported): D == 16, F <= 64, B < 1024 -- RSX_EUNSUPPORTED outside,
 * where the caller runs the two entries above one after the other.  S / y1 / y2 / w1 may be NULL as in rsx_gather_fm_fwd. */
int rsx_gather_tower_fwd0(const float* tables, const float* w1, const int32_t* row_off, const int32_t* ids, float* E,
                          float* S, float* y1, float* y2, uint64_t w1_field_mask, int F, int D, const float* W,
                          const float* bias, float* a_out, double* fstat_out, int B, int N, const rsx_sort_job* sort_h,
                          const rsx_adam_slice* sweep_h, double* zero_stats, int zero_n, rsx_stream_t stream);
int rsx_gather_tower_fwd0_supported(int B, int F, int D);
/* o = dropout(BN(a_last)); u = o.wd + bd; z = wo[0]*act0(s0+c0) + wo[1]*s1 + wo[2]*act2(u) + bo (wo NULL: plain sum);
 * prob = sigmoid(z); per-row-tile partials of the loss and of every head gradient; dy_last / bstat_last = gradient
 * wrt the last BN output; gs0 / gs1 = d loss / d s0, d s1.  loss_scale = 1/(B*replicas).                    */
int rsx_tower_head(const float* a_last, const double* fstat_last, const float* gamma, const float* beta,
                   const float* mask, float* bn_out, const float* wd, const float* bd, const float* s0,
                   const float* c0, const float* s1, const float* wo, const float* bo, const float* labels,
                   float* prob, float* dy_last, double* bstat_last, float* dwd_part, double* hpart, float* gs0,
                   float* gs1, const uint32_t* rng_step, uint32_t seed, int layer, float dropout_rate,
                   float loss_scale, int relu0, int relu2, int B, int N, const rsx_adam_slice* sweep_h,
                   rsx_stream_t stream);
/* FM head with its backward (fm/fm.py:120-133,146-149): z = wo[0]*relu(y1 + c0) + wo[1]*y2 + bo; prob = sigmoid(z);
 * loss = mean sigmoid-CE; gy1 / gy2 = d loss / d y1, y2 (inputs of rsx_segsum_bwd); dwo[2], dbo, dc0.
 * loss_scale = 1/(B * replicas).  sweep_h (nullable): a slice of the untouched-row optimizer sweep carried by extra
 * workgroups.                                                                                                       */
int rsx_fm_head(const float* y1, const float* y2, const float* c0, const float* wo, const float* bo,
                const float* labels, float* prob, float* gy1, float* gy2, float* dwo, float* dbo, float* dc0,
                float* loss, float loss_scale, int B, const rsx_adam_slice* sweep_h, rsx_stream_t stream);
/* The same with the reduction left to the optimizer launch (round 4): terms != NULL -> terms [ceil(B/16), term_stride] receives
 * the rows of rsx_gather_fm_head's contract -- the same additions in the same order, so the two schedules train the same bits
 * (dwo / dbo / dc0 / loss are not written and may be NULL).  terms == NULL: rsx_fm_head.                                   */
int rsx_fm_head_terms(const float* y1, const float* y2, const float* c0, const float* wo, const float* bo,
                      const float* labels, float* prob, float* gy1, float* gy2, float* dwo, float* dbo, float* dc0,
                      float* loss, float* terms, int term_stride, int n_dense, int off_c0, int off_wo, int off_bo,
                      float loss_scale, int B, const rsx_adam_slice* sweep_h, rsx_stream_t stream);
/* Backward of layer l: BN backward + relu mask on load; writes dW, db, dgamma, dbeta, and dy_prev = gradient wrt
 * the previous layer's BN output (+ its bstat_prev partials), or dX for the first layer (bn_prev == NULL).
 * With hpart != NULL (last layer) one extra workgroup reduces the head partials into dwd, dbd, dwo[3], dbo, dc0, loss. */
int rsx_tower_bwd_layer(const float* in, const float* W, const float* a, const float* dy, const double* bstat,
                        const float* bn, const float* gamma, float* dW, float* db, float* dgamma, float* dbeta,
                        const float* bn_prev, const float* gamma_prev, const float* beta_prev,
                        const float* mask_prev, float* dy_prev, double* bstat_prev, const double* hpart,
                        const float* dwd_part, float* dwd, float* dbd, float* dwo, float* dbo, float* dc0,
                        float* loss, const uint32_t* rng_step, uint32_t seed, int layer, float dropout_rate, int B,
                        int K, int N, const rsx_sort_job* sort_h, const rsx_adam_slice* sweep_h, float* dw_partials,
                        rsx_stream_t stream);
/* dw_partials (nullable): rsx_tower_bwd_workspace_floats(B, K, N) floats that let a large batch (B >= 1024) split every
 * dW tile's reduction over the batch into up to 32 row blocks; a second small launch adds the partial tiles in ascending
 * block order (deterministic).  Without it one workgroup per tile walks the whole batch.                               */
size_t rsx_tower_bwd_workspace_floats(int B, int K, int N);
/* Large batches: a layer's weight gradient leaves rsx_tower_bwd_layer as partial sums over blocks of batch rows, added in
 * block order by a small second launch.  rsx_tower_bwd_layer_defer hands that reduction back as a job instead (reduce_out->sb
 * == 0: nothing to do); rsx_tower_reduce_dw_jobs runs the jobs of ALL layers in one launch at the end of the backward pass
 * (dW is first read by the optimizer).  Same arithmetic, same order per layer.                                          */
#define RSX_DW_REDUCE_MAX_JOBS 4
typedef struct {
  const float* partials;
  float* dW;
  float* db;
  int32_t sb, K, N;
  int32_t layout;            /* 0: [tiles][sb][256] tile-major, 1: [sb][K+1 rounded to 16][N rounded to 16] */
} rsx_dw_reduce_job;
struct rsx_tower_bwd_extra_;   /* (defined with the cross layers' entries below: the cross-layer backward riding in this launch) */
int rsx_tower_bwd_layer_defer(const float* in, const float* W, const float* a, const float* dy, const double* bstat,
                              const float* bn, const float* gamma, float* dW, float* db, float* dgamma, float* dbeta,
                              const float* bn_prev, const float* gamma_prev, const float* beta_prev, const float* mask_prev,
                              float* dy_prev, double* bstat_prev, const double* hpart, const float* dwd_part, float* dwd,
                              float* dbd, float* dwo, float* dbo, float* dc0, float* loss, const uint32_t* rng_step,
                              uint32_t seed, int layer, float dropout_rate, int B, int K, int N, const rsx_sort_job* sort_h,
                              const rsx_adam_slice* sweep_h, float* dw_partials, rsx_dw_reduce_job* reduce_out,
                              double* zero_stats, int zero_n, const struct rsx_tower_bwd_extra_* extra_h, rsx_stream_t stream);
int rsx_tower_reduce_dw_jobs(const rsx_dw_reduce_job* jobs_h, int njobs, rsx_stream_t stream);

/* sweep_h (host pointer, nullable, on all three tower entry points): a slice of the untouched-row optimizer sweep
 * (see rsx_adam_slice) executed by extra workgroups appended after the launch's own ones -- the HBM-bound stream fills the
 * CUs the latency-bound tower workgroups leave idle.
 * sort_h (host pointer, nullable): the step's rsx_field_sort job executed by F extra workgroups of this launch.  The
 * sort depends on ids only and is first consumed by rsx_segsum_bwd, so its latency hides behind the tower backward
 * instead of occupying its own slot on the critical path (same results as a separate rsx_field_sort call).        */

/* ---------------------------------------------------------------------------------------------
 * A tower WITHOUT batch-norm as ONE launch, din/din.py:130-147: 'mlp_layer' =
 * L x [tf.layers.dense(relu) -> tf.layers.dropout] -> dense(1); logit = that + s0 (din.py:139: the target item's bias);
 * loss = mean sigmoid cross-entropy (:146-147); forward AND backward of a 16-row tile by one workgroup with every layer's
 * weights in LDS, weight gradients as per-workgroup partials summed in workgroup order by a second launch (deterministic).
 * Replaces the 2L + 2 launches of rsx_tower_fwd_layer / rsx_tower_head / rsx_tower_bwd_layer for such a tower; the dropout
 * masks are the same counter-based hash (layer index l, element b * widths[l] + c), sums are associated differently.
 * Envelope (rsx_mlp_nobn_supported): 1 <= L <= 3, K0 and widths multiples of 4 and <= 112, 16-byte aligned arrays.
 * dW[l] [K_l, widths[l]], db[l] [widths[l]], dwout [widths[L-1]], dbout [1], loss [1] = sum(ce) / B, prob [B],
 * dX [B, K0] = d loss / d X, gs0 (nullable) [B] = d loss / d s0; loss_scale = 1 / (B * replicas) scales every gradient.
 * workspace: rsx_mlp_nobn_workspace_floats(B, K0, widths, L) floats.
 * ------------------------------------------------------------------------------------------- */
#define RSX_MLP_MAX_LAYERS 3
typedef struct {
  const float* X;                          /* [B, K0] */
  const float* W[RSX_MLP_MAX_LAYERS];
  const float* b[RSX_MLP_MAX_LAYERS];
  const float* masks[RSX_MLP_MAX_LAYERS];  /* nullable entries: explicit keep masks [B, widths[l]] (parity tests) */
  const float* wout;
  const float* bout;
  const float* s0;                         /* nullable [B] */
  const float* labels;                     /* [B] float */
  const uint32_t* rng_step;                /* device: the optimizer's step counter (dropout key) */
  float* prob;
  float* dX;
  float* gs0;                              /* nullable */
  float* workspace;
  float* dW[RSX_MLP_MAX_LAYERS];
  float* db[RSX_MLP_MAX_LAYERS];
  float* dwout;
  float* dbout;
  float* loss;
  uint32_t seed;
  float dropout_rate, loss_scale;
  int32_t B, K0, L;
  int32_t widths[RSX_MLP_MAX_LAYERS];
  int32_t defer_reduce;                    /* 1: rsx_mlp_nobn_train_step leaves the weight-gradient reduce to a later
                                            * rsx_mlp_nobn_reduce(step_h) -- on any stream ordered after the step's launch */
  int32_t reserved;
} rsx_mlp_step;
int rsx_mlp_nobn_supported(int K0, const int32_t* widths, int L);
size_t rsx_mlp_nobn_workspace_floats(int B, int K0, const int32_t* widths, int L);
int rsx_mlp_nobn_train_step(const rsx_mlp_step* step_h, rsx_stream_t stream);
int rsx_mlp_nobn_reduce(const rsx_mlp_step* step_h, rsx_stream_t stream);
/* The same reduce as a JOB another launch of the step carries as extra workgroups (rsx_din_pool_bwd_pair_ride): filled by
 * rsx_mlp_nobn_reduce_job from a step with defer_reduce = 1; plain data, valid while the step's buffers are.               */
typedef struct {
  const float* part;
  long long poff[RSX_MLP_MAX_LAYERS + 1];
  float* dW[RSX_MLP_MAX_LAYERS];
  float* db[RSX_MLP_MAX_LAYERS];
  float* dwout;
  float* dbout;
  float* loss;
  int32_t K[RSX_MLP_MAX_LAYERS], N[RSX_MLP_MAX_LAYERS];
  uint32_t e4_end[RSX_MLP_MAX_LAYERS];     /* float4 elements of the layer regions 0 .. q */
  uint32_t e4_last;                        /* float4 elements in all (0: nothing to do) */
  int32_t L, nwg, NPo, NL;
  double inv_B;
} rsx_mlp_reduce_job;
int rsx_mlp_nobn_reduce_job(const rsx_mlp_step* step_h, rsx_mlp_reduce_job* job_out);

/* ---------------------------------------------------------------------------------------------
 * DCN cross layers, dcn/dcn.py:132-142: x_{l+1} = (x_l . w_l) * x0 + x_l + b_l, all L
 * layers fused per example.  dim % 4 == 0, dim <= 1024, L <= 8.
 * ------------------------------------------------------------------------------------------- */
/* s[B,L] receives the per-layer scalars (saved for backward); xL (nullable) the final x_L; cz (nullable, needs wout)
 * the logit contribution <x_L, wout> of tf.layers.dense(concat[deep, x_L], 1) dcn/dcn.py:151-152.                  */
int rsx_cross_fwd(const float* x0, const float* W, const float* Bc, const float* wout, float* s, float* xL, float* cz,
                  int B, int dim, int L, rsx_stream_t stream);
/* The input_layer lookup (rsx_gather_fm_fwd without first-order / FM outputs; dcn/dcn.py:125-127) and rsx_cross_fwd in ONE launch:
 * a wave gathers an example's rows and runs the cross layers on the registers it holds (D = 16, F <= 64).  E [B, F*D] as the
 * gather writes it, s / cz as rsx_cross_fwd; the same bits as the two launches.                                             */
int rsx_gather_cross_fwd(const float* tables, const int32_t* row_off, const int32_t* ids, float* E, const float* W, const float* Bc,
                         const float* wout, float* s, float* cz, int B, int F, int D, int L, rsx_stream_t stream);
size_t rsx_cross_bwd_workspace_floats(int B, int dim, int L);
/* Given dxL [B,dim] and/or gz [B] (gradient of cz): dX (+)= d loss/d x0, dW[L,dim], dB[L,dim], dwout[dim].
 * x_1..x_L are recomputed from s.  workspace: rsx_cross_bwd_workspace_floats() floats.                             */
int rsx_cross_bwd(const float* x0, const float* W, const float* Bc, const float* s, const float* dxL, const float* gz,
                  const float* wout, float* dX, int accumulate, float* dW, float* dB, float* dwout, float* workspace,
                  int B, int dim, int L, rsx_stream_t stream);
/* The backward's second launch (the sum of the per-wave gradient partials into dW / dB / dwout) as a job: rsx_cross_bwd_defer
 * hands it back instead of launching it (reduce_out nullable: then exactly rsx_cross_bwd); rsx_cross_reduce_run runs it, or the
 * scatter's stage-A launch carries it as extra workgroups (rsx_segsum_partials_ride) -- only the optimizer needs these sums. */
typedef struct {
  const float* part;
  float* dW;
  float* dB;
  float* dwout;            /* nullable */
  int32_t RT, n, L, dim;   /* n == 0: nothing to do */
} rsx_cross_reduce_job;
int rsx_cross_bwd_defer(const float* x0, const float* W, const float* Bc, const float* s, const float* dxL, const float* gz,
                        const float* wout, float* dX, int accumulate, float* dW, float* dB, float* dwout, float* workspace,
                        int B, int dim, int L, rsx_cross_reduce_job* reduce_out, rsx_stream_t stream);
int rsx_cross_reduce_run(const rsx_cross_reduce_job* job_h, rsx_stream_t stream);
/* Round 6 (dcn.py): the cross layers' backward needs only the head's gradient gz (dcn/dcn.py:132-142 feeds the logits beside the
 * tower), so it can leave the step's dependent chain: as `extra_h` of the LAST tower layer's rsx_tower_bwd_layer_defer it runs as
 * extra workgroups of that launch (exactly rsx_cross_bwd_defer(x0, cW, cB, s, NULL, gz, wout, dX, accumulate = 0, ...): dX is
 * WRITTEN, the partials' reduce comes back through reduce_out), and the FIRST tower layer's launch is given accumulate_dx = 1 so
 * that its d(input) tiles add onto that dX (a + b = b + a: the same bits as the separate launch adding afterwards).  Needs at
 * least two tower layers (the two roles are different launches).  rsx_tower_bwd_cross_ride_supported says whether both launches
 * take the kernels that know the roles (both layers through the large-batch backward kernel: batch >= 4096 for a 100-wide layer; L == 3, dim % 4 == 0, no sort / sweep riders in the carrying launch).   */
typedef struct rsx_tower_bwd_extra_ {
  int32_t accumulate_dx;                 /* first layer only: dy_prev (= dX) += instead of = */
  const float* x0;                       /* NULL: no rider; else the arguments of rsx_cross_bwd_defer */
  const float* cW;
  const float* cB;
  const float* s;
  const float* gz;
  const float* wout;
  float* dX;
  float* dcW;
  float* dcB;
  float* dwout;
  float* workspace;
  int32_t dim, L;
  rsx_cross_reduce_job* reduce_out;      /* required with the rider */
} rsx_tower_bwd_extra;
int rsx_tower_bwd_cross_ride_supported(int B, int K_last, int N_last, int K_first, int N_first, int dim, int L);
/* Reductions of other launches of the TRAIN step that only the optimizer reads, carried by the scatter's stage-A launch
 * (rsx_segsum_partials_ride) as extra 256-thread workgroups beside its position tiles: the tower's dW partial-tile reductions
 * (the jobs rsx_tower_bwd_layer_defer hands back) and the cross layers' gradient reduce.  dcn.py at batch 4 096: two launches
 * (6.4 + 4.9 us) leave the step's dependent chain.                                                                       */
/* out[j] = sum over g < G of part[g * n + j] (the G partial gradient vectors of a persistent launch, in workgroup order:
 * 16 contiguous ranges of ceil(G / 16) partials summed one after the other, the 16 sub-sums then added in ascending order --
 * the association of rsx_din_attn_finish's reduce) */
typedef struct {
  const float* part;
  float* out;
  int32_t G, n;
} rsx_vec_reduce_job;
#define RSX_VEC_REDUCE_MAX_JOBS 2
typedef struct {
  rsx_dw_reduce_job dw[RSX_DW_REDUCE_MAX_JOBS];
  int32_t n_dw;
  int32_t n_vec;
  rsx_cross_reduce_job cross;    /* cross.n == 0: none */
  rsx_vec_reduce_job vec[RSX_VEC_REDUCE_MAX_JOBS];   /* din.py: the two attention blocks' weight-gradient partials */
} rsx_scatter_riders;
/* rsx_segsum_partials (the scatter's stage A, above) -- the same launch carrying `riders` (nullable / empty: exactly rsx_segsum_partials).  Where the launch's kernel variant has no
 * rider form (it has for D = 16 dX-only and D = 32 dX + first order: dcn.py, din.py) the riders run as their own launches first. */
int rsx_segsum_partials_ride(const float* tables, const float* S, const float* dX, const float* gy1, const float* gy2,
                             const int32_t* perm, const int32_t* seg_off, const int32_t* uniq_row,
                             const rsx_seg_partials* ws_h, uint64_t w1_field_mask, int B, int F, int D, int stride, int null_row,
                             const rsx_example_blocks* blocks_h, const rsx_scatter_riders* riders, rsx_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * DIN: attention-weighted masked history sum din/din.py:118-124 and the generic
 * sparse-row segment builder for tables whose entry count exceeds one LDS tile (din/din.py:96-105 lookups).
 * K in {4, 8, 16, 32, 64}.
 * ------------------------------------------------------------------------------------------- */
/* out[b,:] = sum_p H[b,p,:] * w[b,p] * (ids[b,p] > 0)      H [B,P,K] gathered rows, w [B,P], ids int32 [B,P]      */
int rsx_din_pool_fwd(const float* H, const float* w, const int32_t* ids, float* out, int B, int P, int K,
                     rsx_stream_t stream);
/* dH[b,p,:] (+)= dout[b,:] * w[b,p] * mask ;  dw[b,p] = <H[b,p,:], dout[b,:]> * mask                              */
int rsx_din_pool_bwd(const float* H, const float* w, const int32_t* ids, const float* dout, float* dH, float* dw,
                     int accumulate, int B, int P, int K, rsx_stream_t stream);
/* The same two with row strides (floats, multiples of 4): `out` / `dout` may be column slices of a wider [B, ld] matrix (the
 * 'mlp_layer' concat din/din.py:131 and its gradient, never materialised as copies), dH rows may be interleaved with another
 * table's gradient rows (the [entries, 2, K] value block of the two-table scatter).                                       */
int rsx_din_pool_fwd_ld(const float* H, const float* w, const int32_t* ids, float* out, int B, int P, int K, int ld_out,
                        rsx_stream_t stream);
int rsx_din_pool_bwd_ld(const float* H, const float* w, const int32_t* ids, const float* dout, float* dH, float* dw,
                        int accumulate, int B, int P, int K, int ld_dout, int ld_dH, rsx_stream_t stream);
/* Both histories of din.py (item ids, category ids) in ONE launch each way: the arguments of two rsx_din_pool_fwd_ld /
 * rsx_din_pool_bwd_ld calls that share B, P, K and the leading dimensions.                                            */
int rsx_din_pool_fwd_pair(const float* H0, const float* w0, const int32_t* ids0, float* out0, const float* H1,
                          const float* w1, const int32_t* ids1, float* out1, int B, int P, int K, int ld_out,
                          rsx_stream_t stream);
int rsx_din_pool_bwd_pair(const float* H0, const float* w0, const int32_t* ids0, const float* dout0, float* dH0, float* dw0,
                          const float* H1, const float* w1, const int32_t* ids1, const float* dout1, float* dH1, float* dw1,
                          int accumulate, int B, int P, int K, int ld_dout, int ld_dH, rsx_stream_t stream);
/* the same launch carrying `rider` (nullable: then exactly rsx_din_pool_bwd_pair) as extra workgroups */
int rsx_din_pool_bwd_pair_ride(const float* H0, const float* w0, const int32_t* ids0, const float* dout0, float* dH0, float* dw0,
                               const float* H1, const float* w1, const int32_t* ids1, const float* dout1, float* dH1, float* dw1,
                               int accumulate, int B, int P, int K, int ld_dout, int ld_dH, const rsx_mlp_reduce_job* rider,
                               rsx_stream_t stream);

/* Several plain row gathers (tf.gather / tf.nn.embedding_lookup of one table each, din/din.py:96-105) in ONE launch:
 * out[e, 0:K] = table[row_base + ids[e], 0:K], e < n, `out` rows ld_out floats apart (K a multiple of 4, or K == 1: see
 * ld_table).  Host array of <= 8 jobs.                                                                                    */
#define RSX_GATHER_MAX_JOBS 8
typedef struct {
  const float* table;
  const int32_t* ids;
  float* out;
  int64_t n;
  int32_t K, ld_out, row_base;
  int32_t ld_table;          /* K == 1 only (scalar rows of a 1-D variable): floats between its elements in `table` */
} rsx_gather_job;
int rsx_gather_rows_multi(const rsx_gather_job* jobs_h, int njobs, rsx_stream_t stream);

/* Sort keys of DIN's two id tables (item, category) for one step: keys2 [B*(P+1), 2] int32; entry e < B = the target lookup
 * of example e (i_id, i_cate: din/din.py:100-101), entry B + b*P + p = history position (b, p) (:105) with padding ids (<= 0,
 * :107) mapped to the table's dummy last row (see rsx_seg_partials.null_row).                                               */
int rsx_din_keys(const int32_t* i_id, const int32_t* i_cate, const int32_t* hist_i, const int32_t* hist_c, int B, int P,
                 int dummy_item_row, int dummy_cate_row, int32_t* keys2, rsx_stream_t stream);
/* rsx_din_keys + rsx_din_valid_rows of BOTH histories in two launches (the fused TRAIN step of din.py): keys2 as above;
 * rows_x / count_x / w_x as rsx_din_valid_rows (ascending list of the positions of history x that are not padding, its length
 * in count_x[0], count_x + 1: ceil(B*P/1024) ints of scratch, w_x nullable: zeroed at the padded positions).              */
int rsx_din_prepare(const int32_t* i_id, const int32_t* i_cate, const int32_t* hist_i, const int32_t* hist_c, int B, int P,
                    int dummy_item_row, int dummy_cate_row, int32_t* keys2, int32_t* rows_i, int32_t* count_i, float* w_i,
                    int32_t* rows_c, int32_t* count_c, float* w_c, rsx_stream_t stream);
/* The same with two options: keys_field_stride > 0 -- the keys are written FIELD-MAJOR ([2, keys_field_stride]: item keys,
 * then category keys), the layout rsx_field_sort_large_t takes without a transpose launch; labels_i64 / labels_f32 (both or
 * neither): the model_fn's cast of the labels to float32 (din/din.py:146) done by the same launch.                       */
int rsx_din_prepare2(const int32_t* i_id, const int32_t* i_cate, const int32_t* hist_i, const int32_t* hist_c, int B, int P,
                     int dummy_item_row, int dummy_cate_row, int32_t* keys2, int keys_field_stride, int32_t* rows_i,
                     int32_t* count_i, float* w_i, int32_t* rows_c, int32_t* count_c, float* w_c, const int64_t* labels_i64,
                     float* labels_f32, rsx_stream_t stream);
/* The same two launches carrying the step's row gathers (rsx_gather_rows_multi's jobs; din/din.py:96-105) as extra workgroups:
 * jobs [0, njobs_first) ride in the first launch, the others in the second -- the lookups depend on the batch's ids only, like
 * the prepare kernels, so the bandwidth-bound gather runs beside those two latency-bound launches instead of after them.    */
int rsx_din_prepare2_gather(const int32_t* i_id, const int32_t* i_cate, const int32_t* hist_i, const int32_t* hist_c, int B, int P,
                            int dummy_item_row, int dummy_cate_row, int32_t* keys2, int keys_field_stride, int32_t* rows_i,
                            int32_t* count_i, float* w_i, int32_t* rows_c, int32_t* count_c, float* w_c,
                            const int64_t* labels_i64, float* labels_f32, const rsx_gather_job* jobs_h, int njobs, int njobs_first,
                            rsx_stream_t stream);


/* Fused attention MLP of `_attention` (din/din.py:111-121): for every history position m = (b, p)
 *   w[m] = W2 . drop(relu(W1^T . drop(relu(W0^T . [h, q, h*q, h-q] + b0)) + b1)) + b2,   h = H[m,:], q = q[b,:]
 * without materialising the [B*P, 4K] concat; a1 [M,N1] / a2 [M,N2] (relu outputs before dropout) are saved for the
 * backward pass.  Dropout: mask1 / mask2 (0/1 keep masks, parity tests) or the counter hash of the fused tower
 * (rng_step, seed, layers layer0 and layer0+1).  Envelope: K in {16, 32}, N1 <= 80, N2 <= 48 (the reference hard-codes
 * 80, 40: din/din.py:85).                                                                                            */
int rsx_din_attn_fwd(const float* H, const float* q, const float* W0, const float* b0, const float* W1, const float* b1,
                     const float* W2, const float* b2, float* a1, float* a2, float* w, const float* mask1,
                     const float* mask2, const uint32_t* rng_step, uint32_t seed, int layer0, float dropout_rate,
                     const int32_t* rows, const int32_t* count, int B, int P, int K, int N1, int N2, rsx_stream_t stream);
/* rows / count (nullable pair, from rsx_din_valid_rows): evaluate only the listed history positions -- the padded ones
 * (id 0) are masked out of the weighted sum anyway (din/din.py:118-124) and receive no gradient, so half of a ragged batch
 * need not go through the MLP at all.  a1 / a2 / w stay indexed by the original position; entries of skipped positions are
 * not written (w: see rsx_din_valid_rows).                                                                            */
/* Row list of the positions with ids[b,p] > 0, ascending: rows int32 [B*P], count int32 [1 + ceil(B*P/1024)] (device;
 * count[0] = the number of rows, the rest is scratch).  w_zero_padded
 * (nullable, [B*P]): set to 0 at every padded position, so that rsx_din_pool_fwd -- which multiplies w by the mask -- never
 * meets an unwritten value.  Two small launches; B*P <= 2^24.                                                                */
int rsx_din_valid_rows(const int32_t* ids, int B, int P, int32_t* rows, int32_t* count, float* w_zero_padded,
                       rsx_stream_t stream);
/* Backward of rsx_din_attn_fwd given dw [B*P] = d loss / d w: dH [B*P, K] (the attention's share; the pooling's share
 * comes from rsx_din_pool_bwd), dq [B, K] (summed over the P positions), and grads = [dW0 (4K x N1) | db0 (N1) |
 * dW1 (N1 x N2) | db1 (N2) | dW2 (N2) | db2 (1)] as one flat array.  Persistent workgroups keep their weight-gradient
 * tiles in registers over all their 64-row blocks and write one partial each; partials are added in workgroup order.
 * workspace: rsx_din_attn_bwd_workspace_floats(B, P, K, N1, N2) floats.  Same masks / rng arguments as the forward.   */
size_t rsx_din_attn_bwd_workspace_floats(int B, int P, int K, int N1, int N2);
int rsx_din_attn_bwd(const float* H, const float* q, const float* W0, const float* W1, const float* W2, const float* a1,
                     const float* a2, const float* dw, float* dH, float* dq, float* grads, float* workspace,
                     const float* mask1, const float* mask2, const uint32_t* rng_step, uint32_t seed, int layer0,
                     float dropout_rate, int accumulate_dH, const int32_t* rows, const int32_t* count, const int32_t* ids,
                     int B, int P, int K, int N1, int N2, rsx_stream_t stream);
/* The same with row strides (floats): dH rows ld_dH apart (interleaved with another table's gradient rows), dq rows ld_dq
 * apart, and optionally dq = dq_add (rows ld_dq_add apart) + the attention's query gradient -- the target item embedding
 * also feeds the final MLP directly (din/din.py:131), so its two gradients leave as one row of the scatter's value block.  */
int rsx_din_attn_bwd_ld(const float* H, const float* q, const float* W0, const float* W1, const float* W2, const float* a1,
                        const float* a2, const float* dw, float* dH, float* dq, float* grads, float* workspace,
                        const float* mask1, const float* mask2, const uint32_t* rng_step, uint32_t seed, int layer0,
                        float dropout_rate, int accumulate_dH, const int32_t* rows, const int32_t* count, const int32_t* ids,
                        int B, int P, int K, int N1, int N2, int ld_dH, int ld_dq, const float* dq_add, int ld_dq_add,
                        rsx_stream_t stream);
/* din.py runs two attention blocks per step (item ids, category ids): the backward launch alone, leaving its per-row query
 * gradients and weight-gradient partials in `workspace` (rsx_din_attn_bwd_workspace_floats floats, ONE PER BLOCK) ...      */
int rsx_din_attn_bwd_nofinish(const float* H, const float* q, const float* W0, const float* W1, const float* W2,
                              const float* a1, const float* a2, const float* dw, float* dH, float* workspace,
                              const float* mask1, const float* mask2, const uint32_t* rng_step, uint32_t seed, int layer0,
                              float dropout_rate, int accumulate_dH, const int32_t* rows, const int32_t* count,
                              const int32_t* ids, int B, int P, int K, int N1, int N2, int ld_dH, rsx_stream_t stream);
/* ... and ONE launch that finishes both: grads_x (the packed [W0 | b0 | W1 | b1 | W2 | b2] gradient of block x) = the sum of its
 * partials in workgroup order, dq_x[b, :] = dq_add_x[b, :] (nullable) + sum over p of the per-row query gradients (padded
 * positions skipped through ids_x when the backward ran over a row list, else ids_x = NULL).                              */
int rsx_din_attn_finish_pair(const float* workspace0, float* grads0, float* dq0, const int32_t* ids0, const float* dq_add0,
                             const float* workspace1, float* grads1, float* dq1, const int32_t* ids1, const float* dq_add1,
                             int B, int P, int K, int N1, int N2, int ld_dq, int ld_dq_add, rsx_stream_t stream);
/* the same with the two weight-gradient reduces handed back as jobs (reduce_out[2], nullable: then exactly the entry above)
 * instead of run by the launch -- only the optimizer reads grads_x, so the scatter's stage-A launch can carry them
 * (rsx_scatter_riders.vec) and this launch is left with the query-gradient sums the scatter waits for.                    */
int rsx_din_attn_finish_pair_defer(const float* workspace0, float* grads0, float* dq0, const int32_t* ids0, const float* dq_add0,
                                   const float* workspace1, float* grads1, float* dq1, const int32_t* ids1, const float* dq_add1,
                                   int B, int P, int K, int N1, int N2, int ld_dq, int ld_dq_add, rsx_vec_reduce_job* reduce_out,
                                   rsx_stream_t stream);
int rsx_vec_reduce_run(const rsx_vec_reduce_job* jobs_h, int njobs, rsx_stream_t stream);

/* accumulate_dH != 0: dH += (rsx_din_pool_bwd has already written its share of the gradient of H into the same buffer).
 * rows / count / ids (nullable together): the forward's row list and the [B*P] ids it came from; only listed positions get
 * their dH written (accumulated) and contribute to dq.                                                                   */
/* Sorted row keys (stable sort done by the caller) -> uniq_row[U], seg_off[U+1], nuniq[0] = U and the row -> j slot
 * map (previous call's entries cleared first): the same workspace contract as rsx_field_sort with F = 1, so
 * rsx_segsum_bwd(F = 1, B = N, perm = the sort permutation) and the TABLE_TF1 Adam kind consume it unchanged.
 * segid (nullable; int32 [stride + 2 + ceil(stride/16)], stride >= N): the two-stage segment-sum workspace, as written by
 * rsx_field_sort.  scratch: int32 [ceil(N/1024) + 1].  Multi-workgroup: three small launches.                         */
int rsx_sorted_segments(const int32_t* sorted_keys, int N, int32_t* uniq_row, int32_t* seg_off, int32_t* nuniq,
                        int32_t* slot, int32_t* segid, int stride, int32_t* scratch, rsx_stream_t stream);
/* Ordered segment-sum of generic per-entry row gradients vals[N,K] (the F = 1 case of rsx_segsum_bwd).  null_row >= 0
 * names a padding row whose entries carry exactly-zero gradients by construction (DIN history padding id 0,
 * din/din.py:107): its segment is not walked and G = 0 is written for it; -1 = no such row.                        */
int rsx_segsum_rows(const float* vals, const int32_t* perm, const int32_t* seg_off, const int32_t* uniq_row,
                    const int32_t* nuniq, float* G, int N, int K, int stride, int null_row,
                    const rsx_seg_partials* partials_h, rsx_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * xDeepFM CIN layer, xdeepfm/xdeepfm.py:145-172, fp32 MFMA.  D must be 16, H <= 128.
 *   out[b,n,d] = relu( sum_{f,h} X0[b,f,d] * Xk[b,h,d] * W[f*H+h, n] + c[n] )     (f major, h minor: Appendix A-13)
 * ------------------------------------------------------------------------------------------- */
int rsx_cin_layer_fwd(const float* X0, const float* Xk, const float* W, const float* c, float* out, int B, int F,
                      int H, int N, int D, const rsx_adam_slice* sweep_h, rsx_stream_t stream);
/* dout = gradient wrt `out` (the relu mask is taken from `out`).  Writes dW[F*H,N], dc[N]; dXk[B,H,D] and dX0[B,F,D]
 * are overwritten or accumulated (acc_* != 0).  When Xk aliases X0 (first layer) pass either distinct dXk / dX0 buffers
 * or ONE buffer for both with acc_dx0 != 0 (the two gradient roles of X0 are then added in place).
 * gs / wout (nullable pair): the layer's direct connection into the 'cin_net' head (xdeepfm.py:175-182) --
 * gs[b] * wout[n], broadcast over d, is added to dout (dout itself may then be NULL: last layer).                    */
int rsx_cin_layer_bwd(const float* X0, const float* Xk, const float* W, const float* out, const float* dout,
                      const float* gs, const float* wout, float* dXk, int acc_dxk, float* dX0, int acc_dx0, float* dW,
                      float* dc, float* dpre_ws, int B, int F, int H, int N, int D, const rsx_adam_slice* sweep_h,
                      rsx_stream_t stream);
/* dpre_ws: rsx_cin_bwd_workspace_floats(B, F, H, N) floats of scratch (the relu-masked dout, written by the dX launch and
 * read by the dW launch, followed by the per-h-tile partial sums of dX0).                                             */
size_t rsx_cin_bwd_workspace_floats(int B, int F, int H, int N);
/* sweep_h (nullable): a slice of the untouched-row optimizer sweep carried by extra workgroups of the dW launch (the
 * MFMA-bound tiles leave HBM idle), as on the tower entry points.                                                  */
/* 'cin_net' output head, xdeepfm/xdeepfm.py:180-182 (concat of the L layer maps on axis 1, reduce_sum over d, dense(1, relu)):
 *   y[b] = relu(bout + sum_k sum_n Wout[off_k + n] * sum_d out_k[b,n,d]),  off_k = n_0 + .. + n_{k-1}
 * outs_h: HOST array of L device pointers [B, n_k, 16]; sizes_h: HOST array of the n_k.  L <= 8.  The concat is never
 * materialised.                                                                                                     */
int rsx_cin_out_fwd(const float* const* outs_h, const int32_t* sizes_h, int L, const float* Wout, const float* bout,
                    float* y, int B, int D, rsx_stream_t stream);
/* Backward of the head: gs[b] = gy[b] * (y[b] > 0) (consumed by rsx_cin_layer_bwd), dWout[off_k + n] =
 * sum_b gs[b] * sum_d out_k[b,n,d], dbout = sum_b gs[b]; fixed summation order.                                      */
int rsx_cin_out_bwd(const float* const* outs_h, const int32_t* sizes_h, int L, const float* y, const float* gy, float* gs,
                    float* dWout, float* dbout, int B, int D, rsx_stream_t stream);
/* The same launch with one more workgroup that computes the gradient of the numeric part of xdeepfm.py's linear_net kernel
 * (xdeepfm/xdeepfm.py:127,131): dwnum[j] = sum_b logx[b, j] * g_lin[b], j < nnum (fixed order) -- instead of a library
 * gemv launch of its own.  dwnum == NULL: exactly rsx_cin_out_bwd.                                                    */
int rsx_cin_out_bwd_lin(const float* const* outs_h, const int32_t* sizes_h, int L, const float* y, const float* gy, float* gs,
                        float* dWout, float* dbout, const float* logx, const float* g_lin, float* dwnum, int nnum, int B,
                        int D, rsx_stream_t stream);
/* xDeepFM's input side in one launch (xdeepfm/xdeepfm.py:125-131,185): the same ids gather the rows of TWO table sets
 * (E1 [B,F*D] for the CIN, E2 for the DNN: the script calls input_layer twice) and y1[b] = sum of the indicator weights
 * w1[row] over the fields of w1_field_mask + <num_x[b,:], num_w> -- the pre-activation of linear_net = dense([ND numeric
 * log-values | one-hot blocks], 1) without its bias.  ND <= 64.                                                        */
int rsx_gather_two_fwd(const float* tables1, const float* w1, const float* tables2, const int32_t* row_off,
                       const int32_t* ids, const float* num_x, const float* num_w, float* E1, float* E2, float* y1,
                       uint64_t w1_field_mask, int B, int F, int D, int ND, rsx_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Host-side ingest ("next" row f-1).  Host pointers only.
 * ------------------------------------------------------------------------------------------- */
/* FarmHash Fingerprint64 of n byte strings (concatenated in bytes_h, offs_h[n+1]); replaces the hash
 * inside categorical_column_with_hash_bucket fm/fm.py:89.  out_h[i] = Fingerprint64(s_i).          */
int rsx_hash_fp64_h(const uint8_t* bytes_h, const int64_t* offs_h, int64_t n, uint64_t* out_h);
uint64_t rsx_fingerprint64_h(const uint8_t* s_h, size_t n);
/* bucketized_column(numeric_column(log(x+shift))) fm/fm.py:76-79: idx = #boundaries <= logf(x+shift) */
int rsx_bucketize_log_h(const float* x_h, int64_t n, const float* boundaries_h, int nb, float shift,
                        int32_t* out_h);
uint32_t rsx_crc32c_h(const uint8_t* data_h, size_t n);
uint32_t rsx_masked_crc32c_h(const uint8_t* data_h, size_t n);

/* TFRecord framing scan of a whole shard image (tf.data.TFRecordDataset fm/fm.py:107): returns the record count
 * (payload offsets / lengths written up to max_records) or RSX_EDATA on truncation / crc mismatch.            */
int64_t rsx_tfrecord_index_h(const uint8_t* buf_h, size_t n, int64_t* offsets_h, int64_t* lengths_h,
                             int64_t max_records, int verify_crc);
/* tf.parse_single_example(feature_description) fm/fm.py:43-44,100-103 fused with the host half of input_layer:
 * records -> label[n], cont_log[n,13] (nullable), ids[n,F] in slot order.  Multi-threaded over records.
 * threads: low 16 bits = worker threads; bit 16 (0x10000) = the label feature `_c0` is optional and parses as 0 when
 * absent -- serialized Examples of a serving request (deepfm/grpc_client.py:50-76) carry no label.                  */
int rsx_criteo_parse_h(const uint8_t* buf_h, const int64_t* offsets_h, const int64_t* lengths_h, int64_t n,
                       const int32_t* slot_src_h, const int32_t* slot_rows_h, const float* bnd_h,
                       const int32_t* bnd_off_h, const float* shift_h, int F, float* label_h, float* cont_log_h,
                       int32_t* ids_h, int threads);
/* din/din.py:44-57: label, i_id, i_cate and the two VarLen histories densified / zero padded to P.               */
int rsx_din_parse_h(const uint8_t* buf_h, const int64_t* offsets_h, const int64_t* lengths_h, int64_t n, int P,
                    int64_t* label_h, int64_t* i_id_h, int64_t* i_cate_h, int64_t* hist_i_h, int64_t* hist_c_h,
                    int threads);
/* deepfm/deepfm.py:28-33,53-56 AS COMMITTED (two int64 features u_id / i_id + int64 label): first value of each named int64
 * feature, out_h[k*n + r]; and categorical_column_with_hash_bucket(dtype=int64) deepfm/deepfm.py:41,46: the key is formatted
 * as a decimal string (TF as_string) and hashed, id = Fingerprint64(str(key)) % buckets.                            */
int rsx_int64_features_parse_h(const uint8_t* buf_h, const int64_t* offsets_h, const int64_t* lengths_h, int64_t n,
                               const char* const* names_h, int k, int64_t* out_h, int threads);
int rsx_hash_int64_keys_h(const int64_t* keys_h, int64_t n, uint64_t buckets, int32_t* out_h);
/* Writers for synthetic shards of the two schemas (the reference's own sample shard is a missing blob).  Return bytes
 * written, or -(needed+16) when cap is too small.                                                               */
int64_t rsx_criteo_encode_h(const float* label_h, const float* cont_h, const uint8_t* cat_bytes_h,
                            const int64_t* cat_offs_h, int64_t n, uint8_t* out_h, int64_t cap);
int64_t rsx_din_encode_h(const int64_t* label_h, const int64_t* i_id_h, const int64_t* i_cate_h, const int64_t* hist_i_h,
                         const int64_t* hist_c_h, int64_t n, int P, int keep_padding, uint8_t* out_h, int64_t cap);

/* bf16 MFMA path of the CIN layer (north_star: "MFMA only on the CIN feature-map contraction where it is genuinely a
 * dense bf16 GEMM"; xdeepfm/xdeepfm.py:145-169).  Same contract as rsx_cin_layer_fwd / rsx_cin_layer_bwd, with the
 * filter W replaced by a bf16 image prepared once per step:
 *   rsx_cin_prep_bf16      W fp32 [F*H, N] -> w16 (rsx_cin_bf16_weight_elems(F,H,N) 16-bit elements, caller-owned):
 *                          the two zero-padded operand layouts (n-contiguous and h-contiguous) the kernels read with
 *                          16-byte loads
 *   rsx_cin_layer_fwd_bf16 Xk and W rounded to bf16 (RNE), fp32 accumulation, X0 / bias / relu in fp32
 *   rsx_cin_layer_bwd_bf16 dpre and W (and the products X0*Xk of the weight gradient) rounded to bf16, every sum fp32;
 *                          dc summed in fp32 from the unrounded dpre.  ws: rsx_cin_bf16_bwd_workspace_bytes(B, N) bytes.
 * fp32 (rsx_cin_layer_fwd/bwd) stays the parity path; the tolerance of this one is stated in DESIGN.md / the tests.  */
size_t rsx_cin_bf16_weight_elems(int F, int H, int N);
size_t rsx_cin_bf16_bwd_workspace_bytes(int B, int N);
int rsx_cin_prep_bf16(const float* W, void* w16, int F, int H, int N, rsx_stream_t stream);
int rsx_cin_layer_fwd_bf16(const float* X0, const float* Xk, const void* w16, const float* c, float* out, int B, int F,
                           int H, int N, int D, const rsx_adam_slice* sweep_h, rsx_stream_t stream);
int rsx_cin_layer_bwd_bf16(const float* X0, const float* Xk, const void* w16, const float* out, const float* dout,
                           const float* gs, const float* wout, float* dXk, int acc_dxk, float* dX0, int acc_dx0,
                           float* dW, float* dc, void* ws, int B, int F, int H, int N, int D,
                           const rsx_adam_slice* sweep_h, rsx_stream_t stream);
/* The same backward in two parts, so that a net of several layers launches the data gradients layer by layer (each needs
 * the previous one's output) and then ALL weight gradients together (they only depend on their own layer's dX launch:
 * one launch instead of L latency-bound ones).  ws: one rsx_cin_bf16_bwd_workspace_bytes(B, N) buffer PER LAYER, written
 * by the dx call and read by the dw call.  rsx_cin_prep_bf16_multi: the filters of all layers in one launch (<= 4).   */
typedef struct {
  const float* Xk;   /* [B, H, 16] input map of the layer */
  const void* ws;    /* the layer's workspace, filled by rsx_cin_layer_bwd_dx_bf16 */
  float* dW;         /* [F*H, N] */
  float* dc;         /* [N] */
  int32_t H, N;
  int32_t dc_rows;   /* rows of bias-gradient partials in ws: 0 = one per example pair (rsx_cin_layer_bwd_dx_bf16), B = one per
                        example (rsx_cin_layer_bwd_dx_bf16_parts) */
} rsx_cin_dw_job;
int rsx_cin_layer_bwd_dx_bf16(const float* X0, const float* Xk, const void* w16, const float* out, const float* dout,
                              const float* gs, const float* wout, float* dXk, int acc_dxk, float* dX0, int acc_dx0,
                              void* ws, int B, int F, int H, int N, int D, const rsx_adam_slice* sweep_h,
                              rsx_stream_t stream);
int rsx_cin_bwd_dw_bf16(const float* X0, const rsx_cin_dw_job* jobs_h, int njobs, int B, int F, int D,
                        const rsx_adam_slice* sweep_h, rsx_stream_t stream);
int rsx_cin_prep_bf16_multi(const float* const* W_h, void* const* w16_h, const int32_t* H_h, const int32_t* N_h, int L,
                            int F, rsx_stream_t stream);
/* Round 5: the CIN contraction on the bf16 matrix cores with SPLIT operands (csrc/cin_split.hip; xdeepfm/xdeepfm.py:145-169).
 * An fp32 value is the exact sum of three bf16 values; ns planes of every contraction operand are kept and the products
 * of planes i, j with i + j <= ns + 1 are accumulated (fp32) by v_mfma_f32_16x16x32_bf16:
 *   ns = 3: 6 MFMAs per k-step, every product exact to 2^-23 of itself -- fp32-grade, the parity path on the bf16 cores;
 *   ns = 2: 3 MFMAs, 2^-16-grade;  ns = 1: 1 MFMA, plain bf16 operands (the arithmetic of rsx_cin_layer_fwd_bf16);
 *   ns = 4: forward and data gradients with TWO fp16 planes per operand (v_mfma_f32_16x16x32_f16, 3 MFMAs per k-step); the
 *           operand of every accumulation chain is scaled by a power of two (per example / per field, divided out in fp32),
 *           products to 2^-22 -- held to the tolerances of ns = 3; the weight gradients run on three bf16 planes.
 * Same contract as rsx_cin_layer_fwd / rsx_cin_layer_bwd otherwise (X0 scaling, bias, relu and every sum in fp32).
 *   rsx_cin_split_prep  W fp32 [F*H, N] of L layers -> w16_h[k] (rsx_cin_split_weight_elems 16-bit elements each)
 * F <= 40, H, N <= 128, D = 16; RSX_EUNSUPPORTED otherwise.                                                          */
size_t rsx_cin_split_weight_elems(int F, int H, int N, int ns);
/* rsx_gather_two_fwd's arguments as a job: rsx_cin_split_prep_gather runs that lookup (xdeepfm/xdeepfm.py:125-131,185; D = 16) as
 * extra workgroups of the filter-preparation launch -- the two launches at the head of xdeepfm.py's step become one.      */
typedef struct rsx_gather_two_job {
  const float* tables1;
  const float* w1;
  const float* tables2;
  const int32_t* row_off;
  const int32_t* ids;
  const float* num_x;
  const float* num_w;
  float* E1;
  float* E2;
  float* y1;
  uint64_t w1_field_mask;
  int32_t B, F, D, ND;
} rsx_gather_two_job;
int rsx_cin_split_prep_gather(const float* const* W_h, void* const* w16_h, const int32_t* H_h, const int32_t* N_h, int L, int F,
                              int ns, const rsx_gather_two_job* g, rsx_stream_t stream);
int rsx_cin_split_prep(const float* const* W_h, void* const* w16_h, const int32_t* H_h, const int32_t* N_h, int L, int F,
                       int ns, rsx_stream_t stream);
int rsx_cin_split_fwd(const float* X0, const float* Xk, const void* w16, const float* c, float* out, int B, int F, int H,
                      int N, int D, int ns, rsx_stream_t stream);
/* Backward of the same layer.  rsx_cin_split_bwd_dx: the data gradients (contract of rsx_cin_layer_bwd_dx_bf16_parts: dXk
 * written or accumulated, dX0 left as one partial per 16-wide tile of h in dx0_parts [ceil(H/16)][B][F*16] for
 * rsx_cin_dx0_reduce); it also leaves dpre = relu'(out) * (dout + gs * wout) -- ns planes of operand fragments -- and the
 * bias gradient's per-example partial sums in ws (rsx_cin_split_bwd_workspace_bytes(B, N, ns) bytes, one buffer per layer).
 * acc_dxk = 2 (H == F, the first layer, whose dXk IS dX0 [B, F, D]; ns = 4): the fields are split over TWO workgroups per tile
 * of h; the first half writes dXk, the second half's share goes to dx0_parts tile ceil(H/16) -- one more [B][F*16] partial the
 * caller allocates behind the tiles and hands to the reduce.
 * rsx_cin_split_bwd_dw: dW [F*H, N] and dc [N] of several layers in ONE launch from those workspaces (rsx_cin_dw_job with
 * ws = the layer's workspace; dc_rows is ignored); the products X0 * Xk are formed in fp32, rounded once and split.    */
size_t rsx_cin_split_bwd_workspace_bytes(int B, int N, int ns);
int rsx_cin_split_bwd_dx(const float* X0, const float* Xk, const void* w16, const float* out, const float* dout,
                         const float* gs, const float* wout, float* dXk, int acc_dxk, float* dx0_parts, void* ws, int B,
                         int F, int H, int N, int D, int ns, rsx_stream_t stream);
int rsx_cin_split_bwd_dw(const float* X0, const rsx_cin_dw_job* jobs_h, int njobs, int B, int F, int D, int ns,
                         rsx_stream_t stream);
/* The same launch with rsx_cin_dx0_reduce riding along as extra workgroups (same sums in the same order: dX0 [B, F, D] =
 * (acc_dx0 ? dX0 : 0) + the nparts layers' tile partials, tiles_h[j] tiles each): one launch less on the step's chain.   */
int rsx_cin_split_bwd_dw_dx0(const float* X0, const rsx_cin_dw_job* jobs_h, int njobs, int B, int F, int D, int ns,
                             const float* const* parts_h, const int32_t* tiles_h, int nparts, float* dX0, int acc_dx0,
                             rsx_stream_t stream);
/* Round 5: the data gradients with EIGHT examples per workgroup (csrc/cin_bf16_wide.hip; xdeepfm/xdeepfm.py:145-169
 * differentiated).  A workgroup sees one 16-wide tile of h, so dX0 (a sum over h) is left as one partial per tile:
 * dx0_parts [ceil(H/16)][B][F*16] floats (rsx_cin_bf16_dx0_parts_floats); rsx_cin_dx0_reduce adds the tiles of all layers
 * in (layer, tile) order: dX0 = (acc ? dX0 : 0) + sum.  dXk as in rsx_cin_layer_bwd_dx_bf16 (the first layer's dXk may BE
 * dX0: the launch itself never touches dX0).  The workspace's bias-gradient partials are one row per example: the
 * weight-gradient job of this layer sets dc_rows = B.  RSX_EUNSUPPORTED when F > 40.                                   */
size_t rsx_cin_bf16_dx0_parts_floats(int B, int F, int H);
int rsx_cin_layer_bwd_dx_bf16_parts(const float* X0, const float* Xk, const void* w16, const float* out, const float* dout,
                                    const float* gs, const float* wout, float* dXk, int acc_dxk, float* dx0_parts, void* ws,
                                    int B, int F, int H, int N, int D, rsx_stream_t stream);
int rsx_cin_dx0_reduce(const float* const* parts_h, const int32_t* tiles_h, int njobs, float* dX0, int acc, int B, int F,
                       int D, rsx_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Streaming reader: the whole `input_fn` front end -- tf.data.TFRecordDataset(filenames)
 * .map(_parse_examples, num_parallel_calls).batch(batch_size)[.repeat(num_epochs)] of fm/fm.py:106-112
 * (deepfm/deepfm.py:60-70, xdeepfm/xdeepfm.py:101-118, dcn/dcn.py:106-112, din/din.py:61-80) -- as one host object:
 * files mmap'ed and touched once, a scanner thread on the record framing, `threads` workers that verify the masked
 * CRC-32C (SSE4.2) and parse each Example straight into its row of a batch buffer, `queue_batches` batches in flight.
 * Records of consecutive files form ONE stream; the final partial batch of an epoch is kept unless drop_remainder;
 * num_epochs < 0 repeats forever.  Data-parallel sharding (MirroredStrategy gives successive batches of the stream to
 * successive replicas, fm/fm.py:184-194): batch b of an epoch belongs to rank b % shard_world; only complete rounds of
 * shard_world FULL batches are delivered, so every rank runs the same number of equal-size steps.
 * drop_remainder == RSX_SHARD_TAIL (evaluation: no collective inside the loop): every batch goes to rank b % shard_world,
 * leftover full batches and the final partial one included -- the ranks together see every record exactly once.
 * next(): fills the caller's arrays (host; pinned on the training path) with the next batch of THIS rank, in stream
 * order; returns the number of rows (batch_size, or fewer for a final partial batch), 0 at the end of the data, or a
 * negative status code (RSX_EDATA: truncated file, crc mismatch, malformed Example, missing required feature --
 * TF raises DataLossError / InvalidArgument there).  open() returns NULL on invalid arguments.  One consumer thread. */
#define RSX_SHARD_TAIL 2            /* value of drop_remainder: sharded evaluation covers every record (see above) */
typedef struct rsx_reader rsx_reader;
rsx_reader* rsx_criteo_reader_open_h(const char* const* paths_h, int n_paths, const int32_t* slot_src_h,
                                     const int32_t* slot_rows_h, const float* bnd_h, const int32_t* bnd_off_h,
                                     const float* shift_h, int F, int batch_size, int num_epochs, int shard_rank,
                                     int shard_world, int drop_remainder, int threads, int verify_crc, int queue_batches);
int rsx_criteo_reader_next_h(rsx_reader* r, float* label_h /*[bs]*/, float* cont_log_h /*[bs,13], nullable*/,
                             int32_t* ids_h /*[bs,F]*/);
rsx_reader* rsx_din_reader_open_h(const char* const* paths_h, int n_paths, int P, int batch_size, int num_epochs,
                                  int shard_rank, int shard_world, int drop_remainder, int threads, int verify_crc,
                                  int queue_batches);
int rsx_din_reader_next_h(rsx_reader* r, int64_t* label_h, int64_t* i_id_h, int64_t* i_cate_h, int64_t* hist_i_h /*[bs,P]*/,
                          int64_t* hist_c_h /*[bs,P]*/);
int64_t rsx_reader_records_parsed_h(const rsx_reader* r);
void rsx_reader_close_h(rsx_reader* r);
/* the table-driven CRC-32C regardless of CPU support (rsx_crc32c_h uses the SSE4.2 instruction when present) */
uint32_t rsx_crc32c_table_h(const uint8_t* data_h, size_t n);

/* ---------------------------------------------------------------------------------------------
 * Streaming eval metrics: eval_metric_ops = {tf.metrics.auc(labels, pred),
 * tf.metrics.accuracy(labels, tf.round(pred))} fm/fm.py:150-153 + the Estimator's running mean of the batch
 * losses, as evaluate(steps=200) fm/fm.py:221 accumulates them.
 * One launch per eval batch adds the batch into `state` (uint64 [rsx_eval_metrics_state_words(T)], device,
 * zeroed by the caller before the first batch; read back once at the end):
 *   state[k], state[T+1+k], k = 0..T  examples with label 1 / label 0 whose prediction exceeds exactly k of the
 *                                     ascending thresholds (strict fp32 `pred > t`), so that
 *                                     tp[i] = sum_{k > i} state[k], fp[i] = sum_{k > i} state[T+1+k]
 *   state[2T+2] examples with rint(pred) == label (tf.round: half to even)   state[2T+3] examples
 *   state[2T+4] batches       state[2T+5] sum of batch_loss[0] over the launches (bits of a double; nullable input)
 * thresholds: fp32 [T] ascending (TF: -1e-7, i/(T-1), 1+1e-7), T <= 256.  Integer atomics only (deterministic). */
int rsx_eval_metrics_state_words(int num_thresholds);
int rsx_eval_metrics_update(const float* prob, const float* labels, const float* thresholds, int num_thresholds,
                            const float* batch_loss, uint64_t* state, int B, rsx_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* RSX_H_ */
