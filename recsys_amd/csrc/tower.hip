// Fused DNN tower + head for the CTR model_fn bodies on gfx950 (TRAIN step, forward and backward).
// Reference call sites: the `dnn` scope deepfm/deepfm.py:100-108 (twins xdeepfm/xdeepfm.py:184-192,
// dcn/dcn.py:144-149): per layer tf.layers.dense(relu) -> tf.layers.batch_normalization(training=True)
// -> tf.layers.dropout; the head deepfm/deepfm.py:90-91,108-112; the loss fm/fm.py:146-149.
//
// Why hand-written: at batch 256 the tower is ~110 MFLOP per step.  Library GEMMs pick 1-3-workgroup
// kernels for M=256 and the framework adds ~70 tiny launches; here the whole tower is 2L+1 launches:
//   tower_fwd_k   (per layer)  A-operand = dropout(BN(prev activations)) applied on load, fp32 MFMA
//                              16x16x4, epilogue bias+relu, per-row-tile column sums for the BN stats
//   tower_head_k               last BN+dropout, 1-unit layer, logits, sigmoid-CE loss, AND the head's
//                              backward (dz, grads of the scalar inputs, d(last activations))
//   tower_bwd_k   (per layer)  BN backward + relu mask applied on load; three tile families in one
//                              launch: d(input) [B x K], dW [K x N] (tiled over the OUTPUT so no
//                              partials), db/dgamma/dbeta
// Every cross-workgroup reduction goes through per-row-tile partial buffers that the consumer sums in
// a fixed order (doubles for the BN statistics): no atomics, deterministic, graph-replayable.
// fp32 MFMA (v_mfma_f32_16x16x4_f32) is an exact-fp32 fmaf chain, so parity with an fp32 restatement is
// at rounding-order level.  One wave per 16x16 output tile: these GEMMs are latency-bound, the lever is
// spreading tiles over the 256 CUs, not per-CU efficiency.
#include "rsx_common.h"
#include "sort_device.h"
#include "adam_device.h"
#include "drop_device.h"
#include "cross_device.h"
#include "gather_device.h"
#include "step_riders_device.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
RSX_STAMP_DECL

// The riders of a tower launch (a slice of the untouched-row optimizer sweep, the step's dedup sort).  -DRSX_ISA_PROBE compiles
// them out, so that scripts/isa_probe.sh can read a kernel's own instruction stream (never part of the product build).
#ifdef RSX_ISA_PROBE
#define RSX_RIDE_SWEEP(...) do { } while (0)
#define RSX_RIDE_SORT(...) do { } while (0)
#define RSX_RIDE_WINDOW(...) do { } while (0)
#else
#define RSX_RIDE_SWEEP(...) adam_block(__VA_ARGS__)
#define RSX_RIDE_SORT(...) field_sort_block(__VA_ARGS__)
// a slice of a full optimizer window's sweep (1 + 7 updates per untouched row) in small blocks (rsx_adam_slice.window_block_u = 2)
#define RSX_RIDE_WINDOW(...) adam_window_block<ADAM_WMAX, 2>(__VA_ARGS__)
#endif

constexpr float TOWER_BN_EPS = 1e-3f;  // tf.layers.batch_normalization default epsilon
constexpr int TM = 16;                 // rows per row tile

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// ---- batch-norm statistics of LARGE batches (B > 512): order-independent fixed-point accumulators (round 6) -----------------
// Small batches: every producer workgroup writes its row tile's partial column sums (sum, sum of squares) as one row of
// st[RT, 2, N] and every consumer workgroup adds the <= 32 rows itself in a fixed order.  At batch 4 096 there are 256 rows;
// rounds 1-5 folded them with a launch of their own per statistics buffer (tower_reduce_partials_k: 4 launches = 19.6 us of
// dcn.py's 191 us step, asked for removal four rounds in a row).  A grid-wide fold inside the producer needs a last-arriver
// hand-off (a returning atomic + an L2 write-back on every producer's tail); a floating-point atomic add
// would make the sums depend on the arrival order.  Integer addition does not: every producer adds its partial sums to ONE
// row as FIXED-POINT integers with plain (non-returning) 64-bit atomics -- bit-identical whatever the order, no launch, no
// tail.  Value v = s * 2^52 (exact: a power-of-two scale) is split into two limbs, hi = rint(v / 2^32) and lo = rint(v - hi *
// 2^32) in [-2^31, 2^31]: the resolution is 2^-52 = 2.2e-16 absolute (the fp64 sums it replaces round at ~1e-16 relative), the
// range |s| < 2^43 = 8.8e12, and 2^20 producers cannot overflow the low limb.  Layout of a fixed row: int64 [4][Npad] = the planes
// hi(sum), lo(sum), hi(sum sq), lo(sum sq) (8 such rows per buffer, see STAT_FIX_ROWS).  The rows must be ZERO when the step's first
// producer runs: the consumers cannot clear it
// (other workgroups of the same launch still read it), so a LATER launch of the same stream does -- the first layer's backward
// launch (the last tower launch of a step) clears every row except the one it consumes itself, and the first layer's forward
// launch of the next step clears that one (`zero_stats`: rsx_tower_fwd_layer / rsx_tower_bwd_layer).
constexpr int TOWER_FIX_MIN_B = 513;              // B >= this: fixed-point rows (== the old "pre-reduced" threshold)
constexpr double STAT_FIX_SCALE = 4503599627370496.0;        // 2^52
constexpr double STAT_FIX_LIMB = 4294967296.0;               // 2^32
// Same-address atomics serialise (~12 ns each: 256 producers on one row cost 3.5 us per launch, measured stand-alone --
// scripts/micro/atomic_shard.hip, profiles/r06_c_atomic_shard.txt -- and more at the tail of a real launch, where all producers
// finish together: dcn.py's step got 12 us SLOWER with one row); 8 rows, producer workgroup b adding to row b & 7, cost 0.3 us.
// A fixed buffer is therefore int64 [STAT_FIX_ROWS][4][Npad], Npad = N rounded up to 16 (rows start on 128-byte lines);
// consumers add the 8 rows as integers (exact, order-free).
constexpr int STAT_FIX_ROWS = 8;
__host__ __device__ __forceinline__ int stat_fix_npad(const int N) { return (N + 15) & ~15; }
__device__ __forceinline__ void stat_fix_add(double* __restrict__ st, const int N, const int col, const double s1, const double s2) {
  const int NP = stat_fix_npad(N);
  unsigned long long* a = reinterpret_cast<unsigned long long*>(st) + (size_t)(blockIdx.x & (STAT_FIX_ROWS - 1)) * 4 * NP;
  const double v1 = s1 * STAT_FIX_SCALE, v2 = s2 * STAT_FIX_SCALE;
  const double h1 = rint(v1 * (1.0 / STAT_FIX_LIMB)), h2 = rint(v2 * (1.0 / STAT_FIX_LIMB));
  const long long l1 = __double2ll_rn(v1 - h1 * STAT_FIX_LIMB), l2 = __double2ll_rn(v2 - h2 * STAT_FIX_LIMB);
  // (results unused: the compiler emits the non-returning form -- fire and forget, nothing on the workgroup's tail)
  // The four limbs live in four PLANES of the row (limb q of column c at [q][c]): a wave's atomics of one limb go to
  // consecutive words.  With a column's limbs side by side instead ([c][4], so that a consumer fetches them with two 16-byte
  // loads) the producers' atomics of four limbs hit the same 32 bytes one after the other: dcn.py 0.1913 against 0.1821 ms
  // (ABAB on one box, profiles/r06_h_stat_layout_ab.txt) -- the consumers' 32 8-byte loads cost nothing measurable.
  atomicAdd(a + col, (unsigned long long)__double2ll_rn(h1));
  atomicAdd(a + NP + col, (unsigned long long)l1);
  atomicAdd(a + 2 * NP + col, (unsigned long long)__double2ll_rn(h2));
  atomicAdd(a + 3 * NP + col, (unsigned long long)l2);
}
__device__ __forceinline__ void stat_fix_read(const double* __restrict__ st, const int N, const int col, double& s1, double& s2) {
  const int NP = stat_fix_npad(N);
  const long long* a = reinterpret_cast<const long long*>(st) + col;
  long long t[STAT_FIX_ROWS][4];
#pragma unroll
  for (int r = 0; r < STAT_FIX_ROWS; ++r)        // all 32 loads in flight: one memory round trip
#pragma unroll
    for (int q = 0; q < 4; ++q) t[r][q] = a[((size_t)r * 4 + q) * NP];
  long long h1 = 0, l1 = 0, h2 = 0, l2 = 0;
#pragma unroll
  for (int r = 0; r < STAT_FIX_ROWS; ++r) {
    h1 += t[r][0]; l1 += t[r][1]; h2 += t[r][2]; l2 += t[r][3];
  }
  s1 = ((double)h1 * STAT_FIX_LIMB + (double)l1) * (1.0 / STAT_FIX_SCALE);
  s2 = ((double)h2 * STAT_FIX_LIMB + (double)l2) * (1.0 / STAT_FIX_SCALE);
}
// one producer's partial column sums: a row of the partials array, or an addition to the fixed row
__device__ __forceinline__ void stat_put(double* __restrict__ st, const int RT, const int row, const int N, const int col,
                                         const double s1, const double s2) {
  if (RT == 0) {
    stat_fix_add(st, N, col, s1, s2);
  } else {
    st[((size_t)row * 2 + 0) * N + col] = s1;
    st[((size_t)row * 2 + 1) * N + col] = s2;
  }
}
// clears fixed rows for the step's later producers (see above): the launch's first STAT_ZERO_WGS workgroups take a slice each
// (n % 2 == 0, 16-byte aligned: the host lays the buffers out so)
constexpr int STAT_ZERO_WGS = 8;
__device__ __forceinline__ void stat_zero(double* __restrict__ z, const int n) {
  if ((int)blockIdx.x >= STAT_ZERO_WGS) return;
  double2* z2 = reinterpret_cast<double2*>(z);
  for (int e = (int)blockIdx.x * (int)blockDim.x + (int)threadIdx.x; e < (n >> 1); e += STAT_ZERO_WGS * (int)blockDim.x)
    z2[e] = make_double2(0.0, 0.0);
}

// fixed-order fp64 sum of the RT per-row-tile partials of one column (RT == 0: the fixed-point row of a large batch).  All loads of a 16-tile block are issued before the
// first add (batch 256 = 16 row tiles = ONE memory round trip per consumer instead of two; every tower launch starts with
// this reduction, so the round trip is on the step's critical path five times).
__device__ __forceinline__ void col_partials(const double* __restrict__ st, int RT, int N, int col, double& s1,
                                             double& s2) {
  if (RT == 0) {
    stat_fix_read(st, N, col, s1, s2);
    return;
  }
  s1 = 0.0;
  s2 = 0.0;
  int r = 0;
  for (; r + 16 <= RT; r += 16) {
    double t1[16], t2[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      t1[u] = st[((size_t)(r + u) * 2 + 0) * N + col];
      t2[u] = st[((size_t)(r + u) * 2 + 1) * N + col];
    }
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      s1 += t1[u];
      s2 += t2[u];
    }
  }
  for (; r + 8 <= RT; r += 8) {
    double t1[8], t2[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      t1[u] = st[((size_t)(r + u) * 2 + 0) * N + col];
      t2[u] = st[((size_t)(r + u) * 2 + 1) * N + col];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      s1 += t1[u];
      s2 += t2[u];
    }
  }
  for (; r < RT; ++r) {
    s1 += st[((size_t)r * 2 + 0) * N + col];
    s2 += st[((size_t)r * 2 + 1) * N + col];
  }
}

// mean / rstd of one column from the per-row-tile partial sums (sum a, sum a^2)
// (inv_B = 1.0 / B from the host: the two fp64 divisions cost ~80 instructions on the critical path of every workgroup of
// every launch that consumes batch-norm statistics -- a lone wave issues ~270 per microsecond)
__device__ __forceinline__ void bn_col_stats(const double* __restrict__ fstat, int RT, int N, int col, double inv_B,
                                             float& mean, float& rstd) {
  double s1, s2;
  col_partials(fstat, RT, N, col, s1, s2);
  const double mu = s1 * inv_B;
  double var = s2 * inv_B - mu * mu;
  if (var < 0.0) var = 0.0;
  mean = (float)mu;
  rstd = 1.0f / sqrtf((float)var + TOWER_BN_EPS);
}

// One 16x16 output tile per 256-thread workgroup: the 4 waves split the K loop (k-steps of 16
// interleaved), each wave keeps two independent accumulators (fp32 MFMA 16x16x4 has a 40-cycle
// dependent latency) and issues the loads of two k-steps before their 8 MFMAs.  The 4 partial tiles
// are summed through LDS in wave order by thread t -> element (t/16, t%16).  `ld(ks, a, b)` fills the
// lane's four A / B operands of k-step ks (operand t <-> k = 16*ks + 4*(lane>>4) + t on both sides).
template <class Ld>
__device__ __forceinline__ float tile_ksplit(int nsteps, float* part /* LDS [4][256] */, Ld ld) {
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
  int ks = w;
  // deep reductions (the first layer's K = 624: 10 k-steps per wave): the loads of FOUR k-steps are issued before their
  // 16 MFMAs, so a wave needs 3 memory round trips instead of 5; the summation order (acc0: even, acc1: odd visits)
  // is the one of the two-step loop below, which finishes the tail
  for (; ks + 12 < nsteps; ks += 16) {
    float a0[4], b0[4], a1[4], b1[4], a2[4], b2[4], a3[4], b3[4];
    ld(ks, a0, b0);
    ld(ks + 4, a1, b1);
    ld(ks + 8, a2, b2);
    ld(ks + 12, a3, b3);
#pragma unroll
    for (int t = 0; t < 4; ++t) acc0 = mfma16(a0[t], b0[t], acc0);
#pragma unroll
    for (int t = 0; t < 4; ++t) acc1 = mfma16(a1[t], b1[t], acc1);
#pragma unroll
    for (int t = 0; t < 4; ++t) acc0 = mfma16(a2[t], b2[t], acc0);
#pragma unroll
    for (int t = 0; t < 4; ++t) acc1 = mfma16(a3[t], b3[t], acc1);
  }
  for (; ks < nsteps; ks += 8) {
    float a0[4], b0[4], a1[4], b1[4];
    ld(ks, a0, b0);
    const bool two = ks + 4 < nsteps;
    if (two) ld(ks + 4, a1, b1);
#pragma unroll
    for (int t = 0; t < 4; ++t) acc0 = mfma16(a0[t], b0[t], acc0);
    if (two) {
#pragma unroll
      for (int t = 0; t < 4; ++t) acc1 = mfma16(a1[t], b1[t], acc1);
    }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) part[w * 256 + ((lane >> 4) * 4 + r) * 16 + (lane & 15)] = acc0[r] + acc1[r];
  __syncthreads();
  return ((part[tid] + part[256 + tid]) + part[512 + tid]) + part[768 + tid];
}

// ---------------------------------------------------------------------------------------------
// forward layer:  a_out = relu(in' . W + bias),  in' = in (first layer) or dropout(BN(in)) (others)
// grid = (ceil(N/16), ceil(B/16)), block = 256 (one 16x16 tile, K split over 4 waves).
// dyn LDS: 2*K + 1024 + 256 floats.
// ---------------------------------------------------------------------------------------------
struct FwdArgs {
  const float* in;        // [B, K]
  const float* W;         // [K, N]
  const float* bias;      // [N]
  float* a_out;           // [B, N]
  double* fstat_out;      // [RT, 2, N]
  // previous layer's BN + dropout (null for the first layer)
  const double* fstat_prev;  // [RT, 2, K]
  const float* gamma_prev;
  const float* beta_prev;
  const float* mask_prev;    // [B, K] 1 keep / 0 drop (null: RNG or no dropout)
  float* bn_prev_out;        // [2, K] mean, rstd (written by block (0,0) for the backward pass)
  const uint32_t* rng_step;
  uint32_t seed, layer_prev;
  float rate;
  int B, K, N, RT;
  double inv_B;            // 1.0 / B
  int ct, n_own;          // column tiles; ct * row tiles = workgroups of the layer itself
  int n_sort;             // extra workgroups running the step's per-field dedup sort (0: none), before the sweep slice
  double* zero_p;         // fixed-point statistics rows this launch clears for the step's later producers (stat_zero), or null
  int zero_n;
  SortArgs sort;
  AdamSlice sweep;        // optional slice of the untouched-row optimizer sweep carried as extra workgroups
};

// epilogue of a forward tile: thread t owns element (r = t/16, c = t%16) of the 16x16 tile (bx, by): bias + relu, the store,
// and the tile's column sums (sum a, sum a^2) for the batch-norm statistics
__device__ __forceinline__ void fwd_tile_epilogue(const FwdArgs& p, const float v, const int bx, const int by, double* cred) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int orow = by * TM + (tid >> 4), ocol = bx * 16 + (tid & 15);
  double s1 = 0.0, s2 = 0.0;
  if (orow < p.B && ocol < p.N) {
    float o = v + p.bias[ocol];
    o = o > 0.f ? o : 0.f;
    p.a_out[(size_t)orow * p.N + ocol] = o;
    s1 = (double)o;
    s2 = (double)o * (double)o;
  }
  if (p.fstat_out != nullptr) {   // column sums over the tile's 16 rows: 4 rows per wave, then 4 waves
    s1 += __shfl_xor(s1, 16); s2 += __shfl_xor(s2, 16);
    s1 += __shfl_xor(s1, 32); s2 += __shfl_xor(s2, 32);
    if (lane < 16) {
      cred[((tid >> 6) * 2 + 0) * 16 + lane] = s1;
      cred[((tid >> 6) * 2 + 1) * 16 + lane] = s2;
    }
    __syncthreads();
    if (tid < 16 && ocol < p.N)
      stat_put(p.fstat_out, p.RT, by, p.N, ocol, ((cred[0 * 16 + tid] + cred[2 * 16 + tid]) + cred[4 * 16 + tid]) + cred[6 * 16 + tid],
               ((cred[1 * 16 + tid] + cred[3 * 16 + tid]) + cred[5 * 16 + tid]) + cred[7 * 16 + tid]);
  }
}

// RID = false: a launch without riders (the optimizer-window form of the step): their code -- most of the kernel's
// instructions and registers -- is compiled out
template <bool RID>
__global__ __launch_bounds__(256) void tower_fwd_k(const FwdArgs p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  if (p.zero_n > 0) stat_zero(p.zero_p, p.zero_n);
  if (RID && (int)blockIdx.x >= p.n_own + p.n_sort) {
    RSX_RIDE_SWEEP(p.sweep.args, p.sweep.blk_lo + (blockIdx.x - p.n_own - p.n_sort));
    return;
  }
  if (RID && (int)blockIdx.x >= p.n_own) {   // piggy-backed dedup sort (ids only; first consumed by later launches)
    RSX_RIDE_SORT(p.sort, blockIdx.x - p.n_own, reinterpret_cast<uint32_t*>(lds));
    return;
  }
  const int bx = blockIdx.x % p.ct, by = blockIdx.x / p.ct;
  const int st0 = p.fstat_prev == nullptr ? 4 : 0;
  RSX_STAMP(st0 + 0, blockIdx.x == 0);
  float* sc = lds;                    // [K] scale
  float* sh = lds + p.K;              // [K] shift
  float* part = lds + 2 * p.K;        // [4][256]
  double* cred = reinterpret_cast<double*>(part + 1024);   // [4][2][16] column-sum exchange (8-byte aligned: K%4==0)
  const int tid = threadIdx.x, lane = tid & 63;
  const bool first = p.fstat_prev == nullptr;
  // (before the statistics reduction: the step counter it reads is one more memory round trip, and behind the barrier below it
  // used to start only when the reduction's had finished)
  const DropRng dr = drop_make(first ? 0.f : p.rate, p.mask_prev, p.rng_step, p.seed, p.layer_prev);
  if (!first) {
    for (int k = tid; k < p.K; k += 256) {
      if (p.gamma_prev == nullptr) {   // previous layer without batch-norm (din/din.py MLP): dropout only
        sc[k] = 1.f;
        sh[k] = 0.f;
        continue;
      }
      float mean, rstd;
      bn_col_stats(p.fstat_prev, p.RT, p.K, k, p.inv_B, mean, rstd);
      const float inv = rstd * p.gamma_prev[k];
      sc[k] = inv;
      sh[k] = p.beta_prev[k] - mean * inv;
      if (bx == 0 && by == 0) {
        p.bn_prev_out[k] = mean;
        p.bn_prev_out[p.K + k] = rstd;
      }
    }
    __syncthreads();
  }
  RSX_STAMP(st0 + 1, blockIdx.x == 0);
  const int i = lane & 15, kq = lane >> 4;
  const int row = by * TM + i;
  const int col = bx * 16 + i;
  const bool rok = row < p.B, cok = col < p.N;
  const size_t irow = (size_t)(rok ? row : p.B - 1) * p.K;
  const int colc = cok ? col : 0;
  const float rokf = rok ? 1.f : 0.f;
  const float v = tile_ksplit((p.K + 15) / 16, part, [&](int ks, float* a, float* b) {
    // operand loads unconditional on clamped indices; a k-step past K is neutralised by zeroing its A operand (a guarded
    // load is compiled into a branch of its own and the loads then complete one after the other)
    const int kk = ks * 16 + 4 * kq;
    const int kc = kk < p.K ? kk : p.K - 4;
    const float okf = kk < p.K ? rokf : 0.f;
    const float* w = p.W + (size_t)kc * p.N + colc;
    b[0] = w[0];
    b[1] = w[p.N];
    b[2] = w[2 * (size_t)p.N];
    b[3] = w[3 * (size_t)p.N];
    float4 av = *reinterpret_cast<const float4*>(p.in + irow + kc);
    if (!first) {
      const float4 s = *reinterpret_cast<const float4*>(sc + kc);
      const float4 h = *reinterpret_cast<const float4*>(sh + kc);
      const size_t e = irow + kc;
      av.x = (av.x * s.x + h.x) * drop_mul(dr, p.mask_prev, e);
      av.y = (av.y * s.y + h.y) * drop_mul(dr, p.mask_prev, e + 1);
      av.z = (av.z * s.z + h.z) * drop_mul(dr, p.mask_prev, e + 2);
      av.w = (av.w * s.w + h.w) * drop_mul(dr, p.mask_prev, e + 3);
    }
    a[0] = av.x * okf; a[1] = av.y * okf; a[2] = av.z * okf; a[3] = av.w * okf;
  });
  RSX_STAMP(st0 + 2, blockIdx.x == 0);
  fwd_tile_epilogue(p, v, bx, by, cred);
  RSX_STAMP(st0 + 3, blockIdx.x == 0);
}

// ---------------------------------------------------------------------------------------------
// gather + FIRST forward layer in one launch (round 4; deepfm.py at batch < 1024, D = 16): the input_layer lookup
// (fm/fm.py:118 = deepfm/deepfm.py:85), the first-order sum and the FM term (deepfm/deepfm.py:88-98) and
// a_0 = relu(E . W_0 + b_0) (deepfm/deepfm.py:103-104).  Workgroup roles by blockIdx.x:
//   [0, n_own)       tile (bx, by): gathers the rows of its 16 examples STRAIGHT INTO LDS (wave w the examples 16 by + 4 w +
//                    {0..3}, 16 row loads per lane in flight) and feeds the MFMA A operand from there.  The wave's B operands
//                    (its k-steps of W_0) are requested first and stay in registers, so nothing behind the gather touches
//                    global memory.  E never makes the HBM round trip on the step's critical path.
//   [.., + n_gout)   the stand-alone gather (gather_device.h, one wave per example): E for backward's dW, S, y1, y2 -- off
//                    the tile workgroups' path (phase stamps: as part of the column-tile-0 workgroups these outputs made
//                    them 2.7 us longer than the other tiles, and the launch as long as the two it replaces).
//   then the riders every tower launch takes: the step's dedup sort, a slice of the untouched-row optimizer sweep.
// The 7 column tiles gather the same rows: 7 x 0.64 MB of L2 reads -- what a kernel boundary costs in time many times over.
// dyn LDS: 16 * (K + 4) + 1024 + 256 floats.
// ---------------------------------------------------------------------------------------------
struct GatherFwdArgs {
  FwdArgs f;                 // in unused, fstat_prev == nullptr, K = F * 16
  const float* tables; const float* w1; const int32_t* row_off; const int32_t* ids;
  float* E; float* S; float* y1; float* y2;
  uint64_t w1_mask;
  int F, n_gout;
};

// NS = k-steps per wave (K / 16 = F k-steps dealt to the 4 waves); RID = false: no sort / sweep workgroups in the launch (the
// optimizer-window form of the step), their code -- and its registers -- compiled out
template <int NS, bool RID>
__global__ __launch_bounds__(256, 1) void tower_gather_fwd_k(const GatherFwdArgs g) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const FwdArgs& p = g.f;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  if (p.zero_n > 0) stat_zero(p.zero_p, p.zero_n);
  if ((int)blockIdx.x >= p.n_own) {
    const int r = blockIdx.x - p.n_own;
    if (r < g.n_gout) {
      const int b = r * 4 + w;
      if (b < p.B) gather_fm_example<16>(g.tables, g.w1, g.row_off, g.ids, g.E, g.S, g.y1, g.y2, g.w1_mask, b, g.F, lane);
    } else if (RID) {
      if (r < g.n_gout + p.n_sort) RSX_RIDE_SORT(p.sort, r - g.n_gout, reinterpret_cast<uint32_t*>(lds));     // (ids only)
      else RSX_RIDE_SWEEP(p.sweep.args, p.sweep.blk_lo + (r - g.n_gout - p.n_sort));
    }
    return;
  }
  constexpr int D = 16, LPR = 4, PPP = 16;
  const int bx = blockIdx.x % p.ct, by = blockIdx.x / p.ct;
  RSX_STAMP(4, blockIdx.x == 0);
  const int ES = p.K + 4;              // row stride of the LDS tile: 16-byte aligned, the 16 rows of an A-operand read start
  float* et = lds;                     // in 16 different 4-bank groups (K = 624: stride 628 = 52 mod 64)
  float* part = lds + 16 * ES;         // [4][256]
  double* cred = reinterpret_cast<double*>(part + 1024);
  const int F = g.F;                   // = the number of k-steps (K = 16 F)
  // ---- B operands of this wave's k-steps ks = w, w + 4, ... (operand t <-> k = 16 ks + 4 (lane >> 4) + t) ----
  const int i = lane & 15, kq = lane >> 4;
  const int col = bx * 16 + i;
  const int colc = col < p.N ? col : 0;
  // (32-bit element offsets from the uniform base: one v_add per load -- the launch is bound by instruction issue, every
  // 64-bit address costs 3-4 instructions of a lone wave's ~270 per microsecond)
  float bw[NS][4];
  const float* __restrict__ Wg = p.W;
  const uint32_t Nn = (uint32_t)p.N;
  // (F < 4, i.e. K < 64 -- test shapes only: a wave whose FIRST k-step lies past K falls back to k-step 0's rows.  Unclamped,
  // wave 3 of an F = 1 tile read 3 KB past a 1 KB weight matrix: harmless inside the allocator's segment, a GPU memory fault
  // when the matrix is the segment's last block -- one abort in ~8 runs of the GPU suite, found by soaking it in round 6)
  const uint32_t woff = (uint32_t)((w < F ? w : 0) * 16 + 4 * kq) * Nn + (uint32_t)colc;
#pragma unroll
  for (int sI = 0; sI < NS; ++sI) {
    const uint32_t o = w + 4 * sI < F ? woff + (uint32_t)(sI * 64) * Nn : woff;   // (a k-step past K is loaded from a valid
    bw[sI][0] = Wg[o];                                                           // address and never multiplied)
    bw[sI][1] = Wg[o + Nn];
    bw[sI][2] = Wg[o + 2u * Nn];
    bw[sI][3] = Wg[o + 3u * Nn];
  }
  __builtin_amdgcn_sched_barrier(0);      // (pins the loads here: the scheduler sinks a prefetch to its first use)
  RSX_STAMP(32, blockIdx.x == 0);
  // ---- rows of the tile's examples: lane = (pair slot j, float4 quarter q), fields j, j + 16, ... ----
  const int q = lane % LPR, j = lane / LPR;
  const f32x4* __restrict__ TV = reinterpret_cast<const f32x4*>(g.tables);
  // (F <= 64 = 4 * PPP: ONE pass, straight-line -- the lane's fields are j + 16 k, k = 0..3, clamped to F - 1)
  const int32_t* __restrict__ idg = g.ids;
  const int32_t* __restrict__ rog = g.row_off;
  {
    int row[4][4];
    uint32_t fc[4];
    int ro[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int f = j + k * PPP;
      fc[k] = (uint32_t)(f < F ? f : F - 1);
      ro[k] = rog[fc[k]];
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int b = by * TM + 4 * w + e;
      const uint32_t ib = (uint32_t)(b < p.B ? b : p.B - 1) * (uint32_t)F;     // (B * F < 2^31: the sort's own bound)
#pragma unroll
      for (int k = 0; k < 4; ++k) row[e][k] = ro[k] + idg[ib + fc[k]];
    }
    RSX_STAMP(33, blockIdx.x == 0);
    f32x4 ev[4][4];                       // (the native vector type: an array of HIP's float4 STRUCT stayed in scratch memory here)
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int k = 0; k < 4; ++k) ev[e][k] = TV[(size_t)row[e][k] * LPR + q];
    __builtin_amdgcn_sched_barrier(0);    // (all 16 row loads in flight before the first store)
    RSX_STAMP(34, blockIdx.x == 0);
    // (unconditional stores on the clamped field index -- a field past F rewrites field F - 1's row with the same bytes: behind
    // a guard the compiler sinks every row load to its store and the 16 round trips run one after the other)
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int k = 0; k < 4; ++k)
        *reinterpret_cast<f32x4*>(et + (4 * w + e) * ES + (int)fc[k] * D + q * 4) = ev[e][k];
  }
  RSX_STAMP(35, blockIdx.x == 0);
  __syncthreads();
  RSX_STAMP(5, blockIdx.x == 0);
  // ---- the tile: acc0 takes this wave's even visits, acc1 the odd ones, each in ascending k (tile_ksplit's order); the two
  // chains are issued interleaved (fp32 MFMA 16x16x4 has a ~32-cycle dependent latency) ----
  // (rows past the batch hold a copy of the last example: an A row only reaches its own output row, which is never stored)
  const float* arow = et + i * ES + 4 * kq;
  f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int sI = 0; sI < NS; sI += 2) {
    const int ks0 = w + 4 * sI, ks1 = ks0 + 4;
    if (ks0 < F) {                                          // (wave-uniform)
      const float4 x0 = *reinterpret_cast<const float4*>(arow + ks0 * 16);
      const float4 x1 = *reinterpret_cast<const float4*>(arow + (ks1 < F ? ks1 : ks0) * 16);
      const float a0[4] = {x0.x, x0.y, x0.z, x0.w};
      const float a1v[4] = {x1.x, x1.y, x1.z, x1.w};
      if (sI + 1 < NS && ks1 < F) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          acc0 = mfma16(a0[t], bw[sI][t], acc0);
          acc1 = mfma16(a1v[t], bw[sI + 1 < NS ? sI + 1 : sI][t], acc1);
        }
      } else {
#pragma unroll
        for (int t = 0; t < 4; ++t) acc0 = mfma16(a0[t], bw[sI][t], acc0);
      }
    }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) part[w * 256 + ((lane >> 4) * 4 + r) * 16 + (lane & 15)] = acc0[r] + acc1[r];
  RSX_STAMP(6, blockIdx.x == 0);
  __syncthreads();
  const float v = ((part[tid] + part[256 + tid]) + part[512 + tid]) + part[768 + tid];
  fwd_tile_epilogue(p, v, bx, by, cred);
  RSX_STAMP(7, blockIdx.x == 0);
}

// ---------------------------------------------------------------------------------------------
// head (DeepFM family):  o = dropout(BN(a_last));  u = o . wd + bd;  t2 = act2(u)
//   z = wo[0]*act0(s0 + c0) + wo[1]*s1 + wo[2]*t2 + bo ;  loss = mean sigmoid-CE(z, y)
// and its backward in the same pass: dz, d s0, d s1, d(BN output of the last layer) + partials.
// act0/act2 = relu when the flag is set.  wo == null means z = s0 + u + s1 (sum head).
// grid = RT (16 rows per workgroup), block = 256 (4 waves x 4 rows).  N <= 256.
// ---------------------------------------------------------------------------------------------
struct HeadArgs {
  const float* a_last;       // [B, N]
  const double* fstat_last;  // [RT, 2, N]
  const float* gamma;
  const float* beta;
  const float* mask;         // [B, N] or null
  float* bn_out;             // [2, N]
  const float* wd;           // [N]   (dnn.Wout [N,1])
  const float* bd;           // [1]
  const float* s0;           // [B] first extra scalar input (y1 pre-activation) or null
  const float* c0;           // [1] bias added to s0 (b1) or null
  const float* s1;           // [B] second extra scalar input (y2) or null
  const float* wo;           // [3] out.W or null
  const float* bo;           // [1] out.b or null
  const float* labels;       // [B]
  float* prob;               // [B] sigmoid(z)
  float* dy_last;            // [B, N] grad wrt BN output of the last layer (after dropout backward)
  double* bstat_last;        // [RT, 2, N] partial (sum dy, sum dy*xhat)
  float* dwd_part;           // [RT, N] partial dWd
  double* hpart;             // [RT, 8] partial: loss, dwo0, dwo1, dwo2, dbo, dc0, dbd, -
  float* gs0;                // [B] d loss / d s0
  float* gs1;                // [B] d loss / d s1
  const uint32_t* rng_step;
  uint32_t seed, layer;
  float rate, loss_scale;    // loss_scale = 1/(B * replicas)
  int relu0, relu2;
  int B, N, RT;
  double inv_B;              // 1.0 / B
  int n_own;
  AdamSlice sweep;
};

// CPL: columns per lane = ceil(N / 16) rounded up to 4 / 8 / 16 (a compile-time bound keeps every load unconditional)
// RID: 0 = no riders; 1 = a slice of the one-step untouched-row sweep; 2 = a slice of a full optimizer WINDOW's sweep (round 4:
// the 16 head workgroups leave 240 CUs idle for ~7 us -- 1/8 of the window's sweep per step fits there, and the stand-alone
// 70 us launch per window disappears from the step's chain)
template <int CPL, int RID>
__global__ __launch_bounds__(256) void tower_head_k(const HeadArgs p) {
  if (RID != 0 && (int)blockIdx.x >= p.n_own) {
    const uint32_t blk = p.sweep.blk_lo + (blockIdx.x - p.n_own);
    if (RID == 2) {
      if (blk == 0 && threadIdx.x == 0 && !p.sweep.args.alpha_src) adam_publish_step_sizes<ADAM_WMAX>(p.sweep.args);
      RSX_RIDE_WINDOW(p.sweep.args, blk);
    } else {
      RSX_RIDE_SWEEP(p.sweep.args, blk);
    }
    return;
  }
  __shared__ float sc[256], sh[256], mu[256], rs[256];
  // per (row of the tile, column): dy, xhat and o * g2 -- the column partials are summed from here by one thread per column
  // (round 4; they used to travel through 80 cross-lane fp64 / fp32 shuffles per wave, a third of the kernel's instructions).
  // Row stride = 16 mod 64 floats: the 4 rows a wave writes at once land in 4 different bank groups.
  constexpr int LDC = 16 * CPL + 16;
  __shared__ float ldy[16][LDC], lxh[16][LDC], log2_[16][LDC];
  __shared__ double hrow[16][8];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  RSX_STAMP(8, blockIdx.x == 0);
  // (the row's loads are issued BEFORE the statistics reduction below: they do not depend on it)
  // One row per 16-lane group: the wave's 4 rows go through the dot product, the loss and its backward SIDE BY SIDE (the
  // per-row chain -- 16-lane reduction, exp / log1p / divide -- used to run 4 times in sequence per wave: 1.6 us per row,
  // measured with the phase stamps).  Lane (g, li) holds columns c = li + 16 k of row 16*blockIdx.x + 4*w + g.
  const int g = lane >> 4, li = lane & 15;
  const int row = blockIdx.x * TM + w * 4 + g;
  const bool rok = row < p.B;
  const size_t rowc = (size_t)(rok ? row : p.B - 1);
  float av[CPL], wdv[CPL];
#pragma unroll
  for (int k = 0; k < CPL; ++k) {     // unconditional on clamped indices (rows past the batch / columns past N are masked below)
    const int c = li + 16 * k;
    const int cc = c < p.N ? c : p.N - 1;
    av[k] = p.a_last[rowc * p.N + cc];
    wdv[k] = p.wd[cc] * (c < p.N ? 1.f : 0.f);
  }
  const float s0v = p.s0 ? p.s0[rowc] : 0.f, s1v = p.s1 ? p.s1[rowc] : 0.f, y = p.labels[rowc];
  const float bd = p.bd[0];
  const float wo0 = p.wo ? p.wo[0] : 1.f, wo1 = p.wo ? p.wo[1] : 1.f, wo2 = p.wo ? p.wo[2] : 1.f;
  const float bo = p.bo ? p.bo[0] : 0.f;
  const float c0 = p.c0 ? p.c0[0] : 0.f;
  const DropRng dr = drop_make(p.rate, p.mask, p.rng_step, p.seed, p.layer);
  for (int c = tid; c < p.N; c += 256) {
    if (p.gamma == nullptr) {   // no batch-norm on the last layer: dy_last is the gradient wrt the dropout input
      sc[c] = 1.f; sh[c] = 0.f; mu[c] = 0.f; rs[c] = 0.f;
      continue;
    }
    float mean, rstd;
    bn_col_stats(p.fstat_last, p.RT, p.N, c, p.inv_B, mean, rstd);
    const float inv = rstd * p.gamma[c];
    sc[c] = inv;
    sh[c] = p.beta[c] - mean * inv;
    mu[c] = mean;
    rs[c] = rstd;
    if (blockIdx.x == 0) {
      p.bn_out[c] = mean;
      p.bn_out[p.N + c] = rstd;
    }
  }
  __syncthreads();
  RSX_STAMP(9, blockIdx.x == 0);
  float o[CPL], xh[CPL], mk[CPL];
  float dot = 0.f;
#pragma unroll
  for (int k = 0; k < CPL; ++k) {
    const int c = li + 16 * k;
    o[k] = 0.f; xh[k] = 0.f; mk[k] = 0.f;
    if (c < p.N) {
      const float a = av[k];
      mk[k] = drop_mul(dr, p.mask, rowc * p.N + c);
      const float v = (a * sc[c] + sh[c]) * mk[k];
      o[k] = v;
      xh[k] = (a - mu[c]) * rs[c];
      dot += v * wdv[k];
    }
  }
#pragma unroll
  for (int m = 1; m < 16; m <<= 1) dot += __shfl_xor(dot, m);      // over the row's 16 lanes (fixed tree)
  RSX_STAMP(12, blockIdx.x == 0);
  const float u = dot + bd;
  const float t2 = (p.relu2 && u <= 0.f) ? 0.f : u;
  const float v0 = p.s0 ? s0v + c0 : 0.f;
  const float t0 = (p.relu0 && v0 <= 0.f) ? 0.f : v0;
  const float zz = wo0 * t0 + wo1 * s1v + wo2 * t2 + bo;
  const float pr = 1.f / (1.f + expf(-zz));
  const float ce = fmaxf(zz, 0.f) - zz * y + log1pf(expf(-fabsf(zz)));
  const float dz = (pr - y) * p.loss_scale;
  const float g0 = (p.relu0 && v0 <= 0.f) ? 0.f : dz * wo0;   // d/d(s0 + c0)
  const float g2 = (p.relu2 && u <= 0.f) ? 0.f : dz * wo2;    // d/du
  if (li == 0) {
    double* hr = hrow[w * 4 + g];
    if (rok) {
      p.prob[row] = pr;
      if (p.gs0) p.gs0[row] = g0;
      if (p.gs1) p.gs1[row] = dz * wo1;
      hr[0] = (double)ce; hr[1] = (double)(dz * t0); hr[2] = (double)(dz * s1v); hr[3] = (double)(dz * t2);
      hr[4] = (double)dz; hr[5] = (double)g0; hr[6] = (double)g2; hr[7] = 0.0;
    } else {
#pragma unroll
      for (int k = 0; k < 8; ++k) hr[k] = 0.0;
    }
  }
  RSX_STAMP(13, blockIdx.x == 0);
  // column partials over the tile's 16 rows, in the order the shuffles used to produce: per wave (r0 + r1) + (r2 + r3), then the
  // waves in order -- the same bits
  {
    const int r = w * 4 + g;
#pragma unroll
    for (int k = 0; k < CPL; ++k) {
      const int c = li + 16 * k;
      const bool ok = rok && c < p.N;
      const float dyv = ok ? g2 * wdv[k] * mk[k] : 0.f;          // d/d(BN output) after dropout backward
      if (ok) p.dy_last[(size_t)row * p.N + c] = dyv;
      ldy[r][c] = dyv;
      lxh[r][c] = xh[k];
      log2_[r][c] = ok ? o[k] * g2 : 0.f;
    }
  }
  RSX_STAMP(10, blockIdx.x == 0);
  __syncthreads();
  for (int c = tid; c < p.N; c += 256) {
    double sdy[4], sdx[4];
    float swd[4];
#pragma unroll
    for (int ww = 0; ww < 4; ++ww) {
      const float d0 = ldy[4 * ww][c], d1 = ldy[4 * ww + 1][c], d2 = ldy[4 * ww + 2][c], d3 = ldy[4 * ww + 3][c];
      const float x0 = lxh[4 * ww][c], x1 = lxh[4 * ww + 1][c], x2 = lxh[4 * ww + 2][c], x3 = lxh[4 * ww + 3][c];
      sdy[ww] = ((double)d0 + (double)d1) + ((double)d2 + (double)d3);
      sdx[ww] = ((double)d0 * (double)x0 + (double)d1 * (double)x1) + ((double)d2 * (double)x2 + (double)d3 * (double)x3);
      swd[ww] = (log2_[4 * ww][c] + log2_[4 * ww + 1][c]) + (log2_[4 * ww + 2][c] + log2_[4 * ww + 3][c]);
    }
    stat_put(p.bstat_last, p.RT, (int)blockIdx.x, p.N, c, sdy[0] + sdy[1] + sdy[2] + sdy[3], sdx[0] + sdx[1] + sdx[2] + sdx[3]);
    p.dwd_part[(size_t)blockIdx.x * p.N + c] = swd[0] + swd[1] + swd[2] + swd[3];
  }
  if (tid < 8) {                                   // the 16 rows of the tile in ascending order
    double s = 0.0;
#pragma unroll
    for (int r = 0; r < 16; ++r) s += hrow[r][tid];
    p.hpart[(size_t)blockIdx.x * 8 + tid] = s;
  }
  RSX_STAMP(11, blockIdx.x == 0);
}

// ---------------------------------------------------------------------------------------------
// backward layer l.  da = relu'(a) * BNbwd(dy) is applied on load.  Tile families (blockIdx.x ranges), every
// workgroup = 256 threads = one 16x16 tile with its reduction dimension split over the 4 waves:
//   [0, n_din)              d(input)[B x K] = da . W^T, then dropout backward of layer l-1 and the
//                           partial sums for ITS BN backward (or the final dX for the first layer)
//   [n_din, +n_dw)          dW[K+1 x N] = [in' | 1]^T . da   (tiled over the OUTPUT, loops over all B rows;
//                           the extra ones-row K is db; the kf == 0 tiles also emit dgamma / dbeta)
//   last block (if head)    reduce the head partials: dWd, dbd, dwo, dbo, dc0, loss
// dyn LDS: 5*N + 4 + 1024 + 256 floats.
// ---------------------------------------------------------------------------------------------
struct BwdArgs {
  // this layer
  const float* in;          // [B, K] raw input of the layer (X, or a_{l-1})
  const float* W;           // [K, N]
  const float* a;           // [B, N] relu output (pre-BN)
  const float* dy;          // [B, N] grad wrt BN output
  const double* bstat;      // [RT, 2, N]
  const float* bn;          // [2, N] mean, rstd
  const float* gamma;       // [N]
  float* dW;                // [K, N]
  float* db;                // [N]
  float* dgamma;
  float* dbeta;
  // previous layer (null for the first layer)
  const float* bn_prev;     // [2, K]
  const float* gamma_prev;
  const float* beta_prev;
  const float* mask_prev;   // [B, K] or null
  float* dy_prev;           // [B, K]   (first layer: dX [B, K])
  double* bstat_prev;       // [RT, 2, K]
  // head partial reduce (null unless this is the last layer)
  const double* hpart;      // [RT, 8]
  const float* dwd_part;    // [RT, N]
  float* dwd; float* dbd; float* dwo; float* dbo; float* dc0; float* loss;
  const uint32_t* rng_step;
  uint32_t seed, layer_prev;
  float rate;
  int has_wo;
  int B, K, N, RT, RTh;      // RT: rows of the (possibly pre-reduced) bstat; RTh: row tiles = rows of hpart / dwd_part
  int n_din, n_dw, ct_k, ct_k1, ct_n;
  // dW tiles split their reduction over the batch into `sb` row blocks of `ksb` k-steps (large batches: the serial chain
  // of one workgroup per tile would span the whole batch); tower_reduce_dw_k then adds the sb partial tiles in order
  int sb, ksb;
  float* dwp;                // [tiles, sb, 256] partial tiles
  int din_rtw;               // row tiles per d(input) workgroup (1, or 4 for large batches)
  int dxg;                   // small batches: column tiles per d(input) workgroup (4: grouped LDS-staged tiles; 0: one tile)
  int n_head;                // 1 when the head-partial reduce block is present
  int n_sort;                // extra workgroups that run the per-field dedup sort of the same step (0: none)
  double* zero_p;            // fixed-point statistics rows this launch clears for the NEXT step's producers (stat_zero), or null
  int zero_n;
  int acc_dx;                // first layer: dy_prev (= dX) += instead of = (a rider of an EARLIER launch wrote its share already)
  // rider (round 6, dcn.py): the cross layers' backward as the launch's LAST n_xr workgroups (tower_bwd_big_k<.., XR = true>)
  int n_xr, xr_epw;
  CrossBwdArgs xr;
  SortArgs sort;
  AdamSlice sweep;           // optional slice of the untouched-row optimizer sweep (after the sort workgroups)
};

struct ColBwd { float mean, rstd, k1, sdy, sdx; };

__device__ __forceinline__ ColBwd bwd_col(const BwdArgs& p, int c) {
  double s1, s2;
  col_partials(p.bstat, p.RT, p.N, c, s1, s2);
  ColBwd o;
  if (p.gamma == nullptr) {   // layer without batch-norm: da_of passes dy through the relu mask
    o.mean = 0.f; o.rstd = 0.f; o.k1 = 0.f; o.sdy = 0.f; o.sdx = 0.f;
    return o;
  }
  o.mean = p.bn[c];
  o.rstd = p.bn[p.N + c];
  o.k1 = p.gamma[c] * o.rstd / (float)p.B;
  o.sdy = (float)s1;
  o.sdx = (float)s2;
  return o;
}
__device__ __forceinline__ float da_of(float a, float dy, const ColBwd& c, float Bf, bool nobn = false) {
  if (a <= 0.f) return 0.f;
  if (nobn) return dy;
  const float xh = (a - c.mean) * c.rstd;
  return c.k1 * (Bf * dy - c.sdy - xh * c.sdx);
}

// d(input) tiles for small batches: one workgroup = one 16-row tile x FOUR 16-column tiles (wave w owns column tile
// 4*grp + w with the full reduction over the N outputs).  The operands are staged through LDS with coalesced float4 loads
// -- da = relu'(a) * BNbwd(dy) is formed ONCE per workgroup instead of once per column tile, W rows are read as rows --
// and every load of the tile is issued before the column constants are waited for, so the tile costs one memory round
// trip.  (The one-tile-per-workgroup form read its operands as 4-byte strided loads, 24 instructions touching up to 64
// lines each, and 624 workgroups of the first layer each re-reduced the BN partials of all N columns: phase stamps
// showed 2.0 us of constants + 2.8 us of k-loop per tile.)  LDS: 5*N + (16 + 64) * (NP + 4) floats, NP = N rounded to 16.
__device__ __forceinline__ void bwd_dx_group_tile(const BwdArgs& p, const int bid, float* lds, const bool first,
                                                  const bool nobn) {
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int i = lane & 15, kq = lane >> 4;
  const int NP = (p.N + 15) & ~15, LD = NP + 4, N4 = p.N >> 2;
  float* Lm = lds; float* Lr = lds + p.N; float* Lk = lds + 2 * p.N; float* Ls = lds + 3 * p.N; float* Lx = lds + 4 * p.N;
  float* sA = lds + ((5 * p.N + 3) & ~3);          // [16][LD]  da tile
  float* sW = sA + 16 * LD;                         // [64][LD]  W rows of the workgroup's 64 input features
  const int ngrp = (p.ct_k + p.dxg - 1) / p.dxg;
  const int grp = bid % ngrp, rt = bid / ngrp;
  const float Bf = (float)p.B;
  // (1) tile loads first: a, dy rows (kept in registers until the constants are there) ...
  float4 av[2], dv[2];                              // 16 * N/4 float4 over 256 threads: <= 2 each (N <= 128, host-checked)
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int e = tid + 256 * u;
    const int r = e / N4, c4 = e - r * N4;
    const bool ok = e < 16 * N4;
    const size_t row = (size_t)((ok && rt * TM + r < p.B) ? rt * TM + r : p.B - 1);
    av[u] = reinterpret_cast<const float4*>(p.a + row * p.N)[ok ? c4 : 0];
    dv[u] = reinterpret_cast<const float4*>(p.dy + row * p.N)[ok ? c4 : 0];
  }
  // ... the epilogue's operands (previous layer's activation / BN statistics of this lane's 4 output elements) ...
  const int ocol = (grp * p.dxg + w) * 16 + i;
  const int occ = ocol < p.K ? ocol : p.K - 1;
  float xin[4] = {0.f, 0.f, 0.f, 0.f}, bnm = 0.f, bnr = 0.f;
  if (!first) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int orow = rt * TM + 4 * kq + r;
      xin[r] = p.in[(size_t)(orow < p.B ? orow : p.B - 1) * p.K + occ];
    }
    bnm = p.bn_prev[occ];
    bnr = p.bn_prev[p.K + occ];
  }
  // ... and the W rows into LDS (rows past K and the k-padding columns n >= N are zero): 4 loads per thread in flight
  // before the first LDS store (a load -> store loop is one memory round trip per iteration: 7 of them measured 5 us)
  {
    const int LD4 = LD >> 2, tot = 64 * LD4;
    auto wsrc = [&](int e) -> const float4* {
      const int ec = e < tot ? e : tot - 1;
      const int r = ec / LD4, c4 = ec - r * LD4;
      const int kcol = grp * p.dxg * 16 + r;
      return reinterpret_cast<const float4*>(p.W + (size_t)(kcol < p.K ? kcol : p.K - 1) * p.N) + (c4 < N4 ? c4 : 0);
    };
    auto wdst = [&](int e, const float4 v) {
      if (e < tot) {
        const int r = e / LD4, c4 = e - r * LD4;
        const bool ok = grp * p.dxg * 16 + r < p.K && c4 < N4;
        reinterpret_cast<float4*>(sW + r * LD)[c4] = ok ? v : F4Z;
      }
    };
    for (int e0 = tid; e0 < tot; e0 += 256 * 4) {
      const float4 t0 = *wsrc(e0), t1 = *wsrc(e0 + 256), t2 = *wsrc(e0 + 512), t3 = *wsrc(e0 + 768);
      wdst(e0, t0);
      wdst(e0 + 256, t1);
      wdst(e0 + 512, t2);
      wdst(e0 + 768, t3);
    }
  }
  // (2) column constants of the N outputs
  for (int c = tid; c < p.N; c += 256) {
    const ColBwd cb = bwd_col(p, c);
    Lm[c] = cb.mean; Lr[c] = cb.rstd; Lk[c] = cb.k1; Ls[c] = cb.sdy; Lx[c] = cb.sdx;
  }
  for (int e = tid; e < 16 * ((LD - p.N) >> 2) ; e += 256) {      // zero the k padding of the da tile
    const int r = e / ((LD - p.N) >> 2), c4 = e - r * ((LD - p.N) >> 2);
    reinterpret_cast<float4*>(sA + r * LD + p.N)[c4] = F4Z;
  }
  __syncthreads();
  RSX_STAMP(first ? 17 : 25, bid == 0);
  // (3) da tile -> LDS
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int e = tid + 256 * u;
    if (e < 16 * N4) {
      const int r = e / N4, c = 4 * (e - r * N4);
      const float rokf = rt * TM + r < p.B ? 1.f : 0.f;
      ColBwd c0, c1, c2, c3;
      c0.mean = Lm[c]; c0.rstd = Lr[c]; c0.k1 = Lk[c]; c0.sdy = Ls[c]; c0.sdx = Lx[c];
      c1.mean = Lm[c + 1]; c1.rstd = Lr[c + 1]; c1.k1 = Lk[c + 1]; c1.sdy = Ls[c + 1]; c1.sdx = Lx[c + 1];
      c2.mean = Lm[c + 2]; c2.rstd = Lr[c + 2]; c2.k1 = Lk[c + 2]; c2.sdy = Ls[c + 2]; c2.sdx = Lx[c + 2];
      c3.mean = Lm[c + 3]; c3.rstd = Lr[c + 3]; c3.k1 = Lk[c + 3]; c3.sdy = Ls[c + 3]; c3.sdx = Lx[c + 3];
      float4 o;
      o.x = da_of(av[u].x, dv[u].x, c0, Bf, nobn) * rokf;
      o.y = da_of(av[u].y, dv[u].y, c1, Bf, nobn) * rokf;
      o.z = da_of(av[u].z, dv[u].z, c2, Bf, nobn) * rokf;
      o.w = da_of(av[u].w, dv[u].w, c3, Bf, nobn) * rokf;
      *reinterpret_cast<float4*>(sA + r * LD + c) = o;
    }
  }
  __syncthreads();
  // (4) wave w: 16 x 16 tile, k-steps of 16 outputs; two accumulators cover the MFMA's dependent latency
  f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
  const float* ar = sA + i * LD + 4 * kq;
  const float* br = sW + (w * 16 + i) * LD + 4 * kq;
  const int nks = NP >> 4;
  for (int ks = 0; ks < nks; ks += 2) {
    const float4 a0 = *reinterpret_cast<const float4*>(ar + 16 * ks);
    const float4 b0 = *reinterpret_cast<const float4*>(br + 16 * ks);
    acc0 = mfma16(a0.x, b0.x, acc0);
    acc0 = mfma16(a0.y, b0.y, acc0);
    acc0 = mfma16(a0.z, b0.z, acc0);
    acc0 = mfma16(a0.w, b0.w, acc0);
    if (ks + 1 < nks) {
      const float4 a1 = *reinterpret_cast<const float4*>(ar + 16 * (ks + 1));
      const float4 b1 = *reinterpret_cast<const float4*>(br + 16 * (ks + 1));
      acc1 = mfma16(a1.x, b1.x, acc1);
      acc1 = mfma16(a1.y, b1.y, acc1);
      acc1 = mfma16(a1.z, b1.z, acc1);
      acc1 = mfma16(a1.w, b1.w, acc1);
    }
  }
  RSX_STAMP(first ? 18 : 26, bid == 0);
  // (5) epilogue: lane (i, kq) holds rows 4 kq + r, column ocol.  Dropout backward of the previous layer and the
  // partial sums of ITS batch-norm backward over this tile's 16 rows (4 in-lane, then the 4 lane quarters in order)
  const bool wok = w < p.dxg && ocol < p.K;
  double s1 = 0.0, s2 = 0.0;
  const DropRng dr = drop_make(first ? 0.f : p.rate, p.mask_prev, p.rng_step, p.seed, p.layer_prev);
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int orow = rt * TM + 4 * kq + r;
    float o = acc0[r] + acc1[r];
    if (orow < p.B && wok) {
      if (!first) {
        o *= drop_mul(dr, p.mask_prev, (size_t)orow * p.K + ocol);
        const float xh = (xin[r] - bnm) * bnr;
        s1 += (double)o;
        s2 += (double)o * (double)xh;
      }
      p.dy_prev[(size_t)orow * p.K + ocol] = o;
    }
  }
  if (!first) {
    s1 += __shfl_xor(s1, 16); s2 += __shfl_xor(s2, 16);
    s1 += __shfl_xor(s1, 32); s2 += __shfl_xor(s2, 32);
    if (kq == 0 && wok) stat_put(p.bstat_prev, p.RT, rt, p.K, ocol, s1, s2);
  }
  RSX_STAMP(first ? 19 : 27, bid == 0);
}

// the last layer's backward launch also folds the head's per-row-tile partials (one workgroup): dWd, dbd, dwo, dbo, dc0, loss
__device__ __forceinline__ void tower_head_reduce(const BwdArgs& p, float* part /* LDS [256] */, double* cred /* LDS [32] */) {
  const int tid = threadIdx.x, lane = tid & 63;
  if (p.hpart != nullptr) {
    // RTh row-tile partials per column: the 4 waves take contiguous quarter ranges of the rows (8 loads in flight),
    // lane = column; the quarters are then added in ascending order -- fixed association, short dependent chains
    const int w = tid >> 6;
    const int per = (p.RTh + 3) / 4;
    const int r0 = w * per, r1 = r0 + per < p.RTh ? r0 + per : p.RTh;
    for (int c0 = 0; c0 < p.N; c0 += 64) {
      const int c = c0 + lane;
      float s = 0.f;
      if (c < p.N) {
        int r = r0;
        for (; r + 8 <= r1; r += 8) {
          float t[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) t[u] = p.dwd_part[(size_t)(r + u) * p.N + c];
#pragma unroll
          for (int u = 0; u < 8; ++u) s += t[u];
        }
        for (; r < r1; ++r) s += p.dwd_part[(size_t)r * p.N + c];
      }
      __syncthreads();
      part[w * 64 + lane] = s;
      __syncthreads();
      if (w == 0 && c < p.N) p.dwd[c] = ((part[lane] + part[64 + lane]) + part[128 + lane]) + part[192 + lane];
    }
    double hs = 0.0;
    if (lane < 8) {
      int r = r0;
      for (; r + 8 <= r1; r += 8) {
        double t[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) t[u] = p.hpart[(size_t)(r + u) * 8 + lane];
#pragma unroll
        for (int u = 0; u < 8; ++u) hs += t[u];
      }
      for (; r < r1; ++r) hs += p.hpart[(size_t)r * 8 + lane];
      cred[w * 8 + lane] = hs;
    }
    __syncthreads();
    if (tid < 8) {
      const double s = ((cred[tid] + cred[8 + tid]) + cred[16 + tid]) + cred[24 + tid];
      if (tid == 0) p.loss[0] = (float)(s / (double)p.B);
      if (p.has_wo && tid >= 1 && tid <= 3) p.dwo[tid - 1] = (float)s;
      if (p.has_wo && tid == 4) p.dbo[0] = (float)s;
      if (p.dc0 != nullptr && tid == 5) p.dc0[0] = (float)s;
      if (tid == 6) p.dbd[0] = (float)s;
    }
  }
}

// SPLIT: the dW tiles' batch reduction is cut into p.sb row blocks (large batches); false keeps the single-block code
// path free of the block arithmetic
template <bool SPLIT, bool RID>
__global__ __launch_bounds__(256) void tower_bwd_k(const BwdArgs p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  if (p.zero_n > 0) stat_zero(p.zero_p, p.zero_n);
  float* Lm = lds; float* Lr = lds + p.N; float* Lk = lds + 2 * p.N; float* Ls = lds + 3 * p.N; float* Lx = lds + 4 * p.N;
  float* part = lds + 5 * p.N + ((4 - (5 * p.N) % 4) % 4);         // [4][256], 16-byte aligned
  double* cred = reinterpret_cast<double*>(part + 1024);              // [4][2][16]
  const int tid = threadIdx.x, lane = tid & 63;
  // dispatch order: the dW tiles first -- their reduction over the batch is the longest dependent chain of the launch,
  // so it must start at t = 0 and let the (many, short) d(input) tiles fill in around it
  int bid = blockIdx.x;
  if (SPLIT) bid = bid < p.n_dw ? bid + p.n_din : (bid < p.n_dw + p.n_din ? bid - p.n_dw : bid);
  const float Bf = (float)p.B;
  const bool first = p.bn_prev == nullptr;
  const bool nobn = p.gamma == nullptr;          // this layer has no batch-norm (uniform)
  const int i = lane & 15, kq = lane >> 4;
  const int sb0 = first ? 16 : 24;
  if (p.dxg > 0 && bid < p.n_din) {
    RSX_STAMP(sb0 + 0, bid == 0);
    bwd_dx_group_tile(p, bid, lds, first, nobn);
    return;
  }
  if (bid < p.n_din) {
    // ---- d(input) tile: rows rt*16.., input columns kc*16.. ; reduction over the N outputs ----------
    RSX_STAMP(sb0 + 0, bid == 0);
    // (the epilogue's dropout key: its step-counter load starts here, not behind the k-loop)
    const DropRng dr_e = drop_make(first ? 0.f : p.rate, p.mask_prev, p.rng_step, p.seed, p.layer_prev);
    for (int c = tid; c < p.N; c += 256) {
      const ColBwd cb = bwd_col(p, c);
      Lm[c] = cb.mean; Lr[c] = cb.rstd; Lk[c] = cb.k1; Ls[c] = cb.sdy; Lx[c] = cb.sdx;
    }
    __syncthreads();
    RSX_STAMP(sb0 + 1, bid == 0);
    // large batches (SPLIT): one workgroup walks din_rtw consecutive row tiles of its column tile, so the column
    // constants above (and the launch's workgroup count) are amortised; small batches: one tile per workgroup
    const int rtw = SPLIT ? p.din_rtw : 1;
    const int kc = bid % p.ct_k, rt0 = (bid / p.ct_k) * rtw;
    const int kcol = kc * 16 + i;   // B-operand "column" = input feature
    const bool cok = kcol < p.K;
    for (int rr = 0; rr < rtw; ++rr) {
    const int rt = rt0 + rr;
    if (rt >= p.RTh) break;         // workgroup-uniform
    if (rr) __syncthreads();        // `part` / `cred` of the previous tile have been consumed
    const int row = rt * TM + i;
    const bool rok = row < p.B;
    // the epilogue's operands (previous layer's activation and BN statistics of this thread's output element) do not depend
    // on the tile: issue them now, unconditionally on clamped indices, so they arrive during the MFMA loop
    const int orow = rt * TM + (tid >> 4), ocol = kc * 16 + (tid & 15);
    float xin_e = 0.f, bnm_e = 0.f, bnr_e = 0.f;
    if (!first) {   // workgroup-uniform
      const int oc = ocol < p.K ? ocol : p.K - 1;
      xin_e = p.in[(size_t)(orow < p.B ? orow : p.B - 1) * p.K + oc];
      bnm_e = p.bn_prev[oc];
      bnr_e = p.bn_prev[p.K + oc];
    }
    const size_t arow = (size_t)(rok ? row : p.B - 1) * p.N, wrow = (size_t)(cok ? kcol : 0) * p.N;
    const bool n4 = (p.N & 3) == 0;                    // (uniform) rows of a / dy / W are 16-byte aligned: one float4 per operand
    const float v = tile_ksplit((p.N + 15) / 16, part, [&](int ks, float* a, float* b) {
      const int nn = ks * 16 + 4 * kq;
      float ar[4], dyr[4], wr[4];
      if (n4) {
        // the lane's 4 consecutive outputs as ONE 16-byte load per operand (3 instead of 12 load instructions per k-step, each
        // touching the same 16 lines; a k-step past N reads the last 4 columns and is masked below like a clamped scalar)
        const int nb = nn < p.N ? nn : p.N - 4;
        const float4 a4 = *reinterpret_cast<const float4*>(p.a + arow + nb);
        const float4 d4 = *reinterpret_cast<const float4*>(p.dy + arow + nb);
        const float4 w4 = *reinterpret_cast<const float4*>(p.W + wrow + nb);
        ar[0] = a4.x; ar[1] = a4.y; ar[2] = a4.z; ar[3] = a4.w;
        dyr[0] = d4.x; dyr[1] = d4.y; dyr[2] = d4.z; dyr[3] = d4.w;
        wr[0] = w4.x; wr[1] = w4.y; wr[2] = w4.z; wr[3] = w4.w;
      } else {
#pragma unroll
      for (int t = 0; t < 4; ++t) {                    // unconditional loads on clamped indices (see the dW tiles)
        const int nc = nn + t < p.N ? nn + t : p.N - 1;
        ar[t] = p.a[arow + nc];
        dyr[t] = p.dy[arow + nc];
        wr[t] = p.W[wrow + nc];
      }
      }
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int n = nn + t;
        const int nc = n < p.N ? n : p.N - 1;
        const float nf = n < p.N ? 1.f : 0.f;
        ColBwd cb; cb.mean = Lm[nc]; cb.rstd = Lr[nc]; cb.k1 = Lk[nc]; cb.sdy = Ls[nc]; cb.sdx = Lx[nc];
        a[t] = da_of(ar[t], dyr[t], cb, Bf, nobn) * (rok ? nf : 0.f);
        b[t] = wr[t] * (cok ? nf : 0.f);
      }
    });
    RSX_STAMP(sb0 + 2, bid == 0);
    double s1 = 0.0, s2 = 0.0;
    if (orow < p.B && ocol < p.K) {
      float o = v;
      if (!first) {
        o *= drop_mul(dr_e, p.mask_prev, (size_t)orow * p.K + ocol);
        const float xh = (xin_e - bnm_e) * bnr_e;
        s1 = (double)o;
        s2 = (double)o * (double)xh;
      }
      p.dy_prev[(size_t)orow * p.K + ocol] = o;
    }
    if (!first) {
      s1 += __shfl_xor(s1, 16); s2 += __shfl_xor(s2, 16);
      s1 += __shfl_xor(s1, 32); s2 += __shfl_xor(s2, 32);
      if (lane < 16) {
        cred[((tid >> 6) * 2 + 0) * 16 + lane] = s1;
        cred[((tid >> 6) * 2 + 1) * 16 + lane] = s2;
      }
      __syncthreads();
      if (tid < 16 && ocol < p.K)
        stat_put(p.bstat_prev, p.RT, rt, p.K, ocol, ((cred[0 * 16 + tid] + cred[2 * 16 + tid]) + cred[4 * 16 + tid]) + cred[6 * 16 + tid],
                 ((cred[1 * 16 + tid] + cred[3 * 16 + tid]) + cred[5 * 16 + tid]) + cred[7 * 16 + tid]);
    }
    }
    RSX_STAMP(sb0 + 3, bid == 0);
    return;
  }
  if (bid < p.n_din + p.n_dw) {
    // ---- dW tile: rows = input features kf*16.. (feature K = the ones-row -> db), cols = n*16.. ------
    const int t_id = bid - p.n_din;
    const int sbi = SPLIT ? t_id % p.sb : 0, tile = SPLIT ? t_id / p.sb : t_id;
    const int nt = tile % p.ct_n, kf = tile / p.ct_n;
    const int ks0 = SPLIT ? sbi * p.ksb : 0;
    const int ks_all = (p.B + 15) / 16;
    const int nks = SPLIT ? (ks0 + p.ksb < ks_all ? p.ksb : ks_all - ks0) : ks_all;
    RSX_STAMP(sb0 + 4, bid == p.n_din);
    const int feat = kf * 16 + i;          // A-operand row (input feature)
    const int ncol = nt * 16 + i;          // B-operand column
    const bool fok = feat < p.K, ones = feat == p.K, nok = ncol < p.N;
    // column constants of the tile's 16 columns: reduced from the row-tile partials by 16 threads and shared through LDS
    // (all 256 threads used to repeat the 32-load reduction; with 280 such workgroups starting together that prologue
    // measured 5 us under the L2 queueing it caused itself)
    float* Lc = reinterpret_cast<float*>(cred);        // [5][16] (cred is not used by this tile family)
    // (the dropout key's step counter and the previous layer's batch-norm constants are requested BEFORE the column constants:
    // behind the barrier below they were a second memory round trip on the launch's longest chain -- phase stamps: 3.7 us from
    // the tile's entry to its k-loop against 1.8 us in the d(input) tiles)
    const DropRng dr = drop_make(first ? 0.f : p.rate, p.mask_prev, p.rng_step, p.seed, p.layer_prev);
    float fsc = 1.f, fsh = 0.f;
    if (!first && fok && p.gamma_prev != nullptr) {   // (previous layer without batch-norm: identity)
      const float inv = p.bn_prev[p.K + feat] * p.gamma_prev[feat];
      fsc = inv;
      fsh = p.beta_prev[feat] - p.bn_prev[feat] * inv;
    }
    if (tid < 16) {
      const int c = nt * 16 + tid;
      const ColBwd t = bwd_col(p, c < p.N ? c : 0);
      Lc[tid] = t.mean; Lc[16 + tid] = t.rstd; Lc[32 + tid] = t.k1; Lc[48 + tid] = t.sdy; Lc[64 + tid] = t.sdx;
    }
    __syncthreads();
    ColBwd cb;
    cb.mean = Lc[i]; cb.rstd = Lc[16 + i]; cb.k1 = Lc[32 + i]; cb.sdy = Lc[48 + i]; cb.sdx = Lc[64 + i];
    if (kf == 0 && sbi == 0 && nok && tid < 16 && !nobn) {   // lanes 0..15 of wave 0 hold the column constants
      p.dgamma[ncol] = cb.sdx;
      p.dbeta[ncol] = cb.sdy;
    }
    const int featc = fok ? feat : 0, ncolc = nok ? ncol : 0;
    RSX_STAMP(sb0 + 5, bid == p.n_din);
    float v = tile_ksplit(nks, part, [&](int ks, float* a, float* b) {
      // operand loads unconditional on clamped indices, masked by multiplication afterwards (a guarded load is compiled
      // into a branch of its own, and the four k-steps' loads then complete one after the other)
      float xr[4], ar[4], dyr[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int bb = (ks0 + ks) * 16 + 4 * kq + t;
        const size_t bc = (size_t)(bb < p.B ? bb : p.B - 1);
        xr[t] = p.in[bc * p.K + featc];
        ar[t] = p.a[bc * p.N + ncolc];
        dyr[t] = p.dy[bc * p.N + ncolc];
      }
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int bb = (ks0 + ks) * 16 + 4 * kq + t;
        const float vf = bb < p.B ? 1.f : 0.f;
        float x = xr[t];
        if (!first) x = (x * fsc + fsh) * drop_mul(dr, p.mask_prev, (size_t)(bb < p.B ? bb : p.B - 1) * p.K + featc);
        x = fok ? x : (ones ? 1.f : 0.f);
        a[t] = x * vf;
        b[t] = da_of(ar[t], dyr[t], cb, Bf, nobn) * (nok ? vf : 0.f);
      }
    });
    RSX_STAMP(sb0 + 6, bid == p.n_din);
    if (SPLIT) {      // partial tile; tower_reduce_dw_k adds the sb partials in ascending block order
      p.dwp[((size_t)tile * p.sb + sbi) * 256 + tid] = v;
      return;
    }
    const int orow = kf * 16 + (tid >> 4), ocol = nt * 16 + (tid & 15);
    if (ocol < p.N) {
      if (orow < p.K) p.dW[(size_t)orow * p.N + ocol] = v;
      else if (orow == p.K) p.db[ocol] = v;
    }
    RSX_STAMP(sb0 + 7, bid == p.n_din);
    return;
  }
  if (RID && bid >= p.n_din + p.n_dw + p.n_head + p.n_sort) {
    RSX_RIDE_SWEEP(p.sweep.args, p.sweep.blk_lo + (bid - (p.n_din + p.n_dw + p.n_head + p.n_sort)));
    return;
  }
  if (RID && bid >= p.n_din + p.n_dw + p.n_head) {
    // ---- piggy-backed dedup sort: independent of the tower, first needed by the segment-sum -----------
    RSX_RIDE_SORT(p.sort, bid - (p.n_din + p.n_dw + p.n_head), reinterpret_cast<uint32_t*>(lds));
    return;
  }
  // ---- head partial reduce (last layer only) ---------------------------------------------------------
  tower_head_reduce(p, part, cred);
}

// Large batches: dW[tile] = sum of the tile's sb partials in ascending row-block order.  grid = tiles, block = 256.
__global__ __launch_bounds__(256) void tower_reduce_dw_k(const float* __restrict__ dwp, float* __restrict__ dW,
                                                         float* __restrict__ db, int sb, int ct_n, int K, int N) {
  const int tile = blockIdx.x, tid = threadIdx.x;
  const float* all = dwp + (size_t)tile * sb * 256;
  float s = 0.f;
  for (int q = 0; q < sb; q += 8) {
    float t[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) t[u] = q + u < sb ? all[(size_t)(q + u) * 256 + tid] : 0.f;
#pragma unroll
    for (int u = 0; u < 8; ++u) s += t[u];
  }
  const int nt = tile % ct_n, kf = tile / ct_n;
  const int orow = kf * 16 + (tid >> 4), ocol = nt * 16 + (tid & 15);
  if (ocol < N) {
    if (orow < K) dW[(size_t)orow * N + ocol] = s;
    else if (orow == K) db[ocol] = s;
  }
}

// =============================================================================================
// LARGE BATCHES (B >= TOWER_BIG_MIN_B).  The tiles above are built for batch 256: one 16x16 output tile per workgroup,
// its reduction split over the 4 waves, operands straight from global memory -- 1 792 workgroups of 7.5 us each for
// dcn.py's first layer at batch 4 096 (phase stamps), 19 us per launch with 7 of them per CU competing for the memory
// pipeline, every one re-reading its 40 KB of input rows and 40 KB of weights.  Here ONE workgroup owns a 16-row tile
// across ALL output columns: the input rows are read once (coalesced float4, all in flight) into LDS with the previous
// layer's batch-norm + dropout applied on the way, each wave takes column tiles w, w + 4, .. with its own accumulators
// and streams its weight columns with the loads of two k-steps in flight.  256 workgroups at batch 4 096 = one per CU.
// =============================================================================================
constexpr int TOWER_BIG_MIN_B = 1024;
constexpr int TOWER_BIG_MIN_K = 256;     // input width from which the large-batch kernels are used (see rsx_tower_fwd_layer)
// Round 4: at batch >= 4096 the BACKWARD of a 100-wide layer also wins through the large-batch kernel (dcn.py's second layer:
// step 0.2094 -> 0.2052 ms, A/B on one box; the forward does not: 0.2077 with it alone, 0.2053 with both) -- the rule stays a
// function of the layer shape and batch only.  RSX_TOWER_BIG_MIN_K_FWD / _BWD override the width threshold (A/B runs).
constexpr int TOWER_BIG_MIN_K_BWD_4096 = 64;
static inline int tower_big_min_k(bool bwd, int B) {
  static const int f = getenv("RSX_TOWER_BIG_MIN_K_FWD") ? atoi(getenv("RSX_TOWER_BIG_MIN_K_FWD")) : 0;
  static const int b = getenv("RSX_TOWER_BIG_MIN_K_BWD") ? atoi(getenv("RSX_TOWER_BIG_MIN_K_BWD")) : 0;
  if (bwd) return b > 0 ? b : (B >= 4096 ? TOWER_BIG_MIN_K_BWD_4096 : TOWER_BIG_MIN_K);
  // (round 6: with the fixed-point statistics every consumer workgroup adds 8 rows of its input columns' accumulators; the
  // small-tile forward of a 100-wide layer at batch 4 096 is 1 792 workgroups doing so, the large-batch form 256 -- dcn.py's
  // step 0.1850 -> 0.1817 ms)
  return f > 0 ? f : (B >= 4096 ? TOWER_BIG_MIN_K_BWD_4096 : TOWER_BIG_MIN_K);
}
// LDS row stride of a 16-row operand tile read with ds_read_b128 by lane (row = lane & 15, k-quarter = lane >> 4): a stride
// == 8 (mod 64) floats puts the 16 lanes of every hardware lane group on distinct banks
__host__ __device__ inline int big_stride(int K) { return (K + 127) / 128 * 128 + 8; }     // (>= K rounded up to 128 floats)

// NTW: column tiles per wave (N <= 64 * NTW).  RID: the launch carries riders (the step's dedup sort / a slice of the
// optimizer sweep as extra workgroups) -- a compile-time switch, because their ~100 KB of code and their registers would
// otherwise sit in the rider-free launches of the windowed step as well; the tiles themselves are the same code.
template <int NTW, bool RID>
__global__ __launch_bounds__(256) void tower_fwd_big_k(const FwdArgs p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  if (p.zero_n > 0) stat_zero(p.zero_p, p.zero_n);
  if (RID) {
    if ((int)blockIdx.x >= p.n_own + p.n_sort) {
      RSX_RIDE_SWEEP(p.sweep.args, p.sweep.blk_lo + (blockIdx.x - p.n_own - p.n_sort));
      return;
    }
    if ((int)blockIdx.x >= p.n_own) {
      RSX_RIDE_SORT(p.sort, blockIdx.x - p.n_own, reinterpret_cast<uint32_t*>(lds));
      return;
    }
  }
  const int stb = p.fstat_prev == nullptr ? 36 : 32;      // (profiling build) stamp slots: first layer 36.., others 32..
  RSX_STAMP(stb + 0, blockIdx.x == 0);
  const int KP = big_stride(p.K);
  float* sA = lds;                    // [16][KP]
  float* sc = lds + 16 * KP;          // [K] scale, [K] shift of the previous layer's batch-norm
  float* sh = sc + p.K;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int rt = blockIdx.x;
  const bool first = p.fstat_prev == nullptr;
  const int K4 = p.K >> 2, nks = ((p.K + 127) >> 7) << 3;      // k-steps of 16, padded to a multiple of 8 (A is zero there)
  if (!first) {
    for (int k = tid; k < p.K; k += 256) {
      if (p.gamma_prev == nullptr) {   // previous layer without batch-norm (din/din.py MLP): dropout only
        sc[k] = 1.f;
        sh[k] = 0.f;
        continue;
      }
      float mean, rstd;
      bn_col_stats(p.fstat_prev, p.RT, p.K, k, p.inv_B, mean, rstd);
      const float inv = rstd * p.gamma_prev[k];
      sc[k] = inv;
      sh[k] = p.beta_prev[k] - mean * inv;
      if (rt == 0) {
        p.bn_prev_out[k] = mean;
        p.bn_prev_out[p.K + k] = rstd;
      }
    }
    __syncthreads();
  }
  // (1) the tile's 16 input rows -> LDS (previous layer's batch-norm + dropout applied), zero beyond K and beyond the batch
  {
    const DropRng dr = drop_make(first ? 0.f : p.rate, p.mask_prev, p.rng_step, p.seed, p.layer_prev);
    const int tot = 16 * K4;
    constexpr int NLD = 12;                      // float4 per thread and batch: K = 624 is 9.75 -> ONE batch, all in flight
    for (int e0 = tid; e0 < tot; e0 += 256 * NLD) {
      float4 v[NLD];
#pragma unroll
      for (int u = 0; u < NLD; ++u) {
        const int e = e0 + 256 * u < tot ? e0 + 256 * u : tot - 1;
        const int r = e / K4, c4 = e - r * K4;
        const int row = rt * TM + r < p.B ? rt * TM + r : p.B - 1;
        v[u] = reinterpret_cast<const float4*>(p.in + (size_t)row * p.K)[c4];
      }
#pragma unroll
      for (int u = 0; u < NLD; ++u) {
        const int e = e0 + 256 * u;
        if (e < tot) {
          const int r = e / K4, c4 = e - r * K4, k = 4 * c4;
          const int row = rt * TM + r;
          float4 a = v[u];
          if (!first) {
            const float4 s4 = *reinterpret_cast<const float4*>(sc + k);
            const float4 h4 = *reinterpret_cast<const float4*>(sh + k);
            const size_t ei = (size_t)(row < p.B ? row : p.B - 1) * p.K + k;
            a.x = (a.x * s4.x + h4.x) * drop_mul(dr, p.mask_prev, ei);
            a.y = (a.y * s4.y + h4.y) * drop_mul(dr, p.mask_prev, ei + 1);
            a.z = (a.z * s4.z + h4.z) * drop_mul(dr, p.mask_prev, ei + 2);
            a.w = (a.w * s4.w + h4.w) * drop_mul(dr, p.mask_prev, ei + 3);
          }
          if (row >= p.B) a = F4Z;
          *reinterpret_cast<float4*>(sA + r * KP + k) = a;
        }
      }
    }
    const int padw = nks * 16 - p.K;               // 0 .. 60 floats per row (a multiple of 4)
    for (int e = tid; e < 16 * (padw >> 2); e += 256) {
      const int r = e / (padw >> 2), c4 = e - r * (padw >> 2);
      *reinterpret_cast<float4*>(sA + r * KP + p.K + 4 * c4) = F4Z;
    }
  }
  __syncthreads();
  RSX_STAMP(stb + 1, blockIdx.x == 0);
  // (2) wave w: column tiles w, w + 4, ...; k = 16 ks + 4 kq + t on both operands
  const int i = lane & 15, kq = lane >> 4;
  const int NT = (p.N + 15) >> 4;
  int colc[NTW];
#pragma unroll
  for (int j = 0; j < NTW; ++j) {
    const int col = (w + 4 * j) * 16 + i;
    colc[j] = col < p.N ? col : p.N - 1;           // (columns past N: computed on a valid address, never stored)
  }
  f32x4 acc[NTW];
#pragma unroll
  for (int j = 0; j < NTW; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  auto ldb = [&](const int ks, float (&b)[NTW][4]) {
    const int kk = ks * 16 + 4 * kq;
    const int kc = kk < p.K ? kk : p.K - 4;        // (a k-step past K: A is zero there)
    const float* wr = p.W + (size_t)kc * p.N;
#pragma unroll
    for (int j = 0; j < NTW; ++j) {
#pragma unroll
      for (int t = 0; t < 4; ++t) b[j][t] = wr[(size_t)t * p.N + colc[j]];
    }
  };
  // four k-steps of weight loads in flight (one k-step of MFMAs is 256-320 cycles, an L2 round trip under load more than
  // twice that: with two in flight every k-step waited for its operands); the column tiles of a wave alternate inside a
  // k-step so that consecutive MFMAs never share an accumulator (40-cycle dependent latency against a 32-cycle issue)
  float bq[4][NTW][4];
  // (every load of the loop is UNCONDITIONAL, on a clamped k-step: behind a condition the compiler can no longer count the
  // loads in flight and waits for all of them -- vmcnt(0) -- at the head of every iteration, which is no prefetch at all)
#pragma unroll
  for (int u = 0; u < 4; ++u) ldb(u, bq[u]);
  const float* ar = sA + i * KP + 4 * kq;
  // (8 k-steps per trip: at the loop's back edge the compiler falls back to "wait for everything", so a trip should be long)
  float4 an = *reinterpret_cast<const float4*>(ar);               // (the A fragment one k-step ahead: its LDS latency hides too)
  for (int ks = 0; ks < nks; ks += 8) {
#pragma unroll
    for (int uu = 0; uu < 8; ++uu) {
      const int u = uu & 3;
      const float4 a = an;
      an = *reinterpret_cast<const float4*>(ar + 16 * (ks + uu + 1 < nks ? ks + uu + 1 : ks + uu));
      const float av[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
      for (int t = 0; t < 4; ++t) {
#pragma unroll
        for (int j = 0; j < NTW; ++j) acc[j] = mfma16(av[t], bq[u][j][t], acc[j]);
      }
      // (a scheduling fence: left alone, the compiler sinks these loads towards their use four k-steps later and the
      // loop runs with ~1 k-step of loads in flight instead of 4)
      __builtin_amdgcn_sched_barrier(0);
      ldb(ks + uu + 4 < nks ? ks + uu + 4 : nks - 1, bq[u]);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  // (3) epilogue: lane (i, kq) holds rows 4 kq + r of column tile j's column i; bias + relu, the per-row-tile column sums
  // for the batch-norm statistics (4 rows in-lane, then the 4 lane quarters in order)
  RSX_STAMP(stb + 2, blockIdx.x == 0);
  float biasv[NTW];
#pragma unroll
  for (int j = 0; j < NTW; ++j) biasv[j] = p.bias[colc[j]];
  __builtin_amdgcn_s_waitcnt(0x0F70);                // vmcnt(0) BEFORE the first store: afterwards a wait for a load waits for the stores
#pragma unroll
  for (int j = 0; j < NTW; ++j) {
    const int ct = w + 4 * j;
    const int col = ct * 16 + i;
    const bool cok = ct < NT && col < p.N;
    const float bias = biasv[j];
    double s1 = 0.0, s2 = 0.0;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = rt * TM + 4 * kq + r;
      float o = acc[j][r] + bias;
      o = o > 0.f ? o : 0.f;
      if (cok && row < p.B) {
        p.a_out[(size_t)row * p.N + col] = o;
        s1 += (double)o;
        s2 += (double)o * (double)o;
      }
    }
    if (p.fstat_out != nullptr) {
      s1 += __shfl_xor(s1, 16); s2 += __shfl_xor(s2, 16);
      s1 += __shfl_xor(s1, 32); s2 += __shfl_xor(s2, 32);
      if (kq == 0 && cok) stat_put(p.fstat_out, p.RT, rt, p.N, col, s1, s2);
    }
  }
  RSX_STAMP(stb + 3, blockIdx.x == 0);
}

// ---------------------------------------------------------------------------------------------
// backward layer, large batches.  Workgroup families of one launch (256 threads each):
//   [0, n_din)        d(input): ONE workgroup per 16-row tile across ALL K input columns.  da = relu'(a) * BNbwd(dy) of the
//                     tile is formed once into LDS; wave w takes column tiles w, w + 4, .. (NTX accumulators per pass) and
//                     streams the W rows of its columns as float4 (the reduction index n is contiguous in W[k, :]), the
//                     loads of the next k-step in flight.
//   [n_din, +n_dw)    dW: workgroup = (64 input features, one block of p.ksb * 16 batch rows); wave w owns 16 features x
//                     ALL N outputs.  Both MFMA operands are indexed [batch row][feature / output], i.e. "transposed": 32
//                     rows at a time are staged in LDS (in' with batch-norm + dropout, da, coalesced float4 loads) and read
//                     back column-wise with ds_read_b32.  Row K of the feature range is the ones-row (-> db).  Partial
//                     result [sb][K + 1 rounded to 16][N rounded to 16] -> tower_reduce_dw_big_k adds the blocks in order.
//   then              the head-partial reduce, the riding sort, the sweep slice -- as in tower_bwd_k.
// ---------------------------------------------------------------------------------------------
constexpr int BIG_DW_ROWS = 32;        // batch rows per LDS stage of a dW workgroup
__host__ __device__ inline int big_ld32(int n) { return ((n + 15) & ~15) + 4; }   // row stride == 4 (mod 8): ds_read_b32 columns

// NTX: d(input) column tiles per wave and pass; NTD: column tiles of the dW family (N <= 16 * NTD); RID: riders present
// XR (round 6): the launch's last n_xr workgroups run dcn.py's cross-layer backward (cross_device.h cross_bwd4_body<3>: it needs
// only the head's gradient, so it leaves the step's dependent chain; the FIRST layer's launch then accumulates onto its dX)
// (XR: two waves per SIMD asked for -- the two bodies together took 276 registers, one workgroup per CU for the whole launch)
template <int NTX, int NTD, bool RID, bool XR = false>
__global__ __launch_bounds__(256, XR ? 2 : 1) void tower_bwd_big_k(const BwdArgs p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  if (p.zero_n > 0) stat_zero(p.zero_p, p.zero_n);
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int i = lane & 15, kq = lane >> 4;
  const int bid = blockIdx.x;
  if constexpr (XR) {
    const int xb = bid - (p.n_din + p.n_dw + p.n_head);
    if (xb >= 0) {
      cross_bwd4_body<3>(p.xr, p.xr_epw, lds, xb);
      return;
    }
  }
  const float Bf = (float)p.B;
  const bool first = p.bn_prev == nullptr;
  const bool nobn = p.gamma == nullptr;
  const int N4 = p.N >> 2, NP = (p.N + 15) & ~15;
  float* Lm = lds; float* Lr = lds + NP; float* Lk = lds + 2 * NP; float* Ls = lds + 3 * NP; float* Lx = lds + 4 * NP;
  float* tile = lds + 5 * NP;                                   // 16-byte aligned (NP % 16 == 0)
  if (bid < p.n_din) {
    // ================= d(input): rows rt*16 .., all K columns =================
    const int rt = bid;
    const int sx = first ? 40 : 48;                              // (profiling build) stamp slots
    RSX_STAMP(sx + 0, bid == 0);
    const int LD = big_stride(p.N);
    float* sA = tile;                                           // [16][LD]
    float4 av[4], dv[4];                                        // 16 * N/4 float4 over 256 threads: <= 4 each (N <= 256)
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int e = tid + 256 * u;
      const int ec = e < 16 * N4 ? e : 16 * N4 - 1;
      const int r = ec / N4, c4 = ec - r * N4;
      const size_t row = (size_t)(rt * TM + r < p.B ? rt * TM + r : p.B - 1);
      av[u] = reinterpret_cast<const float4*>(p.a + row * p.N)[c4];
      dv[u] = reinterpret_cast<const float4*>(p.dy + row * p.N)[c4];
    }
    for (int c = tid; c < NP; c += 256) {
      const ColBwd cb = bwd_col(p, c < p.N ? c : p.N - 1);
      Lm[c] = cb.mean; Lr[c] = cb.rstd; Lk[c] = cb.k1; Ls[c] = cb.sdy; Lx[c] = cb.sdx;
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int e = tid + 256 * u;
      if (e < 16 * N4) {
        const int r = e / N4, c = 4 * (e - r * N4);
        const float rokf = rt * TM + r < p.B ? 1.f : 0.f;
        ColBwd c0, c1, c2, c3;
        c0.mean = Lm[c]; c0.rstd = Lr[c]; c0.k1 = Lk[c]; c0.sdy = Ls[c]; c0.sdx = Lx[c];
        c1.mean = Lm[c + 1]; c1.rstd = Lr[c + 1]; c1.k1 = Lk[c + 1]; c1.sdy = Ls[c + 1]; c1.sdx = Lx[c + 1];
        c2.mean = Lm[c + 2]; c2.rstd = Lr[c + 2]; c2.k1 = Lk[c + 2]; c2.sdy = Ls[c + 2]; c2.sdx = Lx[c + 2];
        c3.mean = Lm[c + 3]; c3.rstd = Lr[c + 3]; c3.k1 = Lk[c + 3]; c3.sdy = Ls[c + 3]; c3.sdx = Lx[c + 3];
        float4 o;
        o.x = da_of(av[u].x, dv[u].x, c0, Bf, nobn) * rokf;
        o.y = da_of(av[u].y, dv[u].y, c1, Bf, nobn) * rokf;
        o.z = da_of(av[u].z, dv[u].z, c2, Bf, nobn) * rokf;
        o.w = da_of(av[u].w, dv[u].w, c3, Bf, nobn) * rokf;
        *reinterpret_cast<float4*>(sA + r * LD + c) = o;
      }
    }
    const int nks = ((p.N + 31) >> 5) << 1;                      // k-steps of 16, padded to an even count (da is zero there)
    for (int e = tid; e < 16 * ((nks * 16 - p.N) >> 2); e += 256) {    // zero the k padding
      const int r = e / ((nks * 16 - p.N) >> 2), c4 = e - r * ((nks * 16 - p.N) >> 2);
      *reinterpret_cast<float4*>(sA + r * LD + p.N + 4 * c4) = F4Z;
    }
    __syncthreads();
    RSX_STAMP(sx + 1, bid == 0);
    const DropRng dr = drop_make(first ? 0.f : p.rate, p.mask_prev, p.rng_step, p.seed, p.layer_prev);
    const float* ar = sA + i * LD + 4 * kq;
    const int tpw = (p.ct_k - w + 3) / 4;                        // column tiles of this wave: w, w + 4, ...
    for (int j0 = 0; j0 < tpw; j0 += NTX) {
      int kcol[NTX];
      const float* wrow[NTX];
#pragma unroll
      for (int j = 0; j < NTX; ++j) {
        const int ct = w + 4 * (j0 + j);
        kcol[j] = ct * 16 + i;
        const int kc = (j0 + j < tpw && kcol[j] < p.K) ? kcol[j] : p.K - 1;
        wrow[j] = p.W + (size_t)kc * p.N + 4 * kq;
      }
      f32x4 acc[NTX];
#pragma unroll
      for (int j = 0; j < NTX; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
      auto ldw = [&](const int ks, float4 (&b)[NTX]) {
        const int off = 16 * ks + 4 * kq < p.N ? 16 * ks : p.N - 4 - 4 * kq;     // (past N: A is zero there)
#pragma unroll
        for (int j = 0; j < NTX; ++j) b[j] = *reinterpret_cast<const float4*>(wrow[j] + off);
      };
      float4 b0[NTX], b1[NTX];
      ldw(0, b0);
      ldw(1, b1);
      for (int ks = 0; ks < nks; ks += 2) {                       // (loads unconditional on clamped k-steps: see tower_fwd_big_k)
        const float4 a0 = *reinterpret_cast<const float4*>(ar + 16 * ks);
        const float4 a1 = *reinterpret_cast<const float4*>(ar + 16 * (ks + 1));
#pragma unroll
        for (int j = 0; j < NTX; ++j) acc[j] = mfma16(a0.x, b0[j].x, acc[j]);
#pragma unroll
        for (int j = 0; j < NTX; ++j) acc[j] = mfma16(a0.y, b0[j].y, acc[j]);
#pragma unroll
        for (int j = 0; j < NTX; ++j) acc[j] = mfma16(a0.z, b0[j].z, acc[j]);
#pragma unroll
        for (int j = 0; j < NTX; ++j) acc[j] = mfma16(a0.w, b0[j].w, acc[j]);
        __builtin_amdgcn_sched_barrier(0);
        ldw(ks + 2 < nks ? ks + 2 : nks - 1, b0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < NTX; ++j) acc[j] = mfma16(a1.x, b1[j].x, acc[j]);
#pragma unroll
        for (int j = 0; j < NTX; ++j) acc[j] = mfma16(a1.y, b1[j].y, acc[j]);
#pragma unroll
        for (int j = 0; j < NTX; ++j) acc[j] = mfma16(a1.z, b1[j].z, acc[j]);
#pragma unroll
        for (int j = 0; j < NTX; ++j) acc[j] = mfma16(a1.w, b1[j].w, acc[j]);
        __builtin_amdgcn_sched_barrier(0);
        ldw(ks + 3 < nks ? ks + 3 : nks - 1, b1);
        __builtin_amdgcn_sched_barrier(0);
      }
      RSX_STAMP(sx + 2, bid == 0 && j0 == 0);
      // epilogue: lane (i, kq) holds rows 4 kq + r of column kcol[j]: dropout backward of the previous layer and the
      // partial sums of ITS batch-norm backward over the tile's 16 rows (4 in-lane, then the 4 lane quarters in order)
#pragma unroll
      for (int j = 0; j < NTX; ++j) {
        const bool cok = j0 + j < tpw && kcol[j] < p.K;
        const int kc = cok ? kcol[j] : p.K - 1;
        float xin[4] = {0.f, 0.f, 0.f, 0.f}, bnm = 0.f, bnr = 0.f;
        if (!first) {     // (uniform)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int orow = rt * TM + 4 * kq + r;
            xin[r] = p.in[(size_t)(orow < p.B ? orow : p.B - 1) * p.K + kc];
          }
          bnm = p.bn_prev[kc];
          bnr = p.bn_prev[p.K + kc];
        }
        double s1 = 0.0, s2 = 0.0;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int orow = rt * TM + 4 * kq + r;
          float o = acc[j][r];
          if (orow < p.B && cok) {
            if (!first) {
              o *= drop_mul(dr, p.mask_prev, (size_t)orow * p.K + kc);
              const float xh = (xin[r] - bnm) * bnr;
              s1 += (double)o;
              s2 += (double)o * (double)xh;
            }
            float* dst = p.dy_prev + (size_t)orow * p.K + kc;
            if (first && p.acc_dx) o = *dst + o;                 // (dX: an earlier launch's rider wrote the cross layers' share)
            *dst = o;
          }
        }
        if (!first) {
          s1 += __shfl_xor(s1, 16); s2 += __shfl_xor(s2, 16);
          s1 += __shfl_xor(s1, 32); s2 += __shfl_xor(s2, 32);
          if (kq == 0 && cok) stat_put(p.bstat_prev, p.RT, rt, p.K, kc, s1, s2);
        }
      }
    }
    RSX_STAMP(sx + 3, bid == 0);
    return;
  }
  if (bid < p.n_din + p.n_dw) {
    // ================= dW: features fg*64 .. +64 (wave w: 16 of them), batch rows sbi * ksb * 16 .. =================
    const int t_id = bid - p.n_din;
    const int sbi = t_id % p.sb, fg = t_id / p.sb;
    const int sw = first ? 44 : 52;
    RSX_STAMP(sw + 0, t_id == 0);
    const int LDI = big_ld32(64), LDD = big_ld32(p.N);
    float* sI = tile;                                            // [2][32][LDI]
    float* sD = tile + 2 * BIG_DW_ROWS * LDI;                    // [2][32][LDD]
    float* fS = sD + 2 * BIG_DW_ROWS * LDD;                      // [64] scale, [64] shift of the workgroup's features
    for (int c = tid; c < NP; c += 256) {
      const ColBwd cb = bwd_col(p, c < p.N ? c : p.N - 1);
      Lm[c] = cb.mean; Lr[c] = cb.rstd; Lk[c] = cb.k1; Ls[c] = cb.sdy; Lx[c] = cb.sdx;
      if (fg == 0 && sbi == 0 && c < p.N && !nobn) {
        p.dgamma[c] = cb.sdx;
        p.dbeta[c] = cb.sdy;
      }
    }
    if (tid < 64) {
      const int feat = fg * 64 + tid;
      float fsc = 1.f, fsh = 0.f;
      if (!first && feat < p.K && p.gamma_prev != nullptr) {
        const float inv = p.bn_prev[p.K + feat] * p.gamma_prev[feat];
        fsc = inv;
        fsh = p.beta_prev[feat] - p.bn_prev[feat] * inv;
      }
      fS[tid] = fsc;
      fS[64 + tid] = fsh;
    }
    const DropRng dr = drop_make(first ? 0.f : p.rate, p.mask_prev, p.rng_step, p.seed, p.layer_prev);
    const int row0 = sbi * p.ksb * 16;
    const int row1 = row0 + p.ksb * 16 < p.B ? row0 + p.ksb * 16 : p.B;
    const int NT = NP >> 4;
    f32x4 acc[NTD];                                              // every wave: all NT <= NTD column tiles
    // in' tile: 32 rows x 64 features = 512 float4, 2 per thread; a, dy: 32 x N/4 float4 each, <= 4 per thread (N <= 128)
#pragma unroll
    for (int j = 0; j < NTD; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int nst = (row1 - row0 + BIG_DW_ROWS - 1) / BIG_DW_ROWS;
    __syncthreads();
    float4 xv[2], avv[4], dvv[4];                    // (N <= 128, host-checked: 32 * N / 4 float4 over 256 threads)
    // global loads of one 32-row stage (unconditional, rows clamped into the block) ...
    auto ld_stage = [&](const int st) {
      const int rb = row0 + st * BIG_DW_ROWS;
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int e = tid + 256 * u, r = e >> 4, c4 = e & 15;
        const int row = rb + r < row1 ? rb + r : row1 - 1;
        const int feat = fg * 64 + 4 * c4;
        xv[u] = *reinterpret_cast<const float4*>(p.in + (size_t)row * p.K + (feat + 3 < p.K ? feat : p.K - 4));
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int e = tid + 256 * u;
        const int ec = e < BIG_DW_ROWS * N4 ? e : BIG_DW_ROWS * N4 - 1;
        const int r = ec / N4, c4 = ec - r * N4;
        const size_t row = (size_t)(rb + r < row1 ? rb + r : row1 - 1);
        avv[u] = reinterpret_cast<const float4*>(p.a + row * p.N)[c4];
        dvv[u] = reinterpret_cast<const float4*>(p.dy + row * p.N)[c4];
      }
    };
    // ... and their way into the stage's LDS buffer: in' = dropout(BN(in)) (feature K: the ones-row), da = relu' * BNbwd(dy)
    auto put_stage = [&](const int st) {
      float* cI = sI + (st & 1) * BIG_DW_ROWS * LDI;
      float* cD = sD + (st & 1) * BIG_DW_ROWS * LDD;
      const int rb = row0 + st * BIG_DW_ROWS;
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int e = tid + 256 * u, r = e >> 4, c4 = e & 15;
        const int row = rb + r;
        const int feat = fg * 64 + 4 * c4;
        const float v[4] = {xv[u].x, xv[u].y, xv[u].z, xv[u].w};      // (K % 4 == 0: a float4 is all inside K or all outside)
        float o[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const int ft = feat + t;
          float x = v[t];
          if (!first) x = (x * fS[4 * c4 + t] + fS[64 + 4 * c4 + t]) *
                          drop_mul(dr, p.mask_prev, (size_t)(row < row1 ? row : row1 - 1) * p.K + (ft < p.K ? ft : p.K - 1));
          x = ft < p.K ? x : (ft == p.K ? 1.f : 0.f);            // feature K: the ones-row that yields db
          o[t] = row < row1 ? x : 0.f;
        }
        *reinterpret_cast<float4*>(cI + r * LDI + 4 * c4) = make_float4(o[0], o[1], o[2], o[3]);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int e = tid + 256 * u;
        if (e < BIG_DW_ROWS * N4) {
          const int r = e / N4, c = 4 * (e - r * N4);
          const float rokf = rb + r < row1 ? 1.f : 0.f;
          ColBwd c0, c1, c2, c3;
          c0.mean = Lm[c]; c0.rstd = Lr[c]; c0.k1 = Lk[c]; c0.sdy = Ls[c]; c0.sdx = Lx[c];
          c1.mean = Lm[c + 1]; c1.rstd = Lr[c + 1]; c1.k1 = Lk[c + 1]; c1.sdy = Ls[c + 1]; c1.sdx = Lx[c + 1];
          c2.mean = Lm[c + 2]; c2.rstd = Lr[c + 2]; c2.k1 = Lk[c + 2]; c2.sdy = Ls[c + 2]; c2.sdx = Lx[c + 2];
          c3.mean = Lm[c + 3]; c3.rstd = Lr[c + 3]; c3.k1 = Lk[c + 3]; c3.sdy = Ls[c + 3]; c3.sdx = Lx[c + 3];
          float4 o;
          o.x = da_of(avv[u].x, dvv[u].x, c0, Bf, nobn) * rokf;
          o.y = da_of(avv[u].y, dvv[u].y, c1, Bf, nobn) * rokf;
          o.z = da_of(avv[u].z, dvv[u].z, c2, Bf, nobn) * rokf;
          o.w = da_of(avv[u].w, dvv[u].w, c3, Bf, nobn) * rokf;
          *reinterpret_cast<float4*>(cD + r * LDD + c) = o;
        }
      }
    };
    for (int e = tid; e < 2 * BIG_DW_ROWS * (NP - p.N); e += 256) {   // the k padding of the da tiles (both buffers), once
      const int r = e / (NP - p.N), c = e - r * (NP - p.N);
      sD[r * LDD + p.N + c] = 0.f;
    }
    RSX_STAMP(sw + 1, t_id == 0);
    ld_stage(0);
    for (int st = 0; st < nst; ++st) {
      const float* cI = sI + (st & 1) * BIG_DW_ROWS * LDI;
      const float* cD = sD + (st & 1) * BIG_DW_ROWS * LDD;
      put_stage(st);
      ld_stage(st + 1 < nst ? st + 1 : st);          // the next stage's rows are in flight under this stage's MFMAs
      __syncthreads();
#pragma unroll
      for (int ks = 0; ks < BIG_DW_ROWS / 16; ++ks) {
        float a[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) a[t] = cI[(16 * ks + 4 * kq + t) * LDI + 16 * w + i];
        float b[NTD][4];
#pragma unroll
        for (int j = 0; j < NTD; ++j) {
#pragma unroll
          for (int t = 0; t < 4; ++t) b[j][t] = cD[(16 * ks + 4 * kq + t) * LDD + 16 * (j < NT ? j : NT - 1) + i];
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
#pragma unroll
          for (int j = 0; j < NTD; ++j)
            if (j < NT) acc[j] = mfma16(a[t], b[j][t], acc[j]);
        }
      }
      // (the next stage writes the other LDS buffer; the barrier at its end also orders this stage's reads before the
      // stage after next overwrites this buffer)
    }
    RSX_STAMP(sw + 2, t_id == 0);
    // partial tile: lane (i, kq) holds features fg*64 + 16 w + 4 kq + r, column 16 j + i
    const int KR = p.ct_k1 * 16;
    float* out = p.dwp + (size_t)sbi * KR * NP;
#pragma unroll
    for (int j = 0; j < NTD; ++j) {
      if (j < NT) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int feat = fg * 64 + 16 * w + 4 * kq + r;
          if (feat < KR) out[(size_t)feat * NP + 16 * j + i] = acc[j][r];
        }
      }
    }
    return;
  }
  if (RID) {
    if (bid >= p.n_din + p.n_dw + p.n_head + p.n_sort) {
      RSX_RIDE_SWEEP(p.sweep.args, p.sweep.blk_lo + (bid - (p.n_din + p.n_dw + p.n_head + p.n_sort)));
      return;
    }
    if (bid >= p.n_din + p.n_dw + p.n_head) {
      RSX_RIDE_SORT(p.sort, bid - (p.n_din + p.n_dw + p.n_head), reinterpret_cast<uint32_t*>(lds));
      return;
    }
  }
  tower_head_reduce(p, tile, reinterpret_cast<double*>(tile + 1024));
}

// dW[k, n] = sum over the sb row blocks (ascending) of the partial [sb][KR][NP]; row K -> db.  grid = KR * NP / 1024 floats
__global__ __launch_bounds__(256) void tower_reduce_dw_big_k(const float* __restrict__ dwp, float* __restrict__ dW,
                                                             float* __restrict__ db, int sb, int K, int N, int KR, int NP) {
  const size_t e4 = (size_t)blockIdx.x * 256 + threadIdx.x;       // float4 index into [KR][NP]
  const size_t tot4 = (size_t)KR * NP / 4;
  if (e4 >= tot4) return;
  const float4* src = reinterpret_cast<const float4*>(dwp);
  float4 s = F4Z;
  for (int q = 0; q < sb; q += 8) {
    float4 t[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) t[u] = src[(size_t)(q + u < sb ? q + u : sb - 1) * tot4 + e4];
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (q + u < sb) s = f4_add(s, t[u]);
  }
  const int k = (int)((e4 * 4) / NP), n = (int)((e4 * 4) - (size_t)k * NP);
  const float v[4] = {s.x, s.y, s.z, s.w};
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    if (n + t < N) {
      if (k < K) dW[(size_t)k * N + n + t] = v[t];
      else if (k == K) db[n + t] = v[t];
    }
  }
}

// The dW reductions of ALL layers of a backward pass in one launch (a layer's reduce is first needed by the optimizer: one
// launch at the end of the backward pass instead of one per layer; every job keeps its own fixed summation order).
__global__ __launch_bounds__(256) void tower_reduce_dw_jobs_k(const DwReduceJobs g) {
  RSX_DW_REDUCE_SELECT(g, blockIdx.x, jb, blk)
  dw_reduce_job_block(jb, blk);
}

extern "C" int rsx_tower_reduce_dw_jobs(const rsx_dw_reduce_job* jobs_h, int njobs, rsx_stream_t stream) {
  if (njobs == 0) return RSX_OK;
  DwReduceJobs g;
  uint32_t end = 0;
  const int rc = dw_reduce_pack(jobs_h, njobs, g, &end);
  if (rc != RSX_OK) return rc;
  RSX_LAUNCH(tower_reduce_dw_jobs_k, dim3(end), dim3(256), 0, rsx_s(stream), g);
  RSX_CHECK_LAUNCH();
  return RSX_OK;
}

// validates a piggy-backed sort job for a 256-thread carrier launch and grows the launch's dynamic LDS if needed
// consumers read ONE fixed-point row when the batch is large (stat_fix_add / stat_fix_read above)
static inline int stat_rows(int B) { return B >= TOWER_FIX_MIN_B ? 0 : (B + TM - 1) / TM; }       // 0: the fixed-point row


// ------------------------------------------------------------------ C ABI ----------------------------
extern "C" int rsx_tower_fwd_layer(const float* in, const float* W, const float* bias, float* a_out,
                                   double* fstat_out, const double* fstat_prev, const float* gamma_prev,
                                   const float* beta_prev, const float* mask_prev, float* bn_prev_out,
                                   const uint32_t* rng_step, uint32_t seed, int layer, float dropout_rate, int B,
                                   int K, int N, const rsx_sort_job* sort_h, const rsx_adam_slice* sweep_h,
                                   double* zero_stats, int zero_n, rsx_stream_t stream) {
  if (B < 0 || K <= 0 || N <= 0) return RSX_EINVAL;
  if (B == 0) return RSX_OK;
  if (!in || !W || !bias || !a_out) return RSX_EINVAL;
  if (zero_n < 0 || (zero_n > 0 && !zero_stats)) return RSX_EINVAL;
  if (K % 4 != 0) return RSX_EUNSUPPORTED;
  // previous layer with batch-norm: gamma, beta and bn_prev_out all given; without (din MLP): all three NULL
  if (fstat_prev != nullptr && gamma_prev != nullptr && (!beta_prev || !bn_prev_out)) return RSX_EINVAL;
  if (gamma_prev == nullptr && (beta_prev || bn_prev_out)) return RSX_EINVAL;
  if (dropout_rate < 0.f || dropout_rate >= 1.f) return RSX_EINVAL;
  FwdArgs p;
  p.in = in; p.W = W; p.bias = bias; p.a_out = a_out; p.fstat_out = fstat_out;
  p.fstat_prev = fstat_prev; p.gamma_prev = gamma_prev; p.beta_prev = beta_prev; p.mask_prev = mask_prev;
  p.bn_prev_out = bn_prev_out;
  p.rng_step = rng_step; p.seed = seed; p.layer_prev = (uint32_t)(layer - 1);
  p.rate = dropout_rate;
  p.B = B; p.K = K; p.N = N; p.RT = stat_rows(B);
  p.zero_p = zero_stats; p.zero_n = zero_n;
  p.inv_B = 1.0 / (double)B;
  p.ct = (N + 15) / 16;
  p.n_own = p.ct * ((B + TM - 1) / TM);
  const int rcs = adam_build_slice(sweep_h, p.sweep);
  if (rcs != RSX_OK) return rcs;
  size_t lds = ((size_t)2 * K + 1024 + 256) * sizeof(float);
  p.n_sort = 0;
  if (sort_h != nullptr) {
    const int rc = sort_job_args(*sort_h, p.sort, &lds);
    if (rc != RSX_OK) return rc;
    p.n_sort = sort_h->F;
  }
  static const int big_env = getenv("RSX_TOWER_BIG") ? atoi(getenv("RSX_TOWER_BIG")) : 1;   // (0: the small-batch tiles, A/B runs)
  const size_t lds_big = ((size_t)16 * big_stride(K) + 2 * (size_t)K) * sizeof(float);
  // (A/B on one box, r03: the large-batch kernels win on the WIDE first layer -- dcn.py K = 624: forward 19 -> 16 us, backward
  // 48 -> 41 us incl. the reduce -- and lose on 100-wide layers, whose launches are latency chains either way: DIN's 96 /
  // 100 / 52-wide MLP at batch 1 024 ran 22 us per step slower through them.  The choice is a function of the layer shape
  // only, so every path of one model and batch size takes the same kernels.)
  if (big_env && B >= TOWER_BIG_MIN_B && K >= tower_big_min_k(false, B) && N <= 256 && lds_big <= 64 * 1024) {
    // one workgroup per 16-row tile across all N columns (see tower_fwd_big_k)
    p.n_own = (B + TM - 1) / TM;
    size_t l2 = lds_big > lds ? lds_big : lds;                     // (lds: what a riding sort needs)
    const dim3 grid(p.n_own + p.n_sort + p.sweep.n_blk);
    const bool rid = p.n_sort + (int)p.sweep.n_blk > 0;
#define RSX_FWD_BIG(NTW)                                                                                      \
  do {                                                                                                        \
    if (rid) RSX_LAUNCH((tower_fwd_big_k<NTW, true>), grid, dim3(256), l2, rsx_s(stream), p);         \
    else RSX_LAUNCH((tower_fwd_big_k<NTW, false>), grid, dim3(256), l2, rsx_s(stream), p);            \
  } while (0)
    if (N <= 64) RSX_FWD_BIG(1);
    else if (N <= 128) RSX_FWD_BIG(2);
    else if (N <= 192) RSX_FWD_BIG(3);
    else RSX_FWD_BIG(4);
#undef RSX_FWD_BIG
    RSX_CHECK_LAUNCH();
    return RSX_OK;
  }
  if (p.n_sort + (int)p.sweep.n_blk > 0)
    RSX_LAUNCH(tower_fwd_k<true>, dim3(p.n_own + p.n_sort + p.sweep.n_blk), dim3(256), lds, rsx_s(stream), p);
  else
    RSX_LAUNCH(tower_fwd_k<false>, dim3(p.n_own), dim3(256), lds, rsx_s(stream), p);
  RSX_CHECK_LAUNCH();
  return RSX_OK;
}

extern "C" int rsx_gather_tower_fwd0(const float* tables, const float* w1, const int32_t* row_off, const int32_t* ids,
                                     float* E, float* S, float* y1, float* y2, uint64_t w1_field_mask, int F, int D,
                                     const float* W, const float* bias, float* a_out, double* fstat_out, int B, int N,
                                     const rsx_sort_job* sort_h, const rsx_adam_slice* sweep_h, double* zero_stats,
                                     int zero_n, rsx_stream_t stream) {
  if (B < 0 || F <= 0 || D <= 0 || N <= 0) return RSX_EINVAL;
  if (B == 0) return RSX_OK;
  if (!tables || !row_off || !ids || !E || !W || !bias || !a_out) return RSX_EINVAL;
  if (zero_n < 0 || (zero_n > 0 && !zero_stats)) return RSX_EINVAL;
  // the envelope (rsx_gather_tower_fwd0_supported): rows of 16 floats, a batch the small-batch tiles serve, an LDS tile that
  // leaves room for 3 workgroups per CU
  if (D != 16 || F > 64 || B >= TOWER_BIG_MIN_B) return RSX_EUNSUPPORTED;
  GatherFwdArgs g;
  FwdArgs& p = g.f;
  p.in = nullptr; p.W = W; p.bias = bias; p.a_out = a_out; p.fstat_out = fstat_out;
  p.fstat_prev = nullptr; p.gamma_prev = nullptr; p.beta_prev = nullptr; p.mask_prev = nullptr; p.bn_prev_out = nullptr;
  p.rng_step = nullptr; p.seed = 0; p.layer_prev = 0u; p.rate = 0.f;
  p.B = B; p.K = F * D; p.N = N; p.RT = stat_rows(B);
  p.zero_p = zero_stats; p.zero_n = zero_n;
  p.inv_B = 1.0 / (double)B;
  p.ct = (N + 15) / 16;
  p.n_own = p.ct * ((B + TM - 1) / TM);
  const int rcs = adam_build_slice(sweep_h, p.sweep);
  if (rcs != RSX_OK) return rcs;
  size_t lds = ((size_t)16 * (p.K + 4) + 1024 + 256) * sizeof(float);
  p.n_sort = 0;
  if (sort_h != nullptr) {
    const int rc = sort_job_args(*sort_h, p.sort, &lds);
    if (rc != RSX_OK) return rc;
    p.n_sort = sort_h->F;
  }
  g.tables = tables; g.w1 = w1; g.row_off = row_off; g.ids = ids; g.E = E; g.S = S; g.y1 = y1; g.y2 = y2;
  g.w1_mask = w1_field_mask; g.F = F;
  g.n_gout = (B + 3) / 4;
  const dim3 grid(p.n_own + g.n_gout + p.n_sort + p.sweep.n_blk);
  const bool rid = p.n_sort + (int)p.sweep.n_blk > 0;
  if (F <= 40) {
    if (rid) RSX_LAUNCH((tower_gather_fwd_k<10, true>), grid, dim3(256), lds, rsx_s(stream), g);
    else RSX_LAUNCH((tower_gather_fwd_k<10, false>), grid, dim3(256), lds, rsx_s(stream), g);
  } else {
    if (rid) RSX_LAUNCH((tower_gather_fwd_k<16, true>), grid, dim3(256), lds, rsx_s(stream), g);
    else RSX_LAUNCH((tower_gather_fwd_k<16, false>), grid, dim3(256), lds, rsx_s(stream), g);
  }
  RSX_CHECK_LAUNCH();
  return RSX_OK;
}

extern "C" int rsx_gather_tower_fwd0_supported(int B, int F, int D) {
  return D == 16 && F > 0 && F <= 64 && B > 0 && B < TOWER_BIG_MIN_B;
}

extern "C" int rsx_tower_head(const float* a_last, const double* fstat_last, const float* gamma, const float* beta,
                              const float* mask, float* bn_out, const float* wd, const float* bd, const float* s0,
                              const float* c0, const float* s1, const float* wo, const float* bo,
                              const float* labels, float* prob, float* dy_last, double* bstat_last,
                              float* dwd_part, double* hpart, float* gs0, float* gs1, const uint32_t* rng_step,
                              uint32_t seed, int layer, float dropout_rate, float loss_scale, int relu0, int relu2,
                              int B, int N, const rsx_adam_slice* sweep_h, rsx_stream_t stream) {
  if (B < 0 || N <= 0) return RSX_EINVAL;
  if (B == 0) return RSX_OK;
  if (N > 256) return RSX_EUNSUPPORTED;
  if (!a_last || !fstat_last || !wd || !bd || !labels || !prob || !dy_last || !bstat_last || !dwd_part || !hpart)
    return RSX_EINVAL;
  if (gamma != nullptr ? (!beta || !bn_out) : (beta != nullptr)) return RSX_EINVAL;   // gamma NULL: no batch-norm
  if (dropout_rate < 0.f || dropout_rate >= 1.f) return RSX_EINVAL;
  HeadArgs p;
  p.a_last = a_last; p.fstat_last = fstat_last; p.gamma = gamma; p.beta = beta; p.mask = mask; p.bn_out = bn_out;
  p.wd = wd; p.bd = bd; p.s0 = s0; p.c0 = c0; p.s1 = s1; p.wo = wo; p.bo = bo; p.labels = labels; p.prob = prob;
  p.dy_last = dy_last; p.bstat_last = bstat_last; p.dwd_part = dwd_part; p.hpart = hpart; p.gs0 = gs0; p.gs1 = gs1;
  p.rng_step = rng_step; p.seed = seed; p.layer = (uint32_t)layer;
  p.rate = dropout_rate;
  p.loss_scale = loss_scale;
  p.relu0 = relu0; p.relu2 = relu2;
  p.B = B; p.N = N; p.RT = stat_rows(B);
  p.inv_B = 1.0 / (double)B;
  p.n_own = (B + TM - 1) / TM;
  const int rcs = adam_build_slice(sweep_h, p.sweep);
  if (rcs != RSX_OK) return rcs;
  const dim3 grid(p.n_own + p.sweep.n_blk);
  // riders: none / a slice of the one-step sweep / a slice of a full window's sweep in small blocks
  const int rid = p.sweep.n_blk == 0 ? 0 : (p.sweep.args.nw == 0 && p.sweep.args.win_u == 0 ? 1 :
                                            (p.sweep.args.nw == ADAM_WMAX && p.sweep.args.win_u == 2 ? 2 : -1));
  if (rid < 0) return RSX_EUNSUPPORTED;      // (partial windows / other block sizes: the stand-alone sweep, rsx_adam_slice_run)
#define RSX_HEAD(CPL)                                                                                   \
  do {                                                                                                  \
    if (rid == 2) RSX_LAUNCH((tower_head_k<CPL, 2>), grid, dim3(256), 0, rsx_s(stream), p);             \
    else if (rid == 1) RSX_LAUNCH((tower_head_k<CPL, 1>), grid, dim3(256), 0, rsx_s(stream), p);        \
    else RSX_LAUNCH((tower_head_k<CPL, 0>), grid, dim3(256), 0, rsx_s(stream), p);                      \
  } while (0)
  if (N <= 64) RSX_HEAD(4);
  else if (N <= 128) RSX_HEAD(8);
  else RSX_HEAD(16);
#undef RSX_HEAD
  RSX_CHECK_LAUNCH();
  return RSX_OK;
}

// ---------------------------------------------------------------------------------------------
// FM head (fm/fm.py:120-133, 146-149): z = wo[0]*relu(y1 + c0) + wo[1]*y2 + bo, prob = sigmoid(z), mean sigmoid-CE, and
// the whole backward of it: gy1 = d loss / d y1, gy2 = d loss / d y2, dwo[2], dbo, dc0, loss.  One workgroup walks the
// batch (the model is the embedding kernels; this is ~10 flops per example) and reduces in fp64 in a fixed order; extra
// workgroups carry a slice of the untouched-row optimizer sweep, as on the tower entry points.
// ---------------------------------------------------------------------------------------------
struct FmHeadArgs {
  const float* y1; const float* y2; const float* c0; const float* wo; const float* bo; const float* labels;
  float* prob; float* gy1; float* gy2; float* dwo; float* dbo; float* dc0; float* loss;
  float loss_scale;
  int B;
  // terms != null (round 4): no reduction here -- every example's contribution to the dense gradients goes to row b of
  // terms [B, stride] in the dense arena's layout, its cross-entropy term to column n_dense (rsx_gather_fm_head's contract: the
  // optimizer launch sums the rows in example order, so this launch and the fused one leave the same bits)
  float* terms;
  int stride, n_dense, off_c0, off_wo, off_bo;
  AdamSlice sweep;
};

__global__ __launch_bounds__(256) void fm_head_k(const FmHeadArgs p) {
  if (blockIdx.x > 0) {
    RSX_RIDE_SWEEP(p.sweep.args, p.sweep.blk_lo + (blockIdx.x - 1));
    return;
  }
  __shared__ double red[4][5];
  __shared__ float tv[256][5];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const float c0 = p.c0[0], w0 = p.wo[0], w1 = p.wo[1], bo = p.bo[0];
  double acc[5] = {0, 0, 0, 0, 0};                   // loss, dwo0, dwo1, dbo, dc0
  if (p.terms != nullptr) {
    // terms mode: groups of 16 consecutive examples added in example order into ONE row each (rsx_gather_fm_head's contract,
    // the same additions in the same order as gather_fm_head_k)
    for (int b0 = 0; b0 < p.B; b0 += 256) {
      const int b = b0 + tid;
      if (b < p.B) {
        const float v0 = p.y1[b] + c0, v1 = p.y2[b], y = p.labels[b];
        const float t0 = v0 > 0.f ? v0 : 0.f;
        const float z = w0 * t0 + w1 * v1 + bo;
        const float pr = 1.f / (1.f + expf(-z));
        const float ce = fmaxf(z, 0.f) - z * y + log1pf(expf(-fabsf(z)));
        const float dz = (pr - y) * p.loss_scale;
        const float g0 = v0 > 0.f ? dz * w0 : 0.f;
        p.prob[b] = pr;
        p.gy1[b] = g0;
        p.gy2[b] = dz * w1;
        tv[tid][0] = g0; tv[tid][1] = dz * t0; tv[tid][2] = dz * v1; tv[tid][3] = dz; tv[tid][4] = ce;
      }
      __syncthreads();
      for (int gi = tid; gi < 16 * p.stride; gi += 256) {             // (group of this chunk, column of its row)
        const int g = gi / p.stride, col = gi - g * p.stride;
        if (b0 + 16 * g >= p.B) continue;
        const int n = p.B - (b0 + 16 * g) < 16 ? p.B - (b0 + 16 * g) : 16;
        const int src = col == p.off_c0 ? 0 : col == p.off_wo ? 1 : col == p.off_wo + 1 ? 2 : col == p.off_bo ? 3 : col == p.n_dense ? 4 : -1;
        float t = 0.f;
        if (src >= 0) {
          t = tv[16 * g][src];
          for (int i = 1; i < n; ++i) t += tv[16 * g + i][src];
        }
        p.terms[(size_t)(b0 / 16 + g) * p.stride + col] = t;
      }
      __syncthreads();
    }
    return;
  }
  for (int b = tid; b < p.B; b += 256) {
    const float v0 = p.y1[b] + c0, v1 = p.y2[b], y = p.labels[b];
    const float t0 = v0 > 0.f ? v0 : 0.f;
    const float z = w0 * t0 + w1 * v1 + bo;
    const float pr = 1.f / (1.f + expf(-z));
    const float ce = fmaxf(z, 0.f) - z * y + log1pf(expf(-fabsf(z)));
    const float dz = (pr - y) * p.loss_scale;
    const float g0 = v0 > 0.f ? dz * w0 : 0.f;
    p.prob[b] = pr;
    p.gy1[b] = g0;
    p.gy2[b] = dz * w1;
    acc[0] += (double)ce;
    acc[1] += (double)(dz * t0);
    acc[2] += (double)(dz * v1);
    acc[3] += (double)dz;
    acc[4] += (double)g0;
  }
#pragma unroll
  for (int k = 0; k < 5; ++k) {
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) acc[k] += __shfl_xor(acc[k], m);
    if (lane == 0) red[w][k] = acc[k];
  }
  __syncthreads();
  if (tid < 5) {
    const double s = ((red[0][tid] + red[1][tid]) + red[2][tid]) + red[3][tid];
    if (tid == 0) p.loss[0] = (float)(s / (double)p.B);
    else if (tid <= 2) p.dwo[tid - 1] = (float)s;
    else if (tid == 3) p.dbo[0] = (float)s;
    else p.dc0[0] = (float)s;
  }
}

extern "C" int rsx_fm_head(const float* y1, const float* y2, const float* c0, const float* wo, const float* bo,
                           const float* labels, float* prob, float* gy1, float* gy2, float* dwo, float* dbo, float* dc0,
                           float* loss, float loss_scale, int B, const rsx_adam_slice* sweep_h, rsx_stream_t stream) {
  return rsx_fm_head_terms(y1, y2, c0, wo, bo, labels, prob, gy1, gy2, dwo, dbo, dc0, loss, nullptr, 0, 0, 0, 0, 0, loss_scale,
                           B, sweep_h, stream);
}

extern "C" int rsx_fm_head_terms(const float* y1, const float* y2, const float* c0, const float* wo, const float* bo,
                                 const float* labels, float* prob, float* gy1, float* gy2, float* dwo, float* dbo, float* dc0,
                                 float* loss, float* terms, int term_stride, int n_dense, int off_c0, int off_wo, int off_bo,
                                 float loss_scale, int B, const rsx_adam_slice* sweep_h, rsx_stream_t stream) {
  if (B <= 0) return B == 0 ? RSX_OK : RSX_EINVAL;
  if (!y1 || !y2 || !c0 || !wo || !bo || !labels || !prob || !gy1 || !gy2) return RSX_EINVAL;
  if (terms == nullptr && (!dwo || !dbo || !dc0 || !loss)) return RSX_EINVAL;
  if (terms != nullptr && (term_stride < n_dense + 1 || term_stride > 64 || (term_stride & 3) || n_dense <= 0 || off_c0 < 0 ||
                           off_wo < 0 || off_bo < 0 || off_c0 >= n_dense || off_wo + 1 >= n_dense || off_bo >= n_dense))
    return RSX_EINVAL;
  FmHeadArgs p{y1, y2, c0, wo, bo, labels, prob, gy1, gy2, dwo, dbo, dc0, loss, loss_scale, B,
               terms, term_stride, n_dense, off_c0, off_wo, off_bo, {}};
  const int rcs = adam_build_slice(sweep_h, p.sweep);
  if (rcs != RSX_OK) return rcs;
  RSX_LAUNCH(fm_head_k, dim3(1 + p.sweep.n_blk), dim3(256), 0, rsx_s(stream), p);
  RSX_CHECK_LAUNCH();
  return RSX_OK;
}

// row blocks a dW tile's batch reduction is split into (1 without workspace or for small batches)
static inline int rsx_tower_dw_blocks(int B, bool have_ws) {
  static const int min_b = getenv("RSX_TOWER_SB_MIN_B") ? atoi(getenv("RSX_TOWER_SB_MIN_B")) : 1024;
  if (!have_ws || B < min_b) return 1;
  // rows per dW row block: 512 measured 2-3 % faster than 256 on dcn.py bs 4096 (fewer partial tiles to write and re-add)
  static const int rows = getenv("RSX_TOWER_SB_ROWS") ? atoi(getenv("RSX_TOWER_SB_ROWS")) : 512;
  const int sb = (B + rows - 1) / rows;
  return sb > 32 ? 32 : sb;
}

extern "C" size_t rsx_tower_bwd_workspace_floats(int B, int K, int N) {
  const size_t small = (size_t)((K + 1 + 15) / 16) * ((N + 15) / 16) * rsx_tower_dw_blocks(B, true) * 256;
  // large-batch layout (tower_bwd_big_k): up to min(32, 320 / feature groups) row blocks of the whole [K + 1 rounded to
  // 16][N rounded to 16] gradient
  const int nfg = (K + 1 + 63) / 64;
  const int sbmax = 320 / nfg < 1 ? 1 : (320 / nfg > 32 ? 32 : 320 / nfg);
  const size_t big = (size_t)sbmax * (((size_t)K + 1 + 15) / 16 * 16) * (((size_t)N + 15) / 16 * 16);
  return B >= TOWER_BIG_MIN_B && big > small ? big : small;
}

// does a backward launch of this shape take tower_bwd_big_k (the kernel that knows the cross rider / the accumulating dX)?
static inline bool tower_bwd_is_big(int B, int K, int N) {
  static const int big_env = getenv("RSX_TOWER_BIG") ? atoi(getenv("RSX_TOWER_BIG")) : 1;
  return big_env && B >= TOWER_BIG_MIN_B && K >= tower_big_min_k(true, B) && (N & 3) == 0 && N <= 128 && (K & 3) == 0;
}
extern "C" int rsx_tower_bwd_cross_ride_supported(int B, int K_last, int N_last, int K_first, int N_first, int dim, int L) {
  if (B <= 0 || dim <= 0) return 0;
  return tower_bwd_is_big(B, K_last, N_last) && K_last <= 128 && tower_bwd_is_big(B, K_first, N_first) && L == 3 &&
         dim % 4 == 0 && dim <= 256 * CROSS_NV && (size_t)4 * (2 * L + 1) * dim * sizeof(float) <= 80 * 1024;
}

extern "C" int rsx_tower_bwd_layer_defer(const float*, const float*, const float*, const float*, const double*, const float*,
                                         const float*, float*, float*, float*, float*, const float*, const float*, const float*,
                                         const float*, float*, double*, const double*, const float*, float*, float*, float*,
                                         float*, float*, float*, const uint32_t*, uint32_t, int, float, int, int, int,
                                         const rsx_sort_job*, const rsx_adam_slice*, float*, rsx_dw_reduce_job*, double*, int,
                                         const rsx_tower_bwd_extra*, rsx_stream_t);
extern "C" int rsx_tower_bwd_layer(const float* in, const float* W, const float* a, const float* dy,
                                   const double* bstat, const float* bn, const float* gamma, float* dW, float* db,
                                   float* dgamma, float* dbeta, const float* bn_prev, const float* gamma_prev,
                                   const float* beta_prev, const float* mask_prev, float* dy_prev,
                                   double* bstat_prev, const double* hpart, const float* dwd_part, float* dwd,
                                   float* dbd, float* dwo, float* dbo, float* dc0, float* loss,
                                   const uint32_t* rng_step, uint32_t seed, int layer, float dropout_rate, int B,
                                   int K, int N, const rsx_sort_job* sort_h, const rsx_adam_slice* sweep_h,
                                   float* dw_partials, rsx_stream_t stream) {
  return rsx_tower_bwd_layer_defer(in, W, a, dy, bstat, bn, gamma, dW, db, dgamma, dbeta, bn_prev, gamma_prev, beta_prev, mask_prev,
                                   dy_prev, bstat_prev, hpart, dwd_part, dwd, dbd, dwo, dbo, dc0, loss, rng_step, seed, layer,
                                   dropout_rate, B, K, N, sort_h, sweep_h, dw_partials, nullptr, nullptr, 0, nullptr, stream);
}

extern "C" int rsx_tower_bwd_layer_defer(const float* in, const float* W, const float* a, const float* dy,
                                         const double* bstat, const float* bn, const float* gamma, float* dW, float* db,
                                         float* dgamma, float* dbeta, const float* bn_prev, const float* gamma_prev,
                                         const float* beta_prev, const float* mask_prev, float* dy_prev,
                                         double* bstat_prev, const double* hpart, const float* dwd_part, float* dwd,
                                         float* dbd, float* dwo, float* dbo, float* dc0, float* loss,
                                         const uint32_t* rng_step, uint32_t seed, int layer, float dropout_rate, int B,
                                         int K, int N, const rsx_sort_job* sort_h, const rsx_adam_slice* sweep_h,
                                         float* dw_partials, rsx_dw_reduce_job* reduce_out, double* zero_stats,
                                         int zero_n, const rsx_tower_bwd_extra* extra_h, rsx_stream_t stream) {
  if (reduce_out != nullptr) reduce_out->sb = 0;          // (0: nothing left to reduce for this layer)
  const bool xride = extra_h != nullptr && extra_h->x0 != nullptr;
  if (xride) {
    if (!extra_h->cW || !extra_h->cB || !extra_h->s || !extra_h->gz || !extra_h->wout || !extra_h->dX || !extra_h->dcW ||
        !extra_h->dcB || !extra_h->dwout || !extra_h->workspace || !extra_h->reduce_out)
      return RSX_EINVAL;
    extra_h->reduce_out->n = 0;
    if (!rsx_tower_bwd_cross_ride_supported(B, K, N, K, N, extra_h->dim, extra_h->L) || sort_h != nullptr ||
        (sweep_h != nullptr) || bn_prev == nullptr)
      return RSX_EUNSUPPORTED;                              // (the carrying launch: a non-first layer through tower_bwd_big_k, no riders)
  }
  if (extra_h != nullptr && extra_h->accumulate_dx && bn_prev != nullptr) return RSX_EINVAL;     // (first layer only)
  if (B < 0 || K <= 0 || N <= 0) return RSX_EINVAL;
  if (B == 0) return RSX_OK;
  if (zero_n < 0 || (zero_n > 0 && !zero_stats)) return RSX_EINVAL;
  if (!in || !W || !a || !dy || !bstat || !bn || !dW || !db || !dy_prev) return RSX_EINVAL;
  if (gamma != nullptr && (!dgamma || !dbeta)) return RSX_EINVAL;           // gamma NULL: this layer has no batch-norm
  if (bn_prev != nullptr && !bstat_prev) return RSX_EINVAL;
  if (bn_prev != nullptr && gamma_prev != nullptr && !beta_prev) return RSX_EINVAL;   // gamma_prev NULL: previous layer has none
  if (hpart != nullptr && (!dwd_part || !dwd || !dbd || !loss)) return RSX_EINVAL;
  if (dropout_rate < 0.f || dropout_rate >= 1.f) return RSX_EINVAL;
  BwdArgs p;
  p.in = in; p.W = W; p.a = a; p.dy = dy; p.bstat = bstat; p.bn = bn; p.gamma = gamma;
  p.dW = dW; p.db = db; p.dgamma = dgamma; p.dbeta = dbeta;
  p.bn_prev = bn_prev; p.gamma_prev = gamma_prev; p.beta_prev = beta_prev; p.mask_prev = mask_prev;
  p.dy_prev = dy_prev; p.bstat_prev = bstat_prev;
  p.hpart = hpart; p.dwd_part = dwd_part; p.dwd = dwd; p.dbd = dbd; p.dwo = dwo; p.dbo = dbo; p.dc0 = dc0;
  p.loss = loss;
  p.rng_step = rng_step; p.seed = seed; p.layer_prev = (uint32_t)(layer - 1);
  p.rate = dropout_rate;
  p.has_wo = dwo != nullptr;
  p.B = B; p.K = K; p.N = N; p.RT = stat_rows(B); p.RTh = (B + TM - 1) / TM;
  p.zero_p = zero_stats; p.zero_n = zero_n;
  p.acc_dx = extra_h != nullptr ? extra_h->accumulate_dx : 0;
  p.n_xr = 0; p.xr_epw = 1;
  p.xr = CrossBwdArgs{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0, 0, 0};
  if (xride) {
    p.xr_epw = B <= 1024 ? 1 : (B <= 8192 ? 2 : 4);              // (cross.hip cross_epw4; 4 / 8 per wave at 4 096: +5 us / equal)
    p.n_xr = (B + 4 * p.xr_epw - 1) / (4 * p.xr_epw);
    p.xr = CrossBwdArgs{extra_h->x0, extra_h->cW, extra_h->cB, extra_h->s, nullptr, extra_h->gz, extra_h->wout, extra_h->dX,
                        extra_h->workspace, 0, B, extra_h->dim, extra_h->L};
  }
  p.ct_k = (K + 15) / 16;
  p.ct_k1 = (K + 1 + 15) / 16;          // +1: the ones-row that yields db
  p.ct_n = (N + 15) / 16;
  p.sb = rsx_tower_dw_blocks(B, dw_partials != nullptr);
  static const int rtw_env = getenv("RSX_TOWER_RTW") ? atoi(getenv("RSX_TOWER_RTW")) : 4;
  p.din_rtw = p.sb > 1 ? rtw_env : 1;
  // grouped LDS-staged d(input) tiles (RSX_TOWER_DXG=0: the one-tile form).  Round 2 measured them 1 % worse in the windowed
  // step and left them off; round 6's phase stamps show what the first layer's launch is bound by at batch 256 -- DISPATCH: 624
  // one-tile d(input) workgroups go out at ~130 per us, so the first dW workgroup (block 624 of 904) starts 4.7 us after the
  // launch does and the launch lasts dispatch + one workgroup's life (10.5 us).  Grouped: 160 + 280 workgroups.  ABAB on one
  // box, deepfm.py bs 256: 0.0583 / 0.0583 ms one-tile against 0.0563 / 0.0563 grouped -- the default since round 6.  NOT
  // chosen per launch: the two forms add the N products of an output in a different order, and the windowed, the step-by-step
  // and the plain path must stay bit-identical to each other.
  static const int dxg_env = getenv("RSX_TOWER_DXG") ? atoi(getenv("RSX_TOWER_DXG")) : 1;
  // Large batches (SPLIT, B >= 1024) always take the grouped form (RSX_TOWER_DXG_SPLIT=0 for A/B runs): there the one-tile
  // workgroups (4 row tiles x 1 column tile each, operands as 4-byte strided loads) are throughput-bound -- phase stamps at
  // batch 4096: 12-15 us per workgroup, 2 496 of them over ~1 000 slots -- while a grouped workgroup stages da and W once
  // for four column tiles.  Every path of one batch size makes the same choice, so they stay bit-identical to each other.
  static const int dxg_split_env = getenv("RSX_TOWER_DXG_SPLIT") ? atoi(getenv("RSX_TOWER_DXG_SPLIT")) : 1;
  // (small batches: only where the one-tile form would launch 256 or more d(input) workgroups -- the 100-wide second layer's 112
  // are not dispatch-bound, and its launch measured 7.3 us grouped against 6.5 one-tile; a rule of the layer's SHAPE, so every
  // path of one model and batch size still makes the same choice)
  const bool dxg_want = p.sb > 1 ? dxg_split_env > 0 : (dxg_env > 0 && p.ct_k * p.RTh >= 256);
  p.dxg = ((N & 3) != 0 || N > 128 || !dxg_want) ? 0 : 4;
  p.n_din = p.dxg > 0 ? ((p.ct_k + p.dxg - 1) / p.dxg) * p.RTh : p.ct_k * ((p.RTh + p.din_rtw - 1) / p.din_rtw);
  p.ksb = ((B + 15) / 16 + p.sb - 1) / p.sb;
  p.dwp = dw_partials;
  p.n_dw = p.ct_k1 * p.ct_n * p.sb;
  p.n_head = hpart != nullptr ? 1 : 0;
  p.n_sort = 0;
  size_t lds = ((size_t)5 * N + 4 + 1024 + 256) * sizeof(float);
  if (p.dxg > 0) {
    const size_t l2 = ((size_t)5 * N + 4 + (size_t)(16 + 64) * ((size_t)((N + 15) & ~15) + 4)) * sizeof(float);
    if (l2 > lds) lds = l2;
    if (lds > 64 * 1024) {       // very wide layers: the one-tile form
      p.dxg = 0;
      p.n_din = p.ct_k * ((p.RTh + p.din_rtw - 1) / p.din_rtw);
      lds = ((size_t)5 * N + 4 + 1024 + 256) * sizeof(float);
    }
  }
  if (sort_h != nullptr) {
    const int rc = sort_job_args(*sort_h, p.sort, &lds);
    if (rc != RSX_OK) return rc;
    p.n_sort = sort_h->F;
  }
  const int rcs = adam_build_slice(sweep_h, p.sweep);
  if (rcs != RSX_OK) return rcs;
  static const int big_env = getenv("RSX_TOWER_BIG") ? atoi(getenv("RSX_TOWER_BIG")) : 1;   // (0: the small-batch tiles, A/B runs)
  if (big_env && B >= TOWER_BIG_MIN_B && K >= tower_big_min_k(true, B) && dw_partials != nullptr && (N & 3) == 0 && N <= 128 &&
      (K & 3) == 0) {
    // one workgroup per 16-row tile for d(input), (64 features x row block) workgroups for dW (see tower_bwd_big_k)
    const int NP = (N + 15) & ~15;
    p.n_din = p.RTh;
    const int nfg = (K + 1 + 63) / 64;
    // dW workgroups wanted: up to 320 -- but a workgroup of either family lives ~20 us and two fit a CU (220 registers), so the
    // launch should not spill a few dozen workgroups into a second round of the 256 CUs: with the 256 d(input) workgroups of
    // batch 4 096, 286 dW workgroups made the launch 41 us, 242 make it 34.7 (round 6; dcn.py 0.173 -> 0.166 ms, ABAB on one box;
    // RSX_TOWER_BIG_DW_WGS: A/B knob)
    static const int dw_env = getenv("RSX_TOWER_BIG_DW_WGS") ? atoi(getenv("RSX_TOWER_BIG_DW_WGS")) : 0;
    const int room = 2 * 256 - p.n_din - p.n_head;
    const int dw_wgs = dw_env > 0 ? dw_env : (room >= 128 && room < 320 ? room : 320);
    int sbw = dw_wgs / nfg < 1 ? 1 : (dw_wgs / nfg > 32 ? 32 : dw_wgs / nfg);         // row blocks wanted (more, shorter blocks
                                                                                      // measured slower: the reduce grows with them)
    int rows = ((B + sbw - 1) / sbw + BIG_DW_ROWS - 1) / BIG_DW_ROWS * BIG_DW_ROWS;  // rows per block: a multiple of the LDS stage
    p.ksb = rows / 16;
    p.sb = (B + rows - 1) / rows;
    p.n_dw = nfg * p.sb;
    size_t lb = (size_t)5 * NP + 16 * (size_t)big_stride(N);
    const size_t ldw = (size_t)5 * NP + 2 * BIG_DW_ROWS * ((size_t)big_ld32(64) + big_ld32(N)) + 128;
    if (ldw > lb) lb = ldw;
    if ((size_t)5 * NP + 1024 + 64 > lb) lb = (size_t)5 * NP + 1024 + 64;
    lb *= sizeof(float);
    if (lb > lds) lds = lb;                                       // (lds: what a riding sort needs)
    const bool wide = K > 128;                                    // d(input) column tiles per wave and pass: 5 (K = 624: two passes), else 2
    const bool rid = p.n_sort + (int)p.sweep.n_blk > 0;
    if (xride) {
      // the cross layers' backward as the launch's last workgroups (4 waves, (2L + 1) dim floats of LDS per wave)
      const size_t lx = (size_t)4 * (2 * extra_h->L + 1) * extra_h->dim * sizeof(float);
      if (lx > lds) lds = lx;
      if (wide || rid || lds > 80 * 1024) return RSX_EUNSUPPORTED;
      const dim3 gx(p.n_din + p.n_dw + p.n_head + p.n_xr);
      auto launch = [&](auto kern) -> int {
        static bool attr_set = false;                             // (> 64 KB of dynamic LDS per workgroup)
        if (!attr_set) {
          if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024) != hipSuccess)
            return RSX_EUNSUPPORTED;
          attr_set = true;
        }
        RSX_LAUNCH(kern, gx, dim3(256), lds, rsx_s(stream), p);
        return RSX_OK;
      };
      const int rcx = N <= 64 ? launch(tower_bwd_big_k<2, 4, false, true>) : launch(tower_bwd_big_k<2, 8, false, true>);
      if (rcx != RSX_OK) return rcx;
      RSX_CHECK_LAUNCH();
      *extra_h->reduce_out = rsx_cross_reduce_job{extra_h->workspace, extra_h->dcW, extra_h->dcB, extra_h->dwout, p.n_xr,
                                                  (2 * extra_h->L + 1) * extra_h->dim, extra_h->L, extra_h->dim};
    } else {
    const dim3 grid(p.n_din + p.n_dw + p.n_head + p.n_sort + (int)p.sweep.n_blk);
#define RSX_BWD_BIG(NTX, NTD)                                                                                 \
  do {                                                                                                        \
    if (rid) RSX_LAUNCH((tower_bwd_big_k<NTX, NTD, true>), grid, dim3(256), lds, rsx_s(stream), p);   \
    else RSX_LAUNCH((tower_bwd_big_k<NTX, NTD, false>), grid, dim3(256), lds, rsx_s(stream), p);      \
  } while (0)
    if (N <= 64) { if (wide) RSX_BWD_BIG(5, 4); else RSX_BWD_BIG(2, 4); }
    else { if (wide) RSX_BWD_BIG(5, 8); else RSX_BWD_BIG(2, 8); }
#undef RSX_BWD_BIG
    RSX_CHECK_LAUNCH();
    }
    const int KR = p.ct_k1 * 16;
    if (reduce_out != nullptr) {
      *reduce_out = rsx_dw_reduce_job{dw_partials, dW, db, p.sb, K, N, 1};
      return RSX_OK;
    }
    RSX_LAUNCH(tower_reduce_dw_big_k, dim3((unsigned)(((size_t)KR * NP / 4 + 255) / 256)), dim3(256), 0, rsx_s(stream),
                       dw_partials, dW, db, p.sb, K, N, KR, NP);
    RSX_CHECK_LAUNCH();
    return RSX_OK;
  }
  if (xride || p.acc_dx) return RSX_EUNSUPPORTED;            // (only tower_bwd_big_k knows these roles: rsx_tower_bwd_cross_ride_supported)
  const int total = p.n_din + p.n_dw + p.n_head + p.n_sort + (int)p.sweep.n_blk;
  const bool rid = p.n_sort + (int)p.sweep.n_blk > 0;        // (no riders: the variant with their code compiled out)
  if (p.sb > 1) {
    if (rid) RSX_LAUNCH((tower_bwd_k<true, true>), dim3(total), dim3(256), lds, rsx_s(stream), p);
    else RSX_LAUNCH((tower_bwd_k<true, false>), dim3(total), dim3(256), lds, rsx_s(stream), p);
  } else {
    if (rid) RSX_LAUNCH((tower_bwd_k<false, true>), dim3(total), dim3(256), lds, rsx_s(stream), p);
    else RSX_LAUNCH((tower_bwd_k<false, false>), dim3(total), dim3(256), lds, rsx_s(stream), p);
  }
  RSX_CHECK_LAUNCH();
  if (p.sb > 1) {
    if (reduce_out != nullptr) {
      *reduce_out = rsx_dw_reduce_job{dw_partials, dW, db, p.sb, K, N, 0};
      return RSX_OK;
    }
    RSX_LAUNCH(tower_reduce_dw_k, dim3(p.ct_k1 * p.ct_n), dim3(256), 0, rsx_s(stream), dw_partials, dW, db, p.sb,
                       p.ct_n, K, N);
    RSX_CHECK_LAUNCH();
  }
  return RSX_OK;
}

#ifdef RSX_STAMPS
// profiling build only: copy the tower kernels' phase stamps (100 MHz wall clock ticks) to the host
extern "C" int rsx_dbg_stamps_tower(unsigned long long* out_h) {
  return hipMemcpyFromSymbol(out_h, HIP_SYMBOL(rsx_stamps_d), sizeof(unsigned long long) * 64) == hipSuccess ? RSX_OK : RSX_ELAUNCH;
}
#endif
