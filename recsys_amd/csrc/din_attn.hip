// DIN attention MLP, fused (fp32 MFMA), forward and backward.  Reference: din/din.py:103-121 `_attention` --
//   att_in = concat[h, q, h*q, h-q]  [B*P, 4K];  a1 = dropout(relu(att_in.W0 + b0));  a2 = dropout(relu(a1.W1 + b1));
//   w = a2.W2 + b2  (one logit per history position; no softmax, no scaling)
// with h = history embedding rows [B*P, K], q = the query embedding [B, K] tiled over the P positions.  SURVEY.md 8a row
// a-10: M = B*P = 102 400 rows at the reference's batch; through library GEMMs + element-wise kernels this is ~100
// launches and ~0.5 GB of [M, 128] / [M, 80] round trips per call.  Here one launch per direction:
//   * the [M, 4K] concat is never materialised: a lane builds its A operands from the h / q float4s it holds;
//   * weights live in LDS (transposed / padded so that one ds_read_b128 feeds four k-steps);
//   * layer outputs change from the MFMA C layout to the A layout through a per-wave LDS tile, never through HBM
//     (a1 / a2 are written once for the backward pass);
//   * dropout is the counter hash of the fused tower (drop_device.h) or an injected mask (parity tests).
// One wave = one 16-row tile; k-permutation as in tower.hip / cin.hip: k-step 4*kb + t uses k = 16*kb + 4*(lane>>4) + t.
#include "drop_device.h"
#include <type_traits>
#include "rsx_common.h"
#include "gather_rows_device.h"
#include "step_riders_device.h"
#include "split_device.h"
RSX_STAMP_DECL

typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 att_mfma(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

struct AttnFwdArgs {
  const float* H;       // [M, K]
  const float* q;       // [B, K]
  const float* W0;      // [4K, N1]
  const float* b0;      // [N1]
  const float* W1;      // [N1, N2]
  const float* b1;      // [N2]
  const float* W2;      // [N2]
  const float* b2;      // [1]
  float* a1;            // [M, N1]  relu output, before dropout (saved for backward)
  float* a2;            // [M, N2]
  float* w;             // [M]
  const float* mask1;   // [M, N1] keep masks (parity tests) or null
  const float* mask2;   // [M, N2]
  const uint32_t* rng_step;
  uint32_t seed, layer0;
  float rate;
  int M, P, N1, N2;
  // optional row list (rsx_din_valid_rows): only the rows[0 .. count[0]) -- the history positions that are not padding --
  // are evaluated; a1 / a2 / w / masks / RNG stay indexed by the ORIGINAL row, so nothing downstream changes
  const int32_t* rows;
  const int32_t* count;
};

// Stages a row-major global matrix src[R][C] (C a run-time count) into LDS -- as it is (dst[r * ld + c]) or transposed
// (dst[c * ld + r]) -- with T threads: 8 unconditional loads in flight per thread and trip, the (row, column) of an element
// advanced incrementally.  (The per-element loops this replaces divided by C and waited for one conditional load per trip:
// 40 + 13 dependent round trips in the forward kernel's prologue, ~a quarter of the launch.)
template <int T, bool TRANS>
__device__ __forceinline__ void stage_matrix(float* __restrict__ dst, const float* __restrict__ src, const int R, const int C,
                                             const int ld, const int tid) {
  const int total = R * C;
  if (total <= 0) return;
  const int dc = T % C, dr = T / C;
  int c = tid % C, r = tid / C;
  for (int base = 0; base < total; base += 8 * T) {
    float v[8];
    int rr[8], cc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int e = base + j * T + tid;
      v[j] = src[e < total ? e : total - 1];
      rr[j] = r;
      cc[j] = c;
      c += dc;
      r += dr;
      if (c >= C) { c -= C; ++r; }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j)
      if (base + j * T + tid < total) dst[TRANS ? cc[j] * ld + rr[j] : rr[j] * ld + cc[j]] = v[j];
  }
}

// The same, untransposed, in 16-byte pieces (C % 4 == 0, src 16-byte aligned, ld % 4 == 0): a quarter of the load and LDS-store
// instructions, up to 8 loads in flight per thread -- the backward kernel's 13 440 weights arrive in one trip instead of four.
template <int T>
__device__ __forceinline__ void stage_matrix4(float* __restrict__ dst, const float* __restrict__ src, const int R, const int C,
                                              const int ld, const int tid) {
  const int C4 = C >> 2, total = R * C4;
  if (total <= 0) return;
  const float4* __restrict__ s4 = reinterpret_cast<const float4*>(src);
  for (int base = 0; base < total; base += 8 * T) {
    float4 v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int e = base + j * T + tid;
      v[j] = s4[e < total ? e : total - 1];
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int e = base + j * T + tid;
      if (e < total) {
        const int r = e / C4, c = e - r * C4;
        *reinterpret_cast<float4*>(dst + r * ld + 4 * c) = v[j];
      }
    }
  }
}

// KB = K/16, NT1 = ceil(N1/16), NT2 = ceil(N2/16).  grid = ceil(M / (16 FWD_WAVES)), block = 64 FWD_WAVES.
// dyn LDS (floats): 16*NT1*(64*KB+4) + 16*NT2*(16*NT1+4) + FWD_WAVES*16*(16*NT1+4) + 16*NT1 + 2*16*NT2
// 16 waves per workgroup (round 3; 4 before): the kernel issues about as many VALU cycles (operand construction, layer
// transitions, dropout hash) as MFMA cycles per wave -- SQ counters: 9.6 k against 11.2 k of a 62 k-cycle wave life -- and
// two workgroups of 4 waves per CU (LDS-limited: 58 KB of staged weights + 5.4 KB of transposition scratch per wave) left 2
// waves per SIMD to overlap them; 16 waves share ONE copy of the weights (144 KB in all), so 4 waves per SIMD fit.
constexpr int FWD_WAVES = 16;
template <int KB, int NT1, int NT2>
__global__ __launch_bounds__(64 * FWD_WAVES) void din_attn_fwd_k(const AttnFwdArgs p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int K = 16 * KB, K4 = 4 * K, LD0 = K4 + 4, N1P = 16 * NT1, LD1 = N1P + 4, N2P = 16 * NT2;
  float* sW0 = lds;                       // [N1P][LD0]  W0 transposed: sW0[n][k]
  float* sW1 = sW0 + N1P * LD0;           // [N2P][LD1]  W1 transposed: sW1[n2][n1]
  float* sS = sW1 + N2P * LD1;            // [FWD_WAVES][16][LD1]  a1 (after dropout) in row-major for the A layout
  float* sb0 = sS + FWD_WAVES * 16 * LD1; // [N1P]
  float* sb1 = sb0 + N1P;                 // [N2P]
  float* sw2 = sb1 + N2P;                 // [N2P]
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int i = lane & 15, kq = lane >> 4;
  // weights: zero fill (padding rows / columns), then coalesced global reads scattered into the transposed layout
  constexpr int FT = 64 * FWD_WAVES;
  for (int e = tid; e < (N1P * LD0 + N2P * LD1) / 4; e += FT) reinterpret_cast<float4*>(sW0)[e] = F4Z;
  __syncthreads();
  stage_matrix<FT, true>(sW0, p.W0, K4, p.N1, LD0, tid);        // sW0[n][k] = W0[k][n]
  stage_matrix<FT, true>(sW1, p.W1, p.N1, p.N2, LD1, tid);      // sW1[n2][n1] = W1[n1][n2]
  for (int e = tid; e < N1P; e += FT) sb0[e] = e < p.N1 ? p.b0[e] : 0.f;
  for (int e = tid; e < N2P; e += FT) {
    sb1[e] = e < p.N2 ? p.b1[e] : 0.f;
    sw2[e] = e < p.N2 ? p.W2[e] : 0.f;
  }
  __syncthreads();
  const DropRng d1 = drop_make(p.rate, p.mask1, p.rng_step, p.seed, p.layer0);
  const DropRng d2 = drop_make(p.rate, p.mask2, p.rng_step, p.seed, p.layer0 + 1);
  const int Mv = p.count ? p.count[0] : p.M;     // rows to evaluate (valid history positions, or all)
  const int nblk = (Mv + 16 * FWD_WAVES - 1) / (16 * FWD_WAVES);
  for (int blk = blockIdx.x; blk < nblk; blk += gridDim.x) {    // persistent: the weights are staged once per workgroup
  const int m0 = (blk * FWD_WAVES + wv) * 16;    // position in the (possibly compacted) row list
  const bool mok = m0 + i < Mv;
  const int jc = mok ? m0 + i : 0;
  const int m = p.rows ? p.rows[jc] : jc;        // A-layout row of this lane (original row index)
  const int mc = mok ? m : 0;
  int mrow[4];                                   // original indices of this lane's C-layout rows 4*kq + r
#pragma unroll
  for (int r = 0; r < 4; ++r) mrow[r] = __shfl(mc, 4 * kq + r);   // lane 4*kq + r (kq' = 0) holds that row's index
  float4 h4[KB], q4[KB];
#pragma unroll
  for (int c = 0; c < KB; ++c) {
    h4[c] = mok ? *reinterpret_cast<const float4*>(p.H + (size_t)mc * K + 16 * c + 4 * kq) : F4Z;
    q4[c] = mok ? *reinterpret_cast<const float4*>(p.q + (size_t)(mc / p.P) * K + 16 * c + 4 * kq) : F4Z;
  }
  // ---- layer 0: z1 = [h, q, h*q, h-q] . W0 ----------------------------------------------------------------------
  f32x4 acc1[NT1];
#pragma unroll
  for (int nt = 0; nt < NT1; ++nt) acc1[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int seg = 0; seg < 4; ++seg)
#pragma unroll
    for (int c = 0; c < KB; ++c) {
      const float4 hv = h4[c], qv = q4[c];
      float4 a;
      if (seg == 0) a = hv;
      else if (seg == 1) a = qv;
      else if (seg == 2) a = make_float4(hv.x * qv.x, hv.y * qv.y, hv.z * qv.z, hv.w * qv.w);
      else a = make_float4(hv.x - qv.x, hv.y - qv.y, hv.z - qv.z, hv.w - qv.w);
      const int kb = seg * KB + c;
#pragma unroll
      for (int nt = 0; nt < NT1; ++nt) {
        const float4 b = *reinterpret_cast<const float4*>(sW0 + (16 * nt + i) * LD0 + 16 * kb + 4 * kq);
        acc1[nt] = att_mfma(a.x, b.x, acc1[nt]);
        acc1[nt] = att_mfma(a.y, b.y, acc1[nt]);
        acc1[nt] = att_mfma(a.z, b.z, acc1[nt]);
        acc1[nt] = att_mfma(a.w, b.w, acc1[nt]);
      }
    }
  // C layout: acc1[nt][r] = z1[row 4*kq + r][n = 16*nt + i]
  float* S = sS + wv * 16 * LD1;
#pragma unroll
  for (int nt = 0; nt < NT1; ++nt) {
    const int n = 16 * nt + i;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = 4 * kq + r, mm = mrow[r];
      float v = 0.f;
      if (n < p.N1 && m0 + row < Mv) {
        v = fmaxf(acc1[nt][r] + sb0[n], 0.f);
        p.a1[(size_t)mm * p.N1 + n] = v;
        v *= drop_mul(d1, p.mask1, (size_t)mm * p.N1 + n);
      }
      S[row * LD1 + n] = v;
    }
  }
  __syncthreads();
  // ---- layer 1: z2 = a1d . W1 -----------------------------------------------------------------------------------
  f32x4 acc2[NT2];
#pragma unroll
  for (int nt = 0; nt < NT2; ++nt) acc2[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int kb = 0; kb < NT1; ++kb) {
    const float4 a = *reinterpret_cast<const float4*>(S + i * LD1 + 16 * kb + 4 * kq);
#pragma unroll
    for (int nt = 0; nt < NT2; ++nt) {
      const float4 b = *reinterpret_cast<const float4*>(sW1 + (16 * nt + i) * LD1 + 16 * kb + 4 * kq);
      acc2[nt] = att_mfma(a.x, b.x, acc2[nt]);
      acc2[nt] = att_mfma(a.y, b.y, acc2[nt]);
      acc2[nt] = att_mfma(a.z, b.z, acc2[nt]);
      acc2[nt] = att_mfma(a.w, b.w, acc2[nt]);
    }
  }
  // ---- layer 2: w = a2d . W2 + b2 (row sums over the 16 column lanes) --------------------------------------------
  float pw[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int nt = 0; nt < NT2; ++nt) {
    const int n = 16 * nt + i;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int mm = mrow[r];
      if (n < p.N2 && m0 + 4 * kq + r < Mv) {
        const float v = fmaxf(acc2[nt][r] + sb1[n], 0.f);
        p.a2[(size_t)mm * p.N2 + n] = v;
        pw[r] += v * drop_mul(d2, p.mask2, (size_t)mm * p.N2 + n) * sw2[n];
      }
    }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
#pragma unroll
    for (int s = 1; s < 16; s <<= 1) pw[r] += __shfl_xor(pw[r], s);
    const int mm = mrow[r];
    if (i == 0 && m0 + 4 * kq + r < Mv) p.w[mm] = pw[r] + p.b2[0];
  }
  __syncthreads();                               // S is rewritten by the next block
  }
}


// =====================================================================================================================
// Forward on the bf16 matrix cores with SPLIT operands (round 6; split_device.h: every product exact to 2^-24 of itself, the
// fp32 kernel's tolerance tests unchanged).  The fp32 MFMA above spends 7 040 matrix-core cycles per 16-row tile (4K = 128:
// 32 + 20 k-steps of 4 over 5 + 3 column tiles, 32 cycles each), this one 2 784 (4 + 3 k-steps of 32, six plane products of 16
// cycles each).  Both GEMMs are computed TRANSPOSED -- the weight fragment is the MFMA's first operand, the activation rows
// its second: acc[r] = z[row = lane & 15][n = 16 nt + 4 (lane >> 4) + r] -- so that
//   * a lane holds four consecutive columns of ITS row: a1 / a2 leave as one 16-byte store per column tile and lane (the fp32
//     kernel stores 4-byte elements of four different rows), and
//   * the relu + dropout output is already where the next layer's operand wants it: the reduction index of a GEMM may be
//     permuted freely as long as both operands agree, so layer 1 takes k-step ks, element j of lane (i, kq) to mean
//     k = 16 (2 ks + (j >> 2)) + 4 kq + (j & 3) -- the lane's own accumulators of column tiles 2 ks and 2 ks + 1 -- and W1's
//     fragments are staged in LDS with the same permutation.  No LDS transposition tile, no barrier inside the row loop.
// Weights: split into their three bf16 planes while they are staged (fragment-major: one ds_read_b128 per plane and fragment).
// dyn LDS: 3 planes x (2 KB x 5 + 3 x 3) fragments x 1 KiB + biases = 87.6 KiB (K = 32), 57.6 KiB (K = 16).
// =====================================================================================================================
constexpr int FSP_WAVES = 16;
template <int KB>
__global__ __launch_bounds__(64 * FSP_WAVES) void din_attn_fwd_split_k(const AttnFwdArgs p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int K = 16 * KB, KS0 = 2 * KB, NT1 = 5, NT2 = 3, KS1 = 3;
  constexpr int FT = 64 * FSP_WAVES;
  sp_bf16x8* sW0 = reinterpret_cast<sp_bf16x8*>(lds);              // [3][KS0][NT1][64]
  sp_bf16x8* sW1 = sW0 + 3 * KS0 * NT1 * 64;                       // [3][KS1][NT2][64]
  float* sb0 = reinterpret_cast<float*>(sW1 + 3 * KS1 * NT2 * 64); // [80]
  float* sb1 = sb0 + 16 * NT1;                                     // [48]
  float* sw2 = sb1 + 16 * NT2;                                     // [48]
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int i = lane & 15, kq = lane >> 4;
  // ---- weights -> split fragments (element j of lane (i, kq) of fragment (ks, nt) = W[k][n = 16 nt + i]) ----
  for (int fr = tid; fr < KS0 * NT1 * 64; fr += FT) {
    const int l = fr & 63, nt = (fr >> 6) % NT1, ks = (fr >> 6) / NT1;
    const int n = 16 * nt + (l & 15), k0 = 32 * ks + 8 * (l >> 4);
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = p.W0[(size_t)(k0 + j) * p.N1 + (n < p.N1 ? n : 0)];
    if (n >= p.N1) {
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = 0.f;
    }
    sp_bf16x8 pl[3];
    sp_split8(v, pl);
#pragma unroll
    for (int s = 0; s < 3; ++s) sW0[((s * KS0 + ks) * NT1 + nt) * 64 + l] = pl[s];
  }
  for (int fr = tid; fr < KS1 * NT2 * 64; fr += FT) {
    const int l = fr & 63, nt = (fr >> 6) % NT2, ks = (fr >> 6) / NT2;
    const int n = 16 * nt + (l & 15), q4 = 4 * (l >> 4);
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int k = 16 * (2 * ks + (j >> 2)) + q4 + (j & 3);       // (layer 1's permuted reduction index, see above)
      const bool ok = k < p.N1 && n < p.N2;
      const float x = p.W1[(size_t)(k < p.N1 ? k : 0) * p.N2 + (n < p.N2 ? n : 0)];
      v[j] = ok ? x : 0.f;
    }
    sp_bf16x8 pl[3];
    sp_split8(v, pl);
#pragma unroll
    for (int s = 0; s < 3; ++s) sW1[((s * KS1 + ks) * NT2 + nt) * 64 + l] = pl[s];
  }
  for (int e = tid; e < 16 * NT1; e += FT) sb0[e] = e < p.N1 ? p.b0[e] : 0.f;
  for (int e = tid; e < 16 * NT2; e += FT) {
    sb1[e] = e < p.N2 ? p.b1[e] : 0.f;
    sw2[e] = e < p.N2 ? p.W2[e] : 0.f;
  }
  const DropRng d1 = drop_make(p.rate, p.mask1, p.rng_step, p.seed, p.layer0);
  const DropRng d2 = drop_make(p.rate, p.mask2, p.rng_step, p.seed, p.layer0 + 1);
  const float bias2 = p.b2[0];
  __syncthreads();
  const int Mv = p.count ? p.count[0] : p.M;
  const int nblk = (Mv + 16 * FSP_WAVES - 1) / (16 * FSP_WAVES);
  const sp_f32x4 zf = {0.f, 0.f, 0.f, 0.f};
  for (int blk = blockIdx.x; blk < nblk; blk += gridDim.x) {      // persistent: the weights are staged once per workgroup
    const int jr = (blk * FSP_WAVES + wv) * 16 + i;               // position in the (possibly compacted) row list
    const bool mok = jr < Mv;
    const int jc = mok ? jr : 0;
    const int m = p.rows ? p.rows[jc] : jc;                       // the lane's row (original index)
    // the lane's eight elements of h and q: K = 32 -> [8 kq, 8 kq + 8); K = 16 -> [8 (kq & 1), ..) (kq >> 1 picks the segment)
    const int c0 = KB == 2 ? 8 * kq : 8 * (kq & 1);
    const float4* hp = reinterpret_cast<const float4*>(p.H + (size_t)m * K + c0);
    const float4* qp = reinterpret_cast<const float4*>(p.q + (size_t)(m / p.P) * K + c0);
    const float4 h0 = hp[0], h1 = hp[1], q0 = qp[0], q1 = qp[1];
    const float hv[8] = {h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w};
    const float qv[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
    // ---- layer 0: z1^T = W0^T . [h, q, h*q, h-q]^T ------------------------------------------------------------
    sp_f32x4 acc1[NT1];
#pragma unroll
    for (int nt = 0; nt < NT1; ++nt) acc1[nt] = zf;
#pragma unroll
    for (int ks = 0; ks < KS0; ++ks) {
      float a[8];
      const int seg = KB == 2 ? ks : 2 * ks + (kq >> 1);          // (K = 16: lane-dependent -- selects, no divergence)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float pr = hv[j] * qv[j], df = hv[j] - qv[j];
        a[j] = seg == 0 ? hv[j] : (seg == 1 ? qv[j] : (seg == 2 ? pr : df));
      }
      sp_bf16x8 ap[3];
      sp_split8(a, ap);
#pragma unroll
      for (int nt = 0; nt < NT1; ++nt) {
        sp_bf16x8 wp[3];
#pragma unroll
        for (int s = 0; s < 3; ++s) wp[s] = sW0[((s * KS0 + ks) * NT1 + nt) * 64 + lane];
        acc1[nt] = sp_mma3(wp, ap, acc1[nt]);
        // (without the fence the scheduler hoists the fragment reads of ALL 20 (k-step, tile) pairs -- 240 registers -- to the top)
        if ((nt & 1) == 1) __builtin_amdgcn_sched_barrier(0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    // acc1[nt][r] = z1[row i][n = 16 nt + 4 kq + r]: bias, relu, the a1 store, dropout -- and it stays in registers
    float a1d[NT1 + 1][4];
#pragma unroll
    for (int nt = 0; nt < NT1; ++nt) {
      const int n0 = 16 * nt + 4 * kq;
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = fmaxf(acc1[nt][r] + sb0[n0 + r], 0.f);
      const bool nok = n0 < p.N1;                                 // (N1 % 4 == 0: a quad is valid as a whole)
      if (mok && nok) *reinterpret_cast<float4*>(p.a1 + (size_t)m * p.N1 + n0) = make_float4(v[0], v[1], v[2], v[3]);
#pragma unroll
      for (int r = 0; r < 4; ++r) a1d[nt][r] = nok ? v[r] * drop_mul(d1, p.mask1, (size_t)m * p.N1 + n0 + r) : 0.f;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) a1d[NT1][r] = 0.f;                // (column tile 5: the padding half of layer 1's last k-step)
    // ---- layer 1: z2^T = W1^T . a1d^T ---------------------------------------------------------------------------
    sp_f32x4 acc2[NT2];
#pragma unroll
    for (int nt = 0; nt < NT2; ++nt) acc2[nt] = zf;
#pragma unroll
    for (int ks = 0; ks < KS1; ++ks) {
      const float a[8] = {a1d[2 * ks][0], a1d[2 * ks][1], a1d[2 * ks][2], a1d[2 * ks][3],
                          a1d[2 * ks + 1][0], a1d[2 * ks + 1][1], a1d[2 * ks + 1][2], a1d[2 * ks + 1][3]};
      sp_bf16x8 ap[3];
      sp_split8(a, ap);
#pragma unroll
      for (int nt = 0; nt < NT2; ++nt) {
        sp_bf16x8 wp[3];
#pragma unroll
        for (int s = 0; s < 3; ++s) wp[s] = sW1[((s * KS1 + ks) * NT2 + nt) * 64 + lane];
        acc2[nt] = sp_mma3(wp, ap, acc2[nt]);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    // ---- layer 2: w = a2d . W2 + b2 -----------------------------------------------------------------------------
    float pw = 0.f;
#pragma unroll
    for (int nt = 0; nt < NT2; ++nt) {
      const int n0 = 16 * nt + 4 * kq;
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = fmaxf(acc2[nt][r] + sb1[n0 + r], 0.f);
      const bool nok = n0 < p.N2;
      if (mok && nok) *reinterpret_cast<float4*>(p.a2 + (size_t)m * p.N2 + n0) = make_float4(v[0], v[1], v[2], v[3]);
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (nok) pw += v[r] * drop_mul(d2, p.mask2, (size_t)m * p.N2 + n0 + r) * sw2[n0 + r];
    }
    pw += __shfl_xor(pw, 16);                                     // the row's four column quarters (kq = 0 .. 3)
    pw += __shfl_xor(pw, 32);
    if (kq == 0 && mok) p.w[m] = pw + bias2;
  }
}

static inline size_t attn_fwd_split_lds_bytes(int KB) {
  return (size_t)3 * (2 * KB * 5 + 3 * 3) * 1024 + (16 * 5 + 2 * 16 * 3) * sizeof(float);
}

static inline size_t attn_fwd_lds_floats(int KB, int NT1, int NT2) {
  const size_t LD0 = 64 * KB + 4, N1P = 16 * NT1, LD1 = N1P + 4, N2P = 16 * NT2;
  return N1P * LD0 + N2P * LD1 + FWD_WAVES * 16 * LD1 + N1P + 2 * N2P;
}

extern "C" int rsx_din_attn_fwd(const float* H, const float* q, const float* W0, const float* b0, const float* W1,
                                const float* b1, const float* W2, const float* b2, float* a1, float* a2, float* w,
                                const float* mask1, const float* mask2, const uint32_t* rng_step, uint32_t seed,
                                int layer0, float dropout_rate, const int32_t* rows, const int32_t* count, int B, int P,
                                int K, int N1, int N2, rsx_stream_t stream) {
  if (B < 0 || P <= 0 || K <= 0 || N1 <= 0 || N2 <= 0) return RSX_EINVAL;
  if (B == 0) return RSX_OK;
  if (!H || !q || !W0 || !b0 || !W1 || !b1 || !W2 || !b2 || !a1 || !a2 || !w) return RSX_EINVAL;
  if ((rows == nullptr) != (count == nullptr)) return RSX_EINVAL;
  if (dropout_rate < 0.f || dropout_rate >= 1.f) return RSX_EINVAL;
  if ((K != 16 && K != 32) || N1 > 80 || N2 > 48) return RSX_EUNSUPPORTED;     // instantiated envelope (din/din.py:85)
  AttnFwdArgs p{H, q, W0, b0, W1, b1, W2, b2, a1, a2, w, mask1, mask2, rng_step, seed, (uint32_t)layer0, dropout_rate,
                B * P, P, N1, N2, rows, count};
  // Round 6: the split-operand kernel on the bf16 matrix cores (din.py's shapes: N1, N2 multiples of 4, 16-byte aligned
  // activations); RSX_DIN_ATTN_SPLIT=0: the fp32 MFMA kernel (A/B runs)
  static const int split_env = getenv("RSX_DIN_ATTN_SPLIT") ? atoi(getenv("RSX_DIN_ATTN_SPLIT")) : 1;
  if (split_env && N1 % 4 == 0 && N2 % 4 == 0 && ((((uintptr_t)H | (uintptr_t)q | (uintptr_t)a1 | (uintptr_t)a2) & 15) == 0)) {
    const int nb = (p.M + 16 * FSP_WAVES - 1) / (16 * FSP_WAVES);
    const dim3 g((unsigned)(nb < 256 ? nb : 256)), b(64 * FSP_WAVES);
    const size_t lds = attn_fwd_split_lds_bytes(K / 16);
    if (K == 32) {
      static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(din_attn_fwd_split_k<2>),
                                                         hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      if (attr != hipSuccess) return RSX_EUNSUPPORTED;
      RSX_LAUNCH((din_attn_fwd_split_k<2>), g, b, lds, rsx_s(stream), p);
    } else {
      RSX_LAUNCH((din_attn_fwd_split_k<1>), g, b, lds, rsx_s(stream), p);
    }
    RSX_CHECK_LAUNCH();
    return RSX_OK;
  }
  const int nblk = (p.M + 16 * FWD_WAVES - 1) / (16 * FWD_WAVES);
  const dim3 grid((unsigned)(nblk < 256 ? nblk : 256)), block(64 * FWD_WAVES);    // one workgroup per CU (144 KB of LDS)
  if (K == 32) {
    const size_t lds = attn_fwd_lds_floats(2, 5, 3) * sizeof(float);
    static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(din_attn_fwd_k<2, 5, 3>),
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (attr != hipSuccess) return RSX_EUNSUPPORTED;
    RSX_LAUNCH((din_attn_fwd_k<2, 5, 3>), grid, block, lds, rsx_s(stream), p);
  } else {
    const size_t lds = attn_fwd_lds_floats(1, 5, 3) * sizeof(float);
    static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(din_attn_fwd_k<1, 5, 3>),
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (attr != hipSuccess) return RSX_EUNSUPPORTED;
    RSX_LAUNCH((din_attn_fwd_k<1, 5, 3>), grid, block, lds, rsx_s(stream), p);
  }
  RSX_CHECK_LAUNCH();
  return RSX_OK;
}

// =====================================================================================================================
// Backward.  Persistent workgroups (grid <= 256, one per CU: ~135 KB of LDS) walk 64-row blocks (wave = 16 rows):
//   g2  = dw * W2 * drop2 * (a2 > 0)                 built in the A layout straight from the a2 loads
//   g1  = (g2 . W1^T) * drop1 * (a1 > 0)             MFMA, B operand = W1 row-major in LDS
//   dx  = g1 . W0^T   -> dH = dx_h + dx_hq*q + dx_h-q ;  dq(row) = dx_q + dx_hq*h - dx_h-q      (per row; the sum of
//                                                     dq over the P positions is a second small kernel)
//   dW1 += a1d^T . g2,  dW0 += [h,q,h*q,h-q]^T . g1   reduction over the block's 64 rows, operands from the block's LDS
//                                                     tiles, output tiles split over the 4 waves and kept in registers
//                                                     across ALL blocks of the workgroup
// Every workgroup writes ONE partial of all weight gradients; din_attn_finish_k adds the partials in workgroup order.
// =====================================================================================================================
struct AttnBwdArgs {
  const float* H; const float* q; const float* W0; const float* W1; const float* W2;
  const float* a1; const float* a2; const float* dw;     // dw [M]: gradient of the logits
  float* dH;            // [M, K]
  float* dqr;           // [M, K] per-row query gradient (summed over p by din_attn_finish_k)
  float* part;          // [G, NPART] weight-gradient partials
  const float* mask1; const float* mask2;
  const uint32_t* rng_step;
  uint32_t seed, layer0;
  float rate;
  int M, P, N1, N2, nblk;
  int acc_dH;           // dH += (the pooling backward already wrote its part of the same gradient)
  int ldh;              // row stride of dH (floats; K when dense)
  const int32_t* rows;  // optional row list, as in the forward: only rows[0 .. count[0]) are walked (their dH / dqr written)
  const int32_t* count;
};

template <int KB, int NT1, int NT2>
struct AttnDims {
  static constexpr int K = 16 * KB, K4 = 4 * K, N1P = 16 * NT1, N2P = 16 * NT2, LD1 = N1P + 4, LD2 = N2P + 4, LDH = K + 4;
  static constexpr int MT0 = K4 / 16;                        // row tiles of dW0 (<= 8: one per wave)
};

// 512 threads = 8 waves per workgroup: wave = (half hf, row tile rt).  The two halves of a row tile split the OUTPUT
// columns of every per-row stage (g1 columns, dx columns), the 8 waves split the weight-gradient tiles; per wave that is
// ~170 registers, so two waves share a SIMD and one's VALU / LDS work overlaps the other's MFMAs (with 4 waves of 417
// registers the kernel was instruction-issue bound: 13 non-MFMA instructions per MFMA, one wave per SIMD).
// element `idx` of a per-row array through a 32-bit BYTE offset from the (uniform) base: the load / store takes the
// scalar-base + 32-bit-offset form -- one address register per access instead of a 64-bit pair and its arithmetic
template <class T>
__device__ __forceinline__ T* at32(T* base, uint32_t idx) {
  return reinterpret_cast<T*>(reinterpret_cast<char*>(const_cast<typename std::remove_const<T>::type*>(base)) + idx * (uint32_t)sizeof(T));
}
// VEC (N1 and N2 multiples of 4 -- din.py's 80 / 40): a2 and a1 arrive as ONE 16-byte load per 16-column tile and lane in the A
// layout (row = lane & 15) instead of four 4-byte loads -- 7 instead of 25 load instructions per wave and block for them: the
// CU's address path, not HBM, bounded the loads --; a1 goes through its transposed LDS tile (which S6 needs anyway) to reach
// the C layout S4 wants, dropout applied on the way.
template <int KB, int NT1, int NT2, bool VEC>
__global__ __launch_bounds__(512) void din_attn_bwd_k(const AttnBwdArgs p) {
  using D = AttnDims<KB, NT1, NT2>;
  constexpr int K = D::K, K4 = D::K4, N1P = D::N1P, N2P = D::N2P, LD1 = D::LD1, LD2 = D::LD2;
  constexpr int NH1 = (NT1 + 1) / 2;       // g1 column tiles per half
  constexpr int CH = (KB + 1) / 2;         // dx column blocks (of each of the 4 segments) per half
  extern __shared__ __attribute__((aligned(16))) float lds[];
  RSX_STAMP(0, blockIdx.x == 0);
  float* sW0 = lds;                        // [K4][LD1]   W0 row-major
  float* sW1 = sW0 + K4 * LD1;             // [N1P][LD2]  W1 row-major
  float* sw2 = sW1 + N1P * LD2;            // [N2P]
  // the block's 64-row tiles are stored TRANSPOSED ([feature][row], row stride LDR): the weight-gradient GEMMs reduce over
  // rows, so a lane fetches the operands of 4 k-steps (4 consecutive rows) with one ds_read_b128
  constexpr int LDR = 68;
  float* sG1 = sw2 + N2P;                  // [N1P][LDR]  g1^T
  float* sA1 = sG1 + N1P * LDR;            // [N1P][LDR]  (a1 after dropout)^T
  float* sG2 = sA1 + N1P * LDR;            // [N2P][LDR]  g2^T
  float* sH = sG2 + N2P * LDR;             // [K][LDR]    h^T
  float* sQ = sH + K * LDR;                // [K][LDR]    q^T
  float* sA2 = sQ + K * LDR;               // [N2P][LDR]  (a2 after dropout * dw)^T: db1 / dW2 are column sums of sG2 / sA2,
                                           //             taken by the matrix cores with an all-ones A operand (waves NT1, NT1+1)
  static_assert(NT1 + 2 <= 8, "two spare waves for the ones-row tiles");
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int rt = wave & 3, hf = wave >> 2;
  const int i = lane & 15, kq = lane >> 4;
  const int nt0 = hf * NH1;                // this half's first g1 column tile
  for (int e = tid; e < (K4 * LD1 + N1P * LD2) / 4; e += 512) reinterpret_cast<float4*>(sW0)[e] = F4Z;   // (padding columns / rows)
  __syncthreads();
  if (VEC && (((uintptr_t)p.W0 | (uintptr_t)p.W1) & 15) == 0) {
    stage_matrix4<512>(sW0, p.W0, K4, p.N1, LD1, tid);          // W0 row-major
    stage_matrix4<512>(sW1, p.W1, p.N1, p.N2, LD2, tid);        // W1 row-major
  } else {
    stage_matrix<512, false>(sW0, p.W0, K4, p.N1, LD1, tid);
    stage_matrix<512, false>(sW1, p.W1, p.N1, p.N2, LD2, tid);
  }
  for (int e = tid; e < N2P; e += 512) sw2[e] = e < p.N2 ? p.W2[e] : 0.f;
  const DropRng d1 = drop_make(p.rate, p.mask1, p.rng_step, p.seed, p.layer0);
  const DropRng d2 = drop_make(p.rate, p.mask2, p.rng_step, p.seed, p.layer0 + 1);
  const f32x4 zf = {0.f, 0.f, 0.f, 0.f};
  // weight-gradient accumulators, alive across all blocks: wave w owns row tile w of dW0 (w < 4*KB) and of dW1 (w < NT1);
  // waves NT1 / NT1 + 1 keep ones^T . g2 = db1 and ones^T . (a2d * dw) = dW2 in accW1 (every row of the tile is the sum)
  f32x4 accW0[NT1], accW1[NT2];
#pragma unroll
  for (int b = 0; b < NT1; ++b) accW0[b] = zf;
#pragma unroll
  for (int b = 0; b < NT2; ++b) accW1[b] = zf;
  float db0acc[NH1], db2acc = 0.f;
#pragma unroll
  for (int a = 0; a < NH1; ++a) db0acc[a] = 0.f;
  __syncthreads();

  const int Mv = p.count ? p.count[0] : p.M;       // rows to walk (valid history positions, or all)
  const int nblk = (Mv + 63) / 64;
  // the row-list entry of the NEXT block is requested one block ahead (one dependent HBM round trip less per block)
  auto row_of = [&](const int blk_) -> int {
    const int jr = blk_ * 64 + 16 * rt + i;
    const int jcl = jr < Mv ? jr : 0;
    return p.rows ? p.rows[jcl] : jcl;
  };
  int mci_next = row_of((int)blockIdx.x);
  // The per-row global operands of a block (dw, a2, a1, h, q, the dH to add to) are requested ONE BLOCK AHEAD, right before the
  // previous block's weight-gradient MFMAs (S6): that phase needs only the LDS tiles and the accumulators, the ~45 registers the
  // operands land in are free there, and the ~4 us the loads took at the top of every block (stamps: 30 % of it) are covered by
  // S6 and the barrier behind it.  All loads are unconditional on clamped addresses (rows past the end read row 0).
  int n_mci;
  float n_dz, n_a2[NT2][4], n_a1[NH1][4], n_dh[CH][4];
  float4 n_h[KB], n_q[KB];
  auto issue = [&](const int blk_) {
    const int jr = blk_ * 64 + 16 * rt + i;
    n_mci = jr < Mv ? mci_next : 0;                // A-layout row (original index; 0 when past the end)
    mci_next = row_of(blk_ + (int)gridDim.x);      // (the row-list entry of the block after: one dependent round trip less)
    const uint32_t mcu = (uint32_t)n_mci;          // (32-bit element offsets: the host refuses shapes whose arrays reach 4 GB)
    uint32_t mr[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) mr[r] = (uint32_t)__shfl(n_mci, 4 * kq + r);
    n_dz = *at32(p.dw, mcu);
    if constexpr (VEC) {
#pragma unroll
      for (int kb = 0; kb < NT2; ++kb) {
        const int n0 = 16 * kb + 4 * kq;
        const float4 v = *reinterpret_cast<const float4*>(at32(p.a2, mcu * (uint32_t)p.N2 + (uint32_t)(n0 < p.N2 ? n0 : p.N2 - 4)));
        n_a2[kb][0] = v.x; n_a2[kb][1] = v.y; n_a2[kb][2] = v.z; n_a2[kb][3] = v.w;
      }
#pragma unroll
      for (int u = 0; u < NH1; ++u) {                // A layout: row i, columns 16*(nt0+u) + 4*kq + t
        const int n0 = 16 * (nt0 + u) + 4 * kq;
        const float4 v = *reinterpret_cast<const float4*>(at32(p.a1, mcu * (uint32_t)p.N1 + (uint32_t)(n0 < p.N1 ? n0 : p.N1 - 4)));
        n_a1[u][0] = v.x; n_a1[u][1] = v.y; n_a1[u][2] = v.z; n_a1[u][3] = v.w;
      }
    } else {
#pragma unroll
      for (int kb = 0; kb < NT2; ++kb)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const int n = 16 * kb + 4 * kq + t;
          n_a2[kb][t] = *at32(p.a2, mcu * (uint32_t)p.N2 + (uint32_t)(n < p.N2 ? n : p.N2 - 1));
        }
#pragma unroll
      for (int u = 0; u < NH1; ++u) {                // C layout: rows 4*kq + r, column 16*(nt0+u) + i
        const int n = 16 * (nt0 + u) + i, nc = n < p.N1 ? n : p.N1 - 1;
#pragma unroll
        for (int r = 0; r < 4; ++r) n_a1[u][r] = *at32(p.a1, mr[r] * (uint32_t)p.N1 + (uint32_t)nc);
      }
    }
#pragma unroll
    for (int c = 0; c < KB; ++c) {                 // (half 1 does not use them: wave-uniform)
      n_h[c] = hf == 0 ? *reinterpret_cast<const float4*>(at32(p.H, mcu * (uint32_t)K + (uint32_t)(16 * c + 4 * kq))) : F4Z;
      n_q[c] = hf == 0 ? *reinterpret_cast<const float4*>(at32(p.q, (mcu / (uint32_t)p.P) * (uint32_t)K + (uint32_t)(16 * c + 4 * kq)))
                       : F4Z;
    }
#pragma unroll
    for (int cu = 0; cu < CH; ++cu) {
      const int c = hf * CH + cu, cc = c < KB ? c : KB - 1;
#pragma unroll
      for (int r = 0; r < 4; ++r)
        n_dh[cu][r] = p.acc_dH ? *at32(p.dH, mr[r] * (uint32_t)p.ldh + (uint32_t)(16 * cc + i)) : 0.f;
    }
  };
  issue((int)blockIdx.x);
  RSX_STAMP(1, blockIdx.x == 0);
  for (int blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
    const bool stamp_ = blockIdx.x == 0 && blk == (int)gridDim.x;        // (profiling build) the workgroup's SECOND block
    RSX_STAMP(2, stamp_);
    const int jrow = blk * 64 + 16 * rt + i;       // position in the (possibly compacted) row list
    const bool mok = jrow < Mv;
    const int mci = n_mci;
    const size_t mc = (size_t)mci;
    int mrow[4];                                   // original indices of this lane's C-layout rows 16*rt + 4*kq + r
#pragma unroll
    for (int r = 0; r < 4; ++r) mrow[r] = __shfl(mci, 4 * kq + r);
    // ---- S1/S2: g2 (A layout, registers; both halves), the h / q / g2 tiles of the block (half 0) -----------------
    const float dz = mok ? n_dz : 0.f;
    float a2N[NT2][4], a1N[NH1][4], dh_old[CH][4];
#pragma unroll
    for (int kb = 0; kb < NT2; ++kb)
#pragma unroll
      for (int t = 0; t < 4; ++t) a2N[kb][t] = n_a2[kb][t];
#pragma unroll
    for (int u = 0; u < NH1; ++u)
#pragma unroll
      for (int r = 0; r < 4; ++r) a1N[u][r] = n_a1[u][r];
#pragma unroll
    for (int cu = 0; cu < CH; ++cu)
#pragma unroll
      for (int r = 0; r < 4; ++r) dh_old[cu][r] = n_dh[cu][r];
    float g2[NT2][4];
#pragma unroll
    for (int kb = 0; kb < NT2; ++kb)
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int n = 16 * kb + 4 * kq + t;
        const bool ok = mok && n < p.N2;
        const float av = a2N[kb][t] * (ok ? 1.f : 0.f);
        const float mul = d2.mode == 0 ? 1.f : (ok ? drop_mul(d2, p.mask2, mc * p.N2 + n) : 0.f);
        const float gv = av > 0.f ? dz * sw2[n] * mul : 0.f;
        g2[kb][t] = gv;
        if (hf == 0) {
          sA2[n * LDR + 16 * rt + i] = av * mul * dz;
          sG2[n * LDR + 16 * rt + i] = gv;
        }
      }
    if constexpr (VEC) {                           // a1 after dropout -> its transposed tile (this wave's rows and column tiles)
#pragma unroll
      for (int u = 0; u < NH1; ++u) {
        if (nt0 + u < NT1) {
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            const int n = 16 * (nt0 + u) + 4 * kq + t;
            const bool ok = mok && n < p.N1;
            const float av = a1N[u][t] * (ok ? 1.f : 0.f);
            const float mul = d1.mode == 0 ? 1.f : (ok ? drop_mul(d1, p.mask1, mc * p.N1 + n) : 0.f);
            sA1[n * LDR + 16 * rt + i] = av * mul;
          }
        }
      }
    }
    if (hf == 0) {
      if (kq == 0) db2acc += dz;
#pragma unroll
      for (int c = 0; c < KB; ++c) {
        const float4 hr = n_h[c], qr = n_q[c];
        const float4 hv = mok ? hr : F4Z;
        const float4 qv = mok ? qr : F4Z;
        const int cc = 16 * c + 4 * kq, rr = 16 * rt + i;
        sH[(cc + 0) * LDR + rr] = hv.x; sH[(cc + 1) * LDR + rr] = hv.y; sH[(cc + 2) * LDR + rr] = hv.z; sH[(cc + 3) * LDR + rr] = hv.w;
        sQ[(cc + 0) * LDR + rr] = qv.x; sQ[(cc + 1) * LDR + rr] = qv.y; sQ[(cc + 2) * LDR + rr] = qv.z; sQ[(cc + 3) * LDR + rr] = qv.w;
      }
    }
    RSX_STAMP(3, stamp_ && g2[0][0] != 12345.f);
    // ---- S3: dg1 = g2 . W1^T for this half's column tiles (C layout: rows 4*kq + r, column n1 = 16*nt + i) ----------
    f32x4 dg1[NH1];
#pragma unroll
    for (int u = 0; u < NH1; ++u) dg1[u] = zf;
#pragma unroll
    for (int kb = 0; kb < NT2; ++kb)
#pragma unroll
      for (int u = 0; u < NH1; ++u) {
        if (nt0 + u < NT1) {                           // wave-uniform
          const float4 b = *reinterpret_cast<const float4*>(sW1 + (16 * (nt0 + u) + i) * LD2 + 16 * kb + 4 * kq);
          dg1[u] = att_mfma(g2[kb][0], b.x, dg1[u]);
          dg1[u] = att_mfma(g2[kb][1], b.y, dg1[u]);
          dg1[u] = att_mfma(g2[kb][2], b.z, dg1[u]);
          dg1[u] = att_mfma(g2[kb][3], b.w, dg1[u]);
        }
      }
    RSX_STAMP(4, stamp_ && dg1[0][0] != 12345.f);
    // ---- S4: g1 = dg1 * drop1 * (a1 > 0); tiles g1^T / a1d^T --------------------------------------------------------
#pragma unroll
    for (int u = 0; u < NH1; ++u) {
      if (nt0 + u < NT1) {
        const int n = 16 * (nt0 + u) + i;
        float gq[4], aq[4];
        if constexpr (VEC) {
          // (a1 * drop1)[rows 4*kq + r][n] back from the tile this wave wrote above (zero where dropped, relu'd away or past
          // the end): positive <=> kept and a1 > 0
          const float4 ad = *reinterpret_cast<const float4*>(sA1 + n * LDR + 16 * rt + 4 * kq);
          const float mulc = d1.mode == 0 ? 1.f : d1.inv_keep;
          const float adr[4] = {ad.x, ad.y, ad.z, ad.w};
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float gv = adr[r] > 0.f ? dg1[u][r] * mulc : 0.f;
            db0acc[u] += gv;
            gq[r] = gv;
          }
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int row = 16 * rt + 4 * kq + r;
            const size_t mm = (size_t)mrow[r];
            const bool ok = blk * 64 + row < Mv && n < p.N1;
            const float av = a1N[u][r] * (ok ? 1.f : 0.f);
            const float mul = d1.mode == 0 ? 1.f : (ok ? drop_mul(d1, p.mask1, mm * p.N1 + n) : 0.f);
            const float gv = av > 0.f ? dg1[u][r] * mul : 0.f;
            db0acc[u] += gv;
            gq[r] = gv;
            aq[r] = av * mul;
          }
          *reinterpret_cast<float4*>(sA1 + n * LDR + 16 * rt + 4 * kq) = make_float4(aq[0], aq[1], aq[2], aq[3]);
        }
        *reinterpret_cast<float4*>(sG1 + n * LDR + 16 * rt + 4 * kq) = make_float4(gq[0], gq[1], gq[2], gq[3]);
      }
    }
    RSX_STAMP(5, stamp_);
    __syncthreads();
    RSX_STAMP(6, stamp_);
    // ---- S5: dx = g1 . W0^T for this half's column blocks of the 4 segments -> dH, per-row dq ---------------------------
#pragma unroll
    for (int cu = 0; cu < CH; ++cu) {
      const int c = hf * CH + cu;                      // wave-uniform
      if (c < KB) {
        f32x4 dx[4] = {zf, zf, zf, zf};
#pragma unroll
        for (int kb = 0; kb < NT1; ++kb) {
          const int nn = 16 * kb + 4 * kq, rr = 16 * rt + i;
          const float a0 = sG1[(nn + 0) * LDR + rr], a1_ = sG1[(nn + 1) * LDR + rr], a2_ = sG1[(nn + 2) * LDR + rr],
                      a3 = sG1[(nn + 3) * LDR + rr];
#pragma unroll
          for (int sg = 0; sg < 4; ++sg) {
            const float4 b = *reinterpret_cast<const float4*>(sW0 + (16 * (sg * KB + c) + i) * LD1 + 16 * kb + 4 * kq);
            dx[sg] = att_mfma(a0, b.x, dx[sg]);
            dx[sg] = att_mfma(a1_, b.y, dx[sg]);
            dx[sg] = att_mfma(a2_, b.z, dx[sg]);
            dx[sg] = att_mfma(a3, b.w, dx[sg]);
          }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = 16 * rt + 4 * kq + r, col = 16 * c + i;
          if (blk * 64 + row < Mv) {
            const float hv = sH[col * LDR + row], qv = sQ[col * LDR + row];
            const float dh = (dx[0][r] + dx[2][r] * qv) + dx[3][r];
            *at32(p.dH, (uint32_t)mrow[r] * (uint32_t)p.ldh + (uint32_t)col) = p.acc_dH ? dh_old[cu][r] + dh : dh;
            *at32(p.dqr, (uint32_t)mrow[r] * (uint32_t)K + (uint32_t)col) = (dx[1][r] + dx[2][r] * hv) - dx[3][r];
          }
        }
      }
    }
    RSX_STAMP(7, stamp_);
    issue(blk + (int)gridDim.x);                       // the NEXT block's operands, in flight during S6
    // ---- S6: weight gradients over the block's 64 rows (k-step (kb, t) <-> row 16*kb + 4*kq + t) --------------------
    if (wave < NT1 + 2) {                              // dW1 row tile `wave`; waves NT1 / NT1 + 1: the ones rows (db1 / dW2)
      float4 av[4];
#pragma unroll
      for (int kb = 0; kb < 4; ++kb)
        av[kb] = wave < NT1 ? *reinterpret_cast<const float4*>(sA1 + (16 * wave + i) * LDR + 16 * kb + 4 * kq)
                            : make_float4(1.f, 1.f, 1.f, 1.f);
      const float* __restrict__ sB = wave == NT1 + 1 ? sA2 : sG2;
#pragma unroll
      for (int jt = 0; jt < NT2; ++jt) {
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
          const float4 bv = *reinterpret_cast<const float4*>(sB + (16 * jt + i) * LDR + 16 * kb + 4 * kq);
          accW1[jt] = att_mfma(av[kb].x, bv.x, accW1[jt]);
          accW1[jt] = att_mfma(av[kb].y, bv.y, accW1[jt]);
          accW1[jt] = att_mfma(av[kb].z, bv.z, accW1[jt]);
          accW1[jt] = att_mfma(av[kb].w, bv.w, accW1[jt]);
        }
      }
    }
    if (wave < D::MT0) {                               // dW0 row tile `wave`: input features 16*wave .. (segment wave / KB)
      const int seg = wave / KB, col = 16 * (wave - seg * KB) + i;
      float4 av[4];
#pragma unroll
      for (int kb = 0; kb < 4; ++kb) {
        const float4 hv = *reinterpret_cast<const float4*>(sH + col * LDR + 16 * kb + 4 * kq);
        const float4 qv = *reinterpret_cast<const float4*>(sQ + col * LDR + 16 * kb + 4 * kq);
        av[kb] = seg == 0 ? hv
                 : seg == 1 ? qv
                 : seg == 2 ? make_float4(hv.x * qv.x, hv.y * qv.y, hv.z * qv.z, hv.w * qv.w)
                            : make_float4(hv.x - qv.x, hv.y - qv.y, hv.z - qv.z, hv.w - qv.w);
      }
#pragma unroll
      for (int jt = 0; jt < NT1; ++jt) {
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
          const float4 bv = *reinterpret_cast<const float4*>(sG1 + (16 * jt + i) * LDR + 16 * kb + 4 * kq);
          accW0[jt] = att_mfma(av[kb].x, bv.x, accW0[jt]);
          accW0[jt] = att_mfma(av[kb].y, bv.y, accW0[jt]);
          accW0[jt] = att_mfma(av[kb].z, bv.z, accW0[jt]);
          accW0[jt] = att_mfma(av[kb].w, bv.w, accW0[jt]);
        }
      }
    }
    RSX_STAMP(8, stamp_ && accW0[0][0] != 12345.f);
    __syncthreads();                                   // the tiles are rewritten by the next block
    RSX_STAMP(9, stamp_);
  }
  RSX_STAMP(10, blockIdx.x == 0);

  // ---- this workgroup's partial of every weight gradient: [dW0 | db0 | dW1 | db1 | dW2 | db2] --------------------------
  float* out = p.part + (size_t)blockIdx.x * ((size_t)K4 * p.N1 + p.N1 + (size_t)p.N1 * p.N2 + 2 * p.N2 + 1);
  float* o_dW0 = out;
  float* o_db0 = o_dW0 + (size_t)K4 * p.N1;
  float* o_dW1 = o_db0 + p.N1;
  float* o_db1 = o_dW1 + (size_t)p.N1 * p.N2;
  float* o_dW2 = o_db1 + p.N2;
  float* o_db2 = o_dW2 + p.N2;
  if (wave < D::MT0) {
#pragma unroll
    for (int jt = 0; jt < NT1; ++jt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int kin = 16 * wave + 4 * kq + r, n = 16 * jt + i;
        if (n < p.N1) o_dW0[(size_t)kin * p.N1 + n] = accW0[jt][r];
      }
  }
  if (wave < NT1) {
#pragma unroll
    for (int jt = 0; jt < NT2; ++jt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int k1 = 16 * wave + 4 * kq + r, n = 16 * jt + i;
        if (k1 < p.N1 && n < p.N2) o_dW1[(size_t)k1 * p.N2 + n] = accW1[jt][r];
      }
  }
  if ((wave == NT1 || wave == NT1 + 1) && kq == 0) {   // db1 / dW2: row 0 of the ones-row tiles
    float* o_ = wave == NT1 ? o_db1 : o_dW2;
#pragma unroll
    for (int jt = 0; jt < NT2; ++jt) {
      const int n = 16 * jt + i;
      if (n < p.N2) o_[n] = accW1[jt][0];
    }
  }
  // bias-like sums: per-lane partials -> fixed-order sums over lanes and waves through LDS (the tiles are idle now)
  float* red = sG1;                                    // [8 waves][64 lanes][NR], spans the g1 / a1 / g2 tiles
  constexpr int NR = NH1 + 1;
  static_assert(8 * 64 * NR <= 68 * (2 * N1P + N2P + 2 * K), "reduction scratch must fit the block tiles");
  {
    float* r_ = red + (wave * 64 + lane) * NR;
#pragma unroll
    for (int a = 0; a < NH1; ++a) r_[a] = db0acc[a];
    r_[NH1] = db2acc;
  }
  __syncthreads();
  // db0[n1 = 16*nt + i]: nt belongs to half nt / NH1; sum over that half's 4 row-tile waves and the 4 kq lanes of column i
  for (int n = tid; n < p.N1; n += 512) {
    const int nt = n >> 4, ii = n & 15, h_ = nt / NH1, u = nt - h_ * NH1;
    float s_ = 0.f;
    for (int w = 0; w < 4; ++w)
      for (int k4 = 0; k4 < 4; ++k4) s_ += red[((h_ * 4 + w) * 64 + 16 * k4 + ii) * NR + u];
    o_db0[n] = s_;
  }
  if (tid == 0) {
    float s_ = 0.f;
    for (int w = 0; w < 4; ++w)
      for (int ii = 0; ii < 16; ++ii) s_ += red[(w * 64 + ii) * NR + NH1];
    o_db2[0] = s_;
  }
  RSX_STAMP(11, blockIdx.x == 0);
  RSX_STAMP_MAX(12, true);
}

// Two small jobs that only need the backward kernel's outputs, as ONE launch of 1024-thread workgroups:
//   * grads[j] = sum over the G workgroup partials: 16 waves x 64 elements, wave w adds the contiguous partial range
//     [w*per, (w+1)*per) (8 loads in flight), the 16 sub-sums are then added in ascending wave order -- a fixed association
//     with 16x shorter dependent chains than one thread walking all G partials;
//   * dq[b, c] = sum_p dqr[b*P + p, c] (+ add[b, c]): 256 threads per example = (256/K) position groups x K columns; group g
//     adds positions g, g+G', ... (ascending), the groups are then added in ascending order.  valid (nullable, [B*P] ids):
//     with a row list only the positions whose id is > 0 were written to dqr -- the others are selected away (never
//     multiplied: they may hold anything).  dq rows ld_dq floats apart; add (nullable, rows ld_add apart): a second gradient of
//     the same query added on the way out (din/din.py:131: the target item embedding also feeds the final MLP directly).
// workgroups [0, nr) reduce 64 weight-gradient elements each, the rest take 4 examples each.
__device__ __forceinline__ void attn_finish_block(float (*sub)[64], const float* __restrict__ part, int G, int n,
                                                  float* __restrict__ grads, int nr, const float* __restrict__ dqr,
                                                  float* __restrict__ dq, int B, int P, int K,
                                                  const int32_t* __restrict__ valid, int ld_dq, const float* __restrict__ add,
                                                  int ld_add) {
  if ((int)blockIdx.x < nr) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int j = blockIdx.x * 64 + lane;
    const int per = (G + 15) / 16;
    const int g0 = w * per, g1 = g0 + per < G ? g0 + per : G;
    float s = 0.f;
    if (j < n) {
      int g = g0;
      for (; g + 8 <= g1; g += 8) {
        float t[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) t[u] = part[(size_t)(g + u) * n + j];
#pragma unroll
        for (int u = 0; u < 8; ++u) s += t[u];
      }
      for (; g < g1; ++g) s += part[(size_t)g * n + j];
    }
    sub[w][lane] = s;
    __syncthreads();
    if (w == 0 && j < n) {
      float t = sub[0][lane];
#pragma unroll
      for (int k = 1; k < 16; ++k) t += sub[k][lane];
      grads[j] = t;
    }
    return;
  }
  float* sb = &sub[0][0] + (threadIdx.x >> 8) * 256;           // this example's 256 slots
  const int lt = threadIdx.x & 255;
  const int b = ((int)blockIdx.x - nr) * 4 + (threadIdx.x >> 8);
  const int c = lt % K, g = lt / K, ng = 256 / K;
  float s = 0.f;
  if (b < B) {
    const float* src = dqr + (size_t)b * P * K + c;
    const int32_t* vid = valid ? valid + (size_t)b * P : nullptr;
    int pp = g;
    for (; pp + 3 * ng < P; pp += 4 * ng) {
      float t0 = src[(size_t)pp * K], t1 = src[(size_t)(pp + ng) * K], t2 = src[(size_t)(pp + 2 * ng) * K],
            t3 = src[(size_t)(pp + 3 * ng) * K];
      if (vid) {
        const int v0 = vid[pp], v1 = vid[pp + ng], v2 = vid[pp + 2 * ng], v3 = vid[pp + 3 * ng];
        t0 = v0 > 0 ? t0 : 0.f; t1 = v1 > 0 ? t1 : 0.f; t2 = v2 > 0 ? t2 : 0.f; t3 = v3 > 0 ? t3 : 0.f;
      }
      s += t0; s += t1; s += t2; s += t3;
    }
    for (; pp < P; pp += ng) {
      const float t = src[(size_t)pp * K];
      s += (vid == nullptr || vid[pp] > 0) ? t : 0.f;
    }
  }
  sb[lt] = s;
  __syncthreads();
  if (b < B && g == 0) {
    float t = sb[c];
    for (int k = 1; k < ng; ++k) t += sb[k * K + c];
    if (add != nullptr) t = add[(size_t)b * ld_add + c] + t;
    dq[(size_t)b * ld_dq + c] = t;
  }
}
__global__ __launch_bounds__(1024) void din_attn_finish_k(const float* __restrict__ part, int G, int n, float* __restrict__ grads,
                                                          int nr, const float* __restrict__ dqr, float* __restrict__ dq, int B,
                                                          int P, int K, const int32_t* __restrict__ valid, int ld_dq,
                                                          const float* __restrict__ add, int ld_add) {
  __shared__ float sub[16][64];
  attn_finish_block(sub, part, G, n, grads, nr, dqr, dq, B, P, K, valid, ld_dq, add, ld_add);
}
// the finish of BOTH attention blocks of din.py in one launch (grid.y = block)
struct AttnFinishSet { const float* part; float* grads; const float* dqr; float* dq; const int32_t* valid; const float* add; };
__global__ __launch_bounds__(1024) void din_attn_finish_pair_k(const AttnFinishSet s0, const AttnFinishSet s1, int G, int n, int nr,
                                                               int B, int P, int K, int ld_dq, int ld_add) {
  __shared__ float sub[16][64];
  if (blockIdx.y == 0) attn_finish_block(sub, s0.part, G, n, s0.grads, nr, s0.dqr, s0.dq, B, P, K, s0.valid, ld_dq, s0.add, ld_add);
  else attn_finish_block(sub, s1.part, G, n, s1.grads, nr, s1.dqr, s1.dq, B, P, K, s1.valid, ld_dq, s1.add, ld_add);
}

// ---- row list of the history positions that are not padding (id > 0), din/din.py:118-124 -------------------------------
// Two small launches over 1024-position tiles: (1) valid positions per tile; (2) every tile adds the counts of the tiles
// before it (<= a few hundred ints), compacts its own positions with ballots in ascending order and -- if asked -- zeroes w
// at the padded positions (the pooling multiplies w by the mask).
__global__ __launch_bounds__(256) void din_tile_counts_k(const int32_t* __restrict__ ids, int M, int32_t* __restrict__ tcnt) {
  __shared__ int wsum[4];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  int c = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int e = blockIdx.x * 1024 + 256 * k + tid;
    c += (e < M && ids[e] > 0) ? 1 : 0;
  }
#pragma unroll
  for (int m = 1; m < 64; m <<= 1) c += __shfl_xor(c, m);
  if (lane == 0) wsum[wv] = c;
  __syncthreads();
  if (tid == 0) tcnt[blockIdx.x] = ((wsum[0] + wsum[1]) + wsum[2]) + wsum[3];
}

__global__ __launch_bounds__(256) void din_valid_rows_k(const int32_t* __restrict__ ids, int M, int32_t* __restrict__ rows,
                                                        int32_t* __restrict__ count, const int32_t* __restrict__ tcnt,
                                                        float* __restrict__ w) {
  __shared__ int wsum[4];
  __shared__ int sbase;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int t0 = blockIdx.x * 1024;
  int before = 0;
  for (int t = tid; t < (int)blockIdx.x; t += 256) before += tcnt[t];
#pragma unroll
  for (int m = 1; m < 64; m <<= 1) before += __shfl_xor(before, m);
  if (lane == 0) wsum[wv] = before;
  __syncthreads();
  if (tid == 0) sbase = ((wsum[0] + wsum[1]) + wsum[2]) + wsum[3];
  __syncthreads();
  int base = sbase;
  // the tile in 4 passes of 256 consecutive positions: wave w of pass k covers positions t0 + 256 k + 64 w ..
  for (int k = 0; k < 4; ++k) {
    const int e = t0 + 256 * k + tid;
    const bool ok = e < M && ids[e] > 0;
    if (e < M && !ok && w != nullptr) w[e] = 0.f;
    const uint64_t bal = __ballot(ok);
    __syncthreads();
    if (lane == 0) wsum[wv] = __popcll(bal);
    __syncthreads();
    int off = base;
    for (int ww = 0; ww < wv; ++ww) off += wsum[ww];
    if (ok) rows[off + __popcll(bal & ((1ull << lane) - 1ull))] = e;
    base += ((wsum[0] + wsum[1]) + wsum[2]) + wsum[3];
  }
  if (blockIdx.x == gridDim.x - 1 && tid == 0) count[0] = base;
}

// ---- both histories' row lists AND the sort keys of both id tables in two launches (fused TRAIN step of din.py) ---------------
// launch 1, grid (tiles, 2): valid positions per 1024-position tile of history y; the y = 0 blocks also write the tile's sort
// keys (entry B + position: the history ids with padding mapped to the tables' dummy rows, rsx_din_keys) and block (0, 0) the B
// target entries' keys.  launch 2, grid (tiles, 2): din_valid_rows_k's body per history.
struct DinPrep {
  const int32_t* hist[2];
  int32_t* rows[2];
  int32_t* count[2];          // count[h][0] = valid rows, count[h] + 1: per-tile scratch
  float* w[2];                // nullable: zeroed at the padded positions
  const int32_t* i_id;
  const int32_t* i_cate;
  int32_t* keys2;             // [B + M, 2], or field-major [2, kt_stride] when kt_stride > 0
  int B, M, dummy[2];
  int kt_stride;
  const int64_t* labels_i64;  // nullable: labels_f32[e] = (float)labels_i64[e] (the model_fn's tf.cast, din/din.py:146)
  float* labels_f32;
};
// RID: the launch carries row gathers of the same step as extra workgroups (gather_rows_device.h): x >= n_tiles, two per x
template <bool RID>
__global__ __launch_bounds__(256) void din_prep_counts_k(const DinPrep p, const GatherJobs gj, const int n_tiles, const uint32_t n_gather) {
  if (RID && (int)blockIdx.x >= n_tiles) {
    const uint32_t r = (blockIdx.x - n_tiles) * 2 + blockIdx.y;
    if (r < n_gather) RSX_GATHER_ROWS_BLOCK(gj, r);
    return;
  }
  __shared__ int wsum[4];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, h = blockIdx.y;
  const int32_t* ids = h ? p.hist[1] : p.hist[0];
  int32_t* tcnt = (h ? p.count[1] : p.count[0]) + 1;
  int c = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int e = blockIdx.x * 1024 + 256 * k + tid;
    const int v = e < p.M ? ids[e] : 0;
    c += v > 0 ? 1 : 0;
    if (h == 0 && e < p.M) {
      const int vc = p.hist[1][e];
      const int k0 = v > 0 ? v : p.dummy[0], k1 = vc > 0 ? vc : p.dummy[1];
      if (p.kt_stride > 0) {
        p.keys2[(size_t)p.B + e] = k0;
        p.keys2[(size_t)p.kt_stride + p.B + e] = k1;
      } else {
        reinterpret_cast<int2*>(p.keys2)[(size_t)p.B + e] = make_int2(k0, k1);
      }
    }
  }
  // the B target entries' keys (+ the labels' cast), a slice per tile workgroup of history 0 (round 4: block (0, 0) used to walk
  // all of them in B / 256 dependent trips -- 4 us longer than every other workgroup of the launch at batch 1 024)
  if (h == 0) {
    const int per = (p.B + n_tiles - 1) / n_tiles;
    const int e0 = (int)blockIdx.x * per, e1 = e0 + per < p.B ? e0 + per : p.B;
    for (int e = e0 + tid; e < e1; e += 256) {
      if (p.kt_stride > 0) {
        p.keys2[e] = p.i_id[e];
        p.keys2[(size_t)p.kt_stride + e] = p.i_cate[e];
      } else {
        reinterpret_cast<int2*>(p.keys2)[e] = make_int2(p.i_id[e], p.i_cate[e]);
      }
      if (p.labels_i64 != nullptr) p.labels_f32[e] = (float)p.labels_i64[e];
    }
  }
#pragma unroll
  for (int m = 1; m < 64; m <<= 1) c += __shfl_xor(c, m);
  if (lane == 0) wsum[wv] = c;
  __syncthreads();
  if (tid == 0) tcnt[blockIdx.x] = ((wsum[0] + wsum[1]) + wsum[2]) + wsum[3];
}
template <bool RID>
__global__ __launch_bounds__(256) void din_prep_rows_k(const DinPrep p, const GatherJobs gj, const int n_tiles, const uint32_t n_gather) {
  if (RID && (int)blockIdx.x >= n_tiles) {
    const uint32_t r = (blockIdx.x - n_tiles) * 2 + blockIdx.y;
    if (r < n_gather) RSX_GATHER_ROWS_BLOCK(gj, r);
    return;
  }
  __shared__ int wsum[4];
  __shared__ int sbase;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, h = blockIdx.y;
  const int32_t* ids = h ? p.hist[1] : p.hist[0];
  int32_t* rows = h ? p.rows[1] : p.rows[0];
  int32_t* count = h ? p.count[1] : p.count[0];
  const int32_t* tcnt = count + 1;
  float* w = h ? p.w[1] : p.w[0];
  const int M = p.M, t0 = blockIdx.x * 1024;
  int before = 0;
  for (int t = tid; t < (int)blockIdx.x; t += 256) before += tcnt[t];
#pragma unroll
  for (int m = 1; m < 64; m <<= 1) before += __shfl_xor(before, m);
  if (lane == 0) wsum[wv] = before;
  __syncthreads();
  if (tid == 0) sbase = ((wsum[0] + wsum[1]) + wsum[2]) + wsum[3];
  __syncthreads();
  int base = sbase;
  for (int k = 0; k < 4; ++k) {
    const int e = t0 + 256 * k + tid;
    const bool ok = e < M && ids[e] > 0;
    if (e < M && !ok && w != nullptr) w[e] = 0.f;
    const uint64_t bal = __ballot(ok);
    __syncthreads();
    if (lane == 0) wsum[wv] = __popcll(bal);
    __syncthreads();
    int off = base;
    for (int ww = 0; ww < wv; ++ww) off += wsum[ww];
    if (ok) rows[off + __popcll(bal & ((1ull << lane) - 1ull))] = e;
    base += ((wsum[0] + wsum[1]) + wsum[2]) + wsum[3];
  }
  if ((int)blockIdx.x == n_tiles - 1 && tid == 0) count[0] = base;
}

extern "C" int rsx_din_prepare2_gather(const int32_t* i_id, const int32_t* i_cate, const int32_t* hist_i, const int32_t* hist_c,
                                       int B, int P, int dummy_item_row, int dummy_cate_row, int32_t* keys2, int keys_field_stride,
                                       int32_t* rows_i, int32_t* count_i, float* w_i, int32_t* rows_c, int32_t* count_c, float* w_c,
                                       const int64_t* labels_i64, float* labels_f32, const rsx_gather_job* jobs_h, int njobs,
                                       int njobs_first, rsx_stream_t stream) {
  if (B <= 0 || P <= 0) return (B == 0 && P > 0) ? RSX_OK : RSX_EINVAL;
  if (!i_id || !i_cate || !hist_i || !hist_c || !keys2 || !rows_i || !count_i || !rows_c || !count_c) return RSX_EINVAL;
  if ((labels_i64 != nullptr) != (labels_f32 != nullptr)) return RSX_EINVAL;
  if (njobs < 0 || njobs > RSX_GATHER_MAX_JOBS || njobs_first < 0 || njobs_first > njobs || (njobs > 0 && !jobs_h)) return RSX_EINVAL;
  const long long M = (long long)B * P;
  if (M > (1ll << 24)) return RSX_EUNSUPPORTED;
  if (keys_field_stride != 0 && keys_field_stride < B + M) return RSX_EINVAL;
  DinPrep p{{hist_i, hist_c}, {rows_i, rows_c}, {count_i, count_c}, {w_i, w_c}, i_id, i_cate, keys2, B, (int)M,
            {dummy_item_row, dummy_cate_row}, keys_field_stride, labels_i64, labels_f32};
  const int nt = (int)((M + 1023) / 1024);
  GatherJobs g1{}, g2{};
  uint32_t n1 = 0, n2 = 0;
  if (njobs_first > 0) {
    const int rc = gather_jobs_pack(jobs_h, 0, njobs_first, g1, &n1);
    if (rc != RSX_OK) return rc;
  }
  if (njobs - njobs_first > 0) {
    const int rc = gather_jobs_pack(jobs_h, njobs_first, njobs - njobs_first, g2, &n2);
    if (rc != RSX_OK) return rc;
  }
  if (n1 > 0) RSX_LAUNCH(din_prep_counts_k<true>, dim3(nt + (n1 + 1) / 2, 2), dim3(256), 0, rsx_s(stream), p, g1, nt, n1);
  else RSX_LAUNCH(din_prep_counts_k<false>, dim3(nt, 2), dim3(256), 0, rsx_s(stream), p, g1, nt, 0u);
  if (n2 > 0) RSX_LAUNCH(din_prep_rows_k<true>, dim3(nt + (n2 + 1) / 2, 2), dim3(256), 0, rsx_s(stream), p, g2, nt, n2);
  else RSX_LAUNCH(din_prep_rows_k<false>, dim3(nt, 2), dim3(256), 0, rsx_s(stream), p, g2, nt, 0u);
  RSX_CHECK_LAUNCH();
  return RSX_OK;
}

extern "C" int rsx_din_prepare2(const int32_t* i_id, const int32_t* i_cate, const int32_t* hist_i, const int32_t* hist_c, int B,
                                int P, int dummy_item_row, int dummy_cate_row, int32_t* keys2, int keys_field_stride,
                                int32_t* rows_i, int32_t* count_i, float* w_i, int32_t* rows_c, int32_t* count_c, float* w_c,
                                const int64_t* labels_i64, float* labels_f32, rsx_stream_t stream) {
  return rsx_din_prepare2_gather(i_id, i_cate, hist_i, hist_c, B, P, dummy_item_row, dummy_cate_row, keys2, keys_field_stride, rows_i,
                                 count_i, w_i, rows_c, count_c, w_c, labels_i64, labels_f32, nullptr, 0, 0, stream);
}

extern "C" int rsx_din_prepare(const int32_t* i_id, const int32_t* i_cate, const int32_t* hist_i, const int32_t* hist_c, int B,
                               int P, int dummy_item_row, int dummy_cate_row, int32_t* keys2, int32_t* rows_i,
                               int32_t* count_i, float* w_i, int32_t* rows_c, int32_t* count_c, float* w_c,
                               rsx_stream_t stream) {
  return rsx_din_prepare2(i_id, i_cate, hist_i, hist_c, B, P, dummy_item_row, dummy_cate_row, keys2, 0, rows_i, count_i, w_i,
                          rows_c, count_c, w_c, nullptr, nullptr, stream);
}

extern "C" int rsx_din_valid_rows(const int32_t* ids, int B, int P, int32_t* rows, int32_t* count, float* w_zero_padded,
                                  rsx_stream_t stream) {
  if (B < 0 || P <= 0) return RSX_EINVAL;
  if (!ids || !rows || !count) return RSX_EINVAL;
  const long long M = (long long)B * P;
  if (M > (1ll << 24)) return RSX_EUNSUPPORTED;
  const int nt = M == 0 ? 1 : (int)((M + 1023) / 1024);
  RSX_LAUNCH(din_tile_counts_k, dim3(nt), dim3(256), 0, rsx_s(stream), ids, (int)M, count + 1);
  RSX_LAUNCH(din_valid_rows_k, dim3(nt), dim3(256), 0, rsx_s(stream), ids, (int)M, rows, count, count + 1,
                     w_zero_padded);
  RSX_CHECK_LAUNCH();
  return RSX_OK;
}

static inline int attn_bwd_groups(int M) {
  const int nblk = (M + 63) / 64;
  return nblk < 256 ? nblk : 256;
}
static inline size_t attn_npart(int K, int N1, int N2) { return (size_t)4 * K * N1 + N1 + (size_t)N1 * N2 + 2 * N2 + 1; }

extern "C" size_t rsx_din_attn_bwd_workspace_floats(int B, int P, int K, int N1, int N2) {
  const size_t M = (size_t)B * P;
  return M * K + (size_t)attn_bwd_groups((int)M) * attn_npart(K, N1, N2);
}

template <int KB, int NT1, int NT2, bool VEC>
static int launch_attn_bwd(const AttnBwdArgs& p, int G, hipStream_t st) {
  using D = AttnDims<KB, NT1, NT2>;
  const size_t fl = (size_t)D::K4 * D::LD1 + (size_t)D::N1P * D::LD2 + D::N2P + (size_t)68 * (2 * D::N1P + 2 * D::N2P + 2 * D::K);
  static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(din_attn_bwd_k<KB, NT1, NT2, VEC>),
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (attr != hipSuccess || fl * sizeof(float) > 160 * 1024) return RSX_EUNSUPPORTED;
  RSX_LAUNCH((din_attn_bwd_k<KB, NT1, NT2, VEC>), dim3(G), dim3(512), fl * sizeof(float), st, p);
  return RSX_OK;
}

extern "C" int rsx_din_attn_bwd_ld(const float*, const float*, const float*, const float*, const float*, const float*,
                                   const float*, const float*, float*, float*, float*, float*, const float*, const float*,
                                   const uint32_t*, uint32_t, int, float, int, const int32_t*, const int32_t*, const int32_t*, int,
                                   int, int, int, int, int, int, const float*, int, rsx_stream_t);
extern "C" int rsx_din_attn_bwd(const float* H, const float* q, const float* W0, const float* W1, const float* W2,
                                const float* a1, const float* a2, const float* dw, float* dH, float* dq, float* grads,
                                float* workspace, const float* mask1, const float* mask2, const uint32_t* rng_step,
                                uint32_t seed, int layer0, float dropout_rate, int accumulate_dH, const int32_t* rows,
                                const int32_t* count, const int32_t* ids, int B, int P, int K, int N1, int N2,
                                rsx_stream_t stream) {
  return rsx_din_attn_bwd_ld(H, q, W0, W1, W2, a1, a2, dw, dH, dq, grads, workspace, mask1, mask2, rng_step, seed, layer0,
                             dropout_rate, accumulate_dH, rows, count, ids, B, P, K, N1, N2, K, K, nullptr, 0, stream);
}

static int attn_bwd_impl(const float* H, const float* q, const float* W0, const float* W1, const float* W2, const float* a1,
                         const float* a2, const float* dw, float* dH, float* dq, float* grads, float* workspace,
                         const float* mask1, const float* mask2, const uint32_t* rng_step, uint32_t seed, int layer0,
                         float dropout_rate, int accumulate_dH, const int32_t* rows, const int32_t* count, const int32_t* ids,
                         int B, int P, int K, int N1, int N2, int ld_dH, int ld_dq, const float* dq_add, int ld_dq_add,
                         bool finish, rsx_stream_t stream) {
  if (B < 0 || P <= 0 || K <= 0 || N1 <= 0 || N2 <= 0) return RSX_EINVAL;
  if (B == 0) return RSX_OK;
  if (!H || !q || !W0 || !W1 || !W2 || !a1 || !a2 || !dw || !dH || !dq || !grads || !workspace) return RSX_EINVAL;
  if (ld_dH < K || ld_dq < K || (dq_add != nullptr && ld_dq_add < K)) return RSX_EINVAL;
  if ((rows == nullptr) != (count == nullptr) || (rows != nullptr && ids == nullptr)) return RSX_EINVAL;
  if (dropout_rate < 0.f || dropout_rate >= 1.f) return RSX_EINVAL;
  if ((K != 16 && K != 32) || N1 > 80 || N2 > 48) return RSX_EUNSUPPORTED;
  {   // the backward kernel addresses its per-row arrays with 32-bit element offsets
    const size_t widest = (size_t)(N1 > ld_dH ? N1 : ld_dH);
    if ((size_t)B * P * widest >= (1ull << 30)) return RSX_EUNSUPPORTED;
  }
  const int M = B * P, G = attn_bwd_groups(M);
  AttnBwdArgs p{H, q, W0, W1, W2, a1, a2, dw, dH, workspace, workspace + (size_t)M * K, mask1, mask2, rng_step, seed,
                (uint32_t)layer0, dropout_rate, M, P, N1, N2, (M + 63) / 64, accumulate_dH != 0, ld_dH, rows, count};
  hipStream_t st = rsx_s(stream);
  const bool vec = N1 % 4 == 0 && N2 % 4 == 0 && ((uintptr_t)a1 & 15) == 0 && ((uintptr_t)a2 & 15) == 0;
  const int rc = K == 32 ? (vec ? launch_attn_bwd<2, 5, 3, true>(p, G, st) : launch_attn_bwd<2, 5, 3, false>(p, G, st))
                         : (vec ? launch_attn_bwd<1, 5, 3, true>(p, G, st) : launch_attn_bwd<1, 5, 3, false>(p, G, st));
  if (rc != RSX_OK) return rc;
  RSX_CHECK_LAUNCH();
  if (!finish) return RSX_OK;
  const int n = (int)attn_npart(K, N1, N2);
  const int nr = (n + 63) / 64;
  RSX_LAUNCH(din_attn_finish_k, dim3(nr + (B + 3) / 4), dim3(1024), 0, st, p.part, G, n, grads, nr, p.dqr, dq, B, P, K,
                     rows ? ids : nullptr, ld_dq, dq_add, ld_dq_add);
  RSX_CHECK_LAUNCH();
  return RSX_OK;
}

extern "C" int rsx_din_attn_bwd_ld(const float* H, const float* q, const float* W0, const float* W1, const float* W2,
                                   const float* a1, const float* a2, const float* dw, float* dH, float* dq, float* grads,
                                   float* workspace, const float* mask1, const float* mask2, const uint32_t* rng_step,
                                   uint32_t seed, int layer0, float dropout_rate, int accumulate_dH, const int32_t* rows,
                                   const int32_t* count, const int32_t* ids, int B, int P, int K, int N1, int N2, int ld_dH,
                                   int ld_dq, const float* dq_add, int ld_dq_add, rsx_stream_t stream) {
  return attn_bwd_impl(H, q, W0, W1, W2, a1, a2, dw, dH, dq, grads, workspace, mask1, mask2, rng_step, seed, layer0, dropout_rate,
                       accumulate_dH, rows, count, ids, B, P, K, N1, N2, ld_dH, ld_dq, dq_add, ld_dq_add, true, stream);
}

// The backward launch alone: the weight-gradient partials and the per-row query gradients stay in `workspace` until
// rsx_din_attn_finish_pair reduces them (din.py has two attention blocks per step: one finish launch instead of two).
extern "C" int rsx_din_attn_bwd_nofinish(const float* H, const float* q, const float* W0, const float* W1, const float* W2,
                                         const float* a1, const float* a2, const float* dw, float* dH, float* workspace,
                                         const float* mask1, const float* mask2, const uint32_t* rng_step, uint32_t seed,
                                         int layer0, float dropout_rate, int accumulate_dH, const int32_t* rows,
                                         const int32_t* count, const int32_t* ids, int B, int P, int K, int N1, int N2, int ld_dH,
                                         rsx_stream_t stream) {
  // (dq / grads are written by the finish: any non-null pointers pass the argument check)
  return attn_bwd_impl(H, q, W0, W1, W2, a1, a2, dw, dH, workspace, workspace, workspace, mask1, mask2, rng_step, seed, layer0,
                       dropout_rate, accumulate_dH, rows, count, ids, B, P, K, N1, N2, ld_dH, K, nullptr, 0, false, stream);
}

extern "C" int rsx_din_attn_finish_pair_defer(const float* workspace0, float* grads0, float* dq0, const int32_t* ids0,
                                              const float* dq_add0, const float* workspace1, float* grads1, float* dq1,
                                              const int32_t* ids1, const float* dq_add1, int B, int P, int K, int N1, int N2,
                                              int ld_dq, int ld_dq_add, rsx_vec_reduce_job* reduce_out, rsx_stream_t stream) {
  if (reduce_out != nullptr) reduce_out[0].n = reduce_out[1].n = 0;
  if (B < 0 || P <= 0 || K <= 0 || N1 <= 0 || N2 <= 0) return RSX_EINVAL;
  if (B == 0) return RSX_OK;
  if (!workspace0 || !grads0 || !dq0 || !workspace1 || !grads1 || !dq1 || ld_dq < K) return RSX_EINVAL;
  if ((dq_add0 != nullptr || dq_add1 != nullptr) && ld_dq_add < K) return RSX_EINVAL;
  const int M = B * P, G = attn_bwd_groups(M);
  const int n = (int)attn_npart(K, N1, N2);
  // reduce_out: the launch keeps the query-gradient sums only (nr = 0 reduce workgroups); the two reduces go back as jobs
  const int nr = reduce_out != nullptr ? 0 : (n + 63) / 64;
  // workspace layout of the backward launch: [M, K] per-row query gradients, then the G partials
  const AttnFinishSet s0{workspace0 + (size_t)M * K, grads0, workspace0, dq0, ids0, dq_add0};
  const AttnFinishSet s1{workspace1 + (size_t)M * K, grads1, workspace1, dq1, ids1, dq_add1};
  if (reduce_out != nullptr) {
    reduce_out[0] = rsx_vec_reduce_job{s0.part, grads0, G, n};
    reduce_out[1] = rsx_vec_reduce_job{s1.part, grads1, G, n};
  }
  RSX_LAUNCH(din_attn_finish_pair_k, dim3(nr + (B + 3) / 4, 2), dim3(1024), 0, rsx_s(stream), s0, s1, G, n, nr, B, P, K,
                     ld_dq, ld_dq_add);
  RSX_CHECK_LAUNCH();
  return RSX_OK;
}
extern "C" int rsx_din_attn_finish_pair(const float* workspace0, float* grads0, float* dq0, const int32_t* ids0,
                                        const float* dq_add0, const float* workspace1, float* grads1, float* dq1,
                                        const int32_t* ids1, const float* dq_add1, int B, int P, int K, int N1, int N2, int ld_dq,
                                        int ld_dq_add, rsx_stream_t stream) {
  return rsx_din_attn_finish_pair_defer(workspace0, grads0, dq0, ids0, dq_add0, workspace1, grads1, dq1, ids1, dq_add1, B, P, K, N1,
                                        N2, ld_dq, ld_dq_add, nullptr, stream);
}

// the deferred reduces as a launch of their own (grid.y = job)
__global__ __launch_bounds__(256) void vec_reduce_k(const rsx_vec_reduce_job a, const rsx_vec_reduce_job b) {
  __shared__ float sub[1024];
  if (blockIdx.y == 0) vec_reduce_block(a.part, a.G, a.n, a.out, blockIdx.x, sub);
  else vec_reduce_block(b.part, b.G, b.n, b.out, blockIdx.x, sub);
}
extern "C" int rsx_vec_reduce_run(const rsx_vec_reduce_job* jobs_h, int njobs, rsx_stream_t stream) {
  if (njobs == 0) return RSX_OK;
  if (!jobs_h || njobs < 0 || njobs > RSX_VEC_REDUCE_MAX_JOBS) return RSX_EINVAL;
  const rsx_vec_reduce_job a = jobs_h[0], b = jobs_h[njobs > 1 ? 1 : 0];
  int nmax = 0;
  for (int k = 0; k < njobs; ++k) {
    if (jobs_h[k].n < 0 || (jobs_h[k].n > 0 && (!jobs_h[k].part || !jobs_h[k].out || jobs_h[k].G <= 0))) return RSX_EINVAL;
    nmax = jobs_h[k].n > nmax ? jobs_h[k].n : nmax;
  }
  if (nmax == 0) return RSX_OK;
  RSX_LAUNCH(vec_reduce_k, dim3((nmax + 63) / 64, njobs), dim3(256), 0, rsx_s(stream), a, b);
  RSX_CHECK_LAUNCH();
  return RSX_OK;
}

#ifdef RSX_STAMPS
// profiling build only: the attention kernels' phase stamps (100 MHz wall clock ticks)
extern "C" int rsx_dbg_stamps_attn_zero() {
  static const unsigned long long z[64] = {0};
  return hipMemcpyToSymbol(HIP_SYMBOL(rsx_stamps_d), z, sizeof(z)) == hipSuccess ? RSX_OK : RSX_ELAUNCH;
}
extern "C" int rsx_dbg_stamps_attn(unsigned long long* out_h) {
  return hipMemcpyFromSymbol(out_h, HIP_SYMBOL(rsx_stamps_d), sizeof(unsigned long long) * 64) == hipSuccess ? RSX_OK : RSX_ELAUNCH;
}
#endif
