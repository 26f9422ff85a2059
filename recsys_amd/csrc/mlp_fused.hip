// A tower WITHOUT batch-norm (din/din.py:130-147: 'mlp_layer' = 3 x [dense(relu) -> dropout] -> dense(1) + the item bias ->
// sigmoid cross-entropy) as ONE launch for forward AND backward, plus one reduce launch for the weight gradients.
//
// Without batch-norm no statistic crosses the rows of a batch, so a 16-row tile can run the whole network -- forward, loss,
// backward -- on its own: the weights of every layer (96x100 + 100x52 + 52x20 + 20 floats = 63 KB for din.py) sit in the
// CU's LDS, every activation tile stays there, and the only cross-tile quantities are the weight gradients, which leave as
// per-workgroup partial tiles and are summed in workgroup order by mlp_reduce_k (deterministic, no atomics).
// The launch-per-layer form (tower.hip: 3 forward + head + 3 backward + the dW reduce = 8 launches, 58 us at batch 1 024)
// is a chain of launch latencies; this one is 2 launches.
// Dropout masks are the SAME counter-based hash as tower.hip's (drop_device.h: layer index, element index b * N + c), so the
// two forms drop the same units; sums are associated differently (fp32 rounding), nothing else differs.
//
// MFMA: v_mfma_f32_16x16x4_f32.  Lane (i = lane & 15, kq = lane >> 4): A operand row i, B operand column i, k = 4 kq + t over
// the four MFMAs t of a 16-wide k-step (same convention as tower.hip); the accumulator holds rows 4 kq + r of column i.
#include "rsx_common.h"
#include "drop_device.h"
#include "mlp_reduce_device.h"

namespace {
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

constexpr int MLP_MAX_W = 112;        // widest layer / input (a multiple of 16): 7 column tiles

struct MlpArgs {
  const float* X;                     // [B, K0]
  const float* W[MLP_MAX_L];          // [K_l, N_l]
  const float* b[MLP_MAX_L];          // [N_l]
  const float* mask[MLP_MAX_L];       // nullable explicit keep masks [B, N_l] (parity tests)
  const float* wout;                  // [N_last]
  const float* bout;                  // [1]
  const float* s0;                    // nullable [B]: added to the logit (din.py: the target item's bias)
  const float* labels;                // [B]
  const uint32_t* rng_step;
  float* prob;                        // [B]
  float* dX;                          // [B, K0]
  float* gs0;                         // nullable [B]: d loss / d s0
  float* part;                        // partial weight gradients, see poff
  uint32_t seed;
  float rate, loss_scale;             // loss_scale = 1 / (B * replicas)
  int B, K0, L;
  int N[MLP_MAX_L];
  // LDS layout (floats, from the host): weights [KP_l][ldw_l] (KP = K rounded to 16, zero rows / columns beyond K / N),
  // biases, the output weights, the activation tiles in'_l [16][ldin_l] (in'_0 = X, in'_{l+1} = dropout(relu(..))), the
  // backward multipliers g_l = keep-mask * relu' [16][ldin_{l+1}], two da tiles [16][ldda], per-row dz / ce
  int oW[MLP_MAX_L], ldw[MLP_MAX_L], oB[MLP_MAX_L], oWout;
  int oIn[MLP_MAX_L + 1], ldin[MLP_MAX_L + 1], oG[MLP_MAX_L];
  int oDa[2], ldda, oRow, lds_floats;
  // partial regions: layer l at part + poff[l]: [nwg][KR_l][NP_l] (KR = K + 1 rounded to 16: row K is the bias gradient;
  // NP = N rounded to 16); the output layer at part + poff[L]: [nwg][NPo] = dwout[N_last], dbout, loss term
  long long poff[MLP_MAX_L + 1];
  int NPo;
  // the entries a run-time index would pick (an index that is not a compile-time constant puts the whole struct in scratch)
  int NL, oInL, ldinL, oGL;
  long long poffL;
};

constexpr int MLP_T = 512, MLP_NW = MLP_T / 64;     // 8 waves = 2 per SIMD: one wave's LDS / MFMA latency hides behind the other's

// one 16x16 output tile, A rows from an LDS tile (float4 along k), B = W[k][col] read down a column (forward).  Even and odd
// k-steps go to two accumulators: consecutive MFMAs never depend on each other (a dependent 16x16x4 fp32 MFMA waits ~40 cycles)
__device__ __forceinline__ f32x4 tile_fwd(const float* __restrict__ arow /* in + i*ldi + 4*kq */,
                                          const float* __restrict__ bcol /* W + 4*kq*ld + col */, const int ld, const int nks) {
  f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
  int ks = 0;
  for (; ks + 1 < nks; ks += 2) {
    const float4 a0 = *reinterpret_cast<const float4*>(arow + 16 * ks);
    const float4 a1 = *reinterpret_cast<const float4*>(arow + 16 * ks + 16);
    const float* b0 = bcol + 16 * ks * ld;
    const float* b1 = b0 + 16 * ld;
    const float b00 = b0[0], b01 = b0[ld], b02 = b0[2 * ld], b03 = b0[3 * ld];
    const float b10 = b1[0], b11 = b1[ld], b12 = b1[2 * ld], b13 = b1[3 * ld];
    acc0 = mfma16(a0.x, b00, acc0);
    acc1 = mfma16(a1.x, b10, acc1);
    acc0 = mfma16(a0.y, b01, acc0);
    acc1 = mfma16(a1.y, b11, acc1);
    acc0 = mfma16(a0.z, b02, acc0);
    acc1 = mfma16(a1.z, b12, acc1);
    acc0 = mfma16(a0.w, b03, acc0);
    acc1 = mfma16(a1.w, b13, acc1);
  }
  if (ks < nks) {
    const float4 a0 = *reinterpret_cast<const float4*>(arow + 16 * ks);
    const float* b0 = bcol + 16 * ks * ld;
    acc0 = mfma16(a0.x, b0[0], acc0);
    acc0 = mfma16(a0.y, b0[ld], acc0);
    acc0 = mfma16(a0.z, b0[2 * ld], acc0);
    acc0 = mfma16(a0.w, b0[3 * ld], acc0);
  }
  return acc0 + acc1;
}
// the same with B = W[col][k] read along a row as float4 (d(input) = da . W^T)
__device__ __forceinline__ f32x4 tile_bwd(const float* __restrict__ arow /* da + i*ldda + 4*kq */,
                                          const float* __restrict__ brow /* W + col*ld + 4*kq */, const int nks) {
  f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
  int ks = 0;
  for (; ks + 1 < nks; ks += 2) {
    const float4 a0 = *reinterpret_cast<const float4*>(arow + 16 * ks);
    const float4 a1 = *reinterpret_cast<const float4*>(arow + 16 * ks + 16);
    const float4 b0 = *reinterpret_cast<const float4*>(brow + 16 * ks);
    const float4 b1 = *reinterpret_cast<const float4*>(brow + 16 * ks + 16);
    acc0 = mfma16(a0.x, b0.x, acc0);
    acc1 = mfma16(a1.x, b1.x, acc1);
    acc0 = mfma16(a0.y, b0.y, acc0);
    acc1 = mfma16(a1.y, b1.y, acc1);
    acc0 = mfma16(a0.z, b0.z, acc0);
    acc1 = mfma16(a1.z, b1.z, acc1);
    acc0 = mfma16(a0.w, b0.w, acc0);
    acc1 = mfma16(a1.w, b1.w, acc1);
  }
  if (ks < nks) {
    const float4 a0 = *reinterpret_cast<const float4*>(arow + 16 * ks);
    const float4 b0 = *reinterpret_cast<const float4*>(brow + 16 * ks);
    acc0 = mfma16(a0.x, b0.x, acc0);
    acc0 = mfma16(a0.y, b0.y, acc0);
    acc0 = mfma16(a0.z, b0.z, acc0);
    acc0 = mfma16(a0.w, b0.w, acc0);
  }
  return acc0 + acc1;
}

__global__ __launch_bounds__(MLP_T) void mlp_nobn_step_k(const MlpArgs p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, i = lane & 15, kq = lane >> 4;
  const int wg = blockIdx.x, row0 = wg * 16;
  const int L = p.L;
  RSX_STAMP(0, wg == 0);
  // ---- every global input of the tile is requested first (input rows: <= 1 float4 per thread; weights: <= 8), the LDS is
  // zeroed while they fly ------------------------------------------------------------------------------------------------
  const int K04 = p.K0 >> 2;
  float4 xv;
  {
    const int ec = tid < 16 * K04 ? tid : 16 * K04 - 1;
    const int r = ec / K04, c4 = ec - r * K04;
    const int row = row0 + r < p.B ? row0 + r : p.B - 1;
    xv = reinterpret_cast<const float4*>(p.X + (size_t)row * p.K0)[c4];
  }
  // (the dropout key's step counter, the row's label / extra logit and the output bias too: a global load costs ~1 us of
  // latency wherever it sits, and the phases below are a fraction of that)
  const uint32_t rng_st = p.rng_step != nullptr ? p.rng_step[0] : 0u;
  const int hrow = row0 + (tid >> 5) < p.B ? row0 + (tid >> 5) : p.B - 1;
  const float s0r = p.s0 ? p.s0[hrow] : 0.f, yr = p.labels[hrow], boutv = p.bout[0];
  constexpr int WPT = (MLP_MAX_W * MLP_MAX_W / 4 + MLP_T - 1) / MLP_T;      // float4 of the widest layer per thread: 7
  float4 wv[MLP_MAX_L][WPT];
#pragma unroll
  for (int l = 0; l < MLP_MAX_L; ++l) {
    const int K = l == 0 ? p.K0 : p.N[l > 0 ? l - 1 : 0], tot = l < L ? K * (p.N[l] >> 2) : 0;
#pragma unroll
    for (int u = 0; u < WPT; ++u) {
      const int e = tid + MLP_T * u;
      wv[l][u] = l < L ? reinterpret_cast<const float4*>(p.W[l])[e < tot ? e : tot - 1] : F4Z;
    }
  }
  for (int e = tid; e < (p.lds_floats >> 2); e += MLP_T) reinterpret_cast<float4*>(lds)[e] = F4Z;
  __syncthreads();
  RSX_STAMP(1, wg == 0);
  if (tid < 16 * K04) {
    const int r = tid / K04, c4 = tid - r * K04;
    *reinterpret_cast<float4*>(lds + p.oIn[0] + r * p.ldin[0] + 4 * c4) = xv;
  }
#pragma unroll
  for (int l = 0; l < MLP_MAX_L; ++l) {
    if (l < L) {
      const int K = l == 0 ? p.K0 : p.N[l > 0 ? l - 1 : 0], N = p.N[l], N4 = N >> 2, tot = K * N4;
      float* dst = lds + p.oW[l];
      const int ld = p.ldw[l];
      int r = tid / N4, c4 = tid - r * N4;                         // (one division; then stepped)
      const int dr = MLP_T / N4, dc = MLP_T - dr * N4;
#pragma unroll
      for (int u = 0; u < WPT; ++u) {
        if (tid + MLP_T * u < tot) *reinterpret_cast<float4*>(dst + r * ld + 4 * c4) = wv[l][u];
        r += dr;
        c4 += dc;
        if (c4 >= N4) { c4 -= N4; ++r; }
      }
      for (int c = tid; c < N; c += MLP_T) lds[p.oB[l] + c] = p.b[l][c];
    }
  }
  const int NL = p.NL;
  for (int c = tid; c < NL; c += MLP_T) lds[p.oWout + c] = p.wout[c];
  __syncthreads();
  RSX_STAMP(2, wg == 0);
  // ---- forward --------------------------------------------------------------------------------------------------------
#pragma unroll
  for (int l = 0; l < MLP_MAX_L; ++l) {
    if (l >= L) break;
    const int K = l == 0 ? p.K0 : p.N[l > 0 ? l - 1 : 0], N = p.N[l];
    const int nks = (K + 15) >> 4, ntj = (N + 15) >> 4;
    const float* in = lds + p.oIn[l];
    const int ldi = p.ldin[l];
    const float* Wl = lds + p.oW[l];
    const int ld = p.ldw[l];
    float* out = lds + p.oIn[l + 1];
    float* gm = lds + p.oG[l];
    const int ldo = p.ldin[l + 1];
    DropRng dr;                                                   // = drop_make(rate, mask, rng_step, seed, l) on the preloaded step
    dr.inv_keep = 1.0f / (1.0f - p.rate);
    dr.mode = p.rate == 0.f ? 0 : (p.mask[l] != nullptr ? 1 : 2);
    dr.thresh = (uint32_t)((double)p.rate * 4294967296.0);
    dr.key = rsx_hash32(p.seed ^ (rng_st * 0x9E3779B9u) ^ ((uint32_t)l * 0x85EBCA6Bu + 0x27220A95u));
    for (int j = w; j < ntj; j += MLP_NW) {
      const int col = 16 * j + i;
      const f32x4 acc = tile_fwd(in + i * ldi + 4 * kq, Wl + 4 * kq * ld + col, ld, nks);
      const float bias = col < N ? lds[p.oB[l] + col] : 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = 4 * kq + r, grow = row0 + row;
        float o = acc[r] + bias;
        o = o > 0.f ? o : 0.f;
        const bool ok = col < N && grow < p.B;
        const float mk = ok ? drop_mul(dr, p.mask[l], (size_t)grow * N + col) : 0.f;
        if (col < N) {
          out[row * ldo + col] = ok ? o * mk : 0.f;
          gm[row * ldo + col] = (ok && o > 0.f) ? mk : 0.f;
        }
      }
    }
    __syncthreads();
    RSX_STAMP(3 + l, wg == 0);
  }
  // ---- logit, loss and its gradient (din/din.py:138-147): 32 lanes per row, columns li, li + 32, .. -------------------
  float* rowb = lds + p.oRow;                                     // [16] dz, [16] ce
  {
    const int row = tid >> 5, li = tid & 31, grow = row0 + row;
    const bool rok = grow < p.B;
    const float* inL = lds + p.oInL + row * p.ldinL;
    float dot = 0.f;
#pragma unroll
    for (int k = 0; k < (MLP_MAX_W + 31) / 32; ++k) {
      const int c = li + 32 * k;
      dot += c < NL ? inL[c] * lds[p.oWout + c] : 0.f;
    }
#pragma unroll
    for (int m = 1; m < 32; m <<= 1) dot += __shfl_xor(dot, m);   // (fixed tree over the row's 32 lanes)
    if (li == 0) {
      const float zz = s0r + (dot + boutv);
      const float y = yr;
      const float pr = 1.f / (1.f + expf(-zz));
      const float ce = fmaxf(zz, 0.f) - zz * y + log1pf(expf(-fabsf(zz)));
      const float dz = rok ? (pr - y) * p.loss_scale : 0.f;
      if (rok) {
        p.prob[grow] = pr;
        if (p.gs0) p.gs0[grow] = dz;
      }
      rowb[row] = dz;
      rowb[16 + row] = rok ? ce : 0.f;
    }
  }
  __syncthreads();
  // output layer's gradients and the loss term of this tile: rows in ascending order (all 16 operands requested first)
  {
    float* po = p.part + p.poffL + (size_t)wg * p.NPo;
    if (tid < NL) {
      float a[16], d[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) { a[r] = lds[p.oInL + r * p.ldinL + tid]; d[r] = rowb[r]; }
      float s = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) s += d[r] * a[r];
      po[tid] = s;
    } else if (tid == NL || tid == NL + 1) {
      float s = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) s += rowb[(tid - NL) * 16 + r];
      po[tid] = s;
    } else if (tid < p.NPo) {
      po[tid] = 0.f;
    }
  }
  // da of the last hidden layer: dz * wout * g (thread = (row, 32-lane column group))
  {
    float* da = lds + p.oDa[0];
    const float* gm = lds + p.oGL;
    const int ldg = p.ldinL;
    const int row = tid >> 5, li = tid & 31;
    const float dzr = rowb[row];
    for (int c = li; c < p.ldda; c += 32) da[row * p.ldda + c] = c < NL ? dzr * lds[p.oWout + c] * gm[row * ldg + c] : 0.f;
  }
  __syncthreads();
  RSX_STAMP(6, wg == 0);
  // ---- backward -------------------------------------------------------------------------------------------------------
#pragma unroll
  for (int lr = 0; lr < MLP_MAX_L; ++lr) {
    const int l = MLP_MAX_L - 1 - lr;                             // (compile-time after unrolling)
    if (l >= L) continue;
    const int K = l == 0 ? p.K0 : p.N[l > 0 ? l - 1 : 0], N = p.N[l];
    const int KR = (K + 1 + 15) & ~15, NP = (N + 15) & ~15;
    const bool odd = ((L - 1 - l) & 1) != 0;
    const float* da = lds + (odd ? p.oDa[1] : p.oDa[0]);
    float* dan = lds + (odd ? p.oDa[0] : p.oDa[1]);               // da of layer l - 1 (written below)
    const float* in = lds + p.oIn[l];
    const int ldi = p.ldin[l];
    const float* Wl = lds + p.oW[l];
    const int ld = p.ldw[l];
    // (a) d(input) = da . W^T: column tiles over K; wave w: tiles w, w + 8, ..
    const int ntk = (K + 15) >> 4, nkn = NP >> 4;
    for (int j = w; j < ntk; j += MLP_NW) {
      const int col = 16 * j + i;
      const f32x4 acc = tile_bwd(da + i * p.ldda + 4 * kq, Wl + col * ld + 4 * kq, nkn);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = 4 * kq + r, grow = row0 + row;
        if (l == 0) {
          if (col < K && grow < p.B) p.dX[(size_t)grow * K + col] = acc[r];
        } else if (col < K) {
          dan[row * p.ldda + col] = acc[r] * lds[p.oG[l > 0 ? l - 1 : 0] + row * ldi + col];
        }
      }
    }
    if (l > 0) {                                                  // the next da tile's columns beyond K: zero
      const int row = tid >> 5, li = tid & 31;
      for (int c = K + li; c < p.ldda; c += 32) dan[row * p.ldda + c] = 0.f;
    }
    // (b) dW partial [KR][NP] = [in' | 1]^T . da over the tile's 16 rows (one k-step); tiles dealt from the last wave downwards
    // (the first waves hold the d(input) tiles), TWO tiles at a time: their MFMA chains interleave
    {
      const int ntm = KR >> 4, ntn = NP >> 4, ntt = ntm * ntn;
      const int rcp = 65536 / ntn + 1;
      float* po = p.part + p.poff[l] + (size_t)wg * KR * NP;
      for (int tt = MLP_NW - 1 - w; tt < ntt; tt += 2 * MLP_NW) {
        const int t2 = tt + MLP_NW < ntt ? tt + MLP_NW : tt;      // (second tile; the last odd one is computed twice, stored once)
        // (tt < 64 and ntn <= 7: an exact multiply-shift instead of two integer divisions -- ~50 instructions per tile pair)
        const int m0 = (tt * rcp) >> 16, j0 = tt - m0 * ntn, m1 = (t2 * rcp) >> 16, j1 = t2 - m1 * ntn;
        const int f0 = 16 * m0 + i, f1 = 16 * m1 + i;
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
        float a0[4], a1[4], b0[4], b1[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const int row = 4 * kq + t;
          // (feature K: the ones-row that yields the bias gradient; in' tiles are zero beyond K, and their stride covers KR)
          a0[t] = f0 == K ? 1.f : in[row * ldi + f0];
          a1[t] = f1 == K ? 1.f : in[row * ldi + f1];
          b0[t] = da[row * p.ldda + 16 * j0 + i];
          b1[t] = da[row * p.ldda + 16 * j1 + i];
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          acc0 = mfma16(a0[t], b0[t], acc0);
          acc1 = mfma16(a1[t], b1[t], acc1);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) po[(16 * m0 + 4 * kq + r) * NP + 16 * j0 + i] = acc0[r];
        if (t2 != tt) {
#pragma unroll
          for (int r = 0; r < 4; ++r) po[(16 * m1 + 4 * kq + r) * NP + 16 * j1 + i] = acc1[r];
        }
      }
    }
    RSX_STAMP(7 + 2 * lr, wg == 0);
    __syncthreads();
    RSX_STAMP(8 + 2 * lr, wg == 0);
  }
}

// one thread per float4 of a region, 32 loads in flight (mlp_reduce_device.h)
// (64-thread workgroups: the 5.6 MB of partials were written by 64 other CUs a moment ago and come back from memory --
// 85 small workgroups pull them through 85 CUs' L1s instead of 22)
__global__ __launch_bounds__(64) void mlp_reduce_k(const MlpRed r) { mlp_reduce_body<32>(r, blockIdx.x * 64 + threadIdx.x); }

inline int up16(int x) { return (x + 15) & ~15; }
inline bool al16(const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15u) == 0; }
}  // namespace

extern "C" size_t rsx_mlp_nobn_workspace_floats(int B, int K0, const int32_t* widths, int L) {
  if (B <= 0 || K0 <= 0 || !widths || L <= 0 || L > MLP_MAX_L) return 0;
  const size_t nwg = ((size_t)B + 15) / 16;
  size_t per = 0;
  int K = K0;
  for (int l = 0; l < L; ++l) {
    per += (size_t)up16(K + 1) * up16(widths[l]);
    K = widths[l];
  }
  per += (size_t)((widths[L - 1] + 2 + 3) & ~3);
  return nwg * per;
}

extern "C" int rsx_mlp_nobn_supported(int K0, const int32_t* widths, int L) {
  if (!widths || L <= 0 || L > MLP_MAX_L || K0 <= 0 || K0 > MLP_MAX_W || (K0 & 3)) return 0;
  for (int l = 0; l < L; ++l)
    if (widths[l] <= 0 || widths[l] > MLP_MAX_W || (widths[l] & 3)) return 0;
  return 1;
}

static int mlp_build(const rsx_mlp_step* s, MlpArgs& p, MlpRed& r, size_t& lds_bytes, unsigned& e4_total) {
  if (!s) return RSX_EINVAL;
  const int L = s->L, B = s->B, K0 = s->K0;
  if (B < 0 || L <= 0 || L > MLP_MAX_L) return RSX_EINVAL;
  if (!rsx_mlp_nobn_supported(K0, s->widths, L)) return RSX_EUNSUPPORTED;
  if (B == 0) { e4_total = 0; return RSX_OK; }
  if (!s->X || !s->wout || !s->bout || !s->labels || !s->prob || !s->dX || !s->workspace || !s->dwout || !s->dbout || !s->loss)
    return RSX_EINVAL;
  if (s->dropout_rate < 0.f || s->dropout_rate >= 1.f) return RSX_EINVAL;
  if (!al16(s->X) || !al16(s->dX) || !al16(s->workspace)) return RSX_EUNSUPPORTED;
  p.X = s->X; p.wout = s->wout; p.bout = s->bout; p.s0 = s->s0; p.labels = s->labels; p.rng_step = s->rng_step;
  p.prob = s->prob; p.dX = s->dX; p.gs0 = s->gs0; p.part = s->workspace;
  p.seed = s->seed; p.rate = s->dropout_rate; p.loss_scale = s->loss_scale;
  p.B = B; p.K0 = K0; p.L = L;
  const int nwg = (B + 15) / 16;
  int off = 0, K = K0;
  long long po = 0;
  unsigned e4 = 0;
  int wmax = K0;
  for (int l = 0; l < MLP_MAX_L; ++l) {
    const bool on = l < L;
    const int N = on ? s->widths[l] : 4;
    p.W[l] = on ? s->W[l] : nullptr; p.b[l] = on ? s->b[l] : nullptr; p.mask[l] = on ? s->masks[l] : nullptr;
    p.N[l] = N;
    r.dW[l] = on ? s->dW[l] : nullptr; r.db[l] = on ? s->db[l] : nullptr; r.K[l] = K; r.N[l] = N;
    p.oW[l] = 0; p.ldw[l] = 0; p.oB[l] = 0; p.poff[l] = 0; r.poff[l] = 0;
    if (!on) { r.e4_end[l] = 0xFFFFFFFFu; continue; }
    if (!p.W[l] || !p.b[l] || !r.dW[l] || !r.db[l]) return RSX_EINVAL;
    if (!al16(p.W[l])) return RSX_EUNSUPPORTED;
    p.ldw[l] = up16(N) + 4;
    p.oW[l] = off; off += up16(K) * p.ldw[l];
    p.poff[l] = po; r.poff[l] = po;
    po += (long long)nwg * up16(K + 1) * up16(N);
    e4 += (unsigned)(up16(K + 1) * up16(N) / 4);
    r.e4_end[l] = e4;
    wmax = N > wmax ? N : wmax;
    K = N;
  }
  const int NL = s->widths[L - 1];
  p.NPo = (NL + 2 + 3) & ~3;
  for (int l = L; l <= MLP_MAX_L; ++l) { p.poff[l] = po; r.poff[l] = po; }
  p.poffL = po;
  e4 += (unsigned)(p.NPo / 4);
  r.e4_last = e4;
  for (int l = 0; l < L; ++l) { p.oB[l] = off; off += (s->widths[l] + 3) & ~3; }
  p.oWout = off; off += (NL + 3) & ~3;
  // activation tiles: stride = (width + 1 rounded to 16) + 8 floats (covers the ones-row feature of the dW operand; == 8 mod 16)
  K = K0;
  for (int l = 0; l <= MLP_MAX_L; ++l) {
    p.oIn[l] = 0; p.ldin[l] = 0;
    if (l < MLP_MAX_L) p.oG[l] = 0;
  }
  for (int l = 0; l <= L; ++l) {
    const int wdt = l == 0 ? K0 : s->widths[l - 1];
    p.ldin[l] = up16(wdt + 1) + 8;
    p.oIn[l] = off; off += 16 * p.ldin[l];
    if (l > 0) { p.oG[l - 1] = off; off += 16 * p.ldin[l]; }
  }
  p.ldda = up16(wmax) + 8;
  p.oDa[0] = off; off += 16 * p.ldda;
  p.oDa[1] = off; off += 16 * p.ldda;
  p.oRow = off; off += 32;
  p.NL = NL; p.oInL = p.oIn[L]; p.ldinL = p.ldin[L]; p.oGL = p.oG[L - 1];
  p.lds_floats = (off + 3) & ~3;
  const size_t lds = (size_t)p.lds_floats * sizeof(float);
  if (lds > 160 * 1024) return RSX_EUNSUPPORTED;
  if (lds > 64 * 1024) {
    static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(mlp_nobn_step_k),
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (attr != hipSuccess) return RSX_EUNSUPPORTED;
  }
  r.part = s->workspace; r.dwout = s->dwout; r.dbout = s->dbout; r.loss = s->loss;
  r.L = L; r.nwg = nwg; r.NPo = p.NPo; r.NL = NL; r.inv_B = 1.0 / (double)B;
  lds_bytes = lds;
  e4_total = e4;
  return RSX_OK;
}

extern "C" int rsx_mlp_nobn_train_step(const rsx_mlp_step* s, rsx_stream_t stream) {
  MlpArgs p;
  MlpRed r;
  size_t lds = 0;
  unsigned e4 = 0;
  const int rc = mlp_build(s, p, r, lds, e4);
  if (rc != RSX_OK || e4 == 0) return rc;
  RSX_LAUNCH(mlp_nobn_step_k, dim3((s->B + 15) / 16), dim3(MLP_T), lds, rsx_s(stream), p);
  // (defer_reduce: the caller issues rsx_mlp_nobn_reduce itself -- on another stream, or later: the weight gradients are
  // first needed by the optimizer)
  if (!s->defer_reduce) RSX_LAUNCH(mlp_reduce_k, dim3((e4 + 63) / 64), dim3(64), 0, rsx_s(stream), r);
  RSX_CHECK_LAUNCH();
  return RSX_OK;
}

extern "C" int rsx_mlp_nobn_reduce_job(const rsx_mlp_step* s, rsx_mlp_reduce_job* job_out) {
  if (!job_out) return RSX_EINVAL;
  MlpArgs p;
  size_t lds = 0;
  unsigned e4 = 0;
  job_out->e4_last = 0;
  return mlp_build(s, p, *job_out, lds, e4);
}

extern "C" int rsx_mlp_nobn_reduce(const rsx_mlp_step* s, rsx_stream_t stream) {
  MlpArgs p;
  MlpRed r;
  size_t lds = 0;
  unsigned e4 = 0;
  const int rc = mlp_build(s, p, r, lds, e4);
  if (rc != RSX_OK || e4 == 0) return rc;
  RSX_LAUNCH(mlp_reduce_k, dim3((e4 + 63) / 64), dim3(64), 0, rsx_s(stream), r);
  RSX_CHECK_LAUNCH();
  return RSX_OK;
}


#ifdef RSX_STAMPS
extern "C" int rsx_dbg_stamps_mlp(unsigned long long* out_h) {
  return hipMemcpyFromSymbol(out_h, HIP_SYMBOL(rsx_stamps_d), sizeof(unsigned long long) * 64) == hipSuccess ? RSX_OK : RSX_ELAUNCH;
}
#endif
