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

namespace {
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

constexpr int MLP_MAX_L = 3;          // hidden layers
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

__global__ __launch_bounds__(256) void mlp_nobn_step_k(const MlpArgs p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, i = lane & 15, kq = lane >> 4;
  const int wg = blockIdx.x, row0 = wg * 16;
  const int L = p.L;
  RSX_STAMP(0, wg == 0);
  // ---- the tile's input rows are requested first; LDS zeroed; weights in ---------------------------------------------
  const int K04 = p.K0 >> 2;
  float4 xv[2];                                                   // 16 * K0/4 float4 over 256 threads: <= 2 each (K0 <= 112)
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int e = tid + 256 * u, ec = e < 16 * K04 ? e : 16 * K04 - 1;
    const int r = ec / K04, c4 = ec - r * K04;
    const int row = row0 + r < p.B ? row0 + r : p.B - 1;
    xv[u] = reinterpret_cast<const float4*>(p.X + (size_t)row * p.K0)[c4];
  }
  for (int e = tid; e < (p.lds_floats >> 2); e += 256) reinterpret_cast<float4*>(lds)[e] = F4Z;
  __syncthreads();
  RSX_STAMP(1, wg == 0);
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int e = tid + 256 * u;
    if (e < 16 * K04) {
      const int r = e / K04, c4 = e - r * K04;
      *reinterpret_cast<float4*>(lds + p.oIn[0] + r * p.ldin[0] + 4 * c4) = xv[u];
    }
  }
#pragma unroll
  for (int l = 0; l < MLP_MAX_L; ++l) {
    if (l < L) {
      const int K = l == 0 ? p.K0 : p.N[l > 0 ? l - 1 : 0], N = p.N[l], N4 = N >> 2;
      const float* Wl = p.W[l];
      float* dst = lds + p.oW[l];
      const int ld = p.ldw[l];
      int e = tid;
      for (; e + 7 * 256 < K * N4; e += 8 * 256) {                 // 8 x 16-byte loads in flight
        float4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = reinterpret_cast<const float4*>(Wl)[e + 256 * u];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int ee = e + 256 * u, r = ee / N4, c4 = ee - r * N4;
          *reinterpret_cast<float4*>(dst + r * ld + 4 * c4) = v[u];
        }
      }
      for (; e < K * N4; e += 256) {
        const int r = e / N4, c4 = e - r * N4;
        *reinterpret_cast<float4*>(dst + r * ld + 4 * c4) = reinterpret_cast<const float4*>(Wl)[e];
      }
      for (int c = tid; c < N; c += 256) lds[p.oB[l] + c] = p.b[l][c];
    }
  }
  const int NL = p.NL;
  for (int c = tid; c < NL; c += 256) lds[p.oWout + c] = p.wout[c];
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
    const DropRng dr = drop_make(p.rate, p.mask[l], p.rng_step, p.seed, (uint32_t)l);
    for (int j = w; j < ntj; j += 4) {
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      for (int ks = 0; ks < nks; ++ks) {
        const float4 a = *reinterpret_cast<const float4*>(in + i * ldi + 16 * ks + 4 * kq);
        const float* br = Wl + (16 * ks + 4 * kq) * ld + 16 * j + i;
        const float b0 = br[0], b1 = br[ld], b2 = br[2 * ld], b3 = br[3 * ld];
        acc = mfma16(a.x, b0, acc);
        acc = mfma16(a.y, b1, acc);
        acc = mfma16(a.z, b2, acc);
        acc = mfma16(a.w, b3, acc);
      }
      const int col = 16 * j + i;
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
  // ---- logit, loss and its gradient: one thread per row (din/din.py:138-147) ----------------------------------------
  float* rowb = lds + p.oRow;                                     // [16] dz, [16] ce
  if (tid < 16) {
    const int grow = row0 + tid;
    const bool rok = grow < p.B;
    const float* inL = lds + p.oInL + tid * p.ldinL;
    float dot = 0.f;
    for (int c = 0; c < NL; ++c) dot += inL[c] * lds[p.oWout + c];
    const size_t rc = (size_t)(rok ? grow : p.B - 1);
    const float zz = (p.s0 ? p.s0[rc] : 0.f) + (dot + p.bout[0]);
    const float y = p.labels[rc];
    const float pr = 1.f / (1.f + expf(-zz));
    const float ce = fmaxf(zz, 0.f) - zz * y + log1pf(expf(-fabsf(zz)));
    const float dz = rok ? (pr - y) * p.loss_scale : 0.f;
    if (rok) {
      p.prob[grow] = pr;
      if (p.gs0) p.gs0[grow] = dz;
    }
    rowb[tid] = dz;
    rowb[16 + tid] = rok ? ce : 0.f;
  }
  __syncthreads();
  // output layer's gradients and the loss term of this tile: rows in ascending order
  {
    float* po = p.part + p.poffL + (size_t)wg * p.NPo;
    if (tid < NL) {
      float s = 0.f;
      for (int r = 0; r < 16; ++r) s += rowb[r] * lds[p.oInL + r * p.ldinL + tid];
      po[tid] = s;
    } else if (tid == NL || tid == NL + 1) {
      float s = 0.f;
      for (int r = 0; r < 16; ++r) s += rowb[(tid - NL) * 16 + r];
      po[tid] = s;
    } else if (tid < p.NPo) {
      po[tid] = 0.f;
    }
  }
  // da of the last hidden layer: dz * wout * g
  {
    float* da = lds + p.oDa[0];
    const float* gm = lds + p.oGL;
    const int ldg = p.ldinL;
    for (int e = tid; e < 16 * p.ldda; e += 256) {
      const int r = e / p.ldda, c = e - r * p.ldda;
      da[e] = c < NL ? rowb[r] * lds[p.oWout + c] * gm[r * ldg + c] : 0.f;
    }
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
    // (a) d(input) = da . W^T: column tiles over K; wave w: tiles w, w + 4, ..
    const int ntk = (K + 15) >> 4, nkn = NP >> 4;
    for (int j = w; j < ntk; j += 4) {
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      for (int ks = 0; ks < nkn; ++ks) {
        const float4 a = *reinterpret_cast<const float4*>(da + i * p.ldda + 16 * ks + 4 * kq);
        const float4 bq = *reinterpret_cast<const float4*>(Wl + (16 * j + i) * ld + 16 * ks + 4 * kq);
        acc = mfma16(a.x, bq.x, acc);
        acc = mfma16(a.y, bq.y, acc);
        acc = mfma16(a.z, bq.z, acc);
        acc = mfma16(a.w, bq.w, acc);
      }
      const int col = 16 * j + i;
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
      const int padw = p.ldda - K;
      for (int e = tid; e < 16 * padw; e += 256) {
        const int r = e / padw, c = e - r * padw;
        dan[r * p.ldda + K + c] = 0.f;
      }
    }
    // (b) dW partial [KR][NP] = [in' | 1]^T . da over the tile's 16 rows (one k-step); tiles dealt from wave 3 downwards
    {
      const int ntm = KR >> 4, ntn = NP >> 4;
      float* po = p.part + p.poff[l] + (size_t)wg * KR * NP;
      for (int tt = 3 - w; tt < ntm * ntn; tt += 4) {
        const int m = tt / ntn, jn = tt - m * ntn;
        const int feat = 16 * m + i;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const int row = 4 * kq + t;
          // (feature K: the ones-row that yields the bias gradient; in' tiles are zero beyond K, and their stride covers KR)
          const float a = feat == K ? 1.f : in[row * ldi + feat];
          const float bv = da[row * p.ldda + 16 * jn + i];
          acc = mfma16(a, bv, acc);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) po[(size_t)(16 * m + 4 * kq + r) * NP + 16 * jn + i] = acc[r];
      }
    }
    RSX_STAMP(7 + 2 * lr, wg == 0);
    __syncthreads();
    RSX_STAMP(8 + 2 * lr, wg == 0);
  }
}

struct MlpRed {
  const float* part;
  long long poff[MLP_MAX_L + 1];
  float* dW[MLP_MAX_L];
  float* db[MLP_MAX_L];
  float* dwout;
  float* dbout;
  float* loss;
  int K[MLP_MAX_L], N[MLP_MAX_L];
  unsigned e4_end[MLP_MAX_L];         // float4 elements of the layer regions 0 .. q (unused layers: never reached)
  unsigned e4_last;
  int L, nwg, NPo, NL;
  double inv_B;
};
// one thread per float4 of a region: the nwg partials in ascending workgroup order, 16 loads in flight
__global__ __launch_bounds__(256) void mlp_reduce_k(const MlpRed r) {
  const unsigned e = blockIdx.x * 256 + threadIdx.x;
  int q = 0;
  unsigned base = 0;
#pragma unroll
  for (int k = 0; k < MLP_MAX_L; ++k) {
    if (k < r.L && e >= r.e4_end[k]) {
      q = k + 1;
      base = r.e4_end[k];
    }
  }
  if (e >= r.e4_last) return;
  const unsigned e4 = e - base;
  const int K = q == 0 ? r.K[0] : (q == 1 ? r.K[1] : r.K[2]), N = q == 0 ? r.N[0] : (q == 1 ? r.N[1] : r.N[2]);
  const int KR = (K + 1 + 15) & ~15, NP = (N + 15) & ~15;
  const size_t reg4 = q < r.L ? (size_t)KR * NP / 4 : (size_t)r.NPo / 4;
  const long long pq = q == r.L ? r.poff[MLP_MAX_L] : (q == 0 ? r.poff[0] : (q == 1 ? r.poff[1] : r.poff[2]));
  const float4* src = reinterpret_cast<const float4*>(r.part + pq) + e4;
  float4 s = F4Z;
  double sl = 0.0;
  for (int g = 0; g < r.nwg; g += 16) {
    float4 t[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) t[u] = src[(size_t)(g + u < r.nwg ? g + u : r.nwg - 1) * reg4];
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      if (g + u < r.nwg) {
        s = f4_add(s, t[u]);
        // (the loss term is summed in fp64: 1 024 terms of ~0.7 in fp32 would cost the reported loss its last digits)
        if (q == r.L) {
          const int c0 = (int)e4 * 4, cl = r.NL + 1 - c0;
          if (cl >= 0 && cl < 4) sl += (double)(cl == 0 ? t[u].x : cl == 1 ? t[u].y : cl == 2 ? t[u].z : t[u].w);
        }
      }
    }
  }
  const float v[4] = {s.x, s.y, s.z, s.w};
  if (q < r.L) {
    const int kk = (int)(((size_t)e4 * 4) / NP), n = (int)(((size_t)e4 * 4) - (size_t)kk * NP);
    float* dW = q == 0 ? r.dW[0] : (q == 1 ? r.dW[1] : r.dW[2]);
    float* db = q == 0 ? r.db[0] : (q == 1 ? r.db[1] : r.db[2]);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      if (n + t < N) {
        if (kk < K) dW[(size_t)kk * N + n + t] = v[t];
        else if (kk == K) db[n + t] = v[t];
      }
    }
  } else {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int c = (int)e4 * 4 + t;
      if (c < r.NL) r.dwout[c] = v[t];
      else if (c == r.NL) r.dbout[0] = v[t];
      else if (c == r.NL + 1) r.loss[0] = (float)(sl * r.inv_B);
    }
  }
}

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

extern "C" int rsx_mlp_nobn_train_step(const rsx_mlp_step* s, rsx_stream_t stream) {
  if (!s) return RSX_EINVAL;
  const int L = s->L, B = s->B, K0 = s->K0;
  if (B < 0 || L <= 0 || L > MLP_MAX_L) return RSX_EINVAL;
  if (!rsx_mlp_nobn_supported(K0, s->widths, L)) return RSX_EUNSUPPORTED;
  if (B == 0) return RSX_OK;
  if (!s->X || !s->wout || !s->bout || !s->labels || !s->prob || !s->dX || !s->workspace || !s->dwout || !s->dbout || !s->loss)
    return RSX_EINVAL;
  if (s->dropout_rate < 0.f || s->dropout_rate >= 1.f) return RSX_EINVAL;
  if (!al16(s->X) || !al16(s->dX) || !al16(s->workspace)) return RSX_EUNSUPPORTED;
  MlpArgs p;
  MlpRed r;
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
  RSX_LAUNCH(mlp_nobn_step_k, dim3(nwg), dim3(256), lds, rsx_s(stream), p);
  RSX_LAUNCH(mlp_reduce_k, dim3((e4 + 255) / 256), dim3(256), 0, rsx_s(stream), r);
  RSX_CHECK_LAUNCH();
  return RSX_OK;
}

#ifdef RSX_STAMPS
extern "C" int rsx_dbg_stamps_mlp(unsigned long long* out_h) {
  return hipMemcpyFromSymbol(out_h, HIP_SYMBOL(rsx_stamps_d), sizeof(unsigned long long) * 64) == hipSuccess ? RSX_OK : RSX_ELAUNCH;
}
#endif
