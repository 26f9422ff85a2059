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

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr float TOWER_BN_EPS = 1e-3f;  // tf.layers.batch_normalization default epsilon
constexpr int TM = 16;                 // rows per row tile

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// mean / rstd of one column from the per-row-tile partial sums (sum a, sum a^2), fixed order, fp64
__device__ __forceinline__ void bn_col_stats(const double* __restrict__ fstat, int RT, int N, int col, int B,
                                             float& mean, float& rstd) {
  double s1 = 0.0, s2 = 0.0;
  for (int r = 0; r < RT; ++r) {
    s1 += fstat[((size_t)r * 2 + 0) * N + col];
    s2 += fstat[((size_t)r * 2 + 1) * N + col];
  }
  const double mu = s1 / B;
  double var = s2 / B - mu * mu;
  if (var < 0.0) var = 0.0;
  mean = (float)mu;
  rstd = 1.0f / sqrtf((float)var + TOWER_BN_EPS);
}

// ---------------------------------------------------------------------------------------------
// forward layer:  a_out = relu(in' . W + bias),  in' = in (first layer) or dropout(BN(in)) (others)
// grid = (ceil(N/16), ceil(B/16)), block = 64 (one wave per 16x16 tile).  dyn LDS: 2*K floats.
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
  const float* mask_prev;    // [B, K] 1 keep / 0 drop (null: no dropout)
  float* bn_prev_out;        // [2, K] mean, rstd (written by block (0,0) for the backward pass)
  float inv_keep;
  int B, K, N, RT;
};

__global__ __launch_bounds__(64) void tower_fwd_k(const FwdArgs p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* sc = lds;          // [K] scale
  float* sh = lds + p.K;    // [K] shift
  const int lane = threadIdx.x;
  const bool first = p.fstat_prev == nullptr;
  if (!first) {
    for (int k = lane; k < p.K; k += 64) {
      float mean, rstd;
      bn_col_stats(p.fstat_prev, p.RT, p.K, k, p.B, mean, rstd);
      const float inv = rstd * p.gamma_prev[k];
      sc[k] = inv;
      sh[k] = p.beta_prev[k] - mean * inv;
      if (blockIdx.x == 0 && blockIdx.y == 0) {
        p.bn_prev_out[k] = mean;
        p.bn_prev_out[p.K + k] = rstd;
      }
    }
    __syncthreads();
  }
  const int i = lane & 15, kq = lane >> 4;
  const int row = blockIdx.y * TM + i;
  const int col = blockIdx.x * 16 + i;   // B-operand column for this lane (j = lane & 15)
  const bool rok = row < p.B, cok = col < p.N;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int k0 = 0; k0 < p.K; k0 += 16) {
    const int kk = k0 + 4 * kq;
    float4 a = z4;
    if (rok && kk < p.K) {
      a = *reinterpret_cast<const float4*>(p.in + (size_t)row * p.K + kk);
      if (!first) {
        const float4 s = *reinterpret_cast<const float4*>(sc + kk);
        const float4 h = *reinterpret_cast<const float4*>(sh + kk);
        a = make_float4(a.x * s.x + h.x, a.y * s.y + h.y, a.z * s.z + h.z, a.w * s.w + h.w);
        if (p.mask_prev != nullptr) {
          const float4 m = *reinterpret_cast<const float4*>(p.mask_prev + (size_t)row * p.K + kk);
          a = make_float4(a.x * m.x * p.inv_keep, a.y * m.y * p.inv_keep, a.z * m.z * p.inv_keep,
                          a.w * m.w * p.inv_keep);
        }
      }
    }
    float b0 = 0.f, b1 = 0.f, b2 = 0.f, b3 = 0.f;
    if (cok && kk < p.K) {
      const float* w = p.W + (size_t)kk * p.N + col;
      b0 = w[0];
      b1 = w[p.N];
      b2 = w[2 * (size_t)p.N];
      b3 = w[3 * (size_t)p.N];
    }
    acc = mfma16(a.x, b0, acc);
    acc = mfma16(a.y, b1, acc);
    acc = mfma16(a.z, b2, acc);
    acc = mfma16(a.w, b3, acc);
  }
  // epilogue: C layout col = lane & 15, row = (lane >> 4) * 4 + r
  const int ocol = blockIdx.x * 16 + (lane & 15);
  const float bv = ocol < p.N ? p.bias[ocol] : 0.f;
  double s1 = 0.0, s2 = 0.0;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int orow = blockIdx.y * TM + (lane >> 4) * 4 + r;
    float v = acc[r] + bv;
    v = v > 0.f ? v : 0.f;
    if (orow < p.B && ocol < p.N) {
      p.a_out[(size_t)orow * p.N + ocol] = v;
      s1 += (double)v;
      s2 += (double)v * (double)v;
    }
  }
  s1 += __shfl_xor(s1, 16);
  s2 += __shfl_xor(s2, 16);
  s1 += __shfl_xor(s1, 32);
  s2 += __shfl_xor(s2, 32);
  if (lane < 16 && ocol < p.N && p.fstat_out != nullptr) {
    p.fstat_out[((size_t)blockIdx.y * 2 + 0) * p.N + ocol] = s1;
    p.fstat_out[((size_t)blockIdx.y * 2 + 1) * p.N + ocol] = s2;
  }
}

// ---------------------------------------------------------------------------------------------
// head (DeepFM family):  o = dropout(BN(a_last));  u = o . wd + bd;  t2 = act2(u)
//   z = wo[0]*act0(s0 + c0) + wo[1]*s1 + wo[2]*t2 + bo ;  loss = mean sigmoid-CE(z, y)
// and its backward in the same pass: dz, d s0, d s1, d(BN output of the last layer) + partials.
// act0/act2 = relu when the flag is set.  wo == null means z = s0 + u + s1 (DCN-style sum head).
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
  float inv_keep, loss_scale;  // loss_scale = 1/(B * world)
  int relu0, relu2;
  int B, N, RT;
};

__global__ __launch_bounds__(256) void tower_head_k(const HeadArgs p) {
  __shared__ float sc[256], sh[256], mu[256], rs[256];
  __shared__ double red[4][2][256];
  __shared__ float redw[4][256];
  __shared__ double hred[4][8];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  for (int c = tid; c < p.N; c += 256) {
    float mean, rstd;
    bn_col_stats(p.fstat_last, p.RT, p.N, c, p.B, mean, rstd);
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
  constexpr int CPL = 4;  // columns per lane (N <= 256)
  double sdy[CPL], sdx[CPL];
  float swd[CPL];
#pragma unroll
  for (int k = 0; k < CPL; ++k) { sdy[k] = 0.0; sdx[k] = 0.0; swd[k] = 0.f; }
  double hp[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const float bd = p.bd[0];
  const float wo0 = p.wo ? p.wo[0] : 1.f, wo1 = p.wo ? p.wo[1] : 1.f, wo2 = p.wo ? p.wo[2] : 1.f;
  const float bo = p.bo ? p.bo[0] : 0.f;
  const float c0 = p.c0 ? p.c0[0] : 0.f;
  for (int rr = 0; rr < 4; ++rr) {
    const int row = blockIdx.x * TM + w * 4 + rr;
    if (row >= p.B) break;  // wave-uniform
    float o[CPL], xh[CPL], mk[CPL];
    float dot = 0.f;
#pragma unroll
    for (int k = 0; k < CPL; ++k) {
      const int c = lane + 64 * k;
      o[k] = 0.f; xh[k] = 0.f; mk[k] = 0.f;
      if (c < p.N) {
        const float a = p.a_last[(size_t)row * p.N + c];
        float v = a * sc[c] + sh[c];
        mk[k] = p.mask ? p.mask[(size_t)row * p.N + c] * p.inv_keep : 1.f;
        v *= mk[k];
        o[k] = v;
        xh[k] = (a - mu[c]) * rs[c];
        dot += v * p.wd[c];
      }
    }
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) dot += __shfl_xor(dot, m);
    const float u = dot + bd;
    const float t2 = (p.relu2 && u <= 0.f) ? 0.f : u;
    const float v0 = p.s0 ? p.s0[row] + c0 : 0.f;
    const float t0 = (p.relu0 && v0 <= 0.f) ? 0.f : v0;
    const float v1 = p.s1 ? p.s1[row] : 0.f;
    const float zz = wo0 * t0 + wo1 * v1 + wo2 * t2 + bo;
    const float y = p.labels[row];
    const float pr = 1.f / (1.f + expf(-zz));
    const float ce = fmaxf(zz, 0.f) - zz * y + log1pf(expf(-fabsf(zz)));
    const float dz = (pr - y) * p.loss_scale;
    const float g0 = (p.relu0 && v0 <= 0.f) ? 0.f : dz * wo0;   // d/d(s0 + c0)
    const float g2 = (p.relu2 && u <= 0.f) ? 0.f : dz * wo2;    // d/du
    if (lane == 0) {
      p.prob[row] = pr;
      if (p.gs0) p.gs0[row] = g0;
      if (p.gs1) p.gs1[row] = dz * wo1;
      hp[0] += (double)ce;
      hp[1] += (double)(dz * t0);
      hp[2] += (double)(dz * v1);
      hp[3] += (double)(dz * t2);
      hp[4] += (double)dz;
      hp[5] += (double)g0;
      hp[6] += (double)g2;
    }
#pragma unroll
    for (int k = 0; k < CPL; ++k) {
      const int c = lane + 64 * k;
      if (c < p.N) {
        const float dyv = g2 * p.wd[c] * mk[k];    // d/d(BN output) after dropout backward
        p.dy_last[(size_t)row * p.N + c] = dyv;
        sdy[k] += (double)dyv;
        sdx[k] += (double)dyv * (double)xh[k];
        swd[k] += o[k] * g2;
      }
    }
  }
#pragma unroll
  for (int k = 0; k < CPL; ++k) {
    const int c = lane + 64 * k;
    red[w][0][c] = sdy[k];
    red[w][1][c] = sdx[k];
    redw[w][c] = swd[k];
  }
  if (lane == 0)
    for (int k = 0; k < 8; ++k) hred[w][k] = hp[k];
  __syncthreads();
  for (int c = tid; c < p.N; c += 256) {
    p.bstat_last[((size_t)blockIdx.x * 2 + 0) * p.N + c] = red[0][0][c] + red[1][0][c] + red[2][0][c] + red[3][0][c];
    p.bstat_last[((size_t)blockIdx.x * 2 + 1) * p.N + c] = red[0][1][c] + red[1][1][c] + red[2][1][c] + red[3][1][c];
    p.dwd_part[(size_t)blockIdx.x * p.N + c] = redw[0][c] + redw[1][c] + redw[2][c] + redw[3][c];
  }
  if (tid < 8) p.hpart[(size_t)blockIdx.x * 8 + tid] = hred[0][tid] + hred[1][tid] + hred[2][tid] + hred[3][tid];
}

// ---------------------------------------------------------------------------------------------
// backward layer l.  da = relu'(a) * BNbwd(dy) is applied on load.  Tile families (blockIdx.x ranges):
//   [0, n_din)              d(input)[B x K] = da . W^T, then dropout backward of layer l-1 and the
//                           partial sums for ITS BN backward (or the final dX for the first layer)
//   [n_din, +n_dw)          dW[K x N] = in'^T . da   (tiled over the output; loops over all B rows)
//   [.., +n_vec)            db = sum_b da, dgamma = sum dy*xhat, dbeta = sum dy   (one lane per column)
//   last block (if head)    reduce the head partials: dWd, dbd, dwo, dbo, dc0, loss
// block = 64.  dyn LDS: 5*N floats (din tiles).
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
  float inv_keep;
  int has_wo;
  int B, K, N, RT;
  int n_din, n_dw, n_vec, ct_k, ct_n;
};

struct ColBwd { float mean, rstd, k1, sdy, sdx; };

__device__ __forceinline__ ColBwd bwd_col(const BwdArgs& p, int c) {
  double s1 = 0.0, s2 = 0.0;
  for (int r = 0; r < p.RT; ++r) {
    s1 += p.bstat[((size_t)r * 2 + 0) * p.N + c];
    s2 += p.bstat[((size_t)r * 2 + 1) * p.N + c];
  }
  ColBwd o;
  o.mean = p.bn[c];
  o.rstd = p.bn[p.N + c];
  o.k1 = p.gamma[c] * o.rstd / (float)p.B;
  o.sdy = (float)s1;
  o.sdx = (float)s2;
  return o;
}
__device__ __forceinline__ float da_of(float a, float dy, const ColBwd& c, float Bf) {
  if (a <= 0.f) return 0.f;
  const float xh = (a - c.mean) * c.rstd;
  return c.k1 * (Bf * dy - c.sdy - xh * c.sdx);
}

__global__ __launch_bounds__(64) void tower_bwd_k(const BwdArgs p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int lane = threadIdx.x;
  const int bid = blockIdx.x;
  const float Bf = (float)p.B;
  const bool first = p.bn_prev == nullptr;
  const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
  if (bid < p.n_din) {
    // ---- d(input) tile: rows rt*16.., input columns kc*16.. ; K' = N ------------------------------
    float* Lm = lds; float* Lr = lds + p.N; float* Lk = lds + 2 * p.N; float* Ls = lds + 3 * p.N; float* Lx = lds + 4 * p.N;
    for (int c = lane; c < p.N; c += 64) {
      const ColBwd cb = bwd_col(p, c);
      Lm[c] = cb.mean; Lr[c] = cb.rstd; Lk[c] = cb.k1; Ls[c] = cb.sdy; Lx[c] = cb.sdx;
    }
    __syncthreads();
    const int kc = bid % p.ct_k, rt = bid / p.ct_k;
    const int i = lane & 15, kq = lane >> 4;
    const int row = rt * TM + i;
    const int kcol = kc * 16 + i;   // B-operand "column" = input feature
    const bool rok = row < p.B, cok = kcol < p.K;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int n0 = 0; n0 < p.N; n0 += 16) {
      const int nn = n0 + 4 * kq;
      float av[4] = {0.f, 0.f, 0.f, 0.f};
      float bv[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int n = nn + t;
        if (n < p.N) {
          if (rok) {
            const float a = p.a[(size_t)row * p.N + n], dy = p.dy[(size_t)row * p.N + n];
            ColBwd cb; cb.mean = Lm[n]; cb.rstd = Lr[n]; cb.k1 = Lk[n]; cb.sdy = Ls[n]; cb.sdx = Lx[n];
            av[t] = da_of(a, dy, cb, Bf);
          }
          if (cok) bv[t] = p.W[(size_t)kcol * p.N + n];
        }
      }
#pragma unroll
      for (int t = 0; t < 4; ++t) acc = mfma16(av[t], bv[t], acc);
    }
    const int ocol = kc * 16 + (lane & 15);
    double s1 = 0.0, s2 = 0.0;
    float pm = 0.f, pr = 0.f;
    if (!first && ocol < p.K) { pm = p.bn_prev[ocol]; pr = p.bn_prev[p.K + ocol]; }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int orow = rt * TM + (lane >> 4) * 4 + r;
      if (orow < p.B && ocol < p.K) {
        float v = acc[r];
        if (!first) {
          if (p.mask_prev) v *= p.mask_prev[(size_t)orow * p.K + ocol] * p.inv_keep;
          const float xh = (p.in[(size_t)orow * p.K + ocol] - pm) * pr;
          s1 += (double)v;
          s2 += (double)v * (double)xh;
        }
        p.dy_prev[(size_t)orow * p.K + ocol] = v;
      }
    }
    if (!first) {
      s1 += __shfl_xor(s1, 16); s2 += __shfl_xor(s2, 16);
      s1 += __shfl_xor(s1, 32); s2 += __shfl_xor(s2, 32);
      if (lane < 16 && ocol < p.K) {
        p.bstat_prev[((size_t)rt * 2 + 0) * p.K + ocol] = s1;
        p.bstat_prev[((size_t)rt * 2 + 1) * p.K + ocol] = s2;
      }
    }
    return;
  }
  if (bid < p.n_din + p.n_dw) {
    // ---- dW tile: rows = input features kf*16.., cols = n*16.. ; K'' = B ---------------------------
    const int t_id = bid - p.n_din;
    const int nt = t_id % p.ct_n, kf = t_id / p.ct_n;
    const int i = lane & 15, kq = lane >> 4;
    const int feat = kf * 16 + i;          // A-operand row (input feature)
    const int ncol = nt * 16 + i;          // B-operand column
    const bool fok = feat < p.K, nok = ncol < p.N;
    ColBwd cb = {0.f, 0.f, 0.f, 0.f, 0.f};
    if (nok) cb = bwd_col(p, ncol);
    float fsc = 1.f, fsh = 0.f;
    if (!first && fok) {
      const float inv = p.bn_prev[p.K + feat] * p.gamma_prev[feat];
      fsc = inv;
      fsh = p.beta_prev[feat] - p.bn_prev[feat] * inv;
    }
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int b0 = 0; b0 < p.B; b0 += 16) {
      float av[4] = {0.f, 0.f, 0.f, 0.f};
      float bv[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int b = b0 + 4 * kq + t;
        if (b < p.B) {
          if (fok) {
            float v = p.in[(size_t)b * p.K + feat];
            if (!first) {
              v = v * fsc + fsh;
              if (p.mask_prev) v *= p.mask_prev[(size_t)b * p.K + feat] * p.inv_keep;
            }
            av[t] = v;
          }
          if (nok) bv[t] = da_of(p.a[(size_t)b * p.N + ncol], p.dy[(size_t)b * p.N + ncol], cb, Bf);
        }
      }
#pragma unroll
      for (int t = 0; t < 4; ++t) acc = mfma16(av[t], bv[t], acc);
    }
    const int ocol = nt * 16 + (lane & 15);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int orow = kf * 16 + (lane >> 4) * 4 + r;
      if (orow < p.K && ocol < p.N) p.dW[(size_t)orow * p.N + ocol] = acc[r];
    }
    return;
  }
  if (bid < p.n_din + p.n_dw + p.n_vec) {
    // ---- per-column vectors ------------------------------------------------------------------------
    const int c = (bid - p.n_din - p.n_dw) * 64 + lane;
    if (c < p.N) {
      const ColBwd cb = bwd_col(p, c);
      float s = 0.f;
      for (int b = 0; b < p.B; ++b) s += da_of(p.a[(size_t)b * p.N + c], p.dy[(size_t)b * p.N + c], cb, Bf);
      p.db[c] = s;
      p.dgamma[c] = cb.sdx;
      p.dbeta[c] = cb.sdy;
    }
    return;
  }
  // ---- head partial reduce (last layer only) ---------------------------------------------------------
  if (p.hpart != nullptr) {
    for (int c = lane; c < p.N; c += 64) {
      float s = 0.f;
      for (int r = 0; r < p.RT; ++r) s += p.dwd_part[(size_t)r * p.N + c];
      p.dwd[c] = s;
    }
    if (lane < 8) {
      double s = 0.0;
      for (int r = 0; r < p.RT; ++r) s += p.hpart[(size_t)r * 8 + lane];
      if (lane == 0) p.loss[0] = (float)(s / (double)p.B);
      if (p.has_wo && lane >= 1 && lane <= 3) p.dwo[lane - 1] = (float)s;
      if (p.has_wo && lane == 4) p.dbo[0] = (float)s;
      if (p.dc0 != nullptr && lane == 5) p.dc0[0] = (float)s;
      if (lane == 6) p.dbd[0] = (float)s;
    }
  }
}

// ------------------------------------------------------------------ C ABI ----------------------------
extern "C" int rsx_tower_fwd_layer(const float* in, const float* W, const float* bias, float* a_out,
                                   double* fstat_out, const double* fstat_prev, const float* gamma_prev,
                                   const float* beta_prev, const float* mask_prev, float* bn_prev_out,
                                   float dropout_rate, int B, int K, int N, rsx_stream_t stream) {
  if (B < 0 || K <= 0 || N <= 0) return RSX_EINVAL;
  if (B == 0) return RSX_OK;
  if (!in || !W || !bias || !a_out) return RSX_EINVAL;
  if (K % 4 != 0) return RSX_EUNSUPPORTED;
  if (fstat_prev != nullptr && (!gamma_prev || !beta_prev || !bn_prev_out)) return RSX_EINVAL;
  if (dropout_rate < 0.f || dropout_rate >= 1.f) return RSX_EINVAL;
  FwdArgs p;
  p.in = in; p.W = W; p.bias = bias; p.a_out = a_out; p.fstat_out = fstat_out;
  p.fstat_prev = fstat_prev; p.gamma_prev = gamma_prev; p.beta_prev = beta_prev; p.mask_prev = mask_prev;
  p.bn_prev_out = bn_prev_out;
  p.inv_keep = 1.0f / (1.0f - dropout_rate);
  p.B = B; p.K = K; p.N = N; p.RT = (B + TM - 1) / TM;
  const dim3 grid((N + 15) / 16, p.RT);
  hipLaunchKernelGGL(tower_fwd_k, grid, dim3(64), (size_t)2 * K * sizeof(float), rsx_s(stream), p);
  RSX_CHECK_LAUNCH();
  return RSX_OK;
}

extern "C" int rsx_tower_head(const float* a_last, const double* fstat_last, const float* gamma, const float* beta,
                              const float* mask, float* bn_out, const float* wd, const float* bd, const float* s0,
                              const float* c0, const float* s1, const float* wo, const float* bo,
                              const float* labels, float* prob, float* dy_last, double* bstat_last,
                              float* dwd_part, double* hpart, float* gs0, float* gs1, float dropout_rate,
                              float loss_scale, int relu0, int relu2, int B, int N, rsx_stream_t stream) {
  if (B < 0 || N <= 0) return RSX_EINVAL;
  if (B == 0) return RSX_OK;
  if (N > 256) return RSX_EUNSUPPORTED;
  if (!a_last || !fstat_last || !gamma || !beta || !bn_out || !wd || !bd || !labels || !prob || !dy_last ||
      !bstat_last || !dwd_part || !hpart)
    return RSX_EINVAL;
  HeadArgs p;
  p.a_last = a_last; p.fstat_last = fstat_last; p.gamma = gamma; p.beta = beta; p.mask = mask; p.bn_out = bn_out;
  p.wd = wd; p.bd = bd; p.s0 = s0; p.c0 = c0; p.s1 = s1; p.wo = wo; p.bo = bo; p.labels = labels; p.prob = prob;
  p.dy_last = dy_last; p.bstat_last = bstat_last; p.dwd_part = dwd_part; p.hpart = hpart; p.gs0 = gs0; p.gs1 = gs1;
  p.inv_keep = 1.0f / (1.0f - dropout_rate);
  p.loss_scale = loss_scale;
  p.relu0 = relu0; p.relu2 = relu2;
  p.B = B; p.N = N; p.RT = (B + TM - 1) / TM;
  hipLaunchKernelGGL(tower_head_k, dim3(p.RT), dim3(256), 0, rsx_s(stream), p);
  RSX_CHECK_LAUNCH();
  return RSX_OK;
}

extern "C" int rsx_tower_bwd_layer(const float* in, const float* W, const float* a, const float* dy,
                                   const double* bstat, const float* bn, const float* gamma, float* dW, float* db,
                                   float* dgamma, float* dbeta, const float* bn_prev, const float* gamma_prev,
                                   const float* beta_prev, const float* mask_prev, float* dy_prev,
                                   double* bstat_prev, const double* hpart, const float* dwd_part, float* dwd,
                                   float* dbd, float* dwo, float* dbo, float* dc0, float* loss,
                                   float dropout_rate, int B, int K, int N, rsx_stream_t stream) {
  if (B < 0 || K <= 0 || N <= 0) return RSX_EINVAL;
  if (B == 0) return RSX_OK;
  if (!in || !W || !a || !dy || !bstat || !bn || !gamma || !dW || !db || !dgamma || !dbeta || !dy_prev)
    return RSX_EINVAL;
  if (bn_prev != nullptr && (!gamma_prev || !beta_prev || !bstat_prev)) return RSX_EINVAL;
  if (hpart != nullptr && (!dwd_part || !dwd || !dbd || !loss)) return RSX_EINVAL;
  BwdArgs p;
  p.in = in; p.W = W; p.a = a; p.dy = dy; p.bstat = bstat; p.bn = bn; p.gamma = gamma;
  p.dW = dW; p.db = db; p.dgamma = dgamma; p.dbeta = dbeta;
  p.bn_prev = bn_prev; p.gamma_prev = gamma_prev; p.beta_prev = beta_prev; p.mask_prev = mask_prev;
  p.dy_prev = dy_prev; p.bstat_prev = bstat_prev;
  p.hpart = hpart; p.dwd_part = dwd_part; p.dwd = dwd; p.dbd = dbd; p.dwo = dwo; p.dbo = dbo; p.dc0 = dc0;
  p.loss = loss;
  p.inv_keep = 1.0f / (1.0f - dropout_rate);
  p.has_wo = dwo != nullptr;
  p.B = B; p.K = K; p.N = N; p.RT = (B + TM - 1) / TM;
  p.ct_k = (K + 15) / 16;
  p.ct_n = (N + 15) / 16;
  p.n_din = p.ct_k * p.RT;
  p.n_dw = p.ct_k * p.ct_n;
  p.n_vec = (N + 63) / 64;
  const int total = p.n_din + p.n_dw + p.n_vec + (hpart != nullptr ? 1 : 0);
  hipLaunchKernelGGL(tower_bwd_k, dim3(total), dim3(64), (size_t)5 * N * sizeof(float), rsx_s(stream), p);
  RSX_CHECK_LAUNCH();
  return RSX_OK;
}
