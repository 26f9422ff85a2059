// xDeepFM Compressed Interaction Network layer on gfx950 (fp32 MFMA), forward and backward.
// Reference call site: xdeepfm/xdeepfm.py:145-172 -- Z[b,d,f*H+h] = X0[b,f,d]*Xk[b,h,d] (split + batched matmul +
// reshape + transpose), X^{k+1}[b,n,d] = relu(sum_j Z[b,d,j] W[j,n] + c[n]) (1x1 conv1d).  SURVEY.md 8a row a-8:
// the only genuinely GEMM-shaped op of the path (M = B*D rows, K = F*H, N outputs; 6.8 GFLOP fwd at [128,128], B=256).
//
// Formulation used here (never materialises Z, 25-82 MB in TF): with m = (b, d)
//     pre[m, n] = sum_f X0[m, f] * T_f[m, n],   T_f = Xk[m, :] . W_f[:, n]        (W_f = rows f*H .. f*H+H-1 of W)
// i.e. a grouped GEMM whose A operand (Xk) is shared by all F groups and whose per-group results are row-scaled by
// X0[:, f].  Because D = 16, the 16 rows of one MFMA tile are exactly the 16 embedding dims of ONE example, so
// A[i=d][k=h] = Xk[b][h][d] is a unit-stride 64 B read and the output tile out[b][n0..n0+15][0..15] is one contiguous
// 1 KiB store.  fp32 in / fp32 accumulate (v_mfma_f32_16x16x4_f32, an exact fmaf chain) keeps the 1e-5 parity bar;
// a bf16 path is a later-round item.
//   cin_fwd_k     wave = (2 examples) x (16 outputs); Xk, X0 tiles of the examples staged in LDS
//   cin_bwd_dx_k  workgroup = (2 examples) x all H-tiles; U_f = dpre . W_f^T, dXk += X0_f * U_f, dX0_f = <Xk, U_f>
//   cin_bwd_dw_k  wave = (3 fields) x (16 h) x (16 n); dW = Z^T . dpre with Z generated on load; ones-row -> dc
// All reductions are in fixed order: deterministic, no atomics.
#include "rsx_common.h"
#include "adam_device.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 cin_mfma(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}
constexpr int CIN_D = 16;
#ifndef RSX_CIN_BT
#define RSX_CIN_BT 1
#endif
constexpr int CIN_BT = RSX_CIN_BT;   // examples per workgroup: 1 doubles the waves in flight (B = 256 gives few tiles); 2 was 20-25 % slower

// ----------------------------------------------------------------------------------------------- forward
struct CinFwdArgs {
  const float* X0;   // [B, F, 16]
  const float* Xk;   // [B, H, 16]
  const float* W;    // [F*H, N]
  const float* c;    // [N]
  float* out;        // [B, N, 16]
  int B, F, H, N;
  int nby;           // tile rows of the grid; rows >= nby carry the optimizer sweep slice
  AdamSlice sweep;
};

// grid = (ceil(N/16), ceil(B/4)), block = 256: wave w owns example 4*blockIdx.y + w and the 16 outputs n0.. of the
// workgroup.  The B operand is staged through LDS: per field f the workgroup copies the [H, 16] slice of W_f ONCE
// (coalesced float4 global loads, double-buffered: the loads of f+1 are in flight during the MFMAs of f) and its 4 waves
// read it back transposed, so one L2 read feeds 4 MFMAs instead of 1 and a lane fetches the operands of 4 k-steps with a
// single ds_read_b128.  k-permutation: k-step 4*ks + t uses h = 16*ks + 4*(lane>>4) + t on both operands.
// The A operand Xk[b][h][d] does not depend on f: HSMAX (= k-steps, a multiple of 4: 12 covers H <= 48, 32 covers
// H <= 128) values per lane stay in registers for the whole kernel.
// dyn LDS: 2 * 16 * (4*HSMAX + 4) + 4 * F * 16 floats.
template <int HSMAX>
__global__ __launch_bounds__(256) void cin_fwd_k(const CinFwdArgs p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int HP = 4 * HSMAX + 4;              // padded h-stride of the transposed tile
  constexpr int R = (HSMAX * 16 + 255) / 256;    // float4 per thread per tile
  float* sW = lds;                               // [2][16][HP]
  float* sX0 = lds + 2 * 16 * HP;                // [4][F*16]
  if ((int)blockIdx.y >= p.nby) {   // piggy-backed optimizer sweep (untouched rows): streaming beside the MFMA tiles
    const uint32_t lin = ((uint32_t)blockIdx.y - (uint32_t)p.nby) * gridDim.x + blockIdx.x;
    if (lin < p.sweep.n_blk) adam_block(p.sweep.args, p.sweep.blk_lo + lin);
    return;
  }
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int i = lane & 15, kq = lane >> 4;
  const int n0 = blockIdx.x * 16;
  const int b = blockIdx.y * 4 + wv;
  const bool bok = b < p.B, nok = n0 + i < p.N;
  for (int e = tid; e < 2 * 16 * HP; e += 256) sW[e] = 0.f;          // rows h >= H must read as zero
  for (int e = tid; e < 4 * p.F * 4; e += 256) {
    const int ex = e / (p.F * 4), r = e - ex * (p.F * 4);
    const int bb = blockIdx.y * 4 + ex;
    reinterpret_cast<float4*>(sX0)[e] = bb < p.B ? reinterpret_cast<const float4*>(p.X0 + (size_t)bb * p.F * CIN_D)[r] : F4Z;
  }
  float areg[HSMAX];
  {
    // unconditional on clamped indices, masked by multiplication: HSMAX guarded loads would be HSMAX branches, i.e. as
    // many dependent memory round trips before the first MFMA
    const float* xk = p.Xk + (size_t)(bok ? b : p.B - 1) * p.H * CIN_D + i;
#pragma unroll
    for (int s_ = 0; s_ < HSMAX; ++s_) {
      const int h = 16 * (s_ >> 2) + 4 * kq + (s_ & 3);
      areg[s_] = xk[(size_t)(h < p.H ? h : p.H - 1) * CIN_D] * ((h < p.H && bok) ? 1.f : 0.f);
    }
  }
  float4 wreg[R];
  auto load_tile = [&](int f) {
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int q = tid + 256 * r, h = q >> 2, j = q & 3;
      const int nn = n0 + 4 * j;
      if ((p.N & 3) == 0) {                      // aligned rows: one float4
        const bool ok = h < p.H && nn < p.N;
        wreg[r] = *reinterpret_cast<const float4*>(p.W + ((size_t)f * p.H + (ok ? h : 0)) * p.N + (ok ? nn : 0));
        if (!ok) wreg[r] = F4Z;
      } else {                                   // odd widths (tests, tiny models): element-wise
        const float* w = p.W + ((size_t)f * p.H + (h < p.H ? h : 0)) * p.N;
        const bool hok = h < p.H;
        wreg[r].x = (hok && nn + 0 < p.N) ? w[nn + 0] : 0.f;
        wreg[r].y = (hok && nn + 1 < p.N) ? w[nn + 1] : 0.f;
        wreg[r].z = (hok && nn + 2 < p.N) ? w[nn + 2] : 0.f;
        wreg[r].w = (hok && nn + 3 < p.N) ? w[nn + 3] : 0.f;
      }
    }
  };
  auto store_tile = [&](int buf) {
    float* t = sW + buf * 16 * HP;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int q = tid + 256 * r, h = q >> 2, j = q & 3;
      if (h < p.H) {
        t[(4 * j + 0) * HP + h] = wreg[r].x;
        t[(4 * j + 1) * HP + h] = wreg[r].y;
        t[(4 * j + 2) * HP + h] = wreg[r].z;
        t[(4 * j + 3) * HP + h] = wreg[r].w;
      }
    }
  };
  load_tile(0);
  __syncthreads();                               // zero fill done
  store_tile(0);
  __syncthreads();
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  const int nks = (p.H + 15) >> 4;
  for (int f = 0; f < p.F; ++f) {
    if (f + 1 < p.F) load_tile(f + 1);
    const float* t = sW + (f & 1) * 16 * HP + i * HP + 4 * kq;
    f32x4 T = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < HSMAX / 4; ++ks) {
      if (ks < nks) {                            // wave-uniform
        const float4 bq = *reinterpret_cast<const float4*>(t + 16 * ks);
        T = cin_mfma(areg[4 * ks + 0], bq.x, T);
        T = cin_mfma(areg[4 * ks + 1], bq.y, T);
        T = cin_mfma(areg[4 * ks + 2], bq.z, T);
        T = cin_mfma(areg[4 * ks + 3], bq.w, T);
      }
    }
    const float4 x = *reinterpret_cast<const float4*>(sX0 + wv * p.F * CIN_D + f * CIN_D + kq * 4);  // rows d = 4*kq + r
    acc[0] += x.x * T[0];
    acc[1] += x.y * T[1];
    acc[2] += x.z * T[2];
    acc[3] += x.w * T[3];
    if (f + 1 < p.F) store_tile((f + 1) & 1);
    __syncthreads();
  }
  if (nok && bok) {
    const float cv = p.c[n0 + i];
    float4 o;
    o.x = fmaxf(acc[0] + cv, 0.f);
    o.y = fmaxf(acc[1] + cv, 0.f);
    o.z = fmaxf(acc[2] + cv, 0.f);
    o.w = fmaxf(acc[3] + cv, 0.f);
    *reinterpret_cast<float4*>(p.out + ((size_t)b * p.N + n0 + i) * CIN_D + kq * 4) = o;
  }
}

// ----------------------------------------------------------------------------------------------- backward: dXk, dX0
struct CinBwdDxArgs {
  const float* X0;    // [B, F, 16]
  const float* Xk;    // [B, H, 16]
  const float* W;     // [F*H, N]
  const float* out;   // [B, N, 16] this layer's relu output
  const float* dout;  // [B, N, 16] gradient wrt the relu output (nullable when gs is given)
  const float* gs;    // [B]  nullable: the direct-connect gradient gs[b] * wout[n], broadcast over d, is added to dout
  const float* wout;  // [N]
  float* dXk;         // [B, H, 16]
  float* dX0;         // [B, F, 16]
  float* dpre;        // [B, N, 16] out: dout with the relu mask applied, consumed by cin_bwd_dw_k
  int acc_dxk, acc_dx0;
  int B, F, H, N;
  int HT;             // 16-wide h tiles; the block holds HT * FS waves: for a short H the fields of a tile are dealt
                      // round-robin to FS = 2 or 4 waves so that the workgroup still loads its CU's 4 SIMDs evenly
};

// grid = ceil(B/BT), block = 64 * HT * FS (<= 12 waves).
// dyn LDS: 2*(N + F + H)*16 + HT*2*F*16 floats.  The A operand dpre[b][n][d] does not depend on f: after staging it
// through LDS (the relu mask is applied once) each lane keeps its 4*NSMAX values per example in registers
// (NSMAX = 8 covers N <= 128), so the f loop is "4 float4 W loads in flight -> 32 MFMAs".
template <int NSMAX>
__global__ __launch_bounds__(768) void cin_bwd_dx_k(const CinBwdDxArgs p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int HT = p.HT, FS = (int)(blockDim.x >> 6) / HT;
  float* sDp = lds;                                  // [BT][N*16] dpre
  float* sX0 = sDp + CIN_BT * p.N * CIN_D;           // [BT][F*16]
  float* sXk = sX0 + CIN_BT * p.F * CIN_D;           // [BT][H*16]
  float* sP = sXk + CIN_BT * p.H * CIN_D;            // [HT][BT][F*16] dX0 partials
  float* sDx = sP + HT * CIN_BT * p.F * CIN_D;       // [FS][HT][BT][16*16] dXk tiles
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ht = wave % HT, part = wave / HT;
  const int b0 = blockIdx.x * CIN_BT;
  for (int bt = 0; bt < CIN_BT; ++bt) {
    const int b = b0 + bt;
    for (int e = tid; e < p.N * 4; e += blockDim.x) {
      float4 v = F4Z;
      if (b < p.B) {
        const float4 o = reinterpret_cast<const float4*>(p.out + (size_t)b * p.N * CIN_D)[e];
        float4 g = p.dout ? reinterpret_cast<const float4*>(p.dout + (size_t)b * p.N * CIN_D)[e] : F4Z;
        if (p.gs) {
          const float a = p.gs[b] * p.wout[e >> 2];
          g = make_float4(g.x + a, g.y + a, g.z + a, g.w + a);
        }
        v = make_float4(o.x > 0.f ? g.x : 0.f, o.y > 0.f ? g.y : 0.f, o.z > 0.f ? g.z : 0.f, o.w > 0.f ? g.w : 0.f);
        reinterpret_cast<float4*>(p.dpre + (size_t)b * p.N * CIN_D)[e] = v;
      }
      reinterpret_cast<float4*>(sDp + bt * p.N * CIN_D)[e] = v;
    }
    for (int e = tid; e < p.F * 4; e += blockDim.x)
      reinterpret_cast<float4*>(sX0 + bt * p.F * CIN_D)[e] =
          b < p.B ? reinterpret_cast<const float4*>(p.X0 + (size_t)b * p.F * CIN_D)[e] : F4Z;
    for (int e = tid; e < p.H * 4; e += blockDim.x)
      reinterpret_cast<float4*>(sXk + bt * p.H * CIN_D)[e] =
          b < p.B ? reinterpret_cast<const float4*>(p.Xk + (size_t)b * p.H * CIN_D)[e] : F4Z;
  }
  __syncthreads();
  const int i = lane & 15, kq = lane >> 4;
  const int h = ht * 16 + i;                // A-operand row = h (W rows); the dpre operand's column = d = i
  const bool hok = h < p.H;
  const int ns = (p.N + 15) >> 4;
  // U_f^T tile in C layout: lane (i, kq) holds U_f[h = 16 ht + 4 kq + r][d = i], r = 0..3 -- the <Xk, U_f> reduction over
  // h is then 4 in-lane FMAs + 2 cross-lane steps (over kq) for ONE value, not 4 steps for 4 values
  f32x4 dxk[CIN_BT];
  float xkv[CIN_BT][4];                     // Xk[bt][h = 16 ht + 4 kq + r][d = i], zero for h >= H
  float areg[CIN_BT][NSMAX][4];             // dpre[bt][n = 16 s + 4 kq + t][d = i]
#pragma unroll
  for (int bt = 0; bt < CIN_BT; ++bt) {
    dxk[bt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int hr = ht * 16 + 4 * kq + r;
      xkv[bt][r] = hr < p.H ? sXk[bt * p.H * CIN_D + hr * CIN_D + i] : 0.f;
    }
#pragma unroll
    for (int s_ = 0; s_ < NSMAX; ++s_)
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int n = 16 * s_ + 4 * kq + t;
        areg[bt][s_][t] = n < p.N ? sDp[bt * p.N * CIN_D + n * CIN_D + i] : 0.f;
      }
  }
  const bool vec_ok = (p.N & 3) == 0;
  // W rows of field f for this lane: NSMAX float4 (n = 16 s + 4 kq ..).  Unconditional loads on clamped addresses when
  // N % 4 == 0: a row h >= H only feeds a column that is never stored, and k-steps with n >= N multiply A operands that
  // are zero (areg is zero-padded) -- no mask needed, no branch per load.
  auto load_w = [&](int f, float4* bw) {
    const float* Wr = p.W + ((size_t)(f < p.F ? f : p.F - 1) * p.H + (hok ? h : 0)) * p.N;
    if (vec_ok) {
#pragma unroll
      for (int u = 0; u < NSMAX; ++u) {
        const int nn = 16 * u + 4 * kq;
        bw[u] = *reinterpret_cast<const float4*>(Wr + (nn + 3 < p.N ? nn : p.N - 4));
      }
    } else {
#pragma unroll
      for (int u = 0; u < NSMAX; ++u) {
        const int nn = 16 * u + 4 * kq;
        float4 t = F4Z;
        if (hok && nn < p.N) {
          t.x = Wr[nn];
          t.y = nn + 1 < p.N ? Wr[nn + 1] : 0.f;
          t.z = nn + 2 < p.N ? Wr[nn + 2] : 0.f;
          t.w = nn + 3 < p.N ? Wr[nn + 3] : 0.f;
        }
        bw[u] = t;
      }
    }
  };
  // one field: U_f = dpre . W_f^T (4 * NSMAX MFMAs on the rows already in registers), then dXk += X0_f * U_f and the
  // dX0_f partial of this wave's 16 h
  auto field = [&](int f, const float4* bw) {
    f32x4 U[CIN_BT];
#pragma unroll
    for (int bt = 0; bt < CIN_BT; ++bt) U[bt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < NSMAX; ++u) {
      if (u < ns) {                               // wave-uniform
#pragma unroll
        for (int bt = 0; bt < CIN_BT; ++bt) {
          U[bt] = cin_mfma(bw[u].x, areg[bt][u][0], U[bt]);
          U[bt] = cin_mfma(bw[u].y, areg[bt][u][1], U[bt]);
          U[bt] = cin_mfma(bw[u].z, areg[bt][u][2], U[bt]);
          U[bt] = cin_mfma(bw[u].w, areg[bt][u][3], U[bt]);
        }
      }
    }
#pragma unroll
    for (int bt = 0; bt < CIN_BT; ++bt) {
      const float x = sX0[bt * p.F * CIN_D + f * CIN_D + i];
      dxk[bt][0] += x * U[bt][0];
      dxk[bt][1] += x * U[bt][1];
      dxk[bt][2] += x * U[bt][2];
      dxk[bt][3] += x * U[bt][3];
      // dX0[bt][f][d = i] partial over this wave's 16 h
      float q = ((U[bt][0] * xkv[bt][0] + U[bt][1] * xkv[bt][1]) + U[bt][2] * xkv[bt][2]) + U[bt][3] * xkv[bt][3];
      q += __shfl_xor(q, 16);
      q += __shfl_xor(q, 32);
      if (kq == 0) sP[((ht * CIN_BT + bt) * p.F + f) * CIN_D + i] = q;
    }
  };
  // software pipeline over the fields: the W rows of the next field load while this field's MFMAs run (the workgroup
  // is alone on its CU, so the second register set is free)
  float4 wa[NSMAX], wb[NSMAX];
  load_w(part, wa);
  for (int f = part; f < p.F; f += 2 * FS) {
    load_w(f + FS, wb);
    field(f, wa);
    if (f + FS < p.F) {                           // wave-uniform
      load_w(f + 2 * FS, wa);
      field(f + FS, wb);
    }
  }
  // dXk tiles -> LDS as [part][ht][bt][h_local][d], then float4 rows out (parts added in order)
#pragma unroll
  for (int bt = 0; bt < CIN_BT; ++bt)
#pragma unroll
    for (int r = 0; r < 4; ++r) sDx[(((part * HT + ht) * CIN_BT + bt) * 16 + 4 * kq + r) * CIN_D + i] = dxk[bt][r];
  __syncthreads();
  for (int e = tid; e < HT * CIN_BT * 64; e += blockDim.x) {
    const int t = e >> 6, hl = (e & 63) >> 2, dq = e & 3;      // t = ht * BT + bt
    const int bt = t % CIN_BT, hh = (t / CIN_BT) * 16 + hl, b = b0 + bt;
    if (hh < p.H && b < p.B) {
      float4 o = *reinterpret_cast<const float4*>(sDx + (t * 16 + hl) * CIN_D + dq * 4);
      for (int pp = 1; pp < FS; ++pp)     // field parts in order
        o = f4_add(o, *reinterpret_cast<const float4*>(sDx + ((pp * HT * CIN_BT + t) * 16 + hl) * CIN_D + dq * 4));
      float4* dst = reinterpret_cast<float4*>(p.dXk + ((size_t)b * p.H + hh) * CIN_D + dq * 4);
      if (p.acc_dxk) o = f4_add(*dst, o);
      *dst = o;
    }
  }
  if (p.dXk == p.dX0) __syncthreads();   // first layer, one gradient buffer for both roles of X0: dXk lands before dX0 adds
  for (int e = tid; e < CIN_BT * p.F * 4; e += blockDim.x) {   // sum the HT partials in wave order
    const int bt = e / (p.F * 4), r = e - bt * p.F * 4;
    const int b = b0 + bt;
    if (b >= p.B) continue;
    float4 s = F4Z;
    for (int w = 0; w < HT; ++w) s = f4_add(s, reinterpret_cast<const float4*>(sP + (w * CIN_BT + bt) * p.F * CIN_D)[r]);
    float4* dst = reinterpret_cast<float4*>(p.dX0 + (size_t)b * p.F * CIN_D) + r;
    if (p.acc_dx0) s = f4_add(*dst, s);
    *dst = s;
  }
}

// ----------------------------------------------------------------------------------------------- backward: dXk, dX0 (tiled)
// The same math as cin_bwd_dx_k with the weight slice SHARED between examples: grid = (ceil(H/16), ceil(B/EB)), block =
// 64 * EB * FS.  Wave (e, part) owns example b0 + e and the fields f = part (mod FS); per step the workgroup copies the FS
// slices W_f[16 h][N] once (coalesced float4, double-buffered through registers while the previous step's MFMAs run) and
// its waves read them back as MFMA A operands with ds_read_b128 -- 1/EB of the L2 -> L1 traffic of cin_bwd_dx_k, which is
// bound by exactly that traffic (every workgroup there streams the whole W).  The <Xk, U_f> partial of this h tile goes
// to part[ht][b][f][d]; cin_dx0_reduce_k adds the tiles in order.  Requires N % 4 == 0.
template <int NSMAX, int EB, int FS>
__global__ __launch_bounds__(64 * EB * FS) void cin_bwd_dx2_k(const CinBwdDxArgs p, float* __restrict__ part_ws) {
  constexpr int NW = EB * FS, NTHR = 64 * NW;
  constexpr int RMAX = (FS * 16 * 32 + NTHR - 1) / NTHR;     // float4 per thread per step at N = 128
  static_assert(RMAX <= 4, "copy slots");
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int NP = p.N + 4, N4 = p.N >> 2;
  float* sW = lds;                              // [2][FS][16][NP]      (aliases sDp: dpre is consumed before the loop)
  float* sDp = lds;                             // [EB][N*16]
  const int r0 = 2 * FS * 16 * NP, r1 = EB * p.N * CIN_D;
  float* sX0 = lds + (r0 > r1 ? r0 : r1);       // [EB][F*16]
  float* sXk = sX0 + EB * p.F * CIN_D;          // [EB][16*16]  this h tile
  float* sDx = sXk + EB * 256;                  // [FS][EB][256] dXk tiles
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int e = wave % EB, part = wave / EB;
  const int ht = blockIdx.x, b0 = blockIdx.y * EB, b = b0 + e;
  for (int bt = 0; bt < EB; ++bt) {
    const int bb = b0 + bt;
    for (int e4 = tid; e4 < p.N * 4; e4 += NTHR) {
      float4 v = F4Z;
      if (bb < p.B) {
        const float4 o = reinterpret_cast<const float4*>(p.out + (size_t)bb * p.N * CIN_D)[e4];
        float4 g = p.dout ? reinterpret_cast<const float4*>(p.dout + (size_t)bb * p.N * CIN_D)[e4] : F4Z;
        if (p.gs) {
          const float a = p.gs[bb] * p.wout[e4 >> 2];
          g = make_float4(g.x + a, g.y + a, g.z + a, g.w + a);
        }
        v = make_float4(o.x > 0.f ? g.x : 0.f, o.y > 0.f ? g.y : 0.f, o.z > 0.f ? g.z : 0.f, o.w > 0.f ? g.w : 0.f);
        if (ht == 0) reinterpret_cast<float4*>(p.dpre + (size_t)bb * p.N * CIN_D)[e4] = v;
      }
      reinterpret_cast<float4*>(sDp + bt * p.N * CIN_D)[e4] = v;
    }
    for (int e4 = tid; e4 < p.F * 4; e4 += NTHR)
      reinterpret_cast<float4*>(sX0 + bt * p.F * CIN_D)[e4] =
          bb < p.B ? reinterpret_cast<const float4*>(p.X0 + (size_t)bb * p.F * CIN_D)[e4] : F4Z;
    for (int e4 = tid; e4 < 64; e4 += NTHR) {
      const int hh = ht * 16 + (e4 >> 2);
      reinterpret_cast<float4*>(sXk + bt * 256)[e4] =
          (bb < p.B && hh < p.H) ? reinterpret_cast<const float4*>(p.Xk + ((size_t)bb * p.H + hh) * CIN_D)[e4 & 3] : F4Z;
    }
  }
  __syncthreads();
  const int i = lane & 15, kq = lane >> 4;
  const int ns = (p.N + 15) >> 4;
  f32x4 dxk = {0.f, 0.f, 0.f, 0.f};
  float xkv[4];                             // Xk[b][h = 16 ht + 4 kq + r][d = i] (zero rows for h >= H)
  float areg[NSMAX][4];                     // dpre[b][n = 16 s + 4 kq + t][d = i]
#pragma unroll
  for (int r = 0; r < 4; ++r) xkv[r] = sXk[e * 256 + (4 * kq + r) * CIN_D + i];
#pragma unroll
  for (int s_ = 0; s_ < NSMAX; ++s_)
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int n = 16 * s_ + 4 * kq + t;
      areg[s_][t] = n < p.N ? sDp[e * p.N * CIN_D + n * CIN_D + i] : 0.f;
    }
  __syncthreads();                          // sDp is dead: its space becomes the W double buffer
  const int steps = (p.F + FS - 1) / FS;
  const int slice4 = FS * 16 * N4;          // float4 per step
  // per-thread copy slots (fixed over the steps): field part, offset of the float4 inside a field's [H][N] block, LDS offset
  int cp_pf[RMAX], cp_g[RMAX], cp_l[RMAX];
#pragma unroll
  for (int k = 0; k < RMAX; ++k) {
    const int idx0 = tid + k * NTHR;
    const int idx = idx0 < slice4 ? idx0 : slice4 - 1;
    const int pf = idx / (16 * N4), rem = idx - pf * 16 * N4;
    const int r = rem / N4, c4 = rem - r * N4;
    const int hh = ht * 16 + r < p.H ? ht * 16 + r : p.H - 1;
    cp_pf[k] = pf;
    cp_g[k] = hh * N4 + c4;
    cp_l[k] = idx0 < slice4 ? (pf * 16 + r) * NP + c4 * 4 : -1;
  }
  const float4* W4 = reinterpret_cast<const float4*>(p.W);
  const size_t fstride4 = (size_t)p.H * N4;
  // the double buffer starts as zeros: the row pads and the columns past N that the fixed NSMAX reads touch must be finite
  for (int e4 = tid; e4 < (2 * FS * 16 * NP) / 4; e4 += NTHR) reinterpret_cast<float4*>(sW)[e4] = F4Z;
  __syncthreads();
#pragma unroll
  for (int k = 0; k < RMAX; ++k) {
    const int f = cp_pf[k] < p.F ? cp_pf[k] : p.F - 1;
    const float4 v = W4[(size_t)f * fstride4 + cp_g[k]];
    if (cp_l[k] >= 0) *reinterpret_cast<float4*>(sW + cp_l[k]) = v;
  }
  __syncthreads();
  for (int s_ = 0; s_ < steps; ++s_) {
    // next step's slices: global loads in flight during this step's MFMAs (clamped step: the last one reloads itself)
    const int sn = s_ + 1 < steps ? s_ + 1 : s_;
    float4 st0 = F4Z, st1 = F4Z, st2 = F4Z, st3 = F4Z;
    {
      const int f0_ = sn * FS + cp_pf[0] < p.F ? sn * FS + cp_pf[0] : p.F - 1;
      st0 = W4[(size_t)f0_ * fstride4 + cp_g[0]];
      if (RMAX > 1) { const int f1_ = sn * FS + cp_pf[RMAX > 1 ? 1 : 0] < p.F ? sn * FS + cp_pf[RMAX > 1 ? 1 : 0] : p.F - 1; st1 = W4[(size_t)f1_ * fstride4 + cp_g[RMAX > 1 ? 1 : 0]]; }
      if (RMAX > 2) { const int f2_ = sn * FS + cp_pf[RMAX > 2 ? 2 : 0] < p.F ? sn * FS + cp_pf[RMAX > 2 ? 2 : 0] : p.F - 1; st2 = W4[(size_t)f2_ * fstride4 + cp_g[RMAX > 2 ? 2 : 0]]; }
      if (RMAX > 3) { const int f3_ = sn * FS + cp_pf[RMAX > 3 ? 3 : 0] < p.F ? sn * FS + cp_pf[RMAX > 3 ? 3 : 0] : p.F - 1; st3 = W4[(size_t)f3_ * fstride4 + cp_g[RMAX > 3 ? 3 : 0]]; }
    }
    const int f = s_ * FS + part;
    if (f < p.F) {                          // wave-uniform
      const float* wr = sW + (((s_ & 1) * FS + part) * 16 + i) * NP + 4 * kq;
      float4 bw[NSMAX];
#pragma unroll
      for (int u = 0; u < NSMAX; ++u) bw[u] = *reinterpret_cast<const float4*>(wr + 16 * u);   // all reads up front
      f32x4 U = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int u = 0; u < NSMAX; ++u) {
        if (u < ns) {                       // uniform
          U = cin_mfma(bw[u].x, areg[u][0], U);
          U = cin_mfma(bw[u].y, areg[u][1], U);
          U = cin_mfma(bw[u].z, areg[u][2], U);
          U = cin_mfma(bw[u].w, areg[u][3], U);
        }
      }
      const float x = sX0[e * p.F * CIN_D + f * CIN_D + i];
      dxk[0] += x * U[0];
      dxk[1] += x * U[1];
      dxk[2] += x * U[2];
      dxk[3] += x * U[3];
      float q = ((U[0] * xkv[0] + U[1] * xkv[1]) + U[2] * xkv[2]) + U[3] * xkv[3];
      q += __shfl_xor(q, 16);
      q += __shfl_xor(q, 32);
      if (kq == 0 && b < p.B) part_ws[(((size_t)ht * p.B + b) * p.F + f) * CIN_D + i] = q;
    }
    float* dstb = sW + ((s_ + 1) & 1) * FS * 16 * NP;
    if (cp_l[0] >= 0) *reinterpret_cast<float4*>(dstb + cp_l[0]) = st0;
    if (RMAX > 1 && cp_l[RMAX > 1 ? 1 : 0] >= 0) *reinterpret_cast<float4*>(dstb + cp_l[RMAX > 1 ? 1 : 0]) = st1;
    if (RMAX > 2 && cp_l[RMAX > 2 ? 2 : 0] >= 0) *reinterpret_cast<float4*>(dstb + cp_l[RMAX > 2 ? 2 : 0]) = st2;
    if (RMAX > 3 && cp_l[RMAX > 3 ? 3 : 0] >= 0) *reinterpret_cast<float4*>(dstb + cp_l[RMAX > 3 ? 3 : 0]) = st3;
    __syncthreads();
  }
  // dXk tiles -> LDS as [part][e][h_local][d], then float4 rows out (parts added in order)
#pragma unroll
  for (int r = 0; r < 4; ++r) sDx[((part * EB + e) * 16 + 4 * kq + r) * CIN_D + i] = dxk[r];
  __syncthreads();
  for (int e4 = tid; e4 < EB * 64; e4 += NTHR) {
    const int ex = e4 >> 6, hl = (e4 & 63) >> 2, dq = e4 & 3;
    const int hh = ht * 16 + hl, bb = b0 + ex;
    if (hh < p.H && bb < p.B) {
      float4 o = *reinterpret_cast<const float4*>(sDx + (ex * 16 + hl) * CIN_D + dq * 4);
#pragma unroll
      for (int pp = 1; pp < FS; ++pp)
        o = f4_add(o, *reinterpret_cast<const float4*>(sDx + ((pp * EB + ex) * 16 + hl) * CIN_D + dq * 4));
      float4* dst = reinterpret_cast<float4*>(p.dXk + ((size_t)bb * p.H + hh) * CIN_D + dq * 4);
      if (p.acc_dxk) o = f4_add(*dst, o);
      *dst = o;
    }
  }
}

// dX0[b][f][d] (=|+=) sum over the h tiles, in tile order
__global__ __launch_bounds__(256) void cin_dx0_reduce_k(const float* __restrict__ part_ws, float* __restrict__ dX0, int HT,
                                                        long long n4, int acc) {
  const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
  if (e >= n4) return;
  float4 s = reinterpret_cast<const float4*>(part_ws)[e];
  for (int t = 1; t < HT; ++t) s = f4_add(s, reinterpret_cast<const float4*>(part_ws)[(long long)t * n4 + e]);
  float4* dst = reinterpret_cast<float4*>(dX0) + e;
  if (acc) s = f4_add(*dst, s);
  *dst = s;
}

// ----------------------------------------------------------------------------------------------- backward: dW, dc
struct CinBwdDwArgs {
  const float* X0; const float* Xk; const float* dpre;   // dpre [B, N, 16] = relu-masked dout (written by cin_bwd_dx_k)
  float* dW;   // [F*H, N]
  float* dc;   // [N]
  int B, F, H, N, FG;   // FG = ceil(F/FT) field groups
  AdamSlice sweep;      // optional slice of the untouched-row optimizer sweep: extra z-planes of the grid (the MFMA-bound
                        // tiles leave HBM idle; the sweep is pure streaming)
};

// grid = (ceil(N/(16 NT)), ceil(H/16), FG), block = 64 NW.  A wave owns the (CIN_FT fields) x (16 h) x (NT 16-wide n tiles)
// output tiles: per example it loads Xk (1 float4 / lane), X0 (CIN_FT) and dpre (NT) and issues 4 CIN_FT NT MFMAs, so the
// operand bytes per MFMA fall with NT (the launch is bound by L1/L2 operand traffic as much as by the matrix pipe).
// Reduction over all m = (b, d): one example per k-step group; the NW waves of the workgroup take b = w, w+NW, ... and
// their partial tiles are added in wave order through LDS.
template <int CIN_FT, int NT, int NW>
__global__ __launch_bounds__(64 * NW) void cin_bwd_dw_k(const CinBwdDwArgs p) {
  const int zt = blockIdx.z, zsw = (int)blockIdx.z - p.FG;     // tile plane / sweep plane (after the tiles)
  if (zsw >= 0) {   // piggy-backed optimizer sweep: one 256-thread sweep block per 4 waves of the workgroup
    const uint32_t lin = (((uint32_t)zsw * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * (NW / 4) + (threadIdx.x >> 8);
    if (lin < p.sweep.n_blk) adam_block(p.sweep.args, p.sweep.blk_lo + lin, threadIdx.x & 255);
    return;
  }
  constexpr int TT = CIN_FT * NT;            // output tiles per wave
  constexpr int CH = 4;                      // tiles per reduction pass (<= NW)
  __shared__ float red[NW][CH][256];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int i = lane & 15, kq = lane >> 4;
  const int nb = blockIdx.x * 16 * NT;      // first column of this wave's n tiles
  const int h = blockIdx.y * 16 + i;        // A-operand row (within each field)
  const int f0 = zt * CIN_FT;
  const bool hok = h < p.H;
  const bool want_dc = blockIdx.y == 0 && zt == 0;
  f32x4 acc[CIN_FT][NT], accc[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    accc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < CIN_FT; ++t) acc[t][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  const float one = i == 0 ? 1.f : 0.f;
  // Operand loads are unconditional on clamped indices: out-of-range columns / rows / fields only feed output elements
  // that are never stored, and an example past the batch is neutralised by zeroing its A operand (xk) and the ones-row.
  int nc[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) nc[j] = nb + 16 * j + i < p.N ? nb + 16 * j + i : 0;
  const int hc = hok ? h : 0;
  struct Ops { float4 dp[NT]; float4 xk; float4 x0[CIN_FT]; float ob; };
  auto load = [&](int b, Ops& o) {
    const bool bok = b < p.B;
    const size_t bc = bok ? (size_t)b : (size_t)p.B - 1;
    o.ob = bok ? one : 0.f;
#pragma unroll
    for (int j = 0; j < NT; ++j) o.dp[j] = *reinterpret_cast<const float4*>(p.dpre + (bc * p.N + nc[j]) * CIN_D + kq * 4);
    const float4 xr = *reinterpret_cast<const float4*>(p.Xk + (bc * p.H + hc) * CIN_D + kq * 4);
    const float okf = bok ? 1.f : 0.f;
    o.xk = make_float4(xr.x * okf, xr.y * okf, xr.z * okf, xr.w * okf);
#pragma unroll
    for (int t = 0; t < CIN_FT; ++t) {
      const int ff = f0 + t < p.F ? f0 + t : p.F - 1;
      o.x0[t] = *reinterpret_cast<const float4*>(p.X0 + (bc * p.F + ff) * CIN_D + kq * 4);
    }
  };
  auto fma_b = [&](const Ops& o) {
    float a[CIN_FT][4];
#pragma unroll
    for (int t = 0; t < CIN_FT; ++t) {
      a[t][0] = o.x0[t].x * o.xk.x; a[t][1] = o.x0[t].y * o.xk.y; a[t][2] = o.x0[t].z * o.xk.z; a[t][3] = o.x0[t].w * o.xk.w;
    }
    // component-major order: consecutive MFMAs go to different accumulators
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int t = 0; t < CIN_FT; ++t)
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          const float bv = c == 0 ? o.dp[j].x : c == 1 ? o.dp[j].y : c == 2 ? o.dp[j].z : o.dp[j].w;
          acc[t][j] = cin_mfma(a[t][c], bv, acc[t][j]);
        }
    if (want_dc) {   // ones-row: row 0 of the tile accumulates sum_m dpre[m, n]
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        accc[j] = cin_mfma(o.ob, o.dp[j].x, accc[j]);
        accc[j] = cin_mfma(o.ob, o.dp[j].y, accc[j]);
        accc[j] = cin_mfma(o.ob, o.dp[j].z, accc[j]);
        accc[j] = cin_mfma(o.ob, o.dp[j].w, accc[j]);
      }
    }
  };
  // software pipeline: the operands of the wave's examples two steps ahead are loading while two feed the matrix pipe
  Ops A, Bq, Cq, Dq;
  load(wv, A);
  load(wv + NW, Bq);
  for (int b = wv; b < p.B; b += 4 * NW) {
    load(b + 2 * NW, Cq);
    load(b + 3 * NW, Dq);
    fma_b(A);
    fma_b(Bq);
    load(b + 4 * NW, A);
    load(b + 5 * NW, Bq);
    fma_b(Cq);
    fma_b(Dq);
  }
  // per-wave partial tiles -> LDS in C layout order (row = 4*kq + r, col = lane & 15), summed in wave order.  CH tiles
  // per pass keep the buffer small: the sweep workgroups of this launch inherit its LDS footprint.
  f32x4 tl[TT + NT];
#pragma unroll
  for (int t = 0; t < CIN_FT; ++t)
#pragma unroll
    for (int j = 0; j < NT; ++j) tl[t * NT + j] = acc[t][j];
#pragma unroll
  for (int j = 0; j < NT; ++j) tl[TT + j] = accc[j];
#pragma unroll
  for (int c0 = 0; c0 < TT + NT; c0 += CH) {
    if (c0) __syncthreads();
#pragma unroll
    for (int u = 0; u < CH; ++u)
      if (c0 + u < TT + NT) {
#pragma unroll
        for (int r = 0; r < 4; ++r) red[wv][u][(kq * 4 + r) * 16 + i] = tl[c0 + u][r];
      }
    __syncthreads();
    // tile c0 + u is summed and stored by wave u (u < CH <= NW).  C layout: col = lane & 15 (n), row = 4*kq + r
    const int tt = c0 + wv;
    if (wv < CH && tt < TT + NT) {
      const bool is_dc = tt >= TT;
      const int t = is_dc ? 0 : tt / NT, j = is_dc ? tt - TT : tt % NT;
      const int n = nb + 16 * j + i;
      if (n < p.N) {
        if (is_dc) {
          if (want_dc && kq == 0) {
            float a = 0.f;
#pragma unroll
            for (int w = 0; w < NW; ++w) a += red[w][wv][i];     // row 0
            p.dc[n] = a;
          }
        } else if (f0 + t < p.F) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int e = (kq * 4 + r) * 16 + i;
            float a = 0.f;
#pragma unroll
            for (int w = 0; w < NW; ++w) a += red[w][wv][e];
            const int hr = blockIdx.y * 16 + kq * 4 + r;
            if (hr < p.H) p.dW[((size_t)(f0 + t) * p.H + hr) * p.N + n] = a;
          }
        }
      }
    }
  }
}

// ----------------------------------------------------------------------------------------------- C ABI
extern "C" int rsx_cin_layer_fwd(const float* X0, const float* Xk, const float* W, const float* c, float* out, int B,
                                 int F, int H, int N, int D, const rsx_adam_slice* sweep_h, rsx_stream_t stream) {
  if (B < 0 || F <= 0 || H <= 0 || N <= 0) return RSX_EINVAL;
  if (B == 0) return RSX_OK;
  if (!X0 || !Xk || !W || !c || !out) return RSX_EINVAL;
  if (D != CIN_D) return RSX_EUNSUPPORTED;
  if (H > 128) return RSX_EUNSUPPORTED;
  const int hsmax = H <= 48 ? 12 : 32;
  const size_t lds = ((size_t)2 * 16 * (4 * hsmax + 4) + (size_t)4 * F * CIN_D) * sizeof(float);
  if (lds > 64 * 1024) return RSX_EUNSUPPORTED;
  CinFwdArgs p{X0, Xk, W, c, out, B, F, H, N, (B + 3) / 4, {}};
  const int rcs = adam_build_slice(sweep_h, p.sweep);
  if (rcs != RSX_OK) return rcs;
  const unsigned gx = (unsigned)((N + 15) / 16);
  const dim3 grid(gx, p.nby + (p.sweep.n_blk + gx - 1) / gx);
  if (H <= 48) RSX_LAUNCH(cin_fwd_k<12>, grid, dim3(256), lds, rsx_s(stream), p);
  else RSX_LAUNCH(cin_fwd_k<32>, grid, dim3(256), lds, rsx_s(stream), p);
  RSX_CHECK_LAUNCH();
  return RSX_OK;
}

extern "C" size_t rsx_cin_bwd_workspace_floats(int B, int F, int H, int N) {
  if (B <= 0 || F <= 0 || H <= 0 || N <= 0) return 0;
  return (size_t)B * N * CIN_D + (size_t)((H + 15) / 16) * B * F * CIN_D;     // dpre + per-h-tile dX0 partials
}

extern "C" int rsx_cin_layer_bwd(const float* X0, const float* Xk, const float* W, const float* out, const float* dout,
                                 const float* gs, const float* wout, float* dXk, int acc_dxk, float* dX0, int acc_dx0, float* dW, float* dc, float* dpre_ws,
                                 int B, int F, int H, int N, int D, const rsx_adam_slice* sweep_h, rsx_stream_t stream) {
  if (B < 0 || F <= 0 || H <= 0 || N <= 0) return RSX_EINVAL;
  if (B == 0) return RSX_OK;
  if (!X0 || !Xk || !W || !out || !dXk || !dX0 || !dW || !dc || !dpre_ws) return RSX_EINVAL;
  if ((!dout && !gs) || (gs && !wout)) return RSX_EINVAL;
  if (dXk == dX0 && !(Xk == X0 && acc_dx0)) return RSX_EINVAL;   // one buffer only for the first layer, accumulating
  if (D != CIN_D || H > 128 || N > 128) return RSX_EUNSUPPORTED;
  const int HT = (H + 15) / 16;
  CinBwdDxArgs a{X0, Xk, W, out, dout, gs, wout, dXk, dX0, dpre_ws, acc_dxk, acc_dx0, B, F, H, N, HT};
  static const int dx_force = getenv("RSX_CIN_DX") ? atoi(getenv("RSX_CIN_DX")) : -1;
  // Many h tiles (H > 64): weight slices shared by 4 examples through LDS (measured 131 vs 137 us backward at H = N = 128);
  // few: one workgroup per example with the rows in registers (66 vs 75 us at H = 39).
  const bool tiled = (N & 3) == 0 && (dx_force >= 0 ? dx_force != 0 : HT >= 5);
  if (tiled) {
    const int EB = 4, FS = 1;
    const int NP = N + 4;
    const size_t r0 = (size_t)2 * FS * 16 * NP, r1 = (size_t)EB * N * CIN_D;
    const size_t lds = ((r0 > r1 ? r0 : r1) + (size_t)EB * F * CIN_D + (size_t)EB * 256 + (size_t)FS * EB * 256) * sizeof(float);
    if (lds > 64 * 1024) return RSX_EUNSUPPORTED;
    float* part_ws = dpre_ws + (size_t)B * N * CIN_D;
    const dim3 grid(HT, (B + EB - 1) / EB), block(64 * EB * FS);
    if (N <= 32) RSX_LAUNCH((cin_bwd_dx2_k<2, 4, 1>), grid, block, lds, rsx_s(stream), a, part_ws);
    else RSX_LAUNCH((cin_bwd_dx2_k<8, 4, 1>), grid, block, lds, rsx_s(stream), a, part_ws);
    RSX_CHECK_LAUNCH();
    const long long n4 = (long long)B * F * 4;
    RSX_LAUNCH(cin_dx0_reduce_k, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, rsx_s(stream), part_ws, dX0, HT, n4,
                       acc_dx0);
    RSX_CHECK_LAUNCH();
  } else {
    const int FS = HT <= 3 ? 4 : HT <= 4 ? 2 : 1;   // waves = HT * FS <= 12, an even load for the 4 SIMDs
    const size_t lds = ((size_t)CIN_BT * (N + F + H) * CIN_D + (size_t)HT * CIN_BT * F * CIN_D + (size_t)FS * HT * CIN_BT * 256) * sizeof(float);
    if (lds > 160 * 1024) return RSX_EUNSUPPORTED;
    if (lds > 64 * 1024) {   // gfx950 has 160 KiB of LDS per CU; above 64 KiB the kernel must opt in (host-side attribute)
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(cin_bwd_dx_k<8>), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)lds) != hipSuccess ||
          hipFuncSetAttribute(reinterpret_cast<const void*>(cin_bwd_dx_k<2>), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)lds) != hipSuccess)
        return RSX_ELAUNCH;
    }
    const dim3 grid((B + CIN_BT - 1) / CIN_BT), block(64 * HT * FS);
    if (N <= 32) RSX_LAUNCH(cin_bwd_dx_k<2>, grid, block, lds, rsx_s(stream), a);
    else RSX_LAUNCH(cin_bwd_dx_k<8>, grid, block, lds, rsx_s(stream), a);
    RSX_CHECK_LAUNCH();
  }
  // Wave tile = FT fields x 16 h x (NT x 16) n, NW waves split the batch.  The configuration is chosen for BALANCE first
  // (workgroups are equal-sized: ceil(WGs / 256 CUs) rounds of FT*NT work each), then for the larger tile (fewer operand
  // bytes per MFMA).
  static const int force = getenv("RSX_CIN_DW_CFG") ? atoi(getenv("RSX_CIN_DW_CFG")) : -1;
  static const int cfgs[5][3] = {{5, 2, 8}, {5, 1, 4}, {3, 1, 4}, {2, 2, 8}, {2, 1, 4}};
  int best = 0;
  double best_cost = 1e30;
  for (int c = 0; c < 5; ++c) {
    const int ft = cfgs[c][0], nt = cfgs[c][1];
    const long wgs = (long)((N + 16 * nt - 1) / (16 * nt)) * HT * ((F + ft - 1) / ft);
    const double cost = (double)((wgs + 255) / 256) * ft * nt * (1.0 + 0.02 * c);   // earlier entries win ties
    if (cost < best_cost) { best_cost = cost; best = c; }
  }
  if (sweep_h != nullptr) best = 2;   // carrying sweep workgroups: they inherit the tile kernel's footprint, and the small
                                      // 4-wave / 104-VGPR configuration co-runs best with them (measured end to end)
  if (force >= 0 && force < 5) best = force;
  const int FT = cfgs[best][0], NT = cfgs[best][1], NW = cfgs[best][2];
  CinBwdDwArgs w{X0, Xk, dpre_ws, dW, dc, B, F, H, N, (F + FT - 1) / FT, {}};
  const int rcs = adam_build_slice(sweep_h, w.sweep);
  if (rcs != RSX_OK) return rcs;
  const unsigned gx = (unsigned)((N + 16 * NT - 1) / (16 * NT));
  const unsigned plane = gx * (unsigned)HT * (unsigned)(NW / 4);
  const unsigned zs = (w.sweep.n_blk + plane - 1) / plane;           // extra z-planes that carry the sweep
  const dim3 grid(gx, HT, w.FG + zs);
  switch (best) {
    case 0: RSX_LAUNCH((cin_bwd_dw_k<5, 2, 8>), grid, dim3(512), 0, rsx_s(stream), w); break;
    case 1: RSX_LAUNCH((cin_bwd_dw_k<5, 1, 4>), grid, dim3(256), 0, rsx_s(stream), w); break;
    case 2: RSX_LAUNCH((cin_bwd_dw_k<3, 1, 4>), grid, dim3(256), 0, rsx_s(stream), w); break;
    case 3: RSX_LAUNCH((cin_bwd_dw_k<2, 2, 8>), grid, dim3(512), 0, rsx_s(stream), w); break;
    default: RSX_LAUNCH((cin_bwd_dw_k<2, 1, 4>), grid, dim3(256), 0, rsx_s(stream), w); break;
  }
  RSX_CHECK_LAUNCH();
  return RSX_OK;
}

// ----------------------------------------------------------------------------------------------- 'cin_net' output head
// xdeepfm/xdeepfm.py:180-182: result = reduce_sum(concat(final_result, 1), -1); cin_y = dense(result, 1, relu).
// The concat and the [B, sum N] reduce_sum are never materialised: forward and backward read the layer maps in place.
constexpr int CIN_MAXL = 8;
struct CinOutArgs {
  const float* out[CIN_MAXL];   // [B, n_k, 16] each
  int n[CIN_MAXL], off[CIN_MAXL];
  int L, B, tot;
  const float* Wout;   // [tot]
  const float* bout;   // [1]
  float* y;            // [B]
  const float* gy;     // [B]       (backward)
  float* gs;           // [B]  out: gy * relu'(y)
  float* dWout;        // [tot]
  float* dbout;        // [1]
  // optional rider (xdeepfm/xdeepfm.py:127,131): gradient of the numeric part of the linear_net kernel,
  // dwnum[j] = sum_b logx[b, j] * g_lin[b] -- one more workgroup of this launch instead of a library gemv launch
  const float* logx;   // [B, nnum]
  const float* g_lin;  // [B]
  float* dwnum;        // [nnum]
  int nnum;
};

// one wave per example: lane e walks the float4 of each map, fixed-order butterfly at the end
__global__ __launch_bounds__(256) void cin_out_fwd_k(const CinOutArgs p) {
  const int lane = threadIdx.x & 63;
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= p.B) return;
  float s = 0.f;
  for (int k = 0; k < p.L; ++k) {
    const float4* src = reinterpret_cast<const float4*>(p.out[k] + (size_t)b * p.n[k] * CIN_D);
    const float* w = p.Wout + p.off[k];
    const int n4 = p.n[k] * 4;
    for (int e0 = lane; e0 < n4; e0 += 64 * 8) {      // 8 independent loads in flight (clamped index, masked weight)
      float4 v[8];
      float wv[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int e = e0 + 64 * u;
        const int ec = e < n4 ? e : n4 - 1;
        v[u] = src[ec];
        wv[u] = w[ec >> 2] * (e < n4 ? 1.f : 0.f);
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) s += wv[u] * ((v[u].x + v[u].y) + (v[u].z + v[u].w));
    }
  }
#pragma unroll
  for (int m = 1; m < 64; m <<= 1) s += __shfl_xor(s, m);
  if (lane == 0) p.y[b] = fmaxf(s + p.bout[0], 0.f);
}

// grid = ceil(tot/16) + 1, block = 1024: workgroup j owns 16 columns of the concatenated map (a layer's width is a
// multiple of 16 or the tile is clipped to its layer); wave w takes examples w, w+16, ...; partials added in wave order.
// The last workgroup writes gs and dbout.
__global__ __launch_bounds__(1024) void cin_out_bwd_k(const CinOutArgs p) {
  __shared__ float red[16][64];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  if (p.dwnum != nullptr && blockIdx.x == gridDim.x - 2) {      // wave w: columns w, w + 16, ...; lanes stride the batch
    for (int j = wv; j < p.nnum; j += 16) {
      float s = 0.f;
      for (int b0 = lane; b0 < p.B; b0 += 64 * 8) {
        float x[8], g[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int b = b0 + 64 * u;
          const int bc = b < p.B ? b : p.B - 1;
          x[u] = p.logx[(size_t)bc * p.nnum + j];
          g[u] = b < p.B ? p.g_lin[bc] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) s += x[u] * g[u];
      }
#pragma unroll
      for (int m = 1; m < 64; m <<= 1) s += __shfl_xor(s, m);
      if (lane == 0) p.dwnum[j] = s;
    }
    return;
  }
  if (blockIdx.x == gridDim.x - 1) {
    float s = 0.f;
    for (int b = tid; b < p.B; b += 1024) {
      const float g = p.y[b] > 0.f ? p.gy[b] : 0.f;
      p.gs[b] = g;
    }
    // dbout: one wave, fixed order
    if (wv == 0) {
      for (int b = lane; b < p.B; b += 64) s += p.y[b] > 0.f ? p.gy[b] : 0.f;
#pragma unroll
      for (int m = 1; m < 64; m <<= 1) s += __shfl_xor(s, m);
      if (lane == 0) p.dbout[0] = s;
    }
    return;
  }
  // tile -> (layer k, first column n0): tiles are counted per layer
  int k = 0, t = blockIdx.x;
  while (k < p.L - 1 && t >= (p.n[k] + 15) / 16) { t -= (p.n[k] + 15) / 16; ++k; }
  const int n0 = t * 16, nl = lane >> 2, dq = lane & 3;
  const int n = n0 + nl;
  const bool ok = n < p.n[k];
  const float* src = p.out[k] + ((size_t)(ok ? n : 0)) * CIN_D + dq * 4;
  float s = 0.f;
  for (int b0 = wv; b0 < p.B; b0 += 16 * 8) {      // 8 independent loads in flight (clamped index, masked weight)
    float4 v[8];
    float g[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int b = b0 + 16 * u;
      const int bc = b < p.B ? b : p.B - 1;
      v[u] = *reinterpret_cast<const float4*>(src + (size_t)bc * p.n[k] * CIN_D);
      const float yv = p.y[bc], gv = p.gy[bc];
      g[u] = (b < p.B && yv > 0.f) ? gv : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) s += g[u] * ((v[u].x + v[u].y) + (v[u].z + v[u].w));
  }
  s += __shfl_xor(s, 1);
  s += __shfl_xor(s, 2);
  red[wv][lane] = s;
  __syncthreads();
  if (tid < 16) {
    float a = 0.f;
#pragma unroll
    for (int w = 0; w < 16; ++w) a += red[w][tid * 4];
    if (n0 + tid < p.n[k]) p.dWout[p.off[k] + n0 + tid] = a;
  }
}

static int cin_out_args(CinOutArgs& a, const float* const* outs_h, const int32_t* sizes_h, int L, int B, int D) {
  if (!outs_h || !sizes_h || L <= 0 || B < 0) return RSX_EINVAL;
  if (L > CIN_MAXL || D != CIN_D) return RSX_EUNSUPPORTED;
  a.L = L; a.B = B; a.tot = 0;
  for (int k = 0; k < L; ++k) {
    if (!outs_h[k] || sizes_h[k] <= 0) return RSX_EINVAL;
    a.out[k] = outs_h[k]; a.n[k] = sizes_h[k]; a.off[k] = a.tot; a.tot += sizes_h[k];
  }
  return RSX_OK;
}

extern "C" int rsx_cin_out_fwd(const float* const* outs_h, const int32_t* sizes_h, int L, const float* Wout,
                               const float* bout, float* y, int B, int D, rsx_stream_t stream) {
  CinOutArgs a{};
  const int rc = cin_out_args(a, outs_h, sizes_h, L, B, D);
  if (rc != RSX_OK) return rc;
  if (B == 0) return RSX_OK;
  if (!Wout || !bout || !y) return RSX_EINVAL;
  a.Wout = Wout; a.bout = bout; a.y = y;
  RSX_LAUNCH(cin_out_fwd_k, dim3((B + 3) / 4), dim3(256), 0, rsx_s(stream), a);
  RSX_CHECK_LAUNCH();
  return RSX_OK;
}

extern "C" int rsx_cin_out_bwd_lin(const float* const* outs_h, const int32_t* sizes_h, int L, const float* y, const float* gy,
                                   float* gs, float* dWout, float* dbout, const float* logx, const float* g_lin,
                                   float* dwnum, int nnum, int B, int D, rsx_stream_t stream) {
  CinOutArgs a{};
  const int rc = cin_out_args(a, outs_h, sizes_h, L, B, D);
  if (rc != RSX_OK) return rc;
  if (!y || !gy || !gs || !dWout || !dbout) return RSX_EINVAL;
  if (dwnum != nullptr && (!logx || !g_lin || nnum <= 0)) return RSX_EINVAL;
  a.y = const_cast<float*>(y); a.gy = gy; a.gs = gs; a.dWout = dWout; a.dbout = dbout;
  a.logx = logx; a.g_lin = g_lin; a.dwnum = dwnum; a.nnum = nnum;
  int tiles = 0;
  for (int k = 0; k < L; ++k) tiles += (sizes_h[k] + 15) / 16;
  RSX_LAUNCH(cin_out_bwd_k, dim3(tiles + 1 + (dwnum != nullptr ? 1 : 0)), dim3(1024), 0, rsx_s(stream), a);
  RSX_CHECK_LAUNCH();
  return RSX_OK;
}

extern "C" int rsx_cin_out_bwd(const float* const* outs_h, const int32_t* sizes_h, int L, const float* y, const float* gy,
                               float* gs, float* dWout, float* dbout, int B, int D, rsx_stream_t stream) {
  return rsx_cin_out_bwd_lin(outs_h, sizes_h, L, y, gy, gs, dWout, dbout, nullptr, nullptr, nullptr, 0, B, D, stream);
}
