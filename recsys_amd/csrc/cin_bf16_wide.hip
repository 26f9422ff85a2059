// xDeepFM CIN layer, bf16 MFMA path, "wide" launches (round 5): EIGHT examples per workgroup.
// Reference call site: xdeepfm/xdeepfm.py:145-172.  Same arithmetic and the same rounding points as cin_bf16.hip (Xk, W and
// dpre are bf16 MFMA operands, X0 / bias / relu / every sum fp32); what changes is who shares what:
//
//   cin_fwd_bf16_k / cin_bwd_dx_bf16_k (cin_bf16.hip) give a workgroup TWO examples.  Every 1-KiB filter fragment a wave
//   loads feeds two MFMAs, a CU with 16 resident waves pulls 640 KiB of fragments through its vector-memory path per launch,
//   and each wave walks ten dependent load -> multiply steps of ~0.8 us each: the launches (12 / 17 us at batch 256) are
//   bound by that chain, the matrix cores are busy 17 % of the time (profiles/r03_*_pmc_*).
//
//   Here a workgroup of 8 waves owns 8 examples and ONE 16-wide tile of the filter's other index (16 outputs n in the
//   forward, 16 inputs h in the backward); the waves split the FIELDS (f = wave, wave + 8, ...: five each at F = 39).  The
//   stationary operand (Xk, or dpre) of all eight examples sits in registers (8 x KS 16-byte quads), so a filter fragment
//   feeds EIGHT MFMAs, a wave's whole filter stream is 5 x KS KiB -- requested before anything else in the kernel, up to
//   three fields in flight -- and a CU moves 160 KiB instead of 640.  Grid = (tiles, B / 8): consecutive workgroups are
//   dealt to consecutive XCDs, so with 8 tiles each XCD's L2 holds one tile's slice of the filters (160 KiB).
//
//   Backward: a workgroup sees one h tile only, so dX0[b, f, :] = sum_h Xk[b,h,:] * U_f[h,:] comes out as one partial per
//   h tile (dx0_parts [HT][B][F*16]); cin_dx0_reduce_k adds the tiles in order (one launch for all layers).
// All sums in fixed order: deterministic, no atomics.
#include "rsx_common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16_t;

namespace {

constexpr int CW_D = 16;
constexpr int CW_E = 8;        // examples per workgroup
constexpr int CW_NFW = 5;      // fields per wave: F <= 40

__device__ __forceinline__ f32x4 mfma_bf16(bf16x8 a, bf16x8 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ bf16x8 ld_bf16x8(const bf16_t* p) {
  return __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(p));
}
__device__ __forceinline__ uint32_t pack2(float lo, float hi) {
  bf16x2 v;
  v[0] = (bf16_t)lo;
  v[1] = (bf16_t)hi;
  return __builtin_bit_cast(uint32_t, v);
}
inline int rup(int x, int m) { return (x + m - 1) / m * m; }

// [E][rows][16] fp32 (through `load(e, row, quarter)`, rows >= the real count must come back as zeros) -> bf16, transposed
// to dst[e][d][RPP] (rows contiguous: what the MFMA's k index walks).  A thread takes TWO rows of one d-quarter and writes
// four packed dwords; lanes = (quarter fastest, row pair): the 64 dwords of a wave's store land in 64 different banks
// (row stride RPP / 2 = 68 / 52 / 36 / 20 dwords: the quarter's stride of four rows is 16 banks).
template <int KS>
struct StageRows {
  float4 a[KS], b[KS];
  // 512 threads, CW_E * 16 KS * 4 = 512 KS items: KS per thread, every load requested before the first store (written as a
  // loop of load -> store the compiler waits for each round trip in turn)
  template <typename Load>
  __device__ __forceinline__ void load(int tid, Load ld) {
#pragma unroll
    for (int u = 0; u < KS; ++u) {
      const int it = tid + 512 * u;
      const int dq = it & 3, rp = (it >> 2) % (16 * KS), e = it / (64 * KS);
      a[u] = ld(e, 2 * rp, dq);
      b[u] = ld(e, 2 * rp + 1, dq);
    }
  }
  __device__ __forceinline__ void store(bf16_t* dst, int tid) const {
    constexpr int RPP = 32 * KS + 8;
#pragma unroll
    for (int u = 0; u < KS; ++u) {
      const int it = tid + 512 * u;
      const int dq = it & 3, rp = (it >> 2) % (16 * KS), e = it / (64 * KS);
      uint32_t* t = reinterpret_cast<uint32_t*>(dst + ((size_t)e * 16 + dq * 4) * RPP + 2 * rp);
      t[0 * (RPP / 2)] = pack2(a[u].x, b[u].x);
      t[1 * (RPP / 2)] = pack2(a[u].y, b[u].y);
      t[2 * (RPP / 2)] = pack2(a[u].z, b[u].z);
      t[3 * (RPP / 2)] = pack2(a[u].w, b[u].w);
    }
  }
};

// X0 of the eight examples -> LDS [E][CW_FP * 16], zeros for the fields F .. CW_FP - 1 (a wave's five fields need no bounds
// test: a field past F multiplies by zero) -- 3 float4 per thread, requested together
constexpr int CW_FP = 8 * CW_NFW;
struct StageX0 {
  float4 v[3];
  __device__ __forceinline__ void load(const float* X0, int b0, int B, int F, int tid) {
#pragma unroll
    for (int u = 0; u < 3; ++u) {
      const int e4 = tid + 512 * u;
      const int ex = e4 / (CW_FP * 4), r = e4 % (CW_FP * 4);
      // (unconditional loads from clamped addresses, zeroed afterwards: a load under a condition becomes a branch, and the
      // compiler waits for each of them in turn)
      const int exc = ex < CW_E ? ex : CW_E - 1;
      const bool ok = e4 < CW_E * CW_FP * 4 && (r >> 2) < F && b0 + ex < B;
      const int bc = b0 + exc < B ? b0 + exc : B - 1, rc = (r >> 2) < F ? r : 0;
      const float4 t = reinterpret_cast<const float4*>(X0 + (size_t)bc * F * CW_D)[rc];
      v[u] = make_float4(ok ? t.x : 0.f, ok ? t.y : 0.f, ok ? t.z : 0.f, ok ? t.w : 0.f);
    }
  }
  __device__ __forceinline__ void store(float* sX0, int tid) const {
#pragma unroll
    for (int u = 0; u < 3; ++u) {
      const int e4 = tid + 512 * u;
      if (e4 < CW_E * CW_FP * 4) reinterpret_cast<float4*>(sX0)[e4] = v[u];
    }
  }
};

// ------------------------------------------------------------------------------------------------------------ forward
struct CwFwdArgs {
  const float* X0;      // [B, F, 16]
  const float* Xk;      // [B, H, 16]
  const bf16_t* Wt16;   // fragment-major [F][N16/16][Hp/32][64][8] (cin_prep_bf16_k)
  const float* c;       // [N]
  float* out;           // [B, N, 16]
  int B, F, H, N, N16, Hp;
};

// grid = (N16 / 16, ceil(B / 8)), block = 512.  KS = Hp / 32 k-steps per field; RING fields of filter fragments in flight.
// dyn LDS: 8*40*16 floats (X0) + 8*16*(32 KS + 8) bf16 (Xk^T) + 8*8*256 floats (the waves' partial sums).
template <int KS>
__global__ __launch_bounds__(512) void cin_fwd_bf16_wide_k(const CwFwdArgs p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int E = CW_E, NFW = CW_NFW;
  constexpr int RING = KS >= 4 ? 2 : NFW;
  constexpr int HPP = 32 * KS + 8;
  float* sX0 = lds;                                                   // [E][CW_FP*16]
  float* sR = lds + E * CW_FP * CW_D;                                 // [8 waves][E][4][64]
  bf16_t* sXk = reinterpret_cast<bf16_t*>(sR + 8 * E * 256);          // [E][16][HPP]
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int i = lane & 15, kq = lane >> 4;
  const int n0 = blockIdx.x * 16, b0 = blockIdx.y * E;
  const bool st0 = KS == 4 && blockIdx.x == 0 && blockIdx.y == 0, st1 = KS == 4 && blockIdx.x == 7 && blockIdx.y == gridDim.y - 1;
  (void)st0; (void)st1;
  RSX_STAMP(0, st0); RSX_STAMP(8, st1); RSX_STAMP_MAX(16, KS == 4);
  // the wave's filter stream first: nothing below depends on it until the first MFMA
  const bf16_t* wbase = p.Wt16 + ((size_t)blockIdx.x * KS * 64 + lane) * 8;
  const size_t fstride = (size_t)p.N16 * p.Hp;
  bf16x8 w[RING][KS];
  auto load_w = [&](int g, bf16x8* dst) {
    const int ff = wv + 8 * g;
    const int f = ff < p.F ? ff : p.F - 1;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) dst[ks] = ld_bf16x8(wbase + (size_t)f * fstride + (size_t)ks * 512);
  };
#pragma unroll
  for (int g = 0; g < RING; ++g) load_w(g, w[g]);
  StageX0 sx;
  StageRows<KS> sr;
  sx.load(p.X0, b0, p.B, p.F, tid);
  sr.load(tid, [&](int e, int h, int dq) {
    const bool ok = b0 + e < p.B && h < p.H;
    const int bc = b0 + e < p.B ? b0 + e : p.B - 1, hc = h < p.H ? h : p.H - 1;
    const float4 t = reinterpret_cast<const float4*>(p.Xk + ((size_t)bc * p.H + hc) * CW_D)[dq];
    return make_float4(ok ? t.x : 0.f, ok ? t.y : 0.f, ok ? t.z : 0.f, ok ? t.w : 0.f);
  });
  RSX_STAMP(1, st0); RSX_STAMP(9, st1);
  sx.store(sX0, tid);
  sr.store(sXk, tid);
  RSX_STAMP(2, st0); RSX_STAMP(10, st1);
  __syncthreads();
  RSX_STAMP(3, st0); RSX_STAMP(11, st1);
  bf16x8 a[E][KS];                                 // Xk[b0 + e][h = 32 ks + 8 kq + j][d = i]: the same for every field
#pragma unroll
  for (int e = 0; e < E; ++e)
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) a[e][ks] = ld_bf16x8(sXk + ((size_t)e * 16 + i) * HPP + 32 * ks + 8 * kq);
  f32x4 acc[E];
#pragma unroll
  for (int e = 0; e < E; ++e) acc[e] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int g = 0; g < NFW; ++g) {
    const int f = wv + 8 * g;                      // (f >= F: X0 reads as zero)
#pragma unroll
    for (int e = 0; e < E; ++e) {
      f32x4 T = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) T = mfma_bf16(a[e][ks], w[g % RING][ks], T);
      const float4 x = *reinterpret_cast<const float4*>(sX0 + (e * CW_FP + f) * CW_D + kq * 4);
      acc[e][0] = __builtin_fmaf(x.x, T[0], acc[e][0]);
      acc[e][1] = __builtin_fmaf(x.y, T[1], acc[e][1]);
      acc[e][2] = __builtin_fmaf(x.z, T[2], acc[e][2]);
      acc[e][3] = __builtin_fmaf(x.w, T[3], acc[e][3]);
    }
    if (g + RING < NFW) load_w(g + RING, w[g % RING]);
    __builtin_amdgcn_sched_barrier(0);             // (register budget: nothing of field g + 1 is hoisted into field g)
  }
#pragma unroll
  for (int e = 0; e < E; ++e)
#pragma unroll
    for (int r = 0; r < 4; ++r) sR[((wv * E + e) * 4 + r) * 64 + lane] = acc[e][r];
  RSX_STAMP(4, st0); RSX_STAMP(12, st1);
  __syncthreads();
  RSX_STAMP(5, st0); RSX_STAMP(13, st1);
  {                                                // wave e finishes example e: the waves' partials in wave order
    const int e = wv, b = b0 + e;
    const bool nok = n0 + i < p.N;
    const float cv = p.c[nok ? n0 + i : 0];
    float o[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float s = sR[((0 * E + e) * 4 + r) * 64 + lane];
#pragma unroll
      for (int w8 = 1; w8 < 8; ++w8) s += sR[((w8 * E + e) * 4 + r) * 64 + lane];
      o[r] = fmaxf(s + cv, 0.f);
    }
    if (nok && b < p.B)
      *reinterpret_cast<float4*>(p.out + ((size_t)b * p.N + n0 + i) * CW_D + kq * 4) = make_float4(o[0], o[1], o[2], o[3]);
  }
  RSX_STAMP(6, st0); RSX_STAMP(14, st1); RSX_STAMP_MAX(17, KS == 4);
}

// ------------------------------------------------------------------------------------------------ backward: dXk, dX0
struct CwDxArgs {
  const float* X0;      // [B, F, 16]
  const float* Xk;      // [B, H, 16]
  const bf16_t* W16;    // fragment-major [F][H16/16][Np/32][64][8] (cin_prep_bf16_k)
  const float* out;     // [B, N, 16] this layer's relu output
  const float* dout;    // [B, N, 16] gradient wrt the relu output (nullable when gs is given)
  const float* gs;      // [B] nullable: direct-connect gradient gs[b] * wout[n], broadcast over d, added to dout
  const float* wout;    // [N]
  float* dXk;           // [B, H, 16]
  float* dx0_parts;     // [HT][B][F*16] out: tile ht's share of dX0
  bf16_t* dpre16;       // [ceil(B/2)][N16/16][64][8] out (workgroups of tile 0): the dW kernel's B fragments (cin_bf16.hip)
  float* dc_part;       // [B][N16] out (workgroups of tile 0): per-example column sums of the UNROUNDED dpre
  int acc_dxk;
  int B, F, H, N, H16, N16, Np;
};

// grid = (H16 / 16, ceil(B / 8)), block = 512: workgroup = 8 examples x the 16 inputs h of tile blockIdx.x, wave w takes the
// fields w, w + 8, ...  U_f^T[h, d] = sum_n W_f[h, n] dpre[b, n, d]: A = W16 fragments (the wave's stream), B = dpre[b]^T
// (k = n, column = d) of the eight examples in registers.  dXk[b, h, d] += X0[b, f, d] U_f^T[h, d] (summed over the wave's
// fields in registers, over the waves in order through LDS); dX0[b, f, d] = sum_h Xk[b, h, d] U_f^T[h, d] over this tile's 16
// h: four lane-quarter partials through LDS, added in order, written to dx0_parts[tile].
// dyn LDS: sDpT 8*16*(32 KSN + 8) bf16 | sX0 8*40*16 | sXkT 8*64*4 | sP 8*40*64 floats (sDx 8*8*256 floats aliases sP).
template <int KSN>
__global__ __launch_bounds__(512) void cin_bwd_dx_bf16_wide_k(const CwDxArgs p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int E = CW_E, NFW = CW_NFW;
  constexpr int RING = KSN >= 4 ? 2 : NFW;
  constexpr int NPP = 32 * KSN + 8;
  bf16_t* sDpT = reinterpret_cast<bf16_t*>(lds);                      // [E][16][NPP]
  float* sX0 = lds + E * 16 * NPP / 2;                                // [E][CW_FP*16]
  float* sXkT = sX0 + E * CW_FP * CW_D;                               // [E][4 kq][16 i][4 r]
  float* sP = sXkT + E * 256;                                         // [E][CW_FP][4 kq][16 i]
  float* sDx = sP;                                                    // [8 waves][E][4][64]   (after sP was consumed)
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int i = lane & 15, kq = lane >> 4;
  const int ht = blockIdx.x, b0 = blockIdx.y * E;
  const bool st0 = p.dout == nullptr && blockIdx.x == 0 && blockIdx.y == 0;      // (the last layer's launch)
  (void)st0;
  RSX_STAMP(32, st0); RSX_STAMP_MAX(48, p.dout == nullptr);
  const bf16_t* wbase = p.W16 + ((size_t)ht * KSN * 64 + lane) * 8;
  const size_t fstride = (size_t)p.H16 * p.Np;
  bf16x8 w[RING][KSN];
  auto load_w = [&](int g, bf16x8* dst) {
    const int ff = wv + 8 * g;
    const int f = ff < p.F ? ff : p.F - 1;
#pragma unroll
    for (int ks = 0; ks < KSN; ++ks) dst[ks] = ld_bf16x8(wbase + (size_t)f * fstride + (size_t)ks * 512);
  };
#pragma unroll
  for (int g = 0; g < RING; ++g) load_w(g, w[g]);
  StageX0 sx;
  sx.load(p.X0, b0, p.B, p.F, tid);
  {   // Xk of this tile: thread = (example, h in tile, d-quarter) -> sXkT[e][h >> 2][d][h & 3]
    const int e = tid >> 6, h16 = (tid >> 2) & 15, dq = tid & 3;
    const int h = 16 * ht + h16;
    const bool ok = b0 + e < p.B && h < p.H;
    const int bc = b0 + e < p.B ? b0 + e : p.B - 1, hc = h < p.H ? h : p.H - 1;
    float4 v = reinterpret_cast<const float4*>(p.Xk + ((size_t)bc * p.H + hc) * CW_D)[dq];
    v = make_float4(ok ? v.x : 0.f, ok ? v.y : 0.f, ok ? v.z : 0.f, ok ? v.w : 0.f);
    float* t = sXkT + ((e * 4 + (h16 >> 2)) * 16 + dq * 4) * 4 + (h16 & 3);
    t[0] = v.x;
    t[4] = v.y;
    t[8] = v.z;
    t[12] = v.w;
  }
  // dpre = relu'(out) * (dout + gs * wout) of the eight examples: bf16, transposed, into LDS; the workgroups of tile 0 also
  // leave the dW launch's fragments and the bias gradient's per-example partial sums
  const bool lead = ht == 0, has_dout = p.dout != nullptr, has_gs = p.gs != nullptr;
  StageRows<KSN> sr;
  sr.load(tid, [&](int e, int n, int dq) {
    const int b = b0 + e;
    const bool ok = b < p.B && n < p.N;
    const int bc = b < p.B ? b : p.B - 1, nc = n < p.N ? n : p.N - 1;
    const size_t at = ((size_t)bc * p.N + nc) * 4 + dq;
    const float4 o = reinterpret_cast<const float4*>(p.out)[at];
    float4 g = F4Z;
    if (has_dout) g = reinterpret_cast<const float4*>(p.dout)[at];         // (workgroup-uniform)
    if (has_gs) {
      const float a = p.gs[bc] * p.wout[nc];
      g = make_float4(g.x + a, g.y + a, g.z + a, g.w + a);
    }
    const float4 v = make_float4((ok && o.x > 0.f) ? g.x : 0.f, (ok && o.y > 0.f) ? g.y : 0.f, (ok && o.z > 0.f) ? g.z : 0.f,
                                 (ok && o.w > 0.f) ? g.w : 0.f);
    if (lead) {                                    // (workgroup-uniform)
      float s = (v.x + v.y) + (v.z + v.w);
      s += __shfl_xor(s, 1);                       // the 4 d-quarters of row n sit in adjacent lanes
      s += __shfl_xor(s, 2);
      if (b < 2 * ((p.B + 1) / 2) && n < p.N16) {      // (the last pair's missing example: zero fragments)
        if (dq == 0 && b < p.B) p.dc_part[(size_t)b * p.N16 + n] = s;
        const size_t fr = ((((size_t)(b >> 1) * (p.N16 >> 4) + (n >> 4)) * 64 + (2 * (b & 1) + (dq >> 1)) * 16 + (n & 15)) * 8) + (dq & 1) * 4;
        uint2 q;
        q.x = pack2(v.x, v.y);
        q.y = pack2(v.z, v.w);
        *reinterpret_cast<uint2*>(p.dpre16 + fr) = q;
      }
    }
    return v;
  });
  RSX_STAMP(33, st0);
  sx.store(sX0, tid);
  sr.store(sDpT, tid);
  RSX_STAMP(34, st0);
  __syncthreads();
  RSX_STAMP(35, st0);
  bf16x8 bd[E][KSN];                               // dpre[b0 + e][n = 32 ks + 8 kq + j][d = i]
#pragma unroll
  for (int e = 0; e < E; ++e)
#pragma unroll
    for (int ks = 0; ks < KSN; ++ks) bd[e][ks] = ld_bf16x8(sDpT + ((size_t)e * 16 + i) * NPP + 32 * ks + 8 * kq);
  f32x4 dxk[E];
#pragma unroll
  for (int e = 0; e < E; ++e) dxk[e] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int g = 0; g < NFW; ++g) {
    const int f = wv + 8 * g;                      // (f >= F: X0 reads as zero, the dX0 slot is never read)
#pragma unroll
    for (int e = 0; e < E; ++e) {
      f32x4 U = {0.f, 0.f, 0.f, 0.f};              // U_f^T[h = 16 ht + 4 kq + r][d = i]
#pragma unroll
      for (int ks = 0; ks < KSN; ++ks) U = mfma_bf16(w[g % RING][ks], bd[e][ks], U);
      const float x = sX0[(e * CW_FP + f) * CW_D + i];
      const float4 xk = *reinterpret_cast<const float4*>(sXkT + ((e * 4 + kq) * 16 + i) * 4);
      dxk[e][0] = __builtin_fmaf(x, U[0], dxk[e][0]);
      dxk[e][1] = __builtin_fmaf(x, U[1], dxk[e][1]);
      dxk[e][2] = __builtin_fmaf(x, U[2], dxk[e][2]);
      dxk[e][3] = __builtin_fmaf(x, U[3], dxk[e][3]);
      sP[((e * CW_FP + f) * 4 + kq) * 16 + i] = ((U[0] * xk.x + U[1] * xk.y) + U[2] * xk.z) + U[3] * xk.w;
    }
    if (g + RING < NFW) load_w(g + RING, w[g % RING]);
    __builtin_amdgcn_sched_barrier(0);             // (register budget: nothing of field g + 1 is hoisted into field g)
  }
  RSX_STAMP(36, st0);
  __syncthreads();
  RSX_STAMP(37, st0);
  // this tile's share of dX0: the four lane-quarter partials of every (example, field, d) in order
  for (int e4 = tid; e4 < E * p.F * 4; e4 += 512) {
    const int ex = e4 / (p.F * 4), r = e4 - ex * (p.F * 4);
    const int f = r >> 2, dq = r & 3;
    const float4* q = reinterpret_cast<const float4*>(sP + ((ex * CW_FP + f) * 4) * 16) + dq;
    const float4 s = f4_add(f4_add(f4_add(q[0], q[4]), q[8]), q[12]);
    if (b0 + ex < p.B)
      reinterpret_cast<float4*>(p.dx0_parts + ((size_t)ht * p.B + b0 + ex) * p.F * CW_D)[r] = s;
  }
  __syncthreads();
#pragma unroll
  for (int e = 0; e < E; ++e)
#pragma unroll
    for (int r = 0; r < 4; ++r) sDx[((wv * E + e) * 4 + r) * 64 + lane] = dxk[e][r];
  RSX_STAMP(38, st0);
  __syncthreads();
  {   // wave e finishes example e: dXk[b][h = 16 ht + 4 kq + r][d = i], the waves' field shares in wave order
    const int e = wv, b = b0 + e;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float s = sDx[((0 * E + e) * 4 + r) * 64 + lane];
#pragma unroll
      for (int w8 = 1; w8 < 8; ++w8) s += sDx[((w8 * E + e) * 4 + r) * 64 + lane];
      const int h = 16 * ht + 4 * kq + r;
      if (b < p.B && h < p.H) {
        float* dst = p.dXk + ((size_t)b * p.H + h) * CW_D + i;
        *dst = p.acc_dxk ? *dst + s : s;
      }
    }
  }
  RSX_STAMP(39, st0); RSX_STAMP_MAX(49, p.dout == nullptr);
}

// dX0[e] = (acc ? dX0[e] : 0) + sum over the jobs' tiles, in (job, tile) order; e over B * F * 16 floats
constexpr int CW_MAXJ = 4;
struct CwRedArgs {
  const float* parts[CW_MAXJ];
  int tiles[CW_MAXJ];
  int njobs, acc;
  float* dX0;
  size_t n4;            // B * F * 16 / 4
};
__global__ __launch_bounds__(256) void cin_dx0_reduce_k(const CwRedArgs p) {
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < p.n4; e += (size_t)gridDim.x * 256) {
    float4 s = p.acc ? reinterpret_cast<const float4*>(p.dX0)[e] : F4Z;
#pragma unroll
    for (int j = 0; j < CW_MAXJ; ++j)
      if (j < p.njobs)
        for (int t = 0; t < p.tiles[j]; ++t) s = f4_add(s, reinterpret_cast<const float4*>(p.parts[j])[(size_t)t * p.n4 + e]);
    reinterpret_cast<float4*>(p.dX0)[e] = s;
  }
}

template <typename K>
int opt_in_lds(K kernel, size_t lds) {
  if (lds > 160 * 1024) return RSX_EUNSUPPORTED;
  if (lds > 64 * 1024 &&
      hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
    return RSX_ELAUNCH;
  return RSX_OK;
}

}  // namespace

#ifdef RSX_STAMPS
extern "C" int rsx_dbg_stamps_cin_wide(unsigned long long* out_h) {
  return hipMemcpyFromSymbol(out_h, HIP_SYMBOL(rsx_stamps_d), sizeof(unsigned long long) * 64) == hipSuccess ? RSX_OK : RSX_ELAUNCH;
}
extern "C" int rsx_dbg_stamps_cin_wide_reset() {
  static const unsigned long long z[64] = {0};
  return hipMemcpyToSymbol(HIP_SYMBOL(rsx_stamps_d), z, sizeof(z)) == hipSuccess ? RSX_OK : RSX_ELAUNCH;
}
#endif

// Internal launchers (declared in cin_bf16_wide.h, called by the entry points in cin_bf16.hip).
bool cin_wide_supported(int F, int H, int N) { return F <= 8 * CW_NFW && H <= 128 && N <= 128; }

int cin_wide_fwd(const float* X0, const float* Xk, const void* wt16, const float* c, float* out, int B, int F, int H, int N,
                 hipStream_t stream) {
  const int N16 = rup(N, 16), Hp = rup(H, 32);
  const CwFwdArgs a{X0, Xk, static_cast<const bf16_t*>(wt16), c, out, B, F, H, N, N16, Hp};
  const dim3 grid((unsigned)(N16 / 16), (unsigned)((B + CW_E - 1) / CW_E));
  const size_t lds = ((size_t)CW_E * CW_FP * CW_D + 8 * CW_E * 256) * sizeof(float) + (size_t)CW_E * 16 * (Hp + 8) * 2;
#define RSX_CW_FWD(KS)                                                               \
  {                                                                                  \
    const int rc = opt_in_lds(cin_fwd_bf16_wide_k<KS>, lds);                         \
    if (rc != RSX_OK) return rc;                                                     \
    RSX_LAUNCH(cin_fwd_bf16_wide_k<KS>, grid, dim3(512), lds, stream, a);            \
  }
  switch (Hp / 32) {
    case 1: RSX_CW_FWD(1); break;
    case 2: RSX_CW_FWD(2); break;
    case 3: RSX_CW_FWD(3); break;
    default: RSX_CW_FWD(4); break;
  }
#undef RSX_CW_FWD
  RSX_CHECK_LAUNCH();
  return RSX_OK;
}

int cin_wide_dx(const float* X0, const float* Xk, const void* w16, const float* out, const float* dout, const float* gs,
                const float* wout, float* dXk, int acc_dxk, float* dx0_parts, void* dpre16, float* dc_part, int B, int F,
                int H, int N, hipStream_t stream) {
  const int H16 = rup(H, 16), N16 = rup(N, 16), Np = rup(N, 32);
  const CwDxArgs a{X0, Xk, static_cast<const bf16_t*>(w16), out, dout, gs, wout, dXk, dx0_parts, static_cast<bf16_t*>(dpre16),
                   dc_part, acc_dxk, B, F, H, N, H16, N16, Np};
  const dim3 grid((unsigned)(H16 / 16), (unsigned)((B + CW_E - 1) / CW_E));
  const size_t sp = (size_t)CW_E * CW_FP * 64, sdx = (size_t)8 * CW_E * 256;
  const size_t lds = (size_t)CW_E * 16 * (Np + 8) * 2 + ((size_t)CW_E * CW_FP * CW_D + CW_E * 256 + (sp > sdx ? sp : sdx)) * sizeof(float);
#define RSX_CW_DX(KSN)                                                               \
  {                                                                                  \
    const int rc = opt_in_lds(cin_bwd_dx_bf16_wide_k<KSN>, lds);                     \
    if (rc != RSX_OK) return rc;                                                     \
    RSX_LAUNCH(cin_bwd_dx_bf16_wide_k<KSN>, grid, dim3(512), lds, stream, a);        \
  }
  switch (Np / 32) {
    case 1: RSX_CW_DX(1); break;
    case 2: RSX_CW_DX(2); break;
    case 3: RSX_CW_DX(3); break;
    default: RSX_CW_DX(4); break;
  }
#undef RSX_CW_DX
  RSX_CHECK_LAUNCH();
  return RSX_OK;
}

int cin_wide_dx0_reduce(const float* const* parts, const int* tiles, int njobs, float* dX0, int acc, int B, int F,
                        hipStream_t stream) {
  if (njobs <= 0 || njobs > CW_MAXJ) return RSX_EUNSUPPORTED;
  CwRedArgs a{};
  for (int j = 0; j < njobs; ++j) {
    a.parts[j] = parts[j];
    a.tiles[j] = tiles[j];
  }
  a.njobs = njobs;
  a.acc = acc;
  a.dX0 = dX0;
  a.n4 = (size_t)B * F * CW_D / 4;
  const unsigned blocks = (unsigned)((a.n4 + 255) / 256 < 1024 ? (a.n4 + 255) / 256 : 1024);
  RSX_LAUNCH(cin_dx0_reduce_k, dim3(blocks), dim3(256), 0, stream, a);
  RSX_CHECK_LAUNCH();
  return RSX_OK;
}
