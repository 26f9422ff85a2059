// xDeepFM CIN layer on the bf16 matrix cores with SPLIT operands (round 5).
// Reference call site: xdeepfm/xdeepfm.py:145-172 (formulation: cin.hip).
//
// v_mfma_f32_16x16x32_bf16 runs at 16x the rate of v_mfma_f32_16x16x4_f32 (2.5 PFLOP/s against 157 TFLOP/s), multiplies its
// 8-bit-significand operands EXACTLY and accumulates in fp32.  An fp32 number is the exact sum of three bf16 numbers
// (x = x1 + x2 + x3, x1 = bf16(x), x2 = bf16(x - x1), x3 = x - x1 - x2: 8 + 8 + 8 significand bits), so a product of two fp32
// numbers is the sum of nine exact bf16 products; the six with i + j <= 4 carry everything above 2^-24 of the product:
//   NS = 3   x1y1 + (x1y2 + x2y1) + (x1y3 + x2y2 + x3y1): fp32-grade (|dropped| < 3 * 2^-24 |xy|), 6 MFMAs = 3/8 of the fp32
//            MFMA's time for the same k -- this is the PARITY path of the bf16 matrix cores (1e-5 against the oracle)
//   NS = 2   x1y1 + x1y2 + x2y1: 2^-16-grade products (3 MFMAs)
//   NS = 1   x1y1: plain bf16 operands (what cin_bf16.hip computes)
// Everything that is not a contraction operand (X0 row scaling, bias, relu, every sum) is fp32, as in the other two paths.
//
// Work split (forward; the backward mirrors it): a workgroup of 8 waves owns 8 examples and one 16-wide tile of outputs n,
// and walks the fields two at a time -- waves 0,2,4,6 take the even field, waves 1,3,5,7 the odd one, wave pair p the
// examples 2p, 2p + 1.  The filter fragments of a step (2 fields x NS planes x KS k-steps, 1 KiB each) come through LDS
// once per workgroup (double buffered, one barrier per step, the next step's loads in flight during the MFMAs), so a
// fragment is read from L2 once per 8 examples; the Xk operand of a wave's two examples is split once and stays in
// registers (NS x 2 x KS quads).  Per step and wave: 2 examples x (1 | 3 | 6) terms x KS MFMAs.
// Sums in fixed order (terms smallest first, k-steps, fields in step order, the two field parities at the end).
#include "rsx_common.h"
#include "gather_two_device.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16_t;
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) float gfloat_t;

namespace {

constexpr int CS_D = 16;
constexpr int CS_FP = 40;       // fields, padded (X0 reads as zero past F): F <= 40
constexpr int CS_MAXJ = 4;

__device__ __forceinline__ f32x4 mfma_bf16(bf16x8 a, bf16x8 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ uint32_t pack2(float lo, float hi) {
  bf16x2 v;
  v[0] = (bf16_t)lo;
  v[1] = (bf16_t)hi;
  return __builtin_bit_cast(uint32_t, v);
}
inline int rup(int x, int m) { return (x + m - 1) / m * m; }

// two fp32 values -> NS packed bf16 pairs: plane s holds bf16 of what the planes before it left over
template <int NS>
__device__ __forceinline__ void split2(float lo, float hi, uint32_t (&out)[NS]) {
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    const uint32_t pk = pack2(lo, hi);
    out[s] = pk;
    if (s + 1 < NS) {
      lo -= __uint_as_float(pk << 16);
      hi -= __uint_as_float(pk & 0xffff0000u);
    }
  }
}
// eight fp32 values (two float4) -> NS operand quads
template <int NS>
__device__ __forceinline__ void split8(float4 a, float4 b, bf16x8 (&out)[NS]) {
  uint32_t p0[NS], p1[NS], p2[NS], p3[NS];
  split2<NS>(a.x, a.y, p0);
  split2<NS>(a.z, a.w, p1);
  split2<NS>(b.x, b.y, p2);
  split2<NS>(b.z, b.w, p3);
#pragma unroll
  for (int s = 0; s < NS; ++s) out[s] = __builtin_bit_cast(bf16x8, (u32x4){p0[s], p1[s], p2[s], p3[s]});
}

// T += sum over the kept terms (smallest first) and the k-steps of A-plane x B-plane
// RSX_CIN_DBG (probe builds only, scripts/cin_split_where.sh; results are WRONG): 1 = one VALU fma in place of every MFMA, 2 = no
// global filter loads inside the field loop, 3 = no barrier inside the field loop
#ifndef RSX_CIN_DBG
#define RSX_CIN_DBG 0
#endif
template <int NS, int KS>
__device__ __forceinline__ f32x4 split_mma(const bf16x8 (&a)[NS][KS], const bf16x8 (&b)[NS][KS], f32x4 T) {
#pragma unroll
  for (int lvl = NS - 1; lvl >= 0; --lvl)          // lvl = i + j (zero based): 2^-8lvl relative size
#pragma unroll
    for (int sa = 0; sa <= lvl; ++sa)
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
#if RSX_CIN_DBG == 1
        T[0] = __builtin_fmaf(__builtin_bit_cast(f32x4, a[sa][ks])[0], __builtin_bit_cast(f32x4, b[lvl - sa][ks])[0], T[0]);
#else
        T = mfma_bf16(a[sa][ks], b[lvl - sa][ks], T);
#endif
      }
  return T;
}
template <int NS, int KS>
__device__ __forceinline__ f32x4 split_mma_ba(const bf16x8 (&a)[NS][KS], const bf16x8 (&b)[NS][KS], f32x4 T) {   // (b as the MFMA's first operand)
#pragma unroll
  for (int lvl = NS - 1; lvl >= 0; --lvl)
#pragma unroll
    for (int sa = 0; sa <= lvl; ++sa)
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) T = mfma_bf16(b[lvl - sa][ks], a[sa][ks], T);
  return T;
}

// ------------------------------------------------------------------------------------------------ filter preparation
// Fragment-major bf16 images, NS planes each (plane stride = the image size):
//   W16  [NS][F][H16/16][Np/32][64][8]: element j of lane (i, kq) = W[f][h = 16 ht + i][n = 32 ks + 8 kq + j]   (A of dX)
//   Wt16 [NS][F][N16/16][Hp/32][64][8]: element j of lane (i, kq) = W[f][h = 32 ks + 8 kq + j][n = 16 nt + i]   (B of fwd)
struct CsPrepJob { const float* W; bf16_t* W16; bf16_t* Wt16; float* winv; int H, N, H16, N16, Hp, Np; long long end; };
struct CsPrepArgs {
  CsPrepJob job[CS_MAXJ];
  int njobs, F;
  // rider (rsx_cin_split_prep_gather): the first n_gather workgroups (4 waves = 4 examples each) run xdeepfm.py's two input_layer
  // lookups + the linear_net pre-activation -- the launch before this one on the step's chain, as short as this one
  int n_gather;
  GatherTwoArgs gt;
};
// (256-thread workgroups, D = 16; a macro: the fields are read off the kernel's own by-value parameter)
#define CS_PREP_GATHER_ROLE(P)                                                                \
  if ((int)blockIdx.x < (P).n_gather) {                                                       \
    const int b_ = (int)blockIdx.x * 4 + (int)(threadIdx.x >> 6);                             \
    if (b_ < (P).gt.B) RSX_GATHER_TWO_EXAMPLE(16, (P).gt, b_, (int)(threadIdx.x & 63));       \
    return;                                                                                   \
  }
template <int NS>
__global__ __launch_bounds__(256) void cin_split_prep_k(const CsPrepArgs p) {
  CS_PREP_GATHER_ROLE(p)
  const long long total = p.job[p.njobs - 1].end;           // operand quads (of one plane) over all jobs
  for (long long e = (long long)((int)blockIdx.x - p.n_gather) * 256 + threadIdx.x; e < total; e += (long long)((int)gridDim.x - p.n_gather) * 256) {
    int ji = 0;
#pragma unroll
    for (int k = 1; k < CS_MAXJ; ++k)
      if (k < p.njobs && e >= p.job[k - 1].end) ji = k;
    const CsPrepJob& jb = p.job[ji];
    const long long le = e - (ji ? p.job[ji - 1].end : 0);
    const long long n1 = ((long long)p.F * jb.H16 * jb.Np) >> 3, n2 = ((long long)p.F * jb.N16 * jb.Hp) >> 3;
    const bool first = le < n1;
    const long long q = first ? le : le - n1;
    const int lane = (int)(q & 63);
    int r = (int)(q >> 6);
    const int KS = first ? jb.Np >> 5 : jb.Hp >> 5;
    const int ks = r % KS;
    r /= KS;
    const int T = first ? jb.H16 >> 4 : jb.N16 >> 4;
    const int t = r % T, f = r / T;
    float v[8];
    if (first) {                                   // W16: h = 16 t + (lane & 15), n = 32 ks + 8 (lane >> 4) + j
      const int h = 16 * t + (lane & 15), n0 = 32 * ks + 8 * (lane >> 4);
      const float* src = jb.W + ((size_t)f * jb.H + (h < jb.H ? h : jb.H - 1)) * jb.N;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int n = n0 + j;
        v[j] = src[n < jb.N ? n : jb.N - 1] * ((h < jb.H && n < jb.N) ? 1.f : 0.f);
      }
    } else {                                       // Wt16: n = 16 t + (lane & 15), h = 32 ks + 8 (lane >> 4) + j
      const int n = 16 * t + (lane & 15), h0 = 32 * ks + 8 * (lane >> 4);
      const float* src = jb.W + (size_t)f * jb.H * jb.N + (n < jb.N ? n : jb.N - 1);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int h = h0 + j;
        v[j] = src[(size_t)(h < jb.H ? h : jb.H - 1) * jb.N] * ((h < jb.H && n < jb.N) ? 1.f : 0.f);
      }
    }
    bf16x8 o[NS];
    split8<NS>(make_float4(v[0], v[1], v[2], v[3]), make_float4(v[4], v[5], v[6], v[7]), o);
    bf16_t* dst = (first ? jb.W16 : jb.Wt16) + q * 8;
    const size_t plane = (first ? n1 : n2) * 8;
#pragma unroll
    for (int s = 0; s < NS; ++s) *reinterpret_cast<bf16x8*>(dst + (size_t)s * plane) = o[s];
  }
}

// X0 of the workgroup's E examples -> LDS [E][CS_FP * 16], zeros for the fields past F: 3 float4 per thread (E * 160 items over
// 64 E threads), requested together
template <int E>
struct StageX0 {
  static constexpr int NTHR = 64 * E;
  float4 v[3];
  __device__ __forceinline__ void load(const float* X0, int b0, int B, int F, int tid) {
#pragma unroll
    for (int u = 0; u < 3; ++u) {
      const int e4 = tid + NTHR * u;
      const int ex = e4 / (CS_FP * 4), r = e4 % (CS_FP * 4);
      // (unconditional loads from clamped addresses, zeroed afterwards: a load under a condition becomes a branch, and the
      // compiler waits for each of them in turn)
      const int exc = ex < E ? ex : E - 1;
      const bool ok = e4 < E * CS_FP * 4 && (r >> 2) < F && b0 + ex < B;
      const int bc = b0 + exc < B ? b0 + exc : B - 1, rc = (r >> 2) < F ? r : 0;
      const float4 t = reinterpret_cast<const float4*>(X0 + (size_t)bc * F * CS_D)[rc];
      v[u] = make_float4(ok ? t.x : 0.f, ok ? t.y : 0.f, ok ? t.z : 0.f, ok ? t.w : 0.f);
    }
  }
  __device__ __forceinline__ void store(float* sX0, int tid) const {
#pragma unroll
    for (int u = 0; u < 3; ++u) {
      const int e4 = tid + NTHR * u;
      if (e4 < E * CS_FP * 4) reinterpret_cast<float4*>(sX0)[e4] = v[u];
    }
  }
};

// [E][rows][16] fp32 (through `ld(e, row, quarter)`, zeros past the real rows) -> LDS, transposed to dst[e][d][RP] fp32 (rows
// contiguous: what the MFMA's k index walks), RP = 32 KS + 4.  2 KS float4 per thread, all requested before the first store;
// lanes = (quarter fastest, row): a wave's 64 scalar stores hit 64 banks.
template <int KS, int E>
struct StageRowsF32 {
  static constexpr int RP = 32 * KS + 4, NTHR = 64 * E;
  float4 v[2 * KS];
  template <typename Load>
  __device__ __forceinline__ void load(int tid, Load ld) {
#pragma unroll
    for (int u = 0; u < 2 * KS; ++u) {
      const int it = tid + NTHR * u;
      const int dq = it & 3, row = (it >> 2) % (32 * KS), e = it / (128 * KS);
      v[u] = ld(e, row, dq);
    }
  }
  __device__ __forceinline__ void store(float* dst, int tid) const {
#pragma unroll
    for (int u = 0; u < 2 * KS; ++u) {
      const int it = tid + NTHR * u;
      const int dq = it & 3, row = (it >> 2) % (32 * KS), e = it / (128 * KS);
      float* t = dst + ((size_t)e * 16 + dq * 4) * RP + row;
      t[0 * RP] = v[u].x;
      t[1 * RP] = v[u].y;
      t[2 * RP] = v[u].z;
      t[3 * RP] = v[u].w;
    }
  }
};

// The filter fragments of one step: 2 fields x NS planes x KS k-steps, 1 KiB each, contiguous in LDS in that order, copied by
// global_load_lds_dwordx4 (L2 -> LDS without staging registers or a ds_write pass: the register-staged form spent 13 LDS cycles
// per 1-KiB ds_write_b128 AFTER the step's MFMAs, before its barrier).  Item = one 16-byte lane quad, U per thread; a wave's 64
// items are one fragment = 1 KiB contiguous on both sides (the LDS destination of an LDS-DMA is wave-uniform base + lane * 16).
// f >= F is clamped (its X0 is zero); a ring slot is padded to U * NTHR quads (no conditional issue).
typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void glb_void_t;
template <int NS, int KS, int E>
struct StageW {
  static constexpr int FRAGS = 2 * NS * KS, NTHR = 64 * E;
  static constexpr int U = (FRAGS * 64 + NTHR - 1) / NTHR;
  static constexpr int SLOT = U * NTHR * 8;        // bf16 elements per ring slot
  // base: image + (tile * KS) * 512 elements; fstride: elements per field; plane: elements per plane
  static __device__ __forceinline__ void issue(const bf16_t* base, size_t fstride, size_t plane, int f0, int F, int tid, bf16_t* slot) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int it = tid + NTHR * u;
      const int lane = it & 63, frag = (it >> 6) < FRAGS ? (it >> 6) : FRAGS - 1;
      const int ks = frag % KS, sp = (frag / KS) % NS, par = frag / (KS * NS);
      const int f = f0 + par < F ? f0 + par : F - 1;
      const bf16_t* src = base + (size_t)sp * plane + (size_t)f * fstride + (size_t)ks * 512 + lane * 8;
      __builtin_amdgcn_global_load_lds((glb_void_t*)src, (lds_void_t*)(slot + (size_t)(it - lane) * 8), 16, 0, 0);
    }
  }
};

constexpr size_t cs_max(size_t a, size_t b) { return a > b ? a : b; }
#if RSX_CIN_DBG == 2
#define CS_DBG_LOAD(X)
#else
#define CS_DBG_LOAD(X) X
#endif
#if RSX_CIN_DBG == 3
#define CS_DBG_BARRIER
#else
#define CS_DBG_BARRIER __syncthreads();
#endif
// LDS of the forward / data-gradient launches: sX0 | ring slot 0 | ring slot 1 | staged operand rows [E][16][32 KS + 4] f32 (dead
// once the operands are in registers; the waves' partial sums reuse it).  Where two 256-thread workgroups would not fit a CU
// side by side (80 KiB each), slot 1 ALIASES the staged rows and is filled after they were consumed (one exposed L2 round
// trip in the prologue).
template <int NS, int KS, int E>
struct CsLds {
  static constexpr size_t X0 = (size_t)E * CS_FP * CS_D * 4, SLOT = (size_t)StageW<NS, KS, E>::SLOT * 2;
  static constexpr size_t ROWS = cs_max((size_t)E * 16 * (32 * KS + 4) * 4, (size_t)E * 2 * 256 * 4);
  static constexpr bool ALIAS = E == 4 && X0 + 2 * SLOT + ROWS > 80 * 1024;
  static constexpr size_t TOTAL = X0 + SLOT + (ALIAS ? cs_max(SLOT, ROWS) : SLOT + ROWS);
};

// ------------------------------------------------------------------------------------------------------------ forward
struct CsFwdArgs {
  const float* X0;      // [B, F, 16]
  const float* Xk;      // [B, H, 16]
  const bf16_t* Wt16;   // NS planes, see cin_split_prep_k
  const float* c;       // [N]
  float* out;           // [B, N, 16]
  int B, F, H, N, N16, Hp;
  const float* winv;    // [F] inverse scales of the filter planes (mode 4)
};

// grid = (N16 / 16, ceil(B / E)), block = 64 E (E = 4: two workgroups per CU, their barriers independent -- one's MFMAs cover
// the other's loads, LDS traffic and barrier waits; E = 8: one workgroup per CU, half the filter traffic from L2).
// dyn LDS: CsLds.
template <int NS, int KS, int E>
__global__ __launch_bounds__(64 * E, E == 4 ? 2 : 1) void cin_split_fwd_k(const CsFwdArgs p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int HP = 32 * KS + 4, SLOT = StageW<NS, KS, E>::SLOT;
  float* sX0 = lds;                                                   // [E][CS_FP*16]
  bf16_t* sW0 = reinterpret_cast<bf16_t*>(sX0 + E * CS_FP * CS_D);    // ring slot 0
  bf16_t* sW1 = sW0 + SLOT;                                           // ring slot 1
  constexpr bool ALIAS = CsLds<NS, KS, E>::ALIAS;
  float* sXk = reinterpret_cast<float*>(ALIAS ? sW1 : sW1 + SLOT);    // [E][16][HP]
  float* sR = sXk;                                                    // [E waves][2][4][64] at the end
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int i = lane & 15, kq = lane >> 4;
  const int par = wv & 1, e0 = (wv >> 1) * 2;                         // field parity, first of the wave's two examples
  const int n0 = blockIdx.x * 16, b0 = blockIdx.y * E;
  const bf16_t* wbase = p.Wt16 + (size_t)blockIdx.x * KS * 512;
  const size_t fstride = (size_t)p.N16 * p.Hp, plane = (size_t)p.F * fstride;
  const int nstep = (p.F + 1) / 2;
  using GW = StageW<NS, KS, E>;
  GW::issue(wbase, fstride, plane, 0, p.F, tid, sW0);
  if constexpr (!ALIAS) GW::issue(wbase, fstride, plane, 2, p.F, tid, sW1);
  StageX0<E> sx;
  sx.load(p.X0, b0, p.B, p.F, tid);
  {
    StageRowsF32<KS, E> sr;
    sr.load(tid, [&](int e, int h, int dq) {
      const bool ok = b0 + e < p.B && h < p.H;
      const int bc = b0 + e < p.B ? b0 + e : p.B - 1, hc = h < p.H ? h : p.H - 1;
      const float4 t = reinterpret_cast<const float4*>(p.Xk + ((size_t)bc * p.H + hc) * CS_D)[dq];
      return make_float4(ok ? t.x : 0.f, ok ? t.y : 0.f, ok ? t.z : 0.f, ok ? t.w : 0.f);
    });
    sx.store(sX0, tid);
    sr.store(sXk, tid);
  }
  __syncthreads();
  bf16x8 a[2][NS][KS];                             // Xk[b0 + e0 + e][h = 32 ks + 8 kq + j][d = i], split
#pragma unroll
  for (int e = 0; e < 2; ++e)
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const float4* src = reinterpret_cast<const float4*>(sXk + ((size_t)(e0 + e) * 16 + i) * HP + 32 * ks + 8 * kq);
      bf16x8 t[NS];
      split8<NS>(src[0], src[1], t);
#pragma unroll
      for (int s = 0; s < NS; ++s) a[e][s][ks] = t[s];
    }
  f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
  // Step st multiplies the fragments its predecessor read from ring slot st & 1 into registers, while (a) the reads of step
  // st + 1's fragments (the other slot, complete since the last barrier) and (b) the LDS-DMA of step st + 2's into slot st & 1
  // -- every wave's reads of it completed before the last barrier -- are in flight; the step's barrier waits for (b).  Two
  // steps per trip: the two register sets swap roles without copies.  A step past the last field multiplies by X0 = 0.
  // (macros, not lambdas: captured by reference the staging registers and the fragment sets go to scratch memory)
#define CS_READ_W(ST, W)                                                                                              \
  {                                                                                                                   \
    const bf16_t* wb_ = (((ST) & 1) ? sW1 : sW0) + (par * NS * KS) * 512 + lane * 8;                                  \
    _Pragma("unroll") for (int s_ = 0; s_ < NS; ++s_)                                                                 \
      _Pragma("unroll") for (int ks_ = 0; ks_ < KS; ++ks_)                                                            \
        W[s_][ks_] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(wb_ + (s_ * KS + ks_) * 512));        \
  }
#define CS_FWD_STEP(ST, W, WN)                                                                                        \
  {                                                                                                                   \
    CS_DBG_LOAD(GW::issue(wbase, fstride, plane, 2 * ((ST) + 2), p.F, tid, ((ST) & 1) ? sW1 : sW0);)                  \
    CS_READ_W((ST) + 1, WN)                                                                                           \
    const int f_ = 2 * (ST) + par;                                                                                    \
    const float m_ = f_ < CS_FP ? 1.f : 0.f;                                                                          \
    _Pragma("unroll") for (int e = 0; e < 2; ++e) {                                                                   \
      const f32x4 T = split_mma<NS, KS>(a[e], W, (f32x4){0.f, 0.f, 0.f, 0.f});                                        \
      const float4 x = *reinterpret_cast<const float4*>(sX0 + ((e0 + e) * CS_FP + (f_ < CS_FP ? f_ : CS_FP - 1)) * CS_D + kq * 4); \
      acc[e][0] = __builtin_fmaf(x.x * m_, T[0], acc[e][0]);                                                          \
      acc[e][1] = __builtin_fmaf(x.y * m_, T[1], acc[e][1]);                                                          \
      acc[e][2] = __builtin_fmaf(x.z * m_, T[2], acc[e][2]);                                                          \
      acc[e][3] = __builtin_fmaf(x.w * m_, T[3], acc[e][3]);                                                          \
    }                                                                                                                 \
    CS_DBG_BARRIER /* (its fence waits for the step's LDS-DMA: vmcnt(0)) */                                           \
  }
  bf16x8 w0[NS][KS], w1[NS][KS];
  CS_READ_W(0, w0)
  __syncthreads();                                 // every wave has its operands (sXk is free) and slot 0's fragments
  if constexpr (ALIAS) {
    GW::issue(wbase, fstride, plane, 2, p.F, tid, sW1);
    __syncthreads();
  }
  for (int st = 0; st < nstep; st += 2) {
    CS_FWD_STEP(st, w0, w1)
    CS_FWD_STEP(st + 1, w1, w0)
  }
#undef CS_FWD_STEP
#undef CS_READ_W
#pragma unroll
  for (int e = 0; e < 2; ++e)
#pragma unroll
    for (int r = 0; r < 4; ++r) sR[((wv * 2 + e) * 4 + r) * 64 + lane] = acc[e][r];
  __syncthreads();
  {   // wave w finishes example w: even fields' sum + odd fields' sum + bias, relu
    const int ex = wv, b = b0 + ex;
    const int w0i = (ex >> 1) * 2, sl = ex & 1;
    const bool nok = n0 + i < p.N;
    const float cv = p.c[nok ? n0 + i : 0];
    float o[4];
#pragma unroll
    for (int r = 0; r < 4; ++r)
      o[r] = fmaxf((sR[((w0i * 2 + sl) * 4 + r) * 64 + lane] + sR[(((w0i + 1) * 2 + sl) * 4 + r) * 64 + lane]) + cv, 0.f);
    if (nok && b < p.B)
      *reinterpret_cast<float4*>(p.out + ((size_t)b * p.N + n0 + i) * CS_D + kq * 4) = make_float4(o[0], o[1], o[2], o[3]);
  }
}

// ------------------------------------------------------------------------ deep-ring kernels (the default): forward, dXk / dX0
// What bounded cin_split_fwd_k / cin_split_dx_k (profiles/r05_y_cin_split_where.txt: 18 us with every MFMA replaced by one VALU
// op, 12.8 us of MFMA issue at 2.4 GHz, 29 us together):
//   * the compiler cannot tell an LDS-DMA's destination from the slot a ds_read reads and puts `s_waitcnt vmcnt(0)` in front of
//     the first LDS read after every DMA issue: the prefetch distance of the ring was never more than the rest of ONE step;
//   * the whole-step register double buffer (96 VGPRs of filter fragments) did not survive register allocation: the reads were
//     sunk to their uses, every few MFMAs waited for an LDS read issued just before them;
//   * the DMA addresses cost ~60 VALU instructions per step (64-bit multiplies of clamped field numbers) between dependent MFMAs;
//   * and the clock under this load is 2.0 GHz (SQ_BUSY_CYCLES / 32 shader engines / duration), not 2.4: 15.4 us of MFMA issue.
// Here: ONE workgroup of 8 waves per CU and 8 examples (a fragment is read from L2 once per 8 examples), a ring of R = 5..8
// slots, the DMA issued from inline assembly (invisible to the compiler's LDS alias check; addresses = SGPR base + lane * 16, all
// scalar), the step's barrier waits for `vmcnt((R - 3) U)` -- a piece has R - 2 whole steps to land -- and the filter fragments go
// through a ring of PF <= 6 register quads, each requested PF - 2 fragments ahead of its use (the next step's first ones before
// the barrier).  A fragment's MFMAs go to one accumulator per (example, magnitude level): 2 NS independent chains; the levels
// are added smallest first at the end of the step.  Stamps + probe builds (scripts/stamp_probe_cin_split.py): the field loop is
// MFMA issue now -- without DMA, barrier or fragment reads it is 4-14 % shorter, without the MFMAs half.
//
// MODE 4 ("h2"): the same contraction with TWO fp16 planes per operand (x = x1 + x2, 11 + 11 significand bits; the operand of
// one accumulation chain is scaled by a power of two so that its largest element sits at 2^14: per example for Xk / dpre, per
// field for the filters, the scales divided out in fp32 after the chain) and the three products x1y1 + x1y2 + x2y1 -- HALF the
// MFMAs of the three-plane bf16 form.  What it drops (x2y2, the planes' own rounding) is below 2^-22 of a product for every
// element within 2^17 of its chain's largest and 2^-39 of the largest product below that: 6e-8 of the layer's output in the
// tests, under the fp32 accumulation's own 2e-6.  The weight gradients stay on three bf16 planes (their chain runs over the
// batch: no per-example scale can be divided out).
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
constexpr int CS_H2 = 4;

template <int MODE>
struct SplitMode {                                   // MODE = 1..3: bf16 planes
  static constexpr int NS = MODE;
  static constexpr bool SCALED = false;
  typedef bf16x8 quad;
  static __device__ __forceinline__ f32x4 mma(quad a, quad b, f32x4 c) { return mfma_bf16(a, b, c); }
  static __device__ __forceinline__ void split(float4 a, float4 b, quad (&out)[NS]) { split8<NS>(a, b, out); }
};
template <>
struct SplitMode<CS_H2> {
  static constexpr int NS = 2;
  static constexpr bool SCALED = true;
  typedef f16x8 quad;
  static __device__ __forceinline__ f32x4 mma(quad a, quad b, f32x4 c) {
#ifdef RSX_CIN_H2_AS_BF16      // (probe builds only, WRONG results: is the fp16 MFMA itself slower than the bf16 one under load?)
    return mfma_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c);
#else
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
#endif
  }
  static __device__ __forceinline__ void split2h(float lo, float hi, uint32_t (&out)[2]) {
    f16x2 p, q;
    p[0] = (_Float16)lo;
    p[1] = (_Float16)hi;
    q[0] = (_Float16)(lo - (float)p[0]);
    q[1] = (_Float16)(hi - (float)p[1]);
    out[0] = __builtin_bit_cast(uint32_t, p);
    out[1] = __builtin_bit_cast(uint32_t, q);
  }
  static __device__ __forceinline__ void split(float4 a, float4 b, quad (&out)[2]) {
    uint32_t p0[2], p1[2], p2[2], p3[2];
    split2h(a.x, a.y, p0);
    split2h(a.z, a.w, p1);
    split2h(b.x, b.y, p2);
    split2h(b.z, b.w, p3);
#pragma unroll
    for (int s = 0; s < 2; ++s) out[s] = __builtin_bit_cast(f16x8, (u32x4){p0[s], p1[s], p2[s], p3[s]});
  }
};
// power-of-two scale that puts `mx` (>= 0) into [2^14, 2^15), and its inverse
__device__ __forceinline__ void cs_pow2_scale(float mx, float& scale, float& inv) {
  int e = (int)((__float_as_uint(mx) >> 23) & 0xffu);
  e = e < 16 ? 16 : (e > 254 ? 254 : e);
  scale = __uint_as_float((uint32_t)(268 - e) << 23);
  inv = __uint_as_float((uint32_t)(e - 14) << 23);
}
__device__ __forceinline__ float cs_wave_max(float v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v = fmaxf(v, __shfl_xor(v, m));
  return v;
}

__device__ __forceinline__ void cs_dma16(const void* sbase, uint32_t voff, uint32_t lds_addr) {
  // (M0 = wave-uniform LDS byte address of the piece; the hardware adds lane * 16.  The base goes through readfirstlane: an "s"
  // operand the compiler could not prove uniform is handed over in VGPRs, which the instruction's SGPR-base form rejects)
  const uint64_t b = reinterpret_cast<uint64_t>(sbase);
  const uint64_t bs = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(b >> 32)) << 32) |
                      (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)b);
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(bs),
               "s"(__builtin_amdgcn_readfirstlane((int)lds_addr)) : "memory");
}
template <int VM>
__device__ __forceinline__ void cs_wait_barrier() {
  asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(VM) : "memory");
}

template <int NS, int KS, int EXTRA>
struct Cs8 {
  static constexpr int E = 8, NTHR = 512;
  static constexpr int FR = NS * KS;                          // fragments of one field
  static constexpr int FRAGS = 2 * FR;                        // of one step (two fields)
  static constexpr int U = (FRAGS + 7) / 8;                   // DMA pieces per wave and step
  static constexpr int SLOTB = U * 8 * 1024;                  // bytes per ring slot (padded to whole rounds of the 8 waves)
  static constexpr int X0B = E * CS_FP * CS_D * 4;
  static constexpr int RMAX = (160 * 1024 - X0B - EXTRA) / SLOTB;
  static constexpr int R = RMAX > 8 ? 8 : RMAX;
  // register ring of filter fragments: the largest divisor of FR up to 6 (FR % PF == 0: a fragment's quad is the same every step)
  static constexpr int PF = FR % 6 == 0 ? 6 : (FR % 4 == 0 ? 4 : (FR % 3 == 0 ? 3 : (FR % 2 == 0 ? 2 : 1)));
  static constexpr int VM = (R - 3) * U;                      // pieces that may still be in flight at a step's barrier
  static constexpr size_t TOTAL = (size_t)X0B + (size_t)R * SLOTB + EXTRA;
  static_assert(R >= 3, "the ring needs three slots");
  static_assert(VM <= 63, "vmcnt is a 6-bit counter");
  // the pieces of step `stn` into ring slot `sl`: wave wv takes fragments wv, wv + 8, ... (clamped: the slot's padding)
  static __device__ __forceinline__ void issue(const char* img, uint32_t fstrideB, uint32_t planeB, int stn, int F, int wv,
                                               uint32_t lane16, uint32_t slot_lds) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int fi = wv + 8 * u;
      const int fr = fi < FRAGS ? fi : FRAGS - 1;
      const int ks = fr % KS, sp = (fr / KS) % NS, par = fr / FR;
      const int f = 2 * stn + par < F ? 2 * stn + par : F - 1;          // (a field past F: its X0 is zero)
      cs_dma16(img + (size_t)sp * planeB + (size_t)f * fstrideB + (size_t)ks * 1024, lane16, slot_lds + (uint32_t)fi * 1024u);
    }
  }
};

// One step's MFMAs of a wave: the FR fragments of its field (ring slot `cur`, from the register ring w) against the split
// operands of its two examples, T[e][level] += ...; WA: the filter fragment is the MFMA's first operand (data gradients).
template <int MODE, int KS, int PF, bool WA>
__device__ __forceinline__ void cs8_step_mma(const typename SplitMode<MODE>::quad (&a)[2][SplitMode<MODE>::NS][KS],
                                             typename SplitMode<MODE>::quad (&w)[PF], const char* cur, const char* nxt,
                                             f32x4 (&T)[2][SplitMode<MODE>::NS]) {
  using M = SplitMode<MODE>;
  constexpr int NS = M::NS, FR = NS * KS;
#pragma unroll
  for (int e = 0; e < 2; ++e)
#pragma unroll
    for (int l = 0; l < NS; ++l) T[e][l] = (f32x4){0.f, 0.f, 0.f, 0.f};
  // (the scheduling fences keep the reads where they are written: left alone, the compiler sinks every LDS read to its first use)
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int j = 0; j < FR; ++j) {                   // fragment j = (k-step j / NS, plane j % NS) sits in register quad j % PF
    const int ks = j / NS, s = j % NS;
#pragma unroll
    for (int e = 0; e < 2; ++e)
#pragma unroll
      for (int sa = 0; sa + s < NS; ++sa) {
#if RSX_CIN_DBG == 1
        T[e][s + sa][0] = __builtin_fmaf(__builtin_bit_cast(f32x4, a[e][sa][ks])[0], __builtin_bit_cast(f32x4, w[j % PF])[0], T[e][s + sa][0]);
#else
        T[e][s + sa] = WA ? M::mma(w[j % PF], a[e][sa][ks], T[e][s + sa]) : M::mma(a[e][sa][ks], w[j % PF], T[e][s + sa]);
#endif
      }
    __builtin_amdgcn_sched_barrier(0);
    // the quad of fragment j - 1 (its MFMAs were issued a fragment ago) takes fragment j - 1 + PF: of this step, or the next one's
    constexpr int LAG = PF >= 2 ? 1 : 0;           // (a single quad is refilled straight after its use)
    const int jn = j - LAG + PF;
    const int jj = jn < FR ? jn : jn - FR;
    const char* src = (jn < FR ? cur : nxt) + ((jj % NS) * KS + jj / NS) * 1024;
#if RSX_CIN_DBG != 4
    w[jn % PF] = __builtin_bit_cast(typename M::quad, *reinterpret_cast<const uint4*>(src));
#endif
    __builtin_amdgcn_sched_barrier(0);
  }
}
#if RSX_CIN_DBG == 2
#define CS8_STEP_BARRIER(C8) cs_wait_barrier<0>()
#elif RSX_CIN_DBG == 3
#define CS8_STEP_BARRIER(C8) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#else
#define CS8_STEP_BARRIER(C8) cs_wait_barrier<C8::VM>()
#endif

// grid = (N16 / 16, ceil(B / 8)), block = 512, dyn LDS = Cs8<NS, KS, SCALED ? 256 : 0>::TOTAL: sX0 [8][CS_FP * 16] f32 | R ring slots | the
// filters' inverse scales [CS_FP] (MODE 4)
template <int MODE, int KS>
__global__ __launch_bounds__(512, 1) void cin_split_fwd8_k(const CsFwdArgs p) {
  using M = SplitMode<MODE>;
  constexpr int NS = M::NS;
  using C8 = Cs8<NS, KS, M::SCALED ? 256 : 0>;
  constexpr int E = 8, R = C8::R, PF = C8::PF, FR = C8::FR, SLOTB = C8::SLOTB;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* sX0 = lds;
  char* ring = reinterpret_cast<char*>(lds) + C8::X0B;
  float* sInv = reinterpret_cast<float*>(ring + (size_t)R * SLOTB);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i = lane & 15, kq = lane >> 4;
  const int par = wv & 1, e0 = (wv >> 1) * 2;
  const int n0 = blockIdx.x * 16, b0 = blockIdx.y * E;
  const uint32_t fstrideB = (uint32_t)p.N16 * p.Hp * 2u, planeB = (uint32_t)p.F * fstrideB;
  const char* img = reinterpret_cast<const char*>(p.Wt16) + (size_t)blockIdx.x * KS * 1024;
  const uint32_t ring_lds = (uint32_t)(uintptr_t)ring, lane16 = (uint32_t)lane * 16u;
  const int nstep = (p.F + 1) / 2;
  const bool st0 = KS == 4 && blockIdx.x == 0 && blockIdx.y == 0;
  RSX_STAMP(0, st0); RSX_STAMP_MAX(16, KS == 4);
#pragma unroll
  for (int s = 0; s < R - 1; ++s) C8::issue(img, fstrideB, planeB, s < nstep ? s : nstep - 1, p.F, wv, lane16, ring_lds + s * SLOTB);
  StageX0<E> sx;
  sx.load(p.X0, b0, p.B, p.F, tid);
  if (M::SCALED && tid < CS_FP) sInv[tid] = tid < p.F ? p.winv[tid] : 0.f;
  typename M::quad a[2][NS][KS];                   // Xk[b0 + e0 + e][h = 32 ks + 8 kq + j][d = i], split; straight from L2
  float inv_x[2] = {1.f, 1.f};
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    const int b = b0 + e0 + e;
    const float* xb = p.Xk + (size_t)(b < p.B ? b : p.B - 1) * p.H * CS_D + i;
    const float bm = b < p.B ? 1.f : 0.f;
    float v[KS][8];
    float mx = 0.f;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int h = 32 * ks + 8 * kq + j;
        v[ks][j] = xb[(size_t)(h < p.H ? h : p.H - 1) * CS_D];
      }
    // (every load requested before the first use: with the masks and the maximum folded into the load loop the scheduler kept
    // the source order -- load, wait, multiply -- and the prologue cost 15 us)
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        v[ks][j] *= (32 * ks + 8 * kq + j) < p.H ? bm : 0.f;
        if (M::SCALED) mx = fmaxf(mx, fabsf(v[ks][j]));
      }
    float sc = 1.f;
    if (M::SCALED) cs_pow2_scale(cs_wave_max(mx), sc, inv_x[e]);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      typename M::quad t[NS];
      M::split(make_float4(v[ks][0] * sc, v[ks][1] * sc, v[ks][2] * sc, v[ks][3] * sc),
               make_float4(v[ks][4] * sc, v[ks][5] * sc, v[ks][6] * sc, v[ks][7] * sc), t);
#pragma unroll
      for (int s = 0; s < NS; ++s) a[e][s][ks] = t[s];
    }
  }
  RSX_STAMP(1, st0);
  sx.store(sX0, tid);
  cs_wait_barrier<0>();                            // sX0 and the first R - 1 slots
  RSX_STAMP(2, st0);
  const char* rd = ring + (size_t)par * FR * 1024 + lane16;      // this wave's fragments of slot 0
  typename M::quad w[PF];                          // at a step's entry: its fragments 0 .. PF - 2 (requested during the step before)
#pragma unroll
  for (int j = 0; j < PF; ++j)
    w[j] = __builtin_bit_cast(typename M::quad, *reinterpret_cast<const uint4*>(rd + ((j % NS) * KS + j / NS) * 1024));
  f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
  int sl = 0;                                      // ring slot of the current step
  for (int st = 0; st < nstep; ++st) {
    const int sln = sl + 1 == R ? 0 : sl + 1, slp = sl == 0 ? R - 1 : sl - 1;
    const int stn = st + R - 1 < nstep ? st + R - 1 : nstep - 1;
    CS_DBG_LOAD(C8::issue(img, fstrideB, planeB, stn, p.F, wv, lane16, ring_lds + slp * SLOTB);)      // (slot of step st - 1: free since its barrier)
    const int f_ = 2 * st + par;
    float4 x[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) x[e] = *reinterpret_cast<const float4*>(sX0 + ((e0 + e) * CS_FP + f_) * CS_D + kq * 4);
    const float wi = M::SCALED ? sInv[f_] : 1.f;
    f32x4 T[2][NS];
    cs8_step_mma<MODE, KS, PF, false>(a, w, rd + (size_t)sl * SLOTB, rd + (size_t)sln * SLOTB, T);
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      f32x4 t = T[e][NS - 1];
#pragma unroll
      for (int l = NS - 2; l >= 0; --l) t += T[e][l];
      if (M::SCALED) x[e] = f4_scale(wi * inv_x[e], x[e]);
      acc[e][0] = __builtin_fmaf(x[e].x, t[0], acc[e][0]);
      acc[e][1] = __builtin_fmaf(x[e].y, t[1], acc[e][1]);
      acc[e][2] = __builtin_fmaf(x[e].z, t[2], acc[e][2]);
      acc[e][3] = __builtin_fmaf(x[e].w, t[3], acc[e][3]);
    }
    CS8_STEP_BARRIER(C8);
    sl = sln;
  }
  RSX_STAMP(3, st0);
  cs_wait_barrier<0>();                            // the tail's redundant pieces have landed: the ring is free
  RSX_STAMP(4, st0);
  float* sR = reinterpret_cast<float*>(ring);      // [8 waves][2][4][64]
#pragma unroll
  for (int e = 0; e < 2; ++e)
#pragma unroll
    for (int r = 0; r < 4; ++r) sR[((wv * 2 + e) * 4 + r) * 64 + lane] = acc[e][r];
  __syncthreads();
  {   // wave w finishes example w: even fields' sum + odd fields' sum + bias, relu
    const int ex = wv, b = b0 + ex;
    const int w0i = (ex >> 1) * 2, sl2 = ex & 1;
    const bool nok = n0 + i < p.N;
    const float cv = p.c[nok ? n0 + i : 0];
    float o[4];
#pragma unroll
    for (int r = 0; r < 4; ++r)
      o[r] = fmaxf((sR[((w0i * 2 + sl2) * 4 + r) * 64 + lane] + sR[(((w0i + 1) * 2 + sl2) * 4 + r) * 64 + lane]) + cv, 0.f);
    if (nok && b < p.B)
      *reinterpret_cast<float4*>(p.out + ((size_t)b * p.N + n0 + i) * CS_D + kq * 4) = make_float4(o[0], o[1], o[2], o[3]);
  }
  RSX_STAMP(5, st0); RSX_STAMP_MAX(17, KS == 4);
}

// Filter images of mode 4: CS_H2_PARTS workgroups per (layer, field) -- the field's largest |W| (a power-of-two scale puts it at 2^14; its
// inverse goes to winv[f]), then both fragment-major images of the field as two fp16 planes of W * scale.
constexpr int CS_H2_PARTS = 4;                     // workgroups per (layer, field): each finds the field's maximum, converts a quarter
// value j of operand quad lq of field Wf in layout `first` (W16: h = 16 t + (lane & 15), n = 32 ks + 8 (lane >> 4) + j; Wt16:
// n = 16 t + (lane & 15), h = 32 ks + 8 (lane >> 4) + j): clamped address + the mask the loaded value is multiplied by
__device__ __forceinline__ const float* cs_h2_src(const CsPrepJob& jb, const float* Wf, bool first, int lq, int j, float& m) {
  const int lane = lq & 63, r = lq >> 6;
  const int KS = first ? jb.Np >> 5 : jb.Hp >> 5;
  const int ks = r % KS, t = r / KS;
  const int a = 16 * t + (lane & 15), c = 32 * ks + 8 * (lane >> 4) + j;
  const int h = first ? a : c, n = first ? c : a;
  m = (h < jb.H && n < jb.N) ? 1.f : 0.f;
  return Wf + (size_t)(h < jb.H ? h : jb.H - 1) * jb.N + (n < jb.N ? n : jb.N - 1);
}
__global__ __launch_bounds__(256) void cin_split_prep_h2_k(const CsPrepArgs p) {
  __shared__ float red[4];
  CS_PREP_GATHER_ROLE(p)
  const int bid = (int)blockIdx.x - p.n_gather;
  const int part = bid % CS_H2_PARTS, lf = bid / CS_H2_PARTS;
  const int ji = lf / p.F, f = lf % p.F;
  CsPrepJob jb = p.job[0];                         // (compile-time indices into the by-value parameter: a dynamic one goes through scratch)
#pragma unroll
  for (int k = 1; k < CS_MAXJ; ++k)
    if (ji == k) jb = p.job[k];
  const int tid = threadIdx.x;
  const float* Wf = jb.W + (size_t)f * jb.H * jb.N;
  const int tot = jb.H * jb.N;
  // the field's largest |W|: ONE memory round trip (64 elements per thread requested together; the filters were written by the
  // previous step's optimizer launch and come from HBM)
  float mx = 0.f;
  if ((tot & 3) == 0 && (reinterpret_cast<uintptr_t>(Wf) & 15) == 0) {
    const int tot4 = tot >> 2;
    for (int e0 = tid; e0 < tot4; e0 += 256 * 16) {
      float4 t[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) t[u] = reinterpret_cast<const float4*>(Wf)[e0 + 256 * u < tot4 ? e0 + 256 * u : tot4 - 1];
#pragma unroll
      for (int u = 0; u < 16; ++u) mx = fmaxf(fmaxf(mx, fmaxf(fabsf(t[u].x), fabsf(t[u].y))), fmaxf(fabsf(t[u].z), fabsf(t[u].w)));
    }
  } else {
    for (int e0 = tid; e0 < tot; e0 += 256 * 16) {
      float t[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) t[u] = Wf[e0 + 256 * u < tot ? e0 + 256 * u : tot - 1];
#pragma unroll
      for (int u = 0; u < 16; ++u) mx = fmaxf(mx, fabsf(t[u]));
    }
  }
  mx = cs_wave_max(mx);
  if ((tid & 63) == 0) red[tid >> 6] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  float sc, inv;
  cs_pow2_scale(mx, sc, inv);
  if (tid == 0 && part == 0) jb.winv[f] = inv;
  const int n1 = (jb.H16 * jb.Np) >> 3, n2 = (jb.N16 * jb.Hp) >> 3;       // quads of this field in the two layouts
  const size_t pl1 = ((size_t)p.F * jb.H16 * jb.Np), pl2 = ((size_t)p.F * jb.N16 * jb.Hp);
  // one layout at a time (a per-quad choice of layout compiles to branches around the loads: every quad's loads were waited for
  // in turn), two quads per thread and trip, their 16 loads requested together (L2 hits: the max pass has just read the field)
#define CS_H2_LAYOUT(FIRST, NQ, DSTBASE, PLANE)                                                                        \
  for (int q0 = part * 256 + tid; q0 < (NQ); q0 += 2 * 256 * CS_H2_PARTS) {                                          \
    float v[2][8];                                                                                                   \
    _Pragma("unroll") for (int u = 0; u < 2; ++u) {                                                                  \
      const int q = q0 + u * 256 * CS_H2_PARTS;                                                                      \
      const int qc = q < (NQ) ? q : (NQ) - 1;                                                                        \
      _Pragma("unroll") for (int j = 0; j < 8; ++j) {                                                                \
        float m;                                                                                                     \
        const float* src = cs_h2_src(jb, Wf, FIRST, qc, j, m);                                                       \
        v[u][j] = *src * (m * sc);                                                                                   \
      }                                                                                                              \
    }                                                                                                                \
    _Pragma("unroll") for (int u = 0; u < 2; ++u) {                                                                  \
      const int q = q0 + u * 256 * CS_H2_PARTS;                                                                      \
      if (q < (NQ)) {                                                                                                \
        f16x8 o[2];                                                                                                  \
        SplitMode<CS_H2>::split(make_float4(v[u][0], v[u][1], v[u][2], v[u][3]), make_float4(v[u][4], v[u][5], v[u][6], v[u][7]), o); \
        bf16_t* dst = (DSTBASE) + ((size_t)f * (NQ) + q) * 8;   /* (quad index inside a plane of the layout) */          \
        _Pragma("unroll") for (int s2 = 0; s2 < 2; ++s2) *reinterpret_cast<f16x8*>(dst + (size_t)s2 * (PLANE)) = o[s2]; \
      }                                                                                                              \
    }                                                                                                                \
  }
  CS_H2_LAYOUT(true, n1, jb.W16, pl1)
  CS_H2_LAYOUT(false, n2, jb.Wt16, pl2)
#undef CS_H2_LAYOUT
}

// ------------------------------------------------------------------------------------------------ backward: dXk, dX0
struct CsDxArgs {
  const float* X0;      // [B, F, 16]
  const float* Xk;      // [B, H, 16]
  const bf16_t* W16;    // NS planes, see cin_split_prep_k
  const float* out;     // [B, N, 16] this layer's relu output
  const float* dout;    // [B, N, 16] gradient wrt the relu output (nullable when gs is given)
  const float* gs;      // [B] nullable: direct-connect gradient gs[b] * wout[n], broadcast over d, added to dout
  const float* wout;    // [N]
  float* dXk;           // [B, H, 16]
  float* dx0_parts;     // [HT][B][F*16] out: tile ht's share of dX0
  bf16_t* dpre16;       // [NS][ceil(B/2)][N16/16][64][8] out (workgroups of tile 0): the dW kernel's B fragments, split
  float* dc_part;       // [B][N16] out (workgroups of tile 0): per-example column sums of dpre
  int acc_dxk;
  int B, F, H, N, H16, N16, Np;
  const float* winv;    // [F] inverse scales of the filter planes (mode 4)
  int dw_planes;        // bf16 planes of dpre16 (what the weight-gradient launch multiplies)
};

// grid = (H16 / 16, ceil(B / E)), block = 64 E: workgroup = E examples x the 16 inputs h of tile blockIdx.x; waves as in the
// forward (field parity x example pair).  U_f^T[h, d] = sum_n W_f[h, n] dpre[b, n, d]: A = W16 fragments (through the LDS ring),
// B = dpre[b]^T (k = n, column = d) of the wave's two examples, split, in registers.
//   dXk[b, h, d] += X0[b, f, d] U_f^T[h, d]   summed over the wave's fields in registers, the two parities through LDS
//   dX0[b, f, d]  = sum_h Xk[b, h, d] U_f^T[h, d] over this tile's 16 h: four rows per lane, the four lane quarters by two
//                   butterfly exchanges, collected in LDS, written to dx0_parts[tile] at the end (rsx_cin_dx0_reduce adds the tiles)
// dyn LDS: CsLds.
template <int NS, int KSN, int E>
__global__ __launch_bounds__(64 * E, E == 4 ? 2 : 1) void cin_split_dx_k(const CsDxArgs p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int NP = 32 * KSN + 4, SLOT = StageW<NS, KSN, E>::SLOT, NTHR = 64 * E;
  float* sX0 = lds;                                                   // [E][CS_FP*16]
  bf16_t* sW0 = reinterpret_cast<bf16_t*>(sX0 + E * CS_FP * CS_D);    // ring slot 0
  bf16_t* sW1 = sW0 + SLOT;                                           // ring slot 1
  constexpr bool ALIAS = CsLds<NS, KSN, E>::ALIAS;
  float* sDp = reinterpret_cast<float*>(ALIAS ? sW1 : sW1 + SLOT);    // [E][16][NP]
  float* sR = sDp;                                                    // [E waves][2][4][64] at the end
  float* sP = reinterpret_cast<float*>(reinterpret_cast<char*>(lds) + CsLds<NS, KSN, E>::TOTAL);   // [E][CS_FP][16]: dX0 of this tile
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int i = lane & 15, kq = lane >> 4;
  const int par = wv & 1, e0 = (wv >> 1) * 2;
  const int ht = blockIdx.x, b0 = blockIdx.y * E;
  const bf16_t* wbase = p.W16 + (size_t)ht * KSN * 512;
  const size_t fstride = (size_t)p.H16 * p.Np, plane = (size_t)p.F * fstride;
  const int nstep = (p.F + 1) / 2;
  const float* dsrc = p.dout ? p.dout : p.out;
  const float* gsrc = p.gs ? p.gs : p.out;
  const float* wsrc = p.gs ? p.wout : p.out;
  const float dmul = p.dout ? 1.f : 0.f, gmul = p.gs ? 1.f : 0.f;
  using GW = StageW<NS, KSN, E>;
  GW::issue(wbase, fstride, plane, 0, p.F, tid, sW0);
  if constexpr (!ALIAS) GW::issue(wbase, fstride, plane, 2, p.F, tid, sW1);
  StageX0<E> sx;
  sx.load(p.X0, b0, p.B, p.F, tid);
  float xkv[2][4];                                 // Xk[b0 + e0 + e][h = 16 ht + 4 kq + r][d = i]
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    const int b = b0 + e0 + e;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int h = 16 * ht + 4 * kq + r;
      xkv[e][r] = p.Xk[((size_t)(b < p.B ? b : p.B - 1) * p.H + (h < p.H ? h : p.H - 1)) * CS_D + i] * ((b < p.B && h < p.H) ? 1.f : 0.f);
    }
  }
  {
    // dpre = relu'(out) * (dout + gs * wout) of the E examples -> LDS, transposed (fp32: each wave splits its own two)
    StageRowsF32<KSN, E> sr;
    sr.load(tid, [&](int e, int n, int dq) {
      const int b = b0 + e;
      const bool ok = b < p.B && n < p.N;
      const int bc = b < p.B ? b : p.B - 1, nc = n < p.N ? n : p.N - 1;
      const size_t at = ((size_t)bc * p.N + nc) * 4 + dq;
      const float4 o = reinterpret_cast<const float4*>(p.out)[at];
      // (no branches: a load under a condition is waited for in its own block -- absent operands read `out` and count zero)
      float4 g = f4_scale(dmul, reinterpret_cast<const float4*>(dsrc)[at]);
      const float a = gsrc[bc] * wsrc[nc] * gmul;
      g = make_float4(g.x + a, g.y + a, g.z + a, g.w + a);
      return make_float4((ok && o.x > 0.f) ? g.x : 0.f, (ok && o.y > 0.f) ? g.y : 0.f, (ok && o.z > 0.f) ? g.z : 0.f,
                         (ok && o.w > 0.f) ? g.w : 0.f);
    });
    sx.store(sX0, tid);
    sr.store(sDp, tid);
  }
  __syncthreads();
  bf16x8 bd[2][NS][KSN];                           // dpre[b0 + e0 + e][n = 32 ks + 8 kq + j][d = i], split
#pragma unroll
  for (int e = 0; e < 2; ++e)
#pragma unroll
    for (int ks = 0; ks < KSN; ++ks) {
      const float4* src = reinterpret_cast<const float4*>(sDp + ((size_t)(e0 + e) * 16 + i) * NP + 32 * ks + 8 * kq);
      bf16x8 t[NS];
      split8<NS>(src[0], src[1], t);
#pragma unroll
      for (int s = 0; s < NS; ++s) bd[e][s][ks] = t[s];
    }
  bf16x8 w0[NS][KSN], w1[NS][KSN];
#define CS_READ_WA(ST, W)                                                                                             \
  {                                                                                                                   \
    const bf16_t* wb_ = (((ST) & 1) ? sW1 : sW0) + (par * NS * KSN) * 512 + lane * 8;                                 \
    _Pragma("unroll") for (int s_ = 0; s_ < NS; ++s_)                                                                 \
      _Pragma("unroll") for (int ks_ = 0; ks_ < KSN; ++ks_)                                                           \
        W[s_][ks_] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(wb_ + (s_ * KSN + ks_) * 512));       \
  }
  CS_READ_WA(0, w0)
  {   // the weight-gradient launch's operands and the bias gradient's per-example partials, from the LDS copy (kept in registers
      // across this block the staged values go to scratch memory): every tile's workgroup writes an equal share of the rows n
      // (tile 0 alone finished after the others and set the launch's duration, see cin_split_dx8_k)
    const size_t dplane = (size_t)((p.B + 1) / 2) * 2 * p.N16 * CS_D;
    const int nshare = (32 * KSN + (int)gridDim.x - 1) / (int)gridDim.x;
#pragma unroll
    for (int u = 0; u < 2 * KSN; ++u) {
      const int it = tid + NTHR * u;
      const int dq = it & 3, n = (it >> 2) % (32 * KSN), e = it / (128 * KSN);
      if (n / nshare != ht) continue;              // (whole lane quads: the four quarters of a row go together)
      const int b = b0 + e;
      const float* t = sDp + ((size_t)e * 16 + dq * 4) * NP + n;
      const float4 v = make_float4(t[0], t[NP], t[2 * NP], t[3 * NP]);
      float s = (v.x + v.y) + (v.z + v.w);
      s += __shfl_xor(s, 1);                       // the 4 d-quarters of row n sit in adjacent lanes
      s += __shfl_xor(s, 2);
      if (b < 2 * ((p.B + 1) / 2) && n < p.N16) {            // (the last pair's missing example: zero fragments)
        if (dq == 0 && b < p.B) p.dc_part[(size_t)b * p.N16 + n] = s;
        // fragment of the dW kernel: lane (i = n & 15, kq = 2 (b & 1) + (d >> 3)), elements j = d & 7
        const size_t fr = ((((size_t)(b >> 1) * (p.N16 >> 4) + (n >> 4)) * 64 + (2 * (b & 1) + (dq >> 1)) * 16 + (n & 15)) * 8) + (dq & 1) * 4;
        uint32_t q0[NS], q1[NS];
        split2<NS>(v.x, v.y, q0);
        split2<NS>(v.z, v.w, q1);
#pragma unroll
        for (int sp = 0; sp < NS; ++sp) *reinterpret_cast<uint2*>(p.dpre16 + (size_t)sp * dplane + fr) = make_uint2(q0[sp], q1[sp]);
      }
    }
  }
  __syncthreads();                                 // every wave has its operands (sDp is free) and slot 0's fragments
  if constexpr (ALIAS) {
    GW::issue(wbase, fstride, plane, 2, p.F, tid, sW1);
    __syncthreads();
  }
  f32x4 dxk[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
  // (no branch inside a step: across one the staging registers go to scratch)
#define CS_DX_BODY(ST, W)                                                                                             \
    const int f_ = 2 * (ST) + par;                                                                                    \
    const int fc_ = f_ < CS_FP ? f_ : CS_FP - 1;                                                                      \
    const float m_ = f_ < CS_FP ? 1.f : 0.f;                                                                          \
    _Pragma("unroll") for (int e = 0; e < 2; ++e) {                                                                   \
      const f32x4 U = split_mma_ba<NS, KSN>(bd[e], W, (f32x4){0.f, 0.f, 0.f, 0.f});                                   \
      const float x = sX0[((e0 + e) * CS_FP + fc_) * CS_D + i] * m_;                                                  \
      dxk[e][0] = __builtin_fmaf(x, U[0], dxk[e][0]);                                                                 \
      dxk[e][1] = __builtin_fmaf(x, U[1], dxk[e][1]);                                                                 \
      dxk[e][2] = __builtin_fmaf(x, U[2], dxk[e][2]);                                                                 \
      dxk[e][3] = __builtin_fmaf(x, U[3], dxk[e][3]);                                                                 \
      float q = ((U[0] * xkv[e][0] + U[1] * xkv[e][1]) + U[2] * xkv[e][2]) + U[3] * xkv[e][3];                       \
      q += __shfl_xor(q, 16);                                                                                         \
      q += __shfl_xor(q, 32);                                                                                         \
      sP[((e0 + e) * CS_FP + fc_) * CS_D + i] = q;   /* (the four lane quarters hold the same sum and store it four times; */ \
                                                      /*  LDS, not dx0_parts: a global store inside the step sends the      */ \
                                                      /*  staging registers through scratch memory)                          */ \
    }
#define CS_DX_STEP(ST, W, WN)                                                                                         \
  {                                                                                                                   \
    GW::issue(wbase, fstride, plane, 2 * ((ST) + 2), p.F, tid, ((ST) & 1) ? sW1 : sW0);                               \
    CS_READ_WA((ST) + 1, WN)                                                                                          \
    CS_DX_BODY(ST, W)                                                                                                 \
    __syncthreads();                                                                                                  \
  }
  for (int st = 0; st < nstep; st += 2) {
    CS_DX_STEP(st, w0, w1)
    CS_DX_STEP(st + 1, w1, w0)
  }
#undef CS_DX_BODY
#undef CS_DX_STEP
#undef CS_READ_WA
#pragma unroll
  for (int e = 0; e < 2; ++e)
#pragma unroll
    for (int r = 0; r < 4; ++r) sR[((wv * 2 + e) * 4 + r) * 64 + lane] = dxk[e][r];
  for (int e4 = tid; e4 < E * p.F * 4; e4 += NTHR) {        // this tile's share of dX0 (the last step's barrier published sP)
    const int ex = e4 / (p.F * 4), r = e4 - ex * (p.F * 4);
    if (b0 + ex < p.B)
      reinterpret_cast<float4*>(p.dx0_parts + ((size_t)ht * p.B + b0 + ex) * p.F * CS_D)[r] = reinterpret_cast<const float4*>(sP + ex * CS_FP * CS_D)[r];
  }
  __syncthreads();
  {   // wave w finishes example w: dXk[b][h = 16 ht + 4 kq + r][d = i] = even fields' share + odd fields' share
    const int ex = wv, b = b0 + ex;
    const int w0i = (ex >> 1) * 2, sl = ex & 1;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float s = sR[((w0i * 2 + sl) * 4 + r) * 64 + lane] + sR[(((w0i + 1) * 2 + sl) * 4 + r) * 64 + lane];
      const int h = 16 * ht + 4 * kq + r;
      if (b < p.B && h < p.H) {
        float* dst = p.dXk + ((size_t)b * p.H + h) * CS_D + i;
        *dst = p.acc_dxk ? *dst + s : s;
      }
    }
  }
}

// ------------------------------------------------------------------------------------ backward dXk / dX0, deep ring (default)
// grid = (H16 / 16, ceil(B / 8)), block = 512: workgroup = 8 examples x the 16 inputs h of tile blockIdx.x, waves = (field
// parity, example pair) and the ring / register pipeline of cin_split_fwd8_k.  U_f^T[h, d] = sum_n W_f[h, n] dpre[b, n, d]:
// A = W16 fragments, B = dpre[b]^T of the wave's two examples, formed from out / dout / gs * wout straight from L2 and split.
// dyn LDS = Cs8<NS, KSN, 20480 + 256>::TOTAL: sX0 | ring | sP [8][CS_FP][16] (this tile's dX0) | inverse filter scales.
template <int MODE, int KSN>
__global__ __launch_bounds__(512, 1) void cin_split_dx8_k(const CsDxArgs p) {
  using M = SplitMode<MODE>;
  constexpr int NS = M::NS;
  constexpr int PB = 8 * CS_FP * CS_D * 4;
  using C8 = Cs8<NS, KSN, PB + (M::SCALED ? 256 : 0)>;
  constexpr int E = 8, R = C8::R, PF = C8::PF, FR = C8::FR, SLOTB = C8::SLOTB, NTHR = 512;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* sX0 = lds;
  char* ring = reinterpret_cast<char*>(lds) + C8::X0B;
  float* sP = reinterpret_cast<float*>(ring + (size_t)R * SLOTB);
  float* sInv = sP + E * CS_FP * CS_D;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i = lane & 15, kq = lane >> 4;
  const int par = wv & 1, e0 = (wv >> 1) * 2;
  // acc_dxk == 2 (layer 0, dXk IS dX0 [B, F, 16]): TWO workgroups per (tile, examples) split the fields -- with three tiles of h
  // the launch otherwise fills 96 of 256 CUs; the second half's share of dXk goes to dx0_parts tile HT, which the reduce adds
  const int HT = p.H16 >> 4;
  const int ht = (int)blockIdx.x % HT, half = (int)blockIdx.x / HT, b0 = blockIdx.y * E;
  const uint32_t fstrideB = (uint32_t)p.H16 * p.Np * 2u, planeB = (uint32_t)p.F * fstrideB;
  const char* img = reinterpret_cast<const char*>(p.W16) + (size_t)ht * KSN * 1024;
  const uint32_t ring_lds = (uint32_t)(uintptr_t)ring, lane16 = (uint32_t)lane * 16u;
  const int nstep_all = (p.F + 1) / 2;
  const int nhalf = p.acc_dxk == 2 ? (nstep_all + 1) / 2 : nstep_all;
  const int sbase = half * nhalf;                                   // this workgroup's steps: sbase .. sbase + nstep - 1
  const int nstep = nstep_all - sbase < nhalf ? nstep_all - sbase : nhalf;
  const bool st0 = KSN == 4 && blockIdx.x == 1 && blockIdx.y == 0;
  RSX_STAMP(32, st0); RSX_STAMP_MAX(48, KSN == 4);
  // The dpre transposition scratch (8 waves x 8 KiB) sits at the END of the ring: the first NE slots' fragments are requested
  // now and land while the operands are formed, the others once the scratch is free.
  constexpr int SCR_B = 8 * 128 * CS_D * 4;
  static_assert(R * SLOTB >= SCR_B, "the ring holds the transposition scratch");
  constexpr int NE_ = (R * SLOTB - SCR_B) / SLOTB;
  constexpr int NE = NE_ > R - 1 ? R - 1 : NE_;
#pragma unroll
  for (int s = 0; s < NE; ++s) C8::issue(img, fstrideB, planeB, sbase + (s < nstep ? s : nstep - 1), p.F, wv, lane16, ring_lds + s * SLOTB);
  const float* dsrc = p.dout ? p.dout : p.out;     // (absent operands read `out` and count zero: no branches around loads)
  const float* gsrc = p.gs ? p.gs : p.out;
  const float* wsrc = p.gs ? p.wout : p.out;
  const float dmul = p.dout ? 1.f : 0.f, gmul = p.gs ? 1.f : 0.f;
  StageX0<E> sx;
  sx.load(p.X0, b0, p.B, p.F, tid);
  if (M::SCALED && tid < CS_FP) sInv[tid] = tid < p.F ? p.winv[tid] : 0.f;
  float xkv[2][4];                                 // Xk[b0 + e0 + e][h = 16 ht + 4 kq + r][d = i]
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    const int b = b0 + e0 + e;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int h = 16 * ht + 4 * kq + r;
      xkv[e][r] = p.Xk[((size_t)(b < p.B ? b : p.B - 1) * p.H + (h < p.H ? h : p.H - 1)) * CS_D + i] * ((b < p.B && h < p.H) ? 1.f : 0.f);
    }
  }
  typename M::quad bd[2][NS][KSN];                 // dpre[b0 + e0 + e][n = 32 ks + 8 kq + j][d = i], split
  float inv_d[2] = {1.f, 1.f};
  {
    // dpre = relu'(out) * (dout + gs * wout) of the wave's two examples: whole rows by coalesced float4 loads (32 KSN per lane,
    // all requested together), transposed through 8 KiB of LDS per wave (the ring's last 64 KiB: the filter DMA of those slots
    // starts after this block), lane (i, kq) then picks rows n = 32 ks + 8 kq + j of column d = i.  (Fetching the 64 elements per lane straight
    // from L2 cost 6-10 us: 4-byte accesses to 64-byte row pieces, re-read by the 8 tiles' workgroups.)
    float* scr = reinterpret_cast<float*>(ring + (size_t)R * SLOTB - SCR_B) + (size_t)wv * (128 * CS_D);
    float4 o4[2][2 * KSN], g4[2][2 * KSN];
    float wo[2 * KSN], gbv[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int b = b0 + e0 + e;
      const int bc = b < p.B ? b : p.B - 1;
      gbv[e] = gsrc[bc];
#pragma unroll
      for (int u = 0; u < 2 * KSN; ++u) {
        const int it = lane + 64 * u, n = it >> 2, dq = it & 3;
        const size_t at = ((size_t)bc * p.N + (n < p.N ? n : p.N - 1)) * 4 + dq;
        o4[e][u] = reinterpret_cast<const float4*>(p.out)[at];
        g4[e][u] = reinterpret_cast<const float4*>(dsrc)[at];
      }
    }
#pragma unroll
    for (int u = 0; u < 2 * KSN; ++u) {
      const int n = (lane + 64 * u) >> 2;
      wo[u] = wsrc[n < p.N ? n : p.N - 1] * gmul;
    }
    __builtin_amdgcn_sched_barrier(0);             // (every load requested before the first use)
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int b = b0 + e0 + e;
#pragma unroll
      for (int u = 0; u < 2 * KSN; ++u) {
        const int it = lane + 64 * u, n = it >> 2;
        const float m = (b < p.B && n < p.N) ? 1.f : 0.f;
        const float a = gbv[e] * wo[u];
        const float4 o = o4[e][u], g = g4[e][u];
        reinterpret_cast<float4*>(scr)[it] = make_float4((o.x > 0.f ? g.x * dmul + a : 0.f) * m, (o.y > 0.f ? g.y * dmul + a : 0.f) * m,
                                                         (o.z > 0.f ? g.z * dmul + a : 0.f) * m, (o.w > 0.f ? g.w * dmul + a : 0.f) * m);
      }
      float v[KSN][8];
      float mx = 0.f;
#pragma unroll
      for (int ks = 0; ks < KSN; ++ks)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          v[ks][j] = scr[(32 * ks + 8 * kq + j) * CS_D + i];
          if (M::SCALED) mx = fmaxf(mx, fabsf(v[ks][j]));
        }
      float sc = 1.f;
      if (M::SCALED) cs_pow2_scale(cs_wave_max(mx), sc, inv_d[e]);
#pragma unroll
      for (int ks = 0; ks < KSN; ++ks) {
        typename M::quad t[NS];
        M::split(make_float4(v[ks][0] * sc, v[ks][1] * sc, v[ks][2] * sc, v[ks][3] * sc),
                 make_float4(v[ks][4] * sc, v[ks][5] * sc, v[ks][6] * sc, v[ks][7] * sc), t);
#pragma unroll
        for (int s = 0; s < NS; ++s) bd[e][s][ks] = t[s];
      }
    }
  }
  {   // the weight-gradient launch's operands (THREE bf16 planes in every mode) and the bias gradient's per-example partials of
      // this example group: every workgroup of the group (all tiles of h, both field halves) takes an equal share of the rows n
      // -- done by tile 0 alone, those workgroups finished ~3 us after the others and set the launch's duration.
      // item = (example, row n of the share, quarter of d); rows past N16 are not stored
    const size_t dplane = (size_t)((p.B + 1) / 2) * 2 * p.N16 * CS_D;
    const int nparts = (int)gridDim.x, part = (int)blockIdx.x;
    const int nshare = (p.N16 + nparts - 1) / nparts;
    const int items = E * nshare * 4;
    for (int it = tid; it < items; it += NTHR) {
      const int dq = it & 3, n = part * nshare + (it >> 2) % nshare, e = it / (4 * nshare);
      if (n >= p.N16) continue;                    // (whole lane quads: the four quarters of a row are skipped together)
      const int b = b0 + e;
      const int bc = b < p.B ? b : p.B - 1, nc = n < p.N ? n : p.N - 1;
      const size_t at = ((size_t)bc * p.N + nc) * 4 + dq;
      const float4 o = reinterpret_cast<const float4*>(p.out)[at];
      float4 g = f4_scale(dmul, reinterpret_cast<const float4*>(dsrc)[at]);
      const float a = gsrc[bc] * wsrc[nc] * gmul;
      const bool ok = b < p.B && n < p.N;
      const float4 v = make_float4((ok && o.x > 0.f) ? g.x + a : 0.f, (ok && o.y > 0.f) ? g.y + a : 0.f,
                                   (ok && o.z > 0.f) ? g.z + a : 0.f, (ok && o.w > 0.f) ? g.w + a : 0.f);
      float s = (v.x + v.y) + (v.z + v.w);
      s += __shfl_xor(s, 1);                       // the 4 d-quarters of row n sit in adjacent lanes
      s += __shfl_xor(s, 2);
      if (b < 2 * ((p.B + 1) / 2)) {               // (the last pair's missing example: zero fragments)
        if (dq == 0 && b < p.B) p.dc_part[(size_t)b * p.N16 + n] = s;
        // fragment of the dW kernel: lane (i = n & 15, kq = 2 (b & 1) + (d >> 3)), elements j = d & 7
        const size_t fr = ((((size_t)(b >> 1) * (p.N16 >> 4) + (n >> 4)) * 64 + (2 * (b & 1) + (dq >> 1)) * 16 + (n & 15)) * 8) + (dq & 1) * 4;
        uint32_t q0[3], q1[3];
        split2<3>(v.x, v.y, q0);
        split2<3>(v.z, v.w, q1);
#pragma unroll
        for (int sp = 0; sp < 3; ++sp)
          if (sp < p.dw_planes) *reinterpret_cast<uint2*>(p.dpre16 + (size_t)sp * dplane + fr) = make_uint2(q0[sp], q1[sp]);
      }
    }
  }
  RSX_STAMP(33, st0);
  sx.store(sX0, tid);
  __syncthreads();                                 // every wave is done with its transposition scratch: the ring is free
#pragma unroll
  for (int s = NE; s < R - 1; ++s) C8::issue(img, fstrideB, planeB, sbase + (s < nstep ? s : nstep - 1), p.F, wv, lane16, ring_lds + s * SLOTB);
  // sX0 and the slots of steps 0 and 1 (the pieces requested just now may still be in flight when they are not those)
  cs_wait_barrier<(NE >= 2 ? (R - 1 - NE) * C8::U : 0)>();
  RSX_STAMP(34, st0);
  const char* rd = ring + (size_t)par * FR * 1024 + lane16;
  typename M::quad w[PF];
#pragma unroll
  for (int j = 0; j < PF; ++j)
    w[j] = __builtin_bit_cast(typename M::quad, *reinterpret_cast<const uint4*>(rd + ((j % NS) * KSN + j / NS) * 1024));
  f32x4 dxk[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
  int sl = 0;
  for (int st = 0; st < nstep; ++st) {
    const int sln = sl + 1 == R ? 0 : sl + 1, slp = sl == 0 ? R - 1 : sl - 1;
    const int stn = st + R - 1 < nstep ? st + R - 1 : nstep - 1;
    CS_DBG_LOAD(C8::issue(img, fstrideB, planeB, sbase + stn, p.F, wv, lane16, ring_lds + slp * SLOTB);)
    const int f_ = 2 * (sbase + st) + par;
    float x[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) x[e] = sX0[((e0 + e) * CS_FP + f_) * CS_D + i];
    const float wi = M::SCALED ? sInv[f_] : 1.f;
    f32x4 T[2][NS];
    cs8_step_mma<MODE, KSN, PF, true>(bd, w, rd + (size_t)sl * SLOTB, rd + (size_t)sln * SLOTB, T);
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      f32x4 U = T[e][NS - 1];
#pragma unroll
      for (int l = NS - 2; l >= 0; --l) U += T[e][l];
      if (M::SCALED) U *= wi * inv_d[e];
      dxk[e][0] = __builtin_fmaf(x[e], U[0], dxk[e][0]);
      dxk[e][1] = __builtin_fmaf(x[e], U[1], dxk[e][1]);
      dxk[e][2] = __builtin_fmaf(x[e], U[2], dxk[e][2]);
      dxk[e][3] = __builtin_fmaf(x[e], U[3], dxk[e][3]);
      float q = ((U[0] * xkv[e][0] + U[1] * xkv[e][1]) + U[2] * xkv[e][2]) + U[3] * xkv[e][3];
      q += __shfl_xor(q, 16);
      q += __shfl_xor(q, 32);
      sP[((e0 + e) * CS_FP + f_) * CS_D + i] = q;  // (the four lane quarters hold the same sum and store it four times)
    }
    CS8_STEP_BARRIER(C8);
    sl = sln;
  }
  RSX_STAMP(35, st0);
  cs_wait_barrier<0>();                            // the tail's redundant pieces have landed (the ring is free); sP is complete
  RSX_STAMP(36, st0);
  float* sR = reinterpret_cast<float*>(ring);      // [8 waves][2][4][64]
#pragma unroll
  for (int e = 0; e < 2; ++e)
#pragma unroll
    for (int r = 0; r < 4; ++r) sR[((wv * 2 + e) * 4 + r) * 64 + lane] = dxk[e][r];
  {   // this tile's share of dX0: the fields of this workgroup's steps
    const int fa = 2 * sbase, fb = 2 * (sbase + nstep) < p.F ? 2 * (sbase + nstep) : p.F;
    const int nf4 = (fb - fa) * 4;
    for (int e4 = tid; e4 < E * nf4; e4 += NTHR) {
      const int ex = e4 / nf4, r = fa * 4 + (e4 - ex * nf4);
      if (b0 + ex < p.B)
        reinterpret_cast<float4*>(p.dx0_parts + ((size_t)ht * p.B + b0 + ex) * p.F * CS_D)[r] = reinterpret_cast<const float4*>(sP + ex * CS_FP * CS_D)[r];
    }
  }
  __syncthreads();
  {   // wave w finishes example w: dXk[b][h = 16 ht + 4 kq + r][d = i] = even fields' share + odd fields' share
    const int ex = wv, b = b0 + ex;
    const int w0i = (ex >> 1) * 2, sl2 = ex & 1;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float s = sR[((w0i * 2 + sl2) * 4 + r) * 64 + lane] + sR[(((w0i + 1) * 2 + sl2) * 4 + r) * 64 + lane];
      const int h = 16 * ht + 4 * kq + r;
      if (b < p.B && h < p.H) {
        if (half == 0) {
          float* dst = p.dXk + ((size_t)b * p.H + h) * CS_D + i;
          *dst = p.acc_dxk == 1 ? *dst + s : s;
        } else {                                   // (H == F: a row of dXk is a row of dX0)
          p.dx0_parts[((size_t)HT * p.B + b) * p.F * CS_D + (size_t)h * CS_D + i] = s;
        }
      }
    }
  }
  RSX_STAMP(37, st0); RSX_STAMP_MAX(49, KSN == 4);
}

template <typename K>
int opt_in_lds(K kernel, size_t lds) {
  if (lds > 160 * 1024) return RSX_EUNSUPPORTED;
  if (lds > 64 * 1024 &&
      hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
    return RSX_ELAUNCH;
  return RSX_OK;
}

// examples per workgroup of the first form: 4 (two independent 256-thread workgroups per CU; 8 measured slower, round 5)

template <int NS, int KS, int E>
int launch_fwd(const CsFwdArgs& a, hipStream_t stream) {
  const dim3 grid((unsigned)(a.N16 / 16), (unsigned)((a.B + E - 1) / E));
  const size_t lds = CsLds<NS, KS, E>::TOTAL;
  const int rc = opt_in_lds(cin_split_fwd_k<NS, KS, E>, lds);
  if (rc != RSX_OK) return rc;
  RSX_LAUNCH((cin_split_fwd_k<NS, KS, E>), grid, dim3(64 * E), lds, stream, a);
  RSX_CHECK_LAUNCH();
  return RSX_OK;
}
template <int NS, int E>
int launch_fwd_ns(const CsFwdArgs& a, hipStream_t stream) {
  switch (a.Hp / 32) {
    case 1: return launch_fwd<NS, 1, E>(a, stream);
    case 2: return launch_fwd<NS, 2, E>(a, stream);
    case 3: return launch_fwd<NS, 3, E>(a, stream);
    default: return launch_fwd<NS, 4, E>(a, stream);
  }
}

// Which kernels run a mode: 1 = the first form (4 examples per 256-thread workgroup, two workgroups per CU), 2 = the deep-ring
// form (8 examples per 512-thread workgroup).  Measured inside xdeepfm.py's step (profiles/r05_y_*): for the bf16 modes 1..3 the
// first form is the faster one (0.2550 against 0.2638 ms per step at ns = 3: two independent workgroups per CU cover each
// other's barriers and prologues), mode 4 exists in the deep-ring form only.  RSX_CIN_SPLIT_V=1|2 forces one (A/B runs).
// Round 6 (pruned): the losing forms are no longer instantiated -- ns = 3 runs the first form with 4 examples per workgroup, mode
// 4 the deep-ring form; ns = 1 / 2 (A/B arithmetic of round 5: plain bf16 / 2^-16-grade) are gone (RSX_EUNSUPPORTED; the plain
// bf16-operand path is csrc/cin_bf16.hip).
int cs_version(int ns) { return ns == CS_H2 ? 2 : 1; }
template <int MODE, int KS>
int launch_fwd8(const CsFwdArgs& a, hipStream_t stream) {
  using M = SplitMode<MODE>;
  const dim3 grid((unsigned)(a.N16 / 16), (unsigned)((a.B + 7) / 8));
  const size_t lds = Cs8<M::NS, KS, M::SCALED ? 256 : 0>::TOTAL;
  const int rc = opt_in_lds(cin_split_fwd8_k<MODE, KS>, lds);
  if (rc != RSX_OK) return rc;
  RSX_LAUNCH((cin_split_fwd8_k<MODE, KS>), grid, dim3(512), lds, stream, a);
  RSX_CHECK_LAUNCH();
  return RSX_OK;
}
template <int MODE>
int launch_fwd8_ns(const CsFwdArgs& a, hipStream_t stream) {
  switch (a.Hp / 32) {
    case 1: return launch_fwd8<MODE, 1>(a, stream);
    case 2: return launch_fwd8<MODE, 2>(a, stream);
    case 3: return launch_fwd8<MODE, 3>(a, stream);
    default: return launch_fwd8<MODE, 4>(a, stream);
  }
}
template <int MODE, int KSN>
int launch_dx8(const CsDxArgs& a, hipStream_t stream) {
  using M = SplitMode<MODE>;
  const dim3 grid((unsigned)(a.H16 / 16) * (a.acc_dxk == 2 ? 2u : 1u), (unsigned)((a.B + 7) / 8));
  const size_t lds = Cs8<M::NS, KSN, 8 * CS_FP * CS_D * 4 + (M::SCALED ? 256 : 0)>::TOTAL;
  const int rc = opt_in_lds(cin_split_dx8_k<MODE, KSN>, lds);
  if (rc != RSX_OK) return rc;
  RSX_LAUNCH((cin_split_dx8_k<MODE, KSN>), grid, dim3(512), lds, stream, a);
  RSX_CHECK_LAUNCH();
  return RSX_OK;
}
template <int MODE>
int launch_dx8_ns(const CsDxArgs& a, hipStream_t stream) {
  switch (a.Np / 32) {
    case 1: return launch_dx8<MODE, 1>(a, stream);
    case 2: return launch_dx8<MODE, 2>(a, stream);
    case 3: return launch_dx8<MODE, 3>(a, stream);
    default: return launch_dx8<MODE, 4>(a, stream);
  }
}
inline int cs_planes(int ns) { return ns == CS_H2 ? 2 : ns; }          // 16-bit planes of the filter images
inline int cs_dw_planes(int ns) { return ns == CS_H2 ? 3 : ns; }       // bf16 planes the weight-gradient launch multiplies

template <int NS, int KSN, int E>
int launch_dx(const CsDxArgs& a, hipStream_t stream) {
  const dim3 grid((unsigned)(a.H16 / 16), (unsigned)((a.B + E - 1) / E));
  const size_t lds = CsLds<NS, KSN, E>::TOTAL + (size_t)E * CS_FP * CS_D * 4;
  const int rc = opt_in_lds(cin_split_dx_k<NS, KSN, E>, lds);
  if (rc != RSX_OK) return rc;
  RSX_LAUNCH((cin_split_dx_k<NS, KSN, E>), grid, dim3(64 * E), lds, stream, a);
  RSX_CHECK_LAUNCH();
  return RSX_OK;
}
template <int NS, int E>
int launch_dx_ns(const CsDxArgs& a, hipStream_t stream) {
  switch (a.Np / 32) {
    case 1: return launch_dx<NS, 1, E>(a, stream);
    case 2: return launch_dx<NS, 2, E>(a, stream);
    case 3: return launch_dx<NS, 3, E>(a, stream);
    default: return launch_dx<NS, 4, E>(a, stream);
  }
}

size_t image_elems(int F, int H, int N) {        // one plane of both layouts
  return (size_t)F * rup(H, 16) * rup(N, 32) + (size_t)F * rup(N, 16) * rup(H, 32);
}

}  // namespace

// ------------------------------------------------------------------------------------------------------ entry points
extern "C" size_t rsx_cin_split_weight_elems(int F, int H, int N, int ns) {
  if (F <= 0 || H <= 0 || N <= 0 || ns < 1 || ns > CS_H2) return 0;
  return (size_t)cs_planes(ns) * image_elems(F, H, N) + (ns == CS_H2 ? 2 * CS_FP : 0);     // (+ the fields' inverse scales, fp32)
}

static int cs_launch_prep(const float* const* W_h, void* const* w16_h, const int32_t* H_h, const int32_t* N_h, int L, int F, int ns,
                          const rsx_gather_two_job* g, rsx_stream_t stream);
extern "C" int rsx_cin_split_prep(const float* const* W_h, void* const* w16_h, const int32_t* H_h, const int32_t* N_h, int L,
                                  int F, int ns, rsx_stream_t stream) {
  return cs_launch_prep(W_h, w16_h, H_h, N_h, L, F, ns, nullptr, stream);
}
extern "C" int rsx_cin_split_prep_gather(const float* const* W_h, void* const* w16_h, const int32_t* H_h, const int32_t* N_h, int L,
                                         int F, int ns, const rsx_gather_two_job* g, rsx_stream_t stream) {
  if (!g) return RSX_EINVAL;
  if (g->B < 0 || g->F <= 0 || g->F > 64 || g->ND <= 0 || g->ND > 64) return RSX_EINVAL;
  if (g->D != 16) return RSX_EUNSUPPORTED;
  if (g->B > 0 && (!g->tables1 || !g->w1 || !g->tables2 || !g->row_off || !g->ids || !g->num_x || !g->num_w || !g->E1 || !g->E2 || !g->y1))
    return RSX_EINVAL;
  return cs_launch_prep(W_h, w16_h, H_h, N_h, L, F, ns, g, stream);
}
static int cs_launch_prep(const float* const* W_h, void* const* w16_h, const int32_t* H_h, const int32_t* N_h, int L, int F, int ns,
                          const rsx_gather_two_job* g, rsx_stream_t stream) {
  if (!W_h || !w16_h || !H_h || !N_h || L <= 0 || F <= 0 || ns < 1 || ns > CS_H2) return RSX_EINVAL;
  if (ns < 3) return RSX_EUNSUPPORTED;       // (round 6: only ns = 3 and mode 4 are built)
  if (L > CS_MAXJ || (ns == CS_H2 && F > CS_FP)) return RSX_EUNSUPPORTED;
  const int np = cs_planes(ns);
  CsPrepArgs a{};
  a.njobs = L;
  a.F = F;
  long long tot = 0;
  for (int k = 0; k < L; ++k) {
    const int H = H_h[k], N = N_h[k];
    if (!W_h[k] || !w16_h[k] || H <= 0 || N <= 0) return RSX_EINVAL;
    if (H > 128 || N > 128) return RSX_EUNSUPPORTED;
    CsPrepJob& j = a.job[k];
    j.W = W_h[k]; j.H = H; j.N = N; j.H16 = rup(H, 16); j.N16 = rup(N, 16); j.Hp = rup(H, 32); j.Np = rup(N, 32);
    j.W16 = static_cast<bf16_t*>(w16_h[k]);
    j.Wt16 = j.W16 + (size_t)np * F * j.H16 * j.Np;
    j.winv = reinterpret_cast<float*>(j.W16 + (size_t)np * image_elems(F, H, N));
    tot += ((long long)F * j.H16 * j.Np + (long long)F * j.N16 * j.Hp) >> 3;
    j.end = tot;
  }
  if (g && g->B > 0) {
    a.n_gather = (g->B + 3) / 4;
    a.gt = GatherTwoArgs{g->tables1, g->w1, g->tables2, g->row_off, g->ids, g->num_x, g->num_w, g->E1, g->E2, g->y1, g->w1_field_mask,
                         g->B, g->F, g->ND};
  }
  if (ns == CS_H2) {                                // CS_H2_PARTS workgroups per (layer, field): the field's largest |W| first
    RSX_LAUNCH(cin_split_prep_h2_k, dim3((unsigned)(a.n_gather + L * F * CS_H2_PARTS)), dim3(256), 0, rsx_s(stream), a);
    RSX_CHECK_LAUNCH();
    return RSX_OK;
  }
  const unsigned blocks = (unsigned)a.n_gather + (unsigned)((tot + 255) / 256 < 4096 ? (tot + 255) / 256 : 4096);
  RSX_LAUNCH(cin_split_prep_k<3>, dim3(blocks), dim3(256), 0, rsx_s(stream), a);
  RSX_CHECK_LAUNCH();
  return RSX_OK;
}

extern "C" int rsx_cin_split_fwd(const float* X0, const float* Xk, const void* w16, const float* c, float* out, int B, int F,
                                 int H, int N, int D, int ns, rsx_stream_t stream) {
  if (B < 0 || F <= 0 || H <= 0 || N <= 0 || ns < 1 || ns > CS_H2) return RSX_EINVAL;
  if (ns < 3) return RSX_EUNSUPPORTED;       // (round 6: only ns = 3 and mode 4 are built)
  if (B == 0) return RSX_OK;
  if (!X0 || !Xk || !w16 || !c || !out) return RSX_EINVAL;
  if (D != CS_D || H > 128 || N > 128 || F > CS_FP) return RSX_EUNSUPPORTED;
  const int H16 = rup(H, 16), N16 = rup(N, 16), Hp = rup(H, 32), Np = rup(N, 32);
  const int np = cs_planes(ns);
  const bf16_t* wt = static_cast<const bf16_t*>(w16) + (size_t)np * F * H16 * Np;
  const float* winv = reinterpret_cast<const float*>(static_cast<const bf16_t*>(w16) + (size_t)np * image_elems(F, H, N));
  const CsFwdArgs a{X0, Xk, wt, c, out, B, F, H, N, N16, Hp, winv};
  if (ns == CS_H2) return launch_fwd8_ns<CS_H2>(a, rsx_s(stream));
  return launch_fwd_ns<3, 4>(a, rsx_s(stream));
}

// ws: [ns planes of dpre fragments | B x N16 bias-gradient partials]
extern "C" size_t rsx_cin_split_bwd_workspace_bytes(int B, int N, int ns) {
  if (B <= 0 || N <= 0 || ns < 1 || ns > CS_H2) return 0;
  return (size_t)cs_dw_planes(ns) * ((B + 1) / 2) * 2 * rup(N, 16) * CS_D * 2 + (size_t)B * rup(N, 16) * sizeof(float);
}

extern "C" int rsx_cin_split_bwd_dx(const float* X0, const float* Xk, const void* w16, const float* out, const float* dout,
                                    const float* gs, const float* wout, float* dXk, int acc_dxk, float* dx0_parts, void* ws,
                                    int B, int F, int H, int N, int D, int ns, rsx_stream_t stream) {
  if (B < 0 || F <= 0 || H <= 0 || N <= 0 || ns < 1 || ns > CS_H2) return RSX_EINVAL;
  if (ns < 3) return RSX_EUNSUPPORTED;       // (round 6: only ns = 3 and mode 4 are built)
  if (B == 0) return RSX_OK;
  if (!X0 || !Xk || !w16 || !out || !dXk || !dx0_parts || !ws) return RSX_EINVAL;
  if ((!dout && !gs) || (gs && !wout)) return RSX_EINVAL;
  if (D != CS_D || H > 128 || N > 128 || F > CS_FP) return RSX_EUNSUPPORTED;
  if (acc_dxk < 0 || acc_dxk > 2) return RSX_EINVAL;
  // acc_dxk = 2: the fields split over two workgroups per tile, the second half's dXk in dx0_parts tile ceil(H/16) (H == F, deep-ring kernels)
  if (acc_dxk == 2 && (H != F || cs_version(ns) != 2)) return RSX_EUNSUPPORTED;
  const int H16 = rup(H, 16), N16 = rup(N, 16), Np = rup(N, 32);
  const int dwp = cs_dw_planes(ns);
  float* dc_part = reinterpret_cast<float*>(static_cast<char*>(ws) + (size_t)dwp * ((B + 1) / 2) * 2 * N16 * CS_D * 2);
  const float* winv = reinterpret_cast<const float*>(static_cast<const bf16_t*>(w16) + (size_t)cs_planes(ns) * image_elems(F, H, N));
  const CsDxArgs a{X0, Xk, static_cast<const bf16_t*>(w16), out, dout, gs, wout, dXk, dx0_parts, static_cast<bf16_t*>(ws),
                   dc_part, acc_dxk, B, F, H, N, H16, N16, Np, winv, dwp};
  if (ns == CS_H2) return launch_dx8_ns<CS_H2>(a, rsx_s(stream));
  return launch_dx_ns<3, 4>(a, rsx_s(stream));
}

// ------------------------------------------------------------------------------------------------ backward: dW, dc
// dW_f[h, n] = sum over (b, d) of Z[b, f, h, d] * dpre[b, n, d], Z = X0 * Xk rounded once to fp32 (as cin.hip forms it) and
// split into NS planes on the fly; dpre's planes come from the data-gradient launch (fragments of example pairs).
// One launch for all layers.  Workgroup = 8 waves that split the k-steps (example pairs) of ONE tile = FT fields x 16 h x 64
// n, partial tiles added in fixed order through LDS; FT is a per-job choice (so that the tiles of all layers together fill
// the CUs in one round: 4 x 40 + ... see cs_launch_dw).  The X0 slab of the tile's fields sits in LDS (cin_bf16.hip: X0L).
namespace {

constexpr int CS_NT = 4;
struct CsDwJob {
  const float* Xk;        // [B, H, 16]
  const bf16_t* dpre16;   // NS planes of fragments, see CsDxArgs
  const float* dc_part;   // [B][N16]
  float* dW;              // [F*H, N]
  float* dc;              // [N]
  int H, N, N16;
  int ft;                 // fields per tile (3 or 4)
  int gx, HT;             // tile grid of the job: gx n-groups x HT h tiles x ceil(F / ft) field groups
  int tile_end;           // exclusive prefix sum of the jobs' tile counts
};
struct CsDwArgs {
  CsDwJob job[CS_MAXJ];
  int njobs;
  const float* X0;        // [B, F, 16]
  int B, F;
  int x0l;                // the tiles' X0 slabs sit in LDS (B <= 640)
  // riders (rsx_cin_split_bwd_dw_dx0): red_blocks extra workgroups add the data-gradient launches' dX0 tile partials -- what
  // cin_dx0_reduce_k does in a launch of its own (5.4 us of launch latency for 7 MB; the tiles leave ~18 CUs free)
  const float* red_parts[CS_MAXJ];
  int red_tiles[CS_MAXJ];
  int red_njobs, red_blocks, red_acc;
  float* red_dX0;
  unsigned long long red_n4;
};

// (A tile of 6 fields x 3 n tiles -- 42 % fewer dpre bytes per MFMA -- measured SLOWER: xdeepfm.py 0.265 against 0.238 ms; the
// launch is bound by forming and splitting the Z operand on the VALU, which grows with the fields per tile.)
template <int NS, int FT, bool X0L>
__device__ __forceinline__ void cs_dw_tile(const CsDwArgs& p, const CsDwJob& jb, int local, float4* dw_lds) {
  constexpr int NT = CS_NT;
  float (*red)[FT * NT][256] = reinterpret_cast<float (*)[FT * NT][256]>(dw_lds);
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int bx = local % jb.gx, by = (local / jb.gx) % jb.HT, bz = local / (jb.gx * jb.HT);
  const int i = lane & 15, kq = lane >> 4;
  const int ntg = bx * NT, ht = by, f0 = bz * FT;
  const int NT16 = jb.N16 >> 4;
  const int H = jb.H;
  const int h = 16 * ht + i;
  const int hc = h < H ? h : H - 1;
  const int d0 = (kq & 1) * 8, eb = kq >> 1;      // k = 8 kq + j  <->  example 2 ks + (kq >> 1), dims d0 .. d0 + 7
  const int nks = (p.B + 1) / 2;
  const size_t dplane = (size_t)nks * 2 * jb.N16 * CS_D;
  f32x4 acc[FT][NT];
#pragma unroll
  for (int ft = 0; ft < FT; ++ft)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[ft][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  if constexpr (X0L) {                            // X0[b, f0 + ft, :] -> x0s[(ft * B + b) * 4 + quarter]
    const int n4 = FT * p.B * 4;
    for (int e = tid; e < n4; e += 512) {
      const int qd = e & 3, b = (e >> 2) % p.B, ft = (e >> 2) / p.B;
      const int f = f0 + ft < p.F ? f0 + ft : p.F - 1;
      dw_lds[e] = reinterpret_cast<const float4*>(p.X0 + ((size_t)b * p.F + f) * CS_D)[qd];
    }
    __syncthreads();
  }
  struct Ld {
    float4 xk[2];
    uint4 dp[NS][NT];
    float m;
    int bc;
  };
  auto load = [&](int ks, Ld& L) {
    const int b = 2 * ks + eb;
    L.bc = b < p.B ? b : p.B - 1;
    const int ksc = ks < nks ? ks : nks - 1;
    L.m = (b < p.B && ks < nks && h < H) ? 1.f : 0.f;
    const float4* xk = reinterpret_cast<const float4*>(jb.Xk + ((size_t)L.bc * H + hc) * CS_D + d0);
    L.xk[0] = xk[0];
    L.xk[1] = xk[1];
#pragma unroll
    for (int s = 0; s < NS; ++s)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {             // n tiles past N16 re-read the last tile (never stored)
        const int t = ntg + nt < NT16 ? ntg + nt : NT16 - 1;
        L.dp[s][nt] = *reinterpret_cast<const uint4*>(jb.dpre16 + (size_t)s * dplane + (((size_t)ksc * NT16 + t) * 64 + lane) * 8);
      }
  };
  // One k-step of a stage, and the stage's NEXT k-step (ksn: two k-steps of this wave ahead) requested piece by piece as its
  // registers fall free: Xk as soon as the Z factors are formed, the dpre fragments of n tile nt after the last field's MFMAs on
  // them.  (With whole stages reloaded between the runs a request was one k-step ahead of its use -- ~2 500 cycles with the
  // SIMD's other wave interleaved, under a loaded L2's round trip: the launch waited for its operands, 45 % MFMA-busy.)
  auto run = [&](Ld& L, const int ksn) {
    const float4 k0 = f4_scale(L.m, L.xk[0]), k1 = f4_scale(L.m, L.xk[1]);
    const int bn = 2 * ksn + eb;
    const int bcn = bn < p.B ? bn : p.B - 1;
    const int kscn = ksn < nks ? ksn : nks - 1;
    {
      const float4* xk = reinterpret_cast<const float4*>(jb.Xk + ((size_t)bcn * H + hc) * CS_D + d0);
      L.xk[0] = xk[0];
      L.xk[1] = xk[1];
    }
#pragma unroll
    for (int ft = 0; ft < FT; ++ft) {
      const int fx = f0 + ft < p.F ? f0 + ft : p.F - 1;
      const float4* x0 = X0L ? dw_lds + ((size_t)ft * p.B + L.bc) * 4 + (d0 >> 2)       // (batches whose slab does not fit the LDS
                             : reinterpret_cast<const float4*>(p.X0 + ((size_t)L.bc * p.F + fx) * CS_D + d0);   //  read X0 from L2)
      bf16x8 a[NS][1];
      {
        bf16x8 t[NS];
        split8<NS>(f4_mul(x0[0], k0), f4_mul(x0[1], k1), t);
#pragma unroll
        for (int s = 0; s < NS; ++s) a[s][0] = t[s];
      }
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        bf16x8 b[NS][1];
#pragma unroll
        for (int s = 0; s < NS; ++s) b[s][0] = __builtin_bit_cast(bf16x8, L.dp[s][nt]);
        acc[ft][nt] = split_mma<NS, 1>(a, b, acc[ft][nt]);
        if (ft == FT - 1) {
          const int t = ntg + nt < NT16 ? ntg + nt : NT16 - 1;
#pragma unroll
          for (int s = 0; s < NS; ++s)
            L.dp[s][nt] = *reinterpret_cast<const uint4*>(jb.dpre16 + (size_t)s * dplane + (((size_t)kscn * NT16 + t) * 64 + lane) * 8);
        }
      }
    }
    L.bc = bcn;
    L.m = (bn < p.B && ksn < nks && h < H) ? 1.f : 0.f;
  };
  Ld La, Lb;
  load(wv, La);
  load(wv + 8, Lb);                               // (clamped + masked past the batch)
  for (int ks = wv; ks < nks; ks += 16) {         // this wave's k-steps: wv, wv + 8, ...
    run(La, ks + 16);
    run(Lb, ks + 24);
  }
  __syncthreads();                                // (the reduce buffer aliases the X0 slab)
  // partial tiles, fixed order: ((w0 + w4) + (w1 + w5) ... ) as two rounds through one 4-slot LDS buffer
  if (wv >= 4) {
#pragma unroll
    for (int ft = 0; ft < FT; ++ft)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) red[wv - 4][ft * NT + nt][r * 64 + lane] = acc[ft][nt][r];
  }
  __syncthreads();
  if (wv < 4) {
#pragma unroll
    for (int ft = 0; ft < FT; ++ft)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[ft][nt][r] += red[wv][ft * NT + nt][r * 64 + lane];
  }
  __syncthreads();
  if (wv >= 1 && wv < 4) {
#pragma unroll
    for (int ft = 0; ft < FT; ++ft)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) red[wv][ft * NT + nt][r * 64 + lane] = acc[ft][nt][r];
  }
  __syncthreads();
  if (wv == 0) {
#pragma unroll
    for (int ft = 0; ft < FT; ++ft) {
      const int f = f0 + ft;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const int n = 16 * (ntg + nt) + i;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int hh = 16 * ht + 4 * kq + r;
          const float s = ((acc[ft][nt][r] + red[1][ft * NT + nt][r * 64 + lane]) + red[2][ft * NT + nt][r * 64 + lane]) +
                          red[3][ft * NT + nt][r * 64 + lane];
          if (f < p.F && hh < H && n < jb.N) jb.dW[((size_t)f * H + hh) * jb.N + n] = s;
        }
      }
    }
  }
}

// grid = njobs (the bias gradients: per-example partials added in order) + the jobs' tiles, block = 512
template <int NS>
__global__ __launch_bounds__(512) void cin_split_dw_k(const CsDwArgs p) {
  extern __shared__ float4 dw_lds[];
  const int tid = threadIdx.x;
  if ((int)blockIdx.x < p.njobs) {
    const CsDwJob& jb = p.job[blockIdx.x];
    for (int n = tid; n < jb.N; n += 512) {
      float s = 0.f;
      int g = 0;
      for (; g + 8 <= p.B; g += 8) {
        float t[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) t[u] = jb.dc_part[(size_t)(g + u) * jb.N16 + n];
#pragma unroll
        for (int u = 0; u < 8; ++u) s += t[u];
      }
      for (; g < p.B; ++g) s += jb.dc_part[(size_t)g * jb.N16 + n];
      jb.dc[n] = s;
    }
    return;
  }
  const int tile = (int)blockIdx.x - p.njobs;
  if (tile >= p.job[p.njobs - 1].tile_end) {      // rider: dX0[e] = (acc ? dX0[e] : 0) + the tiles' partials in (job, tile) order
    const unsigned long long first = (unsigned long long)(tile - p.job[p.njobs - 1].tile_end) * 512 + tid;
    for (unsigned long long e = first; e < p.red_n4; e += (unsigned long long)p.red_blocks * 512) {
      float4 s = p.red_acc ? reinterpret_cast<const float4*>(p.red_dX0)[e] : F4Z;
#pragma unroll
      for (int j = 0; j < CS_MAXJ; ++j)
        if (j < p.red_njobs)
          for (int t = 0; t < p.red_tiles[j]; ++t) s = f4_add(s, reinterpret_cast<const float4*>(p.red_parts[j])[(size_t)t * p.red_n4 + e]);
      reinterpret_cast<float4*>(p.red_dX0)[e] = s;
    }
    return;
  }
  int ji = 0;
#pragma unroll
  for (int k = 1; k < CS_MAXJ; ++k)
    if (k < p.njobs && tile >= p.job[k - 1].tile_end) ji = k;
  const CsDwJob& jb = p.job[ji];
  const int local = tile - (ji ? p.job[ji - 1].tile_end : 0);
  if (p.x0l) {
    if (jb.ft == 4) cs_dw_tile<NS, 4, true>(p, jb, local, dw_lds);
    else cs_dw_tile<NS, 3, true>(p, jb, local, dw_lds);
  } else {
    cs_dw_tile<NS, 3, false>(p, jb, local, dw_lds);
  }
}

}  // namespace

/* jobs_h: rsx_cin_dw_job with ws = the layer's rsx_cin_split_bwd_dx workspace (dc_rows is ignored: one row per example) */
static int cs_launch_dw(const float* X0, const rsx_cin_dw_job* jobs_h, int njobs, int B, int F, int D, int ns,
                        const float* const* parts_h, const int32_t* tiles_h, int nparts, float* dX0, int acc_dx0,
                        rsx_stream_t stream);
extern "C" int rsx_cin_split_bwd_dw(const float* X0, const rsx_cin_dw_job* jobs_h, int njobs, int B, int F, int D, int ns,
                                    rsx_stream_t stream) {
  return cs_launch_dw(X0, jobs_h, njobs, B, F, D, ns, nullptr, nullptr, 0, nullptr, 0, stream);
}
extern "C" int rsx_cin_split_bwd_dw_dx0(const float* X0, const rsx_cin_dw_job* jobs_h, int njobs, int B, int F, int D, int ns,
                                        const float* const* parts_h, const int32_t* tiles_h, int nparts, float* dX0, int acc_dx0,
                                        rsx_stream_t stream) {
  if (!parts_h || !tiles_h || nparts <= 0 || !dX0) return RSX_EINVAL;
  if (nparts > CS_MAXJ) return RSX_EUNSUPPORTED;
  for (int j = 0; j < nparts; ++j)
    if (!parts_h[j] || tiles_h[j] <= 0) return RSX_EINVAL;
  return cs_launch_dw(X0, jobs_h, njobs, B, F, D, ns, parts_h, tiles_h, nparts, dX0, acc_dx0, stream);
}
static int cs_launch_dw(const float* X0, const rsx_cin_dw_job* jobs_h, int njobs, int B, int F, int D, int ns,
                        const float* const* parts_h, const int32_t* tiles_h, int nparts, float* dX0, int acc_dx0,
                        rsx_stream_t stream) {
  if (!X0 || !jobs_h || njobs <= 0 || B < 0 || F <= 0 || ns < 1 || ns > CS_H2) return RSX_EINVAL;
  if (ns < 3) return RSX_EUNSUPPORTED;       // (round 6: only ns = 3 and mode 4 are built)
  if (njobs > CS_MAXJ || D != CS_D || F > CS_FP) return RSX_EUNSUPPORTED;
  if (B == 0) return RSX_OK;
  ns = cs_dw_planes(ns);                            // (mode 4: the data-gradient launch left three bf16 planes of dpre)
  CsDwArgs w{};
  w.njobs = njobs; w.X0 = X0; w.B = B; w.F = F;
  // fields per tile: 4 for the job with the most tiles, 3 for the others -- [128, 128] at F = 39: 160 + 78 = 238 workgroups, one
  // round on 256 CUs (3 everywhere: 286, a second round for 30 of them)
  int big = 0;
  for (int k = 1; k < njobs; ++k)
    if ((long long)jobs_h[k].H * jobs_h[k].N > (long long)jobs_h[big].H * jobs_h[big].N) big = k;
  static const int ft_env = getenv("RSX_CIN_SPLIT_DW_FT") ? atoi(getenv("RSX_CIN_SPLIT_DW_FT")) : 0;
  int tiles = 0;
  size_t x0_bytes = 0;
  for (int k = 0; k < njobs; ++k) {
    const rsx_cin_dw_job& j = jobs_h[k];
    if (!j.Xk || !j.ws || !j.dW || !j.dc || j.H <= 0 || j.N <= 0) return RSX_EINVAL;
    if (j.H > 128 || j.N > 128) return RSX_EUNSUPPORTED;
    const int H16 = rup(j.H, 16), N16 = rup(j.N, 16);
    CsDwJob& d = w.job[k];
    d.Xk = j.Xk;
    d.dpre16 = static_cast<const bf16_t*>(j.ws);
    d.dc_part = reinterpret_cast<const float*>(static_cast<const char*>(j.ws) + (size_t)ns * ((B + 1) / 2) * 2 * N16 * CS_D * 2);
    d.dW = j.dW; d.dc = j.dc; d.H = j.H; d.N = j.N; d.N16 = N16;
    d.ft = ft_env == 3 || ft_env == 4 ? ft_env : ((k == big && njobs > 1) ? 4 : 3);
    d.gx = (N16 + 16 * CS_NT - 1) / (16 * CS_NT);
    d.HT = H16 / 16;
    tiles += d.gx * d.HT * ((F + d.ft - 1) / d.ft);
    d.tile_end = tiles;
    const size_t xb = (size_t)d.ft * B * CS_D * 4;
    x0_bytes = xb > x0_bytes ? xb : x0_bytes;
  }
  const size_t red_bytes = (size_t)4 * 4 * CS_NT * 256 * 4;
  w.x0l = x0_bytes <= 160 * 1024;
  if (!w.x0l) {                                   // large batches: three fields per tile everywhere, X0 from L2
    tiles = 0;
    for (int k = 0; k < njobs; ++k) {
      CsDwJob& d = w.job[k];
      d.ft = 3;
      tiles += d.gx * d.HT * ((F + 2) / 3);
      d.tile_end = tiles;
    }
  }
  const size_t lds = (w.x0l && x0_bytes > red_bytes) ? x0_bytes : red_bytes;
  if (nparts > 0) {                                 // the dX0 reduce rides along: what the tiles leave of 256 CUs, at least 8 workgroups
    for (int j = 0; j < nparts; ++j) {
      w.red_parts[j] = parts_h[j];
      w.red_tiles[j] = tiles_h[j];
    }
    w.red_njobs = nparts;
    w.red_acc = acc_dx0;
    w.red_dX0 = dX0;
    w.red_n4 = (unsigned long long)B * F * CS_D / 4;
    const int used = (tiles + njobs) % 256;
    int rb = used == 0 ? 8 : 256 - used;
    if (rb < 8) rb = 8;
    const unsigned long long need = (w.red_n4 + 511) / 512;
    w.red_blocks = (unsigned long long)rb < need ? rb : (int)need;
  }
  const unsigned grid = (unsigned)tiles + (unsigned)njobs + (unsigned)w.red_blocks;
#define RSX_CS_DW(NS_)                                                       \
  {                                                                          \
    const int rc = opt_in_lds(cin_split_dw_k<NS_>, lds);                     \
    if (rc != RSX_OK) return rc;                                             \
    RSX_LAUNCH(cin_split_dw_k<NS_>, dim3(grid), dim3(512), lds, rsx_s(stream), w); \
  }
  RSX_CS_DW(3);
#undef RSX_CS_DW
  RSX_CHECK_LAUNCH();
  return RSX_OK;
}

#ifdef RSX_STAMPS
extern "C" int rsx_dbg_stamps_cin_split(unsigned long long* out_h) {
  return hipMemcpyFromSymbol(out_h, HIP_SYMBOL(rsx_stamps_d), sizeof(rsx_stamps_d)) == hipSuccess ? 0 : 1;
}
extern "C" int rsx_dbg_stamps_cin_split_reset() {
  unsigned long long z[64] = {};
  return hipMemcpyToSymbol(HIP_SYMBOL(rsx_stamps_d), z, sizeof(z)) == hipSuccess ? 0 : 1;
}
#endif
