// Split-operand products on the bf16 matrix cores (round 5: csrc/cin_split.hip, which keeps its own copy of these helpers;
// round 6: csrc/din_attn.hip).  v_mfma_f32_16x16x32_bf16 multiplies its 8-bit-significand operands EXACTLY and accumulates in
// fp32 at 16x the rate of v_mfma_f32_16x16x4_f32.  An fp32 number is the exact sum of three bf16 numbers (x = x1 + x2 + x3,
// x1 = bf16(x), x2 = bf16(x - x1), x3 = x - x1 - x2), so a product of two fp32 numbers is the sum of nine exact bf16 products;
// the six with i + j <= 4 carry everything above 2^-24 of the product: fp32-grade, 6 MFMAs = 3/8 of the fp32 MFMA's time.
#pragma once
#include "rsx_common.h"

typedef float sp_f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 sp_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 sp_bf16x2 __attribute__((ext_vector_type(2)));
typedef uint32_t sp_u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t sp_pack2(float lo, float hi) {
  sp_bf16x2 v;
  v[0] = (__bf16)lo;
  v[1] = (__bf16)hi;
  return __builtin_bit_cast(uint32_t, v);
}
// two fp32 values -> 3 packed bf16 pairs: plane s holds bf16 of what the planes before it left over
__device__ __forceinline__ void sp_split2(float lo, float hi, uint32_t (&out)[3]) {
#pragma unroll
  for (int s = 0; s < 3; ++s) {
    const uint32_t pk = sp_pack2(lo, hi);
    out[s] = pk;
    if (s + 1 < 3) {
      lo -= __uint_as_float(pk << 16);
      hi -= __uint_as_float(pk & 0xffff0000u);
    }
  }
}
// eight fp32 values -> 3 operand quads (element j of the quad = v[j])
__device__ __forceinline__ void sp_split8(const float (&v)[8], sp_bf16x8 (&out)[3]) {
  uint32_t p0[3], p1[3], p2[3], p3[3];
  sp_split2(v[0], v[1], p0);
  sp_split2(v[2], v[3], p1);
  sp_split2(v[4], v[5], p2);
  sp_split2(v[6], v[7], p3);
#pragma unroll
  for (int s = 0; s < 3; ++s) out[s] = __builtin_bit_cast(sp_bf16x8, (sp_u32x4){p0[s], p1[s], p2[s], p3[s]});
}
// T[m][n] += sum_k X[m][k] Y[k][n] with x / y the three planes of the two operands' fragments (x: the MFMA's first operand,
// rows m; y: its second, columns n): the six kept plane products, smallest first
__device__ __forceinline__ sp_f32x4 sp_mma3(const sp_bf16x8 (&x)[3], const sp_bf16x8 (&y)[3], sp_f32x4 T) {
  T = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x[0], y[2], T, 0, 0, 0);
  T = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x[1], y[1], T, 0, 0, 0);
  T = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x[2], y[0], T, 0, 0, 0);
  T = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x[0], y[1], T, 0, 0, 0);
  T = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x[1], y[0], T, 0, 0, 0);
  T = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x[0], y[0], T, 0, 0, 0);
  return T;
}
