// Packed (two fp32 per instruction: v_pk_fma_f32 / v_pk_mul_f32) square root and division for the zero-gradient Adam update
// of the window sweep, CORRECTLY ROUNDED on a restricted domain -- the same bits as sqrtf() and '/' there, at a third of
// their instruction count (the compiler's IEEE expansions are scalar per element and carry denormal / overflow scaling and
// special-case fix-ups that these domains exclude).  Both claims are checked by rsx_adam_fast_math_selftest
// (tests/test_gpu_fast_math.py), the first exhaustively:
//   rsx_sqrt2_fast(x):   2^-96 <= x < 2^41.  g = x * rsq(x), one residual step g += (x - g * g) * (rsq(x) / 2).  (LLVM's own
//                        expansion refines g and the half-reciprocal once more before the residual step; over EVERY float of
//                        the domain the result is the same without.)
//   rsx_div2_fast(n, d): 2^-30 <= d <= 2^21 and (n == +0 or 2^-94 <= |n| <= 2^34).  v_rcp_f32, one Newton step on the
//                        reciprocal, q = n * r, one residual correction q += (n - d * q) * r.  The compiler's sequence for
//                        '/' is this plus v_div_scale_f32 / v_div_fmas_f32 / v_div_fixup_f32 -- which on this domain are the
//                        identity (numerator exponent > 23, |exponent difference| < 96, quotient and reciprocal normal) --
//                        and a second residual correction.  Without the second correction the quotient is still the
//                        correctly rounded one for ALL 2^46 pairs of mantissas (exponents 0 / 0; every operation of the
//                        sequence commutes with scaling an operand by a power of two while nothing leaves the normal range,
//                        which the domain guarantees), and on 3.4e9 structured random pairs over the domain's exponents.
#pragma once
#include <hip/hip_runtime.h>

typedef float rsx_f2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ rsx_f2 rsx_pk_fma(rsx_f2 a, rsx_f2 b, rsx_f2 c) { return __builtin_elementwise_fma(a, b, c); }

__device__ __forceinline__ rsx_f2 rsx_sqrt2_fast(const rsx_f2 x) {
  const rsx_f2 r = {__builtin_amdgcn_rsqf(x.x), __builtin_amdgcn_rsqf(x.y)};
  const rsx_f2 g = x * r;
  const rsx_f2 h = r * 0.5f;
  const rsx_f2 d = rsx_pk_fma(-g, g, x);
  return rsx_pk_fma(d, h, g);
}

__device__ __forceinline__ rsx_f2 rsx_div2_fast(const rsx_f2 n, const rsx_f2 d) {
  rsx_f2 r = {__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
  rsx_f2 e = rsx_pk_fma(-d, r, (rsx_f2){1.f, 1.f});
  r = rsx_pk_fma(e, r, r);
  const rsx_f2 q = n * r;
  e = rsx_pk_fma(-d, q, n);
  return rsx_pk_fma(e, r, q);
}
