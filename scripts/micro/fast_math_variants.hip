// Which shorter forms of the packed sqrt / div are still correctly rounded?  (Development aid for csrc/adam_fast.h.)
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off scripts/micro/fast_math_variants.hip -o scripts/_build/fmv && scripts/_build/fmv [exhaustive_div_variant]
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
typedef float f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f2 pfma(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ float as_f(uint32_t u) { return __builtin_bit_cast(float, u); }
__device__ __forceinline__ uint32_t as_u(float f) { return __builtin_bit_cast(uint32_t, f); }

template <int V>
__device__ __forceinline__ f2 sqrt_v(const f2 x) {
  const f2 r = {__builtin_amdgcn_rsqf(x.x), __builtin_amdgcn_rsqf(x.y)};
  f2 g = x * r;
  f2 h = r * 0.5f;
  if (V == 0 || V == 1) {
    const f2 e = pfma(-h, g, (f2){0.5f, 0.5f});
    if (V == 0) h = pfma(h, e, h);
    g = pfma(g, e, g);
  }
  const f2 d = pfma(-g, g, x);
  g = pfma(d, h, g);
  if (V == 3) {                       // second residual step instead of the e-refinement
    const f2 d2 = pfma(-g, g, x);
    g = pfma(d2, h, g);
  }
  return g;
}
template <int V>
__device__ __forceinline__ f2 div_v(const f2 n, const f2 d) {
  f2 r = {__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
  if (V != 2) {
    const f2 e = pfma(-d, r, (f2){1.f, 1.f});
    r = pfma(e, r, r);
  }
  f2 q = n * r;
  f2 e = pfma(-d, q, n);
  q = pfma(e, r, q);
  if (V != 1) {
    e = pfma(-d, q, n);
    q = pfma(e, r, q);
  }
  return q;
}

template <int V>
__global__ void sqrt_all(uint32_t lo, uint32_t hi, unsigned long long* bad) {
  const uint32_t n = hi - lo;
  unsigned long long mine = 0;
  for (uint64_t i = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * 2u; i < n; i += (uint64_t)gridDim.x * blockDim.x * 2u) {
    const float x0 = as_f(lo + (uint32_t)i), x1 = as_f(lo + (uint32_t)i + 1);
    const f2 s = sqrt_v<V>((f2){x0, x1});
    mine += as_u(s.x) != as_u(sqrtf(x0));
    mine += as_u(s.y) != as_u(sqrtf(x1));
  }
  if (mine) atomicAdd(bad, mine);
}
__device__ __forceinline__ uint32_t mix(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}
template <int V>
__global__ void div_rand(uint32_t seed, int iters, unsigned long long* bad) {
  uint32_t s = mix(seed ^ (blockIdx.x * blockDim.x + threadIdx.x) * 0x9e3779b9u);
  unsigned long long mine = 0;
  for (int it = 0; it < iters; ++it) {
    float n[2], d[2];
    for (int k = 0; k < 2; ++k) {
      s = mix(s + 0x632be5abu);
      const uint32_t en = 127 - 94 + s % (94 + 34);
      s = mix(s + 1);
      const uint32_t ed = 127 - 30 + s % (30 + 21);
      s = mix(s + 2);
      uint32_t mn = s & 0x7fffffu;
      const uint32_t sg = s >> 31;
      if ((s >> 23 & 31u) == 0u) mn = (s >> 28) & 1u ? 0x7fffffu : 0u;
      s = mix(s + 3);
      uint32_t md = s & 0x7fffffu;
      if ((s >> 23 & 15u) == 0u) md = (s >> 27) & 1u ? 0x7fffffu : 0u;
      n[k] = as_f((sg << 31) | (en << 23) | mn);
      d[k] = as_f((ed << 23) | md);
    }
    const f2 q = div_v<V>((f2){n[0], n[1]}, (f2){d[0], d[1]});
    mine += as_u(q.x) != as_u(n[0] / d[0]);
    mine += as_u(q.y) != as_u(n[1] / d[1]);
  }
  if (mine) atomicAdd(bad, mine);
}
// every (numerator mantissa, denominator mantissa) pair, exponents 0 / 0: thread <- denominator mantissas md, md + stride, ...
template <int V>
__global__ void div_all(uint32_t md_lo, uint32_t md_hi, unsigned long long* bad) {
  unsigned long long mine = 0;
  const uint32_t md = md_lo + blockIdx.x;
  if (md >= md_hi) return;
  const float d = as_f(0x3f800000u | md);
  for (uint32_t mn = threadIdx.x * 2u; mn < (1u << 23); mn += blockDim.x * 2u) {
    const float n0 = as_f(0x3f800000u | mn), n1 = as_f(0x3f800000u | (mn + 1u));
    const f2 q = div_v<V>((f2){n0, n1}, (f2){d, d});
    mine += as_u(q.x) != as_u(n0 / d);
    mine += as_u(q.y) != as_u(n1 / d);
  }
  if (mine) atomicAdd(bad, mine);
}

template <int V>
unsigned long long run_sqrt(unsigned long long* bad) {
  unsigned long long hb = 0;
  hipMemcpy(bad, &hb, 8, hipMemcpyHostToDevice);
  sqrt_all<V><<<4096, 256>>>((127u - 96u) << 23, (127u + 41u) << 23, bad);
  hipDeviceSynchronize();
  hipMemcpy(&hb, bad, 8, hipMemcpyDeviceToHost);
  return hb;
}
template <int V>
unsigned long long run_div(unsigned long long* bad) {
  unsigned long long hb = 0;
  hipMemcpy(bad, &hb, 8, hipMemcpyHostToDevice);
  div_rand<V><<<8192, 256>>>(777u, 4096, bad);
  hipDeviceSynchronize();
  hipMemcpy(&hb, bad, 8, hipMemcpyDeviceToHost);
  return hb;
}
template <int V>
unsigned long long run_div_all(unsigned long long* bad) {
  unsigned long long hb = 0;
  hipMemcpy(bad, &hb, 8, hipMemcpyHostToDevice);
  for (uint32_t lo = 0; lo < (1u << 23); lo += (1u << 18)) {       // 32 launches of 262 144 workgroups
    div_all<V><<<1u << 18, 256>>>(lo, lo + (1u << 18), bad);
    hipDeviceSynchronize();
  }
  hipMemcpy(&hb, bad, 8, hipMemcpyDeviceToHost);
  return hb;
}
int main(int argc, char** argv) {
  unsigned long long* bad;
  hipMalloc(&bad, 16);
  printf("sqrt mismatches over [2^-96, 2^41): full %llu | no h refinement %llu | no refinement %llu | two residual steps %llu\n",
         run_sqrt<0>(bad), run_sqrt<1>(bad), run_sqrt<2>(bad), run_sqrt<3>(bad));
  printf("div mismatches over 1.7e10 random pairs: full %llu | one correction %llu | unrefined reciprocal, two corrections %llu\n",
         run_div<0>(bad), run_div<1>(bad), run_div<2>(bad));
  const int ex = argc > 1 ? atoi(argv[1]) : -1;
  if (ex == 0) printf("div exhaustive (2^46 mantissa pairs), full: %llu\n", run_div_all<0>(bad));
  if (ex == 1) printf("div exhaustive (2^46 mantissa pairs), one correction: %llu\n", run_div_all<1>(bad));
  if (ex == 2) printf("div exhaustive (2^46 mantissa pairs), unrefined reciprocal: %llu\n", run_div_all<2>(bad));
  return 0;
}
