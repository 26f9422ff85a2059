// VALU issue rates on gfx950, 4 waves / SIMD resident: v_fma_f32 vs v_pk_fma_f32 vs v_rcp_f32 / v_rsq_f32 (per wave64 instruction).
//   hipcc --offload-arch=gfx950 -O3 scripts/micro/pk_rate.hip -o scripts/_build/pk_rate && scripts/_build/pk_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
constexpr int IT = 4096;
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, float a, float b) {
  float x[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) x[i] = threadIdx.x * 1e-3f + i;
  for (int it = 0; it < IT; ++it) {
    if (MODE == 0) {
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(a), "v"(b));
    } else if (MODE == 1) {
#pragma unroll
      for (int i = 0; i < 16; i += 2) {
        f2 v = {x[i], x[i + 1]};
        f2 av = {a, a}, bv = {b, b};
        asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(v) : "v"(av), "v"(bv));
        x[i] = v.x; x[i + 1] = v.y;
      }
    } else if (MODE == 2) {
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_rcp_f32 %0, %0" : "+v"(x[i]));
    } else if (MODE == 3) {
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_rsq_f32 %0, %0" : "+v"(x[i]));
    } else if (MODE == 4) {
#pragma unroll
      for (int i = 0; i < 16; i += 2) {
        f2 v = {x[i], x[i + 1]};
        f2 av = {a, a};
        asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(v) : "v"(av));
        x[i] = v.x; x[i + 1] = v.y;
      }
    } else if (MODE == 5) {
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_max_f32 %0, %0, %1" : "+v"(x[i]) : "v"(a));
    }
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += x[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int MODE>
void run(const char* name, int instr_per_iter, float* out) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int grid = 256 * 4;
  k<MODE><<<grid, 256>>>(out, 0.999f, 1e-3f);
  hipEventRecord(e0);
  k<MODE><<<grid, 256>>>(out, 0.999f, 1e-3f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  // per SIMD: 4 waves (one per workgroup x 4 workgroups / CU ... 4 waves of a 256-thread block land on the 4 SIMDs)
  const double wave_instr_per_simd = 4.0 * IT * instr_per_iter;
  printf("%-14s %.3f ms  -> %.2f cycles per wave64 instruction at 2.4 GHz\n", name, ms, ms * 1e-3 * 2.4e9 / wave_instr_per_simd);
}
int main() {
  float* out;
  hipMalloc(&out, 1024 * 256 * 4);
  run<0>("v_fma_f32", 16, out);
  run<1>("v_pk_fma_f32", 8, out);
  run<4>("v_pk_mul_f32", 8, out);
  run<2>("v_rcp_f32", 16, out);
  run<3>("v_rsq_f32", 16, out);
  run<5>("v_max_f32", 16, out);
  return 0;
}
