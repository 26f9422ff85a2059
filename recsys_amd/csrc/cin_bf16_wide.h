// Launchers of cin_bf16_wide.hip (8 examples per workgroup), called by the entry points in cin_bf16.hip.
#pragma once
#include <hip/hip_runtime.h>

bool cin_wide_supported(int F, int H, int N);
int cin_wide_fwd(const float* X0, const float* Xk, const void* wt16, const float* c, float* out, int B, int F, int H, int N,
                 hipStream_t stream);
int cin_wide_dx(const float* X0, const float* Xk, const void* w16, const float* out, const float* dout, const float* gs,
                const float* wout, float* dXk, int acc_dxk, float* dx0_parts, void* dpre16, float* dc_part, int B, int F,
                int H, int N, hipStream_t stream);
int cin_wide_dx0_reduce(const float* const* parts, const int* tiles, int njobs, float* dX0, int acc, int B, int F,
                        hipStream_t stream);
