// Micro-benchmark: cost of reducing 10 per-lane partials over a wave64 on gfx950, several schemes.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include "../gs-slam-analytica_jacobian_amd/csrc/wave_reduce.h"
#define ITERS 2048
template <int CTRL> __device__ __forceinline__ float dppadd_rm(float v, int rm) {
  return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
// canonical reduce-to-lane-63: row_shr 1,2,3 ; row_shr 4 (bank 0xe), row_shr 8 (bank 0xc); row_bcast15 (row 0xa); row_bcast31 (row 0xc)
__device__ __forceinline__ float reduce_dpp(float v) {
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x111, 0xF, 0xF, true));
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x112, 0xF, 0xF, true));
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x114, 0xF, 0xF, true));
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x118, 0xF, 0xF, true));
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x142, 0xA, 0xF, true));
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x143, 0xC, 0xF, true));
  return v;
}
template <int MODE>
__global__ void k(float *out, float a) {
  float v[10];
  for (int c = 0; c < 10; c++) v[c] = threadIdx.x * 0.001f + c;
  float acc = 0.f;
  for (int i = 0; i < ITERS; i++) {
    if (MODE == 0) {
      float x0, x1, x2;
      reduce10(v, x0, x1, x2);
      acc += x0 + x1 + x2;
    } else if (MODE == 1) {
      for (int c = 0; c < 10; c++) acc += reduce_dpp(v[c]);
    } else if (MODE == 2) {
      for (int c = 0; c < 10; c++) { float t = v[c]; for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o); acc += t; }
    } else {  // no reduction: cheap dependency only
      for (int c = 0; c < 10; c++) acc += v[c];
    }
    for (int c = 0; c < 10; c++) v[c] = v[c] * a + acc * 1e-9f;  // keep a dependency so nothing hoists
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
template <int MODE> void run(const char *name, float *d) {
  hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  hipLaunchKernelGGL(k<MODE>, dim3(256 * 4), dim3(256), 0, 0, d, 1.0001f);
  (void)hipEventRecord(a);
  hipLaunchKernelGGL(k<MODE>, dim3(256 * 4), dim3(256), 0, 0, d, 1.0001f);
  (void)hipEventRecord(b); (void)hipEventSynchronize(b);
  float ms; (void)hipEventElapsedTime(&ms, a, b);
  printf("%-28s %.3f ms -> %.1f ns per reduction-of-10 per SIMD-slot (4 waves/SIMD)\n", name, ms, ms * 1e6 / (ITERS * 4.0));
}
int main() {
  float *d; (void)hipMalloc(&d, 256 * 4 * 256 * sizeof(float));
  run<3>("baseline (no reduction)", d); run<0>("permlane swap merge + DPP", d); run<1>("DPP scan-style x10", d); run<2>("__shfl_xor butterfly x10", d);
  return 0;
}
