// Micro-benchmark: wave64 VALU issue rate on gfx950 -- plain v_fma_f32 vs packed v_pk_fma_f32 vs v_exp_f32,
// as a function of waves per SIMD.  Build: hipcc --offload-arch=gfx950 -O3 tools/valu_rate.hip -o tools/valu_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float float2v __attribute__((ext_vector_type(2)));
#define ITERS 4096
template <int MODE>
__global__ void k(float *out, float a, float b) {
  float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
  float2v p0 = {x0, x1}, p1 = {x2, x3}, p2 = {x4, x5}, p3 = {x6, x7}, pa = {a, a}, pb = {b, b};
  for (int i = 0; i < ITERS; i++) {
    if (MODE == 0) {  // 8 independent scalar fma chains
      asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                   "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                   : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));
    } else if (MODE == 1) {  // 4 packed fma chains (= 8 fmas) x2 to match instruction count 8
      asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
                   "v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
                   : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pa), "v"(pb));
    } else {  // 8 v_exp_f32
      asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n"
                   "v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7\n"
                   : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7));
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y;
}
template <int MODE>
void run(const char *name, int waves_per_simd, float *d) {
  int blocks = 256 * waves_per_simd;  // 256 threads = 4 waves = one per SIMD of a CU
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 1.0001f, 0.5f);
  hipEventRecord(a);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 1.0001f, 0.5f);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  double instr_per_simd = (double)ITERS * 8 * waves_per_simd;  // wave-instructions issued by each SIMD
  double cycles = ms * 1e-3 * 2.4e9;
  printf("%-12s waves/SIMD=%d  %.3f ms  -> %.2f cycles per wave-instruction per SIMD (at 2.4 GHz)\n", name, waves_per_simd, ms,
         cycles / instr_per_simd);
}
int main() {
  float *d; hipMalloc(&d, 256 * 8 * 256 * sizeof(float) * 4);
  for (int w : {1, 2, 4, 8}) { run<0>("v_fma_f32", w, d); run<1>("v_pk_fma_f32", w, d); run<2>("v_exp_f32", w, d); }
  return 0;
}
