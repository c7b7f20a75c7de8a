// The batched contraction cov2D = T^T Sigma T of the EWA projection (forward.cu:76-115; its transpose chain in
// backward.cu:226-273) on the MATRIX cores, against the per-lane VALU form the rasteriser uses -- BASELINE.json's north star
// allows MFMA "only for the batched J Sigma J^T contractions"; this settles with a measurement whether it pays.
//
//   VALU: lane = Gaussian; Sigma (6 floats), T (3x3: the third column of J W is zero, kept general here) in registers;
//         M = Sigma T (27 fma), cov = T^T M (upper triangle of the 2x2 block: 9 fma).  No cross-lane traffic, no LDS.
//   MFMA: v_mfma_f32_4x4x1_16b_f32 -- 16 independent 4x4 += (4x1)(1x4) blocks per instruction, the one shape made for batches
//         of tiny matrices.  A wave takes 64 Gaussians in 4 groups of 16; 3x3 matrices padded to 4x4.  Lane (b, r) of a group
//         supplies A[b][r] and B[b][r] and receives D[b][0..3][r] (4 registers).  First product M = Sigma T: step k feeds
//         column k of Sigma (A) and row k of T (B).  Second product cov = T^T M: step k feeds row k of T as A (already in
//         the lane) and row k of M as B -- which is exactly register k of the first product's D: no data moves between the two
//         products.  What does move: the operands from "lane = Gaussian" into "lane = (Gaussian, row)" through LDS.
//
// Both kernels repeat the contraction REPS times on data kept on chip (the inputs are 15 floats per Gaussian either way), so
// the figure is the arithmetic path's cost, not HBM's.
// Build: hipcc --offload-arch=gfx950 -O3 tools/mfma_jsj.hip -o tools/mfma_jsj.co ; run on the GPU box: ./tools/mfma_jsj.co
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define REPS 256
typedef float f4v __attribute__((ext_vector_type(4)));

// in: per Gaussian 15 floats {Sigma xx xy xz yy yz zz, T row-major 3x3}; out: 3 floats {cov00, cov01, cov11}
__global__ __launch_bounds__(256) void k_valu(int n, const float *__restrict__ in, float *__restrict__ out, float eps) {
  const int g = blockIdx.x * 256 + threadIdx.x;
  if (g >= n) return;
  float v[15];
#pragma unroll
  for (int i = 0; i < 15; i++) v[i] = in[(size_t)g * 15 + i];
  float c00 = 0.f, c01 = 0.f, c11 = 0.f;
  for (int r = 0; r < REPS; r++) {
    const float S[3][3] = {{v[0], v[1], v[2]}, {v[1], v[3], v[4]}, {v[2], v[4], v[5]}};
    float M[3][3];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
      for (int j = 0; j < 3; j++) M[i][j] = S[i][0] * v[6 + j] + S[i][1] * v[9 + j] + S[i][2] * v[12 + j];
    c00 += v[6] * M[0][0] + v[9] * M[1][0] + v[12] * M[2][0];
    c01 += v[6] * M[0][1] + v[9] * M[1][1] + v[12] * M[2][1];
    c11 += v[7] * M[0][1] + v[10] * M[1][1] + v[13] * M[2][1];
#pragma unroll
    for (int i = 0; i < 15; i++) asm volatile("" : "+v"(v[i]));  // (nothing of the contraction may be hoisted out of the loop)
  }
  out[(size_t)g * 3 + 0] = c00;
  out[(size_t)g * 3 + 1] = c01;
  out[(size_t)g * 3 + 2] = c11;
}

__global__ __launch_bounds__(256) void k_mfma(int n, const float *__restrict__ in, float *__restrict__ out, float eps) {
  // per wave: S rows and T columns of its 64 Gaussians, 4 floats per (Gaussian, row): [wave][gaussian][row][4]
  __shared__ float4 sS[4][64][4], sTt[4][64][4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = blockIdx.x * 256 + threadIdx.x;
  float v[15];
#pragma unroll
  for (int i = 0; i < 15; i++) v[i] = g < n ? in[(size_t)g * 15 + i] : 0.f;
  f4v acc[4];
#pragma unroll
  for (int q = 0; q < 4; q++) acc[q] = (f4v){0.f, 0.f, 0.f, 0.f};
  for (int r = 0; r < REPS; r++) {
    // lane = Gaussian -> LDS (rows of Sigma; columns of T, i.e. rows of T^T; padded to 4x4 with zeros)
    sS[wave][lane][0] = make_float4(v[0], v[1], v[2], 0.f);
    sS[wave][lane][1] = make_float4(v[1], v[3], v[4], 0.f);
    sS[wave][lane][2] = make_float4(v[2], v[4], v[5], 0.f);
    sS[wave][lane][3] = make_float4(0.f, 0.f, 0.f, 0.f);
    sTt[wave][lane][0] = make_float4(v[6], v[9], v[12], 0.f);
    sTt[wave][lane][1] = make_float4(v[7], v[10], v[13], 0.f);
    sTt[wave][lane][2] = make_float4(v[8], v[11], v[14], 0.f);
    sTt[wave][lane][3] = make_float4(0.f, 0.f, 0.f, 0.f);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int q = 0; q < 4; q++) {  // group q: Gaussians 16 q .. 16 q + 15 of the wave; this lane = (b, row) = (lane / 4, lane % 4)
      const int gg = q * 16 + (lane >> 2), row = lane & 3;
      const float4 s = sS[wave][gg][row];    // Sigma[row][0..3]: A operand of step k is s[k] (column k of Sigma, element `row`)
      const float4 t = sTt[wave][gg][row];   // T[0..3][row]:     B operand of step k is t[k] (row k of T, element `row`)
      f4v m = {0.f, 0.f, 0.f, 0.f};          // M[0..3][row] of this Gaussian
      m = __builtin_amdgcn_mfma_f32_4x4x1f32(s.x, t.x, m, 0, 0, 0);
      m = __builtin_amdgcn_mfma_f32_4x4x1f32(s.y, t.y, m, 0, 0, 0);
      m = __builtin_amdgcn_mfma_f32_4x4x1f32(s.z, t.z, m, 0, 0, 0);
      // cov = T^T M: A of step k = T[k][row] = t[k], B of step k = M[k][row] = m[k]
      f4v c = acc[q];
      c = __builtin_amdgcn_mfma_f32_4x4x1f32(t.x, m.x, c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_4x4x1f32(t.y, m.y, c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_4x4x1f32(t.z, m.z, c, 0, 0, 0);
      acc[q] = c;
    }
#pragma unroll
    for (int i = 0; i < 15; i++) asm volatile("" : "+v"(v[i]));
    __builtin_amdgcn_wave_barrier();
  }
  // acc[q][i] at lane (b, j) = cov[i][j] of Gaussian 16 q + b
#pragma unroll
  for (int q = 0; q < 4; q++) {
    const int gg = blockIdx.x * 256 + wave * 64 + q * 16 + (lane >> 2), col = lane & 3;
    if (gg < n) {
      if (col == 0) out[(size_t)gg * 3 + 0] = acc[q].x;
      if (col == 1) {
        out[(size_t)gg * 3 + 1] = acc[q].x;
        out[(size_t)gg * 3 + 2] = acc[q].y;
      }
    }
  }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)

int main() {
  const int n = 256 * 2048;  // 524 288 Gaussians: 8 workgroups per CU
  std::vector<float> h((size_t)n * 15);
  srand(7);
  for (size_t g = 0; g < (size_t)n; g++) {
    float L[3][3] = {{0}};
    for (int i = 0; i < 3; i++) for (int j = 0; j <= i; j++) L[i][j] = (float)(rand() & 0xffff) / 65535.0f - 0.3f;
    float S[3][3];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) S[i][j] = L[i][0] * L[j][0] + L[i][1] * L[j][1] + L[i][2] * L[j][2];
    float *v = &h[g * 15];
    v[0] = S[0][0]; v[1] = S[0][1]; v[2] = S[0][2]; v[3] = S[1][1]; v[4] = S[1][2]; v[5] = S[2][2];
    for (int i = 0; i < 9; i++) v[6 + i] = (float)(rand() & 0xffff) / 65535.0f - 0.5f;
  }
  float *din, *do1, *do2;
  CK(hipMalloc(&din, h.size() * 4));
  CK(hipMalloc(&do1, (size_t)n * 12));
  CK(hipMalloc(&do2, (size_t)n * 12));
  CK(hipMemcpy(din, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  float ms[2] = {0, 0};
  for (int which = 0; which < 2; which++) {
    for (int it = 0; it < 6; it++) {  // (first launches warm up)
      CK(hipEventRecord(e0));
      if (which == 0) hipLaunchKernelGGL(k_valu, dim3(n / 256), dim3(256), 0, 0, n, din, do1, 0.f);
      else hipLaunchKernelGGL(k_mfma, dim3(n / 256), dim3(256), 0, 0, n, din, do2, 0.f);
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      float t;
      CK(hipEventElapsedTime(&t, e0, e1));
      if (it == 1 || t < ms[which]) ms[which] = t;
    }
  }
  std::vector<float> o1((size_t)n * 3), o2((size_t)n * 3);
  CK(hipMemcpy(o1.data(), do1, o1.size() * 4, hipMemcpyDeviceToHost));
  CK(hipMemcpy(o2.data(), do2, o2.size() * 4, hipMemcpyDeviceToHost));
  double worst = 0, scale = 0;
  for (size_t i = 0; i < o1.size(); i++) {
    worst = fmax(worst, fabs((double)o1[i] - o2[i]));
    scale = fmax(scale, fabs((double)o1[i]));
  }
  // CPU check of the VALU kernel on a few Gaussians (fp64)
  double worst_cpu = 0;
  for (int g = 0; g < 1000; g++) {
    const float *v = &h[(size_t)g * 15];
    const double S[3][3] = {{v[0], v[1], v[2]}, {v[1], v[3], v[4]}, {v[2], v[4], v[5]}};
    double M[3][3], c[3];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) M[i][j] = S[i][0] * v[6 + j] + S[i][1] * v[9 + j] + S[i][2] * v[12 + j];
    c[0] = v[6] * M[0][0] + v[9] * M[1][0] + v[12] * M[2][0];
    c[1] = v[6] * M[0][1] + v[9] * M[1][1] + v[12] * M[2][1];
    c[2] = v[7] * M[0][1] + v[10] * M[1][1] + v[13] * M[2][1];
    for (int k = 0; k < 3; k++) worst_cpu = fmax(worst_cpu, fabs(c[k] * REPS - o1[(size_t)g * 3 + k]));
  }
  const double per = 1e-3 / ((double)n * REPS);  // ms -> s per contraction
  printf("cov2D = T^T Sigma T, %d Gaussians x %d repetitions on chip (MI355X)\n", n, REPS);
  printf("  VALU (lane = Gaussian, 36 fma)              : %.3f ms = %.1f ps per contraction = %.2f G contractions/s\n", ms[0], ms[0] * per * 1e12, 1e-9 / (ms[0] * per));
  printf("  MFMA (v_mfma_f32_4x4x1_16b_f32, 6 per 16)   : %.3f ms = %.1f ps per contraction = %.2f G contractions/s\n", ms[1], ms[1] * per * 1e12, 1e-9 / (ms[1] * per));
  printf("  MFMA / VALU time: %.2f\n", ms[1] / ms[0]);
  printf("  max |MFMA - VALU| = %.3e of max |value| %.3e (relative %.2e); VALU vs fp64 on 1000 Gaussians: %.3e\n", worst, scale, worst / scale, worst_cpu);
  printf("  a frame of 10^6 Gaussians spends %.1f us (VALU) / %.1f us (MFMA) in this contraction\n", ms[0] * per * 1e6 * 1e6, ms[1] * per * 1e6 * 1e6);
  return worst / scale < 1e-5 ? 0 : 2;
}
