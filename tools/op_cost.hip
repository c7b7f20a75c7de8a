// Micro-benchmark: issue cost (cycles per wave-instruction per SIMD at 2.4 GHz) of the VALU / cross-lane / LDS
// instruction kinds the compositors are made of, at 1..8 waves per SIMD, plus SALU co-issue.
// Build (travels as a code object so the GPU box gets it): hipcc --offload-arch=gfx950 -O3 tools/op_cost.hip -o tools/op_cost.co
#include <hip/hip_runtime.h>
#include <stdio.h>
#define ITERS 2048
typedef float f4v __attribute__((ext_vector_type(4)));
typedef float f2v __attribute__((ext_vector_type(2)));
typedef unsigned u2v __attribute__((ext_vector_type(2)));
#define REP8(X) X X X X X X X X
template <int MODE>
__global__ void k(float *out, float a, float b) {
  __shared__ float4 lds[1024];
  float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3;
  f4v q = {0, 0, 0, 0};
  f2v p = {x0, x1}, p2 = {x2, x3};
  u2v u = {threadIdx.x, 3};
  lds[threadIdx.x] = make_float4(x0, x1, x2, x3);
  __syncthreads();
  unsigned addr_b = (threadIdx.x >> 6) * 48;           // wave-uniform LDS address (broadcast read)
  unsigned addr_l = threadIdx.x * 16;                  // lane-distinct 16-B reads
  unsigned long long m = 0x5555555555555555ull;
  int si = 0;
  double d2 = 1.0 + threadIdx.x, e2 = 2.0 + threadIdx.x, e3 = 3.5;
  for (int i = 0; i < ITERS; i++) {
    if (MODE == 0) asm volatile(REP8("v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %3, %3, %1, %2\n") : "+v"(x0) : "v"(a), "v"(b), "v"(x1));
    if (MODE == 1) asm volatile(REP8("v_cmp_gt_f32_e64 s[20:21], %0, %1\n v_cmp_lt_f32_e64 s[22:23], %0, %2\n") : : "v"(x0), "v"(a), "v"(b) : "s20", "s21", "s22", "s23");
    if (MODE == 2) asm volatile(REP8("v_cndmask_b32_e64 %0, %1, %2, %3\n v_cndmask_b32_e64 %4, %2, %1, %3\n") : "+v"(x0) : "v"(a), "v"(b), "s"(m), "v"(x1));
    if (MODE == 3) asm volatile(REP8("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n") : "+v"(x0), "+v"(x1));
    if (MODE == 4) asm volatile(REP8("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n") : "+v"(x0), "+v"(x1));
    if (MODE == 5) asm volatile(REP8("v_add_f32_dpp %0, %0, %0 row_ror:4 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %1, %1, %1 row_ror:8 row_mask:0xf bank_mask:0xf\n") : "+v"(x0), "+v"(x1));
    if (MODE == 6) asm volatile(REP8("v_permlane32_swap_b32 %0, %1\n v_permlane16_swap_b32 %2, %3\n") : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3));
    if (MODE == 7) asm volatile(REP8("ds_read_b128 %0, %1\n ds_read_b128 %0, %1 offset:16\n") "s_waitcnt lgkmcnt(0)\n" : "=v"(q) : "v"(addr_b) : "memory");
    if (MODE == 8) asm volatile(REP8("ds_read_b128 %0, %1\n ds_read_b128 %0, %1 offset:16384\n") "s_waitcnt lgkmcnt(0)\n" : "=v"(q) : "v"(addr_l) : "memory");
    if (MODE == 9) asm volatile(REP8("v_fma_f32 %0, %0, %2, %3\n s_add_u32 %1, %1, 1\n") : "+v"(x0), "+s"(si) : "v"(a), "v"(b) : "scc");
    if (MODE == 10) asm volatile(REP8("v_mul_f32 %0, %0, %1\n v_sub_f32 %2, %2, %1\n") : "+v"(x0) : "v"(a), "v"(x1));
    if (MODE == 11) asm volatile(REP8("v_pk_mul_f32 %0, %0, %1\n v_pk_fma_f32 %0, %0, %1, %1\n") : "+v"(p) : "v"(p2));
    if (MODE == 12) asm volatile(REP8("v_readlane_b32 s20, %0, 3\n v_readfirstlane_b32 s21, %1\n") : : "v"(x0), "v"(x1) : "s20", "s21");
    if (MODE == 13) asm volatile(REP8("v_mad_u64_u32 %0, s[20:21], %1, %1, %0\n v_mov_b32 %2, %1\n") : "+v"(u), "+v"(x0) : "v"(x1) : "s20", "s21");
    if (MODE == 14) asm volatile(REP8("ds_read_b64 %0, %1\n ds_read_b64 %0, %1 offset:8\n") "s_waitcnt lgkmcnt(0)\n" : "=v"(p) : "v"(addr_l / 2 + (threadIdx.x & 7) * 520) : "memory");
    if (MODE == 15) asm volatile(REP8("ds_write_b64 %1, %0\n ds_write_b64 %1, %0 offset:8192\n") "s_waitcnt lgkmcnt(0)\n" : : "v"(p), "v"(addr_l / 2) : "memory");
    if (MODE == 16) asm volatile(REP8("v_cmp_gt_f32_e32 vcc, %0, %1\n v_cmp_lt_f32_e32 vcc, %0, %2\n") : : "v"(x0), "v"(a), "v"(b) : "vcc");
    if (MODE == 17) asm volatile(REP8("v_cndmask_b32_e32 %0, %1, %2, vcc\n v_cndmask_b32_e32 %3, %2, %1, vcc\n") : "+v"(x0) : "v"(a), "v"(b), "v"(x1) : "vcc");
    if (MODE == 18) asm volatile(REP8("v_cmp_gt_f32_e32 vcc, %0, %1\n v_cndmask_b32_e32 %0, %1, %2, vcc\n") : "+v"(x0) : "v"(a), "v"(b) : "vcc");
    if (MODE == 19) asm volatile(REP8("v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %3, %3, %1, %2\n") : "+v"(x0) : "s"(a), "v"(b), "v"(x1));
    if (MODE == 20) asm volatile(REP8("v_min_f32 %0, %0, %1\n v_med3_f32 %2, %2, %1, %0\n") : "+v"(x0) : "v"(a), "v"(x1));
    if (MODE == 21) asm volatile(REP8("s_and_b64 s[20:21], s[20:21], %0\n s_bcnt1_i32_b64 s22, s[20:21]\n") : : "s"(m) : "s20", "s21", "s22", "scc");
    if (MODE == 22) asm volatile(REP8("v_mbcnt_lo_u32_b32 %0, -1, 0\n v_mbcnt_hi_u32_b32 %0, -1, %0\n") : "+v"(u.x));
    if (MODE == 23) asm volatile(REP8("ds_bpermute_b32 %0, %1, %0\n ds_swizzle_b32 %2, %2 offset:0x041F\n") "s_waitcnt lgkmcnt(0)\n" : "+v"(x0) : "v"(addr_l), "v"(x1) : "memory");
    if (MODE == 24) asm volatile(REP8("ds_read_b128 %0, %1\n v_fma_f32 %2, %2, %3, %4\n v_fma_f32 %2, %2, %3, %4\n v_fma_f32 %2, %2, %3, %4\n v_fma_f32 %2, %2, %3, %4\n") "s_waitcnt lgkmcnt(0)\n" : "=v"(q), "+v"(addr_b), "+v"(x0) : "v"(a), "v"(b) : "memory");
    if (MODE == 25) asm volatile(REP8("v_cmp_gt_f32_e64 s[20:21], %0, %1\n s_and_b64 s[22:23], s[20:21], %2\n") : : "v"(x0), "v"(a), "s"(m) : "s20", "s21", "s22", "s23", "scc");
    if (MODE == 26) asm volatile(REP8("v_add_u32 %0, %0, %1\n v_lshlrev_b32 %2, 1, %2\n") : "+v"(u.x) : "v"(u.y), "v"(u.y));
    if (MODE == 27) asm volatile(REP8("v_mul_f32 %0, 0x3fb8aa3b, %0\n v_mul_f32 %1, 0x3fb8aa3b, %1\n") : "+v"(x0), "+v"(x1));
    if (MODE == 29) asm volatile(REP8("v_cmp_lt_u64_e64 s[20:21], %0, %1\n v_cmp_gt_u64_e64 s[22:23], %0, %1\n") : : "v"(d2), "v"(e2) : "s20", "s21", "s22", "s23");
    if (MODE == 30) asm volatile(REP8("v_min_f64 %0, %0, %1\n v_max_f64 %2, %2, %1\n") : "+v"(d2), "+v"(e2) : "v"(e3));
    if (MODE == 31) asm volatile(REP8("v_cmp_lt_f64_e64 s[20:21], %0, %1\n v_cmp_gt_f64_e64 s[22:23], %0, %1\n") : : "v"(d2), "v"(e2) : "s20", "s21", "s22", "s23");
    if (MODE == 32) asm volatile(REP8("v_mov_b64 %0, %1\n v_mov_b64 %2, %1\n") : "=v"(d2), "+v"(e3), "=v"(e2));
    if (MODE == 28) asm volatile(REP8("v_mov_b32_dpp %0, %1 row_bcast:15 row_mask:0xa bank_mask:0xf\n v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n") : "+v"(x0) : "v"(x1));
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + q.x + q.y + q.z + q.w + si + p.x + p.y + u.x;
}
template <int MODE>
void run(const char *name, float *d) {
  printf("%-44s", name);
  for (int w : {1, 2, 4, 8}) {
    int blocks = 256 * w;  // 256 threads = 4 waves = one per SIMD of a CU
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 1.0001f, 0.5f);
    hipEventRecord(a);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 1.0001f, 0.5f);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    double instr_per_simd = (double)ITERS * (MODE == 24 ? 40 : 16) * w;
    printf("  w%d: %5.2f", w, ms * 1e-3 * 2.4e9 / instr_per_simd);
  }
  printf("   cycles/instr/SIMD\n");
  fflush(stdout);
}
int main() {
  float *d; hipMalloc(&d, 256 * 8 * 256 * sizeof(float) * 4);
  run<0>("v_fma_f32 (2 chains)", d);
  run<10>("v_mul_f32 / v_sub_f32", d);
  run<1>("v_cmp_*_e64 -> SGPR pair", d);
  run<2>("v_cndmask_b32_e64 (SGPR mask)", d);
  run<3>("v_exp_f32", d);
  run<4>("v_rcp_f32", d);
  run<5>("v_add_f32_dpp row_ror", d);
  run<6>("v_permlane32_swap / v_permlane16_swap", d);
  run<11>("v_pk_mul_f32 / v_pk_fma_f32", d);
  run<12>("v_readlane / v_readfirstlane", d);
  run<13>("v_mad_u64_u32 / v_mov_b32", d);
  run<9>("v_fma_f32 + s_add_u32 (pairs)", d);
  run<7>("ds_read_b128 wave-uniform address", d);
  run<8>("ds_read_b128 lane-distinct (1 KB/instr)", d);
  run<14>("ds_read_b64 stride-65 pattern", d);
  run<15>("ds_write_b64 contiguous", d);
  run<16>("v_cmp_*_e32 -> vcc", d);
  run<17>("v_cndmask_b32_e32 (vcc)", d);
  run<18>("v_cmp_e32 + v_cndmask_e32 pairs", d);
  run<19>("v_fma_f32 with SGPR operand", d);
  run<20>("v_min_f32 / v_med3_f32", d);
  run<21>("s_and_b64 / s_bcnt1_i32_b64 (SALU only)", d);
  run<22>("v_mbcnt_lo / v_mbcnt_hi", d);
  run<23>("ds_bpermute_b32 / ds_swizzle_b32", d);
  run<24>("1 ds_read_b128 + 4 v_fma (per instr)", d);
  run<25>("v_cmp_e64 + s_and_b64 pairs", d);
  run<26>("v_add_u32 / v_lshlrev_b32", d);
  run<27>("v_mul_f32 with literal", d);
  run<28>("v_mov_b32_dpp row_bcast / row_shr", d);
  run<29>("v_cmp_lt_u64 / v_cmp_gt_u64 -> SGPR pair", d);
  run<30>("v_min_f64 / v_max_f64", d);
  run<31>("v_cmp_lt_f64 / v_cmp_gt_f64 -> SGPR pair", d);
  run<32>("v_mov_b64", d);
  return 0;
}
