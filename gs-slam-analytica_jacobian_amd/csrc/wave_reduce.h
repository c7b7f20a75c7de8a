// wave_reduce.h -- register-only wave64 reduction of 10 per-lane partials on gfx950.
//
// v_permlane32_swap / v_permlane16_swap exchange half-waves / odd-even rows between two
// registers, so ONE swap + ONE add both halves the number of live registers and folds one
// lane bit: 10 -> 5 -> 3 registers.  Four DPP row rotations then finish each 16-lane row.
// After reduce10():  row r = lane>>4 holds  x0 -> v[{0,2,1,3}[r]],  x1 -> v[{4,6,5,7}[r]],
//                    x2 -> v[{8,8,9,9}[r]]   in every lane of the row.
#pragma once
#include <hip/hip_runtime.h>

typedef unsigned gsaj_u32x2 __attribute__((ext_vector_type(2)));

// (A, B) -> lanes 0-31: A_lo + A_hi, lanes 32-63: B_lo + B_hi   (sum across lane bit 5)
__device__ __forceinline__ float merge32(float a, float b) {
  gsaj_u32x2 r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
// (A, B) -> even rows: A_r + A_{r+1}, odd rows: B_{r-1} + B_r      (sum across lane bit 4)
__device__ __forceinline__ float merge16(float a, float b) {
  gsaj_u32x2 r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
template <int CTRL>
__device__ __forceinline__ float dpp_add(float v) {
  return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float row_allreduce(float v) {  // sum over the 16 lanes of each row
  v = dpp_add<0x128>(v);  // row_ror:8
  v = dpp_add<0x124>(v);  // row_ror:4
  v = dpp_add<0x122>(v);  // row_ror:2
  v = dpp_add<0x121>(v);  // row_ror:1
  return v;
}
__device__ __forceinline__ void reduce10(const float (&v)[10], float &x0, float &x1, float &x2) {
  const float w0 = merge32(v[0], v[1]), w1 = merge32(v[2], v[3]), w2 = merge32(v[4], v[5]), w3 = merge32(v[6], v[7]),
              w4 = merge32(v[8], v[9]);
  x0 = row_allreduce(merge16(w0, w1));
  x1 = row_allreduce(merge16(w2, w3));
  x2 = row_allreduce(merge16(w4, w4));
}
// Lane 0 of each row stores its three totals into a 12-float slot (a[0..9] = v[0..9]).
__device__ __forceinline__ void store10(float *a, int lane, float x0, float x1, float x2) {
  if ((lane & 15) == 0) {
    const int r = lane >> 4;
    const int k = ((r & 1) << 1) | (r >> 1);
    a[k] = x0;
    a[4 + k] = x1;
    if ((r & 1) == 0) a[8 + (r >> 1)] = x2;
  }
}
