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

// ---- lane-parallel relevance test of one list entry against an 8x8 pixel quadrant -----------------
// Returns false only if NO pixel of the quadrant [X0, X0+7] x [Y0, Y0+7] can pass the compositor's
// per-pixel tests (power <= 0 and alpha = o * exp(power) >= 1/255): the quadratic form
// q = a dx^2 + 2 b dx dy + c dy^2 (power = -q/2) is minimised over the continuous box, and the
// entry is dropped when o * exp(-q_min / 2) stays below 1/255 by a safety factor that covers the
// fp32 / v_exp_f32 rounding of the per-pixel evaluation.  Non positive-definite conics are kept.
// This only removes work whose result is "skip" for all 64 lanes -- results are unchanged.
__device__ __forceinline__ bool quadrant_relevant(float mx, float my, float a, float b, float c, float o, float X0,
                                                  float Y0) {
  const float dxl = mx - (X0 + 7.0f), dxh = mx - X0;
  const float dyl = my - (Y0 + 7.0f), dyh = my - Y0;
  const bool pd = a > 0.f && c > 0.f && (a * c - b * b) > 0.f;
  const bool inside = dxl <= 0.f && dxh >= 0.f && dyl <= 0.f && dyh >= 0.f;
  const float ic = __builtin_amdgcn_rcpf(c), ia = __builtin_amdgcn_rcpf(a);
  float q = 3.0e38f;
  {
    const float y0 = fminf(fmaxf(-b * dxl * ic, dyl), dyh);
    q = fminf(q, a * dxl * dxl + 2.f * b * dxl * y0 + c * y0 * y0);
    const float y1 = fminf(fmaxf(-b * dxh * ic, dyl), dyh);
    q = fminf(q, a * dxh * dxh + 2.f * b * dxh * y1 + c * y1 * y1);
    const float x0 = fminf(fmaxf(-b * dyl * ia, dxl), dxh);
    q = fminf(q, a * x0 * x0 + 2.f * b * x0 * dyl + c * dyl * dyl);
    const float x1 = fminf(fmaxf(-b * dyh * ia, dxl), dxh);
    q = fminf(q, a * x1 * x1 + 2.f * b * x1 * dyh + c * dyh * dyh);
  }
  // keep if o * exp(-q/2) >= (1/255) * 0.99, evaluated conservatively (q shrunk by a relative 1e-3 and 1e-3 absolute)
  const float qs = fmaxf(q * 0.999f - 1.0e-3f, 0.f);
  const bool reach = o * __expf(-0.5f * qs) >= (0.99f / 255.0f);
  return !pd || inside || reach;
}
