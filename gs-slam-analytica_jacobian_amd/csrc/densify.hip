// densify.hip -- bookkeeping the densify / prune logic reads after a backward, gfx950.
//
// Reference (callers' side of the render path): utils/slam_backend.py:113-121, :276-285, :344-352 -- per rendered view
//     max_radii2D[vis] = max(max_radii2D[vis], radii[vis]);  add_densification_stats(viewspace_points, vis)
// with vis = radii > 0 and gaussian_model.py:767-771
//     xyz_gradient_accum[vis] += ||viewspace_points.grad[vis, :2]||;   denom[vis] += 1
// and :236-250: n_obs = number of window keyframes in which the Gaussian was touched (n_touched > 0).
// The reference runs ~6 small torch kernels with boolean-mask gathers per view; here ONE launch handles the K views of a
// window (K = 1: one view), one lane per Gaussian, views in order (deterministic), every array touched once.
#include "gsaj_common.h"

__global__ __launch_bounds__(256) void k_densification_stats(int K, int P, const float *__restrict__ dL_dmean2D,
                                                             const int *__restrict__ radii, const int *__restrict__ n_touched,
                                                             float *__restrict__ xyz_gradient_accum, float *__restrict__ denom,
                                                             float *__restrict__ max_radii2D, int *__restrict__ n_obs) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= P) return;
  float acc = xyz_gradient_accum ? xyz_gradient_accum[i] : 0.f, den = denom ? denom[i] : 0.f;
  float mr = max_radii2D ? max_radii2D[i] : 0.f;
  int obs = 0;
  for (int v = 0; v < K; v++) {
    const size_t row = (size_t)v * P + i;
    const int r = radii[row];
    if (r > 0) {  // visibility_filter
      const float gx = dL_dmean2D[3 * row], gy = dL_dmean2D[3 * row + 1];
      acc += sqrtf(gx * gx + gy * gy);
      den += 1.f;
      mr = fmaxf(mr, (float)r);
    }
    if (n_touched && n_touched[row] > 0) obs++;
  }
  if (xyz_gradient_accum) xyz_gradient_accum[i] = acc;
  if (denom) denom[i] = den;
  if (max_radii2D) max_radii2D[i] = mr;
  if (n_obs) n_obs[i] = obs;
}

extern "C" int gsaj_densification_stats(int K, int P, const float *dL_dmean2D, const int *radii, const int *n_touched,
                                        float *xyz_gradient_accum, float *denom, float *max_radii2D, int *n_obs, void *stream) {
  if (K <= 0 || P <= 0 || !dL_dmean2D || !radii) {
    gsaj_set_error("gsaj_densification_stats: invalid argument (K=%d P=%d)", K, P);
    return GSAJ_ERR_INVALID_ARGUMENT;
  }
  hipLaunchKernelGGL(k_densification_stats, dim3((P + 255) / 256), dim3(256), 0, (hipStream_t)stream, K, P, dL_dmean2D, radii,
                     n_touched, xyz_gradient_accum, denom, max_radii2D, n_obs);
  GSAJ_HIP_CHECK(hipGetLastError());
  return GSAJ_OK;
}
