// dense.hip -- the reference's CPU/NumPy analytic path on gfx950 ("dense" semantics,
// SURVEY Appendix A.4): every Gaussian at every pixel, global depth order, alpha = clip(o G, 0, 1),
// no tile culling, no rasteriser cut-offs.
//
//   k_dense_bwd      <- compute_gradients_2D_vectorized_chunked  (Loss_Derivative_script_compare.py:1173-1351)
//   k_dense_render   <- rendered_Image_from_Projected_Gaussians_vectorized (:973-1018) + depth
//   k_pose_jacobians <- GetAnalyticalJcobian / compute_analytical_jacobians_all_gaussians (:633-760)
//   k_dense_tau      <- the dL/dtau chain-rule loop (:1587-1695) incl. compute_sh_backward_single
//                       (:452-532) and dnormvdv (:434-449)
//
// One pixel per lane; the Gaussian list is staged through LDS in chunks.  Pass 1 composites
// front-to-back for the per-pixel totals; pass 2 repeats the walk, forms S_i (the sum over the
// Gaussians behind i) as total - prefix in fp64, and reduces the 10 per-pixel partials of each
// Gaussian with the same register-only wave reduction as the tiled backward.  Workgroup partials
// go to a [workgroup][N][12] slab that a second kernel sums in workgroup order: no float atomics,
// bit-reproducible.
#include "gsaj_common.h"
#include "wave_reduce.h"

struct DensePixelMap {  // where pixel (col, row) sits in the coordinates of means2D / covs2D
  int normalised;       // 0: (col, row) themselves (compare.py:1226-1229)
  double fx, fy, cx, cy;  // 1: ((col - cx) / fx, (row - cy) / fy)   (script.py:873-877)
};

#define DCHUNK 128  // Gaussians staged per LDS chunk
#define DPAR 12     // floats per staged Gaussian: mu(2) inv(4) colour(3) depth opacity pad

__device__ __forceinline__ void stage_chunk(float *par, int tid, int base, int n, const float *means2D, const float *covs2D,
                                            const float *colors, const float *depths, const float *opac) {
  if (tid < n) {
    const int i = base + tid;
    const float a = covs2D[4 * i], b = covs2D[4 * i + 1], c = covs2D[4 * i + 2], d = covs2D[4 * i + 3];
    const float det = a * d - b * c;
    float *p = par + tid * DPAR;
    p[0] = means2D[2 * i]; p[1] = means2D[2 * i + 1];
    p[2] = d / det; p[3] = -b / det; p[4] = -c / det; p[5] = a / det;  // inv[0][0], [0][1], [1][0], [1][1]
    p[6] = colors[3 * i]; p[7] = colors[3 * i + 1]; p[8] = colors[3 * i + 2];
    p[9] = depths[i]; p[10] = opac[i]; p[11] = 0.f;
  }
}

__device__ __forceinline__ float dense_alpha(const float *p, float u, float v, float &dx, float &dy, float &qx, float &qy,
                                             float &rx, float &ry) {
  dx = u - p[0]; dy = v - p[1];
  qx = p[2] * dx + p[3] * dy;  // q = S^-1 D       (dalpha/dmu direction, :1324)
  qy = p[4] * dx + p[5] * dy;
  rx = dx * p[2] + dy * p[4];  // r = D^T S^-1     (exponent, :1266-1267)
  ry = dx * p[3] + dy * p[5];
  const float e = -0.5f * (rx * dx + ry * dy);
  return fminf(fmaxf(p[10] * expf(e), 0.0f), 1.0f);
}

__global__ __launch_bounds__(256) void k_dense_bwd(int N, int W, int H, const float *__restrict__ means2D,
                                                   const float *__restrict__ covs2D, const float *__restrict__ colors,
                                                   const float *__restrict__ depths, const float *__restrict__ opac,
                                                   const float *__restrict__ seed_color,
                                                   const float *__restrict__ seed_depth, float *__restrict__ slab, int naive,
                                                   DensePixelMap pm) {
  __shared__ float par[DCHUNK * DPAR];
  __shared__ float acc[DCHUNK * 4 * IGRAD_F];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const size_t HW = (size_t)W * H;
  const size_t pid = (size_t)blockIdx.x * 256 + tid;
  const bool inside = pid < HW;
  float u = inside ? (float)(pid % W) : 0.f, v = inside ? (float)(pid / W) : 0.f;
  if (pm.normalised && inside) {
    // the normalised-coordinate variant (Loss_Derivative_script.py:873-877): x_n = (u - cx) / fx in float64, then float32
    u = (float)(((double)(pid % W) - pm.cx) / pm.fx);
    v = (float)(((double)(pid / W) - pm.cy) / pm.fy);
  }
  float gC[3] = {0.f, 0.f, 0.f}, gD = 0.f;
  if (inside) {
    gC[0] = seed_color[3 * pid]; gC[1] = seed_color[3 * pid + 1]; gC[2] = seed_color[3 * pid + 2];
    gD = seed_depth[pid];
  }
  // ---- pass 1: per-pixel totals sum_i (c_i, z_i) alpha_i T_i ----
  double tot[4] = {0.0, 0.0, 0.0, 0.0};
  {
    float T = 1.0f;
    for (int base = 0; base < N; base += DCHUNK) {
      const int n = min(DCHUNK, N - base);
      __syncthreads();
      stage_chunk(par, tid, base, n, means2D, covs2D, colors, depths, opac);
      __syncthreads();
      for (int j = 0; j < n; j++) {
        const float *p = par + j * DPAR;
        float dx, dy, qx, qy, rx, ry;
        const float alpha = dense_alpha(p, u, v, dx, dy, qx, qy, rx, ry);
        const float aT = alpha * T;
        tot[0] += (double)(p[6] * aT); tot[1] += (double)(p[7] * aT); tot[2] += (double)(p[8] * aT);
        tot[3] += (double)(p[9] * aT);
        T = T * (1.0f - alpha);
      }
    }
  }
  // ---- pass 2: gradients ----
  double pre[4] = {0.0, 0.0, 0.0, 0.0};
  float T = 1.0f;
  for (int base = 0; base < N; base += DCHUNK) {
    const int n = min(DCHUNK, N - base);
    __syncthreads();
    stage_chunk(par, tid, base, n, means2D, covs2D, colors, depths, opac);
    __syncthreads();
    for (int j = 0; j < n; j++) {
      const float *p = par + j * DPAR;
      float dx, dy, qx, qy, rx, ry;
      const float alpha = dense_alpha(p, u, v, dx, dy, qx, qy, rx, ry);
      const float aT = alpha * T;
      const float den = alpha < 0.999f ? 1.0f - alpha : 1.0f;
      // naive-loop semantics (GSAJ_DENSE_NAIVE_GUARDS): at alpha >= 0.999 the suffix term is dropped, not divided by 1
      const float keep = (naive && !(alpha < 0.999f)) ? 0.0f : 1.0f;
      float dLda = 0.f;
#pragma unroll
      for (int ch = 0; ch < 4; ch++) {
        const float val = p[6 + ch];
        pre[ch] += (double)(val * aT);
        const float after = (float)(tot[ch] - pre[ch]);
        const float g = ch < 3 ? gC[ch] : gD;
        dLda += g * (val * T - keep * (after / den));
      }
      // (naive loop: an entry with abs(alpha) < 1e-8 adds nothing to dL/dmu, dL/dSigma)
      const float w = (inside && !(naive && fabsf(alpha) < 1e-8f)) ? dLda * alpha : 0.f;
      const float m = inside ? aT : 0.f;
      float vals[10];
      vals[0] = w * qx; vals[1] = w * qy;                          // dL/dmu
      vals[2] = 0.5f * w * qx * rx; vals[3] = 0.5f * w * qx * ry;  // dL/dSigma [0][0], [0][1]
      vals[4] = 0.5f * w * qy * rx; vals[5] = 0.5f * w * qy * ry;  //           [1][0], [1][1]
      vals[6] = m * gD;                                            // dL/dz
      vals[7] = m * gC[0]; vals[8] = m * gC[1]; vals[9] = m * gC[2];  // dL/dc
      T = T * (1.0f - alpha);
      float x0, x1, x2;
      reduce10(vals, x0, x1, x2);
      store10(acc + (j * 4 + wave) * IGRAD_F, lane, x0, x1, x2);
    }
    __syncthreads();
    if (tid < n) {
      const float *a = acc + tid * 4 * IGRAD_F;
      float *dst = slab + ((size_t)blockIdx.x * N + base + tid) * IGRAD_F;
#pragma unroll
      for (int k = 0; k < 10; k++) dst[k] = (a[k] + a[IGRAD_F + k]) + (a[2 * IGRAD_F + k] + a[3 * IGRAD_F + k]);
    }
  }
}

// Fixed-order sum over workgroup slabs (fp64 accumulation, one lane per (Gaussian, component)).
__global__ __launch_bounds__(256) void k_dense_reduce(int N, int nblk, const float *__restrict__ slab,
                                                      float *__restrict__ grad_mu, float *__restrict__ grad_Sigma,
                                                      float *__restrict__ grad_depth, float *__restrict__ grad_color) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= N * 10) return;
  const int i = t / 10, k = t - i * 10;
  double s = 0.0;
  for (int b = 0; b < nblk; b++) s += (double)slab[((size_t)b * N + i) * IGRAD_F + k];
  const float r = (float)s;
  if (k < 2) grad_mu[2 * i + k] = r;
  else if (k < 6) grad_Sigma[4 * i + (k - 2)] = r;
  else if (k == 6) grad_depth[i] = r;
  else grad_color[3 * i + (k - 7)] = r;
}

__global__ __launch_bounds__(256) void k_dense_render(int N, int W, int H, const float *__restrict__ means2D,
                                                      const float *__restrict__ covs2D, const float *__restrict__ colors,
                                                      const float *__restrict__ depths, const float *__restrict__ opac,
                                                      float *__restrict__ out_color, float *__restrict__ out_depth) {
  __shared__ float par[DCHUNK * DPAR];
  const int tid = threadIdx.x;
  const size_t HW = (size_t)W * H;
  const size_t pid = (size_t)blockIdx.x * 256 + tid;
  const bool inside = pid < HW;
  const float u = inside ? (float)(pid % W) : 0.f, v = inside ? (float)(pid / W) : 0.f;
  float T = 1.0f, C0 = 0.f, C1 = 0.f, C2 = 0.f, D = 0.f;
  for (int base = 0; base < N; base += DCHUNK) {
    const int n = min(DCHUNK, N - base);
    __syncthreads();
    stage_chunk(par, tid, base, n, means2D, covs2D, colors, depths, opac);
    __syncthreads();
    for (int j = 0; j < n; j++) {
      const float *p = par + j * DPAR;
      float dx, dy, qx, qy, rx, ry;
      const float alpha = dense_alpha(p, u, v, dx, dy, qx, qy, rx, ry);
      const float aT = alpha * T;
      C0 += p[6] * aT; C1 += p[7] * aT; C2 += p[8] * aT; D += p[9] * aT;
      T = T * (1.0f - alpha);
    }
  }
  if (inside) {
    out_color[3 * pid] = C0; out_color[3 * pid + 1] = C1; out_color[3 * pid + 2] = C2;
    out_depth[pid] = D;
  }
}

// ---- closed-form pose Jacobians (fp64) ------------------------------------------------------
// Sigma_I = J R S R^T J^T with J = [[1/z,0,-x/z^2],[0,1/z,-y/z^2]] (normalised image coords, no
// clamp / dilation);  d mu_c = [I, -[mu_c]x] d tau,  dR = [d theta]x R  (left perturbation).
__device__ __forceinline__ void mat23x33(const double A[2][3], const double B[3][3], double O[2][3]) {
  for (int r = 0; r < 2; r++)
    for (int c = 0; c < 3; c++) O[r][c] = A[r][0] * B[0][c] + A[r][1] * B[1][c] + A[r][2] * B[2][c];
}

__global__ __launch_bounds__(128) void k_pose_jacobians(int N, const double *__restrict__ T_cw,
                                                        const double *__restrict__ mu_w, const double *__restrict__ cov6,
                                                        double fx, double fy, int W, int H, double *__restrict__ dmu_out,
                                                        double *__restrict__ dcov_out) {
  const int i = blockIdx.x * 128 + threadIdx.x;
  if (i >= N) return;
  double R[3][3], tr[3];
  for (int r = 0; r < 3; r++) {
    for (int c = 0; c < 3; c++) R[r][c] = T_cw[4 * r + c];
    tr[r] = T_cw[4 * r + 3];
  }
  const double mx = mu_w[3 * i], my = mu_w[3 * i + 1], mz = mu_w[3 * i + 2];
  const double x = R[0][0] * mx + R[0][1] * my + R[0][2] * mz + tr[0];
  const double y = R[1][0] * mx + R[1][1] * my + R[1][2] * mz + tr[1];
  const double z = R[2][0] * mx + R[2][1] * my + R[2][2] * mz + tr[2];
  const double iz = 1.0 / z, iz2 = iz * iz, iz3 = iz2 * iz;
  const double J[2][3] = {{iz, 0.0, -x * iz2}, {0.0, iz, -y * iz2}};
  // d mu_c / d tau = [I, -[mu_c]x]
  const double dmc[3][6] = {{1, 0, 0, 0, z, -y}, {0, 1, 0, -z, 0, x}, {0, 0, 1, y, -x, 0}};
  const double sx = 2.0 * fx / W, sy = 2.0 * fy / H;
  for (int k = 0; k < 6; k++) {
    dmu_out[(size_t)i * 12 + k] = sx * (J[0][0] * dmc[0][k] + J[0][2] * dmc[2][k]);
    dmu_out[(size_t)i * 12 + 6 + k] = sy * (J[1][1] * dmc[1][k] + J[1][2] * dmc[2][k]);
  }
  const double *c = cov6 + 6 * (size_t)i;
  const double S[3][3] = {{c[0], c[1], c[2]}, {c[1], c[3], c[4]}, {c[2], c[4], c[5]}};
  double RS[3][3], RSR[3][3];  // RS = R S, RSR = R S R^T
  for (int r = 0; r < 3; r++)
    for (int cc = 0; cc < 3; cc++) RS[r][cc] = R[r][0] * S[0][cc] + R[r][1] * S[1][cc] + R[r][2] * S[2][cc];
  for (int r = 0; r < 3; r++)
    for (int cc = 0; cc < 3; cc++) RSR[r][cc] = RS[r][0] * R[cc][0] + RS[r][1] * R[cc][1] + RS[r][2] * R[cc][2];
  double JA[2][3];  // J (R S R^T)
  mat23x33(J, RSR, JA);
  const double scale[4] = {fx * fx, fx * fy, fy * fx, fy * fy};
  for (int k = 0; k < 6; k++) {
    // dJ = dJ/dx dmc_x + dJ/dy dmc_y + dJ/dz dmc_z
    const double ax = dmc[0][k], ay = dmc[1][k], az = dmc[2][k];
    const double dJ[2][3] = {{-iz2 * az, 0.0, -iz2 * ax + 2.0 * x * iz3 * az}, {0.0, -iz2 * az, -iz2 * ay + 2.0 * y * iz3 * az}};
    double d[2][2];
    for (int r = 0; r < 2; r++)
      for (int cc = 0; cc < 2; cc++) {
        double v = 0.0;
        for (int q = 0; q < 3; q++) v += dJ[r][q] * JA[cc][q] + JA[r][q] * dJ[cc][q];  // dJ A J^T + J A dJ^T (A symmetric)
        d[r][cc] = v;
      }
    if (k >= 3) {
      const int m = k - 3;  // dR = [e_m]x R
      double E[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
      if (m == 0) { E[1][2] = -1; E[2][1] = 1; }
      if (m == 1) { E[0][2] = 1; E[2][0] = -1; }
      if (m == 2) { E[0][1] = -1; E[1][0] = 1; }
      double dR[3][3], B[3][3];
      for (int r = 0; r < 3; r++)
        for (int cc = 0; cc < 3; cc++) dR[r][cc] = E[r][0] * R[0][cc] + E[r][1] * R[1][cc] + E[r][2] * R[2][cc];
      for (int r = 0; r < 3; r++)  // B = dR S R^T + R S dR^T
        for (int cc = 0; cc < 3; cc++) {
          double v = 0.0;
          for (int q = 0; q < 3; q++) v += dR[r][q] * RS[cc][q] + RS[r][q] * dR[cc][q];
          B[r][cc] = v;
        }
      double JB[2][3];
      mat23x33(J, B, JB);
      for (int r = 0; r < 2; r++)
        for (int cc = 0; cc < 2; cc++) d[r][cc] += JB[r][0] * J[cc][0] + JB[r][1] * J[cc][1] + JB[r][2] * J[cc][2];
    }
    dcov_out[(size_t)i * 24 + 0 * 6 + k] = scale[0] * d[0][0];
    dcov_out[(size_t)i * 24 + 1 * 6 + k] = scale[1] * d[0][1];
    dcov_out[(size_t)i * 24 + 2 * 6 + k] = scale[2] * d[1][0];
    dcov_out[(size_t)i * 24 + 3 * 6 + k] = scale[3] * d[1][1];
  }
}

// ---- dL/dtau chain rule over sorted Gaussians (fp64) ------------------------------------------
__device__ __forceinline__ void sh_basis16(int deg, double x, double y, double z, double *B) {
  const double C0 = 0.28209479177387814, C1 = 0.4886025119029199;
  const double C2[5] = {1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396};
  const double C3[7] = {-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154,
                        -0.4570457994644658, 1.445305721320277, -0.5900435899266435};
  for (int k = 0; k < 16; k++) B[k] = 0.0;
  B[0] = C0;
  if (deg > 0) { B[1] = -C1 * y; B[2] = C1 * z; B[3] = -C1 * x; }
  if (deg > 1) {
    const double xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
    B[4] = C2[0] * xy; B[5] = C2[1] * yz; B[6] = C2[2] * (2.0 * zz - xx - yy); B[7] = C2[3] * xz; B[8] = C2[4] * (xx - yy);
    if (deg > 2) {
      B[9] = C3[0] * y * (3 * xx - yy); B[10] = C3[1] * xy * z; B[11] = C3[2] * y * (4 * zz - xx - yy);
      B[12] = C3[3] * z * (2 * zz - 3 * xx - 3 * yy); B[13] = C3[4] * x * (4 * zz - xx - yy);
      B[14] = C3[5] * z * (xx - yy); B[15] = C3[6] * x * (xx - 3 * yy);
    }
  }
}
// d(basis_k)/d(x,y,z)
__device__ __forceinline__ void sh_dbasis16(int deg, double x, double y, double z, double *Bx, double *By, double *Bz) {
  const double C1 = 0.4886025119029199;
  const double C2[5] = {1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396};
  const double C3[7] = {-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154,
                        -0.4570457994644658, 1.445305721320277, -0.5900435899266435};
  for (int k = 0; k < 16; k++) Bx[k] = By[k] = Bz[k] = 0.0;
  if (deg > 0) { By[1] = -C1; Bz[2] = C1; Bx[3] = -C1; }
  if (deg > 1) {
    Bx[4] = C2[0] * y; By[4] = C2[0] * x;
    By[5] = C2[1] * z; Bz[5] = C2[1] * y;
    Bx[6] = -2 * C2[2] * x; By[6] = -2 * C2[2] * y; Bz[6] = 4 * C2[2] * z;
    Bx[7] = C2[3] * z; Bz[7] = C2[3] * x;
    Bx[8] = 2 * C2[4] * x; By[8] = -2 * C2[4] * y;
    if (deg > 2) {
      const double xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
      Bx[9] = 6 * C3[0] * xy; By[9] = 3 * C3[0] * (xx - yy);
      Bx[10] = C3[1] * yz; By[10] = C3[1] * xz; Bz[10] = C3[1] * xy;
      Bx[11] = -2 * C3[2] * xy; By[11] = C3[2] * (-3 * yy + 4 * zz - xx); Bz[11] = 8 * C3[2] * yz;
      Bx[12] = -6 * C3[3] * xz; By[12] = -6 * C3[3] * yz; Bz[12] = 3 * C3[3] * (2 * zz - xx - yy);
      Bx[13] = C3[4] * (-3 * xx + 4 * zz - yy); By[13] = -2 * C3[4] * xy; Bz[13] = 8 * C3[4] * xz;
      Bx[14] = 2 * C3[5] * xz; By[14] = -2 * C3[5] * yz; Bz[14] = C3[5] * (xx - yy);
      Bx[15] = 3 * C3[6] * (xx - yy); By[15] = -6 * C3[6] * xy;
    }
  }
}

// ---- NumPy-path front end: project, colour and depth-order the Gaussians (fp64 on fp32 inputs, like the reference's
// Python floats) -- GetImagePlaneMeanAndCovs + compute_cov2d + ndc2Pix + compute_colors_from_sh + OrderGaussiansByDepth
// (Loss_Derivative_script_compare.py:854-971, :772-848, :851-852, :535-588, :764-769).  A.4 semantics: NO z <= 0.2 cull,
// NO radius / tile test, colours clamped below at 0 only, ONE global stable depth order.
__global__ __launch_bounds__(128) void k_dense_project(int N, int M, int deg, const float *__restrict__ means3D,
                                                       const float *__restrict__ cov3D6, const float *__restrict__ shs,
                                                       const float *__restrict__ vm, const float *__restrict__ pm,
                                                       const float *__restrict__ campos, double fx, double fy, double tanx,
                                                       double tany, int W, int H, double *__restrict__ mean2D,
                                                       double *__restrict__ cov2D, double *__restrict__ color,
                                                       double *__restrict__ color_raw, double *__restrict__ depth) {
  const int i = blockIdx.x * 128 + threadIdx.x;
  if (i >= N) return;
  const double x = means3D[3 * i], y = means3D[3 * i + 1], z = means3D[3 * i + 2];
  // the rasteriser's transposed matrices: flat index 4*c + r = element (r, c) of W2C / P W2C
  double ph[4], t[3];
  for (int r = 0; r < 4; r++) ph[r] = (double)pm[r] * x + (double)pm[4 + r] * y + (double)pm[8 + r] * z + (double)pm[12 + r];
  for (int r = 0; r < 3; r++) t[r] = (double)vm[r] * x + (double)vm[4 + r] * y + (double)vm[8 + r] * z + (double)vm[12 + r];
  const double pw = ph[3] + 0.0000001;
  mean2D[2 * i] = ((ph[0] / pw + 1.0) * W - 1.0) * 0.5;
  mean2D[2 * i + 1] = ((ph[1] / pw + 1.0) * H - 1.0) * 0.5;
  depth[i] = t[2];
  const double limx = 1.3 * tanx, limy = 1.3 * tany;
  const double tx = fmin(limx, fmax(-limx, t[0] / t[2])) * t[2], ty = fmin(limy, fmax(-limy, t[1] / t[2])) * t[2];
  const double J[2][3] = {{fx / t[2], 0.0, -(fx * tx) / (t[2] * t[2])}, {0.0, fy / t[2], -(fy * ty) / (t[2] * t[2])}};
  double T[2][3];  // T = J W, W[r][c] = element (r, c) of the rotation block of W2C
  for (int r = 0; r < 2; r++)
    for (int c = 0; c < 3; c++) T[r][c] = J[r][0] * (double)vm[4 * c + 0] + J[r][1] * (double)vm[4 * c + 1] + J[r][2] * (double)vm[4 * c + 2];
  const float *c6 = cov3D6 + 6 * (size_t)i;
  const double V[3][3] = {{c6[0], c6[1], c6[2]}, {c6[1], c6[3], c6[4]}, {c6[2], c6[4], c6[5]}};
  double TV[2][3];
  for (int r = 0; r < 2; r++)
    for (int c = 0; c < 3; c++) TV[r][c] = T[r][0] * V[0][c] + T[r][1] * V[1][c] + T[r][2] * V[2][c];
  for (int r = 0; r < 2; r++)
    for (int c = 0; c < 2; c++)
      cov2D[4 * i + 2 * r + c] = TV[r][0] * T[c][0] + TV[r][1] * T[c][1] + TV[r][2] * T[c][2] + (r == c ? 0.3 : 0.0);
  // colours: SH (degree `deg`, M coefficients stored) at the world-frame direction (gaussian - camera) / (|.| + 1e-8)
  const double d0 = x - (double)campos[0], d1 = y - (double)campos[1], d2 = z - (double)campos[2];
  const double nrm = sqrt(d0 * d0 + d1 * d1 + d2 * d2) + 1e-8;
  double B[16];
  sh_basis16(deg, d0 / nrm, d1 / nrm, d2 / nrm, B);
  for (int ch = 0; ch < 3; ch++) {
    double raw = 0.5;
    for (int k = 0; k < M && k < 16; k++) raw += B[k] * (double)shs[((size_t)i * M + k) * 3 + ch];
    color_raw[3 * i + ch] = raw;
    color[3 * i + ch] = raw < 0.0 ? 0.0 : raw;
  }
}

// rank of every Gaussian in the global stable depth order (Python's list.sort is stable: ties keep index order).
// N is small on this path (every Gaussian meets every pixel), so the O(N^2) count with an LDS-staged key block is ample.
__global__ __launch_bounds__(256) void k_dense_rank(int N, const double *__restrict__ depth, int *__restrict__ order) {
  __shared__ double zs[256];
  const int i = blockIdx.x * 256 + threadIdx.x;
  const double zi = i < N ? depth[i] : 0.0;
  int rank = 0;
  for (int base = 0; base < N; base += 256) {
    __syncthreads();
    if (base + threadIdx.x < N) zs[threadIdx.x] = depth[base + threadIdx.x];
    __syncthreads();
    const int n = min(256, N - base);
    for (int j = 0; j < n; j++) {
      const double zj = zs[j];
      rank += (zj < zi || (zj == zi && base + j < i)) ? 1 : 0;
    }
  }
  if (i < N) order[rank] = i;
}

__global__ __launch_bounds__(256) void k_dense_gather(int N, const int *__restrict__ order, const double *__restrict__ in,
                                                      double *__restrict__ out, int width) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= N * width) return;
  const int r = t / width, c = t - r * width;
  out[t] = in[(size_t)order[r] * width + c];
}

__global__ __launch_bounds__(256) void k_dense_tau(int N, int M, int deg, const int *__restrict__ order,
                                                   const float *__restrict__ g_mu, const float *__restrict__ g_S,
                                                   const float *__restrict__ g_z, const float *__restrict__ g_c,
                                                   const double *__restrict__ dmu, const double *__restrict__ dcov,
                                                   const double *__restrict__ mu_w, const double *__restrict__ T_cw,
                                                   const double *__restrict__ campos, const double *__restrict__ shs,
                                                   double *__restrict__ out, double *__restrict__ parts) {
  __shared__ double sh[256];
  const int tid = threadIdx.x;
  double acc[4][6];
  for (int a = 0; a < 4; a++)
    for (int k = 0; k < 6; k++) acc[a][k] = 0.0;
  for (int i = tid; i < N; i += 256) {
    const int idx = order[i];
    const double *A = dmu + (size_t)idx * 12, *Bc = dcov + (size_t)idx * 24;
    const double mx = mu_w[3 * idx], my = mu_w[3 * idx + 1], mz = mu_w[3 * idx + 2];
    const double xc = T_cw[0] * mx + T_cw[1] * my + T_cw[2] * mz + T_cw[3];
    const double yc = T_cw[4] * mx + T_cw[5] * my + T_cw[6] * mz + T_cw[7];
    for (int k = 0; k < 6; k++) {
      acc[0][k] += (double)g_mu[2 * i] * A[k] + (double)g_mu[2 * i + 1] * A[6 + k];
      acc[1][k] += (double)g_S[4 * i] * Bc[k] + (double)g_S[4 * i + 1] * Bc[6 + k] + (double)g_S[4 * i + 2] * Bc[12 + k] +
                   (double)g_S[4 * i + 3] * Bc[18 + k];
    }
    const double gz = (double)g_z[i];
    acc[2][2] += gz; acc[2][3] += gz * yc; acc[2][4] += -gz * xc;
    if (shs && M > 0) {
      const double d0 = mx - campos[0], d1 = my - campos[1], d2 = mz - campos[2];
      const double nrm = sqrt(d0 * d0 + d1 * d1 + d2 * d2);
      const double x = d0 / (nrm + 1e-8), y = d1 / (nrm + 1e-8), z = d2 / (nrm + 1e-8);
      double B[16], Bx[16], By[16], Bz[16];
      sh_basis16(deg, x, y, z, B);
      sh_dbasis16(deg, x, y, z, Bx, By, Bz);
      const double *s = shs + (size_t)idx * M * 3;
      double ddir[3] = {0.0, 0.0, 0.0};
      for (int ch = 0; ch < 3; ch++) {
        double raw = 0.5, ex = 0.0, ey = 0.0, ez = 0.0;
        for (int k = 0; k < M && k < 16; k++) {
          const double cf = s[k * 3 + ch];
          raw += B[k] * cf; ex += Bx[k] * cf; ey += By[k] * cf; ez += Bz[k] * cf;
        }
        const double g = raw < 0.0 ? 0.0 : (double)g_c[3 * i + ch];
        ddir[0] += ex * g; ddir[1] += ey * g; ddir[2] += ez * g;
      }
      if (nrm >= 1e-8) {  // dnormvdv: (g - (g.vh) vh) / |v|
        const double vh0 = d0 / nrm, vh1 = d1 / nrm, vh2 = d2 / nrm;
        const double dot = ddir[0] * vh0 + ddir[1] * vh1 + ddir[2] * vh2;
        acc[3][0] -= (ddir[0] - dot * vh0) / nrm;
        acc[3][1] -= (ddir[1] - dot * vh1) / nrm;
        acc[3][2] -= (ddir[2] - dot * vh2) / nrm;
      }
    }
  }
  for (int a = 0; a < 4; a++)
    for (int k = 0; k < 6; k++) {
      sh[tid] = acc[a][k];
      __syncthreads();
      for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) sh[tid] += sh[tid + o];
        __syncthreads();
      }
      if (tid == 0) {
        if (parts) parts[a * 6 + k] = sh[0];
        acc[a][k] = sh[0];
      }
      __syncthreads();
    }
  if (tid == 0)
    for (int k = 0; k < 6; k++) out[k] = acc[0][k] + acc[1][k] + acc[2][k] + acc[3][k];
}

extern "C" {

size_t gsaj_dense_workspace_bytes(int N, int W, int H) {
  const size_t nblk = ((size_t)W * H + 255) / 256;
  return nblk * (size_t)(N > 0 ? N : 1) * IGRAD_F * sizeof(float) + 256;
}

int gsaj_dense_backward(int N, int W, int H, const float *means2D, const float *covs2D, const float *colors,
                        const float *depths, const float *opac, const float *seed_color, const float *seed_depth,
                        float *grad_mu, float *grad_Sigma, float *grad_depth, float *grad_color, void *dense_ws,
                        int flags, const double *intrinsics, void *stream) {
  if (N <= 0 || W <= 0 || H <= 0 || !means2D || !covs2D || !colors || !depths || !opac || !seed_color || !seed_depth ||
      !grad_mu || !grad_Sigma || !grad_depth || !grad_color || !dense_ws) {
    gsaj_set_error("gsaj_dense_backward: invalid argument");
    return GSAJ_ERR_INVALID_ARGUMENT;
  }
  DensePixelMap pm = {0, 1.0, 1.0, 0.0, 0.0};
  if (flags & GSAJ_DENSE_NORMALISED_COORDS) {
    if (!intrinsics || !(intrinsics[0] != 0.0) || !(intrinsics[1] != 0.0)) {
      gsaj_set_error("gsaj_dense_backward: GSAJ_DENSE_NORMALISED_COORDS needs intrinsics = {fx, fy, cx, cy} with fx, fy != 0");
      return GSAJ_ERR_INVALID_ARGUMENT;
    }
    pm = DensePixelMap{1, intrinsics[0], intrinsics[1], intrinsics[2], intrinsics[3]};
  }
  hipStream_t s = (hipStream_t)stream;
  const int nblk = (int)(((size_t)W * H + 255) / 256);
  float *slab = reinterpret_cast<float *>(gsaj_align(reinterpret_cast<size_t>(dense_ws)));
  {
    GsajProfScope ps(ST_DENSE_BWD, s);
    hipLaunchKernelGGL(k_dense_bwd, dim3(nblk), dim3(256), 0, s, N, W, H, means2D, covs2D, colors, depths, opac, seed_color,
                       seed_depth, slab, (flags & GSAJ_DENSE_NAIVE_GUARDS) ? 1 : 0, pm);
  }
  GsajProfScope ps(ST_DENSE_REDUCE, s);
  hipLaunchKernelGGL(k_dense_reduce, dim3((N * 10 + 255) / 256), dim3(256), 0, s, N, nblk, slab, grad_mu, grad_Sigma,
                     grad_depth, grad_color);
  GSAJ_HIP_CHECK(hipGetLastError());
  return GSAJ_OK;
}

int gsaj_dense_render(int N, int W, int H, const float *means2D, const float *covs2D, const float *colors,
                      const float *depths, const float *opac, float *out_color, float *out_depth, void *stream) {
  if (N <= 0 || W <= 0 || H <= 0 || !means2D || !covs2D || !colors || !depths || !opac || !out_color || !out_depth) {
    gsaj_set_error("gsaj_dense_render: invalid argument");
    return GSAJ_ERR_INVALID_ARGUMENT;
  }
  const int nblk = (int)(((size_t)W * H + 255) / 256);
  hipLaunchKernelGGL(k_dense_render, dim3(nblk), dim3(256), 0, (hipStream_t)stream, N, W, H, means2D, covs2D, colors,
                     depths, opac, out_color, out_depth);
  GSAJ_HIP_CHECK(hipGetLastError());
  return GSAJ_OK;
}

size_t gsaj_dense_project_workspace_bytes(int N) { return sizeof(double) * 13 * (size_t)(N > 0 ? N : 1) + 256; }

int gsaj_dense_project(int N, int sh_coeffs, int sh_degree, int W, int H, const float *means3D, const float *cov3D,
                       const float *shs, const float *viewmatrix, const float *projmatrix, const float *campos, double fx,
                       double fy, int *order, double *mean2D, double *cov2D, double *color, double *color_raw, double *depth,
                       void *project_ws, void *stream) {
  if (N <= 0 || W <= 0 || H <= 0 || !means3D || !cov3D || !shs || sh_coeffs <= 0 || sh_coeffs > 16 || sh_degree < 0 ||
      (sh_degree + 1) * (sh_degree + 1) > sh_coeffs || !viewmatrix || !projmatrix || !campos || !order || !mean2D || !cov2D ||
      !color || !depth || !project_ws) {
    gsaj_set_error("gsaj_dense_project: invalid argument");
    return GSAJ_ERR_INVALID_ARGUMENT;
  }
  hipStream_t s = (hipStream_t)stream;
  double *w = reinterpret_cast<double *>(gsaj_align(reinterpret_cast<size_t>(project_ws)));
  const size_t n = (size_t)N;
  double *u_mean = w, *u_cov = w + 2 * n, *u_col = w + 6 * n, *u_raw = w + 9 * n, *u_dep = w + 12 * n;
  hipLaunchKernelGGL(k_dense_project, dim3((N + 127) / 128), dim3(128), 0, s, N, sh_coeffs, sh_degree, means3D, cov3D, shs,
                     viewmatrix, projmatrix, campos, fx, fy, W / (2.0 * fx), H / (2.0 * fy), W, H, u_mean, u_cov, u_col, u_raw,
                     u_dep);
  hipLaunchKernelGGL(k_dense_rank, dim3((N + 255) / 256), dim3(256), 0, s, N, u_dep, order);
  const struct { const double *in; double *out; int width; } cols[5] = {
      {u_mean, mean2D, 2}, {u_cov, cov2D, 4}, {u_col, color, 3}, {u_raw, color_raw, 3}, {u_dep, depth, 1}};
  for (int k = 0; k < 5; k++)
    if (cols[k].out)
      hipLaunchKernelGGL(k_dense_gather, dim3((N * cols[k].width + 255) / 256), dim3(256), 0, s, N, order, cols[k].in, cols[k].out,
                         cols[k].width);
  GSAJ_HIP_CHECK(hipGetLastError());
  return GSAJ_OK;
}

int gsaj_pose_jacobians(int N, const double *T_cw, const double *mu_w, const double *cov3D, double fx, double fy, int W,
                        int H, double *dmu_dtau, double *dcov_dtau, void *stream) {
  if (N <= 0 || !T_cw || !mu_w || !cov3D || !dmu_dtau || !dcov_dtau) {
    gsaj_set_error("gsaj_pose_jacobians: invalid argument");
    return GSAJ_ERR_INVALID_ARGUMENT;
  }
  hipLaunchKernelGGL(k_pose_jacobians, dim3((N + 127) / 128), dim3(128), 0, (hipStream_t)stream, N, T_cw, mu_w, cov3D, fx,
                     fy, W, H, dmu_dtau, dcov_dtau);
  GSAJ_HIP_CHECK(hipGetLastError());
  return GSAJ_OK;
}

int gsaj_dense_tau(int N, int sh_coeffs, int sh_degree, const int *order, const float *grad_mu, const float *grad_Sigma,
                   const float *grad_depth, const float *grad_color, const double *dmu_dtau, const double *dcov_dtau,
                   const double *mu_w, const double *T_cw, const double *campos, const double *shs, double *dL_dtau,
                   double *parts, void *stream) {
  if (N <= 0 || !order || !grad_mu || !grad_Sigma || !grad_depth || !grad_color || !dmu_dtau || !dcov_dtau || !mu_w ||
      !T_cw || !dL_dtau || (shs && !campos)) {
    gsaj_set_error("gsaj_dense_tau: invalid argument");
    return GSAJ_ERR_INVALID_ARGUMENT;
  }
  hipLaunchKernelGGL(k_dense_tau, dim3(1), dim3(256), 0, (hipStream_t)stream, N, sh_coeffs, sh_degree, order, grad_mu,
                     grad_Sigma, grad_depth, grad_color, dmu_dtau, dcov_dtau, mu_w, T_cw, campos, shs, dL_dtau, parts);
  GSAJ_HIP_CHECK(hipGetLastError());
  return GSAJ_OK;
}

}  // extern "C"
