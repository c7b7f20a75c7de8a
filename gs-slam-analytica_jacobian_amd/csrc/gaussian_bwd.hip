// gaussian_bwd.hip -- fused per-Gaussian backward (gfx950): instance-gradient gather,
// conic -> cov2D -> (cov3D, mean3D, tau), mean2D -> (mean3D, tau), depth -> (mean3D, tau),
// colour -> (SH, mean3D, tau), cov3D -> (scale, rotation), and the reduction of dL/dtau.
//
// The reference runs computeCov2DCUDA (backward.cu:150-422), preprocessCUDA (:494-624, with
// computeColorFromSH :21-145 and computeCov3D :426-489) as two kernels that `+=` into
// dL_dmean3D / dL_dtau through global memory, then torch.sum over [P,6]
// (diff_gaussian_rasterization/__init__.py:162).  Here one lane owns one Gaussian end to end:
// it first sums the partial gradients of its (tile, Gaussian) instances in emission order
// (deterministic -- replaces the float atomics of backward.cu:852-869), keeps every
// intermediate in registers, writes each output once, and the 6 pose components are reduced
// wave -> (last-arriving workgroup) fixed-order final pass.  HBM-bound streaming stage: one wave per
// workgroup (P/64 groups spread over the 256 CUs), the instance partials of a Gaussian are one
// contiguous run (emission-slot order), and the [64][M*3] SH blocks are staged through LDS so that the
// global loads / stores of dL_dsh are fully coalesced (padded LDS rows, conflict-free per-lane reads).
#include "gsaj_common.h"

__constant__ float bSH_C0 = 0.28209479177387814f;
__constant__ float bSH_C1 = 0.4886025119029199f;
__constant__ float bSH_C2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f, -1.0925484305920792f,
                                0.5462742152960396f};
__constant__ float bSH_C3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f, 0.3731763325901154f,
                                -0.4570457994644658f, 1.445305721320277f, -0.5900435899266435f};

__device__ __forceinline__ float3 dnormvdv(float3 v, float3 dv) {
  const float sum2 = v.x * v.x + v.y * v.y + v.z * v.z;
  const float inv = 1.0f / sqrtf(sum2 * sum2 * sum2);
  float3 o;
  o.x = ((+sum2 - v.x * v.x) * dv.x - v.y * v.x * dv.y - v.z * v.x * dv.z) * inv;
  o.y = (-v.x * v.y * dv.x + (sum2 - v.y * v.y) * dv.y - v.z * v.y * dv.z) * inv;
  o.z = (-v.x * v.z * dv.x - v.y * v.z * dv.y + (sum2 - v.z * v.z) * dv.z) * inv;
  return o;
}

// SH backward.  `sh` holds this Gaussian's coefficients [M][3]; dL/dsh goes to `dL_dsh` (zeros above the active degree).
// dL_dsh may alias sh (the single-view kernel works in place): within a band every coefficient is read before the band's
// gradients overwrite it.  Returns dL/dmean through the view direction.
// MODE 0: dL_dsh[k][ch] = w_k g[ch] (w_k = the basis weights of this view direction, g = the colour gradient masked by the clamp
// flags); MODE 1: dL_dsh[k] = w_k only (the batched path sums w_k g[ch] over views itself; nothing is written above the active
// degree); MODE 2: dL_dsh[k][ch] += w_k g[ch].
template <int MODE>
__device__ __forceinline__ float3 sh_backward(int deg, int M, float3 pos, float3 campos, const float *sh, float *dL_dsh,
                                              const uint8_t *__restrict__ clamped, float3 gcol) {
  const float3 dorig = make_float3(pos.x - campos.x, pos.y - campos.y, pos.z - campos.z);
  const float len = sqrtf(dorig.x * dorig.x + dorig.y * dorig.y + dorig.z * dorig.z);
  const float x = dorig.x / len, y = dorig.y / len, z = dorig.z / len;
  const float g[3] = {gcol.x * (clamped[0] ? 0.f : 1.f), gcol.y * (clamped[1] ? 0.f : 1.f), gcol.z * (clamped[2] ? 0.f : 1.f)};
  float dx[3] = {0.f, 0.f, 0.f}, dy[3] = {0.f, 0.f, 0.f}, dz[3] = {0.f, 0.f, 0.f};
#define SH(k, ch) sh[(k) * 3 + (ch)]
#define OUT(k, w)                                                                  \
  {                                                                                \
    const float _w = (w);                                                          \
    if (MODE == 1) {                                                               \
      dL_dsh[(k)] = _w;                                                            \
    } else if (MODE == 2) {                                                        \
      dL_dsh[(k) * 3 + 0] += _w * g[0];                                            \
      dL_dsh[(k) * 3 + 1] += _w * g[1];                                            \
      dL_dsh[(k) * 3 + 2] += _w * g[2];                                            \
    } else {                                                                       \
      dL_dsh[(k) * 3 + 0] = _w * g[0];                                             \
      dL_dsh[(k) * 3 + 1] = _w * g[1];                                             \
      dL_dsh[(k) * 3 + 2] = _w * g[2];                                             \
    }                                                                              \
  }
  OUT(0, bSH_C0)
  if (deg > 0) {
#pragma unroll
    for (int ch = 0; ch < 3; ch++) {
      dx[ch] = -bSH_C1 * SH(3, ch);
      dy[ch] = -bSH_C1 * SH(1, ch);
      dz[ch] = bSH_C1 * SH(2, ch);
    }
    OUT(1, -bSH_C1 * y) OUT(2, bSH_C1 * z) OUT(3, -bSH_C1 * x)
    if (deg > 1) {
      const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
#pragma unroll
      for (int ch = 0; ch < 3; ch++) {
        dx[ch] += bSH_C2[0] * y * SH(4, ch) + bSH_C2[2] * 2.f * -x * SH(6, ch) + bSH_C2[3] * z * SH(7, ch) + bSH_C2[4] * 2.f * x * SH(8, ch);
        dy[ch] += bSH_C2[0] * x * SH(4, ch) + bSH_C2[1] * z * SH(5, ch) + bSH_C2[2] * 2.f * -y * SH(6, ch) + bSH_C2[4] * 2.f * -y * SH(8, ch);
        dz[ch] += bSH_C2[1] * y * SH(5, ch) + bSH_C2[2] * 2.f * 2.f * z * SH(6, ch) + bSH_C2[3] * x * SH(7, ch);
      }
      OUT(4, bSH_C2[0] * xy) OUT(5, bSH_C2[1] * yz) OUT(6, bSH_C2[2] * (2.f * zz - xx - yy))
      OUT(7, bSH_C2[3] * xz) OUT(8, bSH_C2[4] * (xx - yy))
      if (deg > 2) {
#pragma unroll
        for (int ch = 0; ch < 3; ch++) {
          dx[ch] += (bSH_C3[0] * SH(9, ch) * 3.f * 2.f * xy + bSH_C3[1] * SH(10, ch) * yz + bSH_C3[2] * SH(11, ch) * -2.f * xy +
                     bSH_C3[3] * SH(12, ch) * -3.f * 2.f * xz + bSH_C3[4] * SH(13, ch) * (-3.f * xx + 4.f * zz - yy) +
                     bSH_C3[5] * SH(14, ch) * 2.f * xz + bSH_C3[6] * SH(15, ch) * 3.f * (xx - yy));
          dy[ch] += (bSH_C3[0] * SH(9, ch) * 3.f * (xx - yy) + bSH_C3[1] * SH(10, ch) * xz +
                     bSH_C3[2] * SH(11, ch) * (-3.f * yy + 4.f * zz - xx) + bSH_C3[3] * SH(12, ch) * -3.f * 2.f * yz +
                     bSH_C3[4] * SH(13, ch) * -2.f * xy + bSH_C3[5] * SH(14, ch) * -2.f * yz + bSH_C3[6] * SH(15, ch) * -3.f * 2.f * xy);
          dz[ch] += (bSH_C3[1] * SH(10, ch) * xy + bSH_C3[2] * SH(11, ch) * 4.f * 2.f * yz +
                     bSH_C3[3] * SH(12, ch) * 3.f * (2.f * zz - xx - yy) + bSH_C3[4] * SH(13, ch) * 4.f * 2.f * xz +
                     bSH_C3[5] * SH(14, ch) * (xx - yy));
        }
        OUT(9, bSH_C3[0] * y * (3.f * xx - yy)) OUT(10, bSH_C3[1] * xy * z) OUT(11, bSH_C3[2] * y * (4.f * zz - xx - yy))
        OUT(12, bSH_C3[3] * z * (2.f * zz - 3.f * xx - 3.f * yy)) OUT(13, bSH_C3[4] * x * (4.f * zz - xx - yy))
        OUT(14, bSH_C3[5] * z * (xx - yy)) OUT(15, bSH_C3[6] * x * (xx - 3.f * yy))
      }
    }
  }
#undef SH
#undef OUT
  if (MODE == 0)
    for (int k = (deg + 1) * (deg + 1) * 3; k < 3 * M; k++) dL_dsh[k] = 0.f;  // coefficients above the active degree
  const float3 ddir = make_float3(dx[0] * g[0] + dx[1] * g[1] + dx[2] * g[2], dy[0] * g[0] + dy[1] * g[1] + dy[2] * g[2],
                                  dz[0] * g[0] + dz[1] * g[1] + dz[2] * g[2]);
  return dnormvdv(dorig, ddir);
}

// dL/dSigma (6-vector, off-diagonals doubled) -> dL/dscale, dL/drot  (backward.cu:426-489): Sigma = R S^2 R^T.
__device__ __forceinline__ void cov3d_backward(const float (&gcov)[6], float3 sc, float4 q, float scale_modifier, float3 &o_scale,
                                               float4 &o_rot) {
  const float r = q.x, x = q.y, y = q.z, z = q.w;
  const float R[3][3] = {{1.f - 2.f * (y * y + z * z), 2.f * (x * y - r * z), 2.f * (x * z + r * y)},
                         {2.f * (x * y + r * z), 1.f - 2.f * (x * x + z * z), 2.f * (y * z - r * x)},
                         {2.f * (x * z - r * y), 2.f * (y * z + r * x), 1.f - 2.f * (x * x + y * y)}};
  const float s[3] = {scale_modifier * sc.x, scale_modifier * sc.y, scale_modifier * sc.z};
  const float dS[3][3] = {{gcov[0], 0.5f * gcov[1], 0.5f * gcov[2]},
                          {0.5f * gcov[1], gcov[3], 0.5f * gcov[4]},
                          {0.5f * gcov[2], 0.5f * gcov[4], gcov[5]}};
  float dA[3][3];  // A = S R^T, dL/dA = 2 A dSigma
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int cc = 0; cc < 3; cc++)
      dA[i][cc] = 2.0f * ((s[i] * R[0][i]) * dS[0][cc] + (s[i] * R[1][i]) * dS[1][cc] + (s[i] * R[2][i]) * dS[2][cc]);
  o_scale.x = R[0][0] * dA[0][0] + R[1][0] * dA[0][1] + R[2][0] * dA[0][2];
  o_scale.y = R[0][1] * dA[1][0] + R[1][1] * dA[1][1] + R[2][1] * dA[1][2];
  o_scale.z = R[0][2] * dA[2][0] + R[1][2] * dA[2][1] + R[2][2] * dA[2][2];
  float gR[3][3];  // dL/dR[j][i] = s_i dA[i][j]
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) gR[j][i] = s[i] * dA[i][j];
  float4 dq;
  dq.x = 2 * z * (gR[1][0] - gR[0][1]) + 2 * y * (gR[0][2] - gR[2][0]) + 2 * x * (gR[2][1] - gR[1][2]);
  dq.y = 2 * y * (gR[0][1] + gR[1][0]) + 2 * z * (gR[0][2] + gR[2][0]) + 2 * r * (gR[2][1] - gR[1][2]) - 4 * x * (gR[2][2] + gR[1][1]);
  dq.z = 2 * x * (gR[0][1] + gR[1][0]) + 2 * r * (gR[0][2] - gR[2][0]) + 2 * z * (gR[2][1] + gR[1][2]) - 4 * y * (gR[2][2] + gR[0][0]);
  dq.w = 2 * r * (gR[1][0] - gR[0][1]) + 2 * x * (gR[0][2] + gR[2][0]) + 2 * y * (gR[2][1] + gR[1][2]) - 4 * z * (gR[1][1] + gR[0][0]);
  o_rot = dq;
}

// Everything one Gaussian contributes for ONE view, from the reverse compositor's sums (s0, s1, s2 = the 10 partials) to
// dL/d{mean3D, cov3D, scale, rotation, SH} and the 6 pose components -- shared by the single-view and the batched kernel so
// that both produce the same bits per view.  p.viewmatrix / projmatrix / campos are that view's.
struct GaussianGrads {
  float m2x, m2y, ca, cb, cc, op, dz;
  float3 col, gm, scale;
  float4 rot;
  float cov[6];
};
// gaussian_chain_geom: everything but the colour -> SH / view-direction part and cov3D -> (scale, rotation): o.gm and tau
// lack the view-direction term (sh_backward's return value: gm += d, tau[0..2] -= d), o.scale / o.rot are not set.
__device__ __forceinline__ void gaussian_chain_geom(const BwdParams &p, float3 mean, const float (&c6)[6], float4 s0, float4 s1, float4 s2,
                                                    GaussianGrads &o, float (&tau)[6]) {
  const float g2x = s0.x, g2y = s0.y;           // dL/dmean2D (NDC-scaled)
  const float gcx = s0.z, gcy = s0.w, gcz = s1.x;  // dL/dconic a, b, c
  const float gop = s1.y;
  const float3 gcol = make_float3(s1.z, s1.w, s2.x);
  const float gz = s2.y;
  // ---- 2. conic -> cov2D -> cov3D, M = J Rcw, t ----
  const float *vm = p.viewmatrix;
  const float fx = p.focal_x, fy = p.focal_y;
  float3 t = xform4x3(vm, mean);
  const float3 pC = t;  // un-clamped camera-space point
  const float limx = 1.3f * p.tanfovx, limy = 1.3f * p.tanfovy;
  const float txtz = t.x / t.z, tytz = t.y / t.z;
  t.x = fminf(limx, fmaxf(-limx, txtz)) * t.z;
  t.y = fminf(limy, fmaxf(-limy, tytz)) * t.z;
  const float xmul = (txtz < -limx || txtz > limx) ? 0.f : 1.f;
  const float ymul = (tytz < -limy || tytz > limy) ? 0.f : 1.f;
  const float J00 = fx / t.z, J02 = -(fx * t.x) / (t.z * t.z);
  const float J11 = fy / t.z, J12 = -(fy * t.y) / (t.z * t.z);
  float Rc[3][3], M[2][3];
#pragma unroll
  for (int r = 0; r < 3; r++)
#pragma unroll
    for (int c = 0; c < 3; c++) Rc[r][c] = vm[4 * c + r];
#pragma unroll
  for (int k = 0; k < 3; k++) {
    M[0][k] = Rc[0][k] * J00 + Rc[2][k] * J02;
    M[1][k] = Rc[1][k] * J11 + Rc[2][k] * J12;
  }
  const float V[3][3] = {{c6[0], c6[1], c6[2]}, {c6[1], c6[3], c6[4]}, {c6[2], c6[4], c6[5]}};
  float MV[2][3];
#pragma unroll
  for (int r = 0; r < 2; r++)
#pragma unroll
    for (int k = 0; k < 3; k++) MV[r][k] = M[r][0] * V[k][0] + M[r][1] * V[k][1] + M[r][2] * V[k][2];
  const float a = (MV[0][0] * M[0][0] + MV[0][1] * M[0][1] + MV[0][2] * M[0][2]) + 0.3f;
  const float b = MV[1][0] * M[0][0] + MV[1][1] * M[0][1] + MV[1][2] * M[0][2];
  const float c = (MV[1][0] * M[1][0] + MV[1][1] * M[1][1] + MV[1][2] * M[1][2]) + 0.3f;
  const float denom = a * c - b * b;
  float dL_da = 0.f, dL_db = 0.f, dL_dc = 0.f;
  const float denom2inv = 1.0f / ((denom * denom) + 0.0000001f);
  float gcov[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (denom2inv != 0.f) {
    dL_da = denom2inv * (-c * c * gcx + 2 * b * c * gcy + (denom - a * c) * gcz);
    dL_dc = denom2inv * (-a * a * gcz + 2 * a * b * gcy + (denom - a * c) * gcx);
    dL_db = denom2inv * 2 * (b * c * gcx - (denom + 2 * b * b) * gcy + a * b * gcz);
    gcov[0] = (M[0][0] * M[0][0] * dL_da + M[0][0] * M[1][0] * dL_db + M[1][0] * M[1][0] * dL_dc);
    gcov[3] = (M[0][1] * M[0][1] * dL_da + M[0][1] * M[1][1] * dL_db + M[1][1] * M[1][1] * dL_dc);
    gcov[5] = (M[0][2] * M[0][2] * dL_da + M[0][2] * M[1][2] * dL_db + M[1][2] * M[1][2] * dL_dc);
    gcov[1] = 2 * M[0][0] * M[0][1] * dL_da + (M[0][0] * M[1][1] + M[0][1] * M[1][0]) * dL_db + 2 * M[1][0] * M[1][1] * dL_dc;
    gcov[2] = 2 * M[0][0] * M[0][2] * dL_da + (M[0][0] * M[1][2] + M[0][2] * M[1][0]) * dL_db + 2 * M[1][0] * M[1][2] * dL_dc;
    gcov[4] = 2 * M[0][2] * M[0][1] * dL_da + (M[0][1] * M[1][2] + M[0][2] * M[1][1]) * dL_db + 2 * M[1][1] * M[1][2] * dL_dc;
  }
#pragma unroll
  for (int k = 0; k < 6; k++) o.cov[k] = gcov[k];
  float dM[2][3];
#pragma unroll
  for (int k = 0; k < 3; k++) {
    dM[0][k] = 2 * MV[0][k] * dL_da + MV[1][k] * dL_db;
    dM[1][k] = 2 * MV[1][k] * dL_dc + MV[0][k] * dL_db;
  }
  const float dJ00 = Rc[0][0] * dM[0][0] + Rc[0][1] * dM[0][1] + Rc[0][2] * dM[0][2];
  const float dJ02 = Rc[2][0] * dM[0][0] + Rc[2][1] * dM[0][1] + Rc[2][2] * dM[0][2];
  const float dJ11 = Rc[1][0] * dM[1][0] + Rc[1][1] * dM[1][1] + Rc[1][2] * dM[1][2];
  const float dJ12 = Rc[2][0] * dM[1][0] + Rc[2][1] * dM[1][1] + Rc[2][2] * dM[1][2];
  const float tz = 1.f / t.z, tz2 = tz * tz, tz3 = tz2 * tz;
  float3 gt;
  gt.x = xmul * -fx * tz2 * dJ02;
  gt.y = ymul * -fy * tz2 * dJ12;
  gt.z = -fx * tz2 * dJ00 - fy * tz2 * dJ11 + (2 * fx * t.x) * tz3 * dJ02 + (2 * fy * t.y) * tz3 * dJ12;
  // tau: rho += g, theta += t x g (clamped t) + sum_k col_k(Rcw) x dL/dcol_k(Rcw)
  const float3 txg = cross3(t, gt);
  tau[0] += gt.x; tau[1] += gt.y; tau[2] += gt.z;
  tau[3] += txg.x; tau[4] += txg.y; tau[5] += txg.z;
  float3 gm = make_float3(Rc[0][0] * gt.x + Rc[1][0] * gt.y + Rc[2][0] * gt.z,
                          Rc[0][1] * gt.x + Rc[1][1] * gt.y + Rc[2][1] * gt.z,
                          Rc[0][2] * gt.x + Rc[1][2] * gt.y + Rc[2][2] * gt.z);
  {
    float3 th = make_float3(0.f, 0.f, 0.f);
#pragma unroll
    for (int k = 0; k < 3; k++) {
      const float3 ck = make_float3(Rc[0][k], Rc[1][k], Rc[2][k]);
      const float3 gk = make_float3(J00 * dM[0][k], J11 * dM[1][k], J02 * dM[0][k] + J12 * dM[1][k]);
      const float3 cr = cross3(ck, gk);
      th.x += cr.x; th.y += cr.y; th.z += cr.z;
    }
    tau[3] += th.x; tau[4] += th.y; tau[5] += th.z;
  }

  // ---- 3. mean2D -> mean3D, tau ----
  const float *pj = p.projmatrix;
  const float4 mh = xform4x4(pj, mean);
  const float mw = 1.0f / (mh.w + 0.0000001f);
  const float mul1 = (pj[0] * mean.x + pj[4] * mean.y + pj[8] * mean.z + pj[12]) * mw * mw;
  const float mul2 = (pj[1] * mean.x + pj[5] * mean.y + pj[9] * mean.z + pj[13]) * mw * mw;
  gm.x += (pj[0] * mw - pj[3] * mul1) * g2x + (pj[1] * mw - pj[3] * mul2) * g2y;
  gm.y += (pj[4] * mw - pj[7] * mul1) * g2x + (pj[5] * mw - pj[7] * mul2) * g2y;
  gm.z += (pj[8] * mw - pj[11] * mul1) * g2x + (pj[9] * mw - pj[11] * mul2) * g2y;
  {
    const float alpha_ = 1.0f * mw, beta_ = -mh.x * mw * mw, gamma_ = -mh.y * mw * mw;
    const float pa = p.projmatrix_raw[0], pb = p.projmatrix_raw[5], pe = p.projmatrix_raw[11];
    const float3 d1 = make_float3(alpha_ * pa, 0.f, beta_ * pe), d2 = make_float3(0.f, alpha_ * pb, gamma_ * pe);
    const float3 c1 = cross3(pC, d1), c2 = cross3(pC, d2);
    tau[0] += g2x * d1.x + g2y * d2.x; tau[1] += g2x * d1.y + g2y * d2.y; tau[2] += g2x * d1.z + g2y * d2.z;
    tau[3] += g2x * c1.x + g2y * c2.x; tau[4] += g2x * c1.y + g2y * c2.y; tau[5] += g2x * c1.z + g2y * c2.z;
  }
  // ---- 4. depth -> mean3D, tau: dz/dtau = [0,0,1, y, -x, 0] ----
  gm.x += gz * vm[2]; gm.y += gz * vm[6]; gm.z += gz * vm[10];
  tau[2] += gz;
  tau[3] += gz * pC.y;
  tau[4] += gz * -pC.x;
  o.m2x = g2x; o.m2y = g2y; o.ca = gcx; o.cb = gcy; o.cc = gcz; o.op = gop; o.col = gcol; o.dz = gz; o.gm = gm;
}
__device__ __forceinline__ void gaussian_chain(const BwdParams &p, float3 mean, const float (&c6)[6], float3 sc, float4 q,
                                               const uint8_t (&cl)[3], float4 s0, float4 s1, float4 s2, const float *sh_row,
                                               float *dsh_row, bool want_scale_rot, GaussianGrads &o, float (&tau)[6]) {
  gaussian_chain_geom(p, mean, c6, s0, s1, s2, o, tau);
  // ---- 5. colour -> SH, view direction -> mean3D, tau ----
  if (p.shs) {
    const float3 cam = make_float3(p.campos[0], p.campos[1], p.campos[2]);
    const float3 dmean = sh_backward<0>(p.D, p.M, mean, cam, sh_row, dsh_row, cl, o.col);
    o.gm.x += dmean.x; o.gm.y += dmean.y; o.gm.z += dmean.z;
    tau[0] -= dmean.x; tau[1] -= dmean.y; tau[2] -= dmean.z;
  }
  // ---- 6. cov3D -> scale, rotation ----
  if (p.scales && want_scale_rot) cov3d_backward(o.cov, sc, q, p.scale_modifier, o.scale, o.rot);
}

// Four workgroups' dL/dtau partials (8 floats each, 6 used) as seen by ANOTHER workgroup after the ticket: 16-byte loads that
// bypass this CU's L1 and this XCD's L2 (sc0 sc1).  The partials were stored write-through (agent-scope atomic stores) by CUs of
// any XCD.  Coherent loads cost ~100 ns EACH on this part and do not overlap (tools/chain_trace.py: 6 dword loads per partial
// made the last workgroup's sum 20 us of the batched kernel's 80; two dwordx4 loads: 8 us), so they are as wide as the ISA allows,
// eight to a batch, and the batch's s_waitcnt sits in the SAME asm statement: the compiler takes an asm's outputs for ready when
// the statement ends and may copy them at once.
__device__ __forceinline__ void gsaj_load_partials4(const float *p0, const float *p1, const float *p2, const float *p3,
                                                    float4 (&lo)[4], float4 (&hi)[4]) {
  asm volatile(
      "global_load_dwordx4 %0, %8, off sc0 sc1\n\tglobal_load_dwordx4 %1, %8, off offset:16 sc0 sc1\n\t"
      "global_load_dwordx4 %2, %9, off sc0 sc1\n\tglobal_load_dwordx4 %3, %9, off offset:16 sc0 sc1\n\t"
      "global_load_dwordx4 %4, %10, off sc0 sc1\n\tglobal_load_dwordx4 %5, %10, off offset:16 sc0 sc1\n\t"
      "global_load_dwordx4 %6, %11, off sc0 sc1\n\tglobal_load_dwordx4 %7, %11, off offset:16 sc0 sc1\n\t"
      "s_waitcnt vmcnt(0)"
      : "=&v"(lo[0]), "=&v"(hi[0]), "=&v"(lo[1]), "=&v"(hi[1]), "=&v"(lo[2]), "=&v"(hi[2]), "=&v"(lo[3]), "=&v"(hi[3])
      : "v"(p0), "v"(p1), "v"(p2), "v"(p3)
      : "memory");
}
// sum over workgroups i = lane, lane + 64, ... < nblk of the 6 components, in that order, in fp64
__device__ __forceinline__ void gsaj_sum_partials(const float *partials, int nblk, int lane, double (&acc)[6]) {
  for (int i0 = lane; i0 < nblk; i0 += 4 * 64) {
    float4 lo[4], hi[4];
    const float *src[4];
#pragma unroll
    for (int u = 0; u < 4; u++) src[u] = partials + (size_t)min(i0 + u * 64, nblk - 1) * 8;  // unconditional loads, masked below
    gsaj_load_partials4(src[0], src[1], src[2], src[3], lo, hi);
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const bool in = i0 + u * 64 < nblk;
      const float t6[6] = {lo[u].x, lo[u].y, lo[u].z, lo[u].w, hi[u].x, hi[u].y};
#pragma unroll
      for (int k = 0; k < 6; k++) acc[k] += in ? (double)t6[k] : 0.0;
    }
  }
}

// SHW = 3*M as a compile-time constant (0: runtime) -- the staging loops divide by it per element
GSAJ_TRACE_DEFINE(gbwd)

template <int SHW>
__global__ __launch_bounds__(GB_BLOCK) void k_gaussian_bwd(BwdParams p, GeomWS g, uint32_t *__restrict__ counters,
                                                           const float4 *__restrict__ inst_grad,
                                                           const uint8_t *__restrict__ reached) {
  extern __shared__ float sh_lds[];  // [GB_BLOCK][3M+1]: SH coefficients in, overwritten in place by dL/dSH (padded rows)
  __shared__ uint32_t s_ticket;
  if (counters[4]) return;  // aborted async frame
  GSAJ_TRACE_BEGIN(gbwd)
#ifdef GSAJ_BLOCK_TRACE
  unsigned long long tr_[5] = {0, 0, 0, 0, 0}, tr_t = wall_clock64();
#define TRM(i) { const unsigned long long n_ = wall_clock64(); tr_[i] += n_ - tr_t; tr_t = n_; }
#else
#define TRM(i)
#endif
  const int tid = threadIdx.x;
  const int idx = blockIdx.x * GB_BLOCK + tid;
  const int shw = SHW > 0 ? SHW : 3 * p.M, shs_stride = shw + 1;
  float *sh_io = sh_lds;
  // ---- 0. every input of this Gaussian is requested up front (this stage is latency-bound: one wave
  //         per SIMD, so the loads must be in flight together, not one s_waitcnt apart) ----
  const int radius = idx < p.P ? p.radii[idx] : 0;
  const bool vis = radius > 0;
  const size_t ii = (size_t)(idx < p.P ? idx : 0);
  // emission slots: Gaussian idx owns rows [first, first + cnt) of inst_grad; consecutive Gaussians
  // own consecutive runs, so the rows of this wave's 64 Gaussians are ONE contiguous block
  const uint32_t cnt = idx < p.P ? g.tiles_touched[ii] : 0u;
  // (point_offsets counts inside the Gaussian's block of PRE_BLOCK; block_sums holds the blocks' exclusive offsets)
  const uint32_t endi = idx < p.P ? g.block_sums[ii / PRE_BLOCK] + g.point_offsets[ii] : 0u;
  const uint32_t first = endi - cnt;
  // The wave's rows [F, E) of inst_grad are gathered in trips of 256; the first trip's flag and row loads (two dependent
  // round trips) are issued NOW, so that they overlap the input loads and the SH staging below.
  const uint32_t F = (uint32_t)__shfl((int)first, 0);
  uint32_t E = endi;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) E = max(E, (uint32_t)__shfl_xor((int)E, o));
  float4 a[4][3];
  auto load_trip = [&](uint32_t lo, uint32_t hi) {
#pragma unroll
    for (int w = 0; w < 4; w++) {
      const uint32_t r = lo + (uint32_t)(w * GB_BLOCK + tid);
      a[w][0] = a[w][1] = a[w][2] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (r < hi && reached[r]) {  // rows the reverse compositor never wrote are zero by definition
        const float4 *src = inst_grad + (size_t)r * REC_F4;
        a[w][0] = src[0]; a[w][1] = src[1]; a[w][2] = src[2];
      }
    }
  };
  load_trip(F, min(F + 256u, E));
  const float3 mean = make_float3(p.means3D[3 * ii], p.means3D[3 * ii + 1], p.means3D[3 * ii + 2]);
  float c6[6];
#pragma unroll
  for (int k = 0; k < 6; k++) c6[k] = p.cov3Ds[6 * ii + k];
  float3 sc = make_float3(0.f, 0.f, 0.f);
  float4 q = make_float4(1.f, 0.f, 0.f, 0.f);
  if (p.scales) {
    sc = make_float3(p.scales[3 * ii], p.scales[3 * ii + 1], p.scales[3 * ii + 2]);
    q = reinterpret_cast<const float4 *>(p.rotations)[ii];
  }
  uint8_t cl[3] = {0, 0, 0};
  if (p.shs) {
    cl[0] = g.clamped[3 * ii]; cl[1] = g.clamped[3 * ii + 1]; cl[2] = g.clamped[3 * ii + 2];
    // coalesced load of this workgroup's contiguous [GB_BLOCK][M*3] SH block, 24 loads in flight per batch
    const size_t base = (size_t)blockIdx.x * GB_BLOCK * shw;
    const int count = min(GB_BLOCK, p.P - blockIdx.x * GB_BLOCK) * shw;
    for (int e0 = 0; e0 < count; e0 += 24 * GB_BLOCK) {
      float v[24];
#pragma unroll
      for (int b = 0; b < 24; b++) {
        const int e = e0 + b * GB_BLOCK + tid;
        v[b] = e < count ? p.shs[base + e] : 0.f;
      }
#pragma unroll
      for (int b = 0; b < 24; b++) {
        const int e = e0 + b * GB_BLOCK + tid;
        if (e < count) {
          const int gi = e / shw, k = e - gi * shw;
          sh_io[gi * shs_stride + k] = v[b];
        }
      }
    }
    __syncthreads();
  }
  TRM(0)
  float tau[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  // ---- 1. gather the instance partials.  Tiles-per-Gaussian is heavy-tailed (mean ~8, max > 100), so a
  //         per-lane loop over global memory makes the whole wave wait for its largest Gaussian.
  //         Instead the wave streams its contiguous block of rows through LDS with fully coalesced
  //         loads (256 rows = 12 KB per trip, 12 loads per lane in flight) and each lane then sums its
  //         own rows from LDS in emission order (bit-reproducible). ----
  float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0, s2 = s0;
  {
    __shared__ float4 rows[256 * REC_F4];
    for (uint32_t lo = F; lo < E; lo += 256) {
      const uint32_t hi = min(lo + 256u, E);
      if (lo != F) load_trip(lo, hi);
#pragma unroll
      for (int w = 0; w < 4; w++) {
        rows[(w * GB_BLOCK + tid) * REC_F4 + 0] = a[w][0];
        rows[(w * GB_BLOCK + tid) * REC_F4 + 1] = a[w][1];
        rows[(w * GB_BLOCK + tid) * REC_F4 + 2] = a[w][2];
      }
      __syncthreads();
      const uint32_t ub = max(first, lo), ue = min(first + cnt, hi);
      for (uint32_t u = ub; u < ue; u++) {
        const float4 a0 = rows[(u - lo) * REC_F4 + 0], a1 = rows[(u - lo) * REC_F4 + 1], a2 = rows[(u - lo) * REC_F4 + 2];
        s0.x += a0.x; s0.y += a0.y; s0.z += a0.z; s0.w += a0.w;
        s1.x += a1.x; s1.y += a1.y; s1.z += a1.z; s1.w += a1.w;
        s2.x += a2.x; s2.y += a2.y;
      }
      __syncthreads();
    }
  }
  TRM(1)
  // every output of this Gaussian stays in registers until the dL/dtau hand-off below has been made:
  // the hand-off drains this wave's outstanding stores (s_waitcnt vmcnt(0)), so the bulk stores come last
  GaussianGrads o;
  o.m2x = o.m2y = o.ca = o.cb = o.cc = o.op = o.dz = 0.f;
  o.col = o.gm = o.scale = make_float3(0.f, 0.f, 0.f);
  o.rot = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int k = 0; k < 6; k++) o.cov[k] = 0.f;
  if (vis)
    gaussian_chain(p, mean, c6, sc, q, cl, s0, s1, s2, sh_io + tid * shs_stride, sh_io + tid * shs_stride,
                   p.dL_dscale != nullptr, o, tau);
  TRM(2)
  // ---- 7. wave partial of dL/dtau (fixed butterfly) ----
#pragma unroll
  for (int k = 0; k < 6; k++) {
    float v = tau[k];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    if (tid == 0)  // write-through (sc1) store: part of the fence-free hand-off below
      __hip_atomic_store(&g.tau_partials[(size_t)blockIdx.x * 8 + k], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (p.dL_dtau_sum) {
  // ---- 8. the last workgroup to arrive sums the partials in workgroup order (fp64): deterministic,
  // no extra launch (replaces torch.sum over [P,6], diff_gaussian_rasterization/__init__.py:162).
  // Hand-off without fences (MI355X_MICROARCH.md, hand-offs measured with sc1 loads in place of the
  // acquire): the 6 partials are stored write-through (sc1) by lane 0, drained with vmcnt(0), then
  // the ticket is drawn with a relaxed agent-scope atomic; the workgroup that draws the last ticket
  // reads every partial with sc1 loads only after its add has returned.  An agent release here
  // (buffer_wbl2) would write back this kernel's ~25 MB of dirty output once per workgroup.
  if (tid == 0) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    s_ticket = __hip_atomic_fetch_add(&counters[3], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  if (s_ticket == gridDim.x - 1) {
  const int nblk = (int)gridDim.x;
  double acc6[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
  gsaj_sum_partials(g.tau_partials, nblk, tid, acc6);
#pragma unroll
  for (int k = 0; k < 6; k++) {
    double v = acc6[k];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    if (tid == 0) p.dL_dtau_sum[k] = (float)v;
  }
  if (tid == 0) counters[3] = 0u;  // ready for the next backward over this workspace
  }
  }
  TRM(3)
  // ---- 9. outputs: one row per Gaussian, zeros for culled ones (the reference's binding memsets first) ----
  if (idx < p.P && p.dL_dmean2D) {  // pose-only mode (all per-Gaussian outputs NULL) stores nothing per Gaussian
    const size_t i = (size_t)idx;
    p.dL_dmean2D[3 * i] = o.m2x; p.dL_dmean2D[3 * i + 1] = o.m2y; p.dL_dmean2D[3 * i + 2] = 0.f;
    reinterpret_cast<float4 *>(p.dL_dconic)[i] = make_float4(o.ca, o.cb, 0.f, o.cc);
    p.dL_dopacity[i] = o.op;
    p.dL_dcolor[3 * i] = o.col.x; p.dL_dcolor[3 * i + 1] = o.col.y; p.dL_dcolor[3 * i + 2] = o.col.z;
    p.dL_ddepth[i] = o.dz;
    p.dL_dmean3D[3 * i] = o.gm.x; p.dL_dmean3D[3 * i + 1] = o.gm.y; p.dL_dmean3D[3 * i + 2] = o.gm.z;
#pragma unroll
    for (int k = 0; k < 6; k++) p.dL_dcov3D[6 * i + k] = o.cov[k];
    if (p.scales) {
      p.dL_dscale[3 * i] = o.scale.x; p.dL_dscale[3 * i + 1] = o.scale.y; p.dL_dscale[3 * i + 2] = o.scale.z;
      reinterpret_cast<float4 *>(p.dL_drot)[i] = o.rot;
    }
  }
  if (idx < p.P && p.dL_dtau) {
#pragma unroll
    for (int k = 0; k < 6; k++) p.dL_dtau[6 * (size_t)idx + k] = tau[k];
  }
  // dL/dSH block: coalesced store (rows of culled Gaussians and coefficients above the active degree are zero)
  if (p.shs && p.dL_dsh) {
    if (!vis) {
      for (int k = 0; k < shw; k++) sh_io[tid * shs_stride + k] = 0.f;
    }
    __syncthreads();
    const size_t base = (size_t)blockIdx.x * GB_BLOCK * shw;
    const int count = min(GB_BLOCK, p.P - blockIdx.x * GB_BLOCK) * shw;
    for (int e = tid; e < count; e += GB_BLOCK) {
      const int gi = e / shw, k = e - gi * shw;
      p.dL_dsh[base + e] = sh_io[gi * shs_stride + k];
    }
  }
  TRM(4)
  GSAJ_TRACE_END(gbwd)
#ifdef GSAJ_BLOCK_TRACE
  if (tid == 0 && blockIdx.x < GSAJ_TRACE_MAX) {
    unsigned long long *t = g_trace_gbwd + 4 * blockIdx.x;
    t[2] = (tr_[0] << 42) | (tr_[1] << 21) | tr_[2];
    t[3] = (tr_[3] << 21) | tr_[4];
  }
#endif
}

int launch_gaussian_backward(const BwdParams &p, const GeomWS &g, const BinWS &b, const ImageWS &im, hipStream_t s) {
  const int nblk = (p.P + GB_BLOCK - 1) / GB_BLOCK;
  {
    GsajProfScope ps(ST_GAUSSIAN_BWD, s);
    const size_t lds = p.shs ? sizeof(float) * GB_BLOCK * (3 * (size_t)p.M + 1) : 0;
    switch (p.shs ? p.M : -1) {
      case 1: hipLaunchKernelGGL(k_gaussian_bwd<3>, dim3(nblk), dim3(GB_BLOCK), lds, s, p, g, im.counters, b.inst_grad, b.reached); break;
      case 4: hipLaunchKernelGGL(k_gaussian_bwd<12>, dim3(nblk), dim3(GB_BLOCK), lds, s, p, g, im.counters, b.inst_grad, b.reached); break;
      case 9: hipLaunchKernelGGL(k_gaussian_bwd<27>, dim3(nblk), dim3(GB_BLOCK), lds, s, p, g, im.counters, b.inst_grad, b.reached); break;
      case 16: hipLaunchKernelGGL(k_gaussian_bwd<48>, dim3(nblk), dim3(GB_BLOCK), lds, s, p, g, im.counters, b.inst_grad, b.reached); break;
      default: hipLaunchKernelGGL(k_gaussian_bwd<0>, dim3(nblk), dim3(GB_BLOCK), lds, s, p, g, im.counters, b.inst_grad, b.reached); break;
    }
  }
  GSAJ_HIP_CHECK(hipGetLastError());
  return GSAJ_OK;
}

// ---- batched: K views of one map, per-Gaussian parameter gradients summed over the views IN-KERNEL --------------------
// A mapping window renders every keyframe against the same Gaussians and back-propagates once, so the per-Gaussian
// gradients accumulate over keyframes while every keyframe keeps its own dL/dtau (reference utils/slam_backend.py:168-232).
// Three launches:
//  k_gather_sums (grid x K, HBM-streaming, many waves in flight): a Gaussian's per-instance partial-gradient rows -- one
//    contiguous run by emission slot, a wave's 64 Gaussians one contiguous block -- are streamed with coalesced loads and
//    added in emission order; 48 bytes per Gaussian and view out (gsum).
//  k_chain_window + k_tau_sum: below.
// one step of the keyed wave scan: lanes that receive a value through the DPP pattern CTRL (row mask ROWS) add it iff it comes from
// the same owner; lanes the pattern does not reach see the key -1 and add nothing
// Each value and step is ONE v_fmac_f32 with a DPP operand: v += dpp(v) * same, same = 1.0 where the incoming value is the lane's own
// owner's, else 0.0 (sums are finite, so x * 0 = 0 and x * 1 + v is the add, bit for bit); a lane the pattern does not reach is not
// written (bound_ctrl off).  As "v += same ? dpp(v) : 0" every value and step cost a zero initialisation, a DPP move, a select and an
// add, and the kernel was as VALU-bound (0.71) as it was HBM-bound.  (Masking the adds with EXEC instead does not work: a DPP read
// of a lane that EXEC disables is an invalid read, and the lanes to be skipped are other lanes' sources.)
#define GSAJ_SCAN_STEP(DPPSTR)                                                                                          \
  asm volatile("v_fmac_f32_dpp %[v0], %[v0], %[m] " DPPSTR "\n\tv_fmac_f32_dpp %[v1], %[v1], %[m] " DPPSTR "\n\t"           \
               "v_fmac_f32_dpp %[v2], %[v2], %[m] " DPPSTR "\n\tv_fmac_f32_dpp %[v3], %[v3], %[m] " DPPSTR "\n\t"           \
               "v_fmac_f32_dpp %[v4], %[v4], %[m] " DPPSTR "\n\tv_fmac_f32_dpp %[v5], %[v5], %[m] " DPPSTR "\n\t"           \
               "v_fmac_f32_dpp %[v6], %[v6], %[m] " DPPSTR "\n\tv_fmac_f32_dpp %[v7], %[v7], %[m] " DPPSTR "\n\t"           \
               "v_fmac_f32_dpp %[v8], %[v8], %[m] " DPPSTR "\n\tv_fmac_f32_dpp %[v9], %[v9], %[m] " DPPSTR                   \
               : [v0] "+v"(v[0]), [v1] "+v"(v[1]), [v2] "+v"(v[2]), [v3] "+v"(v[3]), [v4] "+v"(v[4]), [v5] "+v"(v[5]),       \
                 [v6] "+v"(v[6]), [v7] "+v"(v[7]), [v8] "+v"(v[8]), [v9] "+v"(v[9])                                          \
               : [m] "v"(same))
template <int CTRL, int ROWS>
__device__ __forceinline__ void scan_by_key_step(int own, float (&v)[10]) {
  const int own_in = __builtin_amdgcn_update_dpp(-1, own, CTRL, ROWS, 0xF, false);
  const float same = own_in == own ? 1.0f : 0.0f;
  static_assert(CTRL == 0x111 || CTRL == 0x112 || CTRL == 0x114 || CTRL == 0x118 || CTRL == 0x142 || CTRL == 0x143, "the six steps of the wave scan");
  if constexpr (CTRL == 0x111) GSAJ_SCAN_STEP("row_shr:1 row_mask:0xf bank_mask:0xf");
  else if constexpr (CTRL == 0x112) GSAJ_SCAN_STEP("row_shr:2 row_mask:0xf bank_mask:0xf");
  else if constexpr (CTRL == 0x114) GSAJ_SCAN_STEP("row_shr:4 row_mask:0xf bank_mask:0xf");
  else if constexpr (CTRL == 0x118) GSAJ_SCAN_STEP("row_shr:8 row_mask:0xf bank_mask:0xf");
  else if constexpr (CTRL == 0x142) GSAJ_SCAN_STEP("row_bcast:15 row_mask:0xa bank_mask:0xf");
  else GSAJ_SCAN_STEP("row_bcast:31 row_mask:0xc bank_mask:0xf");
}
#undef GSAJ_SCAN_STEP

// k_gather_sums: lane = ROW.  A wave owns 64 consecutive Gaussians, whose instance rows are ONE contiguous block ordered by
// owner; it walks the block 64 rows at a time with fully coalesced loads (row + `reached` flag requested together: one memory
// round trip; the next 64 rows are requested before the current ones are reduced), finds every row's owner by a 6-step binary
// search over the owners' end slots (lane shuffles), runs a scan-by-key over the 64 rows (Hillis-Steele, add when the owner of
// lane l equals the owner of lane l - d: exact for contiguous segments; DPP, no LDS), and each owner picks up its segment's total
// from its segment's last row of the group.  No LDS staging, no per-lane loop over a heavy-tailed run length (that loop kept ~20 % of
// the lanes busy and its scattered ds_read_b128 conflicted 4-way: 100 us; this form: see profiles/).  The additions follow a
// fixed tree per 64-row group and the groups in order: bit-reproducible.
GSAJ_TRACE_DEFINE(gath)
__global__ __launch_bounds__(256) void k_gather_sums(int P, const int *__restrict__ radii0, GeomWS g0, ImageWS im0,
                                                     const float4 *__restrict__ inst_grad0, const uint8_t *__restrict__ reached0,
                                                     ViewStrides vs) {
  GSAJ_TRACE_BEGIN(gath)
  const size_t view = blockIdx.y;
  const GeomWS g = geom_view(g0, view * vs.geom);
  const float4 *inst_grad = gsaj_shift(inst_grad0, view * vs.bin);
  const uint8_t *reached = gsaj_shift(reached0, view * vs.bin);
  const uint32_t *counters = gsaj_shift(im0.counters, view * vs.image);
  const int lane = threadIdx.x & 63;
  const int idx = blockIdx.x * 256 + threadIdx.x;
  const size_t ii = (size_t)(idx < P ? idx : 0);
  const bool live = idx < P && counters[4] == 0u;  // (an aborted async frame contributes nothing)
  const uint32_t cnt = live ? g.tiles_touched[ii] : 0u;
  const uint32_t endi_raw = live ? g.block_sums[ii / PRE_BLOCK] + g.point_offsets[ii] : 0u;  // (block-local scan + the block's offset)
  // owners' end slots, made non-decreasing over the lanes (culled / out-of-range owners repeat their predecessor's end)
  uint32_t endi = endi_raw;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const uint32_t up = (uint32_t)__shfl_up((int)endi, o);
    if (lane >= o) endi = max(endi, up);
  }
  const uint32_t first = live ? endi_raw - cnt : endi;
  uint32_t F = (cnt > 0) ? first : 0xffffffffu;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) F = min(F, (uint32_t)__shfl_xor((int)F, o));
  const uint32_t E = (uint32_t)__shfl((int)endi, 63);
  float sum[10];
#pragma unroll
  for (int c = 0; c < 10; c++) sum[c] = 0.f;
  if (F != 0xffffffffu && F < E) {
    // rows the reverse compositor never wrote (`reached` = 0: no pixel of the tile got that far -- most of an opaque scene's
    // instances) are not fetched: the one-byte flags run TWO groups ahead of the rows they admit, the rows one group ahead of
    // their use; only the wave's first group is fetched without waiting for its flags
    float4 a0, a1, a2;
    uint8_t fl = 0, fl_a = 0, fl_b = 0;  // flags of the group in (a0, a1, a2), of the next group, of the one after
    auto flags_of = [&](uint32_t base) -> uint8_t { return base + (uint32_t)lane < E ? reached[base + lane] : (uint8_t)0; };
    auto rows_of = [&](uint32_t base) {
      const float4 *src = inst_grad + (size_t)(base + lane) * REC_F4;
      a0 = src[0]; a1 = src[1]; a2 = src[2];
    };
    fl = flags_of(F);
    if (F + (uint32_t)lane < E) rows_of(F);
    fl_a = flags_of(F + 64);
    fl_b = flags_of(F + 128);
    for (uint32_t base = F; base < E; base += 64) {
      const uint32_t r = base + (uint32_t)lane;
      const bool ok = fl != 0;
      float v[10] = {ok ? a0.x : 0.f, ok ? a0.y : 0.f, ok ? a0.z : 0.f, ok ? a0.w : 0.f, ok ? a1.x : 0.f,
                     ok ? a1.y : 0.f, ok ? a1.z : 0.f, ok ? a1.w : 0.f, ok ? a2.x : 0.f, ok ? a2.y : 0.f};
      if (base + 64 < E) {
        fl = fl_a;
        if (fl) rows_of(base + 64);
        fl_a = fl_b;
        fl_b = flags_of(base + 192);
      }
      // a group none of whose rows was reached adds nothing (opaque scenes: most groups of the Gaussians that lie behind)
      if (__builtin_amdgcn_ballot_w64(ok) == 0ull) continue;
      // owner of row r = number of owners whose end slot is <= r (end slots are non-decreasing): binary search by shuffles
      int own = 0;
#pragma unroll
      for (int step = 32; step > 0; step >>= 1) {
        const uint32_t e = (uint32_t)__shfl((int)endi, own + step - 1);
        if (e <= r) own += step;
      }
      // scan by key: after the last step lane l holds the sum of the rows of its owner in [max(segment start, base), r].
      // All in the VALU (DPP): four row_shr steps inside each 16-lane row, then row_bcast:15 (rows 1, 3 take lane 15 / 47) and
      // row_bcast:31 (rows 2, 3 take lane 31, which by then carries rows 0-1) -- the keyed form of the classic wave scan: a lane
      // adds the incoming value iff it comes from the same owner, exact because an owner's rows are contiguous.
      scan_by_key_step<0x111, 0xF>(own, v);  // row_shr:1
      scan_by_key_step<0x112, 0xF>(own, v);  // row_shr:2
      scan_by_key_step<0x114, 0xF>(own, v);  // row_shr:4
      scan_by_key_step<0x118, 0xF>(own, v);  // row_shr:8
      scan_by_key_step<0x142, 0xA>(own, v);  // row_bcast:15 -> rows 1, 3
      scan_by_key_step<0x143, 0xC>(own, v);  // row_bcast:31 -> rows 2, 3
      // every owner whose run meets this group takes the total at its run's last row inside the group
      const uint32_t run_lo = max(first, base), run_hi = min(first + cnt, min(base + 64u, E));
      const bool mine = cnt > 0 && run_lo < run_hi;
      const int q = mine ? (int)(run_hi - 1u - base) : 0;
#pragma unroll
      for (int c = 0; c < 10; c++) {
        const float t = __shfl(v[c], q);
        sum[c] += mine ? t : 0.f;
      }
    }
  }
  if (idx < P) {
    g.gsum[3 * ii + 0] = make_float4(sum[0], sum[1], sum[2], sum[3]);
    g.gsum[3 * ii + 1] = make_float4(sum[4], sum[5], sum[6], sum[7]);
    g.gsum[3 * ii + 2] = make_float4(sum[8], sum[9], 0.f, 0.f);
  }
  (void)radii0;
  GSAJ_TRACE_END(gath)
}

// ---- the batched per-Gaussian backward ------------------------------------------------------------------------------
// (round 2 ran it as 64 Gaussians x 4 waves, wave w taking views w, w + 4 one after the other with the SH backward inside:
// 213 VGPRs, two waves per SIMD, a wave alive for 27 us of which 5 us issued instructions -- 70 us per cfg2 window.)
//
//  k_chain_window  one workgroup = 32 Gaussians x 8 views, lane = (view, Gaussian): every lane runs the whole per-view chain on
//                  that view's sums (gaussian_chain_geom + the view-direction term of the SH colour) -- K x P independent lanes
//                  fill the chip -- writes the view's own outputs (dL/dmean2D, ..., a dL/dtau partial per 32 Gaussians), and
//                  leaves what the sums over views need in LDS: (dL/dopacity, dL/dmean3D, dL/dcov3D, the colour gradient
//                  masked by the clamp flags, the SH basis weights of the view direction).  After a barrier the workgroup adds
//                  the views IN VIEW ORDER (fixed order: bit-reproducible): thread t owns up to 8 of the 32 x (10 + 3 M)
//                  elements -- component c of a Gaussian, or dL/dSH[k][ch] = sum_v w_k(view) g_v[ch] -- through every group
//                  launch of 8 views (K > 8: further launches ADD to the outputs, in view order); dL/dscale, dL/drot are formed ONCE from the summed dL/dcov3D (they are linear in it with
//                  view-independent coefficients); the block's outputs go out transposed through LDS, contiguous rows.
//                  Nothing per (view, Gaussian) goes through memory except what the caller asked for.
//  k_tau_sum       one workgroup per view adds that view's dL/dtau partials up in slot order (fp64).  No tickets: 31 256
//                  workgroups (cfg5 window) drawing tickets from ONE word per view run at the ~88 returning atomics per us a
//                  single address sustains (MI355X_MICROARCH.md, dequeue) -- 0.36 ms of a 1.2 ms kernel when tried.
#define CW_G 32   // Gaussians per workgroup
#define CW_V 8    // views per pass
GSAJ_TRACE_DEFINE(gbb)

template <int SHW>
__global__ __launch_bounds__(CW_G * CW_V, 4) void k_chain_window(BwdParams p, int v0, int K, GeomWS g0, ImageWS im0, ViewStrides vs,
                                                              float *__restrict__ pv_mean2D, float *__restrict__ pv_conic,
                                                              float *__restrict__ pv_color, float *__restrict__ pv_depth,
                                                              float *__restrict__ pv_tau, int accumulate) {
  constexpr int MC = SHW / 3;           // SH coefficients stored
  constexpr int NV = 13 + MC;           // values a (view, Gaussian) lane leaves for the sums
  constexpr int NE = 10 + SHW;          // summed outputs per Gaussian
  constexpr int EPT = (NE * CW_G + CW_G * CW_V - 1) / (CW_G * CW_V);  // elements per thread
  __shared__ float rowv[CW_V * NV * CW_G];  // [view lane][value][Gaussian]
  __shared__ float outv[NE * CW_G];         // [element][Gaussian]: the sums, for the transposed store
  GSAJ_TRACE_BEGIN(gbb)
  const int tid = threadIdx.x, gl = tid & (CW_G - 1), vl = tid / CW_G;
  const int idx = blockIdx.x * CW_G + gl;
  const size_t ii = (size_t)(idx < p.P ? idx : 0);
  // inputs that do not depend on the view: once
  const float3 mean = make_float3(p.means3D[3 * ii], p.means3D[3 * ii + 1], p.means3D[3 * ii + 2]);
  float c6[6];
#pragma unroll
  for (int k = 0; k < 6; k++) c6[k] = p.cov3Ds[6 * ii + k];  // (view 0's copy: Sigma = R S^2 R^T does not depend on the view)
  const float *vm0 = p.viewmatrix, *pj0 = p.projmatrix;
  float acc[EPT];
#pragma unroll
  for (int j = 0; j < EPT; j++) acc[j] = 0.f;
  // views [v0, min(v0 + 8, K)) of the window.  (A loop over the groups of 8 views INSIDE the kernel cost 70 VGPRs: 201 instead of
  // 132 -- the compiler kept view-independent values alive across it, whatever was hidden from it.)
  {
    const int v = v0 + vl;
    GaussianGrads o;
    o.m2x = o.m2y = o.ca = o.cb = o.cc = o.op = o.dz = 0.f;
    o.col = o.gm = make_float3(0.f, 0.f, 0.f);
#pragma unroll
    for (int k = 0; k < 6; k++) o.cov[k] = 0.f;
    float tau[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float w[MC > 0 ? MC : 1];  // SH basis weights of this view's direction
#pragma unroll
    for (int k = 0; k < MC; k++) w[k] = 0.f;
    float3 gmask = make_float3(0.f, 0.f, 0.f);
    if (v < K) {
      const GeomWS g = geom_view(g0, (size_t)v * vs.geom);
      const size_t iv = ii;
      const float3 mean_v = mean;
      const float (&c6v)[6] = c6;
      // every input of this (view, Gaussian) requested up front and unconditionally
      const uint32_t aborted = gsaj_shift(im0.counters, (size_t)v * vs.image)[4];  // aborted async frame: contributes nothing
      const int rad = p.radii[(size_t)v * p.P + ii];
      const float4 s0 = g.gsum[3 * ii + 0], s1 = g.gsum[3 * ii + 1], s2 = g.gsum[3 * ii + 2];
      uint8_t cl[3] = {0, 0, 0};
      if (SHW > 0) { cl[0] = g.clamped[3 * ii]; cl[1] = g.clamped[3 * ii + 1]; cl[2] = g.clamped[3 * ii + 2]; }
      p.viewmatrix = vm0 + 16 * v;
      p.projmatrix = pj0 + 16 * v;
      const bool vis = idx < p.P && !aborted && rad > 0;
      if (vis) {
        // the view-direction term first: its 3 M coefficient loads are consumed (and their registers free) before the geometric
        // chain's own peak
        float3 dmean = make_float3(0.f, 0.f, 0.f);
        if (SHW > 0) {
          const float *cam = p.campos + 3 * v;
          constexpr int DEG_MAX = MC >= 16 ? 3 : (MC >= 9 ? 2 : (MC >= 4 ? 1 : 0));  // (the storage bounds the degree: dead bands compile away)
          const float3 gcol = make_float3(s1.z, s1.w, s2.x);
          dmean = sh_backward<1>(min(p.D, DEG_MAX), p.M, mean_v, make_float3(cam[0], cam[1], cam[2]), p.shs + iv * SHW, w, cl, gcol);
          gmask = make_float3(cl[0] ? 0.f : gcol.x, cl[1] ? 0.f : gcol.y, cl[2] ? 0.f : gcol.z);
          __builtin_amdgcn_sched_barrier(0);
        }
        gaussian_chain_geom(p, mean_v, c6v, s0, s1, s2, o, tau);
        if (SHW > 0) {
          o.gm.x += dmean.x; o.gm.y += dmean.y; o.gm.z += dmean.z;
          tau[0] -= dmean.x; tau[1] -= dmean.y; tau[2] -= dmean.z;
        }
      }
      // the view's own outputs
      if (idx < p.P) {
        const size_t row = (size_t)v * p.P + ii;
        if (pv_mean2D) { pv_mean2D[3 * row] = o.m2x; pv_mean2D[3 * row + 1] = o.m2y; pv_mean2D[3 * row + 2] = 0.f; }
        if (pv_conic) reinterpret_cast<float4 *>(pv_conic)[row] = make_float4(o.ca, o.cb, 0.f, o.cc);
        if (pv_color) { pv_color[3 * row] = o.col.x; pv_color[3 * row + 1] = o.col.y; pv_color[3 * row + 2] = o.col.z; }
        if (pv_depth) pv_depth[row] = o.dz;
        if (pv_tau) {
#pragma unroll
          for (int k = 0; k < 6; k++) pv_tau[6 * row + k] = tau[k];
        }
      }
      // this view's dL/dtau: partial over the 32 Gaussians (the lanes of one half-wave), one slot per (view, workgroup); k_tau_sum
      // adds the slots up
      float t6[6];
#pragma unroll
      for (int k = 0; k < 6; k++) {
        float t = tau[k];
#pragma unroll
        for (int o2 = CW_G / 2; o2 > 0; o2 >>= 1) t += __shfl_xor(t, o2);
        t6[k] = t;
      }
      if (gl == 0) {
        float4 *tp = reinterpret_cast<float4 *>(g.tau_partials + (size_t)blockIdx.x * 8);
        tp[0] = make_float4(t6[0], t6[1], t6[2], t6[3]);
        tp[1] = make_float4(t6[4], t6[5], 0.f, 0.f);
      }
    }
    // ---- what the sums over views need, through LDS ----
    float *mine = rowv + vl * NV * CW_G + gl;
    mine[0 * CW_G] = o.op; mine[1 * CW_G] = o.gm.x; mine[2 * CW_G] = o.gm.y; mine[3 * CW_G] = o.gm.z;
#pragma unroll
    for (int k = 0; k < 6; k++) mine[(4 + k) * CW_G] = o.cov[k];
    mine[10 * CW_G] = gmask.x; mine[11 * CW_G] = gmask.y; mine[12 * CW_G] = gmask.z;
#pragma unroll
    for (int k = 0; k < MC; k++) mine[(13 + k) * CW_G] = w[k];
    __syncthreads();
    const int nv = min(CW_V, K - v0);
#pragma unroll
    for (int j = 0; j < EPT; j++) {
      const int e = tid + j * (CW_G * CW_V);  // element e = c * 32 + Gaussian: this thread's Gaussian is always gl
      const int c = e / CW_G;
      if (c < NE) {
        float sacc = acc[j];
        if (c < 10) {
#pragma unroll 2
          for (int u = 0; u < nv; u++) sacc += rowv[u * NV * CW_G + c * CW_G + gl];
        } else {
          const int kk = (c - 10) / 3, ch = (c - 10) - 3 * kk;
#pragma unroll 2
          for (int u = 0; u < nv; u++) sacc += rowv[u * NV * CW_G + (13 + kk) * CW_G + gl] * rowv[u * NV * CW_G + (10 + ch) * CW_G + gl];
        }
        acc[j] = sacc;
      }
    }
    __syncthreads();
  }
  // ---- the sums, transposed through LDS: [element][Gaussian] -> rows of the outputs ----
#pragma unroll
  for (int j = 0; j < EPT; j++) {
    const int e = tid + j * (CW_G * CW_V);
    if (e < NE * CW_G) outv[e] = acc[j];
  }
  __syncthreads();
  // accumulate: this call's sums are ADDED to what the buffers hold (a window processed in several calls, or keyframes of several
  // windows accumulated on one rank before the optimiser step: callers add in a fixed order, so the result stays reproducible)
#define OUT(dst, x) dst = accumulate ? dst + (x) : (x)
  const int nG = min(CW_G, p.P - (int)blockIdx.x * CW_G);  // Gaussians of this workgroup
  const size_t b0 = (size_t)blockIdx.x * CW_G;
  for (int e = tid; e < nG; e += CW_G * CW_V) OUT(p.dL_dopacity[b0 + e], outv[0 * CW_G + e]);
  for (int e = tid; e < 3 * nG; e += CW_G * CW_V) OUT(p.dL_dmean3D[3 * b0 + e], outv[(1 + e % 3) * CW_G + e / 3]);
  for (int e = tid; e < 6 * nG; e += CW_G * CW_V) OUT(p.dL_dcov3D[6 * b0 + e], outv[(4 + e % 6) * CW_G + e / 6]);
  if (SHW > 0 && p.dL_dsh) {  // (coefficients above the active degree have zero weights: their gradients come out zero)
    for (int e = tid; e < SHW * nG; e += CW_G * CW_V) OUT(p.dL_dsh[(size_t)SHW * b0 + e], outv[(10 + e % SHW) * CW_G + e / SHW]);
  }
  if (p.scales && tid < nG) {  // dL/dscale, dL/drot from the summed dL/dcov3D (this call's part: both are linear in it)
    const size_t i = b0 + tid;
    float gcov[6];
#pragma unroll
    for (int k = 0; k < 6; k++) gcov[k] = outv[(4 + k) * CW_G + tid];
    const float3 sc = make_float3(p.scales[3 * i], p.scales[3 * i + 1], p.scales[3 * i + 2]);
    const float4 q = reinterpret_cast<const float4 *>(p.rotations)[i];
    float3 dscale;
    float4 drot;
    cov3d_backward(gcov, sc, q, p.scale_modifier, dscale, drot);
    OUT(p.dL_dscale[3 * i], dscale.x); OUT(p.dL_dscale[3 * i + 1], dscale.y); OUT(p.dL_dscale[3 * i + 2], dscale.z);
    float4 *dr = reinterpret_cast<float4 *>(p.dL_drot) + i;
    if (accumulate) { const float4 o4 = *dr; drot.x += o4.x; drot.y += o4.y; drot.z += o4.z; drot.w += o4.w; }
    *dr = drot;
  }
#undef OUT
  GSAJ_TRACE_END(gbb)
}

// workgroup `view`: dL_dtau_sum[view] = sum of the view's partials (one per 32 Gaussians) in slot order, fp64
__global__ __launch_bounds__(256) void k_tau_sum(BwdParams p, GeomWS g0, ViewStrides vs) {
  __shared__ double red[256 * 6];
  const int tid = threadIdx.x, view = blockIdx.x, nslot = (p.P + CW_G - 1) / CW_G;
  const float4 *tp = reinterpret_cast<const float4 *>(gsaj_shift(g0.tau_partials, (size_t)view * vs.geom));
  double a[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
  for (int i = tid; i < nslot; i += 256) {
    const float4 lo = tp[2 * i], hi = tp[2 * i + 1];
    a[0] += (double)lo.x; a[1] += (double)lo.y; a[2] += (double)lo.z; a[3] += (double)lo.w; a[4] += (double)hi.x; a[5] += (double)hi.y;
  }
#pragma unroll
  for (int k = 0; k < 6; k++) red[k * 256 + tid] = a[k];
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {  // fixed tree
    if (tid < o) {
#pragma unroll
      for (int k = 0; k < 6; k++) red[k * 256 + tid] += red[k * 256 + tid + o];
    }
    __syncthreads();
  }
  if (tid < 6 && p.dL_dtau_sum) p.dL_dtau_sum[6 * view + tid] = (float)red[tid * 256];
}

template <int SHW>
static void launch_chain(const BwdParams &p, int K, const GeomWS &g, const ImageWS &im, ViewStrides vs, int accumulate, hipStream_t s) {
  for (int v0 = 0; v0 < K; v0 += CW_V)  // 8 views per launch; later launches add to the first one's sums (same stream: in view order)
    hipLaunchKernelGGL(k_chain_window<SHW>, dim3((p.P + CW_G - 1) / CW_G), dim3(CW_G * CW_V), 0, s, p, v0, K, g, im, vs, p.dL_dmean2D,
                       p.dL_dconic, p.dL_dcolor, p.dL_ddepth, p.dL_dtau, (accumulate || v0 > 0) ? 1 : 0);
  hipLaunchKernelGGL(k_tau_sum, dim3(K), dim3(256), 0, s, p, g, vs);
}

int launch_gather_sums(int P, int K, const int *radii, const GeomWS &g, const BinWS &b, const ImageWS &im, ViewStrides vs, hipStream_t s) {
  GsajProfScope ps(ST_GATHER_SUMS, s);
  hipLaunchKernelGGL(k_gather_sums, dim3((P + 255) / 256, K), dim3(256), 0, s, P, radii, g, im, b.inst_grad, b.reached, vs);
  GSAJ_HIP_CHECK(hipGetLastError());
  return GSAJ_OK;
}

int launch_gaussian_backward_batch(const BwdParams &p, int K, const GeomWS &g, const BinWS &b, const ImageWS &im, ViewStrides vs,
                                   int accumulate, hipStream_t s) {
  (void)b;
  GsajProfScope ps(ST_GAUSSIAN_BWD, s);
  switch (p.shs ? p.M : 0) {
    case 0: launch_chain<0>(p, K, g, im, vs, accumulate, s); break;
    case 1: launch_chain<3>(p, K, g, im, vs, accumulate, s); break;
    case 4: launch_chain<12>(p, K, g, im, vs, accumulate, s); break;
    case 9: launch_chain<27>(p, K, g, im, vs, accumulate, s); break;
    case 16: launch_chain<48>(p, K, g, im, vs, accumulate, s); break;
    default:
      gsaj_set_error("batched backward: SH storage of %d coefficients is not supported (1, 4, 9 or 16)", p.M);
      return GSAJ_ERR_INVALID_ARGUMENT;
  }
  GSAJ_HIP_CHECK(hipGetLastError());
  return GSAJ_OK;
}
