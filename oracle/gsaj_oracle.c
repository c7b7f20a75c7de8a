/*
 * gsaj_oracle.c -- CPU ORACLE (TEST INFRASTRUCTURE ONLY, never a product path).
 *
 * A plain-C, single-threaded, fp32 restatement of the *rasteriser semantics* of the
 * reference's Gaussian-splat hot path: per-Gaussian projection, 16x16 tile binning,
 * front-to-back alpha compositing of colour + depth, and the closed-form backward that
 * yields dL/dmean2D, dL/dconic, per-Gaussian parameter gradients and the SE(3) pose
 * Jacobian dL/dtau.  It follows the algorithm of (paths relative to /root/reference):
 *
 *   submodules/diff-gaussian-rasterization/cuda_rasterizer/forward.cu:22-73   (SH -> RGB)
 *   .../forward.cu:76-115   (EWA cov2D, +0.3 dilation)      .../forward.cu:120-154 (cov3D)
 *   .../forward.cu:157-401  (per-Gaussian preprocess)       .../forward.cu:406-535 (tile compositor)
 *   .../auxiliary.h:41-56   (ndc2Pix, getRect)              .../auxiliary.h:139-164 (in_frustum, z<=0.2 cull)
 *   .../rasterizer_impl.cu:70-138,327-368 (key = tile<<32|depth bits, stable sort, tile ranges)
 *   .../backward.cu:648-872 (reverse compositor)            .../backward.cu:150-422 (cov2D backward + tau)
 *   .../backward.cu:494-624 (preprocess backward + tau)     .../backward.cu:21-145  (SH backward + tau)
 *   .../backward.cu:426-489 (cov3D -> scale/rot backward)
 *   .../diff_gaussian_rasterization/__init__.py:162-164 (sum of per-Gaussian dL/dtau -> 6 numbers)
 *
 * It is NOT a copy: the code below is written from the maths in
 * row-major "maths" notation, scalar loops, no GLM, no cooperative groups.  Where the
 * reference's evaluation order decides an integer result (radius, tile rectangle) the
 * same association order is kept so that integer outputs are reproducible.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this.
 *
 * Pinning: the CUDA side of the reference cannot run here (no nvcc / NVIDIA GPU), so this
 * tiled-mode restatement is pinned through (i) the reference's importable NumPy functions
 * for the shared sub-steps (compute_cov2d, ndc2Pix, eval_sh, compute_sh_backward_single,
 * dnormvdv, the dense compositing backward where thresholds are inactive) -- see
 * tests/golden/make_goldens.py -- and (ii) finite differences of its own forward.
 * Against CUDA outputs themselves: parity unpinned (no runnable reference, inputs of the
 * recorded grad_tau prints are missing blobs).
 *
 * Build: gcc -O2 -ffp-contract=off -fPIC -shared (see oracle/Makefile).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define TILE 16

static const float SH_C0 = 0.28209479177387814f;
static const float SH_C1 = 0.4886025119029199f;
static const float SH_C2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f,
                               -1.0925484305920792f, 0.5462742152960396f};
static const float SH_C3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f,
                               0.3731763325901154f,  -0.4570457994644658f, 1.445305721320277f,
                               -0.5900435899266435f};

/* viewmatrix / projmatrix arrive as 16 floats = column-major W2C / P*W2C
 * (reference: auxiliary.h:58-77). */
static void xform4x3(const float *m, const float *p, float *o) {
  o[0] = m[0] * p[0] + m[4] * p[1] + m[8] * p[2] + m[12];
  o[1] = m[1] * p[0] + m[5] * p[1] + m[9] * p[2] + m[13];
  o[2] = m[2] * p[0] + m[6] * p[1] + m[10] * p[2] + m[14];
}
static void xform4x4(const float *m, const float *p, float *o) {
  o[0] = m[0] * p[0] + m[4] * p[1] + m[8] * p[2] + m[12];
  o[1] = m[1] * p[0] + m[5] * p[1] + m[9] * p[2] + m[13];
  o[2] = m[2] * p[0] + m[6] * p[1] + m[10] * p[2] + m[14];
  o[3] = m[3] * p[0] + m[7] * p[1] + m[11] * p[2] + m[15];
}

/* pixel = ((ndc + 1) * S - 1) / 2, evaluated in double like the reference's
 * un-suffixed constants (auxiliary.h:41-44). */
static float ndc2pix(float v, int S) { return (float)((((double)v + 1.0) * (double)S - 1.0) * 0.5); }

static int imin(int a, int b) { return a < b ? a : b; }
static int imax(int a, int b) { return a > b ? a : b; }

/* Tile rectangle [x0,x1) x [y0,y1) touched by a disc of integer radius r (auxiliary.h:46-56). */
static void tile_rect(float px, float py, int r, int gx, int gy, int *x0, int *y0, int *x1, int *y1) {
  *x0 = imin(gx, imax(0, (int)((px - (float)r) / (float)TILE)));
  *y0 = imin(gy, imax(0, (int)((py - (float)r) / (float)TILE)));
  *x1 = imin(gx, imax(0, (int)((px + (float)r + (float)(TILE - 1)) / (float)TILE)));
  *y1 = imin(gy, imax(0, (int)((py + (float)r + (float)(TILE - 1)) / (float)TILE)));
}

/* Rotation matrix (row-major) of an un-normalised quaternion (r,x,y,z); forward.cu:129-142. */
static void quat_to_R(const float *q, float R[3][3]) {
  float r = q[0], x = q[1], y = q[2], z = q[3];
  R[0][0] = 1.f - 2.f * (y * y + z * z); R[0][1] = 2.f * (x * y - r * z);       R[0][2] = 2.f * (x * z + r * y);
  R[1][0] = 2.f * (x * y + r * z);       R[1][1] = 1.f - 2.f * (x * x + z * z); R[1][2] = 2.f * (y * z - r * x);
  R[2][0] = 2.f * (x * z - r * y);       R[2][1] = 2.f * (y * z + r * x);       R[2][2] = 1.f - 2.f * (x * x + y * y);
}

/* Sigma = R S^2 R^T, upper triangle (xx,xy,xz,yy,yz,zz); forward.cu:120-154.
 * A[i][j] = s_i * R[j][i]  (A = S R^T), Sigma[r][c] = sum_k A[k][r] A[k][c]. */
static void cov3d_from_scale_rot(const float *scale, float mod, const float *rot, float *c6) {
  float R[3][3], A[3][3], s[3] = {mod * scale[0], mod * scale[1], mod * scale[2]};
  quat_to_R(rot, R);
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) A[i][j] = s[i] * R[j][i];
  float S[3][3];
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 3; c++) S[r][c] = A[0][r] * A[0][c] + A[1][r] * A[1][c] + A[2][r] * A[2][c];
  c6[0] = S[0][0]; c6[1] = S[0][1]; c6[2] = S[0][2]; c6[3] = S[1][1]; c6[4] = S[1][2]; c6[5] = S[2][2];
}

/* Shared by forward and backward: clamped camera-space point t, the 2x3 matrix
 * M = J * Rcw (rows of the projective Jacobian times the rotation block), and the
 * dilated 2D covariance (a,b,c).  forward.cu:76-115 / backward.cu:176-205. */
typedef struct {
  float t[3], txtz, tytz, limx, limy;
  float J00, J02, J11, J12;
  float Rcw[3][3]; /* row-major rotation block of W2C */
  float M[2][3];
  float V[3][3];
  float a, b, c;
} Cov2D;

static void cov2d_eval(const float *mean, float fx, float fy, float tanx, float tany, const float *c6,
                       const float *vm, Cov2D *o) {
  xform4x3(vm, mean, o->t);
  o->limx = 1.3f * tanx; o->limy = 1.3f * tany;
  o->txtz = o->t[0] / o->t[2]; o->tytz = o->t[1] / o->t[2];
  o->t[0] = fminf(o->limx, fmaxf(-o->limx, o->txtz)) * o->t[2];
  o->t[1] = fminf(o->limy, fmaxf(-o->limy, o->tytz)) * o->t[2];
  float tz = o->t[2];
  o->J00 = fx / tz; o->J02 = -(fx * o->t[0]) / (tz * tz);
  o->J11 = fy / tz; o->J12 = -(fy * o->t[1]) / (tz * tz);
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 3; c++) o->Rcw[r][c] = vm[4 * c + r];
  for (int k = 0; k < 3; k++) {
    o->M[0][k] = o->Rcw[0][k] * o->J00 + o->Rcw[1][k] * 0.0f + o->Rcw[2][k] * o->J02;
    o->M[1][k] = o->Rcw[0][k] * 0.0f + o->Rcw[1][k] * o->J11 + o->Rcw[2][k] * o->J12;
  }
  o->V[0][0] = c6[0]; o->V[0][1] = c6[1]; o->V[0][2] = c6[2];
  o->V[1][0] = c6[1]; o->V[1][1] = c6[3]; o->V[1][2] = c6[4];
  o->V[2][0] = c6[2]; o->V[2][1] = c6[4]; o->V[2][2] = c6[5];
  float X[2][3];
  for (int r = 0; r < 2; r++)
    for (int c = 0; c < 3; c++) X[r][c] = o->M[r][0] * o->V[c][0] + o->M[r][1] * o->V[c][1] + o->M[r][2] * o->V[c][2];
  o->a = (X[0][0] * o->M[0][0] + X[0][1] * o->M[0][1] + X[0][2] * o->M[0][2]) + 0.3f;
  o->b = X[1][0] * o->M[0][0] + X[1][1] * o->M[0][1] + X[1][2] * o->M[0][2];
  o->c = (X[1][0] * o->M[1][0] + X[1][1] * o->M[1][1] + X[1][2] * o->M[1][2]) + 0.3f;
}

/* SH basis -> RGB (+0.5, clamp at 0 with flags); forward.cu:22-73. sh is [M][3]. */
static void sh_to_rgb(int deg, const float *pos, const float *campos, const float *sh, float *rgb, uint8_t *clamped) {
  float d[3] = {pos[0] - campos[0], pos[1] - campos[1], pos[2] - campos[2]};
  float len = sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
  float x = d[0] / len, y = d[1] / len, z = d[2] / len;
  for (int ch = 0; ch < 3; ch++) {
#define SHC(k) sh[(k) * 3 + ch]
    float res = SH_C0 * SHC(0);
    if (deg > 0) {
      res = res - SH_C1 * y * SHC(1) + SH_C1 * z * SHC(2) - SH_C1 * x * SHC(3);
      if (deg > 1) {
        float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
        res = res + SH_C2[0] * xy * SHC(4) + SH_C2[1] * yz * SHC(5) + SH_C2[2] * (2.0f * zz - xx - yy) * SHC(6) +
              SH_C2[3] * xz * SHC(7) + SH_C2[4] * (xx - yy) * SHC(8);
        if (deg > 2) {
          res = res + SH_C3[0] * y * (3.0f * xx - yy) * SHC(9) + SH_C3[1] * xy * z * SHC(10) +
                SH_C3[2] * y * (4.0f * zz - xx - yy) * SHC(11) +
                SH_C3[3] * z * (2.0f * zz - 3.0f * xx - 3.0f * yy) * SHC(12) +
                SH_C3[4] * x * (4.0f * zz - xx - yy) * SHC(13) + SH_C3[5] * z * (xx - yy) * SHC(14) +
                SH_C3[6] * x * (xx - 3.0f * yy) * SHC(15);
        }
      }
    }
#undef SHC
    res += 0.5f;
    clamped[ch] = (res < 0.0f);
    rgb[ch] = res < 0.0f ? 0.0f : res;
  }
}

/* ------------------------------------------------------------------------------------ */
/* Stage 1: per-Gaussian preprocess.  Returns the number of (Gaussian, tile) instances R,
 * or -1 if `prefiltered` is set and a point is culled (the reference traps,
 * auxiliary.h:156-160).  All output arrays have P rows and are fully written. */
int gsaj_oracle_preprocess(int P, int D, int M, int W, int H,
                           const float *means3D, const float *shs, const float *colors_precomp,
                           const float *opacities, const float *scales, float scale_modifier,
                           const float *rotations, const float *cov3D_precomp,
                           const float *viewmatrix, const float *projmatrix, const float *campos,
                           float tanfovx, float tanfovy, int prefiltered,
                           /* out */ int *radii, float *means2D, float *depths, float *cov3D, float *conic_opacity,
                           float *rgb, uint8_t *clamped, int *tiles_touched) {
  const float fy = (float)H / (2.0f * tanfovy), fx = (float)W / (2.0f * tanfovx);
  const int gx = (W + TILE - 1) / TILE, gy = (H + TILE - 1) / TILE;
  long R = 0;
  for (int i = 0; i < P; i++) {
    radii[i] = 0; tiles_touched[i] = 0;
    means2D[2 * i] = means2D[2 * i + 1] = 0.f; depths[i] = 0.f;
    for (int k = 0; k < 6; k++) cov3D[6 * i + k] = 0.f;
    for (int k = 0; k < 4; k++) conic_opacity[4 * i + k] = 0.f;
    for (int k = 0; k < 3; k++) { rgb[3 * i + k] = 0.f; clamped[3 * i + k] = 0; }
    const float *p = means3D + 3 * i;
    float ph[4], pv[3];
    xform4x4(projmatrix, p, ph);
    float pw = 1.0f / (ph[3] + 0.0000001f);
    float pproj[3] = {ph[0] * pw, ph[1] * pw, ph[2] * pw};
    xform4x3(viewmatrix, p, pv);
    if (pv[2] <= 0.2f) {
      if (prefiltered) return -1;
      continue;
    }
    const float *c6;
    if (cov3D_precomp) c6 = cov3D_precomp + 6 * i;
    else { cov3d_from_scale_rot(scales + 3 * i, scale_modifier, rotations + 4 * i, cov3D + 6 * i); c6 = cov3D + 6 * i; }
    Cov2D cv;
    cov2d_eval(p, fx, fy, tanfovx, tanfovy, c6, viewmatrix, &cv);
    float det = cv.a * cv.c - cv.b * cv.b;
    if (det == 0.0f) continue;
    float det_inv = 1.f / det;
    float conic[3] = {cv.c * det_inv, -cv.b * det_inv, cv.a * det_inv};
    float mid = 0.5f * (cv.a + cv.c);
    float sq = sqrtf(fmaxf(0.1f, mid * mid - det));
    float l1 = mid + sq, l2 = mid - sq;
    float my_radius = ceilf(3.f * sqrtf(fmaxf(l1, l2)));
    float px = ndc2pix(pproj[0], W), py = ndc2pix(pproj[1], H);
    int x0, y0, x1, y1;
    tile_rect(px, py, (int)my_radius, gx, gy, &x0, &y0, &x1, &y1);
    if ((x1 - x0) * (y1 - y0) == 0) continue;
    if (!colors_precomp) sh_to_rgb(D, p, campos, shs + (size_t)i * M * 3, rgb + 3 * i, clamped + 3 * i);
    depths[i] = pv[2];
    radii[i] = (int)my_radius;
    means2D[2 * i] = px; means2D[2 * i + 1] = py;
    conic_opacity[4 * i] = conic[0]; conic_opacity[4 * i + 1] = conic[1];
    conic_opacity[4 * i + 2] = conic[2]; conic_opacity[4 * i + 3] = opacities[i];
    tiles_touched[i] = (y1 - y0) * (x1 - x0);
    R += tiles_touched[i];
  }
  return (int)R;
}

/* Frustum test only (rasterizer_impl.cu:54-66). */
void gsaj_oracle_mark_visible(int P, const float *means3D, const float *viewmatrix, uint8_t *present) {
  for (int i = 0; i < P; i++) {
    float pv[3];
    xform4x3(viewmatrix, means3D + 3 * i, pv);
    present[i] = pv[2] > 0.2f;
  }
}

/* Stage 2: emit keys, stable sort by (tile, depth bits), tile ranges.
 * point_list[R]: Gaussian ids; ranges[2*tiles]: [start,end). rasterizer_impl.cu:70-138,339-368. */
typedef struct { uint64_t key; uint32_t val; uint32_t seq; } KV;
static int kv_cmp(const void *a, const void *b) {
  const KV *x = (const KV *)a, *y = (const KV *)b;
  if (x->key != y->key) return x->key < y->key ? -1 : 1;
  return x->seq < y->seq ? -1 : (x->seq > y->seq ? 1 : 0); /* stable: emission order */
}
int gsaj_oracle_bin(int P, int W, int H, int R, const int *radii, const float *means2D, const float *depths,
                    /* out */ uint32_t *point_list, uint64_t *keys_sorted, int *ranges) {
  const int gx = (W + TILE - 1) / TILE, gy = (H + TILE - 1) / TILE;
  KV *kv = (KV *)malloc(sizeof(KV) * (size_t)(R > 0 ? R : 1));
  if (!kv) return -2;
  uint32_t off = 0;
  for (int i = 0; i < P; i++) {
    if (radii[i] <= 0) continue;
    int x0, y0, x1, y1;
    tile_rect(means2D[2 * i], means2D[2 * i + 1], radii[i], gx, gy, &x0, &y0, &x1, &y1);
    uint32_t dbits;
    memcpy(&dbits, depths + i, 4);
    for (int y = y0; y < y1; y++)
      for (int x = x0; x < x1; x++) {
        kv[off].key = ((uint64_t)(uint32_t)(y * gx + x) << 32) | dbits;
        kv[off].val = (uint32_t)i; kv[off].seq = off; off++;
      }
  }
  if ((int)off != R) { free(kv); return -3; }
  qsort(kv, (size_t)R, sizeof(KV), kv_cmp);
  memset(ranges, 0, sizeof(int) * 2 * (size_t)gx * gy);
  for (int k = 0; k < R; k++) {
    point_list[k] = kv[k].val;
    if (keys_sorted) keys_sorted[k] = kv[k].key;
    uint32_t cur = (uint32_t)(kv[k].key >> 32);
    if (k == 0) ranges[2 * cur] = 0;
    else {
      uint32_t prev = (uint32_t)(kv[k - 1].key >> 32);
      if (cur != prev) { ranges[2 * prev + 1] = k; ranges[2 * cur] = k; }
    }
    if (k == R - 1) ranges[2 * cur + 1] = R;
  }
  free(kv);
  return 0;
}

/* Stage 3: per-pixel front-to-back compositing (forward.cu:406-535).
 * out_color [3,H,W], out_depth [H,W], out_opacity [H,W], final_T [H,W], n_contrib [H,W],
 * n_touched [P] (must be zeroed by caller). Returns sum over pixels of n_contrib
 * (= the interaction count I of SURVEY 8d). */
long gsaj_oracle_render(int W, int H, const int *ranges, const uint32_t *point_list, const float *means2D,
                        const float *features, const float *conic_opacity, const float *depths, const float *bg,
                        float *out_color, float *out_depth, float *out_opacity, float *final_T,
                        uint32_t *n_contrib, int *n_touched) {
  const int gx = (W + TILE - 1) / TILE;
  long interactions = 0;
  for (int py = 0; py < H; py++)
    for (int px = 0; px < W; px++) {
      int tile = (py / TILE) * gx + (px / TILE);
      int beg = ranges[2 * tile], end = ranges[2 * tile + 1];
      float T = 1.0f, C[3] = {0, 0, 0}, Dp = 0.0f;
      uint32_t contributor = 0, last = 0;
      float pxf = (float)px, pyf = (float)py;
      for (int k = beg; k < end; k++) {
        contributor++;
        uint32_t g = point_list[k];
        float dx = means2D[2 * g] - pxf, dy = means2D[2 * g + 1] - pyf;
        const float *co = conic_opacity + 4 * g;
        float power = -0.5f * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
        if (power > 0.0f) continue;
        float alpha = fminf(0.99f, co[3] * expf(power));
        if (alpha < 1.0f / 255.0f) continue;
        float test_T = T * (1 - alpha);
        if (test_T < 0.0001f) break;
        for (int ch = 0; ch < 3; ch++) C[ch] += features[3 * g + ch] * alpha * T;
        Dp += depths[g] * alpha * T;
        if (test_T > 0.5f) n_touched[g]++;
        T = test_T;
        last = contributor;
      }
      size_t pid = (size_t)py * W + px;
      final_T[pid] = T; n_contrib[pid] = last;
      for (int ch = 0; ch < 3; ch++) out_color[(size_t)ch * H * W + pid] = C[ch] + T * bg[ch];
      out_depth[pid] = Dp; out_opacity[pid] = 1 - T;
      interactions += last;
    }
  return interactions;
}

/* Stage 4: reverse compositor (backward.cu:648-872).  Per-Gaussian sums are kept in
 * double and rounded once (the reference's float atomics have no defined order).
 * dL_dmean2D [P,3] (z unused), dL_dconic [P,4] (slots 0,1,3), dL_dopacity [P], dL_dcolor [P,3], dL_ddepth [P]. */
void gsaj_oracle_render_backward(int P, int W, int H, const int *ranges, const uint32_t *point_list,
                                 const float *means2D, const float *conic_opacity, const float *colors,
                                 const float *depths, const float *bg, const float *final_T,
                                 const uint32_t *n_contrib, const float *dL_dpix, const float *dL_dpix_depth,
                                 float *dL_dmean2D, float *dL_dconic, float *dL_dopacity, float *dL_dcolor,
                                 float *dL_ddepth) {
  const int gx = (W + TILE - 1) / TILE;
  double *acc = (double *)calloc((size_t)P * 10, sizeof(double));
  const float ddelx_dx = 0.5f * W, ddely_dy = 0.5f * H;
  for (int py = 0; py < H; py++)
    for (int px = 0; px < W; px++) {
      int tile = (py / TILE) * gx + (px / TILE);
      int beg = ranges[2 * tile], end = ranges[2 * tile + 1];
      size_t pid = (size_t)py * W + px;
      const float T_final = final_T[pid];
      float T = T_final;
      int last = (int)n_contrib[pid];
      float dLdC[3] = {dL_dpix[pid], dL_dpix[(size_t)H * W + pid], dL_dpix[2 * (size_t)H * W + pid]};
      float dLdD = dL_dpix_depth[pid];
      float accum_rec[3] = {0, 0, 0}, last_color[3] = {0, 0, 0};
      float accum_rec_depth = 0, last_depth = 0, last_alpha = 0;
      float pxf = (float)px, pyf = (float)py;
      float bg_dot = 0.f;
      for (int ch = 0; ch < 3; ch++) bg_dot += bg[ch] * dLdC[ch];
      /* list position `contributor` (1-based) of entry k is k-beg+1; entries with
       * contributor-1 >= last were never reached by the forward. */
      for (int k = imin(end, beg + last) - 1; k >= beg; k--) {
        uint32_t g = point_list[k];
        float dx = means2D[2 * g] - pxf, dy = means2D[2 * g + 1] - pyf;
        const float *co = conic_opacity + 4 * g;
        float power = -0.5f * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
        if (power > 0.0f) continue;
        float G = expf(power);
        float alpha = fminf(0.99f, co[3] * G);
        if (alpha < 1.0f / 255.0f) continue;
        T = T / (1.f - alpha);
        float dchannel_dcolor = alpha * T;
        float dL_dalpha = 0.0f;
        double *a = acc + (size_t)g * 10;
        for (int ch = 0; ch < 3; ch++) {
          float c = colors[3 * g + ch];
          accum_rec[ch] = last_alpha * last_color[ch] + (1.f - last_alpha) * accum_rec[ch];
          last_color[ch] = c;
          dL_dalpha += (c - accum_rec[ch]) * dLdC[ch];
          a[6 + ch] += (double)(dchannel_dcolor * dLdC[ch]);
        }
        float depth = depths[g];
        accum_rec_depth = last_alpha * last_depth + (1.f - last_alpha) * accum_rec_depth;
        last_depth = depth;
        dL_dalpha += (depth - accum_rec_depth) * dLdD;
        a[9] += (double)(dchannel_dcolor * dLdD);
        dL_dalpha *= T;
        last_alpha = alpha;
        dL_dalpha += (-T_final / (1.f - alpha)) * bg_dot;
        float dL_dG = co[3] * dL_dalpha;
        float gdx = G * dx, gdy = G * dy;
        float dG_ddelx = -gdx * co[0] - gdy * co[1];
        float dG_ddely = -gdy * co[2] - gdx * co[1];
        a[0] += (double)(dL_dG * dG_ddelx * ddelx_dx);
        a[1] += (double)(dL_dG * dG_ddely * ddely_dy);
        a[2] += (double)(-0.5f * gdx * dx * dL_dG);
        a[3] += (double)(-0.5f * gdx * dy * dL_dG);
        a[4] += (double)(-0.5f * gdy * dy * dL_dG);
        a[5] += (double)(G * dL_dalpha);
      }
    }
  for (int g = 0; g < P; g++) {
    const double *a = acc + (size_t)g * 10;
    dL_dmean2D[3 * g] = (float)a[0]; dL_dmean2D[3 * g + 1] = (float)a[1]; dL_dmean2D[3 * g + 2] = 0.f;
    dL_dconic[4 * g] = (float)a[2]; dL_dconic[4 * g + 1] = (float)a[3]; dL_dconic[4 * g + 2] = 0.f; dL_dconic[4 * g + 3] = (float)a[4];
    dL_dopacity[g] = (float)a[5];
    dL_dcolor[3 * g] = (float)a[6]; dL_dcolor[3 * g + 1] = (float)a[7]; dL_dcolor[3 * g + 2] = (float)a[8];
    dL_ddepth[g] = (float)a[9];
  }
  free(acc);
}

static void cross3(const float *a, const float *b, float *o) {
  o[0] = a[1] * b[2] - a[2] * b[1];
  o[1] = a[2] * b[0] - a[0] * b[2];
  o[2] = a[0] * b[1] - a[1] * b[0];
}

/* d(v/|v|)/dv applied to dv (auxiliary.h:109-119). */
static void dnormvdv3(const float *v, const float *dv, float *o) {
  float sum2 = v[0] * v[0] + v[1] * v[1] + v[2] * v[2];
  float inv = 1.0f / sqrtf(sum2 * sum2 * sum2);
  o[0] = ((+sum2 - v[0] * v[0]) * dv[0] - v[1] * v[0] * dv[1] - v[2] * v[0] * dv[2]) * inv;
  o[1] = (-v[0] * v[1] * dv[0] + (sum2 - v[1] * v[1]) * dv[1] - v[2] * v[1] * dv[2]) * inv;
  o[2] = (-v[0] * v[2] * dv[0] - v[1] * v[2] * dv[1] + (sum2 - v[2] * v[2]) * dv[2]) * inv;
}

/* SH backward: dL/dsh, dL/dmean (view-direction path) and tau[0:3] -= dL/dmean (backward.cu:21-145). */
static void sh_backward(int deg, int M, const float *pos, const float *campos, const float *sh,
                        const uint8_t *clamped, const float *dL_dcolor, float *dL_dmean_acc, float *dL_dsh,
                        float *dL_dtau) {
  float dorig[3] = {pos[0] - campos[0], pos[1] - campos[1], pos[2] - campos[2]};
  float len = sqrtf(dorig[0] * dorig[0] + dorig[1] * dorig[1] + dorig[2] * dorig[2]);
  float x = dorig[0] / len, y = dorig[1] / len, z = dorig[2] / len;
  float g[3];
  for (int ch = 0; ch < 3; ch++) g[ch] = dL_dcolor[ch] * (clamped[ch] ? 0.f : 1.f);
  float dx[3] = {0, 0, 0}, dy[3] = {0, 0, 0}, dz[3] = {0, 0, 0};
  (void)M;
#define SH(k, ch) sh[(k) * 3 + (ch)]
#define OUT(k, w) for (int ch = 0; ch < 3; ch++) dL_dsh[(k) * 3 + ch] = (w) * g[ch];
  OUT(0, SH_C0)
  if (deg > 0) {
    OUT(1, -SH_C1 * y) OUT(2, SH_C1 * z) OUT(3, -SH_C1 * x)
    for (int ch = 0; ch < 3; ch++) { dx[ch] = -SH_C1 * SH(3, ch); dy[ch] = -SH_C1 * SH(1, ch); dz[ch] = SH_C1 * SH(2, ch); }
    if (deg > 1) {
      float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
      OUT(4, SH_C2[0] * xy) OUT(5, SH_C2[1] * yz) OUT(6, SH_C2[2] * (2.f * zz - xx - yy))
      OUT(7, SH_C2[3] * xz) OUT(8, SH_C2[4] * (xx - yy))
      for (int ch = 0; ch < 3; ch++) {
        dx[ch] += SH_C2[0] * y * SH(4, ch) + SH_C2[2] * 2.f * -x * SH(6, ch) + SH_C2[3] * z * SH(7, ch) + SH_C2[4] * 2.f * x * SH(8, ch);
        dy[ch] += SH_C2[0] * x * SH(4, ch) + SH_C2[1] * z * SH(5, ch) + SH_C2[2] * 2.f * -y * SH(6, ch) + SH_C2[4] * 2.f * -y * SH(8, ch);
        dz[ch] += SH_C2[1] * y * SH(5, ch) + SH_C2[2] * 2.f * 2.f * z * SH(6, ch) + SH_C2[3] * x * SH(7, ch);
      }
      if (deg > 2) {
        OUT(9, SH_C3[0] * y * (3.f * xx - yy)) OUT(10, SH_C3[1] * xy * z) OUT(11, SH_C3[2] * y * (4.f * zz - xx - yy))
        OUT(12, SH_C3[3] * z * (2.f * zz - 3.f * xx - 3.f * yy)) OUT(13, SH_C3[4] * x * (4.f * zz - xx - yy))
        OUT(14, SH_C3[5] * z * (xx - yy)) OUT(15, SH_C3[6] * x * (xx - 3.f * yy))
        for (int ch = 0; ch < 3; ch++) {
          dx[ch] += (SH_C3[0] * SH(9, ch) * 3.f * 2.f * xy + SH_C3[1] * SH(10, ch) * yz + SH_C3[2] * SH(11, ch) * -2.f * xy +
                     SH_C3[3] * SH(12, ch) * -3.f * 2.f * xz + SH_C3[4] * SH(13, ch) * (-3.f * xx + 4.f * zz - yy) +
                     SH_C3[5] * SH(14, ch) * 2.f * xz + SH_C3[6] * SH(15, ch) * 3.f * (xx - yy));
          dy[ch] += (SH_C3[0] * SH(9, ch) * 3.f * (xx - yy) + SH_C3[1] * SH(10, ch) * xz +
                     SH_C3[2] * SH(11, ch) * (-3.f * yy + 4.f * zz - xx) + SH_C3[3] * SH(12, ch) * -3.f * 2.f * yz +
                     SH_C3[4] * SH(13, ch) * -2.f * xy + SH_C3[5] * SH(14, ch) * -2.f * yz + SH_C3[6] * SH(15, ch) * -3.f * 2.f * xy);
          dz[ch] += (SH_C3[1] * SH(10, ch) * xy + SH_C3[2] * SH(11, ch) * 4.f * 2.f * yz +
                     SH_C3[3] * SH(12, ch) * 3.f * (2.f * zz - xx - yy) + SH_C3[4] * SH(13, ch) * 4.f * 2.f * xz +
                     SH_C3[5] * SH(14, ch) * (xx - yy));
        }
      }
    }
  }
#undef SH
#undef OUT
  float ddir[3] = {dx[0] * g[0] + dx[1] * g[1] + dx[2] * g[2], dy[0] * g[0] + dy[1] * g[1] + dy[2] * g[2],
                   dz[0] * g[0] + dz[1] * g[1] + dz[2] * g[2]};
  float dmean[3];
  dnormvdv3(dorig, ddir, dmean);
  for (int k = 0; k < 3; k++) { dL_dmean_acc[k] += dmean[k]; dL_dtau[k] += -dmean[k]; }
}

/* Stage 5: per-Gaussian backward -- conic -> cov2D -> (cov3D, mean3D, tau), mean2D -> (mean3D, tau),
 * depth -> (mean3D, tau), colour -> (SH, mean3D, tau), cov3D -> (scale, rot).
 * backward.cu:150-345 (computeCov2DCUDA), :494-624 (preprocessCUDA), :426-489 (computeCov3D).
 * Outputs must be zero-initialised by the caller (rows of culled Gaussians stay zero). */
void gsaj_oracle_preprocess_backward(int P, int D, int M, int W, int H, const float *means3D, const int *radii,
                                     const float *shs, const uint8_t *clamped, const float *scales,
                                     const float *rotations, float scale_modifier, const float *cov3Ds,
                                     const float *viewmatrix, const float *projmatrix, const float *projmatrix_raw,
                                     const float *campos, float tanfovx, float tanfovy, const float *dL_dmean2D,
                                     const float *dL_dconic, const float *dL_dcolor, const float *dL_ddepth,
                                     float *dL_dmean3D, float *dL_dcov3D, float *dL_dsh, float *dL_dscale,
                                     float *dL_drot, float *dL_dtau) {
  const float fy = (float)H / (2.0f * tanfovy), fx = (float)W / (2.0f * tanfovx);
  for (int i = 0; i < P; i++) {
    if (!(radii[i] > 0)) continue;
    const float *mean = means3D + 3 * i;
    float *tau = dL_dtau + 6 * i;
    /* ---- conic -> cov2D -> cov3D / T / J / t ---- */
    Cov2D cv;
    cov2d_eval(mean, fx, fy, tanfovx, tanfovy, cov3Ds + 6 * i, viewmatrix, &cv);
    const float xmul = (cv.txtz < -cv.limx || cv.txtz > cv.limx) ? 0.f : 1.f;
    const float ymul = (cv.tytz < -cv.limy || cv.tytz > cv.limy) ? 0.f : 1.f;
    float gcx = dL_dconic[4 * i], gcy = dL_dconic[4 * i + 1], gcz = dL_dconic[4 * i + 3];
    float a = cv.a, b = cv.b, c = cv.c;
    float denom = a * c - b * b;
    float dL_da = 0, dL_db = 0, dL_dc = 0;
    float denom2inv = 1.0f / ((denom * denom) + 0.0000001f);
    float (*Mx)[3] = cv.M;
    float *gcov = dL_dcov3D + 6 * i;
    if (denom2inv != 0) {
      dL_da = denom2inv * (-c * c * gcx + 2 * b * c * gcy + (denom - a * c) * gcz);
      dL_dc = denom2inv * (-a * a * gcz + 2 * a * b * gcy + (denom - a * c) * gcx);
      dL_db = denom2inv * 2 * (b * c * gcx - (denom + 2 * b * b) * gcy + a * b * gcz);
      gcov[0] = (Mx[0][0] * Mx[0][0] * dL_da + Mx[0][0] * Mx[1][0] * dL_db + Mx[1][0] * Mx[1][0] * dL_dc);
      gcov[3] = (Mx[0][1] * Mx[0][1] * dL_da + Mx[0][1] * Mx[1][1] * dL_db + Mx[1][1] * Mx[1][1] * dL_dc);
      gcov[5] = (Mx[0][2] * Mx[0][2] * dL_da + Mx[0][2] * Mx[1][2] * dL_db + Mx[1][2] * Mx[1][2] * dL_dc);
      gcov[1] = 2 * Mx[0][0] * Mx[0][1] * dL_da + (Mx[0][0] * Mx[1][1] + Mx[0][1] * Mx[1][0]) * dL_db + 2 * Mx[1][0] * Mx[1][1] * dL_dc;
      gcov[2] = 2 * Mx[0][0] * Mx[0][2] * dL_da + (Mx[0][0] * Mx[1][2] + Mx[0][2] * Mx[1][0]) * dL_db + 2 * Mx[1][0] * Mx[1][2] * dL_dc;
      gcov[4] = 2 * Mx[0][2] * Mx[0][1] * dL_da + (Mx[0][1] * Mx[1][2] + Mx[0][2] * Mx[1][1]) * dL_db + 2 * Mx[1][1] * Mx[1][2] * dL_dc;
    } else {
      for (int k = 0; k < 6; k++) gcov[k] = 0;
    }
    /* dL/dM (2x3): MV[r][k] = sum_j M[r][j] V[k][j] */
    float MV[2][3], dM[2][3];
    for (int r = 0; r < 2; r++)
      for (int k = 0; k < 3; k++) MV[r][k] = Mx[r][0] * cv.V[k][0] + Mx[r][1] * cv.V[k][1] + Mx[r][2] * cv.V[k][2];
    for (int k = 0; k < 3; k++) {
      dM[0][k] = 2 * MV[0][k] * dL_da + MV[1][k] * dL_db;
      dM[1][k] = 2 * MV[1][k] * dL_dc + MV[0][k] * dL_db;
    }
    /* M = J Rcw  ->  dL/dJ[i][j] = sum_k dM[i][k] Rcw[j][k] */
    float (*Rc)[3] = cv.Rcw;
    float dJ00 = Rc[0][0] * dM[0][0] + Rc[0][1] * dM[0][1] + Rc[0][2] * dM[0][2];
    float dJ02 = Rc[2][0] * dM[0][0] + Rc[2][1] * dM[0][1] + Rc[2][2] * dM[0][2];
    float dJ11 = Rc[1][0] * dM[1][0] + Rc[1][1] * dM[1][1] + Rc[1][2] * dM[1][2];
    float dJ12 = Rc[2][0] * dM[1][0] + Rc[2][1] * dM[1][1] + Rc[2][2] * dM[1][2];
    float tz = 1.f / cv.t[2], tz2 = tz * tz, tz3 = tz2 * tz;
    float gt[3];
    gt[0] = xmul * -fx * tz2 * dJ02;
    gt[1] = ymul * -fy * tz2 * dJ12;
    gt[2] = -fx * tz2 * dJ00 - fy * tz2 * dJ11 + (2 * fx * cv.t[0]) * tz3 * dJ02 + (2 * fy * cv.t[1]) * tz3 * dJ12;
    /* tau: rho += g, theta += t x g  with the CLAMPED t (backward.cu:275-290) */
    float txg[3];
    cross3(cv.t, gt, txg);
    for (int k = 0; k < 3; k++) { tau[k] += gt[k]; tau[3 + k] += txg[k]; }
    /* mean3D (covariance part) = Rcw^T g; assignment, not += (backward.cu:300) */
    float *gm = dL_dmean3D + 3 * i;
    for (int k = 0; k < 3; k++) gm[k] = Rc[0][k] * gt[0] + Rc[1][k] * gt[1] + Rc[2][k] * gt[2];
    /* dL/dRcw[j][k] = sum_i J[i][j] dM[i][k]; theta += sum_k col_k(Rcw) x col_k(dL/dRcw) (backward.cu:301-345) */
    float dR[3][3];
    for (int k = 0; k < 3; k++) {
      dR[0][k] = cv.J00 * dM[0][k];
      dR[1][k] = cv.J11 * dM[1][k];
      dR[2][k] = cv.J02 * dM[0][k] + cv.J12 * dM[1][k];
    }
    float th[3] = {0, 0, 0};
    for (int k = 0; k < 3; k++) {
      float ck[3] = {Rc[0][k], Rc[1][k], Rc[2][k]}, gk[3] = {dR[0][k], dR[1][k], dR[2][k]}, cr[3];
      cross3(ck, gk, cr);
      /* reference evaluates dot(g_k, (-[c_k]x).col_m): same value, keep its sum order over k */
      th[0] += cr[0]; th[1] += cr[1]; th[2] += cr[2];
    }
    for (int k = 0; k < 3; k++) tau[3 + k] += th[k];

    /* ---- mean2D -> mean3D and tau (backward.cu:512-597) ---- */
    float mh[4];
    xform4x4(projmatrix, mean, mh);
    float mw = 1.0f / (mh[3] + 0.0000001f);
    const float *pj = projmatrix;
    float g2x = dL_dmean2D[3 * i], g2y = dL_dmean2D[3 * i + 1];
    float mul1 = (pj[0] * mean[0] + pj[4] * mean[1] + pj[8] * mean[2] + pj[12]) * mw * mw;
    float mul2 = (pj[1] * mean[0] + pj[5] * mean[1] + pj[9] * mean[2] + pj[13]) * mw * mw;
    gm[0] += (pj[0] * mw - pj[3] * mul1) * g2x + (pj[1] * mw - pj[3] * mul2) * g2y;
    gm[1] += (pj[4] * mw - pj[7] * mul1) * g2x + (pj[5] * mw - pj[7] * mul2) * g2y;
    gm[2] += (pj[8] * mw - pj[11] * mul1) * g2x + (pj[9] * mw - pj[11] * mul2) * g2y;
    float alpha_ = 1.0f * mw, beta_ = -mh[0] * mw * mw, gamma_ = -mh[1] * mw * mw;
    float pa = projmatrix_raw[0], pb = projmatrix_raw[5], pe = projmatrix_raw[11];
    float pC[3];
    xform4x3(viewmatrix, mean, pC); /* un-clamped camera-space point */
    float d1[3] = {alpha_ * pa, 0.f, beta_ * pe}, d2[3] = {0.f, alpha_ * pb, gamma_ * pe};
    float c1[3], c2[3];
    cross3(pC, d1, c1); /* (-[p]x)^T d = p x d */
    cross3(pC, d2, c2);
    for (int k = 0; k < 3; k++) {
      tau[k] += g2x * d1[k] + g2y * d2[k];
      tau[3 + k] += g2x * c1[k] + g2y * c2[k];
    }
    /* ---- depth -> mean3D and tau (backward.cu:599-613): dz/dtau = [0,0,1, y, -x, 0] ---- */
    float gz = dL_ddepth[i];
    gm[0] += gz * viewmatrix[2]; gm[1] += gz * viewmatrix[6]; gm[2] += gz * viewmatrix[10];
    tau[2] += gz * 1.f;
    tau[3] += gz * pC[1];
    tau[4] += gz * -pC[0];
    tau[5] += gz * 0.f;
    /* ---- colour -> SH, view direction -> mean3D and tau ---- */
    if (shs) sh_backward(D, M, mean, campos, shs + (size_t)i * M * 3, clamped + 3 * i, dL_dcolor + 3 * i, gm,
                         dL_dsh + (size_t)i * M * 3, tau);
    /* ---- cov3D -> scale, rotation (backward.cu:426-489) ---- */
    if (scales) {
      float R[3][3], s[3] = {scale_modifier * scales[3 * i], scale_modifier * scales[3 * i + 1], scale_modifier * scales[3 * i + 2]};
      const float *q = rotations + 4 * i;
      quat_to_R(q, R);
      float dS[3][3] = {{gcov[0], 0.5f * gcov[1], 0.5f * gcov[2]},
                        {0.5f * gcov[1], gcov[3], 0.5f * gcov[4]},
                        {0.5f * gcov[2], 0.5f * gcov[4], gcov[5]}};
      /* A = S R^T (A[i][j] = s_i R[j][i]); dL/dA = 2 A dSigma */
      float dA[3][3];
      for (int r = 0; r < 3; r++)
        for (int cc = 0; cc < 3; cc++) {
          float v = 0;
          for (int k = 0; k < 3; k++) v += (s[r] * R[k][r]) * dS[k][cc];
          dA[r][cc] = 2.0f * v;
        }
      for (int k = 0; k < 3; k++) dL_dscale[3 * i + k] = R[0][k] * dA[k][0] + R[1][k] * dA[k][1] + R[2][k] * dA[k][2];
      /* dL/dR[j][i] = s_i dA[i][j] */
      float g[3][3];
      for (int ii = 0; ii < 3; ii++)
        for (int j = 0; j < 3; j++) g[j][ii] = s[ii] * dA[ii][j];
      float r = q[0], x = q[1], y = q[2], z = q[3];
      dL_drot[4 * i + 0] = 2 * z * (g[1][0] - g[0][1]) + 2 * y * (g[0][2] - g[2][0]) + 2 * x * (g[2][1] - g[1][2]);
      dL_drot[4 * i + 1] = 2 * y * (g[0][1] + g[1][0]) + 2 * z * (g[0][2] + g[2][0]) + 2 * r * (g[2][1] - g[1][2]) - 4 * x * (g[2][2] + g[1][1]);
      dL_drot[4 * i + 2] = 2 * x * (g[0][1] + g[1][0]) + 2 * r * (g[0][2] - g[2][0]) + 2 * z * (g[2][1] + g[1][2]) - 4 * y * (g[2][2] + g[0][0]);
      dL_drot[4 * i + 3] = 2 * r * (g[1][0] - g[0][1]) + 2 * x * (g[0][2] + g[2][0]) + 2 * y * (g[2][1] + g[1][2]) - 4 * z * (g[1][1] + g[0][0]);
    }
  }
}
