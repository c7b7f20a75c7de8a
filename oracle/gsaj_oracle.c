/*
 * gsaj_oracle.c -- CPU ORACLE (TEST INFRASTRUCTURE ONLY, never a product path).
 *
 * A plain-C, fp32 restatement of the *rasteriser semantics* of the
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
 * Threads: the per-Gaussian loops and the per-tile loops are OpenMP-parallel (tiles are independent); the number
 * of threads is set with gsaj_oracle_set_threads (default 1).  Results do not depend on the thread count: every
 * floating-point sum has a fixed order (per-pixel sums inside one tile by one thread; a Gaussian's per-tile partials
 * are added in sorted-instance order by a serial pass), integer counters use atomic increments.
 *
 * Build: gcc -O2 -ffp-contract=off -fopenmp -fPIC -shared (see oracle/Makefile).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define TILE 16

static int g_threads = 1;
/* Number of OpenMP threads of the parallel loops (test / benchmark harness knob; 0 = all cores). Returns the value in use. */
int gsaj_oracle_set_threads(int n) {
#ifdef _OPENMP
  if (n <= 0) n = omp_get_num_procs();
  g_threads = n;
#else
  (void)n;
  g_threads = 1;
#endif
  return g_threads;
}

static const float SH_C0 = 0.28209479177387814f;
static const float SH_C1 = 0.4886025119029199f;
static const float SH_C2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f,
                               -1.0925484305920792f, 0.5462742152960396f};
static const float SH_C3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f,
                               0.3731763325901154f,  -0.4570457994644658f, 1.445305721320277f,
                               -0.5900435899266435f};

/* The small-matrix helpers, the shared projection step (cov2d_eval) and the whole per-Gaussian backward live in
 * chain_body.inc, instantiated twice: in fp32 (the restatement of the reference's arithmetic; un-suffixed names) and in
 * fp64 (suffix _f64; same operations, used by the parity tests as the "truth" that tells rounding noise of ANY fp32
 * evaluation from a wrong formula). */
#define REAL float
#define RN(x) x
#define R_SQRT sqrtf
#define R_FMIN fminf
#define R_FMAX fmaxf
#include "chain_body.inc"
#undef REAL
#undef RN
#undef R_SQRT
#undef R_FMIN
#undef R_FMAX
#define REAL double
#define RN(x) x##_f64
#define R_SQRT sqrt
#define R_FMIN fmin
#define R_FMAX fmax
#include "chain_body.inc"
#undef REAL
#undef RN
#undef R_SQRT
#undef R_FMIN
#undef R_FMAX

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
  int culled_prefiltered = 0;
#pragma omp parallel for num_threads(g_threads) schedule(static) reduction(+ : R) reduction(| : culled_prefiltered)
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
      if (prefiltered) culled_prefiltered |= 1;
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
  if (culled_prefiltered) return -1;
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
 * point_list[R]: Gaussian ids; ranges[2*tiles]: [start,end). rasterizer_impl.cu:70-138,339-368.
 * The stable radix sort of the 64-bit keys is evaluated as: stable counting sort by tile id (keeps emission order),
 * then a stable sort by depth bits inside every tile (ties keep emission order = Gaussian index order) -- the same
 * permutation, and the tiles sort in parallel. */
typedef struct { uint64_t key; uint32_t val; uint32_t seq; } KV;
static int kv_cmp(const void *a, const void *b) {
  const KV *x = (const KV *)a, *y = (const KV *)b;
  if (x->key != y->key) return x->key < y->key ? -1 : 1;
  return x->seq < y->seq ? -1 : (x->seq > y->seq ? 1 : 0); /* stable: emission order */
}
int gsaj_oracle_bin(int P, int W, int H, int R, const int *radii, const float *means2D, const float *depths,
                    /* out */ uint32_t *point_list, uint64_t *keys_sorted, int *ranges) {
  const int gx = (W + TILE - 1) / TILE, gy = (H + TILE - 1) / TILE, tiles = gx * gy;
  KV *kv = (KV *)malloc(sizeof(KV) * (size_t)(R > 0 ? R : 1));
  uint32_t *start = (uint32_t *)calloc((size_t)tiles + 1, sizeof(uint32_t));
  uint32_t *cursor = (uint32_t *)malloc(sizeof(uint32_t) * ((size_t)tiles + 1));
  if (!kv || !start || !cursor) { free(kv); free(start); free(cursor); return -2; }
  long total = 0;
  for (int i = 0; i < P; i++) {
    if (radii[i] <= 0) continue;
    int x0, y0, x1, y1;
    tile_rect(means2D[2 * i], means2D[2 * i + 1], radii[i], gx, gy, &x0, &y0, &x1, &y1);
    for (int y = y0; y < y1; y++)
      for (int x = x0; x < x1; x++) { start[y * gx + x + 1]++; total++; }
  }
  if (total != (long)R) { free(kv); free(start); free(cursor); return -3; }
  for (int t = 0; t < tiles; t++) start[t + 1] += start[t];
  memcpy(cursor, start, sizeof(uint32_t) * ((size_t)tiles + 1));
  uint32_t off = 0;
  for (int i = 0; i < P; i++) {
    if (radii[i] <= 0) continue;
    int x0, y0, x1, y1;
    tile_rect(means2D[2 * i], means2D[2 * i + 1], radii[i], gx, gy, &x0, &y0, &x1, &y1);
    uint32_t dbits;
    memcpy(&dbits, depths + i, 4);
    for (int y = y0; y < y1; y++)
      for (int x = x0; x < x1; x++) {
        const uint32_t t = (uint32_t)(y * gx + x), k = cursor[t]++;
        kv[k].key = ((uint64_t)t << 32) | dbits;
        kv[k].val = (uint32_t)i; kv[k].seq = off++;
      }
  }
#pragma omp parallel for num_threads(g_threads) schedule(dynamic, 8)
  for (int t = 0; t < tiles; t++) {
    const uint32_t b = start[t], e = start[t + 1];
    if (e > b + 1) qsort(kv + b, (size_t)(e - b), sizeof(KV), kv_cmp);
    ranges[2 * t] = e > b ? (int)b : 0;
    ranges[2 * t + 1] = e > b ? (int)e : 0;
    for (uint32_t k = b; k < e; k++) {
      point_list[k] = kv[k].val;
      if (keys_sorted) keys_sorted[k] = kv[k].key;
    }
  }
  free(kv); free(start); free(cursor);
  return 0;
}

/* Stage 3: per-pixel front-to-back compositing (forward.cu:406-535).
 * out_color [3,H,W], out_depth [H,W], out_opacity [H,W], final_T [H,W], n_contrib [H,W],
 * n_touched [P] (must be zeroed by caller). Returns sum over pixels of n_contrib
 * (= the interaction count I of SURVEY 8d).  Tiles are independent (one thread per tile at a time). */
long gsaj_oracle_render(int W, int H, const int *ranges, const uint32_t *point_list, const float *means2D,
                        const float *features, const float *conic_opacity, const float *depths, const float *bg,
                        float *out_color, float *out_depth, float *out_opacity, float *final_T,
                        uint32_t *n_contrib, int *n_touched) {
  const int gx = (W + TILE - 1) / TILE, gy = (H + TILE - 1) / TILE;
  long interactions = 0;
#pragma omp parallel for num_threads(g_threads) schedule(dynamic, 4) reduction(+ : interactions)
  for (int tile = 0; tile < gx * gy; tile++) {
    const int ty = tile / gx, tx = tile - ty * gx;
    const int beg = ranges[2 * tile], end = ranges[2 * tile + 1];
    for (int py = ty * TILE; py < imin(H, (ty + 1) * TILE); py++)
      for (int px = tx * TILE; px < imin(W, (tx + 1) * TILE); px++) {
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
          if (test_T > 0.5f) {
#pragma omp atomic
            n_touched[g]++;
          }
          T = test_T;
          last = contributor;
        }
        size_t pid = (size_t)py * W + px;
        final_T[pid] = T; n_contrib[pid] = last;
        for (int ch = 0; ch < 3; ch++) out_color[(size_t)ch * H * W + pid] = C[ch] + T * bg[ch];
        out_depth[pid] = Dp; out_opacity[pid] = 1 - T;
        interactions += last;
      }
  }
  return interactions;
}

/* Stage 4: reverse compositor (backward.cu:648-872).  The reference adds every pixel's 10 partials to the Gaussian's
 * row with float atomics (no defined order).  Here every (tile, Gaussian) instance first gets its own double-precision
 * partial sums (pixels of the tile in row-major order), then a Gaussian's instances are added in sorted-instance order
 * and rounded once.
 * dL_dmean2D [P,3] (z unused), dL_dconic [P,4] (slots 0,1,3), dL_dopacity [P], dL_dcolor [P,3], dL_ddepth [P]. */
void gsaj_oracle_render_backward(int P, int W, int H, const int *ranges, const uint32_t *point_list,
                                 const float *means2D, const float *conic_opacity, const float *colors,
                                 const float *depths, const float *bg, const float *final_T,
                                 const uint32_t *n_contrib, const float *dL_dpix, const float *dL_dpix_depth,
                                 float *dL_dmean2D, float *dL_dconic, float *dL_dopacity, float *dL_dcolor,
                                 float *dL_ddepth) {
  const int gx = (W + TILE - 1) / TILE, gy = (H + TILE - 1) / TILE;
  int R = 0;
  for (int t = 0; t < gx * gy; t++) R = imax(R, ranges[2 * t + 1]);
  double *inst = (double *)calloc((size_t)(R > 0 ? R : 1) * 10, sizeof(double));
  double *acc = (double *)calloc((size_t)P * 10, sizeof(double));
  const float ddelx_dx = 0.5f * W, ddely_dy = 0.5f * H;
#pragma omp parallel for num_threads(g_threads) schedule(dynamic, 4)
  for (int tile = 0; tile < gx * gy; tile++) {
    const int ty = tile / gx, tx = tile - ty * gx;
    const int beg = ranges[2 * tile], end = ranges[2 * tile + 1];
    for (int py = ty * TILE; py < imin(H, (ty + 1) * TILE); py++)
      for (int px = tx * TILE; px < imin(W, (tx + 1) * TILE); px++) {
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
          double *a = inst + (size_t)k * 10;
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
  }
  for (int k = 0; k < R; k++) { /* serial: a Gaussian's instances in sorted-instance order */
    double *a = acc + (size_t)point_list[k] * 10;
    const double *b = inst + (size_t)k * 10;
    for (int c = 0; c < 10; c++) a[c] += b[c];
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
  free(inst);
}

/* ------------------------------------------------------------------------------------------------------------
 * Error model of the reverse compositor's per-Gaussian sums (TEST ANALYSIS, not part of the restated algorithm).
 *
 * Two correct fp32 implementations of backward.cu:648-872 (this oracle, the reference's CUDA kernel, the HIP kernels)
 * differ in three ways, each bounded here per Gaussian and per output component c (order: mean2D x,y, conic a,b,c,
 * opacity, colour r,g,b, depth):
 *   term_mass[g,c]   = sum over pixels of |term|: a different summation order / accumulator width moves the sum by at
 *                      most (a small multiple of eps) * term_mass;
 *   cond_slack[g,c]  = sum over pixels of |term| * eps * (1 + mag + n), mag = (|a| dx^2 + |c| dy^2)/2 + |b dx dy|: power =
 *                      -(a dx^2 + c dy^2)/2 - b dx dy is a difference of products that cancel for elongated, rotated
 *                      Gaussians, so ANY fp32 evaluation of alpha = o exp(power) has a relative error of a few eps * mag;
 *                      n = how many contributors the walk has already un-blended at that pixel: T is recovered by
 *                      repeated division T <- T / (1 - alpha) (backward.cu:779), each adding a rounding of its own;
 *                      plus, for the six terms that carry dL/dalpha, (factor) * eps * (1 + n) * T * sum over the channels and
 *                      the depth of (|c| + |accum_rec|) |dL/dC|: dL/dalpha is a sum of DIFFERENCES c - accum_rec
 *                      (backward.cu:799-823), which behind nearly opaque layers of the same colour cancel to a small
 *                      fraction of their operands -- every fp32 evaluation, this oracle's float recurrences included, rounds
 *                      them at the size of the operands (a map painted in ONE colour: tools/fuzz_uniform.py);
 *   flip_budget[g,c] = sum over BORDERLINE pixels of |term(all borderline decisions taken one way) - term(taken the other
 *                      way)|, where a pixel is borderline if some entry's alpha is within (border_rel + 4 eps mag) of 1/255,
 *                      T(1-alpha) within the accumulated relative uncertainty of 1e-4, or power within its rounding of 0:
 *                      there the cut-offs of forward.cu:406-535 may legitimately fall either way, and the whole pixel's
 *                      backward (this entry, the nearer ones through accum_rec, the farther ones through T, the last
 *                      contributor and T_final) is re-evaluated, forward and backward, with the per-entry decisions and the
 *                      termination decision each taken both ways (four outcomes); the budget is the spread of the term.
 *   border_mask[pix] = 1 for those pixels (the image / n_contrib checks allow differences only there).
 * All arrays are outputs; P-major [P,10] floats, border_mask [H*W] bytes. */
typedef struct {
  const uint32_t *point_list; const float *means2D, *conic_opacity, *colors, *depths, *bg;
  float ddelx_dx, ddely_dy, border_rel, border_rel_T;
} EMScene;

/* forward walk of one pixel with borderline decisions biased: `bias` for the per-entry tests (power <= 0, alpha >= 1/255: +1
 * include, -1 exclude), `bias_stop` for the termination test (+1 continue, -1 stop), 0 as computed.  The two kinds are biased
 * SEPARATELY: an entry at the alpha cut-off moves T by its (1 - alpha), i.e. by 0.4 %, and that can carry a termination test
 * hundreds of entries later across 1e-4 -- "entry excluded, walk continued" is an outcome neither all-included nor
 * all-excluded reaches (tests/test_gpu_random.py fuzz seed 8345).  Returns 1 if a borderline decision was met. */
static int em_forward(const EMScene *sc, int beg, int end, float pxf, float pyf, int bias, int bias_stop, float *T_out, int *last_out) {
  const double eps = 1.1920929e-07, thr = 1.0 / 255.0;
  float T = 1.0f;
  int last = 0, border = 0;
  double t_rel = sc->border_rel_T;
  for (int k = beg; k < end; k++) {
    uint32_t g = sc->point_list[k];
    float dx = sc->means2D[2 * g] - pxf, dy = sc->means2D[2 * g + 1] - pyf;
    const float *co = sc->conic_opacity + 4 * g;
    float power = -0.5f * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
    double mag = 0.5 * (fabs((double)co[0]) * dx * dx + fabs((double)co[2]) * dy * dy) + fabs((double)co[1] * dx * dy);
    double a_rel = sc->border_rel + 4.0 * eps * mag;
    int skip = power > 0.0f;
    if (fabs((double)power) <= 4.0 * eps * mag + 1e-7) { border = 1; if (bias) skip = bias < 0; }
    if (skip) continue;
    float alpha = fminf(0.99f, co[3] * expf(fminf(power, 0.0f)));
    int low = alpha < 1.0f / 255.0f;
    if (fabs((double)alpha - thr) <= a_rel * thr) { border = 1; if (bias) low = bias < 0; }
    if (low) continue;
    float test_T = T * (1 - alpha);
    t_rel += a_rel * alpha / (1.0 - alpha);
    int stop = test_T < 0.0001f;
    if (fabs((double)test_T - 1e-4) <= t_rel * 1e-4) { border = 1; if (bias_stop) stop = bias_stop < 0; }
    if (stop) break;
    T = test_T;
    last = k - beg + 1;
  }
  *T_out = T; *last_out = last;
  return border;
}

/* backward walk of one pixel under the same bias; adds the 10 terms of every visited entry k to v[(k-beg)*10 + c]
 * and, if m / cs are given, |term| and |term| * eps * mag. */
static void em_backward(const EMScene *sc, int beg, int end, float pxf, float pyf, int bias, float T_final, int last,
                        const float *dLdC, float dLdD, double *v, double *m, double *cs) {
  const double eps = 1.1920929e-07, thr = 1.0 / 255.0;
  float T = T_final, accum_rec[3] = {0, 0, 0}, last_color[3] = {0, 0, 0};
  float accum_rec_depth = 0, last_depth = 0, last_alpha = 0, bg_dot = 0.f;
  double ndiv = 0.0;
  for (int ch = 0; ch < 3; ch++) bg_dot += sc->bg[ch] * dLdC[ch];
  for (int k = imin(end, beg + last) - 1; k >= beg; k--) {
    uint32_t g = sc->point_list[k];
    float dx = sc->means2D[2 * g] - pxf, dy = sc->means2D[2 * g + 1] - pyf;
    const float *co = sc->conic_opacity + 4 * g;
    float power = -0.5f * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
    double mag = 0.5 * (fabs((double)co[0]) * dx * dx + fabs((double)co[2]) * dy * dy) + fabs((double)co[1] * dx * dy);
    double a_rel = sc->border_rel + 4.0 * eps * mag;
    int skip = power > 0.0f;
    if (bias && fabs((double)power) <= 4.0 * eps * mag + 1e-7) skip = bias < 0;
    if (skip) continue;
    float G = expf(fminf(power, 0.0f));
    float alpha = fminf(0.99f, co[3] * G);
    int low = alpha < 1.0f / 255.0f;
    if (bias && fabs((double)alpha - thr) <= a_rel * thr) low = bias < 0;
    if (low) continue;
    T = T / (1.f - alpha);
    ndiv += 1.0;
    float dchannel_dcolor = alpha * T, dL_dalpha = 0.0f;
    double t[10], operands = 0.0;  /* sum |c| |dL/dC| + |accum_rec| |dL/dC| over the channels and the depth: what (c - accum_rec) rounds at */
    for (int ch = 0; ch < 3; ch++) {
      float c = sc->colors[3 * g + ch];
      accum_rec[ch] = last_alpha * last_color[ch] + (1.f - last_alpha) * accum_rec[ch];
      last_color[ch] = c;
      dL_dalpha += (c - accum_rec[ch]) * dLdC[ch];
      operands += (fabs((double)c) + fabs((double)accum_rec[ch])) * fabs((double)dLdC[ch]);
      t[6 + ch] = (double)(dchannel_dcolor * dLdC[ch]);
    }
    float depth = sc->depths[g];
    accum_rec_depth = last_alpha * last_depth + (1.f - last_alpha) * accum_rec_depth;
    last_depth = depth;
    dL_dalpha += (depth - accum_rec_depth) * dLdD;
    operands += (fabs((double)depth) + fabs((double)accum_rec_depth)) * fabs((double)dLdD);
    t[9] = (double)(dchannel_dcolor * dLdD);
    dL_dalpha *= T;
    /* |rounding of dL/dalpha| that is NOT proportional to |dL/dalpha|: c - accum_rec is a difference (behind nearly opaque layers
     * of the same colour it cancels to a small fraction of its operands), accum_rec carries the roundings of the ndiv steps of
     * its recurrence, and the background term is added to the sum */
    const double dalpha_abs = eps * (1.0 + ndiv) * (operands * (double)T + fabs((double)((-T_final / (1.f - alpha)) * bg_dot)));
    last_alpha = alpha;
    dL_dalpha += (-T_final / (1.f - alpha)) * bg_dot;
    float dL_dG = co[3] * dL_dalpha, gdx = G * dx, gdy = G * dy;
    t[0] = (double)(dL_dG * (-gdx * co[0] - gdy * co[1]) * sc->ddelx_dx);
    t[1] = (double)(dL_dG * (-gdy * co[2] - gdx * co[1]) * sc->ddely_dy);
    t[2] = (double)(-0.5f * gdx * dx * dL_dG);
    t[3] = (double)(-0.5f * gdx * dy * dL_dG);
    t[4] = (double)(-0.5f * gdy * dy * dL_dG);
    t[5] = (double)(G * dL_dalpha);
    double *vk = v + (size_t)(k - beg) * 10;
    for (int c = 0; c < 10; c++) vk[c] += t[c];
    if (m) {
      double *mk = m + (size_t)(k - beg) * 10, *ck = cs + (size_t)(k - beg) * 10;
      const double adG = fabs((double)dL_dG);
      double a[10];
      a[0] = adG * (fabs((double)(gdx * co[0])) + fabs((double)(gdy * co[1]))) * sc->ddelx_dx;  /* a sum of two products */
      a[1] = adG * (fabs((double)(gdy * co[2])) + fabs((double)(gdx * co[1]))) * sc->ddely_dy;
      for (int c = 2; c < 10; c++) a[c] = fabs(t[c]);
      for (int c = 0; c < 10; c++) { mk[c] += a[c]; ck[c] += a[c] * eps * (1.0 + mag + ndiv); }
      /* terms 0..5 are dL/dalpha times a factor: its absolute rounding (above) times that factor */
      const double o = fabs((double)co[3]);
      ck[0] += dalpha_abs * o * (fabs((double)(gdx * co[0])) + fabs((double)(gdy * co[1]))) * sc->ddelx_dx;
      ck[1] += dalpha_abs * o * (fabs((double)(gdy * co[2])) + fabs((double)(gdx * co[1]))) * sc->ddely_dy;
      ck[2] += dalpha_abs * o * 0.5 * fabs((double)(gdx * dx));
      ck[3] += dalpha_abs * o * 0.5 * fabs((double)(gdx * dy));
      ck[4] += dalpha_abs * o * 0.5 * fabs((double)(gdy * dy));
      ck[5] += dalpha_abs * (double)G;
    }
  }
}

void gsaj_oracle_error_model(int P, int W, int H, const int *ranges, const uint32_t *point_list, const float *means2D,
                             const float *conic_opacity, const float *colors, const float *depths, const float *bg,
                             const float *dL_dpix, const float *dL_dpix_depth, float border_rel, float border_rel_T,
                             float *term_mass, float *cond_slack, float *flip_budget, uint8_t *border_mask) {
  const int gx = (W + TILE - 1) / TILE, gy = (H + TILE - 1) / TILE;
  int R = 0;
  for (int t = 0; t < gx * gy; t++) R = imax(R, ranges[2 * t + 1]);
  const size_t Rn = (size_t)(R > 0 ? R : 1) * 10;
  double *im = (double *)calloc(Rn, sizeof(double)), *ic = (double *)calloc(Rn, sizeof(double));
  double *ifl = (double *)calloc(Rn, sizeof(double));
  EMScene sc = {point_list, means2D, conic_opacity, colors, depths, bg, 0.5f * W, 0.5f * H, border_rel, border_rel_T};
#pragma omp parallel for num_threads(g_threads) schedule(dynamic, 4)
  for (int tile = 0; tile < gx * gy; tile++) {
    const int ty = tile / gx, tx = tile - ty * gx;
    const int beg = ranges[2 * tile], end = ranges[2 * tile + 1], n = end - beg;
    double *scratch = (double *)malloc(sizeof(double) * 10 * (size_t)(n > 0 ? n : 1) * 4);
    double *v0 = scratch, *vp = scratch + 10 * (size_t)(n > 0 ? n : 1), *vm = vp + 10 * (size_t)(n > 0 ? n : 1);
    for (int py = ty * TILE; py < imin(H, (ty + 1) * TILE); py++)
      for (int px = tx * TILE; px < imin(W, (tx + 1) * TILE); px++) {
        size_t pid = (size_t)py * W + px;
        float dLdC[3] = {dL_dpix[pid], dL_dpix[(size_t)H * W + pid], dL_dpix[2 * (size_t)H * W + pid]};
        float dLdD = dL_dpix_depth[pid], Tf;
        int last;
        const int border = em_forward(&sc, beg, end, (float)px, (float)py, 0, 0, &Tf, &last);
        border_mask[pid] = (uint8_t)border;
        if (n <= 0) continue;
        memset(v0, 0, sizeof(double) * 10 * (size_t)n);
        em_backward(&sc, beg, end, (float)px, (float)py, 0, Tf, last, dLdC, dLdD, v0, im + (size_t)beg * 10, ic + (size_t)beg * 10);
        if (border) {
          /* the four outcomes (entries included / excluded) x (walk continued / stopped): per term, the spread of the five
           * evaluations (running minimum and maximum in vp / vm) */
          double *vc = scratch + 10 * (size_t)n * 3;
          memcpy(vp, v0, sizeof(double) * 10 * (size_t)n);
          memcpy(vm, v0, sizeof(double) * 10 * (size_t)n);
          for (int combo = 0; combo < 4; combo++) {
            const int ba = (combo & 1) ? +1 : -1, bs = (combo & 2) ? +1 : -1;
            float Tc;
            int lc;
            memset(vc, 0, sizeof(double) * 10 * (size_t)n);
            em_forward(&sc, beg, end, (float)px, (float)py, ba, bs, &Tc, &lc);
            em_backward(&sc, beg, end, (float)px, (float)py, ba, Tc, lc, dLdC, dLdD, vc, NULL, NULL);
            for (size_t i = 0; i < (size_t)n * 10; i++) { vp[i] = fmax(vp[i], vc[i]); vm[i] = fmin(vm[i], vc[i]); }
          }
          for (size_t i = 0; i < (size_t)n * 10; i++) ifl[(size_t)beg * 10 + i] += vp[i] - vm[i];
        }
      }
    free(scratch);
  }
  double *acc = (double *)calloc((size_t)P * 30, sizeof(double));
  for (int k = 0; k < R; k++) {
    const size_t g = point_list[k];
    for (int c = 0; c < 10; c++) {
      acc[g * 30 + c] += im[(size_t)k * 10 + c];
      acc[g * 30 + 10 + c] += ic[(size_t)k * 10 + c];
      acc[g * 30 + 20 + c] += ifl[(size_t)k * 10 + c];
    }
  }
  for (size_t g = 0; g < (size_t)P; g++)
    for (int c = 0; c < 10; c++) {
      term_mass[g * 10 + c] = (float)acc[g * 30 + c];
      cond_slack[g * 10 + c] = (float)acc[g * 30 + 10 + c];
      flip_budget[g * 10 + c] = (float)acc[g * 30 + 20 + c];
    }
  free(acc); free(im); free(ic); free(ifl);
}

