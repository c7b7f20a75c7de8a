/* A consumer of include/gsaj.h written in plain C: what a maintainer's binding does, without Python or torch.
 *
 *   c_abi_frame <scene.bin> <out.bin>
 *
 * scene.bin (little endian, written by tests/test_gpu_c_abi_consumer.py):
 *   int32  P, D, M, W, H;  float32 tanfovx, tanfovy, bg[3];
 *   float32 viewmatrix[16], projmatrix[16], projmatrix_raw[16], campos[3];
 *   float32 means3D[P*3], opacities[P], scales[P*3], rotations[P*4], shs[P*M*3];
 *   float32 dL_dpix[3*H*W], dL_dpix_depth[H*W]
 * out.bin:
 *   int32 num_rendered; float32 color[3*H*W], depth[H*W], opacity[H*W]; int32 radii[P], n_touched[P];
 *   float32 dL_dmean3D[P*3], dL_dtau_sum[6]
 *
 * One synchronous forward (gsaj_rasterize_forward: it reports the R the binning workspace must hold, like the reference's
 * resize callbacks do, rasterize_points.cu:27-33), one backward, every buffer owned by this program (hipMalloc).
 * Build: gcc -std=c11 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -I../include c_abi_frame.c -L<lib dir> -lgsaj_hip -L/opt/rocm/lib -lamdhip64
 */
#include <hip/hip_runtime_api.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "gsaj.h"

#define HIP(x)                                                                              \
  do {                                                                                      \
    hipError_t e_ = (x);                                                                    \
    if (e_ != hipSuccess) {                                                                 \
      fprintf(stderr, "%s:%d: %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_));  \
      exit(3);                                                                              \
    }                                                                                       \
  } while (0)
#define GSAJ(x)                                                                       \
  do {                                                                                \
    int r_ = (x);                                                                     \
    if (r_ < 0) {                                                                     \
      fprintf(stderr, "%s:%d: %s -> %d: %s\n", __FILE__, __LINE__, #x, r_, gsaj_last_error()); \
      exit(4);                                                                        \
    }                                                                                 \
  } while (0)

static void *rd(FILE *f, size_t bytes) {
  void *p = malloc(bytes ? bytes : 1);
  if (!p || fread(p, 1, bytes, f) != bytes) {
    fprintf(stderr, "short read\n");
    exit(2);
  }
  return p;
}
static void *to_dev(const void *host, size_t bytes) {
  void *d = NULL;
  HIP(hipMalloc(&d, bytes ? bytes : 4));
  if (host) HIP(hipMemcpy(d, host, bytes, hipMemcpyHostToDevice));
  else HIP(hipMemset(d, 0, bytes ? bytes : 4));
  return d;
}
static void wr(FILE *f, const void *dev, size_t bytes) {
  void *h = malloc(bytes ? bytes : 1);
  HIP(hipMemcpy(h, dev, bytes, hipMemcpyDeviceToHost));
  fwrite(h, 1, bytes, f);
  free(h);
}

int main(int argc, char **argv) {
  if (argc != 3) {
    fprintf(stderr, "usage: %s scene.bin out.bin\n", argv[0]);
    return 1;
  }
  FILE *f = fopen(argv[1], "rb");
  if (!f) return 2;
  int hdr[5];
  float cam[5];
  if (fread(hdr, 4, 5, f) != 5 || fread(cam, 4, 5, f) != 5) return 2;
  const int P = hdr[0], D = hdr[1], M = hdr[2], W = hdr[3], H = hdr[4];
  const float tanfovx = cam[0], tanfovy = cam[1];
  const size_t HW = (size_t)H * W, F = sizeof(float);
  float *bg = to_dev(cam + 2, 3 * F);
  float *view = to_dev(rd(f, 16 * F), 16 * F), *proj = to_dev(rd(f, 16 * F), 16 * F), *praw = to_dev(rd(f, 16 * F), 16 * F);
  float *campos = to_dev(rd(f, 3 * F), 3 * F);
  float *means = to_dev(rd(f, P * 3 * F), P * 3 * F), *opac = to_dev(rd(f, P * F), P * F);
  float *scales = to_dev(rd(f, P * 3 * F), P * 3 * F), *rots = to_dev(rd(f, P * 4 * F), P * 4 * F);
  float *shs = to_dev(rd(f, (size_t)P * M * 3 * F), (size_t)P * M * 3 * F);
  float *dLc = to_dev(rd(f, 3 * HW * F), 3 * HW * F), *dLd = to_dev(rd(f, HW * F), HW * F);
  fclose(f);

  /* outputs and workspaces: sizes come from the library, memory from the caller */
  float *color = to_dev(NULL, 3 * HW * F), *depth = to_dev(NULL, HW * F), *opacity = to_dev(NULL, HW * F);
  int *radii = to_dev(NULL, P * sizeof(int)), *n_touched = to_dev(NULL, P * sizeof(int));
  void *geom = to_dev(NULL, gsaj_geom_workspace_bytes(P));
  void *img = to_dev(NULL, gsaj_image_workspace_bytes(W, H)); /* zeroed once */
  size_t bin_bytes = gsaj_binning_workspace_bytes(1024);     /* a guess; the forward says what it needs */
  void *bin = to_dev(NULL, bin_bytes);
  int R = 0;
  int rc = gsaj_rasterize_forward(P, D, M, bg, W, H, means, shs, NULL, opac, scales, 1.0f, rots, NULL, view, proj, campos, tanfovx,
                                  tanfovy, 0, color, depth, opacity, radii, n_touched, geom, bin, bin_bytes, img, &R, 0, NULL);
  if (rc == GSAJ_ERR_WORKSPACE_TOO_SMALL) { /* grow to the reported R and go again (the reference's resize callback) */
    HIP(hipFree(bin));
    bin_bytes = gsaj_binning_workspace_bytes(R);
    bin = to_dev(NULL, bin_bytes);
    rc = gsaj_rasterize_forward(P, D, M, bg, W, H, means, shs, NULL, opac, scales, 1.0f, rots, NULL, view, proj, campos, tanfovx,
                                tanfovy, 0, color, depth, opacity, radii, n_touched, geom, bin, bin_bytes, img, &R, 0, NULL);
  }
  GSAJ(rc);

  float *g_mean2D = to_dev(NULL, P * 3 * F), *g_conic = to_dev(NULL, P * 4 * F), *g_opac = to_dev(NULL, P * F);
  float *g_color = to_dev(NULL, P * 3 * F), *g_depth = to_dev(NULL, P * F), *g_mean3D = to_dev(NULL, P * 3 * F);
  float *g_cov3D = to_dev(NULL, P * 6 * F), *g_sh = to_dev(NULL, (size_t)P * M * 3 * F), *g_scale = to_dev(NULL, P * 3 * F);
  float *g_rot = to_dev(NULL, P * 4 * F), *g_tau = to_dev(NULL, P * 6 * F), *g_tau_sum = to_dev(NULL, 6 * F);
  GSAJ(gsaj_rasterize_backward(P, D, M, R, bg, W, H, means, shs, NULL, scales, 1.0f, rots, NULL, view, proj, praw, campos, tanfovx,
                               tanfovy, radii, geom, bin, img, dLc, dLd, g_mean2D, g_conic, g_opac, g_color, g_depth, g_mean3D,
                               g_cov3D, g_sh, g_scale, g_rot, g_tau, g_tau_sum, NULL));
  HIP(hipDeviceSynchronize());

  FILE *o = fopen(argv[2], "wb");
  if (!o) return 2;
  fwrite(&R, 4, 1, o);
  wr(o, color, 3 * HW * F);
  wr(o, depth, HW * F);
  wr(o, opacity, HW * F);
  wr(o, radii, P * sizeof(int));
  wr(o, n_touched, P * sizeof(int));
  wr(o, g_mean3D, P * 3 * F);
  wr(o, g_tau_sum, 6 * F);
  fclose(o);
  printf("gsaj %d: P=%d %dx%d num_rendered=%d\n", gsaj_version(), P, W, H, R);
  return 0;
}
