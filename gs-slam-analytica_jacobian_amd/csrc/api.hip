// api.hip -- extern "C" entry points of libgsaj_hip.so (see include/gsaj.h).
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <atomic>
#include <mutex>

#include "gsaj_common.h"
#include "loss_terms.h"

static thread_local char g_err[512] = "";

void gsaj_set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

// ---- event-based per-stage profiler ------------------------------------------------------------
// Shared by every thread / stream of the process: the record table is guarded by a mutex, a record is claimed with an
// atomic index, and the record a launch scope has open is thread-local (a scope opens and closes on one thread), so
// concurrent launches from a tracking and a mapping thread each time their own kernels.
struct ProfRec { int stage; hipEvent_t a, b; };
static std::mutex g_prof_mu;
static ProfRec *g_prof = nullptr;
static std::atomic<bool> g_prof_on{false};
static int g_prof_cap = 0;
static std::atomic<int> g_prof_n{0};
static thread_local int t_prof_open = -1;

void gsaj_prof_mark(int stage, int is_stop, hipStream_t s) {
  if (!g_prof_on.load(std::memory_order_acquire)) return;  // the common case: one relaxed-cost load, no lock
  std::lock_guard<std::mutex> lk(g_prof_mu);
  if (!g_prof) return;
  if (!is_stop) {
    const int i = g_prof_n.fetch_add(1);
    if (i >= g_prof_cap) { g_prof_n.store(g_prof_cap); t_prof_open = -1; return; }
    t_prof_open = i;
    g_prof[i].stage = stage;
    (void)hipEventRecord(g_prof[i].a, s);
  } else if (t_prof_open >= 0 && t_prof_open < g_prof_cap) {
    (void)hipEventRecord(g_prof[t_prof_open].b, s);
    t_prof_open = -1;
  }
}

extern "C" {

int gsaj_profile_begin(int max_records) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  if (g_prof || max_records <= 0) {
    gsaj_set_error("gsaj_profile_begin: already active or bad size");
    return GSAJ_ERR_INVALID_ARGUMENT;
  }
  g_prof = new ProfRec[max_records];
  for (int i = 0; i < max_records; i++) {
    GSAJ_HIP_CHECK(hipEventCreate(&g_prof[i].a));
    GSAJ_HIP_CHECK(hipEventCreate(&g_prof[i].b));
  }
  g_prof_cap = max_records;
  g_prof_n.store(0);
  g_prof_on.store(true, std::memory_order_release);
  return GSAJ_OK;
}

int gsaj_profile_end(float *stage_ms, int *stage_launches) {
  if (!stage_ms || !stage_launches) {
    gsaj_set_error("gsaj_profile_end: null output");
    return GSAJ_ERR_INVALID_ARGUMENT;
  }
  g_prof_on.store(false, std::memory_order_release);  // no new records; launches in flight finish under the lock
  GSAJ_HIP_CHECK(hipDeviceSynchronize());
  std::lock_guard<std::mutex> lk(g_prof_mu);
  if (!g_prof) {
    gsaj_set_error("gsaj_profile_end: not active");
    return GSAJ_ERR_INVALID_ARGUMENT;
  }
  for (int i = 0; i < ST_COUNT; i++) { stage_ms[i] = 0.f; stage_launches[i] = 0; }
  const int nrec = g_prof_n.load() < g_prof_cap ? g_prof_n.load() : g_prof_cap;
  for (int i = 0; i < nrec; i++) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, g_prof[i].a, g_prof[i].b) == hipSuccess) {
      stage_ms[g_prof[i].stage] += ms;
      stage_launches[g_prof[i].stage] += 1;
    }
  }
  for (int i = 0; i < g_prof_cap; i++) { (void)hipEventDestroy(g_prof[i].a); (void)hipEventDestroy(g_prof[i].b); }
  delete[] g_prof;
  g_prof = nullptr;
  g_prof_cap = 0;
  g_prof_n.store(0);
  return GSAJ_OK;
}

const char *gsaj_last_error(void) { return g_err; }
int gsaj_version(void) { return 100; }

size_t gsaj_geom_workspace_bytes(int P) { return geom_carve(nullptr, (size_t)(P > 0 ? P : 0), nullptr) + 256; }
size_t gsaj_image_workspace_bytes(int W, int H) { return image_carve(nullptr, W, H, nullptr) + 256; }
size_t gsaj_binning_workspace_bytes(int R) {
  return bin_carve(nullptr, (size_t)(R > 0 ? R : 0), nullptr) + 256;
}

static char *align_base(void *p) { return reinterpret_cast<char *>(gsaj_align(reinterpret_cast<size_t>(p))); }

int gsaj_forward_preprocess(int P, int D, int M, int W, int H, const float *means3D, const float *shs,
                            const float *colors_precomp, const float *opacities, const float *scales,
                            float scale_modifier, const float *rotations, const float *cov3D_precomp,
                            const float *viewmatrix, const float *projmatrix, const float *campos, float tanfovx,
                            float tanfovy, int prefiltered, int *radii, int *n_touched, void *geom_ws, void *image_ws,
                            void *stream) {
  return gsaj_forward_preprocess_cap(P, D, M, W, H, means3D, shs, colors_precomp, opacities, scales, scale_modifier, rotations,
                                     cov3D_precomp, viewmatrix, projmatrix, campos, tanfovx, tanfovy, prefiltered, radii,
                                     n_touched, geom_ws, image_ws, 0, 0, stream);
}

int gsaj_forward_preprocess_cap(int P, int D, int M, int W, int H, const float *means3D, const float *shs,
                                const float *colors_precomp, const float *opacities, const float *scales,
                                float scale_modifier, const float *rotations, const float *cov3D_precomp,
                                const float *viewmatrix, const float *projmatrix, const float *campos, float tanfovx,
                                float tanfovy, int prefiltered, int *radii, int *n_touched, void *geom_ws, void *image_ws,
                                int capacity, int tile_list_capacity, void *stream) {
  if (P <= 0 || W <= 0 || H <= 0 || !means3D || !opacities || !viewmatrix || !projmatrix || !geom_ws || !image_ws ||
      !n_touched) {
    gsaj_set_error("gsaj_forward_preprocess: invalid argument (P=%d W=%d H=%d)", P, W, H);
    return GSAJ_ERR_INVALID_ARGUMENT;
  }
  if ((shs == nullptr) == (colors_precomp == nullptr)) {
    gsaj_set_error("Please provide excatly one of either SHs or precomputed colors!");
    return GSAJ_ERR_INVALID_ARGUMENT;
  }
  if (((scales == nullptr || rotations == nullptr) && cov3D_precomp == nullptr) ||
      ((scales != nullptr || rotations != nullptr) && cov3D_precomp != nullptr)) {
    gsaj_set_error("Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!");
    return GSAJ_ERR_INVALID_ARGUMENT;
  }
  if (shs && (!campos || M <= 0 || D < 0 || (D + 1) * (D + 1) > M || D > 3)) {
    gsaj_set_error("gsaj_forward_preprocess: SH degree %d needs %d coefficients, got M=%d", D, (D + 1) * (D + 1), M);
    return GSAJ_ERR_INVALID_ARGUMENT;
  }
  GeomWS g;
  geom_carve(align_base(geom_ws), (size_t)P, &g);
  FwdParams p;
  p.P = P; p.D = D; p.M = M; p.W = W; p.H = H;
  p.means3D = means3D; p.shs = shs; p.colors_precomp = colors_precomp; p.opacities = opacities;
  p.scales = scales; p.rotations = rotations; p.cov3D_precomp = cov3D_precomp;
  p.viewmatrix = viewmatrix; p.projmatrix = projmatrix; p.campos = campos;
  p.scale_modifier = scale_modifier; p.tanfovx = tanfovx; p.tanfovy = tanfovy;
  p.focal_y = H / (2.0f * tanfovy);
  p.focal_x = W / (2.0f * tanfovx);
  p.prefiltered = prefiltered;
  p.grid_x = (W + TILE - 1) / TILE; p.grid_y = (H + TILE - 1) / TILE;
  p.capacity = capacity;
  p.sort_cap = (tile_list_capacity > 0 && tile_list_capacity < SORT_CAP) ? tile_list_capacity : SORT_CAP;
  p.views = 1;
  ImageWS im;
  image_carve(align_base(image_ws), W, H, &im);
  return launch_preprocess(p, radii ? radii : g.internal_radii, n_touched, g, im, ViewStrides{0, 0, 0}, (hipStream_t)stream);
}

int gsaj_forward_num_rendered(int W, int H, const void *image_ws, void *stream, int *num_rendered, int *max_tile_list) {
  if (W <= 0 || H <= 0 || !image_ws || !num_rendered) {
    gsaj_set_error("gsaj_forward_num_rendered: invalid argument");
    return GSAJ_ERR_INVALID_ARGUMENT;
  }
  ImageWS im;
  image_carve(align_base(const_cast<void *>(image_ws)), W, H, &im);
  uint32_t host[4] = {0, 0, 0, 0};
  GSAJ_HIP_CHECK(hipMemcpyAsync(host, im.counters, sizeof(host), hipMemcpyDeviceToHost, (hipStream_t)stream));
  GSAJ_HIP_CHECK(hipStreamSynchronize((hipStream_t)stream));
  *num_rendered = (int)host[0];
  if (max_tile_list) *max_tile_list = (int)host[2];
  if (host[1] & ERR_PREFILTERED) {
    gsaj_set_error("Point is filtered although prefiltered is set. This shouldn't happen!");
    return GSAJ_ERR_PREFILTERED_CULLED;
  }
  if (host[1] & ERR_INTERNAL) {
    gsaj_set_error("internal error: tile histogram total != instance total");
    return GSAJ_ERR_HIP;
  }
  if (host[1] & ERR_CAPACITY) {
    gsaj_set_error("async forward aborted: binning arena too small (R=%u, longest tile list=%u)", host[0], host[2]);
    return GSAJ_ERR_WORKSPACE_TOO_SMALL;
  }
  return GSAJ_OK;
}

int gsaj_forward_aborted_count(int W, int H, void *image_ws, void *stream, int *count) {
  if (W <= 0 || H <= 0 || !image_ws || !count) {
    gsaj_set_error("gsaj_forward_aborted_count: invalid argument");
    return GSAJ_ERR_INVALID_ARGUMENT;
  }
  ImageWS im;
  image_carve(align_base(image_ws), W, H, &im);
  uint32_t host = 0;
  GSAJ_HIP_CHECK(hipMemcpyAsync(&host, im.sticky, sizeof(host), hipMemcpyDeviceToHost, (hipStream_t)stream));
  GSAJ_HIP_CHECK(hipStreamSynchronize((hipStream_t)stream));
  *count = (int)host;
  // read and clear: the next call reports the aborts since this one (a caller that re-sized its arena starts from zero)
  if (host) GSAJ_HIP_CHECK(hipMemsetAsync(im.sticky, 0, sizeof(uint32_t), (hipStream_t)stream));
  return GSAJ_OK;
}

const uint32_t *gsaj_forward_abort_flag(int W, int H, void *image_ws) {
  if (W <= 0 || H <= 0 || !image_ws) {
    gsaj_set_error("gsaj_forward_abort_flag: invalid argument");
    return nullptr;
  }
  ImageWS im;
  image_carve(align_base(image_ws), W, H, &im);
  return im.counters + 4;
}

int gsaj_set_tile_band(int W, int H, void *image_ws, int tile_row_begin, int tile_row_end, void *stream) {
  const int gy = (H + TILE - 1) / TILE;
  if (W <= 0 || H <= 0 || !image_ws || tile_row_begin < 0 || tile_row_end < tile_row_begin || tile_row_end > gy || gy > 0xffff) {
    gsaj_set_error("gsaj_set_tile_band: rows [%d, %d) are not a band of the %d tile rows of a %dx%d frame", tile_row_begin,
                   tile_row_end, gy, W, H);
    return GSAJ_ERR_INVALID_ARGUMENT;
  }
  ImageWS im;
  image_carve(align_base(image_ws), W, H, &im);
  // 0 = the whole frame; an empty band [b, b) with b > 0 is kept as such (the rank renders nothing); [0, 0) cannot be
  // told from "whole frame" and is refused
  const bool whole = tile_row_begin == 0 && tile_row_end == gy;
  if (!whole && tile_row_end == 0) {
    gsaj_set_error("gsaj_set_tile_band: the empty band [0, 0) cannot be expressed; give an empty band as [b, b) with b > 0");
    return GSAJ_ERR_INVALID_ARGUMENT;
  }
  const uint32_t band = whole ? 0u : ((uint32_t)tile_row_begin | ((uint32_t)tile_row_end << 16));
  GSAJ_HIP_CHECK(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(im.sticky + 1), (int)band, 1, (hipStream_t)stream));
  GSAJ_HIP_CHECK(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(im.sticky + 2), (int)~band, 1, (hipStream_t)stream));
  return GSAJ_OK;
}

int gsaj_forward_render(int P, int R, int max_tile_list, int W, int H, const float *bg, const float *colors_precomp,
                        const int *radii,
                        void *geom_ws, void *binning_ws, size_t binning_ws_bytes, void *image_ws, float *out_color,
                        float *out_depth, float *out_opacity, int *n_touched, int flags, void *stream) {
  if (P <= 0 || R < 0 || W <= 0 || H <= 0 || !bg || !geom_ws || !binning_ws || !image_ws || !out_color || !out_depth ||
      !out_opacity || !n_touched) {
    gsaj_set_error("gsaj_forward_render: invalid argument");
    return GSAJ_ERR_INVALID_ARGUMENT;
  }
  const size_t need = gsaj_binning_workspace_bytes(R);
  if (binning_ws_bytes < need) {
    gsaj_set_error("binning workspace too small: have %zu bytes, need %zu for R=%d", binning_ws_bytes, need, R);
    return GSAJ_ERR_WORKSPACE_TOO_SMALL;
  }
  hipStream_t s = (hipStream_t)stream;
  GeomWS g;
  geom_carve(align_base(geom_ws), (size_t)P, &g);
  ImageWS im;
  image_carve(align_base(image_ws), W, H, &im);
  BinWS b;
  bin_carve(align_base(binning_ws), (size_t)R, &b);
  const int gx = (W + TILE - 1) / TILE, gy = (H + TILE - 1) / TILE;
  (void)colors_precomp;  // (already in the splat rows: gsaj_forward_preprocess)
  (void)radii;
  // per-tile lists sorted by the tile's workgroup: in one LDS pass when the list fits the capacity sized from max_tile_list,
  // in LDS-sized chunks + merge passes when it does not (max_tile_list < 0 forces that path with the smallest capacity)
  const int sort_cap = max_tile_list < 0 ? 128 : (max_tile_list > SORT_CAP ? SORT_CAP : max_tile_list);
  int rc;
  if ((rc = launch_tile_binning(P, sort_cap, (flags & GSAJ_FWD_RECORDS_FP16) ? 1 : 0, gx, gy, g, b, im, 1, ViewStrides{0, 0, 0}, s)) != GSAJ_OK) return rc;
  return launch_render_forward(P, W, H, gx, gy, bg, g, b, im, out_color, out_depth, out_opacity, n_touched, 1, ViewStrides{0, 0, 0}, s);
}

int gsaj_rasterize_forward(int P, int D, int M, const float *bg, int W, int H, const float *means3D, const float *shs,
                           const float *colors_precomp, const float *opacities, const float *scales,
                           float scale_modifier, const float *rotations, const float *cov3D_precomp,
                           const float *viewmatrix, const float *projmatrix, const float *campos, float tanfovx,
                           float tanfovy, int prefiltered, float *out_color, float *out_depth, float *out_opacity,
                           int *radii, int *n_touched, void *geom_ws, void *binning_ws, size_t binning_ws_bytes,
                           void *image_ws, int *num_rendered_out, int flags, void *stream) {
  int rc = gsaj_forward_preprocess(P, D, M, W, H, means3D, shs, colors_precomp, opacities, scales, scale_modifier,
                                   rotations, cov3D_precomp, viewmatrix, projmatrix, campos, tanfovx, tanfovy,
                                   prefiltered, radii, n_touched, geom_ws, image_ws, stream);
  if (rc != GSAJ_OK) return rc;
  int R = 0, max_tile = 0;
  rc = gsaj_forward_num_rendered(W, H, image_ws, stream, &R, &max_tile);
  if (num_rendered_out) *num_rendered_out = R;
  if (rc != GSAJ_OK) return rc;
  rc = gsaj_forward_render(P, R, max_tile, W, H, bg, colors_precomp, radii, geom_ws, binning_ws, binning_ws_bytes, image_ws,
                           out_color, out_depth, out_opacity, n_touched, flags, stream);
  return rc == GSAJ_OK ? R : rc;
}

static int forward_async_impl(int P, int D, int M, const float *bg, int W, int H, const float *means3D, const float *shs,
                              const float *colors_precomp, const float *opacities, const float *scales,
                              float scale_modifier, const float *rotations, const float *cov3D_precomp,
                              const float *viewmatrix, const float *projmatrix, const float *campos, float tanfovx,
                              float tanfovy, int prefiltered, float *out_color, float *out_depth, float *out_opacity,
                              int *radii, int *n_touched, void *geom_ws, void *binning_ws, size_t binning_ws_bytes,
                              int capacity, int tile_list_capacity, void *image_ws, int flags, void *stream, const FusedLoss *fl,
                              float *out_scalars) {
  if (capacity <= 0 || !binning_ws || !bg || !out_color || !out_depth || !out_opacity || !n_touched) {
    gsaj_set_error("gsaj_rasterize_forward_async: invalid argument");
    return GSAJ_ERR_INVALID_ARGUMENT;
  }
  if (binning_ws_bytes < gsaj_binning_workspace_bytes(capacity)) {
    gsaj_set_error("binning workspace smaller than gsaj_binning_workspace_bytes(capacity=%d)", capacity);
    return GSAJ_ERR_WORKSPACE_TOO_SMALL;
  }
  int rc = gsaj_forward_preprocess_cap(P, D, M, W, H, means3D, shs, colors_precomp, opacities, scales, scale_modifier,
                                       rotations, cov3D_precomp, viewmatrix, projmatrix, campos, tanfovx, tanfovy,
                                       prefiltered, radii, n_touched, geom_ws, image_ws, capacity, tile_list_capacity, stream);
  if (rc != GSAJ_OK) return rc;
  // No host round trip: grids do not depend on R, the arena is carved for `capacity` instances, and
  // k_scan aborts the frame on the device if R or a tile list does not fit (gsaj_forward_num_rendered
  // reports it whenever the caller chooses to look).
  hipStream_t s = (hipStream_t)stream;
  GeomWS g;
  geom_carve(align_base(geom_ws), (size_t)P, &g);
  ImageWS im;
  image_carve(align_base(image_ws), W, H, &im);
  BinWS b;
  bin_carve(align_base(binning_ws), (size_t)capacity, &b);
  const int gx = (W + TILE - 1) / TILE, gy = (H + TILE - 1) / TILE;
  const int sort_cap = (tile_list_capacity > 0 && tile_list_capacity < SORT_CAP) ? tile_list_capacity : SORT_CAP;
  if ((rc = launch_tile_binning(P, sort_cap, (flags & GSAJ_FWD_RECORDS_FP16) ? 1 : 0, gx, gy, g, b, im, 1, ViewStrides{0, 0, 0}, s)) != GSAJ_OK) return rc;
  if ((rc = launch_render_forward(P, W, H, gx, gy, bg, g, b, im, out_color, out_depth, out_opacity, n_touched, 1, ViewStrides{0, 0, 0}, s, fl)) != GSAJ_OK) return rc;
  if (fl) return launch_loss_finalize(*fl, gsaj_fwd_loss_slots(W, H), W, H, im.counters + 4, out_scalars, s);
  return GSAJ_OK;
}

int gsaj_rasterize_forward_async(int P, int D, int M, const float *bg, int W, int H, const float *means3D, const float *shs,
                                 const float *colors_precomp, const float *opacities, const float *scales,
                                 float scale_modifier, const float *rotations, const float *cov3D_precomp,
                                 const float *viewmatrix, const float *projmatrix, const float *campos, float tanfovx,
                                 float tanfovy, int prefiltered, float *out_color, float *out_depth, float *out_opacity,
                                 int *radii, int *n_touched, void *geom_ws, void *binning_ws, size_t binning_ws_bytes,
                                 int capacity, int tile_list_capacity, void *image_ws, int flags, void *stream) {
  return forward_async_impl(P, D, M, bg, W, H, means3D, shs, colors_precomp, opacities, scales, scale_modifier, rotations, cov3D_precomp,
                            viewmatrix, projmatrix, campos, tanfovx, tanfovy, prefiltered, out_color, out_depth, out_opacity, radii,
                            n_touched, geom_ws, binning_ws, binning_ws_bytes, capacity, tile_list_capacity, image_ws, flags, stream,
                            nullptr, nullptr);
}

static int fused_loss_args(const char *who, int loss_flags, const float *gt_color, const float *gt_depth, const float *exposure_a,
                           const float *exposure_b) {
  const bool mono = loss_flags & GSAJ_LOSS_MONOCULAR, noexp = loss_flags & GSAJ_LOSS_NO_EXPOSURE;
  if ((loss_flags & GSAJ_LOSS_COMPUTE_LOSS) || !gt_color || (!mono && !gt_depth) || (!noexp && (!exposure_a || !exposure_b))) {
    gsaj_set_error("%s: invalid loss arguments (flags=%d; GSAJ_LOSS_COMPUTE_LOSS has no fused form)", who, loss_flags);
    return GSAJ_ERR_INVALID_ARGUMENT;
  }
  return GSAJ_OK;
}

size_t gsaj_fused_loss_workspace_bytes(int W, int H) { return (size_t)gsaj_fwd_loss_slots(W, H) * 4 * sizeof(float) + 256; }

int gsaj_rasterize_forward_loss(int P, int D, int M, const float *bg, int W, int H, const float *means3D, const float *shs,
                                const float *colors_precomp, const float *opacities, const float *scales, float scale_modifier,
                                const float *rotations, const float *cov3D_precomp, const float *viewmatrix,
                                const float *projmatrix, const float *campos, float tanfovx, float tanfovy, int prefiltered,
                                float *out_color, float *out_depth, float *out_opacity, int *radii, int *n_touched, void *geom_ws,
                                void *binning_ws, size_t binning_ws_bytes, int capacity, int tile_list_capacity, void *image_ws,
                                int flags, int loss_flags, float alpha, float rgb_boundary_threshold, const float *gt_color,
                                const float *gt_depth, const uint8_t *grad_mask, const float *exposure_a, const float *exposure_b,
                                float *out_scalars, void *loss_ws, void *stream) {
  int rc = fused_loss_args("gsaj_rasterize_forward_loss", loss_flags, gt_color, gt_depth, exposure_a, exposure_b);
  if (rc != GSAJ_OK) return rc;
  if (!out_scalars || !loss_ws) {
    gsaj_set_error("gsaj_rasterize_forward_loss: out_scalars and loss_ws are required");
    return GSAJ_ERR_INVALID_ARGUMENT;
  }
  FusedLoss fl{};
  fl.flags = loss_flags; fl.alpha = alpha; fl.rgb_thr = rgb_boundary_threshold;
  fl.gt_color = gt_color; fl.gt_depth = gt_depth; fl.grad_mask = grad_mask; fl.exp_a = exposure_a; fl.exp_b = exposure_b;
  fl.partials = reinterpret_cast<float *>(align_base(loss_ws));
  return forward_async_impl(P, D, M, bg, W, H, means3D, shs, colors_precomp, opacities, scales, scale_modifier, rotations, cov3D_precomp,
                            viewmatrix, projmatrix, campos, tanfovx, tanfovy, prefiltered, out_color, out_depth, out_opacity, radii,
                            n_touched, geom_ws, binning_ws, binning_ws_bytes, capacity, tile_list_capacity, image_ws, flags, stream, &fl,
                            out_scalars);
}

// ---- batched multi-view entry points: K views of ONE Gaussian map (a mapping window) --------------------------------
static int batch_workspaces(int K, int P, int capacity, int W, int H, void *geom_ws, void *binning_ws, size_t binning_ws_bytes,
                            void *image_ws, GeomWS *g, BinWS *b, ImageWS *im, ViewStrides *vs) {
  vs->geom = gsaj_geom_workspace_bytes(P);
  vs->image = gsaj_image_workspace_bytes(W, H);
  vs->bin = gsaj_binning_workspace_bytes(capacity);
  if (((uintptr_t)geom_ws | (uintptr_t)binning_ws | (uintptr_t)image_ws) & 255) {
    gsaj_set_error("batched entry points need 256-byte aligned workspaces");
    return GSAJ_ERR_INVALID_ARGUMENT;
  }
  if (binning_ws_bytes < vs->bin * (size_t)K) {
    gsaj_set_error("binning workspace smaller than K x gsaj_binning_workspace_bytes(capacity=%d)", capacity);
    return GSAJ_ERR_WORKSPACE_TOO_SMALL;
  }
  geom_carve(reinterpret_cast<char *>(geom_ws), (size_t)P, g);
  image_carve(reinterpret_cast<char *>(image_ws), W, H, im);
  bin_carve(reinterpret_cast<char *>(binning_ws), (size_t)capacity, b);
  return GSAJ_OK;
}

int gsaj_rasterize_forward_batch(int K, int P, int D, int M, const float *bg, int W, int H, const float *means3D, const float *shs,
                                 const float *colors_precomp, const float *opacities, const float *scales, float scale_modifier,
                                 const float *rotations, const float *cov3D_precomp, const float *viewmatrices,
                                 const float *projmatrices, const float *campos, float tanfovx, float tanfovy, int prefiltered,
                                 float *out_color, float *out_depth, float *out_opacity, int *radii, int *n_touched, void *geom_ws,
                                 void *binning_ws, size_t binning_ws_bytes, int capacity, int tile_list_capacity, void *image_ws,
                                 int flags, void *stream) {
  if (K <= 0 || P <= 0 || W <= 0 || H <= 0 || capacity <= 0 || !means3D || !opacities || !viewmatrices || !projmatrices || !bg ||
      !out_color || !out_depth || !out_opacity || !radii || !n_touched || !geom_ws || !binning_ws || !image_ws) {
    gsaj_set_error("gsaj_rasterize_forward_batch: invalid argument (K=%d P=%d W=%d H=%d capacity=%d)", K, P, W, H, capacity);
    return GSAJ_ERR_INVALID_ARGUMENT;
  }
  if ((shs == nullptr) == (colors_precomp == nullptr)) {
    gsaj_set_error("Please provide excatly one of either SHs or precomputed colors!");
    return GSAJ_ERR_INVALID_ARGUMENT;
  }
  if (((scales == nullptr || rotations == nullptr) && cov3D_precomp == nullptr) ||
      ((scales != nullptr || rotations != nullptr) && cov3D_precomp != nullptr)) {
    gsaj_set_error("Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!");
    return GSAJ_ERR_INVALID_ARGUMENT;
  }
  if (shs && (!campos || M <= 0 || D < 0 || (D + 1) * (D + 1) > M || D > 3)) {
    gsaj_set_error("gsaj_rasterize_forward_batch: SH degree %d needs %d coefficients, got M=%d", D, (D + 1) * (D + 1), M);
    return GSAJ_ERR_INVALID_ARGUMENT;
  }
  GeomWS g; BinWS b; ImageWS im; ViewStrides vs;
  int rc = batch_workspaces(K, P, capacity, W, H, geom_ws, binning_ws, binning_ws_bytes, image_ws, &g, &b, &im, &vs);
  if (rc != GSAJ_OK) return rc;
  hipStream_t s = (hipStream_t)stream;
  FwdParams p;
  p.P = P; p.D = D; p.M = M; p.W = W; p.H = H;
  p.means3D = means3D; p.shs = shs; p.colors_precomp = colors_precomp; p.opacities = opacities;
  p.scales = scales; p.rotations = rotations; p.cov3D_precomp = cov3D_precomp;
  p.viewmatrix = viewmatrices; p.projmatrix = projmatrices; p.campos = campos;
  p.scale_modifier = scale_modifier; p.tanfovx = tanfovx; p.tanfovy = tanfovy;
  p.focal_y = H / (2.0f * tanfovy);
  p.focal_x = W / (2.0f * tanfovx);
  p.prefiltered = prefiltered;
  p.grid_x = (W + TILE - 1) / TILE; p.grid_y = (H + TILE - 1) / TILE;
  p.capacity = capacity;
  p.sort_cap = (tile_list_capacity > 0 && tile_list_capacity < SORT_CAP) ? tile_list_capacity : SORT_CAP;
  p.views = K;
  if ((rc = launch_preprocess(p, radii, n_touched, g, im, vs, s)) != GSAJ_OK) return rc;
  if ((rc = launch_tile_binning(P, p.sort_cap, (flags & GSAJ_FWD_RECORDS_FP16) ? 1 : 0, p.grid_x, p.grid_y, g, b, im, K, vs, s)) != GSAJ_OK)
    return rc;
  return launch_render_forward(P, W, H, p.grid_x, p.grid_y, bg, g, b, im, out_color, out_depth, out_opacity, n_touched, K, vs, s);
}

int gsaj_rasterize_backward_batch(int K, int P, int D, int M, int capacity, const float *bg, int W, int H, const float *means3D,
                                  const float *shs, const float *colors_precomp, const float *scales, float scale_modifier,
                                  const float *rotations, const float *cov3D_precomp, const float *viewmatrices,
                                  const float *projmatrices, const float *projmatrix_raw, const float *campos, float tanfovx,
                                  float tanfovy, const int *radii, void *geom_ws, void *binning_ws, void *image_ws,
                                  const float *dL_dpix, const float *dL_dpix_depth, float *dL_dmean2D, float *dL_dconic,
                                  float *dL_dopacity, float *dL_dcolor, float *dL_ddepth, float *dL_dmean3D, float *dL_dcov3D,
                                  float *dL_dsh, float *dL_dscale, float *dL_drot, float *dL_dtau, float *dL_dtau_sum, int flags,
                                  void *stream) {
  if (K <= 0 || P <= 0 || capacity <= 0 || W <= 0 || H <= 0 || !bg || !means3D || !viewmatrices || !projmatrices || !projmatrix_raw ||
      !radii || !geom_ws || !binning_ws || !image_ws || !dL_dpix || !dL_dpix_depth || !dL_dopacity || !dL_dmean3D || !dL_dcov3D ||
      !dL_dtau_sum) {
    gsaj_set_error("gsaj_rasterize_backward_batch: invalid argument");
    return GSAJ_ERR_INVALID_ARGUMENT;
  }
  if ((shs && (!dL_dsh || !campos)) || (scales && (!rotations || !dL_dscale || !dL_drot)) || (!scales && !cov3D_precomp)) {
    gsaj_set_error("gsaj_rasterize_backward_batch: missing gradient buffer for a provided input");
    return GSAJ_ERR_INVALID_ARGUMENT;
  }
  GeomWS g; BinWS b; ImageWS im; ViewStrides vs;
  int rc = batch_workspaces(K, P, capacity, W, H, geom_ws, binning_ws, gsaj_binning_workspace_bytes(capacity) * (size_t)K, image_ws,
                            &g, &b, &im, &vs);
  if (rc != GSAJ_OK) return rc;
  hipStream_t s = (hipStream_t)stream;
  const int gx = (W + TILE - 1) / TILE, gy = (H + TILE - 1) / TILE;
  if (!(flags & GSAJ_BWD_ONLY_CHAIN)) {
    if ((rc = launch_render_backward(capacity, W, H, gx, gy, bg, g, b, im, dL_dpix, dL_dpix_depth, K, vs, s)) != GSAJ_OK) return rc;
    if ((rc = launch_gather_sums(P, K, radii, g, b, im, vs, s)) != GSAJ_OK) return rc;
  }
  if (flags & GSAJ_BWD_ONLY_COMPOSITE) return GSAJ_OK;
  BwdParams p;
  p.P = P; p.D = D; p.M = M; p.W = W; p.H = H;
  p.means3D = means3D; p.shs = shs; p.scales = scales; p.rotations = rotations;
  p.cov3Ds = cov3D_precomp ? cov3D_precomp : g.cov3D;
  p.viewmatrix = viewmatrices; p.projmatrix = projmatrices; p.projmatrix_raw = projmatrix_raw; p.campos = campos;
  p.scale_modifier = scale_modifier; p.tanfovx = tanfovx; p.tanfovy = tanfovy;
  p.focal_y = H / (2.0f * tanfovy);
  p.focal_x = W / (2.0f * tanfovx);
  p.grid_x = gx; p.grid_y = gy;
  p.radii = radii;
  p.dL_dmean2D = dL_dmean2D; p.dL_dconic = dL_dconic; p.dL_dopacity = dL_dopacity; p.dL_dcolor = dL_dcolor;
  p.dL_ddepth = dL_ddepth; p.dL_dmean3D = dL_dmean3D; p.dL_dcov3D = dL_dcov3D; p.dL_dsh = dL_dsh;
  p.dL_dscale = dL_dscale; p.dL_drot = dL_drot; p.dL_dtau = dL_dtau; p.dL_dtau_sum = dL_dtau_sum;
  (void)colors_precomp;
  return launch_gaussian_backward_batch(p, K, g, b, im, vs, (flags & GSAJ_BWD_ACCUMULATE) ? 1 : 0, s);
}

static int backward_impl(int P, int D, int M, int R, const float *bg, int W, int H, const float *means3D,
                         const float *shs, const float *colors_precomp, const float *scales, float scale_modifier,
                         const float *rotations, const float *cov3D_precomp, const float *viewmatrix,
                         const float *projmatrix, const float *projmatrix_raw, const float *campos, float tanfovx,
                         float tanfovy, const int *radii, void *geom_ws, void *binning_ws, void *image_ws,
                         const float *dL_dpix, const float *dL_dpix_depth, float *dL_dmean2D, float *dL_dconic,
                         float *dL_dopacity, float *dL_dcolor, float *dL_ddepth, float *dL_dmean3D,
                         float *dL_dcov3D, float *dL_dsh, float *dL_dscale, float *dL_drot, float *dL_dtau,
                         float *dL_dtau_sum, void *stream, const FusedLoss *fl) {
  // pose-only mode (tracking: only the camera is optimised): every per-Gaussian output pointer NULL, dL_dtau_sum given
  const bool pose_only = !dL_dmean2D && !dL_dconic && !dL_dopacity && !dL_dcolor && !dL_ddepth && !dL_dmean3D && !dL_dcov3D &&
                         !dL_dsh && !dL_dscale && !dL_drot && dL_dtau_sum;
  if (P <= 0 || R < 0 || W <= 0 || H <= 0 || !bg || !means3D || !viewmatrix || !projmatrix || !projmatrix_raw ||
      !geom_ws || !binning_ws || !image_ws || (!fl && (!dL_dpix || !dL_dpix_depth)) ||
      (!pose_only && (!dL_dmean2D || !dL_dconic || !dL_dopacity || !dL_dcolor || !dL_ddepth || !dL_dmean3D || !dL_dcov3D))) {
    gsaj_set_error("gsaj_rasterize_backward: invalid argument");
    return GSAJ_ERR_INVALID_ARGUMENT;
  }
  if ((shs && ((!pose_only && !dL_dsh) || !campos)) || (scales && (!rotations || (!pose_only && (!dL_dscale || !dL_drot)))) ||
      (!scales && !cov3D_precomp)) {
    gsaj_set_error("gsaj_rasterize_backward: missing gradient buffer for a provided input");
    return GSAJ_ERR_INVALID_ARGUMENT;
  }
  hipStream_t s = (hipStream_t)stream;
  GeomWS g;
  geom_carve(align_base(geom_ws), (size_t)P, &g);
  ImageWS im;
  image_carve(align_base(image_ws), W, H, &im);
  BinWS b;
  bin_carve(align_base(binning_ws), (size_t)R, &b);
  const int gx = (W + TILE - 1) / TILE, gy = (H + TILE - 1) / TILE;
  // every output row is written by the kernels (zeros for culled Gaussians): no memsets
  int rc = launch_render_backward(R, W, H, gx, gy, bg, g, b, im, dL_dpix, dL_dpix_depth, 1, ViewStrides{0, 0, 0}, s, fl);
  if (rc != GSAJ_OK) return rc;
  BwdParams p;
  p.P = P; p.D = D; p.M = M; p.W = W; p.H = H;
  p.means3D = means3D; p.shs = shs; p.scales = scales; p.rotations = rotations;
  p.cov3Ds = cov3D_precomp ? cov3D_precomp : g.cov3D;
  p.viewmatrix = viewmatrix; p.projmatrix = projmatrix; p.projmatrix_raw = projmatrix_raw; p.campos = campos;
  p.scale_modifier = scale_modifier; p.tanfovx = tanfovx; p.tanfovy = tanfovy;
  p.focal_y = H / (2.0f * tanfovy);
  p.focal_x = W / (2.0f * tanfovx);
  p.grid_x = gx; p.grid_y = gy;
  p.radii = radii ? radii : g.internal_radii;
  p.dL_dmean2D = dL_dmean2D; p.dL_dconic = dL_dconic; p.dL_dopacity = dL_dopacity; p.dL_dcolor = dL_dcolor;
  p.dL_ddepth = dL_ddepth; p.dL_dmean3D = dL_dmean3D; p.dL_dcov3D = dL_dcov3D; p.dL_dsh = dL_dsh;
  p.dL_dscale = dL_dscale; p.dL_drot = dL_drot; p.dL_dtau = dL_dtau; p.dL_dtau_sum = dL_dtau_sum;
  (void)colors_precomp;
  return launch_gaussian_backward(p, g, b, im, s);
}

int gsaj_rasterize_backward(int P, int D, int M, int R, const float *bg, int W, int H, const float *means3D,
                            const float *shs, const float *colors_precomp, const float *scales, float scale_modifier,
                            const float *rotations, const float *cov3D_precomp, const float *viewmatrix,
                            const float *projmatrix, const float *projmatrix_raw, const float *campos, float tanfovx,
                            float tanfovy, const int *radii, void *geom_ws, void *binning_ws, void *image_ws,
                            const float *dL_dpix, const float *dL_dpix_depth, float *dL_dmean2D, float *dL_dconic,
                            float *dL_dopacity, float *dL_dcolor, float *dL_ddepth, float *dL_dmean3D,
                            float *dL_dcov3D, float *dL_dsh, float *dL_dscale, float *dL_drot, float *dL_dtau,
                            float *dL_dtau_sum, void *stream) {
  return backward_impl(P, D, M, R, bg, W, H, means3D, shs, colors_precomp, scales, scale_modifier, rotations, cov3D_precomp, viewmatrix,
                       projmatrix, projmatrix_raw, campos, tanfovx, tanfovy, radii, geom_ws, binning_ws, image_ws, dL_dpix, dL_dpix_depth,
                       dL_dmean2D, dL_dconic, dL_dopacity, dL_dcolor, dL_ddepth, dL_dmean3D, dL_dcov3D, dL_dsh, dL_dscale, dL_drot, dL_dtau,
                       dL_dtau_sum, stream, nullptr);
}

int gsaj_rasterize_backward_loss(int P, int D, int M, int R, const float *bg, int W, int H, const float *means3D, const float *shs,
                                 const float *colors_precomp, const float *scales, float scale_modifier, const float *rotations,
                                 const float *cov3D_precomp, const float *viewmatrix, const float *projmatrix,
                                 const float *projmatrix_raw, const float *campos, float tanfovx, float tanfovy, const int *radii,
                                 void *geom_ws, void *binning_ws, void *image_ws, int loss_flags, float alpha,
                                 float rgb_boundary_threshold, const float *color, const float *depth, const float *opacity,
                                 const float *gt_color, const float *gt_depth, const uint8_t *grad_mask, const float *exposure_a,
                                 const float *exposure_b, float *dL_dmean2D, float *dL_dconic, float *dL_dopacity, float *dL_dcolor,
                                 float *dL_ddepth, float *dL_dmean3D, float *dL_dcov3D, float *dL_dsh, float *dL_dscale,
                                 float *dL_drot, float *dL_dtau, float *dL_dtau_sum, void *stream) {
  int rc = fused_loss_args("gsaj_rasterize_backward_loss", loss_flags, gt_color, gt_depth, exposure_a, exposure_b);
  if (rc != GSAJ_OK) return rc;
  if (!color || !opacity || (!(loss_flags & GSAJ_LOSS_MONOCULAR) && !depth)) {
    gsaj_set_error("gsaj_rasterize_backward_loss: the forward's color / depth / opacity images are required");
    return GSAJ_ERR_INVALID_ARGUMENT;
  }
  FusedLoss fl{};
  fl.flags = loss_flags; fl.alpha = alpha; fl.rgb_thr = rgb_boundary_threshold;
  fl.gt_color = gt_color; fl.gt_depth = gt_depth; fl.grad_mask = grad_mask; fl.exp_a = exposure_a; fl.exp_b = exposure_b;
  fl.color = color; fl.depth = depth; fl.opacity = opacity;
  return backward_impl(P, D, M, R, bg, W, H, means3D, shs, colors_precomp, scales, scale_modifier, rotations, cov3D_precomp, viewmatrix,
                       projmatrix, projmatrix_raw, campos, tanfovx, tanfovy, radii, geom_ws, binning_ws, image_ws, nullptr, nullptr,
                       dL_dmean2D, dL_dconic, dL_dopacity, dL_dcolor, dL_ddepth, dL_dmean3D, dL_dcov3D, dL_dsh, dL_dscale, dL_drot, dL_dtau,
                       dL_dtau_sum, stream, &fl);
}

int gsaj_mark_visible(int P, const float *means3D, const float *viewmatrix, const float *projmatrix, uint8_t *present,
                      void *stream) {
  (void)projmatrix;  // the reference's test only uses the view-space depth (auxiliary.h:154)
  if (P <= 0 || !means3D || !viewmatrix || !present) {
    gsaj_set_error("gsaj_mark_visible: invalid argument");
    return GSAJ_ERR_INVALID_ARGUMENT;
  }
  return launch_mark_visible(P, means3D, viewmatrix, present, (hipStream_t)stream);
}

int gsaj_debug_export(int P, int R, int W, int H, const void *geom_ws, const void *binning_ws, const void *image_ws,
                      float *means2D, float *depths, float *cov3D, float *conic_opacity, float *rgb, uint8_t *clamped,
                      uint32_t *tiles_touched, uint32_t *point_list, uint32_t *ranges, float *final_T,
                      uint32_t *n_contrib, void *stream) {
  hipStream_t s = (hipStream_t)stream;
  const size_t Pz = (size_t)P, N = (size_t)W * H;
  const size_t tiles = (size_t)((W + TILE - 1) / TILE) * ((H + TILE - 1) / TILE);
#define CP(dst, src, bytes) \
  if (dst) GSAJ_HIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, s))
  if (geom_ws) {
    GeomWS g;
    geom_carve(align_base(const_cast<void *>(geom_ws)), Pz, &g);
    // means2D, conic_opacity, rgb live in the 48-byte splat rows only: strided copies
#define CP2D(dst, src, width) \
  if (dst) GSAJ_HIP_CHECK(hipMemcpy2DAsync(dst, width, src, sizeof(float4) * REC_F4, width, Pz, hipMemcpyDeviceToDevice, s))
    CP2D(means2D, g.splat, sizeof(float2));
    CP2D(conic_opacity, g.splat + 1, sizeof(float4));
    CP2D(rgb, g.splat + 2, sizeof(float) * 3);
#undef CP2D
    CP(depths, g.depths, sizeof(float) * Pz);
    CP(cov3D, g.cov3D, sizeof(float) * 6 * Pz);
    CP(clamped, g.clamped, 3 * Pz);
    CP(tiles_touched, g.tiles_touched, sizeof(uint32_t) * Pz);
  }
  if (binning_ws && R > 0) {
    BinWS b;
    bin_carve(align_base(const_cast<void *>(binning_ws)), (size_t)R, &b);
    CP(point_list, b.point_list, sizeof(uint32_t) * (size_t)R);
  }
  if (image_ws) {
    ImageWS im;
    image_carve(align_base(const_cast<void *>(image_ws)), W, H, &im);
    CP(ranges, im.ranges, sizeof(uint2) * tiles);
    CP(final_T, im.final_T, sizeof(float) * N);
    CP(n_contrib, im.n_contrib, sizeof(uint32_t) * N);
  }
#undef CP
  return GSAJ_OK;
}

int gsaj_debug_export_view_sums(int P, const void *geom_ws, float *sums, void *stream) {
  if (P <= 0 || !geom_ws || !sums) {
    gsaj_set_error("gsaj_debug_export_view_sums: invalid argument");
    return GSAJ_ERR_INVALID_ARGUMENT;
  }
  GeomWS g;
  geom_carve(align_base(const_cast<void *>(geom_ws)), (size_t)P, &g);
  GSAJ_HIP_CHECK(hipMemcpyAsync(sums, g.gsum, sizeof(float4) * 3 * (size_t)P, hipMemcpyDeviceToDevice, (hipStream_t)stream));
  return GSAJ_OK;
}

}  // extern "C"
