/*
 * gsaj.h -- C ABI of libgsaj_hip.so: MI355X (gfx950) Gaussian-splat rasteriser with
 * analytical pose Jacobians.
 *
 * These entry points are what the reference's Python binding for this path would bind in
 * place of its pybind/CUDA layer (paths relative to the reference repository):
 *
 *   gsaj_rasterize_forward  / gsaj_forward_*   <- CudaRasterizer::Rasterizer::forward
 *        submodules/diff-gaussian-rasterization/cuda_rasterizer/rasterizer.h:33-58,
 *        called from rasterize_points.cu:36-130 (RasterizeGaussiansCUDA)
 *   gsaj_rasterize_backward                    <- CudaRasterizer::Rasterizer::backward
 *        rasterizer.h:60-91, called from rasterize_points.cu:132-223
 *   gsaj_mark_visible                          <- CudaRasterizer::Rasterizer::markVisible
 *        rasterizer.h:24-31, rasterize_points.cu:225-246
 *   gsaj_*_workspace_bytes                     <- required<GeometryState/ImageState/BinningState>()
 *        rasterizer_impl.h:21-72 (the std::function<char*(size_t)> resize callbacks of
 *        rasterize_points.cu:27-33 become size queries + caller-owned buffers)
 *   gsaj_dense_* / gsaj_pose_jacobians         <- the CPU/NumPy analytic path
 *        Loss_Derivative_script_compare.py:1173-1351 (compute_gradients_2D_vectorized_chunked),
 *        :633-760 (GetAnalyticalJcobian, compute_analytical_jacobians_all_gaussians),
 *        :1587-1695 (dL/dtau assembly)
 *
 * Conventions
 *   - every pointer marked "dev" is a device (HBM) pointer; fp32, contiguous, caller-owned.
 *   - `stream` is a hipStream_t passed as void* (0 = default stream).  All work is
 *     enqueued asynchronously on it; the only host synchronisation is the 4-byte read of
 *     the instance count in gsaj_forward_num_rendered / gsaj_rasterize_forward
 *     (the reference has the same one, rasterizer_impl.cu:331).
 *   - viewmatrix / projmatrix / projmatrix_raw are 16 floats = the transposed 4x4 tensors
 *     the reference hands over (W2C^T, (P W2C)^T, P^T), i.e. column-major W2C / P W2C / P.
 *   - return value: >= 0 on success (gsaj_rasterize_forward returns num_rendered),
 *     a negative GSAJ_ERR_* on failure; gsaj_last_error() describes the last failure of
 *     the calling thread.
 *   - the caller need not zero anything: every output row is written by the kernels (zeros for
 *     culled Gaussians), where the reference's binding zero-fills first (rasterize_points.cu:84-88,175-185).
 */
#ifndef GSAJ_H_INCLUDED
#define GSAJ_H_INCLUDED

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GSAJ_OK 0
#define GSAJ_ERR_INVALID_ARGUMENT (-1)   /* bad shape / null pointer / bad combination */
#define GSAJ_ERR_HIP (-2)                /* a HIP runtime call failed */
#define GSAJ_ERR_WORKSPACE_TOO_SMALL (-3) /* binning workspace smaller than gsaj_binning_workspace_bytes(R) */
#define GSAJ_ERR_PREFILTERED_CULLED (-4) /* prefiltered=1 but a point failed the frustum test (auxiliary.h:156-160) */

#define GSAJ_TILE 16         /* BLOCK_X = BLOCK_Y of config.h:15-17 */
#define GSAJ_NUM_CHANNELS 3  /* NUM_CHANNELS of config.h */

const char *gsaj_last_error(void);
int gsaj_version(void);

/* ---- workspace sizes (bytes) -------------------------------------------------------- */
size_t gsaj_geom_workspace_bytes(int P);
size_t gsaj_image_workspace_bytes(int W, int H);
size_t gsaj_binning_workspace_bytes(int R);

/* ---- per-call flags of the forward entry points (the library keeps NO process-wide mode: two threads may render
 * frames of different formats on different streams at the same time) ---------------------------------------------
 * GSAJ_FWD_RECORDS_FP16: the compositors of THIS frame read 32-byte per-Gaussian splat rows with conic, opacity and colour
 * rounded to half once (positions, depth, every accumulation and every gradient stay fp32) -- BASELINE config 5,
 * "fp16 splat with fp32 Jacobian accumulation"; integer outputs (radii, lists, ranges) are unchanged, images /
 * gradients move by ~1e-3 relative.  Default: 48-byte fp32 rows.  The format is latched in the frame's image
 * workspace, where the matching backward reads it.  (The name is historical: up to round 2 the rows were per-instance
 * records; since the compositors gather per-Gaussian rows the mode saves 16 of 48 bytes per GAUSSIAN, which no longer
 * buys time on MI355X -- see DESIGN.md section 5.) */
#define GSAJ_FWD_RECORDS_FP16 1

/* ---- forward, two-phase form --------------------------------------------------------- */
/* Phase A: per-Gaussian projection (cov3D, EWA cov2D + 0.3, conic, radius, tile rect,
 * SH -> RGB) and the prefix sum of tiles touched.  Writes radii[P] (int32, may be NULL). */
int gsaj_forward_preprocess(int P, int D, int M, int W, int H,
                            const float *means3D /*dev [P,3]*/, const float *shs /*dev [P,M,3] or NULL*/,
                            const float *colors_precomp /*dev [P,3] or NULL*/, const float *opacities /*dev [P]*/,
                            const float *scales /*dev [P,3] or NULL*/, float scale_modifier,
                            const float *rotations /*dev [P,4] or NULL*/, const float *cov3D_precomp /*dev [P,6] or NULL*/,
                            const float *viewmatrix /*dev [16]*/, const float *projmatrix /*dev [16]*/,
                            const float *campos /*dev [3]*/, float tanfovx, float tanfovy, int prefiltered,
                            int *radii /*dev [P] or NULL*/, int *n_touched /*dev [P], zeroed here*/,
                            void *geom_ws /*dev*/, void *image_ws /*dev*/, void *stream);
/* Blocking: number of (Gaussian, tile) instances produced by phase A. */
int gsaj_forward_num_rendered(int W, int H, const void *image_ws, void *stream, int *num_rendered /*host*/,
                              int *max_tile_list /*host, may be NULL: longest per-tile list*/);
/* Phase B: instance scatter into per-tile id lists, per-tile (depth, id) sort in LDS (a list longer than the LDS capacity
 * is sorted in LDS-sized chunks and merged in place by the same workgroup: no list length is refused), front-to-back
 * compositing straight from the sorted id list (the compositor gathers the per-Gaussian 48-byte rows of phase A).
 * out_color [3,H,W], out_depth [1,H,W], out_opacity [1,H,W], n_touched [P] int32.  R must be the value phase A produced. */
int gsaj_forward_render(int P, int R, int max_tile_list /* from phase A: sizes the LDS sort; < 0 forces the chunk + merge path
                                                           (128-key chunks) for every list longer than 128 -- for tests */, int W, int H,
                        const float *bg /*dev [3]*/,
                        const float *colors_precomp /*dev [P,3] or NULL*/, const int *radii /*dev [P] or NULL*/,
                        void *geom_ws, void *binning_ws, size_t binning_ws_bytes, void *image_ws,
                        float *out_color, float *out_depth, float *out_opacity, int *n_touched,
                        int flags /* GSAJ_FWD_*: takes the place of the reference's `bool debug` */, void *stream);

/* ---- forward, one call (phase A, sync, phase B).  The caller supplies a binning
 * workspace of any capacity; if it is too small the call fails with
 * GSAJ_ERR_WORKSPACE_TOO_SMALL and *num_rendered_out holds the R to size it for. */
int gsaj_rasterize_forward(int P, int D, int M, const float *bg, int W, int H,
                           const float *means3D, const float *shs, const float *colors_precomp,
                           const float *opacities, const float *scales, float scale_modifier,
                           const float *rotations, const float *cov3D_precomp,
                           const float *viewmatrix, const float *projmatrix, const float *campos,
                           float tanfovx, float tanfovy, int prefiltered,
                           float *out_color, float *out_depth, float *out_opacity, int *radii, int *n_touched,
                           void *geom_ws, void *binning_ws, size_t binning_ws_bytes, void *image_ws,
                           int *num_rendered_out /*host, may be NULL*/, int flags /* GSAJ_FWD_* */, void *stream);

/* ---- forward without any host synchronisation (tracking / mapping inner loops) -----------------
 * The caller provides a binning workspace sized for `capacity` instances
 * (gsaj_binning_workspace_bytes(capacity)) and passes the SAME capacity as `R` to
 * gsaj_rasterize_backward.  If the frame needs more instances than that, the frame is aborted on the
 * device (every later kernel of the frame returns at once: outputs are the previous frame's) and
 * gsaj_forward_num_rendered -- which may be called at any later time -- returns
 * GSAJ_ERR_WORKSPACE_TOO_SMALL together with the R to size the arena for; the caller then repeats
 * the frame with a larger arena.  tile_list_capacity (0 = the maximum, 16384): the tile-list length the
 * per-tile LDS sort is sized for (rounded up to a power of two >= 128); the sort is given exactly that much
 * shared memory, so scenes with short lists keep more workgroups resident.  It is a performance hint only:
 * a longer list is sorted in chunks of that size and merged (slower, same result) -- never an abort. */
int gsaj_rasterize_forward_async(int P, int D, int M, const float *bg, int W, int H,
                                 const float *means3D, const float *shs, const float *colors_precomp,
                                 const float *opacities, const float *scales, float scale_modifier,
                                 const float *rotations, const float *cov3D_precomp,
                                 const float *viewmatrix, const float *projmatrix, const float *campos,
                                 float tanfovx, float tanfovy, int prefiltered,
                                 float *out_color, float *out_depth, float *out_opacity, int *radii, int *n_touched,
                                 void *geom_ws, void *binning_ws, size_t binning_ws_bytes, int capacity, int tile_list_capacity,
                                 void *image_ws, int flags /* GSAJ_FWD_* */, void *stream);
/* Blocking: number of async forwards aborted on the device since the previous call (read and clear; the first call counts
 * from when the caller zero-filled the image workspace, which it does once, when it allocates it). */
int gsaj_forward_aborted_count(int W, int H, void *image_ws, void *stream, int *count /*host*/);

/* Device address of the frame's abort word inside the image workspace: non-zero after an asynchronous forward that did not fit
 * its arena (every later kernel of that frame returned at once: images, dL/dtau and per-Gaussian outputs are the previous
 * frame's); cleared by the next forward.  Stream-ordered consumers on the device read it without a host round trip:
 * gsaj_pose_adam_step(skip = this) leaves the pose alone after an aborted frame.  NULL on invalid arguments. */
const uint32_t *gsaj_forward_abort_flag(int W, int H, void *image_ws);

/* Tile-band sharding of ONE frame (tracking on several GPUs; no counterpart in the reference, whose rasteriser is
 * single-device: cuda_rasterizer/rasterizer_impl.cu:224-352 binds every tile of the frame).  Every later forward that uses
 * this image workspace renders only tile rows [tile_row_begin, tile_row_end) of the (H + 15) / 16 rows: a Gaussian's tile
 * rectangle (auxiliary.h:46-58 getRect) is clipped to the band, pixels outside it come out as background / zero depth /
 * zero opacity, Gaussians with no tile inside get radii = 0, and the backward returns the band's share of every gradient
 * and of dL/dtau -- the shares of disjoint bands covering the frame add up to the whole-frame result (every gradient is a
 * sum over pixels).  The setting is kept in the image workspace (stream-ordered) until changed; [0, rows) restores the
 * whole frame.  In a batched workspace set it on each view's block.  A workspace this call never touched renders the whole
 * frame whatever its bytes are (the band word is stored with its complement and ignored unless both agree). */
int gsaj_set_tile_band(int W, int H, void *image_ws, int tile_row_begin, int tile_row_end, void *stream);
/* gsaj_forward_preprocess with the arena capacity check armed (capacity = 0: unchecked). */
int gsaj_forward_preprocess_cap(int P, int D, int M, int W, int H,
                                const float *means3D, const float *shs, const float *colors_precomp,
                                const float *opacities, const float *scales, float scale_modifier,
                                const float *rotations, const float *cov3D_precomp,
                                const float *viewmatrix, const float *projmatrix, const float *campos,
                                float tanfovx, float tanfovy, int prefiltered, int *radii, int *n_touched,
                                void *geom_ws, void *image_ws, int capacity, int tile_list_capacity, void *stream);

/* ---- backward ------------------------------------------------------------------------ */
/* dL_dpix [3,H,W], dL_dpix_depth [1,H,W] -> dL_dmean2D [P,3] (NDC-scaled, z unused),
 * dL_dconic [P,2,2] (slots 0,1,3), dL_dopacity [P], dL_dcolor [P,3], dL_ddepth [P],
 * dL_dmean3D [P,3], dL_dcov3D [P,6], dL_dsh [P,M,3], dL_dscale [P,3], dL_drot [P,4],
 * dL_dtau [P,6] (may be NULL) and dL_dtau_sum [6] = sum over Gaussians, tau = [rho, theta]
 * (may be NULL; replaces torch.sum in diff_gaussian_rasterization/__init__.py:162).
 * Pose-only mode (tracking, where only the camera is optimised): pass NULL for ALL ten per-Gaussian outputs
 * dL_dmean2D .. dL_drot and a dL_dtau_sum; the per-Gaussian parameter gradients are then neither finished nor stored.
 * The three workspaces must be the ones the forward filled. */
int gsaj_rasterize_backward(int P, int D, int M, int R, const float *bg, int W, int H,
                            const float *means3D, const float *shs, const float *colors_precomp,
                            const float *scales, float scale_modifier, const float *rotations,
                            const float *cov3D_precomp, const float *viewmatrix, const float *projmatrix,
                            const float *projmatrix_raw, const float *campos, float tanfovx, float tanfovy,
                            const int *radii, void *geom_ws, void *binning_ws, void *image_ws,
                            const float *dL_dpix, const float *dL_dpix_depth,
                            float *dL_dmean2D, float *dL_dconic, float *dL_dopacity, float *dL_dcolor,
                            float *dL_ddepth, float *dL_dmean3D, float *dL_dcov3D, float *dL_dsh,
                            float *dL_dscale, float *dL_drot, float *dL_dtau, float *dL_dtau_sum, void *stream);

/* ---- batched multi-view entry points: K views of ONE Gaussian map ------------------------------------------------
 * The mapping step of the reference renders every keyframe of the window against the same Gaussians, sums the losses and
 * back-propagates once: per-Gaussian gradients ACCUMULATE over the keyframes, every keyframe keeps its own dL/dtau
 * (utils/slam_backend.py:168-232).  These two calls do that for K views in one set of launches (grids x K): the Gaussians
 * are read once per kernel where the view does not matter, the per-Gaussian parameter gradients are summed over the K views
 * inside the kernel in view order (deterministic) and written once, and K rows of dL/dtau come out.
 *
 * Per-view arrays are K consecutive blocks: viewmatrices / projmatrices [K,16], campos [K,3], out_color [K,3,H,W], out_depth
 * / out_opacity [K,1,H,W], radii / n_touched [K,P], dL_dpix [K,3,H,W], dL_dpix_depth [K,1,H,W].  All K views share W, H,
 * tanfov (one camera model) and projmatrix_raw.  Workspaces are K consecutive blocks of gsaj_geom_workspace_bytes(P),
 * gsaj_image_workspace_bytes(W, H) and gsaj_binning_workspace_bytes(capacity) bytes, 256-byte aligned, the image workspaces
 * zeroed once by the caller; view v's block can be handed to gsaj_forward_num_rendered / gsaj_forward_aborted_count /
 * gsaj_debug_export on its own.  Like gsaj_rasterize_forward_async there is NO host synchronisation: `capacity` instances
 * per view, a view that needs more is aborted on the device, contributes nothing to the sums, and is reported by
 * gsaj_forward_num_rendered(view block); tile_list_capacity as in gsaj_rasterize_forward_async (a hint, never an abort).
 *
 * Backward outputs.  Summed over the views: dL_dopacity [P], dL_dmean3D [P,3], dL_dcov3D [P,6], dL_dsh [P,M,3], dL_dscale
 * [P,3], dL_drot [P,4].  Per view (each may be NULL): dL_dmean2D [K,P,3] (what densification reads as
 * viewspace_points.grad, gaussian_model.py:767-771), dL_dconic [K,P,2,2], dL_dcolor [K,P,3], dL_ddepth [K,P], dL_dtau [K,P,6];
 * and dL_dtau_sum [K,6].  SH storage of 1, 4, 9 or 16 coefficients. */
int gsaj_rasterize_forward_batch(int K, int P, int D, int M, const float *bg, int W, int H, const float *means3D,
                                 const float *shs, const float *colors_precomp, const float *opacities, const float *scales,
                                 float scale_modifier, const float *rotations, const float *cov3D_precomp,
                                 const float *viewmatrices, const float *projmatrices, const float *campos, float tanfovx,
                                 float tanfovy, int prefiltered, float *out_color, float *out_depth, float *out_opacity,
                                 int *radii, int *n_touched, void *geom_ws, void *binning_ws, size_t binning_ws_bytes,
                                 int capacity, int tile_list_capacity, void *image_ws, int flags /* GSAJ_FWD_* */, void *stream);
int gsaj_rasterize_backward_batch(int K, int P, int D, int M, int capacity, const float *bg, int W, int H,
                                  const float *means3D, const float *shs, const float *colors_precomp, const float *scales,
                                  float scale_modifier, const float *rotations, const float *cov3D_precomp,
                                  const float *viewmatrices, const float *projmatrices, const float *projmatrix_raw,
                                  const float *campos, float tanfovx, float tanfovy, const int *radii, void *geom_ws,
                                  void *binning_ws, void *image_ws, const float *dL_dpix, const float *dL_dpix_depth,
                                  float *dL_dmean2D, float *dL_dconic, float *dL_dopacity, float *dL_dcolor, float *dL_ddepth,
                                  float *dL_dmean3D, float *dL_dcov3D, float *dL_dsh, float *dL_dscale, float *dL_drot,
                                  float *dL_dtau, float *dL_dtau_sum, int flags /* GSAJ_BWD_* */, void *stream);
/* flags of gsaj_rasterize_backward_batch:
 * GSAJ_BWD_ACCUMULATE      the summed per-Gaussian outputs are ADDED to what their buffers hold instead of overwriting them:
 *                          a window processed in several calls (views [0,K0) then [K0,K)), or the keyframes of several windows
 *                          accumulated on one rank before the optimiser step (utils/slam_backend.py:168-232: current window +
 *                          2 random older keyframes, ONE backward).  Calls add in the caller's order: reproducible.
 * GSAJ_BWD_ONLY_COMPOSITE  run only the per-view half (reverse compositor + per-Gaussian gather of its partial sums);
 * GSAJ_BWD_ONLY_CHAIN      run only the per-Gaussian chain on views whose per-view half has already run.
 *                          Together they let a caller put the per-view halves of two view groups on two HIP streams (they are
 *                          independent) and serialise only the accumulating chains (gsaj.rasterizer.BatchContext(streams=2)). */
#define GSAJ_BWD_ACCUMULATE 1
#define GSAJ_BWD_ONLY_COMPOSITE 2
#define GSAJ_BWD_ONLY_CHAIN 4

/* ---- frustum test -------------------------------------------------------------------- */
int gsaj_mark_visible(int P, const float *means3D, const float *viewmatrix, const float *projmatrix,
                      uint8_t *present /*dev [P]*/, void *stream);

/* ---- introspection of the forward state (parity tests, debugging) -------------------- */
/* Copies internal arrays to caller-provided DEVICE buffers; any pointer may be NULL.
 * means2D [P,2], depths [P], cov3D [P,6], conic_opacity [P,4], rgb [P,3], clamped [P,3] u8,
 * tiles_touched [P] u32, point_list [R] u32, ranges [tiles,2] u32, final_T [H,W], n_contrib [H,W] u32. */
int gsaj_debug_export(int P, int R, int W, int H, const void *geom_ws, const void *binning_ws, const void *image_ws,
                      float *means2D, float *depths, float *cov3D, float *conic_opacity, float *rgb, uint8_t *clamped,
                      uint32_t *tiles_touched, uint32_t *point_list, uint32_t *ranges, float *final_T,
                      uint32_t *n_contrib, void *stream);

/* After gsaj_rasterize_backward_batch: the reverse compositor's 10 sums per Gaussian of ONE view of the window (that view's block
 * of the geometry workspace), sums [P,12] = (dL/dmean2D x, y | dL/dconic a, b, c | dL/dopacity | dL/dcolor r, g, b | dL/ddepth | 2
 * pads) -- the per-view quantities the batched backward does not return (it returns their sums over the views), for parity tests. */
int gsaj_debug_export_view_sums(int P, const void *geom_ws, float *sums /*dev [P,12]*/, void *stream);

/* ---- per-kernel timing (bench.py's roofline leg) ----------------------------------------
 * Between gsaj_profile_begin and gsaj_profile_end every kernel launch of the library is
 * bracketed by HIP events on the stream it is launched on.  gsaj_profile_end synchronises,
 * and returns per stage the summed duration in ms and the number of launches.
 * Stage order: GSAJ_STAGE_NAMES.  (scan_blocks, emit_keys, sort, ranges_records and tau_finalize belong to kernels that no
 * longer exist; the slots are kept so that the indices of the others do not move, and report 0 launches.  tile_sort_records
 * is k_tile_sort -- it sorts ids now, there are no per-instance records.) */
#define GSAJ_NUM_STAGES 14
#define GSAJ_STAGE_NAMES "preprocess,scan_blocks,emit_keys,sort,ranges_records,render_fwd,render_bwd,gaussian_bwd,tau_finalize,dense_bwd,dense_reduce,scatter_instances,tile_sort_records,gather_sums"
int gsaj_profile_begin(int max_records);
int gsaj_profile_end(float *stage_ms /*host [GSAJ_NUM_STAGES]*/, int *stage_launches /*host [GSAJ_NUM_STAGES]*/);

/* ---- losses + pixel-gradient seeds ---------------------------------------------
 * Replaces the ~15 full-frame torch kernels + autograd of get_loss_tracking / get_loss_mapping
 * (reference utils/slam_utils.py:56-128) by one pass: reads color [3,H,W], depth [1,H,W], opacity [1,H,W], the
 * ground truth gt_color [3,H,W], gt_depth [H,W] (RGB-D only) and the optional tracking grad_mask [H,W] (bytes,
 * non-zero = keep; slam_utils.py:69), the exposure scalars a, b on the device (ignored with NO_EXPOSURE, i.e.
 * get_loss_mapping(initialization=True)), and writes dL/dcolor [3,H,W], dL/ddepth [1,H,W] -- the two inputs of
 * gsaj_rasterize_backward -- optionally dL/dopacity [1,H,W] (the reference's rasteriser ignores it), and
 * out_scalars[5] = {loss, L_rgb, L_depth, dL/da, dL/db} on the device.  Deterministic (no float atomics).
 * loss_ws: gsaj_loss_workspace_bytes(W, H) bytes, ZEROED ONCE by the caller when allocated. */
#define GSAJ_LOSS_TRACKING 1    /* opacity weight, grad_mask, opacity > 0.95 depth gate (get_loss_tracking*) */
#define GSAJ_LOSS_MONOCULAR 2   /* config["Training"]["monocular"]: colour term only */
#define GSAJ_LOSS_NO_EXPOSURE 4 /* image_ab = image (get_loss_mapping(initialization=True)) */
#define GSAJ_LOSS_COMPUTE_LOSS 8 /* compute_loss of the verification harness (Jacobian_test.py:155-196, compare.py:144-185):
                                  * grad_mask = the per-pixel mask; colour = mean over 3HW of |color*mask - gt*mask|; depth = mean
                                  * over the pixels with gt_depth > 0 inside the mask of |depth - gt|; loss = their plain sum
                                  * (alpha, rgb_boundary_threshold, exposure ignored; with MONOCULAR: colour term only).  The
                                  * 10 x isotropic term of compute_loss is per Gaussian: gsaj_isotropic_loss. */
size_t gsaj_loss_workspace_bytes(int W, int H);
int gsaj_loss_seeds(int W, int H, int flags, float alpha, float rgb_boundary_threshold, const float *color,
                    const float *depth, const float *opacity, const float *gt_color, const float *gt_depth,
                    const uint8_t *grad_mask, const float *exposure_a, const float *exposure_b, float *dL_dcolor,
                    float *dL_ddepth, float *dL_dopacity, float *out_scalars, void *loss_ws, void *stream);

/* ---- the same losses FUSED into the compositors (SURVEY 8(f)-1 as written): no seed image, no pass over the frame ------------
 * The reference evaluates get_loss_tracking / get_loss_mapping between render() and backward() (utils/slam_frontend.py:164-176,
 * utils/slam_utils.py:56-128).  gsaj_rasterize_forward_loss = gsaj_rasterize_forward_async whose compositor epilogue also sums the
 * loss terms of its pixels against the ground truth; out_scalars[5] = {loss, L_rgb, L_depth, dL/da, dL/db} (device) are there when
 * the call's work has run.  gsaj_rasterize_backward_loss = gsaj_rasterize_backward whose reverse compositor derives each pixel's
 * seeds dL/dC, dL/dD from the images the forward wrote (color, depth, opacity: pass them back), the ground truth and the exposure
 * scalars, with the arithmetic of gsaj_loss_seeds bit for bit -- the gradients equal those of forward -> gsaj_loss_seeds ->
 * backward exactly; the loss scalars differ from gsaj_loss_seeds' only by the order of their sums.  loss_flags: GSAJ_LOSS_TRACKING,
 * _MONOCULAR, _NO_EXPOSURE (GSAJ_LOSS_COMPUTE_LOSS has no fused form).  loss_ws: gsaj_fused_loss_workspace_bytes(W, H) bytes, no
 * initialisation needed.  An aborted frame leaves out_scalars as they were. */
size_t gsaj_fused_loss_workspace_bytes(int W, int H);
int gsaj_rasterize_forward_loss(int P, int D, int M, const float *bg, int W, int H, const float *means3D, const float *shs,
                                const float *colors_precomp, const float *opacities, const float *scales, float scale_modifier,
                                const float *rotations, const float *cov3D_precomp, const float *viewmatrix,
                                const float *projmatrix, const float *campos, float tanfovx, float tanfovy, int prefiltered,
                                float *out_color, float *out_depth, float *out_opacity, int *radii, int *n_touched, void *geom_ws,
                                void *binning_ws, size_t binning_ws_bytes, int capacity, int tile_list_capacity, void *image_ws,
                                int flags /* GSAJ_FWD_* */, int loss_flags /* GSAJ_LOSS_* */, float alpha,
                                float rgb_boundary_threshold, const float *gt_color /*dev [3,H,W]*/,
                                const float *gt_depth /*dev [H,W] or NULL (monocular)*/, const uint8_t *grad_mask /*dev [H,W] or NULL*/,
                                const float *exposure_a, const float *exposure_b /*dev scalars; NULL with NO_EXPOSURE*/,
                                float *out_scalars /*dev [5]*/, void *loss_ws, void *stream);
int gsaj_rasterize_backward_loss(int P, int D, int M, int R, const float *bg, int W, int H, const float *means3D, const float *shs,
                                 const float *colors_precomp, const float *scales, float scale_modifier, const float *rotations,
                                 const float *cov3D_precomp, const float *viewmatrix, const float *projmatrix,
                                 const float *projmatrix_raw, const float *campos, float tanfovx, float tanfovy, const int *radii,
                                 void *geom_ws, void *binning_ws, void *image_ws, int loss_flags, float alpha,
                                 float rgb_boundary_threshold, const float *color, const float *depth, const float *opacity,
                                 const float *gt_color, const float *gt_depth, const uint8_t *grad_mask, const float *exposure_a,
                                 const float *exposure_b, float *dL_dmean2D, float *dL_dconic, float *dL_dopacity, float *dL_dcolor,
                                 float *dL_ddepth, float *dL_dmean3D, float *dL_dcov3D, float *dL_dsh, float *dL_dscale,
                                 float *dL_drot, float *dL_dtau, float *dL_dtau_sum, void *stream);

/* The same for the K views of a mapping window in ONE launch (utils/slam_backend.py:168-232 sums get_loss_mapping over the
 * keyframes of the window): color / gt_color / dL_dcolor [K,3,H,W], depth / opacity / dL_ddepth / dL_dopacity [K,1,H,W], gt_depth /
 * grad_mask [K,H,W], exposure_a / exposure_b [K] (one pair per keyframe, camera_utils.py:43-48), out_scalars [K,5]; view k gets
 * exactly what gsaj_loss_seeds gives for its slices.  loss_ws: K blocks of gsaj_loss_workspace_bytes(W, H) rounded up to 256 bytes,
 * 256-byte aligned, zeroed once.  GSAJ_LOSS_COMPUTE_LOSS has no batched form. */
int gsaj_loss_seeds_batch(int K, int W, int H, int flags, float alpha, float rgb_boundary_threshold, const float *color,
                          const float *depth, const float *opacity, const float *gt_color, const float *gt_depth,
                          const uint8_t *grad_mask, const float *exposure_a, const float *exposure_b, float *dL_dcolor,
                          float *dL_ddepth, float *dL_dopacity, float *out_scalars, void *loss_ws, void *stream);

/* weight * mean |s_ij - mean_j(s_i.)| over scales [P,C] (C = 1..3) -> out_loss[0] (device), and its gradient into dL_dscales
 * [P,C] (may be NULL; accumulate != 0: added to what is there): the isotropic regulariser of compute_loss (weight 10,
 * Jacobian_test.py:169-171) and of the mapping loss (slam_backend.py:229-231).  iso_ws: gsaj_isotropic_workspace_bytes(P),
 * ZEROED ONCE by the caller when allocated.  Deterministic. */
size_t gsaj_isotropic_workspace_bytes(int P);
int gsaj_isotropic_loss(int P, int C, float weight, const float *scales, float *dL_dscales, int accumulate, float *out_loss,
                        void *iso_ws, void *stream);

/* ---- densification / pruning bookkeeping ----------------------------------------------------------
 * What the reference's mapping loop does per rendered view after the backward (utils/slam_backend.py:113-121, 276-285):
 *   vis = radii > 0;  max_radii2D[vis] = max(max_radii2D[vis], radii[vis]);
 *   xyz_gradient_accum[vis] += ||viewspace_points.grad[vis, :2]||;  denom[vis] += 1        (gaussian_model.py:767-771)
 * and, for pruning, n_obs = number of views that touched the Gaussian (n_touched > 0; slam_backend.py:236-250), for the K
 * views of a window in ONE launch (K = 1: one view).  dL_dmean2D [K,P,3] (the backward's per-view output), radii [K,P],
 * n_touched [K,P] (may be NULL); xyz_gradient_accum [P], denom [P], max_radii2D [P] are updated in place (each may be NULL),
 * n_obs [P] int32 is written (may be NULL). */
int gsaj_densification_stats(int K, int P, const float *dL_dmean2D, const int *radii, const int *n_touched,
                             float *xyz_gradient_accum, float *denom, float *max_radii2D, int *n_obs, void *stream);

/* ---- tracking pose step on the device -------------------------------------------
 * One launch = torch.optim.Adam.step() on (cam_trans_delta, cam_rot_delta, exposure_a, exposure_b) as set up in
 * slam_frontend.py:135-160 + update_pose (reference utils/pose_utils.py:76-93) + the camera matrices of
 * camera_utils.py:95-109, with no host read-back.  dL_dtau = [rho(3), theta(3)] is the dL_dtau_sum of
 * gsaj_rasterize_backward; dL_dexposure = {dL/da, dL/db} (out_scalars + 3 of gsaj_loss_seeds) or NULL.
 * projection_matrix = the camera's projection_matrix (P^T, row-major [16]) or NULL.
 * pose_state: GSAJ_POSE_STATE_FLOATS device floats owned by the caller:
 *   [0:16)  W2C row-major (in: current pose; out: Exp(tau) * W2C)      [16:24) Adam m   [24:32) Adam v
 *   [32]    step count    [33:35) exposure a, b (updated in place)
 *   [35:51) out world_view_transform = W2C^T     [51:67) out full_proj_transform    [67:70) out camera_center
 *   [70:76) out tau = [rho, theta] applied       [76] out |tau|     [77] out converged (1.0 / 0.0: |tau| < threshold)
 * Initialise [0:16) with the pose and zero the rest.
 * skip (device, may be NULL): if the 32-bit word it points to is non-zero the call changes NOTHING (no Adam moment, no step
 * count, no pose): pass gsaj_forward_abort_flag() of the frame the gradients come from -- an aborted asynchronous frame leaves
 * the previous iteration's dL/dtau in place -- or, with the frame sharded over ranks, a word of the all-reduced buffer that is
 * non-zero when ANY rank's share was aborted (any non-zero bit pattern counts, e.g. a positive float). */
#define GSAJ_POSE_STATE_FLOATS 80
int gsaj_pose_state_floats(void);
int gsaj_pose_adam_step(const float *dL_dtau, const float *dL_dexposure, float lr_rot, float lr_trans, float lr_exp_a,
                        float lr_exp_b, float beta1, float beta2, float eps, float converged_threshold,
                        const float *projection_matrix, float *pose_state, const uint32_t *skip, void *stream);

/* The same step for K poses in one launch (the keyframe poses of a mapping window, each with its own Adam state:
 * utils/slam_backend.py:255-262 steps the keyframe optimiser and calls update_pose per keyframe, skipping uid 0):
 * dL_dtau [K,6] (the dL_dtau_sum rows of gsaj_rasterize_backward_batch), dL_dexposure [K,2] or NULL, active [K] bytes or NULL
 * (0: the pose and its Adam state are left untouched), pose_states [K, GSAJ_POSE_STATE_FLOATS]; the learning rates are shared.
 * skip (may be NULL) / skip_stride_bytes: pose k is left untouched if the word at skip + k * skip_stride_bytes is non-zero -- with
 * skip = gsaj_forward_abort_flag(view 0's image workspace) and skip_stride_bytes = gsaj_image_workspace_bytes(W, H), the views of a
 * batched window that were aborted on the device. */
int gsaj_pose_adam_step_batch(int K, const float *dL_dtau, const float *dL_dexposure, const uint8_t *active, float lr_rot,
                              float lr_trans, float lr_exp_a, float lr_exp_b, float beta1, float beta2, float eps,
                              float converged_threshold, const float *projection_matrix, float *pose_states, const uint32_t *skip,
                              size_t skip_stride_bytes, void *stream);

/* ---- distCUDA2 -------------------------------------------------------------------
 * simple_knn._C.distCUDA2 (reference submodules/simple-knn/simple_knn.cu:45-220, spatial.cu): for P points
 * [P,3] the mean of the squared distances to the 3 nearest other points -> mean_dists [P].  Exact search;
 * fewer than 3 neighbours leaves FLT_MAX terms (-> inf) as in the reference.  No host synchronisation.
 * knn_ws: gsaj_dist2_workspace_bytes(P) bytes. */
size_t gsaj_dist2_workspace_bytes(int P);
int gsaj_dist2(int P, const float *points, float *mean_dists, void *knn_ws, void *stream);
/* Tests only: the Morton order gsaj_dist2 left in its workspace -- the 30-bit codes in sorted order and the point index of every
 * position (device arrays [P]; blocking).  The sort is this library's own stable radix sort (thrust::sort_by_key in the reference,
 * simple_knn.cu:211): ascending codes, equal codes in ascending index order. */
int gsaj_debug_dist2_order(int P, void *knn_ws, uint32_t *codes_sorted, uint32_t *idx_sorted, void *stream);

/* ---- dense analytic path (NumPy-path semantics) ------------------ */
size_t gsaj_dense_workspace_bytes(int N, int W, int H);
/* N depth-sorted Gaussians: means2D [N,2] (pixels), covs2D [N,2,2], colors [N,3], depths [N], opac [N];
 * per-pixel seeds seed_color [H,W,3], seed_depth [H,W]  ->
 * grad_mu [N,2], grad_Sigma [N,2,2], grad_depth [N], grad_color [N,3]. */
/* flags: GSAJ_DENSE_NAIVE_GUARDS selects the edge semantics of the naive per-pixel loop
 * (Loss_Derivative_wrt_mu_and_cov.py:3-118 = compare.py:1050-1169) instead of the vectorised golden producer's (:1311): where
 * alpha_i >= 0.999 the suffix term is dropped rather than divided by 1.0, and an entry with abs(alpha_i) < 1e-8 adds nothing to
 * grad_mu / grad_Sigma. */
#define GSAJ_DENSE_NAIVE_GUARDS 1
/* GSAJ_DENSE_NORMALISED_COORDS: the variant of Loss_Derivative_script.py:820-979 -- means2D / covs2D are in NORMALISED image
 * coordinates and pixel (col, row) sits at ((col - cx) / fx, (row - cy) / fy) (float64 arithmetic rounded to float32 as in
 * :873-877, where the reference reads the module globals cx, fx, cy, fy); `intrinsics` = host {fx, fy, cx, cy}, read only
 * with this flag (NULL otherwise).  Same arithmetic; the gradients come out in normalised units. */
#define GSAJ_DENSE_NORMALISED_COORDS 2
int gsaj_dense_backward(int N, int W, int H, const float *means2D, const float *covs2D, const float *colors,
                        const float *depths, const float *opac, const float *seed_color, const float *seed_depth,
                        float *grad_mu, float *grad_Sigma, float *grad_depth, float *grad_color,
                        void *dense_ws, int flags, const double *intrinsics /*host [4] or NULL*/, void *stream);
/* Front end of the NumPy path (GetImagePlaneMeanAndCovs + compute_cov2d + ndc2Pix + compute_colors_from_sh +
 * OrderGaussiansByDepth, compare.py:854-971, 772-852, 535-588, 764-769), fp64 arithmetic on the fp32 inputs like the
 * reference's Python floats, Appendix A.4 semantics: no z <= 0.2 cull, no tile test, colours clamped below at 0 only, one
 * global STABLE order by view-space z.  means3D [N,3], cov3D [N,6] (xx,xy,xz,yy,yz,zz), shs [N,sh_coeffs,3]; viewmatrix /
 * projmatrix as for the rasteriser (W2C^T, (P W2C)^T); campos [3].  Outputs (device): order [N] int32 (order[i] = original
 * index of sorted position i) and, IN SORTED ORDER, mean2D [N,2] pixels, cov2D [N,2,2] pixels^2 (+0.3 dilation), color [N,3],
 * color_raw [N,3] (before the clamp; may be NULL), depth [N].  project_ws: gsaj_dense_project_workspace_bytes(N). */
size_t gsaj_dense_project_workspace_bytes(int N);
int gsaj_dense_project(int N, int sh_coeffs, int sh_degree, int W, int H, const float *means3D, const float *cov3D,
                       const float *shs, const float *viewmatrix, const float *projmatrix, const float *campos, double fx,
                       double fy, int *order, double *mean2D, double *cov2D, double *color, double *color_raw, double *depth,
                       void *project_ws, void *stream);
/* Dense forward compositor: out_color [H,W,3], out_depth [H,W]. */
int gsaj_dense_render(int N, int W, int H, const float *means2D, const float *covs2D, const float *colors,
                      const float *depths, const float *opac, float *out_color, float *out_depth, void *stream);
/* Closed-form d(mu_I)/d(tau) [N,2,6] and d(vec Sigma_I)/d(tau) [N,4,6] (fp64), already
 * scaled to NDC / pixel^2 units like compute_analytical_jacobians_all_gaussians.
 * T_cw: 16 doubles row-major; mu_w [N,3] doubles; cov3D [N,6] doubles. */
int gsaj_pose_jacobians(int N, const double *T_cw /*dev*/, const double *mu_w, const double *cov3D, double fx,
                        double fy, int W, int H, double *dmu_dtau, double *dcov_dtau, void *stream);
/* dL/dtau (6 doubles, dev) of the NumPy path: chain rule over sorted Gaussians
 * (order[i] = original index of sorted position i) including depth and SH view-direction terms. */
int gsaj_dense_tau(int N, int sh_coeffs, int sh_degree, const int *order, const float *grad_mu,
                   const float *grad_Sigma, const float *grad_depth, const float *grad_color,
                   const double *dmu_dtau, const double *dcov_dtau, const double *mu_w, const double *T_cw,
                   const double *campos, const double *shs /*[N,sh_coeffs,3]*/, double *dL_dtau /*dev [6]*/,
                   double *parts /*dev [4,6] mu,cov,depth,sh or NULL*/, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* GSAJ_H_INCLUDED */
