// gsaj_common.h -- shared host/device declarations of libgsaj_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "../../include/gsaj.h"

#define GSAJ_WAVE 64
#define TILE GSAJ_TILE
#define TILE_PIXELS (TILE * TILE)
#define PRE_BLOCK 256   // Gaussians per workgroup in the forward per-Gaussian kernels
#define GB_BLOCK 64     // Gaussians per workgroup in the per-Gaussian backward (one wave: spreads P/64 groups over the CUs)
#define REC_F4 3        // float4s per splat row / per-instance gradient row
#define IGRAD_F 12      // floats per per-instance gradient slot (10 used, padded to 3 x float4)

// ---- per-Gaussian splat row (one per Gaussian and view; GeomWS.splat) --------------------
//   r0 = (mean2D.x, mean2D.y, tile rectangle x0 | y0 << 10 | w << 20, first emission slot inside the Gaussian's
//         256-Gaussian block)
//   r1 = (conic.a, conic.b, conic.c, opacity)
//   r2 = (colour.r, colour.g, colour.b, depth)
// The compositors walk a tile's sorted id list (point_list) and GATHER these 48-byte rows -- one 16-byte-aligned row
// instead of the four arrays the reference chases through the index (forward.cu:491-498 / backward.cu:742-752).
// There is no per-(tile, Gaussian) copy of them any more: an opaque scene reaches a tenth of its tile lists, and the
// rows of a frame (2.4 MB at cfg2) stay in the XCD's L2.
// The emission slot u = block_sums[id / 256] + first slot + index of the tile inside the rectangle is where the reverse
// compositor stores this instance's partial gradients, so that the per-Gaussian backward reads each Gaussian's partials
// as one contiguous run.

// ---- workspace layouts ------------------------------------------------------------------
static inline __host__ __device__ size_t gsaj_align(size_t x) { return (x + 255) & ~(size_t)255; }

struct GeomWS {  // per-Gaussian state (reference: GeometryState, rasterizer_impl.h:29-44)
  float *depths;          // [P] view-space depth: the tile sort gathers it by id for its (depth, id) keys
  float *cov3D;           // [P*6] (a batched launch writes view 0's only: Sigma = R S^2 R^T does not depend on the view)
  uint8_t *clamped;       // [P*3]
  uint32_t *tiles_touched;  // [P]
  uint32_t *point_offsets;  // [P] inclusive scan of tiles_touched INSIDE the Gaussian's block of PRE_BLOCK; + block_sums[block] = global
  int *internal_radii;      // [P]
  uint32_t *block_sums;     // [nblk] per-workgroup totals, then (frame scan) their exclusive offsets
  float *tau_partials;      // [nblk*8] per-workgroup dL/dtau partial sums
  float4 *splat;            // [P*3] the per-Gaussian row both compositors gather through the sorted id list (above)
  float4 *splat16;          // [P*2] fp16-storage form of the row (GSAJ_FWD_RECORDS_FP16; written by k_pack_splat16):
                            //   (mean2D.x, mean2D.y, depth, first emission slot) (half2 a b, half2 c o, half2 r g, half2 b 0)
  uint2 *scat;              // [P] what the instance scatter reads: the tile rectangle (x0 | x1 << 16, y0 | y1 << 16); x1 == x0: no tile
  float4 *gsum;             // [P*3] batched backward: the Gaussian's 10 reverse-compositor sums (its instance rows added in
                            //   emission order by k_gather_sums), read by k_chain_window
};

static inline __host__ __device__ size_t geom_carve(char *base, size_t P, GeomWS *g) {
  size_t off = 0;
  size_t nblk = (P + PRE_BLOCK - 1) / PRE_BLOCK;
  if (nblk == 0) nblk = 1;
#define CARVE(field, type, count)                                \
  do {                                                           \
    if (g) g->field = reinterpret_cast<type *>(base + off);      \
    off += gsaj_align(sizeof(type) * (size_t)(count));           \
  } while (0)
  CARVE(depths, float, P);
  CARVE(cov3D, float, P * 6);
  CARVE(clamped, uint8_t, P * 3);
  CARVE(tiles_touched, uint32_t, P);
  CARVE(point_offsets, uint32_t, P);
  CARVE(internal_radii, int, P);
  CARVE(block_sums, uint32_t, nblk);
  CARVE(tau_partials, float, ((P + 31) / 32 + 1) * 8);  // (one slot per 64 Gaussians single view, per 32 batched)
  CARVE(splat, float4, P * 3);
  CARVE(splat16, float4, P * 2);
  CARVE(scat, uint2, P);
  CARVE(gsum, float4, P * 3);
  return off;
}

#define TILE_REP 1        // replicas of each tile's instance counter (1: contention is removed by LDS aggregation)
#define LDS_TILES_MAX 8192  // images with more tiles than this bin with direct global atomics
#define SORT_CAP 16384    // largest tile list sorted in ONE LDS pass (128 KB of the CU's 160 KB LDS); longer lists: LDS-sized chunks + merge passes
#define N_COUNTERS 64     // frame counters: [0] num_rendered [1] error flags [2] longest tile list [3] tau ticket
                          //   [5] (unused)
                          //   [4] abort (async forward: arena too small / tile list too long -> later kernels return)
#define ERR_PREFILTERED 1u
#define ERR_INTERNAL 2u
#define ERR_CAPACITY 4u   // async forward: R exceeds the binning arena the caller provided
#define ERR_TILE_LIST 8u  // (no longer raised: tile lists of any length are sorted on the device)

struct ImageWS {  // reference: ImageState, rasterizer_impl.h:46-53
  float *final_T;         // [H*W]
  uint32_t *n_contrib;    // [H*W]
  uint2 *ranges;          // [tiles]
  uint32_t *counters;     // [N_COUNTERS]            -+
  uint32_t *tile_count;   // [tiles*TILE_REP]         | zeroed by ONE memset per forward
  uint32_t *tile_cursor;  // [tiles*TILE_REP]        -+
  uint32_t *tile_offset;  // [tiles*TILE_REP + 1] exclusive scan of tile_count (tile-major)
  uint32_t *sticky;       // [16] never zeroed by a forward: [0] number of aborted async forwards, [1] tile band, [2] its complement (gsaj_tile_band)
  uint32_t *tile_order;   // [tiles] tile indices, longest list first (frame_scan): the order both compositors take their tiles in
  size_t zero_bytes;      // bytes from counters to the end of tile_cursor
};

static inline __host__ __device__ size_t image_carve(char *base, int W, int H, ImageWS *s) {
  size_t off = 0;
  size_t N = (size_t)W * H;
  size_t tiles = (size_t)((W + TILE - 1) / TILE) * ((H + TILE - 1) / TILE);
  ImageWS *g = s;
  CARVE(final_T, float, N);
  CARVE(n_contrib, uint32_t, N);
  CARVE(ranges, uint2, tiles);
  const size_t z0 = off;
  CARVE(counters, uint32_t, N_COUNTERS);
  CARVE(tile_count, uint32_t, tiles * TILE_REP);
  CARVE(tile_cursor, uint32_t, tiles * TILE_REP);
  if (g) g->zero_bytes = off - z0;
  CARVE(tile_offset, uint32_t, tiles * TILE_REP + 1);
  CARVE(sticky, uint32_t, 16);
  CARVE(tile_order, uint32_t, tiles);
  return off;
}

struct BinWS {  // reference: BinningState, rasterizer_impl.h:55-66
  uint64_t *keys_unsorted;  // [R] the two key buffers of tile lists longer than the LDS sort capacity (chunk sort -> merge passes);
  uint64_t *keys;           // [R]   keys are (depth bits << 32 | id)
  uint32_t *point_list;     // [R] Gaussian ids grouped by tile: as scattered (k_scatter_instances), then sorted in place (k_tile_sort)
  float4 *inst_grad;        // [R*3] per-instance partial gradients (backward), indexed by emission slot
  uint8_t *reached;         // [R] by emission slot: 1 = the reverse compositor wrote that row, 0 = no pixel of the tile got that far
                            //   (zeroed by the tile sort, set by the reverse compositor)
};

// ---- batched multi-view launches (gsaj_rasterize_*_batch): K views of ONE Gaussian map -------------------------------
// Every per-view workspace is one of K consecutive, identically carved blocks; kernels run with gridDim.y = K and shift
// the view-0 pointers by blockIdx.y * stride bytes (wave-uniform: scalar adds).  Single-view launches pass zero strides.
struct ViewStrides {
  size_t geom, image, bin;  // bytes between consecutive views' workspaces
};
#ifdef __HIPCC__
#include <type_traits>
template <typename T>
__device__ __forceinline__ T *gsaj_shift(T *p, size_t bytes) {
  // (pointer arithmetic, not integer arithmetic: through an integer the compiler loses track of where the pointer came from, takes it
  // for a generic pointer, and every load through it becomes a FLAT load -- counted by the LDS counter too, so that a wait for an LDS
  // read also waits for the rows requested chunks ahead)
  typedef typename std::conditional<std::is_const<T>::value, const char, char>::type byte_t;
  return reinterpret_cast<T *>(reinterpret_cast<byte_t *>(p) + bytes);
}
__device__ __forceinline__ GeomWS geom_view(GeomWS g, size_t off) {
  g.depths = gsaj_shift(g.depths, off); g.cov3D = gsaj_shift(g.cov3D, off); g.clamped = gsaj_shift(g.clamped, off);
  g.tiles_touched = gsaj_shift(g.tiles_touched, off); g.point_offsets = gsaj_shift(g.point_offsets, off);
  g.internal_radii = gsaj_shift(g.internal_radii, off); g.block_sums = gsaj_shift(g.block_sums, off);
  g.tau_partials = gsaj_shift(g.tau_partials, off); g.splat = gsaj_shift(g.splat, off); g.gsum = gsaj_shift(g.gsum, off);
  g.splat16 = gsaj_shift(g.splat16, off); g.scat = gsaj_shift(g.scat, off);
  return g;
}
__device__ __forceinline__ ImageWS image_view(ImageWS m, size_t off) {
  m.final_T = gsaj_shift(m.final_T, off); m.n_contrib = gsaj_shift(m.n_contrib, off); m.ranges = gsaj_shift(m.ranges, off);
  m.counters = gsaj_shift(m.counters, off); m.tile_count = gsaj_shift(m.tile_count, off);
  m.tile_cursor = gsaj_shift(m.tile_cursor, off); m.tile_offset = gsaj_shift(m.tile_offset, off);
  m.sticky = gsaj_shift(m.sticky, off); m.tile_order = gsaj_shift(m.tile_order, off);
  return m;
}
#endif

static inline size_t bin_carve(char *base, size_t R, BinWS *g) {
  size_t off = 0;
  size_t Rn = R ? R : 1;
  CARVE(keys_unsorted, uint64_t, Rn);
  CARVE(keys, uint64_t, Rn);
  CARVE(point_list, uint32_t, Rn);
  CARVE(inst_grad, float4, Rn * REC_F4);
  CARVE(reached, uint8_t, Rn);
  return off;
}
#undef CARVE

// ---- error plumbing (api.hip) -------------------------------------------------------------
void gsaj_set_error(const char *fmt, ...);
#define GSAJ_HIP_CHECK(expr)                                                          \
  do {                                                                                \
    hipError_t _e = (expr);                                                           \
    if (_e != hipSuccess) {                                                           \
      gsaj_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
      return GSAJ_ERR_HIP;                                                            \
    }                                                                                 \
  } while (0)

// ---- optional per-stage timing with HIP events on the launch stream (api.hip) --------------
enum GsajStage {
  ST_PREPROCESS = 0, ST_SCAN, ST_EMIT_KEYS, ST_SORT, ST_RANGES_RECORDS, ST_RENDER_FWD, ST_RENDER_BWD, ST_GAUSSIAN_BWD,
  ST_TAU_FINALIZE, ST_DENSE_BWD, ST_DENSE_REDUCE, ST_SCATTER, ST_TILE_SORT, ST_GATHER_SUMS, ST_COUNT
};
void gsaj_prof_mark(int stage, int is_stop, hipStream_t s);
struct GsajProfScope {
  int stage;
  hipStream_t s;
  GsajProfScope(int st, hipStream_t stream) : stage(st), s(stream) { gsaj_prof_mark(stage, 0, s); }
  ~GsajProfScope() { gsaj_prof_mark(stage, 1, s); }
};

// ---- launchers implemented in the .hip files -----------------------------------------------
struct FwdParams {
  int P, D, M, W, H;
  const float *means3D, *shs, *colors_precomp, *opacities, *scales, *rotations, *cov3D_precomp;
  const float *viewmatrix, *projmatrix, *campos;
  float scale_modifier, tanfovx, tanfovy, focal_x, focal_y;
  int prefiltered;
  int grid_x, grid_y;
  int capacity;  // > 0: async forward, binning arena holds this many instances
  int sort_cap;  // async forward: longest tile list the caller sized the LDS sort for (0: SORT_CAP)
  int bpw;       // k_preprocess: blocks of 256 Gaussians per workgroup (set by launch_preprocess)
  int views;     // gridDim.y of the per-view kernels (1: single view); view v reads viewmatrix + 16 v, projmatrix + 16 v,
                 // campos + 3 v and writes radii + P v, n_touched + P v
};

int launch_preprocess(const FwdParams &p, int *radii, int *n_touched, const GeomWS &g, const ImageWS &im, ViewStrides vs,
                      hipStream_t s);
// tile binning of `views` frames: instance scatter -> per-tile (depth, id) sort -> point_list, ranges, cleared `reached` flags.
// sort_cap: LDS sort capacity in keys the caller sizes for (rounded up to a power of two in [128, SORT_CAP]); longer lists take
// the chunk + merge path inside the same kernel.  rec16: also write the fp16-storage rows (GeomWS.splat16).
int launch_tile_binning(int P, int sort_cap, int rec16, int grid_x, int grid_y, const GeomWS &g, const BinWS &b, const ImageWS &im,
                        int views, ViewStrides vs, hipStream_t s);
// out_color [views,3,H,W], out_depth / out_opacity [views,1,H,W], n_touched [views,P]
struct FusedLoss;  // loss_terms.h
// fl != NULL (single view only): the forward's epilogue also sums the loss terms of its pixels into fl->partials
// [gsaj_fwd_loss_slots(W, H)][4]
int launch_render_forward(int P, int W, int H, int grid_x, int grid_y, const float *bg, const GeomWS &g, const BinWS &b,
                          const ImageWS &im, float *out_color, float *out_depth, float *out_opacity, int *n_touched, int views,
                          ViewStrides vs, hipStream_t s, const FusedLoss *fl = nullptr);
int gsaj_fwd_loss_slots(int W, int H);  // workgroups of the forward compositor = loss partial slots
int launch_loss_finalize(const FusedLoss &fl, int nslots, int W, int H, const uint32_t *aborted, float *out_scalars, hipStream_t s);
// dL_dpix [views,3,H,W], dL_dpix_depth [views,1,H,W]
// fl != NULL (single view only): the pixel seeds are derived from fl's images + ground truth instead of read from dL_dpix
int launch_render_backward(int R, int W, int H, int grid_x, int grid_y, const float *bg, const GeomWS &g, const BinWS &b,
                           const ImageWS &im, const float *dL_dpix, const float *dL_dpix_depth, int views, ViewStrides vs,
                           hipStream_t s, const FusedLoss *fl = nullptr);
struct BwdParams {
  int P, D, M, W, H;
  const float *means3D, *shs, *scales, *rotations, *cov3Ds;
  const float *viewmatrix, *projmatrix, *projmatrix_raw, *campos;
  float scale_modifier, tanfovx, tanfovy, focal_x, focal_y;
  int grid_x, grid_y;
  const int *radii;
  float *dL_dmean2D, *dL_dconic, *dL_dopacity, *dL_dcolor, *dL_ddepth, *dL_dmean3D, *dL_dcov3D, *dL_dsh, *dL_dscale,
      *dL_drot, *dL_dtau, *dL_dtau_sum;
};
int launch_gaussian_backward(const BwdParams &p, const GeomWS &g, const BinWS &b, const ImageWS &im, hipStream_t s);
// K views of one map: per-Gaussian parameter gradients SUMMED over the views in a fixed order (p.dL_dmean3D, dL_dcov3D,
// dL_dsh, dL_dscale, dL_drot, dL_dopacity: [P, .]); per-view outputs (any may be NULL): dL_dmean2D [K,P,3], dL_dconic [K,P,4],
// dL_dcolor [K,P,3], dL_ddepth [K,P], dL_dtau [K,P,6]; dL_dtau_sum [K,6].  p.viewmatrix / projmatrix / campos: [K,.];
// p.radii [K,P].
int launch_gather_sums(int P, int K, const int *radii, const GeomWS &g, const BinWS &b, const ImageWS &im, ViewStrides vs, hipStream_t s);
int launch_gaussian_backward_batch(const BwdParams &p, int K, const GeomWS &g, const BinWS &b, const ImageWS &im, ViewStrides vs,
                                   int accumulate, hipStream_t s);
int launch_mark_visible(int P, const float *means3D, const float *viewmatrix, uint8_t *present, hipStream_t s);

// ---- small device helpers -------------------------------------------------------------------
#ifdef __HIPCC__
__device__ __forceinline__ float3 xform4x3(const float *m, float3 p) {
  return make_float3(m[0] * p.x + m[4] * p.y + m[8] * p.z + m[12], m[1] * p.x + m[5] * p.y + m[9] * p.z + m[13],
                     m[2] * p.x + m[6] * p.y + m[10] * p.z + m[14]);
}
__device__ __forceinline__ float4 xform4x4(const float *m, float3 p) {
  return make_float4(m[0] * p.x + m[4] * p.y + m[8] * p.z + m[12], m[1] * p.x + m[5] * p.y + m[9] * p.z + m[13],
                     m[2] * p.x + m[6] * p.y + m[10] * p.z + m[14], m[3] * p.x + m[7] * p.y + m[11] * p.z + m[15]);
}
// pixel = ((ndc + 1) * S - 1) / 2 in double, as the reference's un-suffixed literals do (auxiliary.h:41-44)
__device__ __forceinline__ float ndc2pix(float v, int S) { return (float)((((double)v + 1.0) * (double)S - 1.0) * 0.5); }

// band: the tile rows this frame renders, begin | end << 16 (ImageWS.sticky[1], gsaj_set_tile_band); 0 = all of them.
// A banded frame clips every Gaussian's tile rectangle to rows [begin, end): tiles outside get empty lists, Gaussians
// with no tile inside get radius 0, and every per-Gaussian sum covers the band's tiles only (tile-band sharding of one
// frame over ranks, DESIGN.md "Multi-GPU").
// The band word is honoured only with its complement beside it (sticky[2] == ~sticky[1]): the synchronous entry points do not
// ask for a zeroed image workspace, and uninitialised memory must not pass for a band.
__device__ __forceinline__ uint32_t gsaj_tile_band(const uint32_t *__restrict__ sticky) {
  const uint32_t band = sticky[1];
  return sticky[2] == ~band ? band : 0u;
}
__device__ __forceinline__ void tile_rect(float px, float py, int r, int gx, int gy, uint32_t band, int &x0, int &y0, int &x1,
                                          int &y1) {
  const int ylo = band ? (int)(band & 0xffffu) : 0, yhi = band ? min(gy, (int)(band >> 16)) : gy;
  x0 = min(gx, max(0, (int)((px - (float)r) / (float)TILE)));
  y0 = min(yhi, max(ylo, (int)((py - (float)r) / (float)TILE)));
  x1 = min(gx, max(0, (int)((px + (float)r + (float)(TILE - 1)) / (float)TILE)));
  y1 = min(yhi, max(ylo, (int)((py + (float)r + (float)(TILE - 1)) / (float)TILE)));
}
// ---- the compositors' alpha, ONE definition for the forward and the reverse pass ----------------------------------
// power = -(a dx^2 + c dy^2)/2 - b dx dy is evaluated as log2(e) * power from the pre-scaled conic
// k = (-log2(e)/2 a, -log2(e) b, -log2(e)/2 c) with explicit fused operations, so that both passes get the SAME bits for
// every (pixel, entry) and therefore take the same power <= 0 / alpha >= 1/255 decisions (the reference evaluates the
// identical expression in forward.cu:470-486 and backward.cu:757-766).  G = exp(power) = v_exp_f32(p2).
#define GSAJ_LOG2E 1.4426950408889634f
__device__ __forceinline__ float3 gsaj_prescale_conic(float a, float b, float c) {
  return make_float3((-0.5f * GSAJ_LOG2E) * a, -GSAJ_LOG2E * b, (-0.5f * GSAJ_LOG2E) * c);
}
__device__ __forceinline__ float gsaj_power2(float dx, float dy, float kx, float ky, float kz) {
  const float t = ky * dy;
  const float u = kz * dy;
  return __builtin_fmaf(dx, __builtin_fmaf(kx, dx, t), u * dy);
}
__device__ __forceinline__ float3 cross3(float3 a, float3 b) {
  return make_float3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
#endif

// ---- splat row formats ----------------------------------------------------------------------------------------
// fp32 (default): GeomWS.splat, 3 float4 = {mean x, mean y, rect, first slot | conic a, b, c, opacity | r, g, b, depth}.
// fp16 storage (GSAJ_FWD_RECORDS_FP16; BASELINE config 5 "fp16 splat with fp32 Jacobian accumulation"): GeomWS.splat16, 2 float4 =
// {mean x, mean y, depth, first slot | half2(a, b), half2(c, opacity), half2(r, g), half2(b, 0)} -- positions, depth and
// every accumulation stay fp32; conic / opacity / colour are rounded to half once (k_pack_splat16).  The tile rectangle
// then comes from GeomWS.scat (reverse compositor only).  counters[7] of the frame says which form the frame uses.
#define REC16_F4 2
#ifdef __HIPCC__
#include <hip/hip_fp16.h>
__device__ __forceinline__ uint32_t gsaj_pack_h2(float a, float b) {
  const __half2 h = __floats2half2_rn(a, b);
  return *reinterpret_cast<const uint32_t *>(&h);
}
__device__ __forceinline__ float2 gsaj_unpack_h2(uint32_t u) { return __half22float2(*reinterpret_cast<const __half2 *>(&u)); }
// Gather Gaussian `id`'s row: q0 = (mean x, mean y, depth, first slot), q1 = (conic a, b, c, opacity), q2 = (r, g, b, rect pack
// x0 | y0 << 10 | w << 20 -- fp32 rows only; 0 for fp16 rows, whose rectangle is in GeomWS.scat).
__device__ __forceinline__ void gsaj_load_row(const float4 *__restrict__ splat, const float4 *__restrict__ splat16, uint32_t id,
                                              bool rec16, float4 &q0, float4 &q1, float4 &q2) {
  if (!rec16) {
    const float4 *s = splat + (size_t)id * REC_F4;
    const float4 r0 = s[0], r1 = s[1], r2 = s[2];
    q0 = make_float4(r0.x, r0.y, r2.w, r0.w);
    q1 = r1;
    q2 = make_float4(r2.x, r2.y, r2.z, r0.z);
    return;
  }
  const float4 *s = splat16 + (size_t)id * REC16_F4;
  const float4 p0 = s[0], p1 = s[1];
  const float2 ab = gsaj_unpack_h2(__float_as_uint(p1.x)), co = gsaj_unpack_h2(__float_as_uint(p1.y));
  const float2 rg = gsaj_unpack_h2(__float_as_uint(p1.z)), bz = gsaj_unpack_h2(__float_as_uint(p1.w));
  q0 = p0;
  q1 = make_float4(ab.x, ab.y, co.x, co.y);
  q2 = make_float4(rg.x, rg.y, bz.x, 0.f);
}
#endif

#ifdef __HIPCC__
// Four consecutive words written by OTHER workgroups of this launch (L2 atomics / write-through stores): one 16-byte load that
// bypasses this CU's L1 and this XCD's L2 (sc0 sc1), with its wait inside the statement (the compiler takes an asm's outputs
// for ready when the statement ends).  Coherent loads cost ~100 ns each and do not overlap (tools/chain_trace.py), so the
// "last workgroup sums the partials" tails use as few and as wide ones as they can.
// 16 bytes from a 4-byte-aligned address (global_load_dwordx4 needs no more): one load instruction where the C++ type system would
// want four.
// The pointer is stated to be GLOBAL memory (a flat load is counted by the LDS counter too, so every LDS wait would wait for it).
__device__ __forceinline__ float4 gsaj_load_f4_unaligned(const void *src) {
  typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
  const f4u v = *reinterpret_cast<const __attribute__((address_space(1))) f4u *>((unsigned long long)src);
  return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ uint32_t gsaj_load_u32_global(const uint32_t *src) {
  return *reinterpret_cast<const __attribute__((address_space(1))) uint32_t *>((unsigned long long)src);
}
__device__ __forceinline__ uint4 gsaj_coherent_load_x4(const void *src) {
  uint4 v;
  asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(src) : "memory");
  return v;
}
#endif

// ---- optional per-wave schedule trace (build with -DGSAJ_BLOCK_TRACE; tools/block_trace.py) -----------------
// Each traced kernel records, per wave: start and end of the 100 MHz wall clock, HW_ID and XCC_ID.
#ifdef GSAJ_BLOCK_TRACE
#define GSAJ_TRACE_MAX 65536
#define GSAJ_TRACE_DEFINE(NAME)                                                                               \
  __device__ unsigned long long g_trace_##NAME[4 * GSAJ_TRACE_MAX];                                           \
  extern "C" int gsaj_trace_read_##NAME(unsigned long long *host, int nwaves) {                               \
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_trace_##NAME), sizeof(unsigned long long) * 4 * nwaves); \
  }
#define GSAJ_TRACE_BEGIN(NAME)                                                                                \
  {                                                                                                           \
    const unsigned tw_ = (blockIdx.y * gridDim.x + blockIdx.x) * (blockDim.x >> 6) + (threadIdx.x >> 6);      \
    if ((threadIdx.x & 63) == 0 && tw_ < GSAJ_TRACE_MAX) {                                                    \
      g_trace_##NAME[4 * tw_ + 0] = wall_clock64();                                                           \
      g_trace_##NAME[4 * tw_ + 2] = __builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11));                   \
      g_trace_##NAME[4 * tw_ + 3] = __builtin_amdgcn_s_getreg((20) | (0 << 6) | (31 << 11));                  \
    }                                                                                                         \
  }
#define GSAJ_TRACE_END(NAME)                                                                                  \
  {                                                                                                           \
    const unsigned tw_ = (blockIdx.y * gridDim.x + blockIdx.x) * (blockDim.x >> 6) + (threadIdx.x >> 6);      \
    if ((threadIdx.x & 63) == 0 && tw_ < GSAJ_TRACE_MAX) g_trace_##NAME[4 * tw_ + 1] = wall_clock64();        \
  }
#else
#define GSAJ_TRACE_DEFINE(NAME)
#define GSAJ_TRACE_BEGIN(NAME)
#define GSAJ_TRACE_END(NAME)
#endif
