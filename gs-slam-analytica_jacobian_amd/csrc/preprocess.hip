// preprocess.hip -- per-Gaussian forward stage and instance-key emission (gfx950).
//
// One lane per Gaussian, SoA streaming loads/stores (HBM-bound stage, SURVEY 8d regime 1).
// The workgroup also produces the block-local inclusive scan of tiles_touched, so the
// prefix sum over all Gaussians costs no extra pass over P (the reference runs a separate
// cub::DeviceScan over tiles_touched, rasterizer_impl.cu:327).
//
// Compiled with -ffp-contract=off: radius and tile rectangle are integers derived from
// fp32 arithmetic, and must not depend on FMA contraction.
//
// Behaviour follows forward.cu:157-401 (preprocessCUDA), :76-115 (computeCov2D),
// :120-154 (computeCov3D), :22-73 (computeColorFromSH), auxiliary.h:139-164 (in_frustum),
// rasterizer_impl.cu:70-111 (duplicateWithKeys), :54-66 (checkFrustum).
#include "gsaj_common.h"
#include "wave_reduce.h"

__constant__ float kSH_C0 = 0.28209479177387814f;
__constant__ float kSH_C1 = 0.4886025119029199f;
__constant__ float kSH_C2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f, -1.0925484305920792f,
                                0.5462742152960396f};
__constant__ float kSH_C3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f, 0.3731763325901154f,
                                -0.4570457994644658f, 1.445305721320277f, -0.5900435899266435f};

__device__ __forceinline__ float3 ld3(const float *p, size_t i) { return make_float3(p[3 * i], p[3 * i + 1], p[3 * i + 2]); }

// Sigma = R S^2 R^T (upper triangle) from scale and an un-normalised quaternion (r,x,y,z).
__device__ __forceinline__ void cov3d_from_scale_rot(float3 sc, float mod, float4 q, float *c6) {
  const float r = q.x, x = q.y, y = q.z, z = q.w;
  float R[3][3] = {{1.f - 2.f * (y * y + z * z), 2.f * (x * y - r * z), 2.f * (x * z + r * y)},
                   {2.f * (x * y + r * z), 1.f - 2.f * (x * x + z * z), 2.f * (y * z - r * x)},
                   {2.f * (x * z - r * y), 2.f * (y * z + r * x), 1.f - 2.f * (x * x + y * y)}};
  const float s[3] = {mod * sc.x, mod * sc.y, mod * sc.z};
  float A[3][3];  // A = S R^T
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) A[i][j] = s[i] * R[j][i];
  float S[3][3];
#pragma unroll
  for (int a = 0; a < 3; a++)
#pragma unroll
    for (int b = a; b < 3; b++) S[a][b] = A[0][a] * A[0][b] + A[1][a] * A[1][b] + A[2][a] * A[2][b];
  c6[0] = S[0][0]; c6[1] = S[0][1]; c6[2] = S[0][2]; c6[3] = S[1][1]; c6[4] = S[1][2]; c6[5] = S[2][2];
}

// Dilated EWA covariance (a,b,c) of the projected Gaussian.
__device__ __forceinline__ float3 cov2d_forward(float3 mean, float fx, float fy, float tanx, float tany, const float *c6,
                                                const float *vm) {
  float3 t = xform4x3(vm, mean);
  const float limx = 1.3f * tanx, limy = 1.3f * tany;
  const float txtz = t.x / t.z, tytz = t.y / t.z;
  t.x = fminf(limx, fmaxf(-limx, txtz)) * t.z;
  t.y = fminf(limy, fmaxf(-limy, tytz)) * t.z;
  const float J00 = fx / t.z, J02 = -(fx * t.x) / (t.z * t.z);
  const float J11 = fy / t.z, J12 = -(fy * t.y) / (t.z * t.z);
  float M[2][3];
#pragma unroll
  for (int k = 0; k < 3; k++) {  // M = J * Rcw, Rcw[r][k] = vm[4k + r]
    M[0][k] = vm[4 * k + 0] * J00 + vm[4 * k + 1] * 0.0f + vm[4 * k + 2] * J02;
    M[1][k] = vm[4 * k + 0] * 0.0f + vm[4 * k + 1] * J11 + vm[4 * k + 2] * J12;
  }
  const float V[3][3] = {{c6[0], c6[1], c6[2]}, {c6[1], c6[3], c6[4]}, {c6[2], c6[4], c6[5]}};
  float X[2][3];
#pragma unroll
  for (int r = 0; r < 2; r++)
#pragma unroll
    for (int c = 0; c < 3; c++) X[r][c] = M[r][0] * V[c][0] + M[r][1] * V[c][1] + M[r][2] * V[c][2];
  float3 cov;
  cov.x = (X[0][0] * M[0][0] + X[0][1] * M[0][1] + X[0][2] * M[0][2]) + 0.3f;
  cov.y = X[1][0] * M[0][0] + X[1][1] * M[0][1] + X[1][2] * M[0][2];
  cov.z = (X[1][0] * M[1][0] + X[1][1] * M[1][1] + X[1][2] * M[1][2]) + 0.3f;
  return cov;
}

__device__ __forceinline__ float3 sh_to_rgb(int deg, float3 pos, float3 campos, const float *sh, uint8_t *clamped) {
  float3 d = make_float3(pos.x - campos.x, pos.y - campos.y, pos.z - campos.z);
  const float len = sqrtf(d.x * d.x + d.y * d.y + d.z * d.z);
  const float x = d.x / len, y = d.y / len, z = d.z / len;
  float out[3];
#pragma unroll
  for (int ch = 0; ch < 3; ch++) {
#define SHC(k) sh[(k) * 3 + ch]
    float res = kSH_C0 * SHC(0);
    if (deg > 0) {
      res = res - kSH_C1 * y * SHC(1) + kSH_C1 * z * SHC(2) - kSH_C1 * x * SHC(3);
      if (deg > 1) {
        const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
        res = res + kSH_C2[0] * xy * SHC(4) + kSH_C2[1] * yz * SHC(5) + kSH_C2[2] * (2.0f * zz - xx - yy) * SHC(6) +
              kSH_C2[3] * xz * SHC(7) + kSH_C2[4] * (xx - yy) * SHC(8);
        if (deg > 2) {
          res = res + kSH_C3[0] * y * (3.0f * xx - yy) * SHC(9) + kSH_C3[1] * xy * z * SHC(10) +
                kSH_C3[2] * y * (4.0f * zz - xx - yy) * SHC(11) + kSH_C3[3] * z * (2.0f * zz - 3.0f * xx - 3.0f * yy) * SHC(12) +
                kSH_C3[4] * x * (4.0f * zz - xx - yy) * SHC(13) + kSH_C3[5] * z * (xx - yy) * SHC(14) +
                kSH_C3[6] * x * (xx - 3.0f * yy) * SHC(15);
        }
      }
    }
#undef SHC
    res += 0.5f;
    clamped[ch] = (res < 0.0f);
    out[ch] = res < 0.0f ? 0.0f : res;
  }
  return make_float3(out[0], out[1], out[2]);
}

GSAJ_TRACE_DEFINE(pre)

// SH16: the SH colours come from 16 stored coefficients per channel (degree-3 maps): all 48 floats are requested up front, in
// registers (108 VGPRs: four waves per SIMD).  Any other storage (SH-0 maps, precomputed colours) reads its few coefficients
// where it needs them and runs at twice the occupancy -- this kernel is latency-bound.
// MULTI: the workgroup loops over p.bpw blocks (the loop costs ~20 VGPRs: only launches that need it get this instantiation).
template <bool SH16, bool MULTI>
__global__ __launch_bounds__(PRE_BLOCK) void k_preprocess(FwdParams p, int *__restrict__ radii, int *__restrict__ n_touched,
                                                         GeomWS g, ImageWS im, ViewStrides vs) {
  __shared__ uint32_t scan[PRE_BLOCK / 64];
  extern __shared__ uint32_t hist[];  // [tiles] workgroup-local tile histogram (when tiles <= LDS_TILES_MAX)
  {  // batched launch: blockIdx.y = view (its own camera, workspaces and per-view outputs; the Gaussians are shared)
    const size_t view = blockIdx.y;
    g = geom_view(g, view * vs.geom);
    im = image_view(im, view * vs.image);
    p.viewmatrix += 16 * view;
    p.projmatrix += 16 * view;
    if (p.campos) p.campos += 3 * view;
    radii += view * (size_t)p.P;
    n_touched += view * (size_t)p.P;
  }
  GSAJ_TRACE_BEGIN(pre)
#ifdef GSAJ_BLOCK_TRACE
  unsigned long long trp_[4] = {0, 0, 0, 0}, trp_t = wall_clock64();
#define TRP(i) { const unsigned long long n_ = wall_clock64(); trp_[i] += n_ - trp_t; trp_t = n_; }
#else
#define TRP(i)
#endif
  const int tid = threadIdx.x;
  const int tiles = p.grid_x * p.grid_y;
  const bool use_lds = tiles <= LDS_TILES_MAX;
  if (use_lds) {
    for (int t = tid; t < tiles; t += PRE_BLOCK) hist[t] = 0u;
    __syncthreads();
  }
  // A workgroup takes p.bpw consecutive blocks of 256 Gaussians, one after the other, and flushes its tile histogram ONCE: a
  // 256-Gaussian block of a 10^6-Gaussian 1280x720 frame leaves ~2800 instances in ~2000 of the 3600 tiles -- LDS aggregation
  // saves nothing there and the flush was 62 M global atomics per window (launch_preprocess picks bpw > 1 only for such launches).
  // Everything per-Gaussian stays per 256-block (block scan, block_sums: the emission slots are counted inside a block).
  const int nblk_all = (p.P + PRE_BLOCK - 1) / PRE_BLOCK;
#pragma clang loop unroll(disable)
  for (int it = 0; it < (MULTI ? p.bpw : 1); it++) {
  const int blk = MULTI ? (int)blockIdx.x * p.bpw + it : (int)blockIdx.x;
  if (blk >= nblk_all) break;
  const int idx = blk * PRE_BLOCK + tid;
  uint32_t touched = 0;
  uint32_t rect_x = 0, rect_y = 0;  // x0 | x1 << 16, y0 | y1 << 16 (GeomWS.scat)
  float depth = 0.f;
  if (idx < p.P) {
    int my_radius_i = 0;
    uint32_t rect_pack = 0;
    float2 xy = make_float2(0.f, 0.f);
    float4 con_o = make_float4(0.f, 0.f, 0.f, 0.f);
    float3 rgb = make_float3(0.f, 0.f, 0.f);
    uint8_t cl[3] = {0, 0, 0};
    float c6s[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    // every input of this Gaussian is requested up front (one memory round trip instead of three dependent ones:
    // the kernel is latency-bound, < 1 workgroup per CU); culled Gaussians simply do not use theirs
    const float3 p_orig = ld3(p.means3D, idx);
    float3 sc_in = make_float3(0.f, 0.f, 0.f);
    float4 q_in = make_float4(1.f, 0.f, 0.f, 0.f);
    if (!p.cov3D_precomp) {
      sc_in = ld3(p.scales, idx);
      q_in = reinterpret_cast<const float4 *>(p.rotations)[idx];
    }
    const float opac_in = p.opacities[idx];
    constexpr bool sh_regs = SH16;  // degree-3 storage: 48 floats = 12 x 16-byte loads
    float shv[SH16 ? 48 : 1];
    if (sh_regs) {
      const float4 *s4 = reinterpret_cast<const float4 *>(p.shs + (size_t)idx * 48);
#pragma unroll
      for (int k = 0; k < 12; k++) {
        const float4 t = s4[k];
        shv[4 * k] = t.x; shv[4 * k + 1] = t.y; shv[4 * k + 2] = t.z; shv[4 * k + 3] = t.w;
      }
    }
    const float3 p_view = xform4x3(p.viewmatrix, p_orig);
    if (p_view.z <= 0.2f) {
      if (p.prefiltered) atomicOr(&im.counters[1], ERR_PREFILTERED);  // the reference traps here (auxiliary.h:156-160)
    } else {
      const float4 p_hom = xform4x4(p.projmatrix, p_orig);
      const float p_w = 1.0f / (p_hom.w + 0.0000001f);
      const float3 p_proj = make_float3(p_hom.x * p_w, p_hom.y * p_w, p_hom.z * p_w);
      float c6[6];  // (in registers: a pointer chosen between the caller's array and a local one put the local one in scratch memory)
      if (p.cov3D_precomp) {
#pragma unroll
        for (int k = 0; k < 6; k++) c6[k] = p.cov3D_precomp[6 * (size_t)idx + k];
      } else {
        cov3d_from_scale_rot(sc_in, p.scale_modifier, q_in, c6s);
#pragma unroll
        for (int k = 0; k < 6; k++) c6[k] = c6s[k];
      }
      const float3 cov = cov2d_forward(p_orig, p.focal_x, p.focal_y, p.tanfovx, p.tanfovy, c6, p.viewmatrix);
      const float det = cov.x * cov.z - cov.y * cov.y;
      if (det != 0.0f) {
        const float det_inv = 1.f / det;
        const float3 conic = make_float3(cov.z * det_inv, -cov.y * det_inv, cov.x * det_inv);
        const float mid = 0.5f * (cov.x + cov.z);
        const float sq = sqrtf(fmaxf(0.1f, mid * mid - det));
        const float lambda1 = mid + sq, lambda2 = mid - sq;
        const float my_radius = ceilf(3.f * sqrtf(fmaxf(lambda1, lambda2)));
        const float2 pim = make_float2(ndc2pix(p_proj.x, p.W), ndc2pix(p_proj.y, p.H));
        int x0, y0, x1, y1;
        tile_rect(pim.x, pim.y, (int)my_radius, p.grid_x, p.grid_y, gsaj_tile_band(im.sticky), x0, y0, x1, y1);
        const int area = (x1 - x0) * (y1 - y0);
        if (area != 0) {
          if (!p.colors_precomp) {
            const float3 cam = make_float3(p.campos[0], p.campos[1], p.campos[2]);
            if constexpr (sh_regs) rgb = sh_to_rgb(p.D, p_orig, cam, shv, cl);
            else rgb = sh_to_rgb(p.D, p_orig, cam, p.shs + (size_t)idx * p.M * 3, cl);
          }
          rect_pack = (uint32_t)x0 | ((uint32_t)y0 << 10) | ((uint32_t)(x1 - x0) << 20);
          rect_x = (uint32_t)x0 | ((uint32_t)x1 << 16);
          rect_y = (uint32_t)y0 | ((uint32_t)y1 << 16);
          depth = p_view.z;
          my_radius_i = (int)my_radius;
          xy = pim;
          con_o = make_float4(conic.x, conic.y, conic.z, opac_in);
          touched = (uint32_t)area;
          // per-tile instance histogram: LDS atomics here, one coalesced global flush per workgroup
          // (a scattered global atomic wave-instruction costs ~17x a contiguous one)
          for (int y = y0; y < y1; y++)
            for (int x = x0; x < x1; x++) {
              if (use_lds) atomicAdd(&hist[y * p.grid_x + x], 1u);
              else atomicAdd(&im.tile_count[y * p.grid_x + x], 1u);
            }
        }
      }
    }
    radii[idx] = my_radius_i;
    n_touched[idx] = 0;
    g.depths[idx] = depth;
    if (!p.cov3D_precomp && blockIdx.y == 0) {  // (a batched launch: the views share the Gaussians, view 0's copy serves all)
#pragma unroll
      for (int k = 0; k < 6; k++) g.cov3D[6 * (size_t)idx + k] = c6s[k];
    }
    if (!p.colors_precomp) {
      g.clamped[3 * (size_t)idx] = cl[0]; g.clamped[3 * (size_t)idx + 1] = cl[1]; g.clamped[3 * (size_t)idx + 2] = cl[2];
    }
    g.tiles_touched[idx] = touched;
    if (p.colors_precomp) rgb = ld3(p.colors_precomp, idx);
    // the row both compositors gather through the sorted id list (its .w of the first quarter, the first emission slot, follows the scan)
    g.splat[3 * (size_t)idx + 0] = make_float4(xy.x, xy.y, __uint_as_float(rect_pack), 0.f);
    g.splat[3 * (size_t)idx + 1] = con_o;
    g.splat[3 * (size_t)idx + 2] = make_float4(rgb.x, rgb.y, rgb.z, depth);
  }
  TRP(0)
  // block-local inclusive scan of tiles_touched: wave scans (shuffles) + one LDS hop for the four wave totals
  uint32_t incl = touched;
  {
    const int lane = tid & 63;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t v = (uint32_t)__shfl_up((int)incl, o);
      if (lane >= o) incl += v;
    }
    if (lane == 63) scan[tid >> 6] = incl;
  }
  __syncthreads();
  uint32_t block_total = 0;
#pragma unroll
  for (int w = 0; w < PRE_BLOCK / 64; w++) {
    const uint32_t v = scan[w];
    if (w < (tid >> 6)) incl += v;
    block_total += v;
  }
  if (idx < p.P) {
    // emission slots are counted inside the block; + block_sums[block] (its exclusive offset once the frame scan below has run)
    const uint32_t first = incl - touched;
    g.point_offsets[idx] = incl;
    reinterpret_cast<float *>(g.splat)[12 * (size_t)idx + 3] = __uint_as_float(first);
    g.scat[idx] = make_uint2(rect_x, rect_y);
  }
  if (tid == 0) g.block_sums[blk] = block_total;  // block totals -> exclusive offsets: k_frame_scan
  __syncthreads();  // (scan[] is written again by the next block; after the last one: every thread's LDS histogram atomics are done)
  }
  if (use_lds)
    for (int t = tid; t < tiles; t += PRE_BLOCK) {
      const uint32_t c = hist[t];
      if (c) atomicAdd(&im.tile_count[t], c);
    }
  TRP(1)
  GSAJ_TRACE_END(pre)
#ifdef GSAJ_BLOCK_TRACE
  if ((threadIdx.x & 63) == 0) {
    unsigned long long *t = g_trace_pre + 4 * (blockIdx.x * (PRE_BLOCK / 64) + (threadIdx.x >> 6));
    t[2] = (trp_[0] << 32) | trp_[1];
    t[3] = 0;
  }
#endif
}

// Exclusive scan of n items by ONE workgroup of PRE_BLOCK lanes: each lane sums a contiguous run, the
// run totals are scanned with wave shuffles + one LDS hop, then each lane rewrites its run.  Returns the total.
__device__ uint32_t tail_exclusive_scan(const uint32_t *in, uint32_t *out, int n, uint32_t *run_max) {
  __shared__ uint32_t wsum[PRE_BLOCK / 64];
  __shared__ uint32_t wmax[PRE_BLOCK / 64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int per = (n + PRE_BLOCK - 1) / PRE_BLOCK;
  const int b0 = min(n, tid * per), b1 = min(n, b0 + per);
  uint32_t s = 0, mx = 0;
  for (int i = b0; i < b1; i++) {
    const uint32_t v = in[i];
    s += v;
    mx = max(mx, v);
  }
  uint32_t incl = s;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const uint32_t v = __shfl_up((int)incl, o);
    if (lane >= o) incl += v;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = max(mx, (uint32_t)__shfl_xor((int)mx, o));
  if (lane == 63) wsum[wave] = incl;
  if (lane == 0) wmax[wave] = mx;
  __syncthreads();
  uint32_t woff = 0, total = 0, m = 0;
#pragma unroll
  for (int w = 0; w < PRE_BLOCK / 64; w++) {
    const uint32_t v = wsum[w];
    if (w < wave) woff += v;
    total += v;
    m = max(m, wmax[w]);
  }
  if (run_max) *run_max = m;
  uint32_t run = woff + incl - s;
  for (int i = b0; i < b1; i++) {
    const uint32_t v = in[i];
    out[i] = run;
    run += v;
  }
  __syncthreads();
  return total;
}

// The per-tile histogram the other workgroups flushed -> (1) its exclusive scan (tile_offset), (2) the longest list, (3) the order
// in which both compositors take their tiles: longest list first (64 length classes, counting sort in LDS).  A launch that
// oversubscribes the chip (a batched window: 9600 tiles for 1280 workgroup slots) then ends on its short tiles instead of waiting
// for a long one dispatched last; a single frame, resident as a whole, does not care.  The order inside a length class is whatever
// the LDS atomics give -- it changes which workgroup runs where, never a result.  `len`: the histogram copied ONCE into LDS
// ([tiles], the dead workgroup-local histogram area) with wide coherent loads; every later pass reads LDS.
__device__ uint32_t tile_scan_and_schedule(int tiles, ImageWS im, uint32_t *len, uint32_t *longest_out) {
  __shared__ uint32_t wsum[PRE_BLOCK / 64], wmax[PRE_BLOCK / 64], cls[64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int c = tid; 4 * c < tiles; c += PRE_BLOCK) {  // (reads up to 3 words past `tiles`: still inside the zeroed counter block)
    const uint4 v = *reinterpret_cast<const uint4 *>(im.tile_count + 4 * c);
    len[4 * c] = v.x;
    if (4 * c + 1 < tiles) len[4 * c + 1] = v.y;
    if (4 * c + 2 < tiles) len[4 * c + 2] = v.z;
    if (4 * c + 3 < tiles) len[4 * c + 3] = v.w;
  }
  if (tid < 64) cls[tid] = 0u;
  __syncthreads();
  // (1) + (2): each lane owns a contiguous run, run totals scanned with wave shuffles + one LDS hop
  const int per = (tiles + PRE_BLOCK - 1) / PRE_BLOCK;
  const int b0 = min(tiles, tid * per), b1 = min(tiles, b0 + per);
  uint32_t s = 0, mx = 0;
  for (int i = b0; i < b1; i++) {
    s += len[i];
    mx = max(mx, len[i]);
  }
  uint32_t incl = s;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const uint32_t v = __shfl_up((int)incl, o);
    if (lane >= o) incl += v;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = max(mx, (uint32_t)__shfl_xor((int)mx, o));
  if (lane == 63) wsum[wave] = incl;
  if (lane == 0) wmax[wave] = mx;
  __syncthreads();
  uint32_t woff = 0, total = 0, longest = 0;
#pragma unroll
  for (int w = 0; w < PRE_BLOCK / 64; w++) {
    const uint32_t v = wsum[w];
    if (w < wave) woff += v;
    total += v;
    longest = max(longest, wmax[w]);
  }
  uint32_t run = woff + incl - s;
  for (int i = b0; i < b1; i++) {
    im.tile_offset[i] = run;
    run += len[i];
  }
  // (3)
  const int shift = longest >= 64u ? (32 - __builtin_clz(longest)) - 6 : 0;  // longest >> shift <= 63
  for (int t = tid; t < tiles; t += PRE_BLOCK) atomicAdd(&cls[63u - min(63u, len[t] >> shift)], 1u);
  __syncthreads();
  if (tid < 64) {  // exclusive scan of the 64 class sizes by one wave
    const uint32_t v = cls[tid];
    uint32_t in2 = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t u = (uint32_t)__shfl_up((int)in2, o);
      if (tid >= o) in2 += u;
    }
    cls[tid] = in2 - v;
  }
  __syncthreads();
  for (int t = tid; t < tiles; t += PRE_BLOCK) im.tile_order[atomicAdd(&cls[63u - min(63u, len[t] >> shift)], 1u)] = (uint32_t)t;
  __syncthreads();
  *longest_out = longest;
  return total;
}

// One workgroup per view, launched behind k_preprocess (replaces cub::DeviceScan of rasterizer_impl.cu:327): (1) exclusive offsets
// of the per-workgroup Gaussian totals and the grand total R; (2) exclusive offsets of the per-tile histogram and the tile
// schedule; (3) the longest tile list; (4) for the async forward, the device-side capacity check.
// (Rounds 1-2 had the LAST workgroup of k_preprocess to arrive do this, found with a ticket counter: a window of 10^6-Gaussian
// frames draws 31 256 tickets, all resident workgroups from ONE word at ~88 returning atomics per us -- 0.36 ms.  A launch
// boundary costs 2 us.)
__global__ __launch_bounds__(PRE_BLOCK) void k_frame_scan(int nblk, int tiles, int capacity, GeomWS g, ImageWS im, ViewStrides vs) {
  extern __shared__ uint32_t lds_len[];  // [tiles] when tiles <= LDS_TILES_MAX
  g = geom_view(g, (size_t)blockIdx.y * vs.geom);
  im = image_view(im, (size_t)blockIdx.y * vs.image);
  uint32_t *block_sums = g.block_sums;
  const uint32_t R = tail_exclusive_scan(block_sums, block_sums, nblk, nullptr);
  uint32_t m = 0, R2;
  if (tiles <= LDS_TILES_MAX) {
    R2 = tile_scan_and_schedule(tiles, im, lds_len, &m);
  } else {  // more tiles than the LDS copy holds (> 8192: beyond 2048 x 1024 pixels): scan from memory, tiles in index order
    R2 = tail_exclusive_scan(im.tile_count, im.tile_offset, tiles, &m);
    for (int t = threadIdx.x; t < tiles; t += PRE_BLOCK) im.tile_order[t] = (uint32_t)t;
  }
  if (threadIdx.x == 0) {
    im.tile_offset[tiles] = R2;
    im.counters[0] = R;
    im.counters[2] = m;
    uint32_t err = 0u;
    if (R2 != R) err |= ERR_INTERNAL;  // cannot happen; guards the invariant sum(tile lists) == sum(tiles_touched)
    if (capacity > 0) {  // async forward: nobody on the host will look at R before the next kernels run
      if (R > (uint32_t)capacity) err |= ERR_CAPACITY;  // (a tile list of any length is sorted on the device: k_tile_sort)
      if (err) {
        im.counters[4] = 1u;  // the remaining kernels of this frame return immediately
        atomicAdd(&im.sticky[0], 1u);
      }
    }
    if (err) atomicOr(&im.counters[1], err);
  }
}

// ---- instance scatter ----------------------------------------------------------------------------------------------
// Every Gaussian drops its id into a free slot of each touched tile's list segment (duplicateWithKeys,
// rasterizer_impl.cu:70-111; the segments are the tiles, and the depth half of the reference's key is gathered by the tile
// sort from the 4-byte depth array, which stays in L2: half the scattered bytes).  Slot order inside a tile is arbitrary;
// the tile sort orders by (depth, id), a total order, so the final lists are deterministic.
//
// Who writes what (MI355X: 8 XCDs, each with its own L2 that writes partial lines back on its own, in 64-byte pieces).  A
// tile's segment is a few hundred contiguous ids; written by 256-Gaussian workgroups of all eight XCDs, two ids at a time,
// it left the L2s as 4x its bytes (WRITE_SIZE, profiles/r02_pmc_summary.json).  So the tile ROWS are dealt into
// SCAT_CLS = 8 classes (row y belongs to class y % 8) and a workgroup emits only its class: blockIdx.x % 8 = class = the XCD
// the dispatcher is observed to put the workgroup on (speed only -- any placement gives the same lists), so one L2 assembles
// a segment's lines.  The workgroup covers gpt x 256 Gaussians (eight workgroups, one per class, read the same 8-byte `scat`
// entries; gpt grows with the launch -- launch_tile_binning -- so that a workgroup leaves tens of ids per tile: the run a
// workgroup appends to a segment is what leaves L2 in one piece): per tile of its class it counts in LDS, reserves ONE slot
// range with a returning global atomic, then hands the slots out with LDS atomics -- into an LDS staging area ordered by
// tile, which the workgroup then writes out with consecutive lanes on consecutive slots of a run.  (Stored straight from the
// hand-out loop every lane of a store instruction hit a different tile: 64 separate transactions per instruction, and the
// store phase was half of a wave's life -- tools/batch_trace.py.)
#ifndef SCAT_CLS
#define SCAT_CLS 8
#endif
#define SCAT_LDS_TILES 4096  // class-local tiles the LDS counters hold (3 words each: 48 KB); more: direct global atomics
#define SCAT_STAGE 4096      // instances a workgroup stages in LDS before writing them out (16 KB of ids + 8 KB of 16-bit tile indices: six
                             // workgroups per CU; with the 32-bit slot staged beside the id: 32 KB, four); more: stored directly
GSAJ_TRACE_DEFINE(scat)

__global__ __launch_bounds__(PRE_BLOCK) void k_scatter_instances(int P, int gx, int gy, int gpt, GeomWS g, ImageWS im,
                                                                 uint32_t *__restrict__ inst_id, ViewStrides vs) {
  extern __shared__ uint32_t lds[];  // [3 * ltiles]: count -> reserved base, fill cursor, staging base (class-local tile index)
  __shared__ uint32_t stage_id[SCAT_STAGE];
  __shared__ uint16_t stage_lt[SCAT_STAGE];  // class-local tile of a staged id (its slot = the tile's reserved base + position in the tile's run)
  __shared__ uint32_t s_wave[PRE_BLOCK / 64], s_carry;
  {
    const size_t view = blockIdx.y;
    g = geom_view(g, view * vs.geom);
    im = image_view(im, view * vs.image);
    inst_id = gsaj_shift(inst_id, view * vs.bin);
  }
  if (im.counters[4]) return;  // aborted async frame
  GSAJ_TRACE_BEGIN(scat)
#ifdef GSAJ_BLOCK_TRACE
  unsigned long long trs_[4] = {0, 0, 0, 0}, trs_t = wall_clock64();
#define TRS(i) { const unsigned long long n_ = wall_clock64(); trs_[i] += n_ - trs_t; trs_t = n_; }
#else
#define TRS(i)
#endif
  const int tid = threadIdx.x;
  const int cls = (int)(blockIdx.x % SCAT_CLS), gb = (int)(blockIdx.x / SCAT_CLS);
  const int rows_c = (gy - cls + SCAT_CLS - 1) / SCAT_CLS;  // tile rows y = cls + SCAT_CLS * j < gy
  const int ltiles = rows_c > 0 ? rows_c * gx : 0;
  const bool use_lds = ((gy + SCAT_CLS - 1) / SCAT_CLS) * gx <= SCAT_LDS_TILES;
  uint32_t *cnt = lds, *fill = lds + ltiles, *lbase = lds + 2 * ltiles;
  if (use_lds) {
    for (int t = tid; t < 2 * ltiles; t += PRE_BLOCK) lds[t] = 0u;
    if (tid == 0) s_carry = 0u;
    __syncthreads();
  }
  // this workgroup's Gaussians: (gb * gpt + i) * 256 + tid, i < gpt -- four rectangles requested at a time; the second walk
  // finds them in L2.  Rows of this class inside [y0, y1): y = ys, ys + SCAT_CLS, ...
#define SCAT_FOR_EACH_TILE(BODY)                                                                     \
  for (int i0 = 0; i0 < gpt; i0 += 4) {                                                              \
    uint2 sc[4];                                                                                     \
    _Pragma("unroll") for (int j = 0; j < 4; j++) {                                                  \
      const int idx = (gb * gpt + i0 + j) * PRE_BLOCK + tid;                                         \
      sc[j] = (i0 + j < gpt && idx < P) ? g.scat[idx] : make_uint2(0u, 0u);                          \
    }                                                                                                \
    _Pragma("unroll") for (int j = 0; j < 4; j++) {                                                  \
      const int x0 = (int)(sc[j].x & 0xffffu), x1 = (int)(sc[j].x >> 16);                            \
      const int y0 = (int)(sc[j].y & 0xffffu), y1 = (int)(sc[j].y >> 16);                            \
      if (x1 > x0) {                                                                                 \
        const uint32_t id = (uint32_t)((gb * gpt + i0 + j) * PRE_BLOCK + tid);                       \
        (void)id;                                                                                    \
        const int ys = y0 + ((cls - y0) % SCAT_CLS + SCAT_CLS) % SCAT_CLS;                           \
        for (int y = ys; y < y1; y += SCAT_CLS)                                                      \
          for (int x = x0; x < x1; x++) {                                                            \
            const int lt = (y / SCAT_CLS) * gx + x; /* class-local tile */                           \
            const int t = y * gx + x;               /* tile */                                       \
            (void)lt; (void)t;                                                                       \
            BODY                                                                                     \
          }                                                                                          \
      }                                                                                              \
    }                                                                                                \
  }
  if (!use_lds) {  // (more than 8 x 8192 tiles: beyond 16 384 x 2048 pixels)
    SCAT_FOR_EACH_TILE({ inst_id[im.tile_offset[t] + atomicAdd(&im.tile_cursor[t], 1u)] = id; })
    return;
  }
  TRS(0)
  SCAT_FOR_EACH_TILE({ atomicAdd(&cnt[lt], 1u); })
  __syncthreads();
  TRS(1)
  // reserve this workgroup's slot range in every tile of its class it touches (returning atomics, their round trips
  // overlapping across the lanes) and lay the tiles' runs out back to back in the staging area: 256 tiles per step, exclusive
  // scan of their counts by wave shuffles + one LDS hop, carry from step to step
  for (int l0 = 0; l0 < ltiles; l0 += PRE_BLOCK) {
    const int lt = l0 + tid, lane = tid & 63;
    uint32_t c = 0u, off = 0u, base = 0u;
    if (lt < ltiles) {
      const int j = lt / gx;
      const int t = (j * SCAT_CLS + cls) * gx + (lt - j * gx);
      c = cnt[lt];
      off = im.tile_offset[t];  // in flight together with the atomic
      if (c) base = atomicAdd(&im.tile_cursor[t], c);
    }
    uint32_t incl = c;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t v = (uint32_t)__shfl_up((int)incl, o);
      if (lane >= o) incl += v;
    }
    if (lane == 63) s_wave[tid >> 6] = incl;
    __syncthreads();
    uint32_t before = s_carry;
#pragma unroll
    for (int w = 0; w < PRE_BLOCK / 64; w++)
      if (w < (tid >> 6)) before += s_wave[w];
    if (lt < ltiles) {
      cnt[lt] = off + base;
      lbase[lt] = before + incl - c;
    }
    __syncthreads();
    if (tid == PRE_BLOCK - 1) s_carry = before + incl;
  }
  __syncthreads();
  const uint32_t total = s_carry;  // this workgroup's instances
  TRS(2)
  if (total <= SCAT_STAGE) {
    SCAT_FOR_EACH_TILE({
      const uint32_t k = atomicAdd(&fill[lt], 1u);
      stage_id[lbase[lt] + k] = id;
      stage_lt[lbase[lt] + k] = (uint16_t)lt;
    })
    __syncthreads();
    for (uint32_t i = tid; i < total; i += PRE_BLOCK) {
      const uint32_t lt = stage_lt[i];
      inst_id[cnt[lt] + (i - lbase[lt])] = stage_id[i];
    }
  } else {  // (large Gaussians: more instances than the staging area holds)
    SCAT_FOR_EACH_TILE({ inst_id[cnt[lt] + atomicAdd(&fill[lt], 1u)] = id; })
  }
#undef SCAT_FOR_EACH_TILE
  TRS(3)
  GSAJ_TRACE_END(scat)
#ifdef GSAJ_BLOCK_TRACE
  if ((threadIdx.x & 63) == 0) {
    const unsigned tw_ = (blockIdx.y * gridDim.x + blockIdx.x) * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (tw_ < GSAJ_TRACE_MAX) {
      g_trace_scat[4 * tw_ + 2] = (trs_[0] << 32) | trs_[1];
      g_trace_scat[4 * tw_ + 3] = (trs_[2] << 32) | trs_[3];
    }
  }
#endif
}

// fp16-storage rows (GSAJ_FWD_RECORDS_FP16): conic / opacity / colour of every Gaussian rounded to half ONCE, here; mean2D,
// depth and the emission slot stay 32-bit.  One 32-byte row instead of 48 for the compositors to gather.
__global__ __launch_bounds__(256) void k_pack_splat16(int P, GeomWS g, const uint32_t *__restrict__ counters, ViewStrides vs) {
  g = geom_view(g, (size_t)blockIdx.y * vs.geom);
  if (gsaj_shift(counters, (size_t)blockIdx.y * vs.image)[4]) return;
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= P) return;
  const float4 a = g.splat[3 * (size_t)idx + 0], bq = g.splat[3 * (size_t)idx + 1], c = g.splat[3 * (size_t)idx + 2];
  g.splat16[2 * (size_t)idx + 0] = make_float4(a.x, a.y, c.w, a.w);
  g.splat16[2 * (size_t)idx + 1] = make_float4(__uint_as_float(gsaj_pack_h2(bq.x, bq.y)), __uint_as_float(gsaj_pack_h2(bq.z, bq.w)),
                                              __uint_as_float(gsaj_pack_h2(c.x, c.y)), __uint_as_float(gsaj_pack_h2(c.z, 0.f)));
}

// ---- register exchange for the wave-local stages of the tile sort -------------------------------
// value of lane (l ^ J) for J = 32, 16 (v_permlane*_swap of a register with itself) and 8, 4, 2, 1 (DPP)
template <int J>
__device__ __forceinline__ uint32_t lane_xor(uint32_t x, int lane) {
  if (J == 32) {
    const gsaj_u32x2 r = __builtin_amdgcn_permlane32_swap(x, x, false, false);  // r[0] = [lo | lo], r[1] = [hi | hi]
    return lane < 32 ? r[1] : r[0];
  } else if (J == 16) {
    const gsaj_u32x2 r = __builtin_amdgcn_permlane16_swap(x, x, false, false);  // r[0] = rows [0,0,2,2], r[1] = rows [1,1,3,3]
    return (lane & 16) ? r[0] : r[1];
  } else if (J == 8) {
    return (uint32_t)__builtin_amdgcn_mov_dpp((int)x, 0x128, 0xF, 0xF, true);  // row_ror:8 (every lane reads a lane: no `old` value to initialise)
  } else if (J == 4) {
    int t = __builtin_amdgcn_mov_dpp((int)x, 0x104, 0xF, 0x5, false);                  // row_shl:4 -> banks 0, 2 (lane <- lane + 4); banks 1, 3: next line
    t = __builtin_amdgcn_update_dpp(t, (int)x, 0x114, 0xF, 0xA, false);              // row_shr:4 -> banks 1, 3 (lane <- lane - 4)
    return (uint32_t)t;
  } else if (J == 2) {
    return (uint32_t)__builtin_amdgcn_mov_dpp((int)x, 0x4E, 0xF, 0xF, true);  // quad_perm [2,3,0,1]
  } else {
    return (uint32_t)__builtin_amdgcn_mov_dpp((int)x, 0xB1, 0xF, 0xF, true);  // quad_perm [1,0,3,2]
  }
}
// value of lane (l ^ MASK) for MASK = 2^j - 1: the mirror inside groups of 2, 4, 8, 16, 32, 64 lanes (DPP quad_perm /
// row_half_mirror / row_mirror, + the half-row / half-wave swaps above)
template <int MASK>
__device__ __forceinline__ uint32_t lane_mirror(uint32_t x, int lane) {
  if (MASK == 1) return (uint32_t)__builtin_amdgcn_mov_dpp((int)x, 0xB1, 0xF, 0xF, true);         // quad_perm [1,0,3,2]
  else if (MASK == 3) return (uint32_t)__builtin_amdgcn_mov_dpp((int)x, 0x1B, 0xF, 0xF, true);    // quad_perm [3,2,1,0]
  else if (MASK == 7) return (uint32_t)__builtin_amdgcn_mov_dpp((int)x, 0x141, 0xF, 0xF, true);   // row_half_mirror
  else if (MASK == 15) return (uint32_t)__builtin_amdgcn_mov_dpp((int)x, 0x140, 0xF, 0xF, true);  // row_mirror
  else if (MASK == 31) return lane_xor<16>(lane_mirror<15>(x, lane), lane);
  else return lane_xor<32>(lane_mirror<31>(x, lane), lane);
}
template <int MASK>
__device__ __forceinline__ uint64_t key_mirror(uint64_t k, int lane) {
  return ((uint64_t)lane_mirror<MASK>((uint32_t)(k >> 32), lane) << 32) | lane_mirror<MASK>((uint32_t)k, lane);
}

// The network is the ALL-ASCENDING form of the bitonic sort: the merge of two sorted runs of length K/2 opens with a "flip"
// (element i against element K - 1 - i of the block) and continues with the half-cleaners of stride K/4 ... 1, every comparator
// putting the smaller key at the lower index.  Keys past the end of a list are KEY_INF and an all-ascending comparator never
// moves KEY_INF down: whole chunks, merges and comparators that would touch only padding are skipped, so a list of 315 keys costs
// what 384 keys cost, not 512 (the alternating-direction form of rounds 1-2 had to sort the padding too).
//
// A comparator is v_min_f64 + v_max_f64.  A key (depth bits << 32 | id) read as a double has sign 0 and an exponent field
// below 0x7F8 (the depth is a finite positive float), i.e. it is a positive normal double, and positive doubles order as their
// bit patterns do: min / max of the doubles ARE the smaller / larger key, bit for bit, in one full-rate instruction each -- where
// the integer form is a 64-bit compare and two selects per key kept (measured: all of these issue at ~4.3 cycles).  The padding key
// is the largest finite double (~0 would be a NaN, which min / max drop).  Written as instructions: through fmin() / fmax() the
// compiler first quiets a possible signalling NaN with an extra v_max_f64 x, x.
#define KEY_INF 0x7FEFFFFFFFFFFFFFull
__device__ __forceinline__ void key_minmax(uint64_t a, uint64_t b, uint64_t &lo, uint64_t &hi) {
  double l, h;
  asm("v_min_f64 %0, %2, %3\n\tv_max_f64 %1, %2, %3" : "=&v"(l), "=&v"(h) : "v"(__longlong_as_double((long long)a)), "v"(__longlong_as_double((long long)b)));
  lo = (uint64_t)__double_as_longlong(l);
  hi = (uint64_t)__double_as_longlong(h);
}
__device__ __forceinline__ uint64_t key_min(uint64_t a, uint64_t b) {
  double l;
  asm("v_min_f64 %0, %1, %2" : "=v"(l) : "v"(__longlong_as_double((long long)a)), "v"(__longlong_as_double((long long)b)));
  return (uint64_t)__double_as_longlong(l);
}
__device__ __forceinline__ uint64_t key_max(uint64_t a, uint64_t b) {
  double h;
  asm("v_max_f64 %0, %1, %2" : "=v"(h) : "v"(__longlong_as_double((long long)a)), "v"(__longlong_as_double((long long)b)));
  return (uint64_t)__double_as_longlong(h);
}
// lanes whose bit J is clear hold the lower-indexed key of their pair
template <int J>
__device__ __forceinline__ constexpr uint64_t lower_lanes() {
  return J == 1 ? 0x5555555555555555ull : J == 2 ? 0x3333333333333333ull : J == 4 ? 0x0F0F0F0F0F0F0F0Full
       : J == 8 ? 0x00FF00FF00FF00FFull : J == 16 ? 0x0000FFFF0000FFFFull : 0x00000000FFFFFFFFull;
}
// The comparators of one stage on the lane's two keys: A <- min(XA, YA) in the lanes of `lower`, max(XA, YA) in the others; B
// likewise.  The two halves run under complementary EXEC masks (scalar constants), so no per-lane select is spent: four VALU
// instructions for the two keys.  (All 64 lanes are active wherever the sort calls this; s_nop: a DPP instruction may follow, and
// it needs five wait states after a scalar write of EXEC that the compiler cannot see.)
__device__ __forceinline__ void key_stage(uint64_t &A, uint64_t XA, uint64_t YA, uint64_t &B, uint64_t XB, uint64_t YB, uint64_t lower) {
  double a, b2;
  uint64_t save;
  asm volatile("s_mov_b64 %[sv], exec\n\t"
               "s_and_b64 exec, %[sv], %[lo]\n\t"
               "v_min_f64 %[a], %[xa], %[ya]\n\t"
               "v_min_f64 %[b], %[xb], %[yb]\n\t"
               "s_andn2_b64 exec, %[sv], %[lo]\n\t"
               "v_max_f64 %[a], %[xa], %[ya]\n\t"
               "v_max_f64 %[b], %[xb], %[yb]\n\t"
               "s_mov_b64 exec, %[sv]\n\t"
               "s_nop 4"
               : [a] "=&v"(a), [b] "=&v"(b2), [sv] "=&s"(save)
               : [xa] "v"(__longlong_as_double((long long)XA)), [ya] "v"(__longlong_as_double((long long)YA)),
                 [xb] "v"(__longlong_as_double((long long)XB)), [yb] "v"(__longlong_as_double((long long)YB)), [lo] "s"(lower)
               : "scc");
  A = (uint64_t)__double_as_longlong(a);
  B = (uint64_t)__double_as_longlong(b2);
}
// the two keys of lanes l and l ^ J for J = 32, 16: v_permlane*_swap of the key with a copy of itself leaves the lower lane's key
// in one register pair and the upper lane's in the other, in BOTH lanes -- the comparator needs no "which is mine" select
template <int J>
__device__ __forceinline__ void key_both(uint64_t k, uint64_t &of_lower, uint64_t &of_upper) {
  const uint32_t hi = (uint32_t)(k >> 32), lo = (uint32_t)k;
  gsaj_u32x2 rh, rl;
  if (J == 32) {
    rh = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);  // r[0] = [lo half | lo half], r[1] = [hi half | hi half]
    rl = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
    of_lower = ((uint64_t)rh[0] << 32) | rl[0];
    of_upper = ((uint64_t)rh[1] << 32) | rl[1];
  } else {
    rh = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);  // r[0] = rows [0,0,2,2], r[1] = rows [1,1,3,3]
    rl = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    of_lower = ((uint64_t)rh[0] << 32) | rl[0];
    of_upper = ((uint64_t)rh[1] << 32) | rl[1];
  }
}
// one half-cleaner stage of stride J < 64 on the two keys a lane holds (indices base + lane and base + 64 + lane)
template <int J>
__device__ __forceinline__ void sort_stage(uint64_t &A, uint64_t &B, int lane) {
  if constexpr (J >= 16) {
    uint64_t al, au, bl, bu;
    key_both<J>(A, al, au);
    key_both<J>(B, bl, bu);
    key_stage(A, al, au, B, bl, bu, lower_lanes<J>());
  } else {
    const uint64_t PA = ((uint64_t)lane_xor<J>((uint32_t)(A >> 32), lane) << 32) | lane_xor<J>((uint32_t)A, lane);
    const uint64_t PB = ((uint64_t)lane_xor<J>((uint32_t)(B >> 32), lane) << 32) | lane_xor<J>((uint32_t)B, lane);
    key_stage(A, A, PA, B, B, PB, lower_lanes<J>());
  }
}
// half-cleaners of stride J, J/2, ..., 1 (J = 64: the in-lane comparator between a lane's two keys first)
template <int J>
__device__ __forceinline__ void merge_strides(uint64_t &A, uint64_t &B, int lane) {
  if constexpr (J == 64) {
    key_minmax(A, B, A, B);
  } else {
    sort_stage<J>(A, B, lane);
  }
  if constexpr (J > 1) merge_strides<J / 2>(A, B, lane);
}
// the flip of a merge of size K <= 128 inside the chunk, then its half-cleaners
template <int K>
__device__ __forceinline__ void local_merge(uint64_t &A, uint64_t &B, int lane) {
  if constexpr (K == 128) {  // index l (key A of lane l) against index 127 - l (key B of lane 63 - l)
    const uint64_t PA = key_mirror<63>(B, lane), PB = key_mirror<63>(A, lane);
    A = key_min(A, PA);
    B = key_max(B, PB);
    merge_strides<32>(A, B, lane);
  } else {  // inside A and inside B: lane l against lane l ^ (K - 1)
    const uint64_t PA = key_mirror<K - 1>(A, lane), PB = key_mirror<K - 1>(B, lane);
    key_stage(A, A, PA, B, B, PB, lower_lanes<K / 2>());
    if constexpr (K >= 4) merge_strides<K / 4>(A, B, lane);
  }
}
// the merges of size K, 2K, ..., 128 (all inside the chunk), stopping at the padded list length m
template <int K>
__device__ __forceinline__ void local_merges(uint64_t &A, uint64_t &B, int lane, int m) {
  if (K > m) return;
  local_merge<K>(A, B, lane);
  if constexpr (K < 128) local_merges<2 * K>(A, B, lane, m);
}

// Sort of keys[0, n) in LDS, ascending; keys[n, ceil(n / 128) * 128) must hold KEY_INF; n >= 1.  Barrier before (the caller's
// stores to `keys`), barrier after.  Comparators with stride <= 64 pair keys inside one aligned 128-key chunk, and a wave holds
// a chunk in REGISTERS (lane l: keys l and l + 64): stride 64 is in-lane, the mirrors and strides 32 / 16 use
// v_permlane32_swap / v_permlane16_swap, the rest DPP -- an LDS round trip per stage was the latency of this kernel.  So the
// chunk sorts (merges up to 128) never touch LDS, and a merge of size k >= 256 goes through LDS only for its flip and its
// strides >= 128 (workgroup barriers), followed by the seven chunk-local stages in registers again.
template <int NT>
__device__ __forceinline__ void lds_bitonic_sort(uint64_t *keys, int n, int tid) {
  const int lane = tid & 63, wave = tid >> 6;
  const int nch = (n + 127) >> 7, nup = nch << 7;  // chunks with at least one key; everything at or past nup is padding nobody reads
  int m = 2;
  while (m < n) m <<= 1;
  for (int chunk = wave; chunk < nch; chunk += NT / 64) {
    const int ia = chunk * 128 + lane, ib = ia + 64;
    uint64_t A = keys[ia], B = keys[ib];
    local_merges<2>(A, B, lane, m);
    keys[ia] = A;
    keys[ib] = B;
  }
  __syncthreads();
  for (int k = 256, hs = 7; (k >> 1) < nup; k <<= 1, hs++) {
    // blocks of k keys whose upper half holds a key: [base, base + k) with base + k / 2 < nup.  The others are sorted already.
    const int half = k >> 1;  // = 1 << hs
    // flip: base + off against base + k - 1 - off
    for (int i = tid; i < (nup >> 1) + half; i += NT) {  // (i enumerates (block, off) over every block that starts below nup)
      const int blk = i >> hs, off = i & (half - 1);
      const int l = (blk << (hs + 1)) + off, r = (blk << (hs + 1)) + k - 1 - off;
      if (r < nup) key_minmax(keys[l], keys[r], keys[l], keys[r]);
    }
    __syncthreads();
    for (int j = k >> 2; j >= 128; j >>= 1) {  // half-cleaners through LDS
      for (int i = tid; i < (nup >> 1); i += NT) {
        const int l = ((i & ~(j - 1)) << 1) | (i & (j - 1)), r = l + j;
        if (r < nup && ((l & ~(k - 1)) + half) < nup) key_minmax(keys[l], keys[r], keys[l], keys[r]);
      }
      __syncthreads();
    }
    for (int chunk = wave; chunk < nch; chunk += NT / 64) {
      if (((chunk * 128) & ~(k - 1)) + half >= nup) continue;  // (its block's upper half is padding)
      const int ia = chunk * 128 + lane, ib = ia + 64;
      uint64_t A = keys[ia], B = keys[ib];
      merge_strides<64>(A, B, lane);
      keys[ia] = A;
      keys[ib] = B;
    }
    __syncthreads();
  }
}

// keys[0, T) holds a BITONIC sequence (ascending, KEY_INF, descending: sort_long_list's merge step), T a power of two >= 128:
// its half-cleaners of stride T/2 ... 1 sort it.
template <int NT>
__device__ __forceinline__ void lds_bitonic_merge(uint64_t *keys, int T, int tid) {
  const int lane = tid & 63, wave = tid >> 6;
  for (int j = T >> 1; j >= 128; j >>= 1) {
    for (int i = tid; i < (T >> 1); i += NT) {
      const int l = ((i & ~(j - 1)) << 1) | (i & (j - 1)), r = l + j;
      key_minmax(keys[l], keys[r], keys[l], keys[r]);
    }
    __syncthreads();
  }
  for (int chunk = wave; chunk * 128 < T; chunk += NT / 64) {
    const int ia = chunk * 128 + lane, ib = ia + 64;
    uint64_t A = keys[ia], B = keys[ib];
    merge_strides<64>(A, B, lane);
    keys[ia] = A;
    keys[ib] = B;
  }
  __syncthreads();
}

// A tile list longer than the LDS holds (n > T keys; the reference's global radix sort takes any length,
// rasterizer_impl.cu:353-368).  (1) chunks of T keys (ids + gathered depths) are sorted in LDS -> b; (2) merge passes b -> a -> b ... over runs of
// T, 2T, ...: the output of a pair of runs is produced T keys at a time -- the merge-path split of every T-th output
// diagonal is found by binary search (one thread per diagonal), the two input pieces (together T keys) are loaded as
// [A ascending | KEY_INF | B descending], a bitonic sequence that ONE merge level sorts.  Everything inside the tile's own
// workgroup and key segment; returns the buffer that holds the sorted keys.
// (split_s: PRE_BLOCK + 1 words of the kernel's DYNAMIC LDS behind the T keys -- as a static array it sat beside every workgroup's 32 KB
// of keys, also those of ordinary lists, and kept the fifth workgroup off a CU.)
template <int NT>
__device__ const uint64_t *sort_long_list(uint64_t *lds, int T, uint32_t *split_s, const uint32_t *ids, const float *__restrict__ depths,
                                          uint64_t *a, uint64_t *b, int n, int tid) {
  for (int c0 = 0; c0 < n; c0 += T) {
    const int len = min(T, n - c0);
    for (int i0 = tid; i0 < T; i0 += 4 * NT) {  // (four ids, then four depths, in flight together: k_tile_sort)
      uint32_t id[4], dz[4];
#pragma unroll
      for (int u = 0; u < 4; u++) id[u] = ids[c0 + min(i0 + u * NT, len - 1)];
#pragma unroll
      for (int u = 0; u < 4; u++) dz[u] = __float_as_uint(depths[id[u]]);
#pragma unroll
      for (int u = 0; u < 4; u++) asm volatile("" : "+v"(dz[u]));
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const int i = i0 + u * NT;
        if (i < T) lds[i] = i < len ? ((uint64_t)dz[u] << 32) | id[u] : KEY_INF;
      }
    }
    __syncthreads();
    lds_bitonic_sort<NT>(lds, T, tid);
    for (int i = tid; i < len; i += NT) b[c0 + i] = lds[i];
    __syncthreads();
  }
  uint64_t *src = b, *dst = a;
  for (long long width = T; width < (long long)n; width <<= 1) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this workgroup's stores of the previous pass, before anyone reads them
    __syncthreads();
    for (long long p0 = 0; p0 < (long long)n; p0 += 2 * width) {
      const int la = (int)min(width, (long long)n - p0);
      const int lb = (int)max(0ll, min(width, (long long)n - p0 - width));
      const uint64_t *A = src + p0, *B = src + p0 + width;
      uint64_t *D = dst + p0;
      if (lb == 0) {
        for (int i = tid; i < la; i += NT) D[i] = A[i];
        continue;
      }
      const int tot = la + lb, nblk = (tot + T - 1) / T;
      for (int j0 = 0; j0 < nblk; j0 += PRE_BLOCK) {
        for (int t = tid; t <= PRE_BLOCK; t += NT) {  // merge-path split of diagonal d: how many of the first d outputs come from A
          const long long dl = (long long)(j0 + t) * T;
          const int d = (int)min(dl, (long long)tot);
          int lo = max(0, d - lb), hi = min(d, la);
          while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (A[mid] <= B[d - 1 - mid]) lo = mid + 1;
            else hi = mid;
          }
          split_s[t] = (uint32_t)lo;
        }
        __syncthreads();
        for (int j = j0; j < min(nblk, j0 + PRE_BLOCK); j++) {
          const int a0 = (int)split_s[j - j0], a1 = (int)split_s[j - j0 + 1];
          const int d0 = j * T, d1 = min(tot, d0 + T);
          const int b0 = d0 - a0, cA = a1 - a0, cB = (d1 - a1) - b0;
          for (int i = tid; i < T; i += NT) lds[i] = i < cA ? A[a0 + i] : (i >= T - cB ? B[b0 + (T - 1 - i)] : KEY_INF);
          __syncthreads();
          lds_bitonic_merge<NT>(lds, T, tid);
          for (int i = tid; i < cA + cB; i += NT) D[d0 + i] = lds[i];
          __syncthreads();
        }
      }
    }
    uint64_t *t = src;
    src = dst;
    dst = t;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  return src;
}

// One workgroup per tile: sort of the tile's (depth, id) keys -> the sorted Gaussian ids (point_list, in place) and the tile's [start, end)
// range (cub::DeviceRadixSort + identifyTileRanges, rasterizer_impl.cu:353-368, 116-138); the tile's `reached` flags cleared.
GSAJ_TRACE_DEFINE(sort)

// NT threads per tile: 256, or 512 from 4096 keys of LDS on -- 32 KB and more per workgroup limit a CU to four or five of them, and
// with four waves each the kernel (dependent LDS round trips between barriers) had 16-20 waves per CU to hide them behind (cfg5:
// 1.66 -> 1.48 ms; at 2048 keys a CU already holds its 32 waves and 512 threads only idle: cfg3 294 -> 389 us, so not there)
#ifndef GSAJ_SORT_128_MAX
#define GSAJ_SORT_128_MAX 1024  // LDS capacity (keys) up to which a tile is sorted by 128 threads: lists of a few hundred keys are two or three
                                 // 128-key chunks, and of four waves one or two only waited at the barriers (cfg2 47 -> 43 us; one wave: 46)
#endif
template <int NT>
__global__ __launch_bounds__(NT) void k_tile_sort(ImageWS im, const float *__restrict__ depths, uint64_t *__restrict__ inst_key,
                                                   uint64_t *__restrict__ keys_b, uint32_t *point_list,
                                                   uint8_t *__restrict__ reached, int cap, int pass, int rec16, ViewStrides vs) {
  {
    const size_t view = blockIdx.y;
    im = image_view(im, view * vs.image);
    depths = gsaj_shift(depths, view * vs.geom);
    inst_key = gsaj_shift(inst_key, view * vs.bin);
    keys_b = gsaj_shift(keys_b, view * vs.bin);
    point_list = gsaj_shift(point_list, view * vs.bin);
    reached = gsaj_shift(reached, view * vs.bin);
  }
  // `cap` keys of dynamic LDS: the host sizes it to the longest tile list it expects (sync path: known exactly; async path: the
  // caller's tile_list_capacity), so short lists do not pay for 32 KB per workgroup; a longer list takes sort_long_list
  extern __shared__ uint64_t keys[];
  if (im.counters[4]) return;  // aborted async frame
  GSAJ_TRACE_BEGIN(sort)
#ifdef GSAJ_BLOCK_TRACE
  unsigned long long tr_a = wall_clock64(), tr_b = 0, tr_c = 0;
#endif
  const int tid = threadIdx.x, tile = blockIdx.x;
  const uint32_t beg = im.tile_offset[tile * TILE_REP], end = im.tile_offset[(tile + 1) * TILE_REP];
  const int n = (int)(end - beg);
  // pass 0: the only launch.  Long lists (cap >= 4096 keys = 32 KB of LDS, where LDS and not wave slots bounds the resident
  // workgroups) are sorted in TWO launches instead: pass 1 has LDS for `cap` keys and takes the tiles whose padded length exceeds
  // cap / 2 (+ the bookkeeping of every tile), pass 2 has half the LDS -- twice the resident workgroups -- and takes the rest.
  // At cfg5 (lists of ~3300, a few above 4096) one launch ran at 2 workgroups per CU: 4.2 of the window's 10.5 ms.
  if (pass != 2) {
    if (tid == 0) im.ranges[tile] = n > 0 ? make_uint2(beg, end) : make_uint2(0u, 0u);
    if (tid == 0 && tile == 0) im.counters[7] = (uint32_t)rec16;  // row format of this frame (read by the compositors)
    // no pixel has reached any of the tile's instances yet (the reverse compositor sets the flags of the rows it writes).  The
    // flags are indexed by emission slot, not by sorted position: the tiles' ranges merely tile [0, R)
    for (uint32_t k = beg + tid; k < end; k += NT) reached[k] = 0;
  }
  if (n == 0) return;
  const uint64_t *sorted = keys;
  if (n > cap) {
    if (pass == 2) return;  // (pass 2 has half of pass 1's LDS: pass 1 took it)
    // chunks of `cap` keys with the merge-path splits behind them; from 4096 keys on (all of a CU's LDS for five workgroups) the
    // chunk is half the capacity and the splits sit in the other half
    const int T = cap >= 4096 ? cap / 2 : cap;
    sorted = sort_long_list<NT>(keys, T, reinterpret_cast<uint32_t *>(keys + T), point_list + beg, depths, inst_key + beg, keys_b + beg, n, tid);
  } else {
    int m = 2;
    while (m < n) m <<= 1;
    if (pass == 1 && 2 * m <= cap) return;  // pass 2's
    if (pass == 2 && m > cap) return;       // pass 1's (here cap = half of pass 1's)
    // the tile's ids as scattered (point_list, sorted in place below) + their depths gathered from the 4-byte depth array (L2)
    const int nup = ((n + 127) >> 7) << 7;  // (KEY_INF up to the end of the last 128-key chunk: lds_bitonic_sort)
    // Four ids, then their four depths, in flight together: one entry at a time the two dependent loads of every trip were
    // most of a short list's life (a position past the list reads the list's last entry and stores KEY_INF).
    for (int i0 = tid; i0 < nup; i0 += 4 * NT) {
      uint32_t id[4], dz[4];
#pragma unroll
      for (int u = 0; u < 4; u++) id[u] = point_list[beg + (uint32_t)min(i0 + u * NT, n - 1)];
#pragma unroll
      for (int u = 0; u < 4; u++) dz[u] = __float_as_uint(depths[id[u]]);
#pragma unroll
      for (int u = 0; u < 4; u++) asm volatile("" : "+v"(dz[u]));  // (all eight loads issued HERE, not sunk into the branches below)
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const int i = i0 + u * NT;
        if (i < nup) keys[i] = i < n ? ((uint64_t)dz[u] << 32) | id[u] : KEY_INF;
      }
    }
    __syncthreads();
#ifdef GSAJ_BLOCK_TRACE
    tr_b = wall_clock64();
#endif
    lds_bitonic_sort<NT>(keys, n, tid);
  }
#ifdef GSAJ_BLOCK_TRACE
  tr_c = wall_clock64();
#endif
  for (int i = tid; i < n; i += NT) point_list[beg + (uint32_t)i] = (uint32_t)sorted[i];
  GSAJ_TRACE_END(sort)
#ifdef GSAJ_BLOCK_TRACE
  if ((threadIdx.x & 63) == 0) {
    unsigned long long *t = g_trace_sort + 4 * (blockIdx.x * (NT / 64) + (threadIdx.x >> 6));
    t[2] = tr_b - tr_a;
    t[3] = tr_c - tr_b;
  }
#endif
}

__global__ __launch_bounds__(PRE_BLOCK) void k_mark_visible(int P, const float *__restrict__ means3D,
                                                            const float *__restrict__ vm, uint8_t *__restrict__ present) {
  const int idx = blockIdx.x * PRE_BLOCK + threadIdx.x;
  if (idx >= P) return;
  const float3 pv = xform4x3(vm, ld3(means3D, idx));
  present[idx] = pv.z > 0.2f;
}

// zero the frame counters + tile histogram / cursors of every view (one launch for K views).  A kernel also for K = 1, not
// hipMemsetAsync: a forward captured into a hipGraph must replay correctly, and on ROCm 7.2 a captured memset node stopped
// taking effect from the second replay on whenever the host had synchronised in between (tests/test_gpu_device_tracker.py)
__global__ __launch_bounds__(256) void k_zero_frame_state(uint32_t *__restrict__ counters, size_t words, size_t view_stride) {
  uint32_t *c = gsaj_shift(counters, blockIdx.y * view_stride);
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < words; i += (size_t)gridDim.x * 256) c[i] = 0u;
}

int launch_preprocess(const FwdParams &p0, int *radii, int *n_touched, const GeomWS &g, const ImageWS &im, ViewStrides vs,
                      hipStream_t s) {
  FwdParams p = p0;
  const int nblk = (p.P + PRE_BLOCK - 1) / PRE_BLOCK;
  const int views = p.views > 0 ? p.views : 1;
  // blocks of 256 Gaussians per workgroup: 1 unless the launch has far more workgroups than the chip holds (k_preprocess)
  p.bpw = 1;
  while (p.bpw < 8 && (long long)nblk * views / (2 * p.bpw) >= 4096) p.bpw *= 2;
  const int nwg = (nblk + p.bpw - 1) / p.bpw;
  {
    const size_t words = im.zero_bytes / sizeof(uint32_t);
    hipLaunchKernelGGL(k_zero_frame_state, dim3((unsigned)((words + 1023) / 1024), views), dim3(256), 0, s, im.counters, words,
                       vs.image);
  }
  {
    GsajProfScope ps(ST_PREPROCESS, s);
    const int tiles = p.grid_x * p.grid_y;
    const size_t lds = tiles <= LDS_TILES_MAX ? sizeof(uint32_t) * (size_t)tiles : 0;
#ifndef GSAJ_PRE_NO_SHREGS
    if (!p.colors_precomp && p.M == 16)
#else
    if (false)
#endif
      if (p.bpw > 1) hipLaunchKernelGGL((k_preprocess<true, true>), dim3(nwg, views), dim3(PRE_BLOCK), lds, s, p, radii, n_touched, g, im, vs);
      else hipLaunchKernelGGL((k_preprocess<true, false>), dim3(nwg, views), dim3(PRE_BLOCK), lds, s, p, radii, n_touched, g, im, vs);
    else
      if (p.bpw > 1) hipLaunchKernelGGL((k_preprocess<false, true>), dim3(nwg, views), dim3(PRE_BLOCK), lds, s, p, radii, n_touched, g, im, vs);
      else hipLaunchKernelGGL((k_preprocess<false, false>), dim3(nwg, views), dim3(PRE_BLOCK), lds, s, p, radii, n_touched, g, im, vs);
    hipLaunchKernelGGL(k_frame_scan, dim3(1, views), dim3(PRE_BLOCK), lds, s, nblk, tiles, p.capacity, g, im, vs);
  }
  GSAJ_HIP_CHECK(hipGetLastError());
  return GSAJ_OK;
}

int launch_tile_binning(int P, int sort_cap, int rec16, int grid_x, int grid_y, const GeomWS &g, const BinWS &b, const ImageWS &im,
                        int views, ViewStrides vs, hipStream_t s) {
  {
    GsajProfScope ps(ST_SCATTER, s);
    const int ltiles = ((grid_y + SCAT_CLS - 1) / SCAT_CLS) * grid_x;
    const size_t lds = ltiles <= SCAT_LDS_TILES ? 3 * sizeof(uint32_t) * (size_t)ltiles : 0;
    // Gaussians per thread: as many as still leave ~1500 workgroups in the launch (4 .. 8; 8 x 256 Gaussians put ~2000-3000
    // instances of one class into the 4096-entry staging area): the longer a workgroup's run of ids per tile, the fewer partial
    // 64-byte pieces leave L2
    int gpt = 4;
    while (gpt < 8 && (long long)P * views * SCAT_CLS / ((long long)PRE_BLOCK * gpt * 2) >= 1536) gpt *= 2;
    // (a launch that misses ONE resident generation -- 6 workgroups x 256 CUs -- by a few workgroups: one or two more blocks each)
    for (int g2 = 9; gpt == 8 && g2 <= 10; g2++) {
      const long long wg8 = (long long)((P + PRE_BLOCK * 8 - 1) / (PRE_BLOCK * 8)) * SCAT_CLS * views;
      const long long wg2 = (long long)((P + PRE_BLOCK * g2 - 1) / (PRE_BLOCK * g2)) * SCAT_CLS * views;
      if (wg8 > 1536 && wg2 <= 1536) gpt = g2;
    }
    const int per = PRE_BLOCK * gpt;
    hipLaunchKernelGGL(k_scatter_instances, dim3((unsigned)((P + per - 1) / per) * SCAT_CLS, views), dim3(PRE_BLOCK), lds, s, P, grid_x,
                       grid_y, gpt, g, im, b.point_list, vs);
    if (rec16)
      hipLaunchKernelGGL(k_pack_splat16, dim3((P + 255) / 256, views), dim3(256), 0, s, P, g, im.counters, vs);
  }
  {
    GsajProfScope ps(ST_TILE_SORT, s);
    int cap = 128;
    while (cap < sort_cap && cap < SORT_CAP) cap <<= 1;
    if (sizeof(uint64_t) * (size_t)cap > 65536)  // lists of 8193 .. 16384 keys: more dynamic LDS than the 64 KB default limit
      (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_tile_sort<512>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)(sizeof(uint64_t) * (size_t)cap));
    // (+ the merge-path splits of a list longer than the capacity: behind the keys below 4096, inside the capacity from there on)
    const size_t split_bytes = cap >= 4096 ? 0 : sizeof(uint32_t) * (PRE_BLOCK + 1);
    // a launch with 32 KB of keys or more per workgroup runs 512 threads per tile (k_tile_sort)
    auto launch = [&](int lds_keys, int pass, size_t extra) {
      const size_t dyn = sizeof(uint64_t) * (size_t)lds_keys + extra;
      if (lds_keys >= 4096)
        hipLaunchKernelGGL(k_tile_sort<512>, dim3(grid_x * grid_y, views), dim3(512), dyn, s, im, g.depths, b.keys_unsorted, b.keys,
                           b.point_list, b.reached, lds_keys, pass, rec16, vs);
      else if (lds_keys > GSAJ_SORT_128_MAX)
        hipLaunchKernelGGL(k_tile_sort<256>, dim3(grid_x * grid_y, views), dim3(256), dyn, s, im, g.depths, b.keys_unsorted, b.keys,
                           b.point_list, b.reached, lds_keys, pass, rec16, vs);
      else
        hipLaunchKernelGGL(k_tile_sort<128>, dim3(grid_x * grid_y, views), dim3(128), dyn, s, im, g.depths, b.keys_unsorted, b.keys,
                           b.point_list, b.reached, lds_keys, pass, rec16, vs);
    };
    if (cap >= 4096) {  // two launches by padded list length: the many lists below half the capacity run with half the LDS
      launch(cap, 1, 0);
      launch(cap / 2, 2, 0);
    } else {
      launch(cap, 0, split_bytes);
    }
  }
  GSAJ_HIP_CHECK(hipGetLastError());
  return GSAJ_OK;
}

int launch_mark_visible(int P, const float *means3D, const float *viewmatrix, uint8_t *present, hipStream_t s) {
  const int nblk = (P + PRE_BLOCK - 1) / PRE_BLOCK;
  hipLaunchKernelGGL(k_mark_visible, dim3(nblk), dim3(PRE_BLOCK), 0, s, P, means3D, viewmatrix, present);
  GSAJ_HIP_CHECK(hipGetLastError());
  return GSAJ_OK;
}
