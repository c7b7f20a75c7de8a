// loss.hip -- tracking / mapping L1 losses and the pixel-gradient seeds of the rasteriser backward in ONE pass
// over the image (gfx950).  SURVEY 8(f)-1.
//
// Semantics: reference utils/slam_utils.py:56-128 -- get_loss_tracking{,_rgb,_rgbd} (opacity-weighted masked L1 on
// the exposure-corrected colour, depth L1 gated by gt_depth > 0.01 and opacity > 0.95) and
// get_loss_mapping{,_rgb,_rgbd} (no opacity weight, no grad_mask, no opacity gate), combined as
// alpha * L_rgb + (1 - alpha) * L_depth (RGB-D) or L_rgb (monocular).  The reference builds these from ~15 full-frame
// torch kernels plus their autograd backward; here one kernel reads colour/depth/opacity + ground truth once
// and writes dL/dcolour, dL/ddepth (what gsaj_rasterize_backward consumes) and, deterministically, the loss value
// and dL/d(exposure a, b): workgroup partials are summed in workgroup order, in fp64, by the last-arriving
// workgroup (ticket; no float atomics).
#include "gsaj_common.h"

#define LOSS_BLOCK 256
#define LOSS_PPT 4  // pixels per thread (strided by the workgroup size: coalesced, 4x the loads in flight, 4x fewer partials)

struct LossParams {
  int W, H, flags;
  float alpha, rgb_thr;
  const float *color, *depth, *opacity, *gt_color, *gt_depth;
  const uint8_t *grad_mask;
  const float *exp_a, *exp_b;
  float *dL_dcolor, *dL_ddepth, *dL_dopacity;
  float *partials;    // [nblocks][4]
  uint32_t *ticket;   // zero between launches (reset by the last workgroup)
  float *out;         // [5]: loss, L_rgb, L_depth, dL/da, dL/db
};

__device__ __forceinline__ float sgn(float x) { return (x > 0.f) ? 1.f : ((x < 0.f) ? -1.f : 0.f); }

__global__ __launch_bounds__(LOSS_BLOCK) void k_loss_seeds(LossParams p) {
  __shared__ float red[4][LOSS_BLOCK / 64];
  __shared__ bool is_last;
  const size_t HW = (size_t)p.W * p.H;
  const bool tracking = p.flags & GSAJ_LOSS_TRACKING, mono = p.flags & GSAJ_LOSS_MONOCULAR, noexp = p.flags & GSAJ_LOSS_NO_EXPOSURE;
  const float ea = noexp ? 1.f : expf(p.exp_a[0]), eb = noexp ? 0.f : p.exp_b[0];
  const float k_rgb = (mono ? 1.f : p.alpha) / (3.f * (float)HW), k_d = (1.f - p.alpha) / (float)HW;
  float s_rgb = 0.f, s_d = 0.f, s_a = 0.f, s_b = 0.f;
#pragma unroll
  for (int q = 0; q < LOSS_PPT; q++) {
    const size_t pix = ((size_t)blockIdx.x * LOSS_PPT + q) * LOSS_BLOCK + threadIdx.x;
    if (pix >= HW) continue;
    const float g0 = p.gt_color[pix], g1 = p.gt_color[HW + pix], g2 = p.gt_color[2 * HW + pix];
    const float c0 = p.color[pix], c1 = p.color[HW + pix], c2 = p.color[2 * HW + pix];
    const float op = p.opacity[pix];
    float m = (g0 + g1 + g2 > p.rgb_thr) ? 1.f : 0.f;
    if (tracking && p.grad_mask) m = p.grad_mask[pix] ? m : 0.f;
    const float wrgb = tracking ? op : 1.f;
    const float r0 = (ea * c0 + eb) * m - g0 * m, r1 = (ea * c1 + eb) * m - g1 * m, r2 = (ea * c2 + eb) * m - g2 * m;
    const float a0 = fabsf(r0), a1 = fabsf(r1), a2 = fabsf(r2);
    const float t0 = wrgb * m * sgn(r0), t1 = wrgb * m * sgn(r1), t2 = wrgb * m * sgn(r2);
    s_rgb += wrgb * (a0 + a1 + a2);
    p.dL_dcolor[pix] = k_rgb * ea * t0;
    p.dL_dcolor[HW + pix] = k_rgb * ea * t1;
    p.dL_dcolor[2 * HW + pix] = k_rgb * ea * t2;
    if (p.dL_dopacity) p.dL_dopacity[pix] = tracking ? k_rgb * (a0 + a1 + a2) : 0.f;
    s_a += ea * (t0 * c0 + t1 * c1 + t2 * c2);
    s_b += t0 + t1 + t2;
    float dd = 0.f;
    if (!mono) {
      const float gd = p.gt_depth[pix], d = p.depth[pix];
      float dm = (gd > 0.01f) ? 1.f : 0.f;
      if (tracking) dm = (op > 0.95f) ? dm : 0.f;
      const float rd = d * dm - gd * dm;
      s_d += fabsf(rd);
      dd = k_d * dm * sgn(rd);
    }
    p.dL_ddepth[pix] = dd;
  }
  // workgroup partials: wave butterfly, then the four waves in order
  float v[4] = {s_rgb, s_d, s_a, s_b};
#pragma unroll
  for (int c = 0; c < 4; c++) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v[c] += __shfl_xor(v[c], o);
  }
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (lane == 0) {
#pragma unroll
    for (int c = 0; c < 4; c++) red[c][wave] = v[c];
  }
  __syncthreads();
  // Hand-off without fences (as in k_gaussian_bwd): lane 0 stores the four workgroup partials write-through
  // (agent-scope atomic stores = sc1), drains them with vmcnt(0) and only then draws its ticket; the workgroup
  // that draws the last ticket reads every partial with sc1 loads.  A __threadfence() here costs a buffer_wbl2
  // of this kernel's 5 MB of dirty seed rows per workgroup (measured: 78 us instead of 6).
  if (threadIdx.x == 0) {
#pragma unroll
    for (int c = 0; c < 4; c++) {
      float s = red[c][0];
      for (int w = 1; w < LOSS_BLOCK / 64; w++) s += red[c][w];
      __hip_atomic_store(&p.partials[(size_t)blockIdx.x * 4 + c], s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    is_last = __hip_atomic_fetch_add(p.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1;
  }
  __syncthreads();
  if (!is_last) return;
  __shared__ double fin[4][LOSS_BLOCK / 64];
  double acc[4] = {0.0, 0.0, 0.0, 0.0};
  for (unsigned b0 = threadIdx.x; b0 < gridDim.x; b0 += 4 * LOSS_BLOCK) {  // unconditional loads, 16 in flight, masked afterwards
    float t[4][4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const unsigned b = b0 + u * LOSS_BLOCK, bc = min(b, gridDim.x - 1);
#pragma unroll
      for (int c = 0; c < 4; c++) {
        const float x = __hip_atomic_load(&p.partials[(size_t)bc * 4 + c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        t[u][c] = b < gridDim.x ? x : 0.f;
      }
    }
#pragma unroll
    for (int u = 0; u < 4; u++)
#pragma unroll
      for (int c = 0; c < 4; c++) acc[c] += (double)t[u][c];
  }
#pragma unroll
  for (int c = 0; c < 4; c++) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc[c] += __shfl_xor(acc[c], o);
  }
  if (lane == 0) {
#pragma unroll
    for (int c = 0; c < 4; c++) fin[c][wave] = acc[c];
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double t[4];
    for (int c = 0; c < 4; c++) {
      t[c] = fin[c][0];
      for (int w = 1; w < LOSS_BLOCK / 64; w++) t[c] += fin[c][w];
    }
    const double n_rgb = 3.0 * (double)HW;
    const double l_rgb = t[0] / n_rgb, l_d = mono ? 0.0 : t[1] / (double)HW;
    p.out[0] = (float)(mono ? l_rgb : (double)p.alpha * l_rgb + (1.0 - (double)p.alpha) * l_d);
    p.out[1] = (float)l_rgb;
    p.out[2] = (float)l_d;
    p.out[3] = noexp ? 0.f : (float)((double)k_rgb * t[2]);
    p.out[4] = noexp ? 0.f : (float)((double)k_rgb * t[3]);
    *p.ticket = 0u;
  }
}

extern "C" size_t gsaj_loss_workspace_bytes(int W, int H) {
  const size_t nblk = ((size_t)W * H + LOSS_BLOCK * LOSS_PPT - 1) / (LOSS_BLOCK * LOSS_PPT);
  return 256 + nblk * 4 * sizeof(float) + 256;
}

extern "C" int gsaj_loss_seeds(int W, int H, int flags, float alpha, float rgb_boundary_threshold, const float *color,
                               const float *depth, const float *opacity, const float *gt_color, const float *gt_depth,
                               const uint8_t *grad_mask, const float *exposure_a, const float *exposure_b,
                               float *dL_dcolor, float *dL_ddepth, float *dL_dopacity, float *out_scalars,
                               void *loss_ws, void *stream) {
  const bool mono = flags & GSAJ_LOSS_MONOCULAR, noexp = flags & GSAJ_LOSS_NO_EXPOSURE;
  if (W <= 0 || H <= 0 || !color || !opacity || !gt_color || !dL_dcolor || !dL_ddepth || !out_scalars || !loss_ws ||
      (!mono && (!depth || !gt_depth)) || (!noexp && (!exposure_a || !exposure_b))) {
    gsaj_set_error("gsaj_loss_seeds: invalid argument (W=%d H=%d flags=%d)", W, H, flags);
    return GSAJ_ERR_INVALID_ARGUMENT;
  }
  LossParams p;
  p.W = W; p.H = H; p.flags = flags; p.alpha = alpha; p.rgb_thr = rgb_boundary_threshold;
  p.color = color; p.depth = depth; p.opacity = opacity; p.gt_color = gt_color; p.gt_depth = gt_depth;
  p.grad_mask = grad_mask; p.exp_a = exposure_a; p.exp_b = exposure_b;
  p.dL_dcolor = dL_dcolor; p.dL_ddepth = dL_ddepth; p.dL_dopacity = dL_dopacity;
  char *base = (char *)(((uintptr_t)loss_ws + 255) & ~(uintptr_t)255);
  p.ticket = (uint32_t *)base;          // the caller zeroes the workspace once, when it allocates it
  p.partials = (float *)(base + 256);
  p.out = out_scalars;
  const unsigned nblk = (unsigned)(((size_t)W * H + LOSS_BLOCK * LOSS_PPT - 1) / (LOSS_BLOCK * LOSS_PPT));
  hipLaunchKernelGGL(k_loss_seeds, dim3(nblk), dim3(LOSS_BLOCK), 0, (hipStream_t)stream, p);
  GSAJ_HIP_CHECK(hipGetLastError());
  return GSAJ_OK;
}
