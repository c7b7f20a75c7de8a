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
#include "loss_terms.h"

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
  const uint32_t *n_valid;  // COMPUTE_LOSS: number of pixels with gt_depth > 0 inside the mask (k_count_valid)
  float *out;         // [5]: loss, L_rgb, L_depth, dL/da, dL/db
  size_t ws_stride;   // batched launch (gridDim.y = views): bytes between consecutive views' workspace blocks
};

__device__ __forceinline__ float sgn(float x) { return loss_sgn(x); }

__global__ __launch_bounds__(LOSS_BLOCK) void k_loss_seeds(LossParams p) {
  __shared__ float red[4][LOSS_BLOCK / 64];
  __shared__ bool is_last;
  const size_t HW = (size_t)p.W * p.H;
  if (blockIdx.y) {  // batched launch: view blockIdx.y of [K,3,H,W] / [K,1,H,W] / [K,H,W] / [K] arrays and K workspace blocks
    const size_t v = blockIdx.y;
    p.color += v * 3 * HW; p.gt_color += v * 3 * HW; p.dL_dcolor += v * 3 * HW;
    p.opacity += v * HW; p.dL_ddepth += v * HW;
    if (p.depth) p.depth += v * HW;
    if (p.gt_depth) p.gt_depth += v * HW;
    if (p.grad_mask) p.grad_mask += v * HW;
    if (p.dL_dopacity) p.dL_dopacity += v * HW;
    if (p.exp_a) p.exp_a += v;
    if (p.exp_b) p.exp_b += v;
    p.out += 5 * v;
    p.partials = reinterpret_cast<float *>(reinterpret_cast<char *>(p.partials) + v * p.ws_stride);
    p.ticket = reinterpret_cast<uint32_t *>(reinterpret_cast<char *>(p.ticket) + v * p.ws_stride);
    p.n_valid = reinterpret_cast<const uint32_t *>(reinterpret_cast<const char *>(p.n_valid) + v * p.ws_stride);
  }
  const bool mono = p.flags & GSAJ_LOSS_MONOCULAR, noexp = p.flags & GSAJ_LOSS_NO_EXPOSURE;
  // compute_loss of the verification harness (Jacobian_test.py:155-196): mask given per pixel, colour term = mean over
  // 3HW, depth term = mean over the valid pixels only, plain sum of the two (no alpha weighting, no exposure)
  const bool cl = p.flags & GSAJ_LOSS_COMPUTE_LOSS;
  const LossConsts L = loss_consts(p.flags, p.alpha, p.rgb_thr, p.exp_a, p.exp_b, HW, cl ? (float)max(p.n_valid[0], 1u) : 1.f);
  const float k_rgb = L.k_rgb;
  float s_rgb = 0.f, s_d = 0.f, s_a = 0.f, s_b = 0.f;
#pragma unroll
  for (int q = 0; q < LOSS_PPT; q++) {
    const size_t pix = ((size_t)blockIdx.x * LOSS_PPT + q) * LOSS_BLOCK + threadIdx.x;
    if (pix >= HW) continue;
    const float g0 = p.gt_color[pix], g1 = p.gt_color[HW + pix], g2 = p.gt_color[2 * HW + pix];
    const float c0 = p.color[pix], c1 = p.color[HW + pix], c2 = p.color[2 * HW + pix];
    const float op = p.opacity[pix];
    const bool mask = p.grad_mask ? p.grad_mask[pix] != 0 : true;
    const float gd = mono ? 0.f : p.gt_depth[pix], d = mono ? 0.f : p.depth[pix];
    const LossPixel o = loss_pixel(L, g0, g1, g2, c0, c1, c2, op, mask, gd, d);  // (loss_terms.h: shared with the fused compositors)
    s_rgb += o.s_rgb; s_d += o.s_d; s_a += o.s_a; s_b += o.s_b;
    p.dL_dcolor[pix] = o.gC0;
    p.dL_dcolor[HW + pix] = o.gC1;
    p.dL_dcolor[2 * HW + pix] = o.gC2;
    if (p.dL_dopacity) p.dL_dopacity[pix] = o.dop;
    p.dL_ddepth[pix] = o.gD;
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
  for (unsigned b = threadIdx.x; b < gridDim.x; b += LOSS_BLOCK) {  // a workgroup's four partials = one 16-byte coherent load
    const uint4 u = gsaj_coherent_load_x4(p.partials + (size_t)b * 4);
    acc[0] += (double)__uint_as_float(u.x);
    acc[1] += (double)__uint_as_float(u.y);
    acc[2] += (double)__uint_as_float(u.z);
    acc[3] += (double)__uint_as_float(u.w);
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
    const double l_rgb = t[0] / n_rgb, l_d = mono ? 0.0 : t[1] / (cl ? (double)max(p.n_valid[0], 1u) : (double)HW);
    p.out[0] = (float)(cl ? (mono ? l_rgb : l_rgb + l_d) : mono ? l_rgb : (double)p.alpha * l_rgb + (1.0 - (double)p.alpha) * l_d);
    p.out[1] = (float)l_rgb;
    p.out[2] = (float)l_d;
    p.out[3] = noexp ? 0.f : (float)((double)k_rgb * t[2]);
    p.out[4] = noexp ? 0.f : (float)((double)k_rgb * t[3]);
    *p.ticket = 0u;
  }
}

// The loss value and dL/d(exposure) of a frame whose forward compositor summed the per-pixel terms itself (render_fwd.hip,
// loss-fused form): one workgroup adds the per-workgroup partials in slot order (fp64, fixed tree) -- the tail of k_loss_seeds
// without its pass over the image.
__global__ __launch_bounds__(LOSS_BLOCK) void k_loss_finalize(FusedLoss fl, int nslots, size_t HW, const uint32_t *__restrict__ aborted,
                                                              float *__restrict__ out) {
  __shared__ double red[4][LOSS_BLOCK];
  if (aborted && aborted[0]) return;  // aborted asynchronous frame: no partials were written; the scalars stay the last frame's
  double acc[4] = {0.0, 0.0, 0.0, 0.0};
  for (int b = threadIdx.x; b < nslots; b += LOSS_BLOCK) {
    const float4 v = reinterpret_cast<const float4 *>(fl.partials)[b];
    acc[0] += (double)v.x; acc[1] += (double)v.y; acc[2] += (double)v.z; acc[3] += (double)v.w;
  }
#pragma unroll
  for (int c = 0; c < 4; c++) red[c][threadIdx.x] = acc[c];
  __syncthreads();
  for (int o = LOSS_BLOCK / 2; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) {
#pragma unroll
      for (int c = 0; c < 4; c++) red[c][threadIdx.x] += red[c][threadIdx.x + o];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const bool mono = fl.flags & GSAJ_LOSS_MONOCULAR, noexp = fl.flags & GSAJ_LOSS_NO_EXPOSURE;
    const LossConsts L = loss_consts(fl.flags, fl.alpha, fl.rgb_thr, fl.exp_a, fl.exp_b, HW, 1.f);
    const double l_rgb = red[0][0] / (3.0 * (double)HW), l_d = mono ? 0.0 : red[1][0] / (double)HW;
    out[0] = (float)(mono ? l_rgb : (double)fl.alpha * l_rgb + (1.0 - (double)fl.alpha) * l_d);
    out[1] = (float)l_rgb;
    out[2] = (float)l_d;
    out[3] = noexp ? 0.f : (float)((double)L.k_rgb * red[2][0]);
    out[4] = noexp ? 0.f : (float)((double)L.k_rgb * red[3][0]);
  }
}

int launch_loss_finalize(const FusedLoss &fl, int nslots, int W, int H, const uint32_t *aborted, float *out_scalars, hipStream_t s) {
  hipLaunchKernelGGL(k_loss_finalize, dim3(1), dim3(LOSS_BLOCK), 0, s, fl, nslots, (size_t)W * H, aborted, out_scalars);
  GSAJ_HIP_CHECK(hipGetLastError());
  return GSAJ_OK;
}

// number of pixels with gt_depth > 0 inside the mask (the denominator of compute_loss's depth term): integer atomics
__global__ __launch_bounds__(LOSS_BLOCK) void k_count_valid(size_t HW, const float *__restrict__ gt_depth,
                                                            const uint8_t *__restrict__ mask, uint32_t *__restrict__ count) {
  uint32_t c = 0;
  for (size_t i = (size_t)blockIdx.x * LOSS_BLOCK + threadIdx.x; i < HW; i += (size_t)gridDim.x * LOSS_BLOCK)
    c += (gt_depth[i] > 0.0f && (!mask || mask[i])) ? 1u : 0u;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) c += (uint32_t)__shfl_xor((int)c, o);
  if ((threadIdx.x & 63) == 0 && c) atomicAdd(count, c);
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
      (!mono && (!depth || !gt_depth)) || (!noexp && !(flags & GSAJ_LOSS_COMPUTE_LOSS) && (!exposure_a || !exposure_b))) {
    gsaj_set_error("gsaj_loss_seeds: invalid argument (W=%d H=%d flags=%d)", W, H, flags);
    return GSAJ_ERR_INVALID_ARGUMENT;
  }
  if (flags & GSAJ_LOSS_COMPUTE_LOSS) flags |= GSAJ_LOSS_NO_EXPOSURE;
  LossParams p;
  p.W = W; p.H = H; p.flags = flags; p.alpha = alpha; p.rgb_thr = rgb_boundary_threshold;
  p.color = color; p.depth = depth; p.opacity = opacity; p.gt_color = gt_color; p.gt_depth = gt_depth;
  p.grad_mask = grad_mask; p.exp_a = exposure_a; p.exp_b = exposure_b;
  p.dL_dcolor = dL_dcolor; p.dL_ddepth = dL_ddepth; p.dL_dopacity = dL_dopacity;
  char *base = (char *)(((uintptr_t)loss_ws + 255) & ~(uintptr_t)255);
  p.ticket = (uint32_t *)base;          // the caller zeroes the workspace once, when it allocates it
  p.partials = (float *)(base + 256);
  p.out = out_scalars;
  p.n_valid = (uint32_t *)(base + 64);
  p.ws_stride = 0;
  if ((flags & GSAJ_LOSS_COMPUTE_LOSS) && !mono) {
    GSAJ_HIP_CHECK(hipMemsetAsync(base + 64, 0, sizeof(uint32_t), (hipStream_t)stream));
    hipLaunchKernelGGL(k_count_valid, dim3(64), dim3(LOSS_BLOCK), 0, (hipStream_t)stream, (size_t)W * H, gt_depth, grad_mask,
                       (uint32_t *)(base + 64));
  }
  const unsigned nblk = (unsigned)(((size_t)W * H + LOSS_BLOCK * LOSS_PPT - 1) / (LOSS_BLOCK * LOSS_PPT));
  hipLaunchKernelGGL(k_loss_seeds, dim3(nblk), dim3(LOSS_BLOCK), 0, (hipStream_t)stream, p);
  GSAJ_HIP_CHECK(hipGetLastError());
  return GSAJ_OK;
}

extern "C" int gsaj_loss_seeds_batch(int K, int W, int H, int flags, float alpha, float rgb_boundary_threshold, const float *color,
                                     const float *depth, const float *opacity, const float *gt_color, const float *gt_depth,
                                     const uint8_t *grad_mask, const float *exposure_a, const float *exposure_b, float *dL_dcolor,
                                     float *dL_ddepth, float *dL_dopacity, float *out_scalars, void *loss_ws, void *stream) {
  const bool mono = flags & GSAJ_LOSS_MONOCULAR, noexp = flags & GSAJ_LOSS_NO_EXPOSURE;
  if (K <= 0 || W <= 0 || H <= 0 || !color || !opacity || !gt_color || !dL_dcolor || !dL_ddepth || !out_scalars || !loss_ws ||
      (!mono && (!depth || !gt_depth)) || (!noexp && (!exposure_a || !exposure_b)) || (flags & GSAJ_LOSS_COMPUTE_LOSS) ||
      ((uintptr_t)loss_ws & 255)) {
    gsaj_set_error("gsaj_loss_seeds_batch: invalid argument (K=%d W=%d H=%d flags=%d; the workspace must be 256-byte aligned; "
                   "GSAJ_LOSS_COMPUTE_LOSS has no batched form)", K, W, H, flags);
    return GSAJ_ERR_INVALID_ARGUMENT;
  }
  LossParams p;
  p.W = W; p.H = H; p.flags = flags; p.alpha = alpha; p.rgb_thr = rgb_boundary_threshold;
  p.color = color; p.depth = depth; p.opacity = opacity; p.gt_color = gt_color; p.gt_depth = gt_depth;
  p.grad_mask = grad_mask; p.exp_a = exposure_a; p.exp_b = exposure_b;
  p.dL_dcolor = dL_dcolor; p.dL_ddepth = dL_ddepth; p.dL_dopacity = dL_dopacity;
  char *base = (char *)loss_ws;
  p.ticket = (uint32_t *)base;
  p.partials = (float *)(base + 256);
  p.out = out_scalars;
  p.n_valid = (uint32_t *)(base + 64);
  p.ws_stride = (gsaj_loss_workspace_bytes(W, H) + 255) & ~(size_t)255;
  const unsigned nblk = (unsigned)(((size_t)W * H + LOSS_BLOCK * LOSS_PPT - 1) / (LOSS_BLOCK * LOSS_PPT));
  hipLaunchKernelGGL(k_loss_seeds, dim3(nblk, (unsigned)K), dim3(LOSS_BLOCK), 0, (hipStream_t)stream, p);
  GSAJ_HIP_CHECK(hipGetLastError());
  return GSAJ_OK;
}

// ---- isotropic regulariser: weight * mean |s_ij - mean_j(s_i.)| over [P,3] scales, and its gradient -----------------
// (compute_loss, Jacobian_test.py:169-171: weight 10; the mapping loss, slam_backend.py:229-231).  Deterministic: workgroup
// partials in fp64, summed in workgroup order by the last-arriving workgroup.
__global__ __launch_bounds__(LOSS_BLOCK) void k_isotropic(int P, int C, float weight, const float *__restrict__ scales,
                                                          float *__restrict__ dL_dscales, int accumulate,
                                                          double *__restrict__ partials, uint32_t *__restrict__ ticket,
                                                          float *__restrict__ out) {
  __shared__ double red[LOSS_BLOCK / 64];
  __shared__ bool is_last;
  const int i = blockIdx.x * LOSS_BLOCK + threadIdx.x;
  const float k = weight / ((float)P * (float)C);
  double s = 0.0;
  if (i < P) {
    float v[3] = {0.f, 0.f, 0.f}, m = 0.f;
    for (int c = 0; c < C; c++) { v[c] = scales[(size_t)i * C + c]; m += v[c]; }
    m /= (float)C;
    float sg[3], ssum = 0.f;
    for (int c = 0; c < C; c++) {
      const float d = v[c] - m;
      s += (double)fabsf(d);
      sg[c] = sgn(d);
      ssum += sg[c];
    }
    if (dL_dscales)
      for (int c = 0; c < C; c++) {
        const float gq = k * (sg[c] - ssum / (float)C);
        dL_dscales[(size_t)i * C + c] = accumulate ? dL_dscales[(size_t)i * C + c] + gq : gq;
      }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = red[0];
    for (int w = 1; w < LOSS_BLOCK / 64; w++) t += red[w];
    __hip_atomic_store(&partials[blockIdx.x], t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    is_last = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1;
  }
  __syncthreads();
  if (!is_last || threadIdx.x != 0) return;
  double t = 0.0;
  for (unsigned b = 0; b < gridDim.x; b++) t += __hip_atomic_load(&partials[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  out[0] = (float)((double)k * t);
  *ticket = 0u;
}

extern "C" size_t gsaj_isotropic_workspace_bytes(int P) {
  return 512 + sizeof(double) * (size_t)((P + LOSS_BLOCK - 1) / LOSS_BLOCK + 1);
}

extern "C" int gsaj_isotropic_loss(int P, int C, float weight, const float *scales, float *dL_dscales, int accumulate,
                                   float *out_loss, void *iso_ws, void *stream) {
  if (P <= 0 || C < 1 || C > 3 || !scales || !out_loss || !iso_ws) {
    gsaj_set_error("gsaj_isotropic_loss: invalid argument (P=%d C=%d)", P, C);
    return GSAJ_ERR_INVALID_ARGUMENT;
  }
  char *base = (char *)(((uintptr_t)iso_ws + 255) & ~(uintptr_t)255);
  hipLaunchKernelGGL(k_isotropic, dim3((P + LOSS_BLOCK - 1) / LOSS_BLOCK), dim3(LOSS_BLOCK), 0, (hipStream_t)stream, P, C, weight,
                     scales, dL_dscales, accumulate, (double *)(base + 256), (uint32_t *)base, out_loss);
  GSAJ_HIP_CHECK(hipGetLastError());
  return GSAJ_OK;
}
