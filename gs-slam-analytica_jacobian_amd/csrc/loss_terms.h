// loss_terms.h -- the per-pixel arithmetic of the tracking / mapping L1 losses (reference utils/slam_utils.py:56-128), shared by
// the stand-alone loss kernel (loss.hip: k_loss_seeds) and by the loss-fused forms of the two compositors (render_fwd.hip:
// the forward's epilogue sums the loss terms; render_bwd.hip: the reverse compositor's prologue re-derives the pixel's
// gradient seeds instead of reading them from memory).  ONE definition, written with explicitly rounded operations, so that
// the three kernels produce the same bits for a pixel whatever each compiler pass would otherwise contract.
#pragma once
#include "gsaj_common.h"

struct LossConsts {
  float ea, eb;        // exposure: image_ab = exp(a) * image + b  (1, 0 with GSAJ_LOSS_NO_EXPOSURE)
  float k_rgb, k_d;    // d loss / d (sum of the colour / depth terms)
  float rgb_thr;
  bool tracking, mono, cl;
};

// Loss terms fused into the compositors (gsaj_rasterize_forward_loss / gsaj_rasterize_backward_loss): what the stand-alone
// kernel takes as arguments, by value in the compositors' kernel arguments.
struct FusedLoss {
  int flags;
  float alpha, rgb_thr;
  const float *gt_color, *gt_depth;    // [3,H,W], [H,W] (NULL: monocular)
  const uint8_t *grad_mask;            // [H,W] or NULL
  const float *exp_a, *exp_b;          // device scalars (NULL with GSAJ_LOSS_NO_EXPOSURE)
  const float *color, *depth, *opacity;  // backward: the images the forward wrote
  float *partials;                     // forward: [workgroups][4] sums of (colour term, depth term, d/da term, d/db term)
};

#ifdef __HIPCC__
__device__ __forceinline__ float loss_sgn(float x) { return (x > 0.f) ? 1.f : ((x < 0.f) ? -1.f : 0.f); }

__device__ __forceinline__ LossConsts loss_consts(int flags, float alpha, float rgb_thr, const float *exp_a, const float *exp_b,
                                                  size_t HW, float n_valid) {
  LossConsts L;
  L.tracking = flags & GSAJ_LOSS_TRACKING; L.mono = flags & GSAJ_LOSS_MONOCULAR; L.cl = flags & GSAJ_LOSS_COMPUTE_LOSS;
  const bool noexp = flags & GSAJ_LOSS_NO_EXPOSURE;
  L.ea = noexp ? 1.f : expf(exp_a[0]);
  L.eb = noexp ? 0.f : exp_b[0];
  L.k_rgb = L.cl ? 1.f / (3.f * (float)HW) : (L.mono ? 1.f : alpha) / (3.f * (float)HW);
  L.k_d = L.cl ? 1.f / n_valid : (1.f - alpha) / (float)HW;
  L.rgb_thr = rgb_thr;
  return L;
}

struct LossPixel {
  float gC0, gC1, gC2, gD;  // dL/dcolour, dL/ddepth of the pixel: the seeds of the reverse compositor
  float s_rgb, s_d, s_a, s_b;  // the pixel's contributions to the four sums the loss value and dL/d(exposure) are made of
  float dop;                // tracking: d(colour term)/d(opacity weight)
};

// mask: grad_mask[pix] != 0 (true where there is no mask); gd, d: ground-truth and rendered depth (ignored when monocular)
__device__ __forceinline__ LossPixel loss_pixel(const LossConsts &L, float g0, float g1, float g2, float c0, float c1, float c2,
                                                float op, bool mask, float gd, float d) {
  LossPixel o;
  float m = (L.cl || __fadd_rn(__fadd_rn(g0, g1), g2) > L.rgb_thr) ? 1.f : 0.f;
  if (L.tracking || L.cl) m = mask ? m : 0.f;
  const float wrgb = L.tracking ? op : 1.f;
  const float r0 = __fsub_rn(__fmul_rn(__builtin_fmaf(L.ea, c0, L.eb), m), __fmul_rn(g0, m));
  const float r1 = __fsub_rn(__fmul_rn(__builtin_fmaf(L.ea, c1, L.eb), m), __fmul_rn(g1, m));
  const float r2 = __fsub_rn(__fmul_rn(__builtin_fmaf(L.ea, c2, L.eb), m), __fmul_rn(g2, m));
  const float a0 = fabsf(r0), a1 = fabsf(r1), a2 = fabsf(r2);
  const float wm = __fmul_rn(wrgb, m);
  const float t0 = __fmul_rn(wm, loss_sgn(r0)), t1 = __fmul_rn(wm, loss_sgn(r1)), t2 = __fmul_rn(wm, loss_sgn(r2));
  const float asum = __fadd_rn(__fadd_rn(a0, a1), a2);
  const float ke = __fmul_rn(L.k_rgb, L.ea);
  o.s_rgb = __fmul_rn(wrgb, asum);
  o.gC0 = __fmul_rn(ke, t0);
  o.gC1 = __fmul_rn(ke, t1);
  o.gC2 = __fmul_rn(ke, t2);
  o.dop = L.tracking ? __fmul_rn(L.k_rgb, asum) : 0.f;
  o.s_a = __fmul_rn(L.ea, __fadd_rn(__fadd_rn(__fmul_rn(t0, c0), __fmul_rn(t1, c1)), __fmul_rn(t2, c2)));
  o.s_b = __fadd_rn(__fadd_rn(t0, t1), t2);
  o.s_d = 0.f;
  o.gD = 0.f;
  if (!L.mono) {
    float dm = (gd > (L.cl ? 0.0f : 0.01f)) ? 1.f : 0.f;
    if (L.tracking) dm = (op > 0.95f) ? dm : 0.f;
    if (L.cl) dm *= m;
    const float rd = __fsub_rn(__fmul_rn(d, dm), __fmul_rn(gd, dm));
    o.s_d = fabsf(rd);
    o.gD = __fmul_rn(__fmul_rn(L.k_d, dm), loss_sgn(rd));
  }
  return o;
}
#endif
