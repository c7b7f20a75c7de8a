// render_bwd.hip -- per-tile reverse compositor: per-pixel dL/dC, dL/dD -> per-instance
// partial gradients of (mean2D, conic, opacity, colour, depth)  (gfx950).
//
// Semantics: backward.cu:648-872 (renderCUDA backward): walk the tile list back to front from
// each pixel's last contributor, T <- T/(1-alpha), recurrences accum_rec / accum_rec_depth,
// dL/dalpha incl. the background term, dL/dmean2D scaled by (W/2, H/2).
//
// MI355X design (differs from the reference on purpose):
//  * the reference reduces the 10 per-pixel partials of EVERY list entry with a 256-thread
//    shared-memory tree (8 barrier rounds, backward.cu:633-644) and then issues 10 global float
//    atomics.  Here each wave64 (an 8x8 pixel quadrant) reduces its 10 partials in registers:
//    v_permlane32_swap / v_permlane16_swap merge two registers per instruction across the
//    half-wave and row boundaries (10 -> 5 -> 3 registers), then four DPP row rotations finish
//    the 16-lane rows.  ~29 VALU ops per entry, no barrier, no LDS traffic for the reduction.
//  * no global atomics at all: the four waves of a tile write their totals to private LDS slots,
//    and after each round of BWD_ROUND entries the workgroup stores one 48-byte partial-gradient
//    row per (tile, Gaussian) instance, coalesced, at the instance's sorted position.  The
//    per-Gaussian kernel (gaussian_bwd.hip) sums a Gaussian's instances in a fixed order, so
//    gradients are bit-reproducible run to run (the reference's float atomics are not).
//  * entries beyond the quadrant's / tile's furthest last-contributor are never visited.
#include "gsaj_common.h"
#include "wave_reduce.h"

#define BWD_ROUND 128

__global__ __launch_bounds__(256) void k_render_bwd(int W, int H, int gx, const uint2 *__restrict__ ranges,
                                                    const float4 *__restrict__ records, const float *__restrict__ bg,
                                                    const float *__restrict__ final_T,
                                                    const uint32_t *__restrict__ n_contrib,
                                                    const float *__restrict__ dL_dpix,
                                                    const float *__restrict__ dL_dpix_depth,
                                                    float4 *__restrict__ inst_grad,
                                                    const uint32_t *__restrict__ counters) {
  __shared__ float4 rec[BWD_ROUND * REC_F4];
  if (counters[4]) return;  // aborted async frame
  __shared__ float acc[BWD_ROUND * 4 * IGRAD_F];  // [entry][wave][12]
  __shared__ uint32_t wave_max[4];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int tile = blockIdx.x;
  const int ty = tile / gx, tx = tile - ty * gx;
  const int px = tx * TILE + (wave & 1) * 8 + (lane & 7);
  const int py = ty * TILE + (wave >> 1) * 8 + (lane >> 3);
  const bool inside = px < W && py < H;
  const float pxf = (float)px, pyf = (float)py;
  const float qx0 = (float)(tx * TILE + (wave & 1) * 8), qy0 = (float)(ty * TILE + (wave >> 1) * 8);
  const uint2 range = ranges[tile];
  const size_t pid = (size_t)py * W + px, HW = (size_t)H * W;

  const float T_final = inside ? final_T[pid] : 0.f;
  float T = T_final;
  const uint32_t last = inside ? n_contrib[pid] : 0u;
  float gC0 = 0.f, gC1 = 0.f, gC2 = 0.f, gD = 0.f;
  if (inside) {
    gC0 = dL_dpix[pid];
    gC1 = dL_dpix[HW + pid];
    gC2 = dL_dpix[2 * HW + pid];
    gD = dL_dpix_depth[pid];
  }
  const float bg_dot = bg[0] * gC0 + bg[1] * gC1 + bg[2] * gC2;
  const float ddelx_dx = 0.5f * W, ddely_dy = 0.5f * H;

  // furthest last-contributor of this quadrant and of the tile
  uint32_t wmax = last;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) wmax = max(wmax, (uint32_t)__shfl_xor((int)wmax, o));
  if (lane == 0) wave_max[wave] = wmax;
  __syncthreads();
  const uint32_t bmax = max(max(wave_max[0], wave_max[1]), max(wave_max[2], wave_max[3]));

  float accC0 = 0.f, accC1 = 0.f, accC2 = 0.f, accD = 0.f;  // accum_rec, accum_rec_depth

  uint32_t hi = range.x + bmax;  // exclusive sorted position
  // entries [hi, range.y) were never reached by any pixel of the tile: their partials are zero
  for (uint32_t k = hi + tid; k < range.y; k += 256) {
    float4 *dst = inst_grad + (size_t)__float_as_uint(records[(size_t)k * REC_F4 + 2].w) * REC_F4;
    dst[0] = make_float4(0.f, 0.f, 0.f, 0.f);
    dst[1] = make_float4(0.f, 0.f, 0.f, 0.f);
    dst[2] = make_float4(0.f, 0.f, 0.f, 0.f);
  }

  while (hi > range.x) {
    const uint32_t lo = (hi - range.x > BWD_ROUND) ? hi - BWD_ROUND : range.x;
    const int n = (int)(hi - lo);
    if (tid < n) {
      const float4 *src = records + (size_t)(lo + tid) * REC_F4;
      rec[tid * REC_F4 + 0] = src[0];
      rec[tid * REC_F4 + 1] = src[1];
      rec[tid * REC_F4 + 2] = src[2];
    }
    {
      float4 *z = reinterpret_cast<float4 *>(acc);
      for (int i = tid; i < n * 4 * (IGRAD_F / 4); i += 256) z[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    __syncthreads();

    const uint32_t first_idx = lo - range.x;  // list index (0-based) of rec[0]
    if (wmax > first_idx) {
      for (int jb = ((n - 1) >> 6) << 6; jb >= 0; jb -= 64) {
        // lane l tests entry jb+l against this wave's quadrant (and its furthest last contributor);
        // the loop then visits only the set bits, back to front
        bool rel = false;
        if (jb + lane < n && first_idx + (uint32_t)(jb + lane) < wmax) {
          const float4 q0 = rec[(jb + lane) * REC_F4 + 0];
          const float4 q1 = rec[(jb + lane) * REC_F4 + 1];
          rel = quadrant_relevant(q0.x, q0.y, q1.x, q1.y, q1.z, q1.w, qx0, qy0);
        }
        unsigned long long todo = __builtin_amdgcn_ballot_w64(rel);
        if (todo == 0ull) continue;
        // software pipeline: the record of the NEXT relevant entry is requested from LDS before the
        // current one is processed, so its ~100-cycle read latency hides under ~90 VALU instructions
        int jj = 63 - __builtin_clzll(todo);
        todo &= ~(1ull << jj);
        float4 n0 = rec[(jb + jj) * REC_F4 + 0], n1 = rec[(jb + jj) * REC_F4 + 1], n2 = rec[(jb + jj) * REC_F4 + 2];
        while (true) {
          const int j = jb + jj;
          const uint32_t idx = first_idx + (uint32_t)j;
          const float4 r0 = n0, r1 = n1, r2 = n2;
          const bool more = todo != 0ull;
          if (more) {
            jj = 63 - __builtin_clzll(todo);
            todo &= ~(1ull << jj);
            n0 = rec[(jb + jj) * REC_F4 + 0];
            n1 = rec[(jb + jj) * REC_F4 + 1];
            n2 = rec[(jb + jj) * REC_F4 + 2];
          }
        const float dx = r0.x - pxf, dy = r0.y - pyf;
        const float power = -0.5f * (r1.x * dx * dx + r1.z * dy * dy) - r1.y * dx * dy;
        const float G0 = __expf(power);
        const float alpha0 = fminf(0.99f, r1.w * G0);
        const bool valid = idx < last && power <= 0.0f && alpha0 >= (1.0f / 255.0f);
        if (__builtin_amdgcn_ballot_w64(valid) != 0ull) {
        // A lane that skips this entry runs the same arithmetic with alpha = G = 0: T, accum_rec and
        // every partial then come out unchanged / zero, so two selects replace ~20 predicated updates.
        const float alpha = valid ? alpha0 : 0.f;
        const float G = valid ? G0 : 0.f;
        const float inv1ma = __builtin_amdgcn_rcpf(1.f - alpha);
        T = T * inv1ma;  // T <- T / (1 - alpha)
        const float dchannel = alpha * T;
        float dL_dalpha = (r2.x - accC0) * gC0 + (r2.y - accC1) * gC1 + (r2.z - accC2) * gC2 + (r0.z - accD) * gD;
        dL_dalpha = dL_dalpha * T - (T_final * inv1ma) * bg_dot;
        // accum_rec for the next (nearer) entry: alpha c + (1 - alpha) accum_rec  (backward.cu:799,811,
        // applied here instead of lazily at the top of the next iteration -- same arithmetic)
        const float oma = 1.f - alpha;
        accC0 = alpha * r2.x + oma * accC0;
        accC1 = alpha * r2.y + oma * accC1;
        accC2 = alpha * r2.z + oma * accC2;
        accD = alpha * r0.z + oma * accD;
        const float dL_dG = r1.w * dL_dalpha;
        const float gdx = G * dx, gdy = G * dy;
        const float dG_ddelx = -gdx * r1.x - gdy * r1.y;
        const float dG_ddely = -gdy * r1.z - gdx * r1.y;
        float v[10];
        v[0] = dL_dG * dG_ddelx * ddelx_dx;
        v[1] = dL_dG * dG_ddely * ddely_dy;
        v[2] = -0.5f * gdx * dx * dL_dG;
        v[3] = -0.5f * gdx * dy * dL_dG;
        v[4] = -0.5f * gdy * dy * dL_dG;
        v[5] = G * dL_dalpha;
        v[6] = dchannel * gC0;
        v[7] = dchannel * gC1;
        v[8] = dchannel * gC2;
        v[9] = dchannel * gD;
        // ---- wave reduction of the 10 partials (registers only), totals -> this wave's LDS slot ----
        float x0, x1, x2;
        reduce10(v, x0, x1, x2);
        store10(acc + (j * 4 + wave) * IGRAD_F, lane, x0, x1, x2);
        }
          if (!more) break;
        }
      }
    }
    __syncthreads();
    if (tid < n) {
      const float4 *a = reinterpret_cast<const float4 *>(acc) + tid * 4 * (IGRAD_F / 4);
      float4 s0 = a[0], s1 = a[1], s2 = a[2];
#pragma unroll
      for (int w = 1; w < 4; w++) {
        const float4 b0 = a[w * 3 + 0], b1 = a[w * 3 + 1], b2 = a[w * 3 + 2];
        s0.x += b0.x; s0.y += b0.y; s0.z += b0.z; s0.w += b0.w;
        s1.x += b1.x; s1.y += b1.y; s1.z += b1.z; s1.w += b1.w;
        s2.x += b2.x; s2.y += b2.y;
      }
      s2.z = 0.f; s2.w = 0.f;
      float4 *dst = inst_grad + (size_t)__float_as_uint(rec[tid * REC_F4 + 2].w) * REC_F4;  // emission slot
      dst[0] = s0;
      dst[1] = s1;
      dst[2] = s2;
    }
    __syncthreads();
    hi = lo;
  }
}

int launch_render_backward(int R, int W, int H, int grid_x, int grid_y, const float *bg, const BinWS &b,
                           const ImageWS &im, const float *dL_dpix, const float *dL_dpix_depth, hipStream_t s) {
  if (R <= 0) return GSAJ_OK;  // (async callers pass the arena capacity as R)
  {
    GsajProfScope ps(ST_RENDER_BWD, s);
    hipLaunchKernelGGL(k_render_bwd, dim3(grid_x * grid_y), dim3(256), 0, s, W, H, grid_x, im.ranges, b.records, bg,
                     im.final_T, im.n_contrib, dL_dpix, dL_dpix_depth, b.inst_grad, im.counters);
  }
  GSAJ_HIP_CHECK(hipGetLastError());
  return GSAJ_OK;
}
