// render_fwd.hip -- per-tile front-to-back alpha compositing of colour + depth (gfx950).
//
// Semantics: forward.cu:406-535 (renderCUDA): power > 0 skipped, alpha = min(0.99, o*exp(power)),
// alpha < 1/255 skipped, stop before T(1-alpha) < 1e-4, n_touched counts T(1-alpha) > 0.5.
//
// MI355X mapping: one 256-thread workgroup per 16x16 tile = four wave64s, each owning an
// 8x8 pixel quadrant (one pixel per lane).  A round stages 256 sorted 48-byte instance
// records in LDS with coalesced loads; the entry loop reads them back as wave-uniform
// (broadcast) ds_read_b128s.  All early-outs are wave-level ballots -- a quadrant whose
// 64 pixels have all saturated stops walking the list, and entries that no lane of the
// quadrant accepts skip the colour fetch -- instead of the reference's block-wide votes.
// n_touched is accumulated into one register per 64 entries (lane l counts entry l) and flushed
// with ONE atomic wave-instruction per 64 entries (the reference issues one atomic per
// pixel per entry, forward.cu:512-514).
#include "gsaj_common.h"
#include "wave_reduce.h"

#define FWD_ROUND 256

__global__ __launch_bounds__(256) void k_render_fwd(int W, int H, int gx, const uint2 *__restrict__ ranges,
                                                    const float4 *__restrict__ records, const float *__restrict__ bg,
                                                    float *__restrict__ final_T, uint32_t *__restrict__ n_contrib,
                                                    float *__restrict__ out_color, float *__restrict__ out_depth,
                                                    float *__restrict__ out_opacity, int *__restrict__ n_touched,
                                                    const uint32_t *__restrict__ counters) {
  __shared__ float4 rec[FWD_ROUND * REC_F4];
  if (counters[4]) return;  // aborted async frame
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int tile = blockIdx.x;
  const int ty = tile / gx, tx = tile - ty * gx;
  const int px = tx * TILE + (wave & 1) * 8 + (lane & 7);
  const int py = ty * TILE + (wave >> 1) * 8 + (lane >> 3);
  const bool inside = px < W && py < H;
  const float pxf = (float)px, pyf = (float)py;
  const float qx0 = (float)(tx * TILE + (wave & 1) * 8), qy0 = (float)(ty * TILE + (wave >> 1) * 8);
  const uint2 range = ranges[tile];

  bool done = !inside;
  float T = 1.0f, Cr = 0.f, Cg = 0.f, Cb = 0.f, Dp = 0.f;
  uint32_t last = 0;

  for (uint32_t base = range.x; base < range.y; base += FWD_ROUND) {
    if (__syncthreads_and(done)) break;  // also fences reuse of rec[]
    const int n = min((uint32_t)FWD_ROUND, range.y - base);
    if (tid < n) {
      const float4 *src = records + (size_t)(base + tid) * REC_F4;
      rec[tid * REC_F4 + 0] = src[0];
      rec[tid * REC_F4 + 1] = src[1];
      rec[tid * REC_F4 + 2] = src[2];
    }
    __syncthreads();
    if (__builtin_amdgcn_ballot_w64(!done) == 0ull) continue;  // this quadrant is finished; keep serving barriers
    for (int jb = 0; jb < n; jb += 64) {
      const int m = min(64, n - jb);
      // lane l tests entry jb+l against this wave's quadrant; the loop then visits only the set bits
      bool rel = false;
      if (lane < m) {
        const float4 q0 = rec[(jb + lane) * REC_F4 + 0];
        const float4 q1 = rec[(jb + lane) * REC_F4 + 1];
        rel = quadrant_relevant(q0.x, q0.y, q1.x, q1.y, q1.z, q1.w, qx0, qy0);
      }
      unsigned long long todo = __builtin_amdgcn_ballot_w64(rel);
      int cnt = 0;  // lane l: #pixels of this wave that count entry jb+l as "touched"
      bool wave_done = false;
      while (todo != 0ull) {
        const int jj = __builtin_ctzll(todo);
        todo &= todo - 1ull;
        const int j = jb + jj;
        const float4 r0 = rec[j * REC_F4 + 0];
        const float4 r1 = rec[j * REC_F4 + 1];
        const float dx = r0.x - pxf, dy = r0.y - pyf;
        const float power = -0.5f * (r1.x * dx * dx + r1.z * dy * dy) - r1.y * dx * dy;
        const float alpha = fminf(0.99f, r1.w * __expf(power));
        const float test_T = T * (1.f - alpha);
        bool ok = !done && power <= 0.0f && alpha >= (1.0f / 255.0f);
        if (ok && test_T < 0.0001f) {
          done = true;
          ok = false;
        }
        if (__builtin_amdgcn_ballot_w64(ok) != 0ull) {
          const float4 r2 = rec[j * REC_F4 + 2];
          if (ok) {
            const float w = alpha * T;
            Cr += r2.x * w;
            Cg += r2.y * w;
            Cb += r2.z * w;
            Dp += r0.z * w;
            T = test_T;
            last = base - range.x + (uint32_t)j + 1u;
          }
          const int touched = __popcll(__builtin_amdgcn_ballot_w64(ok && test_T > 0.5f));
          cnt = (lane == jj) ? touched : cnt;  // each entry is visited once per 64-batch
        }
        if (__builtin_amdgcn_ballot_w64(!done) == 0ull) {
          wave_done = true;
          break;
        }
      }
      if (lane < m && cnt > 0) {
        const uint32_t id = __float_as_uint(rec[(jb + lane) * REC_F4 + 0].w);
        atomicAdd(&n_touched[id], cnt);
      }
      if (wave_done) break;  // whole quadrant saturated
    }
  }

  if (inside) {
    const size_t pid = (size_t)py * W + px;
    const size_t HW = (size_t)H * W;
    final_T[pid] = T;
    n_contrib[pid] = last;
    out_color[pid] = Cr + T * bg[0];
    out_color[HW + pid] = Cg + T * bg[1];
    out_color[2 * HW + pid] = Cb + T * bg[2];
    out_depth[pid] = Dp;
    out_opacity[pid] = 1.f - T;
  }
}

int launch_render_forward(int W, int H, int grid_x, int grid_y, const float *bg, const BinWS &b, const ImageWS &im,
                          float *out_color, float *out_depth, float *out_opacity, int *n_touched, hipStream_t s) {
  {
    GsajProfScope ps(ST_RENDER_FWD, s);
    hipLaunchKernelGGL(k_render_fwd, dim3(grid_x * grid_y), dim3(256), 0, s, W, H, grid_x, im.ranges, b.records, bg,
                     im.final_T, im.n_contrib, out_color, out_depth, out_opacity, n_touched, im.counters);
  }
  GSAJ_HIP_CHECK(hipGetLastError());
  return GSAJ_OK;
}
