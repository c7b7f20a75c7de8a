// render_fwd.hip -- per-tile front-to-back alpha compositing of colour + depth (gfx950).
//
// Semantics: forward.cu:406-535 (renderCUDA): power > 0 skipped, alpha = min(0.99, o*exp(power)),
// alpha < 1/255 skipped, stop before T(1-alpha) < 1e-4, n_touched counts T(1-alpha) > 0.5.
//
// MI355X mapping: one 256-thread workgroup per 16x16 tile = four wave64s, each owning an
// 8x8 pixel quadrant (one pixel per lane).  Each wave stages 64 sorted 48-byte instance
// records at a time in its private LDS area with coalesced loads; the entry loop reads them
// back as wave-uniform (broadcast) ds_read_b128s.  All early-outs are wave-level ballots -- a quadrant whose
// 64 pixels have all saturated stops walking the list, and entries that no lane of the
// quadrant accepts skip the colour fetch -- instead of the reference's block-wide votes.
// n_touched is accumulated into one register per 64 entries (lane l counts entry l) and flushed
// with ONE atomic wave-instruction per 64 entries (the reference issues one atomic per
// pixel per entry, forward.cu:512-514).
#include "gsaj_common.h"
#include "wave_reduce.h"

#define FWD_CHUNK 64  // records staged per wave per trip

GSAJ_TRACE_DEFINE(fwd)

__global__ __launch_bounds__(256) void k_render_fwd(int W, int H, int gx, const uint2 *__restrict__ ranges,
                                                    const float4 *__restrict__ records, const float *__restrict__ bg,
                                                    float *__restrict__ final_T, uint32_t *__restrict__ n_contrib,
                                                    float *__restrict__ out_color, float *__restrict__ out_depth,
                                                    float *__restrict__ out_opacity, int *__restrict__ n_touched,
                                                    const uint32_t *__restrict__ counters) {
  // Each wave (one 8x8 quadrant) walks the tile list on its own: private 64-record staging area, no
  // workgroup barrier anywhere, so a quadrant never waits for a slower neighbour.  The four waves of a
  // tile read the same records; the repeats are served by L1/L2.
  __shared__ float4 rec_all[4 * FWD_CHUNK * REC_F4];
  if (counters[4]) return;  // aborted async frame
  GSAJ_TRACE_BEGIN(fwd)
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  float4 *rec = rec_all + wave * FWD_CHUNK * REC_F4;
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

  if (__builtin_amdgcn_ballot_w64(!done) != 0ull) {
    for (uint32_t base = range.x; base < range.y; base += FWD_CHUNK) {
      const int m = min((uint32_t)FWD_CHUNK, range.y - base);
      // stage: lane l fetches record base+l (coalesced 3 KB), tests it against the quadrant, publishes it
      bool rel = false;
      if (lane < m) {
        const float4 *src = records + (size_t)(base + lane) * REC_F4;
        const float4 q0 = src[0], q1 = src[1], q2 = src[2];
        rec[lane * REC_F4 + 0] = q0;
        rec[lane * REC_F4 + 1] = q1;
        rec[lane * REC_F4 + 2] = q2;
        rel = quadrant_relevant(q0.x, q0.y, q1.x, q1.y, q1.z, q1.w, qx0, qy0);
      }
      unsigned long long todo = __builtin_amdgcn_ballot_w64(rel);
      int cnt = 0;  // lane l: #pixels of this wave that count entry base+l as "touched"
      bool wave_done = false;
      if (todo != 0ull) {
        // software pipeline: the geometry of the NEXT relevant record is requested from LDS before the
        // current one is evaluated (its colour row is fetched only if some lane accepts the entry)
        int jn = __builtin_ctzll(todo);
        todo &= todo - 1ull;
        float4 n0 = rec[jn * REC_F4 + 0], n1 = rec[jn * REC_F4 + 1];
        while (true) {
          const int jj = jn;
          const float4 r0 = n0, r1 = n1;
          const bool more = todo != 0ull;
          if (more) {
            jn = __builtin_ctzll(todo);
            todo &= todo - 1ull;
            n0 = rec[jn * REC_F4 + 0];
            n1 = rec[jn * REC_F4 + 1];
          }
          const float dx = r0.x - pxf, dy = r0.y - pyf;
          const float power = -0.5f * (r1.x * dx * dx + r1.z * dy * dy) - r1.y * dx * dy;
          const float alpha = fminf(0.99f, r1.w * __expf(power));
          const float test_T = T * (1.f - alpha);
          bool ok = !done && power <= 0.0f && alpha >= (1.0f / 255.0f);
          if (ok && test_T < 0.0001f) {  // this pixel is saturated: stop before this entry
            done = true;
            ok = false;
          }
          if (__builtin_amdgcn_ballot_w64(ok) != 0ull) {
            const float4 r2 = rec[jj * REC_F4 + 2];
            if (ok) {
              const float w = alpha * T;
              Cr += r2.x * w;
              Cg += r2.y * w;
              Cb += r2.z * w;
              Dp += r0.z * w;
              T = test_T;
              last = base - range.x + (uint32_t)jj + 1u;
            }
            const int touched = __popcll(__builtin_amdgcn_ballot_w64(ok && test_T > 0.5f));
            cnt = (lane == jj) ? touched : cnt;  // each entry is visited once per chunk
          }
          if (__builtin_amdgcn_ballot_w64(!done) == 0ull) {
            wave_done = true;
            break;
          }
          if (!more) break;
        }
      }
      if (lane < m && cnt > 0) {
        const uint32_t id = __float_as_uint(rec[lane * REC_F4 + 0].w);
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
  GSAJ_TRACE_END(fwd)
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
