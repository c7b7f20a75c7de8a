// render_bwd.hip -- per-tile reverse compositor: per-pixel dL/dC, dL/dD -> per-instance
// partial gradients of (mean2D, conic, opacity, colour, depth)  (gfx950).
//
// Semantics: backward.cu:648-872 (renderCUDA backward): walk the tile list back to front from
// each pixel's last contributor, T <- T/(1-alpha), recurrences accum_rec / accum_rec_depth,
// dL/dalpha incl. the background term, dL/dmean2D scaled by (W/2, H/2).
//
// MI355X design (differs from the reference on purpose).  The reference reduces the 10 per-pixel
// partials of EVERY list entry with a 256-thread shared-memory tree (8 barrier rounds,
// backward.cu:633-644) and then issues 10 global float atomics.  Here a wave64 owns an 8x8 pixel
// quadrant and works in two phases per batch of 8 accepted entries:
//
//  phase 1 (lane = pixel):  the sequential part only.  For an entry the quadrant can see
//    (lane-parallel culling, see wave_reduce.h) each lane advances T and accum_rec and produces
//    just TWO scalars: w = dL/dG * G and u = alpha * T.  All 10 partials are linear in (w, u):
//      d/dmean2D ~ w * (conic . d),  d/dconic ~ w * d d^T,  d/dopacity = w / o,  d/dcolour = u * dL/dC,
//      d/ddepth = u * dL/dD.        (w, u) go to LDS, one 64-pixel row per entry.
//  phase 2 (lane = (entry slot, pixel row)):  8 slots x 8 pixel rows = 64 lanes.  Each lane walks
//    the 8 pixels of its row with three running sums -- w, w dx, w dx^2, dx the pixel's own distance
//    from the Gaussian's mean; dy is constant along a row, so these give every second moment about
//    the mean -- and the 4 products u * seed; the 8 rows of a slot are then combined with
//    three register-merge steps (v_permlane32_swap, v_permlane16_swap, one DPP rotation) -- against
//    ~29 cross-lane instructions (~80 plain-VALU issue slots, measured) for reducing the 10 partials
//    of every entry across the wave directly.
//
//  No global atomics: the four waves of a tile write per-entry totals to private LDS slots, and
//  after each round the workgroup stores one 48-byte partial-gradient row per (tile, Gaussian)
//  instance at the instance's emission slot.  The per-Gaussian kernel (gaussian_bwd.hip) sums a
//  Gaussian's rows in a fixed order, so gradients are bit-reproducible run to run (the reference's
//  float atomics are not).  Entries beyond the quadrant's / tile's furthest last-contributor are
//  never visited -- not even fetched: the list is a list of Gaussian ids, and the 48-byte rows they name
//  (GeomWS.splat) are gathered per round of 48 entries -- one 16-byte piece per wave and entry, a round ahead of their use, the
//  ids two rounds ahead (see the staging below).  Rows beyond the tile's furthest
//  last contributor keep the one-byte `reached = 0` flag the tile sort gave them instead of a zero row.
//  Workgroups take tiles longest list first (ImageWS.tile_order, written by the preprocess kernel's frame scan).
#include "gsaj_common.h"
#include "loss_terms.h"
#include "wave_reduce.h"

#define BWD_ROUND 48   // list entries staged per workgroup round: 31.1 KB LDS + 96 VGPRs = five resident workgroups per CU (all 1200 tiles of a 640x480
                       // frame at once).  56 (with ACC_STRIDE 56: 32.6 KB) measured 460 against 426-445 us; 64 needs cheaper accumulators, and
                       // sharing them by wave pairs through ds_add_f32 measured 520 us (an LDS float atomic costs hundreds of cycles)
#define SLOTS 8        // accepted entries per phase-2 batch
#define WU_STRIDE 65   // float2 per slot row (64 pixels + 1: conflict-free ds_read_b64 in phase 2)
#define ACC_C 10       // partials per (entry, wave)
#define ACC_STRIDE (BWD_ROUND + 1)  // acc[wave][partial][entry]: the end-of-round merge reads consecutive words

GSAJ_TRACE_DEFINE(bwd)
GSAJ_TRACE_DEFINE(bwdph)  // per-wave phase times of the rounds (trace build; tools/batch_trace.py)
#ifdef GSAJ_BLOCK_TRACE
#define BWD_PH(i) { const unsigned long long now_ = wall_clock64(); ph_[i] += (unsigned)(now_ - pht_); pht_ = now_; }
#else
#define BWD_PH(i)
#endif

// LOSS: the loss-fused form (gsaj_rasterize_backward_loss): a pixel's seeds dL/dC, dL/dD are derived in the
// prologue from the images the forward wrote and the ground truth (loss_terms.h: the stand-alone loss kernel's arithmetic, same
// bits) instead of being read from a seed image -- that image is never materialised.
template <bool LOSS>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(5, 5))) void k_render_bwd(int W, int H, int gx, ImageWS im,
                                                    const uint32_t *__restrict__ point_list, GeomWS g,
                                                    const float *__restrict__ bg,
                                                    const float *__restrict__ dL_dpix,
                                                    const float *__restrict__ dL_dpix_depth,
                                                    float4 *__restrict__ inst_grad, uint8_t *__restrict__ reached, ViewStrides vs,
                                                    FusedLoss fl) {
  {  // batched launch: blockIdx.y = view
    const size_t view = blockIdx.y, HWv = (size_t)H * W;
    im = image_view(im, view * vs.image);
    point_list = gsaj_shift(point_list, view * vs.bin);
    g = geom_view(g, view * vs.geom);
    inst_grad = gsaj_shift(inst_grad, view * vs.bin);
    reached = gsaj_shift(reached, view * vs.bin);
    dL_dpix += view * 3 * HWv;
    dL_dpix_depth += view * HWv;
  }
  const uint2 *__restrict__ ranges = im.ranges;
  const float *__restrict__ final_T = im.final_T;
  const uint32_t *__restrict__ n_contrib = im.n_contrib;
  const uint32_t *__restrict__ counters = im.counters;
  __shared__ float4 rec[BWD_ROUND * REC_F4];
  __shared__ __attribute__((aligned(16))) float acc[4 * ACC_C * ACC_STRIDE];  // [wave][partial][entry]
  __shared__ float2 wu_all[4 * SLOTS * WU_STRIDE];     // [wave][slot][pixel] (w, u)
  __shared__ float4 seed_all[4 * 64];                  // [wave][pixel] (dL/dC rgb, dL/dD)
  __shared__ uint32_t wave_max[4];
  __shared__ uint32_t blk_first[BWD_ROUND];
  if (counters[4]) return;  // aborted async frame
  GSAJ_TRACE_BEGIN(bwd)
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  float2 *wu = wu_all + wave * SLOTS * WU_STRIDE;
  float4 *seed = seed_all + wave * 64;
  // longest lists first (frame_scan's schedule; clamped: a stale entry must not become an out-of-range tile)
  const int tile = __builtin_amdgcn_readfirstlane((int)min(im.tile_order[blockIdx.x], gridDim.x - 1));
  const int ty = tile / gx, tx = tile - ty * gx;
  const int px = tx * TILE + (wave & 1) * 8 + (lane & 7);
  const int py = ty * TILE + (wave >> 1) * 8 + (lane >> 3);
  const bool inside = px < W && py < H;
  const float pxf = (float)px, pyf = (float)py;
  const float qx0 = (float)(tx * TILE + (wave & 1) * 8), qy0 = (float)(ty * TILE + (wave >> 1) * 8);
  const uint2 range_v = ranges[tile];  // workgroup-uniform: into scalar registers, with everything derived from it (hi, lo, n, first_idx)
  const uint2 range = make_uint2((uint32_t)__builtin_amdgcn_readfirstlane((int)range_v.x), (uint32_t)__builtin_amdgcn_readfirstlane((int)range_v.y));
  const size_t pid = (size_t)py * W + px, HW = (size_t)H * W;

  const float T_final = inside ? final_T[pid] : 0.f;
  float T = T_final;
  const uint32_t last = inside ? n_contrib[pid] : 0u;
  float gC0 = 0.f, gC1 = 0.f, gC2 = 0.f, gD = 0.f;
  if (inside) {
    if (LOSS) {
      const LossConsts L = loss_consts(fl.flags, fl.alpha, fl.rgb_thr, fl.exp_a, fl.exp_b, HW, 1.f);
      const bool mask = fl.grad_mask ? fl.grad_mask[pid] != 0 : true;
      const float gd = L.mono ? 0.f : fl.gt_depth[pid], d = L.mono ? 0.f : fl.depth[pid];
      const LossPixel o = loss_pixel(L, fl.gt_color[pid], fl.gt_color[HW + pid], fl.gt_color[2 * HW + pid], fl.color[pid], fl.color[HW + pid],
                                     fl.color[2 * HW + pid], fl.opacity[pid], mask, gd, d);
      gC0 = o.gC0; gC1 = o.gC1; gC2 = o.gC2; gD = o.gD;
    } else {
      gC0 = dL_dpix[pid];
      gC1 = dL_dpix[HW + pid];
      gC2 = dL_dpix[2 * HW + pid];
      gD = dL_dpix_depth[pid];
    }
  }
  seed[lane] = make_float4(gC0, gC1, gC2, gD);
  const float Tf_bg = T_final * (bg[0] * gC0 + bg[1] * gC1 + bg[2] * gC2);
  const float ddelx_dx = 0.5f * W, ddely_dy = 0.5f * H;

  // furthest last-contributor of this quadrant and of the tile
  uint32_t wmax = last;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) wmax = max(wmax, (uint32_t)__shfl_xor((int)wmax, o));
  if (lane == 0) wave_max[wave] = wmax;
  __syncthreads();
  const uint32_t bmax = max(max(wave_max[0], wave_max[1]), max(wave_max[2], wave_max[3]));

  float accS = 0.f;  // (accum_rec, accum_rec_depth) . (dL/dC, dL/dD) of this pixel: the only form the recurrences are needed in

  // phase-2 lane roles
  const int p2_slot = lane & (SLOTS - 1), p2_row = lane >> 3;
  const float p2_py = qy0 + (float)p2_row;

  uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(range.x + bmax));  // exclusive sorted position (workgroup-uniform: kept in scalar registers)
  // entries [hi, range.y) were never reached by any pixel of the tile: their partials are zero, and their `reached` flags
  // stay as the tile sort left them (opaque scenes leave most of a long list unreached: neither their ids nor their rows
  // are ever fetched)
  const bool rec16 = counters[7] != 0u;  // fp16-storage rows (gsaj_common.h)
  // Staging, software-pipelined over the rounds.  A round's 48 rows are gathered through two dependent loads (id, then the row
  // it names): waited for at the round's start, that latency was a tenth of a wave's life (per-round phase times of the
  // trace build, tools/batch_trace.py).  Now thread (part, e) = (tid / 48, tid % 48), part < 4, owns ONE 16-byte piece of entry e
  // -- the row's three float4s and the Gaussian's block offset -- requests round r+1's piece at the start of round r (its id
  // was requested a round earlier still) and writes it to LDS at the start of round r+1: four registers per thread instead
  // of a whole row, and nothing is waited for but the very first round.  (part = wave: which array a wave reads is decided by
  // scalar code; every wave issues ONE 16-byte load per round from base(part) + offset(lane).)
  const int part = __builtin_amdgcn_readfirstlane(wave), e = lane;
  auto round_lo = [&](uint32_t h) { return (h - range.x > BWD_ROUND) ? h - BWD_ROUND : range.x; };
  auto load_id = [&](uint32_t l, uint32_t h) -> uint32_t {  // id of entry e of the round [l, h)
    // (no branch around the load: a lane without an entry reads the tile's first id -- the value a load leaves in its
    // registers can stay in flight across the loop's back edge only if no copy merges it with another definition)
    const bool has = e < BWD_ROUND && l + (uint32_t)e < h;
    const uint32_t v = gsaj_load_u32_global(point_list + (has ? l + (uint32_t)e : range.x));
    return has ? v : 0xffffffffu;
  };
  // base and stride of this wave's piece (wave-uniform).  The 16-byte load of the 4-byte block offset / the 8-byte rectangle reads
  // on into the next elements of the same workspace array (its last element: into the carve's alignment gap / the next array).
  const char *piece_base;
  uint32_t piece_stride;
  if (part == 3) piece_base = reinterpret_cast<const char *>(g.block_sums), piece_stride = 0u;
  else if (!rec16) piece_base = reinterpret_cast<const char *>(g.splat + part), piece_stride = 16u * REC_F4;
  else if (part < 2) piece_base = reinterpret_cast<const char *>(g.splat16 + part), piece_stride = 16u * REC16_F4;
  else piece_base = reinterpret_cast<const char *>(g.scat), piece_stride = 8u;
  auto load_piece = [&](uint32_t id) -> float4 {
    const uint32_t i = id != 0xffffffffu ? id : 0u;  // (no entry: element 0, never used)
    const size_t off = piece_stride ? (size_t)i * piece_stride : (size_t)(i / PRE_BLOCK) * 4u;
    return gsaj_load_f4_unaligned(piece_base + off);
  };
  uint32_t id_cur = 0xffffffffu, id_nxt = 0xffffffffu;  // ids of entry e in the round being staged next / the one after
  float4 piece = make_float4(0.f, 0.f, 0.f, 0.f);
  if (hi > range.x) {
    const uint32_t lo0 = round_lo(hi);
    id_cur = load_id(lo0, hi);
    id_nxt = load_id(round_lo(lo0), lo0);
    piece = load_piece(id_cur);
  }

#ifdef GSAJ_BLOCK_TRACE
  unsigned ph_[7] = {0u, 0u, 0u, 0u, 0u, 0u, 0u};
  unsigned tr_flushes = 0u, tr_slots = 0u;  // phase-2 batches of this wave, and the accepted entries in them
  unsigned long long pht_ = wall_clock64();
#endif
  while (hi > range.x) {
    BWD_PH(6)
    const uint32_t lo = round_lo(hi);
    const int n = (int)(hi - lo);
    // this round's pieces -> LDS.  Layout per entry: rec[3j] = {mean x, mean y, rect, first local slot}, rec[3j+1] = {conic
    // PRE-SCALED for v_exp_f32 (gsaj_prescale_conic), opacity}, rec[3j+2] = {r, g, b, depth}; blk_first[j] = block offset
    if (e < n) {
      int e_ = e;  // (LDS addresses below are formed here, from the lane id, not carried through the loop in registers of their own)
      asm volatile("" : "+v"(e_));
      const int e = e_;
      if (part == 3) {
        blk_first[e] = __float_as_uint(piece.x);
      } else if (!rec16) {
        if (part == 1) {
          const float3 kq = gsaj_prescale_conic(piece.x, piece.y, piece.z);
          rec[e * REC_F4 + 1] = make_float4(kq.x, kq.y, kq.z, piece.w);
        } else {
          rec[e * REC_F4 + part] = piece;
        }
      } else {
        float *r = reinterpret_cast<float *>(rec + e * REC_F4);
        if (part == 0) {  // {mean x, mean y, depth, first}
          r[0] = piece.x, r[1] = piece.y, r[3] = piece.w, r[11] = piece.z;
        } else if (part == 1) {  // half2(a, b), half2(c, opacity), half2(r, g), half2(b, 0)
          const float2 ab = gsaj_unpack_h2(__float_as_uint(piece.x)), co = gsaj_unpack_h2(__float_as_uint(piece.y));
          const float2 rg = gsaj_unpack_h2(__float_as_uint(piece.z)), bz = gsaj_unpack_h2(__float_as_uint(piece.w));
          const float3 kq = gsaj_prescale_conic(ab.x, ab.y, co.x);
          rec[e * REC_F4 + 1] = make_float4(kq.x, kq.y, kq.z, co.y);
          r[8] = rg.x, r[9] = rg.y, r[10] = bz.x;
        } else if (part == 2) {  // the rectangle, packed as the fp32 rows carry it: x0 | y0 << 10 | w << 20
          const uint32_t sx = __float_as_uint(piece.x), sy = __float_as_uint(piece.y);
          const uint32_t x0 = sx & 0xffffu, y0 = sy & 0xffffu, w = (sx >> 16) - x0;
          r[2] = __uint_as_float(x0 | (y0 << 10) | (w << 20));
        }
      }
    }
    {  // the next round's pieces, and the ids of the round after it (unconditional: past the list's head the rounds are empty)
      const uint32_t lo2 = round_lo(lo);
      id_cur = id_nxt;
      id_nxt = load_id(round_lo(lo2), lo2);
      piece = load_piece(id_cur);
    }
    {
      float4 *z = reinterpret_cast<float4 *>(acc);
      for (int i = tid; i < 4 * ACC_C * ACC_STRIDE / 4; i += 256) z[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    BWD_PH(0)
    __syncthreads();
    BWD_PH(1)

    const uint32_t first_idx = (uint32_t)__builtin_amdgcn_readfirstlane((int)(lo - range.x));  // list index (0-based) of rec[0]; workgroup-uniform -> scalar
    if (wmax > first_idx) {
      // lane l tests entry l against this wave's quadrant (and its furthest last contributor)
      bool rel = false;
      if (lane < n && first_idx + (uint32_t)lane < wmax) {
        const float4 q0 = rec[lane * REC_F4 + 0];
        const float4 q1 = rec[lane * REC_F4 + 1];
        // (conic back from its pre-scaled form: a = -2/log2e kx, b = -1/log2e ky, c = -2/log2e kz)
        rel = quadrant_relevant(q0.x, q0.y, (-2.0f / GSAJ_LOG2E) * q1.x, (-1.0f / GSAJ_LOG2E) * q1.y, (-2.0f / GSAJ_LOG2E) * q1.z, q1.w, qx0, qy0);
      }
      unsigned long long todo = __builtin_amdgcn_ballot_w64(rel);
      int nslot = 0;      // accepted entries waiting in wu[] (wave-uniform)
      unsigned long long slot_pack = 0ull;  // byte s: round-local index j of the entry in slot s (wave-uniform: scalar registers)

      // ---- phase 2: moments of the queued (w, u) rows -> per-entry totals in this wave's acc slots ----
      auto flush = [&]() {
#ifdef GSAJ_BLOCK_TRACE
        tr_flushes++;
        tr_slots += (unsigned)nslot;
#endif
        // Lane (slot s, pixel row y): three running sums over the row's 8 pixels -- w, w dx, w dx^2 with every pixel's OWN
        // dx = mean x - pixel x -- give every second moment of the row about the Gaussian's mean (dy is constant along the
        // row).  (Moments about the quadrant's centre column -- sums of w x', w x'^2 with compile-time x', shifted to the mean
        // afterwards -- are two operations per pixel cheaper, but the shift cancels: a Gaussian whose only contributing pixel
        // sits 0.01 px from its mean had sum w dx^2 wrong in the third digit -- 12 eps sum|w| against a sum of 1.5e-4 |w|.)
        // Slots beyond nslot hold stale rows; slots never mix, and only live ones are stored.
        const int j = (int)((slot_pack >> (8 * p2_slot)) & 0xffull);
        const float4 e0 = rec[j * REC_F4 + 0];
        float4 e1 = rec[j * REC_F4 + 1];
        e1.x *= -2.0f / GSAJ_LOG2E; e1.y *= -1.0f / GSAJ_LOG2E; e1.z *= -2.0f / GSAJ_LOG2E;  // pre-scaled conic -> (a, b, c)
        const float ax0 = e0.x - qx0, dy = e0.y - p2_py;  // (mean x relative to the row's first pixel: exact, or as well conditioned as dx itself)
        float m0 = 0.f, swdx = 0.f, swdx2 = 0.f, u0 = 0.f, u1 = 0.f, u2 = 0.f, u3 = 0.f;  // sum w, sum w dx, sum w dx^2
#pragma unroll
        for (int it = 0; it < 8; it++) {
          const int pix = p2_row * 8 + it;
          const float2 q = wu[p2_slot * WU_STRIDE + pix];
          const float4 sd = seed[pix];
          const float dx = ax0 - (float)it;  // the pixel's own dx, as phase 1 and the reference form it (backward.cu:736)
          const float wd = q.x * dx;
          m0 += q.x;
          swdx += wd;
          swdx2 = __builtin_fmaf(wd, dx, swdx2);
          u0 += q.y * sd.x;
          u1 += q.y * sd.y;
          u2 += q.y * sd.z;
          u3 += q.y * sd.w;
        }
        const float swdy = dy * m0;                    // sum w dy
        // the 10 partials are linear in the moments: convert per pixel row, then combine the 8 rows
        float v[10];
        v[0] = -(e1.x * swdx + e1.y * swdy) * ddelx_dx;  // dL/dmean2D.x  (dG/ddx = -G (a dx + b dy))
        v[1] = -(e1.z * swdy + e1.y * swdx) * ddely_dy;  // dL/dmean2D.y
        v[2] = -0.5f * swdx2;                            // dL/dconic a
        v[3] = -0.5f * (dy * swdx);                      // dL/dconic b
        v[4] = -0.5f * (dy * swdy);                      // dL/dconic c
        v[5] = m0 * __builtin_amdgcn_rcpf(e1.w);         // dL/dopacity = sum G dL/dalpha = sum w / o
        v[6] = u0; v[7] = u1; v[8] = u2; v[9] = u3;      // dL/dcolour, dL/ddepth
        const float w0 = merge32(v[0], v[1]), w1 = merge32(v[2], v[3]), w2 = merge32(v[4], v[5]), w3 = merge32(v[6], v[7]),
                    w4 = merge32(v[8], v[9]);
        float x0 = merge16(w0, w1), x1 = merge16(w2, w3), x2 = merge16(w4, w4);
        x0 = dpp_add<0x128>(x0);  // row_ror:8: the two pixel rows that share a 16-lane row
        x1 = dpp_add<0x128>(x1);
        x2 = dpp_add<0x128>(x2);
        // 16-lane row r = lane>>4 now holds, for slot lane&7: x0 -> v[{0,2,1,3}[r]], x1 -> v[{4,6,5,7}[r]], x2 -> v[{8,8,9,9}[r]]
        if (p2_slot < nslot && (lane & 8) == 0) {
          const int r = lane >> 4;
          const int k = ((r & 1) << 1) | (r >> 1);
          float *a = acc + wave * ACC_C * ACC_STRIDE + j;
          a[k * ACC_STRIDE] = x0;
          a[(4 + k) * ACC_STRIDE] = x1;
          if ((r & 1) == 0) a[(8 + (r >> 1)) * ACC_STRIDE] = x2;
        }
        nslot = 0;
        slot_pack = 0ull;
      };

      // ---- phase 1: back-to-front walk over the entries this quadrant can see ----
      if (todo != 0ull) {
        int jj = 63 - __builtin_clzll(todo);
        todo &= ~(1ull << jj);
        // the row of the entry being worked on: q01 = (mean x, mean y), k = (pre-scaled conic, opacity), c = (r, g, b, depth).  The NEXT
        // entry's row is requested into the SAME registers as soon as the current values have had their last use -- (mean, conic)
        // after the power, the colour after c . g -- so no copies rotate a pipeline, and the LDS round trip still hides behind the
        // rest of the entry's arithmetic
        float2 q01 = reinterpret_cast<const float2 *>(rec)[jj * (REC_F4 * 2)];
        float4 k = rec[jj * REC_F4 + 1], c = rec[jj * REC_F4 + 2];
        while (true) {
          const int j = jj;
          const uint32_t idx = first_idx + (uint32_t)j;
          const bool more = todo != 0ull;
          if (more) {
            jj = 63 - __builtin_clzll(todo);
            todo &= ~(1ull << jj);
          }
          const float dx = q01.x - pxf, dy = q01.y - pyf;
          // the forward's own expression (gsaj_common.h): both passes decide power <= 0 / alpha >= 1/255 on identical bits
          const float p2 = gsaj_power2(dx, dy, k.x, k.y, k.z);
          const float oG = k.w * __builtin_amdgcn_exp2f(p2);  // opacity x G
          // (unconditional: after the last entry the same row is read once more -- a load under `if (more)` would merge with the old
          // value through register copies)
          q01 = reinterpret_cast<const float2 *>(rec)[jj * (REC_F4 * 2)];
          k = rec[jj * REC_F4 + 1];
          // alpha = min(0.99, o G) >= 1/255  <=>  o G >= 1/255: the clamp is applied after the mask (one select fewer); the
          // decision is the forward's bit for bit
          const bool valid = idx < last && p2 <= 0.0f && oG >= (1.0f / 255.0f);
          // (a chain of plain fmas: paired into v_pk_mul_f32 + adds -- what the compiler makes of the sum of products -- it costs more
          // cycles, a packed operation being worth 1.2 plain ones here, and needs its operands moved into register pairs)
          float cg = __builtin_fmaf(c.x, gC0, __builtin_fmaf(c.y, gC1, __builtin_fmaf(c.z, gC2, c.w * gD)));
          asm volatile("" : "+v"(cg));  // (c . g is formed HERE, not sunk into the branch below: the colour registers are free for the next row)
          c = rec[jj * REC_F4 + 2];
          if (__builtin_amdgcn_ballot_w64(valid) != 0ull) {
            // a lane that skips this entry runs the same arithmetic with alpha = 0: T and the recurrence come out unchanged
            // and (w, u) = 0
            const float oGm = valid ? oG : 0.f;
            float alpha;  // min(0.99, o G); written as the instruction because fminf() on a selected value costs a second one
                          // (the compiler quiets a possible signalling NaN first: v_max_f32 x, x)
            asm("v_min_f32 %0, 0x3f7d70a4, %1" : "=v"(alpha) : "v"(oGm));
            const float inv1ma = __builtin_amdgcn_rcpf(1.f - alpha);
            T = T * inv1ma;  // T <- T / (1 - alpha)
            asm volatile("" : "+v"(T));  // (on its own: not paired with Tf_bg * inv1ma, which would cost two register moves)
            // dL/dalpha needs accum_rec only through its product with this pixel's seeds, sum_ch (c_ch - accum_rec_ch) g_ch
            // (backward.cu:799-823, colour and depth alike): the four recurrences accum_rec <- alpha c + (1 - alpha) accum_rec
            // collapse into ONE for the scalar s = accum_rec . g, s <- s + alpha (c . g - s) -- 6 VALU operations per entry
            // instead of 12, same value up to fp32 rounding (the dot product is linear in accum_rec)
            const float dd = cg - accS;
            const float dL_dalpha = dd * T - Tf_bg * inv1ma;
            accS += alpha * dd;
            // w = dL/dG * G = (o dL/dalpha) G (backward.cu:826-829; G, not the clamped alpha): o G is at hand from the alpha test
            float w_ = dL_dalpha * oGm;
            asm volatile("" : "+v"(w_));  // (two plain multiplies into a register pair, not v_pk_mul_f32 of operands moved into pairs)
            wu[nslot * WU_STRIDE + lane] = make_float2(w_, alpha * T);
            slot_pack |= (unsigned long long)j << (8 * nslot);
            nslot++;
            if (nslot == SLOTS) flush();
          }
          if (!more) break;
        }
        if (nslot > 0) flush();
      }
    }
    BWD_PH(2)
    __syncthreads();
    BWD_PH(3)
    if (part < 3 && e < n) {  // wave `part`, lane e: float4 `part` of entry e's row = partials 4 part .. 4 part + 3
      float t[4];
#pragma unroll
      for (int c = 0; c < 4; c++) {
        const float *a = acc + min(part * 4 + c, ACC_C - 1) * ACC_STRIDE + e;
        t[c] = ((a[0] + a[ACC_C * ACC_STRIDE]) + a[2 * ACC_C * ACC_STRIDE]) + a[3 * ACC_C * ACC_STRIDE];  // fixed order
      }
      if (part == 2) t[2] = t[3] = 0.f;
      // emission slot: where this instance's partial gradients go (Gaussian-major, tiles of the rectangle in row order)
      const float4 q0 = rec[e * REC_F4 + 0];
      const uint32_t rp = __float_as_uint(q0.z);
      const uint32_t x0 = rp & 1023u, y0 = (rp >> 10) & 1023u, w = rp >> 20;
      const uint32_t emit = blk_first[e] + __float_as_uint(q0.w) + ((uint32_t)ty - y0) * w + ((uint32_t)tx - x0);
      inst_grad[(size_t)emit * REC_F4 + part] = make_float4(t[0], t[1], t[2], t[3]);
      if (part == 0) reached[emit] = 1;
    }
    BWD_PH(4)
    __syncthreads();
    BWD_PH(5)
    hi = lo;
  }
#ifdef GSAJ_BLOCK_TRACE
  {
    const unsigned tw_ = (blockIdx.y * gridDim.x + blockIdx.x) * 4 + (threadIdx.x >> 6);
    if ((threadIdx.x & 63) == 0 && tw_ < GSAJ_TRACE_MAX) {
      g_trace_bwdph[4 * tw_ + 0] = ((unsigned long long)ph_[0] << 32) | ph_[1];
      g_trace_bwdph[4 * tw_ + 1] = ((unsigned long long)ph_[2] << 32) | ph_[3];
      g_trace_bwdph[4 * tw_ + 2] = ((unsigned long long)ph_[4] << 32) | ph_[5];
      g_trace_bwdph[4 * tw_ + 3] = ph_[6] | ((unsigned long long)tr_flushes << 32) | ((unsigned long long)tr_slots << 44);
    }
  }
#endif
  GSAJ_TRACE_END(bwd)
}

int launch_render_backward(int R, int W, int H, int grid_x, int grid_y, const float *bg, const GeomWS &g, const BinWS &b,
                           const ImageWS &im, const float *dL_dpix, const float *dL_dpix_depth, int views, ViewStrides vs,
                           hipStream_t s, const FusedLoss *fl) {
  if (R <= 0) return GSAJ_OK;  // (async callers pass the arena capacity as R)
  {
    GsajProfScope ps(ST_RENDER_BWD, s);
    if (fl)
      hipLaunchKernelGGL(k_render_bwd<true>, dim3(grid_x * grid_y, 1), dim3(256), 0, s, W, H, grid_x, im, b.point_list, g, bg, dL_dpix,
                         dL_dpix_depth, b.inst_grad, b.reached, vs, *fl);
    else
      hipLaunchKernelGGL(k_render_bwd<false>, dim3(grid_x * grid_y, views), dim3(256), 0, s, W, H, grid_x, im, b.point_list, g, bg, dL_dpix,
                         dL_dpix_depth, b.inst_grad, b.reached, vs, FusedLoss{});
  }
  GSAJ_HIP_CHECK(hipGetLastError());
  return GSAJ_OK;
}
