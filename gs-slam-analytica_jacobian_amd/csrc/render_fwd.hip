// render_fwd.hip -- per-tile front-to-back alpha compositing of colour + depth (gfx950).
//
// Semantics: forward.cu:406-535 (renderCUDA): power > 0 skipped, alpha = min(0.99, o*exp(power)),
// alpha < 1/255 skipped, stop before T(1-alpha) < 1e-4, n_touched counts T(1-alpha) > 0.5.
//
// MI355X mapping: one 256-thread workgroup per 16x16 tile = four wave64s, each owning an 8x8 pixel quadrant (one
// pixel per lane) and walking the tile list on its own -- no workgroup barrier in the loop.  Per 64 sorted list
// entries: lane l takes entry l -- its Gaussian id from point_list (coalesced; requested TWO chunks ahead) and that
// Gaussian's 48-byte row gathered from GeomWS.splat (requested ONE chunk ahead; the rows of a frame live in L2) -- a
// lane-parallel test of the entry against the quadrant box (wave_reduce.h), and the survivors PACKED in list order into
// the wave's LDS area with the conic pre-scaled for v_exp_f32 -- PAIR-INTERLEAVED (x_A x_B y_A y_B | kx_A kx_B ky_A ky_B | ...), so that
// ds_read_b128 delivers the register pairs the packed fp32 instructions take: the compiler paired entries before, but paid 31
// v_mov shuffles per four entries for it (120 -> 96 VALU instructions per four entries, 231 -> 219 us per cfg2 window).  The
// entry loop is straight-line, two pairs per step, the next half-step's pair requested from LDS before this one's arithmetic;
// a pixel that skips an entry runs the same arithmetic with weight 0 instead of branching.  Early-outs are wave-level ballots (a quadrant whose 64 pixels have all saturated
// stops), not the reference's block-wide votes.  n_touched: lane l counts packed entry l, only while some pixel of the
// quadrant still has T > 0.5, and ONE atomic wave-instruction per 64 entries flushes it (the reference issues one
// atomic per pixel per entry, forward.cu:512-514).  Tiles are taken longest list first (ImageWS.tile_order).
#include "gsaj_common.h"
#include "loss_terms.h"
#include "wave_reduce.h"

#define FWD_CHUNK 64  // records staged per wave per trip
#define FWD_PAD 8     // inert records after the packed ones: the pipelined entry loop reads up to 7 slots past the last
#define FWD_PAIR_F4 6  // float4 per staged PAIR of entries (pair-interleaved layout, see composite2)
typedef float v2f __attribute__((ext_vector_type(2)));

// Workgroup shape.  The four quadrant waves of a tile never talk to each other (no barrier, private LDS areas), so each can be
// its own 64-thread workgroup: a 256-thread workgroup holds its CU slots until its SLOWEST wave is done, and the quadrants of a
// tile finish far apart (per-wave trace of a batched window, tools/batch_trace.py: mean wave life 27 us, 4045 of 5120 wave slots
// busy on average).  Wave-sized workgroups give every finished wave's slot back at once.  The quadrants of a tile keep
// blockIdx.x equal mod 8, i.e. on the same XCD, so that the tile's records are fetched into ONE L2.
#ifndef GSAJ_FWD_THREADS
#define GSAJ_FWD_THREADS 64
#endif
#define FWD_WAVES (GSAJ_FWD_THREADS / 64)

GSAJ_TRACE_DEFINE(fwd)

// LOSS: the loss-fused form (gsaj_rasterize_forward_loss): the epilogue also evaluates the tracking / mapping
// L1 terms of its pixels against the ground truth (loss_terms.h) and leaves the workgroup's four sums in fl.partials -- the loss
// value without a pass of its own over the images.
template <bool LOSS>
__global__ __launch_bounds__(GSAJ_FWD_THREADS) __attribute__((amdgpu_waves_per_eu(5, 8))) void k_render_fwd(int W, int H, int gx, int tiles, int P, ImageWS im,
                                                    const float4 *__restrict__ splat, const float4 *__restrict__ splat16,
                                                    const float *__restrict__ bg,
                                                    float *__restrict__ out_color, float *__restrict__ out_depth,
                                                    float *__restrict__ out_opacity, int *__restrict__ n_touched,
                                                    const uint32_t *__restrict__ point_list, ViewStrides vs, FusedLoss fl) {
  {  // batched launch: blockIdx.y = view
    const size_t view = blockIdx.y, HWv = (size_t)H * W;
    im = image_view(im, view * vs.image);
    splat = gsaj_shift(splat, view * vs.geom);
    splat16 = gsaj_shift(splat16, view * vs.geom);
    point_list = gsaj_shift(point_list, view * vs.bin);
    out_color += view * 3 * HWv;
    out_depth += view * HWv;
    out_opacity += view * HWv;
    n_touched += view * (size_t)P;
  }
  const uint2 *__restrict__ ranges = im.ranges;
  float *__restrict__ final_T = im.final_T;
  uint32_t *__restrict__ n_contrib = im.n_contrib;
  uint32_t *__restrict__ counters = im.counters;
  // Each wave (one 8x8 quadrant) walks the tile list on its own: private 64-record staging area, no
  // workgroup barrier anywhere, so a quadrant never waits for a slower neighbour.  The four waves of a
  // tile gather the same rows; the repeats are served by L1/L2.
  __shared__ float4 rec_all[FWD_WAVES * ((FWD_CHUNK + FWD_PAD) / 2) * FWD_PAIR_F4];
  if (counters[4]) return;  // aborted async frame
  GSAJ_TRACE_BEGIN(fwd)
  const int tid = threadIdx.x, lane = tid & 63;
#if FWD_WAVES == 1
  // 32 consecutive workgroups = 8 tiles x 4 quadrants; tile slot = blockIdx.x & 7 (the XCD the workgroup lands on)
  const int wave = ((int)blockIdx.x >> 3) & 3;
  const int rank = ((int)blockIdx.x >> 5) * 8 + ((int)blockIdx.x & 7);
  if (rank >= tiles) {
    if (LOSS && lane == 0) reinterpret_cast<float4 *>(fl.partials)[blockIdx.x] = make_float4(0.f, 0.f, 0.f, 0.f);
    return;
  }
  const int tile = (int)min(im.tile_order[rank], (uint32_t)(tiles - 1));  // longest lists first (frame_scan)
  float4 *rec = rec_all;
#else
  const int wave = tid >> 6;
  const int tile = (int)min(im.tile_order[blockIdx.x], (uint32_t)(tiles - 1));
  float4 *rec = rec_all + wave * ((FWD_CHUNK + FWD_PAD) / 2) * FWD_PAIR_F4;
#endif
  const int ty = tile / gx, tx = tile - ty * gx;
  const int px = tx * TILE + (wave & 1) * 8 + (lane & 7);
  const int py = ty * TILE + (wave >> 1) * 8 + (lane >> 3);
  const bool inside = px < W && py < H;
  const float pxf = (float)px, pyf = (float)py;
  const float qx0 = (float)(tx * TILE + (wave & 1) * 8), qy0 = (float)(ty * TILE + (wave >> 1) * 8);
  const uint2 range = ranges[tile];

  bool done = !inside;
  float T = 1.0f;
  uint32_t last = 0;
  bool counting = true;  // wave-uniform: some pixel of the quadrant still has T > 0.5 (only those can "touch")

  // TWO packed list entries (A, B: consecutive in list order) for this lane's pixel, stored pair-interleaved in LDS so that the
  // arithmetic both entries share -- dx, dy, the power, opacity x G -- runs as packed fp32 instructions (v_pk_add / v_pk_mul /
  // v_pk_fma_f32) on register pairs exactly as ds_read_b128 delivers them (no v_mov shuffles):
  //   p0 = (x_A, x_B, y_A, y_B)   p1 = (kx_A, kx_B, ky_A, ky_B)   p2 = (kz_A, kz_B, o_A, o_B)   cA / cB = (r, g, b, depth)
  //   pos = (1-based list positions of A, B)
  // with k the conic pre-scaled so that kx dx^2 + ky dx dy + kz dy^2 = log2(e) * power.  The sequential part (T, saturation) is
  // scalar, A before B; colour + depth accumulate as two packed pairs.  A pixel that skips an entry runs the same arithmetic
  // with weight 0: T, C, D come out unchanged.  Every operation is the one gsaj_power2 / the reverse compositor use (IEEE
  // mul / fma, packed or not): both passes decide on identical bits.
  v2f Crg = {0.f, 0.f}, Cbd = {0.f, 0.f};
  const v2f px2 = {pxf, pxf}, py2 = {pyf, pyf};
  auto composite2 = [&](const float4 p0, const float4 p1, const float4 p2, const float4 cA, const float4 cB, const float2 pos,
                        float &T_afterA) {
    const v2f dx = v2f{p0.x, p0.y} - px2, dy = v2f{p0.z, p0.w} - py2;
    const v2f t = v2f{p1.z, p1.w} * dy;
    const v2f u = v2f{p2.x, p2.y} * dy;
    const v2f pw = __builtin_elementwise_fma(dx, __builtin_elementwise_fma(v2f{p1.x, p1.y}, dx, t), u * dy);
    const v2f oe = v2f{p2.z, p2.w} * v2f{__builtin_amdgcn_exp2f(pw.x), __builtin_amdgcn_exp2f(pw.y)};
    {
      const float alpha0 = fminf(0.99f, oe.x);
      bool ok = !done && pw.x <= 0.0f && alpha0 >= (1.0f / 255.0f);
      const float tA = __builtin_fmaf(-alpha0, T, T);
      const bool sat = ok && tA < 0.0001f;  // this pixel is saturated: stop before this entry
      done = done || sat;
      ok = ok && !sat;
      const float w = ok ? alpha0 * T : 0.f;
      const v2f w2 = {w, w};
      Crg = __builtin_elementwise_fma(v2f{cA.x, cA.y}, w2, Crg);
      Cbd = __builtin_elementwise_fma(v2f{cA.z, cA.w}, w2, Cbd);
      T = ok ? tA : T;
      last = ok ? __float_as_uint(pos.x) : last;
      T_afterA = T;
    }
    {
      const float alpha0 = fminf(0.99f, oe.y);
      bool ok = !done && pw.y <= 0.0f && alpha0 >= (1.0f / 255.0f);
      const float tB = __builtin_fmaf(-alpha0, T, T);
      const bool sat = ok && tB < 0.0001f;
      done = done || sat;
      ok = ok && !sat;
      const float w = ok ? alpha0 * T : 0.f;
      const v2f w2 = {w, w};
      Crg = __builtin_elementwise_fma(v2f{cB.x, cB.y}, w2, Crg);
      Cbd = __builtin_elementwise_fma(v2f{cB.z, cB.w}, w2, Cbd);
      T = ok ? tB : T;
      last = ok ? __float_as_uint(pos.y) : last;
    }
  };

  if (__builtin_amdgcn_ballot_w64(!done) != 0ull && range.x < range.y) {
    // two loads in a row per entry (id, then the row it names): ids are requested two chunks ahead and rows one chunk ahead,
    // so that neither round trip is waited for behind the other
    float4 q0, q1, q2;
    uint32_t id_cur = 0u, id_nxt = 0u;
    const bool rec16 = counters[7] != 0u;  // fp16-storage rows (gsaj_common.h)
    auto fetch_id = [&](uint32_t base) {
      if (base + (uint32_t)lane < range.y) id_nxt = point_list[base + lane];
    };
    auto fetch_row = [&](uint32_t base) {  // (of the chunk whose ids fetch_id requested last)
      id_cur = id_nxt;
      if (base + (uint32_t)lane < range.y) gsaj_load_row(splat, splat16, id_nxt, rec16, q0, q1, q2);
    };
    float *recf = reinterpret_cast<float *>(rec);
    for (int i = lane; i < ((FWD_CHUNK + FWD_PAD) / 2) * FWD_PAIR_F4; i += 64) rec[i] = make_float4(0.f, 0.f, 0.f, 0.f);  // (see the sentinels below)
    // packed slot s -> pair s / 2, half s % 2 of the pair-interleaved layout (FWD_PAIR_F4 float4 per pair)
    auto put = [&](int s, float x, float y, float kx, float ky, float kz, float o, float4 c, uint32_t pos, uint32_t id) {
      const int h = s & 1, pb = (s >> 1) * FWD_PAIR_F4;
      float *f = recf + pb * 4 + h;
      f[0] = x; f[2] = y; f[4] = kx; f[6] = ky; f[8] = kz; f[10] = o;
      rec[pb + 3 + h] = c;
      f[20] = __uint_as_float(pos); f[22] = __uint_as_float(id);
    };
    fetch_id(range.x);
    fetch_row(range.x);
    fetch_id(range.x + FWD_CHUNK);
    for (uint32_t base = range.x; base < range.y; base += FWD_CHUNK) {
      const int m = min((uint32_t)FWD_CHUNK, range.y - base);
      // stage: lane l holds the row of entry base+l and tests it against the quadrant; the entries the quadrant can see
      // are packed to the front of the wave's LDS area in list order
      bool rel = false;
      const uint32_t id_here = id_cur;
      if (lane < m) rel = quadrant_relevant(q0.x, q0.y, q1.x, q1.y, q1.z, q1.w, qx0, qy0);
      const unsigned long long todo = __builtin_amdgcn_ballot_w64(rel);
      const int nrel = __popcll(todo);
      __builtin_amdgcn_wave_barrier();
      if (rel) {
        const int slot = __builtin_amdgcn_mbcnt_hi((uint32_t)(todo >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)todo, 0u));
        const float3 k = gsaj_prescale_conic(q1.x, q1.y, q1.z);
        put(slot, q0.x, q0.y, k.x, k.y, k.z, q1.w, make_float4(q2.x, q2.y, q2.z, q0.z), base - range.x + (uint32_t)lane + 1u, id_here);
      }
      // inert sentinels behind the packed entries: opacity 0 is enough -- whatever else the slot holds is a finite number (an earlier
      // entry's, or the zeros the area started with), so the entry's power is finite or positive, its alpha 0 or rejected, its weight 0
      if (lane < FWD_PAD) recf[((nrel + lane) >> 1) * FWD_PAIR_F4 * 4 + 10 + ((nrel + lane) & 1)] = 0.f;
      fetch_row(base + FWD_CHUNK);
      fetch_id(base + 2 * FWD_CHUNK);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      // lane l: #pixels of this wave that count packed entry l as "touched".  The four counts of a step (each <= 64) travel as
      // the bytes of ONE scalar word to the four lanes of the step (one compare + select per FOUR entries instead of per entry);
      // every lane takes its byte after the loop
      uint32_t cnt4 = 0u;
      bool wave_done = false;
      // two pairs per step; the pair of the next half-step is requested from LDS before this half-step's arithmetic
      const float2 *rec2 = reinterpret_cast<const float2 *>(rec);
      float4 a0 = rec[0], a1 = rec[1], a2 = rec[2], a3 = rec[3], a4 = rec[4];
      float2 a5 = rec2[10];
      for (int i = 0; i < nrel; i += 4) {
        const int pb = (i >> 1) * FWD_PAIR_F4;
        const float4 c0 = rec[pb + 6], c1 = rec[pb + 7], c2 = rec[pb + 8], c3 = rec[pb + 9], c4 = rec[pb + 10];
        const float2 c5 = rec2[(pb + 11) * 2];
        uint32_t pack = 0u;  // (scalar)
        // "touched" (forward.cu:512-514): the pixel took the entry and is left with T > 0.5.  A pixel that takes an entry leaves
        // with a strictly smaller T (alpha >= 1/255), one that does not keeps its T: "took it" is T_after < T_before -- two
        // compares on registers, and T_after > 0.5 of the second entry is the "anyone still counting" test as well (a ballot of the
        // accept flag itself -- or of any `a && b` -- is first materialised as 0 / 1 and compared again; the AND of two ballots of plain
        // compares is two v_cmp and a scalar AND).
        {
          const float T0 = T;
          float Ta;
          composite2(a0, a1, a2, a3, a4, a5, Ta);
          if (counting) {
            const unsigned long long hB = __builtin_amdgcn_ballot_w64(T > 0.5f);
            const uint32_t nA = (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(Ta > 0.5f) & __builtin_amdgcn_ballot_w64(Ta < T0));
            const uint32_t nB = (uint32_t)__popcll(hB & __builtin_amdgcn_ballot_w64(T < Ta));
            pack = nA | (nB << 8);
            counting = hB != 0ull;
          }
        }
        // (not before this point: requested earlier, while the first pair still occupies these registers, the next pair lands in others
        // and is copied over at the end of the step -- six moves per step)
        asm volatile("" : "+v"(T), "+v"(last), "+v"(Crg), "+v"(Cbd) : : "memory");  // (the first pair's results are formed: its registers are free)
        a0 = rec[pb + 12], a1 = rec[pb + 13], a2 = rec[pb + 14], a3 = rec[pb + 15], a4 = rec[pb + 16];
        a5 = rec2[(pb + 17) * 2];
        {
          const float T0 = T;
          float Tc;
          composite2(c0, c1, c2, c3, c4, c5, Tc);
          if (counting) {
            const unsigned long long hD = __builtin_amdgcn_ballot_w64(T > 0.5f);
            const uint32_t nC = (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(Tc > 0.5f) & __builtin_amdgcn_ballot_w64(Tc < T0));
            const uint32_t nD = (uint32_t)__popcll(hD & __builtin_amdgcn_ballot_w64(T < Tc));
            pack |= (nC << 16) | (nD << 24);
            counting = hD != 0ull;
          }
        }
        cnt4 = ((lane >> 2) == (i >> 2)) ? pack : cnt4;
        if (__builtin_amdgcn_ballot_w64(!done) == 0ull) {
          wave_done = true;
          break;
        }
      }
      const int cnt = (int)((cnt4 >> (8 * (lane & 3))) & 0xffu);
      if (lane < nrel && cnt > 0) {
        const uint32_t id = __float_as_uint(recf[(lane >> 1) * FWD_PAIR_F4 * 4 + 22 + (lane & 1)]);
        atomicAdd(&n_touched[id], cnt);
      }
      __builtin_amdgcn_wave_barrier();
      if (wave_done) break;  // whole quadrant saturated
    }
  }

  const float Cr = Crg.x, Cg = Crg.y, Cb = Cbd.x, Dp = Cbd.y;
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
  if (LOSS) {  // the loss terms of this quadrant's pixels, from the very values just stored
    float s4[4] = {0.f, 0.f, 0.f, 0.f};
    if (inside) {
      const size_t pid = (size_t)py * W + px, HW = (size_t)H * W;
      const LossConsts L = loss_consts(fl.flags, fl.alpha, fl.rgb_thr, fl.exp_a, fl.exp_b, HW, 1.f);
      const bool mask = fl.grad_mask ? fl.grad_mask[pid] != 0 : true;
      const float gd = L.mono ? 0.f : fl.gt_depth[pid];
      const LossPixel o = loss_pixel(L, fl.gt_color[pid], fl.gt_color[HW + pid], fl.gt_color[2 * HW + pid], Cr + T * bg[0], Cg + T * bg[1],
                                     Cb + T * bg[2], 1.f - T, mask, gd, Dp);
      s4[0] = o.s_rgb; s4[1] = o.s_d; s4[2] = o.s_a; s4[3] = o.s_b;
    }
#pragma unroll
    for (int c = 0; c < 4; c++) {
#pragma unroll
      for (int o2 = 32; o2 > 0; o2 >>= 1) s4[c] += __shfl_xor(s4[c], o2);
    }
    if (lane == 0) reinterpret_cast<float4 *>(fl.partials)[blockIdx.x] = make_float4(s4[0], s4[1], s4[2], s4[3]);
  }
  GSAJ_TRACE_END(fwd)
}

int gsaj_fwd_loss_slots(int W, int H) {
  const int tiles = ((W + TILE - 1) / TILE) * ((H + TILE - 1) / TILE);
  return FWD_WAVES == 1 ? ((tiles + 7) / 8) * 32 : tiles;
}

int launch_render_forward(int P, int W, int H, int grid_x, int grid_y, const float *bg, const GeomWS &g, const BinWS &b,
                          const ImageWS &im, float *out_color, float *out_depth, float *out_opacity, int *n_touched, int views,
                          ViewStrides vs, hipStream_t s, const FusedLoss *fl) {
  static_assert(FWD_WAVES == 1, "the loss-fused epilogue writes one partial per wave-sized workgroup");
  {
    GsajProfScope ps(ST_RENDER_FWD, s);
    const int tiles = grid_x * grid_y;
    const unsigned nblk = FWD_WAVES == 1 ? (unsigned)((tiles + 7) / 8) * 32u : (unsigned)tiles;
    if (fl)
      hipLaunchKernelGGL(k_render_fwd<true>, dim3(nblk, 1), dim3(GSAJ_FWD_THREADS), 0, s, W, H, grid_x, tiles, P, im, g.splat, g.splat16, bg,
                         out_color, out_depth, out_opacity, n_touched, b.point_list, vs, *fl);
    else
      hipLaunchKernelGGL(k_render_fwd<false>, dim3(nblk, views), dim3(GSAJ_FWD_THREADS), 0, s, W, H, grid_x, tiles, P, im, g.splat, g.splat16,
                         bg, out_color, out_depth, out_opacity, n_touched, b.point_list, vs, FusedLoss{});
  }
  GSAJ_HIP_CHECK(hipGetLastError());
  return GSAJ_OK;
}
