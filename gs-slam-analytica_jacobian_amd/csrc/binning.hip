// binning.hip -- (tile, depth) sort of the instance keys, per-tile ranges and the
// per-instance record gather (gfx950).
//
// Reference behaviour: rasterizer_impl.cu:353-358 (cub::DeviceRadixSort::SortPairs on
// 64-bit keys, bits [0, 32 + ceil(log2 tiles))), :116-138 (identifyTileRanges).  The
// LSD radix sort is stable, so equal (tile, depth) keys keep emission order = Gaussian
// index order; the CPU oracle sorts by (key, emission index) and the two agree bit for bit.
//
// Sort: rocPRIM's device radix sort (the ROCm counterpart of the CUB call the reference
// makes) -- scaffolding for round 1; ranges + record gather are hand-written and fused.
#include <cstring>
#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>

#include "gsaj_common.h"

size_t gsaj_sort_temp_bytes(int R) {
  size_t bytes = 0;
  uint64_t *k = nullptr;
  uint32_t *v = nullptr;
  size_t n = R > 0 ? (size_t)R : 1;
  (void)rocprim::radix_sort_pairs(nullptr, bytes, k, k, v, v, n, 0, 64, (hipStream_t)0);
  return bytes + 256;
}

int launch_sort(int R, int end_bit, const BinWS &b, hipStream_t s) {
  if (R <= 0) return GSAJ_OK;
  size_t bytes = b.sort_temp_bytes;
  GsajProfScope ps(ST_SORT, s);
  GSAJ_HIP_CHECK(rocprim::radix_sort_pairs(b.sort_temp, bytes, b.keys_unsorted, b.keys, b.vals_unsorted, b.point_list,
                                           (size_t)R, 0u, (unsigned)end_bit, s));
  return GSAJ_OK;
}

// One lane per sorted instance k:
//   * tile boundaries -> ranges[tile] = [start, end)
//   * record k = {mean2D, depth, id | conic, opacity | colour}: 48 contiguous bytes that the
//     compositors stream instead of chasing point_list -> 4 arrays per entry
//   * the record carries the instance's emission slot, where the reverse compositor stores its
//     partial gradients (the per-Gaussian backward then reads one contiguous run, fixed order).
__global__ __launch_bounds__(256) void k_ranges_records(int R, int gx, int gy, const uint64_t *__restrict__ keys,
                                                        const uint32_t *__restrict__ point_list,
                                                        const int *__restrict__ radii, const float *__restrict__ features,
                                                        GeomWS g, const uint32_t *__restrict__ sticky,
                                                        uint2 *__restrict__ ranges, float4 *__restrict__ records) {
  const int k = blockIdx.x * 256 + threadIdx.x;
  if (k >= R) return;
  const uint64_t key = keys[k];
  const uint32_t tile = (uint32_t)(key >> 32);
  if (k == 0) {
    ranges[tile].x = 0;
  } else {
    const uint32_t prev = (uint32_t)(keys[k - 1] >> 32);
    if (tile != prev) {
      ranges[prev].y = (uint32_t)k;
      ranges[tile].x = (uint32_t)k;
    }
  }
  if (k == R - 1) ranges[tile].y = (uint32_t)R;

  const uint32_t id = point_list[k];
  const float2 xy = g.means2D[id];
  const float4 co = g.conic_opacity[id];
  const float depth = g.depths[id];
  records[(size_t)k * REC_F4 + 0] = make_float4(xy.x, xy.y, depth, __uint_as_float(id));
  records[(size_t)k * REC_F4 + 1] = co;
  int x0, y0, x1, y1;
  tile_rect(xy.x, xy.y, radii[id], gx, gy, gsaj_tile_band(sticky), x0, y0, x1, y1);
  const int ty = (int)tile / gx, tx = (int)tile - ty * gx;
  const uint32_t u = g.point_offsets[id] - g.tiles_touched[id] + (uint32_t)((ty - y0) * (x1 - x0) + (tx - x0));
  records[(size_t)k * REC_F4 + 2] = make_float4(features[3 * (size_t)id], features[3 * (size_t)id + 1],
                                                features[3 * (size_t)id + 2], __uint_as_float(u));
}

int launch_ranges_and_records(int P, int R, int grid_x, int grid_y, const int *radii, const float *features,
                              const GeomWS &g, const BinWS &b, const ImageWS &im, hipStream_t s) {
  (void)P;
  GSAJ_HIP_CHECK(hipMemsetAsync(im.ranges, 0, sizeof(uint2) * (size_t)grid_x * grid_y, s));
  if (R > 0) {
    GsajProfScope ps(ST_RANGES_RECORDS, s);
    hipLaunchKernelGGL(k_ranges_records, dim3((R + 255) / 256), dim3(256), 0, s, R, grid_x, grid_y, b.keys, b.point_list,
                       radii, features, g, im.sticky, im.ranges, b.records);
    GSAJ_HIP_CHECK(hipGetLastError());
  }
  return GSAJ_OK;
}
