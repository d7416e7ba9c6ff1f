// binning.hip - instance emission and tile ranges (K3 / K5 of SURVEY.md 2.3), MI355X design.
//
// The published rasterizer sorts R 64-bit (tile<<32 | depth) keys in one global radix sort (6 passes over
// R pairs).  Here the same total order (tile, depth bits, Gaussian id) is produced with far less traffic:
//   1. Gaussians are depth-sorted ONCE (32-bit keys, P elements)            [api.hip -> gsr_radix_sort_pairs]
//   2. instances are emitted in that depth order                             [k_emit_instances]
//   3. a STABLE sort on the tile id alone (<= 15 bits at 4K: 2 passes)       [gsr_radix_sort_pairs]
// Stability of step 3 keeps (depth, id) order inside each tile, so the per-tile lists are identical to the
// published ordering (ties on equal depth bits broken by ascending Gaussian id because step 1 is stable on
// an id-ordered input).
#include "gsr_common.h"

// One thread per depth-sorted position j.  The 256 Gaussians of a workgroup are consecutive in depth order, so their
// emission slots form ONE contiguous range [slot0, slot0 + count): the (tile, Gaussian) pairs are assembled in LDS and then
// written with coalesced stores (per-thread 4-B stores at scattered addresses ran at ~0.5 TB/s).  Ranges that do not fit
// the LDS window (a few huge splats) are written directly.
#ifndef EMIT_WINDOW
#define EMIT_WINDOW 4096
#endif
__global__ __launch_bounds__(256) void k_emit_instances(int P, int grid_x, const uint32_t* __restrict__ order,
                                                        const uint32_t* __restrict__ offsets_incl,
                                                        const float4* __restrict__ bin_rec,
                                                        uint32_t* __restrict__ tile_key,
                                                        uint32_t* __restrict__ gauss_of_slot,
                                                        uint32_t* __restrict__ slot_start, int tiles,
                                                        uint2* __restrict__ ranges, uint32_t cap,
                                                        uint32_t* __restrict__ sort_head) {
  __shared__ uint32_t lkey[EMIT_WINDOW], lgid[EMIT_WINDOW];
  // the tile sort that follows wants its digit histograms and pass tickets zeroed (sort_scan.hip, head_zeroed)
  if (blockIdx.x == 0)
    for (int i = threadIdx.x; i < GSR_RADIX_HEAD_WORDS; i += 256) sort_head[i] = 0u;
  // the tile ranges are filled in after the tile sort (k_finalize_bins); tiles without instances keep this (0, 0)
  for (int t = blockIdx.x * 256 + threadIdx.x; t < tiles; t += gridDim.x * 256) ranges[t] = make_uint2(0u, 0u);
  const int j0 = blockIdx.x * 256;
  const int j = j0 + threadIdx.x;
  const int jlast = min(j0 + 255, P - 1);
  const uint32_t slot0 = (j0 == 0) ? 0u : offsets_incl[j0 - 1];
  const uint32_t count = offsets_incl[jlast] - slot0;            // block-uniform
  const bool staged = count <= EMIT_WINDOW;
  if (j < P) {
    const uint32_t g = order[j];
    const uint32_t incl = offsets_incl[j];
    const uint32_t n = incl - (j == 0 ? 0u : offsets_incl[j - 1]);   // = tiles_touched[g], without a gather
    if (n != 0) {  // culled Gaussians sort to the end (key 0xFFFFFFFF) and emit nothing
      uint32_t off = incl - n;
      slot_start[g] = off;
      // same inputs (bit copies of the stored record fields) and the same compiled row-interval routine as
      // k_preprocess_fwd -> exactly n tiles
      const float4 r0 = bin_rec[2 * (size_t)g], r1 = bin_rec[2 * (size_t)g + 1];
      const uint32_t rlo = __float_as_uint(r1.z), rhi = __float_as_uint(r1.w);
      const ushort4 r = make_ushort4((unsigned short)(rlo & 0xFFFFu), (unsigned short)(rlo >> 16),
                                     (unsigned short)(rhi & 0xFFFFu), (unsigned short)(rhi >> 16));
      const uint32_t end = off + n;
      uint32_t* kdst = staged ? lkey : tile_key;
      uint32_t* gdst = staged ? lgid : gauss_of_slot;
      const uint32_t bias = staged ? slot0 : 0u;
      for (int y = r.y; y < r.w; y++) {
        const uint32_t iv = gsr_row_interval(r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, y, r.x, r.z);
        const int lo = (int)(iv & 0xFFFFu), hi = (int)(iv >> 16);
        for (int x = lo; x < hi && off < end; x++) {
          if (staged || off < cap) {   // (direct path) slots beyond the binning state's capacity are dropped, see below
            kdst[off - bias] = (uint32_t)(y * grid_x + x);
            gdst[off - bias] = g;
          }
          off++;
        }
      }
      // belt and braces: if fewer tiles passed than were counted (cannot happen with one compiled test body), park the
      // unused slots on this Gaussian's first tile with a sentinel Gaussian id that the render kernels treat as empty
      for (; off < end; off++) {
        if (staged || off < cap) {
          kdst[off - bias] = (uint32_t)(r.y * grid_x + r.x);
          gdst[off - bias] = 0xFFFFFFFFu;
        }
      }
    }
  }
  if (staged) {
    __syncthreads();
    // `cap` = instances the binning state has room for.  On the blocking path cap == num_rendered and the guard never
    // fires; on the non-blocking path (gsr_forward_async) a view with more instances than the caller's estimate loses its
    // LAST slots, i.e. (emission runs in depth order) its farthest splats, instead of writing out of bounds.
    const uint32_t lim = slot0 < cap ? min(count, cap - slot0) : 0u;
    for (uint32_t i = threadIdx.x; i < lim; i += 256) {
      tile_key[slot0 + i] = lkey[i];
      gauss_of_slot[slot0 + i] = lgid[i];
    }
  }
}

// one thread per sorted instance position: tile ranges (the Gaussian id list comes out of the tile sort itself)
__global__ __launch_bounds__(256) void k_finalize_bins(uint32_t cap, const uint32_t* __restrict__ n_dev,
                                                       const uint32_t* __restrict__ tile_sorted,
                                                       uint2* __restrict__ ranges) {
  const uint32_t R = gsr_eff_n(n_dev, cap);
  const uint32_t pos = blockIdx.x * 256 + threadIdx.x;
  if (pos >= R) return;
  const uint32_t t = tile_sorted[pos];
  if (pos == 0) {
    ranges[t].x = 0;
  } else {
    const uint32_t prev = tile_sorted[pos - 1];
    if (prev != t) {
      ranges[prev].y = pos;
      ranges[t].x = pos;
    }
  }
  if (pos == R - 1) ranges[t].y = R;
}

void gsr_launch_emit(int P, int grid_x, int tiles, const char* geom, const GsrGeomLayout& GL, char* bin,
                     const GsrBinLayout& BL, uint32_t cap, hipStream_t st) {
  GSR_LAUNCH("emit_instances", k_emit_instances, dim3((P + 255) / 256), dim3(256), 0, st, P, grid_x,
             (const uint32_t*)(geom + GL.order), (const uint32_t*)(geom + GL.offsets),
             (const float4*)(geom + GL.bin_rec), (uint32_t*)(bin + BL.key_a), (uint32_t*)(bin + BL.gauss_of_slot), (uint32_t*)(geom + GL.slot_start), tiles,
             (uint2*)(bin + BL.ranges), cap, (uint32_t*)(bin + BL.radix_tmp));
}

void gsr_launch_finalize(uint32_t cap, const uint32_t* n_dev, const uint32_t* tile_sorted, char* bin,
                         const GsrBinLayout& BL, hipStream_t st) {
  GSR_LAUNCH("finalize_bins", k_finalize_bins, dim3((cap + 255) / 256), dim3(256), 0, st, cap, n_dev, tile_sorted,
             (uint2*)(bin + BL.ranges));
}
