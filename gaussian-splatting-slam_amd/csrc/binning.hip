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
//
// order == nullptr (tile-local binning form): emission in index order; offsets_incl then holds the instance total of every
// projection workgroup (= workgroup here; bit 31 = "a prefiltered point of it was culled") and the WHOLE prefix sum happens
// here: a workgroup adds up the totals in front of its own (<= P / 256 words, L2-resident) and finishes the scan from
// tiles_touched; workgroup 0 also adds up all of them and leaves the frame's status words in `meta` ([1] cull flag, [2..3]
// num_rendered, the rest 0) and, for a host that waits for the count, in the pinned word `early` (api.hip wait_for_count).
// One launch less than a scan kernel in between, and nothing in the geometry state has to be cleared beforehand.
__global__ __launch_bounds__(256) void k_emit_instances(int P, int grid_x, const uint32_t* __restrict__ order,
                                                        const uint32_t* __restrict__ offsets_incl,
                                                        const uint32_t* __restrict__ tiles_touched,
                                                        const float4* __restrict__ bin_rec,
                                                        uint32_t* __restrict__ tile_key,
                                                        uint32_t* __restrict__ gauss_of_slot,
                                                        uint32_t* __restrict__ slot_start, int tiles,
                                                        uint2* __restrict__ ranges, uint32_t cap,
                                                        uint32_t* __restrict__ sort_head, uint32_t* __restrict__ meta,
                                                        unsigned long long* __restrict__ early, int hist_bits,
                                                        size_t lookback_words,
                                                        const uint32_t* __restrict__ tile_cutoff /* nullptr: emit everything */,
                                                        const uint32_t* __restrict__ depth_key) {
  __shared__ uint32_t lkey[EMIT_WINDOW], lgid[EMIT_WINDOW];
  __shared__ uint32_t wave_tot[4];
  __shared__ unsigned long long wave_pre[4];
  __shared__ uint32_t lhist[2 * GSR_RADIX_SIZE];    // (hist_bits) digit counts of this workgroup's instances, both tile-sort passes
  // hist_bits > 0 (round 4, tile-local form): this kernel writes every tile id anyway, so it also counts the tile sort's digit
  // histograms (LDS, then one global add per non-zero counter into the replica blockIdx picks - gsr_common.h) and clears the
  // sort's look-back table: no k_radix_hist_all launch.  The head (histogram replicas + tickets) was zeroed by the PROJECTION
  // kernel, i.e. before any workgroup of this one can add to it.  hist_bits == 0: the head is zeroed here, for k_radix_hist_all.
  const int hpasses = hist_bits > 0 ? gsr_radix_passes(hist_bits) : 0;     // (<= 2: tile ids have at most 16 bits)
  if (hist_bits > 0) {
    for (int i = threadIdx.x; i < 2 * GSR_RADIX_SIZE; i += 256) lhist[i] = 0u;
    uint4* z = reinterpret_cast<uint4*>(sort_head + GSR_RADIX_HEAD_WORDS);      // look-back words of the sort's first pass
    const size_t n4 = lookback_words >> 2;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) z[i] = make_uint4(0u, 0u, 0u, 0u);
    __syncthreads();
  } else if (blockIdx.x == 0) {
    // the tile sort that follows wants its digit histograms and pass tickets zeroed (sort_scan.hip, head_zeroed)
    for (int i = threadIdx.x; i < GSR_RADIX_HEAD_WORDS; i += 256) sort_head[i] = 0u;
  }
  auto count_digits = [&](uint32_t tile_id) __attribute__((always_inline)) {
#pragma unroll
    for (int p = 0; p < 2; p++)
      if (p < hpasses)
        atomicAdd(&lhist[p * GSR_RADIX_SIZE + ((tile_id >> gsr_radix_shift(hist_bits, p)) & ((1u << gsr_radix_width(hist_bits, p)) - 1u))], 1u);
  };
  // the tile ranges are filled in after the tile sort (k_finalize_bins writes them / the sort's last pass takes maxima into them
  // and k_tile_depth_sort decodes): tiles without instances keep this (0, 0)
  for (int t = blockIdx.x * 256 + threadIdx.x; t < tiles; t += gridDim.x * 256) ranges[t] = make_uint2(0u, 0u);
  const int j0 = blockIdx.x * 256;
  const int j = j0 + threadIdx.x;
  const int jlast = min(j0 + 255, P - 1);
  uint32_t slot0, count, incl = 0, n = 0;                          // slot0, count: block-uniform
  if (order) {
    slot0 = (j0 == 0) ? 0u : offsets_incl[j0 - 1];
    count = offsets_incl[jlast] - slot0;
    if (j < P) {
      incl = offsets_incl[j];
      n = incl - (j == 0 ? 0u : offsets_incl[j - 1]);              // = tiles_touched[g], without a gather
    }
  } else {
    const int nb = (P + 255) / 256;
    count = offsets_incl[blockIdx.x] & ~GSR_BLOCK_CULLED;
    {
      // instances of the workgroups in front of this one (workgroup 0: of all of them = num_rendered, and the cull flags)
      const int upto = blockIdx.x == 0 ? nb : (int)blockIdx.x;
      unsigned long long part = 0ull;
      uint32_t flag = 0u;
      // (16-byte loads, all of a thread's in flight together: <= 4 per thread at 1 M Gaussians; the array is 256-B aligned)
      const uint4* v4 = reinterpret_cast<const uint4*>(offsets_incl);
      const int upto4 = upto >> 2;
#pragma unroll 4
      for (int i = threadIdx.x; i < upto4; i += 256) {
        const uint4 q = v4[i];
        part += (unsigned long long)(q.x & ~GSR_BLOCK_CULLED) + (q.y & ~GSR_BLOCK_CULLED) + (q.z & ~GSR_BLOCK_CULLED) +
                (q.w & ~GSR_BLOCK_CULLED);
        flag |= (q.x | q.y) | (q.z | q.w);
      }
      if ((int)threadIdx.x < (upto & 3)) {
        const uint32_t v = offsets_incl[4 * upto4 + threadIdx.x];
        part += v & ~GSR_BLOCK_CULLED;
        flag |= v;
      }
#pragma unroll
      for (int d = 32; d > 0; d >>= 1) part += __shfl_down(part, d, 64);
      if ((threadIdx.x & 63) == 0) wave_pre[threadIdx.x >> 6] = part;
      const int culled = __syncthreads_or((int)(flag >> 31));
      const unsigned long long sum = (wave_pre[0] + wave_pre[1]) + (wave_pre[2] + wave_pre[3]);
      slot0 = blockIdx.x == 0 ? 0u : (uint32_t)sum;     // (a count beyond 2^30 is refused by the caller: 32 bits suffice)
      if (blockIdx.x == 0 && threadIdx.x == 0) {
        meta[0] = 0u;
        meta[1] = culled ? 1u : 0u;
        meta[2] = (uint32_t)sum;
        meta[3] = (uint32_t)(sum >> 32);
        meta[4] = meta[5] = meta[6] = meta[7] = 0u;
        if (early)
          __hip_atomic_store(early, (1ull << 63) | ((unsigned long long)(culled ? 1u : 0u) << 62) | (sum & ((1ull << 62) - 1ull)),
                             __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
    n = j < P ? tiles_touched[j] : 0u;
    uint32_t inc = n;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const uint32_t t = __shfl_up(inc, d, 64);
      if ((threadIdx.x & 63) >= d) inc += t;
    }
    if ((threadIdx.x & 63) == 63) wave_tot[threadIdx.x >> 6] = inc;
    __syncthreads();
    const int wv = threadIdx.x >> 6;
    incl = slot0 + inc + (wv > 0 ? wave_tot[0] : 0u) + (wv > 1 ? wave_tot[1] : 0u) + (wv > 2 ? wave_tot[2] : 0u);
  }
  const bool staged = count <= EMIT_WINDOW;
  if (j < P) {
    const uint32_t g = order ? order[j] : (uint32_t)j;      // (tile-local ordering form: emission in index order)
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
      const uint32_t zbits = tile_cutoff ? depth_key[g] : 0u;   // (truncation by depth: the comparison k_preprocess_fwd counted with)
      uint32_t* kdst = staged ? lkey : tile_key;
      uint32_t* gdst = staged ? lgid : gauss_of_slot;
      const uint32_t bias = staged ? slot0 : 0u;
      for (int y = r.y; y < r.w; y++) {
        const uint32_t iv = gsr_row_interval(r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, y, r.x, r.z);
        const int lo = (int)(iv & 0xFFFFu), hi = (int)(iv >> 16);
        if (tile_cutoff) {
          // lists truncated by depth: behind a tile's cut-off an instance is not emitted (nor was it counted); four cut-offs in
          // flight per trip
          const uint32_t* crow = tile_cutoff + y * grid_x;
          for (int x = lo; x < hi && off < end; x += 4) {
            uint32_t c[4];
#pragma unroll
            for (int u = 0; u < 4; u++) c[u] = x + u < hi ? crow[x + u] : 0u;
#pragma unroll
            for (int u = 0; u < 4; u++) {
              if (x + u < hi && zbits <= c[u] && off < end) {
                if (staged || off < cap) {
                  kdst[off - bias] = (uint32_t)(y * grid_x + x + u);
                  gdst[off - bias] = g;
                  if (hist_bits > 0 && !staged) count_digits((uint32_t)(y * grid_x + x + u));
                }
                off++;
              }
            }
          }
          continue;
        }
        for (int x = lo; x < hi && off < end; x++) {
          if (staged || off < cap) {   // (direct path) slots beyond the binning state's capacity are dropped, see below
            kdst[off - bias] = (uint32_t)(y * grid_x + x);
            gdst[off - bias] = g;
            if (hist_bits > 0 && !staged) count_digits((uint32_t)(y * grid_x + x));
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
          if (hist_bits > 0 && !staged) count_digits((uint32_t)(r.y * grid_x + r.x));
        }
      }
    }
  }
  if (staged) {
    __syncthreads();
    // `cap` = instances the binning state has room for.  On the blocking path cap == num_rendered and the guard never
    // fires; on the non-blocking path (gsr_forward_async) a view with more instances than the caller's estimate loses its
    // LAST slots - its farthest splats when emission runs in depth order, its highest Gaussian indices in the tile-local
    // form - instead of writing out of bounds.
    const uint32_t lim = slot0 < cap ? min(count, cap - slot0) : 0u;
    for (uint32_t i = threadIdx.x; i < lim; i += 256) {
      const uint32_t k = lkey[i];
      tile_key[slot0 + i] = k;
      gauss_of_slot[slot0 + i] = lgid[i];
      if (hist_bits > 0) count_digits(k);          // exactly the keys that exist for the sort: slots below the capacity
    }
  }
  if (hist_bits > 0) {
    __syncthreads();
    uint32_t* rep = sort_head + gsr_hist_replica((int)(blockIdx.x % GSR_HIST_REPLICAS));
    for (int i = threadIdx.x; i < hpasses * GSR_RADIX_SIZE; i += 256) {
      const uint32_t c = lhist[i];
      if (c) atomicAdd(&rep[i], c);
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

// ---------------------------------------------------------------------------------------------------------------
// Tile-local depth ordering (second form of the binning stage, gsr_forward_async(tile_local_sort = 1)).
//
// The first form depth-sorts all P Gaussians (4 radix passes, each paying ~15 us of inter-workgroup coordination at 1 M keys
// for 16 MB of traffic), scans their tile counts through a gather and emits in depth order, so that the STABLE tile sort
// leaves every tile's list in (depth, id) order.  Here nothing global is ordered by depth: instances are emitted in INDEX
// order (coalesced record reads; the tile-count scan collapses into the projection and emission kernels plus one
// single-workgroup launch), the same stable tile sort leaves every list in id order, and one workgroup per tile orders ITS
// list by (depth bits, position) - position = id order, so ties break exactly as before - with a stable LSD radix sort in
// LDS, then permutes the list (and the emission slots the backward needs) in place.  ~550 entries per tile at C3.  Lists
// longer than GSR_TLO_CAP are ordered by a bitonic network in global memory (the two free ping-pong halves of the tile sort
// hold depth bits and positions): correct, slow, and reported (meta[4] = longest list) so that the caller goes back to the
// first form for scenes that need it.  Results are bit-identical to the first form.
// ---------------------------------------------------------------------------------------------------------------
#define GSR_TLO_CAP 4096
#define GSR_TLO_SMALL 1024
#ifndef GSR_TLO_SPLIT
#define GSR_TLO_SPLIT 1      // lists up to GSR_TLO_SMALL in a launch of their own with a quarter of the LDS (more tiles in flight per CU)
#endif
// DECODE (the first launch over the tiles, round 4): `ranges` arrives as the tile sort's last pass left it - (~first position,
// last position + 1) maxima, (0, 0) for a tile without instances - and leaves as [start, end) for every later reader.  Every
// thread decodes its own copy of the pair first; thread 0's store of the decoded pair is only ever read by later kernels.
template <bool DUAL, int CAP, int ABOVE, bool DECODE>
__global__ __launch_bounds__(256) void k_tile_depth_sort(const uint2* __restrict__ ranges_in, uint2* __restrict__ ranges_out,
                                                         uint32_t* __restrict__ point_list,
                                                         uint32_t* __restrict__ slot_of_pos,
                                                         const uint32_t* __restrict__ depth_key, uint32_t* __restrict__ free_a,
                                                         uint32_t* __restrict__ free_b, uint32_t* __restrict__ free_c,
                                                         uint32_t* __restrict__ meta, uint32_t* __restrict__ walk_cnt) {
  // (the walk-class counters the compositing kernel behind this launch adds to: cleared here, gsr_common.h gsr_walk_class)
  if (walk_cnt && blockIdx.x == 0 && threadIdx.x < GSR_WALK_CLASSES) walk_cnt[threadIdx.x] = 0u;
  // LDS path: stable LSD radix sort of the list's 32-bit depth keys (8-bit digits, passes whose digit is the same for every
  // key are skipped - the keys of one tile usually differ in their low ~20 bits only), ranked like the global sort: each
  // wave ranks a contiguous quarter of the list with ballots and wave-private counters.  The position in the id-ordered
  // input rides along, so equal depths keep ascending id.  Linear in the list length (a bitonic network on the same data
  // moved 20x the bytes through LDS and took 0.17 ms per frame at C3).
  __shared__ uint32_t skey[CAP];                   // one buffer: a pass reads its elements into registers, then scatters
  __shared__ uint16_t sidx[CAP];
  __shared__ uint32_t wave_run[4][256];
  __shared__ uint32_t dstart[256];
  __shared__ uint32_t red[4], scan4[4];
  uint2 range = ranges_in[blockIdx.x];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  if (DECODE) {
    range.x = range.y != 0u ? ~range.x : 0u;
    if (tid == 0) ranges_out[blockIdx.x] = range;
  }
  const uint32_t x = range.x, n = range.y - range.x;
  if (n <= 1u || n <= (uint32_t)ABOVE) return;              // block-uniform (ABOVE: the lists another launch orders)
  if (CAP < GSR_TLO_CAP && n > (uint32_t)CAP) return;
  // longest list of this frame, for the caller's choice of binning form - only lists past half the LDS capacity report
  // (same-address atomics cost ~15 ns each on this part: one per tile would be 0.12 ms at 1080p)
  if (CAP == GSR_TLO_CAP && tid == 0 && n > GSR_TLO_CAP / 2) atomicMax(&meta[4], n);
  if (n <= (uint32_t)CAP) {
    uint32_t diff = 0, k0 = 0;
    for (uint32_t i = tid; i < n; i += 256) {
      const uint32_t g = point_list[x + i];
      const uint32_t d = g != 0xFFFFFFFFu ? depth_key[g] : 0xFFFFFFFFu;   // (padding slot of k_emit_instances: last)
      skey[i] = d;
      sidx[i] = (uint16_t)i;
      if (i == (uint32_t)tid) k0 = d;
      diff |= d ^ k0;                                        // bits in which this thread's keys differ from its first one
    }
    // bits in which ANY two keys of the list differ: OR over threads of (own differences | own first key ^ thread 0's first key)
    __syncthreads();
    diff |= (tid < (int)n ? k0 : skey[0]) ^ skey[0];
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) diff |= __shfl_xor(diff, d, 64);
    if (lane == 0) red[w] = diff;
    __syncthreads();
    diff = red[0] | red[1] | red[2] | red[3];
    const uint32_t q = (((n + 3) >> 2) + 63) & ~63u;         // elements per wave (multiple of 64)
    const uint32_t lo = min(n, (uint32_t)w * q), hi = min(n, lo + q);
    const unsigned long long lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    for (int shift = 0; shift < 32; shift += 8) {
      if (((diff >> shift) & 0xFFu) == 0u) continue;          // every key has the same digit here: identity pass
      wave_run[0][tid] = 0; wave_run[1][tid] = 0; wave_run[2][tid] = 0; wave_run[3][tid] = 0;
      __syncthreads();
      uint32_t rk[CAP / 256], kreg[CAP / 256];   // rk: digit << 24 | input position << 12 | rank in (wave, digit)
#pragma unroll
      for (int sb = 0; sb < CAP / 256; sb++) {
        const uint32_t i = lo + (uint32_t)sb * 64 + lane;
        rk[sb] = 0; kreg[sb] = 0;
        if (lo + (uint32_t)sb * 64 < hi) {                    // wave-uniform
          const bool active = i < hi;
          kreg[sb] = active ? skey[i] : 0u;
          const uint32_t src = active ? (uint32_t)sidx[i] : 0u;
          const uint32_t d = (kreg[sb] >> shift) & 0xFFu;
          unsigned long long peers = __ballot(active);
#pragma unroll
          for (int b = 0; b < 8; b++) {
            const bool bit = (d >> b) & 1u;
            const unsigned long long bal = __ballot(active && bit);
            peers &= bit ? bal : ~bal;
          }
          const uint32_t rank = __popcll(peers & lt_mask);
          const uint32_t run = wave_run[w][d];
          rk[sb] = (d << 24) | (src << 12) | (run + rank);    // run + rank < q <= 1024, src < 4096
          if (active && rank == 0) wave_run[w][d] = run + (uint32_t)__popcll(peers);
        }
      }
      __syncthreads();
      {
        const uint32_t c0 = wave_run[0][tid], c1 = wave_run[1][tid], c2 = wave_run[2][tid], c3 = wave_run[3][tid];
        // exclusive scan of the digit totals over the 256 threads
        uint32_t v = c0 + c1 + c2 + c3, inc = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
          const uint32_t t = __shfl_up(inc, d, 64);
          if (lane >= d) inc += t;
        }
        if (lane == 63) scan4[w] = inc;
        __syncthreads();
        uint32_t base = 0;
#pragma unroll
        for (int i = 0; i < 4; i++) if (i < w) base += scan4[i];
        dstart[tid] = base + inc - v;
        wave_run[0][tid] = 0; wave_run[1][tid] = c0; wave_run[2][tid] = c0 + c1; wave_run[3][tid] = c0 + c1 + c2;
      }
      __syncthreads();
#pragma unroll
      for (int sb = 0; sb < CAP / 256; sb++) {
        const uint32_t i = lo + (uint32_t)sb * 64 + lane;
        if (i < hi) {
          const uint32_t d = rk[sb] >> 24;
          const uint32_t pos = dstart[d] + wave_run[w][d] + (rk[sb] & 0xFFFu);
          skey[pos] = kreg[sb];                               // (every element was read before the barriers above)
          sidx[pos] = (uint16_t)((rk[sb] >> 12) & 0xFFFu);
        }
      }
      __syncthreads();
    }
    // permute the payloads in place: every source is read before anything is written
    uint32_t gsrc[CAP / 256], ssrc[DUAL ? CAP / 256 : 1];
#pragma unroll
    for (int u = 0; u < CAP / 256; u++) {
      const uint32_t i = (uint32_t)u * 256 + tid;
      if (i < n) {
        const uint32_t src = sidx[i];
        gsrc[u] = point_list[x + src];
        if (DUAL) ssrc[u] = slot_of_pos[x + src];
      }
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < CAP / 256; u++) {
      const uint32_t i = (uint32_t)u * 256 + tid;
      if (i < n) {
        point_list[x + i] = gsrc[u];
        if (DUAL) slot_of_pos[x + i] = ssrc[u];
      }
    }
    return;
  }
  uint32_t npad = 2;
  while (npad < n) npad <<= 1;
  // ---- a list beyond the LDS capacity: the same network on (free_a = depth bits, free_b = position) in global memory ----
  for (uint32_t i = tid; i < n; i += 256) {
    const uint32_t g = point_list[x + i];
    free_a[x + i] = g != 0xFFFFFFFFu ? depth_key[g] : 0xFFFFFFFFu;
    free_b[x + i] = i;
  }
  __syncthreads();
  // "normalised" bitonic network: every compare-exchange is ascending (the smaller key goes to the lower index), the first
  // step of each merge pairs p with its mirror k-1-p.  Positions >= n are virtual +infinity: they sit at the top, compare as
  // the larger partner and therefore never move, so no padding has to exist in memory.
  auto cmpx = [&](uint32_t i, uint32_t l) __attribute__((always_inline)) {
    if (l >= n) return;
    const unsigned long long a = ((unsigned long long)free_a[x + i] << 32) | free_b[x + i];
    const unsigned long long b = ((unsigned long long)free_a[x + l] << 32) | free_b[x + l];
    if (a > b) {
      free_a[x + i] = (uint32_t)(b >> 32); free_b[x + i] = (uint32_t)b;
      free_a[x + l] = (uint32_t)(a >> 32); free_b[x + l] = (uint32_t)a;
    }
  };
  for (uint32_t k = 2; k <= npad; k <<= 1) {
    const uint32_t h = k >> 1;
    for (uint32_t t = tid; t < (npad >> 1); t += 256) {
      const uint32_t blk = t / h, p = t - blk * h;
      cmpx(blk * k + p, blk * k + (k - 1 - p));
    }
    __syncthreads();
    for (uint32_t j = h >> 1; j > 0; j >>= 1) {
      for (uint32_t t = tid; t < (npad >> 1); t += 256) {
        const uint32_t i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
        cmpx(i, i | j);
      }
      __syncthreads();
    }
  }
  for (uint32_t i = tid; i < n; i += 256) free_c[x + i] = point_list[x + free_b[x + i]];
  __syncthreads();
  for (uint32_t i = tid; i < n; i += 256) point_list[x + i] = free_c[x + i];
  if (DUAL) {
    __syncthreads();
    for (uint32_t i = tid; i < n; i += 256) free_c[x + i] = slot_of_pos[x + free_b[x + i]];
    __syncthreads();
    for (uint32_t i = tid; i < n; i += 256) slot_of_pos[x + i] = free_c[x + i];
  }
}

// ranges_enc: nullptr (ranges hold [start, end) already: k_finalize_bins ran) or the encoded pairs of the sort's last pass, which
// the first launch decodes into `ranges`
void gsr_launch_tile_depth_sort(int tiles, bool dual, uint2* ranges, const uint2* ranges_enc, uint32_t* point_list,
                                uint32_t* slot_of_pos, const uint32_t* depth_key, uint32_t* free_a, uint32_t* free_b,
                                uint32_t* free_c, uint32_t* meta, uint32_t* walk_cnt, hipStream_t st) {
  const uint2* rin = ranges_enc ? ranges_enc : (const uint2*)ranges;
  // (ABOVEV == 0: the first launch over the tiles - the one that clears the walk-class counters)
#define GSR_TLO(NAME, D, CAPV, ABOVEV, DEC, RIN)                                                                         \
  GSR_LAUNCH(NAME, (k_tile_depth_sort<D, CAPV, ABOVEV, DEC>), dim3(tiles), dim3(256), 0, st, RIN, ranges, point_list, \
             slot_of_pos, depth_key, free_a, free_b, free_c, meta, (ABOVEV) == 0 ? walk_cnt : (uint32_t*)nullptr)
#if GSR_TLO_SPLIT
  if (dual) {
    if (ranges_enc) GSR_TLO("tile_depth_sort", true, GSR_TLO_SMALL, 0, true, rin);
    else GSR_TLO("tile_depth_sort", true, GSR_TLO_SMALL, 0, false, rin);
    GSR_TLO("tile_depth_sort_long", true, GSR_TLO_CAP, GSR_TLO_SMALL, false, (const uint2*)ranges);
  } else {
    if (ranges_enc) GSR_TLO("tile_depth_sort", false, GSR_TLO_SMALL, 0, true, rin);
    else GSR_TLO("tile_depth_sort", false, GSR_TLO_SMALL, 0, false, rin);
    GSR_TLO("tile_depth_sort_long", false, GSR_TLO_CAP, GSR_TLO_SMALL, false, (const uint2*)ranges);
  }
#else
  if (dual) {
    if (ranges_enc) GSR_TLO("tile_depth_sort", true, GSR_TLO_CAP, 0, true, rin);
    else GSR_TLO("tile_depth_sort", true, GSR_TLO_CAP, 0, false, rin);
  } else {
    if (ranges_enc) GSR_TLO("tile_depth_sort", false, GSR_TLO_CAP, 0, true, rin);
    else GSR_TLO("tile_depth_sort", false, GSR_TLO_CAP, 0, false, rin);
  }
#endif
#undef GSR_TLO
}

// count_hist_bits > 0 (tile-local form, sort head zeroed by the projection kernel): the emission also counts the tile sort's
// digit histograms and clears its look-back table; the encoded tile ranges (BL.ranges_enc) are what it zero-initialises then
void gsr_launch_emit(int P, int grid_x, int tiles, char* geom, const GsrGeomLayout& GL, char* bin,
                     const GsrBinLayout& BL, uint32_t cap, bool index_order, unsigned long long* early, int count_hist_bits,
                     const uint32_t* tile_cutoff, hipStream_t st) {
  // (the look-back words of the sort's FIRST pass; every pass clears the next one's itself - sort_scan.hip)
  const size_t lb_words = count_hist_bits > 0 ? (size_t)gsr_radix_blocks(cap) * GSR_RADIX_SIZE : 0;
  GSR_LAUNCH("emit_instances", k_emit_instances, dim3((P + 255) / 256), dim3(256), 0, st, P, grid_x,
             index_order ? (const uint32_t*)nullptr : (const uint32_t*)(geom + GL.order), (const uint32_t*)(geom + GL.offsets),
             (const uint32_t*)(geom + GL.tiles_touched),
             (const float4*)(geom + GL.bin_rec), (uint32_t*)(bin + BL.key_a), (uint32_t*)(bin + BL.gauss_of_slot),
             (uint32_t*)(geom + GL.slot_start), tiles, (uint2*)(bin + (count_hist_bits > 0 ? BL.ranges_enc : BL.ranges)), cap,
             (uint32_t*)(bin + BL.radix_tmp), (uint32_t*)(geom + GL.meta),
             index_order ? early : (unsigned long long*)nullptr, count_hist_bits, lb_words, tile_cutoff,
             (const uint32_t*)(geom + GL.depth_key));
}

void gsr_launch_finalize(uint32_t cap, const uint32_t* n_dev, const uint32_t* tile_sorted, char* bin,
                         const GsrBinLayout& BL, hipStream_t st) {
  GSR_LAUNCH("finalize_bins", k_finalize_bins, dim3((cap + 255) / 256), dim3(256), 0, st, cap, n_dev, tile_sorted,
             (uint2*)(bin + BL.ranges));
}
