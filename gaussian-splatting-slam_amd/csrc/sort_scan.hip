// sort_scan.hip - device-wide prefix sum and stable LSD radix sort for the binning stage (gfx950, wave64).
//
// Replaces the two library calls of the published rasterizer's binning stage (SURVEY.md 2.3 K2
// cub::DeviceScan::InclusiveSum and K4 cub::DeviceRadixSort::SortPairs) with hand-written kernels.
// Both are HBM-bound integer work; the design rules are coalesced 4-B/lane streams, LDS histograms,
// wave64 ballots for stable ranking, and no inter-workgroup hand-offs (each pass is reduce -> scan ->
// scatter with kernel boundaries as the only grid-wide synchronisation).
#include "gsr_common.h"

// ---------------------------------------------------------------------------------------------------
// wave / block primitives
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t wave_incl_scan_u32(uint32_t v) {
  const int lane = gsr_lane();
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    uint32_t t = __shfl_up(v, d, 64);
    if (lane >= d) v += t;
  }
  return v;
}

// exclusive block scan of one value per thread (256 threads); returns exclusive prefix, *total = block sum
__device__ __forceinline__ uint32_t block_excl_scan_u32(uint32_t v, uint32_t* total, uint32_t* lds4) {
  const int lane = gsr_lane(), w = threadIdx.x >> 6;
  uint32_t inc = wave_incl_scan_u32(v);
  if (lane == 63) lds4[w] = inc;
  __syncthreads();
  uint32_t base = 0, tot = 0;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    uint32_t s = lds4[i];
    if (i < w) base += s;
    tot += s;
  }
  __syncthreads();
  *total = tot;
  return base + inc - v;
}

// ---------------------------------------------------------------------------------------------------
// scan: reduce per chunk -> (recursive) scan of chunk sums -> apply
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_scan_reduce(const uint32_t* __restrict__ src,
                                                     const uint32_t* __restrict__ idx, uint32_t* __restrict__ sums,
                                                     size_t n) {
  __shared__ uint32_t lds4[4];
  const size_t base = (size_t)blockIdx.x * GSR_SCAN_CHUNK + (size_t)threadIdx.x * GSR_SCAN_ITEMS;
  uint32_t s = 0;
#pragma unroll
  for (int i = 0; i < GSR_SCAN_ITEMS; i++) {
    size_t k = base + i;
    if (k < n) s += idx ? src[idx[k]] : src[k];
  }
  uint32_t tot;
  block_excl_scan_u32(s, &tot, lds4);
  if (threadIdx.x == 0) sums[blockIdx.x] = tot;
}

__global__ __launch_bounds__(256) void k_scan_apply(const uint32_t* src,
                                                    const uint32_t* __restrict__ idx,
                                                    const uint32_t* __restrict__ sums_excl, uint32_t* out,
                                                    size_t n, int inclusive) {
  __shared__ uint32_t lds4[4];
  const size_t base = (size_t)blockIdx.x * GSR_SCAN_CHUNK + (size_t)threadIdx.x * GSR_SCAN_ITEMS;
  uint32_t v[GSR_SCAN_ITEMS];
  uint32_t s = 0;
#pragma unroll
  for (int i = 0; i < GSR_SCAN_ITEMS; i++) {
    size_t k = base + i;
    v[i] = (k < n) ? (idx ? src[idx[k]] : src[k]) : 0u;
    s += v[i];
  }
  uint32_t tot;
  uint32_t run = block_excl_scan_u32(s, &tot, lds4) + (sums_excl ? sums_excl[blockIdx.x] : 0u);
#pragma unroll
  for (int i = 0; i < GSR_SCAN_ITEMS; i++) {
    size_t k = base + i;
    uint32_t e = run;
    run += v[i];
    if (k < n) out[k] = inclusive ? run : e;
  }
}

// One workgroup walks the whole array chunk by chunk carrying the running total: one launch instead of three for the
// small arrays (digit tables, chunk sums) where launch gaps, not bytes, are the cost.
__global__ __launch_bounds__(256) void k_scan_single(const uint32_t* src, const uint32_t* __restrict__ idx, uint32_t* out,
                                                     size_t n, int inclusive) {
  __shared__ uint32_t lds4[4];
  uint32_t carry = 0;
  for (size_t c0 = 0; c0 < n; c0 += GSR_SCAN_CHUNK) {
    const size_t base = c0 + (size_t)threadIdx.x * GSR_SCAN_ITEMS;
    uint32_t v[GSR_SCAN_ITEMS];
    uint32_t s = 0;
#pragma unroll
    for (int i = 0; i < GSR_SCAN_ITEMS; i++) {
      size_t k = base + i;
      v[i] = (k < n) ? (idx ? src[idx[k]] : src[k]) : 0u;
      s += v[i];
    }
    uint32_t tot;
    uint32_t run = block_excl_scan_u32(s, &tot, lds4) + carry;
#pragma unroll
    for (int i = 0; i < GSR_SCAN_ITEMS; i++) {
      size_t k = base + i;
      uint32_t e = run;
      run += v[i];
      if (k < n) out[k] = inclusive ? run : e;
    }
    carry += tot;
  }
}

#define GSR_SCAN_SINGLE_MAX (4 * GSR_SCAN_CHUNK)   // 8192 elements; beyond that the serial chunk chain of one workgroup
                                                  // (74 us for 62 k elements, measured) loses to three parallel launches

void gsr_scan_u32(const uint32_t* src, const uint32_t* idx, uint32_t* out, size_t n, int inclusive,
                  uint32_t* tmp, hipStream_t st) {
  if (n == 0) return;
  const size_t nblk = (n + GSR_SCAN_CHUNK - 1) / GSR_SCAN_CHUNK;
  if (n <= GSR_SCAN_SINGLE_MAX && nblk > 1) {
    GSR_LAUNCH("scan_single", k_scan_single, dim3(1), dim3(256), 0, st, src, idx, out, n, inclusive);
    return;
  }
  if (nblk == 1) {
    GSR_LAUNCH("scan_apply", k_scan_apply, dim3(1), dim3(256), 0, st, src, idx, (const uint32_t*)nullptr, out, n,
               inclusive);
    return;
  }
  uint32_t* sums = tmp;
  uint32_t* next_tmp = tmp + gsr_align(nblk * 4) / 4;
  GSR_LAUNCH("scan_reduce", k_scan_reduce, dim3((unsigned)nblk), dim3(256), 0, st, src, idx, sums, n);
  gsr_scan_u32(sums, nullptr, sums, nblk, 0, next_tmp, st);  // in-place exclusive scan of chunk sums
  GSR_LAUNCH("scan_apply", k_scan_apply, dim3((unsigned)nblk), dim3(256), 0, st, src, idx, (const uint32_t*)sums,
             out, n, inclusive);
}

// ---------------------------------------------------------------------------------------------------
// radix sort, 8-bit digits, stable.
//   pass = k_radix_hist (per-chunk digit counts, table[digit][chunk]) -> k_radix_rowscan (per-digit exclusive scan
//          over chunks + digit totals) -> k_radix_scatter (re-read the chunk, stable rank by wave ballots, scatter).
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_radix_hist(const uint32_t* __restrict__ keys, uint32_t* __restrict__ table,
                                                    size_t n_max, const uint32_t* __restrict__ n_dev, int shift,
                                                    uint32_t mask, uint32_t nblk, int subtiles) {
  __shared__ uint32_t hist[GSR_RADIX_SIZE];
  const size_t n = gsr_eff_n(n_dev, (uint32_t)n_max);   // chunks beyond n still write their (all-zero) table column
  hist[threadIdx.x] = 0;
  __syncthreads();
  const size_t base = (size_t)blockIdx.x * 256 * subtiles;
#pragma unroll 4
  for (int s = 0; s < subtiles; s++) {
    size_t k = base + (size_t)s * 256 + threadIdx.x;
    if (k < n) atomicAdd(&hist[(keys[k] >> shift) & mask], 1u);
  }
  __syncthreads();
  table[(size_t)threadIdx.x * nblk + blockIdx.x] = hist[threadIdx.x];
}

// One workgroup per digit: exclusive scan of that digit's per-chunk counts (row `d` of the table) in place, row total to
// totals[d].  Replaces a 3-launch device-wide scan of the whole table; the 256-entry scan of the totals is redone by
// every scatter workgroup in LDS (trivial) instead of costing a launch.
__global__ __launch_bounds__(256) void k_radix_rowscan(uint32_t* __restrict__ table, uint32_t* __restrict__ totals,
                                                       uint32_t nblk) {
  __shared__ uint32_t lds4[4];
  uint32_t* row = table + (size_t)blockIdx.x * nblk;
  uint32_t carry = 0;
  for (uint32_t c0 = 0; c0 < nblk; c0 += GSR_SCAN_CHUNK) {
    const uint32_t base = c0 + threadIdx.x * GSR_SCAN_ITEMS;
    uint32_t v[GSR_SCAN_ITEMS];
    uint32_t s = 0;
#pragma unroll
    for (int i = 0; i < GSR_SCAN_ITEMS; i++) {
      v[i] = (base + i < nblk) ? row[base + i] : 0u;
      s += v[i];
    }
    uint32_t tot;
    uint32_t run = block_excl_scan_u32(s, &tot, lds4) + carry;
#pragma unroll
    for (int i = 0; i < GSR_SCAN_ITEMS; i++) {
      if (base + i < nblk) row[base + i] = run;
      run += v[i];
    }
    carry += tot;
  }
  if (threadIdx.x == 0) totals[blockIdx.x] = carry;
}

// Scatter pass.  Each workgroup owns one chunk of 4096 keys; wave w owns the contiguous quarter [1024 w, 1024 (w+1)) of it,
// 16 sub-tiles of 64 keys, so the original order is (wave, sub-tile, lane).
//   phase 1  every wave ranks its own keys inside their (wave, digit) group with ballots and a wave-PRIVATE LDS counter row:
//            no workgroup barrier in the loop (LDS operations of one wave execute in order);
//   phase 2  one barrier, then per digit: totals over the four waves, each wave's base inside the digit, local start of the
//            digit (block scan) - the chunk is sorted locally into LDS;
//   phase 3  LDS is streamed out so that each digit's run lands in consecutive global addresses (64-B+ runs instead of the
//            4-B scattered stores of a direct scatter).
// Stable: equal digits keep (wave, sub-tile, lane) = original order.
// DUAL: a second 32-bit payload rides along (the tile sort carries the Gaussian id next to the emission slot, so no
// gather by slot is needed afterwards).
template <bool DUAL, int SUBTILES>   // chunk = 256 * SUBTILES keys; wave w owns keys [64 SUBTILES w, 64 SUBTILES (w+1))
__global__ __launch_bounds__(256) void k_radix_scatter(const uint32_t* __restrict__ keys_in,
                                                       const uint32_t* __restrict__ vals_in,
                                                       const uint32_t* __restrict__ vals2_in,
                                                       uint32_t* __restrict__ keys_out, uint32_t* __restrict__ vals_out,
                                                       uint32_t* __restrict__ vals2_out,
                                                       const uint32_t* __restrict__ table_excl,
                                                       const uint32_t* __restrict__ totals, size_t n_max,
                                                       const uint32_t* __restrict__ n_dev, int shift, uint32_t mask,
                                                       uint32_t nblk) {
  __shared__ uint32_t wave_run[4][GSR_RADIX_SIZE];  // phase 1: keys of (wave, digit) seen so far; phase 2: the wave's base
  __shared__ uint32_t lstart[GSR_RADIX_SIZE];       // first local (sorted) position of each digit
  __shared__ uint32_t gbase[GSR_RADIX_SIZE];        // global position of the chunk's first key of each digit
  __shared__ uint32_t lbuf[(256 * SUBTILES)];        // ONE staging buffer, reused for keys, values (and the second payload):
                                                    // 21 KB of LDS per workgroup instead of 36 / 52 -> 7 workgroups per CU
  __shared__ uint32_t lds4[4];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const size_t n = gsr_eff_n(n_dev, (uint32_t)n_max);
  if ((size_t)blockIdx.x * (256 * SUBTILES) >= n) return;   // block-uniform: a chunk beyond the keys present
  {
    uint32_t tot;
    const uint32_t digit_start = block_excl_scan_u32(totals[tid], &tot, lds4);   // keys with a smaller digit, whole array
    gbase[tid] = digit_start + table_excl[(size_t)tid * nblk + blockIdx.x];
  }
#pragma unroll
  for (int i = 0; i < 4; i++) wave_run[i][tid] = 0;
  __syncthreads();
  const size_t base = (size_t)blockIdx.x * (256 * SUBTILES);
  const uint32_t count = (uint32_t)min((size_t)(256 * SUBTILES), n - base);
  const unsigned long long lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
  const uint32_t wbase = (uint32_t)w * (64 * SUBTILES);
  uint32_t* my_run = wave_run[w];
  uint32_t key[SUBTILES], val[SUBTILES], rk[SUBTILES];   // rk = digit << 16 | rank in (wave, digit)
  uint32_t val2[DUAL ? SUBTILES : 1];
#pragma unroll
  for (int s = 0; s < SUBTILES; s++) {
    const uint32_t li = wbase + (uint32_t)s * 64 + lane;
    const bool active = li < count;
    key[s] = active ? keys_in[base + li] : 0u;
    val[s] = active ? (vals_in ? vals_in[base + li] : (uint32_t)(base + li)) : 0u;
    if (DUAL) val2[s] = active ? vals2_in[base + li] : 0u;
  }
#pragma unroll
  for (int s = 0; s < SUBTILES; s++) {
    const uint32_t li = wbase + (uint32_t)s * 64 + lane;
    const bool active = li < count;
    const uint32_t d = active ? ((key[s] >> shift) & mask) : 0u;
    rk[s] = 0;
    if (wbase + (uint32_t)s * 64 < count) {   // wave-uniform
      unsigned long long peers = __ballot(active);
#pragma unroll
      for (int b = 0; b < GSR_RADIX_BITS; b++) {
        const bool bit = (d >> b) & 1u;
        const unsigned long long bal = __ballot(active && bit);
        peers &= bit ? bal : ~bal;
      }
      const uint32_t rank = __popcll(peers & lt_mask);
      const uint32_t run = my_run[d];                                   // every lane of the group reads the same counter ...
      rk[s] = (d << 16) | (run + rank);
      if (active && rank == 0) my_run[d] = run + (uint32_t)__popcll(peers);   // ... before its first lane advances it
    }
  }
  __syncthreads();
  {
    // per digit (thread = digit): totals over the waves, each wave's base inside the digit, local start of the digit
    const uint32_t c0 = wave_run[0][tid], c1 = wave_run[1][tid], c2 = wave_run[2][tid], c3 = wave_run[3][tid];
    uint32_t tot;
    lstart[tid] = block_excl_scan_u32(c0 + c1 + c2 + c3, &tot, lds4);
    wave_run[0][tid] = 0;
    wave_run[1][tid] = c0;
    wave_run[2][tid] = c0 + c1;
    wave_run[3][tid] = c0 + c1 + c2;
  }
  __syncthreads();
  // local sorted position of each of this thread's keys
#pragma unroll
  for (int s = 0; s < SUBTILES; s++) {
    const uint32_t d = rk[s] >> 16;
    rk[s] = lstart[d] + my_run[d] + (rk[s] & 0xFFFFu);
  }
  // round 1: keys through LDS; each thread also learns the global position of the sorted positions it streams out
  uint32_t gpos[SUBTILES];
#pragma unroll
  for (int s = 0; s < SUBTILES; s++)
    if (wbase + (uint32_t)s * 64 + lane < count) lbuf[rk[s]] = key[s];
  __syncthreads();
#pragma unroll
  for (int k = 0; k < SUBTILES; k++) {
    const uint32_t i = (uint32_t)k * 256 + tid;
    gpos[k] = 0;
    if (i < count) {
      const uint32_t kk = lbuf[i];
      const uint32_t d = (kk >> shift) & mask;
      gpos[k] = gbase[d] + (i - lstart[d]);
      keys_out[gpos[k]] = kk;
    }
  }
  __syncthreads();
  // round 2: values
#pragma unroll
  for (int s = 0; s < SUBTILES; s++)
    if (wbase + (uint32_t)s * 64 + lane < count) lbuf[rk[s]] = val[s];
  __syncthreads();
#pragma unroll
  for (int k = 0; k < SUBTILES; k++) {
    const uint32_t i = (uint32_t)k * 256 + tid;
    if (i < count) vals_out[gpos[k]] = lbuf[i];
  }
  if (DUAL) {   // round 3: the second payload
    __syncthreads();
#pragma unroll
    for (int s = 0; s < SUBTILES; s++)
      if (wbase + (uint32_t)s * 64 + lane < count) lbuf[rk[s]] = val2[s];
    __syncthreads();
#pragma unroll
    for (int k = 0; k < SUBTILES; k++) {
      const uint32_t i = (uint32_t)k * 256 + tid;
      if (i < count) vals2_out[gpos[k]] = lbuf[i];
    }
  }
}

int gsr_radix_sort_pairs(uint32_t* k0, uint32_t* v0, uint32_t* k1, uint32_t* v1, bool vals_iota, size_t n,
                         int bits, uint32_t* tmp, hipStream_t st, uint32_t* w0, uint32_t* w1, const uint32_t* n_dev) {
  if (n == 0 || bits <= 0) return 0;
  const uint32_t nblk = (uint32_t)gsr_radix_blocks(n);
  const int subtiles = gsr_radix_subtiles(n);
  uint32_t* table = tmp;
  uint32_t* totals = tmp + (size_t)GSR_RADIX_SIZE * nblk;     // sized by gsr_radix_tmp_elems: 256 * (nblk + 1)
  const bool dual = w0 != nullptr && w1 != nullptr;
  int cur = 0;
  for (int shift = 0; shift < bits; shift += GSR_RADIX_BITS) {
    const int nb = (bits - shift) < GSR_RADIX_BITS ? (bits - shift) : GSR_RADIX_BITS;
    const uint32_t mask = (1u << nb) - 1u;
    uint32_t* ki = cur ? k1 : k0;
    uint32_t* vi = cur ? v1 : v0;
    uint32_t* ko = cur ? k0 : k1;
    uint32_t* vo = cur ? v0 : v1;
    const uint32_t* vin = (shift == 0 && vals_iota) ? nullptr : vi;
    GSR_LAUNCH("radix_hist", k_radix_hist, dim3(nblk), dim3(256), 0, st, (const uint32_t*)ki, table, n, n_dev, shift,
               mask, nblk, subtiles);
    GSR_LAUNCH("radix_rowscan", k_radix_rowscan, dim3(GSR_RADIX_SIZE), dim3(256), 0, st, table, totals, nblk);
#define GSR_SCATTER(D, S)                                                                                              \
  GSR_LAUNCH("radix_scatter", (k_radix_scatter<D, S>), dim3(nblk), dim3(256), 0, st, (const uint32_t*)ki, vin,         \
             (const uint32_t*)(D ? (cur ? w1 : w0) : nullptr), ko, vo, (uint32_t*)(D ? (cur ? w0 : w1) : nullptr),      \
             (const uint32_t*)table, (const uint32_t*)totals, n, n_dev, shift, mask, nblk)
    if (dual) {
      if (subtiles == GSR_RADIX_SUBTILES_SMALL) GSR_SCATTER(true, GSR_RADIX_SUBTILES_SMALL);
      else GSR_SCATTER(true, GSR_RADIX_SUBTILES);
    } else {
      if (subtiles == GSR_RADIX_SUBTILES_SMALL) GSR_SCATTER(false, GSR_RADIX_SUBTILES_SMALL);
      else GSR_SCATTER(false, GSR_RADIX_SUBTILES);
    }
#undef GSR_SCATTER
    cur ^= 1;
  }
  return cur;
}
