// sort_scan.hip - device-wide prefix sum and stable LSD radix sort for the binning stage (gfx950, wave64).
//
// Replaces the two library calls of the published rasterizer's binning stage (SURVEY.md 2.3 K2
// cub::DeviceScan::InclusiveSum and K4 cub::DeviceRadixSort::SortPairs) with hand-written kernels.
// Both are HBM-bound integer work; the design rules are coalesced 4-B/lane streams, LDS histograms,
// wave64 ballots for stable ranking, and no inter-workgroup hand-offs (each pass is reduce -> scan ->
// scatter with kernel boundaries as the only grid-wide synchronisation).
#include <stdlib.h>

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
// radix sort, 8-bit digits, stable, ONE kernel per pass (+ one histogram kernel per sort).
//
//   k_radix_hist_all  one read of the keys -> the digit histograms of EVERY pass (hist[pass][digit], LDS counters then one
//                     global add per non-zero counter and workgroup); also zeroes the look-back table of the passes.
//   k_radix_pass      per pass: a workgroup takes a chunk of 4096 keys by TICKET (atomic counter: a chunk's predecessors have
//                     all started before it, whatever order the dispatcher chose), ranks its keys (as before: wave ballots,
//                     wave-private LDS counters, chunk sorted locally in LDS), and gets "keys with this digit in earlier
//                     chunks" by DECOUPLED LOOK-BACK instead of from a table two more kernels had to build: every chunk
//                     publishes per digit first its own count (flag 1), later the inclusive prefix (flag 2), as ONE 32-bit
//                     word {flag:2, count:30} written / polled with relaxed agent-scope atomics (global_store / global_load
//                     sc1): the word is its own flag, so no fence or ordering against other data is needed
//                     (/opt/skills/guides/MI355X_MICROARCH.md "Valid forms": an aligned granule written by ONE store).
//                     A chunk walks back over its predecessors until it meets a flag-2 word.
// The 18 launches of a forward's two sorts (4 + 2 passes x hist / rowscan / scatter) become 8; integer work, results are
// bit-identical to the three-kernel form.  Counts are limited to 2^30 - 1 keys by the 30-bit field.
// ---------------------------------------------------------------------------------------------------
#define GSR_LB_AGG (1u << 30)
#define GSR_LB_INC (2u << 30)
#define GSR_LB_VAL 0x3FFFFFFFu
#ifndef GSR_LB_WINDOW
#define GSR_LB_WINDOW 8
#endif

__global__ __launch_bounds__(256) void k_radix_hist_all(const uint32_t* __restrict__ keys, size_t n_max,
                                                        const uint32_t* __restrict__ n_dev, int bits,
                                                        uint32_t* __restrict__ hist, uint32_t* __restrict__ lookback,
                                                        size_t lookback_words) {
  __shared__ uint32_t lh[GSR_RADIX_MAX_PASSES * GSR_RADIX_SIZE];
  const size_t n = gsr_eff_n(n_dev, (uint32_t)n_max);
  const int passes = gsr_radix_passes(bits);
  for (int i = threadIdx.x; i < passes * GSR_RADIX_SIZE; i += 256) lh[i] = 0;
  __syncthreads();
  // the look-back words of every pass start at "nothing published" (keys_in is 256-B aligned, so is the table)
  {
    uint4* z = reinterpret_cast<uint4*>(lookback);
    const size_t n4 = lookback_words >> 2;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) z[i] = make_uint4(0u, 0u, 0u, 0u);
  }
  auto count = [&](uint32_t k) __attribute__((always_inline)) {
#pragma unroll
    for (int p = 0; p < GSR_RADIX_MAX_PASSES; p++) {
      if (p < passes) {
        const uint32_t d = (k >> gsr_radix_shift(bits, p)) & ((1u << gsr_radix_width(bits, p)) - 1u);
        atomicAdd(&lh[p * GSR_RADIX_SIZE + d], 1u);
      }
    }
  };
  const size_t n4 = n >> 2;
  const uint4* k4 = reinterpret_cast<const uint4*>(keys);
  // four 16-B loads in flight per thread: with 512 workgroups and one load per trip the kernel ran at 1 TB/s - latency x
  // (2048 waves x 1 KB in flight) - not at what HBM delivers
  const size_t stride = (size_t)gridDim.x * 256;
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  for (; i + 3 * stride < n4; i += 4 * stride) {
    const uint4 q0 = k4[i], q1 = k4[i + stride], q2 = k4[i + 2 * stride], q3 = k4[i + 3 * stride];
    count(q0.x); count(q0.y); count(q0.z); count(q0.w);
    count(q1.x); count(q1.y); count(q1.z); count(q1.w);
    count(q2.x); count(q2.y); count(q2.z); count(q2.w);
    count(q3.x); count(q3.y); count(q3.z); count(q3.w);
  }
  for (; i < n4; i += stride) {
    const uint4 q = k4[i];
    count(q.x); count(q.y); count(q.z); count(q.w);
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) count(keys[4 * n4 + threadIdx.x]);
  __syncthreads();
  for (int i = threadIdx.x; i < passes * GSR_RADIX_SIZE; i += 256) {
    const uint32_t c = lh[i];
    if (c) atomicAdd(&hist[i], c);
  }
}

// One pass.  Each workgroup owns one chunk of 256 * SUBTILES keys; wave w owns the contiguous quarter of it, SUBTILES
// sub-tiles of 64 keys, so the original order is (wave, sub-tile, lane).
//   phase 1  every wave ranks its own keys inside their (wave, digit) group with ballots and a wave-PRIVATE LDS counter row:
//            no workgroup barrier in the loop (LDS operations of one wave execute in order);
//   phase 2  one barrier, then per digit: totals over the four waves (published for the look-back at once), each wave's base
//            inside the digit, local start of the digit (block scan) - the chunk is sorted locally into LDS;
//   phase 3  look-back for the global base of every digit, then LDS is streamed out so that each digit's run lands in
//            consecutive global addresses (64-B+ runs instead of the 4-B scattered stores of a direct scatter).
// Stable: equal digits keep (chunk, wave, sub-tile, lane) = original order.
// DUAL: a second 32-bit payload rides along (the tile sort carries the Gaussian id next to the emission slot, so no
// gather by slot is needed afterwards).
// RANGES (the tile sort's last pass, tile-local binning form): the sorted keys are tile ids; every run of equal keys inside the
// chunk's locally sorted buffer is a run of consecutive global positions, so its first / last element leave (~first, last + 1)
// in ranges_enc[tile] by atomicMax (a tile's run may span chunks: the maxima over its pieces are its true start and end) -
// k_tile_depth_sort decodes them: no k_finalize_bins launch.
template <bool DUAL, int SUBTILES, bool RANGES>
__global__ __launch_bounds__(256) void k_radix_pass(const uint32_t* __restrict__ keys_in,
                                                    const uint32_t* __restrict__ vals_in,
                                                    const uint32_t* __restrict__ vals2_in,
                                                    uint32_t* __restrict__ keys_out, uint32_t* __restrict__ vals_out,
                                                    uint32_t* __restrict__ vals2_out,
                                                    const uint32_t* __restrict__ head /* the sort's scratch head */, int pass,
                                                    int hist_reps /* replicas of the histograms in use (gsr_common.h) */,
                                                    uint32_t* __restrict__ ticket, uint32_t* lookback /* [chunks][256] */,
                                                    size_t n_max, const uint32_t* __restrict__ n_dev, int shift,
                                                    uint32_t mask, uint32_t* __restrict__ fail_flags, uint32_t force_timeout,
                                                    uint2* __restrict__ ranges_enc, uint32_t* __restrict__ zero_next,
                                                    size_t zero_next_words) {
  __shared__ uint32_t wave_run[4][GSR_RADIX_SIZE];  // phase 1: keys of (wave, digit) seen so far; phase 2: the wave's base
  __shared__ uint32_t lstart[GSR_RADIX_SIZE];       // first local (sorted) position of each digit
  __shared__ uint32_t gbase[GSR_RADIX_SIZE];        // global position of the chunk's first key of each digit
  __shared__ uint32_t lbuf[(256 * SUBTILES)];       // ONE staging buffer, reused for keys, values (and the second payload)
  __shared__ uint32_t lds4[4];
  __shared__ uint32_t s_chunk;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  // (hist_counted sorts) the NEXT pass's look-back words are cleared by this pass - just ahead of their use, so they are still in
  // L2 when that pass polls them - instead of by the kernel that counted the histograms
  if (zero_next) {
    uint4* z = reinterpret_cast<uint4*>(zero_next);
    const size_t n4 = zero_next_words >> 2;
    for (size_t i = (size_t)blockIdx.x * 256 + tid; i < n4; i += (size_t)gridDim.x * 256) z[i] = make_uint4(0u, 0u, 0u, 0u);
  }
  // keys with digit `tid` in this pass, whole array: the sum of the histogram's replicas (a sort that fills replica 0 only leaves
  // zeros in the others).  All loads in flight together: a loop over a run-time count made them 8 dependent L2 round trips, +5 us
  // per pass (profiles/r04_binning_chain_ab.txt)
  uint32_t hrep[GSR_HIST_REPLICAS];
#pragma unroll
  for (int r = 0; r < GSR_HIST_REPLICAS; r++) hrep[r] = head[gsr_hist_replica(r) + (size_t)pass * GSR_RADIX_SIZE + tid];
  uint32_t hist_tid = 0;
#pragma unroll
  for (int r = 0; r < GSR_HIST_REPLICAS; r++) hist_tid += hrep[r];
  (void)hist_reps;
  {
    // Every key has the same digit in this pass (e.g. the top byte of fp32 depths that span less than a factor of 4): the
    // pass is the identity permutation, so the chunk is copied straight across - no ticket, no ranking, no look-back.
    const size_t n0 = gsr_eff_n(n_dev, (uint32_t)n_max);
    if (__syncthreads_or(hist_tid == (uint32_t)n0 && n0 != 0)) {
      const size_t b0 = (size_t)blockIdx.x * (256 * SUBTILES);
      for (uint32_t i = tid; i < (uint32_t)(256 * SUBTILES) && b0 + i < n0; i += 256) {
        const uint32_t kk = keys_in[b0 + i];
        if (!RANGES) keys_out[b0 + i] = kk;
        vals_out[b0 + i] = vals_in ? vals_in[b0 + i] : (uint32_t)(b0 + i);
        if (DUAL) vals2_out[b0 + i] = vals2_in[b0 + i];
        if (RANGES) {      // (the array is sorted already: neighbours in memory are neighbours in the order)
          const size_t g = b0 + i;
          if (g == 0 || keys_in[g - 1] != kk) atomicMax(&ranges_enc[kk].x, ~(uint32_t)g);
          if (g + 1 == n0 || keys_in[g + 1] != kk) atomicMax(&ranges_enc[kk].y, (uint32_t)g + 1u);
        }
      }
      return;
    }
  }
#ifdef GSR_EXP_NOTICKET   // (timing experiment only)
  if (tid == 0) s_chunk = blockIdx.x;
#else
  if (tid == 0) s_chunk = atomicAdd(ticket, 1u);    // chunks are taken in ticket order: every predecessor is already running
#endif
#pragma unroll
  for (int i = 0; i < 4; i++) wave_run[i][tid] = 0;
  __syncthreads();
  const uint32_t chunk = s_chunk;
  const size_t n = gsr_eff_n(n_dev, (uint32_t)n_max);
  const size_t base = (size_t)chunk * (256 * SUBTILES);
  if (base >= n) return;                            // block-uniform: a chunk beyond the keys present (nobody looks back at it)
  uint32_t digit_start;                             // keys with a smaller digit, whole array
  {
    uint32_t tot;
    digit_start = block_excl_scan_u32(hist_tid, &tot, lds4);
  }
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
  uint32_t my_count;
  {
    // per digit (thread = digit): totals over the waves, each wave's base inside the digit, local start of the digit
    const uint32_t c0 = wave_run[0][tid], c1 = wave_run[1][tid], c2 = wave_run[2][tid], c3 = wave_run[3][tid];
    my_count = c0 + c1 + c2 + c3;
    // publish this chunk's count of digit `tid` at once (chunk 0 knows its inclusive prefix already)
    __hip_atomic_store(&lookback[(size_t)chunk * GSR_RADIX_SIZE + tid], (chunk == 0 ? GSR_LB_INC : GSR_LB_AGG) | my_count,
                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    uint32_t tot;
    lstart[tid] = block_excl_scan_u32(my_count, &tot, lds4);
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
  // round 1: keys into LDS in sorted order
#pragma unroll
  for (int s = 0; s < SUBTILES; s++)
    if (wbase + (uint32_t)s * 64 + lane < count) lbuf[rk[s]] = key[s];
  // look-back (thread = digit): keys of this digit in all earlier chunks.  Every word polled belongs to a chunk whose
  // workgroup holds an earlier ticket, i.e. is resident or finished: the wait is bounded by that workgroup's phase 1.
  {
    uint32_t excl = 0;
#ifdef GSR_EXP_NOLB         // (timing experiment only: wrong bases, every store still inside [0, n))
    if (false) {
#else
    if (chunk != 0) {
#endif
      // GSR_LB_WINDOW predecessors per trip, all loads in flight together: with ~1000 chunks started at once the walk back
      // to the nearest finished prefix is tens of chunks long, and one dependent ~1 us poll per chunk made the pass slower
      // than the three-kernel form it replaces
      int64_t p = (int64_t)chunk - 1;
      bool done = false;
      uint32_t spins = 0;
      while (!done) {
        uint32_t v[GSR_LB_WINDOW];
#pragma unroll
        for (int i = 0; i < GSR_LB_WINDOW; i++)
          v[i] = (p - i >= 0) ? __hip_atomic_load(&lookback[(size_t)(p - i) * GSR_RADIX_SIZE + tid], __ATOMIC_RELAXED,
                                                  __HIP_MEMORY_SCOPE_AGENT)
                              : GSR_LB_INC;
        int used = 0;
#pragma unroll
        for (int i = 0; i < GSR_LB_WINDOW; i++) {
          if (!done && used == i) {
            if ((v[i] & ~GSR_LB_VAL) != 0u) {
              excl += v[i] & GSR_LB_VAL;
              done = (v[i] & GSR_LB_INC) != 0u;
              used = i + 1;
            }
          }
        }
        p -= used;                    // the first word not yet published (if any) heads the next window
        if (!done && used < GSR_LB_WINDOW) {
          // (bounded, like every spin should be: ~1 s; a time-out can only mean a broken protocol - it gives this chunk a
          // wrong base and lets the grid drain instead of hanging the GPU, leaves a mark in the word behind the tickets and
          // raises GSR_STATUS_SORT_TIMEOUT in the frame's status word, which the host reads: a mis-sorted frame is REPORTED)
          if (++spins > (1u << 20)) {
            ticket[GSR_RADIX_MAX_PASSES] = 0xDEADu;
            if (fail_flags) atomicOr(fail_flags, GSR_STATUS_SORT_TIMEOUT);
            break;
          }
          __builtin_amdgcn_s_sleep(1);
        }
      }
      __hip_atomic_store(&lookback[(size_t)chunk * GSR_RADIX_SIZE + tid], GSR_LB_INC | (excl + my_count), __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_AGENT);
    }
    gbase[tid] = digit_start + excl;
    // test hook (GSR_TEST_FORCE_LOOKBACK_TIMEOUT=1, tests/test_sort_gpu.py): behave as if this chunk's wait had timed out -
    // the mark and the status bit, without the second of spinning and without touching the sort's result
    if (force_timeout && chunk == 0 && tid == 0) {
      ticket[GSR_RADIX_MAX_PASSES] = 0xDEADu;
      if (fail_flags) atomicOr(fail_flags, GSR_STATUS_SORT_TIMEOUT);
    }
  }
  __syncthreads();
  uint32_t gpos[SUBTILES];
#pragma unroll
  for (int k = 0; k < SUBTILES; k++) {
    const uint32_t i = (uint32_t)k * 256 + tid;
    gpos[k] = 0;
    if (i < count) {
      const uint32_t kk = lbuf[i];
      const uint32_t d = (kk >> shift) & mask;
      gpos[k] = gbase[d] + (i - lstart[d]);
      // (RANGES = the tile sort's last pass: nobody reads the sorted tile ids - the ranges below are all that is wanted of them)
      if (!RANGES) keys_out[gpos[k]] = kk;
      if (RANGES) {
        if (i == 0 || lbuf[i - 1] != kk) atomicMax(&ranges_enc[kk].x, ~gpos[k]);
        if (i + 1 == count || lbuf[i + 1] != kk) atomicMax(&ranges_enc[kk].y, gpos[k] + 1u);
      }
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
                         int bits, uint32_t* tmp, hipStream_t st, uint32_t* w0, uint32_t* w1, const uint32_t* n_dev,
                         bool head_zeroed, uint32_t* fail_flags, bool hist_counted, uint2* ranges_enc) {
  if (n == 0 || bits <= 0) return 0;
  const char* force_env = getenv("GSR_TEST_FORCE_LOOKBACK_TIMEOUT");      // (read per call: a test switches it inside one process)
  const uint32_t force = (force_env && force_env[0] == '1') ? 1u : 0u;
  const uint32_t nblk = (uint32_t)gsr_radix_blocks(n);
  const int subtiles = gsr_radix_subtiles(n);
  const int passes = gsr_radix_passes(bits);
  // tmp = [hist: MAX_PASSES x 256][tickets: 64 words][look-back: passes x chunks x 256]   (gsr_radix_tmp_elems)
  uint32_t* hist = tmp;
  uint32_t* tickets = tmp + GSR_RADIX_HIST_WORDS;
  uint32_t* lookback = tmp + GSR_RADIX_HEAD_WORDS;
  const size_t lb_words = (size_t)passes * nblk * GSR_RADIX_SIZE;
  if (!head_zeroed && !hist_counted) (void)hipMemsetAsync(tmp, 0, GSR_RADIX_HEAD_WORDS * 4, st);
  const bool dual = w0 != nullptr && w1 != nullptr;
  const int reps = hist_counted ? GSR_HIST_REPLICAS : 1;
  // one workgroup per CU: every workgroup ends with one global add per non-zero counter, and adds to ONE address serialise
  // (~15 ns each): 128 / 256 / 512 / 1024 workgroups -> 22 / 17 / 21 / 29 us at 4.4 M keys
  const unsigned hgrid = nblk < 256u ? nblk : 256u;
  if (!hist_counted)     // (else: an earlier kernel counted the digits into the replicas and cleared the look-back table)
    GSR_LAUNCH("radix_hist", k_radix_hist_all, dim3(hgrid), dim3(256), 0, st, (const uint32_t*)k0, n, n_dev, bits, hist,
               lookback, lb_words);
  int cur = 0;
  for (int pass = 0; pass < passes; pass++) {
    const int shift = gsr_radix_shift(bits, pass);
    const uint32_t mask = (1u << gsr_radix_width(bits, pass)) - 1u;
    uint32_t* ki = cur ? k1 : k0;
    uint32_t* vi = cur ? v1 : v0;
    uint32_t* ko = cur ? k0 : k1;
    uint32_t* vo = cur ? v0 : v1;
    const uint32_t* vin = (pass == 0 && vals_iota) ? nullptr : vi;
#define GSR_PASS(D, S, RG)                                                                                             \
  GSR_LAUNCH("radix_pass", (k_radix_pass<D, S, RG>), dim3(nblk), dim3(256), 0, st, (const uint32_t*)ki, vin,           \
             (const uint32_t*)(D ? (cur ? w1 : w0) : nullptr), ko, vo, (uint32_t*)(D ? (cur ? w0 : w1) : nullptr),      \
             (const uint32_t*)tmp, pass, reps, tickets + pass,                                                         \
             lookback + (size_t)pass * nblk * GSR_RADIX_SIZE, n, n_dev, shift, mask, fail_flags,                       \
             (force && pass == passes - 1) ? 1u : 0u, ranges_enc,                                                      \
             (hist_counted && pass + 1 < passes) ? lookback + (size_t)(pass + 1) * nblk * GSR_RADIX_SIZE : (uint32_t*)nullptr, \
             (size_t)nblk * GSR_RADIX_SIZE)
    const bool rg = ranges_enc != nullptr && pass == passes - 1;
    if (dual) {
      if (subtiles == GSR_RADIX_SUBTILES_SMALL) { if (rg) GSR_PASS(true, GSR_RADIX_SUBTILES_SMALL, true); else GSR_PASS(true, GSR_RADIX_SUBTILES_SMALL, false); }
      else { if (rg) GSR_PASS(true, GSR_RADIX_SUBTILES, true); else GSR_PASS(true, GSR_RADIX_SUBTILES, false); }
    } else {
      if (subtiles == GSR_RADIX_SUBTILES_SMALL) { if (rg) GSR_PASS(false, GSR_RADIX_SUBTILES_SMALL, true); else GSR_PASS(false, GSR_RADIX_SUBTILES_SMALL, false); }
      else { if (rg) GSR_PASS(false, GSR_RADIX_SUBTILES, true); else GSR_PASS(false, GSR_RADIX_SUBTILES, false); }
    }
#undef GSR_PASS
    cur ^= 1;
  }
  return cur;
}
