// gsr_common.h - shared host/device definitions for libgsr_hip (gfx950 only, wave64).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

#include "../../include/gsr.h"

#define GSR_TILE 16                 // 16x16 pixel tiles
#define GSR_BLOCK_CULLED 0x80000000u  // top bit of a projection workgroup's instance total (tile-local form): a prefiltered point of it was culled
#define GSR_TILE_PIX 256
#define GSR_WAVE 64
#define GSR_REC_F4 3                // packed splat record = 3 float4 = 48 B
// The record stores the conic as the render kernels consume it: log2(alpha/opacity) = (As dx + Bs dy) dx + Cs dy dy with
// As = -0.5 log2(e) A, Bs = -log2(e) B, Cs = -0.5 log2(e) C, and the cut-off pmin scaled by log2(e) likewise: five VALU
// and a bare v_exp_f32 per (pixel, Gaussian) instead of seven and a multiply.
#define GSR_LOG2E 1.4426950408889634f
#define GSR_IGRAD_F4 3              // per-instance gradient record = 3 float4 (10 used)
// (round 4) One byte per emission slot behind the records of the backward's scratch buffer: 1 = the compositing backward wrote a
// record there, 0 = the instance lies behind its tile's walk (its record would be all zeros: not written, not read).  62 % of the
// records at 1080p / 1 M Gaussians, 76 % at 4K / 5 M, 91 % on a scene of large splats were such zeros - 48 B each way.

// -------------------------------------------------------------------------------------------------
// Packed per-Gaussian splat record written by preprocess and staged through LDS by the render kernels:
//   r0 = (mean2D.x, mean2D.y, conic.A, conic.B)
//   r1 = (conic.C, opacity*aa, pmin = -ln(255 opacity*aa) - margin, rgb.r)      conic and pmin in the scaled form above
//   r2 = (rgb.g, rgb.b, 1/depth, depth)
// r0 + r1 are all a wave needs to reject a Gaussian for its 64 pixels; r2 is read on hits only.
// One 48-B gather per (tile, instance) instead of four separate arrays.
// -------------------------------------------------------------------------------------------------

struct GsrGeomLayout {
  // all offsets in bytes from the start of the geometry state; every array 256-B aligned
  size_t rec;            // float4[3P]
  size_t depth_key;      // u32[P]   fp32 depth bits, 0xFFFFFFFF when culled   (sort key, buffer A)
  size_t order;          // u32[P]   Gaussian ids (sort value, buffer A) -> depth-sorted ids
  size_t key_tmp;        // u32[P]   sort ping-pong B
  size_t val_tmp;        // u32[P]   sort ping-pong B
  size_t tiles_touched;  // u32[P]
  size_t rect;           // ushort4[P] tile rect (min.x, min.y, max.x, max.y)
  size_t bin_rec;        // float4[2P] everything the emit pass needs of a Gaussian in ONE 32-B piece:
                         //            (mx, my, A', B') (C', pmin', rect.xy | rect.zw as two u32)
  size_t offsets;        // u32[P]   inclusive scan of tiles_touched in depth order
  size_t slot_start;     // u32[P]   first emission slot of Gaussian g
  size_t clamped;        // u8[P]    bit c set: colour channel c clamped at 0
  size_t scan_tmp;       // u32[...] block sums for the scans
  size_t radix_tmp;      // u32[...] digit-count tables
  size_t meta;           // u32[16]  [0]=num_rendered [1]=error flags
  size_t total;
};

struct GsrBinLayout {
  size_t key_a, key_b;   // u32[R] tile ids (ping-pong)
  size_t val_a, val_b;   // u32[R] emission slots (ping-pong); the sorted one (slot of each position) is kept for the backward
  size_t gauss_of_slot;  // u32[R] Gaussian id of each emission slot; second payload of the tile sort, ping-pongs with
  size_t point_list;     // u32[R] -> Gaussian ids sorted by (tile, depth, id) end up in ONE of the two (pass parity)
  size_t ranges;         // uint2[tiles]
  size_t culled_any;     // u32[tiles]: == the frame's tag where tile t lost an instance to its depth cut-off (gsr_forward_async_culled;
  //                        a tag per frame instead of a flag: nothing has to be cleared before the projection kernel sets it)
  size_t ranges_enc;     // uint2[tiles]: (~first position, last position + 1) by atomicMax from the tile sort's last pass
  //                        (tile-local binning form), decoded into `ranges` by k_tile_depth_sort
  size_t scan_tmp;
  size_t radix_tmp;
  size_t total;
};

#define GSR_WALK_CLASSES 64
#ifndef GSR_WALK_SUB
#define GSR_WALK_SUB 2              // mantissa bits of a walk class: 2^SUB classes per octave of the walk length (measured: 2 classes
#endif                              // per octave 0.396 ms at C3, 4 per octave 0.381 = what a full sort reaches, profiles/r04_bwd_lpt_ab.txt)
struct GsrImgLayout {
  size_t final_T;        // f32[N]
  size_t n_contrib;      // u32[N]
  size_t walk_cnt;       // u32[GSR_WALK_CLASSES]: tiles per walk class of this frame (k_render_fwd; cleared by the kernel in front of it)
  size_t walk_list;      // u32[GSR_WALK_CLASSES][tiles]: the tiles of every class, in the order their forward workgroups finished
  size_t walk_of_tile;   // u32[tiles]: the walk length itself (entries behind it get zero records from the workgroup of the tile's INDEX)
  size_t total;
};

static inline __host__ __device__ size_t gsr_align(size_t x) { return (x + 255) & ~(size_t)255; }
static inline __host__ __device__ size_t gsr_igrad_bytes(size_t cap) { return gsr_align((cap < 1 ? 1 : cap) * 16 * 3); }
// the validity flags of the gradient records (see GSR_IGRAD_F4): right behind the `cap` records
template <typename T>
static inline __host__ __device__ unsigned char* gsr_igrad_flags(T* igrad, size_t cap) {
  return (unsigned char*)igrad + gsr_igrad_bytes(cap);
}

// sort/scan tuning shared by the sizing code and the kernels
#define GSR_SCAN_ITEMS 8                         // per thread
#define GSR_SCAN_CHUNK (256 * GSR_SCAN_ITEMS)    // per block
#define GSR_RADIX_BITS 8
#define GSR_RADIX_SIZE 256
#define GSR_RADIX_SUBTILES 16                    // sub-tiles of 256 keys per workgroup (chunk = 4096 keys) for large arrays
#define GSR_RADIX_SUBTILES_SMALL 8               // chunk = 2048 keys below GSR_RADIX_SMALL_N keys: twice the workgroups, so a
#ifndef GSR_RADIX_SMALL_N
#define GSR_RADIX_SMALL_N (2u << 20)             // 1 M-key depth sort still puts two workgroups on every CU
#endif
#define GSR_RADIX_CHUNK (256 * GSR_RADIX_SUBTILES)

static inline size_t gsr_scan_tmp_elems(size_t n) {
  // block sums for level 0, level 1, ... (recursive)
  size_t tot = 0;
  while (n > 1) {
    n = (n + GSR_SCAN_CHUNK - 1) / GSR_SCAN_CHUNK;
    tot += gsr_align(n * 4) / 4;
    if (n == 1) break;
  }
  return tot + 64;
}
static inline int gsr_radix_subtiles(size_t n) { return n < GSR_RADIX_SMALL_N ? GSR_RADIX_SUBTILES_SMALL : GSR_RADIX_SUBTILES; }
static inline size_t gsr_radix_blocks(size_t n) {
  const size_t chunk = (size_t)256 * gsr_radix_subtiles(n);
  return (n + chunk - 1) / chunk;
}
#define GSR_RADIX_MAX_PASSES 4
// `bits` key bits are split into ceil(bits / 8) passes of (nearly) EQUAL width, low bits first: 13 tile-id bits sort as 7 + 6
// (128 / 64 bins: per-digit runs of 32 / 64 keys leave a chunk as 128-B / 256-B pieces) rather than 8 + 5; 32 bits as 4 x 8.
static inline __host__ __device__ int gsr_radix_passes(int bits) { return (bits + GSR_RADIX_BITS - 1) / GSR_RADIX_BITS; }
static inline __host__ __device__ int gsr_radix_width(int bits, int pass) {
  const int p = gsr_radix_passes(bits), q = bits / p, r = bits - q * p;     // the first r passes get q + 1 bits
  return q + (pass < r ? 1 : 0);
}
static inline __host__ __device__ int gsr_radix_shift(int bits, int pass) {
  const int p = gsr_radix_passes(bits), q = bits / p, r = bits - q * p;
  return pass * q + (pass < r ? pass : r);
}
// Head of a sort's scratch: digit histograms of every pass + pass tickets, then GSR_HIST_REPLICAS - 1 further copies of the
// histogram block.  A sort whose histograms are counted by MANY workgroups (round 4: the tile sort's, by k_emit_instances, 3900
// workgroups at 1 M Gaussians) spreads its adds over the replicas by workgroup index - adds to ONE address serialise at ~15 ns -
// and k_radix_pass sums the replicas; the classic path (k_radix_hist_all, <= 256 workgroups) fills replica 0 only.
#ifndef GSR_HIST_REPLICAS
#define GSR_HIST_REPLICAS 8
#endif
#define GSR_RADIX_HIST_WORDS (GSR_RADIX_MAX_PASSES * GSR_RADIX_SIZE)
#define GSR_RADIX_HEAD_WORDS (GSR_RADIX_HIST_WORDS + 64 + (GSR_HIST_REPLICAS - 1) * GSR_RADIX_HIST_WORDS)
static inline __host__ __device__ size_t gsr_hist_replica(int r) {     // word offset of histogram replica r inside the head
  return r == 0 ? (size_t)0 : (size_t)GSR_RADIX_HIST_WORDS + 64 + (size_t)(r - 1) * GSR_RADIX_HIST_WORDS;
}
static inline size_t gsr_radix_tmp_elems(size_t n) {
  // [histograms + tickets + histogram replicas][look-back words: passes x chunks x 256]
  return GSR_RADIX_HEAD_WORDS + (size_t)GSR_RADIX_MAX_PASSES * gsr_radix_blocks(n) * GSR_RADIX_SIZE;
}

static inline GsrGeomLayout gsr_geom_layout(size_t P) {
  GsrGeomLayout L;
  size_t o = 0;
  if (P == 0) P = 1;
  L.rec = o;           o += gsr_align(P * 48);
  L.depth_key = o;     o += gsr_align(P * 4);
  L.order = o;         o += gsr_align(P * 4);
  L.key_tmp = o;       o += gsr_align(P * 4);
  L.val_tmp = o;       o += gsr_align(P * 4);
  L.tiles_touched = o; o += gsr_align(P * 4);
  L.rect = o;          o += gsr_align(P * 8);
  L.bin_rec = o;       o += gsr_align(P * 32);
  L.offsets = o;       o += gsr_align(P * 4);
  L.slot_start = o;    o += gsr_align(P * 4);
  L.clamped = o;       o += gsr_align(P);
  L.scan_tmp = o;      o += gsr_align(gsr_scan_tmp_elems(P) * 4);
  L.meta = o;          o += 256;            // meta and the head of radix_tmp (histograms, tickets) are cleared by ONE memset
  L.radix_tmp = o;     o += gsr_align(gsr_radix_tmp_elems(P) * 4);
  L.total = o;
  return L;
}

static inline GsrBinLayout gsr_bin_layout(size_t R, size_t tiles) {
  GsrBinLayout L;
  size_t o = 0;
  if (R == 0) R = 1;
  L.key_a = o;         o += gsr_align(R * 4);
  L.key_b = o;         o += gsr_align(R * 4);
  L.val_a = o;         o += gsr_align(R * 4);
  L.val_b = o;         o += gsr_align(R * 4);
  L.gauss_of_slot = o; o += gsr_align(R * 4);
  L.point_list = o;    o += gsr_align(R * 4);
  L.ranges = o;        o += gsr_align(tiles * 8);
  L.ranges_enc = o;    o += gsr_align(tiles * 8);
  L.culled_any = o;    o += gsr_align(tiles * 4);
  L.scan_tmp = o;      o += gsr_align(gsr_scan_tmp_elems(R) * 4);
  L.radix_tmp = o;     o += gsr_align(gsr_radix_tmp_elems(R) * 4);
  L.total = o;
  return L;
}

static inline GsrImgLayout gsr_img_layout(int W, int H) {
  GsrImgLayout L;
  const size_t N = (size_t)W * (size_t)H;
  const size_t tiles = (size_t)((W + GSR_TILE - 1) / GSR_TILE) * (size_t)((H + GSR_TILE - 1) / GSR_TILE);
  size_t o = 0;
  L.final_T = o;   o += gsr_align(N * 4);
  L.n_contrib = o; o += gsr_align(N * 4);
  L.walk_cnt = o;  o += gsr_align(GSR_WALK_CLASSES * 4);
  L.walk_list = o; o += gsr_align(GSR_WALK_CLASSES * tiles * 4);
  L.walk_of_tile = o; o += gsr_align(tiles * 4);
  L.total = o;
  return L;
}
// Walk class of a tile (round 4): a coarse logarithm of the number of list entries its backward walks (the deepest contributor of
// any of its pixels) - exponent and GSR_WALK_SUB leading mantissa bits.  k_render_bwd_tile takes the tiles class by class, longest
// walks first; inside a class the tiles keep (roughly) their natural order, which is what keeps neighbouring tiles' record gathers
// in the same L2.
__host__ __device__ static inline int gsr_walk_class(uint32_t w) {
  if (GSR_WALK_SUB == 2 && w > 65535u) w = 65535u;
  if (w < (1u << GSR_WALK_SUB)) return (int)w;
#if defined(__HIP_DEVICE_COMPILE__)
  const int e = 31 - __clz((int)w);
#else
  const int e = 31 - __builtin_clz(w);
#endif
  return (e << GSR_WALK_SUB) + (int)((w >> (e - GSR_WALK_SUB)) & ((1u << GSR_WALK_SUB) - 1u));
}

// optimizer step folded into k_preprocess_bwd (gsr_backward_adam): kernel-side form of gsr_fused_adam
struct GsrAdamArgs {
  float* p[6];      // xyz, f_dc, f_rest, opacity, scaling, rotation (the rasterizer's own input arrays, updated in place)
  float* m[6];
  float* v[6];
  float lr[6], step_size[6], inv_bc2_sqrt[6];
  float beta1, beta2, omb1, omb2, eps;
  const float* dyn;   // optional device copy of (lr, step_size, inv_bc2_sqrt) that overrides the three arrays above (HIP-graph replay)
};

// the per-step factors as the kernel should use them: from the launch arguments, or from device memory when the caller keeps
// them there (gsr_fused_adam.dynamic).  Wave-uniform loads of 18 floats.
__device__ __forceinline__ GsrAdamArgs gsr_adam_resolve(const GsrAdamArgs& in) {
  GsrAdamArgs A = in;
  if (in.dyn) {
#pragma unroll
    for (int i = 0; i < 6; i++) {
      A.lr[i] = in.dyn[i];
      A.step_size[i] = in.dyn[6 + i];
      A.inv_bc2_sqrt[i] = in.dyn[12 + i];
    }
  }
  return A;
}

// -------------------------------------------------------------------------------------------------
// host-side launch helpers (api.hip owns the definitions)
// -------------------------------------------------------------------------------------------------
void gsr_set_error(const char* fmt, ...);
int gsr_check(hipError_t e, const char* what);
void gsr_prof_begin(const char* name, hipStream_t st);
void gsr_prof_end(hipStream_t st);
extern int g_gsr_profile_on;

// A launch that fails (bad configuration, too much LDS ...) is noted with ITS stage name right away (hipGetLastError does not
// wait for the device); the entry point's final gsr_launch_status() then reports the first failing stage, not the last one.
void gsr_note_launch_failure(const char* stage, hipError_t e);
int gsr_launch_status(const char* what);
#define GSR_LAUNCH(name, kern, grid, block, shmem, st, ...)                    \
  do {                                                                         \
    if (g_gsr_profile_on) gsr_prof_begin(name, st);                            \
    hipLaunchKernelGGL(kern, grid, block, shmem, st, __VA_ARGS__);             \
    if (g_gsr_profile_on) gsr_prof_end(st);                                    \
    {                                                                          \
      const hipError_t gsr_e_ = hipGetLastError();                             \
      if (gsr_e_ != hipSuccess) gsr_note_launch_failure(name, gsr_e_);         \
    }                                                                          \
  } while (0)

// sort_scan.hip
// Exclusive (inclusive=0) or inclusive scan of n u32 values: out[i] = scan(src[idx ? idx[i] : i]).
void gsr_scan_u32(const uint32_t* src, const uint32_t* idx, uint32_t* out, size_t n, int inclusive,
                  uint32_t* tmp, hipStream_t st);
// Stable LSD radix sort of (key,value) pairs on key bits [0, bits).  vals_in == nullptr means value = index.
// Buffers ping-pong between (k0,v0) and (k1,v1); returns 0 if the result is in (k0,v0), 1 if in (k1,v1).
// w0 / w1 (both or neither): a second 32-bit payload, input in w0, ping-ponging with w1 like the values.
// n_dev (optional, DEVICE pointer to a 64-bit count as two u32 words): the number of keys actually present is
// min(*n_dev, n) and is read by the kernels themselves - `n` then only sizes the grids and the tables (the capacity of a
// caller that does not know the count on the host: gsr_forward_async).
int gsr_radix_sort_pairs(uint32_t* k0, uint32_t* v0, uint32_t* k1, uint32_t* v1, bool vals_iota, size_t n,
                         int bits, uint32_t* tmp, hipStream_t st, uint32_t* w0 = nullptr, uint32_t* w1 = nullptr,
                         const uint32_t* n_dev = nullptr, bool head_zeroed = false, uint32_t* fail_flags = nullptr,
                         bool hist_counted = false, uint2* ranges_enc = nullptr);
// hist_counted: an earlier kernel has counted every pass's digits into the head's replicas and zeroed the look-back table
// (k_emit_instances for the tile sort): no k_radix_hist_all launch.  ranges_enc: the LAST pass also leaves, per key value t
// (= tile id), (~first, last + 1) of its run of sorted positions in ranges_enc[t] by atomicMax (zero-initialised by the caller).
// fail_flags: the frame's first status word (geometry state meta[0], handed to the host as status word 0): a look-back wait that
// times out (a broken hand-off protocol: the pass then goes on with a WRONG base, the grid drains, the frame is mis-sorted) ORs
// GSR_STATUS_SORT_TIMEOUT into it, so the failure reaches the caller instead of staying a mark in device memory.
#define GSR_STATUS_SORT_TIMEOUT 1u
// (round 4) tile-list truncation by a per-tile depth cut-off (gsr_forward_async_culled): a pixel that reached the END of a truncated
// list without saturating might have blended a culled Gaussian - the frame is not the frame.  The compositing kernel raises this
// bit in meta[0] (every backward kernel then turns the step into a no-op, like a frame beyond the capacity: gsr_overflowed) and
// stores 1 into word 6 of the caller's status slot.
#define GSR_STATUS_CULL_MISS 2u
// head_zeroed: the caller guarantees that the first GSR_RADIX_HEAD_WORDS words of `tmp` are zero when the sort's first kernel
// starts (an earlier kernel of the same stream cleared them); otherwise the sort enqueues a memset of its own.

// -------------------------------------------------------------------------------------------------
// device helpers
// -------------------------------------------------------------------------------------------------
#ifdef __HIPCC__
__device__ __forceinline__ int gsr_lane() { return threadIdx.x & 63; }

// Number of instances the binning / compositing stages work on: min(num_rendered, capacity).  num_rendered lives on the
// DEVICE (geometry state `meta[2..3]`, 64-bit, written by k_sum_tiles); `cap` is what the caller's binning state was sized
// for (= num_rendered itself on the blocking path, an upper estimate on the non-blocking one).  n_dev == nullptr: cap.
__device__ __forceinline__ uint32_t gsr_eff_n(const uint32_t* __restrict__ n_dev, uint32_t cap) {
  if (!n_dev) return cap;
  const uint32_t lo = n_dev[0], hi = n_dev[1];
  return (hi != 0u || lo > cap) ? cap : lo;
}

// Whether this frame's gradient records carry validity flags (GSR_IGRAD_F4): every backward kernel decides it from the frame's
// instance count, which all of them read anyway.  Small frames keep the round-3 form (zero records written and read): there the
// projection backward is one partial round of workgroups whose length is a thread's chain of dependent memory round trips, and the
// flags are one more link of it (100 k Gaussians / 1.3 M instances: +7 .. 13 us on a 61 us kernel, nothing gained elsewhere).
#ifndef GSR_FLAGS_MIN_R
#define GSR_FLAGS_MIN_R 2500000u
#endif
// (the threshold travels as a launch argument: tests set it to 0 / ~0 through gsr_debug_set_flags_min_r to force either form)
extern unsigned g_gsr_flags_min_r;
__device__ __forceinline__ bool gsr_flags_on(const uint32_t* __restrict__ n_dev, uint32_t cap, uint32_t min_r) {
  return gsr_eff_n(n_dev, cap) >= min_r;
}

// A frame of the non-blocking forward that had MORE instances than its binning state could hold was composited from a truncated
// list: its image is not the frame's image, so nothing may be learnt from it.  Every backward kernel tests this (one scalar
// load) and turns the whole step into a no-op: no gradient records, zero gradients, no optimizer update, no statistics.
__device__ __forceinline__ bool gsr_overflowed(const uint32_t* __restrict__ n_dev, uint32_t cap) {
  // (n_dev = meta + 2 wherever it is not null: meta[0] holds the frame's status flags - a list truncated by the depth cut-off
  // that turned out to be too short makes the frame as unusable as one beyond the capacity)
  return n_dev != nullptr && (n_dev[1] != 0u || n_dev[0] > cap || (*(n_dev - 2) & GSR_STATUS_CULL_MISS) != 0u);
}

// One Adam update, shared by k_adam (adam.hip) and the step folded into k_preprocess_bwd (preprocess.hip): every rounding is
// spelled out (explicit fma / separate operations) so that both kernels produce the same bits whatever the compiler would
// contract in their different surroundings.  ADAM = 1: torch.optim.Adam (exp_avg.lerp_, bias-corrected step, eps after the
// corrected sqrt); ADAM = 2: the sparse optimizer of the reference's accelerated path (no bias correction).
template <int ADAM>
__device__ __forceinline__ void adam_elem(float& p, float& m, float& v, const float g, const GsrAdamArgs& A, const int grp) {
  const float gg = __fmul_rn(__fmul_rn(A.omb2, g), g);
  if (ADAM == 2) {
    m = __builtin_fmaf(A.beta1, m, __fmul_rn(A.omb1, g));
    v = __builtin_fmaf(A.beta2, v, gg);
    p = __fadd_rn(p, __fdiv_rn(__fmul_rn(-A.lr[grp], m), __fadd_rn(__fsqrt_rn(v), A.eps)));
  } else {
    m = __builtin_fmaf(__fsub_rn(g, m), A.omb1, m);
    v = __builtin_fmaf(A.beta2, v, gg);
    const float denom = __builtin_fmaf(__fsqrt_rn(v), A.inv_bc2_sqrt[grp], A.eps);
    p = __builtin_fmaf(-A.step_size[grp], __fdiv_rn(m, denom), p);
  }
}

// add_densification_stats + max_radii2D of one Gaussian (reference scene/gaussian_model.py:431-433, train.py:159); shared by
// k_densify_stats and the copy folded into k_preprocess_bwd, roundings spelled out so both give the same bits
__device__ __forceinline__ void gsr_densify_stats_update(float gx, float gy, int radius, float* accum, float* denom,
                                                         float* max_radii) {
  *accum = __fadd_rn(*accum, __fsqrt_rn(__builtin_fmaf(gx, gx, __fmul_rn(gy, gy))));
  *denom = __fadd_rn(*denom, 1.0f);
  *max_radii = fmaxf(*max_radii, (float)radius);
}

// 16-B streaming (non-temporal) global accesses for data that passes through once
typedef float gsr_f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ gsr_f4 gsr_ld_stream(const float* p) {
  return __builtin_nontemporal_load(reinterpret_cast<const gsr_f4*>(p));
}
__device__ __forceinline__ void gsr_st_stream(float* p, gsr_f4 v) {
  __builtin_nontemporal_store(v, reinterpret_cast<gsr_f4*>(p));
}
__device__ __forceinline__ float4 gsr_ld_stream4(const float4* p) {
  const gsr_f4 v = __builtin_nontemporal_load(reinterpret_cast<const gsr_f4*>(p));
  return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void gsr_st_stream4(float4* p, const float4& v) {
  __builtin_nontemporal_store(gsr_f4{v.x, v.y, v.z, v.w}, reinterpret_cast<gsr_f4*>(p));
}

// Exact tile culling shared by preprocess (count) and emit (write).  Pixels that can blend a Gaussian satisfy
// Q(d) = A dx^2 + 2B dx dy + C dy^2 <= q (q = -2 pmin, i.e. power >= pmin).  The part of that ellipse inside the horizontal
// band of tile row ty (pixel centres 16ty .. 16ty+15) is convex, so the tile columns it reaches form ONE interval: its
// x-extent is [x_lo(y*), x_hi(y**)] with x_hi/lo(y) = (-B y +- sqrt(A q - det y^2)) / A evaluated at the band-clamped
// heights of the ellipse's right-/left-most points.  Returns the half-open column interval clipped to [cx0, cx1) packed as
// lo | hi << 16 (lo >= hi: empty).  __noinline__: ONE compiled body, so the count and the emission agree bit for bit.
// Inputs are the STORED record fields (conic and cut-off pre-scaled for the render kernels, see GSR_K*): both callers pass
// the same bits and the un-scaling happens in here, once.
__device__ __noinline__ uint32_t gsr_row_interval(float mx, float my, float As, float Bs, float Cs, float pmins, int ty,
                                                  int cx0, int cx1) {
  const float A = As * (-2.0f / GSR_LOG2E), B = Bs * (-1.0f / GSR_LOG2E), C = Cs * (-2.0f / GSR_LOG2E);
  const float q = pmins * (-2.0f / GSR_LOG2E);
  // (hardware rcp / sqrt, 1 ulp: every use below carries a slack orders of magnitude larger, and both callers run this one
  // compiled body, so count and emission still agree bit for bit)
  const float det = A * C - B * B;
  const float rdet = __builtin_amdgcn_rcpf(det), rA = __builtin_amdgcn_rcpf(A), rC = __builtin_amdgcn_rcpf(C);
  const float hy = __builtin_amdgcn_sqrtf(fmaxf(0.f, q * A * rdet)) * 1.0001f + 0.01f;   // ellipse half-height (+ slack)
  const float yl = (float)(ty * GSR_TILE) - my, yh = yl + (float)(GSR_TILE - 1);
  const float bl = fmaxf(yl, -hy), bh = fminf(yh, hy);
  if (bl > bh) return 0u;                                                   // band misses the ellipse
  const float yr = -B * __builtin_amdgcn_sqrtf(fmaxf(0.f, q * rC * rdet));  // height of the right-most point (left-most: -yr)
  const float y1 = fminf(bh, fmaxf(bl, yr)), y2 = fminf(bh, fmaxf(bl, -yr));
  const float qa = q * 1.0002f + 0.002f;                                   // conservative slack >> fp32 error
  const float xhi = (-B * y1 + __builtin_amdgcn_sqrtf(fmaxf(0.f, A * qa - det * y1 * y1))) * rA + 0.01f;
  const float xlo = (-B * y2 - __builtin_amdgcn_sqrtf(fmaxf(0.f, A * qa - det * y2 * y2))) * rA - 0.01f;
  // tile column t holds pixel centres 16t .. 16t+15 (relative to the mean: subtract mx)
  const float lim = 1.0e9f;
  int lo = (int)ceilf(fmaxf(-lim, (mx + xlo - (float)(GSR_TILE - 1)) / GSR_TILE));
  int hi = (int)floorf(fminf(lim, (mx + xhi) / GSR_TILE)) + 1;
  lo = max(lo, cx0);
  hi = min(hi, cx1);
  if (hi <= lo) return 0u;
  return (uint32_t)lo | ((uint32_t)hi << 16);
}

#endif
