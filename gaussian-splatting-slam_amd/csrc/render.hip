// render.hip - 16x16-tile alpha compositing, forward (K6) and backward (K7).
//
// MI355X mapping: one 256-thread workgroup (4 wave64) per tile; wave w owns the 8x8 pixel quadrant w, lane l
// the pixel (l&7, l>>3) of it, so a Gaussian that misses a quadrant is rejected for 64 pixels by ONE
// wave-uniform ballot + branch.  Batches of packed 48-B splat records (256 forward, 128 backward) are staged through LDS
// (one coalesced gather per record) and read back as wave-uniform broadcasts (conflict-free) through a VGPR base with
// immediate offsets, four entries per trip, prefetched two entries ahead.  Power is evaluated in the exp2 domain from the
// pre-scaled conic of the record (5 VALU, bare v_exp_f32).  A conservative wave-level test `power >= log2(1/(255 opacity)) -
// margin` rejects a Gaussian for a whole quadrant before the exp; survivors take the exact published test, so results are
// unchanged.  Both kernels are bound by VALU issue (profiles/README.md): everything here is about instructions per hit.
//
// Backward: per-pixel back-to-front replay as published (T recovered by division), but NO global atomics: each wave
// reduces its 64 pixels' contributions by recursive halving (v_permlane32/16_swap + DPP, 24 VALU for nine sums) into a
// private LDS slab, the block adds the four slabs in a fixed order and writes one 48-B gradient record per
// (tile, instance) at the instance's emission slot.  The per-Gaussian sum over instances happens in k_preprocess_bwd
// (deterministic).
// Record = (sum h dx, sum h dy, sum h dx^2, sum h dx dy) (sum h dy^2, sum h = dL/dopacity_eff, d_r, d_g) (d_b, d_invdepth, -, -)
// with h = dL/dpower / opacity_eff (the per-pixel dL/dopacity_eff) and d = mean - pixel: raw moments; k_preprocess_bwd turns
// their per-Gaussian totals, times opacity_eff, into dL/dmean2D and dL/dconic.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "gsr_common.h"

#define ALPHA_MIN (1.0f / 255.0f)

// log2 of the Gaussian falloff at d = mean - pixel from the pre-scaled conic of the record (gsr_common.h): 5 VALU
__device__ __forceinline__ float gsr_power2(const float4& r0, const float4& r1, float dx, float dy) {
  const float t = __builtin_fmaf(r0.w, dy, r0.z * dx);
  return __builtin_fmaf(r1.x * dy, dy, t * dx);
}

// ---------------------------------------------------------------------------------------------------------------
// Ten full-wave sums by recursive halving: at every stage a lane keeps half of its values and hands the other half to
// its partner, so the work per stage halves (10 -> 5 -> 3 -> 2 registers) instead of staying at ten DPP adds per stage.
//   stage A  lanes l <-> l^32   v_permlane32_swap_b32 (gfx950): x'=[x.lo,y.lo] y'=[x.hi,y.hi]; x'+y' = [sum x | sum y]
//   stage B  rows  r <-> r^1    v_permlane16_swap_b32 (gfx950): same idea on 16-lane rows
//   stage C  lanes l <-> l^8    DPP row_ror:8 with a lane-bit select
//   stage D-F                   quad_perm xor 1, xor 2, row_half_mirror: every lane of an 8-lane octet gets the octet sum
// 27 VALU instead of 60.  Result: u0 in octet o = lane>>3 holds the total of value OCTET_VALUE[o] = {0,4,2,6,1,5,3,7}[o];
// u1 holds the total of v8 in lanes 0..15 and of v9 in lanes 32..47.  The summation tree is fixed -> deterministic.
// ---------------------------------------------------------------------------------------------------------------
typedef unsigned gsr_u2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float swap32_add(float x, float y) {
  const gsr_u2 r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(y), false, false);
  return __uint_as_float(r.x) + __uint_as_float(r.y);
}
__device__ __forceinline__ float swap16_add(float x, float y) {
  const gsr_u2 r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(y), false, false);
  return __uint_as_float(r.x) + __uint_as_float(r.y);
}
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ float dpp_get(float v) {
  // old = 0 + bound_ctrl: lanes without a source (and rows masked off) read 0.0, so `x + dpp_get(x)` folds into ONE
  // v_add_f32_dpp
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xf, true));
}

__device__ __forceinline__ void wave_sum10_halving(float v0, float v1, float v2, float v3, float v4, float v5, float v6,
                                                   float v7, float v8, float v9, bool lane_bit3, float& u0, float& u1) {
  // A: 10 -> 5
  const float r0 = swap32_add(v0, v1), r1 = swap32_add(v2, v3), r2 = swap32_add(v4, v5), r3 = swap32_add(v6, v7),
              r4 = swap32_add(v8, v9);
  // B: 5 -> 3   (rows: s0 = [v0,v2,v1,v3], s1 = [v4,v6,v5,v7], s2 = [v8,0,v9,0])
  const float s0 = swap16_add(r0, r1), s1 = swap16_add(r2, r3), s2 = swap16_add(r4, 0.0f);
  // C: 3 -> 2   (lanes with bit 3 clear keep s0, the others keep s1; each sends what it does not keep)
  const float keep = lane_bit3 ? s1 : s0;
  const float send = lane_bit3 ? s0 : s1;
  float a = keep + dpp_get<0x128>(send);   // row_ror:8
  float b = s2 + dpp_get<0x128>(s2);
  // D-F: sum the 8 lanes of each octet
  a += dpp_get<0xB1>(a);                   // quad_perm:[1,0,3,2]
  b += dpp_get<0xB1>(b);
  a += dpp_get<0x4E>(a);                   // quad_perm:[2,3,0,1]
  b += dpp_get<0x4E>(b);
  a += dpp_get<0x141>(a);                  // row_half_mirror
  b += dpp_get<0x141>(b);
  asm volatile("" : "+v"(a), "+v"(b));     // (totals formed here, not sunk into the caller's lane-predicated store blocks)
  u0 = a;
  u1 = b;
}

// Nine sums (no inverse-depth gradient, the usual training step): v0..v7 by the same halving tree (18 VALU), and the lone
// ninth value by a plain DPP butterfly (6 VALU): quad xor 1, xor 2, row_half_mirror, row_mirror give every lane its row
// sum; row_bcast:15 adds each row's predecessor, row_bcast:31 then adds rows 0+1 into row 3.  u1 = total of v8 in lanes
// 48..63 (other rows hold partial sums nobody reads).  Fixed tree -> deterministic.
__device__ __forceinline__ void wave_sum9_halving(float v0, float v1, float v2, float v3, float v4, float v5, float v6,
                                                  float v7, float v8, bool lane_bit3, float& u0, float& u1) {
  const float r0 = swap32_add(v0, v1), r1 = swap32_add(v2, v3), r2 = swap32_add(v4, v5), r3 = swap32_add(v6, v7);
  const float s0 = swap16_add(r0, r1), s1 = swap16_add(r2, r3);
  const float keep = lane_bit3 ? s1 : s0;
  const float send = lane_bit3 ? s0 : s1;
  float a = keep + dpp_get<0x128>(send);   // row_ror:8
  float b = v8 + dpp_get<0xB1>(v8);        // quad_perm:[1,0,3,2]
  a += dpp_get<0xB1>(a);
  b += dpp_get<0x4E>(b);                   // quad_perm:[2,3,0,1]
  a += dpp_get<0x4E>(a);
  b += dpp_get<0x141>(b);                  // row_half_mirror
  a += dpp_get<0x141>(a);
  b += dpp_get<0x140>(b);                  // row_mirror
  b += dpp_get<0x142>(b);                  // row_bcast:15: row k += row k-1  (row 3 = r3 + r2, row 1 = r1 + r0)
  b += dpp_get<0x143>(b);                  // row_bcast:31: row 3 += lane 31 = r1 + r0   (rows 0..2: don't care)
  asm volatile("" : "+v"(a), "+v"(b));     // (see wave_sum10_halving)
  u0 = a;
  u1 = b;
}

// ---------------------------------------------------------------------------------------------------------------
// (round 4) The same trees on values kept as register PAIRS (P0 = (v0, v1), P1 = (v2, v3), P2 = (v4, v5), P3 = (v6, v7)): the
// swap partners are chosen so that the adds behind each swap stage act on whole pairs - v_pk_add_f32, one instruction for
// two sums (packed fp32 runs at the rate of plain fp32 on this part) - 3 adds instead of 6.  Which value shares a register with
// which does not enter any value's own summation tree (lanes l + l^32, rows r + r^1, lanes l + l^8, then the octet), so every
// total has the bits of the unpacked tree; only the octet a total ends up in differs: octet o holds value
// {0,1,4,5,2,3,6,7}[o] = o with bits 1 and 2 exchanged (GSR_OCTET_VALUE_PK).
// ---------------------------------------------------------------------------------------------------------------
typedef float gsr_f2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ gsr_f2 swap32_add_pk(gsr_f2 p, gsr_f2 q) {   // -> ([p.x | q.x], [p.y | q.y]) half-wave sums
  const gsr_u2 a = __builtin_amdgcn_permlane32_swap(__float_as_uint(p.x), __float_as_uint(q.x), false, false);
  const gsr_u2 b = __builtin_amdgcn_permlane32_swap(__float_as_uint(p.y), __float_as_uint(q.y), false, false);
  const gsr_f2 lo = {__uint_as_float(a.x), __uint_as_float(b.x)}, hi = {__uint_as_float(a.y), __uint_as_float(b.y)};
  return lo + hi;
}
__device__ __forceinline__ gsr_f2 swap16_add_pk(gsr_f2 p, gsr_f2 q) {
  const gsr_u2 a = __builtin_amdgcn_permlane16_swap(__float_as_uint(p.x), __float_as_uint(q.x), false, false);
  const gsr_u2 b = __builtin_amdgcn_permlane16_swap(__float_as_uint(p.y), __float_as_uint(q.y), false, false);
  const gsr_f2 lo = {__uint_as_float(a.x), __uint_as_float(b.x)}, hi = {__uint_as_float(a.y), __uint_as_float(b.y)};
  return lo + hi;
}
#define GSR_OCTET_VALUE_PK(o) (((o) & 1) | (((o) & 2) << 1) | (((o) & 4) >> 1))

__device__ __forceinline__ void wave_sum9_halving_pk(gsr_f2 P0, gsr_f2 P1, gsr_f2 P2, gsr_f2 P3, float v8, bool lane_bit3,
                                                     float& u0, float& u1) {
  const gsr_f2 R0 = swap32_add_pk(P0, P1), R1 = swap32_add_pk(P2, P3);   // R0 = ([v0|v2], [v1|v3]), R1 = ([v4|v6], [v5|v7])
  const gsr_f2 S = swap16_add_pk(R0, R1);                                // rows: S.x = [v0,v4,v2,v6], S.y = [v1,v5,v3,v7]
  const float keep = lane_bit3 ? S.y : S.x;
  const float send = lane_bit3 ? S.x : S.y;
  float a = keep + dpp_get<0x128>(send);   // row_ror:8
  float b = v8 + dpp_get<0xB1>(v8);        // the lone ninth value: the same DPP butterfly as wave_sum9_halving
  a += dpp_get<0xB1>(a);
  b += dpp_get<0x4E>(b);
  a += dpp_get<0x4E>(a);
  b += dpp_get<0x141>(b);
  a += dpp_get<0x141>(a);
  b += dpp_get<0x140>(b);
  b += dpp_get<0x142>(b);
  b += dpp_get<0x143>(b);
  // (the totals are formed HERE: left to itself the compiler sinks the last add of each chain into the caller's lane-predicated
  // store block and pays a v_mov_b32_dpp + v_add_f32 for what is one v_add_f32_dpp)
  asm volatile("" : "+v"(a), "+v"(b));
  u0 = a;
  u1 = b;
}

__device__ __forceinline__ void wave_sum10_halving_pk(gsr_f2 P0, gsr_f2 P1, gsr_f2 P2, gsr_f2 P3, gsr_f2 P4, bool lane_bit3,
                                                      float& u0, float& u1) {
  const gsr_f2 R0 = swap32_add_pk(P0, P1), R1 = swap32_add_pk(P2, P3);
  const float r4 = swap32_add(P4.x, P4.y);                               // [v8 | v9]
  const gsr_f2 S = swap16_add_pk(R0, R1);
  const float s2 = swap16_add(r4, 0.0f);                                 // rows [v8, 0, v9, 0]
  const float keep = lane_bit3 ? S.y : S.x;
  const float send = lane_bit3 ? S.x : S.y;
  float a = keep + dpp_get<0x128>(send);
  float b = s2 + dpp_get<0x128>(s2);
  a += dpp_get<0xB1>(a);
  b += dpp_get<0xB1>(b);
  a += dpp_get<0x4E>(a);
  b += dpp_get<0x4E>(b);
  a += dpp_get<0x141>(a);
  b += dpp_get<0x141>(b);
  asm volatile("" : "+v"(a), "+v"(b));   // (totals formed here, not sunk into the caller's store blocks: see the nine-value tree)
  u0 = a;
  u1 = b;
}

// test hook (tests/test_parity_gpu.py::test_wave_reduction_primitive): in[10][64] -> out[0..9] (ten-value tree) and
// out[10..18] (nine-value tree on rows 0..8) through the same store patterns the render backward uses
__global__ void k_debug_wave_reduce(const float* __restrict__ in, float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  float u0, u1;
  wave_sum10_halving(in[0 * 64 + lane], in[1 * 64 + lane], in[2 * 64 + lane], in[3 * 64 + lane], in[4 * 64 + lane],
                     in[5 * 64 + lane], in[6 * 64 + lane], in[7 * 64 + lane], in[8 * 64 + lane], in[9 * 64 + lane],
                     (lane & 8) != 0, u0, u1);
  const int o = lane >> 3;
  const int val = ((o & 1) << 2) | (o & 2) | ((o & 4) >> 2);   // {0,4,2,6,1,5,3,7}[o]
  if ((lane & 7) == 0) out[val] = u0;
  if ((lane & 31) == 0) out[8 + (lane >> 5)] = u1;
  // nine-value variant on the first nine rows -> out[10..18]
  wave_sum9_halving(in[0 * 64 + lane], in[1 * 64 + lane], in[2 * 64 + lane], in[3 * 64 + lane], in[4 * 64 + lane],
                    in[5 * 64 + lane], in[6 * 64 + lane], in[7 * 64 + lane], in[8 * 64 + lane], (lane & 8) != 0, u0, u1);
  if ((lane & 7) == 0) out[10 + val] = u0;
  if (lane == 63) out[18] = u1;
}

extern "C" int gsr_debug_wave_reduce(const float* in640, float* out20, void* stream) {
  hipLaunchKernelGGL(k_debug_wave_reduce, dim3(1), dim3(64), 0, (hipStream_t)stream, in640, out20);
  return gsr_launch_status("debug wave reduce");
}

// the packed-pair trees through the store pattern of k_render_bwd_tile<., MASK = true>: same in / out layout as above
__global__ void k_debug_wave_reduce_pk(const float* __restrict__ in, float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  float u0, u1;
  auto P = [&](int i) { return gsr_f2{in[(2 * i) * 64 + lane], in[(2 * i + 1) * 64 + lane]}; };
  wave_sum10_halving_pk(P(0), P(1), P(2), P(3), P(4), (lane & 8) != 0, u0, u1);
  const int val = GSR_OCTET_VALUE_PK(lane >> 3);
  if ((lane & 7) == 0) out[val] = u0;
  if ((lane & 31) == 0) out[8 + (lane >> 5)] = u1;
  wave_sum9_halving_pk(P(0), P(1), P(2), P(3), in[8 * 64 + lane], (lane & 8) != 0, u0, u1);
  if ((lane & 7) == 0) out[10 + val] = u0;
  if (lane == 63) out[18] = u1;
}

extern "C" int gsr_debug_wave_reduce_pk(const float* in640, float* out20, void* stream) {
  hipLaunchKernelGGL(k_debug_wave_reduce_pk, dim3(1), dim3(64), 0, (hipStream_t)stream, in640, out20);
  return gsr_launch_status("debug wave reduce (packed)");
}

// ---------------------------------------------------------------------------------------------------------------
// (round 4) Sub-block masks, computed where the records are STAGED.  The lane that copies list entry e into LDS also decides,
// for that entry, which of the tile's four 8x8 sub-blocks the Gaussian can reach at all: bit s is set unless the maximum of the
// (concave) log2-power over the hull of sub-block s's pixel centres stays below the record's cut-off pmin' (= alpha < 1/255
// everywhere, with the record's own margin), or every pixel of the sub-block finished in front of this entry (entry1 >
// sub_last[s]).  64 entries are tested by 64 lanes at once (~170 VALU wave-instructions per BATCH instead of 8 VALU + two
// ballots + two branches per (entry, sub-block) in the entry loop), and the loop reads an entry's mask into an SGPR
// (v_readlane) and skips missed sub-blocks on the scalar unit.  The test is conservative (max over the continuous rectangle >=
// max over its pixel centres; slack 2e-6 x the magnitude of the form's terms >> fp32 error of either evaluation; the record's
// cut-off already lies 0.0144 below the exact alpha >= 1/255 threshold), and a sub-block that is skipped would only have added
// exact zeros: gradients are bit-identical to the unmasked kernel (tests/test_parity_gpu.py::test_backward_subblock_masks...).
// Max of q(d) = A' dx^2 + B' dx dy + C' dy^2 (A', C' < 0) over a box not containing 0: on one of the four edges, where q is a
// concave parabola in the free coordinate - clamp its vertex into the edge.
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float gsr_clampf(float v, float lo, float hi) { return fminf(fmaxf(v, lo), hi); }

__device__ __forceinline__ uint32_t gsr_subblock_mask(const float4& r0, const float4& r1, float ox, float oy) {
  const float As = r0.z, Bs = r0.w, Cs = r1.x, pmin = r1.z;
  if (!(pmin < 1.0e30f)) return 0u;                        // padding slot: never reached
  const float kx = -0.5f * Bs * __builtin_amdgcn_rcpf(As), ky = -0.5f * Bs * __builtin_amdgcn_rcpf(Cs);
  const float aA = fabsf(As), aB = fabsf(Bs), aC = fabsf(Cs);
  uint32_t m = 0u;
#pragma unroll
  for (int s = 0; s < 4; s++) {
    const float dxh = r0.x - (ox + (float)((s & 1) * 8)), dxl = dxh - 7.0f;      // d = mean - pixel over the sub-block's pixels
    const float dyh = r0.y - (oy + (float)((s >> 1) * 8)), dyl = dyh - 7.0f;
    auto q = [&](float dx, float dy) { return __builtin_fmaf(Cs * dy, dy, __builtin_fmaf(Bs, dy, As * dx) * dx); };
    const float e0 = q(dxl, gsr_clampf(ky * dxl, dyl, dyh)), e1 = q(dxh, gsr_clampf(ky * dxh, dyl, dyh));
    const float e2 = q(gsr_clampf(kx * dyl, dxl, dxh), dyl), e3 = q(gsr_clampf(kx * dyh, dxl, dxh), dyh);
    const bool inside = dxl <= 0.0f && dxh >= 0.0f && dyl <= 0.0f && dyh >= 0.0f;
    const float qmax = inside ? 0.0f : fmaxf(fmaxf(e0, e1), fmaxf(e2, e3));
    const float dxm = fmaxf(fabsf(dxl), fabsf(dxh)), dym = fmaxf(fabsf(dyl), fabsf(dyh));
    const float mag = aA * dxm * dxm + aB * dxm * dym + aC * dym * dym;
    if (qmax >= pmin - 2.0e-6f * mag) m |= 1u << s;
  }
  return m;
}

#ifndef FWD_BATCH
#define FWD_BATCH 256
#endif
#define BALLOT(p) __builtin_amdgcn_ballot_w64(p)

// COUNT = true: the instrumented build behind gsr_debug_count_pairs (SURVEY.md 8(d) "FLOP model": pair evaluations E); it
// additionally counts, per pixel, the list entries evaluated while the pixel was still compositing, and writes nothing else.
// MASKED (round 4; opt-in, GSR_FWD_MASK=1 - measured SLOWER than the plain loop at C3, 0.208 against 0.179 ms: the forward stops
// after ~200 of a tile's 532 entries, so the 256 masks of a batch are mostly computed for nothing, and the walk over set bits loses
// the four-entry unrolled prefetch): the thread that stages entry e also computes which of the tile's
// four 8x8 quadrants its alpha >= 1/255 ellipse can reach (gsr_subblock_mask, conservative) and leaves the four bits in LDS; every
// wave takes the bit of ITS quadrant of all 256 staged entries with four ballots and walks only the entries whose bit is set -
// an entry out of reach costs a scalar bit scan instead of 8 VALU + a ballot + a branch.  Entries that ARE walked take the exact
// published test, in the published order: images, final_T and n_contrib are bit-identical.
template <bool COUNT, bool MASKED>
__global__ __launch_bounds__(256) void k_render_fwd(int W, int H, int grid_x, const uint2* __restrict__ ranges,
                                                    const uint32_t* __restrict__ point_list,
                                                    const float4* __restrict__ rec, const float* __restrict__ bg,
                                                    float* __restrict__ out_color, float* __restrict__ out_invdepth,
                                                    float* __restrict__ final_T, uint32_t* __restrict__ n_contrib,
                                                    uint32_t* __restrict__ pairs,
                                                    const uint32_t* __restrict__ status_src,
                                                    uint32_t* __restrict__ status_dst,
                                                    uint32_t* __restrict__ tile_cutoff /* in/out, or nullptr */,
                                                    const uint32_t* __restrict__ depth_key,
                                                    const uint32_t* __restrict__ culled_any /* nullptr: lists not truncated */,
                                                    uint32_t frame_tag, uint32_t* __restrict__ meta, uint32_t margin_q8,
                                                    uint32_t margin_add, uint32_t* __restrict__ walk_cnt,
                                                    uint32_t* __restrict__ walk_list, uint32_t* __restrict__ walk_of_tile) {
  __shared__ float4 s0[FWD_BATCH + 6], s1[FWD_BATCH + 6], s2[FWD_BATCH];  // +6: the prefetch may touch [n+5]
  __shared__ uint32_t s_need;
  __shared__ uint32_t s_wave_last[4];
  __shared__ uint32_t smask[MASKED ? FWD_BATCH : 1];                      // (MASKED) reachable quadrants of every staged entry
  const int tile = blockIdx.x;
  // non-blocking forward: the frame's status words (flags, num_rendered, longest tile list - final since the previous
  // kernel) go straight to the caller's pinned host slot; a 32-byte hipMemcpyAsync here cost ~10 us of stream time
  // Word 4 (the longest tile list) is the one the host's sentinel sits in (_workspace.py _StatusArrived): it leaves LAST, behind a
  // system-scope fence, so a host that sees it overwritten also sees flags and num_rendered of THIS frame (not a replay's
  // predecessor's).
  if (status_dst && blockIdx.x == 0 && threadIdx.x < 8) {
    // (word 6 is not copied: it is the host's to clear and any workgroup's to set - "a truncated tile list was too short")
    if (threadIdx.x != 4 && threadIdx.x != 6) status_dst[threadIdx.x] = status_src[threadIdx.x];
    __threadfence_system();
    if (threadIdx.x == 4) status_dst[4] = status_src[4];
  }
  const int tile_x = tile % grid_x, tile_y = tile / grid_x;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int px = tile_x * GSR_TILE + (w & 1) * 8 + (lane & 7);
  const int py = tile_y * GSR_TILE + (w >> 1) * 8 + (lane >> 3);
  const bool inside = px < W && py < H;
  const float pxf = (float)px, pyf = (float)py;
  const uint2 range = ranges[tile];
  int toDo = (int)(range.y - range.x);
  const int rounds = (toDo + FWD_BATCH - 1) / FWD_BATCH;

  // A finished (or outside) pixel is moved far away: its power becomes hugely negative, so it fails the wave-level
  // reject test AND the exact alpha test (alpha = 0) with no extra instruction in the loop; "done" is never tested per
  // lane.  `live` (wave-uniform, SGPR pair) holds the lanes still compositing.
  float pxe = inside ? pxf : 1.0e15f;
  uint64_t live = BALLOT(inside);
  float T = 1.0f, C0 = 0.f, C1 = 0.f, C2 = 0.f, D = 0.f;
  uint32_t last = 0, visited = 0, blended = 0;
  uint32_t stop_at = 0;     // 1-based list position of the entry this pixel saturated at (0: it has not)
  if (tile_cutoff && threadIdx.x == 0) s_need = 0u;
  int vzero;   // keeps the LDS base in a VGPR (see k_render_bwd)
  asm volatile("v_mov_b32 %0, 0" : "=v"(vzero));
  const float4 *s0v = s0 + vzero, *s1v = s1 + vzero, *s2v = s2 + vzero;

  for (int r = 0; r < rounds; r++, toDo -= FWD_BATCH) {
    if (__syncthreads_count(pxe > 1.0e14f) == 256) break;
    const uint32_t progress = range.x + (uint32_t)(r * FWD_BATCH + tid);
    uint32_t mymask = 0u;
    if (progress < range.y) {
      const uint32_t id32 = point_list[progress];
      if (id32 != 0xFFFFFFFFu) {
        const size_t id = id32;
        const float4 r0 = rec[3 * id + 0], r1 = rec[3 * id + 1];
        s0[tid] = r0;
        s1[tid] = r1;
        s2[tid] = rec[3 * id + 2];
        if (MASKED) mymask = gsr_subblock_mask(r0, r1, (float)(tile_x * GSR_TILE), (float)(tile_y * GSR_TILE));
      } else {  // padding slot (see k_emit_instances): a record that can never pass the reject test
        s0[tid] = make_float4(0.f, 0.f, 0.f, 0.f);
        s1[tid] = make_float4(0.f, 0.f, 3.0e38f, 0.f);
        s2[tid] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
    if (MASKED) smask[tid] = mymask;
    __syncthreads();
    const int n = toDo < FWD_BATCH ? toDo : FWD_BATCH;
    // one list entry against this wave's 64 pixels
    auto step = [&](const float4& a, const float4& b, const int j) __attribute__((always_inline)) {
      const float dx = a.x - pxe, dy = a.y - pyf;
      const float power = gsr_power2(a, b, dx, dy);
      if (COUNT) visited += pxe < 1.0e14f ? 1u : 0u;
      // conservative wave-level reject (b.z = log2 of 1/(255 opacity), minus a margin): no lane can reach alpha >= 1/255
      if (BALLOT(power >= b.z) != 0ull) {
        const float alpha = fminf(0.99f, b.y * __builtin_amdgcn_exp2f(power));
        const bool ok = power <= 0.0f && alpha >= ALPHA_MIN;
        const float4 c = s2v[j];
        const float test_T = T * (1.0f - alpha);
        const bool stop = ok && test_T < 0.0001f;    // the stopping Gaussian is NOT blended (A.5)
        const bool blend = ok && !stop;
        const float wgt = blend ? alpha * T : 0.f;
        if (COUNT) blended += blend ? 1u : 0u;
        C0 += b.w * wgt;
        C1 += c.x * wgt;
        C2 += c.y * wgt;
        D += c.z * wgt;
        T = blend ? test_T : T;
        last = blend ? (uint32_t)(r * FWD_BATCH + j + 1) : last;
        stop_at = stop ? (uint32_t)(r * FWD_BATCH + j + 1) : stop_at;
        pxe = stop ? 1.0e15f : pxe;
        // (ballots taken straight off the three compares: the AND happens on the scalar unit)
        live &= ~(BALLOT(power <= 0.0f) & BALLOT(alpha >= ALPHA_MIN) & BALLOT(test_T < 0.0001f));  // quadrant saturated -> leave
      }
    };
    if (MASKED) {
      // this wave's quadrant bit of all staged entries (threads beyond the list staged a zero mask), then only the set bits
      uint64_t todo[FWD_BATCH / 64];
#pragma unroll
      for (int k = 0; k < FWD_BATCH / 64; k++) todo[k] = BALLOT(((smask[k * 64 + lane] >> w) & 1u) != 0u);
#pragma unroll
      for (int k = 0; k < FWD_BATCH / 64; k++) {
        uint64_t td = todo[k];
        if (td == 0ull || live == 0ull) continue;
        int j = k * 64 + (int)__builtin_ctzll(td);
        float4 a = s0v[j], b = s1v[j];
        while (true) {
          td &= td - 1ull;
          const int jn = td != 0ull ? k * 64 + (int)__builtin_ctzll(td) : j;
          const float4 an = s0v[jn], bn = s1v[jn];      // the next record is requested before this one is worked on
          step(a, b, j);
          if (td == 0ull || live == 0ull) break;
          a = an; b = bn; j = jn;
        }
      }
      continue;
    }
    // four entries per trip, records prefetched two entries ahead into rotating register sets
    float4 a0 = s0v[0], b0 = s1v[0], a1 = s0v[1], b1 = s1v[1];
    for (int j = 0; j < n && live != 0ull; j += 4) {
      const float4 a2 = s0v[j + 2], b2 = s1v[j + 2];
      step(a0, b0, j);
      const float4 a3 = s0v[j + 3], b3 = s1v[j + 3];
      if (j + 1 < n && live != 0ull) step(a1, b1, j + 1);
      a0 = s0v[j + 4];
      b0 = s1v[j + 4];
      if (j + 2 < n && live != 0ull) step(a2, b2, j + 2);
      a1 = s0v[j + 5];
      b1 = s1v[j + 5];
      if (j + 3 < n && live != 0ull) step(a3, b3, j + 3);
    }
  }
  if (COUNT) {   // pairs[0 .. N) = entries evaluated per pixel, pairs[N .. 2N) = entries blended per pixel
    if (inside) {
      pairs[(size_t)py * W + px] = visited;
      pairs[(size_t)W * H + (size_t)py * W + px] = blended;
    }
    return;
  }
  if (tile_cutoff) {
    // (round 4) How much of this tile's list the frame NEEDED: the deepest entry any of its pixels saturated at - or all of it
    // and more, if a pixel never saturated.  Its depth, with a margin (1.75 x as many entries + 48), is the tile's cut-off for
    // the next render of this view (gsr_forward_async_culled): instances behind it are not even emitted then.  And if THIS
    // frame's list was truncated (culled_any) and a pixel ran off its end unsaturated, a culled Gaussian might have been
    // blended: the frame is flagged, every backward kernel skips it (gsr_overflowed), the caller renders it again untruncated.
    uint32_t need = inside ? (stop_at != 0u ? stop_at : 0xFFFFFFFFu) : 0u;
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) need = max(need, (uint32_t)__shfl_xor((int)need, d, 64));
    __syncthreads();
    if (lane == 0) atomicMax(&s_need, need);
    __syncthreads();
    if (tid == 0) {
      const uint32_t nd = s_need, len = range.y - range.x;
      const bool truncated = culled_any != nullptr && culled_any[tile] == frame_tag;
      if (nd == 0xFFFFFFFFu) {
        if (truncated) {
          atomicOr(&meta[0], GSR_STATUS_CULL_MISS);
          if (status_dst) status_dst[6] = 1u;
        }
        tile_cutoff[tile] = 0xFFFFFFFFu;
      } else {
        // margin: need x margin_q8 / 256 + margin_add entries (GSR_CULL_MARGIN="q8,add" overrides the launcher's default)
        const uint32_t k = (uint32_t)(((unsigned long long)nd * margin_q8) >> 8) + margin_add;   // (positions are 1-based: index k is entry k + 1)
        if (k < len) {
          const uint32_t id = point_list[range.x + k];
          tile_cutoff[tile] = id != 0xFFFFFFFFu ? depth_key[id] : 0xFFFFFFFFu;
        } else if (!truncated) {
          tile_cutoff[tile] = 0xFFFFFFFFu;                  // the margin reaches past the whole list: nothing to cut
        }                                                   // (else: keep the wider cut-off this frame was rendered with)
      }
    }
  }
  if (walk_cnt) {
    // (round 4) The backward walks this tile's list up to the deepest contributor of any of its pixels.  The tile is filed under
    // the class of that length (gsr_walk_class: a coarse logarithm) - one atomic per tile, spread over the launch - and
    // k_render_bwd_tile takes the classes longest first: its launch ends with its longest walk, so the long walks must START
    // first and the short ones fill the second round of resident waves (0.415 -> 0.38 ms at 1080p).
    uint32_t m = last;
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, d, 64));
    if (lane == 0) s_wave_last[w] = m;
    __syncthreads();
    if (tid == 0) {
      const uint32_t walk = max(max(s_wave_last[0], s_wave_last[1]), max(s_wave_last[2], s_wave_last[3]));
      const int cls = gsr_walk_class(walk);
      walk_of_tile[tile] = walk;
      const uint32_t r = atomicAdd(&walk_cnt[cls], 1u);
      if (r < gridDim.x) walk_list[(size_t)cls * gridDim.x + r] = (uint32_t)tile;
    }
  }
  if (inside) {
    const size_t pix = (size_t)py * W + px;
    const size_t N = (size_t)W * H;
    final_T[pix] = T;
    n_contrib[pix] = last;
    out_color[pix] = C0 + T * bg[0];
    out_color[N + pix] = C1 + T * bg[1];
    out_color[2 * N + pix] = C2 + T * bg[2];
    out_invdepth[pix] = D;
  }
}

#ifndef BWD_BATCH
#define BWD_BATCH 128   // entries staged per round in the backward (4 per-wave gradient slabs must fit LDS)
#endif

// DEPTH = false: no gradient arrives on the inverse-depth image (the usual training step): its recurrence and its
// reduction are compiled out.
template <bool DEPTH>
__global__ __launch_bounds__(256) void k_render_bwd(int W, int H, int grid_x, const uint2* __restrict__ ranges,
                                                    const uint32_t* __restrict__ point_list,
                                                    const float4* __restrict__ rec, const float* __restrict__ bg,
                                                    const float* __restrict__ final_T,
                                                    const uint32_t* __restrict__ n_contrib,
                                                    const float* __restrict__ dL_dpix,
                                                    const float* __restrict__ dL_dinvdepth,
                                                    const uint32_t* __restrict__ slot_of_pos,
                                                    float4* __restrict__ igrad, const uint32_t* __restrict__ n_dev,
                                                    uint32_t cap, const uint32_t* __restrict__ walk_cnt,
                                                    const uint32_t* __restrict__ walk_list,
                                                    const uint32_t* __restrict__ walk_of_tile, uint32_t flags_min_r) {
  __shared__ float4 s0[BWD_BATCH + 6], s1[BWD_BATCH + 6], s2[BWD_BATCH];  // +6: the prefetch may touch [n+5]
  if (gsr_overflowed(n_dev, cap)) return;   // grid-uniform: a truncated frame teaches nothing (gsr_common.h)
  unsigned char* const iflags = gsr_igrad_flags(igrad, cap);
  const bool flags_on = gsr_flags_on(n_dev, cap, flags_min_r);   // (grid-uniform: validity bytes instead of zero records, gsr_common.h)
  // one private slab per wave: no LDS atomics, and the 4 partial sums are added in a FIXED order at flush time,
  // so gradients are bitwise reproducible
  __shared__ float4 slab[4][BWD_BATCH * GSR_IGRAD_F4];
  __shared__ int s_max;

  // (round 4) walk classes, as in k_render_bwd_tile below: workgroup b writes the zero records behind the walk of tile b (index
  // order: neighbouring tiles' partial-line writes combine) and composites the b-th tile of the order "longest walk first"; every
  // wave works the mapping out for itself (a 64-lane suffix sum of the class sizes: no barrier)
  int tile = blockIdx.x;
  uint32_t csum = walk_cnt ? walk_cnt[threadIdx.x & 63] : 0u;
  if (walk_cnt) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const uint32_t o = (uint32_t)__shfl_down((int)csum, d, 64);
      if ((int)(threadIdx.x & 63) + d < 64) csum += o;
    }
    if ((uint32_t)__builtin_amdgcn_readfirstlane((int)csum) != gridDim.x) walk_cnt = nullptr;   // (see k_render_bwd_tile)
  }
  if (walk_cnt) {
    {
      const uint2 zr = ranges[blockIdx.x];
      const int zlen = (int)(zr.y - zr.x);
      const int zfrom = min(zlen, (int)walk_of_tile[blockIdx.x]);
      for (int i = zfrom + (int)threadIdx.x; i < zlen; i += 256) {
        const uint32_t zslot = slot_of_pos[zr.x + i];
        if (flags_on) iflags[zslot] = 0;      // (no record: gsr_common.h GSR_IGRAD_F4)
        else {
          float4* dst = igrad + (size_t)GSR_IGRAD_F4 * zslot;
          dst[0] = dst[1] = dst[2] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
      }
    }
    const uint64_t mcls = BALLOT(csum > blockIdx.x);
    const int cs = 63 - (int)__builtin_clzll(mcls);
    const uint32_t above = cs < 63 ? (uint32_t)__builtin_amdgcn_readlane((int)csum, cs + 1) : 0u;
    tile = (int)min(walk_list[(size_t)cs * gridDim.x + (blockIdx.x - above)], gridDim.x - 1u);
  }
  const int tile_x = tile % grid_x, tile_y = tile / grid_x;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int px = tile_x * GSR_TILE + (w & 1) * 8 + (lane & 7);
  const int py = tile_y * GSR_TILE + (w >> 1) * 8 + (lane >> 3);
  const bool inside = px < W && py < H;
  const float pxf = (float)px, pyf = (float)py;
  const uint2 range = ranges[tile];
  const int len = (int)(range.y - range.x);
  if (len == 0) return;  // block-uniform

  const size_t pix = (size_t)py * W + px;
  const size_t N = (size_t)W * H;
  const float T_final = inside ? final_T[pix] : 0.f;
  const int last = inside ? (int)n_contrib[pix] : 0;
  float gp0 = 0.f, gp1 = 0.f, gp2 = 0.f, gd = 0.f;
  if (inside) {
    gp0 = dL_dpix[pix];
    gp1 = dL_dpix[N + pix];
    gp2 = dL_dpix[2 * N + pix];
    if (DEPTH) gd = dL_dinvdepth[pix];
  }
  const float neg_Tf_bg = -T_final * (bg[0] * gp0 + bg[1] * gp1 + bg[2] * gp2);

  // entries beyond the deepest contributor of any pixel are never visited
  if (tid == 0) s_max = 0;
  __syncthreads();
  int wave_last = last;   // deepest contributor among this wave's 64 pixels
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) wave_last = max(wave_last, __shfl_xor(wave_last, d, 64));
  wave_last = __builtin_amdgcn_readfirstlane(wave_last);
  if (lane == 0) atomicMax(&s_max, wave_last);
  __syncthreads();
  const int toDo = min(len, s_max);
  const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
  // Gradient records are stored by EMISSION SLOT (slot_of_pos = the tile sort's value array), i.e. grouped per
  // Gaussian, so the per-Gaussian sum in k_preprocess_bwd streams contiguous memory; the scattered 48-B stores here
  // are fire-and-forget.  (With walk classes the zero records were written above, by the workgroup of the tile's index.)
  if (!walk_cnt)
    for (int i = toDo + tid; i < len; i += 256) {
      const uint32_t zslot = slot_of_pos[range.x + i];
      if (flags_on) iflags[zslot] = 0;      // (no record: gsr_common.h GSR_IGRAD_F4)
      else {
        float4* dst = igrad + (size_t)GSR_IGRAD_F4 * zslot;
        dst[0] = dst[1] = dst[2] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }

  const int rounds = (toDo + BWD_BATCH - 1) / BWD_BATCH;
  float T = T_final;
  // The published per-channel recurrence "colour accumulated behind" ar_c enters the gradient only through
  // sum_c (c_c - ar_c) gp_c, and the recurrence is linear, so ONE scalar S = sum_c ar_c gp_c (the loss-weighted colour
  // behind this pixel, depth channel included) carries it.  After an entry with alpha a and colour c:
  // S <- a (c . gp) + (1 - a) S = S + a (c . gp - S), and (c . gp - S) is the very term dL/dalpha needs: one fma.
  float S = 0.f;
  float4* myslab = slab[w];
  // The staged records are read at wave-uniform addresses; a zero the compiler cannot see through keeps the base in a
  // VGPR (ds_read takes its address from one), advanced once per trip, instead of an SGPR re-copied before every read.
  int vzero;
  asm volatile("v_mov_b32 %0, 0" : "=v"(vzero));
  const float4 *s0v = s0 + vzero, *s1v = s1 + vzero, *s2v = s2 + vzero;
  const bool lane_bit3 = (lane & 8) != 0, octet_lead = (lane & 7) == 0;
  const bool u1_lead = DEPTH ? (lane & 31) == 0 : lane == 63;   // lanes holding the v8 / v9 totals
  const int u1_slot = DEPTH ? 8 + (lane >> 5) : 8;
  const int octet_val = (((lane >> 3) & 1) << 2) | ((lane >> 3) & 2) | (((lane >> 3) & 4) >> 2);

  for (int b = 0; b < rounds; b++) {
    __syncthreads();
    const int e_idx = toDo - 1 - (b * BWD_BATCH + tid);  // back-to-front staging (threads 0..127)
    if (tid < BWD_BATCH && e_idx >= 0) {
      const uint32_t id32 = point_list[range.x + e_idx];
      if (id32 != 0xFFFFFFFFu) {
        const size_t id = id32;
        s0[tid] = rec[3 * id + 0];
        s1[tid] = rec[3 * id + 1];
        s2[tid] = rec[3 * id + 2];
      } else {
        s0[tid] = make_float4(0.f, 0.f, 0.f, 0.f);
        s1[tid] = make_float4(0.f, 0.f, 3.0e38f, 0.f);
        s2[tid] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
    // each wave clears its own slab (384 float4 / 64 lanes)
#pragma unroll
    for (int i = 0; i < (BWD_BATCH * GSR_IGRAD_F4) / 64; i++) myslab[i * 64 + lane] = z4;
    __syncthreads();
    const int n = min(BWD_BATCH, toDo - b * BWD_BATCH);
    auto step = [&](const float4& a, const float4& bb, const int j) __attribute__((always_inline)) {
      const int entry1 = toDo - (b * BWD_BATCH + j);  // 1-based list position of this entry
      const float dx = a.x - pxf, dy = a.y - pyf;
      const float power = gsr_power2(a, bb, dx, dy);
      const bool pre = entry1 <= last && power >= bb.z;   // conservative wave-level reject (see forward)
      // (every ballot is taken straight off ONE compare; the ANDs run on the scalar unit)
      const uint64_t m_pre = BALLOT(entry1 <= last) & BALLOT(power >= bb.z);
      if (m_pre == 0ull) return;
      const float G = __builtin_amdgcn_exp2f(power);
      const float alpha = fminf(0.99f, bb.y * G);
      const bool ok = pre && power <= 0.0f && alpha >= ALPHA_MIN;
      if ((m_pre & BALLOT(power <= 0.0f) & BALLOT(alpha >= ALPHA_MIN)) == 0ull) return;
      const float4 c = s2v[j];
      // Lanes that do not blend this entry run the same arithmetic with alpha = G = 0: T, the "accumulated behind"
      // recurrences and every gradient term then stay exactly unchanged / zero, so no per-lane branch is needed.
      const float a_e = ok ? alpha : 0.f;
      const float G_e = ok ? G : 0.f;
      const float rcp = __builtin_amdgcn_rcpf(1.0f - a_e);
      T = T * rcp;
      const float dch = a_e * T;
      float cg = bb.w * gp0 + c.x * gp1 + c.y * gp2;                  // this entry's colour . dL/dpixel
      if (DEPTH) cg += c.z * gd;
      const float diff = cg - S;
      S = __builtin_fmaf(a_e, diff, S);
      const float dL_dalpha = diff * T + neg_Tf_bg * rcp;
      const float v6 = dch * gp0, v7 = dch * gp1, v8 = dch * gp2, v9 = DEPTH ? dch * gd : 0.f;
      // Geometry: only the raw moments of v5 = dL/dpower / opacity_eff are reduced here; the per-Gaussian linear map to
      // (dL/dmean2D, dL/dconic) uses wave-uniform factors (opacity_eff, conic, W/2, H/2) and is applied once per Gaussian AFTER
      // the sum over its instances, in k_preprocess_bwd.
      const float v5 = G_e * dL_dalpha;                               // -> dL/dopacity_eff; g = opacity_eff v5 = dL/dpower
      const float v0 = v5 * dx, v1 = v5 * dy;                         // sum v5 dx, sum v5 dy   (the factor opacity_eff: k_preprocess_bwd)
      const float v2 = v0 * dx, v3 = v0 * dy, v4 = v1 * dy;           // sum v5 dx^2, v5 dx dy, v5 dy^2
      float u0, u1;
      if (DEPTH) wave_sum10_halving(v0, v1, v2, v3, v4, v5, v6, v7, v8, v9, lane_bit3, u0, u1);
      else wave_sum9_halving(v0, v1, v2, v3, v4, v5, v6, v7, v8, lane_bit3, u0, u1);
      float* dst = reinterpret_cast<float*>(myslab) + 12 * j;
      if (octet_lead) dst[octet_val] = u0;        // 8 lanes store v0..v7 totals
      if (u1_lead) dst[u1_slot] = u1;             // v8 (/ v9) totals
    };
    // four entries per trip, records prefetched two entries ahead into rotating register sets; constant LDS offsets from
    // one base per trip (no per-entry address arithmetic)
    // entries deeper than this wave's deepest contributor come first in the back-to-front order: skip them wholesale
    const int jstart = max(0, (toDo - b * BWD_BATCH) - wave_last) & ~3;
    float4 a0 = s0v[jstart], b0 = s1v[jstart], a1 = s0v[jstart + 1], b1 = s1v[jstart + 1];
    for (int j = jstart; j < n; j += 4) {
      const float4 a2 = s0v[j + 2], b2 = s1v[j + 2];
      step(a0, b0, j);
      const float4 a3 = s0v[j + 3], b3 = s1v[j + 3];
      if (j + 1 < n) step(a1, b1, j + 1);
      a0 = s0v[j + 4];
      b0 = s1v[j + 4];
      if (j + 2 < n) step(a2, b2, j + 2);
      a1 = s0v[j + 5];
      b1 = s1v[j + 5];
      if (j + 3 < n) step(a3, b3, j + 3);
    }
    __syncthreads();
    // flush: 128 entries x 3 float4 = 384 float4, fixed summation order over the 4 waves
    for (int q = tid; q < n * GSR_IGRAD_F4; q += 256) {
      const int j = q / GSR_IGRAD_F4, part = q - j * GSR_IGRAD_F4;
      const float4 a0 = slab[0][q], a1 = slab[1][q], a2 = slab[2][q], a3 = slab[3][q];
      float4 r;
      r.x = ((a0.x + a1.x) + a2.x) + a3.x;
      r.y = ((a0.y + a1.y) + a2.y) + a3.y;
      r.z = ((a0.z + a1.z) + a2.z) + a3.z;
      r.w = ((a0.w + a1.w) + a2.w) + a3.w;
      const int e = toDo - 1 - (b * BWD_BATCH + j);
      const uint32_t slot = slot_of_pos[range.x + e];
      igrad[(size_t)GSR_IGRAD_F4 * slot + part] = r;
      if (flags_on && part == 0) iflags[slot] = 1;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Backward, second form: ONE wave per tile, four pixels per lane.
//
// The micro-benchmark (profiles/r02_valu_microbench.txt) prices the cross-lane instructions of the reduction at 1.6x
// (DPP add) and 3.2x (v_permlane32/16_swap) a plain VALU instruction, which makes the nine-value wave reduction ~45 % of
// a hit's issue cost in the four-waves-per-tile kernel above - and it is paid per (8x8 quadrant, Gaussian).  Here a lane owns
// the same (x, y) of all four 8x8 sub-blocks of the tile, accumulates the nine sums of a Gaussian over its sub-blocks in
// registers (plain fma) and the wave reduces ONCE per (tile, Gaussian): the reduction is shared by up to four hits, the
// wave's total IS the tile's total (no per-wave slabs, no cross-wave pass, no workgroup barrier, 6 KB of LDS instead of
// 31 KB), and a sub-block the Gaussian misses still costs only the 9-instruction reject test behind a scalar skip.
// Summation order is fixed (sub-blocks 0..3 in a lane, then the halving tree): bitwise reproducible like the first form.
// ---------------------------------------------------------------------------------------------------------------
#ifndef BWD1_BATCH
#define BWD1_BATCH 64
#endif
// MASK = true (round 4, the default): sub-block masks from the staging step (gsr_subblock_mask) replace the per-(entry, sub-block)
// wave-level reject tests of the loop; MASK = false (GSR_BWD_MASK=0): the round-3 loop, kept for the A/B and the bit-identity test.
#ifndef BWD_LDS_REDUCE
#define BWD_LDS_REDUCE 0   // measured out: 0.440 ms against 0.416 for the swap tree at C3 (profiles/r04_bwd_lds_reduce_ab.txt)
#endif
#ifndef BWD_TILE_WAVES
#define BWD_TILE_WAVES 5   // waves per SIMD the register allocation must admit (<= 96 VGPRs; measured: 4, 5 and 6 within 2 %, 6 needs spills)
#endif
template <bool DEPTH, bool MASK>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(BWD_TILE_WAVES, 8))) void k_render_bwd_tile(int W, int H, int grid_x, const uint2* __restrict__ ranges,
                                                        const uint32_t* __restrict__ point_list,
                                                        const float4* __restrict__ rec, const float* __restrict__ bg,
                                                        const float* __restrict__ final_T,
                                                        const uint32_t* __restrict__ n_contrib,
                                                        const float* __restrict__ dL_dpix,
                                                        const float* __restrict__ dL_dinvdepth,
                                                        const uint32_t* __restrict__ slot_of_pos,
                                                        float4* __restrict__ igrad, const uint32_t* __restrict__ n_dev,
                                                        uint32_t cap, int prio1, int prio2, int prio3,
                                                        const uint32_t* __restrict__ walk_cnt,
                                                        const uint32_t* __restrict__ walk_list,
                                                        const uint32_t* __restrict__ walk_of_tile, uint32_t flags_min_r) {
  // (one object: the three arrays sit at fixed distances, so an entry's reads share ONE address register and differ in the
  // instruction's immediate offset - two v_add_u32 per walked entry less than three separate __shared__ arrays cost)
  __shared__ struct { float4 s0[BWD1_BATCH + 2], s1[BWD1_BATCH + 2], s2[BWD1_BATCH]; } stg;   // +2: the prefetch may touch [n+1]
  float4 (&s0)[BWD1_BATCH + 2] = stg.s0;
  float4 (&s1)[BWD1_BATCH + 2] = stg.s1;
  float4 (&s2)[BWD1_BATCH] = stg.s2;
  if (gsr_overflowed(n_dev, cap)) return;   // grid-uniform: a truncated frame teaches nothing (gsr_common.h)
  unsigned char* const iflags = gsr_igrad_flags(igrad, cap);
  const bool flags_on = gsr_flags_on(n_dev, cap, flags_min_r);   // (grid-uniform: validity bytes instead of zero records, gsr_common.h)
  __shared__ float4 outb[BWD1_BATCH * GSR_IGRAD_F4];                          // the batch's gradient records
#if BWD_LDS_REDUCE
  // The first two stages of the halving tree (lanes l + l^32, rows r + r^1: six v_permlane swaps at 3.2 x the price of a plain
  // VALU instruction) through LDS instead: every lane stores its sums ([value][lane], rows 80 floats apart: two lanes per bank,
  // the minimum for 64 lanes), lane (row q, column c) reads the four rows' entries of values q and 4 + q at column c and adds them
  // in the tree's own order, (x0 + x2) + (x1 + x3) - the same bits - and the remaining stages run as before.  The reads are
  // issued at once and consumed after the NEXT entry's visit, so their latency hides behind it.
  __shared__ float redbuf[(MASK ? 10 : 1) * 80];
#endif
  // (round 4) Workgroup b takes the b-th tile of the order "walk classes descending, inside a class as filed by the forward"
  // (k_render_fwd): lane l holds the size of class l, a suffix sum over the 64 lanes gives the tiles in classes >= l, the class of
  // position b is the highest one whose suffix sum exceeds b.  Which workgroup takes which tile does not enter any result.
  int tile = blockIdx.x;
  // (class sizes that do not add up to the grid - an image state no forward-with-backward has filed - mean index order, not a fault)
  uint32_t csum = walk_cnt ? walk_cnt[threadIdx.x] : 0u;
  if (walk_cnt) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const uint32_t o = (uint32_t)__shfl_down((int)csum, d, 64);
      if ((int)threadIdx.x + d < 64) csum += o;
    }
    if ((uint32_t)__builtin_amdgcn_readfirstlane((int)csum) != gridDim.x) walk_cnt = nullptr;
  }
  if (walk_cnt) {
    // Two duties, two orders.  The entries BEHIND a tile's walk get zero records, hundreds per tile on scenes of large splats, and
    // neighbouring tiles write them into the same Gaussians' record regions: in index order those partial-line writes combine in
    // the caches, in walk order they do not (2 x splats: 0.53 -> 0.58 ms, 4K: 1.39 -> 1.55 with the zeros written by the walking
    // workgroup).  So workgroup b writes the zeros of tile b - index order, as ever - and walks the b-th tile of the walk order.
    {
      const uint2 zr = ranges[blockIdx.x];
      const int zlen = (int)(zr.y - zr.x);
      const int zfrom = min(zlen, (int)walk_of_tile[blockIdx.x]);
      for (int i = zfrom + (int)threadIdx.x; i < zlen; i += 64) {
        const uint32_t zslot = slot_of_pos[zr.x + i];
        if (flags_on) iflags[zslot] = 0;      // (no record: gsr_common.h GSR_IGRAD_F4)
        else {
          float4* dst = igrad + (size_t)GSR_IGRAD_F4 * zslot;
          dst[0] = dst[1] = dst[2] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
      }
    }
    const uint64_t mcls = BALLOT(csum > blockIdx.x);                // (not empty: the sizes add up to the grid)
    const int cs = 63 - (int)__builtin_clzll(mcls);
    const uint32_t above = cs < 63 ? (uint32_t)__builtin_amdgcn_readlane((int)csum, cs + 1) : 0u;
    tile = (int)min(walk_list[(size_t)cs * gridDim.x + (blockIdx.x - above)], gridDim.x - 1u);
  }
  const int tile_x = tile % grid_x, tile_y = tile / grid_x;
  const int lane = threadIdx.x;
  const uint2 range = ranges[tile];
  const int len = (int)(range.y - range.x);
  if (len == 0) return;

  const size_t N = (size_t)W * H;
  const float bg0 = bg[0], bg1 = bg[1], bg2 = bg[2];
  // sub-block s: origin (8 (s & 1), 8 (s >> 1)) inside the tile
  const float px0 = (float)(tile_x * GSR_TILE + (lane & 7)), py0 = (float)(tile_y * GSR_TILE + (lane >> 3));
  float T[4], S[4], gp0[4], gp1[4], gp2[4], gd[4], nTb[4];
  int last[4];
  int sub_last[4];
#pragma unroll
  for (int s = 0; s < 4; s++) {
    const int px = tile_x * GSR_TILE + (s & 1) * 8 + (lane & 7), py = tile_y * GSR_TILE + (s >> 1) * 8 + (lane >> 3);
    const bool inside = px < W && py < H;
    const size_t pix = (size_t)py * W + px;
    T[s] = inside ? final_T[pix] : 0.f;
    last[s] = inside ? (int)n_contrib[pix] : 0;
    gp0[s] = inside ? dL_dpix[pix] : 0.f;
    gp1[s] = inside ? dL_dpix[N + pix] : 0.f;
    gp2[s] = inside ? dL_dpix[2 * N + pix] : 0.f;
    gd[s] = (DEPTH && inside) ? dL_dinvdepth[pix] : 0.f;
    nTb[s] = -T[s] * (bg0 * gp0[s] + bg1 * gp1[s] + bg2 * gp2[s]);
    S[s] = 0.f;
    int m = last[s];
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) m = max(m, __shfl_xor(m, d, 64));
    sub_last[s] = __builtin_amdgcn_readfirstlane(m);   // deepest contributor among this sub-block's 64 pixels
  }
  const int toDo = min(len, max(max(sub_last[0], sub_last[1]), max(sub_last[2], sub_last[3])));
  // (round 4) One wave walks one tile's list from end to end, so the launch cannot end before its LONGEST walk has: at 1080p 380
  // entries against a mean of 203 (tools/tile_stats.py), ~2000 cycles per entry for a wave on its own and ~2700 with five
  // waves sharing a SIMD - the 0.43 ms the launch took.  Waves with long walks therefore ask for issue priority (s_setprio):
  // they run at close to a lone wave's pace while the short walks fill the slots they leave (-3.5 %, profiles/r04_bwd_prio_ab.txt).
  // "Long" is judged against the frame itself: the mean n_contrib of 64 pixels sampled across the image (one load per lane) -
  // at 1080p the mean walk is 1.35 x that figure; 1.4 / 1.65 / 1.9 x it earn priority 1 / 2 / 3.  prio1 >= 0: fixed thresholds
  // (GSR_BWD_PRIO="t1,t2,t3", experiments); prio1 == -2: no priorities.  Scheduling only: results cannot depend on it.
  if (prio1 != -2) {
    int t1 = prio1, t2 = prio2, t3 = prio3;
    if (prio1 < 0) {
      const uint32_t sp = ((uint32_t)lane * 2654435761u + (uint32_t)tile * 40503u + 12345u) % (uint32_t)N;
      float m = (float)n_contrib[sp];
#pragma unroll
      for (int d = 32; d > 0; d >>= 1) m += __shfl_xor(m, d, 64);
      m *= (1.0f / 64.0f);
      t1 = (int)(1.4f * m); t2 = (int)(1.65f * m); t3 = (int)(1.9f * m);
    }
    if (toDo >= t3) __builtin_amdgcn_s_setprio(3);
    else if (toDo >= t2) __builtin_amdgcn_s_setprio(2);
    else if (toDo >= t1) __builtin_amdgcn_s_setprio(1);
  }
  const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
  // entries beyond the deepest contributor of any pixel are never visited: their records are zeros (with walk classes: written by
  // the workgroup of the tile's index, above)
  if (!walk_cnt)
    for (int i = toDo + lane; i < len; i += 64) {
      const uint32_t zslot = slot_of_pos[range.x + i];
      if (flags_on) iflags[zslot] = 0;      // (no record: gsr_common.h GSR_IGRAD_F4)
      else {
        float4* dst = igrad + (size_t)GSR_IGRAD_F4 * zslot;
        dst[0] = dst[1] = dst[2] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
  int vzero;   // keeps the LDS base in a VGPR (see k_render_bwd)
  asm volatile("v_mov_b32 %0, 0" : "=v"(vzero));
  const char* const stgv = reinterpret_cast<const char*>(&stg) + vzero;
  const float4* const s0v = reinterpret_cast<const float4*>(stgv);
  const float4* const s1v = reinterpret_cast<const float4*>(stgv + sizeof(stg.s0));
  const float4* const s2v = reinterpret_cast<const float4*>(stgv + sizeof(stg.s0) + sizeof(stg.s1));
  const bool lane_bit3 = (lane & 8) != 0, octet_lead = (lane & 7) == 0;
  const bool u1_lead = DEPTH ? (lane & 31) == 0 : lane == 63;
  const int u1_slot = DEPTH ? 8 + (lane >> 5) : 8;
  const int octet_val = (((lane >> 3) & 1) << 2) | ((lane >> 3) & 2) | (((lane >> 3) & 4) >> 2);
  const int octet_val_pk = GSR_OCTET_VALUE_PK(lane >> 3);   // (packed-pair trees: another value order over the octets)
  // (nine-value form: the lane holding the ninth total, 63, leads no octet - one per-lane slot offset serves both stores of a record)
  float* const st_pk = reinterpret_cast<float*>(outb) + (octet_lead ? octet_val_pk : 8);
  const int octet_val_lds = (((lane >> 3) & 1) << 2) | (lane >> 4);   // (LDS form: row q holds values q | 4 + q)
  (void)octet_val_pk; (void)octet_val_lds;
  float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f, acc3 = 0.f, acc4 = 0.f, acc5 = 0.f, acc6 = 0.f, acc7 = 0.f, acc8 = 0.f,
        acc9 = 0.f;
  // MASK form: the same sums as register PAIRS (A01 = (acc0, acc1) ...): v_pk_fma_f32 / v_pk_mul_f32 do two of the body's
  // accumulations per instruction and v_pk_add_f32 two of the reduction's adds, with the very operations (and contractions) of the
  // scalar loop per component, so the bits do not change
  gsr_f2 A01 = {0.f, 0.f}, A23 = {0.f, 0.f}, A45 = {0.f, 0.f}, A67 = {0.f, 0.f}, A89 = {0.f, 0.f};
  gsr_f2 gp01[4];
#pragma unroll
  for (int s = 0; s < 4; s++) gp01[s] = gsr_f2{gp0[s], gp1[s]};

  const int rounds = (toDo + BWD1_BATCH - 1) / BWD1_BATCH;
  for (int b = 0; b < rounds; b++) {
    __syncthreads();   // (one wave: orders this wave's LDS reads of the previous batch before the stores below)
    const int e_idx = toDo - 1 - (b * BWD1_BATCH + lane);   // back-to-front staging
    uint32_t mymask = 0u;                                    // (MASK) bit s: entry `lane` of this batch can touch sub-block s
    if (e_idx >= 0) {
      const uint32_t id32 = point_list[range.x + e_idx];
      if (id32 != 0xFFFFFFFFu) {
        const size_t id = id32;
        const float4 r0 = rec[3 * id + 0], r1 = rec[3 * id + 1];
        s0[lane] = r0;
        s1[lane] = r1;
        s2[lane] = rec[3 * id + 2];
        if (MASK) {
          mymask = gsr_subblock_mask(r0, r1, (float)(tile_x * GSR_TILE), (float)(tile_y * GSR_TILE));
#pragma unroll
          for (int s = 0; s < 4; s++)
            if (e_idx + 1 > sub_last[s]) mymask &= ~(1u << s);   // every pixel of the sub-block finished in front of it
        }
      } else {
        s0[lane] = z4;
        s1[lane] = make_float4(0.f, 0.f, 3.0e38f, 0.f);
        s2[lane] = z4;
      }
    }
#pragma unroll
    for (int i = 0; i < (BWD1_BATCH * GSR_IGRAD_F4) / 64; i++) outb[i * 64 + lane] = z4;
    __syncthreads();
    const int n = min(BWD1_BATCH, toDo - b * BWD1_BATCH);
    // one list entry (record halves a, bb already in registers) against the tile's four sub-blocks; -> true if any pixel blended it
    auto visit = [&](const int j, const float4& a, const float4& bb, const uint32_t emask) __attribute__((always_inline)) -> bool {
      const int entry1 = toDo - (b * BWD1_BATCH + j);  // 1-based list position of this entry
      bool any = false;                                // (wave-uniform)
      const float dx0 = a.x - px0, dy0 = a.y - py0;
      float4 c;
      bool have_c = false;
#pragma unroll
      for (int s = 0; s < 4; s++) {
        float power, G, alpha;
        bool ok;
        float dx, dy;
        if (MASK) {
          if (!(emask & (1u << s))) continue;          // scalar: the Gaussian cannot reach this sub-block (or nobody is left in it)
          dx = dx0 - (float)((s & 1) * 8);
          dy = dy0 - (float)((s >> 1) * 8);
          power = gsr_power2(a, bb, dx, dy);
          G = __builtin_amdgcn_exp2f(power);
          alpha = fminf(0.99f, bb.y * G);
          // (alpha >= 1/255 implies power >= bb.z: the cut-off lies 0.0144 below the threshold, so this IS the round-3 `ok`)
          ok = entry1 <= last[s] && power <= 0.0f && alpha >= ALPHA_MIN;
          if ((BALLOT(entry1 <= last[s]) & BALLOT(power <= 0.0f) & BALLOT(alpha >= ALPHA_MIN)) == 0ull) continue;
        } else {
          if (entry1 > sub_last[s]) continue;            // scalar: every pixel of this sub-block finished earlier
          dx = dx0 - (float)((s & 1) * 8);
          dy = dy0 - (float)((s >> 1) * 8);
          power = gsr_power2(a, bb, dx, dy);
          const bool pre = entry1 <= last[s] && power >= bb.z;
          const uint64_t m_pre = BALLOT(entry1 <= last[s]) & BALLOT(power >= bb.z);
          if (m_pre == 0ull) continue;
          G = __builtin_amdgcn_exp2f(power);
          alpha = fminf(0.99f, bb.y * G);
          ok = pre && power <= 0.0f && alpha >= ALPHA_MIN;
          if ((m_pre & BALLOT(power <= 0.0f) & BALLOT(alpha >= ALPHA_MIN)) == 0ull) continue;
        }
        if (!have_c) { c = s2v[j]; have_c = true; }
        any = true;
        const float a_e = ok ? alpha : 0.f;
        const float G_e = ok ? G : 0.f;
        const float rcp = __builtin_amdgcn_rcpf(1.0f - a_e);
        T[s] = T[s] * rcp;
        const float dch = a_e * T[s];
        float cg = bb.w * gp0[s] + c.x * gp1[s] + c.y * gp2[s];
        if (DEPTH) cg += c.z * gd[s];
        const float diff = cg - S[s];
        S[s] = __builtin_fmaf(a_e, diff, S[s]);
        const float dL_dalpha = diff * T[s] + nTb[s] * rcp;
        // v5 = this pixel's dL/dopacity_eff; the geometry moments are taken of v5, not of g = opacity_eff v5 = dL/dpower: the
        // factor is wave-uniform and is applied once per Gaussian, after the sum over its instances (k_preprocess_bwd)
        const float v5 = G_e * dL_dalpha;
        if (MASK) {
          // per component exactly what the scalar branch below compiles to: acc0 / acc1 = fma(d, v5, acc) (the compiler
          // contracts `acc += v5 * d`), t = v5 * d, acc2 / acc3 = fma(t0, d, acc), acc5 = fma(G_e, dL/dalpha, acc5)
          const gsr_f2 dxy = {dx, dy}, v55 = {v5, v5};
          A01 = __builtin_elementwise_fma(dxy, v55, A01);
          const gsr_f2 t01 = v55 * dxy;
          A23 = __builtin_elementwise_fma(gsr_f2{t01.x, t01.x}, dxy, A23);
          A45.x = __builtin_fmaf(t01.y, dy, A45.x);
          A45.y = __builtin_fmaf(G_e, dL_dalpha, A45.y);
          A67 = __builtin_elementwise_fma(gsr_f2{dch, dch}, gp01[s], A67);
          A89.x = __builtin_fmaf(dch, gp2[s], A89.x);
          if (DEPTH) A89.y = __builtin_fmaf(dch, gd[s], A89.y);
        } else {
          const float t0 = v5 * dx, t1 = v5 * dy;
          acc0 += t0;
          acc1 += t1;
          acc2 = __builtin_fmaf(t0, dx, acc2);
          acc3 = __builtin_fmaf(t0, dy, acc3);
          acc4 = __builtin_fmaf(t1, dy, acc4);
          acc5 += v5;
          acc6 = __builtin_fmaf(dch, gp0[s], acc6);
          acc7 = __builtin_fmaf(dch, gp1[s], acc7);
          acc8 = __builtin_fmaf(dch, gp2[s], acc8);
          if (DEPTH) acc9 = __builtin_fmaf(dch, gd[s], acc9);
        }
      }
      return any;
    };
    // the wave's nine (ten) sums of entry j -> its gradient record in LDS; the sums start over
#if BWD_LDS_REDUCE
    // (MASK form) pending reduction: the eight partial rows read back from LDS, the ninth (tenth) value's butterfly, the entry
    float px0 = 0.f, px1 = 0.f, px2 = 0.f, px3 = 0.f, py0 = 0.f, py1 = 0.f, py2 = 0.f, py3 = 0.f, pz0 = 0.f, pz1 = 0.f, pz2 = 0.f,
          pz3 = 0.f, pb = 0.f;
    int pend_j = -1;                                     // (wave-uniform)
    auto consume = [&]() __attribute__((always_inline)) {
      if (pend_j < 0) return;
      const float s0 = (px0 + px2) + (px1 + px3);        // value q  at column c: (lane + lane^32) + the same of row r^1
      const float s1 = (py0 + py2) + (py1 + py3);        // value 4 + q
      const float keep = lane_bit3 ? s1 : s0;
      const float send = lane_bit3 ? s0 : s1;
      float a = keep + dpp_get<0x128>(send);             // row_ror:8
      a += dpp_get<0xB1>(a);
      a += dpp_get<0x4E>(a);
      a += dpp_get<0x141>(a);
      float b = pb;
      if (DEPTH) {                                       // rows [v8, 0, v9, 0] as in wave_sum10_halving
        const float s2 = (pz0 + pz2) + (pz1 + pz3);
        b = s2 + dpp_get<0x128>(s2);
        b += dpp_get<0xB1>(b);
        b += dpp_get<0x4E>(b);
        b += dpp_get<0x141>(b);
      }
      float* dst = reinterpret_cast<float*>(outb) + 12 * pend_j;
      if (octet_lead) dst[octet_val_lds] = a;
      if (u1_lead) dst[u1_slot] = b;
      pend_j = -1;
    };
#endif
    auto reduce = [&](const int j) __attribute__((always_inline)) {
      float u0, u1;
      float* dst = reinterpret_cast<float*>(outb) + 12 * j;
      if (MASK) {
#if BWD_LDS_REDUCE
        float* rb = redbuf + lane;
        rb[0 * 80] = A01.x; rb[1 * 80] = A01.y; rb[2 * 80] = A23.x; rb[3 * 80] = A23.y;
        rb[4 * 80] = A45.x; rb[5 * 80] = A45.y; rb[6 * 80] = A67.x; rb[7 * 80] = A67.y;
        if (DEPTH) { rb[8 * 80] = A89.x; rb[9 * 80] = A89.y; }
        const int q = lane >> 4, c = lane & 15;
        const float* r0 = redbuf + q * 80 + c;
        const float* r1 = redbuf + (4 + q) * 80 + c;
        px0 = r0[0]; px1 = r0[16]; px2 = r0[32]; px3 = r0[48];
        py0 = r1[0]; py1 = r1[16]; py2 = r1[32]; py3 = r1[48];
        if (DEPTH) {
          // rows 0 / 2 take values 8 / 9, rows 1 / 3 nothing (their lanes add zeros, as the swap tree's `swap16_add(r4, 0)` does)
          const float* r2 = redbuf + (8 + (q >> 1)) * 80 + c;
          const bool on = (q & 1) == 0;
          pz0 = on ? r2[0] : 0.f; pz1 = on ? r2[16] : 0.f; pz2 = on ? r2[32] : 0.f; pz3 = on ? r2[48] : 0.f;
        } else {
          float b = A89.x + dpp_get<0xB1>(A89.x);        // the lone ninth value: the DPP butterfly of wave_sum9_halving
          b += dpp_get<0x4E>(b);
          b += dpp_get<0x141>(b);
          b += dpp_get<0x140>(b);
          b += dpp_get<0x142>(b);
          b += dpp_get<0x143>(b);
          pb = b;
        }
        pend_j = j;
        (void)u0; (void)u1; (void)dst;
#else
        if (DEPTH) {
          wave_sum10_halving_pk(A01, A23, A45, A67, A89, lane_bit3, u0, u1);
          if (octet_lead) dst[octet_val_pk] = u0;
          if (u1_lead) dst[u1_slot] = u1;
        } else {
          wave_sum9_halving_pk(A01, A23, A45, A67, A89.x, lane_bit3, u0, u1);
          if (octet_lead) st_pk[12 * j] = u0;
          if (u1_lead) st_pk[12 * j] = u1;
        }
#endif
        A01 = A23 = A45 = A67 = A89 = gsr_f2{0.f, 0.f};
        return;
      }
      if (DEPTH) wave_sum10_halving(acc0, acc1, acc2, acc3, acc4, acc5, acc6, acc7, acc8, acc9, lane_bit3, u0, u1);
      else wave_sum9_halving(acc0, acc1, acc2, acc3, acc4, acc5, acc6, acc7, acc8, lane_bit3, u0, u1);
      if (octet_lead) dst[octet_val] = u0;
      if (u1_lead) dst[u1_slot] = u1;
      acc0 = acc1 = acc2 = acc3 = acc4 = acc5 = acc6 = acc7 = acc8 = 0.f;
      if (DEPTH) acc9 = 0.f;
    };
    if (MASK) {
      // Only the entries some live sub-block can be reached by are looked at, at all: the others (55 % of a 1080p frame's visits
      // blend nothing - their pixels finished in front of them) cost a scalar bit scan instead of a trip through the loop.
      uint64_t todo = BALLOT(mymask != 0u);
      if (todo != 0ull) {
        int j = (int)__builtin_ctzll(todo);
        float4 a = s0v[j], bb = s1v[j];
        while (true) {
          todo &= todo - 1ull;
          const int jn = todo != 0ull ? (int)__builtin_ctzll(todo) : j;
          const uint32_t emask = (uint32_t)__builtin_amdgcn_readlane((int)mymask, j);   // SGPR
          const bool any = visit(j, a, bb, emask);
          // the next record is requested behind the last use of this one, into the same registers (see the unmasked loop)
          __builtin_amdgcn_sched_barrier(0);
          a = s0v[jn];
          bb = s1v[jn];
          __builtin_amdgcn_sched_barrier(0);
#if BWD_LDS_REDUCE
          consume();                 // the previous hit's sums have come back from LDS while this entry was looked at
#endif
          if (any) reduce(j);
          if (todo == 0ull) break;
          j = jn;
        }
#if BWD_LDS_REDUCE
        consume();
#endif
      }
    } else {
      float4 a = s0v[0], bb = s1v[0];
      for (int j = 0; j < n; j++) {
        const bool any = visit(j, a, bb, 0xFu);
        // The next entry's record is requested HERE, behind the last use of this one and into the same registers (a rotating
        // pair of register sets cost four v_mov_b64 per entry); the reduction below, or the other waves, cover the LDS latency.
        __builtin_amdgcn_sched_barrier(0);
        a = s0v[j + 1];
        bb = s1v[j + 1];
        __builtin_amdgcn_sched_barrier(0);
        if (any) reduce(j);
      }
    }
    __syncthreads();
    // (the nine sums are zero between entries; saying so here lets their registers go free across the staging code above)
    acc0 = acc1 = acc2 = acc3 = acc4 = acc5 = acc6 = acc7 = acc8 = acc9 = 0.f;
    A01 = A23 = A45 = A67 = A89 = gsr_f2{0.f, 0.f};
    // flush the batch: 3 float4 per entry, at the entry's emission slot (grouped per Gaussian for k_preprocess_bwd)
    for (int q = lane; q < n * GSR_IGRAD_F4; q += 64) {
      const int j = q / GSR_IGRAD_F4, part = q - j * GSR_IGRAD_F4;
      const int e = toDo - 1 - (b * BWD1_BATCH + j);
      const uint32_t slot = slot_of_pos[range.x + e];
      igrad[(size_t)GSR_IGRAD_F4 * slot + part] = outb[q];
      if (flags_on && part == 0) iflags[slot] = 1;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// (round 4) k_render_bwd_tile_mx: the same walk, the per-(tile, Gaussian) sums taken by the MATRIX pipe.
//
// What the cross-lane reduction computes is a contraction: for one list entry, the ten record values are
//     sum over the tile's 256 pixels of  h(p) * {1, dx, dy, dx^2, dx dy, dy^2}   and   w(p) * dL/dC_c(p),
// and with d = mean - pixel the first six are a fixed linear map (applied once per entry) of the moments of h against the PIXEL
// basis {1, X, Y, X^2, XY, Y^2} - which does not depend on the entry and is separable in X and Y.  A lane owns pixel
// (X, Y) = (x7 + 8 (s & 1), y7 + 8 (s >> 1)), s = sub-block, x7 = lane & 7, y7 = lane >> 3; v_mfma_f32_16x16x4_f32 reads ONE
// f32 per lane as A[i = lane & 15][k = lane >> 4] (an exact k-ordered fma chain, MI355X_MICROARCH.md "FP32-input MFMA"), so with
//   stage 1   D1[i][j] += h_s(i, k) * B_s[k][j]      one MFMA per blended sub-block, A = the lane's h itself (no accumulation
//                                                    instructions at all), B_s = the Y-side basis of sub-block s (constants)
//             D1[i][5 + c] += C_c(i, k)              one MFMA per colour channel: the lane's sum of w * dL/dC_c over its pixels
//   stage 2   D2[i'][j] = sum_i G[i'][i] * D1[i][j]  four MFMAs (D1's registers ARE the B operands: row = 4 (lane >> 4) + reg),
//                                                    G = the X-side basis
// D2 holds every product moment: the 27-instruction v_permlane / DPP tree, its nine zeroing moves and six of the eight
// accumulation instructions per blended sub-block become 4 + 3 + (blended sub-blocks) MFMAs, each of which holds the vector issue
// port for 8 cycles (two plain instructions) and otherwise runs beside the other waves' VALU work.  Coordinates are taken about
// the tile centre (|X'|, |Y'| <= 7.5, all basis values exact in f32) to keep the cancellation of the final map small:
//   X' = Xi' + Xs',  Xi' = (i & 7) - 3.5, Xs' = 8 (s & 1) - 4;     Y' = Yi' + Yks',  Yi' = (i >> 3) - 0.5, Yks' = 2 k + 8 (s >> 1) - 7
//   columns j: 0: 1, 1: Yks', 2: Xs', 3: Yks'^2, 4: Xs' Yks', 5..8: colour / inverse-depth sums;  rows i': 0: 1, 1: Xi', 2: Yi',
//   3: Xi'^2, 4: Xi' Yi'   (Xs'^2 = 16 and Yi'^2 = 1/4 are constants).
// Lanes 0..2 store rows 0..3 of columns 0..2 (one ds_write_b128), lanes 3..8 and 16 one float each: 19 floats per entry in LDS; once
// per batch lane j turns entry j's 19 numbers into its gradient record (gsr_mx_record) and stores it at the emission slot.
// Every sum is still taken in a fixed order (bitwise reproducible); it is a DIFFERENT order than the swap tree's, so this form agrees
// with the other two to rounding, not bit for bit (tests: parity against the float64 oracle at the same bar, and the basis / lane maps
// against exact integer data, tests/test_parity_gpu.py::test_matrix_pipe_reduction_primitive).
//
// MEASURED OUT (opt-in, GSR_BWD_REDUCE=mfma): 27 % fewer vector instructions per launch (1.88e8 -> 1.38e8 + 1.38e7 MFMAs) and
// 0.52 ms against the tree's 0.42 at C3.  The premise was wrong for THIS data type: on gfx950 an FP32-input MFMA executes on the
// vector FP32 datapath - SQ_VALU_MFMA_COEXEC_CYCLES = 0 for the launch, and tools/mfma_coexec_microbench.hip shows a v_fma_f32
// wave taking exactly its own time PLUS the 32 cycles of every v_mfma_f32_16x16x4_f32 a second wave of its SIMD issues (a bf16
// MFMA costs it ~6) - so each of the ~8.3 MFMAs per entry is worth eight plain instructions, more than the tree it replaces.  A
// bf16 form would need every f32 value split in three (5-6 instructions each): no gain left.  Kept as evidence and for its tests.
// ---------------------------------------------------------------------------------------------------------------
typedef float gsr_f4v __attribute__((ext_vector_type(4)));
#define GSR_MFMA4(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)
#define GSR_MX_SLOT_F4 5            // 19 floats per entry, padded to 80 B

struct gsr_mx_basis {
  float Bs[4];   // stage 1, B operand of sub-block s:  B_s[k = lane >> 4][j = lane & 15]
  float Bc[4];   // stage 1, B operand of colour channel c: 1 in column 5 + c
  float G[4];    // stage 2, A operand of step r:  G[i' = lane & 15][i = 4 (lane >> 4) + r]
  int qofs;      // byte offset of this lane's piece inside an entry's LDS slot (lanes 0..2: 16 lane; 3..8: 48 + 4 (lane - 3); 16: 72)
  bool lane_a, lane_b;
};

__device__ __forceinline__ void gsr_mx_basis_init(const int lane, gsr_mx_basis& K) {
  const int j = lane & 15, k = lane >> 4;
#pragma unroll
  for (int s = 0; s < 4; s++) {
    const float yks = (float)(2 * k + 8 * (s >> 1) - 7), xs = (float)(8 * (s & 1) - 4);
    K.Bs[s] = j == 0 ? 1.0f : j == 1 ? yks : j == 2 ? xs : j == 3 ? yks * yks : j == 4 ? xs * yks : 0.0f;
  }
#pragma unroll
  for (int c = 0; c < 4; c++) K.Bc[c] = j == 5 + c ? 1.0f : 0.0f;
#pragma unroll
  for (int r = 0; r < 4; r++) {
    const int i = 4 * k + r;
    const float xi = (float)(i & 7) - 3.5f, yi = (float)(i >> 3) - 0.5f;
    K.G[r] = j == 0 ? 1.0f : j == 1 ? xi : j == 2 ? yi : j == 3 ? xi * xi : j == 4 ? xi * yi : 0.0f;
  }
  K.qofs = lane < 3 ? 16 * lane : lane < 9 ? 48 + 4 * (lane - 3) : 72;
  K.lane_a = lane < 3;
  K.lane_b = ((0x101F8ull >> lane) & 1ull) != 0ull;   // lanes 3..8 and 16
}

// stage 2 + the entry's 19 numbers into its LDS slot (slot = base of the entry's 80 bytes)
__device__ __forceinline__ void gsr_mx_finish(const gsr_f4v D1, const gsr_mx_basis& K, char* slot) {
  const gsr_f4v zero = {0.f, 0.f, 0.f, 0.f};
  gsr_f4v D2 = GSR_MFMA4(K.G[0], D1[0], zero);
  D2 = GSR_MFMA4(K.G[1], D1[1], D2);
  D2 = GSR_MFMA4(K.G[2], D1[2], D2);
  D2 = GSR_MFMA4(K.G[3], D1[3], D2);
  char* dst = slot + K.qofs;
  if (K.lane_a) *reinterpret_cast<float4*>(dst) = make_float4(D2[0], D2[1], D2[2], D2[3]);
  if (K.lane_b) *reinterpret_cast<float*>(dst) = D2[0];
}

// an entry's 19 numbers -> its gradient record; (mux, muy) = the Gaussian's 2-D mean relative to the tile CENTRE (tile origin + 7.5)
__device__ __forceinline__ void gsr_mx_record(const float4 q0, const float4 q1, const float4 q2, const float4 q3, const float4 q4,
                                              const float mux, const float muy, float4& o0, float4& o1, float4& o2) {
  // q0 = column 0 (1): rows (1, Xi', Yi', Xi'^2); q1 = column 1 (Yks'); q2 = column 2 (Xs'); q3 = (Q[1][Yks'^2], Q[1][Xs'Yks'], c0, c1);
  // q4 = (c2, c3, Q[Xi'Yi'][1], -)
  const float M00 = q0.x;
  const float M10 = q0.y + q2.x;                                   // sum h X'
  const float M01 = q0.z + q1.x;                                   // sum h Y'
  const float M20 = __builtin_fmaf(16.0f, M00, __builtin_fmaf(2.0f, q2.y, q0.w));
  const float M02 = __builtin_fmaf(0.25f, M00, __builtin_fmaf(2.0f, q1.z, q3.x));
  const float M11 = (q1.y + q4.z) + (q3.y + q2.z);
  // d = mean - pixel = mu - X'
  o0.x = __builtin_fmaf(mux, M00, -M10);
  o0.y = __builtin_fmaf(muy, M00, -M01);
  o0.z = __builtin_fmaf(mux, __builtin_fmaf(mux, M00, -2.0f * M10), M20);
  o0.w = __builtin_fmaf(mux, __builtin_fmaf(muy, M00, -M01), __builtin_fmaf(-muy, M10, M11));
  o1.x = __builtin_fmaf(muy, __builtin_fmaf(muy, M00, -2.0f * M01), M02);
  o1.y = M00;
  o1.z = q3.z;
  o1.w = q3.w;
  o2.x = q4.x;
  o2.y = q4.y;
  o2.z = 0.f;
  o2.w = 0.f;
}

// test hook (tests/test_parity_gpu.py::test_matrix_pipe_reduction_primitive): in = h[4][64] (sub-block s, lane), c[4][64] (the lane's
// four channel sums), mu[2]; out = the 10 record values, through the very stage-1 / stage-2 / store / record code of the kernel
__global__ void k_debug_mx_reduce(const float* __restrict__ in, float* __restrict__ out) {
  __shared__ float4 q[GSR_MX_SLOT_F4];
  const int lane = threadIdx.x & 63;
  gsr_mx_basis K;
  gsr_mx_basis_init(lane, K);
  if (lane < GSR_MX_SLOT_F4) q[lane] = make_float4(0.f, 0.f, 0.f, 0.f);
  __syncthreads();
  gsr_f4v D1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int s = 0; s < 4; s++) D1 = GSR_MFMA4(in[s * 64 + lane], K.Bs[s], D1);
#pragma unroll
  for (int c = 0; c < 4; c++) D1 = GSR_MFMA4(in[(4 + c) * 64 + lane], K.Bc[c], D1);
  gsr_mx_finish(D1, K, reinterpret_cast<char*>(q));
  __syncthreads();
  if (lane == 0) {
    float4 o0, o1, o2;
    gsr_mx_record(q[0], q[1], q[2], q[3], q[4], in[512], in[513], o0, o1, o2);
    out[0] = o0.x; out[1] = o0.y; out[2] = o0.z; out[3] = o0.w; out[4] = o1.x; out[5] = o1.y; out[6] = o1.z; out[7] = o1.w;
    out[8] = o2.x; out[9] = o2.y;
  }
}

extern "C" int gsr_debug_mx_reduce(const float* in514, float* out10, void* stream) {
  hipLaunchKernelGGL(k_debug_mx_reduce, dim3(1), dim3(64), 0, (hipStream_t)stream, in514, out10);
  return gsr_launch_status("debug matrix-pipe reduce");
}

#ifndef BWD_MX_WAVES
#define BWD_MX_WAVES 4
#endif
template <bool DEPTH>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(BWD_MX_WAVES, 8))) void k_render_bwd_tile_mx(
    int W, int H, int grid_x, const uint2* __restrict__ ranges, const uint32_t* __restrict__ point_list,
    const float4* __restrict__ rec, const float* __restrict__ bg, const float* __restrict__ final_T,
    const uint32_t* __restrict__ n_contrib, const float* __restrict__ dL_dpix, const float* __restrict__ dL_dinvdepth,
    const uint32_t* __restrict__ slot_of_pos, float4* __restrict__ igrad, const uint32_t* __restrict__ n_dev, uint32_t cap,
    int prio1, int prio2, int prio3, uint32_t flags_min_r) {
  __shared__ float4 s0[BWD1_BATCH + 2], s1[BWD1_BATCH + 2], s2[BWD1_BATCH];
  if (gsr_overflowed(n_dev, cap)) return;
  unsigned char* const iflags = gsr_igrad_flags(igrad, cap);
  const bool flags_on = gsr_flags_on(n_dev, cap, flags_min_r);   // (grid-uniform: validity bytes instead of zero records, gsr_common.h)
  __shared__ float4 qbuf[BWD1_BATCH * GSR_MX_SLOT_F4];
  const int tile = blockIdx.x;
  const int tile_x = tile % grid_x, tile_y = tile / grid_x;
  const int lane = threadIdx.x;
  const uint2 range = ranges[tile];
  const int len = (int)(range.y - range.x);
  if (len == 0) return;

  const size_t N = (size_t)W * H;
  const float bg0 = bg[0], bg1 = bg[1], bg2 = bg[2];
  const float px0 = (float)(tile_x * GSR_TILE + (lane & 7)), py0 = (float)(tile_y * GSR_TILE + (lane >> 3));
  float T[4], S[4], gp2[4], gd[4], nTb[4];
  gsr_f2 gp01[4];
  int last[4];
  int sub_last[4];
#pragma unroll
  for (int s = 0; s < 4; s++) {
    const int px = tile_x * GSR_TILE + (s & 1) * 8 + (lane & 7), py = tile_y * GSR_TILE + (s >> 1) * 8 + (lane >> 3);
    const bool inside = px < W && py < H;
    const size_t pix = (size_t)py * W + px;
    T[s] = inside ? final_T[pix] : 0.f;
    last[s] = inside ? (int)n_contrib[pix] : 0;
    const float g0 = inside ? dL_dpix[pix] : 0.f, g1 = inside ? dL_dpix[N + pix] : 0.f;
    gp01[s] = gsr_f2{g0, g1};
    gp2[s] = inside ? dL_dpix[2 * N + pix] : 0.f;
    gd[s] = (DEPTH && inside) ? dL_dinvdepth[pix] : 0.f;
    nTb[s] = -T[s] * (bg0 * g0 + bg1 * g1 + bg2 * gp2[s]);
    S[s] = 0.f;
    int m = last[s];
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) m = max(m, __shfl_xor(m, d, 64));
    sub_last[s] = __builtin_amdgcn_readfirstlane(m);
  }
  const int toDo = min(len, max(max(sub_last[0], sub_last[1]), max(sub_last[2], sub_last[3])));
  if (prio1 != -2) {   // issue priority for long walks (see k_render_bwd_tile)
    int t1 = prio1, t2 = prio2, t3 = prio3;
    if (prio1 < 0) {
      const uint32_t sp = ((uint32_t)lane * 2654435761u + (uint32_t)tile * 40503u + 12345u) % (uint32_t)N;
      float m = (float)n_contrib[sp];
#pragma unroll
      for (int d = 32; d > 0; d >>= 1) m += __shfl_xor(m, d, 64);
      m *= (1.0f / 64.0f);
      t1 = (int)(1.4f * m); t2 = (int)(1.65f * m); t3 = (int)(1.9f * m);
    }
    if (toDo >= t3) __builtin_amdgcn_s_setprio(3);
    else if (toDo >= t2) __builtin_amdgcn_s_setprio(2);
    else if (toDo >= t1) __builtin_amdgcn_s_setprio(1);
  }
  const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int i = toDo + lane; i < len; i += 64) {   // entries behind the deepest contributor: zero records
    const uint32_t zslot = slot_of_pos[range.x + i];
    if (flags_on) iflags[zslot] = 0;      // (no record: gsr_common.h GSR_IGRAD_F4)
    else {
      float4* dst = igrad + (size_t)GSR_IGRAD_F4 * zslot;
      dst[0] = dst[1] = dst[2] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  int vzero;
  asm volatile("v_mov_b32 %0, 0" : "=v"(vzero));
  const float4 *s0v = s0 + vzero, *s1v = s1 + vzero, *s2v = s2 + vzero;
  gsr_mx_basis K;
  gsr_mx_basis_init(lane, K);
  const float cx = (float)(tile_x * GSR_TILE) + 7.5f, cy = (float)(tile_y * GSR_TILE) + 7.5f;
  const gsr_f4v zero4 = {0.f, 0.f, 0.f, 0.f};
  gsr_f4v D1 = zero4;
  gsr_f2 C01 = {0.f, 0.f}, C23 = {0.f, 0.f};

  const int rounds = (toDo + BWD1_BATCH - 1) / BWD1_BATCH;
  for (int b = 0; b < rounds; b++) {
    __syncthreads();
    const int e_idx = toDo - 1 - (b * BWD1_BATCH + lane);
    uint32_t mymask = 0u;
    if (e_idx >= 0) {
      const uint32_t id32 = point_list[range.x + e_idx];
      if (id32 != 0xFFFFFFFFu) {
        const size_t id = id32;
        const float4 r0 = rec[3 * id + 0], r1 = rec[3 * id + 1];
        s0[lane] = r0;
        s1[lane] = r1;
        s2[lane] = rec[3 * id + 2];
        mymask = gsr_subblock_mask(r0, r1, (float)(tile_x * GSR_TILE), (float)(tile_y * GSR_TILE));
#pragma unroll
        for (int s = 0; s < 4; s++)
          if (e_idx + 1 > sub_last[s]) mymask &= ~(1u << s);
      } else {
        s0[lane] = z4;
        s1[lane] = make_float4(0.f, 0.f, 3.0e38f, 0.f);
        s2[lane] = z4;
      }
    }
    __syncthreads();
    const int n = min(BWD1_BATCH, toDo - b * BWD1_BATCH);
    uint64_t done = 0ull;                               // (scalar) entries of this batch that left numbers in qbuf
    uint64_t todo = BALLOT(mymask != 0u);
    if (todo != 0ull) {
      int j = (int)__builtin_ctzll(todo);
      float4 a = s0v[j], bb = s1v[j];
      while (true) {
        todo &= todo - 1ull;
        const int jn = todo != 0ull ? (int)__builtin_ctzll(todo) : j;
        const uint32_t emask = (uint32_t)__builtin_amdgcn_readlane((int)mymask, j);
        const int entry1 = toDo - (b * BWD1_BATCH + j);
        bool any = false;
        const float dx0 = a.x - px0, dy0 = a.y - py0;
        float4 c;
        bool have_c = false;
#pragma unroll
        for (int s = 0; s < 4; s++) {
          if (!(emask & (1u << s))) continue;
          const float dx = dx0 - (float)((s & 1) * 8), dy = dy0 - (float)((s >> 1) * 8);
          const float power = gsr_power2(a, bb, dx, dy);
          const float G = __builtin_amdgcn_exp2f(power);
          const float alpha = fminf(0.99f, bb.y * G);
          const bool ok = entry1 <= last[s] && power <= 0.0f && alpha >= ALPHA_MIN;
          if ((BALLOT(entry1 <= last[s]) & BALLOT(power <= 0.0f) & BALLOT(alpha >= ALPHA_MIN)) == 0ull) continue;
          if (!have_c) { c = s2v[j]; have_c = true; }
          any = true;
          const float a_e = ok ? alpha : 0.f;
          const float G_e = ok ? G : 0.f;
          const float rcp = __builtin_amdgcn_rcpf(1.0f - a_e);
          T[s] = T[s] * rcp;
          const float dch = a_e * T[s];
          float cg = bb.w * gp01[s].x + c.x * gp01[s].y + c.y * gp2[s];
          if (DEPTH) cg += c.z * gd[s];
          const float diff = cg - S[s];
          S[s] = __builtin_fmaf(a_e, diff, S[s]);
          const float dL_dalpha = diff * T[s] + nTb[s] * rcp;
          const float v5 = G_e * dL_dalpha;               // this pixel's dL/dopacity_eff (see k_render_bwd_tile)
          D1 = GSR_MFMA4(v5, K.Bs[s], D1);
          C01 = __builtin_elementwise_fma(gsr_f2{dch, dch}, gp01[s], C01);
          C23.x = __builtin_fmaf(dch, gp2[s], C23.x);
          if (DEPTH) C23.y = __builtin_fmaf(dch, gd[s], C23.y);
        }
        __builtin_amdgcn_sched_barrier(0);
        a = s0v[jn];
        bb = s1v[jn];
        __builtin_amdgcn_sched_barrier(0);
        if (any) {
          D1 = GSR_MFMA4(C01.x, K.Bc[0], D1);
          D1 = GSR_MFMA4(C01.y, K.Bc[1], D1);
          D1 = GSR_MFMA4(C23.x, K.Bc[2], D1);
          if (DEPTH) D1 = GSR_MFMA4(C23.y, K.Bc[3], D1);
          gsr_mx_finish(D1, K, reinterpret_cast<char*>(qbuf) + j * (GSR_MX_SLOT_F4 * 16));
          done |= 1ull << j;
          D1 = zero4;
          C01 = C23 = gsr_f2{0.f, 0.f};
        }
        if (todo == 0ull) break;
        j = jn;
      }
    }
    __syncthreads();
    D1 = zero4;
    C01 = C23 = gsr_f2{0.f, 0.f};
    // flush the batch: lane j turns entry j's numbers into its record, at the entry's emission slot
    if (lane < n) {
      const int e = toDo - 1 - (b * BWD1_BATCH + lane);
      const uint32_t slot = slot_of_pos[range.x + e];
      float4 o0 = z4, o1 = z4, o2 = z4;
      if ((done >> lane) & 1ull) {
        const float4* q = qbuf + GSR_MX_SLOT_F4 * lane;
        const float4 m = s0[lane];
        gsr_mx_record(q[0], q[1], q[2], q[3], q[4], m.x - cx, m.y - cy, o0, o1, o2);
      }
      float4* dst = igrad + (size_t)GSR_IGRAD_F4 * slot;
      dst[0] = o0; dst[1] = o1; dst[2] = o2;
      if (flags_on) iflags[slot] = 1;
    }
  }
}

void gsr_launch_render_fwd(const gsr_settings* s, int tiles, int grid_x, const uint2* ranges,
                           const uint32_t* point_list, const float4* rec, float* out_color, float* out_invdepth,
                           float* final_T, uint32_t* n_contrib, const uint32_t* status_src, uint32_t* status_dst,
                           uint32_t* tile_cutoff, const uint32_t* depth_key, const uint32_t* culled_any, uint32_t frame_tag,
                           uint32_t* meta, uint32_t* walk_cnt, uint32_t* walk_list, uint32_t* walk_of_tile, hipStream_t st) {
  // cut-off margin of the depth-truncated lists: 1.75 x the entries a tile needed + 48 (measured, profiles/r04_tile_cull.txt: 1.25 x + 16
  // flags 54 % of the frames of a run that trains from scratch, 1.5 x + 32 1 %, 1.75 x + 48 none; C3 and the 2 x splats scene)
  unsigned mq8 = 448, madd = 48;
  if (const char* mg = getenv("GSR_CULL_MARGIN")) sscanf(mg, "%u,%u", &mq8, &madd);
  // GSR_FWD_MASK=1 selects the masked walk (measured: 0.208 against 0.179 ms at C3, profiles/r04_fwd_mask_ab.txt - not the default)
  const char* mk = getenv("GSR_FWD_MASK");          // (read per call: the tests switch inside one process)
  if (!(mk && !strcmp(mk, "1")))
    GSR_LAUNCH("render_fwd", (k_render_fwd<false, false>), dim3(tiles), dim3(256), 0, st, s->image_width, s->image_height, grid_x,
               ranges, point_list, rec, s->bg, out_color, out_invdepth, final_T, n_contrib, (uint32_t*)nullptr, status_src,
               status_dst, tile_cutoff, depth_key, culled_any, frame_tag, meta, mq8, madd, walk_cnt, walk_list, walk_of_tile);
  else
    GSR_LAUNCH("render_fwd", (k_render_fwd<false, true>), dim3(tiles), dim3(256), 0, st, s->image_width, s->image_height, grid_x,
               ranges, point_list, rec, s->bg, out_color, out_invdepth, final_T, n_contrib, (uint32_t*)nullptr, status_src,
               status_dst, tile_cutoff, depth_key, culled_any, frame_tag, meta, mq8, madd, walk_cnt, walk_list, walk_of_tile);
}

void gsr_launch_count_pairs(const gsr_settings* s, int tiles, int grid_x, const uint2* ranges, const uint32_t* point_list,
                            const float4* rec, uint32_t* pairs, hipStream_t st) {
  // (the instrumented build counts the entries the PUBLISHED loop evaluates per pixel - the unit of SURVEY 8(d)'s FLOP model -
  // i.e. the unmasked walk; the masked production kernel evaluates fewer)
  hipLaunchKernelGGL((k_render_fwd<true, false>), dim3(tiles), dim3(256), 0, st, s->image_width, s->image_height, grid_x, ranges,
                     point_list, rec, s->bg, (float*)nullptr, (float*)nullptr, (float*)nullptr, (uint32_t*)nullptr, pairs,
                     (const uint32_t*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr, (const uint32_t*)nullptr,
                     (const uint32_t*)nullptr, 0u, (uint32_t*)nullptr, 0u, 0u, (uint32_t*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr);
}

void gsr_launch_render_bwd(const gsr_settings* s, int tiles, int grid_x, const uint2* ranges,
                           const uint32_t* point_list, const float4* rec, const float* final_T,
                           const uint32_t* n_contrib, const float* dL_dpix, const float* dL_dinvdepth,
                           const uint32_t* slot_of_pos, float4* igrad, const uint32_t* n_dev, uint32_t cap,
                           const uint32_t* walk_cnt, const uint32_t* walk_list, const uint32_t* walk_of_tile, hipStream_t st) {
  // GSR_BWD_LPT=0: tiles in index order (the A/B of the walk classes, profiles/r04_bwd_lpt_ab.txt)
  if (const char* lp = getenv("GSR_BWD_LPT")) {
    if (lp[0] == '0') walk_cnt = nullptr;
  }
  // One wave per tile needs enough tiles to keep 1024 SIMDs busy: below ~6 tiles per SIMD (720p: 3600 tiles) the four-waves-per-tile form
  // (same results up to summation order inside a tile) has the shorter critical path.  GSR_BWD_FORM=quad|tile forces one.
  const char* form = getenv("GSR_BWD_FORM");          // (read per call: the tests switch forms inside one process)
  const char* mk = getenv("GSR_BWD_MASK");
  const bool quad = form ? !strcmp(form, "quad") : tiles < 6000;
  // (the walk order pays where a launch runs in several rounds of resident workgroups; a grid that is resident at once - 256 CUs x 5
  // four-wave workgroups - has no late starters to reorder: 256 tiles 0.054 -> 0.056 ms with it, 3600 tiles 0.166 -> 0.150)
  // ... nor does a launch of many rounds gain what its record gathers lose in locality (4K, 32 400 tiles = 6.3 rounds of one-wave
  // workgroups: 1.381 -> 1.407 ms with the walk order); GSR_BWD_LPT=1 forces the order for any grid
  if (tiles <= 1280 || tiles > 24000) {
    const char* lp = getenv("GSR_BWD_LPT");
    if (!(lp && lp[0] == '1')) walk_cnt = nullptr;
  }
  const bool mask = !(mk && !strcmp(mk, "0"));
  // issue priority for long walks: thresholds from the frame itself (default), GSR_BWD_PRIO="t1,t2,t3" fixes them, "0" = none
  int p1 = -1, p2 = -1, p3 = -1;
  if (const char* pr = getenv("GSR_BWD_PRIO")) {
    if (sscanf(pr, "%d,%d,%d", &p1, &p2, &p3) != 3) p1 = p2 = p3 = -2;
  }
  // GSR_BWD_REDUCE=mfma (opt-in, measured SLOWER: 0.52 against 0.42 ms at C3, profiles/r04_bwd_mx_ab.txt) takes the sums with
  // v_mfma_f32_16x16x4_f32 (k_render_bwd_tile_mx); the default is the v_permlane / DPP halving tree
  const char* red = getenv("GSR_BWD_REDUCE");
  const bool mx = mask && red && !strcmp(red, "mfma");
  if ((!quad || (form && !strcmp(form, "tile"))) && mx) {
    if (dL_dinvdepth)
      GSR_LAUNCH("render_bwd", (k_render_bwd_tile_mx<true>), dim3(tiles), dim3(64), 0, st, s->image_width, s->image_height, grid_x,
                 ranges, point_list, rec, s->bg, final_T, n_contrib, dL_dpix, dL_dinvdepth, slot_of_pos, igrad, n_dev, cap, p1, p2, p3, g_gsr_flags_min_r);
    else
      GSR_LAUNCH("render_bwd", (k_render_bwd_tile_mx<false>), dim3(tiles), dim3(64), 0, st, s->image_width, s->image_height, grid_x,
                 ranges, point_list, rec, s->bg, final_T, n_contrib, dL_dpix, dL_dinvdepth, slot_of_pos, igrad, n_dev, cap, p1, p2, p3, g_gsr_flags_min_r);
    return;
  }
  if (!quad || (form && !strcmp(form, "tile"))) {
#define GSR_BWD_TILE_LAUNCH(D, M)                                                                                          \
  GSR_LAUNCH("render_bwd", (k_render_bwd_tile<D, M>), dim3(tiles), dim3(64), 0, st, s->image_width, s->image_height, grid_x, \
             ranges, point_list, rec, s->bg, final_T, n_contrib, dL_dpix, dL_dinvdepth, slot_of_pos, igrad, n_dev, cap, p1, p2, p3, \
             walk_cnt, walk_list, walk_of_tile, g_gsr_flags_min_r)
    if (dL_dinvdepth) {
      if (mask) GSR_BWD_TILE_LAUNCH(true, true); else GSR_BWD_TILE_LAUNCH(true, false);
    } else {
      if (mask) GSR_BWD_TILE_LAUNCH(false, true); else GSR_BWD_TILE_LAUNCH(false, false);
    }
#undef GSR_BWD_TILE_LAUNCH
    return;
  }
  if (dL_dinvdepth)
    GSR_LAUNCH("render_bwd", k_render_bwd<true>, dim3(tiles), dim3(256), 0, st, s->image_width, s->image_height,
               grid_x, ranges, point_list, rec, s->bg, final_T, n_contrib, dL_dpix, dL_dinvdepth, slot_of_pos, igrad, n_dev, cap,
               walk_cnt, walk_list, walk_of_tile, g_gsr_flags_min_r);
  else
    GSR_LAUNCH("render_bwd", k_render_bwd<false>, dim3(tiles), dim3(256), 0, st, s->image_width, s->image_height,
               grid_x, ranges, point_list, rec, s->bg, final_T, n_contrib, dL_dpix, dL_dinvdepth, slot_of_pos, igrad, n_dev, cap,
               walk_cnt, walk_list, walk_of_tile, g_gsr_flags_min_r);
}
