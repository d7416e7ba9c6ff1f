// render.hip - 16x16-tile alpha compositing, forward (K6) and backward (K7).
//
// MI355X mapping: one 256-thread workgroup (4 wave64) per tile; wave w owns the 8x8 pixel quadrant w, lane l
// the pixel (l&7, l>>3) of it, so a Gaussian that misses a quadrant is rejected for 64 pixels by ONE
// wave-uniform ballot + branch.  Batches of 256 packed 48-B splat records are staged through LDS (one
// coalesced gather per record) and read back as wave-uniform broadcasts (conflict-free), software-prefetched one
// entry ahead.  A conservative wave-level test `power >= -ln(255 opacity) - margin` rejects a Gaussian for a whole
// quadrant before the exp; survivors take the exact published test, so results are unchanged.
//
// Backward: per-pixel back-to-front replay as published (T recovered by division), but NO global atomics:
// each wave reduces its 64 pixels' contributions with a DPP scan (6 VALU/value), lane 63 adds the wave total
// into a per-entry LDS accumulator, and the block writes one 48-B gradient record per (tile, instance) with
// plain coalesced stores.  The per-Gaussian sum over instances happens in k_preprocess_bwd (deterministic).
#include "gsr_common.h"

#define ALPHA_MIN (1.0f / 255.0f)

// In-place full-wave sums of ten registers; totals valid in lane 63.  gfx9 DPP: a lane whose DPP source is out of
// range (bound_ctrl:0) or whose row is masked keeps its value, so `v_add_f32_dpp v, v, v <ctrl>` accumulates in place
// with no v_mov.  Steps are interleaved over the ten values so no instruction reads a register written by one of
// the two preceding instructions (VALU-write -> DPP-read needs 2 wait states; hipcc pads nothing inside asm).
#define GSR_DPP10(ctrl)                                   \
  "v_add_f32_dpp %0, %0, %0 " ctrl "\n"                   \
  "v_add_f32_dpp %1, %1, %1 " ctrl "\n"                   \
  "v_add_f32_dpp %2, %2, %2 " ctrl "\n"                   \
  "v_add_f32_dpp %3, %3, %3 " ctrl "\n"                   \
  "v_add_f32_dpp %4, %4, %4 " ctrl "\n"                   \
  "v_add_f32_dpp %5, %5, %5 " ctrl "\n"                   \
  "v_add_f32_dpp %6, %6, %6 " ctrl "\n"                   \
  "v_add_f32_dpp %7, %7, %7 " ctrl "\n"                   \
  "v_add_f32_dpp %8, %8, %8 " ctrl "\n"                   \
  "v_add_f32_dpp %9, %9, %9 " ctrl "\n"

#define GSR_DPP9(ctrl)                                    \
  "v_add_f32_dpp %0, %0, %0 " ctrl "\n"                   \
  "v_add_f32_dpp %1, %1, %1 " ctrl "\n"                   \
  "v_add_f32_dpp %2, %2, %2 " ctrl "\n"                   \
  "v_add_f32_dpp %3, %3, %3 " ctrl "\n"                   \
  "v_add_f32_dpp %4, %4, %4 " ctrl "\n"                   \
  "v_add_f32_dpp %5, %5, %5 " ctrl "\n"                   \
  "v_add_f32_dpp %6, %6, %6 " ctrl "\n"                   \
  "v_add_f32_dpp %7, %7, %7 " ctrl "\n"                   \
  "v_add_f32_dpp %8, %8, %8 " ctrl "\n"

__device__ __forceinline__ void wave_sum9_to_lane63(float& v0, float& v1, float& v2, float& v3, float& v4, float& v5,
                                                    float& v6, float& v7, float& v8) {
  asm volatile(
      "s_nop 1\n"
      GSR_DPP9("row_shr:1 row_mask:0xf bank_mask:0xf")
      GSR_DPP9("row_shr:2 row_mask:0xf bank_mask:0xf")
      GSR_DPP9("row_shr:4 row_mask:0xf bank_mask:0xf")
      GSR_DPP9("row_shr:8 row_mask:0xf bank_mask:0xf")
      GSR_DPP9("row_bcast:15 row_mask:0xa bank_mask:0xf")
      GSR_DPP9("row_bcast:31 row_mask:0xc bank_mask:0xf")
      "s_nop 1\n"
      : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7), "+v"(v8));
}

__device__ __forceinline__ void wave_sum10_to_lane63(float& v0, float& v1, float& v2, float& v3, float& v4, float& v5,
                                                     float& v6, float& v7, float& v8, float& v9) {
  asm volatile(
      "s_nop 1\n"
      GSR_DPP10("row_shr:1 row_mask:0xf bank_mask:0xf")
      GSR_DPP10("row_shr:2 row_mask:0xf bank_mask:0xf")
      GSR_DPP10("row_shr:4 row_mask:0xf bank_mask:0xf")
      GSR_DPP10("row_shr:8 row_mask:0xf bank_mask:0xf")
      GSR_DPP10("row_bcast:15 row_mask:0xa bank_mask:0xf")
      GSR_DPP10("row_bcast:31 row_mask:0xc bank_mask:0xf")
      "s_nop 1\n"
      : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7), "+v"(v8), "+v"(v9));
}

#define FWD_BATCH 256
#define BALLOT(p) __builtin_amdgcn_ballot_w64(p)

__global__ __launch_bounds__(256) void k_render_fwd(int W, int H, int grid_x, const uint2* __restrict__ ranges,
                                                    const uint32_t* __restrict__ point_list,
                                                    const float4* __restrict__ rec, const float* __restrict__ bg,
                                                    float* __restrict__ out_color, float* __restrict__ out_invdepth,
                                                    float* __restrict__ final_T, uint32_t* __restrict__ n_contrib) {
  __shared__ float4 s0[FWD_BATCH + 2], s1[FWD_BATCH + 2], s2[FWD_BATCH + 2];  // +2: the prefetch may touch [n+1]
  const int tile = blockIdx.x;
  const int tile_x = tile % grid_x, tile_y = tile / grid_x;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int px = tile_x * GSR_TILE + (w & 1) * 8 + (lane & 7);
  const int py = tile_y * GSR_TILE + (w >> 1) * 8 + (lane >> 3);
  const bool inside = px < W && py < H;
  const float pxf = (float)px, pyf = (float)py;
  const uint2 range = ranges[tile];
  int toDo = (int)(range.y - range.x);
  const int rounds = (toDo + FWD_BATCH - 1) / FWD_BATCH;

  bool done = !inside;
  // A finished (or outside) pixel is moved far away: its power becomes hugely negative and fails the wave-level
  // reject test with no extra instruction in the loop.
  float pxe = done ? 1.0e15f : pxf;
  float T = 1.0f, C0 = 0.f, C1 = 0.f, C2 = 0.f, D = 0.f;
  uint32_t last = 0;

  for (int r = 0; r < rounds; r++, toDo -= FWD_BATCH) {
    if (__syncthreads_count(done) == 256) break;
    const uint32_t progress = range.x + (uint32_t)(r * FWD_BATCH + tid);
    if (progress < range.y) {
      const size_t id = point_list[progress];
      s0[tid] = rec[3 * id + 0];
      s1[tid] = rec[3 * id + 1];
      s2[tid] = rec[3 * id + 2];
    }
    __syncthreads();
    const int n = toDo < FWD_BATCH ? toDo : FWD_BATCH;
    bool wave_live = BALLOT(!done) != 0ull;
    // one list entry against this wave's 64 pixels
    auto step = [&](const float4& a, const float4& b, const int j) __attribute__((always_inline)) {
      const float dx = a.x - pxe, dy = a.y - pyf;
      const float power = -0.5f * (a.z * dx * dx + b.x * dy * dy) - a.w * dx * dy;
      // conservative wave-level reject (b.z = -ln(255 opacity) - margin): no lane can reach alpha >= 1/255
      if (BALLOT(power >= b.z) != 0ull) {
        const float alpha = fminf(0.99f, b.y * __expf(power));
        const bool ok = !done && power <= 0.0f && alpha >= ALPHA_MIN;
        const float4 c = s2[j];
        const float test_T = T * (1.0f - alpha);
        const bool stop = ok && test_T < 0.0001f;    // the stopping Gaussian is NOT blended (A.5)
        const bool blend = ok && !stop;
        const float wgt = blend ? alpha * T : 0.f;
        C0 += b.w * wgt;
        C1 += c.x * wgt;
        C2 += c.y * wgt;
        D += c.z * wgt;
        T = blend ? test_T : T;
        last = blend ? (uint32_t)(r * FWD_BATCH + j + 1) : last;
        done = done || stop;
        pxe = done ? 1.0e15f : pxf;
        wave_live = BALLOT(!done) != 0ull;           // `done` only changes here: whole quadrant saturated -> leave
      }
    };
    // software prefetch, two register sets in ping-pong (entry j+1 / j+2 in flight while j / j+1 is evaluated)
    float4 a0 = s0[0], b0 = s1[0];
    for (int j = 0; j < n && wave_live; j += 2) {
      const float4 a1 = s0[j + 1], b1 = s1[j + 1];
      step(a0, b0, j);
      a0 = s0[j + 2];
      b0 = s1[j + 2];
      if (j + 1 < n && wave_live) step(a1, b1, j + 1);
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

#define BWD_BATCH 128   // entries staged per round in the backward (4 per-wave gradient slabs must fit LDS)

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
                                                    float4* __restrict__ igrad) {
  __shared__ float4 s0[BWD_BATCH + 2], s1[BWD_BATCH + 2], s2[BWD_BATCH + 2];
  // one private slab per wave: no LDS atomics, and the 4 partial sums are added in a FIXED order at flush time,
  // so gradients are bitwise reproducible
  __shared__ float4 slab[4][BWD_BATCH * GSR_IGRAD_F4];
  __shared__ int s_max;

  const int tile = blockIdx.x;
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
  {
    int m = last;
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) m = max(m, __shfl_xor(m, d, 64));
    if (lane == 0) atomicMax(&s_max, m);
  }
  __syncthreads();
  const int toDo = min(len, s_max);
  const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
  // Gradient records are stored by EMISSION SLOT (slot_of_pos = the tile sort's value array), i.e. grouped per
  // Gaussian, so the per-Gaussian sum in k_preprocess_bwd streams contiguous memory; the scattered 48-B stores here
  // are fire-and-forget.
  for (int i = toDo + tid; i < len; i += 256) {
    float4* dst = igrad + (size_t)GSR_IGRAD_F4 * slot_of_pos[range.x + i];
    dst[0] = z4; dst[1] = z4; dst[2] = z4;
  }

  const int rounds = (toDo + BWD_BATCH - 1) / BWD_BATCH;
  float T = T_final;
  float ar0 = 0.f, ar1 = 0.f, ar2 = 0.f, ad = 0.f;        // colour / invdepth accumulated behind
  float lc0 = 0.f, lc1 = 0.f, lc2 = 0.f, ld = 0.f, last_alpha = 0.f;
  const float halfW = 0.5f * W, halfH = 0.5f * H;
  float4* myslab = slab[w];

  for (int b = 0; b < rounds; b++) {
    __syncthreads();
    const int e_idx = toDo - 1 - (b * BWD_BATCH + tid);  // back-to-front staging (threads 0..127)
    if (tid < BWD_BATCH && e_idx >= 0) {
      const size_t id = point_list[range.x + e_idx];
      s0[tid] = rec[3 * id + 0];
      s1[tid] = rec[3 * id + 1];
      s2[tid] = rec[3 * id + 2];
    }
    // each wave clears its own slab (384 float4 / 64 lanes)
#pragma unroll
    for (int i = 0; i < (BWD_BATCH * GSR_IGRAD_F4) / 64; i++) myslab[i * 64 + lane] = z4;
    __syncthreads();
    const int n = min(BWD_BATCH, toDo - b * BWD_BATCH);
    auto step = [&](const float4& a, const float4& bb, const int j) __attribute__((always_inline)) {
      const int entry1 = toDo - (b * BWD_BATCH + j);  // 1-based list position of this entry
      const float dx = a.x - pxf, dy = a.y - pyf;
      const float power = -0.5f * (a.z * dx * dx + bb.x * dy * dy) - a.w * dx * dy;
      const bool pre = entry1 <= last && power >= bb.z;   // conservative wave-level reject (see forward)
      if (BALLOT(pre) == 0ull) return;
      const float G = __expf(power);
      const float alpha = fminf(0.99f, bb.y * G);
      const bool ok = pre && power <= 0.0f && alpha >= ALPHA_MIN;
      if (BALLOT(ok) == 0ull) return;
      const float4 c = s2[j];
      // Lanes that do not blend this entry run the same arithmetic with alpha = G = 0: T, the "accumulated behind"
      // recurrences and every gradient term then stay exactly unchanged / zero, so no per-lane branch is needed.
      const float a_e = ok ? alpha : 0.f;
      const float G_e = ok ? G : 0.f;
      const float rcp = __builtin_amdgcn_rcpf(1.0f - a_e);
      T = T * rcp;
      const float dch = a_e * T;
      ar0 = last_alpha * lc0 + (1.f - last_alpha) * ar0; lc0 = bb.w;
      ar1 = last_alpha * lc1 + (1.f - last_alpha) * ar1; lc1 = c.x;
      ar2 = last_alpha * lc2 + (1.f - last_alpha) * ar2; lc2 = c.y;
      if (DEPTH) { ad = last_alpha * ld + (1.f - last_alpha) * ad; ld = c.z; }
      last_alpha = a_e;
      float dL_dalpha = (bb.w - ar0) * gp0 + (c.x - ar1) * gp1 + (c.y - ar2) * gp2;
      if (DEPTH) dL_dalpha += (c.z - ad) * gd;
      dL_dalpha = dL_dalpha * T + neg_Tf_bg * rcp;
      float v6 = dch * gp0, v7 = dch * gp1, v8 = dch * gp2, v9 = DEPTH ? dch * gd : 0.f;
      const float dL_dG = bb.y * dL_dalpha;
      const float gdx = G_e * dx, gdy = G_e * dy;
      const float dG_ddelx = -gdx * a.z - gdy * a.w;
      const float dG_ddely = -gdy * bb.x - gdx * a.w;
      float v0 = dL_dG * dG_ddelx * halfW;
      float v1 = dL_dG * dG_ddely * halfH;
      float v2 = -0.5f * gdx * dx * dL_dG;
      float v3 = -gdx * dy * dL_dG;
      float v4 = -0.5f * gdy * dy * dL_dG;
      float v5 = G_e * dL_dalpha;
      if (DEPTH) wave_sum10_to_lane63(v0, v1, v2, v3, v4, v5, v6, v7, v8, v9);
      else wave_sum9_to_lane63(v0, v1, v2, v3, v4, v5, v6, v7, v8);
      if (lane == 63) {
        myslab[3 * j + 0] = make_float4(v0, v1, v2, v3);
        myslab[3 * j + 1] = make_float4(v4, v5, v6, v7);
        myslab[3 * j + 2] = make_float4(v8, v9, 0.f, 0.f);
      }
    };
    float4 a0 = s0[0], b0 = s1[0];
    for (int j = 0; j < n; j += 2) {
      const float4 a1 = s0[j + 1], b1 = s1[j + 1];
      step(a0, b0, j);
      a0 = s0[j + 2];
      b0 = s1[j + 2];
      if (j + 1 < n) step(a1, b1, j + 1);
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
      igrad[(size_t)GSR_IGRAD_F4 * slot_of_pos[range.x + e] + part] = r;
    }
  }
}

void gsr_launch_render_fwd(const gsr_settings* s, int tiles, int grid_x, const uint2* ranges,
                           const uint32_t* point_list, const float4* rec, float* out_color, float* out_invdepth,
                           float* final_T, uint32_t* n_contrib, hipStream_t st) {
  GSR_LAUNCH("render_fwd", k_render_fwd, dim3(tiles), dim3(256), 0, st, s->image_width, s->image_height, grid_x,
             ranges, point_list, rec, s->bg, out_color, out_invdepth, final_T, n_contrib);
}

void gsr_launch_render_bwd(const gsr_settings* s, int tiles, int grid_x, const uint2* ranges,
                           const uint32_t* point_list, const float4* rec, const float* final_T,
                           const uint32_t* n_contrib, const float* dL_dpix, const float* dL_dinvdepth,
                           const uint32_t* slot_of_pos, float4* igrad, hipStream_t st) {
  if (dL_dinvdepth)
    GSR_LAUNCH("render_bwd", k_render_bwd<true>, dim3(tiles), dim3(256), 0, st, s->image_width, s->image_height,
               grid_x, ranges, point_list, rec, s->bg, final_T, n_contrib, dL_dpix, dL_dinvdepth, slot_of_pos, igrad);
  else
    GSR_LAUNCH("render_bwd", k_render_bwd<false>, dim3(tiles), dim3(256), 0, st, s->image_width, s->image_height,
               grid_x, ranges, point_list, rec, s->bg, final_T, n_contrib, dL_dpix, dL_dinvdepth, slot_of_pos, igrad);
}
