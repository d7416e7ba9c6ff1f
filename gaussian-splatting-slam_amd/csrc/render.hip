// render.hip - 16x16-tile alpha compositing, forward (K6) and backward (K7).
//
// MI355X mapping: one 256-thread workgroup (4 wave64) per tile; wave w owns the 8x8 pixel quadrant w, lane l
// the pixel (l&7, l>>3) of it, so a Gaussian that misses a quadrant is rejected for 64 pixels by ONE
// wave-uniform ballot + branch.  Batches of 256 packed 48-B splat records are staged through LDS (one
// coalesced gather per record) and read back as wave-uniform broadcasts (conflict-free).
//
// Backward: per-pixel back-to-front replay as published (T recovered by division), but NO global atomics:
// each wave reduces its 64 pixels' contributions with a DPP scan (6 VALU/value), lane 63 adds the wave total
// into a per-entry LDS accumulator, and the block writes one 48-B gradient record per (tile, instance) with
// plain coalesced stores.  The per-Gaussian sum over instances happens in k_preprocess_bwd (deterministic).
#include "gsr_common.h"

#define ALPHA_MIN (1.0f / 255.0f)

__global__ __launch_bounds__(256) void k_render_fwd(int W, int H, int grid_x, const uint2* __restrict__ ranges,
                                                    const uint32_t* __restrict__ point_list,
                                                    const float4* __restrict__ rec, const float* __restrict__ bg,
                                                    float* __restrict__ out_color, float* __restrict__ out_invdepth,
                                                    float* __restrict__ final_T, uint32_t* __restrict__ n_contrib) {
  __shared__ float4 s0[256], s1[256], s2[256];
  const int tile = blockIdx.x;
  const int tile_x = tile % grid_x, tile_y = tile / grid_x;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int px = tile_x * GSR_TILE + (w & 1) * 8 + (lane & 7);
  const int py = tile_y * GSR_TILE + (w >> 1) * 8 + (lane >> 3);
  const bool inside = px < W && py < H;
  const float pxf = (float)px, pyf = (float)py;
  const uint2 range = ranges[tile];
  int toDo = (int)(range.y - range.x);
  const int rounds = (toDo + 255) / 256;

  bool done = !inside;
  float T = 1.0f, C0 = 0.f, C1 = 0.f, C2 = 0.f, D = 0.f;
  uint32_t last = 0;

  for (int r = 0; r < rounds; r++, toDo -= 256) {
    if (__syncthreads_count(done) == 256) break;
    const uint32_t progress = range.x + (uint32_t)(r * 256 + tid);
    if (progress < range.y) {
      const size_t id = point_list[progress];
      s0[tid] = rec[3 * id + 0];
      s1[tid] = rec[3 * id + 1];
      s2[tid] = rec[3 * id + 2];
    }
    __syncthreads();
    const int n = toDo < 256 ? toDo : 256;
    for (int j = 0; j < n; j++) {
      if (__ballot(!done) == 0ull) break;  // whole quadrant saturated
      const float4 a = s0[j];
      const float4 b = s1[j];
      const float dx = a.x - pxf, dy = a.y - pyf;
      const float power = -0.5f * (a.z * dx * dx + b.x * dy * dy) - a.w * dx * dy;
      const float alpha = fminf(0.99f, b.y * __expf(power));
      const bool ok = !done && power <= 0.0f && alpha >= ALPHA_MIN;
      if (__ballot(ok) == 0ull) continue;  // Gaussian misses this quadrant
      if (ok) {
        const float test_T = T * (1.0f - alpha);
        if (test_T < 0.0001f) {
          done = true;  // the stopping Gaussian is NOT blended (A.5)
        } else {
          const float4 c = s2[j];
          const float wgt = alpha * T;
          C0 += b.z * wgt;
          C1 += b.w * wgt;
          C2 += c.x * wgt;
          D += c.y * wgt;
          T = test_T;
          last = (uint32_t)(r * 256 + j + 1);
        }
      }
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

__global__ __launch_bounds__(256) void k_render_bwd(int W, int H, int grid_x, const uint2* __restrict__ ranges,
                                                    const uint32_t* __restrict__ point_list,
                                                    const float4* __restrict__ rec, const float* __restrict__ bg,
                                                    const float* __restrict__ final_T,
                                                    const uint32_t* __restrict__ n_contrib,
                                                    const float* __restrict__ dL_dpix,
                                                    const float* __restrict__ dL_dinvdepth, float4* __restrict__ igrad) {
  __shared__ float4 s0[BWD_BATCH], s1[BWD_BATCH], s2[BWD_BATCH];
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
    if (dL_dinvdepth) gd = dL_dinvdepth[pix];
  }
  const float bg_dot = bg[0] * gp0 + bg[1] * gp1 + bg[2] * gp2;

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
  for (int i = toDo + tid; i < len; i += 256) {
    float4* dst = igrad + (size_t)GSR_IGRAD_F4 * (range.x + i);
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
    for (int j = 0; j < n; j++) {
      const int entry1 = toDo - (b * BWD_BATCH + j);  // 1-based list position of this entry
      const float4 a = s0[j];
      const float4 bb = s1[j];
      const float dx = a.x - pxf, dy = a.y - pyf;
      const float power = -0.5f * (a.z * dx * dx + bb.x * dy * dy) - a.w * dx * dy;
      const float G = __expf(power);
      const float alpha = fminf(0.99f, bb.y * G);
      const bool ok = entry1 <= last && power <= 0.0f && alpha >= ALPHA_MIN;
      if (__ballot(ok) == 0ull) continue;
      const float4 c = s2[j];
      float v0 = 0.f, v1 = 0.f, v2 = 0.f, v3 = 0.f, v4 = 0.f, v5 = 0.f, v6 = 0.f, v7 = 0.f, v8 = 0.f, v9 = 0.f;
      if (ok) {
        T = T / (1.0f - alpha);
        const float dch = alpha * T;
        float dL_dalpha;
        ar0 = last_alpha * lc0 + (1.f - last_alpha) * ar0; lc0 = bb.z;
        ar1 = last_alpha * lc1 + (1.f - last_alpha) * ar1; lc1 = bb.w;
        ar2 = last_alpha * lc2 + (1.f - last_alpha) * ar2; lc2 = c.x;
        ad = last_alpha * ld + (1.f - last_alpha) * ad;    ld = c.y;
        dL_dalpha = (bb.z - ar0) * gp0 + (bb.w - ar1) * gp1 + (c.x - ar2) * gp2 + (c.y - ad) * gd;
        v6 = dch * gp0; v7 = dch * gp1; v8 = dch * gp2; v9 = dch * gd;
        dL_dalpha *= T;
        last_alpha = alpha;
        dL_dalpha += (-T_final / (1.f - alpha)) * bg_dot;
        const float dL_dG = bb.y * dL_dalpha;
        const float gdx = G * dx, gdy = G * dy;
        const float dG_ddelx = -gdx * a.z - gdy * a.w;
        const float dG_ddely = -gdy * bb.x - gdx * a.w;
        v0 = dL_dG * dG_ddelx * halfW;
        v1 = dL_dG * dG_ddely * halfH;
        v2 = -0.5f * gdx * dx * dL_dG;
        v3 = -gdx * dy * dL_dG;
        v4 = -0.5f * gdy * dy * dL_dG;
        v5 = G * dL_dalpha;
      }
      v0 = gsr_wave_sum_to_lane63(v0); v1 = gsr_wave_sum_to_lane63(v1);
      v2 = gsr_wave_sum_to_lane63(v2); v3 = gsr_wave_sum_to_lane63(v3);
      v4 = gsr_wave_sum_to_lane63(v4); v5 = gsr_wave_sum_to_lane63(v5);
      v6 = gsr_wave_sum_to_lane63(v6); v7 = gsr_wave_sum_to_lane63(v7);
      v8 = gsr_wave_sum_to_lane63(v8); v9 = gsr_wave_sum_to_lane63(v9);
      if (lane == 63) {
        myslab[3 * j + 0] = make_float4(v0, v1, v2, v3);
        myslab[3 * j + 1] = make_float4(v4, v5, v6, v7);
        myslab[3 * j + 2] = make_float4(v8, v9, 0.f, 0.f);
      }
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
      igrad[(size_t)GSR_IGRAD_F4 * (range.x + e) + part] = r;
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
                           const uint32_t* n_contrib, const float* dL_dpix, const float* dL_dinvdepth, float4* igrad,
                           hipStream_t st) {
  GSR_LAUNCH("render_bwd", k_render_bwd, dim3(tiles), dim3(256), 0, st, s->image_width, s->image_height, grid_x,
             ranges, point_list, rec, s->bg, final_T, n_contrib, dL_dpix, dL_dinvdepth, igrad);
}
