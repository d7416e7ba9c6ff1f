// tools/experiments/bwd_gauss.hip - EXPERIMENT (round 3): the "per-Gaussian" / bucketed form of the compositing
// backward - what the reference's accelerated rasterizer branch (3dgs_accel, reference README.md:508) is built around - against
// the product's k_render_bwd_tile, on the state buffers of a finished product forward.  NOT part of libgsr_hip.so.
//
// Form: one WAVE per (tile, bucket of 64 consecutive list entries); lane = Gaussian; the wave walks the tile's 256 pixels, lane l
// working on pixel t - l at step t, so a pixel's running state (transmittance T and a = (colour accumulated so far) . dL/dpixel)
// enters at lane 0 from a per-bucket checkpoint and is handed from lane to lane (DPP wave_shr:1) in depth order.  Every lane
// keeps its Gaussian's nine gradient sums in registers: NO cross-lane reduction, perfectly balanced waves.  With
//   C = sum_j c_j alpha_j T_j + T_final bg,   A_j = sum_{k<=j} c_k alpha_k T_k:
//   dL/dalpha_j = T_j (c_j . g) - (C . g - A_j . g) / (1 - alpha_j)                     (g = dL/dpixel)
// so only the two scalars (T, a = A . g) travel.  The checkpoints (T, a at every 64th entry, per pixel) come from a pre-pass
// that replays the forward per tile (k_bg_checkpoints; the product would write them in its forward instead: +8 B per pixel and
// bucket) - its time is reported separately and NOT charged to the challenger.
#include <hip/hip_runtime.h>
#include <stdint.h>

#define TILE 16
#define ALPHA_MIN (1.0f / 255.0f)

__device__ __forceinline__ float power2(const float4& r0, const float4& r1, float dx, float dy) {
  const float t = __builtin_fmaf(r0.w, dy, r0.z * dx);
  return __builtin_fmaf(r1.x * dy, dy, t * dx);
}

// ---- pre-pass: per tile, thread = pixel: replay the forward, store (T, a) at every 64th entry; tile_todo = deepest contributor
__global__ __launch_bounds__(256) void k_bg_checkpoints(int W, int H, int grid_x, const uint2* __restrict__ ranges,
                                                        const uint32_t* __restrict__ point_list,
                                                        const float4* __restrict__ rec, const uint32_t* __restrict__ n_contrib,
                                                        const float* __restrict__ dL_dpix, float2* __restrict__ ckpt,
                                                        uint32_t* __restrict__ tile_todo, uint32_t* __restrict__ worklist,
                                                        uint32_t* __restrict__ work_count) {
  __shared__ float4 s0[256], s1[256], s2[256];
  __shared__ int s_max;
  __shared__ uint32_t s_base;
  const int tile = blockIdx.x, tid = threadIdx.x;
  const int px = (tile % grid_x) * TILE + (tid & 15), py = (tile / grid_x) * TILE + (tid >> 4);
  const bool inside = px < W && py < H;
  const uint2 range = ranges[tile];
  const int len = (int)(range.y - range.x);
  const size_t N = (size_t)W * H, pix = (size_t)py * W + px;
  const int last = inside ? (int)n_contrib[pix] : 0;
  const float g0 = inside ? dL_dpix[pix] : 0.f, g1 = inside ? dL_dpix[N + pix] : 0.f, g2 = inside ? dL_dpix[2 * N + pix] : 0.f;
  if (tid == 0) s_max = 0;
  __syncthreads();
  atomicMax(&s_max, last);
  __syncthreads();
  const int toDo = min(len, s_max);
  const int nb = (toDo + 63) >> 6;
  if (tid == 0) {
    tile_todo[tile] = (uint32_t)toDo;
    s_base = nb ? atomicAdd(work_count, (uint32_t)nb) : 0u;
  }
  __syncthreads();
  if (tid < nb) worklist[s_base + tid] = ((uint32_t)tile << 8) | (uint32_t)tid;     // (tile, bucket)
  const size_t cbase = ((size_t)(range.x >> 6) + (size_t)tile) * 256;                // this tile's first checkpoint row
  float T = 1.f, a = 0.f;
  const float pxf = (float)px, pyf = (float)py;
  for (int r0 = 0; r0 < toDo; r0 += 256) {
    __syncthreads();
    if (r0 + tid < toDo) {
      const uint32_t id = point_list[range.x + r0 + tid];
      if (id != 0xFFFFFFFFu) { s0[tid] = rec[3 * (size_t)id]; s1[tid] = rec[3 * (size_t)id + 1]; s2[tid] = rec[3 * (size_t)id + 2]; }
      else { s0[tid] = make_float4(0, 0, 0, 0); s1[tid] = make_float4(0, 0, 3.0e38f, 0); s2[tid] = make_float4(0, 0, 0, 0); }
    }
    __syncthreads();
    const int n = min(256, toDo - r0);
    for (int j = 0; j < n; j++) {
      const int e = r0 + j;
      if ((e & 63) == 0) ckpt[cbase + (size_t)(e >> 6) * 256 + tid] = make_float2(T, a);
      if (e + 1 > last) continue;
      const float4 A = s0[j], B = s1[j];
      const float dx = A.x - pxf, dy = A.y - pyf;
      const float power = power2(A, B, dx, dy);
      if (power > 0.f || power < B.z) continue;
      const float alpha = fminf(0.99f, B.y * __builtin_amdgcn_exp2f(power));
      if (alpha < ALPHA_MIN) continue;
      const float4 Cc = s2[j];
      const float cg = B.w * g0 + Cc.x * g1 + Cc.y * g2;
      a = __builtin_fmaf(alpha * T, cg, a);
      T *= (1.f - alpha);
    }
  }
}

template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v, float old) {
  return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(v), CTRL, 0xf, 0xf, false));
}

// ---- the challenger: one wave per (tile, bucket); lane = Gaussian; 256 + 63 steps; nine sums in registers
__global__ __launch_bounds__(64) void k_bwd_gauss(int W, int H, int grid_x, const uint2* __restrict__ ranges,
                                                  const uint32_t* __restrict__ point_list, const float4* __restrict__ rec,
                                                  const float* __restrict__ out_color, const uint32_t* __restrict__ n_contrib,
                                                  const float* __restrict__ dL_dpix, const float2* __restrict__ ckpt,
                                                  const uint32_t* __restrict__ tile_todo, const uint32_t* __restrict__ worklist,
                                                  const uint32_t* __restrict__ work_count, float4* __restrict__ igrad_pos) {
  __shared__ float4 pixg[256];          // (g0, g1, g2, C . g) of the tile's pixels
  __shared__ float2 pixs[256];          // checkpoint (T, a) of this bucket
  __shared__ int pixn[256];             // n_contrib
  __shared__ uint8_t act[256 + 64];     // COMPACT: the pixels still compositing when this bucket starts (n_contrib > 64 b), in order
  if (blockIdx.x >= *work_count) return;
  const uint32_t wk = worklist[blockIdx.x];
  const int tile = (int)(wk >> 8), b = (int)(wk & 255u), lane = threadIdx.x;
  const int tx = (tile % grid_x) * TILE, ty = (tile / grid_x) * TILE;
  const uint2 range = ranges[tile];
  const int toDo = (int)tile_todo[tile];
  const size_t N = (size_t)W * H;
  const size_t cbase = ((size_t)(range.x >> 6) + (size_t)tile + (size_t)b) * 256;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const int p = k * 64 + lane;
    const int px = tx + (p & 15), py = ty + (p >> 4);
    const bool inside = px < W && py < H;
    const size_t pix = (size_t)py * W + px;
    const float g0 = inside ? dL_dpix[pix] : 0.f, g1 = inside ? dL_dpix[N + pix] : 0.f, g2 = inside ? dL_dpix[2 * N + pix] : 0.f;
    const float c0 = inside ? out_color[pix] : 0.f, c1 = inside ? out_color[N + pix] : 0.f, c2 = inside ? out_color[2 * N + pix] : 0.f;
    pixg[p] = make_float4(g0, g1, g2, c0 * g0 + c1 * g1 + c2 * g2);
    pixn[p] = inside ? (int)n_contrib[pix] : 0;
    pixs[p] = ckpt[cbase + p];
  }
#ifdef BG_COMPACT
  int n_act = 0;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const int p = k * 64 + lane;
    const int px = tx + (p & 15), py = ty + (p >> 4);
    const bool on = px < W && py < H && (int)n_contrib[(size_t)py * W + px] > 64 * b;
    const unsigned long long m = __builtin_amdgcn_ballot_w64(on);
    if (on) act[n_act + __popcll(m & ((1ull << lane) - 1ull))] = (uint8_t)p;
    n_act += __popcll(m);
  }
  const int n_steps = n_act + 63;
#else
  const int n_act = 256, n_steps = 256 + 63;
#endif
  const int e = b * 64 + lane;                       // this lane's list entry
  const bool have = e < toDo;
  float4 A = make_float4(0, 0, 0, 0), B = make_float4(0, 0, 3.0e38f, 0), Cc = make_float4(0, 0, 0, 0);
  if (have) {
    const uint32_t id = point_list[range.x + e];
    if (id != 0xFFFFFFFFu) { A = rec[3 * (size_t)id]; B = rec[3 * (size_t)id + 1]; Cc = rec[3 * (size_t)id + 2]; }
  }
  __syncthreads();
  const float mx = A.x - (float)tx, my = A.y - (float)ty;     // mean relative to the tile origin
  float acc0 = 0, acc1 = 0, acc2 = 0, acc3 = 0, acc4 = 0, acc5 = 0, acc6 = 0, acc7 = 0, acc8 = 0;
  float T_out = 0.f, a_out = 0.f;
  for (int t = 0; t < n_steps; t++) {
    // state of the pixel this lane works on now = what the previous lane produced at the previous step (lane 0: the checkpoint)
    float T = dpp_mov<0x138>(T_out, 0.f), a = dpp_mov<0x138>(a_out, 0.f);       // wave_shr:1
    const int q = t - lane;
    const bool inrange = q >= 0 && q < n_act;
#ifdef BG_COMPACT
    const int pc = inrange ? (int)act[q] : 0;
#else
    const int pc = inrange ? q : 0;
#endif
    if (lane == 0) { const float2 s = pixs[pc]; T = s.x; a = s.y; }
    const float4 g = pixg[pc];
    const int last = pixn[pc];
    const float dx = mx - (float)(pc & 15), dy = my - (float)(pc >> 4);
    const float power = power2(A, B, dx, dy);
    const float Gx = __builtin_amdgcn_exp2f(power);
    const float alpha = fminf(0.99f, B.y * Gx);
    const bool ok = inrange && have && (e + 1 <= last) && power <= 0.f && power >= B.z && alpha >= ALPHA_MIN;
    const float a_e = ok ? alpha : 0.f, G_e = ok ? Gx : 0.f;
    const float cg = B.w * g.x + Cc.x * g.y + Cc.y * g.z;
    const float w = a_e * T;
    a = __builtin_fmaf(w, cg, a);
    const float rc = __builtin_amdgcn_rcpf(1.f - a_e);
    const float dLda = T * cg - (g.w - a) * rc;
    const float v5 = G_e * dLda, gg = B.y * v5;
    const float t0 = gg * dx, t1 = gg * dy;
    acc0 += t0; acc1 += t1;
    acc2 = __builtin_fmaf(t0, dx, acc2); acc3 = __builtin_fmaf(t0, dy, acc3); acc4 = __builtin_fmaf(t1, dy, acc4);
    acc5 += v5;
    acc6 = __builtin_fmaf(w, g.x, acc6); acc7 = __builtin_fmaf(w, g.y, acc7); acc8 = __builtin_fmaf(w, g.z, acc8);
    T_out = T * (1.f - a_e);
    a_out = a;
  }
  if (have) {      // one 48-B record per list position (the product stores by emission slot; position order is as good for the A/B)
    float4* dst = igrad_pos + 3 * (size_t)(range.x + e);
    dst[0] = make_float4(acc0, acc1, acc2, acc3);
    dst[1] = make_float4(acc4, acc5, acc6, acc7);
    dst[2] = make_float4(acc8, 0.f, 0.f, 0.f);
  }
}

extern "C" int bg_checkpoints(int W, int H, int tiles, int grid_x, const void* ranges, const void* point_list, const void* rec,
                              const void* n_contrib, const void* dL_dpix, void* ckpt, void* tile_todo, void* worklist,
                              void* work_count, void* stream) {
  hipMemsetAsync(work_count, 0, 4, (hipStream_t)stream);
  hipLaunchKernelGGL(k_bg_checkpoints, dim3(tiles), dim3(256), 0, (hipStream_t)stream, W, H, grid_x, (const uint2*)ranges,
                     (const uint32_t*)point_list, (const float4*)rec, (const uint32_t*)n_contrib, (const float*)dL_dpix,
                     (float2*)ckpt, (uint32_t*)tile_todo, (uint32_t*)worklist, (uint32_t*)work_count);
  return (int)hipGetLastError();
}

extern "C" int bg_backward(int W, int H, int grid_x, int max_work, const void* ranges, const void* point_list, const void* rec,
                           const void* out_color, const void* n_contrib, const void* dL_dpix, const void* ckpt,
                           const void* tile_todo, const void* worklist, const void* work_count, void* igrad_pos, void* stream) {
  hipLaunchKernelGGL(k_bwd_gauss, dim3(max_work), dim3(64), 0, (hipStream_t)stream, W, H, grid_x, (const uint2*)ranges,
                     (const uint32_t*)point_list, (const float4*)rec, (const float*)out_color, (const uint32_t*)n_contrib,
                     (const float*)dL_dpix, (const float2*)ckpt, (const uint32_t*)tile_todo, (const uint32_t*)worklist,
                     (const uint32_t*)work_count, (float4*)igrad_pos);
  return (int)hipGetLastError();
}
