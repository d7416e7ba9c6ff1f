// ssim.hip - fused 11x11 (sigma 1.5) separable SSIM map, forward and backward.
//
// Serves the two interfaces the reference tries (both native modules are absent from the reference tree):
//   utils/loss_utils.py:16-38,162-164   _C.fusedssim(C1,C2,img1,img2), _C.fusedssim_backward(C1,C2,img1,img2,dL_dmap)
//   train.py:31-35,116-117              fused_ssim.fused_ssim(img1, img2)
// and computes exactly what the reference's pure-PyTorch ssim() computes (utils/loss_utils.py:100-159: Gaussian
// window 11, sigma 1.5, zero "same" padding, C1 = 0.01^2, C2 = 0.03^2), which on MI355X costs five MIOpen grouped
// convolutions forward plus their backward (~10.7 ms per 1080p step, profiles/r01_*) against ~0.2 ms here.
//
// One 256-thread workgroup per 32x16 output tile of one image plane (26 KB of LDS -> 6 workgroups per CU; the 32x32 tile
// of the first version needed 42 KB, 3 per CU, and ran 17 % slower): the (32+10)x(16+10) halo of both images is staged in
// LDS once, a horizontal 11-tap pass produces the five moments (x, y, xx, yy, xy) for 42 rows, a vertical pass
// finishes them; everything else is per-pixel arithmetic.  HBM traffic: 8 B in + 16 B out per pixel-channel forward,
// 24 B in + 4 B out backward.
#include "gsr_common.h"

#define ST 32           // output tile width
#ifndef SSIM_STY
#define SSIM_STY 16     // output tile height (multiple of 8: a thread of the vertical pass owns SSIM_STY / 8 adjacent rows)
#endif
#define STY SSIM_STY
#define SR 5            // window radius
#define SH (ST + 2 * SR)    // halo width  42
#define SHY (STY + 2 * SR)  // halo height
#define RPT (STY / 8)       // rows per thread in the vertical passes (256 threads = 32 columns x 8 row groups)

struct SsimWindow { float g[11]; };

static SsimWindow make_window() {
  SsimWindow w;
  double v[11], s = 0.0;
  for (int i = 0; i < 11; i++) { v[i] = exp(-(double)((i - 5) * (i - 5)) / (2.0 * 1.5 * 1.5)); s += v[i]; }
  // the reference builds the 1-D window in fp32 (torch.Tensor of python floats) and normalises in fp32
  float f[11], fs = 0.f;
  for (int i = 0; i < 11; i++) { f[i] = (float)v[i]; fs += f[i]; }
  for (int i = 0; i < 11; i++) w.g[i] = f[i] / fs;
  (void)s;
  return w;
}


// XCD-aware tile order.  Consecutive workgroup ids go round-robin to the 8 XCDs (each with its own L2), so with the natural
// (x, y, plane) order two x-adjacent tiles - which share a 10-pixel halo - never share an L2 and every halo is fetched from
// the fabric again (PMC: k_ssim_bwd read 339 MB for 125 MB of input).  Workgroup b instead takes tile
// (b % 8) * ceil(n / 8) + (b / 8) * tpw: each XCD walks ONE contiguous strip of the (plane, y, x) order and keeps its halos in
// L2.  The launch is 1-D with 8 * ceil(ceil(n / 8) / tpw) workgroups; the few past the end return at once.
#define SSIM_XCDS 8
// A workgroup of the FORWARD walks `tpw` consecutive tiles of its strip and keeps the NEXT tile's halo in flight (registers) while
// it convolves the current one: all workgroups of a launch start together and take the same time, so their load phases and their
// compute phases coincide on a CU (round 4 PMC: half of all wave cycles parked).  With the halo of the next tile in flight, the L1
// term moved into the horizontal pass and the tile sums reduced per wave, a tile costs two barriers instead of seven and the
// parked share drops from 48 % to 30 % of the wave cycles: 69 -> 65.5 us at 1080p, at five waves per SIMD instead of six (88
// VGPRs).  The same treatment of the backward (19 more registers: 8 -> 5 waves per SIMD) ran 53 -> 62-72 us in every variant
// tried (double-buffered LDS, 4 / 5 / 6 waves, 1 / 2 / 4 / 8 tiles): it stays one tile per workgroup.  What bounds both after
// that is the LDS pipe (ACTIVE_INST_LDS 63 % of the kernel per CU beside 45 % VALU), `gpurun_out/r5c`.
#define NTRIP ((SHY * SH + 255) / 256)   // halo elements per thread
#ifndef SSIM_FWD_WAVES
#define SSIM_FWD_WAVES 5                 // registers (88, with the next halo in flight) admit five waves per SIMD; LDS six
#endif
struct SsimTile { int x0, y0; size_t plane; };
struct SsimWalk { long long first; int count; };   // tiles [first, first + count) of the (plane, y, x) order
__device__ __forceinline__ SsimWalk ssim_walk(int gx, int gy, int planes, int tpw) {
  const long long n = (long long)gx * gy * planes;
  const long long per = (n + SSIM_XCDS - 1) / SSIM_XCDS;
  const long long strip = blockIdx.x % SSIM_XCDS, off = (long long)(blockIdx.x / SSIM_XCDS) * tpw;
  SsimWalk w;
  w.first = strip * per + off;
  long long last = strip * per + (off + tpw < per ? off + tpw : per);
  if (last > n) last = n;
  w.count = last > w.first ? (int)(last - w.first) : 0;
  return w;
}
__device__ __forceinline__ SsimTile ssim_tile_at(long long lin, int gx, int gy, int H, int W) {
  const int plane = (int)(lin / ((long long)gx * gy));
  const int rem = (int)(lin - (long long)plane * gx * gy);
  const int ty = rem / gx;
  SsimTile t;
  t.x0 = (rem - ty * gx) * ST;
  t.y0 = ty * STY;
  t.plane = (size_t)plane * H * W;
  return t;
}
static inline unsigned ssim_grid(int gx, int gy, int planes, int tpw) {
  const long long n = (long long)gx * gy * planes;
  const long long per = (n + SSIM_XCDS - 1) / SSIM_XCDS;
  return (unsigned)(((per + tpw - 1) / tpw) * SSIM_XCDS);
}
// tiles per workgroup of the forward: the pipeline pays once the launch runs in several rounds of resident workgroups
static inline int ssim_fwd_tpw(int gx, int gy, int planes) {
  static const int forced = getenv("GSR_SSIM_TPW") ? atoi(getenv("GSR_SSIM_TPW")) : 0;
  if (forced > 0) return forced;
  return (long long)gx * gy * planes >= 4096 ? 2 : 1;
}

// `partials` != NULL: the kernel also reduces sum(ssim) and sum(|img1-img2|) over its tile into partials[2*block .. +1]
// (fixed in-block order; the host adds the per-block pairs in index order -> deterministic), which is all the fused
// L1 + D-SSIM training loss needs; `ssim_map` may then be NULL.
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(SSIM_FWD_WAVES, SSIM_FWD_WAVES))) void k_ssim_fwd(int H, int W, int planes, float C1, float C2, SsimWindow win,
                                                  const float* __restrict__ img1, const float* __restrict__ img2,
                                                  float* __restrict__ ssim_map, float* __restrict__ dm_dmu1,
                                                  float* __restrict__ dm_dsigma1_sq, float* __restrict__ dm_dsigma12,
                                                  float* __restrict__ partials, int tpw) {
  __shared__ float sx[SHY][SH + 1], sy[SHY][SH + 1];
  __shared__ float hm[5][SHY][ST + 1];
  __shared__ float wsum[4][2];                     // per wave: (ssim sum, L1 sum) of the tile just finished
  const int tid0 = threadIdx.x;
  const int gxn = (W + ST - 1) / ST, gyn = (H + STY - 1) / STY;
  const SsimWalk walk = ssim_walk(gxn, gyn, planes, tpw);
  if (walk.count <= 0) return;                     // block-uniform
  // halo of one tile: NTRIP elements per thread; (row, column) of flat index i is carried from trip to trip (i += 256 = 6 rows +
  // 4 columns at SH = 42)
  float ra[NTRIP], rb[NTRIP];
  auto fetch = [&](long long lin, int tid) {
    const SsimTile t = ssim_tile_at(lin, gxn, gyn, H, W);
    int r = tid / SH, c = tid - r * SH;
#pragma unroll
    for (int k = 0; k < NTRIP; k++) {
      const int gy = t.y0 + r - SR, gx = t.x0 + c - SR;
      float a = 0.f, b = 0.f;
      if (tid + 256 * k < SHY * SH && gy >= 0 && gy < H && gx >= 0 && gx < W) {
        a = (img1 + t.plane)[gy * W + gx];           // uniform plane base + 32-bit offset (H * W < 2^31, checked by the host)
        b = (img2 + t.plane)[gy * W + gx];
      }
      ra[k] = a;
      rb[k] = b;
      r += 256 / SH; c += 256 % SH;
      if (c >= SH) { c -= SH; r++; }
    }
  };
  fetch(walk.first, tid0);
  for (int ti = 0; ti < walk.count; ti++) {
    // the thread index is re-read per tile behind an opaque move: everything derived from it (LDS and image offsets of five code
    // sections) would otherwise be hoisted out of this loop and held in registers across it (125 VGPRs instead of 80)
    int tid = tid0;
    asm volatile("" : "+v"(tid));
    const long long lin = walk.first + ti;
    const SsimTile tile = ssim_tile_at(lin, gxn, gyn, H, W);
    const int x0 = tile.x0, y0 = tile.y0;
    const size_t plane = tile.plane;
    {
      int r = tid / SH, c = tid - r * SH;
#pragma unroll
      for (int k = 0; k < NTRIP; k++) {
        if (tid + 256 * k < SHY * SH) { sx[r][c] = ra[k]; sy[r][c] = rb[k]; }
        r += 256 / SH; c += 256 % SH;
        if (c >= SH) { c -= SH; r++; }
      }
    }
    __syncthreads();
    if (ti > 0 && tid == 0 && partials) {            // the previous tile's sums (its waves wrote them before this barrier)
      partials[2 * (size_t)(lin - 1)] = (wsum[0][0] + wsum[1][0]) + (wsum[2][0] + wsum[3][0]);
      partials[2 * (size_t)(lin - 1) + 1] = (wsum[0][1] + wsum[1][1]) + (wsum[2][1] + wsum[3][1]);
    }
    if (ti + 1 < walk.count) fetch(lin + 1, tid);    // in flight while this tile is convolved
    // horizontal pass: SH rows x ST columns, FOUR adjacent columns per thread: the 14 taps they share are read from LDS once
    // (sliding window) instead of 11 per output.  The L1 term is summed here too (taps 5..8 of an interior row ARE the item's
    // four pixels): the staged images are then dead behind this pass and the next tile's may land while the vertical pass runs.
    float acc_ssim = 0.f, acc_l1 = 0.f;
    for (int i = tid; i < SHY * (ST / 4); i += 256) {
      const int r = i / (ST / 4), c0 = (i - r * (ST / 4)) * 4;
      float xa[14], ya[14];
#pragma unroll
      for (int k = 0; k < 14; k++) { xa[k] = sx[r][c0 + k]; ya[k] = sy[r][c0 + k]; }
      if (r >= SR && r < SR + STY && y0 + r - SR < H) {
#pragma unroll
        for (int j = 0; j < 4; j++)
          if (x0 + c0 + j < W) acc_l1 += fabsf(xa[SR + j] - ya[SR + j]);
      }
#pragma unroll
      for (int j = 0; j < 4; j++) {
        float m1 = 0.f, m2 = 0.f, s11 = 0.f, s22 = 0.f, s12 = 0.f;
#pragma unroll
        for (int k = 0; k < 11; k++) {
          const float w = win.g[k], a = xa[j + k], b = ya[j + k];
          m1 += w * a; m2 += w * b; s11 += w * a * a; s22 += w * b * b; s12 += w * a * b;
        }
        hm[0][r][c0 + j] = m1; hm[1][r][c0 + j] = m2; hm[2][r][c0 + j] = s11; hm[3][r][c0 + j] = s22; hm[4][r][c0 + j] = s12;
      }
    }
    __syncthreads();
    // vertical pass + SSIM: each thread owns column c and RPT adjacent rows (RPT + 10 shared taps per moment)
    {
      const int c = tid & (ST - 1), r0 = (tid / ST) * RPT;
      float col[5][RPT + 10];
#pragma unroll
      for (int q = 0; q < 5; q++)
#pragma unroll
        for (int k = 0; k < RPT + 10; k++) col[q][k] = hm[q][r0 + k][c];
#pragma unroll
      for (int j = 0; j < RPT; j++) {
        const int r = r0 + j;
        const int gy = y0 + r, gx = x0 + c;
        if (gy >= H || gx >= W) continue;
        float mu1 = 0.f, mu2 = 0.f, e11 = 0.f, e22 = 0.f, e12 = 0.f;
#pragma unroll
        for (int k = 0; k < 11; k++) {
          const float w = win.g[k];
          mu1 += w * col[0][j + k]; mu2 += w * col[1][j + k]; e11 += w * col[2][j + k];
          e22 += w * col[3][j + k]; e12 += w * col[4][j + k];
        }
        const float mu1_sq = mu1 * mu1, mu2_sq = mu2 * mu2, mu12 = mu1 * mu2;
        const float sigma1_sq = e11 - mu1_sq, sigma2_sq = e22 - mu2_sq, sigma12 = e12 - mu12;
        const float A = 2.f * mu12 + C1, B = 2.f * sigma12 + C2;
        const float Cc = mu1_sq + mu2_sq + C1, D = sigma1_sq + sigma2_sq + C2;
        // (v_rcp_f32, 1 ulp: three IEEE divisions per pixel-channel were ~10 % of this kernel's instructions)
        const float rCc = __builtin_amdgcn_rcpf(Cc), rD = __builtin_amdgcn_rcpf(D);
        const float inv = rCc * rD;
        const float m = A * B * inv;
        const size_t o = plane + (size_t)gy * W + gx;
        if (ssim_map) ssim_map[o] = m;
        acc_ssim += m;
        if (dm_dmu1) {
          // partials holding E[xx], E[yy], E[xy] fixed (sigma's depend on mu1 through -mu1^2, -mu1*mu2)
          dm_dmu1[o] = 2.f * mu2 * (B - A) * inv - m * 2.f * mu1 * rCc + m * 2.f * mu1 * rD;
          dm_dsigma1_sq[o] = -m * rD;
          dm_dsigma12[o] = 2.f * A * inv;
        }
      }
    }
    if (partials) {
      // fixed order: a halving tree inside each wave, the four wave sums added as (0 + 1) + (2 + 3) by thread 0 behind the next
      // barrier of the workgroup (the next tile's first one, or the one below); the finalize kernel adds the tiles in index order
#pragma unroll
      for (int sft = 32; sft > 0; sft >>= 1) {
        acc_ssim += __shfl_down(acc_ssim, sft, 64);
        acc_l1 += __shfl_down(acc_l1, sft, 64);
      }
      if ((tid & 63) == 0) { wsum[tid >> 6][0] = acc_ssim; wsum[tid >> 6][1] = acc_l1; }
    }
  }
  if (partials) {
    __syncthreads();
    if (tid0 == 0) {
      const size_t lin = (size_t)(walk.first + walk.count - 1);
      partials[2 * lin] = (wsum[0][0] + wsum[1][0]) + (wsum[2][0] + wsum[3][0]);
      partials[2 * lin + 1] = (wsum[0][1] + wsum[1][1]) + (wsum[2][1] + wsum[3][1]);
    }
  }
}

// dL/dimg1 = w*(g dm_dmu1) + 2 x (w*(g dm_dsigma1_sq)) + y (w*(g dm_dsigma12)),   g = dL/dmap
// dL_dmap == NULL: the upstream gradient of the map is the constant g_const (mean reduction), and g_l1 * sign(img1 - img2)
// is added (the L1 term of the fused training loss).
__global__ __launch_bounds__(256) void k_ssim_bwd(int H, int W, int planes, SsimWindow win, const float* __restrict__ img1,
                                                  const float* __restrict__ img2, const float* __restrict__ dL_dmap,
                                                  float g_const, float g_l1, const float* __restrict__ upstream,
                                                  const float* __restrict__ dm_dmu1,
                                                  const float* __restrict__ dm_dsigma1_sq,
                                                  const float* __restrict__ dm_dsigma12, float* __restrict__ dL_dimg1) {
  // The horizontal pass runs IN PLACE: item i = (row i / 8, column group i % 8), so the eight threads of a row sit in one wave
  // and a wave owns its rows exclusively; every lane of the wave has its 14 taps in registers before the first result is
  // stored, and no other wave touches those rows.  13 KB of LDS instead of 24 KB: 8 workgroups per CU instead of 6 for a kernel
  // that spends 60 % of its wave cycles waiting on memory (61 -> 54 us at 1080p).
  __shared__ float sa[3][SHY][SH + 1];
  const int tid = threadIdx.x;
  const int gxn = (W + ST - 1) / ST, gyn = (H + STY - 1) / STY;
  const SsimWalk walk = ssim_walk(gxn, gyn, planes, 1);
  if (walk.count <= 0) return;                     // block-uniform
  const SsimTile tile = ssim_tile_at(walk.first, gxn, gyn, H, W);
  const int x0 = tile.x0, y0 = tile.y0;
  const size_t plane = tile.plane;
  if (upstream) {                                  // scalar dL/dloss lives on the device: no host round trip
    const float u = upstream[0];
    g_const *= u;
    g_l1 *= u;
  }
  int lr = tid / SH, lc = tid - lr * SH;             // (row, column) of the flat halo index, carried incrementally
  for (int i = tid; i < SHY * SH; i += 256) {
    const int r = lr, c = lc;
    lr += 256 / SH; lc += 256 % SH;
    if (lc >= SH) { lc -= SH; lr++; }
    const int gy = y0 + r - SR, gx = x0 + c - SR;
    float a = 0.f, b = 0.f, d = 0.f;
    if (gy >= 0 && gy < H && gx >= 0 && gx < W) {
      const size_t o = plane + (size_t)gy * W + gx;
      const float g = dL_dmap ? dL_dmap[o] : g_const;
      a = g * dm_dmu1[o];
      b = g * dm_dsigma1_sq[o];
      d = g * dm_dsigma12[o];
    }
    sa[0][r][c] = a; sa[1][r][c] = b; sa[2][r][c] = d;
  }
  __syncthreads();
  for (int i = tid; i < SHY * (ST / 4); i += 256) {           // horizontal pass, 4 adjacent columns per thread
    const int r = i / (ST / 4), c0 = (i - r * (ST / 4)) * 4;
    float v0[14], v1[14], v2[14];
#pragma unroll
    for (int k = 0; k < 14; k++) { v0[k] = sa[0][r][c0 + k]; v1[k] = sa[1][r][c0 + k]; v2[k] = sa[2][r][c0 + k]; }
    // every load of the wave is issued before its first store (LDS executes a wave's operations in order); the compiler must
    // not sink a load below a store it can prove disjoint for ONE lane - a neighbouring lane's store hits it
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int j = 0; j < 4; j++) {
      float t0 = 0.f, t1 = 0.f, t2 = 0.f;
#pragma unroll
      for (int k = 0; k < 11; k++) {
        const float w = win.g[k];
        t0 += w * v0[j + k]; t1 += w * v1[j + k]; t2 += w * v2[j + k];
      }
      sa[0][r][c0 + j] = t0; sa[1][r][c0 + j] = t1; sa[2][r][c0 + j] = t2;      // in place (see above)
    }
  }
  __syncthreads();
  {                                                            // vertical pass, 4 adjacent rows per thread
    const int c = tid & (ST - 1), r0 = (tid / ST) * RPT;
    float col[3][RPT + 10];
#pragma unroll
    for (int q = 0; q < 3; q++)
#pragma unroll
      for (int k = 0; k < RPT + 10; k++) col[q][k] = sa[q][r0 + k][c];
#pragma unroll
    for (int j = 0; j < RPT; j++) {
      const int gy = y0 + r0 + j, gx = x0 + c;
      if (gy >= H || gx >= W) continue;
      float t0 = 0.f, t1 = 0.f, t2 = 0.f;
#pragma unroll
      for (int k = 0; k < 11; k++) {
        const float w = win.g[k];
        t0 += w * col[0][j + k]; t1 += w * col[1][j + k]; t2 += w * col[2][j + k];
      }
      const size_t o = plane + (size_t)gy * W + gx;
      const float x = img1[o], y = img2[o];
      float out = t0 + 2.f * x * t1 + y * t2;
      if (!dL_dmap) out += g_l1 * ((x > y) ? 1.f : ((x < y) ? -1.f : 0.f));      // torch.sign semantics (0 at equality)
      dL_dimg1[o] = out;
    }
  }
}

extern "C" {

int gsr_fused_ssim_forward(int32_t planes, int32_t H, int32_t W, float C1, float C2, const float* img1,
                           const float* img2, float* ssim_map, float* dm_dmu1, float* dm_dsigma1_sq,
                           float* dm_dsigma12, void* stream) {
  if (planes < 0 || H <= 0 || W <= 0 || !img1 || !img2 || !ssim_map ||
      ((dm_dmu1 != nullptr) != (dm_dsigma1_sq != nullptr)) || ((dm_dmu1 != nullptr) != (dm_dsigma12 != nullptr))) {
    gsr_set_error("fused_ssim_forward: bad arguments");
    return GSR_ERR_INVALID_ARGUMENT;
  }
  if (planes == 0) return 0;
  static const SsimWindow win = make_window();
  hipStream_t st = (hipStream_t)stream;
  dim3 grid((W + ST - 1) / ST, (H + STY - 1) / STY, planes);
  const int tpw = ssim_fwd_tpw(grid.x, grid.y, planes);
  GSR_LAUNCH("ssim_fwd", k_ssim_fwd, dim3(ssim_grid(grid.x, grid.y, planes, tpw)), dim3(256), 0, st, H, W, planes, C1, C2, win, img1, img2,
             ssim_map, dm_dmu1, dm_dsigma1_sq, dm_dsigma12, (float*)nullptr, tpw);
  return gsr_launch_status("ssim forward launch");
}

// Fused training loss of reference train.py:114-121: (1-lambda) * mean|img1-img2| + lambda * (1 - mean(ssim_map)).
// Forward writes the three dm_* maps and per-block partial sums partials[2*nblocks] (ssim sum, L1 sum);
// gsr_fused_loss_blocks() gives nblocks.  The caller adds the partials (in index order) and forms the scalar.
int64_t gsr_fused_loss_blocks(int32_t planes, int32_t H, int32_t W) {
  return (int64_t)((W + ST - 1) / ST) * ((H + STY - 1) / STY) * planes;
}

// one workgroup adds the per-tile partial sums in a fixed order and forms the scalar loss: thread t takes tiles t, t + 1024, ...
// (four loads in flight), a halving tree inside each wave, the sixteen wave sums through LDS and one more tree in wave 0 - one
// barrier instead of the ten of a workgroup-wide LDS tree (8.5 -> ~5 us, most of what is left is the launch itself)
#define LOSS_FIN_THREADS 1024
__global__ __launch_bounds__(LOSS_FIN_THREADS) void k_loss_finalize(const float* __restrict__ partials, long long nblk,
                                                                    float lambda, float inv_n, float* __restrict__ loss) {
  __shared__ float red[2][LOSS_FIN_THREADS / 64];
  const float2* pairs = reinterpret_cast<const float2*>(partials);   // (ssim sum, L1 sum) per tile, 8-B aligned
  float a = 0.f, b = 0.f;
  long long i = threadIdx.x;
  for (; i + 3 * LOSS_FIN_THREADS < nblk; i += 4 * LOSS_FIN_THREADS) {
    const float2 v0 = pairs[i], v1 = pairs[i + LOSS_FIN_THREADS], v2 = pairs[i + 2 * LOSS_FIN_THREADS],
                 v3 = pairs[i + 3 * LOSS_FIN_THREADS];
    a += v0.x; b += v0.y;
    a += v1.x; b += v1.y;
    a += v2.x; b += v2.y;
    a += v3.x; b += v3.y;
  }
  for (; i < nblk; i += LOSS_FIN_THREADS) {
    const float2 v = pairs[i];
    a += v.x;
    b += v.y;
  }
#pragma unroll
  for (int sft = 32; sft > 0; sft >>= 1) {
    a += __shfl_down(a, sft, 64);
    b += __shfl_down(b, sft, 64);
  }
  if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = a; red[1][threadIdx.x >> 6] = b; }
  __syncthreads();
  if (threadIdx.x < 64) {
    a = threadIdx.x < LOSS_FIN_THREADS / 64 ? red[0][threadIdx.x] : 0.f;
    b = threadIdx.x < LOSS_FIN_THREADS / 64 ? red[1][threadIdx.x] : 0.f;
#pragma unroll
    for (int sft = LOSS_FIN_THREADS / 128; sft > 0; sft >>= 1) {
      a += __shfl_down(a, sft, 64);
      b += __shfl_down(b, sft, 64);
    }
    if (threadIdx.x == 0) loss[0] = (1.0f - lambda) * (b * inv_n) + lambda * (1.0f - a * inv_n);
  }
}

int gsr_fused_l1_ssim_forward(int32_t planes, int32_t H, int32_t W, float C1, float C2, float lambda_dssim,
                              const float* img1, const float* img2, float* dm_dmu1, float* dm_dsigma1_sq,
                              float* dm_dsigma12, float* partials, float* loss, void* stream) {
  if (planes <= 0 || H <= 0 || W <= 0 || !img1 || !img2 || !dm_dmu1 || !dm_dsigma1_sq || !dm_dsigma12 || !partials ||
      !loss) {
    gsr_set_error("fused_l1_ssim_forward: bad arguments");
    return GSR_ERR_INVALID_ARGUMENT;
  }
  static const SsimWindow win = make_window();
  hipStream_t st = (hipStream_t)stream;
  dim3 grid((W + ST - 1) / ST, (H + STY - 1) / STY, planes);
  const int tpw = ssim_fwd_tpw(grid.x, grid.y, planes);
  GSR_LAUNCH("loss_fwd", k_ssim_fwd, dim3(ssim_grid(grid.x, grid.y, planes, tpw)), dim3(256), 0, st, H, W, planes, C1, C2, win, img1, img2,
             (float*)nullptr, dm_dmu1, dm_dsigma1_sq, dm_dsigma12, partials, tpw);
  const long long nblk = (long long)grid.x * grid.y * grid.z;
  const float inv_n = 1.0f / ((float)planes * (float)H * (float)W);
  GSR_LAUNCH("loss_finalize", k_loss_finalize, dim3(1), dim3(LOSS_FIN_THREADS), 0, st, (const float*)partials, nblk, lambda_dssim,
             inv_n, loss);
  return gsr_launch_status("fused loss forward launch");
}

// dL/dimg1 = upstream[0] * dloss/dimg1 (`upstream`: DEVICE scalar dL/dloss, or NULL for 1).
int gsr_fused_l1_ssim_backward(int32_t planes, int32_t H, int32_t W, float lambda_dssim, const float* img1,
                               const float* img2, const float* upstream, const float* dm_dmu1,
                               const float* dm_dsigma1_sq, const float* dm_dsigma12, float* dL_dimg1, void* stream) {
  if (planes <= 0 || H <= 0 || W <= 0 || !img1 || !img2 || !dm_dmu1 || !dm_dsigma1_sq || !dm_dsigma12 || !dL_dimg1) {
    gsr_set_error("fused_l1_ssim_backward: bad arguments");
    return GSR_ERR_INVALID_ARGUMENT;
  }
  static const SsimWindow win = make_window();
  hipStream_t st = (hipStream_t)stream;
  dim3 grid((W + ST - 1) / ST, (H + STY - 1) / STY, planes);
  const float inv_n = 1.0f / ((float)planes * (float)H * (float)W);
  GSR_LAUNCH("loss_bwd", k_ssim_bwd, dim3(ssim_grid(grid.x, grid.y, planes, 1)), dim3(256), 0, st, H, W, planes, win, img1, img2, (const float*)nullptr,
             -lambda_dssim * inv_n, (1.0f - lambda_dssim) * inv_n, upstream, dm_dmu1, dm_dsigma1_sq, dm_dsigma12,
             dL_dimg1);
  return gsr_launch_status("fused loss backward launch");
}

int gsr_fused_ssim_backward(int32_t planes, int32_t H, int32_t W, const float* img1, const float* img2,
                            const float* dL_dmap, const float* dm_dmu1, const float* dm_dsigma1_sq,
                            const float* dm_dsigma12, float* dL_dimg1, void* stream) {
  if (planes < 0 || H <= 0 || W <= 0 || !img1 || !img2 || !dL_dmap || !dm_dmu1 || !dm_dsigma1_sq || !dm_dsigma12 ||
      !dL_dimg1) {
    gsr_set_error("fused_ssim_backward: bad arguments");
    return GSR_ERR_INVALID_ARGUMENT;
  }
  if (planes == 0) return 0;
  static const SsimWindow win = make_window();
  hipStream_t st = (hipStream_t)stream;
  dim3 grid((W + ST - 1) / ST, (H + STY - 1) / STY, planes);
  GSR_LAUNCH("ssim_bwd", k_ssim_bwd, dim3(ssim_grid(grid.x, grid.y, planes, 1)), dim3(256), 0, st, H, W, planes, win, img1, img2, dL_dmap, 0.f, 0.f,
             (const float*)nullptr, dm_dmu1, dm_dsigma1_sq, dm_dsigma12, dL_dimg1);
  return gsr_launch_status("ssim backward launch");
}

// ---------------------------------------------------------------------------------------------------------------
// weight * mean |(a - b) mask| and its gradient w.r.t. a: the inverse-depth regularisation term of a training step, reference
// train.py:124-132 (`torch.abs((invDepth - mono_invdepth) * depth_mask).mean()`; mask optional) - in torch ~12 element-wise /
// reduction launches forward + backward, 0.13 ms at 4K.
// Forward: per-workgroup partial sums (fixed order inside a workgroup), then one workgroup adds them in index order:
// deterministic.  Backward: upstream[0] * weight * sign((a - b) mask) mask / n (sign(0) = 0 like torch).
// ---------------------------------------------------------------------------------------------------------------
#define L1_BLOCKS 1024
__global__ __launch_bounds__(256) void k_l1_partial(const float* __restrict__ a, const float* __restrict__ b,
                                                    const float* __restrict__ mask, long long n, float* __restrict__ partials) {
  __shared__ float red[256];
  float acc = 0.f;
  const long long n4 = n >> 2;
  const float4* a4 = reinterpret_cast<const float4*>(a);
  const float4* b4 = reinterpret_cast<const float4*>(b);
  const float4* m4 = reinterpret_cast<const float4*>(mask);
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
    const float4 x = a4[i], y = b4[i];
    const float4 m = mask ? m4[i] : make_float4(1.f, 1.f, 1.f, 1.f);
    acc += (fabsf((x.x - y.x) * m.x) + fabsf((x.y - y.y) * m.y)) + (fabsf((x.z - y.z) * m.z) + fabsf((x.w - y.w) * m.w));
  }
  if (blockIdx.x == 0 && (long long)threadIdx.x < (n & 3)) {
    const long long i = 4 * n4 + threadIdx.x;
    acc += fabsf((a[i] - b[i]) * (mask ? mask[i] : 1.f));
  }
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int sft = 128; sft > 0; sft >>= 1) {
    if ((int)threadIdx.x < sft) red[threadIdx.x] += red[threadIdx.x + sft];
    __syncthreads();
  }
  if (threadIdx.x == 0) partials[blockIdx.x] = red[0];
}

__global__ __launch_bounds__(L1_BLOCKS) void k_l1_finalize(const float* __restrict__ partials, int nblk, float scale,
                                                           float* __restrict__ out) {
  __shared__ float red[L1_BLOCKS];
  red[threadIdx.x] = (int)threadIdx.x < nblk ? partials[threadIdx.x] : 0.f;
  __syncthreads();
  for (int sft = L1_BLOCKS / 2; sft > 0; sft >>= 1) {
    if ((int)threadIdx.x < sft) red[threadIdx.x] += red[threadIdx.x + sft];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = red[0] * scale;
}

__global__ __launch_bounds__(256) void k_l1_bwd(const float* __restrict__ a, const float* __restrict__ b,
                                                const float* __restrict__ mask, long long n, const float* __restrict__ upstream,
                                                float scale, float* __restrict__ grad) {
  const float g = (upstream ? upstream[0] : 1.0f) * scale;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const float m = mask ? mask[i] : 1.f;
    const float d = (a[i] - b[i]) * m;
    grad[i] = d > 0.f ? g * m : (d < 0.f ? -g * m : 0.f);
  }
}

// partials: L1_BLOCKS floats of scratch (gsr_l1_mean_blocks()); a, b, mask 16-byte aligned; mask may be NULL
int32_t gsr_l1_mean_blocks(void) { return L1_BLOCKS; }

int gsr_l1_mean_forward(int64_t n, float weight, const float* a, const float* b, const float* mask, float* partials, float* out,
                        void* stream) {
  if (n <= 0 || !a || !b || !partials || !out || (((uintptr_t)a | (uintptr_t)b | (uintptr_t)mask) & 15)) {
    gsr_set_error("l1_mean_forward: bad arguments");
    return GSR_ERR_INVALID_ARGUMENT;
  }
  hipStream_t st = (hipStream_t)stream;
  const int nblk = (int)min((long long)L1_BLOCKS, (long long)((n / 4 + 255) / 256 > 0 ? (n / 4 + 255) / 256 : 1));
  GSR_LAUNCH("l1_partial", k_l1_partial, dim3(nblk), dim3(256), 0, st, a, b, mask, (long long)n, partials);
  GSR_LAUNCH("l1_finalize", k_l1_finalize, dim3(1), dim3(L1_BLOCKS), 0, st, (const float*)partials, nblk, weight / (float)n, out);
  return gsr_launch_status("l1 mean forward launch");
}

int gsr_l1_mean_backward(int64_t n, float weight, const float* a, const float* b, const float* mask, const float* upstream,
                         float* grad, void* stream) {
  if (n <= 0 || !a || !b || !grad) {
    gsr_set_error("l1_mean_backward: bad arguments");
    return GSR_ERR_INVALID_ARGUMENT;
  }
  const int nblk = (int)min((long long)4096, (long long)((n + 255) / 256));
  GSR_LAUNCH("l1_bwd", k_l1_bwd, dim3(nblk), dim3(256), 0, (hipStream_t)stream, a, b, mask, (long long)n, upstream, weight / (float)n, grad);
  return gsr_launch_status("l1 mean backward launch");
}

}  // extern "C"
