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
// (b % 8) * ceil(n / 8) + b / 8: each XCD walks ONE contiguous strip of the (plane, y, x) order and keeps its halos in L2.
// The launch is 1-D with 8 * ceil(n / 8) workgroups; the few past the end return at once.
#define SSIM_XCDS 8
struct SsimTile { int tx, ty, plane; long long lin; bool valid; };
__device__ __forceinline__ SsimTile ssim_tile(int gx, int gy, int planes) {
  const long long n = (long long)gx * gy * planes;
  const long long per = (n + SSIM_XCDS - 1) / SSIM_XCDS;
  const long long lin = (long long)(blockIdx.x % SSIM_XCDS) * per + blockIdx.x / SSIM_XCDS;
  SsimTile t;
  t.lin = lin;
  t.valid = lin < n;
  const long long l = t.valid ? lin : 0;
  t.plane = (int)(l / ((long long)gx * gy));
  const int rem = (int)(l - (long long)t.plane * gx * gy);
  t.ty = rem / gx;
  t.tx = rem - t.ty * gx;
  return t;
}
static inline unsigned ssim_grid(int gx, int gy, int planes) {
  const long long n = (long long)gx * gy * planes;
  return (unsigned)(((n + SSIM_XCDS - 1) / SSIM_XCDS) * SSIM_XCDS);
}

// `partials` != NULL: the kernel also reduces sum(ssim) and sum(|img1-img2|) over its tile into partials[2*block .. +1]
// (fixed in-block order; the host adds the per-block pairs in index order -> deterministic), which is all the fused
// L1 + D-SSIM training loss needs; `ssim_map` may then be NULL.
__global__ __launch_bounds__(256) void k_ssim_fwd(int H, int W, int planes, float C1, float C2, SsimWindow win,
                                                  const float* __restrict__ img1, const float* __restrict__ img2,
                                                  float* __restrict__ ssim_map, float* __restrict__ dm_dmu1,
                                                  float* __restrict__ dm_dsigma1_sq, float* __restrict__ dm_dsigma12,
                                                  float* __restrict__ partials) {
  __shared__ float sx[SHY][SH + 1], sy[SHY][SH + 1];
  __shared__ float hm[5][SHY][ST + 1];
  const int tid = threadIdx.x;
  const SsimTile tile = ssim_tile((W + ST - 1) / ST, (H + STY - 1) / STY, planes);
  if (!tile.valid) return;                         // block-uniform
  const int x0 = tile.tx * ST, y0 = tile.ty * STY;
  const size_t plane = (size_t)tile.plane * H * W;
  // halo load; (row, column) of flat index i is carried from trip to trip (i += 256 = 6 rows + 4 columns at SH = 42)
  {
    int r = tid / SH, c = tid - r * SH;
    for (int i = tid; i < SHY * SH; i += 256) {
      const int gy = y0 + r - SR, gx = x0 + c - SR;
      float a = 0.f, b = 0.f;
      if (gy >= 0 && gy < H && gx >= 0 && gx < W) {
        a = img1[plane + (size_t)gy * W + gx];
        b = img2[plane + (size_t)gy * W + gx];
      }
      sx[r][c] = a;
      sy[r][c] = b;
      r += 256 / SH; c += 256 % SH;
      if (c >= SH) { c -= SH; r++; }
    }
  }
  __syncthreads();
  // horizontal pass: SH rows x ST columns, FOUR adjacent columns per thread: the 14 taps they share are read from LDS once
  // (sliding window) instead of 11 per output
  for (int i = tid; i < SHY * (ST / 4); i += 256) {
    const int r = i / (ST / 4), c0 = (i - r * (ST / 4)) * 4;
    float xa[14], ya[14];
#pragma unroll
    for (int k = 0; k < 14; k++) { xa[k] = sx[r][c0 + k]; ya[k] = sy[r][c0 + k]; }
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
  // vertical pass + SSIM: each thread owns column c and FOUR adjacent rows (14 shared taps per moment)
  float acc_ssim = 0.f, acc_l1 = 0.f;
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
      acc_l1 += fabsf(sx[r + SR][c + SR] - sy[r + SR][c + SR]);
      if (dm_dmu1) {
        // partials holding E[xx], E[yy], E[xy] fixed (sigma's depend on mu1 through -mu1^2, -mu1*mu2)
        dm_dmu1[o] = 2.f * mu2 * (B - A) * inv - m * 2.f * mu1 * rCc + m * 2.f * mu1 * rD;
        dm_dsigma1_sq[o] = -m * rD;
        dm_dsigma12[o] = 2.f * A * inv;
      }
    }
  }
  if (partials) {
    __syncthreads();                               // hm is free now: reuse it for the block reduction
    float* red = &hm[0][0][0];
    red[tid] = acc_ssim;
    red[256 + tid] = acc_l1;
    __syncthreads();
    for (int sft = 128; sft > 0; sft >>= 1) {
      if (tid < sft) {
        red[tid] += red[tid + sft];
        red[256 + tid] += red[256 + tid + sft];
      }
      __syncthreads();
    }
    if (tid == 0) {
      const size_t b = (size_t)tile.lin;           // (plane, y, x) order: the host adds the pairs in this fixed order
      partials[2 * b] = red[0];
      partials[2 * b + 1] = red[256];
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
  const SsimTile tile = ssim_tile((W + ST - 1) / ST, (H + STY - 1) / STY, planes);
  if (!tile.valid) return;                         // block-uniform
  const int x0 = tile.tx * ST, y0 = tile.ty * STY;
  const size_t plane = (size_t)tile.plane * H * W;
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
  GSR_LAUNCH("ssim_fwd", k_ssim_fwd, dim3(ssim_grid(grid.x, grid.y, planes)), dim3(256), 0, st, H, W, planes, C1, C2, win, img1, img2, ssim_map, dm_dmu1,
             dm_dsigma1_sq, dm_dsigma12, (float*)nullptr);
  return gsr_launch_status("ssim forward launch");
}

// Fused training loss of reference train.py:114-121: (1-lambda) * mean|img1-img2| + lambda * (1 - mean(ssim_map)).
// Forward writes the three dm_* maps and per-block partial sums partials[2*nblocks] (ssim sum, L1 sum);
// gsr_fused_loss_blocks() gives nblocks.  The caller adds the partials (in index order) and forms the scalar.
int64_t gsr_fused_loss_blocks(int32_t planes, int32_t H, int32_t W) {
  return (int64_t)((W + ST - 1) / ST) * ((H + STY - 1) / STY) * planes;
}

// one workgroup adds the per-block partial sums in a fixed order and forms the scalar loss
#define LOSS_FIN_THREADS 1024
__global__ __launch_bounds__(LOSS_FIN_THREADS) void k_loss_finalize(const float* __restrict__ partials, long long nblk,
                                                                    float lambda, float inv_n, float* __restrict__ loss) {
  __shared__ float red[2 * LOSS_FIN_THREADS];
  const float2* pairs = reinterpret_cast<const float2*>(partials);   // (ssim sum, L1 sum) per tile, 8-B aligned
  float a = 0.f, b = 0.f;
  for (long long i = threadIdx.x; i < nblk; i += LOSS_FIN_THREADS) {
    const float2 v = pairs[i];
    a += v.x;
    b += v.y;
  }
  red[threadIdx.x] = a;
  red[LOSS_FIN_THREADS + threadIdx.x] = b;
  __syncthreads();
  for (int sft = LOSS_FIN_THREADS / 2; sft > 0; sft >>= 1) {          // fixed tree: deterministic
    if ((int)threadIdx.x < sft) {
      red[threadIdx.x] += red[threadIdx.x + sft];
      red[LOSS_FIN_THREADS + threadIdx.x] += red[LOSS_FIN_THREADS + threadIdx.x + sft];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) loss[0] = (1.0f - lambda) * (red[LOSS_FIN_THREADS] * inv_n) + lambda * (1.0f - red[0] * inv_n);
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
  GSR_LAUNCH("loss_fwd", k_ssim_fwd, dim3(ssim_grid(grid.x, grid.y, planes)), dim3(256), 0, st, H, W, planes, C1, C2, win, img1, img2, (float*)nullptr, dm_dmu1,
             dm_dsigma1_sq, dm_dsigma12, partials);
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
  GSR_LAUNCH("loss_bwd", k_ssim_bwd, dim3(ssim_grid(grid.x, grid.y, planes)), dim3(256), 0, st, H, W, planes, win, img1, img2, (const float*)nullptr,
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
  GSR_LAUNCH("ssim_bwd", k_ssim_bwd, dim3(ssim_grid(grid.x, grid.y, planes)), dim3(256), 0, st, H, W, planes, win, img1, img2, dL_dmap, 0.f, 0.f,
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
