// preprocess.hip - per-Gaussian stages of the rasterizer (gfx950).
//
//   k_preprocess_fwd : K1 of SURVEY.md 2.3 (near cull, projection, EWA covariance, dilation / AA,
//                      conic, radius, tile rect, SH -> RGB) - writes the packed 48-B splat record.
//   k_preprocess_bwd : K8 + K9 fused, preceded by a deterministic per-Gaussian gather-sum of the
//                      per-instance gradient records written (grouped per Gaussian) by the render backward (no float atomics:
//                      MI355X global float atomics hit 64 different rows at ~0.08 TB/s, plain stores +
//                      a gather pass run at HBM rate and make the gradients bitwise reproducible).
//   k_mark_visible   : K10.
//
// Semantics follow SURVEY.md Appendix A (the published algorithm of
// graphdeco-inria/diff-gaussian-rasterization@9c5c2028, absent from the reference tree) and the boundary
// contract of reference gaussian_renderer/__init__.py:18-121.  SH basis / constants: reference
// utils/sh_utils.py:26-100.  Quaternion -> R and Sigma packing: reference utils/general_utils.py:64-110.
#include <string.h>

#include "gsr_common.h"

// No implicit fma contraction in this file: k_preprocess_bwd exists in three instantiations (plain, Adam folded in, sparse Adam
// folded in) whose gradients must agree bit for bit, and what the compiler contracts depends on how a value is used further
// down.  These kernels are HBM-bound; the few extra multiplies do not show.
#pragma clang fp contract(off)

#define SH_C0 0.28209479177387814f
#define SH_C1 0.4886025119029199f
__constant__ float SH_C2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f,
                               -1.0925484305920792f, 0.5462742152960396f};
__constant__ float SH_C3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f,
                               0.3731763325901154f,  -0.4570457994644658f, 1.445305721320277f,
                               -0.5900435899266435f};

struct PreView {
  float V[16];   // viewmatrix, flat (column-major W2C)
  float PV[16];  // projmatrix, flat
  float cam[3];
};

__device__ __forceinline__ void load_view(const float* vm, const float* pm, const float* campos, PreView& v) {
#pragma unroll
  for (int i = 0; i < 16; i++) {
    v.V[i] = vm[i];
    v.PV[i] = pm[i];
  }
  v.cam[0] = campos[0];
  v.cam[1] = campos[1];
  v.cam[2] = campos[2];
}

// Sigma (6-vector) from scale / rotation (A.2): Sigma = R diag(mod*s)^2 R^T
__device__ __forceinline__ void cov3d_from_sr(const float* s3, const float* q4, float mod, float* cov6,
                                              float* Rout /*9, row-major, may be null*/) {
  const float r = q4[0], x = q4[1], y = q4[2], z = q4[3];
  float R[9] = {1.f - 2.f * (y * y + z * z), 2.f * (x * y - r * z),       2.f * (x * z + r * y),
                2.f * (x * y + r * z),       1.f - 2.f * (x * x + z * z), 2.f * (y * z - r * x),
                2.f * (x * z - r * y),       2.f * (y * z + r * x),       1.f - 2.f * (x * x + y * y)};
  const float sx = mod * s3[0], sy = mod * s3[1], sz = mod * s3[2];
  float L[9];
#pragma unroll
  for (int i = 0; i < 3; i++) {
    L[3 * i + 0] = R[3 * i + 0] * sx;
    L[3 * i + 1] = R[3 * i + 1] * sy;
    L[3 * i + 2] = R[3 * i + 2] * sz;
  }
  cov6[0] = L[0] * L[0] + L[1] * L[1] + L[2] * L[2];
  cov6[1] = L[0] * L[3] + L[1] * L[4] + L[2] * L[5];
  cov6[2] = L[0] * L[6] + L[1] * L[7] + L[2] * L[8];
  cov6[3] = L[3] * L[3] + L[4] * L[4] + L[5] * L[5];
  cov6[4] = L[3] * L[6] + L[4] * L[7] + L[5] * L[8];
  cov6[5] = L[6] * L[6] + L[7] * L[7] + L[8] * L[8];
  if (Rout) {
#pragma unroll
    for (int i = 0; i < 9; i++) Rout[i] = R[i];
  }
}

// EWA projection pieces shared by forward and backward.
struct Ewa {
  float tx, ty, tz;   // (clamped) view-space mean
  bool in_x, in_y;    // clamp masks
  float m0[3], m1[3]; // rows of M = J W
  float a0, b, c0;    // cov2D before dilation
};

__device__ __forceinline__ void ewa_project(const float t[3], const PreView& v, const float* cov6, float fx,
                                            float fy, float tanfovx, float tanfovy, Ewa& e) {
  const float limx = 1.3f * tanfovx, limy = 1.3f * tanfovy;
  const float txtz = t[0] / t[2], tytz = t[1] / t[2];
  e.in_x = !(txtz < -limx || txtz > limx);
  e.in_y = !(tytz < -limy || tytz > limy);
  e.tx = fminf(limx, fmaxf(-limx, txtz)) * t[2];
  e.ty = fminf(limy, fmaxf(-limy, tytz)) * t[2];
  e.tz = t[2];
  const float j00 = fx / e.tz, j02 = -(fx * e.tx) / (e.tz * e.tz);
  const float j11 = fy / e.tz, j12 = -(fy * e.ty) / (e.tz * e.tz);
  // W row k = (V[k], V[4+k], V[8+k])
#pragma unroll
  for (int j = 0; j < 3; j++) {
    e.m0[j] = j00 * v.V[4 * j + 0] + j02 * v.V[4 * j + 2];
    e.m1[j] = j11 * v.V[4 * j + 1] + j12 * v.V[4 * j + 2];
  }
  // Sigma m
  const float s0x = cov6[0] * e.m0[0] + cov6[1] * e.m0[1] + cov6[2] * e.m0[2];
  const float s0y = cov6[1] * e.m0[0] + cov6[3] * e.m0[1] + cov6[4] * e.m0[2];
  const float s0z = cov6[2] * e.m0[0] + cov6[4] * e.m0[1] + cov6[5] * e.m0[2];
  const float s1x = cov6[0] * e.m1[0] + cov6[1] * e.m1[1] + cov6[2] * e.m1[2];
  const float s1y = cov6[1] * e.m1[0] + cov6[3] * e.m1[1] + cov6[4] * e.m1[2];
  const float s1z = cov6[2] * e.m1[0] + cov6[4] * e.m1[1] + cov6[5] * e.m1[2];
  e.a0 = e.m0[0] * s0x + e.m0[1] * s0y + e.m0[2] * s0z;
  e.b = e.m0[0] * s1x + e.m0[1] * s1y + e.m0[2] * s1z;
  e.c0 = e.m1[0] * s1x + e.m1[1] * s1y + e.m1[2] * s1z;
}

// Scale / rotation / opacity of Gaussian idx as the rasterizer consumes them.  raw == 0: the arrays hold activated values
// (the reference's call form).  raw != 0: they hold the model's RAW parameters and the activations of reference
// scene/gaussian_model.py:38-46 are applied on load - exp, F.normalize (x / max(|x|, 1e-12)), sigmoid - so the model needs no
// activation kernels of its own; the backward then returns gradients w.r.t. the raw parameters (chain rule in
// k_preprocess_bwd).  qden = max(|q_raw|, 1e-12) (1 when not raw).
__device__ __forceinline__ void load_scale_rot(const float* __restrict__ scales, const float* __restrict__ rotations,
                                               size_t idx, int raw, float* s3, float* q4, float& qden) {
#pragma unroll
  for (int j = 0; j < 3; j++) s3[j] = scales[3 * idx + j];
#pragma unroll
  for (int j = 0; j < 4; j++) q4[j] = rotations[4 * idx + j];
  qden = 1.0f;
  if (raw) {
#pragma unroll
    for (int j = 0; j < 3; j++) s3[j] = expf(s3[j]);
    qden = fmaxf(sqrtf(q4[0] * q4[0] + q4[1] * q4[1] + q4[2] * q4[2] + q4[3] * q4[3]), 1e-12f);
#pragma unroll
    for (int j = 0; j < 4; j++) q4[j] = q4[j] / qden;
  }
}
__device__ __forceinline__ float load_opacity(const float* __restrict__ opacities, size_t idx, int raw) {
  const float x = opacities[idx];
  return raw ? 1.0f / (1.0f + expf(-x)) : x;
}

__device__ __forceinline__ float sh_coef(const float* dc, const float* shs, int stride, int idx, int k, int c) {
  // coefficient k of channel c of Gaussian idx; with `dc`, band 0 lives there and shs holds k-1
  if (dc) return (k == 0) ? dc[3 * (size_t)idx + c] : shs[((size_t)idx * stride + (k - 1)) * 3 + c];
  return shs[((size_t)idx * stride + k) * 3 + c];
}

// ---- LDS staging of a block's SH rows (north_star: coalesced per-Gaussian attribute loads) -------------------------
// The 256 rows of `shs` that a workgroup needs are one contiguous span of 256*S floats (S = 3*sh_stride; 48 KB at SH
// degree 3).  Reading it as per-thread 12-B pieces at a 192-B stride costs 48 uncoalesced dword loads per lane; instead
// the span is copied with flat 16-B/lane loads into LDS rows of ODD stride Sp = S|1 (a thread walking its own row then
// hits 32 distinct banks across the wave), and the gradient rows go back to HBM the same way.
// (row, column) of flat element e = 4 i is carried incrementally from trip to trip (i advances by 256, e by 1024): one
// runtime division per thread instead of one per 16-B piece.
// `need` (optional; an LDS array, one int per row, > 0 = wanted): rows nobody will read - culled Gaussians, a third of the
// bench scene and most of a room-scale capture - are not fetched; their LDS rows stay undefined.  All the loads of a thread
// are issued before the first LDS store (a predicated load followed by its own store would serialise the fetches).
#define GSR_STAGE_MAX_TRIPS 16     // 256 rows x S floats / 4 / 256 threads = S / 4 <= 16 (can_stage_sh caps S at 63)
template <int BT = 256>
__device__ __forceinline__ void stage_rows_in(const float* __restrict__ src, int nflt, int S, int Sp, float* lds,
                                              const int32_t* need = nullptr) {
  const int n4 = nflt >> 2;
  const int dr = (4 * BT) / S, dc = (4 * BT) - dr * S;                 // (row, column) advance per trip
  const int r0 = (threadIdx.x * 4) / S, c0 = threadIdx.x * 4 - r0 * S;
  float4 v[GSR_STAGE_MAX_TRIPS];
  bool want[GSR_STAGE_MAX_TRIPS];
  {
    int r = r0, c = c0;
#pragma unroll
    for (int it = 0; it < GSR_STAGE_MAX_TRIPS; it++) {
      const int i = threadIdx.x + BT * it;
      want[it] = i < n4;
      if (want[it] && need) want[it] = need[r] > 0 || (c + 3 >= S && (r + 1) * S < nflt && need[r + 1] > 0);   // may straddle 2 rows
      if (want[it]) v[it] = gsr_ld_stream4(reinterpret_cast<const float4*>(src) + i);   // SH rows pass through once per kernel
      r += dr; c += dc;
      if (c >= S) { c -= S; r++; }
    }
  }
  {
    int r = r0, c = c0;
#pragma unroll
    for (int it = 0; it < GSR_STAGE_MAX_TRIPS; it++) {
      if (want[it]) {
        const float vv[4] = {v[it].x, v[it].y, v[it].z, v[it].w};
        int rr = r, cc = c;
#pragma unroll
        for (int k = 0; k < 4; k++) {
          lds[rr * Sp + cc] = vv[k];
          if (++cc == S) { cc = 0; rr++; }
        }
      }
      r += dr; c += dc;
      if (c >= S) { c -= S; r++; }
    }
  }
  for (int e = n4 * 4 + threadIdx.x; e < nflt; e += BT) {
    const int r2 = e / S, c2 = e - r2 * S;
    lds[r2 * Sp + c2] = src[e];
  }
}
template <int BT = 256>
__device__ __forceinline__ void stage_rows_out(float* __restrict__ dst, int nflt, int S, int Sp, const float* lds) {
  const int n4 = nflt >> 2;
  const int dr = (4 * BT) / S, dc = (4 * BT) - dr * S;
  int r = (threadIdx.x * 4) / S, c = threadIdx.x * 4 - r * S;
  for (int i = threadIdx.x; i < n4; i += BT) {
    int rr = r, cc = c;
    float vv[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
      vv[k] = lds[rr * Sp + cc];
      if (++cc == S) { cc = 0; rr++; }
    }
    gsr_st_stream4(reinterpret_cast<float4*>(dst) + i, make_float4(vv[0], vv[1], vv[2], vv[3]));
    r += dr; c += dc;
    if (c >= S) { c -= S; r++; }
  }
  for (int e = n4 * 4 + threadIdx.x; e < nflt; e += BT) {
    const int r2 = e / S, c2 = e - r2 * S;
    dst[e] = lds[r2 * Sp + c2];
  }
}
// coefficient k of channel c of this thread's Gaussian, staged variant
__device__ __forceinline__ float sh_coef_lds(const float* dc, const float* row, int idx, int k, int c) {
  if (dc) return (k == 0) ? dc[3 * (size_t)idx + c] : row[(k - 1) * 3 + c];
  return row[k * 3 + c];
}

__device__ __forceinline__ void sh_basis_eval(int deg, float x, float y, float z, float* b /*16*/) {
  b[0] = SH_C0;
  if (deg > 0) {
    b[1] = -SH_C1 * y;
    b[2] = SH_C1 * z;
    b[3] = -SH_C1 * x;
    if (deg > 1) {
      const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
      b[4] = SH_C2[0] * xy;
      b[5] = SH_C2[1] * yz;
      b[6] = SH_C2[2] * (2.f * zz - xx - yy);
      b[7] = SH_C2[3] * xz;
      b[8] = SH_C2[4] * (xx - yy);
      if (deg > 2) {
        b[9] = SH_C3[0] * y * (3.f * xx - yy);
        b[10] = SH_C3[1] * xy * z;
        b[11] = SH_C3[2] * y * (4.f * zz - xx - yy);
        b[12] = SH_C3[3] * z * (2.f * zz - 3.f * xx - 3.f * yy);
        b[13] = SH_C3[4] * x * (4.f * zz - xx - yy);
        b[14] = SH_C3[5] * z * (xx - yy);
        b[15] = SH_C3[6] * x * (xx - 3.f * yy);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// K1 forward
// ---------------------------------------------------------------------------------------------------
template <bool STAGE>
__global__ __launch_bounds__(256) void k_preprocess_fwd(
    int P, int deg, int sh_stride, const float* __restrict__ means3D, const float* __restrict__ dc,
    const float* __restrict__ shs, const float* __restrict__ colors_precomp, const float* __restrict__ opacities,
    const float* __restrict__ scales, const float* __restrict__ rotations, const float* __restrict__ cov3D_precomp,
    float scale_modifier, const float* __restrict__ viewmatrix, const float* __restrict__ projmatrix,
    const float* __restrict__ campos, int W, int H, float tanfovx, float tanfovy, int prefiltered, int antialiasing,
    int defer_color, int raw_act, int32_t* __restrict__ radii, float4* __restrict__ rec, uint32_t* __restrict__ depth_key,
    uint32_t* __restrict__ order, uint32_t* __restrict__ tiles_touched, ushort4* __restrict__ rect,
    float4* __restrict__ bin_rec,
    uint8_t* __restrict__ clamped, uint32_t* __restrict__ meta, uint32_t* __restrict__ block_sums,
    uint32_t* __restrict__ zero_words, int n_zero_words,
    const uint32_t* __restrict__ tile_cutoff /* nullptr, or per tile the depth bits beyond which nothing is emitted */,
    uint32_t* __restrict__ culled_any /* [tiles]: set to frame_tag where an instance is dropped */, uint32_t frame_tag) {
  extern __shared__ __attribute__((aligned(16))) float sh_lds[];
  // (round 4) the head of the tile sort's scratch - histogram replicas + pass tickets - is cleared HERE, one kernel ahead of
  // k_emit_instances, whose workgroups all add their digit counts to it (binning.hip)
  if (zero_words && blockIdx.x == 0)
    for (int i = threadIdx.x; i < n_zero_words; i += 256) zero_words[i] = 0u;
  __shared__ int32_t need_sh[256];     // STAGE: which SH rows of this workgroup will be evaluated
  __shared__ uint32_t tile_sum[4];
  const int idx = blockIdx.x * 256 + threadIdx.x;
  const int S = 3 * sh_stride, Sp = S | 1;
  const bool in_range = idx < P;
  const float* my_row = sh_lds + threadIdx.x * Sp;

  // ---- phase 1: geometry (no SH needed); what phase 2 needs stays in registers ----
  int32_t out_radius = 0;                       // defaults for a culled Gaussian
  uint32_t out_tiles = 0, out_key = 0xFFFFFFFFu;
  bool on_screen = false;                       // passed the near plane, det != 0, non-empty 3-sigma rectangle
  bool culled_flag = false;                     // prefiltered = true and this Gaussian fails the near-plane test
  float px = 0.f, py = 0.f, sA = 0.f, sB = 0.f, sC = 0.f, op = 0.f, spmin = 0.f, depth = 1.f;
  int cx0 = 0, cy0 = 0, cx1 = 0, cy1 = 0;
  float p[3] = {0.f, 0.f, 0.f};
  if (in_range) {
    PreView v;
    load_view(viewmatrix, projmatrix, campos, v);
    p[0] = means3D[3 * (size_t)idx];
    p[1] = means3D[3 * (size_t)idx + 1];
    p[2] = means3D[3 * (size_t)idx + 2];
    float t[3];
#pragma unroll
    for (int i = 0; i < 3; i++) t[i] = v.V[i] * p[0] + v.V[4 + i] * p[1] + v.V[8 + i] * p[2] + v.V[12 + i];
    depth = t[2];

    if (t[2] > 0.2f) {  // A.1 near-plane cull only
      float hom[4];
#pragma unroll
      for (int i = 0; i < 4; i++) hom[i] = v.PV[i] * p[0] + v.PV[4 + i] * p[1] + v.PV[8 + i] * p[2] + v.PV[12 + i];
      const float pw = 1.0f / (hom[3] + 0.0000001f);
      const float ndcx = hom[0] * pw, ndcy = hom[1] * pw;

      float cov6[6];
      if (cov3D_precomp) {
#pragma unroll
        for (int i = 0; i < 6; i++) cov6[i] = cov3D_precomp[6 * (size_t)idx + i];
      } else {
        float s3[3], q4[4], qden;
        load_scale_rot(scales, rotations, (size_t)idx, raw_act, s3, q4, qden);
        cov3d_from_sr(s3, q4, scale_modifier, cov6, nullptr);
      }
      const float fx = W / (2.0f * tanfovx), fy = H / (2.0f * tanfovy);
      Ewa e;
      ewa_project(t, v, cov6, fx, fy, tanfovx, tanfovy, e);

      // A.3 dilation / AA
      const float det0 = e.a0 * e.c0 - e.b * e.b;
      const float a = e.a0 + 0.3f, c = e.c0 + 0.3f, b = e.b;
      const float det = a * c - b * b;
      float h = 1.0f;
      if (antialiasing) h = sqrtf(fmaxf(0.000025f, det0 / det));
      if (det != 0.0f) {
        const float det_inv = 1.0f / det;
        const float cA = c * det_inv, cB = -b * det_inv, cC = a * det_inv;
        // A.4 extent
        const float mid = 0.5f * (a + c);
        const float root = sqrtf(fmaxf(0.1f, mid * mid - det));
        const float lam = fmaxf(mid + root, mid - root);
        const float radius = ceilf(3.0f * sqrtf(lam));
        px = ((ndcx + 1.0f) * W - 1.0f) * 0.5f;
        py = ((ndcy + 1.0f) * H - 1.0f) * 0.5f;
        const int gx = (W + GSR_TILE - 1) / GSR_TILE, gy = (H + GSR_TILE - 1) / GSR_TILE;
        // C-style truncation then clamp; guard the float->int conversion against huge values
        const float lim = 1.0e9f;
        const int x0 = min(gx, max(0, (int)fminf(lim, fmaxf(-lim, (px - radius) / GSR_TILE))));
        const int y0 = min(gy, max(0, (int)fminf(lim, fmaxf(-lim, (py - radius) / GSR_TILE))));
        const int x1 = min(gx, max(0, (int)fminf(lim, fmaxf(-lim, (px + radius + (GSR_TILE - 1)) / GSR_TILE))));
        const int y1 = min(gy, max(0, (int)fminf(lim, fmaxf(-lim, (py + radius + (GSR_TILE - 1)) / GSR_TILE))));
        const int area = (x1 - x0) * (y1 - y0);
        if (area > 0) {
          on_screen = true;
          op = load_opacity(opacities, (size_t)idx, raw_act) * h;
          // conservative cut-off for the render kernels: alpha = op*exp(power) >= 1/255  <=>  power >= -ln(255 op);
          // the margin (>> fp32 error of power / exp) keeps the test a pure accelerator (exact test follows it)
          const float pmin = (op > 0.f) ? (-logf(255.0f * op) - 0.01f) : 1.0f;
          // Exact tile culling: a pixel can only blend this Gaussian if d^T conic d <= q = -2 pmin.  That ellipse lies in
          // the axis-aligned box |dx| <= sqrt(q * cov_xx), |dy| <= sqrt(q * cov_yy) (cov = dilated 2-D covariance =
          // conic^-1), so tiles of the published 3-sigma rectangle outside the box cannot contribute to the image or to
          // any gradient and are not emitted.  Outputs are unchanged; only the internal instance lists get shorter
          // (-26 % at C3).  `rect` keeps the (shrunk) box; emit applies the per-row test below inside it.
          cx0 = x0; cy0 = y0; cx1 = x1; cy1 = y1;
          {
            const float q = -2.0f * pmin;
            if (q <= 0.f) {
              cx1 = cx0;  // opacity below 1/255: never blended anywhere
            } else {
              const float hx = sqrtf(q * a) * 1.0001f + 0.01f, hy = sqrtf(q * c) * 1.0001f + 0.01f;
              // tile t holds pixel centres 16t .. 16t+15
              const int tx_lo = (int)ceilf(fmaxf(-lim, (px - hx - (GSR_TILE - 1)) / GSR_TILE));
              const int tx_hi = (int)floorf(fminf(lim, (px + hx) / GSR_TILE));
              const int ty_lo = (int)ceilf(fmaxf(-lim, (py - hy - (GSR_TILE - 1)) / GSR_TILE));
              const int ty_hi = (int)floorf(fminf(lim, (py + hy) / GSR_TILE));
              cx0 = max(cx0, tx_lo); cx1 = min(cx1, tx_hi + 1);
              cy0 = max(cy0, ty_lo); cy1 = min(cy1, ty_hi + 1);
            }
          }
          // second stage inside the box: exact per-row column intervals of the ellipse (another -12 % at C3); emit repeats
          // the same computation bit-identically (same stored inputs, same compiled body)
          sA = (-0.5f * GSR_LOG2E) * cA; sB = -GSR_LOG2E * cB; sC = (-0.5f * GSR_LOG2E) * cC;
          spmin = GSR_LOG2E * pmin;
          int kept = 0;
          const uint32_t zbits = __float_as_uint(t[2]);
          for (int ty = cy0; ty < cy1; ty++) {
            const uint32_t iv = gsr_row_interval(px, py, sA, sB, sC, spmin, ty, cx0, cx1);
            if (tile_cutoff) {
              // (round 4) lists truncated by depth: a tile whose every pixel saturated in front of `tile_cutoff[tile]` when this view
              // was last rendered takes no instance behind it - the emission applies the very same comparison
              // (four cut-offs in flight per trip: one dependent L2 round trip per tile made this loop +12 us at C3, +53 us on big splats)
              const int lo = (int)(iv & 0xFFFFu), hi = (int)(iv >> 16);
              const uint32_t* crow = tile_cutoff + ty * gx;
              for (int tx = lo; tx < hi; tx += 4) {
                uint32_t c[4];
#pragma unroll
                for (int u = 0; u < 4; u++) c[u] = tx + u < hi ? crow[tx + u] : 0xFFFFFFFFu;
#pragma unroll
                for (int u = 0; u < 4; u++) {
                  if (tx + u < hi) {
                    if (zbits <= c[u]) kept++;
                    else culled_any[ty * gx + tx + u] = frame_tag;   // the tile's list is not whole in this frame (same-value races are fine)
                  }
                }
              }
            } else {
              kept += (int)(iv >> 16) - (int)(iv & 0xFFFFu);
            }
          }
          if (kept == 0) { cx0 = cx1 = cy0 = cy1 = 0; }
          out_radius = (int32_t)radius;        // radii / visibility are the published ones (3-sigma rectangle non-empty)
          out_tiles = (uint32_t)kept;
          out_key = __float_as_uint(t[2]);
        }
      }
    } else if (prefiltered) {
      // prefiltered point failed the near-plane test (hard error upstream).  Tile-local form: the flag rides in the top bit of
      // this workgroup's instance total (below), so nothing has to be cleared before this kernel starts
      if (block_sums) culled_flag = true;
      else meta[1] = 1u;
    }
  }

  // ---- phase 2: colour.  Only Gaussians that reach at least one tile need their SH row: the workgroup fetches just those
  // rows (a third of the bench scene is culled, most of a room-scale capture is), still as flat coalesced 16-B pieces ----
  const bool want_sh = on_screen && out_tiles != 0 && !colors_precomp && !defer_color;
  if (STAGE) {
    need_sh[threadIdx.x] = want_sh ? 1 : 0;
    __syncthreads();
    const size_t row0 = (size_t)blockIdx.x * 256;
    const int rows = (int)min((size_t)256, (size_t)P - row0);
    stage_rows_in(shs + row0 * S, rows * S, S, Sp, sh_lds, need_sh);
    __syncthreads();
  }
  if (on_screen) {
    // rgb stays 0 for deferred colour (k_shade fills it, and `clamped`, right before compositing) and for a Gaussian whose
    // alpha >= 1/255 ellipse reaches no tile (never composited, and the backward skips it)
    float rgb[3] = {0.f, 0.f, 0.f};
    uint8_t cl = 0;
    if (colors_precomp) {
      rgb[0] = colors_precomp[3 * (size_t)idx];
      rgb[1] = colors_precomp[3 * (size_t)idx + 1];
      rgb[2] = colors_precomp[3 * (size_t)idx + 2];
    } else if (want_sh) {
      float dx = p[0] - campos[0], dy = p[1] - campos[1], dz = p[2] - campos[2];
      const float inv = 1.0f / sqrtf(dx * dx + dy * dy + dz * dz);
      dx *= inv; dy *= inv; dz *= inv;
      float bs[16];
      sh_basis_eval(deg, dx, dy, dz, bs);
      const int K = (deg + 1) * (deg + 1);
      for (int k = 0; k < K; k++) {
#pragma unroll
        for (int ch = 0; ch < 3; ch++)
          rgb[ch] += bs[k] * (STAGE ? sh_coef_lds(dc, my_row, idx, k, ch) : sh_coef(dc, shs, sh_stride, idx, k, ch));
      }
#pragma unroll
      for (int ch = 0; ch < 3; ch++) {
        rgb[ch] += 0.5f;
        if (rgb[ch] < 0.f) { cl |= (1u << ch); rgb[ch] = 0.f; }
      }
    }
    rec[3 * (size_t)idx + 0] = make_float4(px, py, sA, sB);
    rec[3 * (size_t)idx + 1] = make_float4(sC, op, spmin, rgb[0]);
    rec[3 * (size_t)idx + 2] = make_float4(rgb[1], rgb[2], 1.0f / depth, depth);
    rect[idx] = make_ushort4((unsigned short)cx0, (unsigned short)cy0, (unsigned short)cx1, (unsigned short)cy1);
    // the emit pass walks the Gaussians in DEPTH order: one 32-B gather per Gaussian instead of three (record, rect, count)
    bin_rec[2 * (size_t)idx + 0] = make_float4(px, py, sA, sB);
    bin_rec[2 * (size_t)idx + 1] = make_float4(sC, spmin, __uint_as_float((uint32_t)cx0 | ((uint32_t)cy0 << 16)),
                                               __uint_as_float((uint32_t)cx1 | ((uint32_t)cy1 << 16)));
    clamped[idx] = cl;
  }
  if (in_range) {
    radii[idx] = out_radius;
    tiles_touched[idx] = out_tiles;
    depth_key[idx] = out_key;   // `order` is not written: the depth sort takes value = index on its first pass
  }
  // tile-local binning form: instances are emitted in index order, so the prefix sum of the tile counts is taken over the
  // workgroups of THIS kernel: each leaves its total, and k_emit_instances does the rest (start slot of its workgroup from the
  // totals in front of it, the scan inside the workgroup, num_rendered) - no scan launch at all
  if (block_sums) {                                                    // (kernel argument: uniform)
    uint32_t acc = out_tiles;
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) acc += __shfl_down(acc, d, 64);
    if ((threadIdx.x & 63) == 0) tile_sum[threadIdx.x >> 6] = acc;
    const int any_culled = __syncthreads_or(culled_flag ? 1 : 0);
    // (a workgroup's 256 Gaussians reach < 2^31 tiles: bit 31 is free for the "prefiltered point culled" flag)
    if (threadIdx.x == 0)
      block_sums[blockIdx.x] = ((tile_sum[0] + tile_sum[1]) + (tile_sum[2] + tile_sum[3])) | (any_culled ? GSR_BLOCK_CULLED : 0u);
  }
}

// num_rendered = sum of tiles_touched does not depend on the depth order: summed right after the projection (integer
// atomics: exact), so the host can read it back while the depth sort and the offset scan are still running (api.hip:
// forward_prepare_impl).  Same-address atomics cost ~15 ns each on this part, so: few workgroups, 16-B loads, one atomic per
// workgroup.
#define SUM_TILES_BLOCKS 128
__global__ __launch_bounds__(256) void k_sum_tiles(int P, const uint32_t* __restrict__ tiles_touched,
                                                   unsigned long long* __restrict__ total) {
  __shared__ uint32_t wave_sum[4];
  uint32_t acc = 0;
  const int P4 = P >> 2;                                              // tiles_touched is 256-B aligned (gsr_geom_layout)
  const uint4* v = reinterpret_cast<const uint4*>(tiles_touched);
  for (int i = blockIdx.x * 256 + threadIdx.x; i < P4; i += gridDim.x * 256) {
    const uint4 q = v[i];
    acc += (q.x + q.y) + (q.z + q.w);
  }
  if (blockIdx.x == 0 && threadIdx.x < (P & 3)) acc += tiles_touched[4 * P4 + threadIdx.x];
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) acc += __shfl_down(acc, d, 64);   // all 64 lanes are alive here
  if ((threadIdx.x & 63) == 0) wave_sum[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned long long s = (unsigned long long)wave_sum[0] + wave_sum[1] + wave_sum[2] + wave_sum[3];
    if (s != 0) atomicAdd(total, s);
  }
}

void gsr_launch_sum_tiles(int P, const char* geom, const GsrGeomLayout& L, uint32_t* meta, hipStream_t st) {
  const int blocks = min(SUM_TILES_BLOCKS, (P + 1023) / 1024);
  GSR_LAUNCH("sum_tiles", k_sum_tiles, dim3(blocks), dim3(256), 0, st, P, (const uint32_t*)(geom + L.tiles_touched),
             reinterpret_cast<unsigned long long*>(meta + 2));
}

// ---------------------------------------------------------------------------------------------------
// Deferred colour pass (SH -> RGB of K1 as its own kernel).  Used by the view-sharded data-parallel trainer: the SH
// coefficients are 81 % of the gradient bytes, and with the colour evaluated HERE - after projection, depth sort, emission and
// tile sort - their all-reduce + Adam update can still be in flight on another stream while those geometry stages of the
// next step run.  Same arithmetic as the fused K1 path (bitwise identical colours).
// ---------------------------------------------------------------------------------------------------
template <bool STAGE, int BT>
__global__ __launch_bounds__(BT) void k_shade(int P, int deg, int sh_stride, const float* __restrict__ means3D,
                                               const float* __restrict__ dc, const float* __restrict__ shs,
                                               const float* __restrict__ campos,
                                               const uint32_t* __restrict__ tiles_touched, float4* __restrict__ rec,
                                               uint8_t* __restrict__ clamped) {
  extern __shared__ __attribute__((aligned(16))) float sh_lds[];
  __shared__ int32_t need_sh[BT];
  const int idx = blockIdx.x * BT + threadIdx.x;
  const int S = 3 * sh_stride, Sp = S | 1;
  if (STAGE) {
    const size_t row0 = (size_t)blockIdx.x * BT;
    const int rows = (int)min((size_t)BT, (size_t)P - row0);
    need_sh[threadIdx.x] = (int)threadIdx.x < rows ? (int32_t)min(tiles_touched[row0 + threadIdx.x], 1u) : 0;
    __syncthreads();
    stage_rows_in<BT>(shs + row0 * S, rows * S, S, Sp, sh_lds, need_sh);
    __syncthreads();
  }
  if (idx >= P || tiles_touched[idx] == 0) return;
  const float* my_row = sh_lds + threadIdx.x * Sp;
  float dx = means3D[3 * (size_t)idx] - campos[0], dy = means3D[3 * (size_t)idx + 1] - campos[1],
        dz = means3D[3 * (size_t)idx + 2] - campos[2];
  const float inv = 1.0f / sqrtf(dx * dx + dy * dy + dz * dz);
  dx *= inv; dy *= inv; dz *= inv;
  float bs[16];
  sh_basis_eval(deg, dx, dy, dz, bs);
  const int K = (deg + 1) * (deg + 1);
  float rgb[3] = {0.f, 0.f, 0.f};
  for (int k = 0; k < K; k++) {
#pragma unroll
    for (int ch = 0; ch < 3; ch++)
      rgb[ch] += bs[k] * (STAGE ? sh_coef_lds(dc, my_row, idx, k, ch) : sh_coef(dc, shs, sh_stride, idx, k, ch));
  }
  uint8_t cl = 0;
#pragma unroll
  for (int ch = 0; ch < 3; ch++) {
    rgb[ch] += 0.5f;
    if (rgb[ch] < 0.f) { cl |= (1u << ch); rgb[ch] = 0.f; }
  }
  float* r = reinterpret_cast<float*>(rec + 3 * (size_t)idx);
  r[7] = rgb[0];      // r1.w
  r[8] = rgb[1];      // r2.x
  r[9] = rgb[2];      // r2.y
  clamped[idx] = cl;
}

// ---------------------------------------------------------------------------------------------------
// K8 + K9 backward, fused with the per-Gaussian gather of per-instance gradients.
// One thread per Gaussian index (all per-Gaussian arrays coalesced).
// igrad record (render backward): (S_x, S_y, S_xx, S_xy) (S_yy, d_opacity_eff, d_r, d_g) (d_b, d_invdepth, -, -) with
// S_* = opacity_eff x the record's sums over the instance's pixels of h dx, h dx^2 ... (h = dL/dpower / opacity_eff, d = mean -
// pixel; the multiplication happens here, once per Gaussian).  With
// power = -0.5(A dx^2 + C dy^2) - B dx dy:  dL/dA = -S_xx/2, dL/dB = -S_xy, dL/dC = -S_yy/2 and
// dL/dmean2D(ndc) = (W/2)(-A S_x - B S_y), (H/2)(-C S_y - B S_x): linear in the sums, so applied once, after the gather.
// ---------------------------------------------------------------------------------------------------
// Optimizer step folded into this kernel (gsr_backward_adam, include/gsr.h): ADAM = 1 torch.optim.Adam semantics on every row,
// ADAM = 2 SparseGaussianAdam semantics (rows with radii == 0 untouched, no bias correction).  Same arithmetic, element for
// element, as k_adam (adam.hip) applied to the gradients this kernel would have written - which then never travel through HBM:
// the 59 floats per Gaussian are neither stored here nor re-read there (2 x 236 MB per step at 1 M Gaussians, SH 3).
// the n (<= 4) consecutive elements of one Gaussian's row of a small parameter group
template <int ADAM, int N>
__device__ __forceinline__ void adam_row(const GsrAdamArgs& A, const int grp, const size_t idx, const float* g) {
  float* P = A.p[grp] + N * idx;
  float* M = A.m[grp] + N * idx;
  float* V = A.v[grp] + N * idx;
  float p[N], m[N], v[N];
#pragma unroll
  for (int j = 0; j < N; j++) { p[j] = P[j]; m[j] = M[j]; v[j] = V[j]; }
#pragma unroll
  for (int j = 0; j < N; j++) adam_elem<ADAM == 2 ? 2 : 1>(p[j], m[j], v[j], g[j], A, grp);
#pragma unroll
  for (int j = 0; j < N; j++) { P[j] = p[j]; M[j] = m[j]; V[j] = v[j]; }
}

#ifndef GSR_SHADE_BT
#define GSR_SHADE_BT 64
#endif
#ifndef GSR_BWD_ADAM_BT
#define GSR_BWD_ADAM_BT 64    // Gaussians per workgroup of the folded-optimizer backward with staged SH rows (256 / 128 / 64: 0.351 / 0.347 / 0.337 ms at C3)
#endif
#ifndef GSR_BWD_PLAIN_BT
#define GSR_BWD_PLAIN_BT 64   // ... of the plain backward (gradients stored; what the multi-GPU schedules run): 256 / 128 / 64 -> 0.158 / 0.151 / 0.148 ms
#endif
#ifndef GSR_ADAM_AHEAD
#define GSR_ADAM_AHEAD 6      // trips of f_rest moments in flight ahead of the one being updated (2 -> 6: -2 % of the kernel)
#endif
// BT = Gaussians (threads) per workgroup.  The folded-optimizer instantiations stage 64 floats per row in LDS: 256-row workgroups
// fit two per CU, one-wave workgroups of 64 rows eight or nine - the same waves, but four times as many independent phases
// (gather / arithmetic / streaming) in flight, and the barriers become free.
template <bool STAGE, int ADAM, int BT>
__global__ __launch_bounds__(BT) void k_preprocess_bwd(
    int P, int deg, int sh_stride, const float* __restrict__ means3D, const float* __restrict__ dc,
    const float* __restrict__ shs, const float* __restrict__ colors_precomp, const float* __restrict__ opacities,
    const float* __restrict__ scales, const float* __restrict__ rotations, const float* __restrict__ cov3D_precomp,
    float scale_modifier, const float* __restrict__ viewmatrix, const float* __restrict__ projmatrix,
    const float* __restrict__ campos, int W, int H, float tanfovx, float tanfovy, int antialiasing, int raw_act,
    const int32_t* __restrict__ radii, const uint8_t* __restrict__ clamped, const uint32_t* __restrict__ tiles_touched,
    const uint32_t* __restrict__ slot_start, const float4* __restrict__ igrad, const uint32_t* __restrict__ n_dev,
    uint32_t cap, float* __restrict__ dL_dmeans3D, float* __restrict__ dL_dmeans2D,
    float* __restrict__ dL_ddc, float* __restrict__ dL_dshs, float* __restrict__ dL_dcolors,
    float* __restrict__ dL_dopacities, float* __restrict__ dL_dscales, float* __restrict__ dL_drotations,
    float* __restrict__ dL_dcov3D, float* __restrict__ st_accum, float* __restrict__ st_denom,
    float* __restrict__ st_max_radii, const GsrAdamArgs A_in, uint32_t flags_min_r) {
  extern __shared__ __attribute__((aligned(16))) float sh_lds[];
  const GsrAdamArgs A = ADAM ? gsr_adam_resolve(A_in) : A_in;
  __shared__ int32_t need_sh[BT];
  const int S = 3 * sh_stride, Sp = S | 1;
  const size_t row0 = (size_t)blockIdx.x * BT;
  const int rows = (int)min((size_t)BT, (size_t)P - row0);
  // (round 4) where this thread's gradient records start, how many there are, and the validity flags of the first sixteen - asked
  // for HERE so that they are on their way while the coefficient rows are staged below (the record reads depend on them)
  uint32_t pre_s0 = 0u, pre_n = 0u, pre_fw[4] = {0u, 0u, 0u, 0u};
  {
    const int pidx = (int)min((size_t)(blockIdx.x * BT + threadIdx.x), (size_t)P - 1);
    // (both words asked for at once - the count is not waited for before the start slot is requested; a Gaussian without instances
    // has a stale start slot, which is then not used)
    const uint32_t tt = tiles_touched[pidx], ss = slot_start[pidx];
    if (!gsr_overflowed(n_dev, cap) && tt > 0) {
      const uint32_t n_eff = gsr_eff_n(n_dev, cap);
      pre_s0 = ss;
      pre_n = pre_s0 < n_eff ? min(tt, n_eff - pre_s0) : 0u;
      if (pre_n && gsr_flags_on(n_dev, cap, flags_min_r)) __builtin_memcpy(pre_fw, gsr_igrad_flags(igrad, cap) + pre_s0, 16);
    }
  }
  if (STAGE) {
    // rows to fetch: those with instances (their coefficients enter the gradient); with the optimizer folded in, every
    // row that will be UPDATED (all of them / the visible ones), since the update reads the parameter from the staged copy
    // (ADAM = 3: dense Adam on the rows WITH instances only; the others got their zero-gradient update from k_adam_culled_rows)
    if (ADAM == 1) need_sh[threadIdx.x] = (int)threadIdx.x < rows ? 1 : 0;
    else if (ADAM == 2) need_sh[threadIdx.x] = (int)threadIdx.x < rows ? (radii[row0 + threadIdx.x] > 0 ? 1 : 0) : 0;
    else
    need_sh[threadIdx.x] = (int)threadIdx.x < rows ? (int32_t)min(tiles_touched[row0 + threadIdx.x], 1u) : 0;
    __syncthreads();
    stage_rows_in<BT>(shs + row0 * S, rows * S, S, Sp, sh_lds, need_sh);   // only the rows of Gaussians with instances are read below
    __syncthreads();
  }
  float* my_row = sh_lds + threadIdx.x * Sp;
  const bool active = blockIdx.x * BT + threadIdx.x < P;
  const int idx = active ? blockIdx.x * BT + threadIdx.x : P - 1;   // idle tail threads mirror the last Gaussian (no stores)
  const int K = (deg + 1) * (deg + 1);
  // A Gaussian that reached no tile (culled, or its alpha >= 1/255 ellipse misses every tile centre row) has no gradient
  // records: all its gradients are exact zeros, written below without touching its inputs.
  // A frame whose instance list was truncated (non-blocking forward beyond its capacity) is treated as if NO Gaussian had been
  // visible: zero gradients everywhere, and neither the folded optimizer step nor the folded statistics happen (gsr_overflowed).
  const bool overflow = gsr_overflowed(n_dev, cap);       // grid-uniform
  const bool visible = !overflow && tiles_touched[idx] > 0;

  float g_mean[3] = {0.f, 0.f, 0.f};
  float g_m2d[2] = {0.f, 0.f};
  float g_opac = 0.f;
  float g_cov6[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  float g_col[3] = {0.f, 0.f, 0.f};
  float g_scale[3] = {0.f, 0.f, 0.f};
  float g_rot[4] = {0.f, 0.f, 0.f, 0.f};
  float bs[16];
  bool have_sh = false;

  if (visible) {
    // ---- deterministic gather-sum over this Gaussian's instances (slot order) ----
    float acc[10];
#pragma unroll
    for (int i = 0; i < 10; i++) acc[i] = 0.f;
    // (a non-blocking forward whose instance count exceeded the binning capacity dropped the slots >= n_eff: no records)
    const uint32_t s0 = pre_s0, n = pre_n;      // (idx == the row the prologue looked at: an active thread's own)
    // this Gaussian's records are contiguous (slot order): stream them, 4 records in flight per thread.
    // Summation order = slot order (deterministic).
    const float4* rows = igrad + (size_t)GSR_IGRAD_F4 * s0;
    // (round 4) one validity byte per slot: an instance behind its tile's walk has no record (it would be all zeros) - the bytes
    // of a Gaussian's slots are contiguous like its records
    const unsigned char* fl = gsr_igrad_flags(igrad, cap) + s0;
    const bool flags_on = gsr_flags_on(n_dev, cap, flags_min_r);   // (grid-uniform; off: every slot holds a record, zeros included)
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    // The flags of sixteen slots come in ONE load (from wherever the Gaussian's slots start: the read may run up to fifteen bytes
    // past them - the next Gaussian's flags, or the array's padding, gsr_backward_scratch_bytes), then four records per trip, masked
    // by the count: one memory round trip per trip as before the flags (a flag load in front of every trip cost 100 k Gaussians
    // with 13 instances each 12 us).
    for (uint32_t it0 = 0; it0 < n; it0 += 16) {
      uint32_t fw[4];
      if (it0 == 0) { fw[0] = pre_fw[0]; fw[1] = pre_fw[1]; fw[2] = pre_fw[2]; fw[3] = pre_fw[3]; }
      else if (flags_on) __builtin_memcpy(fw, fl + it0, 16);
      else fw[0] = fw[1] = fw[2] = fw[3] = 0u;
#pragma unroll
      for (int t = 0; t < 4; t++) {
        const uint32_t it = it0 + 4 * t;
        if (it >= n) break;
        float4 q[12];
        if (!flags_on) {
          // (small frames: the round-3 form - every slot was written, zeros included; nothing to wait for in front of the reads)
#pragma unroll
          for (int u = 0; u < 4; u++)
            if (it + u < n) {
              q[3 * u] = rows[3 * (size_t)(it + u)];
              q[3 * u + 1] = rows[3 * (size_t)(it + u) + 1];
              q[3 * u + 2] = rows[3 * (size_t)(it + u) + 2];
            } else {
              q[3 * u] = q[3 * u + 1] = q[3 * u + 2] = z4;
            }
        } else {
#pragma unroll
          for (int u = 0; u < 12; u++) q[u] = z4;
          // (plain loads on purpose: the records were just written by the render backward and are largely still in the
          // infinity cache - streaming hints on either side cost 35 % here.  Explicit branches: written as `flag ? load : zero`
          // the compiler selects between POINTERS and reads the records dword by dword through flat loads.)
#pragma unroll
          for (int u = 0; u < 4; u++)
            if (((fw[t] >> (8 * u)) & 0xFFu) != 0u && it + u < n) {
              q[3 * u] = rows[3 * (size_t)(it + u)];
              q[3 * u + 1] = rows[3 * (size_t)(it + u) + 1];
              q[3 * u + 2] = rows[3 * (size_t)(it + u) + 2];
            }
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
          acc[0] += q[3 * u].x; acc[1] += q[3 * u].y; acc[2] += q[3 * u].z; acc[3] += q[3 * u].w;
          acc[4] += q[3 * u + 1].x; acc[5] += q[3 * u + 1].y; acc[6] += q[3 * u + 1].z; acc[7] += q[3 * u + 1].w;
          acc[8] += q[3 * u + 2].x; acc[9] += q[3 * u + 2].y;
        }
      }
    }
    const float g_op_eff = acc[5];
    float g_rgb[3] = {acc[6], acc[7], acc[8]};
    const float g_invd = acc[9];

    PreView v;
    load_view(viewmatrix, projmatrix, campos, v);
    const float p[3] = {means3D[3 * (size_t)idx], means3D[3 * (size_t)idx + 1], means3D[3 * (size_t)idx + 2]};
    float t[3];
#pragma unroll
    for (int i = 0; i < 3; i++) t[i] = v.V[i] * p[0] + v.V[4 + i] * p[1] + v.V[8 + i] * p[2] + v.V[12 + i];

    // ---- colour: SH backward (A.7 iv) or pass-through ----
    if (colors_precomp) {
      g_col[0] = g_rgb[0]; g_col[1] = g_rgb[1]; g_col[2] = g_rgb[2];
    } else {
      const uint8_t cl = clamped[idx];
#pragma unroll
      for (int ch = 0; ch < 3; ch++)
        if (cl & (1u << ch)) g_rgb[ch] = 0.f;
      float vx = p[0] - v.cam[0], vy = p[1] - v.cam[1], vz = p[2] - v.cam[2];
      const float inv = 1.0f / sqrtf(vx * vx + vy * vy + vz * vz);
      const float x = vx * inv, y = vy * inv, z = vz * inv;
      sh_basis_eval(deg, x, y, z, bs);
      have_sh = true;
      // direction derivative: g_dir = sum_k dbasis_k/ddir * (sh_k . g_rgb)
      float gdx = 0.f, gdy = 0.f, gdz = 0.f;
      if (deg > 0) {
        float q[16];
        for (int k = 1; k < K; k++) {
          q[k] = STAGE ? (sh_coef_lds(dc, my_row, idx, k, 0) * g_rgb[0] + sh_coef_lds(dc, my_row, idx, k, 1) * g_rgb[1] +
                          sh_coef_lds(dc, my_row, idx, k, 2) * g_rgb[2])
                       : (sh_coef(dc, shs, sh_stride, idx, k, 0) * g_rgb[0] + sh_coef(dc, shs, sh_stride, idx, k, 1) * g_rgb[1] +
                          sh_coef(dc, shs, sh_stride, idx, k, 2) * g_rgb[2]);
        }
        gdx += -SH_C1 * q[3];
        gdy += -SH_C1 * q[1];
        gdz += SH_C1 * q[2];
        if (deg > 1) {
          const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
          gdx += SH_C2[0] * y * q[4] + SH_C2[2] * (-2.f * x) * q[6] + SH_C2[3] * z * q[7] + SH_C2[4] * 2.f * x * q[8];
          gdy += SH_C2[0] * x * q[4] + SH_C2[1] * z * q[5] + SH_C2[2] * (-2.f * y) * q[6] + SH_C2[4] * (-2.f * y) * q[8];
          gdz += SH_C2[1] * y * q[5] + SH_C2[2] * 4.f * z * q[6] + SH_C2[3] * x * q[7];
          if (deg > 2) {
            gdx += SH_C3[0] * 6.f * xy * q[9] + SH_C3[1] * yz * q[10] + SH_C3[2] * (-2.f * xy) * q[11] +
                   SH_C3[3] * (-6.f * xz) * q[12] + SH_C3[4] * (4.f * zz - 3.f * xx - yy) * q[13] +
                   SH_C3[5] * 2.f * xz * q[14] + SH_C3[6] * (3.f * xx - 3.f * yy) * q[15];
            gdy += SH_C3[0] * (3.f * xx - 3.f * yy) * q[9] + SH_C3[1] * xz * q[10] +
                   SH_C3[2] * (4.f * zz - xx - 3.f * yy) * q[11] + SH_C3[3] * (-6.f * yz) * q[12] +
                   SH_C3[4] * (-2.f * xy) * q[13] + SH_C3[5] * (-2.f * yz) * q[14] + SH_C3[6] * (-6.f * xy) * q[15];
            gdz += SH_C3[1] * xy * q[10] + SH_C3[2] * 8.f * yz * q[11] +
                   SH_C3[3] * (6.f * zz - 3.f * xx - 3.f * yy) * q[12] + SH_C3[4] * 8.f * xz * q[13] +
                   SH_C3[5] * (xx - yy) * q[14];
          }
        }
        // through dir = v/|v|
        const float dot = x * gdx + y * gdy + z * gdz;
        g_mean[0] += (gdx - x * dot) * inv;
        g_mean[1] += (gdy - y * dot) * inv;
        g_mean[2] += (gdz - z * dot) * inv;
      }
      g_col[0] = g_rgb[0]; g_col[1] = g_rgb[1]; g_col[2] = g_rgb[2];  // masked dL/drgb, used for dL/dsh below
    }

    // ---- covariance chain (A.7 i, ii) ----
    float cov6[6];
    float R[9];
    float s3[3] = {0.f, 0.f, 0.f}, q4[4] = {0.f, 0.f, 0.f, 0.f}, qden = 1.0f;
    if (cov3D_precomp) {
#pragma unroll
      for (int i = 0; i < 6; i++) cov6[i] = cov3D_precomp[6 * (size_t)idx + i];
    } else {
      load_scale_rot(scales, rotations, (size_t)idx, raw_act, s3, q4, qden);
      cov3d_from_sr(s3, q4, scale_modifier, cov6, R);
    }
    const float fx = W / (2.0f * tanfovx), fy = H / (2.0f * tanfovy);
    Ewa e;
    ewa_project(t, v, cov6, fx, fy, tanfovx, tanfovy, e);
    const float a = e.a0 + 0.3f, c = e.c0 + 0.3f, b = e.b;
    const float det = a * c - b * b;
    const float opac = load_opacity(opacities, (size_t)idx, raw_act);
    {
      // the records hold the moments of h = dL/dpower / opacity_eff (render.hip): the factor, as the forward formed it
      const float op_eff = antialiasing ? opac * sqrtf(fmaxf(0.000025f, (e.a0 * e.c0 - b * b) / det)) : opac;
#pragma unroll
      for (int i = 0; i < 5; i++) acc[i] *= op_eff;
    }
    const float gA = -0.5f * acc[2], gB = -acc[3], gC = -0.5f * acc[4];
    {
      // conic exactly as the forward computed it (a visible Gaussian has det != 0)
      const float det_inv = 1.0f / det;
      const float cA = c * det_inv, cB = -b * det_inv, cC = a * det_inv;
      g_m2d[0] = (0.5f * W) * (-cA * acc[0] - cB * acc[1]);
      g_m2d[1] = (0.5f * H) * (-cC * acc[1] - cB * acc[0]);
    }
    const float det2inv = 1.0f / (det * det + 0.0000001f);
    float g_a = det2inv * (-c * c * gA + b * c * gB - b * b * gC);
    float g_c = det2inv * (-b * b * gA + a * b * gB - a * a * gC);
    float g_b = det2inv * (2.f * b * c * gA - (det + 2.f * b * b) * gB + 2.f * a * b * gC);
    if (antialiasing) {
      const float det0 = e.a0 * e.c0 - b * b;
      const float f = det0 / det;
      const float h = sqrtf(fmaxf(0.000025f, f));
      g_opac = g_op_eff * h;
      if (f > 0.000025f) {
        const float g_f = (g_op_eff * opac) / (2.f * h);
        const float di = 1.0f / (det * det);
        g_a += g_f * (e.c0 * det - det0 * c) * di;
        g_c += g_f * (e.a0 * det - det0 * a) * di;
        g_b += g_f * (2.f * b * (det0 - det)) * di;
      }
    } else {
      g_opac = g_op_eff;
    }
    // cov2D -> Sigma (6-vector; off-diagonals appear twice)
    const float* m0 = e.m0;
    const float* m1 = e.m1;
    g_cov6[0] = g_a * m0[0] * m0[0] + g_b * m0[0] * m1[0] + g_c * m1[0] * m1[0];
    g_cov6[3] = g_a * m0[1] * m0[1] + g_b * m0[1] * m1[1] + g_c * m1[1] * m1[1];
    g_cov6[5] = g_a * m0[2] * m0[2] + g_b * m0[2] * m1[2] + g_c * m1[2] * m1[2];
    g_cov6[1] = 2.f * g_a * m0[0] * m0[1] + g_b * (m0[0] * m1[1] + m0[1] * m1[0]) + 2.f * g_c * m1[0] * m1[1];
    g_cov6[2] = 2.f * g_a * m0[0] * m0[2] + g_b * (m0[0] * m1[2] + m0[2] * m1[0]) + 2.f * g_c * m1[0] * m1[2];
    g_cov6[4] = 2.f * g_a * m0[1] * m0[2] + g_b * (m0[1] * m1[2] + m0[2] * m1[1]) + 2.f * g_c * m1[1] * m1[2];
    // cov2D -> M rows
    float Sm0[3], Sm1[3];
    Sm0[0] = cov6[0] * m0[0] + cov6[1] * m0[1] + cov6[2] * m0[2];
    Sm0[1] = cov6[1] * m0[0] + cov6[3] * m0[1] + cov6[4] * m0[2];
    Sm0[2] = cov6[2] * m0[0] + cov6[4] * m0[1] + cov6[5] * m0[2];
    Sm1[0] = cov6[0] * m1[0] + cov6[1] * m1[1] + cov6[2] * m1[2];
    Sm1[1] = cov6[1] * m1[0] + cov6[3] * m1[1] + cov6[4] * m1[2];
    Sm1[2] = cov6[2] * m1[0] + cov6[4] * m1[1] + cov6[5] * m1[2];
    float gm0[3], gm1[3];
#pragma unroll
    for (int j = 0; j < 3; j++) {
      gm0[j] = 2.f * g_a * Sm0[j] + g_b * Sm1[j];
      gm1[j] = 2.f * g_c * Sm1[j] + g_b * Sm0[j];
    }
    // M -> J (W row k = (V[k], V[4+k], V[8+k]))
    float gJ00 = 0.f, gJ02 = 0.f, gJ11 = 0.f, gJ12 = 0.f;
#pragma unroll
    for (int j = 0; j < 3; j++) {
      gJ00 += gm0[j] * v.V[4 * j + 0];
      gJ02 += gm0[j] * v.V[4 * j + 2];
      gJ11 += gm1[j] * v.V[4 * j + 1];
      gJ12 += gm1[j] * v.V[4 * j + 2];
    }
    const float tz1 = 1.0f / e.tz, tz2 = tz1 * tz1, tz3 = tz2 * tz1;
    float g_t[3];
    g_t[0] = e.in_x ? (-fx * tz2 * gJ02) : 0.f;
    g_t[1] = e.in_y ? (-fy * tz2 * gJ12) : 0.f;
    g_t[2] = -fx * tz2 * gJ00 - fy * tz2 * gJ11 + 2.f * fx * e.tx * tz3 * gJ02 + 2.f * fy * e.ty * tz3 * gJ12;
    // inverse depth output: invd = 1/t.z
    g_t[2] += -g_invd * tz2;
    // t = V p
#pragma unroll
    for (int j = 0; j < 3; j++) g_mean[j] += v.V[4 * j + 0] * g_t[0] + v.V[4 * j + 1] * g_t[1] + v.V[4 * j + 2] * g_t[2];

    // ---- mean2D (NDC) -> mean3D (A.7 iii) ----
    {
      float hom[4];
#pragma unroll
      for (int i = 0; i < 4; i++) hom[i] = v.PV[i] * p[0] + v.PV[4 + i] * p[1] + v.PV[8 + i] * p[2] + v.PV[12 + i];
      const float pw = 1.0f / (hom[3] + 0.0000001f);
      const float mul1 = hom[0] * pw * pw, mul2 = hom[1] * pw * pw;
#pragma unroll
      for (int j = 0; j < 3; j++) {
        g_mean[j] += (v.PV[4 * j + 0] * pw - v.PV[4 * j + 3] * mul1) * g_m2d[0] +
                     (v.PV[4 * j + 1] * pw - v.PV[4 * j + 3] * mul2) * g_m2d[1];
      }
    }

    // ---- Sigma -> scale, rotation (A.7 v) ----
    if (!cov3D_precomp) {
      const float sp[3] = {scale_modifier * s3[0], scale_modifier * s3[1], scale_modifier * s3[2]};
      // G = full symmetric gradient matrix; dL/dL = 2 G L, L = R diag(sp)
      const float G[9] = {g_cov6[0],       0.5f * g_cov6[1], 0.5f * g_cov6[2], 0.5f * g_cov6[1], g_cov6[3],
                          0.5f * g_cov6[4], 0.5f * g_cov6[2], 0.5f * g_cov6[4], g_cov6[5]};
      float gR[9];
#pragma unroll
      for (int j = 0; j < 3; j++) {
        float gs = 0.f;
#pragma unroll
        for (int i = 0; i < 3; i++) {
          // dL/dL_ij = 2 * sum_k G_ik L_kj,  L_kj = R_kj sp_j
          const float dLij = 2.f * sp[j] * (G[3 * i + 0] * R[0 + j] + G[3 * i + 1] * R[3 + j] + G[3 * i + 2] * R[6 + j]);
          gs += dLij * R[3 * i + j];
          gR[3 * i + j] = dLij * sp[j];
        }
        g_scale[j] = scale_modifier * gs;
      }
      const float r = q4[0], x = q4[1], y = q4[2], z = q4[3];
      g_rot[0] = 2.f * (-z * gR[1] + y * gR[2] + z * gR[3] - x * gR[5] - y * gR[6] + x * gR[7]);
      g_rot[1] = 2.f * (y * gR[1] + z * gR[2] + y * gR[3] - 2.f * x * gR[4] - r * gR[5] + z * gR[6] + r * gR[7] -
                        2.f * x * gR[8]);
      g_rot[2] = 2.f * (-2.f * y * gR[0] + x * gR[1] + r * gR[2] + x * gR[3] + z * gR[5] - r * gR[6] + z * gR[7] -
                        2.f * y * gR[8]);
      g_rot[3] = 2.f * (-2.f * z * gR[0] - r * gR[1] + x * gR[2] + r * gR[3] - 2.f * z * gR[4] + y * gR[5] + x * gR[6] +
                        y * gR[7]);
      if (raw_act) {   // through exp and F.normalize (q4 is the unit quaternion, qden = max(|q_raw|, eps))
#pragma unroll
        for (int j = 0; j < 3; j++) g_scale[j] *= s3[j];
        const float dot = qden > 1e-12f ? q4[0] * g_rot[0] + q4[1] * g_rot[1] + q4[2] * g_rot[2] + q4[3] * g_rot[3] : 0.f;
#pragma unroll
        for (int j = 0; j < 4; j++) g_rot[j] = (g_rot[j] - q4[j] * dot) / qden;
      }
    }
    if (raw_act) g_opac *= opac * (1.0f - opac);     // through sigmoid
  }

  // ---- write every output (zeros for culled Gaussians: no memset pass needed) ----
  if (active) {
    dL_dmeans2D[3 * (size_t)idx + 0] = g_m2d[0];
    dL_dmeans2D[3 * (size_t)idx + 1] = g_m2d[1];
    dL_dmeans2D[3 * (size_t)idx + 2] = 0.f;
    if (st_accum && !overflow) {     // this view's densification statistics, folded in (gsr_grads.xyz_gradient_accum / denom / max_radii2D)
      const int rad = radii[idx];
      if (rad > 0) gsr_densify_stats_update(g_m2d[0], g_m2d[1], rad, st_accum + idx, st_denom + idx, st_max_radii + idx);
    }
  }
  if (ADAM) {
    if (overflow) return;      // the step is a no-op: parameters and both moments keep their bits
    // ---- optimizer step instead of gradient stores (raw-parameter call form: the gradients above ARE the leaves') ----
    const bool upd = active && (ADAM == 1 || (ADAM == 2 ? radii[idx] > 0 : visible));
    if (upd) {
      // the five small groups (xyz 3, f_dc 3, opacity 1, scaling 3, rotation 4 elements per row): ALL their parameter and
      // moment loads first, then the arithmetic, then the stores - group after group exposed a memory round trip each
      const float bk0 = have_sh ? bs[0] : 0.f;
      const float gg[14] = {g_mean[0], g_mean[1], g_mean[2], bk0 * g_col[0], bk0 * g_col[1], bk0 * g_col[2], g_opac,
                            g_scale[0], g_scale[1], g_scale[2], g_rot[0], g_rot[1], g_rot[2], g_rot[3]};
      constexpr int GRP[14] = {0, 0, 0, 1, 1, 1, 3, 4, 4, 4, 5, 5, 5, 5};
      constexpr int WID[14] = {3, 3, 3, 3, 3, 3, 1, 3, 3, 3, 4, 4, 4, 4};
      constexpr int COL[14] = {0, 1, 2, 0, 1, 2, 0, 0, 1, 2, 0, 1, 2, 3};
      float pv[14], mv[14], vv[14];
#pragma unroll
      for (int e = 0; e < 14; e++) {
        const size_t o = (size_t)WID[e] * idx + COL[e];
        pv[e] = A.p[GRP[e]][o]; mv[e] = A.m[GRP[e]][o]; vv[e] = A.v[GRP[e]][o];
      }
#pragma unroll
      for (int e = 0; e < 14; e++) adam_elem<ADAM == 2 ? 2 : 1>(pv[e], mv[e], vv[e], gg[e], A, GRP[e]);
#pragma unroll
      for (int e = 0; e < 14; e++) {
        const size_t o = (size_t)WID[e] * idx + COL[e];
        A.p[GRP[e]][o] = pv[e]; A.m[GRP[e]][o] = mv[e]; A.v[GRP[e]][o] = vv[e];
      }
    }
    if (STAGE) {
      // f_rest: dL/dsh[k][c] = basis_k * dL/drgb_c is rank one, so a row's gradient is 19 numbers (basis values, masked
      // dL/drgb) kept in LDS next to the staged PARAMETER rows; the update then runs over the block's span of f_rest as flat
      // 16-B pieces (coalesced m / v / p traffic), forming each element's gradient on the fly.
      float* fac = sh_lds + BT * Sp + threadIdx.x * 19;
#pragma unroll
      for (int k = 0; k < 16; k++) fac[k] = (have_sh && k < K) ? bs[k] : 0.f;
      fac[16] = g_col[0]; fac[17] = g_col[1]; fac[18] = g_col[2];
      __syncthreads();
      const float* facs = sh_lds + BT * Sp;
      const int nflt = rows * S, n4 = nflt >> 2;
      const int dr = (4 * BT) / S, dcol = (4 * BT) - dr * S;
      int r = (threadIdx.x * 4) / S, c = threadIdx.x * 4 - r * S;
      float* Pg = A.p[2] + row0 * S;
      float* Mg = A.m[2] + row0 * S;
      float* Vg = A.v[2] + row0 * S;
      if (ADAM == 1) {
        // every piece is updated: the moments of the next GSR_ADAM_AHEAD trips are in flight while this one is computed and stored (the
        // loop is pure streaming - 32 B in, 48 B out per piece - and with two workgroups per CU it lives on loads in flight)
        constexpr int AHEAD = GSR_ADAM_AHEAD;
        gsr_f4 mq[AHEAD + 1], vq[AHEAD + 1];
#pragma unroll
        for (int u = 0; u < AHEAD; u++) {
          const int iu = threadIdx.x + BT * u;
          if (iu < n4) { mq[u] = gsr_ld_stream(Mg + 4 * (size_t)iu); vq[u] = gsr_ld_stream(Vg + 4 * (size_t)iu); }
        }
        for (int i = threadIdx.x; i < n4; i += BT) {
          if (i + BT * AHEAD < n4) {
            mq[AHEAD] = gsr_ld_stream(Mg + 4 * (size_t)(i + BT * AHEAD));
            vq[AHEAD] = gsr_ld_stream(Vg + 4 * (size_t)(i + BT * AHEAD));
          }
          float pp[4], mm[4] = {mq[0].x, mq[0].y, mq[0].z, mq[0].w}, vv[4] = {vq[0].x, vq[0].y, vq[0].z, vq[0].w};
          int rr = r, cc = c;
#pragma unroll
          for (int k = 0; k < 4; k++) {
            pp[k] = sh_lds[rr * Sp + cc];
            const int kk = cc / 3, ch = cc - 3 * kk;
            adam_elem<1>(pp[k], mm[k], vv[k], facs[rr * 19 + kk + 1] * facs[rr * 19 + 16 + ch], A, 2);
            if (++cc == S) { cc = 0; rr++; }
          }
          gsr_st_stream(Pg + 4 * (size_t)i, gsr_f4{pp[0], pp[1], pp[2], pp[3]});
          gsr_st_stream(Mg + 4 * (size_t)i, gsr_f4{mm[0], mm[1], mm[2], mm[3]});
          gsr_st_stream(Vg + 4 * (size_t)i, gsr_f4{vv[0], vv[1], vv[2], vv[3]});
#pragma unroll
          for (int u = 0; u < AHEAD; u++) { mq[u] = mq[u + 1]; vq[u] = vq[u + 1]; }
          r += dr; c += dcol;
          if (c >= S) { c -= S; r++; }
        }
      } else {
        // pieces of visible rows only.  Two phases, fully unrolled: first every needed piece's moments are requested (the
        // one-load-then-wait loop this replaces exposed a memory round trip per trip: the sparse update was SLOWER than the
        // dense one, 0.44 vs 0.34 ms at C3, although it moves two thirds of the bytes), then the updates run over the
        // arrived data.  Two workgroups per CU: up to 16 trips x 8 registers fit.
        gsr_f4 mq[GSR_STAGE_MAX_TRIPS], vq[GSR_STAGE_MAX_TRIPS];
        bool want[GSR_STAGE_MAX_TRIPS];
        {
          int r1 = r, c1 = c;
#pragma unroll
          for (int u = 0; u < GSR_STAGE_MAX_TRIPS; u++) {
            const int i = threadIdx.x + BT * u;
            want[u] = i < n4 && (need_sh[r1] > 0 || need_sh[min(r1 + (c1 + 3 >= S ? 1 : 0), rows - 1)] > 0);
            if (want[u]) { mq[u] = gsr_ld_stream(Mg + 4 * (size_t)i); vq[u] = gsr_ld_stream(Vg + 4 * (size_t)i); }
            r1 += dr; c1 += dcol;
            if (c1 >= S) { c1 -= S; r1++; }
          }
        }
#pragma unroll
        for (int u = 0; u < GSR_STAGE_MAX_TRIPS; u++) {
          const int i = threadIdx.x + BT * u;
          if (want[u]) {                          // (else: a piece of invisible rows only, neither read nor written)
            const gsr_f4 m4 = mq[u], v4 = vq[u];
            float pp[4], mm[4] = {m4.x, m4.y, m4.z, m4.w}, vv[4] = {v4.x, v4.y, v4.z, v4.w};
            int rr = r, cc = c;
            bool any = false, all = true;
#pragma unroll
            for (int k = 0; k < 4; k++) {
              // ADAM = 3 owns every 16-B piece that touches a row with instances, WHOLE: the elements of a neighbouring row
              // without instances in it get their (zero-gradient) dense update here too - their `fac` entries are zeros and
              // the staging fetched the straddling piece - so no piece is ever split between this kernel and
              // k_adam_culled_rows
              const bool vis = ADAM == 3 ? true : need_sh[rr] > 0;
              pp[k] = sh_lds[rr * Sp + cc];                   // (undefined for rows that were not fetched: not stored then)
              if (vis) {
                const int kk = cc / 3, ch = cc - 3 * kk;
                const float g = facs[rr * 19 + kk + 1] * facs[rr * 19 + 16 + ch];
                adam_elem<ADAM == 2 ? 2 : 1>(pp[k], mm[k], vv[k], g, A, 2);
                any = true;
              } else {
                all = false;
              }
              if (++cc == S) { cc = 0; rr++; }
            }
            if (all) {
              gsr_st_stream(Pg + 4 * (size_t)i, gsr_f4{pp[0], pp[1], pp[2], pp[3]});
              gsr_st_stream(Mg + 4 * (size_t)i, gsr_f4{mm[0], mm[1], mm[2], mm[3]});
              gsr_st_stream(Vg + 4 * (size_t)i, gsr_f4{vv[0], vv[1], vv[2], vv[3]});
            } else if (any) {   // (sparse only) a piece straddling a visible and an invisible row: element stores
              int r2 = r, c2 = c;
#pragma unroll
              for (int k = 0; k < 4; k++) {
                if (need_sh[r2] > 0) { Pg[4 * (size_t)i + k] = pp[k]; Mg[4 * (size_t)i + k] = mm[k]; Vg[4 * (size_t)i + k] = vv[k]; }
                if (++c2 == S) { c2 = 0; r2++; }
              }
            }
          }
          r += dr; c += dcol;
          if (c >= S) { c -= S; r++; }
        }
      }
      for (int e = n4 * 4 + threadIdx.x; e < nflt; e += BT) {   // tail of a span whose length is not a multiple of 4
        const int r2 = e / S, c2 = e - r2 * S;
        if (need_sh[r2] > 0) {
          const int kk = c2 / 3, ch = c2 - 3 * kk;
          float pv = sh_lds[r2 * Sp + c2], mv = Mg[e], vvv = Vg[e];
          adam_elem<ADAM == 2 ? 2 : 1>(pv, mv, vvv, facs[r2 * 19 + kk + 1] * facs[r2 * 19 + 16 + ch], A, 2);
          Pg[e] = pv; Mg[e] = mv; Vg[e] = vvv;
        }
      }
    }
    return;
  }
  if (active) {
#pragma unroll
    for (int j = 0; j < 3; j++) dL_dmeans3D[3 * (size_t)idx + j] = g_mean[j];
    dL_dopacities[idx] = g_opac;
    if (dL_dcolors) {
#pragma unroll
      for (int j = 0; j < 3; j++) dL_dcolors[3 * (size_t)idx + j] = g_col[j];
    }
    if (dL_dcov3D) {
#pragma unroll
      for (int j = 0; j < 6; j++) dL_dcov3D[6 * (size_t)idx + j] = g_cov6[j];
    }
    if (dL_dscales) {
#pragma unroll
      for (int j = 0; j < 3; j++) dL_dscales[3 * (size_t)idx + j] = g_scale[j];
    }
    if (dL_drotations) {
#pragma unroll
      for (int j = 0; j < 4; j++) dL_drotations[4 * (size_t)idx + j] = g_rot[j];
    }
  }
  if (dL_dshs || dL_ddc) {
    // stored coefficients beyond the active degree get zero gradient
    const int stored = sh_stride + (dL_ddc ? 1 : 0);
    // dL_dshs == NULL with dL_ddc set (the view-sharded exchange "sh_rank1": only dL/df_dc travels, the other coefficients'
    // gradients are rebuilt from it, csrc/exchange.hip): the rest rows are neither formed nor stored
    if (STAGE) {
      // the thread has consumed its own SH row: overwrite it with the gradient row, then one flat coalesced copy-out
      for (int k = 0; k < stored; k++) {
        const float bk = (have_sh && k < K) ? bs[k] : 0.f;
        if (dL_ddc && k == 0) {
          if (active) {
            float* d0 = dL_ddc + 3 * (size_t)idx;
            d0[0] = bk * g_col[0]; d0[1] = bk * g_col[1]; d0[2] = bk * g_col[2];
          }
        } else if (dL_dshs) {
          float* dst = my_row + (dL_ddc ? (k - 1) : k) * 3;
          dst[0] = bk * g_col[0]; dst[1] = bk * g_col[1]; dst[2] = bk * g_col[2];
        }
      }
      if (dL_dshs) {          // (kernel argument: uniform)
        __syncthreads();
        stage_rows_out<BT>(dL_dshs + row0 * S, rows * S, S, Sp, sh_lds);
      }
    } else if (active) {
      for (int k = 0; k < stored; k++) {
        const float bk = (have_sh && k < K) ? bs[k] : 0.f;
        float* dst = dL_ddc ? ((k == 0) ? dL_ddc + 3 * (size_t)idx : (dL_dshs ? dL_dshs + ((size_t)idx * sh_stride + (k - 1)) * 3 : nullptr))
                            : dL_dshs + ((size_t)idx * sh_stride + k) * 3;
        if (!dst) continue;
        dst[0] = bk * g_col[0];
        dst[1] = bk * g_col[1];
        dst[2] = bk * g_col[2];
      }
    }
  }
}

// Dense Adam on the rows that reached no tile in this forward (their gradient is exactly zero): m, v decay and the parameter
// follows its momentum, as torch.optim.Adam does for a zero gradient.  Needs only `tiles_touched` of the forward, so it can run on
// another stream WHILE the compositing kernels (VALU-bound, little HBM traffic) run; k_preprocess_bwd<., 3> then updates the
// rows with instances.  Same adam_elem as everywhere: the two kernels together equal k_preprocess_bwd<., 1> bit for bit.
__global__ __launch_bounds__(256) void k_adam_culled_rows(int P, int sh_stride, const uint32_t* __restrict__ tiles_touched,
                                                          const GsrAdamArgs A_in, const uint32_t* __restrict__ n_dev,
                                                          uint32_t cap) {
  const GsrAdamArgs A = gsr_adam_resolve(A_in);
  __shared__ int32_t culled[256];
  if (gsr_overflowed(n_dev, cap)) return;   // the other half of the update (k_preprocess_bwd<., 3>) skips the frame too
  const size_t row0 = (size_t)blockIdx.x * 256;
  const int rows = (int)min((size_t)256, (size_t)P - row0);
  const bool mine = (int)threadIdx.x < rows && tiles_touched[row0 + threadIdx.x] == 0;
  culled[threadIdx.x] = mine ? 1 : 0;
  __syncthreads();
  if (mine) {
    const size_t idx = row0 + threadIdx.x;
    const float z[4] = {0.f, 0.f, 0.f, 0.f};
    adam_row<1, 3>(A, 0, idx, z);
    adam_row<1, 3>(A, 1, idx, z);
    adam_row<1, 1>(A, 3, idx, z);
    adam_row<1, 3>(A, 4, idx, z);
    adam_row<1, 4>(A, 5, idx, z);
  }
  if (sh_stride == 0) return;
  const int S = 3 * sh_stride;
  const int nflt = rows * S, n4 = nflt >> 2;
  const int dr = 1024 / S, dcol = 1024 - dr * S;
  int r = (threadIdx.x * 4) / S, c = threadIdx.x * 4 - r * S;
  float* Pg = A.p[2] + row0 * S;
  float* Mg = A.m[2] + row0 * S;
  float* Vg = A.v[2] + row0 * S;
  for (int i = threadIdx.x; i < n4; i += 256) {
    // a 16-B piece is ours only if EVERY row it touches is culled; a piece shared with a row that has instances belongs,
    // whole, to k_preprocess_bwd<., 3> (which gives our elements in it the same zero-gradient update)
    const int r_end = min(r + (c + 3 >= S ? 1 : 0), rows - 1);
    if (culled[r] && culled[r_end]) {
      const gsr_f4 p4 = gsr_ld_stream(Pg + 4 * (size_t)i), m4 = gsr_ld_stream(Mg + 4 * (size_t)i),
                   v4 = gsr_ld_stream(Vg + 4 * (size_t)i);
      float pp[4] = {p4.x, p4.y, p4.z, p4.w}, mm[4] = {m4.x, m4.y, m4.z, m4.w}, vv[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
      for (int k = 0; k < 4; k++) adam_elem<1>(pp[k], mm[k], vv[k], 0.f, A, 2);
      gsr_st_stream(Pg + 4 * (size_t)i, gsr_f4{pp[0], pp[1], pp[2], pp[3]});
      gsr_st_stream(Mg + 4 * (size_t)i, gsr_f4{mm[0], mm[1], mm[2], mm[3]});
      gsr_st_stream(Vg + 4 * (size_t)i, gsr_f4{vv[0], vv[1], vv[2], vv[3]});
    }
    r += dr; c += dcol;
    if (c >= S) { c -= S; r++; }
  }
  for (int e = n4 * 4 + threadIdx.x; e < nflt; e += 256) {
    const int r2 = e / S;
    if (culled[r2]) {
      float pv = Pg[e], mv = Mg[e], vvv = Vg[e];
      adam_elem<1>(pv, mv, vvv, 0.f, A, 2);
      Pg[e] = pv; Mg[e] = mv; Vg[e] = vvv;
    }
  }
}

void gsr_launch_adam_culled_rows(int P, int sh_stride, const char* geom, const GsrGeomLayout& L, const GsrAdamArgs& A,
                                 uint32_t cap, hipStream_t st) {
  GSR_LAUNCH("adam_culled_rows", k_adam_culled_rows, dim3((P + 255) / 256), dim3(256), 0, st, P, sh_stride,
             (const uint32_t*)(geom + L.tiles_touched), A, (const uint32_t*)(geom + L.meta) + 2, cap);
}

__global__ __launch_bounds__(256) void k_mark_visible(int P, const float* __restrict__ means3D,
                                                      const float* __restrict__ viewmatrix, uint8_t* __restrict__ present) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= P) return;
  const float z = viewmatrix[2] * means3D[3 * (size_t)idx] + viewmatrix[6] * means3D[3 * (size_t)idx + 1] +
                  viewmatrix[10] * means3D[3 * (size_t)idx + 2] + viewmatrix[14];
  present[idx] = z > 0.2f ? 1 : 0;
}

// ---------------------------------------------------------------------------------------------------
// host launchers
// ---------------------------------------------------------------------------------------------------
// LDS staging applies when the rasterizer reads every stored coefficient of `shs` (active degree == stored degree),
// the span is 16-B aligned and 256 padded rows fit comfortably (<= 64 KB so that two workgroups share a CU).
static bool can_stage_sh(const gsr_settings* s, const gsr_gaussians* g, size_t* lds_bytes) {
  if (!g->shs || g->colors_precomp || g->sh_coeffs <= 0) return false;
  const int K = (s->sh_degree + 1) * (s->sh_degree + 1);
  if (K - (g->dc ? 1 : 0) != g->sh_coeffs) return false;
  if (((uintptr_t)g->shs & 15) != 0) return false;
  const size_t bytes = (size_t)256 * ((3 * g->sh_coeffs) | 1) * sizeof(float);
  if (bytes > 64 * 1024) return false;
  *lds_bytes = bytes;
  return true;
}

// beside_other_work: the pass runs on a side stream next to the (latency-bound) tile sort.  There the 256-row form is kept on
// purpose: 64-row workgroups finish the pass in 0.065 instead of 0.096 ms at C3 but take so much more of the machine while they
// run that the radix passes beside them slow down by more (step 1.408 vs 1.390 ms; throttling the pass further - two or one
// workgroup per CU through an LDS pad - gains nothing / costs 0.04 ms).  On the caller's own stream - blocking
// forward, late colour pass of the data-parallel overlap, forward-only renders - the pass is on the critical path and gets 64.
void gsr_launch_shade(const gsr_settings* s, const gsr_gaussians* g, char* geom, const GsrGeomLayout& L, bool beside_other_work,
                      hipStream_t st) {
  const int P = g->P;
  if (P == 0 || g->colors_precomp) return;
  size_t lds = 0;
  const bool stage = can_stage_sh(s, g, &lds);
#define GSR_SHADE_ARGS                                                                                           \
  P, s->sh_degree, g->sh_coeffs, g->means3D, g->dc, g->shs, s->campos, (const uint32_t*)(geom + L.tiles_touched), \
      (float4*)(geom + L.rec), (uint8_t*)(geom + L.clamped)
  // (rows per workgroup: 64, as in the backward - 12 KB of staged rows instead of 46 KB, 13 workgroups per CU instead of 3)
  if (stage && beside_other_work)
    GSR_LAUNCH("shade", (k_shade<true, 256>), dim3((P + 255) / 256), dim3(256), lds, st, GSR_SHADE_ARGS);
  else if (stage)
    GSR_LAUNCH("shade", (k_shade<true, GSR_SHADE_BT>), dim3((P + GSR_SHADE_BT - 1) / GSR_SHADE_BT), dim3(GSR_SHADE_BT),
               (lds / 256) * GSR_SHADE_BT, st, GSR_SHADE_ARGS);
  else
    GSR_LAUNCH("shade", (k_shade<false, 256>), dim3((P + 255) / 256), dim3(256), 0, st, GSR_SHADE_ARGS);
#undef GSR_SHADE_ARGS
}

// block_sums: the tile-local binning form's per-workgroup instance totals go to the head of the (otherwise unused) `offsets`
// array: [0, nb) totals, [nb, 2 nb) start slots, nb = ceil(P / 256)
void gsr_launch_preprocess_fwd(const gsr_settings* s, const gsr_gaussians* g, int32_t* radii, char* geom,
                               const GsrGeomLayout& L, bool defer_color, bool block_sums, uint32_t* zero_words,
                               int n_zero_words, const uint32_t* tile_cutoff, uint32_t* culled_any, uint32_t frame_tag,
                               hipStream_t st) {
  const int P = g->P;
  size_t lds = 0;
  const bool stage = !defer_color && can_stage_sh(s, g, &lds);
#define GSR_PRE_FWD_ARGS                                                                                              \
  P, s->sh_degree, g->sh_coeffs, g->means3D, g->dc, g->shs, g->colors_precomp, g->opacities, g->scales, g->rotations, \
      g->cov3D_precomp, s->scale_modifier, s->viewmatrix, s->projmatrix, s->campos, s->image_width, s->image_height,  \
      s->tanfovx, s->tanfovy, s->prefiltered, s->antialiasing, (int)defer_color, (int)g->raw_activations, radii,    \
      (float4*)(geom + L.rec),                                                                                       \
      (uint32_t*)(geom + L.depth_key), (uint32_t*)(geom + L.order), (uint32_t*)(geom + L.tiles_touched),              \
      (ushort4*)(geom + L.rect), (float4*)(geom + L.bin_rec), (uint8_t*)(geom + L.clamped), (uint32_t*)(geom + L.meta), \
      block_sums ? (uint32_t*)(geom + L.offsets) : (uint32_t*)nullptr, zero_words, n_zero_words, tile_cutoff,       \
      culled_any, frame_tag
  if (stage)
    GSR_LAUNCH("preprocess_fwd", k_preprocess_fwd<true>, dim3((P + 255) / 256), dim3(256), lds, st, GSR_PRE_FWD_ARGS);
  else
    GSR_LAUNCH("preprocess_fwd", k_preprocess_fwd<false>, dim3((P + 255) / 256), dim3(256), 0, st, GSR_PRE_FWD_ARGS);
#undef GSR_PRE_FWD_ARGS
}

// adam: nullptr (plain backward) or the folded optimizer step; adam_mode 1 dense / 2 sparse.  Returns 0, or -1 when the
// folded form does not apply to these inputs (the caller reports the error).
int gsr_launch_preprocess_bwd(const gsr_settings* s, const gsr_gaussians* g, const int32_t* radii,
                              const char* geom, const GsrGeomLayout& L, const float4* igrad, uint32_t cap,
                              const gsr_grads* gr, const GsrAdamArgs* adam, int adam_mode, hipStream_t st) {
  const int P = g->P;
  size_t lds = 0;
  GsrAdamArgs A;
  memset(&A, 0, sizeof(A));
  bool stage;
  if (adam) {
    // raw-parameter call form with dc / rest passed separately: every gradient of this kernel is a leaf's gradient
    if (!g->raw_activations || !g->dc || g->colors_precomp || g->cov3D_precomp || !g->scales || !g->rotations) return -1;
    stage = g->shs != nullptr;
    if (stage && !can_stage_sh(s, g, &lds)) return -1;     // (f_rest rows are updated from their staged copy)
    if (!stage && g->sh_coeffs != 0) return -1;
    if (stage) lds = (lds / 256) * GSR_BWD_ADAM_BT + (size_t)GSR_BWD_ADAM_BT * 19 * sizeof(float);   // rows of this kernel's workgroup
    A = *adam;
  } else {
    stage = can_stage_sh(s, g, &lds) && (gr->dL_dshs ? (((uintptr_t)gr->dL_dshs & 15) == 0) : gr->dL_ddc != nullptr);
    if (stage) lds = (lds / 256) * GSR_BWD_PLAIN_BT;
  }
#define GSR_PRE_BWD_ARGS                                                                                              \
  P, s->sh_degree, g->sh_coeffs, g->means3D, g->dc, g->shs, g->colors_precomp, g->opacities, g->scales, g->rotations, \
      g->cov3D_precomp, s->scale_modifier, s->viewmatrix, s->projmatrix, s->campos, s->image_width, s->image_height,  \
      s->tanfovx, s->tanfovy, s->antialiasing, (int)g->raw_activations, radii, (const uint8_t*)(geom + L.clamped),     \
      (const uint32_t*)(geom + L.tiles_touched), (const uint32_t*)(geom + L.slot_start), igrad,                       \
      (const uint32_t*)(geom + L.meta) + 2, cap, gr->dL_dmeans3D,                                                      \
      gr->dL_dmeans2D, gr->dL_ddc, gr->dL_dshs, gr->dL_dcolors, gr->dL_dopacities, gr->dL_dscales, gr->dL_drotations, \
      gr->dL_dcov3D, gr->xyz_gradient_accum, gr->denom, gr->max_radii2D, A, g_gsr_flags_min_r
#define GSR_PRE_BWD(ST, AD, BT_)                                                                                       \
  do {                                                                                                                 \
    if (lds > 48 * 1024)                                                                                               \
      (void)hipFuncSetAttribute((const void*)k_preprocess_bwd<ST, AD, BT_>,                                            \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                                 \
    GSR_LAUNCH(AD ? "preprocess_bwd_adam" : "preprocess_bwd", (k_preprocess_bwd<ST, AD, BT_>),                         \
               dim3((P + BT_ - 1) / BT_), dim3(BT_), ST ? lds : 0, st, GSR_PRE_BWD_ARGS);                              \
  } while (0)
  if (!adam) {
    if (stage) GSR_PRE_BWD(true, 0, GSR_BWD_PLAIN_BT); else GSR_PRE_BWD(false, 0, 256);
  } else if (adam_mode == 2) {
    if (stage) GSR_PRE_BWD(true, 2, GSR_BWD_ADAM_BT); else GSR_PRE_BWD(false, 2, 256);
  } else if (adam_mode == 3) {
    if (stage) GSR_PRE_BWD(true, 3, GSR_BWD_ADAM_BT); else GSR_PRE_BWD(false, 3, 256);
  } else {
    if (stage) GSR_PRE_BWD(true, 1, GSR_BWD_ADAM_BT); else GSR_PRE_BWD(false, 1, 256);
  }
#undef GSR_PRE_BWD
#undef GSR_PRE_BWD_ARGS
  return 0;
}

void gsr_launch_mark_visible(int P, const float* means3D, const float* viewmatrix, uint8_t* present,
                             hipStream_t st) {
  GSR_LAUNCH("mark_visible", k_mark_visible, dim3((P + 255) / 256), dim3(256), 0, st, P, means3D, viewmatrix,
             present);
}
