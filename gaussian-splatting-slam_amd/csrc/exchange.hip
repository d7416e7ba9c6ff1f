// exchange.hip - kernels of the view-sharded data-parallel gradient exchange (no counterpart in the reference,
// which is single-GPU).
//
// k_sh_rank1_expand: the SH gradient of ONE view is rank one per Gaussian - dL/dsh[k][c] = basis_k(dir) * dL/drgb_c (SURVEY.md
// A.7 iv, csrc/preprocess.hip) - and band 0 has the constant basis C0, so dL/df_dc = C0 * dL/drgb (masked where the colour was
// clamped) already carries everything the other 15 coefficients' gradients are made of.  With one view per rank per step the
// ranks therefore exchange their dL/df_dc [P, 3] (12 B per Gaussian and rank, all-gather) and their camera centres instead of
// all-reducing dL/df_rest [P, 15, 3] (180 B per Gaussian); every rank then rebuilds
//     mean_r dL/df_rest[k][c] = (1 / N) sum_r (basis_k(dir_r) / C0) * dL/df_dc_r[c],      dir_r = normalize(xyz - campos_r)
// here, summing in RANK ORDER: the same bits on every rank, and the all-reduce schedule's values up to fp32 rounding of the
// individual products (each term is rounded once more: (b / C0) (C0 g) against b g).
#include <math.h>
#include <string.h>

#include "gsr_common.h"

// No implicit fma contraction in this file (as in preprocess.hip): k_sh_rank1_expand and k_sh_rank1_adam must form the same
// gradient bits whatever the compiler would contract in their different surroundings; the one fma of the rebuild is spelled out.
#pragma clang fp contract(off)

#define GSR_X_C0 0.28209479177387814f
#define GSR_X_C1 0.4886025119029199f

__device__ __forceinline__ void x_sh_basis(int deg, float x, float y, float z, float* b /*16*/) {
  // basis / signs of reference utils/sh_utils.py:74-100 (as csrc/preprocess.hip sh_basis_eval)
  b[0] = GSR_X_C0;
  if (deg > 0) {
    b[1] = -GSR_X_C1 * y;
    b[2] = GSR_X_C1 * z;
    b[3] = -GSR_X_C1 * x;
    if (deg > 1) {
      const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
      b[4] = 1.0925484305920792f * xy;
      b[5] = -1.0925484305920792f * yz;
      b[6] = 0.31539156525252005f * (2.f * zz - xx - yy);
      b[7] = -1.0925484305920792f * xz;
      b[8] = 0.5462742152960396f * (xx - yy);
      if (deg > 2) {
        b[9] = -0.5900435899266435f * y * (3.f * xx - yy);
        b[10] = 2.890611442640554f * xy * z;
        b[11] = -0.4570457994644658f * y * (4.f * zz - xx - yy);
        b[12] = 0.3731763325901154f * z * (2.f * zz - 3.f * xx - 3.f * yy);
        b[13] = -0.4570457994644658f * x * (4.f * zz - xx - yy);
        b[14] = 1.445305721320277f * z * (xx - yy);
        b[15] = -0.5900435899266435f * x * (xx - 3.f * yy);
      }
    }
  }
}

#define XBT 64        // Gaussians per workgroup: 64 rows x 45 floats staged in LDS for the flat copy-out / the flat update

// Mean gradients of one Gaussian: acc_dc[3] and acc[3 KREST] (scaled), ranks summed in order.
template <int KREST>
__device__ __forceinline__ void x_rank1_rows(int P, int n_ranks, int deg, int idx, const float* __restrict__ means3D,
                                             const float* __restrict__ gathered, float scale, float* acc_dc, float* acc) {
  constexpr int S = 3 * KREST;
  const size_t rank_stride = 3 * ((size_t)P + 1);
  acc_dc[0] = acc_dc[1] = acc_dc[2] = 0.f;
#pragma unroll
  for (int i = 0; i < S; i++) acc[i] = 0.f;
  if (idx >= P) return;
  const float px = means3D[3 * (size_t)idx], py = means3D[3 * (size_t)idx + 1], pz = means3D[3 * (size_t)idx + 2];
  const int K = (deg + 1) * (deg + 1);
  for (int r = 0; r < n_ranks; r++) {
    const float* gr = gathered + (size_t)r * rank_stride;
    const float g0 = gr[3 * (size_t)idx], g1 = gr[3 * (size_t)idx + 1], g2 = gr[3 * (size_t)idx + 2];
    acc_dc[0] += g0; acc_dc[1] += g1; acc_dc[2] += g2;
    if (KREST > 0 && (g0 != 0.f || g1 != 0.f || g2 != 0.f)) {      // (a Gaussian without instances in rank r's view: exact zeros)
      const float cx = gr[3 * (size_t)P], cy = gr[3 * (size_t)P + 1], cz = gr[3 * (size_t)P + 2];   // rank r's camera centre
      float dx = px - cx, dy = py - cy, dz = pz - cz;
      const float inv = 1.0f / sqrtf(dx * dx + dy * dy + dz * dz);
      dx *= inv; dy *= inv; dz *= inv;
      float b[16];
      x_sh_basis(deg, dx, dy, dz, b);
#pragma unroll
      for (int k = 1; k <= KREST; k++) {
        if (k < K) {
          const float wgt = b[k] * (1.0f / GSR_X_C0);
          acc[3 * (k - 1) + 0] = __builtin_fmaf(wgt, g0, acc[3 * (k - 1) + 0]);
          acc[3 * (k - 1) + 1] = __builtin_fmaf(wgt, g1, acc[3 * (k - 1) + 1]);
          acc[3 * (k - 1) + 2] = __builtin_fmaf(wgt, g2, acc[3 * (k - 1) + 2]);
        }
      }
    }
  }
  acc_dc[0] *= scale; acc_dc[1] *= scale; acc_dc[2] *= scale;
#pragma unroll
  for (int i = 0; i < S; i++) acc[i] *= scale;
}

template <int KREST>  // stored "rest" coefficients per Gaussian (15 at SH degree 3), compile-time for the register arrays
__global__ __launch_bounds__(XBT) void k_sh_rank1_expand(int P, int n_ranks, int deg, const float* __restrict__ means3D,
                                                         const float* __restrict__ gathered /* [n_ranks][P + 1][3] */,
                                                         float scale, float* __restrict__ out_dc,
                                                         float* __restrict__ out_rest) {
  constexpr int S = 3 * KREST, SP = S | 1;
  __shared__ float rows[XBT * SP];
  const int idx = blockIdx.x * XBT + threadIdx.x;
  float acc_dc[3];
  float acc[KREST > 0 ? S : 1];
  x_rank1_rows<KREST>(P, n_ranks, deg, idx, means3D, gathered, scale, acc_dc, acc);
  if (idx < P) {
    out_dc[3 * (size_t)idx + 0] = acc_dc[0];
    out_dc[3 * (size_t)idx + 1] = acc_dc[1];
    out_dc[3 * (size_t)idx + 2] = acc_dc[2];
  }
  if (KREST > 0) {
    // the thread's row goes to LDS (odd stride: conflict-free), the workgroup's span leaves as flat 16-B pieces
#pragma unroll
    for (int i = 0; i < S; i++) rows[threadIdx.x * SP + i] = acc[i];
    __syncthreads();
    const size_t row0 = (size_t)blockIdx.x * XBT;
    const int nrows = (int)min((size_t)XBT, (size_t)P - row0);
    const int nflt = nrows * S;
    float* dst = out_rest + row0 * S;
    if ((((uintptr_t)dst) & 15) == 0) {
      const int n4 = nflt >> 2;
      for (int i = threadIdx.x; i < n4; i += XBT) {
        float v[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
          const int e = 4 * i + k, r = e / S, c = e - r * S;
          v[k] = rows[r * SP + c];
        }
        reinterpret_cast<float4*>(dst)[i] = make_float4(v[0], v[1], v[2], v[3]);
      }
      for (int e = n4 * 4 + threadIdx.x; e < nflt; e += XBT) dst[e] = rows[(e / S) * SP + (e % S)];
    } else {
      for (int e = threadIdx.x; e < nflt; e += XBT) dst[e] = rows[(e / S) * SP + (e % S)];
    }
  }
}

// The same rebuild with the dense Adam update of the two SH groups folded in: the 48 rebuilt floats per Gaussian are never written
// to memory (nor re-read by an optimizer kernel): parameters and both moments stream through once.  Same gradient bits as
// k_sh_rank1_expand, same adam_elem as k_adam (gsr_common.h): bit-identical to expand + gsr_adam_step.  A.*[1] = f_dc, A.*[2] = f_rest.
template <int KREST>
__global__ __launch_bounds__(XBT) void k_sh_rank1_adam(int P, int n_ranks, int deg, const float* __restrict__ means3D,
                                                       const float* __restrict__ gathered, float scale, const GsrAdamArgs A) {
  constexpr int S = 3 * KREST, SP = S | 1;
  __shared__ float rows[XBT * SP];
  const int idx = blockIdx.x * XBT + threadIdx.x;
  float acc_dc[3];
  float acc[KREST > 0 ? S : 1];
  x_rank1_rows<KREST>(P, n_ranks, deg, idx, means3D, gathered, scale, acc_dc, acc);
  if (idx < P) {
    float* Pd = A.p[1] + 3 * (size_t)idx; float* Md = A.m[1] + 3 * (size_t)idx; float* Vd = A.v[1] + 3 * (size_t)idx;
    float p[3] = {Pd[0], Pd[1], Pd[2]}, m[3] = {Md[0], Md[1], Md[2]}, v[3] = {Vd[0], Vd[1], Vd[2]};
#pragma unroll
    for (int j = 0; j < 3; j++) adam_elem<1>(p[j], m[j], v[j], acc_dc[j], A, 1);
#pragma unroll
    for (int j = 0; j < 3; j++) { Pd[j] = p[j]; Md[j] = m[j]; Vd[j] = v[j]; }
  }
  if (KREST > 0) {
#pragma unroll
    for (int i = 0; i < S; i++) rows[threadIdx.x * SP + i] = acc[i];
    __syncthreads();
    const size_t row0 = (size_t)blockIdx.x * XBT;
    const int nrows = (int)min((size_t)XBT, (size_t)P - row0);
    const int nflt = nrows * S;
    float* Pg = A.p[2] + row0 * S; float* Mg = A.m[2] + row0 * S; float* Vg = A.v[2] + row0 * S;
    const bool vec = ((((uintptr_t)Pg) | ((uintptr_t)Mg) | ((uintptr_t)Vg)) & 15) == 0;
    const int n4 = vec ? (nflt >> 2) : 0;
    for (int i = threadIdx.x; i < n4; i += XBT) {
      const gsr_f4 p4 = gsr_ld_stream(Pg + 4 * (size_t)i), m4 = gsr_ld_stream(Mg + 4 * (size_t)i), v4 = gsr_ld_stream(Vg + 4 * (size_t)i);
      float pp[4] = {p4.x, p4.y, p4.z, p4.w}, mm[4] = {m4.x, m4.y, m4.z, m4.w}, vv[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const int e = 4 * i + k, r = e / S, c = e - r * S;
        adam_elem<1>(pp[k], mm[k], vv[k], rows[r * SP + c], A, 2);
      }
      gsr_st_stream(Pg + 4 * (size_t)i, gsr_f4{pp[0], pp[1], pp[2], pp[3]});
      gsr_st_stream(Mg + 4 * (size_t)i, gsr_f4{mm[0], mm[1], mm[2], mm[3]});
      gsr_st_stream(Vg + 4 * (size_t)i, gsr_f4{vv[0], vv[1], vv[2], vv[3]});
    }
    for (int e = n4 * 4 + threadIdx.x; e < nflt; e += XBT) {
      float pv = Pg[e], mv = Mg[e], vvv = Vg[e];
      adam_elem<1>(pv, mv, vvv, rows[(e / S) * SP + (e % S)], A, 2);
      Pg[e] = pv; Mg[e] = mv; Vg[e] = vvv;
    }
  }
}

static int x_check(int32_t P, int32_t n_ranks, int32_t sh_degree, int32_t sh_coeffs_rest, const char* what) {
  if (P < 0 || n_ranks < 1 || sh_degree < 0 || sh_degree > 3 || sh_coeffs_rest < 0) {
    gsr_set_error("%s: bad arguments", what);
    return GSR_ERR_INVALID_ARGUMENT;
  }
  if ((sh_degree + 1) * (sh_degree + 1) - 1 > sh_coeffs_rest) {
    gsr_set_error("%s: sh_degree %d needs %d rest coefficients, %d stored", what, sh_degree,
                  (sh_degree + 1) * (sh_degree + 1) - 1, sh_coeffs_rest);
    return GSR_ERR_INVALID_ARGUMENT;
  }
  if (sh_coeffs_rest != 0 && sh_coeffs_rest != 3 && sh_coeffs_rest != 8 && sh_coeffs_rest != 15) {
    gsr_set_error("%s: %d stored rest coefficients (0, 3, 8 or 15 = SH degree 0..3)", what, sh_coeffs_rest);
    return GSR_ERR_INVALID_ARGUMENT;
  }
  return 0;
}

extern "C" int gsr_sh_rank1_adam(int32_t P, int32_t n_ranks, int32_t sh_degree, int32_t sh_coeffs_rest, const float* means3D,
                                 const float* gathered, float scale, float* f_dc, float* f_rest, const gsr_fused_adam* opt,
                                 void* stream) {
  int rc = x_check(P, n_ranks, sh_degree, sh_coeffs_rest, "sh_rank1_adam");
  if (rc) return rc;
  if (!opt || opt->sparse != 0 || (P > 0 && (!means3D || !gathered || !f_dc || !opt->exp_avg[1] || !opt->exp_avg_sq[1] ||
                                             (sh_coeffs_rest > 0 && (!f_rest || !opt->exp_avg[2] || !opt->exp_avg_sq[2]))))) {
    gsr_set_error("sh_rank1_adam: bad arguments (dense Adam; moments of groups 1 = f_dc and 2 = f_rest)");
    return GSR_ERR_INVALID_ARGUMENT;
  }
  if (P == 0) return 0;
  GsrAdamArgs A;
  memset(&A, 0, sizeof(A));
  float* params[6] = {nullptr, f_dc, f_rest, nullptr, nullptr, nullptr};
  for (int i = 1; i <= 2; i++) {
    A.p[i] = params[i]; A.m[i] = opt->exp_avg[i]; A.v[i] = opt->exp_avg_sq[i];
    A.lr[i] = opt->lr[i];
    const double bc1 = 1.0 - pow(opt->beta1, (double)opt->step[i]);
    const double bc2 = 1.0 - pow(opt->beta2, (double)opt->step[i]);
    A.step_size[i] = (float)((double)opt->lr[i] / bc1);
    A.inv_bc2_sqrt[i] = (float)(1.0 / sqrt(bc2));
  }
  A.beta1 = (float)opt->beta1; A.beta2 = (float)opt->beta2;
  A.omb1 = (float)(1.0 - opt->beta1); A.omb2 = (float)(1.0 - opt->beta2);
  A.eps = (float)opt->eps;
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid((P + XBT - 1) / XBT), block(XBT);
#define XA_LAUNCH(K) \
  GSR_LAUNCH("sh_rank1_adam", k_sh_rank1_adam<K>, grid, block, 0, st, P, n_ranks, sh_degree, means3D, gathered, scale, A)
  switch (sh_coeffs_rest) {
    case 0: XA_LAUNCH(0); break;
    case 3: XA_LAUNCH(3); break;
    case 8: XA_LAUNCH(8); break;
    default: XA_LAUNCH(15); break;
  }
#undef XA_LAUNCH
  return gsr_launch_status("sh_rank1_adam");
}

extern "C" int gsr_sh_rank1_expand(int32_t P, int32_t n_ranks, int32_t sh_degree, int32_t sh_coeffs_rest, const float* means3D,
                                   const float* gathered, float scale, float* dL_ddc_mean, float* dL_dsh_rest_mean,
                                   void* stream) {
  int rc = x_check(P, n_ranks, sh_degree, sh_coeffs_rest, "sh_rank1_expand");
  if (rc) return rc;
  if (P > 0 && (!means3D || !gathered || !dL_ddc_mean || (sh_coeffs_rest > 0 && !dL_dsh_rest_mean))) {
    gsr_set_error("sh_rank1_expand: bad arguments");
    return GSR_ERR_INVALID_ARGUMENT;
  }
  if (P == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid((P + XBT - 1) / XBT), block(XBT);
#define X_LAUNCH(K) \
  GSR_LAUNCH("sh_rank1_expand", k_sh_rank1_expand<K>, grid, block, 0, st, P, n_ranks, sh_degree, means3D, gathered, scale, \
             dL_ddc_mean, dL_dsh_rest_mean)
  switch (sh_coeffs_rest) {
    case 0: X_LAUNCH(0); break;
    case 3: X_LAUNCH(3); break;
    case 8: X_LAUNCH(8); break;
    default: X_LAUNCH(15); break;
  }
#undef X_LAUNCH
  return gsr_launch_status("sh_rank1_expand");
}
