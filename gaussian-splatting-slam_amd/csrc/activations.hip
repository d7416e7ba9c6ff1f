// activations.hip - the three parameter activations of the Gaussian model in one launch each way.
//
// Reference scene/gaussian_model.py:38-46 (setup_functions) and :101-121 (get_scaling / get_rotation / get_opacity):
//   scaling = exp(_scaling), rotation = normalize(_rotation) (x / max(|x|, 1e-12)), opacity = sigmoid(_opacity).
// As separate PyTorch ops these are 7 launches forward and ~18 backward (F.normalize alone differentiates into 14
// elementwise / reduce kernels), each a few microseconds of launch-bound work per training step; here one thread per
// Gaussian does all three (32 B in, 32 B out forward; 96 B in, 32 B out backward): HBM-bound streaming kernels.
#include "gsr_common.h"

__global__ __launch_bounds__(256) void k_activations_fwd(int P, const float* __restrict__ raw_s,
                                                         const float4* __restrict__ raw_q,
                                                         const float* __restrict__ raw_o, float* __restrict__ s,
                                                         float4* __restrict__ q, float* __restrict__ o) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= P) return;
#pragma unroll
  for (int k = 0; k < 3; k++) s[3 * (size_t)i + k] = expf(raw_s[3 * (size_t)i + k]);
  const float4 r = raw_q[i];
  // (true divisions, like F.normalize: the rounding of the quaternion matches the PyTorch op bit for bit)
  const float den = fmaxf(sqrtf(r.x * r.x + r.y * r.y + r.z * r.z + r.w * r.w), 1e-12f);
  q[i] = make_float4(r.x / den, r.y / den, r.z / den, r.w / den);
  o[i] = 1.0f / (1.0f + expf(-raw_o[i]));
}

// Gradients: d exp = g * out;  d sigmoid = g * out (1 - out);  d normalize = (g - n (n . g)) / |x| with n = x / |x|
// (for |x| <= eps the forward is x / eps, so the gradient is g / eps).  A missing upstream gradient (nullptr) is zero.
__global__ __launch_bounds__(256) void k_activations_bwd(int P, const float4* __restrict__ raw_q,
                                                         const float* __restrict__ s, const float* __restrict__ o,
                                                         const float* __restrict__ g_s, const float4* __restrict__ g_q,
                                                         const float* __restrict__ g_o, float* __restrict__ d_s,
                                                         float4* __restrict__ d_q, float* __restrict__ d_o) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= P) return;
#pragma unroll
  for (int k = 0; k < 3; k++) d_s[3 * (size_t)i + k] = g_s ? g_s[3 * (size_t)i + k] * s[3 * (size_t)i + k] : 0.f;
  if (g_q) {
    const float4 r = raw_q[i], g = g_q[i];
    const float len = sqrtf(r.x * r.x + r.y * r.y + r.z * r.z + r.w * r.w);
    if (len > 1e-12f) {
      const float inv = 1.0f / len;
      const float nx = r.x * inv, ny = r.y * inv, nz = r.z * inv, nw = r.w * inv;
      const float dot = nx * g.x + ny * g.y + nz * g.z + nw * g.w;
      d_q[i] = make_float4((g.x - nx * dot) * inv, (g.y - ny * dot) * inv, (g.z - nz * dot) * inv, (g.w - nw * dot) * inv);
    } else {
      d_q[i] = make_float4(g.x * 1e12f, g.y * 1e12f, g.z * 1e12f, g.w * 1e12f);
    }
  } else {
    d_q[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  const float ov = o[i];
  d_o[i] = g_o ? g_o[i] * ov * (1.0f - ov) : 0.f;
}

extern "C" int gsr_gaussian_activations_forward(int32_t P, const float* raw_scaling, const float* raw_rotation,
                                                const float* raw_opacity, float* scaling, float* rotation,
                                                float* opacity, void* stream) {
  if (P < 0 || (P > 0 && (!raw_scaling || !raw_rotation || !raw_opacity || !scaling || !rotation || !opacity))) {
    gsr_set_error("activations forward: null buffer");
    return GSR_ERR_INVALID_ARGUMENT;
  }
  if (P == 0) return 0;
  GSR_LAUNCH("activations_fwd", k_activations_fwd, dim3((P + 255) / 256), dim3(256), 0, (hipStream_t)stream, P,
             raw_scaling, (const float4*)raw_rotation, raw_opacity, scaling, (float4*)rotation, opacity);
  return gsr_launch_status("activations forward");
}

extern "C" int gsr_gaussian_activations_backward(int32_t P, const float* raw_rotation, const float* scaling,
                                                 const float* opacity, const float* dL_dscaling,
                                                 const float* dL_drotation, const float* dL_dopacity,
                                                 float* dL_draw_scaling, float* dL_draw_rotation,
                                                 float* dL_draw_opacity, void* stream) {
  if (P < 0 || (P > 0 && (!raw_rotation || !scaling || !opacity || !dL_draw_scaling || !dL_draw_rotation ||
                          !dL_draw_opacity))) {
    gsr_set_error("activations backward: null buffer");
    return GSR_ERR_INVALID_ARGUMENT;
  }
  if (P == 0) return 0;
  GSR_LAUNCH("activations_bwd", k_activations_bwd, dim3((P + 255) / 256), dim3(256), 0, (hipStream_t)stream, P,
             (const float4*)raw_rotation, scaling, opacity, dL_dscaling, (const float4*)dL_drotation, dL_dopacity,
             dL_draw_scaling, (float4*)dL_draw_rotation, dL_draw_opacity);
  return gsr_launch_status("activations backward");
}
