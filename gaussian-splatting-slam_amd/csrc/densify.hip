// densify.hip - GaussianModel.densify_and_prune as two HIP passes.
//
// Reference: scene/gaussian_model.py:367-429 (densify_and_split :367, densify_and_clone :389, densify_and_prune :412) with
// the optimizer surgery of :274-344 (_prune_optimizer, cat_tensors_to_optimizer).  There it is ~20 small PyTorch kernels and
// three rounds of torch.cat / boolean-mask reallocation of every parameter AND both Adam moments.  Here:
//   k_densify_plan  : one thread per Gaussian decides keep / clone / split-children (same predicates, same order of tests)
//   (three prefix sums give every output row its position)
//   k_densify_apply : one thread per Gaussian writes its surviving rows of all six parameters and their exp_avg / exp_avg_sq
//                     straight into the new arrays (moments kept for survivors, zero for new rows)
// Output order = the reference's: [originals that are neither split nor pruned] ++ [clones] ++ [children copy 0] ++
// [children copy 1]  (`.repeat(N,1)` tiles the selected set, gaussian_model.py:375-385).
//
// What the reference's sequence reduces to (traced line by line):
//   grads = accum/denom, NaN -> 0                                           (:414-415)
//   clone  <=> |grads| >= max_grad and max(exp(scaling)) <= percent_dense*extent            (:390-392)
//   split  <=> grads >= max_grad and max(exp(scaling)) >  percent_dense*extent              (:369-373; clones have padded grad 0)
//   after both, prune <=> sigmoid(opacity) < min_opacity  or (max_screen_size and max(exp(scaling)) > 0.1*extent);
//   `max_radii2D > max_screen_size` can never fire: densification_postfix zeroes max_radii2D (:362-364) before it is read (:420).
//   A clone carries its source's opacity/scaling, a child the source's opacity and scaling/(0.8*2): the prune test of every
//   output row is therefore a function of the source row alone.
// Split samples: xyz + R(normalize(q)) . (exp(scaling) * z), z ~ N(0,1)^3 - a counter-based hash RNG (seed, source index, copy,
// axis) replaces torch.normal: same distribution, different stream (the reference's stream is not reproducible across devices
// either).
#include "gsr_common.h"

#define DENS_KEEP 1u
#define DENS_CLONE 2u
#define DENS_CHILD 4u

__global__ __launch_bounds__(256) void k_densify_plan(int P, const float* __restrict__ grad_accum,
                                                      const float* __restrict__ denom, const float* __restrict__ scaling,
                                                      const float* __restrict__ opacity, float max_grad, float min_opacity,
                                                      float extent, float percent_dense, int use_world_size,
                                                      uint32_t* __restrict__ f_keep, uint32_t* __restrict__ f_clone,
                                                      uint32_t* __restrict__ f_child) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= P) return;
  float g = grad_accum[i] / denom[i];
  if (g != g) g = 0.f;                                             // grads[grads.isnan()] = 0.0
  const float s0 = expf(scaling[3 * (size_t)i]), s1 = expf(scaling[3 * (size_t)i + 1]), s2 = expf(scaling[3 * (size_t)i + 2]);
  const float smax = fmaxf(s0, fmaxf(s1, s2));
  const bool big = smax > percent_dense * extent;
  const bool sel = g >= max_grad;
  const bool clone = sel && !big;
  const bool split = sel && big;
  const float op = 1.0f / (1.0f + expf(-opacity[i]));
  const bool low = op < min_opacity;
  const bool prune_self = low || (use_world_size && smax > 0.1f * extent);
  // children: scaling / (0.8 * 2); exp(log(x)) round trip as in the reference (scaling_inverse_activation then get_scaling)
  const float cmax = expf(logf(smax / 1.6f));
  const bool prune_child = low || (use_world_size && cmax > 0.1f * extent);
  f_keep[i] = (!split && !prune_self) ? 1u : 0u;
  f_clone[i] = (clone && !prune_self) ? 1u : 0u;
  f_child[i] = (split && !prune_child) ? 1u : 0u;
}

__device__ __forceinline__ uint32_t hash_u32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}
// standard normal from a counter (Box-Muller on two hashed uniforms)
__device__ __forceinline__ float normal_from_counter(uint32_t seed, uint32_t idx, uint32_t stream) {
  const uint32_t a = hash_u32(seed ^ hash_u32(idx * 2654435761u + stream * 0x9E3779B9u));
  const uint32_t b = hash_u32(a ^ 0x68bc21ebu ^ (stream << 8));
  const float u1 = ((a >> 8) + 1u) * (1.0f / 16777216.0f);        // (0,1]
  const float u2 = (b >> 8) * (1.0f / 16777216.0f);               // [0,1)
  return sqrtf(-2.0f * logf(u1)) * cospif(2.0f * u2);
}

struct DensTensors {
  // six parameters x (value, exp_avg, exp_avg_sq): in / out pointers and floats per row
  const float* in[18];
  float* out[18];
  int row[6];
};

__device__ __forceinline__ void copy_row(const float* __restrict__ src, float* __restrict__ dst, int n) {
  for (int k = 0; k < n; k++) dst[k] = src[k];
}
__device__ __forceinline__ void zero_row(float* __restrict__ dst, int n) {
  for (int k = 0; k < n; k++) dst[k] = 0.f;
}

// parameter order: 0 xyz(3) 1 f_dc(3) 2 f_rest(R) 3 opacity(1) 4 scaling(3) 5 rotation(4)
__global__ __launch_bounds__(256) void k_densify_apply(int P, DensTensors t, const uint32_t* __restrict__ f_keep,
                                                       const uint32_t* __restrict__ f_clone,
                                                       const uint32_t* __restrict__ f_child,
                                                       const uint32_t* __restrict__ p_keep,
                                                       const uint32_t* __restrict__ p_clone,
                                                       const uint32_t* __restrict__ p_child, uint32_t n_keep,
                                                       uint32_t n_clone, uint32_t n_child, uint32_t seed,
                                                       int32_t* __restrict__ source_of_row) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= P) return;
  const bool keep = f_keep[i], clone = f_clone[i], child = f_child[i];
  if (keep) {
    const size_t o = p_keep[i];
#pragma unroll
    for (int p = 0; p < 6; p++) {
      const int n = t.row[p];
#pragma unroll
      for (int q = 0; q < 3; q++)
        if (t.out[3 * p + q]) copy_row(t.in[3 * p + q] + (size_t)i * n, t.out[3 * p + q] + o * n, n);
    }
    if (source_of_row) source_of_row[o] = i;
  }
  if (clone) {
    const size_t o = (size_t)n_keep + p_clone[i];
#pragma unroll
    for (int p = 0; p < 6; p++) {
      const int n = t.row[p];
      copy_row(t.in[3 * p] + (size_t)i * n, t.out[3 * p] + o * n, n);
      if (t.out[3 * p + 1]) zero_row(t.out[3 * p + 1] + o * n, n);
      if (t.out[3 * p + 2]) zero_row(t.out[3 * p + 2] + o * n, n);
    }
    if (source_of_row) source_of_row[o] = i;
  }
  if (child) {
    // R from the NORMALISED raw quaternion (build_rotation, utils/general_utils.py:78-99)
    const float* q = t.in[15] + 4 * (size_t)i;
    const float qn = rsqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    const float r = q[0] * qn, x = q[1] * qn, y = q[2] * qn, z = q[3] * qn;
    const float R[9] = {1.f - 2.f * (y * y + z * z), 2.f * (x * y - r * z),       2.f * (x * z + r * y),
                        2.f * (x * y + r * z),       1.f - 2.f * (x * x + z * z), 2.f * (y * z - r * x),
                        2.f * (x * z - r * y),       2.f * (y * z + r * x),       1.f - 2.f * (x * x + y * y)};
    const float* sc = t.in[12] + 3 * (size_t)i;
    const float s[3] = {expf(sc[0]), expf(sc[1]), expf(sc[2])};
    const float* xyz = t.in[0] + 3 * (size_t)i;
    for (int c = 0; c < 2; c++) {
      const size_t o = (size_t)n_keep + n_clone + (size_t)c * n_child + p_child[i];
      float smp[3];
#pragma unroll
      for (int a = 0; a < 3; a++) smp[a] = s[a] * normal_from_counter(seed, (uint32_t)i, (uint32_t)(c * 3 + a));
#pragma unroll
      for (int a = 0; a < 3; a++)
        t.out[0][o * 3 + a] = R[3 * a] * smp[0] + R[3 * a + 1] * smp[1] + R[3 * a + 2] * smp[2] + xyz[a];
#pragma unroll
      for (int a = 0; a < 3; a++) t.out[12][o * 3 + a] = logf(s[a] / 1.6f);   // scaling_inverse_activation(s/(0.8*N))
      // the other parameters are repeated
      const int others[4] = {1, 2, 3, 5};
#pragma unroll
      for (int w = 0; w < 4; w++) {
        const int p = others[w], n = t.row[p];
        copy_row(t.in[3 * p] + (size_t)i * n, t.out[3 * p] + o * n, n);
      }
#pragma unroll
      for (int p = 0; p < 6; p++) {
        const int n = t.row[p];
        if (t.out[3 * p + 1]) zero_row(t.out[3 * p + 1] + o * n, n);
        if (t.out[3 * p + 2]) zero_row(t.out[3 * p + 2] + o * n, n);
      }
      if (source_of_row) source_of_row[o] = i;
    }
  }
}

// add_densification_stats (reference scene/gaussian_model.py:431-433) + the max_radii2D update of train.py:159 in one pass:
// for visible Gaussians (radii > 0): accum += ||dL/dmeans2D.xy||, denom += 1, max_radii = max(max_radii, radii).
__global__ __launch_bounds__(256) void k_densify_stats(int P, const float* __restrict__ grad_means2D,
                                                       const int32_t* __restrict__ radii, float* __restrict__ accum,
                                                       float* __restrict__ denom, float* __restrict__ max_radii) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= P) return;
  const int r = radii[i];
  if (r > 0) {
    const float gx = grad_means2D[3 * (size_t)i], gy = grad_means2D[3 * (size_t)i + 1];
    gsr_densify_stats_update(gx, gy, r, accum + i, denom + i, max_radii + i);
  }
}

extern "C" {

int gsr_densification_stats(int64_t P, const float* grad_means2D, const int32_t* radii, float* xyz_gradient_accum,
                            float* denom, float* max_radii2D, void* stream) {
  if (P < 0 || (P > 0 && (!grad_means2D || !radii || !xyz_gradient_accum || !denom || !max_radii2D))) {
    gsr_set_error("densification_stats: bad arguments");
    return GSR_ERR_INVALID_ARGUMENT;
  }
  if (P == 0) return 0;
  GSR_LAUNCH("densify_stats", k_densify_stats, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
             (int)P, grad_means2D, radii, xyz_gradient_accum, denom, max_radii2D);
  return gsr_launch_status("densification_stats launch");
}

size_t gsr_densify_workspace_bytes(int64_t P) {
  const size_t p = (size_t)(P < 1 ? 1 : P);
  return 6 * gsr_align(p * 4) + gsr_align(gsr_scan_tmp_elems(p) * 4) + 256;
}

// Plan + prefix sums.  Synchronises the stream once to return the three counts (host int64[3]: keep, clone, child pairs).
int gsr_densify_plan(int64_t P, const float* xyz_gradient_accum, const float* denom, const float* scaling_raw,
                     const float* opacity_raw, float max_grad, float min_opacity, float extent, float percent_dense,
                     int32_t use_world_size_prune, void* workspace, size_t workspace_bytes, int64_t* counts_host,
                     void* stream) {
  if (P < 0 || !counts_host || (P > 0 && (!xyz_gradient_accum || !denom || !scaling_raw || !opacity_raw || !workspace))) {
    gsr_set_error("densify_plan: bad arguments");
    return GSR_ERR_INVALID_ARGUMENT;
  }
  counts_host[0] = counts_host[1] = counts_host[2] = 0;
  if (P == 0) return 0;
  if (P > 0x7FFFFFFF || workspace_bytes < gsr_densify_workspace_bytes(P)) {
    gsr_set_error("densify_plan: workspace too small or P too large");
    return GSR_ERR_STATE_TOO_SMALL;
  }
  hipStream_t st = (hipStream_t)stream;
  char* ws = (char*)workspace;
  const size_t a = gsr_align((size_t)P * 4);
  uint32_t* f[3] = {(uint32_t*)ws, (uint32_t*)(ws + a), (uint32_t*)(ws + 2 * a)};
  uint32_t* pfx[3] = {(uint32_t*)(ws + 3 * a), (uint32_t*)(ws + 4 * a), (uint32_t*)(ws + 5 * a)};
  uint32_t* scan_tmp = (uint32_t*)(ws + 6 * a);
  GSR_LAUNCH("densify_plan", k_densify_plan, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, st, (int)P,
             xyz_gradient_accum, denom, scaling_raw, opacity_raw, max_grad, min_opacity, extent, percent_dense,
             (int)use_world_size_prune, f[0], f[1], f[2]);
  for (int k = 0; k < 3; k++) gsr_scan_u32(f[k], nullptr, pfx[k], (size_t)P, 0, scan_tmp, st);
  uint32_t last_f[3], last_p[3];
  for (int k = 0; k < 3; k++) {
    int rc;
    if ((rc = gsr_check(hipMemcpyAsync(&last_f[k], f[k] + (P - 1), 4, hipMemcpyDeviceToHost, st), "densify readback")))
      return rc;
    if ((rc = gsr_check(hipMemcpyAsync(&last_p[k], pfx[k] + (P - 1), 4, hipMemcpyDeviceToHost, st), "densify readback")))
      return rc;
  }
  int rc = gsr_check(hipStreamSynchronize(st), "densify sync");
  if (rc) return rc;
  for (int k = 0; k < 3; k++) counts_host[k] = (int64_t)last_f[k] + (int64_t)last_p[k];
  return 0;
}

// Gather.  in_ptrs/out_ptrs: HOST arrays of 18 device pointers, parameter-major: (xyz, f_dc, f_rest, opacity, scaling,
// rotation) x (value, exp_avg, exp_avg_sq); exp_avg / exp_avg_sq entries may be NULL (no optimizer state yet).
// row_floats: HOST int32[6].  source_of_row (optional, device int32[new P]) receives the source index of every output row.
int gsr_densify_apply(int64_t P, const void* workspace, const float* const* in_ptrs, float* const* out_ptrs,
                      const int32_t* row_floats, int64_t n_keep, int64_t n_clone, int64_t n_child, uint32_t seed,
                      int32_t* source_of_row, void* stream) {
  if (P < 0 || !in_ptrs || !out_ptrs || !row_floats) {
    gsr_set_error("densify_apply: bad arguments");
    return GSR_ERR_INVALID_ARGUMENT;
  }
  // nothing to read, or nothing survives (every row pruned, none cloned or split): no row to write
  if (P == 0 || n_keep + n_clone + n_child == 0) return 0;
  DensTensors t;
  for (int p = 0; p < 6; p++) t.row[p] = row_floats[p];
  for (int k = 0; k < 18; k++) {
    const bool empty_rows = t.row[k / 3] == 0;      // e.g. f_rest at SH degree 0: [P, 0, 3] - its pointers may be NULL
    t.in[k] = empty_rows ? nullptr : in_ptrs[k];
    t.out[k] = empty_rows ? nullptr : out_ptrs[k];
    if (empty_rows) continue;
    if ((k % 3) != 0 && (in_ptrs[k] == nullptr) != (out_ptrs[k] == nullptr)) {
      gsr_set_error("densify_apply: optimizer state %d present on one side only", k);
      return GSR_ERR_INVALID_ARGUMENT;
    }
    if ((k % 3) == 0 && (!in_ptrs[k] || !out_ptrs[k])) {
      gsr_set_error("densify_apply: parameter %d missing", k / 3);
      return GSR_ERR_INVALID_ARGUMENT;
    }
  }
  if (t.row[0] != 3 || t.row[4] != 3 || t.row[5] != 4) {
    gsr_set_error("densify_apply: xyz/scaling/rotation rows must be 3/3/4 floats");
    return GSR_ERR_INVALID_ARGUMENT;
  }
  const char* ws = (const char*)workspace;
  const size_t a = gsr_align((size_t)P * 4);
  hipStream_t st = (hipStream_t)stream;
  GSR_LAUNCH("densify_apply", k_densify_apply, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, st, (int)P, t,
             (const uint32_t*)ws, (const uint32_t*)(ws + a), (const uint32_t*)(ws + 2 * a), (const uint32_t*)(ws + 3 * a),
             (const uint32_t*)(ws + 4 * a), (const uint32_t*)(ws + 5 * a), (uint32_t)n_keep, (uint32_t)n_clone,
             (uint32_t)n_child, seed, source_of_row);
  return gsr_launch_status("densify_apply launch");
}

}  // extern "C"
