// adam.hip - one-launch fused Adam over all Gaussian parameter groups.
//
//   dense  : torch.optim.Adam semantics (bias correction, eps added after the corrected sqrt), the reference's default
//            optimizer (scene/gaussian_model.py:169-170: Adam(l, lr=0.0, eps=1e-15)), all groups in ONE kernel launch
//            instead of the ~8 multi_tensor_apply launches PyTorch issues per step.
//   sparse : the `SparseGaussianAdam.step(visibility, N)` form the reference uses when available (train.py:37-41,
//            173-176): only rows of Gaussians with visibility != 0 are touched, NO bias correction
//            [published behaviour of the 3dgs_accel `adamUpdate`, SURVEY.md 2.2 N2].
// Pure HBM streaming: 16 B read + 12 B written per element; 16-B vector accesses when the tensor allows.
#include "gsr_common.h"

#define GSR_ADAM_MAX_TENSORS 8

struct AdamBatch {
  float* p[GSR_ADAM_MAX_TENSORS];
  const float* g[GSR_ADAM_MAX_TENSORS];
  float* m[GSR_ADAM_MAX_TENSORS];
  float* v[GSR_ADAM_MAX_TENSORS];
  long long n[GSR_ADAM_MAX_TENSORS];          // elements
  int row[GSR_ADAM_MAX_TENSORS];              // elements per Gaussian (visibility granularity)
  float lr[GSR_ADAM_MAX_TENSORS];
  float step_size[GSR_ADAM_MAX_TENSORS];      // lr / bias_correction1            (dense)
  float inv_bc2_sqrt[GSR_ADAM_MAX_TENSORS];   // 1 / sqrt(bias_correction2)       (dense)
  unsigned block_begin[GSR_ADAM_MAX_TENSORS + 1];
  int count;
};

#define ADAM_ELEMS_PER_BLOCK 4096   // 256 threads x 4 x float4

template <bool SPARSE>
__global__ __launch_bounds__(256) void k_adam(AdamBatch b, float beta1, float beta2, float omb1, float omb2, float eps,
                                              const uint8_t* __restrict__ visible) {
  GsrAdamArgs A;     // the per-element update is the one shared with the step folded into k_preprocess_bwd (gsr_common.h)
  A.beta1 = beta1; A.beta2 = beta2; A.omb1 = omb1; A.omb2 = omb2; A.eps = eps;
  int t = 0;
#pragma unroll
  for (int i = 1; i < GSR_ADAM_MAX_TENSORS; i++)
    if (i < b.count && blockIdx.x >= b.block_begin[i]) t = i;
  float* __restrict__ P = b.p[t];
  const float* __restrict__ G = b.g[t];
  float* __restrict__ M = b.m[t];
  float* __restrict__ V = b.v[t];
  const long long n = b.n[t];
  const int row = b.row[t];
  A.lr[0] = b.lr[t]; A.step_size[0] = b.step_size[t]; A.inv_bc2_sqrt[0] = b.inv_bc2_sqrt[t];
  const long long base = (long long)(blockIdx.x - b.block_begin[t]) * ADAM_ELEMS_PER_BLOCK;
  const bool vec_ok = (n % 4 == 0) && ((((uintptr_t)P | (uintptr_t)G | (uintptr_t)M | (uintptr_t)V) & 15) == 0);
#pragma unroll
  for (int it = 0; it < 4; it++) {
    const long long i0 = base + ((long long)it * 256 + threadIdx.x) * 4;
    if (i0 >= n) break;
    float p4[4], g4[4], m4[4], v4[4];
    const bool full = vec_ok && (i0 + 3 < n);
    if (full) {
      // streaming (non-temporal) accesses: 1.65 GB pass through once per step; keeping them out of L2 / the infinity
      // cache is worth 7 % here and leaves the next kernel's working set alone
      *(gsr_f4*)p4 = gsr_ld_stream(P + i0); *(gsr_f4*)g4 = gsr_ld_stream(G + i0);
      *(gsr_f4*)m4 = gsr_ld_stream(M + i0); *(gsr_f4*)v4 = gsr_ld_stream(V + i0);
    } else {
#pragma unroll
      for (int k = 0; k < 4; k++)
        if (i0 + k < n) { p4[k] = P[i0 + k]; g4[k] = G[i0 + k]; m4[k] = M[i0 + k]; v4[k] = V[i0 + k]; }
    }
    bool any = !SPARSE;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      if (i0 + k >= n) continue;
      if (SPARSE) {
        if (!visible[(i0 + k) / row]) continue;
        any = true;
        adam_elem<2>(p4[k], m4[k], v4[k], g4[k], A, 0);
      } else {
        // torch: exp_avg.lerp_(grad, 1 - beta1); 1 - beta formed in double on the host, as torch does
        adam_elem<1>(p4[k], m4[k], v4[k], g4[k], A, 0);
      }
    }
    if (!any) continue;
    if (full) {
      gsr_st_stream(P + i0, *(gsr_f4*)p4); gsr_st_stream(M + i0, *(gsr_f4*)m4); gsr_st_stream(V + i0, *(gsr_f4*)v4);
    } else {
#pragma unroll
      for (int k = 0; k < 4; k++)
        if (i0 + k < n) { P[i0 + k] = p4[k]; M[i0 + k] = m4[k]; V[i0 + k] = v4[k]; }
    }
  }
}

extern "C" {

// Dense Adam step (torch.optim.Adam semantics, amsgrad/weight_decay/maximize off) on `count` tensors in one launch.
// step[i] is the 1-based step number of tensor i AFTER this update (used for the bias corrections).
int gsr_adam_step(int32_t count, float* const* params, const float* const* grads, float* const* exp_avg,
                  float* const* exp_avg_sq, const int64_t* numel, const float* lr, const int64_t* step, double beta1,
                  double beta2, double eps, void* stream) {
  if (count < 0 || count > GSR_ADAM_MAX_TENSORS) {
    gsr_set_error("adam: at most %d tensors per call", GSR_ADAM_MAX_TENSORS);
    return GSR_ERR_INVALID_ARGUMENT;
  }
  AdamBatch b;
  b.count = count;
  unsigned blocks = 0;
  for (int i = 0; i < count; i++) {
    b.p[i] = params[i]; b.g[i] = grads[i]; b.m[i] = exp_avg[i]; b.v[i] = exp_avg_sq[i];
    b.n[i] = numel[i]; b.row[i] = 1; b.lr[i] = lr[i];
    const double bc1 = 1.0 - pow(beta1, (double)step[i]);
    const double bc2 = 1.0 - pow(beta2, (double)step[i]);
    b.step_size[i] = (float)((double)lr[i] / bc1);
    b.inv_bc2_sqrt[i] = (float)(1.0 / sqrt(bc2));
    b.block_begin[i] = blocks;
    blocks += (unsigned)((numel[i] + ADAM_ELEMS_PER_BLOCK - 1) / ADAM_ELEMS_PER_BLOCK);
  }
  b.block_begin[count] = blocks;
  if (blocks == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  GSR_LAUNCH("adam_dense", k_adam<false>, dim3(blocks), dim3(256), 0, st, b, (float)beta1, (float)beta2,
             (float)(1.0 - beta1), (float)(1.0 - beta2), (float)eps, (const uint8_t*)nullptr);
  return gsr_launch_status("adam launch");
}

// Sparse (visibility-masked) Adam: tensor i has N rows of numel[i]/N elements; rows with visible[row]==0 are untouched.
int gsr_sparse_adam_step(int32_t count, float* const* params, const float* const* grads, float* const* exp_avg,
                         float* const* exp_avg_sq, const int64_t* numel, const float* lr, int64_t N,
                         const uint8_t* visible, double beta1, double beta2, double eps, void* stream) {
  if (N == 0) return 0;     // an empty model (everything pruned): nothing to update
  if (count < 0 || count > GSR_ADAM_MAX_TENSORS || N < 0 || !visible) {
    gsr_set_error("sparse adam: bad arguments");
    return GSR_ERR_INVALID_ARGUMENT;
  }
  AdamBatch b;
  b.count = count;
  unsigned blocks = 0;
  for (int i = 0; i < count; i++) {
    if (numel[i] % N != 0) {
      gsr_set_error("sparse adam: tensor %d has %lld elements, not a multiple of N=%lld", i, (long long)numel[i],
                    (long long)N);
      return GSR_ERR_INVALID_ARGUMENT;
    }
    b.p[i] = params[i]; b.g[i] = grads[i]; b.m[i] = exp_avg[i]; b.v[i] = exp_avg_sq[i];
    b.n[i] = numel[i]; b.row[i] = (int)(numel[i] / N); b.lr[i] = lr[i];
    b.step_size[i] = 0.f; b.inv_bc2_sqrt[i] = 0.f;
    b.block_begin[i] = blocks;
    blocks += (unsigned)((numel[i] + ADAM_ELEMS_PER_BLOCK - 1) / ADAM_ELEMS_PER_BLOCK);
  }
  b.block_begin[count] = blocks;
  if (blocks == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  GSR_LAUNCH("adam_sparse", k_adam<true>, dim3(blocks), dim3(256), 0, st, b, (float)beta1, (float)beta2,
             (float)(1.0 - beta1), (float)(1.0 - beta2), (float)eps, visible);
  return gsr_launch_status("sparse adam launch");
}

}  // extern "C"
