// mfma_coexec_microbench.hip - does an FP32-input MFMA (v_mfma_f32_16x16x4_f32) run BESIDE another wave's VALU work on the same
// SIMD of an MI355X, the way a bf16 MFMA does?  (Round 4: k_render_bwd_tile_mx moved the compositing backward's cross-lane sums
// onto the matrix pipe, 27 % fewer vector instructions - and got slower; PMC showed SQ_VALU_MFMA_COEXEC_CYCLES = 0.)
//
//   hipcc --offload-arch=gfx950 -O3 -o mfma_coexec_microbench mfma_coexec_microbench.hip && ./mfma_coexec_microbench
//
// Method: one 512-thread workgroup per CU (160 KiB of dynamic LDS keeps a second one out) = two waves per SIMD; waves 0..3 play
// role A (v_fma_f32, 64 independent instructions per trip), waves 4..7 role B (MFMAs on 4 independent accumulators, 16 per trip).
// Each role alone, then both together; s_memtime stamps per wave.  If the two pipes are independent, "together" costs each role
// what it cost alone; if the MFMA occupies the vector datapath, role A's time grows by role B's.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)
#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define REP64(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X)
typedef float f4 __attribute__((ext_vector_type(4)));
typedef short bf8 __attribute__((ext_vector_type(8)));

// mode bit 0: role A runs (fma), bit 1: role B runs; kind: 0 = f32 16x16x4 MFMA, 1 = bf16 16x16x32 MFMA
template <int KIND>
__global__ __launch_bounds__(512) void k_coexec(int mode, int iters_a, int iters_b, float seed, float* out, unsigned long long* cycles) {
  extern __shared__ float4 lds[];
  const int wave = threadIdx.x >> 6;
  const bool roleA = wave < 4;
  unsigned long long t0 = 0, t1 = 0;
  float sink = 0.f;
  if (roleA && (mode & 1)) {
    float r[8];
#pragma unroll
    for (int i = 0; i < 8; i++) r[i] = seed + (float)(threadIdx.x * 8 + i) * 1e-6f;
    const float a = 1.0000001f, b = 1e-9f;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int it = 0; it < iters_a; it++) {
#define X(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(a), "v"(b));
      REP64(X)
#undef X
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
#pragma unroll
    for (int i = 0; i < 8; i++) sink += r[i];
  } else if (!roleA && (mode & 2)) {
    f4 acc[4];
#pragma unroll
    for (int i = 0; i < 4; i++) acc[i] = f4{seed, seed, seed, seed};
    const float a = seed * 1e-3f, b = 1.0f + seed * 1e-6f;
    bf8 a8, b8;
#pragma unroll
    for (int i = 0; i < 8; i++) { a8[i] = (short)0x3c00; b8[i] = (short)0x3f80; }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int it = 0; it < iters_b; it++) {
#pragma unroll
      for (int u = 0; u < 4; u++) {
#pragma unroll
        for (int i = 0; i < 4; i++) {
          if (KIND == 0) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
          else acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a8, b8, acc[i], 0, 0, 0);
        }
      }
    }
    asm volatile("s_nop 15\n\ts_nop 15\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
#pragma unroll
    for (int i = 0; i < 4; i++) sink += acc[i][0] + acc[i][3];
  }
  out[(size_t)blockIdx.x * 512 + threadIdx.x] = sink;
  if ((threadIdx.x & 63) == 0) cycles[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int KIND>
static void run(const char* name, int mode, int iters_a, int iters_b, float* out, unsigned long long* cyc) {
  CHECK(hipFuncSetAttribute((const void*)k_coexec<KIND>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  std::vector<unsigned long long> host(256 * 8);
  for (int rep = 0; rep < 2; rep++) {   // (first launch warms the clocks)
    hipLaunchKernelGGL(k_coexec<KIND>, dim3(256), dim3(512), 160 * 1024, 0, mode, iters_a, iters_b, 1.0f, out, cyc);
    CHECK(hipDeviceSynchronize());
  }
  CHECK(hipMemcpy(host.data(), cyc, sizeof(unsigned long long) * 256 * 8, hipMemcpyDeviceToHost));
  std::vector<double> A, B;
  for (int b = 0; b < 256; b++)
    for (int w = 0; w < 8; w++) (w < 4 ? A : B).push_back((double)host[b * 8 + w]);
  std::sort(A.begin(), A.end());
  std::sort(B.begin(), B.end());
  const double a = A[A.size() / 2], b = B[B.size() / 2];
  printf("%-44s role A (v_fma_f32): %9.0f cycles (%5.2f / instr)   role B (MFMA): %9.0f cycles (%6.2f / MFMA)\n", name, a,
         (mode & 1) ? a / (64.0 * iters_a) : 0.0, b, (mode & 2) ? b / (16.0 * iters_b) : 0.0);
}

int main() {
  float* out;
  unsigned long long* cyc;
  CHECK(hipMalloc(&out, sizeof(float) * 256 * 512));
  CHECK(hipMalloc(&cyc, sizeof(unsigned long long) * 256 * 8));
  // role A: 2000 x 64 fmas = 128 k instructions (~4 cycles each alone); role B sized to take about as long alone
  const int ia = 2000;
  run<0>("fma alone", 1, ia, 0, out, cyc);
  run<0>("f32 MFMA 16x16x4 alone", 2, ia, 1000, out, cyc);
  run<0>("fma + f32 MFMA 16x16x4, same SIMD", 3, ia, 1000, out, cyc);
  run<1>("bf16 MFMA 16x16x32 alone", 2, ia, 2000, out, cyc);
  run<1>("fma + bf16 MFMA 16x16x32, same SIMD", 3, ia, 2000, out, cyc);
  CHECK(hipFree(out));
  CHECK(hipFree(cyc));
  return 0;
}
