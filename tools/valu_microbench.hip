// valu_microbench.hip - what one SIMD-32 of an MI355X really issues per cycle for the instruction kinds the compositing
// kernels are made of (render.hip): v_fma_f32, v_exp_f32, DPP-fused v_add_f32, v_permlane32_swap_b32, and an LDS broadcast read,
// at 1, 2, 4, 5 and 8 resident waves per SIMD.  bench.py's `roofline.valu` prices k_render_fwd / k_render_bwd against the
// v_fma_f32 rate measured here ("commit a micro-benchmark ... that measures the real per-SIMD rate").
//
//   hipcc --offload-arch=gfx950 -O3 -o valu_microbench valu_microbench.hip && ./valu_microbench
//
// Method: every workgroup is 256 threads = one wave per SIMD; a dynamic-LDS request of 160 KiB / k lets exactly k workgroups
// share a CU (k waves per SIMD); grid = 256 CUs x k.  Each wave runs ITERS x 64 independent instructions (8 rotating
// registers, so no dependency closer than 8 instructions) between two s_memtime stamps; cycles per wave-instruction on one
// SIMD = stamp difference / (ITERS x 64) / k is reported beside the wall-clock rate (which includes the clock the chip holds).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define REP64(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X)

enum { K_FMA = 0, K_EXP, K_DPP, K_SWAP, K_LDS, K_MIX, K_PKFMA, K_COUNT };
typedef float f2_t __attribute__((ext_vector_type(2)));
static const char* kNames[K_COUNT] = {"v_fma_f32", "v_exp_f32", "v_add_f32 dpp row_ror:8", "v_permlane32_swap_b32",
                                      "ds_read_b128 (broadcast)", "mix 5 fma : 1 exp : 2 dpp-add",
                                      "v_pk_fma_f32 (2 fma per lane)"};

template <int KIND>
__global__ __launch_bounds__(256) void k_bench(int iters, float seed, float* out, unsigned long long* cycles) {
  extern __shared__ float4 lds[];
  float r[8];
#pragma unroll
  for (int i = 0; i < 8; i++) r[i] = seed + (float)(threadIdx.x * 8 + i) * 1e-6f;
  if (KIND == K_LDS) {
    for (int i = threadIdx.x; i < 64; i += 256) lds[i] = make_float4(seed, seed, seed, seed);
    __syncthreads();
  }
  const float a = 1.0000001f, b = 1e-9f;
  unsigned long long t0, t1;
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  for (int it = 0; it < iters; it++) {
    if (KIND == K_FMA) {
#define X(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(a), "v"(b));
      REP64(X)
#undef X
    } else if (KIND == K_PKFMA) {
      f2_t* r2 = reinterpret_cast<f2_t*>(r);      // four register pairs
      const f2_t a2 = {a, a}, b2 = {b, b};
#define X(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(r2[(i) & 3]) : "v"(a2), "v"(b2));
      REP64(X)
#undef X
    } else if (KIND == K_EXP) {
#define X(i) asm volatile("v_exp_f32 %0, %0" : "+v"(r[i]));
      REP64(X)
#undef X
    } else if (KIND == K_DPP) {
#define X(i) asm volatile("v_add_f32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf" : "+v"(r[i]));
      REP64(X)
#undef X
    } else if (KIND == K_SWAP) {
#define X(i) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(r[i]), "+v"(r[(i + 4) & 7]));
      REP64(X)
#undef X
    } else if (KIND == K_LDS) {
      float4 q[8];
      const float4* base = lds + (it & 7);
#define X(i) q[i] = base[i * 4];
      REP8(X) REP8(X) REP8(X) REP8(X)
#undef X
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int i = 0; i < 8; i++) r[i] += q[i].x;
    } else {
      // the rough per-hit shape of the compositing loops: 5 fma-class : 1 transcendental : 2 cross-lane adds
#define X(i)                                                                                          \
  asm volatile("v_fma_f32 %0, %0, %2, %3\n\tv_fma_f32 %1, %1, %2, %3\n\tv_fma_f32 %0, %0, %2, %3\n\t" \
               "v_fma_f32 %1, %1, %2, %3\n\tv_fma_f32 %0, %0, %2, %3\n\tv_exp_f32 %1, %1\n\t"           \
               "v_add_f32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"                      \
               "v_add_f32_dpp %1, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"                \
               : "+v"(r[i]), "+v"(r[(i + 4) & 7])                                                       \
               : "v"(a), "v"(b));
      REP8(X)
#undef X
    }
  }
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; i++) s += r[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cycles[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int KIND>
static void run(int k, int iters, float* out, unsigned long long* cyc_dev, std::vector<unsigned long long>& cyc_host) {
  const int per_iter = (KIND == K_LDS) ? 32 : 64;   // instructions of the measured kind per iteration and wave
  const int grid = 256 * k;
  size_t lds = (160 * 1024) / k;
  lds = lds - (lds % 1024);
  if (k == 8) lds = 20 * 1024 - 512;   // leave room for the launch's own allocation granularity
  CHECK(hipFuncSetAttribute((const void*)k_bench<KIND>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL(k_bench<KIND>, dim3(grid), dim3(256), lds, 0, iters / 8 + 1, 1.0f, out, cyc_dev);   // warm-up
  CHECK(hipEventRecord(e0, 0));
  hipLaunchKernelGGL(k_bench<KIND>, dim3(grid), dim3(256), lds, 0, iters, 1.0f, out, cyc_dev);
  CHECK(hipEventRecord(e1, 0));
  CHECK(hipEventSynchronize(e1));
  float ms = 0.f;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  CHECK(hipMemcpy(cyc_host.data(), cyc_dev, sizeof(unsigned long long) * grid * 4, hipMemcpyDeviceToHost));
  std::vector<unsigned long long> c(cyc_host.begin(), cyc_host.begin() + grid * 4);
  std::sort(c.begin(), c.end());
  const double med = (double)c[c.size() / 2];
  const double n_per_wave = (double)iters * per_iter;
  const double n_total = n_per_wave * grid * 4;
  printf("  %-32s waves/SIMD %d : %6.2f cycles per wave-instr on one SIMD (s_memtime, median wave), wall %7.3f ms = %7.1f G "
         "wave-instr/s chip-wide\n",
         kNames[KIND], k, med / n_per_wave / k, ms, n_total / (ms * 1e-3) / 1e9);
}

int main() {
  const int iters = 2000;
  float* out;
  unsigned long long* cyc;
  CHECK(hipMalloc(&out, sizeof(float) * 256 * 8 * 256));
  CHECK(hipMalloc(&cyc, sizeof(unsigned long long) * 256 * 8 * 4));
  std::vector<unsigned long long> host(256 * 8 * 4);
  hipDeviceProp_t p;
  CHECK(hipGetDeviceProperties(&p, 0));
  printf("device: %s, %d CUs, clock %d kHz\n", p.gcnArchName, p.multiProcessorCount, p.clockRate);
  printf("peak by the guide: 2 cycles per wave64 VALU instruction on a SIMD-32 -> 256 CUs x 4 SIMDs x 2.4 GHz / 2 = 1228.8 G "
         "wave-instr/s\n");
  const int ks[] = {1, 2, 4, 5, 8};
  for (int k : ks) {
    run<K_FMA>(k, iters, out, cyc, host);
    run<K_EXP>(k, iters, out, cyc, host);
    run<K_DPP>(k, iters, out, cyc, host);
    run<K_SWAP>(k, iters, out, cyc, host);
    run<K_LDS>(k, iters, out, cyc, host);
    run<K_MIX>(k, iters, out, cyc, host);
    run<K_PKFMA>(k, iters, out, cyc, host);
  }
  CHECK(hipFree(out));
  CHECK(hipFree(cyc));
  return 0;
}
