// Device-scope integer atomics on a small table of counters - what a counting sort by tile id would cost on this part.
//   hipcc -O2 --offload-arch=gfx950 tools/atomic_microbench.hip -o tools/atomic_microbench && tools/atomic_microbench
// N atomic adds (returning and not) onto T counters, addresses random or clustered the way splats touch neighbouring tiles.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__device__ __forceinline__ uint32_t hash32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
  return x;
}

// one thread = one "Gaussian": k consecutive tiles of a random row segment (clustered) or k random tiles
template <bool RETURNING, bool CLUSTERED>
__global__ __launch_bounds__(256) void k_atomics(uint32_t* counters, uint32_t T, int grid_x, int k, uint32_t* out, int n_threads) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n_threads) return;
  const uint32_t h = hash32((uint32_t)i * 2654435761u + 17u);
  uint32_t acc = 0;
  for (int j = 0; j < k; j++) {
    uint32_t t;
    if (CLUSTERED) t = (h % T + (uint32_t)(j & 1) + (uint32_t)(j >> 1) * grid_x) % T;     // a 2-wide block of tiles going down
    else t = hash32(h + (uint32_t)j * 0x9E3779B9u) % T;
    if (RETURNING) acc += atomicAdd(&counters[t], 1u);
    else atomicAdd(&counters[t], 1u);
  }
  if (RETURNING) out[i] = acc;
}

template <bool R, bool C>
static float run(uint32_t* counters, uint32_t T, int k, uint32_t* out, int n_threads) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipMemset(counters, 0, T * 4);
  hipLaunchKernelGGL((k_atomics<R, C>), dim3((n_threads + 255) / 256), dim3(256), 0, 0, counters, T, 120, k, out, n_threads);
  hipMemset(counters, 0, T * 4);
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL((k_atomics<R, C>), dim3((n_threads + 255) / 256), dim3(256), 0, 0, counters, T, 120, k, out, n_threads);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  return ms;
}

int main() {
  const int n_threads = 1000000;
  uint32_t *counters, *out;
  CHECK(hipMalloc(&counters, 4 << 20));
  CHECK(hipMalloc(&out, (size_t)n_threads * 4));
  const uint32_t Ts[] = {8160, 32400, 1u << 20};
  for (uint32_t T : Ts)
    for (int k : {4, 8}) {
      printf("T = %7u counters, %d atomics per thread x 1 M threads (%.1f M atomics): ", T, k, n_threads * (double)k / 1e6);
      printf("random no-return %.1f us, random returning %.1f us, clustered no-return %.1f us, clustered returning %.1f us\n",
             run<false, false>(counters, T, k, out, n_threads) * 1e3, run<true, false>(counters, T, k, out, n_threads) * 1e3,
             run<false, true>(counters, T, k, out, n_threads) * 1e3, run<true, true>(counters, T, k, out, n_threads) * 1e3);
    }
  return 0;
}
