// tile_atomic_microbench.hip - can the binning stage place its instances with per-tile atomic cursors?  One thread per Gaussian (1 M),
// each touching a w x h block of tiles around a random position of a 120 x 68 tile grid (4.5 instances per Gaussian on average, as
// at C3), two variants: atomicAdd without / with the returned value (+ an 8-byte store at base[tile] + returned position).
//   hipcc --offload-arch=gfx950 -O3 -o tile_atomic_microbench tile_atomic_microbench.hip && ./tile_atomic_microbench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

__device__ inline uint32_t hash(uint32_t x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }

template <int MODE>   // 0: count (no return), 1: place (returning atomic + store)
__global__ __launch_bounds__(256) void k_place(int P, int gx, int gy, uint32_t* __restrict__ cnt, const uint32_t* __restrict__ base,
                                               uint2* __restrict__ list, uint32_t cap) {
  const int g = blockIdx.x * 256 + threadIdx.x;
  if (g >= P) return;
  const uint32_t h = hash((uint32_t)g * 2654435761u + 12345u);
  const int tx = (int)(h % (uint32_t)gx), ty = (int)((h >> 12) % (uint32_t)gy);
  const int w = 1 + (int)((h >> 24) & 1) + (int)((h >> 25) & 1), hh = 1 + (int)((h >> 26) & 1) + (int)((h >> 27) & 1);   // 1..3 x 1..3: mean 4
  uint32_t k = 0;
  for (int y = ty; y < min(gy, ty + hh); y++)
    for (int x = tx; x < min(gx, tx + w); x++) {
      const int t = y * gx + x;
      if (MODE == 0) atomicAdd(&cnt[t], 1u);
      else {
        const uint32_t pos = atomicAdd(&cnt[t], 1u);
        const uint32_t at = base[t] + pos;
        if (at < cap) list[at] = make_uint2((uint32_t)g, k);
      }
      k++;
    }
}

int main() {
  const int P = 1000000, gx = 120, gy = 68, tiles = gx * gy;
  uint32_t *cnt, *base; uint2* list;
  const uint32_t cap = 6000000;
  CHECK(hipMalloc(&cnt, tiles * 4)); CHECK(hipMalloc(&base, tiles * 4)); CHECK(hipMalloc(&list, (size_t)cap * 8));
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  std::vector<uint32_t> h(tiles);
  for (int rep = 0; rep < 3; rep++) {
    CHECK(hipMemset(cnt, 0, tiles * 4));
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_place<0>, dim3((P + 255) / 256), dim3(256), 0, 0, P, gx, gy, cnt, base, list, cap);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms0; CHECK(hipEventElapsedTime(&ms0, e0, e1));
    CHECK(hipMemcpy(h.data(), cnt, tiles * 4, hipMemcpyDeviceToHost));
    std::vector<uint32_t> b(tiles); uint64_t tot = 0; uint32_t mx = 0;
    for (int t = 0; t < tiles; t++) { b[t] = (uint32_t)tot; tot += h[t]; mx = h[t] > mx ? h[t] : mx; }
    CHECK(hipMemcpy(base, b.data(), tiles * 4, hipMemcpyHostToDevice));
    CHECK(hipMemset(cnt, 0, tiles * 4));
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_place<1>, dim3((P + 255) / 256), dim3(256), 0, 0, P, gx, gy, cnt, base, list, cap);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms1; CHECK(hipEventElapsedTime(&ms1, e0, e1));
    printf("rep %d: %llu instances (%.2f per Gaussian, longest tile list %u): count pass (atomicAdd, no return) %.1f us; place pass (returning atomicAdd + 8-B store) %.1f us\n",
           rep, (unsigned long long)tot, (double)tot / P, mx, ms0 * 1e3, ms1 * 1e3);
  }
  return 0;
}
