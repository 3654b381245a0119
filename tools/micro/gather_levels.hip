// (GPU box, developer tool) random 4-byte gathers from float tables of 2^k entries, k = 10 .. 24: the chip's
// gather rate by cache level (4 KB: L1; 64 KB .. 2 MB: one XCD's L2; 16 MB: the bench weight table, Infinity
// Cache; 64 MB) with 16 wavefronts per CU as in k_sweep, all 64 lanes active and 40 of 64 active (the sweep's
// typical occupancy of a gather instruction).  Answers "is k_sweep (375 G gathers/s) near a gather-issue
// ceiling?" (DESIGN section 4).
//   hipcc --offload-arch=gfx950 -O3 tools/micro/gather_levels.hip -o build/micro/gather_levels
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int ACTIVE>
__global__ void __launch_bounds__(64) k_gather(const float* __restrict__ table, uint32_t mask, uint32_t per_lane, float* out) {
  if ((int)(threadIdx.x & 63) >= ACTIVE) return;
  uint64_t x = 0x9E3779B97F4A7C15ull * (blockIdx.x * blockDim.x + threadIdx.x + 1);
  float acc = 0.f;
  for (uint32_t i = 0; i < per_lane; i += 8) {
    uint32_t idx[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      x ^= x << 13; x ^= x >> 7; x ^= x << 17;
      idx[j] = (uint32_t)(x >> 20) & mask;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) acc += table[idx[j]];
  }
  if (acc == 12345.678f) out[0] = acc;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

int main() {
  float* out;
  CK(hipMalloc(&out, 64));
  const int ks[] = {10, 14, 17, 19, 20, 22, 24};
  for (int k : ks) {
    const uint64_t n = 1ull << k;
    float* tab;
    CK(hipMalloc(&tab, n * 4));
    CK(hipMemset(tab, 0, n * 4));
    const uint32_t blocks = 256 * 16 * 4, per_lane = 512;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int act = 0; act < 2; ++act) {
      float best = 1e30f;
      for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0));
        if (act == 0) hipLaunchKernelGGL(k_gather<64>, dim3(blocks), dim3(64), 0, 0, tab, (uint32_t)(n - 1), per_lane, out);
        else hipLaunchKernelGGL(k_gather<40>, dim3(blocks), dim3(64), 0, 0, tab, (uint32_t)(n - 1), per_lane, out);
        CK(hipEventRecord(e1));
        CK(hipDeviceSynchronize());
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
      }
      const double gathers = (double)blocks * (act == 0 ? 64 : 40) * per_lane;
      printf("table 2^%d floats (%8.0f KB), %d of 64 lanes: %.3f ms = %.1f G gathers/s (%.1f G gather instructions/s)\n", k, n * 4 / 1024.0,
             act == 0 ? 64 : 40, best, gathers / best / 1e6, (double)blocks * per_lane / best / 1e6);
    }
    CK(hipFree(tab));
  }
  return 0;
}
