// (GPU box, developer tool) calibration of rocprofv3's FETCH_SIZE for the access pattern of the perceptron
// gathers: random 4-byte loads from a float table of 2^k entries (k = 22: the bench table, L2/Infinity-Cache
// resident; k = 26: 256 MB; k = 28: 1 GB, beyond the Infinity Cache), next to a coalesced 16 B/lane streaming
// read of the same number of bytes (the guide's reference pattern: FETCH_SIZE reports 1/2 of it on gfx950).
//   hipcc --offload-arch=gfx950 -O3 tools/micro/gather_calib.hip -o build/micro/gather_calib
//   rocprofv3 --pmc FETCH_SIZE -d out -o calib -- build/micro/gather_calib
// Each kernel prints the bytes it requested; tools/gpu_session_r02f.sh divides FETCH_SIZE by them.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

__global__ void k_gather(const float* __restrict__ table, uint32_t mask, uint32_t per_lane, float* out) {
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

__global__ void k_stream(const float4* __restrict__ table, uint64_t n4, float* out) {
  float acc = 0.f;
  for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n4; i += (uint64_t)gridDim.x * blockDim.x) {
    float4 v = table[i];
    acc += v.x + v.y + v.z + v.w;
  }
  if (acc == 12345.678f) out[0] = acc;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

int main() {
  float* out;
  CK(hipMalloc(&out, 64));
  const int ks[3] = {22, 26, 28};
  for (int t = 0; t < 3; ++t) {
    const uint64_t n = 1ull << ks[t];
    float* tab;
    CK(hipMalloc(&tab, n * 4));
    CK(hipMemset(tab, 0, n * 4));
    const uint32_t blocks = 256 * 16, threads = 256, per_lane = 256;   // 2^28 gathers = 1 GiB of 4-byte requests
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 2; ++rep) {
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(k_gather, dim3(blocks), dim3(threads), 0, 0, tab, (uint32_t)(n - 1), per_lane, out);
      CK(hipEventRecord(e1));
      CK(hipDeviceSynchronize());
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      const double gathers = (double)blocks * threads * per_lane;
      printf("gather k=%d rep=%d: %.0f gathers (%.0f bytes requested) in %.3f ms = %.1f G gathers/s\n", ks[t], rep, gathers, gathers * 4, ms, gathers / ms / 1e6);
    }
    for (int rep = 0; rep < 2; ++rep) {
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(k_stream, dim3(256 * 8), dim3(256), 0, 0, (const float4*)tab, n / 4, out);
      CK(hipEventRecord(e1));
      CK(hipDeviceSynchronize());
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      printf("stream k=%d rep=%d: %.0f bytes in %.3f ms = %.1f GB/s\n", ks[t], rep, (double)n * 4, ms, (double)n * 4 / ms / 1e6);
    }
    CK(hipFree(tab));
  }
  return 0;
}
