// (GPU box, developer tool) Does the cache policy of a 4-byte random gather change what an L2 miss costs?
// k_sweep / k_t0 on the SURVEY 8(d) workload (2^24 weights = 64 MB, beyond the 32 MB of aggregate L2) run at the chip's
// rate of gather MISSES (profiles/r04_a_*), so the request a miss sends to the fabric is what bounds them.
// Variants of the same kernel: plain global_load_dword, nt, sc1, sc0 sc1, sc0 sc1 nt -- rate by table size; run under
//   rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum
// to see the request sizes per variant (the kernel names differ by the template argument).
//   hipcc --offload-arch=gfx950 -O3 tools/micro/gather_policy.hip -o build/micro/gather_policy
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int POLICY>
__device__ __forceinline__ float load_w(const float* table, uint32_t idx) {
  float v;
  const uint32_t off = idx * 4u;
  if constexpr (POLICY == 0) asm volatile("global_load_dword %0, %1, %2" : "=v"(v) : "v"(off), "s"(table) : "memory");
  if constexpr (POLICY == 1) asm volatile("global_load_dword %0, %1, %2 nt" : "=v"(v) : "v"(off), "s"(table) : "memory");
  if constexpr (POLICY == 2) asm volatile("global_load_dword %0, %1, %2 sc1" : "=v"(v) : "v"(off), "s"(table) : "memory");
  if constexpr (POLICY == 3) asm volatile("global_load_dword %0, %1, %2 sc0 sc1" : "=v"(v) : "v"(off), "s"(table) : "memory");
  if constexpr (POLICY == 4) asm volatile("global_load_dword %0, %1, %2 sc0 sc1 nt" : "=v"(v) : "v"(off), "s"(table) : "memory");
  if constexpr (POLICY == 5) asm volatile("global_load_dword %0, %1, %2 sc0" : "=v"(v) : "v"(off), "s"(table) : "memory");
  if constexpr (POLICY == 6) asm volatile("global_load_dword %0, %1, %2 sc0 nt" : "=v"(v) : "v"(off), "s"(table) : "memory");
  if constexpr (POLICY == 7) asm volatile("global_load_dword %0, %1, %2 sc1 nt" : "=v"(v) : "v"(off), "s"(table) : "memory");
  return v;
}

template <int POLICY>
__global__ void __launch_bounds__(64) k_gather(const float* __restrict__ table, uint32_t mask, uint32_t per_lane, float* out) {
  uint64_t x = 0x9E3779B97F4A7C15ull * (blockIdx.x * blockDim.x + threadIdx.x + 1);
  float acc = 0.f;
  for (uint32_t i = 0; i < per_lane; i += 8) {
    uint32_t idx[8];
    float w[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      x ^= x << 13; x ^= x >> 7; x ^= x << 17;
      idx[j] = (uint32_t)(x >> 20) & mask;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) w[j] = load_w<POLICY>(table, idx[j]);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int j = 0; j < 8; ++j) acc += w[j];
  }
  if (acc == 12345.678f) out[0] = acc;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

template <int POLICY>
int run(const char* name, const float* tab, uint64_t n, int k, float* out) {
  const uint32_t blocks = 256 * 16 * 4, per_lane = 256;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e30f;
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_gather<POLICY>, dim3(blocks), dim3(64), 0, 0, tab, (uint32_t)(n - 1), per_lane, out);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  const double gathers = (double)blocks * 64 * per_lane;
  printf("table 2^%d floats (%7.0f MB)  %-12s (k_gather<%d>): %.3f ms = %.1f G gathers/s\n", k, n * 4 / 1048576.0, name, POLICY, best, gathers / best / 1e6);
  return 0;
}

int main(int argc, char** argv) {
  float* out;
  CK(hipMalloc(&out, 64));
  // allocation flavours: 0 = hipMalloc (cached in L2), 1 = hipDeviceMallocUncached, 2 = hipDeviceMallocFinegrained
  const int ks[] = {24, 26};
  for (int flavour = 0; flavour < 3; ++flavour) {
    for (int k : ks) {
      const uint64_t n = 1ull << k;
      float* tab = nullptr;
      if (flavour == 0) CK(hipMalloc(&tab, n * 4));
      if (flavour == 1) CK(hipExtMallocWithFlags((void**)&tab, n * 4, hipDeviceMallocUncached));
      if (flavour == 2) CK(hipExtMallocWithFlags((void**)&tab, n * 4, hipDeviceMallocFinegrained));
      CK(hipMemset(tab, 0, n * 4));
      printf("-- allocation: %s\n", flavour == 0 ? "hipMalloc" : flavour == 1 ? "hipDeviceMallocUncached" : "hipDeviceMallocFinegrained");
      run<0>("plain", tab, n, k, out);
      run<1>("nt", tab, n, k, out);
      run<3>("sc0 sc1", tab, n, k, out);
      CK(hipFree(tab));
    }
  }
  return 0;
}
