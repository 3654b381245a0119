// Runtime layer: the product build is HIP/gfx950 only.  The JPP_EMU branch
// exists solely so that tests/emu can run the same kernel sources on a CPU
// fiber emulator (tests/emu/hip_emu.h); it is never part of libjppgpu.so.
#ifndef JPP_RT_H
#define JPP_RT_H

#include <cstddef>
#include <cstdint>

#if defined(JPP_EMU)
#include "hip_emu.h"
typedef void* jpp_stream_t;
#define JPP_LAUNCH(kernel, grid, block, stream, ...) \
  hip_emu::launch(dim3(grid), dim3(block), [=]() { kernel(__VA_ARGS__); })
#define JPP_LAUNCH_LDS(kernel, grid, block, lds_bytes, stream, ...) JPP_LAUNCH(kernel, grid, block, stream, __VA_ARGS__)
#else
#include <hip/hip_runtime.h>
typedef hipStream_t jpp_stream_t;
#define JPP_LAUNCH(kernel, grid, block, stream, ...) \
  hipLaunchKernelGGL(kernel, dim3(grid), dim3(block), 0, stream, __VA_ARGS__)
// with `lds_bytes` of dynamic LDS on top of the kernel's own (occupancy experiments: tools/gpu_session_r03a.sh)
#define JPP_LAUNCH_LDS(kernel, grid, block, lds_bytes, stream, ...) \
  hipLaunchKernelGGL(kernel, dim3(grid), dim3(block), lds_bytes, stream, __VA_ARGS__)
#endif

// developer phase timers: real only in a -DJPP_DEV_PROF build of the HIP library (dev/jpp_dev_prof.h)
#if defined(JPP_DEV_PROF) && !defined(JPP_EMU)
#include "dev/jpp_dev_prof.h"
#else
#define JPP_PROF_DECL
#define JPP_PROF(i)
#define JPP_PROF_FLUSH
#define JPP_RPROF_DECL
#define JPP_RPROF(i)
#define JPP_RPROF_COUNT(cn)
#define JPP_RPROF_FLUSH
#define JPP_PPROF_DECL
#define JPP_PPROF(i)
#define JPP_PPROF_FLUSH
#endif

typedef uint8_t u8;
typedef uint16_t u16;
typedef uint32_t u32;
typedef uint64_t u64;
typedef int32_t i32;
typedef int64_t i64;

// Pointers read out of a structure in memory (DevModel) are generic to the compiler, and loads through
// them become flat_load: those tick the LDS counter as well as the vector-memory counter, so every wait for
// an LDS read behind a weight gather would also wait for the gather's HBM round trip.  The model tables are
// re-typed as global (address space 1) where the hot kernels pick them up.
#if defined(JPP_EMU) || !defined(__HIP_DEVICE_COMPILE__)
#define JPP_GLOBAL
#else
#define JPP_GLOBAL __attribute__((address_space(1)))
#endif

namespace jpp {

template <typename T>
__device__ __forceinline__ const T JPP_GLOBAL* as_global(const T* p) {
  return (const T JPP_GLOBAL*)p;
}

// ... and a pointer into LDS typed as such.  Where one branch reads a record from LDS and the other the same field
// from HBM, the compiler otherwise merges the two loads into one flat_load through a selected pointer: an LDS read
// through the vector-memory path, with the vmcnt(0) waits that come with it.
#if defined(JPP_EMU) || !defined(__HIP_DEVICE_COMPILE__)
#define JPP_LDS
#else
#define JPP_LDS __attribute__((address_space(3)))
#endif
template <typename T>
__device__ __forceinline__ const T JPP_LDS* as_lds(const T* p) {
  return (const T JPP_LDS*)p;
}

// ---- wavefront (64 lanes) helpers -------------------------------------------
__device__ __forceinline__ int lane_id() {
#if defined(JPP_EMU)
  return hip_emu::lane();
#else
  return (int)(threadIdx.x & 63);
#endif
}

// The lane index recomputed on the spot (v_mbcnt_lo / v_mbcnt_hi, two instructions) and opaque to the optimiser.
// k_sweep's boundary loop takes its lane index from here at the top of every iteration: whatever is derived from a
// lane index that lives outside the loop (lane * 16, lane >> 3, LDS addresses per array, ...) is hoisted out of the
// loop by LLVM and kept in registers across it -- sixty-odd VGPRs in round 2, the difference between four and five
// wavefronts per SIMD -- and a spilled lane index costs a scratch reload, which on this in-order memory pipeline also
// waits for every prefetch in flight.
__device__ __forceinline__ int lane_now() {
#if defined(JPP_EMU)
  return hip_emu::lane();
#elif defined(__HIP_DEVICE_COMPILE__)
  int v = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
  asm volatile("" : "+v"(v));
  return v;
#else
  return 0;
#endif
}

// Synchronise the lanes of ONE wavefront around LDS traffic (multi-wave workgroups whose
// waves work on independent sentences cannot use the workgroup barrier inside ragged loops).
// LDS operations of a wave complete in order; the fences make the compiler emit the waits.
__device__ __forceinline__ void wave_sync() {
#if defined(JPP_EMU)
  hip_emu::wave_barrier();
#else
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#endif
}

// Order this wavefront's global stores and atomics before its own later loads (other LANES read what a lane wrote):
// workgroup scope, which is the waits alone.  __threadfence() is device scope -- `buffer_wbl2 sc1` + `buffer_inv sc1`, a
// write-back and an invalidate of the XCD's whole L2 per call, paid by every wavefront of the chip that shares it
// (k_lat_count with three of them per sentence: 8 192 sentences took 7 x the time of 256; k_decode's note).
__device__ __forceinline__ void wave_fence_global() {
#if !defined(JPP_EMU)
  __threadfence_block();
#endif
}

// A load served by L2, not by the CU's vector cache: for words this wavefront has just changed with atomics (they
// execute in L2; a neighbour wavefront of the CU may have left the line in the vector cache)
template <typename T>
__device__ __forceinline__ T load_l2(const T* p) {
#if defined(JPP_EMU) || !defined(__HIP_DEVICE_COMPILE__)
  return *p;
#else
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
}

// Asynchronous global -> LDS copy (gfx950 `global_load_lds_dword` / `_dwordx4`): every lane for which
// `act` holds copies BYTES (4 or 16) from its own global address to lds_base + lane * BYTES without
// passing through VGPRs.  lds_base must be wave-uniform.  The data is visible after lds_async_wait()
// (a workgroup barrier of a single-wavefront workgroup does not imply the vmcnt wait).
template <int BYTES>
__device__ __forceinline__ void lds_async_load(void* lds_base, const void* gptr, bool act) {
  static_assert(BYTES == 4 || BYTES == 16, "global_load_lds moves a dword or four dwords per lane");
#if defined(JPP_EMU)
  if (act) __builtin_memcpy(static_cast<char*>(lds_base) + (size_t)lane_id() * BYTES, gptr, BYTES);
#else
#if defined(__HIP_DEVICE_COMPILE__)
  if (act) {
    auto g = (const void __attribute__((address_space(1)))*)gptr;
    auto l = (void __attribute__((address_space(3)))*)lds_base;
    if constexpr (BYTES == 16) __builtin_amdgcn_global_load_lds(g, l, 16, 0, 0);
    else __builtin_amdgcn_global_load_lds(g, l, 4, 0, 0);
  }
#else
  (void)lds_base; (void)gptr; (void)act;
#endif
#endif
}

// wait until every lds_async_load of this wavefront has landed in LDS
__device__ __forceinline__ void lds_async_wait() {
#if !defined(JPP_EMU) && defined(__HIP_DEVICE_COMPILE__)
  __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0), expcnt/lgkmcnt untouched (gfx9 encoding)
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  asm volatile("" ::: "memory");
#endif
}

// wait for every outstanding global load / store of this wavefront
__device__ __forceinline__ void vm_wait_all() {
#if !defined(JPP_EMU) && defined(__HIP_DEVICE_COMPILE__)
  __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0), expcnt/lgkmcnt untouched (gfx9 encoding)
#endif
}

// occupancy target of a kernel (wavefronts per SIMD): caps its VGPRs accordingly
#if defined(JPP_EMU)
#define JPP_WAVES_PER_EU(n)
#define JPP_WAVES_PER_EU_RANGE(lo, hi)
#else
#define JPP_WAVES_PER_EU(n) __attribute__((amdgpu_waves_per_eu(n)))
#define JPP_WAVES_PER_EU_RANGE(lo, hi) __attribute__((amdgpu_waves_per_eu(lo, hi)))
#endif

// workgroup barrier that orders LDS traffic only: outstanding global loads / stores of the wavefront stay in flight
// (__syncthreads() drains them: its workgroup-scope release waits for vmcnt(0)).  For phases that hand data to other
// wavefronts through LDS alone.
__device__ __forceinline__ void lds_barrier() {
#if defined(JPP_EMU)
  __syncthreads();
#elif defined(__HIP_DEVICE_COMPILE__)
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#endif
}

// the SIMD (0..3) of the compute unit this wavefront runs on: HW_ID bits 5:4
__device__ __forceinline__ u32 wave_simd_id() {
#if defined(JPP_EMU)
  return (threadIdx.x >> 6) & 3u;
#elif defined(__HIP_DEVICE_COMPILE__)
  return (u32)__builtin_amdgcn_s_getreg((1 << 11) | (4 << 6) | 4);   // 2 bits at offset 4 of HW_REG_HW_ID
#else
  return 0;
#endif
}

// v_mfma_f32_16x16x4_f32: D = A(16x4) * B(4x16) + C on the matrix cores.  Lane l supplies A[l & 15][l >> 4] and
// B[l >> 4][l & 15] and holds D[4 * (l >> 4) + i][l & 15] in element i.  Every D element is the f32 fused chain
// fma(a3, b3, fma(a2, b2, fma(a1, b1, fma(a0, b0, c)))): one rounding per step, k ascending, nothing wider inside.
struct MfmaAcc {
  float v[4];
};
__device__ __forceinline__ MfmaAcc mfma_f32_16x16x4(float a, float b, MfmaAcc c) {
#if defined(JPP_EMU)
  return MfmaAcc{hip_emu::mfma_f32_16x16x4(a, b, c.v[0], 0), hip_emu::mfma_f32_16x16x4(a, b, c.v[1], 1),
                 hip_emu::mfma_f32_16x16x4(a, b, c.v[2], 2), hip_emu::mfma_f32_16x16x4(a, b, c.v[3], 3)};
#elif defined(__HIP_DEVICE_COMPILE__)
  typedef float f32x4_t __attribute__((ext_vector_type(4)));
  f32x4_t cc = {c.v[0], c.v[1], c.v[2], c.v[3]};
  cc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, cc, 0, 0, 0);
  return MfmaAcc{cc[0], cc[1], cc[2], cc[3]};
#else
  (void)a;
  (void)b;
  return c;
#endif
}

__device__ __forceinline__ u64 wave_ballot(bool p) {
#if defined(JPP_EMU)
  return hip_emu::ballot(p);
#else
  return __ballot(p);
#endif
}

__device__ __forceinline__ u32 wave_shfl_u32(u32 v, int src) {
#if defined(JPP_EMU)
  return (u32)hip_emu::shfl_u64(v, src);
#else
  return (u32)__shfl((int)v, src, 64);
#endif
}

__device__ __forceinline__ float wave_shfl_f32(float v, int src) {
  u32 x;
  __builtin_memcpy(&x, &v, 4);
  x = wave_shfl_u32(x, src);
  float r;
  __builtin_memcpy(&r, &x, 4);
  return r;
}

// value held by the lane N positions above this one (N = 1..15) inside its row of 16 lanes, as a DPP
// modifier (row_shl:N) -- no LDS crossbar traffic, unlike __shfl/ds_bpermute.  Lanes whose source would
// fall outside the row receive 0.  Used for the sums inside 8-lane groups: the group leader (lane 0 or 8
// of the row) reads its members 1..7 this way.
template <int N>
__device__ __forceinline__ float row_shl_f32(float v) {
  static_assert(N >= 1 && N <= 15, "row_shl:1..15");
#if defined(JPP_EMU)
  const int l = lane_id();
  float r = wave_shfl_f32(v, (l + N) & 63);
  return ((l & 15) + N) > 15 ? 0.f : r;
#else
  int x;
  __builtin_memcpy(&x, &v, 4);
  x = __builtin_amdgcn_update_dpp(0, x, 0x100 + N, 0xf, 0xf, false);
  float r;
  __builtin_memcpy(&r, &x, 4);
  return r;
#endif
}

// A value that IS the same in every lane (a boundary record read from LDS at a uniform address, a count derived from
// one), moved to a scalar register: loops over it become scalar loops (s_cmp / s_cbranch instead of an exec-mask
// dance per iteration), comparisons take an SGPR operand, address arithmetic goes to the scalar unit and the VGPR is
// free.  The caller guarantees the uniformity; the emulator returns the value as it is.
__device__ __forceinline__ u32 uni(u32 v) {
#if defined(JPP_EMU) && defined(JPP_EMU_CHECK_UNI)
  hip_emu::check_uniform(v, __FILE__, __LINE__);   // (test build: the claim is checked across the wavefront)
  return v;
#elif defined(JPP_EMU) || !defined(__HIP_DEVICE_COMPILE__)
  return v;
#else
  return (u32)__builtin_amdgcn_readfirstlane((int)v);
#endif
}
__device__ __forceinline__ int uni(int v) { return (int)uni((u32)v); }

// broadcast lane `src` (wave-uniform) of v to every lane: v_readlane_b32
__device__ __forceinline__ u32 wave_bcast_u32(u32 v, int src) {
#if defined(JPP_EMU)
  return wave_shfl_u32(v, src);
#else
  return (u32)__builtin_amdgcn_readlane((int)v, src);
#endif
}
__device__ __forceinline__ float wave_bcast_f32(float v, int src) {
#if defined(JPP_EMU)
  return wave_shfl_f32(v, src);
#else
  int x;
  __builtin_memcpy(&x, &v, 4);
  x = __builtin_amdgcn_readlane(x, src);
  float r;
  __builtin_memcpy(&r, &x, 4);
  return r;
#endif
}

__device__ __forceinline__ u64 mulhi_u64(u64 a, u64 b) {
#if defined(JPP_EMU)
  return (u64)(((unsigned __int128)a * b) >> 64);
#else
  return __umul64hi(a, b);
#endif
}

// x % m with magic = floor((2^64 - 1) / m): the estimate of the quotient is short by at most 2
__device__ __forceinline__ u64 fastmod_u64(u64 x, u64 m, u64 magic) {
  u64 q = mulhi_u64(x, magic);
  u64 r = x - q * m;
  while (r >= m) r -= m;
  return r;
}

__device__ __forceinline__ u64 wave_shfl_u64(u64 v, int src) {
#if defined(JPP_EMU)
  return hip_emu::shfl_u64(v, src);
#else
  u32 lo = (u32)__shfl((int)(u32)v, src, 64);
  u32 hi = (u32)__shfl((int)(u32)(v >> 32), src, 64);
  return ((u64)hi << 32) | lo;
#endif
}

// max over the whole wave (all 64 lanes must call)
__device__ __forceinline__ u64 wave_max_u64(u64 v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    u64 o = wave_shfl_u64(v, lane_id() ^ off);
    v = o > v ? o : v;
  }
  return v;
}

__device__ __forceinline__ u32 wave_sum_u32(u32 v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    v += wave_shfl_u32(v, lane_id() ^ off);
  }
  return v;
}

__device__ __forceinline__ u64 wave_sum_u64(u64 v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    v += wave_shfl_u64(v, lane_id() ^ off);
  }
  return v;
}

__device__ __forceinline__ u32 wave_max_u32(u32 v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    u32 o = wave_shfl_u32(v, lane_id() ^ off);
    v = o > v ? o : v;
  }
  return v;
}

__device__ __forceinline__ int popc64(u64 v) { return __builtin_popcountll(v); }

}  // namespace jpp

#endif  // JPP_RT_H
