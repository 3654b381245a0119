// TEST INFRASTRUCTURE ONLY.  Functional CPU emulation of the tiny subset of
// HIP that jumanpp_amd's kernels use, so that the *actual kernel sources* can
// be run against the oracle without a GPU (pytest -m "not gpu").  It is never
// compiled into the product library (libjppgpu.so); the product fails loudly
// when no HIP device is present.
//
// Model: a launch runs its blocks one after another; the threads of a block
// are ucontext fibers on one OS thread.  __syncthreads() and the wave-level
// exchange primitives are rendezvous points; a pass over all fibers without
// progress is reported as a deadlock (this catches divergent barriers).
#ifndef JPP_TESTS_HIP_EMU_H
#define JPP_TESTS_HIP_EMU_H

#include <ucontext.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
#include <vector>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __shared__ static
#define __launch_bounds__(...)
#define __restrict__

struct dim3 {
  unsigned x, y, z;
  dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};

namespace hip_emu {

constexpr int kWave = 64;

// Fibers switch with __builtin_setjmp / __builtin_longjmp: swapcontext saves and restores the signal mask, two system
// calls per switch, and a kernel like k_sweep switches at every wave_sync() of every lane.  A fiber is created once
// (makecontext, entered with setcontext) and then lives in a loop that runs the launch body once per block it is part of.
struct Fiber {
  ucontext_t ctx;
  void* jb[5];
  char* stack = nullptr;
  bool started = false;
  bool done = false;
  int wait = 0;  // 0 runnable, 1 block barrier, 2 wave barrier
};

struct State {
  dim3 grid, block;
  dim3 bidx, tidx;
  int cur = 0;
  std::vector<Fiber> fibers;
  void* sched_jb[5];
  std::function<void()> body;
  // per-wave exchange slots for shuffles / ballots
  std::vector<uint64_t> xchg;
};

inline State& st() {
  static State s;
  return s;
}

__attribute__((noinline)) inline void yield_to_sched() {
  State& s = st();
  Fiber& f = s.fibers[s.cur];
  if (__builtin_setjmp(f.jb) == 0) __builtin_longjmp(s.sched_jb, 1);
}

inline void block_barrier() {
  State& s = st();
  s.fibers[s.cur].wait = 1;
  yield_to_sched();
}

inline void wave_barrier() {
  State& s = st();
  s.fibers[s.cur].wait = 2;
  yield_to_sched();
}

inline void fiber_main() {
  for (;;) {
    st().body();
    State& s = st();
    s.fibers[s.cur].done = true;
    yield_to_sched();
  }
}

// scheduler side: run fiber f until it yields
__attribute__((noinline)) inline void resume(Fiber& f) {
  State& s = st();
  if (__builtin_setjmp(s.sched_jb) == 0) {
    if (!f.started) {
      f.started = true;
      setcontext(&f.ctx);
    } else {
      __builtin_longjmp(f.jb, 1);
    }
  }
}

// LDS is NOT preserved between workgroups and holds whatever the previous wavefronts of the CU left there.  `__shared__`
// is `static` here, which would make every never-written element a stable, small, valid-looking value (zero, or what the
// previous block wrote): a kernel that reads such an element as an index passes here and faults on the MI355X (round 4:
// the eight-keys-per-lane global beam read enn[0] of a boundary WITHOUT left nodes and loaded through it,
// DESIGN.md section 8).  The emulator's link (tests/emu/lds.ld) gathers every function-local static of the kernel
// sources into one section, and every block starts with that section poisoned: an index made of 0xA5 bytes points
// gigabytes away and the process dies on the spot (or AddressSanitizer names the load).
extern "C" char __start_jpp_lds[] __attribute__((weak));
extern "C" char __stop_jpp_lds[] __attribute__((weak));
#if defined(__SANITIZE_ADDRESS__)
// (an AddressSanitizer build puts red zones between the statics: the section is filled by a loop the sanitizer does not
// instrument -- the red zones' shadow is untouched, so out-of-bounds accesses to an LDS array are still reported)
__attribute__((no_sanitize_address, noinline)) inline void poison_range(char* a, char* b) {
  for (volatile char* p = a; p < b; ++p) *p = (char)0xA5;
}
#else
inline void poison_range(char* a, char* b) { memset(a, 0xA5, (size_t)(b - a)); }
#endif
inline void poison_lds() {
  static const bool off = std::getenv("JPP_EMU_NO_LDS_POISON") != nullptr;   // (A/B timing of the poisoning itself)
  if (off) return;
  if (__start_jpp_lds != nullptr && __stop_jpp_lds > __start_jpp_lds) poison_range(__start_jpp_lds, __stop_jpp_lds);
}
inline size_t lds_section_bytes() { return __start_jpp_lds ? (size_t)(__stop_jpp_lds - __start_jpp_lds) : 0; }

inline void run_block(unsigned nthreads) {
  State& s = st();
  poison_lds();
  constexpr size_t kStack = 256 * 1024;
  if (s.fibers.size() < nthreads) {
    // (the vector may not move fibers that are parked inside it: reserve once for the largest block there is)
    if (s.fibers.capacity() < 1024) s.fibers.reserve(1024);
    if (nthreads > 1024) {
      fprintf(stderr, "hip_emu: blocks of more than 1024 threads are not supported\n");
      abort();
    }
    s.fibers.resize(nthreads);
  }
  s.xchg.assign(nthreads, 0);
  for (unsigned t = 0; t < nthreads; ++t) {
    Fiber& f = s.fibers[t];
    if (!f.stack) {
      f.stack = static_cast<char*>(malloc(kStack));
      getcontext(&f.ctx);
      f.ctx.uc_stack.ss_sp = f.stack;
      f.ctx.uc_stack.ss_size = kStack;
      f.ctx.uc_link = nullptr;
      makecontext(&f.ctx, (void (*)())fiber_main, 0);
      f.started = false;
    }
    f.done = false;
    f.wait = 0;
  }
  for (;;) {
    bool progress = false;
    bool alldone = true;
    for (unsigned t = 0; t < nthreads; ++t) {
      Fiber& f = s.fibers[t];
      if (f.done) continue;
      alldone = false;
      if (f.wait != 0) continue;
      s.cur = (int)t;
      s.tidx = dim3(t % s.block.x, 0, 0);
      resume(f);
      progress = true;
    }
    if (alldone) break;
    // release wave barriers
    for (unsigned w0 = 0; w0 < nthreads; w0 += kWave) {
      unsigned w1 = w0 + kWave < nthreads ? w0 + kWave : nthreads;
      bool any = false, all = true;
      for (unsigned t = w0; t < w1; ++t) {
        if (s.fibers[t].done) continue;
        if (s.fibers[t].wait == 2) any = true; else all = false;
      }
      if (any && all) {
        for (unsigned t = w0; t < w1; ++t) if (!s.fibers[t].done) s.fibers[t].wait = 0;
        progress = true;
      }
    }
    // release the block barrier
    {
      bool any = false, all = true;
      for (unsigned t = 0; t < nthreads; ++t) {
        if (s.fibers[t].done) continue;
        if (s.fibers[t].wait == 1) any = true; else all = false;
      }
      if (any && all) {
        for (unsigned t = 0; t < nthreads; ++t) if (!s.fibers[t].done) s.fibers[t].wait = 0;
        progress = true;
      }
    }
    if (!progress) {
      fprintf(stderr, "hip_emu: deadlock (divergent barrier?) in block %u\n", s.bidx.x);
      abort();
    }
  }
}

// launches are serialised: the emulator has one fiber state, and the host layer may drive two
// contexts from two threads (jumanpp_gpu's alternating analyzers)
inline std::mutex& launch_mutex() {
  static std::mutex mu;
  return mu;
}

template <typename F>
void launch(dim3 grid, dim3 block, F&& f) {
  std::lock_guard<std::mutex> lock(launch_mutex());
  State& s = st();
  s.grid = grid;
  s.block = block;
  s.body = f;
  for (unsigned b = 0; b < grid.x; ++b) {
    s.bidx = dim3(b, 0, 0);
    run_block(block.x);
  }
}

}  // namespace hip_emu

#define threadIdx (hip_emu::st().tidx)
#define blockIdx (hip_emu::st().bidx)
#define blockDim (hip_emu::st().block)
#define gridDim (hip_emu::st().grid)

inline void __syncthreads() { hip_emu::block_barrier(); }

// ---- wave primitives used through jpp_rt.h wrappers -------------------------
namespace hip_emu {
inline int lane() { return st().cur % kWave; }
inline uint64_t shfl_u64(uint64_t v, int src) {
  State& s = st();
  int base = s.cur - s.cur % kWave;
  s.xchg[s.cur] = v;
  wave_barrier();
  uint64_t r = s.xchg[base + (src & (kWave - 1))];
  wave_barrier();
  return r;
}
inline uint64_t ballot(bool p) {
  State& s = st();
  int base = s.cur - s.cur % kWave;
  s.xchg[s.cur] = p ? 1 : 0;
  wave_barrier();
  uint64_t r = 0;
  for (int i = 0; i < kWave && base + i < (int)s.block.x; ++i) {
    if (!s.fibers[base + i].done && s.xchg[base + i]) r |= (uint64_t{1} << i);
  }
  wave_barrier();
  return r;
}
// -DJPP_EMU_CHECK_UNI (tools/emu_asan.sh): uni(v) = v_readfirstlane on the device is only right when v IS the same in
// every lane that executes it; the plain emulator returns v as it is and would never notice.  Every lane of the
// wavefront publishes its value and compares (a uni() under divergent control flow shows up as the emulator's deadlock).
inline void check_uniform(uint64_t v, const char* file, int line) {
  State& s = st();
  int base = s.cur - s.cur % kWave;
  s.xchg[s.cur] = v;
  wave_barrier();
  for (int i = 0; i < kWave && base + i < (int)s.block.x; ++i) {
    if (!s.fibers[base + i].done && s.xchg[base + i] != v) {
      fprintf(stderr, "hip_emu: uni() of a value that differs between lanes (%s:%d, block %u, lanes %d / %d: %llu / %llu)\n", file, line,
              s.bidx.x, s.cur % kWave, i, (unsigned long long)v, (unsigned long long)s.xchg[base + i]);
      abort();
    }
  }
  wave_barrier();
}
// element `i` of a lane's v_mfma_f32_16x16x4_f32 result: the lanes publish (a, b), then lane l forms
// D[4 * (l >> 4) + i][l & 15] as the k-ascending fused chain the matrix core computes
inline float mfma_f32_16x16x4(float a, float b, float c, int i) {
  State& s = st();
  int base = s.cur - s.cur % kWave;
  uint32_t ua, ub;
  std::memcpy(&ua, &a, 4);
  std::memcpy(&ub, &b, 4);
  s.xchg[s.cur] = (uint64_t)ua | ((uint64_t)ub << 32);
  wave_barrier();
  const int l = s.cur % kWave;
  const int m = 4 * (l >> 4) + i, n = l & 15;
  float acc = c;
  for (int k = 0; k < 4; ++k) {
    const uint32_t xa = (uint32_t)s.xchg[base + k * 16 + m];
    const uint32_t xb = (uint32_t)(s.xchg[base + k * 16 + n] >> 32);
    float fa, fb;
    std::memcpy(&fa, &xa, 4);
    std::memcpy(&fb, &xb, 4);
    acc = std::fma(fa, fb, acc);
  }
  wave_barrier();
  return acc;
}
}  // namespace hip_emu

template <typename T>
inline T atomicAdd(T* p, T v) {
  T o = *p;
  *p = o + v;
  return o;
}
template <typename T>
inline T atomicMin(T* p, T v) {
  T o = *p;
  if (v < o) *p = v;
  return o;
}
template <typename T>
inline T atomicMax(T* p, T v) {
  T o = *p;
  if (v > o) *p = v;
  return o;
}
inline unsigned atomicCAS(unsigned* p, unsigned expect, unsigned v) {
  const unsigned old = *p;
  if (old == expect) *p = v;
  return old;
}
inline unsigned atomicExch(unsigned* p, unsigned v) {
  const unsigned old = *p;
  *p = v;
  return old;
}
inline void __threadfence() {}
inline unsigned atomicOr(unsigned* p, unsigned v) {
  unsigned o = *p;
  *p = o | v;
  return o;
}
inline unsigned long long atomicOr(unsigned long long* p, unsigned long long v) {
  unsigned long long o = *p;
  *p = o | v;
  return o;
}

#endif  // JPP_TESTS_HIP_EMU_H
