// DEVELOPER BUILDS ONLY (-DJPP_DEV_PROF): per-phase cycle counters of k_sweep and k_rnn_score.
// Lane 0 of every wavefront adds the s_memtime cycles it spent per phase to g_sweep_prof[]; read with
// the debug entry point jppgpu_debug_sweep_prof (tools/gpu_sweep_phases.py).  Slots 8..13 are shared: k_sweep's finer
// marks inside its phases (JPP_PROF(8..13), read with --fine) and k_rnn_score's phases (read with --rnn; one or the other).  The shipped library is
// built without this header: the JPP_PROF* / JPP_RPROF* macros are then empty (jpp_rt.h).
#ifndef JPP_DEV_PROF_H
#define JPP_DEV_PROF_H
__device__ unsigned long long g_sweep_prof[16];
__device__ unsigned long long g_rnn_cnt[2];
#define JPP_PROF_DECL unsigned long long prof_t = __builtin_readcyclecounter(), prof_acc[14] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}
// (the scheduling barriers and the memory clobbers keep the compiler from moving the counter read across the phase's
// loads and stores; the wait makes the phase that issued them pay for its own memory operations)
#define JPP_PROF(i)                                          \
  do {                                                       \
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); \
    __builtin_amdgcn_sched_barrier(0);                       \
    unsigned long long now_ = __builtin_readcyclecounter();  \
    __builtin_amdgcn_sched_barrier(0);                       \
    asm volatile("" ::: "memory");                           \
    prof_acc[i] += now_ - prof_t;                            \
    prof_t = now_;                                           \
  } while (0)
#define JPP_PROF_FLUSH                                                                 \
  do {                                                                                 \
    if (lane == 0)                                                                     \
      for (int q_ = 0; q_ < 14; ++q_) atomicAdd(&g_sweep_prof[q_], prof_acc[q_]);      \
  } while (0)
#define JPP_RPROF_DECL unsigned long long rprof_t = __builtin_readcyclecounter(), rprof_acc[6] = {0, 0, 0, 0, 0, 0}; \
  const unsigned long long rprof_start = rprof_t; \
  unsigned long long rprof_pass = 0, rprof_nodes = 0
#define JPP_RPROF(i)                                         \
  do {                                                       \
    unsigned long long now_ = __builtin_readcyclecounter();  \
    rprof_acc[i] += now_ - rprof_t;                          \
    rprof_t = now_;                                          \
  } while (0)
#define JPP_RPROF_COUNT(cn) do { if (lane == 0) { rprof_pass += 1; rprof_nodes += (unsigned long long)(cn); } } while (0)
#define JPP_RPROF_FLUSH                                                                    \
  do {                                                                                     \
    if (lane == 0) {                                                                       \
      for (int q_ = 0; q_ < 6; ++q_) atomicAdd(&g_sweep_prof[8 + q_], rprof_acc[q_]);      \
      atomicAdd(&g_sweep_prof[14], __builtin_readcyclecounter() - rprof_start);            \
      atomicAdd(&g_sweep_prof[15], 1ull);                                                  \
      atomicAdd(&g_rnn_cnt[0], rprof_pass);                                                \
      atomicAdd(&g_rnn_cnt[1], rprof_nodes);                                               \
    }                                                                                      \
  } while (0)
// k_rnn_prep<true> (long sentences): cycles per phase, lane 0 of every wavefront (jppgpu_debug_prep_prof, tools/gpu_prep_phases.py)
__device__ unsigned long long g_prep_prof[8];
#define JPP_PPROF_DECL unsigned long long pprof_t = __builtin_readcyclecounter(), pprof_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define JPP_PPROF(i)                                         \
  do {                                                       \
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); \
    __builtin_amdgcn_sched_barrier(0);                       \
    unsigned long long now_ = __builtin_readcyclecounter();  \
    __builtin_amdgcn_sched_barrier(0);                       \
    asm volatile("" ::: "memory");                           \
    pprof_acc[i] += now_ - pprof_t;                          \
    pprof_t = now_;                                          \
  } while (0)
#define JPP_PPROF_FLUSH                                                                \
  do {                                                                                 \
    if (lane == 0)                                                                     \
      for (int q_ = 0; q_ < 8; ++q_) atomicAdd(&g_prep_prof[q_], pprof_acc[q_]);       \
  } while (0)
#endif  // JPP_DEV_PROF_H
