// C-ABI implementation (include/jppgpu.h): model upload, batch pipeline,
// result views.  Compiled with hipcc for gfx950 into libjppgpu.so.  There is
// no CPU path: without a HIP device jppgpu_ctx_create fails with
// JPPGPU_NO_DEVICE.  (tests/emu builds this same file with -DJPP_EMU against a
// fiber emulator to exercise the kernel sources in CI; that library is test
// infrastructure and is never loaded by the product.)
#include "../../include/jppgpu.h"

#include <cstdio>
#include <cstdlib>
#include <atomic>
#include <chrono>
#include <cstring>
#include <memory>
#include <mutex>
#include <set>
#include <string>
#include <thread>
#include <vector>

#include "jpp_rt.h"
#include "jpp_types.h"
#include "k_adjust.h"
#include "k_decode.h"
#include "k_format.h"
#include "k_latfmt.h"
#include "k_gold.h"
#include "k_lattice.h"
#include "k_rnn.h"
#include "k_seeds.h"
#include "k_sweep.h"
#include "k_sweep_full.h"
#include "k_t0.h"
#include "k_train.h"

using namespace jpp;

namespace {

thread_local std::string g_err;

int fail(int code, const std::string& msg) {
  g_err = msg;
  return code;
}

// HIP's current device is a property of the calling host thread.  A process may hold contexts on several
// GPUs (jumanpp_gpu --devices=...) and drive them from several threads, so every entry point that touches
// the device first binds the thread to its context's GPU.
#if defined(JPP_EMU)
inline bool bind_device(int) { return true; }
#else
inline bool bind_device(int device) { return hipSetDevice(device) == hipSuccess; }
#endif

// ---- thin device-memory layer ------------------------------------------------
#if defined(JPP_EMU)
bool rt_ok(int) { return true; }
void* rt_malloc(size_t n) {   // (hipMalloc does not zero either: reads of never-written memory should show up here too)
  void* p = malloc(n ? n : 1);
  if (p) memset(p, 0xA5, n ? n : 1);
  return p;
}
void rt_free(void* p) { free(p); }
void rt_h2d(void* d, const void* h, size_t n, jpp_stream_t) { memcpy(d, h, n); }
void rt_d2h(void* h, const void* d, size_t n, jpp_stream_t) { memcpy(h, d, n); }
void rt_sync(jpp_stream_t) {}
void rt_device_sync() {}
struct SyncPoint {
  void init() {}
  void destroy() {}
  void mark(jpp_stream_t) {}
  void wait(jpp_stream_t) {}
  void make_stream_wait(jpp_stream_t) {}
};
jpp_stream_t rt_stream_create() { return nullptr; }
void rt_stream_destroy(jpp_stream_t) {}
void* rt_host_alloc(size_t n) { return malloc(n ? n : 1); }
void* rt_host_alloc_pinned(size_t n) { return malloc(n ? n : 1); }
void rt_host_free(void* p) { free(p); }
// (emulator: "device" pointers are host pointers)
void* rt_mailbox_alloc(size_t n, void** dev) {
  void* p = calloc(1, n);
  *dev = p;
  return p;
}
void rt_mailbox_free(void* p) { free(p); }
struct Timer {
  void init() {}
  void destroy() {}
  void mark(int, jpp_stream_t) {}
  void collect(float* ms) {
    for (int i = 0; i < 12; ++i) ms[i] = 0;
  }
};
struct FmtTimer {
  void init() {}
  void destroy() {}
  void mark(int, jpp_stream_t) {}
  void collect(float* ms) { ms[0] = ms[1] = 0.f; }
};
#else
void* rt_malloc(size_t n) {
  void* p = nullptr;
  if (hipMalloc(&p, n ? n : 1) != hipSuccess) return nullptr;
  return p;
}
void rt_free(void* p) {
  if (p) (void)hipFree(p);
}
void rt_h2d(void* d, const void* h, size_t n, jpp_stream_t s) {
  (void)hipMemcpyAsync(d, h, n, hipMemcpyHostToDevice, s);
}
void rt_d2h(void* h, const void* d, size_t n, jpp_stream_t s) {
  (void)hipMemcpyAsync(h, d, n, hipMemcpyDeviceToHost, s);
}
// (Polling the stream with sched_yield() instead of sleeping in the driver was measured in round 3: it buys ~0.02 ms per
// sync when the host is idle, and it DOUBLED the batch time inside jumanpp_gpu, whose 32 format workers own the cores
// the polling thread yields to.  The driver's blocking wait it is.)
void rt_sync(jpp_stream_t s) { (void)hipStreamSynchronize(s); }
void rt_device_sync() { (void)hipDeviceSynchronize(); }
// "Everything enqueued before mark() has completed": lets the host wait for a copy while kernels enqueued behind
// the mark keep the GPU busy.
struct SyncPoint {
  hipEvent_t ev = nullptr;
  void init() { (void)hipEventCreateWithFlags(&ev, hipEventDisableTiming); }
  void destroy() {
    if (ev) (void)hipEventDestroy(ev);
    ev = nullptr;
  }
  void mark(jpp_stream_t s) { (void)hipEventRecord(ev, s); }
  void wait(jpp_stream_t s) {
    if (!ev) {
      rt_sync(s);
      return;
    }
    (void)hipEventSynchronize(ev);
  }
  // device-side: work enqueued on `s` after this call starts once everything before mark() has completed
  void make_stream_wait(jpp_stream_t s) {
    if (ev) (void)hipStreamWaitEvent(s, ev, 0);
  }
};
// a context's own stream: contexts used from different host threads do not serialise on the null stream
jpp_stream_t rt_stream_create() {
  hipStream_t s = nullptr;
  if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) return nullptr;
  return s;
}
void rt_stream_destroy(jpp_stream_t s) {
  if (s) (void)hipStreamDestroy(s);
}
// host blocks for the result copies.  Page-locking them (hipHostMalloc) was measured and rejected: the
// copies got faster but pinning tens of megabytes per batch made the analysis stage 2-3x slower and
// erratic; ordinary memory, recycled through HostPool so that it is neither re-allocated nor re-zeroed
// per batch, is what is used.
// (developer knob JPPGPU_DEV_PINNED=1: blocks of at least 1 MB page-locked -- they are recycled through HostPool, so a
// steady-state batch pins nothing; profiles/r03_u_pinned_results.txt)
std::mutex g_pinned_mu;
std::set<void*> g_pinned;
void* rt_host_alloc(size_t n) {
  static const bool pinned = std::getenv("JPPGPU_DEV_PINNED") && std::atoi(std::getenv("JPPGPU_DEV_PINNED")) != 0;
  if (pinned && n >= (size_t{1} << 20)) {
    void* p = nullptr;
    if (hipHostMalloc(&p, n, hipHostMallocDefault) == hipSuccess && p) {
      std::lock_guard<std::mutex> l(g_pinned_mu);
      g_pinned.insert(p);
      return p;
    }
  }
  return malloc(n ? n : 1);
}
// page-locked blocks for the formatted output text (150 MB per 65 536-sentence batch: pageable memory crosses PCIe at
// ~8 GB/s through the runtime's staging buffer, pinned memory at the link rate); recycled through a HostPool of their
// own, so a steady-state batch pins nothing
void* rt_host_alloc_pinned(size_t n) {
  void* p = nullptr;
  if (hipHostMalloc(&p, n ? n : 1, hipHostMallocDefault) == hipSuccess && p) {
    std::lock_guard<std::mutex> l(g_pinned_mu);
    g_pinned.insert(p);
    return p;
  }
  return malloc(n ? n : 1);
}
// a few words of page-locked host memory the device writes directly (hipHostMallocMapped): the batch totals that size
// the next allocations reach the host without a copy engine -- a 16-byte hipMemcpyAsync queues behind whatever the
// engine is doing, e.g. the 150 MB text copy of the previous batch on the context next door (3-4 ms per sync in
// jumanpp_gpu, profiles/r04_f_cli_stages.txt)
void* rt_mailbox_alloc(size_t n, void** dev) {
  void* p = nullptr;
  *dev = nullptr;
  if (hipHostMalloc(&p, n, hipHostMallocMapped) != hipSuccess || !p) return nullptr;
  if (hipHostGetDevicePointer(dev, p, 0) != hipSuccess) {
    (void)hipHostFree(p);
    return nullptr;
  }
  memset(p, 0, n);
  return p;
}
void rt_mailbox_free(void* p) {
  if (p) (void)hipHostFree(p);
}
void rt_host_free(void* p) {
  if (!p) return;
  {
    std::lock_guard<std::mutex> l(g_pinned_mu);
    auto it = g_pinned.find(p);
    if (it != g_pinned.end()) {
      g_pinned.erase(it);
      (void)hipHostFree(p);
      return;
    }
  }
  free(p);
}
struct Timer {
  hipEvent_t ev[15];
  bool have = false;
  bool chain = false;   // ev[13] .. ev[14] bracket k_rnn_chain (recorded only when it ran)
  void init() {
    for (auto& e : ev) (void)hipEventCreate(&e);
    have = true;
  }
  void destroy() {
    if (have)
      for (auto& e : ev) (void)hipEventDestroy(e);
    have = false;
  }
  void mark(int i, jpp_stream_t s) {
    // JPPGPU_DEBUG_SYNC=1: synchronise after every phase and say which one finished (fault triage)
    static const bool dbg = std::getenv("JPPGPU_DEBUG_SYNC") != nullptr;
    if (dbg) {
      hipError_t e = hipStreamSynchronize(s);
      std::fprintf(stderr, "[jppgpu] phase mark %d reached: %s\n", i, hipGetErrorString(e));
    }
    (void)hipEventRecord(ev[i], s);
    if (i == 14) chain = true;
    if (i == 0) chain = false;
  }
  void collect(float* ms) {
    // ev[0]..ev[7] bracket the seven phases
    for (int i = 0; i < 7; ++i) {
      ms[i] = 0;
      (void)hipEventElapsedTime(&ms[i], ev[i], ev[i + 1]);
    }
    ms[7] = 0;
    (void)hipEventElapsedTime(&ms[7], ev[0], ev[7]);
    // the sweep phase by class (classes 1 and 2 may run on the context's second stream, beside class 0):
    // class 0 ev[8] .. ev[11], class 1 ev[9] .. ev[10], class 2 ev[10] .. ev[12]
    static const int from[3] = {8, 9, 10}, to[3] = {11, 10, 12};
    for (int i = 0; i < 3; ++i) {
      ms[8 + i] = 0;
      (void)hipEventElapsedTime(&ms[8 + i], ev[from[i]], ev[to[i]]);
    }
    ms[11] = 0;
    if (chain) (void)hipEventElapsedTime(&ms[11], ev[13], ev[14]);
  }
};
// the format kernels of a result (jppgpu_result_format_top1 / _lattice): ev[0] .. ev[1] count pass + offset scan (the
// host then reads the byte total), ev[2] .. ev[3] write pass
struct FmtTimer {
  hipEvent_t ev[4];
  bool have = false;
  void init() {
    for (auto& e : ev) (void)hipEventCreate(&e);
    have = true;
  }
  void destroy() {
    if (have)
      for (auto& e : ev) (void)hipEventDestroy(e);
    have = false;
  }
  void mark(int i, jpp_stream_t s) {
    if (have) (void)hipEventRecord(ev[i], s);
  }
  void collect(float* ms) {   // (after the stream was synchronised)
    ms[0] = ms[1] = 0.f;
    if (!have) return;
    (void)hipEventElapsedTime(&ms[0], ev[0], ev[1]);
    (void)hipEventElapsedTime(&ms[1], ev[2], ev[3]);
  }
};
#endif

// device (re)allocations of the process: a steady pipeline makes none after its first batch (jppgpu_ctx_stats)
std::atomic<unsigned long long> g_dev_allocs{0};
struct DevBuf {
  void* p = nullptr;
  size_t cap = 0;
  bool ensure(size_t bytes) {
    if (bytes <= cap) return true;
    ++g_dev_allocs;
    rt_free(p);
    size_t want = bytes + bytes / 4 + 256;
    p = rt_malloc(want);
    cap = p ? want : 0;
    return p != nullptr;
  }
  void release() {
    rt_free(p);
    p = nullptr;
    cap = 0;
  }
  template <typename T>
  T* as() const {
    return static_cast<T*>(p);
  }
};

u64 host_varint(const u8* p, size_t& pos) {
  u64 r = 0;
  int shift = 0;
  for (;;) {
    u32 b = p[pos++];
    r |= (u64)(b & 0x7f) << shift;
    if (b < 0x80 || shift >= 63) break;
    shift += 7;
  }
  return r;
}

}  // namespace

// Host copies of a result live in blocks that a context recycles across batches (the per-batch sizes
// repeat; a fresh std::vector would be allocated and zero-filled every time).  A block goes back to its
// pool when the result is released; the pool is freed with the context.
struct HostBlock {
  void* p;
  size_t cap;
};
// page-locked blocks pinned ahead of time by jppgpu_host_prepin (any thread, before or while contexts are made): the
// text pools of the contexts take from here before they pin anything themselves
static std::mutex g_prepin_mu;
static std::vector<HostBlock> g_prepinned;

struct HostPool {
  typedef HostBlock Block;
  bool pinned = false;   // blocks come from rt_host_alloc_pinned
  // A result may be released on another thread than the one that analyses on its context (jumanpp_gpu hands the text of
  // a batch to a writer thread while the analysis and format threads keep taking blocks): the free list is locked.
  std::mutex mu;
  std::vector<Block> free_blocks;
  Block take(size_t bytes) {
    {
      std::lock_guard<std::mutex> l(mu);
      int best = -1;
      for (int i = 0; i < (int)free_blocks.size(); ++i)
        if (free_blocks[i].cap >= bytes && (best < 0 || free_blocks[i].cap < free_blocks[best].cap)) best = i;
      if (best >= 0 && free_blocks[best].cap <= 2 * bytes + 4096) {
        Block b = free_blocks[best];
        free_blocks.erase(free_blocks.begin() + best);
        return b;
      }
    }
    if (pinned) {
      std::lock_guard<std::mutex> l(g_prepin_mu);
      int best = -1;
      for (int i = 0; i < (int)g_prepinned.size(); ++i)
        if (g_prepinned[i].cap >= bytes && (best < 0 || g_prepinned[i].cap < g_prepinned[best].cap)) best = i;
      if (best >= 0) {
        Block b = g_prepinned[best];
        g_prepinned.erase(g_prepinned.begin() + best);
        return b;
      }
    }
    size_t want = bytes + bytes / 8 + 256;
    return Block{pinned ? rt_host_alloc_pinned(want) : rt_host_alloc(want), want};
  }
  void give(Block b) {
    if (!b.p) return;
    std::lock_guard<std::mutex> l(mu);
    free_blocks.push_back(b);
  }
  void clear() {
    std::vector<Block> all;
    {
      std::lock_guard<std::mutex> l(mu);
      all.swap(free_blocks);
    }
    for (auto& b : all) rt_host_free(b.p);
  }
  ~HostPool() { clear(); }
};

template <typename T>
struct HostVec {
  HostPool* pool = nullptr;
  HostPool::Block blk{nullptr, 0};
  size_t n = 0;
  HostVec() = default;
  HostVec(const HostVec&) = delete;
  HostVec& operator=(const HostVec&) = delete;
  ~HostVec() {
    if (pool) pool->give(blk);
    else rt_host_free(blk.p);
  }
  // contents are unspecified after a resize (always followed by a full copy or fill)
  bool resize(size_t count) {
    if (count * sizeof(T) > blk.cap) {
      if (pool) pool->give(blk);
      else rt_host_free(blk.p);
      blk = pool ? pool->take(count * sizeof(T)) : HostPool::Block{rt_host_alloc(count * sizeof(T)), count * sizeof(T)};
      if (!blk.p) {
        blk.cap = 0;
        n = 0;
        return false;
      }
    }
    n = count;
    return true;
  }
  void assign(size_t count, T v) {
    if (resize(count))
      for (size_t i = 0; i < count; ++i) data()[i] = v;
  }
  T* data() { return static_cast<T*>(blk.p); }
  const T* data() const { return static_cast<const T*>(blk.p); }
  T& operator[](size_t i) { return data()[i]; }
  const T& operator[](size_t i) const { return data()[i]; }
  size_t size() const { return n; }
};

struct jppgpu_result {
  // declared first = destroyed last: a result released after its context keeps the pool alive until its
  // own blocks are back in it
  std::shared_ptr<HostPool> pool_ref;
  std::shared_ptr<HostPool> text_pool_ref;
  jppgpu_ctx* ctx = nullptr;
  Batch B{};
  u64 generation = 0;
  Config cfg{};  // configuration the batch was analysed with (jppgpu_ctx_set_beams may change the context's later)
  bool fetched_basic = false, fetched_full = false, fetched_top1 = false;
  // host copies
  HostVec<i32> status;
  HostVec<u32> ncp, nnodes, path_len, path_nodes;
  HostVec<u64> node_base, bnd_base;
  HostVec<jppgpu_node> nodes;
  HostVec<jppgpu_unk> unk;
  HostVec<u32> bnd_first, bnd_cnt, end_first, end_cnt, end_nodes, ngb, gbeam;
  HostVec<i32> entry_rows;
  HostVec<u64> patterns;
  HostVec<float> t0, cells;
  HostVec<jppgpu_beam_slot> beams;
  HostVec<u8> kept;
  HostVec<u32> byte_off;
  // JPPGPU_FETCH_TOP1: compact tables of the top-1 paths
  HostVec<i32> t1_status;
  HostVec<u32> t1_ncp, t1_len, t1_idx;
  HostVec<u64> t1_base, t1_zero;
  HostVec<jppgpu_node> t1_nodes;
  HostVec<jppgpu_unk> t1_unk;
  // jppgpu_result_format_top1
  bool fm_have = false;
  int fm_kind = 0;         // 1: top-1 text (format_top1), 2: lattice text (format_lattice)
  int fm_nbest = 0;
  HostVec<u32> fm_head;    // format_lattice: header bytes per sentence
  HostVec<i32> fm_status;
  HostVec<u64> fm_off;
  HostVec<char> fm_text;   // (page-locked: text_pool_ref)
  // jppgpu_result_fetch_nbest
  int nb_n = 0;
  HostVec<i32> nb_status;
  HostVec<u32> nb_ncp, nb_nnodes;
  HostVec<u64> nb_first;
  HostVec<jppgpu_beam_slot> nb_eos;
  HostVec<jppgpu_nbest_item> nb_items;
  // jppgpu_result_fetch_top1_ngrams
  bool ng_have = false;
  HostVec<u64> ng_first;
  HostVec<u32> ng_nodes, ng_feat;
  // jppgpu_result_fetch_path_ngrams
  HostVec<u64> gp_first;
  HostVec<u32> gp_nodes, gp_feat;
  void bind(HostPool* pool);
};

// What a context holds of the MODEL in HBM: the dictionary blobs, the weight table, the RNN tables, the per-entry T0
// records and the format table -- read-only for the kernels, so every context of a device can use the same copy
// (jppgpu_ctx_create_shared).  Freed with its last context.
struct ModelBufs {
  DevBuf trie, eptrs, edata, weights, dyn_spec;
  DevBuf rnn_known, rnn_unk, rnn_wt, rnn_emb, rnn_nce, rnn_maxent;
  DevModel* dmodel = nullptr;
  // per-entry T0 memo (k_t0_memo): device table + what its weight-dependent half is rebuilt from
  DevBuf t0_memo;
  u32 t0_memo_slots = 0;
  struct MemoSeed {
    u32 slot, len;
    i32 row[spec::kNumDicFeatures];
  };
  std::vector<MemoSeed> t0_memo_seeds;
  std::vector<DevBuf> field_blobs;       // column storages of the length primitives (DevSpec::storages)
  bool t0_memo_from_image = false;       // uploaded from jppgpu_config::t0_memo_image: no seeds to rebuild it from
  std::vector<T0Memo> t0_memo_host;      // keep_t0_memo_image: what jppgpu_ctx_t0_memo_image hands out
  bool keep_memo_host = false;           // ... refilled in place by jppgpu_ctx_set_weights
  // output text on the device (jppgpu_ctx_set_format_table)
  bool fmt_have = false;
  DevBuf fmt_slots, fmt_rows, fmt_blob, fmt_table;
  // the lattice format on the device (jppgpu_ctx_set_lattice_table)
  bool lat_have = false;
  DevBuf lat_slots, lat_rows, lat_blob, lat_table;
  int device = 0;
  // jppgpu_ctx_set_weights / jppgpu_ctx_set_format_table change tables that every context of the copy reads: one writer
  // at a time, and a writer first waits for the whole device (the batches its sibling contexts have enqueued)
  std::mutex mu;
  ~ModelBufs();
};

struct jppgpu_ctx {
  Config cfg{};
  int device = 0;
  DevModel hmodel{};
  std::shared_ptr<ModelBufs> mb = std::make_shared<ModelBufs>();
  UnkRank unk_rank{};   // creation order of the UNK makers (k_ends numbers the UNK entry pointers with it)
  bool dynamic_spec = false;   // a spec other than the built-in jumandic tables: table-driven kernels
  u32 row_stride = 8;          // columns of a node's entry row in node_entry: 8, or 16 for models with more than 8 feature columns
  bool builtin_spec = false;   // the spec equals the compiled-in tables (k_path_ngrams reads them), also when dynamic_spec is forced
  DevBuf rnn_conn, rnn_id, rnn_gi, rnn_assign, rnn_prev, rnn_nid, rnn_nlen, rnn_cnt, rnn_ctx, rnn_rec, rnn_rscore, rnn_noff, rnn_rows, rnn_rowbase, rnn_ord, pack_cnt, pack_off, top1_nodes, top1_aux, nbest_cnt, nbest_off, nbest_items, nbest_eos, ng_nodes, ng_feat, gstats;
  // workspace
  DevBuf text, offs;
  DevBuf cp_code, cp_class, cp_boff, cl_nodes, pos_cnt1, pos_cntN, pos_norm, pos_cnt2, pos_ends, pos_walk, reach;
  DevBuf sent_ncp, sent_status, sent_flags, sent_nodes, sent_nodes2, node_base, node_base2;
  DevBuf path_len, bnd_meta, sweep_scratch, sent_maxr, sweep_list;
  u32 last_class_n[3] = {0, 0, 0};   // sentences per sweep class of the last batch (jppgpu_last_timings)
  DevBuf pc_nb_off, pc_nb, pc_b_off, pc_b, pc_node_off, pc_nodes, pc_tags, node_penalty;
  bool partial_pending = false;  // constraints uploaded for the next analyze call
  jppgpu_score_plugin_fn plugin_fn = nullptr;  // host plugin of the next analyze call (jppgpu_analyze_batch_plugin)
  jppgpu_connection_plugin_fn pair_fn = nullptr;   // per-connection plugin of the next analyze call (jppgpu_analyze_batch_pairs)
  DevBuf pair_penalty, pair_base;
  void* plugin_user = nullptr;
  jpp_stream_t aux_stream = nullptr;   // the sweep variants of the rare wide sentences run here, beside the main variant
  SyncPoint sweep_fork, sweep_join;
  SyncPoint front_fork, front_join;   // the normalize maker's emit passes run on aux_stream beside k_seeds<1> / <2>
  // scorers: slot 0 perceptron, slot 1 the RNN when use_rnn, then the host scorers (jppgpu_analyze_batch_scored)
  bool use_rnn = false;
  int n_host_scorers = 0;
  ScoreWeights score_weights{};
  const jppgpu_score_lattice_fn* scored_fns = nullptr;   // of the next analyze call
  void* const* scored_users = nullptr;
  DevBuf adj_stack;
  jppgpu_seed_hook_fn seed_hook = nullptr;     // gold-seed hook of the next analyze call (jppgpu_analyze_batch_seeds)
  void* seed_user = nullptr;
  DevBuf node_info2, node_aux2, gold_off, gold, gold_base;
  DevBuf full_scratch, full_locks;   // k_sweep_full's HBM slices for boundaries beyond its LDS staging
  DevBuf norm_scratch, norm_locks;   // k_norm's HBM slices for starts beyond the per-lane result / state arrays
  DevBuf bnd_first, bnd_cnt, end_first, end_cnt, bnd_ngb, bnd_gbeam;
  DevBuf node_info, node_aux, end_nodes, node_entry, node_pat, node_t0, node_beam, node_cells, node_kept,
      path_nodes;
  u64 generation = 0;
  Timer timer;
  SyncPoint rnn_sync;
  float last_ms[12] = {0};   // [11] = k_rnn_chain
  FmtTimer fmt_timer;
  float last_fmt_ms[2] = {0.f, 0.f};   // count + scan / write pass of the last jppgpu_result_format_* call
  u64 last_fmt_bytes = 0;
  u64 last_rnn_rows = 0;     // hidden-state rows of the last batch (rnn nodes + 2 per sentence)
  jpp_stream_t last_stream = nullptr;
  jpp_stream_t own_stream = nullptr;  // used by the host-buffer entry points
  std::shared_ptr<HostPool> host_pool = std::make_shared<HostPool>();
  volatile u64* mail_host = nullptr;   // [16] written by k_mail
  u64* mail_dev = nullptr;
  std::shared_ptr<HostPool> text_pool = std::make_shared<HostPool>();   // page-locked blocks (constructor sets the flag)
  // output text on the device (jppgpu_ctx_set_format_table): the table in HBM and the per-batch buffers
  DevBuf fmt_len, fmt_cnt, fmt_off, fmt_text, fmt_st;
  DevBuf lat_mask, lat_best, lat_id, lat_list, lat_marked, lat_head;   // k_latfmt.h: per-node sets of the N best paths, header bytes per sentence
  bool timing_pending = false;
  // one enqueue per batch (k_lattice.h: k_cap_guard): grids of the rare sweep classes and the scratch geometry the next
  // batch is launched with before its totals are known; statistics for the bench / tests
  u32 spec_grid1 = 64, spec_grid2 = 16, spec_maxr_cap = 0;
  DevBuf scan_ws;        // k_scan_mb: tile sums + epoch-stamped flags
  u32 scan_epoch = 0;
  u64 n_spec_batches = 0, n_spec_overflows = 0, n_exact_batches = 0, n_dev_allocs_at_last_batch = 0;
};

static_assert(sizeof(jppgpu_node) == sizeof(NodeInfo), "node layout");
static_assert(sizeof(jppgpu_unk) == sizeof(NodeAux), "unk layout");
static_assert(sizeof(jppgpu_beam_slot) == sizeof(BeamSlot), "beam layout");

void jppgpu_result::bind(HostPool* pool) {
  status.pool = pool; ncp.pool = pool; nnodes.pool = pool; path_len.pool = pool; path_nodes.pool = pool;
  node_base.pool = pool; bnd_base.pool = pool; nodes.pool = pool; unk.pool = pool;
  bnd_first.pool = pool; bnd_cnt.pool = pool; end_first.pool = pool; end_cnt.pool = pool; end_nodes.pool = pool;
  ngb.pool = pool; gbeam.pool = pool; entry_rows.pool = pool; patterns.pool = pool; t0.pool = pool; cells.pool = pool;
  beams.pool = pool; kept.pool = pool; byte_off.pool = pool;
  t1_status.pool = pool; t1_ncp.pool = pool; t1_len.pool = pool; t1_idx.pool = pool; t1_base.pool = pool;
  t1_zero.pool = pool; t1_nodes.pool = pool; t1_unk.pool = pool;
  ng_first.pool = pool; ng_nodes.pool = pool; ng_feat.pool = pool;
  gp_first.pool = pool; gp_nodes.pool = pool; gp_feat.pool = pool;
  fm_status.pool = pool; fm_off.pool = pool; fm_head.pool = pool;
  nb_status.pool = pool; nb_ncp.pool = pool; nb_nnodes.pool = pool; nb_first.pool = pool; nb_eos.pool = pool; nb_items.pool = pool;
}

namespace {
// The flattened FeaturesSpec descriptors of a model (jppgpu_model::feature_spec; i32 counts and lists, see
// include/jppgpu.h) into the device tables of the table-driven kernels.  Returns an empty string or why the spec is
// outside what those kernels hold.
std::string parse_feature_spec(const void* blob, size_t bytes, int numFeatures, DevSpec* out,
                               const jppgpu_field_storage* storages = nullptr, uint32_t numStorages = 0,
                               std::vector<int>* usedStorages = nullptr) {
  const i32* p = static_cast<const i32*>(blob);
  const size_t n = bytes / 4;
  size_t pos = 0;
  bool ok = true;
  auto rd = [&]() -> i32 {
    if (pos >= n) {
      ok = false;
      return 0;
    }
    return p[pos++];
  };
  auto rdList = [&](std::vector<i32>* v) {
    const i32 c = rd();
    v->clear();
    for (i32 i = 0; ok && i < c; ++i) v->push_back(rd());
  };
  std::memset(out, 0, sizeof(*out));
  std::vector<i32> a, b;
  const i32 nprims = rd();
  if (!ok || nprims < 0 || nprims > kDynMaxPrims) return "too many primitive features";
  out->nprims = nprims;
  for (i32 i = 0; i < nprims; ++i) {
    const i32 kind = rd();
    rdList(&a);
    if (!ok) return "truncated spec";
    // spec::PrimitiveFeatureKind: Copy 1, SingleBit 2, Provided 3, ByteLength 4, CodepointSize 5,
    // SurfaceCodepointSize 6, CodepointType 7, Codepoint 8
    if (kind < 1 || kind > 8) return "primitive feature kind " + std::to_string(kind);
    out->prims[i].kind = kind;
    out->prims[i].a = a.size() > 0 ? a[0] : 0;
    out->prims[i].b = a.size() > 1 ? a[1] : 0;
    if (kind == 4 || kind == 5) {
      // ByteLength / CodepointSize over column a (feature_impl_prim.cc:51-90): the column's value storage must be here
      if (a.size() != 1 || a[0] < 0 || a[0] >= numFeatures) return "length feature over a column outside the entry row";
      int found = -1;
      for (uint32_t q = 0; q < numStorages; ++q)
        if (storages[q].column == a[0] && (storages[q].kind == 1 || storages[q].kind == 2) && storages[q].data != nullptr) found = (int)q;
      if (found < 0)
        return "length feature over column " + std::to_string(a[0]) + " whose value storage was not given (jppgpu_config::field_storages)";
      int slot = -1;
      for (size_t q = 0; usedStorages && q < usedStorages->size(); ++q)
        if ((*usedStorages)[q] == found) slot = (int)q;
      if (slot < 0) {
        if (!usedStorages || usedStorages->size() >= (size_t)kDynMaxStorages) return "more column storages than the device tables hold";
        slot = (int)usedStorages->size();
        usedStorages->push_back(found);
      }
      out->prims[i].b = slot;
    }
    if ((kind == 1 || kind == 2) && (out->prims[i].a < 0 || out->prims[i].a >= numFeatures)) return "primitive feature reads a column outside the entry row";
    if (kind == 3 && (out->prims[i].a < 0 || out->prims[i].a > 1)) return "more than two placeholders";
  }
  const i32 ncomp = rd();
  if (!ok || ncomp < 0 || ncomp > kDynMaxComputes) return "too many computed features";
  out->ncomputes = ncomp;
  for (i32 i = 0; i < ncomp; ++i) {
    const i32 prim = rd();
    rdList(&a);
    rdList(&b);
    if (!ok) return "truncated spec";
    auto& c = out->computes[i];
    if (a.empty() && b.empty()) {   // plain primitive
      c.cond = -1;
      c.nt = 1;
      c.t[0] = prim;
      c.nf = 0;
    } else {
      if (a.size() > (size_t)kDynMaxBranch || b.size() > (size_t)kDynMaxBranch) return "computed feature with too many branch members";
      c.cond = prim;
      c.nt = (i32)a.size();
      c.nf = (i32)b.size();
      for (size_t q = 0; q < a.size(); ++q) c.t[q] = a[q];
      for (size_t q = 0; q < b.size(); ++q) c.f[q] = b[q];
    }
    if (prim < 0 || prim >= nprims) return "computed feature refers to an unknown primitive";
    for (i32 q = 0; q < c.nt; ++q)
      if (c.t[q] < 0 || c.t[q] >= nprims) return "computed feature refers to an unknown primitive";
    for (i32 q = 0; q < c.nf; ++q)
      if (c.f[q] < 0 || c.f[q] >= nprims) return "computed feature refers to an unknown primitive";
  }
  const i32 npat = rd();
  if (!ok || npat < 0 || npat > kDynMaxPatterns) return "too many pattern features";
  out->npatterns = npat;
  for (i32 i = 0; i < npat; ++i) {
    const i32 idx = rd();
    rdList(&a);
    if (!ok) return "truncated spec";
    if (idx != i) return "pattern features out of order";
    if (a.size() > (size_t)kDynMaxArgs) return "pattern feature with too many arguments";
    auto& pt = out->patterns[i];
    pt.nargs = (i32)a.size();
    pt.slot = -1;
    pt.prefix = hmix(hmix(hmix(kHashSeed0, (u64)(u32)idx), (u64)a.size()), kPatternSeed);   // feature_impl_pattern.h:28-41
    for (size_t q = 0; q < a.size(); ++q) {
      if (a[q] < 0 || a[q] >= ncomp) return "pattern feature refers to an unknown feature";
      pt.args[q] = a[q];
    }
  }
  struct Ng {
    i32 index;
    std::vector<i32> refs;
  };
  std::vector<Ng> uni, bi, tri;
  const i32 nng = rd();
  for (i32 i = 0; ok && i < nng; ++i) {
    Ng g;
    g.index = rd();
    rdList(&g.refs);
    if (!ok) break;
    for (i32 r : g.refs)
      if (r < 0 || r >= npat) return "n-gram feature refers to an unknown pattern";
    if (g.refs.size() == 1) uni.push_back(g);
    else if (g.refs.size() == 2) bi.push_back(g);
    else if (g.refs.size() == 3) tri.push_back(g);
    else return "n-gram feature of order " + std::to_string(g.refs.size());
  }
  if (!ok) return "truncated spec";
  if (uni.size() > (size_t)kDynMaxUni || bi.size() > (size_t)kDynMaxBi || tri.size() > (size_t)kDynMaxTri)
    return "more n-gram features than the sweep's lane layout holds (64 unigrams, 40 bigrams, 4 trigrams)";
  // stored patterns: those a bigram or trigram reads (the unigram-only ones are consumed where they are computed),
  // in pattern order
  for (auto& g : bi)
    for (i32 r : g.refs) out->patterns[r].slot = 0;
  for (auto& g : tri)
    for (i32 r : g.refs) out->patterns[r].slot = 0;
  i32 slot = 0;
  for (i32 i = 0; i < npat; ++i)
    if (out->patterns[i].slot == 0) out->patterns[i].slot = slot++;
  if (slot > kPat) return "more than " + std::to_string(kPat) + " patterns are read by bigram / trigram features";
  out->nstored = slot;
  out->nuni = (i32)uni.size();
  out->nbi = (i32)bi.size();
  out->ntri = (i32)tri.size();
  // hash prefixes: Hasher{}.mix(order + 2).mix(index).mix(seed) (feature_impl_ngram_partial.h:23-31,52-60,98-106)
  for (size_t i = 0; i < uni.size(); ++i) {
    out->uni[i].prefix = hmix(hmix(hmix(kHashSeed0, 3), (u64)(u32)uni[i].index), kUnigramSeed);
    out->uni[i].t0 = uni[i].refs[0];
  }
  for (size_t i = 0; i < bi.size(); ++i) {
    out->bi_prefix[i] = hmix(hmix(hmix(kHashSeed0, 4), (u64)(u32)bi[i].index), kBigramSeed);
    out->bi_t01[i] = (u8)((out->patterns[bi[i].refs[0]].slot << 4) | out->patterns[bi[i].refs[1]].slot);
  }
  for (size_t i = 0; i < tri.size(); ++i) {
    out->tri_prefix[i] = hmix(hmix(hmix(kHashSeed0, 5), (u64)(u32)tri[i].index), kTrigramSeed);
    for (int q = 0; q < 3; ++q) out->tri_t[i][q] = (u8)out->patterns[tri[i].refs[q]].slot;
  }
  return std::string();
}
}  // namespace

namespace {
// ---- per-entry T0 memo (k_t0.h: T0Memo) ------------------------------------------------------------------------------
// Every key of the double array with the codepoint length of its surface: depth-first over the units (darts-clone
// layout as in jpp_device.h: trie_step).
struct TrieKey {
  i32 value;
  u32 len;
};
unsigned memo_threads() {
  unsigned hc = std::thread::hardware_concurrency();
  return hc == 0 ? 1u : hc > 16 ? 16u : hc;
}

// the subtree below `start` (inclusive)
void trie_subtree(const u32* units, size_t nunits, u32 start_id, u32 start_len, std::vector<TrieKey>* out) {
  auto offset_of = [](u32 unit) { return (unit >> 10) << ((unit & (1u << 9)) >> 6); };
  struct Item {
    u32 id, len;
  };
  std::vector<Item> stack;
  stack.push_back(Item{start_id, start_len});
  size_t visited = 0;
  while (!stack.empty() && visited <= nunits) {
    const Item it = stack.back();
    stack.pop_back();
    ++visited;
    const u32 unit = units[it.id];
    const u32 base = it.id ^ offset_of(unit);
    if ((unit >> 8) & 1) {
      if (base < nunits) out->push_back(TrieKey{(i32)(units[base] & 0x7fffffffu), it.len});
    }
    for (u32 b = 1; b < 256; ++b) {
      const u32 child = base ^ b;
      if (child >= nunits || child == it.id) continue;
      if ((units[child] & ((1u << 31) | 0xFFu)) != b) continue;
      stack.push_back(Item{child, it.len + ((b & 0xC0u) != 0x80u ? 1u : 0u)});
    }
  }
}

void trie_keys(const u32* units, size_t nunits, std::vector<TrieKey>* out) {
  if (nunits == 0) return;
  // the first two byte levels by hand, their subtrees on worker threads
  auto offset_of = [](u32 unit) { return (unit >> 10) << ((unit & (1u << 9)) >> 6); };
  struct Start {
    u32 id, len;
  };
  std::vector<Start> level{Start{0, 0}}, starts;
  for (int depth = 0; depth < 2; ++depth) {
    std::vector<Start> next;
    for (const Start& st : level) {
      const u32 unit = units[st.id];
      const u32 base = st.id ^ offset_of(unit);
      if (((unit >> 8) & 1) && base < nunits) out->push_back(TrieKey{(i32)(units[base] & 0x7fffffffu), st.len});
      for (u32 b = 1; b < 256; ++b) {
        const u32 child = base ^ b;
        if (child >= nunits || child == st.id) continue;
        if ((units[child] & ((1u << 31) | 0xFFu)) != b) continue;
        next.push_back(Start{child, st.len + ((b & 0xC0u) != 0x80u ? 1u : 0u)});
      }
    }
    level.swap(next);
  }
  starts.swap(level);
  const unsigned nt = memo_threads();
  std::vector<std::vector<TrieKey>> parts(nt);
  std::vector<std::thread> pool;
  for (unsigned t = 0; t < nt; ++t)
    pool.emplace_back([&, t]() {
      for (size_t i = t; i < starts.size(); i += nt) trie_subtree(units, nunits, starts[i].id, starts[i].len, &parts[t]);
    });
  for (auto& th : pool) th.join();
  for (auto& pt : parts) out->insert(out->end(), pt.begin(), pt.end());
}

// the seeds of the memo: (slot, surface length, entry row) of every dictionary entry reachable through the trie
void collect_memo_seeds(const jppgpu_model* m, std::vector<ModelBufs::MemoSeed>* seeds, u32* nslots) {
  std::vector<TrieKey> keys;
  trie_keys(static_cast<const u32*>(m->trie), m->trie_bytes / 4, &keys);
  const u8* eptrs = static_cast<const u8*>(m->entry_ptrs);
  const u8* edata = static_cast<const u8*>(m->entry_data);
  const u32 slots = (u32)(m->entry_data_bytes / 8 + 1);
  std::vector<u32> seen(slots, 0);   // 0: free, otherwise 1 + index into seeds, ~0u: ambiguous
  for (const TrieKey& k : keys) {
    size_t pos = (size_t)(u32)k.value;
    if (pos >= m->entry_ptrs_bytes) continue;
    const u64 cnt = host_varint(eptrs, pos);
    i32 ptr = 0;
    for (u64 q = 0; q < cnt && pos < m->entry_ptrs_bytes; ++q) {
      ptr += (i32)host_varint(eptrs, pos);
      if (ptr < 0) break;
      const size_t at = (size_t)((u32)ptr >> 1);
      const u32 slot = (u32)(at >> 3);
      if (slot >= slots || at + 8 * 10 > m->entry_data_bytes) continue;   // (entries at the very end of the blob take the full path)
      ModelBufs::MemoSeed sd{};
      sd.slot = slot;
      sd.len = k.len;
      size_t rp = at;
      for (int f = 0; f < spec::kNumDicFeatures; ++f) sd.row[f] = (i32)host_varint(edata, rp);
      if (seen[slot] == 0) {
        seeds->push_back(sd);
        seen[slot] = (u32)seeds->size();
      } else if (seen[slot] != ~0u) {
        const ModelBufs::MemoSeed& o = (*seeds)[seen[slot] - 1];
        if (o.len != sd.len || memcmp(o.row, sd.row, sizeof(sd.row)) != 0) {   // one record cannot serve two different nodes
          (*seeds)[seen[slot] - 1].len = 0;
          seen[slot] = ~0u;
        }
      }
    }
  }
  *nslots = slots;
}

// records from the seeds and a weight table (host pointers)
void fill_memo_range(const std::vector<ModelBufs::MemoSeed>& seeds, size_t from, size_t to, const float* weights, u32 wmask, T0Memo* table) {
  for (size_t i = from; i < to; ++i) {
    const auto& sd = seeds[i];
    if (sd.len == 0) continue;
    T0Memo& r = table[sd.slot];
    u64 prim[spec::kNumPrims];
    u64 pat[spec::kNumPatterns];
    t0_entry_prims(sd.row, sd.len, prim);
    t0_pattern_hashes(prim, pat);
    float w[spec::kNumUni];
    for (int u = 0; u < spec::kNumUni; ++u)
      w[u] = weights[(u32)hmix(uni_prefix(spec::kUni[u].index), pat[spec::kUni[u].t0]) & wmask];
    bool ok = true;
    for (int j = 0; j < 4; ++j) {
      u32 bits;
      memcpy(&bits, &w[j], 4);
      ok = ok && bits != 0x80000000u;   // (0.f + -0.f differs from -0.f: the last node of a boundary starts its sums from 0.f)
    }
    if (!ok) continue;
    for (int f = 0; f < spec::kNumDicFeatures; ++f) r.row[f] = sd.row[f];
    for (int j = 0; j < 4; ++j) {
      float acc = w[j];
      for (int u = j + 4; u < kT0CtxFirst; u += 4) acc += w[u];
      r.pre[j] = acc;
    }
    for (int u = kT0CtxLast + 1; u < spec::kNumUni; ++u) r.raw[u - kT0CtxLast - 1] = w[u];
    r.len = sd.len;
  }
}

// records from the seeds and a weight table (host pointers); the weight gathers miss the host caches, hence the threads
void fill_memo(const std::vector<ModelBufs::MemoSeed>& seeds, const float* weights, u32 wmask, std::vector<T0Memo>* table) {
  const unsigned nt = memo_threads();
  std::vector<std::thread> pool;
  const size_t per = (seeds.size() + nt - 1) / nt;
  for (unsigned t = 0; t < nt; ++t) {
    const size_t a = (size_t)t * per, b = a + per < seeds.size() ? a + per : seeds.size();
    if (a >= b) break;
    pool.emplace_back([&, a, b]() { fill_memo_range(seeds, a, b, weights, wmask, table->data()); });
  }
  for (auto& th : pool) th.join();
}

bool upload_memo(jppgpu_ctx* ctx, const float* weights, bool keep_host = false) {
  // A copy made with keep_t0_memo_image keeps its host records for good: jppgpu_ctx_set_weights refills them IN PLACE
  // (same storage: a pointer jppgpu_ctx_t0_memo_image handed out stays valid until the last context is destroyed)
  ctx->mb->keep_memo_host = ctx->mb->keep_memo_host || keep_host;
  std::vector<T0Memo> local;
  std::vector<T0Memo>& table = ctx->mb->keep_memo_host ? ctx->mb->t0_memo_host : local;
  if (table.size() != (size_t)ctx->mb->t0_memo_slots) table.resize((size_t)ctx->mb->t0_memo_slots);
  memset(static_cast<void*>(table.data()), 0, table.size() * sizeof(T0Memo));
  fill_memo(ctx->mb->t0_memo_seeds, weights, ctx->hmodel.wmask, &table);
  if (!ctx->mb->t0_memo.ensure(table.size() * sizeof(T0Memo))) return false;
  rt_h2d(ctx->mb->t0_memo.p, table.data(), table.size() * sizeof(T0Memo), nullptr);
  rt_sync(nullptr);
  return true;
}
}  // namespace

ModelBufs::~ModelBufs() {
  (void)bind_device(device);
  DevBuf* bufs[] = {&trie, &eptrs, &edata, &weights, &dyn_spec, &rnn_known, &rnn_unk, &rnn_wt, &rnn_emb, &rnn_nce, &rnn_maxent,
                    &t0_memo, &fmt_slots, &fmt_rows, &fmt_blob, &fmt_table, &lat_slots, &lat_rows, &lat_blob, &lat_table};
  for (auto* b : bufs) b->release();
  for (auto& b : field_blobs) b.release();
  rt_free(dmodel);
}

extern "C" const char* jppgpu_last_error(void) { return g_err.c_str(); }

namespace {
// Beams beyond kMaxBeam.  With the global beam on, a node's beam is filled from the boundary's global beam -- at most
// global_beam candidates, the other slots are fake (makeT0Beam, score_processor.cc:426-468; the EOS beam likewise,
// remakeEosBeam :553-577) -- and nothing else reads the beam size but the partition threshold beam * 4 / 3 of that
// function, which global_beam <= 32 stays below for every beam >= 32.  A beam of 40 or 500 (`-s N` widens the beam to N,
// jumanpp_args.cc:261-264) therefore gives the lattice of beam 32 with more fake slots behind it: the device keeps 32.
// Without the global beam the beam size is the real width of makeBeams (score_processor.cc:210-245): not beyond 32.
bool beams_fit(int beam, int global_beam) {
  if (global_beam > kMaxGbeam) return false;
  return beam <= kMaxBeam || global_beam >= 1;
}
int device_beam(int beam, int global_beam) { return beam > kMaxBeam && global_beam >= 1 ? kMaxBeam : beam; }

// the caller's struct may be older (shorter) or newer (longer) than ours: read what both sides know; then the
// configuration checks of AnalyzerImpl::initScorers (analyzer_impl.cc:43-89)
int read_config(const jppgpu_config* c_in, jppgpu_config* outc) {
  jppgpu_config& c_local = *outc;
  memset(&c_local, 0, sizeof(c_local));
  {
    const uint32_t sz = c_in->struct_size;
    if (sz < JPPGPU_CONFIG_MIN_SIZE || sz % 4 != 0 || sz > 4096)
      return fail(JPPGPU_INVALID_PARAMETER, "jppgpu_config::struct_size is not a size this library knows (set it to sizeof(jppgpu_config))");
    memcpy(&c_local, c_in, sz < sizeof(c_local) ? sz : sizeof(c_local));
    const unsigned char* tail = reinterpret_cast<const unsigned char*>(c_in);
    for (uint32_t i = (uint32_t)sizeof(c_local); i < sz; ++i)
      if (tail[i] != 0) return fail(JPPGPU_NOT_IMPLEMENTED, "jppgpu_config carries a non-zero field this library does not know");
    c_local.struct_size = (uint32_t)sizeof(c_local);
  }
  const jppgpu_config* c = &c_local;
#if !defined(JPP_EMU)
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
    return fail(JPPGPU_NO_DEVICE, "jppgpu: no HIP device available (the analysis path has no CPU fallback)");
  }
  if (c->device < 0 || c->device >= ndev) return fail(JPPGPU_INVALID_PARAMETER, "bad device ordinal");
  if (hipSetDevice(c->device) != hipSuccess) return fail(JPPGPU_NO_DEVICE, "hipSetDevice failed");
#endif
  if (c->beam <= 0) return fail(JPPGPU_INVALID_PARAMETER, "AnalyzerImpl: beam size can not be zero for scoring");
  if (c->num_host_scorers < 0 || c->num_host_scorers > 2 || 1 + (c->use_rnn ? 1 : 0) + c->num_host_scorers > kMaxScorers)
    return fail(JPPGPU_INVALID_PARAMETER, "jppgpu: num_host_scorers outside 0 .. 2");
  if (c->global_beam <= 0 && (c->use_rnn || c->num_host_scorers > 0))
    return fail(JPPGPU_INVALID_STATE, "additional scorers are supported only with global beam enabled");
  if (c->global_beam > 0 && c->right_check > 0 && c->right_beam <= 0)
    return fail(JPPGPU_INVALID_PARAMETER, "right global beam size should not be zero if you enable it");
  if (c->right_check < 0) return fail(JPPGPU_INVALID_PARAMETER, "right_check < 0");
  if (!beams_fit(c->beam, c->global_beam))
    return fail(JPPGPU_NOT_IMPLEMENTED,
                "jppgpu: a global beam > 32, or a beam > 32 without a global beam, is not supported");
  return JPPGPU_OK;
}

// Config, scorer weights of a new context
void apply_config(jppgpu_ctx* ctx, const jppgpu_config* c) {
  ctx->device = c->device;
  ctx->cfg = Config{device_beam(c->beam, c->global_beam), c->global_beam > 0 ? c->global_beam : 0, c->right_check, c->right_beam,
                    c->max_input_bytes > 0 ? c->max_input_bytes : 4096, 1 + (c->use_rnn ? 1 : 0) + c->num_host_scorers,
                    (c->use_rnn || c->num_host_scorers > 0) ? c->weight_perceptron : 1.0f, c->use_rnn ? c->weight_rnn : 0.0f};
  ctx->use_rnn = c->use_rnn != 0;
  ctx->n_host_scorers = c->num_host_scorers;
  int k = 0;
  ctx->score_weights.w[k++] = ctx->cfg.w_perceptron;
  if (ctx->use_rnn) ctx->score_weights.w[k++] = ctx->cfg.w_rnn;
  for (int h = 0; h < c->num_host_scorers; ++h) ctx->score_weights.w[k++] = c->weight_host[h];
  for (; k < kMaxScorers; ++k) ctx->score_weights.w[k] = 0.f;
  if (ctx->cfg.max_input_bytes > 65535) ctx->cfg.max_input_bytes = 65535;
}

// streams, events, mailbox: what every context has of its own
void finish_context(jppgpu_ctx* ctx) {
  ctx->own_stream = rt_stream_create();
  ctx->aux_stream = rt_stream_create();
  ctx->sweep_fork.init();
  ctx->sweep_join.init();
  ctx->front_fork.init();
  ctx->front_join.init();
  ctx->fmt_timer.init();
  ctx->timer.init();
  ctx->rnn_sync.init();
  void* dev = nullptr;
  ctx->mail_host = static_cast<volatile u64*>(rt_mailbox_alloc(16 * 8, &dev));
  ctx->mail_dev = static_cast<u64*>(dev);
  ctx->text_pool->pinned = true;
  if (ctx->scan_ws.ensure(sizeof(ScanWs))) {
    const ScanWs zero{};
    rt_h2d(ctx->scan_ws.p, &zero, sizeof(zero), nullptr);
    rt_sync(nullptr);
  }
}

// exclusive scan of n u32 counts into u64 offsets (+ *base), out[n] = total: one tile per workgroup when the tiles fit
// the look-back window (k_lattice.h: k_scan_mb), the single-workgroup form otherwise
void launch_scan(jppgpu_ctx* ctx, jpp_stream_t st, const u32* in, u64* out, u32 n, const u64* base) {
  const u32 tiles = (n + 1024 * kScanPer - 1) / (1024 * kScanPer);
  if (ctx->scan_ws.p != nullptr && tiles <= kScanMbBlocks) {
    if (++ctx->scan_epoch == 0) ctx->scan_epoch = 1;
    JPP_LAUNCH(k_scan_mb, tiles ? tiles : 1, 1024, st, in, out, n, base, ctx->scan_ws.as<ScanWs>(), ctx->scan_epoch);
  } else {
    JPP_LAUNCH(k_scan, 1, 1024, st, in, out, n, base);
  }
}
}  // namespace

extern "C" int jppgpu_ctx_create(const jppgpu_model* m, const jppgpu_config* c_in, jppgpu_ctx** out) {
  if (!m || !c_in || !out) return fail(JPPGPU_INVALID_PARAMETER, "null argument");
  *out = nullptr;
  jppgpu_config c_local;
  if (int rc = read_config(c_in, &c_local)) return rc;
  const jppgpu_config* c = &c_local;
  // The reference runs its generated static feature code when the spec hash matches it and its table-driven dynamic
  // feature objects otherwise (features_api.cc:20-60); here: the compiled-in jumandic tables when the flattened
  // descriptors equal them, the table-driven kernels (k_t0_dyn, k_sweep<.., DYN>: the dynamic code's summation
  // orders) for any other spec that fits the device layout.
  const bool builtinSpec = m->num_features == spec::kNumDicFeatures && m->num_placeholders == spec::kNumPlaceholders &&
                           m->feature_spec_bytes == spec::kSpecBlobSize &&
                           memcmp(m->feature_spec, spec::kSpecBlob, spec::kSpecBlobSize) == 0;
  std::unique_ptr<DevSpec> dynSpec;
  std::vector<int> usedStorages;   // indices into c->field_storages of the storages the spec's length primitives read
  if (!builtinSpec || c->dynamic_features) {
    if (m->num_features < 1 || m->num_features > kMaxDicFeatures || m->num_placeholders < 0 ||
        m->num_placeholders > spec::kNumPlaceholders)
      return fail(JPPGPU_NOT_IMPLEMENTED, "jppgpu: entry rows of more than 16 columns (JPP_MAX_DIC_FIELDS) / more than 2 placeholders are not supported");
    if (!m->feature_spec || m->feature_spec_bytes < 16) return fail(JPPGPU_INVALID_PARAMETER, "model has no feature spec");
    dynSpec.reset(new DevSpec());
    const std::string why = parse_feature_spec(m->feature_spec, m->feature_spec_bytes, m->num_features, dynSpec.get(),
                                               c->field_storages, c->field_storages ? c->num_field_storages : 0, &usedStorages);
    if (!why.empty()) return fail(JPPGPU_NOT_IMPLEMENTED, "jppgpu: feature spec outside the table-driven kernels: " + why);
  }
  if (m->weight_exponent >= 32 || !m->weights) return fail(JPPGPU_INVALID_PARAMETER, "bad perceptron weights");
  if (m->num_unk_makers > kMaxUnkMakers - 1) return fail(JPPGPU_NOT_IMPLEMENTED, "too many UNK makers");
  if (m->trie_bytes % 4 != 0 || m->trie_bytes == 0) return fail(JPPGPU_INVALID_PARAMETER, "bad trie blob");

  auto* ctx = new jppgpu_ctx();
  apply_config(ctx, c);
  ctx->mb->device = c->device;
  DevModel& H = ctx->hmodel;
  size_t wbytes = (size_t{1} << m->weight_exponent) * sizeof(float);
  bool ok = ctx->mb->trie.ensure(m->trie_bytes) && ctx->mb->eptrs.ensure(m->entry_ptrs_bytes + 16) &&
            ctx->mb->edata.ensure(m->entry_data_bytes + 16) && ctx->mb->weights.ensure(wbytes);
  if (!ok) {
    jppgpu_ctx_destroy(ctx);
    return fail(JPPGPU_OUT_OF_MEMORY, "device allocation failed (model)");
  }
  rt_h2d(ctx->mb->trie.p, m->trie, m->trie_bytes, nullptr);
  rt_h2d(ctx->mb->eptrs.p, m->entry_ptrs, m->entry_ptrs_bytes, nullptr);
  rt_h2d(ctx->mb->edata.p, m->entry_data, m->entry_data_bytes, nullptr);
  rt_h2d(ctx->mb->weights.p, m->weights, wbytes, nullptr);
  H.spec = nullptr;
  if (dynSpec) {
    dynSpec->nstorages = (i32)usedStorages.size();
    ctx->mb->field_blobs.resize(usedStorages.size());
    for (size_t q = 0; q < usedStorages.size(); ++q) {
      const jppgpu_field_storage& fs = c->field_storages[usedStorages[q]];
      if (!ctx->mb->field_blobs[q].ensure((size_t)fs.bytes + 16)) {
        jppgpu_ctx_destroy(ctx);
        return fail(JPPGPU_OUT_OF_MEMORY, "device allocation failed (column storages)");
      }
      rt_h2d(ctx->mb->field_blobs[q].p, fs.data, (size_t)fs.bytes, nullptr);
      dynSpec->storages[q].data = ctx->mb->field_blobs[q].as<u8>();
      dynSpec->storages[q].bytes = fs.bytes;
      dynSpec->storages[q].kind = (u32)fs.kind;
      dynSpec->storages[q].align = fs.align_power;
    }
    if (!ctx->mb->dyn_spec.ensure(sizeof(DevSpec))) {
      jppgpu_ctx_destroy(ctx);
      return fail(JPPGPU_OUT_OF_MEMORY, "device allocation failed (feature tables)");
    }
    rt_h2d(ctx->mb->dyn_spec.p, dynSpec.get(), sizeof(DevSpec), nullptr);
    rt_sync(nullptr);
    H.spec = ctx->mb->dyn_spec.as<DevSpec>();
    ctx->dynamic_spec = true;
    // (k_path_ngrams addresses the stored patterns by the compiled-in numbering)
    if (builtinSpec) {
      bool same = dynSpec->nstored == spec::kNumStoredPatterns;
      for (int i = 0; i < dynSpec->npatterns; ++i)
        same = same && dynSpec->patterns[i].slot == (i < spec::kNumStoredPatterns ? i : -1);
      ctx->builtin_spec = same;
    }
  } else {
    ctx->builtin_spec = true;
  }
  H.trie = ctx->mb->trie.as<u32>();
  H.entry_ptrs = ctx->mb->eptrs.as<u8>();
  H.entry_data = ctx->mb->edata.as<u8>();
  H.weights = ctx->mb->weights.as<float>();
  H.trie_units = (u32)(m->trie_bytes / 4);
  H.entry_ptrs_bytes = (u32)m->entry_ptrs_bytes;
  H.entry_data_bytes = (u32)m->entry_data_bytes;
  H.wmask = (u32)((size_t{1} << m->weight_exponent) - 1);
  H.num_features = m->num_features;
  ctx->row_stride = m->num_features <= 8 ? 8u : (u32)kMaxDicFeatures;
  // makers: [stage-1 except normalize][stage-2][normalize]
  std::vector<UnkMaker> st1, st2, norm;
  bool seenNorm = false;
  for (int i = 0; i < m->num_unk_makers; ++i) {
    const jppgpu_unk_maker& u = m->unk_makers[i];
    UnkMaker k{};
    k.type = u.type;
    k.char_class = u.char_class;
    k.pattern_ptr = u.pattern_ptr;
    k.priority = u.priority;
    k.placeholder = u.placeholder;
    k.replace_mask = u.replace_mask;
    k.pattern_mask = ~u.replace_mask & ((1u << m->num_features) - 1);
    k.spec_index = i;
    if (u.pattern_ptr < 0 || (size_t)(u.pattern_ptr >> 1) >= m->entry_data_bytes) {
      jppgpu_ctx_destroy(ctx);
      return fail(JPPGPU_INVALID_PARAMETER, "UNK template pointer outside entry data");
    }
    size_t pos = (size_t)(u.pattern_ptr >> 1);
    for (int f = 0; f < m->num_features; ++f)
      k.tmpl[f] = (i32)host_varint(static_cast<const u8*>(m->entry_data), pos);
    if (u.type < UNK_SINGLE || u.type > UNK_NORMALIZE || u.priority < 0 || u.priority > 1) {
      jppgpu_ctx_destroy(ctx);
      return fail(JPPGPU_NOT_IMPLEMENTED, "unsupported UNK maker type/priority");
    }
    if (u.type == UNK_NORMALIZE) {
      if (u.priority != 0 || seenNorm) {
        jppgpu_ctx_destroy(ctx);
        return fail(JPPGPU_NOT_IMPLEMENTED, "normalize maker must be a single stage-1 maker");
      }
      seenNorm = true;
      norm.push_back(k);
    } else if (u.priority == 0) {
      if (seenNorm) {
        jppgpu_ctx_destroy(ctx);
        return fail(JPPGPU_NOT_IMPLEMENTED, "normalize maker must be the last stage-1 maker");
      }
      st1.push_back(k);
    } else {
      st2.push_back(k);
    }
  }
  int idx = 0;
  for (auto& k : st1) H.makers[idx++] = k;
  H.n_stage1 = idx;
  for (auto& k : st2) H.makers[idx++] = k;
  H.n_unk = idx;
  H.norm_maker = -1;
  if (!norm.empty()) {
    H.norm_maker = idx;
    H.makers[idx++] = norm[0];
  }
  for (int q = 0; q < idx; ++q) H.maker_of_spec[H.makers[q].spec_index] = q;
  {
    // the reference makes its UNK nodes maker by maker: stage 1 in spec order (the normalize maker is its last
    // member, checked above), then stage 2 (analyzer_impl.cc:100-126, unk_nodes.cc:40-52)
    u32 r = 0;
    for (auto& k : st1) ctx->unk_rank.rank[k.spec_index] = (u8)r++;
    for (auto& k : norm) ctx->unk_rank.rank[k.spec_index] = (u8)r++;
    for (auto& k : st2) ctx->unk_rank.rank[k.spec_index] = (u8)r++;
    ctx->unk_rank.n = r;
    for (int q = 0; q < idx; ++q) H.makers[q].rank = ctx->unk_rank.rank[H.makers[q].spec_index];
  }
  H.has_rnn = 0;
  if (c->use_rnn) {
    // AnalyzerImpl::initScorers: scorer count must match the weights; RNN needs the global beam
    if (!m->has_rnn) {
      jppgpu_ctx_destroy(ctx);
      return fail(JPPGPU_INVALID_PARAMETER, "use_rnn set but the model has no RNN part");
    }
    const u64 E = m->rnn_layer_size, V = m->rnn_vocab_size;
    if (E == 0 || E > (u64)kMaxRnnE || m->rnn_maxent_order > 4 ||
        m->rnn_num_fields > 8 || m->rnn_maxent_size <= V) {
      jppgpu_ctx_destroy(ctx);
      return fail(JPPGPU_NOT_IMPLEMENTED, "RNN shape outside the supported range (E<=256, maxent order<=4)");
    }
    const u64 EP = E <= 64 ? 64 : E <= 128 ? 128 : 256;
    std::vector<float> wt(EP * EP, 0.f);
    for (u64 i = 0; i < E; ++i)
      for (u64 k = 0; k < E; ++k) wt[k * EP + i] = m->rnn_matrix[i * E + k];
    bool ok2 = ctx->mb->rnn_known.ensure(m->rnn_known_index_bytes) && ctx->mb->rnn_unk.ensure(m->rnn_unk_index_bytes) &&
               ctx->mb->rnn_wt.ensure(EP * EP * 4) && ctx->mb->rnn_emb.ensure(V * E * 4) && ctx->mb->rnn_nce.ensure(V * E * 4) &&
               ctx->mb->rnn_maxent.ensure(m->rnn_maxent_size * 4);
    if (!ok2) {
      jppgpu_ctx_destroy(ctx);
      return fail(JPPGPU_OUT_OF_MEMORY, "device allocation failed (rnn)");
    }
    rt_h2d(ctx->mb->rnn_known.p, m->rnn_known_index, m->rnn_known_index_bytes, nullptr);
    rt_h2d(ctx->mb->rnn_unk.p, m->rnn_unk_index, m->rnn_unk_index_bytes, nullptr);
    rt_h2d(ctx->mb->rnn_wt.p, wt.data(), EP * EP * 4, nullptr);
    rt_h2d(ctx->mb->rnn_emb.p, m->rnn_embeddings, V * E * 4, nullptr);
    rt_h2d(ctx->mb->rnn_nce.p, m->rnn_nce_embeddings, V * E * 4, nullptr);
    rt_h2d(ctx->mb->rnn_maxent.p, m->rnn_maxent, m->rnn_maxent_size * 4, nullptr);
    rt_sync(nullptr);
    H.has_rnn = 1;
    H.rnn_known = ctx->mb->rnn_known.as<u32>();
    H.rnn_unk = ctx->mb->rnn_unk.as<u32>();
    H.rnn_wt = ctx->mb->rnn_wt.as<float>();
    H.rnn_emb = ctx->mb->rnn_emb.as<float>();
    H.rnn_nce = ctx->mb->rnn_nce.as<float>();
    H.rnn_maxent = ctx->mb->rnn_maxent.as<float>();
    H.rnn_E = (u32)E;
    H.rnn_EP = (u32)EP;
    H.rnn_order = m->rnn_maxent_order;
    H.rnn_hash_max = m->rnn_maxent_size - V;
    H.rnn_hash_magic = ~0ull / H.rnn_hash_max;
    {
      // PRIMES (src/rnn/mikolov_rnn.h:18-25); hash_i = P0*P1 + sum_{j=1..i} P[(i*P[j] + j) % 36] * (ctx_j + 1),
      // and every ctx_j is the previous word id (rnn_scorer_gbeam.cc:171-188), so the sum factors out
      static const u64 P[36] = {108641969, 116049371, 125925907, 133333309, 145678979, 175308587, 197530793, 234567803,
                                251851741, 264197411, 330864029, 399999781, 407407183, 459258997, 479012069, 545678687,
                                560493491, 607407037, 629629243, 656789717, 716048933, 718518067, 725925469, 733332871,
                                753085943, 755555077, 782715551, 790122953, 812345159, 814814293, 893826581, 923456189,
                                940740127, 953085797, 985184539, 990122807};
      H.rnn_mx_base = P[0] * P[1];
      for (u64 i = 0; i < 4; ++i) {
        u64 c = 0;
        for (u64 j = 1; j <= i; ++j) c += P[(i * P[j] + j) % 36];
        H.rnn_mx_coef[i] = c;
      }
    }
    H.rnn_nce_const = m->rnn_nce_constant;
    H.rnn_unk_id = m->rnn_unk_id;
    H.rnn_unk_const = m->rnn_unk_constant;
    H.rnn_unk_len = m->rnn_unk_length;
    H.rnn_nfields = m->rnn_num_fields;
    for (u32 f = 0; f < m->rnn_num_fields; ++f) H.rnn_fields[f] = m->rnn_fields[f];
  }
  ctx->mb->dmodel = static_cast<DevModel*>(rt_malloc(sizeof(DevModel)));
  if (!ctx->mb->dmodel) {
    jppgpu_ctx_destroy(ctx);
    return fail(JPPGPU_OUT_OF_MEMORY, "device allocation failed (model header)");
  }
  rt_h2d(ctx->mb->dmodel, &H, sizeof(DevModel), nullptr);
  rt_sync(nullptr);
  // (developer knob JPPGPU_DEV_T0_MEMO=0: k_t0 without the per-entry memo)
  static const bool devT0Memo = !(std::getenv("JPPGPU_DEV_T0_MEMO") && std::atoi(std::getenv("JPPGPU_DEV_T0_MEMO")) == 0);
  const u32 memoSlotsOfModel = (u32)(m->entry_data_bytes / 8 + 1);
  if (!ctx->dynamic_spec && devT0Memo && c->t0_memo_image != nullptr && c->t0_memo_slots == memoSlotsOfModel &&
      c->t0_memo_image_bytes == (uint64_t)memoSlotsOfModel * sizeof(T0Memo)) {
    // the records as an earlier process derived them from this model: uploaded as they are
    if (!ctx->mb->t0_memo.ensure((size_t)c->t0_memo_image_bytes)) {
      jppgpu_ctx_destroy(ctx);
      return fail(JPPGPU_OUT_OF_MEMORY, "device allocation failed (T0 memo)");
    }
    rt_h2d(ctx->mb->t0_memo.p, c->t0_memo_image, (size_t)c->t0_memo_image_bytes, nullptr);
    rt_sync(nullptr);
    ctx->mb->t0_memo_slots = memoSlotsOfModel;
    ctx->mb->t0_memo_from_image = true;
  } else if (!ctx->dynamic_spec && devT0Memo) {
    const auto t_a = std::chrono::steady_clock::now();
    collect_memo_seeds(m, &ctx->mb->t0_memo_seeds, &ctx->mb->t0_memo_slots);
    const auto t_b = std::chrono::steady_clock::now();
    if (!upload_memo(ctx, m->weights, c->keep_t0_memo_image != 0)) {
      jppgpu_ctx_destroy(ctx);
      return fail(JPPGPU_OUT_OF_MEMORY, "device allocation failed (T0 memo)");
    }
    if (std::getenv("JPPGPU_DEV_T0_MEMO")) {   // =1 / =2: report
      size_t valid = 0;
      for (const auto& sd : ctx->mb->t0_memo_seeds) valid += sd.len != 0;
      std::fprintf(stderr, "[jppgpu] T0 memo: %zu entries (%zu with a record) in %u slots, trie walk %.1f ms, records + upload %.1f ms\n",
                   ctx->mb->t0_memo_seeds.size(), valid, ctx->mb->t0_memo_slots,
                   std::chrono::duration<double, std::milli>(t_b - t_a).count(),
                   std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_b).count());
    }
  }
  finish_context(ctx);
  *out = ctx;
  return JPPGPU_OK;
}

// A second context on the device of `base` that uses base's copy of the model in HBM (dictionary, weights, RNN tables,
// per-entry T0 records, format table): only workspaces, streams and the configuration are its own.  The model tables
// live until the last context that uses them is destroyed, in any order.  jppgpu_ctx_set_weights on either context
// changes the table both read.
extern "C" int jppgpu_ctx_create_shared(jppgpu_ctx* base, const jppgpu_config* c_in, jppgpu_ctx** out) {
  if (!base || !c_in || !out) return fail(JPPGPU_INVALID_PARAMETER, "null argument");
  *out = nullptr;
  jppgpu_config c_local;
  if (int rc = read_config(c_in, &c_local)) return rc;
  const jppgpu_config* c = &c_local;
  if (c->device != base->device) return fail(JPPGPU_INVALID_PARAMETER, "jppgpu_ctx_create_shared: the model copy lives on another device");
  if (c->use_rnn && !base->hmodel.has_rnn)
    return fail(JPPGPU_INVALID_PARAMETER, "jppgpu_ctx_create_shared: use_rnn set but the shared model copy has no RNN tables");
  if (base->builtin_spec && ((c->dynamic_features != 0) != base->dynamic_spec))
    return fail(JPPGPU_INVALID_PARAMETER, "jppgpu_ctx_create_shared: dynamic_features differs from the base context");
  auto* ctx = new jppgpu_ctx();
  apply_config(ctx, c);
  ctx->mb = base->mb;
  ctx->hmodel = base->hmodel;
  ctx->unk_rank = base->unk_rank;
  ctx->dynamic_spec = base->dynamic_spec;
  ctx->builtin_spec = base->builtin_spec;
  ctx->row_stride = base->row_stride;
  finish_context(ctx);
  *out = ctx;
  return JPPGPU_OK;
}

extern "C" int jppgpu_ctx_set_beams(jppgpu_ctx* ctx, int32_t beam, int32_t global_beam, int32_t right_check,
                                    int32_t right_beam) {
  if (!ctx) return fail(JPPGPU_INVALID_PARAMETER, "null argument");
  // the same checks as at construction (AnalyzerImpl::initScorers, analyzer_impl.cc:43-89)
  if (beam <= 0) return fail(JPPGPU_INVALID_PARAMETER, "AnalyzerImpl: beam size can not be zero for scoring");
  if (global_beam <= 0 && ctx->cfg.nscorers > 1)
    return fail(JPPGPU_INVALID_STATE, "additional scorers are supported only with global beam enabled");
  if (global_beam > 0 && right_check > 0 && right_beam <= 0)
    return fail(JPPGPU_INVALID_PARAMETER, "right global beam size should not be zero if you enable it");
  if (right_check < 0) return fail(JPPGPU_INVALID_PARAMETER, "right_check < 0");
  if (!beams_fit(beam, global_beam))
    return fail(JPPGPU_NOT_IMPLEMENTED, "jppgpu: a global beam > 32, or a beam > 32 without a global beam, is not supported");
  ctx->cfg.beam = device_beam(beam, global_beam);
  ctx->cfg.gbeam = global_beam > 0 ? global_beam : 0;
  ctx->cfg.rcheck = right_check;
  ctx->cfg.rbeam = right_beam;
  return JPPGPU_OK;
}

extern "C" void jppgpu_ctx_destroy(jppgpu_ctx* ctx) {
  if (!ctx) return;
  (void)bind_device(ctx->device);
  DevBuf* bufs[] = {&ctx->text,
                    &ctx->offs,       &ctx->cp_code,   &ctx->cp_class,   &ctx->cp_boff,   &ctx->cl_nodes,
                    &ctx->pos_cnt1,   &ctx->pos_cntN,  &ctx->pos_norm,  &ctx->pos_cnt2,   &ctx->pos_ends,  &ctx->pos_walk,  &ctx->reach,     &ctx->sent_ncp,
                    &ctx->sent_status, &ctx->sent_flags, &ctx->sent_nodes, &ctx->sent_nodes2, &ctx->node_base,
                    &ctx->node_base2, &ctx->path_len,  &ctx->bnd_first,  &ctx->bnd_cnt,   &ctx->end_first,
                    &ctx->end_cnt,    &ctx->bnd_ngb,   &ctx->bnd_gbeam,  &ctx->node_info, &ctx->node_aux,
                    &ctx->end_nodes,  &ctx->node_entry, &ctx->node_pat,  &ctx->node_t0,   &ctx->node_beam,
                    &ctx->node_cells, &ctx->node_kept, &ctx->path_nodes, &ctx->rnn_conn,  &ctx->rnn_id,  &ctx->rnn_gi,
                    &ctx->rnn_assign, &ctx->rnn_prev, &ctx->rnn_nid,   &ctx->rnn_nlen,
                    &ctx->rnn_cnt,    &ctx->rnn_ctx,    &ctx->rnn_rec, &ctx->rnn_rscore, &ctx->rnn_noff, &ctx->rnn_rows, &ctx->rnn_rowbase,    &ctx->rnn_ord,    &ctx->pack_cnt,  &ctx->pack_off,  &ctx->top1_nodes, &ctx->top1_aux, &ctx->nbest_cnt, &ctx->nbest_off, &ctx->nbest_items, &ctx->nbest_eos,
                    &ctx->gstats,     &ctx->bnd_meta,  &ctx->sweep_scratch, &ctx->sent_maxr, &ctx->sweep_list,  &ctx->pc_nb_off,  &ctx->pc_nb,     &ctx->pc_b_off,
                    &ctx->pc_b,       &ctx->pc_node_off, &ctx->pc_nodes, &ctx->pc_tags,   &ctx->node_penalty,
                    &ctx->node_info2, &ctx->node_aux2, &ctx->gold_off, &ctx->gold, &ctx->gold_base, &ctx->full_scratch, &ctx->full_locks, &ctx->norm_scratch, &ctx->norm_locks,
                    &ctx->adj_stack, &ctx->pair_penalty, &ctx->pair_base, &ctx->fmt_len, &ctx->fmt_cnt, &ctx->fmt_off, &ctx->fmt_text, &ctx->fmt_st, &ctx->scan_ws,
                    &ctx->lat_mask, &ctx->lat_best, &ctx->lat_id, &ctx->lat_list, &ctx->lat_marked, &ctx->lat_head};
  for (auto* b : bufs) b->release();
  ctx->mb.reset();   // (the model tables go with their last context)
  rt_stream_destroy(ctx->own_stream);
  rt_stream_destroy(ctx->aux_stream);
  ctx->sweep_fork.destroy();
  ctx->sweep_join.destroy();
  ctx->front_fork.destroy();
  ctx->front_join.destroy();
  ctx->fmt_timer.destroy();
  rt_mailbox_free((void*)ctx->mail_host);
  ctx->host_pool->clear();
  ctx->text_pool->clear();
  ctx->timer.destroy();
  ctx->rnn_sync.destroy();
  delete ctx;
}

namespace {
// widest boundary (right nodes) the sweep variants of class 0 / class 1 take: 64 / kMaxRight staged in LDS, with
// right-check * R prescores inside 2 * the staging
void sweep_class_thresholds(const Config& cfg, u32* t0, u32* t1) {
  const u32 rc = cfg.rcheck > 0 ? (u32)cfg.rcheck : 1u;
  *t0 = rc <= 2 ? 64u : 0u;
  u32 w = (2u * (u32)kMaxRight) / rc;
  *t1 = w < (u32)kMaxRight ? w : (u32)kMaxRight;
  if (*t1 < *t0) *t1 = *t0;
}
}  // namespace

namespace {
// the workspaces whose size follows from the input alone (codepoint / boundary / sentence index spaces)
bool ensure_front(jppgpu_ctx* ctx, size_t n, size_t total_bytes) {
  const size_t cpN = total_bytes + n + 8;
  const size_t bbN = total_bytes + 4 * n + 8;
  const int G = ctx->cfg.gbeam;
  bool ok = ctx->cp_code.ensure(cpN * 4) && ctx->cp_class.ensure(cpN * 4) && ctx->cp_boff.ensure(cpN * 2) &&
            ctx->cl_nodes.ensure(cpN * sizeof(ClNodes)) && ctx->pos_cnt1.ensure(cpN * 2) &&
            ctx->pos_cntN.ensure(cpN * 2) && ctx->pos_norm.ensure(cpN * 8 * kNormCache) && ctx->pos_cnt2.ensure(cpN * 2) && ctx->pos_ends.ensure(cpN * 8) && ctx->pos_walk.ensure(cpN * sizeof(WalkCache)) && ctx->reach.ensure(cpN) &&
            ctx->sent_ncp.ensure((n + 1) * 4) && ctx->sent_status.ensure((n + 1) * 4) &&
            ctx->sent_flags.ensure((n + 1) * 4) && ctx->sent_nodes.ensure((n + 1) * 4) &&
            ctx->sent_nodes2.ensure((n + 1) * 4) && ctx->node_base.ensure((n + 2) * 8) &&
            ctx->node_base2.ensure((n + 2) * 8) && ctx->path_len.ensure((n + 1) * 4) &&
            ctx->bnd_first.ensure(bbN * 4) && ctx->bnd_cnt.ensure(bbN * 4) && ctx->end_first.ensure(bbN * 4) &&
            ctx->end_cnt.ensure(bbN * 4) && ctx->bnd_ngb.ensure(bbN * 4) && ctx->bnd_meta.ensure(bbN * sizeof(BndMeta)) &&
            ctx->bnd_gbeam.ensure(bbN * G * sizeof(GbeamEntry)) &&
            (!ctx->use_rnn || (ctx->rnn_conn.ensure(bbN * G * 4) && ctx->rnn_id.ensure(bbN * G * 4) && ctx->rnn_gi.ensure(bbN * G * 4) &&
              ctx->rnn_assign.ensure(bbN * G * 4) && ctx->rnn_prev.ensure(bbN * G * 4) &&
              ctx->rnn_nid.ensure(bbN * G * 4) &&
              ctx->rnn_nlen.ensure(bbN * G * 4) && ctx->rnn_cnt.ensure(bbN * 4) && ctx->rnn_ord.ensure((2 * n + 2 * kRnnOrderBins + 2) * 4) &&
              ctx->rnn_noff.ensure(bbN * 4) && ctx->rnn_rows.ensure(((size_t)n + 1) * 4) && ctx->rnn_rowbase.ensure(((size_t)n + 2) * 8)));
  ok = ok && ctx->gstats.ensure(64) && ctx->sent_maxr.ensure(((size_t)n + 1) * 4) && ctx->sweep_list.ensure((3 * (size_t)n + 1) * 4);
  return ok;
}
}  // namespace

namespace {
// the HBM slices (and their locks) of the normalize maker's long starts: once per context
bool ensure_norm_scratch(jppgpu_ctx* ctx, jpp_stream_t st) {
  if (ctx->norm_locks.p) return true;
  static_assert(sizeof(NormState) == 16 && sizeof(NormResult) == 8, "slice layout");
  const std::vector<u32> zeros(kNormSlotGroups * 64, 0u);
  if (!(ctx->norm_scratch.ensure((size_t)kNormSlotGroups * 64 * norm_slice_bytes()) && ctx->norm_locks.ensure(zeros.size() * 4))) return false;
  rt_h2d(ctx->norm_locks.p, zeros.data(), zeros.size() * 4, st);
  rt_sync(st);
  return true;
}
}  // namespace

// Takes every device buffer a batch of up to `max_sentences` sentences / `max_total_bytes` input bytes needs at its final
// size NOW, so that the batches themselves allocate nothing and run as one enqueue from the first one on: the
// workspaces that follow from the input size exactly, the node tables / lattice arrays / hidden-state rows from
// nodes_per_byte (0: kDefaultNodesPerByte), the output-text buffers when text_bytes_per_byte != 0.  A batch that needs
// more than was reserved still works: it is run again the sized way and the buffers grow (jppgpu_ctx_stats counts it).
extern "C" int jppgpu_ctx_reserve(jppgpu_ctx* ctx, const jppgpu_reserve* r) {
  if (!ctx || !r) return fail(JPPGPU_INVALID_PARAMETER, "null argument");
  if (r->struct_size != sizeof(jppgpu_reserve)) return fail(JPPGPU_INVALID_PARAMETER, "jppgpu_reserve::struct_size is not the size this library was built with");
  if (!bind_device(ctx->device)) return fail(JPPGPU_INVALID_STATE, "hipSetDevice failed for the context's device");
  const size_t n = r->max_sentences, bytes = (size_t)r->max_total_bytes;
  if (n == 0) return JPPGPU_OK;
  const auto t_reserve0 = std::chrono::steady_clock::now();
  constexpr double kDefaultNodesPerByte = 3.0;   // (the 10^6-row bench dictionary: 2.1; 220-codepoint sentences: 2.2)
  const double npb = r->nodes_per_byte > 0.f ? (double)r->nodes_per_byte : kDefaultNodesPerByte;
  const u64 nodes = (u64)((double)bytes * npb) + 16 * (u64)n + 8;
  const int G = ctx->cfg.gbeam > 0 ? ctx->cfg.gbeam : 1, beam = ctx->cfg.beam;
  bool ok = ensure_front(ctx, n, bytes) && ctx->text.ensure(bytes + 64) && ctx->offs.ensure((n + 1) * 4);
  if (ok && ctx->hmodel.norm_maker >= 0) ok = ensure_norm_scratch(ctx, ctx->own_stream);
  // (the node tables also hold the relocation area of the stage-2 sentences: k_layout's upper bound is twice the nodes)
  const u64 seeds = 2 * nodes + nodes / 8;
  ok = ok && ctx->node_info.ensure(seeds * sizeof(NodeInfo)) && ctx->node_aux.ensure(seeds * sizeof(NodeAux));
  ok = ok && ctx->end_nodes.ensure(nodes * 4) && ctx->node_entry.ensure(nodes * ctx->row_stride * 4) &&
       ctx->node_pat.ensure(nodes * kPat * 8) && ctx->node_t0.ensure(nodes * 4) && ctx->node_beam.ensure(nodes * beam * sizeof(BeamSlot)) &&
       ctx->node_cells.ensure(nodes * G * 4 * ctx->cfg.nscorers) && ctx->node_kept.ensure(nodes) && ctx->path_nodes.ensure(nodes * 4);
  if (ctx->use_rnn) {
    const u64 rows = nodes / 6 + 2 * (u64)n + 8;   // (an rnn node per ~8 lattice nodes on 40-codepoint sentences, ~9 on long ones)
    ok = ok && ctx->rnn_ctx.ensure(rows * (size_t)ctx->hmodel.rnn_EP * 4) && ctx->rnn_rec.ensure(rows * sizeof(RnnRec)) && ctx->rnn_rscore.ensure(rows * 4);
  }
  if (ok && ctx->cfg.gbeam != 0) {
    // 64 slices of the wide-lattice variant, 2 048 right nodes each
    const u64 rc = ctx->cfg.rcheck > 0 ? (u64)ctx->cfg.rcheck : 1;
    const u64 mr = std::max<u64>(2048, ctx->spec_maxr_cap);
    const u64 stride = (((rc + 1) * mr * 4 + mr * 2) + 63) & ~u64{63};
    ok = ctx->sweep_scratch.ensure((size_t)(stride * 64));
    if (ok) ctx->spec_maxr_cap = (u32)mr;
  }
  // packed / top-1 / text read-outs
  ok = ok && ctx->pack_cnt.ensure((n + 1) * 4) && ctx->pack_off.ensure((n + 2) * 8);
  if (ok && r->text_bytes_per_byte > 0.f) {
    const u64 tb = (u64)((double)bytes * (double)r->text_bytes_per_byte) + 64 * (u64)n + 64;
    ok = ctx->fmt_len.ensure((nodes + 1) * 4) && ctx->fmt_cnt.ensure((n + 1) * 4) && ctx->fmt_off.ensure((n + 2) * 8) &&
         ctx->fmt_st.ensure((n + 1) * 4) && ctx->fmt_text.ensure(tb + 64);
    // the lattice formatter's per-node words, when that is the text this context prints
    if (ok && ctx->mb->lat_have)
      ok = ctx->lat_mask.ensure((nodes + 1) * 8) && ctx->lat_best.ensure((nodes + 1) * 8) && ctx->lat_id.ensure((nodes + 1) * 4) &&
           ctx->lat_list.ensure((nodes + 1) * sizeof(LatRec)) && ctx->lat_marked.ensure((n + 1) * 4) && ctx->lat_head.ensure((n + 1) * 4);
    // page-locked host blocks of the text, as many as the caller keeps in flight (jumanpp_gpu: analysed, being written, next)
    std::vector<HostPool::Block> blocks;
    for (uint32_t k = 0; ok && k < r->text_host_blocks; ++k) {
      HostPool::Block b = ctx->text_pool->take((size_t)tb);
      ok = b.p != nullptr;
      blocks.push_back(b);
    }
    for (auto& b : blocks) ctx->text_pool->give(b);
  }
  if (!ok) return fail(JPPGPU_OUT_OF_MEMORY, "device allocation failed (jppgpu_ctx_reserve)");
  if (std::getenv("JPPGPU_HOST_TIMING") != nullptr)
    std::fprintf(stderr, "[jppgpu] reserve: %u sentences, %llu bytes, %llu nodes: %.1f ms\n", (unsigned)n, (unsigned long long)bytes,
                 (unsigned long long)nodes, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_reserve0).count());
  return JPPGPU_OK;
}

// `count` page-locked host blocks of `bytes` each, pinned NOW on the calling thread and left where the contexts' text
// pools find them (jppgpu_result_format_top1 copies the text of a batch into such a block).  Page-locking runs at
// ~1-2 GB/s: jumanpp_gpu calls this on a thread of its own while the model is loaded and the analyzers are made, so that
// neither the first batches (round 4: 0.2-0.9 s stalls) nor the start-up (jppgpu_ctx_reserve with text_host_blocks:
// 2.3 s for 8 x 215 MB, profiles/r05b) wait for it.  Blocks nobody took are freed at process exit.
extern "C" int jppgpu_host_prepin(int32_t device, uint64_t bytes, uint32_t count) {
  if (bytes == 0 || count == 0) return JPPGPU_OK;
  if (!bind_device(device)) return fail(JPPGPU_INVALID_STATE, "hipSetDevice failed");
  for (uint32_t k = 0; k < count; ++k) {
    void* p = rt_host_alloc_pinned((size_t)bytes);
    if (!p) return fail(JPPGPU_OUT_OF_MEMORY, "host allocation failed (jppgpu_host_prepin)");
    std::lock_guard<std::mutex> l(g_prepin_mu);
    g_prepinned.push_back(HostBlock{p, (size_t)bytes});
  }
  return JPPGPU_OK;
}

extern "C" uint64_t jppgpu_t0_memo_format(void) {
  // record size, the split of the unigram list it folds, the spec it was generated from
  // (the CONTENTS of the compiled-in spec, not only its size: a regenerated jumandic_spec.inc changes the key)
  u64 h = 0xcbf29ce484222325ull;
  for (unsigned i = 0; i < spec::kSpecBlobSize; ++i) h = (h ^ spec::kSpecBlob[i]) * 0x100000001b3ull;
  return (u64{0x54304d52} << 32) ^ ((u64)sizeof(T0Memo) << 16) ^ ((u64)kT0CtxFirst << 8) ^ (u64)kT0CtxLast ^ ((u64)spec::kSpecBlobSize << 40) ^ 2u ^ (h << 1);
}

extern "C" int jppgpu_ctx_t0_memo_image(jppgpu_ctx* ctx, const void** data, uint64_t* bytes, uint32_t* slots) {
  if (!ctx || !data || !bytes || !slots) return fail(JPPGPU_INVALID_PARAMETER, "null argument");
  *data = ctx->mb->t0_memo_host.empty() ? nullptr : ctx->mb->t0_memo_host.data();
  *bytes = (uint64_t)ctx->mb->t0_memo_host.size() * sizeof(T0Memo);
  *slots = ctx->mb->t0_memo_host.empty() ? 0u : ctx->mb->t0_memo_slots;
  return JPPGPU_OK;
}

extern "C" int jppgpu_ctx_stats(jppgpu_ctx* ctx, jppgpu_ctx_statistics* out) {
  if (!ctx || !out) return fail(JPPGPU_INVALID_PARAMETER, "null argument");
  if (out->struct_size != sizeof(jppgpu_ctx_statistics)) return fail(JPPGPU_INVALID_PARAMETER, "jppgpu_ctx_statistics::struct_size is not the size this library was built with");
  out->one_enqueue_batches = ctx->n_spec_batches;
  out->one_enqueue_overflows = ctx->n_spec_overflows;
  out->sized_batches = ctx->n_exact_batches;
  out->device_allocations = g_dev_allocs.load();
  return JPPGPU_OK;
}

extern "C" int jppgpu_analyze_batch_device(jppgpu_ctx* ctx, const void* d_utf8, const void* d_offsets, uint32_t n,
                                           uint32_t total_bytes, void* stream_, jppgpu_result** out) {
  if (!ctx || !out || (!d_utf8 && total_bytes) || !d_offsets) return fail(JPPGPU_INVALID_PARAMETER, "null argument");
  if (!bind_device(ctx->device)) return fail(JPPGPU_INVALID_STATE, "hipSetDevice failed for the context's device");
  jpp_stream_t st = static_cast<jpp_stream_t>(stream_);
  *out = nullptr;
  const size_t cpN = (size_t)total_bytes + n + 8;
  const size_t bbN = (size_t)total_bytes + 4 * (size_t)n + 8;
  (void)cpN;
  const int G = ctx->cfg.gbeam, beam = ctx->cfg.beam;
  bool ok = ensure_front(ctx, n, total_bytes);
  if (!ok) return fail(JPPGPU_OUT_OF_MEMORY, "device allocation failed (batch workspace)");

  ctx->generation++;
  jppgpu_result* Rp = new jppgpu_result();
  // (every early return below -- allocation failures, a failing host callback -- drops the result and its pooled blocks)
  std::unique_ptr<jppgpu_result, void (*)(jppgpu_result*)> result_guard(Rp, jppgpu_result_release);
  Rp->pool_ref = ctx->host_pool;
  Rp->text_pool_ref = ctx->text_pool;
  Rp->bind(ctx->host_pool.get());
  Rp->fm_text.pool = ctx->text_pool.get();
  jppgpu_result& R = *Rp;
  R.ctx = ctx;
  R.generation = ctx->generation;
  R.cfg = ctx->cfg;
  Batch& B = R.B;
  B.text = static_cast<const u8*>(d_utf8);
  B.byte_off = static_cast<const u32*>(d_offsets);
  B.n_sent = n;
  B.total_bytes = total_bytes;
  B.cp_code = ctx->cp_code.as<u32>();
  B.cp_class = ctx->cp_class.as<i32>();
  B.cp_boff = ctx->cp_boff.as<u16>();
  B.cl_nodes = ctx->cl_nodes.as<ClNodes>();
  B.pos_cnt1 = ctx->pos_cnt1.as<u16>();
  B.pos_cntN = ctx->pos_cntN.as<u16>();
  B.pos_norm = ctx->pos_norm.as<u64>();
  B.pos_cnt2 = ctx->pos_cnt2.as<u16>();
  B.pos_ends = ctx->pos_ends.as<u64>();
  B.pos_walk = ctx->pos_walk.as<WalkCache>();
  B.reach = ctx->reach.as<u8>();
  B.sent_ncp = ctx->sent_ncp.as<u32>();
  B.sent_status = ctx->sent_status.as<i32>();
  B.sent_flags = ctx->sent_flags.as<u32>();
  B.sent_nodes = ctx->sent_nodes.as<u32>();
  B.sent_nodes2 = ctx->sent_nodes2.as<u32>();
  B.node_base = ctx->node_base.as<u64>();
  B.node_base2 = ctx->node_base2.as<u64>();
  B.path_len = ctx->path_len.as<u32>();
  B.bnd_first = ctx->bnd_first.as<u32>();
  B.bnd_cnt = ctx->bnd_cnt.as<u32>();
  B.end_first = ctx->end_first.as<u32>();
  B.bnd_meta = ctx->bnd_meta.as<BndMeta>();
  B.end_cnt = ctx->end_cnt.as<u32>();
  B.bnd_ngb = ctx->bnd_ngb.as<u32>();
  B.bnd_gbeam = ctx->bnd_gbeam.as<GbeamEntry>();
  B.gstats = ctx->gstats.as<u32>();
  B.sent_maxr = ctx->sent_maxr.as<u32>();
  B.sweep_list = ctx->sweep_list.as<u32>();
  B.rnn_conn = ctx->rnn_conn.as<u32>();
  B.rnn_id = ctx->rnn_id.as<i32>();
  B.rnn_gi = ctx->rnn_gi.as<u32>();
  B.rnn_assign = ctx->rnn_assign.as<u32>();
  B.rnn_prev = ctx->rnn_prev.as<u32>();
  B.rnn_nid = ctx->rnn_nid.as<i32>();
  B.rnn_nlen = ctx->rnn_nlen.as<u32>();
  B.rnn_cnt = ctx->rnn_cnt.as<u32>();
  B.rnn_order = nullptr;
  B.rnn_hist = ctx->rnn_ord.as<u32>();
  B.rnn_offs = B.rnn_hist ? B.rnn_hist + kRnnOrderBins : nullptr;
  B.rnn_slow = B.rnn_hist ? B.rnn_hist + 2 * kRnnOrderBins : nullptr;
  B.rnn_key = B.rnn_hist ? B.rnn_hist + 2 * kRnnOrderBins + 2 : nullptr;
  B.rnn_ctx = ctx->rnn_ctx.as<float>();   // (sized after k_rnn_prep, once the number of rnn nodes is known)
  B.rnn_rec = ctx->rnn_rec.as<RnnRec>();
  B.rnn_rscore = ctx->rnn_rscore.as<float>();
  B.rnn_noff = ctx->rnn_noff.as<u32>();
  B.rnn_rows = ctx->rnn_rows.as<u32>();
  B.rnn_rowbase = ctx->rnn_rowbase.as<u64>();
  B.norm_scratch = nullptr;
  B.norm_locks = nullptr;
  B.norm_slots = 0;
  if (ctx->hmodel.norm_maker >= 0) {
    if (!ensure_norm_scratch(ctx, st)) return fail(JPPGPU_OUT_OF_MEMORY, "device allocation failed (normalize scratch)");
    B.norm_scratch = ctx->norm_scratch.as<unsigned char>();
    B.norm_locks = ctx->norm_locks.as<u32>();
    B.norm_slots = kNormSlotGroups;
  }
  if (n == 0) {
    B.total_nodes = 0;
    *out = result_guard.release();
    return JPPGPU_OK;
  }

  // The pipeline, enqueued one of two ways (k_lattice.h: k_cap_guard).  spec: against the capacity the context holds,
  // no host wait until the end, *overflowed says whether the batch fitted.  Otherwise (first batch of a context that
  // was not reserved, a batch that did not fit, the host-callback entry points, full-beam scoring): three waits, each
  // sizing the next group of buffers from the device's totals.
  auto pipeline = [&](const bool spec, bool* overflowed) -> int {
  *overflowed = false;
  const u32 sblocks = (n + 255) / 256;
  const u32 wblocks = (n + kLatWaves - 1) / kLatWaves;
  Timer& T = ctx->timer;
  T.mark(0, st);
  JPP_LAUNCH(k_decode, (n + kDecWaves - 1) / kDecWaves, 64 * kDecWaves, st, B, ctx->cfg);
  T.mark(1, st);
  // (developer knob JPPGPU_DEV_SEEDS_WAVES=6: the count / emit passes compiled for 6 instead of 8 wavefronts per SIMD)
  static const int devSeedsWaves = std::getenv("JPPGPU_DEV_SEEDS_WAVES") ? std::atoi(std::getenv("JPPGPU_DEV_SEEDS_WAVES")) : 8;
  if (devSeedsWaves == 6) JPP_LAUNCH((k_seeds<0, 6>), n, 64, st, B, (const DevModel*)ctx->mb->dmodel);
  else JPP_LAUNCH(k_seeds<0>, n, 64, st, B, (const DevModel*)ctx->mb->dmodel);
  JPP_LAUNCH(k_norm<0>, n, 64, st, B, (const DevModel*)ctx->mb->dmodel);
  JPP_LAUNCH(k_layout<1>, wblocks, 64 * kLatWaves, st, B);
  launch_scan(ctx, st, (const u32*)B.sent_nodes, B.node_base, n, (const u64*)nullptr);
  launch_scan(ctx, st, (const u32*)B.sent_nodes2, B.node_base2, n, (const u64*)nullptr);
  if (spec) {
    // the node tables the context holds against the totals, on the device (no host wait)
    const u64 seedCapHave = std::min(ctx->node_info.cap / sizeof(NodeInfo), ctx->node_aux.cap / sizeof(NodeAux));
    JPP_LAUNCH(k_cap_guard, sblocks, 256, st, B, (const u64*)(B.node_base + n), (const u64*)(B.node_base2 + n), seedCapHave);
  } else {
    u64 totals[2] = {0, 0};
    if (ctx->mail_host) {
      JPP_LAUNCH(k_mail, 1, 64, st, (const u64*)(B.node_base + n), (const u64*)(B.node_base2 + n), (const u32*)nullptr, ctx->mail_dev);
      rt_sync(st);
      totals[0] = ctx->mail_host[0];
      totals[1] = ctx->mail_host[1];
    } else {
      rt_d2h(&totals[0], B.node_base + n, 8, st);
      rt_d2h(&totals[1], B.node_base2 + n, 8, st);
      rt_sync(st);
    }
    const u64 seedCap = totals[0] + totals[1] + 8;
    if (!(ctx->node_info.ensure(seedCap * sizeof(NodeInfo)) && ctx->node_aux.ensure(seedCap * sizeof(NodeAux))))
      return fail(JPPGPU_OUT_OF_MEMORY, "device allocation failed (node table)");
  }
  B.node_info = ctx->node_info.as<NodeInfo>();
  B.node_aux = ctx->node_aux.as<NodeAux>();
  // The two emit passes of a stage write disjoint slots of the node tables (the normalize maker's nodes follow the
  // others of their start) and read only what the count passes left: they run side by side, k_norm on the context's
  // second stream -- both are chains of dependent loads, and each fills the other's tail (JPPGPU_DEV_FRONT_SERIAL=1: one
  // after the other, as until round 6).
  static const bool frontSerial = std::getenv("JPPGPU_DEV_FRONT_SERIAL") != nullptr;
  const bool frontSplit = !frontSerial && ctx->aux_stream != nullptr && ctx->hmodel.norm_maker >= 0;
  auto emit_pair = [&](int mode) {
    jpp_stream_t s2 = st;
    if (frontSplit) {
      ctx->front_fork.mark(st);
      ctx->front_fork.make_stream_wait(ctx->aux_stream);
      s2 = ctx->aux_stream;
    }
    if (mode == 1) {
      JPP_LAUNCH(k_norm<1>, n, 64, s2, B, (const DevModel*)ctx->mb->dmodel);
      if (devSeedsWaves == 6) JPP_LAUNCH((k_seeds<1, 6>), n, 64, st, B, (const DevModel*)ctx->mb->dmodel);
      else JPP_LAUNCH(k_seeds<1>, n, 64, st, B, (const DevModel*)ctx->mb->dmodel);
    } else {
      JPP_LAUNCH(k_norm<2>, n, 64, s2, B, (const DevModel*)ctx->mb->dmodel);
      JPP_LAUNCH(k_seeds<2>, n, 64, st, B, (const DevModel*)ctx->mb->dmodel);
    }
    if (frontSplit) {
      ctx->front_join.mark(s2);
      ctx->front_join.make_stream_wait(st);
    }
  };
  emit_pair(1);
  JPP_LAUNCH(k_connect<1>, wblocks, 64 * kLatWaves, st, B);
  // stage 2 for disconnected sentences: relocate them behind the stage-1 region
  JPP_LAUNCH(k_layout<2>, wblocks, 64 * kLatWaves, st, B);
  {
    u32 t0c, t1c;
    sweep_class_thresholds(ctx->cfg, &t0c, &t1c);
    JPP_LAUNCH(k_sweep_classify, sblocks, 256, st, B, t0c, t1c);
  }
  launch_scan(ctx, st, (const u32*)B.sent_nodes2, B.node_base2, n, (const u64*)(B.node_base + n));
  JPP_LAUNCH(k_relocate, sblocks, 256, st, B);
  emit_pair(2);
  JPP_LAUNCH(k_connect<2>, wblocks, 64 * kLatWaves, st, B);
  u64 totalNodes = 0;
  u32 gstats[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const u64 rcS = ctx->cfg.rcheck > 0 ? (u64)ctx->cfg.rcheck : 1;
  auto scratch_stride = [&](u64 maxr) { return (((rcS + 1) * maxr * 4 + maxr * 2) + 63) & ~u64{63}; };
  u32 specSlots = 0;   // scratch slices the class-2 variant may use (spec)
  if (spec) {
    // what the lattice arrays hold, in nodes, under this batch's beam configuration
    u64 latCap = ctx->end_nodes.cap / 4;
    latCap = std::min<u64>(latCap, ctx->node_entry.cap / ((size_t)ctx->row_stride * 4));
    latCap = std::min<u64>(latCap, ctx->node_pat.cap / (kPat * 8));
    latCap = std::min<u64>(latCap, ctx->node_t0.cap / 4);
    latCap = std::min<u64>(latCap, ctx->node_beam.cap / ((size_t)beam * sizeof(BeamSlot)));
    latCap = std::min<u64>(latCap, ctx->node_cells.cap / ((size_t)(G > 0 ? G : 1) * 4 * ctx->cfg.nscorers));
    latCap = std::min<u64>(latCap, ctx->node_kept.cap);
    latCap = std::min<u64>(latCap, ctx->path_nodes.cap / 4);
    if (ctx->spec_maxr_cap != 0) {
      const u64 slots = ctx->sweep_scratch.cap / scratch_stride(ctx->spec_maxr_cap);
      specSlots = (u32)std::min<u64>(slots, 0x7fffffffu);
    }
    JPP_LAUNCH(k_cls_guard, 1, 64, st, B, std::min(ctx->spec_grid1, n), std::min(std::min(ctx->spec_grid2, specSlots), n), ctx->spec_maxr_cap, specSlots);
    JPP_LAUNCH(k_cap_guard, sblocks, 256, st, B, (const u64*)(B.node_base2 + n), (const u64*)nullptr, latCap);
  } else if (ctx->mail_host) {
    JPP_LAUNCH(k_mail, 1, 64, st, (const u64*)(B.node_base2 + n), (const u64*)nullptr, (const u32*)B.gstats, ctx->mail_dev);
    rt_sync(st);
    totalNodes = ctx->mail_host[0];
    for (int q = 0; q < 8; ++q) gstats[q] = (u32)ctx->mail_host[2 + q];
  } else {
    rt_d2h(&totalNodes, B.node_base2 + n, 8, st);
    rt_d2h(gstats, B.gstats, sizeof(gstats), st);
    rt_sync(st);
  }
  B.gold_off = nullptr;
  B.gold = nullptr;
  bool goldInserted = false;
  if (ctx->seed_hook) {
    // the trainer's gold nodes (jppgpu_analyze_batch_seeds): the hook sees the seeds of the batch and returns the ones to add
    jppgpu_seed_hook_fn hook = ctx->seed_hook;
    ctx->seed_hook = nullptr;
    const size_t N = (size_t)totalNodes;
    std::vector<i32> h_status(n);
    std::vector<u32> h_ncp(n), h_nn(n), h_ns(n);
    std::vector<u64> h_base(n + 1), h_sbase(n);
    std::vector<NodeInfo> h_nodes(N);
    std::vector<NodeAux> h_aux(N);
    rt_d2h(h_status.data(), B.sent_status, n * 4, st);
    rt_d2h(h_ncp.data(), B.sent_ncp, n * 4, st);
    rt_d2h(h_nn.data(), B.sent_nodes, n * 4, st);
    rt_d2h(h_base.data(), B.node_base, (n + 1) * 8, st);
    if (N) {
      rt_d2h(h_nodes.data(), B.node_info, N * sizeof(NodeInfo), st);
      rt_d2h(h_aux.data(), B.node_aux, N * sizeof(NodeAux), st);
    }
    rt_sync(st);
    auto live = [&](u32 q) { return h_status[q] == ST_OK || h_status[q] == ST_NO_LATTICE; };
    for (u32 q = 0; q < n; ++q) {
      h_ns[q] = live(q) && h_nn[q] >= 3 ? h_nn[q] - 3 : 0;   // without the two BOS nodes and EOS
      h_sbase[q] = h_base[q] + 2;
    }
    static_assert(sizeof(jppgpu_node) == sizeof(NodeInfo) && sizeof(jppgpu_unk) == sizeof(NodeAux), "ABI node records");
    static_assert(sizeof(jppgpu_extra_seed) == sizeof(ExtraSeed), "ABI extra seed record");
    jppgpu_seed_view view{};
    view.n_sentences = n;
    view.status = h_status.data();
    view.n_codepoints = h_ncp.data();
    view.n_seeds = h_ns.data();
    view.seed_base = h_sbase.data();
    view.seeds = reinterpret_cast<const jppgpu_node*>(h_nodes.data());
    view.unk = reinterpret_cast<const jppgpu_unk*>(h_aux.data());
    jppgpu_extra_seeds extra{nullptr, nullptr};
    if (hook(ctx->seed_user, &view, &extra) != 0) return fail(JPPGPU_INVALID_STATE, "the seed hook reported an error");
    const u64 totalExtra = extra.offsets ? extra.offsets[n] : 0;
    if (totalExtra) {
      if (!extra.seeds || extra.offsets[0] != 0) return fail(JPPGPU_INVALID_PARAMETER, "extra seeds: bad offsets");
      std::vector<u64> nbase(n + 1);
      u64 acc = 0;
      for (u32 q = 0; q < n; ++q) {
        const u32 a = extra.offsets[q], b = extra.offsets[q + 1];
        if (b < a || b - a > 0xffffu) return fail(JPPGPU_INVALID_PARAMETER, "extra seeds: bad offsets");
        if (b > a && !live(q)) return fail(JPPGPU_INVALID_PARAMETER, "extra seeds for a sentence that has no seed table");
        for (u32 k = a; k < b; ++k) {
          const jppgpu_extra_seed& e = extra.seeds[k];
          if (!(e.start < e.end && e.end <= h_ncp[q]) || (k > a && extra.seeds[k - 1].start > e.start))
            return fail(JPPGPU_INVALID_PARAMETER, "extra seeds: span outside the sentence or starts not ascending");
        }
        nbase[q] = acc;
        acc += h_nn[q] + (b - a);
      }
      nbase[n] = acc;
      if (!(ctx->node_info2.ensure((acc + 8) * sizeof(NodeInfo)) && ctx->node_aux2.ensure((acc + 8) * sizeof(NodeAux)) &&
            ctx->gold_off.ensure(((size_t)n + 1) * 4) && ctx->gold.ensure((size_t)totalExtra * sizeof(ExtraSeed)) &&
            ctx->gold_base.ensure(((size_t)n + 1) * 8)))
        return fail(JPPGPU_OUT_OF_MEMORY, "device allocation failed (gold seeds)");
      rt_h2d(ctx->gold_off.p, extra.offsets, ((size_t)n + 1) * 4, st);
      rt_h2d(ctx->gold.p, extra.seeds, (size_t)totalExtra * sizeof(ExtraSeed), st);
      rt_h2d(ctx->gold_base.p, nbase.data(), ((size_t)n + 1) * 8, st);
      B.gold_off = ctx->gold_off.as<u32>();
      B.gold = ctx->gold.as<ExtraSeed>();
      JPP_LAUNCH(k_gold_insert, wblocks, 64 * kLatWaves, st, B, (const u64*)ctx->gold_base.as<u64>(), ctx->node_info2.as<NodeInfo>(),
                 ctx->node_aux2.as<NodeAux>(), ctx->unk_rank.n);
      JPP_LAUNCH(k_gold_bounds, wblocks, 64 * kLatWaves, st, B, (const u64*)ctx->gold_base.as<u64>(),
                 (const NodeInfo*)ctx->node_info2.as<NodeInfo>());
      std::swap(ctx->node_info, ctx->node_info2);
      std::swap(ctx->node_aux, ctx->node_aux2);
      B.node_info = ctx->node_info.as<NodeInfo>();
      B.node_aux = ctx->node_aux.as<NodeAux>();
      {
        u32 t0c, t1c;
        sweep_class_thresholds(ctx->cfg, &t0c, &t1c);
        JPP_LAUNCH(k_sweep_classify, sblocks, 256, st, B, t0c, t1c);
      }
      rt_d2h(gstats, B.gstats, sizeof(gstats), st);
      rt_sync(st);   // (also: the hook's arrays are no longer read after this point)
      totalNodes = acc;
      goldInserted = true;
    }
  }
  const u32 maxR = gstats[0];
  B.total_nodes = totalNodes;   // (spec: known at the end of the batch)
  if (!spec) {
    const u64 cap = totalNodes + 8;
    ok = ctx->end_nodes.ensure(cap * 4) && ctx->node_entry.ensure(cap * ctx->row_stride * 4) &&
         ctx->node_pat.ensure(cap * kPat * 8) && ctx->node_t0.ensure(cap * 4) &&
         ctx->node_beam.ensure(cap * beam * sizeof(BeamSlot)) && ctx->node_cells.ensure(cap * G * 4 * ctx->cfg.nscorers) &&
         ctx->node_kept.ensure(cap) && ctx->path_nodes.ensure(cap * 4);
    if (!ok) return fail(JPPGPU_OUT_OF_MEMORY, "device allocation failed (lattice)");
  }
  B.end_nodes = ctx->end_nodes.as<u32>();
  B.node_entry = ctx->node_entry.as<i32>();
  B.row_stride = ctx->row_stride;
  B.node_pat = ctx->node_pat.as<u64>();
  B.node_t0 = ctx->node_t0.as<float>();
  B.node_beam = ctx->node_beam.as<BeamSlot>();
  B.node_cells = ctx->node_cells.as<float>();
  B.node_kept = ctx->node_kept.as<u8>();
  B.path_nodes = ctx->path_nodes.as<u32>();
  T.mark(2, st);
  {
    UnkRank rk = ctx->unk_rank;
    if (goldInserted) rk.n += 1;   // the gold nodes are created after every maker's (Trainer::prepare)
    JPP_LAUNCH(k_ends, wblocks, 64 * kLatWaves, st, B, ctx->cfg, rk);
  }
  T.mark(3, st);
  if (ctx->dynamic_spec) JPP_LAUNCH(k_t0_dyn, n, 64, st, B, (const DevModel*)ctx->mb->dmodel);
  else if (ctx->mb->t0_memo_slots && ctx->hmodel.wmask <= 0xffffffu)
    JPP_LAUNCH(k_t0_memo<true>, n, 64, st, B, (const DevModel*)ctx->mb->dmodel, (const T0Memo*)ctx->mb->t0_memo.as<T0Memo>(), ctx->mb->t0_memo_slots);
  else if (ctx->mb->t0_memo_slots)
    JPP_LAUNCH(k_t0_memo<false>, n, 64, st, B, (const DevModel*)ctx->mb->dmodel, (const T0Memo*)ctx->mb->t0_memo.as<T0Memo>(), ctx->mb->t0_memo_slots);
  else if (ctx->hmodel.wmask <= 0xffffffu) JPP_LAUNCH(k_t0<true>, n, 64, st, B, (const DevModel*)ctx->mb->dmodel);
  else JPP_LAUNCH(k_t0<false>, n, 64, st, B, (const DevModel*)ctx->mb->dmodel);
  B.node_penalty = nullptr;
  B.pair_penalty = nullptr;
  B.pair_base = nullptr;
  if (ctx->pair_fn) {
    // the per-connection ScorePlugin, batched: the plugin sees nodes, right-node ranges and ends lists of every boundary
    // and fills an L x R matrix of amounts per boundary (k_ends, launched above, wrote the ends lists)
    jppgpu_connection_plugin_fn fn = ctx->pair_fn;
    ctx->pair_fn = nullptr;
    const size_t N = (size_t)B.total_nodes;
    const size_t NB = bbN;
    std::vector<i32> h_status(n);
    std::vector<u32> h_ncp(n), h_nn(n), h_off(n + 1);
    std::vector<u64> h_base(n + 1), h_bbase(n);
    std::vector<NodeInfo> h_nodes(N);
    std::vector<NodeAux> h_aux(N);
    std::vector<i32> h_rows(N * ctx->row_stride);
    std::vector<u32> h_bf(NB), h_bc(NB), h_ef(NB), h_ec(NB), h_en(N);
    rt_d2h(h_status.data(), B.sent_status, n * 4, st);
    rt_d2h(h_ncp.data(), B.sent_ncp, n * 4, st);
    rt_d2h(h_nn.data(), B.sent_nodes, n * 4, st);
    rt_d2h(h_base.data(), B.node_base, (n + 1) * 8, st);
    rt_d2h(h_off.data(), B.byte_off, (n + 1) * 4, st);
    rt_d2h(h_bf.data(), B.bnd_first, NB * 4, st);
    rt_d2h(h_bc.data(), B.bnd_cnt, NB * 4, st);
    rt_d2h(h_ef.data(), B.end_first, NB * 4, st);
    rt_d2h(h_ec.data(), B.end_cnt, NB * 4, st);
    if (N) {
      rt_d2h(h_nodes.data(), B.node_info, N * sizeof(NodeInfo), st);
      rt_d2h(h_aux.data(), B.node_aux, N * sizeof(NodeAux), st);
      rt_d2h(h_rows.data(), B.node_entry, N * ctx->row_stride * 4, st);
      rt_d2h(h_en.data(), B.end_nodes, N * 4, st);
    }
    rt_sync(st);
    std::vector<u64> h_pair(NB + 1, 0);
    u64 acc = 0;
    for (u32 q = 0; q < n; ++q) {
      h_bbase[q] = (u64)h_off[q] + 4ull * q;
      if (h_status[q] != ST_OK) h_nn[q] = 0;
    }
    {
      // only boundaries of live sentences carry a matrix; the rest of the boundary index space stays empty
      std::vector<u8> liveB(NB, 0);
      for (u32 q = 0; q < n; ++q)
        if (h_status[q] == ST_OK)
          for (u32 b = 0; b < h_ncp[q] + 3 && h_bbase[q] + b < NB; ++b) liveB[h_bbase[q] + b] = 1;
      for (size_t bb = 0; bb < NB; ++bb) {
        h_pair[bb] = acc;
        if (liveB[bb]) acc += (u64)h_bc[bb] * h_ec[bb];
      }
      h_pair[NB] = acc;
    }
    jppgpu_lattice_pairs view{};
    view.n_sentences = n;
    view.num_features = (int32_t)ctx->row_stride;
    view.status = h_status.data();
    view.n_codepoints = h_ncp.data();
    view.n_nodes = h_nn.data();
    view.node_base = h_base.data();
    view.total_nodes = N;
    view.nodes = reinterpret_cast<const jppgpu_node*>(h_nodes.data());
    view.unk = reinterpret_cast<const jppgpu_unk*>(h_aux.data());
    view.entry_rows = h_rows.data();
    view.bnd_base = h_bbase.data();
    view.total_boundaries = NB;
    view.bnd_first = h_bf.data();
    view.bnd_count = h_bc.data();
    view.end_first = h_ef.data();
    view.end_count = h_ec.data();
    view.end_nodes = h_en.data();
    view.pair_base = h_pair.data();
    view.total_pairs = acc;
    std::vector<float> pen((size_t)acc, 0.f);
    fn(ctx->plugin_user, &view, pen.data());
    if (!(ctx->pair_penalty.ensure((size_t)acc * 4 + 4) && ctx->pair_base.ensure((NB + 1) * 8)))
      return fail(JPPGPU_OUT_OF_MEMORY, "device allocation failed (connection plugin)");
    if (acc) rt_h2d(ctx->pair_penalty.p, pen.data(), (size_t)acc * 4, st);
    rt_h2d(ctx->pair_base.p, h_pair.data(), (NB + 1) * 8, st);
    rt_sync(st);
    B.pair_penalty = ctx->pair_penalty.as<float>();
    B.pair_base = ctx->pair_base.as<u64>();
  }
  if (ctx->partial_pending) {
    ctx->partial_pending = false;
    if (!ctx->node_penalty.ensure((size_t)B.total_nodes * 4)) return fail(JPPGPU_OUT_OF_MEMORY, "device allocation failed (partial)");
    B.pc_nb_off = ctx->pc_nb_off.as<u32>();
    B.pc_nb = ctx->pc_nb.as<u16>();
    B.pc_b_off = ctx->pc_b_off.as<u32>();
    B.pc_b = ctx->pc_b.as<u16>();
    B.pc_node_off = ctx->pc_node_off.as<u32>();
    B.pc_nodes = ctx->pc_nodes.as<PcNode>();
    B.pc_tags = ctx->pc_tags.as<PcTag>();
    B.node_penalty = ctx->node_penalty.as<float>();
    JPP_LAUNCH(k_penalty, n, 64, st, B);
  }
  if (ctx->plugin_fn) {
    // the batched ScorePlugin hook: the plugin sees the built lattice on the host and returns per-node penalties
    jppgpu_score_plugin_fn fn = ctx->plugin_fn;
    ctx->plugin_fn = nullptr;
    const size_t N = (size_t)B.total_nodes;
    std::vector<i32> h_status(n);
    std::vector<u32> h_ncp(n), h_nn(n);
    std::vector<u64> h_base(n + 1);
    std::vector<NodeInfo> h_nodes(N);
    std::vector<NodeAux> h_aux(N);
    std::vector<i32> h_rows(N * ctx->row_stride);
    rt_d2h(h_status.data(), B.sent_status, n * 4, st);
    rt_d2h(h_ncp.data(), B.sent_ncp, n * 4, st);
    rt_d2h(h_nn.data(), B.sent_nodes, n * 4, st);
    rt_d2h(h_base.data(), B.node_base, (n + 1) * 8, st);
    if (N) {
      rt_d2h(h_nodes.data(), B.node_info, N * sizeof(NodeInfo), st);
      rt_d2h(h_aux.data(), B.node_aux, N * sizeof(NodeAux), st);
      rt_d2h(h_rows.data(), B.node_entry, N * ctx->row_stride * 4, st);
    }
    rt_sync(st);
    for (u32 q = 0; q < n; ++q)
      if (h_status[q] != ST_OK) h_nn[q] = 0;
    static_assert(sizeof(jppgpu_node) == sizeof(NodeInfo) && sizeof(jppgpu_unk) == sizeof(NodeAux), "ABI node records");
    jppgpu_lattice_nodes view{};
    view.n_sentences = n;
    view.num_features = (int32_t)ctx->row_stride;
    view.status = h_status.data();
    view.n_codepoints = h_ncp.data();
    view.n_nodes = h_nn.data();
    view.node_base = h_base.data();
    view.total_nodes = N;
    view.nodes = reinterpret_cast<const jppgpu_node*>(h_nodes.data());
    view.unk = reinterpret_cast<const jppgpu_unk*>(h_aux.data());
    view.entry_rows = h_rows.data();
    std::vector<float> pen(N, 0.f);
    fn(ctx->plugin_user, &view, pen.data());
    if (!ctx->node_penalty.ensure(N * 4 + 4)) return fail(JPPGPU_OUT_OF_MEMORY, "device allocation failed (plugin)");
    if (N) rt_h2d(ctx->node_penalty.p, pen.data(), N * 4, st);
    rt_sync(st);
    B.node_penalty = ctx->node_penalty.as<float>();
  }
  T.mark(4, st);
  // Every sentence runs the sweep variant of its own widest boundary (k_sweep_classify; the class thresholds are
  // sweep_class_thresholds()).  The <8, *> variants have no makeT0Beam replay (util::partition / introsort): they
  // take the configurations whose beams are plain stable ranks, i.e. at most 8 candidates and global beam <= beam*4/3.
  const bool narrow = ctx->cfg.gbeam <= 8 && ctx->cfg.beam <= 8 && ctx->cfg.gbeam <= ctx->cfg.beam * 4 / 3;
  // (spec: the GRIDS of the classes -- every sentence for class 0, what earlier batches suggest for the rare wide
  // classes; k_sweep leaves a workgroup beyond its class's list at once, k_cls_guard has checked that the lists fit)
  const u32 nCls[3] = {spec ? n : gstats[1], spec ? std::min(ctx->spec_grid1, n) : gstats[2],
                       spec ? std::min(std::min(ctx->spec_grid2, specSlots), n) : gstats[3]};
  const u32* lists[3] = {B.sweep_list, B.sweep_list + n, B.sweep_list + 2 * (size_t)n};
  B.sweep_scratch = nullptr;
  B.sweep_scratch_stride = 0;
  B.sweep_scratch_maxr = 0;
  if (spec) {
    if (nCls[2] != 0) {
      B.sweep_scratch = ctx->sweep_scratch.as<unsigned char>();
      B.sweep_scratch_stride = scratch_stride(ctx->spec_maxr_cap);
      B.sweep_scratch_maxr = ctx->spec_maxr_cap;
    }
  } else if (ctx->cfg.gbeam != 0 && nCls[2] != 0) {
    // class 2: the per-right-node arrays (prescores, their sums, cutoff order) in an HBM slice per workgroup
    // (kept at least as wide / as many as a later one-enqueue batch is promised: spec_maxr_cap)
    const u64 mr = std::max<u64>(maxR, ctx->spec_maxr_cap);
    const u64 stride = scratch_stride(mr);
    if (!ctx->sweep_scratch.ensure((size_t)(stride * std::max<u64>(nCls[2], 16))))
      return fail(JPPGPU_OUT_OF_MEMORY, "device allocation failed (wide-lattice scratch)");
    ctx->spec_maxr_cap = (u32)mr;
    B.sweep_scratch = ctx->sweep_scratch.as<unsigned char>();
    B.sweep_scratch_stride = stride;
    B.sweep_scratch_maxr = (u32)mr;
  }
  const DevModel* dmS = (const DevModel*)ctx->mb->dmodel;
  B.full_scratch = nullptr;
  B.full_locks = nullptr;
  B.full_slots = 0;
  B.full_cap = 0;
  if (ctx->cfg.gbeam == 0) {
    if (!ctx->full_locks.p) {
      const std::vector<u32> zeros(kFullSlots, 0u);
      if (!(ctx->full_scratch.ensure((size_t)kFullSlots * full_slot_bytes(kFullSlotCand)) && ctx->full_locks.ensure(kFullSlots * 4)))
        return fail(JPPGPU_OUT_OF_MEMORY, "device allocation failed (full-beam scratch)");
      rt_h2d(ctx->full_locks.p, zeros.data(), kFullSlots * 4, st);
      rt_sync(st);
    }
    B.full_scratch = ctx->full_scratch.as<unsigned char>();
    B.full_locks = ctx->full_locks.as<u32>();
    B.full_slots = kFullSlots;
    B.full_cap = kFullSlotCand;
  }
  if (ctx->cfg.gbeam == 0) {
    if (ctx->dynamic_spec) JPP_LAUNCH(k_sweep_full<true>, n, 64, st, B, dmS, ctx->cfg);
    else JPP_LAUNCH(k_sweep_full<false>, n, 64, st, B, dmS, ctx->cfg);
    T.mark(8, st); T.mark(11, st); T.mark(9, st); T.mark(10, st); T.mark(12, st);
  } else {
    const bool def = narrow && ctx->cfg.beam == 5 && ctx->cfg.gbeam == 6 && ctx->cfg.rcheck == 1 && ctx->cfg.rbeam == 5;
    // (developer knob: JPPGPU_DEV_SWEEP_LDS_PAD=bytes of dynamic LDS added to the launch, i.e. fewer wavefronts per
    // CU -- the occupancy curve of profiles/r03_a_occupancy.txt)
    static const unsigned devPad = std::getenv("JPPGPU_DEV_SWEEP_LDS_PAD") ? (unsigned)std::atoi(std::getenv("JPPGPU_DEV_SWEEP_LDS_PAD")) : 0u;
    // The wider classes hold few sentences on ordinary text, and a kernel of a handful of wavefronts lasts as long as its
    // longest sentence (one 300-node-per-boundary sentence: 1.9 ms): they run on the context's second stream, beside
    // the main class instead of behind it (profiles/r03_r_bench.json realism.homographs: 10.4 + 1.9 ms in sequence).
    jpp_stream_t s12 = st;
    if (nCls[0] != 0 && (nCls[1] != 0 || nCls[2] != 0) && ctx->aux_stream) {
      ctx->sweep_fork.mark(st);
      ctx->sweep_fork.make_stream_wait(ctx->aux_stream);
      s12 = ctx->aux_stream;
    }
    T.mark(8, st);
    if (ctx->dynamic_spec) {
      // table-driven variants (a spec other than the built-in jumandic tables)
      if (nCls[0]) {
        if (narrow) JPP_LAUNCH((k_sweep<8, 64, false, false, kSweepWaves, 0, true>), nCls[0], 64, st, B, dmS, ctx->cfg, lists[0]);
        else JPP_LAUNCH((k_sweep<32, 64, false, false, kSweepWideWaves, 0, true>), nCls[0], 64, st, B, dmS, ctx->cfg, lists[0]);
      }
      T.mark(11, st);
      T.mark(9, s12);
      if (nCls[1]) {
        if (narrow) JPP_LAUNCH((k_sweep<8, kMaxRight, false, false, kSweepWaves, 0, true>), nCls[1], 64, s12, B, dmS, ctx->cfg, lists[1]);
        else JPP_LAUNCH((k_sweep<32, kMaxRight, false, false, kSweepWideWaves, 0, true>), nCls[1], 64, s12, B, dmS, ctx->cfg, lists[1]);
      }
      T.mark(10, s12);
      if (nCls[2]) {
        if (narrow) JPP_LAUNCH((k_sweep<8, 0, false, false, kSweepWaves, 0, true>), nCls[2], 64, s12, B, dmS, ctx->cfg, lists[2]);
        else JPP_LAUNCH((k_sweep<32, 0, false, false, kSweepWideWaves, 0, true>), nCls[2], 64, s12, B, dmS, ctx->cfg, lists[2]);
      }
    } else {
    if (nCls[0]) {   // at most 64 right nodes per boundary, right-check <= 2
      // the CLI defaults: the one-row lean LDS layout (6.6 KB), 5 wavefronts per SIMD (96 VGPRs, no scratch); weight
      // tables of up to 2^24 entries get the 24-bit index arithmetic.  (developer knob JPPGPU_DEV_SWEEP_WAVES: 0 =
      // the round-2 layout (9.8 KB LDS, 4 waves), 4 / 5 = lean with two row buffers (7.7 KB) compiled for 4 / 5
      // waves, 15 / 6 = one row buffer compiled for 5 / 6 waves; profiles/r03_f_sweep_layouts.txt)
      static const int devWaves = std::getenv("JPPGPU_DEV_SWEEP_WAVES") ? std::atoi(std::getenv("JPPGPU_DEV_SWEEP_WAVES")) : 15;
      const bool w24 = ctx->hmodel.wmask <= 0xffffffu;
      if (def && w24 && devWaves == 4) JPP_LAUNCH_LDS((k_sweep<8, 64, true, true, 4, 1>), nCls[0], 64, devPad, st, B, dmS, ctx->cfg, lists[0]);
      else if (def && w24 && devWaves == 6) JPP_LAUNCH_LDS((k_sweep<8, 64, true, true, 6, 2>), nCls[0], 64, devPad, st, B, dmS, ctx->cfg, lists[0]);
      else if (def && w24 && devWaves == 15) JPP_LAUNCH_LDS((k_sweep<8, 64, true, true, 5, 2>), nCls[0], 64, devPad, st, B, dmS, ctx->cfg, lists[0]);
      else if (def && w24 && devWaves == 0) JPP_LAUNCH_LDS((k_sweep<8, 64, true, true>), nCls[0], 64, devPad, st, B, dmS, ctx->cfg, lists[0]);
      else if (def && w24 && devWaves == 5) JPP_LAUNCH_LDS((k_sweep<8, 64, true, true, 5, 1>), nCls[0], 64, devPad, st, B, dmS, ctx->cfg, lists[0]);
      else if (def && w24) JPP_LAUNCH_LDS((k_sweep<8, 64, true, true, 5, 2>), nCls[0], 64, devPad, st, B, dmS, ctx->cfg, lists[0]);
      else if (def) JPP_LAUNCH((k_sweep<8, 64, true, false, 5, 2>), nCls[0], 64, st, B, dmS, ctx->cfg, lists[0]);
      else if (narrow) JPP_LAUNCH((k_sweep<8, 64>), nCls[0], 64, st, B, dmS, ctx->cfg, lists[0]);
      else JPP_LAUNCH((k_sweep<32, 64, false, false, kSweepWideWaves>), nCls[0], 64, st, B, dmS, ctx->cfg, lists[0]);   // 6 KB less LDS per wavefront than the 512-wide staging
    }
    T.mark(11, st);
    T.mark(9, s12);
    if (nCls[1]) {   // at most kMaxRight right nodes per boundary (and right-check * R prescores within the staging)
      if (narrow) JPP_LAUNCH((k_sweep<8, kMaxRight>), nCls[1], 64, s12, B, dmS, ctx->cfg, lists[1]);
      else JPP_LAUNCH((k_sweep<32, kMaxRight, false, false, kSweepWideWaves>), nCls[1], 64, s12, B, dmS, ctx->cfg, lists[1]);
    }
    T.mark(10, s12);
    if (nCls[2]) {   // any width
      if (narrow) JPP_LAUNCH((k_sweep<8, 0>), nCls[2], 64, s12, B, dmS, ctx->cfg, lists[2]);
      else JPP_LAUNCH((k_sweep<32, 0, false, false, kSweepWideWaves>), nCls[2], 64, s12, B, dmS, ctx->cfg, lists[2]);
    }
    }
    T.mark(12, s12);
    if (s12 != st) {   // the main stream continues when both are done
      ctx->sweep_join.mark(s12);
      ctx->sweep_join.make_stream_wait(st);
    }
  }
  if (!spec) { ctx->last_class_n[0] = nCls[0]; ctx->last_class_n[1] = nCls[1]; ctx->last_class_n[2] = nCls[2]; }
  T.mark(5, st);
  if (ctx->use_rnn) {
    JPP_LAUNCH(k_rnn_paths, (u32)(((u64)n * ctx->cfg.gbeam + 255) / 256), 256, st, B, ctx->cfg);
    JPP_LAUNCH(k_rnn_prep, (n + kRnnPrepWaves - 1) / kRnnPrepWaves, 64 * kRnnPrepWaves, st, B,
               (const DevModel*)ctx->mb->dmodel, ctx->cfg);
    // SORT: remakeEosBeam needs the makeT0Beam replay (more than 16 EOS candidates or global beam > beam*4/3)
    const bool sortE = ctx->cfg.gbeam > 16 || ctx->cfg.gbeam > ctx->cfg.beam * 4 / 3;
    const DevModel* dm = (const DevModel*)ctx->mb->dmodel;
    // hidden states: one row of EP floats per rnn node (+ parking and BOS rows per sentence), i.e. ~31 rows per
    // 40-codepoint sentence instead of the (codepoints + 3) * G = 258 of a boundary-indexed table (1.0 GB instead of
    // 8.6 GB per 65 536 sentences).  The row total is only known here: a third host sync; the sentence ordering of
    // the lock-step recurrence (which does not need the table) is enqueued before the host waits.
    launch_scan(ctx, st, (const u32*)B.rnn_rows, B.rnn_rowbase, n, (const u64*)nullptr);
    u64 rnnRows = 0;
    if (spec) {
      const u64 rowsCap = std::min<u64>(ctx->rnn_ctx.cap / ((size_t)ctx->hmodel.rnn_EP * 4),
                                        std::min<u64>(ctx->rnn_rec.cap / sizeof(RnnRec), ctx->rnn_rscore.cap / 4));
      JPP_LAUNCH(k_cap_guard, sblocks, 256, st, B, (const u64*)(B.rnn_rowbase + n), (const u64*)nullptr, rowsCap);
    } else {
      if (ctx->mail_host) JPP_LAUNCH(k_mail, 1, 64, st, (const u64*)(B.rnn_rowbase + n), (const u64*)nullptr, (const u32*)nullptr, ctx->mail_dev + 12);
      else rt_d2h(&rnnRows, B.rnn_rowbase + n, 8, st);
      ctx->rnn_sync.mark(st);
    }
    if (ctx->hmodel.rnn_EP <= 128) {
      // lock-step workgroups take sentences of equal chain length
      B.rnn_order = B.rnn_key + n;
      // zeroed every batch (2 us): inferring "already zero" from the buffer address is wrong when a grown
      // buffer comes back at the same address
      JPP_LAUNCH(k_rnn_order_zero, 1, kRnnOrderBins, st, B.rnn_hist);
      JPP_LAUNCH(k_rnn_order_key, (n + 255) / 256, 256, st, B, ctx->cfg);
      JPP_LAUNCH(k_rnn_order_scan, 1, kRnnOrderBins, st, B);
      JPP_LAUNCH(k_rnn_order_fill, (n + 255) / 256, 256, st, B);
    }
    if (!spec) {
      ctx->rnn_sync.wait(st);
      if (ctx->mail_host) rnnRows = ctx->mail_host[12];
      ctx->last_rnn_rows = rnnRows;
      if (!(ctx->rnn_ctx.ensure((rnnRows + 8) * (size_t)ctx->hmodel.rnn_EP * 4) && ctx->rnn_rec.ensure((rnnRows + 8) * sizeof(RnnRec)) &&
            ctx->rnn_rscore.ensure((rnnRows + 8) * 4)))
        return fail(JPPGPU_OUT_OF_MEMORY, "device allocation failed (RNN hidden states)");
    }
    B.rnn_ctx = ctx->rnn_ctx.as<float>();
    B.rnn_rec = ctx->rnn_rec.as<RnnRec>();
    B.rnn_rscore = ctx->rnn_rscore.as<float>();
    // the rnn nodes in row order: what the recurrence and the scoring of long sentences read (E <= 128)
    if (ctx->hmodel.rnn_EP <= 128)
      JPP_LAUNCH(k_rnn_dense, (n + kRnnPrepWaves - 1) / kRnnPrepWaves, 64 * kRnnPrepWaves, st, B, ctx->cfg);
    const u32 slowGrid = (n + 3) / 4 < 4096u ? (n + 3) / 4 : 4096u;   // k_rnn_score_long (4 sentences per workgroup) loops over its list
    if (ctx->hmodel.rnn_EP == 64) {
      T.mark(13, st);
      JPP_LAUNCH((k_rnn_chain<1>), (n + 31) / 32, 1024, st, B, dm, ctx->cfg);
      T.mark(14, st);
      if (sortE) JPP_LAUNCH((k_rnn_score<1, true, 2>), (n + 3) / 4, 256, st, B, dm, ctx->cfg);
      else JPP_LAUNCH((k_rnn_score<1, false, 2>), (n + 3) / 4, 256, st, B, dm, ctx->cfg);
      if (sortE) JPP_LAUNCH((k_rnn_score_long<1, true>), slowGrid, 256, st, B, dm, ctx->cfg);
      else JPP_LAUNCH((k_rnn_score_long<1, false>), slowGrid, 256, st, B, dm, ctx->cfg);
    } else if (ctx->hmodel.rnn_EP == 128) {
      T.mark(13, st);
      JPP_LAUNCH((k_rnn_chain<2>), (n + 31) / 32, 1024, st, B, dm, ctx->cfg);
      T.mark(14, st);
      if (sortE) JPP_LAUNCH((k_rnn_score<2, true, 2>), (n + 3) / 4, 256, st, B, dm, ctx->cfg);
      else JPP_LAUNCH((k_rnn_score<2, false, 2>), (n + 3) / 4, 256, st, B, dm, ctx->cfg);
      if (sortE) JPP_LAUNCH((k_rnn_score_long<2, true>), slowGrid, 256, st, B, dm, ctx->cfg);
      else JPP_LAUNCH((k_rnn_score_long<2, false>), slowGrid, 256, st, B, dm, ctx->cfg);
    } else {
      if (sortE) JPP_LAUNCH((k_rnn_score<4, true, 0>), (n + 3) / 4, 256, st, B, dm, ctx->cfg);
      else JPP_LAUNCH((k_rnn_score<4, false, 0>), (n + 3) / 4, 256, st, B, dm, ctx->cfg);
    }
  }
  T.mark(6, st);
  JPP_LAUNCH(k_path, sblocks, 256, st, B, ctx->cfg);
  T.mark(7, st);
  ctx->last_stream = st;
  ctx->timing_pending = true;
  if (spec) {
    // the batch's ONE host wait: totals, class sizes and the overflow verdict in one mapped-memory record
    JPP_LAUNCH(k_mail_all, 1, 64, st, (const u64*)(B.node_base + n), (const u64*)(B.node_base2 + n), (const u32*)B.gstats,
               ctx->use_rnn ? (const u64*)(B.rnn_rowbase + n) : (const u64*)nullptr, ctx->mail_dev);
    rt_sync(st);
    volatile u64* mh = ctx->mail_host;
    B.total_nodes = mh[1];
    ctx->last_class_n[0] = (u32)mh[3]; ctx->last_class_n[1] = (u32)mh[4]; ctx->last_class_n[2] = (u32)mh[5];
    ctx->last_rnn_rows = mh[11];
    *overflowed = mh[10] != 0;
  }
  if (!*overflowed) {
    // what the next one-enqueue batch launches the rare classes with: twice what this one held
    // (and no less than three quarters of the last grid: a batch without wide sentences between two with many does not
    // send the second one through the sized path)
    ctx->spec_grid1 = std::max<u32>(std::max<u32>(64u, 2 * ctx->last_class_n[1]), ctx->spec_grid1 - ctx->spec_grid1 / 4);
    ctx->spec_grid2 = std::max<u32>(std::max<u32>(16u, 2 * ctx->last_class_n[2]), ctx->spec_grid2 - ctx->spec_grid2 / 4);
  }
  return JPPGPU_OK;
  };
  const bool hostInLoop = ctx->seed_hook || ctx->pair_fn || ctx->plugin_fn || ctx->partial_pending || ctx->scored_fns;
  static const bool devExact = std::getenv("JPPGPU_DEV_EXACT") != nullptr;   // (developer knob: always the three-wait path)
  const bool canSpec = !devExact && !hostInLoop && ctx->mail_host != nullptr && ctx->cfg.gbeam != 0 && ctx->node_info.cap != 0 &&
                       ctx->node_beam.cap != 0 && (!ctx->use_rnn || ctx->rnn_ctx.cap != 0);
  bool overflowed = false;
  {
    int prc = pipeline(canSpec, &overflowed);
    if (prc != JPPGPU_OK) return prc;
    if (canSpec) ctx->n_spec_batches++;
    else ctx->n_exact_batches++;
    if (overflowed) {
      ctx->n_spec_overflows++;
      ctx->n_exact_batches++;
      prc = pipeline(false, &overflowed);
      if (prc != JPPGPU_OK) return prc;
    }
  }
  ctx->n_dev_allocs_at_last_batch = g_dev_allocs.load();
  if (ctx->scored_fns != nullptr) {
    // ScorerDef::others on the host (AnalyzerImpl::computeScoresGbeam, analyzer_impl.cc:286-294): every host scorer sees
    // the scored lattice and fills its slot of the cells; the device then re-makes the totals and the EOS beam from the
    // weighted cells (k_adjust.h) and the top-1 paths are traced again
    const jppgpu_score_lattice_fn* fns = ctx->scored_fns;
    void* const* users = ctx->scored_users;
    ctx->scored_fns = nullptr;
    jppgpu_result_view view;
    int frc = jppgpu_result_fetch(Rp, JPPGPU_FETCH_FULL, &view);
    if (frc != JPPGPU_OK) return frc;
    const int S = ctx->cfg.nscorers, G = ctx->cfg.gbeam;
    const int firstHost = 1 + (ctx->use_rnn ? 1 : 0);
    const size_t ncell = (size_t)B.total_nodes * (size_t)G;
    float* cells = Rp->cells.data();
    for (int slot = firstHost; slot < S; ++slot)
      for (size_t q = 0; q < ncell; ++q) cells[q * S + slot] = 0.f;
    for (int h = 0; h < ctx->n_host_scorers; ++h) {
      if (fns[h] == nullptr || fns[h](users ? users[h] : nullptr, &view, (uint32_t)(firstHost + h), cells) != 0)
        return fail(JPPGPU_INVALID_STATE, "jppgpu: a host scorer failed (ScoreComputer::scoreLattice)");
    }
    if (ncell) rt_h2d(B.node_cells, cells, ncell * S * 4, st);
    if (!ctx->adj_stack.ensure(bbN * (size_t)G * 4 + 64)) return fail(JPPGPU_OUT_OF_MEMORY, "device allocation failed (score adjustment)");
    const bool sortE = ctx->cfg.gbeam > 16 || ctx->cfg.gbeam > ctx->cfg.beam * 4 / 3;
    if (sortE) JPP_LAUNCH((k_adjust<true>), (n + 3) / 4, 256, st, B, ctx->cfg, ctx->score_weights, ctx->adj_stack.as<u32>());
    else JPP_LAUNCH((k_adjust<false>), (n + 3) / 4, 256, st, B, ctx->cfg, ctx->score_weights, ctx->adj_stack.as<u32>());
    JPP_LAUNCH(k_path, (n + 255) / 256, 256, st, B, ctx->cfg);
    rt_sync(st);   // (the host copy of the cells may be recycled by the next fetch)
    // beam totals, the EOS beam and the paths have changed: later fetches copy them again
    Rp->fetched_basic = Rp->fetched_full = Rp->fetched_top1 = false;
  }
#if !defined(JPP_EMU)
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(JPPGPU_INVALID_STATE, std::string("kernel launch failed: ") + hipGetErrorString(e));
#endif
  *out = result_guard.release();
  return JPPGPU_OK;
}

extern "C" int jppgpu_analyze_batch(jppgpu_ctx* ctx, const char* utf8, const uint32_t* offsets, uint32_t n,
                                    jppgpu_result** out) {
  if (!ctx || !offsets || !out) return fail(JPPGPU_INVALID_PARAMETER, "null argument");
  if (!bind_device(ctx->device)) return fail(JPPGPU_INVALID_STATE, "hipSetDevice failed for the context's device");
  u32 total = offsets[n];
  if (!(ctx->text.ensure((size_t)total + 64) && ctx->offs.ensure(((size_t)n + 1) * 4)))
    return fail(JPPGPU_OUT_OF_MEMORY, "device allocation failed (input)");
  if (total) rt_h2d(ctx->text.p, utf8, total, ctx->own_stream);
  rt_h2d(ctx->offs.p, offsets, ((size_t)n + 1) * 4, ctx->own_stream);
  rt_sync(ctx->own_stream);
  return jppgpu_analyze_batch_device(ctx, ctx->text.p, ctx->offs.p, n, total, ctx->own_stream, out);
}

extern "C" int jppgpu_analyze_batch_plugin(jppgpu_ctx* ctx, const char* utf8, const uint32_t* offsets, uint32_t n,
                                           jppgpu_score_plugin_fn plugin, void* user, jppgpu_result** out) {
  if (!ctx || !offsets || !out || !plugin) return fail(JPPGPU_INVALID_PARAMETER, "null argument");
  ctx->plugin_fn = plugin;
  ctx->plugin_user = user;
  int rc = jppgpu_analyze_batch(ctx, utf8, offsets, n, out);
  ctx->plugin_fn = nullptr;
  return rc;
}

extern "C" int jppgpu_analyze_batch_pairs(jppgpu_ctx* ctx, const char* utf8, const uint32_t* offsets, uint32_t n,
                                          jppgpu_connection_plugin_fn plugin, void* user, jppgpu_result** out) {
  if (!ctx || !offsets || !out || !plugin) return fail(JPPGPU_INVALID_PARAMETER, "null argument");
  if (ctx->cfg.gbeam <= 0)
    return fail(JPPGPU_INVALID_STATE, "jppgpu: the score plugin acts on global-beam scoring only (as in the reference, analyzer_impl.cc:236-238)");
  ctx->pair_fn = plugin;
  ctx->plugin_user = user;
  int rc = jppgpu_analyze_batch(ctx, utf8, offsets, n, out);
  ctx->pair_fn = nullptr;
  return rc;
}

extern "C" int jppgpu_analyze_batch_scored(jppgpu_ctx* ctx, const char* utf8, const uint32_t* offsets, uint32_t n,
                                           const jppgpu_score_lattice_fn* scorers, void* const* users, uint32_t n_scorers,
                                           jppgpu_result** out) {
  if (!ctx || !offsets || !out || !scorers) return fail(JPPGPU_INVALID_PARAMETER, "null argument");
  if ((int)n_scorers != ctx->n_host_scorers || n_scorers == 0)
    return fail(JPPGPU_INVALID_PARAMETER, "jppgpu_analyze_batch_scored: as many scorers as jppgpu_config::num_host_scorers, at least one");
  ctx->scored_fns = scorers;
  ctx->scored_users = users;
  int rc = jppgpu_analyze_batch(ctx, utf8, offsets, n, out);
  ctx->scored_fns = nullptr;
  return rc;
}

extern "C" int jppgpu_analyze_batch_seeds(jppgpu_ctx* ctx, const char* utf8, const uint32_t* offsets, uint32_t n,
                                          jppgpu_seed_hook_fn hook, void* user, jppgpu_result** out) {
  if (!ctx || !offsets || !out || !hook) return fail(JPPGPU_INVALID_PARAMETER, "null argument");
  if (!ctx->dynamic_spec)
    return fail(JPPGPU_INVALID_STATE, "jppgpu: gold seeds need a context created with dynamic_features = 1 (the trainer's feature code)");
  if (ctx->cfg.nscorers > 1) return fail(JPPGPU_INVALID_STATE, "jppgpu: gold seeds are not scored by the RNN (the trainer runs the perceptron only)");
  if (ctx->row_stride != 8)   // (jppgpu_extra_seed::row holds eight columns)
    return fail(JPPGPU_NOT_IMPLEMENTED, "jppgpu: gold seeds for models with more than 8 feature columns");
  ctx->seed_hook = hook;
  ctx->seed_user = user;
  int rc = jppgpu_analyze_batch(ctx, utf8, offsets, n, out);
  ctx->seed_hook = nullptr;
  return rc;
}

extern "C" int jppgpu_analyze_batch_partial(jppgpu_ctx* ctx, const char* utf8, const uint32_t* offsets, uint32_t n,
                                            const jppgpu_partial* p, jppgpu_result** out) {
  if (!ctx || !offsets || !out || !p) return fail(JPPGPU_INVALID_PARAMETER, "null argument");
  if (!p->nobreak_offsets || !p->boundary_offsets || !p->node_offsets)
    return fail(JPPGPU_INVALID_PARAMETER, "partial annotation offsets are null");
  if (!bind_device(ctx->device)) return fail(JPPGPU_INVALID_STATE, "hipSetDevice failed for the context's device");
  static_assert(sizeof(jppgpu_node_constraint) == sizeof(PcNode) && sizeof(jppgpu_tag_constraint) == sizeof(PcTag),
                "ABI and device constraint records must match");
  const size_t nNb = p->nobreak_offsets[n], nB = p->boundary_offsets[n], nNodes = p->node_offsets[n];
  for (size_t q = 0; q < nNodes; ++q) {
    const jppgpu_node_constraint& c = p->nodes[q];
    if ((size_t)c.tag_first + c.tag_count > p->num_tags) return fail(JPPGPU_INVALID_PARAMETER, "tag range outside tags[]");
    for (uint32_t t = 0; t < c.tag_count; ++t) {
      int32_t f = p->tags[c.tag_first + t].field;
      if (f < 0 || f >= ctx->hmodel.num_features) return fail(JPPGPU_INVALID_PARAMETER, "tag field outside the entry row");
    }
  }
  bool ok = ctx->pc_nb_off.ensure(((size_t)n + 1) * 4) && ctx->pc_b_off.ensure(((size_t)n + 1) * 4) &&
            ctx->pc_node_off.ensure(((size_t)n + 1) * 4) && ctx->pc_nb.ensure(nNb * 2 + 2) && ctx->pc_b.ensure(nB * 2 + 2) &&
            ctx->pc_nodes.ensure(nNodes * sizeof(PcNode) + 4) && ctx->pc_tags.ensure((size_t)p->num_tags * sizeof(PcTag) + 4);
  if (!ok) return fail(JPPGPU_OUT_OF_MEMORY, "device allocation failed (partial)");
  rt_h2d(ctx->pc_nb_off.p, p->nobreak_offsets, ((size_t)n + 1) * 4, ctx->own_stream);
  rt_h2d(ctx->pc_b_off.p, p->boundary_offsets, ((size_t)n + 1) * 4, ctx->own_stream);
  rt_h2d(ctx->pc_node_off.p, p->node_offsets, ((size_t)n + 1) * 4, ctx->own_stream);
  if (nNb) rt_h2d(ctx->pc_nb.p, p->nobreak, nNb * 2, ctx->own_stream);
  if (nB) rt_h2d(ctx->pc_b.p, p->boundaries, nB * 2, ctx->own_stream);
  if (nNodes) rt_h2d(ctx->pc_nodes.p, p->nodes, nNodes * sizeof(PcNode), ctx->own_stream);
  if (p->num_tags) rt_h2d(ctx->pc_tags.p, p->tags, (size_t)p->num_tags * sizeof(PcTag), ctx->own_stream);
  ctx->partial_pending = true;
  int rc = jppgpu_analyze_batch(ctx, utf8, offsets, n, out);
  ctx->partial_pending = false;
  return rc;
}

#if defined(JPP_DEV_PROF) && !defined(JPP_EMU)
// developer build only (dev/jpp_dev_prof.h): cycles per k_sweep phase accumulated since the last call (then reset)
extern "C" int jppgpu_debug_sweep_prof(unsigned long long* out16) {
  hipDeviceSynchronize();
  hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_sweep_prof), 16 * sizeof(unsigned long long));
  unsigned long long rc[2] = {0, 0};
  hipMemcpyFromSymbol(rc, HIP_SYMBOL(g_rnn_cnt), sizeof(rc));
  if (rc[0]) std::fprintf(stderr, "[jppgpu prof] rnn passes %llu nodes %llu (%.2f nodes per pass)\n", rc[0], rc[1], (double)rc[1] / (double)rc[0]);
  unsigned long long z2[2] = {0, 0};
  hipMemcpyToSymbol(HIP_SYMBOL(g_rnn_cnt), z2, sizeof(z2));
  unsigned long long z[16] = {};
  hipMemcpyToSymbol(HIP_SYMBOL(g_sweep_prof), z, sizeof(z));
  return 0;
}
extern "C" int jppgpu_debug_prep_prof(unsigned long long* out8) {
  hipDeviceSynchronize();
  hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_prep_prof), 8 * sizeof(unsigned long long));
  unsigned long long z[8] = {};
  hipMemcpyToSymbol(HIP_SYMBOL(g_prep_prof), z, sizeof(z));
  return 0;
}
#endif

extern "C" int jppgpu_last_timings(jppgpu_ctx* ctx, float* ms, int n) {
  if (!ctx || !ms) return fail(JPPGPU_INVALID_PARAMETER, "null argument");
  if (!bind_device(ctx->device)) return fail(JPPGPU_INVALID_STATE, "hipSetDevice failed for the context's device");
  if (ctx->timing_pending) {
    rt_sync(ctx->last_stream);
    ctx->timer.collect(ctx->last_ms);
    ctx->timing_pending = false;
  }
  for (int i = 0; i < n && i < 11; ++i) ms[i] = ctx->last_ms[i];
  for (int i = 11; i < n && i < 14; ++i) ms[i] = (float)ctx->last_class_n[i - 11];
  if (n > 14) ms[14] = (float)ctx->last_rnn_rows;
  if (n > 15) ms[15] = ctx->last_ms[11];
  if (n > 16) ms[16] = ctx->last_fmt_ms[0];
  if (n > 17) ms[17] = ctx->last_fmt_ms[1];
  if (n > 18) ms[18] = (float)((double)ctx->last_fmt_bytes / 1.0e6);
  return JPPGPU_OK;
}

namespace {
template <typename T>
bool pull(HostVec<T>& v, const void* d, size_t count, jpp_stream_t st) {
  if (!v.resize(count)) return false;
  if (count) rt_d2h(v.data(), d, count * sizeof(T), st);
  return true;
}
template <typename T>
void pull(std::vector<T>& v, const void* d, size_t count, jpp_stream_t st) {
  v.resize(count);
  if (count) rt_d2h(v.data(), d, count * sizeof(T), st);
}
}  // namespace

extern "C" int jppgpu_result_stats(jppgpu_result* res, uint64_t* total_nodes, uint64_t* total_path) {
  if (!res || !res->ctx) return fail(JPPGPU_INVALID_PARAMETER, "null result");
  if (res->generation != res->ctx->generation) return fail(JPPGPU_INVALID_STATE, "result was invalidated");
  if (!bind_device(res->ctx->device)) return fail(JPPGPU_INVALID_STATE, "hipSetDevice failed for the context's device");
  jpp_stream_t st = res->ctx->last_stream;
  std::vector<u32> pl;
  pull(pl, res->B.path_len, res->B.n_sent, st);
  rt_sync(st);
  u64 sum = 0;
  for (auto x : pl) sum += x;
  if (total_nodes) *total_nodes = res->B.total_nodes;
  if (total_path) *total_path = sum;
  return JPPGPU_OK;
}

extern "C" int jppgpu_result_pack(jppgpu_result* res, void* d_offsets, void* d_items, uint64_t cap_items) {
  if (!res || !res->ctx || !d_offsets || (!d_items && cap_items)) return fail(JPPGPU_INVALID_PARAMETER, "null argument");
  jppgpu_ctx* ctx = res->ctx;
  if (res->generation != ctx->generation) return fail(JPPGPU_INVALID_STATE, "result was invalidated");
  if (!bind_device(ctx->device)) return fail(JPPGPU_INVALID_STATE, "hipSetDevice failed for the context's device");
  const Batch& B = res->B;
  const u32 n = B.n_sent;
  if (!(ctx->pack_cnt.ensure(((size_t)n + 1) * 4) && ctx->pack_off.ensure(((size_t)n + 2) * 8)))
    return fail(JPPGPU_OUT_OF_MEMORY, "device allocation failed (pack)");
  jpp_stream_t st = ctx->last_stream;
  const u32 sblocks = (n + 1 + 255) / 256;
  if (n) JPP_LAUNCH(k_pack_count, sblocks, 256, st, B, ctx->pack_cnt.as<u32>());
  launch_scan(ctx, st, (const u32*)ctx->pack_cnt.as<u32>(), ctx->pack_off.as<u64>(), n, (const u64*)nullptr);
  JPP_LAUNCH(k_pack_write, sblocks, 256, st, B, (const u64*)ctx->pack_off.as<u64>(), static_cast<u32*>(d_offsets),
             static_cast<NodeInfo*>(d_items), (u64)cap_items);
  return JPPGPU_OK;
}

// ---- output text on the device (k_format.h) -----------------------------------------------------------------------------
extern "C" int jppgpu_ctx_set_format_table(jppgpu_ctx* ctx, const jppgpu_format_table* t) {
  if (!ctx || !t) return fail(JPPGPU_INVALID_PARAMETER, "null argument");
  if (t->struct_size != sizeof(jppgpu_format_table))
    return fail(JPPGPU_INVALID_PARAMETER, "jppgpu_format_table::struct_size is not the size this library was built with");
  if (!t->slot_first_row || !t->rows || !t->blob || t->n_rows == 0)
    return fail(JPPGPU_INVALID_PARAMETER, "format table: empty");
  if (t->n_rows >= 0xffffffffull || t->blob_bytes >= 0xffffffffull || t->n_slots >= 0xffffffffull)
    return fail(JPPGPU_NOT_IMPLEMENTED, "format table: more than 2^32 rows / blob bytes / slots");
  if (t->n_escapes > 4 || t->n_flags > 16 || t->flag_label_len > 32 || t->eos_len > 16 || t->error_len > 32 || t->flag_placeholder > 1)
    return fail(JPPGPU_INVALID_PARAMETER, "format table: literal beyond its field");
  static_assert(sizeof(FmtRow) == sizeof(jppgpu_format_row), "row layout");
  // The kernels index with what the table says (k_format.h: rows[slot_first_row[..] - 1], blob + blob_off, rows walked
  // to the one flagged "last of its entry"): a table from a file is checked once, here, not trusted.
  for (uint64_t i = 0; i < t->n_slots; ++i)
    if (t->slot_first_row[i] > t->n_rows) return fail(JPPGPU_INVALID_PARAMETER, "format table: a slot names a row beyond the table");
  for (uint64_t i = 0; i < t->n_rows; ++i) {
    const jppgpu_format_row& r = t->rows[i];
    const uint64_t pieces = (uint64_t)r.len_pre + r.len_s + 1 + r.len_r + 1 + r.len_b + r.len_mid + r.len_feat;
    if ((uint64_t)r.blob_off + r.len_total > t->blob_bytes || pieces > r.len_total)
      return fail(JPPGPU_INVALID_PARAMETER, "format table: a row's text lies outside the blob");
  }
  if ((t->rows[t->n_rows - 1].flags & 2) == 0) return fail(JPPGPU_INVALID_PARAMETER, "format table: the last row does not end its entry");
  if (!bind_device(ctx->device)) return fail(JPPGPU_INVALID_STATE, "hipSetDevice failed for the context's device");
  std::lock_guard<std::mutex> shared_lock(ctx->mb->mu);
  // the buffers below may be freed and re-allocated: no kernel of a context that shares the copy may still read them
  if (ctx->mb.use_count() > 1) rt_device_sync();
  if (!(ctx->mb->fmt_slots.ensure(t->n_slots * 4) && ctx->mb->fmt_rows.ensure(t->n_rows * sizeof(FmtRow)) &&
        ctx->mb->fmt_blob.ensure(t->blob_bytes + 64) && ctx->mb->fmt_table.ensure(sizeof(FmtTable))))
    return fail(JPPGPU_OUT_OF_MEMORY, "device allocation failed (format table)");
  jpp_stream_t st = ctx->own_stream;
  rt_h2d(ctx->mb->fmt_slots.p, t->slot_first_row, t->n_slots * 4, st);
  rt_h2d(ctx->mb->fmt_rows.p, t->rows, t->n_rows * sizeof(FmtRow), st);
  rt_h2d(ctx->mb->fmt_blob.p, t->blob, t->blob_bytes, st);
  FmtTable T;
  memset(&T, 0, sizeof(T));
  T.slot_first_row = ctx->mb->fmt_slots.as<u32>();
  T.n_slots = t->n_slots;
  T.rows = ctx->mb->fmt_rows.as<FmtRow>();
  T.n_rows = t->n_rows;
  T.blob = ctx->mb->fmt_blob.as<u8>();
  memcpy(T.maker_replaces, t->maker_replaces, 16);
  T.n_escapes = t->n_escapes;
  memcpy(T.escape_from, t->escape_from, 4);
  memcpy(T.escape_len, t->escape_len, 4);
  memcpy(T.escape_to, t->escape_to, 32);
  T.flag_placeholder = t->flag_placeholder;
  T.flag_label_len = t->flag_label_len;
  memcpy(T.flag_label, t->flag_label, 32);
  T.n_flags = t->n_flags;
  memcpy(T.flag_mask, t->flag_mask, sizeof(T.flag_mask));
  memcpy(T.flag_char, t->flag_char, 16);
  T.eos_len = t->eos_len;
  T.error_len = t->error_len;
  memcpy(T.eos_text, t->eos_text, 16);
  memcpy(T.error_text, t->error_text, 32);
  rt_h2d(ctx->mb->fmt_table.p, &T, sizeof(T), st);
  rt_sync(st);   // (T and the caller's arrays may go away)
  ctx->mb->fmt_have = true;
  return JPPGPU_OK;
}

extern "C" int jppgpu_result_format_top1(jppgpu_result* res, jppgpu_text_view* v) {
  if (!res || !res->ctx || !v) return fail(JPPGPU_INVALID_PARAMETER, "null argument");
  jppgpu_ctx* ctx = res->ctx;
  if (!ctx->mb->fmt_have) return fail(JPPGPU_INVALID_STATE, "jppgpu_result_format_top1 needs jppgpu_ctx_set_format_table");
  if (!bind_device(ctx->device)) return fail(JPPGPU_INVALID_STATE, "hipSetDevice failed for the context's device");
  const Batch& B = res->B;
  const u32 n = B.n_sent;
  if (res->fm_have && res->fm_kind != 1) return fail(JPPGPU_INVALID_STATE, "the result holds the lattice text already");
  if (!res->fm_have) {
    if (res->generation != ctx->generation)
      return fail(JPPGPU_INVALID_STATE, "result was invalidated by a later jppgpu_analyze_batch on the same context");
    jpp_stream_t st = ctx->last_stream;
    if (!(ctx->fmt_len.ensure((B.total_nodes + 1) * 4) && ctx->fmt_cnt.ensure(((size_t)n + 1) * 4) && ctx->fmt_off.ensure(((size_t)n + 2) * 8) && ctx->fmt_st.ensure(((size_t)n + 1) * 4)))
      return fail(JPPGPU_OUT_OF_MEMORY, "device allocation failed (format)");
    const FmtTable* T = ctx->mb->fmt_table.as<FmtTable>();
    ctx->fmt_timer.mark(0, st);
    if (n) JPP_LAUNCH(k_fmt_count, (n + 3) / 4, 256, st, B, T, ctx->fmt_len.as<u32>(), ctx->fmt_cnt.as<u32>(), ctx->fmt_st.as<i32>());
    launch_scan(ctx, st, (const u32*)ctx->fmt_cnt.as<u32>(), ctx->fmt_off.as<u64>(), n, (const u64*)nullptr);
    ctx->fmt_timer.mark(1, st);
    bool ok = pull(res->fm_off, ctx->fmt_off.p, (size_t)n + 1, st);
    rt_sync(st);   // the byte total sizes the text buffers
    if (!ok) return fail(JPPGPU_OUT_OF_MEMORY, "host allocation failed (format offsets)");
    const u64 total = res->fm_off.data()[n];
    if (!ctx->fmt_text.ensure(total + 64)) return fail(JPPGPU_OUT_OF_MEMORY, "device allocation failed (format text)");
    ctx->fmt_timer.mark(2, st);
    if (n) JPP_LAUNCH(k_fmt_write, (n + 3) / 4, 256, st, B, T, (const u32*)ctx->fmt_len.as<u32>(), (const u64*)ctx->fmt_off.as<u64>(),
                      ctx->fmt_text.as<u8>(), (const i32*)ctx->fmt_st.as<i32>());
    ctx->fmt_timer.mark(3, st);
    ok = pull(res->fm_text, ctx->fmt_text.p, (size_t)total, st);
    ok &= pull(res->fm_status, ctx->fmt_st.p, n, st);   // (the text's own status: a sentence the table cannot render answers ST_CAPACITY)
    rt_sync(st);
    ctx->fmt_timer.collect(ctx->last_fmt_ms);
    ctx->last_fmt_bytes = total;
    if (!ok) return fail(JPPGPU_OUT_OF_MEMORY, "host allocation failed (format text)");
    res->fm_have = true;
    res->fm_kind = 1;
  }
  v->n_sentences = n;
  v->offsets = res->fm_off.data();
  v->text = res->fm_text.data();
  v->status = res->fm_status.data();
  return JPPGPU_OK;
}

// ---- the lattice format on the device (k_latfmt.h) ------------------------------------------------------------------------
extern "C" int jppgpu_ctx_set_lattice_table(jppgpu_ctx* ctx, const jppgpu_lattice_table* t) {
  if (!ctx || !t) return fail(JPPGPU_INVALID_PARAMETER, "null argument");
  if (t->struct_size != sizeof(jppgpu_lattice_table))
    return fail(JPPGPU_INVALID_PARAMETER, "jppgpu_lattice_table::struct_size is not the size this library was built with");
  if (!t->slot_first_row || !t->rows || !t->blob || t->n_rows == 0)
    return fail(JPPGPU_INVALID_PARAMETER, "lattice table: empty");
  if (t->n_rows >= 0xffffffffull || t->blob_bytes >= 0xffffffffull || t->n_slots >= 0xffffffffull)
    return fail(JPPGPU_NOT_IMPLEMENTED, "lattice table: more than 2^32 rows / blob bytes / slots");
  if (t->n_escapes > 4 || t->n_flags > 16 || t->flag_label_len > 32 || t->flag_placeholder > 1 || t->head_len > 16 || t->rank_len > 8 ||
      t->feat_len > 32 || t->lm_len > 32 || t->total_len > 32 || t->ranks_len > 16 || t->eos_len > 16 || t->error_len > 32 ||
      t->n_weights < 1 || t->n_weights > 2)
    return fail(JPPGPU_INVALID_PARAMETER, "lattice table: literal beyond its field");
  for (int e = 0; e < (int)t->n_escapes; ++e)
    if (t->escape_len[e] > 8) return fail(JPPGPU_INVALID_PARAMETER, "lattice table: literal beyond its field");
  static_assert(sizeof(LatRow) == sizeof(jppgpu_lattice_row), "row layout");
  // checked once, not trusted (the table may come from a file): k_latfmt.h indexes with these
  for (uint64_t i = 0; i < t->n_slots; ++i)
    if (t->slot_first_row[i] > t->n_rows) return fail(JPPGPU_INVALID_PARAMETER, "lattice table: a slot names a row beyond the table");
  for (uint64_t i = 0; i < t->n_rows; ++i) {
    const jppgpu_lattice_row& r = t->rows[i];
    const uint64_t lenX = r.len_c ? (uint64_t)r.len_c : (uint64_t)r.len_b + 1 + r.len_r;
    const uint64_t total = (uint64_t)r.len_s + 1 + lenX + 1 + r.len_r + 1 + r.len_b + 1 + r.len_rest;
    if ((uint64_t)r.blob_off + total > t->blob_bytes) return fail(JPPGPU_INVALID_PARAMETER, "lattice table: a row's text lies outside the blob");
  }
  if ((t->rows[t->n_rows - 1].flags & 2) == 0) return fail(JPPGPU_INVALID_PARAMETER, "lattice table: the last row does not end its entry");
  if (!bind_device(ctx->device)) return fail(JPPGPU_INVALID_STATE, "hipSetDevice failed for the context's device");
  std::lock_guard<std::mutex> shared_lock(ctx->mb->mu);
  if (ctx->mb.use_count() > 1) rt_device_sync();
  if (!(ctx->mb->lat_slots.ensure(t->n_slots * 4) && ctx->mb->lat_rows.ensure(t->n_rows * sizeof(LatRow)) &&
        ctx->mb->lat_blob.ensure(t->blob_bytes + 64) && ctx->mb->lat_table.ensure(sizeof(LatTable))))
    return fail(JPPGPU_OUT_OF_MEMORY, "device allocation failed (lattice table)");
  jpp_stream_t st = ctx->own_stream;
  rt_h2d(ctx->mb->lat_slots.p, t->slot_first_row, t->n_slots * 4, st);
  rt_h2d(ctx->mb->lat_rows.p, t->rows, t->n_rows * sizeof(LatRow), st);
  rt_h2d(ctx->mb->lat_blob.p, t->blob, t->blob_bytes, st);
  LatTable T;
  memset(&T, 0, sizeof(T));
  T.slot_first_row = ctx->mb->lat_slots.as<u32>();
  T.n_slots = t->n_slots;
  T.rows = ctx->mb->lat_rows.as<LatRow>();
  T.n_rows = t->n_rows;
  T.blob = ctx->mb->lat_blob.as<u8>();
  memcpy(T.maker_replaces, t->maker_replaces, 16);
  T.n_escapes = t->n_escapes;
  memcpy(T.escape_from, t->escape_from, 4);
  memcpy(T.escape_len, t->escape_len, 4);
  memcpy(T.escape_to, t->escape_to, 32);
  T.flag_placeholder = t->flag_placeholder;
  T.flag_label_len = t->flag_label_len;
  memcpy(T.flag_label, t->flag_label, 32);
  T.n_flags = t->n_flags;
  memcpy(T.flag_mask, t->flag_mask, sizeof(T.flag_mask));
  memcpy(T.flag_char, t->flag_char, 16);
  T.head_len = t->head_len; T.rank_len = t->rank_len; T.feat_len = t->feat_len; T.lm_len = t->lm_len;
  T.total_len = t->total_len; T.ranks_len = t->ranks_len; T.eos_len = t->eos_len; T.error_len = t->error_len;
  memcpy(T.head_text, t->head_text, 16);
  memcpy(T.rank_text, t->rank_text, 8);
  memcpy(T.feat_text, t->feat_text, 32);
  memcpy(T.lm_text, t->lm_text, 32);
  memcpy(T.total_text, t->total_text, 32);
  memcpy(T.ranks_text, t->ranks_text, 16);
  memcpy(T.eos_text, t->eos_text, 16);
  memcpy(T.error_text, t->error_text, 32);
  T.n_weights = t->n_weights;
  T.weights[0] = t->weights[0];
  T.weights[1] = t->weights[1];
  rt_h2d(ctx->mb->lat_table.p, &T, sizeof(T), st);
  rt_sync(st);
  ctx->mb->lat_have = true;
  return JPPGPU_OK;
}

extern "C" int jppgpu_result_format_lattice(jppgpu_result* res, int32_t n_best, jppgpu_lattice_text_view* v) {
  if (!res || !res->ctx || !v) return fail(JPPGPU_INVALID_PARAMETER, "null argument");
  jppgpu_ctx* ctx = res->ctx;
  if (!ctx->mb->lat_have) return fail(JPPGPU_INVALID_STATE, "jppgpu_result_format_lattice needs jppgpu_ctx_set_lattice_table");
  if (n_best < 1 || n_best > 64) return fail(JPPGPU_INVALID_PARAMETER, "jppgpu_result_format_lattice: n_best outside 1..64");
  if (res->cfg.gbeam <= 0) return fail(JPPGPU_NOT_IMPLEMENTED, "lattice format needs the global beam (score cells)");
  if (!bind_device(ctx->device)) return fail(JPPGPU_INVALID_STATE, "hipSetDevice failed for the context's device");
  const Batch& B = res->B;
  const u32 n = B.n_sent;
  if (res->fm_have && (res->fm_kind != 2 || res->fm_nbest != n_best)) return fail(JPPGPU_INVALID_STATE, "the result holds another text already");
  if (!res->fm_have) {
    if (res->generation != ctx->generation)
      return fail(JPPGPU_INVALID_STATE, "result was invalidated by a later jppgpu_analyze_batch on the same context");
    jpp_stream_t st = ctx->last_stream;
    const size_t N = (size_t)B.total_nodes + 1;
    if (!(ctx->lat_mask.ensure(N * 8) && ctx->lat_best.ensure(N * 8) && ctx->lat_id.ensure(N * 4) && ctx->lat_list.ensure(N * sizeof(LatRec)) && ctx->lat_marked.ensure(((size_t)n + 1) * 4) &&
          ctx->lat_head.ensure(((size_t)n + 1) * 4) && ctx->fmt_cnt.ensure(((size_t)n + 1) * 4) && ctx->fmt_off.ensure(((size_t)n + 2) * 8) &&
          ctx->fmt_st.ensure(((size_t)n + 1) * 4)))
      return fail(JPPGPU_OUT_OF_MEMORY, "device allocation failed (lattice format)");
    const LatTable* T = ctx->mb->lat_table.as<LatTable>();
    const LatScratch S{ctx->lat_mask.as<u64>(), ctx->lat_best.as<u64>(), ctx->lat_id.as<u32>(), ctx->lat_list.as<LatRec>(), ctx->lat_marked.as<u32>()};
    // (developer switches for the tests: k_latfmt.h kLatDevWinMask / kLatDevManyPrev)
    static const u32 latDev = [] {
      u32 v = 0;
      if (const char* w = std::getenv("JPPGPU_DEV_LAT_WIN")) v |= (u32)std::strtoul(w, nullptr, 10) & kLatDevWinMask;
      if (std::getenv("JPPGPU_DEV_LAT_MANY_PREV") != nullptr) v |= kLatDevManyPrev;
      return v;
    }();
    ctx->fmt_timer.mark(0, st);
    if (n) JPP_LAUNCH(k_lat_count, (n + 3) / 4, 256, st, B, res->cfg, T, S, (int)n_best, ctx->fmt_cnt.as<u32>(), ctx->lat_head.as<u32>(), ctx->fmt_st.as<i32>(), latDev);
    launch_scan(ctx, st, (const u32*)ctx->fmt_cnt.as<u32>(), ctx->fmt_off.as<u64>(), n, (const u64*)nullptr);
    ctx->fmt_timer.mark(1, st);
    bool ok = pull(res->fm_off, ctx->fmt_off.p, (size_t)n + 1, st);
    rt_sync(st);   // the byte total sizes the text buffers
    if (!ok) return fail(JPPGPU_OUT_OF_MEMORY, "host allocation failed (format offsets)");
    const u64 total = res->fm_off.data()[n];
    if (!ctx->fmt_text.ensure(total + 64)) return fail(JPPGPU_OUT_OF_MEMORY, "device allocation failed (format text)");
    ctx->fmt_timer.mark(2, st);
    if (n) JPP_LAUNCH(k_lat_write, (n + 3) / 4, 256, st, B, res->cfg, T, S, (int)n_best, (const u64*)ctx->fmt_off.as<u64>(), (const u32*)ctx->lat_head.as<u32>(),
                      ctx->fmt_text.as<u8>(), (const i32*)ctx->fmt_st.as<i32>(), latDev);
    ctx->fmt_timer.mark(3, st);
    ok = pull(res->fm_text, ctx->fmt_text.p, (size_t)total, st);
    ok &= pull(res->fm_status, ctx->fmt_st.p, n, st);
    ok &= pull(res->fm_head, ctx->lat_head.p, n, st);
    rt_sync(st);
    ctx->fmt_timer.collect(ctx->last_fmt_ms);
    ctx->last_fmt_bytes = total;
    if (!ok) return fail(JPPGPU_OUT_OF_MEMORY, "host allocation failed (format text)");
    res->fm_have = true;
    res->fm_kind = 2;
    res->fm_nbest = n_best;
  }
  v->n_sentences = n;
  v->offsets = res->fm_off.data();
  v->text = res->fm_text.data();
  v->status = res->fm_status.data();
  v->head_len = res->fm_head.data();
  return JPPGPU_OK;
}

extern "C" int jppgpu_result_fetch(jppgpu_result* res, int full, jppgpu_result_view* v) {
  if (!res || !res->ctx || !v) return fail(JPPGPU_INVALID_PARAMETER, "null argument");
  jppgpu_ctx* ctx = res->ctx;
  if (res->generation != ctx->generation)
    return fail(JPPGPU_INVALID_STATE, "result was invalidated by a later jppgpu_analyze_batch on the same context");
  if (!bind_device(ctx->device)) return fail(JPPGPU_INVALID_STATE, "hipSetDevice failed for the context's device");
  jpp_stream_t st = ctx->last_stream;
  const Batch& B = res->B;
  const u32 n = B.n_sent;
  const u64 N = B.total_nodes;
  const int G = res->cfg.gbeam, beam = res->cfg.beam;
  bool ok = true;  // host blocks obtained
  if (full == JPPGPU_FETCH_TOP1) {
    if (!res->fetched_top1) {
      // an upper bound of the compact size: a path holds at most one node per codepoint, plus EOS
      const u64 cap = (u64)B.total_bytes + n + 1;
      if (!(ctx->pack_cnt.ensure(((size_t)n + 1) * 4) && ctx->pack_off.ensure(((size_t)n + 2) * 8) &&
            ctx->top1_nodes.ensure(cap * sizeof(NodeInfo)) && ctx->top1_aux.ensure(cap * sizeof(NodeAux))))
        return fail(JPPGPU_OUT_OF_MEMORY, "device allocation failed (top-1 fetch)");
      if (n) JPP_LAUNCH(k_top1_count, (n + 255) / 256, 256, st, B, ctx->pack_cnt.as<u32>());
      launch_scan(ctx, st, (const u32*)ctx->pack_cnt.as<u32>(), ctx->pack_off.as<u64>(), n, (const u64*)nullptr);
      if (n) JPP_LAUNCH(k_top1_gather, (n + 3) / 4, 256, st, B, (const u64*)ctx->pack_off.as<u64>(),
                        ctx->top1_nodes.as<NodeInfo>(), ctx->top1_aux.as<NodeAux>());
      ok &= pull(res->t1_status, B.sent_status, n, st);
      ok &= pull(res->t1_ncp, B.sent_ncp, n, st);
      ok &= pull(res->t1_len, ctx->pack_cnt.p, n, st);
      ok &= pull(res->t1_base, ctx->pack_off.p, (size_t)n + 1, st);
      rt_sync(st);
      if (!ok) return fail(JPPGPU_OUT_OF_MEMORY, "host allocation failed (result copies)");
      const u64 M = n ? res->t1_base[n] : 0;
      ok &= pull(res->t1_nodes, ctx->top1_nodes.p, M, st);
      ok &= pull(res->t1_unk, ctx->top1_aux.p, M, st);
      // every sentence's path is 0, 1, 2, ... in its compact table
      if (!(ok && res->t1_idx.resize((size_t)M))) return fail(JPPGPU_OUT_OF_MEMORY, "host allocation failed (result copies)");
      for (u32 s = 0; s < n; ++s)
        for (u32 k = 0; k < res->t1_len[s]; ++k) res->t1_idx[res->t1_base[s] + k] = k;
      res->t1_zero.assign(n, 0);
      rt_sync(st);
      res->fetched_top1 = true;
    }
    if (!ok) return fail(JPPGPU_OUT_OF_MEMORY, "host allocation failed (result copies)");
    memset(v, 0, sizeof(*v));
    v->n_sentences = n;
    v->status = res->t1_status.data();
    v->n_codepoints = res->t1_ncp.data();
    v->n_nodes = res->t1_len.data();
    v->node_base = res->t1_base.data();
    v->bnd_base = res->t1_zero.data();
    v->total_nodes = n ? res->t1_base[n] : 0;
    v->total_boundaries = 0;
    v->beam = beam;
    v->global_beam = G;
    v->num_scorers = res->cfg.nscorers;
    v->path_len = res->t1_len.data();
    v->path_nodes = res->t1_idx.data();
    v->nodes = res->t1_nodes.data();
    v->unk = res->t1_unk.data();
    return JPPGPU_OK;
  }
  if (!res->fetched_basic) {
    ok &= pull(res->status, B.sent_status, n, st);
    ok &= pull(res->ncp, B.sent_ncp, n, st);
    ok &= pull(res->nnodes, B.sent_nodes, n, st);
    ok &= pull(res->node_base, B.node_base, n, st);
    ok &= pull(res->path_len, B.path_len, n, st);
    ok &= pull(res->byte_off, B.byte_off, (size_t)n + 1, st);
    if (n) {
      // nodes live in [0, seed region end); copy the whole region that holds any sentence
      ok &= pull(res->nodes, B.node_info, N, st);
      ok &= pull(res->unk, B.node_aux, N, st);
      ok &= pull(res->path_nodes, B.path_nodes, N, st);
    }
    rt_sync(st);
    if (!(ok && res->bnd_base.resize(n))) return fail(JPPGPU_OUT_OF_MEMORY, "host allocation failed (result copies)");
    for (u32 s = 0; s < n; ++s) res->bnd_base[s] = (u64)res->byte_off[s] + 4ull * s;
    for (u32 s = 0; s < n; ++s) {
      if (res->status[s] != ST_OK) {
        res->nnodes[s] = 0;
        res->path_len[s] = 0;
      }
    }
    res->fetched_basic = true;
  }
  const u64 NB = n ? (u64)B.total_bytes + 4ull * n : 0;
  if (full && !res->fetched_full && n) {
    ok &= pull(res->bnd_first, B.bnd_first, NB, st);
    ok &= pull(res->bnd_cnt, B.bnd_cnt, NB, st);
    ok &= pull(res->end_first, B.end_first, NB, st);
    ok &= pull(res->end_cnt, B.end_cnt, NB, st);
    ok &= pull(res->ngb, B.bnd_ngb, NB, st);
    ok &= pull(res->gbeam, B.bnd_gbeam, NB * G * 2, st);
    ok &= pull(res->end_nodes, B.end_nodes, N, st);
    ok &= pull(res->entry_rows, B.node_entry, N * ctx->row_stride, st);
    ok &= pull(res->patterns, B.node_pat, N * kPat, st);
    ok &= pull(res->t0, B.node_t0, N, st);
    ok &= pull(res->beams, B.node_beam, N * beam, st);
    ok &= pull(res->cells, B.node_cells, N * G * res->cfg.nscorers, st);
    ok &= pull(res->kept, B.node_kept, N, st);
    rt_sync(st);
    res->fetched_full = true;
  }
  if (!ok) return fail(JPPGPU_OUT_OF_MEMORY, "host allocation failed (result copies)");
  memset(v, 0, sizeof(*v));
  v->n_sentences = n;
  v->status = res->status.data();
  v->n_codepoints = res->ncp.data();
  v->n_nodes = res->nnodes.data();
  v->node_base = res->node_base.data();
  v->bnd_base = res->bnd_base.data();
  v->total_nodes = N;
  v->total_boundaries = NB;
  v->beam = beam;
  v->global_beam = G;
  v->num_scorers = res->cfg.nscorers;
  v->entry_row_stride = (int32_t)ctx->row_stride;
  v->path_len = res->path_len.data();
  v->path_nodes = res->path_nodes.data();
  v->nodes = res->nodes.data();
  v->unk = res->unk.data();
  if (res->fetched_full) {
    v->bnd_first = res->bnd_first.data();
    v->bnd_count = res->bnd_cnt.data();
    v->end_first = res->end_first.data();
    v->end_count = res->end_cnt.data();
    v->end_nodes = res->end_nodes.data();
    v->entry_rows = res->entry_rows.data();
    v->patterns = res->patterns.data();
    v->t0_scores = res->t0.data();
    v->beams = res->beams.data();
    v->cells = res->cells.data();
    v->kept = res->kept.data();
    v->gbeam_count = res->ngb.data();
    v->gbeam = res->gbeam.data();
  }
  return JPPGPU_OK;
}

extern "C" int jppgpu_result_fetch_nbest(jppgpu_result* res, int32_t n_best, jppgpu_nbest_view* v) {
  if (!res || !res->ctx || !v) return fail(JPPGPU_INVALID_PARAMETER, "null argument");
  if (n_best <= 0 || n_best > 64) return fail(JPPGPU_INVALID_PARAMETER, "n_best must be in 1..64");
  static_assert(sizeof(jppgpu_nbest_item) == sizeof(NbestItem), "nbest item layout");
  jppgpu_ctx* ctx = res->ctx;
  if (!bind_device(ctx->device)) return fail(JPPGPU_INVALID_STATE, "hipSetDevice failed for the context's device");
  const Batch& B = res->B;
  const u32 n = B.n_sent;
  if (res->nb_n != n_best) {
    if (res->generation != ctx->generation)
      return fail(JPPGPU_INVALID_STATE, "result was invalidated by a later jppgpu_analyze_batch on the same context");
    jpp_stream_t st = ctx->last_stream;
    const u64 paths = (u64)n * (u32)n_best;
    if (paths >= 0xffffffffull) return fail(JPPGPU_INVALID_PARAMETER, "too many paths");
    if (!(ctx->nbest_cnt.ensure((paths + 1) * 4) && ctx->nbest_off.ensure((paths + 2) * 8) &&
          ctx->nbest_eos.ensure((paths + 1) * sizeof(BeamSlot))))
      return fail(JPPGPU_OUT_OF_MEMORY, "device allocation failed (n-best fetch)");
    bool ok = true;
    if (n) {
      JPP_LAUNCH((k_nbest<false>), (n + 3) / 4, 256, st, B, res->cfg, (int)n_best, ctx->nbest_cnt.as<u32>(),
                 (const u64*)nullptr, (NbestItem*)nullptr, (BeamSlot*)nullptr);
    }
    launch_scan(ctx, st, (const u32*)ctx->nbest_cnt.as<u32>(), ctx->nbest_off.as<u64>(), (u32)paths, (const u64*)nullptr);
    ok &= pull(res->nb_first, ctx->nbest_off.p, (size_t)paths + 1, st);
    rt_sync(st);
    if (!ok) return fail(JPPGPU_OUT_OF_MEMORY, "host allocation failed (result copies)");
    const u64 M = res->nb_first[paths];
    if (!ctx->nbest_items.ensure((M + 1) * sizeof(NbestItem))) return fail(JPPGPU_OUT_OF_MEMORY, "device allocation failed (n-best fetch)");
    if (n) {
      JPP_LAUNCH((k_nbest<true>), (n + 3) / 4, 256, st, B, res->cfg, (int)n_best, ctx->nbest_cnt.as<u32>(),
                 (const u64*)ctx->nbest_off.as<u64>(), ctx->nbest_items.as<NbestItem>(), ctx->nbest_eos.as<BeamSlot>());
    }
    ok &= pull(res->nb_status, B.sent_status, n, st);
    ok &= pull(res->nb_ncp, B.sent_ncp, n, st);
    ok &= pull(res->nb_nnodes, B.sent_nodes, n, st);
    ok &= pull(res->nb_eos, ctx->nbest_eos.p, (size_t)paths, st);
    ok &= pull(res->nb_items, ctx->nbest_items.p, (size_t)M, st);
    rt_sync(st);
    if (!ok) return fail(JPPGPU_OUT_OF_MEMORY, "host allocation failed (result copies)");
    for (u32 s = 0; s < n; ++s)
      if (res->nb_status[s] != ST_OK) res->nb_nnodes[s] = 0;
    res->nb_n = n_best;
  }
  memset(v, 0, sizeof(*v));
  v->n_sentences = n;
  v->n_best = n_best;
  v->beam = res->cfg.beam;
  v->global_beam = res->cfg.gbeam;
  v->num_scorers = res->cfg.nscorers;
  v->status = res->nb_status.data();
  v->n_codepoints = res->nb_ncp.data();
  v->n_nodes = res->nb_nnodes.data();
  v->eos = res->nb_eos.data();
  v->path_first = res->nb_first.data();
  v->items = res->nb_items.data();
  return JPPGPU_OK;
}

extern "C" int jppgpu_result_fetch_top1_ngrams(jppgpu_result* res, jppgpu_top1_ngrams_view* v) {
  if (!res || !res->ctx || !v) return fail(JPPGPU_INVALID_PARAMETER, "null argument");
  jppgpu_ctx* ctx = res->ctx;
  if (!bind_device(ctx->device)) return fail(JPPGPU_INVALID_STATE, "hipSetDevice failed for the context's device");
  const Batch& B = res->B;
  const u32 n = B.n_sent;
  if (!ctx->builtin_spec)
    return fail(JPPGPU_NOT_IMPLEMENTED, "jppgpu: the trainer read-out exists for the built-in jumandic spec only");
  if (!res->ng_have) {
    if (res->generation != ctx->generation)
      return fail(JPPGPU_INVALID_STATE, "result was invalidated by a later jppgpu_analyze_batch on the same context");
    jpp_stream_t st = ctx->last_stream;
    if (!(ctx->nbest_cnt.ensure(((size_t)n + 1) * 4) && ctx->nbest_off.ensure(((size_t)n + 2) * 8)))
      return fail(JPPGPU_OUT_OF_MEMORY, "device allocation failed (n-gram fetch)");
    if (n) JPP_LAUNCH(k_path_count, (n + 255) / 256, 256, st, B, ctx->nbest_cnt.as<u32>());
    launch_scan(ctx, st, (const u32*)ctx->nbest_cnt.as<u32>(), ctx->nbest_off.as<u64>(), n, (const u64*)nullptr);
    bool ok = pull(res->ng_first, ctx->nbest_off.p, (size_t)n + 1, st);
    rt_sync(st);
    if (!ok) return fail(JPPGPU_OUT_OF_MEMORY, "host allocation failed (result copies)");
    const u64 M = res->ng_first[n];
    if (!(ctx->ng_nodes.ensure((M + 1) * 4) && ctx->ng_feat.ensure((M + 1) * kNumNgram * 4)))
      return fail(JPPGPU_OUT_OF_MEMORY, "device allocation failed (n-gram fetch)");
    if (n) JPP_LAUNCH(k_path_ngrams, n, 64, st, B, (const u64*)ctx->nbest_off.as<u64>(), ctx->ng_nodes.as<u32>(), ctx->ng_feat.as<u32>());
    ok = pull(res->ng_nodes, ctx->ng_nodes.p, (size_t)M, st) && pull(res->ng_feat, ctx->ng_feat.p, (size_t)M * kNumNgram, st);
    rt_sync(st);
    if (!ok) return fail(JPPGPU_OUT_OF_MEMORY, "host allocation failed (result copies)");
    res->ng_have = true;
  }
  v->n_sentences = n;
  v->n_ngram = (uint32_t)kNumNgram;
  v->path_first = res->ng_first.data();
  v->path_nodes = res->ng_nodes.data();
  v->features = res->ng_feat.data();
  return JPPGPU_OK;
}

extern "C" int jppgpu_result_fetch_path_ngrams(jppgpu_result* res, const uint64_t* path_first, const uint32_t* path_nodes,
                                               jppgpu_top1_ngrams_view* v) {
  if (!res || !res->ctx || !v || !path_first) return fail(JPPGPU_INVALID_PARAMETER, "null argument");
  jppgpu_ctx* ctx = res->ctx;
  if (!bind_device(ctx->device)) return fail(JPPGPU_INVALID_STATE, "hipSetDevice failed for the context's device");
  const Batch& B = res->B;
  const u32 n = B.n_sent;
  if (!ctx->builtin_spec)
    return fail(JPPGPU_NOT_IMPLEMENTED, "jppgpu: the trainer read-out exists for the built-in jumandic spec only");
  if (res->generation != ctx->generation)
    return fail(JPPGPU_INVALID_STATE, "result was invalidated by a later jppgpu_analyze_batch on the same context");
  const u64 M = path_first[n];
  if (path_first[0] != 0 || (M && !path_nodes)) return fail(JPPGPU_INVALID_PARAMETER, "bad path offsets");
  for (u32 q = 0; q < n; ++q)
    if (path_first[q + 1] < path_first[q]) return fail(JPPGPU_INVALID_PARAMETER, "bad path offsets");
  jpp_stream_t st = ctx->last_stream;
  if (!(ctx->nbest_off.ensure(((size_t)n + 2) * 8) && ctx->ng_nodes.ensure((M + 1) * 4) && ctx->ng_feat.ensure((M + 1) * kNumNgram * 4)))
    return fail(JPPGPU_OUT_OF_MEMORY, "device allocation failed (n-gram fetch)");
  rt_h2d(ctx->nbest_off.p, path_first, ((size_t)n + 1) * 8, st);
  if (M) rt_h2d(ctx->ng_nodes.p, path_nodes, (size_t)M * 4, st);
  if (n) JPP_LAUNCH(k_given_path_ngrams, n, 64, st, B, (const u64*)ctx->nbest_off.as<u64>(), (const u32*)ctx->ng_nodes.as<u32>(), ctx->ng_feat.as<u32>());
  bool ok = res->gp_first.resize((size_t)n + 1) && res->gp_nodes.resize((size_t)M) && pull(res->gp_feat, ctx->ng_feat.p, (size_t)M * kNumNgram, st);
  rt_sync(st);
  if (!ok) return fail(JPPGPU_OUT_OF_MEMORY, "host allocation failed (result copies)");
  memcpy(res->gp_first.data(), path_first, ((size_t)n + 1) * 8);
  if (M) memcpy(res->gp_nodes.data(), path_nodes, (size_t)M * 4);
  v->n_sentences = n;
  v->n_ngram = (uint32_t)kNumNgram;
  v->path_first = res->gp_first.data();
  v->path_nodes = res->gp_nodes.data();
  v->features = res->gp_feat.data();
  return JPPGPU_OK;
}

extern "C" int jppgpu_ctx_set_weights(jppgpu_ctx* ctx, const float* weights, uint64_t n) {
  if (!ctx || !weights) return fail(JPPGPU_INVALID_PARAMETER, "null argument");
  if (n != (uint64_t)ctx->hmodel.wmask + 1) return fail(JPPGPU_INVALID_PARAMETER, "weight count does not match the model's table");
  if (!bind_device(ctx->device)) return fail(JPPGPU_INVALID_STATE, "hipSetDevice failed for the context's device");
  std::lock_guard<std::mutex> shared_lock(ctx->mb->mu);
  // (ordered behind every batch already enqueued on the context's streams -- and, when the model copy is shared
  // (jppgpu_ctx_create_shared), behind every batch of the sibling contexts: the table and the T0 records are theirs too.
  // The caller must not START a batch on a sibling while this call runs.)
  if (ctx->mb.use_count() > 1) rt_device_sync();
  if (ctx->last_stream) rt_sync(ctx->last_stream);
  rt_sync(ctx->own_stream);
  rt_h2d(ctx->mb->weights.p, weights, (size_t)n * 4, nullptr);
  rt_sync(nullptr);
  if (ctx->mb->t0_memo_from_image) {
    // (the records came from a cache of the model file's own weights and there are no seeds to re-derive them from:
    // the context goes on without them -- k_t0 instead of k_t0_memo, same results)
    ctx->mb->t0_memo_slots = 0;
    ctx->mb->t0_memo_from_image = false;
  }
  if (ctx->mb->t0_memo_slots && !upload_memo(ctx, weights)) return fail(JPPGPU_OUT_OF_MEMORY, "device allocation failed (T0 memo)");
  return JPPGPU_OK;
}

extern "C" void jppgpu_result_release(jppgpu_result* res) {
  // results are views on the context workspace (same lifetime rule as the reference:
  // valid until the next analyze on that Analyzer); this drops the handle and host copies.
  delete res;
}
