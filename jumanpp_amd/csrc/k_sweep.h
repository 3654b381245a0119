// Kernel 5: the boundary sweep -- global-beam Viterbi with bigram/trigram
// perceptron scoring.  One wavefront (64-lane workgroup) per sentence walks
// the boundaries in order (the recurrence is sequential over boundaries); all
// per-boundary work is spread over the 64 lanes:
//   * global beam  : wave-wide arg-max over the unique u64 keys, G rounds
//   * prescores    : 8 lanes per (gbeam head, right node), round-robin partial sums
//   * tail scoring : 4 lanes per (kept right node, unique T1), 1 lane per (node, tail entry)
//   * beams        : stable rank of each candidate among <= G totals
// The per-boundary working set (gbeam, T1/T2 patterns, prescores, totals) is
// staged in ~10 KB of LDS; node patterns / beams / weights stay in HBM.
//
// Every float sum below reproduces the association order of the reference
// (SURVEY section 7 "Float summation order"); the comments name the code.
//
// Reference behaviour reproduced:
//   AnalyzerImpl::computeScoresGbeam        src/core/analysis/analyzer_impl.cc:250-297
//   ScoreProcessor::makeGlobalBeam          src/core/analysis/score_processor.cc:246-282
//   processBeamCandidates / BeamCandidate   score_processor.cc:193-206, score_processor.h:81-115
//   ScoreProcessor::computeGbeamScores      score_processor.cc:284-361
//   dedupT1 / gatherT1 / gatherT2           score_processor.cc:363-409
//   computeT0Prescores                      score_processor.cc:497-511
//   generated applyBiStep2 / applyTriStep3  (8 / 4 round-robin accumulators; last row
//                                            computeUnrolled4RawPerceptron, perceptron.h:46-72)
//   makeT0cutoffBeam (std::nth_element)     score_processor.cc:471-495
//   applyBiTriFullKernel                    src/core/impl/feature_impl_ngram_partial_kernels.h:19-111
//   copyT0Scores / makeT0Beam               score_processor.cc:411-469
#ifndef JPP_K_SWEEP_H
#define JPP_K_SWEEP_H

#include "jpp_device.h"
#include "jpp_select.h"
#include "k_t0.h"

namespace jpp {

struct NgramTables {
  u64 bi_pre[spec::kNumBi];
  int bi_t0[spec::kNumBi];
  int bi_t1[spec::kNumBi];
  u64 tri_pre[spec::kNumTri];
  int tri_t0[spec::kNumTri];
  int tri_t1[spec::kNumTri];
  int tri_t2[spec::kNumTri];
  constexpr NgramTables() : bi_pre{}, bi_t0{}, bi_t1{}, tri_pre{}, tri_t0{}, tri_t1{}, tri_t2{} {
    for (int k = 0; k < spec::kNumBi; ++k) {
      bi_pre[k] = bi_prefix(spec::kBi[k].index);
      bi_t0[k] = spec::kBi[k].t0;
      bi_t1[k] = spec::kBi[k].t1;
    }
    for (int k = 0; k < spec::kNumTri; ++k) {
      tri_pre[k] = tri_prefix(spec::kTri[k].index);
      tri_t0[k] = spec::kTri[k].t0;
      tri_t1[k] = spec::kTri[k].t1;
      tri_t2[k] = spec::kTri[k].t2;
    }
  }
};
constexpr NgramTables kNg{};

constexpr int kSweepWaves = 4;  // wavefronts per SIMD the LDS footprint of the narrow variant allows
constexpr int kSweepWideWaves = 3;  // ... and of the wide variant (beam up to 32): 168 VGPRs, about 15 KB of LDS
constexpr int kChunk = 8;       // right nodes processed per pass (= 8-lane groups per wave)
constexpr int kPresCap = 1024;  // rcheck * R prescores staged in LDS

__device__ __forceinline__ bool slot_fake(const BeamSlot& s) { return s.left == kFake16 && s.beam == kFake16; }

// The value itself, but opaque to the optimiser: what is computed from it inside the boundary loop is not hoisted out
// of the loop into (sixty-odd, in round 2) loop-invariant VGPRs, which is what bounds the wavefronts per SIMD.
__device__ __forceinline__ int opaque_i32(int v) {
#if !defined(JPP_EMU) && defined(__HIP_DEVICE_COMPILE__)
  asm volatile("" : "+v"(v));
#endif
  return v;
}
__device__ __forceinline__ u32 opaque_u32(u32 v) {
#if !defined(JPP_EMU) && defined(__HIP_DEVICE_COMPILE__)
  asm volatile("" : "+v"(v));
#endif
  return v;
}

// cold paths of the sweep as real calls: inlined, their temporaries (six butterfly addresses of the wave-wide maximum,
// the state of the nth_element replay) are live across the whole boundary loop
// more candidates than the LDS staging holds (rare): G rounds of a wave-wide maximum over the beam slots in HBM
__device__ __attribute__((noinline)) int global_beam_from_hbm(const BeamSlot* beams, const u32* ends, u32 ncand, int beam,
                                                              int G, u64* gb_key) {
  const int lane = lane_id();
  u64 last = ~u64{0};
  int ngb = 0;
  for (int r = 0; r < G; ++r) {
    u64 best = 0;
    for (u32 q = (u32)lane; q < ncand; q += 64) {
      u32 l = q / (u32)beam, k = q - l * (u32)beam;
      BeamSlot sl = beams[(u64)ends[l] * beam + k];
      if (!slot_fake(sl)) {
        u64 key = ((u64)f32_sortable(sl.total) << 32) | ((u64)l << 16) | k;
        if (key < last && key > best) best = key;
      }
    }
    u64 win = wave_max_u64(best);
    if (win == 0) break;
    if (lane == 0) gb_key[r] = win;
    last = win;
    ++ngb;
  }
  return ngb;
}

__device__ __attribute__((noinline)) void cutoff_replay(u16* order, const float* csum, u32 rbeam, u32 R) {
  ScoreGreater cmp{csum};
  nth_element_u16(order, order + rbeam, order + R, cmp);
}

// q / d for the small runtime divisors of the sweep (beam sizes, global-beam counts: d <= 32) and q < 2048: one
// multiply by ceil(2^16 / d) and a shift instead of the ~25-instruction reciprocal sequence of a 32-bit division.
// Exact: q * (d * inv - 2^16) < q * d < 2^16.
struct SmallDivTab {
  u32 inv[33];
  constexpr SmallDivTab() : inv{} {
    inv[0] = 0;
    for (u32 d = 1; d <= 32; ++d) inv[d] = (65536u + d - 1) / d;
  }
};
constexpr SmallDivTab kSmallDiv{};
__device__ __forceinline__ u32 small_div_inv(u32 d) { return kSmallDiv.inv[d]; }   // d wave-uniform: a scalar load
__device__ __forceinline__ u32 small_div(u32 q, u32 inv) {
#if defined(JPP_EMU)
  return (q * inv) >> 16;
#else
  return __umul24(q, inv) >> 16;   // both factors below 2^24: the full-rate multiplier
#endif
}

// GM = compile-time capacity of the global beam / per-node beam (8 for the CLI defaults, 32 for wide beams)

// ---- 8-lane group helpers -------------------------------------------------------
// The bigram features of one (right node, T1 row) pair are spread over an 8-lane group:
// lane j of the group owns features k = j + 8m (m < 5), its table entries live in
// registers (loaded once per kernel), so all weight gathers of a group are issued
// together.  The reference's summation orders are then rebuilt with shuffles.
constexpr int kBiPerLane = (kDynMaxBi + 7) / 8;   // feature slots per lane of an 8-lane group (kDynMaxBi >= spec::kNumBi)
static_assert(spec::kNumBi <= kDynMaxBi && spec::kNumTri <= kDynMaxTri, "lane layout of the bigram / trigram features");

// tables in LDS (filled once per kernel)
struct LaneBi {
  const u64* pre;   // [kNumBi] hash prefixes
  const u8* t01;    // [kNumBi] (t0 << 4) | t1
  u32 t1pack;       // this lane's T1 pattern indices: 4 bits per feature slot m (features gj + 8 m)
};
#define JPP_LBI_T1(t, m, j) (((t).t1pack >> (4 * (m))) & 15u)

// weights of this lane's bigram features for one (right node, T1 row) pair, from the cached first-stage
// states of the right node: s1[k] = hmix(prefix_k, p0[t0_k])
template <bool W24>
// (nBi: the number of bigram features -- spec::kNumBi, a constant after inlining, or the count of a table-driven spec)
__device__ __forceinline__ void bi_gather_s1(const LaneBi& t, int j, const u64* s1, const u64* t1r,
                                             const float JPP_GLOBAL* __restrict__ W, u32 wmask, bool act, float* w, int nBi) {
  u32 idx[kBiPerLane];
#pragma unroll
  for (int m = 0; m < kBiPerLane; ++m) {
    const int k = (j + 8 * m) < nBi ? (j + 8 * m) : 0;
    idx[m] = hmix_index<W24>(s1[k], t1r[JPP_LBI_T1(t, m, j)], wmask);
  }
#pragma unroll
  for (int m = 0; m < kBiPerLane; ++m) w[m] = (act && (j + 8 * m) < nBi) ? W[idx[m]] : 0.f;
}

// generated applyBiStep2: f_j = 0 + w_j + w_{j+8} + ..., then f_0 + f_1 + ... + f_7.
// The three sums below are valid on the group leader (gj == 0) only: it reads its members through
// row_shl DPP modifiers in exactly the order of the scalar code.
__device__ __forceinline__ float bi_sum8(const float* w, int lane, int j, int nBi) {
  float f = 0.f;
#pragma unroll
  for (int m = 0; m < kBiPerLane; ++m)
    if (j + 8 * m < nBi) f += w[m];
  float total = f;
  total += row_shl_f32<1>(f);
  total += row_shl_f32<2>(f);
  total += row_shl_f32<3>(f);
  total += row_shl_f32<4>(f);
  total += row_shl_f32<5>(f);
  total += row_shl_f32<6>(f);
  total += row_shl_f32<7>(f);
  return total;
}

// computeUnrolled4RawPerceptron: r_q = 0 + w_q + w_{q+4} + ... (q < 4), then ((r0 + r1) + r2) + r3.
// Feature q + 4n sits in lane q (n even) or lane q + 4 (n odd) of the group.
__device__ __forceinline__ float bi_sum4(const float* w, int lane, int j, int nBi) {
  float r = 0.f;
#pragma unroll
  for (int m = 0; m < kBiPerLane; ++m) {
    float other = row_shl_f32<4>(w[m]);  // lanes 0..3 of the group read lanes 4..7
    if (j + 8 * m < nBi) r += w[m];
    if (j + 4 + 8 * m < nBi) r += other;
  }
  float total = r;
  total += row_shl_f32<1>(r);
  total += row_shl_f32<2>(r);
  total += row_shl_f32<3>(r);
  return total;
}

// applyBiTriFullKernel: r1 = 0 + w_0 + w_2 + ..., r2 = 0 + w_1 + w_3 + ..., result r1 + r2
// (lane 0 of the group accumulates the even features, lane 1 the odd ones)
__device__ __forceinline__ float bi_sum2(const float* w, int lane, int j, int nBi) {
  const int par = j & 1;
  float r = 0.f;
#pragma unroll
  for (int m = 0; m < kBiPerLane; ++m) {
    float v0 = w[m], v2 = row_shl_f32<2>(w[m]), v4 = row_shl_f32<4>(w[m]), v6 = row_shl_f32<6>(w[m]);
    if (0 + par + 8 * m < nBi) r += v0;
    if (2 + par + 8 * m < nBi) r += v2;
    if (4 + par + 8 * m < nBi) r += v4;
    if (6 + par + 8 * m < nBi) r += v6;
  }
  return r + row_shl_f32<1>(r);
}

// Partial-annotation ScorePlugin: penalty of every lattice node, one lane per node.
// PartialExample::checkViolation (src/core/input/partial_example.cc:23-73) decides from the RIGHT node of
// a connection only, so PexStreamReaderImpl::updateScore (pex_stream_reader.cc:24-39) subtracts the same
// amount from every connection into that node: 10000 for a word start/end on a no-break position, a required
// boundary inside the node or a length mismatch, 1000 for a tag mismatch, nothing otherwise.
__global__ void k_penalty(Batch B) {
  const u32 s = blockIdx.x;
  if (B.sent_status[s] != ST_OK) return;
  const u32 N = B.sent_nodes[s];
  const u64 nb = B.node_base[s];
  const u32 nb0 = B.pc_nb_off[s], nb1 = B.pc_nb_off[s + 1];
  const u32 b0 = B.pc_b_off[s], b1 = B.pc_b_off[s + 1];
  const u32 n0 = B.pc_node_off[s], n1 = B.pc_node_off[s + 1];
  for (u32 k = threadIdx.x; k < N; k += blockDim.x) {
    float pen = 0.f;
    if (k >= 2) {
      const NodeInfo ni = B.node_info[nb + k];
      const u32 boundary = (u32)ni.start + 2;
      const u32 len = (u32)ni.end - ni.start;
      const u32 end = boundary + len;
      bool hard = false, tag = false, done = false;
      for (u32 q = nb0; q < nb1 && !done; ++q) {
        u32 bnd = B.pc_nb[q];
        if (bnd == boundary || bnd == end) {
          hard = true;
          done = true;
        } else if (bnd > end) {
          break;
        }
      }
      for (u32 q = b0; q < b1 && !done; ++q) {
        u32 bnd = B.pc_b[q];
        if (bnd <= boundary) continue;
        if (bnd >= end) break;
        hard = true;
        done = true;
      }
      for (u32 q = n0; q < n1 && !done; ++q) {
        PcNode c = B.pc_nodes[q];
        if (c.boundary != boundary) continue;
        if (len != c.length) {
          hard = true;
        } else {
          const i32* row = B.node_entry + (nb + k) * B.row_stride;
          for (u32 t = 0; t < c.tag_count; ++t) {
            PcTag tg = B.pc_tags[c.tag_first + t];
            if (row[tg.field] != tg.value) {
              tag = true;
              break;
            }
          }
        }
        done = true;  // std::find_if: only the first constraint at that boundary counts
      }
      pen = hard ? 10000.f : (tag ? 1000.f : 0.f);
    }
    B.node_penalty[nb + k] = pen;
  }
}

__device__ __attribute__((noinline)) BndMeta load_bnd_meta(const BndMeta* g, u32 q) { return g[q]; }

// RM = capacity of right nodes per boundary staged in LDS (every sentence runs the variant of its own widest
// boundary: k_sweep_classify)
// The workgroup is one wavefront: the phases are separated by wave_sync() (compiler + LDS ordering only).
// A __syncthreads() would additionally drain the vector-memory counter, i.e. wait for every outstanding
// global store (beams, cells) ~12 times per boundary.
// DEF: the configuration is the CLI default (beam 5, global beam 6, right-check 1, right-beam 5,
// jumanpp_args.h:50-54): the four numbers become compile-time constants (no runtime divisions by the beam
// size, fixed trip counts); any other configuration runs the same code with the values read from `cfg`.
// W24: the weight table has at most 2^24 entries (hmix_index).
// WAVES: wavefronts per SIMD the variant is compiled for (VGPR budget 512 / WAVES); its LDS footprint must allow as many.
// LEAN: the small-LDS layouts of the default-configuration variants (see kLean below): 0 = off, 1 = lean with the next
// boundary's pattern rows in a buffer of their own, 2 = the tightest layout (one row buffer).
// DYN: a spec other than the built-in jumandic tables: the n-gram descriptors come from DevModel::spec, and every
// sum takes the association of the reference's DYNAMIC feature code (PartialNgramFeatureApplyImpl,
// feature_impl_ngram_partial.h:188-357): computeUnrolled4RawPerceptron for every row of every n-gram order.
template <int GM, int RM, bool DEF = false, bool W24 = false, int WAVES = kSweepWaves, int LEAN = 0, bool DYN = false>
__global__ void __launch_bounds__(64) JPP_WAVES_PER_EU_RANGE(WAVES, WAVES)
k_sweep(Batch B, const DevModel* __restrict__ Mp, Config cfg, const u32* __restrict__ slist) {
  const DevModel& M = *Mp;
  // workgroup -> sentence through the list of the variant's class (k_sweep_classify)
  // (the one-enqueue path launches a class with a grid chosen before its size is known: the list holds
  // gstats[1 + class] sentences, k_sweep_classify; the lists are sweep_list + class * n_sent)
  if (blockIdx.x >= B.gstats[1 + (u32)((u64)(slist - B.sweep_list) / B.n_sent)]) return;
  const u32 s = slist[blockIdx.x];
  if (B.sent_status[s] != ST_OK) return;
  const int lane = (int)threadIdx.x;
  const u32 off = B.byte_off[s];
  const u32 bb0 = off + 4 * s;
  const u32 n = B.sent_ncp[s];
  const u64 nb = B.node_base[s];
  static_assert(!DEF || GM >= 6, "default configuration needs a global beam of 6");
  const int beam = DEF ? 5 : cfg.beam;
  const int G = DEF ? 6 : cfg.gbeam;
  const int rcheck = DEF ? 1 : cfg.rcheck;
  const int rbeam = DEF ? 5 : cfg.rbeam;
  const float JPP_GLOBAL* __restrict__ W = as_global(M.weights);
  const u32 wmask = M.wmask;
  const u32* en = B.end_nodes + nb;
  BeamSlot* beams = B.node_beam + nb * beam;
  const u64* pats = B.node_pat + nb * kPat;
  const float* t0s = B.node_t0 + nb;

  __shared__ u64 gb_key[GM];
  __shared__ u16 gb_left[GM];
  __shared__ u16 gb_slot[GM];
  __shared__ float gb_score[GM];
  __shared__ u32 gb_lnode[GM];
  __shared__ u32 gb_pnode[GM];
  __shared__ u32 gb_t1[GM];
  __shared__ u32 t1node[GM];
  // kLean (the CLI-default variants): the LDS footprint is what bounds the wavefronts per CU, and the kernel is a
  // chain of dependent round trips that only other wavefronts hide (6.4 / 8.2 / 10.6 ms at 4 / 3 / 2 per SIMD,
  // profiles/r03_a_occupancy.txt) -- every array is as small as the default configuration allows: rows for the 6
  // global-beam entries, one prescore per right node that doubles as the cutoff sum, ONE pattern-row buffer (rows
  // are dead once their first-stage hash states exist), a 16-entry layout ring.  6.5 KB instead of 9.8 KB.
  static_assert(!LEAN || (DEF && RM > 0), "the lean layout is written for the default configuration");
  static_assert(!DYN || (!DEF && !W24), "the table-driven variant is the generic one");
  const int nBi = DYN ? M.spec->nbi : spec::kNumBi;
  const int nTri = DYN ? M.spec->ntri : spec::kNumTri;
  constexpr bool kLean = LEAN != 0;
  // kOneRow: the tightest layout (6 wavefronts per SIMD: 6.5 KB) has ONE pattern-row buffer and small rings; the
  // 5-wavefront layout (8 KB) keeps the next boundary's rows apart (requested a whole boundary ahead)
  constexpr bool kOneRow = LEAN == 2;
  constexpr int kGR = kLean ? 6 : GM;   // rows of the per-global-beam-entry arrays
  __shared__ u64 t1pat[kGR][kPat];
  constexpr int kT2 = DYN ? kPat : 4;  // pattern fields of the T2 node the trigrams read (built-in spec: indices 0..3)
  static_assert(spec::kTri[0].t2 < 4 && spec::kTri[1].t2 < 4 && spec::kTri[2].t2 < 4 && spec::kTri[3].t2 < 4 && spec::kNumTri == 4,
                "t2pat holds pattern fields 0..3 only");
  __shared__ u64 t2pat[kGR][kT2];
  // the three per-right-node arrays: in LDS (capacity RM), or -- RM == 0, the variant of sentences with a boundary
  // wider than the LDS variants stage -- in an HBM scratch slice of the workgroup sized from the batch maximum, so
  // that no lattice is ever too wide (the reference has no limit, lattice_builder.cc:70-93)
  constexpr int kRMs = RM > 0 ? RM : 1;
  // (kLean: right-check 1, so one prescore per right node, and the cutoff sum 0.f + prescore IS the prescore --
  // except that it turns -0.f into +0.f, which no comparison of the cutoff distinguishes)
  __shared__ float pres_lds[(kLean ? 1 : 2) * kRMs];
  __shared__ float csum_lds[kLean ? 1 : kRMs];
  __shared__ u16 order_lds[kRMs];
  float* pres = pres_lds;
  float* csum = kLean ? pres_lds : csum_lds;
  u16* order = order_lds;
  if constexpr (RM == 0) {
    unsigned char* base = B.sweep_scratch + (u64)blockIdx.x * B.sweep_scratch_stride;
    pres = reinterpret_cast<float*>(base);                                   // [rcheck][maxR]
    csum = pres + (size_t)(cfg.rcheck > 0 ? cfg.rcheck : 1) * B.sweep_scratch_maxr;   // [maxR]
    order = reinterpret_cast<u16*>(csum + B.sweep_scratch_maxr);           // [maxR]
  }
  __shared__ float biS[kChunk][kGR];
  // default configuration: tail-association bigram sum of (right node t, T1 row 0), formed in the prescore pass
  constexpr bool kHeadShare = DEF && RM > 0;
  __shared__ float biS0[kHeadShare ? RM : 1];
  __shared__ __attribute__((aligned(16))) float tot[kChunk][GM];
  __shared__ u8 hpart[GM > 16 ? 2 : 1][2][GM > 16 ? 36 : 1];   // per half-wave: positions of the k-th stop of the upward / downward scan (B1)
  __shared__ u8 shave[GM > 16 ? kChunk : 1];   // per node of the pass: length of the sorted range | 0x80 if already in final order
  __shared__ u8 sreplay[GM > 16 ? kChunk : 1];  // per node of the pass: its totals tie, the sort is replayed
  __shared__ float t0R[kChunk];
  // static data of a boundary, fetched asynchronously (global_load_lds) while the previous boundary is
  // being scored: patterns / T0 of its first kChunk right nodes and its ends list; double buffered
  // candidate slots (left nodes x beam) staged in LDS per boundary; more take global_beam_from_hbm (wide variant: 8 left
  // nodes x 32 -- at 512 the buffer alone held the kernel at 6 wavefronts per CU)
  constexpr int kCandCap = GM <= 8 ? 64 : 256;   // (wide variant: KEYS in registers, 4 per lane and chunk -- no slots staged in LDS)
  constexpr int kCandSlots = GM <= 8 ? 64 : 0;   // beam slots of the candidates staged in LDS (narrow variant only)
  // pattern rows of up to kChunk right nodes: kLean has ONE buffer -- a boundary's rows are dead as soon as their
  // first-stage states are in s1b / s1t, the next boundary's (or the next pass's) rows are requested right then;
  // the other variants keep the next boundary's rows (pRn[par ^ 1]) apart from the pass buffer pR
  __shared__ __attribute__((aligned(16))) u64 pRn[kOneRow ? 1 : 2][kChunk][kPat];
  __shared__ __attribute__((aligned(16))) u64 pR_lds[kLean ? 1 : kChunk][kPat];
  // patterns of the right nodes of the current pass (R > kChunk only): kLean re-uses the boundary's own row buffer
  // (its rows are dead once their first-stage states exist)
  auto pRbuf = [&](int cur) -> u64(*)[kPat] { return kLean ? pRn[kOneRow ? 0 : cur] : pR_lds; };
  __shared__ __attribute__((aligned(16))) float t0n[2][kChunk];
  constexpr u32 kEnnCap = kLean ? 16 : GM <= 8 ? 32 : 64;   // ends-list entries staged per boundary
  __shared__ __attribute__((aligned(16))) u32 enn[2][kEnnCap];
  // first-stage hash states of the right nodes of the current pass: every bigram / trigram index starts with
  // hmix(prefix_k, p0[t0_k]), which depends on the right node only and is shared by all its T1 / T2 partners
  constexpr int kS1 = kOneRow ? spec::kNumBi : 40;  // >= kNumBi, row stride
  static_assert(spec::kNumBi <= kS1, "state row too short");
  // The candidate slots / keys of phase 1 and the bigram states of phases 3-5 are never live together
  // (the states die with the tail of a boundary, the candidates of the next one are requested after it),
  // so they share one buffer.
  constexpr u32 kCandBytes = kCandSlots * sizeof(BeamSlot) + 64 * sizeof(u64);
  constexpr u32 kS1Bytes = kChunk * kS1 * sizeof(u64);
  __shared__ __attribute__((aligned(16))) unsigned char u_buf[kCandBytes > kS1Bytes ? kCandBytes : kS1Bytes];
  BeamSlot* const cand = reinterpret_cast<BeamSlot*>(u_buf);                    // live beam slots of the left nodes
  u64* const ckey = reinterpret_cast<u64*>(u_buf + kCandSlots * sizeof(BeamSlot));  // their keys (rank selection)
  u64(*const s1b)[kS1] = reinterpret_cast<u64(*)[kS1]>(u_buf);
  // makeT0Beam replay (wide variant, ties only): (total bits << 32 | candidate index) per candidate of the pass's nodes.
  // Written and read in 5c only, when the candidates' keys (phase 1) and the bigram states (dead after 5a) are gone.
  u64(*const skey)[GM] = reinterpret_cast<u64(*)[GM]>(u_buf);
  static_assert(GM <= 16 || sizeof(u_buf) >= kChunk * GM * sizeof(u64), "replay keys share the candidate / state buffer");
  __shared__ u64 s1t[kChunk][kDynMaxTri];
  __shared__ u64 s_tripre[kDynMaxTri];
  __shared__ u8 s_trit[kDynMaxTri][4];

  const int grp = lane >> 3, gj = lane & 7;
  LaneBi lbi;
  __shared__ u64 s_bipre[DYN ? kDynMaxBi : spec::kNumBi];
  __shared__ u8 s_bit01[DYN ? kDynMaxBi : spec::kNumBi];
  static_assert(kPat <= 16, "pattern indices are packed in 4 bits");
  if constexpr (DYN) {
    if (lane < nBi) {
      s_bipre[lane] = M.spec->bi_prefix[lane];
      s_bit01[lane] = M.spec->bi_t01[lane];
    }
    if (lane < kDynMaxTri) {
      s_tripre[lane] = lane < nTri ? M.spec->tri_prefix[lane] : 0;
      for (int q = 0; q < 3; ++q) s_trit[lane][q] = lane < nTri ? M.spec->tri_t[lane][q] : (u8)0;
    }
  } else {
    if (lane < spec::kNumBi) {
      s_bipre[lane] = kNg.bi_pre[lane];
      s_bit01[lane] = (u8)((kNg.bi_t0[lane] << 4) | kNg.bi_t1[lane]);
    }
    if (lane < spec::kNumTri) {
      s_tripre[lane] = kNg.tri_pre[lane];
      s_trit[lane][0] = (u8)kNg.tri_t0[lane];
      s_trit[lane][1] = (u8)kNg.tri_t1[lane];
      s_trit[lane][2] = (u8)kNg.tri_t2[lane];
    }
  }
  lbi.pre = s_bipre;
  lbi.t01 = s_bit01;
  wave_sync();
  // the T1 pattern indices of the features of group lane j (4 bits per feature slot m), fetched once per boundary
  __shared__ u32 s_t1pack[8];
  if (lane < 8) {
    u32 pk = 0;
#pragma unroll
    for (int m = 0; m < kBiPerLane; ++m) pk |= (u32)(s_bit01[(lane + 8 * m) < nBi ? (lane + 8 * m) : 0] & 15) << (4 * m);
    s_t1pack[lane] = pk;
  }
  wave_sync();
  lbi.t1pack = s_t1pack[gj];

  if (n == 0) {
    // empty input: the reference returns before scoring anything
    // (computeScoresGbeam: bndCount <= 3, analyzer_impl.cc:255-258)
    for (int q = lane; q < beam; q += 64) beams[(u64)2 * beam + q] = BeamSlot{kFake16, kFake16, 0.f, 0xffffffffu, 0};
    if (lane == 0) B.bnd_ngb[bb0 + 2] = 0;
    return;
  }
  // per-boundary layout records: a kRing-entry LDS ring (slot b mod kRing), filled asynchronously;
  // entries that are not resident (kRing or more boundaries ahead) are read from HBM
  constexpr u32 kRing = kLean ? 16 : 32;
  __shared__ __attribute__((aligned(16))) BndMeta meta[kRing];
  const BndMeta* gmeta = B.bnd_meta + bb0;
  u32 metaEnd = (n + 3) < kRing ? (n + 3) : kRing;  // records below metaEnd have been requested ...
  lds_async_load<16>(&meta[0], gmeta + lane, (u32)lane < metaEnd);
  lds_async_wait();
  wave_sync();
  u32 metaReady = metaEnd;                      // ... and those below metaReady have landed
  // The HBM read sits behind a noinline call so that the two loads cannot be merged into one flat load
  // through a selected pointer: flat access to the LDS aperture faults on this platform.
  auto metaAt = [&](u32 q) -> BndMeta {
    if (q < metaReady) return meta[q & (kRing - 1)];
    return load_bnd_meta(gmeta, q);
  };
  // the next boundary at or after `from` where nodes start, and its record (read once, carried into the
  // prefetch and into the next iteration)
  auto next_nonempty = [&](u32 from, BndMeta& out) {
    u32 q = from;
    for (; q <= n + 2; ++q) {
      out = metaAt(q);
      if (out.cnt != 0) break;
    }
    return q;
  };
  // pattern rows of the first kChunk right nodes of boundary bq into the row buffer `rbuf`
  auto prefetch_rows = [&](u32 bq, const BndMeta& mq, int rbuf) {
    if (bq > n + 2) return;
    const int lane = lane_now();
    const u32 nxr = mq.cnt < (u32)kChunk ? mq.cnt : (u32)kChunk;
    static_assert(kPat * 8 % 16 == 0 && kChunk * kPat * 8 / 16 <= 64, "one dwordx4 per lane covers a chunk");
    lds_async_load<16>(&pRn[rbuf][0][0], reinterpret_cast<const char*>(pats + (u64)mq.first * kPat) + lane * 16,
                       (u32)lane < nxr * (kPat * 8 / 16));
  };
  auto prefetch = [&](u32 bq, const BndMeta& mq, int buf) {
    if (bq > n + 2) return;
    const int lane = lane_now();
    const u32 Rq = mq.cnt, rf = mq.first, Lq = mq.ecnt, ef = mq.efirst;
    const u32 nxr = Rq < (u32)kChunk ? Rq : (u32)kChunk;
    if constexpr (!kOneRow) prefetch_rows(bq, mq, buf);   // (kOneRow: requested when the single row buffer falls free)
    lds_async_load<4>(&t0n[buf][0], t0s + rf + lane, (u32)lane < nxr);
    lds_async_load<4>(&enn[buf][0], en + ef + lane, (u32)lane < (Lq < kEnnCap ? Lq : kEnnCap));
  };
  // first-stage states of `nx` right nodes whose pattern rows are rows[0..nx): lane = feature (37 bigram + 4 trigram
  // features: one pass of 41 lanes), a loop over the nodes.  The lane's hash prefix and pattern index stay in
  // registers across the nodes; a lane-per-(node, feature) layout paid a division by 41 and three table reads per
  // element for nothing.
  auto compute_s1 = [&](const u64(*rows)[kPat], u32 nx) {
    const u32 kF = (u32)(nBi + nTri);
    static_assert(kDynMaxBi + kDynMaxTri <= 64, "one lane per n-gram feature");
    const int fl = lane_now();   // (the table reads below stay inside the loop)
    if ((u32)fl < kF) {
      const bool bi = fl < nBi;
      const int kt = bi ? 0 : fl - nBi;
      const u64 pre = bi ? s_bipre[fl] : s_tripre[kt];
      const u32 t0i = bi ? (u32)(s_bit01[fl] >> 4) : (u32)s_trit[kt][0];
      for (u32 x = 0; x < nx; ++x) {
        const u64 v = hmix(pre, rows[x][t0i]);
        if (bi) s1b[x][fl] = v;
        else s1t[x][kt] = v;
      }
    }
  };
  JPP_PROF_DECL;
  BndMeta mbn{0, 0, 0, 0};
  u32 bn = next_nonempty(2, mbn);
  // (records and everything counted from them are wave-uniform: keep them in scalar registers, see uni())
  auto uni_meta = [](BndMeta& m) {
    m.first = uni(m.first);
    m.cnt = uni(m.cnt);
    m.efirst = uni(m.efirst);
    m.ecnt = uni(m.ecnt);
  };
  bn = uni(bn);
  uni_meta(mbn);
  int par = 0;
  prefetch(bn, mbn, par);
  if constexpr (kOneRow) prefetch_rows(bn, mbn, 0);
  lds_async_wait();
  for (u32 b = bn; b <= n + 2; b = bn, par ^= 1) {
    // The records / rows requested during the previous boundary are needed from here on.  They were
    // requested before that boundary's candidate slots / pattern rows, and vector memory operations complete
    // in order, so the waits of phases 1-2 have covered them: no wait here -- it would only drain the beam
    // and cell stores the previous boundary has just issued (every path that skips those phases waits itself).
    wave_sync();
    // this iteration's own lane index (lane_now): nothing derived from it can be hoisted out of the loop
    const int lane = lane_now();
    const int grp = lane >> 3, gj = lane & 7;
    lbi.t1pack = s_t1pack[gj];
    const BndMeta mb = mbn;
    const u32 R = mb.cnt;
    const u32 rfirst = mb.first;
    const u32 L = mb.ecnt;
    const u32 efirst = mb.efirst;
    if (RM > 0 && R > (u32)RM) {   // cannot happen: the sentence was routed here by its widest boundary
      if (lane == 0) B.sent_status[s] = ST_CAPACITY;
      return;
    }
    // The ring records requested during the previous boundary have landed -- INVARIANT (nothing checks it; the emulator
    // copies synchronously): every path through an iteration ends with a vmcnt(0) behind its prefetch() and ring refill:
    //   (1) fastCand: the lds_async_wait() behind the candidate-slot copy of phase 1,
    //   (2) !fastCand: the lds_async_wait() in front of global_beam_from_hbm(),
    //   (3) no global beam entry (ngb == 0): the lds_async_wait() before `continue`,
    // and the first iteration starts behind the lds_async_wait() in front of the loop.  A new path through the loop body
    // needs its own wait before the next iteration reads `meta` / the prefetched rows.
    // Boundaries below b are done: recycle their ring slots for the records up to kRing - 1 ahead.
    metaReady = metaEnd;
    if (metaEnd < n + 3 && metaEnd <= b + kRing / 2) {   // refill when half of the window is used up
      const u32 lo = metaEnd, hi = (b + kRing) < (n + 3) ? (b + kRing) : (n + 3);
      // slots lo..hi-1 (mod 64) may wrap: issue the two contiguous pieces separately
      const u32 s0 = lo & (kRing - 1), cntAll = hi - lo;
      const u32 c0 = (s0 + cntAll) <= kRing ? cntAll : kRing - s0;
      lds_async_load<16>(&meta[s0], gmeta + lo + lane, (u32)lane < c0);
      lds_async_load<16>(&meta[0], gmeta + lo + c0 + lane, (u32)lane < cntAll - c0);
      metaEnd = hi;
    }
    bn = uni(next_nonempty(b + 1, mbn));
    uni_meta(mbn);
    prefetch(bn, mbn, par ^ 1);
    const u32* enL = enn[par];  // ends list of this boundary (first 64 entries)
    u64(*const pR)[kPat] = pRbuf(par);

    JPP_PROF(0);
    // ---- 1. global beam: top-G of all live (left, slot) by the packed key ----
    int ngb = 0;
    const u32 ncand = L * (u32)beam;
    const u32 invBeam = small_div_inv((u32)beam);   // (candidate index -> (left, slot); ncand <= 2048 is checked below)
    // (wide variant: chunks of kCandCap keys, any number of them up to the staged ends list)
    const bool fastCand = (GM > 8 ? ncand <= 2048u : ncand <= (u32)kCandCap) && L <= kEnnCap;
    if constexpr (GM > 8) {
      // WIDE VARIANT (round 4).  Up to 512 candidates = 16 left nodes at beam 32: every lane reads the 8 bytes of its
      // candidates' slots that make the key (left | beam | total) straight into registers, eight loads in flight; nothing
      // is staged in LDS.  (Until round 4: 256 slots of 16 bytes in LDS, and anything beyond that -- every boundary with
      // more than 8 left nodes -- took G rounds of a wave-wide maximum over HBM: 42 % of the kernel on the configs[4]
      // shape, profiles/r04_m_phases5.txt.)
      if (fastCand) {
        // Chunks of 256 candidates (4 keys per lane): the G best keys of every chunk go to a pool in LDS, the G best of
        // the pool are the global beam (the top G of a union lie in the union of the parts' top G).  One chunk is the
        // common case; 2 048 candidates = 64 left nodes at beam 32 make eight chunks and one pool pass.
        u64* const pool = ckey + 64;   // [256]: u_buf holds 64 scratch keys + the pool (2 560 bytes)
        static_assert(kCandCap == 256 && sizeof(u_buf) >= (64 + 256) * sizeof(u64), "pool of the chunked global beam");
        // the bits a key can have: 63..32 total, 16 + log2(kEnnCap) .. 16 left index, 4..0 (beam <= 32) slot
        auto key_bit = [](int bit) { return bit >= 32 || (bit >= 16 && bit <= 22) || bit <= 5; };
        // top G of the (<= 256) keys in `k4`, ranked, written to dst[0 .. return value)
        auto select4 = [&](u64 (&k4)[4], u64* dst) -> int {
          int live = 0;
#pragma unroll
          for (int jx = 0; jx < 4; ++jx) live += popc64(wave_ballot(k4[jx] != 0));
          const int want = live < G ? live : G;
          u64 lo = 1;   // every live key is >= 1
          if (live > G) {
            // the G-th largest of the 64 per-lane maxima is a lower bound of the G-th largest key
            u64 mx = 0;
#pragma unroll
            for (int jx = 0; jx < 4; ++jx) mx = k4[jx] > mx ? k4[jx] : mx;
            if (popc64(wave_ballot(mx != 0)) > G) {
              // every lane ranks its maximum among the 64 (LDS broadcast reads, no serial chain: the search bit by bit
              // that stood here until round 4 was 44 dependent ballots); the keys are unique, so one lane holds rank G-1
              struct alignas(16) K2 { u64 k[2]; };
              const K2* k2 = reinterpret_cast<const K2*>(ckey);
              ckey[lane] = mx;
              wave_sync();
              u32 rk = 0;
#pragma unroll
              for (int z2 = 0; z2 < 32; ++z2) {
                const K2 o2 = k2[z2];
                rk += o2.k[0] > mx ? 1u : 0u;
                rk += o2.k[1] > mx ? 1u : 0u;
              }
              const u64 hit = wave_ballot(mx != 0 && rk == (u32)(G - 1));
              lo = wave_shfl_u64(mx, hit ? __builtin_ctzll(hit) : 0);
              wave_sync();   // (ckey is written again below)
            }
            int c0 = 0;
#pragma unroll
            for (int jx = 0; jx < 4; ++jx) c0 += popc64(wave_ballot(k4[jx] >= lo));
            if (c0 > 64) {
              // (a few lanes hold most of the large keys) the exact threshold, bit by bit over all registers
              lo = 0;
              for (int bit = 63; bit >= 0; --bit) {
                if (!key_bit(bit)) continue;
                const u64 c2 = lo | (u64{1} << bit);
                int cntGe = 0;
#pragma unroll
                for (int jx = 0; jx < 4; ++jx) cntGe += popc64(wave_ballot(k4[jx] >= c2));
                if (cntGe >= G) lo = c2;
              }
            }
          }
          // the survivors (at most 64), one per lane, ranked among themselves (the keys are unique)
          int base = 0;
#pragma unroll
          for (int jx = 0; jx < 4; ++jx) {
            const bool sel = k4[jx] != 0 && k4[jx] >= lo;
            const u64 m = wave_ballot(sel);
            if (sel) ckey[base + popc64(m & ((u64{1} << lane) - 1))] = k4[jx];
            base += popc64(m);
          }
          if (lane >= base) ckey[lane] = 0;
          wave_sync();
          {
            struct alignas(16) K2 { u64 k[2]; };
            const K2* k2 = reinterpret_cast<const K2*>(ckey);
            const u64 me = ckey[lane];
            u32 rank = 0;
            for (int z0 = 0; z0 < base; z0 += 8) {   // (base <= 64 is uniform; eight keys in flight per round; the slots
#pragma unroll                                     //  beyond `base` hold 0, which is greater than no key)
              for (int u = 0; u < 4; ++u) {
                const K2 o2 = k2[z0 / 2 + u];
                rank += o2.k[0] > me ? 1u : 0u;
                rank += o2.k[1] > me ? 1u : 0u;
              }
            }
            if (lane < base && rank < (u32)want) dst[rank] = me;
          }
          wave_sync();
          return want;
        };
        const bool oneChunk = ncand <= 256u;
        int npool = 0;
        for (u32 c0 = 0; c0 < ncand; c0 += 256u) {
          u64 raw[4];
          u64 k4[4];
#pragma unroll
          for (int jx = 0; jx < 4; ++jx) {
            const u32 q = c0 + (u32)lane + 64u * jx;
            const u32 qq = q < ncand ? q : 0;
            const u32 l = small_div(qq, invBeam), k = qq - l * (u32)beam;
            raw[jx] = *reinterpret_cast<const u64*>(&beams[(u64)as_lds(enL)[l] * beam + k]);   // {u16 left, u16 beam, f32 total}
          }
          if (c0 == 0) lds_async_wait();   // invariant (1): this boundary's prefetch() and ring refill were issued before these loads
          JPP_PROF(11);
#pragma unroll
          for (int jx = 0; jx < 4; ++jx) {
            const u32 q = c0 + (u32)lane + 64u * jx;
            const u32 l = small_div(q < ncand ? q : 0, invBeam), k = q - l * (u32)beam;
            const bool fake = (u32)raw[jx] == 0xffffffffu;   // left == beam == 0xffff
            k4[jx] = (q < ncand && !fake) ? (((u64)f32_sortable(__builtin_bit_cast(float, (u32)(raw[jx] >> 32))) << 32) | ((u64)l << 16) | k) : 0;
          }
          npool += select4(k4, oneChunk ? gb_key : pool + npool);
          JPP_PROF(12);
        }
        if (oneChunk) {
          ngb = npool;
        } else {
          u64 k4[4];
#pragma unroll
          for (int jx = 0; jx < 4; ++jx) k4[jx] = ((int)lane + 64 * jx) < npool ? pool[lane + 64 * jx] : 0;
          wave_sync();
          ngb = select4(k4, gb_key);
        }
      } else {
        lds_async_wait();   // invariant (2) above
        ngb = global_beam_from_hbm(beams, en + efirst, ncand, beam, G, gb_key);
      }
    } else {
      if (fastCand) {
        // the candidates' beam slots go straight to LDS (one dwordx4 per slot); they stay there for the winners
        for (u32 q0 = 0; q0 < ncand; q0 += 64) {
          const u32 q = q0 + (u32)lane;
          const u32 l = small_div(q, invBeam), k = q - l * (u32)beam;
          lds_async_load<16>(&cand[q0], &beams[(u64)enL[q < ncand ? l : 0] * beam + k], q < ncand);
        }
        lds_async_wait();   // invariant (1) above: also covers this boundary's prefetch() and ring refill
        wave_sync();
      }
      {
        u64 last = ~u64{0};
        if (fastCand) {
          // every lane keeps its <= 4 candidate keys in registers
          u64 mykey[kCandCap / 64];
  #pragma unroll
          for (int jx = 0; jx < kCandCap / 64; ++jx) {
            u32 q = (u32)lane + 64u * jx;
            u64 key = 0;
            if (q < ncand) {
              u32 l = small_div(q, invBeam), k = q - l * (u32)beam;
              BeamSlot sl = cand[q];
              if (!slot_fake(sl)) key = ((u64)f32_sortable(sl.total) << 32) | ((u64)l << 16) | k;
            }
            mykey[jx] = key;
          }
          if (ncand <= 64u) {
            // the keys are unique, so the global beam is "every key with fewer than G larger ones":
            // each lane ranks its own key against the others (LDS broadcast reads)
            ckey[lane] = mykey[0];
            wave_sync();
            const u64 me = mykey[0];
            u32 rank = 0;
            for (u32 z = 0; z < ncand; ++z) rank += ckey[z] > me ? 1u : 0u;
            const int live = popc64(wave_ballot(me != 0));
            ngb = live < G ? live : G;
            if (me != 0 && rank < (u32)G) gb_key[rank] = me;
          } else {
            // More than 64 candidates (wide beams): the G-th largest key by a bitwise threshold search -- per
            // bit one compare per register-resident key and a ballot count, no cross-lane reduction chain --
            // then the selected keys are compacted and ranked among themselves (the keys are unique).
            int live = 0;
  #pragma unroll
            for (int jx = 0; jx < kCandCap / 64; ++jx) live += popc64(wave_ballot(mykey[jx] != 0));
            ngb = live < G ? live : G;
            u64 thr = 1;   // every live key is >= 1
            if (live > G) {
              thr = 0;
              // bits that can be set in a key: 63..32 total, 16 + log2(kEnnCap).. 16 left index, 4..0 (beam <= 32) slot
              for (int bit = 63; bit >= 0; --bit) {
                if (bit < 32 && !((bit >= 16 && bit <= 22) || bit <= 5)) continue;
                const u64 c2 = thr | (u64{1} << bit);
                int cntGe = 0;
  #pragma unroll
                for (int jx = 0; jx < kCandCap / 64; ++jx) cntGe += popc64(wave_ballot(mykey[jx] >= c2));
                if (cntGe >= G) thr = c2;
              }
            }
            // compaction: position = number of selected keys in earlier registers / lower lanes
            int base = 0;
  #pragma unroll
            for (int jx = 0; jx < kCandCap / 64; ++jx) {
              const bool sel = mykey[jx] != 0 && mykey[jx] >= thr;
              const u64 m = wave_ballot(sel);
              if (sel) ckey[base + popc64(m & ((u64{1} << lane) - 1))] = mykey[jx];
              base += popc64(m);
            }
            wave_sync();
            if (lane < ngb) {
              const u64 me = ckey[lane];
              u32 rank = 0;
              for (int z = 0; z < ngb; ++z) rank += ckey[z] > me ? 1u : 0u;
              gb_key[rank] = me;
            }
          }
        } else {
          lds_async_wait();   // invariant (2) above
          (void)last;
          ngb = global_beam_from_hbm(beams, en + efirst, ncand, beam, G, gb_key);
        }
      }

    }
    wave_sync();
    ngb = uni(ngb);
    JPP_PROF(13);
    if (lane < ngb) {
      u64 key = gb_key[lane];
      u32 l = (u32)(key >> 16) & 0xffff, k = (u32)key & 0xffff;
      u32 lnode;
      u32 pnode;
      if (fastCand && GM <= 8) {
        lnode = enL[l];
        pnode = as_lds(cand)[l * (u32)beam + k].prev_node;   // (a plain ds_read: see as_lds)
      } else if (fastCand) {
        lnode = as_lds(enL)[l];
        pnode = beams[(u64)lnode * beam + k].prev_node;   // (wide variant: one gather for the <= 32 winners)
      } else {
        lnode = en[efirst + l];
        pnode = beams[(u64)lnode * beam + k].prev_node;
      }
      gb_left[lane] = (u16)l;
      gb_slot[lane] = (u16)k;
      gb_score[lane] = sortable_f32((u32)(key >> 32));
      gb_lnode[lane] = lnode;
      gb_pnode[lane] = pnode;
      B.bnd_gbeam[(u64)(bb0 + b) * G + lane] = GbeamEntry{(u16)l, (u16)k, gb_score[lane]};
    }
    if (lane == 0) B.bnd_ngb[bb0 + b] = (u32)ngb;
    wave_sync();

    if (ngb == 0) {
      // unreachable boundary: every right node gets an all-fake beam (makeT0Beam with an empty gbeam)
      for (u32 q = lane; q < R * (u32)beam; q += 64) {
        beams[(u64)rfirst * beam + q] = BeamSlot{kFake16, kFake16, 0.f, 0xffffffffu, 0};
      }
      for (u32 q = lane; q < R; q += 64) B.node_kept[nb + rfirst + q] = 0;
      if constexpr (kOneRow) prefetch_rows(bn, mbn, 0);   // this boundary's rows are not needed
      lds_async_wait();   // invariant (3) above
      wave_sync();
      continue;
    }

    JPP_PROF(1);
    // ---- 2. T1 dedup in first-seen order, gather T1 / T2 pattern rows ----
    int U;
    {
      // first occurrence of every left node among the gbeam entries; the unique T1 rows are numbered in
      // first-seen order exactly like the sequential dedupT1
      int first = lane;
      if (lane < ngb) {
        const u16 left = gb_left[lane];
        for (int j = 0; j < lane; ++j) {
          if (gb_left[j] == left) {
            first = j;
            break;
          }
        }
      }
      const bool isFirst = lane < ngb && first == lane;
      const u64 fmask = wave_ballot(isFirst);
      U = popc64(fmask);
      if (lane < ngb) {
        const u32 u = (u32)popc64(fmask & ((u64{1} << first) - 1));
        gb_t1[lane] = u;
        if (isFirst) t1node[u] = gb_lnode[lane];
      }
    }
    wave_sync();
    // (Requesting these rows into registers and computing the first-stage hash states of the right nodes while
    // they travel was measured in round 2: 6.92 -> 7.11 ms.  The extra live registers cost more than the
    // overlapped round trip saves at 128 VGPRs.)
    // All row elements of a pass are requested before the first one is stored: written as one loop with a T1 and a
    // T2 branch this was two divergent halves per iteration, each waiting for its own HBM/L2 round trip (three to
    // four serial round trips per boundary instead of one).
    {
      // Row elements by SLOT, a slot = one load per lane: a T1 slot covers four rows with 16 lanes each (kPat = 14 of them
      // busy), a T2 slot 64 / kT2w rows with kT2w lanes each -- the (row, field) of a lane is a shift and a mask.  (Until
      // round 3 the elements were numbered consecutively: two divisions by 14 and by 4 per element and load, a tenth of the
      // boundary's vector instructions.)  Up to kRowIter slots are in flight together.
      constexpr int kRowIter = 3;                 // GM = 8: two T1 slots + one T2 slot = one round
      constexpr int kT2w = kT2 <= 4 ? 4 : 16;     // lanes per T2 row
      constexpr int kT2rows = 64 / kT2w;          // T2 rows per slot
      static_assert(kPat <= 16 && kT2 <= kT2w, "a row fits its lane group");
      const int nT1s = (U + 3) >> 2;
      const int nSlots = nT1s + (ngb + kT2rows - 1) / kT2rows;
      for (int s0 = 0; s0 < nSlots; s0 += kRowIter) {
        u64 v[kRowIter];
#pragma unroll
        for (int z = 0; z < kRowIter; ++z) {
          const int sl = s0 + z;
          const bool isT1 = sl < nT1s;
          const int row = isT1 ? sl * 4 + (lane >> 4) : (sl - nT1s) * kT2rows + lane / kT2w;
          const int pp = isT1 ? (lane & 15) : (lane & (kT2w - 1));
          const bool ok = sl < nSlots && (isT1 ? (row < U && pp < kPat) : (row < ngb && pp < kT2));
          // (lanes without an element read element 0 again: a load behind a branch would make the compiler wait for the
          // previous one before it, see k_rnn_chain)
          const u32 node = ok ? (isT1 ? t1node[row] : gb_pnode[row]) : t1node[0];
          v[z] = pats[(u64)node * kPat + (ok ? pp : 0)];
        }
#pragma unroll
        for (int z = 0; z < kRowIter; ++z) {
          const int sl = s0 + z;
          const bool isT1 = sl < nT1s;
          const int row = isT1 ? sl * 4 + (lane >> 4) : (sl - nT1s) * kT2rows + lane / kT2w;
          const int pp = isT1 ? (lane & 15) : (lane & (kT2w - 1));
          if (sl < nSlots) {
            if (isT1) {
              if (row < U && pp < kPat) t1pat[row][pp] = v[z];
            } else {
              if (row < ngb && pp < kT2) t2pat[row][pp] = v[z];
            }
          }
        }
      }
    }
    wave_sync();

    JPP_PROF(2);
    // ---- 3. prescores for the first c gbeam entries over all right nodes ----
    int c = rcheck;
    if (c > (int)R) c = (int)R;
    if (c > ngb) c = ngb;
    if (RM > 0 && (u32)c * R > (u32)(2 * RM)) {
      if (lane == 0) B.sent_status[s] = ST_CAPACITY;
      return;
    }
    for (u32 tc = 0; tc < R && c > 0; tc += kChunk) {
      const u32 nx = (R - tc) < (u32)kChunk ? (R - tc) : (u32)kChunk;
      const float* t0c = tc == 0 ? t0n[par] : t0R;
      if (tc != 0) {
        for (u32 q = lane; q < nx * kPat; q += 64) pR[q / kPat][q % kPat] = pats[(u64)(rfirst + tc) * kPat + q];
        if ((u32)lane < nx) t0R[lane] = t0s[rfirst + tc + lane];
        wave_sync();
      }
      compute_s1(tc == 0 ? pRn[kOneRow ? 0 : par] : pR, nx);
      wave_sync();
      // kOneRow: the rows are dead now.  With a single pass per phase the states of phase 3 serve phase 5 as well and
      // the buffer is free for the next boundary's rows; otherwise the later passes re-stage rows in it first.
      if constexpr (kOneRow) {
        if (R <= (u32)kChunk) prefetch_rows(bn, mbn, 0);
      }
      for (int i = 0; i < c; ++i) {
        const bool act = (u32)grp < nx;
        const u32 t = tc + (u32)grp;
        const int xr = act ? grp : 0;
        const u64* t1r = t1pat[gb_t1[i]];
        const u64* t2r = t2pat[i];
        float w[kBiPerLane];
        bi_gather_s1<W24>(lbi, gj, s1b[xr], t1r, W, wmask, act, w, nBi);
        float g = 0.f;
        if (act && gj < nTri) {
          u32 idx = hmix_index<W24>(hmix(s1t[xr][gj], t1r[s_trit[gj][1]]), t2r[s_trit[gj][2]], wmask);
          g += W[idx];
        }
        // generated applyBiStep2 (8 round-robin sums; last right node: unrolled-4) and applyTriStep3
        const float b8 = DYN ? 0.f : bi_sum8(w, lane, gj, nBi);
        const float b4 = bi_sum4(w, lane, gj, nBi);
        if constexpr (kHeadShare) {
          // The tail (5a) adds up the same 37 weights of (right node, T1 row 0) again, only in the
          // association of applyBiTriFullKernel: form that sum here, from the weights already gathered,
          // and 5a skips row 0 (with one head entry, c == 1, its row is always row 0).
          const float tailS = (U == 1) ? b4 : bi_sum2(w, lane, gj, nBi);   // row 0 is the last T1 row iff U == 1
          if (act && gj == 0) biS0[t] = tailS;
        }
        static_assert(spec::kNumTri == 4, "the trigram sum below reads group members 1..3");
        float tsum = g;
        tsum += row_shl_f32<1>(g);
        tsum += row_shl_f32<2>(g);
        tsum += row_shl_f32<3>(g);
        if (act && gj == 0) {
          float sc = t0c[grp];
          sc += (DYN || t == R - 1) ? b4 : b8;   // (DYN: computeUnrolled4RawPerceptron for every row)
          sc += tsum;
          // applyPluginToPrescores (score_processor.cc:578-596)
          if (B.node_penalty) sc -= B.node_penalty[nb + rfirst + t];
          // a per-connection plugin: the amount of (left node of head entry i, right node t)
          if (B.pair_penalty) sc -= B.pair_penalty[B.pair_base[bb0 + b] + (u64)gb_left[i] * R + t];
          pres[(u32)i * R + t] = sc;
        }
      }
      wave_sync();
    }

    JPP_PROF(3);
    // ---- 4. right-node cutoff (std::nth_element semantics) ----
    const u32 K = (rcheck > 0) ? ((u32)rbeam < R ? (u32)rbeam : R) : R;
    if (!(rcheck > 0 && R > (u32)rbeam)) {
      for (u32 t = lane; t < R; t += 64) order[t] = (u16)t;   // no cutoff: natural order
    } else {
      // (the rank pass below writes every entry of `order`: a strict total order gives a permutation)
      if constexpr (!kLean) {   // (kLean: csum is pres, see the declarations)
        for (u32 t = lane; t < R; t += 64) {
          float sc = 0.f;
          for (int i = 0; i < c; ++i) sc += pres[i * R + t];
          csum[t] = sc;
        }
        wave_sync();
      }
      // Fast path: only the SET of the first rbeam entries matters downstream (kept nodes are scored
      // independently).  If no tie straddles the cut, that set is the unique top-rbeam by score and a
      // parallel stable rank gives it; otherwise replay std::nth_element step by step on one lane.
      for (u32 t = lane; t < R; t += 64) {
        float me = csum[t];
        u32 rank = 0;
        for (u32 u = 0; u < R; ++u) {
          float o = csum[u];
          rank += (o > me || (o == me && u < t)) ? 1u : 0u;
        }
        order[rank] = (u16)t;
      }
      wave_sync();
      const bool tieAtCut = csum[order[rbeam - 1]] == csum[order[rbeam]];
      wave_sync();
      if (tieAtCut) {
        for (u32 t = lane; t < R; t += 64) order[t] = (u16)t;
        wave_sync();
        if (lane == 0) cutoff_replay(order, csum, (u32)rbeam, R);
      }
    }
    wave_sync();

    JPP_PROF(4);
    // ---- 5. score + beams, kChunk right nodes at a time in cutoff order ----
    const int ntail = ngb - c;
    const u32 invNgb = small_div_inv((u32)ngb);   // (ngb <= 32, lane indices < 2048)
    for (u32 op0 = 0; op0 < R; op0 += kChunk) {
      const int nx = (int)((R - op0) < (u32)kChunk ? (R - op0) : (u32)kChunk);
      // patterns / T0 of this pass's right nodes in cutoff order: rows of the prefetched buffer when the
      // whole boundary fits one chunk, staged from HBM otherwise
      const bool small = R <= (u32)kChunk;
      if (!small) {
        for (int q = lane; q < nx * kPat; q += 64) pR[q / kPat][q % kPat] = pats[(u64)(rfirst + order[op0 + q / kPat]) * kPat + q % kPat];
        if (lane < nx) t0R[lane] = t0s[rfirst + order[op0 + lane]];
        wave_sync();
      }
      // first-stage states: in the small case they are those of the prescore pass (natural node order,
      // addressed through the cutoff order); otherwise they are rebuilt for the rows just staged
      if (!small) {
        compute_s1(pR, (u32)nx);
        wave_sync();
        // kOneRow: after the last pass the row buffer is free for the next boundary
        if constexpr (kOneRow) {
          if (op0 + (u32)kChunk >= R) prefetch_rows(bn, mbn, 0);
        }
      } else if (c == 0) {
        static_assert(!kLean || DEF, "kLean: c >= 1 whenever a boundary is scored, the rows are gone by now");
        compute_s1(pRn[kOneRow ? 0 : par], (u32)nx);
        wave_sync();
      }
      auto s1Row = [&](int x) -> int { return small ? (int)order[op0 + x] : x; };
      auto t0Of = [&](int x) -> float { return small ? t0n[par][order[op0 + x]] : t0R[x]; };
      // The trigram weights of 5b are requested before the bigram pass 5a, so that both sets of gathers are in flight
      // together instead of one HBM round trip after the other.  Narrow variant: nx * ngb <= 64, one (node, entry) per
      // lane.  Wide variant (round 4): up to kChunk * GM / 64 = 4 per lane, 16 gathers in flight (until then 5b ran
      // its items one after the other, four gathers and a full round trip each).  The lane that requests a weight is
      // the lane that sums it in 5b.
      constexpr int kItems = GM <= 8 ? 1 : kChunk * GM / 64;
      static_assert(kChunk * GM <= 64 * kItems, "kItems (node, entry) pairs per lane cover a pass");
      float wtri[kItems][spec::kNumTri];
#pragma unroll
      for (int jx = 0; jx < kItems; ++jx) {
#pragma unroll
        for (int f = 0; f < spec::kNumTri; ++f) wtri[jx][f] = 0.f;
        const int q = lane + 64 * jx;
        if (q < nx * ngb) {
          const int x = (int)small_div((u32)q, invNgb), i = q - x * ngb;
          if (i >= c && (op0 + x) < K) {
            const u64* st = s1t[s1Row(x)];
            const u64* t1r = t1pat[gb_t1[i]];
            const u64* t2r = t2pat[i];
#pragma unroll
            for (int f = 0; f < spec::kNumTri; ++f) {
              if (DYN && f >= nTri) continue;
              const int i1 = DYN ? (int)s_trit[f][1] : kNg.tri_t1[f], i2 = DYN ? (int)s_trit[f][2] : kNg.tri_t2[f];
              u32 idx = hmix_index<W24>(hmix(st[f], t1r[i1]), t2r[i2], wmask);
              wtri[jx][f] = W[idx];
            }
          }
        }
      }
      // 5a. bigram sums per (kept node, unique T1 row) -- applyBiTriFullKernel rows
      if (ntail > 0) {
        // (kHeadShare: row 0 came out of the prescore pass, the units cover rows 1 .. U-1)
        const int Ur = kHeadShare ? U - 1 : U;
        const u32 invUr = small_div_inv((u32)Ur);
        const int units = nx * Ur;
        // (wide variant: two rounds of units per iteration, their gathers in flight together)
        constexpr int kRounds = GM <= 8 ? 1 : 2;
        for (int base = 0; base < units; base += 8 * kRounds) {
          float w[kRounds][kBiPerLane];
          int xs[kRounds], tus[kRounds];
          bool acts[kRounds];
#pragma unroll
          for (int r = 0; r < kRounds; ++r) {
            int u = base + 8 * r + grp;
            bool act = u < units;
            int x = act ? (int)small_div((u32)u, invUr) : 0, tu = act ? u - x * Ur : 0;
            if (kHeadShare) tu += 1;
            act = act && (op0 + x) < K;
            bi_gather_s1<W24>(lbi, gj, s1b[s1Row(x)], t1pat[tu], W, wmask, act, w[r], nBi);
            xs[r] = x;
            tus[r] = tu;
            acts[r] = act;
          }
#pragma unroll
          for (int r = 0; r < kRounds; ++r) {
            const float s2 = DYN ? 0.f : bi_sum2(w[r], lane, gj, nBi);
            const float s4 = bi_sum4(w[r], lane, gj, nBi);
            if (acts[r] && gj == 0) biS[xs[r]][tus[r]] = (DYN || tus[r] == U - 1) ? s4 : s2;
          }
        }
      }
      wave_sync();
      JPP_PROF(5);
      // 5b. cells and totals per (node, gbeam entry)
#pragma unroll
      for (int jx = 0; jx < kItems; ++jx) {
        const int q = lane + 64 * jx;
        if (q >= nx * ngb) continue;
        int x = (int)small_div((u32)q, invNgb), i = q - x * ngb;
        bool kept = (op0 + x) < K;
        u32 t = order[op0 + x];
        float cell, total;
        bool defined = true;
        if (i < c) {
          // copyT0Scores(head, result, 0): v += 0; cell = v; v += gb.score()
          float v = pres[i * R + t];
          v += 0.f;
          cell = v;
          v += gb_score[i];
          total = v;
        } else if (kept) {
          float w[spec::kNumTri];
#pragma unroll
          for (int f = 0; f < spec::kNumTri; ++f) w[f] = wtri[jx][f];   // requested before 5a
          static_assert(spec::kNumTri == 4, "trigram association below is written for 4 features");
          float S;
          if (kHeadShare && gb_t1[i] == 0) S = biS0[t];
          else S = biS[x][gb_t1[i]];
          float res;
          if (!DYN && i < ngb - 1) {   // (DYN: every row through computeUnrolled4RawPerceptron)
            float r1 = 0.f, r2 = 0.f;
            r1 += w[0];
            r2 += w[1];
            r1 += w[2];
            r2 += w[3];
            res = S + r1 + r2;
          } else {
            float q1 = 0.f, q2 = 0.f, q3 = 0.f, q4 = 0.f;
            q1 += w[0];
            q2 += w[1];
            q3 += w[2];
            q4 += w[3];
            res = S + (q1 + q2 + q3 + q4);
          }
          // applyPluginToGbeam (score_processor.cc:598-613), then copyT0Scores(tail, resultTail, t0Score)
          float v = res;
          if (B.node_penalty) v -= B.node_penalty[nb + rfirst + t];
          if (B.pair_penalty) v -= B.pair_penalty[B.pair_base[bb0 + b] + (u64)gb_left[i] * R + t];
          v += t0Of(x);
          cell = v;
          v += gb_score[i];
          total = v;
        } else {
          defined = false;
          cell = 0.f;
          total = 0.f;
        }
        tot[x][i] = total;
        if (defined) B.node_cells[((nb + rfirst + t) * G + i) * cfg.nscorers] = cell;
      }
      wave_sync();
      JPP_PROF(6);
      // 5c. beams: stable descending rank among the node's candidates (makeT0Beam; for <= 16
      //     candidates std::sort is an insertion sort, i.e. stable)
      const int partB = beam * 4 / 3;  // makeT0Beam: partitionBoundary
      if constexpr (GM > 16) {
        // Wide variant (the host sends every configuration with more than 16 candidates or a global beam
        // above beam*4/3 here; the narrow one stays free of scratch).  Beyond 16 candidates libstdc++'s
        // std::sort is an introsort (not stable), beyond beam*4/3 util::partition runs first
        // (score_processor.cc:434-437).  Both only permute: when the node's candidate totals are pairwise
        // distinct, the front util::partition leaves is the true top-sz for some sz >= beam and the sorted
        // order is the unique descending one, so the parallel rank below IS makeT0Beam's result.  Only a
        // node that holds two equal totals replays the two algorithms step by step on one lane.
        static_assert(GM == 32, "one half-wave per right node");
        // Three passes over the nodes of the chunk (round 4; until then the replay ran inside the rank loop, two nodes
        // per iteration, i.e. up to four serial replays one after the other per chunk): A ranks and ties, nodes without
        // a tie write their beams; B the nodes that tie replay the partitioning steps, one half-wave per node (B1) or,
        // when util::partition comes first, one lane per node (B2); C the parallel stable rank of the partitioned keys.
        for (int q0 = 0; q0 < nx * GM; q0 += 64) {
          const int q = q0 + lane;
          const bool in = q < nx * GM;
          const int x = in ? q / GM : 0, i = in ? q - x * GM : GM;
          const bool kept = (op0 + x) < K;
          const u32 t = order[op0 + x];
          const int cnt = kept ? ngb : c;
          BeamSlot* row = beams + (u64)(rfirst + t) * beam;
          const bool wide = cnt > 16 || cnt > partB;
          float me = 0.f;
          int rank = 0;
          bool tie = false;
          {
            // every lane goes over all GM totals of its node, four per LDS read, whatever its own count, and keeps the
            // two comparisons as bit masks by position -- no branch, no short-circuit (round 4: a loop up to `cnt` is
            // divergent and is not unrolled: 32 dependent LDS round trips per lane and pass, a third of the kernel on
            // the configs[4] shape, profiles/r04_v_phases5.txt).  Positions beyond the count hold stale totals: masked.
            struct alignas(16) F4 { float v[4]; };
            const F4* t4 = reinterpret_cast<const F4*>(tot[x]);
            me = tot[x][i < GM ? i : 0];
            u32 gtm = 0, eqm = 0;
#pragma unroll
            for (int j4 = 0; j4 < GM / 4; ++j4) {
              const F4 o4 = t4[j4];
#pragma unroll
              for (int u = 0; u < 4; ++u) {
                const u32 bit = 1u << (j4 * 4 + u);
                gtm |= o4.v[u] > me ? bit : 0u;
                eqm |= o4.v[u] == me ? bit : 0u;
              }
            }
            const u32 cm = cnt >= 32 ? 0xffffffffu : ((1u << cnt) - 1u);
            const u32 below = i >= 32 ? 0xffffffffu : ((1u << i) - 1u), self = i >= 32 ? 0u : (1u << i);
            rank = __builtin_popcount(gtm & cm) + __builtin_popcount(eqm & below & cm);
            tie = i < cnt && (eqm & cm & ~self) != 0;
          }
          const u64 tb = wave_ballot(wide && tie);
          const bool replay = wide && ((tb >> (lane & 32)) & 0xffffffffull) != 0;
          if (in && i == 0) sreplay[x] = replay ? (cnt > partB ? 2 : 1) : 0;   // 2: util::partition first (serial lane)
          if (!replay) {
            if (i < cnt) {
              if (rank < beam) row[rank] = BeamSlot{gb_left[i], gb_slot[i], me, gb_lnode[i], (u32)i};
            } else if (i < beam) {
              row[i] = BeamSlot{kFake16, kFake16, 0.f, 0xffffffffu, 0};
            }
          }
          if (i == 0) B.node_kept[nb + rfirst + t] = kept ? 1 : 0;
        }
        wave_sync();
        JPP_PROF(8);
        // B. Ties: the partitioning steps of the reference's sort (util::partition beyond beam*4/3, then the Hoare
        // partitions of introsort while a range is longer than 16), replayed step by step.  What follows in std::sort is
        // libstdc++'s final insertion pass, a STABLE sort of the array as partitioned -- i.e. a rank again: greater
        // totals first, equal totals in their order after partitioning (std_sort_partition_only, jpp_select.h).
        //
        // B1 (nodes without util::partition, i.e. at most beam*4/3 candidates): one HALF-WAVE per node, lane = position
        // in the array, the keys in registers.  A Hoare partition around the median-of-three pivot is then a handful of
        // wave operations instead of a serial scan through LDS: the pivot's value comes from three shuffles; ONE pair
        // of ballots gives, by position, the elements the upward scan stops at (not greater than the pivot) and the ones
        // the downward scan stops at (not less); which positions the loop exchanges follows from the two masks alone
        // (below), and all exchanges of the partition -- every position takes part in at most one -- are applied as
        // one permutation by a single shuffle at the end.  Element for element the sequence of util / libstdc++ steps
        // (sel_move_median_to_first, sel_unguarded_partition, std_sort_partition_only_le32 in jpp_select.h).
        for (int x0 = 0; x0 < nx; x0 += 2) {
          const int x = x0 + (lane >> 5);
          const int hl = lane & 31, hb = lane & 32;
          const bool act0 = x < nx && sreplay[x < nx ? x : 0] == 1;
          if (wave_ballot(act0) == 0) continue;
          const int cnt = act0 ? (((op0 + x) < K) ? ngb : c) : 0;
          float v = 0.f;
          u32 idx = (u32)hl;
          if (hl < cnt) v = tot[x][hl];
          int f = 0, l = cnt;
          int depth = 0;
          while ((cnt >> (depth + 1)) != 0) ++depth;
          depth *= 2;
          bool act = act0, fail = false;
          for (;;) {
            bool run = act && (l - f > 16);
            if (run && depth == 0) {   // introsort's heap-sort fallback: the serial lane replays all of it
              fail = true;
              run = false;
              act = false;
            }
            if (wave_ballot(run) == 0) break;
            --depth;
            const int a = f + 1, mid = f + (l - f) / 2, cc = l - 1;
            const float va = wave_shfl_f32(v, run ? hb + a : lane);
            const float vb = wave_shfl_f32(v, run ? hb + mid : lane);
            const float vc = wave_shfl_f32(v, run ? hb + cc : lane);
            const float vf = wave_shfl_f32(v, run ? hb + f : lane);
            int m;   // the median of the three goes to the front (comp(x, y) = x > y)
            if (va > vb) m = vb > vc ? mid : va > vc ? cc : a;
            else m = va > vc ? a : vb > vc ? cc : mid;
            const float pv = m == a ? va : m == mid ? vb : vc;
            const float myv = hl == f ? pv : hl == m ? vf : v;
            // The scans of sel_unguarded_partition in closed form.  a_1 < a_2 < ... : the positions from f + 1 up whose
            // element is not greater than the pivot (where the upward scan stops); b_1 > b_2 > ... : the positions from
            // l - 1 down to f whose element is not less (where the downward scan stops; f holds the pivot itself).  The
            // k-th round of the loop exchanges a_k and b_k as long as a_k < b_k: in between, the array is untouched, and
            // an exchanged position stops the other scan (it now holds an element from the other side), so round k + 1
            // ends at min(a_{k+1}, b_k) and max(b_{k+1}, a_k) -- which cross exactly when a_{k+1} >= b_{k+1}.  With K
            // exchanges the function returns a_1 (K = 0) or min(a_{K+1}, b_K).  Every lane finds its own rank among the
            // a's / b's with a population count and its partner through two small tables in LDS: no loop.
            const u32 range = l >= 32 ? 0xffffffffu : ((1u << l) - 1u);
            const u32 ngm = (u32)(wave_ballot(!(myv > pv)) >> hb) & range & ~((2u << f) - 1u);
            const u32 nlm = (u32)(wave_ballot(!(pv > myv)) >> hb) & range & ~((1u << f) - 1u);
            const int nL = __builtin_popcount(ngm), nR = __builtin_popcount(nlm);
            const bool isL = run && ((ngm >> hl) & 1u) != 0, isR = run && ((nlm >> hl) & 1u) != 0;
            const int kL = __builtin_popcount(ngm & ((2u << hl) - 1u));   // 1-based, from the bottom
            const int kR = __builtin_popcount(nlm >> hl);                 // 1-based, from the top
            u8* const pl = hpart[lane >> 5][0];
            u8* const pr = hpart[lane >> 5][1];
            if (isL) pl[kL] = (u8)hl;
            if (isR) pr[kR] = (u8)hl;
            wave_sync();
            int src = hl;
            bool partL = false;
            if (isL && kL <= nR) {
              const int bk = pr[kL];
              if (hl < bk) {
                src = bk;
                partL = true;
              }
            }
            if (!partL && isR && kR <= nL) {
              const int ak = pl[kR];
              if (ak < hl) src = ak;
            }
            const int Kx = __builtin_popcount((u32)(wave_ballot(partL) >> hb));
            if (run) {
              int cut;
              if (Kx == 0) {
                cut = ngm ? __builtin_ctz(ngm) : l;
              } else {
                const int aN = Kx + 1 <= nL ? (int)pl[Kx + 1] : 64;
                const int bK = pr[Kx];
                cut = aN < bK ? aN : bK;
              }
              src = src == f ? m : src == m ? f : src;   // the pivot's exchange came first
              if (l - cut > 16) f = cut;   // the right part is the long one
              else l = cut;
            }
            wave_sync();   // (the tables are written again by the next partition)
            v = wave_shfl_f32(v, hb + src);
            idx = wave_shfl_u32(idx, hb + src);
          }
          if (act0 && fail) {
            if (hl == 0) sreplay[x] = 2;
          } else if (act0) {
            if (hl < cnt) {
              u32 bits;
              __builtin_memcpy(&bits, &v, 4);
              skey[x][hl] = ((u64)bits << 32) | idx;
            }
            if (hl == 0) shave[x] = (u8)cnt;
          }
        }
        wave_sync();
        JPP_PROF(9);
        // B2 (util::partition first, or the depth limit): ONE LANE per node
        if (lane < nx && sreplay[lane] == 2) {
          const int x = lane;
          const bool kept = (op0 + x) < K;
          const int cnt = kept ? ngb : c;
          u64* keys = skey[x];   // in LDS: as a private array it would live in scratch (HBM latency per access)
          const float* tr = tot[x];
          // the reference sorts indices with `scores[i1] > scores[i2]`: the same predicate on the packed totals
          auto comp = [](u64 a, u64 bb) {
            const u32 xa = (u32)(a >> 32), xb = (u32)(bb >> 32);
            float fa, fb;
            __builtin_memcpy(&fa, &xa, 4);
            __builtin_memcpy(&fb, &xb, 4);
            return fa > fb;
          };
          auto fill = [&]() {
            for (int z = 0; z < cnt; ++z) {
              u32 bits;
              __builtin_memcpy(&bits, &tr[z], 4);
              keys[z] = ((u64)bits << 32) | (u32)z;
            }
          };
          fill();
          u64* itr = keys + cnt;
          if (cnt > partB) itr = jpp_partition(keys, itr, comp, (long)beam, (long)partB);
          bool sorted = false;
          static_assert(GM <= 32, "stackless partition replay covers at most 32 candidates");
          if (!std_sort_partition_only_le32(keys, itr, comp)) {
            // depth limit of introsort hit (heap-sort fallback, not stable): replay all of it from the start
            fill();
            itr = keys + cnt;
            if (cnt > partB) itr = jpp_partition(keys, itr, comp, (long)beam, (long)partB);
            std_sort(keys, itr, comp);
            sorted = true;
          }
          shave[x] = (u8)((itr - keys) | (sorted ? 0x80 : 0));
        }
        wave_sync();
        JPP_PROF(10);
        // C. the replayed nodes' beams
        for (int q0 = 0; q0 < nx * GM; q0 += 64) {
          const int q = q0 + lane;
          const bool in = q < nx * GM;
          const int x = in ? q / GM : 0, i = in ? q - x * GM : GM;
          if (!in || sreplay[x] == 0) continue;
          const u32 t = order[op0 + x];
          BeamSlot* row = beams + (u64)(rfirst + t) * beam;
          const int have = shave[x] & 0x7f;
          const bool sorted = (shave[x] & 0x80) != 0;
          if (i < have) {
            const u64 mine = skey[x][i];
            const u32 vb = (u32)(mine >> 32);
            float mv;
            __builtin_memcpy(&mv, &vb, 4);
            int pos = i;
            if (!sorted) {
              pos = 0;
              struct alignas(16) K2 { u64 k[2]; };
              const K2* k2 = reinterpret_cast<const K2*>(skey[x]);
              u32 gtm = 0, eqm = 0;
#pragma unroll
              for (int j2 = 0; j2 < GM / 2; ++j2) {   // fixed trip count, two keys per LDS read, masks by position (see pass A)
                const K2 o2 = k2[j2];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                  const u32 bit = 1u << (j2 * 2 + u);
                  const float ov = __builtin_bit_cast(float, (u32)(o2.k[u] >> 32));
                  gtm |= ov > mv ? bit : 0u;
                  eqm |= ov == mv ? bit : 0u;
                }
              }
              const u32 hm = have >= 32 ? 0xffffffffu : ((1u << have) - 1u);
              pos = __builtin_popcount(gtm & hm) + __builtin_popcount(eqm & hm & ((1u << i) - 1u));
            }
            const u32 iz = (u32)mine & 0xffu;
            if (pos < beam) row[pos] = BeamSlot{gb_left[iz], gb_slot[iz], mv, gb_lnode[iz], iz};
          }
          // slots beyond the sorted range stay fake (have >= beam whenever util::partition ran)
          if (i >= have && i < beam) row[i] = BeamSlot{kFake16, kFake16, 0.f, 0xffffffffu, 0};
        }
      } else {
        for (int q = lane; q < nx * GM; q += 64) {
          int x = q / GM, i = q - x * GM;
          bool kept = (op0 + x) < K;
          u32 t = order[op0 + x];
          int cnt = kept ? ngb : c;
          BeamSlot* row = beams + (u64)(rfirst + t) * beam;
          if (i < cnt) {
            float me = tot[x][i];
            int rank = 0;
            for (int jx = 0; jx < cnt; ++jx) {
              float o = tot[x][jx];
              if (o > me || (o == me && jx < i)) ++rank;
            }
            if (rank < beam) row[rank] = BeamSlot{gb_left[i], gb_slot[i], me, gb_lnode[i], (u32)i};
          } else if (i < beam) {
            row[i] = BeamSlot{kFake16, kFake16, 0.f, 0xffffffffu, 0};
          }
          if (i == 0) B.node_kept[nb + rfirst + t] = kept ? 1 : 0;
        }
      }
      wave_sync();
      JPP_PROF(7);
    }
  }
  JPP_PROF_FLUSH;
}

// top-1 path: follow the EOS beam's best slot back to BOS (AnalysisPath::fillIn,
// src/core/analysis/analysis_result.cc:25-76).  One lane per sentence.
__global__ void k_path(Batch B, Config cfg) {
  u32 s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= B.n_sent) return;
  B.path_len[s] = 0;
  if (B.sent_status[s] != ST_OK) return;
  u32 N = B.sent_nodes[s];
  u64 nb = B.node_base[s];
  const BeamSlot* beams = B.node_beam + nb * cfg.beam;
  u32* out = B.path_nodes + nb;
  u32 node = N - 1;
  u32 slot = 0;
  u32 len = 0;
  while (node >= 2 && len < N) {
    BeamSlot sl = beams[(u64)node * cfg.beam + slot];
    if (slot_fake(sl)) break;
    out[len++] = node;
    node = sl.prev_node;
    slot = sl.beam;
  }
  B.path_len[s] = len;
}

// Packed top-1 results for the host / for the cross-GPU gather: per sentence the
// morphemes of the best path in text order (EOS dropped), 8 bytes each.
__global__ void k_pack_count(Batch B, u32* counts) {
  u32 s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= B.n_sent) return;
  u32 pl = B.sent_status[s] == ST_OK ? B.path_len[s] : 0;
  counts[s] = pl > 0 ? pl - 1 : 0;
}

__global__ void k_pack_write(Batch B, const u64* offs, u32* out_off, NodeInfo* items, u64 cap) {
  u32 s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s > B.n_sent) return;
  u64 o = offs[s];
  out_off[s] = (u32)o;
  if (s == B.n_sent) return;
  u32 pl = B.sent_status[s] == ST_OK ? B.path_len[s] : 0;
  if (pl <= 1) return;
  u64 nb = B.node_base[s];
  for (u32 k = 0; k + 1 < pl; ++k) {
    // path_nodes is EOS first; emit first morpheme first
    u32 node = B.path_nodes[nb + (pl - 1 - k)];
    if (o + k < cap) items[o + k] = B.node_info[nb + node];
  }
}

// jppgpu_result_fetch(JPPGPU_FETCH_TOP1): the node and UNK records of every sentence's top-1 path,
// in path order (EOS first), compacted to offs[s]...  One wavefront per sentence.
__global__ void k_top1_count(Batch B, u32* counts) {
  u32 s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= B.n_sent) return;
  counts[s] = B.sent_status[s] == ST_OK ? B.path_len[s] : 0;
}

__global__ void __launch_bounds__(256) k_top1_gather(Batch B, const u64* offs, NodeInfo* nodes, NodeAux* aux) {
  const u32 s = blockIdx.x * 4 + (threadIdx.x >> 6);
  const u32 lane = threadIdx.x & 63;
  if (s >= B.n_sent || B.sent_status[s] != ST_OK) return;
  const u32 pl = B.path_len[s];
  const u64 nb = B.node_base[s], o = offs[s];
  for (u32 k = lane; k < pl; k += 64) {
    u32 node = B.path_nodes[nb + k];
    nodes[o + k] = B.node_info[nb + node];
    aux[o + k] = B.node_aux[nb + node];
  }
}

// jppgpu_result_fetch_nbest: what the lattice output format reads -- the beam slots, node records and
// score cells along the n best paths from EOS -- compacted on the device (LatticeFormatInfo::fillInfo,
// src/jumandic/shared/lattice_format.cc:13-43, walks exactly these).  One wavefront per sentence, lane i =
// path i.  WRITE = false counts the items of every path, WRITE = true emits them at the scanned offsets.
struct NbestItem {
  u32 node;       // sentence-local lattice node
  u32 slot;       // its beam slot the path runs through
  BeamSlot beam;  // that slot
  NodeInfo info;
  NodeAux aux;
  float cells[2];  // score cells of the connection (one per scorer)
};

template <bool WRITE>
__global__ void __launch_bounds__(256) k_nbest(Batch B, Config cfg, int n_best, u32* counts, const u64* offs,
                                               NbestItem* items, BeamSlot* eos_slots) {
  const u32 s = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = (int)(threadIdx.x & 63);
  if (s >= B.n_sent) return;
  if (lane >= n_best) return;
  const u64 q = (u64)s * (u32)n_best + (u32)lane;
  const BeamSlot fake{kFake16, kFake16, 0.f, 0xffffffffu, 0};
  if (B.sent_status[s] != ST_OK || B.sent_nodes[s] <= 3) {
    if (!WRITE) counts[q] = 0;
    else eos_slots[q] = fake;
    return;
  }
  const int beam = cfg.beam, G = cfg.gbeam, S = cfg.nscorers;
  const u64 nb = B.node_base[s];
  const u32 N = B.sent_nodes[s];
  const BeamSlot* beams = B.node_beam + nb * beam;
  BeamSlot el = lane < beam ? beams[(u64)(N - 1) * beam + lane] : fake;
  if (WRITE) eos_slots[q] = el;
  u32 cnt = 0;
  if (!(el.left == kFake16 && el.beam == kFake16)) {
    u64 o = WRITE ? offs[q] : 0;
    u32 node = el.prev_node, slot = el.beam;
    while (node >= 2 && node != 0xffffffffu && cnt <= N) {
      const BeamSlot c = beams[(u64)node * beam + slot];
      if (WRITE) {
        NbestItem it;
        it.node = node;
        it.slot = slot;
        it.beam = c;
        it.info = B.node_info[nb + node];
        it.aux = B.node_aux[nb + node];
        const float* cell = B.node_cells + ((nb + node) * G + c.pad) * S;
        it.cells[0] = G > 0 ? cell[0] : 0.f;
        it.cells[1] = (G > 0 && S > 1) ? cell[1] : 0.f;
        items[o + cnt] = it;
      }
      ++cnt;
      node = c.prev_node;
      slot = c.beam;
    }
  }
  if (!WRITE) counts[q] = cnt;
}

}  // namespace jpp

#endif  // JPP_K_SWEEP_H
