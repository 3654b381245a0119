// Kernel 4 ("T0"): per lattice node -- entry row, 20 primitive features, 38
// pattern hashes (14 stored), 32 unigram weight gathers summed in the exact
// association of the reference's generated static code.
// One workgroup (64 lanes) per sentence, one lane per node.  The spec tables
// are constexpr, so every loop below unrolls into straight-line integer code;
// the only memory traffic per node is the varint entry row (8-24 B), 32
// scattered 4-byte weight gathers and the 112+32+4 byte result row.
//
// Reference behaviour reproduced:
//   generated PatternFeatureStaticApply_JumandicStatic::patternsAndUnigramsApply
//     (emitted by src/core/codegen/pattern_feature_codegen.cc; called from
//      ScoreProcessor::computeT0All, src/core/analysis/score_processor.cc:123-134)
//   PrimitiveFeatureContext::fillEntryBuffer / providedFeature
//     src/core/impl/feature_impl_types.h:104-148
//   primitive features  src/core/impl/feature_impl_prim.h:62-236
//   compute features    src/core/impl/feature_impl_compute.cc:12-26,59-63
//   pattern hash        src/core/impl/feature_impl_pattern.h:28-41
//   UnigramFeature::maskedValueFor  src/core/impl/feature_impl_ngram_partial.h:29-32
//   computeUnrolled4RawPerceptron   src/core/analysis/perceptron.h:46-72 (last row)
#ifndef JPP_K_T0_H
#define JPP_K_T0_H

#include "jpp_device.h"
#include "jumandic_spec.inc"
#include "k_seeds.h"

namespace jpp {

__host__ __device__ constexpr u64 pattern_prefix(int idx, int nargs) {
  return hmix(hmix(hmix(kHashSeed0, (u64)(u32)idx), (u64)nargs), kPatternSeed);
}
__host__ __device__ constexpr u64 uni_prefix(int index) {
  return hmix(hmix(hmix(kHashSeed0, 3), (u64)(u32)index), kUnigramSeed);
}
__host__ __device__ constexpr u64 bi_prefix(int index) {
  return hmix(hmix(hmix(kHashSeed0, 4), (u64)(u32)index), kBigramSeed);
}
__host__ __device__ constexpr u64 tri_prefix(int index) {
  return hmix(hmix(hmix(kHashSeed0, 5), (u64)(u32)index), kTrigramSeed);
}

// hash of pattern P over the primitive values (PatternFeatureStaticApply / feature_impl_pattern.h:28-41, compute features
// feature_impl_compute.cc:12-26)
template <int P>
__host__ __device__ inline u64 t0_pattern_one(const u64 (&prim)[spec::kNumPrims]) {
  u64 h = pattern_prefix(P, spec::kPatterns[P].nargs);
#pragma unroll
  for (int q = 0; q < spec::kPatterns[P].nargs; ++q) {
    const int c = spec::kPatterns[P].args[q];
    if (spec::kComputes[c].cond < 0) {
      h = hmix(h, prim[spec::kComputes[c].t[0]]);
    } else {
      u64 ht = h, hf = h;
#pragma unroll
      for (int z = 0; z < spec::kComputes[c].nt; ++z) ht = hmix(ht, prim[spec::kComputes[c].t[z]]);
#pragma unroll
      for (int z = 0; z < spec::kComputes[c].nf; ++z) hf = hmix(hf, prim[spec::kComputes[c].f[z]]);
      h = prim[spec::kComputes[c].cond] != 0 ? ht : hf;
    }
  }
  return h;
}
template <int P = 0>
__host__ __device__ inline void t0_pattern_hashes(const u64 (&prim)[spec::kNumPrims], u64 (&pat)[spec::kNumPatterns]) {
  if constexpr (P < spec::kNumPatterns) {
    pat[P] = t0_pattern_one<P>(prim);
    t0_pattern_hashes<P + 1>(prim, pat);
  }
}

// Which patterns look at the sentence around / under the node (codepoints, character classes) and which are a function
// of the dictionary entry and the length of its surface alone: the latter are the same for every lattice node of one
// dictionary entry, which is what the per-entry memo of k_t0_memo stores.
constexpr bool t0_prim_is_context(int p) { return spec::kPrims[p].kind == spec::Codepoint || spec::kPrims[p].kind == spec::CodepointType; }
constexpr bool t0_compute_is_context(int c) {
  bool r = false;
  if (spec::kComputes[c].cond < 0) return t0_prim_is_context(spec::kComputes[c].t[0]);
  r = t0_prim_is_context(spec::kComputes[c].cond);
  for (int z = 0; z < spec::kComputes[c].nt; ++z) r = r || t0_prim_is_context(spec::kComputes[c].t[z]);
  for (int z = 0; z < spec::kComputes[c].nf; ++z) r = r || t0_prim_is_context(spec::kComputes[c].f[z]);
  return r;
}
constexpr bool t0_pattern_is_context(int p) {
  bool r = false;
  for (int q = 0; q < spec::kPatterns[p].nargs; ++q) r = r || t0_compute_is_context(spec::kPatterns[p].args[q]);
  return r;
}

// Primitive features and the kNumPatterns pattern hashes of one node (generated
// PatternFeatureStaticApply_JumandicStatic::patternsAndUnigramsApply, first half; dynamic equivalent
// InNodeFeatureComputer + PatternDynamicApplyImpl::apply, innode_features.cc:11-33, feature_impl_pattern.h:59-65)
// from its entry row, its span and the codepoints / classes of the sentence.
__device__ __forceinline__ void t0_prims(const i32 (&entry)[spec::kNumDicFeatures], const NodeInfo& ni, const NodeAux& na,
                                         bool isUnk, const u32* cps, const i32* cls, u32 n, u64 (&prim)[spec::kNumPrims]) {
#pragma unroll
  for (int p = 0; p < spec::kNumPrims; ++p) {
    const int kind = spec::kPrims[p].kind;
    const int a = spec::kPrims[p].a;
    const int bsh = spec::kPrims[p].b;
    u64 v = 0;
    if (kind == spec::Copy) {
      v = (u32)entry[a];
    } else if (kind == spec::SingleBit) {
      v = ((u32)entry[a] >> bsh) & 1u;
    } else if (kind == spec::Provided) {
      v = isUnk ? (u64)(u32)(a == 0 ? na.ph0 : na.ph1) : 0;
    } else if (kind == spec::SurfaceCodepointSize) {
      v = (u64)((i32)ni.end - (i32)ni.start);
    } else if (kind == spec::Codepoint) {
      v = ~u64{0};
      if (a > 0) {
        u32 pos = (u32)ni.end + (u32)(a - 1);
        if (pos < n) v = cps[pos];
      } else {
        i32 pos = (i32)ni.start + a;
        if (pos >= 0 && (u32)pos < n) v = cps[pos];
      }
    } else if (kind == spec::CodepointType) {
      v = 0;
      if (a == 0) {
        for (u32 q = ni.start; q < ni.end; ++q) v |= (u32)cls[q];
      } else if (a > 0) {
        u32 pos = (u32)ni.end + (u32)(a - 1);
        if (pos < n) v = (u32)cls[pos];
      } else {
        i32 pos = (i32)ni.start + a;
        if (pos >= 0 && (u32)pos < n) v = (u32)cls[pos];
      }
    }
    prim[p] = v;
  }
}

__device__ __forceinline__ void t0_patterns(const i32 (&entry)[spec::kNumDicFeatures], const NodeInfo& ni, const NodeAux& na,
                                            bool isUnk, const u32* cps, const i32* cls, u32 n, u64 (&pat)[spec::kNumPatterns]) {
  u64 prim[spec::kNumPrims];
  t0_prims(entry, ni, na, isUnk, cps, cls, n, prim);
  t0_pattern_hashes(prim, pat);
}

// ---- per-entry memo ------------------------------------------------------------------------------------------------
// A dictionary node's entry row, its 14 stored patterns and 26 of its 32 unigram weights depend on the dictionary entry
// only (placeholders are 0 for dictionary nodes, the surface length is the key's): one 64-byte record per entry, built
// on the host when the model (or a new weight table) is loaded, replaces the varint decode and 26 scattered weight
// gathers -- the gathers are what bound k_t0 (3.2x line amplification at the fabric ceiling).
// Record of the entry at EntryPtr e: slot (e >> 1) >> 3 (an entry row is at least 8 bytes long, so slots are unique).
//   row:    the decoded entry row;
//   pre[j]: the weights of the features u = j, j + 4, ... < 23 summed in that order -- the four accumulators of
//           computeUnrolled4RawPerceptron up to the first feature that looks at the context;
//   raw[]:  the weights of the three features behind the context block (u = 29, 30, 31);
//   len:    codepoints of the key (0: no record -- the node takes the full path).
// The context features (u = 23..28) are hashed and gathered per node and added between the two, in the reference's order.
// Round 6: the record no longer carries the 14 stored patterns (176 -> 64 bytes: one 128-byte line per node instead of
// 2.4 on average, and a table a third the size); they are hashed from the row again -- 38 multiply-mixes per node on
// arithmetic units that a kernel at the fabric ceiling leaves idle.
struct alignas(16) U4 {
  u32 x, y, z, w;
};
__host__ __device__ inline float bits_f32(u32 v) { return __builtin_bit_cast(float, v); }

struct alignas(64) T0Memo {
  i32 row[spec::kNumDicFeatures];
  float pre[4];
  float raw[3];
  u32 len;
};
static_assert(sizeof(T0Memo) == 64 && spec::kNumDicFeatures == 8 && spec::kNumStoredPatterns == 14, "memo record layout");
constexpr int kT0CtxFirst = 23, kT0CtxLast = 28;   // unigram positions (summation order) that read the context
constexpr bool t0_memo_layout_ok() {
  if (spec::kNumUni != 32) return false;
  for (int u = 0; u < spec::kNumUni; ++u)
    if (t0_pattern_is_context(spec::kUni[u].t0) != (u >= kT0CtxFirst && u <= kT0CtxLast)) return false;
  for (int p = 0; p < spec::kNumStoredPatterns; ++p)
    if (t0_pattern_is_context(p)) return false;
  return true;
}
static_assert(t0_memo_layout_ok(), "the memo assumes which unigram features read the context");

// the first kNumStoredPatterns patterns only (what bigrams / trigrams read of a node)
template <int P = 0>
__host__ __device__ inline void t0_stored_patterns(const u64 (&prim)[spec::kNumPrims], u64 (&pat)[spec::kNumStoredPatterns]) {
  if constexpr (P < spec::kNumStoredPatterns) {
    pat[P] = t0_pattern_one<P>(prim);
    t0_stored_patterns<P + 1>(prim, pat);
  }
}

// entry-only primitives (host: building the memo; placeholders 0, context primitives unused)
__host__ __device__ inline void t0_entry_prims(const i32 (&entry)[spec::kNumDicFeatures], u32 len, u64 (&prim)[spec::kNumPrims]) {
  for (int p = 0; p < spec::kNumPrims; ++p) {
    const int kind = spec::kPrims[p].kind, a = spec::kPrims[p].a, bsh = spec::kPrims[p].b;
    u64 v = 0;
    if (kind == spec::Copy) v = (u32)entry[a];
    else if (kind == spec::SingleBit) v = ((u32)entry[a] >> bsh) & 1u;
    else if (kind == spec::SurfaceCodepointSize) v = len;
    prim[p] = v;
  }
}

// the weights of the context features (summation positions kT0CtxFirst..kT0CtxLast) of one node
template <bool W24, int U = kT0CtxFirst, typename WP>
__device__ __forceinline__ void t0_context_weights(const u64 (&prim)[spec::kNumPrims], WP weights, u32 wmask, float (&wc)[kT0CtxLast - kT0CtxFirst + 1]) {
  if constexpr (U <= kT0CtxLast) {
    const u32 idx = hmix_index<W24>(uni_prefix(spec::kUni[U].index), t0_pattern_one<spec::kUni[U].t0>(prim), wmask);
    wc[U - kT0CtxFirst] = weights[idx];
    t0_context_weights<W24, U + 1>(prim, weights, wmask, wc);
  }
}

// what a sentence's T0 pass needs of its sentence
struct T0Sent {
  u32 n, N, bb0;
  u64 nb;
  const u32* cps;
  const i32* cls;
};

// one node, everything computed from scratch
template <bool W24>
__device__ __forceinline__ void t0_node_full(const Batch& B, const DevModel& M, const T0Sent& S, u32 k) {
  NodeInfo ni = B.node_info[S.nb + k];
  NodeAux na = B.node_aux[S.nb + k];
  u32 b = (k == S.N - 1) ? S.n + 2 : (u32)ni.start + 2;
  u32 first = B.bnd_first[S.bb0 + b];
  u32 R = B.bnd_cnt[S.bb0 + b];
  bool isLast = (k - first) == R - 1;

  // ---- entry row ----
  i32 entry[spec::kNumDicFeatures];
  bool isUnk = false;
  if (ni.eptr == kEptrEOS) {
#pragma unroll
    for (int f = 0; f < spec::kNumDicFeatures; ++f) entry[f] = kEptrEOS;
  } else if (ni.eptr >= 0) {
    read_entry_row(M, ni.eptr, entry, spec::kNumDicFeatures);
  } else {
    isUnk = true;
    const UnkMaker& mk = M.makers[M.maker_of_spec[na.maker]];
    if (mk.type == UNK_NORMALIZE) {
      read_entry_row(M, na.tmpl, entry, spec::kNumDicFeatures);
    } else {
#pragma unroll
      for (int f = 0; f < spec::kNumDicFeatures; ++f) entry[f] = mk.tmpl[f];
    }
#pragma unroll
    for (int f = 0; f < spec::kNumDicFeatures; ++f) {
      if ((mk.replace_mask >> f) & 1) entry[f] = na.hash;
    }
  }
#pragma unroll
  for (int f = 0; f < spec::kNumDicFeatures; ++f) B.node_entry[(S.nb + k) * spec::kNumDicFeatures + f] = entry[f];

  u64 pat[spec::kNumPatterns];
  t0_patterns(entry, ni, na, isUnk, S.cps, S.cls, S.n, pat);
#pragma unroll
  for (int p = 0; p < spec::kNumStoredPatterns; ++p) B.node_pat[(S.nb + k) * kPat + p] = pat[p];

  // ---- unigram perceptron ----
  static_assert(spec::kNumUni >= 4, "unigram count");
  float w[spec::kNumUni];
#pragma unroll
  for (int u = 0; u < spec::kNumUni; ++u) {
    u32 idx = hmix_index<W24>(uni_prefix(spec::kUni[u].index), pat[spec::kUni[u].t0], M.wmask);
    w[u] = as_global(M.weights)[idx];
  }
  float part[4];
  if (isLast) {
#pragma unroll
    for (int j = 0; j < 4; ++j) part[j] = 0.f;
#pragma unroll
    for (int u = 0; u < spec::kNumUni; ++u) part[u & 3] += w[u];
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j) part[j] = w[j];
#pragma unroll
    for (int u = 4; u < spec::kNumUni; ++u) part[u & 3] += w[u];
  }
  B.node_t0[S.nb + k] = part[0] + part[1] + part[2] + part[3];
}

__device__ __forceinline__ T0Sent t0_sentence(const Batch& B, u32 s) {
  const u32 off = B.byte_off[s];
  const u32 g0 = off + s;
  return T0Sent{B.sent_ncp[s], B.sent_nodes[s], off + 4 * s, B.node_base[s], B.cp_code + g0, B.cp_class + g0};
}

// (more wavefronts per SIMD do not help here: 5 at 94 VGPRs 2.01 ms, 6 at 80 2.05 ms, 8 at 64 with spills 2.68 ms)
template <bool W24>   // W24: at most 2^24 weights (hmix_index)
__global__ void __launch_bounds__(64) k_t0(Batch B, const DevModel* __restrict__ Mp) {
  const DevModel& M = *Mp;
  u32 s = blockIdx.x;
  if (B.sent_status[s] != ST_OK) return;
  const T0Sent S = t0_sentence(B, s);
  for (u32 k = 2 + threadIdx.x; k < S.N; k += blockDim.x) t0_node_full<W24>(B, M, S, k);
}

// The same with the per-entry memo (T0Memo above).  Pass A: every lane takes one node; a dictionary node whose record is
// valid copies its row and stored patterns from it, hashes the six context patterns, gathers their six weights and
// finishes the four accumulators; every other node (UNK makers' nodes, EOS, entries without a record) is put on a list
// in LDS.  Pass B: the listed nodes, one per lane again, from scratch -- so that the long path runs with full wavefronts
// instead of on the few lanes of pass A that missed.
constexpr u32 kT0MissCap = 512;
template <bool W24>
__global__ void __launch_bounds__(64) k_t0_memo(Batch B, const DevModel* __restrict__ Mp, const T0Memo* __restrict__ memo, u32 nslots) {
  const DevModel& M = *Mp;
  const u32 s = blockIdx.x;
  if (B.sent_status[s] != ST_OK) return;
  const T0Sent S = t0_sentence(B, s);
  const int lane = (int)threadIdx.x;
  __shared__ u32 miss[kT0MissCap];
  u32 nmiss = 0;   // wave-uniform
  auto drain = [&]() {
    wave_sync();
    for (u32 q = (u32)lane; q < nmiss; q += 64) t0_node_full<W24>(B, M, S, miss[q]);
    wave_sync();
    nmiss = 0;
  };
  for (u32 k0 = 2; k0 < S.N; k0 += 64) {
    const u32 k = k0 + (u32)lane;
    bool hit = false;
    if (k < S.N) {
      const NodeInfo ni = B.node_info[S.nb + k];
      const u32 len = (u32)ni.end - (u32)ni.start;
      const u32 slot = ni.eptr >= 0 ? (u32)ni.eptr >> 4 : nslots;
      if (slot < nslots) {
        const U4 JPP_GLOBAL* rec = reinterpret_cast<const U4 JPP_GLOBAL*>(as_global(memo + slot));
        // (the whole record at once -- it is one line -- instead of its length word first and the rest behind the test)
        const U4 q0 = rec[0], q1 = rec[1], q2 = rec[2];
        const U4 tail = rec[3];   // raw[0..2], len
        if (tail.w == len) {
          hit = true;
          // entry row, as read
          U4* orow = reinterpret_cast<U4*>(B.node_entry + (S.nb + k) * spec::kNumDicFeatures);
          orow[0] = q0;
          orow[1] = q1;
          i32 entry[spec::kNumDicFeatures] = {(i32)q0.x, (i32)q0.y, (i32)q0.z, (i32)q0.w, (i32)q1.x, (i32)q1.y, (i32)q1.z, (i32)q1.w};
          u64 prim[spec::kNumPrims];
          t0_prims(entry, ni, NodeAux{0, 0, 0, 0, 0, 0}, false, S.cps, S.cls, S.n, prim);
          // (the six gathers first: the hashing below runs while they are in flight)
          float wc[kT0CtxLast - kT0CtxFirst + 1];
          t0_context_weights<W24>(prim, as_global(M.weights), M.wmask, wc);
          // the stored patterns, hashed from the row (they are functions of the entry and its length alone)
          u64 sp[spec::kNumStoredPatterns];
          t0_stored_patterns(prim, sp);
          U4* opat = reinterpret_cast<U4*>(B.node_pat + (S.nb + k) * kPat);
#pragma unroll
          for (int z = 0; z < 7; ++z) opat[z] = U4{(u32)sp[2 * z], (u32)(sp[2 * z] >> 32), (u32)sp[2 * z + 1], (u32)(sp[2 * z + 1] >> 32)};
          float part[4] = {bits_f32(q2.x), bits_f32(q2.y), bits_f32(q2.z), bits_f32(q2.w)};
          const float raw[3] = {bits_f32(tail.x), bits_f32(tail.y), bits_f32(tail.z)};
#pragma unroll
          for (int u = kT0CtxFirst; u < spec::kNumUni; ++u) part[u & 3] += u <= kT0CtxLast ? wc[u - kT0CtxFirst] : raw[u - kT0CtxLast - 1];
          B.node_t0[S.nb + k] = part[0] + part[1] + part[2] + part[3];
        }
      }
    }
    const u64 bal = wave_ballot(k < S.N && !hit);
    if (bal != 0) {
      if (k < S.N && !hit) miss[nmiss + (u32)popc64(bal & ((u64{1} << lane) - 1))] = k;
      nmiss += (u32)popc64(bal);
      if (nmiss + 64 > kT0MissCap) drain();
    }
  }
  if (nmiss) drain();
}

// The same kernel for a spec other than the built-in jumandic tables: every descriptor is read from the DevSpec
// tables (staged in LDS), and the unigram weights are summed the way the reference's DYNAMIC feature code sums
// them -- PartialNgramFeatureApplyImpl::applyUni, feature_impl_ngram_partial.h:188-214: every row through
// computeUnrolled4RawPerceptron (perceptron.h:46-72), four partial sums from 0.f in feature order, ((r1+r2)+r3)+r4.
// Patterns: PatternDynamicApplyImpl (feature_impl_pattern.h:28-65); only those a bigram / trigram reads are stored
// (DevSpec::Pattern::slot), the reference's dynamic lattice stores all of them.
__global__ void __launch_bounds__(64) k_t0_dyn(Batch B, const DevModel* __restrict__ Mp) {
  const DevModel& M = *Mp;
  __shared__ DevSpec S;
  {
    const u32* src = reinterpret_cast<const u32*>(M.spec);
    u32* dst = reinterpret_cast<u32*>(&S);
    for (u32 i = threadIdx.x; i < sizeof(DevSpec) / 4; i += blockDim.x) dst[i] = src[i];
  }
  __syncthreads();
  u32 s = blockIdx.x;
  if (B.sent_status[s] != ST_OK) return;
  u32 off = B.byte_off[s];
  u32 g0 = off + s;
  u32 n = B.sent_ncp[s];
  u32 N = B.sent_nodes[s];
  u64 nb = B.node_base[s];
  const u32* cps = B.cp_code + g0;
  const i32* cls = B.cp_class + g0;
  const u16* boffs = B.cp_boff + g0;
  const int nf = M.num_features;
  const float JPP_GLOBAL* W = as_global(M.weights);

  for (u32 k = 2 + threadIdx.x; k < N; k += blockDim.x) {
    NodeInfo ni = B.node_info[nb + k];
    NodeAux na = B.node_aux[nb + k];
    // ---- entry row (up to kMaxDicFeatures = JPP_MAX_DIC_FIELDS columns, round 5) ----
    constexpr int kRowMax = kMaxDicFeatures;
    i32 entry[kRowMax];
#pragma unroll
    for (int f = 0; f < kRowMax; ++f) entry[f] = 0;
    bool isUnk = false;
    if (ni.eptr == kEptrEOS) {
#pragma unroll
      for (int f = 0; f < kRowMax; ++f) entry[f] = f < nf ? kEptrEOS : 0;
    } else if (ni.eptr >= 0) {
      read_entry_row(M, ni.eptr, entry, nf);
    } else if (na.maker == kGoldMaker) {
      // a gold node of the trainer: the row is given (gold_example.cc:118-136)
      isUnk = true;
      const ExtraSeed* g = B.gold + B.gold_off[s] + na.pad;
#pragma unroll
      for (int f = 0; f < kRowMax; ++f) entry[f] = (f < nf && f < 8) ? g->row[f < 8 ? f : 0] : 0;   // (the trainer's gold rows: 8 columns)
    } else {
      isUnk = true;
      const UnkMaker& mk = M.makers[M.maker_of_spec[na.maker]];
      if (mk.type == UNK_NORMALIZE) {
        read_entry_row(M, na.tmpl, entry, nf);
      } else {
#pragma unroll
        for (int f = 0; f < kRowMax; ++f) entry[f] = f < nf ? mk.tmpl[f] : 0;
      }
#pragma unroll
      for (int f = 0; f < kRowMax; ++f) {
        if (f < nf && ((mk.replace_mask >> f) & 1)) entry[f] = na.hash;
      }
    }
#pragma unroll
    for (int f = 0; f < kRowMax; ++f) if (f < (int)B.row_stride) B.node_entry[(nb + k) * B.row_stride + f] = entry[f];

    // ---- primitive features (feature_impl_prim.h:62-236) ----
    auto primitive = [&](int p) -> u64 {
      const int kind = S.prims[p].kind, a = S.prims[p].a, bsh = S.prims[p].b;
      u32 col = 0;
#pragma unroll
      for (int f = 0; f < kRowMax; ++f) col = f == a ? (u32)entry[f] : col;   // (no dynamic register index)
      if (kind == spec::Copy) return col;
      if (kind == spec::SingleBit) return (col >> bsh) & 1u;
      if (kind == spec::Provided) return isUnk ? (u64)(u32)(a == 0 ? na.ph0 : na.ph1) : 0;
      if (kind == spec::SurfaceCodepointSize) return (u64)((i32)ni.end - (i32)ni.start);
      if (kind == spec::ByteLength || kind == spec::CodepointSize) {
        // PrimitiveFeatureContext::lengthOf (feature_impl_types.h:156-174): a negative column value is an UNK node's
        // surface hash -> the BYTES of its surface, whichever of the two primitives asks (extra_nodes.h:89-93);
        // otherwise the column's storage: int lists ("positions") their count, strings their byte length or codepoints
        const i32 fp = (i32)col;
        if (fp < 0) return (u64)((u32)boffs[ni.end] - (u32)boffs[ni.start]);
        const DevSpec::Storage st = S.storages[bsh];
        const u64 at = st.kind == 2 ? (u64)(u32)fp : (u64)(u32)fp << st.align;
        if (at >= st.bytes) return ~u64{0};   // (cannot happen with a model the reference built)
        const u8* p = st.data + at;
        u32 len = 0;
        int used = 0;
        for (int sh = 0; sh < 35; sh += 7) {
          const u32 b = p[used++];
          len |= (b & 0x7fu) << sh;
          if (b < 0x80u) break;
        }
        if (st.kind == 2 || kind == spec::ByteLength) return (u64)len;
        // chars::numCodepoints (src/util/characters.cc:278-301): steps by the class of the lead byte; -1 when the last
        // step overshoots the string
        u32 q = 0, ncp = 0;
        while (q < len && at + used + q < st.bytes) {
          const u32 b = p[used + q];
          q += b > 0xefu ? 4u : b > 0xdfu ? 3u : b > 0x7fu ? 2u : 1u;
          ++ncp;
        }
        return q != len ? ~u64{0} : (u64)ncp;
      }
      if (kind == spec::Codepoint) {
        if (a > 0) {
          const u32 pos = (u32)ni.end + (u32)(a - 1);
          return pos < n ? (u64)cps[pos] : ~u64{0};
        }
        const i32 pos = (i32)ni.start + a;
        return (pos >= 0 && (u32)pos < n) ? (u64)cps[pos] : ~u64{0};
      }
      // CodepointType
      if (a == 0) {
        u64 v = 0;
        for (u32 q = ni.start; q < ni.end; ++q) v |= (u32)cls[q];
        return v;
      }
      if (a > 0) {
        const u32 pos = (u32)ni.end + (u32)(a - 1);
        return pos < n ? (u64)(u32)cls[pos] : 0;
      }
      const i32 pos = (i32)ni.start + a;
      return (pos >= 0 && (u32)pos < n) ? (u64)(u32)cls[pos] : 0;
    };
    // ---- patterns, unigram weights ----
    float r[4] = {0.f, 0.f, 0.f, 0.f};
    // (pattern by pattern: a pattern's hash feeds the unigram features over it and, if stored, its slot; unigram
    // features are summed in FEATURE order, so their weights are gathered first and added afterwards)
    float wu[kDynMaxUni];
    for (int p = 0; p < S.npatterns; ++p) {
      u64 h = S.patterns[p].prefix;
      for (int q = 0; q < S.patterns[p].nargs; ++q) {
        const DevSpec::Compute& c = S.computes[S.patterns[p].args[q]];
        if (c.cond < 0) {
          h = hmix(h, primitive(c.t[0]));
        } else if (primitive(c.cond) != 0) {
          for (int z = 0; z < c.nt; ++z) h = hmix(h, primitive(c.t[z]));
        } else {
          for (int z = 0; z < c.nf; ++z) h = hmix(h, primitive(c.f[z]));
        }
      }
      if (S.patterns[p].slot >= 0) B.node_pat[(nb + k) * kPat + S.patterns[p].slot] = h;
      for (int u = 0; u < S.nuni; ++u)
        if (S.uni[u].t0 == p) wu[u] = W[(u32)hmix(S.uni[u].prefix, h) & M.wmask];
    }
    for (int q = S.nstored; q < kPat; ++q) B.node_pat[(nb + k) * kPat + q] = 0;
    {
      int u = 0;
      for (; u + 4 <= S.nuni; u += 4) {
        r[0] += wu[u];
        r[1] += wu[u + 1];
        r[2] += wu[u + 2];
        r[3] += wu[u + 3];
      }
      const int rest = S.nuni - u;
      if (rest >= 3) r[2] += wu[u + 2];
      if (rest >= 2) r[1] += wu[u + 1];
      if (rest >= 1) r[0] += wu[u];
    }
    B.node_t0[nb + k] = r[0] + r[1] + r[2] + r[3];
  }
}

}  // namespace jpp

#endif  // JPP_K_T0_H
