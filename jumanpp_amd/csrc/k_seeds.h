// Kernel 2: lattice node seeds.  One 64-lane workgroup per sentence, one lane
// per start codepoint: dictionary double-array walk + entry-pointer list
// expansion, then the rule-based UNK makers for that start, in the order that
// the reference's stable sort by start produces:
//   dictionary seeds (end asc, list order), then every stage-1 maker in spec order.
// Stage-2 (lowPriority) makers are counted/emitted separately; they only join
// the lattice of sentences whose stage-1 lattice is disconnected.
//
// Reference behaviour reproduced:
//   DictionaryNodeCreator::spawnNodes  src/core/analysis/dictionary_node_creator.cc:11-38
//   IndexedEntries::readOnePtr         src/core/dic/dic_entries.h:171-179 (+ field_reader.h:78-85,176-187)
//   SingleUnkMaker / ChunkingUnkMaker  src/core/analysis/unk_nodes_creator.cc:19-103
//   UnkNodesContext::makePtr           src/core/analysis/unk_nodes_creator.cc:105-142
//   UnkNodesContext::dicPatternMatches src/core/analysis/unk_nodes_creator.cc:144-168
//   NumericUnkMaker                    src/core/analysis/numeric_creator.cc:57-274
//   OnomatopoeiaUnkMaker               src/core/analysis/onomatopoeia_creator.cc:16-104
//   NormalizedNodeMaker + CharLattceTraversal
//                                      src/core/analysis/normalized_node_creator.cc:15-48,
//                                      src/core/analysis/charlattice.cc:266-415
#ifndef JPP_K_SEEDS_H
#define JPP_K_SEEDS_H

#include "jpp_device.h"
#include "jpp_select.h"
#include "k_decode.h"

namespace jpp {

// The decoded sentence as the makers see it (`V` below): in HBM, or -- ordinary sentences -- a copy in LDS made once per
// workgroup.  The emit pass alone issued ~175 dependent vector-memory reads per wavefront, most of them one lane
// looking at the class / codepoint / bytes of a neighbouring position; from LDS they cost a tenth of an L2 round trip.
struct SentViewG {
  const u8* txt;     // sentence bytes
  const u32* cp;     // codepoints
  const i32* cls;    // classes
  const u16* boff;   // byte offsets (n+1 entries)
  u32 n;
};
struct SentViewL {
  const u8 JPP_LDS* txt;
  const u32 JPP_LDS* cp;
  const i32 JPP_LDS* cls;
  const u16 JPP_LDS* boff;
  u32 n;
};
constexpr u32 kSentLdsCp = 128, kSentLdsBytes = 512;   // sentences up to this size are staged in LDS
struct SentLds {
  u32 cp[kSentLdsCp];
  i32 cls[kSentLdsCp];
  u16 boff[kSentLdsCp + 2];
  u8 txt[kSentLdsBytes + 8];
};
// (all lanes of the workgroup call; `bytes` = byte length of the sentence)
__device__ __forceinline__ void stage_sentence(SentLds& L, const SentViewG& S, u32 bytes) {
  for (u32 i = threadIdx.x; i < S.n; i += blockDim.x) {
    L.cp[i] = S.cp[i];
    L.cls[i] = S.cls[i];
  }
  for (u32 i = threadIdx.x; i <= S.n; i += blockDim.x) L.boff[i] = S.boff[i];
  for (u32 i = threadIdx.x; i < bytes; i += blockDim.x) L.txt[i] = S.txt[i];
  __syncthreads();
}
__device__ __forceinline__ SentViewL lds_view(const SentLds& L, u32 n) {
  return SentViewL{as_lds(L.txt), as_lds(L.cp), as_lds(L.cls), as_lds(L.boff), n};
}

struct SeedSink {
  bool emit;
  NodeInfo* ni;
  NodeAux* na;
  u32 n;
  u64 ends;  // count pass: bit e for every node end e <= 63 (reachability without emitting, k_layout<1>)
  __device__ __forceinline__ void mark_end(u32 e) { ends |= e < 64 ? (u64{1} << e) : u64{0}; }
  __device__ __forceinline__ void dic(i32 eptr, u32 s, u32 e) {
    mark_end(e);
    if (emit) {
      ni[n] = NodeInfo{eptr, (u16)s, (u16)e};
      na[n] = NodeAux{0, 0, 0, 0, 0, 0};
    }
    ++n;
  }
  // count pass: `cnt` dictionary nodes ending at e
  __device__ __forceinline__ void dic_count(u32 cnt, u32 e) {
    if (cnt) mark_end(e);
    n += cnt;
  }
  // `rank`: the maker's position in the creation sequence; the provisional entry pointer -(1 + rank) carries it to
  // k_ends, which numbers the UNK nodes maker by maker
  __device__ __forceinline__ void unk(i32 tmpl, i32 hash, i32 ph0, i32 ph1, u32 maker, i32 rank, u32 s, u32 e) {
    mark_end(e);
    if (emit) {
      ni[n] = NodeInfo{-(1 + rank), (u16)s, (u16)e};  // final ~index is assigned by k_ends
      na[n] = NodeAux{tmpl, hash, (u16)ph0, (u16)ph1, (u16)maker, 0};
    }
    ++n;
  }
};

template <typename V>
__device__ __forceinline__ int step_cp(const DevModel& M, const V& S, TrieCursor& c, u32 j) {
  return trie_step(as_global(M.trie), c, S.txt + S.boff[j], (int)(S.boff[j + 1] - S.boff[j]));
}

template <typename V>
__device__ __forceinline__ i32 surface_hash(const V& S, u32 s, u32 e) {
  return unk_string_hash(S.txt + S.boff[s], (u32)(S.boff[e] - S.boff[s]));
}

// makePtr(surface, conf, notPrefix)
template <typename V>
__device__ __forceinline__ void emit_unk(const DevModel& M, const UnkMaker& mk, const V& S,
                                         SeedSink& out, u32 s, u32 e, bool notPrefix) {
  if (!out.emit) {
    out.mark_end(e);
    ++out.n;
    return;
  }
  i32 ph[2] = {0, 0};
  if (mk.placeholder >= 0 && mk.placeholder < 2) ph[mk.placeholder] = notPrefix ? 1 : 0;
  out.unk(mk.pattern_ptr, surface_hash(S, s, e), ph[0], ph[1], (u32)mk.spec_index, mk.rank, s, e);
}

// expand the entry-pointer list at trie value `v`
template <typename F>
__device__ __forceinline__ void for_each_entry(const DevModel& M, i32 v, F&& f) {
  u32 pos = (u32)v;
  VarintWindow<const u8 JPP_GLOBAL*> win(as_global(M.entry_ptrs), pos);
  i32 cnt = (i32)win.next(pos);
  i32 ptr = 0;
  for (i32 k = 0; k < cnt; ++k) {
    ptr += (i32)win.next(pos);
    f(ptr);
  }
}

// the length of the entry-pointer list at trie value `v` (all the count pass needs of it)
__device__ __forceinline__ u32 entry_count(const DevModel& M, i32 v) {
  u32 pos = (u32)v;
  return (u32)read_varint(as_global(M.entry_ptrs), pos);
}

// dictionary nodes of one key: counted (count pass) or emitted
__device__ __forceinline__ void dic_nodes(const DevModel& M, i32 v, u32 i, u32 e, SeedSink& out) {
  if (!out.emit) out.dic_count(entry_count(M, v), e);
  else for_each_entry(M, v, [&](i32 ptr) { out.dic(ptr, i, e); });
}

// What the dictionary walk from one start learned about the prefixes of the input: the UNK makers ask the
// same questions ("is this prefix a word / a prefix of a word / neither"), and every one of them would
// otherwise walk the double array again from the same start.
struct WalkInfo {
  u32 ok_len;  // prefixes of up to ok_len codepoints are nodes of the trie (status != NoNode)
  u64 leaf;    // bit (len - 1): the prefix of `len` codepoints is a dictionary key (status Ok)
  bool valid;  // false if the walk went past 64 codepoints (makers then walk themselves)
};

__device__ __forceinline__ int walk_status(const WalkInfo& w, u32 len) {
  if (len > w.ok_len) return TRIE_NONODE;
  return ((w.leaf >> (len - 1)) & 1) ? TRIE_OK : TRIE_NOLEAF;
}

// `rec` (count pass): where to record the walk for the emit passes.  The record is written straight to
// memory: a local WalkCache indexed by the running key count would live in scratch.
template <typename V>
__device__ __forceinline__ WalkInfo dic_seeds(const DevModel& M, const V& S, u32 i, SeedSink& out,
                                              WalkCache* rec = nullptr) {
  TrieCursor c{0, 0};
  WalkInfo w{0, 0, true};
  u32 nvals = 0;
  for (u32 j = i; j < S.n; ++j) {
    int st = step_cp(M, S, c, j);
    if (st == TRIE_NONODE) break;
    const u32 len = j - i + 1;
    if (len > 64) {
      w.valid = false;
    } else {
      w.ok_len = len;
      if (st == TRIE_OK) w.leaf |= u64{1} << (len - 1);
    }
    if (st == TRIE_OK) {
      if (rec && nvals < (u32)kWalkCacheVals) rec->vals[nvals] = c.value;
      ++nvals;
      u32 e = j + 1;
      dic_nodes(M, c.value, i, e, out);
    }
  }
  if (rec) {
    rec->leaf = w.leaf;
    rec->ok_len = (u8)w.ok_len;
    rec->cached = (nvals <= (u32)kWalkCacheVals && w.valid) ? 1 : 0;
  }
  return w;
}

// the same seeds from the recorded walk: no trie access
__device__ __forceinline__ WalkInfo dic_seeds_replay(const DevModel& M, const WalkCache* wc, u64 leaf, u32 ok_len, u32 i,
                                                     SeedSink& out) {
  WalkInfo w{ok_len, leaf, true};
  u64 bits = leaf;
  u32 k = 0;
  while (bits) {
    const u32 len = (u32)__builtin_ctzll(bits) + 1;
    bits &= bits - 1;
    const u32 e = i + len;
    dic_nodes(M, wc->vals[k], i, e, out);
    ++k;
  }
  return w;
}

template <typename V>
__device__ __forceinline__ void single_maker(const DevModel& M, const UnkMaker& mk, const V& S,
                                             u32 i, const WalkInfo& w, SeedSink& out) {
  if ((S.cls[i] & mk.char_class) == 0) return;
  int st;
  if (w.valid) {
    st = walk_status(w, 1);
  } else {
    TrieCursor c{0, 0};
    st = step_cp(M, S, c, i);
  }
  if (st == TRIE_OK) return;
  emit_unk(M, mk, S, out, i, i + 1, st == TRIE_NONODE);
}

template <typename V>
__device__ __forceinline__ void chunking_maker(const DevModel& M, const UnkMaker& mk, const V& S,
                                               u32 i, const WalkInfo& w, SeedSink& out) {
  if ((S.cls[i] & mk.char_class) == 0) return;
  TrieCursor c{0, 0};
  for (u32 j = i; j < S.n; ++j) {
    if ((S.cls[j] & mk.char_class) == 0) break;
    int st = w.valid ? walk_status(w, j - i + 1) : step_cp(M, S, c, j);
    if (st == TRIE_NONODE) {
      for (; j < S.n; ++j) {
        if ((S.cls[j] & mk.char_class) == 0) break;
        emit_unk(M, mk, S, out, i, j + 1, true);
      }
      return;
    } else if (st == TRIE_NOLEAF) {
      emit_unk(M, mk, S, out, i, j + 1, false);
    }
  }
}

// ---- numeric ---------------------------------------------------------------
template <typename V>
__device__ __forceinline__ bool cls_has(const V& S, u32 k, i32 mask) { return (S.cls[k] & mask) != 0; }

template <typename V>
__device__ __forceinline__ u32 num_check_suffix(const V& S, u32 start, u32 pos) {
  // suffixPatterns キロ メガ ギガ テラ ミリ
  const u32 pats[5][2] = {{U'キ', U'ロ'}, {U'メ', U'ガ'}, {U'ギ', U'ガ'}, {U'テ', U'ラ'}, {U'ミ', U'リ'}};
  pos &= 0xffff;
  u32 rest = S.n - (start + pos);
  if (pos > 0) {
    for (int p = 0; p < 5; ++p) {
      bool isException = cls_has(S, start + pos - 1, CC_FAMILY_EXCEPTION);
      if (isException && rest >= 2) {
        if (S.cp[start + pos] == pats[p][0] && S.cp[start + pos + 1] == pats[p][1]) return 2;
      }
    }
  }
  return 0;
}

template <typename V>
__device__ __forceinline__ u32 num_check_prefix(const V& S, u32 start, u32 pos) {
  const u32 pats[3] = {U'数', U'何', U'幾'};
  for (int p = 0; p < 3; ++p) {
    u32 suffixLength = num_check_suffix(S, start, pos + 1) & 0xffff;
    if (start + pos + 1 < S.n && (cls_has(S, start + pos + 1, CC_FIGURE_DIGIT) || suffixLength > 0)) {
      if (S.cp[start + pos] == pats[p]) return 1 + suffixLength;
    }
  }
  return 0;
}

template <typename V>
__device__ __forceinline__ u32 num_check_interfix(const V& S, u32 start, u32 pos, i32 cc) {
  u32 rest = S.n - (start + pos);
  if (pos > 0) {
    // ぶんの
    if (cls_has(S, start + pos - 1, cc) && rest > 3 && cls_has(S, start + pos + 3, cc)) {
      if (S.cp[start + pos] == U'ぶ' && S.cp[start + pos + 1] == U'ん' && S.cp[start + pos + 2] == U'の') return 3;
    }
    // 分の
    if (cls_has(S, start + pos - 1, cc) && rest > 2 && cls_has(S, start + pos + 2, cc)) {
      if (S.cp[start + pos] == U'分' && S.cp[start + pos + 1] == U'の') return 2;
    }
  }
  return 0;
}

template <typename V>
__device__ __forceinline__ u32 num_check_comma(const V& S, u32 start, u32 pos) {
  u32 posComma = (start + pos) & 0xffff;
  if (pos == 0) return 0;
  if (!cls_has(S, posComma, CC_COMMA)) return 0;
  u32 k = 0;
  for (k = 0; k <= 4 && posComma + 1 + k < S.n; ++k) {
    if (!cls_has(S, posComma + 1 + k, CC_FIGURE)) break;
  }
  return k == 3 ? 1 : 0;
}

template <typename V>
__device__ __forceinline__ u32 num_check_period(const V& S, u32 start, u32 pos, i32 cc) {
  u32 pp = start + pos;
  if (pos == 0) return 0;
  if (!cls_has(S, pp, CC_FAMILY_NUM_PERIOD)) return 0;
  if (!cls_has(S, pp - 1, cc)) return 0;
  if (pp + 1 < S.n && cls_has(S, pp + 1, cc)) return 1;
  return 0;
}

template <typename V>
__device__ __forceinline__ u32 num_find_longest(const V& S, u32 start, i32 cc) {
  u32 pos = 0;
  for (pos = 0; pos <= 64 && start + pos < S.n; pos = (pos + 1) & 0xffff) {
    if (!cls_has(S, start + pos, cc)) {
      u32 len = num_check_prefix(S, start, pos);
      if (len == 0) len = num_check_interfix(S, start, pos, cc);
      if (len == 0) len = num_check_suffix(S, start, pos);
      if (len == 0) len = num_check_comma(S, start, pos);
      if (len == 0) len = num_check_period(S, start, pos, cc);
      if (len > 0) {
        pos = (pos + len - 1) & 0xffff;
      } else {
        return pos;
      }
    }
  }
  return pos;
}

// decode the first `nf` ints of the entry row at EntryPtr `eptr`
__device__ __forceinline__ void read_entry_row(const DevModel& M, i32 eptr, i32* row, int nf) {
  u32 pos = (u32)(eptr >> 1);
  VarintWindow<const u8 JPP_GLOBAL*> win(as_global(M.entry_data), pos);
  for (int k = 0; k < nf; ++k) row[k] = (i32)win.next(pos);
}

__device__ __forceinline__ bool dic_pattern_matches(const DevModel& M, const UnkMaker& mk, i32 value) {
  bool match = false;
  for_each_entry(M, value, [&](i32 ptr) {
    if (match) return;
    i32 row[kMaxDicFeatures];
    read_entry_row(M, ptr, row, M.num_features);
    bool same = true;
    for (int f = 0; f < M.num_features; ++f) {
      if ((mk.pattern_mask >> f) & 1) {
        if (row[f] != mk.tmpl[f]) same = false;
      }
    }
    if (same) match = true;
  });
  return match;
}

template <typename V>
__device__ __forceinline__ void numeric_maker(const DevModel& M, const UnkMaker& mk, const V& S,
                                              u32 i, const WalkInfo& w, SeedSink& out) {
  u32 length = num_find_longest(S, i, mk.char_class);
  if (length == 0) return;
  if (w.valid) {
    // the number is not a dictionary key: the walk already knows which of the two UNK flavours applies
    // (a key needs the entry list, i.e. the real walk below)
    const int ws = walk_status(w, length);
    if (ws != TRIE_OK) {
      emit_unk(M, mk, S, out, i, i + length, ws == TRIE_NONODE);
      return;
    }
  }
  TrieCursor c{0, 0};
  int st = TRIE_NONODE;
  bool nonode = false;
  for (u32 k = i; k < i + length; ++k) {
    st = step_cp(M, S, c, k);
    if (st == TRIE_NONODE) nonode = true;
  }
  if (nonode) st = TRIE_NONODE;
  u32 e = i + length;
  if (st == TRIE_NONODE) {
    emit_unk(M, mk, S, out, i, e, true);
  } else if (st == TRIE_NOLEAF) {
    emit_unk(M, mk, S, out, i, e, false);
  } else {
    if (!dic_pattern_matches(M, mk, c.value)) emit_unk(M, mk, S, out, i, e, false);
  }
}

// ---- onomatopoeia ----------------------------------------------------------
template <typename V>
__device__ __forceinline__ u32 onoma_find(const V& S, u32 start, i32 cc) {
  if (start + 4 >= S.n) return 0;
  if ((S.cls[start] & cc) == 0) return 0;
  i32 c1 = S.cls[start];
  if ((S.cls[start + 1] & c1) == 0) return 0;
  u32 pattern = 0;
  for (u32 half = 2; half * 2 <= 8 && start + half * 2 - 1 < S.n; ++half) {
    if ((S.cls[start + half] & c1) == 0) return pattern;
    if (S.cp[start] != S.cp[start + half]) continue;
    bool ok = true;
    for (u32 p = 1; p < half; ++p) {
      if (S.cp[start + p] != S.cp[start + half + p]) {
        ok = false;
        break;
      }
    }
    if (ok) pattern |= (1u << half) & 0x1c;
  }
  return pattern;
}

template <typename V>
__device__ __forceinline__ void onoma_maker(const DevModel& M, const UnkMaker& mk, const V& S,
                                            u32 i, SeedSink& out) {
  u32 pattern = onoma_find(S, i, mk.char_class);
  if (pattern == 0) return;
  TrieCursor c{0, 0};
  u32 next = i;
  int st = TRIE_NONODE;
  for (u32 half = 2; half * 2 <= 8; ++half) {
    if (pattern & ((1u << half) & 0x1c)) {
      for (; next < i + half * 2; ++next) st = step_cp(M, S, c, next);
      if (st == TRIE_NONODE) emit_unk(M, mk, S, out, i, i + half * 2, true);
      else if (st == TRIE_NOLEAF) emit_unk(M, mk, S, out, i, i + half * 2, false);
    }
  }
}

template <typename V>
__device__ __forceinline__ void run_maker(const DevModel& M, const UnkMaker& mk, const V& S, u32 i,
                                          const WalkInfo& w, SeedSink& out) {
  switch (mk.type) {
    case UNK_SINGLE: single_maker(M, mk, S, i, w, out); break;
    case UNK_CHUNKING: chunking_maker(M, mk, S, i, w, out); break;
    case UNK_NUMERIC: numeric_maker(M, mk, S, i, w, out); break;
    case UNK_ONOMATOPOEIA: onoma_maker(M, mk, S, i, out); break;
    default: break;  // UNK_NORMALIZE is handled by k_norm
  }
}

// ---- normalize (charlattice traversal from one start) ----------------------
struct NormState {
  u32 node_pos;
  i32 value;
  u16 end;
  u16 flags;
  u8 key_pos;
  u8 last;
};
struct NormResult {
  i32 ptr;
  u16 flags;
  u16 end;
};

__device__ __forceinline__ int utf8_encode3(u32 cp, u8* b) {
  // every replacement codepoint of CharDb is a 3-byte kana
  b[0] = (u8)(0xE0 | (cp >> 12));
  b[1] = (u8)(0x80 | ((cp >> 6) & 0x3f));
  b[2] = (u8)(0x80 | (cp & 0x3f));
  return 3;
}

// The reference's traversal has no bound on its state and result lists (charlattice.cc:266-353).  The per-lane arrays
// hold kMaxNormStates / kMaxNormResults; a start that needs more repeats the traversal in an HBM slice of these sizes:
constexpr int kBigNormStates = 2048;
constexpr int kBigNormResults = 8192;
constexpr u32 kNormSlotGroups = 16;   // x 64 lanes x 128 KB = 128 MB, allocated when the first batch needs it ... (see jppgpu_api.cc)
__host__ __device__ constexpr size_t norm_slice_bytes() { return (size_t)kBigNormResults * 8 + 2 * (size_t)kBigNormStates * 16; }

// returns number of results or -1 on capacity overflow; results are sorted and uniqued.
// a, b: two state lists of maxs entries; res: maxr entries
template <typename V>
__device__ inline int norm_lookup_into(const DevModel& M, const V& S, const ClNodes* cl, u32 start, NormResult* res, int maxr,
                                       NormState* a, NormState* b, int maxs) {
  NormState* s1 = a;
  NormState* s2 = b;
  int n1 = 0, n2 = 0, nres = 0;
  bool overflow = false;
  TrieCursor c{0, 0x7fffffff};
  int st = step_cp(M, S, c, start);
  if (st == TRIE_NONODE) return 0;
  s1[n1++] = NormState{c.node_pos, c.value, (u16)(start + 1), (u16)CL_ORIGINAL,
                       (u8)(S.boff[start + 1] - S.boff[start]), (u8)st};
  u32 step = start + 1;
  while (step < S.n && n1 > 0) {
    n2 = 0;
    const ClNodes& xn = cl[step];
    for (int k = 0; k < n1; ++k) {
      const NormState ps = s1[k];
      for (int w = -1; w < (int)xn.n; ++w) {
        u16 newFlag = w < 0 ? (u16)CL_ORIGINAL : xn.type[w];
        bool doStep = w < 0 ? true : (xn.type[w] & CL_DELETE) == 0;
        NormState ns = ps;
        u32 e = (u32)ps.end + 1;
        ns.end = (u16)(e < S.n ? e : S.n);
        ns.flags = ps.flags | newFlag;
        int status;
        if (doStep) {
          TrieCursor tc{ps.node_pos, ps.value};
          if (w < 0) {
            status = step_cp(M, S, tc, step);
            ns.key_pos = (u8)(S.boff[step + 1] - S.boff[step]);
          } else {
            u8 bytes[4];
            int nb = utf8_encode3(xn.cp[w], bytes);
            status = trie_step(as_global(M.trie), tc, bytes, nb);
            ns.key_pos = (u8)nb;
          }
          // a failed step leaves key_pos at the failing byte in the reference, but such
          // states are discarded, so only successful steps matter
          ns.node_pos = tc.node_pos;
          ns.value = tc.value;
        } else {
          status = ps.last;
        }
        if (status == TRIE_NONODE) continue;
        ns.last = (u8)status;
        if (status == TRIE_OK && ns.flags != CL_ORIGINAL) {
          u16 flags = ns.flags;
          if (newFlag & CL_DELETE) flags |= CL_DELETE_LAST;
          for_each_entry(M, ns.value, [&](i32 ptr) {
            if (nres < maxr) res[nres++] = NormResult{ptr, flags, ns.end};
            else overflow = true;
          });
        }
        if (n2 < maxs) s2[n2++] = ns;
        else overflow = true;
      }
    }
    ++step;
    // consecutive-duplicate removal (charlattice.cc:266-296 + :336-345)
    int m = 0;
    for (int k = 0; k < n2; ++k) {
      if (m > 0) {
        const NormState& p = s2[m - 1];
        const NormState& q = s2[k];
        if (p.end == q.end && p.flags == q.flags && p.node_pos == q.node_pos && p.key_pos == q.key_pos &&
            p.value == q.value)
          continue;
      }
      s2[m++] = s2[k];
    }
    NormState* t = s1;
    s1 = s2;
    s2 = t;
    n1 = m;
  }
  if (overflow) return -1;
  if (nres == 0) return 0;
  // util::sort == std::sort (charlattice.cc:347-353): exact libstdc++ order incl. ties
  std_sort(res, res + nres, [](const NormResult& c1, const NormResult& c2) {
    return c1.end == c2.end ? c1.ptr < c2.ptr : c1.end < c2.end;
  });
  int m = 0;
  for (int x = 0; x < nres; ++x) {
    if (m > 0 && res[m - 1].ptr == res[x].ptr && res[m - 1].end == res[x].end) continue;
    res[m++] = res[x];
  }
  return m;
}

template <typename V>
__device__ inline int norm_lookup(const DevModel& M, const V& S, const ClNodes* cl, u32 start, NormResult* res) {
  NormState a[kMaxNormStates], b[kMaxNormStates];
  return norm_lookup_into(M, S, cl, start, res, kMaxNormResults, a, b, kMaxNormStates);
}

// The same in this lane's HBM slice.  A pool slot is 64 slices, one per lane, and is owned by ONE WAVEFRONT at a time
// (norm_slot_acquire / norm_slot_release, wave-uniform): the lock is taken by lane 0 while no lane of the wavefront is
// inside a critical section, a wavefront holds at most one slot and never waits while it holds one, so the waits cannot
// form a cycle.  (Until round 3 every lane took a lock of its own inside divergent code: a lane that had its lock stayed
// masked off until its siblings left their spin loops, so two wavefronts of the same slot could take their lanes' locks
// crosswise and hang -- ADVICE r03.)  The results stay in the slice until the caller releases the slot.
// Returns the results through *out; -1: no pool, or beyond the slice as well.
template <typename V>
__device__ inline int norm_lookup_big(const Batch& B, const DevModel& M, const V& S, const ClNodes* cl, u32 start, u32 slot,
                                      const NormResult** out) {
  if (B.norm_slots == 0) return -1;
  const u32 li = slot * 64u + (threadIdx.x & 63u);
  unsigned char* base = B.norm_scratch + (size_t)li * norm_slice_bytes();
  NormResult* res = reinterpret_cast<NormResult*>(base);
  NormState* a = reinterpret_cast<NormState*>(base + (size_t)kBigNormResults * 8);
  NormState* b = a + kBigNormStates;
  *out = res;
  return norm_lookup_into(M, S, cl, start, res, kBigNormResults, a, b, kBigNormStates);
}
// wave-uniform: every lane of the (single-wavefront) workgroup calls these together
__device__ inline u32 norm_slot_acquire(const Batch& B) {
  const u32 slot = B.norm_slots ? blockIdx.x % B.norm_slots : 0u;
  if (B.norm_slots != 0 && (threadIdx.x & 63u) == 0) {
    while (atomicCAS(&B.norm_locks[slot * 64u], 0u, 1u) != 0u) {
#if !defined(JPP_EMU)
      __builtin_amdgcn_s_sleep(32);
#endif
    }
    __threadfence();
  }
  __syncthreads();
  return slot;
}
__device__ inline void norm_slot_release(const Batch& B, u32 slot) {
  __threadfence();
  __syncthreads();
  if (B.norm_slots != 0 && (threadIdx.x & 63u) == 0) atomicExch(&B.norm_locks[slot * 64u], 0u);
}

// makePtr(surface, conf, eptr, feature): entry row of `eptr` with the replace
// fields overwritten by the hash of the *input* surface, placeholder = flags
template <typename V>
__device__ __forceinline__ void norm_emit(const DevModel& M, const UnkMaker& mk, const V& S,
                                          SeedSink& out, u32 s, const NormResult& r) {
  if (!out.emit) {
    ++out.n;
    return;
  }
  i32 ph[2] = {0, 0};
  if (mk.placeholder >= 0 && mk.placeholder < 2) ph[mk.placeholder] = (i32)r.flags;
  out.unk(r.ptr, surface_hash(S, s, r.end), ph[0], ph[1], (u32)M.makers[M.norm_maker].spec_index, M.makers[M.norm_maker].rank, s, r.end);
}

// MODE 0: count (writes pos_cntA / pos_cnt2); MODE 1: emit stage 1; MODE 2: emit stage 1+2
// for sentences with the stage-2 flag (into their relocated region).
// (the trie walk is a chain of dependent loads: 8 wavefronts per SIMD at 64 VGPRs and a few spilled registers beat
// 4 at 97 -- count pass 642 -> 491 us, emit pass 722 -> 668 us; the rarely taken stage-2 pass gains nothing)
// the seeds of one sentence (see k_seeds), on either view of it
template <int MODE, typename V>
__device__ __forceinline__ void seeds_of_sentence(const Batch& B, const DevModel& M, const V& S, u32 s, u32 g0, u32 bb0) {
  u64 nbase = MODE == 0 ? 0 : B.node_base[s];
  for (u32 i = threadIdx.x; i < S.n; i += blockDim.x) {
    SeedSink out;
    out.emit = MODE != 0;
    out.n = 0;
    out.ends = 0;
    if (MODE != 0) {
      u32 first = B.bnd_first[bb0 + i + 2];
      out.ni = B.node_info + nbase + first;
      out.na = B.node_aux + nbase + first;
    } else {
      out.ni = nullptr;
      out.na = nullptr;
    }
    WalkInfo w{0, 0, false};
    if (MODE == 0) {
      w = dic_seeds(M, S, i, out, &B.pos_walk[g0 + i]);
    } else {
      const WalkCache* wc = &B.pos_walk[g0 + i];
      w = wc->cached ? dic_seeds_replay(M, wc, wc->leaf, wc->ok_len, i, out) : dic_seeds(M, S, i, out);
    }
    for (int m = 0; m < M.n_stage1; ++m) run_maker(M, M.makers[m], S, i, w, out);
    if (MODE == 0) {
      B.pos_cnt1[g0 + i] = (u16)(out.n > 0xffff ? 0xffff : out.n);
      B.pos_ends[g0 + i] = out.ends;   // stage-1 ends only; k_norm<0> adds the normalized nodes'
      u32 n1 = out.n;
      for (int m = M.n_stage1; m < M.n_unk; ++m) run_maker(M, M.makers[m], S, i, w, out);
      B.pos_cnt2[g0 + i] = (u16)(out.n - n1);
    } else if (MODE == 2) {
      // stage-2 nodes follow the stage-1 nodes (incl. normalize) of this start
      u32 skip = B.pos_cntN[g0 + i];
      out.n += skip;
      for (int m = M.n_stage1; m < M.n_unk; ++m) run_maker(M, M.makers[m], S, i, w, out);
    }
  }
}

template <int MODE, int WAVES = (MODE == 2 ? 4 : 8)>
__global__ void __launch_bounds__(64) JPP_WAVES_PER_EU(WAVES) k_seeds(Batch B, const DevModel* __restrict__ Mp) {
  const DevModel& M = *Mp;
  u32 s = blockIdx.x;
  if (B.sent_status[s] != ST_OK) return;
  if (MODE == 2 && (B.sent_flags[s] & 2) == 0) return;
  if (MODE == 1 && (B.sent_flags[s] & 2) != 0) return;  // known from the count pass to need stage 2: emitted once, by MODE 2
  u32 off = B.byte_off[s];
  u32 g0 = off + s;
  u32 bb0 = off + 4 * s;
  const u32 bytes = B.byte_off[s + 1] - off;
  SentViewG S{B.text + off, B.cp_code + g0, B.cp_class + g0, B.cp_boff + g0, B.sent_ncp[s]};
  __shared__ SentLds L;
  if (S.n <= kSentLdsCp && bytes <= kSentLdsBytes) {
    stage_sentence(L, S, bytes);
    seeds_of_sentence<MODE>(B, M, lds_view(L, S.n), s, g0, bb0);
  } else {
    seeds_of_sentence<MODE>(B, M, S, s, g0, bb0);
  }
}

// normalize maker: same modes.  Its nodes are the last stage-1 nodes of a start.
template <int MODE, typename V>
__device__ __forceinline__ void norm_of_sentence(const Batch& B, const DevModel& M, const V& S, u32 s, u32 g0, u32 bb0, u32 n) {
  const UnkMaker& mk = M.makers[M.norm_maker];
  u64 nbase = MODE == 0 ? 0 : B.node_base[s];
  static_assert(sizeof(NormResult) == 8, "cached as one 64-bit word");
  NormResult* cache = reinterpret_cast<NormResult*>(B.pos_norm) + (u64)g0 * kNormCache;
  // Count pass: a traversal from start i only ever looks at the charlattice nodes of the positions i + 1 .. i + d, d
  // = the depth of the plain dictionary walk from i (the un-normalised state dies with the trie path, and a state
  // can only leave that path THROUGH a charlattice node).  k_seeds<0> recorded d; sentences of up to 64 codepoints
  // keep the positions that have charlattice nodes in one ballot, and most starts are dismissed without a walk.
  u64 clmask = ~u64{0};
  if (MODE == 0 && n <= 64) clmask = wave_ballot(threadIdx.x < n && B.cl_nodes[g0 + threadIdx.x].n != 0);
  // The trip count is uniform over the wavefront (lanes beyond n idle along): whether any lane of a round needs the HBM
  // slice is decided with one ballot, and the slot is taken and released by the wavefront as a whole.
  for (u32 i0 = 0; i0 < n; i0 += blockDim.x) {
    const u32 i = i0 + threadIdx.x;
    NormResult res_local[kMaxNormResults];
    const NormResult* res = res_local;
    int nr = 0;
    bool act = i < n;
    bool needBig = false;
    if (act && MODE == 0) {
      const u32 depth = B.pos_walk[g0 + i].ok_len;
      const bool reachable = depth >= 63 || i >= 63 || ((clmask >> (i + 1)) & ((u64{1} << depth) - 1)) != 0;
      if (!reachable) {
        B.pos_cntN[g0 + i] = 0;
        act = false;
      } else {
        nr = norm_lookup(M, S, B.cl_nodes + g0, i, res_local);
        needBig = nr < 0;   // beyond the per-lane arrays: once more in this lane's HBM slice
      }
    } else if (act) {
      // the count pass left the number of results and, for short lists, the results themselves
      nr = (int)B.pos_cntN[g0 + i];
      if (nr == 0) {
        act = false;
      } else if (nr <= kNormCache) {
        for (int k = 0; k < nr; ++k) res_local[k] = cache[(u64)i * kNormCache + k];
      } else if (nr <= kMaxNormResults) {
        nr = norm_lookup(M, S, B.cl_nodes + g0, i, res_local);
        needBig = nr < 0;   // (more states than the per-lane lists hold, though the results fit)
      } else {
        needBig = true;
      }
    }
    const bool anyBig = wave_ballot(needBig) != 0;   // uniform
    u32 slot = 0;
    if (anyBig) {
      slot = norm_slot_acquire(B);
      if (needBig) nr = norm_lookup_big(B, M, S, B.cl_nodes + g0, i, slot, &res);
    }
    if (act && MODE == 0) {
      if (nr < 0 || nr > 0xffff) {
        atomicMax(&B.sent_status[s], (i32)ST_CAPACITY);
        nr = 0;
      }
      B.pos_cntN[g0 + i] = (u16)nr;
      u64 ends = 0;
      for (int k = 0; k < nr; ++k) ends |= res[k].end < 64 ? (u64{1} << res[k].end) : u64{0};
      if (ends) B.pos_ends[g0 + i] |= ends;   // k_seeds<0> of this launch sequence wrote the word already
      if (nr <= kNormCache)
        for (int k = 0; k < nr; ++k) cache[(u64)i * kNormCache + k] = res[k];
    } else if (act) {
      if (nr < 0) nr = 0;   // (cannot happen: the count pass went through the same traversal)
      SeedSink out;
      out.emit = true;
      out.ends = 0;
      u32 first = B.bnd_first[bb0 + i + 2] + B.pos_cnt1[g0 + i];
      out.ni = B.node_info + nbase + first;
      out.na = B.node_aux + nbase + first;
      out.n = 0;
      for (int k = 0; k < nr; ++k) norm_emit(M, mk, S, out, i, res[k]);
    }
    if (anyBig) norm_slot_release(B, slot);
  }
}

template <int MODE>
__global__ void __launch_bounds__(64) k_norm(Batch B, const DevModel* __restrict__ Mp) {
  const DevModel& M = *Mp;
  u32 s = blockIdx.x;
  if (B.sent_status[s] != ST_OK) return;
  u32 off = B.byte_off[s];
  u32 g0 = off + s;
  u32 bb0 = off + 4 * s;
  u32 n = B.sent_ncp[s];
  bool applicable = M.norm_maker >= 0 && (B.sent_flags[s] & 1);
  if (MODE == 0 && !applicable) {
    for (u32 i = threadIdx.x; i < n; i += blockDim.x) B.pos_cntN[g0 + i] = 0;
    return;
  }
  if (!applicable) return;
  if (MODE == 2 && (B.sent_flags[s] & 2) == 0) return;
  if (MODE == 1 && (B.sent_flags[s] & 2) != 0) return;
  const u32 bytes = B.byte_off[s + 1] - off;
  SentViewG S{B.text + off, B.cp_code + g0, B.cp_class + g0, B.cp_boff + g0, n};
  __shared__ SentLds L;
  if (n <= kSentLdsCp && bytes <= kSentLdsBytes) {
    stage_sentence(L, S, bytes);
    norm_of_sentence<MODE>(B, M, lds_view(L, n), s, g0, bb0, n);
  } else {
    norm_of_sentence<MODE>(B, M, S, s, g0, bb0, n);
  }
}

}  // namespace jpp

#endif  // JPP_K_SEEDS_H
