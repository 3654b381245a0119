// Kernel 1: UTF-8 decode + character classes + charlattice pre-parse.
// One wavefront per sentence: a lane per byte for the decoder, a lane per codepoint for the charlattice parse
// (round 3; one lane per sentence before: 1024 wavefronts of serial byte loads, 0.29 ms).
//
// Reference behaviour reproduced:
//   AnalysisInput::reset              src/core/analysis/analysis_input.cc:12-33
//   chars::preprocessRawData          src/util/characters.cc:259-276
//   charlattice::CharLattice::Parse   src/core/analysis/charlattice.cc:197-264
//   (+ helper predicates charlattice.cc:84-195 and the CharDb maps :14-77)
#ifndef JPP_K_DECODE_H
#define JPP_K_DECODE_H

#include "jpp_device.h"

namespace jpp {

// charlattice::Modifiers (src/core/analysis/charlattice.h:22-34)
enum ClMod : u32 {
  CL_ORIGINAL = 0x1, CL_REPLACE_SMALLKANA = 0x2, CL_REPLACE = 0x4, CL_DELETE = 0x8,
  CL_REPLACE_PROLONG = 0x10, CL_DELETE_LAST = 0x20, CL_DELETE_PROLONG = 0x40,
  CL_DELETE_HASTSUON = 0x80, CL_DELETE_SMALLKANA = 0x100, CL_REPLACE_EROW_WITH_E = 0x200,
};

// CharDb::prolongedMap (charlattice.cc:37-50); 0 = not in map
__device__ __forceinline__ u32 cl_prolonged(u32 c) {
  switch (c) {
    case U'か': case U'が': case U'ば': case U'ま': case U'ゃ':
      return U'あ';
    case U'い': case U'き': case U'し': case U'ち': case U'に': case U'ひ': case U'じ':
    case U'け': case U'せ': case U'へ': case U'め': case U'れ': case U'げ': case U'ぜ':
    case U'で': case U'べ': case U'ぺ': case U'え': case U'ね':
      return U'い';
    case U'く': case U'す': case U'つ': case U'ふ': case U'ゆ': case U'ぐ': case U'ず':
    case U'ぷ': case U'ゅ': case U'お': case U'こ': case U'そ': case U'と': case U'の':
    case U'ほ': case U'も': case U'よ': case U'ろ': case U'ご': case U'ぞ': case U'ど':
    case U'ぼ': case U'ぽ': case U'ょ':
      return U'う';
    default:
      return 0;
  }
}

// CharDb::prolongedMapForErow (charlattice.cc:51-54)
__device__ __forceinline__ u32 cl_prolonged_erow(u32 c) {
  switch (c) {
    case U'え': case U'け': case U'げ': case U'せ': case U'ぜ': case U'て': case U'で':
    case U'ね': case U'へ': case U'べ': case U'め': case U'れ':
      return U'え';
    default:
      return 0;
  }
}

// CharDb::lower2upper (charlattice.cc:33-36)
__device__ __forceinline__ u32 cl_lower2upper(u32 c) {
  switch (c) {
    case U'ぁ': return U'あ';
    case U'ぃ': return U'い';
    case U'ぅ': return U'う';
    case U'ぇ': return U'え';
    case U'ぉ': return U'お';
    case U'ゎ': return U'わ';
    case U'ヶ': return U'ケ';
    case U'ケ': return U'ヶ';
    default: return 0;
  }
}

// CharDb::lowerMap (charlattice.cc:56-71); duplicate keys in the initializer
// list keep their FIRST value (FlatMap::insert does not overwrite).
__device__ __forceinline__ u32 cl_lower_map(u32 c) {
  switch (c) {
    case U'か': case U'さ': case U'た': case U'な': case U'は': case U'ま': case U'や':
    case U'ら': case U'わ': case U'が': case U'ざ': case U'だ': case U'ば': case U'ぱ':
      return U'ぁ';
    case U'い': case U'し': case U'に': case U'り': case U'ぎ': case U'じ': case U'ね':
    case U'れ': case U'ぜ':
      return U'ぃ';
    case U'う': case U'く': case U'す': case U'ふ': case U'む': case U'る': case U'よ':
      return U'ぅ';
    case U'け': case U'せ': case U'て': case U'め': case U'で':
      return U'ぇ';
    case U'こ': case U'そ': case U'の': case U'も': case U'ろ': case U'ぞ': case U'ど':
      return U'ぉ';
    default:
      return 0;
  }
}

__device__ __forceinline__ bool cl_lower_list(u32 c) {
  return c == U'ぁ' || c == U'ぃ' || c == U'ぅ' || c == U'ぇ' || c == U'ぉ';
}

// CharLattice::Parse for one position (charlattice.cc:197-264): the extra nodes of the position and whether the next
// position sees a preceding deletion.  `preDel`: the previous position produced a deletion node.
struct ClEval {
  ClNodes cn;
  bool nextPreDel;
};
__device__ __forceinline__ ClEval cl_eval(u32 pos, u32 n, u32 cur, i32 cls, u32 prev, i32 prevCls, u32 nextCp, i32 nextCls,
                                          bool preDel) {
  ClEval r;
  ClNodes& cn = r.cn;
  cn.n = 0;
  cn.cp[0] = cn.cp[1] = cn.cp[2] = 0;
  cn.type[0] = cn.type[1] = cn.type[2] = 0;
  r.nextPreDel = false;
  if (cls & CC_FAMILY_DOUBLE) {
    bool choon = pos > 0 && (cls & CC_CHOON);
    u32 sub = choon ? cl_prolonged(prev) : 0;
    if (sub != 0) {
      cn.cp[cn.n] = sub;
      cn.type[cn.n] = CL_REPLACE | CL_REPLACE_PROLONG;
      cn.n++;
      u32 sub2 = cl_prolonged_erow(prev);
      if (sub2 != 0) {
        cn.cp[cn.n] = sub2;
        cn.type[cn.n] = CL_REPLACE | CL_REPLACE_PROLONG | CL_REPLACE_EROW_WITH_E;
        cn.n++;
      }
    } else {
      u32 up = cl_lower2upper(cur);
      if (up != 0) {
        cn.cp[cn.n] = up;
        cn.type[cn.n] = CL_REPLACE | CL_REPLACE_SMALLKANA;
        cn.n++;
      }
    }
    // deletions
    bool removableProlong = false;
    if (pos >= 1 && (cls & CC_CHOON)) {
      removableProlong = preDel || (prevCls & CC_FAMILY_PROLONGABLE) != 0;
    }
    u16 delType = 0;
    if (removableProlong) {
      delType = CL_DELETE | CL_DELETE_PROLONG;
    } else {
      bool hatsuon = false;
      if (pos != 0 && (cur == 0x3063 || cur == 0x30C3)) {
        if (preDel) {
          hatsuon = true;
        } else if (pos + 1 >= n) {
          hatsuon = true;
        } else {
          const i32 always = CC_SPACE | CC_IDEOGRAPHIC_PUNC | CC_FIGURE | CC_PERIOD | CC_MIDDLE_DOT |
                             CC_ALPH | CC_SYMBOL | CC_BRACKET | CC_SLASH | CC_COLON | CC_COMMA;
          if (nextCls & always) hatsuon = true;
          else if (cur == nextCp) hatsuon = true;
          else hatsuon = ((nextCls & prevCls & cls) & CC_FAMILY_FULL_KANA) != 0;
        }
      }
      if (hatsuon) {
        delType = CL_DELETE | CL_DELETE_HASTSUON;
      } else if (pos != 0) {
        u32 lm = cl_lower_map(prev);
        bool youon = (lm != 0 && lm == cur) || (preDel && cl_lower_list(cur) && cur == prev);
        if (youon) delType = CL_DELETE | CL_DELETE_SMALLKANA;
      }
    }
    if (delType != 0) {
      cn.cp[cn.n] = 0;
      cn.type[cn.n] = delType;
      cn.n++;
      r.nextPreDel = true;
    }
  }
  return r;
}

// One wavefront per sentence, four per workgroup.
//  * UTF-8: one lane per byte.  Every byte that is not a continuation byte starts a codepoint and decodes it; the
//    codepoint index is the number of starts before it (ballot + popcount).  The sequential decoder accepts the input
//    iff every start decodes and the codepoints tile the bytes (each start sits where the previous one ended, the last
//    one ends at the end) -- checked the same way, so the verdict (InvalidParameter on the first bad sequence) and
//    the decoded arrays are those of chars::preprocessRawData.
//  * CharLattice::Parse: one lane per codepoint.  The only thing a position takes from its predecessor's RESULT is
//    "did it produce a deletion" (preDel); each lane evaluates its position for both answers, the chain is resolved
//    by a prefix scan over the 2-bit functions preDel -> nextPreDel, and the lane keeps the variant that applies.
constexpr int kDecWaves = 4;
__device__ __forceinline__ u32 cl_compose(u32 first, u32 then) {   // apply `first`, then `then` (bit q = image of q)
  const u32 r0 = (first & 1u) ? (then >> 1) & 1u : then & 1u;
  const u32 r1 = (first & 2u) ? (then >> 1) & 1u : then & 1u;
  return r0 | (r1 << 1);
}

constexpr u32 kDecLds = 256;   // codepoints of a sentence kept in LDS for the charlattice parse
__global__ void __launch_bounds__(64 * kDecWaves) k_decode(Batch B, Config cfg) {
  const int lane = (int)(threadIdx.x & 63);
  const int wv = (int)(threadIdx.x >> 6);
  const u32 s = blockIdx.x * kDecWaves + (u32)wv;
  __shared__ u32 l_cp_all[kDecWaves][kDecLds];
  __shared__ i32 l_cls_all[kDecWaves][kDecLds];
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    for (int q = 0; q < 16; ++q) B.gstats[q] = 0;   // ([8]: overflow flag of the one-enqueue path, k_lattice.h)
  }
  if (s >= B.n_sent) return;
  const u32 off = B.byte_off[s];
  const u32 len = B.byte_off[s + 1] - off;
  const u32 g0 = off + s;  // codepoint index base (one spare slot per sentence)
  const u8* txt = B.text + off;
  if (lane == 0) B.sent_flags[s] = 0;
  if (len > (u32)cfg.max_input_bytes) {
    if (lane == 0) {
      B.sent_status[s] = ST_TOO_LONG;
      B.sent_ncp[s] = 0;
    }
    return;
  }
  const u64 below = (u64{1} << lane) - 1;
  u32 n = 0, endOfLast = 0;
  bool bad = false;
  for (u32 p0 = 0; p0 < len; p0 += 64) {
    const u32 p = p0 + (u32)lane;
    const bool act = p < len;
    const u32 b0 = act ? txt[p] : 0x80u;
    const bool isStart = act && (b0 & 0xC0u) != 0x80u;
    u32 cp = 0;
    int l = 0;
    if (isStart) l = utf8_decode(txt + p, (int)(len - p), cp);
    const u64 starts = wave_ballot(isStart);
    bad = bad || wave_ballot(isStart && l == 0) != 0;
    // the start before this one (in this chunk, else the last one of the earlier chunks) must end here
    const u64 before = starts & below;
    const int prevLane = before ? 63 - __builtin_clzll(before) : 0;
    const u32 prevEnd = wave_shfl_u32(p + (u32)l, prevLane);
    const u32 expect = before ? prevEnd : endOfLast;
    bad = bad || wave_ballot(isStart && p != expect) != 0;
    if (isStart) {
      const u32 idx = n + (u32)popc64(before);
      const i32 cc = char_class(cp);
      B.cp_code[g0 + idx] = cp;
      B.cp_class[g0 + idx] = cc;
      B.cp_boff[g0 + idx] = (u16)p;
      if (idx < kDecLds) {
        l_cp_all[wv][idx] = cp;
        l_cls_all[wv][idx] = cc;
      }
    }
    if (starts) {
      endOfLast = wave_shfl_u32(p + (u32)l, 63 - __builtin_clzll(starts));
      n += (u32)popc64(starts);
    }
  }
  bad = bad || endOfLast != len;
  if (bad) {
    if (lane == 0) {
      B.sent_status[s] = ST_BAD_UTF8;
      B.sent_ncp[s] = 0;
    }
    return;
  }
  if (lane == 0) {
    B.cp_boff[g0 + n] = (u16)len;
    B.sent_ncp[s] = n;
    B.sent_status[s] = ST_OK;
  }
  // The parse looks at the neighbouring codepoints: they are kept in LDS (sentences of up to kDecLds codepoints) --
  // reading them back from HBM needs a device-scope fence per wavefront, which cost 2 ms per batch when tried.
  // Longer sentences take the fence.
  const bool viaLds = n <= kDecLds;
  if (!viaLds) {
#if !defined(JPP_EMU)
    __threadfence();
#endif
  }
  wave_sync();
  const u32* l_cp = l_cp_all[wv];
  const i32* l_cls = l_cls_all[wv];
  auto cpAt = [&](u32 q) -> u32 { return viaLds ? l_cp[q] : B.cp_code[g0 + q]; };
  auto clsAt = [&](u32 q) -> i32 { return viaLds ? l_cls[q] : B.cp_class[g0 + q]; };

  // ---- CharLattice::Parse ----
  u32 carry = 0;        // preDel entering the chunk
  u32 notNormal = 0;
  for (u32 q0 = 0; q0 < n; q0 += 64) {
    const u32 pos = q0 + (u32)lane;
    const bool act = pos < n;
    const u32 cur = act ? cpAt(pos) : 0;
    const i32 cls = act ? clsAt(pos) : 0;
    const u32 prev = (act && pos > 0) ? cpAt(pos - 1) : 0;
    const i32 prevCls = (act && pos > 0) ? clsAt(pos - 1) : 0;
    const u32 nextCp = (act && pos + 1 < n) ? cpAt(pos + 1) : 0;
    const i32 nextCls = (act && pos + 1 < n) ? clsAt(pos + 1) : 0;
    const ClEval e0 = cl_eval(pos, n, cur, cls, prev, prevCls, nextCp, nextCls, false);
    const ClEval e1 = cl_eval(pos, n, cur, cls, prev, prevCls, nextCp, nextCls, true);
    // inclusive scan of the functions preDel -> nextPreDel (inactive lanes: identity)
    u32 f = act ? ((e0.nextPreDel ? 1u : 0u) | (e1.nextPreDel ? 2u : 0u)) : 2u;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const u32 o = wave_shfl_u32(f, lane >= d ? lane - d : lane);
      if (lane >= d) f = cl_compose(o, f);
    }
    // preDel of this lane = (composite of the lanes before it)(carry)
    const u32 fprev = wave_shfl_u32(f, lane > 0 ? lane - 1 : 0);
    const u32 preDel = lane == 0 ? carry : (fprev >> carry) & 1u;
    const ClEval& e = preDel ? e1 : e0;
    if (act) B.cl_nodes[g0 + pos] = e.cn;
    notNormal += wave_sum_u32(act ? (u32)e.cn.n : 0u);
    carry = (wave_shfl_u32(f, 63) >> carry) & 1u;
  }
  if (notNormal != 0 && lane == 0) B.sent_flags[s] = 1;
}

}  // namespace jpp

#endif  // JPP_K_DECODE_H
