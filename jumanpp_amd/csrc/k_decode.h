// Kernel 1: UTF-8 decode + character classes + charlattice pre-parse.
// One lane per sentence (the work is ~120 bytes of strictly sequential
// decoding; 64k sentences give 1024 full waves).
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

__global__ void k_decode(Batch B, Config cfg) {
  u32 s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s == 0) {
    for (int q = 0; q < 8; ++q) B.gstats[q] = 0;
  }
  if (s >= B.n_sent) return;
  u32 off = B.byte_off[s];
  u32 len = B.byte_off[s + 1] - off;
  u32 g0 = off + s;  // codepoint index base (one spare slot per sentence)
  const u8* txt = B.text + off;
  B.sent_flags[s] = 0;
  if (len > (u32)cfg.max_input_bytes) {
    B.sent_status[s] = ST_TOO_LONG;
    B.sent_ncp[s] = 0;
    return;
  }
  u32 n = 0, p = 0;
  while (p < len) {
    u32 cp;
    int l = utf8_decode(txt + p, (int)(len - p), cp);
    if (l == 0) {
      B.sent_status[s] = ST_BAD_UTF8;
      B.sent_ncp[s] = 0;
      return;
    }
    B.cp_code[g0 + n] = cp;
    B.cp_class[g0 + n] = char_class(cp);
    B.cp_boff[g0 + n] = (u16)p;
    p += l;
    ++n;
  }
  B.cp_boff[g0 + n] = (u16)len;
  B.sent_ncp[s] = n;
  B.sent_status[s] = ST_OK;

  // ---- CharLattice::Parse ----
  bool preDel = false;
  u32 notNormal = 0;
  for (u32 pos = 0; pos < n; ++pos) {
    ClNodes cn;
    cn.n = 0;
    u32 cur = B.cp_code[g0 + pos];
    i32 cls = B.cp_class[g0 + pos];
    bool nextPreDel = false;
    if (cls & CC_FAMILY_DOUBLE) {
      u32 prev = pos > 0 ? B.cp_code[g0 + pos - 1] : 0;
      i32 prevCls = pos > 0 ? B.cp_class[g0 + pos - 1] : 0;
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
            u32 nextCp = B.cp_code[g0 + pos + 1];
            i32 nextCls = B.cp_class[g0 + pos + 1];
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
        nextPreDel = true;
      }
    }
    notNormal += cn.n;
    B.cl_nodes[g0 + pos] = cn;
    preDel = nextPreDel;
  }
  if (notNormal != 0) B.sent_flags[s] = 1;
}

}  // namespace jpp

#endif  // JPP_K_DECODE_H
