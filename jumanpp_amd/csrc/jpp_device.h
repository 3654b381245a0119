// Device-side primitives of the analysis hot path (hash, varint, double-array
// step, character classes, murmur3).  Each function cites the reference code
// whose *behaviour* it reproduces bit-for-bit; the implementation is written
// for one GPU lane working on HBM-resident model blobs.
#ifndef JPP_DEVICE_H
#define JPP_DEVICE_H

#include <cmath>

#include "jpp_rt.h"
#include "jpp_types.h"

namespace jpp {

// ----------------------------------------------------------------------------
// FastHashRot (reference src/util/fast_hash_rot.h:30-55, constants
// src/util/seahash.h:15-17): v = (state ^ x) * Mult; rotl(v, 32)
// ----------------------------------------------------------------------------
constexpr u64 kHashSeed0 = 0x16f11fe89b0d677cULL;
constexpr u64 kHashMult = 0x6eed0e9da4d94a4fULL;
// src/core/impl/feature_impl_types.h:21-24
constexpr u64 kPatternSeed = 0x7a11ed00000000ULL;
constexpr u64 kUnigramSeed = 0x5123a31421fULL;
constexpr u64 kBigramSeed = 0x5123a68442fULL;
constexpr u64 kTrigramSeed = 0x51239ab41f1fULL;

__host__ __device__ constexpr u64 hmix(u64 state, u64 x) {
  u64 v = (state ^ x) * kHashMult;
  return (v << 32) | (v >> 32);
}

// ----------------------------------------------------------------------------
// LEB128 varints (reference src/util/coded_io.h:130-158): value is read as u64
// and truncated by the caller (readInt<T>).
// ----------------------------------------------------------------------------
__device__ __forceinline__ u64 read_varint(const u8* __restrict__ p, u32& pos) {
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

// ----------------------------------------------------------------------------
// Darts-clone double-array step (reference src/core/dic/darts.h:55-76 unit
// decode, :512-533 traverse; src/core/dic/darts_trie.cc:104-117 status map).
// The traversal state is just node_pos; a NoNode leaves node_pos at the last
// matched byte, exactly like the reference (callers that keep stepping after
// NoNode -- the onomatopoeia maker -- rely on that).
// ----------------------------------------------------------------------------
enum TrieStatus : int { TRIE_OK = 0, TRIE_NOLEAF = 1, TRIE_NONODE = 2 };

__device__ __forceinline__ u32 da_offset(u32 unit) {
  return (unit >> 10) << ((unit & (1u << 9)) >> 6);
}
__device__ __forceinline__ u32 da_label(u32 unit) { return unit & ((1u << 31) | 0xFF); }

struct TrieCursor {
  u32 node_pos;
  i32 value;  // valid after TRIE_OK
};

__device__ __forceinline__ int trie_step(const u32* __restrict__ units, TrieCursor& c,
                                         const u8* __restrict__ bytes, int nbytes) {
  u32 id = c.node_pos;
  u32 unit = units[id];
  for (int k = 0; k < nbytes; ++k) {
    u32 b = bytes[k];
    id ^= da_offset(unit) ^ b;
    unit = units[id];
    if (da_label(unit) != b) return TRIE_NONODE;
    c.node_pos = id;
  }
  if (((unit >> 8) & 1) == 0) return TRIE_NOLEAF;
  u32 leaf = units[id ^ da_offset(unit)];
  c.value = (i32)(leaf & ((1u << 31) - 1));
  return TRIE_OK;
}

// ----------------------------------------------------------------------------
// Character classes (reference src/util/characters.h:31-75 bit values,
// src/util/characters.cc:135-257 getCodeType ladder -- the order of the
// tests is part of the behaviour, e.g. '[' is ALPH, not BRACKET).
// ----------------------------------------------------------------------------
enum CharClass : i32 {
  CC_SPACE = 0x1, CC_IDEOGRAPHIC_PUNC = 0x2, CC_KANJI = 0x4, CC_FIGURE = 0x8,
  CC_PERIOD = 0x10, CC_MIDDLE_DOT = 0x20, CC_COMMA = 0x40, CC_ALPH = 0x80,
  CC_SYMBOL = 0x100, CC_KATAKANA = 0x200, CC_HIRAGANA = 0x400, CC_KANJI_FIGURE = 0x800,
  CC_SLASH = 0x1000, CC_COLON = 0x2000, CC_ERA = 0x4000, CC_CHOON = 0x8000,
  CC_HANKAKU_KANA = 0x10000, CC_BRACKET = 0x20000, CC_FIGURE_EXCEPTION = 0x40000,
  CC_FIGURE_DIGIT = 0x80000, CC_SMALL_KANA = 0x100000,
  CC_FAMILY_FIGURE = CC_FIGURE | CC_PERIOD | CC_MIDDLE_DOT | CC_KANJI_FIGURE | CC_SLASH | CC_COLON,
  CC_FAMILY_NUM_PERIOD = CC_PERIOD | CC_MIDDLE_DOT,
  CC_FAMILY_KANA = CC_KATAKANA | CC_HIRAGANA | CC_HANKAKU_KANA | CC_SMALL_KANA,
  CC_FAMILY_DOUBLE = CC_KATAKANA | CC_HIRAGANA | CC_HANKAKU_KANA | CC_SMALL_KANA | CC_KANJI | CC_CHOON,
  CC_FAMILY_DIGITS = CC_FIGURE | CC_KANJI_FIGURE | CC_FIGURE_DIGIT,
  CC_FAMILY_EXCEPTION = CC_FIGURE | CC_KANJI_FIGURE | CC_FIGURE_EXCEPTION,
  CC_FAMILY_PROLONGABLE = CC_KANJI | CC_HIRAGANA | CC_KATAKANA,
  CC_FAMILY_FULL_KANA = CC_HIRAGANA | CC_KATAKANA,
};

__device__ __forceinline__ bool is_small_kana(u32 c) {
  switch (c) {
    case 0x3041: case 0x3043: case 0x3045: case 0x3047: case 0x3049: case 0x3063:
    case 0x3083: case 0x3085: case 0x3087: case 0x308E: case 0x3095: case 0x3096:
    case 0x30A1: case 0x30A3: case 0x30A5: case 0x30A7: case 0x30A9: case 0x30C3:
    case 0x30E3: case 0x30E5: case 0x30E7: case 0x30EE: case 0x30F5: case 0x30F6:
      return true;
    default:
      return false;
  }
}

// bracket set of characters.cc:38-102 written as ranges/pairs
__device__ __forceinline__ bool is_bracket(u32 c) {
  if (c < 0x100) return c == 0x28 || c == 0x29 || c == 0x5B || c == 0x5D || c == 0x7B || c == 0x7D;
  if (c >= 0x0F3A && c <= 0x0F3D) return true;
  if (c == 0x169B || c == 0x169C) return true;
  if (c == 0x2045 || c == 0x2046 || c == 0x207D || c == 0x207E || c == 0x208D || c == 0x208E) return true;
  if (c >= 0x2308 && c <= 0x230B) return true;
  if (c == 0x2329 || c == 0x232A) return true;
  if (c >= 0x2768 && c <= 0x2775) return true;
  if (c == 0x27C5 || c == 0x27C6) return true;
  if (c >= 0x27E6 && c <= 0x27EF) return true;
  if (c >= 0x2983 && c <= 0x2998) return true;
  if (c >= 0x29D8 && c <= 0x29DB) return true;
  if (c == 0x29FC || c == 0x29FD) return true;
  if (c >= 0x2E22 && c <= 0x2E29) return true;
  if (c >= 0x3008 && c <= 0x3011) return true;
  if (c >= 0x3014 && c <= 0x301B) return true;
  if (c >= 0xFE59 && c <= 0xFE5E) return true;
  if (c == 0xFF08 || c == 0xFF09 || c == 0xFF3B || c == 0xFF3D || c == 0xFF5B || c == 0xFF5D) return true;
  if (c == 0xFF5F || c == 0xFF60 || c == 0xFF62 || c == 0xFF63) return true;
  return false;
}

__device__ __forceinline__ i32 char_class(u32 code) {
  if (code == 0x20 || code == 0x3000 || code == 0xA0 || code == 0x1680 || code == 0x180E ||
      code == 0x202F || code == 0x205F || code == 0xFEFF || (0x2000 <= code && code <= 0x200B)) {
    return CC_SPACE;
  } else if (code > 0x3000 && code < 0x3003) {
    return CC_IDEOGRAPHIC_PUNC;
  } else if (0x337B <= code && code <= 0x337E) {
    return CC_SYMBOL | CC_ERA;
  } else if ((code > 0x303f && code < 0x30a0) || code == 0x309D || code == 0x309E || code == 0x309F) {
    return is_small_kana(code) ? (CC_HIRAGANA | CC_SMALL_KANA) : CC_HIRAGANA;
  } else if ((code > 0x309f && code < 0x30fb) || code == 0x30FD || code == 0x30FE || code == 0x30FF) {
    return is_small_kana(code) ? (CC_KATAKANA | CC_SMALL_KANA) : CC_KATAKANA;
  } else if (code == 0x30FC || code == 0x301C || code == 0xFF5E || code == 0x223C) {
    return CC_FAMILY_FULL_KANA | CC_CHOON;
  } else if (code == 0xFF70) {
    return CC_HANKAKU_KANA | CC_CHOON;
  } else if (0xFF66 <= code && code <= 0xFF9F) {
    return CC_HANKAKU_KANA;
  } else if (code == 0x00B7 || code == 0x30fb) {
    return CC_MIDDLE_DOT;
  } else if (code == 0x002C || code == 0xff0c) {
    return CC_COMMA;
  } else if (code == 0x002F || code == 0xff0f) {
    return CC_SLASH;
  } else if (code == 0x003A || code == 0xff1a) {
    return CC_COLON;
  } else if (code == 0xff0e) {
    return CC_PERIOD;
  } else if ((code > 0x2f && code < 0x3a) || (code > 0xff0f && code < 0xff1a)) {
    return CC_FIGURE;
  } else if (code == 0x25cb || code == 0x3007 || code == 0x96f6 || code == 0x4e00 || code == 0x4e8c ||
             code == 0x4e09 || code == 0x56db || code == 0x4e94 || code == 0x516d || code == 0x4e03 ||
             code == 0x516b || code == 0x4e5d) {
    return CC_KANJI_FIGURE | CC_KANJI;
  } else if (code == 0x5341 || code == 0x767e || code == 0x5343 || code == 0x4e07 || code == 0x5104 ||
             code == 0x5146 || code == 0x6570 || code == 0x4F55 || code == 0x5E7E) {
    if (code == 0x6570 || code == 0x4F55 || code == 0x5E7E) return CC_FIGURE_EXCEPTION | CC_KANJI;
    return CC_KANJI_FIGURE | CC_FIGURE_DIGIT;
  } else if ((code >= 0x40 && code <= 0x5b) || (code >= 0x60 && code <= 0x7b) ||
             (code >= 0xbf && code <= 0x0100) || (code >= 0xff20 && code <= 0xff3b) ||
             (code >= 0xff40 && code <= 0xff5b) || (code >= 0x370 && code <= 0x3ff) ||
             (code >= 0x400 && code <= 0x4ff)) {
    return CC_ALPH;
  } else if ((code > 0x4dff && code < 0xa000) || code == 0x3005 || code == 0x3007) {
    return CC_KANJI;
  } else if (is_bracket(code)) {
    return CC_BRACKET;
  }
  return CC_SYMBOL;
}

// UTF-8 decode of one codepoint (reference src/util/characters.h:86-131).
// returns byte length, 0 on an invalid sequence.
__device__ __forceinline__ int utf8_decode(const u8* __restrict__ p, int avail, u32& cp) {
  u32 b0 = p[0];
  if (b0 > 0xef) {
    if (avail < 4 || (b0 & ~0x7u) != 0xf0) return 0;
    u32 b1 = p[1], b2 = p[2], b3 = p[3];
    if ((b1 & 0xc0) != 0x80 || (b2 & 0xc0) != 0x80 || (b3 & 0xc0) != 0x80) return 0;
    cp = ((b0 & 7u) << 18) | ((b1 & 0x3fu) << 12) | ((b2 & 0x3fu) << 6) | (b3 & 0x3fu);
    return 4;
  } else if (b0 > 0xdf) {
    if (avail < 3 || (b0 & ~0xfu) != 0xe0) return 0;
    u32 b1 = p[1], b2 = p[2];
    if ((b1 & 0xc0) != 0x80 || (b2 & 0xc0) != 0x80) return 0;
    cp = ((b0 & 0xfu) << 12) | ((b1 & 0x3fu) << 6) | (b2 & 0x3fu);
    return 3;
  } else if (b0 > 0x7f) {
    if (avail < 2 || (b0 & ~0x1fu) != 0xc0) return 0;
    u32 b1 = p[1];
    if ((b1 & 0xc0) != 0x80) return 0;
    cp = ((b0 & 0x1fu) << 6) | (b1 & 0x3fu);
    return 2;
  }
  if (avail < 1) return 0;
  cp = b0 & 0x7fu;
  return 1;
}

// ----------------------------------------------------------------------------
// UNK content hash (reference src/core/analysis/unk_nodes_creator.cc:170-177 +
// src/util/murmur_hash.h:116-160).  NB: the reference copies only
// sizeof(size_t)=8 bytes of every 16-byte block (murmur_hash.h:131), the
// second half of the block is a value-initialised zero -- reproduced here.
// ----------------------------------------------------------------------------
__device__ __forceinline__ u64 rotl64(u64 x, int r) { return (x << r) | (x >> (64 - r)); }
__device__ __forceinline__ u64 fmix64(u64 k) {
  k ^= k >> 33;
  k *= 0xff51afd7ed558ccdULL;
  k ^= k >> 33;
  k *= 0xc4ceb9fe1a85ec53ULL;
  k ^= k >> 33;
  return k;
}

__device__ __forceinline__ i32 unk_string_hash(const u8* __restrict__ data, u32 len) {
  const u64 C1 = 0x87c37b91114253d5ULL, C2 = 0x4cf5ad432745937fULL;
  u64 v1 = 0xa76210bfULL, v2 = 0xa76210bfULL;
  u32 nblocks = len / 16;
  for (u32 i = 0; i < nblocks; ++i) {
    u64 b1 = 0;
    for (int k = 0; k < 8; ++k) b1 |= (u64)data[i * 16 + k] << (8 * k);
    u64 b2 = 0;
    b1 *= C1; b1 = rotl64(b1, 31); b1 *= C2;
    b2 *= C2; b2 = rotl64(b2, 33); b2 *= C1;
    v1 ^= b1; v1 = rotl64(v1, 27); v1 += v2; v1 = v1 * 5 + 0x52dce729;
    v2 ^= b2; v2 = rotl64(v2, 31); v2 += v1; v2 = v2 * 5 + 0x38495ab5;
  }
  const u8* tail = data + nblocks * 16;
  u32 rem = len & 0xf;
  u64 t1 = 0, t2 = 0;
  for (u32 k = 0; k < rem; ++k) {
    if (k < 8) t1 ^= (u64)tail[k] << (8 * k);
    else t2 ^= (u64)tail[k] << (8 * (k - 8));
  }
  t1 *= C1; t1 = rotl64(t1, 31); t1 *= C2;
  t2 *= C2; t2 = rotl64(t2, 33); t2 *= C1;
  v1 ^= t1; v2 ^= t2;
  v1 ^= len; v2 ^= len;
  v1 += v2; v2 += v1;
  v1 = fmix64(v1); v2 = fmix64(v2);
  v1 += v2;
  u32 trimmed = (u32)v1;
  return (i32)(trimmed | 0x80000000u);
}

// sortable float <-> u32 (reference BeamCandidate::pack/score,
// src/core/analysis/score_processor.h:87-114)
__device__ __forceinline__ u32 f32_sortable(float f) {
  u32 v;
  __builtin_memcpy(&v, &f, 4);
  return (v & 0x80000000u) ? ~v : (v ^ 0x80000000u);
}
__device__ __forceinline__ float sortable_f32(u32 v) {
  v = (v & 0x80000000u) == 0 ? ~v : (v ^ 0x80000000u);
  float f;
  __builtin_memcpy(&f, &v, 4);
  return f;
}

}  // namespace jpp

#endif  // JPP_DEVICE_H
