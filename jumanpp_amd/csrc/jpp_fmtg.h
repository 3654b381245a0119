// printf("%g") of a float on the device, exactly.
//
// The lattice output format prints three scores per line through the reference's printer (util::io::Printer <<
// float -> fmt's BasicWriter << double, i.e. the C library's "%g" of the float widened to double;
// src/jumandic/shared/lattice_format.cc:110-116,218-232).  "%g" is a correctly rounded conversion: six significant
// decimal digits of the EXACT binary value, ties to even, then the choice between fixed and exponent notation and the
// removal of trailing zeros.  A float is m * 2^e with m < 2^24: its exact decimal expansion is finite, and six digits of
// it plus "is the rest above, at or below one half" need nothing but integer arithmetic --
//   * |v| >= 10^6: the integer part (up to 128 bits, four 32-bit limbs) is divided by ten until six digits are left;
//     the last digit removed and whether anything non-zero was removed before it decide the rounding;
//   * |v| < 10^6: the integer part is small, the fraction is a 160-bit fixed-point number (binary point above limb 4)
//     that is multiplied by ten; every carry out of the top limb is the next decimal digit.
// No floating-point operation takes part, so the result does not depend on rounding modes, contraction or the
// precision of a device pow / log.  tests/host/fmtg_test.cc compares it with snprintf over the float range.
#ifndef JPP_FMTG_H
#define JPP_FMTG_H

#include "jpp_rt.h"

namespace jpp {

struct GDigits {
  u32 digits;   // 100000 .. 999999 (six significant digits), 0 for a zero
  i32 exp10;    // decimal exponent X of the first digit: |v| = d.ddddd * 10^X
  u32 neg;      // sign bit
  u32 special;  // 0 finite, 1 inf, 2 nan
};

// six significant digits of |v|, correctly rounded (ties to even)
__device__ __forceinline__ GDigits g_digits(float v) {
  u32 bits;
  __builtin_memcpy(&bits, &v, 4);
  GDigits r;
  r.neg = bits >> 31;
  r.digits = 0;
  r.exp10 = 0;
  r.special = 0;
  const u32 ex = (bits >> 23) & 0xffu;
  u32 m = bits & 0x7fffffu;
  if (ex == 0xffu) {
    r.special = m ? 2u : 1u;
    return r;
  }
  i32 e;   // |v| = m * 2^e
  if (ex == 0) {
    if (m == 0) return r;
    e = -149;
  } else {
    m |= 0x800000u;
    e = (i32)ex - 150;
  }
  u32 D = 0;      // digits so far
  i32 nd = 0;     // how many
  i32 X = 0;
  bool up = false, tie = false;
  // the integer part: m << e for e >= 0, m >> -e otherwise (0 when -e >= 24)
  u32 I[4] = {0, 0, 0, 0};
  bool frac_nonzero = false;
  if (e >= 0) {
    const u32 w = (u32)e >> 5, sh = (u32)e & 31u;
    const u64 lo = (u64)m << sh;   // < 2^55
#pragma unroll
    for (u32 k = 0; k < 4; ++k) {
      if (k == w) I[k] = (u32)lo;
      if (k == w + 1) I[k] = (u32)(lo >> 32);
    }
  } else if (e > -24) {
    I[0] = m >> (u32)(-e);
    frac_nonzero = (m & ((1u << (u32)(-e)) - 1u)) != 0;
  } else {
    frac_nonzero = true;
  }
  const bool big = (I[3] | I[2] | I[1]) != 0 || I[0] >= 1000000u;
  if (big) {
    // divide by ten until six digits are left
    u32 last = 0;
    bool sticky = frac_nonzero;
    i32 removed = 0;
    while ((I[3] | I[2] | I[1]) != 0 || I[0] >= 1000000u) {
      sticky = sticky || last != 0;
      u64 rem = 0;
#pragma unroll
      for (int k = 3; k >= 0; --k) {
        const u64 cur = (rem << 32) | I[k];
        const u64 q = cur / 10u;
        I[k] = (u32)q;
        rem = cur - q * 10u;
      }
      last = (u32)rem;
      ++removed;
    }
    D = I[0];
    X = 5 + removed;
    up = last > 5 || (last == 5 && sticky);
    tie = last == 5 && !sticky;
  } else {
    // digits of the integer part
    const u32 ip = I[0];
    if (ip != 0) {
      D = ip;
      nd = ip >= 100000u ? 6 : ip >= 10000u ? 5 : ip >= 1000u ? 4 : ip >= 100u ? 3 : ip >= 10u ? 2 : 1;
      X = nd - 1;
    }
    // the fraction as 160-bit fixed point: F[4] is the most significant limb, the binary point lies above it
    u32 F[5] = {0, 0, 0, 0, 0};
    if (e < 0) {
      const u32 s = (u32)(-e);                                          // fraction = (m mod 2^s) / 2^s, s <= 149
      const u32 fm = s >= 24 ? m : (m & ((1u << s) - 1u));              // < 2^min(s, 24)
      const u32 p = 160u - s;                                           // bit position of the fraction's unit
      const u32 w = p >> 5, sh = p & 31u;
      const u64 lo = (u64)fm << sh;
#pragma unroll
      for (u32 k = 0; k < 5; ++k) {
        if (k == w) F[k] = (u32)lo;
        if (k == w + 1) F[k] = (u32)(lo >> 32);
      }
    }
    i32 lead = 0;   // zeros between the point and the first digit (values below 1)
    while (nd < 6) {
      u64 carry = 0;
#pragma unroll
      for (int k = 0; k < 5; ++k) {
        const u64 t = (u64)F[k] * 10u + carry;
        F[k] = (u32)t;
        carry = t >> 32;
      }
      const u32 d = (u32)carry;
      if (nd == 0 && d == 0) {
        ++lead;
        continue;   // (terminates: the fraction is non-zero here, since v != 0 and its integer part is 0)
      }
      if (nd == 0) X = -1 - lead;
      D = D * 10u + d;
      ++nd;
    }
    const bool half = (F[4] >> 31) != 0;
    const bool rest = ((F[4] << 1) | F[3] | F[2] | F[1] | F[0]) != 0;
    up = half && rest;
    tie = half && !rest;
  }
  if (up || (tie && (D & 1u))) {
    ++D;
    if (D == 1000000u) {
      D = 100000u;
      ++X;
    }
  }
  r.digits = D;
  r.exp10 = X;
  return r;
}

// the text of "%g" for the digits: appends to out (may be null: count only), returns the number of bytes
template <typename P>
__device__ __forceinline__ u32 g_emit(const GDigits& g, P out) {
  u32 n = 0;
  auto put = [&](char c) {
    if (out) out[n] = (u8)c;
    ++n;
  };
  if (g.neg) put('-');
  if (g.special == 1) {
    put('i'); put('n'); put('f');
    return n;
  }
  if (g.special == 2) {
    put('n'); put('a'); put('n');
    return n;
  }
  if (g.digits == 0) {
    put('0');
    return n;
  }
  // the six digits, most significant first, and how many remain without the trailing zeros
  u32 dg[6];
  u32 t = g.digits;
#pragma unroll
  for (int k = 5; k >= 0; --k) {
    dg[k] = t % 10u;
    t /= 10u;
  }
  int nd = 6;
#pragma unroll
  for (int k = 5; k >= 1; --k)
    if (nd == k + 1 && dg[k] == 0) nd = k;
  const i32 X = g.exp10;
  auto digit = [&](int k) -> char {   // (a select chain instead of a dynamically indexed register array)
    u32 d = dg[0];
#pragma unroll
    for (int j = 1; j < 6; ++j) d = k == j ? dg[j] : d;
    return (char)('0' + d);
  };
  if (X < -4 || X >= 6) {
    put(digit(0));
    if (nd > 1) {
      put('.');
      for (int k = 1; k < nd; ++k) put(digit(k));
    }
    put('e');
    put(X < 0 ? '-' : '+');
    const u32 ax = (u32)(X < 0 ? -X : X);   // < 100 for a float
    put((char)('0' + ax / 10u));
    put((char)('0' + ax % 10u));
    return n;
  }
  if (X >= 0) {
    for (int k = 0; k <= X; ++k) put(k < nd ? digit(k) : '0');
    if (nd > X + 1) {
      put('.');
      for (int k = X + 1; k < nd; ++k) put(digit(k));
    }
    return n;
  }
  put('0');
  put('.');
  for (int k = 0; k < -X - 1; ++k) put('0');
  for (int k = 0; k < nd; ++k) put(digit(k));
  return n;
}

template <typename P>
__device__ __forceinline__ u32 g_format(float v, P out) {
  return g_emit(g_digits(v), out);
}

// decimal text of an unsigned integer
template <typename P>
__device__ __forceinline__ u32 u32_format(u32 v, P out) {
  const u32 n = v >= 1000000000u ? 10 : v >= 100000000u ? 9 : v >= 10000000u ? 8 : v >= 1000000u ? 7 : v >= 100000u ? 6 :
                v >= 10000u ? 5 : v >= 1000u ? 4 : v >= 100u ? 3 : v >= 10u ? 2 : 1;
  if (out) {
    u32 t = v;
    for (int k = (int)n - 1; k >= 0; --k) {
      out[k] = (u8)('0' + t % 10u);
      t /= 10u;
    }
  }
  return n;
}

}  // namespace jpp

#endif  // JPP_FMTG_H
