// TEST ONLY: csrc/jpp_fmtg.h (the device's exact "%g" of a float) against the C library over the float range.
//   g++ -std=c++17 -O2 -DJPP_EMU -Itests/emu -Ijumanpp_amd/csrc tests/host/fmtg_test.cc -o build/fmtg_test && build/fmtg_test [N]
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "jpp_fmtg.h"

static unsigned long long checked = 0, bad = 0;

static void check(unsigned bits) {
  float v;
  std::memcpy(&v, &bits, 4);
  char want[64];
  const int wn = std::snprintf(want, sizeof(want), "%g", (double)v);
  unsigned char got[64];
  const unsigned gn = jpp::g_format(v, got);
  const unsigned cn = jpp::g_format(v, (unsigned char*)nullptr);
  ++checked;
  if ((int)gn != wn || cn != gn || std::memcmp(want, got, gn) != 0) {
    if (bad < 20) std::fprintf(stderr, "MISMATCH bits=%08x want=%s got=%.*s (count %u)\n", bits, want, (int)gn, got, cn);
    ++bad;
  }
}

int main(int argc, char** argv) {
  const unsigned long long n = argc > 1 ? std::strtoull(argv[1], nullptr, 10) : 3000000ull;
  // every exponent with structured mantissas, both signs
  for (unsigned ex = 0; ex < 256; ++ex)
    for (unsigned sign = 0; sign < 2; ++sign) {
      const unsigned base = (sign << 31) | (ex << 23);
      for (unsigned k = 0; k < 64; ++k) {
        check(base | k);
        check(base | (0x7fffffu - k));
        check(base | (0x400000u + k));
        check(base | (0x400000u - k - 1));
      }
      for (unsigned b = 0; b < 23; ++b) check(base | (1u << b));
    }
  // neighbourhoods of the powers of ten and of d.ddddd5 ties
  for (int p = -45; p <= 38; ++p) {
    char buf[32];
    for (int lead = 1; lead <= 9; ++lead) {
      std::snprintf(buf, sizeof(buf), "%de%d", lead, p);
      float f = std::strtof(buf, nullptr);
      unsigned bits;
      std::memcpy(&bits, &f, 4);
      for (int d = -40; d <= 40; ++d) check(bits + (unsigned)d);
    }
    for (int q = 0; q < 400; ++q) {
      std::snprintf(buf, sizeof(buf), "%d.%05d5e%d", 1 + q % 9, (q * 7919) % 100000, p);
      float f = std::strtof(buf, nullptr);
      unsigned bits;
      std::memcpy(&bits, &f, 4);
      for (int d = -3; d <= 3; ++d) check(bits + (unsigned)d);
    }
  }
  // small integers and halves (exact ties at six digits need more than 24 bits; 2^k * odd patterns cover the exact cases)
  for (unsigned i = 0; i < 2000000; i += 7) {
    float f = (float)i * 0.5f;
    unsigned bits;
    std::memcpy(&bits, &f, 4);
    check(bits);
    f = (float)i * 0.015625f;
    std::memcpy(&bits, &f, 4);
    check(bits);
  }
  // random bit patterns
  unsigned long long x = 0x9E3779B97F4A7C15ull;
  for (unsigned long long i = 0; i < n; ++i) {
    x ^= x << 13;
    x ^= x >> 7;
    x ^= x << 17;
    check((unsigned)(x >> 16));
  }
  std::printf("checked %llu floats, %llu mismatches\n", checked, bad);
  return bad == 0 ? 0 : 1;
}
