// TEST ONLY: DoubleArrayBuilder (jumanpp_amd/host/rnn_external.cc) on random key sets: every key must be
// found with its value, near-misses must not, and the arrays must stay inside their bounds for any byte.
#include <cstdio>
#include <random>
#include <set>
#include <string>

#include "rnn_external.h"

using namespace jumanpp_amd;

static uint32_t offsetOf(uint32_t u) { return (u >> 10) << ((u & (1u << 9)) >> 6); }

int main() {
  std::mt19937 rng(12345);
  long failures = 0;
  for (int round = 0; round < 6; ++round) {
    const int nkeys = round == 0 ? 1 : round == 1 ? 300 : 20000 * round;
    std::set<std::string> keys;
    while ((int)keys.size() < nkeys) {
      std::string k;
      if (!keys.empty() && rng() % 4 == 0) {  // extend an existing key: prefixes of each other
        auto it = keys.begin();
        std::advance(it, rng() % std::min<size_t>(keys.size(), 50));
        k = *it;
      }
      int len = 1 + rng() % 10;
      for (int i = 0; i < len; ++i) k.push_back((char)(1 + rng() % (round == 2 ? 3 : 255)));  // round 2: tiny alphabet, deep sharing
      keys.insert(k);
    }
    DoubleArrayBuilder b;
    int v = 0;
    std::vector<std::pair<std::string, int>> kv;
    for (auto& k : keys) {
      kv.emplace_back(k, v);
      b.add(k, v++);
    }
    std::vector<uint32_t> units;
    Status s = b.build(&units);
    if (!s.isOk()) { printf("round %d: build failed\n", round); return 1; }
    for (auto& e : kv) {
      int32_t got = -1;
      if (!DoubleArrayBuilder::find(units, e.first, &got) || got != e.second) ++failures;
      std::string miss = e.first + "\x7f";
      if (!keys.count(miss) && DoubleArrayBuilder::find(units, miss, &got)) ++failures;
      miss = e.first.substr(0, e.first.size() - 1);
      if (!miss.empty() && !keys.count(miss) && DoubleArrayBuilder::find(units, miss, &got)) ++failures;
    }
    // the device does not bound-check: from every reachable unit, every byte must land inside the array
    for (auto& e : kv) {
      uint32_t id = 0, unit = units[0];
      for (unsigned char c : e.first) {
        for (int probe = 0; probe < 256; probe += 51)
          if ((id ^ offsetOf(unit) ^ (uint32_t)probe) >= units.size()) ++failures;
        id ^= offsetOf(unit) ^ c;
        unit = units[id];
      }
    }
    printf("round %d: %d keys, %zu units, failures so far %ld\n", round, nkeys, units.size(), failures);
  }
  return failures == 0 ? 0 : 1;
}
