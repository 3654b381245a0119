// TEST INFRASTRUCTURE ONLY (oracle).  NOT part of the product.
//
// Independent cross-check of the oracle build's RNN arithmetic: the reference's
// own re-implementation (src/rnn/mikolov_rnn.cc, whose six Eigen expressions run here through
// oracle/shim/eigen3/Eigen/Core) against the in-tree legacy faster-rnnlm evaluator
// (src/rnn/legacy/rnnlmlib_static.cpp, no Eigen, its own loops, exp() and hashing) on a synthetic
// version-6 NCE model (tools/gen_rnn.py).  The reference's tests make the same comparison at 1e-3 on
// a model that is not in the repository (rnn/mikolov_rnn_test.cc:295-397, "rnn/testlm"); here it is
// 1e-4 on hidden states and on log10 scores, over seeded random word chains.  Both sides are the
// reference's code, linked from oracle/_ref; this file only drives them.
//
//   legacy_check <rnn-model-prefix> [chains=200] [length=8] [seed=1]
//
// prints one JSON line; exit code 0 iff every comparison is within tolerance.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <string>
#include <vector>

#include "rnn/legacy/rnnlmlib_static.h"
#include "rnn/mikolov_rnn.h"
#include "util/logging.hpp"
#include "util/memory.hpp"

using namespace jumanpp;
using namespace jumanpp::rnn::mikolov;

namespace {

struct State {
  std::vector<i32> prev;   // most recent word first, at most maxentOrder - 1 entries
  std::vector<float> ctx;  // hidden layer
};

struct NewSide {
  MikolovModelReader rdr;
  MikolovRnn rnn;
  util::memory::Manager mgr{4 * 1024 * 1024};
  std::shared_ptr<util::memory::PoolAlloc> alloc;
  NewSide() : alloc{mgr.core()} {}

  bool open(const std::string& path) {
    if (!rdr.open(path) || !rdr.parse()) return false;
    return (bool)rnn.init(rdr.header(), rdr.rnnMatrix(), rdr.maxentWeights());
  }

  // one step through the batched API the global-beam scorer uses (rnn_scorer_gbeam.cc:142-233):
  // new context from (old context, embedding of the previous word), then the score of `word`
  State step(const State& s, i32 word, float* score) {
    alloc->reset();
    const auto& h = rdr.header();
    const u32 E = h.layerSize;
    auto ctxIds = alloc->allocate2d<i32>(1, s.prev.size());
    for (size_t i = 0; i < s.prev.size(); ++i) ctxIds.row(0).at(i) = s.prev[i];
    auto oldCtx = alloc->allocate2d<float>(1, E, 64);
    auto leftEmb = alloc->allocate2d<float>(1, E, 64);
    auto nceEmb = alloc->allocate2d<float>(1, E, 64);
    auto newCtx = alloc->allocate2d<float>(1, E, 64);
    for (u32 i = 0; i < E; ++i) {
      oldCtx.row(0).at(i) = s.ctx[i];
      leftEmb.row(0).at(i) = rdr.embeddings().at((size_t)s.prev[0] * E + i);
      nceEmb.row(0).at(i) = rdr.nceEmbeddings().at((size_t)word * E + i);
    }
    auto scores = alloc->allocateBuf<float>(1, 64);
    scores[0] = 0;
    std::vector<i32> words{word};
    ParallelContextData pcd{oldCtx, leftEmb, newCtx};
    rnn.computeNewParCtx(&pcd);
    ParallelStepData psd{ctxIds, words, newCtx, nceEmb, scores};
    rnn.applyParallel(&psd);
    *score = scores[0];
    State n;
    n.ctx.assign(newCtx.row(0).begin(), newCtx.row(0).end());
    n.prev.push_back(word);
    for (size_t i = 0; i + 1 < std::min<size_t>(s.prev.size() + 1, h.maxentOrder - 1); ++i) n.prev.push_back(s.prev[i]);
    return n;
  }
};

}  // namespace

int main(int argc, char** argv) {
  if (argc < 2) {
    std::fprintf(stderr, "usage: legacy_check <rnn-model-prefix> [chains] [length] [seed]\n");
    return 2;
  }
  const std::string path = argv[1];
  const int chains = argc > 2 ? std::atoi(argv[2]) : 200;
  const int length = argc > 3 ? std::atoi(argv[3]) : 8;
  const unsigned seed = argc > 4 ? (unsigned)std::atoi(argv[4]) : 1u;

  util::logging::CurrentLogLevel = util::logging::Level::Warning;  // the legacy evaluator logs every step at debug level
  NewSide nw;
  if (!nw.open(path)) {
    std::fprintf(stderr, "cannot read %s with MikolovModelReader\n", path.c_str());
    return 2;
  }
  RNNLM_legacy::CRnnLM_stat lm;
  lm.setDebugMode(0);
  lm.setRnnLMFile(path.c_str());
  RNNLM_legacy::context lc0;
  lm.get_initial_context_FR(&lc0);  // reads the network (restoreNet_FR) and makes the sentence-start context

  const auto& h = nw.rdr.header();
  const u32 E = h.layerSize;
  const auto& words = nw.rdr.words();
  std::mt19937 rng(seed);
  double maxScore = 0, maxCtx = 0;
  long steps = 0, bad = 0;
  for (int c = 0; c < chains; ++c) {
    State s;
    s.prev = {0};
    s.ctx.assign(E, 0.f);
    RNNLM_legacy::context lc = lc0;
    for (int t = 0; t < length; ++t) {
      const i32 w = 1 + (i32)(rng() % (words.size() - 1));
      float sc = 0;
      State n = nw.step(s, w, &sc);
      RNNLM_legacy::context ln;
      const std::string word = words[w].str();
      const float lsc = lm.test_word_selfnm(&lc, &ln, word, word.size());
      const double mine = std::log10(std::exp((double)sc));  // the reference test's normalizedScore
      const double ds = std::fabs(mine - (double)lsc);
      double dc = 0;
      for (u32 i = 0; i < E; ++i) dc = std::max(dc, (double)std::fabs(n.ctx[i] - ln.l1_neuron[i]));
      maxScore = std::max(maxScore, ds);
      maxCtx = std::max(maxCtx, dc);
      if (!(ds <= 1e-4 * std::max(1.0, std::fabs((double)lsc))) || !(dc <= 1e-4)) {
        if (bad < 5)
          std::fprintf(stderr, "chain %d step %d word %d: score %.7g vs legacy %.7g, max ctx diff %.3g\n", c, t, w, mine,
                       (double)lsc, dc);
        ++bad;
      }
      ++steps;
      s = std::move(n);
      lc = ln;
    }
  }
  std::printf("{\"steps\": %ld, \"mismatches\": %ld, \"max_abs_score_diff_log10\": %.3g, \"max_abs_context_diff\": %.3g, "
              "\"layer_size\": %u, \"vocab\": %zu, \"maxent_order\": %u}\n",
              steps, bad, maxScore, maxCtx, E, words.size(), (unsigned)h.maxentOrder);
  return bad == 0 ? 0 : 1;
}
