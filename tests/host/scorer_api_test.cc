// TEST (CPU emulator / MI355X): the C++14 host mirror of ScorerDef{feature, others, scoreWeights} and of the
// per-connection ScorePlugin (jumanpp_amd/host/gpu_analyzer.h), driven the way a reference user drives
// core::analysis::Analyzer: a ScorerFactory in ScorerDef::others and a ScorePlugin passed to analyzeBatch.
// The scorer and the plugin are the TestScorer / TestPlugin of oracle/ref_dump.cc `top1x`; this program prints the
// packed top-1 analyses in the same binary layout so that tests/test_scorers.py can compare the two files.
//   scorer_api_test <model> <out.bin> <scorer-weight|none> <plugin 0|1> [--rnn] < corpus
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <string>
#include <vector>

#include "gpu_analyzer.h"
#include "model_image.h"

using namespace jumanpp_amd;

namespace {
struct TestScorer : public ScoreComputer {
  Status scoreLattice(const jppgpu_result_view& v, uint32_t idx, float* cells) override {
    const int G = v.global_beam, S = v.num_scorers;
    for (uint32_t s = 0; s < v.n_sentences; ++s) {
      if (v.status[s] != JPPGPU_SENT_OK || v.n_codepoints[s] == 0) continue;
      const uint64_t nb = v.node_base[s], bb = v.bnd_base[s];
      for (uint32_t b = 2; b < v.n_codepoints[s] + 3; ++b) {
        const uint32_t R = v.bnd_count[bb + b], first = v.bnd_first[bb + b];
        if (R == 0) continue;
        const uint32_t ngb = v.gbeam_count[bb + b], ef = v.end_first[bb + b];
        for (uint32_t i = 0; i < ngb; ++i) {
          // gbeam entry: {u16 left, u16 beam, float score}
          const uint16_t left = *reinterpret_cast<const uint16_t*>(reinterpret_cast<const char*>(v.gbeam) + ((bb + b) * G + i) * 8);
          const jppgpu_node& ln = v.nodes[nb + v.end_nodes[nb + ef + left]];
          float base = 0.f;
          if (ln.entry_ptr < 0) base -= 0.25f;
          base -= 0.0625f * (float)(ln.end - ln.start);
          for (uint32_t r = 0; r < R; ++r) {
            float x = base;
            if (v.nodes[nb + first + r].entry_ptr < 0) x += 0.125f;
            cells[((nb + first + r) * G + i) * S + idx] = x;
          }
        }
      }
    }
    return Status::Ok();
  }
};
struct TestScorerFactory : public ScorerFactory {
  bool loaded = false;
  Status load(const ModelImage& model) override {   // ScorerFactory::load(ModelInfo): what the scorer reads from the model
    loaded = model.numFeatures() > 0;
    return Status::Ok();
  }
  Status makeInstance(std::unique_ptr<ScoreComputer>* result) override {
    if (!loaded) return Status::InvalidState("TestScorerFactory: load() was not called");
    result->reset(new TestScorer());
    return Status::Ok();
  }
};
struct TestPlugin : public ScorePlugin {
  bool perConnection() const override { return true; }
  void connectionPenalties(const jppgpu_lattice_pairs& v, const std::vector<uint32_t>&, float* pen) override {
    for (uint32_t s = 0; s < v.n_sentences; ++s) {
      if (v.status[s] != JPPGPU_SENT_OK) continue;
      const uint64_t nb = v.node_base[s], bb = v.bnd_base[s];
      for (uint32_t b = 2; b < v.n_codepoints[s] + 3; ++b) {
        const uint32_t R = v.bnd_count[bb + b], first = v.bnd_first[bb + b], L = v.end_count[bb + b], ef = v.end_first[bb + b];
        for (uint32_t l = 0; l < L; ++l) {
          const jppgpu_node& ln = v.nodes[nb + v.end_nodes[nb + ef + l]];
          for (uint32_t r = 0; r < R; ++r) {
            const jppgpu_node& rn = v.nodes[nb + first + r];
            float a = 0.f;
            if (ln.entry_ptr < 0) a += 1.0f;
            if (rn.end - rn.start > 2 && (ln.start & 1) != 0) a += 0.5f;
            pen[v.pair_base[bb + b] + (uint64_t)l * R + r] = a;
          }
        }
      }
    }
  }
};
template <typename T>
void put(std::string& o, T v) { o.append(reinterpret_cast<const char*>(&v), sizeof(T)); }
}  // namespace

int main(int argc, char** argv) {
  if (argc < 5) return 2;
  ModelImage model;
  Status s = model.loadModel(argv[1]);
  if (!s) { std::cerr << s << "\n"; return 1; }
  const bool useRnn = argc > 5 && std::strcmp(argv[5], "--rnn") == 0;
  ScorerDef def;
  ModelRnnScorerFactory rnn;
  TestScorerFactory factory;
  RnnScoreWeights w = model.savedScoreWeights();
  def.scoreWeights.push_back(useRnn ? w.perceptron : 1.0f);
  if (useRnn) {
    s = rnn.load(model);
    if (!s) { std::cerr << s << "\n"; return 1; }
    def.others.push_back(&rnn);   // the reference's RnnHolder factory, first in `others`
    def.scoreWeights.push_back(w.rnn);
  } else if (model.hasRnn() == false && rnn.load(model)) {
    std::cerr << "ModelRnnScorerFactory::load accepted a model without an RNN part\n";
    return 1;
  }
  if (std::strcmp(argv[3], "none") != 0) {
    s = factory.load(model);
    if (!s) { std::cerr << s << "\n"; return 1; }
    def.others.push_back(&factory);
    def.scoreWeights.push_back((float)std::atof(argv[3]));
  }
  AnalyzerConfig ac;
  ac.globalBeamSize = 6;
  ac.rightGbeamCheck = 1;
  ac.rightGbeamSize = 5;
  ScoringConfig sc;
  sc.beamSize = 5;
  sc.numScorers = def.numScorers();
  GpuAnalyzer an;
  s = an.initialize(&model, ac, sc, &def);
  if (!s) { std::cerr << s << "\n"; return 1; }
  std::vector<std::string> lines;
  std::string line;
  while (std::getline(std::cin, line)) lines.push_back(line);
  std::vector<StringPiece> inputs(lines.begin(), lines.end());
  TestPlugin plugin;
  s = std::atoi(argv[4]) ? an.analyzeBatch(inputs, &plugin, false) : an.analyzeBatch(inputs, false);
  if (!s) { std::cerr << s << "\n"; return 1; }
  std::string out;
  put<uint32_t>(out, 0x31504f54u);
  put<uint32_t>(out, (uint32_t)lines.size());
  for (size_t i = 0; i < lines.size(); ++i) {
    if (!an.sentenceStatus(i)) {
      put<uint32_t>(out, 1);
      put<uint32_t>(out, 0);
      continue;
    }
    SentenceResult r = an.sentence(i);
    put<uint32_t>(out, 0);
    put<uint32_t>(out, r.pathLen > 0 ? r.pathLen - 1 : 0);
    for (uint32_t k = r.pathLen; k-- > 1;) {   // pathNodes is EOS first
      const jppgpu_node& nd = r.nodes[r.pathNodes[k]];
      put<int32_t>(out, nd.entry_ptr);
      put<uint16_t>(out, nd.start);
      put<uint16_t>(out, nd.end);
    }
  }
  FILE* f = std::fopen(argv[2], "wb");
  if (!f) return 1;
  std::fwrite(out.data(), 1, out.size(), f);
  std::fclose(f);
  return 0;
}
