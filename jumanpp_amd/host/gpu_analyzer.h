// GpuAnalyzer: the C++14 host-side mirror of core::analysis::Analyzer
// (src/core/analysis/analyzer.h:37-57) above the C ABI of libjppgpu.so.
//
//   reference                                   here
//   Analyzer::initialize(core, cfg, sconf, def)  GpuAnalyzer::initialize(model, cfg, sconf, def)
//   Analyzer::analyze(StringPiece)               GpuAnalyzer::analyze(StringPiece)        (batch of one)
//                                                GpuAnalyzer::analyzeBatch(inputs)        (the fast path)
//   Analyzer::output() / impl()->lattice()       GpuAnalyzer::sentence(i) -> SentenceResult
//
// Same argument meaning (AnalyzerConfig / ScoringConfig / ScorerDef), same
// validation and the same error kinds: input longer than maxInputBytes and
// invalid UTF-8 are InvalidParameter, an unconnectable lattice is InvalidState
// (src/core/analysis/analyzer_impl.cc:19-25,131-136, util/characters.cc:267-269).
// Results stay valid until the next analyze/analyzeBatch on the same object,
// like the reference's pool-allocated lattice (analyzer_impl.h:48-55).
// There is no CPU path: initialize fails with NoDevice without an MI355X.
#ifndef JUMANPP_AMD_HOST_GPU_ANALYZER_H
#define JUMANPP_AMD_HOST_GPU_ANALYZER_H

#include <cstdint>
#include <memory>
#include <string>
#include <vector>

#include "jpp_status.h"
#include "jppgpu.h"
#include "model_image.h"
#include "partial_example.h"

namespace jumanpp_amd {

// core::analysis::AnalyzerConfig (analyzer.h:14-25).  With autoBeamStep > 0 every sentence gets
// beam = global beam = min(autoBeamBase + codepoints / autoBeamStep, autoBeamMax)
// (AnalyzerImpl::autoBeamSizes, analyzer_impl.cc:350-361): the batch is analysed in groups of equal beam.
struct AnalyzerConfig {
  size_t pageSize = 4 * 1024 * 1024;  // unused: device workspaces are sized per batch
  size_t maxInputBytes = 4 * 1024;
  int32_t globalBeamSize = 0;
  int32_t otherScorersTopN = 0;
  int32_t rightGbeamCheck = 0;
  int32_t rightGbeamSize = 0;
  bool storeAllPatterns = false;
  int32_t autoBeamStep = 0;
  int32_t autoBeamBase = 0;
  int32_t autoBeamMax = 0;
};

// core::ScoringConfig (src/core/core_types.h)
struct ScoringConfig {
  int32_t beamSize = 1;
  int32_t numScorers = 1;
};

// core::analysis::ScoreComputer (score_api.h:54-59): a scorer beyond the perceptron.  The reference hands it the
// pointer-linked Lattice of ONE sentence; here it sees the index-form lattice of a whole batch (the full result view:
// nodes, ends lists, global beams, beams, cells) and writes its slot of the score cells,
//     cells[((node_base[i] + node) * global_beam + gbeamIndex) * num_scorers + scorerIdx],
// for the connections it scores -- what the reference's scorer does through
// scores->nodeScores(right).beamLeft(beam, left).at(scorerIdx).
class ScoreComputer {
 public:
  virtual ~ScoreComputer() = default;
  virtual Status scoreLattice(const jppgpu_result_view& lattice, uint32_t scorerIdx, float* cells) = 0;
};

// core::analysis::ScorerFactory (score_api.h:61-64).  `load` is the owner's call, as in the reference (RnnHolder::load ->
// RnnScorerGbeamFactory::load(ModelInfo), src/core/env.cc:52-63): a factory reads what it needs from the loaded model
// -- the ModelImage stands where the reference passes model::ModelInfo -- before analyzers are made from it.
class ScorerFactory {
 public:
  virtual ~ScorerFactory() = default;
  virtual Status load(const ModelImage& model) {
    (void)model;
    return Status::Ok();
  }
  virtual Status makeInstance(std::unique_ptr<ScoreComputer>* result) = 0;
  // the model's own RNN (RnnScorerGbeamFactory): scored by the device kernels, no ScoreComputer instance is made
  virtual bool isModelRnn() const { return false; }
};

// the RnnHolder's scorer factory (src/core/analysis/rnn_scorer_gbeam.h): stands for the RNN part of the loaded model
class ModelRnnScorerFactory : public ScorerFactory {
 public:
  // RnnScorerGbeamFactory::load fails on a model without an RNN part (rnn_scorer_gbeam.cc:377-393)
  Status load(const ModelImage& model) override {
    if (!model.hasRnn()) return Status::InvalidState("the model has no RNN part");
    return Status::Ok();
  }
  Status makeInstance(std::unique_ptr<ScoreComputer>*) override { return Status::Ok(); }
  bool isModelRnn() const override { return true; }
};

// core::analysis::FeatureScorer as far as a device can honour it (score_api.h:44-52): the hashed perceptron IS its weight
// table (HashedFeaturePerceptron, perceptron.h:75-111); a table given here replaces the model's (same size).
struct FeatureScorer {
  const float* weights = nullptr;   // null: the model's own table
  size_t size = 0;
};

// core::analysis::ScorerDef (score_api.h:66-72): `feature` (null = the model's perceptron), the other scorers in order,
// one weight per scorer.  The model's RNN, when used, must be others[0] (its cells are slot 1 on the device); any other
// factory makes a host ScoreComputer that is called once per batch (jppgpu_analyze_batch_scored), after which the device
// re-makes the beam totals and the EOS beam from the weighted cells like adjustBeamScores / remakeEosBeam.
// `useRnn = true` is shorthand for "others starts with the model's RNN" (the CLI's only case).
struct ScorerDef {
  bool useRnn = false;
  const FeatureScorer* feature = nullptr;
  std::vector<ScorerFactory*> others;
  std::vector<float> scoreWeights;
  int32_t numScorers() const { return (int32_t)(1 + (useRnn ? 1 : 0) + others.size()); }
};

// core::analysis::ScorePlugin (score_plugin.h:14-19) in its batched forms.  The reference asks the plugin for every scored
// connection (updateScore(lattice, connection, &score)); here the plugin sees the built lattice of a batch once and says
//   nodePenalties:        per node, what every connection INTO that node loses (jppgpu_analyze_batch_plugin), or
//   connectionPenalties:  per (left node, right node) pair of every boundary, what a connection between the two loses
//                         (jppgpu_analyze_batch_pairs; perConnection() must return true).
// One subtraction per connection; an amount that depends on the beam slot or the path history cannot be expressed.
class ScorePlugin {
 public:
  virtual ~ScorePlugin() = default;
  virtual bool perConnection() const { return false; }
  // lattice.n_sentences sentences in the order of the batch (or of one beam group of it, see groupSentences);
  // penalty[lattice.node_base[i] + k] belongs to node k of sentence i and arrives zeroed
  virtual void nodePenalties(const jppgpu_lattice_nodes& lattice, const std::vector<uint32_t>& sentenceIds, float* penalty) {
    (void)lattice; (void)sentenceIds; (void)penalty;
  }
  // penalty[lattice.pair_base[bb] + left * lattice.bnd_count[bb] + right], zeroed
  virtual void connectionPenalties(const jppgpu_lattice_pairs& lattice, const std::vector<uint32_t>& sentenceIds, float* penalty) {
    (void)lattice; (void)sentenceIds; (void)penalty;
  }
};

// the formatted text of one batch, detached from the analyzer that produced it (host copies of a result stay valid
// until it is released, also across later batches of the context): lets a writer keep the bytes while the analyzer
// takes its next batch
struct TextBatch {
  jppgpu_result* result = nullptr;
  jppgpu_text_view view{};
  const uint32_t* headLen = nullptr;   // lattice text: bytes of the "# MA-SCORE" line per sentence (jppgpu_lattice_text_view)
  TextBatch() = default;
  TextBatch(const TextBatch&) = delete;
  TextBatch& operator=(const TextBatch&) = delete;
  TextBatch(TextBatch&& o) noexcept : result(o.result), view(o.view), headLen(o.headLen) { o.result = nullptr; }
  TextBatch& operator=(TextBatch&& o) noexcept {
    if (this != &o) {
      reset();
      result = o.result;
      view = o.view;
      headLen = o.headLen;
      o.result = nullptr;
    }
    return *this;
  }
  ~TextBatch() { reset(); }
  void reset() {
    if (result) jppgpu_result_release(result);
    result = nullptr;
  }
};

struct SentenceResult {
  StringPiece input;
  uint32_t numCodepoints = 0;
  uint32_t numNodes = 0;
  // [numNodes].  After analyzeBatch(inputs, fullLattice=true): the lattice's node table (0/1 = BOS,
  // last = EOS); otherwise only the nodes of the top-1 path, in path order (JPPGPU_FETCH_TOP1)
  const jppgpu_node* nodes = nullptr;
  const jppgpu_unk* unk = nullptr;     // [numNodes]
  const uint32_t* pathNodes = nullptr; // top-1 path, EOS first
  uint32_t pathLen = 0;
  const uint32_t* cpByteOffsets = nullptr;  // [numCodepoints + 1]
  // byte span of a node's surface in `input`
  StringPiece surface(const jppgpu_node& n) const {
    return StringPiece(input.data() + cpByteOffsets[n.start], cpByteOffsets[n.end] - cpByteOffsets[n.start]);
  }
};

class GpuAnalyzer {
  const ModelImage* model_ = nullptr;
  jppgpu_ctx* ctx_ = nullptr;
  // one device result per group of sentences analysed with the same beam (a single group without auto-beam)
  struct Group {
    jppgpu_result* result = nullptr;
    jppgpu_result_view view{};
    jppgpu_nbest_view nbest{};   // filled instead of the lattice arrays of `view` in n-best mode
    bool hasNbest = false;
    int32_t beam = 0;
  };
  int32_t latticeNBest_ = 0;
  bool textMode_ = false;          // results are fetched as formatted text (jppgpu_result_format_top1)
  bool haveFormatTable_ = false, haveLatticeTable_ = false;
  int32_t latticeTextN_ = 0;       // > 0: the text is the lattice format of the N best paths (jppgpu_result_format_lattice)
  bool deferText_ = false, textFetched_ = false;
  const void* memoImage_ = nullptr;
  uint64_t memoImageBytes_ = 0;
  uint32_t memoImageSlots_ = 0;
  bool keepMemoImage_ = false;
  jppgpu_text_view text_{};
  const uint32_t* textHeads_ = nullptr;   // lattice text mode: header bytes per sentence
  float lastFormatMs_[2] = {0.f, 0.f};    // device time of the last fetchText(): count pass + scan, write pass
  std::vector<Group> groups_;
  std::vector<uint32_t> groupOf_, localIdx_;  // sentence -> (group, index inside the group's batch)
  AnalyzerConfig cfg_;
  ScoringConfig sconf_;
  std::vector<std::unique_ptr<ScoreComputer>> hostScorers_;   // ScorerDef::others that are not the model's RNN
  std::vector<StringPiece> inputs_;
  mutable std::vector<uint32_t> cpOffsets_;  // concatenated per-sentence codepoint -> byte offset tables
  mutable std::vector<uint8_t> cpOffsetsReady_;
  std::vector<uint64_t> cpOffsetsBase_;
  std::string singleInput_;
  PartialBatch partial_;
  std::vector<const PartialExample*> partialExamples_;
  Status runBatch(const std::vector<StringPiece>& inputs, bool fullLattice, const jppgpu_partial* partial,
                  ScorePlugin* plugin = nullptr);

  void releaseResult();

 public:
  GpuAnalyzer() = default;
  GpuAnalyzer(const GpuAnalyzer&) = delete;
  GpuAnalyzer& operator=(const GpuAnalyzer&) = delete;
  ~GpuAnalyzer();

  // shareModelWith: an initialised analyzer of the same model on the same device whose copy of the model in HBM
  // (dictionary, weights, RNN tables, T0 records, format table) this one uses instead of uploading its own -- the
  // second Analyzer over one CoreHolder (JumanppEnv::makeAnalyzer, src/core/env.cc:109-121)
  Status initialize(const ModelImage* model, const AnalyzerConfig& cfg, const ScoringConfig& sconf,
                    const ScorerDef* scorer, int device = 0, const GpuAnalyzer* shareModelWith = nullptr);
  // one sentence; the Status is the sentence's own status
  Status analyze(StringPiece input);
  // n sentences, one launch sequence; a failing sentence does not fail the batch (see sentenceStatus)
  Status analyzeBatch(const std::vector<StringPiece>& inputs, bool fullLattice = false);

  // Analyzer::analyze(input, plugin) for a batch, with any plugin of the batched form above
  Status analyzeBatch(const std::vector<StringPiece>& inputs, ScorePlugin* plugin, bool fullLattice = false);

  // Analyzer::analyze(surface, plugin) with the partial-annotation ScorePlugin for every example
  // (PexStreamReader::analyzeWith, pex_stream_reader.cc:61-66); a null entry is analysed as the empty string
  Status analyzeBatchPartial(const std::vector<const PartialExample*>& examples, bool fullLattice = false);

  // With n > 0, analyzeBatch(inputs, fullLattice = true) copies only what the lattice output format reads
  // -- the n best paths' beam slots, nodes and score cells, gathered on the device
  // (jppgpu_result_fetch_nbest) -- instead of the whole lattice; with auto-beam n is the sentence's beam.
  void setLatticeNBest(int32_t n) { latticeNBest_ = n; }
  // OutputFormat::format on the device (include/jppgpu.h: jppgpu_format_table): after setFormatTable, text mode makes
  // analyzeBatch(inputs) fetch the formatted top-1 analyses of the batch -- bytes and one offset per sentence -- instead
  // of node tables.  Not with auto-beam (several groups per batch) or a full-lattice fetch; sentence(i) is not
  // available in text mode, sentenceStatus(i) and sentenceText(i) are.
  Status setFormatTable(const jppgpu_format_table& table);
  bool setTextMode(bool on) {
    textMode_ = on && haveFormatTable_ && cfg_.autoBeamStep <= 0;
    latticeTextN_ = 0;
    return textMode_ == on;
  }
  // LatticeFormat::format on the device (jppgpu_lattice_table, csrc/k_latfmt.h): after setLatticeTable, lattice text
  // mode makes analyzeBatch fetch the lattice-format text of the n best paths of every sentence (also when called with
  // fullLattice = true: nothing of the lattice itself crosses PCIe).  batchTextHeads() says how many bytes of a
  // sentence's text are the "# MA-SCORE" line a comment replaces.  Not with auto-beam; n <= 64.
  Status setLatticeTable(const jppgpu_lattice_table& table);
  bool setLatticeTextMode(int32_t n) {
    const bool ok = n > 0 && n <= 64 && haveLatticeTable_ && cfg_.autoBeamStep <= 0 && cfg_.globalBeamSize > 0;
    textMode_ = ok;
    latticeTextN_ = ok ? n : 0;
    return ok;
  }
  bool textMode() const { return textMode_; }
  // deferred: analyzeBatch only analyses; fetchText() -- from any thread, before the analyzer's next batch -- runs the
  // format kernels and copies the text (so that the copy of batch k overlaps the analysis of batch k + 1 on the
  // device's other analyzer)
  void setDeferredText(bool on) { deferText_ = on; }
  Status fetchText();
  // device time of the format kernels of the last fetchText(): [0] count pass + offset scan, [1] write pass (ms)
  void lastFormatTimings(float ms[2]) const {
    ms[0] = lastFormatMs_[0];
    ms[1] = lastFormatMs_[1];
  }
  // hands the batch's text (and the result that owns it) to the caller; sentenceStatus / sentenceText are gone with it
  TextBatch takeText();
  StringPiece sentenceText(size_t i) const {
    return StringPiece(text_.text + text_.offsets[i], (size_t)(text_.offsets[i + 1] - text_.offsets[i]));
  }
  // the whole batch: sentence i is text[offsets[i] .. offsets[i + 1])
  const jppgpu_text_view& batchText() const { return text_; }
  // lattice text mode: [n] bytes of the "# MA-SCORE" line each sentence's text starts with (a comment replaces it); else null
  const uint32_t* batchTextHeads() const { return textHeads_; }
  // the n-best view holding sentence i (nullptr unless the batch was fetched in n-best mode)
  const jppgpu_nbest_view* nbestOf(size_t i, uint32_t* local) const {
    *local = localIdx_[i];
    const Group& g = groups_[groupOf_[i]];
    return g.hasNbest ? &g.nbest : nullptr;
  }

  bool ready() const { return ctx_ != nullptr; }   // initialize succeeded
  size_t numSentences() const { return inputs_.size(); }
  Status sentenceStatus(size_t i) const;
  SentenceResult sentence(size_t i) const;
  // the result view holding sentence i and its index inside that view
  const jppgpu_result_view& viewOf(size_t i, uint32_t* local) const {
    *local = localIdx_[i];
    return groups_[groupOf_[i]].view;
  }
  // AnalyzerImpl::autoBeamSizes for sentence i: its beam in auto-beam mode, 0 otherwise
  int32_t autoBeamSize(size_t i) const { return cfg_.autoBeamStep > 0 ? groups_[groupOf_[i]].beam : 0; }
  const ModelImage& model() const { return *model_; }
  const AnalyzerConfig& cfg() const { return cfg_; }
  const ScoringConfig& scoringConfig() const { return sconf_; }
  // [0]decode [1]seeds [2]layout [3]t0 [4]sweep [5]rnn [6]path [7]total, ms of the last batch
  void lastTimings(float ms[8]) const;
  // the derived per-entry T0 records of the model (jppgpu_config::t0_memo_image): handed in from a cache before
  // initialize(), or kept by initialize() for exportT0MemoImage() so that the caller can write the cache
  void setT0MemoImage(const void* data, uint64_t bytes, uint32_t slots) {
    memoImage_ = data;
    memoImageBytes_ = bytes;
    memoImageSlots_ = slots;
  }
  void setKeepT0MemoImage(bool keep) { keepMemoImage_ = keep; }
  bool exportT0MemoImage(const void** data, uint64_t* bytes, uint32_t* slots) const;
  // Every buffer of a batch of up to maxSentences sentences / maxBytes input bytes now (jppgpu_ctx_reserve): the batches
  // then allocate nothing and are one enqueue each.  textBytesPerByte != 0 (text mode): also the device text buffer and
  // `textBlocks` page-locked host blocks of that size.  A batch beyond the reservation still works (it is run again).
  Status reserve(uint32_t maxSentences, uint64_t maxBytes, float textBytesPerByte = 0.f, uint32_t textBlocks = 0);
  // one-enqueue batches / of which re-run / sized batches / device allocations of the process (jppgpu_ctx_stats)
  void pipelineStats(uint64_t out[4]) const;
};

}  // namespace jumanpp_amd

#endif  // JUMANPP_AMD_HOST_GPU_ANALYZER_H
