#include "gpu_analyzer.h"

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>

#include <cstring>

namespace jumanpp_amd {

namespace {
Status fromCode(int rc) {
  const char* msg = jppgpu_last_error();
  std::string m = msg ? msg : "";
  switch (rc) {
    case JPPGPU_OK: return Status::Ok();
    case JPPGPU_INVALID_PARAMETER: return Status(StatusCode::InvalidParameter, m);
    case JPPGPU_INVALID_STATE: return Status(StatusCode::InvalidState, m);
    case JPPGPU_NOT_IMPLEMENTED: return Status(StatusCode::NotImplemented, m);
    case JPPGPU_NO_DEVICE: return Status(StatusCode::NoDevice, m);
    case JPPGPU_OUT_OF_MEMORY: return Status(StatusCode::OutOfMemory, m);
    default: return Status(StatusCode::InvalidState, m);
  }
}
}  // namespace

GpuAnalyzer::~GpuAnalyzer() {
  releaseResult();
  if (ctx_) jppgpu_ctx_destroy(ctx_);
}

void GpuAnalyzer::releaseResult() {
  for (auto& g : groups_)
    if (g.result) jppgpu_result_release(g.result);
  groups_.clear();
  groupOf_.clear();
  localIdx_.clear();
}

Status GpuAnalyzer::initialize(const ModelImage* model, const AnalyzerConfig& cfg, const ScoringConfig& sconf,
                               const ScorerDef* scorer, int device, const GpuAnalyzer* shareModelWith) {
  if (model == nullptr) return Status::InvalidParameter("model was null");
  if (scorer == nullptr) return Status::InvalidParameter("scorer was null");
  // Analyzer::initialize / AnalyzerImpl::initScorers (analyzer.cc:16-36, analyzer_impl.cc:43-89)
  // ScorerDef::others: the model's RNN (device kernels) first when it is used, then host scorers
  bool useRnn = scorer->useRnn;
  std::vector<ScorerFactory*> hostFactories;
  for (size_t k = 0; k < scorer->others.size(); ++k) {
    ScorerFactory* f = scorer->others[k];
    if (f == nullptr) return Status::InvalidParameter("ScorerDef::others holds a null factory");
    if (f->isModelRnn()) {
      if (useRnn || k != 0) return Status::NotImplemented("the model's RNN must be the first (and only RNN) entry of ScorerDef::others");
      useRnn = true;
    } else {
      hostFactories.push_back(f);
    }
  }
  if (hostFactories.size() > 2) return Status::NotImplemented("at most two host scorers in ScorerDef::others");
  const int32_t nScorers = (int32_t)(1 + (useRnn ? 1 : 0) + hostFactories.size());
  if (sconf.numScorers != nScorers) {
    return Status::InvalidParameter() << "number of scorers in ScoringConfig (" << sconf.numScorers
                                      << ") does not match the ScorerDef (" << nScorers << ")";
  }
  if ((int32_t)scorer->scoreWeights.size() != nScorers) {
    return Status::InvalidParameter() << "ScorerDef has " << scorer->scoreWeights.size() << " score weights for "
                                      << nScorers << " scorers";
  }
  if (cfg.autoBeamStep < 0 || (cfg.autoBeamStep > 0 && (cfg.autoBeamBase <= 0 || cfg.autoBeamMax < cfg.autoBeamBase))) {
    return Status::InvalidParameter("auto beam: step must be positive and base <= max");
  }
  jppgpu_config c = JPPGPU_CONFIG_INIT;
  c.beam = sconf.beamSize;
  c.global_beam = cfg.globalBeamSize;
  c.right_check = cfg.rightGbeamCheck;
  c.right_beam = cfg.rightGbeamSize;
  c.max_input_bytes = (int32_t)cfg.maxInputBytes;
  c.device = device;
  c.use_rnn = useRnn ? 1 : 0;
  c.weight_perceptron = scorer->scoreWeights[0];
  c.weight_rnn = useRnn ? scorer->scoreWeights[1] : 0.f;
  c.num_host_scorers = (int32_t)hostFactories.size();
  for (size_t k = 0; k < hostFactories.size(); ++k) c.weight_host[k] = scorer->scoreWeights[1 + (useRnn ? 1 : 0) + k];
  // AnalyzerImpl::initScorers: one ScoreComputer instance per factory (analyzer_impl.cc:64-70)
  hostScorers_.clear();
  for (ScorerFactory* f : hostFactories) {
    std::unique_ptr<ScoreComputer> comp;
    JPPA_RETURN_IF_ERROR(f->makeInstance(&comp));
    if (!comp) return Status::InvalidState("a ScorerFactory made no ScoreComputer");
    hostScorers_.push_back(std::move(comp));
  }
  if (ctx_) {
    releaseResult();
    jppgpu_ctx_destroy(ctx_);
    ctx_ = nullptr;
    haveFormatTable_ = haveLatticeTable_ = false;
    textMode_ = false;
    latticeTextN_ = 0;
  }
  const std::vector<jppgpu_field_storage> storages = model->fieldStorages();
  c.field_storages = storages.empty() ? nullptr : storages.data();
  c.num_field_storages = (uint32_t)storages.size();
  c.t0_memo_image = memoImage_;
  c.t0_memo_image_bytes = memoImageBytes_;
  c.t0_memo_slots = memoImageSlots_;
  c.keep_t0_memo_image = keepMemoImage_ ? 1 : 0;
  int rc;
  if (shareModelWith != nullptr && shareModelWith->ctx_ != nullptr && shareModelWith->model_ == model) {
    rc = jppgpu_ctx_create_shared(shareModelWith->ctx_, &c, &ctx_);
    if (rc == JPPGPU_OK) {   // (the tables are part of the shared copy)
      haveFormatTable_ = shareModelWith->haveFormatTable_;
      haveLatticeTable_ = shareModelWith->haveLatticeTable_;
    }
  } else {
    rc = jppgpu_ctx_create(&model->cmodel(), &c, &ctx_);
  }
  if (rc != JPPGPU_OK) return fromCode(rc);
  model_ = model;
  cfg_ = cfg;
  sconf_ = sconf;
  // ScorerDef::feature: another weight table for the same hashed perceptron (HashedFeaturePerceptron, perceptron.h:75-111)
  if (scorer->feature != nullptr && scorer->feature->weights != nullptr) {
    rc = jppgpu_ctx_set_weights(ctx_, scorer->feature->weights, scorer->feature->size);
    if (rc != JPPGPU_OK) return fromCode(rc);
  }
  return Status::Ok();
}

Status GpuAnalyzer::analyze(StringPiece input) {
  singleInput_.assign(input.data(), input.size());
  std::vector<StringPiece> one{StringPiece(singleInput_)};
  JPPA_RETURN_IF_ERROR(analyzeBatch(one, false));
  return sentenceStatus(0);
}

Status GpuAnalyzer::analyzeBatch(const std::vector<StringPiece>& inputs, bool fullLattice) {
  return runBatch(inputs, fullLattice, nullptr);
}

namespace {

// PexStreamReaderImpl::updateScore + PartialExample::checkViolation (pex_stream_reader.cc:24-39,
// partial_example.cc:23-73) as a host-side ScorePlugin of the batched form: the same decision the device
// makes in k_penalty, taken here per lattice node.  Used when JPPGPU_PARTIAL_VIA_PLUGIN is set (tests of the
// generic plugin entry point); the default path evaluates the constraints on the device.
class PartialExamplePlugin : public ScorePlugin {
  const std::vector<const PartialExample*>& examples_;

 public:
  explicit PartialExamplePlugin(const std::vector<const PartialExample*>& ex) : examples_(ex) {}
  void nodePenalties(const jppgpu_lattice_nodes& lat, const std::vector<uint32_t>& ids, float* penalty) override {
    for (uint32_t s = 0; s < lat.n_sentences; ++s) {
      const PartialExample* ex = examples_[ids[s]];
      if (ex == nullptr) continue;
      for (uint32_t k = 2; k < lat.n_nodes[s]; ++k) {
        const jppgpu_node& nd = lat.nodes[lat.node_base[s] + k];
        const int32_t boundary = (int32_t)nd.start + 2, len = (int32_t)nd.end - nd.start, end = boundary + len;
        bool hard = false, tag = false, done = false;
        for (int32_t b : ex->noBreak) {
          if (b == boundary || b == end) {
            hard = done = true;
            break;
          }
          if (b > end) break;
        }
        if (!done)
          for (int32_t b : ex->boundaries) {
            if (b <= boundary) continue;
            if (b >= end) break;
            hard = done = true;
            break;
          }
        if (!done)
          for (const NodeConstraint& c : ex->nodes) {
            if (c.boundary != boundary) continue;
            if (len != c.length) hard = true;
            else {
              const int32_t* row = lat.entry_rows + (lat.node_base[s] + k) * (uint64_t)lat.num_features;
              for (const TagConstraint& t : c.tags)
                if (row[t.field] != t.value) {
                  tag = true;
                  break;
                }
            }
            break;  // std::find_if: only the first constraint at that boundary counts
          }
        penalty[lat.node_base[s] + k] = hard ? 10000.f : (tag ? 1000.f : 0.f);
      }
    }
  }
};

struct PluginCall {
  ScorePlugin* plugin;
  const std::vector<uint32_t>* ids;
};

void pluginTrampoline(void* user, const jppgpu_lattice_nodes* lattice, float* penalty) {
  auto* c = static_cast<PluginCall*>(user);
  c->plugin->nodePenalties(*lattice, *c->ids, penalty);
}

void pairTrampoline(void* user, const jppgpu_lattice_pairs* lattice, float* penalty) {
  auto* c = static_cast<PluginCall*>(user);
  c->plugin->connectionPenalties(*lattice, *c->ids, penalty);
}

int scorerTrampoline(void* user, const jppgpu_result_view* lattice, uint32_t scorerIdx, float* cells) {
  return static_cast<ScoreComputer*>(user)->scoreLattice(*lattice, scorerIdx, cells).isOk() ? 0 : 1;
}

}  // namespace

Status GpuAnalyzer::analyzeBatch(const std::vector<StringPiece>& inputs, ScorePlugin* plugin, bool fullLattice) {
  return runBatch(inputs, fullLattice, nullptr, plugin);
}

Status GpuAnalyzer::analyzeBatchPartial(const std::vector<const PartialExample*>& examples, bool fullLattice) {
  partialExamples_ = examples;
  std::vector<StringPiece> inputs;
  for (auto e : examples) inputs.push_back(e ? StringPiece(e->surface) : StringPiece(""));
  static const bool viaPlugin = std::getenv("JPPGPU_PARTIAL_VIA_PLUGIN") != nullptr;
  if (viaPlugin) {
    PartialExamplePlugin plugin(partialExamples_);
    return runBatch(inputs, fullLattice, nullptr, &plugin);
  }
  partial_.build(examples);
  return runBatch(inputs, fullLattice, &partial_.view);
}

Status GpuAnalyzer::runBatch(const std::vector<StringPiece>& inputs, bool fullLattice, const jppgpu_partial* partial,
                             ScorePlugin* plugin) {
  if (!ctx_) return Status::InvalidState("GpuAnalyzer was not initialized");
  releaseResult();
  inputs_ = inputs;
  const size_t n = inputs.size();
  size_t total = 0;
  for (auto& s : inputs) total += s.size();
  if (total >= 0xffffffffull) return Status::InvalidParameter("batch is larger than 4 GiB");
  groupOf_.assign(n, 0);
  localIdx_.assign(n, 0);
  // sentences grouped by their beam: one group unless auto-beam is on
  std::vector<int32_t> beams;               // beam of group g
  std::vector<std::vector<uint32_t>> members;
  if (cfg_.autoBeamStep > 0) {
    for (size_t i = 0; i < n; ++i) {
      size_t ncp = 0;  // AnalysisInput::numCodepoints (lead bytes)
      for (size_t b = 0; b < inputs[i].size(); ++b) ncp += ((unsigned char)inputs[i][b] & 0xc0) != 0x80;
      int32_t beam = cfg_.autoBeamBase + (int32_t)(ncp / (size_t)cfg_.autoBeamStep);
      if (beam > cfg_.autoBeamMax) beam = cfg_.autoBeamMax;
      size_t g = 0;
      while (g < beams.size() && beams[g] != beam) ++g;
      if (g == beams.size()) {
        beams.push_back(beam);
        members.emplace_back();
      }
      groupOf_[i] = (uint32_t)g;
      localIdx_[i] = (uint32_t)members[g].size();
      members[g].push_back((uint32_t)i);
    }
  } else {
    beams.push_back(sconf_.beamSize);
    members.emplace_back();
    for (size_t i = 0; i < n; ++i) {
      localIdx_[i] = (uint32_t)i;
      members[0].push_back((uint32_t)i);
    }
  }
  std::string text;
  std::vector<uint32_t> offsets;
  PartialBatch sub;
  const bool hostTiming = std::getenv("JPPGPU_HOST_TIMING") != nullptr;  // developer: stage times to stderr
  auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  double tStart = now(), tAnalyze = 0, tFetch = 0;
  for (size_t g = 0; g < beams.size(); ++g) {
    text.clear();
    offsets.assign(1, 0);
    for (uint32_t i : members[g]) {
      text.append(inputs[i].data(), inputs[i].size());
      offsets.push_back((uint32_t)text.size());
    }
    const jppgpu_partial* pg = partial;
    if (cfg_.autoBeamStep > 0) {
      int rc = jppgpu_ctx_set_beams(ctx_, beams[g], beams[g], cfg_.rightGbeamCheck, cfg_.rightGbeamSize);
      if (rc != JPPGPU_OK) return fromCode(rc);
      if (partial) {  // the group's slice of the constraints
        std::vector<const PartialExample*> ex;
        for (uint32_t i : members[g]) ex.push_back(partialExamples_[i]);
        sub.build(ex);
        pg = &sub.view;
      }
    }
    groups_.emplace_back();
    Group& G = groups_.back();
    G.beam = beams[g];
    const uint32_t ng = (uint32_t)members[g].size();
    double t0 = now();
    PluginCall call{plugin, &members[g]};
    int rc;
    if (!hostScorers_.empty()) {
      // ScorerDef::others on the host (no plugin on this path: the reference's plugin acts inside the sweep, the host
      // scorers after it -- both at once would need the pair matrix AND the scored lattice; not built)
      if (pg || plugin) return Status::NotImplemented("host scorers together with a score plugin");
      jppgpu_score_lattice_fn fns[2] = {scorerTrampoline, scorerTrampoline};
      void* users[2] = {hostScorers_[0].get(), hostScorers_.size() > 1 ? hostScorers_[1].get() : nullptr};
      rc = jppgpu_analyze_batch_scored(ctx_, text.data(), offsets.data(), ng, fns, users, (uint32_t)hostScorers_.size(), &G.result);
    } else {
      rc = pg       ? jppgpu_analyze_batch_partial(ctx_, text.data(), offsets.data(), ng, pg, &G.result)
           : plugin ? (plugin->perConnection()
                           ? jppgpu_analyze_batch_pairs(ctx_, text.data(), offsets.data(), ng, pairTrampoline, &call, &G.result)
                           : jppgpu_analyze_batch_plugin(ctx_, text.data(), offsets.data(), ng, pluginTrampoline, &call, &G.result))
                    : jppgpu_analyze_batch(ctx_, text.data(), offsets.data(), ng, &G.result);
    }
    if (rc != JPPGPU_OK) return fromCode(rc);
    double t1 = now();
    tAnalyze += t1 - t0;
    // host copies are taken right away: the next group's batch invalidates the device side of this result
    const bool asText = textMode_ && (latticeTextN_ > 0 || !fullLattice);
    if (asText) {
      textFetched_ = false;
      text_ = jppgpu_text_view{};
      G.view = jppgpu_result_view{};
      if (!deferText_) JPPA_RETURN_IF_ERROR(fetchText());
    } else if (fullLattice && latticeNBest_ > 0) {
      const int32_t nBest = std::min<int32_t>(64, cfg_.autoBeamStep > 0 ? G.beam : latticeNBest_);
      rc = jppgpu_result_fetch_nbest(G.result, nBest, &G.nbest);
      if (rc != JPPGPU_OK) return fromCode(rc);
      G.hasNbest = true;
      G.view = jppgpu_result_view{};
      G.view.n_sentences = G.nbest.n_sentences;
      G.view.status = G.nbest.status;
      G.view.n_codepoints = G.nbest.n_codepoints;
      G.view.n_nodes = G.nbest.n_nodes;
      G.view.beam = G.nbest.beam;
      G.view.global_beam = G.nbest.global_beam;
      G.view.num_scorers = G.nbest.num_scorers;
    } else {
      rc = jppgpu_result_fetch(G.result, fullLattice ? JPPGPU_FETCH_FULL : JPPGPU_FETCH_TOP1, &G.view);
      if (rc != JPPGPU_OK) return fromCode(rc);
    }
    tFetch += now() - t1;
  }
  double tGroups = now();
  // codepoint -> byte offset tables (for surfaces): sized here, filled by the first sentence(i) call,
  // which may come from any format worker (distinct sentences write distinct ranges)
  cpOffsetsBase_.assign(n + 1, 0);
  uint64_t cpTotal = 0;
  for (size_t i = 0; i < n && !(textMode_ && (latticeTextN_ > 0 || !fullLattice)); ++i) {
    cpOffsetsBase_[i] = cpTotal;
    const jppgpu_result_view& v = groups_[groupOf_[i]].view;
    if (v.status[localIdx_[i]] == JPPGPU_SENT_OK) cpTotal += (uint64_t)v.n_codepoints[localIdx_[i]] + 1;
  }
  cpOffsets_.assign(cpTotal, 0);
  cpOffsetsReady_.assign(n, 0);
  cpOffsetsBase_[n] = cpTotal;
  if (hostTiming)
    std::fprintf(stderr, "runBatch n=%zu total=%.2f ms: prepare %.2f analyze %.2f fetch %.2f offsets %.2f\n", n, now() - tStart,
                 tGroups - tStart - tAnalyze - tFetch, tAnalyze, tFetch, now() - tGroups);
  return Status::Ok();
}

Status GpuAnalyzer::sentenceStatus(size_t i) const {
  if (groups_.empty() || i >= inputs_.size()) return Status::InvalidState("no result for this sentence");
  if (groups_[groupOf_.empty() ? 0 : groupOf_[i]].view.status == nullptr) return Status::InvalidState("the batch's text was not fetched");
  switch (groups_[groupOf_[i]].view.status[localIdx_[i]]) {
    case JPPGPU_SENT_OK: return Status::Ok();
    case JPPGPU_SENT_TOO_LONG:
      return Status::InvalidParameter() << "byte size of input string (" << inputs_[i].size()
                                        << ") is greater than maximum allowed (" << cfg_.maxInputBytes << ")";
    case JPPGPU_SENT_BAD_UTF8: return Status::InvalidParameter() << "Invalid UTF8 sequence: " << inputs_[i];
    case JPPGPU_SENT_NO_LATTICE: return Status::InvalidState("could not build lattice");
    default: return Status::InvalidState("a device staging capacity was exceeded for this sentence");
  }
}

SentenceResult GpuAnalyzer::sentence(size_t i) const {
  SentenceResult r;
  const jppgpu_result_view& v = groups_[groupOf_[i]].view;
  const uint32_t k = localIdx_[i];
  r.input = inputs_[i];
  if (v.n_codepoints == nullptr) return r;   // text mode: no node tables were fetched
  r.numCodepoints = v.n_codepoints[k];
  r.numNodes = v.n_nodes[k];
  if (v.node_base != nullptr) {  // (n-best mode carries no node table: the paths' node records are in the n-best items)
    r.nodes = v.nodes + v.node_base[k];
    r.unk = v.unk + v.node_base[k];
    r.pathNodes = v.path_nodes + v.node_base[k];
    r.pathLen = v.path_len[k];
  }
  uint32_t* offs = cpOffsets_.data() + cpOffsetsBase_[i];
  if (!cpOffsetsReady_[i] && v.status[k] == JPPGPU_SENT_OK) {
    const unsigned char* p = reinterpret_cast<const unsigned char*>(inputs_[i].data());
    const size_t nb = inputs_[i].size();
    uint32_t* o = offs;
    for (size_t b = 0; b < nb;) {
      *o++ = (uint32_t)b;
      unsigned char c = p[b];
      b += c < 0x80 ? 1 : c < 0xe0 ? 2 : c < 0xf0 ? 3 : 4;
    }
    *o = (uint32_t)nb;
    cpOffsetsReady_[i] = 1;
  }
  r.cpByteOffsets = offs;
  return r;
}

Status GpuAnalyzer::fetchText() {
  if (!textMode_ || groups_.size() != 1 || groups_[0].result == nullptr) return Status::InvalidState("no text-mode batch to fetch");
  if (textFetched_) return Status::Ok();
  Group& G = groups_[0];
  static const bool hostTiming = std::getenv("JPPGPU_HOST_TIMING") != nullptr;
  const auto t0 = std::chrono::steady_clock::now();
  int rc;
  textHeads_ = nullptr;
  if (latticeTextN_ > 0) {
    jppgpu_lattice_text_view lv{};
    rc = jppgpu_result_format_lattice(G.result, latticeTextN_, &lv);
    text_ = jppgpu_text_view{lv.n_sentences, lv.offsets, lv.text, lv.status};
    textHeads_ = lv.head_len;
  } else {
    rc = jppgpu_result_format_top1(G.result, &text_);
  }
  if (rc != JPPGPU_OK) return fromCode(rc);
  if (hostTiming)
    std::fprintf(stderr, "fetchText n=%u bytes=%llu total=%.2f ms\n", text_.n_sentences, (unsigned long long)text_.offsets[text_.n_sentences],
                 std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
  {
    float ms[19] = {0};
    if (jppgpu_last_timings(ctx_, ms, 19) == JPPGPU_OK) {
      lastFormatMs_[0] = ms[16];
      lastFormatMs_[1] = ms[17];
    }
  }
  G.view.n_sentences = text_.n_sentences;
  G.view.status = text_.status;
  textFetched_ = true;
  return Status::Ok();
}

TextBatch GpuAnalyzer::takeText() {
  TextBatch tb;
  if (textFetched_ && groups_.size() == 1) {
    tb.result = groups_[0].result;
    tb.view = text_;
    tb.headLen = textHeads_;
    textHeads_ = nullptr;
    groups_[0].result = nullptr;
    groups_.clear();
    textFetched_ = false;
    text_ = jppgpu_text_view{};
  }
  return tb;
}

Status GpuAnalyzer::setFormatTable(const jppgpu_format_table& table) {
  if (!ctx_) return Status::InvalidState("GpuAnalyzer::setFormatTable before initialize");
  int rc = jppgpu_ctx_set_format_table(ctx_, &table);
  if (rc != JPPGPU_OK) return fromCode(rc);
  haveFormatTable_ = true;
  return Status::Ok();
}

Status GpuAnalyzer::setLatticeTable(const jppgpu_lattice_table& table) {
  if (!ctx_) return Status::InvalidState("GpuAnalyzer::setLatticeTable before initialize");
  int rc = jppgpu_ctx_set_lattice_table(ctx_, &table);
  if (rc != JPPGPU_OK) return fromCode(rc);
  haveLatticeTable_ = true;
  return Status::Ok();
}

Status GpuAnalyzer::reserve(uint32_t maxSentences, uint64_t maxBytes, float textBytesPerByte, uint32_t textBlocks) {
  if (!ctx_) return Status::InvalidState("GpuAnalyzer::reserve before initialize");
  jppgpu_reserve r;
  std::memset(&r, 0, sizeof(r));
  r.struct_size = (uint32_t)sizeof(r);
  r.max_sentences = maxSentences;
  r.max_total_bytes = maxBytes;
  // lattice nodes per input byte to provide for: 2.1 on the 10^6-row bench dictionary, 2.2 on 220-codepoint sentences.
  // Every GB reserved is a GB the driver has to hand out (and scrub after the previous process) before the first batch:
  // 40 GB across four analyzers cost 2.4 s in a session where the same run with warm memory took 8 ms
  // (profiles/r05h); a denser lattice only costs the second run of its first batch.
  r.nodes_per_byte = 2.5f;
  r.text_bytes_per_byte = textMode_ ? textBytesPerByte : 0.f;
  r.text_host_blocks = textMode_ ? textBlocks : 0;
  int rc = jppgpu_ctx_reserve(ctx_, &r);
  return rc == JPPGPU_OK ? Status::Ok() : fromCode(rc);
}

bool GpuAnalyzer::exportT0MemoImage(const void** data, uint64_t* bytes, uint32_t* slots) const {
  if (!ctx_) return false;
  return jppgpu_ctx_t0_memo_image(ctx_, data, bytes, slots) == JPPGPU_OK && *bytes != 0;
}

void GpuAnalyzer::pipelineStats(uint64_t out[4]) const {
  out[0] = out[1] = out[2] = out[3] = 0;
  if (!ctx_) return;
  jppgpu_ctx_statistics st;
  std::memset(&st, 0, sizeof(st));
  st.struct_size = (uint32_t)sizeof(st);
  if (jppgpu_ctx_stats(ctx_, &st) != JPPGPU_OK) return;
  out[0] = st.one_enqueue_batches;
  out[1] = st.one_enqueue_overflows;
  out[2] = st.sized_batches;
  out[3] = st.device_allocations;
}

void GpuAnalyzer::lastTimings(float ms[8]) const {
  for (int i = 0; i < 8; ++i) ms[i] = 0.f;
  if (ctx_) jppgpu_last_timings(ctx_, ms, 8);
}

}  // namespace jumanpp_amd
