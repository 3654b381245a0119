#include "gpu_analyzer.h"

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
  if (result_) {
    jppgpu_result_release(result_);
    result_ = nullptr;
  }
  std::memset(&view_, 0, sizeof(view_));
}

Status GpuAnalyzer::initialize(const ModelImage* model, const AnalyzerConfig& cfg, const ScoringConfig& sconf,
                               const ScorerDef* scorer, int device) {
  if (model == nullptr) return Status::InvalidParameter("model was null");
  if (scorer == nullptr) return Status::InvalidParameter("scorer was null");
  // Analyzer::initialize / AnalyzerImpl::initScorers (analyzer.cc:16-36, analyzer_impl.cc:43-89)
  const int32_t nScorers = scorer->useRnn ? 2 : 1;
  if (sconf.numScorers != nScorers) {
    return Status::InvalidParameter() << "number of scorers in ScoringConfig (" << sconf.numScorers
                                      << ") does not match the ScorerDef (" << nScorers << ")";
  }
  if ((int32_t)scorer->scoreWeights.size() != nScorers) {
    return Status::InvalidParameter() << "ScorerDef has " << scorer->scoreWeights.size() << " score weights for "
                                      << nScorers << " scorers";
  }
  if (cfg.autoBeamStep != 0 || cfg.autoBeamBase != 0 || cfg.autoBeamMax != 0) {
    return Status::NotImplemented("auto beam is not implemented on the GPU path");
  }
  jppgpu_config c{};
  c.beam = sconf.beamSize;
  c.global_beam = cfg.globalBeamSize;
  c.right_check = cfg.rightGbeamCheck;
  c.right_beam = cfg.rightGbeamSize;
  c.max_input_bytes = (int32_t)cfg.maxInputBytes;
  c.device = device;
  c.use_rnn = scorer->useRnn ? 1 : 0;
  c.weight_perceptron = scorer->scoreWeights[0];
  c.weight_rnn = scorer->useRnn ? scorer->scoreWeights[1] : 0.f;
  if (ctx_) {
    releaseResult();
    jppgpu_ctx_destroy(ctx_);
    ctx_ = nullptr;
  }
  int rc = jppgpu_ctx_create(&model->cmodel(), &c, &ctx_);
  if (rc != JPPGPU_OK) return fromCode(rc);
  model_ = model;
  cfg_ = cfg;
  sconf_ = sconf;
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

Status GpuAnalyzer::analyzeBatchPartial(const std::vector<const PartialExample*>& examples, bool fullLattice) {
  partial_.build(examples);
  std::vector<StringPiece> inputs;
  for (auto e : examples) inputs.push_back(e ? StringPiece(e->surface) : StringPiece(""));
  return runBatch(inputs, fullLattice, &partial_.view);
}

Status GpuAnalyzer::runBatch(const std::vector<StringPiece>& inputs, bool fullLattice, const jppgpu_partial* partial) {
  if (!ctx_) return Status::InvalidState("GpuAnalyzer was not initialized");
  releaseResult();
  inputs_ = inputs;
  text_.clear();
  offsets_.assign(1, 0);
  size_t total = 0;
  for (auto& s : inputs) total += s.size();
  if (total >= 0xffffffffull) return Status::InvalidParameter("batch is larger than 4 GiB");
  text_.reserve(total);
  for (auto& s : inputs) {
    text_.append(s.data(), s.size());
    offsets_.push_back((uint32_t)text_.size());
  }
  int rc = partial ? jppgpu_analyze_batch_partial(ctx_, text_.data(), offsets_.data(), (uint32_t)inputs.size(), partial, &result_)
                   : jppgpu_analyze_batch(ctx_, text_.data(), offsets_.data(), (uint32_t)inputs.size(), &result_);
  if (rc != JPPGPU_OK) return fromCode(rc);
  rc = jppgpu_result_fetch(result_, fullLattice ? 1 : 0, &view_);
  if (rc != JPPGPU_OK) return fromCode(rc);
  // codepoint -> byte offset tables of the well-formed sentences (for surfaces)
  cpOffsets_.clear();
  cpOffsetsBase_.assign(inputs.size() + 1, 0);
  for (size_t i = 0; i < inputs.size(); ++i) {
    cpOffsetsBase_[i] = cpOffsets_.size();
    if (view_.status[i] != JPPGPU_SENT_OK) continue;
    const unsigned char* p = reinterpret_cast<const unsigned char*>(inputs[i].data());
    size_t n = inputs[i].size();
    for (size_t b = 0; b < n;) {
      cpOffsets_.push_back((uint32_t)b);
      unsigned char c = p[b];
      b += c < 0x80 ? 1 : c < 0xe0 ? 2 : c < 0xf0 ? 3 : 4;
    }
    cpOffsets_.push_back((uint32_t)n);
  }
  cpOffsetsBase_[inputs.size()] = cpOffsets_.size();
  return Status::Ok();
}

Status GpuAnalyzer::sentenceStatus(size_t i) const {
  if (!result_ || i >= inputs_.size()) return Status::InvalidState("no result for this sentence");
  switch (view_.status[i]) {
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
  r.input = inputs_[i];
  r.numCodepoints = view_.n_codepoints[i];
  r.numNodes = view_.n_nodes[i];
  r.nodes = view_.nodes + view_.node_base[i];
  r.unk = view_.unk + view_.node_base[i];
  r.pathNodes = view_.path_nodes + view_.node_base[i];
  r.pathLen = view_.path_len[i];
  r.cpByteOffsets = cpOffsets_.data() + cpOffsetsBase_[i];
  return r;
}

void GpuAnalyzer::lastTimings(float ms[8]) const {
  for (int i = 0; i < 8; ++i) ms[i] = 0.f;
  if (ctx_) jppgpu_last_timings(ctx_, ms, 8);
}

}  // namespace jumanpp_amd
