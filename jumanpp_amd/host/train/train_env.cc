#include "train_env.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <fstream>
#include <random>
#include <sstream>
#include <thread>

namespace jumanpp_amd {
namespace train {

Status GlobalBeamParams::validate() const {   // training_env.cc:235-264
  if (leftEnabled() && maxLeftBeam < minLeftBeam) {
    return Status::InvalidParameter() << "min left beam size (" << minLeftBeam << ") should be smaller than max left beam size ("
                                      << maxLeftBeam << ")";
  }
  if (rightEnabled()) {
    if (!leftEnabled()) return Status::InvalidParameter() << "right beam won't work without left beam";
    if (minRightCheck < 1) return Status::InvalidParameter() << "right beam should check at least 1 left beam";
    if (maxRightCheck < minRightCheck) return Status::InvalidParameter() << "right beam check: max is lesser than min";
    if (maxRightBeam < minRightBeam) return Status::InvalidParameter() << "right beam: max is lesser than min";
  }
  return Status::Ok();
}

TrainingEnv::~TrainingEnv() {
  if (ctx_) jppgpu_ctx_destroy(ctx_);
}

namespace {
template <typename T>
T interpolate(T min, T max, float v) {
  T diff = max - min;
  return min + static_cast<T>(diff * v);
}
Status abiError(const char* what) { return Status::InvalidState() << what << ": " << jppgpu_last_error(); }
}  // namespace

Status TrainingEnv::initialize(const TrainingArguments& args, const ModelImage* model) {
  args_ = args;
  model_ = model;
  if (args.sizeExponent > 31) return Status::InvalidState() << "size exponent was too large: " << args.sizeExponent << ", maximum allowed is 31";
  JPPA_RETURN_IF_ERROR(args.globalBeam.validate());
  JPPA_RETURN_IF_ERROR(tio_.initialize(*model));
  JPPA_RETURN_IF_ERROR(resolver_.initialize(*model));
  reader_.initialize(&tio_, resolver_.surfaceColumn());
  scw_.reset(new SoftConfidenceWeighted(args.scw, args.sizeExponent, args.randomSeed));
  JPPA_RETURN_IF_ERROR(changeGlobalBeam(0.f));
  jppgpu_model m = model->cmodel();
  m.weights = scw_->weights().data();
  m.weight_exponent = args.sizeExponent;
  m.has_rnn = 0;
  jppgpu_config c = JPPGPU_CONFIG_INIT;
  c.beam = args.beamSize;
  c.global_beam = leftBeam_;
  c.right_check = rightCheck_;
  c.right_beam = rightBeam_;
  c.max_input_bytes = 4096;
  c.device = args.device;
  c.use_rnn = 0;
  c.weight_perceptron = 1.0f;
  c.weight_rnn = 0.f;
  c.dynamic_features = 1;   // TrainingEnv::initFeatures(nullptr): the trainer runs the dynamic feature code
  if (jppgpu_ctx_create(&m, &c, &ctx_) != JPPGPU_OK) return abiError("jppgpu_ctx_create");
  return Status::Ok();
}

Status TrainingEnv::loadInput(const std::string& filename) {
  std::ifstream f(filename, std::ios::binary);
  if (!f) return Status::InvalidParameter() << "could not open the corpus " << filename;
  std::ostringstream ss;
  ss << f.rdbuf();
  corpus_ = ss.str();
  resetInput();
  return Status::Ok();
}

Status TrainingEnv::changeGlobalBeam(float rawRatio) {
  const GlobalBeamParams& g = args_.globalBeam;
  const float func = 1.09574f * std::exp(-3.04689f * rawRatio) - 0.0957439f;
  const float ratio = std::max(0.0f, func);
  leftBeam_ = g.leftEnabled() ? interpolate(g.minLeftBeam, g.maxLeftBeam, ratio) : 0;
  if (g.rightEnabled()) {
    rightBeam_ = interpolate(g.minRightBeam, g.maxRightBeam, ratio);
    rightCheck_ = interpolate(g.minRightCheck, g.maxRightCheck, ratio);
  } else {
    rightBeam_ = 0;
    rightCheck_ = 0;
  }
  if (ctx_ && jppgpu_ctx_set_beams(ctx_, args_.beamSize, leftBeam_, rightCheck_, rightBeam_) != JPPGPU_OK) return abiError("jppgpu_ctx_set_beams");
  return Status::Ok();
}

// TrainerBatch::readFullBatch (trainer.cc:93-108) + shuffleData (trainer.cc:110-134)
Status TrainingEnv::readOneBatch() {
  batch_.clear();
  for (uint32_t i = 0; i < args_.batchSize; ++i) {
    GoldExample ex;
    JPPA_RETURN_IF_ERROR(reader_.readExample(&ex));
    if (reader_.finished()) break;   // (the example read together with the end of input is not used)
    batch_.push_back(std::move(ex));
  }
  order_.clear();
  for (size_t i = 0; i < batch_.size(); ++i) order_.push_back((int32_t)i);
  std::minstd_rand rng{args_.randomSeed * (static_cast<uint32_t>(numShuffles_) * 31 + 5)};
  std::shuffle(order_.begin(), order_.end(), rng);
  numShuffles_ += 1;
  return Status::Ok();
}

int TrainingEnv::seedHook(void* user, const jppgpu_seed_view* view, jppgpu_extra_seeds* out) {
  TrainingEnv* self = static_cast<TrainingEnv*>(user);
  const uint32_t n = view->n_sentences;
  self->goldPaths_.assign(n, {});
  self->extraOffsets_.assign((size_t)n + 1, 0);
  self->extraSeeds_.clear();
  for (uint32_t q = 0; q < n; ++q) {
    const GoldExample& ex = self->batch_[(size_t)self->order_[q]];
    Status s = self->resolver_.resolve(ex, *view, q, &self->goldPaths_[q], &self->extraSeeds_);
    if (!s) {
      self->hookStatus_ = Status(s.code(), s.message() + " [example on line " + std::to_string(ex.line()) + "]");
      return 1;
    }
    self->extraOffsets_[(size_t)q + 1] = (uint32_t)self->extraSeeds_.size();
  }
  out->offsets = self->extraOffsets_.data();
  out->seeds = self->extraSeeds_.data();
  return 0;
}

// TrainingEnv::trainOneBatch (training_env.cc:55-89): Trainer::prepare + compute for every example of the batch (one
// device pass), then handleProcessedTrainer per example in submission order
Status TrainingEnv::trainOneBatch(int32_t iter) {
  const uint32_t n = (uint32_t)order_.size();
  if (leftBeam_ > 0) {   // TrainingEnv::trainOneBatch (training_env.cc:64-71): ITrainer::setGlobalBeam per submitted trainer
    const bool full = args_.globalBeam.fullFirstIter && iter == 0;
    if (jppgpu_ctx_set_beams(ctx_, args_.beamSize, full ? 0 : leftBeam_, full ? 0 : rightCheck_, full ? 0 : rightBeam_) != JPPGPU_OK)
      return abiError("jppgpu_ctx_set_beams");
  }
  std::string text;
  std::vector<uint32_t> offsets(1, 0);
  for (uint32_t q = 0; q < n; ++q) {
    text += batch_[(size_t)order_[q]].surface();
    offsets.push_back((uint32_t)text.size());
  }
  hookStatus_ = Status::Ok();
  auto now = []() { return std::chrono::steady_clock::now(); };
  auto since = [&](std::chrono::steady_clock::time_point t) { return std::chrono::duration<double, std::milli>(now() - t).count(); };
  auto t0 = now();
  jppgpu_result* res = nullptr;
  if (jppgpu_analyze_batch_seeds(ctx_, text.data(), offsets.data(), n, &TrainingEnv::seedHook, this, &res) != JPPGPU_OK) {
    if (!hookStatus_) return hookStatus_;
    return abiError("jppgpu_analyze_batch_seeds");
  }
  struct Release {
    jppgpu_result* r;
    ~Release() { jppgpu_result_release(r); }
  } release{res};
  goldNodesAdded_ += extraSeeds_.size();
  stageMs_[0] += since(t0);
  t0 = now();
  jppgpu_result_view view{};
  if (jppgpu_result_fetch(res, JPPGPU_FETCH_FULL, &view) != JPPGPU_OK) return abiError("jppgpu_result_fetch");
  for (uint32_t q = 0; q < n; ++q) {
    if (view.status[q] != JPPGPU_SENT_OK) {
      const GoldExample& ex = batch_[(size_t)order_[q]];
      return Status::InvalidState() << (view.status[q] == JPPGPU_SENT_NO_LATTICE ? "could not build lattice for gold example"
                                                                                   : "the analyser rejected the example")
                                    << " on line " << ex.line();
    }
  }
  stageMs_[1] += since(t0);
  t0 = now();
  jppgpu_top1_ngrams_view top{};
  if (jppgpu_result_fetch_top1_ngrams(res, &top) != JPPGPU_OK) return abiError("jppgpu_result_fetch_top1_ngrams");
  // gold paths as node indices, EOS last (LossCalculator::resolveGold)
  std::vector<uint64_t> gfirst((size_t)n + 1, 0);
  std::vector<uint32_t> gnodes;
  for (uint32_t q = 0; q < n; ++q) {
    for (const auto& g : goldPaths_[q]) gnodes.push_back(view.bnd_first[view.bnd_base[q] + g.boundary] + g.position);
    gnodes.push_back(view.n_nodes[q] - 1);
    gfirst[(size_t)q + 1] = gnodes.size();
  }
  jppgpu_top1_ngrams_view gold{};
  if (jppgpu_result_fetch_path_ngrams(res, gfirst.data(), gnodes.data(), &gold) != JPPGPU_OK) return abiError("jppgpu_result_fetch_path_ngrams");
  stageMs_[2] += since(t0);
  t0 = now();
  const uint32_t mask = scw_->mask();
  // Every example of the batch is judged with the weights it was analysed with.  Gold scores, comparison, loss and
  // feature difference of an example depend on nothing but its own lattice (the reference computes them on its worker
  // threads, OwningFullTrainer::compute): one slice of the batch per host thread.  The SCW updates follow in batch order.
  std::vector<LossCalculator> loss(n);
  std::vector<float> lossValue(n, 0.f);
  std::vector<Status> st(n);
  auto judge = [&](uint32_t q) {
    LossCalculator& lc = loss[q];
    lc.initialize(&model_->trainingSpec());
    lc.computeGoldScores(scw_->weights().data(), mask, gold.features + gold.path_first[q] * gold.n_ngram, gold.n_ngram,
                         (size_t)(gold.path_first[q + 1] - gold.path_first[q]));
    SentenceLattice L;
    L.view = &view;
    L.s = q;
    L.numFeatures = model_->numFeatures();
    st[q] = lc.compare(L, goldPaths_[q], top.features + top.path_first[q] * top.n_ngram, (size_t)(top.path_first[q + 1] - top.path_first[q]));
    if (!st[q]) return;
    int32_t used = lc.fullSize();   // Trainer::computeTrainingLoss (trainer.cc:49-67)
    if (args_.mode == TrainingMode::FalloffBeam) used = lc.fallOffBeam();
    else if (args_.mode == TrainingMode::MaxViolation) used = lc.maxViolation();
    lossValue[q] = lc.computeLoss(used);
    lc.computeFeatureDiff(mask);
  };
  unsigned nt = std::thread::hardware_concurrency();
  nt = nt == 0 ? 1u : nt > 16 ? 16u : nt;
  if (n < 16 || nt == 1) {
    for (uint32_t q = 0; q < n; ++q) judge(q);
  } else {
    std::vector<std::thread> pool;
    for (unsigned t = 0; t < nt; ++t)
      pool.emplace_back([&, t]() {
        for (uint32_t q = t; q < n; q += nt) judge(q);
      });
    for (auto& th : pool) th.join();
  }
  for (uint32_t q = 0; q < n; ++q)
    if (!st[q]) return Status(st[q].code(), st[q].message() + " [example on line " + std::to_string(batch_[(size_t)order_[q]].line()) + "]");
  stageMs_[3] += since(t0);
  t0 = now();
  double curLoss = 0;
  for (uint32_t q = 0; q < n; ++q) {
    curLoss += lossValue[q];
    scw_->update(lossValue[q], loss[q].featureDiff());   // handleProcessedTrainer (training_env.cc:91-106)
    examplesSeen_ += 1;
  }
  batchLoss_ = curLoss;
  stageMs_[4] += since(t0);
  t0 = now();
  if (jppgpu_ctx_set_weights(ctx_, scw_->weights().data(), (uint64_t)scw_->weights().size()) != JPPGPU_OK) return abiError("jppgpu_ctx_set_weights");
  stageMs_[5] += since(t0);
  return Status::Ok();
}

Status TrainingEnv::trainOneEpoch() {   // training_env.cc:14-53
  double lastLoss = 0.f;
  double lossSum = 0.0f;
  while (!reader_.finished()) {
    JPPA_RETURN_IF_ERROR(readOneBatch());
    if (batch_.empty()) break;
    for (uint32_t it = 0; it < args_.batchMaxIterations; ++it) {
      JPPA_RETURN_IF_ERROR(trainOneBatch((int32_t)it));
      const double normLoss = std::abs(lastLoss - batchLoss_) / (double)batch_.size();
      lastLoss = batchLoss_;
      if (normLoss < args_.batchLossEpsilon) break;
    }
    lossSum += lastLoss;
  }
  if (firstEpoch_) {
    scw_->subtractInitValues();
    firstEpoch_ = false;
    if (jppgpu_ctx_set_weights(ctx_, scw_->weights().data(), (uint64_t)scw_->weights().size()) != JPPGPU_OK) return abiError("jppgpu_ctx_set_weights");
  }
  totalLoss_ = lossSum;
  return Status::Ok();
}

Status trainModel(TrainingEnv* env, const TrainingArguments& args) {
  const float lastLoss = 0.0f;
  for (uint32_t e = 0; e < args.maxEpochs; ++e) {
    env->resetInput();
    JPPA_RETURN_IF_ERROR(env->changeGlobalBeam(static_cast<float>(e) / args.maxEpochs));
    JPPA_RETURN_IF_ERROR(env->trainOneEpoch());
    if (std::abs((double)lastLoss - env->epochLoss()) < args.batchLossEpsilon) break;
  }
  return Status::Ok();
}

}  // namespace train
}  // namespace jumanpp_amd
