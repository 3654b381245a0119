// The training loop above the C ABI: the host counterpart of core::training::TrainingEnv / TrainerBatch / Trainer /
// TrainingExecutor (src/core/training/training_env.{h,cc}, trainer.{h,cc}, training_executor.{h,cc}) with the
// reference's worker threads replaced by the device: a batch of examples is ONE jppgpu_analyze_batch_seeds call (gold
// seeds resolved in its hook = Trainer::prepare), the lattice, top-1 n-grams and gold-path n-grams are fetched
// (= Trainer::compute's inputs), then loss, feature difference and the SCW update run per example in batch order on the
// host, and the updated table goes back with jppgpu_ctx_set_weights.
//
// Batch semantics.  The reference's executor threads analyse the examples of a batch while the main thread applies the
// updates of the ones already finished, so with more than one example in flight the weights an example sees depend on
// thread timing.  Here every example of a batch is analysed with the weights of the batch's start; with --batch 1 (the
// reference's default) both are the same sequential algorithm and the resulting model is bit-identical
// (tests/test_train_parity.py).
#ifndef JUMANPP_AMD_HOST_TRAIN_ENV_H
#define JUMANPP_AMD_HOST_TRAIN_ENV_H

#include <memory>
#include <string>
#include <vector>

#include "../model_image.h"
#include "gold_nodes.h"
#include "jppgpu.h"
#include "scw_update.h"
#include "train_example.h"
#include "train_loss.h"

namespace jumanpp_amd {
namespace train {

// GlobalBeamParams (training_env.h:21-47)
struct GlobalBeamParams {
  int32_t minLeftBeam = -1, maxLeftBeam = -1;
  int32_t minRightBeam = -1, maxRightBeam = -1;
  int32_t minRightCheck = -1, maxRightCheck = -1;
  bool fullFirstIter = false;   // --gb-first-full: full-beam scoring on the first iteration over every batch
  bool leftEnabled() const { return minLeftBeam > 0; }
  bool rightEnabled() const { return minRightBeam > 0; }
  Status validate() const;
};

// TrainingArguments (training_env.h:49-66) as jumanpp_v2_train's flags fill them (jumanpp_train.cc:17-166)
struct TrainingArguments {
  std::string modelFilename, outputFilename, corpusFilename, comment;
  uint32_t sizeExponent = 15;
  uint32_t randomSeed = 0xdeadbeefU;
  TrainingMode mode = TrainingMode::Full;
  CorpusFormat inputFormat = CorpusFormat::Morph;
  ScwConfig scw;
  int32_t beamSize = 5;
  uint32_t batchSize = 1;
  uint32_t batchMaxIterations = 1;
  uint32_t maxEpochs = 1;
  float batchLossEpsilon = 1e-3f;
  GlobalBeamParams globalBeam;
  int32_t device = 0;
};

class TrainingEnv {
  TrainingArguments args_;
  const ModelImage* model_ = nullptr;
  TrainFieldsIndex tio_;
  GoldNodeResolver resolver_;
  GoldExampleReader reader_;
  std::unique_ptr<SoftConfidenceWeighted> scw_;
  jppgpu_ctx* ctx_ = nullptr;
  std::string corpus_;
  std::vector<GoldExample> batch_;       // TrainerBatch::trainers_
  std::vector<int32_t> order_;           // TrainerBatch::indices_
  uint32_t numShuffles_ = 0;
  bool firstEpoch_ = true;
  double batchLoss_ = 0, totalLoss_ = 0;
  int32_t leftBeam_ = 0, rightBeam_ = 0, rightCheck_ = 0;
  uint64_t examplesSeen_ = 0, goldNodesAdded_ = 0;
  double stageMs_[6] = {0, 0, 0, 0, 0, 0};   // analyse (incl. the seed hook), fetch, n-gram read-outs, loss, SCW, weights upload

  // hook state of the batch in flight
  std::vector<std::vector<GoldPosition>> goldPaths_;
  std::vector<uint32_t> extraOffsets_;
  std::vector<jppgpu_extra_seed> extraSeeds_;
  Status hookStatus_;
  static int seedHook(void* user, const jppgpu_seed_view* view, jppgpu_extra_seeds* out);

  Status readOneBatch();
  Status trainOneBatch(int32_t iter);

 public:
  TrainingEnv() = default;
  TrainingEnv(const TrainingEnv&) = delete;
  TrainingEnv& operator=(const TrainingEnv&) = delete;
  ~TrainingEnv();
  // lib: the C ABI entry points come from the library the process is linked with
  Status initialize(const TrainingArguments& args, const ModelImage* model);
  Status loadInput(const std::string& filename);
  void resetInput() {
    reader_.setInput(StringPiece(corpus_), args_.inputFormat);
    batchLoss_ = totalLoss_ = 0;
  }
  // TrainingEnv::changeGlobalBeam (training_env.cc:207-233)
  Status changeGlobalBeam(float ratio);
  Status trainOneEpoch();
  double epochLoss() const { return totalLoss_; }
  const SoftConfidenceWeighted& scw() const { return *scw_; }
  uint64_t examplesSeen() const { return examplesSeen_; }
  uint64_t goldNodesAdded() const { return goldNodesAdded_; }
  int32_t leftBeam() const { return leftBeam_; }
  const double* stageMs() const { return stageMs_; }
};

// doTrain (jumanpp_train.cc:168-198)
Status trainModel(TrainingEnv* env, const TrainingArguments& args);

}  // namespace train
}  // namespace jumanpp_amd

#endif  // JUMANPP_AMD_HOST_TRAIN_ENV_H
