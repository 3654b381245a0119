// Loss of one analysed training example: the host counterpart of core::training::LossCalculator
// (src/core/training/loss.{h,cc}) over the index-form lattice of jppgpu_result_view -- beam slots name their previous
// node by index, so "pointer to the beam element of the top-1 path" becomes (node, slot).
#ifndef JUMANPP_AMD_HOST_TRAIN_LOSS_H
#define JUMANPP_AMD_HOST_TRAIN_LOSS_H

#include <vector>

#include "../model_image.h"
#include "gold_nodes.h"
#include "jppgpu.h"

namespace jumanpp_amd {
namespace train {

enum class TrainingMode { Full, FalloffBeam, MaxViolation };   // training_types.h:15-24

// ScoredFeature (trainer_base.h): one entry of the sparse update vector
struct ScoredFeature {
  uint32_t feature;
  float score;
};

enum class CompareClass { Both, TopOnly, GoldOnly };

struct ComparisonStep {   // loss.h:19-65
  CompareClass cls = CompareClass::Both;
  float lastGoldScore = 0, violation = 0, mismatchWeight = 0;
  int32_t numMismatches = 0;
  int32_t boundary = 0;
  int32_t topPath = -1;      // position on the top-1 path (0 = EOS, like jppgpu_top1_ngrams_view), -1: none
  float topScore = 0;
  int32_t goldPosition = -1;
  int32_t numGold = -1;
  bool goldInBeam = false;
  bool hasError() const { return cls != CompareClass::Both || violation > 0.001f || numMismatches > 0; }
};

// what the loss reads of one sentence
struct SentenceLattice {
  const jppgpu_result_view* view = nullptr;
  uint32_t s = 0;
  int32_t numFeatures = 0;
  const jppgpu_node* nodes() const { return view->nodes + view->node_base[s]; }
  // the device writes rows of 8 or 16 columns whatever the model's numFeatures (jppgpu_result_view::entry_row_stride)
  uint64_t rowStride() const { return view->entry_row_stride > 0 ? (uint64_t)view->entry_row_stride : 8u; }
  const int32_t* row(uint32_t node) const { return view->entry_rows + (view->node_base[s] + node) * rowStride(); }
  const jppgpu_beam_slot* beam(uint32_t node) const { return view->beams + (view->node_base[s] + node) * (uint64_t)view->beam; }
  uint32_t bndFirst(uint32_t b) const { return view->bnd_first[view->bnd_base[s] + b]; }
  uint32_t numNodes() const { return view->n_nodes[s]; }
  uint32_t eosBoundary() const { return view->n_codepoints[s] + 2; }
};

class LossCalculator {
  const TrainingSpecInfo* spec_ = nullptr;
  float fullWeight_ = 0;
  std::vector<ComparisonStep> comparison_;
  std::vector<float> goldNodeScores_, goldScores_;
  std::vector<uint32_t> top1Features_, goldFeatures_;
  std::vector<ScoredFeature> scored_;
  // top-1 path in text order: node, its slot on the path, the slot's total score
  struct PathItem {
    uint32_t node, slot;
    float total;
    int32_t pathPos;   // position in the EOS-first path arrays
  };
  std::vector<PathItem> top_;
  const uint32_t* goldNgrams_ = nullptr;
  const uint32_t* topNgrams_ = nullptr;
  uint32_t numNgram_ = 0;
  std::vector<uint32_t> goldNodes_;

  bool goldStillInBeam(const SentenceLattice& L, uint32_t goldNode, int32_t goldIdx) const;
  void mergeOne(uint32_t target, float score);

 public:
  void initialize(const TrainingSpecInfo* spec);
  // LossCalculator::computeGoldScores: every row of gold n-gram features summed like
  // HashedFeaturePerceptron::compute (perceptron.cc: computeUnrolled4RawPerceptron per row), then the running sum
  void computeGoldScores(const float* weights, uint32_t mask, const uint32_t* goldNgrams, uint32_t numNgram, size_t rows);
  // resolveTop1 + computeComparison.  gold: one position per word; topNgrams: the sentence's rows of
  // jppgpu_result_fetch_top1_ngrams (EOS first)
  Status compare(const SentenceLattice& L, const std::vector<GoldPosition>& gold, const uint32_t* topNgrams, size_t topRows);
  int32_t fullSize() const { return (int32_t)comparison_.size(); }
  int32_t fallOffBeam() const;
  int32_t maxViolation() const;
  float computeLoss(int32_t till);
  void computeFeatureDiff(uint32_t mask);
  const std::vector<ScoredFeature>& featureDiff() const { return scored_; }
  const std::vector<ComparisonStep>& comparison() const { return comparison_; }
};

}  // namespace train
}  // namespace jumanpp_amd

#endif  // JUMANPP_AMD_HOST_TRAIN_LOSS_H
