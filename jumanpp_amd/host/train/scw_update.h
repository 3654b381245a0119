// Soft confidence-weighted learning, the update step of the reference's trainer
// (core::training::SoftConfidenceWeighted, src/core/training/scw.{h,cc}): diagonal covariance, one update per example
// from its loss and sparse feature difference.  Runs on the host like the reference's; the weight table it maintains is
// what jppgpu_ctx_set_weights uploads between batches.
#ifndef JUMANPP_AMD_HOST_TRAIN_SCW_UPDATE_H
#define JUMANPP_AMD_HOST_TRAIN_SCW_UPDATE_H

#include <cstdint>
#include <vector>

#include "train_loss.h"

namespace jumanpp_amd {
namespace train {

struct ScwConfig {
  float C = 1.0f;
  float phi = 5.0f;
};

class SoftConfidenceWeighted {
  std::vector<float> weights_;
  std::vector<float> diagonal_;
  double phi_, C_, zeta_, psi_;
  uint32_t exponent_;

 public:
  // SoftConfidenceWeighted::SoftConfidenceWeighted (scw.cc:97-118): weights uniform in +-1/sqrt(n) from
  // std::default_random_engine{seed}, covariance diagonal 1
  SoftConfidenceWeighted(const ScwConfig& cfg, uint32_t exponent, uint32_t seed);
  void update(float loss, const std::vector<ScoredFeature>& features);
  // substractInitValues (scw.cc:214-226): |w| < 1.01/sqrt(n) -> 0; returns the number zeroed
  uint64_t subtractInitValues();
  const std::vector<float>& weights() const { return weights_; }
  uint32_t exponent() const { return exponent_; }
  uint32_t mask() const { return (uint32_t)(weights_.size() - 1); }
};

}  // namespace train
}  // namespace jumanpp_amd

#endif  // JUMANPP_AMD_HOST_TRAIN_SCW_UPDATE_H
