#include "scw_update.h"

#include <cmath>
#include <random>

namespace jumanpp_amd {
namespace train {

SoftConfidenceWeighted::SoftConfidenceWeighted(const ScwConfig& cfg, uint32_t exponent, uint32_t seed)
    : phi_(cfg.phi), C_(cfg.C), zeta_(1 + phi_ * phi_), psi_(1 + phi_ * phi_ / 2), exponent_(exponent) {
  const size_t n = size_t{1} << exponent;
  diagonal_.assign(n, 1.0f);
  weights_.reserve(n);
  std::default_random_engine eng{seed};
  const float boundary = (float)(1.0 / std::sqrt((double)(uint32_t)n));
  std::uniform_real_distribution<float> dist{-boundary, boundary};
  for (size_t i = 0; i < n; ++i) weights_.push_back(dist(eng));
}

void SoftConfidenceWeighted::update(float loss, const std::vector<ScoredFeature>& features) {
  if (loss < 1e-5) return;
  // (the two tables are 2^k floats each and the features of an example are scattered over them: ask for every line
  // before the four passes below walk them in order -- same arithmetic, the cache misses overlap instead of queueing)
  for (const auto& v : features) {
    __builtin_prefetch(&weights_[v.feature], 1, 1);
    __builtin_prefetch(&diagonal_[v.feature], 1, 1);
  }
  // calcScore / calcVt (scw.cc:47-63): double accumulators over float products
  double score = 0, vt = 0;
  for (const auto& v : features) score += weights_[v.feature] * v.score;
  for (const auto& v : features) vt += v.score * v.score * diagonal_[v.feature];
  // calcAlpha (scw.cc:75-84)
  const double mt = loss * score;
  double alpha = (1.0f / (vt * zeta_)) *
                 (-mt * psi_ + std::sqrt((mt * mt) * (phi_ * phi_ * phi_ * phi_ / 4.0f) + vt * phi_ * phi_ * zeta_));
  if (alpha < 0.0) alpha = 0.0;
  if (alpha > C_) alpha = C_;
  // calcUt, calcBeta (scw.cc:65-73)
  const double t = (-alpha * vt * phi_ + std::sqrt(alpha * alpha * vt * vt * phi_ * phi_ + 4 * vt));
  const double ut = (1.0 / 4.0) * t * t;
  const double beta = (alpha * phi_) / (std::sqrt(ut) + (vt * alpha * phi_));
  if (vt == 0) return;
  const float a = (float)alpha, b = (float)beta;
  for (const auto& v : features) {   // updateWeights (scw.cc:38-45)
    const float upd = a * loss * diagonal_[v.feature] * v.score;
    weights_[v.feature] += upd;
  }
  for (const auto& f : features) {   // updateMatrix (scw.cc:86-95)
    const float cur = diagonal_[f.feature];
    const float upd = cur * cur * f.score * f.score;
    diagonal_[f.feature] -= b * upd;
  }
}

uint64_t SoftConfidenceWeighted::subtractInitValues() {
  const double boundary = 1.01 / std::sqrt((double)weights_.size());
  uint64_t zeroed = 0;
  for (auto& w : weights_) {
    if (std::fabs(w) < boundary) {
      w = 0;
      zeroed += 1;
    }
  }
  return zeroed;
}

}  // namespace train
}  // namespace jumanpp_amd
