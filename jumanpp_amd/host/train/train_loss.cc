#include "train_loss.h"

#include <algorithm>

namespace jumanpp_amd {
namespace train {

void LossCalculator::initialize(const TrainingSpecInfo* spec) {
  spec_ = spec;
  fullWeight_ = 0;
  for (const auto& f : spec->fields) fullWeight_ += f.weight;
}

void LossCalculator::computeGoldScores(const float* weights, uint32_t mask, const uint32_t* goldNgrams, uint32_t numNgram, size_t rows) {
  goldNgrams_ = goldNgrams;
  numNgram_ = numNgram;
  goldNodeScores_.assign(rows, 0.f);
  goldScores_.assign(rows, 0.f);
  for (size_t r = 0; r < rows; ++r) {
    const uint32_t* f = goldNgrams + r * numNgram;
    float r1 = 0, r2 = 0, r3 = 0, r4 = 0;
    uint32_t i = 0;
    for (; i + 4 <= numNgram; i += 4) {
      r1 += weights[f[i] & mask];
      r2 += weights[f[i + 1] & mask];
      r3 += weights[f[i + 2] & mask];
      r4 += weights[f[i + 3] & mask];
    }
    const uint32_t rest = numNgram - i;
    if (rest >= 3) r3 += weights[f[i + 2] & mask];
    if (rest >= 2) r2 += weights[f[i + 1] & mask];
    if (rest >= 1) r1 += weights[f[i] & mask];
    goldNodeScores_[r] = r1 + r2 + r3 + r4;
  }
  float acc = 0;   // std::partial_sum over floats
  for (size_t r = 0; r < rows; ++r) {
    acc = r == 0 ? goldNodeScores_[0] : acc + goldNodeScores_[r];
    goldScores_[r] = acc;
  }
}

// LossCalculator::isGoldStillInBeam (loss.cc:171-193): does a live slot of the gold node's beam continue the previous
// gold node?  A BOS slot (no previous) ends the search.
bool LossCalculator::goldStillInBeam(const SentenceLattice& L, uint32_t goldNode, int32_t goldIdx) const {
  if (goldIdx == 0) return true;
  const uint32_t prevGold = goldNodes_[(size_t)goldIdx - 1];
  const jppgpu_beam_slot* bm = L.beam(goldNode);
  for (int32_t q = 0; q < L.view->beam; ++q) {
    if (bm[q].left == 0xffff && bm[q].beam == 0xffff) break;
    if (bm[q].prev_node == 0xffffffffu) return false;
    if (bm[q].prev_node == prevGold) return true;
  }
  return false;
}

Status LossCalculator::compare(const SentenceLattice& L, const std::vector<GoldPosition>& gold, const uint32_t* topNgrams, size_t topRows) {
  topNgrams_ = topNgrams;
  comparison_.clear();
  const uint32_t N = L.numNodes();
  const uint32_t eosNode = N - 1;
  const uint32_t eosBnd = L.eosBoundary();
  // --- AnalysisPath::fillIn (analysis_result.cc:25-76): from the best EOS slot back to the first word ---
  top_.clear();
  {
    uint32_t node = eosNode, slot = 0;
    int32_t pos = 0;
    for (;;) {
      const jppgpu_beam_slot& bs = L.beam(node)[slot];
      const uint32_t prev = bs.prev_node;
      if (prev == 0xffffffffu || prev < 2) break;
      if (prev >= N || bs.beam >= (uint32_t)L.view->beam) return Status::InvalidState() << "broken top-1 path";
      pos += 1;
      top_.push_back(PathItem{prev, bs.beam, L.beam(prev)[bs.beam].total, pos});
      node = prev;
      slot = bs.beam;
      if (top_.size() > N) return Status::InvalidState() << "cycle on the top-1 path";
    }
    std::reverse(top_.begin(), top_.end());
  }
  if (top_.size() + 1 != topRows) return Status::InvalidState() << "top-1 path and its n-gram rows disagree";
  // --- gold nodes ---
  goldNodes_.clear();
  for (const auto& g : gold) {
    const uint32_t node = L.bndFirst(g.boundary) + g.position;
    if (node >= N) return Status::InvalidState() << "gold node outside the lattice";
    goldNodes_.push_back(node);
  }
  if (goldScores_.size() != gold.size() + 1) return Status::InvalidState() << "gold scores are not computed";
  // --- LossCalculator::computeComparison (loss.cc:68-169) ---
  float goldScore = 0;
  size_t curTop = 0, curGold = 0;
  const size_t totalGold = gold.size();
  auto boundaryOf = [&](uint32_t node) { return (int32_t)L.nodes()[node].start + 2; };
  while (curTop < top_.size() || curGold < totalGold) {
    const int32_t goldBnd = curGold < totalGold ? (int32_t)gold[curGold].boundary : (int32_t)eosBnd;
    const int32_t topBnd = curTop < top_.size() ? boundaryOf(top_[curTop].node) : (int32_t)eosBnd;
    if (goldBnd == topBnd) {
      // findWorstTopNode (loss.cc:19-66) over the one element the path has at this boundary
      if (curTop >= top_.size() || curGold >= totalGold) return Status::InvalidState() << "could not find a worst top node for position #" << topBnd;
      const PathItem& t = top_[curTop];
      const uint32_t gnode = goldNodes_[curGold];
      const int32_t* topRow = L.row(t.node);
      const int32_t* goldRow = L.row(gnode);
      ComparisonStep st;
      st.cls = CompareClass::Both;
      int32_t mism = 0;
      float w = 0;
      for (const auto& sf : spec_->fields) {
        // (spec index used as the row column, like the reference: loss.cc:43)
        if (sf.fieldIdx < 0 || sf.fieldIdx >= L.numFeatures) continue;
        if (topRow[sf.fieldIdx] != goldRow[sf.fieldIdx]) {
          mism += 1;
          w += sf.weight;
        }
      }
      st.boundary = topBnd;
      st.topPath = t.pathPos;
      st.topScore = t.total;
      st.goldPosition = gold[curGold].position;
      st.numMismatches = mism;
      st.mismatchWeight = w;
      goldScore = goldScores_[curGold];
      st.lastGoldScore = goldScore;
      st.violation = t.total - goldScore;
      st.goldInBeam = goldStillInBeam(L, gnode, (int32_t)curGold);
      st.numGold = (int32_t)curGold;
      comparison_.push_back(st);
      ++curTop;
      ++curGold;
      continue;
    }
    if (goldBnd > topBnd) {
      const PathItem& t = top_[curTop];
      ComparisonStep st;
      st.cls = CompareClass::TopOnly;
      st.lastGoldScore = goldScore;
      st.boundary = topBnd;
      st.topPath = t.pathPos;
      st.topScore = t.total;
      st.violation = t.total - goldScore;
      comparison_.push_back(st);
      ++curTop;
    } else {
      const uint32_t gnode = goldNodes_[curGold];
      goldScore = goldScores_[curGold];
      ComparisonStep st;
      st.cls = CompareClass::GoldOnly;
      st.goldPosition = gold[curGold].position;
      st.boundary = goldBnd;
      st.lastGoldScore = goldScore;
      st.goldInBeam = goldStillInBeam(L, gnode, (int32_t)curGold);
      st.numGold = (int32_t)curGold;
      comparison_.push_back(st);
      ++curGold;
    }
  }
  {
    const jppgpu_beam_slot& eos = L.beam(eosNode)[0];
    ComparisonStep st;
    st.cls = CompareClass::Both;
    goldScore = goldScores_[curGold];
    st.boundary = (int32_t)eosBnd;
    st.topPath = 0;
    st.topScore = eos.total;
    st.goldPosition = 0;
    st.lastGoldScore = goldScore;
    st.violation = eos.total - goldScore;
    st.goldInBeam = goldStillInBeam(L, eosNode, (int32_t)curGold);
    st.numGold = (int32_t)curGold;
    comparison_.push_back(st);
  }
  return Status::Ok();
}

int32_t LossCalculator::fallOffBeam() const {
  const int32_t sz = fullSize();
  for (int32_t i = 0; i < sz; ++i) {
    const auto& c = comparison_[(size_t)i];
    if (c.cls != CompareClass::TopOnly && !c.goldInBeam) return std::min(i + 2, sz - 1);
  }
  return sz;
}

int32_t LossCalculator::maxViolation() const {
  const int32_t sz = fullSize();
  int32_t val = 0;
  float viol = 0;
  for (int32_t i = 0; i < sz; ++i) {
    const auto& c = comparison_[(size_t)i];
    if (c.cls != CompareClass::GoldOnly && c.violation > viol) {
      val = i;
      viol = c.violation;
    }
  }
  return std::min(val + 2, sz - 1);
}

// LossCalculator::computeLoss (loss.cc:302-333) with addGoldNgrams / addTopNgrams (without JPP_TRAIN_MID_NGRAMS)
float LossCalculator::computeLoss(int32_t till) {
  goldFeatures_.clear();
  top1Features_.clear();
  float loss = 0;
  const int32_t size = fullSize();
  auto addGold = [&](int32_t numGold) {
    const uint32_t* row = goldNgrams_ + (size_t)numGold * numNgram_;
    goldFeatures_.insert(goldFeatures_.end(), row, row + numNgram_);
  };
  auto addTop = [&](int32_t pathPos) {
    const uint32_t* row = topNgrams_ + (size_t)pathPos * numNgram_;
    top1Features_.insert(top1Features_.end(), row, row + numNgram_);
  };
  for (int32_t i = 0; i < size; ++i) {
    const auto& c = comparison_[(size_t)i];
    if (c.cls == CompareClass::GoldOnly) {
      loss += fullWeight_;
      if (i < till) addGold(c.numGold);
    } else if (c.cls == CompareClass::TopOnly) {
      loss += fullWeight_;
      if (i < till) addTop(c.topPath);
    } else if (c.hasError()) {
      loss += c.mismatchWeight;
      if (i < till) {
        addGold(c.numGold);
        addTop(c.topPath);
      }
    }
  }
  return loss / (size * fullWeight_);
}

void LossCalculator::mergeOne(uint32_t target, float score) {
  if (scored_.back().feature != target) scored_.push_back(ScoredFeature{target, 0});
  scored_.back().score += score;
}

// LossCalculator::computeFeatureDiff (loss.cc:195-280)
void LossCalculator::computeFeatureDiff(uint32_t mask) {
  size_t topPos = 0, goldPos = 0;
  scored_.clear();
  scored_.push_back(ScoredFeature{0, 0});
  for (auto& f : top1Features_) f &= mask;
  std::sort(top1Features_.begin(), top1Features_.end());
  for (auto& f : goldFeatures_) f &= mask;
  std::sort(goldFeatures_.begin(), goldFeatures_.end());
  const size_t topCnt = top1Features_.size(), goldCnt = goldFeatures_.size();
  while (topPos < topCnt && goldPos < goldCnt) {
    const uint32_t g = goldFeatures_[goldPos], t = top1Features_[topPos];
    if (g == t) {
      ++goldPos;
      ++topPos;
    } else if (g < t) {
      mergeOne(g, 1.0f);
      ++goldPos;
    } else {
      mergeOne(t, -1.0f);
      ++topPos;
    }
  }
  // (the two tails carry the signs the reference gives them)
  for (; topPos < topCnt; ++topPos) mergeOne(top1Features_[topPos], 1.0f);
  for (; goldPos < goldCnt; ++goldPos) mergeOne(goldFeatures_[goldPos], -1.0f);
  float numGold = 0, numTop = 0;
  for (const auto& s : scored_) {
    if (s.score > 0) numGold += s.score;
    else numTop -= s.score;
  }
  float weight = 1.0f;
  bool updateTop = true;
  if (numTop != 0) weight = numGold / numTop;
  if (weight > 2 && numGold != 0) {
    weight = numTop / numGold;
    updateTop = false;
  }
  for (auto& s : scored_) {
    if (updateTop ? s.score < 0 : s.score > 0) s.score *= weight;
  }
}

}  // namespace train
}  // namespace jumanpp_amd
