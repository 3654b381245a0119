// Gold nodes of a training example against the seeds of its sentence: the host half of
// TrainingExampleAdapter / UnkAllowedField / GoldenPath (src/core/training/gold_example.{h,cc}).  Runs inside the seed
// hook of jppgpu_analyze_batch_seeds: for every word of the example either the position of a matching seed among the
// seeds starting at the word's first codepoint, or a new seed for the device to insert behind them.
#ifndef JUMANPP_AMD_HOST_TRAIN_GOLD_NODES_H
#define JUMANPP_AMD_HOST_TRAIN_GOLD_NODES_H

#include <unordered_map>
#include <vector>

#include "../model_image.h"
#include "jppgpu.h"
#include "train_example.h"

namespace jumanpp_amd {
namespace train {

// LatticeNodePtr of every gold word: boundary = first codepoint + 2, position among the nodes starting there
struct GoldPosition {
  uint16_t boundary = 0;
  uint16_t position = 0;
};

// UnkAllowedField (gold_example.cc:145-215): an UNK node of the lattice may stand for a gold word when the value its
// template carries under `sourceKey` of the key-value field equals the gold value of the target field
class UnkAllowedFields {
  struct Info {
    std::unordered_map<int32_t, int32_t> templateToGold;  // template EntryPtr raw -> string pointer in the target field
    int32_t goldColumn = 0;
  };
  std::vector<Info> fields_;

 public:
  Status initialize(const ModelImage& model);
  bool isAllowed(int32_t templatePtr, const GoldWord& w) const;
};

class GoldNodeResolver {
  const ModelImage* model_ = nullptr;
  const TrainingSpecInfo* spec_ = nullptr;
  UnkAllowedFields allowed_;
  int32_t surfaceColumn_ = 0;
  std::vector<int32_t> row_;

  void dicRow(int32_t entryPtr);                                  // DictionaryEntries::entryAtPtr(..).fill
  void unkRow(const jppgpu_unk& unk);                             // ExtraNode::content of a maker's UNK node
  bool matchDic(const GoldWord& w) const;                         // matchDicNodeData
  bool matchUnk(const GoldWord& w, int32_t surfaceHash) const;    // matchUnkNodeData

 public:
  Status initialize(const ModelImage& model);
  int32_t surfaceColumn() const { return surfaceColumn_; }
  // TrainingExampleAdapter::ensureNodes for sentence `s` of the view: fills path (one entry per word) and appends the
  // seeds to add.  `text` is the sentence as given to the analyser.
  Status resolve(const GoldExample& ex, const jppgpu_seed_view& view, uint32_t s, std::vector<GoldPosition>* path,
                 std::vector<jppgpu_extra_seed>* extra) const;
};

}  // namespace train
}  // namespace jumanpp_amd

#endif  // JUMANPP_AMD_HOST_TRAIN_GOLD_NODES_H
