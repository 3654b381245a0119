// An RNN language model given as a separate faster-rnnlm file (`--rnn-model=PATH`: vocabulary in PATH,
// weights in PATH.nnet) instead of one embedded in the .jppmdl: RnnScorerGbeamFactory::make
// (src/core/analysis/rnn_scorer_gbeam.cc:312-334) = MikolovModelReader (src/rnn/mikolov_rnn.cc:16-76,
// 131-215) + RnnIdResolver::build (src/core/analysis/rnn_id_resolver.cc:20-155).
//
// The reference indexes the RNN vocabulary with two darts-clone double arrays (words made of dictionary
// strings only / words with out-of-dictionary parts); the device walks arrays of that unit format, so
// DoubleArrayBuilder below produces them.  It is a plain first-fit builder (no suffix sharing): the arrays
// differ from the reference's byte for byte but answer every lookup identically.
#ifndef JUMANPP_AMD_HOST_RNN_EXTERNAL_H
#define JUMANPP_AMD_HOST_RNN_EXTERNAL_H

#include <cstdint>
#include <string>
#include <utility>
#include <vector>

#include "model_image.h"

namespace jumanpp_amd {

class DoubleArrayBuilder {
  std::vector<std::pair<std::string, int32_t>> keys_;

 public:
  void add(std::string key, int32_t value) { keys_.emplace_back(std::move(key), value); }
  // units in the darts-clone layout: label in bits 0-7 (and bit 31 for value units), bit 8 = the node has
  // a value, bits 10-30 = offset to the children (<< 8 when bit 9 is set)
  Status build(std::vector<uint32_t>* units);
  // exact-match lookup, the same steps the device takes (DoubleArray::traversal().step == Ok)
  static bool find(const std::vector<uint32_t>& units, const std::string& key, int32_t* value);
};

// analysis::rnn::RnnInferenceConfig, the string part (rnn_arg_parse.h: --rnn-fields, --rnn-separator,
// --rnn-unk, --rnn-eos)
struct ExternalRnnConfig {
  std::vector<std::string> fields;
  std::string separator = "_";
  std::string unkSymbol = "<unk>";
  std::string eosSymbol = "</s>";
};

class ExternalRnn {
  std::vector<float> embeddings_, nceEmbeddings_, matrix_, maxent_;
  std::vector<uint32_t> known_, unk_;
  jppgpu_model part_{};  // only the rnn_* members are meaningful

 public:
  // reads the model and builds the word-id indices against `dictionary`'s string storages
  Status load(const std::string& path, const ModelImage& dictionary, const ExternalRnnConfig& cfg);
  // rnn_* members for jppgpu_model (pointers into this object: keep it alive while contexts are created)
  const jppgpu_model& part() const { return part_; }
  float nceLnz() const { return part_.rnn_nce_constant; }
};

}  // namespace jumanpp_amd

#endif  // JUMANPP_AMD_HOST_RNN_EXTERNAL_H
