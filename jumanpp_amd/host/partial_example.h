// Partially annotated input (`--partial-input`): the host counterpart of
// core::input::PartialExample / PartialExampleReader / TrainFieldsIndex / PexStreamReader
// (src/core/input/partial_example.{h,cc}, partial_example_io.cc:24-143,
// training_io.cc:37-55, pex_stream_reader.cc:41-69).  The constraints are applied on
// the device by k_penalty + the plugin hooks of k_sweep (jppgpu_analyze_batch_partial).
#ifndef JUMANPP_AMD_HOST_PARTIAL_EXAMPLE_H
#define JUMANPP_AMD_HOST_PARTIAL_EXAMPLE_H

#include <istream>
#include <string>
#include <unordered_map>
#include <vector>

#include "jpp_status.h"
#include "jppgpu.h"
#include "model_image.h"

namespace jumanpp_amd {

struct TagConstraint {
  int32_t field;
  int32_t value;
};

struct NodeConstraint {
  int32_t boundary = 0;
  int32_t length = 0;
  std::string surface;
  std::vector<TagConstraint> tags;
};

struct PartialExample {
  std::string comment;
  std::string surface;
  std::vector<int32_t> boundaries;
  std::vector<int32_t> noBreak;
  std::vector<NodeConstraint> nodes;
};

// TrainFieldsIndex: the training fields of the spec with their string -> pointer maps
class TrainFieldsIndex {
 public:
  struct Field {
    std::string name;
    int32_t dicFieldIdx;
    const std::unordered_map<std::string, int32_t>* str2int;
  };

 private:
  std::vector<std::unordered_map<std::string, int32_t>> storages_;
  std::vector<Field> fields_;

 public:
  Status initialize(const ModelImage& model);
  const std::vector<Field>& fields() const { return fields_; }
  const Field* byName(StringPiece name) const;
};

// analysis::hashUnkString (src/core/analysis/unk_nodes_creator.cc:170-177)
int32_t hashUnkString(StringPiece sp);

class PartialExampleReader {
  const TrainFieldsIndex* tio_ = nullptr;
  char32_t noBreakToken_ = U'&';
  // the reference reuses one PartialExample and never clears its comment: a comment stays in force for
  // the following examples until another one replaces it (partial_example_io.cc:24-43)
  mutable std::string lastComment_;

 public:
  Status initialize(const TrainFieldsIndex* tio, char32_t noBreakToken = U'&') {
    tio_ = tio;
    noBreakToken_ = noBreakToken;
    return Status::Ok();
  }
  // PexStreamReader::readExample: the lines up to the next empty line form one example
  Status readExample(std::istream* stream, PartialExample* result) const;
  // PartialExampleReader::readExample on an in-memory block of lines
  Status parse(StringPiece data, PartialExample* result) const;
};

// flat CSR arrays for jppgpu_analyze_batch_partial
struct PartialBatch {
  std::vector<uint32_t> nobreakOff, boundaryOff, nodeOff;
  std::vector<uint16_t> nobreak, boundaries;
  std::vector<jppgpu_node_constraint> nodes;
  std::vector<jppgpu_tag_constraint> tags;
  jppgpu_partial view{};
  void build(const std::vector<const PartialExample*>& examples);
};

}  // namespace jumanpp_amd

#endif  // JUMANPP_AMD_HOST_PARTIAL_EXAMPLE_H
