// Model file -> jppgpu_model + the dictionary field storages the output formats read.
// Accepts the reference's own `.jppmdl` container (jppmdl_reader.cc) and the flat model image
// (JPPGPUI1) that `oracle/_ref/ref_dump export` writes for the tests.  Plays the role of JumanppEnv::loadModel /
// CoreHolder for the host layer (src/core/env.cc:28-121, src/core/core.cc:11-40):
// one immutable, shared object that must outlive every GpuAnalyzer.
//
// The image is written by `oracle/_ref/ref_dump export` from a .jppmdl; its
// sections are the model's own blobs, verbatim (see jumanpp_amd/native.py for the
// Python twin of this reader).
#ifndef JUMANPP_AMD_HOST_MODEL_IMAGE_H
#define JUMANPP_AMD_HOST_MODEL_IMAGE_H

#include <cstdint>
#include <string>
#include <unordered_map>
#include <utility>
#include <vector>

#include "jpp_status.h"
#include "jppgpu.h"

namespace jumanpp_amd {

// spec::FieldType (src/core/spec/spec_types.h:18)
enum class FieldType : int32_t { String = 0, Int = 1, StringList = 2, StringKVList = 3, Error = 4 };

// dic::DictionaryField (src/core/dic/dictionary.h:19-30)
struct DictionaryField {
  int32_t idxInEntry = 0;      // >= 0: feature column, < 0: ~data column
  int32_t specIndex = 0;
  FieldType columnType = FieldType::Error;
  int32_t stringStorage = -1;
  int32_t intStorage = -1;
  uint32_t alignPower = 0;
  bool isTrieKey = false;
  std::string name;
  std::string emptyValue;
};

// spec::TrainingField as TrainFieldsIndex needs it (src/core/input/training_io.cc:37-55)
struct TrainField {
  std::string name;   // dictionary field name
  int32_t dicIdx = 0; // entry-row column
};

// spec::TrainingSpec (src/core/spec/spec_types.h:181-198), what the trainer reads
struct TrainingFieldSpec {
  int32_t number = 0;    // column in the training example
  int32_t fieldIdx = 0;  // index of the field in the dictionary spec
  int32_t dicIdx = 0;    // entry-row column
  float weight = 0;
  std::string name;      // dictionary field name
};
struct AllowedUnkFieldSpec {
  int32_t targetField = 0, sourceField = 0;   // spec indices
  std::string targetName, sourceName, sourceKey;
  int32_t sourceDicIndex = 0;                 // FieldDescriptor::dicIndex of the source field (< 0: data column)
};
struct TrainingSpecInfo {
  int32_t surfaceIdx = 0;
  std::vector<TrainingFieldSpec> fields;
  std::vector<AllowedUnkFieldSpec> allowedUnk;
};

struct RnnScoreWeights {
  float perceptron = 1.0f;
  float rnn = 0.0f;
};

// analysis::rnn::RnnInferenceConfig as command-line / config-file flags give it (rnn_arg_parse.h):
// value + "was it given" for the five numeric parameters
struct RnnConfigOverride {
  float nceBias = -9.0f, unkConstantTerm = -6.0f, unkLengthPenalty = -1.5f, perceptronWeight = 1.0f, rnnWeight = 1.0f;
  bool hasNceBias = false, hasUnkConstantTerm = false, hasUnkLengthPenalty = false, hasPerceptronWeight = false,
       hasRnnWeight = false;
  bool isDefault() const {
    return !(hasNceBias || hasUnkConstantTerm || hasUnkLengthPenalty || hasPerceptronWeight || hasRnnWeight);
  }
};

// the bytes of a model file: mapped, not read (a 0.7 GB model was zero-filled and copied into a vector, ~0.3 s of
// every process start).  8 readable zero bytes follow the file (varint readers may look ahead): the mapping is made over
// an anonymous reservation one page longer than the file.
class FileBytes {
  char* p_ = nullptr;
  size_t map_ = 0, size_ = 0;

 public:
  FileBytes() = default;
  FileBytes(const FileBytes&) = delete;
  FileBytes& operator=(const FileBytes&) = delete;
  ~FileBytes() { reset(); }
  void reset();
  bool open(const std::string& path);
  const char* data() const { return p_; }
  size_t size() const { return size_; }
};

class ModelImage {
  FileBytes data_;
  jppgpu_model model_{};
  std::vector<jppgpu_unk_maker> makers_;
  std::vector<DictionaryField> fields_;
  std::vector<StringPiece> stringStorages_;
  std::vector<StringPiece> intStorages_;
  int32_t numFeatures_ = 0, numData_ = 0, numPlaceholders_ = 0;
  bool hasRnn_ = false;
  RnnScoreWeights rnnWeights_;
  // the RNN part's saved RnnInferenceConfig as read (jppmdl only)
  bool hasSavedRnnConfig_ = false;
  RnnConfigOverride savedRnnConfig_;
  std::unordered_map<uint64_t, uint64_t> posMap_, conjMap_;
  bool hasIdMap_ = false;
  std::vector<TrainField> trainFields_;
  TrainingSpecInfo trainingSpec_;
  // the container's parts as loaded (jppmdl only): what saveWithPerceptron re-emits
  struct RawPart {
    int32_t kind = 0;
    std::string comment;
    std::vector<std::pair<size_t, size_t>> blocks;   // (offset, size) in data_
  };
  std::vector<RawPart> rawParts_;
  bool allowUntrained_ = false;
  std::vector<char> ownedFeatureSpec_;
  size_t fileSize_ = 0;
  Status loadImage(const std::string& fn);
  Status loadJppmdl(const std::string& fn);  // jppmdl_reader.cc

 public:
  ModelImage() = default;
  ModelImage(const ModelImage&) = delete;
  ModelImage& operator=(const ModelImage&) = delete;

  Status loadModel(StringPiece filename);
  // the trainer starts from a model without a perceptron part (jpp_jumandic_bootstrap output): weights stay null
  Status loadModelForTraining(StringPiece filename) {
    allowUntrained_ = true;
    return loadModel(filename);
  }
  const TrainingSpecInfo& trainingSpec() const { return trainingSpec_; }
  // the trainer's output: the loaded .jppmdl with a perceptron part appended (jumanpp_train.cc doTrainJpp:
  // env.modelInfoCopy() + SoftConfidenceWeighted::exportModel, scw.cc:128-151; container: ModelSaver::save,
  // src/core/impl/model_io.cc:47-99).  Readable by the reference and by loadModel.
  Status saveWithPerceptron(const std::string& path, const float* weights, uint32_t exponent, const std::string& comment) const;

  const jppgpu_model& cmodel() const { return model_; }
  bool hasRnn() const { return hasRnn_; }
  // JumanppEnv::setRnnConfig + RnnScorerGbeamFactory::setConfig (src/core/env.cc:81-107,
  // rnn_scorer_gbeam.cc:339-346) for a model that carries an RNN: the given values override the ones saved
  // in the model, the NCE constant follows the merged nceBias, and the score weights become the
  // *override's* perceptron / RNN weights (1.0 where not given -- what the reference does).
  // A given RNN weight of 0 switches the RNN off (*useRnn = false).  Only for natively loaded .jppmdl.
  Status applyRnnConfig(const RnnConfigOverride& o, bool* useRnn, RnnScoreWeights* weights);
  // JumanppEnv::setRnnHolder for a scorer made from a separate RNN model file (jumandic_env.cc:44-48,
  // env.cc:65-79, rnn_scorer_gbeam.cc:312-346): `part` supplies the rnn_* members (see ExternalRnn), the
  // numeric parameters are the given ones or the RnnInferenceConfig defaults, the NCE constant is the
  // file's unless --rnn-nce-bias is given.  Replaces an embedded RNN.
  void attachExternalRnn(const jppgpu_model& part, const RnnConfigOverride& o, RnnScoreWeights* weights);
  // ScorerDef::scoreWeights saved with the model (RnnInferenceConfig, src/core/env.cc:86-100)
  RnnScoreWeights savedScoreWeights() const { return rnnWeights_; }
  int32_t numFeatures() const { return numFeatures_; }
  int32_t numData() const { return numData_; }
  int32_t numPlaceholders() const { return numPlaceholders_; }
  const jppgpu_unk_maker& unkMaker(size_t i) const { return makers_[i]; }
  size_t numUnkMakers() const { return makers_.size(); }

  // DictionaryHolder::fieldByName (src/core/dic/dictionary.h)
  const DictionaryField* fieldByName(StringPiece name) const;
  StringPiece stringStorage(int32_t idx) const { return stringStorages_[idx]; }
  size_t numStringStorages() const { return stringStorages_.size(); }
  const std::vector<TrainField>& trainFields() const { return trainFields_; }
  StringPiece intStorage(int32_t idx) const { return intStorages_[idx]; }
  StringPiece entryData() const { return StringPiece((const char*)model_.entry_data, model_.entry_data_bytes); }
  // the value storages of the feature columns (jppgpu_config::field_storages: what a spec's length primitives read):
  // string columns their string storage, string-list columns their int-list storage
  std::vector<jppgpu_field_storage> fieldStorages() const {
    std::vector<jppgpu_field_storage> out;
    for (const DictionaryField& f : fields_) {
      if (f.idxInEntry < 0) continue;
      jppgpu_field_storage st;
      st.column = f.idxInEntry;
      st.align_power = f.alignPower;
      st.reserved = 0;
      if (f.columnType == FieldType::String && f.stringStorage >= 0 && (size_t)f.stringStorage < stringStorages_.size()) {
        st.kind = 1;
        st.data = stringStorages_[f.stringStorage].data();
        st.bytes = stringStorages_[f.stringStorage].size();
      } else if (f.columnType == FieldType::StringList && f.intStorage >= 0 && (size_t)f.intStorage < intStorages_.size()) {
        st.kind = 2;
        st.data = intStorages_[f.intStorage].data();
        st.bytes = intStorages_[f.intStorage].size();
      } else {
        continue;
      }
      out.push_back(st);
    }
    return out;
  }

  // JumandicIdResolver::dicToJuman (src/jumandic/shared/jumandic_id_resolver.cc:80-86), resolved at export time
  bool hasIdMap() const { return hasIdMap_; }
  void dicToJuman(int32_t pos, int32_t subpos, int32_t conjType, int32_t conjForm, int32_t out[4]) const;
};

}  // namespace jumanpp_amd

#endif  // JUMANPP_AMD_HOST_MODEL_IMAGE_H
