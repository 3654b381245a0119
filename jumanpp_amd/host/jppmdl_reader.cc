// Native reader of the reference's model container (`.jppmdl`), so that the host layer does not
// depend on the reference-linked exporter.  Restates the read side of
//   model::FilesystemModel::open/load          src/core/impl/model_io.cc:115-176 (magic "jp2Mdl!", varint
//                                              header size, ModelInfoRaw; blocks are (offset, size) pairs)
//   util::serialization (varint ints, length-prefixed strings, fixed32 floats, counted vectors)
//                                              src/util/serialization.h
//   BuiltDictionary / AnalysisSpec Serialize   src/core/dic/dic_builder.cc:73-86, src/core/spec/spec_ser.h
//   fixupDictionary (block order)              src/core/dic/dic_builder.cc:118-183
//   PerceptronInfo                             src/core/impl/perceptron_io.h
//   RnnModelHeader                             src/core/analysis/rnn_scorer_gbeam.cc:353-398,426-470
//   JumandicIdResolver::initialize             src/jumandic/shared/jumandic_id_resolver.cc:32-78
// and produces what `oracle/ref_dump export` writes into a model image; tests/test_host_cli.py runs the same
// analyses from both and against the reference CLI.
#include <cstring>
#include <fstream>
#include <map>

#include "model_image.h"
#include "output.h"

namespace jumanpp_amd {

namespace {

#include "jumandic_ids.inc"

struct Loader {
  const unsigned char* p;
  const unsigned char* end;
  bool ok = true;
  Loader(StringPiece s) : p((const unsigned char*)s.data()), end(p + s.size()) {}
  uint64_t varint() {
    uint64_t v = 0;
    int shift = 0;
    while (p < end) {
      unsigned char b = *p++;
      v |= (uint64_t)(b & 0x7f) << shift;
      if (!(b & 0x80)) return v;
      shift += 7;
      if (shift > 63) break;
    }
    ok = false;
    return 0;
  }
  int32_t i32() { return (int32_t)(uint32_t)varint(); }
  bool boolean() { return i32() == 1; }
  float f32() {
    if (p + 4 > end) {
      ok = false;
      return 0.f;
    }
    float f;
    std::memcpy(&f, p, 4);
    p += 4;
    return f;
  }
  std::string str() {
    uint64_t n = varint();
    if (!ok || p + n > end) {
      ok = false;
      return std::string();
    }
    std::string s((const char*)p, (size_t)n);
    p += n;
    return s;
  }
  std::vector<int32_t> ints() {
    uint64_t n = varint();
    std::vector<int32_t> v;
    for (uint64_t i = 0; i < n && ok; ++i) v.push_back(i32());
    return v;
  }
  std::vector<std::string> strs() {
    uint64_t n = varint();
    std::vector<std::string> v;
    for (uint64_t i = 0; i < n && ok; ++i) v.push_back(str());
    return v;
  }
  bool atEnd() const { return p == end; }
};

struct FieldDesc {  // spec::FieldDescriptor
  int32_t specIndex, position, dicIndex;
  std::string name;
  bool isTrieKey;
  int32_t fieldType;
  std::string emptyString, listSeparator, kvSeparator;
  int32_t stringStorage, intStorage, alignment;
};
struct UnkDesc {  // spec::UnkProcessorDescriptor
  int32_t type, patternPtr, priority, charClass;
  int32_t placeholder;  // features[0].targetPlaceholder or -1
  std::vector<int32_t> replaceFields;
};
struct TrainDesc {
  int32_t number, fieldIdx, dicIdx;
  float weight;
};
struct AllowedDesc {
  int32_t target, source;
  std::string key;
};
struct BuiltFieldDesc {
  int32_t dicIndex, specIndex, uniqueValues;
};

struct Bytes {
  std::vector<char> buf;
  void i32(int32_t v) {
    const char* c = reinterpret_cast<const char*>(&v);
    buf.insert(buf.end(), c, c + 4);
  }
  void ints(const std::vector<int32_t>& v) {
    i32((int32_t)v.size());
    for (auto x : v) i32(x);
  }
};

constexpr int32_t kInvalidInt = INT32_MIN;  // spec::InvalidInt

}  // namespace

Status ModelImage::loadJppmdl(const std::string& fn) {
  const char* base = data_.data();
  const size_t sz = fileSize_;
  // ---- container header ----
  Loader hl(StringPiece(base + 8, (sz < 4096 ? sz : 4096) - 8));
  uint64_t hdrSize = hl.varint();
  if (!hl.ok || hl.p + hdrSize > hl.end) return Status::InvalidState() << "could not read header size from " << fn;
  Loader l(StringPiece((const char*)hl.p, (size_t)hdrSize));
  struct Part {
    int32_t kind;
    std::vector<StringPiece> data;
  };
  std::vector<Part> parts;
  uint64_t nparts = l.varint();
  rawParts_.clear();
  for (uint64_t i = 0; i < nparts && l.ok; ++i) {
    Part pt;
    RawPart rp;
    pt.kind = l.i32();
    rp.kind = pt.kind;
    rp.comment = l.str();
    uint64_t nb = l.varint();
    for (uint64_t b = 0; b < nb && l.ok; ++b) {
      uint64_t off = l.varint(), size = l.varint();
      if (off + size > sz) return Status::InvalidState() << "model file " << fn << " has a block outside the file";
      pt.data.push_back(StringPiece(base + off, (size_t)size));
      rp.blocks.emplace_back((size_t)off, (size_t)size);
    }
    (void)l.varint();  // start
    (void)l.varint();  // end
    parts.push_back(std::move(pt));
    rawParts_.push_back(std::move(rp));
  }
  if (!l.ok || !l.atEnd()) return Status::InvalidState() << "model file " << fn << " has corrupted model header";
  auto firstPartOf = [&](int32_t kind) -> const Part* {
    for (auto& p : parts)
      if (p.kind == kind) return &p;
    return nullptr;
  };
  // ModelPartKind: Dictionary 0, Perceprton 1, Rnn 2, ScwDump 3
  const Part* dic = firstPartOf(0);
  if (dic == nullptr) return Status::InvalidParameter("there was no dictionary information in saved model");
  if (dic->data.size() < 2) {
    return Status::InvalidParameter("dictionary info must have at least two fragments, probably corrupted model file");
  }

  // ---- BuiltDictionary: entryCount, fieldData, timestamp, spec ----
  Loader d(dic->data[0]);
  const int32_t entryCount = d.i32();
  std::vector<BuiltFieldDesc> built;
  for (uint64_t i = 0, n = d.varint(); i < n && d.ok; ++i) {
    BuiltFieldDesc b;
    b.dicIndex = d.i32();
    b.specIndex = d.i32();
    b.uniqueValues = d.i32();
    built.push_back(b);
  }
  (void)d.varint();  // timestamp (i64)
  const uint32_t magic1 = (uint32_t)d.varint();
  const uint32_t version = (uint32_t)d.varint();
  // DictionarySpec
  std::vector<FieldDesc> fdesc;
  for (uint64_t i = 0, n = d.varint(); i < n && d.ok; ++i) {
    FieldDesc f;
    f.specIndex = d.i32();
    f.position = d.i32();
    f.dicIndex = d.i32();
    f.name = d.str();
    f.isTrieKey = d.boolean();
    f.fieldType = d.i32();
    f.emptyString = d.str();
    f.listSeparator = d.str();
    f.kvSeparator = d.str();
    f.stringStorage = d.i32();
    f.intStorage = d.i32();
    f.alignment = d.i32();
    fdesc.push_back(std::move(f));
  }
  (void)d.ints();  // aliasingSet
  (void)d.i32();   // indexColumn
  const int32_t numIntStorage = d.i32();
  const int32_t numStringStorage = d.i32();
  // FeaturesSpec
  for (uint64_t i = 0, n = d.varint(); i < n && d.ok; ++i) {  // dictionary imports
    (void)d.i32(); (void)d.i32(); (void)d.i32(); (void)d.str(); (void)d.i32(); (void)d.ints();
  }
  Bytes feat;
  {
    uint64_t n = d.varint();
    feat.i32((int32_t)n);
    for (uint64_t i = 0; i < n && d.ok; ++i) {  // primitive: index, name, kind, references, matchData
      (void)d.i32();
      (void)d.str();
      int32_t kind = d.i32();
      std::vector<int32_t> refs = d.ints();
      (void)d.strs();
      feat.i32(kind);
      feat.ints(refs);
    }
    n = d.varint();
    feat.i32((int32_t)n);
    for (uint64_t i = 0; i < n && d.ok; ++i) {  // computation: name, index, primitiveFeature, true, false
      (void)d.str();
      (void)d.i32();
      int32_t prim = d.i32();
      std::vector<int32_t> t = d.ints(), f = d.ints();
      feat.i32(prim);
      feat.ints(t);
      feat.ints(f);
    }
  }
  int32_t numPatterns = 0;
  {
    uint64_t n = d.varint();
    numPatterns = (int32_t)n;
    feat.i32((int32_t)n);
    for (uint64_t i = 0; i < n && d.ok; ++i) {  // pattern: index, usage, references
      int32_t idx = d.i32();
      (void)d.i32();
      std::vector<int32_t> refs = d.ints();
      feat.i32(idx);
      feat.ints(refs);
    }
    n = d.varint();
    feat.i32((int32_t)n);
    for (uint64_t i = 0; i < n && d.ok; ++i) {  // ngram: index, references
      int32_t idx = d.i32();
      std::vector<int32_t> refs = d.ints();
      feat.i32(idx);
      feat.ints(refs);
    }
  }
  numPlaceholders_ = d.i32();
  (void)d.i32();  // totalPrimitives
  numFeatures_ = d.i32();
  numData_ = d.i32();
  const int32_t numUniOnlyPats = d.i32();
  (void)numUniOnlyPats;
  (void)numPatterns;
  (void)entryCount;
  // unk creators
  std::vector<UnkDesc> unks;
  for (uint64_t i = 0, n = d.varint(); i < n && d.ok; ++i) {
    UnkDesc u;
    (void)d.i32();  // index
    (void)d.str();  // name
    u.type = d.i32();
    (void)d.i32();  // patternRow
    u.patternPtr = d.i32();
    u.priority = d.i32();
    u.charClass = d.i32();
    u.placeholder = -1;
    for (uint64_t q = 0, nf = d.varint(); q < nf && d.ok; ++q) {
      int32_t target = d.i32();
      (void)d.i32();  // featureType
      if (q == 0) u.placeholder = target;
    }
    u.replaceFields = d.ints();
    unks.push_back(std::move(u));
  }
  // training spec
  const int32_t trainSurfaceIdx = d.i32();
  std::vector<TrainDesc> trains;
  for (uint64_t i = 0, n = d.varint(); i < n && d.ok; ++i) {
    TrainDesc t;
    t.number = d.i32();
    t.fieldIdx = d.i32();
    t.dicIdx = d.i32();
    t.weight = d.f32();
    trains.push_back(t);
  }
  std::vector<AllowedDesc> allowed;
  for (uint64_t i = 0, n = d.varint(); i < n && d.ok; ++i) {  // allowedUnk
    AllowedDesc a;
    a.target = d.i32();
    a.source = d.i32();
    a.key = d.str();
    allowed.push_back(std::move(a));
  }
  const uint32_t magic2 = (uint32_t)d.varint();
  if (!d.ok || !d.atEnd()) return Status::InvalidParameter("failed to load dictionary metadata from model file");
  if (magic1 != 0xfeed0000u || magic2 != magic1 || version != 3) {
    return Status::InvalidParameter("dictionary spec of the model has an unsupported version or is corrupted");
  }

  // ---- fixupDictionary: block order ----
  if ((size_t)(numStringStorage + numIntStorage + 4) != dic->data.size()) {
    return Status::InvalidParameter("model file did not have all dictionary chunks");
  }
  if (built.size() != fdesc.size()) {
    return Status::InvalidParameter("number of columns in spec was not equal to loaded number of columns");
  }
  jppgpu_model& m = model_;
  std::memset(&m, 0, sizeof(m));
  m.trie = dic->data[1].data();
  m.trie_bytes = dic->data[1].size();
  m.entry_ptrs = dic->data[2].data();
  m.entry_ptrs_bytes = dic->data[2].size();
  m.entry_data = dic->data[3].data();
  m.entry_data_bytes = dic->data[3].size();
  stringStorages_.clear();
  intStorages_.clear();
  for (int i = 0; i < numStringStorage; ++i) stringStorages_.push_back(dic->data[4 + i]);
  for (int i = 0; i < numIntStorage; ++i) intStorages_.push_back(dic->data[4 + numStringStorage + i]);
  m.num_features = numFeatures_;
  m.num_placeholders = numPlaceholders_;
  ownedFeatureSpec_.swap(feat.buf);
  m.feature_spec = ownedFeatureSpec_.data();
  m.feature_spec_bytes = ownedFeatureSpec_.size();
  makers_.clear();
  for (auto& u : unks) {
    jppgpu_unk_maker k{};
    k.type = u.type;
    k.char_class = u.charClass;
    k.pattern_ptr = u.patternPtr;
    k.priority = u.priority;
    k.placeholder = u.placeholder;
    for (auto f : u.replaceFields) k.replace_mask |= 1u << f;
    makers_.push_back(k);
  }
  m.unk_makers = makers_.data();
  m.num_unk_makers = (int32_t)makers_.size();
  fields_.clear();
  for (auto& b : built) {
    if (b.specIndex < 0 || (size_t)b.specIndex >= fdesc.size()) return Status::InvalidParameter("bad field index in the model");
    const FieldDesc& sf = fdesc[b.specIndex];
    if (b.dicIndex != sf.dicIndex) {
      return Status::InvalidParameter() << "something went wrong and built field dicIndex !=  spec field dicIndex: "
                                        << b.dicIndex << " vs " << sf.dicIndex;
    }
    DictionaryField f;
    f.idxInEntry = b.dicIndex;
    f.specIndex = b.specIndex;
    f.columnType = (FieldType)sf.fieldType;
    f.stringStorage = sf.stringStorage == kInvalidInt ? -1 : sf.stringStorage;
    f.intStorage = sf.intStorage == kInvalidInt ? -1 : sf.intStorage;
    f.alignPower = (uint32_t)sf.alignment;
    f.isTrieKey = sf.isTrieKey;
    f.name = sf.name;
    f.emptyValue = sf.emptyString;
    fields_.push_back(std::move(f));
  }
  trainFields_.clear();
  for (auto& t : trains) {
    if (t.fieldIdx < 0 || (size_t)t.fieldIdx >= fdesc.size()) return Status::InvalidParameter("bad training field in the model");
    trainFields_.push_back(TrainField{fdesc[t.fieldIdx].name, t.dicIdx});
  }
  trainingSpec_ = TrainingSpecInfo();
  trainingSpec_.surfaceIdx = trainSurfaceIdx;
  for (auto& t : trains) {
    TrainingFieldSpec f;
    f.number = t.number;
    f.fieldIdx = t.fieldIdx;
    f.dicIdx = t.dicIdx;
    f.weight = t.weight;
    f.name = fdesc[t.fieldIdx].name;
    trainingSpec_.fields.push_back(std::move(f));
  }
  for (auto& a : allowed) {
    if (a.target < 0 || (size_t)a.target >= fdesc.size() || a.source < 0 || (size_t)a.source >= fdesc.size())
      return Status::InvalidParameter("bad allowed-unk field in the model");
    AllowedUnkFieldSpec f;
    f.targetField = a.target;
    f.sourceField = a.source;
    f.targetName = fdesc[a.target].name;
    f.sourceName = fdesc[a.source].name;
    f.sourceKey = a.key;
    f.sourceDicIndex = fdesc[a.source].dicIndex;
    trainingSpec_.allowedUnk.push_back(std::move(f));
  }

  // ---- perceptron ----
  const Part* perc = firstPartOf(1);
  if ((perc == nullptr || perc->data.size() < 2) && !allowUntrained_)
    return Status::InvalidParameter("model image has no perceptron weights (untrained model)");
  if (perc != nullptr && perc->data.size() >= 2) {
    Loader pl(perc->data[0]);
    int32_t exponent = pl.i32();
    if (!pl.ok || exponent < 0 || exponent > 40 || perc->data[1].size() != ((size_t)4 << exponent)) {
      return Status::InvalidParameter("bad perceptron header");
    }
    m.weights = reinterpret_cast<const float*>(perc->data[1].data());
    m.weight_exponent = (uint32_t)exponent;
  }

  // ---- JUMAN ids: JumandicIdResolver::initialize ----
  {
    auto readFieldToMap = [&](const char* name, std::map<std::string, int32_t>* out) -> bool {
      const DictionaryField* fld = fieldByName(name);
      if (fld == nullptr || fld->stringStorage < 0) return false;
      StringPiece data = stringStorages_[fld->stringStorage];
      const uint32_t align = 1u << fld->alignPower;
      VarintReader rdr(data, 0);
      const unsigned char* b = (const unsigned char*)data.data();
      while (rdr.p < rdr.end) {
        int32_t pos = (int32_t)((size_t)(rdr.p - b) >> fld->alignPower);
        StringPiece sp;
        if (!rdr.readString(&sp)) break;
        (*out)[sp.str()] = pos;
        size_t off = ((size_t)(rdr.p - b) + align - 1) & ~(size_t)(align - 1);
        rdr.p = b + (off < data.size() ? off : data.size());
      }
      return true;
    };
    std::map<std::string, int32_t> pos2id, sub2id, cf2id, ct2id;
    posMap_.clear();
    conjMap_.clear();
    hasIdMap_ = readFieldToMap("pos", &pos2id) && readFieldToMap("subpos", &sub2id) && readFieldToMap("conjform", &cf2id) &&
                readFieldToMap("conjtype", &ct2id);
    if (hasIdMap_) {
      const unsigned char* t = kJumandicIds;
      uint32_t nPos, nConj;
      std::memcpy(&nPos, t, 4);
      std::memcpy(&nConj, t + 4, 4);
      t += 8;
      auto findOr = [](const std::map<std::string, int32_t>& mp, const std::string& k) {
        auto it = mp.find(k);
        return it == mp.end() ? 0 : it->second;
      };
      auto key2 = [](int32_t a, int32_t b) { return ((uint64_t)(uint32_t)a << 32) | (uint32_t)b; };
      for (uint32_t i = 0; i < nPos + nConj; ++i) {
        std::string p1((const char*)t + 1, t[0]);
        t += 1 + t[0];
        std::string p2((const char*)t + 1, t[0]);
        t += 1 + t[0];
        uint16_t id1, id2;
        std::memcpy(&id1, t, 2);
        std::memcpy(&id2, t + 2, 2);
        t += 4;
        if (i < nPos) {
          int32_t a = findOr(pos2id, p1), b = findOr(sub2id, p2);
          if (a != 0 || b != 0) posMap_[key2(a, b)] = key2(id1, id2);
        } else {
          int32_t a = findOr(ct2id, p1), b = findOr(cf2id, p2);
          if (a != 0) {
            conjMap_[key2(a, b)] = key2(id1, id2);
            conjMap_[key2(a, 0)] = key2(id1, 0);
          }
        }
      }
    }
  }

  // ---- RNN ----
  hasRnn_ = false;
  if (const Part* rp = firstPartOf(2)) {
    if (rp->data.size() < 7) return Status::InvalidParameter("failed to read RNN header");
    Loader r(rp->data[0]);
    auto cfgFloat = [&](bool* defined) {
      uint64_t flag = r.varint();
      float v = r.f32();
      if (defined) *defined = flag != 0;
      return v;
    };
    RnnConfigOverride& sc = savedRnnConfig_;
    sc.nceBias = cfgFloat(&sc.hasNceBias);
    sc.unkConstantTerm = cfgFloat(&sc.hasUnkConstantTerm);
    sc.unkLengthPenalty = cfgFloat(&sc.hasUnkLengthPenalty);
    sc.perceptronWeight = cfgFloat(&sc.hasPerceptronWeight);
    sc.rnnWeight = cfgFloat(&sc.hasRnnWeight);
    hasSavedRnnConfig_ = true;
    m.rnn_unk_constant = sc.unkConstantTerm;
    m.rnn_unk_length = sc.unkLengthPenalty;
    rnnWeights_.perceptron = sc.perceptronWeight;
    rnnWeights_.rnn = sc.rnnWeight;
    const bool rnnWeightDefined = sc.hasRnnWeight;
    (void)r.varint(); (void)r.str();   // eosSymbol
    (void)r.varint(); (void)r.str();   // unkSymbol
    (void)r.varint(); (void)r.strs();  // rnnFields
    (void)r.varint(); (void)r.str();   // fieldSeparator
    m.rnn_unk_id = r.i32();
    std::vector<int32_t> flds = r.ints();
    m.rnn_layer_size = (uint32_t)r.varint();
    m.rnn_maxent_order = (uint32_t)r.varint();
    m.rnn_maxent_size = r.varint();
    m.rnn_vocab_size = r.varint();
    const float nceLnz = r.f32();
    if (!r.ok || !r.atEnd() || flds.size() > 8) return Status::InvalidParameter("failed to read RNN header");
    // RnnScorerGbeamFactory::load: the NCE constant is nceLnz, replaced by rnnWeight when that is defined
    m.rnn_nce_constant = rnnWeightDefined ? rnnWeights_.rnn : nceLnz;
    m.rnn_num_fields = (uint32_t)flds.size();
    for (size_t i = 0; i < flds.size(); ++i) m.rnn_fields[i] = (uint32_t)flds[i];
    m.rnn_known_index = rp->data[1].data();
    m.rnn_known_index_bytes = rp->data[1].size();
    m.rnn_unk_index = rp->data[2].data();
    m.rnn_unk_index_bytes = rp->data[2].size();
    m.rnn_matrix = reinterpret_cast<const float*>(rp->data[3].data());
    m.rnn_embeddings = reinterpret_cast<const float*>(rp->data[4].data());
    m.rnn_nce_embeddings = reinterpret_cast<const float*>(rp->data[5].data());
    m.rnn_maxent = reinterpret_cast<const float*>(rp->data[6].data());
    m.has_rnn = 1;
    hasRnn_ = true;
  }
  return Status::Ok();
}

void ModelImage::attachExternalRnn(const jppgpu_model& part, const RnnConfigOverride& o, RnnScoreWeights* weights) {
  model_.has_rnn = 1;
  model_.rnn_layer_size = part.rnn_layer_size;
  model_.rnn_maxent_order = part.rnn_maxent_order;
  model_.rnn_maxent_size = part.rnn_maxent_size;
  model_.rnn_vocab_size = part.rnn_vocab_size;
  model_.rnn_nce_constant = o.hasNceBias ? o.nceBias : part.rnn_nce_constant;
  model_.rnn_unk_id = part.rnn_unk_id;
  model_.rnn_unk_constant = o.unkConstantTerm;
  model_.rnn_unk_length = o.unkLengthPenalty;
  model_.rnn_num_fields = part.rnn_num_fields;
  for (uint32_t i = 0; i < part.rnn_num_fields && i < 8; ++i) model_.rnn_fields[i] = part.rnn_fields[i];
  model_.rnn_known_index = part.rnn_known_index;
  model_.rnn_known_index_bytes = part.rnn_known_index_bytes;
  model_.rnn_unk_index = part.rnn_unk_index;
  model_.rnn_unk_index_bytes = part.rnn_unk_index_bytes;
  model_.rnn_matrix = part.rnn_matrix;
  model_.rnn_embeddings = part.rnn_embeddings;
  model_.rnn_nce_embeddings = part.rnn_nce_embeddings;
  model_.rnn_maxent = part.rnn_maxent;
  hasRnn_ = true;
  hasSavedRnnConfig_ = false;
  rnnWeights_.perceptron = o.perceptronWeight;
  rnnWeights_.rnn = o.rnnWeight;
  *weights = rnnWeights_;
}

Status ModelImage::applyRnnConfig(const RnnConfigOverride& o, bool* useRnn, RnnScoreWeights* weights) {
  if (!hasRnn_) return Status::InvalidState("the model has no RNN part");
  if (!hasSavedRnnConfig_)
    return Status::NotImplemented("RNN parameter overrides need the model's saved RNN configuration: load the .jppmdl file");
  if (o.rnnWeight == 0.0f) {  // JumanppEnv::setRnnConfig: "disable rnn"
    *useRnn = false;
    weights->perceptron = o.perceptronWeight;
    weights->rnn = 0.0f;
    return Status::Ok();
  }
  // state_->config.mergeWith(config): a given value replaces the saved one
  RnnConfigOverride& sc = savedRnnConfig_;
  auto merge = [](float& v, bool& has, float ov, bool ohas) {
    if (ohas) {
      v = ov;
      has = true;
    }
  };
  merge(sc.nceBias, sc.hasNceBias, o.nceBias, o.hasNceBias);
  merge(sc.unkConstantTerm, sc.hasUnkConstantTerm, o.unkConstantTerm, o.hasUnkConstantTerm);
  merge(sc.unkLengthPenalty, sc.hasUnkLengthPenalty, o.unkLengthPenalty, o.hasUnkLengthPenalty);
  merge(sc.perceptronWeight, sc.hasPerceptronWeight, o.perceptronWeight, o.hasPerceptronWeight);
  merge(sc.rnnWeight, sc.hasRnnWeight, o.rnnWeight, o.hasRnnWeight);
  if (sc.hasNceBias) model_.rnn_nce_constant = sc.nceBias;  // setNceConstant(config.nceBias) when it is not default
  model_.rnn_unk_constant = sc.unkConstantTerm;
  model_.rnn_unk_length = sc.unkLengthPenalty;
  *useRnn = true;
  weights->perceptron = o.perceptronWeight;  // scoreWeights come from the override itself, not from the merge
  weights->rnn = o.rnnWeight;
  return Status::Ok();
}

namespace {
void putVarint(std::string* o, uint64_t v) {
  while (v >= 0x80) {
    o->push_back((char)(v | 0x80));
    v >>= 7;
  }
  o->push_back((char)v);
}
}  // namespace

Status ModelImage::saveWithPerceptron(const std::string& path, const float* weights, uint32_t exponent, const std::string& comment) const {
  if (rawParts_.empty()) return Status::InvalidState() << "only a natively loaded .jppmdl can be saved";
  struct OutPart {
    int32_t kind;
    std::string comment;
    std::vector<StringPiece> data;
  };
  std::vector<OutPart> parts;
  for (const auto& rp : rawParts_) {
    OutPart o{rp.kind, rp.comment, {}};
    for (const auto& b : rp.blocks) o.data.push_back(StringPiece(data_.data() + b.first, b.second));
    parts.push_back(std::move(o));
  }
  std::string percHeader;
  putVarint(&percHeader, exponent);   // PerceptronInfo::modelSizeExponent (perceptron_io.h)
  {
    OutPart o{1, comment, {}};
    o.data.push_back(StringPiece(percHeader));
    o.data.push_back(StringPiece(reinterpret_cast<const char*>(weights), (size_t)4 << exponent));
    parts.push_back(std::move(o));
  }
  // layout: every block on its own 4096-byte boundary, the header in the first page
  auto align4k = [](size_t v) { return (v + 4095) & ~(size_t)4095; };
  std::string hdr;
  putVarint(&hdr, parts.size());
  size_t offset = 4096;
  std::vector<std::pair<size_t, StringPiece>> placed;
  for (const auto& pt : parts) {
    putVarint(&hdr, (uint32_t)pt.kind);
    putVarint(&hdr, pt.comment.size());
    hdr += pt.comment;
    putVarint(&hdr, pt.data.size());
    const size_t start = offset;
    for (const auto& b : pt.data) {
      putVarint(&hdr, offset);
      putVarint(&hdr, b.size());
      placed.emplace_back(offset, b);
      offset = align4k(offset + b.size());
    }
    putVarint(&hdr, start);
    putVarint(&hdr, offset);
  }
  if (hdr.size() > 4080) return Status::NotImplemented() << "model header size >4080 bytes is not implemented";
  std::ofstream out(path, std::ios::binary | std::ios::trunc);
  if (!out) return Status::InvalidParameter() << "could not open " << path << " for writing";
  std::string page(4096, '\0');
  std::memcpy(&page[0], "jp2Mdl!", 8);
  std::string sz;
  putVarint(&sz, hdr.size());
  std::memcpy(&page[8], sz.data(), sz.size());
  std::memcpy(&page[8 + sz.size()], hdr.data(), hdr.size());
  out.write(page.data(), (std::streamsize)page.size());
  size_t at = 4096;
  const std::string zeros(4096, '\0');
  for (const auto& pb : placed) {
    if (pb.first > at) out.write(zeros.data(), (std::streamsize)(pb.first - at));
    out.write(pb.second.data(), (std::streamsize)pb.second.size());
    at = pb.first + pb.second.size();
  }
  if (offset > at) out.write(zeros.data(), (std::streamsize)(offset - at));
  out.flush();
  if (!out) return Status::InvalidState() << "failed to write " << path;
  return Status::Ok();
}

}  // namespace jumanpp_amd
