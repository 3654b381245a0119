// Dictionary-side view of a lattice node for the output formats: the host
// counterpart of core::analysis::OutputManager / NodeWalker / StringField /
// KVListField (src/core/analysis/output.{h,cc}) and of DicEntryBuffer
// (src/core/dic/dic_entries.h:18-160).  Reads the model's own varint blobs.
#ifndef JUMANPP_AMD_HOST_OUTPUT_H
#define JUMANPP_AMD_HOST_OUTPUT_H

#include <cstdint>

#include "gpu_analyzer.h"
#include "model_image.h"

namespace jumanpp_amd {

constexpr int kMaxDicFields = 32;  // JPP_MAX_DIC_FIELDS

// util::CodedBufferParser::readVarint64 (src/util/coded_io.h:130-137)
struct VarintReader {
  const unsigned char* p = nullptr;
  const unsigned char* end = nullptr;
  VarintReader() = default;
  VarintReader(StringPiece s, size_t from)
      : p((const unsigned char*)s.data() + (from < s.size() ? from : s.size())), end((const unsigned char*)s.data() + s.size()) {}
  bool read(uint64_t* out) {
    uint64_t v = 0;
    int shift = 0;
    while (p < end) {
      unsigned char b = *p++;
      v |= (uint64_t)(b & 0x7f) << shift;
      if (!(b & 0x80)) {
        *out = v;
        return true;
      }
      shift += 7;
      if (shift > 63) return false;
    }
    return false;
  }
  bool readString(StringPiece* out) {
    uint64_t n;
    if (!read(&n) || p + n > end) return false;
    *out = StringPiece((const char*)p, (size_t)n);
    p += n;
    return true;
  }
};

// NodeWalker: one lattice node = one dictionary entry row, or several for an alias entry
class NodeWalker {
  friend class OutputManager;
  int32_t features_[kMaxDicFields];
  int32_t data_[kMaxDicFields];
  int32_t numFeatures_ = 0, numData_ = 0;
  int32_t eptr_ = 0;
  bool special_ = false;      // UNK node (EntryPtr::isSpecial)
  int32_t remaining_ = 0;     // rows still to be produced by next()
  bool aliasRows_ = false;    // data rows come from the alias list
  VarintReader rest_;
  StringPiece unkSurface_;
  uint16_t placeholders_[2] = {0, 0};

 public:
  // DicEntryBuffer::nextData
  bool next();
  int32_t eptr() const { return eptr_; }
  bool isSpecial() const { return special_; }
  const int32_t* features() const { return features_; }
  // NodeWalker::valueOf: idx >= 0 feature column, < 0 data column ~idx
  int32_t valueOf(int32_t fieldIdx) const { return fieldIdx >= 0 ? features_[fieldIdx] : data_[~fieldIdx]; }
  StringPiece unkSurface() const { return unkSurface_; }
  int32_t placeholder(int i) const { return placeholders_[i]; }
};

class StringField {
  friend class OutputManager;
  int32_t index_ = 0;
  StringPiece storage_;
  uint32_t alignPower_ = 0;

 public:
  // StringField::operator[] (output.cc:112-130): negative value = surface of the UNK node
  StringPiece operator[](const NodeWalker& w) const;
  int32_t index() const { return index_; }   // >= 0: feature column of the entry row, < 0: ~data column
};

class KVListIterator {
  StringPiece strings_;
  uint32_t alignPower_ = 0;
  VarintReader rdr_;
  int32_t length_ = 0, position_ = 0;
  int32_t lastKey_ = 0, key_ = 0, value_ = 0;
  bool hasValue_ = false;
  StringPiece readAt(int32_t ptr) const;

 public:
  KVListIterator(StringPiece strings, uint32_t alignPower, StringPiece ints, int32_t ptr);
  bool hasNext() const { return position_ < length_; }
  bool next();  // KeyValueListTraversal::moveNext (field_reader.h:131-148)
  StringPiece key() const { return readAt(key_); }
  bool hasValue() const { return hasValue_; }
  StringPiece value() const { return readAt(value_); }
};

class KVListField {
  friend class OutputManager;
  int32_t index_ = 0;
  StringPiece strings_, ints_;
  uint32_t alignPower_ = 0;

 public:
  KVListIterator operator[](const NodeWalker& w) const {
    int32_t ptr = w.valueOf(index_);
    if (ptr == -1) ptr = 0;
    return KVListIterator(strings_, alignPower_, ints_, ptr);
  }
};

class OutputManager {
  const ModelImage* model_ = nullptr;

 public:
  explicit OutputManager(const ModelImage* m) : model_(m) {}
  Status stringField(StringPiece name, StringField* result) const;
  Status kvListField(StringPiece name, KVListField* result) const;
  // OutputManager::locate (output.cc:65-105) for node `k` of a sentence
  bool locate(const SentenceResult& s, uint32_t k, NodeWalker* result) const;
  // the same for a node given by its records (s: input text and codepoint offsets for UNK surfaces)
  bool locate(const SentenceResult& s, const jppgpu_node& nd, const jppgpu_unk& unk, NodeWalker* result) const;
};

}  // namespace jumanpp_amd

#endif  // JUMANPP_AMD_HOST_OUTPUT_H
