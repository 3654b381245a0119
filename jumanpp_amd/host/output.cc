#include "output.h"

namespace jumanpp_amd {

namespace {
// StringStorageReader::readAt (src/core/dic/field_reader.h:26-31)
StringPiece readStringAt(StringPiece storage, uint32_t alignPower, int32_t ptr) {
  size_t real = (size_t)ptr << alignPower;
  VarintReader r(storage, real);
  StringPiece out;
  if (r.readString(&out)) return out;
  return StringPiece("----READ_ERROR!!!----");
}
}  // namespace

bool NodeWalker::next() {
  if (!aliasRows_) {  // data is stored in features fully
    remaining_ -= 1;
    return remaining_ >= 0;
  }
  if (remaining_ <= 0) return false;
  for (int32_t i = 0; i < numData_; ++i) {
    uint64_t v;
    if (!rest_.read(&v)) return false;
    data_[i] = (int32_t)v;
  }
  remaining_ -= 1;
  return true;
}

StringPiece StringField::operator[](const NodeWalker& w) const {
  int32_t value = w.valueOf(index_);
  if (value < 0) return w.unkSurface();
  return readStringAt(storage_, alignPower_, value);
}

KVListIterator::KVListIterator(StringPiece strings, uint32_t alignPower, StringPiece ints, int32_t ptr)
    : strings_(strings), alignPower_(alignPower), rdr_(ints, (size_t)ptr) {
  uint64_t n;
  length_ = rdr_.read(&n) ? (int32_t)n : -1;
}

StringPiece KVListIterator::readAt(int32_t ptr) const { return readStringAt(strings_, alignPower_, ptr); }

bool KVListIterator::next() {
  if (!hasNext()) return false;
  uint64_t d;
  if (!rdr_.read(&d)) return false;
  position_ += 1;
  key_ = lastKey_ + (int32_t)(d >> 1);
  lastKey_ = key_;
  hasValue_ = (d & 1) != 0;
  if (!hasValue_) return true;
  if (!rdr_.read(&d)) return false;
  value_ = (int32_t)d;
  return true;
}

Status OutputManager::stringField(StringPiece name, StringField* result) const {
  auto fld = model_->fieldByName(name);
  if (fld == nullptr) return Status::InvalidParameter() << "dictionary field with name " << name << " was not found";
  if (fld->columnType != FieldType::String) return Status::InvalidParameter() << "field " << name << " was not string typed";
  result->index_ = fld->idxInEntry;
  result->storage_ = model_->stringStorage(fld->stringStorage);
  result->alignPower_ = fld->alignPower;
  return Status::Ok();
}

Status OutputManager::kvListField(StringPiece name, KVListField* result) const {
  auto fld = model_->fieldByName(name);
  if (fld == nullptr) return Status::InvalidParameter() << "dictionary field with name " << name << " was not found";
  if (fld->columnType != FieldType::StringKVList) return Status::InvalidParameter() << "field " << name << " was not kvlist";
  result->index_ = fld->idxInEntry;
  result->strings_ = model_->stringStorage(fld->stringStorage);
  result->ints_ = model_->intStorage(fld->intStorage);
  result->alignPower_ = fld->alignPower;
  return Status::Ok();
}

bool OutputManager::locate(const SentenceResult& s, uint32_t k, NodeWalker* w) const {
  if (k >= s.numNodes || s.nodes == nullptr) return false;
  return locate(s, s.nodes[k], s.unk[k], w);
}

bool OutputManager::locate(const SentenceResult& s, const jppgpu_node& nd, const jppgpu_unk& unk, NodeWalker* w) const {
  const int32_t nf = model_->numFeatures(), ndata = model_->numData();
  if (nf + ndata + 1 > kMaxDicFields) return false;
  w->numFeatures_ = nf;
  w->numData_ = ndata;
  w->eptr_ = nd.entry_ptr;
  w->special_ = false;
  w->aliasRows_ = false;
  w->remaining_ = 0;
  w->unkSurface_ = StringPiece();
  if (nd.entry_ptr == JPPGPU_ENTRY_BOS || nd.entry_ptr == JPPGPU_ENTRY_EOS) {  // fillFeaturesWithValue(ptr), fillDataWithValue(0)
    for (int i = 0; i < nf; ++i) w->features_[i] = nd.entry_ptr;
    for (int i = 0; i < ndata; ++i) w->data_[i] = 0;
    w->special_ = true;
    w->remaining_ = 1;
    return true;
  }
  int32_t actual = nd.entry_ptr;
  if (nd.entry_ptr < 0) {  // UNK: template row, surface-bearing features replaced
    w->special_ = true;
    actual = unk.template_ptr;
    w->unkSurface_ = s.surface(nd);
    w->placeholders_[0] = unk.placeholder[0];
    w->placeholders_[1] = unk.placeholder[1];
  }
  // DicEntryBuffer::fillFromStorage (dic_entries.h:102-127)
  const bool alias = (actual & 1) != 0;
  VarintReader r(model_->entryData(), (size_t)((uint32_t)actual >> 1));
  uint64_t v;
  for (int i = 0; i < nf; ++i) {
    if (!r.read(&v)) return false;
    w->features_[i] = (int32_t)v;
  }
  if (alias) {
    if (!r.read(&v)) return false;
    w->aliasRows_ = true;
    w->remaining_ = (int32_t)v;
    w->rest_ = r;
  } else {
    for (int i = 0; i < ndata; ++i) {
      if (!r.read(&v)) return false;
      w->data_[i] = (int32_t)v;
    }
    w->remaining_ = 1;
  }
  if (nd.entry_ptr < 0) {
    // ExtraNodesContext node content: the maker's replaced fields carry the (negative) surface hash
    const uint32_t mask = model_->unkMaker(unk.maker).replace_mask;
    for (int i = 0; i < nf; ++i)
      if ((mask >> i) & 1) w->features_[i] = unk.content_hash;
  }
  return true;
}

}  // namespace jumanpp_amd
