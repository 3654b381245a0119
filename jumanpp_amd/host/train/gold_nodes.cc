#include "gold_nodes.h"

#include <cstring>

#include "../output.h"

namespace jumanpp_amd {
namespace train {

Status UnkAllowedFields::initialize(const ModelImage& model) {
  fields_.clear();
  const TrainingSpecInfo& ts = model.trainingSpec();
  for (const auto& a : ts.allowedUnk) {
    const DictionaryField* src = model.fieldByName(a.sourceName);
    const DictionaryField* trg = model.fieldByName(a.targetName);
    if (src == nullptr || trg == nullptr) return Status::InvalidState() << "allowed-unk fields are not in the dictionary";
    if (a.sourceDicIndex >= 0) continue;   // "source field was in feature section": skipped by the reference too
    if (src->stringStorage < 0 || src->intStorage < 0 || trg->stringStorage < 0) continue;
    Info info;
    // the strings of the target field by content
    std::unordered_map<std::string, int32_t> targetStrings;
    {
      StringPiece data = model.stringStorage(trg->stringStorage);
      const uint32_t align = 1u << trg->alignPower;
      VarintReader rdr(data, 0);
      const unsigned char* base = (const unsigned char*)data.data();
      while (rdr.p < rdr.end) {
        const int32_t pos = (int32_t)((size_t)(rdr.p - base) >> trg->alignPower);
        StringPiece sp;
        if (!rdr.readString(&sp)) break;
        targetStrings[sp.str()] = pos;   // (the reference keeps the last position of equal strings as well)
        size_t off = ((size_t)(rdr.p - base) + align - 1) & ~(size_t)(align - 1);
        rdr.p = base + (off < data.size() ? off : data.size());
      }
    }
    const int32_t nf = model.numFeatures(), nd = model.numData();
    for (size_t q = 0; q < model.numUnkMakers(); ++q) {
      const int32_t ptr = model.unkMaker(q).pattern_ptr;
      // first data row of the template entry (DicEntryBuffer::fillBuffer + nextData)
      VarintReader r(model.entryData(), (size_t)((uint32_t)ptr >> 1));
      uint64_t v = 0;
      bool ok = true;
      for (int i = 0; i < nf && ok; ++i) ok = r.read(&v);
      if (ok && (ptr & 1)) ok = r.read(&v);   // alias entry: the row count precedes the data rows
      std::vector<int32_t> data((size_t)nd, 0);
      for (int i = 0; i < nd && ok; ++i) {
        ok = r.read(&v);
        data[(size_t)i] = (int32_t)v;
      }
      const int32_t col = ~a.sourceDicIndex;
      if (!ok || col < 0 || col >= nd) continue;
      KVListIterator kv(model.stringStorage(src->stringStorage), src->alignPower, model.intStorage(src->intStorage), data[(size_t)col]);
      while (kv.hasNext() && kv.next()) {
        StringPiece k = kv.key();
        if (k.size() != a.sourceKey.size() || std::memcmp(k.data(), a.sourceKey.data(), k.size()) != 0) continue;
        if (!kv.hasValue()) continue;
        auto it = targetStrings.find(kv.value().str());
        if (it != targetStrings.end()) info.templateToGold[ptr] = it->second;
      }
    }
    for (const auto& tf : ts.fields)
      if (tf.fieldIdx == a.targetField) info.goldColumn = tf.number;
    fields_.push_back(std::move(info));
  }
  return Status::Ok();
}

bool UnkAllowedFields::isAllowed(int32_t templatePtr, const GoldWord& w) const {
  for (const auto& f : fields_) {
    auto it = f.templateToGold.find(templatePtr);
    if (it == f.templateToGold.end()) return false;
    if (it->second != w.data[f.goldColumn]) return false;
  }
  return true;
}

Status GoldNodeResolver::initialize(const ModelImage& model) {
  model_ = &model;
  spec_ = &model.trainingSpec();
  if (spec_->fields.empty()) return Status::InvalidState() << "the model has no training spec";
  if (spec_->surfaceIdx < 0 || (size_t)spec_->surfaceIdx >= spec_->fields.size()) return Status::InvalidState() << "bad surface field in the training spec";
  surfaceColumn_ = spec_->fields[(size_t)spec_->surfaceIdx].number;
  row_.assign((size_t)model.numFeatures(), 0);
  return allowed_.initialize(model);
}

void GoldNodeResolver::dicRow(int32_t entryPtr) {
  VarintReader r(model_->entryData(), (size_t)((uint32_t)entryPtr >> 1));
  uint64_t v = 0;
  for (size_t i = 0; i < row_.size(); ++i) row_[i] = r.read(&v) ? (int32_t)v : 0;
}

void GoldNodeResolver::unkRow(const jppgpu_unk& unk) {
  dicRow(unk.template_ptr);
  const uint32_t mask = unk.maker < model_->numUnkMakers() ? model_->unkMaker(unk.maker).replace_mask : 0u;
  for (size_t i = 0; i < row_.size(); ++i)
    if ((mask >> i) & 1u) row_[i] = unk.content_hash;
}

bool GoldNodeResolver::matchDic(const GoldWord& w) const {
  for (int32_t i = 0; i < w.numFields; ++i) {
    const TrainingFieldSpec& tf = spec_->fields[(size_t)i];
    if (tf.weight == 0) continue;
    if (w.data[tf.number] != row_[(size_t)tf.dicIdx]) return false;
  }
  return true;
}

bool GoldNodeResolver::matchUnk(const GoldWord& w, int32_t surfaceHash) const {
  for (int32_t i = 0; i < w.numFields; ++i) {
    const TrainingFieldSpec& tf = spec_->fields[(size_t)i];
    if (tf.weight == 0) continue;
    // (the reference indexes the node's row with the field's SPEC index here, gold_example.cc:30-31; kept)
    const int32_t col = tf.fieldIdx;
    const int32_t lat = col >= 0 && (size_t)col < row_.size() ? row_[(size_t)col] : 0;
    if (lat < 0) {
      if (lat != surfaceHash) return false;
    } else if (lat != w.data[tf.number]) {
      return false;
    }
  }
  return true;
}

Status GoldNodeResolver::resolve(const GoldExample& ex, const jppgpu_seed_view& view, uint32_t s, std::vector<GoldPosition>* path,
                                 std::vector<jppgpu_extra_seed>* extra) const {
  GoldNodeResolver* self = const_cast<GoldNodeResolver*>(this);   // (row_ is scratch)
  path->clear();
  const std::string& text = ex.surface();
  // byte offset of every codepoint of the sentence
  std::vector<uint32_t> cpOff;
  for (size_t i = 0; i < text.size();) {
    cpOff.push_back((uint32_t)i);
    const unsigned char c = (unsigned char)text[i];
    i += c < 0x80 ? 1 : (c >> 5) == 6 ? 2 : (c >> 4) == 14 ? 3 : 4;
  }
  cpOff.push_back((uint32_t)text.size());
  const uint32_t ncp = (uint32_t)cpOff.size() - 1;
  if (view.status[s] != JPPGPU_SENT_OK && view.status[s] != JPPGPU_SENT_NO_LATTICE) {
    return Status::InvalidState() << "the analyser rejected the example (status " << view.status[s] << ")";
  }
  if (ncp != view.n_codepoints[s]) return Status::InvalidState() << "codepoint count of the example differs from the analyser's";
  const jppgpu_node* seeds = view.seeds + view.seed_base[s];
  const jppgpu_unk* unk = view.unk + view.seed_base[s];
  const uint32_t nseeds = view.n_seeds[s];
  uint32_t cursor = 0;   // seeds are sorted by start; words come in text order
  for (int32_t wi = 0; wi < ex.numWords(); ++wi) {
    const GoldWord w = ex.word(wi);
    if ((uint32_t)(w.position + w.length) > ncp || w.length <= 0) return Status::InvalidState() << "gold word outside the sentence";
    while (cursor < nseeds && seeds[cursor].start < (uint32_t)w.position) ++cursor;
    uint32_t last = cursor;
    while (last < nseeds && seeds[last].start == (uint32_t)w.position) ++last;
    // TrainingExampleAdapter::nodeSeedExists (gold_example.cc:60-116)
    int32_t found = -1;
    if (w.data[surfaceColumn_] < 0) {
      const int32_t hash = hashUnkString(w.surface);
      auto sameSurface = [&](uint32_t k) {
        const uint32_t b = cpOff[seeds[k].start], e = cpOff[seeds[k].end];
        return e - b == w.surface.size() && std::memcmp(text.data() + b, w.surface.data(), e - b) == 0;
      };
      for (uint32_t k = cursor; k < last && found < 0; ++k) {
        if (seeds[k].entry_ptr >= 0 || !sameSurface(k)) continue;
        self->unkRow(unk[k]);
        if (matchUnk(w, hash)) found = (int32_t)(k - cursor);
      }
      for (uint32_t k = cursor; k < last && found < 0; ++k) {
        if (seeds[k].entry_ptr >= 0 || !sameSurface(k)) continue;
        if (allowed_.isAllowed(unk[k].template_ptr, w)) found = (int32_t)(k - cursor);
      }
    } else {
      for (uint32_t k = cursor; k < last && found < 0; ++k) {
        if (seeds[k].entry_ptr < 0) continue;
        self->dicRow(seeds[k].entry_ptr);
        if (matchDic(w)) found = (int32_t)(k - cursor);
      }
    }
    GoldPosition gp;
    gp.boundary = (uint16_t)(w.position + 2);
    if (found >= 0) {
      gp.position = (uint16_t)found;
    } else {
      // makeUnkTrainingNode (gold_example.cc:118-136)
      gp.position = (uint16_t)(last - cursor);
      jppgpu_extra_seed e;
      std::memset(&e, 0, sizeof(e));
      e.start = (uint16_t)w.position;
      e.end = (uint16_t)(w.position + w.length);
      e.content_hash = hashUnkString(w.surface);
      for (int32_t i = 0; i < w.numFields; ++i) {
        const int32_t col = spec_->fields[(size_t)i].dicIdx;
        if (col < 0 || col >= 8 || col >= model_->numFeatures()) continue;
        e.row[col] = w.data[i] >= 0 ? w.data[i] : e.content_hash;
      }
      extra->push_back(e);
    }
    path->push_back(gp);
  }
  return Status::Ok();
}

}  // namespace train
}  // namespace jumanpp_amd
