#include "juman_format.h"

#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>

namespace jumanpp_amd {

Status JumandicFields::initialize(const OutputManager& om) {
  JPPA_RETURN_IF_ERROR(om.stringField("surface", &surface));
  JPPA_RETURN_IF_ERROR(om.stringField("pos", &pos));
  JPPA_RETURN_IF_ERROR(om.stringField("subpos", &subpos));
  JPPA_RETURN_IF_ERROR(om.stringField("conjtype", &conjType));
  JPPA_RETURN_IF_ERROR(om.stringField("conjform", &conjForm));
  JPPA_RETURN_IF_ERROR(om.stringField("baseform", &baseform));
  JPPA_RETURN_IF_ERROR(om.stringField("reading", &reading));
  JPPA_RETURN_IF_ERROR(om.stringField("canonic", &canonicForm));
  JPPA_RETURN_IF_ERROR(om.kvListField("features", &features));
  return Status::Ok();
}

namespace {

// charlattice::Modifiers (src/core/analysis/charlattice.h:22-34)
enum : int32_t {
  M_REPLACE_SMALLKANA = 0x2, M_REPLACE = 0x4, M_DELETE = 0x8, M_REPLACE_PROLONG = 0x10, M_DELETE_LAST = 0x20,
  M_DELETE_PROLONG = 0x40, M_DELETE_HASTSUON = 0x80, M_DELETE_SMALLKANA = 0x100, M_REPLACE_EROW_WITH_E = 0x200,
};

inline void put(std::string& p, StringPiece s) { p.append(s.data(), s.size()); }

// escapeForJumanOutput (juman_format.cc:42-54)
inline StringPiece escapeForJumanOutput(StringPiece in) {
  if (in.size() == 1) {
    switch (in[0]) {
      case '\t': return StringPiece("\\t");
      case ' ': return StringPiece("\\\xe2\x90\xa3");  // backslash + U+2423
    }
  }
  return in;
}

inline StringPiece ifEmpty(StringPiece s, StringPiece dflt) { return s.empty() ? dflt : s; }

}  // namespace

// formatNormalizedFeature (juman_format.cc:57-92); ExistFlag = any common bit (charlattice.h:55-57)
void formatNormalizedFeature(std::string& p, int32_t v) {
  put(p, "非標準表記:");
  auto has = [v](int32_t f) { return (v & f) != 0; };
  if (has(M_REPLACE)) p += 'R';
  if (has(M_REPLACE_SMALLKANA)) p += 's';
  if (has(M_REPLACE_PROLONG)) p += 'p';
  if (has(M_REPLACE_EROW_WITH_E)) p += 'e';
  if (has(M_DELETE)) p += 'D';
  if (has(M_DELETE_PROLONG)) p += 'P';
  if (has(M_DELETE_SMALLKANA)) p += 'S';
  if (has(M_DELETE_HASTSUON)) p += 'H';
  if (has(M_DELETE_LAST)) p += 'L';
}

NodeTextCache::NodeTextCache(size_t entryDataBytes) : nslots_(entryDataBytes / 8 + 1) {
  slots_.reset(new std::atomic<const Record*>[nslots_]);
  for (size_t i = 0; i < nslots_; ++i) slots_[i].store(nullptr, std::memory_order_relaxed);
}

NodeTextCache::~NodeTextCache() {
  for (size_t i = 0; i < nslots_; ++i) std::free(const_cast<Record*>(slots_[i].load(std::memory_order_relaxed)));
}

void NodeTextCache::publish(int32_t eptr, StringPiece text) {
  const size_t slot = (size_t)((uint32_t)eptr >> 4);
  if (eptr < 0 || slot >= nslots_ || text.size() > 0xffffffu) return;
  if (slots_[slot].load(std::memory_order_relaxed) != nullptr) return;
  Record* r = static_cast<Record*>(std::malloc(sizeof(Record) + text.size()));
  if (r == nullptr) return;
  r->eptr = eptr;
  r->len = (uint32_t)text.size();
  std::memcpy(r + 1, text.data(), text.size());
  const Record* expected = nullptr;
  if (!slots_[slot].compare_exchange_strong(expected, r, std::memory_order_release, std::memory_order_relaxed)) std::free(r);
}

std::shared_ptr<NodeTextCache> NodeTextCache::forModel(const ModelImage* model) {
  static std::mutex mu;
  static std::map<const ModelImage*, std::weak_ptr<NodeTextCache>> caches;
  std::lock_guard<std::mutex> l(mu);
  std::shared_ptr<NodeTextCache> c = caches[model].lock();
  if (!c) {
    c = std::make_shared<NodeTextCache>(model->entryData().size());
    caches[model] = c;
  }
  return c;
}

Status JumanFormat::initialize(const ModelImage* model) {
  model_ = model;
  if (!model->hasIdMap()) {
    return Status::InvalidState("model image has no JUMAN id tables (re-export it with the current ref_dump)");
  }
  cache_ = NodeTextCache::forModel(model);
  OutputManager om(model);
  return flds_.initialize(om);
}

// one output line of a node: the row the walker stands on (juman_format.cc:100-165).  `pieces`, when given, receives the
// byte lengths of the parts the device-side formatter replaces or appends to (format_table.cc)
void formatJumanRow(const ModelImage& model, const JumandicFields& flds, const NodeWalker& walker, bool first, std::string& printer,
                    JumanRowPieces* pieces) {
  const size_t at0 = printer.size();
  if (!first) put(printer, "@ ");
  const size_t atS = printer.size();
  const int32_t* fb = walker.features();
  int32_t ids[4];
  // conjForm and conjType are reversed in the entry row (juman_format.cc:104-106)
  model.dicToJuman(fb[1], fb[2], fb[4], fb[3], ids);
  put(printer, escapeForJumanOutput(flds.surface[walker]));
  const size_t endS = printer.size();
  printer += ' ';
  put(printer, escapeForJumanOutput(flds.reading[walker]));
  const size_t endR = printer.size();
  printer += ' ';
  put(printer, escapeForJumanOutput(flds.baseform[walker]));
  const size_t endB = printer.size();
  printer += ' ';
  put(printer, ifEmpty(flds.pos[walker], "*"));
  printer += ' ';
  printer += std::to_string(ids[0]);
  printer += ' ';
  put(printer, ifEmpty(flds.subpos[walker], "*"));
  printer += ' ';
  printer += std::to_string(ids[1]);
  printer += ' ';
  put(printer, ifEmpty(flds.conjType[walker], "*"));
  printer += ' ';
  printer += std::to_string(ids[2]);
  printer += ' ';
  put(printer, ifEmpty(flds.conjForm[walker], "*"));
  printer += ' ';
  printer += std::to_string(ids[3]);
  printer += ' ';
  const size_t endMid = printer.size();
  KVListIterator res = flds.features[walker];
  StringPiece canonic = flds.canonicForm[walker];
  const bool special = walker.isSpecial();
  const bool hasFeatures = special || res.hasNext() || !canonic.empty();
  size_t featBytes = 0;
  if (!hasFeatures) {
    put(printer, "NIL");
  } else {
    bool output = false;
    printer += '"';
    const size_t featAt = printer.size();
    if (!canonic.empty()) {
      put(printer, "代表表記:");
      put(printer, canonic);
      if (res.hasNext()) printer += ' ';
      output = true;
    }
    while (res.next()) {
      output = true;
      put(printer, res.key());
      if (res.hasValue()) {
        printer += ':';
        put(printer, res.value());
      }
      if (res.hasNext()) printer += ' ';
    }
    featBytes = printer.size() - featAt;
    if (special) {
      int32_t ufld = walker.placeholder(NormalizedPlaceholderIdx);
      if (ufld != 0) {
        if (output) printer += ' ';
        formatNormalizedFeature(printer, ufld);
      }
    }
    printer += '"';
  }
  printer += '\n';
  if (pieces != nullptr) {
    pieces->pre = (uint32_t)(atS - at0);
    pieces->s = (uint32_t)(endS - atS);
    pieces->r = (uint32_t)(endR - endS - 1);
    pieces->b = (uint32_t)(endB - endR - 1);
    pieces->mid = (uint32_t)(endMid - endB);
    pieces->feat = (uint32_t)featBytes;
    pieces->hasFeatures = hasFeatures;
    pieces->total = (uint32_t)(printer.size() - at0);
  }
}

bool JumanFormat::formatOne(const OutputManager& om, const SentenceResult& s, uint32_t node, bool first) {
  // a dictionary node as the first alternative of its position: its text is cached per entry
  if (node >= s.numNodes || s.nodes == nullptr) return false;
  const int32_t eptr = s.nodes[node].entry_ptr;
  const bool cacheable = first && eptr >= 0 && cache_ != nullptr;
  if (cacheable) {
    const StringPiece hit = cache_->find(eptr);
    if (!hit.empty()) {
      printer_.append(hit.data(), hit.size());
      return true;
    }
  }
  const size_t startOfNode = printer_.size();
  if (!om.locate(s, node, &walker_)) return false;
  while (walker_.next()) {
    formatJumanRow(*model_, flds_, walker_, first, printer_, nullptr);
    first = false;
  }
  if (cacheable) cache_->publish(eptr, StringPiece(printer_.data() + startOfNode, printer_.size() - startOfNode));
  return true;
}

Status JumanFormat::format(const GpuAnalyzer& analysis, size_t sentence, StringPiece comment) {
  printer_.clear();
  JPPA_RETURN_IF_ERROR(analysis.sentenceStatus(sentence));
  SentenceResult s = analysis.sentence(sentence);
  OutputManager om(model_);
  if (!comment.empty()) {
    put(printer_, "# ");
    put(printer_, comment);
    printer_ += '\n';
  }
  // AnalysisPath::fillIn + nextBoundary/nextNode (analysis_result.cc:25-72): the top-1 path in text
  // order, one node per chunk; pathNodes is EOS first
  for (uint32_t k = s.pathLen; k-- > 1;) {
    if (!formatOne(om, s, s.pathNodes[k], true)) return Status::InvalidState("failed to load a node");
  }
  put(printer_, "EOS\n");
  return Status::Ok();
}

}  // namespace jumanpp_amd
