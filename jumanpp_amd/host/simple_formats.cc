#include "simple_formats.h"

#include <algorithm>

namespace jumanpp_amd {

namespace {
inline void put(std::string& p, StringPiece s) { p.append(s.data(), s.size()); }
inline StringPiece ifEmpty(StringPiece s, StringPiece d) { return s.empty() ? d : s; }
}  // namespace

Status MorphFormat::initialize(const ModelImage* model) {
  model_ = model;
  OutputManager om(model);
  return fields_.initialize(om);
}

Status MorphFormat::format(const GpuAnalyzer& analysis, size_t sentence, StringPiece comment) {
  printer_.clear();
  JPPA_RETURN_IF_ERROR(analysis.sentenceStatus(sentence));
  SentenceResult s = analysis.sentence(sentence);
  OutputManager om(model_);
  auto& f = fields_;
  for (uint32_t k = s.pathLen; k-- > 1;) {
    if (!om.locate(s, s.pathNodes[k], &walker_)) return Status::InvalidState() << "could not find a ready node: " << s.pathNodes[k];
    if (!walker_.next()) return Status::InvalidState("could not walk on the node");
    if (fmrp_) {
      put(printer_, f.surface[walker_]);
      printer_ += '_';
      put(printer_, f.reading[walker_]);
      printer_ += '_';
      put(printer_, f.baseform[walker_]);
      printer_ += '_';
      put(printer_, f.pos[walker_]);
      printer_ += '_';
      put(printer_, ifEmpty(f.subpos[walker_], "*"));
      printer_ += '_';
      put(printer_, ifEmpty(f.conjType[walker_], "*"));
      printer_ += '_';
      put(printer_, ifEmpty(f.conjForm[walker_], "*"));
    } else {
      put(printer_, f.surface[walker_]);
      printer_ += '_';
      put(printer_, f.pos[walker_]);
      printer_ += ':';
      put(printer_, ifEmpty(f.subpos[walker_], "*"));
    }
    printer_ += ' ';
  }
  if (comment.size() > 0) {
    put(printer_, "# ");
    put(printer_, comment);
  }
  printer_ += '\n';
  return Status::Ok();
}

Status SegmentedFormat::initialize(const ModelImage* model, StringPiece separator) {
  model_ = model;
  separator_ = separator.str();
  OutputManager om(model);
  // the index column of the jumandic spec (spec().dictionary.indexColumn) is "surface"
  return om.stringField("surface", &surface_);
}

Status SegmentedFormat::format(const GpuAnalyzer& analysis, size_t sentence, StringPiece) {
  printer_.clear();
  JPPA_RETURN_IF_ERROR(analysis.sentenceStatus(sentence));
  SentenceResult s = analysis.sentence(sentence);
  OutputManager om(model_);
  for (uint32_t k = s.pathLen; k-- > 1;) {
    if (!om.locate(s, s.pathNodes[k], &walker_)) return Status::InvalidParameter() << "failed to find a node at " << s.pathNodes[k];
    put(printer_, surface_[walker_]);
    if (k != 1) printer_ += separator_;
  }
  printer_ += '\n';
  return Status::Ok();
}

namespace {
// fmt::printMaybeQuoted (mdic_format.cc:41-62): CSV quoting, embedded quotes doubled
void putMaybeQuoted(std::string& p, StringPiece data) {
  const char* beg = data.data();
  const char* end = beg + data.size();
  const bool noCommas = std::find(beg, end, ',') == end;
  const char* it = std::find(beg, end, '"');
  if (noCommas && it == end) {
    p.append(beg, end);
    return;
  }
  p += '"';
  while (it != end) {
    p.append(beg, it);
    p += '"';
    beg = it;
    it = std::find(it + 1, end, '"');
  }
  p.append(beg, end);
  p += '"';
}

// fmt::printMaybeQuoteStringList (mdic_format.cc:64-125): the whole key:value list is quoted if any
// key or value needs it
void putMaybeQuotedList(std::string& p, KVListIterator items) {
  auto needs = [](StringPiece sp) {
    return std::find_if(sp.data(), sp.data() + sp.size(), [](char c) { return c == '"' || c == ','; }) != sp.data() + sp.size();
  };
  bool shouldQuote = false;
  for (KVListIterator copy = items; copy.next();) {
    if (needs(copy.key()) || (copy.hasValue() && needs(copy.value()))) {
      shouldQuote = true;
      break;
    }
  }
  auto putQuoted = [&](StringPiece sp) {
    const char* beg = sp.data();
    const char* end = beg + sp.size();
    const char* it = std::find(beg, end, '"');
    while (it != end) {
      p.append(beg, it);
      p += '"';
      beg = it;
      it = std::find(it + 1, end, '"');
    }
    p.append(beg, end);
  };
  if (shouldQuote) p += '"';
  while (items.next()) {
    if (shouldQuote) putQuoted(items.key());
    else put(p, items.key());
    if (items.hasValue()) {
      p += ':';
      if (shouldQuote) putQuoted(items.value());
      else put(p, items.value());
    }
    if (items.hasNext()) p += ' ';
  }
  if (shouldQuote) p += '"';
}
}  // namespace

Status MdicFormat::initialize(const ModelImage* model) {
  model_ = model;
  OutputManager om(model);
  return fields_.initialize(om);
}

Status MdicFormat::format(const GpuAnalyzer& analysis, size_t sentence, StringPiece) {
  printer_.clear();
  JPPA_RETURN_IF_ERROR(analysis.sentenceStatus(sentence));
  SentenceResult s = analysis.sentence(sentence);
  OutputManager om(model_);
  auto& f = fields_;
  // boundaries 2 .. EOS-1 in order, every node: the node table is laid out that way.  (The reference keeps
  // a `displayed_` set but never inserts into it, so an entry is printed once per node that carries it.)
  for (uint32_t k = 2; k + 1 < s.numNodes; ++k) {
    if (s.nodes[k].entry_ptr < 0) continue;  // eptr.isDic()
    if (!om.locate(s, k, &walker_)) return Status::InvalidState() << "could not find a ready node with eptr: " << s.nodes[k].entry_ptr;
    while (walker_.next()) {
      putMaybeQuoted(printer_, f.surface[walker_]);
      put(printer_, ",0,0,0,");
      putMaybeQuoted(printer_, f.pos[walker_]);
      printer_ += ',';
      putMaybeQuoted(printer_, f.subpos[walker_]);
      printer_ += ',';
      putMaybeQuoted(printer_, f.conjForm[walker_]);
      printer_ += ',';
      putMaybeQuoted(printer_, f.conjType[walker_]);
      printer_ += ',';
      putMaybeQuoted(printer_, f.baseform[walker_]);
      printer_ += ',';
      putMaybeQuoted(printer_, f.reading[walker_]);
      printer_ += ',';
      putMaybeQuoted(printer_, f.canonicForm[walker_]);
      printer_ += ',';
      putMaybeQuotedList(printer_, f.features[walker_]);
      printer_ += '\n';
    }
  }
  return Status::Ok();
}

Status SubsetFormat::initialize(const ModelImage* model) {
  JPPA_RETURN_IF_ERROR(morph_.initialize(model));
  return mdic_.initialize(model);
}

Status SubsetFormat::format(const GpuAnalyzer& analysis, size_t sentence, StringPiece comment) {
  JPPA_RETURN_IF_ERROR(morph_.format(analysis, sentence, comment));
  JPPA_RETURN_IF_ERROR(mdic_.format(analysis, sentence, comment));
  buffer_.assign("#### MRPH output ####\n");
  put(buffer_, morph_.result());
  buffer_ += "\n\n### SUBSET OF DICTIONARY\n";
  put(buffer_, mdic_.result());
  return Status::Ok();
}

}  // namespace jumanpp_amd
