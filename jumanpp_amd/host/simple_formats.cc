#include "simple_formats.h"

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

}  // namespace jumanpp_amd
