// JUMAN output format, byte-for-byte what the reference's
// jumandic::output::JumanFormat prints for the top-1 path
// (src/jumandic/shared/juman_format.cc:12-168, docs/output.md).
#ifndef JUMANPP_AMD_HOST_JUMAN_FORMAT_H
#define JUMANPP_AMD_HOST_JUMAN_FORMAT_H

#include <string>

#include "gpu_analyzer.h"
#include "output.h"

namespace jumanpp_amd {

// jumandic::output::JumandicFields (juman_format.h:22-45)
struct JumandicFields {
  StringField surface, pos, subpos, conjType, conjForm, baseform, reading, canonicForm;
  KVListField features;
  Status initialize(const OutputManager& om);
};

constexpr int NormalizedPlaceholderIdx = 0;  // src/jumandic/shared/jumandic_spec.h:14

// core::OutputFormat (src/core/env.h:73-79), per sentence of the last batch
class OutputFormat {
 public:
  virtual ~OutputFormat() = default;
  virtual Status format(const GpuAnalyzer& analysis, size_t sentence, StringPiece comment) = 0;
  virtual StringPiece result() const = 0;
};

class JumanFormat : public OutputFormat {
  const ModelImage* model_ = nullptr;
  JumandicFields flds_;
  std::string printer_;
  NodeWalker walker_;
  bool formatOne(const OutputManager& om, const SentenceResult& s, uint32_t node, bool first);

 public:
  Status initialize(const ModelImage* model);
  // OutputFormat::format(const Analyzer&, StringPiece comment) for sentence i of the last batch
  Status format(const GpuAnalyzer& analysis, size_t sentence, StringPiece comment) override;
  StringPiece result() const override { return StringPiece(printer_); }
  // JumanppExec::emptyResult (jumandic_env.cc:211-222)
  static StringPiece emptyResult() { return StringPiece("# ERROR\nEOS\n"); }
};

}  // namespace jumanpp_amd

#endif  // JUMANPP_AMD_HOST_JUMAN_FORMAT_H
