// Morph / full-morph (`-M`, `-F`), segmented (`--segment`) and dictionary-subset (`--dic-subset`)
// output formats: jumandic::output::MorphFormat (src/jumandic/shared/morph_format.cc:17-66),
// core::output::SegmentedFormat (src/core/impl/segmented_format.cc:12-38),
// jumandic::output::MdicFormat / SubsetFormat (mdic_format.cc:12-164, subset_format.cc:11-26).
#ifndef JUMANPP_AMD_HOST_SIMPLE_FORMATS_H
#define JUMANPP_AMD_HOST_SIMPLE_FORMATS_H

#include <string>

#include "juman_format.h"

namespace jumanpp_amd {

class MorphFormat : public OutputFormat {
  const ModelImage* model_ = nullptr;
  JumandicFields fields_;
  std::string printer_;
  NodeWalker walker_;
  bool fmrp_;

 public:
  explicit MorphFormat(bool fullMorph) : fmrp_(fullMorph) {}
  Status initialize(const ModelImage* model);
  Status format(const GpuAnalyzer& analysis, size_t sentence, StringPiece comment) override;
  StringPiece result() const override { return StringPiece(printer_); }
};

class SegmentedFormat : public OutputFormat {
  const ModelImage* model_ = nullptr;
  StringField surface_;
  std::string separator_;
  std::string printer_;
  NodeWalker walker_;

 public:
  Status initialize(const ModelImage* model, StringPiece separator = " ");
  Status format(const GpuAnalyzer& analysis, size_t sentence, StringPiece comment) override;
  StringPiece result() const override { return StringPiece(printer_); }
};

// every dictionary node of the lattice as a line of the dictionary CSV it came from (needs the whole
// lattice: analyzeBatch(inputs, fullLattice = true))
class MdicFormat : public OutputFormat {
  const ModelImage* model_ = nullptr;
  JumandicFields fields_;
  std::string printer_;
  NodeWalker walker_;

 public:
  Status initialize(const ModelImage* model);
  Status format(const GpuAnalyzer& analysis, size_t sentence, StringPiece comment) override;
  StringPiece result() const override { return StringPiece(printer_); }
};

class SubsetFormat : public OutputFormat {
  MorphFormat morph_{true};
  MdicFormat mdic_;
  std::string buffer_;

 public:
  Status initialize(const ModelImage* model);
  Status format(const GpuAnalyzer& analysis, size_t sentence, StringPiece comment) override;
  StringPiece result() const override { return StringPiece(buffer_); }
};

}  // namespace jumanpp_amd

#endif  // JUMANPP_AMD_HOST_SIMPLE_FORMATS_H
