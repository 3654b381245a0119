// The per-model table behind the device-side JUMAN formatter (include/jppgpu.h: jppgpu_format_table,
// csrc/k_format.h): every entry row of the dictionary rendered ONCE, by the same code JumanFormat prints a node with
// (formatJumanRow), plus the few literals of the format.  Replaces, for the top-1 JUMAN output, the per-sentence
// JumanFormat::format of the reference (src/jumandic/shared/juman_format.cc:94-168) and its per-entry text cache here.
#ifndef JUMANPP_AMD_HOST_FORMAT_TABLE_H
#define JUMANPP_AMD_HOST_FORMAT_TABLE_H

#include <string>
#include <vector>

#include "jppgpu.h"
#include "model_image.h"

namespace jumanpp_amd {

// Changes whenever build() or formatJumanRow render an entry differently: part of the derived-image cache key
// (host/derived_cache.cc), so that a table written by an older builder is not adopted after an update.
constexpr uint32_t kFormatTableBuilderVersion = 1;

class JumanFormatTable {
  std::vector<uint32_t> slots_;
  std::vector<jppgpu_format_row> rows_;
  std::string blob_;
  jppgpu_format_table view_{};
  size_t entries_ = 0;
  double buildMs_ = 0;

 public:
  // InvalidState / NotImplemented when the model cannot be rendered by the table (an UNK maker that replaces a field
  // the table has no slot for, no JUMAN id map, entry pointers that are not the builder's consecutive lists):
  // the caller keeps formatting on the host
  Status build(const ModelImage* model, unsigned threads);
  // the table as an earlier process built it for this model (host/derived_cache.h): the arrays stay where they are
  void adopt(const jppgpu_format_table& cached, size_t entries) {
    view_ = cached;
    entries_ = entries;
    buildMs_ = 0;
  }
  const jppgpu_format_table& view() const { return view_; }
  size_t numEntries() const { return entries_; }
  size_t numRows() const { return (size_t)view_.n_rows; }
  size_t blobBytes() const { return (size_t)view_.blob_bytes; }
  double buildMs() const { return buildMs_; }
};

// The same for the lattice (-s N) format (jppgpu_lattice_table, csrc/k_latfmt.h): the entry-row columns of a line --
// surface, canonic form or baseform/reading, reading, baseform, grammar columns with ids, feature list -- rendered once per
// entry row by formatLatticeRow, the printer LatticeFormat itself uses (src/jumandic/shared/lattice_format.cc:168-205).
class LatticeFormatTable {
  std::vector<uint32_t> slots_;
  std::vector<jppgpu_lattice_row> rows_;
  std::string blob_;
  jppgpu_lattice_table view_{};
  size_t entries_ = 0;
  double buildMs_ = 0;

 public:
  // scoreWeights: ScorerDef::scoreWeights (the format multiplies the score cells by them, lattice_format.cc:127,218-227)
  Status build(const ModelImage* model, const std::vector<float>& scoreWeights, unsigned threads);
  // the table as an earlier process built it for this model (host/derived_cache.h), with THIS run's score weights
  Status adopt(const jppgpu_lattice_table& cached, size_t entries, const std::vector<float>& scoreWeights) {
    if (scoreWeights.empty() || scoreWeights.size() > 2) return Status::NotImplemented("lattice table: one or two score weights");
    view_ = cached;
    view_.n_weights = (uint32_t)scoreWeights.size();
    view_.weights[0] = scoreWeights[0];
    view_.weights[1] = scoreWeights.size() > 1 ? scoreWeights[1] : 0.f;
    entries_ = entries;
    buildMs_ = 0;
    return Status::Ok();
  }
  const jppgpu_lattice_table& view() const { return view_; }
  size_t numEntries() const { return entries_; }
  size_t numRows() const { return (size_t)view_.n_rows; }
  size_t blobBytes() const { return (size_t)view_.blob_bytes; }
  double buildMs() const { return buildMs_; }
};

}  // namespace jumanpp_amd

#endif  // JUMANPP_AMD_HOST_FORMAT_TABLE_H
