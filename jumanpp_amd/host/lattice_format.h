// Lattice output format (`-s N` / `--lattice N`): the N best paths as a graph of
// morphemes, byte-for-byte what jumandic::output::LatticeFormat prints
// (src/jumandic/shared/lattice_format.{h,cc}).  Reads either the full lattice view (beams + score
// cells: analyzeBatch(inputs, fullLattice = true)) or, after GpuAnalyzer::setLatticeNBest(n), the n best
// paths gathered on the device.
#ifndef JUMANPP_AMD_HOST_LATTICE_FORMAT_H
#define JUMANPP_AMD_HOST_LATTICE_FORMAT_H

#include <map>
#include <unordered_map>
#include <string>
#include <vector>

#include "juman_format.h"

namespace jumanpp_amd {

class LatticeFormat : public OutputFormat {
  // LatticeNodeInfo (lattice_format.h:17-25) keyed by sentence-local node id, which is already
  // ordered by (boundary, position) like publishResult's sort
  struct NodeInfo {
    std::vector<uint16_t> ranks;
    std::vector<uint32_t> prev;    // distinct previous lattice nodes
    std::vector<uint32_t> slots;   // distinct beam slots of this node = the ConnectionPtr set
    int32_t id = 0;
  };
  const ModelImage* model_ = nullptr;
  JumandicFields flds_;
  std::string printer_;
  NodeWalker walker_;
  std::map<uint32_t, NodeInfo> info_;
  std::unordered_map<uint64_t, const jppgpu_nbest_item*> nbItems_;  // (node << 8 | slot) of the n-best view
  float fakeCells_[2] = {0.f, 0.f};
  int32_t topN_ = 1;
  std::vector<float> weights_;

 public:
  explicit LatticeFormat(int32_t topN) : topN_(topN) {}
  // scoreWeights = ScorerDef::scoreWeights of the analyzer (lattice_format.cc:127)
  Status initialize(const ModelImage* model, const std::vector<float>& scoreWeights);
  Status format(const GpuAnalyzer& analysis, size_t sentence, StringPiece comment) override;
  StringPiece result() const override { return StringPiece(printer_); }
};

}  // namespace jumanpp_amd

#endif  // JUMANPP_AMD_HOST_LATTICE_FORMAT_H
