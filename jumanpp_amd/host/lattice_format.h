// Lattice output format (`-s N` / `--lattice N`): the N best paths as a graph of
// morphemes, byte-for-byte what jumandic::output::LatticeFormat prints
// (src/jumandic/shared/lattice_format.{h,cc}).  Reads either the full lattice view (beams + score
// cells: analyzeBatch(inputs, fullLattice = true)) or, after GpuAnalyzer::setLatticeNBest(n), the n best
// paths gathered on the device.
#ifndef JUMANPP_AMD_HOST_LATTICE_FORMAT_H
#define JUMANPP_AMD_HOST_LATTICE_FORMAT_H

#include <string>
#include <vector>

#include "juman_format.h"

namespace jumanpp_amd {

// byte lengths of the parts of the entry-row columns of one lattice line (include/jppgpu.h: jppgpu_lattice_row)
struct LatticeRowPieces {
  uint32_t s = 0, c = 0, r = 0, b = 0, rest = 0, total = 0;
  bool tabField = false;
};
void formatLatticeRow(const ModelImage& model, const JumandicFields& flds, const NodeWalker& walker, std::string& printer,
                      LatticeRowPieces* pieces);

class LatticeFormat : public OutputFormat {
  // LatticeNodeInfo (lattice_format.h:17-25) of the nodes on the printed paths.  Flat records reused from sentence to
  // sentence (the reference keeps a map of vectors per node: at beam 32 on 220-codepoint sentences that bookkeeping,
  // not the printing, was most of the 0.8 ms a sentence cost -- round 5): at most kMaxPaths ranks, distinct beam slots
  // (= the ConnectionPtr set) and distinct previous nodes per node.
  static constexpr int kMaxPaths = 64;
  struct NodeInfo {
    uint32_t node = 0;
    int32_t id = 0;
    uint16_t nRanks = 0, nSlots = 0, nPrev = 0;
    uint16_t ranks[kMaxPaths];
    uint16_t slots[kMaxPaths];
    const jppgpu_nbest_item* items[kMaxPaths];   // (n-best view) the record of (node, slots[k])
    uint32_t prev[kMaxPaths];
  };
  const ModelImage* model_ = nullptr;
  JumandicFields flds_;
  std::string printer_;
  NodeWalker walker_;
  std::vector<NodeInfo> info_;          // in order of first visit
  std::vector<int32_t> infoOf_;         // sentence-local node -> index in info_, -1 (reset through order_)
  std::vector<uint32_t> order_;         // the visited nodes sorted by id = (boundary, position), publishResult's order
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
