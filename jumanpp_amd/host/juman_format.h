// JUMAN output format, byte-for-byte what the reference's
// jumandic::output::JumanFormat prints for the top-1 path
// (src/jumandic/shared/juman_format.cc:12-168, docs/output.md).
#ifndef JUMANPP_AMD_HOST_JUMAN_FORMAT_H
#define JUMANPP_AMD_HOST_JUMAN_FORMAT_H

#include <atomic>
#include <memory>
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

// byte lengths of the parts of one JUMAN output line (see include/jppgpu.h: jppgpu_format_row)
struct JumanRowPieces {
  uint32_t pre = 0, s = 0, r = 0, b = 0, mid = 0, feat = 0, total = 0;
  bool hasFeatures = false;
};
// the line JumanFormat prints for the entry row the walker stands on, appended to `printer`
void formatJumanRow(const ModelImage& model, const JumandicFields& flds, const NodeWalker& walker, bool first, std::string& printer,
                    JumanRowPieces* pieces);
void formatNormalizedFeature(std::string& p, int32_t v);

// core::OutputFormat (src/core/env.h:73-79), per sentence of the last batch
class OutputFormat {
 public:
  virtual ~OutputFormat() = default;
  virtual Status format(const GpuAnalyzer& analysis, size_t sentence, StringPiece comment) = 0;
  virtual StringPiece result() const = 0;
};

// The text JumanFormat prints for a DICTIONARY node is a function of its entry alone (all rows of an alias entry
// included): formatted once, by whichever format worker meets the entry first, and copied from then on.  One slot per
// entry ((EntryPtr >> 1) >> 3, unique because an entry row is at least 8 bytes long), published with a compare-and-swap;
// a record carries its entry pointer, so a slot that ever served two pointers would simply never hit for the second.
class NodeTextCache {
  struct Record {
    int32_t eptr;
    uint32_t len;
    // text follows
  };
  std::unique_ptr<std::atomic<const Record*>[]> slots_;
  size_t nslots_ = 0;

 public:
  explicit NodeTextCache(size_t entryDataBytes);
  ~NodeTextCache();
  NodeTextCache(const NodeTextCache&) = delete;
  NodeTextCache& operator=(const NodeTextCache&) = delete;
  // the cached text of the dictionary node `eptr`, or an empty piece
  StringPiece find(int32_t eptr) const {
    const size_t slot = (size_t)((uint32_t)eptr >> 4);
    if (eptr < 0 || slot >= nslots_) return StringPiece();
    const Record* r = slots_[slot].load(std::memory_order_acquire);
    if (r == nullptr || r->eptr != eptr) return StringPiece();
    return StringPiece(reinterpret_cast<const char*>(r + 1), r->len);
  }
  void publish(int32_t eptr, StringPiece text);
  // one cache per model, shared by the format objects of all workers
  static std::shared_ptr<NodeTextCache> forModel(const ModelImage* model);
};

class JumanFormat : public OutputFormat {
  const ModelImage* model_ = nullptr;
  JumandicFields flds_;
  std::string printer_;
  NodeWalker walker_;
  std::shared_ptr<NodeTextCache> cache_;
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
