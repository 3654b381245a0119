// What a process derives from a model file before its first batch -- the per-entry T0 records (k_t0_memo) and the
// rendered-row tables of the device formatters (JUMAN top-1, lattice) -- kept in a file beside the model, so that the next process maps it
// instead of walking the trie and printing every dictionary entry again ("device-image cache").
// The reference has no such thing: it re-reads and re-derives on every start (src/core/impl/model_io.cc:115-176).
//
// Key: size and mtime (ns) of the model file, the library's record format (jppgpu_t0_memo_format) and the size of
// jppgpu_format_table.  File: <model>.jppgpu-cache, or, when the model's directory is not writable,
// $TMPDIR/jppgpu-cache-<uid>/<name>.<size>.<mtime>.  Written to a temporary name and renamed; a file that does not match
// is ignored and rewritten.
#ifndef JUMANPP_AMD_HOST_DERIVED_CACHE_H
#define JUMANPP_AMD_HOST_DERIVED_CACHE_H

#include <cstdint>
#include <string>

#include "jppgpu.h"

namespace jumanpp_amd {

class DerivedCache {
  void* map_ = nullptr;
  size_t mapBytes_ = 0;
  const void* memo_ = nullptr;
  uint64_t memoBytes_ = 0;
  uint32_t memoSlots_ = 0;
  bool hasTable_ = false;
  jppgpu_format_table table_{};
  uint64_t tableEntries_ = 0;
  bool hasLattice_ = false;
  jppgpu_lattice_table lattice_{};
  uint64_t latticeEntries_ = 0;

 public:
  DerivedCache() = default;
  DerivedCache(const DerivedCache&) = delete;
  DerivedCache& operator=(const DerivedCache&) = delete;
  ~DerivedCache();

  // maps the cache of `modelPath` when there is one that matches; false otherwise
  bool load(const std::string& modelPath);
  const void* memo() const { return memo_; }
  uint64_t memoBytes() const { return memoBytes_; }
  uint32_t memoSlots() const { return memoSlots_; }
  bool hasFormatTable() const { return hasTable_; }
  const jppgpu_format_table& formatTable() const { return table_; }   // (pointers into the mapping)
  uint64_t formatTableEntries() const { return tableEntries_; }
  // the rendered rows of the lattice format (its score weights are the writer's: the adopter sets its own)
  bool hasLatticeTable() const { return hasLattice_; }
  const jppgpu_lattice_table& latticeTable() const { return lattice_; }
  uint64_t latticeTableEntries() const { return latticeEntries_; }

  // writes the cache of `modelPath`; every part may be absent (null)
  static bool store(const std::string& modelPath, const void* memo, uint64_t memoBytes, uint32_t memoSlots,
                    const jppgpu_format_table* table, uint64_t tableEntries,
                    const jppgpu_lattice_table* lattice = nullptr, uint64_t latticeEntries = 0);
};

}  // namespace jumanpp_amd

#endif  // JUMANPP_AMD_HOST_DERIVED_CACHE_H
