#include "derived_cache.h"

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace jumanpp_amd {

namespace {

struct Header {
  char magic[8];            // "JPPGPUDC"
  uint32_t version;
  uint32_t tableStructSize;
  uint64_t memoFormat;
  uint64_t modelSize;
  int64_t modelMtimeSec;
  int64_t modelMtimeNsec;
  uint64_t memoOffset, memoBytes;
  uint32_t memoSlots;
  uint32_t hasTable;
  uint64_t tableEntries;
  uint64_t slotsOffset, nSlots;     // u32[nSlots]
  uint64_t rowsOffset, nRows;       // jppgpu_format_row[nRows]
  uint64_t blobOffset, blobBytes;
  uint64_t totalBytes;
  jppgpu_format_table literals;     // the table with its three pointers null
};
constexpr uint32_t kVersion = 1;

bool statModel(const std::string& path, struct stat* st) { return ::stat(path.c_str(), st) == 0 && S_ISREG(st->st_mode); }

std::string baseName(const std::string& p) {
  const size_t k = p.find_last_of('/');
  return k == std::string::npos ? p : p.substr(k + 1);
}

// the two places a cache may live: beside the model, or in a per-user directory under $TMPDIR
std::vector<std::string> candidates(const std::string& modelPath, const struct stat& st) {
  std::vector<std::string> out;
  out.push_back(modelPath + ".jppgpu-cache");
  const char* tmp = std::getenv("TMPDIR");
  char buf[96];
  std::snprintf(buf, sizeof(buf), ".%llu.%lld", (unsigned long long)st.st_size, (long long)st.st_mtim.tv_sec);
  out.push_back(std::string(tmp && *tmp ? tmp : "/tmp") + "/jppgpu-cache-" + std::to_string((unsigned)::getuid()) + "/" + baseName(modelPath) + buf);
  return out;
}

bool writeAll(int fd, const void* p, size_t n) {
  const char* c = static_cast<const char*>(p);
  while (n > 0) {
    const ssize_t w = ::write(fd, c, n > (size_t{1} << 30) ? (size_t{1} << 30) : n);
    if (w <= 0) return false;
    c += w;
    n -= (size_t)w;
  }
  return true;
}

bool padTo(int fd, uint64_t* pos, uint64_t align) {
  static const char zeros[64] = {0};
  const uint64_t pad = (align - *pos % align) % align;
  if (pad && !writeAll(fd, zeros, (size_t)pad)) return false;
  *pos += pad;
  return true;
}

}  // namespace

DerivedCache::~DerivedCache() {
  if (map_) ::munmap(map_, mapBytes_);
}

bool DerivedCache::load(const std::string& modelPath) {
  struct stat st;
  if (!statModel(modelPath, &st)) return false;
  for (const std::string& path : candidates(modelPath, st)) {
    const int fd = ::open(path.c_str(), O_RDONLY);
    if (fd < 0) continue;
    struct stat cs;
    if (::fstat(fd, &cs) != 0 || (size_t)cs.st_size < sizeof(Header)) {
      ::close(fd);
      continue;
    }
    // MAP_POPULATE: the pages are wanted at once (the records are uploaded, the table's blob as well)
    void* m = ::mmap(nullptr, (size_t)cs.st_size, PROT_READ, MAP_PRIVATE | MAP_POPULATE, fd, 0);
    ::close(fd);
    if (m == MAP_FAILED) continue;
    const Header* h = static_cast<const Header*>(m);
    const bool ok = std::memcmp(h->magic, "JPPGPUDC", 8) == 0 && h->version == kVersion && h->memoFormat == jppgpu_t0_memo_format() &&
                    h->tableStructSize == sizeof(jppgpu_format_table) && h->modelSize == (uint64_t)st.st_size &&
                    h->modelMtimeSec == (int64_t)st.st_mtim.tv_sec && h->modelMtimeNsec == (int64_t)st.st_mtim.tv_nsec &&
                    h->totalBytes == (uint64_t)cs.st_size && h->memoOffset + h->memoBytes <= h->totalBytes &&
                    (!h->hasTable || (h->slotsOffset + h->nSlots * 4 <= h->totalBytes && h->rowsOffset + h->nRows * sizeof(jppgpu_format_row) <= h->totalBytes &&
                                      h->blobOffset + h->blobBytes <= h->totalBytes));
    if (!ok) {
      ::munmap(m, (size_t)cs.st_size);
      continue;
    }
    map_ = m;
    mapBytes_ = (size_t)cs.st_size;
    const char* base = static_cast<const char*>(m);
    if (h->memoBytes) {
      memo_ = base + h->memoOffset;
      memoBytes_ = h->memoBytes;
      memoSlots_ = h->memoSlots;
    }
    if (h->hasTable) {
      table_ = h->literals;
      table_.slot_first_row = reinterpret_cast<const uint32_t*>(base + h->slotsOffset);
      table_.n_slots = h->nSlots;
      table_.rows = reinterpret_cast<const jppgpu_format_row*>(base + h->rowsOffset);
      table_.n_rows = h->nRows;
      table_.blob = base + h->blobOffset;
      table_.blob_bytes = h->blobBytes;
      tableEntries_ = h->tableEntries;
      hasTable_ = true;
    }
    return true;
  }
  return false;
}

bool DerivedCache::store(const std::string& modelPath, const void* memo, uint64_t memoBytes, uint32_t memoSlots,
                         const jppgpu_format_table* table, uint64_t tableEntries) {
  struct stat st;
  if (!statModel(modelPath, &st)) return false;
  if (memo == nullptr && table == nullptr) return false;
  for (const std::string& path : candidates(modelPath, st)) {
    const size_t slash = path.find_last_of('/');
    if (slash != std::string::npos) (void)::mkdir(path.substr(0, slash).c_str(), 0700);   // (the per-user directory; EEXIST otherwise)
    const std::string tmp = path + ".tmp" + std::to_string((long)::getpid());
    const int fd = ::open(tmp.c_str(), O_CREAT | O_TRUNC | O_WRONLY, 0644);
    if (fd < 0) continue;
    Header h;
    std::memset(&h, 0, sizeof(h));
    std::memcpy(h.magic, "JPPGPUDC", 8);
    h.version = kVersion;
    h.tableStructSize = (uint32_t)sizeof(jppgpu_format_table);
    h.memoFormat = jppgpu_t0_memo_format();
    h.modelSize = (uint64_t)st.st_size;
    h.modelMtimeSec = (int64_t)st.st_mtim.tv_sec;
    h.modelMtimeNsec = (int64_t)st.st_mtim.tv_nsec;
    uint64_t pos = (sizeof(Header) + 63) / 64 * 64;
    h.memoOffset = pos;
    h.memoBytes = memo ? memoBytes : 0;
    h.memoSlots = memo ? memoSlots : 0;
    pos = (pos + h.memoBytes + 63) / 64 * 64;
    if (table) {
      h.hasTable = 1;
      h.tableEntries = tableEntries;
      h.literals = *table;
      h.literals.slot_first_row = nullptr;
      h.literals.rows = nullptr;
      h.literals.blob = nullptr;
      h.slotsOffset = pos;
      h.nSlots = table->n_slots;
      pos = (pos + h.nSlots * 4 + 63) / 64 * 64;
      h.rowsOffset = pos;
      h.nRows = table->n_rows;
      pos = (pos + h.nRows * sizeof(jppgpu_format_row) + 63) / 64 * 64;
      h.blobOffset = pos;
      h.blobBytes = table->blob_bytes;
      pos += h.blobBytes;
    }
    h.totalBytes = pos;
    uint64_t at = 0;
    bool ok = writeAll(fd, &h, sizeof(h));
    at = sizeof(h);
    ok = ok && padTo(fd, &at, 64);
    if (ok && h.memoBytes) {
      ok = writeAll(fd, memo, (size_t)h.memoBytes);
      at += h.memoBytes;
      ok = ok && padTo(fd, &at, 64);
    }
    if (ok && table) {
      ok = writeAll(fd, table->slot_first_row, (size_t)h.nSlots * 4);
      at += h.nSlots * 4;
      ok = ok && padTo(fd, &at, 64);
      ok = ok && writeAll(fd, table->rows, (size_t)h.nRows * sizeof(jppgpu_format_row));
      at += h.nRows * sizeof(jppgpu_format_row);
      ok = ok && padTo(fd, &at, 64);
      ok = ok && writeAll(fd, table->blob, (size_t)h.blobBytes);
      at += h.blobBytes;
    }
    ok = ok && at == h.totalBytes;
    ::close(fd);
    if (ok && ::rename(tmp.c_str(), path.c_str()) == 0) return true;
    ::unlink(tmp.c_str());
  }
  return false;
}

}  // namespace jumanpp_amd
