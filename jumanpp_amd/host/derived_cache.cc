#include "derived_cache.h"

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cstdio>
#include <cstdlib>
#include <atomic>
#include <cstring>
#include <thread>
#include <vector>

#include "format_table.h"

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
  uint32_t tableBuilder;            // kFormatTableBuilderVersion of the writer
  uint32_t reserved;
  uint64_t contentHash;             // of everything behind the header (hashPayload); the header's own fields are checked one by one
  jppgpu_format_table literals;     // the table with its three pointers null
  // the lattice format's table (round 6), same shape
  uint32_t hasLattice;
  uint32_t latticeStructSize;
  uint64_t latticeEntries;
  uint64_t latSlotsOffset, nLatSlots;
  uint64_t latRowsOffset, nLatRows;
  uint64_t latBlobOffset, latBlobBytes;
  jppgpu_lattice_table latLiterals;
};
constexpr uint32_t kVersion = 3;

// offset + len within total, without wrapping
bool within(uint64_t off, uint64_t len, uint64_t total) { return off <= total && len <= total - off; }
bool withinN(uint64_t off, uint64_t n, uint64_t each, uint64_t total) { return off <= total && n <= (total - off) / each; }

// Content hash of the payload: 4 MB pieces, each mixed 32 bytes at a time in four lanes (multiply-rotate; ~10 GB/s per
// core), the piece hashes folded in order.  Pieces are hashed by a few threads: a 300 MB image is checked in ~10 ms.
uint64_t rotl(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }
uint64_t hashPiece(const unsigned char* p, size_t n, uint64_t seed) {
  const uint64_t P1 = 0x9E3779B185EBCA87ull, P2 = 0xC2B2AE3D27D4EB4Full;
  uint64_t a = seed + P1, b = seed ^ P2, c = seed * P1 + 1, d = seed - P2;
  size_t i = 0;
  for (; i + 32 <= n; i += 32) {
    uint64_t w[4];
    std::memcpy(w, p + i, 32);
    a = rotl(a + w[0] * P2, 31) * P1;
    b = rotl(b + w[1] * P2, 31) * P1;
    c = rotl(c + w[2] * P2, 31) * P1;
    d = rotl(d + w[3] * P2, 31) * P1;
  }
  uint64_t h = rotl(a, 1) + rotl(b, 7) + rotl(c, 12) + rotl(d, 18) + (uint64_t)n;
  for (; i < n; ++i) h = rotl(h ^ (p[i] * P1), 11) * P2;
  h ^= h >> 33;
  h *= P2;
  h ^= h >> 29;
  return h;
}
uint64_t hashPayload(const unsigned char* p, uint64_t n) {
  const uint64_t piece = uint64_t{4} << 20;
  const size_t np = (size_t)((n + piece - 1) / piece);
  std::vector<uint64_t> hs(np, 0);
  unsigned nt = std::thread::hardware_concurrency();
  nt = nt == 0 ? 1 : nt > 8 ? 8 : nt;
  if (np < 4) nt = 1;
  std::atomic<size_t> next{0};
  auto work = [&]() {
    for (size_t k; (k = next.fetch_add(1)) < np;) {
      const uint64_t lo = k * piece, len = n - lo < piece ? n - lo : piece;
      hs[k] = hashPiece(p + lo, (size_t)len, (uint64_t)k);
    }
  };
  std::vector<std::thread> pool;
  for (unsigned t = 1; t < nt; ++t) pool.emplace_back(work);
  work();
  for (auto& th : pool) th.join();
  uint64_t h = 0x4a50504750554443ull ^ n;
  for (uint64_t x : hs) h = rotl(h ^ x, 27) * 0x9E3779B185EBCA87ull + 0x165667B19E3779F9ull;
  return h;
}

// The per-user fallback directory must be OURS: a directory (not a link), owned by this user, closed to everyone else.
// Anything else -- somebody else made it first -- is not used, neither for reading nor for writing.
bool ownPrivateDir(const std::string& dir, bool create) {
  if (create) (void)::mkdir(dir.c_str(), 0700);
  struct stat ds;
  if (::lstat(dir.c_str(), &ds) != 0) return false;
  return S_ISDIR(ds.st_mode) && ds.st_uid == ::getuid() && (ds.st_mode & 077) == 0;
}

bool statModel(const std::string& path, struct stat* st) { return ::stat(path.c_str(), st) == 0 && S_ISREG(st->st_mode); }

std::string baseName(const std::string& p) {
  const size_t k = p.find_last_of('/');
  return k == std::string::npos ? p : p.substr(k + 1);
}

// the two places a cache may live: beside the model, or in a per-user directory under $TMPDIR
struct Candidate {
  std::string path;
  std::string privateDir;   // non-empty: the per-user directory the file lives in (ownPrivateDir)
};
std::vector<Candidate> candidates(const std::string& modelPath, const struct stat& st) {
  std::vector<Candidate> out;
  out.push_back({modelPath + ".jppgpu-cache", ""});
  const char* tmp = std::getenv("TMPDIR");
  char buf[96];
  std::snprintf(buf, sizeof(buf), ".%llu.%lld", (unsigned long long)st.st_size, (long long)st.st_mtim.tv_sec);
  const std::string dir = std::string(tmp && *tmp ? tmp : "/tmp") + "/jppgpu-cache-" + std::to_string((unsigned)::getuid());
  out.push_back({dir + "/" + baseName(modelPath) + buf, dir});
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
  for (const Candidate& cand : candidates(modelPath, st)) {
    if (!cand.privateDir.empty() && !ownPrivateDir(cand.privateDir, false)) continue;
    const int fd = ::open(cand.path.c_str(), O_RDONLY | O_NOFOLLOW | O_CLOEXEC);
    if (fd < 0) continue;
    struct stat cs;
    // a regular file written by this user or by the owner of the model (whoever can replace the model is trusted with
    // its cache), not writable by group or others
    if (::fstat(fd, &cs) != 0 || !S_ISREG(cs.st_mode) || (size_t)cs.st_size < sizeof(Header) ||
        (cs.st_uid != ::getuid() && cs.st_uid != st.st_uid) || (cs.st_mode & 022) != 0) {
      ::close(fd);
      continue;
    }
    // MAP_POPULATE: the pages are wanted at once (the records are uploaded, the table's blob as well)
    void* m = ::mmap(nullptr, (size_t)cs.st_size, PROT_READ, MAP_PRIVATE | MAP_POPULATE, fd, 0);
    ::close(fd);
    if (m == MAP_FAILED) continue;
    const Header* h = static_cast<const Header*>(m);
    const uint64_t total = (uint64_t)cs.st_size;
    const uint64_t body = (sizeof(Header) + 63) / 64 * 64;
    bool ok = std::memcmp(h->magic, "JPPGPUDC", 8) == 0 && h->version == kVersion && h->memoFormat == jppgpu_t0_memo_format() &&
              h->tableStructSize == sizeof(jppgpu_format_table) && h->modelSize == (uint64_t)st.st_size &&
              h->modelMtimeSec == (int64_t)st.st_mtim.tv_sec && h->modelMtimeNsec == (int64_t)st.st_mtim.tv_nsec &&
              h->totalBytes == total && total >= body && within(h->memoOffset, h->memoBytes, total) && (h->memoBytes == 0 || h->memoOffset >= body) &&
              (!h->hasTable || (h->tableBuilder == kFormatTableBuilderVersion && h->slotsOffset >= body && h->rowsOffset >= body && h->blobOffset >= body &&
                                withinN(h->slotsOffset, h->nSlots, 4, total) && withinN(h->rowsOffset, h->nRows, sizeof(jppgpu_format_row), total) &&
                                within(h->blobOffset, h->blobBytes, total))) &&
              (!h->hasLattice || (h->tableBuilder == kFormatTableBuilderVersion && h->latticeStructSize == sizeof(jppgpu_lattice_table) &&
                                  h->latSlotsOffset >= body && h->latRowsOffset >= body && h->latBlobOffset >= body &&
                                  withinN(h->latSlotsOffset, h->nLatSlots, 4, total) &&
                                  withinN(h->latRowsOffset, h->nLatRows, sizeof(jppgpu_lattice_row), total) &&
                                  within(h->latBlobOffset, h->latBlobBytes, total)));
    ok = ok && hashPayload(static_cast<const unsigned char*>(m) + body, total - body) == h->contentHash;
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
    if (h->hasLattice) {
      lattice_ = h->latLiterals;
      lattice_.slot_first_row = reinterpret_cast<const uint32_t*>(base + h->latSlotsOffset);
      lattice_.n_slots = h->nLatSlots;
      lattice_.rows = reinterpret_cast<const jppgpu_lattice_row*>(base + h->latRowsOffset);
      lattice_.n_rows = h->nLatRows;
      lattice_.blob = base + h->latBlobOffset;
      lattice_.blob_bytes = h->latBlobBytes;
      latticeEntries_ = h->latticeEntries;
      hasLattice_ = true;
    }
    return true;
  }
  return false;
}

bool DerivedCache::store(const std::string& modelPath, const void* memo, uint64_t memoBytes, uint32_t memoSlots,
                         const jppgpu_format_table* table, uint64_t tableEntries, const jppgpu_lattice_table* lattice,
                         uint64_t latticeEntries) {
  struct stat st;
  if (!statModel(modelPath, &st)) return false;
  if (memo == nullptr && table == nullptr && lattice == nullptr) return false;
  for (const Candidate& cand : candidates(modelPath, st)) {
    const std::string& path = cand.path;
    if (!cand.privateDir.empty() && !ownPrivateDir(cand.privateDir, true)) continue;
    const std::string tmp = path + ".tmp" + std::to_string((long)::getpid());
    (void)::unlink(tmp.c_str());
    const int fd = ::open(tmp.c_str(), O_CREAT | O_EXCL | O_WRONLY | O_NOFOLLOW | O_CLOEXEC, 0644);
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
      pos = (pos + h.blobBytes + 63) / 64 * 64;
    }
    if (lattice) {
      h.hasLattice = 1;
      h.latticeStructSize = (uint32_t)sizeof(jppgpu_lattice_table);
      h.latticeEntries = latticeEntries;
      h.latLiterals = *lattice;
      h.latLiterals.slot_first_row = nullptr;
      h.latLiterals.rows = nullptr;
      h.latLiterals.blob = nullptr;
      h.latSlotsOffset = pos;
      h.nLatSlots = lattice->n_slots;
      pos = (pos + h.nLatSlots * 4 + 63) / 64 * 64;
      h.latRowsOffset = pos;
      h.nLatRows = lattice->n_rows;
      pos = (pos + h.nLatRows * sizeof(jppgpu_lattice_row) + 63) / 64 * 64;
      h.latBlobOffset = pos;
      h.latBlobBytes = lattice->blob_bytes;
      pos = (pos + h.latBlobBytes + 63) / 64 * 64;
    }
    h.totalBytes = pos;
    h.tableBuilder = kFormatTableBuilderVersion;
    // the payload exactly as it will lie in the file (pads are zeros), hashed piece by piece without assembling it:
    // hashPayload works on one buffer, so the file is written first and hashed from its own mapping below
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
      ok = ok && padTo(fd, &at, 64);
    }
    if (ok && lattice) {
      ok = writeAll(fd, lattice->slot_first_row, (size_t)h.nLatSlots * 4);
      at += h.nLatSlots * 4;
      ok = ok && padTo(fd, &at, 64);
      ok = ok && writeAll(fd, lattice->rows, (size_t)h.nLatRows * sizeof(jppgpu_lattice_row));
      at += h.nLatRows * sizeof(jppgpu_lattice_row);
      ok = ok && padTo(fd, &at, 64);
      ok = ok && writeAll(fd, lattice->blob, (size_t)h.latBlobBytes);
      at += h.latBlobBytes;
      ok = ok && padTo(fd, &at, 64);
    }
    ok = ok && at == h.totalBytes;
    ::close(fd);
    if (ok) {   // the content hash: over the bytes the file holds, then the header once more
      const int rfd = ::open(tmp.c_str(), O_RDWR | O_NOFOLLOW | O_CLOEXEC);
      ok = rfd >= 0;
      if (ok) {
        void* m = ::mmap(nullptr, (size_t)h.totalBytes, PROT_READ, MAP_SHARED, rfd, 0);
        ok = m != MAP_FAILED;
        if (ok) {
          const uint64_t body = (sizeof(Header) + 63) / 64 * 64;
          h.contentHash = hashPayload(static_cast<const unsigned char*>(m) + body, h.totalBytes - body);
          ::munmap(m, (size_t)h.totalBytes);
          ok = ::pwrite(rfd, &h, sizeof(h), 0) == (ssize_t)sizeof(h);
        }
        ::close(rfd);
      }
    }
    if (ok && ::rename(tmp.c_str(), path.c_str()) == 0) return true;
    ::unlink(tmp.c_str());
  }
  return false;
}

}  // namespace jumanpp_amd
