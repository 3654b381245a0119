#include "model_image.h"

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cstring>
#include <fstream>

namespace jumanpp_amd {

namespace {

enum : uint32_t {
  SEC_INFO = 1, SEC_TRIE = 2, SEC_ENTRY_PTRS = 3, SEC_ENTRY_DATA = 4, SEC_WEIGHTS = 5, SEC_UNK = 6,
  SEC_FEATURES = 7, SEC_FIELDS = 8, SEC_STRINGS = 9, SEC_INTS = 10, SEC_RNN = 11, SEC_IDMAP = 12, SEC_TRAIN = 13,
};

struct Cursor {
  const char* p;
  const char* end;
  bool ok = true;
  template <typename T>
  T get() {
    T v{};
    if (p + sizeof(T) > end) {
      ok = false;
      return v;
    }
    std::memcpy(&v, p, sizeof(T));
    p += sizeof(T);
    return v;
  }
  StringPiece bytes(size_t n) {
    if (p + n > end) {
      ok = false;
      return StringPiece();
    }
    StringPiece r(p, n);
    p += n;
    return r;
  }
  void align8(const char* base) {
    size_t off = (size_t)(p - base);
    p = base + ((off + 7) & ~(size_t)7);
  }
};

inline uint64_t key2(int32_t a, int32_t b) { return ((uint64_t)(uint32_t)a << 32) | (uint32_t)b; }

}  // namespace

void FileBytes::reset() {
  if (p_) ::munmap(p_, map_);
  p_ = nullptr;
  map_ = size_ = 0;
}

bool FileBytes::open(const std::string& path) {
  reset();
  const int fd = ::open(path.c_str(), O_RDONLY);
  if (fd < 0) return false;
  struct stat st;
  if (::fstat(fd, &st) != 0 || !S_ISREG(st.st_mode)) {
    ::close(fd);
    return false;
  }
  const size_t sz = (size_t)st.st_size, page = 4096;
  const size_t total = (sz + page - 1) / page * page + page;
  void* r = ::mmap(nullptr, total, PROT_READ, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
  if (r == MAP_FAILED) {
    ::close(fd);
    return false;
  }
  if (sz != 0 && ::mmap(r, sz, PROT_READ, MAP_PRIVATE | MAP_FIXED | MAP_POPULATE, fd, 0) == MAP_FAILED) {
    ::munmap(r, total);
    ::close(fd);
    return false;
  }
  ::close(fd);
  p_ = static_cast<char*>(r);
  map_ = total;
  size_ = sz;
  return true;
}

Status ModelImage::loadModel(StringPiece filename) {
  std::string fn = filename.str();
  // (every section payload starts on an 8-byte boundary of the file, and the mapping is page aligned)
  if (!data_.open(fn)) return Status::InvalidParameter() << "could not open model image " << fn;
  const std::streamsize sz = (std::streamsize)data_.size();
  fileSize_ = (size_t)sz;
  const char* base = data_.data();
  if (sz >= 24 && std::memcmp(base, "jp2Mdl!", 8) == 0) return loadJppmdl(fn);
  if (sz < 24 || std::memcmp(base, "JPPGPUI1", 8) != 0) {
    return Status::InvalidParameter() << "model file " << fn << " has corrupted header";
  }
  return loadImage(fn);
}

Status ModelImage::loadImage(const std::string& fn) {
  const char* base = data_.data();
  const std::streamsize sz = (std::streamsize)fileSize_;

  struct Sec {
    uint32_t tag, aux;
    StringPiece body;
  };
  std::vector<Sec> secs;
  size_t pos = 8;
  for (;;) {
    pos = (pos + 7) & ~(size_t)7;
    if (pos + 16 > (size_t)sz) return Status::InvalidParameter() << "truncated model image " << fn;
    uint32_t tag, aux;
    uint64_t size;
    std::memcpy(&tag, base + pos, 4);
    std::memcpy(&aux, base + pos + 4, 4);
    std::memcpy(&size, base + pos + 8, 8);
    pos += 16;
    if (tag == 0) break;
    if (pos + size > (size_t)sz) return Status::InvalidParameter() << "truncated section in model image " << fn;
    secs.push_back(Sec{tag, aux, StringPiece(base + pos, (size_t)size)});
    pos += size;
  }
  auto find = [&](uint32_t tag, uint32_t aux, bool anyAux) -> const Sec* {
    for (auto& s : secs)
      if (s.tag == tag && (anyAux || s.aux == aux)) return &s;
    return nullptr;
  };

  const Sec* info = find(SEC_INFO, 0, true);
  if (!info || info->body.size() < 32) return Status::InvalidParameter() << "model image has no INFO section";
  int32_t iv[8];
  std::memcpy(iv, info->body.data(), 32);
  numFeatures_ = iv[0];
  numData_ = iv[1];
  numPlaceholders_ = iv[2];
  stringStorages_.assign((size_t)iv[6], StringPiece());
  intStorages_.assign((size_t)iv[7], StringPiece());

  jppgpu_model& m = model_;
  std::memset(&m, 0, sizeof(m));
  const Sec *trie = find(SEC_TRIE, 0, true), *eptrs = find(SEC_ENTRY_PTRS, 0, true), *edata = find(SEC_ENTRY_DATA, 0, true);
  const Sec *wts = find(SEC_WEIGHTS, 0, true), *unk = find(SEC_UNK, 0, true), *feat = find(SEC_FEATURES, 0, true);
  if (!trie || !eptrs || !edata || !unk || !feat) return Status::InvalidParameter() << "model image lacks a dictionary section";
  if (!wts) return Status::InvalidParameter() << "model image has no perceptron weights (untrained model)";
  m.trie = trie->body.data();
  m.trie_bytes = trie->body.size();
  m.entry_ptrs = eptrs->body.data();
  m.entry_ptrs_bytes = eptrs->body.size();
  m.entry_data = edata->body.data();
  m.entry_data_bytes = edata->body.size();
  m.weights = reinterpret_cast<const float*>(wts->body.data());
  m.weight_exponent = wts->aux;
  m.num_features = numFeatures_;
  m.num_placeholders = numPlaceholders_;
  m.feature_spec = feat->body.data();
  m.feature_spec_bytes = feat->body.size();
  {
    Cursor c{unk->body.data(), unk->body.data() + unk->body.size()};
    int32_t n = c.get<int32_t>();
    for (int32_t i = 0; i < n && c.ok; ++i) {
      jppgpu_unk_maker k{};
      k.type = c.get<int32_t>();
      k.char_class = c.get<int32_t>();
      k.pattern_ptr = c.get<int32_t>();
      k.priority = c.get<int32_t>();
      k.placeholder = c.get<int32_t>();
      int32_t nrep = c.get<int32_t>();
      for (int32_t r = 0; r < nrep; ++r) k.replace_mask |= 1u << c.get<int32_t>();
      makers_.push_back(k);
    }
    if (!c.ok) return Status::InvalidParameter() << "bad UNK section";
    m.unk_makers = makers_.data();
    m.num_unk_makers = (int32_t)makers_.size();
  }
  if (const Sec* fs = find(SEC_FIELDS, 0, true)) {
    const char* b = fs->body.data();
    Cursor c{b, b + fs->body.size()};
    int32_t n = c.get<int32_t>();
    for (int32_t i = 0; i < n && c.ok; ++i) {
      DictionaryField d;
      d.idxInEntry = c.get<int32_t>();
      d.specIndex = c.get<int32_t>();
      d.columnType = (FieldType)c.get<int32_t>();
      d.stringStorage = c.get<int32_t>();
      d.intStorage = c.get<int32_t>();
      d.alignPower = (uint32_t)c.get<int32_t>();
      d.isTrieKey = c.get<int32_t>() != 0;
      d.name = c.bytes((size_t)c.get<int32_t>()).str();
      d.emptyValue = c.bytes((size_t)c.get<int32_t>()).str();
      c.align8(b);
      fields_.push_back(std::move(d));
    }
    if (!c.ok) return Status::InvalidParameter() << "bad FIELDS section";
  }
  for (auto& s : secs) {
    if (s.tag == SEC_STRINGS && s.aux < stringStorages_.size()) stringStorages_[s.aux] = s.body;
    if (s.tag == SEC_INTS && s.aux < intStorages_.size()) intStorages_[s.aux] = s.body;
  }
  for (uint32_t which = 0; which < 2; ++which) {
    if (const Sec* im = find(SEC_IDMAP, which, false)) {
      hasIdMap_ = true;
      auto& map = which == 0 ? posMap_ : conjMap_;
      for (size_t o = 0; o + 16 <= im->body.size(); o += 16) {
        int32_t v[4];
        std::memcpy(v, im->body.data() + o, 16);
        map[key2(v[0], v[1])] = key2(v[2], v[3]);
      }
    }
  }
  if (const Sec* ts = find(SEC_TRAIN, 0, true)) {
    const char* b = ts->body.data();
    Cursor c{b, b + ts->body.size()};
    int32_t n = c.get<int32_t>();
    for (int32_t i = 0; i < n && c.ok; ++i) {
      TrainField tf;
      tf.dicIdx = c.get<int32_t>();
      tf.name = c.bytes((size_t)c.get<int32_t>()).str();
      c.align8(b);
      trainFields_.push_back(std::move(tf));
    }
    if (!c.ok) return Status::InvalidParameter() << "bad TRAIN section";
  }
  if (const Sec* rh = find(SEC_RNN, 100, false)) {
    Cursor c{rh->body.data(), rh->body.data() + rh->body.size()};
    m.rnn_layer_size = c.get<uint32_t>();
    m.rnn_maxent_order = c.get<uint32_t>();
    m.rnn_maxent_size = c.get<uint64_t>();
    m.rnn_vocab_size = c.get<uint64_t>();
    m.rnn_nce_constant = c.get<float>();
    m.rnn_unk_id = c.get<int32_t>();
    m.rnn_unk_constant = c.get<float>();
    m.rnn_unk_length = c.get<float>();
    rnnWeights_.perceptron = c.get<float>();
    rnnWeights_.rnn = c.get<float>();
    m.rnn_num_fields = c.get<uint32_t>();
    if (!c.ok || m.rnn_num_fields > 8) return Status::InvalidParameter() << "bad RNN header section";
    for (uint32_t i = 0; i < m.rnn_num_fields; ++i) m.rnn_fields[i] = c.get<uint32_t>();
    const Sec* blk[7] = {};
    for (uint32_t i = 1; i <= 6; ++i) {
      blk[i] = find(SEC_RNN, i, false);
      if (!blk[i]) return Status::InvalidParameter() << "model image lacks RNN block " << i;
    }
    m.rnn_known_index = blk[1]->body.data();
    m.rnn_known_index_bytes = blk[1]->body.size();
    m.rnn_unk_index = blk[2]->body.data();
    m.rnn_unk_index_bytes = blk[2]->body.size();
    m.rnn_matrix = reinterpret_cast<const float*>(blk[3]->body.data());
    m.rnn_embeddings = reinterpret_cast<const float*>(blk[4]->body.data());
    m.rnn_nce_embeddings = reinterpret_cast<const float*>(blk[5]->body.data());
    m.rnn_maxent = reinterpret_cast<const float*>(blk[6]->body.data());
    m.has_rnn = 1;
    hasRnn_ = true;
  }
  return Status::Ok();
}

const DictionaryField* ModelImage::fieldByName(StringPiece name) const {
  for (auto& f : fields_)
    if (f.name.size() == name.size() && std::memcmp(f.name.data(), name.data(), name.size()) == 0) return &f;
  return nullptr;
}

void ModelImage::dicToJuman(int32_t pos, int32_t subpos, int32_t conjType, int32_t conjForm, int32_t out[4]) const {
  out[0] = out[1] = out[2] = out[3] = 0;
  auto p = posMap_.find(key2(pos, subpos));
  if (p != posMap_.end()) {
    out[0] = (int32_t)(p->second >> 32);
    out[1] = (int32_t)(uint32_t)p->second;
  }
  auto c = conjMap_.find(key2(conjType, conjForm));
  if (c != conjMap_.end()) {
    out[2] = (int32_t)(c->second >> 32);
    out[3] = (int32_t)(uint32_t)c->second;
  }
}

}  // namespace jumanpp_amd
