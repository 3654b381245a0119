#include "rnn_external.h"

#include <algorithm>
#include <cstring>
#include <fstream>
#include <unordered_map>

#include "output.h"

namespace jumanpp_amd {

// ------------------------------------------------------------ double array ----
namespace {

inline uint32_t unitOffset(uint32_t u) { return (u >> 10) << ((u & (1u << 9)) >> 6); }
constexpr uint32_t kUnusedUnit = 0x80000000u;  // its label (bit 31 | low byte) equals no input byte

struct Builder {
  const std::vector<std::pair<std::string, int32_t>>& keys;
  std::vector<uint32_t> units;
  std::vector<uint8_t> used;
  std::vector<uint8_t> usedBases;  // two nodes with one base would accept each other's children
  uint32_t firstFree = 1;
  uint32_t maxUsed = 0;
  Status status;

  void reserve(uint32_t pos) {
    if (pos >= units.size()) {
      size_t n = std::max<size_t>(units.size() * 2, (size_t)pos + 257);
      units.resize(n, kUnusedUnit);
      used.resize(n, 0);
      usedBases.resize(n + 256, 0);
    }
  }

  static bool encodable(uint32_t offset) { return offset < (1u << 21) || ((offset & 0xffu) == 0 && offset < (1u << 29)); }

  // a base such that base ^ label is free for every label, the offset from `id` is encodable, and no
  // other node uses it
  uint32_t findBase(uint32_t id, const std::vector<uint8_t>& labels) {
    // like darts-clone, only the last few thousand units are candidates: holes further back are given up,
    // which keeps the search linear in the number of nodes
    constexpr uint32_t kWindow = 4096;
    const uint32_t from = std::max<uint32_t>(firstFree, maxUsed > kWindow ? maxUsed - kWindow : 1);
    for (uint32_t p = from;; ++p) {
      reserve(p + 256);
      if (used[p]) continue;
      const uint32_t base = p ^ labels[0];
      if (!encodable(id ^ base) || usedBases[base]) continue;
      bool ok = true;
      for (uint8_t c : labels) {
        const uint32_t q = base ^ c;
        reserve(q);
        if (q == 0 || used[q]) {
          ok = false;
          break;
        }
      }
      if (ok) return base;
    }
  }

  // keys [lo, hi) share their first `depth` bytes; `id` is the unit of that prefix
  void place(uint32_t id, size_t lo, size_t hi, size_t depth) {
    if (!status.isOk()) return;
    // children: the terminal (value) first, then the distinct next bytes in order
    bool terminal = keys[lo].first.size() == depth;
    std::vector<uint8_t> labels;
    std::vector<std::pair<size_t, size_t>> ranges;
    size_t k = lo + (terminal ? 1 : 0);
    if (terminal) labels.push_back(0);
    while (k < hi) {
      const uint8_t c = (uint8_t)keys[k].first[depth];
      size_t e = k;
      while (e < hi && (uint8_t)keys[e].first[depth] == c) ++e;
      if (c == 0 && terminal) {
        status = Status::NotImplemented("RNN vocabulary key with a zero byte where another key ends");
        return;
      }
      labels.push_back(c);
      ranges.emplace_back(k, e);
      k = e;
    }
    const uint32_t base = findBase(id, labels);
    usedBases[base] = 1;
    const uint32_t offset = id ^ base;
    uint32_t u = units[id] & 0x800000ffu;  // keep this unit's own label
    u |= offset < (1u << 21) ? (offset << 10) : (((offset >> 8) << 10) | (1u << 9));
    if (terminal) u |= 1u << 8;
    units[id] = u;
    for (uint8_t c : labels) {
      used[base ^ c] = 1;
      maxUsed = std::max<uint32_t>(maxUsed, base ^ c);
    }
    while (firstFree < used.size() && used[firstFree]) ++firstFree;
    size_t r = 0;
    for (size_t li = 0; li < labels.size(); ++li) {
      const uint32_t q = base ^ labels[li];
      if (terminal && li == 0) {
        units[q] = (uint32_t)keys[lo].second | (1u << 31);  // value unit
        continue;
      }
      units[q] = labels[li];
      place(q, ranges[r].first, ranges[r].second, depth + 1);
      ++r;
    }
  }
};

}  // namespace

Status DoubleArrayBuilder::build(std::vector<uint32_t>* out) {
  std::sort(keys_.begin(), keys_.end());
  for (size_t i = 1; i < keys_.size(); ++i)
    if (keys_[i].first == keys_[i - 1].first) return Status::InvalidParameter("duplicate key in the RNN vocabulary index");
  for (auto& kv : keys_)
    if (kv.second < 0) return Status::InvalidParameter("negative value in the RNN vocabulary index");
  Builder b{keys_, {}, {}, {}, 1, 0, Status::Ok()};
  b.reserve(512);
  b.used[0] = 1;
  b.units[0] = 0;
  if (!keys_.empty()) b.place(0, 0, keys_.size(), 0);
  if (!b.status.isOk()) return b.status;
  size_t n = b.units.size();
  while (n > 256 && !b.used[n - 1]) --n;
  n = (n + 255) & ~size_t(255);
  b.units.resize(n, kUnusedUnit);
  *out = std::move(b.units);
  return Status::Ok();
}

bool DoubleArrayBuilder::find(const std::vector<uint32_t>& units, const std::string& key, int32_t* value) {
  if (units.empty()) return false;
  uint32_t id = 0, unit = units[0];
  for (unsigned char c : key) {
    id ^= unitOffset(unit) ^ c;
    if (id >= units.size()) return false;
    unit = units[id];
    if ((unit & ((1u << 31) | 0xffu)) != c) return false;
  }
  if (((unit >> 8) & 1) == 0) return false;
  const uint32_t leaf = id ^ unitOffset(unit);
  if (leaf >= units.size()) return false;
  *value = (int32_t)(units[leaf] & 0x7fffffffu);
  return true;
}

// ------------------------------------------------------------ model reader ----
namespace {

void putVarint(std::string& s, uint32_t v) {  // RnnReprBuilder::addInt (rnn_id_resolver.h:27)
  while (v >= 0x80) {
    s.push_back((char)(v | 0x80));
    v >>= 7;
  }
  s.push_back((char)v);
}

bool readFile(const std::string& path, std::string* out) {
  std::ifstream f(path, std::ios::binary);
  if (!f) return false;
  out->assign((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
  return true;
}

}  // namespace

Status ExternalRnn::load(const std::string& path, const ModelImage& dic, const ExternalRnnConfig& cfg) {
  std::string vocab, nnet;
  if (!readFile(path, &vocab)) return Status::InvalidParameter() << "failed to open the RNN vocabulary " << path;
  if (!readFile(path + ".nnet", &nnet)) return Status::InvalidParameter() << "failed to open the RNN model " << path << ".nnet";

  // readHeader (mikolov_rnn.cc:36-76): packed fields, unaligned
  constexpr size_t kLayerName = 64;
  const size_t headerBytes = 8 + 8 + 4 + 1 + 4 + 1 + kLayerName + 4 + 4;
  if (nnet.size() < headerBytes) return Status::InvalidParameter("RNN model file is too short");
  size_t off = 0;
  auto take = [&](void* dst, size_t n) {
    std::memcpy(dst, nnet.data() + off, n);
    off += n;
  };
  uint64_t sizeVersion, maxentSize;
  uint32_t maxentOrder, layerCount, hsArity;
  uint8_t useNce, reversed;
  float nceLnz;
  char layerType[kLayerName + 1] = {0};
  take(&sizeVersion, 8);
  take(&maxentSize, 8);
  take(&maxentOrder, 4);
  take(&useNce, 1);
  take(&nceLnz, 4);
  take(&reversed, 1);
  take(layerType, kLayerName);
  take(&layerCount, 4);
  take(&hsArity, 4);
  if (sizeVersion / 10000 != 6) return Status::InvalidParameter() << "invalid rnn model version " << sizeVersion / 10000 << " can handle only 6";
  if (!useNce) return Status::InvalidParameter("model was trained without nce, we support only nce models");
  if (std::strcmp(layerType, "sigmoid") != 0) return Status::InvalidParameter() << "only sigmoid activation is supported, model had " << layerType;
  const uint64_t E = sizeVersion % 10000;

  // vocabulary: first space-separated field of every line (MikolovModelReader::parse, mikolov_rnn.cc:166-174)
  std::vector<std::string> words;
  for (size_t p = 0; p < vocab.size();) {
    size_t e = vocab.find('\n', p);
    if (e == std::string::npos) e = vocab.size();
    size_t sp = vocab.find(' ', p);
    if (sp == std::string::npos || sp > e) sp = e;
    if (e > p) words.emplace_back(vocab, p, sp - p);
    p = e + 1;
  }
  const uint64_t V = words.size();
  if (V == 0 || E == 0 || E > 256) return Status::InvalidParameter("unsupported RNN model size");
  const uint64_t need = (2 * V * E + E * E + maxentSize) * 4;
  if (nnet.size() - off != need) return Status::InvalidState("did not read rnn model file fully");
  auto copy = [&](std::vector<float>& v, uint64_t n) {
    v.resize(n);
    std::memcpy(v.data(), nnet.data() + off, n * 4);
    off += n * 4;
  };
  copy(embeddings_, V * E);
  copy(nceEmbeddings_, V * E);
  copy(matrix_, E * E);
  copy(maxent_, maxentSize);

  // RnnIdResolverBuilder::resolveFields (rnn_id_resolver.cc:33-58): string -> storage position per field
  if (cfg.fields.empty() || cfg.fields.size() > 8) return Status::InvalidParameter("--rnn-fields must name 1 to 8 dictionary fields");
  if (cfg.separator.size() != 1) return Status::InvalidState("we support RNN separators only of 1 byte length");
  std::vector<std::unordered_map<std::string, int32_t>> fld2pos(cfg.fields.size());
  for (size_t i = 0; i < cfg.fields.size(); ++i) {
    const DictionaryField* fld = dic.fieldByName(cfg.fields[i]);
    if (fld == nullptr) return Status::InvalidParameter() << "could not find a field with name: " << cfg.fields[i] << " in dictionary";
    if (fld->columnType != FieldType::String)
      return Status::InvalidParameter() << "can use only string-typed field in RNN, " << cfg.fields[i] << " was not";
    part_.rnn_fields[i] = (uint32_t)fld->idxInEntry;
    StringPiece data = dic.stringStorage(fld->stringStorage);
    const uint32_t align = 1u << fld->alignPower;
    VarintReader rdr(data, 0);
    const unsigned char* base = (const unsigned char*)data.data();
    while (rdr.p < rdr.end) {  // StringStorageTraversal (field_reader.h:215-239)
      const int32_t pos = (int32_t)((size_t)(rdr.p - base) >> fld->alignPower);
      StringPiece sp;
      if (!rdr.readString(&sp)) break;
      fld2pos[i][sp.str()] = pos;
      size_t o = (size_t)(rdr.p - base);
      o = (o + align - 1) & ~(size_t)(align - 1);
      rdr.p = base + (o < data.size() ? o : data.size());
    }
  }

  // loadData (rnn_id_resolver.cc:82-129)
  DoubleArrayBuilder knownB, unkB;
  int32_t eosId = -1, unkId = -1;
  const char sep = cfg.separator[0];
  for (size_t w = 0; w < words.size(); ++w) {
    const std::string& word = words[w];
    if (word == cfg.eosSymbol) {
      eosId = (int32_t)w;
      continue;
    }
    if (word == cfg.unkSymbol) {
      unkId = (int32_t)w;
      continue;
    }
    std::vector<std::string> parts;
    for (size_t p = 0;;) {
      size_t e = word.find(sep, p);
      if (e == std::string::npos) {
        parts.emplace_back(word, p);
        break;
      }
      parts.emplace_back(word, p, e - p);
      p = e + 1;
    }
    if (parts.size() != cfg.fields.size())
      return Status::InvalidParameter() << "failed to split " << word << " into " << cfg.fields.size() << " components using "
                                        << cfg.separator << " as separator word=" << word << " at line=" << w;
    std::string repr;
    bool known = true;
    for (size_t i = 0; i < parts.size(); ++i) {
      auto it = fld2pos[i].find(parts[i]);
      if (it == fld2pos[i].end()) {
        known = false;
        repr += parts[i];  // RnnReprBuilder::addString: the bytes, then varint(1)
        putVarint(repr, 1);
      } else {
        putVarint(repr, (uint32_t)it->second);
      }
    }
    (known ? knownB : unkB).add(std::move(repr), (int32_t)w);
  }
  if (eosId == -1) return Status::InvalidParameter() << "rnn dic file did not contain BOS/EOS marker (" << cfg.eosSymbol << ")";
  if (!cfg.unkSymbol.empty() && unkId == -1) return Status::InvalidParameter() << "rnn dic did not contain UNK word marker (" << cfg.unkSymbol << ")";
  if (eosId != 0) return Status::NotImplemented("we don't support if EOS/BOS token is not 0");
  JPPA_RETURN_IF_ERROR(knownB.build(&known_));
  JPPA_RETURN_IF_ERROR(unkB.build(&unk_));

  part_.has_rnn = 1;
  part_.rnn_layer_size = (uint32_t)E;
  part_.rnn_maxent_order = maxentOrder;
  part_.rnn_maxent_size = maxentSize;
  part_.rnn_vocab_size = V;
  part_.rnn_nce_constant = nceLnz;
  part_.rnn_unk_id = unkId;
  part_.rnn_num_fields = (uint32_t)cfg.fields.size();
  part_.rnn_known_index = known_.data();
  part_.rnn_known_index_bytes = known_.size() * 4;
  part_.rnn_unk_index = unk_.data();
  part_.rnn_unk_index_bytes = unk_.size() * 4;
  part_.rnn_matrix = matrix_.data();
  part_.rnn_embeddings = embeddings_.data();
  part_.rnn_nce_embeddings = nceEmbeddings_.data();
  part_.rnn_maxent = maxent_.data();
  return Status::Ok();
}

}  // namespace jumanpp_amd
