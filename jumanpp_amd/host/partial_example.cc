#include "partial_example.h"

#include <cstring>

#include "output.h"

namespace jumanpp_amd {

namespace {

inline uint64_t rotl64(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }
inline uint64_t fmix64(uint64_t k) {
  k ^= k >> 33;
  k *= 0xff51afd7ed558ccdULL;
  k ^= k >> 33;
  k *= 0xc4ceb9fe1a85ec53ULL;
  k ^= k >> 33;
  return k;
}

// decodes one UTF-8 sequence; 0 on malformed input (chars::preprocessRawData, util/characters.cc:259-276)
int utf8Len(const unsigned char* p, size_t avail) {
  unsigned char c = p[0];
  int n = c < 0x80 ? 1 : (c & 0xe0) == 0xc0 ? 2 : (c & 0xf0) == 0xe0 ? 3 : (c & 0xf8) == 0xf0 ? 4 : 0;
  if (n == 0 || (size_t)n > avail) return 0;
  for (int i = 1; i < n; ++i)
    if ((p[i] & 0xc0) != 0x80) return 0;
  return n;
}

char32_t utf8Cp(const unsigned char* p, int n) {
  if (n == 1) return p[0];
  if (n == 2) return ((p[0] & 0x1f) << 6) | (p[1] & 0x3f);
  if (n == 3) return ((p[0] & 0x0f) << 12) | ((p[1] & 0x3f) << 6) | (p[2] & 0x3f);
  return ((p[0] & 0x07) << 18) | ((p[1] & 0x3f) << 12) | ((p[2] & 0x3f) << 6) | (p[3] & 0x3f);
}

}  // namespace

// murmurhash3_memory with the seed and the 0x80000000 mark of hashUnkString; the block loop mixes only the
// first 8 bytes of every 16-byte block, exactly like the reference's implementation (util/murmur_hash.cc)
int32_t hashUnkString(StringPiece sp) {
  const unsigned char* data = reinterpret_cast<const unsigned char*>(sp.data());
  const uint32_t len = (uint32_t)sp.size();
  const uint64_t C1 = 0x87c37b91114253d5ULL, C2 = 0x4cf5ad432745937fULL;
  uint64_t v1 = 0xa76210bfULL, v2 = 0xa76210bfULL;
  uint32_t nblocks = len / 16;
  for (uint32_t i = 0; i < nblocks; ++i) {
    uint64_t b1 = 0;
    for (int k = 0; k < 8; ++k) b1 |= (uint64_t)data[i * 16 + k] << (8 * k);
    uint64_t b2 = 0;
    b1 *= C1; b1 = rotl64(b1, 31); b1 *= C2;
    b2 *= C2; b2 = rotl64(b2, 33); b2 *= C1;
    v1 ^= b1; v1 = rotl64(v1, 27); v1 += v2; v1 = v1 * 5 + 0x52dce729;
    v2 ^= b2; v2 = rotl64(v2, 31); v2 += v1; v2 = v2 * 5 + 0x38495ab5;
  }
  const unsigned char* tail = data + nblocks * 16;
  uint32_t rem = len & 0xf;
  uint64_t t1 = 0, t2 = 0;
  for (uint32_t k = 0; k < rem; ++k) {
    if (k < 8) t1 ^= (uint64_t)tail[k] << (8 * k);
    else t2 ^= (uint64_t)tail[k] << (8 * (k - 8));
  }
  t1 *= C1; t1 = rotl64(t1, 31); t1 *= C2;
  t2 *= C2; t2 = rotl64(t2, 33); t2 *= C1;
  v1 ^= t1; v2 ^= t2;
  v1 ^= len; v2 ^= len;
  v1 += v2; v2 += v1;
  v1 = fmix64(v1); v2 = fmix64(v2);
  v1 += v2;
  return (int32_t)((uint32_t)v1 | 0x80000000u);
}

Status TrainFieldsIndex::initialize(const ModelImage& model) {
  fields_.clear();
  storages_.assign(model.numStringStorages(), {});
  for (auto& tf : model.trainFields()) {
    const DictionaryField* fld = model.fieldByName(tf.name);
    if (fld == nullptr || fld->stringStorage < 0) {
      return Status::InvalidState() << "training field " << tf.name << " has no string storage";
    }
    auto& map = storages_[fld->stringStorage];
    if (map.empty()) {
      // readStr2IdMap (training_io.cc:12-35): the field's empty-value marker ("*") maps to pointer 0, then
      // every string of the storage to its position (StringStorageTraversal, field_reader.h:215-239)
      if (!fld->emptyValue.empty()) map[fld->emptyValue] = 0;
      StringPiece data = model.stringStorage(fld->stringStorage);
      const uint32_t align = 1u << fld->alignPower;
      VarintReader rdr(data, 0);
      const unsigned char* base = (const unsigned char*)data.data();
      while (rdr.p < rdr.end) {
        int32_t pos = (int32_t)((size_t)(rdr.p - base) >> fld->alignPower);
        StringPiece sp;
        if (!rdr.readString(&sp)) break;
        map[sp.str()] = pos;
        size_t off = (size_t)(rdr.p - base);
        off = (off + align - 1) & ~(size_t)(align - 1);
        rdr.p = base + (off < data.size() ? off : data.size());
      }
    }
    fields_.push_back(Field{tf.name, tf.dicIdx, &map});
  }
  return Status::Ok();
}

const TrainFieldsIndex::Field* TrainFieldsIndex::byName(StringPiece name) const {
  for (auto& f : fields_)
    if (f.name.size() == name.size() && std::memcmp(f.name.data(), name.data(), name.size()) == 0) return &f;
  return nullptr;
}

Status PartialExampleReader::readExample(std::istream* stream, PartialExample* result) const {
  std::string buf, tmp;
  while (std::getline(*stream, tmp)) {
    if (tmp.size() == 0) break;
    buf.append(tmp);
    buf.push_back('\n');
  }
  return parse(buf, result);
}

Status PartialExampleReader::parse(StringPiece data, PartialExample* result) const {
  result->comment = lastComment_;
  result->surface.clear();
  result->boundaries.clear();
  result->noBreak.clear();
  result->nodes.clear();
  int32_t boundary = 2;
  int lineNo = 0;
  size_t pos = 0;
  std::vector<StringPiece> fields;
  while (pos < data.size()) {
    size_t eol = pos;
    while (eol < data.size() && data[eol] != '\n') ++eol;
    StringPiece line(data.data() + pos, eol - pos);
    pos = eol + 1;
    ++lineNo;
    if (line.size() >= 2 && line[0] == '#' && line[1] == ' ') {  // "# " + comment (+ '\n' in the reference's view)
      result->comment.assign(line.data() + 2, line.size() - 2);
      lastComment_ = result->comment;
      continue;
    }
    // util::CsvReader{'\t', '\0'}: tab separated, no quoting
    fields.clear();
    size_t fs = 0;
    for (size_t i = 0; i <= line.size(); ++i) {
      if (i == line.size() || line[i] == '\t') {
        fields.push_back(StringPiece(line.data() + fs, i - fs));
        fs = i + 1;
      }
    }
    auto decode = [&](StringPiece sp, std::vector<std::pair<char32_t, StringPiece>>* cps) -> bool {
      const unsigned char* p = (const unsigned char*)sp.data();
      size_t i = 0;
      while (i < sp.size()) {
        int n = utf8Len(p + i, sp.size() - i);
        if (n == 0) return false;
        cps->push_back({utf8Cp(p + i, n), StringPiece(sp.data() + i, (size_t)n)});
        i += (size_t)n;
      }
      return true;
    };
    std::vector<std::pair<char32_t, StringPiece>> cps;
    if (fields.size() == 1) {
      StringPiece d = fields[0];
      if (d.empty()) {
        if (!result->boundaries.empty()) result->boundaries.pop_back();
        return Status::Ok();
      }
      if (!decode(d, &cps)) return Status::InvalidParameter() << "Invalid UTF8 sequence: " << d << "at <memory>:" << lineNo;
      for (auto& c : cps) {
        if (c.first == noBreakToken_) {
          result->noBreak.push_back(boundary);
        } else {
          result->surface.append(c.second.data(), c.second.size());
          boundary += 1;
        }
      }
      result->boundaries.push_back(boundary);
      continue;
    }
    if (!fields[0].empty()) {
      return Status::InvalidParameter() << "in file: <memory>:" << lineNo << " first field was not empty, but" << fields[0];
    }
    NodeConstraint nc;
    StringPiece surface = fields[1];
    if (!decode(surface, &cps)) return Status::InvalidParameter() << "Invalid UTF8 sequence: " << surface << " at <memory>:" << lineNo;
    nc.surface = surface.str();
    nc.length = (int32_t)cps.size();
    for (int i = 1; i < nc.length; ++i) result->noBreak.push_back(boundary + i);
    nc.boundary = boundary;
    boundary += nc.length;
    result->surface.append(nc.surface);
    result->boundaries.push_back(boundary);
    for (size_t idx = 2; idx < fields.size(); ++idx) {
      StringPiece fd = fields[idx];
      size_t colon = 0;
      while (colon < fd.size() && fd[colon] != ':') ++colon;
      if (colon == fd.size()) {
        return Status::InvalidParameter() << "in file: <memory>:" << lineNo << " an entry [" << fd
                                          << "] did not contain field name (<name>:<value>)";
      }
      StringPiece name(fd.data(), colon), value(fd.data() + colon + 1, fd.size() - colon - 1);
      cps.clear();
      if (!decode(value, &cps)) return Status::InvalidParameter() << "Invalid UTF8 sequence: " << value << " at <memory>:" << lineNo;
      const TrainFieldsIndex::Field* f = tio_->byName(name);
      if (f == nullptr) {
        return Status::InvalidParameter() << "in file: <memory>:" << lineNo << " the field name of an entry [" << fd
                                          << "] was not present in the dictionary spec";
      }
      auto it = f->str2int->find(value.str());
      nc.tags.push_back(TagConstraint{f->dicFieldIdx, it == f->str2int->end() ? hashUnkString(value) : it->second});
    }
    result->nodes.push_back(std::move(nc));
  }
  return Status::Ok();
}

void PartialBatch::build(const std::vector<const PartialExample*>& examples) {
  nobreakOff.assign(1, 0);
  boundaryOff.assign(1, 0);
  nodeOff.assign(1, 0);
  nobreak.clear();
  boundaries.clear();
  nodes.clear();
  tags.clear();
  for (const PartialExample* e : examples) {
    if (e != nullptr) {
      for (auto b : e->noBreak) nobreak.push_back((uint16_t)b);
      for (auto b : e->boundaries) boundaries.push_back((uint16_t)b);
      for (auto& n : e->nodes) {
        jppgpu_node_constraint c{(uint16_t)n.boundary, (uint16_t)n.length, (uint32_t)tags.size(), (uint32_t)n.tags.size()};
        for (auto& t : n.tags) tags.push_back(jppgpu_tag_constraint{t.field, t.value});
        nodes.push_back(c);
      }
    }
    nobreakOff.push_back((uint32_t)nobreak.size());
    boundaryOff.push_back((uint32_t)boundaries.size());
    nodeOff.push_back((uint32_t)nodes.size());
  }
  view.nobreak_offsets = nobreakOff.data();
  view.nobreak = nobreak.data();
  view.boundary_offsets = boundaryOff.data();
  view.boundaries = boundaries.data();
  view.node_offsets = nodeOff.data();
  view.nodes = nodes.data();
  view.tags = tags.data();
  view.num_tags = (uint32_t)tags.size();
}

}  // namespace jumanpp_amd
