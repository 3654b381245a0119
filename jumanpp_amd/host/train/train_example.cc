#include "train_example.h"

#include <cstring>

namespace jumanpp_amd {
namespace train {

GoldWord GoldExample::word(int32_t idx) const {
  GoldWord w;
  const int32_t nf = lengths_.empty() ? 0 : (int32_t)(data_.size() / lengths_.size());
  for (int32_t i = 0; i < idx; ++i) w.position += lengths_[i];
  w.length = lengths_[idx];
  w.data = data_.data() + (size_t)idx * nf;
  w.numFields = nf;
  // FullyAnnotatedExample::nodeAt (full_example.h:53-74): the last negative value of the row names the string
  int32_t neg = 0;
  for (int32_t i = 0; i < nf; ++i)
    if (w.data[i] < 0) neg = w.data[i];
  if (neg < 0) w.surface = strings_[(size_t)~neg];
  return w;
}

bool CsvLine::parse(StringPiece line, char sep) {
  fields.clear();
  hadQuoted = false;
  const char* p = line.data();
  const char* end = p + line.size();
  if (p == end) return false;
  const char* start = p;
  bool emit = true;   // false right after a quoted field (its text was emitted already)
  for (;; ++p) {
    if (p == end) {
      if (start != p && emit) fields.emplace_back(start, p);
      break;
    }
    const char ch = *p;
    if (ch == sep) {
      if (emit) fields.emplace_back(start, p);
      start = p + 1;
      emit = true;
      continue;
    }
    if (ch == '"') {
      if (p != start) return false;   // a quote opens a field or is an error
      std::string text;
      const char* q = p + 1;
      bool closed = false;
      while (q < end) {
        if (*q == '"') {
          if (q + 1 < end && q[1] == '"') {
            text.push_back('"');
            q += 2;
            continue;
          }
          closed = true;
          break;
        }
        text.push_back(*q++);
      }
      if (!closed) return false;
      if (q + 1 != end && q[1] != sep) return false;
      fields.push_back(std::move(text));
      hadQuoted = true;
      emit = false;
      p = q;
    }
  }
  return !fields.empty();
}

bool GoldExampleReader::nextLine(StringPiece* line) {
  if (pos_ >= data_.size()) return false;
  const char* base = data_.data();
  size_t e = pos_;
  bool inQuote = false;
  // (a line break inside a quoted field belongs to the field)
  while (e < data_.size()) {
    const char c = base[e];
    if (c == '"') inQuote = !inQuote;
    if (c == '\n' && !inQuote) break;
    ++e;
  }
  size_t len = e - pos_;
  if (len > 0 && base[pos_ + len - 1] == '\r') --len;
  *line = StringPiece(base + pos_, len);
  pos_ = e < data_.size() ? e + 1 : e;
  lineNo_ += 1;
  return true;
}

namespace {
// chars::preprocessRawData: the number of codepoints of a well-formed UTF-8 string, -1 otherwise
int32_t countCodepoints(const std::string& s) {
  int32_t n = 0;
  size_t i = 0;
  while (i < s.size()) {
    const unsigned char c = (unsigned char)s[i];
    int len = c < 0x80 ? 1 : (c >> 5) == 6 ? 2 : (c >> 4) == 14 ? 3 : (c >> 3) == 30 ? 4 : 0;
    if (len == 0 || i + len > s.size()) return -1;
    for (int k = 1; k < len; ++k)
      if (((unsigned char)s[i + k] >> 6) != 2) return -1;
    i += len;
    ++n;
  }
  return n;
}
}  // namespace

// FullExampleReader::readSingleExampleFragment (full_example.cc:114-150)
Status GoldExampleReader::addWord(const CsvLine& csv, GoldExample* result) {
  const auto& fields = tio_->fields();
  if ((size_t)surfaceColumn_ >= csv.fields.size()) {
    return Status::InvalidParameter() << "a word from the line #" << lineNo_ << " has no surface column";
  }
  const std::string& surf = csv.fields[surfaceColumn_];
  const int32_t ncp = countCodepoints(surf);
  if (ncp < 0) return Status::InvalidParameter() << "invalid UTF-8 in the line #" << lineNo_;
  if (csv.fields.size() < fields.size()) {
    return Status::InvalidParameter() << "a word from the line #" << lineNo_ << " had " << csv.fields.size()
                                      << " fields, expected " << fields.size();
  }
  result->lengths_.push_back(ncp);
  result->surface_ += surf;
  for (size_t i = 0; i < fields.size(); ++i) {
    const auto& map = *fields[i].str2int;
    const std::string& v = csv.fields[i];
    auto it = map.find(v);
    if (it == map.end()) {
      result->data_.push_back(~(int32_t)result->strings_.size());
      result->strings_.push_back(v);
    } else {
      result->data_.push_back(it->second);
    }
  }
  return Status::Ok();
}

Status GoldExampleReader::readExample(GoldExample* result) {
  result->reset();
  StringPiece line;
  if (format_ == CorpusFormat::Morph) {
    // FullExampleReader::readFullExampleDblCsv (full_example.cc:76-112)
    if (!nextLine(&line)) {
      finished_ = true;
      return Status::Ok();
    }
    if (!outer_.parse(line, ' ')) {
      // (an empty line is one empty word to the reference's reader, which then fails on it; so does a malformed quote)
      return Status::InvalidParameter() << "failed to read word #0 from the line #" << lineNo_;
    }
    result->line_ = lineNo_;
    for (size_t i = 0; i < outer_.fields.size(); ++i) {
      const std::string& tok = outer_.fields[i];
      if (tok == "#") {
        for (size_t j = i + 1; j < outer_.fields.size(); ++j) {
          result->comment_ += outer_.fields[j];
          if (j + 1 != outer_.fields.size()) result->comment_.push_back(' ');
        }
        break;
      }
      if (!inner_.parse(tok, '_')) {
        return Status::InvalidParameter() << "failed to read word #" << i << " from the line #" << lineNo_;
      }
      JPPA_RETURN_IF_ERROR(addWord(inner_, result));
    }
    return Status::Ok();
  }
  // FullExampleReader::readFullExampleCsv (full_example.cc:56-74)
  while (nextLine(&line)) {
    if (result->line_ == 0) result->line_ = lineNo_;
    if (line.size() >= 2 && line[0] == '#' && line[1] == ' ') {
      result->comment_ = line.str();
      continue;
    }
    const bool parsed = outer_.parse(line, ',');
    if (!parsed || (outer_.fields.size() == 1 && (outer_.fields[0].empty() || outer_.fields[0] == "EOS"))) {
      return Status::Ok();
    }
    JPPA_RETURN_IF_ERROR(addWord(outer_, result));
  }
  finished_ = true;
  return Status::Ok();
}

}  // namespace train
}  // namespace jumanpp_amd
