// Fully annotated training examples: the host counterpart of core::training::FullyAnnotatedExample /
// FullExampleReader (src/core/training/full_example.{h,cc}).  Two corpus formats, like the reference:
//   Morph ("DoubleCsv", the default of jumanpp_v2_train): one sentence per line, words separated by ' ', the fields of
//       a word by '_' (what `jumanpp_v2 --full-morph` prints), `# comment` to the end of the line;
//   Csv ("SimpleCsv"): one word per line, comma separated, an empty line or EOS ends the sentence.
// Every field value is looked up in the string storage of its dictionary field; a value the dictionary does not have
// becomes ~(index into the example's own strings).
#ifndef JUMANPP_AMD_HOST_TRAIN_EXAMPLE_H
#define JUMANPP_AMD_HOST_TRAIN_EXAMPLE_H

#include <cstdint>
#include <string>
#include <vector>

#include "../jpp_status.h"
#include "../partial_example.h"

namespace jumanpp_amd {
namespace train {

// ExampleNode (full_example.h:24-29)
struct GoldWord {
  std::string surface;        // the string of the LAST field of the word the dictionary did not know ("" if it knew all)
  const int32_t* data = nullptr;
  int32_t numFields = 0;
  int32_t position = 0;       // first codepoint
  int32_t length = 0;         // codepoints
};

class GoldExample {
  friend class GoldExampleReader;
  std::string surface_;
  std::vector<std::string> strings_;
  std::vector<int32_t> data_;
  std::vector<int32_t> lengths_;
  std::string comment_;
  int64_t line_ = 0;

 public:
  const std::string& surface() const { return surface_; }
  const std::string& comment() const { return comment_; }
  int64_t line() const { return line_; }
  int32_t numWords() const { return (int32_t)lengths_.size(); }
  GoldWord word(int32_t idx) const;
  void reset() {
    surface_.clear();
    strings_.clear();
    data_.clear();
    lengths_.clear();
    comment_.clear();
  }
};

enum class CorpusFormat { Csv, Morph };

// util::CsvReader as the training reader uses it (src/util/csv_reader.cc): separator-delimited fields, a field may be
// enclosed in double quotes, "" inside quotes is one quote.
struct CsvLine {
  std::vector<std::string> fields;
  bool hadQuoted = false;
  bool parse(StringPiece line, char sep);
};

class GoldExampleReader {
  const TrainFieldsIndex* tio_ = nullptr;
  int32_t surfaceColumn_ = 0;
  CorpusFormat format_ = CorpusFormat::Morph;
  StringPiece data_;
  size_t pos_ = 0;
  int64_t lineNo_ = 0;
  bool finished_ = true;
  CsvLine outer_, inner_;

  bool nextLine(StringPiece* line);
  Status addWord(const CsvLine& csv, GoldExample* result);

 public:
  void initialize(const TrainFieldsIndex* tio, int32_t surfaceColumn) {
    tio_ = tio;
    surfaceColumn_ = surfaceColumn;
  }
  void setInput(StringPiece data, CorpusFormat fmt) {
    data_ = data;
    format_ = fmt;
    pos_ = 0;
    lineNo_ = 0;
    finished_ = false;
  }
  bool finished() const { return finished_; }
  int64_t lineNumber() const { return lineNo_; }
  // FullExampleReader::readFullExample: an example with no words at the end of the input sets finished()
  Status readExample(GoldExample* result);
};

}  // namespace train
}  // namespace jumanpp_amd

#endif  // JUMANPP_AMD_HOST_TRAIN_EXAMPLE_H
