// jumanpp_gpu: the reference CLI's read-analyze-format loop
// (src/jumandic/main/jumanpp.cc:100-182, PlainStreamReader
// src/core/input/stream_reader.cc:12-38) with the per-line Analyzer::analyze
// replaced by batched GpuAnalyzer::analyzeBatch.  Output is byte-identical to
// `jumanpp_v2 --model=... ` in the JUMAN format.
//
// usage: jumanpp_gpu --model=MODEL.img [--beam=5] [--global-beam=6] [--right-check=1]
//                    [--right-beam=5] [--no-rnn] [--batch=65536] [--device=0] [-o OUT] [INPUT...]
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <memory>
#include <string>
#include <vector>

#include "gpu_analyzer.h"
#include "juman_format.h"
#include "lattice_format.h"
#include "simple_formats.h"

using namespace jumanpp_amd;

namespace {

struct Conf {
  std::string model;
  int beam = 5, globalBeam = 6, rightCheck = 1, rightBeam = 5;  // jumanpp_args.h:50-54
  bool noRnn = false;
  size_t batch = 65536;
  int device = 0;
  std::string output;
  std::vector<std::string> inputs;
  bool timing = false;
  int lattice = 0;  // -s N / --lattice N / --specifics N: LatticeFormat with the N best paths; -1 = beam width
  enum { Juman, Morph, FullMorph, Segment } kind = Juman;
  std::string segmentSeparator = " ";
  bool partialInput = false;  // --partial-input: InputType::PartiallyAnnotated
  int autoStep = 0;           // --auto-nbest=base:step:max (jumanpp_args.cc:270-279)
};

bool argValue(int argc, const char** argv, int& i, const char* name, std::string* out) {
  size_t n = std::strlen(name);
  const char* a = argv[i];
  if (std::strncmp(a, name, n) != 0) return false;
  if (a[n] == '=') {
    *out = a + n + 1;
    return true;
  }
  if (a[n] == 0 && i + 1 < argc) {
    *out = argv[++i];
    return true;
  }
  if (n == 2 && a[n] != 0) {  // -s5
    *out = a + n;
    return true;
  }
  return false;
}

struct Example {
  std::string comment;
  std::string input;
  Status readStatus;
  PartialExample partial;
};

}  // namespace

int main(int argc, const char** argv) {
  Conf conf;
  for (int i = 1; i < argc; ++i) {
    std::string v;
    if (argValue(argc, argv, i, "--model", &v)) conf.model = v;
    else if (argValue(argc, argv, i, "--beam", &v)) conf.beam = std::atoi(v.c_str());
    else if (argValue(argc, argv, i, "--global-beam", &v)) conf.globalBeam = std::atoi(v.c_str());
    else if (argValue(argc, argv, i, "--right-check", &v)) conf.rightCheck = std::atoi(v.c_str());
    else if (argValue(argc, argv, i, "--right-beam", &v)) conf.rightBeam = std::atoi(v.c_str());
    else if (argValue(argc, argv, i, "--batch", &v)) conf.batch = (size_t)std::atoll(v.c_str());
    else if (argValue(argc, argv, i, "--device", &v)) conf.device = std::atoi(v.c_str());
    else if (argValue(argc, argv, i, "--output", &v) || argValue(argc, argv, i, "-o", &v)) conf.output = v;
    else if (argValue(argc, argv, i, "--lattice", &v) || argValue(argc, argv, i, "--specifics", &v) ||
             argValue(argc, argv, i, "-s", &v) || argValue(argc, argv, i, "-L", &v)) conf.lattice = std::atoi(v.c_str());
    else if (argValue(argc, argv, i, "--segment-separator", &v)) conf.segmentSeparator = v;
    else if (std::strcmp(argv[i], "--segment") == 0) conf.kind = Conf::Segment;
    else if (std::strcmp(argv[i], "--morph") == 0 || std::strcmp(argv[i], "-M") == 0) conf.kind = Conf::Morph;
    else if (std::strcmp(argv[i], "--full-morph") == 0 || std::strcmp(argv[i], "-F") == 0) conf.kind = Conf::FullMorph;
    else if (std::strcmp(argv[i], "--juman") == 0 || std::strcmp(argv[i], "-j") == 0) conf.kind = Conf::Juman;
    else if (argValue(argc, argv, i, "--auto-nbest", &v)) {
      // ^(\d+):(\d+):(\d+)$ -> beamSize, autoStep, globalBeam; anything else is ignored like the reference does
      int a = 0, b = 0, c = 0;
      char tail = 0;
      if (std::sscanf(v.c_str(), "%d:%d:%d%c", &a, &b, &c, &tail) == 3 && a >= 0 && b >= 0 && c >= 0) {
        conf.beam = a;
        conf.autoStep = b;
        conf.globalBeam = c;
      }
    } else if (std::strcmp(argv[i], "--partial-input") == 0) conf.partialInput = true;
    else if (std::strcmp(argv[i], "--no-rnn") == 0) conf.noRnn = true;
    else if (std::strcmp(argv[i], "--timing") == 0) conf.timing = true;
    else if (argv[i][0] == '-' && argv[i][1] != 0) {
      std::cerr << "unknown option " << argv[i] << "\n";
      return 1;
    } else conf.inputs.push_back(argv[i]);
  }
  if (conf.model.empty()) {
    std::cerr << "Model file was not specified\n";
    return 1;
  }
  ModelImage model;
  Status s = model.loadModel(conf.model);
  if (!s) {
    std::cerr << "failed to load model from disk: " << s << "\n";
    return 1;
  }
  // JumanppEnv::makeAnalyzer: the RNN scorer is used whenever the model carries one (env.cc:86-121)
  AnalyzerConfig acfg;
  acfg.globalBeamSize = conf.globalBeam;
  acfg.rightGbeamCheck = conf.rightCheck;
  acfg.rightGbeamSize = conf.rightBeam;
  if (conf.autoStep > 0) {  // env.setAutoBeam(conf.beamSize, conf.autoStep, conf.globalBeam), jumandic_env.cc:34-36
    acfg.autoBeamBase = conf.beam;
    acfg.autoBeamStep = conf.autoStep;
    acfg.autoBeamMax = conf.globalBeam;
  }
  ScoringConfig sconf;
  sconf.beamSize = conf.beam;
  ScorerDef def;
  def.useRnn = model.hasRnn() && !conf.noRnn;
  if (def.useRnn) {
    sconf.numScorers = 2;
    def.scoreWeights = {model.savedScoreWeights().perceptron, model.savedScoreWeights().rnn};
  } else {
    sconf.numScorers = 1;
    def.scoreWeights = {1.0f};
  }
  GpuAnalyzer analyzer;
  s = analyzer.initialize(&model, acfg, sconf, &def, conf.device);
  if (!s) {
    std::cerr << "failed to initialize the analyzer: " << s << "\n";
    return 1;
  }
  // JumanppExec::initOutput (jumandic_env.cc:55-150) and emptyResult (:211-222)
  std::unique_ptr<OutputFormat> format;
  StringPiece emptyResult = "# ERROR\nEOS\n";
  const bool useLattice = conf.lattice != 0;
  if (useLattice) {
    auto f = new LatticeFormat(conf.lattice == -1 ? conf.beam : conf.lattice);
    format.reset(f);
    s = f->initialize(&model, def.scoreWeights);
  } else if (conf.kind == Conf::Morph || conf.kind == Conf::FullMorph) {
    auto f = new MorphFormat(conf.kind == Conf::FullMorph);
    format.reset(f);
    s = f->initialize(&model);
    emptyResult = "# ERROR\n";
  } else if (conf.kind == Conf::Segment) {
    auto f = new SegmentedFormat();
    format.reset(f);
    s = f->initialize(&model, conf.segmentSeparator);
    emptyResult = "";
  } else {
    auto f = new JumanFormat();
    format.reset(f);
    s = f->initialize(&model);
  }
  if (!s) {
    std::cerr << "Failed to initialize I/O: " << s << "\n";
    return 1;
  }

  std::unique_ptr<std::ofstream> ofile;
  std::ostream* out = &std::cout;
  if (!conf.output.empty() && conf.output != "-") {
    ofile.reset(new std::ofstream(conf.output, std::ios::binary));
    out = ofile.get();
  }
  std::ios::sync_with_stdio(false);

  size_t fileIdx = 0;
  std::unique_ptr<std::ifstream> ifile;
  std::istream* in = &std::cin;
  auto openNext = [&]() -> bool {
    if (fileIdx >= conf.inputs.size()) return false;
    ifile.reset(new std::ifstream(conf.inputs[fileIdx++], std::ios::binary));
    in = ifile.get();
    return true;
  };
  if (!conf.inputs.empty()) openNext();
  // InputOutput::hasNext (jumanpp.cc:82-97)
  auto hasNext = [&]() -> bool {
    for (;;) {
      if (in->good() && in->peek() != std::char_traits<char>::eof()) return true;
      if (conf.inputs.empty() || !openNext()) return false;
    }
  };

  const size_t maxInput = 65535, maxComment = 1024;  // rdr->setMaxSizes(65535, 1024), jumanpp.cc:72
  TrainFieldsIndex tfi;
  PartialExampleReader pexReader;
  if (conf.partialInput) {  // PexStreamReader::initialize(core, '&'), jumanpp.cc:74-77
    s = tfi.initialize(model);
    if (s) s = pexReader.initialize(&tfi, U'&');
    if (!s) {
      std::cerr << "Failed to initialize I/O: " << s << "\n";
      return 1;
    }
  }
  std::vector<Example> batch;
  std::vector<StringPiece> pieces;
  int result = 0;
  double gpuMs = 0;
  size_t sentences = 0;
  auto flush = [&]() {
    Status bs;
    if (conf.partialInput) {
      std::vector<const PartialExample*> exs;
      for (auto& e : batch) exs.push_back(e.readStatus.isOk() ? &e.partial : nullptr);
      bs = analyzer.analyzeBatchPartial(exs, useLattice);
    } else {
      pieces.clear();
      for (auto& e : batch) pieces.push_back(e.readStatus.isOk() ? StringPiece(e.input) : StringPiece(""));
      bs = analyzer.analyzeBatch(pieces, useLattice);
    }
    if (conf.timing) {
      float ms[8];
      analyzer.lastTimings(ms);
      gpuMs += ms[7];
    }
    for (size_t i = 0; i < batch.size(); ++i) {
      if (!batch[i].readStatus.isOk()) {
        std::cerr << "failed to read an example: " << batch[i].readStatus;
        result = 1;
        continue;
      }
      result = 0;
      Status st = bs.isOk() ? analyzer.sentenceStatus(i) : bs;
      if (!st) {
        std::cerr << st;
        *out << emptyResult;
        continue;
      }
      StringPiece comment = batch[i].comment.size() < 2 ? StringPiece("") : StringPiece(batch[i].comment.data() + 2, batch[i].comment.size() - 2);
      if (conf.partialInput) comment = StringPiece(batch[i].partial.comment);
      st = format->format(analyzer, i, comment);
      if (!st) std::cerr << st;
      else *out << format->result();
    }
    sentences += batch.size();
    batch.clear();
  };
  while (hasNext()) {
    Example e;
    if (conf.partialInput) {
      e.readStatus = pexReader.readExample(in, &e.partial);
      batch.push_back(std::move(e));
      if (batch.size() >= conf.batch) flush();
      continue;
    }
    // PlainStreamReader::readExample
    for (;;) {
      e.input.clear();
      std::getline(*in, e.input);
      if (e.input.size() > 2 && e.input[0] == '#' && e.input[1] == ' ') std::swap(e.comment, e.input);
      else break;
    }
    if (e.comment.size() > maxComment) {
      e.readStatus = Status::InvalidParameter() << "Comment size was: " << e.comment.size() << " which is more than max: " << maxComment;
    } else if (e.input.size() > maxInput) {
      e.readStatus = Status::InvalidParameter() << "Input size was: " << e.input.size() << " which is more than max: " << maxInput;
    }
    batch.push_back(std::move(e));
    if (batch.size() >= conf.batch) flush();
  }
  if (!batch.empty()) flush();
  out->flush();
  if (conf.timing) std::cerr << "sentences=" << sentences << " gpu_ms=" << gpuMs << "\n";
  return result;
}
