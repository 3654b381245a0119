// TEST INFRASTRUCTURE ONLY (oracle).  NOT part of the product.
//
// Golden-vector dumper / model exporter linked against the *real* reference
// (oracle/_ref/libjpp_ref.a, built by oracle/Makefile from /root/reference).
// Everything here only *calls* reference APIs; no reference source is copied.
//
//   ref_dump export  <model.jppmdl> <out.img>
//       flat "device model image" consumed by jumanpp_amd (see
//       jumanpp_amd/csrc/model_image.h for the section tags)
//   ref_dump mkmodel <dic-only.jppmdl> <out.jppmdl> <sizeExp> <seed> <sigma>
//       attach a random N(0, sigma) perceptron of 2^sizeExp weights
//   ref_dump dump    <model.jppmdl> <out.gold> [beam gbeam rcheck rbeam] < corpus
//       per-sentence golden vectors: seeds/nodes, entry rows, 14 stored
//       patterns, T0 scores, global beams, per-node beams, score cells, top-1
//       path and the Juman-format text.  Drives the same stage sequence as
//       AnalyzerImpl::computeScoresGbeam (src/core/analysis/analyzer_impl.cc:250-297)
//       through the reference's own ScoreProcessor so T0 can be captured per
//       boundary.
//   ref_dump shim    <model.jppmdl> <libjppgpu(.so|_emu.so)> <lattice-N|0> [beam gbeam rcheck rbeam] < corpus
//       the drop-in claim of SURVEY 8(b), executed: the model is handed to the C ABI from the reference's own
//       structures, the batch is analysed by the library, and for every sentence a
//       reference Lattice is re-materialised from the result view inside an ordinary reference Analyzer (seeds
//       from the device node table through the reference's LatticeBuilder, beams and score cells written with
//       host pointers rebuilt from the index form).  The reference's UNMODIFIED
//       JumanFormat / LatticeFormat then format that analyzer; the bytes must equal those of a plain
//       Analyzer::analyze run.  Prints one JSON line, exit code 0 iff everything is identical.
//   ref_dump ngrams  <model.jppmdl> <out.bin> [beam gbeam rcheck rbeam] < corpus
//       the trainer's read-out: NgramFeaturesComputer::calculateNgramFeatures for every connection of the top-1
//       path on an analyzer that stores all patterns (what jppgpu_result_fetch_top1_ngrams must reproduce)
//   ref_dump bootstrapv <dict.mdic> <out.jppmdl> <drop|add|len|cols>
//       jpp_jumandic_bootstrap with a VARIANT of the jumandic spec, so that the spec hash no longer matches the
//       reference's generated static feature code and the reference runs its dynamic feature objects
//       (features_api.cc:20-60): `drop` removes the last n-gram feature of the spec, `add` appends a unigram, a
//       bigram and swaps two bigrams, `len` adds a unigram over three LENGTH primitives (byte length / codepoints of dictionary
//       strings, and of a column UNK makers overwrite), `cols` adds four dictionary columns (12 feature columns per entry row) with
//       unigram features over them.  The checker of the table-driven kernels.
//   ref_dump top1    <model.jppmdl> <out.bin> [beam gbeam rcheck rbeam] < corpus
//       the packed top-1 result of Analyzer::analyze per sentence: u32 status (0 ok / 1 failed), u32 count, then count
//       records {i32 EntryPtr raw, u16 start, u16 end} in text order (EOS dropped) -- the layout of jppgpu_result_pack.
//       UNK nodes carry the reference's own EntryPtr (creation-order numbering, extra_nodes.cc:41-52).  Checker of
//       the at-scale parity tests and of bench.py's parity_sample.
//   ref_dump top1x   <model.jppmdl> <out.bin> <scorer-weight|none> <plugin 0|1> [beam gbeam rcheck rbeam] < corpus
//       `top1` with a TEST ScoreComputer appended to ScorerDef::others (weight given) and / or a TEST per-connection
//       ScorePlugin passed to Analyzer::analyze; also writes <out.bin>.totals (EOS beam totals).  Checker of
//       jppgpu_analyze_batch_scored / jppgpu_analyze_batch_pairs.
//   ref_dump time    <model.jppmdl> [beam gbeam rcheck rbeam] < corpus
//       wall-clock of Analyzer::analyze (+JumanFormat) over the corpus, phases split
//       as BASELINE.md section 3.
#include <chrono>
#include <dlfcn.h>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iostream>
#include <random>
#include <string>
#include <vector>

#include "core/analysis/analyzer_impl.h"
#include "core/analysis/perceptron.h"
#include "core/analysis/rnn_scorer.h"
#include "core/analysis/rnn_scorer_gbeam.h"
#include "rnn/mikolov_rnn.h"
#include "util/cfg.h"
#include "core/analysis/score_processor.h"
#include "core/analysis/unk_nodes_creator.h"
#include "core/core.h"
#include "core/dic/dic_builder.h"
#include "core/dic/dictionary.h"
#include "core/env.h"
#include "core/impl/feature_computer.h"
#include "core/impl/feature_impl_types.h"
#include "core/impl/model_io.h"
#include "core/impl/perceptron_io.h"
#include "jpp_jumandic_cg.h"
#include "jumandic/shared/jumandic_spec.h"
#include "util/mmap.h"
#include "jumandic/shared/juman_format.h"
#include "jumandic/shared/lattice_format.h"
#include "../include/jppgpu.h"
#include "util/serialization.h"

using namespace jumanpp;
using namespace jumanpp::core;
using namespace jumanpp::core::analysis;

#define CHECK_OK(expr)                                             \
  do {                                                             \
    Status _s = (expr);                                            \
    if (!_s) {                                                     \
      std::cerr << "FAILED: " #expr << " : " << _s << "\n";        \
      std::exit(1);                                                \
    }                                                              \
  } while (0)

namespace {

struct Writer {
  std::vector<char> buf;
  template <typename T>
  void put(const T& v) {
    const char* p = reinterpret_cast<const char*>(&v);
    buf.insert(buf.end(), p, p + sizeof(T));
  }
  void bytes(const void* p, size_t n) {
    const char* c = reinterpret_cast<const char*>(p);
    buf.insert(buf.end(), c, c + n);
  }
  void align8() {
    while (buf.size() % 8) buf.push_back(0);
  }
  void save(const char* name) {
    std::ofstream f(name, std::ios::binary);
    f.write(buf.data(), buf.size());
    if (!f) {
      std::cerr << "could not write " << name << "\n";
      std::exit(1);
    }
  }
};

// RNN part header: same field order as the (file-local) RnnModelHeader of
// src/core/analysis/rnn_scorer_gbeam.cc:353-373, declared here because that
// type is not visible outside its translation unit.
struct OracleRnnHeader {
  core::analysis::rnn::RnnInferenceConfig config;
  i32 unkIdx = 0;
  std::vector<u32> fields;
  jumanpp::rnn::mikolov::MikolovRnnModelHeader rnnHeader{};
};

template <typename Arch>
void Serialize(Arch& a, OracleRnnHeader& o) {
  a& o.config.nceBias;
  a& o.config.unkConstantTerm;
  a& o.config.unkLengthPenalty;
  a& o.config.perceptronWeight;
  a& o.config.rnnWeight;
  a& o.config.eosSymbol;
  a& o.config.unkSymbol;
  a& o.config.rnnFields;
  a& o.config.fieldSeparator;
  a& o.unkIdx;
  a& o.fields;
  a& o.rnnHeader.layerSize;
  a& o.rnnHeader.maxentOrder;
  a& o.rnnHeader.maxentSize;
  a& o.rnnHeader.vocabSize;
  a& o.rnnHeader.nceLnz;
}

// ---------------------------------------------------------------- export ---
// section tags, must match jumanpp_amd/csrc/model_image.h
enum : u32 {
  SEC_INFO = 1,
  SEC_TRIE = 2,
  SEC_ENTRY_PTRS = 3,
  SEC_ENTRY_DATA = 4,
  SEC_WEIGHTS = 5,
  SEC_UNK = 6,
  SEC_FEATURES = 7,
  SEC_FIELDS = 8,
  SEC_STRINGS = 9,   // one per string storage
  SEC_INTS = 10,     // one per int storage
  SEC_RNN = 11,
  SEC_IDMAP = 12,    // aux 0: (pos, subpos) -> JUMAN ids, aux 1: (conjtype, conjform) -> JUMAN ids
  SEC_TRAIN = 13,    // training fields (partial-annotation tags): entry-row column + dictionary field name
};

void section(Writer& w, u32 tag, u32 aux, const void* data, u64 size) {
  w.align8();
  w.put<u32>(tag);
  w.put<u32>(aux);
  w.put<u64>(size);
  w.bytes(data, size);
  w.align8();
}

void putInts(Writer& w, const std::vector<i32>& v) {
  w.put<i32>((i32)v.size());
  for (auto x : v) w.put<i32>(x);
}

int doExport(const char* modelFile, const char* out) {
  std::string modelS{modelFile};
  model::FilesystemModel fs;
  CHECK_OK(fs.open(StringPiece{modelS}));
  model::ModelInfo info;
  CHECK_OK(fs.load(&info));
  dic::BuiltDictionary bd;
  CHECK_OK(bd.restoreDictionary(info));
  dic::DictionaryHolder holder;
  CHECK_OK(holder.load(bd));
  auto& spec = bd.spec;

  Writer w;
  w.bytes("JPPGPUI1", 8);

  {  // INFO
    Writer s;
    s.put<i32>(spec.features.numDicFeatures);
    s.put<i32>(spec.features.numDicData);
    s.put<i32>(spec.features.numPlaceholders);
    s.put<i32>(bd.entryCount);
    s.put<i32>((i32)spec.features.pattern.size());
    s.put<i32>(spec.features.numUniOnlyPats);
    s.put<i32>((i32)bd.stringStorages.size());
    s.put<i32>((i32)bd.intStorages.size());
    section(w, SEC_INFO, 0, s.buf.data(), s.buf.size());
  }
  section(w, SEC_TRIE, 0, bd.trieContent.data(), bd.trieContent.size());
  section(w, SEC_ENTRY_PTRS, 0, bd.entryPointers.data(), bd.entryPointers.size());
  section(w, SEC_ENTRY_DATA, 0, bd.entryData.data(), bd.entryData.size());

  if (auto pp = info.firstPartOf(model::ModelPartKind::Perceprton)) {
    util::serialization::Loader ldr{pp->data[0]};
    PerceptronInfo pi{};
    if (!ldr.load(&pi)) {
      std::cerr << "bad perceptron header\n";
      return 1;
    }
    section(w, SEC_WEIGHTS, (u32)pi.modelSizeExponent, pp->data[1].data(),
            pp->data[1].size());
  }

  {  // UNK makers, in spec order (src/core/analysis/unk_nodes.cc:39-95)
    Writer s;
    s.put<i32>((i32)spec.unkCreators.size());
    for (auto& u : spec.unkCreators) {
      s.put<i32>((i32)u.type);
      s.put<i32>((i32)u.charClass);
      s.put<i32>(u.patternPtr);
      s.put<i32>(u.priority);
      s.put<i32>(u.features.empty() ? -1 : u.features[0].targetPlaceholder);
      s.put<i32>((i32)u.replaceFields.size());
      for (auto f : u.replaceFields) s.put<i32>(f);
    }
    section(w, SEC_UNK, 0, s.buf.data(), s.buf.size());
  }

  {  // feature descriptors (src/core/spec/spec_types.h FeaturesSpec)
    Writer s;
    auto& fs_ = spec.features;
    s.put<i32>((i32)fs_.primitive.size());
    for (auto& p : fs_.primitive) {
      s.put<i32>((i32)p.kind);
      putInts(s, p.references);
    }
    s.put<i32>((i32)fs_.computation.size());
    for (auto& c : fs_.computation) {
      s.put<i32>(c.primitiveFeature);
      putInts(s, c.trueBranch);
      putInts(s, c.falseBranch);
    }
    s.put<i32>((i32)fs_.pattern.size());
    for (auto& p : fs_.pattern) {
      s.put<i32>(p.index);
      putInts(s, p.references);
    }
    s.put<i32>((i32)fs_.ngram.size());
    for (auto& n : fs_.ngram) {
      s.put<i32>(n.index);
      putInts(s, n.references);
    }
    section(w, SEC_FEATURES, 0, s.buf.data(), s.buf.size());
  }

  {  // dictionary fields (for output formatting: src/core/analysis/output.cc)
    Writer s;
    s.put<i32>((i32)bd.fieldData.size());
    for (auto& f : bd.fieldData) {
      auto& sf = spec.dictionary.fields.at(f.specIndex);
      s.put<i32>(f.dicIndex);
      s.put<i32>(f.specIndex);
      s.put<i32>((i32)sf.fieldType);
      s.put<i32>(sf.stringStorage);
      s.put<i32>(sf.intStorage);
      s.put<i32>(sf.alignment);
      s.put<i32>(sf.isTrieKey ? 1 : 0);
      s.put<i32>((i32)sf.name.size());
      s.bytes(sf.name.data(), sf.name.size());
      s.put<i32>((i32)sf.emptyString.size());
      s.bytes(sf.emptyString.data(), sf.emptyString.size());
      s.align8();
    }
    section(w, SEC_FIELDS, 0, s.buf.data(), s.buf.size());
  }
  for (size_t i = 0; i < bd.stringStorages.size(); ++i) {
    section(w, SEC_STRINGS, (u32)i, bd.stringStorages[i].data(),
            bd.stringStorages[i].size());
  }
  for (size_t i = 0; i < bd.intStorages.size(); ++i) {
    section(w, SEC_INTS, (u32)i, bd.intStorages[i].data(),
            bd.intStorages[i].size());
  }
  {  // JUMAN grammar ids used by the Juman output format: the resolved maps of
     // JumandicIdResolver (src/jumandic/shared/jumandic_id_resolver.cc:32-88), enumerated over
     // every string pointer the four fields can take
    jumandic::JumandicIdResolver res;
    auto st = res.initialize(holder);
    if (st.isOk()) {
      auto positions = [&](const char* name) {
        std::vector<i32> r{0};
        auto fld = holder.fieldByName(StringPiece{name, std::strlen(name)});
        dic::impl::StringStorageTraversal trav(fld->strings);
        StringPiece sp;
        while (trav.next(&sp)) r.push_back(trav.position());
        return r;
      };
      auto pos = positions("pos"), sub = positions("subpos"), ct = positions("conjtype"), cf = positions("conjform");
      Writer a, b;
      i32 na = 0, nb2 = 0;
      for (auto p : pos)
        for (auto q : sub) {
          auto r = res.dicToJuman(jumandic::JumandicPosId{p, q, 0, 0});
          if (r.pos != 0 || r.subpos != 0) {
            a.put<i32>(p); a.put<i32>(q); a.put<i32>(r.pos); a.put<i32>(r.subpos);
            ++na;
          }
        }
      for (auto p : ct)
        for (auto q : cf) {
          auto r = res.dicToJuman(jumandic::JumandicPosId{0, 0, p, q});
          if (r.conjType != 0 || r.conjForm != 0) {
            b.put<i32>(p); b.put<i32>(q); b.put<i32>(r.conjType); b.put<i32>(r.conjForm);
            ++nb2;
          }
        }
      section(w, SEC_IDMAP, 0, a.buf.data(), a.buf.size());
      section(w, SEC_IDMAP, 1, b.buf.data(), b.buf.size());
      (void)na; (void)nb2;
    }
  }
  {  // spec.training.fields as TrainFieldsIndex::initialize reads them (src/core/input/training_io.cc:37-55)
    Writer s;
    s.put<i32>((i32)spec.training.fields.size());
    for (auto& tf : spec.training.fields) {
      auto& fldSpec = spec.dictionary.fields[tf.fieldIdx];
      s.put<i32>(tf.dicIdx);
      s.put<i32>((i32)fldSpec.name.size());
      s.bytes(fldSpec.name.data(), fldSpec.name.size());
      s.align8();
    }
    section(w, SEC_TRAIN, 0, s.buf.data(), s.buf.size());
  }
  if (auto rp = info.firstPartOf(model::ModelPartKind::Rnn)) {
    // RNN part blocks verbatim (src/core/analysis/rnn_scorer_gbeam.cc:375-398,426-470)
    for (size_t i = 0; i < rp->data.size(); ++i) {
      section(w, SEC_RNN, (u32)i, rp->data[i].data(), rp->data[i].size());
    }
    // decoded parameters (aux = 100).  The effective NCE constant follows
    // RnnScorerGbeamFactory::load (rnn_scorer_gbeam.cc:426-470): nceLnz, replaced by
    // rnnWeight when that is defined (:465-467) -- no CLI override is applied here.
    OracleRnnHeader h;
    util::serialization::Loader l{rp->data[0]};
    if (!l.load(&h)) {
      std::cerr << "bad rnn header\n";
      return 1;
    }
    float nce = h.rnnHeader.nceLnz;
    if (h.config.rnnWeight.defined()) nce = h.config.rnnWeight;
    Writer s;
    s.put<u32>(h.rnnHeader.layerSize);
    s.put<u32>(h.rnnHeader.maxentOrder);
    s.put<u64>(h.rnnHeader.maxentSize);
    s.put<u64>(h.rnnHeader.vocabSize);
    s.put<float>(nce);
    s.put<i32>(h.unkIdx);
    s.put<float>(h.config.unkConstantTerm);
    s.put<float>(h.config.unkLengthPenalty);
    s.put<float>(h.config.perceptronWeight);
    s.put<float>(h.config.rnnWeight);
    s.put<u32>((u32)h.fields.size());
    for (auto f : h.fields) s.put<u32>(f);
    section(w, SEC_RNN, 100, s.buf.data(), s.buf.size());
  }
  w.align8();
  w.put<u32>(0);
  w.put<u32>(0);
  w.put<u64>(0);
  w.save(out);
  std::cerr << "exported " << w.buf.size() << " bytes to " << out << "\n";
  return 0;
}

// --------------------------------------------------------------- mkmodel ---
int doMkModel(const char* in, const char* out, int sizeExp, u64 seed,
              float sigma) {
  std::string inS{in}, outS{out};
  model::FilesystemModel fs;
  CHECK_OK(fs.open(StringPiece{inS}));
  model::ModelInfo info;
  CHECK_OK(fs.load(&info));
  model::ModelInfo result;
  for (auto& p : info.parts) {
    if (p.kind == model::ModelPartKind::Dictionary) result.parts.push_back(p);
  }
  std::vector<float> weights(size_t{1} << sizeExp);
  std::mt19937_64 rng{seed};
  std::normal_distribution<float> nd{0.f, sigma};
  for (auto& x : weights) x = nd(rng);

  PerceptronInfo pi{};
  pi.modelSizeExponent = sizeExp;
  util::serialization::Saver sv;
  sv.save(pi);
  model::ModelPart part;
  part.kind = model::ModelPartKind::Perceprton;
  part.comment = "random perceptron (oracle/ref_dump mkmodel)";
  part.data.push_back(sv.result());
  part.data.push_back(
      StringPiece{reinterpret_cast<const char*>(weights.data()),
                  reinterpret_cast<const char*>(weights.data() + weights.size())});
  result.parts.push_back(part);
  for (auto& p : info.parts) {
    if (p.kind == model::ModelPartKind::Rnn) result.parts.push_back(p);
  }
  model::ModelSaver saver;
  CHECK_OK(saver.open(StringPiece{outS}));
  CHECK_OK(saver.save(result));
  return 0;
}

// ------------------------------------------------------------------ dump ---
struct DumpAnalyzer : public AnalyzerImpl {
  DumpAnalyzer(const CoreHolder* core, const ScoringConfig& sconf,
               const AnalyzerConfig& cfg)
      : AnalyzerImpl(core, sconf, cfg) {}
  ScoreProcessor& sproc() { return *sproc_; }
  AnalysisInput& inputRef() { return input_; }
  LatticeConfig& lcfg() { return latticeConfig_; }
  size_t numExtraScorers() const { return scorers_.size(); }
  Status runExtraScorers(const ScorerDef* sconf) {
    if (!scorers_.empty()) {
      u32 idx = 1;
      for (auto& s : scorers_) {
        JPP_RETURN_IF_ERROR(s->scoreLattice(&lattice_, &xtra_, idx));
        ++idx;
      }
      sproc_->adjustBeamScores(sconf->scoreWeights);
      sproc_->remakeEosBeam(sconf->scoreWeights);
    }
    return Status::Ok();
  }
};

struct Env {
  JumanppEnv env;
  jumanpp_generated::JumandicStatic features;
  i32 beam = 5, gbeam = 6, rcheck = 1, rbeam = 5;
  std::string modelS;
  void init(const char* model, char** extra, int nextra) {
    if (nextra >= 4) {
      beam = atoi(extra[0]);
      gbeam = atoi(extra[1]);
      rcheck = atoi(extra[2]);
      rbeam = atoi(extra[3]);
    }
    modelS = model;
    CHECK_OK(env.loadModel(StringPiece{modelS}));
    env.setBeamSize(beam);
    env.setGlobalBeam(gbeam, rcheck, rbeam);
    CHECK_OK(env.initFeatures(&features));
  }
};

const ConnectionBeamElement* asBeamElem(const ConnectionPtr* p) {
  return reinterpret_cast<const ConnectionBeamElement*>(p);
}

int doDump(const char* modelFile, const char* out, char** extra, int nextra) {
  Env e;
  e.init(modelFile, extra, nextra);
  auto core = e.env.coreHolder();
  auto sconf = e.env.scorers();
  ScoringConfig sc{e.beam, (i32)sconf->scoreWeights.size()};
  AnalyzerConfig ac;
  ac.globalBeamSize = e.gbeam;
  ac.rightGbeamCheck = e.rcheck;
  ac.rightGbeamSize = e.rbeam;
  DumpAnalyzer an{core, sc, ac};
  CHECK_OK(an.initScorers(*sconf));

  // a second, ordinary analyzer only to produce the formatted text
  Analyzer fmtAnalyzer;
  CHECK_OK(e.env.makeAnalyzer(&fmtAnalyzer));
  jumandic::output::JumanFormat jfmt;
  CHECK_OK(jfmt.initialize(fmtAnalyzer.output()));

  auto numPat = an.lattice()->config().numFeaturePatterns;
  auto entrySize = an.lattice()->config().entrySize;
  u32 numScorers = (u32)sconf->scoreWeights.size();
  u32 numPlaceholders = (u32)core->spec().features.numPlaceholders;

  Writer w;
  w.bytes("JPPGOLD1", 8);
  w.put<u32>(e.beam);
  w.put<u32>(e.gbeam);
  w.put<u32>(e.rcheck);
  w.put<u32>(e.rbeam);
  w.put<u32>(numScorers);
  w.put<u32>(numPat);
  w.put<u32>(entrySize);
  w.put<u32>(numPlaceholders);
  size_t countPos = w.buf.size();
  w.put<u32>(0);

  std::string line;
  u32 nsent = 0;
  while (std::getline(std::cin, line)) {
    ++nsent;
    Status s = an.resetForInput(line);
    if (s) s = an.prepareNodeSeeds();
    if (s) s = an.buildLattice();
    if (s) s = an.bootstrapAnalysis();
    if (!s) {
      w.put<u32>(1);
      w.put<u32>(0);
      continue;
    }
    auto lat = an.lattice();
    auto nb = lat->createdBoundaryCount();
    auto& proc = an.sproc();
    features::impl::PrimitiveFeatureContext pfc{
        an.extraNodesContext(), an.dic().fields(), an.dic().entries(),
        an.inputRef().codepoints()};

    std::vector<std::vector<float>> t0(nb);
    std::vector<std::vector<u8>> kept(nb);
    std::vector<std::vector<BeamCandidate>> gbeams(nb);
    // same stage order as analyzer_impl.cc:258-283 (global beam) / :206-245 (full beam)
    if (nb > 3 && e.gbeam <= 0) {
      for (u32 b = 2; b < nb; ++b) {
        auto bnd = lat->boundary(b);
        auto R = bnd->localNodeCount();
        auto left = bnd->ends()->nodePtrs();
        EntryBeam::initializeBlock(bnd->starts()->beamData().data());
        proc.startBoundary(R);
        if (R > 0) {
          // (AnalyzerImpl::computeScoresFull, analyzer_impl.cc:215-219: generated code when the spec matches it, else applyT0)
          if (proc.patternIsStatic()) proc.computeT0All(b, sconf->feature, &pfc);
          else proc.applyT0(b, sconf->feature);
          auto t0buf = proc.scores_.bufferT0();
          t0[b].assign(t0buf.begin(), t0buf.begin() + R);
        }
        kept[b].assign(R, 1);
        for (i32 t1idx = 0; t1idx < (i32)left.size(); ++t1idx) {
          auto& t1node = left[t1idx];
          proc.applyT1(t1node.boundary, t1node.position, sconf->feature);
          proc.resolveBeamAt(t1node.boundary, t1node.position);
          i32 activeBeam = proc.activeBeamSize();
          for (i32 beamIdx = 0; beamIdx < activeBeam; ++beamIdx) {
            proc.applyT2(beamIdx, sconf->feature);
            proc.copyFeatureScores(t1idx, beamIdx, bnd->scores());
          }
        }
        proc.makeBeams(b, bnd, sconf);
      }
    } else if (nb > 3) {
      for (u32 b = 2; b < nb; ++b) {
        auto bnd = lat->boundary(b);
        auto R = bnd->localNodeCount();
        if (R == 0) continue;
        proc.startBoundary(R);
        // (analyzer_impl.cc:270-280: static pattern code computes patterns + T0 here; with a spec whose hash does not
        // match the generated code the patterns were made while the lattice was built and T0 comes from applyT0)
        if (proc.patternIsStatic()) proc.computeT0All(b, sconf->feature, &pfc);
        else proc.applyT0(b, sconf->feature);
        auto t0buf = proc.scores_.bufferT0();
        t0[b].assign(t0buf.begin(), t0buf.begin() + R);
        auto gb = proc.makeGlobalBeam(b, lat->config().globalBeamSize);
        gbeams[b].assign(gb.begin(), gb.end());
        proc.computeGbeamScores(b, gb, sconf->feature);
        kept[b].assign(R, 0);
        if (e.rcheck > 0) {
          u32 toKeep = std::min<u32>(e.rbeam, R);
          for (u32 i = 0; i < toKeep; ++i) kept[b][proc.t0cutoffIdxBuffer_.at(i)] = 1;
        } else {
          kept[b].assign(R, 1);
        }
      }
      CHECK_OK(an.runExtraScorers(sconf));
    }

    w.put<u32>(0);
    w.put<u32>((u32)an.inputRef().numCodepoints());
    w.put<u32>(nb);
    for (u32 b = 0; b < nb; ++b) {
      auto bnd = lat->boundary(b);
      u32 R = bnd->localNodeCount();
      auto ends = bnd->ends()->nodePtrs();
      u32 L = (u32)ends.size();
      w.put<u32>(R);
      w.put<u32>(L);
      for (auto& p : ends) {
        w.put<u16>(p.boundary);
        w.put<u16>(p.position);
      }
      auto starts = bnd->starts();
      bool scored = b >= 2 && R > 0 && nb > 3;
      u32 ngb = scored ? (u32)gbeams[b].size() : 0;
      w.put<u32>(ngb);
      for (u32 i = 0; i < ngb; ++i) {
        w.put<u16>(gbeams[b][i].left());
        w.put<u16>(gbeams[b][i].beam());
        w.put<float>(gbeams[b][i].score());
      }
      for (u32 r = 0; r < R; ++r) {
        auto& ni = starts->nodeInfo().at(r);
        w.put<i32>(ni.entryPtr().rawValue());
        w.put<u16>(ni.start());
        w.put<u16>(ni.end());
        // unk info
        i32 unk[4] = {0, 0, 0, 0};
        auto eptr = ni.entryPtr();
        if (eptr.isSpecial() && eptr != EntryPtr::BOS() && eptr != EntryPtr::EOS()) {
          auto node = an.extraNodesContext()->node(eptr);
          unk[0] = node->header.unk.templatePtr.rawValue();
          unk[1] = node->header.unk.contentHash;
          for (u32 p = 0; p < numPlaceholders && p < 2; ++p) {
            unk[2 + p] = an.extraNodesContext()->placeholderData(eptr, p);
          }
        }
        for (auto x : unk) w.put<i32>(x);
        auto ed = starts->entryData().row(r);
        for (u32 k = 0; k < entrySize; ++k) w.put<i32>(scored ? ed.at(k) : 0);
        auto pat = starts->patternFeatureData().row(r);
        for (u32 k = 0; k < numPat; ++k) w.put<u64>((scored || b < 2) ? pat.at(k) : 0);
        w.put<float>(scored ? t0[b][r] : 0.f);
        w.put<u32>(scored ? kept[b][r] : 0);
        // beam
        auto beam = starts->beamData().row(r);
        for (u32 k = 0; k < (u32)e.beam; ++k) {
          auto& el = beam.at(k);
          bool valid = (scored || b < 2) && !EntryBeam::isFake(el);
          if (b < 2 && k > 0) valid = false;
          if (!valid) {
            for (int q = 0; q < 8; ++q) w.put<u16>(0xffff);
            w.put<float>(0.f);
            w.put<u32>(0);
            continue;
          }
          w.put<u16>(el.ptr.boundary);
          w.put<u16>(el.ptr.left);
          w.put<u16>(el.ptr.right);
          w.put<u16>(el.ptr.beam);
          if (el.ptr.previous != nullptr) {
            auto prev = el.ptr.previous;
            w.put<u16>(prev->boundary);
            w.put<u16>(prev->right);
            // slot of prev inside its beam row
            auto prow = lat->boundary(prev->boundary)->starts()->beamData().row(prev->right);
            u16 slot = (u16)(asBeamElem(prev) - prow.begin());
            w.put<u16>(slot);
            w.put<u16>(0);
          } else {
            for (int q = 0; q < 4; ++q) w.put<u16>(0xffff);
          }
          w.put<float>(el.totalScore);
          w.put<u32>(1);
        }
        // score cells for each gbeam entry (defined for i<rcheck on non-kept nodes)
        for (u32 i = 0; i < ngb; ++i) {
          auto cells = bnd->scores()->nodeScores(r).beamLeft(gbeams[b][i].beam(),
                                                            gbeams[b][i].left());
          bool defined = kept[b][r] || (e.rcheck > 0 && i < (u32)e.rcheck);
          for (u32 q = 0; q < numScorers; ++q) w.put<float>(defined ? cells.at(q) : 0.f);
        }
      }
    }
    // top-1 path from the EOS beam
    {
      std::vector<std::pair<u16, u16>> path;
      auto eos = lat->boundary(nb - 1)->starts()->beamData().row(0);
      const ConnectionPtr* p = nullptr;
      if (nb > 3 && !EntryBeam::isFake(eos.at(0))) p = &eos.at(0).ptr;
      while (p != nullptr && p->boundary >= 2) {
        path.emplace_back(p->boundary, p->right);
        p = p->previous;
      }
      w.put<u32>((u32)path.size());
      for (auto& x : path) {
        w.put<u16>(x.first);
        w.put<u16>(x.second);
      }
    }
    // formatted output through the normal public path
    {
      std::string text;
      Status fs_ = fmtAnalyzer.analyze(line);
      if (fs_) fs_ = jfmt.format(fmtAnalyzer, StringPiece{""});
      if (fs_) text = jfmt.result().str();
      w.put<u32>((u32)text.size());
      w.bytes(text.data(), text.size());
      w.align8();
    }
  }
  std::memcpy(w.buf.data() + countPos, &nsent, 4);
  w.save(out);
  std::cerr << "dumped " << nsent << " sentences, " << w.buf.size() << " bytes\n";
  return 0;
}

// ------------------------------------------------------------------ time ---
int doBootstrapVariant(const char* mdic, const char* out, const char* variant) {
  core::spec::AnalysisSpec spec;
  if (std::string(variant) == "cols") {
    // MORE THAN 8 FEATURE COLUMNS (JPP_MAX_DIC_FIELDS = 16): the jumandic spec as SpecFactory fills it, plus
    // four string columns read from CSV columns 13-16 (the test appends them to the generated dictionary) that four new
    // unigram features read -- 12 feature columns per entry row.
    core::spec::dsl::ModelSpecBuilder bldr;
    jumandic::SpecFactory::fillSpec(bldr);
    auto& x1 = bldr.field(13, "extra1").strings().emptyValue("*").align(3);
    auto& x2 = bldr.field(14, "extra2").strings().emptyValue("*").align(3);
    auto& x3 = bldr.field(15, "extra3").strings().emptyValue("*").align(3);
    auto& x4 = bldr.field(16, "extra4").strings().emptyValue("*").align(3);
    // (unigram features only: the jumandic spec already fills the 14 stored-pattern slots of the device's sweep)
    bldr.unigram({x1});
    bldr.unigram({x2, x3});
    bldr.unigram({x4, x1});
    bldr.unigram({x3, x4, x2, x1});
    CHECK_OK(bldr.build(&spec));
  } else {
    CHECK_OK(jumandic::SpecFactory::makeSpec(&spec));
  }
  auto& ng = spec.features.ngram;
  const std::string v{variant};
  if (v == "drop") {
    ng.pop_back();
  } else if (v == "add") {
    i32 maxIdx = 0;
    for (auto& f : ng) maxIdx = std::max(maxIdx, f.index);
    // patterns 0 and 1 are read by bigrams already, i.e. they are stored ones
    core::spec::NgramFeatureDescriptor uni, bi;
    uni.index = maxIdx + 1;
    uni.references = {1};
    bi.index = maxIdx + 2;
    bi.references = {1, 0};
    ng.push_back(uni);
    ng.push_back(bi);
    // and a different order of two existing bigram features
    i32 a = -1, b = -1;
    for (i32 i = 0; i < (i32)ng.size(); ++i)
      if (ng[i].references.size() == 2) {
        if (a < 0) a = i;
        else if (b < 0) b = i;
      }
    std::swap(ng[a], ng[b]);
  } else if (v == "len") {
    // LENGTH primitives: byte length of the baseform string, codepoints of the reading string, byte length
    // of the surface (the column UNK makers overwrite with a hash: PrimitiveFeatureContext::lengthOf's UNK branch) --
    // each a plain computed feature, all three in one new pattern that only a new unigram reads (it goes last: the
    // trailing numUniOnlyPats patterns are the unigram-only ones, feature_impl_pattern.cc:39)
    auto& F = spec.features;
    auto dicIndexOf = [&](const char* name) -> i32 {
      for (auto& f : spec.dictionary.fields)
        if (f.name == name) return f.dicIndex;
      std::cerr << "no dictionary field " << name << "\n";
      std::exit(2);
    };
    const struct {
      const char* name;
      core::spec::PrimitiveFeatureKind kind;
      const char* field;
    } add[3] = {{"len_baseform_bytes", core::spec::PrimitiveFeatureKind::ByteLength, "baseform"},
                {"len_reading_cps", core::spec::PrimitiveFeatureKind::CodepointSize, "reading"},
                {"len_surface_bytes", core::spec::PrimitiveFeatureKind::ByteLength, "surface"}};
    core::spec::PatternFeatureDescriptor pat;
    pat.index = (i32)F.pattern.size();
    pat.usage = 1;
    for (auto& a : add) {
      core::spec::PrimitiveFeatureDescriptor pd;
      pd.index = (i32)F.primitive.size();
      pd.name = a.name;
      pd.kind = a.kind;
      pd.references = {dicIndexOf(a.field)};
      F.primitive.push_back(pd);
      core::spec::ComputationFeatureDescriptor cd;
      cd.name = a.name;
      cd.index = (i32)F.computation.size();
      cd.primitiveFeature = pd.index;
      F.computation.push_back(cd);
      pat.references.push_back(cd.index);
    }
    F.pattern.push_back(pat);
    F.numUniOnlyPats += 1;
    F.totalPrimitives = (i32)F.primitive.size();
    i32 maxIdx = 0;
    for (auto& f : ng) maxIdx = std::max(maxIdx, f.index);
    core::spec::NgramFeatureDescriptor uni;
    uni.index = maxIdx + 1;
    uni.references = {pat.index};
    ng.push_back(uni);
  } else if (v == "cols") {
    // (built above)
  } else {
    std::cerr << "unknown variant " << v << "\n";
    return 2;
  }
  const std::string mdicS{mdic}, outS{out};
  util::FullyMappedFile file;
  CHECK_OK(file.open(StringPiece{mdicS}, util::MMapType::ReadOnly));
  core::dic::DictionaryBuilder builder;
  CHECK_OK(builder.importSpec(&spec));
  CHECK_OK(builder.importCsv(StringPiece{mdicS}, file.contents()));
  core::dic::DictionaryHolder holder;
  CHECK_OK(holder.load(builder.result()));
  core::model::ModelInfo minfo{};
  minfo.parts.emplace_back();
  CHECK_OK(builder.fillModelPart(&minfo.parts.back(), "variant"));
  core::model::ModelSaver saver;
  CHECK_OK(saver.open(StringPiece{outS}));
  CHECK_OK(saver.save(minfo));
  return 0;
}

int doTop1(const char* modelFile, const char* out, char** extra, int nextra) {
  Env e;
  e.init(modelFile, extra, nextra);
  Analyzer an;
  CHECK_OK(e.env.makeAnalyzer(&an));
  std::vector<std::string> lines;
  std::string line;
  while (std::getline(std::cin, line)) lines.push_back(line);
  Writer w;
  w.put<u32>(0x31504f54u);  // "TOP1"
  w.put<u32>((u32)lines.size());
  std::vector<const ConnectionPtr*> path;
  for (auto& l : lines) {
    Status st = an.analyze(l);
    if (!st) {
      w.put<u32>(1);
      w.put<u32>(0);
      continue;
    }
    auto* lat = an.impl()->lattice();
    const int last = (int)lat->createdBoundaryCount() - 1;
    path.clear();
    const ConnectionBeamElement& top = lat->boundary(last)->starts()->beamData().row(0).at(0);
    // the EOS connection itself is dropped; an empty input has no path (analyzer_impl.cc:255-258)
    const ConnectionPtr* p = (last <= 2 || EntryBeam::isFake(top)) ? nullptr : top.ptr.previous;
    while (p != nullptr && p->boundary >= 2) {
      path.push_back(p);
      p = p->previous;
    }
    w.put<u32>(0);
    w.put<u32>((u32)path.size());
    for (size_t k = path.size(); k-- > 0;) {
      auto& ni = lat->boundary(path[k]->boundary)->starts()->nodeInfo().at(path[k]->right);
      w.put<i32>(ni.entryPtr().rawValue());
      w.put<u16>(ni.start());
      w.put<u16>(ni.end());
    }
  }
  w.save(out);
  return 0;
}

// ---- top1x: the reference with a TEST scorer in ScorerDef::others and / or a TEST ScorePlugin ---------------------------
// The checker of jppgpu_analyze_batch_scored and jppgpu_analyze_batch_pairs.
//
// Test scorer (a ScoreComputer, score_api.h:54-59): for every boundary b >= 2 with nodes, every right node r and every
// element i of the boundary's global beam, the cell of (r, i) gets
//     -0.25 * [left node is special] - 0.0625 * (codepoints of the left node) + 0.125 * [right node is special]
// (dyadic values: exact in float).  Test plugin (score_plugin.h:14-19): every connection whose LEFT node is special
// loses 1, every connection into a node of more than two codepoints whose left node starts at an odd position loses 0.5.
namespace {
struct TestScorer : public ScoreComputer {
  Status scoreLattice(Lattice* l, const ExtraNodesContext*, u32 scorerIdx) override {
    const u32 nb = l->createdBoundaryCount();
    for (u32 b = 2; b < nb; ++b) {
      auto bnd = l->boundary(b);
      const u32 R = bnd->localNodeCount();
      if (R == 0) continue;
      auto ends = bnd->ends()->nodePtrs();
      auto gbeam = bnd->ends()->globalBeam();
      for (auto el : gbeam) {
        // (left, beam) of the element: its node's place in this boundary's ends list, its slot in that node's beam
        i32 left = -1;
        for (u32 q = 0; q < ends.size(); ++q)
          if (ends.at(q).boundary == el->ptr.boundary && ends.at(q).position == el->ptr.right) left = (i32)q;
        if (left < 0) return JPPS_INVALID_STATE << "test scorer: gbeam element outside the ends list";
        auto lstarts = l->boundary(el->ptr.boundary)->starts();
        auto row = lstarts->beamData().row(el->ptr.right);
        const i32 beam = (i32)(el - row.begin());
        auto& lni = lstarts->nodeInfo().at(el->ptr.right);
        for (u32 r = 0; r < R; ++r) {
          auto& rni = bnd->starts()->nodeInfo().at(r);
          float v = 0.f;
          if (lni.entryPtr().isSpecial()) v -= 0.25f;
          v -= 0.0625f * (float)(lni.end() - lni.start());
          if (rni.entryPtr().isSpecial()) v += 0.125f;
          bnd->scores()->nodeScores((i32)r).beamLeft(beam, left).at(scorerIdx) = v;
        }
      }
    }
    return Status::Ok();
  }
};
struct TestScorerFactory : public ScorerFactory {
  Status load(const model::ModelInfo&) override { return Status::Ok(); }
  Status makeInstance(std::unique_ptr<ScoreComputer>* result) override {
    result->reset(new TestScorer());
    return Status::Ok();
  }
};
struct TestPlugin : public ScorePlugin {
  bool updateScore(const Lattice* l, const ConnectionPtr& ptr, float* score) const override {
    auto bnd = l->boundary(ptr.boundary);
    auto& lp = bnd->ends()->nodePtrs().at(ptr.left);
    auto& lni = l->boundary(lp.boundary)->starts()->nodeInfo().at(lp.position);
    auto& rni = bnd->starts()->nodeInfo().at(ptr.right);
    // ONE subtraction per connection: the batched form of the hook hands the device an amount per connection
    float amount = 0.f;
    if (lni.entryPtr().isSpecial()) amount += 1.0f;
    if (rni.end() - rni.start() > 2 && (lni.start() & 1) != 0) amount += 0.5f;
    if (amount == 0.f) return false;
    *score -= amount;
    return true;
  }
};
}  // namespace

// ref_dump top1x <model> <out.bin> <scorer-weight|none> <plugin 0|1> [beam gbeam rcheck rbeam] < corpus
int doTop1x(const char* modelFile, const char* out, const char* scorerW, const char* usePlugin, char** extra, int nextra) {
  Env e;
  e.init(modelFile, extra, nextra);
  ScorerDef def = *e.env.scorers();
  TestScorerFactory factory;
  if (std::string(scorerW) != "none") {
    def.others.push_back(&factory);
    def.scoreWeights.push_back((float)atof(scorerW));
  }
  AnalyzerConfig ac;
  ac.globalBeamSize = e.gbeam;
  ac.rightGbeamCheck = e.rcheck;
  ac.rightGbeamSize = e.rbeam;
  ScoringConfig sc{e.beam, (i32)def.scoreWeights.size()};
  Analyzer an;
  CHECK_OK(an.initialize(e.env.coreHolder(), ac, sc, &def));
  TestPlugin plugin;
  const bool withPlugin = atoi(usePlugin) != 0;
  std::vector<std::string> lines;
  std::string line;
  while (std::getline(std::cin, line)) lines.push_back(line);
  Writer w;
  w.put<u32>(0x31504f54u);  // "TOP1": the layout of `top1`, plus the EOS beam totals (float bits) behind every path
  w.put<u32>((u32)lines.size());
  std::vector<const ConnectionPtr*> path;
  for (auto& l : lines) {
    Status st = an.analyze(l, withPlugin ? &plugin : nullptr);
    if (!st) {
      w.put<u32>(1);
      w.put<u32>(0);
      continue;
    }
    auto* lat = an.impl()->lattice();
    const int last = (int)lat->createdBoundaryCount() - 1;
    path.clear();
    const ConnectionBeamElement& top = lat->boundary(last)->starts()->beamData().row(0).at(0);
    const ConnectionPtr* p = (last <= 2 || EntryBeam::isFake(top)) ? nullptr : top.ptr.previous;
    while (p != nullptr && p->boundary >= 2) {
      path.push_back(p);
      p = p->previous;
    }
    w.put<u32>(0);
    w.put<u32>((u32)path.size());
    for (size_t k = path.size(); k-- > 0;) {
      auto& ni = lat->boundary(path[k]->boundary)->starts()->nodeInfo().at(path[k]->right);
      w.put<i32>(ni.entryPtr().rawValue());
      w.put<u16>(ni.start());
      w.put<u16>(ni.end());
    }
  }
  w.save(out);
  // second file: the EOS beam totals of every sentence (beam floats, fake slots as 0)
  {
    Writer t;
    t.put<u32>((u32)lines.size());
    t.put<u32>((u32)e.beam);
    for (auto& l : lines) {
      Status st = an.analyze(l, withPlugin ? &plugin : nullptr);
      auto* lat = an.impl()->lattice();
      const int last = (int)lat->createdBoundaryCount() - 1;
      for (int k = 0; k < e.beam; ++k) {
        float v = 0.f;
        if (st && last > 2) {
          auto& el = lat->boundary(last)->starts()->beamData().row(0).at(k);
          if (!EntryBeam::isFake(el)) v = el.totalScore;
        }
        t.put<float>(v);
      }
    }
    t.save((std::string(out) + ".totals").c_str());
  }
  return 0;
}

int doTime(const char* modelFile, char** extra, int nextra) {
  Env e;
  e.init(modelFile, extra, nextra);
  Analyzer an;
  CHECK_OK(e.env.makeAnalyzer(&an));
  jumandic::output::JumanFormat jfmt;
  CHECK_OK(jfmt.initialize(an.output()));
  std::vector<std::string> lines;
  std::string line;
  while (std::getline(std::cin, line)) lines.push_back(line);
  using clk = std::chrono::steady_clock;
  double best = 1e30, bestFmt = 1e30;
  size_t outBytes = 0;
  for (int rep = 0; rep < 3; ++rep) {
    double tAnalyze = 0, tFmt = 0;
    outBytes = 0;
    for (auto& l : lines) {
      auto t0 = clk::now();
      Status s = an.analyze(l);
      auto t1 = clk::now();
      if (s) {
        s = jfmt.format(an, StringPiece{""});
        outBytes += jfmt.result().size();
      }
      auto t2 = clk::now();
      tAnalyze += std::chrono::duration<double>(t1 - t0).count();
      tFmt += std::chrono::duration<double>(t2 - t1).count();
    }
    if (tAnalyze + tFmt < best + bestFmt) {
      best = tAnalyze;
      bestFmt = tFmt;
    }
  }
  std::printf(
      "{\"sentences\": %zu, \"analyze_s\": %.6f, \"format_s\": %.6f, "
      "\"sent_per_s_analyze\": %.1f, \"sent_per_s_total\": %.1f, \"out_bytes\": %zu}\n",
      lines.size(), best, bestFmt, lines.size() / best,
      lines.size() / (best + bestFmt), outBytes);
  return 0;
}


// ------------------------------------------------------------------ shim ---
struct GpuLib {
  void* h = nullptr;
  decltype(&jppgpu_ctx_create) ctx_create = nullptr;
  decltype(&jppgpu_ctx_destroy) ctx_destroy = nullptr;
  decltype(&jppgpu_analyze_batch) analyze_batch = nullptr;
  decltype(&jppgpu_result_fetch) result_fetch = nullptr;
  decltype(&jppgpu_result_release) result_release = nullptr;
  decltype(&jppgpu_last_error) last_error = nullptr;
  bool open(const char* path) {
    h = dlopen(path, RTLD_NOW | RTLD_LOCAL);
    if (!h) {
      std::cerr << "dlopen failed: " << dlerror() << "\n";
      return false;
    }
#define JPP_SYM(n) n = reinterpret_cast<decltype(n)>(dlsym(h, "jppgpu_" #n)); if (!n) { std::cerr << "missing symbol jppgpu_" #n "\n"; return false; }
    JPP_SYM(ctx_create) JPP_SYM(ctx_destroy) JPP_SYM(analyze_batch) JPP_SYM(result_fetch) JPP_SYM(result_release) JPP_SYM(last_error)
#undef JPP_SYM
    return true;
  }
};

// the flattened FeaturesSpec the library compares with its built-in tables (same bytes as SEC_FEATURES above)
void flattenFeatures(const spec::AnalysisSpec& spec, Writer& s) {
  auto& fs_ = spec.features;
  s.put<i32>((i32)fs_.primitive.size());
  for (auto& p : fs_.primitive) {
    s.put<i32>((i32)p.kind);
    putInts(s, p.references);
  }
  s.put<i32>((i32)fs_.computation.size());
  for (auto& c : fs_.computation) {
    s.put<i32>(c.primitiveFeature);
    putInts(s, c.trueBranch);
    putInts(s, c.falseBranch);
  }
  s.put<i32>((i32)fs_.pattern.size());
  for (auto& p : fs_.pattern) {
    s.put<i32>(p.index);
    putInts(s, p.references);
  }
  s.put<i32>((i32)fs_.ngram.size());
  for (auto& n : fs_.ngram) {
    s.put<i32>(n.index);
    putInts(s, n.references);
  }
}

int doShim(const char* modelFile, const char* libPath, int latticeN, char** extra, int nextra) {
  Env e;
  e.init(modelFile, extra, nextra);
  GpuLib lib;
  if (!lib.open(libPath)) return 2;

  // ---- the model handed over from the reference's own structures ----
  std::string modelS{modelFile};
  model::FilesystemModel fs;
  CHECK_OK(fs.open(StringPiece{modelS}));
  model::ModelInfo info;
  CHECK_OK(fs.load(&info));
  dic::BuiltDictionary bd;
  CHECK_OK(bd.restoreDictionary(info));
  auto& spec = bd.spec;
  jppgpu_model m{};
  m.trie = bd.trieContent.data();
  m.trie_bytes = bd.trieContent.size();
  m.entry_ptrs = bd.entryPointers.data();
  m.entry_ptrs_bytes = bd.entryPointers.size();
  m.entry_data = bd.entryData.data();
  m.entry_data_bytes = bd.entryData.size();
  auto pp = info.firstPartOf(model::ModelPartKind::Perceprton);
  if (!pp) {
    std::cerr << "model has no perceptron\n";
    return 2;
  }
  {
    util::serialization::Loader ldr{pp->data[0]};
    PerceptronInfo pi{};
    if (!ldr.load(&pi)) return 2;
    m.weights = reinterpret_cast<const float*>(pp->data[1].data());
    m.weight_exponent = (u32)pi.modelSizeExponent;
  }
  m.num_features = spec.features.numDicFeatures;
  m.num_placeholders = spec.features.numPlaceholders;
  std::vector<jppgpu_unk_maker> unk;
  for (auto& u : spec.unkCreators) {
    u32 mask = 0;
    for (auto f : u.replaceFields) mask |= 1u << f;
    unk.push_back(jppgpu_unk_maker{(i32)u.type, (i32)u.charClass, u.patternPtr, u.priority,
                                   u.features.empty() ? -1 : u.features[0].targetPlaceholder, mask});
  }
  m.unk_makers = unk.data();
  m.num_unk_makers = (i32)unk.size();
  Writer flat;
  flattenFeatures(spec, flat);
  m.feature_spec = flat.buf.data();
  m.feature_spec_bytes = flat.buf.size();
  auto sconf = e.env.scorers();
  const bool rnn = sconf->scoreWeights.size() == 2;
  OracleRnnHeader rh;
  if (rnn) {
    auto rp = info.firstPartOf(model::ModelPartKind::Rnn);
    util::serialization::Loader l{rp->data[0]};
    if (!l.load(&rh)) return 2;
    m.has_rnn = 1;
    m.rnn_known_index = rp->data[1].data();
    m.rnn_known_index_bytes = rp->data[1].size();
    m.rnn_unk_index = rp->data[2].data();
    m.rnn_unk_index_bytes = rp->data[2].size();
    m.rnn_matrix = reinterpret_cast<const float*>(rp->data[3].data());
    m.rnn_embeddings = reinterpret_cast<const float*>(rp->data[4].data());
    m.rnn_nce_embeddings = reinterpret_cast<const float*>(rp->data[5].data());
    m.rnn_maxent = reinterpret_cast<const float*>(rp->data[6].data());
    m.rnn_layer_size = rh.rnnHeader.layerSize;
    m.rnn_maxent_order = rh.rnnHeader.maxentOrder;
    m.rnn_maxent_size = rh.rnnHeader.maxentSize;
    m.rnn_vocab_size = rh.rnnHeader.vocabSize;
    float nce = rh.rnnHeader.nceLnz;
    if (rh.config.rnnWeight.defined()) nce = rh.config.rnnWeight;   // RnnScorerGbeamFactory::load, :465-467
    m.rnn_nce_constant = nce;
    m.rnn_unk_id = rh.unkIdx;
    m.rnn_unk_constant = rh.config.unkConstantTerm;
    m.rnn_unk_length = rh.config.unkLengthPenalty;
    m.rnn_num_fields = (u32)rh.fields.size();
    for (size_t i = 0; i < rh.fields.size() && i < 8; ++i) m.rnn_fields[i] = rh.fields[i];
  }
  jppgpu_config c = JPPGPU_CONFIG_INIT;
  c.beam = e.beam;
  c.global_beam = e.gbeam;
  c.right_check = e.rcheck;
  c.right_beam = e.rbeam;
  c.max_input_bytes = 4096;
  c.device = 0;
  c.use_rnn = rnn ? 1 : 0;
  c.weight_perceptron = sconf->scoreWeights[0];
  c.weight_rnn = rnn ? sconf->scoreWeights[1] : 0.f;
  jppgpu_ctx* ctx = nullptr;
  if (lib.ctx_create(&m, &c, &ctx) != JPPGPU_OK) {
    std::cerr << "jppgpu_ctx_create: " << lib.last_error() << "\n";
    return 2;
  }

  // ---- one batched call instead of the per-line loop ----
  std::vector<std::string> lines;
  std::string line;
  while (std::getline(std::cin, line)) lines.push_back(line);
  std::string text;
  std::vector<uint32_t> offs{0};
  for (auto& l : lines) {
    text += l;
    offs.push_back((uint32_t)text.size());
  }
  jppgpu_result* res = nullptr;
  if (lib.analyze_batch(ctx, text.data(), offs.data(), (uint32_t)lines.size(), &res) != JPPGPU_OK) {
    std::cerr << "jppgpu_analyze_batch: " << lib.last_error() << "\n";
    return 2;
  }
  jppgpu_result_view v{};
  if (lib.result_fetch(res, JPPGPU_FETCH_FULL, &v) != JPPGPU_OK) {
    std::cerr << "jppgpu_result_fetch: " << lib.last_error() << "\n";
    return 2;
  }

  // ---- the Analyzer shim ----
  // (refAn2: a second ordinary reference analyzer.  LatticeFormat chooses among exactly tied connections of a node
  // by iterating a set hashed on host addresses, lattice_config.h:109-124, so two reference analyzers may print
  // different lattice lines for the same sentence; such sentences are counted, not held against the shim.)
  Analyzer refAn, refAn2, shimAn;
  CHECK_OK(e.env.makeAnalyzer(&refAn));
  CHECK_OK(e.env.makeAnalyzer(&refAn2));
  CHECK_OK(e.env.makeAnalyzer(&shimAn));
  jumandic::output::JumanFormat jRef, jShim;
  CHECK_OK(jRef.initialize(refAn.output()));
  CHECK_OK(jShim.initialize(shimAn.output()));
  jumandic::output::LatticeFormat lRef(latticeN > 0 ? latticeN : 1), lRef2(latticeN > 0 ? latticeN : 1),
      lShim(latticeN > 0 ? latticeN : 1);
  if (latticeN > 0) {
    CHECK_OK(lRef.initialize(refAn.output()));
    CHECK_OK(lRef2.initialize(refAn2.output()));
    CHECK_OK(lShim.initialize(shimAn.output()));
  }
  const int beam = v.beam, G = v.global_beam, S = v.num_scorers;
  long same = 0, sameLattice = 0, failed = 0, statusMismatch = 0, refUnstable = 0;
  for (size_t si = 0; si < lines.size(); ++si) {
    Status rs = refAn.analyze(lines[si]);
    if (!rs || v.status[si] != JPPGPU_SENT_OK) {
      if ((bool)rs != (v.status[si] == JPPGPU_SENT_OK)) ++statusMismatch;
      ++failed;
      continue;
    }
    CHECK_OK(jRef.format(refAn, StringPiece{""}));
    std::string expect = jRef.result().str();
    std::string expectLattice, expectLattice2;
    if (latticeN > 0) {
      CHECK_OK(lRef.format(refAn, StringPiece{""}));
      expectLattice = lRef.result().str();
      CHECK_OK(refAn2.analyze(lines[si]));
      CHECK_OK(lRef2.format(refAn2, StringPiece{""}));
      expectLattice2 = lRef2.result().str();
      // ... and such a choice exists wherever two live entries of one node's beam have bit-equal totals: two
      // analyzers in one process often make the same choice (similar heaps), a third heap need not
      bool tied = expectLattice2 != expectLattice;
      {
        auto* lat = refAn.impl()->lattice();
        for (int b = 2; !tied && b < (int)lat->createdBoundaryCount(); ++b) {
          auto bnd = lat->boundary(b);
          if (bnd->localNodeCount() == 0) continue;
          auto beams = bnd->starts()->beamData();
          for (int r = 0; !tied && r < (int)bnd->localNodeCount(); ++r) {
            auto row = beams.row(r);
            for (int q1 = 0; !tied && q1 < (int)row.size(); ++q1) {
              if (EntryBeam::isFake(row.at(q1))) continue;
              for (int q2 = q1 + 1; q2 < (int)row.size(); ++q2) {
                if (EntryBeam::isFake(row.at(q2))) continue;
                if (row.at(q1).totalScore == row.at(q2).totalScore) {
                  tied = true;
                  break;
                }
              }
            }
          }
        }
      }
      if (tied) ++refUnstable;
    }
    // -- re-materialise --
    AnalyzerImpl* impl = shimAn.impl();
    CHECK_OK(impl->resetForInput(lines[si]));
    const uint64_t nb = v.node_base[si], bb = v.bnd_base[si];
    const uint32_t N = v.n_nodes[si], ncp = v.n_codepoints[si];
    auto* xtra = impl->extraNodesContext();
    auto& input = impl->input();
    for (uint32_t k = 2; k + 1 < N; ++k) {
      const jppgpu_node& nd = v.nodes[nb + k];
      if (nd.entry_ptr >= 0) {
        impl->latticeBldr()->appendSeed(EntryPtr{nd.entry_ptr}, nd.start, nd.end);
      } else {
        const jppgpu_unk& u = v.unk[nb + k];
        auto node = xtra->makeZeroedUnk();
        auto data = xtra->nodeContent(node);
        for (int f = 0; f < (int)data.size(); ++f) data.at(f) = v.entry_rows[(nb + k) * 8 + f];
        node->header.unk.surface = input.surface(nd.start, nd.end);
        node->header.unk.contentHash = u.content_hash;
        node->header.unk.templatePtr = EntryPtr{u.template_ptr};
        xtra->putPlaceholderData(node, 0, (i32)u.placeholder[0]);
        xtra->putPlaceholderData(node, 1, (i32)u.placeholder[1]);
        impl->latticeBldr()->appendSeed(node->ptr(), nd.start, nd.end);
      }
    }
    if (!impl->latticeBldr()->checkConnectability()) {
      std::cerr << "sentence " << si << ": device lattice is not connected\n";
      return 1;
    }
    CHECK_OK(impl->latticeBldr()->prepare());
    CHECK_OK(impl->buildLattice());
    CHECK_OK(impl->bootstrapAnalysis());
    Lattice* lat = impl->lattice();
    if ((uint32_t)lat->createdBoundaryCount() != ncp + 3) {
      std::cerr << "sentence " << si << ": boundary count\n";
      return 1;
    }
    // node index -> (boundary, position)
    auto locate = [&](uint32_t node, u16* bnd, u16* pos) {
      if (node == 0) { *bnd = 0; *pos = 0; return; }
      if (node == 1) { *bnd = 1; *pos = 0; return; }
      const uint32_t b = (node + 1 == N) ? ncp + 2 : (uint32_t)v.nodes[nb + node].start + 2;
      *bnd = (u16)b;
      *pos = (u16)(node - v.bnd_first[bb + b]);
    };
    for (uint32_t b = 2; b <= ncp + 2; ++b) {
      const uint32_t R = v.bnd_count[bb + b], first = v.bnd_first[bb + b];
      auto bnd = lat->boundary(b);
      if (bnd->localNodeCount() != R) {
        std::cerr << "sentence " << si << " boundary " << b << ": node count\n";
        return 1;
      }
      if (R == 0) continue;
      auto beams = bnd->starts()->beamData();
      const uint32_t ngb = v.gbeam_count[bb + b];
      for (uint32_t r = 0; r < R; ++r) {
        const uint64_t k = nb + first + r;
        auto row = beams.row(r);
        for (int q = 0; q < beam; ++q) {
          const jppgpu_beam_slot& sl = v.beams[k * beam + q];
          if (sl.left == 0xffff && sl.beam == 0xffff) {
            std::memset(&row.at(q), 0xff, sizeof(ConnectionBeamElement));
            continue;
          }
          u16 pb, pr;
          locate(sl.prev_node, &pb, &pr);
          const ConnectionPtr* prev = &lat->boundary(pb)->starts()->beamData().row(pr).at(sl.beam).ptr;
          row.at(q) = ConnectionBeamElement{ConnectionPtr{(u16)b, sl.left, (u16)r, sl.beam, prev}, sl.total};
        }
        // score cells: (beam, left) of global-beam entry i -> cells[node][i][scorer]
        auto ns = bnd->scores()->nodeScores(r);
        for (uint32_t i = 0; i < ngb; ++i) {
          const uint32_t lb = v.gbeam[((bb + b) * G + i) * 2];   // left | beam << 16
          auto dst = ns.beamLeft((i32)(lb >> 16), (i32)(lb & 0xffff));
          for (int sc = 0; sc < S; ++sc) dst.at(sc) = v.cells[(k * G + i) * S + sc];
        }
      }
    }
    // the reference's formatters, unmodified, on the shim analyzer
    CHECK_OK(jShim.format(shimAn, StringPiece{""}));
    if (jShim.result().str() == expect) ++same;
    else if (same + 3 > (long)si) std::cerr << "sentence " << si << " differs:\n" << expect << "---\n" << jShim.result().str();
    if (latticeN > 0) {
      CHECK_OK(lShim.format(shimAn, StringPiece{""}));
      if (lShim.result().str() == expectLattice || lShim.result().str() == expectLattice2) ++sameLattice;
    }
  }
  lib.result_release(res);
  lib.ctx_destroy(ctx);
  const long ok = (long)lines.size() - failed;
  std::printf("{\"sentences\": %zu, \"analysed\": %ld, \"status_mismatch\": %ld, \"juman_identical\": %ld, "
              "\"lattice_n\": %d, \"lattice_identical\": %ld, \"lattice_unstable_in_reference\": %ld}\n",
              lines.size(), ok, statusMismatch, same, latticeN, sameLattice, refUnstable);
  return (statusMismatch == 0 && same == ok) ? 0 : 1;
}

}  // namespace

// `ngrams MODEL OUT [beams] < text`: what the trainer's loss computation reads off an analysed lattice
// (LossCalculator::addTopNgrams, core/training/loss.cc:289-300): for every connection on the top-1 path, from the
// EOS side back to the first morpheme, the u32 n-gram feature values of NgramFeaturesComputer::calculateNgramFeatures
// for (t2, t1, t0).  Binary: u32 magic 'NGR1', u32 sentences, u32 features per node; per sentence u32 status
// (0 ok), u32 positions; per position u16 boundary, u16 position of t0, then the features.
int doNgrams(const char* modelFile, const char* out, char** extra, int nextra) {
  Env e;
  e.init(modelFile, extra, nextra);
  Analyzer an;
  CHECK_OK(e.env.makeAnalyzer(&an));
  an.impl()->setStoreAllPatterns(true);   // the trainer's analyzer keeps the unigram-only patterns too (training_env.h:80)
  std::vector<std::string> lines;
  std::string line;
  while (std::getline(std::cin, line)) lines.push_back(line);
  std::ofstream os(out, std::ios::binary);
  auto put32 = [&](u32 v) { os.write(reinterpret_cast<const char*>(&v), 4); };
  auto put16 = [&](u16 v) { os.write(reinterpret_cast<const char*>(&v), 2); };
  const u32 nf = (u32)an.impl()->core().spec().features.ngram.size();
  put32(0x3152474eu);
  put32((u32)lines.size());
  put32(nf);
  std::vector<u32> buf(nf);
  for (auto& l : lines) {
    Status st = an.analyze(l);
    if (!st) {
      put32(1);
      put32(0);
      continue;
    }
    auto* lat = an.impl()->lattice();
    core::features::NgramFeaturesComputer nfc{lat, an.impl()->core().features()};
    const int last = (int)lat->createdBoundaryCount() - 1;
    std::vector<const ConnectionPtr*> path;
    const ConnectionBeamElement& top = lat->boundary(last)->starts()->beamData().row(0).at(0);
    // (an empty input leaves the previous sentence's beams in place: no path)
    const ConnectionPtr* p = (last <= 2 || EntryBeam::isFake(top)) ? nullptr : &top.ptr;
    while (p != nullptr && p->boundary >= 2) {   // boundaries 0 and 1 are the BOS nodes
      path.push_back(p);
      p = p->previous;
    }
    put32(0);
    put32((u32)path.size());
    for (auto* t0 : path) {
      const ConnectionPtr& t1 = *t0->previous;
      const ConnectionPtr& t2 = *t1.previous;
      core::features::NgramFeatureRef nfr{t2.latticeNodePtr(), t1.latticeNodePtr(), t0->latticeNodePtr()};
      nfc.calculateNgramFeatures(nfr, &buf);
      put16(t0->boundary);
      put16(t0->right);
      os.write(reinterpret_cast<const char*>(buf.data()), nf * 4);
    }
  }
  return os.good() ? 0 : 1;
}

int main(int argc, char** argv) {
  if (argc < 2) {
    std::cerr << "usage: ref_dump export|mkmodel|dump|time|shim|ngrams|top1 ...\n";
    return 2;
  }
  std::string cmd = argv[1];
  if (cmd == "export" && argc == 4) return doExport(argv[2], argv[3]);
  if (cmd == "mkmodel" && argc == 7)
    return doMkModel(argv[2], argv[3], atoi(argv[4]), strtoull(argv[5], nullptr, 0),
                     (float)atof(argv[6]));
  if (cmd == "dump" && argc >= 4) return doDump(argv[2], argv[3], argv + 4, argc - 4);
  if (cmd == "time" && argc >= 3) return doTime(argv[2], argv + 3, argc - 3);
  if (cmd == "shim" && argc >= 5) return doShim(argv[2], argv[3], atoi(argv[4]), argv + 5, argc - 5);
  if (cmd == "ngrams" && argc >= 4) return doNgrams(argv[2], argv[3], argv + 4, argc - 4);
  if (cmd == "top1" && argc >= 4) return doTop1(argv[2], argv[3], argv + 4, argc - 4);
  if (cmd == "top1x" && argc >= 6) return doTop1x(argv[2], argv[3], argv[4], argv[5], argv + 6, argc - 6);
  if (cmd == "bootstrapv" && argc == 5) return doBootstrapVariant(argv[2], argv[3], argv[4]);
  std::cerr << "bad arguments\n";
  return 2;
}
