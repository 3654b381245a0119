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
//   ref_dump time    <model.jppmdl> [beam gbeam rcheck rbeam] < corpus
//       wall-clock of Analyzer::analyze (+JumanFormat) over the corpus, phases split
//       as BASELINE.md section 3.
#include <chrono>
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
#include "core/impl/feature_impl_types.h"
#include "core/impl/model_io.h"
#include "core/impl/perceptron_io.h"
#include "jpp_jumandic_cg.h"
#include "jumandic/shared/juman_format.h"
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
          proc.computeT0All(b, sconf->feature, &pfc);
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
        proc.computeT0All(b, sconf->feature, &pfc);
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

}  // namespace

int main(int argc, char** argv) {
  if (argc < 2) {
    std::cerr << "usage: ref_dump export|mkmodel|dump|time ...\n";
    return 2;
  }
  std::string cmd = argv[1];
  if (cmd == "export" && argc == 4) return doExport(argv[2], argv[3]);
  if (cmd == "mkmodel" && argc == 7)
    return doMkModel(argv[2], argv[3], atoi(argv[4]), strtoull(argv[5], nullptr, 0),
                     (float)atof(argv[6]));
  if (cmd == "dump" && argc >= 4) return doDump(argv[2], argv[3], argv + 4, argc - 4);
  if (cmd == "time" && argc >= 3) return doTime(argv[2], argv + 3, argc - 3);
  std::cerr << "bad arguments\n";
  return 2;
}
