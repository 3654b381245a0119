#include "lattice_format.h"

#include <algorithm>
#include <cstdio>

namespace jumanpp_amd {

namespace {

inline void put(std::string& p, StringPiece s) { p.append(s.data(), s.size()); }
inline void putInt(std::string& p, long long v) { p += std::to_string(v); }
// the reference prints floats through fmt's BasicWriter << double, i.e. printf("%g")
inline void putFloat(std::string& p, float v) {
  char buf[48];
  int n = std::snprintf(buf, sizeof(buf), "%g", (double)v);
  p.append(buf, (size_t)n);
}
inline StringPiece escapeTab(StringPiece sp) {
  if (sp.size() == 1 && sp[0] == '\t') return StringPiece("\\t");
  return sp;
}
inline StringPiece ifEmpty(StringPiece s, StringPiece d) { return s.empty() ? d : s; }
inline bool isFake(const jppgpu_beam_slot& s) { return s.left == 0xffff && s.beam == 0xffff; }

}  // namespace

void formatNormalizedFeature(std::string& p, int32_t v);  // juman_format.cc

Status LatticeFormat::initialize(const ModelImage* model, const std::vector<float>& scoreWeights) {
  model_ = model;
  weights_ = scoreWeights;
  if (weights_.empty()) return Status::InvalidParameter("score weights are empty");
  if (!model->hasIdMap()) {
    return Status::InvalidState("model image has no JUMAN id tables (re-export it with the current ref_dump)");
  }
  OutputManager om(model);
  return flds_.initialize(om);
}

Status LatticeFormat::format(const GpuAnalyzer& analysis, size_t sentence, StringPiece comment) {
  printer_.clear();
  JPPA_RETURN_IF_ERROR(analysis.sentenceStatus(sentence));
  uint32_t local = 0;
  const jppgpu_result_view& v = analysis.viewOf(sentence, &local);
  if (v.beams == nullptr || v.cells == nullptr) {
    return Status::InvalidState("the lattice format needs analyzeBatch(inputs, fullLattice = true)");
  }
  if (v.global_beam <= 0) return Status::NotImplemented("lattice format needs the global beam (score cells)");
  SentenceResult s = analysis.sentence(sentence);
  if (s.numNodes <= 3) {  // createdBoundaryCount() == 3: empty input
    printer_ = "EOS\n";
    return Status::Ok();
  }
  const uint64_t nb = v.node_base[local];
  const int32_t beam = v.beam, G = v.global_beam, S = v.num_scorers;
  const jppgpu_beam_slot* beams = v.beams + nb * (uint64_t)beam;
  const float* cells = v.cells + nb * (uint64_t)G * S;
  const uint32_t eos = s.numNodes - 1;
  // lattice_format.cc:98-101: with auto-beam the number of printed paths is the sentence's beam
  const int32_t outputN = analysis.autoBeamSize(sentence) > 0 ? analysis.autoBeamSize(sentence) : topN_;

  // LatticeFormatInfo::fillInfo (lattice_format.cc:13-43)
  info_.clear();
  const int32_t maxN = std::min<int32_t>(beam, outputN);
  for (int32_t i = 0; i < maxN; ++i) {
    const jppgpu_beam_slot& el = beams[(uint64_t)eos * beam + i];
    if (isFake(el)) break;
    uint32_t node = el.prev_node;
    uint32_t slot = el.beam;
    while (node >= 2 && node != 0xffffffffu) {
      const jppgpu_beam_slot& c = beams[(uint64_t)node * beam + slot];
      NodeInfo& ni = info_[node];
      ni.ranks.push_back((uint16_t)i);
      if (std::find(ni.slots.begin(), ni.slots.end(), slot) == ni.slots.end()) ni.slots.push_back(slot);
      const uint32_t pnode = c.prev_node;
      if (std::find(ni.prev.begin(), ni.prev.end(), pnode) == ni.prev.end()) ni.prev.push_back(pnode);
      node = pnode;
      slot = c.beam;
    }
  }
  // publishResult: prev lists sorted by (boundary, position) = node id; ids from 1 in node order
  int32_t nextId = 1;
  for (auto& kv : info_) {
    std::sort(kv.second.prev.begin(), kv.second.prev.end());
    kv.second.id = nextId++;
  }
  auto idOf = [&](uint32_t node) -> int32_t {
    auto it = info_.find(node);
    return it == info_.end() ? 0 : it->second.id;
  };

  std::string& printer = printer_;
  if (!comment.empty()) {
    put(printer, "# ");
    put(printer, comment);
    printer += '\n';
  } else {
    put(printer, "# MA-SCORE\t");
    for (int32_t i = 0; i < outputN && i < beam; ++i) {
      const jppgpu_beam_slot& bel = beams[(uint64_t)eos * beam + i];
      if (isFake(bel)) break;
      put(printer, "rank");
      putInt(printer, i + 1);
      printer += ':';
      putFloat(printer, bel.total);
      printer += ' ';
    }
    printer += '\n';
  }

  OutputManager om(model_);
  for (auto& kv : info_) {
    const uint32_t node = kv.first;
    const NodeInfo& ni = kv.second;
    if (!om.locate(s, node, &walker_)) {
      return Status::InvalidState() << "failed to locate node: " << (s.nodes[node].start + 2) << ":" << node;
    }
    // std::max_element with `total1 > total2` as the ordering (lattice_format.cc:129-141) selects the
    // connection with the SMALLEST weighted score among those the N best paths use (first one on ties)
    auto total = [&](uint32_t slot) {
      const float* sc = cells + ((uint64_t)node * G + beams[(uint64_t)node * beam + slot].pad) * S;
      float t = 0;
      for (size_t i = 0; i < weights_.size(); ++i) t += sc[i] * weights_[i];
      return t;
    };
    uint32_t best = ni.slots[0];
    float bestTotal = total(best);
    for (size_t q = 1; q < ni.slots.size(); ++q) {
      float t = total(ni.slots[q]);
      if (bestTotal > t) {
        best = ni.slots[q];
        bestTotal = t;
      }
    }
    const float* scores = cells + ((uint64_t)node * G + beams[(uint64_t)node * beam + best].pad) * S;
    const jppgpu_node& nd = s.nodes[node];
    while (walker_.next()) {
      put(printer, "-\t");
      putInt(printer, ni.id);
      printer += '\t';
      for (size_t i = 0; i < ni.prev.size(); ++i) {
        putInt(printer, idOf(ni.prev[i]));
        if (i != ni.prev.size() - 1) printer += ';';
      }
      printer += '\t';
      const int32_t position = nd.start;  // cptr.boundary - 2
      putInt(printer, position);
      printer += '\t';
      putInt(printer, position + (nd.end - nd.start) - 1);
      printer += '\t';
      put(printer, escapeTab(flds_.surface[walker_]));
      printer += '\t';
      StringPiece canFrm = flds_.canonicForm[walker_];
      if (!canFrm.empty()) {
        put(printer, canFrm);
      } else {
        put(printer, flds_.baseform[walker_]);
        printer += '/';
        put(printer, flds_.reading[walker_]);
      }
      printer += '\t';
      put(printer, escapeTab(flds_.reading[walker_]));
      printer += '\t';
      put(printer, escapeTab(flds_.baseform[walker_]));
      printer += '\t';
      const int32_t* fb = walker_.features();
      int32_t ids[4];
      model_->dicToJuman(fb[1], fb[2], fb[4], fb[3], ids);
      put(printer, flds_.pos[walker_]);
      printer += '\t';
      putInt(printer, ids[0]);
      printer += '\t';
      put(printer, ifEmpty(flds_.subpos[walker_], "*"));
      printer += '\t';
      putInt(printer, ids[1]);
      printer += '\t';
      put(printer, ifEmpty(flds_.conjType[walker_], "*"));
      printer += '\t';
      putInt(printer, ids[2]);
      printer += '\t';
      put(printer, ifEmpty(flds_.conjForm[walker_], "*"));
      printer += '\t';
      putInt(printer, ids[3]);
      printer += '\t';
      KVListIterator features = flds_.features[walker_];
      while (features.next()) {
        put(printer, features.key());
        if (features.hasValue()) {
          printer += ':';
          put(printer, features.value());
        }
        printer += '|';
      }
      if (walker_.isSpecial()) {
        int32_t u = walker_.placeholder(NormalizedPlaceholderIdx);
        if (u != 0) {
          formatNormalizedFeature(printer, u);
          printer += '|';
        }
      }
      float totalScore = scores[0] * weights_[0];
      put(printer, "特徴量スコア:");
      putFloat(printer, totalScore);
      printer += '|';
      if (weights_.size() == 2) {  // have RNN
        float rnnScore = scores[1] * weights_[1];
        put(printer, "言語モデルスコア:");
        putFloat(printer, rnnScore);
        printer += '|';
        totalScore += rnnScore;
      }
      put(printer, "形態素解析スコア:");
      putFloat(printer, totalScore);
      printer += '|';
      put(printer, "ランク:");
      for (size_t i = 0; i < ni.ranks.size(); ++i) {
        putInt(printer, ni.ranks[i] + 1);
        if (i != ni.ranks.size() - 1) printer += ';';
      }
      printer += '\n';
    }
  }
  put(printer, "EOS\n");
  return Status::Ok();
}

}  // namespace jumanpp_amd
