#include "lattice_format.h"

#include <algorithm>
#include <cmath>
#include <cstdio>

namespace jumanpp_amd {

namespace {

inline void put(std::string& p, StringPiece s) { p.append(s.data(), s.size()); }
inline void putInt(std::string& p, long long v) {
  char buf[24];
  char* e = buf + sizeof(buf);
  char* q = e;
  unsigned long long u = v < 0 ? 0ull - (unsigned long long)v : (unsigned long long)v;
  do {
    *--q = (char)('0' + u % 10);
    u /= 10;
  } while (u != 0);
  if (v < 0) *--q = '-';
  p.append(q, (size_t)(e - q));
}
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

// the columns of a line that come from the entry row the walker stands on (lattice_format.cc:168-205): surface, canonic
// form (or baseform '/' reading), reading, baseform, the four grammar columns with their JUMAN ids, the feature list
void formatLatticeRow(const ModelImage& model, const JumandicFields& flds, const NodeWalker& walker, std::string& printer,
                      LatticeRowPieces* pieces) {
  const size_t at0 = printer.size();
  const StringPiece surface = flds.surface[walker], reading = flds.reading[walker], baseform = flds.baseform[walker];
  put(printer, escapeTab(surface));
  const size_t endS = printer.size();
  printer += '\t';
  const StringPiece canFrm = flds.canonicForm[walker];
  if (!canFrm.empty()) {
    put(printer, canFrm);
  } else {
    put(printer, baseform);
    printer += '/';
    put(printer, reading);
  }
  printer += '\t';
  const size_t atR = printer.size();
  put(printer, escapeTab(reading));
  const size_t endR = printer.size();
  printer += '\t';
  put(printer, escapeTab(baseform));
  const size_t endB = printer.size();
  printer += '\t';
  const int32_t* fb = walker.features();
  int32_t ids[4];
  model.dicToJuman(fb[1], fb[2], fb[4], fb[3], ids);   // conjForm and conjType are reversed in the entry row
  put(printer, flds.pos[walker]);
  printer += '\t';
  putInt(printer, ids[0]);
  printer += '\t';
  put(printer, ifEmpty(flds.subpos[walker], "*"));
  printer += '\t';
  putInt(printer, ids[1]);
  printer += '\t';
  put(printer, ifEmpty(flds.conjType[walker], "*"));
  printer += '\t';
  putInt(printer, ids[2]);
  printer += '\t';
  put(printer, ifEmpty(flds.conjForm[walker], "*"));
  printer += '\t';
  putInt(printer, ids[3]);
  printer += '\t';
  KVListIterator features = flds.features[walker];
  while (features.next()) {
    put(printer, features.key());
    if (features.hasValue()) {
      printer += ':';
      put(printer, features.value());
    }
    printer += '|';
  }
  if (pieces != nullptr) {
    pieces->s = (uint32_t)(endS - at0);
    pieces->c = (uint32_t)canFrm.size();
    pieces->r = (uint32_t)(endR - atR);
    pieces->b = (uint32_t)(endB - endR - 1);
    pieces->rest = (uint32_t)(printer.size() - endB - 1);
    pieces->total = (uint32_t)(printer.size() - at0);
    // a lone tab in one of the three escaped columns prints differently in the second column: not a table row
    auto lone = [](StringPiece x) { return x.size() == 1 && x[0] == '\t'; };
    pieces->tabField = lone(surface) || lone(reading) || lone(baseform);
  }
}

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
  const jppgpu_nbest_view* nbv = analysis.nbestOf(sentence, &local);
  if (nbv == nullptr && (v.beams == nullptr || v.cells == nullptr)) {
    return Status::InvalidState("the lattice format needs analyzeBatch(inputs, fullLattice = true)");
  }
  if (v.global_beam <= 0) return Status::NotImplemented("lattice format needs the global beam (score cells)");
  SentenceResult s = analysis.sentence(sentence);
  if (s.numNodes <= 3) {  // createdBoundaryCount() == 3: empty input
    printer_ = "EOS\n";
    return Status::Ok();
  }
  const int32_t beam = v.beam, G = v.global_beam, S = v.num_scorers;
  // lattice_format.cc:98-101: with auto-beam the number of printed paths is the sentence's beam
  const int32_t outputN = analysis.autoBeamSize(sentence) > 0 ? analysis.autoBeamSize(sentence) : topN_;
  // Two sources for the same lattice reads: the full arrays, or the n best paths gathered on the device
  // (jppgpu_result_fetch_nbest), indexed here by (node, slot).
  const jppgpu_beam_slot* beams = nullptr;
  const float* cells = nullptr;
  uint32_t eos = 0;
  const jppgpu_beam_slot fakeSlot{0xffff, 0xffff, 0.f, 0xffffffffu, 0};
  const jppgpu_nbest_item* nbFirst = nullptr;   // (n-best view) the records of this sentence's paths, path after path
  const uint64_t* nbPath = nullptr;
  if (nbv != nullptr) {
    if (nbv->n_best < std::min<int32_t>(beam, outputN)) return Status::InvalidState("n-best view holds fewer paths than the format prints");
    nbFirst = nbv->items;
    nbPath = nbv->path_first + (uint64_t)local * (uint32_t)nbv->n_best;
  } else {
    const uint64_t nb = v.node_base[local];
    beams = v.beams + nb * (uint64_t)beam;
    cells = v.cells + nb * (uint64_t)G * S;
    eos = s.numNodes - 1;
  }
  auto eosSlot = [&](int32_t i) -> const jppgpu_beam_slot& {
    if (nbv != nullptr) return i < nbv->n_best ? nbv->eos[(uint64_t)local * (uint32_t)nbv->n_best + (uint32_t)i] : fakeSlot;
    return beams[(uint64_t)eos * beam + i];
  };

  // LatticeFormatInfo::fillInfo (lattice_format.cc:13-43)
  for (uint32_t nd : order_) infoOf_[nd] = -1;
  order_.clear();
  info_.clear();
  if (infoOf_.size() < s.numNodes) infoOf_.resize(s.numNodes, -1);
  const int32_t maxN = std::min<int32_t>(std::min<int32_t>(beam, outputN), kMaxPaths);
  for (int32_t i = 0; i < maxN; ++i) {
    const jppgpu_beam_slot& el = eosSlot(i);
    if (isFake(el)) break;
    uint32_t node = el.prev_node;
    uint32_t slot = el.beam;
    // (n-best view) the connections of path i lie in path order, from the one the EOS slot points at back to BOS (k_nbest)
    uint64_t q = nbv != nullptr ? nbPath[i] : 0;
    const uint64_t qEnd = nbv != nullptr ? nbPath[i + 1] : 0;
    while (node >= 2 && node != 0xffffffffu) {
      const jppgpu_nbest_item* item = nullptr;
      if (nbv != nullptr) {
        if (q >= qEnd || nbFirst[q].node != node || nbFirst[q].slot != slot) {
          return Status::InvalidState("n-best view does not cover a path of the lattice format");
        }
        item = &nbFirst[q++];
      }
      if (node >= s.numNodes) return Status::InvalidState("lattice format: node outside the sentence");
      const jppgpu_beam_slot& c = item ? item->beam : beams[(uint64_t)node * beam + slot];
      if (isFake(c)) return Status::InvalidState("n-best view does not cover a path of the lattice format");
      int32_t at = infoOf_[node];
      if (at < 0) {
        at = infoOf_[node] = (int32_t)info_.size();
        info_.emplace_back();
        info_.back().node = node;
        order_.push_back(node);
      }
      NodeInfo& ni = info_[at];
      ni.ranks[ni.nRanks++] = (uint16_t)i;
      uint16_t k = 0;
      while (k < ni.nSlots && ni.slots[k] != slot) ++k;
      if (k == ni.nSlots) {
        ni.slots[k] = (uint16_t)slot;
        ni.items[k] = item;
        ni.nSlots++;
      }
      const uint32_t pnode = c.prev_node;
      k = 0;
      while (k < ni.nPrev && ni.prev[k] != pnode) ++k;
      if (k == ni.nPrev) ni.prev[ni.nPrev++] = pnode;
      node = pnode;
      slot = c.beam;
    }
  }
  // publishResult: prev lists sorted by (boundary, position) = node id; ids from 1 in node order
  std::sort(order_.begin(), order_.end());
  int32_t nextId = 1;
  for (uint32_t nd : order_) {
    NodeInfo& ni = info_[infoOf_[nd]];
    std::sort(ni.prev, ni.prev + ni.nPrev);
    ni.id = nextId++;
  }
  auto idOf = [&](uint32_t node) -> int32_t {
    return node < infoOf_.size() && infoOf_[node] >= 0 ? info_[infoOf_[node]].id : 0;
  };

  std::string& printer = printer_;
  if (!comment.empty()) {
    put(printer, "# ");
    put(printer, comment);
    printer += '\n';
  } else {
    put(printer, "# MA-SCORE\t");
    for (int32_t i = 0; i < outputN && i < beam; ++i) {
      const jppgpu_beam_slot& bel = eosSlot(i);
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
  for (uint32_t node : order_) {
    const NodeInfo& ni = info_[infoOf_[node]];
    const jppgpu_nbest_item* rec = ni.items[0];
    const jppgpu_node& nd = rec ? rec->info : s.nodes[node];
    if (!(rec ? om.locate(s, rec->info, rec->unk, &walker_) : om.locate(s, node, &walker_))) {
      return Status::InvalidState() << "failed to locate node: " << (nd.start + 2) << ":" << node;
    }
    auto cellsOf = [&](uint16_t k) -> const float* {
      if (nbv != nullptr) return ni.items[k]->cells;
      return cells + ((uint64_t)node * G + beams[(uint64_t)node * beam + ni.slots[k]].pad) * S;
    };
    // std::max_element with `total1 > total2` as the ordering (lattice_format.cc:129-141) selects the
    // connection with the SMALLEST weighted score among those the N best paths use (first one on ties)
    auto total = [&](uint16_t k) {
      const float* sc = cellsOf(k);
      // `total += s[i] * weights[i]`: one fused multiply-add per scorer in the reference's FMA build
      // (-march=native / haswell; the same contraction as in adjustBeamScores), so near-equal connections
      // compare as they do there
      float t = 0;
      for (size_t i = 0; i < weights_.size(); ++i) t = std::fma(sc[i], weights_[i], t);
      return t;
    };
    uint16_t best = 0;
    float bestTotal = total(0);
    for (uint16_t q = 1; q < ni.nSlots; ++q) {
      float t = total(q);
      if (bestTotal > t) {
        best = q;
        bestTotal = t;
      }
    }
    const float* scores = cellsOf(best);
    while (walker_.next()) {
      put(printer, "-\t");
      putInt(printer, ni.id);
      printer += '\t';
      for (uint16_t i = 0; i < ni.nPrev; ++i) {
        putInt(printer, idOf(ni.prev[i]));
        if (i + 1 != ni.nPrev) printer += ';';
      }
      printer += '\t';
      const int32_t position = nd.start;  // cptr.boundary - 2
      putInt(printer, position);
      printer += '\t';
      putInt(printer, position + (nd.end - nd.start) - 1);
      printer += '\t';
      formatLatticeRow(*model_, flds_, walker_, printer, nullptr);
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
      for (uint16_t i = 0; i < ni.nRanks; ++i) {
        putInt(printer, ni.ranks[i] + 1);
        if (i + 1 != ni.nRanks) printer += ';';
      }
      printer += '\n';
    }
  }
  put(printer, "EOS\n");
  return Status::Ok();
}

}  // namespace jumanpp_amd
