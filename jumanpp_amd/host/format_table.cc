#include "format_table.h"

#include <chrono>
#include <cstring>
#include <thread>

#include "juman_format.h"
#include "lattice_format.h"
#include "output.h"

namespace jumanpp_amd {

namespace {
// the entry pointers of the dictionary: the index blob is the trie values' lists written one after the other
// (DicTrieBuilder: entryPtrBuffer.position() per key, impl::writePtrsAsDeltas; src/core/dic/entry_builder.cc:19-23,
// field_reader.h IntListTraversal) -- [count][delta ...] with the pointers of a list summed up from 0
bool collectEntryPointers(const jppgpu_model& m, std::vector<int32_t>* out) {
  VarintReader r(StringPiece((const char*)m.entry_ptrs, m.entry_ptrs_bytes), 0);
  const unsigned char* end = (const unsigned char*)m.entry_ptrs + m.entry_ptrs_bytes;
  while (r.p < end) {
    uint64_t cnt;
    if (!r.read(&cnt)) return false;
    int64_t ptr = 0;
    for (uint64_t q = 0; q < cnt; ++q) {
      uint64_t d;
      if (!r.read(&d)) return false;
      ptr += (int64_t)d;
      if (ptr < 0 || ptr > 0x7fffffff || (size_t)(ptr >> 1) >= m.entry_data_bytes) return false;
      out->push_back((int32_t)ptr);
    }
  }
  return true;
}
// Every entry of the dictionary (and the template entries of the UNK makers) rendered row by row: `render(walker, first,
// blob, &row)` appends the text of the row the walker stands on and fills the row record but for blob_off / the "last
// row" flag; false = the entry cannot be a table row.  slots: (EntryPtr >> 4) -> 1 + first row.
template <typename Row, typename Render>
Status renderEntryRows(const ModelImage* model, unsigned threads, const Render& render, std::vector<uint32_t>* slots,
                       std::vector<Row>* rowsOut, std::string* blobOut, size_t* entries) {
  OutputManager om(model);
  const jppgpu_model& m = model->cmodel();
  std::vector<int32_t> eptrs;
  if (!collectEntryPointers(m, &eptrs) || eptrs.empty())
    return Status::InvalidState("the entry-pointer index is not a sequence of pointer lists");
  // the template entries of the UNK makers are addressed by the spec, not through the trie
  for (size_t k = 0; k < model->numUnkMakers(); ++k) {
    const int32_t tp = model->unkMaker(k).pattern_ptr;
    if (tp >= 0 && (size_t)((uint32_t)tp >> 1) < m.entry_data_bytes) eptrs.push_back(tp);
  }
  const size_t nslots = m.entry_data_bytes / 8 + 1;
  slots->assign(nslots, 0);
  // render: every thread its share of the entries into a blob and a row list of its own
  if (threads == 0) threads = 1;
  if (threads > 32) threads = 32;
  struct Part {
    std::string blob;
    std::vector<Row> rows;
    std::vector<uint32_t> firstRow;   // per entry of the share: index into rows
    bool failed = false;
  };
  std::vector<Part> parts(threads);
  const size_t per = (eptrs.size() + threads - 1) / threads;
  auto work = [&](unsigned t) {
    Part& P = parts[t];
    const size_t lo = t * per, hi = std::min(eptrs.size(), lo + per);
    if (lo >= hi) return;
    P.blob.reserve((hi - lo) * 110);
    P.rows.reserve((hi - lo) + (hi - lo) / 8);
    NodeWalker w;
    SentenceResult none;
    jppgpu_unk nounk{};
    for (size_t i = lo; i < hi; ++i) {
      jppgpu_node nd{eptrs[i], 0, 0};
      if (!om.locate(none, nd, nounk, &w)) {
        P.failed = true;
        return;
      }
      P.firstRow.push_back((uint32_t)P.rows.size());
      bool first = true;
      while (w.next()) {
        Row r{};
        r.blob_off = (uint32_t)P.blob.size();
        if (!render(w, first, P.blob, &r)) {
          P.failed = true;
          return;
        }
        P.rows.push_back(r);
        first = false;
      }
      if (first) {   // an entry without a row cannot be printed
        P.failed = true;
        return;
      }
      P.rows.back().flags |= 2;
    }
  };
  std::vector<std::thread> pool;
  for (unsigned t = 1; t < threads; ++t) pool.emplace_back(work, t);
  work(0);
  for (auto& th : pool) th.join();
  size_t nrows = 0, nblob = 0;
  for (auto& P : parts) {
    if (P.failed) return Status::InvalidState("a dictionary entry could not be rendered");
    nrows += P.rows.size();
    nblob += P.blob.size();
  }
  if (nrows >= 0xfffffff0ull || nblob >= 0xfffffff0ull) return Status::NotImplemented("format table beyond 2^32 rows / bytes");
  rowsOut->clear();
  rowsOut->reserve(nrows);
  blobOut->clear();
  blobOut->reserve(nblob + 64);
  std::vector<int32_t> slotOwner(slots->size(), 0);
  for (unsigned t = 0; t < threads; ++t) {
    Part& P = parts[t];
    const uint32_t rowBase = (uint32_t)rowsOut->size(), blobBase = (uint32_t)blobOut->size();
    const size_t lo = t * per;
    for (size_t k = 0; k < P.firstRow.size(); ++k) {
      const size_t slot = (size_t)((uint32_t)eptrs[lo + k] >> 4);
      // one slot, one entry (an entry listed under two keys renders the same rows).  Two DIFFERENT entries whose rows
      // start within the same 8 bytes of the entry data cannot share a slot: the table says so and the caller keeps
      // the host formatter (the T0 memo does the same with its records, jppgpu_api.cc: collect_memo_seeds)
      if ((*slots)[slot] == 0) {
        (*slots)[slot] = 1 + rowBase + P.firstRow[k];
        slotOwner[slot] = eptrs[lo + k];
      } else if (slotOwner[slot] != eptrs[lo + k]) {
        return Status::NotImplemented("format table: two dictionary entries within 8 bytes of entry data");
      }
    }
    for (Row r : P.rows) {
      r.blob_off += blobBase;
      rowsOut->push_back(r);
    }
    blobOut->append(P.blob);
    std::string().swap(P.blob);
  }
  *entries = eptrs.size();
  return Status::Ok();
}

// formatNormalizedFeature as table literals: label and letters in the order it tests the bits (the letters come out of
// the function itself)
template <typename Table>
Status fillFlagLiterals(Table* v) {
  v->flag_placeholder = NormalizedPlaceholderIdx;
  std::string label;
  formatNormalizedFeature(label, 0);
  if (label.size() > sizeof(v->flag_label)) return Status::InvalidState("flag label too long");
  v->flag_label_len = (uint8_t)label.size();
  std::memcpy(v->flag_label, label.data(), label.size());
  // letter of every single bit; the order of the letters in a combined value is the order of the function's tests, found
  // by printing all bits at once
  std::string all;
  formatNormalizedFeature(all, 0xffff);
  const std::string letters = all.substr(label.size());
  if (letters.size() > 16) return Status::InvalidState("too many flag letters");
  v->n_flags = 0;
  for (char c : letters) {
    uint32_t bitOf = 0;
    for (uint32_t bit = 1; bit < 0x10000u; bit <<= 1) {
      std::string one;
      formatNormalizedFeature(one, (int32_t)bit);
      if (one.size() == label.size() + 1 && one[label.size()] == c) bitOf |= bit;
    }
    if (bitOf == 0) return Status::InvalidState("flag letter without a bit");
    v->flag_mask[v->n_flags] = bitOf;
    v->flag_char[v->n_flags] = c;
    v->n_flags++;
  }
  return Status::Ok();
}

// which of the replaceable strings an UNK maker prints from the input: its replace mask over the entry-row columns
template <typename Table>
Status fillMakerReplaces(const ModelImage* model, const int32_t* cols, int ncols, Table* v) {
  if (model->numUnkMakers() > 16) return Status::NotImplemented("more than 16 UNK makers");
  for (size_t k = 0; k < model->numUnkMakers(); ++k) {
    uint32_t mask = model->unkMaker(k).replace_mask;
    uint8_t rep = 0;
    for (int f = 0; f < ncols; ++f) {
      if (cols[f] >= 0 && ((mask >> cols[f]) & 1)) {
        rep |= (uint8_t)(1u << f);
        mask &= ~(1u << cols[f]);
      }
    }
    // a maker that overwrites another column (pos, features, ...) changes text the table has rendered already
    if (mask != 0) return Status::NotImplemented("an UNK maker replaces a field other than surface / reading / baseform / canonic form");
    v->maker_replaces[k] = rep;
  }
  return Status::Ok();
}
}  // namespace

Status JumanFormatTable::build(const ModelImage* model, unsigned threads) {
  const auto t0 = std::chrono::steady_clock::now();
  if (!model->hasIdMap()) return Status::InvalidState("model has no JUMAN id tables");
  OutputManager om(model);
  JumandicFields flds;
  JPPA_RETURN_IF_ERROR(flds.initialize(om));
  const int32_t cols[3] = {flds.surface.index(), flds.reading.index(), flds.baseform.index()};
  jppgpu_format_table& v = view_;
  std::memset(&v, 0, sizeof(v));
  v.struct_size = (uint32_t)sizeof(v);
  JPPA_RETURN_IF_ERROR(fillMakerReplaces(model, cols, 3, &v));
  auto render = [&](const NodeWalker& w, bool first, std::string& blob, jppgpu_format_row* r) -> bool {
    JumanRowPieces pc;
    formatJumanRow(*model, flds, w, first, blob, &pc);
    if (pc.pre > 0xffff || pc.s > 0xffff || pc.r > 0xffff || pc.b > 0xffff || pc.mid > 0xffff) return false;
    r->len_pre = (uint16_t)pc.pre;
    r->len_s = (uint16_t)pc.s;
    r->len_r = (uint16_t)pc.r;
    r->len_b = (uint16_t)pc.b;
    r->len_mid = (uint16_t)pc.mid;
    r->flags = pc.hasFeatures ? 1 : 0;
    r->len_feat = pc.feat;
    r->len_total = pc.total;
    return true;
  };
  JPPA_RETURN_IF_ERROR(renderEntryRows<jppgpu_format_row>(model, threads, render, &slots_, &rows_, &blob_, &entries_));
  v.slot_first_row = slots_.data();
  v.n_slots = slots_.size();
  v.rows = rows_.data();
  v.n_rows = rows_.size();
  v.blob = blob_.data();
  v.blob_bytes = blob_.size();
  // escapeForJumanOutput (juman_format.cc:42-54)
  v.n_escapes = 2;
  v.escape_from[0] = '\t';
  v.escape_len[0] = 2;
  std::memcpy(v.escape_to[0], "\\t", 2);
  v.escape_from[1] = ' ';
  v.escape_len[1] = 4;
  std::memcpy(v.escape_to[1], "\\\xe2\x90\xa3", 4);
  JPPA_RETURN_IF_ERROR(fillFlagLiterals(&v));
  v.eos_len = 4;
  std::memcpy(v.eos_text, "EOS\n", 4);
  const StringPiece err = JumanFormat::emptyResult();
  if (err.size() > sizeof(v.error_text)) return Status::InvalidState("error text too long");
  v.error_len = (uint8_t)err.size();
  std::memcpy(v.error_text, err.data(), err.size());
  buildMs_ = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  return Status::Ok();
}

Status LatticeFormatTable::build(const ModelImage* model, const std::vector<float>& scoreWeights, unsigned threads) {
  const auto t0 = std::chrono::steady_clock::now();
  if (!model->hasIdMap()) return Status::InvalidState("model has no JUMAN id tables");
  if (scoreWeights.empty() || scoreWeights.size() > 2) return Status::NotImplemented("lattice table: one or two score weights");
  OutputManager om(model);
  JumandicFields flds;
  JPPA_RETURN_IF_ERROR(flds.initialize(om));
  const int32_t cols[4] = {flds.surface.index(), flds.reading.index(), flds.baseform.index(), flds.canonicForm.index()};
  jppgpu_lattice_table& v = view_;
  std::memset(&v, 0, sizeof(v));
  v.struct_size = (uint32_t)sizeof(v);
  JPPA_RETURN_IF_ERROR(fillMakerReplaces(model, cols, 4, &v));
  auto render = [&](const NodeWalker& w, bool, std::string& blob, jppgpu_lattice_row* r) -> bool {
    LatticeRowPieces pc;
    formatLatticeRow(*model, flds, w, blob, &pc);
    if (pc.s > 0xffff || pc.c > 0xffff || pc.r > 0xffff || pc.b > 0xffff || pc.tabField) return false;
    r->len_s = (uint16_t)pc.s;
    r->len_c = (uint16_t)pc.c;
    r->len_r = (uint16_t)pc.r;
    r->len_b = (uint16_t)pc.b;
    r->len_rest = pc.rest;
    r->flags = 0;
    return true;
  };
  JPPA_RETURN_IF_ERROR(renderEntryRows<jppgpu_lattice_row>(model, threads, render, &slots_, &rows_, &blob_, &entries_));
  v.slot_first_row = slots_.data();
  v.n_slots = slots_.size();
  v.rows = rows_.data();
  v.n_rows = rows_.size();
  v.blob = blob_.data();
  v.blob_bytes = blob_.size();
  // escapeTab (lattice_format.cc:74-79)
  v.n_escapes = 1;
  v.escape_from[0] = '\t';
  v.escape_len[0] = 2;
  std::memcpy(v.escape_to[0], "\\t", 2);
  JPPA_RETURN_IF_ERROR(fillFlagLiterals(&v));
  auto lit = [](char* dst, size_t cap, uint8_t* len, const char* text) -> bool {
    const size_t n = std::strlen(text);
    if (n > cap) return false;
    std::memcpy(dst, text, n);
    *len = (uint8_t)n;
    return true;
  };
  const StringPiece err = JumanFormat::emptyResult();
  const std::string errText(err.data(), err.size());
  if (!(lit(v.head_text, sizeof(v.head_text), &v.head_len, "# MA-SCORE\t") && lit(v.rank_text, sizeof(v.rank_text), &v.rank_len, "rank") &&
        lit(v.feat_text, sizeof(v.feat_text), &v.feat_len, "特徴量スコア:") && lit(v.lm_text, sizeof(v.lm_text), &v.lm_len, "言語モデルスコア:") &&
        lit(v.total_text, sizeof(v.total_text), &v.total_len, "形態素解析スコア:") && lit(v.ranks_text, sizeof(v.ranks_text), &v.ranks_len, "ランク:") &&
        lit(v.eos_text, sizeof(v.eos_text), &v.eos_len, "EOS\n") && lit(v.error_text, sizeof(v.error_text), &v.error_len, errText.c_str())))
    return Status::InvalidState("a literal of the lattice format is longer than its table field");
  v.n_weights = (uint32_t)scoreWeights.size();
  v.weights[0] = scoreWeights[0];
  v.weights[1] = scoreWeights.size() > 1 ? scoreWeights[1] : 0.f;
  buildMs_ = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  return Status::Ok();
}

}  // namespace jumanpp_amd
