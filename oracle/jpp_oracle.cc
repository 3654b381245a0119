// TEST INFRASTRUCTURE ONLY -- never linked into or called by the product.
//
// Plain serial C++ restatement of the scoring core of the Juman++ analysis hot
// path: UTF-8 decode + character classes, entry rows, primitive/pattern
// feature hashing, unigram (T0) scores, the global-beam boundary sweep with
// bigram/trigram perceptron scores, right-node cutoff and per-node beams, and
// the top-1 back-trace; and, for models with an RNN part, the RNNLM re-ranker:
// RNN word ids, the RnnIdContainer path merge, Mikolov NCE contexts and scores
// with the maxent hash, adjustBeamScores and remakeEosBeam.  Each function
// names the reference code it follows (paths relative to the ku-nlp/jumanpp tree).
//
// Pinning: `jpp_oracle check <model.img> <corpus.txt> <file.gold>` recomputes
// all of the above from the lattice node table of a golden file written by the
// REAL reference (oracle/_ref/ref_dump) and requires bit-identical patterns,
// T0 scores, global beams, beams (structure + float bits), perceptron score
// cells and paths; RNN score cells, RNN-adjusted totals and the re-made EOS
// beam within the 1e-4 float contract (the reference computes them with Eigen's
// vectorised exp and reductions, which a scalar loop matches only to ulps).  The
// lattice *construction* (dictionary walk, UNK makers) is not restated here:
// for that part the executable oracle is the reference itself (oracle/_ref),
// see DESIGN.md section 5.  Uses libstdc++'s own std::nth_element/std::sort,
// i.e. the same library routines the reference calls.
//
//   jpp_oracle check <model.img> <corpus.txt> <gold>   -> exit 0 iff everything matches
//   jpp_oracle time  <model.img> <corpus.txt> <gold>   -> sentences/s of the restated scoring core
#include <algorithm>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <cmath>
#include <fstream>
#include <map>
#include <numeric>
#include <set>
#include <string>
#include <tuple>
#include <vector>

typedef uint8_t u8;
typedef uint16_t u16;
typedef uint32_t u32;
typedef uint64_t u64;
typedef int32_t i32;

// ---------------------------------------------------------------- model ----
struct Prim { int kind, a, b; };
struct Comp { int cond; std::vector<int> t, f; };
struct Model {
  std::vector<u8> entryData;
  std::vector<float> weights;
  u32 wmask = 0;
  int numFeatures = 0;
  std::vector<Prim> prims;
  std::vector<Comp> comps;
  std::vector<std::vector<int>> patterns;
  std::vector<std::vector<int>> uni, bi, tri;  // {index, pattern refs...}
  struct Unk { int type, cls, tmpl, prio, ph; u32 replace; };
  std::vector<Unk> unks;
  // RNN part (RnnScorerGbeamFactory::load  src/core/analysis/rnn_scorer_gbeam.cc:426-470)
  struct Rnn {
    bool present = false;
    u32 E = 0, order = 0; u64 msize = 0, vsize = 0;
    float nce = 0; i32 unkId = 0; float unkConst = 0, unkLen = 0, wPerc = 1, wRnn = 1;
    std::vector<u32> fields, known, unk;
    std::vector<float> W, emb, nceEmb, maxent, bos;
  } rnn;
};

static std::vector<char> readFile(const char* p) {
  std::ifstream f(p, std::ios::binary);
  if (!f) { fprintf(stderr, "cannot open %s\n", p); exit(2); }
  return std::vector<char>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}

static Model loadModel(const char* path) {
  auto d = readFile(path);
  Model m;
  size_t pos = 8;
  auto rd32 = [&](const char* p) { i32 v; memcpy(&v, p, 4); return v; };
  for (;;) {
    pos = (pos + 7) & ~size_t(7);
    u32 tag, aux; u64 size;
    memcpy(&tag, &d[pos], 4); memcpy(&aux, &d[pos + 4], 4); memcpy(&size, &d[pos + 8], 8);
    pos += 16;
    if (tag == 0) break;
    const char* p = &d[pos];
    if (tag == 1) m.numFeatures = rd32(p);
    if (tag == 4) m.entryData.assign(p, p + size);
    if (tag == 5) { m.weights.resize(size / 4); memcpy(m.weights.data(), p, size); m.wmask = (u32)(size / 4 - 1); }
    if (tag == 6) {
      int n = rd32(p); const char* q = p + 4;
      for (int i = 0; i < n; ++i) {
        Model::Unk u{rd32(q), rd32(q + 4), rd32(q + 8), rd32(q + 12), rd32(q + 16), 0};
        int nr = rd32(q + 20); q += 24;
        for (int k = 0; k < nr; ++k, q += 4) u.replace |= 1u << rd32(q);
        m.unks.push_back(u);
      }
    }
    if (tag == 7) {
      const char* q = p;
      auto ints = [&]() { int n = rd32(q); q += 4; std::vector<int> v(n); for (auto& x : v) { x = rd32(q); q += 4; } return v; };
      int np = rd32(q); q += 4;
      for (int i = 0; i < np; ++i) { int kind = rd32(q); q += 4; auto r = ints(); m.prims.push_back({kind, r.size() > 0 ? r[0] : 0, r.size() > 1 ? r[1] : 0}); }
      int nc = rd32(q); q += 4;
      for (int i = 0; i < nc; ++i) { Comp c; c.cond = rd32(q); q += 4; c.t = ints(); c.f = ints(); m.comps.push_back(c); }
      int npat = rd32(q); q += 4;
      for (int i = 0; i < npat; ++i) { q += 4; m.patterns.push_back(ints()); }
      int nn = rd32(q); q += 4;
      for (int i = 0; i < nn; ++i) {
        int idx = rd32(q); q += 4; auto r = ints(); r.insert(r.begin(), idx);
        (r.size() == 2 ? m.uni : r.size() == 3 ? m.bi : m.tri).push_back(r);
      }
    }
    if (tag == 11) {
      auto floats = [&](std::vector<float>& v) { v.resize(size / 4); memcpy(v.data(), p, size); };
      auto words = [&](std::vector<u32>& v) { v.resize(size / 4); memcpy(v.data(), p, size); };
      if (aux == 1) words(m.rnn.known);
      if (aux == 2) words(m.rnn.unk);
      if (aux == 3) floats(m.rnn.W);
      if (aux == 4) floats(m.rnn.emb);
      if (aux == 5) floats(m.rnn.nceEmb);
      if (aux == 6) floats(m.rnn.maxent);
      if (aux == 100) {
        Model::Rnn& r = m.rnn; const char* q = p;
        auto get = [&](void* out, size_t n) { memcpy(out, q, n); q += n; };
        get(&r.E, 4); get(&r.order, 4); get(&r.msize, 8); get(&r.vsize, 8); get(&r.nce, 4); get(&r.unkId, 4);
        get(&r.unkConst, 4); get(&r.unkLen, 4); get(&r.wPerc, 4); get(&r.wRnn, 4);
        u32 nf; get(&nf, 4); r.fields.resize(nf); get(r.fields.data(), 4 * nf);
        r.present = true;
      }
    }
    pos += size;
  }
  return m;
}

// ----------------------------------------------------------- primitives ----
// FastHashRot::mix  src/util/fast_hash_rot.h:30-55, seeds src/util/seahash.h:15-17,
// src/core/impl/feature_impl_types.h:21-24
static inline u64 mix(u64 s, u64 x) { u64 v = (s ^ x) * 0x6eed0e9da4d94a4fULL; return (v << 32) | (v >> 32); }
static const u64 kSeed = 0x16f11fe89b0d677cULL, kPat = 0x7a11ed00000000ULL, kUni = 0x5123a31421fULL,
                 kBi = 0x5123a68442fULL, kTri = 0x51239ab41f1fULL;

// CodedBufferParser::readVarint64  src/util/coded_io.h:130-158
static u64 varint(const u8* p, size_t& pos) {
  u64 r = 0; int sh = 0;
  for (;;) { u32 b = p[pos++]; r |= (u64)(b & 0x7f) << sh; if (b < 0x80 || sh >= 63) break; sh += 7; }
  return r;
}

// chars::getCodeType  src/util/characters.cc:135-257 (ladder order matters)
static bool inList(u32 c, std::initializer_list<u32> l) { for (u32 x : l) if (x == c) return true; return false; }
static bool isBracket(u32 c) {
  static const u32 pairs[][2] = {{0x28,0x29},{0x5B,0x5B},{0x5D,0x5D},{0x7B,0x7B},{0x7D,0x7D},{0x0F3A,0x0F3D},{0x169B,0x169C},
    {0x2045,0x2046},{0x207D,0x207E},{0x208D,0x208E},{0x2308,0x230B},{0x2329,0x232A},{0x2768,0x2775},{0x27C5,0x27C6},
    {0x27E6,0x27EF},{0x2983,0x2998},{0x29D8,0x29DB},{0x29FC,0x29FD},{0x2E22,0x2E29},{0x3008,0x3011},{0x3014,0x301B},
    {0xFE59,0xFE5E},{0xFF08,0xFF09},{0xFF3B,0xFF3B},{0xFF3D,0xFF3D},{0xFF5B,0xFF5B},{0xFF5D,0xFF5D},{0xFF5F,0xFF60},{0xFF62,0xFF63}};
  for (auto& p : pairs) if (c >= p[0] && c <= p[1]) return true;
  return false;
}
static bool isSmallKana(u32 c) {
  return inList(c, {0x3041,0x3043,0x3045,0x3047,0x3049,0x3063,0x3083,0x3085,0x3087,0x308E,0x3095,0x3096,
                    0x30A1,0x30A3,0x30A5,0x30A7,0x30A9,0x30C3,0x30E3,0x30E5,0x30E7,0x30EE,0x30F5,0x30F6});
}
static i32 codeType(u32 c) {
  enum { SPACE=1, IPUNC=2, KANJI=4, FIGURE=8, PERIOD=0x10, MDOT=0x20, COMMA=0x40, ALPH=0x80, SYMBOL=0x100, KATA=0x200,
         HIRA=0x400, KFIG=0x800, SLASH=0x1000, COLON=0x2000, ERA=0x4000, CHOON=0x8000, HKANA=0x10000, BRACKET=0x20000,
         FEXC=0x40000, FDIGIT=0x80000, SMALL=0x100000 };
  if (inList(c, {0x20,0x3000,0xA0,0x1680,0x180E,0x202F,0x205F,0xFEFF}) || (c >= 0x2000 && c <= 0x200B)) return SPACE;
  if (c > 0x3000 && c < 0x3003) return IPUNC;
  if (c >= 0x337B && c <= 0x337E) return SYMBOL | ERA;
  if ((c > 0x303f && c < 0x30a0)) return isSmallKana(c) ? (HIRA | SMALL) : HIRA;
  if ((c > 0x309f && c < 0x30fb) || inList(c, {0x30FD,0x30FE,0x30FF})) return isSmallKana(c) ? (KATA | SMALL) : KATA;
  if (inList(c, {0x30FC,0x301C,0xFF5E,0x223C})) return HIRA | KATA | CHOON;
  if (c == 0xFF70) return HKANA | CHOON;
  if (c >= 0xFF66 && c <= 0xFF9F) return HKANA;
  if (c == 0xB7 || c == 0x30fb) return MDOT;
  if (c == 0x2C || c == 0xff0c) return COMMA;
  if (c == 0x2F || c == 0xff0f) return SLASH;
  if (c == 0x3A || c == 0xff1a) return COLON;
  if (c == 0xff0e) return PERIOD;
  if ((c > 0x2f && c < 0x3a) || (c > 0xff0f && c < 0xff1a)) return FIGURE;
  if (inList(c, {0x25cb,0x3007,0x96f6,0x4e00,0x4e8c,0x4e09,0x56db,0x4e94,0x516d,0x4e03,0x516b,0x4e5d})) return KFIG | KANJI;
  if (inList(c, {0x5341,0x767e,0x5343,0x4e07,0x5104,0x5146})) return KFIG | FDIGIT;
  if (inList(c, {0x6570,0x4F55,0x5E7E})) return FEXC | KANJI;
  if ((c >= 0x40 && c <= 0x5b) || (c >= 0x60 && c <= 0x7b) || (c >= 0xbf && c <= 0x100) || (c >= 0xff20 && c <= 0xff3b) ||
      (c >= 0xff40 && c <= 0xff5b) || (c >= 0x370 && c <= 0x3ff) || (c >= 0x400 && c <= 0x4ff)) return ALPH;
  if ((c > 0x4dff && c < 0xa000) || c == 0x3005 || c == 0x3007) return KANJI;
  if (isBracket(c)) return BRACKET;
  return SYMBOL;
}

// chars::getCodepoint / preprocessRawData  src/util/characters.h:86-131, characters.cc:259-276
static bool decode(const std::string& s, std::vector<u32>& cps, std::vector<i32>& cls) {
  size_t p = 0, n = s.size();
  while (p < n) {
    u8 b0 = (u8)s[p]; u32 cp; int l;
    auto cont = [&](size_t k) { return k < n && (((u8)s[k]) & 0xc0) == 0x80; };
    if (b0 > 0xef) { if ((b0 & ~7u) != 0xf0 || !cont(p+1) || !cont(p+2) || !cont(p+3)) return false;
      cp = ((b0 & 7u) << 18) | (((u8)s[p+1] & 0x3fu) << 12) | (((u8)s[p+2] & 0x3fu) << 6) | ((u8)s[p+3] & 0x3fu); l = 4; }
    else if (b0 > 0xdf) { if ((b0 & ~0xfu) != 0xe0 || !cont(p+1) || !cont(p+2)) return false;
      cp = ((b0 & 0xfu) << 12) | (((u8)s[p+1] & 0x3fu) << 6) | ((u8)s[p+2] & 0x3fu); l = 3; }
    else if (b0 > 0x7f) { if ((b0 & ~0x1fu) != 0xc0 || !cont(p+1)) return false;
      cp = ((b0 & 0x1fu) << 6) | ((u8)s[p+1] & 0x3fu); l = 2; }
    else { cp = b0; l = 1; }
    cps.push_back(cp); cls.push_back(codeType(cp)); p += l;
  }
  return true;
}

// --------------------------------------------------------------- golden ----
struct GNode {
  i32 eptr; u16 start, end; i32 unk[4]; std::vector<i32> entry; std::vector<u64> pat; float t0; u32 kept;
  struct Slot { u16 cp[4]; u16 prev[4]; float total; u32 valid; };
  std::vector<Slot> beam;
  std::vector<float> cells;
};
struct GBnd { u32 R, L; std::vector<std::pair<u16, u16>> ends; std::vector<u16> gbLeft, gbBeam; std::vector<float> gbScore; std::vector<GNode> nodes; };
struct GSent { u32 status, ncp; std::vector<GBnd> bnds; std::vector<std::pair<u16, u16>> path; };
struct Gold { u32 beam, gbeam, rcheck, rbeam, nsc, npat, esz, nph; std::vector<GSent> sents; };

static Gold loadGold(const char* path) {
  auto d = readFile(path);
  Gold g; size_t pos = 8;
  auto get = [&](void* out, size_t n) { memcpy(out, &d[pos], n); pos += n; };
  u32 hdr[9]; get(hdr, 36);
  g.beam = hdr[0]; g.gbeam = hdr[1]; g.rcheck = hdr[2]; g.rbeam = hdr[3]; g.nsc = hdr[4]; g.npat = hdr[5]; g.esz = hdr[6]; g.nph = hdr[7];
  for (u32 si = 0; si < hdr[8]; ++si) {
    GSent s; get(&s.status, 4); get(&s.ncp, 4);
    if (s.status == 0) {
      u32 nb; get(&nb, 4);
      for (u32 b = 0; b < nb; ++b) {
        GBnd bd; get(&bd.R, 4); get(&bd.L, 4);
        for (u32 l = 0; l < bd.L; ++l) { u16 x[2]; get(x, 4); bd.ends.push_back({x[0], x[1]}); }
        u32 ngb; get(&ngb, 4);
        for (u32 i = 0; i < ngb; ++i) { u16 x[2]; float sc; get(x, 4); get(&sc, 4); bd.gbLeft.push_back(x[0]); bd.gbBeam.push_back(x[1]); bd.gbScore.push_back(sc); }
        for (u32 r = 0; r < bd.R; ++r) {
          GNode n; get(&n.eptr, 4); get(&n.start, 2); get(&n.end, 2); get(n.unk, 16);
          n.entry.resize(g.esz); get(n.entry.data(), 4 * g.esz);
          n.pat.resize(g.npat); get(n.pat.data(), 8 * g.npat);
          get(&n.t0, 4); get(&n.kept, 4);
          n.beam.resize(g.beam);
          for (auto& sl : n.beam) { get(sl.cp, 8); get(sl.prev, 8); get(&sl.total, 4); get(&sl.valid, 4); }
          n.cells.resize(ngb * g.nsc); get(n.cells.data(), 4 * n.cells.size());
          bd.nodes.push_back(std::move(n));
        }
        s.bnds.push_back(std::move(bd));
      }
      u32 np; get(&np, 4);
      for (u32 i = 0; i < np; ++i) { u16 x[2]; get(x, 4); s.path.push_back({x[0], x[1]}); }
      u32 tl; get(&tl, 4); pos += tl; pos = (pos + 7) & ~size_t(7);
    }
    g.sents.push_back(std::move(s));
  }
  return g;
}

// ---------------------------------------------------------- T0 features ----
static const i32 kBOS = (i32)0x80000000, kEOS = (i32)0x80000002;

// PrimitiveFeatureContext::fillEntryBuffer  src/core/impl/feature_impl_types.h:128-148
// (UNK rows: UnkNodesContext::makePtr  src/core/analysis/unk_nodes_creator.cc:105-142)
static void entryRow(const Model& m, const GNode& n, i32* row) {
  int nf = m.numFeatures;
  if (n.eptr == kEOS) { for (int f = 0; f < nf; ++f) row[f] = kEOS; return; }
  i32 src = n.eptr >= 0 ? n.eptr : n.unk[0];
  size_t pos = (size_t)(src >> 1);
  for (int f = 0; f < nf; ++f) row[f] = (i32)varint(m.entryData.data(), pos);
  if (n.eptr < 0) {
    // which maker: a template pointer identifies it; anything else is a normalized dictionary entry
    u32 mask = 0; bool found = false;
    for (auto& u : m.unks) if (u.type != 5 && u.tmpl == n.unk[0]) { mask = u.replace; found = true; break; }
    if (!found) for (auto& u : m.unks) if (u.type == 5) mask = u.replace;
    for (int f = 0; f < nf; ++f) if ((mask >> f) & 1) row[f] = n.unk[1];
  }
}

// primitive features  src/core/impl/feature_impl_prim.h:62-236
static u64 primitive(const Prim& p, const GNode& n, const i32* row, const std::vector<u32>& cps, const std::vector<i32>& cls) {
  i32 N = (i32)cps.size();
  switch (p.kind) {
    case 1: return (u32)row[p.a];
    case 2: return ((u32)row[p.a] >> p.b) & 1u;
    case 3: return (n.eptr < 0 && n.eptr != kBOS && n.eptr != kEOS) ? (u64)(u32)n.unk[2 + p.a] : 0;
    case 6: return (u64)(n.end - n.start);
    case 8: { u64 v = ~u64(0);
      if (p.a > 0) { i32 pos = n.end + p.a - 1; if (pos < N) v = cps[pos]; }
      else { i32 pos = (i32)n.start + p.a; if (pos >= 0 && pos < N) v = cps[pos]; }
      return v; }
    case 7: { u64 v = 0;
      if (p.a == 0) { for (i32 i = n.start; i < n.end; ++i) v |= (u32)cls[i]; }
      else if (p.a > 0) { i32 pos = n.end + p.a - 1; if (pos < N) v = (u32)cls[pos]; }
      else { i32 pos = (i32)n.start + p.a; if (pos >= 0 && pos < N) v = (u32)cls[pos]; }
      return v; }
  }
  fprintf(stderr, "unsupported primitive kind %d\n", p.kind); exit(2);
}

// DynamicPatternFeatureImpl::apply  src/core/impl/feature_impl_pattern.h:28-41
// ExprComputeFeatureImpl / NoopComputeFeatureImpl  src/core/impl/feature_impl_compute.cc:12-26,59-63
static void patternsOf(const Model& m, const GNode& n, const i32* row, const std::vector<u32>& cps,
                       const std::vector<i32>& cls, std::vector<u64>& pat) {
  std::vector<u64> prim(m.prims.size());
  for (size_t i = 0; i < m.prims.size(); ++i) prim[i] = primitive(m.prims[i], n, row, cps, cls);
  pat.resize(m.patterns.size());
  for (size_t p = 0; p < m.patterns.size(); ++p) {
    u64 h = mix(mix(mix(kSeed, (u32)p), m.patterns[p].size()), kPat);
    for (int c : m.patterns[p]) {
      const Comp& cf = m.comps[c];
      if (cf.t.empty() && cf.f.empty()) h = mix(h, prim[cf.cond]);
      else for (int x : (prim[cf.cond] != 0 ? cf.t : cf.f)) h = mix(h, prim[x]);
    }
    pat[p] = h;
  }
}

// unigram sum: generated patternsAndUnigramsApply (4 round-robin partial sums, last row
// computeUnrolled4RawPerceptron, src/core/analysis/perceptron.h:46-72)
static float unigramScore(const Model& m, const std::vector<u64>& pat, bool lastRow) {
  float part[4]; size_t nu = m.uni.size();
  std::vector<float> w(nu);
  for (size_t u = 0; u < nu; ++u)
    w[u] = m.weights[(u32)mix(mix(mix(mix(kSeed, 3), (u32)m.uni[u][0]), kUni), pat[m.uni[u][1]]) & m.wmask];
  if (lastRow) { for (float& x : part) x = 0.f; for (size_t u = 0; u < nu; ++u) part[u & 3] += w[u]; }
  else { for (int j = 0; j < 4; ++j) part[j] = w[j]; for (size_t u = 4; u < nu; ++u) part[u & 3] += w[u]; }
  return part[0] + part[1] + part[2] + part[3];
}

// util::part_step / util::partition  src/util/stl_util.h:51-135: a quickselect that stops as soon as
// between minSize and maxSize of the best elements stand in front; makeT0Beam sorts only those.
// Restated over indices [lo, hi) of `v`; returns the index one past the selected front.
template <typename Cmp>
static size_t partStep(std::vector<u32>& v, size_t lo, size_t hi, Cmp comp) {
  const size_t sz = hi - lo;
  if (sz == 1) return hi;
  if (sz == 2) {
    if (comp(v[lo + 1], v[lo])) std::swap(v[lo + 1], v[lo]);
    return lo + 1;
  }
  if (sz == 3) {
    if (comp(v[lo + 1], v[lo])) std::swap(v[lo + 1], v[lo]);
    if (comp(v[lo + 2], v[lo + 1])) std::swap(v[lo + 2], v[lo + 1]);
    if (comp(v[lo + 1], v[lo])) std::swap(v[lo + 1], v[lo]);
    return lo + 1;
  }
  size_t pivot = hi - 1;               // the middle element goes to the last place and is the pivot
  std::swap(v[lo + sz / 2], v[pivot]);
  size_t a = lo, b = hi - 2;           // a walks up over elements better than the pivot, b collects the others
  while (a != b) {
    if (comp(v[a], v[pivot])) ++a;
    else { std::swap(v[a], v[b]); --b; }
  }
  if (comp(v[pivot], v[b])) std::swap(v[pivot], v[b]);
  else { ++b; std::swap(v[pivot], v[b]); }
  return b;
}

template <typename Cmp>
static size_t partitionFront(std::vector<u32>& v, size_t lo, size_t hi, Cmp comp, size_t minSize, size_t maxSize) {
  for (;;) {
    const size_t mid = partStep(v, lo, hi, comp);
    size_t sz = mid - lo;
    if (minSize <= sz && sz <= maxSize) return mid;
    if (sz > maxSize) { hi = mid; continue; }
    sz += 1;
    lo = mid + 1;
    minSize -= sz;
    maxSize -= sz;
    if (minSize == 0) return lo;
  }
}

// ScoreProcessor::makeT0Beam's ordering  score_processor.cc:426-441: best `beam` candidates first
static size_t beamOrder(std::vector<u32>& idx, const std::vector<float>& tot, u32 beam) {
  auto comp = [&](u32 a, u32 b) { return tot[a] > tot[b]; };
  size_t end = idx.size();
  const size_t bound = (size_t)beam * 4 / 3;
  if (idx.size() > bound) end = partitionFront(idx, 0, idx.size(), comp, beam, bound);
  std::sort(idx.begin(), idx.begin() + end, comp);
  return end;
}

// ------------------------------------------------------------- the sweep ----
struct Slot { u16 left, beam; float total; int pb, pr; bool live; int gi; };  // gi: index of (left, beam) in the boundary's global beam

static inline u32 sortable(float f) { u32 v; memcpy(&v, &f, 4); return (v & 0x80000000u) ? ~v : (v ^ 0x80000000u); }

struct Scorer {
  const Model& m;
  u32 biIdx(int k, const u64* p0, const u64* t1) const {
    return (u32)mix(mix(mix(mix(mix(kSeed, 4), (u32)m.bi[k][0]), kBi), p0[m.bi[k][1]]), t1[m.bi[k][2]]) & m.wmask; }
  u32 triIdx(int k, const u64* p0, const u64* t1, const u64* t2) const {
    return (u32)mix(mix(mix(mix(mix(mix(kSeed, 5), (u32)m.tri[k][0]), kTri), p0[m.tri[k][1]]), t1[m.tri[k][2]]), t2[m.tri[k][3]]) & m.wmask; }
  // W-way round-robin accumulation starting from 0, partial sums added left to right
  float rr(const std::vector<float>& w, int W) const {
    std::vector<float> f(W, 0.f);
    for (size_t k = 0; k < w.size(); ++k) f[k % W] += w[k];
    float s = f[0]; for (int j = 1; j < W; ++j) s += f[j];
    return s;
  }
};


// ------------------------------------------------------------ RNN re-rank ----
// exact-match lookup in a darts-clone double array (DoubleArray::traversal().step == Ok,
// src/core/dic/darts_trie.cc; unit layout of the darts-clone library the reference vendors)
static bool daFind(const std::vector<u32>& units, const std::string& key, i32& value) {
  if (units.empty()) return false;
  auto off = [](u32 u) { return (u >> 10) << ((u & (1u << 9)) >> 6); };
  u32 id = 0, unit = units[0];
  for (unsigned char c : key) {
    id ^= off(unit) ^ c;
    if (id >= units.size()) return false;
    unit = units[id];
    if ((unit & ((1u << 31) | 0xFFu)) != c) return false;
  }
  if (((unit >> 8) & 1) == 0) return false;
  value = (i32)(units[id ^ off(unit)] & 0x7fffffffu);
  return true;
}

// a ConnectionPtr by value: ConnPtrHasher's equality compares the fields, and `previous` is a
// function of (boundary, left, beam)  (rnn_id_resolver.h:88-103)
struct Conn {
  int b, r, left, beam;
  bool operator<(const Conn& o) const { return std::tie(b, r, left, beam) < std::tie(o.b, o.r, o.left, o.beam); }
};

static const u64 kPrimes[36] = {  // src/rnn/mikolov_rnn.h:18-25
    108641969, 116049371, 125925907, 133333309, 145678979, 175308587, 197530793, 234567803, 251851741,
    264197411, 330864029, 399999781, 407407183, 459258997, 479012069, 545678687, 560493491, 607407037,
    629629243, 656789717, 716048933, 718518067, 725925469, 733332871, 753085943, 755555077, 782715551,
    790122953, 812345159, 814814293, 893826581, 923456189, 940740127, 953085797, 985184539, 990122807};

struct RnnPass {
  const Model& m;
  const GSent& S;
  const std::string& line;
  const std::vector<u32>& boff;                               // byte offset of every codepoint (+ end)
  const std::vector<std::vector<std::vector<i32>>>& rows;     // entry rows [b][r]
  std::vector<std::vector<std::vector<Slot>>>& beams;
  const std::vector<std::vector<Conn>>& gbeams;               // global beam of every boundary: {left node b, r, left index, its beam slot}
  const std::vector<std::vector<std::vector<float>>>& cell0;  // perceptron score cells [b][r][gi]
  std::vector<std::vector<std::vector<float>>>& cell1;        // out: rnn score cells [b][r][gi]

  struct Node { i32 id, idx, boundary, length; int prev, nextInBnd; u64 hash; bool published; };
  struct Score { Conn lat; int next, rnn; };
  struct Bnd { int node = -1, scores = -1, nodeCnt = 0, scoreCnt = 0; };
  struct Coord { int boundary, length; i32 id; };
  std::vector<Node> nodes;
  std::vector<Score> scores;
  std::vector<Bnd> bnds;
  std::map<std::tuple<int, int, i32>, int> crdCache;
  std::map<Conn, int> ptrCache;
  std::map<std::pair<int, int>, Coord> nodeCache;

  Conn previousOf(const Conn& c) const {
    auto e = S.bnds[c.b].ends[c.left];
    const Slot& sl = beams[e.first][e.second][c.beam];
    return Conn{e.first, e.second, sl.left, sl.beam};
  }

  // RnnIdResolver::reprOf + RnnIdContainer::resolveId  rnn_id_resolver.cc:157-171,291-323
  // (RnnReprBuilder: addInt = varint of the u32, addString = bytes + varint(1), rnn_id_resolver.h:22-35)
  const Coord& resolveId(const Conn& c) {
    auto it = nodeCache.find({c.b, c.r});
    if (it != nodeCache.end()) return it->second;
    const GNode& n = S.bnds[c.b].nodes[c.r];
    std::string repr;
    for (u32 f : m.rnn.fields) {
      i32 v = rows[c.b][c.r][f];
      if (v >= 0) { u32 x = (u32)v; while (x >= 0x80) { repr.push_back((char)(x | 0x80)); x >>= 7; } repr.push_back((char)x); }
      else { repr.append(line, boff[n.start], boff[n.end] - boff[n.start]); repr.push_back((char)1); }
    }
    i32 id;
    if (!daFind(n.eptr >= 0 ? m.rnn.known : m.rnn.unk, repr, id)) id = m.rnn.unkId;
    return nodeCache[{c.b, c.r}] = Coord{c.b, (int)(n.end - n.start), id};
  }

  void addScore(int node, const Conn& c) {  // :277-285
    Bnd& b = bnds[nodes[node].boundary];
    scores.push_back(Score{c, b.scores, node});
    b.scores = (int)scores.size() - 1;
    b.scoreCnt += 1;
  }

  // RnnIdContainer::addPrevChain  :206-251 -- returns (first, last) of the chain of rnn nodes for `c`;
  // a hash hit attaches the connection to the NEWEST node of the coordinate (it->second), not to the
  // node whose hash matched: that is what the reference does and what the scores depend on
  std::pair<int, int> addPrevChain(const Conn& c) {
    auto ins = ptrCache.emplace(c, -1);
    if (!ins.second) return {ins.first->second, ins.first->second};
    auto span = addPrevChain(previousOf(c));
    int prev = span.second;
    Coord crd = resolveId(c);
    u64 data = (u64)(u32)crd.id | ((u64)(u32)crd.length << 32);
    u64 v = (nodes[prev].hash ^ data) * 0x6eed0e9da4d94a4fULL;   // FastHash1::mix  src/util/fast_hash.h:46-51
    u64 hash = v ^ (v >> 32);
    auto it = crdCache.find(std::make_tuple(crd.boundary, crd.length, crd.id));
    if (it != crdCache.end()) {
      for (int cached = it->second; cached >= 0; cached = nodes[cached].nextInBnd) {
        if (nodes[cached].hash == hash) {
          ptrCache[c] = it->second;
          addScore(it->second, c);
          return {it->second, it->second};
        }
      }
    }
    nodes.push_back(Node{crd.id, -1, c.b, crd.length, prev, -1, hash, false});
    int fresh = (int)nodes.size() - 1;
    ptrCache[c] = fresh;
    return {span.first, fresh};
  }

  void addPath(Conn c) {  // :253-275
    auto path = addPrevChain(c);
    for (int last = path.second; last != path.first; last = nodes[last].prev) {
      Bnd& b = bnds[nodes[last].boundary];
      nodes[last].idx = b.nodeCnt;
      nodes[last].nextInBnd = b.node;
      nodes[last].published = true;
      b.node = last;
      b.nodeCnt += 1;
      addScore(last, c);
      Coord crd = resolveId(c);
      crdCache[std::make_tuple(crd.boundary, crd.length, crd.id)] = last;
      c = previousOf(c);
    }
  }

  // MikolovRnnImplParallel::computeNewContext  src/rnn/mikolov_rnn_impl.h:202-215
  void newContext(const float* in, const float* emb, float* out) const {
    const u32 E = m.rnn.E;
    for (u32 i = 0; i < E; ++i) {
      float acc = 0.f;
      for (u32 k = 0; k < E; ++k) acc += m.rnn.W[(size_t)i * E + k] * in[k];
      acc += emb[i];
      out[i] = 1.0f / (1.0f + std::exp(-acc));
    }
  }

  // MikolovIndexCalculator::calcIndices / MikolovScoreCalculator::calcScoresN  mikolov_rnn_impl.h:21-131
  float maxent(const std::vector<i32>& ctx, i32 word) const {
    const u64 hashMax = m.rnn.msize - m.rnn.vsize;
    float res = 0.f;
    for (size_t i = 0; i <= ctx.size(); ++i) {
      u64 x = kPrimes[0] * kPrimes[1];
      for (size_t j = 1; j <= i; ++j) x += kPrimes[(i * kPrimes[j] + j) % 36] * ((u64)(int64_t)ctx[j - 1] + 1);
      u64 idx = ((x % hashMax) + (u64)(int64_t)word) % hashMax;
      res = i == 0 ? m.rnn.maxent[idx] : res + m.rnn.maxent[idx];
    }
    return res;
  }

  void run(float wPerc, float wRnn) {
    const int nb = (int)S.bnds.size(), eos = nb - 1;
    const u32 E = m.rnn.E;
    // RnnIdContainer::reset + addBos  :325-363
    bnds.assign(nb, Bnd{});
    nodes.push_back(Node{0, 0, 0, 0, -1, -1, 0, true});
    nodes.push_back(Node{0, 0, 1, 0, 0, -1, 0xdeadbeef0000ULL, true});
    bnds[1].node = 1; bnds[1].nodeCnt = 1;
    crdCache[std::make_tuple(1, 0, 0)] = 1;
    nodeCache[{1, 0}] = Coord{1, 0, 0};
    nodeCache[{eos, 0}] = Coord{eos, 0, 0};
    // RnnIdResolver::resolveIdsAtGbeam  :173-195 (one fake connection per EOS global-beam entry)
    ptrCache[Conn{1, 0, 0, 0}] = 1;
    const auto& eg = gbeams[eos];
    for (size_t i = 0; i < eg.size(); ++i) addPath(Conn{eos, 0, eg[i].left, eg[i].beam});  // fakeConnection :365-375
    // GbeamRnnState::computeContext  rnn_scorer_gbeam.cc:142-157 (bos state: computeBosState :37-46)
    std::vector<std::vector<float>> ctx(nb);
    ctx[1].assign(E, 0.f);
    { std::vector<float> zero(E, 0.f); newContext(zero.data(), &m.rnn.emb[0], ctx[1].data()); }
    for (int b = 2; b < eos; ++b) {
      ctx[b].assign((size_t)bnds[b].nodeCnt * E, 0.f);
      for (int nd = bnds[b].node; nd >= 0; nd = nodes[nd].nextInBnd) {
        const Node& N = nodes[nd]; const Node& P = nodes[N.prev];
        size_t embId = N.id == -1 ? 0 : (size_t)N.id;
        newContext(&ctx[P.boundary][(size_t)P.idx * E], &m.rnn.emb[embId * E], &ctx[b][(size_t)N.idx * E]);
      }
    }
    // scoreBoundary + copyScoresToLattice  :159-267 (every maxent history slot holds prev->id, :171-188)
    for (int b = 2; b <= eos; ++b) {
      if (bnds[b].nodeCnt == 0) continue;
      for (int sc = bnds[b].scores; sc >= 0; sc = scores[sc].next) {
        const Node& N = nodes[scores[sc].rnn]; const Node& P = nodes[N.prev];
        float score;
        if (N.id == m.rnn.unkId) score = m.rnn.unkConst + m.rnn.unkLen * N.length;
        else {
          size_t embId = N.id == -1 ? 0 : (size_t)N.id;
          const float* c = &ctx[P.boundary][(size_t)P.idx * E]; const float* e = &m.rnn.nceEmb[embId * E];
          float dot = 0.f; for (u32 k = 0; k < E; ++k) dot += e[k] * c[k];
          std::vector<i32> hist(m.rnn.order - 1, P.id);
          dot += maxent(hist, N.id);
          score = dot - m.rnn.nce;
        }
        const Conn& lat = scores[sc].lat;
        for (size_t i = 0; i < gbeams[b].size(); ++i)
          if (gbeams[b][i].left == lat.left && gbeams[b][i].beam == lat.beam) cell1[b][lat.r][i] = score;
      }
    }
    // ScoreProcessor::adjustBeamScores  src/core/analysis/score_processor.cc:521-549
    for (int b = 3; b < nb; ++b)
      for (const Conn& el : gbeams[b]) {
        Slot& e = beams[el.b][el.r][el.beam];
        // `localScore += scores.at(i) * scoreWeights.at(i)`: one fused multiply-add per scorer on an FMA
        // target (GCC contracts it; vfmadd231ss in the reference's -march=haswell/native object code)
        float local = 0.f;
        local = std::fma(cell0[el.b][el.r][e.gi], wPerc, local);
        local = std::fma(cell1[el.b][el.r][e.gi], wRnn, local);
        local += beams[e.pb][e.pr][e.beam].total;
        e.total = local;
      }
    // ScoreProcessor::remakeEosBeam  :551-576 (+ makeT0Beam :426-469)
    const int G = (int)eg.size();
    std::vector<float> full(G);
    for (int i = 0; i < G; ++i) {
      float beamScore = beams[eg[i].b][eg[i].r][eg[i].beam].total;
      float local = 0.f;
      local = std::fma(cell0[eos][0][i], wPerc, local);
      local = std::fma(cell1[eos][0][i], wRnn, local);
      full[i] = local + beamScore;
    }
    std::vector<u32> idx(G); std::iota(idx.begin(), idx.end(), 0);
    const size_t have = beamOrder(idx, full, (u32)beams[eos][0].size());
    auto& row = beams[eos][0];
    for (size_t q = 0; q < row.size(); ++q) {
      if (q < have) { const Conn& el = eg[idx[q]]; row[q] = Slot{(u16)el.left, (u16)el.beam, full[idx[q]], el.b, el.r, true, (int)idx[q]}; }
      else row[q].live = false;
    }
  }
};

static size_t G_eos(const std::vector<std::vector<Conn>>& gb, size_t nb) { return gb[nb - 1].size(); }

int main(int argc, char** argv) {
  if (argc != 5) { fprintf(stderr, "usage: jpp_oracle check|time model.img corpus.txt file.gold\n"); return 2; }
  bool timing = std::string(argv[1]) == "time";
  Model m = loadModel(argv[2]);
  std::vector<std::string> lines; { std::ifstream f(argv[3]); std::string l; while (std::getline(f, l)) lines.push_back(l); }
  Gold g = loadGold(argv[4]);
  if (g.nsc > 2 || (g.nsc == 2 && !m.rnn.present)) { fprintf(stderr, "golden has %u scorers, the model %s RNN part\n", g.nsc, m.rnn.present ? "an" : "no"); return 2; }
  if (g.nsc == 2 && (m.rnn.order < 1 || m.rnn.order > 4 || m.rnn.msize <= m.rnn.vsize)) { fprintf(stderr, "unsupported RNN header\n"); return 2; }
  const bool rnn = g.nsc == 2;
  const float kTol = 1e-4f;  // relative to max(1, |reference|): the float contract for RNN scores
  auto close = [&](float a, float e) { return std::fabs(a - e) <= kTol * std::max(1.0f, std::fabs(e)); };
  Scorer sc{m};
  const int nbi = (int)m.bi.size(), ntri = (int)m.tri.size(), NP = (int)g.npat;
  long bad = 0, checked = 0;
  auto fail = [&](size_t s, const char* what, int b, int r) { if (++bad < 20) fprintf(stderr, "MISMATCH sent %zu %s b%d n%d\n", s, what, b, r); };
  auto t_start = std::chrono::steady_clock::now();
  size_t nsent = 0;
  for (size_t si = 0; si < g.sents.size() && si < lines.size(); ++si) {
    GSent& S = g.sents[si];
    std::vector<u32> cps; std::vector<i32> cls;
    bool ok = decode(lines[si], cps, cls) && lines[si].size() <= 4096;
    if (S.status != 0) { if (ok && !timing) { /* lattice failures are not restated */ } continue; }
    if (!ok || cps.size() != S.ncp) { fail(si, "decode", 0, 0); continue; }
    ++nsent;
    size_t nb = S.bnds.size();
    if (nb <= 3) continue;  // computeScoresGbeam returns early (analyzer_impl.cc:255-258)
    // all patterns / T0 of the sentence
    std::vector<std::vector<std::vector<u64>>> P(nb);
    std::vector<std::vector<float>> T0(nb);
    std::vector<std::vector<std::vector<Slot>>> beams(nb);
    std::vector<std::vector<std::vector<i32>>> rows(nb);
    std::vector<std::vector<Conn>> gbeams(nb);
    std::vector<std::vector<std::vector<float>>> cell0(nb), cell1(nb);
    for (size_t b = 0; b < nb; ++b) {
      GBnd& B = S.bnds[b];
      P[b].resize(B.R); T0[b].resize(B.R); beams[b].assign(B.R, std::vector<Slot>(g.beam, Slot{0, 0, 0.f, 0, 0, false, 0}));
      rows[b].resize(B.R); cell0[b].resize(B.R); cell1[b].resize(B.R);
      for (u32 r = 0; r < B.R; ++r) {
        if (b < 2) { P[b][r].assign(NP, (u64)(u32)kBOS); continue; }  // LatticeConstructionContext::addBos lattice_builder.cc:173-179
        i32 row[16]; entryRow(m, B.nodes[r], row);
        rows[b][r].assign(row, row + m.numFeatures);
        std::vector<u64> pat; patternsOf(m, B.nodes[r], row, cps, cls, pat);
        T0[b][r] = unigramScore(m, pat, r == B.R - 1);
        P[b][r].assign(pat.begin(), pat.begin() + NP);
        if (!timing) {
          ++checked;
          if (memcmp(row, B.nodes[r].entry.data(), 4 * m.numFeatures)) fail(si, "entry row", (int)b, (int)r);
          if (memcmp(P[b][r].data(), B.nodes[r].pat.data(), 8 * NP)) fail(si, "patterns", (int)b, (int)r);
          if (memcmp(&T0[b][r], &B.nodes[r].t0, 4)) fail(si, "T0", (int)b, (int)r);
        }
      }
    }
    // AnalyzerImpl::bootstrapAnalysis  analyzer_impl.cc:179-195
    beams[0][0][0] = Slot{0, 0, 0.f, -1, -1, true, 0};
    beams[1][0][0] = Slot{0, 0, 0.f, 0, 0, true, 0};
    for (size_t b = 2; b < nb; ++b) {
      GBnd& B = S.bnds[b];
      const u32 R = B.R;
      if (R == 0) continue;
      // ScoreProcessor::makeGlobalBeam  score_processor.cc:246-282 (+ BeamCandidate keys score_processor.h:81-115)
      std::vector<u64> keys;
      for (u32 l = 0; l < B.L; ++l) {
        auto& row = beams[B.ends[l].first][B.ends[l].second];
        for (u32 k = 0; k < g.beam; ++k) { if (!row[k].live) break; keys.push_back(((u64)sortable(row[k].total) << 32) | (l << 16) | k); }
      }
      std::sort(keys.begin(), keys.end(), std::greater<u64>());
      if (keys.size() > g.gbeam) keys.resize(g.gbeam);
      const int G = (int)keys.size();
      std::vector<u16> gl(G), gs(G); std::vector<float> gsc(G);
      std::vector<const u64*> t1(G), t2(G); std::vector<int> t1id(G);
      std::vector<std::pair<int, int>> lnode(G);
      int U = 0; std::vector<int> firstOf;
      for (int i = 0; i < G; ++i) {
        gl[i] = (u16)(keys[i] >> 16); gs[i] = (u16)keys[i];
        u32 bits = (u32)(keys[i] >> 32); bits = (bits & 0x80000000u) == 0 ? ~bits : (bits ^ 0x80000000u); memcpy(&gsc[i], &bits, 4);
        lnode[i] = {B.ends[gl[i]].first, B.ends[gl[i]].second};
        const Slot& sl = beams[lnode[i].first][lnode[i].second][gs[i]];
        t1[i] = P[lnode[i].first][lnode[i].second].data();
        t2[i] = P[sl.pb][sl.pr].data();                     // gatherT2 score_processor.cc:391-409
        int found = -1; for (int j = 0; j < i; ++j) if (gl[j] == gl[i]) { found = t1id[j]; break; }  // dedupT1 :363-377
        if (found >= 0) t1id[i] = found; else { t1id[i] = U++; firstOf.push_back(i); }
        gbeams[b].push_back(Conn{lnode[i].first, lnode[i].second, gl[i], gs[i]});
      }
      if (!timing) {
        if ((size_t)G != B.gbLeft.size()) fail(si, "gbeam size", (int)b, 0);
        else for (int i = 0; i < G; ++i) if (gl[i] != B.gbLeft[i] || gs[i] != B.gbBeam[i]) fail(si, "gbeam entry", (int)b, i);
      }
      // computeGbeamScores  score_processor.cc:284-361
      const int c = std::min<int>({(int)g.rcheck, (int)R, G});
      const u32 K = g.rcheck > 0 ? std::min<u32>(g.rbeam, R) : R;
      std::vector<std::vector<float>> pres(c, std::vector<float>(R));
      for (int i = 0; i < c; ++i)
        for (u32 t = 0; t < R; ++t) {  // computeT0Prescores :497-511 with generated applyBiStep2 / applyTriStep3
          const u64* p0 = P[b][t].data();
          std::vector<float> wb(nbi), wt(ntri);
          for (int k = 0; k < nbi; ++k) wb[k] = m.weights[sc.biIdx(k, p0, t1[i])];
          for (int k = 0; k < ntri; ++k) wt[k] = m.weights[sc.triIdx(k, p0, t1[i], t2[i])];
          float v = T0[b][t];
          v += sc.rr(wb, t == R - 1 ? 4 : 8);
          v += sc.rr(wt, 4);
          pres[i][t] = v;
        }
      std::vector<u32> order(R); std::iota(order.begin(), order.end(), 0);
      if (g.rcheck > 0 && R > g.rbeam) {  // makeT0cutoffBeam :471-495
        std::vector<float> cs(R);
        for (u32 t = 0; t < R; ++t) { float s = 0; for (int i = 0; i < c; ++i) s += pres[i][t]; cs[t] = s; }
        std::nth_element(order.begin(), order.begin() + g.rbeam, order.end(), [&](u32 a, u32 bb) { return cs[a] > cs[bb]; });
      }
      for (u32 op = 0; op < R; ++op) {
        u32 t = order[op]; bool kept = op < K;
        const u64* p0 = P[b][t].data();
        int cnt = kept ? G : c;
        std::vector<float> tot(cnt);
        cell0[b][t].assign(G, 0.f); cell1[b][t].assign(G, 0.f);
        for (int i = 0; i < c; ++i) { float v = pres[i][t]; v += 0.f; cell0[b][t][i] = v; v += gsc[i]; tot[i] = v; }  // copyT0Scores(head, 0) :411-424
        if (kept && G > c) {
          // applyBiTriFullKernel  src/core/impl/feature_impl_ngram_partial_kernels.h:19-111
          std::vector<float> biS(U);
          for (int u = 0; u < U; ++u) {
            const u64* tp = t1[firstOf[u]];
            std::vector<float> wb(nbi); for (int k = 0; k < nbi; ++k) wb[k] = m.weights[sc.biIdx(k, p0, tp)];
            biS[u] = sc.rr(wb, u == U - 1 ? 4 : 2);
          }
          for (int i = c; i < G; ++i) {
            std::vector<float> wt(ntri); for (int k = 0; k < ntri; ++k) wt[k] = m.weights[sc.triIdx(k, p0, t1[i], t2[i])];
            float res;
            if (i < G - 1) { float r1 = 0, r2 = 0; for (int k = 0; k < ntri; ++k) (k & 1 ? r2 : r1) += wt[k]; res = biS[t1id[i]] + r1 + r2; }
            else res = biS[t1id[i]] + sc.rr(wt, 4);
            float v = res; v += T0[b][t]; cell0[b][t][i] = v; v += gsc[i]; tot[i] = v;    // copyT0Scores(tail, t0Score)
          }
        }
        // makeT0Beam :426-469
        std::vector<u32> idx(cnt); std::iota(idx.begin(), idx.end(), 0);
        const size_t have = beamOrder(idx, tot, g.beam);
        auto& row = beams[b][t];
        for (u32 q = 0; q < g.beam; ++q) {
          if (q < (u32)have) row[q] = Slot{gl[idx[q]], gs[idx[q]], tot[idx[q]], lnode[idx[q]].first, lnode[idx[q]].second, true, (int)idx[q]};
          else row[q].live = false;
        }
        if (!timing) {
          const GNode& gn = B.nodes[t];
          if (G > 0 && (gn.kept != 0) != kept) fail(si, "kept", (int)b, (int)t);
          for (int i = 0; i < cnt; ++i)   // perceptron score cells of the connections that were scored
            if (memcmp(&gn.cells[(size_t)i * g.nsc], &cell0[b][t][i], 4)) fail(si, "perceptron cell", (int)b, (int)t);
          if (rnn && b == nb - 1) continue;   // the EOS beam is re-made after the RNN pass
          for (u32 q = 0; q < g.beam; ++q) {
            const auto& gsl = gn.beam[q];
            if ((gsl.valid != 0) != row[q].live) { fail(si, "beam liveness", (int)b, (int)t); continue; }
            if (!row[q].live) continue;
            if (gsl.cp[1] != row[q].left || gsl.cp[3] != row[q].beam || gsl.prev[0] != row[q].pb || gsl.prev[1] != row[q].pr ||
                (!rnn && memcmp(&gsl.total, &row[q].total, 4))) fail(si, "beam slot", (int)b, (int)t);
          }
        }
      }
    }
    bool topTie = false;
    if (rnn) {
      // AnalyzerImpl::computeScoresGbeam's scorer loop  analyzer_impl.cc:286-294
      std::vector<u32> boff; { std::vector<u32> c2; std::vector<i32> k2; size_t p = 0; const std::string& l = lines[si];
        while (p < l.size()) { boff.push_back((u32)p); u8 b0 = (u8)l[p]; p += b0 > 0xef ? 4 : b0 > 0xdf ? 3 : b0 > 0x7f ? 2 : 1; } boff.push_back((u32)l.size()); }
      RnnPass pass{m, S, lines[si], boff, rows, beams, gbeams, cell0, cell1};
      pass.run(m.rnn.wPerc, m.rnn.wRnn);
      if (!timing) {
        // totals and RNN cells are defined on the connections of the surviving EOS paths only
        const GNode& ge = S.bnds[nb - 1].nodes[0];
        std::set<std::tuple<int, int, int>> onPath; std::vector<std::tuple<int, int, int>> stack;
        for (u32 q = 0; q < g.beam; ++q) if (ge.beam[q].valid) stack.push_back(std::make_tuple((int)nb - 1, 0, (int)q));
        while (!stack.empty()) {
          auto key = stack.back(); stack.pop_back();
          int b = std::get<0>(key), r = std::get<1>(key), q = std::get<2>(key);
          if (b < 2 || !onPath.insert(key).second) continue;
          const auto& gsl = S.bnds[b].nodes[r].beam[q];
          stack.push_back(std::make_tuple((int)gsl.prev[0], (int)gsl.prev[1], (int)gsl.prev[2]));
          if (b == (int)nb - 1) continue;
          const Slot& sl = beams[b][r][q];
          if (!sl.live) { fail(si, "on-path slot not live", b, r); continue; }
          if (!close(sl.total, gsl.total)) fail(si, "rnn-adjusted total", b, r);
          if (!close(cell1[b][r][sl.gi], S.bnds[b].nodes[r].cells[(size_t)sl.gi * 2 + 1])) fail(si, "rnn cell", b, r);
        }
        // re-made EOS beam: same candidates and totals; candidates tied within the tolerance may swap ranks
        std::vector<std::tuple<int, int, float>> ref, dev;
        for (u32 q = 0; q < g.beam; ++q) {
          if (ge.beam[q].valid) ref.push_back(std::make_tuple((int)ge.beam[q].cp[1], (int)ge.beam[q].cp[3], ge.beam[q].total));
          const Slot& sl = beams[nb - 1][0][q];
          if (sl.live) dev.push_back(std::make_tuple((int)sl.left, (int)sl.beam, sl.total));
        }
        if (ref.size() != dev.size()) fail(si, "EOS beam size", (int)nb - 1, 0);
        float cutoff = 0.f; for (auto& x : ref) cutoff = (&x == &ref[0]) ? std::get<2>(x) : std::min(cutoff, std::get<2>(x));
        for (auto& x : ref) {
          bool found = false;
          for (auto& d : dev) if (std::get<0>(d) == std::get<0>(x) && std::get<1>(d) == std::get<1>(x)) { found = true; if (!close(std::get<2>(d), std::get<2>(x))) fail(si, "EOS total", (int)nb - 1, 0); }
          if (!found && !close(std::get<2>(x), cutoff)) fail(si, "EOS candidate missing", (int)nb - 1, 0);
        }
        for (u32 i = 0; i < (u32)G_eos(gbeams, nb); ++i)
          if (!close(cell1[nb - 1][0][i], ge.cells[(size_t)i * 2 + 1])) fail(si, "EOS rnn cell", (int)nb - 1, (int)i);
        topTie = ref.size() > 1 && close(std::get<2>(ref[0]), std::get<2>(ref[1]));
      }
    }
    // top-1 path (AnalysisPath::fillIn  src/core/analysis/analysis_result.cc:25-76)
    if (!timing && !topTie) {
      std::vector<std::pair<u16, u16>> path; int pb = (int)nb - 1, pr = 0; u32 slot = 0;
      while (pb >= 2 && beams[pb][pr][slot].live) { path.push_back({(u16)pb, (u16)pr}); Slot s = beams[pb][pr][slot]; pb = s.pb; pr = s.pr; slot = s.beam; }
      if (path != S.path) fail(si, "path", 0, 0);
    }
  }
  double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count();
  if (timing) { printf("{\"sentences\": %zu, \"seconds\": %.4f, \"sent_per_s\": %.1f}\n", nsent, secs, nsent / secs); return 0; }
  printf("jpp_oracle: %zu sentences, %ld nodes checked, %ld mismatches\n", nsent, checked, bad);
  return bad == 0 ? 0 : 1;
}
