// TEST INFRASTRUCTURE ONLY -- never linked into or called by the product.
//
// Plain serial C++ restatement of the scoring core of the Juman++ analysis hot
// path: UTF-8 decode + character classes, entry rows, primitive/pattern
// feature hashing, unigram (T0) scores, the global-beam boundary sweep with
// bigram/trigram perceptron scores, right-node cutoff and per-node beams, and
// the top-1 back-trace.  Each function names the reference code it follows
// (paths relative to the ku-nlp/jumanpp tree).
//
// Pinning: `jpp_oracle check <model.img> <corpus.txt> <file.gold>` recomputes
// all of the above from the lattice node table of a golden file written by the
// REAL reference (oracle/_ref/ref_dump) and requires bit-identical patterns,
// T0 scores, global beams, beams (structure + float bits) and paths.  The
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
#include <fstream>
#include <numeric>
#include <string>
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

// ------------------------------------------------------------- the sweep ----
struct Slot { u16 left, beam; float total; int pb, pr; bool live; };

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

int main(int argc, char** argv) {
  if (argc != 5) { fprintf(stderr, "usage: jpp_oracle check|time model.img corpus.txt file.gold\n"); return 2; }
  bool timing = std::string(argv[1]) == "time";
  Model m = loadModel(argv[2]);
  std::vector<std::string> lines; { std::ifstream f(argv[3]); std::string l; while (std::getline(f, l)) lines.push_back(l); }
  Gold g = loadGold(argv[4]);
  if (g.nsc != 1) { fprintf(stderr, "jpp_oracle restates the perceptron path only (golden has %u scorers)\n", g.nsc); return 2; }
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
    for (size_t b = 0; b < nb; ++b) {
      GBnd& B = S.bnds[b];
      P[b].resize(B.R); T0[b].resize(B.R); beams[b].assign(B.R, std::vector<Slot>(g.beam, Slot{0, 0, 0.f, 0, 0, false}));
      for (u32 r = 0; r < B.R; ++r) {
        if (b < 2) { P[b][r].assign(NP, (u64)(u32)kBOS); continue; }  // LatticeConstructionContext::addBos lattice_builder.cc:173-179
        i32 row[16]; entryRow(m, B.nodes[r], row);
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
    beams[0][0][0] = Slot{0, 0, 0.f, -1, -1, true};
    beams[1][0][0] = Slot{0, 0, 0.f, 0, 0, true};
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
        for (int i = 0; i < c; ++i) { float v = pres[i][t]; v += 0.f; v += gsc[i]; tot[i] = v; }  // copyT0Scores(head, 0) :411-424
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
            float v = res; v += T0[b][t]; v += gsc[i]; tot[i] = v;    // copyT0Scores(tail, t0Score)
          }
        }
        // makeT0Beam :426-469 (std::sort on indices, like the reference)
        std::vector<u32> idx(cnt); std::iota(idx.begin(), idx.end(), 0);
        std::sort(idx.begin(), idx.end(), [&](u32 a, u32 bb) { return tot[a] > tot[bb]; });
        auto& row = beams[b][t];
        for (u32 q = 0; q < g.beam; ++q) {
          if (q < (u32)cnt) row[q] = Slot{gl[idx[q]], gs[idx[q]], tot[idx[q]], lnode[idx[q]].first, lnode[idx[q]].second, true};
          else row[q].live = false;
        }
        if (!timing) {
          const GNode& gn = B.nodes[t];
          if (G > 0 && (gn.kept != 0) != kept) fail(si, "kept", (int)b, (int)t);
          for (u32 q = 0; q < g.beam; ++q) {
            const auto& gsl = gn.beam[q];
            if ((gsl.valid != 0) != row[q].live) { fail(si, "beam liveness", (int)b, (int)t); continue; }
            if (!row[q].live) continue;
            if (gsl.cp[1] != row[q].left || gsl.cp[3] != row[q].beam || gsl.prev[0] != row[q].pb || gsl.prev[1] != row[q].pr ||
                memcmp(&gsl.total, &row[q].total, 4)) fail(si, "beam slot", (int)b, (int)t);
          }
        }
      }
    }
    // top-1 path (AnalysisPath::fillIn  src/core/analysis/analysis_result.cc:25-76)
    if (!timing) {
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
