// The lattice (-s N) output format on the device (BASELINE configs[4] "lattice-format output"): the
// N best analyses of every sentence as the bytes the reference's LatticeFormat prints, assembled from a per-model table
// of rendered entry rows (include/jppgpu.h: jppgpu_lattice_table; built by host/lattice_table.cc).
//
// Reference behaviour reproduced (every literal comes from the table):
//   LatticeFormatInfo::fillInfo      src/jumandic/shared/lattice_format.cc:13-43   the N best paths walked back from the
//                                                                                 EOS beam; per lattice node on a path:
//                                                                                 its connections, ranks, previous nodes
//   LatticeNodeInfo::addElem/fixPrevs :250-270                                      ranks in path order, distinct
//                                                                                 previous nodes sorted by (boundary, position)
//   LatticeFormatInfo::publishResult :45-66                                        ids from 1 in (boundary, position) order
//   LatticeFormat::format            :83-242                                       "# MA-SCORE" line, per node and entry
//                                                                                 row one line; scores of the connection
//                                                                                 std::max_element picks under `total1 >
//                                                                                 total2` (= the smallest weighted total)
//   Printer << float                 util/printer.h (fmt "%g")                      jpp_fmtg.h
//
// Data flow: the host formatter of rounds 1-5 read the N best paths as 64 B per path and node over PCIe (0.2 MB per
// 220-codepoint sentence at N = 32) and rebuilt the per-node sets with a few thousand steps per sentence.  Here one
// wavefront per sentence does it where the lattice lies:
//   k_lat_count  lane = path: walks its path back from EOS and leaves, per lattice node, the set of paths through it
//                (= the ranks) and the set of its beam slots they use (= the distinct connections) in ONE word of HBM
//                (atomicOr), the connection with the smallest total in a second (atomicMin of an ordered key); then
//                numbers the marked nodes (ballot + popcount = publishResult's ids); then lane = node: a 48-byte record
//                of what the node's lines print -- first row, the connection's weighted scores, the ids of the previous
//                nodes, the ranks -- and the bytes of those lines;
//   k_lat_write  lane = node again: every lane prints the lines of its own node from its record -- ids, "%g" digits,
//                ranks, the entry-row text read from the blob eight bytes at a time -- into the wavefront's LDS window,
//                at the offset a wave scan of the byte counts gives it; the window leaves as whole dwords.
// (The first form printed ONE line at a time with all 64 lanes copying its pieces: 25 us per line -- every line waited
// for its own chain of dependent loads and for the table's literals, read byte by byte from HBM -- 19 ms of format
// kernels per 4 096-sentence batch at beam 32, more than the analysis.  profiles/r06_c_*.)
#ifndef JPP_K_LATFMT_H
#define JPP_K_LATFMT_H

#include "jpp_device.h"
#include "jpp_fmtg.h"
#include "k_format.h"
#include "k_lattice.h"

namespace jpp {

struct LatRow {
  u32 blob_off;
  u32 len_rest;
  u16 len_s, len_c, len_r, len_b;
  u32 flags;   // bit 1: last row of its entry
};
static_assert(sizeof(LatRow) == 20, "jppgpu_lattice_row");

// device copy of jppgpu_lattice_table
struct LatTable {
  const u32* slot_first_row;
  u64 n_slots;
  const LatRow* rows;
  u64 n_rows;
  const u8* blob;
  u8 maker_replaces[16];   // bit 0 / 1 / 2 / 3: surface / reading / baseform / canonic form print the input surface
  u8 n_escapes;
  u8 escape_from[4];
  u8 escape_len[4];
  u8 escape_to[4][8];
  i32 flag_placeholder;
  u8 flag_label_len;
  u8 flag_label[32];
  u8 n_flags;
  u32 flag_mask[16];
  u8 flag_char[16];
  u8 head_len, rank_len, feat_len, lm_len, total_len, ranks_len, eos_len, error_len;
  u8 head_text[16], rank_text[8], feat_text[32], lm_text[32], total_text[32], ranks_text[16], eos_text[16], error_text[32];
  u32 n_weights;
  float weights[2];
};

// What k_lat_count found out about a marked node, for k_lat_write: both kernels printed from the lattice at first, and
// each paid some ten random 128-byte lines per node for it (four of its beam slots, two of its score cells, its sets, the
// ids of its previous nodes) -- 8 GB per 8 192 sentences of 220 codepoints at N = 32 between them.  The record is read
// in order, 48 bytes a lane.
struct alignas(16) LatRec {
  u32 node;
  u32 row;        // the first row of its entry
  float f0, f1;   // the weighted scores of the chosen connection
  u32 pid[4];     // the ids of its previous nodes, ascending
  u32 ranks;      // the paths through it
  u32 flags;      // bits 0-2: how many of pid; kLatRecMany: more than four -- k_lat_write derives them again
  u32 bytes;      // of its lines
  u32 pad;
};
static_assert(sizeof(LatRec) == 48, "three 16-byte loads");
constexpr u32 kLatRecMany = 8u;

// Developer switches of the two kernels (their last argument; jppgpu_api.cc reads JPPGPU_DEV_LAT_WIN / JPPGPU_DEV_LAT_MANY_PREV):
// a smaller LDS window and the many-previous-nodes form for every node, so that the tests walk those paths -- windows
// split between nodes, a node beyond the window, every alignment of the flush -- with ordinary sentences ON THE DEVICE.
constexpr u32 kLatDevWinMask = 0xffffu;    // window bytes (0: the whole window)
constexpr u32 kLatDevManyPrev = 0x10000u;

// per-node scratch of the formatter (HBM, [total_nodes] each)
struct LatScratch {
  u64* mask;    // low half: paths (ranks) through the node; high half: beam slots of the node those paths use (both at
                // most 32: a path is an EOS beam slot, and the device keeps at most 32 slots per node) -- ONE atomic for both
  u64* best;    // min over those connections of (ordered total << 32 | path << 8 | slot)
  u32* id;      // publishResult's id, 0 = not on a path
  LatRec* rec;  // [node_base + id - 1] = the node with that id and what its lines print (the marked nodes in output order)
  u32* marked;  // [sentence] how many
};

__device__ __forceinline__ u32 lat_order_f32(float t) {
  u32 b;
  __builtin_memcpy(&b, &t, 4);
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}

__device__ __forceinline__ u32 lat_first_row(const LatTable& T, i32 eptr) {
  const u32 slot = ((u32)eptr >> 1) >> 3;
  if (eptr < 0 || slot >= T.n_slots) return ~0u;
  const u32 v = T.slot_first_row[slot];
  return v == 0 ? ~0u : v - 1;
}

// One lane's cursor into the output: `p` is where its text goes (WRITE = false only counts, p is unused).  Everything
// here is per lane -- different lanes print different lattice nodes.
template <bool WRITE, typename P = u8*>
struct LatOut {
  P p;      // (u8* into the output, or a typed pointer into the wavefront's LDS window)
  u64 n;
  __device__ __forceinline__ void ch(u8 c) {
    if (WRITE) p[n] = c;
    ++n;
  }
  __device__ __forceinline__ void num(u32 v) { n += u32_format(v, WRITE ? p + n : P(nullptr)); }
  __device__ __forceinline__ void flt(const GDigits& g) { n += g_emit(g, WRITE ? p + n : P(nullptr)); }
  // a literal of the table (the table's copy in LDS)
  __device__ __forceinline__ void lit(const u8* src, u32 len) {
    if (WRITE)
      for (u32 i = 0; i < len; ++i) p[n + i] = src[i];
    n += len;
  }
  // a run of bytes from HBM (the blob, the input text): read eight at a time (unaligned reads are the hardware's business)
  __device__ __forceinline__ void run(const u8* src, u32 len) {
    if (WRITE) {
      P d = p + n;
      u32 i = 0;
      for (; i + 8 <= len; i += 8) {
        u64 v;
        __builtin_memcpy(&v, src + i, 8);
#pragma unroll
        for (int k = 0; k < 8; ++k) d[i + k] = (u8)(v >> (8 * k));
      }
      for (; i < len; ++i) d[i] = src[i];
    }
    n += len;
  }
};

// What the lines of one lattice node print besides its entry rows (ONE lane; k_lat_count).  Returns false when the table
// has no row for it.  T: the table's copy in LDS (its pointers point into HBM).
__device__ __forceinline__ bool lat_node_facts(LatRec& r, const Batch& B, const Config& cfg, const LatTable& T, const LatScratch& S, u64 nb,
                                              u32 node, bool forceMany) {
  const int beam = cfg.beam, G = cfg.gbeam, NS = cfg.nscorers;
  const u64 gn = nb + node;
  const NodeInfo ni = B.node_info[gn];
  const u32 row = lat_first_row(T, ni.eptr < 0 ? B.node_aux[gn].tmpl : ni.eptr);
  if (row == ~0u) return false;
  const u64 both = load_l2(&S.mask[gn]);
  const u64 slots = both >> 32;
  const u32 bslot = (u32)(load_l2(&S.best[gn]) & 0xffu);
  const BeamSlot* beams = B.node_beam + gn * (u64)beam;
  // the scores of the chosen connection (once per node, printed on every row)
  const float* cell = B.node_cells + (gn * (u64)G + beams[bslot].pad) * (u64)NS;
  const bool two = T.n_weights == 2 && NS > 1;
  r.node = node;
  r.row = row;
  r.f0 = cell[0] * T.weights[0];
  r.f1 = two ? cell[1] * T.weights[1] : 0.f;
  r.ranks = (u32)both;
  // The distinct previous nodes, ascending.  At beam 32 the paths through a node use a dozen or two of its beam slots
  // (they differ EARLIER in the sentence), but those slots have one to three distinct left nodes: ONE pass over the
  // slots, four loads in flight, into a sorted set of four (a pass per distinct previous node -- the first form -- made a
  // few hundred dependent loads per node and the sentences with such nodes the tail of both kernels)
  u32 pv[4] = {~0u, ~0u, ~0u, ~0u};
  u32 nPrev = 0;
  bool manyPrev = false;
  for (u64 sm = slots; sm != 0;) {
    u32 a[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      a[q] = ~0u;
      if (sm != 0) {
        a[q] = beams[__builtin_ctzll(sm)].prev_node;
        sm &= sm - 1;
      }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const u32 x = a[q];
      if (x == ~0u || x == pv[0] || x == pv[1] || x == pv[2] || x == pv[3]) continue;
      if (nPrev == 4) {
        manyPrev = true;
        continue;
      }
      // insertion into the ascending set (the unused places hold ~0u, the largest value)
      if (x < pv[0]) {
        pv[3] = pv[2]; pv[2] = pv[1]; pv[1] = pv[0]; pv[0] = x;
      } else if (x < pv[1]) {
        pv[3] = pv[2]; pv[2] = pv[1]; pv[1] = x;
      } else if (x < pv[2]) {
        pv[3] = pv[2]; pv[2] = x;
      } else {
        pv[3] = x;
      }
      ++nPrev;
    }
  }
  if (forceMany) manyPrev = true;   // (kLatDevManyPrev: every node through the pass-per-previous-node form of lat_node_lines)
#pragma unroll
  for (int q = 0; q < 4; ++q) r.pid[q] = (u32)q < nPrev && !manyPrev ? load_l2(&S.id[nb + pv[q]]) : 0u;
  r.flags = manyPrev ? kLatRecMany : nPrev;
  r.bytes = 0;
  r.pad = 0;
  return true;
}

// One lattice node of the output: its lines (one per row of its entry), printed by ONE lane from its record.
template <bool WRITE, typename P>
__device__ __forceinline__ void lat_node_lines(LatOut<WRITE, P>& w, const Batch& B, const Config& cfg, const LatTable& T, const LatScratch& S,
                                              u64 nb, const LatRec& rec, u32 id, const u8* text, const u16* boff) {
  const u64 gn = nb + rec.node;
  const NodeInfo ni = B.node_info[gn];
  const bool unk = ni.eptr < 0;
  NodeAux na{0, 0, 0, 0, 0, 0};
  if (unk) na = B.node_aux[gn];
  const bool two = T.n_weights == 2 && cfg.nscorers > 1;
  const GDigits g0 = g_digits(rec.f0), g1 = g_digits(rec.f1), gt = g_digits(two ? rec.f0 + rec.f1 : rec.f0);
  const u32 rep = unk && na.maker < 16 ? T.maker_replaces[na.maker] : 0;
  FmtSurface raw{text, 0}, esc{text, 0};
  if (unk) {
    const u32 b0 = boff[ni.start], b1 = boff[ni.end];
    raw = FmtSurface{text + b0, b1 - b0};
    esc = raw;
    if (raw.len == 1)
      for (int e = 0; e < (int)T.n_escapes; ++e)
        if (text[b0] == T.escape_from[e]) esc = FmtSurface{T.escape_to[e], T.escape_len[e]};
  }
  const u32 fv = !unk ? 0u : T.flag_placeholder == 0 ? na.ph0 : T.flag_placeholder == 1 ? na.ph1 : 0u;
  const bool manyPrev = (rec.flags & kLatRecMany) != 0;
  const u32 nPrev = rec.flags & 7u;
  for (u32 row = rec.row;; ++row) {
    const LatRow r = T.rows[row];
    // "-\t" id "\t" prevs "\t" start "\t" end "\t"
    w.ch('-');
    w.ch('\t');
    w.num(id);
    w.ch('\t');
    if (!manyPrev) {
      for (u32 q = 0; q < nPrev; ++q) {
        if (q) w.ch(';');
        w.num(rec.pid[q]);
      }
    } else {
      // more than four distinct previous nodes: the smallest one above the last printed, again and again
      const u64 slots = load_l2(&S.mask[gn]) >> 32;
      const BeamSlot* beams = B.node_beam + gn * (u64)cfg.beam;
      i64 last = -1;
      bool first = true;
      for (;;) {
        u32 m = ~0u;
        for (u64 sm = slots; sm != 0; sm &= sm - 1) {
          const u32 pj = beams[__builtin_ctzll(sm)].prev_node;
          if ((i64)pj > last && pj < m) m = pj;
        }
        if (m == ~0u) break;
        if (!first) w.ch(';');
        w.num(load_l2(&S.id[nb + m]));
        last = (i64)m;
        first = false;
      }
    }
    w.ch('\t');
    w.num(ni.start);
    w.ch('\t');
    w.num((u32)ni.end - 1u);
    w.ch('\t');
    // the entry row: S \t X \t R \t B \t REST
    const u8* pS = T.blob + r.blob_off;
    const u32 lenS = r.len_s, lenC = r.len_c, lenR = r.len_r, lenB = r.len_b, lenRest = r.len_rest;
    const u32 lenX = lenC ? lenC : lenB + 1 + lenR;
    if (!unk) {
      w.run(pS, lenS + 1 + lenX + 1 + lenR + 1 + lenB + 1 + lenRest);
    } else {
      const u8* pX = pS + lenS + 1;
      const u8* pR = pX + lenX + 1;
      const u8* pB = pR + lenR + 1;
      const u8* pRest = pB + lenB + 1;
      if (rep & 1) w.run(esc.p, esc.len);
      else w.run(pS, lenS);
      w.ch('\t');
      // canonic form, or baseform '/' reading when it is empty (lattice_format.cc:172-177); no tab escape here
      const u32 cLen = (rep & 8) ? raw.len : lenC;
      if (cLen != 0) {
        if (rep & 8) w.run(raw.p, raw.len);
        else w.run(pX, lenC);
      } else {
        if (rep & 4) w.run(raw.p, raw.len);
        else w.run(pB, lenB);
        w.ch('/');
        if (rep & 2) w.run(raw.p, raw.len);
        else w.run(pR, lenR);
      }
      w.ch('\t');
      if (rep & 2) w.run(esc.p, esc.len);
      else w.run(pR, lenR);
      w.ch('\t');
      if (rep & 4) w.run(esc.p, esc.len);
      else w.run(pB, lenB);
      w.ch('\t');
      w.run(pRest, lenRest);
      if (fv != 0) {
        w.lit(T.flag_label, T.flag_label_len);
        for (int f = 0; f < (int)T.n_flags; ++f)
          if (fv & T.flag_mask[f]) w.ch(T.flag_char[f]);
        w.ch('|');
      }
    }
    // scores and ranks
    w.lit(T.feat_text, T.feat_len);
    w.flt(g0);
    w.ch('|');
    if (two) {
      w.lit(T.lm_text, T.lm_len);
      w.flt(g1);
      w.ch('|');
    }
    w.lit(T.total_text, T.total_len);
    w.flt(gt);
    w.ch('|');
    w.lit(T.ranks_text, T.ranks_len);
    for (u32 mm = rec.ranks; mm != 0;) {
      const u32 j = (u32)__builtin_ctz(mm);
      mm &= mm - 1;
      w.num(j + 1);
      if (mm != 0) w.ch(';');
    }
    w.ch('\n');
    if (r.flags & 2) break;
  }
}

// "# MA-SCORE\t" "rank" i ":" total " " ... "\n": lane i prints rank i + 1 at the offset a wave scan gives it (one
// lane printing all of them -- 32 "%g" conversions in a row on one lane of 64 -- was a third of both kernels).  All lanes
// call; returns the bytes of the line.
template <bool WRITE, typename P>
__device__ __forceinline__ u32 lat_header(P base, const LatTable& T, const BeamSlot* eos, int beam, int n_best, u32 lane) {
  const int maxN = n_best < beam ? n_best : beam;
  BeamSlot el{kFake16, kFake16, 0.f, 0xffffffffu, 0};
  if ((int)lane < maxN) el = eos[lane];
  const u64 fakes = wave_ballot(el.left == kFake16 && el.beam == kFake16);
  const u32 npaths = fakes ? (u32)__builtin_ctzll(fakes) : 64u;
  const bool mine = lane < npaths;
  const GDigits g = g_digits(mine ? el.total : 0.f);
  LatOut<false> c{nullptr, 0};
  if (mine) {
    c.n = T.rank_len;
    c.num(lane + 1);
    c.ch(':');
    c.flt(g);
    c.ch(' ');
  }
  const u32 bytes = (u32)c.n;
  const u32 incl = wave_scan_incl_u32(bytes, (int)lane);
  const u32 total = (u32)T.head_len + wave_bcast_u32(incl, 63) + 1u;
  if (WRITE) {
    if (lane == 0) {
      LatOut<true, P> h{base, 0};
      h.lit(T.head_text, T.head_len);
      base[total - 1] = (u8)'\n';
    }
    if (mine) {
      LatOut<true, P> w{base + T.head_len + (incl - bytes), 0};
      w.lit(T.rank_text, T.rank_len);
      w.num(lane + 1);
      w.ch(':');
      w.flt(g);
      w.ch(' ');
    }
  }
  return total;
}

// the table's literals, copied to LDS once per workgroup (its pointers keep pointing into HBM)
__device__ __forceinline__ void lat_stage_table(LatTable* dst, const LatTable* __restrict__ src) {
  static_assert(sizeof(LatTable) % 4 == 0, "copied by words");
  for (u32 i = threadIdx.x; i < sizeof(LatTable) / 4; i += blockDim.x) reinterpret_cast<u32*>(dst)[i] = reinterpret_cast<const u32*>(src)[i];
  __syncthreads();
}

__device__ __forceinline__ u32 wave_or_u32(u32 v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v |= wave_shfl_u32(v, lane_id() ^ off);
  return v;
}

// sentence s -> the bytes of its text and of its header; leaves the per-node sets, the ids and the records of the marked
// nodes (with the bytes of their lines) in the scratch for k_lat_write
__global__ void __launch_bounds__(256) k_lat_count(Batch B, Config cfg, const LatTable* __restrict__ Tp, LatScratch S, int n_best,
                                                   u32* sent_bytes, u32* head_bytes, i32* fmt_status, u32 dev) {
  __shared__ LatTable s_T;
  lat_stage_table(&s_T, Tp);
  const LatTable& T = s_T;
  const u32 s = blockIdx.x * 4 + (threadIdx.x >> 6);
  const u32 lane = threadIdx.x & 63;
  if (s >= B.n_sent) return;
  if (B.sent_status[s] != ST_OK) {
    if (lane == 0) {
      sent_bytes[s] = T.error_len;
      head_bytes[s] = 0;
      S.marked[s] = 0;
      fmt_status[s] = B.sent_status[s];
    }
    return;
  }
  const u32 N = B.sent_nodes[s];
  if (N <= 3) {   // createdBoundaryCount() == 3: the empty sentence prints "EOS\n" alone (lattice_format.cc:87-91)
    if (lane == 0) {
      sent_bytes[s] = T.eos_len;
      head_bytes[s] = 0;
      S.marked[s] = 0;
      fmt_status[s] = ST_OK;
    }
    return;
  }
  const u64 nb = B.node_base[s];
  const int beam = cfg.beam, G = cfg.gbeam, NS = cfg.nscorers;
  for (u32 k = lane; k < N; k += 64) {
    S.mask[nb + k] = 0;
    S.best[nb + k] = ~0ull;
    S.id[nb + k] = 0;
  }
  wave_fence_global();
  wave_sync();
  // fillInfo: lane i walks path i back from the EOS beam (paths behind the first fake slot do not exist).  The paths
  // run in lock step, and the N best paths of a sentence mostly run through the SAME nodes: the lanes standing on one
  // node are found with a ballot (which IS that node's share of the rank mask), their slot sets and smallest keys are
  // reduced across the wavefront, and one lane sends the two atomics -- half a dozen instead of a hundred per step.
  const BeamSlot* beams = B.node_beam + nb * (u64)beam;
  int maxN = n_best < beam ? n_best : beam;
  if (maxN > 32) maxN = 32;   // (the device keeps at most 32 slots per node: jppgpu_api.cc device_beam; the packed masks rely on it)
  BeamSlot el{kFake16, kFake16, 0.f, 0xffffffffu, 0};
  if ((int)lane < maxN) el = beams[(u64)(N - 1) * beam + lane];
  const u64 fakes = wave_ballot(el.left == kFake16 && el.beam == kFake16);
  const u32 npaths = fakes ? (u32)__builtin_ctzll(fakes) : 64u;
  {
    const bool two = T.n_weights == 2 && NS > 1;
    const float w0 = T.weights[0], w1 = T.weights[1];
    bool act = lane < npaths;
    u32 node = el.prev_node, slot = el.beam, steps = 0;
    const BeamSlot fake{kFake16, kFake16, 0.f, 0xffffffffu, 0};
    auto live = [&](u32 nd, u32 sl) { return nd >= 2 && nd < N && sl < (u32)beam; };
    // the slot of the first step; from then on the NEXT step's slot is requested before this step's score cell is waited
    // for (both hang off the slot just read: one round trip per step instead of two)
    act = act && live(node, slot);
    BeamSlot c = act ? beams[(u64)node * beam + slot] : fake;
    for (;;) {
      act = act && steps++ <= N && !(c.left == kFake16 && c.beam == kFake16);
      u64 todo = wave_ballot(act);
      if (todo == 0) break;
      const u32 nnode = c.prev_node, nslot = c.beam;
      const bool nact = act && live(nnode, nslot);
      const BeamSlot cn = nact ? beams[(u64)nnode * beam + nslot] : fake;
      u32 ord = ~0u;   // the connection's total as an ordered word; its key is (ord << 32 | path << 8 | slot)
      if (act) {
        const float* cell = B.node_cells + ((nb + node) * (u64)G + c.pad) * (u64)NS;
        // `total += s[i] * weights[i]` is one fused multiply-add per scorer in the reference's build (host/lattice_format.cc)
        float t = __builtin_fmaf(cell[0], w0, 0.f);
        if (two) t = __builtin_fmaf(cell[1], w1, t);
        ord = lat_order_f32(t);
      }
      while (todo) {
        const int leader = __builtin_ctzll(todo);
        const u32 ln = wave_bcast_u32(node, leader);
        const bool mine = act && node == ln;
        const u64 grp = wave_ballot(mine);
        // (32-bit reductions: a slot is below 32, and among equal totals the smallest key is the lowest lane's)
        const u64 used = wave_or_u32(mine ? (1u << slot) : 0u);
        const u32 omin = ~wave_max_u32(mine ? ~ord : 0u);
        const int first = __builtin_ctzll(wave_ballot(mine && ord == omin));
        const u64 kmin = ((u64)omin << 32) | ((u32)first << 8) | wave_bcast_u32(slot, first);
        if ((int)lane == leader) {
          atomicOr((unsigned long long*)&S.mask[nb + ln], (unsigned long long)(grp | (used << 32)));
          atomicMin((unsigned long long*)&S.best[nb + ln], (unsigned long long)kmin);
        }
        todo &= ~grp;
      }
      act = nact;
      node = nnode;
      slot = nslot;
      c = cn;
    }
  }
  wave_fence_global();
  wave_sync();
  // publishResult: ids from 1 in node order, and the marked nodes listed in that order
  u32 next = 1;
  for (u32 base = 0; base < N; base += 64) {
    const u32 nd = base + lane;
    const bool on = nd >= 2 && nd + 1 < N && load_l2(&S.mask[nb + nd]) != 0;   // (0, 1 = BOS and N - 1 = EOS are never on a list)
    const u64 bal = wave_ballot(on);
    if (on) {
      const u32 id = next + (u32)popc64(bal & ((1ull << lane) - 1ull));
      S.id[nb + nd] = id;
      S.rec[nb + id - 1].node = nd;
    }
    next += (u32)popc64(bal);
  }
  const u32 M = next - 1;
  wave_fence_global();
  wave_sync();
  // the bytes of every marked node's lines, a lane per node
  const u32 off = B.byte_off[s];
  const u8* text = B.text + off;
  const u16* boff = B.cp_boff + off + s;
  u64 sum = 0;
  bool ok = true;
  for (u32 base = 0; base < M; base += 64) {
    const u32 i = base + lane;
    if (i < M) {
      LatRec r;
      LatOut<false> w{nullptr, 0};
      if (lat_node_facts(r, B, cfg, T, S, nb, load_l2(&S.rec[nb + i].node), (dev & kLatDevManyPrev) != 0)) {
        lat_node_lines(w, B, cfg, T, S, nb, r, i + 1, text, boff);
        r.bytes = (u32)w.n;
        S.rec[nb + i] = r;
        sum += w.n;
      } else {
        ok = false;
      }
    }
  }
  sum = wave_sum_u64(sum);
  const bool allOk = wave_ballot(!ok) == 0;
  const u32 hn = lat_header<false>((u8*)nullptr, T, beams + (u64)(N - 1) * beam, beam, n_best, lane);
  if (lane == 0) {
    fmt_status[s] = allOk ? ST_OK : ST_CAPACITY;   // (a node the table cannot render: the text answers like a failed sentence)
    sent_bytes[s] = allOk ? (u32)(hn + sum + T.eos_len) : T.error_len;
    head_bytes[s] = allOk ? hn : 0;
    S.marked[s] = M;
  }
}

// The text leaves through an LDS window per wavefront: the lanes print their nodes' lines into it side by side (byte
// stores into LDS cost nothing; byte stores into HBM are one 64-byte transaction EACH -- 1.5 G of them per batch made
// the first lane-per-node form no faster than the serial one), and the window goes out as whole dwords, 256 contiguous
// bytes per instruction.  The window starts at the output offset's own alignment so that the dwords of both sides match.
#if !defined(JPP_LAT_WIN)
#define JPP_LAT_WIN 12288   // (the tests shrink it at run time: kLatDevWinMask)
#endif
constexpr u32 kLatWin = JPP_LAT_WIN;

__device__ __forceinline__ void lat_flush(u8* out, u64 o, const u8 JPP_LDS* buf, u32 b0, u32 bytes, u32 lane) {
  wave_sync();
  u32 a = (4u - b0) & 3u;
  if (a > bytes) a = bytes;
  if (lane < a) out[o + lane] = buf[b0 + lane];
  const u32 mid = (bytes - a) >> 2;
  const u32 JPP_LDS* src = (const u32 JPP_LDS*)(buf + b0 + a);
  u32* dst = reinterpret_cast<u32*>(out + o + a);
  for (u32 i = lane; i < mid; i += 64) dst[i] = src[i];
  const u32 tail = bytes - a - 4u * mid;
  if (lane < tail) out[o + a + 4u * mid + lane] = buf[b0 + a + 4u * mid + lane];
  wave_sync();
}

// (two wavefronts per SIMD: left to itself the compiler took 274 vector registers for the inlined line printer -- ONE
// workgroup per CU, 1.9 ms per 8 192 sentences where one round of wavefronts takes 0.2.  Held to 256 registers it
// spills 96 bytes and two workgroups share a CU; at three per SIMD (168 registers, 432 bytes of scratch) the same launch
// takes 6 % longer: profiles/r06_ar_*, r06_bc_*)
__global__ void __launch_bounds__(256) JPP_WAVES_PER_EU(2) k_lat_write(Batch B, Config cfg, const LatTable* __restrict__ Tp, LatScratch S, int n_best,
                                                   const u64* sent_off, const u32* head_bytes, u8* out,
                                                   const i32* fmt_status, u32 dev) {
  __shared__ LatTable s_T;
  __shared__ __attribute__((aligned(16))) u8 s_win[4][kLatWin + 16];
  lat_stage_table(&s_T, Tp);
  const LatTable& T = s_T;
  const u32 wv = threadIdx.x >> 6;
  const u32 s = blockIdx.x * 4 + wv;
  const u32 lane = threadIdx.x & 63;
  if (s >= B.n_sent) return;
  u64 o = sent_off[s];
  if (fmt_status[s] != ST_OK) {
    fmt_put(out, o, T.error_text, T.error_len, lane);
    return;
  }
  const u32 N = B.sent_nodes[s];
  if (N <= 3) {
    fmt_put(out, o, T.eos_text, T.eos_len, lane);
    return;
  }
  typedef u8 JPP_LDS* LP;
  const LP win = (LP)s_win[wv];
  u32 winB = kLatWin;
  if ((dev & kLatDevWinMask) != 0 && (dev & kLatDevWinMask) < kLatWin) winB = (dev & kLatDevWinMask) < 16u ? 16u : (dev & kLatDevWinMask);
  const u64 nb = B.node_base[s];
  const u32 off = B.byte_off[s];
  const u8* text = B.text + off;
  const u16* boff = B.cp_boff + off + s;
  // the "# MA-SCORE" line (at most 64 ranks of some 22 bytes)
  {
    const u32 hb = head_bytes[s];
    const u32 b0 = (u32)o & 3u;
    if (hb <= winB) {
      (void)lat_header<true>(win + b0, T, B.node_beam + (nb + (N - 1)) * (u64)cfg.beam, cfg.beam, n_best, lane);
      lat_flush(out, o, win, b0, hb, lane);
    } else {
      (void)lat_header<true>(out + o, T, B.node_beam + (nb + (N - 1)) * (u64)cfg.beam, cfg.beam, n_best, lane);
    }
    o += hb;
  }
  // a lane per marked node; the nodes of a round whose lines fit the window together are printed into it and flushed
  const u32 M = S.marked[s];
  for (u32 base = 0; base < M; base += 64) {
    const u32 i = base + lane;
    LatRec rec{};
    if (i < M) rec = S.rec[nb + i];
    const u32 bytes = rec.bytes;
    const u32 incl = wave_scan_incl_u32(bytes, (int)lane);
    const u32 start = incl - bytes;
    const u32 total = wave_bcast_u32(incl, 63);
    u32 f = 0;   // lanes before f are done
    while (f < 64 && base + f < M) {
      const u32 fstart = wave_bcast_u32(start, (int)f);
      const u32 b0 = (u32)(o + fstart) & 3u;
      const u64 fits = wave_ballot(lane >= f && i < M && incl - fstart <= winB - 4u);
      if (((fits >> f) & 1) == 0) {
        // a single node beyond the window: straight to the output, byte by byte (never seen; a kilobyte feature list)
        if (lane == f) {
          LatOut<true> w{out + o + start, 0};
          lat_node_lines(w, B, cfg, T, S, nb, rec, i + 1, text, boff);
        }
        f += 1;
        continue;
      }
      const u32 cnt = (u32)popc64(fits);   // (the fitting lanes are f .. f + cnt - 1: incl is monotonic)
      const u32 wend = wave_bcast_u32(incl, (int)(f + cnt - 1));
      if ((fits >> lane) & 1) {
        LatOut<true, LP> w{win + b0 + (start - fstart), 0};
        lat_node_lines(w, B, cfg, T, S, nb, rec, i + 1, text, boff);
      }
      lat_flush(out, o + fstart, win, b0, wend - fstart, lane);
      f += cnt;
    }
    o += total;
  }
  fmt_put(out, o, T.eos_text, T.eos_len, lane);
}

}  // namespace jpp

#endif  // JPP_K_LATFMT_H
