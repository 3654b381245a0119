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
//                (64-bit mask = the ranks), the set of beam slots used (= the distinct connections) and the connection
//                with the smallest total (atomicMin of an ordered key) in four per-node words of HBM; then numbers the
//                marked nodes (ballot + popcount = publishResult's ids), and counts the bytes of every line;
//   k_lat_write  the same walk over the marked nodes in id order, one line at a time: numbers and scores are printed
//                by lane 0 into an LDS line buffer and flushed by all lanes, entry-row text is copied blob -> output by
//                all 64 lanes (a dictionary node's row is ONE contiguous run), so stores are coalesced.
// What is wave-uniform is kept uniform (node, masks, lengths): a line is a straight sequence of cooperative copies.
#ifndef JPP_K_LATFMT_H
#define JPP_K_LATFMT_H

#include "jpp_device.h"
#include "jpp_fmtg.h"
#include "k_format.h"

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

// per-node scratch of the formatter (HBM, [total_nodes] each)
struct LatScratch {
  u64* mask;    // paths (ranks) through the node
  u64* slots;   // beam slots of the node those paths use
  u64* best;    // min over those connections of (ordered total << 32 | path << 8 | slot)
  u32* id;      // publishResult's id, 0 = not on a path
};

constexpr u32 kLatTmp = 1024;       // LDS line buffer per wavefront (room() flushes before it would overflow)

__device__ __forceinline__ u32 lat_order_f32(float t) {
  u32 b;
  __builtin_memcpy(&b, &t, 4);
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}

__device__ __forceinline__ u32 wave_min_u32(u32 v) { return ~wave_max_u32(~v); }

__device__ __forceinline__ u32 lat_first_row(const LatTable& T, i32 eptr) {
  const u32 slot = ((u32)eptr >> 1) >> 3;
  if (eptr < 0 || slot >= T.n_slots) return ~0u;
  const u32 v = T.slot_first_row[slot];
  return v == 0 ? ~0u : v - 1;
}

// The cursor of one sentence's text: `o` is the next output byte, `n` the fill of the LDS line buffer.  WRITE = false
// only counts.  All arguments of its members are wave-uniform.
template <bool WRITE>
struct LatOut {
  u8* out;
  u8* tmp;
  u64 o;
  u32 n;
  u32 lane;
  __device__ __forceinline__ void flush() {
    if (WRITE) {
      wave_sync();
      for (u32 i = lane; i < n; i += 64) out[o + i] = tmp[i];
      wave_sync();
    }
    o += n;
    n = 0;
  }
  // lane 0's cursor into the line buffer (null for the other lanes and when counting)
  __device__ __forceinline__ u8* cur() const { return (WRITE && lane == 0) ? tmp + n : nullptr; }
  __device__ __forceinline__ void ch(u8 c) {
    if (WRITE && lane == 0) tmp[n] = c;
    ++n;
  }
  __device__ __forceinline__ void num(u32 v) { n += u32_format(v, cur()); }
  __device__ __forceinline__ void flt(float v) { n += g_format(v, cur()); }
  __device__ __forceinline__ void lit(const u8* p, u32 len) {   // a short literal of the table (by value in the kernel argument's copy)
    if (WRITE && lane == 0)
      for (u32 i = 0; i < len; ++i) tmp[n + i] = p[i];
    n += len;
  }
  // a run of bytes from HBM (the blob, the input text) straight to the output, after what the line buffer holds
  __device__ __forceinline__ void run(const u8* src, u32 len) {
    if (n) flush();
    if (WRITE)
      for (u32 i = lane; i < len; i += 64) out[o + i] = src[i];
    o += len;
  }
  __device__ __forceinline__ void room(u32 need) {
    if (n + need > kLatTmp) flush();
  }
};

// one lattice node of the output: its lines.  Returns false when the table has no row for it.
template <bool WRITE>
__device__ __forceinline__ bool lat_node_lines(LatOut<WRITE>& w, const Batch& B, const Config& cfg, const LatTable& T, const LatScratch& S,
                                              u64 nb, u32 node, const u8* text, const u16* boff) {
  const u32 lane = w.lane;
  const int beam = cfg.beam, G = cfg.gbeam, NS = cfg.nscorers;
  const u64 gn = nb + node;
  const NodeInfo ni = B.node_info[gn];
  const bool unk = ni.eptr < 0;
  NodeAux na{0, 0, 0, 0, 0, 0};
  if (unk) na = B.node_aux[gn];
  u32 row = uni(lat_first_row(T, unk ? na.tmpl : ni.eptr));
  if (row == ~0u) return false;
  const u64 mask = S.mask[gn], slots = S.slots[gn];
  const u32 bslot = (u32)(S.best[gn] & 0xffu);
  const BeamSlot* beams = B.node_beam + gn * (u64)beam;
  // distinct previous nodes, ascending: lane j holds the previous node of slot j
  const u32 pj = (lane < (u32)beam && ((slots >> lane) & 1)) ? beams[lane].prev_node : ~0u;
  // the scores of the chosen connection
  const float* cell = B.node_cells + (gn * (u64)G + beams[bslot].pad) * (u64)NS;
  const float f0 = cell[0] * T.weights[0];
  const bool two = T.n_weights == 2 && NS > 1;
  const float f1 = two ? cell[1] * T.weights[1] : 0.f;
  const float ftot = two ? f0 + f1 : f0;
  const u32 id = S.id[gn];
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
  for (;; ++row) {
    const LatRow r = T.rows[row];
    // "-\t" id "\t" prevs "\t" start "\t" end "\t"
    w.room(64);
    w.ch('-');
    w.ch('\t');
    w.num(id);
    w.ch('\t');
    {
      i64 last = -1;
      bool first = true;
      for (;;) {
        const u32 cand = (pj != ~0u && (i64)pj > last) ? pj : ~0u;
        const u32 m = uni(wave_min_u32(cand));
        if (m == ~0u) break;
        w.room(16);
        if (!first) w.ch(';');
        w.num(S.id[nb + m]);
        last = (i64)m;
        first = false;
      }
    }
    w.room(32);
    w.ch('\t');
    w.num(ni.start);
    w.ch('\t');
    w.num((u32)ni.end - 1u);
    w.ch('\t');
    // the entry row: S \t X \t R \t B \t REST
    const u8* pS = T.blob + r.blob_off;
    const u32 lenS = uni((u32)r.len_s), lenC = uni((u32)r.len_c), lenR = uni((u32)r.len_r), lenB = uni((u32)r.len_b), lenRest = uni(r.len_rest);
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
        w.room(64);
        w.lit(T.flag_label, T.flag_label_len);
        for (int f = 0; f < (int)T.n_flags; ++f)
          if (fv & T.flag_mask[f]) w.ch(T.flag_char[f]);
        w.ch('|');
      }
    }
    // scores and ranks
    w.room(160);
    w.lit(T.feat_text, T.feat_len);
    w.flt(f0);
    w.ch('|');
    if (two) {
      w.lit(T.lm_text, T.lm_len);
      w.flt(f1);
      w.ch('|');
    }
    w.lit(T.total_text, T.total_len);
    w.flt(ftot);
    w.ch('|');
    w.lit(T.ranks_text, T.ranks_len);
    for (u64 mm = mask; mm != 0;) {
      const u32 j = (u32)__builtin_ctzll(mm);
      mm &= mm - 1;
      w.room(8);
      w.num(j + 1);
      if (mm != 0) w.ch(';');
    }
    w.ch('\n');
    w.flush();
    if (r.flags & 2) break;
  }
  return true;
}

// "# MA-SCORE\t" "rank" i ":" total " " ... "\n"
template <bool WRITE>
__device__ __forceinline__ void lat_header(LatOut<WRITE>& w, const LatTable& T, const BeamSlot* eos, int beam, int n_best) {
  w.lit(T.head_text, T.head_len);
  for (int i = 0; i < n_best && i < beam; ++i) {
    const BeamSlot el = eos[i];
    if (el.left == kFake16 && el.beam == kFake16) break;
    w.room(48);
    w.lit(T.rank_text, T.rank_len);
    w.num((u32)i + 1);
    w.ch(':');
    w.flt(el.total);
    w.ch(' ');
  }
  w.room(4);
  w.ch('\n');
  w.flush();
}

// the body of both kernels from the numbered nodes on: header, lines, EOS.  Returns the bytes; *head = bytes of the header.
template <bool WRITE>
__device__ __forceinline__ u64 lat_sentence_text(const Batch& B, const Config& cfg, const LatTable& T, const LatScratch& S, u32 s, int n_best,
                                                 u8* out, u64 o0, u8* tmp, u32 lane, u32* head, bool* ok) {
  const u64 nb = B.node_base[s];
  const u32 N = B.sent_nodes[s];
  const u32 off = B.byte_off[s];
  const u8* text = B.text + off;
  const u16* boff = B.cp_boff + off + s;
  LatOut<WRITE> w{out, tmp, o0, 0, lane};
  lat_header(w, T, B.node_beam + (nb + (N - 1)) * (u64)cfg.beam, cfg.beam, n_best);
  *head = (u32)(w.o - o0);
  *ok = true;
  for (u32 base = 2; base + 1 < N; base += 64) {   // (0, 1 = BOS, N - 1 = EOS are never on the list)
    const u32 nd = base + lane;
    u64 bal = wave_ballot(nd + 1 < N && S.id[nb + nd] != 0);
    while (bal) {
      const u32 j = (u32)__builtin_ctzll(bal);
      bal &= bal - 1;
      if (!lat_node_lines<WRITE>(w, B, cfg, T, S, nb, uni(base + j), text, boff)) *ok = false;
    }
  }
  w.lit(T.eos_text, T.eos_len);
  w.flush();
  return w.o - o0;
}

// sentence s -> its number of text bytes and of header bytes; leaves the per-node sets in the scratch for k_lat_write
__global__ void __launch_bounds__(256) k_lat_count(Batch B, Config cfg, const LatTable* __restrict__ Tp, LatScratch S, int n_best,
                                                   u32* sent_bytes, u32* head_bytes, i32* fmt_status) {
  const LatTable& T = *Tp;
  const u32 s = blockIdx.x * 4 + (threadIdx.x >> 6);
  const u32 lane = threadIdx.x & 63;
  if (s >= B.n_sent) return;
  if (B.sent_status[s] != ST_OK) {
    if (lane == 0) {
      sent_bytes[s] = T.error_len;
      head_bytes[s] = 0;
      fmt_status[s] = B.sent_status[s];
    }
    return;
  }
  const u32 N = B.sent_nodes[s];
  if (N <= 3) {   // createdBoundaryCount() == 3: the empty sentence prints "EOS\n" alone (lattice_format.cc:87-91)
    if (lane == 0) {
      sent_bytes[s] = T.eos_len;
      head_bytes[s] = 0;
      fmt_status[s] = ST_OK;
    }
    return;
  }
  const u64 nb = B.node_base[s];
  const int beam = cfg.beam, G = cfg.gbeam, NS = cfg.nscorers;
  for (u32 k = lane; k < N; k += 64) {
    S.mask[nb + k] = 0;
    S.slots[nb + k] = 0;
    S.best[nb + k] = ~0ull;
    S.id[nb + k] = 0;
  }
  __threadfence();
  wave_sync();
  // fillInfo: lane i walks path i back from the EOS beam (paths behind the first fake slot do not exist)
  const BeamSlot* beams = B.node_beam + nb * (u64)beam;
  const int maxN = n_best < beam ? n_best : beam;
  BeamSlot el{kFake16, kFake16, 0.f, 0xffffffffu, 0};
  if ((int)lane < maxN) el = beams[(u64)(N - 1) * beam + lane];
  const u64 fakes = wave_ballot(el.left == kFake16 && el.beam == kFake16);
  const u32 npaths = fakes ? (u32)__builtin_ctzll(fakes) : 64u;
  if (lane < npaths) {
    const bool two = T.n_weights == 2 && NS > 1;
    u32 node = el.prev_node, slot = el.beam, steps = 0;
    while (node >= 2 && node < N && slot < (u32)beam && steps++ <= N) {
      const BeamSlot c = beams[(u64)node * beam + slot];
      if (c.left == kFake16 && c.beam == kFake16) break;
      const float* cell = B.node_cells + ((nb + node) * (u64)G + c.pad) * (u64)NS;
      // `total += s[i] * weights[i]` is one fused multiply-add per scorer in the reference's build (host/lattice_format.cc)
      float t = __builtin_fmaf(cell[0], T.weights[0], 0.f);
      if (two) t = __builtin_fmaf(cell[1], T.weights[1], t);
      atomicOr((unsigned long long*)&S.mask[nb + node], 1ull << lane);
      atomicOr((unsigned long long*)&S.slots[nb + node], 1ull << slot);
      atomicMin((unsigned long long*)&S.best[nb + node], ((unsigned long long)lat_order_f32(t) << 32) | (lane << 8) | slot);
      node = c.prev_node;
      slot = c.beam;
    }
  }
  __threadfence();
  wave_sync();
  // publishResult: ids from 1 in node order
  u32 next = 1;
  for (u32 base = 0; base < N; base += 64) {
    const u32 nd = base + lane;
    const bool on = nd < N && S.mask[nb + nd] != 0;
    const u64 bal = wave_ballot(on);
    if (on) S.id[nb + nd] = next + (u32)popc64(bal & ((1ull << lane) - 1ull));
    next += (u32)popc64(bal);
  }
  __threadfence();
  wave_sync();
  u32 head = 0;
  bool ok = true;
  const u64 bytes = lat_sentence_text<false>(B, cfg, T, S, s, n_best, nullptr, 0, nullptr, lane, &head, &ok);
  if (lane == 0) {
    fmt_status[s] = ok ? ST_OK : ST_CAPACITY;   // (a node the table cannot render: the text answers like a failed sentence)
    sent_bytes[s] = ok ? (u32)bytes : T.error_len;
    head_bytes[s] = ok ? head : 0;
  }
}

__global__ void __launch_bounds__(256) k_lat_write(Batch B, Config cfg, const LatTable* __restrict__ Tp, LatScratch S, int n_best,
                                                   const u64* sent_off, u8* out, const i32* fmt_status) {
  __shared__ u8 s_tmp[4][kLatTmp];
  const LatTable& T = *Tp;
  const u32 wv = threadIdx.x >> 6;
  const u32 s = blockIdx.x * 4 + wv;
  const u32 lane = threadIdx.x & 63;
  if (s >= B.n_sent) return;
  const u64 o = sent_off[s];
  if (fmt_status[s] != ST_OK) {
    fmt_put(out, o, T.error_text, T.error_len, lane);
    return;
  }
  if (B.sent_nodes[s] <= 3) {
    fmt_put(out, o, T.eos_text, T.eos_len, lane);
    return;
  }
  u32 head = 0;
  bool ok = true;
  (void)lat_sentence_text<true>(B, cfg, T, S, s, n_best, out, o, s_tmp[wv], lane, &head, &ok);
}

}  // namespace jpp

#endif  // JPP_K_LATFMT_H
