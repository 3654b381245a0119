// Output text on the device: the top-1 analysis of every sentence of a batch as the bytes the
// reference's output format prints, assembled from a per-model table of rendered entry rows (include/jppgpu.h:
// jppgpu_format_table; built by host/format_table.cc from the dictionary's string storages, once per model).
//
// Reference behaviour reproduced (the table holds every literal; nothing here knows the JUMAN grammar):
//   JumanFormat::format / formatOne          src/jumandic/shared/juman_format.cc:94-168   top-1 path in text order, one
//                                                                                       line per entry row, "@ " rows
//   OutputManager::locate + StringField[]    src/core/analysis/output.cc:65-130           UNK node = template row with the
//                                                                                       replaced fields read from the input
//   escapeForJumanOutput                     juman_format.cc:42-54
//   formatNormalizedFeature                  juman_format.cc:57-92
//   AnalysisPath::fillIn                     src/core/analysis/analysis_result.cc:25-76   (k_path wrote the path already)
//
// k_fmt_count: one wavefront per sentence, a lane per path node -> bytes of the node, bytes of the sentence.
// k_fmt_write: one wavefront per sentence; the nodes in text order, all 64 lanes copying the bytes of one node
// (dictionary node: ONE contiguous run of the blob per row), so stores are coalesced 64-byte runs.
#ifndef JPP_K_FORMAT_H
#define JPP_K_FORMAT_H

#include "jpp_device.h"

namespace jpp {

struct FmtRow {
  u32 blob_off;
  u16 len_pre, len_s, len_r, len_b;
  u16 len_mid;
  u16 flags;   // bit 0: has_features, bit 1: last row of its entry
  u32 len_feat;
  u32 len_total;
};
static_assert(sizeof(FmtRow) == 24, "jppgpu_format_row");

// device copy of jppgpu_format_table (in HBM: pointers into HBM, the small literals by value)
struct FmtTable {
  const u32* slot_first_row;
  u64 n_slots;
  const FmtRow* rows;
  u64 n_rows;
  const u8* blob;
  u8 maker_replaces[16];
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
  u8 eos_len, error_len;
  u8 eos_text[16];
  u8 error_text[32];
};

// the input bytes of a node's span as an output format prints them: (pointer, length) after escapeForJumanOutput
struct FmtSurface {
  const u8* p;
  u32 len;
};
__device__ __forceinline__ FmtSurface fmt_surface(const FmtTable& T, const u8* text, const u16* boff, NodeInfo ni) {
  const u32 b0 = boff[ni.start], b1 = boff[ni.end];
  FmtSurface s{text + b0, b1 - b0};
  if (s.len == 1) {
    for (int e = 0; e < (int)T.n_escapes; ++e) {
      if (text[b0] == T.escape_from[e]) {
        s.p = T.escape_to[e];
        s.len = T.escape_len[e];
      }
    }
  }
  return s;
}

// first row of the entry `eptr` (raw EntryPtr >= 0), or ~0u when the table has none
__device__ __forceinline__ u32 fmt_first_row(const FmtTable& T, i32 eptr) {
  const u32 slot = ((u32)eptr >> 1) >> 3;
  if (eptr < 0 || slot >= T.n_slots) return ~0u;
  const u32 v = T.slot_first_row[slot];
  return v == 0 ? ~0u : v - 1;
}

__device__ __forceinline__ u32 fmt_flag_bytes(const FmtTable& T, u32 value, u32 len_feat) {
  if (value == 0) return 0;
  u32 n = (len_feat != 0 ? 1u : 0u) + T.flag_label_len;
  for (int f = 0; f < (int)T.n_flags; ++f) n += (value & T.flag_mask[f]) != 0 ? 1u : 0u;
  return n;
}

// bytes one path node prints.  ok = false: the table cannot render it (no row for its entry / template).
__device__ __forceinline__ u32 fmt_node_bytes(const FmtTable& T, const u8* text, const u16* boff, NodeInfo ni, NodeAux na, bool* ok) {
  const bool unk = ni.eptr < 0;
  u32 row = fmt_first_row(T, unk ? na.tmpl : ni.eptr);
  if (row == ~0u) {
    *ok = false;
    return 0;
  }
  u32 total = 0;
  if (!unk) {
    for (;; ++row) {
      const FmtRow r = T.rows[row];
      total += r.len_total;
      if (r.flags & 2) break;
    }
    return total;
  }
  const u32 rep = na.maker < 16 ? T.maker_replaces[na.maker] : 0;
  const u32 slen = fmt_surface(T, text, boff, ni).len;
  const u32 fv = T.flag_placeholder == 0 ? na.ph0 : T.flag_placeholder == 1 ? na.ph1 : 0;
  for (;; ++row) {
    const FmtRow r = T.rows[row];
    total += r.len_pre + ((rep & 1) ? slen : r.len_s) + 1 + ((rep & 2) ? slen : r.len_r) + 1 + ((rep & 4) ? slen : r.len_b) +
             r.len_mid + 1 + r.len_feat + fmt_flag_bytes(T, fv, r.len_feat) + 2;
    if (r.flags & 2) break;
  }
  return total;
}

// sentence s -> its number of text bytes; per path node (text order k = 0 .. pl - 2) the bytes at fmt_len[node_base + k]
// fmt_status: the status the TEXT of a sentence answers with -- the analysis' own, or ST_CAPACITY when the table cannot
// render a node of the path (B.sent_status is not touched: earlier fetches of the result have reported it)
__global__ void __launch_bounds__(256) k_fmt_count(Batch B, const FmtTable* __restrict__ Tp, u32* fmt_len, u32* sent_bytes, i32* fmt_status) {
  const FmtTable& T = *Tp;
  const u32 s = blockIdx.x * 4 + (threadIdx.x >> 6);
  const u32 lane = threadIdx.x & 63;
  if (s >= B.n_sent) return;
  const u32 pl = B.sent_status[s] == ST_OK ? B.path_len[s] : 0;
  if (B.sent_status[s] != ST_OK) {
    if (lane == 0) {
      sent_bytes[s] = T.error_len;
      fmt_status[s] = B.sent_status[s];
    }
    return;
  }
  const u64 nb = B.node_base[s];
  const u32 off = B.byte_off[s];
  const u8* text = B.text + off;
  const u16* boff = B.cp_boff + off + s;
  u32 sum = 0;
  bool ok = true;
  // path_nodes is EOS first: text-order node k is path_nodes[pl - 1 - k], k < pl - 1 (EOS is not printed)
  for (u32 k = lane; k + 1 < pl; k += 64) {
    const u32 node = B.path_nodes[nb + (pl - 1 - k)];
    const u32 n = fmt_node_bytes(T, text, boff, B.node_info[nb + node], B.node_aux[nb + node], &ok);
    fmt_len[nb + k] = n;
    sum += n;
  }
  sum = wave_sum_u32(sum);
  const bool allOk = wave_ballot(!ok) == 0;
  if (lane == 0) {
    fmt_status[s] = allOk ? ST_OK : ST_CAPACITY;   // (a node the table cannot render: the text answers like a failed sentence)
    sent_bytes[s] = allOk ? sum + T.eos_len : T.error_len;
  }
}

// all lanes of the wavefront copy `len` bytes (wave-uniform arguments); returns the advanced output position
__device__ __forceinline__ u64 fmt_put(u8* out, u64 o, const u8* src, u32 len, u32 lane) {
  for (u32 i = lane; i < len; i += 64) out[o + i] = src[i];
  return o + len;
}
__device__ __forceinline__ u64 fmt_put1(u8* out, u64 o, u8 c, u32 lane) {
  if (lane == 0) out[o] = c;
  return o + 1;
}

__global__ void __launch_bounds__(256) k_fmt_write(Batch B, const FmtTable* __restrict__ Tp, const u32* fmt_len, const u64* sent_off, u8* out, const i32* fmt_status) {
  const FmtTable& T = *Tp;
  const u32 s = blockIdx.x * 4 + (threadIdx.x >> 6);
  const u32 lane = threadIdx.x & 63;
  if (s >= B.n_sent) return;
  u64 o = sent_off[s];
  if (fmt_status[s] != ST_OK) {
    fmt_put(out, o, T.error_text, T.error_len, lane);
    return;
  }
  const u32 pl = B.path_len[s];
  const u64 nb = B.node_base[s];
  const u32 off = B.byte_off[s];
  const u8* text = B.text + off;
  const u16* boff = B.cp_boff + off + s;
  for (u32 k = 0; k + 1 < pl; ++k) {
    const u32 node = uni(B.path_nodes[nb + (pl - 1 - k)]);
    const NodeInfo ni = B.node_info[nb + node];
    const bool unk = ni.eptr < 0;
    NodeAux na{0, 0, 0, 0, 0, 0};
    if (unk) na = B.node_aux[nb + node];
    u32 row = uni(fmt_first_row(T, unk ? na.tmpl : ni.eptr));
    if (!unk) {
      for (;; ++row) {
        const FmtRow r = T.rows[row];
        o = fmt_put(out, o, T.blob + r.blob_off, uni(r.len_total), lane);
        if (r.flags & 2) break;
      }
      continue;
    }
    const u32 rep = na.maker < 16 ? T.maker_replaces[na.maker] : 0;
    const FmtSurface sf = fmt_surface(T, text, boff, ni);
    const u32 slen = uni(sf.len);
    const u32 fv = T.flag_placeholder == 0 ? na.ph0 : T.flag_placeholder == 1 ? na.ph1 : 0;
    for (;; ++row) {
      const FmtRow r = T.rows[row];
      const u8* p = T.blob + r.blob_off;
      o = fmt_put(out, o, p, r.len_pre, lane);
      p += r.len_pre;
      o = (rep & 1) ? fmt_put(out, o, sf.p, slen, lane) : fmt_put(out, o, p, r.len_s, lane);
      p += r.len_s + 1;
      o = fmt_put1(out, o, ' ', lane);
      o = (rep & 2) ? fmt_put(out, o, sf.p, slen, lane) : fmt_put(out, o, p, r.len_r, lane);
      p += r.len_r + 1;
      o = fmt_put1(out, o, ' ', lane);
      o = (rep & 4) ? fmt_put(out, o, sf.p, slen, lane) : fmt_put(out, o, p, r.len_b, lane);
      p += r.len_b;
      o = fmt_put(out, o, p, r.len_mid, lane);
      p += r.len_mid;
      // an UNK node always prints the quoted form (hasFeatures = special || ..., juman_format.cc:127-129)
      o = fmt_put1(out, o, '"', lane);
      if (r.flags & 1) o = fmt_put(out, o, p + 1, r.len_feat, lane);   // (p[0] is the opening quote of the stored TAIL)
      if (fv != 0) {
        if (r.len_feat != 0) o = fmt_put1(out, o, ' ', lane);
        o = fmt_put(out, o, T.flag_label, T.flag_label_len, lane);
        for (int f = 0; f < (int)T.n_flags; ++f)
          if (fv & T.flag_mask[f]) o = fmt_put1(out, o, T.flag_char[f], lane);
      }
      o = fmt_put1(out, o, '"', lane);
      o = fmt_put1(out, o, '\n', lane);
      if (r.flags & 2) break;
    }
  }
  fmt_put(out, o, T.eos_text, T.eos_len, lane);
}

}  // namespace jpp

#endif  // JPP_K_FORMAT_H
