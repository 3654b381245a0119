// Kernels 6: RNNLM re-ranking of the paths that survive at EOS (Mikolov faster-rnnlm NCE model), then
// adjustBeamScores + remakeEosBeam.  Launch order for E <= 128 (jppgpu_api.cc):
//
//   k_rnn_paths    one thread per (sentence, EOS path): walks the beam pointers back from EOS -- the lattice
//                  connection of every surviving path at every boundary, its global-beam index, its node's length.
//   k_rnn_prep     one wavefront per sentence: resolves the RNN vocabulary id of every node on the paths and builds the
//                  RNN lattice of RnnIdContainer::addPath / addPrevChain exactly (including its attach-to-it->second
//                  quirk): which rnn node scores which connection, and each rnn node's predecessor.  Boundary by
//                  boundary, lane = path, all paths of a boundary at once (the rule that makes that exact is at the
//                  kernel); everything in registers, the rows in and out coalesced.
//   k_rnn_dense    the rnn nodes as one record per hidden-state row.
//   k_rnn_order_*  counting sort of the sentences by the length of their recurrence.
//   k_rnn_chain    the recurrence (hidden states of all rnn nodes before EOS) on the matrix cores, 32 sentences per
//                  workgroup in lock step.
//   k_rnn_score<J, SORT, 2>   maxent sums, NCE scores, score cells, adjustBeamScores, remakeEosBeam.
//   k_rnn_score_long<J, SORT> the same for sentences beyond the LDS staging limits of k_rnn_score (per-row scores).
// E > 128 (up to 256): k_rnn_paths, k_rnn_prep, k_rnn_score<4, SORT, 0> (everything in one launch, W from L2).
// The hidden vector of a node is spread over the lanes of its wavefront (EP = E rounded up to 64 / 128 / 256).
//
// Reference behaviour reproduced:
//   RnnIdResolver::resolveIdsAtGbeam / RnnIdContainer::resolveId / reprOf
//                                   src/core/analysis/rnn_id_resolver.cc:157-196,291-323
//   GbeamRnnState::computeContext / scoreBoundary / copyScoresToLattice
//                                   src/core/analysis/rnn_scorer_gbeam.cc:142-267
//     (incl. the quirk that every maxent context slot holds prev->id, :171-188,
//      and embedding id -1 -> 0, :130-133,214-217)
//   MikolovRnnImplParallel          src/rnn/mikolov_rnn_impl.h:196-256
//   MikolovIndexCalculator / ScoreCalculator  mikolov_rnn_impl.h:21-131, PRIMES mikolov_rnn.h:18-25
//   ScoreProcessor::adjustBeamScores / remakeEosBeam  score_processor.cc:521-576
// Float rules: bit-exact to the oracle build of the reference (FMA target, Eigen stand-in of oracle/shim): k-ascending
// fused chains for the matrix-vector product (vector ALU or v_mfma_f32_16x16x4_f32), sequential rounded-product NCE dot,
// glibc's expf restated in f64 (expf_libm), fused multiply-adds where GCC contracts them; see DESIGN.md section 3.
#ifndef JPP_K_RNN_H
#define JPP_K_RNN_H

#include "jpp_device.h"
#include "jpp_select.h"
#include "k_lattice.h"
#include "k_sweep.h"


namespace jpp {

// sum over the 64 lanes, the same value on every lane: DPP row shifts inside the rows of 16, then the four
// row sums through v_readlane -- no LDS crossbar round trips (a ds_bpermute butterfly costs six)
__device__ __forceinline__ float wave_sum_f32(float v) {
  v += row_shl_f32<8>(v);
  v += row_shl_f32<4>(v);
  v += row_shl_f32<2>(v);
  v += row_shl_f32<1>(v);
  return (wave_bcast_f32(v, 0) + wave_bcast_f32(v, 16)) + (wave_bcast_f32(v, 32) + wave_bcast_f32(v, 48));
}

// one byte at a time through a double array; returns false once a label mismatches
template <typename UP>
__device__ __forceinline__ bool rnn_trie_byte(UP units, u32& id, u32& unit, u32 b) {
  id ^= da_offset(unit) ^ b;
  unit = units[id];
  return da_label(unit) == b;
}

// RnnIdContainer::resolveId: word id of lattice node `k` (not EOS)
__device__ inline i32 rnn_resolve_id(const DevModel& M, const Batch& B, u32 s, u64 nb, u32 k) {
  NodeInfo ni = B.node_info[nb + k];
  const i32* entry = B.node_entry + (nb + k) * B.row_stride;
  const u32 JPP_GLOBAL* units = as_global(ni.eptr >= 0 ? M.rnn_known : M.rnn_unk);
  u32 id = 0;
  u32 unit = units[0];
  bool ok = true;
  u32 off = B.byte_off[s];
  const u16* boff = B.cp_boff + off + s;
  for (u32 f = 0; f < M.rnn_nfields && ok; ++f) {
    i32 v = entry[M.rnn_fields[f]];
    if (v >= 0) {
      u32 x = (u32)v;  // RnnReprBuilder::addInt: varint of the u32 value
      for (;;) {
        u32 b = x & 0x7f;
        x >>= 7;
        if (x) b |= 0x80;
        ok = rnn_trie_byte(units, id, unit, b);
        if (!ok || !x) break;
      }
    } else {
      // addString(surface): raw bytes, then varint(1)
      const u8* p = B.text + off + boff[ni.start];
      u32 len = (u32)boff[ni.end] - boff[ni.start];
      for (u32 q = 0; q < len && ok; ++q) ok = rnn_trie_byte(units, id, unit, p[q]);
      if (ok) ok = rnn_trie_byte(units, id, unit, 1);
    }
  }
  if (!ok) return M.rnn_unk_id;
  if (((unit >> 8) & 1) == 0) return M.rnn_unk_id;
  u32 leaf = units[id ^ da_offset(unit)];
  return (i32)(leaf & ((1u << 31) - 1));
}

constexpr int kRnnPrepWaves = 4;    // sentences (wavefronts) per k_rnn_prep workgroup
constexpr int kRnnCN = 4;           // rnn nodes of one boundary evaluated per pass
constexpr u32 kNoConn = 0xffffffffu;

// `localScore += scores.at(i) * scoreWeights.at(i)` of adjustBeamScores / remakeEosBeam
// (score_processor.cc:536-538,567-569), i = perceptron, RNN.  GCC contracts each step into one fused
// multiply-add on an FMA target (the reference's recommended -march=native build; oracle/_ref:
// `vfmadd231ss` in both functions), i.e. the products are NOT rounded before they are added.  The device
// does the same so that totals built from bit-equal cells are bit-equal to the reference's, and two EOS
// paths that tie exactly there tie exactly here.
__device__ __forceinline__ float weighted_score2(float perceptron, float rnn, const Config& cfg) {
  float local = __builtin_fmaf(perceptron, cfg.w_perceptron, 0.f);
  local = __builtin_fmaf(rnn, cfg.w_rnn, local);
  return local;
}

// FastHash1::mix (src/util/fast_hash.h:39-64)
__device__ __forceinline__ u64 fh1_mix(u64 state, u64 data) {
  u64 v = (state ^ data) * kHashMult;
  return v ^ (v >> 32);
}

// The lattice connection of every surviving EOS path at every boundary (RnnIdContainer::addPath walks them the same
// way, rnn_id_resolver.cc): one thread per (sentence, path) follows the beam pointers back from EOS -- a chain of
// dependent HBM reads, so it runs as its own launch with every path of the batch in flight at once instead of on
// six lanes of the wavefront that builds the rnn lattice.  conn[b][p] = node | beam slot << 26, or kNoConn; the
// connection's global-beam index (BeamSlot::pad) and the node's length come from the same records.
__global__ void __launch_bounds__(256) k_rnn_paths(Batch B, Config cfg) {
  const u32 G = (u32)cfg.gbeam;
  const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
  const u32 s = t / G, p = t - s * G;
  if (s >= B.n_sent || B.sent_status[s] != ST_OK) return;
  const u32 n = B.sent_ncp[s];
  if (n == 0) return;
  const u32 bb0 = B.byte_off[s] + 4 * s;
  const u32 bE = n + 2;
  const u32 ngb = B.bnd_ngb[bb0 + bE];
  if (ngb == 0) return;
  const u32 N = B.sent_nodes[s];
  const u64 nb = B.node_base[s];
  const u32 beam = (u32)cfg.beam;
  u32* conn = B.rnn_conn + (u64)bb0 * G;
  u32* gi = B.rnn_gi + (u64)bb0 * G;   // global-beam index | node length << 16
  for (u32 b = 0; b <= bE; ++b) conn[(u64)b * G + p] = kNoConn;   // (gi / clen are read only where a connection exists)
  if (p >= ngb) return;
  const BeamSlot* beams = B.node_beam + nb * beam;
  const GbeamEntry ge = B.bnd_gbeam[(u64)(bb0 + bE) * G + p];
  conn[(u64)bE * G + p] = (N - 1) | (p << 26);  // fake EOS connection, "slot" = path index
  gi[(u64)bE * G + p] = p;   // (length 0)
  u32 nd = B.end_nodes[nb + B.end_first[bb0 + bE] + ge.left];
  u32 k = ge.beam;
  u32 guard = 0;
  while (nd >= 2 && guard++ <= n) {
    const NodeInfo ni = B.node_info[nb + nd];
    const u32 b = (u32)ni.start + 2;
    const BeamSlot sl = beams[(u64)nd * beam + k];
    conn[(u64)b * G + p] = nd | (k << 26);
    gi[(u64)b * G + p] = (sl.pad & 0xffffu) | ((u32)(ni.end - ni.start) << 16);
    nd = sl.prev_node;
    k = sl.beam;
  }
}

// The RNN lattice of a sentence (RnnIdContainer::addPath / addPrevChain over its surviving EOS paths, rnn_id_resolver.cc),
// one wavefront per sentence, lane = path, everything in registers.
//
// The reference adds the paths one after the other, each from BOS to EOS.  What path p does at boundary b depends on the
// earlier paths at b (the rnn nodes they published there, the ptrCache entry of a shared connection) and on p's own
// previous node, so the boundaries can be taken in order with all paths of a boundary side by side -- as long as the
// DISTINCT connections of the boundary are handled in path order.  Per boundary: one coalesced row of connections /
// lengths / ids comes in (requested a boundary ahead), the lanes are grouped by connection with ballots, and for every
// distinct connection (there are one or two, not G) the boundary's published nodes -- node x in lane x's registers --
// are searched with two ballots.  The rnn nodes and the paths' assignments leave as coalesced rows.
//
// Until round 5 lane p replayed path p on a diagonal (boundary t - p in step t) with the bookkeeping in LDS (short
// sentences, 40 KB per workgroup) or in the HBM arrays (long ones): every lane in a cache line of its own, a dependent
// round trip per published node -- 6.8 ms per batch of the configs[4] shape, 4.8 ms with the chain moved into an LDS
// ring (profiles/r05i-r05l: ~9 000 cycles of a wavefront's life per step, the texture path's line rate and HBM latency,
// not arithmetic).  The shortcut tried before (prefetching on the diagonal) made it slower: 10.3-10.7 ms, r05f / r05g.
__global__ void __launch_bounds__(64 * kRnnPrepWaves) k_rnn_prep(Batch B, const DevModel* __restrict__ Mp, Config cfg) {
  const DevModel& M = *Mp;
  const int wv = (int)(threadIdx.x >> 6);
  const int lane = (int)(threadIdx.x & 63);
  const u32 s = blockIdx.x * kRnnPrepWaves + wv;
  if (s >= B.n_sent) return;
  // hidden-state rows of the sentence (rnn_rows, scanned into rnn_rowbase): every sentence owns at least its
  // parking row, sentences with an RNN lattice get parking + BOS + one row per rnn node (end of this kernel)
  bool live = B.sent_status[s] == ST_OK;
  const u32 off = B.byte_off[s];
  const u32 bb0 = off + 4 * s;
  const u32 n = live ? B.sent_ncp[s] : 0u;
  const u32 bE = n + 2;
  const u32 G = (u32)cfg.gbeam;
  const u32 ngb = n != 0 ? B.bnd_ngb[bb0 + bE] : 0u;
  live = live && n != 0 && ngb != 0;
  if (!live) {
    if (lane == 0) B.rnn_rows[s] = 1;
    return;
  }
  const u32 N = B.sent_nodes[s];
  const u64 nb = B.node_base[s];
  const u32* g_conn = B.rnn_conn + (u64)bb0 * G;   // lattice connection of path p at boundary b (k_rnn_paths)
  const u32* g_gi = B.rnn_gi + (u64)bb0 * G;       // (k_rnn_paths: global-beam index | node length << 16)
  u32* g_assign = B.rnn_assign + (u64)bb0 * G;     // rnn node (index within boundary) scoring the connection
  u32* g_prev = B.rnn_prev + (u64)bb0 * G;         // rnn node -> handle (pb * G + pidx) of its predecessor
  i32* g_id = B.rnn_nid + (u64)bb0 * G;
  u32* g_len = B.rnn_nlen + (u64)bb0 * G;
  u32* g_cnt = B.rnn_cnt + bb0;                    // rnn nodes per boundary
  i32* wid = B.rnn_id + (u64)bb0 * G;              // vocabulary id of the connection's lattice node, at the FIRST path through it
  constexpr u32 kNodeMask = 0x03ffffffu;
  const u64 laneBit = u64{1} << lane;
  JPP_PPROF_DECL;

  // ---- A. vocabulary ids ----
  // The surviving paths mostly run through the same lattice nodes: the id of a node is resolved once, by the first path
  // through it.  The first occurrences are gathered into a dense list (in `assign`, not written before B): resolving an
  // id is a chain of a dozen dependent loads through the double array, paid once per 64 list entries instead of once
  // per 64 (boundary, path) slots, most of which are empty or repeats.  64 / G boundaries per round, lane = (boundary,
  // path); equal nodes lie in the same boundary, so the lanes are simply grouped by value, one ballot per distinct node.
  const u32 rowsPer = 64u / G;   // (G <= kMaxGbeam = 32)
  const u32 r = (u32)lane / G, p = (u32)lane - r * G;
  u32 nfirst = 0;
  constexpr u32 kTrip = 4;   // rounds requested together (a round alone waits a whole HBM latency for one line)
  for (u32 b0 = 0; b0 <= bE; b0 += kTrip * rowsPer) {
    u32 cs[kTrip];
#pragma unroll
    for (u32 k = 0; k < kTrip; ++k) {
      const u32 b = b0 + k * rowsPer + r;
      cs[k] = (r < rowsPer && b <= bE) ? g_conn[b * G + p] : kNoConn;
    }
#pragma unroll
    for (u32 k = 0; k < kTrip; ++k) {
      const u32 c = cs[k];
      const u32 nd = c & kNodeMask;
      u64 todo = wave_ballot(c != kNoConn), firsts = 0;
      while (todo != 0) {
        const int L = __builtin_ctzll(todo);
        const u32 ndL = wave_bcast_u32(nd, L);
        firsts |= u64{1} << L;
        todo &= ~wave_ballot(c != kNoConn && nd == ndL);
      }
      if ((firsts & laneBit) != 0) g_assign[nfirst + (u32)popc64(firsts & (laneBit - 1))] = (b0 + k * rowsPer + r) * G + p;
      nfirst += (u32)popc64(firsts);
    }
  }
  wave_sync();
  JPP_PPROF(0);
  for (u32 i = (u32)lane; i < nfirst; i += 64) {
    const u32 q = g_assign[i];
    const u32 nd = g_conn[q] & kNodeMask;
    wid[q] = (nd == N - 1) ? 0 : rnn_resolve_id(M, B, s, nb, nd);
  }
  wave_sync();
  JPP_PPROF(1);

  // ---- B. the rnn lattice, boundary by boundary ----
  // BOS node (boundary 1): RnnIdContainer::addBos
  if (lane == 0) {
    g_cnt[0] = 0;
    g_cnt[1] = 1;
    g_id[G] = 0;
    g_len[G] = 0;
    g_prev[G] = kNoConn;
  }
  if ((u32)lane < 2 * G) g_assign[lane] = 0xffu << 24;   // (boundaries 0 and 1: no connections)
  u32 curB = 1, curX = 0;             // lane = path: its current rnn node (the BOS node) ...
  u64 curHash = 0xdeadbeef0000ULL;    // ... and that node's prefix hash
  const bool isPath = (u32)lane < G;
  // The rows of connections / lengths / ids come in 64 / G boundaries per load, lane = (boundary, path) as in A and two
  // rounds ahead; a boundary's row is brought to the path lanes by shuffle, its results go back the same way, and the
  // rows of assignments / rnn nodes leave coalesced, once per round.
  auto load_round = [&](u32 b0, u32& cc, u32& gg, i32& ww) {
    const u32 b = b0 + r;
    cc = kNoConn;
    if (r < rowsPer && b <= bE) {
      const u32 q = b * G + p;
      cc = g_conn[q];
      gg = g_gi[q];
      ww = wid[q];
    }
  };
  u32 c1 = kNoConn, g1 = 0, c2 = kNoConn, g2 = 0;
  i32 w1 = 0, w2 = 0;
  load_round(2, c1, g1, w1);
  load_round(2 + rowsPer, c2, g2, w2);
  const u64 rowBits = G >= 64 ? ~u64{0} : (u64{1} << G) - 1;
  for (u32 b0 = 2; b0 <= bE; b0 += rowsPer) {
    const u32 cR = c1, gR = g1;
    const i32 wR = w1;
    c1 = c2;
    g1 = g2;
    w1 = w2;
    load_round(b0 + 2 * rowsPer, c2, g2, w2);
    const u64 anyConn = wave_ballot(cR != kNoConn);
    JPP_PPROF(2);
    u32 outA = 0xffu << 24, outLen = 0, outPrev = 0, outCnt = 0;
    i32 outId = 0;
    for (u32 k = 0; anyConn != 0 && k < rowsPer && b0 + k <= bE; ++k) {
      const u64 act = (anyConn >> (k * G)) & rowBits;
      if (act == 0) continue;
      const u32 b = b0 + k;
      const int src = isPath ? (int)(k * G) + lane : lane;
      const u32 c = wave_shfl_u32(cR, src), len = wave_shfl_u32(gR, src) >> 16;
      const i32 w = (i32)wave_shfl_u32((u32)wR, src);
      const bool has = isPath && c != kNoConn;
      // the boundary's rnn nodes: node x in lane x
      u32 cnt = 0, myres = 0;
      i32 nid = 0;
      u32 nlen = 0, nprev = 0;
      u64 nhash = 0;
      if (act != 0) {
        // every path takes the id of the first path through its node
        const u32 nd = c & kNodeMask;
        i32 myid = w;
        u64 todo = act;
        while (todo != 0) {
          const int L = __builtin_ctzll(todo);
          const u32 ndL = wave_bcast_u32(nd, L);
          const i32 idv = (i32)wave_bcast_u32((u32)w, L);
          const bool inNode = has && nd == ndL;
          if (inNode) myid = idv;
          todo &= ~wave_ballot(inNode);
        }
        // What the reference does for a path at this boundary, paths in order (addPrevChain): a path whose connection an
        // earlier path went through takes that path's rnn node (ptrCache_).  Otherwise h = hash(previous rnn node, id,
        // length); `it` = the NEWEST published node of the boundary with the same (id, length), crdCache_; if there is one
        // and a node not newer than it carries the hash h, the connection is attached to `it` (not to the node with the
        // hash), else a new node (id, length, h) is published.
        // All paths at once: every lane hashes its own step.  Lanes with equal h form a group (equal connections have
        // equal histories, so they sit in one group), and as long as equal hashes mean equal (id, length):
        //   - the first lane of a group finds no node with its hash: it publishes the group's node;
        //   - every other first-through-its-connection lane of the group finds it, and is attached to the newest node
        //     with its (id, length) published so far -- the group's own node unless a LATER group with the same (id,
        //     length), another history reaching the same word, has published in between;
        //   - the others copy.
        // So: rnn node of lane p = the last node with p's (id, length) published not later than the first lane through
        // p's connection.  One round of ballots per group (one or two per boundary; the paths differ in part-of-speech
        // far more often than in words); the first lanes of the connections are worked out only where a second group
        // with the same (id, length) exists.  Equal hashes with different (id, length) -- a 64-bit collision inside one
        // boundary of one sentence -- take the path-by-path replay below.
        const u64 h = fh1_mix(curHash, (u64)(u32)myid | ((u64)len << 32));
        const u32 prevMine = curB * G + curX;
        u32 ncreate = 0;       // (node lanes) the lane that published the node
        bool needFirst = false;
        u64 collide = 0;
        todo = act;
        while (todo != 0) {
          const int L = __builtin_ctzll(todo);
          const u64 hv = (u64)wave_bcast_u32((u32)h, L) | ((u64)wave_bcast_u32((u32)(h >> 32), L) << 32);
          const i32 idv = (i32)wave_bcast_u32((u32)myid, L);
          const u32 lenv = wave_bcast_u32(len, L);
          const u32 prevv = wave_bcast_u32(prevMine, L);
          const bool inGroup = has && h == hv;
          const bool sameWord = myid == idv && len == lenv;
          collide |= wave_ballot(inGroup && !sameWord);
          if (wave_ballot((u32)lane < cnt && nid == idv && nlen == lenv) != 0)   // a second history reaching this word
            needFirst = needFirst || (has && sameWord && lane > L);
          if ((u32)lane == cnt) {
            nid = idv;
            nlen = lenv;
            nhash = hv;
            nprev = prevv;
            ncreate = (u32)L;
          }
          if (inGroup) myres = cnt;
          cnt += 1;
          todo &= ~wave_ballot(inGroup);
        }
        u64 newHash = h;
#if defined(JPP_RNN_PREP_SERIAL)
        collide = ~u64{0};   // (test builds: every boundary through the path-by-path replay)
#endif
        const u64 fix = wave_ballot(needFirst);
        if (collide == 0 && fix != 0) {
          // first lane through the connection, for the lanes that need it
          u32 firstLane = (u32)lane;
          todo = fix;
          while (todo != 0) {
            const int L = __builtin_ctzll(todo);
            const u32 cL = wave_bcast_u32(c, L);
            const u64 mConn = wave_ballot(has && c == cL);
            if (has && c == cL) firstLane = (u32)__builtin_ctzll(mConn);
            todo &= ~mConn;
          }
          for (u32 x = 0; x < cnt; ++x) {
            const i32 idx = (i32)wave_bcast_u32((u32)nid, (int)x);
            const u32 lenx = wave_bcast_u32(nlen, (int)x);
            const u32 crx = wave_bcast_u32(ncreate, (int)x);
            if (needFirst && myid == idx && len == lenx && crx <= firstLane) myres = x;
          }
          const u64 hres = (u64)wave_shfl_u32((u32)nhash, (int)myres) | ((u64)wave_shfl_u32((u32)(nhash >> 32), (int)myres) << 32);
          if (needFirst) newHash = hres;
        }
        if (collide != 0) {
          // path by path: the distinct connections in path order, the boundary's nodes searched with two ballots each
          cnt = 0;
          todo = act;
          while (todo != 0) {
            const int L = __builtin_ctzll(todo);
            const u32 cL = wave_bcast_u32(c, L);
            const bool inConn = has && c == cL;
            const i32 idL = (i32)wave_bcast_u32((u32)myid, L);
            const u32 lenL = wave_bcast_u32(len, L);
            const u64 hL = (u64)wave_bcast_u32((u32)h, L) | ((u64)wave_bcast_u32((u32)(h >> 32), L) << 32);
            const u32 prevL = wave_bcast_u32(prevMine, L);
            const bool published = (u32)lane < cnt;
            const u64 mSame = wave_ballot(published && nid == idL && nlen == lenL);
            const u64 mHash = wave_ballot(published && nhash == hL);
            const int it = mSame != 0 ? 63 - __builtin_clzll(mSame) : -1;
            const bool merged = it >= 0 && (mHash & ((u64{2} << it) - 1)) != 0;
            u32 res;
            u64 nh;
            if (merged) {
              res = (u32)it;
              nh = (u64)wave_bcast_u32((u32)nhash, it) | ((u64)wave_bcast_u32((u32)(nhash >> 32), it) << 32);
            } else {
              res = cnt;
              nh = hL;
              if ((u32)lane == cnt) {
                nid = idL;
                nlen = lenL;
                nhash = hL;
                nprev = prevL;
              }
              cnt += 1;
            }
            if (inConn) {
              myres = res;
              newHash = nh;
            }
            todo &= ~wave_ballot(inConn);
          }
        }
        if (has) {
          curHash = newHash;
          curB = b;
          curX = myres;
        }
      }
      // back to the row's lanes
      const int back = r == k ? (int)p : lane;
      const u32 vA = wave_shfl_u32(has ? myres : 0xffu << 24, back);
      if (r == k) outA = vA;
      if (cnt != 0) {
        const i32 vI = (i32)wave_shfl_u32((u32)nid, back);
        const u32 vL = wave_shfl_u32(nlen, back), vP = wave_shfl_u32(nprev, back);
        if (r == k) {
          outId = vI;
          outLen = vL;
          outPrev = vP;
          outCnt = cnt;
        }
      }
    }
    JPP_PPROF(3);
    const u32 bq = b0 + r;
    if (r < rowsPer && bq <= bE) {
      const u32 q = bq * G + p;
      g_assign[q] = outA;
      if (p < outCnt) {
        g_id[q] = outId;
        g_len[q] = outLen;
        g_prev[q] = outPrev;
      }
      if (p == 0) g_cnt[bq] = outCnt;
    }
    JPP_PPROF(4);
  }
  wave_sync();
  // dense hidden-state rows: node idx of boundary b lives in row rnn_noff[b] + idx of the sentence's slice of
  // rnn_ctx; row 0 = parking row (boundary 0), row 1 = BOS state (boundary 1), then the rnn nodes in boundary order
  {
    u32 carry = 0;
    for (u32 b0 = 0; b0 <= bE; b0 += 64) {
      const u32 b = b0 + (u32)lane;
      const u32 c = b > bE ? 0u : b < 2 ? 1u : g_cnt[b];
      const u32 incl = wave_scan_incl_u32(c, lane);
      if (b <= bE) B.rnn_noff[bb0 + b] = carry + incl - c;
      carry += wave_shfl_u32(incl, 63);
    }
    if (lane == 0) B.rnn_rows[s] = carry;
  }
  JPP_PPROF(5);
  JPP_PPROF_FLUSH;
}

// The rnn nodes of a sentence in hidden-state row order (round 5): one 16-byte record per row -- handle, word id, row of
// the predecessor, length.  The recurrence (k_rnn_chain) and the scoring of long sentences (k_rnn_score_long) read the
// rnn lattice from here, 64 records per load, instead of staging per-(boundary, index) arrays in LDS: a sentence of any
// length and any beam takes the lock-step matrix-core recurrence (until round 4: at most 288 (boundary, path) slots;
// everything longer ran boundary by boundary in k_rnn_score<.., 3>, a dependent L2 round trip per step).  Launched
// behind the scan of the row counts (the records live at rnn_rowbase[s] + row).
__global__ void __launch_bounds__(64 * kRnnPrepWaves) k_rnn_dense(Batch B, Config cfg) {
  const int wv = (int)(threadIdx.x >> 6);
  const int lane = (int)(threadIdx.x & 63);
  const u32 s = blockIdx.x * kRnnPrepWaves + wv;
  if (s >= B.n_sent) return;
  if (B.gstats[kGstatOverflow] != 0) return;   // (a batch beyond its capacity: the row bases lie outside the table)
  RnnRec* recs = B.rnn_rec + B.rnn_rowbase[s];
  if (lane == 0) recs[0] = RnnRec{0, 0, 0, 0};
  if (B.sent_status[s] != ST_OK) return;
  const u32 n = B.sent_ncp[s];
  if (n == 0) return;
  const u32 bb0 = B.byte_off[s] + 4 * s;
  const u32 bE = n + 2;
  if (B.bnd_ngb[bb0 + bE] == 0) return;
  const u32 G = (u32)cfg.gbeam;
  const u32 nq = (bE + 1) * G;
  const u32* rn_prev = B.rnn_prev + (u64)bb0 * G;
  const i32* rn_id = B.rnn_nid + (u64)bb0 * G;
  const u32* rn_len = B.rnn_nlen + (u64)bb0 * G;
  const u32* rn_cnt = B.rnn_cnt + bb0;
  const u32* noff = B.rnn_noff + bb0;
  if (lane == 0) recs[1] = RnnRec{G, 0, 0, 0};   // BOS (boundary 1, index 0)
  for (u32 b0 = 2; b0 <= bE; b0 += 64) {
    const u32 b = b0 + (u32)lane;
    if (b > bE) continue;
    const u32 cnt = rn_cnt[b], base = noff[b];
    for (u32 idx = 0; idx < cnt; ++idx) {
      const u32 q = b * G + idx;
      const u32 hp = rn_prev[q];
      u32 prow = 0;
      if (hp < nq) {
        const u32 pb = hp / G;
        prow = noff[pb] + (hp - pb * G);
      }
      recs[base + idx] = RnnRec{q, rn_id[q], prow, rn_len[q]};
    }
  }
}

// out[p][:] += W^T[k][:] * ctx[p][k] for every k; lane owns outputs lane*J .. lane*J+J-1
template <int J, int CN>
__device__ __forceinline__ void rnn_matvec(const float* __restrict__ Wt, const float (&ctx)[kRnnCN][J],
                                           float (&acc)[kRnnCN][J], int lane) {
  constexpr int EP = 64 * J;
#pragma unroll 8
  for (int kk = 0; kk < 64; ++kk) {
#pragma unroll
    for (int j2 = 0; j2 < J; ++j2) {
      const float* row = Wt + (kk * J + j2) * EP + lane * J;
      float w[J];
#pragma unroll
      for (int j = 0; j < J; ++j) w[j] = row[j];
#pragma unroll
      for (int p = 0; p < CN; ++p) {
        float c = wave_bcast_f32(ctx[p][j2], kk);
#pragma unroll
        for (int j = 0; j < J; ++j) acc[p][j] = __builtin_fmaf(w[j], c, acc[p][j]);
      }
    }
  }
}

#if !defined(JPP_EMU)
typedef float rnn_f2 __attribute__((ext_vector_type(2)));
#endif

// index of W^T[k][i] in the LDS copy (EP = 64 J, J <= 2): rows are interleaved in pairs so that one read
// gives a lane the 2 J weights of rows k, k+1 for its J outputs -- [k/2][lane][k&1][j]
template <int J>
__device__ __forceinline__ u32 rnn_w2_index(u32 k, u32 i) {
  return ((((k >> 1) * 64u + i / J) * 2u) + (k & 1u)) * J + (i % J);
}

// same sum, same order as rnn_matvec (k ascending), on the pair-interleaved LDS copy: half the LDS
// reads, and for J = 2 both outputs of a lane advance in one packed FMA
template <int J, int CN>
__device__ __forceinline__ void rnn_matvec_lds(const float* __restrict__ W2, const float (&ctx)[kRnnCN][J],
                                               float (&acc)[kRnnCN][J], int lane) {
  static_assert(J == 1 || J == 2, "the LDS copy exists for E <= 128 only");
  constexpr int EP = 64 * J;
#pragma unroll 8
  for (int kp = 0; kp < EP / 2; ++kp) {
    const float* row = W2 + (kp * 64 + lane) * 2 * J;
    float w[2][J];
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int j = 0; j < J; ++j) w[r][j] = row[r * J + j];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int src = J == 2 ? kp : 2 * kp + r;   // lane holding ctx element k = 2 kp + r
      constexpr int kZero = 0;
      const int j2 = J == 2 ? r : kZero;
#pragma unroll
      for (int p = 0; p < CN; ++p) {
        const float c = wave_bcast_f32(j2 == 0 ? ctx[p][0] : ctx[p][J - 1], src);
#if !defined(JPP_EMU)
        if (J == 2) {
          rnn_f2 a = {acc[p][0], acc[p][J - 1]};
          const rnn_f2 ww = {w[r][0], w[r][J - 1]};
          const rnn_f2 cc = {c, c};
          a = __builtin_elementwise_fma(ww, cc, a);
          acc[p][0] = a.x;
          acc[p][J - 1] = a.y;
          continue;
        }
#endif
#pragma unroll
        for (int j = 0; j < J; ++j) acc[p][j] = __builtin_fmaf(w[r][j], c, acc[p][j]);
      }
    }
  }
}

template <int J, int CN, bool WLDS>
__device__ __forceinline__ void rnn_matvec_any(const float* __restrict__ Wt, const float (&ctx)[kRnnCN][J],
                                               float (&acc)[kRnnCN][J], int lane) {
  if constexpr (WLDS) rnn_matvec_lds<J, CN>(Wt, ctx, acc, lane);
  else rnn_matvec<J, CN>(Wt, ctx, acc, lane);
}

// MikolovIndexCalculator::calcIndices + the weight gathers of MikolovScoreCalculator::calcScoresN for one
// (word, history) pair; the caller adds w[0] + w[1] + ... left to right.  Every history slot holds
// prev->id (reference quirk, rnn_scorer_gbeam.cc:171-188), so the context hash of order i is
// base + (prevId + 1) * coef[i].
__device__ __forceinline__ void rnn_maxent_gather(const float JPP_GLOBAL* __restrict__ maxentT, i32 myid, i32 pid, u32 order, u64 mxBase,
                                                  const u64 (&mxCoef)[4], u64 hashMax, u64 hashMagic, float (&w)[4]) {
#pragma unroll
  for (u32 i = 0; i < 4; ++i) {
    if (i < order) {
      const u64 xx = mxBase + ((u64)(i64)pid + 1) * mxCoef[i];
      const u64 h = fastmod_u64(xx, hashMax, hashMagic);
      // (h + id) % hash_max: h < hash_max, so one conditional subtraction does it for any id below
      // hash_max; the general path covers id = -1 (wraps) and oversized ids
      u64 idx = h + (u64)(i64)myid;
      if ((u64)(i64)myid < hashMax) idx = idx >= hashMax ? idx - hashMax : idx;
      else idx = fastmod_u64(idx, hashMax, hashMagic);
      w[i] = maxentT[idx];
    }
  }
}

// MikolovRnnImplParallel::computeContextScores (mikolov_rnn_impl.h:216-223): the element-wise products of
// the NCE row and the context, each rounded to float, added one after the other for k ascending from 0 --
// the order the oracle build's plain loop over `cwiseProduct(...).colwise().sum()` has.  One lane per rnn
// node; the chain is sequential by definition, the loads run ahead of it.
__device__ __forceinline__ float rnn_dot_seq(const float JPP_GLOBAL* __restrict__ nce, const float* __restrict__ ctx, u32 E) {
  float s = 0.f;
  u32 k = 0;
  if ((E & 3u) == 0 && ((size_t)nce & 15u) == 0 && ((size_t)ctx & 15u) == 0) {
    struct alignas(16) F4 {
      float v[4];
    };
    const F4 JPP_GLOBAL* a4 = (const F4 JPP_GLOBAL*)nce;
    const F4* c4 = reinterpret_cast<const F4*>(ctx);
#pragma unroll 4
    for (; k + 4 <= E; k += 4) {
      const F4 a = a4[k >> 2], c = c4[k >> 2];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const float pr = a.v[t] * c.v[t];
        s += pr;
      }
    }
  }
  for (; k < E; ++k) {
    const float pr = nce[k] * ctx[k];
    s += pr;
  }
  return s;
}

// ---- grouping sentences by the length of their recurrence ----
// A lock-step workgroup runs as many rounds as its longest chain, so the sentences are handed to workgroups in
// order of chain length (a counting sort over kRnnOrderBins classes; which sentence shares a workgroup with which
// has no influence on any result).  Sentences k_rnn_score does not stage in LDS form the last class.
constexpr u32 kRnnOrderBins = 128;
constexpr u32 kRnnStageCap = 288, kRnnStageCapB = 48;   // k_rnn_chain's / k_rnn_score's staging limits (connections, boundaries)
constexpr u32 kRnnNodeCap = kRnnStageCap;               // rnn nodes of a staged sentence: at most one per connection
// (a cap of 96 nodes used to send one or two 40-codepoint sentences per 65536 down the serial path -- 0.4 ms for the
// batch, because that launch then lasts as long as its slowest sentence)

// whether a sentence's rnn lattice is staged in LDS by k_rnn_chain / k_rnn_score<.., 2>.  k_rnn_order_key files the
// others in the last class, which k_rnn_score_long serves.
__device__ __forceinline__ bool rnn_stageable(u32 bE, int G, int beam, u32 N) {
  const u32 nq = (bE + 1) * (u32)G;
  return nq <= kRnnStageCap && (bE + 1) <= kRnnStageCapB && (bE + 1) * (((u32)G + kRnnCN - 1) / kRnnCN) <= 2 * kRnnStageCapB &&
         G <= 32 && beam <= 64 && N <= 65535;
}

__global__ void __launch_bounds__(256) k_rnn_order_key(Batch B, Config cfg) {
  __shared__ u32 h[kRnnOrderBins];
  if (threadIdx.x < kRnnOrderBins) h[threadIdx.x] = 0;
  __syncthreads();
  const u32 s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s < B.n_sent) {
    u32 key = 0;
    const u32 n = B.sent_status[s] == ST_OK ? B.sent_ncp[s] : 0u;
    if (n) {
      const u32 bb0 = B.byte_off[s] + 4 * s;
      const u32 bE = n + 2;
      if (B.bnd_ngb[bb0 + bE]) {
        key = kRnnOrderBins - 1;
        if (rnn_stageable(bE, cfg.gbeam, cfg.beam, B.sent_nodes[s])) {
          u32 c = 0;
          for (u32 b = 2; b < bE; ++b) c += B.rnn_cnt[bb0 + b];
          key = c < kRnnOrderBins - 2 ? c : kRnnOrderBins - 2;
        }
      }
    }
    B.rnn_key[s] = key;
    atomicAdd(&h[key], 1u);
  }
  __syncthreads();
  if (threadIdx.x < kRnnOrderBins && h[threadIdx.x]) atomicAdd(&B.rnn_hist[threadIdx.x], h[threadIdx.x]);
}

// one workgroup of kRnnOrderBins threads: class sizes -> first slots; leaves the histogram zeroed for the next batch
__global__ void __launch_bounds__(kRnnOrderBins) k_rnn_order_scan(Batch B) {
  __shared__ u32 h[kRnnOrderBins];
  h[threadIdx.x] = B.rnn_hist[threadIdx.x];
  B.rnn_hist[threadIdx.x] = 0;
  __syncthreads();
  if (threadIdx.x == 0) {
    u32 run = 0;
    for (u32 i = 0; i < kRnnOrderBins; ++i) {
      const u32 x = h[i];
      h[i] = run;
      run += x;
    }
  }
  __syncthreads();
  B.rnn_offs[threadIdx.x] = h[threadIdx.x];
  if (threadIdx.x == kRnnOrderBins - 1) {
    B.rnn_slow[0] = h[threadIdx.x];
    B.rnn_slow[1] = B.n_sent - h[threadIdx.x];   // the last class ends the order
  }
}

__global__ void k_rnn_order_zero(u32* hist) { hist[threadIdx.x] = 0; }

__global__ void __launch_bounds__(256) k_rnn_order_fill(Batch B) {
  // ranks within the workgroup from LDS counters, then one global add per class and workgroup
  __shared__ u32 h[kRnnOrderBins], base[kRnnOrderBins];
  if (threadIdx.x < kRnnOrderBins) h[threadIdx.x] = 0;
  __syncthreads();
  const u32 s = blockIdx.x * blockDim.x + threadIdx.x;
  u32 key = 0, local = 0;
  if (s < B.n_sent) {
    key = B.rnn_key[s];
    local = atomicAdd(&h[key], 1u);
  }
  __syncthreads();
  if (threadIdx.x < kRnnOrderBins && h[threadIdx.x]) base[threadIdx.x] = atomicAdd(&B.rnn_offs[threadIdx.x], h[threadIdx.x]);
  __syncthreads();
  if (s < B.n_sent) B.rnn_order[base[key] + local] = s;
}

// ---- the recurrence (GbeamRnnState::computeContext for every rnn node before EOS), E <= 128 ----
// A workgroup of 16 wavefronts runs 32 sentences in lock step, as two groups of 16 (wavefront w owns sentence w of
// either group).  A group's round r: every owner writes the input context of its r-th chained rnn node (the nodes
// before EOS in boundary order, so a node's predecessor always belongs to an earlier round) into column w of the
// group's 128 x 16 B tile; four wavefronts, one per SIMD (each SIMD has its own matrix pipe; the roles follow
// HW_ID because which wavefronts share a SIMD is the dispatcher's choice), multiply their rows of W -- held in
// registers for the whole kernel -- with the tile: EP/4 v_mfma_f32_16x16x4_f32 steps per row tile, per output
// bitwise the k-ascending fused chain of rnn_matvec (Eigen's gemv with the reference's build flags); the 16 x 16
// results are parked in LDS, where each owner picks up its column, adds the embedding row, applies the sigmoid and
// stores the new context.  The two groups alternate: while the matrix pipes work on one group's tile the owners
// finish the other group's previous round and write its next tile, so a half step costs about one tile product.
// LDS traffic per rnn node: the 4 KB of B operands (a per-sentence matvec would read the 64 KB of W^T per node).
// B tile layout: element (k, n) at ((k >> 4) * 64 + (k & 3) * 16 + n) * 4 + ((k >> 2) & 3), so that the lane
// l = (k & 3) * 16 + n of a computing wavefront reads the operands of four consecutive steps with one 16-byte read.
// Global loads and stores are issued in the same number by every wavefront and half step whether it has a node or
// not (idle ones use row 0 of the context array, the b = 0 row no rnn node owns): with a fixed count the s_waitcnt
// before a use leaves the younger stores in flight instead of draining them (vmcnt is in order; behind a branch the
// compiler has to assume vmcnt(0)).
template <int J>
__global__ void __launch_bounds__(1024) k_rnn_chain(Batch B, const DevModel* __restrict__ Mp, Config cfg) {
  constexpr int EP = 64 * J;
  constexpr int kWaves = 16, NG = 2;
  constexpr int MT = 4;                 // computing wavefronts
  constexpr int TPW = EP / 16 / MT;     // row tiles of W each (two independent accumulator chains for E = 128:
                                        // a single chain waits 40 cycles per 32-cycle step)
  constexpr int kOutStride = EP + 4;
  const DevModel& M = *Mp;
  const int wv = (int)(threadIdx.x >> 6);
  const int lane = (int)(threadIdx.x & 63);
  __shared__ u64 s_exptab[kExp2fN];
  __shared__ __attribute__((aligned(16))) float s_B[NG][EP * 16];
  __shared__ __attribute__((aligned(16))) float s_out[NG][16 * kOutStride];
  __shared__ u32 s_rounds;
  __shared__ u32 s_first[4];            // lowest wavefront of the workgroup on every SIMD
  struct alignas(16) F4 {
    float x, y, z, w;
  };
  if (threadIdx.x < (u32)kExp2fN) s_exptab[threadIdx.x] = exp2f_tab((int)threadIdx.x);
  for (u32 i = threadIdx.x; i < (u32)(NG * EP * 16); i += blockDim.x) (&s_B[0][0])[i] = 0.f;
  if (threadIdx.x == 0) s_rounds = 0;
  if (threadIdx.x < 4) s_first[threadIdx.x] = 0xffffffffu;
  __syncthreads();
  const int G = cfg.gbeam;
  const u32 E = M.rnn_E;
  const float JPP_GLOBAL* __restrict__ embT = as_global(M.rnn_emb);

  // ---- per sentence: BOS state, length of the chain, the rnn-node records (k_rnn_dense) ----
  // Round r of a sentence makes the context of row 2 + r (rows are in boundary order: a predecessor's row is always
  // lower).  The records of 64 rounds sit in one register per field, lane i holding row 2 + recBase + i, reloaded every
  // 64 rounds.  Every wavefront issues the same loads at the same rounds (the rounds are the workgroup's), sentences
  // without a chain read their parking row.
  u32 nchain[NG];
  float* rn_ctx[NG];
  const RnnRec* recs[NG];
  u32 recBase[NG];
  u32 curPrev[NG];
  i32 curId[NG];
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    const u32 slot = (blockIdx.x * NG + (u32)g) * kWaves + (u32)wv;
    const u32 s = slot < B.n_sent ? B.rnn_order[slot] : 0u;
    bool own = slot < B.n_sent && B.sent_status[s] == ST_OK;
    const u32 n = own ? B.sent_ncp[s] : 0u;
    const u32 bb0 = B.byte_off[s] + 4 * s;
    const u32 bE = n + 2;
    own = own && n != 0 && B.bnd_ngb[bb0 + bE] != 0;
    // (not own: some sentence's row 0 serves as the parking row; a batch that overflowed its capacity -- every sentence
    // failed by k_cap_guard, the row bases beyond the table -- parks in the table's first row)
    const u64 rowBase = B.gstats[kGstatOverflow] != 0 ? u64{0} : B.rnn_rowbase[s];
    rn_ctx[g] = B.rnn_ctx + rowBase * (u64)EP;
    recs[g] = B.rnn_rec + rowBase;
    nchain[g] = 0;
    if (own) {
      // BOS state: sigmoid(W^T 0 + emb[0])  (GbeamRnnFactoryState::computeBosState)
#pragma unroll
      for (int j = 0; j < J; ++j) {
        const u32 i = (u32)lane + 64u * j;
        float v = 0.f;
        if (i < E) {
          const float x = 0.f + embT[i];
          v = sigmoid_ref(x, s_exptab);
        }
        rn_ctx[g][(u64)1 * EP + i] = v;   // row 1
      }
      nchain[g] = B.rnn_noff[bb0 + bE] - 2;   // rnn nodes before the EOS boundary (the EOS nodes are only scored)
    }
    recBase[g] = 0;
    {
      const u32 r0 = (u32)lane;
      const RnnRec a = recs[g][r0 < nchain[g] ? 2 + r0 : 0];
      curPrev[g] = a.prevrow;
      curId[g] = a.id;
    }
  }
  wave_sync();
  const u32 simd = wave_simd_id();
  if (lane == 0) {
    const u32 m = nchain[0] > nchain[1] ? nchain[0] : nchain[1];
    if (m) atomicMax(&s_rounds, m);
    atomicMin(&s_first[simd], (u32)wv);
  }
  __syncthreads();
  const u32 rounds = s_rounds;
  // (a workgroup that does not reach all four SIMDs falls back to its first four wavefronts)
  const bool spread = s_first[0] != 0xffffffffu && s_first[1] != 0xffffffffu && s_first[2] != 0xffffffffu && s_first[3] != 0xffffffffu;
  const int role = spread ? (s_first[simd] == (u32)wv ? (int)simd : -1) : (wv < MT ? wv : -1);
  // A operands of this wavefront's row tiles: lane l holds W[16 (TPW role + t) + (l & 15)][4 kk + (l >> 4)] for every step kk
  float wA[TPW][EP / 4];
#pragma unroll
  for (int t = 0; t < TPW; ++t)
#pragma unroll
    for (int kk = 0; kk < EP / 4; ++kk)
      wA[t][kk] = role >= 0 ? as_global(M.rnn_wt)[(u32)(4 * kk + (lane >> 4)) * EP + 16u * (u32)(TPW * role + t) + (u32)(lane & 15)] : 0.f;

  float lastY[NG][J], emb1[NG][J], c1[NG][J], embv[NG][J];
  u32 lastQ[NG], q1[NG], hnd1[NG], qcur[NG];
  bool has[NG];
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    lastQ[g] = 0xffffffffu;
    q1[g] = hnd1[g] = qcur[g] = 0;
    has[g] = false;
#pragma unroll
    for (int j = 0; j < J; ++j) lastY[g][j] = emb1[g][j] = c1[g][j] = embv[g][j] = 0.f;
  }
  // loads of group g's round r: predecessor's context (this lane owns elements lane and lane + 64) and embedding row
  auto fetch = [&](int g, u32 r) {
    u32 eid = 0;
    q1[g] = hnd1[g] = 0;   // (rows: the node of round r is row 2 + r, hnd1 the row of its predecessor; row 0 parks)
    if (r >= recBase[g] + 64u) {   // (wave- and workgroup-uniform: r is the workgroup's round; one wait per 64 rounds)
      recBase[g] += 64u;
      const u32 rn = recBase[g] + (u32)lane;
      const RnnRec c = recs[g][rn < nchain[g] ? 2 + rn : 0];
      curPrev[g] = c.prevrow;
      curId[g] = c.id;
    }
    {
      const int src = (int)(r - recBase[g]);
      const u32 pr = wave_bcast_u32(curPrev[g], src);
      const i32 id = (i32)wave_bcast_u32((u32)curId[g], src);
      if (r < nchain[g]) {
        q1[g] = 2 + r;
        hnd1[g] = pr;
        eid = id == -1 ? 0u : (u32)id;
      }
    }
#pragma unroll
    for (int j = 0; j < J; ++j) c1[g][j] = rn_ctx[g][(u64)hnd1[g] * EP + (u32)lane + 64u * j];   // (stale while that row is being made: then lastY is used)
#pragma unroll
    for (int j = 0; j < J; ++j) {
      const u32 k = (u32)lane + 64u * j;
      emb1[g][j] = embT[(u64)eid * E + (k < E ? k : 0u)];
    }
  };
  // group g's round r: column of the B tile (a wavefront without a node writes its column too: nobody looks at the result)
  auto write_b = [&](int g, u32 r) {
    has[g] = r < nchain[g];
    qcur[g] = q1[g];
#pragma unroll
    for (int j = 0; j < J; ++j) {
      const u32 k = (u32)lane + 64u * j;
      embv[g][j] = emb1[g][j];
      s_B[g][((k >> 4) * 64 + (k & 3) * 16 + (u32)wv) * 4 + ((k >> 2) & 3)] = hnd1[g] == lastQ[g] ? lastY[g][j] : c1[g][j];
    }
  };
  auto tile_product = [&](int g) {
    MfmaAcc acc[TPW];
#pragma unroll
    for (int t = 0; t < TPW; ++t) acc[t] = MfmaAcc{{0.f, 0.f, 0.f, 0.f}};
    const F4* bt = reinterpret_cast<const F4*>(s_B[g]);
#pragma unroll
    for (int x4 = 0; x4 < EP / 16; ++x4) {
      const F4 b4 = bt[x4 * 64 + lane];
      const float bv[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
      for (int x = 0; x < 4; ++x)
#pragma unroll
        for (int t = 0; t < TPW; ++t) acc[t] = mfma_f32_16x16x4(wA[t][4 * x4 + x], bv[x], acc[t]);
    }
    // D[4 (l >> 4) + i][l & 15]: outputs 16 tile + 4 (l >> 4) + i of the sentence in column l & 15
#pragma unroll
    for (int t = 0; t < TPW; ++t)
      *reinterpret_cast<F4*>(&s_out[g][(lane & 15) * kOutStride + 16 * (TPW * role + t) + 4 * (lane >> 4)]) =
          F4{acc[t].v[0], acc[t].v[1], acc[t].v[2], acc[t].v[3]};
  };
  // group g's pending round: column of the product + embedding row -> sigmoid -> the node's context
  auto finish = [&](int g) {
    float y[J];
#pragma unroll
    for (int j = 0; j < J; ++j) y[j] = 0.f;
    if (has[g]) {
#pragma unroll
      for (int j = 0; j < J; ++j) {
        const u32 k = (u32)lane + 64u * j;
        const float x = s_out[g][wv * kOutStride + (int)k] + embv[g][j];
        y[j] = k < E ? sigmoid_ref(x, s_exptab) : 0.f;
        lastY[g][j] = y[j];
      }
      lastQ[g] = qcur[g];
    }
#pragma unroll
    for (int j = 0; j < J; ++j) rn_ctx[g][(u64)qcur[g] * EP + (u32)lane + 64u * j] = y[j];   // (row 0, the parking row, without a node)
    has[g] = false;
    qcur[g] = 0;
  };
  fetch(0, 0);
  fetch(1, 0);
  vm_wait_all();   // (the loop is entered with nothing in flight: its waits are then the ones of the steady state)
  write_b(0, 0);
  lds_barrier();
  for (u32 r = 0; r < rounds; ++r) {
    // matrix pipes: group 0, round r | owners: finish group 1's round r - 1, write its round r
    fetch(0, r + 1);
    if (role >= 0) tile_product(0);
    finish(1);
    write_b(1, r);
    lds_barrier();
    // matrix pipes: group 1, round r | owners: finish group 0's round r, write its round r + 1
    fetch(1, r + 1);
    if (role >= 0) tile_product(1);
    finish(0);
    write_b(0, r + 1);
    lds_barrier();
  }
  finish(1);
}

// ScoreProcessor::remakeEosBeam (score_processor.cc:548-576): the EOS beam from the adjusted totals of the EOS paths
// (full[p] = weighted local score + total of the path's previous element, prev_total[p] = the latter), one wavefront.
// makeT0Beam on the EOS candidates: util::partition beyond beam*4/3, introsort beyond 16 -- both only permute, so with
// pairwise distinct totals the stable rank is their result; the step-by-step replay runs when two candidates tie exactly.
template <bool SORT>
__device__ __forceinline__ void rnn_remake_eos(const Batch& B, BeamSlot* beams, const u32* en, u32 efirstE, u32 bb0, u32 bE, u32 N,
                                               int G, int beam, int ngb, const float* full, const float* prev_total, int lane) {
    BeamSlot* row = beams + (u64)(N - 1) * beam;
    const int partB = beam * 4 / 3;
    // makeT0Beam on the EOS candidates: util::partition beyond beam*4/3, introsort beyond 16.  Both only
    // permute, so with pairwise distinct totals the stable rank below is their result (see k_sweep 5c); the
    // step-by-step replay runs only when two candidates tie exactly.
    bool replay = false;
    if (SORT && (ngb > 16 || ngb > partB)) {
      bool tie = false;
      if (lane < ngb) {
        const float me = full[lane];
        for (int j = 0; j < ngb; ++j) tie = tie || (j != lane && full[j] == me);
      }
      replay = wave_ballot(tie) != 0;
    }
    if (replay) {
      if (lane == 0) {
        u8 idx[kMaxGbeam];
        for (int z = 0; z < ngb; ++z) idx[z] = (u8)z;
        auto comp = [full](u8 a, u8 bb) { return full[a] > full[bb]; };
        u8* itr = idx + ngb;
        if (ngb > partB) itr = jpp_partition(idx, itr, comp, (long)beam, (long)partB);
        std_sort(idx, itr, comp);
        const int have = (int)(itr - idx);
        for (int z = 0; z < beam; ++z) {
          if (z < have) {
            GbeamEntry ge = B.bnd_gbeam[(u64)(bb0 + bE) * G + idx[z]];
            row[z] = BeamSlot{ge.left, ge.beam, full[idx[z]], en[efirstE + ge.left], (u32)idx[z]};
          } else {
            row[z] = BeamSlot{kFake16, kFake16, 0.f, 0xffffffffu, 0};
          }
        }
      }
      wave_sync();
      if (lane < ngb) B.bnd_gbeam[(u64)(bb0 + bE) * G + lane].score = prev_total[lane];
    } else if (lane < kMaxGbeam) {
      if (lane < ngb) {
        float me = full[lane];
        int rank = 0;
        for (int j = 0; j < ngb; ++j) {
          float o = full[j];
          if (o > me || (o == me && j < lane)) ++rank;
        }
        GbeamEntry ge = B.bnd_gbeam[(u64)(bb0 + bE) * G + lane];
        if (rank < beam) row[rank] = BeamSlot{ge.left, ge.beam, me, en[efirstE + ge.left], (u32)lane};
        B.bnd_gbeam[(u64)(bb0 + bE) * G + lane].score = prev_total[lane];
      } else if (lane < beam) {
        row[lane] = BeamSlot{kFake16, kFake16, 0.f, 0xffffffffu, 0};
      }
    }
}

// SORT: compile the makeT0Beam replay for remakeEosBeam (needed beyond 16 candidates / beam*4/3 only)
// MODE 0 (E > 128): one launch does everything, W streamed from L2.
// MODE 2 (E <= 128, after k_rnn_chain): everything but the recurrence (maxent sums, NCE dot products, score cells,
// adjustBeamScores, remakeEosBeam) for the sentences k_rnn_chain handled, reading the contexts it left in HBM/L2.
// MODE 3 (E <= 128): the complete boundary-by-boundary path for the sentences beyond the LDS staging limits, which
// the other two kernels skip; 16 wavefronts per workgroup share a pair-interleaved copy of W^T in LDS
// (rnn_matvec_lds).  A launch of its own so that its registers and LDS do not set MODE 2's occupancy.
// Four wavefronts (sentences) per workgroup and no workgroup barrier after the start: several workgroups per CU
// hide each other's load latency.
template <int J, bool SORT, int MODE>
__global__ void __launch_bounds__(MODE == 3 ? 1024 : 256) k_rnn_score(Batch B, const DevModel* __restrict__ Mp, Config cfg) {
  static_assert(MODE == 0 || ((MODE == 2 || MODE == 3) && J <= 2), "k_rnn_chain covers E <= 128");
  constexpr int kWaves = MODE == 3 ? 16 : 4;
  constexpr bool kWinLds = MODE == 3;
  constexpr int EP = 64 * J;
  const DevModel& M = *Mp;
  const int wv = (int)(threadIdx.x >> 6);
  const int lane = (int)(threadIdx.x & 63);
  __shared__ u64 s_exptab[kExp2fN];   // 2^(i/32) table of expf_libm: lanes index it divergently
  if (threadIdx.x < (u32)kExp2fN) s_exptab[threadIdx.x] = exp2f_tab((int)threadIdx.x);
  __shared__ float s_W[kWinLds ? EP * EP : 1];
  if (kWinLds) {
    for (u32 q = threadIdx.x; q < (u32)(EP * EP); q += blockDim.x) s_W[rnn_w2_index<(J <= 2 ? J : 1)>(q / EP, q % EP)] = M.rnn_wt[q];
  }
  __syncthreads();
  const float* __restrict__ Wt = kWinLds ? s_W : M.rnn_wt;
  // MODE 3: a fixed, small grid walks the sentences of the last class (the tail of the chain-length order; usually
  // there are none, and a full-size launch of empty workgroups costs more than the other RNN kernels' early exits)
  u32 slot = blockIdx.x * kWaves + wv;
  const u32 nslow = MODE == 3 ? B.rnn_slow[1] : 0u;
  if (MODE == 3 && slot >= nslow) return;
  do {
  const u32 s = MODE == 3 ? B.rnn_order[B.rnn_slow[0] + slot] : slot;
  if (s >= B.n_sent) return;
  if (B.sent_status[s] != ST_OK) return;
  const u32 off = B.byte_off[s];
  const u32 bb0 = off + 4 * s;
  const u32 n = B.sent_ncp[s];
  if (n == 0) return;
  const u32 N = B.sent_nodes[s];
  const u64 nb = B.node_base[s];
  const int beam = cfg.beam;
  const int G = cfg.gbeam;
  const int S = cfg.nscorers;
  const u32 bE = n + 2;
  const int ngb = (int)B.bnd_ngb[bb0 + bE];
  if (ngb == 0) return;
  const u32 E = M.rnn_E;
  // model scalars and table pointers, read once: going through `M` inside the loops makes the compiler
  // re-issue the scalar loads after every store (it cannot prove the header is not aliased)
  const float JPP_GLOBAL* __restrict__ embT = as_global(M.rnn_emb);
  const float JPP_GLOBAL* __restrict__ nceT = as_global(M.rnn_nce);
  const float JPP_GLOBAL* __restrict__ maxentT = as_global(M.rnn_maxent);
  const u32 mxOrder = M.rnn_order;
  const u64 hashMax = M.rnn_hash_max, hashMagic = M.rnn_hash_magic, mxBase = M.rnn_mx_base;
  const u64 mxCoef[4] = {M.rnn_mx_coef[0], M.rnn_mx_coef[1], M.rnn_mx_coef[2], M.rnn_mx_coef[3]};
  const float nceConst = M.rnn_nce_const, unkConst = M.rnn_unk_const, unkLen = M.rnn_unk_len;
  const i32 unkId = M.rnn_unk_id;
  BeamSlot* beams = B.node_beam + nb * beam;
  const u32* en = B.end_nodes + nb;
  const u32* conn = B.rnn_conn + (u64)bb0 * G;
  const u32* assign = B.rnn_assign + (u64)bb0 * G;
  const u32* g_len = B.rnn_nlen + (u64)bb0 * G;
  float* rn_ctx = B.rnn_ctx + B.rnn_rowbase[s] * (u64)EP;   // the sentence's hidden-state rows (rnn_noff)

  // the per-node fields the boundary loop depends on are staged in LDS when they fit
  constexpr u32 kCap = kRnnStageCap, kCapB = kRnnStageCapB;
  __shared__ u16 l_prev_all[kWaves][kCap];
  __shared__ i32 l_id_all[kWaves][kCap];
  __shared__ u8 l_cnt_all[kWaves][kCapB];
  __shared__ u16 l_noff_all[kWaves][kCapB];
  // per connection (boundary, path): lattice node | slot << 16 | rnn node << 22 | gbeam index << 27, perceptron score cell
  __shared__ u32 l_conn_all[kWaves][kCap];
  __shared__ float l_cell0_all[kWaves][kCap];
  __shared__ float l_mx_all[kWaves][kCap];
  constexpr u32 kPassCap = 2 * kCapB;
  __shared__ u16 l_pass_all[kWaves][kPassCap];
  __shared__ u32 l_npass_all[kWaves];
  constexpr u32 kNodeCap = kRnnNodeCap;
  __shared__ u16 l_node_all[kWaves][kNodeCap];   // rnn nodes (boundary * G + index) in boundary order
  __shared__ u32 l_nnode_all[kWaves];
  __shared__ float nscore_all[kWaves][kMaxGbeam];
  __shared__ float full_all[kWaves][kMaxGbeam];
  __shared__ float prev_total_all[kWaves][kMaxGbeam];
  float* nscore = nscore_all[wv];
  float* full = full_all[wv];
  float* prev_total = prev_total_all[wv];
  const u32 nq = (bE + 1) * (u32)G;
  const u32* rn_prev = B.rnn_prev + (u64)bb0 * G;
  const i32* rn_id = B.rnn_nid + (u64)bb0 * G;
  const u32* rn_cnt = B.rnn_cnt + bb0;
  const bool staged = MODE != 3 && rnn_stageable(bE, G, beam, N);
  if (MODE == 2 && !staged) return;   // (k_rnn_score_long takes it)
  const bool inLds = MODE == 2 || (MODE == 0 && staged);   // a compile-time constant in MODE 2 and 3
  if (inLds) {
    for (u32 q = lane; q < nq; q += 64) {
      l_prev_all[wv][q] = (u16)rn_prev[q];   // handles are < nq; the BOS node's "none" is never followed
      l_id_all[wv][q] = rn_id[q];
    }
    for (u32 q = lane; q <= bE; q += 64) {
      l_cnt_all[wv][q] = (u8)rn_cnt[q];
      l_noff_all[wv][q] = (u16)B.rnn_noff[bb0 + q];
    }
    rn_id = l_id_all[wv];
    // connections: every load below is independent, so they are all in flight together
    const u32* g_gi = B.rnn_gi + (u64)bb0 * G;
    for (u32 q = lane; q < nq; q += 64) {
      const u32 c = conn[q];
      const u32 gi = g_gi[q] & 0xffffu;
      l_conn_all[wv][q] = c == kNoConn ? kNoConn : ((c & 0xffffu) | ((c >> 26) << 16) | ((assign[q] & 31u) << 22) | ((gi & 31u) << 27));
      l_cell0_all[wv][q] = c != kNoConn ? B.node_cells[((nb + (c & 0x03ffffffu)) * G + gi) * S] : 0.f;
    }
  }
  const u32* l_conn = l_conn_all[wv];
  const float* l_cell0 = l_cell0_all[wv];
  // hidden-state row of rnn-node handle h = b * G + idx
  const u32 invG = small_div_inv((u32)G);
  const u32* g_noff = B.rnn_noff + bb0;
  auto ctx_row = [&](u32 h) -> u32 {
    if (inLds) {
      const u32 hb = small_div(h, invG);   // (staged: h < kRnnStageCap)
      return (u32)l_noff_all[wv][hb] + (h - hb * (u32)G);
    }
    const u32 hb = h / (u32)G;
    return g_noff[hb] + (h - hb * (u32)G);
  };
  float prevT = 0.f;  // running total of this lane's path (adjustBeamScores), BOS element total = 0
  JPP_RPROF_DECL;
  JPP_RPROF(0);
  // ---- C. contexts and scores, boundary by boundary ----
  // BOS state: sigmoid(W^T 0 + emb[0])  (GbeamRnnFactoryState::computeBosState)
#pragma unroll
  for (int j = 0; j < J; ++j) {
    u32 i = (u32)lane * J + j;
    float v = 0.f;
    if (i < E) {
      float x = 0.f + embT[i];
      v = sigmoid_ref(x, s_exptab);
    }
    if (MODE != 2) rn_ctx[(u64)1 * EP + i] = v;   // row 1 (k_rnn_chain made the staged sentences' BOS state)
  }
  wave_sync();
  float* l_mx = l_mx_all[wv];
  const u16* l_prev = l_prev_all[wv];
  const u8* l_cnt = l_cnt_all[wv];
  u16* l_pass = l_pass_all[wv];
  u16* l_node = l_node_all[wv];
  u32 npass = 0, nnode = 0;
  if (inLds) {
    // Every word id is known before the recurrence starts, so everything that does not feed the next
    // context is taken off the serial chain.  Prologue: the maxent sums of all rnn nodes, one lane per
    // node, the pass list and the node list.  Chain: context -> matvec -> sigmoid -> store, nothing else.
    // Epilogue: the NCE dot products and scores, one lane per rnn node (the contexts are all in HBM/L2 by
    // then), the score cells, and adjustBeamScores along the paths.
    {
      // all gathers of all rounds go out before the first sum (one HBM round trip, not one per round)
      constexpr u32 kRounds = (kCap + 63) / 64;
      float mw[kRounds][4];
#pragma unroll
      for (u32 r = 0; r < kRounds; ++r) {
        const u32 q = (u32)lane + 64u * r;
        const u32 bq = q / (u32)G, iq = q - bq * (u32)G;
#pragma unroll
        for (u32 i = 0; i < 4; ++i) mw[r][i] = 0.f;
        if (q < nq && bq >= 2 && bq <= bE && iq < l_cnt[bq])
          rnn_maxent_gather(maxentT, rn_id[q], rn_id[l_prev[q]], mxOrder, mxBase, mxCoef, hashMax, hashMagic, mw[r]);
      }
#pragma unroll
      for (u32 r = 0; r < kRounds; ++r) {
        const u32 q = (u32)lane + 64u * r;
        float me = mw[r][0];   // calcScoresN: w0 + w1 + ... left to right
#pragma unroll
        for (u32 i = 1; i < 4; ++i)
          if (i < mxOrder) me += mw[r][i];
        if (q < nq) l_mx[q] = me;
      }
    }
    // pass list: boundary | first node << 6 | (nodes - 1) << 11 | last pass of the boundary << 13;
    // node list: q = boundary * G + index of every rnn node, in boundary order
    {
      // lane b lists the passes / nodes of boundary b (bE + 1 <= kCapB <= 64) at the offsets exclusive scans give it
      const u32 bq = (u32)lane;
      const u32 cnt = (bq >= 2 && bq <= bE) ? l_cnt[bq] : 0u;
      const u32 mine = (cnt + kRnnCN - 1) / kRnnCN;
      const u32 incl = wave_scan_incl_u32(mine, lane);
      u32 np = incl - mine;
      for (u32 c0 = 0; c0 < cnt; c0 += kRnnCN) {
        const u32 cn = (cnt - c0) < (u32)kRnnCN ? (cnt - c0) : (u32)kRnnCN;
        l_pass[np++] = (u16)(bq | (c0 << 6) | ((cn - 1) << 11) | ((c0 + kRnnCN >= cnt ? 1u : 0u) << 13));
      }
      const u32 inclN = wave_scan_incl_u32(cnt, lane);
      u32 nn = inclN - cnt;
      for (u32 i = 0; i < cnt; ++i) l_node[nn++] = (u16)(bq * (u32)G + i);
      if (lane == 63) {
        l_npass_all[wv] = incl;
        l_nnode_all[wv] = inclN;
      }
    }
    wave_sync();
    npass = l_npass_all[wv];
    nnode = l_nnode_all[wv];
  }
  JPP_RPROF(1);
  if (inLds && MODE == 0) {
    float embR[kRnnCN][J];                    // embedding rows of the current pass
    float lastOut[kRnnCN][J];                 // contexts produced by the previous pass, kept in registers
    u32 lastBase = 0xffffffffu, lastCn = 0;
#pragma unroll
    for (int p = 0; p < kRnnCN; ++p)
#pragma unroll
      for (int j = 0; j < J; ++j) embR[p][j] = lastOut[p][j] = 0.f;
    auto load_rows = [&](u32 pe, const float JPP_GLOBAL* __restrict__ table, bool wanted, float (&out)[kRnnCN][J]) {
      const u32 b = pe & 63u, c0 = (pe >> 6) & 31u, cn = ((pe >> 11) & 3u) + 1;
#pragma unroll
      for (int p = 0; p < kRnnCN; ++p) {
        if ((u32)p < cn) {
          const i32 id = rn_id[b * (u32)G + c0 + p];
          const u32 eid = id == -1 ? 0u : (u32)id;
#pragma unroll
          for (int j = 0; j < J; ++j) {
            const u32 i = (u32)lane * J + j;
            out[p][j] = (i < E && wanted) ? table[(u64)eid * E + i] : 0.f;
          }
        }
      }
    };
    for (u32 pi = 0; pi < npass; ++pi) {
      const u32 pe = l_pass[pi];
      const u32 b = pe & 63u, c0 = (pe >> 6) & 31u;
      const int cn = (int)((pe >> 11) & 3u) + 1;
      if (b >= bE) break;   // the EOS nodes are only scored; passes are in boundary order, so nothing follows
      // contexts of the predecessors: straight from the registers when the previous pass made them,
      // otherwise from HBM/L2 (written at least one boundary, i.e. one wave_sync, ago)
      float ctx[kRnnCN][J];
#pragma unroll
      for (int p = 0; p < kRnnCN; ++p) {
#pragma unroll
        for (int j = 0; j < J; ++j) ctx[p][j] = 0.f;
        if (p < cn) {
          const u32 hnd = l_prev[b * (u32)G + c0 + p];
          const u32 rel = hnd - lastBase;
          if (rel < lastCn) {
#pragma unroll
            for (int j = 0; j < J; ++j)
              ctx[p][j] = rel == 0 ? lastOut[0][j] : rel == 1 ? lastOut[1][j] : rel == 2 ? lastOut[2][j] : lastOut[3][j];
          } else {
            const float* cp = rn_ctx + (u64)ctx_row(hnd) * EP + (u32)lane * J;
#pragma unroll
            for (int j = 0; j < J; ++j) ctx[p][j] = cp[j];
          }
        }
      }
      load_rows(pe, embT, true, embR);
      JPP_RPROF_COUNT(cn);
      JPP_RPROF(2);
      {  // new contexts (GbeamRnnState::computeContext)
        float acc[kRnnCN][J];
#pragma unroll
        for (int p = 0; p < kRnnCN; ++p)
#pragma unroll
          for (int j = 0; j < J; ++j) acc[p][j] = 0.f;
          switch (cn) {
            case 1: rnn_matvec_any<J, 1, kWinLds>(Wt, ctx, acc, lane); break;
            case 2: rnn_matvec_any<J, 2, kWinLds>(Wt, ctx, acc, lane); break;
            case 3: rnn_matvec_any<J, 3, kWinLds>(Wt, ctx, acc, lane); break;
            default: rnn_matvec_any<J, 4, kWinLds>(Wt, ctx, acc, lane); break;
          }
#pragma unroll
        for (int p = 0; p < kRnnCN; ++p) {
          if (p < cn) {
            float* op = rn_ctx + (u64)ctx_row(b * (u32)G + c0 + (u32)p) * EP + (u32)lane * J;
#pragma unroll
            for (int j = 0; j < J; ++j) {
              const u32 i = (u32)lane * J + j;
              const float x = acc[p][j] + embR[p][j];
              const float y = i < E ? sigmoid_ref(x, s_exptab) : 0.f;
              op[j] = y;
              lastOut[p][j] = y;
            }
          }
        }
        lastBase = b * (u32)G + c0;
        lastCn = (u32)cn;
      }
      JPP_RPROF(3);
    }
  }
  if (inLds) {
    wave_sync();
    // ---- scores: one lane per rnn node ----
    // computeContextScores: (nce row .* context).colwise().sum() = the rounded products added for k ascending;
    // computeMaxentScores: += w0 + w1 + ...; applyNceConstant: -= c   (mikolov_rnn_impl.h:216-243);
    // the UNK word scores unkConstantTerm + unkLengthPenalty * length (one fused multiply-add on the
    // reference's FMA build, rnn_scorer_gbeam.cc:239)
    for (u32 k0 = 0; k0 < nnode; k0 += 64) {
      const u32 k = k0 + (u32)lane;
      if (k < nnode) {
        const u32 q = l_node[k];
        const i32 id = rn_id[q];
        float score;
        if (id == unkId) {
          score = __builtin_fmaf(unkLen, (float)g_len[q], unkConst);
        } else if (mxOrder == 0) {
          score = 0.f - nceConst;  // MikolovScoreCalculator::addScores case 0 zero-fills the result
        } else {
          const u32 eid = id == -1 ? 0u : (u32)id;
          score = rnn_dot_seq(nceT + (u64)eid * E, rn_ctx + (u64)ctx_row(l_prev[q]) * EP, E);
          score += l_mx[q];
          score -= nceConst;
        }
        l_mx[q] = score;
      }
    }
    wave_sync();
    JPP_RPROF(4);
    // ---- score cells of the connections (one lane per connection) ----
    for (u32 q = lane; q < nq; q += 64) {
      const u32 c = l_conn[q];
      if (c != kNoConn) {
        const u32 b = q / (u32)G;
        B.node_cells[((nb + (c & 0xffffu)) * G + (c >> 27)) * S + 1] = l_mx[b * (u32)G + ((c >> 22) & 31u)];
      }
    }
    // ---- ScoreProcessor::adjustBeamScores along the EOS paths (lane = path), remakeEosBeam inputs ----
    if (lane < ngb) {
      for (u32 b = 2; b <= bE; ++b) {
        const u32 q = b * (u32)G + (u32)lane;
        const u32 c = l_conn[q];
        if (c == kNoConn) continue;
        const u32 nd = c & 0xffffu, k = (c >> 16) & 63u;
        const float rs = l_mx[b * (u32)G + ((c >> 22) & 31u)];
        const float local = weighted_score2(l_cell0[q], rs, cfg);
        if (b < bE) {
          const float tot = local + prevT;
          beams[(u64)nd * beam + k].total = tot;
          prevT = tot;
        } else {
          full[lane] = local + prevT;  // remakeEosBeam: fullScores[i] = localScore + beamScore
          prev_total[lane] = prevT;
        }
      }
    }
    wave_sync();
  } else {
    for (u32 b = 2; b <= bE; ++b) {
      const int cnt = (int)rn_cnt[b];
      if (cnt == 0) continue;
      for (int c0 = 0; c0 < cnt; c0 += kRnnCN) {
        const int cn = (cnt - c0) < kRnnCN ? (cnt - c0) : kRnnCN;
        // maxent part of the scores first: its gathers depend on the word ids only, so they are in flight
        // together with the context / embedding loads below.  Every context slot holds prev->id (reference
        // quirk, rnn_scorer_gbeam.cc:171-188), so the context hash of order i is base + (prevId + 1) * coef[i].
        float mw[4] = {0.f, 0.f, 0.f, 0.f};  // consumed only when the scores are formed, after the other loads went out
        i32 myid = 0;
        if (lane < cn) {
          myid = rn_id[(u64)b * G + c0 + lane];
          const i32 pid = rn_id[rn_prev[(u64)b * G + c0 + lane]];
          const u32 order = mxOrder;
#pragma unroll
          for (u32 i = 0; i < 4; ++i) {
            mw[i] = 0.f;
            if (i < order) {
              const u64 xx = mxBase + ((u64)(i64)pid + 1) * mxCoef[i];
              const u64 h = fastmod_u64(xx, hashMax, hashMagic);
              // (h + id) % hash_max: h < hash_max, so one conditional subtraction does it for any id below
              // hash_max; the general path covers id = -1 (wraps) and oversized ids
              u64 idx = h + (u64)(i64)myid;
              if ((u64)(i64)myid < hashMax) idx = idx >= hashMax ? idx - hashMax : idx;
              else idx = fastmod_u64(idx, hashMax, hashMagic);
              mw[i] = maxentT[idx];
            }
          }
        }
        JPP_RPROF(1);
        JPP_RPROF_COUNT(cn);
        float ctx[kRnnCN][J];
        float embv[kRnnCN][J];
#pragma unroll
        for (int p = 0; p < kRnnCN; ++p) {
#pragma unroll
          for (int j = 0; j < J; ++j) {
            ctx[p][j] = 0.f;
            embv[p][j] = 0.f;
          }
          if (p < cn) {
            const u32 hnd = rn_prev[(u64)b * G + c0 + p];
            const i32 id = rn_id[(u64)b * G + c0 + p];
            const u32 eid = id == -1 ? 0u : (u32)id;
            const float* cp = rn_ctx + (u64)ctx_row(hnd) * EP + (u32)lane * J;
#pragma unroll
            for (int j = 0; j < J; ++j) ctx[p][j] = cp[j];
#pragma unroll
            for (int j = 0; j < J; ++j) {
              u32 i = (u32)lane * J + j;
              if (i < E && b < bE) embv[p][j] = embT[(u64)eid * E + i];
            }
          }
        }
        // score of rnn node c0 + x on lane x: the sequential NCE dot product (see rnn_dot_seq), the maxent
        // sum, the NCE constant; the UNK word's score is one fused multiply-add (rnn_scorer_gbeam.cc:239)
        if (lane < cn) {
          const int x = lane;
          const i32 id = myid;
          const u32 order = mxOrder;
          float score;
          if (id == unkId) {
            score = __builtin_fmaf(unkLen, (float)g_len[(u64)b * G + c0 + x], unkConst);
          } else if (order == 0) {
            score = 0.f - nceConst;  // MikolovScoreCalculator::addScores case 0 zero-fills the result
          } else {
            const u32 eid = id == -1 ? 0u : (u32)id;
            score = rnn_dot_seq(nceT + (u64)eid * E, rn_ctx + (u64)ctx_row(rn_prev[(u64)b * G + c0 + x]) * EP, E);
            float me = mw[0];
#pragma unroll
            for (u32 i = 1; i < 4; ++i)
              if (i < order) me += mw[i];
            score += me;
            score -= nceConst;
          }
          nscore[c0 + x] = score;
        }
        JPP_RPROF(2);
        // new contexts (GbeamRnnState::computeContext; not needed for EOS)
        if (b < bE) {
          float acc[kRnnCN][J];
#pragma unroll
          for (int p = 0; p < kRnnCN; ++p)
#pragma unroll
            for (int j = 0; j < J; ++j) acc[p][j] = 0.f;
          switch (cn) {
            case 1: rnn_matvec_any<J, 1, kWinLds>(Wt, ctx, acc, lane); break;
            case 2: rnn_matvec_any<J, 2, kWinLds>(Wt, ctx, acc, lane); break;
            case 3: rnn_matvec_any<J, 3, kWinLds>(Wt, ctx, acc, lane); break;
            default: rnn_matvec_any<J, 4, kWinLds>(Wt, ctx, acc, lane); break;
          }
#pragma unroll
          for (int p = 0; p < kRnnCN; ++p) {
            if (p < cn) {
              float* op = rn_ctx + (u64)ctx_row(b * (u32)G + c0 + (u32)p) * EP + (u32)lane * J;
#pragma unroll
              for (int j = 0; j < J; ++j) {
                u32 i = (u32)lane * J + j;
                float x = acc[p][j] + embv[p][j];
                op[j] = i < E ? sigmoid_ref(x, s_exptab) : 0.f;
              }
            }
          }
        }
      }
      JPP_RPROF(3);
      wave_sync();
      if (lane < ngb) {
        u32 c = conn[(u64)b * G + lane];
        if (c != kNoConn) {
          u32 nd = c & 0x03ffffffu, k = c >> 26;
          u32 gi = (nd == N - 1) ? k : beams[(u64)nd * beam + k].pad;
          B.node_cells[((nb + nd) * G + gi) * S + 1] = nscore[assign[(u64)b * G + lane]];
        }
      }
      wave_sync();
    }

  }
  JPP_RPROF(4);
  // ---- D. adjustBeamScores along the EOS paths ----
  const u32 efirstE = B.end_first[bb0 + bE];
  if (lane < ngb && !inLds) {
    for (u32 b = 2; b < bE; ++b) {
      u32 c = conn[(u64)b * G + lane];
      if (c == kNoConn) continue;
      u32 nd = c & 0x03ffffffu, k = c >> 26;
      BeamSlot* sl = &beams[(u64)nd * beam + k];
      const float* cell = B.node_cells + ((nb + nd) * G + sl->pad) * S;
      const float local = weighted_score2(cell[0], cell[1], cfg) + prevT;
      sl->total = local;
      prevT = local;
    }
    // ---- E. remakeEosBeam ----
    const float* cell = B.node_cells + ((nb + N - 1) * G + lane) * S;
    full[lane] = weighted_score2(cell[0], cell[1], cfg) + prevT;
    prev_total[lane] = prevT;
  }
  wave_sync();
  rnn_remake_eos<SORT>(B, beams, en, efirstE, bb0, bE, N, G, beam, ngb, full, prev_total, lane);
  JPP_RPROF(5);
  JPP_RPROF_FLUSH;
  wave_sync();
  slot += gridDim.x * kWaves;
  } while (MODE == 3 && slot < nslow);
}

// Scores, score cells, adjustBeamScores and remakeEosBeam of the sentences beyond k_rnn_score<.., 2>'s LDS staging
// (long sentences, wide global beams), after k_rnn_chain has left every hidden state in HBM (round 5; until then such a
// sentence ran k_rnn_score<.., 3>: recurrence and scores boundary by boundary, ~17 us of dependent round trips per
// boundary -- 7.7 ms per batch of the configs[4] shape).  Same arithmetic as MODE 2, the rnn lattice read from the row
// records (k_rnn_dense):
//   per ROW (lane = row): maxent sum (w0 + w1 + ...), NCE dot product (rnn_dot_seq), - nceConstant; UNK: one fused
//     multiply-add                                                        -> rnn_rscore[row]
//   per (boundary, path) slot: the connection's score cell               <- rnn_rscore[noff[b] + assign]
//   per path (lane = path): adjustBeamScores front to back, remakeEosBeam inputs
// One wavefront per sentence, a fixed grid walking the last class of the chain-length order (the sentences that are
// not staged).
template <int J, bool SORT>
__global__ void __launch_bounds__(256) k_rnn_score_long(Batch B, const DevModel* __restrict__ Mp, Config cfg) {
  constexpr int kWaves = 4;
  constexpr int EP = 64 * J;
  const DevModel& M = *Mp;
  const int wv = (int)(threadIdx.x >> 6);
  const int lane = (int)(threadIdx.x & 63);
  __shared__ float full_all[kWaves][kMaxGbeam];
  __shared__ float prev_total_all[kWaves][kMaxGbeam];
  float* full = full_all[wv];
  float* prev_total = prev_total_all[wv];
  if (B.gstats[kGstatOverflow] != 0) return;
  u32 slot = blockIdx.x * kWaves + wv;
  const u32 nslow = B.rnn_slow[1];
  for (; slot < nslow; slot += gridDim.x * kWaves) {
    const u32 s = B.rnn_order[B.rnn_slow[0] + slot];
    if (s >= B.n_sent || B.sent_status[s] != ST_OK) continue;
    const u32 n = B.sent_ncp[s];
    if (n == 0) continue;
    const u32 bb0 = B.byte_off[s] + 4 * s;
    const u32 bE = n + 2;
    const int ngb = (int)B.bnd_ngb[bb0 + bE];
    if (ngb == 0) continue;
    const u32 N = B.sent_nodes[s];
    const u64 nb = B.node_base[s];
    const int beam = cfg.beam, G = cfg.gbeam, S = cfg.nscorers;
    const u32 E = M.rnn_E;
    const float JPP_GLOBAL* __restrict__ nceT = as_global(M.rnn_nce);
    const float JPP_GLOBAL* __restrict__ maxentT = as_global(M.rnn_maxent);
    const u32 mxOrder = M.rnn_order;
    const u64 hashMax = M.rnn_hash_max, hashMagic = M.rnn_hash_magic, mxBase = M.rnn_mx_base;
    const u64 mxCoef[4] = {M.rnn_mx_coef[0], M.rnn_mx_coef[1], M.rnn_mx_coef[2], M.rnn_mx_coef[3]};
    const float nceConst = M.rnn_nce_const, unkConst = M.rnn_unk_const, unkLen = M.rnn_unk_len;
    const i32 unkId = M.rnn_unk_id;
    BeamSlot* beams = B.node_beam + nb * beam;
    const u32* en = B.end_nodes + nb;
    const u32* conn = B.rnn_conn + (u64)bb0 * G;
    const u32* assign = B.rnn_assign + (u64)bb0 * G;
    const u32* g_gi = B.rnn_gi + (u64)bb0 * G;
    const u32* noff = B.rnn_noff + bb0;
    const u64 rowBase = B.rnn_rowbase[s];
    const RnnRec* recs = B.rnn_rec + rowBase;
    float* rsc = B.rnn_rscore + rowBase;
    const float* rn_ctx = B.rnn_ctx + rowBase * (u64)EP;
    const u32 rows = B.rnn_rows[s];
    const u32 nq = (bE + 1) * (u32)G;
    // ---- scores: one lane per row (rows 2 .. rows - 1 are the rnn nodes, the EOS boundary's included) ----
    for (u32 j0 = 2; j0 < rows; j0 += 64) {
      const u32 j = j0 + (u32)lane;
      if (j < rows) {
        const RnnRec rc = recs[j];
        float score;
        if (rc.id == unkId) {
          score = __builtin_fmaf(unkLen, (float)rc.len, unkConst);
        } else if (mxOrder == 0) {
          score = 0.f - nceConst;  // MikolovScoreCalculator::addScores case 0 zero-fills the result
        } else {
          float mw[4] = {0.f, 0.f, 0.f, 0.f};
          rnn_maxent_gather(maxentT, rc.id, recs[rc.prevrow].id, mxOrder, mxBase, mxCoef, hashMax, hashMagic, mw);
          float me = mw[0];   // calcScoresN: w0 + w1 + ... left to right
#pragma unroll
          for (u32 i = 1; i < 4; ++i)
            if (i < mxOrder) me += mw[i];
          const u32 eid = rc.id == -1 ? 0u : (u32)rc.id;
          score = rnn_dot_seq(nceT + (u64)eid * E, rn_ctx + (u64)rc.prevrow * EP, E);
          score += me;
          score -= nceConst;
        }
        rsc[j] = score;
      }
    }
    wave_sync();   // (the scores are read by other lanes below: written and read by this wavefront only)
#if !defined(JPP_EMU)
    __threadfence_block();
#endif
    // Both loops below are chains of three dependent loads per step (connection -> its rnn node / cell index -> the
    // score); four steps are loaded level by level -- unconditionally, from addresses that are always inside the
    // sentence's arrays -- before any of them is used, so a round trip is paid once per four.
    constexpr int kU = 4;
    const u32 invG = small_div_inv((u32)G);   // (q < 2048 * ... does not hold here: plain division below for the boundary of a slot)
    (void)invG;
    // ---- score cells of the connections (one lane per (boundary, path) slot) ----
    for (u32 q0 = 2u * (u32)G + (u32)lane; q0 < nq; q0 += 64u * kU) {
      u32 c[kU], qq[kU];
#pragma unroll
      for (int u = 0; u < kU; ++u) {
        const u32 q = q0 + 64u * (u32)u;
        qq[u] = q < nq ? q : (u32)lane;   // (slot `lane` of boundary 0 / 1: allocated, never a connection's)
        c[u] = q < nq ? conn[q] : kNoConn;
      }
      u32 gi[kU], row[kU];
#pragma unroll
      for (int u = 0; u < kU; ++u) {
        gi[u] = g_gi[qq[u]] & 0xffffu;
        row[u] = noff[qq[u] / (u32)G] + (assign[qq[u]] & 0xffffu);
      }
#pragma unroll
      for (int u = 0; u < kU; ++u) {
        if (c[u] == kNoConn) continue;
        const u32 nd = c[u] & 0x03ffffffu;
        B.node_cells[((nb + nd) * G + gi[u]) * S + 1] = rsc[row[u]];
      }
    }
    // ---- ScoreProcessor::adjustBeamScores along the EOS paths (lane = path), remakeEosBeam inputs ----
    if (lane < ngb) {
      float prevT = 0.f;   // BOS element total = 0
      for (u32 b0 = 2; b0 <= bE; b0 += kU) {
        u32 c[kU], qq[kU], bb[kU];
#pragma unroll
        for (int u = 0; u < kU; ++u) {
          bb[u] = b0 + (u32)u;
          qq[u] = (bb[u] <= bE ? bb[u] : bE) * (u32)G + (u32)lane;
          c[u] = bb[u] <= bE ? conn[qq[u]] : kNoConn;
        }
        u32 gi[kU], row[kU];
#pragma unroll
        for (int u = 0; u < kU; ++u) {
          gi[u] = g_gi[qq[u]] & 0xffffu;
          row[u] = noff[bb[u] <= bE ? bb[u] : bE] + (assign[qq[u]] & 0xffffu);
        }
        float rs[kU], cell0[kU];
#pragma unroll
        for (int u = 0; u < kU; ++u) {
          const bool live = c[u] != kNoConn;
          const u32 nd = live ? (c[u] & 0x03ffffffu) : 0u;
          rs[u] = live ? rsc[row[u]] : 0.f;
          cell0[u] = live ? B.node_cells[((nb + nd) * G + gi[u]) * S] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < kU; ++u) {
          if (c[u] == kNoConn) continue;
          const u32 nd = c[u] & 0x03ffffffu, k = c[u] >> 26;
          const float local = weighted_score2(cell0[u], rs[u], cfg);
          if (bb[u] < bE) {
            const float tot = local + prevT;
            beams[(u64)nd * beam + k].total = tot;
            prevT = tot;
          } else {
            full[lane] = local + prevT;  // remakeEosBeam: fullScores[i] = localScore + beamScore
            prev_total[lane] = prevT;
          }
        }
      }
    }
    wave_sync();
    rnn_remake_eos<SORT>(B, beams, en, B.end_first[bb0 + bE], bb0, bE, N, G, beam, ngb, full, prev_total, lane);
    wave_sync();
  }
}

}  // namespace jpp

#endif  // JPP_K_RNN_H
