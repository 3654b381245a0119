// Kernels 3: lattice layout: short passes over ~C positions / ~N nodes, one wavefront per sentence
// (lane per position / per boundary, prefix sums by wave scan).
//
// Reference behaviour reproduced:
//   LatticeBuilder::prepare            src/core/analysis/lattice_builder.cc:16-39
//   LatticeBuilder::checkConnectability src/core/analysis/lattice_builder.cc:41-52
//   AnalyzerImpl::prepareNodeSeeds     src/core/analysis/analyzer_impl.cc:128-139
//   LatticeBuilder::makeBos/makeEos/fillEnds  lattice_builder.cc:95-145
#ifndef JPP_K_LATTICE_H
#define JPP_K_LATTICE_H

#include <type_traits>

#include "jpp_device.h"

namespace jpp {

constexpr int kLatWaves = 4;  // sentences (wavefronts) per workgroup of the wave-per-sentence kernels below

// inclusive prefix sum over the 64 lanes
__device__ __forceinline__ u32 wave_scan_incl_u32(u32 v, int lane) {
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    u32 o = wave_shfl_u32(v, lane >= d ? lane - d : lane);
    if (lane >= d) v += o;
  }
  return v;
}

// STAGE 1: layout with dic + stage-1 makers.  STAGE 2: (flagged sentences only) all makers.
// One wavefront per sentence, one lane per start position.
template <int STAGE>
__global__ void __launch_bounds__(64 * kLatWaves) k_layout(Batch B) {
  const int lane = (int)(threadIdx.x & 63);
  const u32 s = blockIdx.x * kLatWaves + (threadIdx.x >> 6);
  if (s >= B.n_sent) return;
  if (STAGE == 2 && (B.sent_status[s] != ST_OK || (B.sent_flags[s] & 2) == 0)) {
    if (lane == 0) B.sent_nodes2[s] = 0;
    return;
  }
  const u32 off = B.byte_off[s];
  const u32 g0 = off + s;
  const u32 bb0 = off + 4 * s;
  const u32 n = B.sent_status[s] == ST_OK ? B.sent_ncp[s] : 0;
  // local node ids: 0,1 = BOS; boundary b = i + 2 starts at position i
  if (lane < 2) {
    B.bnd_first[bb0 + lane] = (u32)lane;
    B.bnd_cnt[bb0 + lane] = 1;
  }
  u32 next = 2;
  u32 sum2 = 0;
  u32 maxR = 0;
  bool overflow = false;
  for (u32 i0 = 0; i0 < n; i0 += 64) {
    const u32 i = i0 + (u32)lane;
    u32 c = 0, c2 = 0;
    bool ovf = false;
    if (i < n) {
      u32 c1 = B.pos_cnt1[g0 + i];
      c = c1 + B.pos_cntN[g0 + i];
      ovf = c1 == 0xffff;
      c2 = B.pos_cnt2[g0 + i];
      if (STAGE == 2) c += c2;
    }
    const u32 incl = wave_scan_incl_u32(c, lane);
    if (i < n) {
      B.bnd_first[bb0 + i + 2] = next + incl - c;
      B.bnd_cnt[bb0 + i + 2] = c;
    }
    next += wave_shfl_u32(incl, 63);
    sum2 += wave_sum_u32(c2);
    const u32 m = wave_max_u32(c);
    if (m > maxR) maxR = m;
    overflow = overflow || wave_ballot(ovf) != 0;
  }
  if (STAGE == 1 && n > 0 && n <= 63 && B.sent_status[s] == ST_OK) {
    // does the end of input connect through dictionary + stage-1 nodes?  Their ends are known from the
    // count pass, so a sentence that needs the stage-2 makers is flagged before anything is emitted and
    // its nodes are written once (k_seeds<2>) instead of twice.  Longer sentences: k_connect<1>.
    const u64 mask = (u32)lane < n ? B.pos_ends[g0 + lane] : u64{0};
    u64 reach = 1;
    for (u32 i = 0; i < n; ++i) {
      const u64 m = wave_shfl_u64(mask, (int)i);
      if ((reach >> i) & 1) reach |= m;
    }
    if (((reach >> n) & 1) == 0 && lane == 0) B.sent_flags[s] |= 2;
  }
  if (lane != 0) return;
  // one atomic per sentence would serialise 64k updates of one address: skip it when the published
  // maximum (monotonic, possibly stale) already covers this sentence
  if (maxR > B.gstats[0]) atomicMax(&B.gstats[0], maxR);
  B.sent_maxr[s] = maxR;
  // EOS boundary
  B.bnd_first[bb0 + n + 2] = next;
  B.bnd_cnt[bb0 + n + 2] = 1;
  next += 1;
  if (overflow) B.sent_status[s] = ST_CAPACITY;
  if (STAGE == 1) {
    B.sent_nodes[s] = next;
    B.sent_nodes2[s] = next + sum2;  // upper bound used to size the relocation area
  } else {
    B.sent_nodes2[s] = next;
  }
}

// single-workgroup exclusive scan of u32 counts into u64 offsets (+ base); out[n] receives the total.
// n is at most a few hundred thousand.  Tiles of 8 elements per thread go through LDS so that HBM is
// read and written coalesced while every thread still scans a contiguous run.
constexpr u32 kScanPer = 8;
__global__ void __launch_bounds__(1024) k_scan(const u32* in, u64* out, u32 n, const u64* base_ptr) {
  __shared__ u32 tile[1024 * kScanPer];
  __shared__ u64 otile[1024 * kScanPer];
  __shared__ u64 part[16];
  const u32 t = threadIdx.x;
  const u32 nt = blockDim.x;
  const int lane = (int)(t & 63);
  const u32 tileN = nt * kScanPer;
  u64 carry = base_ptr ? *base_ptr : 0;
  for (u32 t0 = 0; t0 < n; t0 += tileN) {
#pragma unroll
    for (u32 q = 0; q < kScanPer; ++q) {
      u32 i = t0 + q * nt + t;
      tile[q * nt + t] = i < n ? in[i] : 0u;
    }
    __syncthreads();
    u32 v[kScanPer];
    u64 sum = 0;
#pragma unroll
    for (u32 q = 0; q < kScanPer; ++q) {
      v[q] = tile[t * kScanPer + q];
      sum += v[q];
    }
    // exclusive scan of the per-thread sums: wave scan + scan of the (<= 16) wave totals
    u64 incl = sum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      u64 o = wave_shfl_u64(incl, lane >= d ? lane - d : lane);
      if (lane >= d) incl += o;
    }
    if (lane == 63) part[t >> 6] = incl;
    __syncthreads();
    u64 wbase = carry, all = carry;
    const u32 nw = (nt + 63) / 64;
    for (u32 k = 0; k < nw; ++k) {
      u64 pv = part[k];
      if (k < (t >> 6)) wbase += pv;
      all += pv;
    }
    u64 acc = wbase + incl - sum;
#pragma unroll
    for (u32 q = 0; q < kScanPer; ++q) {
      otile[t * kScanPer + q] = acc;
      acc += v[q];
    }
    __syncthreads();
#pragma unroll
    for (u32 q = 0; q < kScanPer; ++q) {
      u32 i = t0 + q * nt + t;
      if (i < n) out[i] = otile[q * nt + t];
    }
    carry = all;
    __syncthreads();
  }
  if (t == 0) out[n] = carry;
}

// The same scan over several workgroups (round 5): the single-workgroup form walks a 65 536-sentence batch in eight
// tiles, three barriers each -- 44 us, five times per batch.  Here every workgroup takes ONE tile: it publishes the
// tile's sum (value, then an epoch-stamped flag), reads the sums of the workgroups before it as they appear (at most
// kScanMbBlocks - 1 of them: one lane each, no chain), and scans its tile from that base.  Workgroups are dispatched in
// index order and all fit the chip at once, so the wait cannot deadlock; the epoch (a counter of the context) makes a
// flag of an earlier launch meaningless, nothing is reset.  n up to kScanMbBlocks tiles; the host falls back to k_scan
// beyond that.
constexpr u32 kScanMbBlocks = 64;
struct ScanWs {
  u64 part[kScanMbBlocks];
  u32 flag[kScanMbBlocks];
};
__global__ void __launch_bounds__(1024) k_scan_mb(const u32* in, u64* out, u32 n, const u64* base_ptr, ScanWs* ws, u32 epoch) {
  __shared__ u32 tile[1024 * kScanPer];
  __shared__ u64 otile[1024 * kScanPer];
  __shared__ u64 part[16];
  __shared__ u64 s_carry;
  const u32 t = threadIdx.x;
  const int lane = (int)(t & 63);
  const u32 b = blockIdx.x;
  const u32 t0 = b * 1024u * kScanPer;
#pragma unroll
  for (u32 q = 0; q < kScanPer; ++q) {
    const u32 i = t0 + q * 1024u + t;
    tile[q * 1024u + t] = i < n ? in[i] : 0u;
  }
  __syncthreads();
  u32 v[kScanPer];
  u64 sum = 0;
#pragma unroll
  for (u32 q = 0; q < kScanPer; ++q) {
    v[q] = tile[t * kScanPer + q];
    sum += v[q];
  }
  u64 incl = sum;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const u64 o = wave_shfl_u64(incl, lane >= d ? lane - d : lane);
    if (lane >= d) incl += o;
  }
  if (lane == 63) part[t >> 6] = incl;
  __syncthreads();
  u64 wbase = 0, all = 0;
  for (u32 k = 0; k < 16; ++k) {
    const u64 pv = part[k];
    if (k < (t >> 6)) wbase += pv;
    all += pv;
  }
  // publish this tile's sum, then collect the tiles before it
  if (t == 0) {
    ws->part[b] = all;
    __threadfence();
    atomicExch(&ws->flag[b], epoch);
  }
  if (t < 64) {
    u64 mine = 0;
    if (t < b) {
      volatile u32* f = ws->flag + t;
      while (*f != epoch) {
      }
      __threadfence();
      mine = *(volatile u64*)(ws->part + t);
    }
    const u64 before = wave_sum_u64(mine);
    if (t == 0) s_carry = before + (base_ptr ? *base_ptr : 0);
  }
  __syncthreads();
  const u64 carry = s_carry;
  u64 acc = carry + wbase + incl - sum;
#pragma unroll
  for (u32 q = 0; q < kScanPer; ++q) {
    otile[t * kScanPer + q] = acc;
    acc += v[q];
  }
  __syncthreads();
#pragma unroll
  for (u32 q = 0; q < kScanPer; ++q) {
    const u32 i = t0 + q * 1024u + t;
    if (i < n) out[i] = otile[q * 1024u + t];
  }
  if (t == 0 && b == gridDim.x - 1) out[n] = carry + all;
}

// Sentences are routed to the sweep variant that fits THEIR widest boundary, not the batch's (one sentence with a
// 100-homograph boundary must not move the other 65 535 to the generic kernel): class 0 = at most t0 right nodes at
// every boundary (the variants that stage 64 in LDS), class 1 = at most t1 (LDS staging of kMaxRight), class 2 =
// wider (per-right-node arrays in an HBM scratch slice; the reference has no limit, lattice_builder.cc:70-93).
// Counting sort of the sentence indices by class: sweep_list[c * n + k], gstats[1 + c] = sentences of class c.
// One thread per sentence, one atomic per wavefront and class.  Failed sentences are in no list.
__global__ void __launch_bounds__(256) k_sweep_classify(Batch B, u32 t0, u32 t1) {
  const u32 s = blockIdx.x * blockDim.x + threadIdx.x;
  const int lane = (int)(threadIdx.x & 63);
  const bool act = s < B.n_sent && B.sent_status[s] == ST_OK;
  u32 cls = 3;
  if (act) {
    const u32 m = B.sent_maxr[s];
    cls = m <= t0 ? 0u : m <= t1 ? 1u : 2u;
    if (m > 0xffffu) {   // the cutoff order of the sweep is u16 (node spans are u16 as well: not reachable with real dictionaries)
      B.sent_status[s] = ST_CAPACITY;
      cls = 3;
    }
  }
  for (u32 c = 0; c < 3; ++c) {
    const u64 bal = wave_ballot(cls == c);
    if (bal == 0) continue;
    const int leader = __builtin_ctzll(bal);
    u32 base = 0;
    if (lane == leader) base = atomicAdd(&B.gstats[1 + c], (u32)popc64(bal));
    base = wave_shfl_u32(base, leader);
    if (cls == c) B.sweep_list[(u64)c * B.n_sent + base + (u32)popc64(bal & ((u64{1} << lane) - 1))] = s;
  }
}

// relocate flagged sentences: node_base[s] = node_base2[s]
__global__ void k_relocate(Batch B) {
  u32 s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= B.n_sent) return;
  if (B.sent_status[s] == ST_OK && (B.sent_flags[s] & 2)) {
    B.node_base[s] = B.node_base2[s];
    B.sent_nodes[s] = B.sent_nodes2[s];
  }
}

// reachability of the end of input through the emitted nodes.
// PASS 1: sets the stage-2 flag; PASS 2: (flagged only) final verdict.
// One wavefront per sentence.  Up to 63 codepoints the reachable set is one u64: lane i builds the mask
// of the ends of the nodes starting at i, then the positions are swept in order.  Longer sentences run the
// plain sequential pass on lane 0.
template <int PASS>
__global__ void __launch_bounds__(64 * kLatWaves) k_connect(Batch B) {
  const int lane = (int)(threadIdx.x & 63);
  const u32 s = blockIdx.x * kLatWaves + (threadIdx.x >> 6);
  if (s >= B.n_sent) return;
  if (B.sent_status[s] != ST_OK) return;
  if (PASS == 2 && (B.sent_flags[s] & 2) == 0) return;
  if (PASS == 1 && (B.sent_flags[s] & 2) != 0) return;  // flagged by k_layout<1>: nothing was emitted for it yet
  const u32 off = B.byte_off[s];
  const u32 g0 = off + s;
  const u32 bb0 = off + 4 * s;
  const u32 n = B.sent_ncp[s];
  const NodeInfo* ni = B.node_info + B.node_base[s];
  bool ok;
  if (n <= 63) {
    u64 mask = 0;
    if ((u32)lane < n) {
      const u32 first = B.bnd_first[bb0 + lane + 2];
      const u32 cnt = B.bnd_cnt[bb0 + lane + 2];
      for (u32 k = 0; k < cnt; k += 4) {
        u16 e[4];
#pragma unroll
        for (u32 q = 0; q < 4; ++q) e[q] = (k + q < cnt) ? ni[first + k + q].end : (u16)0;
#pragma unroll
        for (u32 q = 0; q < 4; ++q)
          if (k + q < cnt) mask |= u64{1} << e[q];
      }
    }
    u64 reach = 1;
    for (u32 i = 0; i < n; ++i) {
      u64 m = wave_shfl_u64(mask, (int)i);
      if ((reach >> i) & 1) reach |= m;
    }
    ok = ((reach >> n) & 1) != 0;
  } else {
    // Longer sentences (round 4; until then one lane walked every node of the sentence, a dependent HBM read each:
    // 3 ms per batch of 220-codepoint sentences).  64 start positions at a time: lane i builds the mask of the ends of
    // its position's nodes RELATIVE to the position (bit d: a node of d codepoints), then the window `win` -- bit d:
    // position p + d is reachable -- slides over the 64 positions with one broadcast per position.  A node longer than
    // 63 codepoints (a long run of digits or letters) sends the sentence to the sequential pass below.
    u64 win = 1;   // position 0 is reachable
    bool far = false;
    for (u32 p0 = 0; p0 < n && !far; p0 += 64) {
      const u32 p = p0 + (u32)lane;
      u64 rel = 0;
      bool fr = false;
      if (p < n) {
        const u32 first = B.bnd_first[bb0 + p + 2];
        const u32 cnt = B.bnd_cnt[bb0 + p + 2];
        for (u32 k = 0; k < cnt; k += 4) {
          u16 e[4];
#pragma unroll
          for (u32 q = 0; q < 4; ++q) e[q] = (k + q < cnt) ? ni[first + k + q].end : (u16)0;
#pragma unroll
          for (u32 q = 0; q < 4; ++q) {
            if (k + q >= cnt) continue;
            const u32 d = (u32)e[q] - p;
            if (d <= 63) rel |= u64{1} << d;
            else fr = true;
          }
        }
      }
      far = wave_ballot(fr) != 0;
      if (far) break;
      const u32 m = n - p0 < 64u ? n - p0 : 64u;
      for (u32 i = 0; i < m; ++i) {
        const u64 r = wave_shfl_u64(rel, (int)i);
        if (win & 1) win |= r;
        win >>= 1;
      }
    }
    ok = (win & 1) != 0;   // bit 0 after n steps: the end of input
    if (far) {
      if (lane != 0) return;
      u8* reach = B.reach + g0;  // n + 1 entries
      for (u32 i = 0; i <= n; ++i) reach[i] = 0;
      reach[0] = 1;
      for (u32 i = 0; i < n; ++i) {
        if (!reach[i]) continue;
        u32 first = B.bnd_first[bb0 + i + 2];
        u32 cnt = B.bnd_cnt[bb0 + i + 2];
        for (u32 k = 0; k < cnt; ++k) reach[ni[first + k].end] = 1;
      }
      ok = reach[n] != 0;
    }
  }
  if (!ok && lane == 0) {
    if (PASS == 1) B.sent_flags[s] |= 2;
    else B.sent_status[s] = ST_NO_LATTICE;
  }
}

// BOS/EOS nodes, UNK entry pointers, ends lists.  One wavefront per sentence: the node ends are staged
// in LDS, then one lane per boundary collects the nodes ending there in node order (= seed order,
// LatticeBuilder::fillEnds).  Sentences with more than kEndsNodeCap nodes read the ends from the node table instead
// (until round 5 they ran sequentially on lane 0: one such sentence in a batch of 220-codepoint sentences, 1 454 nodes on
// average, held the launch for 1.1 ms instead of 0.36, profiles/r05_y_bench.json config5.kernel_ms_per_step.layout).
constexpr u32 kEndsNodeCap = 2048;

// UNK entry pointers are numbered in CREATION order, like the reference's: ExtraNodesContext::allocateExtra
// gives every extra node the next index (extra_nodes.cc:41-52), and the nodes are made maker by maker
// (AnalyzerImpl::makeUnkNodes1/2, analyzer_impl.cc:100-126: stage 1 in spec order, then stage 2), each maker
// walking the start positions in order, one extra node per seed.  Node order here is (start, seed order)
// (the stable sort of LatticeBuilder::sortSeeds), so the creation index of an UNK node is
//   #(UNK nodes of makers created earlier) + #(UNK nodes of its own maker before it in node order).
// rank[spec index] = position of the maker in that creation sequence.
struct UnkRank {
  u8 rank[kMaxUnkMakers];
  u32 n;
};

__global__ void __launch_bounds__(64 * kLatWaves) k_ends(Batch B, Config cfg, UnkRank rk) {
  const int lane = (int)(threadIdx.x & 63);
  const int wv = (int)(threadIdx.x >> 6);
  const u32 s = blockIdx.x * kLatWaves + wv;
  if (s >= B.n_sent) return;
  if (B.sent_status[s] != ST_OK) return;
  const u32 off = B.byte_off[s];
  const u32 bb0 = off + 4 * s;
  const u32 n = B.sent_ncp[s];
  const u32 N = B.sent_nodes[s];
  const u64 nb = B.node_base[s];
  NodeInfo* ni = B.node_info + nb;
  NodeAux* na = B.node_aux + nb;
  u32* ecnt = B.end_cnt + bb0;
  u32* efirst = B.end_first + bb0;
  u32* en = B.end_nodes + nb;
  __shared__ u16 l_end_all[kLatWaves][kEndsNodeCap];
  u16* l_end = l_end_all[wv];
  if (lane == 0) {
    ni[0] = NodeInfo{kEptrBOS, 0, 0};
    ni[1] = NodeInfo{kEptrBOS, 0, 0};
    ni[N - 1] = NodeInfo{kEptrEOS, (u16)n, (u16)n};
    na[0] = na[1] = na[N - 1] = NodeAux{0, 0, 0, 0, 0, 0};
  }
  const bool staged = N <= kEndsNodeCap;
  {
    // UNK entry pointers ~0, ~1, ... in creation order; stage the ends.  k_seeds left -(1 + rank of the maker) in the
    // entry pointer of every UNK node.  First the number of UNK nodes per maker (wave-uniform counters), ...
    u32 ubase[kMaxUnkMakers];
    u32 maxLen = 1;   // codepoints of the sentence's longest node
#pragma unroll
    for (int c = 0; c < kMaxUnkMakers; ++c) ubase[c] = 0;
    for (u32 k0 = 2; k0 + 1 < N; k0 += 64) {
      const u32 k = k0 + (u32)lane;
      const bool act = k + 1 < N;
      const i32 ep = act ? ni[k].eptr : 0;
      const u32 cls = ep < 0 ? (u32)(-1 - ep) : 0xffu;
      if (wave_ballot(ep < 0) == 0) continue;
#pragma unroll
      for (int c = 0; c < kMaxUnkMakers; ++c)
        if ((u32)c < rk.n) ubase[c] += (u32)popc64(wave_ballot(cls == (u32)c));
    }
    // ... their exclusive prefix = first index of every maker, ...
    {
      u32 acc = 0;
#pragma unroll
      for (int c = 0; c < kMaxUnkMakers; ++c) {
        const u32 v = ubase[c];
        ubase[c] = acc;
        acc += v;
      }
    }
    // ... then every UNK node takes the next index of its maker in node order
    for (u32 k0 = 2; k0 + 1 < N; k0 += 64) {
      const u32 k = k0 + (u32)lane;
      const bool act = k + 1 < N;
      NodeInfo x = act ? ni[k] : NodeInfo{0, 0, 0};
      const bool isUnk = act && x.eptr < 0;
      if (wave_ballot(isUnk) != 0) {
        const u32 cls = isUnk ? (u32)(-1 - x.eptr) : 0xffu;
        u32 mine = 0;
#pragma unroll
        for (int c = 0; c < kMaxUnkMakers; ++c) {
          if ((u32)c < rk.n) {
            const u64 bal = wave_ballot(cls == (u32)c);
            if (cls == (u32)c) mine = ubase[c] + (u32)popc64(bal & ((u64{1} << lane) - 1));
            ubase[c] += (u32)popc64(bal);
          }
        }
        if (isUnk) {
          x.eptr = ~(i32)mine;
          ni[k] = x;
        }
      }
      if (act && staged) l_end[k] = x.end;
      const u32 len = act ? (u32)x.end - (u32)x.start : 0u;
      const u32 ml = wave_max_u32(len);
      if (ml > maxLen) maxLen = ml;
    }
    wave_sync();
    // lane per boundary: count, scan, fill.  The nodes are in start order and none is longer than maxLen codepoints:
    // the nodes that end at position `want` are among those that start at want - maxLen .. want - 1, a few dozen
    // (round 4; until then every lane went over all nodes of the sentence, twice: 2 ms per batch of 220-codepoint
    // sentences).
    // (two copies of the loop, chosen once per sentence: a per-node choice between the LDS copy and the node table costs
    // the common case 0.15 ms per batch -- measured, r05x)
    auto fill = [&](auto stagedTag) {
    constexpr bool kStaged = decltype(stagedTag)::value;
    auto end_of = [&](u32 k) -> u32 { return kStaged ? (u32)l_end[k] : (u32)ni[k].end; };
    u32 carry = 0;
    for (u32 b0 = 0; b0 <= n + 2; b0 += 64) {
      const u32 b = b0 + (u32)lane;
      const bool act = b <= n + 2;
      u32 cnt = (act && (b == 1 || b == 2)) ? 1u : 0u;  // the two BOS nodes end at boundaries 1 and 2
      u32 klo = 2, khi = 2;
      if (act && b >= 2) {
        const u32 want = b - 2;
        klo = B.bnd_first[bb0 + (b - 2 > maxLen ? b - maxLen : 2u)];
        khi = B.bnd_first[bb0 + b];   // (nodes that start at `want` end later)
        for (u32 k = klo; k < khi; ++k) cnt += (end_of(k) == want) ? 1u : 0u;
      }
      const u32 incl = wave_scan_incl_u32(cnt, lane);
      const u32 first = carry + incl - cnt;
      carry += wave_shfl_u32(incl, 63);
      if (act) {
        ecnt[b] = cnt;
        efirst[b] = first;
        B.bnd_meta[bb0 + b] = BndMeta{B.bnd_first[bb0 + b], B.bnd_cnt[bb0 + b], first, cnt};
        u32 w = first;
        if (b == 1) en[w++] = 0;
        if (b == 2) en[w++] = 1;
        if (b >= 2) {
          const u32 want = b - 2;
          for (u32 k = klo; k < khi; ++k)
            if (end_of(k) == want) en[w++] = k;
        }
      }
    }
    };
    if (staged) fill(std::true_type{});
    else fill(std::false_type{});
  }
  // BOS beams (reference AnalyzerImpl::bootstrapAnalysis, analyzer_impl.cc:179-195)
  BeamSlot* bm = B.node_beam + nb * cfg.beam;
  for (int q = lane; q < 2 * cfg.beam; q += 64) {
    BeamSlot v{kFake16, kFake16, 0.f, 0xffffffffu, 0};
    if (q == 0) v = BeamSlot{0, 0, 0.f, 0xffffffffu, 0};
    if (q == cfg.beam) v = BeamSlot{0, 0, 0.f, 0u, 0};
    bm[q] = v;
  }
  // BOS patterns (LatticeConstructionContext::addBos, lattice_builder.cc:173-179)
  u64* pat = B.node_pat + nb * kPat;
  for (int q = lane; q < 2 * kPat; q += 64) pat[q] = (u64)(u32)kEptrBOS;
}

// the batch totals the host sizes the next buffers from, written straight into mapped host memory (no copy engine)
__global__ void k_mail(const u64* a, const u64* b, const u32* g, u64* out) {
  if (threadIdx.x == 0) {
    out[0] = a ? *a : 0;
    out[1] = b ? *b : 0;
  }
  if (g && threadIdx.x < 8) out[2 + threadIdx.x] = g[threadIdx.x];
}

// ---- one enqueue per batch (round 5) --------------------------------------------------------------------------------
// The reference sizes nothing ahead: its lattice grows in an arena while the sentence is analysed (analyzer_impl.cc:
// 100-195).  Here the node tables, the lattice arrays and the RNN rows of a batch are sized from totals that only the
// device knows, which used to cost three host waits per batch.  Now the pipeline is enqueued against the CAPACITY the
// context already holds (jppgpu_ctx_reserve, or what earlier batches left) and these guards compare the totals with it
// on the device: a batch that does not fit marks every sentence ST_CAPACITY -- every later kernel skips such sentences,
// so nothing is written out of bounds -- and raises gstats[8]; the host reads the flag with the batch totals once, at
// the end, and runs the batch again the exact way (with the waits, growing the buffers).  One thread per sentence.
__global__ void __launch_bounds__(256) k_cap_guard(Batch B, const u64* a, const u64* b, u64 cap) {
  const u32 s = blockIdx.x * blockDim.x + threadIdx.x;
  const u64 need = *a + (b ? *b : 0) + 8;
  // (the flag is only ever raised by a thread that sees the same totals, or by an earlier kernel)
  if (need <= cap && B.gstats[kGstatOverflow] == 0) return;
  if (s < B.n_sent && B.sent_status[s] == ST_OK) B.sent_status[s] = ST_CAPACITY;
  if (s == 0) B.gstats[kGstatOverflow] = 1;
}
// the sweep classes against the grids / the scratch slices they were given (one thread; k_cap_guard, launched behind
// it, spreads the verdict)
__global__ void k_cls_guard(Batch B, u32 grid1, u32 grid2, u32 maxr_cap, u32 slots) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const u32 c1 = B.gstats[2], c2 = B.gstats[3], maxR = B.gstats[0];
  if (c1 > grid1 || c2 > grid2 || (c2 != 0 && (maxR > maxr_cap || c2 > slots))) B.gstats[kGstatOverflow] = 1;
}
// everything the host wants to know about a batch, once: [0] stage-1 nodes, [1] nodes, [2..9] gstats[0..7],
// [10] overflow flag, [11] hidden-state rows
__global__ void k_mail_all(const u64* total1, const u64* total, const u32* g, const u64* rows, u64* out) {
  const u32 t = threadIdx.x;
  if (t == 0) {
    out[0] = *total1;
    out[1] = *total;
    out[10] = g[kGstatOverflow];
    out[11] = rows ? *rows : 0;
  }
  if (t < 8) out[2 + t] = g[t];
}

}  // namespace jpp

#endif  // JPP_K_LATTICE_H
