// Trainer hook-up: gold nodes that no maker produced are added to the seeds of a sentence before its lattice is built.
//   TrainingExampleAdapter::ensureNodes     src/core/training/gold_example.h:88-109   (appendSeed per missing gold node)
//   TrainingExampleAdapter::makeUnkTrainingNode  gold_example.cc:118-136              (zeroed UNK node, row from the example)
//   Trainer::prepare                        src/core/training/trainer.cc:13-47        (sortSeeds, checkConnectability, prepare)
//   LatticeBuilder::sortSeeds               src/core/analysis/lattice_builder.cc       (stable by start: an added seed comes
//                                                                                      after the makers' seeds of its boundary)
// The seeds of this pipeline already sit in their final (start, creation) order when the hook runs, so "append + stable
// sort" is one out-of-place copy: every node moves up by the number of extra seeds that start before it, every extra seed
// goes behind the last node of its own start.  A sentence whose makers left it disconnected is checked again afterwards
// (the gold nodes are what repairs it); a sentence that got no extra seed keeps its status.
#ifndef JPP_K_GOLD_H
#define JPP_K_GOLD_H

#include "k_lattice.h"

namespace jpp {

__device__ __forceinline__ bool gold_live(i32 status) { return status == ST_OK || status == ST_NO_LATTICE; }

// One wavefront per sentence.  ni2 / na2: the new node tables, new_base[s]: the sentence's first node in them.
__global__ void __launch_bounds__(64 * kLatWaves) k_gold_insert(Batch B, const u64* new_base, NodeInfo* ni2, NodeAux* na2, u32 gold_rank) {
  const int lane = (int)(threadIdx.x & 63);
  const u32 s = blockIdx.x * kLatWaves + (threadIdx.x >> 6);
  if (s >= B.n_sent) return;
  const bool live = gold_live(B.sent_status[s]);
  const u32 off = B.byte_off[s];
  const u32 bb0 = off + 4 * s;
  const u32 n = live ? B.sent_ncp[s] : 0;
  const u32 N = B.sent_nodes[s];
  const u64 ob = B.node_base[s];
  const u64 nb = new_base[s];
  const u32 e0 = B.gold_off[s];
  const u32 ne = live ? B.gold_off[s + 1] - e0 : 0;
  const ExtraSeed* g = B.gold + e0;
  for (u32 k = 2 + (u32)lane; k + 1 < N; k += 64) {
    const NodeInfo x = B.node_info[ob + k];
    u32 sh = 0;
    for (u32 q = 0; q < ne; ++q) sh += g[q].start < x.start ? 1u : 0u;
    ni2[nb + k + sh] = x;
    na2[nb + k + sh] = B.node_aux[ob + k];
  }
  for (u32 q = (u32)lane; q < ne; q += 64) {
    const u32 j = g[q].start;
    u32 before = 0;
    for (u32 r = 0; r < ne; ++r) before += (g[r].start < j || (g[r].start == j && r < q)) ? 1u : 0u;
    const u32 idx = B.bnd_first[bb0 + j + 2] + B.bnd_cnt[bb0 + j + 2] + before;
    ni2[nb + idx] = NodeInfo{-(i32)(1 + gold_rank), g[q].start, g[q].end};
    na2[nb + idx] = NodeAux{0, g[q].hash, 0, 0, kGoldMaker, (u16)q};
  }
}

// Second half (own launch: it rewrites the boundary table the first half reads): boundary table, node count, widest
// boundary, and the connectivity verdict of the sentences that received seeds (LatticeBuilder::checkConnectability).
__global__ void __launch_bounds__(64 * kLatWaves) k_gold_bounds(Batch B, const u64* new_base, const NodeInfo* ni2) {
  const int lane = (int)(threadIdx.x & 63);
  const u32 s = blockIdx.x * kLatWaves + (threadIdx.x >> 6);
  if (blockIdx.x == 0 && threadIdx.x < 3) B.gstats[1 + threadIdx.x] = 0;   // the sweep classes are counted again
  if (s >= B.n_sent) return;
  const bool live = gold_live(B.sent_status[s]);
  const u32 off = B.byte_off[s];
  const u32 g0 = off + s;
  const u32 bb0 = off + 4 * s;
  const u32 n = live ? B.sent_ncp[s] : 0;
  const u64 nb = new_base[s];
  const u32 e0 = B.gold_off[s];
  const u32 ne = live ? B.gold_off[s + 1] - e0 : 0;
  const ExtraSeed* g = B.gold + e0;
  u32 maxR = 0;
  for (u32 i0 = 0; i0 <= n; i0 += 64) {
    const u32 i = i0 + (u32)lane;
    u32 c = 0;
    if (i <= n) {   // i == n: the EOS boundary
      u32 before = 0, same = 0;
      for (u32 q = 0; q < ne; ++q) {
        before += g[q].start < i ? 1u : 0u;
        same += g[q].start == i ? 1u : 0u;
      }
      B.bnd_first[bb0 + i + 2] += before;
      c = B.bnd_cnt[bb0 + i + 2] + same;
      B.bnd_cnt[bb0 + i + 2] = c;
    }
    const u32 m = wave_max_u32(i < n ? c : 0u);
    if (m > maxR) maxR = m;
  }
  if (lane != 0) return;
  B.sent_nodes[s] += ne;
  B.node_base[s] = nb;
  if (ne == 0) return;
  if (maxR > B.sent_maxr[s]) B.sent_maxr[s] = maxR;
  if (maxR > B.gstats[0]) atomicMax(&B.gstats[0], maxR);
  // bnd_first / bnd_cnt of this sentence were written by other lanes of this wavefront
#if !defined(JPP_EMU)
  __threadfence();
#endif
  u8* reach = B.reach + g0;
  for (u32 i = 0; i <= n; ++i) reach[i] = 0;
  reach[0] = 1;
  for (u32 i = 0; i < n; ++i) {
    if (!reach[i]) continue;
    u32 first = B.bnd_first[bb0 + i + 2];
    const u32 cnt = B.bnd_cnt[bb0 + i + 2];
    for (u32 k = 0; k < cnt; ++k) reach[ni2[nb + first + k].end] = 1;
  }
  B.sent_status[s] = reach[n] ? ST_OK : ST_NO_LATTICE;
}

}  // namespace jpp

#endif  // JPP_K_GOLD_H
