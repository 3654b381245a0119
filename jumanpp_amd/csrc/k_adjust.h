// Score combination for ScorerDef::others beyond the model's own RNN: after every extra scorer
// has written its slot of the score cells -- the RNN on the device (k_rnn.h, slot 1), any other ScoreComputer on the host
// through jppgpu_analyze_batch_scored -- the beam totals are re-made from the weighted cells.
//
// Reference behaviour reproduced:
//   AnalyzerImpl::computeScoresGbeam tail     src/core/analysis/analyzer_impl.cc:286-294   scoreLattice per scorer, then
//   ScoreProcessor::adjustBeamScores          score_processor.cc:521-550   totalScore = sum_i score_i * weight_i + previous total,
//                                                                          over the global-beam elements in boundary order
//   ScoreProcessor::remakeEosBeam             score_processor.cc:552-576   the EOS beam re-ranked by the adjusted totals
//   makeT0Beam                                score_processor.cc:426-469   (util::partition beyond beam*4/3, std::sort)
//
// One wavefront per sentence, lane = EOS global-beam element = one surviving path.  Only elements on a path from the EOS
// beam can reach any output, so the totals are re-made along those paths (as k_rnn_score does): a lane walks its path
// back to BOS, records it, and re-adds it front to back.  Paths that share a prefix write the same values to the shared
// slots.  The sums are fused multiply-adds from 0 in scorer order, like the reference's -march=haswell object code.
#ifndef JPP_K_ADJUST_H
#define JPP_K_ADJUST_H

#include "jpp_device.h"
#include "jpp_select.h"

namespace jpp {

constexpr int kMaxScorers = 4;   // perceptron + up to three others

struct ScoreWeights {
  float w[kMaxScorers];
};

__device__ __forceinline__ float weighted_cells(const float* cell, const ScoreWeights& W, int S) {
  float local = 0.f;
  for (int i = 0; i < S; ++i) local = __builtin_fmaf(cell[i], W.w[i], local);
  return local;
}

// stack: [total boundaries of the batch][G] u32 scratch, (node | slot << 26) per (boundary, path), like Batch::rnn_conn
template <bool SORT>
__global__ void __launch_bounds__(256) k_adjust(Batch B, Config cfg, ScoreWeights W, u32* stack) {
  const u32 s = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = (int)(threadIdx.x & 63);
  const int wv = (int)(threadIdx.x >> 6);
  if (s >= B.n_sent) return;
  if (B.sent_status[s] != ST_OK) return;
  const u32 n = B.sent_ncp[s];
  if (n == 0) return;
  const u32 off = B.byte_off[s];
  const u32 bb0 = off + 4 * s;
  const u32 N = B.sent_nodes[s];
  const u64 nb = B.node_base[s];
  const int beam = cfg.beam, G = cfg.gbeam, S = cfg.nscorers;
  const u32 bE = n + 2;
  const int ngb = (int)B.bnd_ngb[bb0 + bE];
  if (ngb == 0) return;
  BeamSlot* beams = B.node_beam + nb * beam;
  const u32* en = B.end_nodes + nb;
  const u32 efirstE = B.end_first[bb0 + bE];
  __shared__ float full_all[4][kMaxGbeam];
  __shared__ float prev_all[4][kMaxGbeam];
  float* full = full_all[wv];
  float* prev_total = prev_all[wv];
  u32* mine = stack + (u64)bb0 * G;   // element (depth d of path p) at mine[d * G + p]; depth < n + 1
  if (lane < ngb) {
    const GbeamEntry ge = B.bnd_gbeam[(u64)(bb0 + bE) * G + lane];
    u32 node = en[efirstE + ge.left], slot = ge.beam;
    u32 depth = 0;
    while (node >= 2 && node != 0xffffffffu && depth <= n) {
      mine[(u64)depth * G + lane] = node | (slot << 26);
      const BeamSlot sl = beams[(u64)node * beam + slot];
      node = sl.prev_node;
      slot = sl.beam;
      ++depth;
    }
    float prevT = 0.f;   // the BOS element's total
    for (u32 d = depth; d-- > 0;) {
      const u32 c = mine[(u64)d * G + lane];
      const u32 nd = c & 0x03ffffffu, k = c >> 26;
      BeamSlot* sl = &beams[(u64)nd * beam + k];
      const float* cell = B.node_cells + ((nb + nd) * G + sl->pad) * S;
      const float local = weighted_cells(cell, W, S) + prevT;
      sl->total = local;
      prevT = local;
    }
    const float* cell = B.node_cells + ((nb + N - 1) * G + lane) * S;
    full[lane] = weighted_cells(cell, W, S) + prevT;   // remakeEosBeam: fullScores[i] = localScore + beamScore
    prev_total[lane] = prevT;
  }
  wave_sync();
  BeamSlot* row = beams + (u64)(N - 1) * beam;
  const int partB = beam * 4 / 3;
  // makeT0Beam on the EOS candidates (see k_sweep 5c / k_rnn_score): a stable rank unless two totals tie exactly under a
  // configuration that partitions or introsorts
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
          const GbeamEntry ge = B.bnd_gbeam[(u64)(bb0 + bE) * G + idx[z]];
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
      const float me = full[lane];
      int rank = 0;
      for (int j = 0; j < ngb; ++j) {
        const float o = full[j];
        if (o > me || (o == me && j < lane)) ++rank;
      }
      const GbeamEntry ge = B.bnd_gbeam[(u64)(bb0 + bE) * G + lane];
      if (rank < beam) row[rank] = BeamSlot{ge.left, ge.beam, me, en[efirstE + ge.left], (u32)lane};
      B.bnd_gbeam[(u64)(bb0 + bE) * G + lane].score = prev_total[lane];
    } else if (lane < beam) {
      row[lane] = BeamSlot{kFake16, kFake16, 0.f, 0xffffffffu, 0};
    }
  }
}

}  // namespace jpp

#endif  // JPP_K_ADJUST_H
