// Kernel 5b: full-beam variant of the boundary sweep (`--global-beam 0`).
// Every (left node, live beam slot) is scored against every right node; no
// global pruning, no right-node cutoff.  One wavefront per sentence, 8 lanes per
// (candidate, right node) unit exactly like the prescore phase of k_sweep.
// Not used by the CLI defaults -- built for completeness of
// the Analyzer surface; score cells are not materialised on this path.
//
// Reference behaviour reproduced:
//   AnalyzerImpl::computeScoresFull          src/core/analysis/analyzer_impl.cc:197-248
//   ScoreProcessor::applyT1 / applyT2        src/core/analysis/score_processor.cc:136-156
//     (generated applyBiStep2: 8 round-robin sums, last row unrolled-4; applyTriStep3: 4)
//   DYN: the reference's table-driven feature objects (a spec other than the compiled-in one, and its trainer):
//     PartialNgramDynamicFeatureApply::applyBiStep2 / applyTriStep3, feature_impl_ngram_partial.h:216-273 -- every row
//     through computeUnrolled4RawPerceptron (four round-robin sums), descriptors from DevSpec
//   fillBeamCandidates / processBeamCandidates / makeBeams   score_processor.cc:165-244
#ifndef JPP_K_SWEEP_FULL_H
#define JPP_K_SWEEP_FULL_H

#include "k_sweep.h"

namespace jpp {

constexpr int kFullCand = 512;   // live (left, slot) candidates per boundary staged in LDS
constexpr int kFullChunk = 4;    // right nodes per pass

template <bool DYN = false>
__global__ void __launch_bounds__(64) k_sweep_full(Batch B, const DevModel* __restrict__ Mp, Config cfg) {
  const DevModel& M = *Mp;
  const int nBi = DYN ? M.spec->nbi : spec::kNumBi;
  const int nTri = DYN ? M.spec->ntri : spec::kNumTri;
  const u32 s = blockIdx.x;
  if (B.sent_status[s] != ST_OK) return;
  const int lane = (int)threadIdx.x;
  const u32 off = B.byte_off[s];
  const u32 bb0 = off + 4 * s;
  const u32 n = B.sent_ncp[s];
  const u64 nb = B.node_base[s];
  const int beam = cfg.beam;
  const float JPP_GLOBAL* __restrict__ W = as_global(M.weights);
  const u32 wmask = M.wmask;
  const u32* en = B.end_nodes + nb;
  BeamSlot* beams = B.node_beam + nb * beam;
  const u64* pats = B.node_pat + nb * kPat;
  const float* t0s = B.node_t0 + nb;

  __shared__ u32 c_lnode[kFullCand];   // left node of candidate
  __shared__ u32 c_pnode[kFullCand];   // its previous node (T2)
  __shared__ u32 c_lk[kFullCand];      // (left << 16) | slot
  __shared__ float c_total[kFullCand]; // left element total
  __shared__ u64 keys[kFullChunk][kFullCand];
  __shared__ u32 sh_nc;

  if (n == 0) {
    for (int q = lane; q < beam; q += 64) beams[(u64)2 * beam + q] = BeamSlot{kFake16, kFake16, 0.f, 0xffffffffu, 0};
    return;
  }
  for (u32 b = 2; b <= n + 2; ++b) {
    const u32 R = B.bnd_cnt[bb0 + b];
    if (R == 0) continue;  // makeBeams over zero nodes
    const u32 rfirst = B.bnd_first[bb0 + b];
    const u32 L = B.end_cnt[bb0 + b];
    const u32 efirst = B.end_first[bb0 + b];
    // candidates in (left asc, slot asc) order = fillBeamCandidates order
    if (lane == 0) {
      u32 nc = 0;
      for (u32 l = 0; l < L; ++l) {
        u32 lnode = en[efirst + l];
        for (int k = 0; k < beam; ++k) {
          BeamSlot sl = beams[(u64)lnode * beam + k];
          if (slot_fake(sl)) break;
          if (nc < (u32)kFullCand) {
            c_lnode[nc] = lnode;
            c_pnode[nc] = sl.prev_node;
            c_lk[nc] = (l << 16) | (u32)k;
            c_total[nc] = sl.total;
          }
          ++nc;
        }
      }
      sh_nc = nc;
    }
    __syncthreads();
    const u32 nc = sh_nc;
    if (nc > (u32)kFullCand) {
      if (lane == 0) B.sent_status[s] = ST_CAPACITY;
      return;
    }
    for (u32 t0 = 0; t0 < R; t0 += kFullChunk) {
      const u32 nx = (R - t0) < (u32)kFullChunk ? (R - t0) : (u32)kFullChunk;
      // score cells: 8 lanes per (right node x, candidate i)
      const int grp = lane >> 3, j = lane & 7;
      const u32 units = nx * nc;
      for (u32 base = 0; base < units; base += 8) {
        u32 u = base + grp;
        bool act = u < units;
        u32 x = act ? u / nc : 0, i = act ? u - x * nc : 0;
        u32 t = t0 + x;
        bool lastRow = (t == R - 1);
        int Wd = (DYN || lastRow) ? 4 : 8;
        const u64* p0 = pats + (u64)(rfirst + t) * kPat;
        const u64* t1r = pats + (u64)c_lnode[i] * kPat;
        const u64* t2r = pats + (u64)c_pnode[i] * kPat;
        float f = 0.f;
        if (act && j < Wd) {
          for (int k = j; k < nBi; k += Wd) {
            const u64 pre = DYN ? M.spec->bi_prefix[k] : kNg.bi_pre[k];
            const int i0 = DYN ? (int)(M.spec->bi_t01[k] >> 4) : (int)kNg.bi_t0[k];
            const int i1 = DYN ? (int)(M.spec->bi_t01[k] & 15) : (int)kNg.bi_t1[k];
            u32 idx = (u32)hmix(hmix(pre, p0[i0]), t1r[i1]) & wmask;
            f += W[idx];
          }
        }
        float g = 0.f;
        if (act && j < nTri) {
          const u64 pre = DYN ? M.spec->tri_prefix[j] : kNg.tri_pre[j];
          const int i0 = DYN ? (int)M.spec->tri_t[j][0] : (int)kNg.tri_t0[j];
          const int i1 = DYN ? (int)M.spec->tri_t[j][1] : (int)kNg.tri_t1[j];
          const int i2 = DYN ? (int)M.spec->tri_t[j][2] : (int)kNg.tri_t2[j];
          u32 idx = (u32)hmix(hmix(hmix(pre, p0[i0]), t1r[i1]), t2r[i2]) & wmask;
          g += W[idx];
        }
        float bsum = wave_shfl_f32(f, (grp << 3));
        float tsum = wave_shfl_f32(g, (grp << 3));
#pragma unroll
        for (int jj = 1; jj < 8; ++jj) {
          float v = wave_shfl_f32(f, (grp << 3) + jj);
          float w = wave_shfl_f32(g, (grp << 3) + jj);
          if (jj < Wd) bsum += v;
          if (jj < nTri) tsum += w;
        }
        if (act && j == 0) {
          float cell = t0s[rfirst + t];
          cell += bsum;
          cell += tsum;
          float score = c_total[i] + cell;  // leftElm.totalScore + localScore
          keys[x][i] = ((u64)f32_sortable(score) << 32) | c_lk[i];
        }
      }
      __syncthreads();
      // top `beam` of the unique keys per right node (processBeamCandidates): rank by counting
      for (u32 q = lane; q < nx * nc; q += 64) {
        u32 x = q / nc, i = q - x * nc;
        u64 me = keys[x][i];
        u32 rank = 0;
        for (u32 z = 0; z < nc; ++z) rank += keys[x][z] > me;
        if (rank < (u32)beam) {
          u32 hi = (u32)(me >> 32);
          beams[(u64)(rfirst + t0 + x) * beam + rank] =
              BeamSlot{(u16)(me >> 16), (u16)me, sortable_f32(hi), c_lnode[i], 0};
        }
      }
      for (u32 q = lane; q < nx * (u32)beam; q += 64) {
        u32 x = q / (u32)beam, z = q - x * (u32)beam;
        if (z >= nc) beams[(u64)(rfirst + t0 + x) * beam + z] = BeamSlot{kFake16, kFake16, 0.f, 0xffffffffu, 0};
      }
      __syncthreads();
    }
    if (lane == 0) B.bnd_ngb[bb0 + b] = 0;
    __syncthreads();
  }
}

}  // namespace jpp

#endif  // JPP_K_SWEEP_FULL_H
