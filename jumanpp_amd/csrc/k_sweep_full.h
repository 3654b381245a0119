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

constexpr int kFullCand = 512;   // live (left, slot) candidates per boundary staged in LDS; beyond: an HBM slice (below)
constexpr int kFullChunk = 4;    // right nodes per pass
// The reference has no limit on the candidates of a boundary (score_processor.cc:165-191).  A boundary with more than
// kFullCand takes a slice of the batch's scratch pool: slot = workgroup index mod pool size, held under a spin lock for
// the rest of the sentence (a holder always runs to completion, so waiting for a slot cannot deadlock).
constexpr u32 kFullSlots = 128;
constexpr u32 kFullSlotCand = 16384;   // 128 left nodes x beam 128, or 512 x 32
__host__ __device__ constexpr size_t full_slot_bytes(u32 cap) { return (size_t)cap * (4 * 4 + kFullChunk * 8); }

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

  __shared__ u32 l_c_lnode[kFullCand];   // left node of candidate
  __shared__ u32 l_c_pnode[kFullCand];   // its previous node (T2)
  __shared__ u32 l_c_lk[kFullCand];      // (left << 16) | slot
  __shared__ float l_c_total[kFullCand]; // left element total
  __shared__ u64 l_keys[kFullChunk * kFullCand];
  __shared__ u32 sh_nc;
  u32* c_lnode = l_c_lnode;
  u32* c_pnode = l_c_pnode;
  u32* c_lk = l_c_lk;
  float* c_total = l_c_total;
  u64* keys = l_keys;        // keys[x * cap + i]
  u32 cap = (u32)kFullCand;
  bool haveSlot = false;
  u32 slot = 0;
  auto releaseSlot = [&]() {
    if (haveSlot && lane == 0) {
      __threadfence();
      atomicExch(&B.full_locks[slot], 0u);
    }
  };

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
    auto collect = [&]() {
      if (lane == 0) {
        u32 nc = 0;
        for (u32 l = 0; l < L; ++l) {
          u32 lnode = en[efirst + l];
          for (int k = 0; k < beam; ++k) {
            BeamSlot sl = beams[(u64)lnode * beam + k];
            if (slot_fake(sl)) break;
            if (nc < cap) {
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
    };
    collect();
    u32 nc = sh_nc;
    if (nc > cap && !haveSlot && B.full_slots != 0) {
      // more candidates than the LDS staging holds: the rest of this sentence works in an HBM slice
      slot = blockIdx.x % B.full_slots;
      if (lane == 0) {
        while (atomicCAS(&B.full_locks[slot], 0u, 1u) != 0u) {
#if !defined(JPP_EMU)
          __builtin_amdgcn_s_sleep(32);
#endif
        }
        __threadfence();
      }
      __syncthreads();
      haveSlot = true;
      cap = B.full_cap;
      unsigned char* base = B.full_scratch + (size_t)slot * full_slot_bytes(cap);
      c_lnode = reinterpret_cast<u32*>(base);
      c_pnode = c_lnode + cap;
      c_lk = c_pnode + cap;
      c_total = reinterpret_cast<float*>(c_lk + cap);
      keys = reinterpret_cast<u64*>(c_total + cap);
      collect();
      nc = sh_nc;
    }
    if (nc > cap) {
      if (lane == 0) B.sent_status[s] = ST_CAPACITY;
      releaseSlot();
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
          keys[(size_t)x * cap + i] = ((u64)f32_sortable(score) << 32) | c_lk[i];
        }
      }
      __syncthreads();
      // top `beam` of the unique keys per right node (processBeamCandidates): rank by counting
      for (u32 q = lane; q < nx * nc; q += 64) {
        u32 x = q / nc, i = q - x * nc;
        u64 me = keys[(size_t)x * cap + i];
        u32 rank = 0;
        for (u32 z = 0; z < nc; ++z) rank += keys[(size_t)x * cap + z] > me;
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
  releaseSlot();
}

}  // namespace jpp

#endif  // JPP_K_SWEEP_FULL_H
