// Kernels 3: lattice layout.  All are one lane per sentence: they are short
// sequential passes over ~C positions / ~N nodes, and 64k sentences keep the
// chip busy.
//
// Reference behaviour reproduced:
//   LatticeBuilder::prepare            src/core/analysis/lattice_builder.cc:16-39
//   LatticeBuilder::checkConnectability src/core/analysis/lattice_builder.cc:41-52
//   AnalyzerImpl::prepareNodeSeeds     src/core/analysis/analyzer_impl.cc:128-139
//   LatticeBuilder::makeBos/makeEos/fillEnds  lattice_builder.cc:95-145
#ifndef JPP_K_LATTICE_H
#define JPP_K_LATTICE_H

#include "jpp_device.h"

namespace jpp {

// STAGE 1: layout with dic + stage-1 makers.  STAGE 2: (flagged sentences only) all makers.
template <int STAGE>
__global__ void k_layout(Batch B) {
  u32 s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= B.n_sent) return;
  if (STAGE == 2 && (B.sent_status[s] != ST_OK || (B.sent_flags[s] & 2) == 0)) {
    B.sent_nodes2[s] = 0;
    return;
  }
  u32 off = B.byte_off[s];
  u32 g0 = off + s;
  u32 bb0 = off + 4 * s;
  u32 n = B.sent_status[s] == ST_OK ? B.sent_ncp[s] : 0;
  // local node ids: 0,1 = BOS; boundary b = i + 2 starts at position i
  B.bnd_first[bb0 + 0] = 0;
  B.bnd_cnt[bb0 + 0] = 1;
  B.bnd_first[bb0 + 1] = 1;
  B.bnd_cnt[bb0 + 1] = 1;
  u32 next = 2;
  u32 sum2 = 0;
  u32 maxR = 0;
  bool overflow = false;
  for (u32 i = 0; i < n; ++i) {
    u32 c = (u32)B.pos_cnt1[g0 + i] + B.pos_cntN[g0 + i];
    if (B.pos_cnt1[g0 + i] == 0xffff) overflow = true;
    sum2 += B.pos_cnt2[g0 + i];
    if (STAGE == 2) c += B.pos_cnt2[g0 + i];
    B.bnd_first[bb0 + i + 2] = next;
    B.bnd_cnt[bb0 + i + 2] = c;
    next += c;
    if (c > maxR) maxR = c;
  }
  if (maxR > 0) atomicMax(&B.gstats[0], maxR);
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

// single-workgroup exclusive scan of u32 counts into u64 offsets (+ base);
// out[n] receives the total.  n is at most a few hundred thousand.
__global__ void k_scan(const u32* in, u64* out, u32 n, const u64* base_ptr) {
  __shared__ u64 part[1024];
  u32 t = threadIdx.x;
  u32 nt = blockDim.x;
  u32 per = (n + nt - 1) / nt;
  u32 lo = t * per;
  u32 hi = lo + per < n ? lo + per : n;
  u64 sum = 0;
  for (u32 i = lo; i < hi; ++i) sum += in[i];
  part[t] = sum;
  __syncthreads();
  if (t == 0) {
    u64 acc = base_ptr ? *base_ptr : 0;
    for (u32 k = 0; k < nt; ++k) {
      u64 v = part[k];
      part[k] = acc;
      acc += v;
    }
    out[n] = acc;
  }
  __syncthreads();
  u64 acc = part[t];
  for (u32 i = lo; i < hi; ++i) {
    out[i] = acc;
    acc += in[i];
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
template <int PASS>
__global__ void k_connect(Batch B) {
  u32 s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= B.n_sent) return;
  if (B.sent_status[s] != ST_OK) return;
  if (PASS == 2 && (B.sent_flags[s] & 2) == 0) return;
  u32 off = B.byte_off[s];
  u32 g0 = off + s;
  u32 bb0 = off + 4 * s;
  u32 n = B.sent_ncp[s];
  u8* reach = B.reach + g0;  // n + 1 entries
  for (u32 i = 0; i <= n; ++i) reach[i] = 0;
  reach[0] = 1;
  const NodeInfo* ni = B.node_info + B.node_base[s];
  for (u32 i = 0; i < n; ++i) {
    if (!reach[i]) continue;
    u32 first = B.bnd_first[bb0 + i + 2];
    u32 cnt = B.bnd_cnt[bb0 + i + 2];
    for (u32 k = 0; k < cnt; ++k) reach[ni[first + k].end] = 1;
  }
  if (!reach[n]) {
    if (PASS == 1) B.sent_flags[s] |= 2;
    else B.sent_status[s] = ST_NO_LATTICE;
  }
}

// BOS/EOS nodes, UNK entry pointers, ends lists.
__global__ void k_ends(Batch B, Config cfg) {
  u32 s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= B.n_sent) return;
  if (B.sent_status[s] != ST_OK) return;
  u32 off = B.byte_off[s];
  u32 bb0 = off + 4 * s;
  u32 n = B.sent_ncp[s];
  u32 N = B.sent_nodes[s];
  u64 nb = B.node_base[s];
  NodeInfo* ni = B.node_info + nb;
  NodeAux* na = B.node_aux + nb;
  ni[0] = NodeInfo{kEptrBOS, 0, 0};
  ni[1] = NodeInfo{kEptrBOS, 0, 0};
  ni[N - 1] = NodeInfo{kEptrEOS, (u16)n, (u16)n};
  na[0] = na[1] = na[N - 1] = NodeAux{0, 0, 0, 0, 0, 0};
  // per-boundary end counts
  u32* ecnt = B.end_cnt + bb0;
  u32* efirst = B.end_first + bb0;
  for (u32 b = 0; b <= n + 2; ++b) ecnt[b] = 0;
  ecnt[1] = 1;
  ecnt[2] = 1;
  i32 unk = 0;
  for (u32 k = 2; k + 1 < N; ++k) {
    NodeInfo x = ni[k];
    if (x.eptr < 0) {
      x.eptr = ~unk;
      ++unk;
      ni[k] = x;
    }
    ecnt[x.end + 2] += 1;
  }
  u32 acc = 0;
  for (u32 b = 0; b <= n + 2; ++b) {
    efirst[b] = acc;
    acc += ecnt[b];
    ecnt[b] = 0;
  }
  u32* en = B.end_nodes + nb;
  en[efirst[1] + ecnt[1]++] = 0;
  en[efirst[2] + ecnt[2]++] = 1;
  for (u32 k = 2; k + 1 < N; ++k) {
    u32 b = (u32)ni[k].end + 2;
    en[efirst[b] + ecnt[b]++] = k;
  }
  // BOS beams (reference AnalyzerImpl::bootstrapAnalysis, analyzer_impl.cc:179-195)
  BeamSlot* bm = B.node_beam + nb * cfg.beam;
  for (int q = 0; q < cfg.beam; ++q) {
    bm[q] = BeamSlot{kFake16, kFake16, 0.f, 0xffffffffu, 0};
    bm[cfg.beam + q] = BeamSlot{kFake16, kFake16, 0.f, 0xffffffffu, 0};
  }
  bm[0] = BeamSlot{0, 0, 0.f, 0xffffffffu, 0};
  bm[cfg.beam] = BeamSlot{0, 0, 0.f, 0u, 0};
  // BOS patterns (LatticeConstructionContext::addBos, lattice_builder.cc:173-179)
  u64* pat = B.node_pat + nb * kPat;
  for (int q = 0; q < 2 * kPat; ++q) pat[q] = (u64)(u32)kEptrBOS;
}

}  // namespace jpp

#endif  // JPP_K_LATTICE_H
