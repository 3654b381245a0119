// Training hook: what the reference's trainer reads off an analysed lattice besides the scores.
//   LossCalculator::addTopNgrams                    src/core/training/loss.cc:289-300
//   NgramFeaturesComputer::calculateNgramFeatures   src/core/impl/feature_computer.cc:13-31
//   NgramFeatureImpl<1|2|3>::apply                  src/core/impl/feature_impl_combine.h:41-81
// For every connection on the top-1 path the u32 value of every n-gram feature of the spec for (t2, t1, t0):
// static_cast<u32>(Hasher{}.mix(order + 2).mix(index).mix(seed).mix(t0 pattern)[.mix(t1 pattern)[.mix(t2 pattern)]]),
// i.e. the weight index before it is masked by the table size -- what the perceptron update adds its deltas to.
#ifndef JPP_K_TRAIN_H
#define JPP_K_TRAIN_H

#include "k_t0.h"

namespace jpp {

constexpr int kNumNgram = spec::kNumUni + spec::kNumBi + spec::kNumTri;

// positions of the top-1 path of every sentence (EOS included; 0 for failed sentences)
__global__ void k_path_count(Batch B, u32* counts) {
  const u32 s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= B.n_sent) return;
  counts[s] = B.sent_status[s] == ST_OK ? B.path_len[s] : 0u;
}

// the n-gram feature values of one connection (t2, t1, t0 = node)
__device__ __forceinline__ void ngrams_of(const Batch& B, u32 s, u32 node, u32 t1, u32 t2, u32* out) {
  const u32 g0 = B.byte_off[s] + s;
  const u32 n = B.sent_ncp[s];
  const u64 nb = B.node_base[s];
  const u32* cps = B.cp_code + g0;
  const i32* cls = B.cp_class + g0;
  const NodeInfo ni = B.node_info[nb + node];
  const NodeAux na = B.node_aux[nb + node];
  i32 entry[spec::kNumDicFeatures];
#pragma unroll
  for (int f = 0; f < spec::kNumDicFeatures; ++f) entry[f] = B.node_entry[(nb + node) * B.row_stride + f];
  const bool isUnk = ni.eptr < 0 && ni.eptr != kEptrEOS;
  u64 pat[spec::kNumPatterns];
  t0_patterns(entry, ni, na, isUnk, cps, cls, n, pat);
  const u64* p1 = B.node_pat + (nb + t1) * kPat;
  const u64* p2 = B.node_pat + (nb + t2) * kPat;
#pragma unroll
  for (int u = 0; u < spec::kNumUni; ++u) out[spec::kUni[u].index] = (u32)hmix(uni_prefix(spec::kUni[u].index), pat[spec::kUni[u].t0]);
#pragma unroll
  for (int k = 0; k < spec::kNumBi; ++k)
    out[spec::kBi[k].index] = (u32)hmix(hmix(bi_prefix(spec::kBi[k].index), pat[spec::kBi[k].t0]), p1[spec::kBi[k].t1]);
#pragma unroll
  for (int k = 0; k < spec::kNumTri; ++k)
    out[spec::kTri[k].index] =
        (u32)hmix(hmix(hmix(tri_prefix(spec::kTri[k].index), pat[spec::kTri[k].t0]), p1[spec::kTri[k].t1]), p2[spec::kTri[k].t2]);
}

// one 64-lane workgroup per sentence, one lane per path position (EOS first, like path_nodes)
__global__ void __launch_bounds__(64) k_path_ngrams(Batch B, const u64* off, u32* nodes_out, u32* feat) {
  const u32 s = blockIdx.x;
  if (B.sent_status[s] != ST_OK) return;
  const u32 pl = B.path_len[s];
  const u32* path = B.path_nodes + B.node_base[s];
  const u64 o = off[s];
  for (u32 j = threadIdx.x; j < pl; j += blockDim.x) {
    const u32 node = path[j];
    // the previous two nodes on the path; beyond its start the two BOS nodes (1 = the inner one)
    const u32 t1 = j + 1 < pl ? path[j + 1] : 1u;
    const u32 t2 = j + 2 < pl ? path[j + 2] : (j + 1 < pl ? 1u : 0u);
    ngrams_of(B, s, node, t1, t2, feat + (o + j) * kNumNgram);
    nodes_out[o + j] = node;
  }
}

// The same along given paths in TEXT order (the gold path: LossCalculator::resolveGold, loss.cc:366-389;
// NgramFeatureRef::init / next, feature_computer.h): position j sees positions j-1, j-2 or the BOS nodes.
__global__ void __launch_bounds__(64) k_given_path_ngrams(Batch B, const u64* off, const u32* path_all, u32* feat) {
  const u32 s = blockIdx.x;
  const u64 o = off[s];
  const u32 pl = (u32)(off[s + 1] - o);
  if (pl == 0 || B.sent_status[s] != ST_OK) return;
  const u32* path = path_all + o;
  const u32 N = B.sent_nodes[s];
  for (u32 j = threadIdx.x; j < pl; j += blockDim.x) {
    const u32 node = path[j] < N ? path[j] : N - 1;
    u32 t1 = j >= 1 ? path[j - 1] : 1u;
    u32 t2 = j >= 2 ? path[j - 2] : (j >= 1 ? 1u : 0u);
    if (t1 >= N) t1 = 1;
    if (t2 >= N) t2 = 0;
    ngrams_of(B, s, node, t1, t2, feat + (o + j) * kNumNgram);
  }
}

}  // namespace jpp

#endif  // JPP_K_TRAIN_H
