// Kernel 5, default configuration, two sentences per wavefront.
//
// The boundary recurrence of k_sweep (k_sweep.h) keeps 20-40 of its 64 lanes busy in most phases: a boundary
// of the CLI-default configuration has about 5 right nodes, 4.5 left nodes and 6 global-beam entries.  Here
// every half of a wavefront (32 lanes) walks its own sentence; the two halves execute the same instruction
// stream, so the per-boundary bookkeeping (global beam, T1 dedup, cutoff, cells, beams, loop head) is issued
// once for two sentences and the hash phases run 4 instead of 8 eight-lane groups per sentence.  All loops
// that contain a wave-level operation have wave-uniform trip counts (the maximum over the two halves) and
// predicate the half that is done; everything a half computes is the arithmetic of k_sweep, statement by
// statement, so the results are bit-identical (same goldens, same tests).
//
// Scope: beam 5, global beam 6, right-check 1, right-beam 5 (jumanpp_args.h:50-54), at most 64 right nodes,
// 32 left nodes and 128 candidate slots per boundary.  A sentence that exceeds the staging limits is flagged
// in B.sweep_redo and analysed again, from its first boundary, by k_sweep<8,64,DEF> (launched right after this
// kernel for the flagged sentences only).
//
// Reference behaviour reproduced: see the list at the top of k_sweep.h.
#ifndef JPP_K_SWEEP2_H
#define JPP_K_SWEEP2_H

#include "k_sweep.h"

namespace jpp {

constexpr int kS2Chunk = 4;    // right nodes per pass and half (= 8-lane groups per half)
constexpr int kS2RM = 64;      // right nodes per boundary staged per half
constexpr int kS2Cand = 128;   // candidate slots (left nodes x beam) per half: 4 keys per lane
constexpr u32 kS2Enn = 32;     // ends-list entries staged per half
constexpr u32 kS2Ring = 32;    // layout records resident per half

// lane `src` (a compile-time or wave-uniform lane index) of a half-uniform value, as a wave-uniform value
__device__ __forceinline__ u32 s2_readlane(u32 v, int src) {
#if defined(JPP_EMU)
  return wave_shfl_u32(v, src);
#else
  return (u32)__builtin_amdgcn_readlane((int)v, src);
#endif
}
// maximum over the two halves of a value that is uniform inside each half
__device__ __forceinline__ u32 s2_max2(u32 v) {
  const u32 a = s2_readlane(v, 0), b = s2_readlane(v, 32);
  return a > b ? a : b;
}

template <bool W24>
__global__ void __launch_bounds__(64, 2) k_sweep2(Batch B, const DevModel* __restrict__ Mp, Config cfg) {
  const DevModel& M = *Mp;
  const int lane = (int)threadIdx.x;
  const int h = lane >> 5, hl = lane & 31;
  const u32 sRaw = blockIdx.x * 2u + (u32)h;
  bool live = sRaw < B.n_sent && B.sent_status[sRaw < B.n_sent ? sRaw : 0] == ST_OK;
  const u32 s = sRaw < B.n_sent ? sRaw : 0;   // a safe index for the half that has no sentence
  constexpr int beam = 5, G = 6, rbeam = 5;
  const u32 off = B.byte_off[s];
  const u32 bb0 = off + 4 * s;
  const u32 n = B.sent_ncp[s];
  const u64 nb = B.node_base[s];
  const float JPP_GLOBAL* __restrict__ W = as_global(M.weights);
  const u32 wmask = M.wmask;
  const u32* en = B.end_nodes + nb;
  BeamSlot* beams = B.node_beam + nb * beam;
  const u64* pats = B.node_pat + nb * kPat;
  const float* t0s = B.node_t0 + nb;
  const BndMeta* gmeta = B.bnd_meta + bb0;
  const int S = cfg.nscorers;

  constexpr int GM = 8;
  constexpr int kT2 = 4;
  __shared__ u64 gb_key[2][GM];
  __shared__ u16 gb_left[2][GM];
  __shared__ u16 gb_slot[2][GM];
  __shared__ float gb_score[2][GM];
  __shared__ u32 gb_lnode[2][GM];
  __shared__ u32 gb_pnode[2][GM];
  __shared__ u32 gb_t1[2][GM];
  __shared__ u32 t1node[2][GM];
  __shared__ u64 t1pat[2][G][kPat];
  __shared__ u64 t2pat[2][G][kT2];
  __shared__ float pres[2][kS2RM];
  __shared__ float csum[2][kS2RM];
  __shared__ u16 order[2][kS2RM];
  __shared__ float biS0[2][kS2RM];
  __shared__ float biS[2][kS2Chunk][GM];
  __shared__ float tot[2][kS2Chunk][GM];
  __shared__ u64 pR[2][kS2Chunk][kPat];
  __shared__ float t0R[2][kS2Chunk];
  // prefetch targets (global_load_lds: lane L writes slot L, i.e. [half][32 slots])
  __shared__ __attribute__((aligned(16))) u64 pRn[2][2][64];     // [buffer][half][4 rows x 14 patterns (+ pad)]
  __shared__ __attribute__((aligned(16))) float t0n[2][2][32];   // [buffer][half][first 4 used]
  __shared__ __attribute__((aligned(16))) u32 enn[2][2][kS2Enn];
  __shared__ __attribute__((aligned(16))) BndMeta meta[2][kS2Ring];
  // candidate keys (phase 1) and first-stage hash states (phases 3-5) are never live together
  constexpr int kS1 = 40;
  constexpr u32 kKeyBytes = kS2Cand * sizeof(u64), kS1Bytes = kS2Chunk * kS1 * sizeof(u64);
  __shared__ __attribute__((aligned(16))) unsigned char u_buf[2][kKeyBytes > kS1Bytes ? kKeyBytes : kS1Bytes];
  u64* const ckey = reinterpret_cast<u64*>(u_buf[h]);
  u64(*const s1b)[kS1] = reinterpret_cast<u64(*)[kS1]>(u_buf[h]);
  __shared__ u64 s1t[2][kS2Chunk][spec::kNumTri];
  __shared__ u64 s_tripre[spec::kNumTri];
  __shared__ u8 s_trit[spec::kNumTri][4];
  __shared__ u64 s_bipre[spec::kNumBi];
  __shared__ u8 s_bit01[spec::kNumBi];
  static_assert(spec::kNumBi <= 64 && spec::kNumBi <= kS1, "feature tables");
  if (lane < spec::kNumBi) {
    s_bipre[lane] = kNg.bi_pre[lane];
    s_bit01[lane] = (u8)((kNg.bi_t0[lane] << 4) | kNg.bi_t1[lane]);
  }
  if (lane < spec::kNumTri) {
    s_tripre[lane] = kNg.tri_pre[lane];
    s_trit[lane][0] = (u8)kNg.tri_t0[lane];
    s_trit[lane][1] = (u8)kNg.tri_t1[lane];
    s_trit[lane][2] = (u8)kNg.tri_t2[lane];
  }
  LaneBi lbi;
  lbi.pre = s_bipre;
  lbi.t01 = s_bit01;
  const int grp = hl >> 3, gj = hl & 7;   // 8-lane group inside the half, lane inside the group
  wave_sync();

  if (live && n == 0) {
    // empty input: the reference returns before scoring anything (analyzer_impl.cc:255-258)
    if (hl < beam) beams[(u64)2 * beam + hl] = BeamSlot{kFake16, kFake16, 0.f, 0xffffffffu, 0};
    if (hl == 0) B.bnd_ngb[bb0 + 2] = 0;
    live = false;
  }
  if (wave_ballot(live) == 0) return;

  // layout records: ring of kS2Ring entries per half (slot q mod 32), refilled asynchronously
  u32 metaEnd = (n + 3) < kS2Ring ? (n + 3) : kS2Ring;
  lds_async_load<16>(&meta[0][0], gmeta + hl, live && (u32)hl < metaEnd);
  lds_async_wait();
  wave_sync();
  u32 metaReady = metaEnd;
  auto metaAt = [&](u32 q) -> BndMeta {
    if (q < metaReady) return meta[h][q & (kS2Ring - 1)];
    return load_bnd_meta(gmeta, q);   // noinline: see k_sweep.h
  };
  auto next_nonempty = [&](u32 from, BndMeta& out) {
    u32 q = from;
    for (; q <= n + 2; ++q) {
      out = metaAt(q);
      if (out.cnt != 0) break;
    }
    return q;
  };
  // static data of boundary bq into buffer `buf`: its first 4 right nodes' pattern rows and T0, its ends list
  auto prefetch = [&](bool on, u32 bq, const BndMeta& mq, int buf) {
    const bool ok = on && bq <= n + 2;
    const u32 Rq = mq.cnt, rf = mq.first, Lq = mq.ecnt, ef = mq.efirst;
    const u32 nxr = Rq < (u32)kS2Chunk ? Rq : (u32)kS2Chunk;
    static_assert(kPat * 8 % 16 == 0 && kS2Chunk * kPat * 8 / 16 <= 32, "one dwordx4 per lane covers a chunk");
    lds_async_load<16>(&pRn[buf][0][0], reinterpret_cast<const char*>(pats + (u64)rf * kPat) + hl * 16,
                       ok && (u32)hl < nxr * (kPat * 8 / 16));
    lds_async_load<4>(&t0n[buf][0][0], t0s + rf + hl, ok && (u32)hl < nxr);
    lds_async_load<4>(&enn[buf][0][0], en + ef + hl, ok && (u32)hl < (Lq < kS2Enn ? Lq : kS2Enn));
  };
  JPP_PROF_DECL;
  BndMeta mbn{0, 0, 0, 0};
  u32 bn = live ? next_nonempty(2, mbn) : 0xffffffffu;
  int par = 0;
  prefetch(live, bn, mbn, par);
  lds_async_wait();
  for (;; par ^= 1) {
    const u32 b = bn;
    bool active = live && b <= n + 2;
    if (wave_ballot(active) == 0) break;
    wave_sync();
    const BndMeta mb = mbn;
    const u32 R = active ? mb.cnt : 0;
    const u32 rfirst = mb.first;
    const u32 L = active ? mb.ecnt : 0;
    const u32 ncand = L * (u32)beam;
    if (active && (R > (u32)kS2RM || L > kS2Enn || ncand > (u32)kS2Cand)) {
      // beyond the staging of this kernel: the sentence is analysed again by k_sweep<8,64,DEF>
      if (hl == 0) B.sweep_redo[s] = 1;
      live = false;
      active = false;
    }
    // ring refill (one instruction serves both halves: lane hl can only fill slot hl)
    metaReady = metaEnd;
    {
      const bool want = active && metaEnd < n + 3 && metaEnd <= b + kS2Ring / 2;
      const u32 lo = metaEnd, hi = (b + kS2Ring) < (n + 3) ? (b + kS2Ring) : (n + 3);
      const u32 q = lo + (((u32)hl - lo) & (kS2Ring - 1));   // the record of [lo, lo + 32) that lives in slot hl
      lds_async_load<16>(&meta[0][0], gmeta + q, want && q < hi);
      if (want) metaEnd = hi;
    }
    if (active) bn = next_nonempty(b + 1, mbn);
    else bn = 0xffffffffu;
    prefetch(active, bn, mbn, par ^ 1);
    const u32* enL = enn[par][h];

    JPP_PROF(0);
    // ---- 1. global beam: top-G of all live (left, slot) by the packed key ----
    u64 mykey[kS2Cand / 32];
    u32 myprev[kS2Cand / 32], mylnode[kS2Cand / 32];
#pragma unroll
    for (int jx = 0; jx < kS2Cand / 32; ++jx) {
      const u32 q = (u32)hl + 32u * jx;
      u64 key = 0;
      u32 pv = 0, ln = 0;
      if (q < ncand) {
        const u32 l = q / (u32)beam, k = q - l * (u32)beam;
        ln = enL[l];
        const BeamSlot sl = beams[(u64)ln * beam + k];
        pv = sl.prev_node;
        if (!slot_fake(sl)) key = ((u64)f32_sortable(sl.total) << 32) | ((u64)l << 16) | k;
        ckey[q] = key;
      }
      mykey[jx] = key;
      myprev[jx] = pv;
      mylnode[jx] = ln;
    }
    // The loads above were issued after the prefetch requests of this iteration, and vector memory completes in
    // order: waiting here costs nothing the keys do not need anyway, and it guarantees that the next boundary's
    // static data has landed in LDS before the next iteration reads it (no wait at the loop head, which would
    // drain the stores of phase 5).
    lds_async_wait();
    wave_sync();
    int ngb = 0;
    {
      // the keys are unique: the global beam is "every key with fewer than G larger ones"
      const u32 ncMax = s2_max2(ncand);
      const u32 nj = (ncMax + 31) / 32;
      u32 rank[kS2Cand / 32];
      int liveKeys = 0;
#pragma unroll
      for (int jx = 0; jx < kS2Cand / 32; ++jx) {
        rank[jx] = 0;
        liveKeys += popc64((wave_ballot(mykey[jx] != 0) >> (32 * h)) & 0xffffffffull);
      }
      for (u32 z = 0; z < ncMax; ++z) {
        const u64 kz = z < ncand ? ckey[z] : 0;
#pragma unroll
        for (int jx = 0; jx < kS2Cand / 32; ++jx)
          if ((u32)jx < nj) rank[jx] += kz > mykey[jx] ? 1u : 0u;
      }
      ngb = liveKeys < G ? liveKeys : G;
#pragma unroll
      for (int jx = 0; jx < kS2Cand / 32; ++jx) {
        if (mykey[jx] != 0 && rank[jx] < (u32)G) {
          const u32 r = rank[jx];
          const u64 key = mykey[jx];
          const u32 l = (u32)(key >> 16) & 0xffff, k = (u32)key & 0xffff;
          const float sc0 = sortable_f32((u32)(key >> 32));
          gb_key[h][r] = key;
          gb_left[h][r] = (u16)l;
          gb_slot[h][r] = (u16)k;
          gb_score[h][r] = sc0;
          gb_lnode[h][r] = mylnode[jx];
          gb_pnode[h][r] = myprev[jx];
          B.bnd_gbeam[(u64)(bb0 + b) * G + r] = GbeamEntry{(u16)l, (u16)k, sc0};
        }
      }
    }
    if (active && hl == 0) B.bnd_ngb[bb0 + b] = (u32)ngb;
    wave_sync();
    if (active && ngb == 0) {
      // unreachable boundary: every right node gets an all-fake beam (makeT0Beam with an empty gbeam)
      for (u32 q = hl; q < R * (u32)beam; q += 32) beams[(u64)rfirst * beam + q] = BeamSlot{kFake16, kFake16, 0.f, 0xffffffffu, 0};
      for (u32 q = hl; q < R; q += 32) B.node_kept[nb + rfirst + q] = 0;
    }
    const bool scored = active && ngb > 0;
    const u32 Rs = scored ? R : 0;   // right nodes this half scores at this boundary

    JPP_PROF(1);
    // ---- 2. T1 dedup in first-seen order, gather T1 / T2 pattern rows ----
    int U = 0;
    {
      int first = hl;
      if (scored && hl < ngb) {
        const u16 left = gb_left[h][hl];
        for (int j = 0; j < hl; ++j) {
          if (gb_left[h][j] == left) {
            first = j;
            break;
          }
        }
      }
      const bool isFirst = scored && hl < ngb && first == hl;
      const u32 fmask = (u32)((wave_ballot(isFirst) >> (32 * h)) & 0xffffffffull);
      U = __builtin_popcount(fmask);
      if (scored && hl < ngb) {
        const u32 u = (u32)__builtin_popcount(fmask & ((1u << first) - 1));
        gb_t1[h][hl] = u;
        if (isFirst) t1node[h][u] = gb_lnode[h][hl];
      }
    }
    wave_sync();
    {
      const u32 cnt = scored ? (u32)(U * kPat + ngb * kT2) : 0;
      const u32 cntMax = s2_max2(cnt);
      for (u32 q0 = 0; q0 < cntMax; q0 += 32) {
        const u32 q = q0 + (u32)hl;
        if (q < cnt) {
          if (q < (u32)(U * kPat)) {
            const u32 row = q / kPat, p = q - row * kPat;
            t1pat[h][row][p] = pats[(u64)t1node[h][row] * kPat + p];
          } else {
            const u32 r2 = (q - U * kPat) / kT2, p = (q - U * kPat) - r2 * kT2;
            t2pat[h][r2][p] = pats[(u64)gb_pnode[h][r2] * kPat + p];
          }
        }
      }
    }
    wave_sync();

    JPP_PROF(2);
    // first-stage states of `nx` right nodes whose pattern rows are rows[x * kPat + p]: lane per (node, feature)
    auto compute_s1 = [&](const u64* rows, u32 nx) {
      constexpr u32 kF = spec::kNumBi + spec::kNumTri;
      for (u32 q0 = 0; q0 < kS2Chunk * kF; q0 += 32) {
        const u32 q = q0 + (u32)hl;
        const u32 x = q / kF, k = q - x * kF;
        if (x < nx) {
          if (k < (u32)spec::kNumBi) s1b[x][k] = hmix(s_bipre[k], rows[x * kPat + (s_bit01[k] >> 4)]);
          else s1t[h][x][k - spec::kNumBi] = hmix(s_tripre[k - spec::kNumBi], rows[x * kPat + s_trit[k - spec::kNumBi][0]]);
        }
      }
    };
    // ---- 3. prescores of global-beam entry 0 over all right nodes (c == 1) ----
    const u32 RsMax = s2_max2(Rs);
    for (u32 tc = 0; tc < RsMax; tc += kS2Chunk) {
      const bool on = tc < Rs;
      const u32 nx = on ? ((Rs - tc) < (u32)kS2Chunk ? (Rs - tc) : (u32)kS2Chunk) : 0;
      if (tc != 0) {
        for (u32 q = hl; q < nx * kPat; q += 32) pR[h][q / kPat][q % kPat] = pats[(u64)(rfirst + tc) * kPat + q];
        if ((u32)hl < nx) t0R[h][hl] = t0s[rfirst + tc + hl];
        wave_sync();
      }
      const u64* rows = tc == 0 ? pRn[par][h] : &pR[h][0][0];
      const float* t0c = tc == 0 ? t0n[par][h] : t0R[h];
      compute_s1(rows, nx);
      wave_sync();
      {
        const bool act = (u32)grp < nx;
        const u32 t = tc + (u32)grp;
        const int xr = act ? grp : 0;
        const u64* t1r = t1pat[h][gb_t1[h][0]];
        const u64* t2r = t2pat[h][0];
        float w[kBiPerLane];
        bi_gather_s1<W24>(lbi, gj, s1b[xr], t1r, W, wmask, act, w);
        float g = 0.f;
        if (act && gj < spec::kNumTri) {
          u32 idx = hmix_index<W24>(hmix(s1t[h][xr][gj], t1r[s_trit[gj][1]]), t2r[s_trit[gj][2]], wmask);
          g += W[idx];
        }
        // generated applyBiStep2 (8 round-robin sums; last right node: unrolled-4) and applyTriStep3
        const float b8 = bi_sum8(w, lane, gj);
        const float b4 = bi_sum4(w, lane, gj);
        // the tail-association sum of (right node, T1 row 0) from the same weights (see k_sweep.h)
        const float s2v = bi_sum2(w, lane, gj);
        const float tailS = (U == 1) ? b4 : s2v;   // row 0 is the last T1 row iff U == 1
        static_assert(spec::kNumTri == 4, "the trigram sum below reads group members 1..3");
        float tsum = g;
        tsum += row_shl_f32<1>(g);
        tsum += row_shl_f32<2>(g);
        tsum += row_shl_f32<3>(g);
        if (act && gj == 0) {
          biS0[h][t] = tailS;
          float sc = t0c[grp];
          sc += (t == R - 1) ? b4 : b8;
          sc += tsum;
          if (B.node_penalty) sc -= B.node_penalty[nb + rfirst + t];   // applyPluginToPrescores
          pres[h][t] = sc;
        }
      }
      wave_sync();
    }

    JPP_PROF(3);
    // ---- 4. right-node cutoff (std::nth_element semantics) ----
    const u32 K = (u32)rbeam < Rs ? (u32)rbeam : Rs;
    for (u32 t = hl; t < Rs; t += 32) order[h][t] = (u16)t;
    const bool needCut = Rs > (u32)rbeam;
    if (wave_ballot(needCut) != 0) {
      for (u32 t = hl; t < Rs; t += 32) {
        float sc = 0.f;
        sc += pres[h][t];
        csum[h][t] = sc;
      }
      wave_sync();
      if (needCut) {
        for (u32 t = hl; t < Rs; t += 32) {
          const float me = csum[h][t];
          u32 rank = 0;
          for (u32 u = 0; u < Rs; ++u) {
            const float o = csum[h][u];
            rank += (o > me || (o == me && u < t)) ? 1u : 0u;
          }
          order[h][rank] = (u16)t;
        }
      }
      wave_sync();
      const bool tieAtCut = needCut && csum[h][order[h][rbeam - 1]] == csum[h][order[h][rbeam]];
      wave_sync();
      if (tieAtCut) {
        for (u32 t = hl; t < Rs; t += 32) order[h][t] = (u16)t;
      }
      wave_sync();
      if (tieAtCut && hl == 0) {
        ScoreGreater cmp{csum[h]};
        nth_element_u16(order[h], order[h] + rbeam, order[h] + Rs, cmp);
      }
    }
    wave_sync();

    JPP_PROF(4);
    // ---- 5. score + beams, 4 right nodes at a time in cutoff order ----
    const int ntail = ngb - 1;
    for (u32 op0 = 0; op0 < RsMax; op0 += kS2Chunk) {
      const bool on = op0 < Rs;
      const int nx = on ? (int)((Rs - op0) < (u32)kS2Chunk ? (Rs - op0) : (u32)kS2Chunk) : 0;
      // patterns / T0 of this pass's right nodes in cutoff order: rows of the prefetched buffer when the
      // whole boundary fits one chunk, staged from HBM otherwise
      const bool small = Rs <= (u32)kS2Chunk;
      const bool anyBig = wave_ballot(on && !small) != 0;
      if (anyBig) {
        if (on && !small) {
          for (int q = hl; q < nx * kPat; q += 32) pR[h][q / kPat][q % kPat] = pats[(u64)(rfirst + order[h][op0 + q / kPat]) * kPat + q % kPat];
          if (hl < nx) t0R[h][hl] = t0s[rfirst + order[h][op0 + hl]];
        }
        wave_sync();
        compute_s1(&pR[h][0][0], (on && !small) ? (u32)nx : 0u);
        wave_sync();
      }
      auto s1Row = [&](int x) -> int { return small ? (int)order[h][op0 + x] : x; };
      auto t0Of = [&](int x) -> float { return small ? t0n[par][h][order[h][op0 + x]] : t0R[h][x]; };
      // 5a. bigram sums per (kept node, unique T1 row 1 .. U-1) -- applyBiTriFullKernel rows
      {
        const int Ur = U - 1;
        const int units = (on && ntail > 0 && Ur > 0) ? nx * Ur : 0;
        const int unitsMax = (int)s2_max2((u32)units);
        for (int base = 0; base < unitsMax; base += kS2Chunk) {
          const int u = base + grp;
          bool act = u < units;
          const int x = act ? u / Ur : 0;
          const int tu = act ? (u - x * Ur) + 1 : 0;
          act = act && (op0 + (u32)x) < K;
          float w[kBiPerLane];
          bi_gather_s1<W24>(lbi, gj, s1b[act ? s1Row(x) : 0], t1pat[h][tu], W, wmask, act, w);
          const float s2 = bi_sum2(w, lane, gj);
          const float s4 = bi_sum4(w, lane, gj);
          if (act && gj == 0) biS[h][x][tu] = (tu == U - 1) ? s4 : s2;
        }
      }
      wave_sync();
      JPP_PROF(5);
      // 5b. cells and totals per (node, gbeam entry): nx * ngb <= 24 lanes
      {
        const int q = hl;
        if (on && q < nx * ngb) {
          const int x = q / ngb, i = q - x * ngb;
          const bool kept = (op0 + (u32)x) < K;
          const u32 t = order[h][op0 + x];
          float cell, total;
          bool defined = true;
          if (i < 1) {
            // copyT0Scores(head, result, 0): v += 0; cell = v; v += gb.score()
            float v = pres[h][t];
            v += 0.f;
            cell = v;
            v += gb_score[h][i];
            total = v;
          } else if (kept) {
            const u64* st = s1t[h][s1Row(x)];
            const u64* t1r = t1pat[h][gb_t1[h][i]];
            const u64* t2r = t2pat[h][i];
            float w[spec::kNumTri];
#pragma unroll
            for (int f = 0; f < spec::kNumTri; ++f) {
              u32 idx = hmix_index<W24>(hmix(st[f], t1r[kNg.tri_t1[f]]), t2r[kNg.tri_t2[f]], wmask);
              w[f] = W[idx];
            }
            static_assert(spec::kNumTri == 4, "trigram association below is written for 4 features");
            float Sv;
            if (gb_t1[h][i] == 0) Sv = biS0[h][t];
            else Sv = biS[h][x][gb_t1[h][i]];
            float res;
            if (i < ngb - 1) {
              float r1 = 0.f, r2 = 0.f;
              r1 += w[0];
              r2 += w[1];
              r1 += w[2];
              r2 += w[3];
              res = Sv + r1 + r2;
            } else {
              float q1 = 0.f, q2 = 0.f, q3 = 0.f, q4 = 0.f;
              q1 += w[0];
              q2 += w[1];
              q3 += w[2];
              q4 += w[3];
              res = Sv + (q1 + q2 + q3 + q4);
            }
            // applyPluginToGbeam (score_processor.cc:598-613), then copyT0Scores(tail, resultTail, t0Score)
            float v = res;
            if (B.node_penalty) v -= B.node_penalty[nb + rfirst + t];
            v += t0Of(x);
            cell = v;
            v += gb_score[h][i];
            total = v;
          } else {
            defined = false;
            cell = 0.f;
            total = 0.f;
          }
          tot[h][x][i] = total;
          if (defined) B.node_cells[((nb + rfirst + t) * G + i) * S] = cell;
        }
      }
      wave_sync();
      JPP_PROF(6);
      // 5c. beams: stable descending rank among the node's candidates (makeT0Beam; <= 6 candidates)
      {
        const int x = hl >> 3, i = hl & 7;
        if (on && x < nx) {
          const bool kept = (op0 + (u32)x) < K;
          const u32 t = order[h][op0 + x];
          const int cnt = kept ? ngb : 1;
          BeamSlot* row = beams + (u64)(rfirst + t) * beam;
          if (i < cnt) {
            const float me = tot[h][x][i];
            int rank = 0;
            for (int jx = 0; jx < cnt; ++jx) {
              const float o = tot[h][x][jx];
              if (o > me || (o == me && jx < i)) ++rank;
            }
            if (rank < beam) row[rank] = BeamSlot{gb_left[h][i], gb_slot[h][i], me, gb_lnode[h][i], (u32)i};
          } else if (i < beam) {
            row[i] = BeamSlot{kFake16, kFake16, 0.f, 0xffffffffu, 0};
          }
          if (i == 0) B.node_kept[nb + rfirst + t] = kept ? 1 : 0;
        }
      }
      wave_sync();
      JPP_PROF(7);
    }
  }
  JPP_PROF_FLUSH;
}

}  // namespace jpp

#endif  // JPP_K_SWEEP2_H
