// HBM-resident data layout of one analysis batch (see DESIGN.md "Data layout").
#ifndef JPP_TYPES_H
#define JPP_TYPES_H

#include "jpp_rt.h"

namespace jpp {

constexpr int kMaxUnkMakers = 16;
constexpr int kMaxDicFeatures = 16;
constexpr int kPat = 14;          // stored patterns per node (jumandic spec)
constexpr int kMaxGbeam = 32;     // global beam capacity (k_sweep<8> for the CLI defaults, k_sweep<32> beyond)
constexpr int kMaxBeam = 32;
constexpr int kMaxRight = 512;    // right nodes per boundary staged in LDS by the sweep kernel
constexpr int kMaxNormStates = 64;
constexpr int kMaxNormResults = 160;
constexpr int kNormCache = 8;        // normalized-node results per start kept from the count pass for the emit passes
constexpr int kMaxRnnE = 256;      // RNN hidden size staged per lane (E/64 <= 4)

// entry pointers (reference src/core/core_types.h:44-58)
constexpr i32 kEptrBOS = (i32)0x80000000;
constexpr i32 kEptrEOS = (i32)0x80000002;

// per-sentence status codes (mirror of the reference Status kinds that
// Analyzer::analyze can return, src/core/analysis/analysis_input.cc:12-33,
// src/util/characters.cc:267-269, src/core/analysis/analyzer_impl.cc:133-135)
enum SentStatus : i32 {
  ST_OK = 0,
  ST_TOO_LONG = 1,        // > maxInputBytes (InvalidParameter)
  ST_BAD_UTF8 = 2,        // InvalidParameter
  ST_NO_LATTICE = 3,      // InvalidState "could not build lattice"
  ST_CAPACITY = 4,        // an internal device-side capacity was exceeded (not a reference status)
};

// UNK maker kinds: spec::UnkMakerType (reference src/core/spec/spec_types.h)
enum UnkType : i32 { UNK_SINGLE = 1, UNK_CHUNKING = 2, UNK_ONOMATOPOEIA = 3, UNK_NUMERIC = 4, UNK_NORMALIZE = 5 };

struct PcNode {
  u16 boundary;
  u16 length;
  u32 tag_first;
  u32 tag_count;
};
struct PcTag {
  i32 field;
  i32 value;
};

// the four per-boundary layout words k_sweep needs, in one 16-byte record
struct alignas(16) BndMeta {
  u32 first;   // first node starting at the boundary
  u32 cnt;     // R_b
  u32 efirst;  // offset of the ends list
  u32 ecnt;    // L_b
};

struct UnkMaker {
  i32 type;
  i32 char_class;
  i32 pattern_ptr;      // EntryPtr raw of the template entry
  i32 priority;         // 0 = stage 1, 1 = stage 2 (only if the lattice is disconnected)
  i32 placeholder;      // target placeholder index or -1
  u32 replace_mask;     // bit f set: entry feature f is replaced by the surface hash
  u32 pattern_mask;     // features compared by dicPatternMatches (numeric maker)
  i32 spec_index;       // position in spec.unkCreators (= jppgpu_model::unk_makers), reported in jppgpu_unk::maker
  i32 rank;             // position in the reference's creation sequence of the makers (stage 1 in spec order, then stage 2)
  i32 tmpl[kMaxDicFeatures];  // decoded template entry row
};

// Feature descriptors of a spec other than the compiled-in jumandic tables: what the
// reference's DYNAMIC feature objects are built from when the spec hash does not match its generated code
// (features_api.cc:20-60).  Table-driven k_t0_dyn / k_sweep<.., DYN> read them; the limits are those of the
// device layout (entry rows of at most 8 columns, at most kPat patterns referenced by bigrams / trigrams, ...).
constexpr int kDynMaxPrims = 32, kDynMaxComputes = 32, kDynMaxPatterns = 64, kDynMaxArgs = 8, kDynMaxBranch = 8;
constexpr int kDynMaxUni = 64, kDynMaxBi = 40, kDynMaxTri = 4, kDynMaxStorages = 8;
struct DevSpec {
  i32 nprims, ncomputes, npatterns, nstored, nuni, nbi, ntri, pad0;
  struct Prim { i32 kind, a, b; } prims[kDynMaxPrims];
  struct Compute { i32 cond, nt, nf; i32 t[kDynMaxBranch]; i32 f[kDynMaxBranch]; } computes[kDynMaxComputes];
  struct Pattern { i32 nargs; i32 slot; u64 prefix; i32 args[kDynMaxArgs]; } patterns[kDynMaxPatterns];   // slot: stored position or -1
  struct Uni { u64 prefix; i32 t0; i32 pad; } uni[kDynMaxUni];      // t0: pattern index
  u64 bi_prefix[kDynMaxBi];
  u8 bi_t01[kDynMaxBi];        // (t0 slot << 4) | t1 slot
  u64 tri_prefix[kDynMaxTri];
  u8 tri_t[kDynMaxTri][4];     // t0, t1, t2 slots
  // column storages of the length primitives (ByteLength / CodepointSize: prims[i].b = index here), in HBM
  struct Storage { const u8* data; u64 bytes; u32 kind; u32 align; } storages[kDynMaxStorages];
  i32 nstorages, pad1;
};

struct DevModel {
  const DevSpec* spec;       // null: the built-in jumandic tables
  const u32* trie;
  const u8* entry_ptrs;
  const u8* entry_data;
  const float* weights;
  u32 trie_units;
  u32 entry_ptrs_bytes;
  u32 entry_data_bytes;
  u32 wmask;
  i32 num_features;
  i32 n_unk;
  i32 n_stage1;          // makers[0..n_stage1) are stage 1 in spec order, the rest stage 2
  i32 norm_maker;        // index of the Normalize maker or -1
  UnkMaker makers[kMaxUnkMakers];
  i32 maker_of_spec[kMaxUnkMakers];  // spec order (NodeAux::maker) -> index into makers[]
  // RNN re-ranker (reference RNN model part, rnn_scorer_gbeam.cc:375-398,426-470)
  i32 has_rnn;
  const u32* rnn_known;      // word -> id double array for dictionary nodes
  const u32* rnn_unk;        // word -> id double array for UNK nodes
  const float* rnn_wt;       // W transposed and zero-padded: wt[k * EP + i] = W[i * E + k]
  const float* rnn_emb;      // [V][E]
  const float* rnn_nce;      // [V][E]
  const float* rnn_maxent;   // [M]
  u32 rnn_E;
  u32 rnn_EP;                // E rounded up to 64, 128 or 256 (row stride of rnn_wt and rnn_ctx)
  u32 rnn_order;             // maxent order
  u64 rnn_hash_max;          // maxentSize - vocabSize
  u64 rnn_hash_magic;        // floor((2^64 - 1) / rnn_hash_max) for fastmod_u64
  u64 rnn_mx_base;           // maxent context hash of order i = rnn_mx_base + (prevId + 1) * rnn_mx_coef[i]
  u64 rnn_mx_coef[4];        //   (MikolovIndexCalculator with every context slot = prevId, mikolov_rnn_impl.h:21-60)
  float rnn_nce_const;
  i32 rnn_unk_id;
  float rnn_unk_const;
  float rnn_unk_len;
  u32 rnn_nfields;
  u32 rnn_fields[8];
};

struct Config {
  i32 beam;
  i32 gbeam;
  i32 rcheck;
  i32 rbeam;
  i32 max_input_bytes;
  i32 nscorers;          // 1: perceptron, 2: perceptron + RNN
  float w_perceptron;    // ScorerDef::scoreWeights
  float w_rnn;
};

// One lattice node (reference NodeInfo, src/core/core_types.h:71-90)
struct NodeInfo {
  i32 eptr;     // dictionary EntryPtr raw value; for UNK nodes ~(per-sentence unk index)
  u16 start;
  u16 end;
};

// UNK side data (reference UnkNodeHeader + placeholders, src/core/analysis/extra_nodes.h:25-47)
struct NodeAux {
  i32 tmpl;     // template EntryPtr raw (normalized nodes: the dictionary entry)
  i32 hash;     // content hash (negative) or 0 for dictionary nodes
  u16 ph0;      // placeholder 0 (jumandic: nonstdSurf = charlattice Modifiers bits)
  u16 ph1;      // placeholder 1 (jumandic: notPrefix 0/1)
  u16 maker;    // index into DevModel::makers
  u16 pad;
};

// A node the trainer adds to the seeds of a sentence (TrainingExampleAdapter::makeUnkTrainingNode,
// src/core/training/gold_example.cc:118-136): an UNK node with template EntryPtr 0 whose entry row is given.
// NodeAux of such a node: maker == kGoldMaker, pad = its index among the sentence's extra seeds.
struct ExtraSeed {
  u16 start, end;
  i32 hash;
  i32 row[8];
};
constexpr u16 kGoldMaker = 0xffff;

// One beam slot: the index form of ConnectionBeamElement
// (src/core/analysis/lattice_config.h:37-79).  `prev_node` (sentence-local
// node id of the left node) replaces the host pointer `previous`.
// What the count pass (k_seeds<0>) learned walking the double array from one start: which prefixes are
// trie nodes / dictionary keys and the entry-list pointers of the keys.  The emit passes replay it instead
// of walking again (one 48-byte read instead of a chain of dependent 4-byte loads per input byte).
constexpr int kWalkCacheVals = 8;
struct WalkCache {
  u64 leaf;                  // bit (len - 1): the prefix of len codepoints is a key
  i32 vals[kWalkCacheVals];  // trie values of the first keys, by increasing length
  u8 ok_len;                 // prefixes up to this length are trie nodes
  u8 cached;                 // 1: complete (walk <= 64 codepoints, <= kWalkCacheVals keys)
  u8 pad[6];
};

struct BeamSlot {
  u16 left;       // index into ends[boundary]
  u16 beam;       // slot index inside the left node's beam
  float total;
  u32 prev_node;
  u32 pad;
};
constexpr u16 kFake16 = 0xffff;

struct GbeamEntry {
  u16 left;
  u16 beam;
  float score;
};

// charlattice extra nodes per input position (reference CharLattice::Parse,
// src/core/analysis/charlattice.cc:197-264)
struct ClNodes {
  u32 cp[3];
  u16 type[3];
  u16 n;
};

// All device arrays of one batch.  Index spaces:
//   g   = byte_off[s] + s + i      codepoint i of sentence s   (arrays sized total_bytes + n)
//   bb  = byte_off[s] + 4*s + b    boundary b of sentence s    (arrays sized total_bytes + 4*n)
//   gn  = node_base[s] + k         node k of sentence s        (arrays sized total_nodes)
// sentence-local node ids: 0 = BOS (boundary 0), 1 = BOS (boundary 1),
// 2.. = lattice nodes ordered by (start, seed order), last = EOS.
// one rnn node in hidden-state row order (rows of a sentence: 0 parking, 1 BOS state, then the rnn nodes in boundary order)
struct RnnRec {
  u32 q;         // handle: boundary * G + index within the boundary
  i32 id;        // word id (RnnIdContainer::resolveId)
  u32 prevrow;   // row of the predecessor (0 for BOS / the parking row)
  u32 len;       // codepoints of the node (unkLengthPenalty)
};

// gstats[kGstatOverflow]: the batch did not fit the capacity it was enqueued against (k_lattice.h: k_cap_guard)
constexpr int kGstatOverflow = 8;

struct Batch {
  // input
  const u8* text;
  const u32* byte_off;     // [n+1]
  u32 n_sent;
  u32 total_bytes;
  // decode
  u32* cp_code;
  i32* cp_class;
  u16* cp_boff;            // byte offset of codepoint i inside the sentence; entry [ncp] = byte length
  ClNodes* cl_nodes;
  u32* sent_ncp;
  i32* sent_status;
  u32* sent_flags;         // bit0: charlattice applicable, bit1: stage-2 makers active
  // seeds
  u16* pos_cnt1;           // dictionary + stage-1 maker nodes (w/o normalize) starting at position g
  u16* pos_cntN;           // normalize-maker nodes starting at position g
  u64* pos_norm;           // [cp][kNormCache] NormResult of the count pass (starts with at most kNormCache results)
  u16* pos_cnt2;           // stage-2 maker nodes starting at position g
  WalkCache* pos_walk;     // the count pass's dictionary walk from position g, replayed by the emit passes
  u64* pos_ends;           // bit e set: a stage-1 node starting at position g ends at codepoint e (e <= 63), from the count pass
  u8* reach;               // [g] connectivity scratch
  u32* sent_nodes;         // nodes of sentence incl. 2 BOS + EOS (after stage decision)
  u32* sent_nodes2;        // node count with stage 2 (scratch for the relocation scan)
  u64* node_base;          // [n+1] exclusive scan of sent_nodes
  u64* node_base2;         // [n+1] relocation offsets of stage-2 sentences
  // lattice
  u32* bnd_first;          // [bb] first local node id starting at boundary
  u32* bnd_cnt;            // [bb] R_b
  u32* end_first;          // [bb] offset into end_nodes (relative to node_base[s])
  u32* end_cnt;            // [bb] L_b
  u32* end_nodes;          // [gn] local node ids
  NodeInfo* node_info;     // [gn]
  NodeAux* node_aux;       // [gn]
  i32* node_entry;         // [gn][row_stride]: the entry row of the node (8 columns, 16 for models with more than 8 feature columns)
  u32 row_stride;
  u64* node_pat;           // [gn][14]
  float* node_t0;          // [gn]
  BeamSlot* node_beam;     // [gn][beam]
  float* node_cells;       // [gn][gbeam][nscorers]
  u32* rnn_conn;           // [bb][gbeam] connection of EOS path p at boundary b: node (26 bits) | slot<<26, or ~0
  i32* rnn_id;             // [bb][gbeam] RNN vocabulary id of the connection's lattice node, at the first path through it (k_rnn_prep)
  u32* rnn_gi;             // [bb][gbeam] global-beam index of the connection (= which score cell of its node it owns) | codepoints of its lattice node << 16
  u32* rnn_assign;         // [bb][gbeam] rnn node (index within boundary) a connection is scored with
  u32* rnn_prev;           // [bb][gbeam] rnn node -> prev rnn node handle (b * G + idx)
  i32* rnn_nid;            // [bb][gbeam] word id of the rnn node
  u32* rnn_nlen;           // [bb][gbeam] codepoint length of the rnn node
  u32* rnn_cnt;            // [bb] rnn nodes per boundary
  u32* rnn_order;          // [n_sent] sentences grouped by the length of their recurrence (k_rnn_order_*), or null
  u32* rnn_key;            // [n_sent] that length, capped
  u32* rnn_offs;           // [kRnnOrderBins] next free slot of every length class
  u32* rnn_hist;           // [kRnnOrderBins] sentences per length class (all zero between batches)
  u32* rnn_slow;           // [2] first slot and number of the sentences of the last class (not staged in LDS)
  RnnRec* rnn_rec;         // [row] the rnn node of every hidden-state row (k_rnn_dense): what the recurrence and the
                           // scoring of a sentence of ANY length read, 64 records per load
  float* rnn_rscore;       // [row] RNN score of the row's rnn node (k_rnn_score_long)
  float* rnn_ctx;          // [row][EP] hidden states: row rnn_rowbase[s] + rnn_noff[b] + idx holds rnn node idx of boundary b
                           //   (per sentence: row 0 parking, row 1 the BOS state, then its rnn nodes in boundary order)
  u32* rnn_noff;           // [bb] first row of the boundary's rnn nodes within the sentence's rows (k_rnn_prep)
  u32* rnn_rows;           // [n_sent] rows of the sentence (k_rnn_prep)
  u64* rnn_rowbase;        // [n_sent + 1] exclusive scan of rnn_rows
  u8* node_kept;           // [gn]
  GbeamEntry* bnd_gbeam;   // [bb][gbeam]
  // partial annotation (ScorePlugin): CSR constraint arrays and the resulting per-node penalty (null: off)
  const u32* pc_nb_off;
  const u16* pc_nb;
  const u32* pc_b_off;
  const u16* pc_b;
  const u32* pc_node_off;
  const PcNode* pc_nodes;
  const PcTag* pc_tags;
  float* node_penalty;     // [node] 0, 1000 or 10000
  const float* pair_penalty;   // per (boundary, left, right) amount of a per-connection ScorePlugin, or null
  const u64* pair_base;        // [bb] start of the boundary's L x R matrix in pair_penalty
  // k_sweep<*, 0> only (a boundary with more right nodes than the LDS variants stage): per-sentence slice
  // for the prescores, their sums and the cutoff order
  // normalize maker: a start with more results / traversal states than the per-lane arrays hold (kMaxNormResults /
  // kMaxNormStates) repeats its traversal in an HBM slice: norm_slots groups of 64 (one slice per lane of a wavefront,
  // group = workgroup index mod norm_slots), norm_locks[group * 64 + lane]
  unsigned char* norm_scratch;
  u32* norm_locks;
  u32 norm_slots;
  // full-beam sweep: boundaries with more live candidates than its LDS staging holds take one of `full_slots` HBM
  // slices of `full_cap` candidates each (full_locks[slot]: 0 free / 1 taken)
  unsigned char* full_scratch;
  u32* full_locks;
  u32 full_slots;
  u32 full_cap;
  unsigned char* sweep_scratch;
  u64 sweep_scratch_stride;
  u32 sweep_scratch_maxr;
  BndMeta* bnd_meta;       // [bb] {bnd_first, bnd_cnt, end_first, end_cnt} packed for k_sweep (written by k_ends)
  u32* bnd_ngb;            // [bb]
  // result
  u32* path_len;           // [n]
  u32* path_nodes;         // [gn] top-1 path (local node ids, EOS side first), at node_base[s]
  u32* gstats;             // [8] batch statistics: [0] max right nodes at one boundary, [1..3] sentences per sweep class
  u32* sent_maxr;          // [n] widest boundary (right nodes) of the sentence (k_layout)
  u32* sweep_list;         // [3][n] sentence indices by sweep class (k_sweep_classify)
  // gold nodes injected by the trainer (jppgpu_analyze_batch_seeds): CSR over the sentences; null otherwise
  const u32* gold_off;     // [n + 1]
  const ExtraSeed* gold;   // [gold_off[n]]
  u64 total_nodes;
};

}  // namespace jpp

#endif  // JPP_TYPES_H
