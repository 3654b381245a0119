/*
 * jppgpu -- C ABI of the MI355X-native Juman++ analysis hot path.
 *
 * This is the drop-in boundary: what the reference's host code (C++14) binds
 * instead of running Analyzer::analyze on the CPU.  Plain C types only; the
 * caller owns inputs, the library owns results until jppgpu_result_release.
 *
 * Reference interfaces replaced (paths relative to the ku-nlp/jumanpp tree):
 *   Analyzer::initialize(CoreHolder*, AnalyzerConfig, ScoringConfig, ScorerDef*)
 *       src/core/analysis/analyzer.cc:16-43, analyzer_impl.cc:27-89   -> jppgpu_ctx_create
 *   Analyzer::analyze(StringPiece, ScorePlugin*)
 *       src/core/analysis/analyzer.h:51, analyzer.cc:45-53             -> jppgpu_analyze_batch*
 *   Lattice / LatticeBoundary / ConnectionBeamElement read by formatters
 *       src/core/analysis/lattice_types.h, lattice_config.h:37-79      -> jppgpu_result views
 *   ExtraNodesContext::node(EntryPtr) (UNK nodes)
 *       src/core/analysis/extra_nodes.h:79-87                           -> jppgpu_node.unk_*
 *   AnalysisPath::fillIn + OutputManager::locate (what the 1-best formats read)
 *       src/core/analysis/analysis_result.cc:25-76, output.cc:69-111     -> jppgpu_result_fetch(JPPGPU_FETCH_TOP1)
 *   LatticeFormatInfo::fillInfo (what the N-best lattice format reads)
 *       src/jumandic/shared/lattice_format.cc:13-43,129-141             -> jppgpu_result_fetch_nbest
 *   LatticeFormat::format (the N-best lattice text itself)
 *       src/jumandic/shared/lattice_format.cc:83-242                    -> jppgpu_result_format_lattice
 *   ScorePlugin (partial annotation)  src/core/analysis/score_plugin.h:14-19,
 *       src/core/input/partial_example.cc                                -> jppgpu_analyze_batch_partial
 *   ScorePlugin::updateScore as an extension point (any right-node-dependent plugin)
 *       src/core/analysis/score_plugin.h:14-19, score_processor.cc:578-613 -> jppgpu_analyze_batch_plugin
 *   AnalyzerImpl::setGlobalBeam / autoBeamSizes  analyzer_impl.cc:311-361 -> jppgpu_ctx_set_beams
 *   Status kinds returned by analyze()
 *       analysis_input.cc:12-33, characters.cc:267-269, analyzer_impl.cc:133-135 -> status codes
 */
#ifndef JPPGPU_H
#define JPPGPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* return codes of API calls (mirror of jumanpp::StatusCode, src/util/status.hpp) */
enum {
  JPPGPU_OK = 0,
  JPPGPU_INVALID_PARAMETER = 1,
  JPPGPU_INVALID_STATE = 2,
  JPPGPU_NOT_IMPLEMENTED = 3,
  JPPGPU_NO_DEVICE = 4,
  JPPGPU_OUT_OF_MEMORY = 5
};

/* per-sentence status (jppgpu_result.status[i]) */
enum {
  JPPGPU_SENT_OK = 0,
  JPPGPU_SENT_TOO_LONG = 1,   /* input > max_input_bytes: InvalidParameter in the reference */
  JPPGPU_SENT_BAD_UTF8 = 2,   /* InvalidParameter */
  JPPGPU_SENT_NO_LATTICE = 3, /* InvalidState "could not build lattice" */
  JPPGPU_SENT_CAPACITY = 4    /* a device-side staging capacity was exceeded (see DESIGN.md limits) */
};

typedef struct jppgpu_ctx jppgpu_ctx;
typedef struct jppgpu_result jppgpu_result;

/* One UNK maker (spec::UnkProcessorDescriptor, src/core/spec/spec_types.h) */
typedef struct {
  int32_t type;           /* spec::UnkMakerType: 1 single, 2 chunking, 3 onomatopoeia, 4 numeric, 5 normalize */
  int32_t char_class;     /* chars::CharacterClass mask */
  int32_t pattern_ptr;    /* EntryPtr raw of the template dictionary entry */
  int32_t priority;       /* 0: stage 1, 1: stage 2 (lowPriority) */
  int32_t placeholder;    /* target placeholder or -1 */
  uint32_t replace_mask;  /* bit f: entry feature f receives the surface hash (outputTo) */
} jppgpu_unk_maker;

/* Model blobs, borrowed for the duration of jppgpu_ctx_create (copied to HBM).
 * They are exactly the blocks of the reference's .jppmdl parts
 * (src/core/dic/dic_builder.cc:99-107, src/core/analysis/perceptron.cc:64-123). */
typedef struct {
  const void* trie;            /* dictionary part data[1]: darts-clone units (u32[]) */
  size_t trie_bytes;
  const void* entry_ptrs;      /* data[2]: varint entry-pointer lists */
  size_t entry_ptrs_bytes;
  const void* entry_data;      /* data[3]: varint entry rows */
  size_t entry_data_bytes;
  const float* weights;        /* perceptron part data[1]: float[2^weight_exponent] */
  uint32_t weight_exponent;
  int32_t num_features;        /* spec.features.numDicFeatures */
  int32_t num_placeholders;    /* spec.features.numPlaceholders */
  const jppgpu_unk_maker* unk_makers; /* spec.unkCreators, in spec order */
  int32_t num_unk_makers;
  const void* feature_spec;    /* flattened FeaturesSpec descriptors (i32 lists: primitive, computed, pattern and n-gram
                                * features in spec order).  Equal to the built-in jumandic tables: the compiled-in
                                * kernels run (the reference's generated static code, features_api.cc:38-47); any
                                * other spec within the device layout (entry rows of up to 16 columns, 2 placeholders):
                                * the table-driven kernels with the summation
                                * orders of the reference's dynamic feature objects; outside it: JPPGPU_NOT_IMPLEMENTED
                                * with the reason in jppgpu_last_error() */
  size_t feature_spec_bytes;
  /* optional RNN re-ranker = blocks of the model's Rnn part
   * (src/core/analysis/rnn_scorer_gbeam.cc:375-398): data[1..6] + decoded header */
  int32_t has_rnn;
  const void* rnn_known_index;   /* data[1] double array: word repr -> id (dictionary nodes) */
  size_t rnn_known_index_bytes;
  const void* rnn_unk_index;     /* data[2] double array (UNK nodes) */
  size_t rnn_unk_index_bytes;
  const float* rnn_matrix;       /* data[3] W[E x E] column-major: out[i] = sum_k W[i*E+k] ctx[k] */
  const float* rnn_embeddings;   /* data[4] [V x E] */
  const float* rnn_nce_embeddings; /* data[5] [V x E] */
  const float* rnn_maxent;       /* data[6] [maxent_size] */
  uint32_t rnn_layer_size;       /* E */
  uint32_t rnn_maxent_order;
  uint64_t rnn_maxent_size;
  uint64_t rnn_vocab_size;
  float rnn_nce_constant;        /* effective MikolovRnn::rnnNceConstant */
  int32_t rnn_unk_id;
  float rnn_unk_constant;        /* RnnInferenceConfig::unkConstantTerm */
  float rnn_unk_length;          /* RnnInferenceConfig::unkLengthPenalty */
  uint32_t rnn_num_fields;       /* RnnIdResolver::fields_ (entry feature indices) */
  uint32_t rnn_fields[8];
} jppgpu_model;

/* The value storage of one dictionary column (dic::FieldsHolder / DictionaryField, src/core/dic/dictionary.h:19-58):
 * what the LENGTH primitives of a feature spec read (ByteLength / CodepointSize, src/core/impl/feature_impl_prim.h:114-156,
 * PrimitiveFeatureContext::lengthOf, feature_impl_types.h:156-174).  kind 1: a string storage (length-prefixed strings,
 * the column's value << align_power = byte offset); kind 2: an int-list storage (the column's value = byte offset of a
 * varint count: string-list columns, "positions").  Borrowed for the duration of jppgpu_ctx_create. */
typedef struct {
  int32_t column;        /* entry-row column (DictionaryField::idxInEntry >= 0) */
  int32_t kind;          /* 1 strings, 2 int lists */
  uint32_t align_power;
  uint32_t reserved;
  const void* data;
  uint64_t bytes;
} jppgpu_field_storage;

/* AnalyzerConfig + ScoringConfig subset (src/core/analysis/analyzer.h:15-27,
 * defaults of the CLI: src/jumandic/shared/jumanpp_args.h:50-54) */
typedef struct {
  uint32_t struct_size;     /* = sizeof(jppgpu_config) of the header the CALLER was compiled against (JPPGPU_CONFIG_INIT sets
                             * it).  The struct only ever grows at its end: the library reads the first struct_size bytes
                             * and takes the defaults (0) for fields the caller's header did not have; a size it never
                             * shipped (smaller than JPPGPU_CONFIG_MIN_SIZE, larger than its own with non-zero unknown
                             * tail, not a multiple of 4) is JPPGPU_INVALID_PARAMETER.  ABI break of round 4: up to
                             * round 3 the struct started with `beam` and had no size field. */
  int32_t beam;             /* 5.  Beyond 32 only with a global beam: the result's beams then have 32 slots per node (what
                             * lies behind the global_beam-th slot of a node is fake in the reference as well) */
  int32_t global_beam;      /* 6; <= 32.  0: full-beam scoring (beam <= 32) */
  int32_t right_check;      /* 1 */
  int32_t right_beam;       /* 5 */
  int32_t max_input_bytes;  /* 4096 */
  int32_t device;           /* HIP device ordinal */
  int32_t use_rnn;          /* 1: run the RNN scorer (ScorerDef::others), needs model.has_rnn */
  float weight_perceptron;  /* ScorerDef::scoreWeights[0] (used only with use_rnn) */
  float weight_rnn;         /* ScorerDef::scoreWeights[1] */
  int32_t dynamic_features; /* 1: score with the summation orders of the reference's table-driven (dynamic) feature
                             * objects even when the spec is the compiled-in one -- what its trainer runs
                             * (TrainingEnv::initFeatures(nullptr), src/jumandic/main/jumanpp_train.cc:206,
                             * src/core/features_api.cc:20-60).  0: static code when the spec matches (the analyser). */
  /* ScorerDef::others beyond the model's RNN (score_api.h:44-72): scorers that live on the HOST and fill their slot of
   * the score cells through jppgpu_analyze_batch_scored.  Scorer slots: 0 perceptron, 1 RNN (when use_rnn), then the
   * host scorers in order; at most 4 in all.  Needs a global beam, like every extra scorer (analyzer_impl.cc:81-86). */
  int32_t num_host_scorers; /* 0 .. 2 */
  float weight_host[2];     /* ScorerDef::scoreWeights of the host scorers */
  /* (round 5) What jppgpu_ctx_create DERIVES from the model -- the per-dictionary-entry T0 records of k_t0_memo: a walk
   * over the whole trie and 30-odd weight gathers per entry, 0.3-0.6 s for a 10^6-entry dictionary -- may be handed in
   * instead, as jppgpu_ctx_t0_memo_image returned it for this very model (same file, same weights) and this library
   * (jppgpu_t0_memo_format).  The reference re-reads and re-derives its model on every start as well
   * (src/core/impl/model_io.cc:115-176); jumanpp_gpu keeps the image beside the .jppmdl, keyed by size + mtime.
   * A context made from an image drops its T0 records when jppgpu_ctx_set_weights replaces the table. */
  const void* t0_memo_image;
  uint64_t t0_memo_image_bytes;
  uint32_t t0_memo_slots;
  int32_t keep_t0_memo_image;   /* 1: keep a host copy of the records for jppgpu_ctx_t0_memo_image */
  /* (round 5) the column storages a spec with length primitives needs (any others are ignored); without them such a
   * spec is JPPGPU_NOT_IMPLEMENTED */
  const jppgpu_field_storage* field_storages;
  uint32_t num_field_storages;
  uint32_t reserved1;
} jppgpu_config;
#define JPPGPU_CONFIG_MIN_SIZE 44u   /* struct_size .. dynamic_features: the first layout that carried a size */
#define JPPGPU_CONFIG_INIT {(uint32_t)sizeof(jppgpu_config)}

/* EntryPtr::BOS() / EntryPtr::EOS() raw values (src/core/core_types.h:44-58) */
#define JPPGPU_ENTRY_BOS ((int32_t)0x80000000)
#define JPPGPU_ENTRY_EOS ((int32_t)0x80000002)

typedef struct {
  int32_t entry_ptr;   /* EntryPtr raw: >=0 dictionary, BOS/EOS specials, otherwise ~(unk ordinal) */
  uint16_t start;      /* codepoint span */
  uint16_t end;
} jppgpu_node;

typedef struct {
  int32_t template_ptr; /* UNK: template EntryPtr raw; 0 for dictionary nodes */
  int32_t content_hash; /* UNK: surface hash (negative) */
  uint16_t placeholder[2];
  uint16_t maker;          /* UNK: index into jppgpu_model::unk_makers (spec order) */
  uint16_t pad;
} jppgpu_unk;

typedef struct {
  uint16_t left;        /* index into the ends list of the node's boundary */
  uint16_t beam;        /* slot in the previous node's beam */
  float total;
  uint32_t prev_node;   /* sentence-local node id of the previous node, 0xffffffff for BOS */
  uint32_t pad;
} jppgpu_beam_slot;     /* fake slot: left == beam == 0xffff */

/* Host-side view of one analysed batch.  Index spaces:
 *   node k of sentence i lives at node_base[i] + k   (0,1 = BOS, last = EOS)
 *   boundary b of sentence i lives at bnd_base[i] + b
 * Arrays marked (debug) are only filled when the batch was fetched with full=1. */
typedef struct {
  uint32_t n_sentences;
  const int32_t* status;         /* [n] */
  const uint32_t* n_codepoints;  /* [n] */
  const uint32_t* n_nodes;       /* [n] */
  const uint64_t* node_base;     /* [n] */
  const uint64_t* bnd_base;      /* [n] */
  uint64_t total_nodes;
  uint64_t total_boundaries;
  int32_t beam, global_beam;         /* beam = slots per node in `beams` (min(configured beam, 32) with a global beam) */
  int32_t num_scorers;
  int32_t entry_row_stride;          /* columns per row of entry_rows: 8, or 16 for models with more than 8 feature columns
                                      * (0 from the compact JPPGPU_FETCH_TOP1 view, which has no rows) */
  /* top-1 path: node ids from EOS back to the first morpheme, path_len[i] entries at node_base[i] */
  const uint32_t* path_len;
  const uint32_t* path_nodes;
  const jppgpu_node* nodes;          /* [total_nodes] */
  const jppgpu_unk* unk;             /* [total_nodes] */
  /* lattice detail (full=1) */
  const uint32_t* bnd_first;         /* [total_boundaries] first node starting at boundary */
  const uint32_t* bnd_count;         /* R_b */
  const uint32_t* end_first;         /* offset into end_nodes (relative to node_base) */
  const uint32_t* end_count;         /* L_b */
  const uint32_t* end_nodes;         /* [total_nodes] */
  const int32_t* entry_rows;         /* [total_nodes][entry_row_stride] */
  const uint64_t* patterns;          /* [total_nodes][14] */
  const float* t0_scores;            /* [total_nodes] */
  const jppgpu_beam_slot* beams;     /* [total_nodes][beam] */
  const float* cells;                /* [total_nodes][global_beam][num_scorers] */
  const uint8_t* kept;               /* [total_nodes] */
  const uint32_t* gbeam_count;       /* [total_boundaries] */
  const uint32_t* gbeam;             /* [total_boundaries][global_beam][2]: (left | beam<<16), score bits */
} jppgpu_result_view;

/* Threading: like the reference (one Analyzer per thread over a shared read-only JumanppEnv), a context
 * is used by one thread at a time.  Different contexts are independent -- each owns its model copy, its
 * workspaces and the HIP stream its host-buffer entry points run on -- and may be driven from different
 * threads; jppgpu_last_error is per thread. */
int jppgpu_ctx_create(const jppgpu_model* model, const jppgpu_config* config, jppgpu_ctx** out);
void jppgpu_ctx_destroy(jppgpu_ctx* ctx);
/* Changes the beam configuration of the following batches; the model stays resident.
 * AnalyzerImpl::setGlobalBeam + the per-sentence beam of auto-beam mode
 * (src/core/analysis/analyzer_impl.cc:311-331,350-361).  Same validation as jppgpu_ctx_create. */
/* A further context on the same device that shares the base context's copy of the model in HBM (dictionary blobs, weight
 * table, RNN tables, per-entry T0 records, format table); workspaces, streams and configuration are its own.  What a
 * second Analyzer over the same CoreHolder is in the reference (src/core/env.cc:109-121: makeAnalyzer).  The shared
 * tables are freed with the last context using them; jppgpu_ctx_set_weights / jppgpu_ctx_set_format_table on either
 * context acts on the shared copy: such a call waits for everything enqueued on the device (the sibling contexts'
 * batches included) before it touches the tables, writers exclude each other, and the caller must not start a batch
 * on a sibling context while the call runs. */
int jppgpu_ctx_create_shared(jppgpu_ctx* base, const jppgpu_config* config, jppgpu_ctx** out);

/* Buffers at their final size when the analyzer is made (round 5).  The reference grows its per-sentence arena while it
 * analyses (src/jumandic/main/jumanpp.cc:156-179 is the loop this replaces: one analyze() per line, memory taken as it
 * goes); a batched context knows the size of its batches beforehand -- jumanpp_gpu mmaps its input and fixes --batch --
 * and takes everything at once: after jppgpu_ctx_reserve a batch within the reserved size allocates nothing and is ONE
 * enqueue (no host wait between the decode kernel and the sweep; the totals that used to size the node tables, the
 * lattice arrays and the hidden-state rows are compared with the capacity on the device, jppgpu_ctx_stats counts the
 * batches that did not fit and were run again the sized way). */
typedef struct jppgpu_reserve {
  uint32_t struct_size;        /* sizeof(jppgpu_reserve) */
  uint32_t max_sentences;      /* sentences per batch */
  uint64_t max_total_bytes;    /* input bytes per batch */
  float nodes_per_byte;        /* lattice nodes per input byte to provide for; 0: the library's default (3.0) */
  float text_bytes_per_byte;   /* jppgpu_result_format_top1 output bytes per input byte; 0: no text buffers */
  uint32_t text_host_blocks;   /* page-locked host blocks of that size kept ready (results in flight at once) */
  uint32_t reserved;
} jppgpu_reserve;
/* the derived T0 records of a context made with keep_t0_memo_image (host memory owned by the context's model copy, valid
 * until its last context is destroyed; jppgpu_ctx_set_weights refills the same storage in place -- the pointer stays
 * valid, the records must not be read while that call runs); *bytes = 0 when the context has none.  jppgpu_t0_memo_format: changes whenever
 * the record layout or its arithmetic does -- part of a cache key. */
int jppgpu_ctx_t0_memo_image(jppgpu_ctx* ctx, const void** data, uint64_t* bytes, uint32_t* slots);
uint64_t jppgpu_t0_memo_format(void);
int jppgpu_ctx_reserve(jppgpu_ctx* ctx, const jppgpu_reserve* r);
/* `count` page-locked host blocks of `bytes` each, pinned on the calling thread and left where the text pools of the
 * contexts (jppgpu_result_format_top1) take them from.  Meant for a thread of its own at process start: page-locking a
 * few hundred MB takes longer than a batch. */
int jppgpu_host_prepin(int32_t device, uint64_t bytes, uint32_t count);

typedef struct jppgpu_ctx_statistics {
  uint32_t struct_size;            /* sizeof(jppgpu_ctx_statistics), set by the caller */
  uint32_t reserved;
  uint64_t one_enqueue_batches;    /* batches enqueued against the held capacity, no host wait inside */
  uint64_t one_enqueue_overflows;  /* ... of which did not fit and were run again */
  uint64_t sized_batches;          /* batches run with the three sizing waits (first batch of an unreserved context,
                                      host-callback entry points, full-beam scoring, overflows) */
  uint64_t device_allocations;     /* device (re)allocations of the whole process so far */
} jppgpu_ctx_statistics;
int jppgpu_ctx_stats(jppgpu_ctx* ctx, jppgpu_ctx_statistics* out);
int jppgpu_ctx_set_beams(jppgpu_ctx* ctx, int32_t beam, int32_t global_beam, int32_t right_check,
                         int32_t right_beam);
const char* jppgpu_last_error(void);

/* Analyse n sentences given as one UTF-8 buffer + n+1 byte offsets (host memory). */
int jppgpu_analyze_batch(jppgpu_ctx* ctx, const char* utf8, const uint32_t* offsets, uint32_t n,
                         jppgpu_result** out);
/* Same, but text/offsets are already resident in device memory (HBM) and the
 * work is enqueued on `stream` (a hipStream_t, may be NULL).  total_bytes =
 * offsets[n].  Results stay on the device until fetched. */
/* Partial annotation = the reference's only ScorePlugin (src/core/analysis/score_plugin.h:14-19,
 * PexStreamReaderImpl::updateScore src/core/input/pex_stream_reader.cc:24-39, PartialExample::checkViolation
 * src/core/input/partial_example.cc:23-73): a connection whose right node violates a constraint loses 1000
 * (tag mismatch) or 10000 (word start/end on a no-break position, a required boundary inside the node,
 * wrong length) from its score.  Constraints are CSR arrays over the sentences of the batch;
 * boundaries count from 2 (two BOS boundaries) like LatticeNodePtr::boundary. */
typedef struct {
  uint16_t boundary;  /* NodeConstraint::boundary */
  uint16_t length;    /* codepoints */
  uint32_t tag_first; /* index of the first tag in jppgpu_partial::tags */
  uint32_t tag_count;
} jppgpu_node_constraint;
typedef struct {
  int32_t field;      /* TagConstraint::field: entry-row column */
  int32_t value;      /* string pointer of the value, or hashUnkString(value) when the value is not in the dictionary */
} jppgpu_tag_constraint;
typedef struct {
  const uint32_t* nobreak_offsets;   /* [n + 1] */
  const uint16_t* nobreak;           /* PartialExample::noBreak_, ascending per sentence */
  const uint32_t* boundary_offsets;  /* [n + 1] */
  const uint16_t* boundaries;        /* PartialExample::boundaries_, ascending per sentence */
  const uint32_t* node_offsets;      /* [n + 1] */
  const jppgpu_node_constraint* nodes;
  const jppgpu_tag_constraint* tags;
  uint32_t num_tags;
} jppgpu_partial;

/* Analyzer::analyze(input, ScorePlugin*) with the partial-annotation plugin (analyzer.cc:45-53) */
int jppgpu_analyze_batch_partial(jppgpu_ctx* ctx, const char* utf8, const uint32_t* offsets, uint32_t n,
                                 const jppgpu_partial* partial, jppgpu_result** out);

/* The ScorePlugin extension point itself (src/core/analysis/score_plugin.h:14-19): the reference calls
 * ScorePlugin::updateScore(lattice, connection, &score) for every connection it scores
 * (applyPluginToPrescores / applyPluginToGbeam, score_processor.cc:578-613).  A device kernel cannot call
 * back into host code per connection; the batched form of the hook is: the lattice is built (nodes, UNK
 * records, entry rows), the plugin sees it once per batch on the host and fills `penalty[node]`, and that
 * amount is subtracted from the score of EVERY connection into the node, at the two places the reference
 * applies its plugin.  This covers every plugin whose adjustment depends on the right node of the connection
 * only -- the partial-annotation plugin above is one (PartialExample::checkViolation looks at nothing else)
 * and jppgpu_analyze_batch_partial is its device-side specialisation.  With global_beam == 0 the hook has no
 * effect, as in the reference (analyzer_impl.cc:236-238). */
typedef struct {
  uint32_t n_sentences;
  int32_t num_features;            /* width of an entry row */
  const int32_t* status;           /* [n_sentences] JPPGPU_SENT_* so far (failed sentences have no nodes) */
  const uint32_t* n_codepoints;    /* [n_sentences] */
  const uint32_t* n_nodes;         /* [n_sentences] 0/1 = BOS, last = EOS */
  const uint64_t* node_base;       /* [n_sentences] */
  uint64_t total_nodes;
  const jppgpu_node* nodes;        /* [total_nodes] */
  const jppgpu_unk* unk;           /* [total_nodes] */
  const int32_t* entry_rows;       /* [total_nodes][num_features] */
} jppgpu_lattice_nodes;
typedef void (*jppgpu_score_plugin_fn)(void* user, const jppgpu_lattice_nodes* lattice, float* penalty /* [total_nodes], zeroed */);
int jppgpu_analyze_batch_plugin(jppgpu_ctx* ctx, const char* utf8, const uint32_t* offsets, uint32_t n,
                                jppgpu_score_plugin_fn plugin, void* user, jppgpu_result** out);

/* ScorePlugin::updateScore(lattice, ConnectionPtr, &score) (score_plugin.h:14-19; call sites score_processor.cc:578-613)
 * per CONNECTION, batched: the plugin sees the built lattice of the batch -- nodes, the right nodes of every boundary and
 * its ends list (the left nodes) -- and says for every (left node, right node) pair of every boundary what the score of
 * a connection between the two loses:  penalty[pair_base[bb] + left * bnd_count[bb] + right], left = index into the
 * boundary's ends list, right = index among the nodes starting there (ConnectionPtr::left / ::right), bb = bnd_base[i] + b.
 * Applied where applyPluginToPrescores / applyPluginToGbeam act, as ONE subtraction per connection.  What the batched
 * form cannot express: an amount that depends on the beam slot or the history (ConnectionPtr::beam / ::previous). */
typedef struct {
  uint32_t n_sentences;
  int32_t num_features;
  const int32_t* status;
  const uint32_t* n_codepoints;
  const uint32_t* n_nodes;
  const uint64_t* node_base;
  uint64_t total_nodes;
  const jppgpu_node* nodes;
  const jppgpu_unk* unk;
  const int32_t* entry_rows;      /* [total_nodes][num_features] */
  const uint64_t* bnd_base;       /* [n]; sentence i has n_codepoints[i] + 3 boundaries */
  uint64_t total_boundaries;
  const uint32_t* bnd_first;      /* [total_boundaries] first node starting at the boundary (sentence-local) */
  const uint32_t* bnd_count;      /* right nodes */
  const uint32_t* end_first;      /* [total_boundaries] offset of the boundary's ends list in end_nodes (sentence-local) */
  const uint32_t* end_count;      /* left nodes */
  const uint32_t* end_nodes;      /* [total_nodes] at node_base[i]: sentence-local node ids */
  const uint64_t* pair_base;      /* [total_boundaries + 1] */
  uint64_t total_pairs;
} jppgpu_lattice_pairs;
typedef void (*jppgpu_connection_plugin_fn)(void* user, const jppgpu_lattice_pairs* lattice, float* penalty /* [total_pairs], zeroed */);
int jppgpu_analyze_batch_pairs(jppgpu_ctx* ctx, const char* utf8, const uint32_t* offsets, uint32_t n,
                               jppgpu_connection_plugin_fn plugin, void* user, jppgpu_result** out);

/* Trainer hook-up, first half: gold nodes.  The reference's trainer looks at the node seeds of a sentence after the
 * dictionary and UNK makers ran and before the lattice is built, and appends a seed for every node of the gold
 * analysis that is not among them (Trainer::prepare, src/core/training/trainer.cc:13-47;
 * TrainingExampleAdapter::ensureNodes / makeUnkTrainingNode, src/core/training/gold_example.h:88-109,
 * gold_example.cc:118-136).  The batched form: once the seeds of the whole batch exist on the device, `hook` sees them
 * on the host (in their final order: by start, then creation) and returns the seeds to add; they are inserted behind the
 * makers' seeds of their start position, sentences that received seeds are checked for connectivity again (a sentence the
 * makers left disconnected becomes analysable; one that stays disconnected gets JPPGPU_SENT_NO_LATTICE), and the
 * analysis continues.  An added node is an UNK node with template_ptr 0, the given content hash and entry row, zero
 * placeholders and maker == JPPGPU_GOLD_MAKER; its EntryPtr follows those of the makers' UNK nodes. */
#define JPPGPU_GOLD_MAKER 0xffff
typedef struct {
  uint32_t n_sentences;
  const int32_t* status;         /* [n] JPPGPU_SENT_OK or JPPGPU_SENT_NO_LATTICE: seeds present; anything else: none */
  const uint32_t* n_codepoints;  /* [n] */
  const uint32_t* n_seeds;       /* [n] */
  const uint64_t* seed_base;     /* [n] */
  const jppgpu_node* seeds;      /* entry_ptr < 0: an UNK node (not yet numbered), described by unk[] */
  const jppgpu_unk* unk;
} jppgpu_seed_view;
typedef struct {
  uint16_t start, end;           /* codepoint span */
  int32_t content_hash;          /* hashUnkString(surface) */
  int32_t row[8];                /* entry row, model.num_features values */
} jppgpu_extra_seed;
typedef struct {
  const uint32_t* offsets;       /* [n + 1] CSR over the sentences; NULL: nothing to add */
  const jppgpu_extra_seed* seeds;/* ascending start within a sentence */
} jppgpu_extra_seeds;
/* fills *out with arrays that stay valid until jppgpu_analyze_batch_seeds returns; non-zero return aborts the batch */
typedef int (*jppgpu_seed_hook_fn)(void* user, const jppgpu_seed_view* seeds, jppgpu_extra_seeds* out);
int jppgpu_analyze_batch_seeds(jppgpu_ctx* ctx, const char* utf8, const uint32_t* offsets, uint32_t n,
                               jppgpu_seed_hook_fn hook, void* user, jppgpu_result** out);

/* ScoreComputer::scoreLattice (score_api.h:54-59) for a scorer on the host: called once per batch when the lattice is
 * built, the perceptron cells are written and the beams hold the perceptron totals (with use_rnn: the RNN cells too).
 * `lattice` is the full view of the batch (jppgpu_result_fetch(JPPGPU_FETCH_FULL)): nodes, ends lists, global beams,
 * beams, cells.  The scorer writes ITS slot of the connections it scores:
 *     cells[((node_base[i] + node) * global_beam + gbeam_index) * num_scorers + scorer_idx]
 * (what the reference's scorer writes through scores->nodeScores(right).beamLeft(beam, left).at(scorerIdx));
 * cells it leaves alone read 0.  A non-zero return aborts the batch with JPPGPU_INVALID_STATE.  Afterwards the device
 * re-makes the beam totals and the EOS beam from the weighted cells (adjustBeamScores / remakeEosBeam, k_adjust.h). */
typedef int (*jppgpu_score_lattice_fn)(void* user, const jppgpu_result_view* lattice, uint32_t scorer_idx, float* cells);
int jppgpu_analyze_batch_scored(jppgpu_ctx* ctx, const char* utf8, const uint32_t* offsets, uint32_t n,
                                const jppgpu_score_lattice_fn* scorers, void* const* users, uint32_t n_scorers,
                                jppgpu_result** out);

int jppgpu_analyze_batch_device(jppgpu_ctx* ctx, const void* d_utf8, const void* d_offsets, uint32_t n,
                                uint32_t total_bytes, void* stream, jppgpu_result** out);
/* Copy results to the host.  full=0 (JPPGPU_FETCH_BASIC): status, node table, UNK table, top-1 paths.
 * full=1 (JPPGPU_FETCH_FULL): additionally the whole lattice (patterns, T0, beams, cells, global beams).
 * full=2 (JPPGPU_FETCH_TOP1): only what AnalysisPath::fillIn + OutputManager::locate read for the best
 *   analysis (analysis_result.cc:25-76, output.cc:69-111): the node and UNK tables hold just the nodes
 *   of each sentence's top-1 path, compacted on the device, in path order (EOS first).  In this view
 *   n_nodes[i] == path_len[i], node_base[i] is the sentence's offset in the compact tables and
 *   path_nodes[node_base[i] + k] == k; BOS/EOS are recognised by their entry_ptr, not their index.
 *   About 10x fewer bytes cross PCIe than with full=0. */
#define JPPGPU_FETCH_BASIC 0
#define JPPGPU_FETCH_FULL 1
#define JPPGPU_FETCH_TOP1 2
/* The arrays of a fetched view are host copies owned by the result: they stay valid until
 * jppgpu_result_release, also across later batches on the same context (only the device-resident
 * side -- further fetches, jppgpu_result_pack -- is invalidated by the next batch). */
int jppgpu_result_fetch(jppgpu_result* res, int full, jppgpu_result_view* view);

/* ---- output text on the device (replaces OutputFormat::format for the top-1 formats) ----------------------
 * The reference formats one sentence at a time on the host: JumanFormat::format walks the top-1 path, and for every node
 * prints strings of the dictionary entry (src/jumandic/shared/juman_format.cc:94-168, src/core/analysis/output.cc:65-130).
 * What it prints for a DICTIONARY node is a function of the entry alone, and for an UNK node the same text with the
 * surface-bearing fields replaced by the node's input bytes and a flag feature appended.  The host layer therefore
 * renders every entry row ONCE per model into a table of text pieces (host/format_table.cc), the library keeps it in
 * HBM, and two kernels per batch (k_fmt_count / k_fmt_write, csrc/k_format.h) assemble the text of all sentences; only
 * bytes and one offset per sentence cross PCIe.  The library knows nothing of JUMAN: every literal comes from the table.
 *
 * One row = one output line of a node (an alias entry has several: rows after the first start with their own prefix,
 * "@ " in JUMAN).  Its text lies in the blob exactly as a dictionary node prints it:
 *     PRE S ' ' R ' ' B MID TAIL      TAIL = '"' FEAT '"' '\n'  (has_features)  or  "NIL" '\n'
 * S, R, B are the three fields an UNK maker may replace by the input surface (jumandic: surface, reading, baseform);
 * an UNK node prints  PRE S' ' ' R' ' ' B' MID '"' FEAT [sep flag_label flags] '"' '\n'  where X' is the escaped input
 * surface when the node's maker replaces X, and the bracket appears when the node's flag placeholder is non-zero
 * (JumanFormat: formatNormalizedFeature, juman_format.cc:57-92; sep = ' ' iff FEAT is not empty). */
typedef struct {
  uint32_t blob_off;       /* first byte of the row text in `blob` */
  uint16_t len_pre, len_s, len_r, len_b;
  uint16_t len_mid;        /* from the byte after B to the byte before TAIL */
  uint16_t flags;          /* bit 0: has_features (TAIL is the quoted form), bit 1: last row of its entry */
  uint32_t len_feat;       /* FEAT bytes (inside the quotes; 0 when TAIL is "NIL") */
  uint32_t len_total;      /* whole row text incl. the newline */
} jppgpu_format_row;       /* 24 bytes */

typedef struct {
  uint32_t struct_size;    /* sizeof(jppgpu_format_table) */
  /* entry -> rows: slot = (EntryPtr raw >> 1) >> 3 (an entry row is at least 8 bytes long, so slots are unique);
   * value 0 = no entry starts in that slot, otherwise 1 + index of the entry's first row */
  const uint32_t* slot_first_row;
  uint64_t n_slots;
  const jppgpu_format_row* rows;
  uint64_t n_rows;
  const char* blob;
  uint64_t blob_bytes;
  /* per UNK maker (index as in jppgpu_model::unk_makers): bit 0 / 1 / 2 = S / R / B print the input surface */
  uint8_t maker_replaces[16];
  /* escapeForJumanOutput (juman_format.cc:42-54): an input surface of exactly one byte equal to escape_from[i] prints
   * as escape_to[i][0 .. escape_len[i]) */
  uint8_t n_escapes;
  char escape_from[4];
  uint8_t escape_len[4];
  char escape_to[4][8];
  /* flag feature of UNK nodes: placeholder index (0/1; < 0: none), label bytes, and per bit of the value one letter,
   * printed in table order when (value & flag_mask[i]) != 0 */
  int32_t flag_placeholder;
  uint8_t flag_label_len;
  char flag_label[32];
  uint8_t n_flags;
  uint32_t flag_mask[16];
  char flag_char[16];
  /* sentence frame: text after the last node ("EOS\n"), and the whole text of a sentence that failed ("# ERROR\nEOS\n") */
  uint8_t eos_len, error_len;
  char eos_text[16];
  char error_text[32];
} jppgpu_format_table;

/* copies the table to the device (once per model and context); JPPGPU_NOT_IMPLEMENTED for sizes it cannot index */
int jppgpu_ctx_set_format_table(jppgpu_ctx* ctx, const jppgpu_format_table* table);

typedef struct {
  uint32_t n_sentences;
  const uint64_t* offsets;   /* [n + 1] byte offsets into text: sentence i is text[offsets[i] .. offsets[i + 1]) */
  const char* text;          /* host copy, owned by the result (valid until jppgpu_result_release) */
  const int32_t* status;     /* [n] JPPGPU_SENT_* */
} jppgpu_text_view;
/* the formatted top-1 analyses of the batch; needs jppgpu_ctx_set_format_table.  The device side of the result must
 * still be valid (no later batch on the context). */
int jppgpu_result_format_top1(jppgpu_result* res, jppgpu_text_view* view);

/* The n best analyses, as jumandic::output::LatticeFormat consumes them (LatticeFormatInfo::fillInfo,
 * src/jumandic/shared/lattice_format.cc:13-43; score lookup :129-141): for every EOS beam slot i < n_best
 * the connections of its path from the EOS side back to BOS, each with the beam slot, the node and UNK
 * records and the score cells of that connection.  Gathered on the device, so that a lattice-format run
 * does not copy the whole lattice (beam 32: gigabytes per batch) to the host. */
typedef struct {
  uint32_t node;           /* sentence-local node index, as in the full view */
  uint32_t slot;           /* beam slot of that node on the path */
  jppgpu_beam_slot beam;   /* beams[node][slot]; .pad = index of its score cells in the boundary's global beam */
  jppgpu_node info;
  jppgpu_unk unk;
  float cells[2];          /* cells[node][beam.pad][0..num_scorers) */
} jppgpu_nbest_item;

typedef struct {
  uint32_t n_sentences;
  int32_t n_best, beam, global_beam, num_scorers;
  const int32_t* status;            /* [n_sentences] */
  const uint32_t* n_codepoints;     /* [n_sentences] */
  const uint32_t* n_nodes;          /* [n_sentences] lattice nodes (<= 3: empty input) */
  const jppgpu_beam_slot* eos;      /* [n_sentences][n_best] EOS beam slots, fake beyond the live ones */
  const uint64_t* path_first;       /* [n_sentences * n_best + 1] offsets into items */
  const jppgpu_nbest_item* items;   /* path (s, i) = items[path_first[s*n_best+i] .. path_first[s*n_best+i+1]) */
} jppgpu_nbest_view;

int jppgpu_result_fetch_nbest(jppgpu_result* res, int32_t n_best, jppgpu_nbest_view* view);

/* ---- the lattice (-s N) format on the device (replaces jumandic::output::LatticeFormat::format,
 * src/jumandic/shared/lattice_format.cc:83-242, and the LatticeFormatInfo bookkeeping :13-66,250-270) -------------------------
 * One line per lattice node on any of the N best paths and per row of its dictionary entry:
 *     "-" TAB id TAB prev-ids(;) TAB start TAB end TAB  S TAB X TAB R TAB B TAB REST  [flag '|']  scores  ranks(;) LF
 * S / R / B = surface / reading / baseform with a lone tab escaped, X = the canonic form or B '/' R when it is empty, REST
 * = pos, ids, conjugation columns and the feature list as the reference prints them -- for a DICTIONARY node the whole
 * run S .. REST is a function of the entry row and is rendered once per model by the host (host/lattice_table.cc); an UNK
 * node prints the input surface in the columns its maker replaces.  ids, previous ids, ranks and the three scores ("%g")
 * are computed per sentence by the kernels (csrc/k_latfmt.h).  The library knows nothing of JUMAN: every literal is here. */
typedef struct {
  uint32_t blob_off;       /* S TAB X TAB R TAB B TAB REST, contiguous in `blob` */
  uint32_t len_rest;
  uint16_t len_s, len_c, len_r, len_b;   /* len_c = 0: X is B '/' R */
  uint32_t flags;          /* bit 1: last row of its entry */
} jppgpu_lattice_row;      /* 20 bytes */

typedef struct {
  uint32_t struct_size;    /* sizeof(jppgpu_lattice_table) */
  const uint32_t* slot_first_row;   /* as in jppgpu_format_table */
  uint64_t n_slots;
  const jppgpu_lattice_row* rows;
  uint64_t n_rows;
  const char* blob;
  uint64_t blob_bytes;
  uint8_t maker_replaces[16];   /* per UNK maker: bit 0 / 1 / 2 / 3 = S / R / B / canonic form print the input surface */
  uint8_t n_escapes;            /* escapeTab (lattice_format.cc:74-79) */
  char escape_from[4];
  uint8_t escape_len[4];
  char escape_to[4][8];
  int32_t flag_placeholder;     /* formatNormalizedFeature as in the JUMAN table above; followed by a bar here */
  uint8_t flag_label_len;
  char flag_label[32];
  uint8_t n_flags;
  uint32_t flag_mask[16];
  char flag_char[16];
  uint8_t head_len, rank_len, feat_len, lm_len, total_len, ranks_len, eos_len, error_len;
  char head_text[16];      /* "# MA-SCORE\t" */
  char rank_text[8];       /* "rank" */
  char feat_text[32];      /* label of scores[0] * weights[0] */
  char lm_text[32];        /* label of scores[1] * weights[1] (printed when there are two weights) */
  char total_text[32];     /* label of their sum */
  char ranks_text[16];     /* label of the rank list */
  char eos_text[16];
  char error_text[32];     /* whole text of a sentence that failed */
  uint32_t n_weights;      /* ScorerDef::scoreWeights as the format reads them */
  float weights[2];
} jppgpu_lattice_table;

int jppgpu_ctx_set_lattice_table(jppgpu_ctx* ctx, const jppgpu_lattice_table* table);
/* the fields of jppgpu_text_view -- which callers of jppgpu_result_format_top1 hold at its old size -- plus one array */
typedef struct {
  uint32_t n_sentences;
  const uint64_t* offsets;   /* [n + 1] */
  const char* text;
  const int32_t* status;     /* [n] */
  const uint32_t* head_len;  /* [n] bytes of the "# MA-SCORE ..." line a sentence's text starts with (0: none) -- a caller that
                              * has a comment for the sentence prints "# comment\n" in its place (lattice_format.cc:105-120) */
} jppgpu_lattice_text_view;
/* the lattice-format text of the batch's n_best (<= 64) best analyses.  Needs
 * jppgpu_ctx_set_lattice_table and a context with a global beam (the format reads the score cells); the device side of
 * the result must still be valid.  A result holds ONE text: the first of format_top1 / format_lattice called on it. */
int jppgpu_result_format_lattice(jppgpu_result* res, int32_t n_best, jppgpu_lattice_text_view* view);
/* Training hook.  What the reference's trainer reads off an analysed lattice besides the scores
 * (LossCalculator::addTopNgrams, src/core/training/loss.cc:289-300 -> NgramFeaturesComputer::calculateNgramFeatures,
 * src/core/impl/feature_computer.cc:13-31): for every connection on the top-1 path of every sentence, from the EOS
 * side back to the first morpheme, the u32 value of each n-gram feature of the spec for (t2, t1, t0 = that node) --
 * the weight index before masking by the table size, i.e. where a perceptron update adds its deltas. */
typedef struct {
  uint32_t n_sentences;
  uint32_t n_ngram;                /* features per position, in the spec's feature order (jumandic: 73) */
  const uint64_t* path_first;      /* [n_sentences + 1] offsets into path_nodes / features (0 positions for failed sentences) */
  const uint32_t* path_nodes;      /* [path_first[n]] sentence-local node index of t0, EOS first */
  const uint32_t* features;        /* [path_first[n]][n_ngram] */
} jppgpu_top1_ngrams_view;
int jppgpu_result_fetch_top1_ngrams(jppgpu_result* res, jppgpu_top1_ngrams_view* view);
/* The same values along caller-given paths (the gold path: LossCalculator::resolveGold, loss.cc:366-389):
 * path_first[n + 1] offsets into path_nodes, which lists sentence-local node indices in TEXT order; the first two
 * positions of a path see the BOS nodes as t1 / t2 (NgramFeatureRef::init).  In the returned view path_first /
 * path_nodes repeat the input and features[k] belongs to position k.  Paths of failed sentences must be empty. */
int jppgpu_result_fetch_path_ngrams(jppgpu_result* res, const uint64_t* path_first, const uint32_t* path_nodes,
                                    jppgpu_top1_ngrams_view* view);
/* Replaces the perceptron weight table of the context (the trainer's update step; HashedFeaturePerceptron /
 * FloatBufferWeights, src/core/analysis/perceptron.h:76-94, score_api.h:29-42).  n must equal the model's table
 * size; takes effect for the next jppgpu_analyze_batch* on the context.  (A context that runs the static feature code
 * also rebuilds its per-dictionary-entry T0 records from the new table: ~0.1 s for a 300 k-entry dictionary.) */
int jppgpu_ctx_set_weights(jppgpu_ctx* ctx, const float* weights, uint64_t n);
/* Per-batch statistics without copying the lattice: total nodes, sum of path lengths. */
int jppgpu_result_stats(jppgpu_result* res, uint64_t* total_nodes, uint64_t* total_path);
/* Packed top-1 output written to caller-provided DEVICE buffers (e.g. to be gathered
 * across GPUs with RCCL): d_offsets[n+1] (exclusive scan of morphemes per sentence) and
 * d_items[cap_items] of jppgpu_node in text order.  Enqueued on the batch's stream. */
int jppgpu_result_pack(jppgpu_result* res, void* d_offsets, void* d_items, uint64_t cap_items);
/* May be called from any thread, also while another thread analyses or fetches on the result's context (the host
 * blocks of a result go back to the context's pools under a lock), and after the context was destroyed. */
void jppgpu_result_release(jppgpu_result* res);

/* device timing of the last batch's kernels in milliseconds (HIP events on the launch stream):
 * [0] decode [1] seeds [2] layout [3] t0 [4] sweep [5] rnn [6] path, [7] whole pipeline; and, for n > 8, the sweep by
 * sentence class (every sentence runs the kernel variant of its own widest boundary): [8] [9] [10] ms of the variants
 * for at most 64 / at most 512 / any number of nodes starting at one boundary, [11] [12] [13] sentences in each class,
 * [14] rows of the RNN hidden-state table of the batch (rnn nodes + 2 per sentence; 0 without the RNN), [15] ms of
 * k_rnn_chain alone (0 when it did not run); of the context's last jppgpu_result_format_top1 / _lattice call: [16] ms of
 * the count pass + offset scan, [17] ms of the write pass, [18] MB of text */
int jppgpu_last_timings(jppgpu_ctx* ctx, float* ms, int n);

#ifdef __cplusplus
}
#endif

#endif /* JPPGPU_H */
