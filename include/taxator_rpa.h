/* include/taxator_rpa.h -- C ABI of the MI355X-native realignment-placement (RPA) engine.
 *
 * Drop-in boundary for ONE path of fungs/taxator-tk v1.5.0: RPAPredictionModel::predict and everything below it
 * (pairwise re-alignment, distance ranking, taxon-tree LCA placement).  The reference has no FFI; the seam it
 * offers is the C++ plugin class
 *     template<class ContainerT> class TaxonPredictionModel { virtual void predict(ContainerT& recordset,
 *         PredictionRecord& prec, std::ostream& logsink) = 0; };          core/src/taxonpredictionmodel.hh:36-52
 *     RPAPredictionModel(const Taxonomy*, QStorType&, const DBStorType&, float exclude_factor,
 *         float reeval_bandwidth)                                         core/src/taxonpredictionmodelsequence.hh:329
 * called once per record set from core/taxator.cpp:69-74 (serial) / :159-176 (threads).  This header is what a
 * batched replacement of that seam binds: plain pointers and sizes, no C++ / torch types.  INTEGRATION.md shows
 * the reference-side adapter.  Every entry point returns RPA_OK or an error code; rpa_last_error() gives the text
 * (the reference throws Exception subclasses, core/src/exception.hh:29-73; nothing is thrown across this ABI).
 *
 * All kernels are hand-written HIP for gfx950; there is no CPU fallback behind these symbols.
 */
#ifndef TAXATOR_RPA_H
#define TAXATOR_RPA_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RPA_ABI_VERSION 4

enum { RPA_DNA = 0, RPA_PROTEIN = 1 };              /* taxator -b nucleotide|protein, core/taxator.cpp:280,362-371 */
enum { RPA_STORE_QUERY = 0, RPA_STORE_DB = 1 };      /* query_sequences_ / db_sequences_, taxonpredictionmodelsequence.hh:842-843 */
enum { RPA_MEM_HOST = 0, RPA_MEM_DEVICE = 1 };       /* where caller-owned arrays live */

enum {
  RPA_OK = 0,
  RPA_ERR_INVALID = 1,   /* bad argument / inconsistent batch */
  RPA_ERR_DEVICE = 2,    /* HIP runtime error, no device, out of device memory */
  RPA_ERR_RANGE = 3,     /* SequenceRangeError: segment start beyond the sequence (sequencestorage.hh:341-362) */
  RPA_ERR_SEQ_ID = 4,    /* SequenceNotFound (sequencestorage.hh:348-351): sequence index outside the store */
  RPA_ERR_ASSERT = 5,    /* an assert that is live in the reference build would abort the run (CMakeLists.txt:38) */
  RPA_ERR_UNSUPPORTED = 6
};

/* Pruned taxonomy, flattened (core/src/taxontree.hh:45-70 after TaxonTree::deleteUnmarkedNodes, taxontree.cpp:55-70).
 * Node 0 is the root and parent[0] == 0.  left/right are any valid nested-set numbering; depth = root_pathlength.
 * Replaces TaxonomyInterface::getLCA / isParentOf / getRoot (core/src/taxonomyinterface.cpp:16,52-55,67-77). */
typedef struct {
  uint32_t n_nodes;
  const uint32_t* parent;
  const uint32_t* left;
  const uint32_t* right;
  const uint8_t* depth;
} rpa_taxonomy;

typedef struct {
  int alphabet;                    /* RPA_DNA | RPA_PROTEIN */
  float exclude_alignments_factor; /* -x, default 0.5   (taxator.cpp:286, ctor arg exclude_factor) */
  float reeval_bandwidth;          /* -t, default 0.05  (taxator.cpp:287, ctor arg reeval_bandwidth) */
  int device;                      /* HIP device ordinal */
  /* Speculation chunks: how many hits of a record set get their alignment requested per round of the pass-1 / pass-2
   * anchor loops (:604, :757).  The loops have data-dependent early exits; requesting everything at once (0) wastes
   * alignments the reference would never compute, small chunks cost extra rounds.  Results and the demanded-alignment
   * counters are identical for every setting.  RPA_SPECULATION_DEFAULT picks the engine's tuned values. */
  uint32_t speculation_pass1;
  uint32_t speculation_pass2;
  uint32_t reserved[2];            /* must be zero */
} rpa_params;
#define RPA_SPECULATION_DEFAULT 0xFFFFFFFFu

typedef struct rpa_engine rpa_engine;

int rpa_abi_version(void);
void rpa_default_params(rpa_params* p);

/* RPAPredictionModel ctor (taxonpredictionmodelsequence.hh:329-339): copies the taxonomy to the device. */
int rpa_create(const rpa_taxonomy* tax, const rpa_params* params, rpa_engine** out);
void rpa_destroy(rpa_engine* e);
/* Text of the last error on this engine (or of the last failed rpa_create when e == NULL). */
const char* rpa_last_error(const rpa_engine* e);
/* Diagnostics for a caller's watchdog: a static string naming the blocking runtime call any engine of this process
 * entered last ("predict_part: sync after emit (round)", "rpa_destroy: hipStreamDestroy", ...).  No reference counterpart
 * (the reference has no device to wait for); the `taxator` driver prints it when it is told to report where it is. */
const char* rpa_debug_where(void);

/* Run on a caller-provided hipStream_t (e.g. the current stream of the framework that owns the device buffers).
 * NULL restores the engine's own stream.  The engine's own stream is independent (non-blocking): device-resident
 * inputs (RPA_MEM_DEVICE batches, stores) must be complete before a call unless the engine has been put on the stream
 * that produces them. */
int rpa_set_stream(rpa_engine* e, void* hip_stream);

/* Sequence stores, resident in HBM -- replaces RandomSeqStoreROInterface::getSequence /
 * getSequenceReverseComplement (core/src/sequencestorage.hh:41-52; indexed store :318-466, reader core/src/faidx.h:315-350).
 * `letters` are the raw FASTA residue bytes of all sequences back to back (no line breaks); sequence i occupies
 * [offsets[i], offsets[i+1]).  Bytes are mapped to alphabet ordinals on the device exactly as the indexed store does
 * (seqan/basic/alphabet_residue_tabs.h:107-140 Dna5: ACGT(U) any case, everything else N; :397-430 AminoAcid: 27
 * letters, everything else X) and packed (4 bit/base DNA, 8 bit/residue protein).  Coordinates in a batch are the
 * reference's: 1-based inclusive; rstart > rstop means reverse strand. */
int rpa_store_load(rpa_engine* e, int which, uint32_t n_seqs, const uint8_t* letters, const uint64_t* offsets,
                   int mem_kind);

/* Indexed stores.  The reference's RandomIndexedSeqstoreRO (core/src/sequencestorage.hh:318-406) keeps
 * nothing in memory and reads every region through the .fai index (core/src/faidx.h:315-350).  The counterpart here:
 * rpa_store_reserve makes a store of n_seqs sequences of which none is resident yet (sequence indices = positions in
 * the index file); rpa_store_append packs the letters of the sequences `seq_index[0..n)` (host memory, `offsets` = n + 1
 * entries from 0, as in rpa_store_load) behind what is resident and makes them readable.  The host driver appends the
 * sequences a block of the alignment stream refers to before it submits the block, so HBM and load time scale with the
 * sequences touched, not with the collection.  A record that points at a sequence that was never appended sees an
 * empty sequence (RPA_ERR_RANGE).  Appending an index twice wastes arena space but is harmless. */
int rpa_store_reserve(rpa_engine* e, int which, uint32_t n_seqs);
int rpa_store_append(rpa_engine* e, int which, uint32_t n, const uint32_t* seq_index, const uint8_t* letters, const uint64_t* offsets);
/* Region-granular residency (what the reference's indexed store really does: getSequence(id, start, stop) reads only
 * [start, stop] through the .fai, core/src/sequencestorage.hh:341-369, core/src/faidx.h:315-350 -- a 200 bp hit on a
 * 50 Mbp chromosome costs 200 bp of I/O, not 50 Mbp).  rpa_store_reserve_windows makes a store whose sequences have the
 * given lengths and no resident part; rpa_store_append_windows makes the residues [first, first + n) of sequence
 * seq_index readable (letters of the windows back to back, `offsets` n + 1 entries from 0).  A reference / query segment
 * of a batch must lie inside ONE window (the driver appends, per block of the alignment stream, tile-aligned windows that
 * cover every hit's range extended by its record set's query range, merged with the resident windows they touch); a
 * segment outside every window fails the batch with RPA_ERR_INVALID.  A new window may sit beside the resident windows of
 * its sequence or cover some of them entirely (they are dropped), never overlap one partly.
 * rpa_store_reset (any store made with rpa_store_reserve*): nothing is resident any more, the arena is refilled from
 * its start -- how a driver keeps a store inside an HBM budget (`taxator --store-gb`): reset, then append what the
 * current block needs.  Both wait for the jobs in flight. */
typedef struct {
  uint32_t seq_index;
  uint32_t reserved;   /* must be zero */
  uint64_t first;      /* 0-based first residue of the window inside its sequence */
  uint64_t n;          /* residues */
} rpa_window;
int rpa_store_reserve_windows(rpa_engine* e, int which, uint32_t n_seqs, const uint64_t* lengths);
int rpa_store_append_windows(rpa_engine* e, int which, uint32_t n, const rpa_window* windows, const uint8_t* letters, const uint64_t* offsets);
int rpa_store_reset(rpa_engine* e, int which);
typedef struct {
  uint32_t n_seqs;       /* sequences the store knows */
  uint32_t n_resident;   /* of which (partly) resident in HBM */
  uint64_t residues;     /* residues of the arena in use (windowed stores: including replaced windows) */
  uint64_t device_bytes; /* HBM bytes held by the store (packed arena incl. growth slack + extent / window tables) */
  uint64_t windows;      /* resident windows (whole-sequence stores: resident sequences) */
  uint64_t live_residues;/* residues inside resident windows */
} rpa_store_info;
int rpa_store_info_get(rpa_engine* e, int which, rpa_store_info* out);

/* One batch of record sets = what RecordSetGenerator*::getNext hands to predict() one at a time
 * (core/src/alignmentrecord.hh:457-631), as structure-of-arrays.  Record fields are exactly those predict() reads
 * (SURVEY.md 8b): AlignmentRecord getters core/src/alignmentrecord.hh:40-88, reference node :229-231. */
typedef struct {
  uint32_t n_sets;
  uint32_t n_records;
  const uint32_t* set_offsets; /* n_sets+1; records of set s are [set_offsets[s], set_offsets[s+1]) in generator order */
  const uint32_t* query_seq;   /* n_sets: index of the query sequence in the QUERY store (getQueryIdentifier) */
  const uint32_t* query_len;   /* n_sets: getQueryLength() of the set's first record */
  /* n_records each: */
  const uint32_t* qstart;      /* getQueryStart  */
  const uint32_t* qstop;       /* getQueryStop   */
  const uint32_t* rstart;      /* getReferenceStart */
  const uint32_t* rstop;       /* getReferenceStop  */
  const float* score;          /* getScore       */
  const uint32_t* identities;  /* getIdentities  */
  const uint32_t* alnlen;      /* getAlignmentLength */
  const uint32_t* ref_seq;     /* index of getReferenceIdentifier() in the DB store */
  const uint32_t* node;        /* taxonomy node index of getReferenceNode() */
  const uint8_t* filtered;     /* isFiltered() (leading '*' in the input line); may be NULL = none */
  int mem_kind;                /* RPA_MEM_HOST | RPA_MEM_DEVICE for every pointer above */
} rpa_batch;

#define RPA_FLAG_IVAL_UNTOUCHED 1u /* no active record: setUnclassified(), ival keeps its previous value (:359-368) */
#define RPA_FLAG_SHORTCUT 2u       /* best hit identical to the query range: no alignment (:431-472) */
#define RPA_FLAG_SINGLE 4u         /* one active record (:371-388) */

/* What predict() writes into the PredictionRecord (taxonpredictionmodelsequence.hh:828-833): 32 bytes per set,
 * also the row that ranks gather to rank 0.  Node fields are taxonomy indices. */
typedef struct {
  uint32_t lower;   /* setNodeRange lower node */
  uint32_t upper;   /* setNodeRange upper node */
  uint32_t rtax;    /* setBestReferenceTaxon */
  uint32_t support; /* setNodeRange support */
  uint32_t begin;   /* setQueryFeatureBegin */
  uint32_t end;     /* setQueryFeatureEnd */
  float ival;       /* setInterpolationValue */
  uint32_t flags;
} rpa_prediction;

/* Per-set alignment counters = the reference's STATS log line (:837): n, pass_0/1/2_counter. */
typedef struct {
  uint32_t n, pass0, pass1, pass2;
} rpa_stats;

/* predict() for every set of the batch.  `out` (n_sets rows) and `stats` (n_sets rows or NULL) live where out_mem_kind
 * says.  Results are in input order.  Synchronous: returns when the rows are written (asynchronous form: rpa_submit_batch). */
int rpa_predict_batch(rpa_engine* e, const rpa_batch* batch, rpa_prediction* out, rpa_stats* stats, int out_mem_kind);

/* The reference's per-alignment log ("+ALN i <=> j  dist=..; sim=..", taxonpredictionmodelsequence.hh:533,636,782,802: one
 * line per alignment its control flow computes, in the order it computes them) as records.  rpa_trace_begin(capacity > 0)
 * turns recording on for the synchronous rpa_predict_batch calls of this engine (capacity = entries kept per call;
 * 0 turns it off); rpa_trace_read returns the entries of the last call sorted by (record set, order inside the set) and
 * how many did not fit (capacity 0, out NULL: nothing is copied, *n_dropped = the number of entries the call produced --
 * ask, size a buffer, read).  i, j are hit positions in SortFilter order; j = -1: the query range ("i <=> query"), i = -1:
 * "query <=> j" (pass 2).  sim is what the reference prints there (pass 0 / "query <=> j": after std::max with the
 * identities / the stored similarity).  Only alignments the reference would have computed are recorded, never the
 * engine's speculative ones: per set there are exactly pass0 + pass1 + pass2 (rpa_stats) entries. */
typedef struct {
  uint32_t set;   /* record set inside the batch */
  uint32_t seq;   /* 0, 1, 2, ... inside the set */
  int32_t pass, i, j;
  float dist, sim;
  uint32_t reserved;
} rpa_trace_entry;
int rpa_trace_begin(rpa_engine* e, uint64_t capacity);
int rpa_trace_read(rpa_engine* e, rpa_trace_entry* out, uint64_t capacity, uint64_t* n_entries, uint64_t* n_dropped);

/* Asynchronous form for host batches: the replacement of the reference's producer / consumer pool
 * (core/taxator.cpp:80-124,159-176) at the boundary.  rpa_submit_batch hands the batch to one of the engine's
 * RPA_JOBS_IN_FLIGHT job slots (each with its own HIP stream, working set, staging buffers and host thread) and returns;
 * the host -> device copies of one job run under the kernels of the other.  It blocks only while the slot whose turn it
 * is still works on its previous job.  `out` / `stats` live where the batch lives (batch->mem_kind; a device batch must be
 * complete when it is submitted: the job runs on a stream of the engine's own).  The batch arrays, `out` and `stats` must stay valid until rpa_wait(job) returns,
 * which also reports the job's status (rpa_last_error has the text) and releases the handle.  Jobs may be waited for in
 * any order; rows are in input order within a job, as with rpa_predict_batch (same semantics:
 * core/src/taxonpredictionmodel.hh:36-52).  Store loads / appends wait for the jobs in flight. */
/* Host arrays handed to the engine are read with hipMemcpyAsync.  From pageable memory that copy is staged by the driver
 * on the calling thread (~25 GB/s, and it does not overlap the other job's copies); from memory pinned with
 * rpa_host_register it is a DMA at the PCIe rate that runs beside the kernels.  Register the batch arrays / output arrays
 * you reuse from batch to batch once (page-pinning itself is slow: a few GB/s), unregister before freeing them. */
int rpa_host_register(rpa_engine* e, void* ptr, size_t bytes);
int rpa_host_unregister(rpa_engine* e, void* ptr);
#define RPA_JOBS_IN_FLIGHT 4
typedef struct rpa_job rpa_job;
int rpa_submit_batch(rpa_engine* e, const rpa_batch* batch, rpa_prediction* out, rpa_stats* stats, rpa_job** job);
int rpa_wait(rpa_engine* e, rpa_job* job);

/* The reference's alignment-free models behind the same seam (taxator -a dummy | simple-lca | megan-lca | n-best-lca,
 * core/taxator.cpp:346-361): DummyPredictionModel, LCASimplePredictionModel, MeganLCAPredictionModel,
 * NBestLCAPredictionModel (core/src/taxonpredictionmodel.hh:57-253) with the filters they run
 * (core/src/alignmentsfilter.hh:349-386,493-532,612-626).  The batch is the same rpa_batch (rstart/rstop/identities/
 * alnlen/ref_seq/query_seq are not read and may be NULL); `evalue` (n_records, getEValue(), may be NULL = 0) and
 * `node_unclassified` (n_nodes, Taxon::is_unclassified, core/src/ncbidata.cpp:119-126; may be NULL) live where
 * batch->mem_kind says.  No sequence store is needed.  Rows: lower == upper == the predicted node, support = width of
 * the query range, ival = -1 with RPA_FLAG_IVAL_UNTOUCHED (these models never set it). */
enum { RPA_LCA_DUMMY = 0, RPA_LCA_SIMPLE = 1, RPA_LCA_MEGAN = 2, RPA_LCA_NBEST = 3 };
typedef struct {
  int model;               /* RPA_LCA_* */
  float toppercent;        /* -t, default 0.05 (taxator.cpp:287) */
  float minscore;          /* -m, default 0 */
  double maxevalue;        /* -e, default 1000 (narrowed to float by the reference's filter, alignmentsfilter.hh:351) */
  uint32_t minsupport;     /* -c, default 1 */
  uint32_t nbest;          /* -n, default 1 */
  int ignore_unclassified; /* -u */
  uint32_t reserved;       /* must be zero */
} rpa_lca_params;
void rpa_lca_default_params(rpa_lca_params* p);
int rpa_lca_predict_batch(rpa_engine* e, const rpa_batch* batch, const double* evalue, const uint8_t* node_unclassified,
                          const rpa_lca_params* params, rpa_prediction* out, int out_mem_kind);

/* Pair-level entry: getAlignment<StringType>(A, B) (taxonpredictionmodelsequence.hh:133-171 DNA, :173-242 protein)
 * for explicit pairs of sequences of a store-shaped arena (letters/offsets as in rpa_store_load; not kept).
 * Used by the parity tests and the kernel benchmark; same kernels as rpa_predict_batch. */
typedef struct {
  uint32_t matches, mismatches, gaps, length;
  float distance, similarity;
} rpa_alignment;

int rpa_align_pairs(rpa_engine* e, uint32_t n_seqs, const uint8_t* letters, const uint64_t* offsets, uint32_t n_pairs,
                    const uint32_t* pair_a, const uint32_t* pair_b, rpa_alignment* out, int mem_kind);

/* Live kernel timing (HIP events on the engine's stream around each launch of the alignment kernels). */
typedef struct {
  uint64_t launches;       /* alignment-kernel launches since reset */
  uint64_t pairs;          /* alignments computed (including speculative ones the control flow then ignores) */
  uint64_t residue_bytes;  /* sum over computed pairs of len_a + len_b + 8 (SURVEY.md 8d algorithmic bytes) */
  uint64_t cells;          /* sum over computed pairs of len_a * len_b */
  double align_ms;         /* device time inside alignment kernels */
  double total_ms;         /* device time of whole rpa_predict_batch / rpa_align_pairs calls */
  uint64_t demanded;       /* alignments the reference control flow demanded (sum of pass0+pass1+pass2) */
  uint64_t rounds;         /* plan/align/decide rounds */
  uint64_t cells_updated;  /* DP cells the alignment kernels really updated: the banded / strip kernels sweep only the words and
                              column groups their band touches (cells counts len_a * len_b "cell-equivalents" whatever was swept);
                              protein: every cell of the register strips, padding rows included (ABI 4) */
} rpa_profile;

/* on: 0 = off; 1 = launch timing + the counters (pairs, residue_bytes, cells, cells_updated, demanded: reductions inside the
 * alignment kernels, which cost the dominant nucleotide kernel ~4 % of its time); RPA_PROFILE_TIMING = launch timing only
 * (launches, align_ms, total_ms, rounds advance; the counters keep the values of the last counting pass). */
#define RPA_PROFILE_TIMING 2
int rpa_profile_enable(rpa_engine* e, int on);
int rpa_profile_get(rpa_engine* e, rpa_profile* out);
int rpa_profile_reset(rpa_engine* e);

#ifdef __cplusplus
}
#endif
#endif /* TAXATOR_RPA_H */
