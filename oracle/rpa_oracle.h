/* oracle/rpa_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C CPU restatement of the reference's realignment-placement (RPA) hot path (fungs/taxator-tk v1.5.0).
 * It is the checker for the HIP product: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
 * may load it.  Nothing under taxator-tk_amd/ links, imports or calls it.
 *
 * Pinning status (see DESIGN.md "Oracle"):
 *   - pair level (a1-a4: Myers edit distance, BLOSUM62 NW + traceback statistics, alphabets, reverse complement):
 *     PINNED against the reference's own vendored SeqAn 2.4.0 compiled in place (oracle/_ref/libseqan_ref.so,
 *     recipe oracle/Makefile) and against the committed golden vectors generated from it (tests/golden/).
 *   - record-set level (a5-a9: RPAPredictionModel::predict, BandFactor, LCA, SortFilter, segment fetch):
 *     PARITY UNPINNED by a reference build made here -- core/src/taxonpredictionmodelsequence.hh needs Boost,
 *     which this image lacks, and no stand-in is written for it.  Secondary evidence only: GFF3 + log output
 *     that the survey stage produced with the reference code (tests/golden/survey_*; provenance in its README).
 */
#ifndef RPA_ORACLE_H
#define RPA_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { ORC_DNA = 0, ORC_PROTEIN = 1 };
enum {
  ORC_OK = 0,
  ORC_ERR_RANGE = 1,      /* SequenceRangeError / assert(length(seq)==stop-start+1), sequencestorage.hh:341-362 */
  ORC_ERR_SEQ_ID = 2,     /* SequenceNotFound, sequencestorage.hh:348-351 */
  ORC_ERR_ASSERT = 3,     /* a live assert of the reference build would abort (core/CMakeLists.txt:38 has no -DNDEBUG) */
  ORC_ERR_NOMEM = 4,
  ORC_ERR_NAN_SCORE = 5   /* a NaN score among >= 2 active records: list::sort (:412) on an inconsistent comparator -- refused */
};

/* Flattened pruned taxonomy (core/src/taxontree.hh:45-70): node 0 is the root, parent[0]==0. */
typedef struct {
  uint32_t n_nodes;
  const uint32_t* parent;
  const uint32_t* left;   /* nested-set leftvalue  */
  const uint32_t* right;  /* nested-set rightvalue */
  const uint8_t* depth;   /* root_pathlength */
} orc_taxonomy;

/* RandomSeqStoreROInterface model (core/src/sequencestorage.hh:41-52): residues are alphabet ordinals. */
typedef struct {
  const uint8_t* residues;
  const uint64_t* offsets; /* n_seqs + 1 */
  uint32_t n_seqs;
} orc_store;

/* Fields of AlignmentRecordTaxonomy that predict() reads (core/src/alignmentrecord.hh:95-158,207-238). */
typedef struct {
  uint32_t qstart, qstop, rstart, rstop;
  float score;
  uint32_t identities, alnlen;
  uint32_t ref_seq; /* index into the db store */
  uint32_t node;    /* taxonomy node index of the reference */
  uint32_t filtered;
} orc_record;

#define ORC_FLAG_IVAL_UNTOUCHED 1u /* n==0: setUnclassified leaves ival/signal stale (taxonpredictionmodelsequence.hh:359-368) */
#define ORC_FLAG_SHORTCUT 2u       /* perfect best hit, no alignment (:431-472) */
#define ORC_FLAG_SINGLE 4u         /* n==1 (:371-388) */

typedef struct {
  uint32_t lower, upper, rtax; /* node indices */
  uint32_t support;
  uint32_t begin, end; /* query feature begin/end */
  float ival;
  uint32_t flags;
} orc_prediction;

typedef struct {
  uint32_t n, pass0, pass1, pass2;
} orc_stats;

typedef struct {
  uint32_t matches, mismatches, gaps, length;
  float distance, similarity;
  int32_t selfscore, mutualscore; /* protein only */
} orc_alignment;

/* j == -1 means "query"; called once per alignment actually computed (the +ALN log lines). */
typedef void (*orc_trace_fn)(void* ctx, int pass, int i, int j, float dist, float sim);

/* a1: SeqAn Myers bit-vector global score (<=0), 32-bit blocks (global_alignment_myers_impl.h:62-196). */
int orc_myers_score(const uint8_t* h, uint32_t len_h, const uint8_t* v, uint32_t len_v);
/* textbook Levenshtein (cross-check for a1). */
int orc_levenshtein(const uint8_t* a, uint32_t la, const uint8_t* b, uint32_t lb);
/* a2 / a3+a4 */
void orc_align_dna(const uint8_t* a, uint32_t la, const uint8_t* b, uint32_t lb, orc_alignment* out);
int orc_align_protein(const uint8_t* a, uint32_t la, const uint8_t* b, uint32_t lb, orc_alignment* out);
/* protein, computed the way the device kernel does it (forward-carried counters, self scores as diagonal sums);
   exists so tests can prove the two formulations agree on the CPU before trusting the kernel design. */
int orc_align_protein_forward(const uint8_t* a, uint32_t la, const uint8_t* b, uint32_t lb, orc_alignment* out);

/* a8 */
uint32_t orc_lca(const orc_taxonomy* t, uint32_t a, uint32_t b);
int orc_is_parent_of(const orc_taxonomy* t, uint32_t a, uint32_t b);
/* a5: entries[0] is the anchor (dist 0). */
float orc_bandfactor(const orc_taxonomy* t, const float* dist, const uint32_t* node, uint32_t n);
/* a7: out must hold (stop+right_ext)-(start-left_ext)+1 residues at most; returns ORC_* and the length. */
int orc_get_segment(const orc_store* db, int alphabet, uint32_t seq, uint32_t start, uint32_t stop, uint32_t left_ext,
                    uint32_t right_ext, uint8_t* out, uint32_t* out_len);
uint32_t orc_segment_capacity(uint32_t start, uint32_t stop, uint32_t left_ext, uint32_t right_ext);
void orc_revcomp_dna5(uint8_t* s, uint32_t n);
/* a9: stable order by (score desc, identities desc); writes a permutation of 0..n-1. */
void orc_sort_records(const orc_record* recs, uint32_t n, uint32_t* order);

/* a6: RPAPredictionModel::predict for one record set. */
int orc_predict(const orc_taxonomy* tax, const orc_store* qstore, const orc_store* dbstore, int alphabet,
                float exclude_factor, float reeval_bandwidth, uint32_t query_seq, uint32_t query_len,
                const orc_record* recs, uint32_t nrecs, orc_prediction* out, orc_stats* stats, orc_trace_fn trace,
                void* ctx);

/* Convenience for the CPU baseline: loop orc_predict over a batch (set_offsets has n_sets+1 entries). */
int orc_predict_batch(const orc_taxonomy* tax, const orc_store* qstore, const orc_store* dbstore, int alphabet,
                      float exclude_factor, float reeval_bandwidth, uint32_t n_sets, const uint32_t* set_offsets,
                      const uint32_t* query_seq, const uint32_t* query_len, const orc_record* recs,
                      orc_prediction* out, orc_stats* stats);

/* ---- alignment-free models (lca_oracle.c; PARITY UNPINNED, see there) ---- */
enum { ORC_LCA_DUMMY = 0, ORC_LCA_SIMPLE = 1, ORC_LCA_MEGAN = 2, ORC_LCA_NBEST = 3 }; /* taxator -a dummy|simple-lca|megan-lca|n-best-lca */
typedef struct {
  int model;
  float toppercent;  /* -t */
  float minscore;    /* -m */
  double maxevalue;  /* -e */
  uint32_t minsupport; /* -c */
  uint32_t nbest;      /* -n */
  int ignore_unclassified; /* -u */
} orc_lca_params;
int orc_lca_predict(const orc_taxonomy* tax, const orc_lca_params* p, uint32_t query_len, const orc_record* recs,
                    const double* evalue, const uint8_t* node_unclassified, uint32_t n, orc_prediction* out);
int orc_lca_predict_batch(const orc_taxonomy* tax, const orc_lca_params* p, uint32_t n_sets, const uint32_t* set_offsets,
                          const uint32_t* query_len, const orc_record* recs, const double* evalue,
                          const uint8_t* node_unclassified, orc_prediction* out);

#ifdef __cplusplus
}
#endif
#endif
