/* oracle/rpa_oracle.c -- TEST INFRASTRUCTURE ONLY (see rpa_oracle.h for the rules and the pinning status).
 *
 * CPU restatement, in plain C, of the reference's RPA hot path.  Every function cites the reference lines it
 * follows; paths are relative to /root/reference/core.  "seqan/" = includes-external/seqan (SeqAn 2.4.0).
 * Own code: nothing here is copied from the reference; the algorithms are re-expressed over flat arrays.
 */
#include "rpa_oracle.h"

#include <float.h>
#include <limits.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "blosum62_seqan.h"

/* ------------------------------------------------------------------------------------------------------------
 * a1  seqan/align/global_alignment_myers_impl.h:62-196  -- Myers bit-vector global score, 32-bit blocks.
 * The shorter sequence is the pattern (:70-71).  Per text character and block (:143-184):
 *   X = Peq|VN; D0 = ((VP + (X&VP) + carry) ^ VP) | X; HN = VP&D0; HP = VN | ~(VP|D0);
 *   X = (HP<<1)|carryHP (|1 in block 0: the top row of a *global* alignment grows by one per column);
 *   VN = X&D0; VP = (HN<<1)|carryHN | ~(X|D0);
 * and the score moves by -1/+1 with HP/HN at the last pattern row (:187-190).  The reference's score mask is
 * 1 << ((len_y % 32) - 1) (:94), i.e. a shift by -1 when len_y % 32 == 0; x86 masks the count to 31, which is
 * the intended bit, so the bit is written here as (len_y - 1) % 32.  len_y == 0 is out-of-bounds access in the
 * reference; here it returns -len_x (what the recurrence means).
 * ---------------------------------------------------------------------------------------------------------- */
int orc_myers_score(const uint8_t* h, uint32_t len_h, const uint8_t* v, uint32_t len_v) {
  if (len_h < len_v) return orc_myers_score(v, len_v, h, len_h);
  const uint8_t* x = h;
  const uint8_t* y = v;
  const uint32_t len_x = len_h, len_y = len_v;
  if (len_y == 0) return -(int)len_x;
  const uint32_t alphabet = 27; /* >= both alphabets; only rows that occur are read */
  const uint32_t blocks = (len_y + 31u) / 32u;
  uint32_t* mem = (uint32_t*)calloc((size_t)blocks * (alphabet + 2u), sizeof(uint32_t));
  if (!mem) return INT_MIN;
  uint32_t* VP = mem;
  uint32_t* VN = mem + blocks;
  uint32_t* peq = mem + 2u * blocks;
  for (uint32_t b = 0; b < blocks; ++b) VP[b] = 0xFFFFFFFFu;
  for (uint32_t j = 0; j < len_y; ++j) peq[blocks * y[j] + j / 32u] |= 1u << (j % 32u);
  const uint32_t score_mask = 1u << ((len_y - 1u) % 32u);
  int score = -(int)len_y;
  for (uint32_t pos = 0; pos < len_x; ++pos) {
    const uint32_t* eq = peq + blocks * x[pos];
    uint32_t carry_d0 = 0, carry_hp = 1, carry_hn = 0, HP = 0, HN = 0;
    for (uint32_t b = 0; b < blocks; ++b) {
      const uint32_t vp = VP[b], vn = VN[b];
      const uint32_t X = eq[b] | vn;
      const uint64_t sum = (uint64_t)vp + (uint64_t)(X & vp) + carry_d0;
      carry_d0 = (uint32_t)(sum >> 32);
      const uint32_t D0 = ((uint32_t)sum ^ vp) | X;
      HN = vp & D0;
      HP = vn | ~(vp | D0);
      const uint32_t Xs = (HP << 1) | carry_hp;
      carry_hp = HP >> 31;
      VN[b] = Xs & D0;
      const uint32_t hns = (HN << 1) | carry_hn;
      carry_hn = HN >> 31;
      VP[b] = hns | ~(Xs | D0);
    }
    if (HP & score_mask)
      --score;
    else if (HN & score_mask)
      ++score;
  }
  free(mem);
  return score;
}

/* Textbook unit-cost Levenshtein, two rolling rows: independent cross-check of a1. */
int orc_levenshtein(const uint8_t* a, uint32_t la, const uint8_t* b, uint32_t lb) {
  uint32_t* row = (uint32_t*)malloc(sizeof(uint32_t) * ((size_t)lb + 1));
  if (!row) return -1;
  for (uint32_t j = 0; j <= lb; ++j) row[j] = j;
  for (uint32_t i = 1; i <= la; ++i) {
    uint32_t diag = row[0];
    row[0] = i;
    for (uint32_t j = 1; j <= lb; ++j) {
      uint32_t up = row[j];
      uint32_t best = diag + (a[i - 1] != b[j - 1]);
      if (up + 1 < best) best = up + 1;
      if (row[j - 1] + 1 < best) best = row[j - 1] + 1;
      diag = up;
      row[j] = best;
    }
  }
  int d = (int)row[lb];
  free(row);
  return d;
}

/* a2  src/taxonpredictionmodelsequence.hh:133-171  getAlignmentDNA: lower-bound statistics from the distance. */
void orc_align_dna(const uint8_t* a, uint32_t la, const uint8_t* b, uint32_t lb, orc_alignment* out) {
  const uint8_t* lng = a;
  const uint8_t* sht = b;
  uint32_t llen = la, slen = lb;
  if (la < lb) { /* :144-147 */
    lng = b; llen = lb; sht = a; slen = la;
  }
  int mutualscore = -orc_myers_score(sht, slen, lng, llen); /* :150 */
  int lendiff = (int)llen - (int)slen;                       /* :155 */
  int gap = lendiff;                                         /* :157 */
  int mismatch = mutualscore - lendiff;                      /* :158 */
  int match = (int)slen - mismatch;                          /* :159 */
  out->distance = (float)mutualscore;                        /* :163 */
  out->similarity = (float)match;
  out->matches = (uint32_t)match;
  out->mismatches = (uint32_t)mismatch;
  out->gaps = (uint32_t)gap;
  out->length = (uint32_t)(match + mismatch + gap);
  out->selfscore = 0;
  out->mutualscore = mutualscore;
}

/* a4  seqan/align/dp_formula_linear.h:63-107,189-256 + seqan/align/dp_formula.h:153-163 (left operand wins ties:
 * vertical over horizontal, then diagonal over that) + seqan/align/dp_traceback_impl.h:379-412,498-548 (single
 * trace, GapsLeft; stop at the first row/column, the rest are leading gaps).  Scores: Blosum62, gap open =
 * gap extend = -1 (seqan/score/score_matrix.h:106-110).  A is the horizontal sequence (row 0 of the Align
 * object, src/taxonpredictionmodelsequence.hh:194-197). */
enum { TR_NONE = 0, TR_DIAG = 1, TR_VERT = 2, TR_HORI = 4 };

static int nw_self(const uint8_t* s, uint32_t n) {
  /* globalAlignmentScore(X, X, Blosum62, LinearGaps) by full DP, as the reference computes it (:190-191). */
  int* col = (int*)malloc(sizeof(int) * ((size_t)n + 1));
  if (!col) return INT_MIN;
  for (uint32_t i = 0; i <= n; ++i) col[i] = -(int)i;
  for (uint32_t j = 1; j <= n; ++j) {
    int diag = col[0];
    col[0] = -(int)j;
    for (uint32_t i = 1; i <= n; ++i) {
      int up_old = col[i];
      int d = diag + ORC_BLOSUM62[s[j - 1] * 27 + s[i - 1]];
      int v = col[i - 1] + ORC_GAP_SCORE;
      int h = col[i] + ORC_GAP_SCORE;
      int best = (v < h) ? h : v;
      best = (d < best) ? best : d;
      diag = up_old;
      col[i] = best;
    }
  }
  int r = col[n];
  free(col);
  return r;
}

int orc_align_protein(const uint8_t* a, uint32_t la, const uint8_t* b, uint32_t lb, orc_alignment* out) {
  const size_t W = (size_t)la + 1; /* columns: horizontal = A */
  const size_t Hh = (size_t)lb + 1; /* rows: vertical = B */
  int sa = nw_self(a, la), sb = nw_self(b, lb);
  if (sa == INT_MIN || sb == INT_MIN) return ORC_ERR_NOMEM;
  int selfscore = sa + sb; /* :190-191 */
  uint8_t* trace = (uint8_t*)malloc(W * Hh);
  int* col = (int*)malloc(sizeof(int) * Hh);
  if (!trace || !col) {
    free(trace); free(col);
    return ORC_ERR_NOMEM;
  }
  /* column 0 */
  col[0] = 0;
  trace[0] = TR_NONE;
  for (size_t i = 1; i < Hh; ++i) {
    col[i] = col[i - 1] + ORC_GAP_SCORE;
    trace[i * W] = TR_VERT;
  }
  for (size_t j = 1; j < W; ++j) {
    int diag = col[0];
    col[0] = col[0] + ORC_GAP_SCORE;
    trace[j] = TR_HORI;
    for (size_t i = 1; i < Hh; ++i) {
      int left = col[i];
      int d = diag + ORC_BLOSUM62[a[j - 1] * 27 + b[i - 1]];
      int v = col[i - 1] + ORC_GAP_SCORE;
      int h = left + ORC_GAP_SCORE;
      int best;
      uint8_t t;
      if (v < h) { best = h; t = TR_HORI; } else { best = v; t = TR_VERT; }
      if (!(d < best)) { best = d; t = TR_DIAG; }
      diag = left;
      col[i] = best;
      trace[i * W + j] = t;
    }
  }
  int mutualscore = col[Hh - 1];
  int gap = 0, match = 0, mismatch = 0;
  size_t i = lb, j = la;
  while (i > 0 && j > 0) {
    uint8_t t = trace[i * W + j];
    if (t == TR_DIAG) {
      if (a[j - 1] == b[i - 1]) ++match; else ++mismatch;
      --i; --j;
    } else if (t == TR_VERT) {
      ++gap; --i;
    } else {
      ++gap; --j;
    }
  }
  gap += (int)(i + j); /* leading gaps, dp_traceback_impl.h:541-547 */
  free(trace);
  free(col);
  unsigned len = (unsigned)(gap + match + mismatch);      /* :227 */
  float normfactor = len / (float)selfscore;               /* :228 */
  out->distance = (selfscore - 2 * mutualscore) * normfactor; /* :232 */
  out->similarity = (2 * mutualscore) * normfactor;        /* :233 */
  out->matches = (uint32_t)match;
  out->mismatches = (uint32_t)mismatch;
  out->gaps = (uint32_t)gap;
  out->length = len;
  out->selfscore = selfscore;
  out->mutualscore = mutualscore;
  if (!(selfscore >= mutualscore)) return ORC_ERR_ASSERT; /* :205 */
  if (!(out->distance >= 0)) return ORC_ERR_ASSERT;       /* :240 */
  return ORC_OK;
}

/* Same result computed the way the device kernel will: one forward pass carrying (#diagonal steps, #matches)
 * from the chosen predecessor, self scores as sums of the matrix diagonal (SURVEY a4 / Appendix E claim). */
int orc_align_protein_forward(const uint8_t* a, uint32_t la, const uint8_t* b, uint32_t lb, orc_alignment* out) {
  const size_t Hh = (size_t)lb + 1;
  int selfscore = 0;
  for (uint32_t k = 0; k < la; ++k) selfscore += ORC_BLOSUM62[a[k] * 27 + a[k]];
  for (uint32_t k = 0; k < lb; ++k) selfscore += ORC_BLOSUM62[b[k] * 27 + b[k]];
  int* sc = (int*)malloc(sizeof(int) * Hh);
  uint32_t* nd = (uint32_t*)calloc(Hh, sizeof(uint32_t));
  uint32_t* nm = (uint32_t*)calloc(Hh, sizeof(uint32_t));
  if (!sc || !nd || !nm) { free(sc); free(nd); free(nm); return ORC_ERR_NOMEM; }
  for (size_t i = 0; i < Hh; ++i) sc[i] = -(int)i;
  for (size_t j = 1; j <= la; ++j) {
    int dsc = sc[0]; uint32_t dnd = nd[0], dnm = nm[0];
    sc[0] = -(int)j;
    for (size_t i = 1; i < Hh; ++i) {
      int lsc = sc[i]; uint32_t lnd = nd[i], lnm = nm[i];
      int d = dsc + ORC_BLOSUM62[a[j - 1] * 27 + b[i - 1]];
      int v = sc[i - 1] + ORC_GAP_SCORE;
      int h = lsc + ORC_GAP_SCORE;
      int best; uint32_t bnd, bnm;
      if (v < h) { best = h; bnd = lnd; bnm = lnm; } else { best = v; bnd = nd[i - 1]; bnm = nm[i - 1]; }
      if (!(d < best)) { best = d; bnd = dnd + 1; bnm = dnm + (a[j - 1] == b[i - 1]); }
      dsc = lsc; dnd = lnd; dnm = lnm;
      sc[i] = best; nd[i] = bnd; nm[i] = bnm;
    }
  }
  int mutualscore = sc[lb];
  uint32_t ndiag = nd[lb], nmatch = nm[lb];
  free(sc); free(nd); free(nm);
  uint32_t gap = (la - ndiag) + (lb - ndiag);
  uint32_t mismatch = ndiag - nmatch;
  unsigned len = gap + nmatch + mismatch;
  float normfactor = len / (float)selfscore;
  out->distance = (selfscore - 2 * mutualscore) * normfactor;
  out->similarity = (2 * mutualscore) * normfactor;
  out->matches = nmatch; out->mismatches = mismatch; out->gaps = gap; out->length = len;
  out->selfscore = selfscore; out->mutualscore = mutualscore;
  return ORC_OK;
}

/* ------------------------------------------------------------------------------------------------------------
 * a8  src/taxonomyinterface.cpp:52-55 (isParentOf), :67-77 (getLCA, including its min(A.left, B.right)).
 * ---------------------------------------------------------------------------------------------------------- */
int orc_is_parent_of(const orc_taxonomy* t, uint32_t a, uint32_t b) {
  return t->right[a] > t->left[b] && t->left[a] < t->left[b];
}

uint32_t orc_lca(const orc_taxonomy* t, uint32_t a, uint32_t b) {
  uint32_t left_min = t->left[a] < t->right[b] ? t->left[a] : t->right[b];
  uint32_t right_max = t->right[a] > t->right[b] ? t->right[a] : t->right[b];
  uint32_t lca = a;
  while (t->left[lca] > left_min || t->right[lca] < right_max) lca = t->parent[lca];
  return lca;
}

/* ------------------------------------------------------------------------------------------------------------
 * a5  src/taxonpredictionmodelsequence.hh:259-323  BandFactor.  entries[0] = (0, anchor node).  entries[1..] are
 * sorted ascending by distance (:313-316; std::sort, unstable -- the result does not depend on the order of
 * ties, SURVEY Appendix E, re-checked by tests/test_oracle_placement.py); then the walk of :280-311 and the
 * sqrt of :276.  worstscore_per_rank is a std::map keyed by rank (0..255) -> array + presence flags.
 * ---------------------------------------------------------------------------------------------------------- */
float orc_bandfactor(const orc_taxonomy* t, const float* dist, const uint32_t* node, uint32_t n) {
  float bandfactor = 1.f; /* min_bandfactor, :280-281 */
  float worst[256];
  uint8_t have[256];
  memset(have, 0, sizeof(have));
  uint32_t* order = (uint32_t*)malloc(sizeof(uint32_t) * (n ? n : 1));
  if (!order) return -1.f;
  for (uint32_t i = 0; i < n; ++i) order[i] = i;
  for (uint32_t i = 2; i < n; ++i) { /* insertion sort of [1..n) by dist, stable */
    uint32_t k = order[i];
    uint32_t j = i;
    while (j > 1 && dist[order[j - 1]] > dist[k]) { order[j] = order[j - 1]; --j; }
    order[j] = k;
  }
  uint32_t anchor = node[0];
  uint8_t last_rank = t->depth[anchor]; /* :287 */
  worst[last_rank] = dist[0];
  have[last_rank] = 1;
  for (uint32_t e = 1; e < n; ++e) {
    float score = dist[order[e]];
    uint32_t lca = orc_lca(t, node[order[e]], anchor); /* :292 */
    uint8_t rank = t->depth[lca];
    if (rank == last_rank) {
    } else if (rank < last_rank) {
      worst[rank] = score; have[rank] = 1; last_rank = rank; /* :296-297 */
    } else { /* disorder, :298-308 */
      uint8_t r = (uint8_t)(rank - 1);
      do {
        if (have[r]) {
          float refscore = worst[r];
          if (refscore != 0.f) { float q = score / refscore; if (q > bandfactor) bandfactor = q; }
        }
      } while (r--);
    }
  }
  free(order);
  if (bandfactor > FLT_MAX) bandfactor = FLT_MAX; /* :310 */
  return sqrtf(bandfactor);                        /* :276 */
}

/* ------------------------------------------------------------------------------------------------------------
 * a7  src/taxonpredictionmodelsequence.hh:856-880 (getSequence / getSequenceDNA / getSequenceProtein) on top of
 * src/sequencestorage.hh:341-369,430-457 (indexed store: stop clipped to the sequence length, 1-based inclusive)
 * and seqan reverseComplement.
 * ---------------------------------------------------------------------------------------------------------- */
void orc_revcomp_dna5(uint8_t* s, uint32_t n) {
  for (uint32_t i = 0; i < n / 2; ++i) { uint8_t tmp = s[i]; s[i] = s[n - 1 - i]; s[n - 1 - i] = tmp; }
  for (uint32_t i = 0; i < n; ++i) if (s[i] < 4) s[i] = (uint8_t)(3 - s[i]); /* A<->T, C<->G, N stays */
}

uint32_t orc_segment_capacity(uint32_t start, uint32_t stop, uint32_t left_ext, uint32_t right_ext) {
  uint64_t lo = start <= stop ? start : stop, hi = start <= stop ? stop : start;
  return (uint32_t)(hi - lo + 1 + left_ext + right_ext);
}

static int store_get(const orc_store* st, uint32_t seq, uint32_t start, uint32_t stop, uint8_t* out, uint32_t* out_len) {
  if (seq >= st->n_seqs) return ORC_ERR_SEQ_ID;
  uint64_t len = st->offsets[seq + 1] - st->offsets[seq];
  if (start > stop) return ORC_ERR_ASSERT;           /* assert(start <= stop), sequencestorage.hh:343 */
  if (stop > len) stop = (uint32_t)len;               /* :353 */
  if (start < 1 || start > stop) return ORC_ERR_RANGE; /* assert(length(seq) == stop-start+1), :360 */
  memcpy(out, st->residues + st->offsets[seq] + (start - 1), (size_t)(stop - start + 1));
  *out_len = stop - start + 1;
  return ORC_OK;
}

int orc_get_segment(const orc_store* db, int alphabet, uint32_t seq, uint32_t start, uint32_t stop, uint32_t left_ext,
                    uint32_t right_ext, uint8_t* out, uint32_t* out_len) {
  if (alphabet == ORC_PROTEIN && start > stop) return ORC_ERR_ASSERT; /* getSequenceProtein assert(start <= stop), :864 */
  if (start <= stop) { /* :872-876 */
    uint32_t newstart = left_ext < start ? start - left_ext : 1;
    uint32_t newstop = stop + right_ext;
    return store_get(db, seq, newstart, newstop, out, out_len);
  }
  uint32_t newstart = right_ext < stop ? stop - right_ext : 1; /* :877-879 */
  uint32_t newstop = start + left_ext;
  int rc = store_get(db, seq, newstart, newstop, out, out_len);
  if (rc == ORC_OK) orc_revcomp_dna5(out, *out_len);
  return rc;
}

/* a9  src/alignmentsfilter.hh:171-190 + src/alignmentrecord.hh:89-93: list::sort (stable merge sort) with
 * "greater": a before b iff b < a, where x < y iff (score, identities) lexicographically smaller. */
static int rec_less(const orc_record* x, const orc_record* y) {
  if (x->score < y->score) return 1;
  if (x->score > y->score) return 0;
  return x->identities < y->identities;
}

/* `int x = <float>` as the reference's platform (x86-64) computes it: cvttss2si truncates and answers the "integer indefinite"
 * 0x80000000 for a NaN and for anything outside the int range (:602,657-659: `qsearchscore_thresh_heuristic = score * X`).  In C
 * that conversion is undefined behaviour, so the oracle spells the platform's answer out instead of relying on the compiler. */
static int orc_float_to_int(float v) { return (v >= -2147483648.f && v < 2147483648.f) ? (int)v : INT_MIN; }

void orc_sort_records(const orc_record* recs, uint32_t n, uint32_t* order) {
  for (uint32_t i = 0; i < n; ++i) order[i] = i;
  for (uint32_t i = 1; i < n; ++i) { /* stable insertion sort: move left only past strictly smaller elements */
    uint32_t k = order[i];
    uint32_t j = i;
    while (j > 0 && rec_less(&recs[order[j - 1]], &recs[k])) { order[j] = order[j - 1]; --j; }
    order[j] = k;
  }
}

/* ------------------------------------------------------------------------------------------------------------
 * a6  src/taxonpredictionmodelsequence.hh:341-838  RPAPredictionModel::predict, for one record set.
 * Types follow the reference exactly where they change results: float (f32) distances/similarities/scores,
 * double (f64) percent-identity thresholds, int truncation of outgroup distances (:592,661) and of the pass-1
 * score cut-off (:602,659), float score cut-off in pass 2 (:754).
 * ---------------------------------------------------------------------------------------------------------- */
typedef struct {
  const orc_taxonomy* tax;
  const orc_store* q;
  const orc_store* db;
  int alphabet;
  uint32_t n;
  const orc_record** rec; /* sorted active records */
  uint8_t** seg;          /* lazily fetched segments */
  uint32_t* seglen;
  const uint8_t* qrseq;
  uint32_t qrlen;
  uint32_t qrstart, qrstop;
  orc_trace_fn trace;
  void* ctx;
} pctx;

static int fetch_segment(pctx* c, uint32_t i) { /* :523,626-627,774-775,794 */
  if (c->seg[i]) return ORC_OK; /* seqan::empty(segments[i]) */
  const orc_record* r = c->rec[i];
  uint32_t left_ext = r->qstart - c->qrstart, right_ext = c->qrstop - r->qstop;
  uint32_t cap = orc_segment_capacity(r->rstart, r->rstop, left_ext, right_ext);
  c->seg[i] = (uint8_t*)malloc(cap ? cap : 1);
  if (!c->seg[i]) return ORC_ERR_NOMEM;
  return orc_get_segment(c->db, c->alphabet, r->ref_seq, r->rstart, r->rstop, left_ext, right_ext, c->seg[i], &c->seglen[i]);
}

static int do_align(pctx* c, const uint8_t* a, uint32_t la, const uint8_t* b, uint32_t lb, orc_alignment* aln) { /* :244-256 */
  if (c->alphabet == ORC_DNA) { orc_align_dna(a, la, b, lb, aln); return ORC_OK; }
  return orc_align_protein(a, la, b, lb, aln);
}

#define SETBIT(s, i) ((s)[(i) >> 3] |= (uint8_t)(1u << ((i)&7)))
#define CLRBIT(s, i) ((s)[(i) >> 3] &= (uint8_t)~(1u << ((i)&7)))
#define GETBIT(s, i) (((s)[(i) >> 3] >> ((i)&7)) & 1u)
static int first_bit(const uint8_t* s, uint32_t n) { for (uint32_t i = 0; i < n; ++i) if (GETBIT(s, i)) return (int)i; return -1; }

int orc_predict(const orc_taxonomy* tax, const orc_store* qstore, const orc_store* dbstore, int alphabet,
                float exclude_factor, float reeval_bandwidth, uint32_t query_seq, uint32_t query_len,
                const orc_record* recs, uint32_t nrecs, orc_prediction* out, orc_stats* stats, orc_trace_fn trace,
                void* ctx) {
  const uint32_t root = 0;
  const float exclude_alignments_factor_ = exclude_factor;                 /* :333 */
  const float reeval_bandwidth_factor_ = (float)(1. - reeval_bandwidth);   /* :334 (double arithmetic, then float) */
  int rc = ORC_OK;
  orc_stats st = {0, 0, 0, 0};
  out->begin = 1; out->end = query_len; /* initPredictionRecord, src/taxonpredictionmodel.hh:42-44 */
  out->flags = 0;
  out->ival = 0.f; out->lower = out->upper = out->rtax = root; out->support = 0;

  uint32_t n = 0; /* :350-356 */
  for (uint32_t i = 0; i < nrecs; ++i) if (!recs[i].filtered) ++n;
  st.n = n;
  if (n == 0) { /* :359-368 */
    out->lower = out->upper = out->rtax = root; out->support = 0; out->flags |= ORC_FLAG_IVAL_UNTOUCHED;
    if (stats) *stats = st;
    return ORC_OK;
  }
  orc_record* act = (orc_record*)malloc(sizeof(orc_record) * n);
  uint32_t* order = (uint32_t*)malloc(sizeof(uint32_t) * n);
  if (!act || !order) { free(act); free(order); return ORC_ERR_NOMEM; }
  for (uint32_t i = 0, k = 0; i < nrecs; ++i) if (!recs[i].filtered) act[k++] = recs[i];
  if (n == 1) { /* :371-388 */
    out->begin = act[0].qstart; out->end = act[0].qstop; out->ival = 1.f;
    out->lower = act[0].node; out->upper = root; out->support = act[0].identities; out->rtax = act[0].node;
    out->flags |= ORC_FLAG_SINGLE;
    free(act); free(order);
    if (stats) *stats = st;
    return ORC_OK;
  }
  uint32_t qrstart = act[0].qstart, qrstop = act[0].qstop; /* :391-404 */
  for (uint32_t i = 1; i < n; ++i) {
    if (act[i].qstart < qrstart) qrstart = act[i].qstart;
    if (act[i].qstop > qrstop) qrstop = act[i].qstop;
  }
  const uint32_t qrlength = qrstop - qrstart + 1;
  for (uint32_t i = 0; i < n; ++i)
    if (act[i].score != act[i].score) { free(act); free(order); return ORC_ERR_NAN_SCORE; } /* no ordering to restate: see the header */
  orc_sort_records(act, n, order); /* :412 */

  pctx c;
  memset(&c, 0, sizeof(c));
  c.tax = tax; c.q = qstore; c.db = dbstore; c.alphabet = alphabet; c.n = n; c.trace = trace; c.ctx = ctx;
  c.qrstart = qrstart; c.qrstop = qrstop;
  c.rec = (const orc_record**)malloc(sizeof(orc_record*) * n);
  c.seg = (uint8_t**)calloc(n, sizeof(uint8_t*));
  c.seglen = (uint32_t*)calloc(n, sizeof(uint32_t));
  float* querydistance = (float*)malloc(sizeof(float) * n);
  float* querysimilarity = (float*)malloc(sizeof(float) * n);
  const uint32_t setbytes = (n + 7) / 8;
  uint8_t* qgroup = (uint8_t*)calloc(setbytes, 1);
  uint8_t* outgroup = (uint8_t*)calloc(setbytes, 1);
  float* bf_dist = (float*)malloc(sizeof(float) * (n + 1));
  uint32_t* bf_node = (uint32_t*)malloc(sizeof(uint32_t) * (n + 1));
  uint32_t* og_idx = (uint32_t*)malloc(sizeof(uint32_t) * n);
  int* og_dist = (int*)malloc(sizeof(int) * n);
  uint8_t* qrbuf = (uint8_t*)malloc(qrlength ? qrlength : 1);
  if (!c.rec || !c.seg || !c.seglen || !querydistance || !querysimilarity || !qgroup || !outgroup || !bf_dist ||
      !bf_node || !og_idx || !og_dist || !qrbuf) { rc = ORC_ERR_NOMEM; goto done; }
  for (uint32_t i = 0; i < n; ++i) c.rec[i] = &act[order[i]];
#define REC(i) (c.rec[(i)])

  { /* :415  query range sequence (query store getSequence, forward only) */
    uint32_t got = 0;
    rc = store_get(qstore, query_seq, qrstart, qrstop, qrbuf, &got);
    if (rc != ORC_OK) goto done;
    c.qrseq = qrbuf; c.qrlen = got;
  }
  const float qmax_searchscore = REC(0)->score; /* :427 */

  if (REC(0)->alnlen == qrlength && REC(0)->identities == qrlength) { /* :431-472 shortcut */
    float searchscore_best = REC(0)->score;
    uint32_t lnode = REC(0)->node, unode;
    uint32_t i = 1;
    for (;;) {
      if (i == n) { unode = root; break; }
      float searchscore = REC(i)->score;
      if (searchscore == searchscore_best) {
        lnode = orc_lca(tax, lnode, REC(i)->node);
      } else {
        float uscore = searchscore;
        unode = lnode;
        do { unode = orc_lca(tax, unode, REC(i)->node); } while (++i < n && REC(i)->score == uscore);
        break;
      }
      ++i;
    }
    out->begin = qrstart; out->end = qrstop; out->ival = 0.f;
    out->lower = lnode; out->upper = unode; out->support = qrlength; out->rtax = lnode;
    out->flags |= ORC_FLAG_SHORTCUT;
    goto done;
  }

  for (uint32_t i = 0; i < n; ++i) { querydistance[i] = FLT_MAX; querysimilarity[i] = 0.f; } /* :474-476 */

  uint32_t anchors_support = 0; /* :493 */
  uint32_t rtax;
  uint32_t lca_allnodes = REC(0)->node; /* :495 */

  { /* pass 0, :497-566 */
    float dbalignment_searchscore_threshold = reeval_bandwidth_factor_ * qmax_searchscore; /* :499 */
    uint32_t index_best = 0;
    for (uint32_t i = 0; i < n; ++i) {
      float dist, sim;
      const float qsearchscore = REC(i)->score;
      if (REC(i)->alnlen == qrlength && REC(i)->identities == qrlength) { /* :511-517 */
        SETBIT(qgroup, i);
        dist = 0;
        sim = (float)REC(i)->identities;
      } else if (REC(i)->score >= dbalignment_searchscore_threshold) { /* :518-534 */
        SETBIT(qgroup, i);
        rc = fetch_segment(&c, i);
        if (rc != ORC_OK) goto done;
        orc_alignment aln;
        rc = do_align(&c, c.seg[i], c.seglen[i], c.qrseq, c.qrlen, &aln); /* :526 */
        if (rc != ORC_OK) goto done;
        dist = aln.distance;
        ++st.pass0;
        float idf = (float)REC(i)->identities;
        sim = aln.similarity < idf ? idf : aln.similarity; /* std::max(a,b) = a unless a<b, :531 */
        if (trace) trace(ctx, 0, (int)i, -1, dist, sim);
      } else { /* :536-539 */
        dist = FLT_MAX;
        sim = (float)REC(i)->identities;
      }
      querydistance[i] = dist;
      querysimilarity[i] = sim;
      if (dist < querydistance[index_best]) index_best = i; /* :542-546 */
      else if (dist == querydistance[index_best]) {
        if (sim > querysimilarity[index_best]) index_best = i;
        else if (sim == querysimilarity[index_best] && qsearchscore > REC(index_best)->score) index_best = i;
      }
      uint32_t simu = (uint32_t)sim; /* static_cast<large_unsigned_int>(sim), :547 */
      if (simu > anchors_support) anchors_support = simu;
      lca_allnodes = orc_lca(tax, lca_allnodes, REC(i)->node); /* :548 */
    }
    rtax = REC(index_best)->node; /* :553 */
    int any = 0;
    for (uint32_t i = 0; i < n; ++i) { /* :554-562, ascending like std::set */
      if (!GETBIT(qgroup, i)) continue;
      if (querydistance[i] != querydistance[index_best] || querysimilarity[i] != querysimilarity[index_best] ||
          REC(i)->score != REC(index_best)->score) {
        CLRBIT(qgroup, i);
      } else {
        rtax = orc_lca(tax, rtax, REC(i)->node);
        any = 1;
      }
    }
    if (!any) { rc = ORC_ERR_ASSERT; goto done; } /* assert(!qgroup.empty()), :563 */
  }

  float ival_global = 0.f; /* :570-574 */
  uint32_t lnode_global = rtax, unode_global = rtax;
  float bandfactor_max = 1.f;

  { /* pass 1, :576-733 */
    uint8_t lca_root_dist_min = 255; /* :579 */
    do {
      const uint32_t index_anchor = (uint32_t)first_bit(qgroup, n); /* :582-583 */
      CLRBIT(qgroup, index_anchor);
      const float qdist = querydistance[index_anchor];
      const uint32_t rnode = REC(index_anchor)->node;
      uint32_t nbf = 0;
      bf_dist[nbf] = 0.f; bf_node[nbf] = rnode; ++nbf; /* :586 */
      uint32_t lnode = rtax;
      int have_unode = 0; uint32_t unode = root;
      float ldist = 0.f;
      float udist = FLT_MAX;
      uint32_t nog = 0;
      double qpid_upper = 0., qpid_thresh_guarantee = 0., qpid_thresh_heuristic = 0.; /* :599-601 */
      int qsearchscore_thresh_heuristic = 0;                                          /* :602 */

      for (uint32_t i = 0; lnode != root && i < n && REC(i)->score >= (float)qsearchscore_thresh_heuristic; ++i) { /* :604 */
        const uint32_t cnode = REC(i)->node;
        const uint32_t qsearchmatch = REC(i)->identities;
        const double qsearchpid = (double)qsearchmatch / qrlength;
        const double qpid = (double)querysimilarity[i] / qrlength;
        double qpid_thresh = qpid_thresh_guarantee < qpid_thresh_heuristic ? qpid_thresh_heuristic : qpid_thresh_guarantee;
        if (qpid >= qpid_thresh) { /* :612 */
          float dist;
          if (i == index_anchor) dist = 0.f;
          else if (querydistance[i] == 0.f) dist = querydistance[index_anchor]; /* :620-621 */
          else {
            rc = fetch_segment(&c, index_anchor); if (rc != ORC_OK) goto done;
            rc = fetch_segment(&c, i); if (rc != ORC_OK) goto done;
            orc_alignment aln;
            rc = do_align(&c, c.seg[i], c.seglen[i], c.seg[index_anchor], c.seglen[index_anchor], &aln); /* :630 */
            if (rc != ORC_OK) goto done;
            dist = aln.distance;
            ++st.pass1;
            if (trace) trace(ctx, 1, (int)i, (int)index_anchor, dist, aln.similarity);
          }
          bf_dist[nbf] = dist; bf_node[nbf] = cnode; ++nbf; /* :641 */
          if (dist == 0.f) CLRBIT(qgroup, i); /* :644 */
          else if (dist <= qdist) { /* :646-650 */
            lnode = orc_lca(tax, lnode, cnode);
            if (dist > ldist) ldist = dist;
          } else { /* :651-662 */
            if (dist < udist) {
              udist = dist;
              if (qsearchpid > qpid_upper) {
                qpid_upper = qsearchpid;
                qpid_thresh_guarantee = qsearchpid * 2. - 1.;
                qpid_thresh_heuristic = qsearchpid * exclude_alignments_factor_;
              }
              if (!qsearchscore_thresh_heuristic) qsearchscore_thresh_heuristic = orc_float_to_int(REC(i)->score * exclude_alignments_factor_);
            }
            og_idx[nog] = i; og_dist[nog] = (int)dist; ++nog; /* tuple<uint,int>: truncation, :661 */
          }
        }
      }

      float bandfactor = orc_bandfactor(tax, bf_dist, bf_node, nbf); /* :667 */
      if (bandfactor > bandfactor_max) bandfactor_max = bandfactor;   /* :668 */
      float qdist_ex = qdist * bandfactor;                            /* :669 */
      float min_upper_dist = (float)INT_MAX;                          /* :670 */
      { /* :673-687, erase-in-place on the list */
        uint32_t w = 0;
        for (uint32_t k = 0; k < nog; ++k) {
          float dist = (float)og_dist[k];
          int keep = 1;
          if (dist > qdist_ex) {
            if (dist > min_upper_dist) keep = 0;
            else if (dist < min_upper_dist) min_upper_dist = dist;
          } else {
            if (min_upper_dist > qdist_ex) min_upper_dist = dist;
            else min_upper_dist = min_upper_dist < dist ? dist : min_upper_dist;
          }
          if (keep) { og_idx[w] = og_idx[k]; og_dist[w] = og_dist[k]; ++w; }
        }
        nog = w;
      }
      if (min_upper_dist != FLT_MAX) { unode = lnode; have_unode = 1; } /* :690 (always true: INT_MAX != FLT_MAX) */
      for (uint32_t k = 0; k < nog; ++k) { /* :691-711 */
        uint32_t i = og_idx[k];
        float dist = (float)og_dist[k];
        uint32_t cnode = REC(i)->node;
        if (dist > min_upper_dist) continue;
        unode = orc_lca(tax, cnode, unode);
        const uint8_t lca_root_dist = tax->depth[orc_lca(tax, cnode, rtax)];
        if (lca_root_dist > lca_root_dist_min) continue;
        else if (lca_root_dist < lca_root_dist_min) {
          lca_root_dist_min = lca_root_dist;
          memset(outgroup, 0, setbytes);
        }
        SETBIT(outgroup, i);
      }
      float ival = 0.f; /* :714-719 */
      if (!have_unode) { unode = root; udist = -1; ival = 1.f; }
      else if (unode != lnode && ldist < qdist) ival = (qdist - ldist) / (udist - ldist);
      if (ival_global < ival) ival_global = ival;                /* std::max(ival, ival_global), :724 */
      unode_global = orc_lca(tax, unode_global, unode);          /* :726 */
      lnode_global = orc_lca(tax, lnode_global, lnode);          /* :727 */
    } while (first_bit(qgroup, n) >= 0 && lnode_global != root); /* :729 */
  }

  { /* pass 2, :737-822 */
    int a;
    while ((a = first_bit(outgroup, n)) >= 0) {
      const uint32_t index_anchor = (uint32_t)a;
      CLRBIT(outgroup, index_anchor);
      if (unode_global == lca_allnodes) continue; /* :743-747 */
      const double qpid_anchor = (double)querysimilarity[index_anchor] / qrlength;
      const double qpid_thresh_guarantee = qpid_anchor * 2. - 1.;
      const double qpid_thresh_heuristic = qpid_anchor * exclude_alignments_factor_;
      const double qpid_thresh = qpid_thresh_guarantee < qpid_thresh_heuristic ? qpid_thresh_heuristic : qpid_thresh_guarantee;
      const float qsearchscore_thresh_heuristic = REC(index_anchor)->score * exclude_alignments_factor_; /* :754 */
      for (uint32_t i = 0; i < n && REC(i)->score >= qsearchscore_thresh_heuristic; ++i) { /* :757 */
        const double qpid = (double)querysimilarity[i] / qrlength;
        if (qpid >= qpid_thresh) {
          const uint32_t cnode = REC(i)->node;
          float dist;
          if (i == index_anchor) dist = 0.f;
          else {
            if (orc_is_parent_of(tax, unode_global, cnode) || cnode == unode_global) continue; /* :771 */
            rc = fetch_segment(&c, index_anchor); if (rc != ORC_OK) goto done;
            rc = fetch_segment(&c, i); if (rc != ORC_OK) goto done;
            orc_alignment aln;
            rc = do_align(&c, c.seg[i], c.seglen[i], c.seg[index_anchor], c.seglen[index_anchor], &aln); /* :778 */
            if (rc != ORC_OK) goto done;
            dist = aln.distance;
            ++st.pass2;
            if (trace) trace(ctx, 2, (int)i, (int)index_anchor, dist, aln.similarity);
            querydistance[i] = dist; /* :785 */
          }
          if (dist == 0.f) CLRBIT(outgroup, i); /* :789 */
          else {
            float qdist_ex;
            if (querydistance[index_anchor] == FLT_MAX) { /* :792-808 */
              rc = fetch_segment(&c, index_anchor); if (rc != ORC_OK) goto done;
              orc_alignment aln;
              rc = do_align(&c, c.seg[index_anchor], c.seglen[index_anchor], c.qrseq, c.qrlen, &aln); /* :797 */
              if (rc != ORC_OK) goto done;
              float d2 = aln.distance;
              float s2 = aln.similarity < querysimilarity[index_anchor] ? querysimilarity[index_anchor] : aln.similarity;
              if (trace) trace(ctx, 2, -1, (int)index_anchor, d2, s2);
              querydistance[index_anchor] = d2;
              querysimilarity[index_anchor] = s2;
              qdist_ex = d2 * bandfactor_max;
              ++st.pass2;
            } else qdist_ex = querydistance[index_anchor] * bandfactor_max; /* :809 */
            if (dist <= qdist_ex) unode_global = orc_lca(tax, unode_global, cnode); /* :811-813 */
          }
        }
      }
    }
  }

  if (unode_global == lnode_global) ival_global = 1.f; /* :824 */
  out->begin = qrstart; out->end = qrstop; out->ival = ival_global; /* :828-833 */
  out->lower = lnode_global; out->upper = unode_global; out->support = anchors_support; out->rtax = rtax;
  /* setNodeRange assert (src/predictionrecord.hh:153) */
  if (!(out->lower == out->upper || orc_is_parent_of(tax, out->upper, out->lower))) rc = ORC_ERR_ASSERT;

done:
  if (c.seg) for (uint32_t i = 0; i < n; ++i) free(c.seg[i]);
  free(c.seg); free(c.seglen); free((void*)c.rec); free(querydistance); free(querysimilarity); free(qgroup);
  free(outgroup); free(bf_dist); free(bf_node); free(og_idx); free(og_dist); free(qrbuf); free(act); free(order);
  if (stats) *stats = st;
  return rc;
#undef REC
}

int orc_predict_batch(const orc_taxonomy* tax, const orc_store* qstore, const orc_store* dbstore, int alphabet,
                      float exclude_factor, float reeval_bandwidth, uint32_t n_sets, const uint32_t* set_offsets,
                      const uint32_t* query_seq, const uint32_t* query_len, const orc_record* recs,
                      orc_prediction* out, orc_stats* stats) {
  for (uint32_t s = 0; s < n_sets; ++s) {
    int rc = orc_predict(tax, qstore, dbstore, alphabet, exclude_factor, reeval_bandwidth, query_seq[s], query_len[s],
                         recs + set_offsets[s], set_offsets[s + 1] - set_offsets[s], out + s, stats ? stats + s : 0, 0, 0);
    if (rc != ORC_OK) return rc;
  }
  return ORC_OK;
}
