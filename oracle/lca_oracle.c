/* oracle/lca_oracle.c -- TEST INFRASTRUCTURE ONLY (same rules as rpa_oracle.c).
 *
 * Plain-C restatement of the reference's alignment-free prediction models, line references into
 * /root/reference/core/src:
 *   DummyPredictionModel                     taxonpredictionmodel.hh:57-66
 *   LCASimplePredictionModel<.., false>      taxonpredictionmodel.hh:70-132
 *   MeganLCAPredictionModel<.., false>       taxonpredictionmodel.hh:136-165 with MinScoreMaxEvalueTopPercentFilter
 *                                            (alignmentsfilter.hh:349-386) and RemoveUnclassifiedFilter (:612-626)
 *   NBestLCAPredictionModel<.., false>       taxonpredictionmodel.hh:237-253 with NumBestBitscoreFilter
 *                                            (alignmentsfilter.hh:493-532)
 *   TaxonomyInterface::getLCA(container)     taxonomyinterface.hh:61-74 (a fold of the pairwise getLCA)
 * as selected by core/taxator.cpp:346-361.  PARITY UNPINNED: the reference has no test, fixture or golden output for
 * these models and cannot be built here (Boost); this file follows the source text only.
 *
 * None of the models touches the interpolation value: the PredictionRecord keeps the -1 it was constructed with
 * (predictionrecord.hh:40), so no ";ival=" is ever printed.  Rows carry ival = -1 and ORC_FLAG_IVAL_UNTOUCHED.
 */
#include <stdlib.h>
#include <string.h>

#include "rpa_oracle.h"

static void set_unclassified(orc_prediction* out, uint32_t query_len) { /* initPredictionRecord + setUnclassified, :42-51 */
  out->begin = 1; out->end = query_len;
  out->lower = out->upper = out->rtax = 0; out->support = 0;
  out->ival = -1.f; out->flags = ORC_FLAG_IVAL_UNTOUCHED;
}

/* LCASimplePredictionModel::predict over the records whose mask[] is 0, taxonpredictionmodel.hh:73-131 */
static void lca_simple(const orc_taxonomy* tax, uint32_t query_len, const orc_record* recs, const uint8_t* mask, uint32_t n,
                       orc_prediction* out) {
  set_unclassified(out, query_len);
  uint32_t i = 0;
  while (i < n && mask[i]) ++i; /* firstUnmaskedIter */
  if (i == n) return;           /* :129 */
  uint32_t qrstart = recs[i].qstart, qrstop = recs[i].qstop;
  float maxscore = recs[i].score;
  if (qrstart > qrstop) { uint32_t t = qrstart; qrstart = qrstop; qrstop = t; } /* :88 */
  uint32_t node = recs[i].node; /* LCA of refnodes, folded as the records come (the fold is order independent) */
  for (uint32_t k = i + 1; k < n; ++k) { /* :94-107 */
    if (mask[k]) continue;
    if (recs[k].qstart <= recs[k].qstop) {
      if (recs[k].qstart < qrstart) qrstart = recs[k].qstart;
      if (recs[k].qstop > qrstop) qrstop = recs[k].qstop;
    } else {
      if (recs[k].qstop < qrstart) qrstart = recs[k].qstop;
      if (recs[k].qstart > qrstop) qrstop = recs[k].qstart;
    }
    if (recs[k].score > maxscore) maxscore = recs[k].score;
    node = orc_lca(tax, node, recs[k].node);
  }
  out->begin = qrstart; out->end = qrstop; /* :108-109 */
  /* best reference taxon: LCA of the nodes of the records with the maximum score (:113-126; when the two node sets
     have the same size they are the same set, so the size test only saves a call) */
  int have = 0;
  uint32_t best = 0;
  for (uint32_t k = 0; k < n; ++k) {
    if (mask[k] || !(recs[k].score == maxscore)) continue;
    best = have ? orc_lca(tax, best, recs[k].node) : recs[k].node;
    have = 1;
  }
  out->lower = out->upper = node;
  out->support = qrstop - qrstart + 1; /* setNodePoint(node): support = getQueryFeatureWidth(), predictionrecord.hh:68,178-181 */
  out->rtax = have ? best : node;
}

typedef struct { float score; uint32_t idx; } scored;
static int by_score_desc_stable(const void* a, const void* b) {
  const scored* x = (const scored*)a; const scored* y = (const scored*)b;
  if (x->score > y->score) return -1;
  if (x->score < y->score) return 1;
  return x->idx < y->idx ? -1 : (x->idx > y->idx ? 1 : 0); /* multimap keeps insertion order among equal keys */
}

int orc_lca_predict(const orc_taxonomy* tax, const orc_lca_params* p, uint32_t query_len, const orc_record* recs,
                    const double* evalue, const uint8_t* node_unclassified, uint32_t n, orc_prediction* out) {
  uint8_t* mask = (uint8_t*)malloc(n ? n : 1);
  if (!mask) return ORC_ERR_NOMEM;
  for (uint32_t k = 0; k < n; ++k) mask[k] = recs[k].filtered ? 1 : 0;
  int rc = ORC_OK;
  switch (p->model) {
    case ORC_LCA_DUMMY: /* :61-64 */
      set_unclassified(out, query_len);
      break;
    case ORC_LCA_SIMPLE:
      lca_simple(tax, query_len, recs, mask, n, out);
      break;
    case ORC_LCA_MEGAN: { /* :144-154 */
      /* MinScoreMaxEvalueTopPercentFilter::filter, alignmentsfilter.hh:353-376.  minscore, maxevalue and toppercent
         are floats there (ctor :351; the double -e value is narrowed), support counts how often the running maximum
         rose, and the running maximum starts at 0. */
      const float minscore = p->minscore, maxevalue = (float)p->maxevalue, toppercent = p->toppercent;
      float max_bitscore = .0f;
      unsigned int support = 0;
      for (uint32_t k = 0; k < n; ++k) {
        if (mask[k]) continue;
        if (recs[k].score < minscore || (evalue ? evalue[k] : 0.) > (double)maxevalue) mask[k] = 1;
        else if (recs[k].score > max_bitscore) { max_bitscore = recs[k].score; support++; }
      }
      max_bitscore = (float)((1.0 - (double)toppercent) * (double)max_bitscore); /* :369 */
      for (uint32_t k = 0; k < n; ++k)
        if (recs[k].score < max_bitscore) mask[k] = 1; /* :371-375, masked records included */
      if (p->ignore_unclassified) /* RemoveUnclassifiedFilter, :616-620 */
        for (uint32_t k = 0; k < n; ++k)
          if (node_unclassified && node_unclassified[recs[k].node]) mask[k] = 1;
      if (support >= (unsigned int)p->minsupport) lca_simple(tax, query_len, recs, mask, n, out);
      else set_unclassified(out, query_len);
      break;
    }
    case ORC_LCA_NBEST: { /* :245-249 with NumBestBitscoreFilter::filter, alignmentsfilter.hh:497-527 */
      scored* s = (scored*)malloc(sizeof(scored) * (n ? n : 1));
      if (!s) { rc = ORC_ERR_NOMEM; break; }
      uint32_t m = 0;
      for (uint32_t k = 0; k < n; ++k)
        if (!mask[k]) { s[m].score = recs[k].score; s[m].idx = k; ++m; }
      if (m) {
        qsort(s, m, sizeof(scored), by_score_desc_stable);
        unsigned int count = (unsigned int)p->nbest; /* :509 */
        uint32_t j = 1;
        float lastvalue = s[0].score;
        for (; j < m; ++j) {
          if (s[j].score != lastvalue) {
            if (--count <= 0) break; /* unsigned: only ever true at exactly 0 (nbest = 0 wraps and spares everything) */
            lastvalue = s[j].score;
          }
        }
        for (; j < m; ++j) mask[s[j].idx] = 1;
      }
      free(s);
      lca_simple(tax, query_len, recs, mask, n, out);
      break;
    }
    default:
      rc = ORC_ERR_ASSERT;
  }
  free(mask);
  return rc;
}

int orc_lca_predict_batch(const orc_taxonomy* tax, const orc_lca_params* p, uint32_t n_sets, const uint32_t* set_offsets,
                          const uint32_t* query_len, const orc_record* recs, const double* evalue,
                          const uint8_t* node_unclassified, orc_prediction* out) {
  for (uint32_t s = 0; s < n_sets; ++s) {
    const uint32_t lo = set_offsets[s], hi = set_offsets[s + 1];
    int rc = orc_lca_predict(tax, p, query_len[s], recs + lo, evalue ? evalue + lo : NULL, node_unclassified, hi - lo, out + s);
    if (rc) return rc;
  }
  return ORC_OK;
}
