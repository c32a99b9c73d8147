// taxator-tk_amd/csrc/lca_kernels.hip -- the reference's alignment-free prediction models on the device.
//
// What it replaces (all in /root/reference/core/src): DummyPredictionModel (taxonpredictionmodel.hh:57-66),
// LCASimplePredictionModel (:70-132), MeganLCAPredictionModel (:136-165) with MinScoreMaxEvalueTopPercentFilter
// (alignmentsfilter.hh:349-386) and RemoveUnclassifiedFilter (:612-626), NBestLCAPredictionModel (:237-253) with
// NumBestBitscoreFilter (alignmentsfilter.hh:493-532), as selected by taxator -a (core/taxator.cpp:346-361).
// One thread per record set; the set's records are read straight from the batch columns.  The filters of the
// reference mark records ("filterOut"); here a record's mask is a pure function of the set-level quantities the
// filters compute (running maximum, n-th best distinct score), so no per-record scratch is written.  The LCA of a set
// of taxa is found by climbing from the running LCA until its preorder interval holds the next taxon (any depth).
// None of these models sets the interpolation value: rows carry ival = -1 and RPA_FLAG_IVAL_UNTOUCHED.
#include "rpa_device.h"

namespace rpa {

struct LcaRow {  // = rpa_prediction
  uint32_t lower, upper, rtax, support, begin, end;
  float ival;
  uint32_t flags;
};

// LCA of a running node `lca` and node x: climb from `lca` until its subtree holds x (the reference's getLCA walks the
// same way, core/src/taxonomyinterface.cpp:67-77).  Preorder intervals made at rpa_create; any depth.
__device__ __forceinline__ uint32_t lca_with(const TaxDev& tax, uint32_t lca, uint32_t x) {
  const uint32_t px = tax.pre[x];
  while (!(tax.pre[lca] <= px && px <= tax.last[lca])) lca = tax.parent[lca];  // the root's interval holds every node
  return lca;
}

__global__ __launch_bounds__(kBlock) void k_lca_models(BatchDev b, TaxDev tax, LcaParamsDev p, const double* __restrict__ evalue,
                                                        const uint8_t* __restrict__ unclassified, LcaRow* __restrict__ rows) {
  const uint32_t s = blockIdx.x * kBlock + threadIdx.x;
  if (s >= b.n_sets) return;
  const uint32_t lo = b.set_offsets[s], hi = b.set_offsets[s + 1];
  LcaRow row;  // initPredictionRecord + setUnclassified (taxonpredictionmodel.hh:42-51)
  row.begin = 1; row.end = b.query_len[s]; row.lower = row.upper = row.rtax = kRoot; row.support = 0;
  row.ival = -1.f; row.flags = 1u /* RPA_FLAG_IVAL_UNTOUCHED */;
  bool run_simple = p.model != 0;
  // set-level quantities that decide a record's mask
  float thr = 0.f;        // megan: (1 - toppercent) * running maximum
  float cutoff = 0.f;     // n-best: smallest score that is spared
  bool have_cutoff = false;
  if (p.model == 2) {  // MinScoreMaxEvalueTopPercentFilter::filter, alignmentsfilter.hh:353-376
    float max_bitscore = .0f;
    unsigned int support = 0;
    for (uint32_t r = lo; r < hi; ++r) {
      if (b.filtered && b.filtered[r]) continue;
      const float sc = b.score[r];
      if (sc < p.minscore || (evalue ? evalue[r] : 0.) > (double)p.maxevalue) continue;
      if (sc > max_bitscore) { max_bitscore = sc; support++; }
    }
    thr = (float)((1.0 - (double)p.toppercent) * (double)max_bitscore);
    if (!(support >= p.minsupport)) run_simple = false;  // :147-153
  } else if (p.model == 3) {  // NumBestBitscoreFilter::filter, alignmentsfilter.hh:497-527: spare the nbest highest distinct scores
    // walk the distinct score values downwards; `count` is unsigned in the reference, nbest == 0 wraps and spares all
    unsigned int count = p.nbest;
    bool any = false;
    float last = 0.f;
    for (uint32_t r = lo; r < hi; ++r) {
      if (b.filtered && b.filtered[r]) continue;
      const float sc = b.score[r];
      if (!any || sc > last) { last = sc; any = true; }
    }
    if (any) {
      for (;;) {
        // next distinct value below `last`
        bool found = false;
        float nxt = 0.f;
        for (uint32_t r = lo; r < hi; ++r) {
          if (b.filtered && b.filtered[r]) continue;
          const float sc = b.score[r];
          if (sc < last && (!found || sc > nxt)) { nxt = sc; found = true; }
        }
        if (!found) break;                                   // every value spared
        if (--count <= 0) { cutoff = last; have_cutoff = true; break; }  // records below `last` are masked
        last = nxt;
      }
    }
  }
  auto masked = [&](uint32_t r, float sc) -> bool {
    if (b.filtered && b.filtered[r]) return true;
    if (p.model == 2) {
      if (sc < p.minscore || (evalue ? evalue[r] : 0.) > (double)p.maxevalue) return true;
      if (sc < thr) return true;
      if (p.ignore_unclassified && unclassified && unclassified[b.node[r]]) return true;
    } else if (p.model == 3) {
      if (have_cutoff && sc < cutoff) return true;
    }
    return false;
  };
  if (run_simple) {  // LCASimplePredictionModel::predict, taxonpredictionmodel.hh:73-131
    uint32_t r = lo;
    while (r < hi && masked(r, b.score[r])) ++r;
    if (r < hi) {
      uint32_t qrstart = b.qstart[r], qrstop = b.qstop[r];
      float maxscore = b.score[r];
      if (qrstart > qrstop) { const uint32_t t = qrstart; qrstart = qrstop; qrstop = t; }
      uint32_t lca_all = b.node[r];  // LCA of all unmasked nodes (:95-111)
      for (uint32_t k = r + 1; k < hi; ++k) {
        const float sc = b.score[k];
        if (masked(k, sc)) continue;
        const uint32_t qs = b.qstart[k], qe = b.qstop[k];
        if (qs <= qe) { qrstart = min(qs, qrstart); qrstop = max(qe, qrstop); }
        else { qrstart = min(qe, qrstart); qrstop = max(qs, qrstop); }
        if (sc > maxscore) maxscore = sc;
        lca_all = lca_with(tax, lca_all, b.node[k]);
      }
      // best reference taxon: LCA of the nodes of the records that reach the maximum score (:113-126)
      bool have = false;
      uint32_t lca_best = kRoot;
      for (uint32_t k = r; k < hi; ++k) {
        const float sc = b.score[k];
        if (masked(k, sc) || !(sc == maxscore)) continue;
        const uint32_t nk = b.node[k];
        if (!have) { lca_best = nk; have = true; }
        else lca_best = lca_with(tax, lca_best, nk);
      }
      const uint32_t node = lca_all;
      row.begin = qrstart; row.end = qrstop;
      row.lower = row.upper = node;
      row.support = qrstop - qrstart + 1;  // setNodePoint(node): getQueryFeatureWidth(), predictionrecord.hh:68,178-181
      row.rtax = have ? lca_best : node;
    }
  }
  rows[s] = row;
}

hipError_t launch_lca_models(hipStream_t st, const BatchDev& b, const TaxDev& tax, const LcaParamsDev& p, const double* evalue,
                             const uint8_t* unclassified, void* rows) {
  if (b.n_sets == 0) return hipSuccess;
  const dim3 grid((b.n_sets + kBlock - 1) / kBlock);
  hipLaunchKernelGGL(k_lca_models, grid, dim3(kBlock), 0, st, b, tax, p, evalue, unclassified, (LcaRow*)rows);
  return hipGetLastError();
}

}  // namespace rpa
