// taxator-tk_amd/csrc/lca_kernels.hip -- the reference's alignment-free prediction models on the device.
//
// What it replaces (all in /root/reference/core/src): DummyPredictionModel (taxonpredictionmodel.hh:57-66),
// LCASimplePredictionModel (:70-132), MeganLCAPredictionModel (:136-165) with MinScoreMaxEvalueTopPercentFilter
// (alignmentsfilter.hh:349-386) and RemoveUnclassifiedFilter (:612-626), NBestLCAPredictionModel (:237-253) with
// NumBestBitscoreFilter (alignmentsfilter.hh:493-532), as selected by taxator -a (core/taxator.cpp:346-361).
// One thread per record set; the set's records are read straight from the batch columns.  The filters of the
// reference mark records ("filterOut"); here a record's mask is a pure function of the set-level quantities the
// filters compute (running maximum, n-th best distinct score), so no per-record scratch is written.  The LCA of a set
// of taxa is the common prefix of their root paths (per-node ancestor table, see place_kernels.hip).
// None of these models sets the interpolation value: rows carry ival = -1 and RPA_FLAG_IVAL_UNTOUCHED.
#include "rpa_device.h"

namespace rpa {

struct LcaRow {  // = rpa_prediction
  uint32_t lower, upper, rtax, support, begin, end;
  float ival;
  uint32_t flags;
};

template <int P4>
__device__ __forceinline__ void node_path(const TaxDev& tax, uint32_t node, uint32_t (&path)[4 * P4]) {
  const uint4* a = reinterpret_cast<const uint4*>(tax.anc) + (size_t)node * P4;
#pragma unroll
  for (int k = 0; k < P4; ++k) { const uint4 v = a[k]; path[4 * k] = v.x; path[4 * k + 1] = v.y; path[4 * k + 2] = v.z; path[4 * k + 3] = v.w; }
}
template <int P4>
__device__ __forceinline__ uint32_t common_prefix(const uint32_t (&a)[4 * P4], uint32_t da, const uint32_t (&b)[4 * P4], uint32_t db) {
  uint32_t c = 0;
  bool run = true;
#pragma unroll
  for (int d = 0; d < 4 * P4; ++d) { run = run && (a[d] == b[d]); c += run ? 1u : 0u; }
  return min(c, min(da, db));
}
template <int P4>
__device__ __forceinline__ uint32_t path_node(const uint32_t (&path)[4 * P4], uint32_t depth) {
  uint32_t v = kRoot;
#pragma unroll
  for (int d = 0; d < 4 * P4; ++d) if ((uint32_t)d + 1 == depth) v = path[d];
  return v;
}

template <int P4>
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
      uint32_t ref_path[4 * P4], pth[4 * P4];
      const uint32_t n0 = b.node[r];
      node_path<P4>(tax, n0, ref_path);
      const uint32_t d0 = tax.depth[n0];
      uint32_t d_all = d0;  // LCA of all unmasked nodes as a point on the first node's path
      for (uint32_t k = r + 1; k < hi; ++k) {
        const float sc = b.score[k];
        if (masked(k, sc)) continue;
        const uint32_t qs = b.qstart[k], qe = b.qstop[k];
        if (qs <= qe) { qrstart = min(qs, qrstart); qrstop = max(qe, qrstop); }
        else { qrstart = min(qe, qrstart); qrstop = max(qs, qrstop); }
        if (sc > maxscore) maxscore = sc;
        const uint32_t nk = b.node[k];
        node_path<P4>(tax, nk, pth);
        d_all = min(d_all, common_prefix<P4>(ref_path, d0, pth, tax.depth[nk]));
      }
      // best reference taxon: LCA of the nodes of the records that reach the maximum score (:113-126)
      bool have = false;
      uint32_t best_path[4 * P4];
      uint32_t d_best = 0, db0 = 0;
      for (uint32_t k = r; k < hi; ++k) {
        const float sc = b.score[k];
        if (masked(k, sc) || !(sc == maxscore)) continue;
        const uint32_t nk = b.node[k];
        if (!have) { node_path<P4>(tax, nk, best_path); db0 = tax.depth[nk]; d_best = db0; have = true; }
        else { node_path<P4>(tax, nk, pth); d_best = min(d_best, common_prefix<P4>(best_path, db0, pth, tax.depth[nk])); }
      }
      const uint32_t node = path_node<P4>(ref_path, d_all);
      row.begin = qrstart; row.end = qrstop;
      row.lower = row.upper = node;
      row.support = qrstop - qrstart + 1;  // setNodePoint(node): getQueryFeatureWidth(), predictionrecord.hh:68,178-181
      row.rtax = have ? path_node<P4>(best_path, d_best) : node;
    }
  }
  rows[s] = row;
}

hipError_t launch_lca_models(hipStream_t st, const BatchDev& b, const TaxDev& tax, const LcaParamsDev& p, const double* evalue,
                             const uint8_t* unclassified, void* rows) {
  if (b.n_sets == 0) return hipSuccess;
  const dim3 grid((b.n_sets + kBlock - 1) / kBlock);
  switch (tax.path_quads) {
    case 1: hipLaunchKernelGGL(k_lca_models<1>, grid, dim3(kBlock), 0, st, b, tax, p, evalue, unclassified, (LcaRow*)rows); break;
    case 2: hipLaunchKernelGGL(k_lca_models<2>, grid, dim3(kBlock), 0, st, b, tax, p, evalue, unclassified, (LcaRow*)rows); break;
    case 3: hipLaunchKernelGGL(k_lca_models<3>, grid, dim3(kBlock), 0, st, b, tax, p, evalue, unclassified, (LcaRow*)rows); break;
    case 4: hipLaunchKernelGGL(k_lca_models<4>, grid, dim3(kBlock), 0, st, b, tax, p, evalue, unclassified, (LcaRow*)rows); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

}  // namespace rpa
