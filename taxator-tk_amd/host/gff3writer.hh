// taxator-tk_amd/host/gff3writer.hh -- GFF3 lines straight from the engine's 32-byte prediction rows (boundary, output side).
// Byte-for-byte what PredictionRecord::print (predictionrecord.hh in this directory, mirroring
// core/src/predictionrecord.hh:248-308) writes through an ostream, without the ostream: the columnar pipeline formats
// slices of a batch on all host threads into plain buffers.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>

#include "../../include/taxator_rpa.h"
#include "taxonomy.hh"

namespace taxator {

inline void append_u32(std::string& out, uint32_t v) {
  char tmp[10];
  int n = 0;
  do { tmp[n++] = (char)('0' + v % 10); v /= 10; } while (v);
  while (n) out.push_back(tmp[--n]);
}

// `ival` is the interpolation value the reused PredictionRecord would hold at this point (stale when the row's
// RPA_FLAG_IVAL_UNTOUCHED is set, core/taxator.cpp:66 + taxonpredictionmodelsequence.hh:359-368).
inline void append_gff3(std::string& out, const Taxonomy& tax, const char* qid, size_t qid_len, uint32_t query_length,
                        const rpa_prediction& r, float ival) {
  out.append(qid, qid_len);
  out.append("\ttaxator-tk\tsequence_feature\t");
  append_u32(out, r.begin); out.push_back('\t');
  append_u32(out, r.end);
  out.append("\t0\t.\t.\tseqlen=");  // signal strength is 0 on every RPA path (:722-725,828)
  append_u32(out, query_length);
  out.append(";tax=");
  uint32_t last_support = 0;
  for (uint32_t node = r.lower; node != r.upper; node = tax.parent[node]) {  // printFeatureTax :296-304
    if (r.support != last_support) { out.append(tax.taxid[node]); out.push_back(':'); append_u32(out, r.support); out.push_back('-'); last_support = r.support; }
  }
  out.append(tax.taxid[r.upper]);
  if (r.support != last_support) { out.push_back(':'); append_u32(out, r.support); }
  out.append(";rtax=");
  out.append(tax.taxid[r.rtax]);
  if (ival >= 0.f && ival < 1.f) {  // :255-258; operator<<(float) = %g with 6 significant digits
    char tmp[32];
    const int n = snprintf(tmp, sizeof(tmp), "%g", (double)ival);
    out.append(";ival=");
    out.append(tmp, (size_t)n);
  }
  out.push_back('\n');
}

}  // namespace taxator
