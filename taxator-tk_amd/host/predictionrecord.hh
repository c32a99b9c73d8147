// taxator-tk_amd/host/predictionrecord.hh -- prediction result + GFF3 serialisation (boundary, output side).
// Mirrors PredictionRecordBase / PredictionRecord (core/src/predictionrecord.hh:38-435): initialize (:44-54), setNodeRange
// (:152-175), print (:248-261), printFeatureTax (:291-308), GFF3Header (core/src/predictionrecord.cpp:22-25).
// RPA always writes one uniform support value along the node range, which is all this class stores.
#pragma once
#include <cmath>
#include <cstdint>
#include <ostream>
#include <string>

#include "taxonomy.hh"

namespace taxator {

class PredictionRecord {
 public:
  explicit PredictionRecord(const Taxonomy* tax) : tax_(tax) {}
  void initialize(const std::string& query_identifier, uint32_t query_length) {
    query_identifier_ = query_identifier; query_length_ = query_length; feature_begin_ = 1; feature_end_ = query_length;
  }
  void setQueryFeatureBegin(uint32_t v) { feature_begin_ = v; }
  void setQueryFeatureEnd(uint32_t v) { feature_end_ = v; }
  void setInterpolationValue(float f) { interpolation_value_ = f; }
  void setSignalStrength(float f) { signal_strength_ = f; }
  void setBestReferenceTaxon(uint32_t rtax) { rtax_ = rtax; }
  void setNodeRange(uint32_t lower, uint32_t upper, uint32_t support) { lower_ = lower; upper_ = upper; support_ = support; }
  const std::string& getQueryIdentifier() const { return query_identifier_; }
  uint32_t getLowerNode() const { return lower_; }
  uint32_t getUpperNode() const { return upper_; }
  float getInterpolationValue() const { return interpolation_value_; }

  void print(std::ostream& strm) const {
    strm << query_identifier_ << '\t' << "taxator-tk" << '\t' << "sequence_feature" << '\t' << feature_begin_ << '\t' << feature_end_ << '\t';
    if (std::isnan(signal_strength_)) strm << '.'; else strm << signal_strength_;
    strm << '\t' << '.' << '\t' << '.' << '\t';
    strm << "seqlen=" << query_length_ << ';' << "tax=";
    uint32_t last_support = 0;
    for (uint32_t node = lower_; node != upper_; node = tax_->parent[node]) {  // :296-304
      if (support_ != last_support) { strm << tax_->taxid[node] << ':' << support_ << '-'; last_support = support_; }
    }
    strm << tax_->taxid[upper_];
    if (support_ != last_support) strm << ':' << support_;
    strm << ';' << "rtax=" << tax_->taxid[rtax_];
    if (interpolation_value_ >= 0.f && interpolation_value_ < 1.f) strm << ';' << "ival=" << interpolation_value_;  // :255-258
    strm << '\n';
  }

 private:
  const Taxonomy* tax_;
  std::string query_identifier_;
  uint32_t query_length_ = 0, feature_begin_ = 0, feature_end_ = 0;
  uint32_t lower_ = 0, upper_ = 0, rtax_ = 0, support_ = 0;
  float interpolation_value_ = -1.f;  // predictionrecord.hh:40
  float signal_strength_ = 0.f;
};

inline std::ostream& operator<<(std::ostream& strm, const PredictionRecord& prec) { prec.print(strm); return strm; }
struct GFF3Header {};
inline std::ostream& operator<<(std::ostream& strm, const GFF3Header&) { return strm << "##gff-version 3" << std::endl; }

}  // namespace taxator
