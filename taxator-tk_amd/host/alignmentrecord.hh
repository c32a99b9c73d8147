// taxator-tk_amd/host/alignmentrecord.hh -- alignment TSV records and record-set generation (boundary, input side).
// Mirrors AlignmentRecord::parse (core/src/alignmentrecord.hh:95-158), AlignmentRecordTaxonomy (:207-238),
// FileParser (core/src/fileparser.hh:29-76: '#' lines skipped) and RecordSetGeneratorUnsorted<split> (:457-566):
// consecutive lines of one query id form a group; with split=true the group is sorted by (qstart, qstop, input order)
// and cut into record sets wherever qstart exceeds the running maximum qstop.  (The reference breaks exact ties of
// (qstart, qstop) by heap address, :480; input order is used here.)
#pragma once
#include <cstdint>
#include <istream>
#include <string>
#include <vector>

#include "accessconv.hh"
#include "exception.hh"
#include "sequencestorage.hh"
#include "taxonomy.hh"

namespace taxator {

struct AlignmentRecordTaxonomy {
  std::string query_identifier, reference_identifier;
  uint32_t query_start = 0, query_stop = 0, query_length = 0, reference_start = 0, reference_stop = 0;
  float score = 0.f;
  double evalue = 0.;
  uint32_t identities = 0, alignment_length = 0;
  bool blacklist_this = false;   // isFiltered()
  uint32_t reference_node = 0;   // getReferenceNode()
  uint32_t reference_seq = 0;    // index of the reference in the DB store
  // accessor names of the reference class
  bool isFiltered() const { return blacklist_this; }
  uint32_t getQueryStart() const { return query_start; }
  uint32_t getQueryStop() const { return query_stop; }
  uint32_t getQueryLength() const { return query_length; }
  uint32_t getReferenceStart() const { return reference_start; }
  uint32_t getReferenceStop() const { return reference_stop; }
  float getScore() const { return score; }
  uint32_t getIdentities() const { return identities; }
  uint32_t getAlignmentLength() const { return alignment_length; }
  const std::string& getQueryIdentifier() const { return query_identifier; }
  const std::string& getReferenceIdentifier() const { return reference_identifier; }
  uint32_t getReferenceNode() const { return reference_node; }
};

typedef std::vector<AlignmentRecordTaxonomy> RecordSetType;

class AlignmentRecordFactory {
 public:
  AlignmentRecordFactory(const StrIDConverter& acc2taxid, const Taxonomy* tax, const SequenceStore* db)
      : acc2taxid_(acc2taxid), tax_(tax), db_(db) {}
  void parse(const std::string& line, AlignmentRecordTaxonomy& rec) const;
 private:
  const StrIDConverter& acc2taxid_;
  const Taxonomy* tax_;
  const SequenceStore* db_;
};

class RecordSetGenerator {
 public:
  RecordSetGenerator(std::istream& in, const AlignmentRecordFactory& fac, bool split_alignments, bool alignments_sorted = false);
  bool notEmpty() const { return have_next_ || pending_pos_ < pending_.size(); }
  void getNext(RecordSetType& rset);  // appends one record set
 private:
  bool feed();
  std::istream& in_;
  const AlignmentRecordFactory& fac_;
  bool split_;
  bool sorted_;
  AlignmentRecordTaxonomy next_;
  bool have_next_ = false;
  unsigned line_num_ = 0;
  std::vector<AlignmentRecordTaxonomy> pending_;  // sorted group of the current query
  size_t pending_pos_ = 0;
};

}  // namespace taxator
