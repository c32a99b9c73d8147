// taxator-tk_amd/host/alignmentrecord.cpp -- see alignmentrecord.hh.
#include "alignmentrecord.hh"

#include <algorithm>
#include <cerrno>
#include <cstdlib>

namespace taxator {

static bool to_u32(const std::string& s, uint32_t& out) {  // boost::lexical_cast<large_unsigned_int>: whole field, no sign
  if (s.empty() || s[0] == '-' || s[0] == '+' || isspace((unsigned char)s[0])) return false;
  errno = 0;
  char* end = nullptr;
  unsigned long long v = strtoull(s.c_str(), &end, 10);
  if (errno || *end || v > 0xFFFFFFFFull) return false;
  out = (uint32_t)v;
  return true;
}
static bool to_f32(const std::string& s, float& out) {
  if (s.empty() || isspace((unsigned char)s[0])) return false;
  errno = 0;
  char* end = nullptr;
  out = strtof(s.c_str(), &end);
  return !*end;
}
static bool to_f64(const std::string& s, double& out) {
  if (s.empty() || isspace((unsigned char)s[0])) return false;
  char* end = nullptr;
  out = strtod(s.c_str(), &end);
  return !*end;
}

void AlignmentRecordFactory::parse(const std::string& line_in, AlignmentRecordTaxonomy& rec) const {
  if (line_in.size() <= 1) throw ParsingError("alignment line too short");  // :96
  rec.blacklist_this = line_in[0] == '*';                                       // :98-104
  const std::string line = rec.blacklist_this ? line_in.substr(1) : line_in;
  std::vector<std::string> f;
  size_t pos = 0;
  while (f.size() < 11) {  // tokenizeSingleCharDelim(..., 12): 11 split fields + the rest
    size_t e = line.find('\t', pos);
    if (e == std::string::npos) break;
    f.emplace_back(line, pos, e - pos);
    pos = e + 1;
  }
  f.emplace_back(line, pos);
  if (f.size() < 11) throw ParsingError("bad number of fields in alignment line");  // :157
  if (!to_u32(f[1], rec.query_start) || !to_u32(f[2], rec.query_stop)) throw ParsingError("bad position number or query length");
  if (rec.query_start > rec.query_stop)
    throw ParsingError("reverse query positions not allowed (only reference positions can be swapped to indicate the reverse complement, adjust input");
  if (!to_u32(f[3], rec.query_length) || !to_u32(f[5], rec.reference_start) || !to_u32(f[6], rec.reference_stop))
    throw ParsingError("bad position number or query length");
  if (!to_f32(f[7], rec.score)) throw ParsingError("bad score");
  if (!to_f64(f[8], rec.evalue)) throw ParsingError("bad E-value");
  if (!to_u32(f[9], rec.identities)) throw ParsingError("bad identity value");
  if (!to_u32(f[10], rec.alignment_length)) throw ParsingError("bad alignment length");
  rec.query_identifier = f[0];
  rec.reference_identifier = f[4];
  // AlignmentRecordTaxonomy::parse (:211-228)
  TaxonID taxid;
  try { taxid = acc2taxid_[rec.reference_identifier]; }
  catch (Exception& e) { throw TaxonMappingNotFound(std::string(e.what()) + " (bad taxon mapping for alignment reference sequence)"); }
  try { rec.reference_node = tax_->getNode(taxid); }
  catch (Exception& e) { throw TaxonNotFound(std::string(e.what()) + " (bad alignment reference taxon)"); }
  rec.reference_seq = db_ ? db_->getIndex(rec.reference_identifier) : 0u;  // no sequence store with the alignment-free models
}

RecordSetGenerator::RecordSetGenerator(std::istream& in, const AlignmentRecordFactory& fac, bool split_alignments, bool alignments_sorted)
    : in_(in), fac_(fac), split_(split_alignments), sorted_(alignments_sorted) {
  have_next_ = feed();
}

bool RecordSetGenerator::feed() {
  std::string line;
  while (std::getline(in_, line)) {
    ++line_num_;
    if (!line.empty() && line[0] == '#') continue;  // fileparser.hh:61-67 (empty lines are NOT skipped: they fail parse)
    try { fac_.parse(line, next_); }
    catch (Exception& e) { throw ParsingError(std::string(e.what()) + " [line " + std::to_string(line_num_) + "]"); }
    return true;
  }
  return false;
}

void RecordSetGenerator::getNext(RecordSetType& rset) {
  if (pending_pos_ >= pending_.size()) {  // read the next query group (:461-481)
    if (!have_next_) throw EOFError("alignment recordset parser: read from empty input");
    pending_.clear();
    pending_pos_ = 0;
    const std::string qid = next_.query_identifier;
    do {
      pending_.push_back(next_);
      have_next_ = feed();
    } while (have_next_ && next_.query_identifier == qid);
    if (split_ && !sorted_)  // -o 1 (RecordSetGeneratorSorted, :569-631): same cut rule, input order
      std::stable_sort(pending_.begin(), pending_.end(), [](const AlignmentRecordTaxonomy& a, const AlignmentRecordTaxonomy& b) {
        if (a.query_start != b.query_start) return a.query_start < b.query_start;
        return a.query_stop < b.query_stop;
      });
  }
  if (!split_) {  // :543-566
    for (; pending_pos_ < pending_.size(); ++pending_pos_) rset.push_back(pending_[pending_pos_]);
    return;
  }
  uint32_t rstop = pending_[pending_pos_].query_stop;  // :485-504
  rset.push_back(pending_[pending_pos_++]);
  while (pending_pos_ < pending_.size()) {
    const AlignmentRecordTaxonomy& r = pending_[pending_pos_];
    if (r.query_start > rstop) return;  // split point
    rstop = std::max(rstop, r.query_stop);
    rset.push_back(r);
    ++pending_pos_;
  }
}

}  // namespace taxator
