// taxator-tk_amd/host/binner.cpp -- `binner`: consensus taxonomic assignment of whole sequences from the per-segment
// predictions `taxator` prints (the second CLI BASELINE.json's north_star names).  Host-only, O(#segments).
//
// Mirrors core/binner.cpp:56-346 with core/src/predictionranges.hh:30-266 (combinePredictionRanges and its helpers),
// core/src/predictionrecord.hh:68-86,152-194,196-250,330-415 (GFF3 parsing, support vector, node range),
// core/src/predictionrecordbinning.hh:26-76, core/src/fastnodemap.hh:32-56 and the bioboxes.org binning writer
// (core/src/bioboxes.cpp:24-108): same options (-n -g -s -j -i -f -l, hidden -r -m -d), same three steps
//   0  read GFF3 predictions, group them by the regex-extracted sequence identifier,
//   1  count the support of every taxon over the sample, prune range ends below --sample-min-support,
//   2  per group: pass a single prediction through, combine several (walk down from the root keeping the majority
//      branch; "direct" mode when a node's own support reaches the threshold, else the deepest node whose total support
//      does), apply the per-rank identity constraints, write one bioboxes line,
// same diagnostics on stderr.  Reference behaviour kept on purpose: supports inside the combination are 16-bit
// (medium_unsigned_int, core/src/types.hh:32: larger values wrap), groups come out in std::unordered_map order (same
// container, same libstdc++ => same order), a group identifier that re-appears later in the input joins its first group.
// Plain C++17: std::regex (ECMAScript) stands in for boost::regex (Perl syntax) -- identical for the patterns the
// pipelines use ("(.+)", "([^_]+)_.*", extended/binning-*.bash).
#include <algorithm>
#include <cassert>
#include <cstdint>
#include <cstdlib>
#include <fstream>
#include <iomanip>
#include <iostream>
#include <limits>
#include <list>
#include <map>
#include <memory>
#include <regex>
#include <set>
#include <sstream>
#include <string>
#include <unordered_map>
#include <vector>

#include "exception.hh"
#include "taxonomy.hh"

using namespace taxator;

static const char* program_version = "1.5.0-mi355x";  // core/src/constants.hh:47
typedef uint16_t medium_unsigned_int;                  // core/src/types.hh:32
typedef uint32_t large_unsigned_int;

namespace {

// tokenizeSingleCharDelim (core/src/utils.hh:59-80) as observed from outside: at most `fieldnum` cuts (0 = no limit),
// empty pieces dropped when trimempty, and the rest of the string ALWAYS appended as a last token (so "a-b" gives
// "a", "b", "" -- the tax= parser below relies on that empty last field, predictionrecord.hh:362).
std::vector<std::string> split_fields(const std::string& str, char delim, int fieldnum, bool trimempty) {
  std::vector<std::string> tokens;
  const size_t len = str.size();
  if (!fieldnum) fieldnum = (int)len;
  size_t lastpos = 0;
  while (fieldnum && lastpos < len) {
    const size_t pos = str.find(delim, lastpos);
    if (pos == std::string::npos) {
      if (len != lastpos || !trimempty) tokens.emplace_back(str, lastpos, len - lastpos);
      lastpos = len;
      break;
    }
    if (pos != lastpos || !trimempty) { tokens.emplace_back(str, lastpos, pos - lastpos); --fieldnum; }
    lastpos = pos + 1;
  }
  tokens.emplace_back(str, lastpos, len - lastpos);
  return tokens;
}

template <typename T>
T lexical(const std::string& s, const char* what) {  // boost::lexical_cast: the whole string must convert
  std::istringstream in(s);
  T v;
  in >> std::noskipws >> v;
  if (in.fail() || in.peek() != std::char_traits<char>::eof() || s.empty()) throw ParsingError(what);
  return v;
}
template <>
uint32_t lexical<uint32_t>(const std::string& s, const char* what) {
  if (s.empty() || s[0] == '-' || s[0] == '+' || s[0] == ' ') throw ParsingError(what);
  size_t used = 0;
  unsigned long v = 0;
  try { v = std::stoul(s, &used, 10); } catch (...) { throw ParsingError(what); }
  if (used != s.size() || v > 0xFFFFFFFFul) throw ParsingError(what);
  return (uint32_t)v;
}

// PredictionRecordBinning (core/src/predictionrecordbinning.hh:26-76 over PredictionRecordBase)
struct Prediction {
  const Taxonomy* tax;
  std::string qid;
  large_unsigned_int qlen = 0, begin = 0, end = 0;
  uint32_t lower = 0, upper = 0, rtax = 0;
  float ival = -1.f, signal = 0.f;
  std::vector<large_unsigned_int> taxon_support;  // index 0 = upper node ... back() = lower node

  explicit Prediction(const Taxonomy* t) : tax(t) {}
  large_unsigned_int width() const { return end - begin + 1; }
  large_unsigned_int getSupportAt(uint8_t depth) const {  // predictionrecord.hh:76-86
    const int index = (int)depth - (int)tax->depth[upper];
    if (index >= 0) return index < (int)taxon_support.size() ? taxon_support.at((size_t)index) : taxon_support.back();
    return 0;
  }
  large_unsigned_int getSupportAtNode(uint32_t node) const { return getSupportAt(tax->depth[node]); }
  void setNodeRange(uint32_t lo, large_unsigned_int lo_support, uint32_t up, large_unsigned_int up_support) {  // :152-158
    if (!(lo == up || tax->isParentOf(up, lo))) throw GeneralError("setNodeRange: upper node is not an ancestor of the lower node");
    lower = lo; upper = up;
    taxon_support.assign((size_t)(tax->depth[lo] - tax->depth[up] + 1), up_support);
    taxon_support.back() = lo_support;
  }
  void setNodePoint(uint32_t node, large_unsigned_int support) { setNodeRange(node, support, node, support); }
  void pruneLowerNode(uint32_t node) {  // :183-187
    if (!(tax->isParentOf(node, lower) && tax->depth[node] >= tax->depth[upper])) throw GeneralError("pruneLowerNode: bad node");
    taxon_support.resize((size_t)(tax->depth[node] - tax->depth[upper] + 1));
    lower = node;
  }
  void setSupportAt(uint32_t node, large_unsigned_int support) { taxon_support.at((size_t)(tax->depth[node] - tax->depth[upper])) = support; }

  void parse(const std::string& line) {  // predictionrecord.hh:196-250
    if (line.empty()) throw ParsingError("empty GFF3 line");
    const std::vector<std::string> fields = split_fields(line, '\t', 9, false);
    if (fields.size() < 9) throw ParsingError("too few GFF3 fields in line");
    if (fields[1].size() < 10 || fields[1].compare(0, 10, "taxator-tk") != 0) std::cerr << "warning: gff3 produced by unknown algorithm" << std::endl;
    begin = lexical<uint32_t>(fields[3], "bad GFF3 feature position");
    end = lexical<uint32_t>(fields[4], "bad GFF3 feature position");
    if (begin > end) throw ParsingError("GFF3 reverse query positions");
    signal = fields[5] == "." ? std::numeric_limits<float>::quiet_NaN() : lexical<float>(fields[5], "bad GFF3 taxonomic signal score");
    for (const std::string& kv : split_fields(fields[8], ';', 0, true)) {
      if (kv.empty()) continue;  // the always-appended rest (no key: the reference's parseKeyValue matches nothing)
      const std::vector<std::string> key_value = split_fields(kv, '=', 2, false);
      parseKeyValue(key_value[0], key_value.size() > 1 ? key_value[1] : std::string());
    }
    if (ival == -1.f) ival = 1.f;  // default value for output compression (:244)
    qid = fields[0];
  }

  void parseKeyValue(const std::string& key, const std::string& value) {  // :330-415
    if (key == "seqlen") { qlen = lexical<uint32_t>(value, "bad GFF3 key value seqlen"); return; }
    if (key == "ival") { ival = lexical<float>(value, "bad GFF3 key value ival"); return; }
    if (key == "rtax") { rtax = tax->getNode(value); return; }
    if (key != "tax") return;
    taxon_support.clear();
    const std::vector<std::string> taxpath = split_fields(value, '-', 0, false);
    size_t it = 0;
    std::vector<std::string> ts = split_fields(taxpath[it], ':', 2, false);
    large_unsigned_int support = ts.size() < 2 || ts[1].empty() ? width() : lexical<uint32_t>(ts[1], "bad GFF3 key value tax");
    uint32_t last_node = tax->getNode(ts[0]);
    lower = last_node;
    std::list<large_unsigned_int> tmp;
    while (++it < taxpath.size() && !taxpath[it].empty()) {  // the last field is empty by construction
      ts = split_fields(taxpath[it], ':', 2, false);
      const uint32_t node = tax->getNode(ts[0]);
      if (!tax->isParentOf(node, last_node)) throw ParsingError("bad taxon path " + tax->taxid[node] + " " + tax->taxid[last_node]);
      for (uint32_t p = last_node; p != node; p = tax->parent[p]) tmp.push_front(support);
      if (ts.size() > 1 && !ts[1].empty()) support = lexical<uint32_t>(ts[1], "bad GFF3 key value tax");
      last_node = node;
    }
    tmp.push_front(support);
    upper = last_node;
    taxon_support.assign(tmp.begin(), tmp.end());
    if ((size_t)(tax->depth[lower] - tax->depth[upper] + 1) != taxon_support.size()) throw GeneralError("tax= path and support vector disagree");
  }
};

typedef std::list<Prediction> RecordGroup;

// extractRegex (core/binner.cpp:45-54): the first capture group of a FULL match; the asserts there are live.
std::string extract_regex(const std::string& text, const std::regex& re, bool empty_regex) {
  if (empty_regex) return "consensus_sequence";
  std::smatch m;
  if (!std::regex_match(text, m, re) || m.size() < 2 || m[1].first == m[1].second) {
    std::cerr << "binner: glob identifier regex does not match \"" << text << "\" with a non-empty first group" << std::endl;
    std::abort();
  }
  return m[1].str();
}

// ---- combinePredictionRanges and its helpers, core/src/predictionranges.hh:30-266 ----
struct RangeCombine {
  std::vector<uint32_t> path;  // root ... lower node (CPathDownIterator)
  size_t pos = 0;              // current node = path[pos]
  std::vector<medium_unsigned_int> direct, total;
  bool ended = false;
};
typedef std::list<RangeCombine> CombineList;

void getSupport(const Taxonomy& tax, const CombineList& tl, medium_unsigned_int& direct, medium_unsigned_int& total) {  // :59-68
  medium_unsigned_int d = 0, t = 0;
  for (const RangeCombine& rc : tl) { const uint8_t k = tax.depth[rc.path[rc.pos]]; d = (medium_unsigned_int)(d + rc.direct[k]); t = (medium_unsigned_int)(t + rc.total[k]); }
  direct = d; total = t;
}
void setPathEndState(const Taxonomy& tax, CombineList& tl) {  // :70-72
  for (RangeCombine& rc : tl) rc.ended = tax.depth[rc.path[rc.pos]] == rc.direct.size() - 1;
}
bool reduceToMajority(const Taxonomy& tax, CombineList& tl) {  // :78-112
  if (tl.size() < 2) return false;
  std::map<uint32_t, float> supports;
  uint32_t max_node = 0;
  bool have_max = false;
  float max_support = .0f;
  for (const RangeCombine& rc : tl) {
    const uint32_t node = rc.path[rc.pos];
    const float add = rc.total[tax.depth[node]];
    auto f = supports.find(node);
    if (f != supports.end()) f->second += add; else f = supports.insert(std::make_pair(node, add)).first;
    if (f->second > max_support) { max_support = f->second; max_node = node; have_max = true; }
  }
  if (supports.size() == 1) return false;
  for (auto it = tl.begin(); it != tl.end();) {
    if (!have_max || it->path[it->pos] != max_node) it = tl.erase(it); else ++it;
  }
  return true;
}

struct PathEntry { uint32_t node; medium_unsigned_int direct, total; bool branch; };

std::unique_ptr<Prediction> combinePredictionRanges(const RecordGroup& predictions, const std::string& identifier, const Taxonomy* tax,
                                                    float min_signal_percentage, medium_unsigned_int min_support, std::ostream& debug_output) {
  CombineList tlist;
  medium_unsigned_int summed_support = 0;
  large_unsigned_int summed_length = 0;
  {
    std::set<std::string> processed_identifiers;
    for (const Prediction& p : predictions) {  // :134-158
      int i = tax->depth[p.lower];
      const medium_unsigned_int support = (medium_unsigned_int)p.getSupportAt((uint8_t)i);
      summed_support = (medium_unsigned_int)(summed_support + support);
      if (processed_identifiers.emplace(p.qid).second) summed_length += p.qlen;
      const size_t depth = (size_t)i + 1;
      tlist.emplace_back();
      RangeCombine& rc = tlist.back();
      rc.path.resize(depth);
      { uint32_t n = p.lower; for (size_t k = depth; k-- > 0;) { rc.path[k] = n; n = tax->parent[n]; } }
      rc.direct.assign(depth, 0); rc.total.assign(depth, 0);
      rc.total[(size_t)i] = rc.direct[(size_t)i] = support;
      while (--i >= 0) {
        rc.direct[(size_t)i] = (medium_unsigned_int)p.getSupportAt((uint8_t)i);
        rc.total[(size_t)i] = std::max(rc.total[(size_t)i + 1], rc.direct[(size_t)i]);
      }
    }
  }
  std::unique_ptr<Prediction> prec(new Prediction(tax));  // :160-166
  prec->qid = identifier; prec->qlen = summed_length; prec->begin = 1; prec->end = summed_length;

  const medium_unsigned_int direct_support_thresh = std::max(static_cast<medium_unsigned_int>(min_signal_percentage * summed_support), min_support);
  medium_unsigned_int direct_support, total_support;
  std::vector<PathEntry> path;
  const uint32_t root_node = tax->getRoot();

  debug_output << std::endl << "combining " << predictions.size() << " independent predictions for query " << identifier << ", threshold "
               << direct_support_thresh << " (" << static_cast<unsigned>(min_signal_percentage * 100) << " %)" << std::endl;
  for (const Prediction& p : predictions) {  // :176-183
    debug_output << static_cast<int>(p.getSupportAtNode(p.lower)) << ": ";
    std::vector<uint32_t> chain;
    for (uint32_t n = p.lower; ; n = tax->parent[n]) { chain.push_back(n); if (n == root_node) break; }
    std::reverse(chain.begin(), chain.end());
    size_t k = 0;
    for (; chain[k] != p.upper; ++k) debug_output << tax->name[chain[k]] << ";";
    debug_output << "[";
    for (; chain[k] != p.lower; ++k) debug_output << tax->name[chain[k]] << ";";
    debug_output << tax->name[p.lower] << "]" << std::endl;
  }

  setPathEndState(*tax, tlist);  // :186-187
  getSupport(*tax, tlist, direct_support, total_support);
  int lower_direct_node_index = -1;
  int running_index = 0;
  while (!tlist.empty()) {  // :192-203
    const uint32_t node = tlist.front().path[tlist.front().pos];
    if (direct_support >= direct_support_thresh) lower_direct_node_index = running_index;
    path.push_back(PathEntry{node, direct_support, total_support, false});
    for (auto it = tlist.begin(); it != tlist.end();) { if (it->ended) it = tlist.erase(it); else ++it; }  // removeIf<3>
    for (RangeCombine& rc : tlist) ++rc.pos;                                                                // stepDown
    ++running_index;
    setPathEndState(*tax, tlist);
    path.back().branch = reduceToMajority(*tax, tlist);
    getSupport(*tax, tlist, direct_support, total_support);
  }

  debug_output << std::endl;
  debug_output << std::setprecision(3);
  debug_output << "  L |  direct s. |    total s.| B | name" << std::endl;
  debug_output << "--------------------------------------------" << std::endl;
  for (const PathEntry& pe : path) {
    debug_output << std::fixed << std::setw(3) << static_cast<int>(tax->depth[pe.node]) << " | " << std::setw(10) << static_cast<int>(pe.direct) << " | "
                 << std::setw(10) << static_cast<int>(pe.total) << " | " << pe.branch << " | ";
    if (pe.direct >= direct_support_thresh) debug_output << "*";
    debug_output << tax->name[pe.node] << std::endl;
  }

  if (lower_direct_node_index >= 0) {  // direct mode, :217-252
    debug_output << "using direct binning mode..." << std::endl;
    const uint32_t lower_node = path[(size_t)lower_direct_node_index].node;
    const medium_unsigned_int lower_node_support = path[(size_t)lower_direct_node_index].total;
    medium_unsigned_int upper_node_support = lower_node_support;
    uint32_t upper_node = lower_node;
    int upper_direct_node_index = lower_direct_node_index;
    for (int j = lower_direct_node_index; j >= 0; --j) {
      if (path[(size_t)j].direct >= direct_support_thresh) {
        upper_node_support = path[(size_t)j].total;
        upper_node = path[(size_t)j].node;
        upper_direct_node_index = j;
        if (path[(size_t)j].branch) break;
      }
    }
    prec->setNodeRange(lower_node, lower_node_support, upper_node, upper_node_support);
    for (int i = lower_direct_node_index; i > upper_direct_node_index; --i) prec->setSupportAt(path[(size_t)i].node, path[(size_t)i].direct);
    return prec;
  }
  debug_output << "using fallback binning mode..." << std::endl;  // :255-266: majority LCA above the threshold
  for (int i = (int)path.size() - 1; i >= 0; --i) {
    if (path[(size_t)i].total >= direct_support_thresh) { prec->setNodePoint(path[(size_t)i].node, path[(size_t)i].total); return prec; }
  }
  prec->setNodePoint(path[0].node, path[0].total);
  return prec;
}

struct Options {
  std::string sample_identifier, glob_identifier_regex = "(.+)", min_support_in_sample_str = "0", log_filename = "binning.log";
  bool have_sample = false, delete_unmarked = true;
  large_unsigned_int min_support_per_sequence = 50;
  float signal_majority_per_sequence = .7f;
  std::vector<std::string> ranks, files, identity_constrain;
};

void usage(std::ostream& o) {
  o << "Allowed options:\n"
       "  -h [ --help ]                          show help message\n"
       "  -V [ --version ]                       show program version\n"
       "  -n [ --sample-identifier ] arg         unique sample identifier\n"
       "  -g [ --glob-identifier ] arg (=(.+))   grouping regex for substring matching to glob sequence identifiers for consensus binning\n"
       "  -s [ --sequence-min-support ] arg (=50) minimum number of positions supporting a taxonomic signal for any single sequence\n"
       "  -j [ --signal-majority ] arg (=0.7)    minimum combined fraction of support for any single sequence (> 0.5 to be stable)\n"
       "  -i [ --identity-constrain ] arg        minimum required identity for this rank (e.g. -i species:0.8 -i genus:0.7)\n"
       "  -f [ --files ] arg                     arbitrary number of prediction files (\"-\" = standard input)\n"
       "  -l [ --logfile ] arg (=binning.log)    specify name of file for logging\n"
       "  advanced: -r ranks..., -m sample-min-support (=0), -d delete-notranks (=1)\n";
}

bool parse_bool(const std::string& v) { return !(v == "0" || v == "false" || v == "no" || v == "off"); }

}  // namespace

int main(int argc, char** argv) {
  Options o;
  for (int i = 1; i < argc; ++i) {
    const std::string a = argv[i];
    auto val = [&](const char* name) -> std::string {
      if (i + 1 >= argc) { std::cerr << "the required argument for option '" << name << "' is missing" << std::endl; exit(EXIT_FAILURE); }
      return argv[++i];
    };
    if (a == "-h" || a == "--help") { usage(std::cout); return EXIT_SUCCESS; }
    else if (a == "-V" || a == "--version") { std::cout << program_version << std::endl; return EXIT_SUCCESS; }
    else if (a == "-n" || a == "--sample-identifier") { o.sample_identifier = val("--sample-identifier"); o.have_sample = true; }
    else if (a == "-g" || a == "--glob-identifier") o.glob_identifier_regex = val("--glob-identifier");
    else if (a == "-s" || a == "--sequence-min-support") o.min_support_per_sequence = (large_unsigned_int)strtoul(val("--sequence-min-support").c_str(), nullptr, 10);
    else if (a == "-j" || a == "--signal-majority") o.signal_majority_per_sequence = (float)atof(val("--signal-majority").c_str());
    else if (a == "-i" || a == "--identity-constrain") o.identity_constrain.push_back(val("--identity-constrain"));
    else if (a == "-l" || a == "--logfile") o.log_filename = val("--logfile");
    else if (a == "-m" || a == "--sample-min-support") o.min_support_in_sample_str = val("--sample-min-support");
    else if (a == "-d" || a == "--delete-notranks") o.delete_unmarked = parse_bool(val("--delete-notranks"));
    else if (a == "-f" || a == "--files") { while (i + 1 < argc && (argv[i + 1][0] != '-' || argv[i + 1][1] == '\0')) o.files.push_back(argv[++i]); }
    else if (a == "-r" || a == "--ranks") { while (i + 1 < argc && argv[i + 1][0] != '-') o.ranks.push_back(argv[++i]); }
    else { std::cerr << "unrecognised option '" << a << "'" << std::endl; return EXIT_FAILURE; }
  }
  if (!o.have_sample) { std::cerr << "the option '--sample-identifier' is required but missing" << std::endl; return EXIT_FAILURE; }
  if (o.ranks.empty()) o.ranks = default_ranks;  // binner.cpp:118

  const bool empty_regex = o.glob_identifier_regex.empty();
  const std::regex globbing_regex(o.glob_identifier_regex);

  large_unsigned_int min_support_in_sample = 0;  // :124-126
  float min_support_in_sample_percentage = 0.f;
  try {
    if (o.min_support_in_sample_str.find('.') == std::string::npos) min_support_in_sample = lexical<uint32_t>(o.min_support_in_sample_str, "sample-min-support");
    else min_support_in_sample_percentage = lexical<float>(o.min_support_in_sample_str, "sample-min-support");
  } catch (Exception&) { std::cerr << "bad value for --sample-min-support" << std::endl; return EXIT_FAILURE; }

  std::unique_ptr<Taxonomy> tax(loadTaxonomyFromEnvironment(&o.ranks, o.delete_unmarked));  // :131-133
  if (!tax) return EXIT_FAILURE;

  std::map<std::string, float> pid_per_rank;  // :136-156 (keyed by the rank name; the reference keys by the interned string)
  for (const std::string& v : o.identity_constrain) {
    const std::vector<std::string> fields = split_fields(v, ':', 1, false);
    if (fields[0].empty()) {
      std::cerr << "Could not read identity constrain: rank cannot be empty string, use e.g. \"-i species:0.8\"" << std::endl;
      return EXIT_FAILURE;
    }
    try { pid_per_rank[fields[0]] = lexical<float>(fields.size() > 1 ? fields[1] : std::string(), "identity"); }
    catch (Exception&) {
      std::cerr << "Could not read identity constrain: \"" << (fields.size() > 1 ? fields[1] : std::string()) << "\" for rank \"" << fields[0]
                << "\" as float, use e.g. \"-i species:0.8\"" << std::endl;
      return EXIT_FAILURE;
    }
  }

  try {
    // STEP 0: parse the input, group by glob identifier (:161-205)
    std::unordered_map<std::string, RecordGroup> grouped_records;
    {
      std::vector<std::string> sources = o.files;
      if (sources.empty()) sources.push_back("-");
      std::string old_name;
      bool have_old = false;
      RecordGroup* records = nullptr;
      for (const std::string& src : sources) {
        std::ifstream file;
        std::istream* in = &std::cin;
        if (src != "-") {
          file.open(src.c_str());
          if (!file) { std::cerr << "Could not read file \"" << src << "\"" << std::endl; return EXIT_FAILURE; }
          in = &file;
        }
        std::string line;
        while (std::getline(*in, line)) {
          if (line.empty() || line[0] == '#') continue;  // emptyLine / ignoreLine, core/src/utils.hh:41-48
          Prediction rec(tax.get());
          rec.parse(line);
          const std::string new_name = extract_regex(rec.qid, globbing_regex, empty_regex);
          if (!have_old || new_name != old_name) {
            records = &grouped_records.emplace(new_name, RecordGroup()).first->second;
            old_name = new_name; have_old = true;
          }
          records->push_back(std::move(rec));
        }
      }
    }

    // STEP 1: sample support of every taxon, range pruning (:207-283)
    std::cerr << "Analyzing sample composition: ";
    large_unsigned_int minimum_support_found = std::numeric_limits<large_unsigned_int>::max();
    const uint32_t root_node = tax->getRoot();
    std::map<uint32_t, large_unsigned_int> support;  // FastNodeMap
    support[root_node];
    for (auto& g : grouped_records) {
      for (const Prediction& p : g.second) {
        uint32_t pit = p.lower;
        large_unsigned_int total_node_support = p.getSupportAtNode(pit);
        minimum_support_found = std::min(minimum_support_found, total_node_support);
        support[pit] += total_node_support;
        if (pit != root_node) {
          while ((pit = tax->parent[pit]) != root_node) {
            total_node_support = std::max(total_node_support, p.getSupportAtNode(pit));
            support[pit] += total_node_support;
          }
          total_node_support = std::max(total_node_support, p.getSupportAtNode(root_node));
          support[root_node] += total_node_support;
        }
      }
    }
    std::cerr << support.size() << " nested taxa with total support of " << support[root_node] << " positions" << std::endl;
    if (min_support_in_sample_percentage) min_support_in_sample = (large_unsigned_int)(support[root_node] * min_support_in_sample_percentage);

    std::cerr << "Noise removal: ";
    std::set<uint32_t> pruned_nodes;
    if (minimum_support_found < min_support_in_sample) {
      for (auto& g : grouped_records) {
        RecordGroup& records = g.second;
        for (auto prec_it = records.begin(); prec_it != records.end();) {
          const uint32_t lower_node = prec_it->lower, upper_node = prec_it->upper;
          uint32_t pit = lower_node;
          while (pit != upper_node && support[pit] < min_support_in_sample) { pruned_nodes.insert(pit); pit = tax->parent[pit]; }
          if (pit == upper_node && support[pit] < min_support_in_sample) {  // remove the whole range
            pruned_nodes.insert(pit);
            prec_it = records.erase(prec_it);
            continue;
          }
          if (pit != lower_node) prec_it->pruneLowerNode(pit);
          ++prec_it;
        }
      }
    }
    std::cerr << pruned_nodes.size() << " taxa removed" << std::endl;

    // STEP 2: consensus per group, bioboxes output (:285-340)
    std::cerr << "Consensus taxonomy assignment: ";
    std::ofstream binning_debug_output(o.log_filename.c_str());
    std::ostream& out = std::cout;
    out << "# This is the bioboxes.org binning output format at\n# https://github.com/bioboxes/rfc/tree/master/data-format\n\n";  // bioboxes.cpp:36-38
    out << "@Version:0.9.1\n" << "@SampleID:" << o.sample_identifier << "\n";
    if (!tax->version.empty()) out << "@TaxonomyID:" << tax->version << "\n";
    out << "@_TaxatorTK_Version:" << program_version << "\n\n";
    out << "@@SequenceID\tTaxID\t_TaxatorTK_Support\t_TaxatorTK_Length\n" << std::flush;

    for (auto& g : grouped_records) {
      const std::string& identifier = g.first;
      RecordGroup& records = g.second;
      if (records.empty()) continue;
      std::unique_ptr<Prediction> combined;
      const Prediction* prec;
      if (records.size() > 1) {
        combined = combinePredictionRanges(records, identifier, tax.get(), o.signal_majority_per_sequence,
                                           (medium_unsigned_int)o.min_support_per_sequence, binning_debug_output);
        prec = combined.get();
      } else prec = &records.front();
      if (prec->upper != root_node && !pid_per_rank.empty()) {  // user-defined identity constraints, :312-333
        const double seqlen = static_cast<double>(prec->qlen);
        float min_pid = 0.f;
        uint32_t predict_node = root_node;
        const uint32_t target_node = prec->upper;
        const float rank_pid = (float)(prec->getSupportAtNode(target_node) / seqlen);
        std::vector<uint32_t> chain;  // root ... target
        for (uint32_t n = target_node; ; n = tax->parent[n]) { chain.push_back(n); if (n == root_node) break; }
        std::reverse(chain.begin(), chain.end());
        for (size_t k = 1; k < chain.size(); ++k) {
          const auto f = pid_per_rank.find(tax->rank[chain[k]]);
          if (f != pid_per_rank.end()) min_pid = std::max(min_pid, f->second);
          binning_debug_output << "constraint ctrl: " << rank_pid << " >= " << min_pid << " ?" << std::endl;
          if (rank_pid < min_pid) break;
          predict_node = chain[k];
        }
        out << identifier << '\t' << tax->taxid[predict_node] << '\t' << prec->getSupportAtNode(predict_node) << '\t' << prec->qlen << "\n";
      } else {
        out << identifier << '\t' << tax->taxid[prec->upper] << '\t' << prec->getSupportAtNode(prec->upper) << '\t' << prec->qlen << "\n";
      }
    }
    out << std::flush;
    std::cerr << " done" << std::endl;
    return EXIT_SUCCESS;
  } catch (Exception& e) {
    std::cerr << "An unrecoverable error occurred." << std::endl;  // :341-345
    std::cerr << e.what() << std::endl;
    return EXIT_FAILURE;
  }
}
