// taxator-tk_amd/host/taxator.cpp -- `taxator` driver for the MI355X RPA engine.
// Mirrors core/taxator.cpp: same option names/defaults (:267-294), taxonomy from $TAXATORTK_TAXONOMY_NCBI pruned to the
// given ranks (:337-340), seqid->taxid mapping (:342), alignments streamed from stdin, GFF3 to stdout in input order
// (doPredictionsSerial, :50-77), errors to stderr + EXIT_FAILURE (:381-386).  Only `-a rpa` exists here (the LCA-only
// models are out of scope); -p is accepted and ignored (parallelism = the GPU); -i/-v index files are
// accepted and ignored (the stores are resident in HBM, no on-disk index is needed); new: --device, --batch.
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <memory>
#include <sstream>
#include <string>
#include <vector>

#include "accessconv.hh"
#include "alignmentrecord.hh"
#include "exception.hh"
#include "predictionrecord.hh"
#include "sequencestorage.hh"
#include "taxonomy.hh"
#include "taxonpredictionmodel.hh"

using namespace taxator;

static const char* program_version = "1.5.0-mi355x";  // core/src/constants.hh:47

struct Options {
  std::string algorithm = "rpa", mapping, query, query_index, db, db_index, logfile = "/dev/null", dataformat = "nucleotide";
  std::vector<std::string> ranks;
  bool split_alignments = true, alignments_sorted = false, delete_unmarked = true;
  float filterout = 0.5f, toppercent = 0.05f;
  unsigned processors = 1;
  int device = 0;
  size_t batch = 1u << 16;
};

static void usage(std::ostream& o) {
  o << "Allowed options:\n"
       "  -h [ --help ]                      show help message\n"
       "  -V [ --version ]                   show program version\n"
       "  -a [ --algorithm ] arg (=rpa)      prediction algorithm (only rpa in this build)\n"
       "  -g [ --seqid-taxid-mapping ] arg   filename of seqid->taxid mapping for reference\n"
       "  -q [ --query-sequences ] arg       query sequences FASTA\n"
       "  -v [ --query-sequences-index ] arg accepted, unused (store is resident in HBM)\n"
       "  -f [ --ref-sequences ] arg         reference sequences FASTA\n"
       "  -i [ --ref-sequences-index ] arg   accepted, unused (store is resident in HBM)\n"
       "  -p [ --processors ] arg (=1)       accepted, unused (parallelism is the GPU)\n"
       "  -l [ --logfile ] arg (=/dev/null)  per record set STATS lines are appended here\n"
       "  -b [ --dataformat ] arg (=nucleotide)  nucleotide | protein\n"
       "  advanced: -r ranks..., -s split-alignments (=1), -o alignments-sorted (=0), -d delete-notranks (=1),\n"
       "            -x heuristic-cutoff (=0.5), -t toppercent (=0.05), --device N (=0), --batch N (=65536 record sets)\n";
}

static bool parse_bool(const std::string& v) { return !(v == "0" || v == "false" || v == "no" || v == "off"); }

int main(int argc, char** argv) {
  Options o;
  for (int i = 1; i < argc; ++i) {
    std::string a = argv[i];
    auto val = [&](const char* name) -> std::string {
      if (i + 1 >= argc) { std::cerr << "the required argument for option '" << name << "' is missing" << std::endl; exit(EXIT_FAILURE); }
      return argv[++i];
    };
    if (a == "-h" || a == "--help") { usage(std::cout); return EXIT_SUCCESS; }
    else if (a == "-V" || a == "--version") { std::cout << program_version << std::endl; return EXIT_SUCCESS; }
    else if (a == "-a" || a == "--algorithm") o.algorithm = val("--algorithm");
    else if (a == "-g" || a == "--seqid-taxid-mapping") o.mapping = val("--seqid-taxid-mapping");
    else if (a == "-q" || a == "--query-sequences") o.query = val("--query-sequences");
    else if (a == "-v" || a == "--query-sequences-index") o.query_index = val("--query-sequences-index");
    else if (a == "-f" || a == "--ref-sequences") o.db = val("--ref-sequences");
    else if (a == "-i" || a == "--ref-sequences-index") o.db_index = val("--ref-sequences-index");
    else if (a == "-p" || a == "--processors") o.processors = (unsigned)atoi(val("--processors").c_str());
    else if (a == "-l" || a == "--logfile") o.logfile = val("--logfile");
    else if (a == "-b" || a == "--dataformat") o.dataformat = val("--dataformat");
    else if (a == "-s" || a == "--split-alignments") o.split_alignments = parse_bool(val("--split-alignments"));
    else if (a == "-o" || a == "--alignments-sorted") o.alignments_sorted = parse_bool(val("--alignments-sorted"));
    else if (a == "-d" || a == "--delete-notranks") o.delete_unmarked = parse_bool(val("--delete-notranks"));
    else if (a == "-x" || a == "--heuristic-cutoff") o.filterout = (float)atof(val("--heuristic-cutoff").c_str());
    else if (a == "-t" || a == "--toppercent") o.toppercent = (float)atof(val("--toppercent").c_str());
    else if (a == "--device") o.device = atoi(val("--device").c_str());
    else if (a == "--batch") o.batch = (size_t)atoll(val("--batch").c_str());
    else if (a == "-r" || a == "--ranks") { while (i + 1 < argc && argv[i + 1][0] != '-') o.ranks.push_back(argv[++i]); }
    else { std::cerr << "unrecognised option '" << a << "'" << std::endl; return EXIT_FAILURE; }
  }
  if (o.ranks.empty()) o.ranks = default_ranks;  // taxator.cpp:324-326
  if (o.mapping.empty()) {
    std::cout << "Specify a taxonomy mapping file for the reference sequence identifiers" << std::endl;
    usage(std::cout);
    return EXIT_FAILURE;
  }
  if (o.algorithm != "rpa") {
    std::cout << "classification algorithm can only be rpa in this build (the LCA-only models are not part of the accelerated path)" << std::endl;
    return EXIT_FAILURE;
  }
  if (o.dataformat != "nucleotide" && o.dataformat != "protein") {
    std::cout << "data format can either be nucleotide or protein" << std::endl;
    return EXIT_FAILURE;
  }
  std::unique_ptr<Taxonomy> tax(loadTaxonomyFromEnvironment(&o.ranks, o.delete_unmarked));
  if (!tax) return EXIT_FAILURE;
  try {
    std::unique_ptr<StrIDConverter> seqid2taxid(loadStrIDConverterFromFile(o.mapping, 1000));
    std::ofstream logsink(o.logfile.c_str(), std::ios_base::app);
    SequenceStore query_storage(o.query);
    SequenceStore db_storage(o.db);
    RPAPredictionModel rpa(tax.get(), query_storage, db_storage, o.filterout, o.toppercent, o.dataformat == "protein", o.device);
    AlignmentRecordFactory fac(*seqid2taxid, tax.get(), &db_storage);
    RecordSetGenerator recgen(std::cin, fac, o.split_alignments);
    std::vector<RecordSetType> sets;
    std::vector<PredictionRecord> precs;
    std::cout << GFF3Header();
    std::ostringstream outbuf;
    while (recgen.notEmpty()) {
      sets.clear();
      while (recgen.notEmpty() && sets.size() < o.batch) {
        sets.emplace_back();
        recgen.getNext(sets.back());
      }
      rpa.predict(sets, precs, logsink);
      outbuf.str("");
      for (size_t s = 0; s < sets.size(); ++s) outbuf << precs[s];
      std::cout << outbuf.str();
      const std::vector<rpa_stats>& st = rpa.lastStats();
      for (size_t s = 0; s < sets.size(); ++s)  // the STATS line of the reference log (:837), timing columns as 0
        logsink << "STATS\t" << precs[s].getQueryIdentifier() << '\t' << st[s].n << '\t' << st[s].pass0 << '\t' << st[s].pass1 << '\t'
                << st[s].pass2 << '\t' << (st[s].pass0 + st[s].pass1 + st[s].pass2) << "\n";
    }
    return EXIT_SUCCESS;
  } catch (Exception& e) {
    std::cerr << "An unrecoverable error occurred: " << e.what() << std::endl;  // taxator.cpp:381-386
    return EXIT_FAILURE;
  }
}
