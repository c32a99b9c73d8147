// taxator-tk_amd/host/taxator.cpp -- `taxator` driver for the MI355X RPA engine.
// Mirrors core/taxator.cpp: same option names/defaults (:267-294), taxonomy from $TAXATORTK_TAXONOMY_NCBI pruned to the
// given ranks (:337-340), seqid->taxid mapping (:342), alignments streamed from stdin, GFF3 to stdout in input order
// (doPredictionsSerial, :50-77), errors to stderr + EXIT_FAILURE (:381-386).  -a rpa (default) and the alignment-free
// models dummy | simple-lca | megan-lca | ic-megan-lca | n-best-lca with their options -e -c -m -n -u (:346-361; these
// need no FASTA files and run through the per-record object loop); -p sets the number of HOST threads that parse the input and format the output
// (0 or absent = all); -i / -v select the indexed store (sequencestorage.hh): the FASTA is memory-mapped and a sequence
// goes to HBM when the alignment stream first refers to it, instead of the whole file being loaded up front;
// new: --device N | --devices a,b,... (one engine and one host thread per GPU, batches dealt to whichever is free,
// output in input order: the replacement of doPredictionsParallel, :181-210), --batch, --block-mb, --timing,
// --record-objects.
//
// Default data path (no per-record objects): a reader thread turns blocks of the TSV stream into columnar batches
// (columnar.hh) while the main thread runs the previous batch on the GPU and a writer thread formats the one before
// that into GFF3 (gff3writer.hh).  --record-objects selects the loop that mirrors the reference's object model
// (RecordSetGenerator -> std::vector<AlignmentRecordTaxonomy> -> PredictionRecord); both produce the same bytes.
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <exception>
#include <fstream>
#include <future>
#include <iostream>
#include <map>
#include <memory>
#include <mutex>
#include <sstream>
#include <string>
#include <thread>
#include <vector>

#include "accessconv.hh"
#include "alignmentrecord.hh"
#include "channel.hh"
#include "columnar.hh"
#include "hostcpus.hh"
#include "gff3writer.hh"
#include "exception.hh"
#include "predictionrecord.hh"
#include "sequencestorage.hh"
#include "stage.hh"
#include "taxonomy.hh"
#include "taxonpredictionmodel.hh"

using namespace taxator;

static const char* program_version = "1.5.0-mi355x";  // core/src/constants.hh:47

struct Options {
  std::string algorithm = "rpa", mapping, query, query_index, db, db_index, logfile = "/dev/null", dataformat = "nucleotide";
  std::vector<std::string> ranks;
  bool split_alignments = true, alignments_sorted = false, delete_unmarked = true;
  float filterout = 0.5f, toppercent = 0.05f, minscore = 0.0f;
  size_t readers = 0;  // --readers N: reader teams over a regular input file (0: four / two / one by the number of host threads, at least one per engine)
  double maxevalue = 1000.0;
  unsigned minsupport = 1, nbest = 1;
  bool ignore_unclassified = false;
  unsigned processors = 0;
  bool record_objects = false;
  double block_mb = 128;
  std::vector<int> devices;  // --devices a,b,...; empty: --device
  int device = 0;
  bool timing = false;
  double store_gb = 0;       // --store-gb: HBM budget per indexed store and engine (0 = none)
  uint64_t store_tile = 0;   // --store-tile: residues per tile of the indexed stores (0 = 65536)
  size_t batch = 0;  // 0: one GPU call per parsed block (columnar path) / 65536 record sets (record-object path)
};

static void usage(std::ostream& o) {
  o << "Allowed options:\n"
       "  -h [ --help ]                      show help message\n"
       "  -V [ --version ]                   show program version\n"
       "  -a [ --algorithm ] arg (=rpa)      rpa | dummy | simple-lca | megan-lca | ic-megan-lca | n-best-lca\n"
       "  -g [ --seqid-taxid-mapping ] arg   filename of seqid->taxid mapping for reference\n"
       "  -q [ --query-sequences ] arg       query sequences FASTA\n"
       "  -v [ --query-sequences-index ] arg query sequences FASTA index (.fai): sequences go to HBM when first referred to\n"
       "  -f [ --ref-sequences ] arg         reference sequences FASTA\n"
       "  -i [ --ref-sequences-index ] arg   reference sequences FASTA index (.fai): sequences go to HBM when first referred to\n"
       "  -p [ --processors ] arg (=0)       host threads for parsing and output formatting, 0 = all\n"
       "  -l [ --logfile ] arg (=/dev/null)  per record set STATS lines are appended here\n"
       "  -b [ --dataformat ] arg (=nucleotide)  nucleotide | protein\n"
       "  advanced: -r ranks..., -s split-alignments (=1), -o alignments-sorted (=0), -d delete-notranks (=1; -d 0 keeps the whole\n"
       "            tree: up to 63 levels below the root; a hit row holds 832 bits of taxon path labels, 8 bits per level for record\n"
       "            sets of up to 256 records, 16 up to 65536, 32 beyond: 63 / 52 / 26 levels; making the labels costs\n"
       "            records^2 x levels / 8 comparisons per record set -- seconds for a single set of 50 000 records),\n"
       "            -x heuristic-cutoff (=0.5), -t toppercent (=0.05), -e max-evalue (=1000), -c min-support (=1), -m minscore (=0),\n"
       "            -n nbest (=1), -u ignore-unclassified, --device N (=0), --devices a,b,... (several GPUs), --batch N (record sets per GPU call),\n"
       "            --block-mb N (=128, input block parsed at a time), --timing, --record-objects (per-record object loop),\n"
       "            --store-gb G (HBM budget of an indexed store, -i / -v: when the resident regions would outgrow it the store is\n"
       "            emptied and refilled with what the block at hand needs; 0 = no limit), --store-tile N (=65536 residues: the\n"
       "            granularity at which regions of a sequence become resident), --readers N (reader teams over a regular input file:\n"
       "            the file is cut into slices where the query identifier changes and the teams parse slices side by side;\n"
       "            0 = four teams on eight or more host threads, two on four to seven, one below that, and at least one per engine;\n"
       "            at most one team per host thread, 64 at most; a pipe has one reader)\n";
}

namespace {

struct Work {
  ColumnarBatch batch;
  std::vector<rpa_prediction> rows;
  std::vector<rpa_stats> stats;
  std::vector<rpa_trace_entry> trace;  // -l: the alignments the control flow consumed, sorted by (set, order)
  bool last = false;            // end of stream (or a failure of the reader): no batch inside
  bool failed = false;          // a GPU call failed on this block: nothing from here on is printed
  bool err_after = false;       // the input failed BEHIND the record sets of this block (reader teams): they are printed, then the run ends with err
  std::exception_ptr err;
  uint64_t seq = 0;             // position in the input: the writer prints in this order
  std::vector<std::pair<void*, size_t>> pinned;       // the arrays below that are page-locked
  std::vector<std::pair<void*, size_t>> pinned_want;  // ... and what was asked for last (some may have been refused)
  unsigned uses = 0;            // blocks this object has carried to a GPU (pin_work)
};

// The columns of a Work object are reused from block to block: page-lock them (once per growth of a vector) so that the
// engine's host<->device copies are DMA transfers that run beside the kernels of the other job in flight.
template <class T, class A> void add_range(std::vector<std::pair<void*, size_t>>& v, std::vector<T, A>& a) {
  // only arrays big enough to have a mapping of their own (glibc: >= 128 KiB): small ones share heap pages with their
  // neighbours, and a page cannot be locked twice
  if (a.capacity() * sizeof(T) >= ((size_t)1 << 18)) v.emplace_back((void*)a.data(), a.capacity() * sizeof(T));
}
void unpin_work(Work& w, const RPAPredictionModel& rpa);
// Registration never outlives an allocation: the reader unpins through ColumnarBatch::before_regrow before a column
// grows, the GPU thread through room_for_rows before rows / stats grow; a range is registered again once it has settled.
void room_for_rows(Work& w, size_t S, const RPAPredictionModel& rpa) {
  if (S > w.rows.capacity() || S > w.stats.capacity()) {
    unpin_work(w, rpa);
    w.rows.reserve(S + S / 4); w.stats.reserve(S + S / 4);
  }
  w.rows.resize(S);
  w.stats.resize(S);
}
void pin_work(Work& w, const RPAPredictionModel& rpa) {
  // Page-locking ~100 MB (the columns of a 128 MB block) costs 30-40 ms, a copy from pageable memory ~10 ms more than one
  // from locked memory: an object is locked when it carries its THIRD block (1 M record sets are 8 blocks on 3-6 objects:
  // locking them all was 0.25 s of a 0.6 s run; 10 M record sets reuse every object a dozen times).
  if (++w.uses < 3) return;
  std::vector<std::pair<void*, size_t>> now;
  ColumnarBatch& b = w.batch;
  add_range(now, b.set_offsets); add_range(now, b.query_seq); add_range(now, b.query_len);
  add_range(now, b.rec.qstart); add_range(now, b.rec.qstop); add_range(now, b.rec.rstart); add_range(now, b.rec.rstop);
  add_range(now, b.rec.identities); add_range(now, b.rec.alnlen); add_range(now, b.rec.ref_seq); add_range(now, b.rec.node);
  add_range(now, b.rec.score); add_range(now, b.rec.filtered); add_range(now, w.rows); add_range(now, w.stats);
  if (now == w.pinned || (now.size() == w.pinned_want.size() && now == w.pinned_want)) return;
  w.pinned_want = now;
  for (auto& r : w.pinned) { try { rpa.unpinHost(r.first); } catch (...) {} }
  w.pinned.clear();
  for (auto& r : now) {  // page-locking is an optimisation: an array the driver refuses to lock is copied from pageable memory
    try { rpa.pinHost(r.first, r.second); w.pinned.push_back(r); } catch (...) {}
  }
}
void unpin_work(Work& w, const RPAPredictionModel& rpa) {
  for (auto& r : w.pinned) { try { rpa.unpinHost(r.first); } catch (...) {} }
  w.pinned.clear();
  w.pinned_want.clear();  // (an address that comes back after a reallocation is a new mapping: nothing may match it)
}

// One record set of the log (-l), the lines of the reference's log this build can reproduce (taxonpredictionmodelsequence.hh:
// 360-364 / 375-380 / 407-409 "ID <qrstart>:<qrstop>@<query id>" ("-1:-1@id" for a set without active alignments) and NUMREF,
// :533,636,782,802 one "+ALN" line per alignment the control flow computes -- same operands, distance and similarity as
// there, std::fixed << std::setprecision(2), :347; the reference's other fields of those lines (search scores, cut-offs),
// the printed alignments and the timing columns are left out; "pass=" is this build's own field and comes last --, :837
// "STATS <name> n pass0 pass1 pass2 total" with the timing columns left out, followed by the blank line the reference prints).
inline void append_fixed2(std::string& out, float v) {
  char buf[64];
  const int n = snprintf(buf, sizeof(buf), "%.2f", (double)v);
  out.append(buf, (size_t)(n > 0 ? n : 0));
}
inline void append_seqname(std::string& lg, const char* qid, size_t qid_len, const rpa_stats& st, const rpa_prediction& row) {
  if (st.n == 0) lg.append("-1:-1@");  // seqname_fmt % -1 % -1 % qid, :360
  else { append_u32(lg, row.begin); lg.push_back(':'); append_u32(lg, row.end); lg.push_back('@'); }  // "%d:%d@%s", :344,375,407
  lg.append(qid, qid_len);
}
inline void append_log_set(std::string& lg, const char* qid, size_t qid_len, const rpa_stats& st, const rpa_prediction& row,
                           const rpa_trace_entry* tr, size_t n_tr) {
  lg.append("ID\t"); append_seqname(lg, qid, qid_len, st, row); lg.append("\n  NUMREF\t"); append_u32(lg, st.n); lg.push_back('\n');
  for (size_t k = 0; k < n_tr; ++k) {
    const rpa_trace_entry& e = tr[k];
    lg.append("    +ALN ");
    if (e.i < 0) lg.append("query"); else append_u32(lg, (uint32_t)e.i);
    lg.append(" <=> ");
    if (e.j < 0) lg.append("query"); else append_u32(lg, (uint32_t)e.j);
    lg.append("\tdist="); append_fixed2(lg, e.dist); lg.append("; sim="); append_fixed2(lg, e.sim);
    lg.append("; pass="); append_u32(lg, (uint32_t)e.pass); lg.push_back('\n');
  }
  lg.append("STATS\t"); append_seqname(lg, qid, qid_len, st, row);
  const uint32_t v[5] = {st.n, st.pass0, st.pass1, st.pass2, st.pass0 + st.pass1 + st.pass2};
  for (uint32_t x : v) { lg.push_back('\t'); append_u32(lg, x); }
  lg.append("\n\n");
}

// parse: the busiest reader's seconds (with several teams parsing side by side the SUM of their seconds can exceed the wall time and
// says nothing about what bounds the pipeline); parse_sum: all teams' thread-seconds
struct Timing { double parse = 0, parse_sum = 0, predict = 0, write = 0; size_t sets = 0, records = 0; };

typedef std::chrono::steady_clock Clock;
double secs(Clock::time_point a, Clock::time_point b) { return std::chrono::duration<double>(b - a).count(); }

// reader thread(s) -> two GPU threads per engine -> writer thread; 4 x engines + teams Work objects in circulation.  The
// reference's doPredictionsParallel (core/taxator.cpp:181-210) has N consumer threads call predict() on one model and
// prints in whatever order they finish; here a parsed block goes to whichever GPU is free and the writer puts the
// blocks back into input order (what the reference's serial loop prints, :69-74).
void run_columnar(std::vector<std::unique_ptr<ColumnarReader>>& readers, SliceCutter* cutter, std::vector<RPAPredictionModel*>& models, const Taxonomy& tax,
                  std::ostream* logsink, size_t max_sets, unsigned threads, Timing& tm) {
  // two host threads per engine: each submits a block and waits for it (rpa_submit_batch / rpa_wait), so every engine has
  // two batches in flight -- the host->device copies of one run under the kernels of the other
  const size_t n_gpu = 2 * models.size();
  const size_t n_teams = cutter ? readers.size() : 1;
  std::vector<Work> works(2 * n_gpu + n_teams);
  Channel<Work*> free_q, parsed_q, predicted_q;
  for (Work& w : works) free_q.push(&w);
  std::atomic<bool> stop(false);
  std::mutex tm_mu;
  // ONE reader (a pipe, --readers 1, few host threads): blocks in stream order, the open group at a block's end carried to the next.
  // SEVERAL reader teams (a regular file on stdin: four by default, at least one per engine): the file is cut into slices at query-identifier
  // boundaries (SliceCutter), every team parses whole slices with its share of the host threads, the slice index is the
  // block's place in the output -- the feed scales with the host's cores instead of ending at one reader's ~100 M lines/s.
  // (A team takes a Work object BEFORE it asks for a slice, so the earliest slice not yet printed always owns one: the writer,
  // which holds finished blocks until their turn, cannot starve it.)
  std::vector<std::thread> reader_threads;
  std::atomic<bool> last_sent(false);
  auto reader_loop = [&](size_t team) {
    ColumnarReader& reader = *readers[team];
    uint64_t seq = 0;
    double busy = 0;  // this team's parse seconds
    for (;;) {
      Work* w = free_q.pop();
      if (!w) break;
      w->last = false; w->failed = false; w->err_after = false; w->err = nullptr;
      const Clock::time_point t0 = Clock::now();
      if (!w->batch.before_regrow) w->batch.before_regrow = [w, &models] { unpin_work(*w, *models[0]); };
      if (!cutter) {
        w->seq = seq++;
        if (stop.load()) { w->last = true; parsed_q.push(w); break; }  // a GPU call failed: end the stream here
        try { if (!reader.next(w->batch)) w->last = true; }
        catch (...) { w->err = std::current_exception(); w->last = true; }
      } else {
        const char *b = nullptr, *e = nullptr;
        uint64_t index = 0;
        if (stop.load()) cutter->close();  // the run is ending (a GPU call or a line failed): no further slice is handed out
        const bool got = cutter->next(b, e, index);
        if (!got) {  // the file is exhausted or closed: ONE end-of-stream block, behind every slice handed out (index = their number)
          if (last_sent.exchange(true)) { free_q.push(w); break; }
          w->batch.clear(); w->seq = index; w->last = true;
          parsed_q.push(w);
          break;
        }
        w->seq = index;
        // (an exception out of the parser itself -- std::bad_alloc from the column buffers, a failing unregistration -- must take the
        // same ordered way out as a bad line: what precedes this slice is printed, the engines are destroyed, never std::terminate)
        try { reader.parse_slice(*cutter, b, e, w->batch, w->err); }
        catch (...) { w->err = std::current_exception(); w->batch.clear(); }
        if (w->err) { w->err_after = true; stop.store(true); cutter->close(); }  // what precedes the bad line is printed; no further slice is taken
      }
      {
        const double dt = secs(t0, Clock::now());
        busy += dt;
        std::lock_guard<std::mutex> lk(tm_mu);
        tm.parse_sum += dt;
        tm.parse = std::max(tm.parse, busy);
      }
      parsed_q.push(w);
      if (w->last) break;
    }
  };
  for (size_t t = 0; t < n_teams; ++t) reader_threads.emplace_back(reader_loop, t);
  std::thread reader_done([&] {  // every GPU thread sees the end of the stream once the teams have handed over their last block
    for (auto& t : reader_threads) t.join();
    for (size_t k = 0; k < n_gpu; ++k) parsed_q.push(nullptr);
  });
  float stale_ival = -1.f;  // predictionrecord.hh:40; the reference reuses one PredictionRecord (taxator.cpp:66)
  std::thread writer_thread([&] {
    std::vector<float> ival;
    std::vector<std::string> parts, log_parts;
    std::map<uint64_t, Work*> waiting;  // blocks that finished ahead of their turn
    uint64_t next_seq = 0;
    bool finished = false, discard = false;
    while (!finished) {
      Work* got = predicted_q.pop();
      waiting[got->seq] = got;
      while (!waiting.empty() && waiting.begin()->first == next_seq) {
        Work* w = waiting.begin()->second;
        waiting.erase(waiting.begin());
        ++next_seq;
        if (w->last) { finished = true; break; }  // end of stream (or the reader's failure): everything before it has been printed
        if (w->failed) discard = true;            // the serial loop would have died here: blocks behind it are recycled unprinted
        if (discard) { free_q.push(w); continue; }
        const Clock::time_point t0 = Clock::now();
        const ColumnarBatch& b = w->batch;
        const size_t S = b.n_sets();
        ival.resize(S);
        for (size_t s = 0; s < S; ++s) {
          if (!(w->rows[s].flags & RPA_FLAG_IVAL_UNTOUCHED)) stale_ival = w->rows[s].ival;
          ival[s] = stale_ival;
        }
        const unsigned T = (unsigned)std::max<size_t>(1, std::min<size_t>(threads, S / 4096 + 1));
        parts.assign(T, std::string());
        log_parts.assign(T, std::string());
        auto format = [&](unsigned t) {
          const size_t lo = S * t / T, hi = S * (t + 1) / T;
          std::string& out = parts[t];
          out.reserve((hi - lo) * 112);
          for (size_t s = lo; s < hi; ++s)
            append_gff3(out, tax, b.qid_arena.data() + b.qid_off[s], b.qid_len[s], b.query_len[s], w->rows[s], ival[s]);
          if (logsink) {  // ID / NUMREF / +ALN / STATS per record set (append_log_set)
            std::string& lg = log_parts[t];
            size_t k = std::lower_bound(w->trace.begin(), w->trace.end(), (uint32_t)lo,
                                        [](const rpa_trace_entry& e, uint32_t s0) { return e.set < s0; }) - w->trace.begin();
            for (size_t s = lo; s < hi; ++s) {
              size_t k1 = k;
              while (k1 < w->trace.size() && w->trace[k1].set == s) ++k1;
              append_log_set(lg, b.qid_arena.data() + b.qid_off[s], b.qid_len[s], w->stats[s], w->rows[s], w->trace.data() + k, k1 - k);
              k = k1;
            }
          }
        };
        std::vector<std::thread> th;
        for (unsigned t = 1; t < T; ++t) th.emplace_back(format, t);
        format(0);
        for (auto& x : th) x.join();
        for (const std::string& p : parts) fwrite(p.data(), 1, p.size(), stdout);
        if (logsink) for (const std::string& p : log_parts) logsink->write(p.data(), (std::streamsize)p.size());
        tm.write += secs(t0, Clock::now());
        if (w->err_after) discard = true;  // the line behind this block's record sets does not parse: the serial loop dies there;
        free_q.push(w);                    // slices that were already on their way are recycled unprinted until the end-of-stream block
      }
    }
  });
  std::exception_ptr failure;
  std::mutex failure_mu;
  std::atomic<uint64_t> fail_seq(~0ull);  // the earliest block that failed: what the serial loop would have died on
  auto gpu_loop = [&](size_t g) {
    RPAPredictionModel& rpa = *models[g / 2];
    for (;;) {
      Work* w = parsed_q.pop();
      if (!w) return;
      if (w->last) {
        if (w->err) { std::lock_guard<std::mutex> lk(failure_mu); if (w->seq < fail_seq.load()) { fail_seq = w->seq; failure = w->err; } }
        predicted_q.push(w);
        continue;  // (the end of the stream for every GPU thread is the nullptr behind it)
      }
      const Clock::time_point t0 = Clock::now();
      const size_t S = w->batch.n_sets();
      room_for_rows(*w, S, rpa);
      try {
        if (S >= 4096) pin_work(*w, rpa);  // (not worth it for tiny blocks: page-locking costs more than their copies)
        const size_t step = max_sets ? max_sets : S;
        w->trace.clear();
        std::vector<rpa_trace_entry> part;
        for (size_t lo = 0; lo < S; lo += step) {
          const size_t hi = std::min(S, lo + step);
          rpa.predictColumns(w->batch, lo, hi, w->rows.data() + lo, w->stats.data() + lo, logsink ? &part : nullptr);
          if (logsink) {
            for (rpa_trace_entry& e : part) e.set += (uint32_t)lo;
            w->trace.insert(w->trace.end(), part.begin(), part.end());
          }
        }
      } catch (...) {
        { std::lock_guard<std::mutex> lk(failure_mu); if (w->seq < fail_seq.load()) { fail_seq = w->seq; failure = std::current_exception(); } }
        w->failed = true;  // the writer stops printing in front of this block
        stop.store(true);  // and the reader ends the stream at its next block
        predicted_q.push(w);
        continue;          // keep draining: blocks before this one may still be on their way through other GPUs
      }
      {
        std::lock_guard<std::mutex> lk(failure_mu);
        tm.predict += secs(t0, Clock::now()); tm.sets += S; tm.records += w->batch.n_records();
        if (w->err_after && w->seq < fail_seq.load()) { fail_seq = w->seq; failure = w->err; }  // (a GPU failure on the same block came first: it keeps the slot)
      }
      predicted_q.push(w);
    }
  };
  std::vector<std::thread> gpu_threads;
  for (size_t g = 1; g < n_gpu; ++g) gpu_threads.emplace_back(gpu_loop, g);
  gpu_loop(0);
  for (auto& t : gpu_threads) t.join();
  writer_thread.join();
  stop.store(true);
  for (size_t k = 0; k < works.size() + n_teams; ++k) free_q.push(nullptr);  // wake the readers if they wait for a Work object
  reader_done.join();
  for (Work& w : works) unpin_work(w, *models[0]);
  fflush(stdout);
  if (failure) std::rethrow_exception(failure);
}

}  // namespace

static bool parse_bool(const std::string& v) { return !(v == "0" || v == "false" || v == "no" || v == "off"); }

static int run(int argc, char** argv) {
  stage("options");
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
    else if (a == "-e" || a == "--max-evalue") o.maxevalue = atof(val("--max-evalue").c_str());
    else if (a == "-c" || a == "--min-support") o.minsupport = (unsigned)atoi(val("--min-support").c_str());
    else if (a == "-m" || a == "--minscore") o.minscore = (float)atof(val("--minscore").c_str());
    else if (a == "-n" || a == "--nbest") o.nbest = (unsigned)atoi(val("--nbest").c_str());
    else if (a == "-u" || a == "--ignore-unclassified") o.ignore_unclassified = true;
    else if (a == "--device") o.device = atoi(val("--device").c_str());
    else if (a == "--devices") {
      std::stringstream ss(val("--devices"));
      std::string item;
      while (std::getline(ss, item, ',')) if (!item.empty()) o.devices.push_back(atoi(item.c_str()));
    }
    else if (a == "-w" || a == "--db-whitelist") (void)val("--db-whitelist");  // parsed and never used by the reference either (taxator.cpp:294, :220-254)
    else if (a == "--advanced-options") { usage(std::cout); return EXIT_SUCCESS; }
    else if (a == "--citation") { std::cout << "Droege J, Gregor I, McHardy AC. Taxator-tk: precise taxonomic assignment of metagenomes by fast approximation of evolutionary neighborhoods. Bioinformatics 2015." << std::endl; return EXIT_SUCCESS; }
    else if (a == "--timing") o.timing = true;
    else if (a == "--record-objects") o.record_objects = true;
    else if (a == "--block-mb") o.block_mb = atof(val("--block-mb").c_str());
    else if (a == "--store-gb") o.store_gb = atof(val("--store-gb").c_str());
    else if (a == "--store-tile") o.store_tile = (uint64_t)atoll(val("--store-tile").c_str());
    else if (a == "--readers") {
      const long long n = atoll(val("--readers").c_str());
      if (n < 0) { std::cerr << "--readers takes a number of reader teams >= 0 (0 = chosen by the number of host threads)" << std::endl; return EXIT_FAILURE; }
      o.readers = (size_t)std::min<long long>(n, 64);  // (capped again by the host's threads below: a team is a thread, a reader and ~100 MB of page-locked columns)
    }
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
  const bool alignment_free = o.algorithm == "dummy" || o.algorithm == "simple-lca" || o.algorithm == "megan-lca" ||
                              o.algorithm == "ic-megan-lca" || o.algorithm == "n-best-lca";
  if (o.algorithm != "rpa" && !alignment_free) {
    std::cout << "classification algorithm can either be: rpa (default), simple-lca, megan-lca, ic-megan-lca, n-best-lca" << std::endl;  // :376
    return EXIT_FAILURE;
  }
  if (o.dataformat != "nucleotide" && o.dataformat != "protein") {
    std::cout << "data format can either be nucleotide or protein" << std::endl;
    return EXIT_FAILURE;
  }
  const Clock::time_point t_start = Clock::now();
  double t_parse = 0, t_predict = 0, t_write = 0;
  size_t n_sets = 0, n_records = 0;
  if (o.devices.empty()) o.devices.push_back(o.device);
  try {
    stage("taxonomy");
    std::unique_ptr<Taxonomy> tax(loadTaxonomyFromEnvironment(&o.ranks, o.delete_unmarked));
    if (!tax) return EXIT_FAILURE;
    if (tax->getMaxDepth() > 63) {  // rpa_create's limit (kMaxDepth): the full NCBI tree has ~40 levels
      std::cerr << "the taxonomy keeps " << (int)tax->getMaxDepth() << " levels below the root; this build supports 63" << std::endl;
      return EXIT_FAILURE;
    }
    const Clock::time_point t_tax = Clock::now();
    stage("seqid->taxid mapping");
    std::unique_ptr<StrIDConverter> seqid2taxid(loadStrIDConverterFromFile(o.mapping, 1000));
    const Clock::time_point t_map = Clock::now();
    std::ofstream logsink(o.logfile.c_str(), std::ios_base::app);
    if (alignment_free) {  // taxator.cpp:346-361; doPredictionsSerial (:50-77) in batches
      std::unique_ptr<TaxonPredictionModel> model;
      const int dev = o.devices[0];
      stage("alignment-free model: engine");
      if (o.algorithm == "dummy") model.reset(new DummyPredictionModel(tax.get(), dev));
      else if (o.algorithm == "simple-lca") model.reset(new LCASimplePredictionModel(tax.get(), dev));
      else if (o.algorithm == "n-best-lca") model.reset(new NBestLCAPredictionModel(tax.get(), (int)o.nbest, dev));
      else model.reset(new MeganLCAPredictionModel(tax.get(), o.ignore_unclassified, o.toppercent, o.minscore, (int)o.minsupport, o.maxevalue, dev));
      AlignmentRecordFactory fac(*seqid2taxid, tax.get(), nullptr);
      RecordSetGenerator recgen(std::cin, fac, o.split_alignments, o.alignments_sorted);
      std::vector<RecordSetType> sets;
      std::vector<PredictionRecord> precs;
      std::cout << GFF3Header();
      std::ostringstream outbuf;
      const size_t batch_sets = o.batch ? o.batch : (size_t)1 << 16;
      std::exception_ptr input_failure;  // the reference predicts and prints set by set (:69-74): what was read before a bad line is still printed
      while (recgen.notEmpty() && !input_failure) {
        stage("alignment-free model: reading record sets");
        sets.clear();
        try {
          while (recgen.notEmpty() && sets.size() < batch_sets) { sets.emplace_back(); recgen.getNext(sets.back()); }
        } catch (...) { input_failure = std::current_exception(); sets.pop_back(); }
        stage("alignment-free model: predict");
        model->predict(sets, precs, logsink);
        stage("alignment-free model: output");
        outbuf.str("");
        for (size_t s = 0; s < sets.size(); ++s) outbuf << precs[s];
        std::cout << outbuf.str();
      }
      if (input_failure) { std::cout.flush(); std::rethrow_exception(input_failure); }
      return EXIT_SUCCESS;
    }
    // in memory unless an index file is named (core/taxator.cpp:232-249); an indexed store loads nothing here
    stage("sequence stores (host)");
    std::unique_ptr<SequenceStore> query_ptr(o.query_index.empty() ? new SequenceStore(o.query) : new SequenceStore(o.query, o.query_index));
    std::unique_ptr<SequenceStore> db_ptr(o.db_index.empty() ? new SequenceStore(o.db) : new SequenceStore(o.db, o.db_index));
    SequenceStore& query_storage = *query_ptr;
    SequenceStore& db_storage = *db_ptr;
    const unsigned host_threads = o.processors ? o.processors : usable_cpus();
    std::vector<std::unique_ptr<RPAPredictionModel>> engines;
    const Clock::time_point t_stores = Clock::now();
    // the reader's identifier tables (a hash probe per reference / query identifier: ~0.1 s for a million queries) are built on a
    // thread of their own while the engines come up (runtime start-up, taxonomy tables, stores to HBM)
    std::future<std::shared_ptr<const ReaderTables>> tables_future;
    if (!o.record_objects)
      tables_future = std::async(std::launch::async, [&] { return std::make_shared<const ReaderTables>(*seqid2taxid, tax.get(), &db_storage, &query_storage); });
    stage("engines (rpa_create + stores to HBM)");
    for (int dev : o.devices) {
      engines.emplace_back(new RPAPredictionModel(tax.get(), query_storage, db_storage, o.filterout, o.toppercent, o.dataformat == "protein", dev));
      engines.back()->setHostThreads(std::max(1u, host_threads / (unsigned)o.devices.size()));
      if (o.store_tile) engines.back()->setTile(o.store_tile);
      if (o.store_gb > 0) engines.back()->setStoreBudget((uint64_t)(o.store_gb * 1073741824.0));
      if (const char* v = getenv("TAXATOR_STORE_GARBAGE")) engines.back()->setGarbageSlack((uint64_t)atoll(v));  // (tests: residues of swallowed windows tolerated)
    }
    RPAPredictionModel& rpa = *engines[0];
    const bool logging_on = o.logfile != "/dev/null";
    if (logging_on) rpa.setTrace(true);
    const Clock::time_point t_loaded = Clock::now();
    if (o.timing)  // where the loading goes: taxonomy dump | mapping | FASTA files on the host | engines (runtime start-up, tables, stores to HBM)
      std::cerr << "LOAD\ttaxonomy_s=" << secs(t_start, t_tax) << "\tmapping_s=" << secs(t_tax, t_map) << "\tfasta_s=" << secs(t_map, t_stores)
                << "\tengines_s=" << secs(t_stores, t_loaded) << std::endl;
    auto store_report = [&]() {  // what is resident in HBM (per engine; the engines hold the same sequences)
      if (!o.timing) return;
      for (int which = 0; which < 2; ++which) {
        const rpa_store_info si = rpa.storeInfo(which == 0 ? RPA_STORE_QUERY : RPA_STORE_DB);
        std::cerr << "STORE\t" << (which == 0 ? "query" : "ref") << "\t" << ((which == 0 ? query_storage : db_storage).indexed() ? "indexed" : "in-memory")
                  << "\tsequences=" << si.n_seqs << "\tresident=" << si.n_resident << "\tresidues=" << si.residues << "\tdevice_bytes=" << si.device_bytes
                  << "\twindows=" << si.windows << "\tlive_residues=" << si.live_residues << std::endl;
        if ((which == 0 ? query_storage : db_storage).indexed()) {  // engine 0's share of making regions resident (every engine fetches for itself)
          const RPAPredictionModel::ResidentStats rs = rpa.residentStats(which == 0 ? RPA_STORE_QUERY : RPA_STORE_DB);
          std::cerr << "RESIDENT\t" << (which == 0 ? "query" : "ref") << "\tscan_s=" << rs.scan_s << "\tfetch_s=" << rs.fetch_s << "\tappend_s=" << rs.append_s
                    << "\twait_s=" << rs.wait_s << "\tfetched_residues=" << rs.residues << "\twindows_appended=" << rs.windows << "\tblocks_that_fetched=" << rs.blocks
                    << "\tstore_emptied=" << rs.resets << "\tfetch_MB_per_s=" << (rs.fetch_s > 0 ? rs.residues / rs.fetch_s / 1e6 : 0.0)
                    << "\tappend_MB_per_s=" << (rs.append_s > 0 ? rs.residues / rs.append_s / 1e6 : 0.0) << std::endl;
        }
      }
    };
    if (!o.record_objects) {
      std::cout << GFF3Header() << std::flush;
      // Reader teams when stdin is a regular file (--readers N forces N): at least one per engine, and four (two on fewer than
      // eight host threads) even for ONE engine -- a team parses its slice of the mapped file in place, while the single
      // reader of a stream copies every block out of the page cache first (4 M record sets, one engine: 0.99 s with one
      // reader, 0.78 with two teams, 0.73 with four; profiles/r04_cli_teams_one_engine.txt).  A pipe or --readers 1: ONE
      // reader on all host threads.
      const size_t block_bytes = (size_t)std::max(4096.0, o.block_mb * 1048576.0);
      const unsigned host_cpus = o.processors ? o.processors : usable_cpus();
      const size_t default_teams = host_cpus >= 8 ? 4 : (host_cpus >= 4 ? 2 : 1);
      size_t teams = o.readers ? std::min<size_t>(o.readers, std::max<size_t>(1, host_cpus)) : std::max(engines.size(), default_teams);
      std::unique_ptr<SliceCutter> cutter;
      if (teams > 1) {
        cutter.reset(new SliceCutter(stdin, block_bytes));
        if (!cutter->mapped()) { cutter.reset(); teams = 1; }
      }
      const unsigned reader_threads_total = o.processors ? o.processors : usable_cpus();
      std::vector<std::unique_ptr<ColumnarReader>> readers;
      std::shared_ptr<const ReaderTables> tables = tables_future.get();
      for (size_t t = 0; t < teams; ++t) {
        const unsigned th = teams == 1 ? o.processors : std::max(1u, (unsigned)((reader_threads_total + teams - 1 - t) / teams));
        readers.emplace_back(new ColumnarReader(stdin, *seqid2taxid, tax.get(), &db_storage, &query_storage, o.split_alignments, th, block_bytes,
                                                o.alignments_sorted, tables));
        tables = readers.back()->tables();
      }
      if (o.timing) std::cerr << "READERS	teams=" << teams << "	" << (cutter ? "slices of the mapped input file cut at query identifier boundaries" : "one stream") << std::endl;
      Timing tm;
      const bool logging = o.logfile != "/dev/null";
      std::vector<RPAPredictionModel*> models;
      for (auto& m : engines) models.push_back(m.get());
      stage("columnar pipeline (reader / GPU / writer threads)");
      try { run_columnar(readers, cutter.get(), models, *tax, logging ? &logsink : nullptr, o.batch, host_threads, tm); }
      catch (...) {
        stage("columnar pipeline failed: unwinding (engines are destroyed)");
        if (o.timing) std::cerr << "TIMING\tincomplete" << std::endl;
        throw;
      }
      stage("columnar pipeline done: reports, then engines are destroyed");
      store_report();
      if (o.timing)
        std::cerr << "TIMING\tload_s=" << secs(t_start, t_loaded) << "\tparse_s=" << tm.parse << "\tpredict_s=" << tm.predict << "\twrite_s="
                  << tm.write << "\twall_s=" << secs(t_loaded, Clock::now()) << "\trecord_sets=" << tm.sets << "\trecords=" << tm.records
                  << "\tgpus=" << o.devices.size() << "\tparse_sum_s=" << tm.parse_sum << "\treader_teams=" << teams
                  << "\t(columnar pipeline: the three stages overlap; parse_s = the busiest reader team, parse_sum_s = all teams; predict_s = GPU-thread seconds, two threads per engine)" << std::endl;
      return EXIT_SUCCESS;
    }
    stage("record-object loop: first record");
    AlignmentRecordFactory fac(*seqid2taxid, tax.get(), &db_storage);
    RecordSetGenerator recgen(std::cin, fac, o.split_alignments, o.alignments_sorted);
    std::vector<RecordSetType> sets;
    std::vector<PredictionRecord> precs;
    std::cout << GFF3Header();
    std::ostringstream outbuf;
    const size_t batch_sets = o.batch ? o.batch : (size_t)1 << 16;
    std::exception_ptr input_failure;  // the reference predicts and prints set by set (:69-74): what was read before a bad line is still printed
    while (recgen.notEmpty() && !input_failure) {
      stage("record-object loop: reading record sets");
      const Clock::time_point t0 = Clock::now();
      sets.clear();
      try {
        while (recgen.notEmpty() && sets.size() < batch_sets) {
          sets.emplace_back();
          recgen.getNext(sets.back());
          n_records += sets.back().size();
        }
      } catch (...) { input_failure = std::current_exception(); sets.pop_back(); }
      n_sets += sets.size();
      const Clock::time_point t1 = Clock::now();
      stage("record-object loop: predict");
      rpa.predict(sets, precs, logsink);
      stage("record-object loop: output");
      const Clock::time_point t2 = Clock::now();
      outbuf.str("");
      for (size_t s = 0; s < sets.size(); ++s) outbuf << precs[s];
      std::cout << outbuf.str();
      if (logging_on) {
        const std::vector<rpa_stats>& st = rpa.lastStats();
        const std::vector<rpa_trace_entry>& tr = rpa.lastTrace();
        std::string lg;
        size_t k = 0;
        for (size_t s = 0; s < sets.size(); ++s) {
          size_t k1 = k;
          while (k1 < tr.size() && tr[k1].set == s) ++k1;
          const std::string& qid = precs[s].getQueryIdentifier();
          append_log_set(lg, qid.data(), qid.size(), st[s], rpa.lastRows()[s], tr.data() + k, k1 - k);
          k = k1;
        }
        logsink << lg;
      }
      const Clock::time_point t3 = Clock::now();
      t_parse += secs(t0, t1); t_predict += secs(t1, t2); t_write += secs(t2, t3);
    }
    if (input_failure) { std::cout.flush(); std::rethrow_exception(input_failure); }
    if (o.timing)
      std::cerr << "TIMING\tload_s=" << secs(t_start, t_loaded) << "\tparse_s=" << t_parse << "\tpredict_s=" << t_predict << "\twrite_s="
                << t_write << "\trecord_sets=" << n_sets << "\trecords=" << n_records << std::endl;
    store_report();
    stage("record-object loop done: engines are destroyed");
    return EXIT_SUCCESS;
  } catch (Exception& e) {
    stage("error reported: leaving main");
    std::cerr << "An unrecoverable error occurred: " << e.what() << std::endl;  // taxator.cpp:381-386
    return EXIT_FAILURE;
  } catch (std::exception& e) {  // e.g. std::bad_alloc on a huge input: same way out
    stage("error reported: leaving main");
    std::cerr << "An unrecoverable error occurred: " << e.what() << std::endl;
    return EXIT_FAILURE;
  }
}

// Leaving.  When run() returns, every engine has been destroyed (rpa_destroy: streams synchronised and destroyed, device
// and pinned memory freed), stdout / the log file are complete, and nothing of this process is left on the GPU that the
// kernel driver does not reclaim by itself.  What a plain `return` from main would still do is run the exit handlers of the
// HIP / HSA runtime -- and those can spin for ever when the process leaves within a few milliseconds of creating a
// CU-masked stream (round 2's engine streams; plain streams since): tools/hang_hunt.py caught it twice in 676 runs of `taxator --record-objects` on an input whose first line
// does not parse (stage "exit handlers", engine "rpa_destroy: done", main thread running inside the runtime's teardown,
// its helper threads asleep in kfd_wait_on_events; DESIGN.md section 5 has the stacks).  It is what hung round 2's GPU
// test run.  So the driver flushes its streams and leaves with _exit: no handler of anybody's runs, the exit code is the
// same.  TAXATOR_RUN_EXIT_HANDLERS=1 restores the plain return (the hunt's control arm).
int main(int argc, char** argv) {
  setenv("GPU_MAX_HW_QUEUES", "16", 0);  // before the HIP runtime comes up: a hardware queue per stream of the jobs in flight (rpa_engine.hip: make_stream)
  install_stage_reporter();  // SIGUSR1 / SIGTERM / TAXATOR_WATCHDOG_S: say where the process is (stage.hh)
  atexit([] { stage("exit handlers (runtime teardown)"); });
  const int rc = run(argc, argv);
  stage("run() returned");
  std::cout.flush();
  std::cerr.flush();
  fflush(nullptr);
  const char* keep = getenv("TAXATOR_RUN_EXIT_HANDLERS");
  if (!(keep && atoi(keep) != 0)) _exit(rc);
  return rc;
}
