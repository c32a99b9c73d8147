// taxator-tk_amd/host/taxonpredictionmodel.cpp -- see taxonpredictionmodel.hh.
#include "taxonpredictionmodel.hh"

#include <algorithm>
#include <chrono>
#include <iostream>
#include <memory>
#include <mutex>
#include <thread>

#include "columnar.hh"
#include "exception.hh"

namespace taxator {

static void check(rpa_engine* e, int rc) {
  if (rc == RPA_OK) return;
  const std::string msg = rpa_last_error(e);
  switch (rc) {  // back to the reference's exception kinds (core/src/exception.hh:29-73)
    case RPA_ERR_RANGE: throw SequenceRangeError(msg);
    case RPA_ERR_SEQ_ID: throw SequenceNotFound(msg);
    case RPA_ERR_DEVICE: throw DeviceError(msg);
    default: throw GeneralError(msg);
  }
}

RPAPredictionModel::RPAPredictionModel(const Taxonomy* tax, const SequenceStore& q_storage, const SequenceStore& db_storage,
                                       float exclude_factor, float reeval_bandwidth, bool protein, int device)
    : TaxonPredictionModel(tax), query_sequences_(q_storage), db_sequences_(db_storage) {
  rpa_taxonomy t{(uint32_t)tax->size(), tax->parent.data(), tax->left.data(), tax->right.data(), tax->depth.data()};
  rpa_store_bits_ = protein ? 8 : 4;
  rpa_params p;
  rpa_default_params(&p);
  p.alphabet = protein ? RPA_PROTEIN : RPA_DNA;
  p.exclude_alignments_factor = exclude_factor;
  p.reeval_bandwidth = reeval_bandwidth;
  p.device = device;
  rpa_engine* created = nullptr;
  int rc = rpa_create(&t, &p, &created);
  if (rc != RPA_OK) throw DeviceError(rpa_last_error(nullptr));
  std::unique_ptr<rpa_engine, void (*)(rpa_engine*)> guard(created, rpa_destroy);  // a store that fails to load must not leak the engine
  const SequenceStore* stores[2] = {&q_storage, &db_storage};
  const int which[2] = {RPA_STORE_QUERY, RPA_STORE_DB};
  for (int k = 0; k < 2; ++k) {
    const SequenceStore& st = *stores[k];
    if (st.indexed()) {  // nothing resident yet: regions come in as the alignment stream refers to them (makeResident)
      std::vector<uint64_t> lengths(st.size());
      for (uint32_t i = 0; i < st.size(); ++i) lengths[i] = st.length(i);
      check(created, rpa_store_reserve_windows(created, which[k], st.size(), lengths.data()));
    } else {
      check(created, rpa_store_load(created, which[k], st.size(), st.letters().data(), st.offsets().data(), RPA_MEM_HOST));
    }
  }
  engine_ = guard.release();
}

namespace {
inline bool bit(const std::vector<uint64_t>& v, uint64_t i) { return (v[i >> 6] >> (i & 63)) & 1ull; }
inline void set_bit(std::vector<uint64_t>& v, uint64_t i) { v[i >> 6] |= 1ull << (i & 63); }
}  // namespace

void RPAPredictionModel::releaseResident() {
  { std::lock_guard<std::mutex> g(resident_mu_); if (in_flight_) --in_flight_; }
  resident_cv_.notify_all();
}

namespace {
struct ResidentHold {  // releases the hold on every way out of a predict call
  RPAPredictionModel& m; bool held;
  ~ResidentHold() { if (held) m.releaseResident(); }
};
}  // namespace

// SURVEY quirk 6, consciously normalised: the reference's IN-MEMORY store answers getSequenceReverseComplement with the whole
// forward sequence (core/src/sequencestorage.hh:122-130: the body is commented out, `return db_seq`), so `taxator -f ref.fna`
// without `-i` aligns every reverse-strand hit's query range against the entire reference sequence; its indexed store
// (:364-369, what the pipelines use: extended/binning-last.bash:104-105) reverse-complements the segment.  This engine
// does the latter for both kinds of store; the first reverse-strand hit against an in-memory reference store says so once.
void RPAPredictionModel::noteReverseStrand(const rpa_batch& b) {
  if (rpa_store_bits_ != 4 || db_sequences_.indexed() || reverse_noted_.load(std::memory_order_relaxed)) return;
  const uint32_t n = std::min<uint32_t>(b.n_records, 65536u);  // (a sample per batch: a note, not a check)
  for (uint32_t r = 0; r < n; ++r)
    if (b.rstart[r] > b.rstop[r]) {
      if (!reverse_noted_.exchange(true))
        std::cerr << "taxator: note: reverse-strand alignments are re-aligned against the reverse complement of their reference segment.  The reference "
                     "implementation does that only with an indexed reference store (-i); its in-memory store (-f without -i) hands back the whole "
                     "forward sequence instead (core/src/sequencestorage.hh:122-130), so its output differs from this one on such alignments." << std::endl;
      return;
    }
}

size_t RPAPredictionModel::makeResident(const rpa_batch& b, bool hold) {
  noteReverseStrand(b);
  if (!query_sequences_.indexed() && !db_sequences_.indexed()) return 0;
  std::unique_lock<std::mutex> resident_lock(resident_mu_);
  if (b.n_sets == 0) { if (hold) ++in_flight_; return 0; }
  // how far a hit's reference range can be extended: by the width of its record set's query range (:523; the range of
  // ALL records of the set here, masked ones included: an upper bound)
  std::vector<uint32_t> span(b.n_sets, 0);
  {
    const unsigned T = (unsigned)std::max<size_t>(1, std::min<size_t>(threads_, b.n_sets / 16384 + 1));
    auto work = [&](unsigned t) {
      for (size_t s = (size_t)b.n_sets * t / T; s < (size_t)b.n_sets * (t + 1) / T; ++s) {
        uint32_t lo = 0xFFFFFFFFu, hi = 0;
        for (uint32_t r = b.set_offsets[s]; r < b.set_offsets[s + 1]; ++r) { lo = std::min(lo, b.qstart[r]); hi = std::max(hi, b.qstop[r]); }
        span[s] = hi >= lo ? hi - lo : 0;
      }
    };
    std::vector<std::thread> th;
    for (unsigned t = 1; t < T; ++t) th.emplace_back(work, t);
    work(0);
    for (auto& x : th) x.join();
  }
  size_t loaded = 0;
  if (query_sequences_.indexed()) loaded += makeResidentStore(RPA_STORE_QUERY, b, span, resident_lock);
  if (db_sequences_.indexed()) loaded += makeResidentStore(RPA_STORE_DB, b, span, resident_lock);
  if (hold) ++in_flight_;
  return loaded;
}

// One store.  Pass 1 (all host threads): every region the batch reads -> the tiles under it; a tile that is not resident
// is noted in the thread's own bit set.  Pass 2: the union of those, grown to whole runs of (resident or needed) tiles --
// a new window swallows the resident windows it touches, so that resident windows never sit side by side and "all tiles
// of a region resident" means "one window holds the region".  Pass 3: fetch and append.
size_t RPAPredictionModel::makeResidentStore(int which, const rpa_batch& b, const std::vector<uint32_t>& span, std::unique_lock<std::mutex>& lk) {
  const SequenceStore& st = which == RPA_STORE_QUERY ? query_sequences_ : db_sequences_;
  std::vector<uint64_t>& base = tile_base_[which];
  std::vector<uint64_t>& res = resident_[which];
  if (base.empty()) {
    base.assign((size_t)st.size() + 1, 0);
    for (uint32_t i = 0; i < st.size(); ++i) base[i + 1] = base[i] + (st.length(i) + tile_ - 1) / tile_;
    res.assign((size_t)(base.back() + 63) / 64 + 1, 0);
  }
  const uint64_t n_tiles = base.back();
  const size_t n_units = which == RPA_STORE_QUERY ? b.n_sets : b.n_records;
  bool emptied = false;  // the store was emptied for this batch: what it needs now is all there will be
  typedef std::chrono::steady_clock Clock;
  auto since = [](Clock::time_point a) { return std::chrono::duration<double>(Clock::now() - a).count(); };
  ResidentStats& rs = resident_stats_[which == RPA_STORE_QUERY ? 0 : 1];
  for (;;) {
    const Clock::time_point t_scan = Clock::now();
    const unsigned T = (unsigned)std::max<size_t>(1, std::min<size_t>(threads_, n_units / 65536 + 1));
    std::vector<std::vector<uint64_t>> need(T);
    std::vector<uint8_t> out_of_range(T, 0), any(T, 0);
    auto scan = [&](unsigned t) {
      std::vector<uint64_t>& mine = need[t];
      auto region = [&](uint32_t seq, uint64_t first, uint64_t end) {  // residues [first, end) of sequence seq, already clipped
        if (end <= first) return;
        const uint64_t t0 = base[seq] + first / tile_, t1 = base[seq] + (end - 1) / tile_;
        for (uint64_t k = t0; k <= t1; ++k)
          if (!bit(res, k)) {
            if (mine.empty()) mine.assign(res.size(), 0);
            set_bit(mine, k);
            any[t] = 1;
          }
      };
      if (which == RPA_STORE_QUERY) {
        for (size_t s = (size_t)b.n_sets * t / T; s < (size_t)b.n_sets * (t + 1) / T; ++s) {
          const uint32_t seq = b.query_seq[s];
          if (seq >= st.size()) { out_of_range[t] = 1; return; }
          uint32_t lo = 0xFFFFFFFFu, hi = 0;
          for (uint32_t r = b.set_offsets[s]; r < b.set_offsets[s + 1]; ++r) { lo = std::min(lo, b.qstart[r]); hi = std::max(hi, b.qstop[r]); }
          if (hi < lo || lo == 0) continue;
          region(seq, lo - 1, std::min<uint64_t>(hi, st.length(seq)));
        }
      } else {
        for (size_t s = (size_t)b.n_sets * t / T; s < (size_t)b.n_sets * (t + 1) / T; ++s) {
          const uint64_t ext = span[s];
          for (uint32_t r = b.set_offsets[s]; r < b.set_offsets[s + 1]; ++r) {
            const uint32_t seq = b.ref_seq[r];
            if (seq >= st.size()) { out_of_range[t] = 1; return; }
            const uint64_t lo = std::min(b.rstart[r], b.rstop[r]), hi = std::max(b.rstart[r], b.rstop[r]);
            if (lo == 0) continue;
            const uint64_t first = lo - 1 > ext ? lo - 1 - ext : 0;
            region(seq, first, std::min<uint64_t>(hi + ext, st.length(seq)));
          }
        }
      }
    };
    {
      std::vector<std::thread> th;
      for (unsigned t = 1; t < T; ++t) th.emplace_back(scan, t);
      scan(0);
      for (auto& x : th) x.join();
    }
    for (uint8_t bad : out_of_range)
      if (bad) throw SequenceNotFound("sequence index outside the store");
    std::vector<uint64_t> want;
    for (unsigned t = 0; t < T; ++t) {
      if (!any[t]) continue;
      if (want.empty()) want.swap(need[t]);
      else for (size_t k = 0; k < want.size(); ++k) want[k] |= need[t][k];
    }
    if (want.empty()) { rs.scan_s += since(t_scan); return 0; }
    // runs of (resident | wanted) tiles that hold a wanted tile, sequence by sequence
    std::vector<rpa_window> wins;
    fetch_offsets_.assign(1, 0);
    for (size_t wi = 0; wi < want.size(); ++wi) {
      uint64_t word = want[wi];
      while (word) {
        const uint64_t k = (uint64_t)wi * 64 + (uint64_t)__builtin_ctzll(word);
        word &= word - 1;
        if (k >= n_tiles) break;
        const uint32_t seq = (uint32_t)(std::upper_bound(base.begin(), base.end(), k) - base.begin() - 1);
        if (!wins.empty() && wins.back().seq_index == seq && k < base[seq] + (wins.back().first + wins.back().n + tile_ - 1) / tile_) continue;  // inside the run just made
        uint64_t a = k, e = k + 1;
        while (a > base[seq] && (bit(res, a - 1) || bit(want, a - 1))) --a;
        while (e < base[seq + 1] && (bit(res, e) || bit(want, e))) ++e;
        rpa_window w{};
        w.seq_index = seq;
        w.first = (a - base[seq]) * tile_;
        w.n = std::min<uint64_t>((e - base[seq]) * tile_, st.length(seq)) - w.first;
        wins.push_back(w);
        fetch_offsets_.push_back(fetch_offsets_.back() + w.n);
      }
    }
    const uint64_t total = fetch_offsets_.back();
    rs.scan_s += since(t_scan);
    if (!emptied) {
      const rpa_store_info info = storeInfo(which);
      const uint64_t half_bytes = rpa_store_bits_ / 4;  // per residue: 4-bit DNA = 1 half byte, 8-bit protein = 2
      const bool over_budget = store_budget_ && (info.residues + total) * half_bytes / 2 > store_budget_ && info.residues;
      // Windows are only ever added: a new window that swallows resident ones leaves their residues behind in the arena.  When
      // more than half of the arena (and more than garbage_slack_ residues) is such garbage, emptying the store and fetching
      // what this batch needs again is the compaction (the regions come from the mapped FASTA).
      const bool garbage = info.residues > 2 * info.live_residues && info.residues - info.live_residues > garbage_slack_;
      if (over_budget || garbage) {
        // Drop everything resident and start over with what this batch needs -- once no batch that was made resident earlier
        // is still running on what is there (the wait lets go of the lock: another thread may come through here meanwhile, so
        // the scan above is stale afterwards and is done again either way).
        if (in_flight_) { const Clock::time_point t_w = Clock::now(); resident_cv_.wait(lk, [&] { return in_flight_ == 0; }); rs.wait_s += since(t_w); continue; }
        check(engine_, rpa_store_reset(engine_, which));
        ++rs.resets;
        std::fill(res.begin(), res.end(), 0);
        emptied = true;
        continue;
      }
    }
    if (store_budget_ && total * (rpa_store_bits_ / 4) / 2 > store_budget_)
      throw GeneralError("one block of the alignment stream needs more sequence than --store-gb allows: use smaller blocks (--block-mb) or a larger budget");
    // Fetch and append in pieces of about a GiB of letters (whole windows): the staging buffer keeps its pages from block to block --
    // the first blocks of a stream over a large collection want tens of GB at once (round 5, 32 Gbases: a fresh 48 GB buffer was
    // mostly page faults) -- and what is resident in the engine is marked resident here piece by piece.
    const uint64_t piece_residues = (uint64_t)1 << 30;
    const unsigned F0 = threads_;
    std::exception_ptr err;
    std::mutex err_mu;
    for (size_t w0 = 0; w0 < wins.size();) {
      size_t w1 = w0 + 1;
      while (w1 < wins.size() && fetch_offsets_[w1 + 1] - fetch_offsets_[w0] <= piece_residues) ++w1;
      const uint64_t base_off = fetch_offsets_[w0], piece_total = fetch_offsets_[w1] - base_off;
      const Clock::time_point t_fetch = Clock::now();
      if (fetch_letters_.size() < (size_t)piece_total + 8) fetch_letters_.resize((size_t)piece_total + 8);
      piece_offsets_.resize(w1 - w0 + 1);
      for (size_t k = w0; k <= w1; ++k) piece_offsets_[k - w0] = fetch_offsets_[k] - base_off;
      const unsigned F = (unsigned)std::max<size_t>(1, std::min<size_t>(F0, w1 - w0));
      auto fetch = [&](unsigned t) {  // slices of equal residue count, not of equal window count
        const size_t lo = std::lower_bound(piece_offsets_.begin(), piece_offsets_.end() - 1, piece_total * t / F) - piece_offsets_.begin();
        const size_t hi = t + 1 == F ? w1 - w0 : std::lower_bound(piece_offsets_.begin(), piece_offsets_.end() - 1, piece_total * (t + 1) / F) - piece_offsets_.begin();
        try { for (size_t k = lo; k < hi; ++k) st.fetch_region(wins[w0 + k].seq_index, wins[w0 + k].first, wins[w0 + k].n, fetch_letters_.data() + piece_offsets_[k]); }
        catch (...) { std::lock_guard<std::mutex> g(err_mu); if (!err) err = std::current_exception(); }
      };
      {
        std::vector<std::thread> th;
        for (unsigned t = 1; t < F; ++t) th.emplace_back(fetch, t);
        fetch(0);
        for (auto& x : th) x.join();
      }
      if (err) std::rethrow_exception(err);
      rs.fetch_s += since(t_fetch);
      const Clock::time_point t_app = Clock::now();
      check(engine_, rpa_store_append_windows(engine_, which, (uint32_t)(w1 - w0), wins.data() + w0, fetch_letters_.data(), piece_offsets_.data()));
      rs.append_s += since(t_app);
      for (size_t k = w0; k < w1; ++k) {  // these windows' tiles are resident now
        const uint64_t t0 = base[wins[k].seq_index] + wins[k].first / tile_, t1 = base[wins[k].seq_index] + (wins[k].first + wins[k].n - 1) / tile_;
        for (uint64_t t = t0; t <= t1; ++t) set_bit(res, t);
      }
      w0 = w1;
    }
    rs.residues += total; rs.windows += wins.size(); ++rs.blocks;
    return wins.size();
  }
  return 0;
}

void RPAPredictionModel::pinHost(void* p, size_t bytes) const { check(engine_, rpa_host_register(engine_, p, bytes)); }
void RPAPredictionModel::unpinHost(void* p) const { check(engine_, rpa_host_unregister(engine_, p)); }

rpa_store_info RPAPredictionModel::storeInfo(int which) const {
  rpa_store_info info{};
  check(engine_, rpa_store_info_get(engine_, which, &info));
  return info;
}

RPAPredictionModel::~RPAPredictionModel() { rpa_destroy(engine_); }

void RPAPredictionModel::predict(const std::vector<RecordSetType>& recordsets, std::vector<PredictionRecord>& precs, std::ostream&) {
  const size_t S = recordsets.size();
  if (precs.size() < S) precs.resize(S, PredictionRecord(tax_));
  set_offsets_.assign(1, 0);
  query_seq_.clear(); query_len_.clear();
  qstart_.clear(); qstop_.clear(); rstart_.clear(); rstop_.clear(); identities_.clear(); alnlen_.clear(); ref_seq_.clear();
  node_.clear(); score_.clear(); filtered_.clear();
  for (const RecordSetType& rset : recordsets) {
    if (rset.empty()) throw GeneralError("empty record set");
    const AlignmentRecordTaxonomy& front = rset.front();  // initPredictionRecord: id and length of the first record
    query_seq_.push_back(query_sequences_.getIndex(front.getQueryIdentifier()));
    query_len_.push_back(front.getQueryLength());
    for (const AlignmentRecordTaxonomy& r : rset) {
      qstart_.push_back(r.getQueryStart()); qstop_.push_back(r.getQueryStop());
      rstart_.push_back(r.getReferenceStart()); rstop_.push_back(r.getReferenceStop());
      score_.push_back(r.getScore()); identities_.push_back(r.getIdentities()); alnlen_.push_back(r.getAlignmentLength());
      ref_seq_.push_back(r.reference_seq); node_.push_back(r.getReferenceNode()); filtered_.push_back(r.isFiltered() ? 1 : 0);
    }
    set_offsets_.push_back((uint32_t)qstart_.size());
  }
  rpa_batch b;
  b.n_sets = (uint32_t)S; b.n_records = (uint32_t)qstart_.size();
  b.set_offsets = set_offsets_.data(); b.query_seq = query_seq_.data(); b.query_len = query_len_.data();
  b.qstart = qstart_.data(); b.qstop = qstop_.data(); b.rstart = rstart_.data(); b.rstop = rstop_.data(); b.score = score_.data();
  b.identities = identities_.data(); b.alnlen = alnlen_.data(); b.ref_seq = ref_seq_.data(); b.node = node_.data();
  b.filtered = filtered_.data(); b.mem_kind = RPA_MEM_HOST;
  rows_.resize(S);
  stats_.resize(S);
  makeResident(b, true);
  ResidentHold hold{*this, query_sequences_.indexed() || db_sequences_.indexed()};
  if (trace_on_) predictTraced(b, rows_.data(), stats_.data(), trace_);
  else check(engine_, rpa_predict_batch(engine_, &b, rows_.data(), stats_.data(), RPA_MEM_HOST));
  for (size_t s = 0; s < S; ++s) {
    const rpa_prediction& r = rows_[s];
    PredictionRecord& prec = precs[s];
    prec.initialize(recordsets[s].front().getQueryIdentifier(), recordsets[s].front().getQueryLength());
    prec.setQueryFeatureBegin(r.begin);
    prec.setQueryFeatureEnd(r.end);
    if (!(r.flags & RPA_FLAG_IVAL_UNTOUCHED)) stale_ival_ = r.ival;
    prec.setInterpolationValue(stale_ival_);
    prec.setSignalStrength(0.f);  // anchors_taxsig ends at 0 on every path that sets it (:722-725,828); printed as "0"
    prec.setNodeRange(r.lower, r.upper, r.support);
    prec.setBestReferenceTaxon(r.rtax);
  }
}

// A logging run: synchronous, one traced call at a time.  The trace holds one 32-byte entry per alignment the control flow
// consumes -- pass0 + pass1 + pass2 of the STATS line, about 1.3 per record on typical data, at most a few per record.
// Sized for 4 per record first (a 128 MB block of 1.5 M records: 190 MB instead of the 3 GB that 64 per record cost, and no
// zero-filled host buffer of that size per call); a batch that needs more is run again with exactly what its counters ask for.
void RPAPredictionModel::predictTraced(const rpa_batch& b, rpa_prediction* rows, rpa_stats* stats, std::vector<rpa_trace_entry>& out) {
  std::lock_guard<std::mutex> turn(trace_mu_);
  uint64_t cap = (uint64_t)b.n_records * 4 + 4096;
  for (int attempt = 0;; ++attempt) {
    check(engine_, rpa_trace_begin(engine_, cap));
    check(engine_, rpa_predict_batch(engine_, &b, rows, stats, RPA_MEM_HOST));
    uint64_t n = 0, dropped = 0;
    check(engine_, rpa_trace_read(engine_, nullptr, 0, &n, &dropped));  // how many there are
    const uint64_t total = n + dropped;
    if (total <= cap) {
      trace_buf_.reset(new rpa_trace_entry[(size_t)total + 1]);  // (not value-initialised)
      check(engine_, rpa_trace_read(engine_, trace_buf_.get(), total, &n, &dropped));
      out.assign(trace_buf_.get(), trace_buf_.get() + n);
      break;
    }
    if (attempt) { check(engine_, rpa_trace_begin(engine_, 0)); throw GeneralError("alignment log: more alignments than the trace buffer holds"); }
    cap = total + 16;  // results do not depend on tracing: the second run consumes the same alignments
  }
  check(engine_, rpa_trace_begin(engine_, 0));
}

void RPAPredictionModel::predictColumns(const ColumnarBatch& cb, size_t lo, size_t hi, rpa_prediction* rows, rpa_stats* stats,
                                        std::vector<rpa_trace_entry>* trace) {
  if (hi <= lo) return;
  // (re-entrant: the driver keeps two batches in flight per engine from two host threads)
  const uint32_t base = cb.set_offsets[lo];
  std::vector<uint32_t> rebased;
  const uint32_t* offsets = cb.set_offsets.data() + lo;
  if (base != 0) {
    rebased.resize(hi - lo + 1);
    for (size_t s = lo; s <= hi; ++s) rebased[s - lo] = cb.set_offsets[s] - base;
    offsets = rebased.data();
  }
  rpa_batch b;
  b.n_sets = (uint32_t)(hi - lo); b.n_records = cb.set_offsets[hi] - base;
  b.set_offsets = offsets; b.query_seq = cb.query_seq.data() + lo; b.query_len = cb.query_len.data() + lo;
  b.qstart = cb.rec.qstart.data() + base; b.qstop = cb.rec.qstop.data() + base; b.rstart = cb.rec.rstart.data() + base;
  b.rstop = cb.rec.rstop.data() + base; b.score = cb.rec.score.data() + base; b.identities = cb.rec.identities.data() + base;
  b.alnlen = cb.rec.alnlen.data() + base; b.ref_seq = cb.rec.ref_seq.data() + base; b.node = cb.rec.node.data() + base;
  b.filtered = cb.rec.filtered.data() + base; b.mem_kind = RPA_MEM_HOST;
  makeResident(b, true);
  ResidentHold hold{*this, query_sequences_.indexed() || db_sequences_.indexed()};  // until the batch has left the engine
  if (trace) { predictTraced(b, rows, stats, *trace); return; }
  // asynchronous boundary: the engine runs two jobs at a time, the copies of one under the kernels of the other
  rpa_job* job = nullptr;
  check(engine_, rpa_submit_batch(engine_, &b, rows, stats, &job));
  check(engine_, rpa_wait(engine_, job));
}

// ---- alignment-free models ----
static rpa_lca_params lca_params(int model) {
  rpa_lca_params p;
  rpa_lca_default_params(&p);
  p.model = model;
  return p;
}

LCAPredictionModelBase::LCAPredictionModelBase(const Taxonomy* tax, const rpa_lca_params& params, int device)
    : TaxonPredictionModel(tax), params_(params) {
  rpa_taxonomy t{(uint32_t)tax->size(), tax->parent.data(), tax->left.data(), tax->right.data(), tax->depth.data()};
  rpa_params p;
  rpa_default_params(&p);
  p.device = device;
  int rc = rpa_create(&t, &p, &engine_);
  if (rc != RPA_OK) throw DeviceError(rpa_last_error(nullptr));
}

LCAPredictionModelBase::~LCAPredictionModelBase() { rpa_destroy(engine_); }

void LCAPredictionModelBase::predict(const std::vector<RecordSetType>& recordsets, std::vector<PredictionRecord>& precs, std::ostream&) {
  const size_t S = recordsets.size();
  if (precs.size() < S) precs.resize(S, PredictionRecord(tax_));
  set_offsets_.assign(1, 0);
  query_len_.clear(); qstart_.clear(); qstop_.clear(); node_.clear(); score_.clear(); evalue_.clear(); filtered_.clear();
  for (const RecordSetType& rset : recordsets) {
    if (rset.empty()) throw GeneralError("empty record set");
    query_len_.push_back(rset.front().getQueryLength());  // initPredictionRecord: id and length of the first record
    for (const AlignmentRecordTaxonomy& r : rset) {
      qstart_.push_back(r.getQueryStart()); qstop_.push_back(r.getQueryStop()); score_.push_back(r.getScore());
      evalue_.push_back(r.evalue); node_.push_back(r.getReferenceNode()); filtered_.push_back(r.isFiltered() ? 1 : 0);
    }
    set_offsets_.push_back((uint32_t)qstart_.size());
  }
  rpa_batch b{};
  b.n_sets = (uint32_t)S; b.n_records = (uint32_t)qstart_.size();
  b.set_offsets = set_offsets_.data(); b.query_len = query_len_.data(); b.qstart = qstart_.data(); b.qstop = qstop_.data();
  b.score = score_.data(); b.node = node_.data(); b.filtered = filtered_.data(); b.mem_kind = RPA_MEM_HOST;
  rows_.resize(S);
  check(engine_, rpa_lca_predict_batch(engine_, &b, evalue_.data(), tax_->is_unclassified.empty() ? nullptr : tax_->is_unclassified.data(),
                                       &params_, rows_.data(), RPA_MEM_HOST));
  for (size_t s = 0; s < S; ++s) {
    const rpa_prediction& r = rows_[s];
    PredictionRecord& prec = precs[s];
    prec.initialize(recordsets[s].front().getQueryIdentifier(), recordsets[s].front().getQueryLength());
    prec.setQueryFeatureBegin(r.begin);
    prec.setQueryFeatureEnd(r.end);
    prec.setNodeRange(r.lower, r.upper, r.support);  // setNodePoint
    prec.setBestReferenceTaxon(r.rtax);
    // interpolation value and signal strength keep what the record was constructed with (-1, 0): never printed / "0"
  }
}

DummyPredictionModel::DummyPredictionModel(const Taxonomy* tax, int device) : LCAPredictionModelBase(tax, lca_params(RPA_LCA_DUMMY), device) {}
LCASimplePredictionModel::LCASimplePredictionModel(const Taxonomy* tax, int device)
    : LCAPredictionModelBase(tax, lca_params(RPA_LCA_SIMPLE), device) {}
static rpa_lca_params megan_params(bool ignore_unclassified, float toppercent, float minscore, int minsupport, double maxevalue) {
  rpa_lca_params p = lca_params(RPA_LCA_MEGAN);
  p.ignore_unclassified = ignore_unclassified ? 1 : 0; p.toppercent = toppercent; p.minscore = minscore;
  p.minsupport = (uint32_t)minsupport; p.maxevalue = maxevalue;
  return p;
}
MeganLCAPredictionModel::MeganLCAPredictionModel(const Taxonomy* tax, bool ignore_unclassified, float toppercent, float minscore,
                                                 int minsupport, double maxevalue, int device)
    : LCAPredictionModelBase(tax, megan_params(ignore_unclassified, toppercent, minscore, minsupport, maxevalue), device) {}
static rpa_lca_params nbest_params(int n) { rpa_lca_params p = lca_params(RPA_LCA_NBEST); p.nbest = (uint32_t)n; return p; }
NBestLCAPredictionModel::NBestLCAPredictionModel(const Taxonomy* tax, int n, int device)
    : LCAPredictionModelBase(tax, nbest_params(n), device) {}

}  // namespace taxator
