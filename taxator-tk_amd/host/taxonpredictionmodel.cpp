// taxator-tk_amd/host/taxonpredictionmodel.cpp -- see taxonpredictionmodel.hh.
#include "taxonpredictionmodel.hh"

#include <algorithm>
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
    if (st.indexed()) {  // nothing resident yet: sequences come in as the alignment stream refers to them
      check(created, rpa_store_reserve(created, which[k], st.size()));
      resident_[which[k]].assign(st.size(), 0);
    } else {
      check(created, rpa_store_load(created, which[k], st.size(), st.letters().data(), st.offsets().data(), RPA_MEM_HOST));
    }
  }
  engine_ = guard.release();
}

size_t RPAPredictionModel::makeResident(int which, const uint32_t* idx, size_t n) {
  const SequenceStore& st = which == RPA_STORE_QUERY ? query_sequences_ : db_sequences_;
  if (!st.indexed() || n == 0) return 0;
  std::lock_guard<std::mutex> resident_lock(resident_mu_);
  std::vector<uint8_t>& res = resident_[which];
  // candidates: every host thread scans a slice of the indices against the (read-only) residency marks ...
  const unsigned T = (unsigned)std::max<size_t>(1, std::min<size_t>(threads_, n / 65536 + 1));
  std::vector<std::vector<uint32_t>> cand(T);
  std::vector<uint8_t> out_of_range(T, 0);  // (an exception must not leave a helper thread: noted, thrown after the join)
  auto scan = [&](unsigned t) {
    const size_t lo = n * t / T, hi = n * (t + 1) / T;
    uint32_t last = 0xFFFFFFFFu;
    for (size_t k = lo; k < hi; ++k) {
      const uint32_t i = idx[k];
      if (i == last) continue;
      last = i;
      if (i >= res.size()) { out_of_range[t] = 1; return; }
      if (!res[i]) cand[t].push_back(i);
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
  // ... then one pass lists them (a sequence several threads found is taken once); they are marked resident only when the
  // engine has them, so a fetch that fails leaves nothing half done
  std::vector<uint32_t> need;
  for (auto& c : cand)
    for (uint32_t i : c)
      if (!res[i]) { res[i] = 2; need.push_back(i); }  // 2 = listed in this call
  auto unlist = [&] { for (uint32_t i : need) res[i] = 0; };
  if (need.empty()) return 0;
  fetch_offsets_.assign(need.size() + 1, 0);
  for (size_t k = 0; k < need.size(); ++k) fetch_offsets_[k + 1] = fetch_offsets_[k] + st.length(need[k]);
  fetch_letters_.resize((size_t)fetch_offsets_.back() + 8);
  const unsigned F = (unsigned)std::max<size_t>(1, std::min<size_t>(threads_, need.size()));
  std::exception_ptr err;
  std::mutex err_mu;
  auto fetch = [&](unsigned t) {  // slices of equal residue count, not of equal sequence count
    const uint64_t total = fetch_offsets_.back();
    const size_t lo = std::lower_bound(fetch_offsets_.begin(), fetch_offsets_.end() - 1, total * t / F) - fetch_offsets_.begin();
    const size_t hi = t + 1 == F ? need.size() : std::lower_bound(fetch_offsets_.begin(), fetch_offsets_.end() - 1, total * (t + 1) / F) - fetch_offsets_.begin();
    try { for (size_t k = lo; k < hi; ++k) st.fetch(need[k], fetch_letters_.data() + fetch_offsets_[k]); }
    catch (...) { std::lock_guard<std::mutex> g(err_mu); if (!err) err = std::current_exception(); }
  };
  {
    std::vector<std::thread> th;
    for (unsigned t = 1; t < F; ++t) th.emplace_back(fetch, t);
    fetch(0);
    for (auto& x : th) x.join();
  }
  if (err) { unlist(); std::rethrow_exception(err); }
  try {
    check(engine_, rpa_store_append(engine_, which, (uint32_t)need.size(), need.data(), fetch_letters_.data(), fetch_offsets_.data()));
  } catch (...) { unlist(); throw; }
  for (uint32_t i : need) res[i] = 1;
  return need.size();
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
  makeResident(RPA_STORE_QUERY, query_seq_.data(), query_seq_.size());
  makeResident(RPA_STORE_DB, ref_seq_.data(), ref_seq_.size());
  check(engine_, rpa_predict_batch(engine_, &b, rows_.data(), stats_.data(), RPA_MEM_HOST));
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

void RPAPredictionModel::predictColumns(const ColumnarBatch& cb, size_t lo, size_t hi, rpa_prediction* rows, rpa_stats* stats) {
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
  makeResident(RPA_STORE_QUERY, b.query_seq, b.n_sets);
  makeResident(RPA_STORE_DB, b.ref_seq, b.n_records);
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
