// taxator-tk_amd/host/taxonpredictionmodel.hh -- the plugin seam, batched.
// Mirrors TaxonPredictionModel<ContainerT> (core/src/taxonpredictionmodel.hh:36-52) and RPAPredictionModel
// (core/src/taxonpredictionmodelsequence.hh:326-881).  The reference calls predict(recordset, prec, logsink) once per
// record set; the MI355X engine wants thousands of record sets at a time, so predict() here takes a span of record sets
// and fills one PredictionRecord per set, in input order (what the reference's serial loop prints, taxator.cpp:69-74).
// All computation happens behind the C ABI (include/taxator_rpa.h) in hand-written HIP; there is no CPU path.
#pragma once
#include <atomic>
#include <condition_variable>
#include <memory>
#include <mutex>
#include <ostream>
#include <vector>

#include "../../include/taxator_rpa.h"
#include "alignmentrecord.hh"
#include "predictionrecord.hh"
#include "sequencestorage.hh"
#include "taxonomy.hh"

namespace taxator {

class TaxonPredictionModel {
 public:
  explicit TaxonPredictionModel(const Taxonomy* tax) : tax_(tax) {}
  virtual ~TaxonPredictionModel() {}
  // one PredictionRecord per record set; precs must hold recordsets.size() records (they are reused between calls)
  virtual void predict(const std::vector<RecordSetType>& recordsets, std::vector<PredictionRecord>& precs, std::ostream& logsink) = 0;
 protected:
  const Taxonomy* tax_;
};

class RPAPredictionModel : public TaxonPredictionModel {
 public:
  // same argument meaning as the reference ctor (:329): exclude_factor = -x, reeval_bandwidth = -t
  RPAPredictionModel(const Taxonomy* tax, const SequenceStore& q_storage, const SequenceStore& db_storage, float exclude_factor,
                     float reeval_bandwidth = .1f, bool protein = false, int device = 0);
  ~RPAPredictionModel() override;
  void predict(const std::vector<RecordSetType>& recordsets, std::vector<PredictionRecord>& precs, std::ostream& logsink) override;
  const std::vector<rpa_stats>& lastStats() const { return stats_; }
  // The same call for record sets that are already in columns (columnar.hh): sets [lo, hi) of `batch`, rows and STATS
  // counters written to rows[0 .. hi-lo) / stats[0 .. hi-lo).  No per-record objects on this path.
  // trace != null: the call runs synchronously with the alignment log on (rpa_trace_begin) and *trace receives, sorted by
  // (set - lo, order inside the set), one entry per alignment the reference's control flow computes (its "+ALN" lines)
  void predictColumns(const struct ColumnarBatch& batch, size_t lo, size_t hi, rpa_prediction* rows, rpa_stats* stats,
                      std::vector<rpa_trace_entry>* trace = nullptr);
  // the same for predict(): entries of the last call (set = index into its record sets)
  void setTrace(bool on) { trace_on_ = on; }
  const std::vector<rpa_trace_entry>& lastTrace() const { return trace_; }
  const std::vector<rpa_prediction>& lastRows() const { return rows_; }
  // Indexed stores (taxator -i / -v): make every REGION the batch refers to resident in this engine's HBM if it is not yet
  // -- the reference reads exactly [start, stop] of a sequence through the .fai (sequencestorage.hh:341-369); here a
  // sequence is cut into tiles (65536 residues unless setTile says otherwise) and the tiles under a hit's range, extended
  // by its record set's query range, become resident: fetched from the mapped FASTA by the host threads and appended as
  // windows (rpa_store_append_windows), merged with the resident windows they touch so that any segment lies in one
  // window.  HBM and load time follow the residues the stream touches, not the sequences, let alone the collection.
  // With a budget (setStoreBudget, taxator --store-gb) a store that would outgrow it is emptied first (rpa_store_reset)
  // and refilled with what the batch at hand needs.  predict() / predictColumns() call it; a no-op for in-memory stores.
  // Returns the number of windows that had to be loaded.
  // hold = true: the batch counts as "in flight" from here until releaseResident() -- a store is only ever emptied
  // (budget reached) while no batch that has been made resident is still unfinished; the two GPU threads per engine of
  // the columnar pipeline would otherwise empty the store under each other's batches (a spurious "not resident" failure).
  size_t makeResident(const rpa_batch& b, bool hold = false);
  void releaseResident();
  void noteReverseStrand(const rpa_batch& b);  // quirk 6 (in-memory store and reverse-strand hits): said once on stderr
  void setHostThreads(unsigned t) { threads_ = t ? t : 1; }
  void setTile(uint64_t residues) { tile_ = residues ? residues : 65536; }       // before the first batch
  void setStoreBudget(uint64_t bytes) { store_budget_ = bytes; }                  // 0 = no limit; per store
  void setGarbageSlack(uint64_t residues) { garbage_slack_ = residues; }
  rpa_store_info storeInfo(int which) const;
  // where the time of making regions resident went (taxator --timing: RESIDENT lines), per indexed store, summed over the blocks:
  // scanning the block's columns for tiles that are not resident | reading the regions out of the mapped FASTA (all host threads)
  // | rpa_store_append_windows (waits for the jobs in flight, copies the letters to the device, packs) | waiting for earlier
  // batches before the store may be emptied; residues fetched, windows appended, blocks that fetched anything, times the store was emptied
  struct ResidentStats { double scan_s = 0, fetch_s = 0, append_s = 0, wait_s = 0; uint64_t residues = 0, windows = 0, blocks = 0, resets = 0; };
  ResidentStats residentStats(int which) const { return resident_stats_[which == RPA_STORE_QUERY ? 0 : 1]; }
  // page-lock / release host arrays that are handed to predictColumns() again and again (rpa_host_register)
  void pinHost(void* p, size_t bytes) const;
  void unpinHost(void* p) const;
 private:
  const SequenceStore& query_sequences_;
  const SequenceStore& db_sequences_;
  // per indexed store: tiles of every sequence, numbered consecutively (tile_base_[i] = first tile of sequence i); a
  // tile's bit says "resident".  Resident windows are the maximal runs of resident tiles inside a sequence.
  std::vector<uint64_t> tile_base_[2];
  std::vector<uint64_t> resident_[2];
  uint64_t tile_ = 65536, store_budget_ = 0;
  uint64_t garbage_slack_ = (uint64_t)1 << 30;  // residues of swallowed windows an indexed store may carry before it is emptied and refilled (setGarbageSlack)
  ResidentStats resident_stats_[2];
  unsigned rpa_store_bits_ = 4;  // packed bits per residue in the engine (DNA 4, protein 8): what a residue costs against the budget
  size_t makeResidentStore(int which, const rpa_batch& b, const std::vector<uint32_t>& span, std::unique_lock<std::mutex>& lk);
  std::atomic<bool> reverse_noted_{false};
  unsigned in_flight_ = 0;             // batches between makeResident(hold) and releaseResident(); guarded by resident_mu_
  std::condition_variable resident_cv_;
  std::vector<uint8_t> fetch_letters_;
  std::vector<uint64_t> fetch_offsets_, piece_offsets_;
  unsigned threads_ = 1;
  std::mutex resident_mu_;
  rpa_engine* engine_ = nullptr;
  bool trace_on_ = false;
  std::vector<rpa_trace_entry> trace_;
  std::mutex trace_mu_;  // traced calls are synchronous and take turns
  void predictTraced(const rpa_batch& b, rpa_prediction* rows, rpa_stats* stats, std::vector<rpa_trace_entry>& out);
  std::unique_ptr<rpa_trace_entry[]> trace_buf_;
  float stale_ival_ = -1.f;  // PredictionRecord is reused across calls in the reference (taxator.cpp:66): n==0 keeps the last ival
  std::vector<rpa_stats> stats_;
  std::vector<rpa_prediction> rows_;
  // SoA staging
  std::vector<uint32_t> set_offsets_, query_seq_, query_len_, qstart_, qstop_, rstart_, rstop_, identities_, alnlen_, ref_seq_, node_;
  std::vector<float> score_;
  std::vector<uint8_t> filtered_;
};

// The reference's alignment-free models (core/src/taxonpredictionmodel.hh:57-253), selected by taxator -a
// (core/taxator.cpp:346-361): DummyPredictionModel, LCASimplePredictionModel, MeganLCAPredictionModel (argument order of
// its ctor kept: ignore_unclassified, toppercent, minscore, minsupport, maxevalue), NBestLCAPredictionModel.  Batched like
// RPAPredictionModel; the work is done by rpa_lca_predict_batch (hand-written HIP, no CPU path).
class LCAPredictionModelBase : public TaxonPredictionModel {
 public:
  ~LCAPredictionModelBase() override;
  void predict(const std::vector<RecordSetType>& recordsets, std::vector<PredictionRecord>& precs, std::ostream& logsink) override;
 protected:
  LCAPredictionModelBase(const Taxonomy* tax, const rpa_lca_params& params, int device);
 private:
  rpa_engine* engine_ = nullptr;
  rpa_lca_params params_;
  std::vector<rpa_prediction> rows_;
  std::vector<uint32_t> set_offsets_, query_len_, qstart_, qstop_, node_;
  std::vector<float> score_;
  std::vector<double> evalue_;
  std::vector<uint8_t> filtered_;
};

struct DummyPredictionModel : LCAPredictionModelBase {
  explicit DummyPredictionModel(const Taxonomy* tax, int device = 0);
};
struct LCASimplePredictionModel : LCAPredictionModelBase {
  explicit LCASimplePredictionModel(const Taxonomy* tax, int device = 0);
};
struct MeganLCAPredictionModel : LCAPredictionModelBase {
  MeganLCAPredictionModel(const Taxonomy* tax, bool ignore_unclassified = false, float toppercent = 1.0f, float minscore = 0.0f,
                          int minsupport = 1, double maxevalue = 1.7976931348623157e308, int device = 0);
};
struct NBestLCAPredictionModel : LCAPredictionModelBase {
  explicit NBestLCAPredictionModel(const Taxonomy* tax, int n = 1, int device = 0);
};

}  // namespace taxator
