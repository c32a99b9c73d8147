// taxator-tk_amd/host/sequencestorage.hh -- host side of the resident sequence stores.
// Mirrors the two stores of the reference (core/src/sequencestorage.hh):
//   * RandomInmemorySeqStoreRO (:56-140; taxator without -i / -v, core/taxator.cpp:232-249): the whole FASTA is read
//     once, handed to rpa_store_load and then lives packed in HBM;
//   * RandomIndexedSeqstoreRO (:318-406, AA :410-466; taxator -i / -v): nothing is read up front.  The reference
//     seeks + reads every region it needs through the .fai index under one mutex (core/src/faidx.h:315-350,
//     sequencestorage.hh:355-358).  Here the FASTA is memory-mapped, the .fai gives name / length / offset / line
//     shape per sequence (faidx.h:438-476), and a sequence becomes resident in HBM the first time a block of the
//     alignment stream refers to it (fetch() + rpa_store_append): load time and HBM scale with the sequences touched.
// Identifiers are the header up to the first whitespace (faidx.h:590); residues are kept as raw letters because the
// alphabet mapping of the indexed store (unknown -> N / X) is applied by the engine on the device; getSequence()-style
// coordinates stay in the batch rows.
// Deliberate normalisation: a missing .fai is rebuilt in memory by one scan of the FASTA (the reference throws
// "could not build fasta index" there because it takes the success return of its index builder for a failure,
// sequencestorage.hh:322-326); nothing is written next to the FASTA.
#pragma once
#include <atomic>
#include <cstdint>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

#include "exception.hh"

namespace taxator {

class SequenceStore {
 public:
  explicit SequenceStore(const std::string& fasta_filename);                                  // in memory
  SequenceStore(const std::string& fasta_filename, const std::string& fai_filename);          // through the index
  ~SequenceStore();
  SequenceStore(const SequenceStore&) = delete;
  SequenceStore& operator=(const SequenceStore&) = delete;

  uint32_t getIndex(const std::string& id) const {  // SequenceNotFound as in sequencestorage.hh:348-351
    if (!index_built_) build_index();
    auto it = index_.find(id);
    if (it == index_.end()) throw SequenceNotFound("seqid " + id);
    return it->second;
  }
  uint32_t size() const { return (uint32_t)ids_.size(); }
  uint64_t length(uint32_t i) const { return indexed_ ? entries_[i].length : offsets_[i + 1] - offsets_[i]; }
  const std::vector<std::string>& ids() const { return ids_; }
  bool indexed() const { return indexed_; }
  // in-memory stores only
  const std::vector<uint8_t>& letters() const { return letters_; }
  const std::vector<uint64_t>& offsets() const { return offsets_; }
  // indexed stores only: the length(i) residues of sequence i, blanks and line ends removed (what readRegion returns
  // for the whole sequence, faidx.h:336-348); thread-safe
  void fetch(uint32_t i, uint8_t* dst) const;
  // ... and a region of it: the n residues from residue `first` (0-based) on, i.e. readRegion(first, first + n)
  // (faidx.h:322-350: seek to offset + (first / linebases) * linewidth + first % linebases, then take n non-blank bytes);
  // thread-safe.  The caller keeps first + n <= length(i).
  void fetch_region(uint32_t i, uint64_t first, uint64_t n, uint8_t* dst) const;

 private:
  struct FaiEntry { uint64_t length, offset; uint32_t linebases, linewidth; };  // linebases == 0: irregular lines, scan
  void add_id(const std::string& id);
  void build_index_by_scan();
  bool indexed_ = false;
  std::vector<uint8_t> letters_;
  std::vector<uint64_t> offsets_;
  std::vector<std::string> ids_;
  mutable std::unordered_map<std::string, uint32_t> index_;  // in-memory stores: built by the first getIndex() (build_index)
  mutable std::atomic<bool> index_built_{true};
  mutable std::mutex index_mu_;
  void build_index() const;
  std::vector<FaiEntry> entries_;
  const char* map_ = nullptr;  // the memory-mapped FASTA
  size_t map_len_ = 0;
};

}  // namespace taxator
