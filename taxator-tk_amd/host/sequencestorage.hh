// taxator-tk_amd/host/sequencestorage.hh -- host side of the resident sequence stores.
// Mirrors the role of RandomSeqStoreROInterface / RandomIndexedSeqstoreRO (core/src/sequencestorage.hh:41-52,318-466):
// identifiers are the FASTA header up to the first whitespace (core/src/faidx.h:590), residues are kept as raw letters
// (any case, any byte) because the alphabet mapping of the indexed store (unknown -> N / X) is applied by the engine on
// the device.  Instead of seeking in the file per hit (faidx.h:315-350 under one mutex, sequencestorage.hh:355-358) the
// whole file is read once, handed to rpa_store_load and then lives packed in HBM; getSequence()-style coordinates stay
// in the batch rows.
#pragma once
#include <cstdint>
#include <string>
#include <unordered_map>
#include <vector>

#include "exception.hh"

namespace taxator {

class SequenceStore {
 public:
  explicit SequenceStore(const std::string& fasta_filename);
  uint32_t getIndex(const std::string& id) const {  // SequenceNotFound as in sequencestorage.hh:348-351
    auto it = index_.find(id);
    if (it == index_.end()) throw SequenceNotFound("seqid " + id);
    return it->second;
  }
  uint32_t size() const { return (uint32_t)ids_.size(); }
  uint64_t length(uint32_t i) const { return offsets_[i + 1] - offsets_[i]; }
  const std::vector<uint8_t>& letters() const { return letters_; }
  const std::vector<uint64_t>& offsets() const { return offsets_; }
  const std::vector<std::string>& ids() const { return ids_; }
 private:
  std::vector<uint8_t> letters_;
  std::vector<uint64_t> offsets_;
  std::vector<std::string> ids_;
  std::unordered_map<std::string, uint32_t> index_;
};

}  // namespace taxator
