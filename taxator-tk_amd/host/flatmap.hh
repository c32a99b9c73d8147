// taxator-tk_amd/host/flatmap.hh -- identifier -> 32-bit value table for the columnar TSV reader.
// The reference resolves every alignment line's reference identifier through two std::map lookups
// (core/src/accessconv.hh:60-64 seqid -> taxid, core/src/taxonomyinterface.cpp:8-12 taxid -> node) plus the store's
// name lookup (core/src/sequencestorage.hh:348-351).  At tens of millions of lines per second that is one
// open-addressing probe here: keys live back to back in one arena, slots hold (hash, key offset, key length, value).
#pragma once
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

namespace taxator {

inline uint64_t hash_bytes(const char* p, size_t n) {  // FNV-1a, 64 bit, finalised with a multiply-shift mix
  uint64_t h = 1469598103934665603ull;
  for (size_t i = 0; i < n; ++i) { h ^= (uint8_t)p[i]; h *= 1099511628211ull; }
  h ^= h >> 32;
  h *= 0x9E3779B97F4A7C15ull;
  return h ^ (h >> 29);
}

class FlatStringMap {
 public:
  static constexpr uint32_t npos = 0xFFFFFFFFu;
  void reserve(size_t n) { if (slots_.empty()) rehash(n * 2 + 16); }
  // first insertion of a key wins; returns false if the key was already present
  bool insert(const char* key, size_t len, uint32_t value) {
    if ((count_ + 1) * 10 > slots_.size() * 6) rehash(slots_.size() * 2 + 16);
    const uint64_t h = hash_bytes(key, len);
    size_t i = h & mask_;
    while (slots_[i].len != kEmpty) {
      if (slots_[i].hash == h && slots_[i].len == len && !memcmp(arena_.data() + slots_[i].off, key, len)) return false;
      i = (i + 1) & mask_;
    }
    slots_[i] = Slot{h, (uint64_t)arena_.size(), (uint32_t)len, value};
    arena_.insert(arena_.end(), key, key + len);
    ++count_;
    return true;
  }
  bool insert(const std::string& k, uint32_t v) { return insert(k.data(), k.size(), v); }
  uint32_t find(const char* key, size_t len) const {
    if (slots_.empty()) return npos;
    const uint64_t h = hash_bytes(key, len);
    size_t i = h & mask_;
    while (slots_[i].len != kEmpty) {
      if (slots_[i].hash == h && slots_[i].len == len && !memcmp(arena_.data() + slots_[i].off, key, len)) return slots_[i].value;
      i = (i + 1) & mask_;
    }
    return npos;
  }
  size_t size() const { return count_; }

 private:
  static constexpr uint32_t kEmpty = 0xFFFFFFFFu;
  struct Slot { uint64_t hash; uint64_t off; uint32_t len; uint32_t value; };
  void rehash(size_t want) {
    size_t cap = 16;
    while (cap < want) cap <<= 1;
    std::vector<Slot> old;
    old.swap(slots_);
    slots_.assign(cap, Slot{0, 0, kEmpty, 0});
    mask_ = cap - 1;
    for (const Slot& s : old) {
      if (s.len == kEmpty) continue;
      size_t i = s.hash & mask_;
      while (slots_[i].len != kEmpty) i = (i + 1) & mask_;
      slots_[i] = s;
    }
  }
  std::vector<Slot> slots_;
  std::vector<char> arena_;
  size_t mask_ = 0, count_ = 0;
};

}  // namespace taxator
