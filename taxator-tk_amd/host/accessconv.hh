// taxator-tk_amd/host/accessconv.hh -- seqid -> taxid mapping file (`seqid \t taxid`, '#' comments).
// Mirrors AccessIDConverterFlatfileMemory / loadStrIDConverterFromFile (core/src/accessconv.hh:51-110).
#pragma once
#include <fstream>
#include <string>
#include <unordered_map>

#include "exception.hh"
#include "taxonomy.hh"

namespace taxator {

class StrIDConverter {
 public:
  explicit StrIDConverter(const std::string& filename) : filename_(filename) {
    std::ifstream f(filename);
    if (!f) throw FileNotFound(filename);
    std::string line;
    while (std::getline(f, line)) {
      if (!line.empty() && line[0] == '#') continue;  // ignoreLine(), core/src/utils.hh:46-48
      if (line.empty()) continue;
      const size_t tab = line.find('\t');
      if (tab == std::string::npos) throw ParsingError("Could not parse line: " + line);
      size_t e = line.find('\t', tab + 1);
      map_[line.substr(0, tab)] = line.substr(tab + 1, e == std::string::npos ? std::string::npos : e - tab - 1);
    }
  }
  const TaxonID& operator[](const std::string& acc) const {
    auto it = map_.find(acc);
    if (it == map_.end()) throw TaxonMappingNotFound("seqid " + acc + " in " + filename_);  // accessconv.hh:60-64
    return it->second;
  }
 private:
  std::unordered_map<std::string, TaxonID> map_;
  std::string filename_;
};

inline StrIDConverter* loadStrIDConverterFromFile(const std::string& filename, unsigned = 0) { return new StrIDConverter(filename); }

}  // namespace taxator
