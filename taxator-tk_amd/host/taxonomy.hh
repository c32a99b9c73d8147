// taxator-tk_amd/host/taxonomy.hh -- NCBI dump -> pruned, flattened taxonomy (host side of SURVEY 8a row a8).
// Mirrors loadTaxonomyFromEnvironment / parseNCBIFlatFiles (core/src/ncbidata.cpp:17-209), TaxonTree::deleteUnmarkedNodes
// (core/src/taxontree.cpp:55-70, depths recomputed :169-178) and TaxonomyInterface::getNode (core/src/taxonomyinterface.cpp:8-12).
// The product of loading is what the engine takes (rpa_taxonomy): parent / nested-set left,right / depth arrays with the
// root at index 0, plus the taxid strings for GFF3 output.  Plain (uncompressed) nodes.dmp / names.dmp are read.
#pragma once
#include <cstdint>
#include <string>
#include <unordered_map>
#include <vector>

namespace taxator {

typedef std::string TaxonID;  // core/src/types.hh:41
extern const std::vector<std::string> default_ranks;  // core/src/constants.hh:33
extern const char* const ENVVAR_TAXONOMY_NCBI;        // core/src/constants.hh:37

class Taxonomy {
 public:
  std::vector<uint32_t> parent, left, right;
  std::vector<uint8_t> depth;
  std::vector<TaxonID> taxid;
  std::vector<std::string> name;
  std::vector<std::string> rank;   // TaxonAnnotation::rank of every kept node (nodes.dmp column 3)
  std::string version;             // first line of $TAXATORTK_TAXONOMY_NCBI/version.txt if present (ncbidata.cpp:201-207)
  // Taxon::is_unclassified (core/src/taxontree.hh:68): set while the FULL tree is built, for every node below a taxon
  // whose scientific name contains "unclassified" and for that taxon itself (core/src/ncbidata.cpp:119-126)
  std::vector<uint8_t> is_unclassified;

  uint32_t getRoot() const { return 0; }
  bool isParentOf(uint32_t a, uint32_t b) const { return right[a] > left[b] && left[a] < left[b]; }  // taxonomyinterface.cpp:52-55
  uint8_t getMaxDepth() const { uint8_t m = 0; for (uint8_t d : depth) m = d > m ? d : m; return m; }
  uint32_t getNode(const TaxonID& id) const;  // throws TaxonNotFound (also for taxa pruned away, taxontree.cpp:59-64)
  size_t size() const { return parent.size(); }

 private:
  friend Taxonomy* parseNCBIFlatFiles(const std::string&, const std::string&, const std::vector<std::string>*, bool);
  std::unordered_map<TaxonID, uint32_t> index_;
};

// ranks_to_mark == nullptr keeps every node; delete_unmarked mirrors `-d` (taxator.cpp:285,340)
Taxonomy* parseNCBIFlatFiles(const std::string& nodes_filename, const std::string& names_filename,
                             const std::vector<std::string>* ranks_to_mark, bool delete_unmarked);
Taxonomy* loadTaxonomyFromEnvironment(const std::vector<std::string>* ranks_to_mark, bool delete_unmarked = true);

}  // namespace taxator
