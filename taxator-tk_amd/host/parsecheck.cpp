// taxator-tk_amd/host/parsecheck.cpp -- TEST UTILITY: dumps the record sets the two input paths build from one TSV stream.
//   parsecheck <mapping.tax> <query.fasta> <ref.fasta> objects|columnar <split 0|1> [threads] [block_bytes]
// "objects" = AlignmentRecordFactory + RecordSetGenerator (the mirror of the reference's object model), "columnar" =
// ColumnarReader.  Both print one line per record set and per record and, on failure, "ERROR <text>" (exit code 1), so
// that tests can require identical bytes from the two (no GPU involved).
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <memory>
#include <vector>

#include "columnar.hh"

using namespace taxator;

static void print_set(const char* qid, size_t qid_len, uint32_t qlen, uint32_t qseq, size_t n) {
  fwrite("SET\t", 1, 4, stdout);
  fwrite(qid, 1, qid_len, stdout);
  printf("\t%u\t%u\t%zu\n", qlen, qseq, n);
}
static void print_rec(bool filt, uint32_t qs, uint32_t qe, uint32_t rs, uint32_t re, float score, uint32_t ident, uint32_t alnlen,
                      uint32_t ref_seq, uint32_t node) {
  uint32_t bits;
  memcpy(&bits, &score, 4);
  printf("%d\t%u\t%u\t%u\t%u\t%08x\t%u\t%u\t%u\t%u\n", filt ? 1 : 0, qs, qe, rs, re, bits, ident, alnlen, ref_seq, node);
}

// parsecheck --fasta <file.fasta> <file.fai | -> : the indexed store (memory-mapped FASTA + .fai, or an index built by a
// scan when "-" is given) against the in-memory store: same identifiers, same lengths, same residues.
static int fasta_check(const char* fasta, const char* fai) {
  try {
    SequenceStore mem(fasta);
    SequenceStore idx(fasta, strcmp(fai, "-") ? fai : "");
    if (mem.size() != idx.size()) { printf("MISMATCH sequences %u vs %u\n", mem.size(), idx.size()); return 1; }
    uint64_t total = 0;
    std::vector<uint8_t> buf;
    for (uint32_t i = 0; i < mem.size(); ++i) {
      if (mem.ids()[i] != idx.ids()[i] || idx.getIndex(idx.ids()[i]) != mem.getIndex(mem.ids()[i])) { printf("MISMATCH id %u\n", i); return 1; }
      if (mem.length(i) != idx.length(i)) { printf("MISMATCH length of %s: %llu vs %llu\n", mem.ids()[i].c_str(), (unsigned long long)mem.length(i), (unsigned long long)idx.length(i)); return 1; }
      buf.assign((size_t)idx.length(i) + 1, 0xEE);
      idx.fetch(i, buf.data());
      if (buf[(size_t)idx.length(i)] != 0xEE || memcmp(buf.data(), mem.letters().data() + mem.offsets()[i], (size_t)mem.length(i)) != 0) {
        printf("MISMATCH residues of %s\n", mem.ids()[i].c_str());
        return 1;
      }
      // regions (readRegion, faidx.h:322-350): the ends, single residues, line boundaries and a few pseudo-random cuts
      const uint64_t len = idx.length(i);
      uint64_t x = 88172645463325252ull + i;
      for (int k = 0; k < 24 && len; ++k) {
        x ^= x << 13; x ^= x >> 7; x ^= x << 17;
        uint64_t first = k == 0 ? 0 : k == 1 ? len - 1 : x % len;
        uint64_t n = k == 0 ? len : k == 1 ? 1 : 1 + (x >> 20) % (len - first);
        if (k >= 2 && k < 8) { first = std::min<uint64_t>(len - 1, (uint64_t)(k - 2) * 30); n = std::min<uint64_t>(len - first, 59 + k); }  // around 60 / 61-column line ends
        buf.assign((size_t)n + 1, 0xEE);
        idx.fetch_region(i, first, n, buf.data());
        if (buf[(size_t)n] != 0xEE || memcmp(buf.data(), mem.letters().data() + mem.offsets()[i] + first, (size_t)n) != 0) {
          printf("MISMATCH region %llu+%llu of %s\n", (unsigned long long)first, (unsigned long long)n, mem.ids()[i].c_str());
          return 1;
        }
      }
      total += mem.length(i);
    }
    printf("OK %u %llu\n", mem.size(), (unsigned long long)total);
    return 0;
  } catch (Exception& e) {
    printf("ERROR %s\n", e.what());
    return 1;
  }
}

// parsecheck --taxonomy DELETE_UNMARKED [rank ...]: the taxonomy the driver would hand to the engine ($TAXATORTK_TAXONOMY_NCBI, the
// ranks of `-r` or the default ones, `-d`), one node per line in index order: taxid, parent's taxid, depth, left, right, rank;
// further arguments after "--" are taxids to look up ("NODE taxid index" or "MISSING taxid": the reference's TaxonNotFound)
static int taxonomy_check(int argc, char** argv) {
  const bool delete_unmarked = atoi(argv[2]) != 0;
  std::vector<std::string> ranks, lookups;
  int i = 3;
  for (; i < argc && strcmp(argv[i], "--"); ++i) ranks.push_back(argv[i]);
  for (++i; i < argc; ++i) lookups.push_back(argv[i]);
  if (ranks.empty()) ranks = default_ranks;  // core/taxator.cpp:324-326
  try {
    std::unique_ptr<Taxonomy> tax(loadTaxonomyFromEnvironment(&ranks, delete_unmarked));
    if (!tax) return 2;
    for (size_t k = 0; k < tax->size(); ++k)
      printf("%s\t%s\t%u\t%u\t%u\t%s\n", tax->taxid[k].c_str(), tax->taxid[tax->parent[k]].c_str(), (unsigned)tax->depth[k], tax->left[k], tax->right[k],
             tax->rank[k].c_str());
    for (const std::string& t : lookups) {
      try { printf("NODE %s %u\n", t.c_str(), tax->getNode(t)); }
      catch (Exception& e) { printf("MISSING %s (%s)\n", t.c_str(), e.what()); }
    }
    return 0;
  } catch (Exception& e) {
    printf("ERROR %s\n", e.what());
    return 1;
  }
}

int main(int argc, char** argv) {
  if (argc == 4 && !strcmp(argv[1], "--fasta")) return fasta_check(argv[2], argv[3]);
  if (argc >= 3 && !strcmp(argv[1], "--taxonomy")) return taxonomy_check(argc, argv);
  if (argc < 6) { fprintf(stderr, "usage: parsecheck mapping query.fasta ref.fasta objects|columnar split [threads] [block_bytes]\n"); return 2; }
  const std::string mode = argv[4];
  const bool split = atoi(argv[5]) != 0;
  const unsigned threads = argc > 6 ? (unsigned)atoi(argv[6]) : 0;
  const size_t block = argc > 7 ? (size_t)atoll(argv[7]) : (size_t)64 << 20;
  std::unique_ptr<Taxonomy> tax(loadTaxonomyFromEnvironment(&default_ranks, true));
  if (!tax) return 2;
  try {
    StrIDConverter acc(argv[1]);
    SequenceStore query(argv[2]), db(argv[3]);
    if (mode == "objects") {
      AlignmentRecordFactory fac(acc, tax.get(), &db);
      RecordSetGenerator gen(std::cin, fac, split, getenv("PARSECHECK_SORTED") != nullptr);
      RecordSetType rset;
      while (gen.notEmpty()) {
        rset.clear();
        gen.getNext(rset);
        const AlignmentRecordTaxonomy& f = rset.front();
        print_set(f.query_identifier.data(), f.query_identifier.size(), f.query_length, query.getIndex(f.query_identifier), rset.size());
        for (const AlignmentRecordTaxonomy& r : rset)
          print_rec(r.blacklist_this, r.query_start, r.query_stop, r.reference_start, r.reference_stop, r.score, r.identities,
                    r.alignment_length, r.reference_seq, r.reference_node);
      }
    } else if (mode == "slices") {  // the reader teams' path: the mapped file cut at query identifier boundaries, slice by slice
      SliceCutter cutter(stdin, block);
      if (!cutter.mapped()) { fprintf(stderr, "parsecheck slices: stdin must be a regular file\n"); return 2; }
      ColumnarReader reader(stdin, acc, tax.get(), &db, &query, split, threads, block, getenv("PARSECHECK_SORTED") != nullptr);
      ColumnarBatch b;
      const char *sb, *se;
      uint64_t index;
      const bool quiet_slices = getenv("PARSECHECK_QUIET") != nullptr;  // timing runs: parse only
      while (cutter.next(sb, se, index)) {
        std::exception_ptr err;
        reader.parse_slice(cutter, sb, se, b, err);
        if (quiet_slices) { if (err) std::rethrow_exception(err); continue; }
        for (size_t s = 0; s < b.n_sets(); ++s) {
          print_set(b.qid_arena.data() + b.qid_off[s], b.qid_len[s], b.query_len[s], b.query_seq[s], b.set_offsets[s + 1] - b.set_offsets[s]);
          for (uint32_t r = b.set_offsets[s]; r < b.set_offsets[s + 1]; ++r)
            print_rec(b.rec.filtered[r], b.rec.qstart[r], b.rec.qstop[r], b.rec.rstart[r], b.rec.rstop[r], b.rec.score[r],
                      b.rec.identities[r], b.rec.alnlen[r], b.rec.ref_seq[r], b.rec.node[r]);
        }
        if (err) std::rethrow_exception(err);
      }
    } else {
      ColumnarReader reader(stdin, acc, tax.get(), &db, &query, split, threads, block, getenv("PARSECHECK_SORTED") != nullptr);
      ColumnarBatch b;
      const bool quiet = getenv("PARSECHECK_QUIET") != nullptr;  // timing runs: parse only
      size_t sets = 0, recs = 0;
      while (reader.next(b)) {
        sets += b.n_sets(); recs += b.n_records();
        if (quiet) continue;
        for (size_t s = 0; s < b.n_sets(); ++s) {
          print_set(b.qid_arena.data() + b.qid_off[s], b.qid_len[s], b.query_len[s], b.query_seq[s], b.set_offsets[s + 1] - b.set_offsets[s]);
          for (uint32_t r = b.set_offsets[s]; r < b.set_offsets[s + 1]; ++r)
            print_rec(b.rec.filtered[r], b.rec.qstart[r], b.rec.qstop[r], b.rec.rstart[r], b.rec.rstop[r], b.rec.score[r],
                      b.rec.identities[r], b.rec.alnlen[r], b.rec.ref_seq[r], b.rec.node[r]);
        }
      }
      if (quiet) fprintf(stderr, "columnar: %zu record sets, %zu records\n", sets, recs);
    }
  } catch (Exception& e) {
    fflush(stdout);
    printf("ERROR %s\n", e.what());
    return 1;
  }
  return 0;
}
