// taxator-tk_amd/host/sequencestorage.cpp -- FASTA reader for the resident stores (see sequencestorage.hh).
#include "sequencestorage.hh"

#include <emmintrin.h>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <mutex>
#include <thread>

#include "hostcpus.hh"

namespace taxator {

// The whole FASTA, by all host threads (round 4: one thread loaded config 2's 200 MB of queries, a million records, in
// 0.45 s -- more than the pipeline then needs for the alignments themselves).  The file is mapped and cut into chunks at
// line starts; pass 1 counts, per chunk, the header lines and the residues (every byte of a non-header line that is not a
// blank; bytes in front of the file's first header belong to no record); prefix sums give every chunk its first record and
// its place in the letters; pass 2 copies whole lines (memchr finds the line ends; the usual 60-80 column FASTA has no blanks
// inside lines) and writes identifiers and offsets where they belong.  The identifier -> index map is built on the first
// getIndex(): the columnar reader has a table of its own and never asks.
namespace {
inline bool is_blank(char c) { return c == '\r' || c == ' ' || c == '\t'; }
struct FastaChunk {
  const char* begin; const char* end;
  size_t records = 0, residues = 0;
  size_t first_record = 0, first_residue = 0;
};
// calls header(s, e) / residues(s, e) per line of [begin, end); lines are '\n'-separated, the last may lack its '\n'
template <class H, class R>
inline void for_fasta_lines(const char* begin, const char* end, H&& header, R&& residues) {
  for (const char* s = begin; s < end;) {
    const char* nl = (const char*)memchr(s, '\n', (size_t)(end - s));
    const char* e = nl ? nl : end;
    if (s != e) { if (*s == '>') header(s + 1, e); else residues(s, e); }
    s = nl ? nl + 1 : end;
  }
}
}  // namespace

SequenceStore::SequenceStore(const std::string& fasta_filename) {
  const int fd = open(fasta_filename.c_str(), O_RDONLY);
  if (fd < 0) throw FileNotFound(fasta_filename);
  struct stat sb;
  if (fstat(fd, &sb) != 0) { close(fd); throw FileNotFound(fasta_filename); }
  const size_t len = (size_t)sb.st_size;
  offsets_.push_back(0);
  if (len == 0) { close(fd); return; }
  void* m = mmap(nullptr, len, PROT_READ, MAP_PRIVATE, fd, 0);
  close(fd);
  if (m == MAP_FAILED) throw FileNotFound(fasta_filename + " (mmap failed)");
  (void)madvise(m, len, MADV_SEQUENTIAL);
  const char* base = (const char*)m;
  const char* file_end = base + len;
  size_t per_chunk = (size_t)4 << 20;  // (TAXATOR_FASTA_CHUNK=bytes: tests cut small files into many chunks)
  if (const char* v = getenv("TAXATOR_FASTA_CHUNK")) per_chunk = (size_t)std::max(1ll, atoll(v));
  const unsigned T = (unsigned)std::max<size_t>(1, std::min<size_t>(per_chunk < ((size_t)4 << 20) ? 64 : usable_cpus(), len / per_chunk + 1));
  std::vector<FastaChunk> chunks(T);
  const char* p = base;
  for (unsigned t = 0; t < T; ++t) {
    chunks[t].begin = p;
    const char* q = t + 1 == T ? file_end : base + len * (t + 1) / T;
    if (q < p) q = p;
    if (q < file_end) { const char* nl = (const char*)memchr(q, '\n', (size_t)(file_end - q)); q = nl ? nl + 1 : file_end; }
    chunks[t].end = q;
    p = q;
  }
  auto run = [&](auto&& fn) {
    std::vector<std::thread> th;
    for (unsigned t = 1; t < T; ++t) th.emplace_back([&, t] { fn(chunks[t], t); });
    fn(chunks[0], 0u);
    for (auto& x : th) x.join();
  };
  // ---- pass 1: records and residues per chunk (residues in front of the chunk's first header apart: they belong to a record
  // only if an earlier chunk opened one) ----
  std::vector<size_t> lead(T, 0);
  run([&](FastaChunk& c, unsigned t) {
    size_t before = 0, after = 0;
    bool own_header = false;
    for_fasta_lines(c.begin, c.end,
                    [&](const char*, const char*) { ++c.records; own_header = true; },
                    [&](const char* s, const char* e) {
                      size_t n = (size_t)(e - s);
                      if (memchr(s, '\r', n) || memchr(s, ' ', n) || memchr(s, '\t', n)) { size_t k = 0; for (; s < e; ++s) k += is_blank(*s) ? 0 : 1; n = k; }
                      (own_header ? after : before) += n;
                    });
    lead[t] = before; c.residues = after;
  });
  size_t n_rec = 0, n_res = 0;
  bool seen_header = false;
  for (unsigned t = 0; t < T; ++t) {
    FastaChunk& c = chunks[t];
    if (seen_header) c.residues += lead[t];  // (else: bytes in front of the file's first header, dropped)
    seen_header = seen_header || c.records != 0;
    c.first_record = n_rec; c.first_residue = n_res;
    n_rec += c.records; n_res += c.residues;
  }
  letters_.resize(n_res);
  offsets_.assign(n_rec + 1, 0);
  ids_.resize(n_rec);
  // ---- pass 2: identifiers, offsets, letters ----
  std::vector<uint8_t> headerless(T, 0);
  for (unsigned t = 0, before = 1; t < T; ++t) { headerless[t] = (uint8_t)before; if (chunks[t].records) before = 0; }
  run([&](FastaChunk& c, unsigned t) {
    size_t rec = c.first_record, pos = c.first_residue;
    bool open_record = !headerless[t];
    uint8_t* out = letters_.data();
    for_fasta_lines(c.begin, c.end,
                    [&](const char* s, const char* e) {  // s points behind '>': the identifier ends at the first whitespace (faidx.h:590)
                      offsets_[rec] = pos;
                      std::string& id = ids_[rec++];
                      for (; s < e; ++s) {
                        if (*s == '\r') continue;
                        if (isspace((unsigned char)*s)) break;
                        id.push_back(*s);
                      }
                      open_record = true;
                    },
                    [&](const char* s, const char* e) {
                      if (!open_record) return;
                      const size_t n = (size_t)(e - s);
                      if (!memchr(s, '\r', n) && !memchr(s, ' ', n) && !memchr(s, '\t', n)) { memcpy(out + pos, s, n); pos += n; return; }
                      for (; s < e; ++s)
                        if (!is_blank(*s)) out[pos++] = (uint8_t)*s;  // readRegion skips whitespace (faidx.h:340-347)
                    });
  });
  offsets_[n_rec] = n_res;
  munmap(m, len);
  index_built_ = false;  // (built by the first getIndex)
}

void SequenceStore::build_index() const {
  std::lock_guard<std::mutex> lk(index_mu_);
  if (index_built_) return;
  index_.reserve(ids_.size());
  for (size_t i = 0; i < ids_.size(); ++i) index_.emplace(ids_[i], (uint32_t)i);  // first occurrence wins, like the .fai name lookup
  index_built_ = true;
}

SequenceStore::~SequenceStore() {
  if (map_) munmap(const_cast<char*>(map_), map_len_);
}

void SequenceStore::add_id(const std::string& id) {
  index_.emplace(id, (uint32_t)ids_.size());  // first occurrence wins, like the .fai name lookup
  ids_.push_back(id);
}

// Indexed store: map the FASTA, take the per-sequence geometry from the .fai (samtools format, five tab-separated
// columns: name, length, offset of the first residue, residues per line, bytes per line; core/src/faidx.h:438-476).
SequenceStore::SequenceStore(const std::string& fasta_filename, const std::string& fai_filename) : indexed_(true) {
  const int fd = open(fasta_filename.c_str(), O_RDONLY);
  if (fd < 0) throw FileNotFound(fasta_filename);
  struct stat sb;
  if (fstat(fd, &sb) != 0) { close(fd); throw FileNotFound(fasta_filename); }
  map_len_ = (size_t)sb.st_size;
  if (map_len_) {
    void* m = mmap(nullptr, map_len_, PROT_READ, MAP_PRIVATE, fd, 0);
    if (m == MAP_FAILED) { close(fd); throw FileNotFound(fasta_filename + " (mmap failed)"); }
    map_ = (const char*)m;
    madvise(m, map_len_, MADV_RANDOM);  // regions are read where the alignment stream points, not front to back
  }
  close(fd);
  // The index file, by all host threads (round 5: `std::getline` + `sscanf` + a hash-map insertion per line took 6 s for the ten
  // million queries of a 10 M-record-set run, more than the stream itself): mapped, cut into chunks at line starts, lines counted
  // per chunk, then parsed in place into their slots.  The name -> index map is built by the first getIndex() (the columnar reader
  // has tables of its own and never asks), as for the in-memory store.
  const int ffd = fai_filename.empty() ? -1 : open(fai_filename.c_str(), O_RDONLY);
  struct stat fsb;
  if (ffd >= 0 && fstat(ffd, &fsb) == 0) {
    const size_t flen = (size_t)fsb.st_size;
    const char* fbase = nullptr;
    if (flen) {
      void* fm = mmap(nullptr, flen, PROT_READ, MAP_PRIVATE, ffd, 0);
      if (fm == MAP_FAILED) { close(ffd); throw FileNotFound(fai_filename + " (mmap failed)"); }
      fbase = (const char*)fm;
    }
    close(ffd);
    const char* fend = fbase + flen;
    const unsigned T = (unsigned)std::max<size_t>(1, std::min<size_t>(usable_cpus(), flen / ((size_t)1 << 20) + 1));
    struct Piece { const char* b; const char* e; size_t lines = 0, first = 0; std::string err; };
    std::vector<Piece> pieces(T);
    const char* p = fbase;
    for (unsigned t = 0; t < T; ++t) {
      pieces[t].b = p;
      const char* q = t + 1 == T ? fend : fbase + flen * (t + 1) / T;
      if (q < p) q = p;
      if (q < fend) { const char* nl = (const char*)memchr(q, '\n', (size_t)(fend - q)); q = nl ? nl + 1 : fend; }
      pieces[t].e = q;
      p = q;
    }
    auto for_pieces = [&](auto&& fn) {
      std::vector<std::thread> th;
      for (unsigned t = 1; t < T; ++t) th.emplace_back([&, t] { fn(pieces[t]); });
      fn(pieces[0]);
      for (auto& x : th) x.join();
    };
    auto for_lines = [](const char* b, const char* e, auto&& fn) {  // non-empty lines of [b, e)
      for (const char* s = b; s < e;) {
        const char* nl = (const char*)memchr(s, '\n', (size_t)(e - s));
        const char* le = nl ? nl : e;
        const char* trimmed = le;
        if (trimmed > s && trimmed[-1] == '\r') --trimmed;
        if (trimmed != s) fn(s, trimmed);
        s = nl ? nl + 1 : e;
      }
    };
    for_pieces([&](Piece& pc) { for_lines(pc.b, pc.e, [&](const char*, const char*) { ++pc.lines; }); });
    size_t n_lines = 0;
    for (Piece& pc : pieces) { pc.first = n_lines; n_lines += pc.lines; }
    ids_.resize(n_lines);
    entries_.resize(n_lines);
    index_built_ = false;  // (built by the first getIndex)
    const size_t map_len = map_len_;
    for_pieces([&](Piece& pc) {
      size_t k = pc.first;
      for_lines(pc.b, pc.e, [&](const char* s, const char* e) {
        if (!pc.err.empty()) return;
        const char* t1 = (const char*)memchr(s, '\t', (size_t)(e - s));
        if (!t1) { pc.err = "fasta index line without columns: " + std::string(s, e); return; }
        unsigned long long v[4] = {0, 0, 0, 0};
        const char* q = t1 + 1;
        bool ok = true;
        for (int f = 0; f < 4 && ok; ++f) {  // four unsigned decimal columns separated by tabs (anything behind the fourth is ignored, as sscanf did)
          if (q >= e || *q < '0' || *q > '9') { ok = false; break; }
          unsigned long long x = 0;
          while (q < e && *q >= '0' && *q <= '9') { x = x * 10 + (unsigned long long)(*q - '0'); ++q; }
          v[f] = x;
          if (f < 3) { if (q < e && *q == '\t') ++q; else ok = false; }
        }
        if (!ok) { pc.err = "fasta index line is not name, length, offset, line bases, line width: " + std::string(s, e); return; }
        FaiEntry en{v[0], v[1], (uint32_t)v[2], (uint32_t)v[3]};
        if (en.offset > map_len || (en.length && (en.linebases == 0 || en.linewidth < en.linebases))) { pc.err = "fasta index does not fit the fasta file: " + std::string(s, e); return; }
        ids_[k].assign(s, t1);
        entries_[k] = en;
        ++k;
      });
    });
    if (fbase) munmap((void*)fbase, flen);
    for (const Piece& pc : pieces)
      if (!pc.err.empty()) throw ParsingError(pc.err);
  } else {
    if (ffd >= 0) close(ffd);
    build_index_by_scan();
  }
}

// No .fai next to the FASTA: one pass over the mapped file finds what `samtools faidx` would have written.  A sequence
// whose lines are not all of one length (except the last) is marked irregular and fetched by scanning.
void SequenceStore::build_index_by_scan() {
  const char* p = map_;
  const char* end = map_ + map_len_;
  bool empty_line_pending = false;  // the current record has a line without residues that more residues may still follow
  while (p < end) {
    const char* nl = (const char*)memchr(p, '\n', (size_t)(end - p));
    const char* le = nl ? nl : end;
    if (*p == '>') {
      std::string id;
      for (const char* s = p + 1; s < le && !isspace((unsigned char)*s); ++s) id.push_back(*s);
      add_id(id);
      FaiEntry e{0, (uint64_t)((nl ? nl + 1 : end) - map_), 0, 0};
      entries_.push_back(e);
      empty_line_pending = false;
    } else if (!entries_.empty()) {
      FaiEntry& e = entries_.back();
      size_t bases = 0;
      bool blank_inside = false;
      for (const char* s = p; s < le; ++s) { if (*s == '\r' || *s == ' ' || *s == '\t') blank_inside = blank_inside || *s != '\r'; else ++bases; }
      const uint32_t width = (uint32_t)((nl ? nl + 1 : end) - p);
      if (e.length == 0 && e.linewidth == 0) { e.linebases = (uint32_t)bases; e.linewidth = width; if (blank_inside || bases == 0) e.linebases = 0; }
      else if (e.linebases && (e.length % e.linebases != 0 || bases > e.linebases || blank_inside)) e.linebases = 0;  // a short line that is not the last, or a longer one
      // an empty line in the MIDDLE of a record (length so far a multiple of the line length, so the tests above let it pass)
      // shifts every later line by its bytes: such a record is fetched by scanning, as readRegion skips all whitespace
      if (bases && empty_line_pending) e.linebases = 0;
      if (!bases && e.length) empty_line_pending = true;
      e.length += bases;
      if (e.linewidth == 0) e.linewidth = width ? width : 1;
    }
    p = nl ? nl + 1 : end;
  }
}

void SequenceStore::fetch(uint32_t i, uint8_t* dst) const {
  const FaiEntry& e = entries_[i];
  uint64_t remaining = e.length;
  const char* p = map_ + e.offset;
  const char* end = map_ + map_len_;
  if (e.linebases) {
    while (remaining) {
      const uint64_t take = remaining < e.linebases ? remaining : e.linebases;
      if (p + take > end) throw ParsingError("fasta index points beyond the end of the fasta file (sequence " + ids_[i] + ")");
      if (memchr(p, ' ', take) || memchr(p, '\t', take) || memchr(p, '\n', take) || memchr(p, '\r', take))
        break;  // blanks or line ends inside what the index calls a line: readRegion skips them -> scan the rest
      memcpy(dst, p, take);
      dst += take; p += e.linewidth; remaining -= take;
    }
    if (!remaining) return;
  }
  for (; remaining && p < end; ++p) {  // irregular lines: every non-blank byte up to `length` of them (faidx.h:340-347)
    const char c = *p;
    if (c == '\n' || c == '\r' || c == ' ' || c == '\t') continue;
    *dst++ = (uint8_t)c;
    --remaining;
  }
  if (remaining) throw ParsingError("fasta file ends inside sequence " + ids_[i]);
}

// `n` bytes from s to dst; false when one of them is a blank (space, tab, line break -- every byte <= 0x20; bytes >= 0x80 count as
// blanks too: the slow path below then sorts them out).  ONE pass, 16 bytes at a time: the copy and the check together run at
// memory speed (round 5: four memchr calls and a memcpy per 80-byte line were 0.4 GB/s per thread, 6 GB/s on sixteen).
static inline bool copy_line_checked(uint8_t* dst, const char* s, uint64_t n) {
  const __m128i limit = _mm_set1_epi8(0x21);
  __m128i bad = _mm_setzero_si128();
  uint64_t k = 0;
  for (; k + 16 <= n; k += 16) {
    const __m128i v = _mm_loadu_si128(reinterpret_cast<const __m128i*>(s + k));
    bad = _mm_or_si128(bad, _mm_cmplt_epi8(v, limit));  // signed: 0x80.. are "less" as well
    _mm_storeu_si128(reinterpret_cast<__m128i*>(dst + k), v);
  }
  bool tail_bad = false;
  for (; k < n; ++k) { const char c = s[k]; tail_bad = tail_bad || (signed char)c < 0x21; dst[k] = (uint8_t)c; }
  return !tail_bad && _mm_movemask_epi8(bad) == 0;
}

void SequenceStore::fetch_region(uint32_t i, uint64_t first, uint64_t n, uint8_t* dst) const {
  const FaiEntry& e = entries_[i];
  if (first + n > e.length) throw SequenceRangeError("region beyond the end of sequence " + ids_[i]);
  const char* end = map_ + map_len_;
  uint64_t remaining = n;
  if (e.linebases) {
    uint64_t col = first % e.linebases;  // residues to skip on the first line
    const char* p = map_ + e.offset + (first / e.linebases) * (uint64_t)e.linewidth;
    while (remaining) {
      const uint64_t take = std::min<uint64_t>(remaining, e.linebases - col);
      if (p + col + take > end) throw ParsingError("fasta index points beyond the end of the fasta file (sequence " + ids_[i] + ")");
      // (the residues in FRONT of the region on its first line are looked at too, as before: a blank there shifts the columns)
      if ((col && (memchr(p, ' ', col) || memchr(p, '\t', col) || memchr(p, '\n', col) || memchr(p, '\r', col))) || !copy_line_checked(dst, p + col, take))
        break;  // this line is not what the index says: count residues from the start of the record instead (below)
      dst += take; remaining -= take; first += take; col = 0; p += e.linewidth;
    }
    if (!remaining) return;
  }
  // irregular lines (or a line that broke the index's promise): skip `first` non-blank bytes from the start of the
  // record, then take the rest -- the reference's readRegion does the same from the line the index points at
  uint64_t skip = first;
  const char* p = map_ + e.offset;
  if (e.linebases) {  // regular up to the line that failed: jump to its start
    const uint64_t line = first / e.linebases;
    p += line * (uint64_t)e.linewidth;
    skip = first - line * e.linebases;
  }
  for (; remaining && p < end; ++p) {
    const char c = *p;
    if (c == '\n' || c == '\r' || c == ' ' || c == '\t') continue;
    if (skip) { --skip; continue; }
    *dst++ = (uint8_t)c;
    --remaining;
  }
  if (remaining) throw ParsingError("fasta file ends inside sequence " + ids_[i]);
}

}  // namespace taxator
