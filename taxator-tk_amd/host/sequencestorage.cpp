// taxator-tk_amd/host/sequencestorage.cpp -- FASTA reader for the resident stores (see sequencestorage.hh).
#include "sequencestorage.hh"

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>

namespace taxator {

// Line-wise over 4 MiB reads: a header line opens a record, every other line is appended minus blanks.  memchr finds
// the line ends and whole lines are copied at once (the usual 60-80 column FASTA has no blanks inside lines), so loading
// runs at a few GB/s instead of one branch per byte.
SequenceStore::SequenceStore(const std::string& fasta_filename) {
  FILE* f = fopen(fasta_filename.c_str(), "rb");
  if (!f) throw FileNotFound(fasta_filename);
  if (fseek(f, 0, SEEK_END) == 0) {
    const long sz = ftell(f);
    if (sz > 0) letters_.reserve((size_t)sz);
    fseek(f, 0, SEEK_SET);
  }
  offsets_.push_back(0);
  bool have_record = false;
  auto add_header = [&](const char* s, const char* e) {  // s points behind '>'
    if (have_record) offsets_.push_back(letters_.size());
    have_record = true;
    std::string id;
    for (; s < e; ++s) {
      if (*s == '\r') continue;
      if (isspace((unsigned char)*s)) break;
      id.push_back(*s);
    }
    add_id(id);
  };
  auto add_line = [&](const char* s, const char* e) {
    if (s == e) return;
    if (*s == '>') { add_header(s + 1, e); return; }
    if (!have_record) return;
    const size_t n = (size_t)(e - s);
    if (!memchr(s, '\r', n) && !memchr(s, ' ', n) && !memchr(s, '\t', n)) { letters_.insert(letters_.end(), s, e); return; }
    for (; s < e; ++s)
      if (*s != '\r' && *s != ' ' && *s != '\t') letters_.push_back((uint8_t)*s);  // readRegion skips whitespace (faidx.h:340-347)
  };
  std::vector<char> buf((size_t)4 << 20);
  size_t fill = 0;
  for (;;) {
    if (fill == buf.size()) buf.resize(buf.size() * 2);  // a single line longer than the buffer
    const size_t got = fread(buf.data() + fill, 1, buf.size() - fill, f);
    fill += got;
    const bool eof = got == 0;
    const char* s = buf.data();
    const char* end = buf.data() + fill;
    while (s < end) {
      const char* nl = (const char*)memchr(s, '\n', (size_t)(end - s));
      if (!nl) {
        if (!eof) break;
        nl = end;
      }
      add_line(s, nl);
      s = nl < end ? nl + 1 : end;
    }
    fill = (size_t)(end - s);
    memmove(buf.data(), s, fill);
    if (eof) break;
  }
  fclose(f);
  if (have_record) offsets_.push_back(letters_.size());
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
  std::ifstream fai(fai_filename.c_str());
  if (!fai_filename.empty() && fai) {
    std::string line;
    while (std::getline(fai, line)) {
      if (line.empty()) continue;
      const size_t t1 = line.find('\t');
      if (t1 == std::string::npos) throw ParsingError("fasta index line without columns: " + line);
      FaiEntry e{0, 0, 0, 0};
      unsigned long long len = 0, off = 0, lb = 0, lw = 0;
      if (sscanf(line.c_str() + t1 + 1, "%llu\t%llu\t%llu\t%llu", &len, &off, &lb, &lw) != 4)
        throw ParsingError("fasta index line is not name, length, offset, line bases, line width: " + line);
      e.length = len; e.offset = off; e.linebases = (uint32_t)lb; e.linewidth = (uint32_t)lw;
      if (e.offset > map_len_ || (e.length && (e.linebases == 0 || e.linewidth < e.linebases)))
        throw ParsingError("fasta index does not fit the fasta file: " + line);
      add_id(line.substr(0, t1));
      entries_.push_back(e);
    }
  } else {
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
      const char* s = p + col;
      if (memchr(p, ' ', col + take) || memchr(p, '\t', col + take) || memchr(p, '\n', col + take) || memchr(p, '\r', col + take))
        break;  // this line is not what the index says: count residues from the start of the record instead (below)
      memcpy(dst, s, take);
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
