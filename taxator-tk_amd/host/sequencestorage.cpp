// taxator-tk_amd/host/sequencestorage.cpp -- FASTA reader for the resident stores (see sequencestorage.hh).
#include "sequencestorage.hh"

#include <cstdio>
#include <cstring>

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
    index_.emplace(id, (uint32_t)ids_.size());  // first occurrence wins, like the .fai name lookup
    ids_.push_back(id);
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

}  // namespace taxator
