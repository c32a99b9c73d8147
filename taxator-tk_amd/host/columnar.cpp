// taxator-tk_amd/host/columnar.cpp -- see columnar.hh.
#include "columnar.hh"
#include "hostcpus.hh"

#include <algorithm>
#include <chrono>
#include <cerrno>
#include <cstdlib>
#include <cstring>
#include <new>
#include <thread>

#include <emmintrin.h>  // SSE2: part of the x86-64 baseline
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

namespace taxator {

void* raw_column_alloc(size_t bytes) {
  if (bytes >= ((size_t)4 << 20)) {
    void* p = nullptr;
    const size_t rounded = (bytes + ((size_t)2 << 20) - 1) & ~(((size_t)2 << 20) - 1);
    if (posix_memalign(&p, (size_t)2 << 20, rounded) == 0) {
      static const bool no_thp = getenv("TAXATOR_NO_THP") != nullptr;  // (A/B)
      if (!no_thp) (void)madvise(p, rounded, MADV_HUGEPAGE);
      return p;
    }
  }
  void* p = malloc(bytes ? bytes : 1);
  if (!p) throw std::bad_alloc();
  return p;
}
void raw_column_free(void* p) noexcept { free(p); }

void RecordColumns::resize(size_t n) {
  qstart.resize(n); qstop.resize(n); qlen.resize(n); rstart.resize(n); rstop.resize(n); identities.resize(n); alnlen.resize(n);
  ref_seq.resize(n); node.resize(n); score.resize(n); filtered.resize(n);
}

void RecordColumns::move_range(size_t dst, size_t src, size_t n) {
  if (!n || dst == src) return;
  auto mv = [&](auto& v) { memmove(v.data() + dst, v.data() + src, n * sizeof(v[0])); };
  mv(qstart); mv(qstop); mv(qlen); mv(rstart); mv(rstop); mv(identities); mv(alnlen); mv(ref_seq); mv(node); mv(score); mv(filtered);
}

void ColumnarBatch::make_room(size_t n_records, size_t n_sets) {
  const bool grow = n_records > rec.qstart.capacity() || n_records > rec.filtered.capacity() || n_records > rec.score.capacity() ||
                    n_sets + 1 > set_offsets.capacity() || n_sets > query_seq.capacity() || n_sets > query_len.capacity();
  if (!grow) return;
  if (before_regrow) before_regrow();
  // a quarter more than asked for: blocks of one stream differ by a few per cent, and every growth is a re-registration
  const size_t nr = n_records + n_records / 4, ns = n_sets + n_sets / 4;
  rec.qstart.reserve(nr); rec.qstop.reserve(nr); rec.qlen.reserve(nr); rec.rstart.reserve(nr); rec.rstop.reserve(nr);
  rec.identities.reserve(nr); rec.alnlen.reserve(nr); rec.ref_seq.reserve(nr); rec.node.reserve(nr); rec.score.reserve(nr);
  rec.filtered.reserve(nr);
  set_offsets.reserve(ns + 1); query_seq.reserve(ns); query_len.reserve(ns); qid_off.reserve(ns); qid_len.reserve(ns);
}

void ColumnarBatch::clear() {
  set_offsets.assign(1, 0);
  query_seq.clear(); query_len.clear(); qid_off.clear(); qid_len.clear(); qid_arena.clear();
  rec.resize(0);
}

// ---- field conversions: accept exactly what the strict conversions of alignmentrecord.cpp accept, produce the same values ----

// boost::lexical_cast<large_unsigned_int> as mirrored by to_u32: decimal digits only, whole field, value < 2^32
static inline bool field_u32(const char* p, const char* e, uint32_t& out) {
  if (p == e) return false;
  if (*p == 0) { out = 0; return true; }  // (strtoull of "" with *end == 0)
  uint64_t v = 0;
  for (; p != e; ++p) {
    if (*p == 0) break;  // the strict parser converts c_str(): an embedded NUL ends the field
    const unsigned d = (unsigned char)*p - '0';
    if (d > 9) return false;
    v = v * 10 + d;
    if (v > 0xFFFFFFFFull) v = 0x100000000ull;  // stays out of range, cannot wrap
  }
  if (v > 0xFFFFFFFFull) return false;
  out = (uint32_t)v;
  return true;
}

static const double kPow10[9] = {1e0, 1e1, 1e2, 1e3, 1e4, 1e5, 1e6, 1e7, 1e8};

// strtof over the whole field.  Fast path: [sign] digits [. digits] [e [sign] digits] with at most 7 significant digits
// and a decimal exponent within +-8 -- then mantissa and power of ten are exact doubles, the product is exact and the
// quotient is within 2^-53 of the true value, which cannot straddle a float rounding boundary for such short decimals,
// so one rounding to float gives the correctly rounded result strtof returns.  Anything else goes through strtof.
static bool field_f32(const char* p, const char* e, float& out) {
  if (p == e) return false;
  const char* s = p;
  bool neg = false;
  if (*s == '-' || *s == '+') { neg = *s == '-'; ++s; }
  uint64_t mant = 0;
  int sig = 0, frac = 0, ndig = 0;
  bool fast = true;
  for (; s != e && (unsigned)(*s - '0') <= 9u; ++s, ++ndig) {
    if (mant || *s != '0') { mant = mant * 10 + (unsigned)(*s - '0'); if (++sig > 7) { fast = false; break; } }
  }
  if (fast && s != e && *s == '.') {
    ++s;
    for (; s != e && (unsigned)(*s - '0') <= 9u; ++s, ++ndig) {
      ++frac;
      if (mant || *s != '0') { mant = mant * 10 + (unsigned)(*s - '0'); if (++sig > 7) { fast = false; break; } }
    }
  }
  if (fast && ndig == 0) fast = false;
  int ex = 0;
  if (fast && s != e && (*s == 'e' || *s == 'E')) {
    ++s;
    bool eneg = false;
    if (s != e && (*s == '-' || *s == '+')) { eneg = *s == '-'; ++s; }
    int ed = 0;
    for (; s != e && (unsigned)(*s - '0') <= 9u; ++s, ++ed) { ex = ex * 10 + (*s - '0'); if (ex > 10000) ex = 10000; }
    if (ed == 0) fast = false;
    if (eneg) ex = -ex;
  }
  if (fast && s == e) {
    const int e10 = ex - frac;
    if (mant == 0) { out = neg ? -0.0f : 0.0f; return true; }
    if (e10 >= -8 && e10 <= 8) {
      const double d = e10 < 0 ? (double)mant / kPow10[-e10] : (double)mant * kPow10[e10];
      out = (float)(neg ? -d : d);
      return true;
    }
  }
  // general case: what to_f32 does
  if (isspace((unsigned char)*p)) return false;
  char tmp[64];
  const size_t n = (size_t)(e - p);
  std::string big;
  const char* z;
  if (n < sizeof(tmp)) { memcpy(tmp, p, n); tmp[n] = 0; z = tmp; }
  else { big.assign(p, n); z = big.c_str(); }
  char* end = nullptr;
  out = strtof(z, &end);
  return *end == 0;  // as to_f32: an embedded NUL ends the conversion like the end of the field
}

// the E-value is only validated (strtod over the whole field); the RPA path never reads it
static bool field_f64_valid(const char* p, const char* e) {
  if (p == e) return false;
  const char* s = p;
  if (*s == '-' || *s == '+') ++s;
  int nd = 0;
  for (; s != e && (unsigned)(*s - '0') <= 9u; ++s) ++nd;
  if (s != e && *s == '.') { ++s; for (; s != e && (unsigned)(*s - '0') <= 9u; ++s) ++nd; }
  if (nd > 0) {
    if (s == e) return true;
    if (*s == 'e' || *s == 'E') {
      ++s;
      if (s != e && (*s == '-' || *s == '+')) ++s;
      int ed = 0;
      for (; s != e && (unsigned)(*s - '0') <= 9u; ++s) ++ed;
      if (ed > 0 && s == e) return true;
    }
  }
  if (isspace((unsigned char)*p)) return false;
  std::string z(p, (size_t)(e - p));
  char* end = nullptr;
  (void)strtod(z.c_str(), &end);
  return *end == 0;
}

// ---- reader ----
static void count_lines_fwd(const char* begin, const char* end, size_t& lines, size_t& data_lines);

ReaderTables::ReaderTables(const StrIDConverter& acc2taxid, const Taxonomy* tax, const SequenceStore* db, const SequenceStore* query) {
  // reference identifier -> (taxonomy node, index in the reference store); identifiers with a broken link are left out
  // of the table, so that their lines fall through to the strict parser, which raises the matching exception
  const std::vector<std::string>& ids = db->ids();
  ref_map.reserve(ids.size());
  for (size_t i = 0; i < ids.size(); ++i) {
    if (ref_map.find(ids[i].data(), ids[i].size()) != FlatStringMap::npos) continue;  // first occurrence wins (getIndex)
    uint32_t node;
    try { node = tax->getNode(acc2taxid[ids[i]]); } catch (Exception&) { continue; }
    ref_map.insert(ids[i], (uint32_t)ref_node.size());
    ref_node.push_back(node);
    ref_seqidx.push_back((uint32_t)i);
  }
  const std::vector<std::string>& qids = query->ids();
  query_map.reserve(qids.size());
  for (size_t i = 0; i < qids.size(); ++i) query_map.insert(qids[i], (uint32_t)i);
}

ColumnarReader::ColumnarReader(FILE* in, const StrIDConverter& acc2taxid, const Taxonomy* tax, const SequenceStore* db,
                               const SequenceStore* query, bool split_alignments, unsigned threads, size_t block_bytes, bool alignments_sorted,
                               std::shared_ptr<const ReaderTables> tables)
    : in_(in), legacy_(acc2taxid, tax, db), query_(query), split_(split_alignments), sorted_(alignments_sorted), threads_(threads), block_bytes_(block_bytes),
      tab_(tables ? std::move(tables) : std::make_shared<const ReaderTables>(acc2taxid, tax, db, query)) {
  if (threads_ == 0) threads_ = usable_cpus();
  if (block_bytes_ < 4096) block_bytes_ = 4096;
}

// ---- slices of a mapped file ----
SliceCutter::SliceCutter(FILE* in, size_t block_bytes) : block_(block_bytes < 4096 ? 4096 : block_bytes) {
  struct stat st;
  const int fd = fileno(in);
  const off_t at = fd >= 0 ? lseek(fd, 0, SEEK_CUR) : (off_t)-1;
  if (!getenv("TAXATOR_NO_MMAP") && fd >= 0 && at >= 0 && ftell(in) == (long)at && fstat(fd, &st) == 0 && S_ISREG(st.st_mode) && st.st_size > at) {
    void* m = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
    if (m != MAP_FAILED) { map_ = static_cast<const char*>(m); map_size_ = (size_t)st.st_size; start_ = pos_ = (size_t)at; }
  }
}
SliceCutter::~SliceCutter() { if (map_) munmap(const_cast<char*>(map_), map_size_); }

// query identifier of the line at s (a leading '*' masks the record and is not part of the name, alignmentrecord.hh:98-104)
static inline void line_qid(const char* s, const char* end, const char*& b, size_t& n) {
  if (s < end && *s == '*') ++s;
  const char* e = s;
  while (e < end && *e != '\t' && *e != '\n') ++e;
  b = s; n = (size_t)(e - s);
}
const char* SliceCutter::group_boundary(const char* lo, const char* q, const char* hi) {
  if (q >= hi) return hi;
  if (q <= lo) return lo;
  // s = the first line start at or behind q
  const char* s = q;
  if (s[-1] != '\n') { s = (const char*)memchr(s, '\n', (size_t)(hi - s)); if (!s) return hi; ++s; }
  // the data line in front of it (comment lines, fileparser.hh:61-67, belong to no group)
  const char* pb = nullptr; size_t pn = 0;
  for (const char* e = s; e > lo;) {            // e = one past the '\n' of the line we look at
    const char* b = e - 1;                      // its '\n'
    while (b > lo && b[-1] != '\n') --b;        // its first byte
    if (!(*b == '#' && b != e - 1)) { line_qid(b, e - 1, pb, pn); break; }
    e = b;
  }
  if (!pb) return s;  // nothing but comments in front: any line start will do
  for (const char* l = s; l < hi;) {
    const char* nl = (const char*)memchr(l, '\n', (size_t)(hi - l));
    const char* le = nl ? nl : hi;
    if (!(*l == '#' && l != le)) {
      const char* b; size_t n;
      line_qid(l, le, b, n);
      if (n != pn || memcmp(b, pb, n)) return l;
      pb = b; pn = n;
    }
    l = nl ? nl + 1 : hi;
  }
  return hi;
}
bool SliceCutter::next(const char*& begin, const char*& end, uint64_t& index) {
  std::lock_guard<std::mutex> lk(mu_);
  index = index_;
  if (!map_ || pos_ >= map_size_) return false;
  const char* lo = map_ + pos_;
  const char* hi = map_ + map_size_;
  const char* cut = (size_t)(hi - lo) > block_ ? group_boundary(lo, lo + block_, hi) : hi;
  begin = lo; end = cut;
  pos_ = (size_t)(cut - map_);
  ++index_;
  return true;
}
void SliceCutter::close() {
  std::lock_guard<std::mutex> lk(mu_);
  pos_ = map_size_;
}
uint64_t SliceCutter::lines_before(const char* p) const {
  size_t lines = 0, data = 0;
  const char* b = map_ + start_;
  if (p > b) count_lines_fwd(b, p, lines, data);
  return lines;
}

struct ColumnarReader::Chunk {
  const char* begin;
  const char* end;
  size_t lines = 0, data_lines = 0;  // counted in pass 1
  size_t first_record = 0;
  uint64_t first_line = 0;
  bool bad = false;
  uint64_t bad_line = 0;             // 1-based global line number of the first rejected line
  const char* bad_begin = nullptr;
  const char* bad_end = nullptr;
  size_t bad_record = 0;             // records before that line
  std::vector<size_t> new_group;     // pass 2: records of this chunk (but its first) whose query identifier differs from the record before
};

static inline const char* line_end(const char* p, const char* end) {
  const char* nl = (const char*)memchr(p, '\n', (size_t)(end - p));
  return nl ? nl : end;
}

// One sweep over a line, 16 bytes at a time: its end (the next '\n', or `limit`) and the positions of its first 11 tabs.
// (A memchr call per field costs more than the conversion of the field: lines are ~50 bytes, fields ~4.)
static inline const char* scan_line(const char* s, const char* limit, const char** tab, int& ntab) {
  const __m128i vt = _mm_set1_epi8('\t'), vn = _mm_set1_epi8('\n');
  ntab = 0;
  const char* p = s;
  while (p + 16 <= limit) {
    const __m128i v = _mm_loadu_si128(reinterpret_cast<const __m128i*>(p));
    unsigned mt = (unsigned)_mm_movemask_epi8(_mm_cmpeq_epi8(v, vt));
    const unsigned mn = (unsigned)_mm_movemask_epi8(_mm_cmpeq_epi8(v, vn));
    if (mn) {
      const unsigned first = (unsigned)__builtin_ctz(mn);
      mt &= (1u << first) - 1u;
      while (mt && ntab < 11) { tab[ntab++] = p + __builtin_ctz(mt); mt &= mt - 1u; }
      return p + first;
    }
    while (mt && ntab < 11) { tab[ntab++] = p + __builtin_ctz(mt); mt &= mt - 1u; }
    p += 16;
  }
  for (; p < limit; ++p) {
    if (*p == '\n') return p;
    if (*p == '\t' && ntab < 11) tab[ntab++] = p;
  }
  return limit;
}

// Lines of [begin, end) and how many of them do not start with '#' (an empty line is a data line, fileparser.hh:61-67).
// A line starts at `begin` and behind every '\n' that is not the last byte.
static inline void count_lines(const char* begin, const char* end, size_t& lines, size_t& data_lines) {
  lines = 0; data_lines = 0;
  if (begin >= end) return;
  size_t nl = *begin == '\n' ? 1 : 0, comment = *begin == '#' ? 1 : 0;
  const __m128i vn = _mm_set1_epi8('\n'), vh = _mm_set1_epi8('#'), zero = _mm_setzero_si128();
  const char* p = begin + 1;  // every byte from here on has a byte in front of it
  while (p + 16 <= end) {
    __m128i acc_n = zero, acc_c = zero;  // per byte lane: matches so far (at most 255 before they are summed)
    int it = 0;
    for (; it < 255 && p + 16 <= end; ++it, p += 16) {
      const __m128i v = _mm_loadu_si128(reinterpret_cast<const __m128i*>(p));
      const __m128i before = _mm_loadu_si128(reinterpret_cast<const __m128i*>(p - 1));
      acc_n = _mm_sub_epi8(acc_n, _mm_cmpeq_epi8(v, vn));
      acc_c = _mm_sub_epi8(acc_c, _mm_and_si128(_mm_cmpeq_epi8(before, vn), _mm_cmpeq_epi8(v, vh)));
    }
    const __m128i sn = _mm_sad_epu8(acc_n, zero), sc = _mm_sad_epu8(acc_c, zero);
    nl += (size_t)_mm_cvtsi128_si64(sn) + (size_t)_mm_cvtsi128_si64(_mm_srli_si128(sn, 8));
    comment += (size_t)_mm_cvtsi128_si64(sc) + (size_t)_mm_cvtsi128_si64(_mm_srli_si128(sc, 8));
  }
  for (; p < end; ++p) {
    if (*p == '\n') ++nl;
    else if (*p == '#' && p[-1] == '\n') ++comment;
  }
  lines = 1 + nl - (end[-1] == '\n' ? 1 : 0);
  data_lines = lines - comment;
}

static void count_lines_fwd(const char* begin, const char* end, size_t& lines, size_t& data_lines) { count_lines(begin, end, lines, data_lines); }

void ColumnarReader::raise_failure() {
  if (!fail_query_.empty()) throw SequenceNotFound("seqid " + fail_query_);
  AlignmentRecordTaxonomy rec;
  try { legacy_.parse(fail_line_, rec); }
  catch (Exception& e) { throw ParsingError(std::string(e.what()) + " [line " + std::to_string(fail_line_no_) + "]"); }
  throw GeneralError("alignment line rejected by the columnar reader but accepted by the record parser [line " +
                     std::to_string(fail_line_no_) + "]");
}

// The field checks of parse_block's pass 2 on ONE data line (s up to its line break or `limit`), into scratch values: what the
// team that parses the slice BEHIND this line will decide about it.  parse_slice's look-ahead must decide the same way, or the
// two readers would differ by one record set in front of a failure.
bool ColumnarReader::fast_path_accepts(const char* s, const char* limit) const {
  const char* tab[11];
  int ntab;
  const char* e = scan_line(s, limit, tab, ntab);
  if ((e - s) <= 1) return false;
  const char* f = s;
  if (*f == '*') ++f;
  if (ntab < 10) return false;
  const char* fb[11];
  const char* fe[11];
  fb[0] = f; fe[0] = tab[0];
  for (int k = 1; k < 10; ++k) { fb[k] = tab[k - 1] + 1; fe[k] = tab[k]; }
  fb[10] = tab[9] + 1; fe[10] = ntab > 10 ? tab[10] : e;
  uint32_t qstart, qstop, u;
  float sc;
  return field_u32(fb[1], fe[1], qstart) && field_u32(fb[2], fe[2], qstop) && qstart <= qstop && field_u32(fb[3], fe[3], u) &&
         field_u32(fb[5], fe[5], u) && field_u32(fb[6], fe[6], u) && field_f32(fb[7], fe[7], sc) && field_f64_valid(fb[8], fe[8]) &&
         field_u32(fb[9], fe[9], u) && field_u32(fb[10], fe[10], u) &&
         tab_->ref_map.find(fb[4], (size_t)(fe[4] - fb[4])) != FlatStringMap::npos;
}

void ColumnarReader::parse_slice(const SliceCutter& cutter, const char* begin, const char* end, ColumnarBatch& out, std::exception_ptr& err) {
  carry_ = 0; carry_qid_.clear(); failed_ = false; fail_query_.clear(); line_base_ = 0; eof_ = false;
  err = nullptr;
  // The reference's generator closes a group by READING the record behind it (alignmentrecord.hh:464-479): when that line
  // does not parse, the exception leaves with the group still open -- its record set is never predicted.  A slice ends
  // where the next group begins, so the line to look at is the first data line behind the slice.
  const char* file_end = cutter.file_end();
  std::exception_ptr next_bad;
  for (const char* l = end; l < file_end;) {
    const char* nl = (const char*)memchr(l, '\n', (size_t)(file_end - l));
    const char* le = nl ? nl : file_end;
    if (*l == '#' && l != le) { l = nl ? nl + 1 : file_end; continue; }  // comment lines are skipped (fileparser.hh:61-67)
    // decided the way the team behind this slice will decide it (parse_block's field checks), worded the way raise_failure words it
    if (!fast_path_accepts(l, file_end)) {
      const std::string where = " [line " + std::to_string(cutter.lines_before(l) + 1) + "]";
      AlignmentRecordTaxonomy rec;
      try {
        legacy_.parse(std::string(l, le), rec);
        next_bad = std::make_exception_ptr(GeneralError("alignment line rejected by the columnar reader but accepted by the record parser" + where));
      } catch (Exception& e) { next_bad = std::make_exception_ptr(ParsingError(std::string(e.what()) + where)); }
      catch (...) { next_bad = std::current_exception(); }
    }
    break;
  }
  parse_block(begin, end, !next_bad, out);  // (not closed: the slice's last group stays behind as an open group and is dropped)
  if (failed_ || !fail_query_.empty()) {
    fail_line_no_ += cutter.lines_before(begin);  // (counted only now: a failure is the end of the run)
    try { raise_failure(); } catch (...) { err = std::current_exception(); }
    failed_ = false; fail_query_.clear();
  } else if (next_bad) err = next_bad;
  carry_ = 0; carry_qid_.clear();
}

void ColumnarReader::parse_block(const char* begin, const char* end, bool eof, ColumnarBatch& out) {
  typedef std::chrono::steady_clock Clock;
  static const bool trace = getenv("TAXATOR_READER_TRACE") != nullptr;
  const Clock::time_point t0 = Clock::now();
  // ---- cut the block into per-thread chunks at line boundaries ----
  const size_t total = (size_t)(end - begin);
  const unsigned T = (unsigned)std::max<size_t>(1, std::min<size_t>(threads_, total / 65536 + 1));
  std::vector<Chunk> chunks(T);
  const char* p = begin;
  for (unsigned t = 0; t < T; ++t) {
    chunks[t].begin = p;
    const char* q = t + 1 == T ? end : begin + total * (t + 1) / T;
    if (q < p) q = p;
    if (q < end) { q = line_end(q, end); if (q < end) ++q; }
    chunks[t].end = q;
    p = q;
  }
  auto for_chunks = [&](auto&& fn) {
    if (T == 1) { fn(chunks[0]); return; }
    std::vector<std::thread> th;
    for (unsigned t = 1; t < T; ++t) th.emplace_back([&, t] { fn(chunks[t]); });
    fn(chunks[0]);
    for (auto& x : th) x.join();
  };
  // ---- pass 1: count lines and data lines ----
  for_chunks([](Chunk& c) { count_lines(c.begin, c.end, c.lines, c.data_lines); });
  const Clock::time_point t1 = Clock::now();
  size_t nrec = carry_;
  uint64_t line_no = line_base_;
  for (Chunk& c : chunks) { c.first_record = nrec; c.first_line = line_no; nrec += c.data_lines; line_no += c.lines; }
  cols_.resize(nrec);
  qid_ptr_.resize(nrec);
  qid_len_.resize(nrec);
  const Clock::time_point t2 = Clock::now();
  // ---- pass 2: parse in place ----
  for_chunks([this](Chunk& c) {
    size_t r = c.first_record;
    uint64_t ln = c.first_line;
    const char* prev_qid = nullptr;  // query identifier of the record before (this chunk's records only)
    uint32_t prev_len = 0;
    for (const char* s = c.begin; s < c.end;) {
      const char* tab[11];
      int ntab;
      const char* e = scan_line(s, c.end, tab, ntab);
      ++ln;
      const char* next = e + 1;
      if (*s == '#' && s != e) { s = next; continue; }
      bool ok = (e - s) > 1;
      const char* f = s;
      uint8_t filt = 0;
      if (ok && *f == '*') { filt = 1; ++f; }
      const char* fb[11];
      const char* fe[11];
      if (ok) {
        if (ntab < 10) ok = false;
        else {
          fb[0] = f; fe[0] = tab[0];
          for (int k = 1; k < 10; ++k) { fb[k] = tab[k - 1] + 1; fe[k] = tab[k]; }
          fb[10] = tab[9] + 1; fe[10] = ntab > 10 ? tab[10] : e;
        }
      }
      uint32_t refv = FlatStringMap::npos;
      ok = ok && field_u32(fb[1], fe[1], cols_.qstart[r]) && field_u32(fb[2], fe[2], cols_.qstop[r]) && cols_.qstart[r] <= cols_.qstop[r] &&
           field_u32(fb[3], fe[3], cols_.qlen[r]) && field_u32(fb[5], fe[5], cols_.rstart[r]) && field_u32(fb[6], fe[6], cols_.rstop[r]) &&
           field_f32(fb[7], fe[7], cols_.score[r]) && field_f64_valid(fb[8], fe[8]) && field_u32(fb[9], fe[9], cols_.identities[r]) &&
           field_u32(fb[10], fe[10], cols_.alnlen[r]) &&
           (refv = tab_->ref_map.find(fb[4], (size_t)(fe[4] - fb[4]))) != FlatStringMap::npos;
      if (!ok) {
        c.bad = true; c.bad_line = ln; c.bad_begin = s; c.bad_end = e; c.bad_record = r;
        return;
      }
      cols_.filtered[r] = filt;
      cols_.node[r] = tab_->ref_node[refv];
      cols_.ref_seq[r] = tab_->ref_seqidx[refv];
      qid_ptr_[r] = fb[0];
      qid_len_[r] = (uint32_t)(fe[0] - fb[0]);
      if (prev_qid && (qid_len_[r] != prev_len || memcmp(fb[0], prev_qid, prev_len))) c.new_group.push_back(r);
      prev_qid = fb[0]; prev_len = qid_len_[r];
      ++r;
      s = next;
    }
  });
  const Clock::time_point t3 = Clock::now();
  // ---- the first rejected line ends the usable part of the stream ----
  size_t usable = nrec;
  for (const Chunk& c : chunks) {
    if (!c.bad) continue;
    usable = c.bad_record;
    failed_ = true;
    fail_line_.assign(c.bad_begin, c.bad_end);
    fail_line_no_ = c.bad_line;
    break;
  }
  for (size_t r = 0; r < carry_; ++r) { qid_ptr_[r] = carry_qid_.data(); qid_len_[r] = (uint32_t)carry_qid_.size(); }
  // ---- groups of one query identifier; the last one stays open unless the stream ended ----
  out.clear();
  gstart_.clear();
  size_t i = 0;
  const bool closed = eof && !failed_;
  // (the chunks noted where the identifier changes inside them; what is left are the seams: the carried records against
  // the first parsed one, and the last record of a chunk against the first of the next)
  auto same_qid = [&](size_t a, size_t b) {
    return qid_len_[a] == qid_len_[b] && (qid_ptr_[a] == qid_ptr_[b] || !memcmp(qid_ptr_[a], qid_ptr_[b], qid_len_[a]));
  };
  if (usable) gstart_.push_back(0);
  size_t seam = carry_;  // first record behind a seam (0: no record in front of it)
  for (const Chunk& c : chunks) {
    if (c.first_record >= usable) break;
    const size_t first = c.first_record, stop = std::min(usable, first + c.data_lines);
    if (first < stop && seam == first && first > 0 && !same_qid(first - 1, first)) gstart_.push_back(first);
    for (size_t r : c.new_group) {
      if (r >= stop) break;
      gstart_.push_back(r);
    }
    seam = stop;
  }
  if (!closed && !gstart_.empty()) { i = gstart_.back(); gstart_.pop_back(); }  // open group: more of it may follow (or, after a bad
  else i = usable;                                                              // line, it is lost as in the reference, whose
  gstart_.push_back(i);  // records [0, i) belong to complete groups            // generator throws while still reading this group)
  const Clock::time_point t4 = Clock::now();
  emit_groups(out);
  if (!fail_query_.empty()) { carry_ = 0; return; }  // unknown query identifier: everything after it is dropped
  if (failed_) { carry_ = 0; return; }
  // carry the open group to the front of the columns
  const size_t n_open = usable - i;
  if (n_open) {
    std::string keep(qid_ptr_[i], qid_len_[i]);
    carry_qid_.swap(keep);
    cols_.move_range(0, i, n_open);
  }
  carry_ = n_open;
  line_base_ = line_no;
  if (trace) {
    auto ms = [](Clock::time_point a, Clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    fprintf(stderr, "reader block: %zu bytes, %zu records, %u threads: count %.1f ms, allocate %.1f ms, parse %.1f ms, group %.1f ms, emit %.1f ms\n",
            total, nrec, T, ms(t0, t1), ms(t1, t2), ms(t2, t3), ms(t3, t4), ms(t4, Clock::now()));
  }
}

// Sort every complete group, cut it into record sets and write its records to their final place in `out`.  Groups are
// independent and a group's records land at the positions they came from (only their order inside the group changes),
// so ranges of groups go to different threads; the per-set columns are stitched together afterwards.
void ColumnarReader::emit_groups(ColumnarBatch& out) {
  const size_t G = gstart_.size() - 1;
  RecordColumns& o = out.rec;
  out.make_room(gstart_[G], out.n_sets());  // (the per-set arrays get their room below, once the sets are counted)
  o.resize(gstart_[G]);
  struct Part {
    std::vector<uint32_t> set_len, set_qlen, group_sets, group_qseq;
    size_t unknown = (size_t)-1;  // first group of the range whose query identifier is not in the store
  };
  const unsigned T = (unsigned)std::max<size_t>(1, std::min<size_t>(threads_, G / 2048 + 1));
  std::vector<Part> parts(T);
  auto work = [&](unsigned t) {
    Part& part = parts[t];
    std::vector<uint32_t> order;
    auto before = [&](uint32_t a, uint32_t b) {
      if (cols_.qstart[a] != cols_.qstart[b]) return cols_.qstart[a] < cols_.qstart[b];
      return cols_.qstop[a] < cols_.qstop[b];
    };
    // Records of groups that keep their input order are copied column by column in one piece per run of such groups;
    // a group that changes order is gathered column by column through its permutation.
    size_t run_lo = gstart_[G * t / T];  // first record of the current run of groups in input order
    auto flush_run = [&](size_t hi) {
      if (hi <= run_lo) return;
      const size_t n = hi - run_lo;
      auto cp = [&](auto& dst, const auto& src) { memcpy(dst.data() + run_lo, src.data() + run_lo, n * sizeof(dst[0])); };
      cp(o.qstart, cols_.qstart); cp(o.qstop, cols_.qstop); cp(o.qlen, cols_.qlen); cp(o.rstart, cols_.rstart); cp(o.rstop, cols_.rstop);
      cp(o.identities, cols_.identities); cp(o.alnlen, cols_.alnlen); cp(o.ref_seq, cols_.ref_seq); cp(o.node, cols_.node);
      cp(o.score, cols_.score); cp(o.filtered, cols_.filtered);
    };
    for (size_t g = G * t / T; g < G * (t + 1) / T; ++g) {
      const size_t lo = gstart_[g], n = gstart_[g + 1] - lo;
      const uint32_t qseq = tab_->query_map.find(qid_ptr_[lo], qid_len_[lo]);
      if (qseq == FlatStringMap::npos) { flush_run(lo); part.unknown = g; return; }  // SequenceStore::getIndex in RPAPredictionModel::predict's staging
      bool reorder = false;
      if (split_ && !sorted_) {  // stable by (qstart, qstop), ties keep input order (alignmentrecord.hh:476-481); -o 1: the input is
                                 // taken as sorted and cut in input order (RecordSetGeneratorSorted, :569-631)
        for (size_t k = 1; k < n && !reorder; ++k) reorder = before((uint32_t)(lo + k), (uint32_t)(lo + k - 1));
        if (reorder) {
          order.resize(n);
          for (size_t k = 0; k < n; ++k) order[k] = (uint32_t)(lo + k);
          if (n <= 64) {  // groups are short
            for (size_t k = 1; k < n; ++k) {
              const uint32_t x = order[k];
              size_t m = k;
              while (m > 0 && before(x, order[m - 1])) { order[m] = order[m - 1]; --m; }
              order[m] = x;
            }
          } else std::stable_sort(order.begin(), order.end(), before);
        }
      }
      if (reorder) {
        flush_run(lo);
        run_lo = lo + n;
        auto gather = [&](auto& dst, const auto& src) {
          for (size_t m = 0; m < n; ++m) dst[lo + m] = src[order[m]];
        };
        gather(o.qstart, cols_.qstart); gather(o.qstop, cols_.qstop); gather(o.qlen, cols_.qlen); gather(o.rstart, cols_.rstart);
        gather(o.rstop, cols_.rstop); gather(o.identities, cols_.identities); gather(o.alnlen, cols_.alnlen); gather(o.ref_seq, cols_.ref_seq);
        gather(o.node, cols_.node); gather(o.score, cols_.score); gather(o.filtered, cols_.filtered);
      }
      // (a group in input order has not been copied yet: its records are read where they were parsed)
      const RecordColumns& c = reorder ? o : cols_;
      uint32_t n_sets = 0;
      for (size_t k = 0; k < n;) {  // one record set: [k, e)
        size_t e = k + 1;
        if (split_) {
          uint32_t rstop = c.qstop[lo + k];
          while (e < n && c.qstart[lo + e] <= rstop) { rstop = std::max(rstop, c.qstop[lo + e]); ++e; }  // :485-504
        } else e = n;  // :543-566
        part.set_len.push_back((uint32_t)(e - k));
        part.set_qlen.push_back(c.qlen[lo + k]);  // getQueryLength() of the set's first record
        ++n_sets;
        k = e;
      }
      part.group_sets.push_back(n_sets);
      part.group_qseq.push_back(qseq);
    }
    flush_run(gstart_[G * (t + 1) / T]);
  };
  if (T == 1) work(0);
  else {
    std::vector<std::thread> th;
    for (unsigned t = 1; t < T; ++t) th.emplace_back(work, t);
    work(0);
    for (auto& x : th) x.join();
  }
  {
    size_t total_sets = 0;
    for (const Part& part : parts) total_sets += part.set_len.size();
    out.make_room(o.size(), out.n_sets() + total_sets);
  }
  for (unsigned t = 0; t < T; ++t) {
    const Part& part = parts[t];
    size_t g = G * t / T, si = 0;
    for (size_t k = 0; k < part.group_sets.size(); ++k, ++g) {
      const size_t lo = gstart_[g];
      for (uint32_t c = 0; c < part.group_sets[k]; ++c, ++si) {
        out.query_seq.push_back(part.group_qseq[k]);
        out.query_len.push_back(part.set_qlen[si]);
        out.qid_off.push_back((uint32_t)out.qid_arena.size());
        out.qid_len.push_back(qid_len_[lo]);
        out.qid_arena.insert(out.qid_arena.end(), qid_ptr_[lo], qid_ptr_[lo] + qid_len_[lo]);
        out.set_offsets.push_back(out.set_offsets.back() + part.set_len[si]);
      }
    }
    if (part.unknown != (size_t)-1) {
      failed_ = true;
      fail_query_.assign(qid_ptr_[gstart_[part.unknown]], qid_len_[gstart_[part.unknown]]);
      o.resize(gstart_[part.unknown]);
      return;
    }
  }
}

ColumnarReader::~ColumnarReader() {
  if (map_) munmap(const_cast<char*>(map_), map_size_);
  if (!io_thread_.joinable()) return;
  if (eof_) { io_thread_.join(); return; }
  // abandoned mid-stream (a failure upstream of the end): the thread may sit in a blocking read; let it go
  io_->stop = true;
  io_->free_q.push(nullptr);
  io_thread_.detach();
}

bool ColumnarReader::next(ColumnarBatch& out) {
  out.clear();
  if (!map_tried_) {  // first call: a regular file with nothing buffered yet can be mapped from the current offset on
    map_tried_ = true;
    struct stat st;
    const int fd = fileno(in_);
    const off_t at = fd >= 0 ? lseek(fd, 0, SEEK_CUR) : (off_t)-1;
    if (!getenv("TAXATOR_NO_MMAP") && fd >= 0 && at >= 0 && ftell(in_) == (long)at && fstat(fd, &st) == 0 && S_ISREG(st.st_mode) && st.st_size > at) {
      void* m = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
      if (m != MAP_FAILED) {
        map_ = static_cast<const char*>(m);
        map_size_ = (size_t)st.st_size;
        map_pos_ = (size_t)at;
        (void)madvise(m, map_size_, MADV_SEQUENTIAL);
      }
    }
  }
  if (map_) {
    for (;;) {
      if (failed_) raise_failure();
      if (eof_) return false;
      const char* begin = map_ + map_pos_;
      const char* file_end = map_ + map_size_;
      size_t want = block_bytes_;
      const char* usable;
      for (;;) {  // up to the last line break inside the block; a line longer than the block extends it
        const char* end = (size_t)(file_end - begin) > want ? begin + want : file_end;
        eof_ = end == file_end;
        usable = end;
        if (!eof_) while (usable > begin && usable[-1] != '\n') --usable;
        if (usable > begin || eof_) break;
        want += block_bytes_;
      }
      parse_block(begin, usable, eof_, out);
      map_pos_ = (size_t)(usable - map_);
      if (out.n_sets() > 0) return true;
    }
  }
  if (!io_) {  // first call: start reading ahead
    io_ = std::make_shared<Io>();
    io_->in = in_;
    io_->block = block_bytes_;
    for (IoBuf& b : io_->bufs) io_->free_q.push(&b);
    std::shared_ptr<Io> io = io_;
    io_thread_ = std::thread([io] {
      for (;;) {
        IoBuf* b = io->free_q.pop();
        if (!b || io->stop) return;
        if (b->cap < kHeadroom + io->block) { b->cap = kHeadroom + io->block; b->mem.reset(new char[b->cap]); }
        b->n = fread(b->mem.get() + kHeadroom, 1, io->block, io->in);
        b->eof = b->n < io->block;
        const bool last = b->eof;
        io->full_q.push(b);
        if (last) return;
      }
    });
  }
  for (;;) {
    if (failed_) raise_failure();
    if (eof_) return false;
    IoBuf* b = io_->full_q.pop();
    eof_ = b->eof;
    // the block with the previous tail in front of it
    char* begin;
    char* end;
    if (tail_.size() <= kHeadroom) {
      begin = b->mem.get() + kHeadroom - tail_.size();
      memcpy(begin, tail_.data(), tail_.size());
      end = b->mem.get() + kHeadroom + b->n;
    } else {
      joined_.resize(tail_.size() + b->n);
      memcpy(joined_.data(), tail_.data(), tail_.size());
      memcpy(joined_.data() + tail_.size(), b->mem.get() + kHeadroom, b->n);
      begin = joined_.data();
      end = begin + joined_.size();
    }
    // parse everything up to the last line break (everything, at the end of the stream)
    char* usable = end;
    if (!eof_) while (usable > begin && usable[-1] != '\n') --usable;
    if (usable > begin || eof_) parse_block(begin, usable, eof_, out);
    tail_.assign(usable, (size_t)(end - usable));  // (qid pointers of the open group were copied out by parse_block)
    if (!eof_) io_->free_q.push(b);
    if (out.n_sets() > 0) return true;
    // nothing complete yet (one group or one line spanning blocks), a failure to raise, or the end of the stream
  }
}

}  // namespace taxator
