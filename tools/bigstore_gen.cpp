// tools/bigstore_gen.cpp -- a reference collection that is NOT a toy, with an alignment stream over it (SURVEY 8f N1:
// "the real end-to-end bottleneck" is the region fetch through the .fai, core/src/sequencestorage.hh:318-406, core/src/faidx.h:315-350).
//
//   bigstore_gen OUT_DIR REF_GBASES N_SETS HITS SPECIES_FILE [SEED]
//
// Writes, in the reference's file formats (doc/fileformats.md):
//   ref.fna + ref.fna.fai   REF_GBASES x 1e9 bases in sequences of 10 kbp ... 10 Mbp (lengths log-uniform), 80 bases per line.  Every
//                           sequence is a chain of 10 kbp blocks, each a copy of a block of a 200 Mbase ancestral pool with 2-15 %
//                           substitutions of its own; one copy in ten is stored reverse-complemented.
//   query.fna + .fai        one ~200 nt query per record set: a stretch of a pool block with 5 % substitutions
//   aln.tsv                 HITS alignment lines per query: the same stretch in HITS of the block's copies -- wherever in the collection
//                           those copies happen to lie, i.e. at random loci of random sequences -- with true identities (no indels:
//                           coordinates are exact), score = 2 x identities - length, reverse-strand copies as rstart > rstop
//   mapping.tax             refN -> a taxid of SPECIES_FILE (one taxid per line), round robin
// Measurement tool (tools/cli_bigstore.py drives it); nothing of the product depends on it.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

static inline uint64_t mix(uint64_t x) {  // splitmix64 finaliser
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}
static const char kLetters[4] = {'A', 'C', 'G', 'T'};
static const uint32_t B = 10000;  // block length

struct Copy { uint32_t seq, block; uint32_t pool; uint32_t thr; uint8_t rc; };  // thr: substitution threshold on 32 bits of hash

static std::vector<uint8_t> g_pool;  // codes 0..3
static uint64_t g_seed;

// base x (0 .. B-1, in the pool block's own orientation) of a copy
static inline uint8_t copy_base(const Copy& c, uint64_t copy_id, uint32_t x) {
  const uint8_t p = g_pool[(size_t)c.pool * B + x];
  const uint64_t h = mix(g_seed ^ (copy_id * 0x100000001B3ull + x));
  return ((uint32_t)h < c.thr) ? (uint8_t)((p + 1 + ((h >> 40) % 3)) & 3) : p;
}

int main(int argc, char** argv) {
  if (argc < 6) { fprintf(stderr, "usage: bigstore_gen OUT_DIR REF_GBASES N_SETS HITS SPECIES_FILE [SEED]\n"); return 2; }
  const std::string out = argv[1];
  const double gbases = atof(argv[2]);
  const uint64_t n_sets = strtoull(argv[3], nullptr, 10);
  const uint32_t hits = (uint32_t)atoi(argv[4]);
  g_seed = argc > 6 ? strtoull(argv[6], nullptr, 10) : 12345;
  std::vector<std::string> species;
  {
    FILE* f = fopen(argv[5], "r");
    if (!f) { perror(argv[5]); return 2; }
    char buf[256];
    while (fgets(buf, sizeof(buf), f)) { std::string s(buf); while (!s.empty() && (s.back() == '\n' || s.back() == '\r')) s.pop_back(); if (!s.empty()) species.push_back(s); }
    fclose(f);
  }
  if (species.empty()) { fprintf(stderr, "no species\n"); return 2; }
  const unsigned T = std::max(1u, std::thread::hardware_concurrency() > 64 ? 32u : std::thread::hardware_concurrency());
  const uint32_t P = 20000;  // pool blocks: 200 Mbases
  g_pool.resize((size_t)P * B);
  {
    std::vector<std::thread> th;
    for (unsigned t = 0; t < T; ++t) th.emplace_back([&, t] {
      for (size_t i = (size_t)P * B * t / T; i < (size_t)P * B * (t + 1) / T; ++i) g_pool[i] = (uint8_t)(mix(g_seed * 31 + i) & 3);
    });
    for (auto& x : th) x.join();
  }
  // ---- sequences: lengths log-uniform in [1, 1000] blocks until the target is reached ----
  const uint64_t target_blocks = (uint64_t)(gbases * 1e9 / B);
  std::vector<uint32_t> seq_blocks;
  std::vector<uint64_t> first_copy;  // index of a sequence's first block copy
  uint64_t total_blocks = 0;
  for (uint64_t j = 0; total_blocks < target_blocks; ++j) {
    const double u = (double)(mix(g_seed * 7 + j) >> 11) / 9007199254740992.0;
    uint32_t nb = (uint32_t)std::max(1.0, std::floor(std::pow(1000.0, u)));
    if (total_blocks + nb > target_blocks) nb = (uint32_t)(target_blocks - total_blocks);
    first_copy.push_back(total_blocks);
    seq_blocks.push_back(nb);
    total_blocks += nb;
  }
  const uint32_t n_seq = (uint32_t)seq_blocks.size();
  std::vector<Copy> copies(total_blocks);
  std::vector<uint32_t> pool_count(P + 1, 0);
  for (uint32_t j = 0; j < n_seq; ++j)
    for (uint32_t k = 0; k < seq_blocks[j]; ++k) {
      const uint64_t id = first_copy[j] + k;
      const uint64_t h = mix(g_seed * 13 + id);
      Copy c;
      c.seq = j; c.block = k; c.pool = (uint32_t)(h % P);
      const double rate = 0.02 + 0.13 * (double)((h >> 32) & 0xFFFF) / 65536.0;
      c.thr = (uint32_t)(rate * 4294967296.0);
      c.rc = ((h >> 48) % 10) == 0;
      copies[id] = c;
      ++pool_count[c.pool + 1];
    }
  for (uint32_t p = 0; p < P; ++p) pool_count[p + 1] += pool_count[p];
  std::vector<uint64_t> by_pool(total_blocks);
  {
    std::vector<uint32_t> fill(pool_count.begin(), pool_count.end() - 1);
    for (uint64_t id = 0; id < total_blocks; ++id) by_pool[fill[copies[id].pool]++] = id;
  }
  fprintf(stderr, "bigstore_gen: %u sequences, %llu blocks of %u bases (%.1f Gbases), %u pool blocks, ~%.0f copies each, %u threads\n", n_seq,
          (unsigned long long)total_blocks, B, total_blocks * (double)B / 1e9, P, (double)total_blocks / P, T);
  // ---- ref.fna + .fai + mapping ----
  {
    FILE* fa = fopen((out + "/ref.fna").c_str(), "wb");
    FILE* fai = fopen((out + "/ref.fna.fai").c_str(), "w");
    FILE* map = fopen((out + "/mapping.tax").c_str(), "w");
    if (!fa || !fai || !map) { perror("open"); return 2; }
    setvbuf(fa, nullptr, _IOFBF, 1 << 24);
    uint64_t off = 0;
    std::vector<char> buf;
    for (uint32_t j = 0; j < n_seq; ++j) {
      const uint64_t len = (uint64_t)seq_blocks[j] * B;
      char head[64];
      const int hn = snprintf(head, sizeof(head), ">ref%u\n", j);
      fwrite(head, 1, (size_t)hn, fa);
      off += (uint64_t)hn;
      fprintf(fai, "ref%u\t%llu\t%llu\t80\t81\n", j, (unsigned long long)len, (unsigned long long)off);
      fprintf(map, "ref%u\t%s\n", j, species[j % species.size()].c_str());
      const uint64_t text = len + len / 80;  // B is a multiple of 80: every line is full, the last one too
      buf.resize((size_t)text);
      const uint32_t nb = seq_blocks[j];
      const unsigned W = std::min<unsigned>(T, nb);
      std::vector<std::thread> th;
      for (unsigned t = 0; t < W; ++t) th.emplace_back([&, t] {
        for (uint32_t k = (uint32_t)((uint64_t)nb * t / W); k < (uint32_t)((uint64_t)nb * (t + 1) / W); ++k) {
          const uint64_t id = first_copy[j] + k;
          const Copy& c = copies[id];
          char* dst = buf.data() + (size_t)k * (B + B / 80);
          for (uint32_t x = 0; x < B; ++x) {
            const uint8_t code = c.rc ? (uint8_t)(3 - copy_base(c, id, B - 1 - x)) : copy_base(c, id, x);
            dst[x + x / 80] = kLetters[code];
            if (x % 80 == 79) dst[x + x / 80 + 1] = '\n';
          }
        }
      });
      for (auto& x : th) x.join();
      fwrite(buf.data(), 1, (size_t)text, fa);
      off += text;
    }
    fclose(fa); fclose(fai); fclose(map);
  }
  fprintf(stderr, "bigstore_gen: reference written\n");
  // ---- queries + alignments ----
  {
    FILE* fq = fopen((out + "/query.fna").c_str(), "wb");
    FILE* fqi = fopen((out + "/query.fna.fai").c_str(), "w");
    FILE* ft = fopen((out + "/aln.tsv").c_str(), "wb");
    if (!fq || !fqi || !ft) { perror("open"); return 2; }
    setvbuf(fq, nullptr, _IOFBF, 1 << 24); setvbuf(ft, nullptr, _IOFBF, 1 << 24); setvbuf(fqi, nullptr, _IOFBF, 1 << 22);
    const uint64_t chunk = 1 << 16;
    uint64_t qoff = 0;
    for (uint64_t s0 = 0; s0 < n_sets; s0 += chunk) {
      const uint64_t s1 = std::min(n_sets, s0 + chunk);
      std::vector<std::string> qpart(T), tpart(T), ipart(T);
      std::vector<std::vector<uint32_t>> qlens(T);
      std::vector<std::thread> th;
      for (unsigned t = 0; t < T; ++t) th.emplace_back([&, t] {
        std::string& q = qpart[t]; std::string& tsv = tpart[t];
        std::vector<uint8_t> qc(256);
        char line[256];
        for (uint64_t s = s0 + (s1 - s0) * t / T; s < s0 + (s1 - s0) * (t + 1) / T; ++s) {
          uint64_t h = mix(g_seed * 101 + s);
          uint32_t p = (uint32_t)(h % P);
          while (pool_count[p + 1] == pool_count[p]) p = (p + 1) % P;  // (a pool block nobody copied)
          const uint32_t qlen = 190 + (uint32_t)((h >> 32) % 21);
          const uint32_t o = (uint32_t)((h >> 40) % (B - qlen));
          for (uint32_t i = 0; i < qlen; ++i) {
            const uint64_t hh = mix(g_seed * 211 + s * 256 + i);
            const uint8_t pb = g_pool[(size_t)p * B + o + i];
            qc[i] = ((uint32_t)hh < (uint32_t)(0.05 * 4294967296.0)) ? (uint8_t)((pb + 1 + ((hh >> 40) % 3)) & 3) : pb;
          }
          int n = snprintf(line, sizeof(line), ">q%llu\n", (unsigned long long)s);
          q.append(line, (size_t)n);
          for (uint32_t i = 0; i < qlen; i += 80) { for (uint32_t k = i; k < std::min(qlen, i + 80); ++k) q.push_back(kLetters[qc[k]]); q.push_back('\n'); }
          qlens[t].push_back(qlen | ((uint32_t)n << 16));
          const uint32_t avail = pool_count[p + 1] - pool_count[p];
          const uint32_t nh = std::min(hits, avail);
          const uint32_t start = (uint32_t)((h >> 20) % avail);
          const uint32_t step = avail > nh ? std::max(1u, avail / nh) : 1u;  // nh distinct copies, spread over the list
          for (uint32_t k = 0; k < nh; ++k) {
            const uint64_t id = by_pool[pool_count[p] + (start + k * step) % avail];
            const Copy& c = copies[id];
            uint32_t ident = 0;
            for (uint32_t i = 0; i < qlen; ++i) ident += copy_base(c, id, o + i) == qc[i];
            const uint64_t blk = (uint64_t)c.block * B;
            uint64_t r0, r1;  // 1-based inclusive, in the stored sequence
            if (!c.rc) { r0 = blk + o + 1; r1 = blk + o + qlen; }
            else { r0 = blk + (B - o); r1 = blk + (B - o - qlen) + 1; }  // reverse strand: start > stop
            n = snprintf(line, sizeof(line), "q%llu\t1\t%u\t%u\tref%u\t%llu\t%llu\t%d\t0\t%u\t%u\n", (unsigned long long)s, qlen, qlen, c.seq,
                         (unsigned long long)r0, (unsigned long long)r1, 2 * (int)ident - (int)qlen, ident, qlen);
            tsv.append(line, (size_t)n);
          }
        }
      });
      for (auto& x : th) x.join();
      uint64_t s = s0;
      for (unsigned t = 0; t < T; ++t) {
        fwrite(qpart[t].data(), 1, qpart[t].size(), fq);
        fwrite(tpart[t].data(), 1, tpart[t].size(), ft);
        for (uint32_t v : qlens[t]) {
          const uint32_t qlen = v & 0xFFFF, hn = v >> 16;
          fprintf(fqi, "q%llu\t%u\t%llu\t80\t81\n", (unsigned long long)s, qlen, (unsigned long long)(qoff + hn));
          qoff += hn + qlen + (qlen + 79) / 80;
          ++s;
        }
      }
    }
    fclose(fq); fclose(fqi); fclose(ft);
  }
  fprintf(stderr, "bigstore_gen: queries and alignments written\n");
  return 0;
}
