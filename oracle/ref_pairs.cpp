// oracle/ref_pairs.cpp -- TEST INFRASTRUCTURE (reference-side checker, never shipped, never measured as product).
//
// Thin extern "C" driver around the *reference's own* vendored SeqAn 2.4.0 headers, included IN PLACE from
// /root/reference/core/includes-external (nothing is copied into this repo).  It calls SeqAn exactly the way
// the reference's pair-level wrappers do:
//   DNA     : core/src/taxonpredictionmodelsequence.hh:133-171  (getAlignmentDNA -> globalAlignmentScore(short,long,MyersBitVector))
//   protein : core/src/taxonpredictionmodelsequence.hh:173-242  (getAlignmentProtein -> 2x globalAlignmentScore + globalAlignment + row walk)
// taxonpredictionmodelsequence.hh itself cannot be included (it needs Boost, which this image lacks), so the
// few lines of wrapper arithmetic around the SeqAn calls are restated here next to the call they wrap.
// Built by oracle/Makefile into oracle/_ref/libseqan_ref.so (git-ignored; travels to the GPU box as a binary).
#include <seqan/align.h>
#include <seqan/basic.h>
#include <seqan/sequence.h>
#include <seqan/modifier.h>
#include <seqan/score.h>
#include <chrono>
#include <cstdint>
#include <thread>
#include <vector>

using namespace seqan;

extern "C" {

// SeqAn char->ordinal tables (seqan/basic/alphabet_residue_tabs.h:107-140, 397-430)
int ref_char_to_dna5(int c) { return (int)ordValue(Dna5((char)c)); }
int ref_char_to_aa(int c)   { return (int)ordValue(AminoAcid((char)c)); }
int ref_aa_to_char(int o)   { AminoAcid a; assign(a, (unsigned)o); return (int)(char)a; }
int ref_blosum62(int a, int b) {
  Blosum62 sc; AminoAcid x, y; assign(x, (unsigned)a); assign(y, (unsigned)b);
  return score(sc, x, y);
}
int ref_blosum62_gap() { Blosum62 sc; return scoreGapExtend(sc) * 1000 + scoreGapOpen(sc); }

// reverse complement of a Dna5 string (core/src/sequencestorage.hh:364-369), in/out as ordinals
void ref_revcomp_dna5(uint8_t* s, int n) {
  String<Dna5> x; resize(x, n);
  for (int i = 0; i < n; ++i) assign(x[i], (unsigned)s[i]);
  reverseComplement(x);
  for (int i = 0; i < n; ++i) s[i] = (uint8_t)ordValue(x[i]);
}

static void fill_dna(String<Dna5>& s, const uint8_t* o, int n) { resize(s, n); for (int i = 0; i < n; ++i) assign(s[i], (unsigned)o[i]); }
static void fill_aa(String<AminoAcid>& s, const uint8_t* o, int n) { resize(s, n); for (int i = 0; i < n; ++i) assign(s[i], (unsigned)o[i]); }

// getAlignmentDNA (taxonpredictionmodelsequence.hh:133-171); inputs are Dna5 ordinals.
// out = {mutualscore(=distance), match, mismatch, gap, length}
void ref_align_dna(const uint8_t* a, int la, const uint8_t* b, int lb, int32_t* out) {
  String<Dna5> A, B; fill_dna(A, a, la); fill_dna(B, b, lb);
  const String<Dna5>* long_seq = &A; const String<Dna5>* short_seq = &B;
  if (length(A) < length(B)) { long_seq = &B; short_seq = &A; }
  int mutualscore = -globalAlignmentScore(*short_seq, *long_seq, MyersBitVector());
  int lendiff = length(*long_seq) - length(*short_seq);
  int gap = lendiff;
  int mismatch = mutualscore - lendiff;
  int match = length(*short_seq) - mismatch;
  out[0] = mutualscore; out[1] = match; out[2] = mismatch; out[3] = gap; out[4] = match + mismatch + gap;
}

// getAlignmentProtein (taxonpredictionmodelsequence.hh:173-242); inputs are AminoAcid ordinals.
// out = {selfscore, mutualscore, match, mismatch, gap, len}; fout = {distance, similarity}
void ref_align_protein(const uint8_t* a, int la, const uint8_t* b, int lb, int32_t* out, float* fout) {
  typedef String<AminoAcid> S; typedef Align<S, ArrayGaps> TAlign; typedef Row<TAlign>::Type TRow; typedef Iterator<TRow>::Type TIt;
  S A, B; fill_aa(A, a, la); fill_aa(B, b, lb);
  auto algo = LinearGaps(); auto sc = Blosum62();
  int selfscore = globalAlignmentScore(A, A, sc, algo) + globalAlignmentScore(B, B, sc, algo);
  TAlign aln; resize(rows(aln), 2); assignSource(row(aln, 0), A); assignSource(row(aln, 1), B);
  int mutualscore = globalAlignment(aln, sc, algo);
  TRow& r1 = row(aln, 0); TRow& r2 = row(aln, 1);
  int gap = 0, match = 0, mismatch = 0;
  TIt i1 = begin(r1), e1 = end(r1), i2 = begin(r2);
  for (; i1 != e1; ++i1, ++i2) {
    if (isGap(i1) || isGap(i2)) gap++; else if (*i1 == *i2) match++; else mismatch++;
  }
  unsigned len = gap + match + mismatch;
  float normfactor = len / static_cast<float>(selfscore);
  out[0] = selfscore; out[1] = mutualscore; out[2] = match; out[3] = mismatch; out[4] = gap; out[5] = (int)len;
  fout[0] = (selfscore - 2 * mutualscore) * normfactor;
  fout[1] = (2 * mutualscore) * normfactor;
}

// Timing helpers for bench.py's cpu_reference leg: the same two calls over many pairs without a Python round trip
// per pair.  Sequences are ordinals laid out back to back; sequence i = [offsets[i], offsets[i+1]).
void ref_align_dna_batch(const uint8_t* letters, const uint64_t* offsets, const uint32_t* pa, const uint32_t* pb, int n,
                         int32_t* out5) {
  for (int k = 0; k < n; ++k)
    ref_align_dna(letters + offsets[pa[k]], (int)(offsets[pa[k] + 1] - offsets[pa[k]]), letters + offsets[pb[k]],
                  (int)(offsets[pb[k] + 1] - offsets[pb[k]]), out5 + 5 * (size_t)k);
}
void ref_align_protein_batch(const uint8_t* letters, const uint64_t* offsets, const uint32_t* pa, const uint32_t* pb,
                             int n, int32_t* out6, float* fout2) {
  for (int k = 0; k < n; ++k)
    ref_align_protein(letters + offsets[pa[k]], (int)(offsets[pa[k] + 1] - offsets[pa[k]]), letters + offsets[pb[k]],
                      (int)(offsets[pb[k] + 1] - offsets[pb[k]]), out6 + 6 * (size_t)k, fout2 + 2 * (size_t)k);
}

// The timed form: every String is built BEFORE the clock starts (the reference builds them in getSequence, outside
// getAlignment), then n_threads std::threads run `reps` passes over their slice of the pairs.  Returns the wall seconds
// from the moment all threads are released to the moment the last one is done; *checksum (sum of the distances / mutual
// scores) keeps the optimiser honest and lets the caller compare runs.
double ref_time_dna_pairs(const uint8_t* letters, const uint64_t* offsets, const uint32_t* pa, const uint32_t* pb, int n, int n_threads,
                          int reps, long long* checksum) {
  std::vector<String<Dna5> > A(n), B(n);
  for (int k = 0; k < n; ++k) {
    fill_dna(A[k], letters + offsets[pa[k]], (int)(offsets[pa[k] + 1] - offsets[pa[k]]));
    fill_dna(B[k], letters + offsets[pb[k]], (int)(offsets[pb[k] + 1] - offsets[pb[k]]));
  }
  if (n_threads < 1) n_threads = 1;
  std::vector<long long> sums(n_threads, 0);
  std::vector<std::thread> th;
  const auto t0 = std::chrono::steady_clock::now();
  for (int t = 0; t < n_threads; ++t)
    th.emplace_back([&, t] {
      const int lo = (int)((long long)n * t / n_threads), hi = (int)((long long)n * (t + 1) / n_threads);
      long long s = 0;
      for (int r = 0; r < reps; ++r)
        for (int k = lo; k < hi; ++k) {  // getAlignmentDNA's call (:144-150): shorter sequence first
          const String<Dna5>& x = length(A[k]) < length(B[k]) ? A[k] : B[k];
          const String<Dna5>& y = length(A[k]) < length(B[k]) ? B[k] : A[k];
          s += -globalAlignmentScore(x, y, MyersBitVector());
        }
      sums[t] = s;
    });
  for (auto& x : th) x.join();
  const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  long long total = 0;
  for (long long v : sums) total += v;
  if (checksum) *checksum = total;
  return dt;
}

double ref_time_protein_pairs(const uint8_t* letters, const uint64_t* offsets, const uint32_t* pa, const uint32_t* pb, int n, int n_threads,
                              int reps, long long* checksum) {
  typedef String<AminoAcid> S; typedef Align<S, ArrayGaps> TAlign;
  std::vector<S> A(n), B(n);
  for (int k = 0; k < n; ++k) {
    fill_aa(A[k], letters + offsets[pa[k]], (int)(offsets[pa[k] + 1] - offsets[pa[k]]));
    fill_aa(B[k], letters + offsets[pb[k]], (int)(offsets[pb[k] + 1] - offsets[pb[k]]));
  }
  if (n_threads < 1) n_threads = 1;
  std::vector<long long> sums(n_threads, 0);
  std::vector<std::thread> th;
  const auto t0 = std::chrono::steady_clock::now();
  for (int t = 0; t < n_threads; ++t)
    th.emplace_back([&, t] {
      const int lo = (int)((long long)n * t / n_threads), hi = (int)((long long)n * (t + 1) / n_threads);
      long long s = 0;
      auto algo = LinearGaps(); auto sc = Blosum62();
      for (int r = 0; r < reps; ++r)
        for (int k = lo; k < hi; ++k) {  // getAlignmentProtein's calls (:190-199): two self scores + the traced alignment
          s += globalAlignmentScore(A[k], A[k], sc, algo) + globalAlignmentScore(B[k], B[k], sc, algo);
          TAlign aln; resize(rows(aln), 2); assignSource(row(aln, 0), A[k]); assignSource(row(aln, 1), B[k]);
          s += globalAlignment(aln, sc, algo);
        }
      sums[t] = s;
    });
  for (auto& x : th) x.join();
  const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  long long total = 0;
  for (long long v : sums) total += v;
  if (checksum) *checksum = total;
  return dt;
}

}  // extern "C"
