// taxator-tk_amd/csrc/align_dna_kernels.hip -- nucleotide re-alignment kernels for gfx950 (K1 short, K2 long).
//
// What they replace: getAlignmentDNA (core/src/taxonpredictionmodelsequence.hh:133-171) and the SeqAn routine it
// calls, _globalAlignmentScore(..., MyersBitVector) (seqan/align/global_alignment_myers_impl.h:62-196): the
// unit-cost global edit distance of two Dna5 strings (N matches only N), from which the reference derives
//   distance = d, gaps = |len difference|, mismatches = d - gaps, matches = len_short - mismatches (= len_long - d).
// The reference keeps VP/VN/Peq in heap arrays of 32-bit blocks and walks them block by block per text character.
//
// K1 (pattern <= 256 nt): ONE ALIGNMENT PER LANE.  The delta vectors VP/VN of the shorter sequence live in NW
// 32-bit VGPRs each (NW = ceil(m/32) is a template parameter; pairs are binned by NW so a wave runs one shape);
// the match masks Peq[5][NW] live in LDS, laid out [symbol][word][lane] so that a lane always hits bank (lane % 32)
// whatever its text symbol is -> conflict-free ds_read_b32 with an immediate word offset.  The text is streamed from
// the lane's 16-byte aligned segment slot with dwordx4 loads (32 nt per load).  Integer VALU bound (~13 ops per
// 32-bit word per column); HBM traffic is the algorithmic minimum (each segment byte is read once per pair).
// No MFMA: this is bit-parallel integer DP, not a contraction.
//
// K2 (pattern 257..16384 nt): ONE ALIGNMENT PER WAVE, words striped over lanes, anti-diagonal (skewed) schedule with
// the three inter-word carries handed lane to lane by DPP shifts.  See below.
#include <stdlib.h>

#include "rpa_device.h"

namespace rpa {

// Compress the bit `b` of each of the 8 nibbles of x into 8 contiguous bits (nibble k -> bit k).
__device__ __forceinline__ uint32_t nibble_plane(uint32_t x, int b) {
  uint32_t t = (x >> b) & 0x11111111u;        // bit at 4k
  t = (t | (t >> 3)) & 0x03030303u;           // per byte: 2 bits at positions 0,1
  return (t * 0x01041040u) >> 24;             // gather the four 2-bit fields: b0 | b1<<2 | b2<<4 | b3<<6
}

// Bit planes (32 residues per word) of four packed dwords.
__device__ __forceinline__ void planes32(const uint4 x, uint32_t& p0, uint32_t& p1, uint32_t& pn) {
  p0 = nibble_plane(x.x, 0) | (nibble_plane(x.y, 0) << 8) | (nibble_plane(x.z, 0) << 16) | (nibble_plane(x.w, 0) << 24);
  p1 = nibble_plane(x.x, 1) | (nibble_plane(x.y, 1) << 8) | (nibble_plane(x.z, 1) << 16) | (nibble_plane(x.w, 1) << 24);
  pn = nibble_plane(x.x, 2) | (nibble_plane(x.y, 2) << 8) | (nibble_plane(x.z, 2) << 16) | (nibble_plane(x.w, 2) << 24);
}

// upd = DP cells the kernel really updates for this pair (words x 32 rows x columns it sweeps): what the banded kernels
// save shows as the ratio to la x lb ("cell-equivalents")
__device__ __forceinline__ void profile_add(unsigned long long* acc, bool active, uint32_t la, uint32_t lb, unsigned long long upd) {
  // block-level reduction, then 4 atomics per workgroup (only when profiling is on)
  __shared__ unsigned long long red[4][kBlock / 64];
  unsigned long long v0 = active ? 1ull : 0ull, v1 = active ? (unsigned long long)la + lb + 8ull : 0ull,
                     v2 = active ? (unsigned long long)la * lb : 0ull, v3 = active ? upd : 0ull;
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    v0 += __shfl_down(v0, d, 64);
    v1 += __shfl_down(v1, d, 64);
    v2 += __shfl_down(v2, d, 64);
    v3 += __shfl_down(v3, d, 64);
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) { red[0][wave] = v0; red[1][wave] = v1; red[2][wave] = v2; red[3][wave] = v3; }
  __syncthreads();
  if (threadIdx.x < 4) {
    unsigned long long s = 0;
    for (int k = 0; k < kBlock / 64; ++k) s += red[threadIdx.x][k];
    atomicAdd(&acc[(blockIdx.x & (kProfParts - 1)) * kProfSlots + (threadIdx.x < 3 ? threadIdx.x : 4)], s);  // partitioned: same-address atomics serialise
  }
}
// updated cells that are only known behind the work (the strip kernels' adaptive bands): every lane of the workgroup calls
// this once, converged, at the end of the kernel
__device__ __forceinline__ void profile_add_updated(unsigned long long* acc, unsigned long long upd) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) upd += __shfl_down(upd, d, 64);
  if ((threadIdx.x & 63) == 0 && upd) atomicAdd(&acc[(blockIdx.x & (kProfParts - 1)) * kProfSlots + 4], upd);
}
// word-columns the banded one-strip kernels sweep for a text of n columns (their compile-time window schedules, restated):
// HW words either side of word g for the 32 columns of text group g ...
__device__ __forceinline__ unsigned long long band_word_columns(int NW, int HW, uint32_t n) {
  unsigned long long wc = 0;
  for (uint32_t g = 0; g * 32u < n; ++g) {
    const uint32_t cols = n - g * 32u < 32u ? n - g * 32u : 32u;
    const int lo = (int)g - HW > 0 ? (int)g - HW : 0, hi = (int)g + HW < NW - 1 ? (int)g + HW : NW - 1;
    if (hi >= lo) wc += (unsigned long long)cols * (unsigned)(hi - lo + 1);
  }
  return wc;
}
// ... and the four-word window half a group out of phase, with (PH = 3) or without (PH = 2) the three-word middle (BandGroups4)
__device__ __forceinline__ unsigned long long band4_word_columns(int NW, int PH, uint32_t n) {
  unsigned long long wc = 0;
  const uint32_t kFirst = PH == 3 ? 8u : 16u, kLast = PH == 3 ? 24u : 16u;
  auto span = [&](int lo, int hi) { lo = lo > 0 ? lo : 0; hi = hi < NW - 1 ? hi : NW - 1; return hi >= lo ? (unsigned)(hi - lo + 1) : 0u; };
  for (uint32_t g = 0; g * 32u < n; ++g) {
    const uint32_t cols = n - g * 32u < 32u ? n - g * 32u : 32u;
    const uint32_t ca = cols < kFirst ? cols : kFirst;
    wc += (unsigned long long)ca * span((int)g - 2, (int)g + 1);
    if (cols > kFirst) {
      if (PH == 3) wc += (unsigned long long)((cols < kLast ? cols : kLast) - kFirst) * span((int)g - 1, (int)g + 1);
      if (cols > kLast) wc += (unsigned long long)(cols - kLast) * span((int)g - 1, (int)g + 2);
    }
  }
  return wc;
}

// ---- pieces shared by the one-strip kernels ----------------------------------------------------------------
// The five match masks of pattern word `wi` (32 residues from the stream; residues past the pattern end match nothing).
__device__ __forceinline__ void peq_word(NtStream& ps, uint32_t wi, uint32_t m, uint32_t (&mask)[5]) {
#pragma unroll
  for (int c = 0; c < 5; ++c) mask[c] = 0u;
  if (wi * 32u < m) {
    const uint4 x = ps.next32();
    uint32_t p0, p1, pn;
    planes32(x, p0, p1, pn);
    const uint32_t rem = m - wi * 32u;
    const uint32_t vm = rem >= 32u ? 0xFFFFFFFFu : ((1u << rem) - 1u);
    mask[4] = pn & vm;             // N
    mask[3] = p1 & p0 & vm;        // T
    mask[2] = p1 & ~p0 & vm;       // G
    mask[1] = p0 & ~p1 & ~pn & vm; // C
    mask[0] = ~(p0 | p1 | pn) & vm;  // A
  }
}
// ---- the word update of a column, shared by the lane-per-pair kernels ---------------------------------------------------
// Words W0 .. W0 + C - 1 (C <= 4) of VP / VN for one text symbol: the add chain word by word (v_addc_co), then the two
// shifts by one (HP << 1 | carry, HN << 1 | carry) per PAIR of words -- one v_lshl_add_u64 moves two words in 2.3 ns where
// two v_alignbit_b32 take 2 x 1.8 and more inside this instruction mix (tools/valu_rate5.hip; tools/myers_pair.hip: 19.6
// -> 15.9 ns per word for the whole step); the carry into an upper pair is the top bit of the word below it (one shift),
// into a column's first word (kFirst) the constants 1 and 0: global alignment, +1 enters row 0.  An odd last word takes
// v_alignbit.  eq(wi) = the match mask of word wi; carry = the add carry in / out; hp_in / hn_in = HP / HN of the word
// below W0 (their top bits enter) in, of the last word out.  10 VALU ops per word with v_alignbit (or, and, addc, 3x
// v_bitop3 -- gfx950's three-input boolean; truth tables use a=0xF0, b=0xCC, c=0xAA --, 2x and, 2x alignbit), 9.5 in pairs.
#ifndef RPA_PAIR_SHIFT
#define RPA_PAIR_SHIFT 1
#endif
__device__ __forceinline__ uint64_t shl1_add(uint32_t lo, uint32_t hi, uint32_t carry_in) {
  const uint64_t x = ((uint64_t)hi << 32) | lo, c = carry_in;
  uint64_t r;
  asm("v_lshl_add_u64 %0, %1, 1, %2" : "=v"(r) : "v"(x), "v"(c));
  return r;
}
template <int IMM>
__device__ __forceinline__ uint64_t shl1_add_imm(uint32_t lo, uint32_t hi) {
  const uint64_t x = ((uint64_t)hi << 32) | lo;
  uint64_t r;
  if (IMM) asm("v_lshl_add_u64 %0, %1, 1, 1" : "=v"(r) : "v"(x));
  else asm("v_lshl_add_u64 %0, %1, 1, 0" : "=v"(r) : "v"(x));
  return r;
}
template <int NW, int W0, int C, bool kFirst, bool kPairs, typename EqF>
__device__ __forceinline__ void myers_words(uint32_t (&VP)[NW], uint32_t (&VN)[NW], EqF&& eq, uint32_t& carry, uint32_t& hp_in, uint32_t& hn_in) {
  static_assert(C >= 1 && C <= 4 && W0 + C <= NW, "a group of up to four words");
  uint32_t D0[C], HP[C], HN[C], Xs[C], Hs[C];
#pragma unroll
  for (int k = 0; k < C; ++k) {
    const uint32_t vp = VP[W0 + k], vn = VN[W0 + k];
    const uint32_t X = eq(W0 + k) | vn;
    uint32_t cout;
    const uint32_t sum = __builtin_addc(vp, X & vp, carry, &cout);
    carry = cout;
    D0[k] = __builtin_amdgcn_bitop3_b32(sum, vp, X, 0xBE);         // (sum ^ vp) | X
    HN[k] = vp & D0[k];
    HP[k] = __builtin_amdgcn_bitop3_b32(vn, vp, D0[k], 0xF1);      // vn | ~(vp | D0)
  }
  if constexpr (kPairs) {
  constexpr int U = C >= 2 ? 1 : 0;  // (the upper word of a pair; C == 1 has no pair and never gets here)
#pragma unroll
  for (int k = 0; k + 1 < C; k += 2) {
    uint64_t xs, hs;
    if (k == 0 && kFirst) { xs = shl1_add_imm<1>(HP[0], HP[U]); hs = shl1_add_imm<0>(HN[0], HN[U]); }
    else {
      xs = shl1_add(HP[k], HP[k + U], (k ? HP[k ? k - 1 : 0] : hp_in) >> 31);
      hs = shl1_add(HN[k], HN[k + U], (k ? HN[k ? k - 1 : 0] : hn_in) >> 31);
    }
    Xs[k] = (uint32_t)xs; Xs[k + U] = (uint32_t)(xs >> 32); Hs[k] = (uint32_t)hs; Hs[k + U] = (uint32_t)(hs >> 32);
  }
  if constexpr (C & 1) {
    constexpr int k = C - 1;
    Xs[k] = __builtin_amdgcn_alignbit(HP[k], k ? HP[k ? k - 1 : 0] : hp_in, 31);  // (HP << 1) | (below >> 31)
    Hs[k] = __builtin_amdgcn_alignbit(HN[k], k ? HN[k ? k - 1 : 0] : hn_in, 31);
  }
  } else {
#pragma unroll
  for (int k = 0; k < C; ++k) {
    Xs[k] = __builtin_amdgcn_alignbit(HP[k], k ? HP[k ? k - 1 : 0] : hp_in, 31);
    Hs[k] = __builtin_amdgcn_alignbit(HN[k], k ? HN[k ? k - 1 : 0] : hn_in, 31);
  }
  }
#pragma unroll
  for (int k = 0; k < C; ++k) {
    VN[W0 + k] = Xs[k] & D0[k];
    VP[W0 + k] = __builtin_amdgcn_bitop3_b32(Hs[k], Xs[k], D0[k], 0xF1);  // Hs | ~(Xs | D0)
  }
  hp_in = HP[C - 1]; hn_in = HN[C - 1];
}
// words W0 .. W1 of a column in groups of four
template <int NW, int W0, int W1, bool kFirst, bool kPairs = (RPA_PAIR_SHIFT != 0), typename EqF>
__device__ __forceinline__ void myers_word_range(uint32_t (&VP)[NW], uint32_t (&VN)[NW], EqF&& eq, uint32_t& carry, uint32_t& hp_in, uint32_t& hn_in) {
  constexpr int C = W1 - W0 + 1 < 4 ? W1 - W0 + 1 : 4;
  myers_words<NW, W0, C, kFirst, kPairs>(VP, VN, eq, carry, hp_in, hn_in);
  if constexpr (W0 + C <= W1) myers_word_range<NW, W0 + C, W1, false, kPairs>(VP, VN, eq, carry, hp_in, hn_in);
}
#ifndef RPA_PAIR_SHIFT_STRIPS
#define RPA_PAIR_SHIFT_STRIPS 1   // k_myers_band / k_myers_band_duo (strips of 5..8 words)
#endif
#ifndef RPA_PAIR_SHIFT_FULL
#define RPA_PAIR_SHIFT_FULL 1     // k_myers_short (every word of the pattern)
#endif
// One text column: NW chained word updates; e[wi * STRIDE] is the match mask of word wi for the column's symbol.
template <int NW, int STRIDE>
__device__ __forceinline__ void myers_column(uint32_t (&VP)[NW], uint32_t (&VN)[NW], const uint32_t* __restrict__ e) {
  uint32_t carry = 0, hp_prev = 0x80000000u /* global alignment: +1 enters row 0 */, hn_prev = 0;
  myers_word_range<NW, 0, NW - 1, true, (RPA_PAIR_SHIFT_FULL != 0)>(VP, VN, [&](int wi) { return e[wi * STRIDE]; }, carry, hp_prev, hn_prev);
}
// Walk the text front to back and call column(symbol) once per nucleotide.  The n mod 8 leading columns go first (a
// short loop that every lane of the wave runs at the same time), so that the main loop only ever sees full dwords: 8
// unrolled columns per dword, no partial-dword branch and no tail in which a wave executes both a fast and a slow
// path; then full mod 4 single dwords, then 32 nucleotides (one 16-byte load, issued a step ahead) per step.
template <typename F>
__device__ __forceinline__ void walk_text(const SegView& txt, F&& column) {
  const uint32_t n = txt.len, head = n & 7u, full = n >> 3;
  NtStream ts;
  ts.init(txt, 0u);
  if (head) {
    const uint32_t d = ts.next8();  // positions 0..7; only the first n mod 8 are consumed here
#pragma unroll 1
    for (uint32_t k = 0; k < head; ++k) column((d >> (4 * k)) & 7u);
    if (full) ts.init(txt, head);   // the main loop starts at position n mod 8
  }
  if (full) {
    const uint32_t singles = full & 3u, chunks = full >> 2;
#pragma unroll 1
    for (uint32_t k = 0; k < singles; ++k) {
      const uint32_t d = ts.next8();
#pragma unroll
      for (int c = 0; c < 8; ++c) column((d >> (4 * c)) & 7u);
    }
    if (chunks) {
      uint4 pending = ts.load4();
      for (uint32_t ch = 0; ch < chunks; ++ch) {
        const uint4 d = ts.take32(pending);
        if (ch + 1 < chunks) pending = ts.load4();  // prefetch for the next step
#pragma unroll
        for (int c = 0; c < 8; ++c) column((d.x >> (4 * c)) & 7u);
#pragma unroll
        for (int c = 0; c < 8; ++c) column((d.y >> (4 * c)) & 7u);
#pragma unroll
        for (int c = 0; c < 8; ++c) column((d.z >> (4 * c)) & 7u);
#pragma unroll
        for (int c = 0; c < 8; ++c) column((d.w >> (4 * c)) & 7u);
      }
    }
  }
}
// The score is not tracked per column (the reference adds the horizontal delta of the pattern's last row after every
// text character, global_alignment_myers_impl.h:120-123): row 0 of a global alignment holds D[0][j] = j, so
// D[m][n] = n + sum of the vertical deltas of the last column = n + popc(VP & rows<m) - popc(VN & rows<m).
// This also makes a kernel exact for ANY m <= 32*NW (rows past the pattern match nothing and sit below row m).
template <int NW>
__device__ __forceinline__ int32_t myers_distance(const uint32_t (&VP)[NW], const uint32_t (&VN)[NW], uint32_t n, uint32_t m) {
  int32_t dist = (int32_t)n;
#pragma unroll
  for (int wi = 0; wi < NW; ++wi) {
    const uint32_t pos = (uint32_t)wi * 32u;
    const uint32_t vm = pos >= m ? 0u : (m - pos >= 32u ? 0xFFFFFFFFu : ((1u << (m - pos)) - 1u));
    dist += (int32_t)__popc(VP[wi] & vm) - (int32_t)__popc(VN[wi] & vm);
  }
  return dist;
}
// distance, similarity = matches = len_short - (d - lendiff) = len_long - d   (:154-164)
__device__ __forceinline__ void store_dna_result(const WorkDev& w, uint32_t dst, int32_t dist, uint32_t n, uint32_t m) {
  const uint32_t llong = n > m ? n : m, lshort = n > m ? m : n;
  w.pair_res[res_index(w, dst)] = make_float2((float)dist, (float)((int32_t)llong - dist));
  if (w.pair_ext) w.pair_ext[dst] = make_uint4(llong - (uint32_t)dist, (uint32_t)dist - (llong - lshort), llong - lshort, llong);  // match+mismatch = len_short, + gap = len_long
}

// The kernel walks up to two work lists back to back (list, then list2): a pattern-length class with too few pairs
// to fill the chip rides along with the next wider class instead of paying a serialised launch of its own.
// n_list_dev != null: `list` is a retry list whose length only the device knows (n_list bounds it) and the grid strides.
template <int NW>
__global__ __launch_bounds__(kBlock) void k_myers_short(const uint2* __restrict__ list, uint32_t n_list, const uint32_t* __restrict__ n_list_dev,
                                                         uint32_t dev_split, const uint2* __restrict__ list2, uint32_t n_list2, WorkDev w,
                                                         unsigned long long* prof_acc) {
  extern __shared__ uint32_t peq[];  // [5][NW][kBlock]
  const int tid = threadIdx.x;
  if (n_list_dev) {
    if (*n_list_dev <= dev_split) return;  // (a list of up to dev_split pairs is k_myers_short_wave's, launched in front of this kernel; 0: all mine)
    n_list = *n_list_dev < n_list ? *n_list_dev : n_list;
    if (blockIdx.x == 0 && tid == 0 && n_list) atomicAdd(&w.counters[kCtrRetryShortTotal], n_list);  // what the band cost this batch
  }
  const uint32_t total = n_list + n_list2;
  for (uint32_t base = blockIdx.x * kBlock; base < total; base += gridDim.x * kBlock) {
    const uint32_t t = base + tid;
    const bool active = t < total;
    uint32_t dst = 0, m = 0, n = 0;
    SegView pat{}, txt{};
    if (active) {
      uint32_t a, b;
      resolve_pair(w, t < n_list ? list[t] : list2[t - n_list], a, b, dst);
      SegView va = seg_view(w, a), vb = seg_view(w, b);
      // The reference takes the shorter sequence as the pattern (:144-147); the distance is symmetric, so any choice is
      // exact.  Cost = words(pattern) x columns(text): when both need the same number of 32-row words the LONGER one
      // becomes the pattern, and the lane walks the shorter one's columns (2-3 % fewer on ~200 nt segments).
      const bool a_pat = (va.len + 31u) / 32u == (vb.len + 31u) / 32u ? va.len >= vb.len : va.len < vb.len;
      if (a_pat) { txt = vb; pat = va; } else { txt = va; pat = vb; }
      n = txt.len; m = pat.len;
    }
    if (prof_acc) profile_add(prof_acc, active, n, m, (unsigned long long)n * (NW * 32u));

    // ---- Peq build (bit planes -> five match masks per 32-residue word); the pattern is read 16 bytes per word ----
    NtStream ps;
    if (active) ps.init(pat, 0u);
#pragma unroll
    for (int wi = 0; wi < NW; ++wi) {
      uint32_t mask[5] = {0u, 0u, 0u, 0u, 0u};
      if (active) peq_word(ps, (uint32_t)wi, m, mask);
#pragma unroll
      for (int c = 0; c < 5; ++c) peq[(c * NW + wi) * kBlock + tid] = mask[c];
    }
    if (!active) continue;  // (no barrier anywhere: a lane only ever touches its own LDS column)

    uint32_t VP[NW], VN[NW];
#pragma unroll
    for (int wi = 0; wi < NW; ++wi) { VP[wi] = 0xFFFFFFFFu; VN[wi] = 0u; }
    walk_text(txt, [&](uint32_t c) { myers_column<NW, kBlock>(VP, VN, peq + c * (NW * kBlock) + tid); });
    store_dna_result(w, dst, myers_distance<NW>(VP, VN, n, m), n, m);
  }
}

// K1 inside Ukkonen's band: the same lane-per-pair kernel, but the 32 columns of text group g only update the pattern
// words g - HW .. g + HW (a window that slides down one word per group; compile-time word indices, the loop over the
// groups is fully unrolled).  Words below the window keep the VP = all ones they started with until the window reaches
// them (an upper bound of the true column), words above it are dropped and "+1 per column" comes down from there.  As
// in k_myers_band every value D' is >= the true D and equal to it along paths inside the band, which here is
//     min(0, delta) - h <= j - i <= max(0, delta) + h,   h = 32 HW - |delta|,   delta = n - m,
// so D'[m][n] <= k = 2 h + |delta| = 64 HW - |delta| is the distance, and anything larger proves the distance is > k: such
// pairs (and pairs with |delta| >= 32 HW) go onto the retry list and through k_myers_short.  The score is followed at the
// top of the window: T = D'[row above the window][column] grows by 1 per column and by the dropped word's vertical
// deltas when the window moves; at the end the vertical deltas of the words from the window's top down are added
// (tools/band_model.py: banded_window).  HW = 2 covers distances up to ~120 on 200 nt pairs: config 2's pass-1 pairs
// (55-65 edits) update 4.3 of their 7 words per column on average.
template <int NW, int WLO, int WHI>
__device__ __forceinline__ void myers_column_window(uint32_t (&VP)[NW], uint32_t (&VN)[NW], const uint32_t* __restrict__ e) {
  uint32_t carry = 0, hp_prev = 0x80000000u, hn_prev = 0;
  if constexpr (WLO <= WHI) myers_word_range<NW, WLO, WHI, true>(VP, VN, [&](int wi) { return e[wi * kBlock]; }, carry, hp_prev, hn_prev);
}

template <int NW, int HW, int G>
struct BandGroups {  // text group G and the ones after it (compile-time recursion: the word window is a constant per group)
  static __device__ __forceinline__ void run(uint32_t (&VP)[NW], uint32_t (&VN)[NW], const uint32_t* __restrict__ peq_lane,
                                             NtStream& ts, uint4& pending, uint32_t n, uint32_t n_groups, int32_t& T) {
    if ((uint32_t)G >= n_groups) return;
    constexpr int WLO = G - HW > 0 ? G - HW : 0, WHI = G + HW < NW - 1 ? G + HW : NW - 1;
    const uint4 buf = ts.take32(pending);
    if ((uint32_t)G + 1u < n_groups) pending = ts.load4();
    const uint32_t cols = n - (uint32_t)G * 32u < 32u ? n - (uint32_t)G * 32u : 32u;
#pragma unroll 1
    for (uint32_t q = 0; q < 4u && q * 8u < cols; ++q) {
      const uint32_t d = q == 0u ? buf.x : q == 1u ? buf.y : q == 2u ? buf.z : buf.w;
      const uint32_t left = cols - q * 8u;
      if (left >= 8u) {
#pragma unroll
        for (int k = 0; k < 8; ++k) myers_column_window<NW, WLO, WHI>(VP, VN, peq_lane + ((d >> (4 * k)) & 7u) * (NW * kBlock));
      } else {
#pragma unroll 1
        for (uint32_t k = 0; k < left; ++k) myers_column_window<NW, WLO, WHI>(VP, VN, peq_lane + ((d >> (4u * k)) & 7u) * (NW * kBlock));
      }
    }
    T += (int32_t)cols;
    if constexpr (G - HW >= 0 && G - HW <= NW - 1) {
      if ((uint32_t)G + 1u < n_groups) T += (int32_t)__popc(VP[G - HW]) - (int32_t)__popc(VN[G - HW]);  // the window moves down a word
    }
    if constexpr (G + 1 < NW + HW) BandGroups<NW, HW, G + 1>::run(VP, VN, peq_lane, ts, pending, n, n_groups, T);
  }
};

template <int NW, int HW>
__global__ __launch_bounds__(kBlock) void k_myers_short_band(const uint2* __restrict__ list, uint32_t n_list,
                                                              const uint2* __restrict__ list2, uint32_t n_list2, WorkDev w,
                                                              uint2* __restrict__ retry, uint32_t* __restrict__ retry_n,
                                                              unsigned long long* prof_acc) {
  extern __shared__ uint32_t peq[];  // [5][NW][kBlock]
  const int tid = threadIdx.x;
  const uint32_t t = blockIdx.x * kBlock + tid;
  const bool active = t < n_list + n_list2;
  uint32_t dst = 0, m = 0, n = 0;
  SegView pat{}, txt{};
  uint2 entry = make_uint2(0u, 0u);
  if (active) {
    uint32_t a, b;
    entry = t < n_list ? list[t] : list2[t - n_list];
    resolve_pair(w, entry, a, b, dst);
    SegView va = seg_view(w, a), vb = seg_view(w, b);
    const bool a_pat = (va.len + 31u) / 32u == (vb.len + 31u) / 32u ? va.len >= vb.len : va.len < vb.len;  // as in k_myers_short
    if (a_pat) { txt = vb; pat = va; } else { txt = va; pat = vb; }
    n = txt.len; m = pat.len;
  }
  if (prof_acc) profile_add(prof_acc, active, n, m, band_word_columns(NW, HW, n) * 32ull);
  NtStream ps;
  if (active) ps.init(pat, 0u);
#pragma unroll
  for (int wi = 0; wi < NW; ++wi) {
    uint32_t mask[5] = {0u, 0u, 0u, 0u, 0u};
    if (active) peq_word(ps, (uint32_t)wi, m, mask);
#pragma unroll
    for (int c = 0; c < 5; ++c) peq[(c * NW + wi) * kBlock + tid] = mask[c];
  }
  if (!active) return;
  const uint32_t ad = n > m ? n - m : m - n;
  if (ad >= 32u * HW) { retry[atomicAdd(retry_n, 1u)] = entry; return; }  // no band of this shape holds the corridor
  uint32_t VP[NW], VN[NW];
#pragma unroll
  for (int wi = 0; wi < NW; ++wi) { VP[wi] = 0xFFFFFFFFu; VN[wi] = 0u; }
  const uint32_t n_groups = (n + 31u) >> 5;  // <= NW + HW: n < m + 32 HW <= 32 (NW + HW)
  int32_t T = 0;
  if (n_groups) {
    NtStream ts;
    ts.init(txt, 0u);
    uint4 pending = ts.load4();
    BandGroups<NW, HW, 0>::run(VP, VN, peq + tid, ts, pending, n, n_groups, T);
  }
  const int32_t wlo_final = (int32_t)n_groups - 1 - HW;  // first word of the last group's window (<= 0: from word 0)
  int32_t dist = T;
#pragma unroll
  for (int wi = 0; wi < NW; ++wi) {
    const uint32_t pos = (uint32_t)wi * 32u;
    const uint32_t vm = pos >= m ? 0u : (m - pos >= 32u ? 0xFFFFFFFFu : ((1u << (m - pos)) - 1u));
    if (wi >= wlo_final) dist += (int32_t)__popc(VP[wi] & vm) - (int32_t)__popc(VN[wi] & vm);
  }
  if ((uint32_t)dist <= 64u * HW - ad) store_dna_result(w, dst, dist, n, m);
  else retry[atomicAdd(retry_n, 1u)] = entry;
}

// The same with a window of FOUR words, half a group out of phase with the text groups: the first 16 columns of group g
// update words g-2 .. g+1, the last 16 columns words g-1 .. g+2.  A window that slides one word per 32 columns guarantees a
// band of 16 (words - 1) diagonals either side when it is centred like this: |j - i| <= 48, i.e. h = 48 - |delta|,
// k = 96 - |delta| (five words, in phase: 64 and 128).  Config 2's pass-1 pairs (55-65 edits, |delta| < 20) still fit; a
// column costs 3.6 word updates on average instead of 4.3 (of 7).  tools/band_model.py: banded_window_phase.  (The default
// since round 3 drops to THREE words in the middle 16 columns of a group: BandGroups4 below.)
// The same column over a RING of four table slots: word w's masks sit in slot w & 3 of a [symbol][4][lane] table (the
// four-word window never holds two words of the same slot), so the table is 20 KB per workgroup whatever NW is.
template <int NW, int WLO, int WHI>
__device__ __forceinline__ void myers_column_ring(uint32_t (&VP)[NW], uint32_t (&VN)[NW], const uint32_t* __restrict__ e) {
  uint32_t carry = 0, hp_prev = 0x80000000u, hn_prev = 0;
  if constexpr (WLO <= WHI) myers_word_range<NW, WLO, WHI, true>(VP, VN, [&](int wi) { return e[(wi & 3) * kBlock]; }, carry, hp_prev, hn_prev);
}

template <int NW, int WLO, int WHI>
__device__ __forceinline__ void band_sweep16(uint32_t (&VP)[NW], uint32_t (&VN)[NW], const uint32_t* __restrict__ peq_lane, uint32_t d0,
                                             uint32_t d1, uint32_t ncols) {
#pragma unroll 1
  for (uint32_t q = 0; q < 2u && q * 8u < ncols; ++q) {
    const uint32_t d = q == 0u ? d0 : d1;
    const uint32_t left = ncols - q * 8u;
    if (left >= 8u) {
#pragma unroll
      for (int k = 0; k < 8; ++k) myers_column_ring<NW, WLO, WHI>(VP, VN, peq_lane + ((d >> (4 * k)) & 7u) * (4 * kBlock));
    } else {
#pragma unroll 1
      for (uint32_t k = 0; k < left; ++k) myers_column_ring<NW, WLO, WHI>(VP, VN, peq_lane + ((d >> (4u * k)) & 7u) * (4 * kBlock));
    }
  }
}

// the five match masks of pattern word `wi` into its ring slot
__device__ __forceinline__ void ring_store(uint32_t* __restrict__ peq_lane, int wi, const uint32_t (&mask)[5]) {
#pragma unroll
  for (int c = 0; c < 5; ++c) peq_lane[(c * 4 + (wi & 3)) * kBlock] = mask[c];
}

// PH = 3 (default): of the 32 columns of text group g the first 8 update words g-2 .. g+1, the middle 16
// words g-1 .. g+1 and the last 8 words g-1 .. g+2: word g-2 leaves a quarter of a group earlier and word g+2 enters a
// quarter later than with two phases of 16 columns (PH = 2, rounds 2-3: four words all the way).  3.5 words
// per column instead of 4; the guaranteed band is |j - i| <= 40 instead of 48, i.e. k = 80 - |delta| instead of 96 -
// |delta| (tools/band_model.py: banded_window_phase3 / banded_window_phase, both checked against plain Levenshtein by
// the CPU suite).  Config 2: 0.5 % of the demanded pairs lie above 75 edits (tools: the distances in the oracle's trace).
// Both are compiled (PH = 3 | 2): the engine falls back to the wide band for a while when more than an eighth of a
// batch's banded pairs left the narrow one (rpa_engine.hip: band4_wide), RPA_BAND4_PHASES=2|3 fixes the choice.
__host__ __device__ constexpr uint32_t band4_reach(int ph) { return ph == 3 ? 40u : 48u; }  // the band holds the diagonals |j - i| <= reach

template <int NW, int G, int PH>
struct BandGroups4 {
  static __device__ __forceinline__ void run(uint32_t (&VP)[NW], uint32_t (&VN)[NW], uint32_t* __restrict__ peq_lane, NtStream& ts,
                                             uint4& pending, NtStream& ps, uint32_t m, uint32_t n, uint32_t n_groups, int32_t& T,
                                             int32_t& top) {
    if ((uint32_t)G >= n_groups) return;
    constexpr int ALO = G - 2 > 0 ? G - 2 : 0, AHI = G + 1 < NW - 1 ? G + 1 : NW - 1;  // the group's first columns
    constexpr int BLO = G - 1 > 0 ? G - 1 : 0, BHI = G + 2 < NW - 1 ? G + 2 : NW - 1;  // its last columns
    constexpr uint32_t kFirst = PH == 3 ? 8u : 16u;   // columns before word G - 2 leaves
    constexpr uint32_t kLast = PH == 3 ? 24u : 16u;   // columns before word G + 2 enters
    const uint4 buf = ts.take32(pending);
    if ((uint32_t)G + 1u < n_groups) pending = ts.load4();
    const uint32_t cols = n - (uint32_t)G * 32u < 32u ? n - (uint32_t)G * 32u : 32u;
    // the pattern word the window takes in (word G + 2): asked for now, turned into masks when it enters
    constexpr bool takes_in = G + 2 <= NW - 1;
    uint4 pat_pending = make_uint4(0u, 0u, 0u, 0u);
    const bool has_word = takes_in && cols > kLast && (uint32_t)(G + 2) * 32u < m;
    if (has_word) pat_pending = ps.load4();
    const uint32_t ca = cols < kFirst ? cols : kFirst;
    if constexpr (PH == 3) band_sweep16<NW, ALO, AHI>(VP, VN, peq_lane, buf.x, 0u, ca);
    else band_sweep16<NW, ALO, AHI>(VP, VN, peq_lane, buf.x, buf.y, ca);
    T += (int32_t)ca;
    top = ALO;
    if (cols > kFirst) {  // columns follow: word G - 2 leaves the window
      if constexpr (G - 2 >= 0 && G - 2 <= NW - 1) T += (int32_t)__popc(VP[G - 2]) - (int32_t)__popc(VN[G - 2]);
      top = BLO;
      if constexpr (PH == 3) {  // the middle 16 columns: three words
        constexpr int MHI = G + 1 < NW - 1 ? G + 1 : NW - 1;
        const uint32_t cm = (cols < kLast ? cols : kLast) - kFirst;
        band_sweep16<NW, BLO, MHI>(VP, VN, peq_lane, buf.y, buf.z, cm);
        T += (int32_t)cm;
      }
      if (cols > kLast) {
        if constexpr (takes_in) {  // word G + 2 enters the ring in the slot word G - 2 has left
          uint32_t mask[5] = {0u, 0u, 0u, 0u, 0u};
          if (has_word) {
            const uint4 x = ps.take32(pat_pending);
            uint32_t p0, p1, pn;
            planes32(x, p0, p1, pn);
            const uint32_t rem = m - (uint32_t)(G + 2) * 32u;
            const uint32_t vm = rem >= 32u ? 0xFFFFFFFFu : ((1u << rem) - 1u);
            mask[4] = pn & vm; mask[3] = p1 & p0 & vm; mask[2] = p1 & ~p0 & vm; mask[1] = p0 & ~p1 & ~pn & vm; mask[0] = ~(p0 | p1 | pn) & vm;
          }
          ring_store(peq_lane, G + 2, mask);
        }
        if constexpr (PH == 3) band_sweep16<NW, BLO, BHI>(VP, VN, peq_lane, buf.w, 0u, cols - kLast);
        else band_sweep16<NW, BLO, BHI>(VP, VN, peq_lane, buf.z, buf.w, cols - kLast);
        T += (int32_t)(cols - kLast);
      }
    }
    if constexpr (G + 1 < NW + 2) BandGroups4<NW, G + 1, PH>::run(VP, VN, peq_lane, ts, pending, ps, m, n, n_groups, T, top);
  }
};

template <int NW, int PH>
__global__ __launch_bounds__(kBlock) void k_myers_short_band4(const uint2* __restrict__ list, uint32_t n_list,
                                                               const uint2* __restrict__ list2, uint32_t n_list2, WorkDev w,
                                                               uint2* __restrict__ retry, uint32_t* __restrict__ retry_n,
                                                               unsigned long long* prof_acc, uint32_t* __restrict__ retry_total) {
  // match masks of the four pattern words in the window, [symbol][word & 3][lane]: 20 KB per workgroup whatever NW is
  // (the flat table of the full kernel is 5 NW KB and holds the kernel at 4 waves per SIMD; this one leaves that to the
  // registers).  A word's masks are made when the window takes the word in.
  __shared__ uint32_t peq[5 * 4 * kBlock];
  const int tid = threadIdx.x;
  const uint32_t t = blockIdx.x * kBlock + tid;
  const bool active = t < n_list + n_list2;
  uint32_t dst = 0, m = 0, n = 0;
  SegView pat{}, txt{};
  uint2 entry = make_uint2(0u, 0u);
  if (active) {
    uint32_t a, b;
    entry = t < n_list ? list[t] : list2[t - n_list];
    resolve_pair(w, entry, a, b, dst);
    SegView va = seg_view(w, a), vb = seg_view(w, b);
    const bool a_pat = (va.len + 31u) / 32u == (vb.len + 31u) / 32u ? va.len >= vb.len : va.len < vb.len;  // as in k_myers_short
    if (a_pat) { txt = vb; pat = va; } else { txt = va; pat = vb; }
    n = txt.len; m = pat.len;
  }
  if (prof_acc) profile_add(prof_acc, active, n, m, band4_word_columns(NW, PH, n) * 32ull);
  if (!active) return;
  const uint32_t ad = n > m ? n - m : m - n;
  constexpr uint32_t kReach = band4_reach(PH);
  if (ad >= kReach) {  // no band of this shape holds the corridor
    retry[atomicAdd(retry_n, 1u)] = entry;
    if (retry_total) atomicAdd(retry_total, 1u);  // (the 9..12-word classes: what the engine's ladder of banded kernels reads)
    return;
  }
  NtStream ps;
  ps.init(pat, 0u);
#pragma unroll
  for (int wi = 0; wi < 2 && wi < NW; ++wi) {  // the first window: words 0 and 1
    uint32_t mask[5];
    peq_word(ps, (uint32_t)wi, m, mask);
    ring_store(peq + tid, wi, mask);
  }
  uint32_t VP[NW], VN[NW];
#pragma unroll
  for (int wi = 0; wi < NW; ++wi) { VP[wi] = 0xFFFFFFFFu; VN[wi] = 0u; }
  const uint32_t n_groups = (n + 31u) >> 5;  // <= NW + 2: n < m + kReach <= 32 NW + 48
  int32_t T = 0, top = 0;
  if (n_groups) {
    NtStream ts;
    ts.init(txt, 0u);
    uint4 pending = ts.load4();
    BandGroups4<NW, 0, PH>::run(VP, VN, peq + tid, ts, pending, ps, m, n, n_groups, T, top);
  }
  int32_t dist = T;
#pragma unroll
  for (int wi = 0; wi < NW; ++wi) {
    const uint32_t pos = (uint32_t)wi * 32u;
    const uint32_t vm = pos >= m ? 0u : (m - pos >= 32u ? 0xFFFFFFFFu : ((1u << (m - pos)) - 1u));
    if (wi >= top) dist += (int32_t)__popc(VP[wi] & vm) - (int32_t)__popc(VN[wi] & vm);
  }
  if ((uint32_t)dist <= 2u * kReach - ad) store_dna_result(w, dst, dist, n, m);
  else {
    retry[atomicAdd(retry_n, 1u)] = entry;
    if (retry_total) atomicAdd(retry_total, 1u);
  }
}

// ------------------------------------------------------------------------------------------------------------
// K1r (round 5): the out-of-phase window of k_myers_short_band4 for patterns of ANY length -- here 13 ... 64 words, 385 ... 2048 nt,
// which until then went straight to the register strips of K1s (55 T cell-equivalents/s at 400-500 nt where the one-strip kernels
// reach 240 T at 384) -- with a window of WW = 6 or 10 words and ONLY the window alive in registers.  Word w of the pattern lives
// in register w mod WW and its match masks in ring slot w mod WW, for as long as it is inside the window: of the 32 columns of
// text group g the first 16 update words g - WW/2 ... g + WW/2 - 1, then word g - WW/2 leaves (its vertical deltas go into the
// running score T, which follows the row above the window as in band4), word g + WW/2 enters in the same register and slot (all
// +1, its masks made from the pattern stream there and then), and the last 16 columns update words g - WW/2 + 1 ... g + WW/2.
// The order of the carry chain through the registers rotates by one per group: a uniform switch over the WW rotations, every one
// of them with compile-time register and slot indices (no register is ever indexed at run time).  Nothing clips the window at
// the pattern's end -- rows past the pattern match nothing and lie below row m -- so no bound on the pattern length is compiled
// in.  Guaranteed band |j - i| <= 16 (WW - 1): exact up to k = 32 (WW - 1) - |delta| edits (160 and 288); what lies beyond, and
// pairs whose lengths differ by 16 (WW - 1) and more, go onto the strip kernels' retry list.  tools/band_model.py:
// banded_window_ring (the same schedule, checked against plain Levenshtein by the CPU suite).
// ------------------------------------------------------------------------------------------------------------
template <int WW> __host__ __device__ constexpr uint32_t window_reach() { return 16u * (uint32_t)(WW - 1); }

// one text column over CNT words of the window, the chain starting at register T0 and wrapping round
template <int WW, int T0, int CNT>
__device__ __forceinline__ void window_column(uint32_t (&VP)[WW], uint32_t (&VN)[WW], const uint32_t* __restrict__ e) {
  static_assert(T0 >= 0 && T0 < WW && CNT >= 1 && CNT <= WW, "window");
  uint32_t carry = 0, hp_prev = 0x80000000u, hn_prev = 0;
  constexpr int kHead = CNT < WW - T0 ? CNT : WW - T0;  // registers T0 ... T0 + kHead - 1, then 0 ... CNT - kHead - 1
  auto eq = [&](int wi) { return e[wi * kBlock]; };
  myers_word_range<WW, T0, T0 + kHead - 1, true>(VP, VN, eq, carry, hp_prev, hn_prev);
  if constexpr (CNT > kHead) myers_word_range<WW, 0, CNT - kHead - 1, false>(VP, VN, eq, carry, hp_prev, hn_prev);
}
// up to 16 columns (the nibbles of d0, then d1), two per trip: the steady state compiles WW rotations of this loop, so it is kept
// small -- unrolled eight deep like band4's it would be 140 KB of code for WW = 10, a multiple of the instruction cache
template <int WW, int T0, int CNT>
__device__ __forceinline__ void window_sweep16(uint32_t (&VP)[WW], uint32_t (&VN)[WW], const uint32_t* __restrict__ peq_lane, uint32_t d0,
                                               uint32_t d1, uint32_t ncols) {
  uint64_t d = ((uint64_t)d1 << 32) | d0;
#pragma unroll 1
  for (uint32_t c = 0; c < ncols; c += 2u) {
    window_column<WW, T0, CNT>(VP, VN, peq_lane + ((uint32_t)d & 7u) * (uint32_t)(WW * kBlock));
    if (c + 1u < ncols) window_column<WW, T0, CNT>(VP, VN, peq_lane + ((uint32_t)(d >> 4) & 7u) * (uint32_t)(WW * kBlock));
    d >>= 8;
  }
}
// the five match masks of pattern word `wi` (the next 32 residues of the stream, or nothing past the pattern's end) into ring slot SLOT
template <int WW, int SLOT>
__device__ __forceinline__ void window_take_in(uint32_t* __restrict__ peq_lane, NtStream& ps, uint32_t wi, uint32_t m) {
  uint32_t mask[5];
  peq_word(ps, wi, m, mask);
#pragma unroll
  for (int c = 0; c < 5; ++c) peq_lane[(c * WW + SLOT) * kBlock] = mask[c];
}
// groups 0 ... WW/2 - 1: the window still starts at word 0 (register 0) and only grows
template <int WW, int G>
struct WindowHead {
  static __device__ __forceinline__ void run(uint32_t (&VP)[WW], uint32_t (&VN)[WW], uint32_t* __restrict__ peq_lane, NtStream& ts, uint4& pending,
                                             NtStream& ps, uint32_t m, uint32_t n, uint32_t n_groups, int32_t& T) {
    constexpr int H = WW / 2;
    if constexpr (G < H) {
      if ((uint32_t)G >= n_groups) return;
      const uint4 buf = ts.take32(pending);
      if ((uint32_t)G + 1u < n_groups) pending = ts.load4();
      const uint32_t cols = n - (uint32_t)G * 32u < 32u ? n - (uint32_t)G * 32u : 32u;
      const uint32_t ca = cols < 16u ? cols : 16u;
      window_sweep16<WW, 0, G + H>(VP, VN, peq_lane, buf.x, buf.y, ca);
      T += (int32_t)ca;
      if (cols > 16u) {
        window_take_in<WW, G + H>(peq_lane, ps, (uint32_t)(G + H), m);  // (register G + H is still in its initial state)
        window_sweep16<WW, 0, G + H + 1>(VP, VN, peq_lane, buf.z, buf.w, cols - 16u);
        T += (int32_t)(cols - 16u);
      }
      WindowHead<WW, G + 1>::run(VP, VN, peq_lane, ts, pending, ps, m, n, n_groups, T);
    }
  }
};
// a group of the steady state whose window starts in register T0 (T0 = (g - WW/2) mod WW); returns false when the text ended in its first half
template <int WW, int T0>
__device__ __forceinline__ bool window_group(uint32_t (&VP)[WW], uint32_t (&VN)[WW], uint32_t* __restrict__ peq_lane, const uint4 buf, uint32_t cols,
                                             NtStream& ps, uint32_t g, uint32_t m, int32_t& T) {
  const uint32_t ca = cols < 16u ? cols : 16u;
  window_sweep16<WW, T0, WW>(VP, VN, peq_lane, buf.x, buf.y, ca);
  T += (int32_t)ca;
  if (cols <= 16u) return false;
  T += (int32_t)__popc(VP[T0]) - (int32_t)__popc(VN[T0]);  // word g - WW/2 leaves: a whole word of the pattern (n < m + reach)
  VP[T0] = 0xFFFFFFFFu; VN[T0] = 0u;                        // word g + WW/2 enters
  window_take_in<WW, T0>(peq_lane, ps, g + (uint32_t)(WW / 2), m);
  window_sweep16<WW, (T0 + 1) % WW, WW>(VP, VN, peq_lane, buf.z, buf.w, cols - 16u);
  T += (int32_t)(cols - 16u);
  return true;
}
template <int WW, int T0>
struct WindowSteady {
  static __device__ __forceinline__ bool run(int t, uint32_t (&VP)[WW], uint32_t (&VN)[WW], uint32_t* __restrict__ peq_lane, const uint4 buf,
                                             uint32_t cols, NtStream& ps, uint32_t g, uint32_t m, int32_t& T) {
    if constexpr (T0 < WW) {
      if (t == T0) return window_group<WW, T0>(VP, VN, peq_lane, buf, cols, ps, g, m, T);
      return WindowSteady<WW, T0 + 1>::run(t, VP, VN, peq_lane, buf, cols, ps, g, m, T);
    } else return false;
  }
};
// word-columns the kernel sweeps for a text of n columns (its schedule restated, for rpa_profile.cells_updated)
__device__ __forceinline__ unsigned long long window_word_columns(int WW, uint32_t n) {
  unsigned long long wc = 0;
  const uint32_t H = (uint32_t)WW / 2u;
  for (uint32_t g = 0; g * 32u < n; ++g) {
    const uint32_t cols = n - g * 32u < 32u ? n - g * 32u : 32u, ca = cols < 16u ? cols : 16u;
    wc += (unsigned long long)ca * (g < H ? g + H : (uint32_t)WW) + (unsigned long long)(cols - ca) * (g < H ? g + H + 1u : (uint32_t)WW);
  }
  return wc;
}

template <int WW>
__global__ __launch_bounds__(kBlock) void k_myers_window(const uint2* __restrict__ list, uint32_t n_list, WorkDev w, uint2* __restrict__ retry,
                                                          uint32_t* __restrict__ retry_n, uint32_t* __restrict__ retry_total,
                                                          unsigned long long* prof_acc) {
  static_assert(WW % 2 == 0 && WW >= 4, "an even number of words");
  extern __shared__ uint32_t peq_ring[];  // [symbol][slot][lane]: 5 x WW KB per workgroup (30 / 50 / 70 KB)
  const int tid = threadIdx.x;
  const uint32_t t = blockIdx.x * kBlock + tid;
  const bool active = t < n_list;
  uint32_t dst = 0, m = 0, n = 0;
  SegView pat{}, txt{};
  uint2 entry = make_uint2(0u, 0u);
  if (active) {
    uint32_t a, b;
    entry = list[t];
    resolve_pair(w, entry, a, b, dst);
    SegView va = seg_view(w, a), vb = seg_view(w, b);
    if (va.len < vb.len) { txt = vb; pat = va; } else { txt = va; pat = vb; }  // the shorter one is the pattern (any choice is exact; the window costs WW words per column either way)
    n = txt.len; m = pat.len;
  }
  if (prof_acc) profile_add(prof_acc, active, n, m, window_word_columns(WW, n) * 32ull);
  if (!active) return;
  const uint32_t ad = n - m;
  constexpr uint32_t kReach = window_reach<WW>();
  if (ad >= kReach || m == 0u) {  // no band of this shape holds the corridor
    retry[atomicAdd(retry_n, 1u)] = entry;
    atomicAdd(retry_total, 1u);
    return;
  }
  uint32_t* __restrict__ peq_lane = peq_ring + tid;
  NtStream ps;
  ps.init(pat, 0u);
  constexpr int H = WW / 2;
#pragma unroll
  for (int wi = 0; wi < H; ++wi) {  // the first window: words 0 ... WW/2 - 1
    uint32_t mask[5];
    peq_word(ps, (uint32_t)wi, m, mask);
#pragma unroll
    for (int c = 0; c < 5; ++c) peq_lane[(c * WW + wi) * kBlock] = mask[c];
  }
  uint32_t VP[WW], VN[WW];
#pragma unroll
  for (int k = 0; k < WW; ++k) { VP[k] = 0xFFFFFFFFu; VN[k] = 0u; }
  const uint32_t n_groups = (n + 31u) >> 5;
  int32_t T = 0;
  uint32_t top_word = 0;  // first word of the window in effect
  NtStream ts;
  ts.init(txt, 0u);
  uint4 pending = ts.load4();
  WindowHead<WW, 0>::run(VP, VN, peq_lane, ts, pending, ps, m, n, n_groups, T);
  for (uint32_t g = (uint32_t)H; g < n_groups; ++g) {
    const uint4 buf = ts.take32(pending);
    if (g + 1u < n_groups) pending = ts.load4();
    const uint32_t cols = n - g * 32u < 32u ? n - g * 32u : 32u;
    const int rot = (int)((g - (uint32_t)H) % (uint32_t)WW);
    const bool moved = WindowSteady<WW, 0>::run(rot, VP, VN, peq_lane, buf, cols, ps, g, m, T);
    top_word = g - (uint32_t)H + (moved ? 1u : 0u);
  }
  // the vertical deltas of the window's words (register r holds word top_word + ((r - top_word) mod WW); before the window has
  // moved that is word r, the words it has not reached being in their initial state anyway) and of the rows below it, all +1
  int32_t dist = T;
  const uint32_t top_reg = top_word % (uint32_t)WW;
#pragma unroll
  for (int r = 0; r < WW; ++r) {
    const uint32_t off = ((uint32_t)r + (uint32_t)WW - top_reg) % (uint32_t)WW;
    const uint32_t pos = (top_word + off) * 32u;
    const uint32_t vm = pos >= m ? 0u : (m - pos >= 32u ? 0xFFFFFFFFu : ((1u << (m - pos)) - 1u));
    dist += (int32_t)__popc(VP[r] & vm) - (int32_t)__popc(VN[r] & vm);
  }
  const uint32_t below = (top_word + (uint32_t)WW) * 32u;
  if (m > below) dist += (int32_t)(m - below);
  if ((uint32_t)dist <= 2u * kReach - ad) store_dna_result(w, dst, dist, n, m);
  else {
    retry[atomicAdd(retry_n, 1u)] = entry;
    atomicAdd(retry_total, 1u);
  }
}

// ------------------------------------------------------------------------------------------------------------
// K1w: the same alignment with a pattern WORD per lane -- eight lanes per pair, eight pairs per wavefront -- for launches of a
// few thousand pairs (the sparse rounds at the end of a batch, the retry lists of the later dense rounds).  With a pair per
// lane such a launch takes what ONE lane takes, ~45 us for 200 columns x 8 words, however few pairs it carries; here lane l
// owns word l (VP, VN, its five match masks in LDS) and runs l + 1 columns behind lane 0, so a column's add carry and the top
// bits of HP / HN reach the word above one step later through ONE `v_mov_b32 ... row_shr:1` (DPP, no LDS, no memory), and so
// does the column's symbol -- a step ahead of the carries, so that the match mask of the next column is already on its way
// from LDS while the current one is computed.  n + 8 steps of one word update instead of n x 8: ~9 us.  Exact for any pattern
// of up to 256 residues, like k_myers_short<8>.  A retry list's length is known to the device only: both kernels are launched
// on it and `dev_split` decides which of them works (this one up to dev_split pairs, k_myers_short above).
// ------------------------------------------------------------------------------------------------------------
constexpr int kWaveWords = 8;  // lanes per pair = words of the widest pattern
__device__ __forceinline__ uint32_t row_shr1(uint32_t v) {
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111 /* row_shr:1 */, 0xf, 0xf, false);
}
__global__ __launch_bounds__(kBlock) void k_myers_short_wave(const uint2* __restrict__ list, uint32_t n_list, const uint32_t* __restrict__ n_list_dev,
                                                              uint32_t dev_split, const uint2* __restrict__ list2, uint32_t n_list2, WorkDev w,
                                                              unsigned long long* prof_acc) {
  __shared__ uint32_t peq[6 * kBlock];  // [symbol][thread]; row 5 = no match (the symbol of the steps before and behind the text)
  const int tid = threadIdx.x;
  const uint32_t sub = (uint32_t)tid & (kWaveWords - 1), group = (uint32_t)tid / kWaveWords;
  constexpr uint32_t kPairs = kBlock / kWaveWords;  // pairs per workgroup and trip
  if (n_list_dev) {
    const uint32_t cnt = *n_list_dev;
    if (cnt > dev_split) return;  // (whole grid: the pair-per-lane kernel behind this launch takes the list)
    n_list = cnt < n_list ? cnt : n_list;
    if (blockIdx.x == 0 && tid == 0 && n_list) atomicAdd(&w.counters[kCtrRetryShortTotal], n_list);  // what the band cost this batch
  }
  const uint32_t total = n_list + n_list2;
  peq[5 * kBlock + tid] = 0u;
  for (uint32_t base = blockIdx.x * kPairs; base < total; base += gridDim.x * kPairs) {
    const uint32_t t = base + group;
    const bool active = t < total;
    uint32_t dst = 0, m = 0, n = 0;
    SegView pat{}, txt{};
    if (active) {
      uint32_t a, b;
      resolve_pair(w, t < n_list ? list[t] : list2[t - n_list], a, b, dst);
      SegView va = seg_view(w, a), vb = seg_view(w, b);
      const bool a_pat = (va.len + 31u) / 32u == (vb.len + 31u) / 32u ? va.len >= vb.len : va.len < vb.len;  // as in k_myers_short
      if (a_pat) { txt = vb; pat = va; } else { txt = va; pat = vb; }
      n = txt.len; m = pat.len;
    }
    if (prof_acc) profile_add(prof_acc, active && sub == 0u, n, m, (unsigned long long)n * (kWaveWords * 32u));
    {
      uint32_t mask[5] = {0u, 0u, 0u, 0u, 0u};
      if (active && sub * 32u < m) {
        NtStream ps;
        ps.init(pat, sub * 32u);
        peq_word(ps, sub, m, mask);
      }
#pragma unroll
      for (int c = 0; c < 5; ++c) peq[c * kBlock + tid] = mask[c];
    }
    // (no barrier: a lane only ever reads its own LDS column)
    uint32_t VP = 0xFFFFFFFFu, VN = 0u;
    uint32_t sym_now = 5u, eq_now = 0u, out_bits = 0u;  // out_bits: add carry | HP's top bit << 1 | HN's top bit << 2 of my last column
    uint32_t step = 0;
    auto advance = [&](uint32_t c_next) {  // c_next: the symbol of column `step` (lane 0 computes it in the NEXT step), 5 behind the text
      // the symbol of my next column is the one the word below computes in this step: its match mask is asked for first and
      // arrives from LDS while this step's column is computed
      uint32_t sym_next = row_shr1(sym_now);
      sym_next = sub == 0u ? c_next : sym_next;
      const uint32_t eq_next = peq[sym_next * kBlock + tid];
      // the carries of the word below for the column I compute now (lane 0: +1 enters row 0 of a global alignment)
      uint32_t in_bits = row_shr1(out_bits);
      in_bits = sub == 0u ? 2u : in_bits;
      const uint32_t X = eq_now | VN;
      const uint64_t s64 = (uint64_t)VP + (uint64_t)(X & VP) + (uint64_t)(in_bits & 1u);
      const uint32_t sum = (uint32_t)s64;
      const uint32_t D0 = __builtin_amdgcn_bitop3_b32(sum, VP, X, 0xBE);   // (sum ^ VP) | X
      const uint32_t HN = VP & D0;
      const uint32_t HP = __builtin_amdgcn_bitop3_b32(VN, VP, D0, 0xF1);   // VN | ~(VP | D0)
      const uint32_t Xs = (HP << 1) | ((in_bits >> 1) & 1u), Hs = (HN << 1) | (in_bits >> 2);
      const bool on = step - sub - 1u < n;  // my column this step: step - sub - 1 (unsigned: not before the text, not behind it)
      VN = on ? (Xs & D0) : VN;
      VP = on ? __builtin_amdgcn_bitop3_b32(Hs, Xs, D0, 0xF1) : VP;       // Hs | ~(Xs | D0)
      out_bits = (uint32_t)(s64 >> 32) | ((HP >> 31) << 1) | ((HN >> 31) << 2);
      sym_now = sym_next;
      eq_now = eq_next;
      ++step;
    };
    if (active) walk_text(txt, [&](uint32_t c) { advance(c); });
    // (every lane of a wavefront must take every step: the text lengths of its eight pairs differ -- the shorter ones idle
    // through no-match columns; an inactive group has n = 0)
    {
      uint32_t longest = n;
#pragma unroll
      for (int d = 32; d >= kWaveWords; d >>= 1) longest = max(longest, (uint32_t)__shfl_xor(longest, d, 64));
      while (step < longest + kWaveWords) advance(5u);
    }
    int32_t part = 0;
    {
      const uint32_t pos = sub * 32u;
      const uint32_t vm = pos >= m ? 0u : (m - pos >= 32u ? 0xFFFFFFFFu : ((1u << (m - pos)) - 1u));
      part = (int32_t)__popc(VP & vm) - (int32_t)__popc(VN & vm);
    }
#pragma unroll
    for (int d = 1; d < kWaveWords; d <<= 1) part += __shfl_xor(part, d, 64);
    if (active && sub == 0u) store_dna_result(w, dst, (int32_t)n + part, n, m);
  }
}

hipError_t launch_align_dna_short_wave(hipStream_t st, const uint2* list, uint32_t n, const uint32_t* n_dev, uint32_t dev_split,
                                       const uint2* list2, uint32_t n2, const WorkDev& w, unsigned long long* prof_acc) {
  if ((uint64_t)n + n2 == 0) return hipSuccess;
  if ((uint64_t)n + n2 > 0xFFFFFF00ull) return hipErrorInvalidValue;
  const uint64_t bound = n_dev ? (uint64_t)(n < dev_split ? n : dev_split) + n2 : (uint64_t)n + n2;
  constexpr uint32_t kPairs = kBlock / kWaveWords;
  uint32_t blocks = (uint32_t)((bound + kPairs - 1) / kPairs);
  if (blocks > 8192u) blocks = 8192u;  // (the grid strides)
  if (blocks == 0) blocks = 1;
  hipLaunchKernelGGL(k_myers_short_wave, dim3(blocks), dim3(kBlock), 0, st, list, n, n_dev, dev_split, list2, n2, w, prof_acc);
  return hipGetLastError();
}

hipError_t launch_align_dna_short(hipStream_t st, int words, const uint2* list, uint32_t n, const uint32_t* n_dev, uint32_t grid_blocks,
                                  const uint2* list2, uint32_t n2, const WorkDev& w, unsigned long long* prof_acc, uint32_t dev_split) {
  if ((uint64_t)n + n2 == 0) return hipSuccess;
  if ((uint64_t)n + n2 > 0xFFFFFF00ull) return hipErrorInvalidValue;
  uint32_t blocks = (n + n2 + kBlock - 1) / kBlock;
  if (n_dev && grid_blocks && blocks > grid_blocks) blocks = grid_blocks;
  size_t lds = (size_t)5 * words * kBlock * sizeof(uint32_t);
  static const size_t lds_pad = getenv("RPA_LDS_PAD") ? (size_t)atoi(getenv("RPA_LDS_PAD")) : 0;  // occupancy experiments only
  lds += lds_pad;
#define RPA_LAUNCH_MYERS(NWV) \
  case NWV: hipLaunchKernelGGL(k_myers_short<NWV>, dim3(blocks), dim3(kBlock), lds, st, list, n, n_dev, dev_split, list2, n2, w, prof_acc); break;
  switch (words) {
    RPA_LAUNCH_MYERS(1) RPA_LAUNCH_MYERS(2) RPA_LAUNCH_MYERS(3) RPA_LAUNCH_MYERS(4)
    RPA_LAUNCH_MYERS(5) RPA_LAUNCH_MYERS(6) RPA_LAUNCH_MYERS(7) RPA_LAUNCH_MYERS(8)
    RPA_LAUNCH_MYERS(9) RPA_LAUNCH_MYERS(10) RPA_LAUNCH_MYERS(11) RPA_LAUNCH_MYERS(12)
    default: return hipErrorInvalidValue;
  }
#undef RPA_LAUNCH_MYERS
  return hipGetLastError();
}

bool dna_short_band_available(int words) { return words >= 5 && words <= kShortWords; }

hipError_t launch_align_dna_short_band(hipStream_t st, int words, const uint2* list, uint32_t n, const uint2* list2, uint32_t n2,
                                       const WorkDev& w, uint2* retry, uint32_t* retry_n, bool shared_gpu, bool wide, unsigned long long* prof_acc,
                                       uint32_t* retry_total) {
  if ((uint64_t)n + n2 == 0) return hipSuccess;
  if ((uint64_t)n + n2 > 0xFFFFFF00ull || !retry || !retry_n) return hipErrorInvalidValue;
  const uint32_t blocks = (n + n2 + kBlock - 1) / kBlock;
  const size_t lds = (size_t)5 * words * kBlock * sizeof(uint32_t);
  const char* bw = getenv("RPA_BAND_WORDS");  // 5: the in-phase five-word window (A/B, tests)
  const bool five = bw && atoi(bw) == 5;
  // Occupancy of the ring kernel: 20 KB of LDS and 73-79 VGPRs per wavefront give six wavefronts per SIMD, which is worth
  // 2 % to a batch that has the GPU to itself and leaves no registers for anybody else.  A job slot (several batches in
  // flight) asks for 20 KB more: four wavefronts per SIMD (config 2, three steps in flight, round 2: 10.17 -> 9.76 ms per step;
  // one synchronous batch: 11.5 -> 11.7).  Not because the other jobs' placement kernels then sit beside them -- four workgroups
  // of 40 KB fill a CU's LDS -- and on round 4's engine 8, 14.5 (LDS left for k_rank_records), 20 and 26 KB (three wavefronts,
  // room for k_decide / k_emit) all give the same step: profiles/r04_ab_decide_occupancy.txt.
  const char* padv = getenv("RPA_BAND4_LDS_PAD");  // (experiments: extra dynamic LDS per workgroup in bytes, whoever launches)
  const size_t pad4 = padv ? (size_t)atoi(padv) : (shared_gpu ? 20000u : 0u);
#define RPA_LAUNCH_MYERS_BAND(NWV) \
  case NWV: \
    if (five) hipLaunchKernelGGL((k_myers_short_band<NWV, 2>), dim3(blocks), dim3(kBlock), lds, st, list, n, list2, n2, w, retry, retry_n, prof_acc); \
    else if (wide) hipLaunchKernelGGL((k_myers_short_band4<NWV, 2>), dim3(blocks), dim3(kBlock), pad4, st, list, n, list2, n2, w, retry, retry_n, prof_acc, retry_total); \
    else hipLaunchKernelGGL((k_myers_short_band4<NWV, 3>), dim3(blocks), dim3(kBlock), pad4, st, list, n, list2, n2, w, retry, retry_n, prof_acc, retry_total); \
    break;
  switch (words) {
    RPA_LAUNCH_MYERS_BAND(5) RPA_LAUNCH_MYERS_BAND(6) RPA_LAUNCH_MYERS_BAND(7) RPA_LAUNCH_MYERS_BAND(8)
    RPA_LAUNCH_MYERS_BAND(9) RPA_LAUNCH_MYERS_BAND(10) RPA_LAUNCH_MYERS_BAND(11) RPA_LAUNCH_MYERS_BAND(12)
    default: return hipErrorInvalidValue;
  }
#undef RPA_LAUNCH_MYERS_BAND
  return hipGetLastError();
}

bool dna_window_available(int window_words) { return window_words == 6 || window_words == 8 || window_words == 10 || window_words == 14; }

hipError_t launch_align_dna_window(hipStream_t st, int window_words, const uint2* list, uint32_t n, const WorkDev& w, uint2* retry,
                                   uint32_t* retry_n, uint32_t* retry_total, unsigned long long* prof_acc) {
  if (n == 0) return hipSuccess;
  if (n > 0xFFFFFF00u || !retry || !retry_n || !retry_total) return hipErrorInvalidValue;
  const uint32_t blocks = (n + kBlock - 1) / kBlock;
  const size_t lds = (size_t)5 * window_words * kBlock * sizeof(uint32_t);
  switch (window_words) {
    case 6: hipLaunchKernelGGL(k_myers_window<6>, dim3(blocks), dim3(kBlock), lds, st, list, n, w, retry, retry_n, retry_total, prof_acc); break;
    case 8: hipLaunchKernelGGL(k_myers_window<8>, dim3(blocks), dim3(kBlock), lds, st, list, n, w, retry, retry_n, retry_total, prof_acc); break;
    case 10: hipLaunchKernelGGL(k_myers_window<10>, dim3(blocks), dim3(kBlock), lds, st, list, n, w, retry, retry_n, retry_total, prof_acc); break;
    case 14: {  // 70 KB of ring: beyond what a launch may ask for without saying so
      // (per launch, not once: the attribute belongs to the function ON THE CURRENT DEVICE, and a process may drive several -- taxator --devices)
      const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(k_myers_window<14>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (attr != hipSuccess) return attr;
      hipLaunchKernelGGL(k_myers_window<14>, dim3(blocks), dim3(kBlock), lds, st, list, n, w, retry, retry_n, retry_total, prof_acc);
      break;
    }
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------------------
// K1s: pattern longer than 256 nt, many pairs -- still ONE ALIGNMENT PER LANE, the pattern cut into register strips,
// and only the cells inside Ukkonen's band are computed.
//
// Strips.  The delta vectors of a strip of NW words (NW in 5..8, chosen per pair to waste the fewest rows) live in
// VGPRs and the strip's match masks in LDS exactly as in K1; the lane sweeps the text once per strip.  What a strip
// hands to the one below is, per text column, the HP and HN bits of its last row -- two bit planes of 32 columns in an
// HBM scratch laid out [group of 32 columns][plane][thread] (a wave reads/writes 256 contiguous bytes per plane and
// group).  The add carry into the strip below needs no plane of its own: the carry out of a word's top bit is set
// exactly when vp & D0 is at that bit, i.e. it IS the HN bit (tools/band_model.py asserts it on every column).
// Rows past the end of the pattern (the last strip is rarely full) match nothing, which leaves rows <= m untouched;
// the score is tracked at the static bottom row of the strip and corrected at the end by the vertical deltas of those
// phantom rows: D[m][n] = D[bottom][n] - (popc(VP & tail) - popc(VN & tail)).
//
// Band.  An alignment of cost <= k of a pattern of m rows with a text of n >= m columns never leaves the diagonals
// -h <= j - i <= (n - m) + h, h = (k - (n - m)) / 2 (Ukkonen 1985: a cell on diagonal e costs at least |e| to reach
// and |n - m - e| to leave).  A strip therefore sweeps only the column groups that band touches: it starts at its
// first group from VP = all ones on top of the value the strip above had in that column (an upper bound of the true
// column), and where the strip above never got to, "+1 per column" comes down (again an upper bound).  Every value
// D' computed this way is >= the true D, and equal to it along any path that stays inside the band, so
//     D'[m][n] <= k  =>  D'[m][n] is the distance;      D'[m][n] > k  =>  the distance is > k.
// A pair that fails the test goes onto a retry list and is re-run with the band wide open (by K2 when the pattern fits
// a wavefront, else by this kernel): results are those of the unbanded recurrence the reference runs
// (global_alignment_myers_impl.h:129-191), bit for bit, whatever k was.  k comes from a probe: the first strip over
// the first columns gives min_j D[HS][j], the edits the first HS rows need; scaled to m rows with a 31 % margin.  h is
// made uniform over the wavefront (its lanes run the same group loop anyway), so the planes stay coalesced.
// With the 10 kbp pairs of config 5 (distances 0.21-0.29 n) a strip sweeps about a third of the columns.
// Same bound as K1 (integer VALU issue); per-column strip overhead ~7 instructions of ~90.  Any pattern length fits.
// ------------------------------------------------------------------------------------------------------------
constexpr uint32_t kProbeMarginDen = 16;  // k = probe rate x m x margin/16 + slack (margin: launcher argument, 21 by default)
constexpr uint32_t kProbeSlack = 48;

template <int NW>
__global__ __launch_bounds__(kBlock) void k_myers_band(const uint2* __restrict__ list, uint32_t n_list,
                                                        const uint32_t* __restrict__ n_list_dev, WorkDev w,
                                                        uint32_t* __restrict__ scratch, uint32_t stride, uint32_t wide_open,
                                                        uint32_t margin, uint2* __restrict__ retry, uint32_t* __restrict__ retry_n,
                                                        unsigned long long* prof_acc) {
  extern __shared__ uint32_t peq[];  // [5][NW][kBlock]
  constexpr uint32_t HS = NW * 32u;
  const int tid = threadIdx.x;
  const uint32_t total = n_list_dev ? (*n_list_dev < n_list ? *n_list_dev : n_list) : n_list;
  if (n_list_dev && (*n_list_dev <= w.dev_count_lo || (w.dev_count_hi && *n_list_dev > w.dev_count_hi))) return;  // (another kernel's share of this list)
  if (n_list_dev && blockIdx.x == 0 && tid == 0 && total) atomicAdd(&w.counters[kCtrRetryTotal], total);
  uint32_t* __restrict__ sc = scratch + (blockIdx.x * kBlock + tid);  // boundary planes: group g, plane p at sc[(2g + p) * stride]
  unsigned long long upd_cells = 0;  // (profile) cells this lane updated: NW x 32 rows x 32 columns per group a strip swept
  for (uint32_t base = blockIdx.x * kBlock; base < total; base += gridDim.x * kBlock) {  // retry launches are sized before the count is known
    const uint32_t t = base + tid;
    const bool active = t < total;
    uint32_t dst = 0, m = 0, n = 0;
    SegView pat{}, txt{};
    uint2 entry = make_uint2(0u, 0u);
    if (active) {
      uint32_t a, b;
      entry = list[t];
      resolve_pair(w, entry, a, b, dst);
      // Segment B is the text, whichever is longer (the distance is symmetric and the cost is rows x columns either way):
      // in a placement round B is the anchor of the record set, i.e. the SAME segment for ~50 consecutive lanes, and the
      // lanes of a wavefront sweep the same column groups -- their text loads hit the same 16 bytes.
      pat = seg_view(w, a); txt = seg_view(w, b);
      n = txt.len; m = pat.len;
    }
    if (prof_acc) profile_add(prof_acc, active, n, m, 0ull);  // (updated cells: added up strip by strip, profile_add_updated)
    if (!active) continue;  // no barrier below: a lane only ever touches its own LDS column

    const uint32_t words = (m + 31u) / 32u;
    const uint32_t nstrips = (words + NW - 1u) / NW;
    const uint32_t n_groups = (n + 31u) >> 5;
    const int64_t delta = (int64_t)n - (int64_t)m, dlo = delta < 0 ? delta : 0, dhi = delta > 0 ? delta : 0;  // rows may outnumber columns
    const uint32_t h_open = n > m ? n : m;
    // band half width beyond the corridor [min(0, delta), max(0, delta)] (h_open = wide open) of the strip above / this strip / the strip below:
    // the estimate is refreshed after every strip (more rows seen, less margin needed) and takes effect two strips on,
    // because a strip must know where the strip below will start while it sweeps
    uint32_t h_prev = h_open, h_cur = h_open, h_next = h_open, h_min = h_open;
    uint32_t probe_min = 0xFFFFFFFFu;
    int32_t bottom = 0, next_top = 0;
    uint32_t VP[NW], VN[NW];
    // it = -1: the probe (strip 0 over the first columns, nothing stored); it >= 0: strip `it`
    for (int32_t it = wide_open ? 0 : -1; it < (int32_t)nstrips; ++it) {
      const bool probe = it < 0;
      const uint32_t s = probe ? 0u : (uint32_t)it;
      const bool first = s == 0, last = !probe && s + 1u == nstrips;
      const uint32_t wbase = s * NW, R = wbase * 32u;
      NtStream ps;  // the strip's part of the pattern, 16 bytes per word (NtStream, rpa_device.h)
      ps.init(pat, R);
#pragma unroll
      for (int wi = 0; wi < NW; ++wi) {
        uint32_t mask[5];
        peq_word(ps, wbase + wi, m, mask);
#pragma unroll
        for (int c = 0; c < 5; ++c) peq[(c * NW + wi) * kBlock + tid] = mask[c];
        VP[wi] = 0xFFFFFFFFu;
        VN[wi] = 0u;
      }
      // column groups this strip sweeps: [g_lo, g_hi); the strip below starts at g_next; the strip above ended at ga_hi
      uint32_t g_lo, g_hi, g_next = 0xFFFFFFFFu, ga_hi = 0u;
      if (probe) {
        g_lo = 0u;
        g_hi = (3u * NW) / 2u < n_groups ? (3u * NW) / 2u : n_groups;  // 1.5 x the strip's height in columns
      } else {
        const int64_t lo = (int64_t)R + dlo - (int64_t)h_cur, hi = (int64_t)R + HS + dhi + (int64_t)h_cur - 1;
        g_lo = lo > 0 ? (uint32_t)(lo >> 5) : 0u;
        g_hi = (hi >> 5) + 1 < (int64_t)n_groups ? (uint32_t)(hi >> 5) + 1u : n_groups;
        if (!last) { const int64_t nlo = (int64_t)R + HS + dlo - (int64_t)h_next; g_next = nlo > 0 ? (uint32_t)(nlo >> 5) : 0u; }
        if (!first) {
          const int64_t ahi = (int64_t)R + dhi + (int64_t)h_prev - 1;
          ga_hi = ahi < 0 ? 1u : ((ahi >> 5) + 1 < (int64_t)n_groups ? (uint32_t)(ahi >> 5) + 1u : n_groups);
        }
        h_min = h_cur < h_min ? h_cur : h_min;
      }
      probe_min = 0xFFFFFFFFu;
      bottom = (first ? 0 : next_top) + (int32_t)HS;  // D'[bottom row of the strip][32 g_lo]: VP = all ones below the row above
      if (g_next == g_lo) next_top = bottom;

      uint32_t Pout = 0, Nout = 0;
      auto column = [&](uint32_t c, uint32_t hp_prev, uint32_t hn_prev) {
        const uint32_t* __restrict__ e = peq + c * (NW * kBlock) + tid;
        uint32_t carry = hn_prev >> 31;  // the add carry out of the strip above == its HN bit
        myers_word_range<NW, 0, NW - 1, false, (RPA_PAIR_SHIFT_STRIPS != 0)>(VP, VN, [&](int wi) { return e[wi * kBlock]; }, carry, hp_prev, hn_prev);  // (leaves the last word's HP / HN)
        Pout = __builtin_amdgcn_alignbit(Pout, hp_prev, 31);  // (Pout << 1) | (HP >> 31): column k of a group ends up at bit 31-k
        Nout = __builtin_amdgcn_alignbit(Nout, hn_prev, 31);
      };

      // Per 32 columns: one 16-byte load of text and two dword loads of the planes of the strip above (all asked for a
      // group ahead), two dword stores of this strip's planes.  (Memory operations retire in order on gfx9: with a load
      // and a store every four columns each load waited for the previous store to reach HBM.)
      if (g_lo < g_hi) {
        upd_cells += (unsigned long long)(g_hi - g_lo) * (32u * HS);
        NtStream ts;
        ts.init(txt, g_lo * 32u);
        uint4 pending = ts.load4();
        // no strip above, or it never got to these columns: +1 enters every column (HP 1, HN 0)
        uint32_t Pn = 0xFFFFFFFFu, Nn = 0u;
        if (g_lo < ga_hi) { Pn = sc[(size_t)(2u * g_lo) * stride]; Nn = sc[(size_t)(2u * g_lo + 1u) * stride]; }
        for (uint32_t g = g_lo; g < g_hi; ++g) {
          const uint4 buf = ts.take32(pending);
          if (g + 1u < g_hi) pending = ts.load4();
          uint32_t Pin = Pn, Nin = Nn;
          Pn = 0xFFFFFFFFu; Nn = 0u;
          if (g + 1u < ga_hi && g + 1u < g_hi) {
            const uint32_t* __restrict__ nx = sc + (size_t)(2u * (g + 1u)) * stride;
            Pn = nx[0]; Nn = nx[stride];
          }
          const uint32_t cols = n - g * 32u;
          Pout = 0u; Nout = 0u;
#pragma unroll 1
          for (uint32_t q = 0; q < 4u && q * 8u < cols; ++q) {  // eight columns: one dword of text, one byte of each plane
            const uint32_t d = q == 0u ? buf.x : q == 1u ? buf.y : q == 2u ? buf.z : buf.w;
            const uint32_t left = cols - q * 8u;
            if (left >= 8u) {
#pragma unroll
              for (int k = 0; k < 8; ++k) column((d >> (4 * k)) & 7u, Pin << k, Nin << k);
              Pin <<= 8; Nin <<= 8;
            } else {
#pragma unroll 1
              for (uint32_t k = 0; k < left; ++k) {
                column((d >> (4u * k)) & 7u, Pin, Nin);
                Pin <<= 1; Nin <<= 1;
              }
            }
          }
          bottom += (int32_t)__popc(Pout) - (int32_t)__popc(Nout);  // D'[bottom row][j] - D'[bottom row][j-1] summed over the group
          if (g + 1u == g_next) next_top = bottom;                  // what the strip below starts on top of
          probe_min = (uint32_t)bottom < probe_min ? (uint32_t)bottom : probe_min;
          if (cols < 32u) { const uint32_t sh = 32u - cols; Pout <<= sh; Nout <<= sh; }
          if (!last && !probe && g >= g_next) {  // the strip below starts at g_next: nobody reads the groups before it
            uint32_t* __restrict__ o = sc + (size_t)(2u * g) * stride;
            o[0] = Pout; o[stride] = Nout;
          }
        }
      }
      if (!wide_open) {
        // The cheapest way to have consumed the rows seen so far (min over the swept columns of the strip's bottom row)
        // -> a budget for all m rows, with a margin that shrinks as the square root of the rows seen (margin / 16 after
        // one strip); then ONE half width for the whole wavefront (its lanes run the same group loop anyway, and the
        // planes stay coalesced).  D' >= D: a band that was too narrow can only raise the estimate.
        const uint32_t rows = R + HS;
        const float f = 1.0f + (float)((int32_t)margin - (int32_t)kProbeMarginDen) * (1.0f / kProbeMarginDen) * __builtin_sqrtf((float)HS / (float)rows);
        const float est = (float)probe_min * ((float)m / (float)rows) * f + (float)kProbeSlack;
        const uint32_t mine = est * 0.5f < (float)h_open ? (uint32_t)(est * 0.5f) : h_open;
        uint32_t hmax = 0u;  // the lanes that are here are on the same strip: scalar loop over them
        for (unsigned long long mk = __ballot(1); mk; mk &= mk - 1ull) {
          const uint32_t o = (uint32_t)__builtin_amdgcn_readlane((int)mine, __ffsll((long long)mk) - 1);
          hmax = o > hmax ? o : hmax;
        }
        if (probe) { h_prev = hmax; h_cur = hmax; h_next = hmax; }
        else { h_prev = h_cur; h_cur = h_next; h_next = hmax < h_cur + HS ? hmax : h_cur + HS; }  // the strip below never starts left of this one
      }
    }
    // phantom rows below the pattern in the last strip: take their vertical deltas off the bottom score
    int32_t dist = bottom;
#pragma unroll
    for (int wi = 0; wi < NW; ++wi) {
      const uint32_t pos = ((nstrips - 1u) * NW + wi) * 32u;
      const uint32_t tail = pos >= m ? 0xFFFFFFFFu : (m - pos >= 32u ? 0u : ~((1u << (m - pos)) - 1u));
      dist -= (int32_t)__popc(VP[wi] & tail) - (int32_t)__popc(VN[wi] & tail);
    }
    const uint64_t k = 2ull * h_min + (uint64_t)(delta < 0 ? -delta : delta);
    if ((uint64_t)(uint32_t)dist <= k) store_dna_result(w, dst, dist, n, m);  // inside the band: this IS the distance (always, when the band is wide open)
    else retry[atomicAdd(retry_n, 1u)] = entry;
  }
  if (prof_acc) profile_add_updated(prof_acc, upd_cells);
}

// K1s, two lanes per pair.  Lanes 2p and 2p+1 of a wavefront work on the same alignment: in round t the even lane ("A")
// sweeps strip 2t, the odd lane ("B") strip 2t+1, ONE column group behind.  What A's strip hands down travels lane to
// lane through the wavefront's crossbar (the planes of group g are produced in one iteration and consumed by B in the next);
// only what B hands to the A lane of the next round goes through the HBM scratch -- half the boundary traffic of one
// lane per pair -- and a launch has twice as many lanes, i.e. half the granularity of the last round of the chip.  B idles
// while A sweeps the ~8 groups by which its window starts later, A while B finishes its window (~8 % of a round).
// Band, probe, retry: as in k_myers_band; the two strips of a round share one half width h (refreshed after every round,
// effective two rounds on), which the exactness argument covers (any h per strip that grows by at most HS from strip to strip).
template <int NW>
__global__ __launch_bounds__(kBlock, 4) void k_myers_band_duo(const uint2* __restrict__ list, uint32_t n_list,
                                                            const uint32_t* __restrict__ n_list_dev, WorkDev w,
                                                            uint32_t* __restrict__ scratch, uint32_t stride, uint32_t wide_open,
                                                            uint32_t margin, uint2* __restrict__ retry, uint32_t* __restrict__ retry_n,
                                                            unsigned long long* prof_acc) {
  extern __shared__ uint32_t peq[];  // [5][NW][kBlock]: every lane the masks of its own strip
  constexpr uint32_t HS = NW * 32u;
  constexpr uint32_t kPairs = kBlock / 2;
  const int tid = threadIdx.x;
  const uint32_t lag = (uint32_t)tid & 1u;  // 0: lane A (upper strip of a round), 1: lane B (lower strip, one group behind)
  const uint32_t total = n_list_dev ? (*n_list_dev < n_list ? *n_list_dev : n_list) : n_list;
  if (n_list_dev && (*n_list_dev <= w.dev_count_lo || (w.dev_count_hi && *n_list_dev > w.dev_count_hi))) return;  // (another kernel's share of this list)
  if (n_list_dev && blockIdx.x == 0 && tid == 0 && total) atomicAdd(&w.counters[kCtrRetryTotal], total);
  uint32_t* __restrict__ sc = scratch + (blockIdx.x * kPairs + ((uint32_t)tid >> 1));  // per PAIR: group g, plane p at sc[(2g + p) * stride]
  unsigned long long upd_cells = 0;  // (profile) cells this lane updated: NW x 32 rows x 32 columns per group it swept
  for (uint32_t base = blockIdx.x * kPairs; base < total; base += gridDim.x * kPairs) {
    const uint32_t t = base + ((uint32_t)tid >> 1);
    const bool active = t < total;
    uint32_t m = 0, n = 0;
    SegView pat{}, txt{};
    if (active) {
      uint32_t a, b, dst0;
      resolve_pair(w, list[t], a, b, dst0);
      pat = seg_view(w, a); txt = seg_view(w, b);  // B = the record set's anchor: the same text for ~50 consecutive pairs
      n = txt.len; m = pat.len;
    }
    if (prof_acc) profile_add(prof_acc, active && lag == 0u, n, m, 0ull);  // (updated cells: added up strip by strip, profile_add_updated)
    if (!active) continue;  // both lanes of a pair leave together; no barrier below

    const uint32_t words = (m + 31u) / 32u;
    const uint32_t nstrips = (words + NW - 1u) / NW;
    const uint32_t nrounds = (nstrips + 1u) / 2u;
    const uint32_t n_groups = (n + 31u) >> 5;
    // (32-bit arithmetic: segments are shorter than 2^29 residues, kLiteLenMask)
    const int32_t delta = (int32_t)n - (int32_t)m, dlo = delta < 0 ? delta : 0, dhi = delta > 0 ? delta : 0;
    const uint32_t h_open = n > m ? n : m;
    uint32_t h_prev = h_open, h_cur = h_open, h_next = h_open, h_min = h_open;  // per ROUND (see k_myers_band)
    int32_t bottom = 0, round_top = 0;  // round_top: what this round's A strip starts on top of (A lanes)
    uint32_t VP[NW], VN[NW];
    // first column group of strip `s` under half width h, and one past its last
    auto win_lo = [&](uint32_t s, uint32_t h) -> uint32_t { const int32_t lo = (int32_t)(s * HS) + dlo - (int32_t)h; return lo > 0 ? (uint32_t)lo >> 5 : 0u; };
    auto win_hi = [&](uint32_t s, uint32_t h) -> uint32_t {
      const int32_t hi = (int32_t)(s * HS + HS) + dhi + (int32_t)h - 1;
      return hi < 0 ? 0u : (((uint32_t)hi >> 5) + 1u < n_groups ? ((uint32_t)hi >> 5) + 1u : n_groups);
    };
    // rnd = -1: the probe (strip 0 by the A lane over the first columns, nothing stored); rnd >= 0: strips 2 rnd, 2 rnd + 1
    for (int32_t rnd = wide_open ? 0 : -1; rnd < (int32_t)nrounds; ++rnd) {
      const bool probe = rnd < 0;
      const uint32_t s = probe ? 0u : 2u * (uint32_t)rnd + lag;  // my strip
      const bool mine = probe ? lag == 0u : s < nstrips;         // (an odd number of strips: B rests in the last round)
      const bool first = s == 0, last = !probe && s + 1u == nstrips;
      const uint32_t wbase = s * NW, R = wbase * 32u;
      if (mine) {
        NtStream ps;
        ps.init(pat, R);
#pragma unroll
        for (int wi = 0; wi < NW; ++wi) {
          uint32_t mask[5];
          peq_word(ps, wbase + wi, m, mask);
#pragma unroll
          for (int c = 0; c < 5; ++c) peq[(c * NW + wi) * kBlock + tid] = mask[c];
        }
      }
#pragma unroll
      for (int wi = 0; wi < NW; ++wi) { VP[wi] = 0xFFFFFFFFu; VN[wi] = 0u; }
      // windows: mine [g_lo, g_hi); the strip above me [ab_lo, ab_hi); the strip below me starts at g_next
      uint32_t g_lo = 0u, g_hi = 0u, g_next = 0xFFFFFFFFu, ab_lo = 0u, ab_hi = 0u;
      // the round's iterations `it`: A works on group it, B on group it - 1
      uint32_t it_lo, it_hi;
      if (probe) {
        const uint32_t pg = (3u * NW) / 2u < n_groups ? (3u * NW) / 2u : n_groups;  // 1.5 x the strip's height in columns
        g_hi = mine ? pg : 0u;
        it_lo = 0u; it_hi = pg;
      } else {
        const uint32_t sa = 2u * (uint32_t)rnd;
        g_lo = win_lo(s, h_cur); g_hi = mine ? win_hi(s, h_cur) : g_lo;
        if (mine && !last) g_next = win_lo(s + 1u, lag ? h_next : h_cur);
        if (!first) { ab_lo = win_lo(s - 1u, lag ? h_cur : h_prev); ab_hi = win_hi(s - 1u, lag ? h_cur : h_prev); }
        const uint32_t a_lo = win_lo(sa, h_cur), a_hi = win_hi(sa, h_cur);
        const uint32_t b_hi = sa + 1u < nstrips ? win_hi(sa + 1u, h_cur) + 1u : 0u;
        it_lo = a_lo; it_hi = a_hi > b_hi ? a_hi : b_hi;
        h_min = h_cur < h_min ? h_cur : h_min;
      }
      uint32_t probe_min = 0xFFFFFFFFu;
      if (lag == 0u) bottom = (first ? 0 : round_top) + (int32_t)HS;  // A: on top of what B left (B: set when A reaches B's first group)
      int32_t next_top = bottom;  // B: what the A lane of the next round starts on top of (g_next == g_lo: right here)
      uint32_t Pout = 0, Nout = 0;
      auto column = [&](uint32_t c, uint32_t hp_prev, uint32_t hn_prev) {
        const uint32_t* __restrict__ e = peq + c * (NW * kBlock) + tid;
        uint32_t carry = hn_prev >> 31;  // the add carry out of the strip above == its HN bit
        myers_word_range<NW, 0, NW - 1, false, (RPA_PAIR_SHIFT_STRIPS != 0)>(VP, VN, [&](int wi) { return e[wi * kBlock]; }, carry, hp_prev, hn_prev);  // (leaves the last word's HP / HN)
        Pout = __builtin_amdgcn_alignbit(Pout, hp_prev, 31);  // (Pout << 1) | (HP >> 31): column k of a group ends up at bit 31-k
        Nout = __builtin_amdgcn_alignbit(Nout, hn_prev, 31);
      };
      NtStream ts;
      uint4 pending = make_uint4(0u, 0u, 0u, 0u);
      if (g_lo < g_hi) { ts.init(txt, g_lo * 32u); pending = ts.load4(); upd_cells += (unsigned long long)(g_hi - g_lo) * (32u * HS); }
      // A lanes: the planes of the strip above come from the scratch, asked for a group ahead
      uint32_t Pn = 0xFFFFFFFFu, Nn = 0u;
      if (lag == 0u && g_lo < g_hi && g_lo >= ab_lo && g_lo < ab_hi) { Pn = sc[(size_t)(2u * g_lo) * stride]; Nn = sc[(size_t)(2u * g_lo + 1u) * stride]; }
      uint32_t Pkeep = 0u, Nkeep = 0u;  // my planes of the group I finished in the previous iteration (A: what B takes next)
      for (uint32_t it = it_lo; it < it_hi; ++it) {
        // from the lane above (the A lane of my pair, for B): its planes of the group it just finished = the group I am about
        // to work on, and its bottom score BEFORE this iteration's group (= at column 32 it)
        const uint32_t upP = (uint32_t)__shfl_up((int)Pkeep, 1, 64), upN = (uint32_t)__shfl_up((int)Nkeep, 1, 64);
        const int32_t up_bottom = __shfl_up(bottom, 1, 64);
        if (lag == 1u && mine && it == g_lo) { bottom = up_bottom + (int32_t)HS; next_top = bottom; }  // B starts here next iteration
        const uint32_t g = it - lag;  // wraps for B at it == 0: not inside any window
        if (g >= g_lo && g < g_hi) {
          const uint4 buf = ts.take32(pending);
          if (g + 1u < g_hi) pending = ts.load4();
          uint32_t Pin, Nin;
          if (lag == 0u) {
            Pin = Pn; Nin = Nn;
            Pn = 0xFFFFFFFFu; Nn = 0u;
            if (g + 1u < g_hi && g + 1u >= ab_lo && g + 1u < ab_hi) {
              const uint32_t* __restrict__ nx = sc + (size_t)(2u * (g + 1u)) * stride;
              Pn = nx[0]; Nn = nx[stride];
            }
          } else {
            const bool have = g >= ab_lo && g < ab_hi;  // did the A lane sweep this group?
            Pin = have ? upP : 0xFFFFFFFFu; Nin = have ? upN : 0u;
          }
          const uint32_t cols = n - g * 32u;
          Pout = 0u; Nout = 0u;
#pragma unroll 1
          for (uint32_t q = 0; q < 4u && q * 8u < cols; ++q) {
            const uint32_t d = q == 0u ? buf.x : q == 1u ? buf.y : q == 2u ? buf.z : buf.w;
            const uint32_t left = cols - q * 8u;
            if (left >= 8u) {
#pragma unroll
              for (int k = 0; k < 8; ++k) column((d >> (4 * k)) & 7u, Pin << k, Nin << k);
              Pin <<= 8; Nin <<= 8;
            } else {
#pragma unroll 1
              for (uint32_t k = 0; k < left; ++k) {
                column((d >> (4u * k)) & 7u, Pin, Nin);
                Pin <<= 1; Nin <<= 1;
              }
            }
          }
          bottom += (int32_t)__popc(Pout) - (int32_t)__popc(Nout);
          if (g + 1u == g_next) next_top = bottom;
          probe_min = (uint32_t)bottom < probe_min ? (uint32_t)bottom : probe_min;
          if (cols < 32u) { const uint32_t sh = 32u - cols; Pout <<= sh; Nout <<= sh; }
          Pkeep = Pout; Nkeep = Nout;
          if (lag == 1u && !last && !probe && g >= g_next) {  // B -> the A lane of the next round: through the scratch
            uint32_t* __restrict__ o = sc + (size_t)(2u * g) * stride;
            o[0] = Pout; o[stride] = Nout;
          }
        }
      }
      // B's strip is the round's lower one: the next round's A strip starts on top of it.  (An A strip that is the last of
      // an odd count has no successor.)
      {
        const int32_t from_b = __shfl_down(next_top, 1, 64);
        if (lag == 0u) round_top = from_b;
      }
      if (!wide_open) {
        // budget for all m rows from the cheapest way to have consumed the rows seen so far (the round's deepest strip);
        // one half width for the whole wavefront, effective two rounds on (k_myers_band)
        const uint32_t rows = R + HS;
        const float f = 1.0f + (float)((int32_t)margin - (int32_t)kProbeMarginDen) * (1.0f / kProbeMarginDen) * __builtin_sqrtf((float)HS / (float)rows);
        const float est = (float)probe_min * ((float)m / (float)rows) * f + (float)kProbeSlack;
        uint32_t mine_h = est * 0.5f < (float)h_open ? (uint32_t)(est * 0.5f) : h_open;
        if (!mine || probe_min == 0xFFFFFFFFu) mine_h = 0u;  // a resting lane has no say
        // within a pair the deeper strip's estimate counts (more rows seen): take B's when B worked
        const uint32_t other = (uint32_t)__shfl_xor((int)mine_h, 1, 64);
        const bool b_worked = probe ? false : (2u * (uint32_t)rnd + 1u < nstrips);
        const uint32_t pair_h = b_worked ? (lag ? mine_h : other) : (lag ? other : mine_h);
        uint32_t hmax = 0u;
        for (unsigned long long mk = __ballot(1); mk; mk &= mk - 1ull) {
          const uint32_t o = (uint32_t)__builtin_amdgcn_readlane((int)pair_h, __ffsll((long long)mk) - 1);
          hmax = o > hmax ? o : hmax;
        }
        if (probe) { h_prev = hmax; h_cur = hmax; h_next = hmax; }
        else { h_prev = h_cur; h_cur = h_next; h_next = hmax < h_cur + HS ? hmax : h_cur + HS; }
      }
    }
    // the lane that swept the last strip holds the answer
    if (((nstrips - 1u) & 1u) == lag) {
      int32_t dist = bottom;
#pragma unroll
      for (int wi = 0; wi < NW; ++wi) {
        const uint32_t pos = ((nstrips - 1u) * NW + wi) * 32u;
        const uint32_t tail = pos >= m ? 0xFFFFFFFFu : (m - pos >= 32u ? 0u : ~((1u << (m - pos)) - 1u));
        dist -= (int32_t)__popc(VP[wi] & tail) - (int32_t)__popc(VN[wi] & tail);
      }
      const uint64_t k = 2ull * h_min + (uint64_t)(delta < 0 ? -delta : delta);
      uint32_t a, b, dst;
      const uint2 entry = list[t];  // (read again: two registers less to carry through the sweep)
      resolve_pair(w, entry, a, b, dst);
      if ((uint64_t)(uint32_t)dist <= k) store_dna_result(w, dst, dist, n, m);
      else retry[atomicAdd(retry_n, 1u)] = entry;
    }
  }
  if (prof_acc) profile_add_updated(prof_acc, upd_cells);
}

static size_t strip_words_per_thread(uint32_t max_len) { return 2 * (((size_t)max_len + 31) / 32) + 2; }  // two bit planes per 32 columns

size_t dna_strip_scratch_words(uint32_t n_pairs, uint32_t max_len) {
  size_t threads = ((size_t)n_pairs + kBlock - 1) / kBlock * kBlock;
  // at most 32 GiB of boundary bytes in flight (of 288 GB of HBM), and never fewer than one workgroup.  The cap decides
  // into how many launches a list is cut, and every launch has a tail in which most of the chip waits for its last
  // workgroups: config 5 ran its 700 k-pair rounds as two launches under a 4 GiB cap and 4 % faster as one.
  const size_t cap = (((size_t)1 << 33) / strip_words_per_thread(max_len)) / kBlock * kBlock;
  if (threads > cap) threads = cap ? cap : kBlock;
  return threads * strip_words_per_thread(max_len);
}

// n_dev != null: the list's length is read on the device (a retry list filled by the launch before); `n` then only
// bounds it, and the grid is `grid_blocks` workgroups that stride over the list.
hipError_t launch_align_dna_strips(hipStream_t st, int strip_words, const uint2* list, uint32_t n, const uint32_t* n_dev,
                                   uint32_t grid_blocks, const WorkDev& w, uint32_t* scratch, size_t scratch_words, uint32_t max_len,
                                   bool wide_open, uint32_t margin, uint2* retry, uint32_t* retry_n, unsigned long long* prof_acc) {
  if (n == 0) return hipSuccess;
  if (!retry || !retry_n) wide_open = true;  // nowhere to put a pair that leaves its band
  const char* bd = getenv("RPA_BAND_DUO");
  const bool duo = !(bd && atoi(bd) == 0);  // two lanes per pair (default) / one (A/B, tests)
  const uint32_t per_block = duo ? kBlock / 2 : kBlock;  // pairs a workgroup holds = scratch slots it needs
  size_t slots = scratch_words / strip_words_per_thread(max_len) / per_block * per_block;
  if (slots == 0) return hipErrorOutOfMemory;
  const size_t lds = (size_t)5 * strip_words * kBlock * sizeof(uint32_t);
  for (uint32_t done = 0; done < n;) {
    uint32_t chunk = (uint32_t)((size_t)(n - done) < slots ? (n - done) : slots);
    uint32_t blocks = (chunk + per_block - 1) / per_block;
    if (n_dev) {  // one launch that strides over whatever the device-side count turns out to be
      chunk = n;
      if (grid_blocks && blocks > grid_blocks) blocks = grid_blocks;
    }
#define RPA_LAUNCH_STRIPS(NWV) \
  case NWV: \
    if (duo) hipLaunchKernelGGL(k_myers_band_duo<NWV>, dim3(blocks), dim3(kBlock), lds, st, list + done, chunk, n_dev, w, scratch, blocks * per_block, wide_open ? 1u : 0u, margin, retry, retry_n, prof_acc); \
    else hipLaunchKernelGGL(k_myers_band<NWV>, dim3(blocks), dim3(kBlock), lds, st, list + done, chunk, n_dev, w, scratch, blocks * per_block, wide_open ? 1u : 0u, margin, retry, retry_n, prof_acc); \
    break;
    switch (strip_words) {
      RPA_LAUNCH_STRIPS(5) RPA_LAUNCH_STRIPS(6) RPA_LAUNCH_STRIPS(7) RPA_LAUNCH_STRIPS(8)
      default: return hipErrorInvalidValue;
    }
#undef RPA_LAUNCH_STRIPS
    done += chunk;
  }
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------------------
// K2: pattern longer than 256 nt (up to 16384 nt), FEW pairs (too few lanes' worth for K1s) -- ONE ALIGNMENT PER WAVEFRONT.
// The pattern's 32-bit words are dealt to lanes in contiguous runs of WPL words (lane l owns words
// [l*WPL, (l+1)*WPL), WPL = ceil(words/64) is a template parameter, pairs are binned by it); at step t lane l
// processes text column t - l for its run, so the words of one column are computed on an anti-diagonal across
// lanes.  What crosses lanes per column is the add carry and the two shift carries (HP/HN top bits) out of a run,
// plus the text symbol: they travel together in one register, one lane to the right per step (__shfl_up by 1, a
// DPP wave shift) -- the wave itself is the systolic array, no LDS and no barrier.  Peq of the lane's run stays in
// registers as three bit planes (the symbol differs per lane, so a shared table would not help); the pattern is
// read from HBM once, the text once (64 dwords per coalesced prefetch, double buffered, handed to lane 0 by
// v_readlane).  Reference: the multi-block branch of the same SeqAn routine (global_alignment_myers_impl.h:129-191).
// ------------------------------------------------------------------------------------------------------------
template <int WPL>
__global__ __launch_bounds__(64) void k_myers_long(const uint2* __restrict__ list, uint32_t n_list,
                                                    const uint32_t* __restrict__ n_list_dev, WorkDev w,
                                                    unsigned long long* prof_acc) {
  // n_list_dev != null: a retry list whose length only the device knows (n_list bounds it); the grid strides over it
  const uint32_t total = n_list_dev ? (*n_list_dev < n_list ? *n_list_dev : n_list) : n_list;
  const int lane = threadIdx.x;
  if (n_list_dev && (*n_list_dev <= w.dev_count_lo || (w.dev_count_hi && *n_list_dev > w.dev_count_hi))) return;  // (another kernel's share of this list)
  if (n_list_dev && blockIdx.x == 0 && lane == 0 && total) atomicAdd(&w.counters[kCtrRetryTotal], total);
  for (uint32_t t = blockIdx.x; t < total; t += gridDim.x) {
  uint32_t a, b, dst;
  resolve_pair(w, list[t], a, b, dst);
  SegView txt = seg_view(w, a), pat = seg_view(w, b);
  if (txt.len < pat.len) { const SegView x = txt; txt = pat; pat = x; }
  const uint32_t n = txt.len, m = pat.len;
  if (prof_acc && lane == 0) {
    unsigned long long* acc = prof_acc + (blockIdx.x & (kProfParts - 1)) * kProfSlots;
    atomicAdd(&acc[0], 1ull);
    atomicAdd(&acc[1], (unsigned long long)n + m + 8ull);
    atomicAdd(&acc[2], (unsigned long long)n * m);
    atomicAdd(&acc[4], (unsigned long long)n * (((m + 31u) / 32u) * 32u));  // every word of the pattern, every column
  }
  const uint32_t txt_words = (n + 7u) / 8u;
  const uint32_t total_words = (m + 31u) / 32u;  // <= 64 * WPL by binning
  const uint32_t w0 = (uint32_t)lane * WPL;
  const uint32_t last_word = total_words - 1u;
  const uint32_t lastbit = (m - 1u) & 31u;
  const uint32_t lanes_used = (total_words + WPL - 1u) / WPL;
  int32_t dist = (int32_t)m;

  uint32_t P0[WPL], P1[WPL], PN[WPL], VM[WPL], VP[WPL], VN[WPL];
#pragma unroll
  for (int k = 0; k < WPL; ++k) {
    const uint32_t wi = w0 + k;
    P0[k] = P1[k] = PN[k] = VM[k] = 0;
    if (wi < total_words) {
      planes32(pat.fetch32(wi * 32u), P0[k], P1[k], PN[k]);
      const uint32_t rem = m - wi * 32u;
      VM[k] = rem >= 32u ? 0xFFFFFFFFu : ((1u << rem) - 1u);
    }
    VP[k] = 0xFFFFFFFFu;
    VN[k] = 0u;
  }
  const uint32_t steps = n + lanes_used - 1u;
  uint32_t msg = 0;  // bits 0..2: carries out of my run for my column; bits 4..6: text symbol; bit 8: valid
  uint32_t tx_cur = (uint32_t)lane < txt_words ? txt.fetch8((uint32_t)lane * 8u) : 0u;  // text dwords [0,64)
  uint32_t tx_nxt = 64u + lane < txt_words ? txt.fetch8((64u + lane) * 8u) : 0u;
  for (uint32_t step = 0; step < steps; ++step) {
    if ((step & 511u) == 0u && step) {  // 512 columns = 64 dwords consumed: rotate the prefetch buffers
      tx_cur = tx_nxt;
      const uint32_t idx = ((step >> 9) + 1u) * 64u + lane;
      tx_nxt = idx < txt_words ? txt.fetch8(idx * 8u) : 0u;
    }
    const uint32_t dword = __shfl(tx_cur, (int)((step >> 3) & 63u), 64);
    uint32_t in = __shfl_up(msg, 1, 64);
    if (lane == 0) in = step < n ? (2u /* add carry 0, HP carry 1 (global), HN carry 0 */ | (((dword >> ((step & 7u) * 4u)) & 7u) << 4) | 0x100u) : 0u;
    uint32_t out = 0;
    if ((in & 0x100u) && (uint32_t)lane < lanes_used) {
      const uint32_t c = (in >> 4) & 7u;
      const uint32_t m0 = (c & 1u) ? 0xFFFFFFFFu : 0u, m1 = (c & 2u) ? 0xFFFFFFFFu : 0u, mn = (c & 4u) ? 0xFFFFFFFFu : 0u;
      uint32_t carry = in & 1u;
      uint32_t hp_prev = (in & 2u) ? 0x80000000u : 0u, hn_prev = (in & 4u) ? 0x80000000u : 0u;
#pragma unroll
      for (int k = 0; k < WPL; ++k) {
        const uint32_t eq = ~(P0[k] ^ m0) & ~(P1[k] ^ m1) & ~(PN[k] ^ mn) & VM[k];
        const uint32_t vp = VP[k], vn = VN[k];
        const uint32_t X = eq | vn;
        const uint64_t s = (uint64_t)vp + (uint64_t)(X & vp) + carry;
        carry = (uint32_t)(s >> 32);
        const uint32_t D0 = ((uint32_t)s ^ vp) | X;
        const uint32_t HN = vp & D0;
        const uint32_t HP = vn | ~(vp | D0);
        const uint32_t Xs = __builtin_amdgcn_alignbit(HP, hp_prev, 31);
        const uint32_t Hs = __builtin_amdgcn_alignbit(HN, hn_prev, 31);
        hp_prev = HP;
        hn_prev = HN;
        VN[k] = Xs & D0;
        VP[k] = Hs | ~(Xs | D0);
        if (w0 + k == last_word) dist += (int32_t)((HP >> lastbit) & 1u) - (int32_t)((HN >> lastbit) & 1u);
      }
      out = carry | ((hp_prev >> 31) << 1) | ((hn_prev >> 31) << 2) | (c << 4) | 0x100u;
    }
    msg = out;
  }
  dist = __shfl(dist, (int)(last_word / WPL), 64);  // the lane that owns the last word holds the distance
  if (lane == 0) {
    w.pair_res[res_index(w, dst)] = make_float2((float)dist, (float)((int32_t)n - dist));
    if (w.pair_ext) w.pair_ext[dst] = make_uint4(n - (uint32_t)dist, (uint32_t)dist - (n - m), n - m, n);
  }
  }
}

hipError_t launch_align_dna_long(hipStream_t st, int wpl, const uint2* list, uint32_t n, const uint32_t* n_dev, uint32_t grid_blocks,
                                 const WorkDev& w, unsigned long long* prof_acc) {
  if (n == 0) return hipSuccess;
  const uint32_t blocks = n_dev && grid_blocks && grid_blocks < n ? grid_blocks : n;
#define RPA_LAUNCH_LONG(V) \
  case V: hipLaunchKernelGGL(k_myers_long<V>, dim3(blocks), dim3(64), 0, st, list, n, n_dev, w, prof_acc); break;
  switch (wpl) {
    RPA_LAUNCH_LONG(1) RPA_LAUNCH_LONG(2) RPA_LAUNCH_LONG(3) RPA_LAUNCH_LONG(4)
    RPA_LAUNCH_LONG(5) RPA_LAUNCH_LONG(6) RPA_LAUNCH_LONG(7) RPA_LAUNCH_LONG(8)
    default: return hipErrorInvalidValue;
  }
#undef RPA_LAUNCH_LONG
  return hipGetLastError();
}

}  // namespace rpa
