// taxator-tk_amd/csrc/align_protein_kernels.hip -- protein re-alignment kernels for gfx950: K3 (any length) and K3p (the
// packed-cell form of the same recurrence for pairs up to ~312 x 312 residues, further down; the engine bins pairs by
// protein_fits_packed(), rpa_device.h).
//
// What it replaces: getAlignmentProtein (core/src/taxonpredictionmodelsequence.hh:173-242) = three SeqAn calls
//   selfscore = NW(A,A) + NW(B,B)                 (:190-191, full DP in the reference)
//   mutual    = globalAlignment(Align(A,B), Blosum62, LinearGaps)  with traceback  (:199-203)
//   walk of the two gapped rows counting gaps / matches / mismatches              (:218-226)
// and the float normalisation (:227-233).  SeqAn cell (seqan/align/dp_formula_linear.h:63-107): linear gap -1,
// best = max(diag + s(a,b), vert - 1, horiz - 1) with the LEFT operand winning ties in _maxScore
// (seqan/align/dp_formula.h:153-163): vertical over horizontal, then diagonal over that; single trace, so the path is
// unique (seqan/align/dp_traceback_impl.h:379-412).  A is the horizontal sequence (:194-197).
//
// Kernel design: ONE ALIGNMENT PER LANE, no trace matrix and no traceback.  Each cell carries, next to its score,
// (#diagonal steps, #matches) of the unique path that reaches it, taken from whichever predecessor wins under the
// tie order above; at the corner gaps = (la - ndiag) + (lb - ndiag), mismatches = ndiag - nmatch -- identical to
// walking the reference's traceback (the CPU checker under /oracle proves the two formulations equal and both are
// pinned to the reference's SeqAn).  NW(X,X) = sum of s(x,x) because every off-diagonal BLOSUM62 entry is below the
// diagonal ones it would replace (checked against SeqAn by the oracle tests), so no self DP is run.
// The vertical sequence B is cut into strips of R rows whose (score, counters) column lives in 2R VGPRs; the lane
// sweeps the columns of A per strip.  The strip's bottom row is handed to the next strip through a [column][lane]
// scratch array in HBM (coalesced 8 B per lane per column, L2 resident).  Substitution scores come from a 24 KB LDS
// table replicated across the 32 banks, T[b][a/4][bank] = four int8 scores, so that a lane always reads its own
// bank: conflict-free ds_read_b32 + v_bfe_i32.  ~14 VALU + 1 LDS op per cell; integer VALU bound; no MFMA (max-plus
// recurrences with data-dependent tie-breaks are not a contraction).
#include "rpa_device.h"
#include "rpa_tables.inc"

namespace rpa {

constexpr int kAA = 27;
constexpr int kAAGroups = 7;  // ceil(27/4)
constexpr int kAX = kAA + 1;   // K3l: the residues and one more code (null column / padding row)

__device__ __constant__ int8_t c_blosum62[kAA * kAA];
static bool g_blosum_uploaded[64] = {false};

static hipError_t upload_blosum() {
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  if (dev < 64 && g_blosum_uploaded[dev]) return hipSuccess;
  e = hipMemcpyToSymbol(HIP_SYMBOL(c_blosum62), RPA_BLOSUM62, kAA * kAA);
  if (e != hipSuccess) return e;
  if (dev < 64) g_blosum_uploaded[dev] = true;
  return hipSuccess;
}

template <int R>
__global__ __launch_bounds__(kBlock) void k_nw_protein(const uint2* __restrict__ list, uint32_t n_list, WorkDev w,
                                                        int2* __restrict__ scratch, uint32_t col_stride,
                                                        unsigned long long* prof_acc) {
  __shared__ uint32_t tab[kAA * kAAGroups * 32];  // [b][a/4][bank] -> 4 x int8 scores s(a, b)
  __shared__ int32_t selfdiag[32];
  for (int idx = threadIdx.x; idx < kAA * kAAGroups * 32; idx += kBlock) {
    const int bres = idx / (kAAGroups * 32), ag = (idx / 32) % kAAGroups;
    uint32_t v = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int a = ag * 4 + k;
      const int sc = a < kAA ? c_blosum62[a * kAA + bres] : 0;
      v |= (uint32_t)(uint8_t)(int8_t)sc << (8 * k);
    }
    tab[idx] = v;
  }
  if (threadIdx.x < 32) selfdiag[threadIdx.x] = threadIdx.x < kAA ? c_blosum62[threadIdx.x * kAA + threadIdx.x] : 0;
  __syncthreads();

  const uint32_t t = blockIdx.x * kBlock + threadIdx.x;
  const bool active = t < n_list;
  uint32_t dst = 0, la = 0, lb = 0;
  const uint8_t* __restrict__ pa = nullptr;
  const uint8_t* __restrict__ pb = nullptr;
  if (active) {
    uint32_t sa, sb;
    resolve_pair(w, list[t], sa, sb, dst);
    const SegView va = seg_view(w, sa), vb = seg_view(w, sb);  // protein segments are plain byte ranges of the store
    la = va.len; lb = vb.len;
    pa = reinterpret_cast<const uint8_t*>(va.words) + va.src;
    pb = reinterpret_cast<const uint8_t*>(vb.words) + vb.src;
  }
  if (prof_acc) {
    // same accounting as the DNA kernels: pairs, len_a+len_b+8, cells
    // (+ cells really updated: every cell of the strips, padding rows of the last strip included)
    __shared__ unsigned long long red[4][kBlock / 64];
    unsigned long long v0 = active ? 1ull : 0ull, v1 = active ? (unsigned long long)la + lb + 8ull : 0ull,
                       v2 = active ? (unsigned long long)la * lb : 0ull, v3 = active ? (unsigned long long)la * (((lb + R - 1) / R) * R) : 0ull;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { v0 += __shfl_down(v0, d, 64); v1 += __shfl_down(v1, d, 64); v2 += __shfl_down(v2, d, 64); v3 += __shfl_down(v3, d, 64); }
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = v0; red[1][threadIdx.x >> 6] = v1; red[2][threadIdx.x >> 6] = v2; red[3][threadIdx.x >> 6] = v3; }
    __syncthreads();
    if (threadIdx.x < 4) {
      unsigned long long s = 0;
      for (int k = 0; k < kBlock / 64; ++k) s += red[threadIdx.x][k];
      atomicAdd(&prof_acc[(blockIdx.x & (kProfParts - 1)) * kProfSlots + (threadIdx.x < 3 ? threadIdx.x : 4)], s);
    }
  }
  if (!active) return;

  const uint32_t bank4 = (threadIdx.x & 31u) * 4u;                 // byte offset of my bank inside a table row
  const uint8_t* __restrict__ tabb = reinterpret_cast<const uint8_t*>(tab);
  int2* __restrict__ sc = scratch + t;                             // my boundary row: element j at sc[j * col_stride]
  int32_t selfscore = 0;
  for (uint32_t j = 0; j < la; ++j) selfscore += selfdiag[pa[j]];  // NW(A,A), :190
  for (uint32_t i = 0; i < lb; ++i) selfscore += selfdiag[pb[i]];  // NW(B,B), :191

  const uint32_t nstrips = (lb + R - 1) / R;
  int32_t res_s = 0;
  uint32_t res_c = 0;
  for (uint32_t k = 0; k < nstrips; ++k) {
    // residues of the strip's rows: B[k*R .. k*R+R)
    uint32_t brow[R / 4];
#pragma unroll
    for (int q = 0; q < R / 4; ++q) {  // pb is byte aligned only: assemble the words from bytes (rows past lb are never used)
      const uint8_t* bp = pb + (size_t)k * R + 4 * q;
      brow[q] = (uint32_t)bp[0] | ((uint32_t)bp[1] << 8) | ((uint32_t)bp[2] << 16) | ((uint32_t)bp[3] << 24);
    }
    int32_t Ls[R];
    uint32_t Lc[R];
#pragma unroll
    for (int r = 0; r < R; ++r) { Ls[r] = -(int32_t)(k * R + r + 1); Lc[r] = 0u; }  // column 0: M[i][0] = -i
    int32_t dtop_s = -(int32_t)(k * R);  // (row k*R, column 0)
    uint32_t dtop_c = 0u;
    const bool first = k == 0, last = k + 1 == nstrips;
    int2 nxt = make_int2(0, 0);
    if (!first) nxt = sc[(size_t)1 * col_stride];
    uint32_t a_next = pa[0];  // column residues are fetched one column ahead of their use
    for (uint32_t j = 1; j <= la; ++j) {
      const uint32_t a = a_next;
      if (j < la) a_next = pa[j];
      const uint32_t colterm = (a >> 2) * 128u + bank4;
      const uint32_t shift = (a & 3u) * 8u;
      int32_t top_s;
      uint32_t top_c;
      if (first) { top_s = -(int32_t)j; top_c = 0u; }  // row 0: M[0][j] = -j
      else {
        top_s = nxt.x; top_c = (uint32_t)nxt.y;
        if (j < la) nxt = sc[(size_t)(j + 1) * col_stride];  // prefetch the next column's boundary
      }
      int32_t up_s = top_s, dg_s = dtop_s;
      uint32_t up_c = top_c, dg_c = dtop_c;
      dtop_s = top_s; dtop_c = top_c;
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const uint32_t b = (brow[r >> 2] >> (8 * (r & 3))) & 0xFFu;
        const uint32_t word = *reinterpret_cast<const uint32_t*>(tabb + b * (kAAGroups * 128u) + colterm);
        const int32_t sub = (int32_t)(word << (24u - shift)) >> 24;  // v_bfe_i32
        const int32_t ls = Ls[r];
        const uint32_t lc = Lc[r];
        const int32_t sd = dg_s + sub, sv = up_s - 1, sh = ls - 1;
        // vertical beats horizontal on ties, diagonal beats both (dp_formula.h:153-163)
        const int32_t ts = sv < sh ? sh : sv;
        const uint32_t tc = sv < sh ? lc : up_c;
        const uint32_t dc = dg_c + 0x10000u + (a == b ? 1u : 0u);
        const int32_t bs = sd < ts ? ts : sd;
        const uint32_t bc = sd < ts ? tc : dc;
        dg_s = ls; dg_c = lc;
        Ls[r] = bs; Lc[r] = bc;
        up_s = bs; up_c = bc;
      }
      if (!last) sc[(size_t)j * col_stride] = make_int2(up_s, (int)up_c);
    }
    if (last) {
      const uint32_t rr = lb - k * R - 1;  // row lb inside this strip
#pragma unroll
      for (int r = 0; r < R; ++r)
        if ((uint32_t)r == rr) { res_s = Ls[r]; res_c = Lc[r]; }
    }
  }
  const int32_t mutual = res_s;
  const uint32_t ndiag = res_c >> 16, nmatch = res_c & 0xFFFFu;
  const uint32_t gap = (la - ndiag) + (lb - ndiag);
  const uint32_t mismatch = ndiag - nmatch;
  const uint32_t len = gap + nmatch + mismatch;                       // :227
  const float normfactor = (float)len / (float)selfscore;             // :228
  float distance = (float)(selfscore - 2 * mutual) * normfactor;      // :232
  const float similarity = (float)(2 * mutual) * normfactor;          // :233
  if (!(selfscore >= mutual) || !(distance >= 0.f)) distance = __builtin_nanf("");  // live asserts :205,:240 -> flagged, raised if consumed
  w.pair_res[res_index(w, dst)] = make_float2(distance, similarity);
  if (w.pair_ext) w.pair_ext[dst] = make_uint4(nmatch, mismatch, gap, len);
}

// ------------------------------------------------------------------------------------------------------------
// K3p: the same recurrence for pairs that pass protein_fits_packed() (up to 312 x 312 residues; every pair of the 67 aa
// workload), with a cell packed into ONE 32-bit word so that the three-way choice AND its tie order are a single
// unsigned maximum:
//     bits 20..31  M[i][j] + i + bias  (row-relative score; bias = la + 32 keeps every candidate non-negative; <= 4095 by the bin's test)
//     bits 18..19  which move produced the candidate: 2 diagonal, 1 vertical, 0 horizontal
//     bits  9..17  #matches            (<= 511)
//     bits  0..8   #diagonal steps     (<= 511)
// max() orders candidates by score first and, on equal scores, by the move field = diagonal over vertical over
// horizontal, the reference's order (dp_formula.h:153-163); the counters ride in the low bits of whichever wins.
// Cells are kept with the move field SET TO 1 ("as a vertical candidate") and scores carry + i, so that
//     vertical   candidate of the cell below = the cell itself               (gap -1 and row +1 cancel)  -- no instruction
//     horizontal candidate = left + (-1 score, move 1 -> 0)                                              -- one add
//     diagonal   candidate = diag + T(a,b),  T = (s(a,b) + 1) << 20 | (move 1 -> 2) | (a == b) << 9 | 1  -- one add
// T is a ready-made 32-bit LDS table entry, the table replicated over the 32 banks (27 x 27 x 32 x 4 B = 91 KB) so
// that a lane always reads its own bank.  Per cell: address add, ds_read_b32, two adds, v_max3_u32, v_and_or_b32 -- five
// VALU instructions with a dependent chain of two (max3 -> and_or), against ~17 with compares and selects through SGPR
// masks in K3, and half the strip-boundary traffic (4 B per column).
// One 1024-lane workgroup per CU holds the table; its wavefronts are persistent and draw runs of 64 pairs from a
// counter until the list is empty, so a long pair never keeps a whole workgroup's table waiting.
// Same results as K3 bit for bit (same recurrence; no field can overflow at these lengths).
// ------------------------------------------------------------------------------------------------------------
#ifndef RPA_PACKED_BLOCK
#define RPA_PACKED_BLOCK 1024
#endif
#ifndef RPA_PACKED_ROWS
#define RPA_PACKED_ROWS 24
#endif
constexpr int kPackedBlock = RPA_PACKED_BLOCK;
constexpr int kPackedRows = RPA_PACKED_ROWS;
constexpr uint32_t kPkScore = 20, kPkMove = 18, kPkMatch = 9;
constexpr uint32_t kPackedScratchHead = 64;  // words in front of the boundary rows: [0] = the work counter

// kWide: the SCORE-AND-LENGTH cell for pairs the cell above cannot hold -- `M[i][j] + i + bias` in bits 18-31 (14 bits: la + lb +
// 11 min(la, lb) + 32 <= 16383, e.g. 1258 x 1258 residues), the move in bits 16-17, the number of diagonal steps in bits
// 0-15, no match counter.  The placement path only ever uses an alignment's distance and similarity, i.e. its score and its
// LENGTH = la + lb - #diagonal steps (taxonpredictionmodelsequence.hh:227-233); matches / mismatches are needed by
// rpa_align_pairs alone, which keeps K3 for such pairs.  Which candidate wins a cell depends on (score, move) only -- the
// counters sit below them -- so the path, and with it the length, is the one the full cell would have taken.
template <int R, bool kWide>
__global__ __launch_bounds__(kPackedBlock) void k_nw_protein_packed(const uint2* __restrict__ list, uint32_t n_list, WorkDev w,
                                                                    uint32_t* __restrict__ scratch, uint32_t col_stride,
                                                                    uint32_t* __restrict__ work_counter, uint32_t counter_first,
                                                                    unsigned long long* prof_acc) {
  constexpr uint32_t kPkScore = kWide ? 18u : rpa::kPkScore, kPkMove = kWide ? 16u : rpa::kPkMove, kPkMoveMask = 3u << kPkMove;
  constexpr uint32_t kDiagMask = kWide ? 0xFFFFu : 0x1FFu;
  __shared__ uint32_t tab[kAA * kAA * 32];  // [b][a][bank] -> T(a,b)
  __shared__ int32_t selfdiag[32];
  __shared__ int8_t blosum[kAA * kAA];       // the 729 scores once, then the expansion reads LDS
  if (threadIdx.x < kAA * kAA) blosum[threadIdx.x] = c_blosum62[threadIdx.x];
  __syncthreads();
  for (int idx = threadIdx.x; idx < kAA * kAA * 32; idx += kPackedBlock) {
    const int bres = idx / (kAA * 32), a = (idx / 32) % kAA;
    const int sc = blosum[a * kAA + bres];
    tab[idx] = ((uint32_t)(sc + 1) << kPkScore) + (1u << kPkMove) + ((!kWide && a == bres) ? 1u << kPkMatch : 0u) + 1u;
  }
  if (threadIdx.x < 32) selfdiag[threadIdx.x] = threadIdx.x < kAA ? blosum[threadIdx.x * kAA + threadIdx.x] : 0;
  __syncthreads();

  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t bank4 = (threadIdx.x & 31u) * 4u;
  const uint8_t* __restrict__ tabb = reinterpret_cast<const uint8_t*>(tab);
  // my boundary row: columns 4g+1 .. 4g+4 at sc4[g * col_stride] (16 bytes per lane and group: coalesced)
  uint4* __restrict__ sc4 = reinterpret_cast<uint4*>(scratch + kPackedScratchHead) + blockIdx.x * kPackedBlock + threadIdx.x;
  // the constants of a cell live in VGPRs (literal / SGPR operands issue at half rate on gfx950)
  uint32_t c_horiz = 0u - (1u << kPkScore) - (1u << kPkMove);  // score - 1, move 1 -> 0
  uint32_t c_keep = ~kPkMoveMask, c_one = 1u << kPkMove;
  asm volatile("" : "+v"(c_horiz), "+v"(c_keep), "+v"(c_one));

  // The first run of a wavefront is fixed -- wavefront v of workgroup g takes run v * gridDim.x + g, so a short list is
  // dealt one run per CU before any CU gets a second -- the following ones come from the counter (a word that is ZERO when
  // the launch begins -- a fresh one of the engine's pool per launch, cleared once per batch, instead of a fill in front of
  // every launch -- plus counter_first: the pairs of the fixed runs).  Every wavefront leaves through the break: the counter only grows.
  uint32_t base = ((threadIdx.x >> 6) * gridDim.x + blockIdx.x) * 64u;
  for (;; ) {
    if (base >= n_list) break;
    const uint32_t run = base;
    if (lane == 0) base = atomicAdd(work_counter, 64u) + counter_first;
    base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
    const uint32_t t = run + lane;
    const bool active = t < n_list;
    uint32_t dst = 0, la = 0, lb = 0;
    const uint8_t* __restrict__ pa = nullptr;
    const uint8_t* __restrict__ pb = nullptr;
    if (active) {
      uint32_t sa, sb;
      resolve_pair(w, list[t], sa, sb, dst);
      const SegView va = seg_view(w, sa), vb = seg_view(w, sb);
      la = va.len; lb = vb.len;
      pa = reinterpret_cast<const uint8_t*>(va.words) + va.src;
      pb = reinterpret_cast<const uint8_t*>(vb.words) + vb.src;
    }
    if (prof_acc) {  // same accounting as the other alignment kernels: pairs, len_a+len_b+8, cells
      unsigned long long v0 = active ? 1ull : 0ull, v1 = active ? (unsigned long long)la + lb + 8ull : 0ull,
                         v2 = active ? (unsigned long long)la * lb : 0ull, v3 = active ? (unsigned long long)la * (((lb + R - 1) / R) * R) : 0ull;
#pragma unroll
      for (int d = 32; d >= 1; d >>= 1) { v0 += __shfl_down(v0, d, 64); v1 += __shfl_down(v1, d, 64); v2 += __shfl_down(v2, d, 64); v3 += __shfl_down(v3, d, 64); }
      if (lane == 0) {
        unsigned long long* acc = prof_acc + ((run >> 6) & (kProfParts - 1)) * kProfSlots;
        atomicAdd(&acc[0], v0); atomicAdd(&acc[1], v1); atomicAdd(&acc[2], v2); atomicAdd(&acc[4], v3);
      }
    }
    if (!active) continue;  // the last run of a list; nothing below is wave-collective

    int32_t selfscore = 0;  // NW(A,A) + NW(B,B) = sums of s(x,x) (:190-191), gathered while the residues pass by
    const uint32_t bias = la + 32u;
    const uint32_t nstrips = (lb + R - 1) / R;
    uint32_t res = bias << kPkScore;  // no rows: K3's answer (score 0, no steps)
    if (nstrips == 0)
      for (uint32_t j = 0; j < la; ++j) selfscore += selfdiag[pa[j]];
    for (uint32_t k = 0; k < nstrips; ++k) {
      const bool first = k == 0, last = k + 1 == nstrips;
      uint32_t rowoff[R];  // LDS byte offset of the table row of each strip row's residue
      {
        uint32_t bw[R / 4];  // unaligned dword loads; reads run at most R - 1 bytes past the segment (store slack)
#pragma unroll
        for (int q = 0; q < R / 4; ++q) __builtin_memcpy(&bw[q], pb + (size_t)k * R + 4 * q, 4);
#pragma unroll
        for (int r = 0; r < R; ++r) {
          const uint32_t bres = (bw[r >> 2] >> (8 * (r & 3))) & 0xFFu;  // rows past lb are never used
          rowoff[r] = bres * (kAA * 128u) + bank4;
          if (k * R + r < lb) selfscore += selfdiag[bres];
        }
      }
      uint32_t L[R];
#pragma unroll
      for (int r = 0; r < R; ++r) L[r] = (bias << kPkScore) | c_one;  // column 0: M[i][0] + i = 0
      uint32_t dtop = (bias << kPkScore) | c_one;                      // (row k*R, column 0)
      // one column: residue a, the cell above the strip; returns the strip's bottom cell
      auto column = [&](uint32_t a, uint32_t top) -> uint32_t {
        const uint32_t colterm = a * 128u;
        uint32_t up = top, dg = dtop;
        dtop = top;
#pragma unroll
        for (int r = 0; r < R; ++r) {
          const uint32_t tv = *reinterpret_cast<const uint32_t*>(tabb + rowoff[r] + colterm);
          const uint32_t left = L[r];
          const uint32_t raw = max(max(dg + tv, up), left + c_horiz);
          const uint32_t cell = __builtin_amdgcn_bitop3_b32(raw, c_keep, c_one, 0xEA);  // (raw & keep) | one: v_bitop3 issues at full rate, v_and_or at half
          dg = left;
          L[r] = cell;
          up = cell;
        }
        return up;
      };
      // Columns go in groups of four: the residues of A come as unaligned dwords asked for two groups ahead, the
      // boundary cells of the strip above as one 16-byte load per group asked for one group ahead, and the strip's own
      // bottom cells leave as one 16-byte store per group.  (Memory operations retire in order on gfx9: with a byte
      // load and a 4-byte store per column, every column waited for the previous column's store to reach memory.)
      uint32_t ad_cur = 0u, ad_nxt = 0u;
      if (la) __builtin_memcpy(&ad_cur, pa, 4);
      if (la > 4u) __builtin_memcpy(&ad_nxt, pa + 4, 4);
      uint4 t_cur = make_uint4(0u, 0u, 0u, 0u);
      if (!first && la) t_cur = sc4[0];
      const uint32_t full = la & ~3u;
      uint32_t jb = 0;
      for (; jb < full; jb += 4u) {  // columns jb+1 .. jb+4
        const uint32_t g = jb >> 2;
        uint4 t_nxt = make_uint4(0u, 0u, 0u, 0u);
        if (!first && jb + 4u < la) t_nxt = sc4[(size_t)(g + 1u) * col_stride];
        uint32_t ad_far = 0u;
        if (jb + 8u < la) __builtin_memcpy(&ad_far, pa + jb + 8u, 4);
        const uint32_t a0 = ad_cur & 0xFFu, a1 = (ad_cur >> 8) & 0xFFu, a2 = (ad_cur >> 16) & 0xFFu, a3 = ad_cur >> 24;
        uint4 top = t_cur;
        if (first) {  // row 0: M[0][j] = -j
          const uint32_t t0 = ((bias - (jb + 1u)) << kPkScore) | c_one;
          top = make_uint4(t0, t0 - (1u << kPkScore), t0 - (2u << kPkScore), t0 - (3u << kPkScore));
          selfscore += selfdiag[a0] + selfdiag[a1] + selfdiag[a2] + selfdiag[a3];
        }
        uint4 out;
        out.x = column(a0, top.x);
        out.y = column(a1, top.y);
        out.z = column(a2, top.z);
        out.w = column(a3, top.w);
        if (!last) sc4[(size_t)g * col_stride] = out;
        t_cur = t_nxt; ad_cur = ad_nxt; ad_nxt = ad_far;
      }
      if (jb < la) {  // the last one to three columns, one at a time
        uint4 out = make_uint4(0u, 0u, 0u, 0u);
        const uint32_t ncol = la - jb;
#pragma unroll 1
        for (uint32_t c = 0; c < ncol; ++c) {
          const uint32_t a = (ad_cur >> (8u * c)) & 0xFFu;
          uint32_t top = c == 0u ? t_cur.x : c == 1u ? t_cur.y : t_cur.z;
          if (first) { top = ((bias - (jb + c + 1u)) << kPkScore) | c_one; selfscore += selfdiag[a]; }
          const uint32_t bottom = column(a, top);
          if (c == 0u) out.x = bottom; else if (c == 1u) out.y = bottom; else out.z = bottom;
        }
        if (!last) sc4[(size_t)(jb >> 2) * col_stride] = out;
      }
      if (last) {
        const uint32_t rr = lb - k * R - 1;
#pragma unroll
        for (int r = 0; r < R; ++r)
          if ((uint32_t)r == rr) res = L[r];
      }
    }
    const int32_t mutual = (int32_t)(res >> kPkScore) - (int32_t)(bias + lb);
    const uint32_t ndiag = res & kDiagMask, nmatch = kWide ? 0u : (res >> kPkMatch) & 0x1FFu;  // (wide: no match counter; length = gaps + diagonal steps either way)
    const uint32_t gap = (la - ndiag) + (lb - ndiag);
    const uint32_t mismatch = ndiag - nmatch;
    const uint32_t len = gap + nmatch + mismatch;                       // :227
    const float normfactor = (float)len / (float)selfscore;             // :228
    float distance = (float)(selfscore - 2 * mutual) * normfactor;      // :232
    const float similarity = (float)(2 * mutual) * normfactor;          // :233
    if (!(selfscore >= mutual) || !(distance >= 0.f)) distance = __builtin_nanf("");
    w.pair_res[res_index(w, dst)] = make_float2(distance, similarity);
    if (w.pair_ext) w.pair_ext[dst] = kWide ? make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, gap, len) : make_uint4(nmatch, mismatch, gap, len);  // (wide: matches / mismatches not counted)
  }
}

// ------------------------------------------------------------------------------------------------------------
// K3l: K3p's cell with the strips of ONE pair in ADJACENT LANES (G lanes per pair, G = number of R-row strips the
// vertical sequence needs, 64 / G pairs per wavefront).  Lane s of a pair owns rows sR+1 .. sR+R and runs s columns
// behind lane s-1, so the cell above its strip for the column it is about to compute is the bottom cell its neighbour
// produced one step earlier: it comes over the DPP network (v_mov_b32 wave_shr:1, no LDS, no memory).  Nothing of the DP
// touches HBM: a launch reads the residues and writes one result per pair (K3p hands every strip boundary through a
// scratch row in memory: 13x the algorithmic bytes at 67 aa).  The price is the s idle steps per lane at either end of
// a pair (G - 1 of la + G - 1 steps), the lanes of strips a shorter pair of the same class does not have, and lane 63 when
// G = 3.  Same packed cell, same results as K3p bit for bit.
// No lane is ever masked inside the column loop (a branch around the column made the compiler copy all R cells at its
// end, +20 % VALU): a lane without a column at some step computes one for the NULL residue, a 28th table column of
// zeros.  In front of a pair that leaves the strip in its column-0 state K = (bias, move 1, no steps) -- diagonal K + 0,
// vertical K, horizontal K - 1 -> K again, and the cell handed down is K, which is what the neighbour's idle steps need --
// and behind a pair the strip's content no longer matters: the corner cell is picked up at the step that computes it.
// ------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t from_lane_below(uint32_t v) {  // lane i receives lane i-1's value (lane 0: 0)
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
}

// One run of 64 / G pairs of a G-lanes-per-pair list, starting at pair `run` (the whole wavefront calls it together).
template <int R, int G, bool kWide>
__device__ __forceinline__ void protein_lanes_run(const uint2* __restrict__ list, uint32_t n_list, uint32_t run, const WorkDev& w,
                                                  const uint8_t* __restrict__ tabb, const int32_t* __restrict__ selfdiag,
                                                  uint32_t c_horiz, uint32_t c_keep, uint32_t c_one, unsigned long long* prof_acc) {
  constexpr uint32_t kPkScore = kWide ? 18u : rpa::kPkScore, kPkMatch = rpa::kPkMatch;  // (kWide: the score-and-length cell of K3p)
  constexpr uint32_t kDiagMask = kWide ? 0xFFFFu : 0x1FFu;
  constexpr uint32_t kPairs = 64u / G;  // pairs per wavefront
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t sub = lane % G, slot = lane / G;  // my strip, my pair of the run (slot == kPairs: the spare lane of G = 3)
  const uint32_t bank4 = (threadIdx.x & 31u) * 4u;
  {
    const uint32_t t = run + slot;
    const bool have = slot < kPairs && t < n_list;
    uint32_t dst = 0, la = 0, lb = 0;
    const uint8_t* __restrict__ pa = nullptr;
    const uint8_t* __restrict__ pb = nullptr;
    if (have) {
      uint32_t sa, sb;
      resolve_pair(w, list[t], sa, sb, dst);
      const SegView va = seg_view(w, sa), vb = seg_view(w, sb);
      la = va.len; lb = vb.len;
      pa = reinterpret_cast<const uint8_t*>(va.words) + va.src;
      pb = reinterpret_cast<const uint8_t*>(vb.words) + vb.src;
    }
    if (prof_acc) {  // pairs, len_a+len_b+8, cells: counted by the first lane of a pair
      const bool cnt = have && sub == 0u;
      unsigned long long v0 = cnt ? 1ull : 0ull, v1 = cnt ? (unsigned long long)la + lb + 8ull : 0ull,
                         v2 = cnt ? (unsigned long long)la * lb : 0ull, v3 = cnt ? (unsigned long long)la * (((lb + R - 1) / R) * R) : 0ull;
#pragma unroll
      for (int d = 32; d >= 1; d >>= 1) { v0 += __shfl_down(v0, d, 64); v1 += __shfl_down(v1, d, 64); v2 += __shfl_down(v2, d, 64); v3 += __shfl_down(v3, d, 64); }
      if (lane == 0) {
        unsigned long long* acc = prof_acc + ((run / kPairs) & (kProfParts - 1)) * kProfSlots;
        atomicAdd(&acc[0], v0); atomicAdd(&acc[1], v1); atomicAdd(&acc[2], v2); atomicAdd(&acc[4], v3);
      }
    }
    const uint32_t nstrips = (lb + R - 1) / R;  // <= G by the bin
    const bool mine = have && sub < nstrips;
    const uint32_t cols = mine ? la : 0u;  // the columns this lane really computes
    const uint32_t bias = la + 32u;
    int32_t selfscore = 0;  // my share of NW(A,A) + NW(B,B): my rows of B; the first lane of the pair adds A's residues
    uint32_t rowoff[R];
    // Strips are aligned to the END of B: strip s > 0 owns rows p + (s-1)R + 1 .. p + sR, strip 0 the first p = lb - (nstrips-1)R
    // rows in its LAST p registers; the R - p registers above them are padding rows (residue code kAA: every table entry is
    // "score - 1, move 0", so the cell of a padding row is the cell above it: row 0 passes through them unchanged).  The
    // corner cell is then always the bottom cell of the pair's last lane.
    const uint32_t pad = mine && sub == 0u ? nstrips * R - lb : 0u;  // padding rows in front of my rows (strip 0 only)
    if (mine) {
      // my first register's row (0-based, may be "negative" for strip 0) relative to pb
      const uint8_t* __restrict__ rows = pb + (size_t)sub * R - (size_t)(nstrips * R - lb);
      uint32_t bw[(R + 3) / 4];  // unaligned dword loads; they start at most R - 1 bytes in front of the segment and end at most 3 bytes behind it (store slack)
#pragma unroll
      for (int q = 0; q < (R + 3) / 4; ++q) __builtin_memcpy(&bw[q], rows + 4 * q, 4);
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const uint32_t bres = (uint32_t)r < pad ? (uint32_t)kAA : (bw[r >> 2] >> (8 * (r & 3))) & 0xFFu;
        rowoff[r] = bres * (kAX * 128u) + bank4;
        selfscore += selfdiag[bres];  // (padding rows: 0)
      }
    } else {
#pragma unroll
      for (int r = 0; r < R; ++r) rowoff[r] = bank4;
    }
    if (have && sub == 0u && nstrips == 0u)  // no rows: K3p's answer (score 0, no steps), A's residues still count
      for (uint32_t j = 0; j < la; ++j) selfscore += selfdiag[pa[j]];
    const uint32_t K = (bias << kPkScore) | c_one;  // column 0: M[i][0] + i = 0
    uint32_t L[R];
#pragma unroll
    for (int r = 0; r < R; ++r) L[r] = K;
    uint32_t dtop = K;                              // (row sub*R, column 0)
    auto column = [&](uint32_t a, uint32_t top) -> uint32_t {
      const uint32_t colterm = a * 128u;
      uint32_t up = top, dg = dtop;
      dtop = top;
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const uint32_t tv = *reinterpret_cast<const uint32_t*>(tabb + rowoff[r] + colterm);
        const uint32_t left = L[r];
        const uint32_t raw = max(max(dg + tv, up), left + c_horiz);
        const uint32_t cell = __builtin_amdgcn_bitop3_b32(raw, c_keep, c_one, 0xEA);
        dg = left;
        L[r] = cell;
        up = cell;
      }
      return up;
    };
    // Steps: at step u my column is u - sub (0-based).  The loop bound is the wavefront's largest la + sub.
    uint32_t n_steps = mine ? la + sub : 0u;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) n_steps = max(n_steps, (uint32_t)__shfl_xor((int)n_steps, d, 64));
    n_steps = (uint32_t)__builtin_amdgcn_readfirstlane((int)n_steps);
    // The residues of A for my columns u-sub .. u-sub+3 as one unaligned dword, asked for one group of steps ahead.
    auto residues = [&](uint32_t u) -> uint32_t {
      uint32_t v = 0u;
      if (mine) {
        if (u >= sub) { if (u - sub < la) __builtin_memcpy(&v, pa + (u - sub), 4); }
        else if (la && sub - u <= 3u) { __builtin_memcpy(&v, pa, 4); v <<= 8u * (sub - u); }  // my first column falls into this group
      }
      return v;
    };
    // the first lane of a pair: row 0 above its strip, and A's self scores (the other lanes add from a table of zeros)
    const bool first = sub == 0u;
    const int32_t* __restrict__ sd = selfdiag + (first ? 0 : 32);
    uint32_t res = K;                                      // no columns or no rows: score 0, no steps
    uint32_t ad_cur = residues(0u);
    uint32_t bottom = K;
    // Per column beside the cells: one counter, one compare shared by the residue select and the corner pick (res follows
    // the bottom cell while my columns last: what it holds at the end is column la's, and only the pair's last strip reads
    // it), row 0 as a running value (M[0][j] = -j: one subtraction per step; only a pair's first lane uses it).
    uint32_t c = 0u - sub;                                 // wraps for the steps in front of my first column
    uint32_t row0 = ((bias - 1u) << kPkScore) | c_one;     // (first lanes: c == the step)
    const uint32_t row0_step = 1u << kPkScore;
    for (uint32_t u4 = 0; u4 < n_steps; u4 += 4u) {
      const uint32_t ad_nxt = residues(u4 + 4u);
#pragma unroll
      for (uint32_t q = 0; q < 4u; ++q) {
        const uint32_t above = from_lane_below(bottom);
        const bool valid = c < cols;
        const uint32_t a = valid ? (ad_cur >> (8u * q)) & 0xFFu : (uint32_t)kAA;  // (kAA: the null residue)
        const uint32_t top = first ? row0 : above;
        selfscore += sd[a];
        bottom = column(a, top);
        res = valid ? bottom : res;
        row0 -= row0_step;
        ++c;
      }
      ad_cur = ad_nxt;
    }
    // the corner cell sits in the pair's last strip; the self scores are summed over the pair's lanes
    int32_t self_all = 0;
#pragma unroll
    for (int k = 0; k < G; ++k) self_all += __shfl(selfscore, (int)(lane - sub) + k, 64);
    if (have && (nstrips ? sub + 1u == nstrips : sub == 0u)) {
      const int32_t selfscore_all = self_all;
      const int32_t mutual = (int32_t)(res >> kPkScore) - (int32_t)(bias + lb);
      const uint32_t ndiag = res & kDiagMask, nmatch = kWide ? 0u : (res >> kPkMatch) & 0x1FFu;
      const uint32_t gap = (la - ndiag) + (lb - ndiag);
      const uint32_t mismatch = ndiag - nmatch;
      const uint32_t len = gap + nmatch + mismatch;                           // :227
      const float normfactor = (float)len / (float)selfscore_all;             // :228
      float distance = (float)(selfscore_all - 2 * mutual) * normfactor;      // :232
      const float similarity = (float)(2 * mutual) * normfactor;              // :233
      if (!(selfscore_all >= mutual) || !(distance >= 0.f)) distance = __builtin_nanf("");
      w.pair_res[res_index(w, dst)] = make_float2(distance, similarity);
      if (w.pair_ext) w.pair_ext[dst] = kWide ? make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, gap, len) : make_uint4(nmatch, mismatch, gap, len);
    }
  }
}

// One kernel per class (one launch that dispatched on the class of each run was tried: the four inlined bodies cost the
// dominant class 15 % -- config 4: 4.1 -> 4.8 ms of alignment per step).
template <int R, int G, bool kWide>
__global__ __launch_bounds__(kPackedBlock) void k_nw_protein_lanes(const uint2* __restrict__ list, uint32_t n_list, WorkDev w,
                                                                   uint32_t* __restrict__ work_counter, uint32_t counter_first,
                                                                   unsigned long long* prof_acc) {
  constexpr uint32_t kPkScore = kWide ? 18u : rpa::kPkScore, kPkMove = kWide ? 16u : rpa::kPkMove, kPkMoveMask = 3u << kPkMove, kPkMatch = rpa::kPkMatch;
  __shared__ uint32_t tab[kAX * kAX * 32];  // [b][a][bank] -> T(a,b) as in K3p; T(null,b) = 0; T(a,padding) = "score - 1, move 0"
  __shared__ int32_t selfdiag[64];          // s(x,x); [27..31] = 0 (null residue), [32..63] = 0 (lanes that do not count A)
  __shared__ int8_t blosum[kAA * kAA];
  if (threadIdx.x < kAA * kAA) blosum[threadIdx.x] = c_blosum62[threadIdx.x];
  __syncthreads();
  for (int idx = threadIdx.x; idx < kAX * kAX * 32; idx += kPackedBlock) {
    const int bres = idx / (kAX * 32), a = (idx / 32) % kAX;
    const int sc = a < kAA && bres < kAA ? blosum[a * kAA + bres] : 0;
    tab[idx] = bres == kAA ? 0u - (1u << kPkScore) - (1u << kPkMove)
               : a < kAA ? ((uint32_t)(sc + 1) << kPkScore) + (1u << kPkMove) + ((!kWide && a == bres) ? 1u << kPkMatch : 0u) + 1u : 0u;
  }
  if (threadIdx.x < 64) selfdiag[threadIdx.x] = threadIdx.x < kAA ? blosum[threadIdx.x * kAA + threadIdx.x] : 0;
  __syncthreads();
  const uint8_t* __restrict__ tabb = reinterpret_cast<const uint8_t*>(tab);
  uint32_t c_horiz = 0u - (1u << kPkScore) - (1u << kPkMove);  // score - 1, move 1 -> 0
  uint32_t c_keep = ~kPkMoveMask, c_one = 1u << kPkMove;
  asm volatile("" : "+v"(c_horiz), "+v"(c_keep), "+v"(c_one));
  // Runs of 64 / G pairs: the first one of a wavefront is fixed (one run per CU before any CU gets a second), the
  // following ones come from the counter, which starts behind the fixed runs and only grows -- every wavefront leaves
  // through the break.
  constexpr uint32_t kPairs = 64u / G;
  uint32_t base = ((threadIdx.x >> 6) * gridDim.x + blockIdx.x) * kPairs;
  for (;;) {
    if (base >= n_list) break;
    const uint32_t run = base;
    if ((threadIdx.x & 63u) == 0u) base = atomicAdd(work_counter, kPairs) + counter_first;  // (the word is zero when the launch begins, see K3p)
    base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
    protein_lanes_run<R, G, kWide>(list, n_list, run, w, tabb, selfdiag, c_horiz, c_keep, c_one, prof_acc);
  }
}

size_t protein_scratch_words(uint32_t n_pairs, uint32_t max_len) {
  size_t threads = ((size_t)n_pairs + kPackedBlock - 1) / kPackedBlock * kPackedBlock;
  const size_t cap = (size_t)1 << 20;  // at most 1 Mi alignments in flight per launch
  if (threads > cap) threads = cap;
  return threads * 2 * ((size_t)max_len + 2);
}

// Strip height per work list: pairs are binned by the R in {8,16,24,32} that wastes the fewest rows for their vertical
// length (67 aa: 3 strips of 24 = 72 rows instead of 3 x 32 = 96).
hipError_t launch_align_protein(hipStream_t st, int strip_rows, const uint2* list, uint32_t n, const WorkDev& w, uint32_t* scratch,
                                size_t scratch_words, uint32_t max_len, unsigned long long* prof_acc) {
  if (n == 0) return hipSuccess;
  hipError_t e = upload_blosum();
  if (e != hipSuccess) return e;
  // counters of a path fit 16 bits each
  if (max_len > 65535u) return hipErrorInvalidValue;
  const size_t per_thread_words = 2 * ((size_t)max_len + 2);
  size_t threads = scratch_words / per_thread_words;
  threads = threads / kBlock * kBlock;
  if (threads == 0) return hipErrorOutOfMemory;
  for (uint32_t done = 0; done < n;) {
    const uint32_t chunk = (uint32_t)((size_t)(n - done) < threads ? (n - done) : threads);
    const uint32_t blocks = (chunk + kBlock - 1) / kBlock;
#define RPA_LAUNCH_NW(RV) \
  case RV: hipLaunchKernelGGL(k_nw_protein<RV>, dim3(blocks), dim3(kBlock), 0, st, list + done, chunk, w, reinterpret_cast<int2*>(scratch), blocks * kBlock, prof_acc); break;
    switch (strip_rows) {
      RPA_LAUNCH_NW(8) RPA_LAUNCH_NW(16) RPA_LAUNCH_NW(24) RPA_LAUNCH_NW(32)
      default: return hipErrorInvalidValue;
    }
#undef RPA_LAUNCH_NW
    done += chunk;
  }
  return hipGetLastError();
}

// K3p over a list of pairs that protein_bin() found eligible (protein_fits_packed).  One persistent
// workgroup per CU; the boundary rows need (max_len + 1) words per resident lane.
hipError_t launch_align_protein_packed(hipStream_t st, const uint2* list, uint32_t n, const WorkDev& w, uint32_t* scratch,
                                       size_t scratch_words, uint32_t max_len, uint32_t* work_counter, unsigned long long* prof_acc, bool wide) {
  if (n == 0) return hipSuccess;
  hipError_t e = upload_blosum();
  if (e != hipSuccess) return e;
  if (max_len > (wide ? kWideMaxLen : kPackedMaxLen)) return hipErrorInvalidValue;
  static int cus[64] = {0};
  int dev = 0;
  e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  int n_cu = dev < 64 ? cus[dev] : 0;
  if (!n_cu) {
    e = hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev);
    if (e != hipSuccess) return e;
    if (dev < 64) cus[dev] = n_cu;
  }
  // one workgroup per CU as long as there is a run of 64 pairs for it: a short list is spread over the chip (one busy
  // wavefront per CU runs ~3.5x faster than four per SIMD); wavefronts that find the counter exhausted leave at once
  uint32_t blocks = (n + 63u) / 64u;
  if (blocks > (uint32_t)n_cu) blocks = (uint32_t)n_cu;
  const size_t per_lane_words = ((size_t)max_len + 3) / 4 * 4;  // whole groups of four columns
  while (blocks > 1 && kPackedScratchHead + (size_t)blocks * kPackedBlock * per_lane_words > scratch_words) --blocks;
  if (kPackedScratchHead + (size_t)blocks * kPackedBlock * per_lane_words > scratch_words) return hipErrorOutOfMemory;
  const uint32_t first = blocks * (kPackedBlock / 64) * 64u;  // the counter's runs start behind the fixed first ones
  if (wide) hipLaunchKernelGGL((k_nw_protein_packed<kPackedRows, true>), dim3(blocks), dim3(kPackedBlock), 0, st, list, n, w, scratch, blocks * kPackedBlock, work_counter, first, prof_acc);
  else hipLaunchKernelGGL((k_nw_protein_packed<kPackedRows, false>), dim3(blocks), dim3(kPackedBlock), 0, st, list, n, w, scratch, blocks * kPackedBlock, work_counter, first, prof_acc);
  return hipGetLastError();
}

// K3l over the list of a lanes-per-pair class (protein_bin(): the vertical sequence needs `lanes` strips of
// kProteinLaneRows).  Persistent workgroups as in K3p; `work_counter` is one word of device memory that is zero when the launch begins.
hipError_t launch_align_protein_lanes(hipStream_t st, int lanes, const uint2* list, uint32_t n, const WorkDev& w, uint32_t* work_counter,
                                      unsigned long long* prof_acc, bool wide) {
  if (n == 0) return hipSuccess;
  hipError_t e = upload_blosum();
  if (e != hipSuccess) return e;
  if (wide ? !(lanes == 16 || lanes == 21 || lanes == 32 || lanes == 64) : (lanes < 1 || lanes > kProteinMaxLanes)) return hipErrorInvalidValue;
  static int cus[64] = {0};
  int dev = 0;
  e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  int n_cu = dev < 64 ? cus[dev] : 0;
  if (!n_cu) {
    e = hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev);
    if (e != hipSuccess) return e;
    if (dev < 64) cus[dev] = n_cu;
  }
  const uint32_t per_wave = 64u / (uint32_t)lanes;
  uint32_t blocks = (n + per_wave - 1u) / per_wave;
  if (blocks > (uint32_t)n_cu) blocks = (uint32_t)n_cu;
  const uint32_t first = blocks * (kPackedBlock / 64) * per_wave;  // behind the fixed first runs
  if (wide) {
#define RPA_LAUNCH_WIDE_LANES(GV) \
  case GV: hipLaunchKernelGGL((k_nw_protein_lanes<kProteinLaneRows, GV, true>), dim3(blocks), dim3(kPackedBlock), 0, st, list, n, w, work_counter, first, prof_acc); break;
    switch (lanes) {
      RPA_LAUNCH_WIDE_LANES(16) RPA_LAUNCH_WIDE_LANES(21) RPA_LAUNCH_WIDE_LANES(32) RPA_LAUNCH_WIDE_LANES(64)
      default: return hipErrorInvalidValue;
    }
#undef RPA_LAUNCH_WIDE_LANES
    return hipGetLastError();
  }
#define RPA_LAUNCH_LANES(GV) \
  case GV: hipLaunchKernelGGL((k_nw_protein_lanes<kProteinLaneRows, GV, false>), dim3(blocks), dim3(kPackedBlock), 0, st, list, n, w, work_counter, first, prof_acc); break;
  switch (lanes) {
    RPA_LAUNCH_LANES(1) RPA_LAUNCH_LANES(2) RPA_LAUNCH_LANES(3) RPA_LAUNCH_LANES(4) RPA_LAUNCH_LANES(5) RPA_LAUNCH_LANES(6) RPA_LAUNCH_LANES(7)
    RPA_LAUNCH_LANES(8) RPA_LAUNCH_LANES(9) RPA_LAUNCH_LANES(10) RPA_LAUNCH_LANES(11) RPA_LAUNCH_LANES(12) RPA_LAUNCH_LANES(13)
    default: return hipErrorInvalidValue;
  }
#undef RPA_LAUNCH_LANES
  return hipGetLastError();
}
static_assert(kProteinMaxLanes == 13 && kProteinLaneClasses == 13, "launch_align_protein_lanes dispatches the classes of protein_class_lanes()");

}  // namespace rpa
