// taxator-tk_amd/csrc/align_protein_kernels.hip -- protein re-alignment kernel (K3) for gfx950.
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
    __shared__ unsigned long long red[3][kBlock / 64];
    unsigned long long v0 = active ? 1ull : 0ull, v1 = active ? (unsigned long long)la + lb + 8ull : 0ull,
                       v2 = active ? (unsigned long long)la * lb : 0ull;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { v0 += __shfl_down(v0, d, 64); v1 += __shfl_down(v1, d, 64); v2 += __shfl_down(v2, d, 64); }
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = v0; red[1][threadIdx.x >> 6] = v1; red[2][threadIdx.x >> 6] = v2; }
    __syncthreads();
    if (threadIdx.x < 3) {
      unsigned long long s = 0;
      for (int k = 0; k < kBlock / 64; ++k) s += red[threadIdx.x][k];
      atomicAdd(&prof_acc[(blockIdx.x & (kProfParts - 1)) * 4 + threadIdx.x], s);
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

size_t protein_scratch_words(uint32_t n_pairs, uint32_t max_len) {
  size_t threads = ((size_t)n_pairs + kBlock - 1) / kBlock * kBlock;
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

}  // namespace rpa
