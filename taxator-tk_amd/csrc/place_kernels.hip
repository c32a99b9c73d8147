// taxator-tk_amd/csrc/place_kernels.hip -- record-set planning and the RPA placement state machine (K4) for gfx950.
//
// What it replaces: the bookkeeping of RPAPredictionModel::predict (core/src/taxonpredictionmodelsequence.hh:341-838),
// BandFactor (:259-323), SortFilter (core/src/alignmentsfilter.hh:171-190 with AlignmentRecord::operator<,
// core/src/alignmentrecord.hh:89-93), the segment coordinate rules of getSequence* (:856-880 on top of
// core/src/sequencestorage.hh:341-369) and TaxonomyInterface::getLCA / isParentOf (core/src/taxonomyinterface.cpp:52-77).
//
// The reference runs predict() start to finish for one record set, calling the aligner from inside its loops.
// Here a record set is a small resumable state machine owned by one thread: every round it (1) consumes the
// alignment results it asked for, replaying the reference's loop for the current pass/anchor exactly -- including
// its early exits, so an alignment that was computed speculatively but that the reference would not have reached is
// neither read nor counted -- and (2) emits the pairs the next loop may need (a superset: everything the loop could
// reach given what is known before it starts).  Between rounds all emitted pairs of all record sets are aligned by
// the batched kernels of align_*_kernels.hip.
//
// Memory shape (this is what makes a thread-per-set kernel viable on a GPU): a set's hits are ONE contiguous array of
// fixed-size rows in SortFilter order; a row carries everything the loops read per hit, including the hit's taxon
// PATH.  Every node predict() ever forms -- rtax, lnode, unode, lca_allnodes and their global versions -- is
// an LCA that includes the best reference's taxon, i.e. a point on the best hit's root path, so the state machine
// keeps DEPTHS on that path instead of nodes, and every getLCA / isParentOf of the reference becomes
//     common(x, i) = length of the common prefix of two paths,   LCA(point d on x's path, node_i) = min(d, common(x, i)),
// pure register arithmetic with no dependent walks over the taxonomy arrays.
// A path is not stored as node ids (4 bytes per level, and a row then weighs 64 bytes at 8 levels) but as per-set LABELS:
//     label_d(i) = the smallest index (input order inside the record set) of a record whose ancestor at depth d is hit i's.
// Two hits of a set share their ancestor at depth d exactly when their labels at level d are equal, so the common prefix
// of two label strings is the depth of the LCA, as with ids -- but a label needs 8 bits (sets of up to 256 records; 16 /
// 32 bits beyond), not 32: 8 levels fit 8 bytes and a row is 32 bytes, a 40-level tree costs 40 bytes.  k_rank_records
// makes them without touching another record's path: record j shares hit i's ancestor at depth d iff j's node lies in
// that ancestor's subtree, i.e. iff pre[node_j] falls into the ancestor's preorder interval (TaxDev::anc_iv), so a thread
// holds its own node's intervals and scans the preorder numbers of the records in front of it (staged in LDS).
// The node at a depth of the best hit's path, needed only when a set's result is written, comes from TaxDev::anc.
// Alignment results come back in a per-hit array parallel to the rows, so a round reads ~40 contiguous bytes per hit.
//
// Float discipline (bit-exact decisions): f32 distances/similarities/scores, f64 percent identities, int truncations
// exactly where the reference has them (:592,602,659,661); compiled with -ffp-contract=off and IEEE divide/sqrt.
#include <float.h>
#include <limits.h>
#include <stdlib.h>

#include <type_traits>
#include "rpa_device.h"

namespace rpa {

struct OutRow {  // = rpa_prediction
  uint32_t lower, upper, rtax, support, begin, end;
  float ival;
  uint32_t flags;
};
struct OutStats {  // = rpa_stats
  uint32_t n, pass0, pass1, pass2;
};
constexpr uint32_t kFlagIvalUntouched = 1u, kFlagShortcut = 2u, kFlagSingle = 4u;

__device__ __forceinline__ bool rec_filtered(const BatchDev& b, uint32_t r) { return b.filtered && b.filtered[r]; }

// `int x = <float expression>` as the reference's platform computes it (:602,657-659 `qsearchscore_thresh_heuristic = score * X`):
// x86-64's cvttss2si truncates and answers the "integer indefinite" 0x80000000 for a NaN and for anything outside the int range;
// v_cvt_i32_f32 would saturate (INT_MAX for 3e9, 0 for a NaN).
__device__ __forceinline__ int float_to_int_x86(float v) { return (v >= -2147483648.f && v < 2147483648.f) ? (int)v : INT_MIN; }

__device__ __forceinline__ void raise_error(const WorkDev& w, uint32_t code, uint32_t set) {
  if (atomicCAS(&w.counters[kCtrErr], 0u, code) == 0u) w.counters[kCtrErrSet] = set;
}

// ---- hit rows -------------------------------------------------------------------------------------------------
// Where the rows of one record set live.  Tiled layout (WorkDev::tile_base set): 64 consecutive record sets form a
// tile; row i, quad q of the tile's set number l sits at quad index (tile_base + 64 i) Q + 64 q + l, so when the 64
// lanes of a wave (= 64 consecutive sets) read "quad q of my row i" they read 1 KB of contiguous memory, and the
// alignment result of row i at slot tile_base + 64 i + l likewise.  Linear layout (fallback for batches whose sets
// differ wildly in size, where padding every tile to its largest set would waste memory): rows of a set contiguous.
template <int Q>
struct Row { uint4 q[Q]; };  // a whole row in registers
template <int Q> constexpr int path_words() { return 4 * Q - 6; }

template <int Q>
struct SetRows {
  uint4* p;         // quad 0 of row 0
  size_t si, sq;    // quads from a row to the next / from a quad of a row to the next
  float2* res;      // result slot of row 0
  size_t sr;        // slots from a row to the next
  __device__ __forceinline__ Row<Q> load(uint32_t i) const {
    const uint4* r = p + (size_t)i * si;
    Row<Q> o;
#pragma unroll
    for (int k = 0; k < Q; ++k) o.q[k] = r[(size_t)k * sq];
    return o;
  }
  __device__ __forceinline__ HitCore core(uint32_t i) const {  // 24 bytes: quad 0 + the first half of quad 1
    const uint4* r = p + (size_t)i * si;
    const uint4 a = r[0];
    const float2 b = reinterpret_cast<const float2*>(r + sq)[0];
    HitCore c;
    c.score = __uint_as_float(a.x); c.ident = a.y; c.node = a.z; c.meta = a.w;
    c.qd = b.x; c.qs = b.y;
    return c;
  }
  __device__ __forceinline__ float score(uint32_t i) const { return __uint_as_float(reinterpret_cast<const uint32_t*>(p + (size_t)i * si)[0]); }
  __device__ __forceinline__ uint32_t meta(uint32_t i) const { return reinterpret_cast<const uint32_t*>(p + (size_t)i * si)[3]; }
  __device__ __forceinline__ void path(uint32_t i, uint32_t (&path)[4 * Q - 6]) const {
    const uint4* r = p + (size_t)i * si;
    const uint2 h = reinterpret_cast<const uint2*>(r + sq)[1];
    path[0] = h.x; path[1] = h.y;
#pragma unroll
    for (int k = 2; k < Q; ++k) {
      const uint4 v = r[(size_t)k * sq];
      path[4 * k - 6] = v.x; path[4 * k - 5] = v.y; path[4 * k - 4] = v.z; path[4 * k - 3] = v.w;
    }
  }
  // the only mutable fields of a row: querydistance / querysimilarity (first 8 bytes of its second quad)
  __device__ __forceinline__ void store_qdqs(uint32_t i, float qd, float qs) const {
    reinterpret_cast<float2*>(p + (size_t)i * si + sq)[0] = make_float2(qd, qs);
  }
  __device__ __forceinline__ void store_qd(uint32_t i, float qd) const { reinterpret_cast<float*>(p + (size_t)i * si + sq)[0] = qd; }
  __device__ __forceinline__ float2 result(uint32_t i) const { return res[(size_t)i * sr]; }
};
// (t0 = w.tile_base[s >> 6], loaded by the caller: k_decide asks for it together with the set's state, not behind it)
template <int Q>
__device__ __forceinline__ SetRows<Q> set_rows(const WorkDev& w, uint32_t s, uint32_t lo, uint32_t tile0) {
  SetRows<Q> r;
  uint4* base = reinterpret_cast<uint4*>(w.hits);
  if (w.tile_base) {
    const size_t t0 = tile0, l = s & 63u;
    r.p = base + t0 * Q + l; r.si = 64 * Q; r.sq = 64;
    r.res = w.pair_res + t0 + l; r.sr = 64;
  } else {
    r.p = base + (size_t)lo * Q; r.si = Q; r.sq = 1;
    r.res = w.pair_res + lo; r.sr = 1;
  }
  return r;
}
template <int Q>
__device__ __forceinline__ SetRows<Q> set_rows(const WorkDev& w, uint32_t s, uint32_t lo) {
  SetRows<Q> r;
  uint4* base = reinterpret_cast<uint4*>(w.hits);
  if (w.tile_base) {
    const size_t t0 = w.tile_base[s >> 6], l = s & 63u;
    r.p = base + t0 * Q + l; r.si = 64 * Q; r.sq = 64;
    r.res = w.pair_res + t0 + l; r.sr = 64;
  } else {
    r.p = base + (size_t)lo * Q; r.si = Q; r.sq = 1;
    r.res = w.pair_res + lo; r.sr = 1;
  }
  return r;
}
template <int Q>
__device__ __forceinline__ HitCore row_core(const Row<Q>& r) {
  HitCore c;
  c.score = __uint_as_float(r.q[0].x); c.ident = r.q[0].y; c.node = r.q[0].z; c.meta = r.q[0].w;
  c.qd = __uint_as_float(r.q[1].x); c.qs = __uint_as_float(r.q[1].y);
  return c;
}
template <int Q>
__device__ __forceinline__ void row_path(const Row<Q>& r, uint32_t (&path)[4 * Q - 6]) {
  path[0] = r.q[1].z; path[1] = r.q[1].w;
#pragma unroll
  for (int k = 2; k < Q; ++k) {
    path[4 * k - 6] = r.q[k].x; path[4 * k - 5] = r.q[k].y; path[4 * k - 4] = r.q[k].z; path[4 * k - 3] = r.q[k].w;
  }
}
__device__ __forceinline__ uint32_t meta_depth(uint32_t meta) { return (meta >> kMetaDepthShift) & 0xFFu; }
__device__ __forceinline__ uint32_t meta_err(uint32_t meta) { return (meta >> kMetaErrShift) & 0xFu; }
// qgroup / outgroup bit sets of one record set (WorkDev::flags)
__device__ __forceinline__ void bit_clear(uint32_t* words, uint32_t i) { words[i >> 5] &= ~(1u << (i & 31u)); }
__device__ __forceinline__ void bit_set(uint32_t* words, uint32_t i) { words[i >> 5] |= 1u << (i & 31u); }
// One of the two bit sets as k_decide sees it: for record sets of up to 32 hits the single word lives in a REGISTER and
// travels in the set's state record (round 4) -- every read of a group word used to be a trip to memory of its own behind
// the set's offset, every membership change of the outgroup loop a read-modify-write (profiles/r04_decide_phases.txt: the
// kernel is bound by the number of such trips); larger sets keep the words in WorkDev::flags.  Indexed like an array.
struct GroupWords {
  uint32_t* mem;  // null: the word is `reg`
  uint32_t reg;
  struct Ref {
    GroupWords& g; uint32_t k;
    __device__ __forceinline__ operator uint32_t() const { return g.mem ? g.mem[k] : g.reg; }
    __device__ __forceinline__ Ref& operator=(uint32_t v) { if (g.mem) g.mem[k] = v; else g.reg = v; return *this; }
    __device__ __forceinline__ Ref& operator&=(uint32_t v) { if (g.mem) g.mem[k] &= v; else g.reg &= v; return *this; }
    __device__ __forceinline__ Ref& operator|=(uint32_t v) { if (g.mem) g.mem[k] |= v; else g.reg |= v; return *this; }
  };
  __device__ __forceinline__ Ref operator[](uint32_t k) { return Ref{*this, k}; }
};
__device__ __forceinline__ void bit_clear(GroupWords& words, uint32_t i) { words[i >> 5] &= ~(1u << (i & 31u)); }
__device__ __forceinline__ void bit_set(GroupWords& words, uint32_t i) { words[i >> 5] |= 1u << (i & 31u); }

// length of the common prefix of two label paths = depth of the LCA of the two nodes: the first bit in which the strings
// differ, in labels (lb_shift = log2 of the label width), never more than either node's own depth
template <int PW>
__device__ __forceinline__ uint32_t common(const uint32_t (&a)[PW], uint32_t da, const uint32_t (&b)[PW], uint32_t db, uint32_t lb_shift) {
  uint32_t bit = 32u * PW;
#pragma unroll
  for (int k = PW - 1; k >= 0; --k) {
    const uint32_t x = a[k] ^ b[k];
    if (x) bit = 32u * (uint32_t)k + (uint32_t)__builtin_ctz(x);
  }
  return min(bit >> lb_shift, min(da, db));
}
// point `depth` on the root path of `node` (depth <= depth of node; 0 = the root)
__device__ __forceinline__ uint32_t anc_node(const TaxDev& tax, uint32_t node, uint32_t depth) {
  return depth == 0u ? kRoot : tax.anc[(size_t)node * tax.max_depth + (depth - 1u)];
}

// ------------------------------------------------------------------------------------------------------------
// Segment planning (a7).  store.getSequence(id, start, stop): stop clipped to the sequence length, start beyond it is
// a range error (sequencestorage.hh:341-362); getSequenceDNA (:870-880): forward [rstart-left_ext, rstop+right_ext]
// with the start floored at 1; reverse (rstart > rstop) [rstop-right_ext, rstart+left_ext] reverse-complemented.
// Protein with reverse coordinates trips a live assert in the reference (:864).  Returns (length, error code).
// ------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ SegDesc plan_segment_desc(const StoreDev& store, bool from_query, uint32_t seq, uint32_t start, uint32_t stop,
                                                     bool rc, uint2& len_err) {
  uint32_t err = 0, len = 0;
  uint64_t src = 0;
  if (seq >= store.n_seqs) err = kErrSeqId;
  else {
    const ulonglong2 ex = store.extent[seq];
    const uint64_t off = store.windows ? (ex.x & ((1ull << kWinFirstBits) - 1ull)) : ex.x, slen = ex.y;
    if (start > stop) err = kErrAssert;
    else {
      if ((uint64_t)stop > slen) stop = (uint32_t)slen;
      if (start < 1 || start > stop) err = kErrRange;
      else if (!store.windows) { len = stop - start + 1; src = off + (start - 1); }
      else {
        // windowed store (the region-granular reads of the reference's indexed store, sequencestorage.hh:341-362): the
        // window that holds residue start-1 must hold the whole segment
        const uint32_t nw = (uint32_t)(ex.x >> kWinFirstBits), a = start - 1u, n_seg = stop - start + 1u;
        const uint4* __restrict__ wv = store.windows + off;
        uint32_t lo = 0, hi = nw;  // largest k with wv[k].x <= a
        while (hi - lo > 1u) {
          const uint32_t mid = (lo + hi) >> 1;
          if (wv[mid].x <= a) lo = mid; else hi = mid;
        }
        uint4 win = make_uint4(1u, 0u, 0u, 0u);
        if (nw) win = wv[lo];
        if (nw && win.x <= a && (uint64_t)a + n_seg <= (uint64_t)win.x + win.y) {
          len = n_seg;
          src = (((uint64_t)win.w << 32) | win.z) + (a - win.x);
        } else err = kErrNotResident;
      }
    }
  }
  len_err = make_uint2(len, err);
  return SegDesc{(uint32_t)src, (uint32_t)(src >> 32), len, (rc ? 1u : 0u) | (from_query ? 2u : 0u) | (err << 4)};
}
__device__ __forceinline__ uint2 plan_segment(const WorkDev& w, uint32_t seg, const StoreDev& store, bool from_query,
                                              uint32_t seq, uint32_t start, uint32_t stop, bool rc) {
  uint2 le;
  w.seg[seg] = plan_segment_desc(store, from_query, seq, start, stop, rc, le);
  return le;
}

// ------------------------------------------------------------------------------------------------------------
// Rank + gather: one thread per record.  It scans the records of its own set once (neighbouring lanes read the same
// few cache lines) to get (a) its position in SortFilter order (score desc, identities desc, input order on full ties
// = what a stable list::sort gives) among the unmasked records, and (b) the set's n / query range (:391-404).  It
// plans its own reference segment, makes its taxon path labels (file header) from its node's ancestor intervals and the
// preorder numbers of the records in front of it in its set, then writes its row at the sorted position.  The thread of a
// set's first record also plans the query-range segment and writes the set header.
// ------------------------------------------------------------------------------------------------------------
#ifndef RPA_RANK_WAVES
#define RPA_RANK_WAVES 1  // (A/B of the register budget: tools/build_variant.sh; 1 = whatever 256 threads allow)
#endif
template <int Q>
__global__ __launch_bounds__(kBlock, RPA_RANK_WAVES) void k_rank_records(BatchDev b, WorkDev w, TaxDev tax, StoreDev qstore, StoreDev dbstore, Params p) {
  // Record -> record set.  The records of a workgroup are consecutive, so their sets are consecutive too: the set of
  // the block's first record comes from k_block_first (one binary search per workgroup, all of them in one small launch
  // instead of 4 dependent probes at the head of every workgroup), the block stages the next kBlock+1 offsets in LDS and
  // every thread finishes with a search in LDS.  Only a block whose range holds more than kBlock set boundaries (runs
  // of empty sets) falls back to the global search.  "Set of r" = largest s with set_offsets[s] <= r (empty sets share
  // an offset: take the last).
  __shared__ uint32_t s_off[kBlock + 1];
  // the fields the per-set scan below compares, for the block's own records: the scan of a 20-hit set is 20 x 4 loads
  // per thread, nearly all of them to records of this block -> LDS (broadcast reads) instead of the vector memory path
  // (+ kHalo records on either side: the sets that straddle the block's ends would otherwise take the global path for
  // up to a whole set's worth of iterations -- 45 vector loads per wave on config 2 instead of 15)
  constexpr uint32_t kHalo = 64, kWin = kBlock + 2 * kHalo;
  __shared__ uint4 s_rec[kWin];      // score bits, identities, qstart, qstop
  __shared__ uint32_t s_pre[kWin];   // preorder number of the record's taxon
  __shared__ uint8_t s_filt[kWin];
  // Rows are written in the tiled layout: the 64-byte row of a record lands 16 bytes at a time in lines it shares with
  // the same row of the 63 neighbouring record sets, i.e. with records handled by this and the next few workgroups.
  // Workgroups are dealt round-robin over the 8 XCDs (each with its own L2), so consecutive LOGICAL blocks are mapped
  // to hardware blocks b, b+8, b+16, ...: the workgroups that complete each other's lines share an L2.
  const uint32_t per_xcd = gridDim.x / 8u;  // the grid is a multiple of 8
  const uint32_t lblock = (blockIdx.x % 8u) * per_xcd + blockIdx.x / 8u;
  const uint32_t r0 = lblock * kBlock;
  if (r0 >= b.n_records) return;  // padding blocks of the grid (whole workgroup)
  const uint32_t w0 = r0 >= kHalo ? r0 - kHalo : 0u;  // first record of the staged window
  for (uint32_t k = threadIdx.x; k < kWin; k += kBlock) {
    const uint32_t rr = w0 + k;
    if (rr < b.n_records) {
      // (a masked record sits in the window with values that change nothing in the scan of a staged set below: a NaN score --
      // never "before" anybody, skipped by the maximum -- and an empty query range)
      const bool filt = rec_filtered(b, rr);
      s_rec[k] = filt ? make_uint4(0x7FC00000u, b.identities[rr], 0xFFFFFFFFu, 0u)
                      : make_uint4(__float_as_uint(b.score[rr]), b.identities[rr], b.qstart[rr], b.qstop[rr]);
      s_pre[k] = w.rec_pre[rr];  // preorder number of the record's taxon, left by k_validate_batch
      s_filt[k] = filt ? 1 : 0;
    }
  }
  const uint32_t first = w.block_first[lblock];
  for (uint32_t k = threadIdx.x; k <= (uint32_t)kBlock; k += kBlock) {
    const uint64_t idx = (uint64_t)first + k;
    s_off[k] = b.set_offsets[idx < b.n_sets ? idx : b.n_sets];
  }
  __syncthreads();
  const uint32_t r = r0 + threadIdx.x;
  // what this thread's record turns into: its row (Q quads, staged in LDS) at row slot `slot` (0xFFFFFFFF: none -- past
  // the end of the batch, or a masked record); written out by the whole workgroup below
  constexpr int PW = 4 * Q - 6;
  __shared__ uint4 s_rows[kBlock * Q];
  uint32_t* const my_row = reinterpret_cast<uint32_t*>(&s_rows[threadIdx.x * Q]);
  uint32_t slot = 0xFFFFFFFFu, my_set = 0, my_rank = 0;
  bool staged_set = false;  // my set's offsets are in the LDS window
  if (r < b.n_records) {
  uint32_t lo = 0, hi = kBlock + 1;  // largest j in [0, kBlock] with s_off[j] <= r  (s_off[0] <= r0 <= r)
  while (hi - lo > 1) {
    const uint32_t mid = (lo + hi) >> 1;
    if (s_off[mid] <= r) lo = mid; else hi = mid;
  }
  uint32_t s = first + lo;
  bool in_window = lo < (uint32_t)kBlock;  // the set's two offsets are s_off[lo], s_off[lo + 1]
  if (lo == (uint32_t)kBlock && s < b.n_sets) {  // window exhausted: more than kBlock boundaries in this block's range
    uint32_t glo = s, ghi = b.n_sets;
    while (ghi - glo > 1) {
      const uint32_t mid = glo + ((ghi - glo) >> 1);
      if (b.set_offsets[mid] <= r) glo = mid; else ghi = mid;
    }
    s = glo;
  }
  if (s >= b.n_sets) { s = b.n_sets - 1; in_window = false; }  // clamped window entries (cannot be selected: set_offsets[n_sets] > r)
  uint32_t beg, end;
  if (in_window) { beg = s_off[lo]; end = s_off[lo + 1]; }  // staged above: no second trip to memory
  else { beg = b.set_offsets[s]; end = b.set_offsets[s + 1]; }
  // (record sets of w.big_min ... kBigSetMax records are k_rank_big_sets' -- a workgroup per set, everything staged: the scans below are
  // one trip over the set PER RECORD, and a set that does not fit this workgroup's window makes every one of them a walk through memory)
  const bool big_set = w.big_min && end - beg >= w.big_min && end - beg <= kBigSetMax;
  if (!big_set) {
  const uint32_t me = r - w0;  // my record inside the window
  const bool me_filtered = s_filt[me] != 0;
  const float sc = __uint_as_float(s_rec[me].x);
  const uint32_t id = s_rec[me].y;
  uint32_t rank = 0, n = 0, qrstart = 0xFFFFFFFFu, qrstop = 0;
  float smax = 0.f;  // score of the set's best record (its first one in SortFilter order)
  auto take = [&](uint32_t j, const uint4 v, bool filt) {  // one record of my set (branch-free: masked records change nothing)
    const float sj = __uint_as_float(v.x);
    const uint32_t ij = v.y;
    const bool act = !filt;
    smax = act ? (n == 0 ? sj : (sj > smax ? sj : smax)) : smax;
    n += act ? 1u : 0u;
    qrstart = act ? min(qrstart, v.z) : qrstart;
    qrstop = act ? max(qrstop, v.w) : qrstop;
    // (bitwise, not short-circuit: the nested branches of `||` / `&&` cost ~14 scalar instructions per record)
    const bool before = (sj > sc) | ((sj == sc) & ((((uint64_t)ij << 32) | (uint32_t)~j) > (((uint64_t)id << 32) | (uint32_t)~r)));
    rank += (act && before) ? 1u : 0u;
  };
  if (beg >= w0 && end <= w0 + kWin) {
    // the whole set is staged (the usual case): a tight loop over LDS, four records in flight.  Per record: the two ends of the
    // query range, the largest score that is not a NaN, the first unmasked record, the count of masked ones, and three compares
    // for "before me" -- 12 instructions (27 with the running values of `take`, whose compound conditions had become branches:
    // the kernel is bound by instruction issue, 1.21 -> 1.12 ms on config 2).
    // The best score of :391-404 starts from the first unmasked record and a NaN there stays; any other NaN is passed over.
    const uint64_t my_key = ((uint64_t)id << 32) | (uint32_t)~r;
    uint32_t n_filt = 0, first_act = 0xFFFFFFFFu;
    float smax_nn = -INFINITY;
    auto scan = [&](uint32_t j, const uint4 v, uint32_t filt) {
      const float sj = __uint_as_float(v.x);
      n_filt += filt;
      qrstart = min(qrstart, v.z);
      qrstop = max(qrstop, v.w);
      smax_nn = __builtin_fmaxf(smax_nn, sj);
      first_act = min(first_act, filt ? 0xFFFFFFFFu : j);
      rank += (uint32_t)((sj > sc) | ((sj == sc) & ((((uint64_t)v.y << 32) | (uint32_t)~j) > my_key)));
    };
    uint32_t j = beg;
#ifdef RPA_STUB_RANK
    j = end;
#endif
    for (; j + 4u <= end; j += 4u) {
      const uint32_t jl = j - w0;
      const uint4 v0 = s_rec[jl], v1 = s_rec[jl + 1], v2 = s_rec[jl + 2], v3 = s_rec[jl + 3];
      const uint32_t f0 = s_filt[jl], f1 = s_filt[jl + 1], f2 = s_filt[jl + 2], f3 = s_filt[jl + 3];
      scan(j, v0, f0); scan(j + 1, v1, f1); scan(j + 2, v2, f2); scan(j + 3, v3, f3);
    }
    for (; j < end; ++j) scan(j, s_rec[j - w0], s_filt[j - w0]);
    n = (end - beg) - n_filt;
    if (n) {
      const float s_first = __uint_as_float(s_rec[first_act - w0].x);
      smax = isnan(s_first) ? s_first : smax_nn;
    }
  } else {
    for (uint32_t j = beg; j < end; ++j) {
      const uint32_t jl = j - w0;  // wraps for j < w0
      if (jl < kWin) take(j, s_rec[jl], s_filt[jl] != 0);
      else take(j, make_uint4(__float_as_uint(b.score[j]), b.identities[j], b.qstart[j], b.qstop[j]), rec_filtered(b, j));
    }
  }
  // A NaN score among the records that are sorted (n >= 2): the reference's list::sort (:412) then runs on a comparator that is
  // no ordering at all, and the order it leaves is an accident of libstdc++'s merge tree -- refused
  // instead of ranked into a row that another record owns.
  if (!me_filtered && n >= 2 && sc != sc) raise_error(w, kErrNanScore, s);
  // Every load of this thread comes before its first store: loads and stores share one counter on gfx9 and may complete
  // out of order with each other, so a load behind a store waits for the store as well.
  const bool set_head = r == beg;
  uint2 qp = make_uint2(0u, 0u);
  SegDesc qdesc{0u, 0u, 0u, 0u};
  if (set_head && n >= 2) qdesc = plan_segment_desc(qstore, true, b.query_seq[s], qrstart, qrstop, false, qp);  // :415
  if (!me_filtered) {
  const uint32_t pos = beg + rank;  // row position = segment id = result slot of this hit
  uint2 sp = make_uint2(0u, 0u);
  SegDesc sdesc{0u, 0u, 0u, 0u};
#ifdef RPA_STUB_SEG
  if (false) {
#else
  if (n >= 2) {
#endif
    const uint32_t left_ext = b.qstart[r] - qrstart, right_ext = qrstop - b.qstop[r];  // :523
    const uint32_t start = b.rstart[r], stop = b.rstop[r];
    if (start <= stop) {
      sdesc = plan_segment_desc(dbstore, false, b.ref_seq[r], left_ext < start ? start - left_ext : 1u, stop + right_ext, false, sp);
    } else if (p.alphabet != 0) {
      sdesc = SegDesc{0u, 0u, 0u, (uint32_t)kErrAssert << 4};
      sp = make_uint2(0u, (uint32_t)kErrAssert);
    } else {
      sdesc = plan_segment_desc(dbstore, false, b.ref_seq[r], right_ext < stop ? stop - right_ext : 1u, start + left_ext, true, sp);
    }
  }
  const uint32_t node = b.node[r];
  const uint32_t L = qrstop - qrstart + 1;
  const bool perfect = b.alnlen[r] == L && id == L;  // :431,511
  const uint32_t meta = (perfect ? kMetaPerfect : 0u) | (sp.y << kMetaErrShift) | ((uint32_t)tax.depth[node] << kMetaDepthShift);
  // :475-476 start at (FLT_MAX, 0); pass 0 then gives every hit it does not re-align (FLT_MAX, identities) before anything
  // reads the pair (:535-537) -- written here so that k_decide only stores the pairs of the hits pass 0 does align
  float qd = FLT_MAX, qs = (float)id;
  if (n == 1) { qd = __uint_as_float(b.qstart[r]); qs = __uint_as_float(b.qstop[r]); }  // n==1 needs them (:373-374)
  const SetRows<Q> R = set_rows<Q>(w, s, beg);
  slot = (uint32_t)((R.res + (size_t)rank * R.sr) - w.pair_res);  // row slot = result slot (tiled: tile_base + 64 rank + l; linear: beg + rank)
  // ---- taxon path labels: label_d = index (input order in the set) of the first record under my ancestor at depth d.
  // Eight levels at a time: my node's eight ancestor intervals in registers, one pass over the records in front of me.
  // Levels below my own depth keep my own index (nobody shares them; common() never looks past a node's depth).
#pragma unroll
  for (int k = 0; k < PW; ++k) my_row[6 + k] = 0xFFFFFFFFu;
  if (n >= 2) {
    const uint32_t D = tax.max_depth, my = r - beg;
    const uint4* __restrict__ iv4 = reinterpret_cast<const uint4*>(tax.anc_iv + (size_t)node * ((D + 7u) & ~7u));  // rows of whole 64-byte chunks
    const bool staged = beg >= w0;  // the records in front of me are all in the LDS window (r itself always is)
    for (uint32_t c = 0; c < D; c += 8u) {
      uint2 v[8];  // (first preorder number, last - first) of my ancestor at levels c .. c+7; (0xFFFFFFFF, 0) = no such ancestor
      uint32_t lab[8];
#pragma unroll
      for (int d = 0; d < 8; d += 2) {
        const uint4 two = iv4[(c + d) / 2u];
        v[d] = make_uint2(two.x, two.y); v[d + 1] = make_uint2(two.z, two.w);
        lab[d] = my; lab[d + 1] = my;
      }
      // back to front, so that the record nearest the front of the set has the last word: three vector instructions per
      // level and record (subtract, compare, select).  (Four records per trip, their LDS reads in flight together, measured
      // slower: 1.45 against 1.27 ms on config 2 -- the extra registers cost a resident wave.)
#ifdef RPA_STUB_LABELS
      for (uint32_t j = beg; j-- > beg;) {
#else
      for (uint32_t j = r; j-- > beg;) {
#endif
        const uint32_t pj = staged ? s_pre[j - w0] : w.rec_pre[j];
        const uint32_t idx = j - beg;
#pragma unroll
        for (int d = 0; d < 8; ++d) lab[d] = (pj - v[d].x <= v[d].y) ? idx : lab[d];
      }
      // pack the chunk's eight labels behind the chunks before it (levels past D are never compared: left as they are)
      if (w.lb_shift == 3u) {
        my_row[6 + c / 4u] = (lab[0] & 255u) | ((lab[1] & 255u) << 8) | ((lab[2] & 255u) << 16) | (lab[3] << 24);
        my_row[6 + c / 4u + 1u] = (lab[4] & 255u) | ((lab[5] & 255u) << 8) | ((lab[6] & 255u) << 16) | (lab[7] << 24);
      } else if (w.lb_shift == 4u) {
#pragma unroll
        for (int k = 0; k < 4; ++k) my_row[6 + c / 2u + k] = (lab[2 * k] & 0xFFFFu) | (lab[2 * k + 1] << 16);
      } else {
#pragma unroll
        for (int k = 0; k < 8; ++k) my_row[6 + c + k] = lab[k];
      }
    }
  }
  // ---- stores ----
  w.seg[pos] = sdesc;  // (all zero for a set of one record: never aligned)
  if (w.res_slot) w.res_slot[pos] = slot;  // where the alignment kernels put this hit's result (linear layout: the row position itself)
  if (sp.x > kLiteLenMask) { raise_error(w, kErrUnsupported, s); sp.x = 0; }
  // compact per-row word for k_emit: segment length + "pass 0 re-aligns this hit" (:511-518, threshold :499)
  w.lite[pos] = sp.x | ((n >= 2 && !perfect && sc >= p.reeval_factor * smax) ? kLiteWant0 : 0u) | (perfect ? kLitePerfect : 0u);
  my_row[0] = __float_as_uint(sc); my_row[1] = id; my_row[2] = node; my_row[3] = meta;
  my_row[4] = __float_as_uint(qd); my_row[5] = __float_as_uint(qs);
  my_set = s; my_rank = rank; staged_set = in_window;
  }  // !me_filtered
  if (set_head) {
    w.seg[b.n_records + s] = qdesc;
    if (qp.x > kLiteLenMask) { raise_error(w, kErrUnsupported, s); qp.x = 0; }
    w.set_info[s] = make_uint4(n, qrstart, qrstop, qp.x | (qp.y << 29));
  }
  w.hit_set[r] = s;  // rows and records of a set share the index range [set_off[s], set_off[s+1])
  }  // !big_set
  }  // r < n_records

  // ---- write the rows out.  Tiled layout: quad q of the row at slot x lives at quad index (x & ~63) Q + 64 q + (x & 63), i.e.
  // the same quad of the same rank of CONSECUTIVE record sets is contiguous.  A thread storing its own row's quads hits
  // 64 different lines per wave-level store (and there are Q of them per record: the kernel spent 0.5 of its 1.2 ms
  // there on config 2); instead the workgroup stages its rows in LDS and walks (rank, quad, set) so that consecutive lanes
  // write consecutive sets: runs of ~13 adjacent quads.  Batches whose workgroup windows do not fit (runs of empty sets,
  // sets far longer than a workgroup) and the linear layout keep the direct stores.
  __shared__ uint32_t s_slot[kBlock];
  __shared__ uint16_t s_who[kWin];      // (row position of the block's sets) - s_off[0]  ->  thread that holds the row
  __shared__ uint32_t s_lim[3];         // highest set, highest rank + 1, "some row cannot be staged"
  const uint32_t tid = threadIdx.x;
  for (uint32_t k = tid; k < kWin; k += kBlock) s_who[k] = 0xFFFFu;
  if (tid < 3) s_lim[tid] = tid == 0 ? first : 0u;
  __syncthreads();
  const bool tiled = w.tile_base != nullptr;
  if (slot != 0xFFFFFFFFu) {
    s_slot[tid] = slot;
    atomicMax(&s_lim[0], my_set);
    atomicMax(&s_lim[1], my_rank + 1u);
    const uint32_t sl = my_set - first;
    bool ok = staged_set && sl < (uint32_t)kBlock;
    if (ok) {
      const uint32_t widx = s_off[sl] - s_off[0] + my_rank;
      ok = widx < kWin;
      if (ok) s_who[widx] = (uint16_t)tid;
    }
    if (!ok) atomicOr(&s_lim[2], 1u);
  }
  __syncthreads();
  const uint32_t ns = s_lim[0] - first + 1u, maxrank = s_lim[1];
  const uint64_t total = (uint64_t)maxrank * Q * ns;
  uint4* __restrict__ hits4 = reinterpret_cast<uint4*>(w.hits);
#ifdef RPA_STUB_WRITEOUT
  if (false) {} else if (false) {
#else
  if (tiled && s_lim[2] == 0u && ns <= (uint32_t)kBlock && total <= 24ull * kBlock) {
#endif
    // idx / ns by one multiply: exact for idx < 2^32 / ns (here idx < 6144, ns <= 256); a 32-bit division is ~35 instructions
    const uint32_t ns_magic = ns == 1u ? 0u : (uint32_t)(0x100000000ull / ns) + 1u;
    for (uint32_t idx = tid; idx < (uint32_t)total; idx += kBlock) {
      const uint32_t t2 = ns == 1u ? idx : __umulhi(idx, ns_magic), sl = idx - t2 * ns, q = t2 % (uint32_t)Q, i = t2 / (uint32_t)Q;
      const uint32_t begs = s_off[sl];
      if (i >= s_off[sl + 1] - begs) continue;          // the set has fewer records than that
      const uint32_t widx = begs - s_off[0] + i;
      if (widx >= kWin) continue;
      const uint32_t src = s_who[widx];
      if (src == 0xFFFFu) continue;                      // another workgroup's record, or fewer ACTIVE records than that
      const uint32_t x = s_slot[src];
      hits4[(size_t)(x & ~63u) * Q + 64u * q + (x & 63u)] = s_rows[src * Q + q];
    }
  } else if (slot != 0xFFFFFFFFu) {
    const size_t base = tiled ? (size_t)(slot & ~63u) * Q + (slot & 63u) : (size_t)slot * Q;
    const size_t sq = tiled ? 64 : 1;
#pragma unroll
    for (int k = 0; k < Q; ++k) hits4[base + (size_t)k * sq] = s_rows[tid * Q + k];
  }
}

// Pair emission bookkeeping: a thread first counts what it will emit per bin, reserves list space with one atomic
// per wave and bin (wave-level prefix sum), then writes.  All lanes of a wave must call this together.
__device__ __forceinline__ uint32_t wave_reserve(uint32_t* counter, uint32_t count) {
  uint32_t inc = count;
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t o = __shfl_up(inc, d, 64);
    if (lane >= d) inc += o;
  }
  const uint32_t total = __shfl(inc, 63, 64);
  uint32_t base = 0;
  if (total) {
    if (lane == 63) base = atomicAdd(counter, total);
    base = __shfl(base, 63, 64);
  }
  return base + inc - count;
}

__device__ __forceinline__ void wave_max_to(uint32_t* counter, uint32_t v) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v = max(v, (uint32_t)__shfl_xor(v, d, 64));
  if ((threadIdx.x & 63) == 0 && v) atomicMax(counter, v);
}

__device__ __forceinline__ int pair_bin(const Params& p, uint32_t la, uint32_t lb) {
  if (p.alphabet == 0) return dna_bin(la < lb ? la : lb);
  (void)la;
  if (p.force_bin >= 0) return p.force_bin;
  return protein_bin(la, lb, p.protein_lanes != 0, p.protein_wide != 0);  // protein: B (second segment) is the vertical sequence cut into register strips
}

// ------------------------------------------------------------------------------------------------------------
// Emit: one thread per hit row (+ one per record set for the anchor<->query alignment of pass 2).  Each thread
// evaluates, for its own hit, the predicate "the loop the set is about to run can reach the alignment of this hit"
// from the set's EmitInfo (written by plan/decide) and its row, and the active ones are stream-compacted per length
// bin into the work lists of the alignment kernels: ballot + one atomic per wave and bin, consecutive lanes write
// consecutive list entries.  Rows and lists are contiguous, so all of this is coalesced -- emission used to be
// the slow half of the per-set kernel when each set's thread wrote its own pair records.
// A list entry is a row position (< H) or H + s; the alignment kernels derive the two segments from it.
//   pass 0 (:511-534): hit not perfect and score >= reeval threshold           -> (segment_i, query range)
//   pass 1 (:604-638): i != anchor and querydistance[i] != 0                  -> (segment_i, segment_anchor)
//   pass 2 (:757-778): i != anchor, score >= cut-off, qpid >= threshold and unode_global not cnode-or-above
//                      -> (segment_i, segment_anchor);  anchor never aligned to the query (:792) -> (segment_anchor, query range)
// ------------------------------------------------------------------------------------------------------------
constexpr int kEmitBlock = 1024;  // one list reservation (atomic) per bin per workgroup: same-address atomics are ~11 ns each
constexpr int kEmitRows = 4;      // row positions per thread: the three dependent loads (set -> EmitInfo -> row) of four
                                  // rows are in flight together; the kernel is bound by that latency chain, not by bytes
constexpr int kEmitSpan = kEmitBlock * kEmitRows;

// "can the loop the set is about to run reach the alignment of the hit at row `pos`?"  Passes 0 and 1 decide from the
// row's 4-byte lite word alone; pass 2 (few sets, late rounds) reads the row.  Returns the bin (or -1), fills entry / maxlen.
// what the emit kernels use of a mode-3 / mode-5 request, from its two quads (emit_desc, emit_ext_quad)
__device__ __forceinline__ EmitInfo emit_info_from(const uint4 desc, const uint4 ext) {
  EmitInfo e{};
  e.mode = desc.x & 7u; e.anchor = desc.x >> 3; e.i_lo = desc.y; e.i_hi = desc.z; e.n = desc.z; e.anchor_len = desc.w;
  e.fcut = __uint_as_float(ext.x); e.qs_min = __uint_as_float(ext.y);
  e.best = ext.z & 0x00FFFFFFu; e.d_unode_g = ext.z >> 24;
  e.qseglen = ext.w & 0x7FFFFFFFu; e.emit_query = ext.w >> 31;
  return e;
}

template <int Q>
__device__ __forceinline__ int emit_predicate(const BatchDev& b, const WorkDev& w, const Params& p, const EmitInfo& info, uint32_t s,
                                              uint32_t lo, uint32_t pos, const uint32_t lite, uint2& entry, uint32_t& maxlen, uint32_t* cb_out = nullptr) {
  const uint32_t H = b.n_records;
  const uint32_t seglen = lite & kLiteLenMask;
  bool want = false;
  const bool vs_query = info.mode == 1 || (info.mode == 4 && pos - lo == info.anchor);
  if (info.mode == 1) want = (lite & kLiteWant0) != 0u;
  else if (info.mode == 4) want = vs_query ? (lite & kLiteWant0) != 0u : !(lite & kLiteZeroQd);
  else if (info.mode == 2) want = !(lite & kLiteZeroQd);
  else if (info.mode == 5) {  // the paused loop's own tests (:604 score cut-off, :612 qpid threshold) with the values at the pause: they only rise
    if (!(lite & kLiteZeroQd)) {
      const HitCore h = set_rows<Q>(w, s, lo).core(pos - lo);
      want = h.score >= info.fcut && h.qs >= info.qs_min;
    }
  } else {
    const SetRows<Q> R = set_rows<Q>(w, s, lo);
    const uint32_t i = pos - lo;
    const HitCore h = R.core(i);
    if (h.score >= info.fcut && h.qs >= info.qs_min) {  // qs_min: never stricter than "(double)qs / L >= qpid_thresh" (qs_threshold)
      uint32_t pb[4 * Q - 6], pth[4 * Q - 6];
      R.path(info.best, pb);
      R.path(i, pth);
      const uint32_t cb = common<4 * Q - 6>(pb, meta_depth(R.core(info.best).meta), pth, meta_depth(h.meta), w.lb_shift);
      want = !(cb >= info.d_unode_g);
      if (cb_out) *cb_out = cb;
    }
  }
  if (!want) return -1;
  const uint32_t other_len = vs_query ? info.qseglen : info.anchor_len;
  const uint32_t other_seg = vs_query ? H + s : lo + info.anchor;
  int bin = pair_bin(p, seglen, other_len);
  maxlen = p.alphabet == 0 ? max(seglen, other_len) : seglen;  // DNA: text = the longer one; protein: A
  entry = make_uint2(pos, other_seg);
  if (bin >= kNumBins) { raise_error(w, kErrUnsupported, s); bin = -1; }
  return bin;
}

#ifndef RPA_EMIT_WAVES
#define RPA_EMIT_WAVES 8  // (A/B: tools/build_variant.sh)
#endif
// (rows of four and more quads -- deep taxonomies, 16- / 32-bit labels: the two paths a pass-2 candidate compares do not fit 64
// registers beside the rest, the kernel spilled 50 / 64 / 232 of them; those shapes get the registers of ONE resident block per CU)
template <int Q>
__global__ __launch_bounds__(kEmitBlock, (Q <= 3 ? RPA_EMIT_WAVES : 4)) void k_emit(BatchDev b, WorkDev w, Params p) {  // two resident blocks per CU: one computes while the other waits on its loads
  constexpr int kVW = kEmitSpan / 64;  // "virtual waves": runs of 64 consecutive row positions
  static_assert(kVW == 64, "the prefix over virtual waves is done by one wavefront");
  __shared__ uint32_t s_cnt[kNumBins][kVW];  // per bin, per virtual wave: number of active lanes, then its list offset
  __shared__ uint32_t s_maxlen[kVW];
  // pass-2 requests (mode 3): depth of LCA(hit, best hit) per row position of the block, 0xFF = not a pass-2 request -- what decides
  // whether a hit waits for the round's second launch (kDeferredMark, rpa_device.h)
  extern __shared__ uint8_t s_cb[];  // kEmitSpan bytes when the round defers (p.defer2), none otherwise: static LDS of this kernel is what it
                                     // takes from the other jobs' kernels on the same CU (config 4 in flight: +4 % per step with 4 KB more)
  __shared__ uint32_t s_deferred;
  if (threadIdx.x == 0) s_deferred = 0u;
  const uint32_t H = b.n_records, total = H + b.n_sets;
  const uint32_t pos0 = blockIdx.x * kEmitSpan + threadIdx.x;
  for (int i = threadIdx.x; i < kNumBins * kVW; i += kEmitBlock) (&s_cnt[0][0])[i] = 0u;  // only bins that occur get a count below
  __syncthreads();
  uint32_t set_of[kEmitRows], lo[kEmitRows], lite[kEmitRows];
  uint4 desc[kEmitRows], ext[kEmitRows];
#pragma unroll
  for (int k = 0; k < kEmitRows; ++k) {  // level 1: row position -> record set; the row's lite word rides along
    const uint32_t pos = pos0 + k * kEmitBlock;
    set_of[k] = pos < H ? w.hit_set[pos] : (pos < total ? pos - H : 0xFFFFFFFFu);
    lite[k] = pos < H ? w.lite[pos] : 0u;
  }
#pragma unroll
  for (int k = 0; k < kEmitRows; ++k) {  // level 2: what the set wants (16-byte descriptor)
    desc[k] = make_uint4(0u, 0u, 0u, 0u); ext[k] = desc[k]; lo[k] = 0;
    if (set_of[k] != 0xFFFFFFFFu) { desc[k] = w.emit_lite[set_of[k]]; ext[k] = w.emit_ext[set_of[k]]; lo[k] = b.set_offsets[set_of[k]]; }
  }
  int bin[kEmitRows];
  uint2 entry[kEmitRows];
  uint32_t maxlen = 0;
  uint32_t pass2_rows = 0;  // bit k: row k of this thread is a pass-2 request
#pragma unroll
  for (int k = 0; k < kEmitRows; ++k) {
    const uint32_t pos = pos0 + k * kEmitBlock;
    const uint32_t mode = desc[k].x & 7u, anchor = desc[k].x >> 3;
    const uint32_t i = pos - lo[k];
    bin[k] = -1;
    entry[k] = make_uint2(0u, 0u);
    uint32_t ml = 0;
    uint32_t cb = 0xFFu;
    if (pos < H) {
      if (mode != 0u && (mode == 4u || i >= desc[k].y) && i < desc[k].z && (mode == 1u || mode == 4u || i != anchor)) {
        if (mode != 3u && mode != 5u) {  // passes 0 and 1: the lite word decides
          const bool vs_query = mode == 1u || (mode == 4u && i == anchor);
          const bool want = vs_query ? (lite[k] & kLiteWant0) != 0u : !(lite[k] & kLiteZeroQd);
          if (want) {
            const uint32_t seglen = lite[k] & kLiteLenMask, other_len = (mode == 4u && vs_query) ? desc[k].y : desc[k].w;
            bin[k] = pair_bin(p, seglen, other_len);
            ml = p.alphabet == 0 ? max(seglen, other_len) : seglen;  // DNA: text = the longer one; protein: A
            entry[k] = make_uint2(pos, vs_query ? H + set_of[k] : lo[k] + anchor);
            if (bin[k] >= kNumBins) { raise_error(w, kErrUnsupported, set_of[k]); bin[k] = -1; }
          }
        } else {
          const EmitInfo info = emit_info_from(desc[k], ext[k]);  // (modes 3 and 5: the two quads say it all -- no trip to the 64-byte record)
          uint32_t cbv = 0xFFu;
          bin[k] = emit_predicate<Q>(b, w, p, info, set_of[k], lo[k], pos, lite[k], entry[k], ml, &cbv);
          if (mode == 3u && bin[k] >= 0 && p.defer2) { cb = cbv < 0xFFu ? cbv : 0xFEu; pass2_rows |= 1u << k; }
        }
      }
    } else if (pos < total && mode == 3u) {  // anchor <-> query range (:792-797)
      const EmitInfo info = emit_info_from(desc[k], ext[k]);
      if (info.emit_query) {
        bin[k] = pair_bin(p, info.anchor_len, info.qseglen);
        ml = p.alphabet == 0 ? max(info.anchor_len, info.qseglen) : info.anchor_len;
        entry[k] = make_uint2(lo[k] + info.anchor, pos | kDstIsB);
        if (bin[k] >= kNumBins) { raise_error(w, kErrUnsupported, set_of[k]); bin[k] = -1; }
      }
    }
    maxlen = max(maxlen, ml);
    if (p.defer2) s_cb[k * kEmitBlock + threadIdx.x] = (uint8_t)cb;  // (uniform: a launch parameter)
  }
  // Which pass-2 requests wait for the round's second launch: those that an EARLIER request of the same chunk could take out of
  // the loop by lifting unode_global -- an earlier row of my set, inside the chunk, whose LCA with the best hit is no deeper than
  // mine.  Rows of a set are consecutive row positions = consecutive threads of one of the block's four runs of 1024; what lies
  // in front of a run is not seen (the hit then simply goes with the first launch, which is always allowed).
  if (p.defer2) __syncthreads();
  if (p.defer2 && pass2_rows) {
#pragma unroll
    for (int k = 0; k < kEmitRows; ++k) {
      if (!((pass2_rows >> k) & 1u)) continue;
      const uint32_t pos = pos0 + k * kEmitBlock, i = pos - lo[k], i_lo = desc[k].y;
      const uint32_t back = min(min(i - i_lo, (uint32_t)threadIdx.x), kDeferSpan - 1u);  // rows of my chunk in front of me that this run holds
      if (i - i_lo >= kDeferSpan) continue;  // (the filter keeps a chunk's held-back hits as one 64-bit set)
      const uint32_t mine = s_cb[k * kEmitBlock + threadIdx.x];
      bool covered = false;
      for (uint32_t j = 1; j <= back; ++j) covered = covered || s_cb[k * kEmitBlock + threadIdx.x - j] <= mine;
      if (covered) {
        w.pair_res[res_index(w, pos)] = make_float2(kDeferredMark, 0.f);
        bin[k] = -1;
        atomicAdd(&s_deferred, 1u);
      }
    }
  }
  // block-level compaction: wave ballots -> LDS -> one atomic per non-empty bin per block
  // (per row position of a thread: one ballot per bin PRESENT in the wavefront -- usually one or none -- instead of one
  // per bin: twelve ballots per row were two thirds of the kernel's vector instructions)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned long long my_mask[kEmitRows];
#pragma unroll
  for (int k = 0; k < kEmitRows; ++k) {
    my_mask[k] = 0ull;
    const int vw = k * (kEmitBlock / 64) + wave;
    unsigned long long todo = __ballot(bin[k] >= 0);
    while (todo) {
      const int leader = (int)__ffsll((long long)todo) - 1;
      const int kk = __builtin_amdgcn_readlane(bin[k], leader);
      const unsigned long long mask = __ballot(bin[k] == kk);
      if (lane == leader) s_cnt[kk][vw] = (uint32_t)__popcll(mask);
      if (bin[k] == kk) my_mask[k] = mask;
      todo &= ~mask;
    }
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) maxlen = max(maxlen, (uint32_t)__shfl_xor(maxlen, d, 64));
  if (lane == 0) s_maxlen[wave] = maxlen;
  __syncthreads();
  if (wave == 0) {  // lane l = virtual wave l: exclusive prefix per bin, one reservation per non-empty bin
#pragma unroll
    for (int kk = 0; kk < kNumBins; ++kk) {
      const uint32_t c = s_cnt[kk][lane];
      if (!__any(c != 0u)) continue;
      uint32_t inc = c;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = __shfl_up(inc, d, 64);
        if (lane >= d) inc += o;
      }
      uint32_t base = 0;
      if (lane == 63) base = atomicAdd(&w.counters[kk], inc);
      base = __shfl(base, 63, 64);
      s_cnt[kk][lane] = base + inc - c;
    }
  } else if (wave == 1) {
    uint32_t m = lane < kEmitBlock / 64 ? s_maxlen[lane] : 0u;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) m = max(m, (uint32_t)__shfl_xor(m, d, 64));
    if (lane == 0 && m) atomicMax(&w.counters[kCtrMaxLen], m);
    if (lane == 0 && s_deferred) atomicAdd(&w.counters[kCtrDeferred], s_deferred);
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < kEmitRows; ++k)
    if (bin[k] >= 0)
      w.bin_list[(size_t)bin[k] * w.T2 + s_cnt[bin[k]][k * (kEmitBlock / 64) + wave] + (uint32_t)__popcll(my_mask[k] & ((1ull << lane) - 1ull))] = entry[k];
}

// ------------------------------------------------------------------------------------------------------------
// Between the two launches of a pass-2 round (kDeferredMark, rpa_device.h): one thread per record set whose request is a pass-2
// chunk.  It walks the chunk the way k_decide's pass-2 loop will (:757-813) with the results the first launch left: `d` follows
// unode_global wherever a result that is there lifts it.  d never lies below the value the real loop will have at the same row --
// every lift applied here happens there too, or the hit is stepped over there because unode_global is already at or above it
// -- so a held-back hit with LCA depth >= d is one the real loop steps over: it stays unaligned.  Every other held-back hit goes
// onto the bin lists of the second launch (counted in counters[kCtrSurvivor + bin]).  Nothing else changes: k_decide consumes
// exactly what it always did, and a mark it should ever consume is a failed assert, not a distance.
// ------------------------------------------------------------------------------------------------------------
constexpr int kFilterBlock = 1024;  // one list reservation per bin per workgroup
template <int Q>
__global__ __launch_bounds__(kFilterBlock) void k_defer_filter(BatchDev b, WorkDev w, Params p) {
  constexpr int PW = 4 * Q - 6;
  __shared__ uint32_t s_count[kNumBins], s_base[kNumBins];
  if (threadIdx.x < (uint32_t)kNumBins) s_count[threadIdx.x] = 0u;
  if (blockIdx.x == 0 && threadIdx.x == 0) { w.counters[kCtrRetry] = 0u; w.counters[kCtrRetryShort] = 0u; w.counters[kCtrWindowFail] = 0u; }  // the retry lists (and the ring window's list) start over for the second launch
  __syncthreads();
  const uint32_t s = blockIdx.x * kFilterBlock + threadIdx.x;
  unsigned long long survivors = 0ull;  // bit k: hit i_lo + k of my chunk goes into the second launch
  uint32_t lo = 0, i_lo = 0, anchor = 0, anchor_len = 0;
  if (s < b.n_sets) {
    const uint4 desc = w.emit_lite[s];
    if ((desc.x & 7u) == 3u) {
      const EmitInfo info = emit_info_from(desc, w.emit_ext[s]);
      const double qpid_thresh = w.emit[s].qpid_thresh;
      const uint4 sinfo = w.set_info[s];
      const uint32_t L = sinfo.z - sinfo.y + 1u;
      const float bf_max = __uint_as_float(w.state[s].q[1].w);
      lo = b.set_offsets[s];
      i_lo = info.i_lo; anchor = info.anchor; anchor_len = info.anchor_len;
      const SetRows<Q> R = set_rows<Q>(w, s, lo);
      uint32_t pb[PW];
      R.path(info.best, pb);
      const uint32_t db = meta_depth(R.meta(info.best));
      float qd_a = R.core(anchor).qd;
      uint32_t d = info.d_unode_g;
      const uint32_t i_hi = min(info.i_hi, sinfo.x);
      for (uint32_t i = i_lo; i < i_hi; ++i) {
        const HitCore h = R.core(i);
        if (!(h.score >= info.fcut)) break;   // :757
        if (i == anchor) continue;           // distance 0: lifts nothing, is never held back
        if (!((double)h.qs / L >= qpid_thresh)) continue;  // :765 (the real test; k_emit's float threshold is a superset)
        uint32_t pth[PW];
        R.path(i, pth);
        const uint32_t cb = common<PW>(pb, db, pth, meta_depth(h.meta), w.lb_shift);
        if (cb >= d) continue;               // :771 with d >= the real unode_global: stepped over there too
        const float2 r2 = R.result(i);
        if (r2.x == kDeferredMark) {         // held back and still reachable
          if (i - i_lo < kDeferSpan) survivors |= 1ull << (i - i_lo);
          continue;
        }
        if (r2.x != 0.f) {                   // :789-813
          if (qd_a == FLT_MAX) qd_a = w.pair_res[w.P + s].x;  // the anchor <-> query alignment (:792-797): asked for with the anchor's first chunk
          if (r2.x <= qd_a * bf_max) d = min(d, cb);
        }
      }
    }
  }
  // survivors -> bin lists: count per bin in LDS, one reservation per bin and workgroup, then write
  uint32_t my_bins = 0;  // (a set's held-back hits nearly always share one bin; the general case walks the bits twice)
  for (unsigned long long m = survivors; m; m &= m - 1ull) {
    const uint32_t i = i_lo + (uint32_t)__ffsll((long long)m) - 1u;
    int bin = pair_bin(p, w.lite[lo + i] & kLiteLenMask, anchor_len);
    if (bin >= kNumBins) { raise_error(w, kErrUnsupported, s); bin = 0; }
    atomicAdd(&s_count[bin], 1u);
    ++my_bins;
  }
  __syncthreads();
  if (threadIdx.x < (uint32_t)kNumBins) {
    const uint32_t c = s_count[threadIdx.x];
    s_base[threadIdx.x] = c ? atomicAdd(&w.counters[kCtrSurvivor + threadIdx.x], c) : 0u;
    s_count[threadIdx.x] = 0u;
  }
  __syncthreads();
  for (unsigned long long m = survivors; m; m &= m - 1ull) {
    const uint32_t i = i_lo + (uint32_t)__ffsll((long long)m) - 1u;
    int bin = pair_bin(p, w.lite[lo + i] & kLiteLenMask, anchor_len);
    if (bin >= kNumBins) bin = 0;
    const uint32_t at = s_base[bin] + atomicAdd(&s_count[bin], 1u);
    w.bin_list[(size_t)bin * w.T2 + at] = make_uint2(lo + i, lo + anchor);
  }
  (void)my_bins;
}

// In a sparse round every launch is latency-bound (a few thousand lanes at most), so all patterns of up to 256 nt go
// into ONE list, run by the widest one-strip kernel (exact for any pattern that fits its strip), instead of one
// serialised ~60 us launch per pattern-length class; those of 257 ... 384 nt (beyond K1w's eight words) into a second one.
__device__ __forceinline__ int sparse_bin(const Params& p, int bin) {
  if (p.alphabet != 0 || bin < 0 || bin >= kShortWords || p.force_bin >= 0) return bin;
  return bin < kWaveMaxWords ? kWaveMaxWords - 1 : kShortWords - 1;
}

// Sparse rounds: when only a small fraction of the record sets is still running, scanning every row of the batch
// would cost more than the work left.  One wavefront per live set; lane j looks at hits j, j+64, ...; lane 0 also
// emits the anchor <-> query alignment.  Same predicate and same list format as k_emit.
constexpr int kSparseBlock = 1024;  // 16 live sets per workgroup: one list reservation per bin per workgroup
template <int Q>
__global__ __launch_bounds__(kSparseBlock) void k_emit_sparse(BatchDev b, WorkDev w, Params p) {
  // The sparse rounds put nearly everything into ONE list (sparse_bin), and a reservation per wavefront means thousands of
  // atomics on one counter at ~11 ns each (6 k live sets: 92 of the kernel's 140 us).  The first 64 hits of every set
  // (all of them, for sets of up to 63 hits: the anchor <-> query pair rides in the lane behind the last hit) are
  // therefore reserved per WORKGROUP: ballots -> LDS -> one atomic per bin; longer sets finish with per-wave reservations.
  __shared__ uint32_t s_cnt[kNumBins][kSparseBlock / 64];
  const uint32_t wave_id = (blockIdx.x * kSparseBlock + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const bool have = wave_id < w.counters[w.ctr_live_in];  // exact list length (the grid is sized by an upper bound)
  const uint32_t H = b.n_records;
  uint32_t s = 0, lo = 0, i_hi = 0, i_lo = 0;
  EmitInfo info{};
  info.mode = 0;
  if (have) {
    s = w.live_in[wave_id];
    const uint4 desc = w.emit_lite[s];
    info.mode = desc.x & 7u;
    if (info.mode == 3u || info.mode == 5u) info = emit_info_from(desc, w.emit_ext[s]);  // pass 2: thresholds, best hit, depth of unode_global; 5: thresholds
    else {                                  // passes 0 and 1: the descriptor says it all (emit_desc)
      info.anchor = desc.x >> 3; info.n = desc.z; info.i_hi = desc.z;
      info.i_lo = info.mode == 4u ? 0u : desc.y;
      info.qseglen = info.mode == 4u ? desc.y : desc.w;
      info.anchor_len = desc.w;
    }
    lo = b.set_offsets[s];
    i_hi = min(info.i_hi, info.n);
    i_lo = info.i_lo;
  }
  const bool work = have && info.mode != 0;
  uint32_t maxlen = 0;
  // ---- first chunk: hits i_lo .. i_lo + 63, workgroup-level reservation ----
  int bin = -1;
  uint2 entry = make_uint2(0u, 0u);
  bool extra_done = false;
  if (work) {
    const uint32_t i = i_lo + (uint32_t)lane;
    if (i < i_hi && (info.mode == 1 || info.mode == 4 || i != info.anchor)) {
      uint32_t ml = 0;
      bin = sparse_bin(p, emit_predicate<Q>(b, w, p, info, s, lo, lo + i, w.lite[lo + i], entry, ml));
      maxlen = max(maxlen, ml);
    }
    const uint32_t span = i_hi > i_lo ? i_hi - i_lo : 0u;
    if (info.mode == 3 && info.emit_query && span < 64u) {  // anchor <-> query range (:792-797) in the first free lane
      extra_done = true;
      if ((uint32_t)lane == span) {
        int xb = pair_bin(p, info.anchor_len, info.qseglen);
        if (xb >= kNumBins) raise_error(w, kErrUnsupported, s);
        else {
          bin = sparse_bin(p, xb);
          maxlen = max(maxlen, p.alphabet == 0 ? max(info.anchor_len, info.qseglen) : info.anchor_len);
          entry = make_uint2(lo + info.anchor, (H + s) | kDstIsB);
        }
      }
    }
  }
  unsigned long long my_mask = 0ull;
#pragma unroll
  for (int k = 0; k < kNumBins; ++k) {
    const unsigned long long mask = __ballot(bin == k);
    if (lane == 0) s_cnt[k][wave] = (uint32_t)__popcll(mask);
    if (bin == k) my_mask = mask;
  }
  __syncthreads();
  if (wave == 0) {  // lane l < 16 = wave l of the workgroup: exclusive prefix per bin, one reservation per non-empty bin
#pragma unroll
    for (int k = 0; k < kNumBins; ++k) {
      const uint32_t c = lane < kSparseBlock / 64 ? s_cnt[k][lane] : 0u;
      if (!__any(c != 0u)) continue;
      uint32_t inc = c;
#pragma unroll
      for (int d = 1; d < kSparseBlock / 64; d <<= 1) {
        const uint32_t o = __shfl_up(inc, d, 64);
        if (lane >= d) inc += o;
      }
      uint32_t base = 0;
      if (lane == kSparseBlock / 64 - 1) base = atomicAdd(&w.counters[k], inc);
      base = __shfl(base, kSparseBlock / 64 - 1, 64);
      if (lane < kSparseBlock / 64) s_cnt[k][lane] = base + inc - c;
    }
  }
  __syncthreads();
  if (bin >= 0) w.bin_list[(size_t)bin * w.T2 + s_cnt[bin][wave] + (uint32_t)__popcll(my_mask & ((1ull << lane) - 1ull))] = entry;
  // ---- sets with more than 64 hits in their range: the rest, one reservation per wavefront and bin ----
  if (work) {
    for (uint32_t i0 = i_lo + 64u; i0 < i_hi; i0 += 64) {
      const uint32_t i = i0 + (uint32_t)lane;
      int bn = -1;
      uint2 en = make_uint2(0u, 0u);
      if (i < i_hi && (info.mode == 1 || info.mode == 4 || i != info.anchor)) {
        uint32_t ml = 0;
        bn = sparse_bin(p, emit_predicate<Q>(b, w, p, info, s, lo, lo + i, w.lite[lo + i], en, ml));
        maxlen = max(maxlen, ml);
      }
#pragma unroll
      for (int k = 0; k < kNumBins; ++k) {
        const unsigned long long mask = __ballot(bn == k);
        if (mask == 0ull) continue;
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(&w.counters[k], (uint32_t)__popcll(mask));
        base = __shfl(base, 0, 64);
        if (bn == k) w.bin_list[(size_t)k * w.T2 + base + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull))] = en;
      }
    }
    if (info.mode == 3 && info.emit_query && !extra_done && lane == 0) {  // anchor <-> query range (:792-797)
      int xb = pair_bin(p, info.anchor_len, info.qseglen);
      if (xb >= kNumBins) raise_error(w, kErrUnsupported, s);
      else {
        xb = sparse_bin(p, xb);
        maxlen = max(maxlen, p.alphabet == 0 ? max(info.anchor_len, info.qseglen) : info.anchor_len);
        w.bin_list[(size_t)xb * w.T2 + atomicAdd(&w.counters[xb], 1u)] = make_uint2(lo + info.anchor, (H + s) | kDstIsB);
      }
    }
  }
  // the longest sequence of any emitted pair: ONE atomic per workgroup (one per wavefront was 6 k atomics on one address in the
  // first sparse round of a config-2 step, ~11 ns each: 66 of the kernel's 80 us)
  __shared__ uint32_t s_maxlen[kSparseBlock / 64];
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) maxlen = max(maxlen, (uint32_t)__shfl_xor(maxlen, d, 64));
  if (lane == 0) s_maxlen[wave] = maxlen;
  __syncthreads();
  if (wave == 0) {
    uint32_t m = lane < kSparseBlock / 64 ? s_maxlen[lane] : 0u;
#pragma unroll
    for (int d = 8; d >= 1; d >>= 1) m = max(m, (uint32_t)__shfl_xor(m, d, 64));
    if (lane == 0 && m) atomicMax(&w.counters[kCtrMaxLen], m);
  }
}

// ---- a set's state between rounds --------------------------------------------------------------------------------
// SetStateMem (64 bytes): phase | loop_live << 8 | combined << 9, support, best, anchor | the four global depths, one
// byte each | lrd_min | ival_g, bf_max | the three alignment counters | i_next, chunk_hi.  The fields of SetState that
// describe the set itself (n, query range, its segment) come back from set_info; the loop variables of a paused pass-1
// loop from LoopSave.  A round reads 80 bytes and writes 64 per live set (the unpacked struct: 144 + 144).
__device__ __forceinline__ void state_store(const WorkDev& w, uint32_t s, const SetState& st, bool save_loop) {
  SetStateMem m;
  m.q[0] = make_uint4(st.phase | (st.loop_live << 8) | (st.combined << 9), st.support, st.best, st.anchor);
  m.q[1] = make_uint4((st.d_lca_all & 255u) | ((st.d_rtax & 255u) << 8) | ((st.d_lnode_g & 255u) << 16) | (st.d_unode_g << 24), st.lrd_min,
                      __float_as_uint(st.ival_g), __float_as_uint(st.bf_max));
  m.q[2] = make_uint4(st.cnt0, st.cnt1, st.cnt2, st.i_next);
  m.q[3] = make_uint4(st.chunk_hi, st.gq, st.go, 0u);
  uint4* dst = w.state[s].q;
  dst[0] = m.q[0]; dst[1] = m.q[1]; dst[2] = m.q[2]; dst[3] = m.q[3];
  if (save_loop) {
    LoopSave ls;
    ls.nog = st.nog; ls.pad_ = st.pad_; ls.d_lnode = st.d_lnode; ls.scut = st.scut; ls.ldist = st.ldist; ls.udist = st.udist;
    ls.qpid_upper = st.qpid_upper; ls.cut_g = st.cut_g; ls.cut_h = st.cut_h;
    w.loop_save[s] = ls;
  }
}
__device__ __forceinline__ void state_fill_static(SetState& st, const uint4 info) {  // from the set's set_info quad
  st.n = info.x; st.qrstart = info.y; st.qrstop = info.z; st.L = info.z - info.y + 1u;
  st.qseglen = info.w & kLiteLenMask; st.qsegerr = info.w >> 29;
}
__device__ __forceinline__ void state_load(const WorkDev& w, uint32_t s, SetState& st, const uint4* pre_info = nullptr) {
  const uint4* src = w.state[s].q;
  const uint4 a = src[0], b = src[1], c = src[2], d = src[3];
  st.phase = a.x & 255u; st.loop_live = (a.x >> 8) & 1u; st.combined = (a.x >> 9) & 1u;
  st.support = a.y; st.best = a.z; st.anchor = a.w;
  st.d_lca_all = b.x & 255u; st.d_rtax = (b.x >> 8) & 255u; st.d_lnode_g = (b.x >> 16) & 255u; st.d_unode_g = b.x >> 24;
  st.lrd_min = b.y; st.ival_g = __uint_as_float(b.z); st.bf_max = __uint_as_float(b.w);
  st.cnt0 = c.x; st.cnt1 = c.y; st.cnt2 = c.z; st.i_next = c.w; st.chunk_hi = d.x; st.gq = d.y; st.go = d.z;
  st.nog = 0; st.pad_ = 0; st.d_lnode = 0; st.scut = 0; st.ldist = 0.f; st.udist = 0.f; st.qpid_upper = st.cut_g = st.cut_h = 0.;
  if (st.phase != kDone) {
    state_fill_static(st, pre_info ? *pre_info : w.set_info[s]);
    if (st.loop_live && st.phase == kPass1Wait) {
      const LoopSave ls = w.loop_save[s];
      st.nog = ls.nog; st.pad_ = ls.pad_; st.d_lnode = ls.d_lnode; st.scut = ls.scut; st.ldist = ls.ldist; st.udist = ls.udist;
      st.qpid_upper = ls.qpid_upper; st.cut_g = ls.cut_g; st.cut_h = ls.cut_h;
    }
  }
}

// Append the record sets that are still running to the next round's live list: one reservation per workgroup
// (same-address atomics serialise at ~11 ns each; a dense round appends up to n_sets entries).  All threads call.
__device__ __forceinline__ void append_live(const WorkDev& w, bool running, uint32_t s) {
  __shared__ uint32_t s_live[kBlock / 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const unsigned long long m = __ballot(running);
  if (lane == 0) s_live[wave] = (uint32_t)__popcll(m);
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t total = 0;
#pragma unroll
    for (int k = 0; k < kBlock / 64; ++k) { const uint32_t c = s_live[k]; s_live[k] = total; total += c; }
    const uint32_t base = total ? atomicAdd(&w.counters[w.ctr_live_out], total) : 0u;
#pragma unroll
    for (int k = 0; k < kBlock / 64; ++k) s_live[k] += base;
  }
  __syncthreads();
  if (running) w.live_out[s_live[wave] + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = s;
}

// The same without making a wavefront wait for its workgroup (k_decide: a wavefront lives for dozens of microseconds and
// the four of a workgroup finish far apart; with the barriers above each of them also sat out the list reservation's trip
// to memory -- 15 % of the kernel's wave time, profiles/r04_decide_phases.txt).  Every wavefront leaves its ballot and its
// sets in LDS and counts itself in; the one that arrives LAST reserves for all four and writes all four's entries (same
// order inside the workgroup as above).  `arrived` must be zero when the first wavefront gets here (the kernel clears it
// behind a barrier at its very start, where the wavefronts still run together).
struct LiveStage {
  uint32_t arrived;
  uint32_t pad_;
  unsigned long long mask[kBlock / 64];
  uint32_t set[kBlock / 64][64];
};
__device__ __forceinline__ void append_live_last_wave(const WorkDev& w, LiveStage& ls, bool running, uint32_t s) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const unsigned long long m = __ballot(running);
  ls.set[wave][lane] = s;
  if (lane == 0) ls.mask[wave] = m;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  uint32_t before = 0;
  if (lane == 0) before = atomicAdd(&ls.arrived, 1u);
  before = (uint32_t)__builtin_amdgcn_readfirstlane((int)before);
  if (before != (uint32_t)(kBlock / 64 - 1)) return;  // (whole wavefront: `before` is uniform)
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  uint32_t total = 0;
#pragma unroll
  for (int k = 0; k < kBlock / 64; ++k) total += (uint32_t)__popcll(ls.mask[k]);
  if (total == 0) return;
  uint32_t base = 0;
  if (lane == 0) base = atomicAdd(&w.counters[w.ctr_live_out], total);
  base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
#pragma unroll
  for (int k = 0; k < kBlock / 64; ++k) {
    const unsigned long long mk = ls.mask[k];
    if ((mk >> lane) & 1ull) w.live_out[base + (uint32_t)__popcll(mk & ((1ull << lane) - 1ull))] = ls.set[k][lane];
    base += (uint32_t)__popcll(mk);
  }
}

// Where a pass-1 chunk ends.  The loop of an anchor stops at the first hit whose score is below its cut-off (:604), and
// the cut-off is X times the score of the first hit the loop finds farther from the anchor than the query is (:659).  A
// chunk is made on the assumption that its FIRST hit -- the one the loop paused at, or the first one a new loop aligns --
// will be that hit (if the loop has no cut-off yet): hits are in score order, so the chunk ends where the scores fall
// below X times the first hit's (binary search over the score words).  When the assumption fails the loop pauses at the
// chunk end like at any other; when the cut-off is already set the end is exact.  tools/spec_waste.py: config 4 asks for
// 1.08 alignments per demanded one instead of 1.20, with the whole loop in one request (1.5 request rounds per set after
// pass 0 instead of 1.95); RPA_PASS1_FILTER=1 switches the assumption off.
// Measured (tools/spec_sweep9/11.sh): protein (config 4, scores spread over more than a factor 1/X) 20 % faster per step;
// nucleotide (configs 2 and 5: the scores of a set lie within that factor, the chunk end is almost always the set's end)
// 1 % slower -- the probes, and in config 5 one round more.  filter1 = 2: protein only, 3: always.
__device__ __forceinline__ bool assume_first_hit_cutoff(const Params& p) { return p.filter1 >= 3 || (p.filter1 == 2 && p.alphabet != 0); }
// nucleotide: an assumed cut-off is acted on only where it trims at least a quarter of the chunk (filter1 = 4: always, as protein does)
__device__ __forceinline__ uint32_t assumed_cut_den(const Params& p) { return (p.alphabet == 0 && p.filter1 == 3) ? 4u : 0u; }
template <int Q>
__device__ __forceinline__ uint32_t chunk_end_by_score(const SetRows<Q>& R, uint32_t first, uint32_t hi, int scut_now, float X, uint32_t p_min_cut_den = 0u) {
  if (first + 1u >= hi) return hi;
  const int scut = scut_now ? scut_now : float_to_int_x86(R.score(first) * X);
  const float fcut = (float)scut;
  if (R.score(hi - 1u) >= fcut) return hi;  // (scores within a factor X of each other: the usual DNA case, one probe)
  uint32_t a = first + 1u, b = hi - 1u;  // smallest r in (first, hi) with !(score_r >= fcut)
  while (a < b) {
    const uint32_t m = (a + b) >> 1;
    if (R.score(m) >= fcut) a = m + 1u; else b = m;
  }
  // (an ASSUMED cut-off that would only trim the chunk's tail is not acted on: where it is wrong the loop pays a pause and a round for
  // the few alignments it saved -- config 2, scores within a factor X: +2 % per step with it; it is for the sets whose scores spread)
  if (!scut_now && p_min_cut_den && (hi - a) * p_min_cut_den < (hi - first)) return hi;
  return a;
}

// ------------------------------------------------------------------------------------------------------------
// Plan: one thread per record set.  Handles everything predict() decides before any alignment (:341-472): n == 0,
// n == 1, perfect-best-hit shortcut; otherwise emits pass 0.  (Masking, query range, sorting and segment plans were
// done per record by k_rank_records.)
// ------------------------------------------------------------------------------------------------------------
template <int Q>
__global__ __launch_bounds__(kBlock) void k_plan_sets(BatchDev b, WorkDev w, TaxDev tax, Params p, OutRow* __restrict__ rows,
                                                       OutStats* __restrict__ stats) {
  constexpr int PW = 4 * Q - 6;
  const uint32_t s = blockIdx.x * kBlock + threadIdx.x;
  const bool live = s < b.n_sets;
  SetState st;
  st.phase = kDone;
  EmitInfo em{};
  em.mode = 0;
  uint32_t spread_set = 0;
  if (live) {
    const uint32_t lo = b.set_offsets[s], hi = b.set_offsets[s + 1];
    uint4 info = make_uint4(0u, 0u, 0u, 0u);
    if (hi > lo) info = w.set_info[s];
    const uint32_t n = info.x, qrstart = info.y, qrstop = info.z;
    state_fill_static(st, info);
    st.support = 0; st.best = 0; st.d_lca_all = 0; st.d_rtax = 0; st.d_lnode_g = 0; st.d_unode_g = 0; st.ival_g = 0.f;
    st.bf_max = 1.f; st.lrd_min = 255; st.anchor = 0; st.cnt0 = st.cnt1 = st.cnt2 = 0; st.loop_live = 0; st.chunk_hi = 0; st.i_next = 0;
    st.combined = 0; st.pad_ = 0; st.gq = 0; st.go = 0;
    OutRow row;
    row.begin = 1; row.end = b.query_len[s];  // initPredictionRecord, core/src/taxonpredictionmodel.hh:42-44
    row.lower = row.upper = row.rtax = kRoot; row.support = 0; row.ival = 0.f; row.flags = 0;
    bool done = true;
    if (n == 0) {  // :359-368
      row.flags = kFlagIvalUntouched;
    } else if (n == 1) {  // :371-388
      const HitCore h = set_rows<Q>(w, s, lo).core(0u);
      row.begin = __float_as_uint(h.qd); row.end = __float_as_uint(h.qs); row.ival = 1.f;
      row.lower = h.node; row.upper = kRoot; row.support = h.ident; row.rtax = h.node;
      row.flags = kFlagSingle;
    } else {
      const SetRows<Q> R = set_rows<Q>(w, s, lo);
      const HitCore h0 = R.core(0u);
      // the reference fetches the query range (:415) BEFORE it looks at the best hit: an unreadable range aborts the run
      // even where the shortcut below would not have needed the sequence
      if (st.qsegerr) raise_error(w, st.qsegerr, s);
      if (h0.meta & kMetaPerfect) {  // :431-472: lnode / unode are points on hit 0's path
        uint32_t path0[PW];
        R.path(0u, path0);
        const uint32_t d0 = meta_depth(h0.meta);
        const float best = h0.score;
        uint32_t d_l = d0, d_u = 0;
        uint32_t i = 1;
        for (;;) {
          if (i == n) { d_u = 0; break; }  // unode = root
          HitCore hi2 = R.core(i);
          uint32_t pth[PW];
          R.path(i, pth);
          if (hi2.score == best) d_l = min(d_l, common<PW>(path0, d0, pth, meta_depth(hi2.meta), w.lb_shift));
          else {
            const float sc = hi2.score;
            d_u = d_l;
            for (;;) {
              d_u = min(d_u, common<PW>(path0, d0, pth, meta_depth(hi2.meta), w.lb_shift));
              if (++i >= n) break;
              hi2 = R.core(i);
              if (!(hi2.score == sc)) break;
              R.path(i, pth);
            }
            break;
          }
          ++i;
        }
        row.begin = qrstart; row.end = qrstop; row.ival = 0.f;
        row.lower = anc_node(tax, h0.node, d_l); row.upper = anc_node(tax, h0.node, d_u); row.support = st.L; row.rtax = row.lower;
        row.flags = kFlagShortcut;
      } else {
        done = false;
        st.phase = kPass0Wait;
        // (do this batch's scores spread?  Sets in which a cut-off assumed from hit 1 would trim a quarter of the hits or more are
        // counted; the engine lets k_decide's nucleotide loops assume their cut-offs when one set in sixteen is like that -- the
        // probes cost config 2, whose scores lie within a factor X, 2-3 % per step for nothing)
        if (p.alphabet == 0 && p.filter1 == 3 && n >= 3u) spread_set = chunk_end_by_score(set_rows<Q>(w, s, lo), 1u, n, 0, p.exclude_factor, 4u) < n ? 1u : 0u;
        em.mode = 1; em.n = n; em.L = st.L; em.qseglen = st.qseglen; em.i_lo = 0; em.i_hi = n;
        em.fcut = p.reeval_factor * h0.score;  // dbalignment_searchscore_threshold, :499
        // Pass 0 and the first pass-1 loop in ONE round.  qgroup = the perfect hits + the hits above the threshold
        // (:511-518).  If that is a single (not perfect) hit -- it is then hit 0, the top score -- every other hit
        // comes out of pass 0 with distance FLT_MAX, so index_best and the first pass-1 anchor are hit 0 whatever its
        // distance turns out to be (:542-546,553-562,582-583).  The pass-1 pairs (hit i, hit 0) can therefore be asked for
        // together with the pass-0 pair (hit 0, query range): mode 4; k_decide consumes pass 0 and, the anchor being as
        // predicted, the pass-1 loop in the same call.  Typical data (one hit within 5 % of the best score): one round,
        // one decide launch and one small alignment launch less per batch.
        uint32_t members = 0;
        for (uint32_t i = 0; i < n; ++i) members += (w.lite[lo + i] & (kLiteWant0 | kLitePerfect)) ? 1u : 0u;
        if (members == 1 && (w.lite[lo] & kLiteWant0) && !p.no_combined) {
          em.mode = 4; em.anchor = 0; em.anchor_len = w.lite[lo] & kLiteLenMask;
          em.i_hi = p.spec1 ? min(n, p.spec1) : n;  // the first speculation chunk of the pass-1 loop (hit 0 is part of it)
          if (assume_first_hit_cutoff(p)) em.i_hi = chunk_end_by_score(set_rows<Q>(w, s, lo), 1u, em.i_hi, 0, p.exclude_factor, assumed_cut_den(p));  // (no perfect hit here: hit 1 is the first one aligned)
          st.chunk_hi = em.i_hi;                    // remembered: k_decide may run with another chunk size by then
          st.combined = 1;
        }
      }
    }
    if (done) {
      rows[s] = row;
      if (stats) stats[s] = OutStats{n, 0, 0, 0};
    }
    w.emit_lite[s] = emit_desc(em);  // (modes 1 and 4: everything k_emit needs)
    st.nog = 0; st.d_lnode = 0; st.scut = 0; st.ldist = st.udist = 0.f; st.qpid_upper = st.cut_g = st.cut_h = 0.;
    state_store(w, s, st, false);
  }
  append_live(w, live && st.phase != kDone, s);
  if (p.alphabet == 0 && p.filter1 == 3) {  // (uniform)
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) spread_set += (uint32_t)__shfl_xor(spread_set, d, 64);
    if ((threadIdx.x & 63) == 0 && spread_set) atomicAdd(&w.counters[kCtrSpreadSets], spread_set);
  }
}

// ------------------------------------------------------------------------------------------------------------
// BandFactor (:259-323) without the sort.  The reference sorts the (distance, node) entries by distance and walks
// them, remembering per taxonomic rank (= depth of LCA(node, anchor)) the distance at which that rank was first
// reached ("worst[rank]", only while ranks keep decreasing) and, for every entry whose rank is higher than the lowest
// rank seen so far ("disorder"), taking max(distance / worst[r]) over the remembered lower ranks r.  Because
// x -> fl(x / y) is monotone, that maximum only needs, per rank, the smallest (fmin) and the largest (dmax) distance:
//   rank rho is remembered  <=>  rho < rank(anchor) and fmin[rho] < fmin[rho'] for every rho' < rho,
//   factor = max(1, max over remembered rho with fmin[rho] != 0 of  (max_{r > rho} dmax[r]) / fmin[rho]),
// and ties (equal distances) only ever contribute ratios <= 1, which is why the reference's unstable std::sort cannot
// change the result either.  tests/test_oracle_placement.py checks this formulation against the
// sorted walk of the oracle on random inputs.  The accumulators live in LDS, [rank][thread] (bank = thread).
// ------------------------------------------------------------------------------------------------------------
struct BandAcc {
  float* fmin;  // this thread's column of s_fmin: element rank at fmin[rank * kBlock]
  float* dmax;
  uint32_t anchor_rank;
  __device__ void reset(uint32_t ra) {
    anchor_rank = ra;
    for (uint32_t r = 0; r <= ra; ++r) { fmin[r * kBlock] = FLT_MAX; dmax[r * kBlock] = -1.f; }
  }
  __device__ void save(float* g) const {  // a paused loop parks its accumulators in HBM
    for (uint32_t r = 0; r <= anchor_rank; ++r) { g[2 * r] = fmin[r * kBlock]; g[2 * r + 1] = dmax[r * kBlock]; }
  }
  __device__ void restore(uint32_t ra, const float* g) {
    anchor_rank = ra;
    for (uint32_t r = 0; r <= ra; ++r) { fmin[r * kBlock] = g[2 * r]; dmax[r * kBlock] = g[2 * r + 1]; }
  }
  __device__ void add(float dist, uint32_t rank) {  // BandFactor::addSequence (:267-269), rank = depth(LCA(node, anchor)) (:292-293)
    const float f = fmin[rank * kBlock], d = dmax[rank * kBlock];
    if (dist < f) fmin[rank * kBlock] = dist;
    if (dist > d) dmax[rank * kBlock] = dist;
  }
  __device__ float factor() const {  // getFactor (:271-277)
    float bandfactor = 1.f;
    unsigned long long remembered = 0ull;  // (ranks 0 .. kMaxDepth = 63)
    float lowmin = FLT_MAX;
    for (uint32_t r = 0; r < anchor_rank; ++r) {
      const float f = fmin[r * kBlock];
      if (f < lowmin) { remembered |= 1ull << r; lowmin = f; }
    }
    float dsuf = dmax[anchor_rank * kBlock];
    for (int r = (int)anchor_rank - 1; r >= 0; --r) {
      const float f = fmin[r * kBlock];
      if ((remembered >> r) & 1ull) {
        if (f != 0.f && dsuf >= 0.f) { const float q = dsuf / f; bandfactor = bandfactor < q ? q : bandfactor; }
      }
      const float d = dmax[r * kBlock];
      dsuf = dsuf < d ? d : dsuf;
    }
    return sqrtf(bandfactor);
  }
};

// A consumed alignment must come from readable segments, and (protein) must not have tripped the reference's asserts
// selfscore >= mutual / distance >= 0 (:205,:240), which the kernel flags by a NaN distance.
__device__ __forceinline__ bool result_ok(const WorkDev& w, uint32_t s, uint32_t err_a, uint32_t err_b, float dist) {
  if (err_a | err_b) { raise_error(w, err_a ? err_a : err_b, s); return false; }
  if (dist != dist || dist < 0.f) { raise_error(w, kErrAssert, s); return false; }  // (negative: a kDeferredMark nobody replaced -- never consumed by design)
  return true;
}

// A float q with   (double)qs / L >= t  =>  qs >= q   for every float qs, and equality of the two tests except within an
// ulp or two of the boundary: IEEE division by a positive constant is monotone, so the similarities that pass the
// reference's test (:765, a double division per hit) are exactly those from some float q* on; q* is one of the two floats
// around t * L (a quotient just below t may still round up to t), and the float below the rounded product is never above
// it.  k_emit uses this as its similarity test for pass-2 candidates -- emission may be a superset (a hit that squeezes
// through here but not through the reference's test gets an alignment nobody reads; k_decide applies the division) --
// instead of a double division per row, whose registers made that kernel spill.  One multiply, once per pass-2 anchor.
__device__ __forceinline__ float qs_threshold(double t, uint32_t L) {
  const float q = (float)(t * (double)L);
  uint32_t bits = __float_as_uint(q);
  if (!(q == q)) return q;                                   // NaN (t is finite in practice): nothing passes either test
  if ((bits & 0x7FFFFFFFu) == 0u) return __uint_as_float(0x80000001u);  // +-0 -> the negative float next to it
  if (q == -INFINITY) return q;
  bits += (bits >> 31) ? 1u : 0xFFFFFFFFu;                   // one float towards -infinity
  return __uint_as_float(bits);
}

// the log line "+ALN i <=> j  dist=..; sim=.." as a record (tracing on only: rpa_trace_begin)
__device__ __forceinline__ void trace_put(const WorkDev& w, uint32_t s, uint32_t seq, int pass, int i, int j, float dist, float sim) {
  const unsigned long long at = atomicAdd(w.trace_n, 1ull);
  if (at < w.trace_cap) w.trace[at] = TraceEntry{w.trace_set_base + s, seq, pass, i, j, dist, sim, 0u};
}

// ------------------------------------------------------------------------------------------------------------
// Decide: one thread per record set, one call per round.  Every loop over the hits keeps the NEXT row (64 bytes) and
// its alignment result in flight while it works on the current one; qgroup / outgroup are bit sets in WorkDev::flags,
// so "first member", "any member left" and "clear" never touch the rows.
// ------------------------------------------------------------------------------------------------------------
#ifndef RPA_DECIDE_WAVES
#define RPA_DECIDE_WAVES 2  // resident waves per SIMD the register budget of k_decide is set for (A/B: tools/build_variant.sh)
#endif
template <int Q, bool kTrace>  // kTrace: the variant that records the consumed alignments (rpa_trace_begin); the other carries no trace code
__global__ __launch_bounds__(kBlock, RPA_DECIDE_WAVES) void k_decide(BatchDev b, WorkDev w, TaxDev tax, Params p, OutRow* __restrict__ rows,
                                                    OutStats* __restrict__ stats, unsigned long long* prof_acc) {
  constexpr int PW = 4 * Q - 6;
  const uint32_t lbs = w.lb_shift;
  // BandFactor accumulators: (deepest taxonomy level + 1) ranks x 256 threads x {fmin, dmax}; sized at launch so that a
  // pruned NCBI tree (8 levels) needs 18 KB and the kernel can share a CU with the alignment kernel of another stream
  extern __shared__ float s_band[];
  float* s_fmin = s_band;
  float* s_dmax = s_band + (size_t)w.band_ranks * kBlock;
  __shared__ LiveStage s_live_stage;
  if (threadIdx.x == 0) s_live_stage.arrived = 0u;
  // the round's counters are spent (the alignment launches in front of this kernel were their last readers; this kernel counts
  // in ctr_live_out only): cleared here for the next round's k_emit instead of by a fill between the launches
  if (blockIdx.x == 0 && threadIdx.x < (uint32_t)kNumRoundCounters) {
    if (threadIdx.x != (uint32_t)kCtrActive) w.counters[threadIdx.x] = 0u;
    if (threadIdx.x == 0) w.counters[w.ctr_live_in] = 0u;
  }
  if (blockIdx.x == 0 && threadIdx.x >= 64u && threadIdx.x <= 64u + (uint32_t)kNumBins + 1u) w.counters[kCtrDeferred + (threadIdx.x - 64u)] = 0u;  // kCtrDeferred, kCtrSurvivor .., kCtrWindowFail
  static_assert(kCtrWindowFail == kCtrDeferred + kNumBins + 1 && 64 + kNumBins + 1 < kBlock, "the counters k_decide clears");
  __syncthreads();  // (the only barrier of the kernel: the wavefronts of a workgroup start together and finish far apart)
  // dense rounds: one thread per record set; sparse rounds (few sets still running): one thread per entry of the live list
  const uint32_t tid = blockIdx.x * kBlock + threadIdx.x;
  bool live = tid < (w.live_in ? w.n_live_in : b.n_sets);
  const uint32_t s = live ? (w.live_in ? w.live_in[tid] : tid) : 0u;
  SetState st;
  st.phase = kDone;
  uint32_t prof_done = 0;
  // Everything that depends on nothing but the set's number is asked for at once: its state, its static quad, its first row
  // position and its tile -- a wavefront of this kernel is bound by the number of dependent trips to memory it makes (state
  // -> offsets -> tile -> rows -> ... : profiles/r04_decide_phases.txt), and these four used to be three trips.
  uint4 pre_info = make_uint4(0u, 0u, 0u, 0u);
  uint32_t pre_lo = 0, pre_tile = 0;
  if (live) {
    pre_info = w.set_info[s];
    pre_lo = b.set_offsets[s];
    if (w.tile_base) pre_tile = w.tile_base[s >> 6];
    state_load(w, s, st, &pre_info);
  }
  live = live && st.phase != kDone;
  EmitInfo em{};
  bool write_back = false;  // this thread decided something: its request and its state go back to memory at the very end
  if (live) {
    em.mode = 0; em.n = st.n; em.L = st.L; em.qseglen = st.qseglen;
    const uint32_t lo = pre_lo;
    const uint32_t n = st.n, L = st.L;
    const float X = p.exclude_factor;
    const SetRows<Q> R = set_rows<Q>(w, s, lo, pre_tile);    // rows and alignment results of this set
    const float2* __restrict__ qres_p = w.pair_res + w.P + s;  // result of the anchor <-> query alignment
    const uint32_t nw = (n + 31u) >> 5;
    GroupWords fq{nw == 1u ? nullptr : w.flags + lo, st.gq};        // qgroup bits
    GroupWords fo{nw == 1u ? nullptr : w.flags + lo + nw, st.go};   // outgroup bits
    bool failed = false;

    // Normally one trip: consume the results of the phase the set waited for, then choose what to align next.  A set
    // whose pass 0 and first pass-1 loop were requested together (mode 4) takes a second trip through the pass-1 branch
    // right away when the anchor it then selects is the predicted one (hit 0).
    const uint32_t combined_hi = st.chunk_hi;  // mode 4: the pass-1 chunk that was requested together with pass 0
    // Pass 0 of a set whose pass 0 and first pass-1 loop were requested together (mode 4): hit 0 is the ONLY member of the
    // query group, so index_best, rtax and the first anchor are hit 0 whatever its distance is, and every other hit leaves
    // pass 0 with (FLT_MAX, identities) -- what k_rank_records wrote.  What is left of the pass-0 walk are two aggregates
    // over all hits (anchors_support = max similarity, lca_allnodes, :547-548): the pass-1 loop below gathers them while
    // it walks the same rows (and a short tail walk takes the rows that loop does not reach), instead of a first walk over
    // every row for their sake.
    bool acc_pending = false;   // the pass-1 walk of this call also gathers pass 0's aggregates
    uint32_t acc_next = 1;      // first hit they have not seen yet
    uint32_t acc_support = 0, acc_d_all = 0;
    if (st.phase == kPass0Wait && st.combined) {
      st.combined = 0;
      const HitCore h0 = R.core(0u);
      const float2 r2 = R.result(0);
      if (!result_ok(w, s, meta_err(h0.meta), st.qsegerr, r2.x)) failed = true;
      else {
        const float idf = (float)h0.ident;
        const float sim0 = r2.y < idf ? idf : r2.y;  // std::max(similarity, identities), :531
        if constexpr (kTrace) trace_put(w, s, st.cnt0 + st.cnt1 + st.cnt2, 0, 0, -1, r2.x, sim0);
        ++st.cnt0;
        R.store_qdqs(0u, r2.x, sim0);
        if (r2.x == 0.f) w.lite[lo] = (w.lite[lo] & kLiteLenMask) | kLiteZeroQd;  // (:620-621)
        for (uint32_t k = 0; k < nw; ++k) { fq[k] = 0u; fo[k] = 0u; }  // qgroup = {0}, taken by the anchor selection; outgroup empty
        const uint32_t d0 = meta_depth(h0.meta);
        st.best = 0; st.d_rtax = d0; st.ival_g = 0.f; st.d_lnode_g = st.d_unode_g = d0; st.bf_max = 1.f; st.lrd_min = 255;  // :553,569-579
        st.anchor = 0; st.phase = kPass1Wait; st.loop_live = 0;  // :582-583
        acc_pending = true; acc_support = (uint32_t)sim0; acc_d_all = d0;
      }
    }
    if (!failed)
    for (int trip = 0; trip < 2; ++trip) {
    if (st.phase == kPass0Wait) {  // ---- pass 0, :497-566 ----
      uint32_t path0[PW];
      uint32_t d0 = 0;
      float thr = 0.f;
      uint32_t index_best = 0;
      float best_d = 0.f, best_s = 0.f, best_sc = 0.f;
      uint32_t support = 0;
      uint32_t d_all = 0;  // lca_allnodes as a point on hit 0's path (:495)
      uint32_t qword = 0;
      Row<Q> nxt = R.load(0u), nx2 = nxt;  // rows i+1 and i+2 are in flight while row i is worked on
      float2 rn = R.result(0), rn2 = rn;
      if (n > 1) { nx2 = R.load(1); rn2 = R.result(1); }
      // (sets of up to 64 hits: members and zero-distance hits are noted as register bit sets and their stores -- the
      // (distance, similarity) pairs, the lite words, the query-group words -- done behind the walk, see pass 1)
      const bool small0 = n <= 64u;
      unsigned long long members0 = 0ull, zero0 = 0ull;
      auto walk0 = [&](auto small_tag) {
      constexpr bool kSmall = decltype(small_tag)::value;
      for (uint32_t i = 0; i < n; ++i) {
        const Row<Q> cur = nxt;
        const float2 r2 = rn;
        nxt = nx2; rn = rn2;
        if (i + 2 < n) { nx2 = R.load(i + 2); rn2 = R.result(i + 2); }
        const HitCore h = row_core<Q>(cur);
        uint32_t pth[PW];
        row_path<Q>(cur, pth);
        if (i == 0) {
#pragma unroll
          for (int d = 0; d < PW; ++d) path0[d] = pth[d];
          d0 = meta_depth(h.meta); d_all = d0;
          thr = p.reeval_factor * h.score;  // :499
        }
        float dist, sim;
        bool member = false;
        if (h.meta & kMetaPerfect) {  // :511-517
          member = true; dist = 0.f; sim = (float)h.ident;
        } else if (h.score >= thr) {  // :518-534
          member = true;
          if (!result_ok(w, s, meta_err(h.meta), st.qsegerr, r2.x)) { failed = true; break; }
          dist = r2.x;
          const float idf = (float)h.ident;
          sim = r2.y < idf ? idf : r2.y;  // std::max(similarity, identities), :531
          if constexpr (kTrace) trace_put(w, s, st.cnt0 + st.cnt1 + st.cnt2, 0, (int)i, -1, dist, sim);
          ++st.cnt0;
        } else {
          dist = FLT_MAX; sim = (float)h.ident;
        }
        if constexpr (kSmall) {
          if (member) members0 |= 1ull << i;
          if (dist == 0.f) zero0 |= 1ull << i;
        } else {
          if (member) R.store_qdqs(i, dist, sim);  // the others keep what k_rank_records wrote: (FLT_MAX, identities)
          if (dist == 0.f) w.lite[lo + i] = (w.lite[lo + i] & kLiteLenMask) | kLiteZeroQd;  // pass 1 will not align this hit (:620-621)
          if (member) qword |= 1u << (i & 31u);
          if ((i & 31u) == 31u || i + 1 == n) { fq[i >> 5] = qword; qword = 0; }
        }
        if (i == 0) { best_d = dist; best_s = sim; best_sc = h.score; }  // index_best starts at 0 (:500)
        else if (dist < best_d) { index_best = i; best_d = dist; best_s = sim; best_sc = h.score; }  // :542-546
        else if (dist == best_d) {
          if (sim > best_s) { index_best = i; best_s = sim; best_sc = h.score; }
          else if (sim == best_s && h.score > best_sc) { index_best = i; best_sc = h.score; }
        }
        support = max(support, (uint32_t)sim);  // :547
        d_all = min(d_all, common<PW>(path0, d0, pth, meta_depth(h.meta), lbs));  // :548
      }
      };
      if (small0) walk0(std::true_type{}); else walk0(std::false_type{});
      if (small0 && !failed) {
        for (unsigned long long m = members0; m; m &= m - 1ull) {  // same values as in the walk, from the same row and result
          const uint32_t k = (uint32_t)__ffsll((long long)m) - 1u;
          const HitCore h = R.core(k);
          const float idf = (float)h.ident;
          if (h.meta & kMetaPerfect) R.store_qdqs(k, 0.f, idf);
          else { const float2 r2 = R.result(k); R.store_qdqs(k, r2.x, r2.y < idf ? idf : r2.y); }
          if ((zero0 >> k) & 1ull) w.lite[lo + k] = (w.lite[lo + k] & kLiteLenMask) | kLiteZeroQd;
        }
        for (uint32_t k = 0; k < nw; ++k) fq[k] = (uint32_t)(members0 >> (32u * k));
      }
      if (!failed) {
        for (uint32_t k = 0; k < nw; ++k) fo[k] = 0u;  // outgroup starts empty (:578)
        const Row<Q> rb = R.load(index_best);
        uint32_t pb[PW];
        row_path<Q>(rb, pb);
        const uint32_t db = meta_depth(row_core<Q>(rb).meta);
        uint32_t d_rtax = db;  // rtax = node[index_best] (:553), then LCA with every member that stays in qgroup
        bool any = false;
        for (uint32_t k = 0; k < nw; ++k) {  // :554-562, members only
          uint32_t word = fq[k], keep = word;
          while (word) {
            const uint32_t bit = (uint32_t)__ffs((int)word) - 1u;
            word &= word - 1u;
            const Row<Q> ri = R.load(k * 32u + bit);
            const HitCore h = row_core<Q>(ri);
            if (h.qd != best_d || h.qs != best_s || h.score != best_sc) keep &= ~(1u << bit);
            else {
              uint32_t pth[PW];
              row_path<Q>(ri, pth);
              d_rtax = min(d_rtax, common<PW>(pb, db, pth, meta_depth(h.meta), lbs));
              any = true;
            }
          }
          fq[k] = keep;
        }
        if (!any) { raise_error(w, kErrAssert, s); failed = true; }  // assert(!qgroup.empty()), :563
        st.support = support; st.best = index_best; st.d_lca_all = d_all; st.d_rtax = d_rtax;
        st.ival_g = 0.f; st.d_lnode_g = st.d_unode_g = d_rtax; st.bf_max = 1.f; st.lrd_min = 255;  // :569-579
        st.phase = kPass1Select;
      }
    } else if (st.phase == kPass1Wait) {  // ---- pass 1 for one anchor, :580-728 ----
      const uint32_t a = st.anchor;
      uint32_t pa[PW], pb[PW];
      // (the loads that depend on the state alone go out together: the anchor's row, the best hit's, the walk's first two)
      const uint32_t i_first = st.loop_live ? st.i_next : 0u;
      const Row<Q> ra = R.load(a);
      const Row<Q> rb = R.load(st.best);
      Row<Q> nxt = ra, nx2 = ra;  // rows i+1 and i+2 are in flight while row i is worked on
      float2 rn = make_float2(0.f, 0.f), rn2 = rn;
      if (i_first < n) { nxt = R.load(i_first); rn = R.result(i_first); }
      if (i_first + 1 < n) { nx2 = R.load(i_first + 1); rn2 = R.result(i_first + 1); }
      const HitCore ha = row_core<Q>(ra);
      row_path<Q>(ra, pa);
      const uint32_t da = meta_depth(ha.meta);
      row_path<Q>(rb, pb);
      const uint32_t db = meta_depth(row_core<Q>(rb).meta);
      const float qdist = ha.qd;
      uint32_t* __restrict__ ogi = w.og_idx + lo;
      int32_t* __restrict__ ogd = w.og_dist + lo;
      BandAcc band{s_fmin + threadIdx.x, s_dmax + threadIdx.x, 0};
      // loop state: fresh for a new anchor, restored when the previous round paused at the end of its chunk
      uint32_t nog, d_lnode, i;
      float ldist, udist;
      double qpid_upper, cut_g, cut_h;  // :599-601
      int scut;                         // :602 (int!)
      if (!st.loop_live) {
        band.reset(da);  // BandFactor(...); addSequence(0, rnode) (:581,586): the anchor entry only sets the start rank
        nog = 0; d_lnode = st.d_rtax /* lnode = rtax, :587 */; ldist = 0.f; udist = FLT_MAX;
        qpid_upper = cut_g = cut_h = 0.; scut = 0; i = 0;
      } else {
        band.restore(da, w.band_save + (size_t)s * 2 * w.band_ranks);
        nog = st.nog; d_lnode = st.d_lnode; ldist = st.ldist; udist = st.udist;
        qpid_upper = st.qpid_upper; cut_g = st.cut_g; cut_h = st.cut_h; scut = st.scut; i = st.i_next;
      }
      // Sets of up to 64 hits keep outgroup_tmp as a BIT SET (hit i is a member; its distance is re-read from its result
      // slot, which nothing overwrites while this anchor's loop lasts) and the query-group bits to clear as another, both in
      // registers: the walk then contains no store at all.  On gfx9 loads and stores share one counter (vmcnt) and may
      // complete out of order with each other, so a single store that may be pending makes every wait for a prefetched row
      // a wait for ALL outstanding memory operations -- the two rows in flight were worth nothing.
      const bool small = n <= 64u;
      unsigned long long ogmask = (small && st.loop_live) ? ((unsigned long long)st.pad_ << 32) | st.nog : 0ull;
      unsigned long long fqclr = 0ull;
      // ... and, for sets of up to 32 hits on trees of up to 15 levels in a loop that starts in this call, the depth of
      // LCA(member, best hit) of every outgroup member as a nibble (the walk has the member's path in registers; the member
      // loop behind it then re-reads the members' RESULTS only, not their rows: in a first pass-1 round nearly every hit is a
      // member, and the rows were fetched twice -- 1.9 KB per record set where the walk needs 0.9)
      const bool cb_packed = small && n <= 32u && w.band_ranks <= 16u && !st.loop_live;
      unsigned long long cb_lo = 0ull, cb_hi = 0ull;
      bool paused = false;
      auto walk = [&](auto small_tag) {
      constexpr bool kSmall = decltype(small_tag)::value;
      for (; d_lnode != 0 && i < n; ++i) {  // :604 (lnode != root)
        const Row<Q> cur = nxt;
        const float2 r2 = rn;
        nxt = nx2; rn = rn2;
        if (i + 2 < n) { nx2 = R.load(i + 2); rn2 = R.result(i + 2); }
        const HitCore h = row_core<Q>(cur);
        if (!(h.score >= (float)scut)) break;
        uint32_t pth[PW];
        row_path<Q>(cur, pth);
        const uint32_t dh = meta_depth(h.meta);
        if (acc_pending && i >= acc_next) {  // pass 0's aggregates for this hit (it is not hit 0: similarity = identities)
          acc_support = max(acc_support, h.ident);
          acc_d_all = min(acc_d_all, common<PW>(pb, db, pth, dh, lbs));  // best = hit 0: pb is its path
          acc_next = i + 1;
        }
        const double qsearchpid = (double)h.ident / L;
        const double qpid = (double)h.qs / L;
        const double qpid_thresh = cut_g < cut_h ? cut_h : cut_g;
        if (!(qpid >= qpid_thresh)) continue;  // :612
        float dist;
        if (i == a) dist = 0.f;
        else if (h.qd == 0.f) dist = qdist;  // :620-621
        else {
          if (i >= st.chunk_hi) { paused = true; break; }  // this alignment was not part of the chunk emitted last round
          if (!result_ok(w, s, meta_err(h.meta), meta_err(ha.meta), r2.x)) { failed = true; break; }
          dist = r2.x;
          if constexpr (kTrace) trace_put(w, s, st.cnt0 + st.cnt1 + st.cnt2, 1, (int)i, (int)a, dist, r2.y);
          ++st.cnt1;
        }
        band.add(dist, common<PW>(pa, da, pth, dh, lbs));  // :641
        if (dist == 0.f) {  // :644
          if constexpr (kSmall) fqclr |= 1ull << i; else bit_clear(fq, i);
        } else if (dist <= qdist) {  // :646-650
          d_lnode = min(d_lnode, common<PW>(pb, db, pth, dh, lbs));
          if (dist > ldist) ldist = dist;
        } else {  // :651-662
          if (dist < udist) {
            udist = dist;
            if (qsearchpid > qpid_upper) {
              qpid_upper = qsearchpid;
              cut_g = qsearchpid * 2. - 1.;
              cut_h = qsearchpid * X;
            }
            if (!scut) scut = float_to_int_x86(h.score * X);
          }
          if constexpr (kSmall) {
            ogmask |= 1ull << i;
            if (cb_packed) {
              const unsigned long long cbn = (unsigned long long)common<PW>(pb, db, pth, dh, lbs) << ((i & 15u) * 4u);
              if (i < 16u) cb_lo |= cbn; else cb_hi |= cbn;
            }
          } else { ogi[nog] = i; ogd[nog] = (int)dist; ++nog; }
        }
      }
      };
      if (small) walk(std::true_type{}); else walk(std::false_type{});
      if (small && fqclr) {
        for (uint32_t k = 0; k < nw; ++k) {
          const uint32_t clr = (uint32_t)(fqclr >> (32u * k));
          if (clr) fq[k] &= ~clr;
        }
      }
      if (acc_pending && !failed) {  // the hits the loop above did not reach (early exit, pause): identities and path only
        Row<Q> nx = ra;
        if (acc_next < n) nx = R.load(acc_next);
        for (uint32_t k = acc_next; k < n; ++k) {
          const Row<Q> cur = nx;
          if (k + 1 < n) nx = R.load(k + 1);
          const HitCore h = row_core<Q>(cur);
          uint32_t pth[PW];
          row_path<Q>(cur, pth);
          acc_support = max(acc_support, h.ident);
          acc_d_all = min(acc_d_all, common<PW>(pb, db, pth, meta_depth(h.meta), lbs));
        }
        st.support = acc_support; st.d_lca_all = acc_d_all;
        acc_pending = false;
      }
      if (paused) {  // ask for the next chunk of this anchor's loop and come back
        st.loop_live = 1; st.i_next = i; st.d_lnode = d_lnode; st.ldist = ldist; st.udist = udist;
        if (small) { st.nog = (uint32_t)ogmask; st.pad_ = (uint32_t)(ogmask >> 32); } else st.nog = nog;
        st.qpid_upper = qpid_upper; st.cut_g = cut_g; st.cut_h = cut_h; st.scut = scut;
        band.save(w.band_save + (size_t)s * 2 * w.band_ranks);
        st.chunk_hi = p.spec1 ? min(n, i + (p.grow ? max(p.spec1, i) : p.spec1)) : n;
        if (assume_first_hit_cutoff(p) || (p.filter1 && scut)) st.chunk_hi = chunk_end_by_score(R, i, st.chunk_hi, scut, X, assumed_cut_den(p));  // (a cut-off the loop already has: exact)
        em.mode = 2; em.anchor = a; em.anchor_len = w.lite[lo + a] & kLiteLenMask; em.i_lo = i; em.i_hi = st.chunk_hi;
        const double cut = cut_g < cut_h ? cut_h : cut_g;
        if (p.filter1 && (scut != 0 || cut > 0.)) {  // the chunk without the hits the loop will break at or step over anyway
          em.mode = 5; em.fcut = (float)scut; em.qs_min = qs_threshold(cut, L);
        }
      } else
      if (!failed) {
        const float bandfactor = band.factor();  // :667
        st.bf_max = st.bf_max < bandfactor ? bandfactor : st.bf_max;
        const float qdist_ex = qdist * bandfactor;  // :669
        float min_upper_dist = (float)INT_MAX;       // :670
        // :673-687.  The reference also erases entries here (distance beyond both the band and the current minimum);
        // an erased entry stays above the final min_upper_dist (that value only rises while it is <= qdist_ex), so the
        // walk below skips it by its own test and the erase needs no copy.
        auto upper = [&](float dist) {
          if (dist > qdist_ex) {
            if (dist < min_upper_dist) min_upper_dist = dist;
          } else {
            if (min_upper_dist > qdist_ex) min_upper_dist = dist;
            else min_upper_dist = min_upper_dist < dist ? dist : min_upper_dist;
          }
        };
        if (small) {  // members in index order, four result loads in flight; (float)(int): outgroup_tmp holds int distances
          unsigned long long m = ogmask;
          while (m) {
            uint32_t io[4];
            float dd[4];
            int cnt = 0;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              io[q] = 0u;
              if (m) { io[q] = (uint32_t)__ffsll((long long)m) - 1u; m &= m - 1ull; cnt = q + 1; }
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) dd[q] = q < cnt ? R.result(io[q]).x : 0.f;
#pragma unroll
            for (int q = 0; q < 4; ++q)
              if (q < cnt) upper((float)(int)dd[q]);
          }
        } else {
          for (uint32_t k = 0; k < nog; ++k) upper((float)ogd[k]);
        }
        uint32_t d_unode = d_lnode;  // unode = lnode, :690 -- (float)INT_MAX != FLT_MAX, so always taken
        auto member_cb = [&](uint32_t io, uint32_t cb) {  // an outgroup_tmp entry within min_upper_dist, :691-711; cb = depth of LCA(cnode, best hit)
          d_unode = min(d_unode, cb);                          // unode = LCA(cnode, unode)
          const uint32_t lca_root_dist = min(st.d_rtax, cb);   // depth(LCA(cnode, rtax))
          if (lca_root_dist > st.lrd_min) return;
          if (lca_root_dist < st.lrd_min) {
            st.lrd_min = lca_root_dist;
            for (uint32_t kk = 0; kk < nw; ++kk) fo[kk] = 0u;  // outgroup.clear()
          }
          bit_set(fo, io);
        };
        auto member = [&](uint32_t io, float dist, const uint32_t (&pth)[PW], uint32_t depth) {
          if (dist > min_upper_dist) return;
          member_cb(io, common<PW>(pb, db, pth, depth, lbs));
        };
        if (cb_packed) {  // (the nibbles of the walk above: a member is its result slot)
          unsigned long long mleft = ogmask;
          while (mleft) {
            uint32_t io[4];
            float dd[4];
            int cnt = 0;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              io[q] = 0u;
              if (mleft) { io[q] = (uint32_t)__ffsll((long long)mleft) - 1u; mleft &= mleft - 1ull; cnt = q + 1; }
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) dd[q] = q < cnt ? (float)(int)R.result(io[q]).x : 0.f;
#pragma unroll
            for (int q = 0; q < 4; ++q)
              if (q < cnt && !(dd[q] > min_upper_dist))
                member_cb(io[q], (uint32_t)(((io[q] < 16u ? cb_lo : cb_hi) >> ((io[q] & 15u) * 4u)) & 15ull));
          }
        } else if (small) {
          // members in index order; what a member needs (its distance, its path, its depth) is asked for FOUR members at a time
          // (narrow rows): a member used to be two dependent trips to memory, its result and then its row
          constexpr int kBatch = PW <= 2 ? 4 : 1;
          unsigned long long mleft = ogmask;
          while (mleft) {
            uint32_t io[kBatch], depth[kBatch], pth[kBatch][PW];
            float dd[kBatch];
            int cnt = 0;
#pragma unroll
            for (int q = 0; q < kBatch; ++q) {
              io[q] = 0u;
              if (mleft) { io[q] = (uint32_t)__ffsll((long long)mleft) - 1u; mleft &= mleft - 1ull; cnt = q + 1; }
            }
#pragma unroll
            for (int q = 0; q < kBatch; ++q) {
              if (q < cnt) {
                dd[q] = (float)(int)R.result(io[q]).x;  // (float)(int): outgroup_tmp holds int distances
                R.path(io[q], pth[q]);
                depth[q] = meta_depth(R.meta(io[q]));
              }
            }
#pragma unroll
            for (int q = 0; q < kBatch; ++q)
              if (q < cnt) member(io[q], dd[q], pth[q], depth[q]);
          }
        } else {
          for (uint32_t k = 0; k < nog; ++k) {
            const uint32_t io = ogi[k];
            const Row<Q> ri = R.load(io);
            uint32_t pth[PW];
            row_path<Q>(ri, pth);
            member(io, (float)ogd[k], pth, meta_depth(row_core<Q>(ri).meta));
          }
        }
        float ival = 0.f;  // :714-719
        if (d_unode != d_lnode && ldist < qdist) ival = (qdist - ldist) / (udist - ldist);
        st.ival_g = st.ival_g < ival ? ival : st.ival_g;   // :724
        st.d_unode_g = min(st.d_unode_g, d_unode);         // :726
        st.d_lnode_g = min(st.d_lnode_g, d_lnode);         // :727
        uint32_t more = 0;
        for (uint32_t k = 0; k < nw; ++k) more |= fq[k];
        st.phase = (more && st.d_lnode_g != 0) ? kPass1Select : kPass2Select;  // :729
        st.loop_live = 0;
      }
    } else if (st.phase == kPass2Wait) {  // ---- pass 2 for one anchor, :750-820 ----
      const uint32_t a = st.anchor;
      // every load that depends on the state alone goes out before anything is looked at: the anchor's and the best hit's
      // rows, the thresholds of the anchor, the first two rows of the walk, the anchor <-> query result (one trip, not four)
      uint32_t i = st.loop_live ? st.i_next : 0u;
      const Row<Q> ra = R.load(a);
      const Row<Q> rb = R.load(st.best);
      const EmitInfo sel = w.emit[s];            // thresholds were fixed when the anchor was selected (:750-754)
      Row<Q> nxt = ra, nx2 = ra;  // rows i+1 and i+2 are in flight while row i is worked on
      float2 rn = make_float2(0.f, 0.f), rn2 = rn;
      if (i < n) { nxt = R.load(i); rn = R.result(i); }
      if (i + 1 < n) { nx2 = R.load(i + 1); rn2 = R.result(i + 1); }
      const float2 q2_pre = *qres_p;             // (read by the walk only when the anchor was never aligned to the query, :792)
      const HitCore ha = row_core<Q>(ra);
      uint32_t pb[PW];
      row_path<Q>(rb, pb);
      const uint32_t db = meta_depth(row_core<Q>(rb).meta);
      float qd_a = ha.qd, qs_a = ha.qs;
      const double qpid_thresh = sel.qpid_thresh;
      const float scut = sel.fcut;
      bool paused = false;
      // (sets of up to 64 hits: the stores of the walk -- querydistance[i], the anchor's pair, outgroup bits to clear -- are
      // noted in registers and done behind it, for the same reason as in pass 1)
      const bool small = n <= 64u;
      unsigned long long qdmask = 0ull, foclr = 0ull;
      bool anchor_pair = false;
      auto walk = [&](auto small_tag) {
      constexpr bool kSmall = decltype(small_tag)::value;
      for (; i < n; ++i) {  // :757
        const Row<Q> cur = nxt;
        const float2 r2 = rn;
        nxt = nx2; rn = rn2;
        if (i + 2 < n) { nx2 = R.load(i + 2); rn2 = R.result(i + 2); }
        const HitCore h = row_core<Q>(cur);
        if (!(h.score >= scut)) break;
        const double qpid = (double)(i == a ? qs_a : h.qs) / L;  // the anchor's row may have been updated below
        if (!(qpid >= qpid_thresh)) continue;
        uint32_t pth[PW];
        row_path<Q>(cur, pth);
        const uint32_t cb = common<PW>(pb, db, pth, meta_depth(h.meta), lbs);
        float dist;
        if (i == a) dist = 0.f;
        else {
          if (cb >= st.d_unode_g) continue;  // isParentOf(unode_global, cnode) || cnode == unode_global, :771
          if (i >= st.chunk_hi) { paused = true; break; }  // not part of the chunk emitted last round
          if (!result_ok(w, s, meta_err(h.meta), meta_err(ha.meta), r2.x)) { failed = true; break; }
          dist = r2.x;
          if constexpr (kTrace) trace_put(w, s, st.cnt0 + st.cnt1 + st.cnt2, 2, (int)i, (int)a, dist, r2.y);
          ++st.cnt2;
          if constexpr (kSmall) qdmask |= 1ull << i; else R.store_qd(i, dist);  // querydistance[i] = dist, :785
        }
        if (dist == 0.f) {  // :789
          if constexpr (kSmall) foclr |= 1ull << i; else bit_clear(fo, i);
        } else {
          float qdist_ex;
          if (qd_a == FLT_MAX) {  // :792-808
            const float2 q2 = q2_pre;
            if (!result_ok(w, s, meta_err(ha.meta), st.qsegerr, q2.x)) { failed = true; break; }
            const float d2 = q2.x;
            const float s2 = q2.y < qs_a ? qs_a : q2.y;
            qd_a = d2; qs_a = s2;
            if constexpr (kSmall) anchor_pair = true; else R.store_qdqs(a, d2, s2);
            qdist_ex = d2 * st.bf_max;
            if constexpr (kTrace) trace_put(w, s, st.cnt0 + st.cnt1 + st.cnt2, 2, -1, (int)a, d2, s2);
            ++st.cnt2;
          } else qdist_ex = qd_a * st.bf_max;  // :809
          if (dist <= qdist_ex) st.d_unode_g = min(st.d_unode_g, cb);  // unode_global = LCA(unode_global, cnode), :811-813
        }
      }
      };
      if (small) walk(std::true_type{}); else walk(std::false_type{});
      if (small && !failed) {
        if (anchor_pair) R.store_qdqs(a, qd_a, qs_a);
        for (unsigned long long m = qdmask; m;) {  // querydistance[i] = dist (:785), four results asked for at a time
          uint32_t kk[4];
          float dd[4];
          int cnt = 0;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            kk[q] = 0u;
            if (m) { kk[q] = (uint32_t)__ffsll((long long)m) - 1u; m &= m - 1ull; cnt = q + 1; }
          }
#pragma unroll
          for (int q = 0; q < 4; ++q) dd[q] = q < cnt ? R.result(kk[q]).x : 0.f;
#pragma unroll
          for (int q = 0; q < 4; ++q)
            if (q < cnt) R.store_qd(kk[q], dd[q]);
        }
        for (uint32_t k = 0; k < nw && foclr; ++k) {
          const uint32_t clr = (uint32_t)(foclr >> (32u * k));
          if (clr) fo[k] &= ~clr;
        }
      }
      if (paused) {
        st.loop_live = 1; st.i_next = i;
        st.chunk_hi = p.spec2 ? min(n, i + (p.grow ? max(p.spec2, i) : p.spec2)) : n;
        em = sel; em.mode = 3; em.i_lo = i; em.i_hi = st.chunk_hi; em.emit_query = 0; em.d_unode_g = st.d_unode_g;
      } else if (!failed) { st.phase = kPass2Select; st.loop_live = 0; }
    }

    // ---- choose what to align next ----
    if (!failed && st.phase == kPass1Select) {  // :582-583: anchor = first member of qgroup
      uint32_t a = n - 1;  // (the reference's scan would stop at the last hit if the group were empty; it never is)
      for (uint32_t k = 0; k < nw; ++k) {
        const uint32_t word = fq[k];
        if (word) { const uint32_t bit = (uint32_t)__ffs((int)word) - 1u; a = k * 32u + bit; fq[k] = word & (word - 1u); break; }
      }
      st.anchor = a; st.phase = kPass1Wait; st.loop_live = 0;
      st.chunk_hi = p.spec1 ? min(n, p.spec1) : n;
      if (assume_first_hit_cutoff(p)) {  // the first hit this loop aligns: not the anchor, not at distance 0 from the query (:620)
        uint32_t r = 0;
        while (r < st.chunk_hi && (r == a || (w.lite[lo + r] & kLiteZeroQd))) ++r;
        if (r < st.chunk_hi) st.chunk_hi = chunk_end_by_score(R, r, st.chunk_hi, 0, X, assumed_cut_den(p));
      }
      em.mode = 2; em.anchor = a; em.anchor_len = w.lite[lo + a] & kLiteLenMask; em.i_lo = 0; em.i_hi = st.chunk_hi;
    } else if (!failed && st.phase == kPass2Select) {  // :739-747
      bool found = false;
      if (st.d_unode_g == st.d_lca_all) {  // unode_global == lca_allnodes (both on the best hit's path): every anchor is skipped
        for (uint32_t k = 0; k < nw; ++k) fo[k] = 0u;
      } else {
        for (uint32_t k = 0; k < nw && !found; ++k) {
          const uint32_t word = fo[k];
          if (!word) continue;
          const uint32_t bit = (uint32_t)__ffs((int)word) - 1u;
          fo[k] = word & (word - 1u);
          const uint32_t a = k * 32u + bit;
          const HitCore h = R.core(a);
          st.anchor = a; st.phase = kPass2Wait; found = true; st.loop_live = 0;
          st.chunk_hi = p.spec2 ? min(n, p.spec2_first ? p.spec2_first : p.spec2) : n;
          em.mode = 3; em.anchor = a; em.anchor_len = w.lite[lo + a] & kLiteLenMask; em.best = st.best; em.d_unode_g = st.d_unode_g;
          em.i_lo = 0; em.i_hi = st.chunk_hi;
          em.emit_query = h.qd == FLT_MAX ? 1u : 0u;             // anchor never aligned to the query yet (:792)
          em.fcut = h.score * X;                                   // qsearchscore_thresh_heuristic, :754
          const double qpid_anchor = (double)h.qs / L;             // :750-753
          const double g2 = qpid_anchor * 2. - 1., h3 = qpid_anchor * X;
          em.qpid_thresh = g2 < h3 ? h3 : g2;
          em.qs_min = qs_threshold(em.qpid_thresh, L);
        }
      }
      if (!found) {  // :824-833
        if (st.d_unode_g == st.d_lnode_g) st.ival_g = 1.f;
        const uint32_t nb = R.core(st.best).node;  // the nodes of the result are points on the best hit's root path
        OutRow row;
        row.begin = st.qrstart; row.end = st.qrstop; row.ival = st.ival_g;
        row.lower = anc_node(tax, nb, st.d_lnode_g); row.upper = anc_node(tax, nb, st.d_unode_g); row.support = st.support;
        row.rtax = anc_node(tax, nb, st.d_rtax); row.flags = 0;
        rows[s] = row;
        if (stats) stats[s] = OutStats{n, st.cnt0, st.cnt1, st.cnt2};
        prof_done = st.cnt0 + st.cnt1 + st.cnt2;  // (demanded alignments of a finished set: added up per wavefront below)
        if (st.d_unode_g > st.d_lnode_g) raise_error(w, kErrAssert, s);  // setNodeRange assert, predictionrecord.hh:153
        st.phase = kDone;
      }
    }
    if (failed || trip == 1 || !st.combined) break;
    st.combined = 0;
    if (!(st.phase == kPass1Wait && st.anchor == 0)) break;  // (cannot happen with a single member; the results are then simply not used)
    st.chunk_hi = combined_hi;  // what was actually requested (the anchor selection just set it from the current chunk size)
    em.mode = 0;                // nothing asked for yet: the second trip decides
    }
    if (failed) { st.phase = kDone; em.mode = 0; }
    st.gq = fq.reg; st.go = fo.reg;
    write_back = true;
  }
  if (prof_acc) {  // (uniform; one atomic per wavefront, not one per finished record set: a million on 64 addresses per config-2 step)
    uint32_t v = prof_done;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += (uint32_t)__shfl_xor(v, d, 64);
    if ((threadIdx.x & 63) == 0 && v) atomicAdd(&prof_acc[(blockIdx.x & (kProfParts - 1)) * kProfSlots + 3], (unsigned long long)v);
  }
  // The live-list hand-over comes BEFORE the thread's last stores (round 5): its release fence waits for every store the
  // wavefront has in flight, and with the 80-byte state record and the request quads just issued that wait was most of the
  // epilogue's 28 % of a wavefront's life (profiles/r04_decide_phases.txt); now the stores are the last thing a wavefront does
  // and it ends without waiting for them.
  append_live_last_wave(w, s_live_stage, live && st.phase != kDone, s);  // still running: next round's live list
  if (write_back) {
    if (em.mode == 3u || em.mode == 5u) {  // thresholds that only the full record holds (k_decide reads it back), and their 16-byte form for k_emit
      w.emit[s] = em;
      w.emit_ext[s] = emit_ext_quad(em);
      if (em.best > 0x00FFFFFFu) raise_error(w, kErrUnsupported, s);  // (a record set of more than 16 M records)
    }
    w.emit_lite[s] = emit_desc(em);
    state_store(w, s, st, st.loop_live != 0u && st.phase == kPass1Wait);
  }
}

// ---- tiled row layout: per tile of 64 record sets, 64 x (largest record count of the tile) row slots ----
__global__ __launch_bounds__(kBlock) void k_tile_caps(const uint32_t* __restrict__ set_offsets, uint32_t n_sets, uint32_t n_tiles,
                                                       uint32_t* __restrict__ tile_base, uint32_t* __restrict__ max_set, bool keep_caps) {
  const uint32_t t = blockIdx.x * kBlock + threadIdx.x;
  uint32_t cap = 0;
  if (t < n_tiles) {
    const uint32_t s0 = t * 64u, s1 = min(n_sets, s0 + 64u);
    uint32_t prev = set_offsets[s0];
    for (uint32_t s = s0; s < s1; ++s) { const uint32_t nx = set_offsets[s + 1]; cap = max(cap, nx - prev); prev = nx; }
    if (keep_caps) tile_base[t] = cap;  // scanned in place by k_tile_scan
  }
  wave_max_to(max_set, cap);  // the largest record set of the batch sets the width of the taxon path labels
}
// exclusive prefix sum of 64 * cap over the tiles, one workgroup; saturates at 0xFFFFFFFF (the caller then uses the linear layout)
__global__ __launch_bounds__(1024) void k_tile_scan(uint32_t n_tiles, uint32_t* __restrict__ tile_base) {
  __shared__ unsigned long long s_sum[1024];
  const uint32_t per = (n_tiles + 1023u) / 1024u;
  const uint32_t lo = min(n_tiles, threadIdx.x * per), hi = min(n_tiles, lo + per);
  unsigned long long sum = 0;
  for (uint32_t t = lo; t < hi; ++t) sum += 64ull * tile_base[t];
  s_sum[threadIdx.x] = sum;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned long long run = 0;
    for (int k = 0; k < 1024; ++k) { const unsigned long long c = s_sum[k]; s_sum[k] = run; run += c; }
    tile_base[n_tiles] = run > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)run;
  }
  __syncthreads();
  unsigned long long run = s_sum[threadIdx.x];
  for (uint32_t t = lo; t < hi; ++t) {
    const unsigned long long c = 64ull * tile_base[t];
    tile_base[t] = run > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)run;
    run += c;
  }
}

hipError_t launch_tile_plan(hipStream_t st, const uint32_t* set_offsets, uint32_t n_sets, uint32_t* tile_base, uint32_t* max_set, bool scan) {
  const uint32_t n_tiles = (n_sets + 63u) / 64u;
  if (n_tiles == 0) return hipSuccess;
  hipLaunchKernelGGL(k_tile_caps, dim3((n_tiles + kBlock - 1) / kBlock), dim3(kBlock), 0, st, set_offsets, n_sets, n_tiles, tile_base, max_set, scan);
  if (scan) hipLaunchKernelGGL(k_tile_scan, dim3(1), dim3(1024), 0, st, n_tiles, tile_base);
  return hipGetLastError();
}

// Batch sanity, before any kernel follows an index of the caller's (the C ABI takes device pointers: nothing else stands
// between a bad index and a page fault): set_offsets runs from 0 to n_records without ever decreasing, every taxonomy node
// index is inside the taxonomy.  err[0] = error code (first one wins), err[1] = offending set / record.
// rec_pre (may be null): preorder number of every record's taxon (0 for a bad index), for k_rank_records' path labels --
// this kernel has the node column in its hands anyway, and k_rank_records then stages the numbers like any other column
// instead of chasing node -> pre at the head of every workgroup.
__global__ __launch_bounds__(kBlock) void k_validate_batch(BatchDev b, uint32_t n_nodes, uint32_t* __restrict__ err,
                                                            const uint32_t* __restrict__ pre, uint32_t* __restrict__ rec_pre) {
  const uint32_t t = blockIdx.x * kBlock + threadIdx.x;
  uint32_t bad = 0xFFFFFFFFu;
  if (t < b.n_sets) {
    const uint32_t lo = b.set_offsets[t], hi = b.set_offsets[t + 1];
    if (lo > hi || hi > b.n_records || (t == 0 && lo != 0u) || (t + 1u == b.n_sets && hi != b.n_records)) bad = t;
  }
  if (t < b.n_records) {
    const uint32_t node = b.node[t];
    if (node >= n_nodes) { if (bad == 0xFFFFFFFFu) bad = t; if (rec_pre) rec_pre[t] = 0u; }
    else if (rec_pre) rec_pre[t] = pre[node];
  }
  if (bad != 0xFFFFFFFFu && atomicCAS(&err[0], 0u, (uint32_t)kErrInvalid) == 0u) err[1] = bad;
}
hipError_t launch_validate_batch(hipStream_t st, const BatchDev& b, uint32_t n_nodes, uint32_t* err, const uint32_t* pre, uint32_t* rec_pre) {
  const uint32_t n = b.n_sets > b.n_records ? b.n_sets : b.n_records;
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(k_validate_batch, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, st, b, n_nodes, err, pre, rec_pre);
  return hipGetLastError();
}

// set offsets of a part of a batch, rebased so that its first record is record 0
__global__ __launch_bounds__(kBlock) void k_rebase_offsets(const uint32_t* __restrict__ in, uint32_t n, uint32_t* __restrict__ out) {
  const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
  if (i < n) out[i] = in[i] - in[0];
}
hipError_t launch_rebase_offsets(hipStream_t st, const uint32_t* in, uint32_t n, uint32_t* out) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(k_rebase_offsets, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, st, in, n, out);
  return hipGetLastError();
}

// ---- launchers: Q = 16-byte quads per hit row (WorkDev::row_quads, one of kRowQuadChoices) ----
#define RPA_DISPATCH_Q(QV, CALL)                 \
  switch (QV) {                                  \
    case 2: { constexpr int Q = 2; CALL; } break; \
    case 3: { constexpr int Q = 3; CALL; } break; \
    case 4: { constexpr int Q = 4; CALL; } break; \
    case 6: { constexpr int Q = 6; CALL; } break; \
    case 8: { constexpr int Q = 8; CALL; } break; \
    default: return hipErrorInvalidValue;        \
  }

// record set of the first record of every k_rank_records workgroup: largest s with set_offsets[s] <= block * kBlock
__global__ __launch_bounds__(kBlock) void k_block_first(const uint32_t* __restrict__ set_offsets, uint32_t n_sets, uint32_t n_blocks,
                                                        uint32_t* __restrict__ block_first) {
  const uint32_t blk = blockIdx.x * kBlock + threadIdx.x;
  if (blk >= n_blocks) return;
  const uint64_t r0 = (uint64_t)blk * kBlock;
  uint32_t lo = 0, hi = n_sets;  // set_offsets[lo] <= r0 (set_offsets[0] = 0); set_offsets[hi] > r0 or hi == n_sets
  while (hi - lo > 1) {
    const uint32_t mid = lo + ((hi - lo) >> 1);
    if (set_offsets[mid] <= r0) lo = mid; else hi = mid;
  }
  block_first[blk] = lo;
}

// ------------------------------------------------------------------------------------------------------------
// Rank + gather for LARGE record sets (round 5): a workgroup per set of w.big_min ... kBigSetMax records.  k_rank_records gives every
// record a trip over its whole set (its rank) and over the records in front of it (its path labels): quadratic per set, and once a
// set no longer fits that kernel's window of 384 staged records every step of those trips is a load from memory -- 11.6 ms instead of
// 1.2 per config-2-sized step when the 20 M records come in sets of 400 (profiles/r05_dna_hits.txt; BLAST's default of 500 targets
// per query makes such sets ordinary).  Here the set's columns are staged once (score, identities, query range, preorder number of
// the taxon: 20 bytes per record), the set's totals come from a workgroup reduction, and the two trips read LDS.  Same outputs,
// bit for bit: rows at their SortFilter rank, segment descriptors, per-row words, set header (see k_rank_records for the semantics).
// The list of such sets comes from k_list_big_sets (a compaction over the set offsets in front of the batch's first host read).
// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_list_big_sets(const uint32_t* __restrict__ set_offsets, uint32_t n_sets, uint32_t big_min,
                                                          uint32_t* __restrict__ list, uint32_t* __restrict__ count) {
  // (the list keeps the sets in order inside a workgroup's 256 -- one reservation per workgroup -- so that the workgroups of
  // k_rank_big_sets that run side by side own NEIGHBOURING sets: their row quads share lines of the tiled layout, four sets to a line)
  __shared__ uint32_t s_wave[kBlock / 64], s_base;
  const uint32_t s = blockIdx.x * kBlock + threadIdx.x;
  uint32_t n = 0;
  if (s < n_sets) n = set_offsets[s + 1] - set_offsets[s];
  const bool big = s < n_sets && n >= big_min && n <= kBigSetMax;  // (k_rank_records skips a record by exactly this test)
  const unsigned long long m = __ballot(big);
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  if (lane == 0) s_wave[wave] = (uint32_t)__popcll(m);
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t total = 0;
    for (int k = 0; k < kBlock / 64; ++k) { const uint32_t c = s_wave[k]; s_wave[k] = total; total += c; }
    s_base = total ? atomicAdd(count, total) : 0u;
  }
  __syncthreads();
  if (big) list[s_base + s_wave[wave] + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = s;
}

// (BLOCK / NMAX: 256 threads for sets of up to 1024 records.  A wavefront per set of up to 256 records was measured for the sets of 64 ... 190
// records: parity-green and 2 ms per step SLOWER than k_rank_records there -- a set's rows lie 64 slots apart in the tiled layout, so a
// kernel that owns one set stores every quad into a line of its own, where k_rank_records writes 64 neighbouring sets' rows together)
template <int Q, int BLOCK, uint32_t NMAX>
__global__ __launch_bounds__(BLOCK) void k_rank_big_sets(BatchDev b, WorkDev w, TaxDev tax, StoreDev qstore, StoreDev dbstore, Params p,
                                                          const uint32_t* __restrict__ big_list) {
  constexpr uint32_t kPer = NMAX / BLOCK;  // records of a set per thread, at most
  constexpr uint32_t kTab = 2u * NMAX, kEmpty = 0xFFFFFFFFu;
  __shared__ uint32_t s_score[NMAX];  // score bits; a masked record: a NaN
  __shared__ uint32_t s_ident[NMAX];
  __shared__ uint32_t s_qs[NMAX], s_qe[NMAX];  // query range of the record; masked: (0xFFFFFFFF, 0)
  __shared__ uint16_t s_ord[NMAX];    // the sort's permutation: s_ord[p] = record at rank p
  __shared__ uint16_t s_rank[NMAX];
  __shared__ uint8_t s_filt[NMAX];
  __shared__ uint32_t s_hkey[kTab], s_hval[kTab];  // per taxonomy level: ancestor at that level -> first record of the set under it
  __shared__ uint32_t s_red[4];  // masked records, first unmasked record, qrstart, qrstop
  __shared__ float s_smax;
  const uint32_t tid = threadIdx.x;
  const uint32_t s = big_list[blockIdx.x];
  const uint32_t beg = b.set_offsets[s], end = b.set_offsets[s + 1], cnt = end - beg;
  uint32_t n_pad = 2u;
  while (n_pad < cnt) n_pad <<= 1;
  if (tid == 0) { s_red[0] = 0u; s_red[1] = 0xFFFFFFFFu; s_red[2] = 0xFFFFFFFFu; s_red[3] = 0u; s_smax = -INFINITY; }
  __syncthreads();
  {
    uint32_t n_filt = 0, first_act = 0xFFFFFFFFu, qlo = 0xFFFFFFFFu, qhi = 0u;
    float smax_nn = -INFINITY;
    for (uint32_t i = tid; i < n_pad; i += BLOCK) {
      s_ord[i] = (uint16_t)i;
      if (i >= cnt) continue;
      const uint32_t r = beg + i;
      const bool filt = rec_filtered(b, r);
      const uint32_t sb = filt ? 0x7FC00000u : __float_as_uint(b.score[r]);
      const uint32_t a = filt ? 0xFFFFFFFFu : b.qstart[r], e = filt ? 0u : b.qstop[r];
      s_score[i] = sb; s_ident[i] = b.identities[r]; s_qs[i] = a; s_qe[i] = e; s_filt[i] = filt ? 1 : 0;
      n_filt += filt ? 1u : 0u;
      first_act = min(first_act, filt ? 0xFFFFFFFFu : i);
      qlo = min(qlo, a); qhi = max(qhi, e);
      smax_nn = __builtin_fmaxf(smax_nn, __uint_as_float(sb));
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
      n_filt += (uint32_t)__shfl_xor(n_filt, d, 64);
      first_act = min(first_act, (uint32_t)__shfl_xor(first_act, d, 64));
      qlo = min(qlo, (uint32_t)__shfl_xor(qlo, d, 64));
      qhi = max(qhi, (uint32_t)__shfl_xor(qhi, d, 64));
      smax_nn = __builtin_fmaxf(smax_nn, __shfl_xor(smax_nn, d, 64));
    }
    if ((tid & 63u) == 0u) {
      atomicAdd(&s_red[0], n_filt); atomicMin(&s_red[1], first_act); atomicMin(&s_red[2], qlo); atomicMax(&s_red[3], qhi);
      // (the largest score that is not a NaN: a float maximum over four wavefronts by compare-and-swap on its bits)
      uint32_t* sm = reinterpret_cast<uint32_t*>(&s_smax);
      uint32_t old = *sm;
      while (__uint_as_float(old) < smax_nn) { const uint32_t seen = atomicCAS(sm, old, __float_as_uint(smax_nn)); if (seen == old) break; old = seen; }
    }
  }
  __syncthreads();
  const uint32_t n = cnt - s_red[0], qrstart = s_red[2], qrstop = s_red[3];
  float smax = 0.f;  // score of the set's best record: starts from the first unmasked one (a NaN there stays), any other NaN is passed over
  if (n) { const float s_first = __uint_as_float(s_score[s_red[1]]); smax = isnan(s_first) ? s_first : s_smax; }
  const uint32_t L = qrstop - qrstart + 1;
  // ---- SortFilter order by a bitonic sort of the record indices in LDS: score desc, identities desc, input order; masked records
  // and the padding behind everything (n log^2 n compare-exchanges per set where a count per record was n^2: 45 stages for 512)
  auto before = [&](uint32_t x, uint32_t y) {  // does record x come before record y?
    const bool rx = x < cnt && !s_filt[x < cnt ? x : 0u], ry = y < cnt && !s_filt[y < cnt ? y : 0u];
    if (!rx || !ry) return rx ? true : (ry ? false : x < y);
    const float sx = __uint_as_float(s_score[x]), sy = __uint_as_float(s_score[y]);
    if (sx > sy) return true;
    if (!(sx == sy)) return sy > sx ? false : x < y;  // (a NaN among the sorted records is refused below: any total order will do here)
    const uint32_t ix = s_ident[x], iy = s_ident[y];
    return ix > iy || (ix == iy && x < y);
  };
  for (uint32_t k = 2u; k <= n_pad; k <<= 1) {
    for (uint32_t jj = k >> 1; jj > 0u; jj >>= 1) {
      for (uint32_t t = tid; t < (n_pad >> 1); t += BLOCK) {
        const uint32_t lo = ((t & ~(jj - 1u)) << 1) | (t & (jj - 1u)), hi = lo | jj;  // the pair (lo, lo ^ jj) with bit jj of lo clear
        const uint32_t a = s_ord[lo], c = s_ord[hi];
        const bool up = (lo & k) == 0u;  // this stretch is being sorted into "before" order
        if (before(c, a) == up) { s_ord[lo] = (uint16_t)c; s_ord[hi] = (uint16_t)a; }
      }
      __syncthreads();
    }
  }
  for (uint32_t q = tid; q < n_pad; q += BLOCK) { const uint32_t x = s_ord[q]; if (x < cnt) s_rank[x] = (uint16_t)q; }
  __syncthreads();
  uint4* __restrict__ hits4 = reinterpret_cast<uint4*>(w.hits);
  const SetRows<Q> R = set_rows<Q>(w, s, beg);
  const bool tiled = w.tile_base != nullptr;
  const size_t sq = tiled ? 64 : 1;
  uint32_t my_node[kPer];
  uint32_t* my_row[kPer];  // where the row of my k-th record starts (null: none -- past the end, or masked)
#pragma unroll
  for (uint32_t k = 0; k < kPer; ++k) {
    const uint32_t i = tid + k * BLOCK;
    my_row[k] = nullptr; my_node[k] = 0u;
    if (i >= cnt) continue;
    const uint32_t r = beg + i;
    w.hit_set[r] = s;
    my_node[k] = b.node[r];  // (a masked record's taxon counts for the path labels of the others, as in k_rank_records)
    if (s_filt[i]) continue;
    const float sc = __uint_as_float(s_score[i]);
    const uint32_t id = s_ident[i];
    const uint32_t rank = s_rank[i];
    if (n >= 2 && sc != sc) raise_error(w, kErrNanScore, s);  // (a NaN among the sorted records: refused, as in k_rank_records)
    const uint32_t pos = beg + rank;
    uint2 sp = make_uint2(0u, 0u);
    SegDesc sdesc{0u, 0u, 0u, 0u};
    if (n >= 2) {
      const uint32_t left_ext = s_qs[i] - qrstart, right_ext = qrstop - s_qe[i];  // :523
      const uint32_t start = b.rstart[r], stop = b.rstop[r];
      if (start <= stop) {
        sdesc = plan_segment_desc(dbstore, false, b.ref_seq[r], left_ext < start ? start - left_ext : 1u, stop + right_ext, false, sp);
      } else if (p.alphabet != 0) {
        sdesc = SegDesc{0u, 0u, 0u, (uint32_t)kErrAssert << 4};
        sp = make_uint2(0u, (uint32_t)kErrAssert);
      } else {
        sdesc = plan_segment_desc(dbstore, false, b.ref_seq[r], right_ext < stop ? stop - right_ext : 1u, start + left_ext, true, sp);
      }
    }
    const uint32_t node = my_node[k];
    const bool perfect = b.alnlen[r] == L && id == L;
    const uint32_t meta = (perfect ? kMetaPerfect : 0u) | (sp.y << kMetaErrShift) | ((uint32_t)tax.depth[node] << kMetaDepthShift);
    float qd = FLT_MAX, qs = (float)id;
    if (n == 1) { qd = __uint_as_float(s_qs[i]); qs = __uint_as_float(s_qe[i]); }
    const uint32_t slot = (uint32_t)((R.res + (size_t)rank * R.sr) - w.pair_res);
    w.seg[pos] = sdesc;
    if (w.res_slot) w.res_slot[pos] = slot;
    if (sp.x > kLiteLenMask) { raise_error(w, kErrUnsupported, s); sp.x = 0; }
    w.lite[pos] = sp.x | ((n >= 2 && !perfect && sc >= p.reeval_factor * smax) ? kLiteWant0 : 0u) | (perfect ? kLitePerfect : 0u);
    // the row with every label word "none"; the labels follow level by level
    const size_t base = tiled ? (size_t)(slot & ~63u) * Q + (slot & 63u) : (size_t)slot * Q;
    hits4[base] = make_uint4(__float_as_uint(sc), id, node, meta);
    hits4[base + sq] = make_uint4(__float_as_uint(qd), __float_as_uint(qs), 0xFFFFFFFFu, 0xFFFFFFFFu);
#pragma unroll
    for (int qd4 = 2; qd4 < Q; ++qd4) hits4[base + (size_t)qd4 * sq] = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu);
    my_row[k] = reinterpret_cast<uint32_t*>(hits4 + base);
  }
  if (tid == 0) {  // the set header (:415): the query range's segment
    uint2 qp = make_uint2(0u, 0u);
    SegDesc qdesc{0u, 0u, 0u, 0u};
    if (n >= 2) qdesc = plan_segment_desc(qstore, true, b.query_seq[s], qrstart, qrstop, false, qp);
    w.seg[b.n_records + s] = qdesc;
    if (qp.x > kLiteLenMask) { raise_error(w, kErrUnsupported, s); qp.x = 0; }
    w.set_info[s] = make_uint4(n, qrstart, qrstop, qp.x | (qp.y << 29));
  }
  if (n < 2) return;  // (uniform: no alignments, no labels)
  // ---- taxon path labels: label_d(record) = the first record of the set (input order) whose taxon lies under the record's ancestor
  // at level d.  k_rank_records asks every record in front of me, level by level; here one hash table per level (ancestor ->
  // smallest record index, LDS atomics) answers all records of the set at once: n x D operations instead of n^2 x D / 2.
  const uint32_t D = tax.max_depth, Dpad = (D + 7u) & ~7u;
  const uint32_t tab = 2u * n_pad, tab_shift = 32u - (uint32_t)__ffs((int)tab) + 1u;  // a table of twice the (padded) set: 8 ... kTab slots, cleared per level
  auto slot_of = [&](uint32_t key) { return (key * 2654435761u) >> tab_shift; };
  for (uint32_t c = 0; c < D; c += 8u) {
    uint32_t lab[kPer][8];
    for (uint32_t d = c; d < c + 8u && d < D; ++d) {
      for (uint32_t t = tid; t < tab; t += BLOCK) { s_hkey[t] = kEmpty; s_hval[t] = 0xFFFFFFFFu; }
      __syncthreads();
      uint32_t key[kPer];
#pragma unroll
      for (uint32_t k = 0; k < kPer; ++k) {
        const uint32_t i = tid + k * BLOCK;
        key[k] = kEmpty;
        if (i >= cnt) continue;
        key[k] = tax.anc_iv[(size_t)my_node[k] * Dpad + d].x;  // first preorder number of my ancestor at this level; 0xFFFFFFFF: none (I am above it)
        if (key[k] == kEmpty) continue;
        uint32_t h = slot_of(key[k]);
        for (;;) {
          const uint32_t seen = atomicCAS(&s_hkey[h], kEmpty, key[k]);
          if (seen == kEmpty || seen == key[k]) { atomicMin(&s_hval[h], i); break; }
          h = (h + 1u) & (tab - 1u);
        }
      }
      __syncthreads();
#pragma unroll
      for (uint32_t k = 0; k < kPer; ++k) {
        const uint32_t i = tid + k * BLOCK;
        uint32_t v = i;  // levels below my own keep my own index (nobody shares them)
        if (i < cnt && key[k] != kEmpty) {
          uint32_t h = slot_of(key[k]);
          while (s_hkey[h] != key[k]) h = (h + 1u) & (tab - 1u);
          v = s_hval[h];
        }
#pragma unroll
        for (uint32_t dd = 0; dd < 8u; ++dd) if (dd == d - c) lab[k][dd] = v;
      }
      __syncthreads();
    }
#pragma unroll
    for (uint32_t k = 0; k < kPer; ++k) {
      if (!my_row[k]) continue;
      const uint32_t i = tid + k * BLOCK;
#pragma unroll
      for (uint32_t dd = 0; dd < 8u; ++dd) if (c + dd >= D) lab[k][dd] = i;  // (levels past the deepest: never compared; what k_rank_records leaves there)
      auto put = [&](uint32_t word, uint32_t value) { my_row[k][(size_t)(word >> 2) * sq * 4u + (word & 3u)] = value; };  // word of the row -> its quad
      if (w.lb_shift == 3u) {
        put(6u + c / 4u, (lab[k][0] & 255u) | ((lab[k][1] & 255u) << 8) | ((lab[k][2] & 255u) << 16) | (lab[k][3] << 24));
        put(6u + c / 4u + 1u, (lab[k][4] & 255u) | ((lab[k][5] & 255u) << 8) | ((lab[k][6] & 255u) << 16) | (lab[k][7] << 24));
      } else if (w.lb_shift == 4u) {
#pragma unroll
        for (uint32_t q4 = 0; q4 < 4u; ++q4) put(6u + c / 2u + q4, (lab[k][2 * q4] & 0xFFFFu) | (lab[k][2 * q4 + 1] << 16));
      } else {
#pragma unroll
        for (uint32_t q8 = 0; q8 < 8u; ++q8) put(6u + c + q8, lab[k][q8]);
      }
    }
  }
}

hipError_t launch_list_big_sets(hipStream_t st, const uint32_t* set_offsets, uint32_t n_sets, uint32_t big_min, uint32_t* list, uint32_t* count) {
  if (n_sets == 0) return hipSuccess;
  hipLaunchKernelGGL(k_list_big_sets, dim3((n_sets + kBlock - 1) / kBlock), dim3(kBlock), 0, st, set_offsets, n_sets, big_min, list, count);
  return hipGetLastError();
}

hipError_t launch_rank_big_sets(hipStream_t st, const BatchDev& b, const WorkDev& w, const TaxDev& tax, const StoreDev& q, const StoreDev& db,
                                const Params& p, const uint32_t* big_list, uint32_t n_big, uint32_t largest_set) {
  if (n_big == 0) return hipSuccess;
  // (a batch without a set of more than 512 records: half the LDS per workgroup, twice the workgroups per CU -- the kernel's stages are
  // barriers apart and wait on LDS: 3.1 -> 2.7 ms per step of 400-hit sets)
  if (largest_set <= kBigSetMax / 2) { RPA_DISPATCH_Q(w.row_quads, hipLaunchKernelGGL((k_rank_big_sets<Q, kBlock, kBigSetMax / 2>), dim3(n_big), dim3(kBlock), 0, st, b, w, tax, q, db, p, big_list)); }
  else { RPA_DISPATCH_Q(w.row_quads, hipLaunchKernelGGL((k_rank_big_sets<Q, kBlock, kBigSetMax>), dim3(n_big), dim3(kBlock), 0, st, b, w, tax, q, db, p, big_list)); }
  return hipGetLastError();
}

hipError_t launch_rank_records(hipStream_t st, const BatchDev& b, const WorkDev& w, const TaxDev& tax, const StoreDev& q,
                               const StoreDev& db, const Params& p) {
  if (b.n_records == 0) return hipSuccess;
  const dim3 grid(((b.n_records + kBlock - 1) / kBlock + 7u) / 8u * 8u);  // multiple of 8: see the XCD mapping in the kernel
  hipLaunchKernelGGL(k_block_first, dim3((grid.x + kBlock - 1) / kBlock), dim3(kBlock), 0, st, b.set_offsets, b.n_sets, grid.x, w.block_first);
  RPA_DISPATCH_Q(w.row_quads, hipLaunchKernelGGL(k_rank_records<Q>, grid, dim3(kBlock), 0, st, b, w, tax, q, db, p));
  return hipGetLastError();
}

hipError_t launch_plan_sets(hipStream_t st, const BatchDev& b, const WorkDev& w, const TaxDev& tax, const Params& p,
                            void* out_rows, void* out_stats) {
  if (b.n_sets == 0) return hipSuccess;
  const dim3 grid((b.n_sets + kBlock - 1) / kBlock);
  RPA_DISPATCH_Q(w.row_quads, hipLaunchKernelGGL(k_plan_sets<Q>, grid, dim3(kBlock), 0, st, b, w, tax, p, (OutRow*)out_rows, (OutStats*)out_stats));
  return hipGetLastError();
}

hipError_t launch_emit(hipStream_t st, const BatchDev& b, const WorkDev& w, const TaxDev& tax, const Params& p) {
  if (w.live_in) {  // sparse round: one wave per live set
    if (w.n_live_in == 0) return hipSuccess;
    const dim3 grid(((size_t)w.n_live_in * 64 + kSparseBlock - 1) / kSparseBlock);
    RPA_DISPATCH_Q(w.row_quads, hipLaunchKernelGGL(k_emit_sparse<Q>, grid, dim3(kSparseBlock), 0, st, b, w, p));
    return hipGetLastError();
  }
  const uint32_t total = b.n_records + b.n_sets;
  if (total == 0) return hipSuccess;
  const dim3 grid((total + kEmitSpan - 1) / kEmitSpan);
  RPA_DISPATCH_Q(w.row_quads, hipLaunchKernelGGL(k_emit<Q>, grid, dim3(kEmitBlock), p.defer2 ? (size_t)kEmitSpan : 0, st, b, w, p));
  return hipGetLastError();
}

hipError_t launch_defer_filter(hipStream_t st, const BatchDev& b, const WorkDev& w, const Params& p) {
  if (b.n_sets == 0) return hipSuccess;
  const dim3 grid((b.n_sets + kFilterBlock - 1) / kFilterBlock);
  RPA_DISPATCH_Q(w.row_quads, hipLaunchKernelGGL(k_defer_filter<Q>, grid, dim3(kFilterBlock), 0, st, b, w, p));
  return hipGetLastError();
}

hipError_t launch_decide(hipStream_t st, const BatchDev& b, const WorkDev& w, const TaxDev& tax, const Params& p,
                         void* out_rows, void* out_stats, unsigned long long* prof_acc) {
  const uint32_t n_threads = w.live_in ? w.n_live_in : b.n_sets;
  if (n_threads == 0) return hipSuccess;
  const dim3 grid((n_threads + kBlock - 1) / kBlock);
  if (w.trace) { RPA_DISPATCH_Q(w.row_quads, hipLaunchKernelGGL((k_decide<Q, true>), grid, dim3(kBlock), (size_t)w.band_ranks * kBlock * 2 * sizeof(float), st, b, w, tax, p, (OutRow*)out_rows, (OutStats*)out_stats, prof_acc)); }
  else { RPA_DISPATCH_Q(w.row_quads, hipLaunchKernelGGL((k_decide<Q, false>), grid, dim3(kBlock), (size_t)w.band_ranks * kBlock * 2 * sizeof(float), st, b, w, tax, p, (OutRow*)out_rows, (OutStats*)out_stats, prof_acc)); }
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------------------
// Pair-level entry (rpa_align_pairs): every sequence of an arena-shaped store is one segment; explicit pair list.
// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_plan_plain_segments(WorkDev w, StoreDev store) {
  const uint32_t seg = blockIdx.x * kBlock + threadIdx.x;
  if (seg >= w.T) return;
  const ulonglong2 ex = store.extent[seg];
  w.seg[seg] = SegDesc{(uint32_t)ex.x, (uint32_t)(ex.x >> 32), (uint32_t)ex.y, 0u};
}

hipError_t launch_plan_plain_segments(hipStream_t st, int alphabet, const WorkDev& w, const StoreDev& store) {
  if (w.T == 0) return hipSuccess;
  (void)alphabet;
  hipLaunchKernelGGL(k_plan_plain_segments, dim3((w.T + kBlock - 1) / kBlock), dim3(kBlock), 0, st, w, store);
  return hipGetLastError();
}

__global__ __launch_bounds__(kBlock) void k_bin_pairs(WorkDev w, uint32_t n_pairs, Params p) {
  const uint32_t t = blockIdx.x * kBlock + threadIdx.x;
  const bool live = t < n_pairs;
  int bin = -1;
  uint32_t maxlen = 0;
  if (live) {
    const uint32_t sa = w.pair_a[t], sb = w.pair_b[t];
    if (sa >= w.T || sb >= w.T) raise_error(w, kErrSeqId, t);
    else {
      const uint32_t la = w.seg[sa].len, lb = w.seg[sb].len;
      bin = pair_bin(p, la, lb);
      maxlen = p.alphabet == 0 ? max(la, lb) : la;
      if (bin >= kNumBins) { raise_error(w, kErrUnsupported, t); bin = -1; }
      else if (la == 0 || lb == 0) { raise_error(w, kErrRange, t); bin = -1; }
    }
  }
  wave_max_to(&w.counters[kCtrMaxLen], maxlen);
#pragma unroll
  for (int k = 0; k < kNumBins; ++k) {
    const uint32_t c = (bin == k) ? 1u : 0u;
    if (!__any(c != 0)) continue;
    const uint32_t base = wave_reserve(&w.counters[k], c);
    if (c) w.bin_list[(size_t)k * w.T2 + base] = make_uint2(t, 0u);
  }
}

hipError_t launch_bin_pairs(hipStream_t st, const Params& params, const WorkDev& w, uint32_t n_pairs) {
  if (n_pairs == 0) return hipSuccess;
  Params p = params;
  // rpa_align_pairs reports matches / mismatches, which the score-and-length cell does not count: K3 for pairs beyond the packed
  // cell (RPA_PAIRS_WIDE=1, tests: the wide cell here too -- its pairs then report 0xFFFFFFFF matches / mismatches)
  const char* pw = getenv("RPA_PAIRS_WIDE");
  const bool pairs_wide = pw && atoi(pw) != 0;
  p.protein_wide = pairs_wide ? params.protein_wide : 0;
  hipLaunchKernelGGL(k_bin_pairs, dim3((n_pairs + kBlock - 1) / kBlock), dim3(kBlock), 0, st, w, n_pairs, p);
  return hipGetLastError();
}

struct OutAlignment {  // = rpa_alignment
  uint32_t matches, mismatches, gaps, length;
  float distance, similarity;
};

__global__ __launch_bounds__(kBlock) void k_export_alignments(WorkDev w, uint32_t n_pairs, OutAlignment* __restrict__ out) {
  const uint32_t t = blockIdx.x * kBlock + threadIdx.x;
  if (t >= n_pairs) return;
  const uint4 e = w.pair_ext[t];
  const float2 r = w.pair_res[t];
  out[t] = OutAlignment{e.x, e.y, e.z, e.w, r.x, r.y};
}

hipError_t launch_export_alignments(hipStream_t st, const WorkDev& w, uint32_t n_pairs, void* out) {
  if (n_pairs == 0) return hipSuccess;
  hipLaunchKernelGGL(k_export_alignments, dim3((n_pairs + kBlock - 1) / kBlock), dim3(kBlock), 0, st, w, n_pairs, (OutAlignment*)out);
  return hipGetLastError();
}

}  // namespace rpa
