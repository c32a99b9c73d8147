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
// PATH (its ancestors at depth 1..P from the flattened parent-pointer taxonomy, copied once per hit from a
// per-node table).  Every node predict() ever forms -- rtax, lnode, unode, lca_allnodes and their global versions -- is
// an LCA that includes the best reference's taxon, i.e. a point on the best hit's root path, so the state machine
// keeps DEPTHS on that path instead of nodes, and every getLCA / isParentOf of the reference becomes
//     common(x, i) = length of the common prefix of two paths,   LCA(point d on x's path, node_i) = min(d, common(x, i)),
// pure register arithmetic with no dependent walks over the taxonomy arrays.  Alignment results come back in a per-hit
// array parallel to the rows, so a round reads ~72 contiguous bytes per hit.
//
// Float discipline (bit-exact decisions): f32 distances/similarities/scores, f64 percent identities, int truncations
// exactly where the reference has them (:592,602,659,661); compiled with -ffp-contract=off and IEEE divide/sqrt.
#include <float.h>
#include <limits.h>

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

__device__ __forceinline__ void raise_error(const WorkDev& w, uint32_t code, uint32_t set) {
  if (atomicCAS(&w.counters[kCtrErr], 0u, code) == 0u) w.counters[kCtrErrSet] = set;
}

// ---- hit rows -------------------------------------------------------------------------------------------------
template <int P4>
__device__ __forceinline__ const uint4* hit_ptr(const WorkDev& w, uint32_t pos) {
  return reinterpret_cast<const uint4*>(w.hits) + (size_t)pos * (2 + P4);
}
__device__ __forceinline__ HitCore load_core(const uint4* p) {
  const uint4 a = p[0], b = p[1];
  HitCore c;
  c.score = __uint_as_float(a.x); c.ident = a.y; c.node = a.z; c.rec = a.w;
  c.qd = __uint_as_float(b.x); c.qs = __uint_as_float(b.y); c.seglen = b.z; c.meta = b.w;
  return c;
}
template <int P4>
__device__ __forceinline__ void load_path(const uint4* p, uint32_t (&path)[4 * P4]) {
#pragma unroll
  for (int k = 0; k < P4; ++k) {
    const uint4 v = p[2 + k];
    path[4 * k] = v.x; path[4 * k + 1] = v.y; path[4 * k + 2] = v.z; path[4 * k + 3] = v.w;
  }
}
// mutable fields of a row (qd, qs, meta) live in its second 16-byte half
__device__ __forceinline__ void store_dyn(const WorkDev& w, int stride4, uint32_t pos, float qd, float qs, uint32_t seglen, uint32_t meta) {
  reinterpret_cast<uint4*>(w.hits)[(size_t)pos * stride4 + 1] = make_uint4(__float_as_uint(qd), __float_as_uint(qs), seglen, meta);
}
__device__ __forceinline__ uint32_t meta_depth(uint32_t meta) { return (meta >> kMetaDepthShift) & 0xFFu; }
__device__ __forceinline__ uint32_t meta_err(uint32_t meta) { return (meta >> kMetaErrShift) & 0xFu; }

// length of the common prefix of two root paths = depth of the LCA of the two nodes
template <int P4>
__device__ __forceinline__ uint32_t common(const uint32_t (&a)[4 * P4], uint32_t da, const uint32_t (&b)[4 * P4], uint32_t db) {
  uint32_t c = 0;
  bool run = true;
#pragma unroll
  for (int d = 0; d < 4 * P4; ++d) {
    run = run && (a[d] == b[d]);
    c += run ? 1u : 0u;
  }
  return min(c, min(da, db));
}
template <int P4>
__device__ __forceinline__ uint32_t node_at(const uint32_t (&path)[4 * P4], uint32_t depth) {  // point `depth` on a path
  uint32_t v = kRoot;
#pragma unroll
  for (int d = 0; d < 4 * P4; ++d)
    if ((uint32_t)d + 1 == depth) v = path[d];
  return v;
}

// ------------------------------------------------------------------------------------------------------------
// Segment planning (a7).  store.getSequence(id, start, stop): stop clipped to the sequence length, start beyond it is
// a range error (sequencestorage.hh:341-362); getSequenceDNA (:870-880): forward [rstart-left_ext, rstop+right_ext]
// with the start floored at 1; reverse (rstart > rstop) [rstop-right_ext, rstart+left_ext] reverse-complemented.
// Protein with reverse coordinates trips a live assert in the reference (:864).  Returns (length, error code).
// ------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint2 plan_segment(const WorkDev& w, uint32_t seg, const StoreDev& store, bool from_query,
                                              uint32_t seq, uint32_t start, uint32_t stop, bool rc) {
  uint32_t err = 0, len = 0;
  uint64_t src = 0;
  if (seq >= store.n_seqs) err = kErrSeqId;
  else {
    const uint64_t off = store.offsets[seq];
    const uint64_t slen = store.offsets[seq + 1] - off;
    if (start > stop) err = kErrAssert;
    else {
      if ((uint64_t)stop > slen) stop = (uint32_t)slen;
      if (start < 1 || start > stop) err = kErrRange;
      else { len = stop - start + 1; src = off + (start - 1); }
    }
  }
  w.seg[seg] = SegDesc{(uint32_t)src, (uint32_t)(src >> 32), len, (rc ? 1u : 0u) | (from_query ? 2u : 0u) | (err << 4)};
  return make_uint2(len, err);
}

// ------------------------------------------------------------------------------------------------------------
// Rank + gather: one thread per record.  It scans the records of its own set once (neighbouring lanes read the same
// few cache lines) to get (a) its position in SortFilter order (score desc, identities desc, input order on full ties
// = what a stable list::sort gives) among the unmasked records, and (b) the set's n / query range (:391-404).  It
// plans its own reference segment, then writes its row -- core fields plus the taxon path copied from the per-node
// ancestor table -- at the sorted position.  The thread of a set's first record also plans the query-range segment
// and writes the set header.
// ------------------------------------------------------------------------------------------------------------
template <int P4>
__global__ __launch_bounds__(kBlock) void k_rank_records(BatchDev b, WorkDev w, TaxDev tax, StoreDev qstore, StoreDev dbstore, Params p) {
  const uint32_t r = blockIdx.x * kBlock + threadIdx.x;
  if (r >= b.n_records) return;
  uint32_t lo = 0, hi = b.n_sets;  // largest s with set_offsets[s] <= r  (empty sets share an offset: take the last one)
  while (hi - lo > 1) {
    const uint32_t mid = (lo + hi) >> 1;
    if (b.set_offsets[mid] <= r) lo = mid; else hi = mid;
  }
  const uint32_t s = lo;
  w.hit_set[r] = s;  // rows and records of a set share the index range [set_off[s], set_off[s+1])
  const uint32_t beg = b.set_offsets[s], end = b.set_offsets[s + 1];
  const bool me_filtered = rec_filtered(b, r);
  const float sc = b.score[r];
  const uint32_t id = b.identities[r];
  uint32_t rank = 0, n = 0, qrstart = 0xFFFFFFFFu, qrstop = 0;
  for (uint32_t j = beg; j < end; ++j) {
    if (rec_filtered(b, j)) continue;
    ++n;
    qrstart = min(qrstart, b.qstart[j]);
    qrstop = max(qrstop, b.qstop[j]);
    const float sj = b.score[j];
    const uint32_t ij = b.identities[j];
    const bool before = (sj > sc) || (sj == sc && (ij > id || (ij == id && j < r)));
    rank += before ? 1u : 0u;
  }
  if (r == beg) {
    uint2 qp = make_uint2(0u, 0u);
    if (n >= 2) qp = plan_segment(w, b.n_records + s, qstore, true, b.query_seq[s], qrstart, qrstop, false);  // :415
    w.set_info[s] = make_uint4(n, qrstart, qrstop, 0u);
    w.set_qseg[s] = qp;
  }
  if (me_filtered) return;
  uint2 sp = make_uint2(0u, 0u);
  if (n >= 2) {
    const uint32_t left_ext = b.qstart[r] - qrstart, right_ext = qrstop - b.qstop[r];  // :523
    const uint32_t start = b.rstart[r], stop = b.rstop[r];
    if (start <= stop) {
      sp = plan_segment(w, r, dbstore, false, b.ref_seq[r], left_ext < start ? start - left_ext : 1u, stop + right_ext, false);
    } else if (p.alphabet != 0) {
      w.seg[r] = SegDesc{0u, 0u, 0u, (uint32_t)kErrAssert << 4};
      sp = make_uint2(0u, (uint32_t)kErrAssert);
    } else {
      sp = plan_segment(w, r, dbstore, false, b.ref_seq[r], right_ext < stop ? stop - right_ext : 1u, start + left_ext, true);
    }
  }
  const uint32_t node = b.node[r];
  const uint32_t L = qrstop - qrstart + 1;
  const bool perfect = b.alnlen[r] == L && id == L;  // :431,511
  const uint32_t meta = (perfect ? kMetaPerfect : 0u) | (sp.y << kMetaErrShift) | ((uint32_t)tax.depth[node] << kMetaDepthShift);
  float qd = FLT_MAX, qs = 0.f;  // :475-476
  if (n == 1) { qd = __uint_as_float(b.qstart[r]); qs = __uint_as_float(b.qstop[r]); }  // n==1 needs them (:373-374)
  uint4* row = reinterpret_cast<uint4*>(w.hits) + (size_t)(beg + rank) * (2 + P4);
  row[0] = make_uint4(__float_as_uint(sc), id, node, r);
  row[1] = make_uint4(__float_as_uint(qd), __float_as_uint(qs), sp.x, meta);
  const uint4* anc = reinterpret_cast<const uint4*>(tax.anc) + (size_t)node * P4;
#pragma unroll
  for (int k = 0; k < P4; ++k) row[2 + k] = anc[k];
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
  return protein_bin(lb);  // protein: B (second segment) is the vertical sequence cut into register strips
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
constexpr int kEmitBlock = 1024;  // one list reservation (atomic) per bin per 1024 rows: same-address atomics are ~11 ns each

template <int P4>
__global__ __launch_bounds__(kEmitBlock) void k_emit(BatchDev b, WorkDev w, Params p) {
  __shared__ uint32_t s_cnt[kNumBins][kEmitBlock / 64];  // per bin, per wave: number of active lanes
  __shared__ uint32_t s_base[kNumBins];
  __shared__ uint32_t s_maxlen[kEmitBlock / 64];
  const uint32_t pos = blockIdx.x * kEmitBlock + threadIdx.x;
  const uint32_t H = b.n_records;
  int bin = -1;
  uint32_t maxlen = 0;
  uint2 entry = make_uint2(0u, 0u);
  if (pos < H + b.n_sets) {
    const bool is_query_pair = pos >= H;
    const uint32_t s = is_query_pair ? pos - H : w.hit_set[pos];
    const EmitInfo info = w.emit[s];
    if (info.mode != 0) {
      const uint32_t lo = b.set_offsets[s];
      if (is_query_pair) {
        if (info.mode == 3) {
          const HitCore ha = load_core(hit_ptr<P4>(w, lo + info.anchor));
          if (info.emit_query && ha.qd == FLT_MAX) { bin = pair_bin(p, ha.seglen, info.qseglen); maxlen = p.alphabet == 0 ? max(ha.seglen, info.qseglen) : ha.seglen; entry = make_uint2(ha.rec, pos | kDstIsB); }
        }
      } else {
        const uint32_t i = pos - lo;
        if (i < info.n && i >= info.i_lo && i < info.i_hi) {
          const uint4* pi = hit_ptr<P4>(w, pos);
          const HitCore h = load_core(pi);
          uint32_t other_len = info.qseglen, other_seg = H + s;
          bool want = false;
          if (info.mode == 1) want = !(h.meta & kMetaPerfect) && h.score >= info.fcut;
          else if (i != info.anchor) {
            if (info.mode == 2) want = h.qd != 0.f;
            else if (h.score >= info.fcut && (double)h.qs / info.L >= info.qpid_thresh) {
              uint32_t pb[4 * P4], pth[4 * P4];
              const uint4* pbp = hit_ptr<P4>(w, lo + info.best);
              load_path<P4>(pbp, pb);
              load_path<P4>(pi, pth);
              want = !(common<P4>(pb, meta_depth(load_core(pbp).meta), pth, meta_depth(h.meta)) >= info.d_unode_g);
            }
            if (want) { const HitCore ha = load_core(hit_ptr<P4>(w, lo + info.anchor)); other_len = ha.seglen; other_seg = ha.rec; }
          }
          if (want) {
            bin = pair_bin(p, h.seglen, other_len);
            maxlen = p.alphabet == 0 ? max(h.seglen, other_len) : h.seglen;  // DNA: text = the longer one; protein: A
            entry = make_uint2(h.rec, other_seg);
            if (bin >= kNumBins) { raise_error(w, kErrUnsupported, s); bin = -1; }
          }
        }
      }
    }
  }
  // block-level compaction: wave ballots -> LDS -> one atomic per non-empty bin per block
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned long long my_mask = 0ull;
#pragma unroll
  for (int k = 0; k < kNumBins; ++k) {
    const unsigned long long mask = __ballot(bin == k);
    if (lane == 0) s_cnt[k][wave] = (uint32_t)__popcll(mask);
    if (bin == k) my_mask = mask;
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) maxlen = max(maxlen, (uint32_t)__shfl_xor(maxlen, d, 64));
  if (lane == 0) s_maxlen[wave] = maxlen;
  __syncthreads();
  if (threadIdx.x < kNumBins) {  // thread k: exclusive prefix over the waves of bin k, then reserve
    uint32_t run = 0;
    for (int wv = 0; wv < kEmitBlock / 64; ++wv) { const uint32_t c = s_cnt[threadIdx.x][wv]; s_cnt[threadIdx.x][wv] = run; run += c; }
    s_base[threadIdx.x] = run ? atomicAdd(&w.counters[threadIdx.x], run) : 0u;
  } else if (threadIdx.x == 64) {
    uint32_t m = 0;
    for (int wv = 0; wv < kEmitBlock / 64; ++wv) m = max(m, s_maxlen[wv]);
    if (m) atomicMax(&w.counters[kCtrMaxLen], m);
  }
  __syncthreads();
  if (bin >= 0)
    w.bin_list[(size_t)bin * w.T2 + s_base[bin] + s_cnt[bin][wave] + (uint32_t)__popcll(my_mask & ((1ull << lane) - 1ull))] = entry;
}

// Sparse rounds: when only a small fraction of the record sets is still running, scanning every row of the batch
// would cost more than the work left.  One wavefront per live set; lane j looks at hits j, j+64, ...; lane 0 also
// handles the anchor <-> query alignment.  Same predicate and same list format as k_emit.
template <int P4>
__global__ __launch_bounds__(kBlock) void k_emit_sparse(BatchDev b, WorkDev w, Params p) {
  const uint32_t wave_id = (blockIdx.x * kBlock + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  if (wave_id >= w.counters[kCtrActive]) return;  // exact list length (the grid is sized by an upper bound); whole wave
  const uint32_t s = w.live_in[wave_id];
  const uint32_t H = b.n_records;
  const EmitInfo info = w.emit[s];
  if (info.mode == 0) return;
  const uint32_t lo = b.set_offsets[s];
  uint32_t anchor_len = 0, anchor_rec = 0;
  float anchor_qd = 0.f;
  if (info.mode != 1) { const HitCore ha = load_core(hit_ptr<P4>(w, lo + info.anchor)); anchor_len = ha.seglen; anchor_rec = ha.rec; anchor_qd = ha.qd; }
  uint32_t pb[4 * P4];
  uint32_t db = 0;
  if (info.mode == 3) { const uint4* pbp = hit_ptr<P4>(w, lo + info.best); load_path<P4>(pbp, pb); db = meta_depth(load_core(pbp).meta); }
  uint32_t maxlen = 0;
  const uint32_t i_hi = min(info.i_hi, info.n);
  // the query pair rides along as a virtual hit index i_hi handled by lane 0 in the first trip
  for (uint32_t i0 = info.i_lo; i0 < i_hi || i0 == info.i_lo; i0 += 64) {
    const uint32_t i = i0 + (uint32_t)lane;
    int bin = -1;
    uint2 entry = make_uint2(0u, 0u);
    if (i < i_hi) {
      const uint4* pi = hit_ptr<P4>(w, lo + i);
      const HitCore h = load_core(pi);
      bool want = false;
      uint32_t other_len = info.qseglen, other_seg = H + s;
      if (info.mode == 1) want = !(h.meta & kMetaPerfect) && h.score >= info.fcut;
      else if (i != info.anchor) {
        if (info.mode == 2) want = h.qd != 0.f;
        else if (h.score >= info.fcut && (double)h.qs / info.L >= info.qpid_thresh) {
          uint32_t pth[4 * P4];
          load_path<P4>(pi, pth);
          want = !(common<P4>(pb, db, pth, meta_depth(h.meta)) >= info.d_unode_g);
        }
        other_len = anchor_len; other_seg = anchor_rec;
      }
      if (want) {
        bin = pair_bin(p, h.seglen, other_len);
        maxlen = max(maxlen, p.alphabet == 0 ? max(h.seglen, other_len) : h.seglen);
        entry = make_uint2(h.rec, other_seg);
        if (bin >= kNumBins) { raise_error(w, kErrUnsupported, s); bin = -1; }
      }
    }
    if (i0 == info.i_lo && lane == 63 && info.mode == 3 && info.emit_query && anchor_qd == FLT_MAX && bin < 0) {
      bin = pair_bin(p, anchor_len, info.qseglen); maxlen = max(maxlen, p.alphabet == 0 ? max(anchor_len, info.qseglen) : anchor_len);
      entry = make_uint2(anchor_rec, (H + s) | kDstIsB);
    }
#pragma unroll
    for (int k = 0; k < kNumBins; ++k) {
      const unsigned long long mask = __ballot(bin == k);
      if (mask == 0ull) continue;
      uint32_t base = 0;
      if (lane == 0) base = atomicAdd(&w.counters[k], (uint32_t)__popcll(mask));
      base = __shfl(base, 0, 64);
      if (bin == k) w.bin_list[(size_t)k * w.T2 + base + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull))] = entry;
    }
    if (i0 + 64 >= i_hi) break;
  }
  wave_max_to(&w.counters[kCtrMaxLen], maxlen);
}

// ------------------------------------------------------------------------------------------------------------
// Plan: one thread per record set.  Handles everything predict() decides before any alignment (:341-472): n == 0,
// n == 1, perfect-best-hit shortcut; otherwise emits pass 0.  (Masking, query range, sorting and segment plans were
// done per record by k_rank_records.)
// ------------------------------------------------------------------------------------------------------------
template <int P4>
__global__ __launch_bounds__(kBlock) void k_plan_sets(BatchDev b, WorkDev w, Params p, OutRow* __restrict__ rows,
                                                       OutStats* __restrict__ stats) {
  const uint32_t s = blockIdx.x * kBlock + threadIdx.x;
  const bool live = s < b.n_sets;
  SetState st;
  st.phase = kDone;
  EmitInfo em{};
  em.mode = 0;
  if (live) {
    const uint32_t lo = b.set_offsets[s], hi = b.set_offsets[s + 1];
    uint4 info = make_uint4(0u, 0u, 0u, 0u);
    uint2 qseg = make_uint2(0u, 0u);
    if (hi > lo) { info = w.set_info[s]; qseg = w.set_qseg[s]; }
    const uint32_t n = info.x, qrstart = info.y, qrstop = info.z;
    st.n = n; st.qrstart = qrstart; st.qrstop = qrstop; st.L = qrstop - qrstart + 1;
    st.qseglen = qseg.x; st.qsegerr = qseg.y;
    st.support = 0; st.best = 0; st.d_lca_all = 0; st.d_rtax = 0; st.d_lnode_g = 0; st.d_unode_g = 0; st.ival_g = 0.f;
    st.bf_max = 1.f; st.lrd_min = 255; st.anchor = 0; st.cnt0 = st.cnt1 = st.cnt2 = 0; st.loop_live = 0; st.chunk_hi = 0; st.i_next = 0;
    OutRow row;
    row.begin = 1; row.end = b.query_len[s];  // initPredictionRecord, core/src/taxonpredictionmodel.hh:42-44
    row.lower = row.upper = row.rtax = kRoot; row.support = 0; row.ival = 0.f; row.flags = 0;
    bool done = true;
    if (n == 0) {  // :359-368
      row.flags = kFlagIvalUntouched;
    } else if (n == 1) {  // :371-388
      const HitCore h = load_core(hit_ptr<P4>(w, lo));
      row.begin = __float_as_uint(h.qd); row.end = __float_as_uint(h.qs); row.ival = 1.f;
      row.lower = h.node; row.upper = kRoot; row.support = h.ident; row.rtax = h.node;
      row.flags = kFlagSingle;
    } else {
      const uint4* p0 = hit_ptr<P4>(w, lo);
      const HitCore h0 = load_core(p0);
      if (h0.meta & kMetaPerfect) {  // :431-472: lnode / unode are points on hit 0's path
        uint32_t path0[4 * P4];
        load_path<P4>(p0, path0);
        const uint32_t d0 = meta_depth(h0.meta);
        const float best = h0.score;
        uint32_t d_l = d0, d_u = 0;
        uint32_t i = 1;
        for (;;) {
          if (i == n) { d_u = 0; break; }  // unode = root
          const uint4* pi = hit_ptr<P4>(w, lo + i);
          HitCore hi2 = load_core(pi);
          uint32_t pth[4 * P4];
          load_path<P4>(pi, pth);
          if (hi2.score == best) d_l = min(d_l, common<P4>(path0, d0, pth, meta_depth(hi2.meta)));
          else {
            const float sc = hi2.score;
            d_u = d_l;
            for (;;) {
              d_u = min(d_u, common<P4>(path0, d0, pth, meta_depth(hi2.meta)));
              if (++i >= n) break;
              pi = hit_ptr<P4>(w, lo + i);
              hi2 = load_core(pi);
              if (!(hi2.score == sc)) break;
              load_path<P4>(pi, pth);
            }
            break;
          }
          ++i;
        }
        row.begin = qrstart; row.end = qrstop; row.ival = 0.f;
        row.lower = node_at<P4>(path0, d_l); row.upper = node_at<P4>(path0, d_u); row.support = st.L; row.rtax = row.lower;
        row.flags = kFlagShortcut;
      } else {
        done = false;
        st.phase = kPass0Wait;
        em.mode = 1; em.n = n; em.L = st.L; em.qseglen = st.qseglen; em.i_lo = 0; em.i_hi = n;
        em.fcut = p.reeval_factor * h0.score;  // dbalignment_searchscore_threshold, :499
      }
    }
    if (done) {
      rows[s] = row;
      if (stats) stats[s] = OutStats{n, 0, 0, 0};
    } else {
      w.live_out[atomicAdd(&w.counters[kCtrActive], 1u)] = s;
    }
    w.emit[s] = em;
    w.state[s] = st;
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
    uint32_t remembered = 0;
    float lowmin = FLT_MAX;
    for (uint32_t r = 0; r < anchor_rank; ++r) {
      const float f = fmin[r * kBlock];
      if (f < lowmin) { remembered |= 1u << r; lowmin = f; }
    }
    float dsuf = dmax[anchor_rank * kBlock];
    for (int r = (int)anchor_rank - 1; r >= 0; --r) {
      const float f = fmin[r * kBlock];
      if ((remembered >> r) & 1u) {
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
  if (dist != dist) { raise_error(w, kErrAssert, s); return false; }
  return true;
}

// ------------------------------------------------------------------------------------------------------------
// Decide: one thread per record set, one call per round.
// ------------------------------------------------------------------------------------------------------------
template <int P4>
__global__ __launch_bounds__(kBlock) void k_decide(BatchDev b, WorkDev w, Params p, OutRow* __restrict__ rows,
                                                    OutStats* __restrict__ stats, unsigned long long* prof_acc) {
  // BandFactor accumulators: (deepest taxonomy level + 1) ranks x 256 threads x {fmin, dmax}; sized at launch so that a
  // pruned NCBI tree (8 levels) needs 18 KB and the kernel can share a CU with the alignment kernel of another stream
  extern __shared__ float s_band[];
  float* s_fmin = s_band;
  float* s_dmax = s_band + (size_t)w.band_ranks * kBlock;
  // dense rounds: one thread per record set; sparse rounds (few sets still running): one thread per entry of the live list
  const uint32_t tid = blockIdx.x * kBlock + threadIdx.x;
  bool live = tid < (w.live_in ? w.n_live_in : b.n_sets);
  const uint32_t s = live ? (w.live_in ? w.live_in[tid] : tid) : 0u;
  SetState st;
  st.phase = kDone;
  if (live) st = w.state[s];
  live = live && st.phase != kDone;
  if (live) {
    EmitInfo em{};
    em.mode = 0; em.n = st.n; em.L = st.L; em.qseglen = st.qseglen;
    const uint32_t lo = b.set_offsets[s];
    const uint32_t n = st.n, L = st.L;
    constexpr int kStride4 = 2 + P4;
    const float X = p.exclude_factor;
    const float2* __restrict__ res = w.pair_res;                       // res[rec]: result of the alignment asked for that hit
    const float2* __restrict__ qres_p = w.pair_res + b.n_records + s;  // result of the anchor <-> query alignment
    bool failed = false;

    if (st.phase == kPass0Wait) {  // ---- pass 0, :497-566 ----
      uint32_t path0[4 * P4];
      const uint4* p0 = hit_ptr<P4>(w, lo);
      const HitCore h0 = load_core(p0);
      load_path<P4>(p0, path0);
      const uint32_t d0 = meta_depth(h0.meta);
      const float thr = p.reeval_factor * h0.score;  // :499
      uint32_t index_best = 0;
      float best_d = 0.f, best_s = 0.f, best_sc = 0.f;
      uint32_t support = 0;
      uint32_t d_all = d0;  // lca_allnodes as a point on hit 0's path (:495)
      for (uint32_t i = 0; i < n; ++i) {
        const uint4* pi = hit_ptr<P4>(w, lo + i);
        const HitCore h = load_core(pi);
        float dist, sim;
        uint32_t flag = 0;
        if (h.meta & kMetaPerfect) {  // :511-517
          flag = kMetaQGroup; dist = 0.f; sim = (float)h.ident;
        } else if (h.score >= thr) {  // :518-534
          flag = kMetaQGroup;
          const float2 r2 = res[h.rec];
          if (!result_ok(w, s, meta_err(h.meta), st.qsegerr, r2.x)) { failed = true; break; }
          dist = r2.x;
          ++st.cnt0;
          const float idf = (float)h.ident;
          sim = r2.y < idf ? idf : r2.y;  // std::max(similarity, identities), :531
        } else {
          dist = FLT_MAX; sim = (float)h.ident;
        }
        store_dyn(w, kStride4, lo + i, dist, sim, h.seglen, (h.meta & ~(kMetaQGroup | kMetaOutGroup)) | flag);
        if (i == 0) { best_d = dist; best_s = sim; best_sc = h.score; }  // index_best starts at 0 (:500)
        else if (dist < best_d) { index_best = i; best_d = dist; best_s = sim; best_sc = h.score; }  // :542-546
        else if (dist == best_d) {
          if (sim > best_s) { index_best = i; best_s = sim; best_sc = h.score; }
          else if (sim == best_s && h.score > best_sc) { index_best = i; best_sc = h.score; }
        }
        support = max(support, (uint32_t)sim);  // :547
        uint32_t pth[4 * P4];
        load_path<P4>(pi, pth);
        d_all = min(d_all, common<P4>(path0, d0, pth, meta_depth(h.meta)));  // :548
      }
      if (!failed) {
        uint32_t pb[4 * P4];
        const uint4* pbp = hit_ptr<P4>(w, lo + index_best);
        load_path<P4>(pbp, pb);
        const uint32_t db = meta_depth(load_core(pbp).meta);
        uint32_t d_rtax = db;  // rtax = node[index_best] (:553), then LCA with every member that stays in qgroup
        bool any = false;
        for (uint32_t i = 0; i < n; ++i) {  // :554-562
          const uint4* pi = hit_ptr<P4>(w, lo + i);
          const HitCore h = load_core(pi);
          if (!(h.meta & kMetaQGroup)) continue;
          if (h.qd != best_d || h.qs != best_s || h.score != best_sc) store_dyn(w, kStride4, lo + i, h.qd, h.qs, h.seglen, h.meta & ~kMetaQGroup);
          else {
            uint32_t pth[4 * P4];
            load_path<P4>(pi, pth);
            d_rtax = min(d_rtax, common<P4>(pb, db, pth, meta_depth(h.meta)));
            any = true;
          }
        }
        if (!any) { raise_error(w, kErrAssert, s); failed = true; }  // assert(!qgroup.empty()), :563
        st.support = support; st.best = index_best; st.d_lca_all = d_all; st.d_rtax = d_rtax;
        st.ival_g = 0.f; st.d_lnode_g = st.d_unode_g = d_rtax; st.bf_max = 1.f; st.lrd_min = 255;  // :569-579
        st.phase = kPass1Select;
      }
    } else if (st.phase == kPass1Wait) {  // ---- pass 1 for one anchor, :580-728 ----
      const uint32_t a = st.anchor;
      uint32_t pa[4 * P4], pb[4 * P4];
      const uint4* pap = hit_ptr<P4>(w, lo + a);
      const HitCore ha = load_core(pap);
      load_path<P4>(pap, pa);
      const uint32_t da = meta_depth(ha.meta);
      const uint4* pbp = hit_ptr<P4>(w, lo + st.best);
      load_path<P4>(pbp, pb);
      const uint32_t db = meta_depth(load_core(pbp).meta);
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
        band.restore(da, w.band_save + (size_t)s * 2 * (kMaxDepth + 1));
        nog = st.nog; d_lnode = st.d_lnode; ldist = st.ldist; udist = st.udist;
        qpid_upper = st.qpid_upper; cut_g = st.cut_g; cut_h = st.cut_h; scut = st.scut; i = st.i_next;
      }
      bool paused = false;
      for (; d_lnode != 0 && i < n; ++i) {  // :604 (lnode != root)
        const uint4* pi = hit_ptr<P4>(w, lo + i);
        const HitCore h = load_core(pi);
        if (!(h.score >= (float)scut)) break;
        const double qsearchpid = (double)h.ident / L;
        const double qpid = (double)h.qs / L;
        const double qpid_thresh = cut_g < cut_h ? cut_h : cut_g;
        if (!(qpid >= qpid_thresh)) continue;  // :612
        float dist;
        if (i == a) dist = 0.f;
        else if (h.qd == 0.f) dist = qdist;  // :620-621
        else {
          if (i >= st.chunk_hi) { paused = true; break; }  // this alignment was not part of the chunk emitted last round
          const float2 r2 = res[h.rec];
          if (!result_ok(w, s, meta_err(h.meta), meta_err(ha.meta), r2.x)) { failed = true; break; }
          dist = r2.x;
          ++st.cnt1;
        }
        uint32_t pth[4 * P4];
        load_path<P4>(pi, pth);
        const uint32_t dh = meta_depth(h.meta);
        band.add(dist, common<P4>(pa, da, pth, dh));  // :641
        if (dist == 0.f) store_dyn(w, kStride4, lo + i, h.qd, h.qs, h.seglen, h.meta & ~kMetaQGroup);  // :644
        else if (dist <= qdist) {  // :646-650
          d_lnode = min(d_lnode, common<P4>(pb, db, pth, dh));
          if (dist > ldist) ldist = dist;
        } else {  // :651-662
          if (dist < udist) {
            udist = dist;
            if (qsearchpid > qpid_upper) {
              qpid_upper = qsearchpid;
              cut_g = qsearchpid * 2. - 1.;
              cut_h = qsearchpid * X;
            }
            if (!scut) scut = (int)(h.score * X);
          }
          ogi[nog] = i; ogd[nog] = (int)dist; ++nog;
        }
      }
      if (paused) {  // ask for the next chunk of this anchor's loop and come back
        st.loop_live = 1; st.i_next = i; st.nog = nog; st.d_lnode = d_lnode; st.ldist = ldist; st.udist = udist;
        st.qpid_upper = qpid_upper; st.cut_g = cut_g; st.cut_h = cut_h; st.scut = scut;
        band.save(w.band_save + (size_t)s * 2 * (kMaxDepth + 1));
        st.chunk_hi = p.spec1 ? min(n, i + p.spec1) : n;
        em.mode = 2; em.anchor = a; em.i_lo = i; em.i_hi = st.chunk_hi;
      } else
      if (!failed) {
        const float bandfactor = band.factor();  // :667
        st.bf_max = st.bf_max < bandfactor ? bandfactor : st.bf_max;
        const float qdist_ex = qdist * bandfactor;  // :669
        float min_upper_dist = (float)INT_MAX;       // :670
        uint32_t keep = 0;
        for (uint32_t k = 0; k < nog; ++k) {  // :673-687
          const float dist = (float)ogd[k];
          bool kept = true;
          if (dist > qdist_ex) {
            if (dist > min_upper_dist) kept = false;
            else if (dist < min_upper_dist) min_upper_dist = dist;
          } else {
            if (min_upper_dist > qdist_ex) min_upper_dist = dist;
            else min_upper_dist = min_upper_dist < dist ? dist : min_upper_dist;
          }
          if (kept) { ogi[keep] = ogi[k]; ogd[keep] = ogd[k]; ++keep; }
        }
        nog = keep;
        uint32_t d_unode = d_lnode;  // unode = lnode, :690 -- (float)INT_MAX != FLT_MAX, so always taken
        for (uint32_t k = 0; k < nog; ++k) {  // :691-711
          const uint32_t i = ogi[k];
          const float dist = (float)ogd[k];
          if (dist > min_upper_dist) continue;
          const uint4* pi = hit_ptr<P4>(w, lo + i);
          const HitCore h = load_core(pi);
          uint32_t pth[4 * P4];
          load_path<P4>(pi, pth);
          const uint32_t cb = common<P4>(pb, db, pth, meta_depth(h.meta));
          d_unode = min(d_unode, cb);                          // unode = LCA(cnode, unode)
          const uint32_t lca_root_dist = min(st.d_rtax, cb);   // depth(LCA(cnode, rtax))
          if (lca_root_dist > st.lrd_min) continue;
          if (lca_root_dist < st.lrd_min) {
            st.lrd_min = lca_root_dist;
            for (uint32_t j = 0; j < n; ++j) {  // outgroup.clear()
              const HitCore hj = load_core(hit_ptr<P4>(w, lo + j));
              if (hj.meta & kMetaOutGroup) store_dyn(w, kStride4, lo + j, hj.qd, hj.qs, hj.seglen, hj.meta & ~kMetaOutGroup);
            }
          }
          const HitCore hn = load_core(pi);  // re-read: the clear above may have touched this row
          store_dyn(w, kStride4, lo + i, hn.qd, hn.qs, hn.seglen, hn.meta | kMetaOutGroup);
        }
        float ival = 0.f;  // :714-719
        if (d_unode != d_lnode && ldist < qdist) ival = (qdist - ldist) / (udist - ldist);
        st.ival_g = st.ival_g < ival ? ival : st.ival_g;   // :724
        st.d_unode_g = min(st.d_unode_g, d_unode);         // :726
        st.d_lnode_g = min(st.d_lnode_g, d_lnode);         // :727
        bool more = false;
        for (uint32_t j = 0; j < n; ++j) more |= (load_core(hit_ptr<P4>(w, lo + j)).meta & kMetaQGroup) != 0;
        st.phase = (more && st.d_lnode_g != 0) ? kPass1Select : kPass2Select;  // :729
        st.loop_live = 0;
      }
    } else if (st.phase == kPass2Wait) {  // ---- pass 2 for one anchor, :750-820 ----
      const uint32_t a = st.anchor;
      const HitCore ha = load_core(hit_ptr<P4>(w, lo + a));
      uint32_t pb[4 * P4];
      const uint4* pbp = hit_ptr<P4>(w, lo + st.best);
      load_path<P4>(pbp, pb);
      const uint32_t db = meta_depth(load_core(pbp).meta);
      float qd_a = ha.qd, qs_a = ha.qs;
      const EmitInfo sel = w.emit[s];            // thresholds were fixed when the anchor was selected (:750-754)
      const double qpid_thresh = sel.qpid_thresh;
      const float scut = sel.fcut;
      bool paused = false;
      uint32_t i = st.loop_live ? st.i_next : 0u;
      for (; i < n; ++i) {  // :757
        const uint4* pi = hit_ptr<P4>(w, lo + i);
        const HitCore h = load_core(pi);
        if (!(h.score >= scut)) break;
        const double qpid = (double)(i == a ? qs_a : h.qs) / L;
        if (!(qpid >= qpid_thresh)) continue;
        uint32_t pth[4 * P4];
        load_path<P4>(pi, pth);
        const uint32_t cb = common<P4>(pb, db, pth, meta_depth(h.meta));
        float dist;
        if (i == a) dist = 0.f;
        else {
          if (cb >= st.d_unode_g) continue;  // isParentOf(unode_global, cnode) || cnode == unode_global, :771
          if (i >= st.chunk_hi) { paused = true; break; }  // not part of the chunk emitted last round
          const float2 r2 = res[h.rec];
          if (!result_ok(w, s, meta_err(h.meta), meta_err(ha.meta), r2.x)) { failed = true; break; }
          dist = r2.x;
          ++st.cnt2;
          store_dyn(w, kStride4, lo + i, dist, h.qs, h.seglen, h.meta);  // querydistance[i] = dist, :785
        }
        if (dist == 0.f) {  // :789
          const HitCore hn = load_core(pi);
          store_dyn(w, kStride4, lo + i, hn.qd, hn.qs, hn.seglen, hn.meta & ~kMetaOutGroup);
        } else {
          float qdist_ex;
          if (qd_a == FLT_MAX) {  // :792-808
            const float2 q2 = *qres_p;
            if (!result_ok(w, s, meta_err(ha.meta), st.qsegerr, q2.x)) { failed = true; break; }
            const float d2 = q2.x;
            const float s2 = q2.y < qs_a ? qs_a : q2.y;
            qd_a = d2; qs_a = s2;
            const HitCore han = load_core(hit_ptr<P4>(w, lo + a));
            store_dyn(w, kStride4, lo + a, d2, s2, han.seglen, han.meta);
            qdist_ex = d2 * st.bf_max;
            ++st.cnt2;
          } else qdist_ex = qd_a * st.bf_max;  // :809
          if (dist <= qdist_ex) st.d_unode_g = min(st.d_unode_g, cb);  // unode_global = LCA(unode_global, cnode), :811-813
        }
      }
      if (paused) {
        st.loop_live = 1; st.i_next = i;
        st.chunk_hi = p.spec2 ? min(n, i + p.spec2) : n;
        em = sel; em.mode = 3; em.i_lo = i; em.i_hi = st.chunk_hi; em.emit_query = 0; em.d_unode_g = st.d_unode_g;
      } else if (!failed) { st.phase = kPass2Select; st.loop_live = 0; }
    }


    // ---- choose what to align next ----
    if (!failed && st.phase == kPass1Select) {  // :582-583
      uint32_t a = 0;
      HitCore h = load_core(hit_ptr<P4>(w, lo));
      while (a + 1 < n && !(h.meta & kMetaQGroup)) { ++a; h = load_core(hit_ptr<P4>(w, lo + a)); }
      store_dyn(w, kStride4, lo + a, h.qd, h.qs, h.seglen, h.meta & ~kMetaQGroup);
      st.anchor = a; st.phase = kPass1Wait; st.loop_live = 0;
      st.chunk_hi = p.spec1 ? min(n, p.spec1) : n;
      em.mode = 2; em.anchor = a; em.i_lo = 0; em.i_hi = st.chunk_hi;
    } else if (!failed && st.phase == kPass2Select) {  // :739-747
      bool found = false;
      for (uint32_t a = 0; a < n; ++a) {
        const HitCore h = load_core(hit_ptr<P4>(w, lo + a));
        if (!(h.meta & kMetaOutGroup)) continue;
        store_dyn(w, kStride4, lo + a, h.qd, h.qs, h.seglen, h.meta & ~kMetaOutGroup);
        if (st.d_unode_g == st.d_lca_all) continue;  // unode_global == lca_allnodes: both lie on the best hit's path
        st.anchor = a; st.phase = kPass2Wait; found = true; st.loop_live = 0;
        st.chunk_hi = p.spec2 ? min(n, p.spec2) : n;
        em.mode = 3; em.anchor = a; em.best = st.best; em.d_unode_g = st.d_unode_g;
        em.i_lo = 0; em.i_hi = st.chunk_hi; em.emit_query = 1;
        em.fcut = h.score * X;                                   // qsearchscore_thresh_heuristic, :754
        const double qpid_anchor = (double)h.qs / L;             // :750-753
        const double g2 = qpid_anchor * 2. - 1., h3 = qpid_anchor * X;
        em.qpid_thresh = g2 < h3 ? h3 : g2;
        break;
      }
      if (!found) {  // :824-833
        if (st.d_unode_g == st.d_lnode_g) st.ival_g = 1.f;
        uint32_t pb[4 * P4];
        load_path<P4>(hit_ptr<P4>(w, lo + st.best), pb);
        OutRow row;
        row.begin = st.qrstart; row.end = st.qrstop; row.ival = st.ival_g;
        row.lower = node_at<P4>(pb, st.d_lnode_g); row.upper = node_at<P4>(pb, st.d_unode_g); row.support = st.support;
        row.rtax = node_at<P4>(pb, st.d_rtax); row.flags = 0;
        rows[s] = row;
        if (stats) stats[s] = OutStats{n, st.cnt0, st.cnt1, st.cnt2};
        if (prof_acc) atomicAdd(&prof_acc[(blockIdx.x & (kProfParts - 1)) * 4 + 3], (unsigned long long)(st.cnt0 + st.cnt1 + st.cnt2));
        if (st.d_unode_g > st.d_lnode_g) raise_error(w, kErrAssert, s);  // setNodeRange assert, predictionrecord.hh:153
        st.phase = kDone;
      }
    }
    if (failed) { st.phase = kDone; em.mode = 0; }
    if (st.phase != kDone) w.live_out[atomicAdd(&w.counters[kCtrActive], 1u)] = s;  // still running: next round's live list
    w.emit[s] = em;
    w.state[s] = st;
  }
}

// ---- launchers: P4 = number of 16-byte quads of the taxon path = ceil(max depth / 4) ----
#define RPA_DISPATCH_P4(P4V, CALL)                 \
  switch (P4V) {                                   \
    case 1: { constexpr int P4 = 1; CALL; } break; \
    case 2: { constexpr int P4 = 2; CALL; } break; \
    case 3: { constexpr int P4 = 3; CALL; } break; \
    case 4: { constexpr int P4 = 4; CALL; } break; \
    default: return hipErrorInvalidValue;          \
  }

hipError_t launch_rank_records(hipStream_t st, const BatchDev& b, const WorkDev& w, const TaxDev& tax, const StoreDev& q,
                               const StoreDev& db, const Params& p) {
  if (b.n_records == 0) return hipSuccess;
  const dim3 grid((b.n_records + kBlock - 1) / kBlock);
  RPA_DISPATCH_P4(tax.path_quads, hipLaunchKernelGGL(k_rank_records<P4>, grid, dim3(kBlock), 0, st, b, w, tax, q, db, p));
  return hipGetLastError();
}

hipError_t launch_plan_sets(hipStream_t st, const BatchDev& b, const WorkDev& w, const TaxDev& tax, const Params& p,
                            void* out_rows, void* out_stats) {
  if (b.n_sets == 0) return hipSuccess;
  const dim3 grid((b.n_sets + kBlock - 1) / kBlock);
  RPA_DISPATCH_P4(tax.path_quads, hipLaunchKernelGGL(k_plan_sets<P4>, grid, dim3(kBlock), 0, st, b, w, p, (OutRow*)out_rows, (OutStats*)out_stats));
  return hipGetLastError();
}

hipError_t launch_emit(hipStream_t st, const BatchDev& b, const WorkDev& w, const TaxDev& tax, const Params& p) {
  if (w.live_in) {  // sparse round: one wave per live set
    if (w.n_live_in == 0) return hipSuccess;
    const dim3 grid(((size_t)w.n_live_in * 64 + kBlock - 1) / kBlock);
    RPA_DISPATCH_P4(tax.path_quads, hipLaunchKernelGGL(k_emit_sparse<P4>, grid, dim3(kBlock), 0, st, b, w, p));
    return hipGetLastError();
  }
  const uint32_t total = b.n_records + b.n_sets;
  if (total == 0) return hipSuccess;
  const dim3 grid((total + kEmitBlock - 1) / kEmitBlock);
  RPA_DISPATCH_P4(tax.path_quads, hipLaunchKernelGGL(k_emit<P4>, grid, dim3(kEmitBlock), 0, st, b, w, p));
  return hipGetLastError();
}

hipError_t launch_decide(hipStream_t st, const BatchDev& b, const WorkDev& w, const TaxDev& tax, const Params& p,
                         void* out_rows, void* out_stats, unsigned long long* prof_acc) {
  const uint32_t n_threads = w.live_in ? w.n_live_in : b.n_sets;
  if (n_threads == 0) return hipSuccess;
  const dim3 grid((n_threads + kBlock - 1) / kBlock);
  RPA_DISPATCH_P4(tax.path_quads, hipLaunchKernelGGL(k_decide<P4>, grid, dim3(kBlock), (size_t)w.band_ranks * kBlock * 2 * sizeof(float), st, b, w, p, (OutRow*)out_rows, (OutStats*)out_stats, prof_acc));
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------------------
// Pair-level entry (rpa_align_pairs): every sequence of an arena-shaped store is one segment; explicit pair list.
// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_plan_plain_segments(WorkDev w, StoreDev store) {
  const uint32_t seg = blockIdx.x * kBlock + threadIdx.x;
  if (seg >= w.T) return;
  const uint64_t off = store.offsets[seg];
  w.seg[seg] = SegDesc{(uint32_t)off, (uint32_t)(off >> 32), (uint32_t)(store.offsets[seg + 1] - off), 0u};
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

hipError_t launch_bin_pairs(hipStream_t st, int alphabet, const WorkDev& w, uint32_t n_pairs) {
  if (n_pairs == 0) return hipSuccess;
  Params p{alphabet, 0.f, 0.f, 0u, 0u, -1};
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
