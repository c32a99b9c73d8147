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
// the batched kernels of align_*_kernels.hip.  One thread per set keeps the data-dependent control flow out of the
// wave-parallel alignment kernels; the work here is O(n) scalar logic plus <= 8-step LCA walks per set and round.
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
constexpr uint32_t kQGroup = 1, kOutGroup = 2;

__device__ __forceinline__ bool rec_filtered(const BatchDev& b, uint32_t r) { return b.filtered && b.filtered[r]; }

__device__ __forceinline__ void raise_error(const WorkDev& w, uint32_t code, uint32_t set) {
  if (atomicCAS(&w.counters[kCtrErr], 0u, code) == 0u) w.counters[kCtrErrSet] = set;
}

// ------------------------------------------------------------------------------------------------------------
// Segment planning (a7).  store.getSequence(id, start, stop): stop clipped to the sequence length, start beyond it is
// a range error (sequencestorage.hh:341-362); getSequenceDNA (:870-880): forward [rstart-left_ext, rstop+right_ext]
// with the start floored at 1; reverse (rstart > rstop) [rstop-right_ext, rstart+left_ext] reverse-complemented.
// Protein with reverse coordinates trips a live assert in the reference (:864).
// ------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void plan_segment(const WorkDev& w, uint32_t seg, const StoreDev& store, bool from_query,
                                             uint32_t seq, uint32_t start, uint32_t stop, bool rc, int bits) {
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
  const uint32_t per16 = 128u / (uint32_t)bits;  // residues per 16-byte unit
  w.seg_len[seg] = len;
  w.seg_cap16[seg] = (len + per16 - 1) / per16;
  w.seg_src[seg] = src;
  w.seg_info[seg] = (uint8_t)((rc ? 1u : 0u) | (from_query ? 2u : 0u) | (err << 4));
}

// ------------------------------------------------------------------------------------------------------------
// Rank + gather: one thread per record.  It scans the records of its own set once (neighbouring lanes read the same
// few cache lines) to get (a) its position in SortFilter order (score desc, identities desc, input order on full ties
// = what a stable list::sort gives) among the unmasked records, and (b) the set's n / query range (:391-404).  It
// then writes its HitRow at that sorted position -- from here on a set's hits are one contiguous 32-byte-per-hit
// array that the per-set threads walk sequentially -- and plans its own reference segment; the thread of a set's
// first record also plans the query-range segment and writes the set header.
// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_rank_records(BatchDev b, WorkDev w, StoreDev qstore, StoreDev dbstore, Params p) {
  const uint32_t r = blockIdx.x * kBlock + threadIdx.x;
  if (r >= b.n_records) return;
  uint32_t lo = 0, hi = b.n_sets;  // largest s with set_offsets[s] <= r  (empty sets share an offset: take the last one)
  while (hi - lo > 1) {
    const uint32_t mid = (lo + hi) >> 1;
    if (b.set_offsets[mid] <= r) lo = mid; else hi = mid;
  }
  const uint32_t s = lo;
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
  const int bits = p.alphabet == 0 ? 4 : 8;
  if (r == beg) {
    w.set_info[s] = make_uint4(n, qrstart, qrstop, 0u);
    if (n >= 2) plan_segment(w, b.n_records + s, qstore, true, b.query_seq[s], qrstart, qrstop, false, bits);  // :415
  }
  if (me_filtered) return;
  HitRow h;
  h.score = sc; h.ident = id; h.alnlen = b.alnlen[r]; h.node = b.node[r]; h.rec = r;
  h.qd = FLT_MAX; h.qs = 0.f; h.flags = 0;  // :475-476
  if (n == 1) { h.qd = __uint_as_float(b.qstart[r]); h.qs = __uint_as_float(b.qstop[r]); }  // n==1 needs them (:373-374)
  w.hits[beg + rank] = h;
  if (n < 2) return;
  const uint32_t left_ext = b.qstart[r] - qrstart, right_ext = qrstop - b.qstop[r];  // :523
  const uint32_t start = b.rstart[r], stop = b.rstop[r];
  if (start <= stop) {
    plan_segment(w, r, dbstore, false, b.ref_seq[r], left_ext < start ? start - left_ext : 1u, stop + right_ext, false, bits);
  } else if (p.alphabet != 0) {
    w.seg_len[r] = 0; w.seg_cap16[r] = 0; w.seg_src[r] = 0; w.seg_info[r] = (uint8_t)(kErrAssert << 4);
  } else {
    plan_segment(w, r, dbstore, false, b.ref_seq[r], right_ext < stop ? stop - right_ext : 1u, start + left_ext, true, bits);
  }
}

hipError_t launch_rank_records(hipStream_t st, const BatchDev& b, const WorkDev& w, const StoreDev& q, const StoreDev& db,
                               const Params& p) {
  if (b.n_records == 0) return hipSuccess;
  hipLaunchKernelGGL(k_rank_records, dim3((b.n_records + kBlock - 1) / kBlock), dim3(kBlock), 0, st, b, w, q, db, p);
  return hipGetLastError();
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

struct Emit {  // what a set wants aligned next
  uint32_t mode;   // 0 nothing, 1 pass 0, 2 pass 1 (anchor), 3 pass 2 (anchor)
  uint32_t anchor;
};

// Enumerate the pairs of an emission in loop order; F(i_or_tag, segA, segB) is called for each.
template <typename F>
__device__ __forceinline__ void for_each_pair(const BatchDev& b, const WorkDev& w, const TaxDev& tax, const Params& p,
                                              const SetState& st, uint32_t s, const Emit& em, F&& f) {
  const HitRow* __restrict__ hits = w.hits + b.set_offsets[s];
  const uint32_t n = st.n;
  const uint32_t qseg = b.n_records + s;
  if (em.mode == 1) {  // pass 0: every non-perfect hit within the re-evaluation band against the query range (:511-534)
    const float thr = p.reeval_factor * hits[0].score;
    for (uint32_t i = 0; i < n; ++i) {
      const HitRow h = hits[i];
      if (h.alnlen == st.L && h.ident == st.L) continue;
      if (h.score >= thr) f(i, h.rec, qseg);
    }
  } else if (em.mode == 2) {  // pass 1: everything the anchor loop can reach (:604-638)
    const uint32_t ra = hits[em.anchor].rec;
    for (uint32_t i = 0; i < n; ++i) {
      if (i == em.anchor || hits[i].qd == 0.f) continue;
      f(i, hits[i].rec, ra);
    }
  } else if (em.mode == 3) {  // pass 2 (:750-797)
    const uint32_t a = em.anchor;
    const HitRow ha = hits[a];
    if (ha.qd == FLT_MAX) f(kQueryTag, ha.rec, qseg);
    const double qpid_anchor = (double)ha.qs / st.L;
    const double g = qpid_anchor * 2. - 1., h = qpid_anchor * p.exclude_factor;
    const double qpid_thresh = g < h ? h : g;
    const float scut = ha.score * p.exclude_factor;
    for (uint32_t i = 0; i < n; ++i) {
      const HitRow hi = hits[i];
      if (!(hi.score >= scut)) break;
      if (i == a) continue;
      if (!((double)hi.qs / st.L >= qpid_thresh)) continue;
      if (d_is_parent_of(tax, st.unode_g, hi.node) || hi.node == st.unode_g) continue;
      f(i, hi.rec, ha.rec);
    }
  }
}

__device__ __forceinline__ int pair_bin(const WorkDev& w, const Params& p, uint32_t sa, uint32_t sb) {
  const uint32_t la = w.seg_len[sa], lb = w.seg_len[sb];
  const uint32_t m = la < lb ? la : lb;
  if (p.alphabet == 0) return dna_bin(m);
  return 0;  // protein: one kernel, one bin
}

// Convergent tail of plan/decide: reserve list space and write the pairs.  `live` lanes have something to emit.
__device__ __forceinline__ void emit_pairs(const BatchDev& b, const WorkDev& w, const TaxDev& tax, const Params& p,
                                           SetState& st, uint32_t s, bool live, const Emit& em) {
  uint32_t cnt[kNumBins];
#pragma unroll
  for (int k = 0; k < kNumBins; ++k) cnt[k] = 0;
  uint32_t bad = 0, maxlen = 0;
  if (live && em.mode) {
    for_each_pair(b, w, tax, p, st, s, em, [&](uint32_t, uint32_t sa, uint32_t sb) {
      const int bin = pair_bin(w, p, sa, sb);
      maxlen = max(maxlen, w.seg_len[sa]);
      if (bin >= kNumBins) { bad = 1; return; }
#pragma unroll
      for (int k = 0; k < kNumBins; ++k) cnt[k] += (k == bin) ? 1u : 0u;
    });
  }
  if (bad) raise_error(w, kErrUnsupported, s);
  wave_max_to(&w.counters[kCtrMaxLen], maxlen);
  uint32_t base[kNumBins];
#pragma unroll
  for (int k = 0; k < kNumBins; ++k) {
    // skip the wave-wide scan for bins nobody in the wave uses
    const bool any = __any(cnt[k] != 0);
    base[k] = any ? wave_reserve(&w.counters[k], cnt[k]) : 0u;
  }
  if (live && em.mode && !bad) {
    const uint32_t slot0 = b.set_offsets[s] + s;
    uint32_t k_slot = 0;
    for_each_pair(b, w, tax, p, st, s, em, [&](uint32_t i, uint32_t sa, uint32_t sb) {
      const uint32_t slot = slot0 + k_slot++;
      w.pair_a[slot] = sa;
      w.pair_b[slot] = sb;
      w.pair_i[slot] = i;
      const int bin = pair_bin(w, p, sa, sb);
#pragma unroll
      for (int k = 0; k < kNumBins; ++k)
        if (k == bin) w.bin_list[(size_t)k * w.T2 + base[k]++] = slot;
    });
    st.n_emitted = k_slot;
  } else if (live) st.n_emitted = 0;
}

// ------------------------------------------------------------------------------------------------------------
// Plan: one thread per record set.  Handles everything predict() decides before any alignment (:341-472): n == 0,
// n == 1, perfect-best-hit shortcut; otherwise emits pass 0.  (Masking, query range, sorting and segment plans were
// done per record by k_rank_records.)
// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_plan_sets(BatchDev b, WorkDev w, TaxDev tax, Params p, OutRow* __restrict__ rows,
                                                       OutStats* __restrict__ stats) {
  const uint32_t s = blockIdx.x * kBlock + threadIdx.x;
  const bool live = s < b.n_sets;
  SetState st;
  st.phase = kDone;
  Emit em{0, 0};
  if (live) {
    const uint32_t lo = b.set_offsets[s], hi = b.set_offsets[s + 1];
    uint4 info = make_uint4(0u, 0u, 0u, 0u);
    if (hi > lo) info = w.set_info[s];
    const uint32_t n = info.x, qrstart = info.y, qrstop = info.z;
    const HitRow* __restrict__ hits = w.hits + lo;
    st.n = n; st.qrstart = qrstart; st.qrstop = qrstop; st.L = qrstop - qrstart + 1;
    st.support = 0; st.lca_all = kRoot; st.rtax = kRoot; st.lnode_g = st.unode_g = kRoot; st.ival_g = 0.f; st.bf_max = 1.f;
    st.lrd_min = 255; st.anchor = 0; st.cnt0 = st.cnt1 = st.cnt2 = 0; st.n_emitted = 0;
    OutRow row;
    row.begin = 1; row.end = b.query_len[s];  // initPredictionRecord, core/src/taxonpredictionmodel.hh:42-44
    row.lower = row.upper = row.rtax = kRoot; row.support = 0; row.ival = 0.f; row.flags = 0;
    bool done = true;
    if (n == 0) {  // :359-368
      row.flags = kFlagIvalUntouched;
    } else if (n == 1) {  // :371-388
      const HitRow h = hits[0];
      row.begin = __float_as_uint(h.qd); row.end = __float_as_uint(h.qs); row.ival = 1.f;
      row.lower = h.node; row.upper = kRoot; row.support = h.ident; row.rtax = h.node;
      row.flags = kFlagSingle;
    } else {
      const HitRow h0 = hits[0];
      const uint32_t L = st.L;
      if (h0.alnlen == L && h0.ident == L) {  // :431-472
        const float best = h0.score;
        uint32_t lnode = h0.node, unode = kRoot;
        uint32_t i = 1;
        for (;;) {
          if (i == n) { unode = kRoot; break; }
          const float sc = hits[i].score;
          if (sc == best) lnode = d_lca(tax, lnode, hits[i].node);
          else {
            unode = lnode;
            do { unode = d_lca(tax, unode, hits[i].node); } while (++i < n && hits[i].score == sc);
            break;
          }
          ++i;
        }
        row.begin = qrstart; row.end = qrstop; row.ival = 0.f;
        row.lower = lnode; row.upper = unode; row.support = L; row.rtax = lnode; row.flags = kFlagShortcut;
      } else {
        done = false;
        st.phase = kPass0Wait;
        em.mode = 1;
      }
    }
    if (done) {
      rows[s] = row;
      if (stats) stats[s] = OutStats{n, 0, 0, 0};
    } else {
      atomicAdd(&w.counters[kCtrActive], 1u);
    }
  }
  emit_pairs(b, w, tax, p, st, s, live, em);
  if (live) w.state[s] = st;
}

hipError_t launch_plan_sets(hipStream_t st, const BatchDev& b, const WorkDev& w, const TaxDev& tax, const Params& p,
                            void* out_rows, void* out_stats) {
  if (b.n_sets == 0) return hipSuccess;
  hipLaunchKernelGGL(k_plan_sets, dim3((b.n_sets + kBlock - 1) / kBlock), dim3(kBlock), 0, st, b, w, tax, p,
                     (OutRow*)out_rows, (OutStats*)out_stats);
  return hipGetLastError();
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

// ------------------------------------------------------------------------------------------------------------
// Decide: one thread per record set, one call per round.
// ------------------------------------------------------------------------------------------------------------
struct ResultCursor {  // walks this set's pair slots in emission order
  const WorkDev& w;
  uint32_t slot, end;
  __device__ bool find(uint32_t i, float2& out, uint32_t& sa, uint32_t& sb) {
    while (slot < end && w.pair_i[slot] != i) ++slot;
    if (slot >= end) return false;
    out = w.pair_res[slot]; sa = w.pair_a[slot]; sb = w.pair_b[slot];
    ++slot;
    return true;
  }
};

// A consumed alignment must come from two readable segments, and (protein) must not have tripped the reference's
// asserts selfscore >= mutual / distance >= 0 (:205,:240), which the kernel flags by a NaN distance.
__device__ __forceinline__ bool seg_ok(const WorkDev& w, uint32_t s, uint32_t sa, uint32_t sb, float dist) {
  const uint32_t e = (w.seg_info[sa] >> 4) | (w.seg_info[sb] >> 4);
  if (e) { raise_error(w, (w.seg_info[sa] >> 4) ? (w.seg_info[sa] >> 4) : (w.seg_info[sb] >> 4), s); return false; }
  if (dist != dist) { raise_error(w, kErrAssert, s); return false; }
  return true;
}

__global__ __launch_bounds__(kBlock) void k_decide(BatchDev b, WorkDev w, TaxDev tax, Params p, OutRow* __restrict__ rows,
                                                    OutStats* __restrict__ stats, unsigned long long* prof_acc) {
  __shared__ float s_fmin[kMaxDepth + 1][kBlock];
  __shared__ float s_dmax[kMaxDepth + 1][kBlock];
  const uint32_t s = blockIdx.x * kBlock + threadIdx.x;
  bool live = s < b.n_sets;
  SetState st;
  st.phase = kDone;
  if (live) st = w.state[s];
  live = live && st.phase != kDone;
  Emit em{0, 0};
  if (live) {
    const uint32_t lo = b.set_offsets[s];
    const uint32_t n = st.n, L = st.L;
    const uint32_t slot0 = lo + s;
    HitRow* __restrict__ hits = w.hits + lo;  // this set's hits in SortFilter order, 32 B each, contiguous
    const float X = p.exclude_factor;
    ResultCursor cur{w, slot0, slot0 + st.n_emitted};
    bool failed = false;

    if (st.phase == kPass0Wait) {  // ---- pass 0, :497-566 ----
      const float thr = p.reeval_factor * hits[0].score;  // :499
      uint32_t index_best = 0;
      float best_d = 0.f, best_s = 0.f, best_sc = 0.f;
      uint32_t support = 0;
      uint32_t lca_all = hits[0].node;
      for (uint32_t i = 0; i < n; ++i) {
        const HitRow h = hits[i];
        float dist, sim;
        uint32_t flag = 0;
        if (h.alnlen == L && h.ident == L) {  // :511-517
          flag = kQGroup; dist = 0.f; sim = (float)h.ident;
        } else if (h.score >= thr) {  // :518-534
          flag = kQGroup;
          float2 res; uint32_t sa, sb;
          if (!cur.find(i, res, sa, sb)) { raise_error(w, kErrInvalid, s); failed = true; break; }
          if (!seg_ok(w, s, sa, sb, res.x)) { failed = true; break; }
          dist = res.x;
          ++st.cnt0;
          const float idf = (float)h.ident;
          sim = res.y < idf ? idf : res.y;  // std::max(similarity, identities), :531
        } else {
          dist = FLT_MAX; sim = (float)h.ident;
        }
        hits[i].qd = dist; hits[i].qs = sim; hits[i].flags = flag;
        if (i == 0) { best_d = dist; best_s = sim; best_sc = h.score; }  // index_best starts at 0 (:500)
        else if (dist < best_d) { index_best = i; best_d = dist; best_s = sim; best_sc = h.score; }  // :542-546
        else if (dist == best_d) {
          if (sim > best_s) { index_best = i; best_s = sim; best_sc = h.score; }
          else if (sim == best_s && h.score > best_sc) { index_best = i; best_sc = h.score; }
        }
        support = max(support, (uint32_t)sim);     // :547
        lca_all = d_lca(tax, lca_all, h.node);      // :548
      }
      if (!failed) {
        uint32_t rtax = hits[index_best].node;
        bool any = false;
        for (uint32_t i = 0; i < n; ++i) {  // :554-562
          const HitRow h = hits[i];
          if (!(h.flags & kQGroup)) continue;
          if (h.qd != best_d || h.qs != best_s || h.score != best_sc) hits[i].flags = 0;
          else { rtax = d_lca(tax, rtax, h.node); any = true; }
        }
        if (!any) { raise_error(w, kErrAssert, s); failed = true; }  // assert(!qgroup.empty()), :563
        st.support = support; st.lca_all = lca_all; st.rtax = rtax;
        st.ival_g = 0.f; st.lnode_g = st.unode_g = rtax; st.bf_max = 1.f; st.lrd_min = 255;  // :569-579
        st.phase = kPass1Select;
      }
    } else if (st.phase == kPass1Wait) {  // ---- pass 1 for one anchor, :580-728 ----
      const uint32_t a = st.anchor;
      const float qdist = hits[a].qd;
      const uint32_t rnode = hits[a].node;
      uint32_t* __restrict__ ogi = w.og_idx + slot0;
      int32_t* __restrict__ ogd = w.og_dist + slot0;
      uint32_t nog = 0;
      BandAcc band{&s_fmin[0][threadIdx.x], &s_dmax[0][threadIdx.x], 0};
      band.reset(tax.depth[rnode]);  // BandFactor(...); addSequence(0, rnode) (:581,586): the anchor entry only sets the start rank
      uint32_t lnode = st.rtax;
      float ldist = 0.f, udist = FLT_MAX;
      double qpid_upper = 0., cut_g = 0., cut_h = 0.;  // :599-601
      int scut = 0;                                     // :602 (int!)
      for (uint32_t i = 0; lnode != kRoot && i < n; ++i) {  // :604
        const HitRow h = hits[i];
        if (!(h.score >= (float)scut)) break;
        const uint32_t cnode = h.node;
        const double qsearchpid = (double)h.ident / L;
        const double qpid = (double)h.qs / L;
        const double qpid_thresh = cut_g < cut_h ? cut_h : cut_g;
        if (!(qpid >= qpid_thresh)) continue;  // :612
        float dist;
        if (i == a) dist = 0.f;
        else if (h.qd == 0.f) dist = qdist;  // :620-621
        else {
          float2 res; uint32_t sa, sb;
          if (!cur.find(i, res, sa, sb)) { raise_error(w, kErrInvalid, s); failed = true; break; }
          if (!seg_ok(w, s, sa, sb, res.x)) { failed = true; break; }
          dist = res.x;
          ++st.cnt1;
        }
        band.add(dist, tax.depth[d_lca(tax, cnode, rnode)]);  // :641
        if (dist == 0.f) hits[i].flags = h.flags & ~(uint32_t)kQGroup;  // :644
        else if (dist <= qdist) {  // :646-650
          lnode = d_lca(tax, lnode, cnode);
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
        uint32_t unode = lnode;  // :690 -- (float)INT_MAX != FLT_MAX, so always taken
        for (uint32_t k = 0; k < nog; ++k) {  // :691-711
          const uint32_t i = ogi[k];
          const float dist = (float)ogd[k];
          const uint32_t cnode = hits[i].node;
          if (dist > min_upper_dist) continue;
          unode = d_lca(tax, cnode, unode);
          const uint32_t lca_root_dist = tax.depth[d_lca(tax, cnode, st.rtax)];
          if (lca_root_dist > st.lrd_min) continue;
          if (lca_root_dist < st.lrd_min) {
            st.lrd_min = lca_root_dist;
            for (uint32_t j = 0; j < n; ++j) hits[j].flags &= ~(uint32_t)kOutGroup;
          }
          hits[i].flags |= kOutGroup;
        }
        float ival = 0.f;  // :714-719
        if (unode != lnode && ldist < qdist) ival = (qdist - ldist) / (udist - ldist);
        st.ival_g = st.ival_g < ival ? ival : st.ival_g;          // :724
        st.unode_g = d_lca(tax, st.unode_g, unode);               // :726
        st.lnode_g = d_lca(tax, st.lnode_g, lnode);               // :727
        bool more = false;
        for (uint32_t j = 0; j < n; ++j) more |= (hits[j].flags & kQGroup) != 0;
        st.phase = (more && st.lnode_g != kRoot) ? kPass1Select : kPass2Select;  // :729
      }
    } else if (st.phase == kPass2Wait) {  // ---- pass 2 for one anchor, :750-820 ----
      const uint32_t a = st.anchor;
      const HitRow ha = hits[a];
      float qd_a = ha.qd, qs_a = ha.qs;
      const double qpid_anchor = (double)qs_a / L;
      const double g = qpid_anchor * 2. - 1., h2 = qpid_anchor * X;
      const double qpid_thresh = g < h2 ? h2 : g;
      const float scut = ha.score * X;  // :754 (float here)
      // the (anchor, query) pair, if emitted, sits in the first slot
      float2 qres = make_float2(0.f, 0.f);
      bool have_q = false;
      if (cur.slot < cur.end && w.pair_i[cur.slot] == kQueryTag) {
        qres = w.pair_res[cur.slot];
        have_q = true;
        ++cur.slot;
      }
      const uint32_t qslot = slot0;
      for (uint32_t i = 0; i < n; ++i) {  // :757
        const HitRow h = hits[i];
        if (!(h.score >= scut)) break;
        const double qpid = (double)(i == a ? qs_a : h.qs) / L;
        if (!(qpid >= qpid_thresh)) continue;
        const uint32_t cnode = h.node;
        float dist;
        if (i == a) dist = 0.f;
        else {
          if (d_is_parent_of(tax, st.unode_g, cnode) || cnode == st.unode_g) continue;  // :771
          float2 res; uint32_t sa, sb;
          if (!cur.find(i, res, sa, sb)) { raise_error(w, kErrInvalid, s); failed = true; break; }
          if (!seg_ok(w, s, sa, sb, res.x)) { failed = true; break; }
          dist = res.x;
          ++st.cnt2;
          hits[i].qd = dist;  // :785
        }
        if (dist == 0.f) hits[i].flags = h.flags & ~(uint32_t)kOutGroup;  // :789
        else {
          float qdist_ex;
          if (qd_a == FLT_MAX) {  // :792-808
            if (!have_q) { raise_error(w, kErrInvalid, s); failed = true; break; }
            if (!seg_ok(w, s, w.pair_a[qslot], w.pair_b[qslot], qres.x)) { failed = true; break; }
            const float d2 = qres.x;
            const float s2 = qres.y < qs_a ? qs_a : qres.y;
            qd_a = d2; qs_a = s2;
            hits[a].qd = d2; hits[a].qs = s2;
            qdist_ex = d2 * st.bf_max;
            ++st.cnt2;
          } else qdist_ex = qd_a * st.bf_max;  // :809
          if (dist <= qdist_ex) st.unode_g = d_lca(tax, st.unode_g, cnode);  // :811-813
        }
      }
      if (!failed) st.phase = kPass2Select;
    }

    // ---- choose what to align next ----
    if (!failed && st.phase == kPass1Select) {  // :582-583
      uint32_t a = 0;
      while (a < n && !(hits[a].flags & kQGroup)) ++a;
      hits[a].flags &= ~(uint32_t)kQGroup;
      st.anchor = a; st.phase = kPass1Wait; em.mode = 2; em.anchor = a;
    } else if (!failed && st.phase == kPass2Select) {  // :739-747
      bool found = false;
      for (uint32_t a = 0; a < n; ++a) {
        const uint32_t f = hits[a].flags;
        if (!(f & kOutGroup)) continue;
        hits[a].flags = f & ~(uint32_t)kOutGroup;
        if (st.unode_g == st.lca_all) continue;
        st.anchor = a; st.phase = kPass2Wait; em.mode = 3; em.anchor = a; found = true;
        break;
      }
      if (!found) {  // :824-833
        if (st.unode_g == st.lnode_g) st.ival_g = 1.f;
        OutRow row;
        row.begin = st.qrstart; row.end = st.qrstop; row.ival = st.ival_g;
        row.lower = st.lnode_g; row.upper = st.unode_g; row.support = st.support; row.rtax = st.rtax; row.flags = 0;
        rows[s] = row;
        if (stats) stats[s] = OutStats{n, st.cnt0, st.cnt1, st.cnt2};
        if (prof_acc) atomicAdd(&prof_acc[3], (unsigned long long)(st.cnt0 + st.cnt1 + st.cnt2));
        if (!(row.lower == row.upper || d_is_parent_of(tax, row.upper, row.lower))) raise_error(w, kErrAssert, s);  // predictionrecord.hh:153
        st.phase = kDone;
      }
    }
    if (failed) st.phase = kDone;
    if (st.phase != kDone) atomicAdd(&w.counters[kCtrActive], 1u);
  }
  emit_pairs(b, w, tax, p, st, s, live, em);
  if (live) w.state[s] = st;
}

hipError_t launch_decide(hipStream_t st, const BatchDev& b, const WorkDev& w, const TaxDev& tax, const Params& p,
                         void* out_rows, void* out_stats, unsigned long long* prof_acc) {
  if (b.n_sets == 0) return hipSuccess;
  hipLaunchKernelGGL(k_decide, dim3((b.n_sets + kBlock - 1) / kBlock), dim3(kBlock), 0, st, b, w, tax, p, (OutRow*)out_rows,
                     (OutStats*)out_stats, prof_acc);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------------------
// Pair-level entry (rpa_align_pairs): every sequence of an arena-shaped store is one segment; explicit pair list.
// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_plan_plain_segments(WorkDev w, StoreDev store, int bits) {
  const uint32_t seg = blockIdx.x * kBlock + threadIdx.x;
  if (seg >= w.T) return;
  const uint64_t off = store.offsets[seg];
  const uint32_t len = (uint32_t)(store.offsets[seg + 1] - off);
  const uint32_t per16 = 128u / (uint32_t)bits;
  w.seg_len[seg] = len;
  w.seg_cap16[seg] = (len + per16 - 1) / per16;
  w.seg_src[seg] = off;
  w.seg_info[seg] = 0;
}

hipError_t launch_plan_plain_segments(hipStream_t st, int alphabet, const WorkDev& w, const StoreDev& store) {
  if (w.T == 0) return hipSuccess;
  hipLaunchKernelGGL(k_plan_plain_segments, dim3((w.T + kBlock - 1) / kBlock), dim3(kBlock), 0, st, w, store, alphabet == 0 ? 4 : 8);
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
      bin = pair_bin(w, p, sa, sb);
      const uint32_t la = w.seg_len[sa], lb = w.seg_len[sb];
      maxlen = la;
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
    if (c) w.bin_list[(size_t)k * w.T2 + base] = t;
  }
}

hipError_t launch_bin_pairs(hipStream_t st, int alphabet, const WorkDev& w, uint32_t n_pairs) {
  if (n_pairs == 0) return hipSuccess;
  Params p{alphabet, 0.f, 0.f};
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
