// taxator-tk_amd/csrc/rpa_device.h -- shared declarations of the gfx950 RPA engine (device views + kernel launchers).
// Hand-written HIP for CDNA4 only; no CPU fallback exists behind these functions.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rpa {

constexpr int kBlock = 256;          // 4 wave64 per workgroup everywhere
constexpr uint32_t kRoot = 0;        // taxonomy node 0 is the root
constexpr int kShortWords = 12;      // per-lane Myers kernels: pattern <= 12*32 = 384 residues in one register strip (8 words until round 5: the
                                     // two-strip form of a 257 ... 384 nt pair ran at a third of the one-strip rate per cell, tools/pairs_bench.py)
constexpr int kWaveMaxWords = 8;     // K1w (a pattern word per lane, eight lanes per pair): patterns of up to 256 residues
constexpr int kStripMinWords = 5;    // longer patterns: register strips of 5..8 words, whichever wastes the fewest rows
constexpr int kStripMaxWords = 8;
constexpr int kStripBins = kStripMaxWords - kStripMinWords + 1;
constexpr int kWindowBins = 3;       // patterns of 13..16 / 17..32 / 33..64 words: the ring-window kernel (K1r, windows of 6 / 10 / 10 words) in front of the strips
constexpr int kWindowMaxWords = 64;
constexpr int kFirstStripBin = kShortWords + kWindowBins;
constexpr int kDnaBins = kFirstStripBin + kStripBins;  // bins 0..11: one strip of bin+1 words; 12, 13, 14: up to 16 / 32 / 64 words; 15..18: strips of bin-10 words
// window words of a class: the narrow instance, and the wide one the engine moves to when more than a quarter of a batch's pairs left the narrow one
__host__ __device__ constexpr int window_words_of_bin(int bin, bool wide = false) { return bin == kShortWords ? (wide ? 8 : 6) : (wide ? 14 : 10); }
// Classes with a ladder of banded kernels the engine climbs per batch (rpa_engine.hip: window_feedback): the one-strip classes of 9..12
// words (the four-word band, 80 edits -> the ring window of 6 words, 160 -> every cell) and the ring-window classes (narrow -> wide -> strips)
constexpr int kLadderBins = kFirstStripBin - kWaveMaxWords;  // bins kWaveMaxWords .. kFirstStripBin - 1
constexpr int kNumBins = 22;         // work lists per round: the nucleotide bins above, or the 22 protein bins below
constexpr uint32_t kBigSetMax = 1024; // largest record set k_rank_big_sets takes (a workgroup per set: 21 bytes of LDS per record, two hash words per table slot)
constexpr int kLongMaxWPL = 8;       // wave-per-pair Myers kernel (few long pairs): pattern <= 64 lanes * 8 words * 32 = 16384 residues
constexpr uint32_t kWaveMaxPattern = 64u * kLongMaxWPL * 32u;
constexpr int kProfParts = 64;        // profiling accumulators are split 64 ways (block index & 63) and summed on the host
constexpr int kProfSlots = 8;         // per part: pairs, residue bytes, cells (len_a x len_b), demanded alignments, cells really UPDATED (band / strip sweeps), 3 spare
constexpr int kMaxDepth = 63;         // deepest taxonomy level: the BandFactor rank sets of k_decide are 64-bit masks and its per-rank LDS
                                      // accumulators hold (depth + 1) x 2 KB (full NCBI: ~40 levels; pruned to the default ranks: 8)

// error codes mirror include/taxator_rpa.h
enum { kOk = 0, kErrInvalid = 1, kErrDevice = 2, kErrRange = 3, kErrSeqId = 4, kErrAssert = 5, kErrUnsupported = 6,
       kErrNotResident = 7,    // device-only: a segment outside the resident windows of its sequence (reported as RPA_ERR_INVALID)
       kErrNanScore = 8 };     // device-only: a NaN score in a record set of two or more unmasked records (reported as RPA_ERR_INVALID)

// Flattened taxonomy (reference: Taxon::leftvalue/rightvalue/root_pathlength, core/src/taxontree.hh:64-66).
struct TaxDev {
  const uint32_t* parent;
  const uint32_t* left;
  const uint32_t* right;
  const uint8_t* depth;
  uint32_t n_nodes;
  // Built at rpa_create from the parent pointers alone (the caller's left / right are only validated):
  //   pre[node] / last[node]: preorder number of the node / of the last node of its subtree, so that
  //       x lies in the subtree of a (a itself included)  <=>  pre[a] <= pre[x] <= last[a];
  //   anc[node * max_depth + (d-1)]    = ancestor of node at depth d (d = 1 .. depth[node]; 0xFFFFFFFF beyond);
  //   anc_iv[node * roundup8(max_depth) + (d-1)] = (pre, last - pre) of that ancestor, so that x lies under it  <=>
  //       pre[x] - anc_iv.x <= anc_iv.y in unsigned arithmetic ((0xFFFFFFFF, 0) beyond the node's depth: nothing does).
  // k_rank_records turns the intervals into per-record-set taxon path LABELS (see place_kernels.hip), the placement
  // kernels compare labels and never walk `parent`; anc gives the node at a depth of the best hit's path for the output.
  const uint32_t* pre;
  const uint32_t* last;
  const uint32_t* anc;
  const uint2* anc_iv;
  uint32_t max_depth;  // deepest level = entries per node of anc / anc_iv (>= 1)
};

// Resident sequence store: residues packed little-endian into 32-bit words, 4 bit (DNA) or 8 bit (protein) each.
struct StoreDev {
  const uint32_t* words;
  // Whole-sequence stores (rpa_store_load / rpa_store_reserve + rpa_store_append): sequence i = residues
  // [extent[i].x, extent[i].x + extent[i].y) of the arena; y == 0: not resident.
  // Windowed stores (rpa_store_reserve_windows): extent[i].y = length of sequence i (known from the index file whether
  // anything of it is resident or not), extent[i].x = first entry of its window list in `windows`, win_count[i] entries:
  // (first residue of the window inside the sequence, residues, arena offset low, high), sorted by first residue, disjoint.
  // A segment is readable iff one window holds all of it (the driver appends windows that way).
  const ulonglong2* extent;
  uint32_t n_seqs;
  const uint4* windows;       // null: whole-sequence store
  const uint32_t* win_count;  // (host-side bookkeeping mirror; the kernels read the count out of extent[i].x, below)
};
// windowed stores: extent[i].x = first entry of sequence i's window list (low kWinFirstBits bits) | its number of entries << kWinFirstBits
constexpr int kWinFirstBits = 40;

// Batch (device pointers), same meaning as rpa_batch in include/taxator_rpa.h.
struct BatchDev {
  uint32_t n_sets, n_records;
  const uint32_t* set_offsets;
  const uint32_t* query_seq;
  const uint32_t* query_len;
  const uint32_t *qstart, *qstop, *rstart, *rstop;
  const float* score;
  const uint32_t *identities, *alnlen, *ref_seq, *node;
  const uint8_t* filtered;  // may be null
};

enum Phase : uint32_t { kDone = 0, kPass0Wait = 1, kPass1Wait = 2, kPass2Wait = 3, kPass1Select = 4, kPass2Select = 5 };

// Per record-set state carried between rounds (one thread owns one set).  Taxon nodes are kept as DEPTHS on the
// best hit's root path (see place_kernels.hip).  This is the form the kernels work on in registers; between rounds a set's
// state lives in memory as SetStateMem (64 bytes: what changes from round to round) + its set_info quad (what never
// changes) + a LoopSave record that is written and read only by a pass-1 loop that pauses at the end of its chunk.
struct SetState {
  uint32_t phase;
  uint32_t n;          // active (unmasked) records
  uint32_t L;          // qrlength
  uint32_t qrstart, qrstop;
  uint32_t qseglen;    // length of the query range segment after clipping
  uint32_t qsegerr;    // its fetch error code (0 = ok)
  uint32_t support;    // anchors_support
  uint32_t best;       // index_best of pass 0: the hit whose path the depths below refer to
  uint32_t d_lca_all;  // lca_allnodes
  uint32_t d_rtax;
  uint32_t d_lnode_g, d_unode_g;
  float ival_g;
  float bf_max;
  uint32_t lrd_min;    // lca_root_dist_min
  uint32_t anchor;
  uint32_t cnt0, cnt1, cnt2;
  // resumable loops: with a speculation chunk smaller than the set, a pass-1 / pass-2 loop runs over several rounds
  uint32_t loop_live;  // 0: the loop of `anchor` has not started, 1: paused at i_next
  uint32_t i_next;     // first hit the paused loop has not processed
  uint32_t chunk_hi;   // hits below this index had their alignment emitted
  uint32_t nog;        // pass 1: entries of outgroup_tmp so far (sets of more than 64 hits: a list in og_idx / og_dist);
                       // sets of up to 64 hits: the low word of outgroup_tmp as a bit set over the hits (k_decide)
  uint32_t d_lnode;    // pass 1: lnode (depth on the best hit's path)
  float ldist, udist;
  int32_t scut;        // qsearchscore_thresh_heuristic (:602)
  double qpid_upper, cut_g, cut_h;
  uint32_t combined;   // pass 0 and the pass-1 loop of hit 0 were requested together (EmitInfo mode 4)
  uint32_t pad_;       // sets of up to 64 hits: the high word of that bit set
  uint32_t gq, go;     // sets of up to 32 hits: the query-group / outgroup word (else WorkDev::flags holds the words)
};

// One hit of a record set at its SortFilter position: Q quads of 16 bytes (Q = WorkDev::row_quads, 2 for taxonomies of up
// to 8 levels and record sets of up to 256 records: 32 bytes per hit).  Words 0-3 = score, identities, node, meta (static
// after k_rank_records); 4-5 = querydistance, querysimilarity (the only mutable fields); 6 .. 4Q-1 = the hit's taxon path
// as per-set LABELS (place_kernels.hip).  The length of a hit's reference segment lives in WorkDev::lite.
struct HitCore {
  float score;       // getScore()
  uint32_t ident;    // getIdentities()
  uint32_t node;     // getReferenceNode()
  uint32_t meta;     // kMeta* bits
  float qd;          // querydistance[i]
  float qs;          // querysimilarity[i]
};
// One alignment the reference's control flow consumed (= one "+ALN" line of its log, taxonpredictionmodelsequence.hh:533,
// 636,782,802): recorded by k_decide when tracing is on (rpa_trace_begin).  j = -1: the query range; i = -1: "query <=> j".
struct TraceEntry { uint32_t set, seq; int32_t pass, i, j; float dist, sim; uint32_t reserved; };
struct SetStateMem { uint4 q[4]; };  // see state_pack / state_unpack (place_kernels.hip)
struct LoopSave {                    // a paused pass-1 loop: :596-602 loop variables
  uint32_t nog, pad_, d_lnode;
  int32_t scut;
  float ldist, udist;
  double qpid_upper, cut_g, cut_h;
};
constexpr uint32_t kLiteLenMask = 0x1FFFFFFFu;  // segments (reference or query range) of 2^29 residues and more are refused
constexpr uint32_t kLitePerfect = 0x20000000u;  // alnlen == L && identities == L: a member of qgroup with distance 0 (:511-517)
constexpr uint32_t kLiteWant0 = 0x80000000u;    // pass 0 aligns this hit to the query range (not perfect, score >= threshold)
constexpr uint32_t kLiteZeroQd = 0x40000000u;   // querydistance[i] == 0 after pass 0: pass 1 never aligns it (:620)
constexpr uint32_t kMetaPerfect = 4u;  // perfect: alnlen == L && identities == L
constexpr int kMetaErrShift = 4, kMetaDepthShift = 8;                         // 4 bits segment error code, 8 bits node depth

// Segment descriptor (a7): `len` residues starting at residue `src` of a store; reverse strand = read backwards + complement.
struct SegDesc {
  uint32_t src_lo, src_hi;  // first residue (global residue index in the store), 64 bit
  uint32_t len;             // 0 if never planned / unreadable
  uint32_t info;            // bit0 reverse-complement, bit1 from the query store, bits 4-7 fetch error code
};

// What a record set wants aligned in the coming round (written by plan/decide, read by k_emit and the alignment kernels).
struct EmitInfo {
  uint32_t mode;        // 0 nothing, 1 pass 0 (vs query range), 2 pass 1 (vs anchor), 3 pass 2 (vs anchor),
                        // 4 pass 0 + pass 1 in one round: the only hit pass 0 aligns (hit `anchor` = 0) vs the query range,
                        //   every other hit vs that hit (see k_plan_sets)
                        // 5 pass 1 like 2, made when the paused loop already has thresholds: hits they exclude are left out
  uint32_t anchor;      // index_anchor
  uint32_t n;           // active hits
  uint32_t L;           // qrlength
  float fcut;           // pass 0: re-evaluation score threshold (:499); pass 2: qsearchscore_thresh_heuristic (:754); mode 5: (:604, :659)
  uint32_t d_unode_g;   // pass 2: depth of unode_global on the best hit's path
  double qpid_thresh;   // pass 2 (:750-753)
  uint32_t best;        // pass 2: index_best
  uint32_t qseglen;     // length of the query range segment
  uint32_t i_lo, i_hi;  // only hits with i_lo <= i < i_hi are emitted this round (speculation chunk of the running loop)
  uint32_t emit_query;  // pass 2: also emit the anchor <-> query alignment (first chunk of an anchor whose querydistance is unset)
  uint32_t anchor_len;  // pass 1/2: length of the anchor's reference segment
  float qs_min;         // pass 2, mode 5: a similarity below which (double)qs / L >= qpid_thresh cannot hold (place_kernels.hip: qs_threshold)
};

// The 16 bytes of EmitInfo that k_emit needs in passes 0 and 1 (one load instead of the 56-byte struct):
// x = mode | anchor << 3, y = i_lo (mode 4: length of the query range), z = i_hi (<= n), w = length of the other segment
// (query range in pass 0, anchor in pass 1/2 and in mode 4)
__host__ __device__ __forceinline__ uint4 emit_desc(const EmitInfo& e) {
  return uint4{e.mode | (e.anchor << 3), e.mode == 4u ? e.qseglen : e.i_lo, e.i_hi < e.n ? e.i_hi : e.n, e.mode == 1u ? e.qseglen : e.anchor_len};
}

// ... and the 16 bytes more that a pass-2 request (mode 3) or a pass-1 request with thresholds (mode 5) adds: k_emit reads them
// beside emit_lite instead of fetching the 64-byte EmitInfo behind it (one dependent load level less per row).
// x = fcut, y = qs_min (float bits), z = best | d_unode_g << 24, w = qseglen | emit_query << 31
__host__ __device__ __forceinline__ uint4 emit_ext_quad(const EmitInfo& e) {
  union { float f; uint32_t u; } a, b;
  a.f = e.fcut; b.f = e.qs_min;
  return uint4{a.u, b.u, (e.best & 0x00FFFFFFu) | (e.d_unode_g << 24), (e.qseglen & 0x7FFFFFFFu) | (e.emit_query ? 0x80000000u : 0u)};
}

// Per-batch working set (all device pointers; sizes in comments; H = n_records, S = n_sets, T = H + S).
struct WorkDev {
  SetStateMem* state;   // S
  LoopSave* loop_save;  // S
  uint32_t* hits;       // P row slots of row_quads quads in SortFilter order per set; tiled or linear (SetRows, place_kernels.hip)
  const uint32_t* tile_base;  // ceil(S/64) first row slot of every tile of 64 record sets; null = linear layout (slot = row position)
  uint32_t* res_slot;   // H   row position -> slot of pair_res that holds the hit's alignment result (null in rpa_align_pairs)
  uint32_t P;           // row slots (= H in the linear layout); the anchor<->query result of set s is at pair_res[P + s]
  uint4* set_info;      // S   (n active, qrstart, qrstop, length of the query range segment | its fetch error code << 29)
  EmitInfo* emit;       // S   written for pass 2 only (mode 3): passes 0 and 1 are described completely by emit_lite
  uint4* emit_lite;     // S   emit_desc(emit[s])
  uint4* emit_ext;      // S   emit_ext_quad(emit[s]): what k_emit needs of a pass-2 / mode-5 request beyond emit_lite (written with emit[s])
  const uint32_t* rec_pre;  // H   record -> preorder number of its taxon (k_validate_batch)
  uint32_t* hit_set;    // H   row position -> record set
  uint32_t* block_first;  // per workgroup of k_rank_records (logical order): record set of its first record
  // qgroup / outgroup membership (:480-481,:578) as bit sets: for a set with n >= 2 active hits at rows [lo, lo+n),
  // words flags[lo .. lo+nw) hold the qgroup bits and flags[lo+nw .. lo+2nw) the outgroup bits, nw = ceil(n/32) (2nw <= n)
  uint32_t* flags;      // H
  // what k_emit needs of a row in passes 0 and 1, 4 bytes instead of the 64-byte row: segment length | kLite* bits
  uint32_t* lite;       // H
  // segments: id pos in [0,H) = reference segment of the hit at row pos, id H+s = query range of set s.  A segment is only a
  // descriptor: the alignment kernels read the residues straight out of the resident stores (funnel-shifted to the
  // segment's start, reverse-complemented on the fly for reverse-strand hits) -- no per-batch copy of the sequences.
  SegDesc* seg;               // T
  const uint32_t* store_q;    // packed query store
  const uint32_t* store_db;   // packed reference store
  // explicit pairs (rpa_align_pairs only; null in rpa_predict_batch, where a work-list entry is a row position and
  // the alignment kernels derive the segments from the row and the set's EmitInfo)
  uint32_t* pair_a;     // T2  segment id
  uint32_t* pair_b;     // T2
  float2* pair_res;     // P+S (distance, similarity); [res_slot[pos]] for the alignment asked for the hit at row pos, [P+s] for anchor<->query
  uint4* pair_ext;      // optional (rpa_align_pairs only): matches, mismatches, gaps, length; indexed like pair_res
  uint2* bin_list;      // kNumBins * T2 entries (segment A, segment B | kDstIsB)
  uint32_t* counters;   // kNumCounters words, layout below
  uint32_t ctr_live_in, ctr_live_out;  // kCtrActive / kCtrActiveB: the word that counts the live list a round reads / appends to
  // pass-1 scratch (per set region like the pair slots)
  float* band_save;     // S * 2*band_ranks: BandFactor accumulators of paused pass-1 loops
  uint32_t band_ranks;  // deepest taxonomy level + 1: rows of the LDS accumulators of k_decide
  // a launch over a list whose length only the device knows (k_myers_band / _band_duo / k_myers_long): the kernel works only if
  // dev_count_lo < length (and length <= dev_count_hi, when that is not 0) -- two kernels queued on one list, the right one runs
  uint32_t dev_count_lo, dev_count_hi;
  uint32_t big_min;     // record sets of big_min ... kBigSetMax records are ranked by k_rank_big_sets (a workgroup per set); 0: none
  // live lists: plan/decide append every set that is still running to live_out; a sparse round (few live sets) runs
  // emit/decide over live_in instead of over the whole batch
  const uint32_t* live_in;  // null in dense rounds
  uint32_t n_live_in;
  uint32_t* live_out;       // S
  uint32_t* og_idx;     // T
  int32_t* og_dist;     // T
  TraceEntry* trace;            // null: tracing off; else trace_cap entries, trace_n counts every attempt
  unsigned long long* trace_n;
  uint64_t trace_cap;
  uint32_t trace_set_base;      // first record set of this part inside the caller's batch
  const uint32_t* set_offsets;  // S+1 (copy of the batch's, for the alignment kernels)
  uint32_t H;    // n_records
  int row_quads; // Q: 16-byte quads per hit row, one of kRowQuadChoices, chosen per batch (row_quads_for)
  uint32_t lb_shift;  // log2 of the bits per path label: 3 (record sets of up to 256 records), 4 (up to 65536) or 5
  uint32_t T;    // number of segments
  uint32_t T2;   // number of pair slots = stride of each bin list
};
// counters: [0..kNumBins) pairs per bin | live sets | longest horizontal sequence among emitted pairs | pairs on the
// retry lists (banded alignments whose distance turned out to lie outside their band;
// (kCtrRetry: the strip kernel's, kCtrRetryShort: the one-strip kernels') | error code | error set/pair
// | retried pairs so far in this batch (long, short: added up by the retry launches; the host reads them a round later)
// | the live count of every other round.  The first kNumRoundCounters are reset every round -- by k_decide itself (its first
// thread; nobody else touches them while it runs), not by a fill between the launches: the list a round's k_decide appends to
// is counted in the OTHER live word (WorkDev::ctr_live_out) than the one the round's k_emit and the host read (ctr_live_in),
// and the two swap every round (round 4: a memset per round was ~10 us of a synchronous step's ~130 us per sparse round).
constexpr int kCtrActive = kNumBins, kCtrMaxLen = kNumBins + 1, kCtrRetry = kNumBins + 2, kCtrRetryShort = kNumBins + 3, kCtrErr = kNumBins + 4,
              kCtrErrSet = kNumBins + 5, kCtrRetryTotal = kNumBins + 6, kCtrRetryShortTotal = kNumBins + 7, kCtrActiveB = kNumBins + 8,
              kCtrDeferred = kNumBins + 9,   // pass-2 hits of this round that wait for the round's second launch (k_emit; see kDeferredMark)
              kCtrSurvivor = kNumBins + 10,  // [kNumBins words] pairs per bin of that second launch (k_defer_filter)
              kCtrWindowFail = 2 * kNumBins + 10,        // pairs of this round that left the ring-window kernel's band (length of its own list; cleared with kCtrSurvivor)
              kCtrRetryWindowTotal = 2 * kNumBins + 11,  // [kLadderBins words] pairs that left their class's banded kernel, added up over the batch, per ladder class
              kCtrSpreadSets = 2 * kNumBins + 11 + kLadderBins,  // record sets (nucleotide) whose scores spread by more than the cut-off factor (k_plan_sets)
              kNumCounters = 2 * kNumBins + 12 + kLadderBins, kNumRoundCounters = kNumBins + 4;
static_assert(kNumCounters <= 64, "launch_publish_words moves up to 64 words");

// Two launches inside ONE pass-2 round.  A pass-2 loop steps over every hit whose taxon lies at or below unode_global (:771), and
// unode_global is lifted by the hits in front of it (:811-813): of the hits a first pass-2 chunk asks for, half are never consumed
// because an earlier hit of the same chunk lifted unode_global over them (tools/spec_waste.py).  So k_emit holds back every hit
// that an EARLIER requested hit of its chunk could take out that way (the depth of its LCA with the best hit is not above the
// earlier one's) -- it leaves kDeferredMark in the hit's result slot instead of a list entry --, the round's first launch runs
// without them, k_defer_filter replays the chunk with the results that are there and lists the held-back hits the loop can still
// reach, and a second launch of the same round computes those.  Consumption (k_decide) is unchanged and exact: a hit that was
// dropped is one the reference's loop steps over; config 2 computes 1.03 alignments per demanded one instead of 1.13.
constexpr float kDeferredMark = -1.f;  // (a distance is never negative; k_decide treats a consumed mark as a failed assert)
constexpr uint32_t kDeferSpan = 64;    // only the first 64 rows of a chunk can be held back (the filter keeps them as one bit set)

// Row shapes the placement kernels are built for.  A row holds 4Q - 6 path words = (4Q - 6) * 32 / label bits levels.
constexpr int kRowQuadChoices[] = {2, 3, 4, 6, 8};
// label bits for a batch whose largest record set has `max_set` records (a label is a record's index inside its set)
__host__ __device__ __forceinline__ uint32_t label_shift_for(uint32_t max_set) { return max_set <= 256u ? 3u : max_set <= 65536u ? 4u : 5u; }
// smallest row shape that holds `levels` labels of (1 << lb_shift) bits; 0 = none does (levels x label bits > 832)
__host__ __device__ __forceinline__ int row_quads_for(uint32_t levels, uint32_t lb_shift) {
  const uint32_t words = ((((levels + 7u) & ~7u) << lb_shift) + 31u) / 32u;  // k_rank_records packs eight levels at a time
  for (int q : kRowQuadChoices)
    if ((uint32_t)(4 * q - 6) >= words) return q;
  return 0;
}

struct Params {
  int alphabet;
  float exclude_factor;   // exclude_alignments_factor_
  float reeval_factor;    // reeval_bandwidth_factor_ = (float)(1. - reeval_bandwidth)
  uint32_t spec1, spec2;  // speculation chunk (hits per round) of the pass-1 / pass-2 loops; 0 = the whole set at once
  int force_bin;          // >= 0: put every pair into this bin (experiments: RPA_FORCE_BIN); -1 normally
  int protein_lanes;      // 0: packed protein pairs all go to K3p (one lane per pair), none to K3l (A/B: RPA_PROTEIN_LANES=0)
  int protein_wide;       // 1: pairs beyond the packed cell go to K3p with the score-and-length cell where it holds them (the
                          //    placement path; A/B: RPA_PROTEIN_WIDE=0 keeps K3); 0 in rpa_align_pairs (match counts wanted)
  int no_combined;        // 1: never ask for pass 0 and the first pass-1 loop in one round (A/B: RPA_NO_COMBINED)
  int filter1;            // >= 1: a pass-1 chunk leaves out the hits that the loop's thresholds at the pause already exclude;
                          // 2 (default) / 3: and ends where the scores fall below the cut-off its first hit would set, for
                          //    protein / always
                          //    (chunk_end_by_score; A/B: RPA_PASS1_FILTER=0|1|3)
  int grow;               // 1: a loop that pauses asks for as many hits again as it has already walked (chunks 2,2,4,8,...):
                          //    a loop that has not stopped early so far probably will not; log2(n) rounds instead of n/chunk
  int defer2;             // 1: pass-2 hits that an earlier hit of their chunk may take out of the loop wait for the round's second launch
                          //    (kDeferredMark; A/B: RPA_DEFER2=0)
  uint32_t spec2_first;   // pass 2: the FIRST chunk of an anchor's loop (0: spec2).  The hits that lift unode_global come early
                          // in a loop and take most of the hits behind them out of it: a short first chunk, then spec2
};

struct ProfDev {  // device-side accumulators (u64): pairs, residue bytes, cells, demanded, cells updated (kProfSlots per part)
  unsigned long long* acc;
};

// ---- launchers (each enqueues on `st`, returns the hipError of the launch) ----
hipError_t launch_pack_letters(hipStream_t st, int alphabet, const uint8_t* letters, uint64_t n_residues, uint32_t* words);
// n <= 64 words from device memory into host-mapped pinned memory, by a kernel (not the copy engine); read them after a stream sync
hipError_t launch_publish_words(hipStream_t st, const uint32_t* src, uint32_t* dst_host_mapped, uint32_t n);
// extent table of a store: from an offsets table (n_seqs + 1 entries, sequence i at base + offsets[i]) for all sequences,
// or scattered for the sequences `index[k]` (k < n) of an append
hipError_t launch_store_extents(hipStream_t st, const uint64_t* offsets, const uint32_t* index, uint32_t n, uint64_t base, ulonglong2* extent);
// windowed stores: lengths of all sequences into extent[.].y (x = 0), win_count = 0
hipError_t launch_store_lengths(hipStream_t st, const uint64_t* lengths, uint32_t n, ulonglong2* extent, uint32_t* win_count);
// ... and the new window lists of the sequences index[k]: extent[index[k]].x = first[k], win_count[index[k]] = count[k]
hipError_t launch_store_window_lists(hipStream_t st, const uint32_t* index, const uint32_t* first, const uint32_t* count, uint32_t n,
                                     ulonglong2* extent, uint32_t* win_count);

// tiled row layout: tile_base[t] = first row slot of tile t (64 record sets, each padded to the tile's largest record
// count), tile_base[n_tiles] = total slots.  Two small launches; the caller reads the total back to size the buffers.
// max_set (zeroed by the caller) <- records of the largest record set; scan = false: only that (linear row layout)
hipError_t launch_tile_plan(hipStream_t st, const uint32_t* set_offsets, uint32_t n_sets, uint32_t* tile_base, uint32_t* max_set, bool scan);

// err[0..1] (zeroed by the caller) <- kErrInvalid + index when set_offsets / node indices of the batch are not usable
// rec_pre != null: also rec_pre[r] = pre[node[r]] for every record (pre = TaxDev::pre)
hipError_t launch_validate_batch(hipStream_t st, const BatchDev& b, uint32_t n_nodes, uint32_t* err, const uint32_t* pre = nullptr,
                                 uint32_t* rec_pre = nullptr);
hipError_t launch_rebase_offsets(hipStream_t st, const uint32_t* in, uint32_t n, uint32_t* out);
hipError_t launch_rank_records(hipStream_t st, const BatchDev& b, const WorkDev& w, const TaxDev& tax, const StoreDev& q,
                               const StoreDev& db, const Params& p);
hipError_t launch_plan_sets(hipStream_t st, const BatchDev& b, const WorkDev& w, const TaxDev& tax, const Params& p,
                            void* out_rows, void* out_stats);
hipError_t launch_emit(hipStream_t st, const BatchDev& b, const WorkDev& w, const TaxDev& tax, const Params& p);
hipError_t launch_list_big_sets(hipStream_t st, const uint32_t* set_offsets, uint32_t n_sets, uint32_t big_min, uint32_t* list, uint32_t* count);
hipError_t launch_rank_big_sets(hipStream_t st, const BatchDev& b, const WorkDev& w, const TaxDev& tax, const StoreDev& q, const StoreDev& db,
                                const Params& p, const uint32_t* big_list, uint32_t n_big, uint32_t largest_set);
hipError_t launch_decide(hipStream_t st, const BatchDev& b, const WorkDev& w, const TaxDev& tax, const Params& p,
                         void* out_rows, void* out_stats, unsigned long long* prof_acc);
// between the two launches of a round (kDeferredMark): the held-back pass-2 hits that are still needed -> bin lists, counters[kCtrSurvivor ..]
hipError_t launch_defer_filter(hipStream_t st, const BatchDev& b, const WorkDev& w, const Params& p);

// alignment-free models (lca_kernels.hip): model 0 dummy, 1 simple-lca, 2 megan-lca, 3 n-best-lca
struct LcaParamsDev {
  int model;
  float toppercent, minscore, maxevalue;  // floats in the reference's filter (alignmentsfilter.hh:351)
  uint32_t minsupport, nbest;
  int ignore_unclassified;
};
hipError_t launch_lca_models(hipStream_t st, const BatchDev& b, const TaxDev& tax, const LcaParamsDev& p, const double* evalue,
                             const uint8_t* unclassified, void* rows);

// pair-level entry (rpa_align_pairs): segments straight from an arena-shaped store, explicit pair list
hipError_t launch_plan_plain_segments(hipStream_t st, int alphabet, const WorkDev& w, const StoreDev& store);
hipError_t launch_bin_pairs(hipStream_t st, const Params& p, const WorkDev& w, uint32_t n_pairs);
hipError_t launch_export_alignments(hipStream_t st, const WorkDev& w, uint32_t n_pairs, void* out);

// alignment kernels over one bin list
// walks `list` then `list2` (n2 may be 0): pairs of a narrower class may ride along, the kernel is exact for any pattern <= 32*words
// n_dev != null: `list` is a retry list whose length is read on the device (bounded by n); grid_blocks workgroups stride over it
hipError_t launch_align_dna_short(hipStream_t st, int words, const uint2* list, uint32_t n, const uint32_t* n_dev, uint32_t grid_blocks,
                                  const uint2* list2, uint32_t n2, const WorkDev& w, unsigned long long* prof_acc, uint32_t dev_split = 0);
// K1w (a pattern word per lane: launches of few pairs); with n_dev: works only when the list holds up to dev_split pairs
hipError_t launch_align_dna_short_wave(hipStream_t st, const uint2* list, uint32_t n, const uint32_t* n_dev, uint32_t dev_split,
                                       const uint2* list2, uint32_t n2, const WorkDev& w, unsigned long long* prof_acc);
// the same inside a band of 2 words either side of the diagonal (patterns of 5..8 words); pairs whose distance lies
// outside it are appended to retry[*retry_n] instead of being answered
bool dna_short_band_available(int words);
hipError_t launch_align_dna_short_band(hipStream_t st, int words, const uint2* list, uint32_t n, const uint2* list2, uint32_t n2,
                                       const WorkDev& w, uint2* retry, uint32_t* retry_n, bool shared_gpu, bool wide, unsigned long long* prof_acc,
                                       uint32_t* retry_total = nullptr);
// multi-strip per-lane kernel inside Ukkonen's band; scratch holds two boundary bits per text column per thread in
// flight.  A pair whose distance lies outside its band is appended to retry[*retry_n] (wide_open: never happens).
// margin / 16 = safety factor on the probe's distance estimate (21 by default; tests use tiny ones to force retries).
// n_dev != null: the length of `list` is read on the device (bounded by n) and grid_blocks workgroups stride over it.
size_t dna_strip_scratch_words(uint32_t n_pairs, uint32_t max_len);
bool dna_window_available(int window_words);
hipError_t launch_align_dna_window(hipStream_t st, int window_words, const uint2* list, uint32_t n, const WorkDev& w, uint2* retry,
                                   uint32_t* retry_n, uint32_t* retry_total, unsigned long long* prof_acc);
hipError_t launch_align_dna_strips(hipStream_t st, int strip_words, const uint2* list, uint32_t n, const uint32_t* n_dev,
                                   uint32_t grid_blocks, const WorkDev& w, uint32_t* scratch, size_t scratch_words, uint32_t max_len,
                                   bool wide_open, uint32_t margin, uint2* retry, uint32_t* retry_n, unsigned long long* prof_acc);
hipError_t launch_align_dna_long(hipStream_t st, int words_per_lane, const uint2* list, uint32_t n, const uint32_t* n_dev,
                                 uint32_t grid_blocks, const WorkDev& w, unsigned long long* prof_acc);
hipError_t launch_align_protein(hipStream_t st, int strip_rows, const uint2* list, uint32_t n, const WorkDev& w, uint32_t* scratch,
                                size_t scratch_words, uint32_t max_len, unsigned long long* prof_acc);
hipError_t launch_align_protein_packed(hipStream_t st, const uint2* list, uint32_t n, const WorkDev& w, uint32_t* scratch,
                                       size_t scratch_words, uint32_t max_len, uint32_t* work_counter, unsigned long long* prof_acc, bool wide = false);
hipError_t launch_align_protein_lanes(hipStream_t st, int lanes, const uint2* list, uint32_t n, const WorkDev& w, uint32_t* work_counter,
                                      unsigned long long* prof_acc, bool wide = false);
size_t protein_scratch_words(uint32_t n_pairs, uint32_t max_len);

// Work-list entry -> (segment A, segment B, result index).  Order matters for protein (A is the horizontal sequence):
// getAlignment(segments[i], qrseq) (:526), getAlignment(segments[i], segments[index_anchor]) (:630,778),
// getAlignment(segments[index_anchor], qrseq) (:797).  k_emit writes the two segment ids into the entry, so an
// alignment lane is two dependent loads away from its residues (entry -> descriptors -> store words).
// Results of hit <-> x alignments are indexed by the hit's row position (= segment id A); the anchor <-> query
// alignment of a set goes to slot H + s, flagged by the top bit of B.
constexpr uint32_t kDstIsB = 0x80000000u;
__device__ __forceinline__ void resolve_pair(const WorkDev& w, uint2 entry, uint32_t& seg_a, uint32_t& seg_b, uint32_t& dst) {
  if (w.pair_a) { seg_a = w.pair_a[entry.x]; seg_b = w.pair_b[entry.x]; dst = entry.x; return; }  // rpa_align_pairs
  seg_a = entry.x;
  seg_b = entry.y & ~kDstIsB;
  dst = (entry.y & kDstIsB) ? seg_b : seg_a;
}

// where an alignment kernel stores the result of destination id `dst` (a row position, or H + s for anchor <-> query)
__device__ __forceinline__ uint32_t res_index(const WorkDev& w, uint32_t dst) {
  if (!w.res_slot) return dst;  // rpa_align_pairs: one slot per pair; linear row layout: slot = row position and P = H
  return dst < w.H ? w.res_slot[dst] : w.P + (dst - w.H);
}

// protein: bin = strip height index (R = 8 * (bin + 1)).  One height for every pair: R = 24.  Measured on MI355X at
// 67 aa (tools/ab_protein.sh, same device): R = 16 and 24 tie, R = 32 is 17 % slower (172 VGPRs -> 2 waves per SIMD),
// R = 8 is 50 % slower (per-column overhead); choosing 16 or 24 per pair costs 7 % more than either alone (two
// half-filled launches per round).  RPA_FORCE_BIN selects another height for experiments.
// Bin 4: the pair fits the packed-cell kernel (K3p, align_protein_kernels.hip): its two 9-bit path counters hold up to
// 511 steps and its 12-bit score field holds la + lb + 11 * min(la, lb) + 32 (row-relative score + bias, BLOSUM62's
// largest entry is 11) -- 312 x 312 residues, or e.g. 511 x 100.
// Bins 5..17: a packed pair whose vertical sequence needs up to 13 strips of kProteinLaneRows rows (lb <= 312; every pair
// of the 67 aa workload: three) goes to K3l, the same cell with the strips of a pair in adjacent lanes: one class per
// strip count (bin = 4 + strips), 64 / strips pairs per wavefront.  (Until round 3 seven classes -- 1, 2, 3, 4, 6, 8, 13
// lanes -- shared twelve bins with the nucleotide kernels: a 200 aa pair, nine strips, ran in the 13-lane class with four
// pairs per wavefront instead of seven.)
constexpr uint32_t kPackedMaxLen = 511;
constexpr uint32_t kWideMaxLen = 16383;  // the score-and-length cell (bin 3): 16-bit step counter, 14-bit score field
constexpr int kProteinBins = 22;
constexpr int kProteinPackedBin = 4;
// Bin 3 (until round 4: K3 with 32-row strips, an experiment's height): pairs beyond the packed cell whose score field fits 14
// bits -- la + lb + 11 min(la, lb) + 32 <= 16383, e.g. 1258 x 1258 -- when only distance and similarity are wanted (the
// placement path; rpa_align_pairs also reports matches / mismatches and keeps K3): K3p with the score-and-length cell.
constexpr int kProteinWideBin = 3;
#ifndef RPA_LANE_ROWS
#define RPA_LANE_ROWS 24
#endif
constexpr int kProteinLaneRows = RPA_LANE_ROWS;
constexpr int kProteinLaneClasses = 13;
constexpr int kProteinMaxLanes = 13;
__host__ __device__ __forceinline__ int protein_class_lanes(int cls) { return cls + 1; }  // lanes per pair of a class
// Bins 18..21 (round 5): K3l with the SCORE-AND-LENGTH cell (kWideMaxLen) for pairs whose vertical sequence needs 14 ... 64 strips
// (313 ... 1536 residues; the placement path only: no match counter) -- 16, 21, 32 or 64 lanes per pair (4, 3, 2, 1 pairs per
// wavefront), a pair takes the narrowest class that holds its strips.  Until then such pairs ran one lane per pair with every strip
// boundary through an HBM scratch (K3p wide, bin 3): 450 aa segments at 0.43 of the cell rate K3l reaches at 200 aa.
constexpr int kProteinWideLaneBin = 18, kProteinWideLaneClasses = 4;
__host__ __device__ __forceinline__ int protein_wide_class_lanes(int cls) { return cls == 0 ? 16 : cls == 1 ? 21 : cls == 2 ? 32 : 64; }
__host__ __device__ __forceinline__ bool protein_fits_packed(uint32_t la, uint32_t lb) {
  return la <= kPackedMaxLen && lb <= kPackedMaxLen && la + lb + 11u * (la < lb ? la : lb) + 32u <= 4095u;
}
__host__ __device__ __forceinline__ bool protein_fits_wide(uint32_t la, uint32_t lb) {
  return la <= kWideMaxLen && lb <= kWideMaxLen && la + lb + 11u * (la < lb ? la : lb) + 32u <= 16383u;
}
__host__ __device__ __forceinline__ int protein_bin(uint32_t la, uint32_t lb, bool lanes = true, bool wide = false) {
  const uint32_t strips = (lb + kProteinLaneRows - 1u) / kProteinLaneRows;
  const bool packed = protein_fits_packed(la, lb);
  if (packed && lanes && strips <= (uint32_t)kProteinMaxLanes) {
    const int cls = strips ? (int)strips - 1 : 0;
    return kProteinPackedBin + 1 + cls;
  }
  if (wide && lanes && strips <= 64u && (packed || protein_fits_wide(la, lb)))  // more strips than a packed class has lanes for: the wide-cell lanes classes
    return kProteinWideLaneBin + (strips <= 16u ? 0 : strips <= 21u ? 1 : strips <= 32u ? 2 : 3);
  if (packed) return kProteinPackedBin;
  return (wide && protein_fits_wide(la, lb)) ? kProteinWideBin : 2;
}
static_assert(kProteinPackedBin + 1 + kProteinLaneClasses <= kProteinWideLaneBin && kProteinWideLaneBin + kProteinWideLaneClasses <= kProteinBins &&
              kProteinBins <= kNumBins && kDnaBins <= kNumBins, "bins");

// ---- reading a segment out of the packed store --------------------------------------------------------------
__device__ __forceinline__ uint32_t revcomp8(uint32_t x) {  // 8 packed nt: reverse the nibble order, A<->T, C<->G, N stays
  x = __builtin_bswap32(x);
  x = ((x & 0x0F0F0F0Fu) << 4) | ((x >> 4) & 0x0F0F0F0Fu);
  const uint32_t lt4 = (~x >> 2) & 0x11111111u;  // 1 per nibble whose code is < 4
  return x ^ (lt4 * 3u);
}

struct SegView {  // a segment as the kernels read it
  const uint32_t* words;
  uint64_t src;
  uint32_t len;
  bool rc;
  // 8 nucleotides (one dword, 4 bit each) starting at residue p of the segment; nucleotides past the end are unspecified.
  // Forward: two adjacent store words funnel-shifted (v_alignbit).  Reverse strand (seqan::reverseComplement,
  // core/src/sequencestorage.hh:364-369): the 8 source nucleotides that END at src+len-1-p, nibble-reversed, complemented.
  __device__ __forceinline__ uint32_t fetch8(uint32_t p) const {
    if (!rc) {
      const uint64_t g = src + p;
      return __builtin_amdgcn_alignbit(words[g / 8 + 1], words[g / 8], (uint32_t)(g % 8) * 4u);  // stores end in slack words
    }
    const int64_t gs = (int64_t)(src + len - 1 - p) - 7;
    uint32_t x;
    if (gs >= 0) x = __builtin_amdgcn_alignbit(words[(uint64_t)gs / 8 + 1], words[(uint64_t)gs / 8], (uint32_t)((uint64_t)gs % 8) * 4u);
    else x = gs > -8 ? words[0] << (uint32_t)(-gs * 4) : 0u;  // within 7 nt of the very start of the store
    return revcomp8(x);
  }
  __device__ __forceinline__ uint4 fetch32(uint32_t p) const { return make_uint4(fetch8(p), fetch8(p + 8), fetch8(p + 16), fetch8(p + 24)); }
  // protein (8 bit per residue, forward only)
  __device__ __forceinline__ uint32_t byte_at(uint32_t p) const { return reinterpret_cast<const uint8_t*>(words)[src + p]; }
};

// A segment read front to back, 8 or 32 nucleotides per step, in SEGMENT order (reverse-strand segments come out
// reversed and complemented).  Consecutive steps share a store word, so a step costs one new 4-byte or one 16-byte load
// plus funnel shifts.  Why 16 bytes where possible: every lane streams its own sequence, i.e. 64 different cache lines
// per wave-level load; with 16 resident waves per CU a lane's line has left the 32 KB L1 again before the lane's next
// load, so each load is an L2 access that moves a whole sector for the few bytes used.  Reads run up to 4 store words
// past either end of the segment: stores carry zeroed slack words in front and behind.
struct NtStream {
  const uint32_t* words;
  int64_t next_i;   // store word the next step consumes first
  uint32_t keep;    // the word shared with the previous step
  uint32_t sh;      // funnel shift in bits
  int32_t dir;      // +1 forward, -1 reverse strand
  bool rc;
  // start at segment position `first` (text position p of a reverse segment is store nibble src+len-1-p)
  __device__ __forceinline__ void init(const SegView& v, uint32_t first) {
    words = v.words; rc = v.rc; dir = v.rc ? -1 : 1;
    const int64_t g0 = v.rc ? (int64_t)(v.src + v.len) - 8 - (int64_t)first : (int64_t)(v.src + first);
    sh = (uint32_t)(g0 & 7) * 4u;
    const int64_t idx = g0 >> 3;  // floor: may be -1 at the very start of the store
    keep = words[v.rc ? idx + 1 : idx];
    next_i = v.rc ? idx : idx + 1;
  }
  __device__ __forceinline__ uint32_t dword_of(uint32_t fresh) {
    const uint32_t hi = rc ? keep : fresh, lo = rc ? fresh : keep;
    const uint32_t d = __builtin_amdgcn_alignbit(hi, lo, sh);
    keep = fresh;
    const uint32_t r = revcomp8(d);
    return rc ? r : d;
  }
  __device__ __forceinline__ uint32_t next8() {  // one 4-byte load
    const uint32_t fresh = words[next_i];
    next_i += dir;
    return dword_of(fresh);
  }
  __device__ __forceinline__ uint4 load4() const {  // the four store words the next next32 consumes, in store order
    const uint32_t* p4 = words + (rc ? next_i - 3 : next_i);
    return make_uint4(p4[0], p4[1], p4[2], p4[3]);
  }
  __device__ __forceinline__ uint4 take32(const uint4 v) {  // v = load4() of this position (possibly issued a step ahead)
    next_i += 4 * dir;
    const uint32_t w0 = rc ? v.w : v.x, w1 = rc ? v.z : v.y, w2 = rc ? v.y : v.z, w3 = rc ? v.x : v.w;
    uint4 o;
    o.x = dword_of(w0); o.y = dword_of(w1); o.z = dword_of(w2); o.w = dword_of(w3);
    return o;
  }
  __device__ __forceinline__ uint4 next32() { return take32(load4()); }
};

__device__ __forceinline__ SegView seg_view(const WorkDev& w, uint32_t seg) {
  const uint4 d = reinterpret_cast<const uint4*>(w.seg)[seg];
  SegView v;
  v.words = (d.w & 2u) ? w.store_q : w.store_db;
  v.src = ((uint64_t)d.y << 32) | d.x;
  v.len = d.z;
  v.rc = d.w & 1u;
  return v;
}

// bin of a pair whose shorter sequence has m residues (DNA).  Up to kShortWords words: one strip of exactly that many words.
// 13 ... 64 words: three classes for the ring-window kernel.  Beyond: the strip height in 5..8 words with the smallest
// strips x (height x 16 + 20) issue-slot estimate, i.e. the fewest wasted rows (65 words -> 9 x 8, 71 -> 9 x 8 ...); any length fits.
__host__ __device__ __forceinline__ int dna_bin(uint32_t m) {
  const uint32_t words = (m + 31u) / 32u;
  if (words <= (uint32_t)kShortWords) return words ? (int)words - 1 : 0;
  if (words <= 16u) return kShortWords;
  if (words <= 32u) return kShortWords + 1;
  if (words <= (uint32_t)kWindowMaxWords) return kShortWords + 2;
  uint32_t best = 0xFFFFFFFFu;
  int best_nw = kStripMaxWords;
  for (int nw = kStripMaxWords; nw >= kStripMinWords; --nw) {  // ties -> taller strips (fewer sweeps of the text)
    const uint32_t strips = (words + (uint32_t)nw - 1u) / (uint32_t)nw;
    const uint32_t cost = strips > 0x00FFFFFFu ? 0xFFFFFFFEu : strips * ((uint32_t)nw * 16u + 20u);
    if (cost < best) { best = cost; best_nw = nw; }
  }
  return kFirstStripBin + best_nw - kStripMinWords;
}
// strip height (words) of the register-strip kernels for a pair of bin `bin` (>= kShortWords); the window classes fall back to the tallest
__host__ __device__ __forceinline__ int strip_words_of_bin(int bin) { return bin < kFirstStripBin ? kStripMaxWords : bin - kFirstStripBin + kStripMinWords; }

}  // namespace rpa
