// taxator-tk_amd/csrc/rpa_engine.hip -- host side of the C ABI (include/taxator_rpa.h) for the gfx950 RPA engine.
//
// One engine = one GPU = one RPAPredictionModel instance of the reference (core/src/taxonpredictionmodelsequence.hh:326-339)
// plus its two sequence stores (core/taxator.cpp:230-249), resident in HBM.  rpa_predict_batch drives the rounds
//   rank -> plan -> { emit -> align bins -> decide }*      (kernels: place_/align_*_kernels.hip)
// on one HIP stream per PART.  The host reads one small counter block per round (pairs per bin, live sets, error word).
// A large batch is cut into a few parts of consecutive record sets, each with its own working set, stream and host
// thread, started one after the other: the alignment kernel is bound by VALU issue and leaves half of every CU's wave
// slots, registers and LDS free, the placement kernels are bound by memory latency, so the placement rounds of one part
// run underneath the alignment kernel of another (same results: record sets are independent).
// There is no CPU compute path: every entry point fails with RPA_ERR_DEVICE when no gfx950 device answers.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "../../include/taxator_rpa.h"
#include "rpa_device.h"

using namespace rpa;

namespace {

thread_local std::string g_create_error;

// Diagnostics for watchdogs (rpa_debug_where): the engine-side blocking call entered last, process-wide.  A relaxed
// store of a string literal's address per synchronisation point; never read by the engine itself.
std::atomic<const char*> g_where{"idle"};
#define RPA_WHERE(text) g_where.store(text, std::memory_order_relaxed)

struct DevBuf {
  void* p = nullptr;
  size_t cap = 0;
  hipError_t ensure(size_t bytes) {
    if (bytes <= cap) return hipSuccess;
    if (p) { hipError_t e = hipFree(p); p = nullptr; cap = 0; if (e != hipSuccess) return e; }
    size_t want = bytes + bytes / 8 + 256;  // a little slack so a slightly bigger next batch does not reallocate
    hipError_t e = hipMalloc(&p, want);
    if (e != hipSuccess) { p = nullptr; return e; }
    cap = want;
    return hipSuccess;
  }
  void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
  template <typename T> T* as() const { return reinterpret_cast<T*>(p); }
};

constexpr size_t kLeadWords = 16, kTailWords = 64;  // slack around the packed residues (16-byte text loads, funnel shifts)

struct Store {
  DevBuf words, extent, tmp_offsets, tmp_index;
  uint32_t n_seqs = 0, n_resident = 0;
  uint64_t n_residues = 0;   // residues of the arena in use (appends start at multiples of 8 residues)
  uint64_t cap_words = 0;    // packed words the arena holds (without the slack in front and behind)
  bool loaded = false;
  // windowed stores (rpa_store_reserve_windows): which parts of which sequence are resident.  The host keeps every
  // sequence's window list (sorted, disjoint); the device sees them through an append-only pool of window records -- a
  // sequence whose list changes gets a fresh copy of it at the end of the pool and its (first entry, count) updated.
  bool windowed = false;
  std::vector<uint64_t> seq_length;
  std::vector<std::vector<uint4>> seq_windows;  // (first residue, residues, arena offset low, high)
  DevBuf win_pool, win_count, tmp_a, tmp_b;
  uint64_t pool_used = 0, pool_cap = 0;   // window records
  uint64_t n_windows = 0, live_residues = 0;
  // the packed words sit behind kLeadWords zero words: a reverse-strand reader may fetch a few words below its segment
  StoreDev view() const {
    return StoreDev{words.as<const uint32_t>() + kLeadWords, extent.as<const ulonglong2>(), n_seqs,
                    windowed ? win_pool.as<const uint4>() : nullptr, windowed ? win_count.as<const uint32_t>() : nullptr};
  }
  void release() {
    words.release(); extent.release(); tmp_offsets.release(); tmp_index.release(); win_pool.release(); win_count.release(); tmp_a.release(); tmp_b.release();
    n_seqs = n_resident = 0; n_residues = cap_words = 0; loaded = false; windowed = false;
    seq_length.clear(); seq_windows.clear(); pool_used = pool_cap = 0; n_windows = live_residues = 0;
  }
};

}  // namespace

struct rpa_engine {
  int device = 0;
  hipStream_t own_stream = nullptr, stream = nullptr;
  Params params{};
  DevBuf tax_parent, tax_left, tax_right, tax_depth, tax_pre, tax_last, tax_anc, tax_anc_iv;
  TaxDev tax{};
  Store stores[2];
  Store pair_store;  // rpa_align_pairs
  std::string err;
  // batch staging (host batches) and working set
  DevBuf in_u32[11], in_f32, in_u8, in_f64, in_uncl;
  DevBuf state, loop_save, rec_pre, hits, set_info, emit, emit_lite, emit_ext, hit_set, block_first, flags, lite, tile_base, res_slot, seg, pair_a, pair_b,
      pair_res, pair_ext, bin_list, counters, og_idx, og_dist, band_save, live_a, live_b, out_rows, out_stats, tmp_letters,
      protein_ws, lanes_counter, retry_list, retry_short, window_fail, big_sets;
  uint32_t* h_counters = nullptr;  // pinned
  // The alignment launches of ONE round that do not depend on each other (one per length class: the strip-height classes of long
  // nucleotide pairs, the lanes-per-pair classes of protein pairs) go to side streams, forked from and joined to the engine's stream
  // by events: on batches of mixed segment lengths most of them fill a fraction of the chip and last as long as their longest pair
  // (tools/ragged_bench.py: 43 launches per step one after the other).  Uniform batches have one launch per round and never fork.
  static constexpr int kSideStreams = 4;
  hipStream_t side_stream[kSideStreams] = {nullptr, nullptr, nullptr, nullptr};
  hipEvent_t ev_fork = nullptr, ev_join[kSideStreams] = {nullptr, nullptr, nullptr, nullptr};
  DevBuf side_ws[kSideStreams];  // the strip kernel's boundary-plane scratch, one per side stream
  bool rank_only = false;        // RPA_DEBUG_RANK_ONLY=1
  bool debug_bins = false;       // RPA_DEBUG_BINS=1: one stderr line per alignment launch group (bin, pairs), to read beside a kernel trace
  bool side_streams_on = true;   // RPA_SIDE_STREAMS=0: every launch on the engine's stream (A/B)
  bool align_low_priority = false;  // RPA_ALIGN_PRIORITY=1: a job slot's alignment launches on a stream of the lowest priority (run_align_bins)
  hipStream_t lo_stream = nullptr;
  hipEvent_t ev_lo_begin = nullptr, ev_lo_end = nullptr;
  int defer_mode = 2;            // RPA_DEFER2=0 | 1 | auto: two launches per pass-2 round never | always | where alignments are expensive (predict_part)
  // profiling
  bool profiling = false;
  bool prof_counters = true;   // rpa_profile_enable(2): events only -- the kernels carry no counters (their reductions cost K1 4 %)
  int long_kernel = 0;        // RPA_LONG_KERNEL=wave|lane: force K2 / K1s for patterns > 256 nt (tests, A/B); 0 = cost model
  uint32_t counter_next = 0;  // next free word of the work-counter pool (lanes_counter), see next_work_counter
  uint32_t short_wave_max = 16384;  // one-strip launches of up to this many pairs run on K1w (RPA_SHORT_WAVE_MAX; 0: never; tools/short_wave_bench.py)
  bool trace_rounds = false;  // RPA_TRACE_ROUNDS=1: print live sets / alignments per round to stderr
  bool linear_rows = false;   // RPA_ROW_LAYOUT=linear: never tile the hit rows (tests, A/B)
  // alignment trace (rpa_trace_begin / rpa_trace_read): synchronous calls on the owner engine only
  DevBuf trace_buf, trace_count;
  uint64_t trace_cap = 0;
  TraceEntry* trace_view = nullptr;           // what this engine's kernels write to (parts of a batch: the owner's buffer)
  unsigned long long* trace_view_n = nullptr;
  uint64_t trace_view_cap = 0;
  uint32_t trace_set_base = 0;
  uint32_t force_label_shift = 0;  // RPA_LABEL_BITS=16|32: wider taxon path labels than the batch needs (tests: the wide row shapes)
  bool poison = false;        // RPA_POISON=1: every working buffer is filled with 0xA5 before each batch (tests: a kernel that
                              // reads a slot nobody wrote this batch then reads garbage every time, not whatever the last batch left)
  // what the bands cost: pairs sent through the banded kernels / re-run wide open, per batch.  A band that sends more
  // than a quarter of its pairs to the retry list (unrelated sequences) is switched off for the next 16 batches.
  // ladder classes (rpa_device.h: kLadderBins; index = bin - kWaveMaxWords): 0..3 the one-strip classes of 9..12 words, then the ring-window classes
  uint64_t banded_short_batch = 0, banded_long_batch = 0, banded_window_batch[kLadderBins] = {};
  int band_short_pause = 0, band_long_pause = 0, band_window_pause[kLadderBins] = {};  // (per length class: 500 nt reads may stay inside their window while 1500 nt contigs do not)
  int band_window_wide[kLadderBins] = {};  // > 0: the class runs the second instance of its ladder for that many more batches (-1: always, RPA_BAND_WINDOW_WIDE=1)
  uint64_t window_seen_pairs[kLadderBins] = {}, window_seen_left[kLadderBins] = {};  // what the current instance's share of the batch's counters starts from
  bool band_window = true;    // RPA_BAND_WINDOW=0: patterns of 13..64 words go straight to the strip kernels (A/B; the same results)
  uint32_t big_set_min = 192;  // (100-record sets: 13.9 ms per step with their own workgroups -- 100 of 256 lanes at work -- against 10.7 without) record sets from this many records on (up to kBigSetMax) are ranked a workgroup per set (RPA_BIG_SET_MIN; 0: never)
  uint32_t window_fail_wave_max = 8192;  // up to this many pairs that left the ring window: one pair per wavefront (RPA_WINDOW_FAIL_WAVE_MAX; 0: always the strips)
  uint32_t band_window_min = 16384;  // fewer pairs than this in a launch: strips / one pair per wavefront as before (RPA_BAND_WINDOW_MIN; mixed-shape stream, in flight / synchronous: 65536: 7.02 / 10.58 ms per step, 16384: 6.35 / 10.57, 4096: 6.49 / 10.63, 512: 6.27 / 10.86)
  int band4_wide = 0;        // > 0: the one-strip band with four words all the way (k = 96 - |delta|) for this many more batches:
                             //      too many pairs of a batch left the narrow one (k = 80 - |delta|); -1 / -2: fixed by RPA_BAND4_PHASES=2 / 3
  bool band_short = true;     // RPA_BAND_SHORT=0: the one-strip kernel computes every cell of 5..8-word patterns too
  uint32_t band_short_min = 65536;  // fewer pairs than this in a launch: not worth a second (retry) launch
  bool band_long = true;      // RPA_BAND=0: the strip kernel computes every cell (A/B; results are the same either way)
  uint32_t band_margin = 21;  // x/16: safety factor on the probe's distance estimate (RPA_BAND_MARGIN; tests force retries with 1)
  double band_gain = 2.5;     // K1s-vs-K2 cost model: what the band is worth on K1s (RPA_BAND_GAIN)
  // a round goes through the live list once live sets x this <= all sets (tools/sparse_sweep.sh: decide from 1/6 on is
  // 5 % of a config-4 step, neutral on config 2; emit only pays from 1/24 on)
  uint32_t sparse_div_emit = 24, sparse_div_decide = 6;
  uint32_t fill_lanes = 0;  // pairs a round should at least ask for when few record sets are left (set in rpa_create)
  bool fixed_speculation = false;  // the caller (or RPA_SPEC1/2) fixed the chunk sizes: keep them in the sparse rounds too
  DevBuf prof_acc;
  rpa_profile prof{};
  std::vector<hipEvent_t> ev_pool;
  size_t ev_used = 0;
  std::vector<std::pair<size_t, size_t>> ev_align;  // (start, stop) indices
  hipEvent_t ev_total0 = nullptr, ev_total1 = nullptr;
  std::mutex mu;  // submit is serialised per engine (SURVEY 8b threading)
  // parts of a batch beyond the first run on worker engines: own stream, working set and counters, the owner's
  // taxonomy / parameters (views, not owned)
  bool is_worker = false;
  std::vector<rpa_engine*> workers;     // owner only
  DevBuf part_offsets;                  // set offsets of this part, rebased to 0
  hipEvent_t ev_inputs = nullptr;       // owner: everything queued on the owner's stream before the parts start
  int parts_override = 0;               // RPA_PARTS=n: cut every batch into n parts (tests, A/B); 0 = by size
  bool part_mode = false;               // running as one of several parts: the call's device time is taken by the owner
  // asynchronous jobs (rpa_submit_batch / rpa_wait): each slot is a worker engine (own stream, working set, staging) with
  // a host thread of its own; RPA_JOBS_IN_FLIGHT slots = that many batches in flight, the copies of one under the kernels of the others
  rpa_engine* owner = nullptr;          // slots: whose stores and taxonomy they read
  std::vector<rpa_engine*> slots;       // owner only
  std::thread job_thread;               // slot only
  std::mutex job_mu;
  std::condition_variable job_cv;
  struct rpa_job* job_pending = nullptr;
  bool job_busy = false, job_quit = false;
  unsigned next_slot = 0;               // owner: jobs go to the slots in turn
  unsigned peak_in_flight = 0;          // owner: most jobs the caller has had in flight at once (rpa_submit_batch: the ring of slots is one larger)
  std::mutex copy_mu;                   // owner: one job copies its batch to the device at a time (see predict_on)
};

struct rpa_job {
  rpa_batch batch;
  rpa_prediction* out = nullptr;
  rpa_stats* stats = nullptr;
  int status = RPA_OK;
  bool done = false;
  std::string err;
  rpa_engine* slot = nullptr;
};

namespace {

int fail(rpa_engine* e, int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  if (e) e->err = buf; else g_create_error = buf;
  return code;
}

#define HIP_TRY(e, call)                                                                                      \
  do {                                                                                                        \
    hipError_t _err = (call);                                                                                 \
    if (_err != hipSuccess) return fail(e, RPA_ERR_DEVICE, "%s failed: %s", #call, hipGetErrorString(_err)); \
  } while (0)

hipEvent_t get_event(rpa_engine* e) {
  if (e->ev_used == e->ev_pool.size()) {
    hipEvent_t ev;
    if (hipEventCreate(&ev) != hipSuccess) return nullptr;
    e->ev_pool.push_back(ev);
  }
  return e->ev_pool[e->ev_used++];
}

const char* kernel_error_text(uint32_t code) {
  switch (code) {
    case kErrRange: return "sequence range error: segment start lies beyond the end of the sequence (SequenceRangeError)";
    case kErrSeqId: return "sequence index outside the store (SequenceNotFound)";
    case kErrAssert: return "a reference assert would abort here (reverse reference coordinates in protein mode, empty best-reference group or inconsistent node range)";
    case kErrUnsupported: return "sequence pair not supported by this build";
    case kErrNotResident: return "a segment reaches outside the resident windows of its sequence (rpa_store_append_windows must cover every region a batch refers to)";
    case kErrNanScore: return "a score that is not a number in a record set of two or more unmasked alignments: the reference sorts such a set with a comparator that is no ordering (core/src/alignmentrecord.hh:89-93) and what it then does is an accident of the sort";
    default: return "inconsistent batch";
  }
}

// Launch the alignment kernels for every non-empty bin list (counts come from the host copy of the counters).
// A one-strip class (pattern <= 256 nt) with fewer pairs than this does not get a launch of its own when a wider
// one-strip class also has work: its list is appended to that launch (a sparse launch costs ~60 us whatever its size,
// and the wider kernel is exact for the narrower patterns at the price of the extra rows).
constexpr uint32_t kRideAlongPairs = 262144;
constexpr uint32_t kResidentLanes = 256u * 4u * 256u;  // one-pair-per-lane kernels: 256 CUs x 4 workgroups (LDS) x 256 lanes

// Timed launch bracket for the profile (HIP events on the engine's stream).
struct LaunchTimer {
  rpa_engine* e; size_t i0 = 0, i1 = 0; bool on = false;
  int begin() {
    if (!e->profiling) return RPA_OK;
    hipEvent_t a = get_event(e), b = get_event(e);
    if (!a || !b) return fail(e, RPA_ERR_DEVICE, "hipEventCreate failed");
    i0 = e->ev_used - 2; i1 = e->ev_used - 1; on = true;
    HIP_TRY(e, hipEventRecord(a, e->stream));
    return RPA_OK;
  }
  int end() {
    if (!on) return RPA_OK;
    HIP_TRY(e, hipEventRecord(e->ev_pool[i1], e->stream));
    e->ev_align.emplace_back(i0, i1);
    e->prof.launches += 1;
    on = false;
    return RPA_OK;
  }
};

// A zero word for the work counter of a persistent protein launch (K3p, K3l): the pool is cleared once per batch and every
// launch takes the next word -- until round 4 a fill in front of every launch, i.e. a second packet and a gap on the
// stream per launch (the nucleotide rounds lost theirs the same way: rpa_device.h, kNumRoundCounters).
constexpr uint32_t kCounterPoolWords = 4096;
int counter_pool_reset(rpa_engine* e) {
  HIP_TRY(e, e->lanes_counter.ensure((size_t)kCounterPoolWords * 4));
  HIP_TRY(e, hipMemsetAsync(e->lanes_counter.p, 0, (size_t)kCounterPoolWords * 4, e->stream));
  e->counter_next = 0;
  return RPA_OK;
}
int next_work_counter(rpa_engine* e, uint32_t** out) {
  if (!e->lanes_counter.p || e->counter_next >= kCounterPoolWords) {  // (a batch of more launches than words: clear behind the ones in flight on this stream)
    const int rc = counter_pool_reset(e);
    if (rc) return rc;
  }
  *out = e->lanes_counter.as<uint32_t>() + e->counter_next++;
  return RPA_OK;
}

int run_align_bins_on_stream(rpa_engine* e, const WorkDev& w, const uint32_t* counts_in);

// The alignment launches of a JOB SLOT go to a stream of the lowest priority, everything else of the job (rank, plan, emit, decide:
// short kernels that wait on memory) stays on the slot's own stream: with several batches in flight the placement kernels of one
// batch are then dispatched ahead of the next workgroups of another batch's long alignment launch instead of queueing behind them
// (RPA_ALIGN_PRIORITY=0: one stream per slot, as until round 5).  The two streams hand over with events.
int run_align_bins(rpa_engine* e, const WorkDev& w, const uint32_t* counts_in) {
  if (!e->owner || !e->align_low_priority || e->profiling) return run_align_bins_on_stream(e, w, counts_in);
  if (!e->lo_stream) {
    int least = 0, greatest = 0;
    HIP_TRY(e, hipDeviceGetStreamPriorityRange(&least, &greatest));
    HIP_TRY(e, hipStreamCreateWithPriority(&e->lo_stream, hipStreamNonBlocking, least));
    HIP_TRY(e, hipEventCreateWithFlags(&e->ev_lo_begin, hipEventDisableTiming));
    HIP_TRY(e, hipEventCreateWithFlags(&e->ev_lo_end, hipEventDisableTiming));
  }
  HIP_TRY(e, hipEventRecord(e->ev_lo_begin, e->stream));
  HIP_TRY(e, hipStreamWaitEvent(e->lo_stream, e->ev_lo_begin, 0));
  hipStream_t own = e->stream;
  e->stream = e->lo_stream;  // (everything run_align_bins_on_stream queues -- launches, retries, side-stream forks -- hangs off this stream)
  const int rc = run_align_bins_on_stream(e, w, counts_in);
  e->stream = own;
  if (rc) return rc;
  HIP_TRY(e, hipEventRecord(e->ev_lo_end, e->lo_stream));
  HIP_TRY(e, hipStreamWaitEvent(e->stream, e->ev_lo_end, 0));
  return RPA_OK;
}

int run_align_bins_on_stream(rpa_engine* e, const WorkDev& w, const uint32_t* counts_in) {
  unsigned long long* acc = (e->profiling && e->prof_counters) ? e->prof_acc.as<unsigned long long>() : nullptr;
  const uint32_t* counts = counts_in;
  // Side streams for the launches that have one: only when the round has several launches to make, and never under the profile
  // (its event pairs time one launch after the other on the engine's stream).
  int n_side_candidates = 0, n_launch_groups = 0;
  for (int bin = 0; bin < kNumBins; ++bin) {
    if (!counts[bin]) continue;
    const bool side_ok = e->params.alphabet == RPA_DNA ? bin >= kShortWords : bin > kProteinPackedBin;
    n_side_candidates += side_ok ? 1 : 0;
    ++n_launch_groups;
  }
  // ... and not on a job slot: several batches in flight already fill each other's gaps, and four slots x five streams are more
  // streams than hardware queues (make_stream): tools/ragged_bench.py, mixed shapes: a synchronous step 28.0 -> 23.1 ms with the
  // side streams, four steps in flight 15.7 -> 17.3 ms per step.
  const bool fork = e->side_streams_on && !e->profiling && e->owner == nullptr && n_side_candidates >= 1 && n_launch_groups >= 2;
  bool side_used[rpa_engine::kSideStreams] = {false, false, false, false};
  int side_next = 0;
  if (fork) {
    if (!e->ev_fork) {
      HIP_TRY(e, hipEventCreateWithFlags(&e->ev_fork, hipEventDisableTiming));
      for (int k = 0; k < rpa_engine::kSideStreams; ++k) {
        HIP_TRY(e, hipStreamCreateWithFlags(&e->side_stream[k], hipStreamNonBlocking));
        HIP_TRY(e, hipEventCreateWithFlags(&e->ev_join[k], hipEventDisableTiming));
      }
    }
    HIP_TRY(e, hipEventRecord(e->ev_fork, e->stream));
  }
  auto take_side = [&](int* which) -> hipStream_t {  // the next side stream (it waits for everything queued on the engine's stream so far)
    const int k = side_next++ % rpa_engine::kSideStreams;
    if (!side_used[k]) { side_used[k] = true; (void)hipStreamWaitEvent(e->side_stream[k], e->ev_fork, 0); }
    *which = k;
    return e->side_stream[k];
  };
  auto join_sides = [&]() -> int {
    for (int k = 0; k < rpa_engine::kSideStreams; ++k)
      if (side_used[k]) {
        HIP_TRY(e, hipEventRecord(e->ev_join[k], e->side_stream[k]));
        HIP_TRY(e, hipStreamWaitEvent(e->stream, e->ev_join[k], 0));
        side_used[k] = false;
      }
    return RPA_OK;
  };
  int rider = -1;  // a narrower one-strip class waiting for the next launch
  uint64_t banded_pairs = 0;  // pairs that went through the banded strip kernel this round: the retry list's bound
  uint64_t banded_short = 0;  // ... through the banded one-strip kernel (its own retry list and counter)
  uint64_t ladder_short = 0;  // ... of those, the classes of 9..12 words (judged per class, not with the narrower ones)
  int widest_short = 0;       // widest pattern class (in words) among the banded one-strip launches
  const bool use_band_long = e->band_long && e->band_long_pause == 0, use_band_short = e->band_short && e->band_short_pause == 0;
  bool use_window = false, window_on[kLadderBins];
  for (int k = 0; k < kLadderBins; ++k) {
    window_on[k] = e->band_window && e->band_window_pause[k] == 0;
    if (k >= kShortWords - kWaveMaxWords) use_window = use_window || window_on[k];  // (the ring-window classes proper: their own list of pairs that left)
  }
  uint64_t window_pairs = 0;  // pairs that went through the ring-window kernel this round: the bound of its own list of pairs that left its band
  if (e->params.alphabet == RPA_DNA && use_window) {
    uint64_t wp = 0;
    for (int bin = kShortWords; bin < kFirstStripBin; ++bin) wp += counts[bin];
    if (wp) HIP_TRY(e, e->window_fail.ensure((size_t)wp * sizeof(uint2)));
  }
  if (e->params.alphabet == RPA_DNA && (use_band_long || use_window)) {
    uint64_t strip_pairs = 0;
    for (int bin = kShortWords; bin < kNumBins; ++bin) strip_pairs += counts[bin];
    if (strip_pairs) HIP_TRY(e, e->retry_list.ensure((size_t)strip_pairs * sizeof(uint2)));
  }
  if (e->params.alphabet == RPA_DNA && e->band_short) {  // (the 9..12-word classes run their banded kernels whatever the narrower classes' pause says)
    uint64_t short_pairs = 0;
    for (int bin = 0; bin < kShortWords; ++bin) short_pairs += counts[bin];
    if (short_pairs) HIP_TRY(e, e->retry_short.ensure((size_t)short_pairs * sizeof(uint2)));
  }
  for (int bin = 0; bin < kNumBins; ++bin) {
    const uint32_t n = counts[bin];
    if (!n) continue;
    const uint2* list = w.bin_list + (size_t)bin * w.T2;
    const uint2* list2 = nullptr;
    uint32_t n2 = 0;
    if (e->params.alphabet == RPA_DNA && bin < kShortWords) {
      bool wider = false;
      for (int k = bin + 1; k < kShortWords; ++k) wider = wider || counts[k] != 0;
      if (rider < 0 && n < kRideAlongPairs && wider) { rider = bin; continue; }  // picked up by the next one-strip launch
      if (rider >= 0) { list2 = w.bin_list + (size_t)rider * w.T2; n2 = counts[rider]; rider = -1; }
    }
    if (e->debug_bins) fprintf(stderr, "BIN\tbin=%d\tpairs=%u\trider_pairs=%u\n", bin, n, n2);
    LaunchTimer timer{e};
    int trc = timer.begin();
    if (trc) return trc;
    if (e->params.alphabet == RPA_DNA) {
      if (bin < kShortWords) {
        // Patterns of 5..8 words inside a band of two words either side of the diagonal (exact: what falls outside goes
        // onto a retry list).  A launch of the dense rounds only: the sparse rounds put every class into one list.
        const int lk = bin - kWaveMaxWords;  // >= 0: a class of 9..12 words with a ladder of its own
        if (lk >= 0 && e->band_short && e->band_window_pause[lk] == 0 && (uint64_t)n + n2 >= std::min(e->band_short_min, e->band_window_min)) {
          // the four-word band (exact up to 80 - |delta| edits), or -- once more than a quarter of a batch's pairs have left it -- the
          // ring window of six words (160 - |delta|: 384 nt at 40 % edits) in its place; both hand what they cannot hold to the
          // every-cell kernel's retry list, and a class that loses its pairs there too computes every cell for 16 batches
          uint32_t* left_total = w.counters + kCtrRetryWindowTotal + lk;
          if (e->band_window_wide[lk] == 0 || !e->band_window)  // (RPA_BAND_WINDOW=0: no ring window anywhere -- the ladder is band -> every cell)
            HIP_TRY(e, launch_align_dna_short_band(e->stream, bin + 1, list, n, list2, n2, w, e->retry_short.as<uint2>(), w.counters + kCtrRetryShort,
                                                   e->owner != nullptr, e->band4_wide > 0 || e->band4_wide == -1, acc, left_total));
          else {
            HIP_TRY(e, launch_align_dna_window(e->stream, 6, list, n, w, e->retry_short.as<uint2>(), w.counters + kCtrRetryShort, left_total, acc));
            if (n2) HIP_TRY(e, launch_align_dna_window(e->stream, 6, list2, n2, w, e->retry_short.as<uint2>(), w.counters + kCtrRetryShort, left_total, acc));
          }
          banded_short += (uint64_t)n + n2;
          ladder_short += (uint64_t)n + n2;
          widest_short = bin + 1;
          e->banded_window_batch[lk] += (uint64_t)n + n2;
        } else if (lk < 0 && use_band_short && dna_short_band_available(bin + 1) && (uint64_t)n + n2 >= e->band_short_min) {
          HIP_TRY(e, launch_align_dna_short_band(e->stream, bin + 1, list, n, list2, n2, w, e->retry_short.as<uint2>(),
                                                 w.counters + kCtrRetryShort, e->owner != nullptr /* a job slot: other batches are in flight */,
                                                 e->band4_wide > 0 || e->band4_wide == -1, acc));
          banded_short += (uint64_t)n + n2;
          widest_short = bin + 1;
        } else if (e->short_wave_max && (uint64_t)n + n2 <= e->short_wave_max && bin < kWaveMaxWords)  // few pairs: a pattern word per lane (K1w), any class of up to eight words
          HIP_TRY(e, launch_align_dna_short_wave(e->stream, list, n, nullptr, 0, list2, n2, w, acc));
        else HIP_TRY(e, launch_align_dna_short(e->stream, bin + 1, list, n, nullptr, 0, list2, n2, w, acc));
      } else {
        // Patterns of more than 256 nt.  Many pairs: one per lane, register strips inside Ukkonen's band (K1s).  Few pairs
        // (not enough to give every SIMD a few waves of lanes): one per wavefront (K2, every cell), as long as the pattern
        // fits its 64 x 8 words.
        const uint32_t max_len = counts[kCtrMaxLen];  // longest sequence of any emitted pair: bounds text and pattern
        // Measured on MI355X (config 5 rounds, 10 kbp pairs, both kernels on the same lists, every cell computed): K1s runs
        // a lane at ~1.0 G cell updates/s while a SIMD holds one or two of its waves (0.9 k pairs: 99 ms, 115 k pairs:
        // 134 ms = 86 T/s) and levels off near 90 T/s as the chip fills (217 k pairs: 240 ms; 115 T/s over several rounds
        // of lanes); K2 reaches 17 T/s at 0.9 k pairs, 29 T/s at 2.2 k and ~50 T/s x (pattern words / (64 x words per
        // lane)) from ~4 k waves on (115 k pairs: 230 ms).  The band leaves K1s about a third of the cells on pairs
        // that differ by 20-30 % (less on closer ones): its rates are scaled by e->band_gain.
        const double words = (max_len + 31u) / 32u, wpl = (uint32_t)((words + 63) / 64);
        const double gain = use_band_long ? e->band_gain : 1.0;
        const double tp_lane = std::min(88e12, n * 1.0e9) * gain;
        const double tp_wave = std::min(1.0, n / 4096.0) * 50e12 * words / (64.0 * wpl);
        bool wave = max_len <= kWaveMaxPattern && tp_wave > tp_lane;
        if (e->long_kernel == 1 && max_len <= kWaveMaxPattern) wave = true;
        if (e->long_kernel == 2) wave = false;
        int sk = -1;
        hipStream_t st = fork ? take_side(&sk) : e->stream;
        // Patterns of up to 64 words (bins kShortWords ... kFirstStripBin - 1), launches of the dense rounds: the ring-window kernel first
        // (K1r: one lane per pair, a window of 6 / 10 words around the diagonal, exact up to 160 / 288 edits minus the length
        // difference); what lies outside goes onto the strip kernels' retry list.
        const int wwords = bin < kFirstStripBin ? window_words_of_bin(bin, e->band_window_wide[bin - kWaveMaxWords] != 0) : 0;
        if (bin < kFirstStripBin && window_on[bin - kWaveMaxWords] && dna_window_available(wwords) && n >= std::min(e->band_short_min, e->band_window_min) && e->long_kernel == 0) {
          HIP_TRY(e, launch_align_dna_window(st, wwords, list, n, w, e->window_fail.as<uint2>(), w.counters + kCtrWindowFail,
                                             w.counters + kCtrRetryWindowTotal + (bin - kWaveMaxWords), acc));
          window_pairs += n;
          e->banded_window_batch[bin - kWaveMaxWords] += n;
        } else if (wave) HIP_TRY(e, launch_align_dna_long(st, (int)wpl, list, n, nullptr, 0, w, acc));
        else {
          DevBuf& ws = sk >= 0 ? e->side_ws[sk] : e->protein_ws;  // (a scratch per stream: the launches of a round run side by side)
          if (dna_strip_scratch_words(n, max_len) * 4 > ws.cap) {
            if (sk >= 0) HIP_TRY(e, hipStreamSynchronize(st));  // (growing frees the old buffer: nothing of an earlier round may still use it -- rounds are joined, this is for safety)
            HIP_TRY(e, ws.ensure(dna_strip_scratch_words(n, max_len) * 4));
          }
          HIP_TRY(e, launch_align_dna_strips(st, strip_words_of_bin(bin), list, n, nullptr, 0, w, ws.as<uint32_t>(),
                                             ws.cap / 4, max_len, !use_band_long, e->band_margin, e->retry_list.as<uint2>(),
                                             w.counters + kCtrRetry, acc));
          if (use_band_long) banded_pairs += n;
        }
      }
    } else {
      const uint32_t max_len = counts[kCtrMaxLen];
      if (bin > kProteinPackedBin) {  // K3l: the strips of a pair in adjacent lanes, nothing but the work counter in memory
        uint32_t* ctr = nullptr;
        trc = next_work_counter(e, &ctr);
        if (trc) return trc;
        int sk = -1;
        hipStream_t st = fork ? take_side(&sk) : e->stream;
        if (bin >= kProteinWideLaneBin) HIP_TRY(e, launch_align_protein_lanes(st, protein_wide_class_lanes(bin - kProteinWideLaneBin), list, n, w, ctr, acc, true));
        else HIP_TRY(e, launch_align_protein_lanes(st, protein_class_lanes(bin - kProteinPackedBin - 1), list, n, w, ctr, acc));
        trc = timer.end();
        if (trc) return trc;
        continue;
      }
      HIP_TRY(e, e->protein_ws.ensure(protein_scratch_words(n, max_len) * 4));
      uint32_t* ctr = nullptr;
      if (bin == kProteinPackedBin || (bin == kProteinWideBin && e->params.protein_wide && e->params.force_bin < 0)) {
        trc = next_work_counter(e, &ctr);
        if (trc) return trc;
      }
      if (bin == kProteinPackedBin)  // every pair of this list passed protein_fits_packed(): no sequence above kPackedMaxLen
        HIP_TRY(e, launch_align_protein_packed(e->stream, list, n, w, e->protein_ws.as<uint32_t>(), e->protein_ws.cap / 4,
                                               std::min(max_len, kPackedMaxLen), ctr, acc));
      else if (bin == kProteinWideBin && e->params.protein_wide && e->params.force_bin < 0)  // ... protein_fits_wide(): the score-and-length cell
        HIP_TRY(e, launch_align_protein_packed(e->stream, list, n, w, e->protein_ws.as<uint32_t>(), e->protein_ws.cap / 4,
                                               std::min(max_len, kWideMaxLen), ctr, acc, true));
      else
        HIP_TRY(e, launch_align_protein(e->stream, 8 * (bin + 1), list, n, w, e->protein_ws.as<uint32_t>(), e->protein_ws.cap / 4, max_len, acc));
    }
    trc = timer.end();
    if (trc) return trc;
  }
  if (banded_short) {
    // One-strip pairs outside their band: every cell, by the kernel of the widest class that was launched (exact for
    // the narrower ones).  The count lives on the device; 2048 workgroups (two rounds of the chip) stride over the list.
    const uint32_t bound = banded_short > 0xFFFFFF00ull ? 0xFFFFFF00u : (uint32_t)banded_short;
    LaunchTimer timer{e};
    int trc = timer.begin();
    if (trc) return trc;
    // (how many pairs the list holds only the device knows; when even the banded launches were few pairs, K1w takes it.  Both
    // kernels on every list, each working on its side of a device-side threshold, was measured: the retry lists of config 2's
    // dense rounds hold 13 k - 100 k pairs, K1w's 41 us + an empty launch against 49 us -- not worth a second launch)
    if (e->short_wave_max && bound <= e->short_wave_max && widest_short <= kWaveMaxWords)
      HIP_TRY(e, launch_align_dna_short_wave(e->stream, e->retry_short.as<uint2>(), bound, w.counters + kCtrRetryShort, bound,
                                             nullptr, 0, w, nullptr));
    else
      HIP_TRY(e, launch_align_dna_short(e->stream, widest_short, e->retry_short.as<uint2>(), bound, w.counters + kCtrRetryShort, 2048u,
                                        nullptr, 0, w, nullptr));
    trc = timer.end();
    if (trc) return trc;
  }
  { const int jrc = join_sides(); if (jrc) return jrc; }  // (the one-strip launches and their retries above ran beside the side streams' launches)
  if (window_pairs) {
    // What left the ring window (a device-side count: normally a few pairs, all of them when the batch is more divergent than the
    // window is wide -- the engine then pauses the window for 16 batches): through the strip kernel with its adaptive band, as
    // if the pairs had gone there in the first place; what leaves that band too joins the retry list below.
    const uint32_t bound = window_pairs > 0xFFFFFF00ull ? 0xFFFFFF00u : (uint32_t)window_pairs;
    const uint32_t max_len = counts[kCtrMaxLen];
    const uint32_t blocks = 1024u;
    LaunchTimer timer{e};
    int trc = timer.begin();
    if (trc) return trc;
    // (a few thousand pairs: one per wavefront, every cell -- a strip launch takes ~0.5 ms however few pairs it carries, as every
    // lane-per-pair kernel takes what one lane takes.  Both are queued; the list's length decides on the device which of them works.)
    const uint32_t few = max_len <= kWaveMaxPattern ? e->window_fail_wave_max : 0u;
    WorkDev wk = w;
    if (few) {
      const uint32_t wpl = ((max_len + 31u) / 32u + 63u) / 64u;
      wk.dev_count_lo = 0; wk.dev_count_hi = few;
      HIP_TRY(e, launch_align_dna_long(e->stream, (int)(wpl ? wpl : 1u), e->window_fail.as<uint2>(), std::min(bound, few), w.counters + kCtrWindowFail,
                                       16384u, wk, nullptr));
    }
    wk.dev_count_lo = few; wk.dev_count_hi = 0;
    HIP_TRY(e, e->protein_ws.ensure(dna_strip_scratch_words(std::min(bound, blocks * (uint32_t)kBlock), max_len) * 4));
    HIP_TRY(e, launch_align_dna_strips(e->stream, kStripMaxWords, e->window_fail.as<uint2>(), bound, w.counters + kCtrWindowFail, blocks, wk,
                                       e->protein_ws.as<uint32_t>(), e->protein_ws.cap / 4, max_len, !use_band_long, e->band_margin,
                                       e->retry_list.as<uint2>(), w.counters + kCtrRetry, nullptr));
    if (use_band_long) banded_pairs += window_pairs;
    trc = timer.end();
    if (trc) return trc;
  }
  if (banded_pairs) {
    // The pairs whose distance lies outside their band, with the band wide open.  How many there are only the device
    // knows (counters[kCtrRetry]): the launch is sized for "a few" and strides over the list.  One pair per wavefront
    // when the pattern fits (a lone lane needs ~0.1 s for 10 kbp x 10 kbp), else the strip kernel again.  Not counted
    // into the profile's pairs / bytes a second time (its time is).
    const uint32_t bound = banded_pairs > 0xFFFFFF00ull ? 0xFFFFFF00u : (uint32_t)banded_pairs;
    const uint32_t max_len = counts[kCtrMaxLen];
    LaunchTimer timer{e};
    int trc = timer.begin();
    if (trc) return trc;
    if (max_len <= kWaveMaxPattern) {
      const uint32_t wpl = ((max_len + 31u) / 32u + 63u) / 64u;
      HIP_TRY(e, launch_align_dna_long(e->stream, (int)(wpl ? wpl : 1u), e->retry_list.as<uint2>(), bound, w.counters + kCtrRetry, 16384u, w, nullptr));
    } else {
      const uint32_t blocks = 1024u;  // 4 workgroups per CU
      HIP_TRY(e, e->protein_ws.ensure(dna_strip_scratch_words(std::min(bound, blocks * (uint32_t)kBlock), max_len) * 4));
      HIP_TRY(e, launch_align_dna_strips(e->stream, kStripMaxWords, e->retry_list.as<uint2>(), bound, w.counters + kCtrRetry, blocks, w,
                                         e->protein_ws.as<uint32_t>(), e->protein_ws.cap / 4, max_len, true, 0u, nullptr, nullptr, nullptr));
    }
    trc = timer.end();
    if (trc) return trc;
  }
  e->banded_short_batch += banded_short - ladder_short;
  e->banded_long_batch += banded_pairs - (use_band_long ? window_pairs : 0);
  return RPA_OK;
}

int profile_begin(rpa_engine* e) {
  if (!e->profiling) return RPA_OK;
  e->ev_used = 0;
  e->ev_align.clear();
  if (!e->ev_total0) { HIP_TRY(e, hipEventCreate(&e->ev_total0)); HIP_TRY(e, hipEventCreate(&e->ev_total1)); }
  HIP_TRY(e, hipEventRecord(e->ev_total0, e->stream));
  return RPA_OK;
}

int profile_end(rpa_engine* e) {
  if (!e->profiling) return RPA_OK;
  HIP_TRY(e, hipEventRecord(e->ev_total1, e->stream));
  RPA_WHERE("profile_end: event sync");
  HIP_TRY(e, hipEventSynchronize(e->ev_total1));
  float ms = 0.f;
  HIP_TRY(e, hipEventElapsedTime(&ms, e->ev_total0, e->ev_total1));
  if (!e->part_mode) e->prof.total_ms += ms;
  for (auto& pr : e->ev_align) {
    HIP_TRY(e, hipEventElapsedTime(&ms, e->ev_pool[pr.first], e->ev_pool[pr.second]));
    e->prof.align_ms += ms;
  }
  if (!e->prof_counters) return RPA_OK;  // (timing only: pairs / bytes / cells / demanded stay what the last counting pass left)
  unsigned long long parts[kProfParts * kProfSlots], acc[kProfSlots] = {};
  HIP_TRY(e, hipMemcpy(parts, e->prof_acc.p, sizeof(parts), hipMemcpyDeviceToHost));
  for (int k = 0; k < kProfParts; ++k)
    for (int j = 0; j < kProfSlots; ++j) acc[j] += parts[k * kProfSlots + j];
  e->prof.pairs = acc[0]; e->prof.residue_bytes = acc[1]; e->prof.cells = acc[2]; e->prof.demanded = acc[3]; e->prof.cells_updated = acc[4];
  return RPA_OK;
}

// Working-set allocation shared by predict and align_pairs.  T = segments, T2 = pair slots.
// P = row slots (tiled layout: padded; linear: H); tiled selects the layout of the rows / results (predict only).
int alloc_work(rpa_engine* e, uint32_t S, uint32_t H, uint32_t T, uint32_t T2, bool want_ext, WorkDev* w, uint32_t P, bool tiled,
               int row_quads = 2, uint32_t lb_shift = 3) {
  HIP_TRY(e, e->state.ensure((size_t)(S ? S : 1) * sizeof(SetStateMem)));
  HIP_TRY(e, e->loop_save.ensure((size_t)(S ? S : 1) * sizeof(LoopSave)));
  HIP_TRY(e, e->hits.ensure((size_t)(P ? P : 1) * (size_t)row_quads * 16));
  if (!want_ext && tiled) HIP_TRY(e, e->res_slot.ensure((size_t)(H ? H : 1) * 4));
  HIP_TRY(e, e->set_info.ensure((size_t)(S ? S : 1) * sizeof(uint4)));
  HIP_TRY(e, e->emit.ensure((size_t)(S ? S : 1) * sizeof(EmitInfo)));
  HIP_TRY(e, e->emit_lite.ensure((size_t)(S ? S : 1) * sizeof(uint4)));
  HIP_TRY(e, e->emit_ext.ensure((size_t)(S ? S : 1) * sizeof(uint4)));
  HIP_TRY(e, e->hit_set.ensure((size_t)(H ? H : 1) * 4));
  HIP_TRY(e, e->block_first.ensure(((size_t)H / kBlock + 16) * 4));  // one word per k_rank_records workgroup (grid rounded up to 8)
  HIP_TRY(e, e->flags.ensure((size_t)(H ? H : 1) * 4));
  HIP_TRY(e, e->lite.ensure((size_t)(H ? H : 1) * 4));
  HIP_TRY(e, e->seg.ensure((size_t)(T ? T : 1) * sizeof(SegDesc)));
  if (want_ext) {  // explicit pair arrays exist only for rpa_align_pairs
    HIP_TRY(e, e->pair_a.ensure((size_t)(T2 ? T2 : 1) * 4));
    HIP_TRY(e, e->pair_b.ensure((size_t)(T2 ? T2 : 1) * 4));
  }
  const size_t n_res = (size_t)(want_ext ? (T > T2 ? T : T2) : (size_t)P + S) + 1;  // predict: one per row slot + one per set; align_pairs: one per pair
  HIP_TRY(e, e->pair_res.ensure(n_res * 8));
  if (want_ext) HIP_TRY(e, e->pair_ext.ensure(n_res * 16));
  HIP_TRY(e, e->bin_list.ensure((size_t)(T2 ? T2 : 1) * sizeof(uint2) * (e->params.alphabet == RPA_DNA ? kDnaBins : kProteinBins)));
  HIP_TRY(e, e->counters.ensure(sizeof(uint32_t) * (kNumCounters + 1)));  // +1: scan total
  HIP_TRY(e, e->live_a.ensure((size_t)(S ? S : 1) * 4));
  HIP_TRY(e, e->live_b.ensure((size_t)(S ? S : 1) * 4));
  HIP_TRY(e, e->band_save.ensure((size_t)(S ? S : 1) * 2 * (e->tax.max_depth + 1) * 4));
  HIP_TRY(e, e->og_idx.ensure((size_t)(T2 ? T2 : 1) * 4));
  HIP_TRY(e, e->og_dist.ensure((size_t)(T2 ? T2 : 1) * 4));
  if (e->poison) {
    DevBuf* bufs[] = {&e->state, &e->loop_save, &e->hits, &e->res_slot, &e->set_info, &e->emit, &e->emit_lite, &e->emit_ext, &e->hit_set, &e->block_first,
                      &e->flags, &e->lite, &e->seg, &e->pair_a, &e->pair_b, &e->pair_res, &e->pair_ext, &e->bin_list, &e->live_a, &e->live_b,
                      &e->band_save, &e->og_idx, &e->og_dist, &e->retry_list, &e->retry_short, &e->window_fail, &e->protein_ws, &e->out_rows, &e->out_stats};
    for (DevBuf* b : bufs)
      if (b->p) HIP_TRY(e, hipMemsetAsync(b->p, 0xA5, b->cap, e->stream));
  }
  w->state = e->state.as<SetStateMem>(); w->loop_save = e->loop_save.as<LoopSave>(); w->hits = e->hits.as<uint32_t>(); w->set_info = e->set_info.as<uint4>();
  w->emit = e->emit.as<EmitInfo>(); w->emit_lite = e->emit_lite.as<uint4>(); w->emit_ext = e->emit_ext.as<uint4>(); w->hit_set = e->hit_set.as<uint32_t>(); w->block_first = e->block_first.as<uint32_t>();
  w->rec_pre = e->rec_pre.as<const uint32_t>();
  w->flags = e->flags.as<uint32_t>();
  w->lite = e->lite.as<uint32_t>();
  w->seg = e->seg.as<SegDesc>(); w->store_q = nullptr; w->store_db = nullptr;
  w->pair_a = want_ext ? e->pair_a.as<uint32_t>() : nullptr; w->pair_b = want_ext ? e->pair_b.as<uint32_t>() : nullptr;
  w->set_offsets = nullptr; w->H = H; w->row_quads = row_quads; w->lb_shift = lb_shift;
  w->trace = want_ext ? nullptr : e->trace_view; w->trace_n = e->trace_view_n; w->trace_cap = e->trace_view_cap; w->trace_set_base = e->trace_set_base;
  w->P = P; w->tile_base = (!want_ext && tiled) ? e->tile_base.as<const uint32_t>() : nullptr;
  w->res_slot = (want_ext || !tiled) ? nullptr : e->res_slot.as<uint32_t>();  // linear layout: result slot = row position (P = H)
  w->pair_res = e->pair_res.as<float2>(); w->pair_ext = want_ext ? e->pair_ext.as<uint4>() : nullptr;
  w->bin_list = e->bin_list.as<uint2>(); w->counters = e->counters.as<uint32_t>();
  w->ctr_live_in = kCtrActive; w->ctr_live_out = kCtrActive;  // (k_plan_sets counts the first live list here; predict_part alternates from then on)
  w->og_idx = e->og_idx.as<uint32_t>(); w->band_save = e->band_save.as<float>();
  w->live_in = nullptr; w->n_live_in = 0; w->live_out = e->live_a.as<uint32_t>();
  w->band_ranks = e->tax.max_depth + 1;
  w->og_dist = e->og_dist.as<int32_t>();
  w->T = T; w->T2 = T2;
  return RPA_OK;
}

int check_device_error(rpa_engine* e, const uint32_t* c) {
  if (c[kCtrErr] == 0) return RPA_OK;
  const uint32_t code = c[kCtrErr];
  int rc = code == kErrRange ? RPA_ERR_RANGE : code == kErrSeqId ? RPA_ERR_SEQ_ID : code == kErrAssert ? RPA_ERR_ASSERT
           : code == kErrUnsupported ? RPA_ERR_UNSUPPORTED : RPA_ERR_INVALID;  // (kErrNotResident: the caller's omission)
  return fail(e, rc, "%s (record set / pair %u)", kernel_error_text(code), c[kCtrErrSet]);
}

// One part of a batch on engine `e` (the owner for the first part, a worker for the others): everything
// rpa_predict_batch does on the device for the record sets of `b` (device pointers, set offsets starting at 0).
struct PartGate {
  std::mutex m; std::condition_variable cv; bool is_open = false;
  void open() { { std::lock_guard<std::mutex> lk(m); is_open = true; } cv.notify_all(); }
  void wait() { std::unique_lock<std::mutex> lk(m); cv.wait(lk, [&] { return is_open; }); }
};
struct GateOpener {  // opens the gate on every way out of predict_part
  PartGate& g;
  ~GateOpener() { g.open(); }
};

// The feedback of a ladder class, from the counters every round publishes (`left_total` = pairs of the class that left its banded
// kernel in this batch so far): too many of the pairs the CURRENT instance saw left it -> the first instance hands over to the
// second (for this batch's remaining rounds and 16 batches), the second to the fallback behind the ladder (a pause of 16 batches).
void window_feedback(rpa_engine* e, int k, uint64_t left_total) {
  const uint64_t pairs = e->banded_window_batch[k] - e->window_seen_pairs[k], left = left_total - e->window_seen_left[k];
  if (e->band_window_pause[k] != 0 || pairs < 1024) return;
  // A pair that leaves the first instance costs that instance and the fallback behind the ladder (the strips: 4 x a narrow window's
  // price; every cell: 2 x the band's), the second instance 1.5 x: one pair in sixteen leaving is where the second instance is cheaper
  // (the one-strip band's own rule).  The second instance still pays while half of its pairs stay.
  const bool second_exists = e->band_window || k >= kShortWords - kWaveMaxWords;  // (the 9..12-word classes' second instance is the ring window)
  if (e->band_window_wide[k] == 0 && second_exists) { if (left * 16 <= pairs) return; e->band_window_wide[k] = 17; }   // (17: the end of this batch takes one off)
  else { if (left * 2 <= pairs) return; e->band_window_pause[k] = 17; }
  e->window_seen_pairs[k] = e->banded_window_batch[k];
  e->window_seen_left[k] = left_total;
}

int predict_part(rpa_engine* e, const BatchDev& b, const StoreDev& q, const StoreDev& db, void* rows_dev, void* stats_dev, PartGate& gate) {
  GateOpener opener{gate};
  const uint32_t S = b.n_sets, H = b.n_records, T = H + S;
  // Row layout.  Tiled (64 record sets side by side, see SetRows in place_kernels.hip) makes the placement kernels'
  // row accesses coalesced; it pads every tile to its largest set, so batches whose padding would exceed ~50 % of the
  // rows (wildly different set sizes) keep the linear layout.  The slot count comes back from the device.
  const uint32_t n_tiles = (S + 63u) / 64u;
  HIP_TRY(e, e->tile_base.ensure((size_t)(n_tiles + 6) * 4));  // + total slots, + the two words of k_validate_batch, + largest set, + number of large sets
  uint32_t P = H;
  bool tiled = false;
  uint32_t lb_shift = 3;
  int row_quads = 2;
  uint32_t n_big_sets = 0, largest_set = 0;
  {
    uint32_t* tail = e->tile_base.as<uint32_t>() + n_tiles;
    HIP_TRY(e, hipMemsetAsync(tail, 0, 20, e->stream));
    HIP_TRY(e, e->rec_pre.ensure((size_t)(H ? H : 1) * 4));
    HIP_TRY(e, launch_validate_batch(e->stream, b, e->tax.n_nodes, tail + 1, e->tax.pre, e->rec_pre.as<uint32_t>()));
    HIP_TRY(e, launch_tile_plan(e->stream, b.set_offsets, S, e->tile_base.as<uint32_t>(), tail + 3, !e->linear_rows));
    if (e->big_set_min && H >= e->big_set_min) {  // the record sets a workgroup of its own will rank (k_rank_big_sets): listed now, counted in tail[4]
      HIP_TRY(e, e->big_sets.ensure((size_t)(H / e->big_set_min + 1) * 4));
      HIP_TRY(e, launch_list_big_sets(e->stream, b.set_offsets, S, e->big_set_min, e->big_sets.as<uint32_t>(), tail + 4));
    }
    HIP_TRY(e, launch_publish_words(e->stream, tail, e->h_counters, 5));
    RPA_WHERE("predict_part: sync after validate/tile plan");
    HIP_TRY(e, hipStreamSynchronize(e->stream));
    if (e->h_counters[1]) return fail(e, RPA_ERR_INVALID, "batch rejected: set_offsets must run from 0 to n_records without decreasing and every node index must lie inside the taxonomy (set / record %u)", e->h_counters[2]);
    const uint64_t slots = e->h_counters[0];
    if (!e->linear_rows && slots != 0xFFFFFFFFull && slots <= (uint64_t)H + H / 2 + 65536u && slots + S < 0xFFFFFFF0ull) { P = (uint32_t)slots; tiled = true; }
    // Row shape of this batch: a taxon path label is a record's index inside its set, so its width follows the largest set
    // (8 bits up to 256 records), and a row must hold one label per taxonomy level.
    const uint32_t max_set = e->h_counters[3];
    n_big_sets = e->h_counters[4]; largest_set = max_set;
    lb_shift = e->force_label_shift ? e->force_label_shift : label_shift_for(max_set);
    if (lb_shift < label_shift_for(max_set)) lb_shift = label_shift_for(max_set);
    row_quads = row_quads_for(e->tax.max_depth, lb_shift);
    if (!row_quads)
      return fail(e, RPA_ERR_UNSUPPORTED, "a record set of %u records on a taxonomy of %u levels: %u levels x %u label bits exceed the widest hit row of this build (832 bits: 63 levels for record sets of up to 256 records, 52 up to 65536, 26 beyond) -- prune the taxonomy (taxator -r / -d 1) or split the record set",
                  max_set, e->tax.max_depth, e->tax.max_depth, 1u << lb_shift);
  }
  WorkDev w{};
  int rc = alloc_work(e, S, H, T, T, false, &w, P, tiled, row_quads, lb_shift);
  if (rc) return rc;

  rc = profile_begin(e);
  if (rc) return rc;
  HIP_TRY(e, hipMemsetAsync(w.counters, 0, sizeof(uint32_t) * (kNumCounters + 1), e->stream));
  if (e->params.alphabet != RPA_DNA) { const int crc = counter_pool_reset(e); if (crc) return crc; }
  // (w.seg needs no clearing: k_rank_records writes the descriptor of every row position and query range a list entry can name)
  w.set_offsets = b.set_offsets;
  w.store_q = q.words; w.store_db = db.words;
  w.big_min = n_big_sets ? e->big_set_min : 0u;
  HIP_TRY(e, launch_rank_records(e->stream, b, w, e->tax, q, db, e->params));
  HIP_TRY(e, launch_rank_big_sets(e->stream, b, w, e->tax, q, db, e->params, e->big_sets.as<uint32_t>(), n_big_sets, largest_set));
  if (e->rank_only) {  // RPA_DEBUG_RANK_ONLY=1 (tools/rank_parts.sh: timing builds of k_rank_records whose rows are not usable): stop here
    HIP_TRY(e, hipStreamSynchronize(e->stream));
    return fail(e, RPA_ERR_UNSUPPORTED, "RPA_DEBUG_RANK_ONLY: stopped behind k_rank_records");
  }
  HIP_TRY(e, launch_plan_sets(e->stream, b, w, e->tax, e->params, rows_dev, stats_dev));
  uint64_t rounds = 0;
  // Two launches per pass-2 round (kDeferredMark) cost a replay kernel, a host round trip and a second set of launches per round:
  // ~0.3 ms on a million record sets.  What they save is alignments -- 8 % of them on the BASELINE shapes -- which pays when an
  // alignment is expensive: decided per batch from round 0's own work lists, "most pairs have patterns of more than 256 nt" (the
  // strip kernels' bins; config 5: -4 % per step, configs 2 and 4: +0 ... +4 %, mixed shapes with a launch per bin: +30 %).
  bool defer_pays = false;
  uint32_t prev_live = S;     // upper bound of the current live list's length (its exact length is counters[kCtrActive])
  bool list_in_a = true;      // plan wrote the first live list into live_a
  for (;;) {
    // plan/decide left, per set, what it wants aligned next (EmitInfo) and appended the set to the live list.
    // Dense round: scan every row / every set.  Sparse round (few sets left): walk the live list only.
    uint32_t* cur_list = list_in_a ? e->live_a.as<uint32_t>() : e->live_b.as<uint32_t>();
    uint32_t* next_list = list_in_a ? e->live_b.as<uint32_t>() : e->live_a.as<uint32_t>();
    const bool sparse_emit = rounds > 0 && (uint64_t)prev_live * e->sparse_div_emit <= S;
    w.live_in = sparse_emit ? cur_list : nullptr;
    w.n_live_in = prev_live;  // grid bound only: the kernel reads the exact length from counters[ctr_live_in]
    w.ctr_live_in = (rounds & 1) ? kCtrActiveB : kCtrActive;   // (plan counted the first list in kCtrActive)
    w.ctr_live_out = (rounds & 1) ? kCtrActive : kCtrActiveB;
    Params pe = e->params;
    pe.defer2 = (e->defer_mode == 1 || (e->defer_mode == 2 && defer_pays)) && !sparse_emit ? 1 : 0;
    HIP_TRY(e, launch_emit(e->stream, b, w, e->tax, pe));
    HIP_TRY(e, launch_publish_words(e->stream, w.counters, e->h_counters, kNumCounters));  // (a kernel, not a copy: see k_publish_words)
    RPA_WHERE("predict_part: sync after emit (round)");
    HIP_TRY(e, hipStreamSynchronize(e->stream));
    rc = check_device_error(e, e->h_counters);
    if (rc) return rc;
    const uint32_t live_now = e->h_counters[w.ctr_live_in];
    if (live_now == 0) {  // the batch is done: what did the bands cost? (totals added up by the retry launches)
      // (the classes of 9..12 words are judged per class below: their pairs are not in banded_short_batch, their share of the retried ones comes off here)
      uint64_t short_left = e->h_counters[kCtrRetryShortTotal];
      for (int k = 0; k < kShortWords - kWaveMaxWords; ++k) short_left -= std::min<uint64_t>(short_left, e->h_counters[kCtrRetryWindowTotal + k]);
      if (e->banded_short_batch >= 65536 && short_left * 4 > e->banded_short_batch) e->band_short_pause = 16;
      else if (e->band_short_pause) --e->band_short_pause;
      // the narrow band: a pair that leaves it costs the band and every cell (2.2 x the wide band's price), one that stays saves an
      // eighth -- beyond one pair in sixteen the wide band is cheaper
      if (e->trace_rounds && e->banded_short_batch)
        fprintf(stderr, "[rpa] batch done: %u of %llu pairs left the one-strip band (%s)\n", (uint32_t)short_left,
                (unsigned long long)e->banded_short_batch, (e->band4_wide > 0 || e->band4_wide == -1) ? "four words: 96 - |delta|" : "three-word middle: 80 - |delta|");
      if (e->band4_wide >= 0) {
        if (e->band4_wide == 0 && e->banded_short_batch >= 65536 && short_left * 16 > e->banded_short_batch) e->band4_wide = 16;
        else if (e->band4_wide > 0) --e->band4_wide;
      }
      // (the retry launches count the whole retry list into kCtrRetryTotal; the ring-window kernel counts its own share)
      uint64_t from_window = 0;
      for (int k = 0; k < kLadderBins; ++k) {
        const uint64_t left = e->h_counters[kCtrRetryWindowTotal + k];
        if (k >= kShortWords - kWaveMaxWords) from_window += left;  // (the ring-window classes proper share the strips' retry counters)
        window_feedback(e, k, left);
        if (e->trace_rounds && e->banded_window_batch[k])
          fprintf(stderr, "[rpa] batch done: %llu of %llu pairs left the ring window (class %d; next batch: %s)\n", (unsigned long long)left, (unsigned long long)e->banded_window_batch[k],
                  k - (kShortWords - kWaveMaxWords),  // (>= 0: the ring-window classes; -4 .. -1: the one-strip classes of 9 .. 12 words)
                  k < kShortWords - kWaveMaxWords ? (e->band_window_pause[k] > 1 ? "every cell" : e->band_window_wide[k] > 1 || e->band_window_wide[k] < 0 ? "ring window" : "four-word band")
                                                  : (e->band_window_pause[k] > 1 ? "strips" : e->band_window_wide[k] > 1 || e->band_window_wide[k] < 0 ? "wide window" : "narrow window"));
        if (e->band_window_pause[k]) { if (--e->band_window_pause[k] == 0) e->band_window_wide[k] = e->band_window_wide[k] < 0 ? -1 : 0; }  // (after a pause: the narrow one again)
        else if (e->band_window_wide[k] > 0) --e->band_window_wide[k];
        e->banded_window_batch[k] = 0; e->window_seen_pairs[k] = 0; e->window_seen_left[k] = 0;
      }
      const uint64_t from_strips = e->h_counters[kCtrRetryTotal] > from_window ? e->h_counters[kCtrRetryTotal] - from_window : 0;
      if (e->banded_long_batch >= 1024 && from_strips * 4 > e->banded_long_batch) e->band_long_pause = 16;
      else if (e->band_long_pause) --e->band_long_pause;
      e->banded_short_batch = 0; e->banded_long_batch = 0;
      break;
    }
    uint32_t counts[kNumRoundCounters];
    memcpy(counts, e->h_counters, sizeof(counts));
    // (a length class whose pairs are leaving the ring window in this very batch -- the totals of the rounds so far came with the
    // counters -- does not try it again in the rounds that follow; the pause then lasts its 16 batches from the end of this one)
    for (int k = 0; k < kLadderBins; ++k) window_feedback(e, k, e->h_counters[kCtrRetryWindowTotal + k]);
    if (rounds == 0 && e->params.alphabet == RPA_DNA) {
      uint64_t all = 0, strips = 0;
      for (int k = 0; k < kNumBins; ++k) { all += counts[k]; if (k >= kShortWords) strips += counts[k]; }
      defer_pays = all > 0 && strips * 2 >= all;
    }
    if (e->trace_rounds) {
      uint64_t pairs = 0;
      for (int k = 0; k < kNumBins; ++k) pairs += counts[k];
      fprintf(stderr, "[rpa] round %llu (%s): live sets %u, alignments %llu, per bin:", (unsigned long long)rounds, sparse_emit ? "sparse" : "dense",
              live_now, (unsigned long long)pairs);
      for (int k = 0; k < kNumBins; ++k) if (counts[k]) fprintf(stderr, " %d:%u", k, counts[k]);
      fprintf(stderr, "\n");
    }
    rc = run_align_bins(e, w, counts);
    if (rc) return rc;
    if (e->h_counters[kCtrDeferred]) {
      // The round's second launch (kDeferredMark, rpa_device.h): pass-2 hits that an earlier hit of their chunk may have taken out
      // of the loop.  The filter replays the chunks with the first launch's results; what is still reachable comes back as bin
      // lists with their lengths in counters[kCtrSurvivor ..].
      const uint32_t held_back = e->h_counters[kCtrDeferred];
      HIP_TRY(e, launch_defer_filter(e->stream, b, w, e->params));
      HIP_TRY(e, launch_publish_words(e->stream, w.counters, e->h_counters, kNumCounters));
      RPA_WHERE("predict_part: sync after the deferred-hit filter (round)");
      HIP_TRY(e, hipStreamSynchronize(e->stream));
      rc = check_device_error(e, e->h_counters);
      if (rc) return rc;
      uint32_t counts2[kNumRoundCounters];
      memcpy(counts2, counts, sizeof(counts2));
      uint64_t again = 0;
      for (int k = 0; k < kNumBins; ++k) { counts2[k] = e->h_counters[kCtrSurvivor + k]; again += counts2[k]; }
      if (e->trace_rounds) fprintf(stderr, "[rpa]   second launch: %u pass-2 alignments were held back, %llu of them are still reachable\n", held_back, (unsigned long long)again);
      if (again) {
        rc = run_align_bins(e, w, counts2);
        if (rc) return rc;
      }
    }
    if (e->trace_rounds) {  // (a synchronisation the normal path does not have)
      uint32_t retried = 0;
      HIP_TRY(e, hipMemcpyAsync(&retried, w.counters + kCtrRetry, 4, hipMemcpyDeviceToHost, e->stream));
      HIP_TRY(e, hipStreamSynchronize(e->stream));
      uint32_t retried_short = 0;
      HIP_TRY(e, hipMemcpyAsync(&retried_short, w.counters + kCtrRetryShort, 4, hipMemcpyDeviceToHost, e->stream));
      HIP_TRY(e, hipStreamSynchronize(e->stream));
      if (retried || retried_short) fprintf(stderr, "[rpa]   banded alignments that left their band and were re-run wide open: %u long, %u short\n", retried, retried_short);
    }
    if (rounds == 1) gate.open();  // the long pass-1 alignment launch is queued: the next part may start underneath it
    // (no fill of the round's counters here: k_decide clears them, rpa_device.h)
    const bool sparse_decide = rounds > 0 && (uint64_t)live_now * e->sparse_div_decide <= S;
    w.live_in = sparse_decide ? cur_list : nullptr;
    w.n_live_in = live_now;   // exact
    w.live_out = next_list;
    // In the sparse rounds at the end of a batch a round costs ~150 us whatever it carries, so the loops that pause there
    // ask for everything they could still need at once (speculation chunk 0) instead of 10 hits per round: a few hundred
    // speculative alignments more, several rounds fewer.  (Chunk sizes never change results, only who computes what when.)
    // Likewise when the record sets still running cannot fill the chip with a default-sized chunk each (one pair per lane,
    // ~262 k lanes resident): lanes that would idle anyway take the next hits of the loop.  The target is three times the
    // resident lanes: a round of long alignments costs a whole chip-round (~240 ms for 10 kbp pairs) however few lanes
    // it fills, and a launch of 1.7 rounds of lanes runs faster than two of 0.85 (config 5: 3 rounds instead of 6).
    Params pr = e->params;
    // nucleotide loops assume their cut-off from the first hit (RPA_PASS1_FILTER 3, the default) only in batches whose scores spread
    // (k_plan_sets counted the sets where that would trim a quarter of the hits: one in sixteen or more); else as protein-only (2)
    if (pr.alphabet == RPA_DNA && pr.filter1 == 3 && (uint64_t)e->h_counters[kCtrSpreadSets] * 16 < S) pr.filter1 = 2;
    if (!e->fixed_speculation) {
      if (sparse_decide) { pr.spec1 = 0; pr.spec2 = 0; pr.spec2_first = 0; }
      else {
        const uint32_t fill = e->fill_lanes / (live_now ? live_now : 1u);
        if (pr.spec1 && fill > pr.spec1) pr.spec1 = fill;
        if (pr.spec2 && fill > pr.spec2) pr.spec2 = fill;
        if (pr.spec2_first && fill > pr.spec2_first) pr.spec2_first = fill;
      }
    }
    HIP_TRY(e, launch_decide(e->stream, b, w, e->tax, pr, rows_dev, stats_dev,
                             (e->profiling && e->prof_counters) ? e->prof_acc.as<unsigned long long>() : nullptr));
    prev_live = live_now;
    list_in_a = !list_in_a;
    ++rounds;
    if (rounds > (uint64_t)H + S + 8) return fail(e, RPA_ERR_INVALID, "placement state machine did not terminate");
  }
  RPA_WHERE("predict_part: final sync");
  HIP_TRY(e, hipStreamSynchronize(e->stream));
  e->prof.rounds += rounds;
  return profile_end(e);
}

// Streams and hardware queues.  The HIP runtime multiplexes ordinary streams onto a small pool of hardware queues
// (GPU_MAX_HW_QUEUES, 4 by default), and two streams that land on one queue run in the order their packets were written:
// a job's kernels then sit behind the barrier packet another job's 80 MB host-to-device copy waits on.  Measured with
// three jobs in flight over host buffers (config 2): 19.6 ms per step on the default pool, 14.3 ms (the copy bound) when
// every stream has a queue of its own.  Round 2 got the queues by creating the streams with an all-ones CU mask (such a
// stream is never pooled) -- and that turned out to be what makes the runtime's exit handler spin for ever in a process
// that leaves within milliseconds of creating one (15 of 850 short-lived drivers, none of 2200 with plain
// streams).  Since round 3 the streams are plain non-blocking ones and the pool is made large enough instead:
// (round 5: 16, not 8 -- the owner's four side streams for the launches of a round, created by a synchronous call, moved the job
// slots' streams onto shared queues: config 4 with three jobs in flight 4.09 against 3.99 ms per step, profiles/r05_ab_hw_queues.txt)
// rpa_create puts GPU_MAX_HW_QUEUES=8 into the environment if nobody set it and the runtime has not been initialised yet
// (taxator, any ctypes caller; a process that initialises HIP first -- PyTorch -- sets it itself, bench.py does): 14.3 ms
// again (profiles/r03_ab_hostbuf.txt); device-resident batches never cared (9.5-9.6 ms either way).
// RPA_CU_MASK_QUEUES=1 brings the CU-masked streams back (A/B, tools/hang_hunt.py's control arm).
static hipError_t make_stream(int device, hipStream_t* out) {
  const char* masked = getenv("RPA_CU_MASK_QUEUES");
  if (masked && atoi(masked) != 0) {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) {
      std::vector<uint32_t> mask((prop.multiProcessorCount + 31) / 32, 0xffffffffu);
      if (prop.multiProcessorCount % 32) mask.back() = (1u << (prop.multiProcessorCount % 32)) - 1u;
      if (hipExtStreamCreateWithCUMask(out, (uint32_t)mask.size(), mask.data()) == hipSuccess) return hipSuccess;
      (void)hipGetLastError();
    }
  }
  return hipStreamCreateWithFlags(out, hipStreamNonBlocking);
}

// Worker engine for the parts of a batch beyond the first: the owner's taxonomy and parameters by view, everything
// else (stream, working set, counters, profiling accumulators) its own.
rpa_engine* make_worker(rpa_engine* e) {
  rpa_engine* w = new (std::nothrow) rpa_engine();
  if (!w) { fail(e, RPA_ERR_DEVICE, "out of host memory"); return nullptr; }
  w->is_worker = true;
  w->device = e->device; w->params = e->params; w->tax = e->tax;
  w->profiling = e->profiling; w->prof_counters = e->prof_counters; w->long_kernel = e->long_kernel; w->band_long = e->band_long; w->band_short = e->band_short; w->band_window = e->band_window; for (int k = 0; k < kLadderBins; ++k) w->band_window_wide[k] = e->band_window_wide[k] < 0 ? -1 : 0; w->band_window_min = e->band_window_min; w->window_fail_wave_max = e->window_fail_wave_max; w->big_set_min = e->big_set_min; w->band_short_min = e->band_short_min; w->band_gain = e->band_gain; w->band_margin = e->band_margin; w->trace_rounds = e->trace_rounds; w->linear_rows = e->linear_rows; w->poison = e->poison; w->force_label_shift = e->force_label_shift; w->short_wave_max = e->short_wave_max;
  w->fixed_speculation = e->fixed_speculation;
  w->side_streams_on = e->side_streams_on;
  w->defer_mode = e->defer_mode;
  w->align_low_priority = e->align_low_priority;
  w->sparse_div_emit = e->sparse_div_emit; w->sparse_div_decide = e->sparse_div_decide; w->fill_lanes = e->fill_lanes;
  hipError_t err = make_stream(w->device, &w->own_stream);
  if (err == hipSuccess) err = hipHostMalloc((void**)&w->h_counters, sizeof(uint32_t) * (kNumCounters + 1), hipHostMallocDefault);
  if (err == hipSuccess) err = w->prof_acc.ensure(kProfParts * kProfSlots * sizeof(unsigned long long));
  if (err == hipSuccess) err = hipMemset(w->prof_acc.p, 0, kProfParts * kProfSlots * sizeof(unsigned long long));
  if (err != hipSuccess) { fail(e, RPA_ERR_DEVICE, "worker engine: %s", hipGetErrorString(err)); rpa_destroy(w); return nullptr; }
  w->stream = w->own_stream;
  return w;
}

// How many parts a batch of S record sets is cut into: ONE unless RPA_PARTS asks for more.  Measured on MI355X with
// config 2 (1 M record sets): 2 parts 15.5 ms, 3 parts 15.3 ms against 16.7 ms whole (4 and more: slower again, every
// part pays its own ramp-up and its own tail of sparse rounds) -- but the alignment kernels of different parts then run
// side by side and each launch takes longer (0.68-0.79 ms instead of 1.11 ms on average at 2-3 parts), which is what
// the per-kernel roofline figure is quoted on; routing all alignment launches through one stream instead kept them
// apart but stalled the parts behind each other's long launches (16.2 / 17.5 ms).  Left as an opt-in.
int parts_for(const rpa_engine* e, uint32_t S) {
  int k = e->parts_override > 0 ? e->parts_override : 1;
  if ((uint32_t)k > S) k = (int)S;
  return k < 1 ? 1 : k;
}

// All parts of one batch.  h_offsets: the host copy of set_offsets when the caller's batch lives on the host, else null
// (the part boundaries are then read back from the device).
int run_parts(rpa_engine* e, const BatchDev& b, const uint32_t* h_offsets, void* rows_dev, void* stats_dev) {
  const uint32_t S = b.n_sets;
  const StoreDev q = e->stores[RPA_STORE_QUERY].view(), db = e->stores[RPA_STORE_DB].view();
  const int K = parts_for(e, S);
  const bool tracing = e->trace_cap != 0;
  e->trace_view = tracing ? e->trace_buf.as<TraceEntry>() : nullptr;
  e->trace_view_n = tracing ? e->trace_count.as<unsigned long long>() : nullptr;
  e->trace_view_cap = tracing ? e->trace_cap : 0;
  e->trace_set_base = 0;
  if (tracing) HIP_TRY(e, hipMemsetAsync(e->trace_count.p, 0, 8, e->stream));
  if (K == 1) {
    PartGate g;
    return predict_part(e, b, q, db, rows_dev, stats_dev, g);
  }
  {  // the part boundaries below are read out of set_offsets: check the batch first (predict_part does it for K == 1)
    HIP_TRY(e, e->tile_base.ensure(16));
    HIP_TRY(e, hipMemsetAsync(e->tile_base.p, 0, 8, e->stream));
    HIP_TRY(e, launch_validate_batch(e->stream, b, e->tax.n_nodes, e->tile_base.as<uint32_t>()));
    HIP_TRY(e, hipMemcpyAsync(e->h_counters, e->tile_base.p, 8, hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(e, hipStreamSynchronize(e->stream));
    if (e->h_counters[0]) return fail(e, RPA_ERR_INVALID, "batch rejected: set_offsets must run from 0 to n_records without decreasing and every node index must lie inside the taxonomy (set / record %u)", e->h_counters[1]);
  }
  std::vector<uint32_t> s_lo(K + 1), r_lo(K + 1);
  for (int k = 0; k <= K; ++k) s_lo[k] = (uint32_t)((uint64_t)S * k / K);
  r_lo[0] = 0; r_lo[K] = b.n_records;
  for (int k = 1; k < K; ++k) {
    if (h_offsets) r_lo[k] = h_offsets[s_lo[k]];
    else HIP_TRY(e, hipMemcpyAsync(&r_lo[k], b.set_offsets + s_lo[k], 4, hipMemcpyDeviceToHost, e->stream));
  }
  if (!h_offsets) HIP_TRY(e, hipStreamSynchronize(e->stream));
  while ((int)e->workers.size() < K - 1) {
    rpa_engine* w = make_worker(e);
    if (!w) return RPA_ERR_DEVICE;
    e->workers.push_back(w);
  }
  if (!e->ev_inputs) HIP_TRY(e, hipEventCreateWithFlags(&e->ev_inputs, hipEventDisableTiming));
  HIP_TRY(e, hipEventRecord(e->ev_inputs, e->stream));  // staged inputs / the caller's earlier work on this stream
  std::vector<PartGate> gates(K);
  std::vector<int> rcs(K, RPA_OK);
  const auto t0 = std::chrono::steady_clock::now();
  auto run = [&](int k) -> int {
    rpa_engine* pe = k == 0 ? e : e->workers[k - 1];
    GateOpener opener{gates[k]};  // whatever happens, the part after this one must not wait forever
    pe->part_mode = true;
    pe->trace_view = e->trace_view; pe->trace_view_n = e->trace_view_n; pe->trace_view_cap = e->trace_view_cap; pe->trace_set_base = s_lo[k];
    if (k) {
      pe->err.clear();
      pe->profiling = e->profiling; pe->prof_counters = e->prof_counters;
      HIP_TRY(pe, hipSetDevice(pe->device));
      HIP_TRY(pe, hipStreamWaitEvent(pe->stream, e->ev_inputs, 0));
      gates[k - 1].wait();  // start underneath the previous part's long alignment launch
    }
    BatchDev pb = b;
    pb.n_sets = s_lo[k + 1] - s_lo[k]; pb.n_records = r_lo[k + 1] - r_lo[k];
    HIP_TRY(pe, pe->part_offsets.ensure((size_t)(pb.n_sets + 1) * 4));
    HIP_TRY(pe, launch_rebase_offsets(pe->stream, b.set_offsets + s_lo[k], pb.n_sets + 1, pe->part_offsets.as<uint32_t>()));
    pb.set_offsets = pe->part_offsets.as<uint32_t>();
    pb.query_seq += s_lo[k]; pb.query_len += s_lo[k];
    pb.qstart += r_lo[k]; pb.qstop += r_lo[k]; pb.rstart += r_lo[k]; pb.rstop += r_lo[k]; pb.score += r_lo[k];
    pb.identities += r_lo[k]; pb.alnlen += r_lo[k]; pb.ref_seq += r_lo[k]; pb.node += r_lo[k];
    if (pb.filtered) pb.filtered += r_lo[k];
    return predict_part(pe, pb, q, db, (char*)rows_dev + (size_t)s_lo[k] * sizeof(rpa_prediction),
                        stats_dev ? (char*)stats_dev + (size_t)s_lo[k] * sizeof(rpa_stats) : nullptr, gates[k]);
  };
  std::vector<std::thread> threads;
  for (int k = 1; k < K; ++k) threads.emplace_back([&, k] { rcs[k] = run(k); });
  rcs[0] = run(0);
  for (auto& t : threads) t.join();
  e->part_mode = false;
  for (rpa_engine* w : e->workers) w->part_mode = false;
  if (e->profiling) e->prof.total_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  for (int k = 0; k < K; ++k)
    if (rcs[k] != RPA_OK) {
      if (k) e->err = e->workers[k - 1]->err;
      return rcs[k];
    }
  return RPA_OK;
}

}  // namespace

extern "C" {

int rpa_abi_version(void) { return RPA_ABI_VERSION; }

void rpa_default_params(rpa_params* p) {
  if (!p) return;
  memset(p, 0, sizeof(*p));
  p->alphabet = RPA_DNA;
  p->exclude_alignments_factor = 0.5f;  // taxator.cpp:286
  p->reeval_bandwidth = 0.05f;          // taxator.cpp:287
  p->device = 0;
  p->speculation_pass1 = RPA_SPECULATION_DEFAULT;
  p->speculation_pass2 = RPA_SPECULATION_DEFAULT;
}

const char* rpa_debug_where(void) { return g_where.load(std::memory_order_relaxed); }

const char* rpa_last_error(const rpa_engine* e) { return e ? e->err.c_str() : g_create_error.c_str(); }

int rpa_create(const rpa_taxonomy* tax, const rpa_params* params, rpa_engine** out) {
  if (!tax || !params || !out) return fail(nullptr, RPA_ERR_INVALID, "rpa_create: null argument");
  *out = nullptr;
  if (tax->n_nodes == 0 || !tax->parent || !tax->left || !tax->right || !tax->depth)
    return fail(nullptr, RPA_ERR_INVALID, "rpa_create: empty taxonomy");
  if (params->alphabet != RPA_DNA && params->alphabet != RPA_PROTEIN) return fail(nullptr, RPA_ERR_INVALID, "rpa_create: bad alphabet");
  if (tax->parent[0] != 0 || tax->depth[0] != 0) return fail(nullptr, RPA_ERR_INVALID, "rpa_create: node 0 must be the root");
  for (uint32_t i = 0; i < tax->n_nodes; ++i) {
    if (tax->depth[i] > kMaxDepth) return fail(nullptr, RPA_ERR_UNSUPPORTED, "rpa_create: taxonomy deeper than %d levels", kMaxDepth);
    if (tax->parent[i] >= tax->n_nodes) return fail(nullptr, RPA_ERR_INVALID, "rpa_create: parent index out of range");
    if (i && !(tax->left[tax->parent[i]] < tax->left[i] && tax->right[tax->parent[i]] > tax->right[i]))
      return fail(nullptr, RPA_ERR_INVALID, "rpa_create: nested-set values of node %u are not inside its parent's", i);
  }
  setenv("GPU_MAX_HW_QUEUES", "16", 0);  // a hardware queue per stream of the jobs in flight (make_stream); no effect once HIP is up
  int n_dev = 0;
  if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0)
    return fail(nullptr, RPA_ERR_DEVICE, "rpa_create: no HIP device visible (this engine has no CPU path)");
  if (params->device < 0 || params->device >= n_dev) return fail(nullptr, RPA_ERR_DEVICE, "rpa_create: device %d of %d", params->device, n_dev);
  rpa_engine* e = new (std::nothrow) rpa_engine();
  if (!e) return fail(nullptr, RPA_ERR_DEVICE, "rpa_create: out of host memory");
  e->device = params->device;
  e->params.alphabet = params->alphabet;
  e->params.exclude_factor = params->exclude_alignments_factor;
  e->params.reeval_factor = (float)(1. - params->reeval_bandwidth);  // reeval_bandwidth_factor_(1. - reeval_bandwidth), :334
  // speculation chunks (hits whose alignment is requested per round of a pass-1 / pass-2 loop): results never depend on
  // them, only the number of alignments computed beyond what the reference demands and the number of rounds
  e->params.spec1 = params->speculation_pass1 == RPA_SPECULATION_DEFAULT ? 0u : params->speculation_pass1;  // the whole loop, cut by chunk_end_by_score (tools/spec_sweep9.sh; protein: 4 hits until round 3)
  e->params.spec2 = params->speculation_pass2 == RPA_SPECULATION_DEFAULT ? (params->alphabet == RPA_DNA ? 10u : 10u) : params->speculation_pass2;  // tools/spec_sweep.sh, tools/spec_sweep4.sh
  e->trace_rounds = getenv("RPA_TRACE_ROUNDS") != nullptr;
  if (const char* v = getenv("RPA_ROW_LAYOUT")) e->linear_rows = !strcmp(v, "linear");
  if (const char* v = getenv("RPA_POISON")) e->poison = atoi(v) != 0;
  if (const char* v = getenv("RPA_LABEL_BITS")) e->force_label_shift = atoi(v) == 32 ? 5u : atoi(v) == 16 ? 4u : atoi(v) == 8 ? 3u : 0u;
  if (const char* v = getenv("RPA_PARTS")) e->parts_override = atoi(v);
  if (const char* v = getenv("RPA_SHORT_WAVE_MAX")) e->short_wave_max = (uint32_t)strtoul(v, nullptr, 10);
  e->params.force_bin = -1;
  if (const char* v = getenv("RPA_FORCE_BIN")) e->params.force_bin = atoi(v);
  e->params.protein_lanes = 1;
  if (const char* v = getenv("RPA_PROTEIN_LANES")) e->params.protein_lanes = atoi(v) != 0;
  e->params.protein_wide = 1;
  if (const char* v = getenv("RPA_PROTEIN_WIDE")) e->params.protein_wide = atoi(v) != 0;
  e->params.no_combined = getenv("RPA_NO_COMBINED") ? 1 : 0;
  if (const char* v = getenv("RPA_BAND4_PHASES")) e->band4_wide = atoi(v) == 2 ? -1 : -2;
  e->params.filter1 = 2;  // (protein only.  3 = nucleotide too, where the assumed cut-off trims a quarter of a chunk or more and one set in sixteen of the batch is like that: references that
                          // share half their residues 1.7-2.2 x faster, config 2 1.7 % slower -- profiles/r05_dna_divergence.txt; 4 = always)
  if (const char* v = getenv("RPA_PASS1_FILTER")) e->params.filter1 = atoi(v);
  // a paused loop's next chunk is as long as what the loop has walked so far (1: on since round 5 -- record sets of 100 / 200 / 400 hits take
  // 10-11 rounds instead of 15 / 19 / 20, -7 / -9 / -6 % per step; the BASELINE shapes do not notice: profiles/r05_dna_hits.txt)
  e->params.grow = 1;
  if (const char* v = getenv("RPA_SPEC_GROW")) e->params.grow = atoi(v);
  if (const char* v = getenv("RPA_BAND")) e->band_long = atoi(v) != 0;
  if (const char* v = getenv("RPA_BAND_SHORT")) e->band_short = atoi(v) != 0;
  if (const char* v = getenv("RPA_BAND_SHORT_MIN")) e->band_short_min = (uint32_t)atoi(v);
  if (const char* v = getenv("RPA_BAND_WINDOW")) e->band_window = atoi(v) != 0;
  if (const char* v = getenv("RPA_BAND_WINDOW_MIN")) e->band_window_min = (uint32_t)atoi(v);
  if (const char* v = getenv("RPA_BAND_WINDOW_WIDE")) { if (atoi(v) != 0) for (int k = 0; k < kLadderBins; ++k) e->band_window_wide[k] = -1; }
  if (const char* v = getenv("RPA_WINDOW_FAIL_WAVE_MAX")) e->window_fail_wave_max = (uint32_t)atoi(v);
  if (const char* v = getenv("RPA_BIG_SET_MIN")) e->big_set_min = (uint32_t)atoi(v);
  if (const char* v = getenv("RPA_BAND_GAIN")) e->band_gain = atof(v);
  if (const char* v = getenv("RPA_BAND_MARGIN")) e->band_margin = (uint32_t)atoi(v);
  if (const char* v = getenv("RPA_LONG_KERNEL")) e->long_kernel = !strcmp(v, "wave") ? 1 : !strcmp(v, "lane") ? 2 : 0;
  e->fixed_speculation = params->speculation_pass1 != RPA_SPECULATION_DEFAULT || params->speculation_pass2 != RPA_SPECULATION_DEFAULT;
  if (const char* v = getenv("RPA_SPEC1")) { e->params.spec1 = (uint32_t)atoi(v); e->fixed_speculation = true; }
  if (const char* v = getenv("RPA_SPEC2")) { e->params.spec2 = (uint32_t)atoi(v); e->fixed_speculation = true; }
  e->params.spec2_first = 0;
  if (const char* v = getenv("RPA_SPEC2_FIRST")) e->params.spec2_first = (uint32_t)atoi(v);
  if (const char* v = getenv("RPA_SIDE_STREAMS")) e->side_streams_on = atoi(v) != 0;
  if (const char* v = getenv("RPA_DEBUG_RANK_ONLY")) e->rank_only = atoi(v) != 0;
  if (const char* v = getenv("RPA_DEBUG_BINS")) e->debug_bins = atoi(v) != 0;
  if (const char* v = getenv("RPA_ALIGN_PRIORITY")) e->align_low_priority = atoi(v) != 0;
  e->params.defer2 = 0;
  if (const char* v = getenv("RPA_DEFER2")) e->defer_mode = !strcmp(v, "auto") ? 2 : (atoi(v) != 0 ? 1 : 0);
  e->fill_lanes = 3u * kResidentLanes;  // RPA_FILL_LANES sweep: config 5 1174 / 1137 / 1053 ms at 1x / 2x / 3x, configs 2 and 4 indifferent
  if (const char* v = getenv("RPA_FILL_LANES")) e->fill_lanes = (uint32_t)atoi(v);
  if (const char* v = getenv("RPA_SPARSE_EMIT")) e->sparse_div_emit = (uint32_t)atoi(v);
  if (const char* v = getenv("RPA_SPARSE_DECIDE")) e->sparse_div_decide = (uint32_t)atoi(v);
  if (getenv("RPA_SPEC_ADAPTIVE")) e->fixed_speculation = false;  // sweeps of the default chunk sizes: keep the per-round adaptation
  auto bail = [&](int code) { g_create_error = e->err; rpa_destroy(e); return code; };
#define CREATE_TRY(call)                                                                                  \
  do {                                                                                                    \
    hipError_t _err = (call);                                                                             \
    if (_err != hipSuccess) { fail(e, RPA_ERR_DEVICE, "%s failed: %s", #call, hipGetErrorString(_err)); return bail(RPA_ERR_DEVICE); } \
  } while (0)
  RPA_WHERE("rpa_create: hipSetDevice (runtime initialisation)");
  CREATE_TRY(hipSetDevice(e->device));
  hipDeviceProp_t prop;
  CREATE_TRY(hipGetDeviceProperties(&prop, e->device));
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
    fail(e, RPA_ERR_DEVICE, "rpa_create: device %d is %s; this library carries gfx950 code only", e->device, prop.gcnArchName);
    return bail(RPA_ERR_DEVICE);
  }
  RPA_WHERE("rpa_create: stream");
  CREATE_TRY(make_stream(e->device, &e->own_stream));
  e->stream = e->own_stream;
  RPA_WHERE("rpa_create: taxonomy copies");
  CREATE_TRY(hipHostMalloc((void**)&e->h_counters, sizeof(uint32_t) * (kNumCounters + 1), hipHostMallocDefault));
  const size_t n = tax->n_nodes;
  CREATE_TRY(e->tax_parent.ensure(n * 4)); CREATE_TRY(e->tax_left.ensure(n * 4)); CREATE_TRY(e->tax_right.ensure(n * 4));
  CREATE_TRY(e->tax_depth.ensure(n));
  CREATE_TRY(hipMemcpy(e->tax_parent.p, tax->parent, n * 4, hipMemcpyHostToDevice));
  CREATE_TRY(hipMemcpy(e->tax_left.p, tax->left, n * 4, hipMemcpyHostToDevice));
  CREATE_TRY(hipMemcpy(e->tax_right.p, tax->right, n * 4, hipMemcpyHostToDevice));
  CREATE_TRY(hipMemcpy(e->tax_depth.p, tax->depth, n, hipMemcpyHostToDevice));
  // Preorder numbering from the parent pointers alone (children in index order, iterative walk): pre[x] = position of x,
  // last[x] = position of the last node of x's subtree.  Then, per node, the ancestors on its root path (ids for the
  // output, preorder intervals for k_rank_records' path labels).
  uint32_t max_depth = 1;
  for (uint32_t i = 0; i < tax->n_nodes; ++i) max_depth = tax->depth[i] > max_depth ? tax->depth[i] : max_depth;
  const size_t D = max_depth;
  {
    std::vector<uint32_t> child_start(n + 1, 0), child(n ? n - 1 : 0), pre(n, 0), last(n, 0);
    for (uint32_t i = 1; i < tax->n_nodes; ++i) {
      if (tax->parent[i] == i) { fail(e, RPA_ERR_INVALID, "rpa_create: node %u is its own parent (only node 0 may be)", i); return bail(RPA_ERR_INVALID); }
      ++child_start[tax->parent[i] + 1];
    }
    for (size_t i = 0; i < n; ++i) child_start[i + 1] += child_start[i];
    {
      std::vector<uint32_t> fill(child_start.begin(), child_start.end() - 1);
      for (uint32_t i = 1; i < tax->n_nodes; ++i) child[fill[tax->parent[i]]++] = i;
    }
    std::vector<uint32_t> stack, next_child(n, 0);
    stack.reserve(max_depth + 2);
    stack.push_back(0);
    uint32_t counter = 0, visited = 1;
    pre[0] = counter++;
    while (!stack.empty()) {
      const uint32_t v = stack.back();
      const uint32_t k = child_start[v] + next_child[v];
      if (k < child_start[v + 1]) {
        ++next_child[v];
        const uint32_t c = child[k];
        if (tax->depth[c] != tax->depth[v] + 1u) { fail(e, RPA_ERR_INVALID, "rpa_create: depth of node %u is not depth(parent)+1", c); return bail(RPA_ERR_INVALID); }
        pre[c] = counter++;
        ++visited;
        stack.push_back(c);
      } else {
        last[v] = counter - 1;
        stack.pop_back();
      }
    }
    if (visited != tax->n_nodes) { fail(e, RPA_ERR_INVALID, "rpa_create: %u node(s) do not descend from node 0", tax->n_nodes - visited); return bail(RPA_ERR_INVALID); }
    std::vector<uint32_t> anc(n * D, 0xFFFFFFFFu);
    const size_t Dp = (D + 7) & ~(size_t)7;  // interval rows in whole chunks of eight levels (64 bytes): k_rank_records reads them 16 bytes at a time
    std::vector<uint32_t> iv(n * Dp * 2);
    for (size_t k = 0; k < n * Dp; ++k) { iv[2 * k] = 0xFFFFFFFFu; iv[2 * k + 1] = 0u; }  // no ancestor at this depth: nothing is inside
    for (uint32_t i = 0; i < tax->n_nodes; ++i) {
      uint32_t v = i;
      for (uint32_t d = tax->depth[i]; d >= 1; --d) {
        anc[(size_t)i * D + (d - 1)] = v;
        iv[2 * ((size_t)i * Dp + (d - 1))] = pre[v];
        iv[2 * ((size_t)i * Dp + (d - 1)) + 1] = last[v] - pre[v];  // x is in the subtree  <=>  pre[x] - pre[v] <= last[v] - pre[v] (unsigned)
        v = tax->parent[v];
      }
    }
    CREATE_TRY(e->tax_pre.ensure(n * 4)); CREATE_TRY(e->tax_last.ensure(n * 4));
    CREATE_TRY(e->tax_anc.ensure(n * D * 4)); CREATE_TRY(e->tax_anc_iv.ensure(n * Dp * 8));
    CREATE_TRY(hipMemcpy(e->tax_pre.p, pre.data(), n * 4, hipMemcpyHostToDevice));
    CREATE_TRY(hipMemcpy(e->tax_last.p, last.data(), n * 4, hipMemcpyHostToDevice));
    CREATE_TRY(hipMemcpy(e->tax_anc.p, anc.data(), n * D * 4, hipMemcpyHostToDevice));
    CREATE_TRY(hipMemcpy(e->tax_anc_iv.p, iv.data(), n * Dp * 8, hipMemcpyHostToDevice));
  }
  e->tax = TaxDev{e->tax_parent.as<const uint32_t>(), e->tax_left.as<const uint32_t>(), e->tax_right.as<const uint32_t>(),
                  e->tax_depth.as<const uint8_t>(), tax->n_nodes, e->tax_pre.as<const uint32_t>(), e->tax_last.as<const uint32_t>(),
                  e->tax_anc.as<const uint32_t>(), e->tax_anc_iv.as<const uint2>(), max_depth};
  CREATE_TRY(e->prof_acc.ensure(kProfParts * kProfSlots * sizeof(unsigned long long)));
  CREATE_TRY(hipMemset(e->prof_acc.p, 0, kProfParts * kProfSlots * sizeof(unsigned long long)));
#undef CREATE_TRY
  RPA_WHERE("rpa_create: done");
  *out = e;
  return RPA_OK;
}

void rpa_destroy(rpa_engine* e) {
  if (!e) return;
  for (rpa_engine* s : e->slots) {
    RPA_WHERE("rpa_destroy: joining a job slot thread");
    { std::lock_guard<std::mutex> lk(s->job_mu); s->job_quit = true; }
    s->job_cv.notify_all();
    if (s->job_thread.joinable()) s->job_thread.join();
    rpa_destroy(s);
  }
  e->slots.clear();
  for (rpa_engine* w : e->workers) rpa_destroy(w);
  e->workers.clear();
  (void)hipSetDevice(e->device);
  RPA_WHERE("rpa_destroy: sync own stream");
  if (e->own_stream) (void)hipStreamSynchronize(e->own_stream);
  DevBuf* bufs[] = {&e->tax_parent, &e->tax_left, &e->tax_right, &e->tax_depth, &e->tax_pre, &e->tax_last, &e->tax_anc, &e->tax_anc_iv, &e->loop_save, &e->rec_pre, &e->trace_buf, &e->trace_count, &e->in_f32, &e->in_u8, &e->in_f64, &e->in_uncl, &e->state, &e->hits,
                    &e->set_info, &e->seg,
                    &e->pair_a, &e->pair_b, &e->emit, &e->emit_lite, &e->emit_ext, &e->hit_set, &e->block_first, &e->flags, &e->lite, &e->tile_base, &e->res_slot, &e->pair_res, &e->pair_ext, &e->bin_list, &e->counters,
                    &e->og_idx, &e->og_dist, &e->band_save, &e->live_a, &e->live_b, &e->out_rows, &e->out_stats, &e->tmp_letters,
                    &e->protein_ws, &e->retry_list, &e->retry_short, &e->window_fail, &e->big_sets, &e->prof_acc, &e->part_offsets};
  RPA_WHERE("rpa_destroy: hipFree of the buffers");
  for (DevBuf* b : bufs) b->release();
  for (auto& b : e->in_u32) b.release();
  for (auto& s : e->stores) s.release();
  e->pair_store.release();
  RPA_WHERE("rpa_destroy: events");
  for (hipEvent_t ev : e->ev_pool) (void)hipEventDestroy(ev);
  if (e->ev_total0) (void)hipEventDestroy(e->ev_total0);
  if (e->ev_total1) (void)hipEventDestroy(e->ev_total1);
  if (e->ev_inputs) (void)hipEventDestroy(e->ev_inputs);
  for (int k = 0; k < rpa_engine::kSideStreams; ++k) {
    if (e->side_stream[k]) { (void)hipStreamSynchronize(e->side_stream[k]); (void)hipStreamDestroy(e->side_stream[k]); }
    if (e->ev_join[k]) (void)hipEventDestroy(e->ev_join[k]);
    e->side_ws[k].release();
  }
  if (e->ev_fork) (void)hipEventDestroy(e->ev_fork);
  if (e->lo_stream) { (void)hipStreamSynchronize(e->lo_stream); (void)hipStreamDestroy(e->lo_stream); }
  if (e->ev_lo_begin) (void)hipEventDestroy(e->ev_lo_begin);
  if (e->ev_lo_end) (void)hipEventDestroy(e->ev_lo_end);
  RPA_WHERE("rpa_destroy: hipHostFree");
  if (e->h_counters) (void)hipHostFree(e->h_counters);
  RPA_WHERE("rpa_destroy: hipStreamDestroy");
  if (e->own_stream) (void)hipStreamDestroy(e->own_stream);
  RPA_WHERE("rpa_destroy: done");
  delete e;
}

int rpa_set_stream(rpa_engine* e, void* hip_stream) {
  if (!e) return RPA_ERR_INVALID;
  std::lock_guard<std::mutex> lk(e->mu);
  e->stream = hip_stream ? (hipStream_t)hip_stream : e->own_stream;
  return RPA_OK;
}

static void quiesce_slots(rpa_engine* e);

// Make room for `more` residues behind the ones in use (rounded up to whole words); the arena grows by doubling, its
// contents move with a device-to-device copy.  Returns the residue index the new residues start at (a multiple of 8).
static int store_grow(rpa_engine* e, Store* st, uint64_t more, uint64_t* first_residue) {
  const int bits = e->params.alphabet == RPA_DNA ? 4 : 8;
  const uint64_t start = (st->n_residues + 7u) & ~7ull;
  const uint64_t need_words = ((start + more) * bits + 31) / 32;
  if (need_words > st->cap_words || !st->words.p) {
    uint64_t cap = st->cap_words ? st->cap_words : 0;
    while (cap < need_words) cap = cap ? cap * 2 : need_words;
    DevBuf bigger;
    HIP_TRY(e, bigger.ensure((size_t)(cap + kLeadWords + kTailWords) * 4));
    cap = (bigger.cap / 4) - kLeadWords - kTailWords;  // what ensure() really gave
    HIP_TRY(e, hipMemsetAsync(bigger.p, 0, kLeadWords * 4, e->stream));
    const uint64_t used_words = (st->n_residues * bits + 31) / 32;
    if (used_words) HIP_TRY(e, hipMemcpyAsync(bigger.as<uint32_t>() + kLeadWords, st->words.as<uint32_t>() + kLeadWords, (size_t)used_words * 4, hipMemcpyDeviceToDevice, e->stream));
    // everything behind the words in use reads as zero (slack for 16-byte text loads and funnel shifts)
    HIP_TRY(e, hipMemsetAsync(bigger.as<uint32_t>() + kLeadWords + used_words, 0, (size_t)(cap - used_words + kTailWords) * 4, e->stream));
    RPA_WHERE("store_grow: sync after move");
    HIP_TRY(e, hipStreamSynchronize(e->stream));
    st->words.release();
    st->words = bigger;
    st->cap_words = cap;
  }
  *first_residue = start;
  return RPA_OK;
}

// letters of `n` sequences (offsets: n + 1 entries from 0) -> packed behind the arena's residues in use; extent table
// entries index[k] (or k when index is null) point at them.
static int store_add(rpa_engine* e, Store* st, uint32_t n, const uint32_t* index, const uint8_t* letters, const uint64_t* offsets, int mem_kind,
                     DevBuf* letters_buf) {
  uint64_t total = 0;
  HIP_TRY(e, st->tmp_offsets.ensure((size_t)(n + 1) * 8));
  if (mem_kind == RPA_MEM_HOST) {
    for (uint32_t i = 0; i < n; ++i)
      if (offsets[i + 1] < offsets[i]) return fail(e, RPA_ERR_INVALID, "store offsets must be non-decreasing (sequence %u)", i);
    if (offsets[0] != 0) return fail(e, RPA_ERR_INVALID, "store offsets[0] must be 0");
    total = offsets[n];
    HIP_TRY(e, hipMemcpyAsync(st->tmp_offsets.p, offsets, (size_t)(n + 1) * 8, hipMemcpyHostToDevice, e->stream));
  } else {
    HIP_TRY(e, hipMemcpyAsync(st->tmp_offsets.p, offsets, (size_t)(n + 1) * 8, hipMemcpyDeviceToDevice, e->stream));
    HIP_TRY(e, hipMemcpyAsync(&total, offsets + n, 8, hipMemcpyDeviceToHost, e->stream));
    RPA_WHERE("store_add: sync after offsets");
    HIP_TRY(e, hipStreamSynchronize(e->stream));
  }
  const uint32_t* dev_index = nullptr;
  if (index) {
    HIP_TRY(e, st->tmp_index.ensure((size_t)n * 4));
    HIP_TRY(e, hipMemcpyAsync(st->tmp_index.p, index, (size_t)n * 4, mem_kind == RPA_MEM_HOST ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice, e->stream));
    dev_index = st->tmp_index.as<const uint32_t>();
  }
  uint64_t first = 0;
  int rc = store_grow(e, st, total, &first);
  if (rc) return rc;
  const int bits = e->params.alphabet == RPA_DNA ? 4 : 8;
  const uint8_t* dev_letters = letters;
  if (mem_kind == RPA_MEM_HOST && total) {
    HIP_TRY(e, letters_buf->ensure((size_t)total + 8));
    HIP_TRY(e, hipMemcpyAsync(letters_buf->p, letters, (size_t)total, hipMemcpyHostToDevice, e->stream));
    dev_letters = letters_buf->as<const uint8_t>();
  } else if (total && ((uintptr_t)letters & 7u)) {
    return fail(e, RPA_ERR_INVALID, "device letter buffer must be 8-byte aligned");
  }
  HIP_TRY(e, launch_pack_letters(e->stream, e->params.alphabet, dev_letters, total, st->words.as<uint32_t>() + kLeadWords + first * bits / 32));
  HIP_TRY(e, launch_store_extents(e->stream, st->tmp_offsets.as<const uint64_t>(), dev_index, n, first, st->extent.as<ulonglong2>()));
  RPA_WHERE("store_add: sync after pack");
  HIP_TRY(e, hipStreamSynchronize(e->stream));
  st->n_residues = first + total;
  st->n_resident += n;
  return RPA_OK;
}

static void forget_windows(Store* st) {
  st->windowed = false; st->seq_length.clear(); st->seq_windows.clear(); st->pool_used = 0; st->n_windows = 0; st->live_residues = 0;
}

static int pack_store(rpa_engine* e, Store* st, uint32_t n_seqs, const uint8_t* letters, const uint64_t* offsets, int mem_kind,
                      DevBuf* letters_buf) {
  forget_windows(st);
  st->n_residues = 0; st->n_resident = 0; st->n_seqs = n_seqs;
  HIP_TRY(e, st->extent.ensure((size_t)(n_seqs ? n_seqs : 1) * sizeof(ulonglong2)));
  int rc = store_add(e, st, n_seqs, nullptr, letters, offsets, mem_kind, letters_buf);
  if (rc) return rc;
  st->loaded = true;
  return RPA_OK;
}

int rpa_store_load(rpa_engine* e, int which, uint32_t n_seqs, const uint8_t* letters, const uint64_t* offsets, int mem_kind) {
  if (!e) return RPA_ERR_INVALID;
  std::lock_guard<std::mutex> lk(e->mu);
  if (which != RPA_STORE_QUERY && which != RPA_STORE_DB) return fail(e, RPA_ERR_INVALID, "rpa_store_load: bad store id");
  if (!offsets || (n_seqs && !letters)) return fail(e, RPA_ERR_INVALID, "rpa_store_load: null argument");
  if (mem_kind != RPA_MEM_HOST && mem_kind != RPA_MEM_DEVICE) return fail(e, RPA_ERR_INVALID, "rpa_store_load: bad mem_kind");
  HIP_TRY(e, hipSetDevice(e->device));
  quiesce_slots(e);
  int rc = pack_store(e, &e->stores[which], n_seqs, letters, offsets, mem_kind, &e->tmp_letters);
  e->tmp_letters.release();  // the raw letters are not kept: the packed store is the resident copy
  return rc;
}

int rpa_store_reserve(rpa_engine* e, int which, uint32_t n_seqs) {
  if (!e) return RPA_ERR_INVALID;
  std::lock_guard<std::mutex> lk(e->mu);
  if (which != RPA_STORE_QUERY && which != RPA_STORE_DB) return fail(e, RPA_ERR_INVALID, "rpa_store_reserve: bad store id");
  HIP_TRY(e, hipSetDevice(e->device));
  quiesce_slots(e);
  Store* st = &e->stores[which];
  forget_windows(st);
  st->n_residues = 0; st->n_resident = 0; st->n_seqs = n_seqs;
  HIP_TRY(e, st->extent.ensure((size_t)(n_seqs ? n_seqs : 1) * sizeof(ulonglong2)));
  HIP_TRY(e, hipMemsetAsync(st->extent.p, 0, (size_t)(n_seqs ? n_seqs : 1) * sizeof(ulonglong2), e->stream));  // nothing resident: every length 0
  uint64_t first = 0;
  int rc = store_grow(e, st, 0, &first);  // an (empty) arena with its slack words, so that a view exists
  if (rc) return rc;
  RPA_WHERE("rpa_store_reserve: sync");
  HIP_TRY(e, hipStreamSynchronize(e->stream));
  st->loaded = true;
  return RPA_OK;
}

int rpa_store_append(rpa_engine* e, int which, uint32_t n, const uint32_t* seq_index, const uint8_t* letters, const uint64_t* offsets) {
  if (!e) return RPA_ERR_INVALID;
  std::lock_guard<std::mutex> lk(e->mu);
  if (which != RPA_STORE_QUERY && which != RPA_STORE_DB) return fail(e, RPA_ERR_INVALID, "rpa_store_append: bad store id");
  Store* st = &e->stores[which];
  if (!st->loaded) return fail(e, RPA_ERR_INVALID, "rpa_store_append: reserve (or load) the store first");
  if (st->windowed) return fail(e, RPA_ERR_INVALID, "rpa_store_append: this store takes windows (rpa_store_append_windows)");
  if (n == 0) return RPA_OK;
  if (!seq_index || !offsets || !letters) return fail(e, RPA_ERR_INVALID, "rpa_store_append: null argument");
  for (uint32_t k = 0; k < n; ++k)
    if (seq_index[k] >= st->n_seqs) return fail(e, RPA_ERR_INVALID, "rpa_store_append: sequence index %u outside the store (%u sequences)", seq_index[k], st->n_seqs);
  HIP_TRY(e, hipSetDevice(e->device));
  quiesce_slots(e);  // the arena may move: no job may be reading it
  int rc = store_add(e, st, n, seq_index, letters, offsets, RPA_MEM_HOST, &e->tmp_letters);
  return rc;
}

int rpa_store_reserve_windows(rpa_engine* e, int which, uint32_t n_seqs, const uint64_t* lengths) {
  if (!e) return RPA_ERR_INVALID;
  std::lock_guard<std::mutex> lk(e->mu);
  if (which != RPA_STORE_QUERY && which != RPA_STORE_DB) return fail(e, RPA_ERR_INVALID, "rpa_store_reserve_windows: bad store id");
  if (n_seqs && !lengths) return fail(e, RPA_ERR_INVALID, "rpa_store_reserve_windows: null argument");
  for (uint32_t i = 0; i < n_seqs; ++i)
    if (lengths[i] > 0xFFFFFFFFull) return fail(e, RPA_ERR_UNSUPPORTED, "rpa_store_reserve_windows: sequence %u has %llu residues (batch coordinates are 32 bit)", i, (unsigned long long)lengths[i]);
  HIP_TRY(e, hipSetDevice(e->device));
  quiesce_slots(e);
  Store* st = &e->stores[which];
  forget_windows(st);
  st->n_residues = 0; st->n_resident = 0; st->n_seqs = n_seqs;
  st->windowed = true;
  st->seq_length.assign(lengths, lengths + n_seqs);
  st->seq_windows.assign(n_seqs, std::vector<uint4>());
  const size_t ns = n_seqs ? n_seqs : 1;
  HIP_TRY(e, st->extent.ensure(ns * sizeof(ulonglong2)));
  HIP_TRY(e, st->win_count.ensure(ns * 4));
  HIP_TRY(e, st->win_pool.ensure(1024 * sizeof(uint4)));
  st->pool_cap = st->win_pool.cap / sizeof(uint4);
  HIP_TRY(e, st->tmp_offsets.ensure(ns * 8));
  if (n_seqs) HIP_TRY(e, hipMemcpyAsync(st->tmp_offsets.p, lengths, (size_t)n_seqs * 8, hipMemcpyHostToDevice, e->stream));
  HIP_TRY(e, launch_store_lengths(e->stream, st->tmp_offsets.as<const uint64_t>(), n_seqs, st->extent.as<ulonglong2>(), st->win_count.as<uint32_t>()));
  uint64_t first = 0;
  int rc = store_grow(e, st, 0, &first);  // an (empty) arena with its slack words, so that a view exists
  if (rc) return rc;
  RPA_WHERE("rpa_store_reserve_windows: sync");
  HIP_TRY(e, hipStreamSynchronize(e->stream));
  st->loaded = true;
  return RPA_OK;
}

int rpa_store_append_windows(rpa_engine* e, int which, uint32_t n, const rpa_window* win, const uint8_t* letters, const uint64_t* offsets) {
  if (!e) return RPA_ERR_INVALID;
  std::lock_guard<std::mutex> lk(e->mu);
  if (which != RPA_STORE_QUERY && which != RPA_STORE_DB) return fail(e, RPA_ERR_INVALID, "rpa_store_append_windows: bad store id");
  Store* st = &e->stores[which];
  if (!st->loaded || !st->windowed) return fail(e, RPA_ERR_INVALID, "rpa_store_append_windows: reserve the store with rpa_store_reserve_windows first");
  if (n == 0) return RPA_OK;
  if (!win || !offsets || !letters) return fail(e, RPA_ERR_INVALID, "rpa_store_append_windows: null argument");
  if (offsets[0] != 0) return fail(e, RPA_ERR_INVALID, "rpa_store_append_windows: offsets[0] must be 0");
  for (uint32_t k = 0; k < n; ++k) {
    if (win[k].seq_index >= st->n_seqs) return fail(e, RPA_ERR_INVALID, "rpa_store_append_windows: sequence index %u outside the store (%u sequences)", win[k].seq_index, st->n_seqs);
    if (offsets[k + 1] < offsets[k] || offsets[k + 1] - offsets[k] != win[k].n) return fail(e, RPA_ERR_INVALID, "rpa_store_append_windows: window %u has %llu residues, its letters %llu", k, (unsigned long long)win[k].n, (unsigned long long)(offsets[k + 1] - offsets[k]));
    if (win[k].n == 0 || win[k].first + win[k].n > st->seq_length[win[k].seq_index]) return fail(e, RPA_ERR_INVALID, "rpa_store_append_windows: window %u is empty or reaches beyond its sequence", k);
  }
  // bookkeeping first, on copies: a window may sit beside the resident windows of its sequence or cover some of them
  // entirely (those are dropped: the new window is their hull), never overlap one partly -- a segment is only readable
  // from ONE window, so overlapping windows are the caller's mistake.  One merge of two sorted lists per sequence (round 5:
  // until then every new window rebuilt its sequence's whole list -- quadratic for a sequence that collects thousands of small
  // windows, which is what a 10 Mbp chromosome does under `taxator --store-tile 2048`).
  struct WinRec { uint4 w; int64_t k; };  // k >= 0: window number k of THIS call (its arena offset is filled in below); -1: resident
  std::vector<uint32_t> changed;
  std::vector<std::vector<WinRec>> lists;
  uint64_t dropped_residues = 0, dropped_windows = 0;
  std::vector<uint32_t> order(n);
  for (uint32_t k = 0; k < n; ++k) order[k] = k;
  std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return win[a].seq_index != win[b].seq_index ? win[a].seq_index < win[b].seq_index : win[a].first < win[b].first; });
  for (uint32_t oi = 0; oi < n;) {
    const uint32_t seq = win[order[oi]].seq_index;
    uint32_t oj = oi;
    while (oj < n && win[order[oj]].seq_index == seq) ++oj;  // the new windows [oi, oj) of this sequence, by first residue
    for (uint32_t t = oi + 1; t < oj; ++t)
      if (win[order[t - 1]].first + win[order[t - 1]].n > win[order[t]].first)
        return fail(e, RPA_ERR_INVALID, "rpa_store_append_windows: windows %u and %u of this call overlap", order[t - 1], order[t]);
    changed.push_back(seq);
    lists.emplace_back();
    std::vector<WinRec>& out = lists.back();
    const std::vector<uint4>& have = st->seq_windows[seq];
    out.reserve(have.size() + (oj - oi));
    uint32_t t = oi;  // next new window that has not been put out
    for (const uint4& x : have) {
      const uint64_t xa = x.x, xb = (uint64_t)x.x + x.y;
      while (t < oj && win[order[t]].first + win[order[t]].n <= xa) {  // new windows that end in front of x
        out.push_back(WinRec{make_uint4((uint32_t)win[order[t]].first, (uint32_t)win[order[t]].n, 0u, 0u), (int64_t)order[t]});
        ++t;
      }
      if (t < oj && win[order[t]].first < xb) {  // x and the next new window share residues: x must lie inside it
        const uint64_t a = win[order[t]].first, b = a + win[order[t]].n;
        if (xa >= a && xb <= b) { dropped_residues += x.y; ++dropped_windows; continue; }  // swallowed: its residues stay behind in the arena
        return fail(e, RPA_ERR_INVALID, "rpa_store_append_windows: window %u (sequence %u, residues %llu..%llu) overlaps a resident window partly", order[t], seq,
                    (unsigned long long)a, (unsigned long long)b);
      }
      out.push_back(WinRec{x, -1});
    }
    for (; t < oj; ++t) out.push_back(WinRec{make_uint4((uint32_t)win[order[t]].first, (uint32_t)win[order[t]].n, 0u, 0u), (int64_t)order[t]});
    if (out.size() >= (1u << 24)) return fail(e, RPA_ERR_UNSUPPORTED, "rpa_store_append_windows: more than 16 M resident windows in one sequence");
    oi = oj;
  }
  HIP_TRY(e, hipSetDevice(e->device));
  const uint64_t total = offsets[n];
  uint64_t need = 0;
  for (auto& l : lists) need += l.size();
  // Jobs in flight read the arena, the pool and the extent table.  An append only ever writes BEHIND what is resident (new residues,
  // new list copies at the pool's end) and then switches a sequence to its new list with ONE 64-bit store (first entry | count,
  // k_store_window_lists) -- a running job sees the old list or the new one, and both hold every window it was made resident
  // for.  So the jobs only have to be waited for when something MOVES: the arena or the pool outgrows its allocation.
  const int bits_now = e->params.alphabet == RPA_DNA ? 4 : 8;
  const bool arena_moves = !st->words.p || ((((st->n_residues + 7u) & ~7ull) + total) * bits_now + 31) / 32 > st->cap_words;  // (store_grow's own test)
  const bool pool_moves = st->pool_used + need > st->pool_cap;
  if (arena_moves || pool_moves) quiesce_slots(e);
  uint64_t first = 0;
  int rc = store_grow(e, st, total, &first);
  if (rc) return rc;
  const int bits = e->params.alphabet == RPA_DNA ? 4 : 8;
  HIP_TRY(e, e->tmp_letters.ensure((size_t)total + 8));
  HIP_TRY(e, hipMemcpyAsync(e->tmp_letters.p, letters, (size_t)total, hipMemcpyHostToDevice, e->stream));
  HIP_TRY(e, launch_pack_letters(e->stream, e->params.alphabet, e->tmp_letters.as<const uint8_t>(), total, st->words.as<uint32_t>() + kLeadWords + first * bits / 32));
  // the changed lists go to the end of the pool
  if (st->pool_used + need > st->pool_cap) {
    uint64_t cap = st->pool_cap ? st->pool_cap : 1024;
    while (cap < st->pool_used + need) cap *= 2;
    DevBuf bigger;
    HIP_TRY(e, bigger.ensure((size_t)cap * sizeof(uint4)));
    if (st->pool_used) HIP_TRY(e, hipMemcpyAsync(bigger.p, st->win_pool.p, (size_t)st->pool_used * sizeof(uint4), hipMemcpyDeviceToDevice, e->stream));
    HIP_TRY(e, hipStreamSynchronize(e->stream));
    st->win_pool.release();
    st->win_pool = bigger;
    st->pool_cap = st->win_pool.cap / sizeof(uint4);
  }
  std::vector<uint4> flat;
  flat.reserve((size_t)need);
  std::vector<uint32_t> firsts(changed.size()), counts(changed.size());
  for (size_t c = 0; c < changed.size(); ++c) {
    firsts[c] = (uint32_t)(st->pool_used + flat.size());
    counts[c] = (uint32_t)lists[c].size();
    for (WinRec& r : lists[c]) {
      if (r.k >= 0) {  // a window of this call: where its residues went
        const uint64_t at = first + offsets[r.k];
        r.w.z = (uint32_t)at; r.w.w = (uint32_t)(at >> 32);
        r.k = -1;
      }
      flat.push_back(r.w);
    }
  }
  if (st->pool_used + flat.size() > 0xFFFFFFFFull) return fail(e, RPA_ERR_UNSUPPORTED, "rpa_store_append_windows: window pool exhausted (rpa_store_reset compacts it)");
  HIP_TRY(e, hipMemcpyAsync(st->win_pool.as<uint4>() + st->pool_used, flat.data(), flat.size() * sizeof(uint4), hipMemcpyHostToDevice, e->stream));
  const size_t nc = changed.size();
  HIP_TRY(e, st->tmp_index.ensure(nc * 4)); HIP_TRY(e, st->tmp_a.ensure(nc * 4)); HIP_TRY(e, st->tmp_b.ensure(nc * 4));
  HIP_TRY(e, hipMemcpyAsync(st->tmp_index.p, changed.data(), nc * 4, hipMemcpyHostToDevice, e->stream));
  HIP_TRY(e, hipMemcpyAsync(st->tmp_a.p, firsts.data(), nc * 4, hipMemcpyHostToDevice, e->stream));
  HIP_TRY(e, hipMemcpyAsync(st->tmp_b.p, counts.data(), nc * 4, hipMemcpyHostToDevice, e->stream));
  HIP_TRY(e, launch_store_window_lists(e->stream, st->tmp_index.as<const uint32_t>(), st->tmp_a.as<const uint32_t>(), st->tmp_b.as<const uint32_t>(), (uint32_t)nc,
                                       st->extent.as<ulonglong2>(), st->win_count.as<uint32_t>()));
  RPA_WHERE("rpa_store_append_windows: sync");
  HIP_TRY(e, hipStreamSynchronize(e->stream));
  // commit
  st->pool_used += flat.size();
  for (size_t c = 0; c < nc; ++c) {
    if (st->seq_windows[changed[c]].empty() && !lists[c].empty()) ++st->n_resident;
    std::vector<uint4>& dst = st->seq_windows[changed[c]];
    dst.clear();
    dst.reserve(lists[c].size());
    for (const WinRec& r : lists[c]) dst.push_back(r.w);
  }
  st->n_residues = first + total;
  st->n_windows += n; st->n_windows -= dropped_windows;
  st->live_residues += total; st->live_residues -= dropped_residues;
  return RPA_OK;
}

int rpa_store_reset(rpa_engine* e, int which) {
  if (!e) return RPA_ERR_INVALID;
  std::lock_guard<std::mutex> lk(e->mu);
  if (which != RPA_STORE_QUERY && which != RPA_STORE_DB) return fail(e, RPA_ERR_INVALID, "rpa_store_reset: bad store id");
  Store* st = &e->stores[which];
  if (!st->loaded) return fail(e, RPA_ERR_INVALID, "rpa_store_reset: nothing to reset");
  HIP_TRY(e, hipSetDevice(e->device));
  quiesce_slots(e);
  // nothing resident any more: the arena and the window pool are refilled from their start (their memory is kept)
  st->n_residues = 0; st->n_resident = 0; st->pool_used = 0; st->n_windows = 0; st->live_residues = 0;
  if (st->windowed) {
    for (auto& l : st->seq_windows) l.clear();
    // every sequence back to "length known, no window": extent = (0, length), count 0 (the kernels read the count out of extent.x)
    HIP_TRY(e, st->tmp_offsets.ensure((size_t)(st->n_seqs ? st->n_seqs : 1) * 8));
    if (st->n_seqs) HIP_TRY(e, hipMemcpyAsync(st->tmp_offsets.p, st->seq_length.data(), (size_t)st->n_seqs * 8, hipMemcpyHostToDevice, e->stream));
    HIP_TRY(e, launch_store_lengths(e->stream, st->tmp_offsets.as<const uint64_t>(), st->n_seqs, st->extent.as<ulonglong2>(), st->win_count.as<uint32_t>()));
  } else {
    HIP_TRY(e, hipMemsetAsync(st->extent.p, 0, (size_t)(st->n_seqs ? st->n_seqs : 1) * sizeof(ulonglong2), e->stream));
  }
  RPA_WHERE("rpa_store_reset: sync");
  HIP_TRY(e, hipStreamSynchronize(e->stream));
  return RPA_OK;
}

int rpa_store_info_get(rpa_engine* e, int which, rpa_store_info* out) {
  if (!e || !out) return RPA_ERR_INVALID;
  std::lock_guard<std::mutex> lk(e->mu);
  if (which != RPA_STORE_QUERY && which != RPA_STORE_DB) return fail(e, RPA_ERR_INVALID, "rpa_store_info_get: bad store id");
  const Store& st = e->stores[which];
  out->n_seqs = st.n_seqs; out->n_resident = st.n_resident; out->residues = st.n_residues;
  out->device_bytes = (uint64_t)st.words.cap + st.extent.cap + st.win_pool.cap + st.win_count.cap;
  out->windows = st.windowed ? st.n_windows : st.n_resident;
  out->live_residues = st.windowed ? st.live_residues : st.n_residues;
  return RPA_OK;
}

// Everything rpa_predict_batch does, on the stream / buffers of `e` with the stores of `so` (e itself, or the owner of a job slot).
static int predict_on(rpa_engine* e, rpa_engine* so, const rpa_batch* batch, rpa_prediction* out, rpa_stats* stats, int out_mem_kind) {
  if (!batch || !out) return fail(e, RPA_ERR_INVALID, "rpa_predict_batch: null argument");
  if (!so->stores[0].loaded || !so->stores[1].loaded) return fail(e, RPA_ERR_INVALID, "rpa_predict_batch: load both sequence stores first");
  const uint32_t S = batch->n_sets, H = batch->n_records;
  if (S == 0) return RPA_OK;
  if (!batch->set_offsets || !batch->query_seq || !batch->query_len ||
      (H && (!batch->qstart || !batch->qstop || !batch->rstart || !batch->rstop || !batch->score || !batch->identities ||
             !batch->alnlen || !batch->ref_seq || !batch->node)))
    return fail(e, RPA_ERR_INVALID, "rpa_predict_batch: null array in batch");
  if ((uint64_t)S + H >= 0xFFFFFFF0ull) return fail(e, RPA_ERR_INVALID, "rpa_predict_batch: batch too large");
  HIP_TRY(e, hipSetDevice(e->device));
  BatchDev b{};
  b.n_sets = S; b.n_records = H;
  if (batch->mem_kind == RPA_MEM_HOST) {
    if (batch->set_offsets[0] != 0 || batch->set_offsets[S] != H) return fail(e, RPA_ERR_INVALID, "set_offsets must run from 0 to n_records");
    const void* src_u32[11] = {batch->set_offsets, batch->query_seq, batch->query_len, batch->qstart, batch->qstop, batch->rstart,
                               batch->rstop, batch->identities, batch->alnlen, batch->ref_seq, batch->node};
    const size_t cnt[11] = {(size_t)S + 1, S, S, H, H, H, H, H, H, H, H};
    for (int k = 0; k < 11; ++k) HIP_TRY(e, e->in_u32[k].ensure((cnt[k] ? cnt[k] : 1) * 4));
    HIP_TRY(e, e->in_f32.ensure((size_t)(H ? H : 1) * 4));
    if (batch->filtered) HIP_TRY(e, e->in_u8.ensure(H ? H : 1));
    {
      // Jobs in flight take turns on the host->device link: left alone, the jobs that were submitted together copy
      // together (each at a share of the PCIe rate) and then compute together -- phases in lockstep, nothing overlaps.
      // One copier at a time makes the copy of job k+1 run beside the kernels of job k.
      std::unique_lock<std::mutex> copy_turn(so->copy_mu, std::defer_lock);
      if (e != so) copy_turn.lock();
      for (int k = 0; k < 11; ++k)
        if (cnt[k]) HIP_TRY(e, hipMemcpyAsync(e->in_u32[k].p, src_u32[k], cnt[k] * 4, hipMemcpyHostToDevice, e->stream));
      if (H) HIP_TRY(e, hipMemcpyAsync(e->in_f32.p, batch->score, (size_t)H * 4, hipMemcpyHostToDevice, e->stream));
      if (batch->filtered && H) HIP_TRY(e, hipMemcpyAsync(e->in_u8.p, batch->filtered, H, hipMemcpyHostToDevice, e->stream));
      RPA_WHERE("predict_on: sync after host->device copies (job slot)");
      if (e != so) HIP_TRY(e, hipStreamSynchronize(e->stream));  // the link is free again when the data has arrived
    }
    b.set_offsets = e->in_u32[0].as<uint32_t>(); b.query_seq = e->in_u32[1].as<uint32_t>(); b.query_len = e->in_u32[2].as<uint32_t>();
    b.qstart = e->in_u32[3].as<uint32_t>(); b.qstop = e->in_u32[4].as<uint32_t>(); b.rstart = e->in_u32[5].as<uint32_t>();
    b.rstop = e->in_u32[6].as<uint32_t>(); b.identities = e->in_u32[7].as<uint32_t>(); b.alnlen = e->in_u32[8].as<uint32_t>();
    b.ref_seq = e->in_u32[9].as<uint32_t>(); b.node = e->in_u32[10].as<uint32_t>();
    b.score = e->in_f32.as<float>();
    b.filtered = batch->filtered ? e->in_u8.as<uint8_t>() : nullptr;
  } else if (batch->mem_kind == RPA_MEM_DEVICE) {
    b.set_offsets = batch->set_offsets; b.query_seq = batch->query_seq; b.query_len = batch->query_len;
    b.qstart = batch->qstart; b.qstop = batch->qstop; b.rstart = batch->rstart; b.rstop = batch->rstop; b.score = batch->score;
    b.identities = batch->identities; b.alnlen = batch->alnlen; b.ref_seq = batch->ref_seq; b.node = batch->node;
    b.filtered = batch->filtered;
  } else return fail(e, RPA_ERR_INVALID, "rpa_predict_batch: bad mem_kind");

  void* rows_dev = out;
  void* stats_dev = stats;
  if (out_mem_kind == RPA_MEM_HOST) {
    HIP_TRY(e, e->out_rows.ensure((size_t)S * sizeof(rpa_prediction)));
    rows_dev = e->out_rows.p;
    if (stats) { HIP_TRY(e, e->out_stats.ensure((size_t)S * sizeof(rpa_stats))); stats_dev = e->out_stats.p; }
  } else if (out_mem_kind != RPA_MEM_DEVICE) return fail(e, RPA_ERR_INVALID, "rpa_predict_batch: bad out_mem_kind");
  int rc;
  if (e == so) rc = run_parts(e, b, batch->mem_kind == RPA_MEM_HOST ? batch->set_offsets : nullptr, rows_dev, stats_dev);
  else {  // a job slot: one part, the owner's stores
    PartGate g;
    rc = predict_part(e, b, so->stores[RPA_STORE_QUERY].view(), so->stores[RPA_STORE_DB].view(), rows_dev, stats_dev, g);
  }
  if (rc) return rc;
  if (out_mem_kind == RPA_MEM_HOST) {
    HIP_TRY(e, hipMemcpyAsync(out, rows_dev, (size_t)S * sizeof(rpa_prediction), hipMemcpyDeviceToHost, e->stream));
    if (stats) HIP_TRY(e, hipMemcpyAsync(stats, stats_dev, (size_t)S * sizeof(rpa_stats), hipMemcpyDeviceToHost, e->stream));
    RPA_WHERE("predict_on: sync after copying the rows back");
    HIP_TRY(e, hipStreamSynchronize(e->stream));
  }
  return RPA_OK;
}

int rpa_predict_batch(rpa_engine* e, const rpa_batch* batch, rpa_prediction* out, rpa_stats* stats, int out_mem_kind) {
  if (!e) return RPA_ERR_INVALID;
  std::lock_guard<std::mutex> lk(e->mu);
  return predict_on(e, e, batch, out, stats, out_mem_kind);
}

int rpa_trace_begin(rpa_engine* e, uint64_t capacity) {
  if (!e) return RPA_ERR_INVALID;
  std::lock_guard<std::mutex> lk(e->mu);
  HIP_TRY(e, hipSetDevice(e->device));
  if (capacity) {
    HIP_TRY(e, e->trace_buf.ensure((size_t)capacity * sizeof(TraceEntry)));
    HIP_TRY(e, e->trace_count.ensure(8));
    HIP_TRY(e, hipMemsetAsync(e->trace_count.p, 0, 8, e->stream));
    HIP_TRY(e, hipStreamSynchronize(e->stream));
  }
  e->trace_cap = capacity;
  return RPA_OK;
}

int rpa_trace_read(rpa_engine* e, rpa_trace_entry* out, uint64_t capacity, uint64_t* n_entries, uint64_t* n_dropped) {
  if (!e || !n_entries) return RPA_ERR_INVALID;
  std::lock_guard<std::mutex> lk(e->mu);
  *n_entries = 0;
  if (n_dropped) *n_dropped = 0;
  if (!e->trace_cap) return fail(e, RPA_ERR_INVALID, "rpa_trace_read: tracing is off (rpa_trace_begin)");
  HIP_TRY(e, hipSetDevice(e->device));
  unsigned long long attempts = 0;
  HIP_TRY(e, hipMemcpy(&attempts, e->trace_count.p, 8, hipMemcpyDeviceToHost));
  const uint64_t have = attempts < e->trace_cap ? attempts : e->trace_cap;
  if (n_dropped) *n_dropped = attempts - have;
  if (capacity == 0) {  // "how many are there?": nothing is copied (n_dropped = every alignment the call consumed)
    if (n_dropped) *n_dropped = attempts;
    return RPA_OK;
  }
  std::vector<TraceEntry> tmp((size_t)have);
  if (have) HIP_TRY(e, hipMemcpy(tmp.data(), e->trace_buf.p, (size_t)have * sizeof(TraceEntry), hipMemcpyDeviceToHost));
  std::sort(tmp.begin(), tmp.end(), [](const TraceEntry& a, const TraceEntry& b) { return a.set != b.set ? a.set < b.set : a.seq < b.seq; });
  const uint64_t n = have < capacity ? have : capacity;
  if (n && !out) return fail(e, RPA_ERR_INVALID, "rpa_trace_read: null output");
  static_assert(sizeof(TraceEntry) == sizeof(rpa_trace_entry), "trace entry layout");
  if (n) memcpy(out, tmp.data(), (size_t)n * sizeof(TraceEntry));
  *n_entries = n;
  if (n_dropped) *n_dropped += have - n;
  return RPA_OK;
}

int rpa_host_register(rpa_engine* e, void* ptr, size_t bytes) {
  if (!e) return RPA_ERR_INVALID;
  if (!ptr || !bytes) return RPA_OK;
  std::lock_guard<std::mutex> lk(e->mu);
  HIP_TRY(e, hipSetDevice(e->device));
  HIP_TRY(e, hipHostRegister(ptr, bytes, hipHostRegisterPortable));  // every device of the process may DMA from it
  return RPA_OK;
}

int rpa_host_unregister(rpa_engine* e, void* ptr) {
  if (!e) return RPA_ERR_INVALID;
  if (!ptr) return RPA_OK;
  std::lock_guard<std::mutex> lk(e->mu);
  HIP_TRY(e, hipSetDevice(e->device));
  HIP_TRY(e, hipHostUnregister(ptr));
  return RPA_OK;
}

// ---- asynchronous jobs ----
static void slot_main(rpa_engine* slot) {
  for (;;) {
    rpa_job* job = nullptr;
    {
      std::unique_lock<std::mutex> lk(slot->job_mu);
      slot->job_cv.wait(lk, [&] { return slot->job_pending || slot->job_quit; });
      if (slot->job_quit && !slot->job_pending) return;
      job = slot->job_pending;
    }
    slot->err.clear();
    const int rc = predict_on(slot, slot->owner, &job->batch, job->out, job->stats, job->batch.mem_kind);  // rows go where the batch lives
    {
      std::lock_guard<std::mutex> lk(slot->job_mu);
      job->status = rc;
      if (rc) job->err = slot->err;
      job->done = true;
      slot->job_pending = nullptr;
      slot->job_busy = false;
    }
    slot->job_cv.notify_all();
  }
}

// every job slot idle (store loads and appends change what the slots read)
static void quiesce_slots(rpa_engine* e) {
  for (rpa_engine* s : e->slots) {
    std::unique_lock<std::mutex> lk(s->job_mu);
    s->job_cv.wait(lk, [&] { return !s->job_busy; });
  }
}

int rpa_submit_batch(rpa_engine* e, const rpa_batch* batch, rpa_prediction* out, rpa_stats* stats, rpa_job** job_out) {
  if (!e || !job_out) return RPA_ERR_INVALID;
  *job_out = nullptr;
  std::lock_guard<std::mutex> lk(e->mu);
  if (!batch || !out) return fail(e, RPA_ERR_INVALID, "rpa_submit_batch: null argument");
  if (batch->mem_kind != RPA_MEM_HOST && batch->mem_kind != RPA_MEM_DEVICE) return fail(e, RPA_ERR_INVALID, "rpa_submit_batch: bad mem_kind");
  HIP_TRY(e, hipSetDevice(e->device));
  static const size_t n_slots = [] { const char* v = getenv("RPA_JOB_SLOTS"); const long n = v ? atol(v) : 0; return (size_t)(n >= 1 && n <= 8 ? n : RPA_JOBS_IN_FLIGHT); }();  // (experiments)
  // The slots take the jobs in turn, over a ring of ONE MORE slot than the caller has ever had in flight (a slot sizes a whole
  // working set -- rows, pair lists, bin lists: several GB for a 128 MB block -- at its first job, so a caller that keeps k jobs in
  // flight touches k + 1 slots, not RPA_JOBS_IN_FLIGHT of them: the CLI, two per engine, three).  Why one more and not "the lowest
  // idle slot": measured on config 4, three jobs in flight -- 4.29 ms per step when a job goes to the slot that has just finished,
  // 4.08 in turn over four (profiles/r05_ab_slot_choice.txt); the spare slot's stream and thread are ready while the finished one
  // still winds down.
  unsigned busy_now = 0;
  for (rpa_engine* s : e->slots) {
    std::lock_guard<std::mutex> sl(s->job_mu);
    busy_now += s->job_busy ? 1u : 0u;
  }
  e->peak_in_flight = std::max(e->peak_in_flight, busy_now + 1u);
  const size_t ring = std::min<size_t>(n_slots, (size_t)e->peak_in_flight + 1u);
  while (e->slots.size() < ring) {
    rpa_engine* s = make_worker(e);
    if (!s) return RPA_ERR_DEVICE;
    s->owner = e;
    s->profiling = false;
    s->job_thread = std::thread(slot_main, s);
    e->slots.push_back(s);
  }
  rpa_engine* slot = e->slots[e->next_slot++ % e->slots.size()];
  rpa_job* job = new (std::nothrow) rpa_job();
  if (!job) return fail(e, RPA_ERR_DEVICE, "out of host memory");
  job->batch = *batch; job->out = out; job->stats = stats; job->slot = slot;
  {
    std::unique_lock<std::mutex> sl(slot->job_mu);
    slot->job_cv.wait(sl, [&] { return !slot->job_busy; });  // the slot's previous job has written its rows
    slot->job_busy = true;
    slot->job_pending = job;
  }
  slot->job_cv.notify_all();
  *job_out = job;
  return RPA_OK;
}

int rpa_wait(rpa_engine* e, rpa_job* job) {
  if (!e || !job) return RPA_ERR_INVALID;
  rpa_engine* slot = job->slot;
  {
    std::unique_lock<std::mutex> sl(slot->job_mu);
    slot->job_cv.wait(sl, [&] { return job->done; });  // (no marker here: the slot thread's markers say where the job is)
  }
  const int rc = job->status;
  if (rc) { std::lock_guard<std::mutex> lk(e->mu); e->err = job->err; }
  delete job;
  return rc;
}

int rpa_align_pairs(rpa_engine* e, uint32_t n_seqs, const uint8_t* letters, const uint64_t* offsets, uint32_t n_pairs,
                    const uint32_t* pair_a, const uint32_t* pair_b, rpa_alignment* out, int mem_kind) {
  if (!e) return RPA_ERR_INVALID;
  std::lock_guard<std::mutex> lk(e->mu);
  if (!offsets || (n_pairs && (!pair_a || !pair_b || !out))) return fail(e, RPA_ERR_INVALID, "rpa_align_pairs: null argument");
  if (mem_kind != RPA_MEM_HOST && mem_kind != RPA_MEM_DEVICE) return fail(e, RPA_ERR_INVALID, "rpa_align_pairs: bad mem_kind");
  if (n_pairs == 0) return RPA_OK;
  HIP_TRY(e, hipSetDevice(e->device));
  Store& tmp = e->pair_store;  // a store of its own, rebuilt per call (its buffers are kept and reused)
  int rc = pack_store(e, &tmp, n_seqs, letters, offsets, mem_kind, &e->tmp_letters);
  if (rc) return rc;
  WorkDev w{};
  rc = alloc_work(e, 1, 1, n_seqs, n_pairs, true, &w, 1, false);
  if (rc) return rc;
  const hipMemcpyKind kind = mem_kind == RPA_MEM_HOST ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice;
  HIP_TRY(e, hipMemcpyAsync(w.pair_a, pair_a, (size_t)n_pairs * 4, kind, e->stream));
  HIP_TRY(e, hipMemcpyAsync(w.pair_b, pair_b, (size_t)n_pairs * 4, kind, e->stream));
  HIP_TRY(e, hipMemsetAsync(w.counters, 0, sizeof(uint32_t) * (kNumCounters + 1), e->stream));
  if (e->params.alphabet != RPA_DNA) { const int crc = counter_pool_reset(e); if (crc) return crc; }
  rc = profile_begin(e);
  if (rc) return rc;
  const StoreDev sv = tmp.view();
  w.store_q = sv.words; w.store_db = sv.words;
  HIP_TRY(e, launch_plan_plain_segments(e->stream, e->params.alphabet, w, sv));
  HIP_TRY(e, launch_bin_pairs(e->stream, e->params, w, n_pairs));
  HIP_TRY(e, hipMemcpyAsync(e->h_counters, w.counters, sizeof(uint32_t) * kNumCounters, hipMemcpyDeviceToHost, e->stream));
  RPA_WHERE("rpa_align_pairs: sync after bin_pairs");
  HIP_TRY(e, hipStreamSynchronize(e->stream));
  rc = check_device_error(e, e->h_counters);
  if (rc) return rc;
  uint32_t counts[kNumRoundCounters];
  memcpy(counts, e->h_counters, sizeof(counts));
  rc = run_align_bins(e, w, counts);
  if (rc) return rc;
  void* out_dev = out;
  if (mem_kind == RPA_MEM_HOST) { HIP_TRY(e, e->out_rows.ensure((size_t)n_pairs * sizeof(rpa_alignment))); out_dev = e->out_rows.p; }
  HIP_TRY(e, launch_export_alignments(e->stream, w, n_pairs, out_dev));
  if (mem_kind == RPA_MEM_HOST) HIP_TRY(e, hipMemcpyAsync(out, out_dev, (size_t)n_pairs * sizeof(rpa_alignment), hipMemcpyDeviceToHost, e->stream));
  RPA_WHERE("rpa_align_pairs: final sync");
  HIP_TRY(e, hipStreamSynchronize(e->stream));
  return profile_end(e);
}

void rpa_lca_default_params(rpa_lca_params* p) {
  if (!p) return;
  memset(p, 0, sizeof(*p));
  p->model = RPA_LCA_SIMPLE;
  p->toppercent = 0.05f;   // taxator.cpp:287
  p->minscore = 0.f;       // :290
  p->maxevalue = 1000.0;   // :288
  p->minsupport = 1;       // :289
  p->nbest = 1;            // :291
}

int rpa_lca_predict_batch(rpa_engine* e, const rpa_batch* batch, const double* evalue, const uint8_t* node_unclassified,
                          const rpa_lca_params* params, rpa_prediction* out, int out_mem_kind) {
  if (!e) return RPA_ERR_INVALID;
  std::lock_guard<std::mutex> lk(e->mu);
  if (!batch || !params || !out) return fail(e, RPA_ERR_INVALID, "rpa_lca_predict_batch: null argument");
  if (params->model < RPA_LCA_DUMMY || params->model > RPA_LCA_NBEST) return fail(e, RPA_ERR_INVALID, "rpa_lca_predict_batch: unknown model %d", params->model);
  const uint32_t S = batch->n_sets, H = batch->n_records;
  if (S == 0) return RPA_OK;
  if (!batch->set_offsets || !batch->query_len || (H && (!batch->qstart || !batch->qstop || !batch->score || !batch->node)))
    return fail(e, RPA_ERR_INVALID, "rpa_lca_predict_batch: null array in batch");
  if (out_mem_kind != RPA_MEM_HOST && out_mem_kind != RPA_MEM_DEVICE) return fail(e, RPA_ERR_INVALID, "rpa_lca_predict_batch: bad out_mem_kind");
  HIP_TRY(e, hipSetDevice(e->device));
  BatchDev b{};
  b.n_sets = S; b.n_records = H;
  const double* ev = evalue;
  const uint8_t* uncl = node_unclassified;
  if (batch->mem_kind == RPA_MEM_HOST) {
    if (batch->set_offsets[0] != 0 || batch->set_offsets[S] != H) return fail(e, RPA_ERR_INVALID, "set_offsets must run from 0 to n_records");
    for (uint32_t r = 0; r < H; ++r)
      if (batch->node[r] >= e->tax.n_nodes) return fail(e, RPA_ERR_INVALID, "record %u: taxonomy node index out of range", r);
    // staging slots shared with rpa_predict_batch: 0 set_offsets, 2 query_len, 3 qstart, 4 qstop, 10 node
    const void* src[5] = {batch->set_offsets, batch->query_len, batch->qstart, batch->qstop, batch->node};
    const int slot[5] = {0, 2, 3, 4, 10};
    const size_t cnt[5] = {(size_t)S + 1, S, H, H, H};
    for (int k = 0; k < 5; ++k) {
      HIP_TRY(e, e->in_u32[slot[k]].ensure((cnt[k] ? cnt[k] : 1) * 4));
      if (cnt[k]) HIP_TRY(e, hipMemcpyAsync(e->in_u32[slot[k]].p, src[k], cnt[k] * 4, hipMemcpyHostToDevice, e->stream));
    }
    HIP_TRY(e, e->in_f32.ensure((size_t)(H ? H : 1) * 4));
    if (H) HIP_TRY(e, hipMemcpyAsync(e->in_f32.p, batch->score, (size_t)H * 4, hipMemcpyHostToDevice, e->stream));
    if (batch->filtered) {
      HIP_TRY(e, e->in_u8.ensure(H ? H : 1));
      if (H) HIP_TRY(e, hipMemcpyAsync(e->in_u8.p, batch->filtered, H, hipMemcpyHostToDevice, e->stream));
    }
    if (evalue) {
      HIP_TRY(e, e->in_f64.ensure((size_t)(H ? H : 1) * 8));
      if (H) HIP_TRY(e, hipMemcpyAsync(e->in_f64.p, evalue, (size_t)H * 8, hipMemcpyHostToDevice, e->stream));
      ev = e->in_f64.as<const double>();
    }
    if (node_unclassified) {
      HIP_TRY(e, e->in_uncl.ensure(e->tax.n_nodes));
      HIP_TRY(e, hipMemcpyAsync(e->in_uncl.p, node_unclassified, e->tax.n_nodes, hipMemcpyHostToDevice, e->stream));
      uncl = e->in_uncl.as<const uint8_t>();
    }
    b.set_offsets = e->in_u32[0].as<uint32_t>(); b.query_len = e->in_u32[2].as<uint32_t>();
    b.qstart = e->in_u32[3].as<uint32_t>(); b.qstop = e->in_u32[4].as<uint32_t>(); b.node = e->in_u32[10].as<uint32_t>();
    b.score = e->in_f32.as<float>();
    b.filtered = batch->filtered ? e->in_u8.as<uint8_t>() : nullptr;
  } else if (batch->mem_kind == RPA_MEM_DEVICE) {
    b.set_offsets = batch->set_offsets; b.query_len = batch->query_len; b.qstart = batch->qstart; b.qstop = batch->qstop;
    b.score = batch->score; b.node = batch->node; b.filtered = batch->filtered;
  } else return fail(e, RPA_ERR_INVALID, "rpa_lca_predict_batch: bad mem_kind");
  void* rows_dev = out;
  if (out_mem_kind == RPA_MEM_HOST) { HIP_TRY(e, e->out_rows.ensure((size_t)S * sizeof(rpa_prediction))); rows_dev = e->out_rows.p; }
  {  // a bad node index or offset must not reach the kernel (host batches were checked above; this covers device batches too)
    HIP_TRY(e, e->tile_base.ensure(16));
    HIP_TRY(e, hipMemsetAsync(e->tile_base.p, 0, 8, e->stream));
    HIP_TRY(e, launch_validate_batch(e->stream, b, e->tax.n_nodes, e->tile_base.as<uint32_t>()));
    HIP_TRY(e, hipMemcpyAsync(e->h_counters, e->tile_base.p, 8, hipMemcpyDeviceToHost, e->stream));
    RPA_WHERE("rpa_lca_predict_batch: sync after validate");
    HIP_TRY(e, hipStreamSynchronize(e->stream));
    if (e->h_counters[0]) return fail(e, RPA_ERR_INVALID, "batch rejected: set_offsets must run from 0 to n_records without decreasing and every node index must lie inside the taxonomy (set / record %u)", e->h_counters[1]);
  }
  const LcaParamsDev p{params->model, params->toppercent, params->minscore, (float)params->maxevalue, params->minsupport, params->nbest,
                       params->ignore_unclassified};
  HIP_TRY(e, launch_lca_models(e->stream, b, e->tax, p, ev, uncl, rows_dev));
  if (out_mem_kind == RPA_MEM_HOST) HIP_TRY(e, hipMemcpyAsync(out, rows_dev, (size_t)S * sizeof(rpa_prediction), hipMemcpyDeviceToHost, e->stream));
  RPA_WHERE("rpa_lca_predict_batch: final sync");
  HIP_TRY(e, hipStreamSynchronize(e->stream));
  return RPA_OK;
}

int rpa_profile_enable(rpa_engine* e, int on) {
  if (!e) return RPA_ERR_INVALID;
  std::lock_guard<std::mutex> lk(e->mu);
  e->profiling = on != 0;
  e->prof_counters = on != RPA_PROFILE_TIMING;
  for (rpa_engine* w : e->workers) { w->profiling = e->profiling; w->prof_counters = e->prof_counters; }
  return RPA_OK;
}

int rpa_profile_reset(rpa_engine* e) {
  if (!e) return RPA_ERR_INVALID;
  std::lock_guard<std::mutex> lk(e->mu);
  HIP_TRY(e, hipSetDevice(e->device));
  memset(&e->prof, 0, sizeof(e->prof));
  HIP_TRY(e, hipMemset(e->prof_acc.p, 0, kProfParts * kProfSlots * sizeof(unsigned long long)));
  for (rpa_engine* w : e->workers) {
    memset(&w->prof, 0, sizeof(w->prof));
    HIP_TRY(e, hipMemset(w->prof_acc.p, 0, kProfParts * kProfSlots * sizeof(unsigned long long)));
  }
  return RPA_OK;
}

int rpa_profile_get(rpa_engine* e, rpa_profile* out) {
  if (!e || !out) return RPA_ERR_INVALID;
  std::lock_guard<std::mutex> lk(e->mu);
  *out = e->prof;
  for (const rpa_engine* w : e->workers) {  // parts of a batch that ran on worker engines
    out->launches += w->prof.launches; out->pairs += w->prof.pairs; out->residue_bytes += w->prof.residue_bytes;
    out->cells += w->prof.cells; out->align_ms += w->prof.align_ms; out->demanded += w->prof.demanded; out->cells_updated += w->prof.cells_updated;
    out->rounds = out->rounds > w->prof.rounds ? out->rounds : w->prof.rounds;
  }
  return RPA_OK;
}

}  // extern "C"
