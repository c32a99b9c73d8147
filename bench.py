#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X-native RPA engine (BASELINE.json: query-segment realignments/sec).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A "step" is one pass of the hot path (rpa_predict_batch: rank -> plan -> {emit, align, decide}* ) over one
synthetic batch that is already resident in HBM: BASELINE config 2, 1 M record sets x 20 hits x ~200 bp per GPU
(weak scaling: every rank gets its own 1 M sets, seed 2 + rank).  The unit counted is the reference's own: one
pairwise alignment demanded by RPAPredictionModel::predict's control flow (its gcounter, STATS column 7), NOT the
(larger) number of alignments the device computes speculatively.  With N > 1 each step ends with the one exchange the
path has: an RCCL gather of the 32-byte PredictionRecord rows to rank 0.

Prints ONE JSON line (rank 0).  `roofline` is for the dominant kernel (batched Myers edit distance): algorithmic
bytes (len_a + len_b + 8 per computed alignment) over that kernel's device time measured with HIP events
on the engine's stream inside the timed region.  `cpu_baseline` times the CPU oracle (oracle/, kind "port") on a bounded
sample of the same workload on this box's host cores; `cpu_reference` (extra) times the reference's own vendored SeqAn
pair call (oracle/_ref, kind "reference") on the first hits of the same batch.
"""
import argparse
import importlib
import json
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")  # before HIP comes up (torch): a hardware queue per engine stream (rpa_engine.hip: make_stream)

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def usable_cpus():
    """Host threads this process may really use: the scheduler affinity mask, capped by the cgroup CPU quota (a GPU box
    hands a 1-GPU job a slice of the host: os.cpu_count() there reports every hardware thread of the machine)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period) + 0.5)))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p_ = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, int(q / p_ + 0.5)))
        except Exception:
            pass
    return max(1, n)


def cpu_baseline(synth, ds, seconds_target=12.0):
    """Oracle (plain C restatement of the reference path) on all host cores, on a prefix of the same record sets."""
    from concurrent.futures import ThreadPoolExecutor
    from helpers import oracle as O
    cores = usable_cpus()
    tax = O.TaxArrays(ds.tax.parent, ds.tax.left, ds.tax.right, ds.tax.depth)
    q = O.StoreArrays(synth.encode(ds.query_letters, ds.alphabet), ds.query_offsets)
    db = O.StoreArrays(synth.encode(ds.ref_letters, ds.alphabet), ds.ref_offsets)
    b = ds.batch
    n_hits = ds.n_hits

    def run(lo, hi):
        offs = (b["set_offsets"][lo:hi + 1] - b["set_offsets"][lo]).astype(np.uint32)
        r0, r1 = int(b["set_offsets"][lo]), int(b["set_offsets"][hi])
        sub = {k: (v[r0:r1] if len(v) == len(b["qstart"]) else v) for k, v in b.items()}
        recs = synth.records_aos(sub, O.RECORD_DTYPE)
        rc, out, st = O.predict_batch(tax, q, db, ds.alphabet, offs, b["query_seq"][lo:hi], b["query_len"][lo:hi], recs)
        assert rc == 0
        return int(st["pass0"].sum() + st["pass1"].sum() + st["pass2"].sum())

    # calibrate on a small slice (about 1.6 G cells on one core: 2000 sets of config 2, ONE set of config 5), then size
    # the sample for ~seconds_target of wall time on all cores
    seg = float(np.mean(b["query_len"])) if len(b["query_len"]) else 1.0
    n_probe = int(max(1, min(2000, ds.n_sets, 1.6e9 / max(1.0, n_hits * seg * seg))))
    t0 = time.perf_counter()
    probe = run(0, n_probe)
    dt = max(time.perf_counter() - t0, 1e-3)
    per_core = probe / dt
    # every thread takes the next block of record sets until the time is up (one core's rate does not predict 256 busy
    # threads -- the 10 kbp oracle runs 25x slower per thread then -- so the sample is bounded by the clock, not by a
    # forecast); a block is ~0.25 s of one idle core
    import itertools
    import threading
    block = int(max(1, min(n_probe, n_probe * 0.25 / dt)))
    ticket = itertools.count()
    lock = threading.Lock()
    deadline = time.perf_counter() + seconds_target
    taken = [0]

    def worker(_):
        done = 0
        while time.perf_counter() < deadline:
            with lock:
                k = next(ticket)
            lo = k * block
            if lo >= ds.n_sets:
                break
            hi = min(ds.n_sets, lo + block)
            done += run(lo, hi)
            with lock:
                taken[0] = max(taken[0], hi)
        return done

    t0 = time.perf_counter()
    with ThreadPoolExecutor(cores) as ex:
        total = sum(ex.map(worker, range(cores)))
    dt = time.perf_counter() - t0
    return {"value": total / dt, "unit": "realignments/s", "cores": cores, "kind": "port", "single_thread_value": per_core,
            "sample": "first %d record sets x %d hits of the same batch in blocks of %d sets (%d realignments, %.1f s wall, oracle C restatement, "
                      "one thread per usable CPU (%d of the %d the machine reports), every thread takes blocks until %.0f s have passed; "
                      "single_thread_value = the first %d sets on one otherwise idle thread)"
                      % (taken[0], n_hits, block, total, dt, cores, os.cpu_count() or 1, seconds_target, n_probe)}


def cpu_reference_pairs(synth, ds, seconds_target=4.0, n_pairs=20000):
    """The reference's own pair-level call (vendored SeqAn compiled in place, oracle/_ref/libseqan_ref.so) on all host
    cores: getAlignmentDNA / getAlignmentProtein of (query range, hit's reference range) for the first hits of the same
    batch -- the call that takes ~all of the reference's time.  Extra information next to cpu_baseline;
    None when the prebuilt checker is absent."""
    import ctypes as C
    from concurrent.futures import ThreadPoolExecutor
    path = os.path.join(ROOT, "oracle", "_ref", "libseqan_ref.so")
    if not os.path.exists(path):
        return None
    R = C.CDLL(path)
    b = ds.batch
    seg = float(np.mean(b["query_len"])) if len(b["query_len"]) else 1.0
    n_pairs = int(min(n_pairs, len(b["qstart"]), max(usable_cpus(), 4e10 / max(1.0, seg * seg))))  # bounded in cells too (10 kbp: one pair per thread)
    set_of = np.searchsorted(b["set_offsets"], np.arange(n_pairs), side="right") - 1
    q = synth.encode(ds.query_letters, ds.alphabet)
    db = synth.encode(ds.ref_letters, ds.alphabet)
    comp = np.array([3, 2, 1, 0, 4], np.uint8)
    seqs = []
    for r in range(n_pairs):
        qo = int(ds.query_offsets[b["query_seq"][set_of[r]]])
        seqs.append(q[qo + int(b["qstart"][r]) - 1: qo + int(b["qstop"][r])])
        ro = int(ds.ref_offsets[b["ref_seq"][r]])
        lo, hi = int(b["rstart"][r]), int(b["rstop"][r])
        seg = db[ro + min(lo, hi) - 1: ro + max(lo, hi)]
        seqs.append(comp[seg[::-1]] if (lo > hi and ds.alphabet == 0) else seg)
    letters = np.ascontiguousarray(np.concatenate(seqs))
    offsets = np.concatenate([[0], np.cumsum([len(x) for x in seqs])]).astype(np.uint64)
    cores = usable_cpus()
    fn = R.ref_time_dna_pairs if ds.alphabet == 0 else R.ref_time_protein_pairs
    fn.restype = C.c_double
    pa = np.arange(n_pairs, dtype=np.uint32) * 2
    pb = pa + 1
    chk = C.c_longlong(0)

    def timed(n, threads, reps):  # the C side builds the SeqAn strings first, then times `reps` passes on `threads` std::threads
        return float(fn(C.c_void_p(letters.ctypes.data), C.c_void_p(offsets.ctypes.data), C.c_void_p(pa.ctypes.data), C.c_void_p(pb.ctypes.data),
                        C.c_int(n), C.c_int(threads), C.c_int(reps), C.byref(chk)))

    # one thread alone (a slice of the sample, ~1 s), then every usable CPU (~seconds_target)
    n1 = max(1, min(n_pairs, n_pairs // cores if cores > 1 else n_pairs))
    dt1 = max(timed(n1, 1, 1), 1e-4)
    reps1 = max(1, int(1.0 / dt1))
    dt1 = timed(n1, 1, reps1) if reps1 > 1 else dt1
    single = n1 * reps1 / dt1
    dta = max(timed(n_pairs, cores, 1), 1e-4)
    reps = max(1, int(seconds_target / dta))
    dt = timed(n_pairs, cores, reps) if reps > 1 else dta
    done = n_pairs * reps
    return {"value": done / dt, "unit": "pair alignments/s", "cores": cores, "kind": "reference", "single_thread_value": single,
            "sample": "(query range, reference range) of the first %d hits of the same batch, %d passes (%d alignments, %.1f s wall), "
                      "the reference's vendored SeqAn 2.4.0 through getAlignment%s's calls on pre-built strings, std::threads inside the "
                      "checker library, one per usable CPU (%d of the %d the machine reports); single_thread_value: %d pairs x %d passes on one thread"
                      % (n_pairs, reps, done, dt, "DNA" if ds.alphabet == 0 else "Protein", cores, os.cpu_count() or 1, n1, reps1)}


# BASELINE.json configs, per-GPU sizes (weak scaling: the whole job is sets x N).  c2: 1 M sets on one GPU (the headline);
# c3: 10 M sets over 8 GPUs = 1.25 M per GPU, one seed per shard (3 + rank: SURVEY 8d "seeds 3..10 per shard");
# c5: 100 k contigs over 8 GPUs = 12.5 k per GPU -- with N > 1 ONE workload of 12.5 k x N contigs is generated (same
# seed on every rank) and cut into N contiguous shards of equal summed alignment cells (sharding.shard_bounds_by_cost).
CONFIGS = {
    "c2": dict(sets=1_000_000, hits=20, seg_len=200, alphabet=0, genome_len=4000, level_rate=0.03, query_rate=0.05, seed=2,
               name="BASELINE config 2", kernel="k_myers_short_band4 / k_myers_short (batched bit-parallel edit distance, one alignment per lane, a four-word window of the pattern around the diagonal: exact Ukkonen band with retries)"),
    "c3": dict(sets=1_250_000, hits=20, seg_len=200, alphabet=0, genome_len=4000, level_rate=0.03, query_rate=0.05, seed=3,
               name="BASELINE config 3 (10 M record sets over 8 GPUs: 1.25 M per GPU)", kernel="k_myers_short_band4 / k_myers_short (batched bit-parallel edit distance, one alignment per lane, a four-word window of the pattern around the diagonal: exact Ukkonen band with retries)"),
    "c4": dict(sets=500_000, hits=20, seg_len=67, alphabet=1, genome_len=1500, level_rate=0.06, query_rate=0.08, seed=2,
               name="BASELINE config 4 (protein, 67 aa)", kernel="k_nw_protein_lanes / k_nw_protein_packed (BLOSUM62 global DP, score + tie order + path counters in one 32-bit cell; the 24-row strips of a pair in adjacent lanes, strip boundaries over DPP)"),
    "c5": dict(sets=12_500, hits=50, seg_len=10_000, alphabet=0, genome_len=40_000, level_rate=0.03, query_rate=0.05, seed=2, cut_by_cost=True,
               name="BASELINE config 5 (10 kbp contigs, per-GPU share of 100 k)", kernel="k_myers_band (bit-parallel edit distance inside Ukkonen's band, one alignment per lane, pattern in register strips of 5-8 words)"),
}


def committed_profile(tcfg):
    """HBM traffic and vector instructions per alignment launch of configuration `tcfg` from the committed rocprofv3 PMC passes
    of this same command (profiles/<round>_traffic.json = config 2, profiles/<round>_cN_traffic.json = config N; they cannot be
    read from inside the process).  Returns (traffic bytes per launch, its file, wave-instructions per launch, its file)."""
    import glob
    import re
    tagged = lambda f, c: ("_%s_" % c) in os.path.basename(f)
    cands = sorted(f for f in glob.glob(os.path.join(ROOT, "profiles", "*traffic.json"))
                   if (tagged(f, tcfg) if tcfg != "c2" else not any(tagged(f, c) for c in CONFIGS)))
    if not cands:
        return None, None, None, None
    tj = json.load(open(cands[-1]))
    traffic, traffic_src = tj["traffic_bytes_per_launch"], os.path.relpath(cands[-1], ROOT)
    valu_per_launch, valu_src = None, None
    sq = cands[-1].replace("_traffic.json", "_pmc_sq.txt")  # the SQ pass of the same profile
    if os.path.exists(sq):
        cur, instr, disp = None, 0.0, 0
        for line in open(sq):
            m = re.search(r"(rpa::.+?)\s+dispatches=(\d+)", line)
            if m:
                cur = m.group(1) if ("k_myers" in m.group(1) or "k_nw_protein" in m.group(1)) else None
                disp += int(m.group(2)) if cur else 0
            m = re.search(r"SQ_INSTS_VALU\s+sum=(\S+)", line)
            if m and cur:
                instr += float(m.group(1))
        if disp:
            valu_per_launch, valu_src = instr / disp, os.path.relpath(sq, ROOT)
    return traffic, traffic_src, valu_per_launch, valu_src


def valu_issue(valu_per_launch, valu_src, align_ms, n_launch):
    """wave-instructions per launch (SQ_INSTS_VALU of the committed profile) over the live launch duration, per SIMD (1024 of
    them); a wave64 instruction takes a SIMD-32 two cycles (MI355X_MICROARCH.md) = 1.2 per ns at 2.4 GHz."""
    if valu_per_launch is None or align_ms <= 0:
        return None
    a = valu_per_launch / (align_ms / n_launch * 1e6) / 1024.0
    return {"achieved": a, "peak": 1.2, "unit": "wave-instr/ns/SIMD", "frac": a / 1.2, "wave_instr_per_launch": valu_per_launch, "source": valu_src}


def measure_e2e(synth, ds):
    """End to end through the product's own driver: the headline's record sets written in the reference's file formats
    (alignments TSV ~1 GB, query / reference FASTA, mapping, taxonomy dump), then `taxator < aln.tsv > out.gff3` as a child
    process -- load, parse, GPU, GFF3 -- twice (page cache warm the second time), and once more with small blocks and small
    GPU calls, whose GFF3 must have the same SHA-1.  Outside every timed region; reported under secondary.e2e."""
    import hashlib
    import shutil
    import subprocess
    import tempfile
    exe = os.path.join(ROOT, "taxator-tk_amd", "bin", "taxator")
    if not os.path.exists(exe):
        return {"error": "taxator-tk_amd/bin/taxator is not built"}
    folder = tempfile.mkdtemp(prefix="rpa_e2e_")
    try:
        t0 = time.perf_counter()
        synth.write_files(ds, folder)
        t_write = time.perf_counter() - t0
        aln = os.path.join(folder, "aln.tsv")
        env = dict(os.environ, TAXATORTK_TAXONOMY_NCBI=os.path.join(folder, "tax"))
        cmd = [exe, "-g", folder + "/mapping.tax", "-q", folder + "/query.fna", "-f", folder + "/ref.fna", "--timing"]

        def run(extra, out_name):
            t1 = time.perf_counter()
            with open(aln) as fin, open(os.path.join(folder, out_name), "wb") as fout:
                r = subprocess.run(cmd + extra, stdin=fin, stdout=fout, stderr=subprocess.PIPE, text=True, env=env, timeout=600)
            wall = time.perf_counter() - t1
            timing = {}
            for line in r.stderr.splitlines():
                if line.startswith("TIMING\t"):
                    for tok in line.split("\t")[1:]:
                        k_, _, v_ = tok.partition("=")
                        try:
                            timing[k_] = float(v_)
                        except ValueError:
                            pass
            h = hashlib.sha1()
            with open(os.path.join(folder, out_name), "rb") as f:
                for chunk in iter(lambda: f.read(1 << 24), b""):
                    h.update(chunk)
            return r.returncode, wall, timing, h.hexdigest()

        runs = [run([], "out.gff3") for _ in range(2)]
        small = run(["--block-mb", "8", "--batch", "50000"], "out_small.gff3")
        rc, wall, tm, sha = min(runs, key=lambda x: x[1])
        n_sets = int(ds.n_sets)
        # which stage bounds the pipeline, from the stage times the run reported: parse_s = the busiest reader team's seconds, predict_s =
        # seconds the GPU threads (two per engine, each submits a block and waits for it) spent in their calls, write_s = the writer's
        wall_p = tm.get("wall_s") or 0.0
        busy = {"reader": (tm.get("parse_s") or 0.0) / wall_p, "gpu": (tm.get("predict_s") or 0.0) / 2.0 / max(1.0, tm.get("gpus") or 1.0) / wall_p,
                "writer": (tm.get("write_s") or 0.0) / wall_p} if wall_p > 0 else None
        bound = None
        if busy:
            top = max(busy, key=busy.get)
            bound = "%s (busy %.0f %% of the pipeline's wall time; reader %.0f %%, GPU threads %.0f %%, writer %.0f %%)" % (
                top, 100 * busy[top], 100 * busy["reader"], 100 * busy["gpu"], 100 * busy["writer"])
        return {"workload": "taxator < aln.tsv > out.gff3: %d record sets, %d alignment lines, %.2f GB of TSV, in-memory FASTA stores" % (
                    n_sets, len(ds.batch["qstart"]), os.path.getsize(aln) / 1e9),
                "ok": rc == 0 and small[0] == 0 and all(r_[0] == 0 for r_ in runs),
                "wall_s": wall, "record_sets_per_s": n_sets / wall,                                    # the whole child process: start-up, loads, pipeline
                "pipeline_s": tm.get("wall_s"), "pipeline_record_sets_per_s": (n_sets / tm["wall_s"]) if tm.get("wall_s") else None,
                "stages_s": {"load": tm.get("load_s"), "parse": tm.get("parse_s"), "parse_all_teams": tm.get("parse_sum_s"), "predict": tm.get("predict_s"),
                             "write": tm.get("write_s"), "reader_teams": tm.get("reader_teams")},
                "stage_busy_over_pipeline_wall": busy, "bound": bound,
                "gff3_sha1": sha, "gff3_sha1_small_blocks": small[3], "same_output_with_small_blocks": sha == small[3],
                "files_written_s": t_write}
    except Exception as e:  # the headline stands whatever happens here
        return {"error": repr(e)}
    finally:
        shutil.rmtree(folder, ignore_errors=True)


def measure_hbm_store(replicas, headline_ms, headline_sync_ms):
    """The headline once more against a reference store that is really spread over HBM (config 2's own 8 MB collection sits in
    L2 / Infinity Cache): `--store-replicas R` lays it out R times and points every alignment row at a random replica.  A child
    process of its own (the 64 GB of letters it packs and the engine's job slots are gone when it ends); after the headline."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--store-replicas", str(replicas), "--steps", "6", "--warmup", "2", "--no-cpu-baseline",
           "--no-secondary"]
    try:
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=420, cwd=ROOT)
        lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
        if r.returncode != 0 or not lines:
            return {"error": "rc %d: %s" % (r.returncode, r.stderr[-400:])}
        o = json.loads(lines[-1])
        return {"workload": o["config"]["workload"] + " -- reference store laid out %d times (%.1f GB packed), every row at a random replica" % (
                    replicas, o["reference_store"]["packed_bytes"] / 1e9),
                "value": o["value"], "unit": o["unit"], "ms_per_step": o["ms_per_step"], "sync_unprofiled_ms_per_step": o["sync_unprofiled_ms_per_step"],
                "align_ms_per_step": o["align_ms_per_step"], "hbm_in_use_gb": o["hbm_in_use_gb"], "store_packed_bytes": o["reference_store"]["packed_bytes"],
                "roofline_frac": o["roofline"]["frac"],
                "over_cache_resident_store": {"ms_per_step": o["ms_per_step"] / headline_ms if headline_ms else None,
                                              "sync_unprofiled_ms_per_step": (o["sync_unprofiled_ms_per_step"] / headline_sync_ms) if headline_sync_ms else None}}
    except Exception as e:  # the headline stands whatever happens here
        return {"error": repr(e)}


def measure_secondary(pkg, synth, torch, cfg, sets, hits, seg_len, steps, in_flight, device, profile_tag=None):
    """One of the other BASELINE configurations on this GPU, after the headline has been measured: `steps` steps through
    rpa_submit_batch with `in_flight` of them running (the same timed region as the headline), then `steps` synchronous
    calls for the per-launch profile.  No CPU leg.  Returns a small dict for the line's `secondary` object."""
    import collections
    t_gen = time.perf_counter()
    ds = synth.make_dataset(n_sets=sets, n_hits=hits, seg_len=seg_len, branching=4, genome_len=cfg["genome_len"], seed=cfg["seed"],
                            device="cuda", alphabet=cfg["alphabet"], level_rate=cfg["level_rate"], query_rate=cfg["query_rate"])
    t_gen = time.perf_counter() - t_gen
    model = pkg.RPAPredictionModel(ds.tax, alphabet=cfg["alphabet"], device=device)
    model.load_store(pkg.STORE_QUERY, ds.query_letters, ds.query_offsets)
    model.load_store(pkg.STORE_DB, ds.ref_letters, ds.ref_offsets)
    dev = {k: torch.from_numpy(np.ascontiguousarray(v).view(np.int32) if v.dtype == np.uint32 else np.ascontiguousarray(v)).cuda()
           for k, v in ds.batch.items()}
    outs = [(torch.zeros((ds.n_sets, 8), dtype=torch.int32, device="cuda"), torch.zeros((ds.n_sets, 4), dtype=torch.int32, device="cuda"))
            for _ in range(in_flight + 1)]
    torch.cuda.synchronize()
    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        model.set_stream(stream.cuda_stream)
        model.predict(dev, mem_kind=pkg.MEM_DEVICE, out=outs[0][0], stats=outs[0][1])  # allocation pass: the engine's own working set
        jobs = collections.deque()

        def run(n):
            for k in range(n):
                r_, s_ = outs[k % len(outs)]
                jobs.append(model.submit(dev, out=r_, stats=s_, mem_kind=pkg.MEM_DEVICE))
                if len(jobs) == in_flight:
                    jobs.popleft().wait()
            while jobs:
                jobs.popleft().wait()

        run(in_flight + 1)  # every job slot sizes its working set, + one warm step
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run(steps)
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        model.profile(enable=True, reset=True)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(steps):
            model.predict(dev, mem_kind=pkg.MEM_DEVICE, out=outs[0][0], stats=outs[0][1])
        torch.cuda.synchronize()
        sync_ms = (time.perf_counter() - t1) / steps * 1e3
        prof = model.profile()
        model.profile(enable=pkg.PROFILE_TIMING, reset=True)  # the launch durations without the kernels' counters (see the headline's leg)
        torch.cuda.synchronize()
        for _ in range(steps):
            model.predict(dev, mem_kind=pkg.MEM_DEVICE, out=outs[0][0], stats=outs[0][1])
        torch.cuda.synchronize()
        prof_t = model.profile()
        if prof_t["launches"] == prof["launches"]:
            prof = dict(prof, align_ms=prof_t["align_ms"], total_ms=prof_t["total_ms"])
    demanded = int(outs[0][1][:, 1:].to(torch.int64).sum().item())
    align_s = max(prof["align_ms"] / 1e3, 1e-12)
    traffic, traffic_src, valu_pl, valu_src = committed_profile(profile_tag) if profile_tag else (None, None, None, None)
    n_launch = max(prof["launches"], 1)
    res = {"workload": "%s: %d record sets x %d hits x %d %s" % (cfg["name"], sets, hits, seg_len, "bp" if cfg["alphabet"] == 0 else "aa"),
           "value": demanded * steps / elapsed, "unit": "realignments/s", "steps": steps, "in_flight": in_flight,
           "ms_per_step": elapsed / steps * 1e3, "sync_ms_per_step": sync_ms, "demanded_per_step": demanded,
           "computed_over_demanded": prof["pairs"] / max(prof["demanded"], 1), "rounds_per_step": prof["rounds"] / steps,
           "align_ms_per_step": prof["align_ms"] / steps,
           "roofline": {"bound": "hbm", "achieved": prof["residue_bytes"] / align_s / 1e9, "peak": 8000.0, "unit": "GB/s",
                        "frac": prof["residue_bytes"] / align_s / 1e9 / 8000.0, "kernel": cfg["kernel"],
                        "cell_updates_per_s_in_kernel": prof["cells"] / align_s, "launches": prof["launches"],
                        "band_cell_updates_per_s_in_kernel": prof["cells_updated"] / align_s,
                        "band_cells_over_cell_equivalents": prof["cells_updated"] / max(prof["cells"], 1),
                        "bytes_per_launch": prof["residue_bytes"] / n_launch, "avg_launch_ms": prof["align_ms"] / n_launch,
                        "traffic": traffic, "traffic_unit": "bytes per launch (2 x FETCH_SIZE + WRITE_SIZE, rocprofv3 PMC)", "traffic_source": traffic_src,
                        "valu_issue": valu_issue(valu_pl, valu_src, prof["align_ms"], n_launch)},
           "datagen_s": t_gen}
    model.close()
    del dev, outs, ds
    torch.cuda.empty_cache()
    return res


def launch_ranks(n, argv):
    """`python bench.py --gpus N` with no launcher around it (WORLD_SIZE unset): start the N ranks ourselves, the way the
    reference's parallel mode is ONE command (`taxator -p N`, core/taxator.cpp:181-210).  Decided before this process has
    imported torch or touched the GPU in any way: the ranks are fresh children of `python -m torch.distributed.run` (never
    an exec of a process that initialised HIP), this process only relays their output and the launcher's exit code.
    RPA_BENCH_RANK_SCRIPT replaces the script the ranks run (tests/test_distributed_cpu.py: a stub that needs no GPU)."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    script = os.environ.get("RPA_BENCH_RANK_SCRIPT") or os.path.abspath(__file__)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: what RCCL needs between the ranks' processes on this driver
    env.setdefault("OMP_NUM_THREADS", "1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), script] + list(argv)
    sys.stderr.write("bench.py: --gpus %d without a launcher: starting %d ranks (%s)\n" % (n, n, " ".join(cmd)))
    sys.stderr.flush()
    p = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True, bufsize=1)
    try:
        for line in p.stdout:   # rank 0's JSON line (and anything else the ranks print) as it comes
            sys.stdout.write(line)
            sys.stdout.flush()
        rc = p.wait()
    except BaseException:
        p.terminate()
        try:
            p.wait(timeout=20)
        except subprocess.TimeoutExpired:
            p.kill()
        raise
    return rc if rc >= 0 else 128 - rc


def main():
    pre = argparse.ArgumentParser(add_help=False)
    pre.add_argument("--gpus", type=int, default=1)
    n_gpus = pre.parse_known_args()[0].gpus
    if n_gpus > 1 and "WORLD_SIZE" not in os.environ and "RANK" not in os.environ:
        raise SystemExit(launch_ranks(n_gpus, sys.argv[1:]))  # (before torch is imported: this process never sees the GPU)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", choices=sorted(CONFIGS), default="c2",
                    help="c2 = the headline configuration (BASELINE config 2); c3 = config 3's per-GPU share (for --gpus 8), c4 protein, "
                         "c5 long contigs are extra measurements")
    ap.add_argument("--in-flight", type=int, default=None, help="jobs submitted before the oldest is waited for (1 .. RPA_JOBS_IN_FLIGHT = 4); default: 4 for the nucleotide configurations (profiles/r04_inflight_sweep.txt), 3 for protein (its lanes kernel holds a 91 KB table per CU: a fourth job costs it 8 %)")
    ap.add_argument("--pipeline", action="store_true",
                    help="(the default, kept for old command lines) the steps go through rpa_submit_batch with --in-flight of them "
                         "running at a time on the engine's job slots: the placement rounds of one step under the alignment kernels of "
                         "another")
    ap.add_argument("--sync", action="store_true",
                    help="one synchronous rpa_predict_batch per step instead (round 1's timed region).  The default run times the "
                         "steps through the asynchronous boundary and then repeats them this way, untimed for `value`, for the "
                         "per-launch durations behind `roofline` (overlapping jobs inflate those)")
    ap.add_argument("--store-replicas", type=int, default=1,
                    help="lay the reference store out R times (R x 4096 sequences; every alignment row points at a random replica of its "
                         "species): the same work against a store that is really spread over HBM instead of sitting in L2 / "
                         "Infinity Cache -- 2000 replicas of config 2's 16 Mbase collection = 32 GB packed.  An extra measurement, "
                         "never the headline")
    ap.add_argument("--dist-backend", choices=["nccl", "gloo"], default="nccl",
                    help="nccl (= RCCL over xGMI, the real thing) or gloo: rows are gathered through host memory, which lets several "
                         "ranks share ONE GPU -- a rehearsal of the N > 1 code (shards, cost cut, uneven gathers) on a 1-GPU box")
    ap.add_argument("--one-device", action="store_true", help="every rank uses cuda:0 (with --dist-backend gloo)")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise torch.distributed (nccl = RCCL) and run the gather path even with one rank (hardware rehearsal of the N > 1 code)")
    ap.add_argument("--sets", type=int, default=None, help="record sets per GPU (default: the configuration's)")
    ap.add_argument("--hits", type=int, default=None)
    ap.add_argument("--seg-len", type=int, default=None)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true",
                    help="the default command (one GPU, config 2 at its full size, CPU legs on) also runs configs 4 and 5 and a 200 aa "
                         "protein workload for 3 steps each AFTER the headline has been measured and reports them under `secondary`; "
                         "this switches that off (--no-cpu-baseline does too)")
    ap.add_argument("--host-buffers", action="store_true",
                    help="hand the batch over in host memory and take the rows back to the host every step (the PCIe-inclusive "
                         "rate noted in DESIGN.md; never the headline value, whose batch is resident in HBM)")
    args = ap.parse_args()
    cfg = CONFIGS[args.config]
    args.sets = args.sets or cfg["sets"]
    args.hits = args.hits or cfg["hits"]
    args.seg_len = args.seg_len or cfg["seg_len"]

    import torch
    import torch.distributed as dist
    import __graft_entry__ as graft

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d does not match WORLD_SIZE %d (launch with torch.distributed.run for N > 1)" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the RPA engine has no CPU path")
    if args.one_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    use_dist = world > 1 or args.force_dist
    gloo = args.dist_backend == "gloo"
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if gloo:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    pkg = graft.load_package()
    synth = importlib.import_module("taxator_tk_amd.synth")
    t_gen = time.perf_counter()
    sharding = importlib.import_module("taxator_tk_amd.sharding")
    cut_by_cost = bool(cfg.get("cut_by_cost")) and world > 1
    if cut_by_cost:
        # ONE workload for the whole job (every rank generates the same one), cut where the summed alignment cells are equal
        ds = synth.make_dataset(n_sets=args.sets * world, n_hits=args.hits, seg_len=args.seg_len, branching=4, genome_len=cfg["genome_len"],
                                seed=cfg["seed"], device="cuda", alphabet=cfg["alphabet"], level_rate=cfg["level_rate"],
                                query_rate=cfg["query_rate"])
        bounds = sharding.shard_bounds_by_cost(sharding.set_costs(ds.batch), world)
        ds.batch = sharding.slice_batch(ds.batch, bounds[rank], bounds[rank + 1])  # query_seq keeps pointing into the whole query store
        ds.n_sets = bounds[rank + 1] - bounds[rank]
        gather_sizes = [bounds[r + 1] - bounds[r] for r in range(world)]
    else:
        ds = synth.make_dataset(n_sets=args.sets, n_hits=args.hits, seg_len=args.seg_len, branching=4, genome_len=cfg["genome_len"],
                                seed=cfg["seed"] + rank, device="cuda", alphabet=cfg["alphabet"], level_rate=cfg["level_rate"],
                                query_rate=cfg["query_rate"])
        gather_sizes = [ds.n_sets] * world
    t_gen = time.perf_counter() - t_gen

    model = pkg.RPAPredictionModel(ds.tax, alphabet=cfg["alphabet"], device=local_rank)
    model.load_store(pkg.STORE_QUERY, ds.query_letters, ds.query_offsets)
    store_bytes = int(ds.ref_offsets[-1]) // (2 if cfg["alphabet"] == 0 else 1)
    if args.store_replicas > 1:
        # R copies of the collection back to back, built on the device (letters R x 16 MB, packed by the engine, then freed);
        # replica k of species s is sequence s + k * n_species
        R_, n_sp, total = args.store_replicas, len(ds.ref_offsets) - 1, int(ds.ref_offsets[-1])
        pad = (-total) % 8                                            # keep every replica 8-byte aligned in the letter buffer
        one = torch.from_numpy(np.concatenate([ds.ref_letters, np.full(pad, ord("A"), np.uint8)])).cuda()
        letters = one.repeat(R_)
        offs = (torch.from_numpy(ds.ref_offsets[:-1].astype(np.int64)).cuda()[None, :] + (torch.arange(R_, device="cuda", dtype=torch.int64) * (total + pad))[:, None]).reshape(-1)
        # a replica's last sequence ends where its padding starts: lengths come from consecutive offsets, so give every replica's
        # pad its own (never referenced) slot of length `pad` by ending the table per replica
        ends = offs.reshape(R_, n_sp)[:, -1] + int(ds.ref_offsets[-1] - ds.ref_offsets[-2])
        table = torch.cat([torch.cat([offs.reshape(R_, n_sp), ends[:, None]], dim=1).reshape(-1), torch.tensor([R_ * (total + pad)], device="cuda", dtype=torch.int64)])
        # sequences per replica: n_sp real ones + 1 pad slot
        model.load_store(pkg.STORE_DB, letters, table, mem_kind=pkg.MEM_DEVICE)
        del letters, one
        torch.cuda.empty_cache()
        g = np.random.default_rng(12345)
        ds.batch["ref_seq"] = (ds.batch["ref_seq"].astype(np.int64) + g.integers(0, R_, len(ds.batch["ref_seq"])) * (n_sp + 1)).astype(np.uint32)
        store_bytes *= R_
    else:
        model.load_store(pkg.STORE_DB, ds.ref_letters, ds.ref_offsets)
    dev = {k: torch.from_numpy(np.ascontiguousarray(v).view(np.int32) if v.dtype == np.uint32 else np.ascontiguousarray(v)).cuda()
           for k, v in ds.batch.items()}
    rows = torch.zeros((ds.n_sets, 8), dtype=torch.int32, device="cuda")
    stats = torch.zeros((ds.n_sets, 4), dtype=torch.int32, device="cuda")
    # Everything from here on runs on a NON-BLOCKING side stream, never on the legacy null stream: a null-stream operation
    # -- e.g. the event a collective records on the current stream before it starts -- is a barrier across every blocking
    # stream of the process (round 2's CU-masked engine streams were blocking ones; since round 3 they are plain
    # non-blocking streams, RPA_CU_MASK_QUEUES=1 brings the old kind back).
    torch.cuda.synchronize()
    stream = torch.cuda.current_stream() if os.environ.get("RPA_BENCH_NULL_STREAM") else torch.cuda.Stream()  # (A/B)
    torch.cuda.set_stream(stream)
    model.set_stream(stream.cuda_stream)

    # X1: PredictionRecord rows of every rank to rank 0 over xGMI (RCCL gather); shard sizes are equal and known, the
    # receive buffers on rank 0 are allocated once
    cdev = "cpu" if gloo else "cuda"  # where the collectives' tensors live
    gather_out = ([torch.empty((max(gather_sizes), 8), dtype=torch.int32, device=cdev) for _ in range(world)]
                  if (use_dist and rank == 0) else None)
    # RCCL: the gather of step k runs beside the kernels of step k + 1 (rows double-buffered; every step's rows still reach
    # rank 0 inside the timed region).  gloo rehearsals keep the plain blocking gather.
    pipelined = not args.sync and not args.host_buffers
    if args.in_flight is None:
        args.in_flight = 3 if cfg["alphabet"] == 1 else 4
    args.in_flight = max(1, min(int(os.environ.get("RPA_JOB_SLOTS", "4")), args.in_flight))
    overlap_gather = use_dist and not gloo and not args.host_buffers and not pipelined
    rows_buf = [rows, torch.zeros_like(rows)] if overlap_gather else [rows]
    stats_buf = [stats, torch.zeros_like(stats)] if overlap_gather else [stats]
    gather_buf = [gather_out, ([torch.empty_like(t) for t in gather_out] if gather_out is not None else None)] if overlap_gather else [gather_out]
    pending_gather = [None, None]

    import collections
    in_flight = collections.deque()
    host_out = []
    if args.host_buffers:
        # the caller's arrays are pinned once (a driver that reuses its batch buffers does the same): DMA at the PCIe rate
        def pinned_copy(a):  # page-locked memory from the runtime's own allocator (torch pin_memory = hipHostMalloc)
            a = np.ascontiguousarray(a)
            t_ = torch.empty(a.nbytes, dtype=torch.uint8, pin_memory=True)
            out_ = t_.numpy().view(a.dtype).reshape(a.shape)
            out_[...] = a
            pinned_keep.append(t_)
            return out_
        pinned_keep = []
        if os.environ.get("RPA_BENCH_REGISTER"):  # A/B: register the numpy arrays instead (rpa_host_register)
            host_batch = {k: np.ascontiguousarray(v) for k, v in ds.batch.items()}
            for v in host_batch.values():
                model.pin(v)
        else:
            host_batch = {k: pinned_copy(v) for k, v in ds.batch.items()}
        for _ in range(4):  # rows / counters of the jobs in flight (+ one being read)
            r_, s_ = pinned_copy(np.zeros(ds.n_sets, pkg.PRED_DTYPE)), pinned_copy(np.zeros(ds.n_sets, pkg.STATS_DTYPE))
            host_out.append((r_, s_))

    # pipelined: a buffer set per job in flight + two whose rows may still be on their way to rank 0
    n_pipe = args.in_flight + 2
    pipe_out = [(rows, stats)] + [(torch.zeros_like(rows), torch.zeros_like(stats)) for _ in range(n_pipe - 1)] if pipelined else []
    ring = sharding.RowGatherRing(gather_sizes, n_pipe, cdev) if (pipelined and use_dist) else None

    def finish_one():
        if pipelined:
            k, job = in_flight.popleft()
            job.wait()  # the job's rows are complete in pipe_out[k]
            if use_dist:
                # the exchange runs beside the jobs still in flight; a buffer set is not reused before its gather is done
                # (RCCL: device rows over xGMI; gloo rehearsal: the same calls on a host copy of the rows)
                ring.post(k, pipe_out[k][0].cpu() if gloo else pipe_out[k][0])
                step.last = k
            return
        r_host, s_host = in_flight.popleft().wait()
        if not step.copied:
            rows.copy_(torch.from_numpy(r_host.view(np.int32).reshape(-1, 8)))
            stats.copy_(torch.from_numpy(s_host.view(np.int32).reshape(-1, 4)))
            step.copied = True

    def step():
        if args.host_buffers:
            # the asynchronous boundary (rpa_submit_batch / rpa_wait): H2D of the batch columns and D2H of rows + counters
            # inside every job, two jobs in flight -- the copies of one under the kernels of the other
            r_, s_ = host_out[step.n % 4]
            step.n += 1
            in_flight.append(model.submit(host_batch, out=r_, stats=s_))
            if len(in_flight) == args.in_flight:
                finish_one()
        elif pipelined:
            k = step.n % n_pipe
            step.n += 1
            if use_dist and ring.pending[k] is not None:    # the exchange that read this buffer set n_pipe steps ago: done long
                ring.reuse(k)                               # since, but the job about to overwrite the rows runs on a stream of
                torch.cuda.current_stream().synchronize()   # its own, so the host makes sure
            in_flight.append((k, model.submit(dev, out=pipe_out[k][0], stats=pipe_out[k][1], mem_kind=pkg.MEM_DEVICE)))
            if len(in_flight) == args.in_flight:
                finish_one()
        elif overlap_gather:
            k = step.n % 2
            step.n += 1
            if pending_gather[k] is not None:   # the exchange that read this buffer two steps ago (stream-level wait, the host goes on)
                pending_gather[k][0].wait()
                pending_gather[k] = None
            model.predict(dev, mem_kind=pkg.MEM_DEVICE, out=rows_buf[k], stats=stats_buf[k])
            pending_gather[k] = sharding.gather_rows(rows_buf[k], dst=0, sizes=gather_sizes, out=gather_buf[k], force=True, async_op=True)
            step.last = k
            return
        else:
            model.predict(dev, mem_kind=pkg.MEM_DEVICE, out=rows, stats=stats)
        if use_dist:
            if gloo:
                torch.cuda.current_stream().synchronize()
            step.gathered = sharding.gather_rows(rows.cpu() if gloo else rows, dst=0, sizes=gather_sizes, out=gather_out, force=True)

    step.copied = False
    step.n = 0
    step.gathered = None
    step.last = 0
    # Allocation pass, before the W warm-up steps: every job slot of the engine (and the engine itself, for the synchronous
    # profile leg) sizes its working set at its first batch -- a few GB of hipMalloc each.  With fewer warm-up steps than
    # slots that would otherwise happen inside the timed region.
    if pipelined:
        model.predict(dev, mem_kind=pkg.MEM_DEVICE, out=rows, stats=stats)
    if pipelined or args.host_buffers:
        for _ in range(args.in_flight):
            step()
        while in_flight:
            finish_one()
    for _ in range(args.warmup):
        step()
    while in_flight:
        finish_one()
    for k_ in range(len(pending_gather)):
        if pending_gather[k_] is not None:
            pending_gather[k_][0].wait()
            pending_gather[k_] = None
    if ring is not None:
        ring.drain()
    # The profile in the timed region: job slots (steps in flight, host buffers) never profile; synchronous steps (--sync,
    # multi-rank runs with a blocking gather) time their launches WITHOUT the kernels' counters here and count in a short pass
    # behind the timed region (RPA_PROFILE_TIMING: the counters are reductions inside the alignment kernels, ~4 % of K1).
    sync_steps = not pipelined and not args.host_buffers
    model.profile(enable=pkg.PROFILE_TIMING if sync_steps else True, reset=True)
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    while in_flight:
        finish_one()
    t_jobs = time.perf_counter() - t0    # this rank's own K steps are done (their rows computed)
    for pg in pending_gather:
        if pg is not None:
            pg[0].wait()
    if ring is not None:
        ring.drain()
    torch.cuda.synchronize()
    t_local = time.perf_counter() - t0   # ... and its part of every exchange
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    prof = model.profile()
    prof_steps = args.steps
    count_steps = args.steps  # steps the counters (pairs, bytes, cells, demanded) of `prof` cover
    sync_ms = None
    sync_plain_ms = None
    sync_timing_ms = None
    prof_t = None
    if sync_steps:
        prof_t = prof
        count_steps = max(1, min(args.steps, 2))
        model.profile(enable=True, reset=True)
        for _ in range(count_steps):
            model.predict(dev, mem_kind=pkg.MEM_DEVICE, out=rows, stats=stats)
        torch.cuda.synchronize()
        prof = model.profile()
    if pipelined:
        # The same steps once more as synchronous calls, for the profile only (launch durations by HIP events on the
        # engine's stream, cells, demanded / computed alignments, rounds): jobs that overlap inflate each other's launches.
        prof_steps = max(1, min(args.steps, 10))
        count_steps = prof_steps
        # (first without the profile: its event pairs around every alignment launch cost a synchronous step ~0.1 ms; this is
        # the call a synchronous caller makes -- then with it, for the roofline)
        model.profile(enable=False)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(prof_steps):
            model.predict(dev, mem_kind=pkg.MEM_DEVICE, out=rows, stats=stats)
        torch.cuda.synchronize()
        sync_plain_ms = (time.perf_counter() - t1) / prof_steps * 1e3
        model.profile(enable=True, reset=True)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(prof_steps):
            model.predict(dev, mem_kind=pkg.MEM_DEVICE, out=rows, stats=stats)
        torch.cuda.synchronize()
        sync_ms = (time.perf_counter() - t1) / prof_steps * 1e3
        prof = model.profile()
        # ... and once more with the launch timing alone: the counters are reductions inside the alignment kernels (a shuffle
        # reduction of four 64-bit sums per wavefront: ~4 % of K1's time), the launch durations `roofline` is quoted on should
        # not contain the instrument
        model.profile(enable=pkg.PROFILE_TIMING, reset=True)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(prof_steps):
            model.predict(dev, mem_kind=pkg.MEM_DEVICE, out=rows, stats=stats)
        torch.cuda.synchronize()
        sync_timing_ms = (time.perf_counter() - t1) / prof_steps * 1e3
        prof_t = model.profile()

    # What each rank saw, for the first multi-GPU run to be diagnosable from its one JSON line: per rank the time its own K
    # steps took, the time until its share of the exchanges was done, and the cost of ONE gather of a step's rows on an
    # otherwise idle job (a blocking gather between barriers, after the timed region; 5 repetitions).
    per_rank, gather_ms = None, None
    if use_dist:
        mine = torch.tensor([t_jobs / args.steps * 1e3, t_local / args.steps * 1e3, elapsed / args.steps * 1e3], dtype=torch.float64, device=cdev)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        per_rank = [[round(float(x), 4) for x in r_.tolist()] for r_ in allr]
        g_rows = rows.cpu() if gloo else rows
        g_out = [torch.empty((max(gather_sizes), 8), dtype=torch.int32, device=cdev) for _ in range(world)] if rank == 0 else None
        if not gloo:
            torch.cuda.synchronize()
        dist.barrier()
        tg = time.perf_counter()
        for _ in range(5):
            sharding.gather_rows(g_rows, dst=0, sizes=gather_sizes, out=g_out, force=True)
        if not gloo:
            torch.cuda.synchronize()
        dist.barrier()
        gather_ms = (time.perf_counter() - tg) / 5 * 1e3
        del g_out

    demanded = int(stats[:, 1:].to(torch.int64).sum().item())  # per step, this rank
    t = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
    d = torch.tensor([demanded], dtype=torch.int64, device=cdev)
    n_sets_all = torch.tensor([ds.n_sets], dtype=torch.int64, device=cdev)
    if use_dist:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(d, op=dist.ReduceOp.SUM)
        dist.all_reduce(n_sets_all, op=dist.ReduceOp.SUM)
    elapsed = float(t.item())
    demanded_all = int(d.item())

    if rank == 0:
        # HBM traffic of the alignment kernels cannot be read from inside the process: it comes from the committed
        # rocprofv3 PMC passes of this same command (profiles/*traffic.json, see profiles/README.md)
        traffic, traffic_src = None, None
        valu_per_launch, valu_src = None, None
        if use_dist and rank == 0:  # the gathered rows of this rank's own shard must be the rows it computed
            if pipelined:
                gather_out, rows = ring.recv[step.last], pipe_out[step.last][0]
                step.gathered = ring.result(step.last)
                assert ring.posted == args.in_flight + args.warmup + args.steps, "a step's rows were not sent to rank 0"
            if overlap_gather:
                gather_out, rows = gather_buf[step.last], rows_buf[step.last]
                step.gathered = gather_out if len(set(gather_sizes)) == 1 else torch.cat([b_[:n_] for b_, n_ in zip(gather_out, gather_sizes)], dim=0)
            own = gather_out[0][: gather_sizes[0]] if isinstance(gather_out, list) else None
            assert own is None or torch.equal(own, rows.cpu() if gloo else rows), "gathered rows differ from the computed ones"
            g = step.gathered
            n_gathered = sum(x.shape[0] for x in g) if isinstance(g, list) else g.shape[0]
            if isinstance(g, list) and len(set(gather_sizes)) == 1:
                n_gathered = sum(gather_sizes)
            assert n_gathered == sum(gather_sizes), "the gather did not return every rank's rows"
        if args.sets == cfg["sets"] and args.hits == cfg["hits"] and args.seg_len == cfg["seg_len"]:
            tcfg = "c2" if args.config == "c3" else args.config  # config 3 runs config 2's kernels on config 2's shapes
            traffic, traffic_src, valu_per_launch, valu_src = committed_profile(tcfg)
        value = demanded_all * args.steps / elapsed
        # launch timing without the counters (prof_t: prof_steps steps) beside the counting pass (prof: count_steps steps of the same
        # batch, i.e. the same launches per step); everything below is per step
        same = prof_t is not None and prof_t["launches"] * count_steps == prof["launches"] * prof_steps
        tprof = prof_t if same else prof
        t_steps = prof_steps if same else count_steps
        n_launch = max(tprof["launches"], 1)                       # launches of the timing pass
        align_s = tprof["align_ms"] / 1e3 * count_steps / t_steps  # ... its alignment time, scaled to the steps the counters cover
        achieved = prof["residue_bytes"] / max(align_s, 1e-12) / 1e9
        out = {
            "metric": "query-segment realignments/sec",
            "value": value, "unit": "realignments/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u32" if cfg["alphabet"] == 0 else "i32", "data": "synthetic",
            "config": {"workload": "%s: %d record sets x %d hits x %d %s RPA per GPU, mock NCBI taxonomy (b=4, 5462 nodes), batch resident in HBM"
                                   % (cfg["name"], args.sets, args.hits, args.seg_len, "bp nucleotide" if cfg["alphabet"] == 0 else "aa protein")
                                   + (" -- HOST BUFFERS: batch copied to the device and rows back inside every step, several steps in flight through rpa_submit_batch / rpa_wait (PCIe-inclusive, not the headline)" if args.host_buffers else "")
                                   + (" -- one synchronous rpa_predict_batch per step" if args.sync else ""),
                       "record_sets_per_gpu": args.sets, "hits_per_set": args.hits, "segment_bp": args.seg_len,
                       "parallelism": ("record sets sharded over %d rank(s)%s; %s gather of 32-byte rows to rank 0 every step"
                                       % (world, ", ONE workload cut by summed alignment cells" if cut_by_cost else "",
                                          "gloo (host memory; REHEARSAL, ranks may share a GPU)" if gloo else "RCCL")) if use_dist else "single GPU",
                       "shard_sets": gather_sizes,
                       "unit_counted": "alignments demanded by the reference control flow (STATS gcounter), not speculative ones"},
            "record_sets_per_s": int(n_sets_all.item()) * args.steps / elapsed,
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": 8000.0, "unit": "GB/s", "frac": achieved / 8000.0,
                         # SURVEY 8(d) defines achieved = realignments/s x bytes on DEMANDED alignments: `frac` above counts every
                         # alignment the kernels computed (speculative ones included) over the kernels' own time; frac_demanded
                         # the demanded ones only over the same time; frac_step the demanded ones over the whole driver-timed step
                         # (placement kernels, host round trips and all), per GPU
                         "frac_demanded": achieved / 8000.0 * prof["demanded"] / max(prof["pairs"], 1),
                         "frac_step": (value / world) * (prof["residue_bytes"] / max(prof["pairs"], 1)) / 8e12,
                         "bytes_per_alignment": prof["residue_bytes"] / max(prof["pairs"], 1),
                         "traffic": traffic, "traffic_unit": "bytes per launch (2 x FETCH_SIZE + WRITE_SIZE, rocprofv3 PMC)",
                         "traffic_source": traffic_src,
                         "kernel": cfg["kernel"],
                         "bytes_per_launch": prof["residue_bytes"] / max(prof["launches"], 1),
                         "avg_launch_ms": tprof["align_ms"] / n_launch, "launches": tprof["launches"],
                         "timing": "HIP events around every alignment launch of %d synchronous calls with the kernels' counters OFF (rpa_profile_enable(RPA_PROFILE_TIMING)); bytes / pairs / cells from the same calls with the counters on (avg launch %.4f ms there)" % (t_steps, prof["align_ms"] / max(prof["launches"], 1)) if tprof is not prof else "HIP events around every alignment launch, counters on",
                         "computed_alignments_per_s_in_kernel": prof["pairs"] / max(align_s, 1e-12),
                         "cell_updates_per_s_in_kernel": prof["cells"] / max(align_s, 1e-12),
                         "cell_updates_note": "cell-EQUIVALENTS: len_a x len_b per computed alignment, whatever part of the matrix the kernel swept",
                         # cells the kernels really updated (words x 32 rows x columns inside the band / the strips' windows)
                         "band_cell_updates_per_s_in_kernel": prof["cells_updated"] / max(align_s, 1e-12),
                         "band_cells_over_cell_equivalents": prof["cells_updated"] / max(prof["cells"], 1),
                         "note": "integer-VALU bound bit-parallel DP: HBM fraction is reported per contract; cell updates/s and valu_issue are the honest bounds",
                         # vector-instruction issue: wave-instructions per launch (rocprofv3 SQ_INSTS_VALU of the committed profile)
                         # over the live launch duration, per SIMD (1024 of them); a wave64 instruction takes a SIMD-32 two cycles
                         # (MI355X_MICROARCH.md) = 1.2 per ns at 2.4 GHz; plain and/or/add measured ~1.0, shifts and carry ops ~0.57
                         "valu_issue": valu_issue(valu_per_launch, valu_src, tprof["align_ms"], n_launch)},
            "device_ms_per_step": tprof["total_ms"] / t_steps, "align_ms_per_step": tprof["align_ms"] / t_steps,
            "align_ms_with_counters_per_step": prof["align_ms"] / count_steps,
            "computed_over_demanded": prof["pairs"] / max(prof["demanded"], 1), "rounds_per_step": prof["rounds"] / count_steps,
            "submission": ("rpa_submit_batch / rpa_wait, %d steps in flight on the engine's job slots (the timed region); `roofline`, device_ms_per_step, "
                           "align_ms_per_step and rounds_per_step come from %d synchronous rpa_predict_batch calls on the same batch after it"
                           % (args.in_flight, prof_steps)) if pipelined else "one synchronous call per step",
            "sync_ms_per_step": sync_ms,   # one rpa_predict_batch per step WITH the profile on (the leg `roofline` comes from)
            "sync_unprofiled_ms_per_step": sync_plain_ms,  # ... the same call as a caller makes it (no event pairs around the launches)
            "sync_timing_only_ms_per_step": sync_timing_ms,  # ... with the event pairs but without the kernels' counters
            "datagen_s": t_gen,
            "hbm_in_use_gb": (lambda fr_to: round((fr_to[1] - fr_to[0]) / 1e9, 2))(torch.cuda.mem_get_info()),  # this rank's device, everything resident
            "reference_store": {"replicas": args.store_replicas, "packed_bytes": store_bytes,
                                "note": "config 2's 4096-species collection is 8 MB packed (cache resident); --store-replicas spreads the same work over HBM"},
        }
        if use_dist:  # what the collective layer saw: an 8-GPU line must show eight ranks on RCCL
            out["dist"] = {"backend": dist.get_backend(), "world_size": dist.get_world_size(),
                           "rccl_version": (".".join(str(x) for x in torch.cuda.nccl.version()) if not gloo else None),
                           "gather": "asynchronous, ring of %d receive-buffer sets (sharding.RowGatherRing)" % n_pipe if ring is not None else "blocking",
                           # per rank, ms per step: its own steps done | its share of the exchanges done | released by the closing barrier
                           "per_rank_ms": [r_[0] for r_ in per_rank], "per_rank_exchange_done_ms": [r_[1] for r_ in per_rank],
                           "per_rank_barrier_ms": [r_[2] for r_ in per_rank],
                           "gather_ms": gather_ms, "gather_bytes": int(sum(gather_sizes)) * 32,
                           "gather_note": "one blocking gather of a step's rows to rank 0 between barriers, nothing else running (5 repetitions); in the timed region the gathers run beside the steps that follow"}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(synth, ds)
            ref = cpu_reference_pairs(synth, ds)
            if ref is not None:
                out["cpu_reference"] = ref
        default_sizes = args.sets == cfg["sets"] and args.hits == cfg["hits"] and args.seg_len == cfg["seg_len"]
        if (world == 1 and args.config == "c2" and default_sizes and pipelined and not use_dist and not args.no_cpu_baseline
                and not args.no_secondary and args.store_replicas == 1):
            # the other single-GPU configurations of BASELINE.json, outside the headline's timed region and after it
            model.set_stream(None)
            model.close()
            del dev, rows, stats, pipe_out
            torch.cuda.empty_cache()
            sec = {}
            c4, c5 = CONFIGS["c4"], CONFIGS["c5"]
            for key, c_, sets_, hits_, len_, tag in (("c4", c4, c4["sets"], c4["hits"], c4["seg_len"], "c4"),
                                                     ("c4_200aa", c4, c4["sets"], c4["hits"], 200, None),
                                                     ("c4_450aa", c4, 60_000, c4["hits"], 450, None),  # beyond the packed cell (312 x 312): the score-and-length cell
                                                     ("c5", c5, c5["sets"], c5["hits"], c5["seg_len"], "c5")):
                try:
                    sec[key] = measure_secondary(pkg, synth, torch, c_, sets_, hits_, len_, 3, 3 if c_["alphabet"] == 1 else args.in_flight, local_rank, profile_tag=tag)
                except Exception as e:  # the headline stands whatever happens here
                    sec[key] = {"error": repr(e)}
            sec["e2e"] = measure_e2e(synth, ds)
            del ds
            try:  # N1 against an indexed reference that is not the toy collection (small variant of tools/cli_bigstore.py: 2 Gbases, 250 k record sets)
                import types
                sys.path.insert(0, os.path.join(ROOT, "tools"))
                import cli_bigstore
                r_ = cli_bigstore.measure(types.SimpleNamespace(gbases=2.0, sets=250_000, hits=20, dir=None, runs="default,turnover:0.5,inmemory", keep=False, json=None), quiet=True)
                sec["e2e_indexed"] = {"workload": "taxator -i ref.fna.fai -v query.fna.fai < aln.tsv: a %.1f GB reference FASTA in %d sequences of 10 kbp ... 10 Mbp, %d record sets x %d hits at random loci (tools/bigstore_gen.cpp); the full-size run (32 Gbases, 10 M record sets) is profiles/r05_cli_bigstore.txt"
                                                  % (r_["files"]["ref.fna"] / 1e9, r_["files"]["sequences"], r_["files"]["record_sets"], r_["files"]["hits"]),
                                      "runs": {k: {kk: v[kk] for kk in ("args", "rc", "wall_s", "record_sets_per_s", "pipeline_s", "pipeline_record_sets_per_s", "load_s", "stages_s", "resident",
                                                                        "same_gff3_as_first_run")} for k, v in r_["runs"].items()}}
            except Exception as e:
                sec["e2e_indexed"] = {"error": repr(e)}
            try:  # record sets of mixed shape (1 ... 500 hits, 50 ... 1500 nt, 500 k-node taxonomy pruned and whole): tools/ragged_bench.py
                sys.path.insert(0, os.path.join(ROOT, "tools"))
                import ragged_bench
                sec["c2_ragged"] = ragged_bench.run(pkg, synth, torch, steps=8, in_flight=args.in_flight)  # (eight steps: the first ones of a job slot are where the length classes settle on their kernels)
            except Exception as e:
                sec["c2_ragged"] = {"error": repr(e)}
            try:  # the pair rate across the kernel families' length classes (tools/pairs_bench.py): no cliff between them
                import pairs_bench
                rates = pairs_bench.sweep(pkg, synth, "dna", [200, 256, 260, 320, 384, 390, 512, 768, 1024, 1536], mut=0.12, calls=2)
                sec["dna_pair_rates"] = {"workload": "launches of up to 400 k random nucleotide pairs of one length (~12 % edits) through rpa_align_pairs, in-kernel time: one-strip kernels up to 384 nt, the ring-window kernel (K1r) up to 2048 nt, register strips beyond",
                                         "rates": [{"length": r["length"], "pairs": r["pairs"], "alignments_per_s": r["alignments_per_s"], "cell_equivalents_per_s": r["cell_equivalents_per_s"]} for r in rates]}
            except Exception as e:
                sec["dna_pair_rates"] = {"error": repr(e)}
            sec["c2_hbm_store"] = measure_hbm_store(4000, out["ms_per_step"], out["sync_unprofiled_ms_per_step"])
            out["secondary"] = sec
        print(json.dumps(out))
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
