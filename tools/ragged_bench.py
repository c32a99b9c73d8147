#!/usr/bin/env python3
"""tools/ragged_bench.py [--variants pruned256,pruned,deep] [--steps K] [--scale F] [--oracle-cells C] -- the engine on record sets
of MIXED shape (bench.py reports the result under secondary.c2_ragged).

Every BASELINE configuration is uniform: exactly 20 or 50 hits per record set, one segment length per batch.  A real alignment
stream is not (core/USAGE.md:90-96): here hits per set are drawn from 1 ... 500 with a long tail, segments of 50 ... 1500 nt sit in
one batch (several length bins and both nucleotide kernel families per round, ragged tiles), and the taxonomy has 500 k nodes
of which the batch names a few thousand.  Three variants of the same stream:
  pruned256  taxonomy pruned to the seven ranks (8 levels), the sets of more than 250 hits left out: 8-bit path labels, rows of 2 quads
  pruned     the same with the sets of up to 500 hits: one such set makes every label of the batch 16 bits wide (rows of 3 quads)
  deep       the same stream on the taxonomy kept whole (taxator -d 0): chains of "no rank" nodes inside the lineages, ~25 levels,
             rows of 6 quads -- the row shapes whose k_decide / k_emit variants spill registers
Reported per variant: demanded realignments/s with jobs in flight, a synchronous step, alignment-kernel ms and launches per
step, computed per demanded alignment, rounds, row shape, and a window of record sets checked against the CPU oracle.
Test infrastructure when --oracle-cells > 0 (uses oracle/); bench.py calls run() the same way."""
import argparse
import collections
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

# (record sets, hits per set, segment length): a long tail in the hits (most sets have a handful, a few have hundreds), the
# lengths spread over 50 ... 1500 nt; `scale` multiplies the number of sets
CLASSES = [(40000, 1, 150), (30000, 2, 100), (30000, 3, 200), (24000, 5, 50), (24000, 5, 300), (20000, 8, 150), (16000, 12, 200),
           (12000, 16, 100), (12000, 20, 200), (8000, 20, 500), (6000, 32, 150), (4000, 32, 800), (4000, 48, 200), (2500, 64, 300),
           (1500, 100, 200), (600, 100, 1500), (800, 150, 100), (500, 250, 200), (200, 350, 150), (100, 500, 200), (60, 500, 400)]


def build(synth, scale, device):
    classes = [(max(2, int(n * scale)), h, L) for (n, h, L) in CLASSES]
    t0 = time.perf_counter()
    ds = synth.make_ragged_dataset(classes, seed=5, device=device)
    return ds, classes, time.perf_counter() - t0


def subset(ds, keep):
    """the record sets keep[s] of a ragged dataset (stores untouched)"""
    so = ds.batch["set_offsets"].astype(np.int64)
    n = np.diff(so)
    rec_keep = np.repeat(keep, n)
    b = {}
    for k, v in ds.batch.items():
        if k == "set_offsets":
            o = np.zeros(int(keep.sum()) + 1, np.int64)
            o[1:] = np.cumsum(n[keep])
            b[k] = o.astype(np.uint32)
        elif len(v) == len(keep):
            b[k] = v[keep]
        else:
            b[k] = v[rec_keep]
    return b


def run_variant(pkg, synth, torch, name, tax, batch, ds, steps, in_flight, oracle_cells, node_map=None):
    if node_map is not None:
        batch = dict(batch, node=node_map[batch["node"]].astype(np.uint32))
    S, H = len(batch["set_offsets"]) - 1, len(batch["qstart"])
    model = pkg.RPAPredictionModel(tax, alphabet=pkg.DNA)
    model.load_store(pkg.STORE_QUERY, ds.query_letters, ds.query_offsets)
    model.load_store(pkg.STORE_DB, ds.ref_letters, ds.ref_offsets)
    dev = {k: torch.from_numpy(np.ascontiguousarray(v).view(np.int32) if v.dtype == np.uint32 else np.ascontiguousarray(v)).cuda() for k, v in batch.items()}
    outs = [(torch.zeros((S, 8), dtype=torch.int32, device="cuda"), torch.zeros((S, 4), dtype=torch.int32, device="cuda")) for _ in range(in_flight + 1)]
    torch.cuda.synchronize()
    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        model.set_stream(stream.cuda_stream)
        model.predict(dev, mem_kind=pkg.MEM_DEVICE, out=outs[0][0], stats=outs[0][1])
        jobs = collections.deque()

        def go(n):
            for k in range(n):
                r_, s_ = outs[k % len(outs)]
                jobs.append(model.submit(dev, out=r_, stats=s_, mem_kind=pkg.MEM_DEVICE))
                if len(jobs) == in_flight:
                    jobs.popleft().wait()
            while jobs:
                jobs.popleft().wait()

        go(in_flight + 1)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        go(steps)
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        model.profile(enable=False)
        torch.cuda.synchronize()
        sync_steps = int(os.environ.get("RAGGED_SYNC_STEPS", steps))  # (tools/kt_ragged_inflight.sh: a trace that is nearly all steps in flight)
        t1 = time.perf_counter()
        for _ in range(sync_steps):
            model.predict(dev, mem_kind=pkg.MEM_DEVICE, out=outs[0][0], stats=outs[0][1])
        torch.cuda.synchronize()
        sync_ms = (time.perf_counter() - t1) / sync_steps * 1e3
        model.profile(enable=True, reset=True)
        for _ in range(sync_steps):
            model.predict(dev, mem_kind=pkg.MEM_DEVICE, out=outs[0][0], stats=outs[0][1])
        torch.cuda.synchronize()
        prof = model.profile()
        prof = dict(prof, **{k: prof[k] * steps / sync_steps for k in ("pairs", "demanded", "rounds", "launches", "align_ms", "total_ms", "cells")})
    rows = outs[0][0].cpu().numpy().view(pkg.PRED_DTYPE).reshape(-1)
    stats = outs[0][1].cpu().numpy().view(pkg.STATS_DTYPE).reshape(-1)
    demanded = int(stats["pass0"].astype(np.int64).sum() + stats["pass1"].astype(np.int64).sum() + stats["pass2"].astype(np.int64).sum())
    so = batch["set_offsets"].astype(np.int64)
    nrec = np.diff(so)
    depth_levels = int(np.asarray(tax.depth).max())
    lb = 8 if nrec.max() <= 256 else (16 if nrec.max() <= 65536 else 32)
    words = ((depth_levels + 7) // 8 * 8 * lb + 31) // 32
    quads = next(q for q in (2, 3, 4, 6, 8) if 4 * q - 6 >= words)
    res = {"variant": name, "record_sets": S, "records": H, "largest_set": int(nrec.max()), "taxonomy_nodes": len(tax.parent), "taxonomy_levels": depth_levels,
           "label_bits": lb, "row_quads": quads,
           "value": demanded * steps / elapsed, "unit": "realignments/s", "steps": steps, "in_flight": in_flight, "ms_per_step": elapsed / steps * 1e3,
           "sync_unprofiled_ms_per_step": sync_ms, "demanded_per_step": demanded, "record_sets_per_s": S * steps / elapsed,
           "computed_over_demanded": prof["pairs"] / max(prof["demanded"], 1), "rounds_per_step": prof["rounds"] / steps,
           "alignment_launches_per_step": prof["launches"] / steps, "align_ms_per_step": prof["align_ms"] / steps,
           "device_ms_per_step": prof["total_ms"] / steps, "placement_and_host_ms_per_step": (prof["total_ms"] - prof["align_ms"]) / steps,
           "cell_updates_per_s_in_kernel": prof["cells"] / max(prof["align_ms"] / 1e3, 1e-12)}
    if oracle_cells > 0:
        from helpers import oracle as O
        # the first record sets of the (shuffled) stream, as many as fit the budget of alignment cells (hits x segment length^2)
        qlen = batch["query_len"].astype(np.float64)
        cost = np.cumsum(nrec * 1.5 * qlen * qlen)
        W = int(max(1, min(S, np.searchsorted(cost, oracle_cells))))
        otax = O.TaxArrays(tax.parent, tax.left, tax.right, tax.depth)
        q = O.StoreArrays(synth.encode(ds.query_letters, 0), ds.query_offsets)
        db = O.StoreArrays(synth.encode(ds.ref_letters, 0), ds.ref_offsets)
        from concurrent.futures import ThreadPoolExecutor
        workers = max(1, min(16, len(os.sched_getaffinity(0))))
        chunk = max(1, W // (4 * workers))

        def check(lo):
            hi = min(W, lo + chunk)
            r0, r1 = int(so[lo]), int(so[hi])
            sub = {k: (v[r0:r1] if len(v) == H else v) for k, v in batch.items()}
            rc, exp, est = O.predict_batch(otax, q, db, 0, (so[lo:hi + 1] - so[lo]).astype(np.uint32), batch["query_seq"][lo:hi], batch["query_len"][lo:hi],
                                           synth.records_aos(sub, O.RECORD_DTYPE))
            return rc == 0 and rows[lo:hi].tobytes() == exp.tobytes() and stats[lo:hi].tobytes() == est.tobytes()

        t0 = time.perf_counter()
        with ThreadPoolExecutor(workers) as ex:
            oks = list(ex.map(check, range(0, W, chunk)))
        res["oracle_window"] = {"record_sets": W, "largest_set_in_window": int(nrec[:W].max()), "identical": bool(all(oks)), "seconds": time.perf_counter() - t0}
    model.close()
    del dev, outs
    torch.cuda.empty_cache()
    return res


def run(pkg, synth, torch, variants=("pruned256", "pruned", "deep"), steps=5, in_flight=4, scale=1.0, oracle_cells=2.5e10):
    ds, classes, t_gen = build(synth, scale, "cuda")
    so = ds.batch["set_offsets"].astype(np.int64)
    nrec = np.diff(so)
    out = {"workload": "record sets of mixed shape: %d sets, %d alignment records; hits per set %d ... %d (median %d, mean %.1f), segments of %d ... %d nt in one batch; "
                       "reference store %.0f Mbases in %d sequences; taxonomy of 500 k nodes (the mock tree's lineages inside random clades)"
                       % (ds.n_sets, len(ds.batch["qstart"]), int(nrec.min()), int(nrec.max()), int(np.median(nrec)), float(nrec.mean()),
                          min(c[2] for c in classes), max(c[2] for c in classes), len(ds.ref_letters) / 1e6, len(ds.ref_offsets) - 1),
           "datagen_s": t_gen, "variants": {}}
    big = synth.embed_in_big_tree(ds.tax, 500_000, 7)
    for v in variants:
        try:
            if v == "pruned64":  # (not in the default list: every set inside k_decide's register-resident path, the yardstick for the larger ones)
                out["variants"][v] = run_variant(pkg, synth, torch, v, big, subset(ds, nrec <= 64), ds, steps, in_flight, oracle_cells)
            elif v == "pruned256":
                out["variants"][v] = run_variant(pkg, synth, torch, v, big, subset(ds, nrec <= 256), ds, steps, in_flight, oracle_cells)
            elif v == "pruned":
                out["variants"][v] = run_variant(pkg, synth, torch, v, big, ds.batch, ds, steps, in_flight, oracle_cells)
            else:
                st = synth.StretchedTaxonomy(ds.tax, lambda i: (i * 7 + 3) % 5)   # 0-4 "no rank" nodes in the edge above every node: ~25 levels
                deep = synth.embed_in_big_tree(st, 500_000, int(st.depth.max()))
                out["variants"][v] = run_variant(pkg, synth, torch, v, deep, ds.batch, ds, steps, in_flight, oracle_cells, node_map=st.new_index)
        except Exception as e:  # (one variant failing must not take the others along)
            out["variants"][v] = {"error": repr(e)}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--variants", default="pruned256,pruned,deep")
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--in-flight", type=int, default=4)
    ap.add_argument("--scale", type=float, default=1.0)
    ap.add_argument("--oracle-cells", type=float, default=2.5e10)
    args = ap.parse_args()
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
    import torch
    import __graft_entry__ as graft
    pkg = graft.load_package()
    synth = importlib.import_module("taxator_tk_amd.synth")
    print(json.dumps(run(pkg, synth, torch, tuple(args.variants.split(",")), args.steps, args.in_flight, args.scale, args.oracle_cells)))


if __name__ == "__main__":
    main()
