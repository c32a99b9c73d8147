#!/usr/bin/env python3
"""tools/dna_divergence.py -- the placement path on config 2's shape (20 hits; 200 nt x 200 k record sets and 500 nt x 40 k) over the
divergence of the synthetic references (level_rate: substitutions per taxonomy level; BASELINE's configurations use 0.03 = ~30 % edits
between distant references): steps in flight, demanded alignments per set, alignment ms, computed per demanded.  Measurement tool."""
import sys, os, importlib, time, collections
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import __graft_entry__ as graft
pkg = graft.load_package(); synth = importlib.import_module("taxator_tk_amd.synth")
import torch
for seg in (200, 500):
  for lr in (0.003, 0.01, 0.03, 0.06):
    ds = synth.make_dataset(n_sets=200000 if seg == 200 else 40000, n_hits=20, seg_len=seg, branching=4, genome_len=4000, seed=1, level_rate=lr, query_rate=lr * 1.5)
    model = pkg.RPAPredictionModel(ds.tax, alphabet=pkg.DNA)
    model.load_store(pkg.STORE_QUERY, ds.query_letters, ds.query_offsets); model.load_store(pkg.STORE_DB, ds.ref_letters, ds.ref_offsets)
    dev = {k: torch.from_numpy(np.ascontiguousarray(v).view(np.int32) if v.dtype == np.uint32 else np.ascontiguousarray(v)).cuda() for k, v in ds.batch.items()}
    S = ds.n_sets
    outs = [(torch.zeros((S, 8), dtype=torch.int32, device="cuda"), torch.zeros((S, 4), dtype=torch.int32, device="cuda")) for _ in range(5)]
    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        model.set_stream(stream.cuda_stream)
        model.predict(dev, mem_kind=pkg.MEM_DEVICE, out=outs[0][0], stats=outs[0][1]); torch.cuda.synchronize()
        jobs = collections.deque()
        def go(n):
            for k in range(n):
                r_, s_ = outs[k % 5]; jobs.append(model.submit(dev, out=r_, stats=s_, mem_kind=pkg.MEM_DEVICE))
                if len(jobs) == 4: jobs.popleft().wait()
            while jobs: jobs.popleft().wait()
        go(5); torch.cuda.synchronize(); t1 = time.perf_counter(); go(8); torch.cuda.synchronize(); infl = (time.perf_counter() - t1) / 8 * 1e3
        model.profile(enable=True, reset=True); model.predict(dev, mem_kind=pkg.MEM_DEVICE, out=outs[0][0], stats=outs[0][1]); torch.cuda.synchronize(); prof = model.profile()
    st = outs[0][1].cpu().numpy().view(pkg.STATS_DTYPE).reshape(-1)
    dem = int(st["pass0"].astype(np.int64).sum() + st["pass1"].astype(np.int64).sum() + st["pass2"].astype(np.int64).sum())
    print("seg %4d level_rate %.3f: %6.2f ms/step in flight, demanded %8d (%.1f per set), %7.1f M/s, align %5.2f ms, rounds %d, computed/demanded %.3f, %6.1f T cell-eq/s" % (
        seg, lr, infl, dem, dem / S, dem / infl / 1e3, prof["align_ms"], prof["rounds"], prof["pairs"] / max(prof["demanded"], 1), prof["cells"] / max(prof["align_ms"], 1e-9) / 1e9), flush=True)
    model.close(); del dev, outs; torch.cuda.empty_cache()
