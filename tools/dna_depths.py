#!/usr/bin/env python3
"""tools/dna_depths.py [extra ...] -- the placement path on config 2's shape (200 nt, 20 hits, 200 k record sets) over taxonomies of
growing depth: `extra` single-child nodes in the edge above every node of the mock tree (0 = the seven ranks; 1, 2, 4, 6, 8 -> 14 ... 63
levels: rows of 2, 3, 4, 6, 8 quads).  Steps in flight, the synchronous step, alignment ms.  Measurement tool."""
import collections, importlib, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft


def main():
    pkg = graft.load_package()
    synth = importlib.import_module("taxator_tk_amd.synth")
    import torch
    extras = [int(x) for x in sys.argv[1:]] or [0, 1, 2, 4, 6, 8]
    ds = synth.make_dataset(n_sets=200000, n_hits=20, seg_len=200, branching=4, genome_len=4000, seed=1, device="cpu")
    for ex in extras:
        st = synth.StretchedTaxonomy(ds.tax, (lambda i: ex)) if ex else None
        tax = st if st else ds.tax
        batch = dict(ds.batch, node=st.new_index[ds.batch["node"]].astype(np.uint32)) if st else ds.batch
        model = pkg.RPAPredictionModel(tax, alphabet=pkg.DNA)
        model.load_store(pkg.STORE_QUERY, ds.query_letters, ds.query_offsets)
        model.load_store(pkg.STORE_DB, ds.ref_letters, ds.ref_offsets)
        dev = {k: torch.from_numpy(np.ascontiguousarray(v).view(np.int32) if v.dtype == np.uint32 else np.ascontiguousarray(v)).cuda() for k, v in batch.items()}
        S = ds.n_sets
        outs = [(torch.zeros((S, 8), dtype=torch.int32, device="cuda"), torch.zeros((S, 4), dtype=torch.int32, device="cuda")) for _ in range(5)]
        stream = torch.cuda.Stream()
        with torch.cuda.stream(stream):
            model.set_stream(stream.cuda_stream)
            model.predict(dev, mem_kind=pkg.MEM_DEVICE, out=outs[0][0], stats=outs[0][1])
            torch.cuda.synchronize()
            steps = 8
            t1 = time.perf_counter()
            for _ in range(steps): model.predict(dev, mem_kind=pkg.MEM_DEVICE, out=outs[0][0], stats=outs[0][1])
            torch.cuda.synchronize()
            sync = (time.perf_counter() - t1) / steps * 1e3
            jobs = collections.deque()

            def go(n):
                for k in range(n):
                    r_, s_ = outs[k % 5]
                    jobs.append(model.submit(dev, out=r_, stats=s_, mem_kind=pkg.MEM_DEVICE))
                    if len(jobs) == 4: jobs.popleft().wait()
                while jobs: jobs.popleft().wait()
            go(5); torch.cuda.synchronize()
            t1 = time.perf_counter(); go(steps); torch.cuda.synchronize()
            infl = (time.perf_counter() - t1) / steps * 1e3
            model.profile(enable=True, reset=True)
            model.predict(dev, mem_kind=pkg.MEM_DEVICE, out=outs[0][0], stats=outs[0][1])
            torch.cuda.synchronize()
            prof = model.profile()
        levels = int(np.asarray(tax.depth).max())
        print("extra %d: %2d levels: %6.2f ms/step in flight, %6.2f sync, align %5.2f, placement+host %5.2f" % (ex, levels, infl, sync, prof["align_ms"], prof["total_ms"] - prof["align_ms"]), flush=True)
        model.close()
        del dev, outs
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
