#!/usr/bin/env python3
"""tools/huge_set.py [N_HITS] -- what ONE huge record set costs (k_rank_records scans a set once per record: O(n^2)): a batch of
2 record sets x N_HITS records (default 30000) on a taxonomy of 8^6 species, 40 nt segments; engine against the oracle, wall time
of the predict call and the engine's own profile.  Test infrastructure (uses oracle/)."""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as graft
from helpers import oracle as O
pkg = graft.load_package()
synth = importlib.import_module("taxator_tk_amd.synth")
n_hits = int(sys.argv[1]) if len(sys.argv) > 1 else 30000
t0 = time.time()
ds = synth.make_dataset(n_sets=2, n_hits=n_hits, seg_len=40, branching=8, genome_len=400, seed=5)
print("dataset in %.1f s: 2 sets x %d records, taxonomy of %d nodes" % (time.time() - t0, n_hits, len(ds.tax.parent)), flush=True)
model = pkg.RPAPredictionModel(ds.tax, alphabet=0)
model.load_store(pkg.STORE_QUERY, ds.query_letters, ds.query_offsets)
model.load_store(pkg.STORE_DB, ds.ref_letters, ds.ref_offsets)
model.predict(ds.batch)
model.profile(enable=True, reset=True)
t0 = time.time()
rows, stats = model.predict(ds.batch)
dt = time.time() - t0
prof = model.profile()
print("predict: %.1f ms wall (device %.1f ms, alignment kernels %.1f ms, %d rounds, %d alignments computed)" % (dt * 1e3, prof["total_ms"], prof["align_ms"], prof["rounds"], prof["pairs"]), flush=True)
tax = O.TaxArrays(ds.tax.parent, ds.tax.left, ds.tax.right, ds.tax.depth)
q = O.StoreArrays(synth.encode(ds.query_letters, 0), ds.query_offsets)
db = O.StoreArrays(synth.encode(ds.ref_letters, 0), ds.ref_offsets)
t0 = time.time()
rc, exp, est = O.predict_batch(tax, q, db, 0, ds.batch["set_offsets"], ds.batch["query_seq"], ds.batch["query_len"], synth.records_aos(ds.batch, O.RECORD_DTYPE))
print("oracle: rc %d in %.1f s; rows equal: %s, stats equal: %s" % (rc, time.time() - t0, rows.tobytes() == exp.tobytes(), stats.tobytes() == est.tobytes()))
