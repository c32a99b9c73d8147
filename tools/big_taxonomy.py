#!/usr/bin/env python3
"""tools/big_taxonomy.py [n_nodes] [max_depth] -- the engine on a taxonomy of NCBI size kept whole (taxator -d 0: ~2.5 M nodes,
~40 levels): time and HBM of rpa_create (preorder numbering, ancestor id / interval tables), then one RPA batch and one
alignment-free batch whose records point at random nodes of the big tree, both against the CPU oracle.
Test infrastructure (uses oracle/); run on the GPU box."""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as graft
from helpers import oracle as O

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_500_000
max_depth = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rng = np.random.default_rng(7)
t0 = time.time()
# random tree, vectorised level by level: node i hangs under a random earlier node whose depth is below the limit
parent = np.zeros(n, np.uint32)
depth = np.zeros(n, np.uint8)
lo, prev_lo = 1, 0
while lo < n:
    hi = min(n, lo + 1000 if lo < 4000 else int(lo * 1.22))
    # four of five new nodes hang under a node of the previous batch (the tree grows a level per batch), the rest anywhere
    deep = rng.random(hi - lo) < 0.8
    cand = np.where(deep, rng.integers(prev_lo, lo, hi - lo), rng.integers(0, lo, hi - lo))
    cand[depth[cand] >= max_depth] = 0
    parent[lo:hi] = cand
    depth[lo:hi] = depth[cand] + 1
    prev_lo, lo = lo, hi
# nested-set numbers by an iterative walk over children lists (numpy argsort by parent)
order = np.argsort(parent[1:], kind="stable") + 1
start = np.searchsorted(parent[order], np.arange(n + 1))
left = np.zeros(n, np.uint32); right = np.zeros(n, np.uint32)
nxt = start[:-1].copy()
stack = [0]; ctr = 1; left[0] = 1
while stack:
    v = stack[-1]
    if nxt[v] < start[v + 1]:
        c = order[nxt[v]]; nxt[v] += 1
        ctr += 1; left[c] = ctr
        stack.append(int(c))
    else:
        ctr += 1; right[v] = ctr
        stack.pop()
print("tree: %d nodes, depth %d, built in %.1f s" % (n, int(depth.max()), time.time() - t0), flush=True)


class T: pass
tax = T(); tax.parent, tax.left, tax.right, tax.depth = parent, left, right, depth
pkg = graft.load_package()
synth = importlib.import_module("taxator_tk_amd.synth")
import torch
free0 = torch.cuda.mem_get_info()[0]
t0 = time.time()
model = pkg.RPAPredictionModel(tax, alphabet=pkg.DNA)
print("rpa_create: %.2f s, %.2f GB of HBM for the taxonomy tables" % (time.time() - t0, (free0 - torch.cuda.mem_get_info()[0]) / 1e9), flush=True)
# a workload on the small mock tree, its taxa re-pointed at random deep nodes of the big one (species -> a random node each)
ds = synth.make_dataset(n_sets=3000, n_hits=20, seg_len=150, branching=4, genome_len=2500, seed=5, variety=True)
remap = rng.integers(1, n, len(ds.tax.parent)).astype(np.uint32)
ds.batch["node"] = remap[ds.batch["node"]]
model.load_store(pkg.STORE_QUERY, ds.query_letters, ds.query_offsets)
model.load_store(pkg.STORE_DB, ds.ref_letters, ds.ref_offsets)
t0 = time.time()
rows, stats = model.predict(ds.batch)
t1 = time.time()
otax = O.TaxArrays(parent, left, right, depth)
q = O.StoreArrays(synth.encode(ds.query_letters, 0), ds.query_offsets)
db = O.StoreArrays(synth.encode(ds.ref_letters, 0), ds.ref_offsets)
rc, exp, est = O.predict_batch(otax, q, db, 0, ds.batch["set_offsets"], ds.batch["query_seq"], ds.batch["query_len"], synth.records_aos(ds.batch, O.RECORD_DTYPE))
ok = rc == 0 and rows.tobytes() == exp.tobytes() and stats.tobytes() == est.tobytes()
print("rpa batch (3000 sets x 20 hits, taxa = random nodes of the big tree): engine %.3f s, %s; results at depths %s" % (
    t1 - t0, "identical to the oracle" if ok else "MISMATCH", sorted(set(int(x) for x in depth[rows["lower"]]))[:12]), flush=True)
import test_lca_models as TL
tree = T(); tree.parent, tree.left, tree.right, tree.depth = parent, left, right, depth
offs, qlen, recs, evalue = TL.random_batch(rng, tree, 50000, 30)
uncl = (rng.random(n) < 0.1).astype(np.uint8)
rc, exp = O.lca_predict_batch(otax, O.LCA_MEGAN, offs, qlen, recs, evalue, uncl, 0.3, 0.0, 1000.0, 2, 1, True)
batch = dict(set_offsets=offs, query_len=qlen, filtered=recs["filtered"].astype(np.uint8))
for f in ("qstart", "qstop", "score", "node"):
    batch[f] = recs[f].copy()
t0 = time.time()
got = model.lca_predict(batch, O.LCA_MEGAN, evalue=evalue, node_unclassified=uncl, toppercent=0.3, minscore=0.0, maxevalue=1000.0, minsupport=2, nbest=1, ignore_unclassified=True)
ok2 = rc == 0 and got.tobytes() == exp.tobytes()
print("megan-lca batch (50000 sets): engine %.3f s, %s" % (time.time() - t0, "identical to the oracle" if ok2 else "MISMATCH"), flush=True)
sys.exit(0 if ok and ok2 else 1)
