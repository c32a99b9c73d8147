#!/usr/bin/env python3
"""tools/soak_lca.py [N] -- differential soak of the alignment-free models (rpa_lca_predict_batch) against the CPU oracle on
the GPU box: random taxonomies, batches and option values (-t -m -e -c -n -u).  Test infrastructure (uses oracle/ and the
generators of tests/test_lca_models.py)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as graft
from helpers import oracle as O
import test_lca_models as T

pkg = graft.load_package()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(2024)
t0 = time.time(); total = 0
for k in range(n):
    tree = T.Tree(rng, n_nodes=int(rng.integers(5, 6000)), max_depth=int(rng.choice([8, 8, 25, 60])))  # (deep trees: chains, see T.Tree)
    uncl = (rng.random(len(tree.parent)) < rng.choice([0.0, 0.2, 0.6])).astype(np.uint8)
    n_sets, max_hits = int(rng.integers(1, 5000)), int(rng.integers(1, 70))
    offs, qlen, recs, evalue = T.random_batch(rng, tree, n_sets, max_hits)
    c = dict(model=int(rng.integers(0, 4)), toppercent=float(rng.choice([0.0, 0.05, 0.3, 1.0])), minscore=float(rng.choice([0.0, 30.0, 200.0])),
             maxevalue=float(rng.choice([1000.0, 1e-5, 1e-40])), minsupport=int(rng.integers(0, 6)), nbest=int(rng.integers(0, 8)),
             ignore_unclassified=bool(rng.integers(0, 2)))
    tax = O.TaxArrays(tree.parent, tree.left, tree.right, tree.depth)
    rc, exp = O.lca_predict_batch(tax, c["model"], offs, qlen, recs, evalue, uncl, c["toppercent"], c["minscore"], c["maxevalue"],
                                  c["minsupport"], c["nbest"], c["ignore_unclassified"])
    assert rc == 0
    model = pkg.RPAPredictionModel(tree, alphabet=pkg.DNA)
    batch = dict(set_offsets=offs, query_len=qlen, filtered=recs["filtered"].astype(np.uint8))
    for f in ("qstart", "qstop", "score", "node"):
        batch[f] = recs[f].copy()
    rows = model.lca_predict(batch, c["model"], evalue=evalue, node_unclassified=uncl, toppercent=c["toppercent"], minscore=c["minscore"],
                             maxevalue=c["maxevalue"], minsupport=c["minsupport"], nbest=c["nbest"], ignore_unclassified=c["ignore_unclassified"])
    ok = rows.tobytes() == exp.tobytes()
    total += len(offs) - 1
    print("batch %d: nodes=%d sets=%d %s -> %s (%.0f s)" % (k, len(tree.parent), len(offs) - 1, c, "ok" if ok else "MISMATCH", time.time() - t0), flush=True)
    if not ok:
        sys.exit(1)
print("lca soak ok: %d batches, %d record sets" % (n, total))
