"""Record-set level checks of the C oracle.

The reference has no tests for this path and its predict() cannot be built here without Boost stand-ins, so these
checks are SECONDARY evidence (tests/golden/make_survey_fixture.py explains the provenance of survey_*):
  - GFF3 identical to what the reference code printed for the same inputs (DNA 400 sets, protein 300 sets);
  - the sequence of alignments the reference logged (+ALN lines: pass, i, j, dist, sim) and its STATS counters are
    reproduced exactly for some ordering of records that tie on (score, identities, qstart, qstop) -- the reference
    orders such full ties by heap address (src/alignmentrecord.hh:480), i.e. arbitrarily;
  - hand-made edge sets (n==0, n==1, perfect-hit shortcut, reverse strand, split query);
  - invariants: BandFactor independent of tie order, LCA/isParentOf against a naive walk, stable sort.
"""
import gzip
import itertools
import json
import os
import shutil

import numpy as np
import pytest

from helpers import refio as R


def _load(folder, alphabet, tmp_path, **kw):
    ext = ".fna" if alphabet == 0 else ".faa"
    gz = os.path.join(folder, "ref" + ext + ".gz")
    if os.path.exists(gz):
        work = tmp_path / os.path.basename(folder)
        shutil.copytree(folder, work)
        with gzip.open(gz, "rb") as f, open(work / ("ref" + ext), "wb") as g:
            g.write(f.read())
        folder = str(work)
    return R.Dataset(folder, alphabet, **kw)


def _arrays(oracle, ds):
    tax = oracle.TaxArrays(ds.tax.parent, ds.tax.left, ds.tax.right, ds.tax.depth)
    q = oracle.StoreArrays(ds.query.residues, ds.query.offsets)
    db = oracle.StoreArrays(ds.ref.residues, ds.ref.offsets)
    return tax, q, db


@pytest.mark.parametrize("name,alphabet", [("survey_dna", 0), ("survey_protein", 1)])
def test_gff3_matches_reference_output(oracle, golden_dir, tmp_path, name, alphabet):
    ds = _load(os.path.join(golden_dir, name), alphabet, tmp_path)
    b = ds.batch()
    tax, q, db = _arrays(oracle, ds)
    rc, out, stats = oracle.predict_batch(tax, q, db, alphabet, b["offsets"], b["qseq"], b["qlen"], b["recs"])
    assert rc == 0
    ours = ds.gff3(out, b["qids"], b["qlen"])
    gold = open(os.path.join(golden_dir, name, "out.gff3")).read()
    assert ours == gold


@pytest.mark.parametrize("name,alphabet,nsets", [("survey_dna", 0, 120), ("survey_protein", 1, 60)])
def test_alignment_trace_matches_reference_log(oracle, golden_dir, tmp_path, name, alphabet, nsets):
    ds = _load(os.path.join(golden_dir, name), alphabet, tmp_path)
    b = ds.batch()
    tax, q, db = _arrays(oracle, ds)
    exp = json.load(open(os.path.join(golden_dir, name, "expected.json")))
    checked = 0
    for s in range(nsets):
        lo, hi = b["offsets"][s], b["offsets"][s + 1]
        r = b["recs"][lo:hi]
        ref = [tuple(x) for x in exp[s]["aln"]]
        groups = {}
        for k, x in enumerate(r):
            groups.setdefault((float(x["score"]), int(x["identities"])), []).append(k)
        glist = list(groups.values())
        total = 1
        for g in glist:
            total *= int(np.prod(np.arange(1, len(g) + 1)))
        if total > 3000:
            continue
        ok = False
        for perm in itertools.product(*[itertools.permutations(g) for g in glist]):
            order = np.arange(len(r))
            for g, pg in zip(glist, perm):
                order[list(g)] = pg
            rc, out, st, tr = oracle.predict(tax, q, db, alphabet, b["qseq"][s], b["qlen"][s], r[order], want_trace=True)
            mine = [(p, i, j, round(d, 2), round(sm, 2)) for p, i, j, d, sm in tr]
            if mine == ref and [int(st["pass0"]), int(st["pass1"]), int(st["pass2"])] == exp[s]["stats"]:
                ok = True
                break
        assert ok, "no tie order reproduces the reference log for set %d (%s)" % (s, exp[s]["id"])
        checked += 1
    assert checked > nsets // 2


def _check_ranges(oracle, ds, exp):
    b = ds.batch()
    tax, q, db = _arrays(oracle, ds)
    name2idx = {ds.tax.names[t]: i for i, t in enumerate(ds.tax.taxid)}
    assert len(exp) == len(b["qids"])
    for s, e in enumerate(exp):
        lo, hi = b["offsets"][s], b["offsets"][s + 1]
        rc, out, st, tr = oracle.predict(tax, q, db, ds.alphabet, b["qseq"][s], b["qlen"][s], b["recs"][lo:hi], want_trace=True)
        assert rc == 0
        assert int(st["n"]) == e["numref"]
        rt, lower, upper = (name2idx[n] for n in e["range"])
        assert (int(out["rtax"]), int(out["lower"]), int(out["upper"])) == (rt, lower, upper), e["id"]
        assert [int(st["pass0"]), int(st["pass1"]), int(st["pass2"])] == e["stats"], e["id"]
        if e["numref"] > 0:
            assert e["id"] == "%d:%d@%s" % (out["begin"], out["end"], b["qids"][s])
        else:
            assert e["id"].startswith("-1:-1@") and out["flags"] & 1


def test_edge_sets(oracle, golden_dir):
    folder = os.path.join(golden_dir, "survey_edge")
    ds = R.Dataset(folder, 0, query_name="query_edge.fna", aln_name="aln_edge.tsv")
    exp = json.load(open(os.path.join(folder, "expected_edge.json")))
    assert [e["id"] for e in exp] == ["1:200@e0", "-1:-1@e1", "5:190@e2", "1:200@e3", "1:80@e4", "100:200@e4"]
    _check_ranges(oracle, ds, exp)
    # the reverse-strand hit of e3 must have gone through the reverse complement (dist 106 in the reference log)
    assert [1, 1, 0, 106.0, 94.0] in exp[3]["aln"]


def test_small_set(oracle, golden_dir):
    folder = os.path.join(golden_dir, "survey_edge")
    ds = R.Dataset(folder, 0)
    _check_ranges(oracle, ds, json.load(open(os.path.join(folder, "expected.json"))))


def test_stale_ival_and_gff3_forms(oracle, golden_dir):
    """Quirks 5 and 9: n==0 prints the previous record's ival; zero support prints no lower node."""
    folder = os.path.join(golden_dir, "survey_edge")
    ds = R.Dataset(folder, 0, query_name="query_edge.fna", aln_name="aln_edge.tsv")
    b = ds.batch()
    tax, q, db = _arrays(oracle, ds)
    rc, out, st = oracle.predict_batch(tax, q, db, 0, b["offsets"], b["qseq"], b["qlen"], b["recs"])
    lines = ds.gff3(out, b["qids"], b["qlen"]).splitlines()
    assert lines[1].endswith(";ival=0")          # e0: shortcut, ival 0
    assert "tax=1;rtax=1;ival=0" in lines[2]      # e1: all masked -> root, support 0, STALE ival of e0
    assert lines[3].split("\t")[3:5] == ["5", "190"] and "ival" not in lines[3]  # e2: n==1 -> ival 1 is not printed


def test_bandfactor_tie_order_invariance(oracle, golden_dir):
    ds = R.Dataset(os.path.join(golden_dir, "survey_edge"), 0)
    tax = oracle.TaxArrays(ds.tax.parent, ds.tax.left, ds.tax.right, ds.tax.depth)
    leaves = np.nonzero(ds.tax.depth == ds.tax.depth.max())[0]
    rng = np.random.default_rng(5)
    for _ in range(3000):
        n = int(rng.integers(2, 30))
        node = rng.choice(leaves, n)
        dist = rng.integers(0, 8, n).astype(np.float32) * np.float32(rng.choice([1.0, 0.37]))
        dist[0] = 0
        f0 = tax.bandfactor(dist, node)
        p = np.concatenate([[0], 1 + rng.permutation(n - 1)])
        assert tax.bandfactor(dist[p], node[p]) == f0
        assert f0 >= 1.0


def test_lca_and_ancestry_against_naive_walk(oracle, golden_dir):
    ds = R.Dataset(os.path.join(golden_dir, "survey_edge"), 0)
    t = ds.tax
    tax = oracle.TaxArrays(t.parent, t.left, t.right, t.depth)
    n = len(t.parent)

    def ancestors(x):
        path = [x]
        while x != 0:
            x = int(t.parent[x])
            path.append(x)
        return path

    rng = np.random.default_rng(3)
    for _ in range(5000):
        a, b = (int(v) for v in rng.integers(0, n, 2))
        pa, pb = ancestors(a), set(ancestors(b))
        naive = next(x for x in pa if x in pb)
        assert tax.lca(a, b) == naive
        assert tax.is_parent_of(a, b) == (a != b and a in pb)
    assert int(t.depth.max()) == 7 and t.depth[0] == 0  # root + 7 kept ranks (src/constants.hh:33)


def test_sort_is_stable_by_score_then_identities(oracle):
    rng = np.random.default_rng(11)
    for _ in range(200):
        n = int(rng.integers(1, 40))
        recs = np.zeros(n, oracle.RECORD_DTYPE)
        recs["score"] = rng.integers(0, 5, n)
        recs["identities"] = rng.integers(0, 3, n)
        order = np.zeros(n, np.uint32)
        oracle.lib().orc_sort_records(recs.ctypes.data, n, order.ctypes.data)
        expect = sorted(range(n), key=lambda i: (-float(recs["score"][i]), -int(recs["identities"][i]), i))
        assert list(order) == expect


def test_segment_fetch_rules(oracle):
    """a7: extension, clipping at both ends, reverse complement (taxonpredictionmodelsequence.hh:870-880)."""
    seq = np.array([0, 1, 2, 3, 4, 0, 0, 1, 1, 2], np.uint8)  # ACGTNAACCG
    st = oracle.StoreArrays(seq, [0, 10])
    rc, s = st.segment(0, 0, 3, 6)
    assert rc == 0 and list(s) == [2, 3, 4, 0]
    rc, s = st.segment(0, 0, 3, 6, left_ext=5, right_ext=9)   # clipped to [1, 10]
    assert rc == 0 and list(s) == list(seq)
    rc, s = st.segment(0, 0, 6, 3)                              # reverse strand: RC of [3, 6]
    assert rc == 0 and list(s) == [3, 4, 0, 1]
    rc, s = st.segment(0, 0, 6, 3, left_ext=1, right_ext=2)    # [stop-right_ext, start+left_ext] = [1, 7] RC
    assert rc == 0 and list(s) == [3, 3, 4, 0, 1, 2, 3]
    rc, s = st.segment(0, 0, 11, 12)
    assert rc == 1                                              # start beyond the sequence: range error
    rc, s = st.segment(1, 0, 6, 3)
    assert rc == 3                                              # protein + reverse coordinates: live assert (:864)


def test_bandfactor_sort_free_formulation(oracle, golden_dir):
    """The device computes BandFactor from per-rank (min, max) distances instead of sorting (place_kernels.hip);
    this is that formulation in numpy against the oracle's sorted walk, bit for bit."""
    ds = R.Dataset(os.path.join(golden_dir, "survey_edge"), 0)
    tax = oracle.TaxArrays(ds.tax.parent, ds.tax.left, ds.tax.right, ds.tax.depth)
    depth = ds.tax.depth
    rng = np.random.default_rng(1)

    def sort_free(dist, node):
        anchor = int(node[0])
        ra = int(depth[anchor])
        fmin = np.full(ra + 1, np.float32(np.finfo(np.float32).max), np.float32)
        dmax = np.full(ra + 1, np.float32(-1), np.float32)
        for d, nd in zip(dist[1:], node[1:]):
            r = int(depth[tax.lca(int(nd), anchor)])
            fmin[r] = min(fmin[r], d)
            dmax[r] = max(dmax[r], d)
        bf = np.float32(1)
        remembered, lowmin = [], np.float32(np.finfo(np.float32).max)
        for r in range(ra):
            if fmin[r] < lowmin:
                remembered.append(r)
                lowmin = fmin[r]
        dsuf = dmax[ra]
        for r in range(ra - 1, -1, -1):
            if r in remembered and fmin[r] != 0 and dsuf >= 0:
                bf = max(bf, np.float32(dsuf) / np.float32(fmin[r]))
            dsuf = max(dsuf, dmax[r])
        return np.sqrt(np.float32(bf))

    for _ in range(4000):
        n = int(rng.integers(1, 25))
        node = rng.integers(0, len(depth), n)
        dist = (rng.integers(0, 7, n).astype(np.float32) * np.float32(rng.choice([1.0, 0.37, 2.5]))).astype(np.float32)
        dist[0] = 0
        assert np.float32(tax.bandfactor(dist, node)).tobytes() == np.float32(sort_free(dist, node)).tobytes()
