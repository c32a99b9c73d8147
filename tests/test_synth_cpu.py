"""The synthetic workload generator produces consistent data, and the oracle runs through it."""
import importlib

import numpy as np

import __graft_entry__ as graft


def test_generator_is_deterministic_and_consistent(oracle, tmp_path):
    graft.load_package()
    synth = importlib.import_module("taxator_tk_amd.synth")
    a = synth.make_dataset(n_sets=120, n_hits=10, seg_len=200, branching=2, genome_len=2000, seed=1, variety=True)
    b = synth.make_dataset(n_sets=120, n_hits=10, seg_len=200, branching=2, genome_len=2000, seed=1, variety=True)
    for k in a.batch:
        assert (a.batch[k] == b.batch[k]).all()
    assert len(a.tax.parent) == 128 and len(a.tax.species) == 64          # config 1 taxonomy
    assert a.batch["rstart"].max() <= (np.diff(a.ref_offsets.astype(np.int64)).max())
    tax = oracle.TaxArrays(a.tax.parent, a.tax.left, a.tax.right, a.tax.depth)
    q = oracle.StoreArrays(synth.encode(a.query_letters, 0), a.query_offsets)
    db = oracle.StoreArrays(synth.encode(a.ref_letters, 0), a.ref_offsets)
    rc, out, st = oracle.predict_batch(tax, q, db, 0, a.batch["set_offsets"], a.batch["query_seq"], a.batch["query_len"],
                                       synth.records_aos(a.batch, oracle.RECORD_DTYPE))
    assert rc == 0
    flags = out["flags"]
    assert (flags & 1).any() and (flags & 2).any() and (flags & 4).any()   # n==0, shortcut, n==1 all occur
    assert (st["pass1"] > 0).any() and (st["pass2"] > 0).any()
    # the true species is usually recovered as lower/rtax when nothing is masked
    plain = flags == 0
    assert (out["rtax"][plain] == a.true_node[plain]).mean() > 0.5
    # files round-trip through the independent python readers
    from helpers import refio as R
    synth.write_files(a, str(tmp_path))
    ds = R.Dataset(str(tmp_path), 0)
    fb = ds.batch()
    assert (fb["recs"]["qstart"] == a.batch["qstart"]).all() and (fb["recs"]["filtered"] == a.batch["filtered"]).all()
    tax2 = oracle.TaxArrays(ds.tax.parent, ds.tax.left, ds.tax.right, ds.tax.depth)
    q2 = oracle.StoreArrays(ds.query.residues, ds.query.offsets)
    db2 = oracle.StoreArrays(ds.ref.residues, ds.ref.offsets)
    rc, out2, st2 = oracle.predict_batch(tax2, q2, db2, 0, fb["offsets"], fb["qseq"], fb["qlen"], fb["recs"])
    assert rc == 0
    # same predictions through the file route (node numbering differs: compare taxids)
    t1 = [a.tax.taxid[i] for i in out["lower"]]
    t2 = [ds.tax.taxid[i] for i in out2["lower"]]
    assert t1 == t2 and (out["support"] == out2["support"]).all() and out["ival"].tobytes() == out2["ival"].tobytes()


def test_ragged_dataset_and_big_tree(oracle):
    """tools/ragged_bench.py's workload (round 5): classes of record sets of different shape laid out in one stream.  A class's
    record sets must come out of the mix exactly as they come out alone (same rows from the oracle), the stores and indices must be
    consistent, and embedding the taxonomy in a 100 k-node tree of random clades must change no row (the extra nodes are clades no
    record refers to; the original nodes keep their indices, depths and ancestry)."""
    graft.load_package()
    synth = importlib.import_module("taxator_tk_amd.synth")
    classes = [(60, 1, 100), (40, 3, 50), (30, 20, 200), (4, 100, 300), (2, 300, 150), (6, 8, 900)]
    ds = synth.make_ragged_dataset(classes, seed=3)
    so = ds.batch["set_offsets"].astype(np.int64)
    n = np.diff(so)
    assert ds.n_sets == sum(c[0] for c in classes) and len(ds.batch["qstart"]) == sum(c[0] * c[1] for c in classes) == int(so[-1])
    assert sorted(set(n.tolist())) == sorted({c[1] for c in classes}) and (n[:40] != n[0]).any()          # shuffled: shapes interleave
    assert int(ds.batch["ref_seq"].max()) < len(ds.ref_offsets) - 1 and int(ds.batch["query_seq"].max()) == ds.n_sets - 1
    assert len(set(ds.batch["query_seq"].tolist())) == ds.n_sets
    tax = oracle.TaxArrays(ds.tax.parent, ds.tax.left, ds.tax.right, ds.tax.depth)
    q = oracle.StoreArrays(synth.encode(ds.query_letters, 0), ds.query_offsets)
    db = oracle.StoreArrays(synth.encode(ds.ref_letters, 0), ds.ref_offsets)
    recs = synth.records_aos(ds.batch, oracle.RECORD_DTYPE)
    rc, rows, st = oracle.predict_batch(tax, q, db, 0, ds.batch["set_offsets"], ds.batch["query_seq"], ds.batch["query_len"], recs)
    assert rc == 0 and int(st["pass1"].sum()) > 1000
    k = 2  # the 20-hit class alone
    alone = synth.make_dataset(n_sets=classes[k][0], n_hits=classes[k][1], seg_len=classes[k][2], branching=4, genome_len=max(2000, 4 * classes[k][2]), seed=3 * 1000 + k)
    aq = oracle.StoreArrays(synth.encode(alone.query_letters, 0), alone.query_offsets)
    adb = oracle.StoreArrays(synth.encode(alone.ref_letters, 0), alone.ref_offsets)
    rc2, rows2, st2 = oracle.predict_batch(tax, aq, adb, 0, alone.batch["set_offsets"], alone.batch["query_seq"], alone.batch["query_len"],
                                           synth.records_aos(alone.batch, oracle.RECORD_DTYPE))
    members = np.nonzero(ds.set_class == k)[0]
    members = members[np.argsort(ds.batch["query_seq"][members])]   # a class's record sets keep their order inside the class's query block
    assert rc2 == 0 and rows[members].tobytes() == rows2.tobytes() and st[members].tobytes() == st2.tobytes()
    big = synth.embed_in_big_tree(ds.tax, 100_000, 7)
    n0 = len(ds.tax.parent)
    assert len(big.parent) == 100_000 and int(big.depth.max()) <= 7
    assert (big.parent[:n0] == np.asarray(ds.tax.parent)).all() and (big.depth[:n0] == np.asarray(ds.tax.depth)).all()
    inside = (big.left[big.parent[1:]] < big.left[1:]) & (big.right[big.parent[1:]] > big.right[1:])
    assert inside.all() and sorted(np.concatenate([big.left, big.right]).tolist()) == list(range(1, 200_001))
    btax = oracle.TaxArrays(big.parent, big.left, big.right, big.depth)
    rc3, rows3, st3 = oracle.predict_batch(btax, q, db, 0, ds.batch["set_offsets"], ds.batch["query_seq"], ds.batch["query_len"], recs)
    assert rc3 == 0 and rows3.tobytes() == rows.tobytes() and st3.tobytes() == st.tobytes()
