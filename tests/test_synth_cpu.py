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
