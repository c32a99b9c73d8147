"""BASELINE configs 2, 3, 4 and 5 at their full per-GPU sizes (1 M sets x 20 hits x 200 bp; 1.25 M sets = one GPU's share of
config 3's 10 M, seeds 3 and 10; 500 k protein sets x 20 hits x 67 aa; 12.5 k contigs x 50 hits x 10 kbp = one GPU's share
of config 5) through properties that do not need the CPU
oracle to finish the whole batch: determinism, independence of record sets (a batch in reversed set order gives the
reversed rows), structural invariants of every row, and bit-exact agreement with the oracle on two windows of the same
batch (front and back) that together cover 10 % of it (the oracle runs them in chunks on the box's host cores).  The
device batch is generated on the GPU exactly as bench.py does."""
import importlib
import os

import numpy as np
import pytest

import __graft_entry__ as graft
from helpers import oracle as O

pytestmark = pytest.mark.gpu


# the per-GPU workloads of bench.py's CONFIGS; window = record sets per oracle window (sized for seconds of CPU time)
FULL = {
    "config2": dict(S=1_000_000, NH=20, seg_len=200, alphabet=0, genome_len=4000, level_rate=0.03, query_rate=0.05, window=50_000,
                    demanded=(20_000_000, 40_000_000)),
    "config4": dict(S=500_000, NH=20, seg_len=67, alphabet=1, genome_len=1500, level_rate=0.06, query_rate=0.08, window=25_000,
                    demanded=(1_500_000, 8_000_000)),
    # config 3 = 10 M record sets over 8 GPUs: one GPU's share, with the first and the last of the per-shard seeds bench.py
    # --config c3 uses (3 + rank); what is left untested is only "across 8 devices"
    "config3_seed3": dict(S=1_250_000, NH=20, seg_len=200, alphabet=0, genome_len=4000, level_rate=0.03, query_rate=0.05, window=25_000,
                          demanded=(25_000_000, 50_000_000), seed=3),
    "config3_seed10": dict(S=1_250_000, NH=20, seg_len=200, alphabet=0, genome_len=4000, level_rate=0.03, query_rate=0.05, window=25_000,
                           demanded=(25_000_000, 50_000_000), seed=10),
    "config5": dict(S=12_500, NH=50, seg_len=10_000, alphabet=0, genome_len=40_000, level_rate=0.03, query_rate=0.05, window=48,
                    demanded=(300_000, 1_500_000)),
}


@pytest.mark.parametrize("name", list(FULL))
def test_full_size_properties(name):
    import torch
    cfg = FULL[name]
    pkg = graft.load_package()
    synth = importlib.import_module("taxator_tk_amd.synth")
    S, NH, alphabet, W = cfg["S"], cfg["NH"], cfg["alphabet"], cfg["window"]
    ds = synth.make_dataset(n_sets=S, n_hits=NH, seg_len=cfg["seg_len"], branching=4, genome_len=cfg["genome_len"], seed=cfg.get("seed", 2), device="cuda",
                            alphabet=alphabet, level_rate=cfg["level_rate"], query_rate=cfg["query_rate"])
    model = pkg.RPAPredictionModel(ds.tax, alphabet=alphabet)
    model.load_store(pkg.STORE_QUERY, ds.query_letters, ds.query_offsets)
    model.load_store(pkg.STORE_DB, ds.ref_letters, ds.ref_offsets)
    dev = {k: torch.from_numpy(np.ascontiguousarray(v).view(np.int32) if v.dtype == np.uint32 else np.ascontiguousarray(v)).cuda()
           for k, v in ds.batch.items()}

    def run(batch):
        rows = torch.zeros((S, 8), dtype=torch.int32, device="cuda")
        stats = torch.zeros((S, 4), dtype=torch.int32, device="cuda")
        model.predict(batch, mem_kind=pkg.MEM_DEVICE, out=rows, stats=stats)
        torch.cuda.synchronize()
        return rows, stats

    rows, stats = run(dev)
    rows2, stats2 = run(dev)
    assert torch.equal(rows, rows2) and torch.equal(stats, stats2)                      # deterministic

    # independence: the same record sets in reversed order (every set has NH records here)
    rev = {}
    for k, v in dev.items():
        if k == "set_offsets":
            rev[k] = v
        elif v.numel() == S:
            rev[k] = torch.flip(v, dims=[0]).contiguous()
        else:
            rev[k] = torch.flip(v.view(S, NH), dims=[0]).contiguous().view(-1)
    torch.cuda.synchronize()  # the engine runs on its own stream: the flips above (torch's stream) must have finished
    rrows, rstats = run(rev)
    assert torch.equal(torch.flip(rrows, dims=[0]), rows) and torch.equal(torch.flip(rstats, dims=[0]), stats)

    r = rows.cpu().numpy().view(pkg.PRED_DTYPE).reshape(-1)
    st = stats.cpu().numpy().view(pkg.STATS_DTYPE).reshape(-1)
    tax = ds.tax
    # every row: upper is lower or an ancestor of it, rtax lies at or below lower's lineage start, ranges inside the query
    left, right, depth = np.asarray(tax.left), np.asarray(tax.right), np.asarray(tax.depth)
    lo, up = r["lower"], r["upper"]
    assert ((left[up] <= left[lo]) & (right[up] >= right[lo])).all() and (depth[up] <= depth[lo]).all()
    assert ((left[lo] <= left[r["rtax"]]) & (right[lo] >= right[r["rtax"]]) | (r["flags"] != 0)).all()   # rtax at or below lower (multi-hit rows)
    qlen = np.asarray(ds.batch["query_len"])
    assert (r["begin"] >= 1).all() and (r["end"] <= qlen).all() and (r["begin"] <= r["end"]).all()
    ival = r["ival"][r["flags"] == 0]
    assert ((ival >= 0) & (ival <= 1)).all()
    assert (st["n"] == NH).all()
    if name.startswith(("config2", "config3")):
        assert int(st["pass0"].sum()) > S // 2 and int(st["pass1"].sum()) > 10 * S
    demanded = int(st["pass0"].astype(np.int64).sum() + st["pass1"].astype(np.int64).sum() + st["pass2"].astype(np.int64).sum())
    assert cfg["demanded"][0] < demanded < cfg["demanded"][1], demanded

    # two windows against the oracle, bit for bit
    otax = O.TaxArrays(tax.parent, tax.left, tax.right, tax.depth)
    q = O.StoreArrays(synth.encode(ds.query_letters, alphabet), ds.query_offsets)
    db = O.StoreArrays(synth.encode(ds.ref_letters, alphabet), ds.ref_offsets)
    b = ds.batch

    def check(lo, hi, r=r, st=st, x=0.5, t=0.05):  # record sets [lo, hi) through the oracle (the C library releases the GIL: chunks run side by side)
        r0, r1 = int(b["set_offsets"][lo]), int(b["set_offsets"][hi])
        sub = {k: (v[r0:r1] if len(v) == len(b["qstart"]) else v) for k, v in b.items()}
        offs = (b["set_offsets"][lo:hi + 1] - b["set_offsets"][lo]).astype(np.uint32)
        rc, exp, est = O.predict_batch(otax, q, db, alphabet, offs, b["query_seq"][lo:hi], b["query_len"][lo:hi], synth.records_aos(sub, O.RECORD_DTYPE),
                                       exclude_factor=x, reeval_bandwidth=t)
        return rc == 0 and r[lo:hi].tobytes() == exp.tobytes() and st[lo:hi].tobytes() == est.tobytes()

    from concurrent.futures import ThreadPoolExecutor
    workers = max(1, min(16, os.cpu_count() or 1))
    chunk = max(1, W // (4 * workers))
    spans = [(lo, min(lo + chunk, w0 + W)) for w0 in (0, S - W) for lo in range(w0, w0 + W, chunk)]
    with ThreadPoolExecutor(workers) as ex:
        results = list(ex.map(lambda sp: check(*sp), spans))
    assert all(results), [sp for sp, ok in zip(spans, results) if not ok][:5]
    assert 2 * W * 10 >= S or name not in ("config2", "config4")   # the two windows cover 10 % of configs 2 and 4 (4 % of config 3's shares)

    # `-x` / `-t` off their defaults at full size (the production pipelines pass -x; extended/binning-last.bash:108): the whole
    # batch through an engine with the setting, one oracle window each (front for the first, back for the second)
    if name in ("config2", "config4"):
        del model
        for k, (x, t) in enumerate(((0.0, 0.1), (1.0, 0.0))):
            mx = pkg.RPAPredictionModel(ds.tax, alphabet=alphabet, exclude_factor=x, reeval_bandwidth=t)
            mx.load_store(pkg.STORE_QUERY, ds.query_letters, ds.query_offsets)
            mx.load_store(pkg.STORE_DB, ds.ref_letters, ds.ref_offsets)
            rows_x = torch.zeros((S, 8), dtype=torch.int32, device="cuda")
            stats_x = torch.zeros((S, 4), dtype=torch.int32, device="cuda")
            mx.predict(dev, mem_kind=pkg.MEM_DEVICE, out=rows_x, stats=stats_x)
            torch.cuda.synchronize()
            rx = rows_x.cpu().numpy().view(pkg.PRED_DTYPE).reshape(-1)
            sx = stats_x.cpu().numpy().view(pkg.STATS_DTYPE).reshape(-1)
            dem_x = int(sx["pass0"].astype(np.int64).sum() + sx["pass1"].astype(np.int64).sum() + sx["pass2"].astype(np.int64).sum())
            assert (dem_x >= demanded) if x == 0.0 else (dem_x < demanded), (x, dem_x, demanded)  # X = 0: no heuristic cut-off; X = 1: nearly everything cut
            w0 = 0 if k == 0 else S - W // 2
            spans_x = [(lo, min(lo + chunk, w0 + W // 2)) for lo in range(w0, w0 + W // 2, chunk)]
            with ThreadPoolExecutor(workers) as ex:
                results = list(ex.map(lambda sp: check(sp[0], sp[1], r=rx, st=sx, x=x, t=t), spans_x))
            assert all(results), (x, t, [sp for sp, ok in zip(spans_x, results) if not ok][:5])
            del mx
