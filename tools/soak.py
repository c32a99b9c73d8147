#!/usr/bin/env python3
"""tools/soak.py [N_SEEDS] -- differential soak on the GPU box: many seeded batches of every shape the parity tests use
(DNA / protein, ragged sets, edge-case injection, random speculation chunks (with and without the paused loop's thresholds) and round-scheduling knobs, taxonomies of up to
63 levels, 8- / 16- / 32-bit path labels, whole or windowed reference stores) through the engine and the CPU oracle; stops at the first row or counter that differs.  Test infrastructure (uses oracle/)."""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as graft
from helpers import oracle as O

KNOBS = ("RPA_PASS1_FILTER", "RPA_LABEL_BITS", "RPA_SPEC1", "RPA_SPEC2", "RPA_SPEC_GROW", "RPA_SPARSE_DECIDE", "RPA_SPARSE_EMIT", "RPA_NO_COMBINED", "RPA_PARTS",
         "RPA_BAND_MARGIN", "RPA_BAND_SHORT_MIN", "RPA_BAND_WORDS", "RPA_BAND4_PHASES", "RPA_BAND_DUO", "RPA_BAND", "RPA_BAND_SHORT", "RPA_LONG_KERNEL",
         "RPA_PROTEIN_LANES", "RPA_SHORT_WAVE_MAX", "RPA_DEFER2", "RPA_SIDE_STREAMS", "RPA_BAND_WINDOW", "RPA_BAND_WINDOW_WIDE", "RPA_BIG_SET_MIN")


def fill_windows(pkg, model, ds, tile):
    """The reference store as a windowed one: the tiles under every hit's range extended by its set's query range, as maximal
    runs (what the host driver's makeResident appends), in one call."""
    b = ds.batch
    roff = ds.ref_offsets.astype(np.int64)
    rlen = np.diff(roff)
    model.reserve_windows(pkg.STORE_DB, rlen)
    so = b["set_offsets"].astype(np.int64)
    set_of = np.repeat(np.arange(len(so) - 1), np.diff(so))
    span = np.array([int(b["qstop"][lo:hi].max()) - int(b["qstart"][lo:hi].min()) if hi > lo else 0 for lo, hi in zip(so[:-1], so[1:])], np.int64)
    seq = b["ref_seq"].astype(np.int64)
    ok = (seq < len(rlen)) & (np.minimum(b["rstart"], b["rstop"]) >= 1)
    lo = np.minimum(b["rstart"], b["rstop"]).astype(np.int64)[ok] - 1 - span[set_of][ok]
    hi = np.maximum(b["rstart"], b["rstop"]).astype(np.int64)[ok] + span[set_of][ok]
    seq = seq[ok]
    lo, hi = np.maximum(lo, 0), np.minimum(hi, rlen[seq])
    keep = hi > lo
    tiles = {}
    for s_, a, e in zip(seq[keep], lo[keep] // tile, (hi[keep] - 1) // tile):
        tiles.setdefault(int(s_), set()).update(range(int(a), int(e) + 1))
    wins = []
    for s_, ts in tiles.items():
        ts = sorted(ts); a = p_ = ts[0]
        for t in ts[1:] + [None]:
            if t is None or t != p_ + 1:
                wins.append((s_, a * tile, min((p_ + 1) * tile, int(rlen[s_]))))
                a = t
            p_ = t
    if wins:
        model.append_windows(pkg.STORE_DB, [w[0] for w in wins], [w[1] for w in wins], [ds.ref_letters[roff[w[0]] + w[1]: roff[w[0]] + w[2]] for w in wins])


def run(n_seeds=40, rng_seed=12345, first_seed=1000, verbose=True):
  """Returns (batches, record sets, batches on which engine and oracle both abort); raises AssertionError at the first difference."""
  pkg = graft.load_package()
  synth = importlib.import_module("taxator_tk_amd.synth")
  rng = np.random.default_rng(rng_seed)
  t0 = time.time()
  total_sets = 0
  aborts = 0
  for k in range(n_seeds):
      alphabet = int(rng.random() < 0.35)
      n_hits = int(rng.integers(2, 41))
      branching = int(rng.integers(2, 5))
      if alphabet == 0:
          seg = int(rng.choice([40, 90, 150, 200, 257, 320, 400, 520, 700, 1000, 1500]))  # (round 5: 257 ... 384 one strip, 385 ... 2048 the ring window first)
          n_sets = int(rng.integers(50, 600)) if seg < 400 else int(rng.integers(10, 60))
          kw = dict(level_rate=float(rng.choice([0.01, 0.03, 0.06])), query_rate=float(rng.choice([0.0, 0.05, 0.1])))
      else:
          seg = int(rng.choice([20, 67, 120, 253, 312, 400]))
          n_sets = int(rng.integers(30, 300))
          kw = dict(level_rate=float(rng.choice([0.03, 0.06, 0.12])), query_rate=float(rng.choice([0.0, 0.08, 0.15])))
      if rng.random() < 0.15: n_hits = int(rng.integers(41, 90))  # beyond k_decide's 64-hit bit sets
      n_species = branching ** 6
      n_hits = min(n_hits, n_species)
      ds = synth.make_dataset(n_sets=n_sets, n_hits=n_hits, seg_len=seg, branching=branching, genome_len=max(4 * seg, 1200),
                              seed=first_seed + k, alphabet=alphabet, variety=True, **kw)
      # round 3: deep taxonomies (chains of extra nodes in the edges: up to 63 levels), wider path labels than needed,
      # and the reference store filled through windows (only the tiles the batch reads) instead of whole
      deep = rng.random() < 0.3
      if deep:
          mode = int(rng.integers(0, 3))
          depth0 = ds.tax.depth.copy()
          extra = {0: (lambda i: 1), 1: (lambda i: 4 if depth0[i] < 7 else 10), 2: (lambda i: (i * 7 + 3) % 10 if depth0[i] > 1 else 0)}[mode]
          synth.stretch_dataset(ds, extra)
      windowed = rng.random() < 0.3
      env = {}
      if rng.random() < 0.2:  # (the widest row holds 832 bits of labels: wider labels only where the tree's depth leaves room)
          bits = int(rng.choice([16, 32]))
          if ((int(ds.tax.depth.max()) + 7) // 8 * 8) * bits <= 832: env["RPA_LABEL_BITS"] = str(bits)
      if rng.random() < 0.5:
          env.update({"RPA_SPEC1": str(int(rng.integers(0, 9))), "RPA_SPEC2": str(int(rng.integers(0, 9))),
                      "RPA_SPEC_GROW": str(int(rng.integers(0, 2))), "RPA_SPARSE_DECIDE": str(int(rng.choice([1, 3, 6, 24, 100000]))),
                      "RPA_SPARSE_EMIT": str(int(rng.choice([1, 6, 24, 100000])))})
          if rng.random() < 0.4:
              env["RPA_PASS1_FILTER"] = str(int(rng.choice([0, 1, 3, 4])))
          if rng.random() < 0.3:
              env["RPA_NO_COMBINED"] = "1"
          if rng.random() < 0.3:
              env["RPA_PARTS"] = str(int(rng.integers(2, 5)))
      # the exact bands: forced onto launches of any size, margins from "every pair is retried" to generous, or switched off
      if rng.random() < 0.6:
          env["RPA_BAND_SHORT_MIN"] = "1"
          env["RPA_BAND_MARGIN"] = str(int(rng.choice([1, 8, 16, 21, 40])))
          if rng.random() < 0.3: env["RPA_BAND_WORDS"] = "5"
          elif rng.random() < 0.3: env["RPA_BAND4_PHASES"] = "2"
          if rng.random() < 0.3: env["RPA_BAND_DUO"] = "0"
          if rng.random() < 0.15: env["RPA_BAND"] = "0"
          if rng.random() < 0.15: env["RPA_BAND_SHORT"] = "0"
          if rng.random() < 0.15: env["RPA_BAND_WINDOW"] = "0"
          elif rng.random() < 0.3: env["RPA_BAND_WINDOW_WIDE"] = "1"
          if alphabet == 0 and seg > 256 and rng.random() < 0.5: env["RPA_LONG_KERNEL"] = "lane"
      if alphabet == 1 and rng.random() < 0.3: env["RPA_PROTEIN_LANES"] = "0"
      if alphabet == 0 and rng.random() < 0.4: env["RPA_SHORT_WAVE_MAX"] = str(int(rng.choice([0, 64, 1000000])))  # round 4: K1w (a pattern word per lane) never / for tiny launches / always
      if rng.random() < 0.6: env["RPA_DEFER2"] = str(rng.choice(["0", "1", "1", "auto"]))  # round 5: two launches per pass-2 round (kDeferredMark) never / always / by pattern length
      if rng.random() < 0.5: env["RPA_BIG_SET_MIN"] = str(int(rng.choice([0, 2, 7, 30])))  # round 5: which record sets a workgroup of its own ranks (default: from 192 records on)
      if rng.random() < 0.3: env["RPA_SIDE_STREAMS"] = "0"  # round 5: the launches of a round one after the other on the engine's stream
      for kk in KNOBS:
          os.environ.pop(kk, None)
      os.environ.update(env)
      # round 4: `-x` / `-t` off their defaults in half of the batches (taxator.cpp:286-287; no range check there)
      x, t = 0.5, 0.05
      if rng.random() < 0.5:
          x = float(rng.choice([0.0, 0.1, 0.25, 0.5, 0.8, 0.95, 1.0, 1.3]))
          t = float(rng.choice([0.0, 0.01, 0.05, 0.1, 0.3, 0.7, 1.0]))
      model = pkg.RPAPredictionModel(ds.tax, alphabet=alphabet, exclude_factor=x, reeval_bandwidth=t)
      model.load_store(pkg.STORE_QUERY, ds.query_letters, ds.query_offsets)
      if not windowed:
          model.load_store(pkg.STORE_DB, ds.ref_letters, ds.ref_offsets)
      else:
          fill_windows(pkg, model, ds, int(rng.choice([64, 256, 1000])))
      tax = O.TaxArrays(ds.tax.parent, ds.tax.left, ds.tax.right, ds.tax.depth)
      q = O.StoreArrays(synth.encode(ds.query_letters, alphabet), ds.query_offsets)
      db = O.StoreArrays(synth.encode(ds.ref_letters, alphabet), ds.ref_offsets)
      rc, exp, est = O.predict_batch(tax, q, db, alphabet, ds.batch["set_offsets"], ds.batch["query_seq"], ds.batch["query_len"],
                                     synth.records_aos(ds.batch, O.RECORD_DTYPE), exclude_factor=x, reeval_bandwidth=t)
      try:
          rows, stats = model.predict(ds.batch)
          ok = rc == 0 and rows.tobytes() == exp.tobytes() and stats.tobytes() == est.tobytes()
          what = "ok" if ok else "MISMATCH (oracle rc %d)" % rc
      except pkg.RpaError as e:  # a batch the reference would abort on: both sides must say so (oracle 1/2/3 = engine 3/4/5)
          ok = rc != 0 and e.code == rc + 2
          aborts += ok
          rows = None
          what = "both abort (oracle rc %d, engine %d)" % (rc, e.code) if ok else "MISMATCH: engine error %d %s, oracle rc %d" % (e.code, e, rc)
      total_sets += n_sets
      if verbose or not ok:
        print("seed %d: %s sets=%d hits=%d seg=%d b=%d depth=%d%s x=%g t=%g env=%s -> %s  (%.0f s)" % (first_seed + k, "protein" if alphabet else "dna", n_sets, n_hits, seg,
                                                                           branching, int(ds.tax.depth.max()), " windows" if windowed else "", x, t, env, what, time.time() - t0), flush=True)
      if not ok:
          for f in (rows.dtype.names if rows is not None else ()):
              bad = np.nonzero(rows[f] != exp[f])[0]
              if len(bad):
                  print("  field", f, "sets", bad[:5], rows[f][bad[:5]], exp[f][bad[:5]])
          for kk in KNOBS: os.environ.pop(kk, None)
          raise AssertionError("soak: seed %d differs from the oracle (%s)" % (first_seed + k, what))
  for kk in KNOBS:
      os.environ.pop(kk, None)
  return n_seeds, total_sets, aborts


if __name__ == "__main__":
    n, sets, aborts = run(int(sys.argv[1]) if len(sys.argv) > 1 else 40)
    print("soak ok: %d batches, %d record sets (%d batches on which engine and oracle both abort)" % (n, sets, aborts))
