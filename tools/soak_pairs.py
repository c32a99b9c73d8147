#!/usr/bin/env python3
"""tools/soak_pairs.py [N_BATCHES] -- pair-level differential soak on the GPU box: batches of random nucleotide / protein
pairs of random lengths (every kernel: one-strip Myers classes, register strips, wave-per-pair, both protein kernels) and
similarity (identical, point mutations + indels, unrelated, reversed, N-rich) through rpa_align_pairs and the CPU oracle.
Test infrastructure (uses oracle/)."""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as graft
from helpers import oracle as O
from helpers import refio as R

LET = {0: np.frombuffer(b"ACGTN", np.uint8), 1: np.frombuffer(R.AA_LETTERS.encode(), np.uint8)}


class OneNodeTax:
    parent = np.zeros(1, np.uint32); left = np.array([1], np.uint32); right = np.array([2], np.uint32); depth = np.zeros(1, np.uint8)


def mutate(a, rng, asize, rate):
    b = a.copy()
    mm = rng.random(len(b)) < rate
    b[mm] = rng.integers(0, asize, int(mm.sum()))
    for _ in range(int(rng.integers(0, 1 + len(a) // 16))):
        if len(b) > 1 and rng.random() < 0.5:
            b = np.delete(b, rng.integers(0, len(b)))
        else:
            b = np.insert(b, rng.integers(0, len(b) + 1), rng.integers(0, asize))
    return b.astype(np.uint8)


KNOBS = ("RPA_LONG_KERNEL", "RPA_BAND_MARGIN", "RPA_BAND_SHORT_MIN", "RPA_BAND_WORDS", "RPA_BAND4_PHASES", "RPA_BAND_DUO", "RPA_BAND_WINDOW_WIDE")


def run(n_batches=20, rng_seed=777, verbose=True):
    """Returns (batches, pairs); raises AssertionError at the first pair that differs from the oracle."""
    pkg = graft.load_package()
    rng = np.random.default_rng(rng_seed)
    t0 = time.time()
    total = 0
    for k in range(n_batches):
        alphabet = int(rng.random() < 0.4)
        asize = 5 if alphabet == 0 else 27
        if alphabet == 0:
            top = int(rng.choice([64, 256, 300, 400, 700, 1100, 2000, 5000, 18000]))
            n = int(rng.integers(200, 3000)) if top <= 1100 else int(rng.integers(20, 200)) if top <= 5000 else int(rng.integers(4, 20))
            for kk in KNOBS: os.environ.pop(kk, None)
            lk = rng.choice(["", "wave", "lane"])
            if lk: os.environ["RPA_LONG_KERNEL"] = str(lk)
            if rng.random() < 0.6:  # the exact bands, forced onto launches of any size, with margins from hopeless to generous
                os.environ["RPA_BAND_SHORT_MIN"] = "1"
                os.environ["RPA_BAND_MARGIN"] = str(int(rng.choice([1, 8, 16, 21, 40])))
                if rng.random() < 0.3: os.environ["RPA_BAND_WORDS"] = "5"
                elif rng.random() < 0.3: os.environ["RPA_BAND4_PHASES"] = "2"
                if rng.random() < 0.3: os.environ["RPA_BAND_DUO"] = "0"
                if rng.random() < 0.3: os.environ["RPA_BAND_WINDOW_WIDE"] = "1"  # (round 5: K1r's wide instances)
        else:
            top = int(rng.choice([30, 80, 260, 330, 520, 800]))
            n = int(rng.integers(100, 1500)) if top <= 330 else int(rng.integers(20, 200))
            lk = ""
        a_codes, b_codes = [], []
        for _ in range(n):
            la = int(rng.integers(1, top + 1))
            a = rng.integers(0, asize if rng.random() < 0.3 else (4 if alphabet == 0 else 20), la).astype(np.uint8)
            mode = rng.integers(0, 5)
            if mode == 0: b = a.copy()
            elif mode == 1: b = mutate(a, rng, asize, float(rng.choice([0.02, 0.1, 0.3])))
            elif mode == 2: b = rng.integers(0, asize, int(rng.integers(1, top + 1))).astype(np.uint8)
            elif mode == 3: b = a[::-1].copy()
            else: b = mutate(a[: max(1, la // 2)], rng, asize, 0.05)
            a_codes.append(a); b_codes.append(b)
        seqs = [LET[alphabet][x] for pair in zip(a_codes, b_codes) for x in pair]
        offsets = np.zeros(len(seqs) + 1, np.uint64); offsets[1:] = np.cumsum([len(s) for s in seqs])
        letters = np.concatenate(seqs)
        m = pkg.RPAPredictionModel(OneNodeTax(), alphabet=alphabet)
        out = m.align_pairs(letters, offsets, np.arange(n) * 2, np.arange(n) * 2 + 1)
        bad = -1
        for i in range(n):
            if alphabet == 0:
                o = O.align_dna(a_codes[i], b_codes[i]); rc = 0
            else:
                rc, o = O.align_protein(a_codes[i], b_codes[i])
            g = out[i]
            same = (g["matches"], g["mismatches"], g["gaps"], g["length"]) == (o["matches"], o["mismatches"], o["gaps"], o["length"]) \
                and np.float32(g["similarity"]).tobytes() == np.float32(o["similarity"]).tobytes() \
                and (rc != 0 or np.float32(g["distance"]).tobytes() == np.float32(o["distance"]).tobytes()) \
                and (rc == 0 or np.isnan(g["distance"]))
            if not same:
                bad = i
                break
        total += n
        if verbose or bad >= 0:
          print("batch %d: %s pairs=%d max_len=%d kernel=%s -> %s  (%.0f s)" % (k, "protein" if alphabet else "dna", n, top, lk or "auto",
                                                                          "ok" if bad < 0 else "MISMATCH at pair %d (%d x %d)" % (bad, len(a_codes[bad]), len(b_codes[bad])),
                                                                          time.time() - t0), flush=True)
        if bad >= 0:
            print("  engine", out[bad], "\n  oracle", o)
            for kk in KNOBS: os.environ.pop(kk, None)
            raise AssertionError("pair soak: batch %d pair %d differs from the oracle" % (k, bad))
    for kk in KNOBS:
        os.environ.pop(kk, None)
    return n_batches, total


if __name__ == "__main__":
    nb, tot = run(int(sys.argv[1]) if len(sys.argv) > 1 else 20)
    print("pair soak ok: %d batches, %d pairs" % (nb, tot))
