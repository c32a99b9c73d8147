#!/usr/bin/env python3
"""Generate pair-level golden vectors by RUNNING THE REFERENCE's SeqAn (oracle/_ref/libseqan_ref.so, built in place
from /root/reference by `make -C oracle ref`).  Container-only; the outputs are committed:

  pairs_dna.json      [{a, b, dist, match, mismatch, gap, length}]          (getAlignmentDNA, :133-171)
  pairs_protein.json  [{a, b, selfscore, mutual, match, mismatch, gap, length, distance_bits, similarity_bits}]
                                                                           (getAlignmentProtein, :173-242)
Sequences are letters; floats are stored as IEEE-754 bit patterns (bit-exact comparison).
"""
import json, os, sys
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from helpers import oracle as O, refio as R

rng = np.random.default_rng(20261003)
DNA = np.frombuffer(b"ACGTN", np.uint8)
AA = np.frombuffer(R.AA_LETTERS.encode(), np.uint8)


def mutate(a, k, rate, indels):
    b = a.copy()
    m = rng.random(len(b)) < rate
    b[m] = rng.integers(0, k, int(m.sum()))
    for _ in range(indels):
        if len(b) > 2 and rng.random() < 0.5:
            b = np.delete(b, rng.integers(0, len(b)))
        else:
            b = np.insert(b, rng.integers(0, len(b) + 1), rng.integers(0, k))
    return b.astype(np.uint8)


def dna_pairs():
    out = []
    edge = [1, 2, 31, 32, 33, 63, 64, 65, 95, 96, 97, 127, 128, 129, 191, 192, 193, 200, 255, 256, 257, 300]
    for la in edge:
        for mode in range(4):
            a = rng.integers(0, 4, la).astype(np.uint8)
            if mode == 0: b = a.copy()
            elif mode == 1: b = mutate(a, 4, 0.1, 3)
            elif mode == 2: b = rng.integers(0, 5, max(1, la + int(rng.integers(-20, 20)))).astype(np.uint8)
            else:
                a[rng.random(la) < 0.05] = 4
                b = mutate(a, 5, 0.2, 6)
            out.append((a, b))
    for _ in range(200):
        la = int(rng.integers(150, 260))
        a = rng.integers(0, 4, la).astype(np.uint8)
        a[rng.random(la) < 0.01] = 4
        out.append((a, mutate(a, 4, float(rng.uniform(0, 0.35)), int(rng.integers(0, 8)))))
    for la in (1000, 2500, 10000):  # long-contig class
        a = rng.integers(0, 4, la).astype(np.uint8)
        out.append((a, mutate(a, 4, 0.08, 30)))
    res = []
    for a, b in out:
        r = O.ref_align_dna(a, b)
        res.append(dict(a=DNA[a].tobytes().decode(), b=DNA[b].tobytes().decode(), dist=int(r[0]), match=int(r[1]),
                        mismatch=int(r[2]), gap=int(r[3]), length=int(r[4])))
    return res


def protein_pairs():
    out = []
    for la in [1, 2, 5, 23, 24, 25, 31, 32, 33, 64, 67, 100, 128, 200, 255, 300]:
        for k in (20, 27):
            a = rng.integers(0, k, la).astype(np.uint8)
            out.append((a, a.copy()))
            out.append((a, mutate(a, k, 0.25, 4)))
            out.append((a, rng.integers(0, k, max(1, la + int(rng.integers(-10, 10)))).astype(np.uint8)))
    for _ in range(150):
        la = int(rng.integers(40, 110))
        a = rng.integers(0, 20, la).astype(np.uint8)
        out.append((a, mutate(a, 20, float(rng.uniform(0, 0.5)), int(rng.integers(0, 6)))))
    res = []
    for a, b in out:
        o, f = O.ref_align_protein(a, b)
        res.append(dict(a=AA[a].tobytes().decode(), b=AA[b].tobytes().decode(), selfscore=int(o[0]), mutual=int(o[1]),
                        match=int(o[2]), mismatch=int(o[3]), gap=int(o[4]), length=int(o[5]),
                        distance_bits=int(f[0:1].view(np.uint32)[0]), similarity_bits=int(f[1:2].view(np.uint32)[0])))
    return res


if __name__ == "__main__":
    assert O.ref_available(), "build oracle/_ref first (make -C oracle ref)"
    d = dna_pairs(); p = protein_pairs()
    json.dump(d, open(os.path.join(HERE, "pairs_dna.json"), "w"), separators=(",", ":"))
    json.dump(p, open(os.path.join(HERE, "pairs_protein.json"), "w"), separators=(",", ":"))
    print(len(d), "dna pairs;", len(p), "protein pairs")
