"""Pins the C oracle's pair level against the reference's own SeqAn.

1. committed golden vectors produced by running the reference's SeqAn (tests/golden/make_pair_vectors.py);
2. live comparison with oracle/_ref/libseqan_ref.so when that binary is present (it is built from
   /root/reference in the build container and travels as a binary; skipped otherwise);
3. the two restated formulations agree (SeqAn-style trace matrix + traceback vs forward-carried counters; SeqAn's
   32-bit blocked Myers vs textbook Levenshtein) -- the device kernels use the second of each pair.
"""
import json
import os

import numpy as np
import pytest

from helpers import refio as R


def _enc(s, alphabet):
    return R.encode(s.encode(), alphabet)


def test_golden_dna_pairs(oracle, golden_dir):
    vec = json.load(open(os.path.join(golden_dir, "pairs_dna.json")))
    assert len(vec) > 250
    for v in vec:
        a, b = _enc(v["a"], 0), _enc(v["b"], 0)
        o = oracle.align_dna(a, b)
        assert (int(o["distance"]), o["matches"], o["mismatches"], o["gaps"], o["length"]) == (
            v["dist"], v["match"], v["mismatch"], v["gap"], v["length"])
        assert oracle.levenshtein(a, b) == v["dist"]
        assert oracle.myers_score(b, a) == -v["dist"]  # symmetric in its arguments


def test_golden_protein_pairs(oracle, golden_dir):
    vec = json.load(open(os.path.join(golden_dir, "pairs_protein.json")))
    assert len(vec) > 200
    for v in vec:
        a, b = _enc(v["a"], 1), _enc(v["b"], 1)
        for forward in (False, True):
            rc, o = oracle.align_protein(a, b, forward=forward)
            # rc 3 = a live assert of the reference build would fire (selfscore < mutual or distance < 0, :205/:240;
            # happens for X/*-rich strings whose self score is negative); the numbers are still SeqAn's
            assert rc == (3 if (not forward and (o["selfscore"] < o["mutualscore"] or not o["distance"] >= 0)) else 0)
            assert (o["selfscore"], o["mutualscore"], o["matches"], o["mismatches"], o["gaps"], o["length"]) == (
                v["selfscore"], v["mutual"], v["match"], v["mismatch"], v["gap"], v["length"])
            assert int(np.float32(o["distance"]).view(np.uint32)) == v["distance_bits"]
            assert int(np.float32(o["similarity"]).view(np.uint32)) == v["similarity_bits"]


def test_alphabet_tables(oracle):
    # seqan/basic/alphabet_residue_tabs.h:107-140, 397-430 as restated in helpers.refio vs the generated C tables
    assert list(R.encode(b"ACGTUNacgtun-RYX*", 0)) == [0, 1, 2, 3, 3, 4, 0, 1, 2, 3, 3, 4, 4, 4, 4, 4, 4]
    assert list(R.encode(b"ABCDEFGHIJKLMNOPQRSTUVWYZX*", 1)) == list(range(27))
    assert list(R.encode(b"x-1", 1)) == [25, 25, 25]


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_live_vs_reference_seqan(oracle, seed):
    if not oracle.ref_available():
        pytest.skip("oracle/_ref not built (needs /root/reference)")
    rng = np.random.default_rng(seed)
    for t in range(400):
        la, lb = int(rng.integers(1, 280)), int(rng.integers(1, 280))
        a = rng.integers(0, 5, la).astype(np.uint8)
        if t % 2:
            b = a.copy()
            m = rng.random(la) < 0.15
            b[m] = rng.integers(0, 5, int(m.sum()))
            b = np.delete(b, rng.integers(0, la, min(3, la - 1))) if la > 4 else b
        else:
            b = rng.integers(0, 5, lb).astype(np.uint8)
        r = oracle.ref_align_dna(a, b)
        o = oracle.align_dna(a, b)
        assert list(r) == [int(o["distance"]), o["matches"], o["mismatches"], o["gaps"], o["length"]]
    for t in range(200):
        k = (4, 20, 27)[t % 3]
        la, lb = int(rng.integers(1, 100)), int(rng.integers(1, 100))
        a = rng.integers(0, k, la).astype(np.uint8)
        b = rng.integers(0, k, lb).astype(np.uint8)
        if t % 2:
            b = a.copy()
            m = rng.random(la) < 0.3
            b[m] = rng.integers(0, k, int(m.sum()))
        ro, rf = oracle.ref_align_protein(a, b)
        rc, o = oracle.align_protein(a, b)
        assert list(ro) == [o["selfscore"], o["mutualscore"], o["matches"], o["mismatches"], o["gaps"], o["length"]]
        assert rf[0].tobytes() == np.float32(o["distance"]).tobytes()
        assert rf[1].tobytes() == np.float32(o["similarity"]).tobytes()
        s = a.copy()
        if k == 4:
            oracle.lib().orc_revcomp_dna5(s.ctypes.data, len(s))
            s2 = a.copy()
            oracle.ref().ref_revcomp_dna5(s2.ctypes.data, len(s2))
            assert (s == s2).all()


def test_myers_edge_lengths(oracle):
    rng = np.random.default_rng(7)
    for la in (1, 31, 32, 33, 64, 65, 96, 128, 160, 192, 224, 256):
        for lb in (1, la, la + 1, max(1, la - 1), 2 * la):
            a = rng.integers(0, 5, la).astype(np.uint8)
            b = rng.integers(0, 5, lb).astype(np.uint8)
            assert -oracle.myers_score(a, b) == oracle.levenshtein(a, b)
    e = np.zeros(0, np.uint8)
    assert oracle.myers_score(e, np.array([1, 2, 3], np.uint8)) == -3  # reference is UB here; documented choice
