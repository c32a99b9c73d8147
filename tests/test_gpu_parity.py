"""Parity tests proper: the HIP path (through the C ABI) against the CPU oracle and the committed golden fixtures.
Run on the MI355X box: python -m pytest tests -m gpu.  Bit-exact bar: integer scores, node indices, supports,
feature coordinates, alignment counters AND the float32 bit patterns of distances / ival."""
import gzip
import importlib
import json
import os
import shutil

import numpy as np
import pytest

import __graft_entry__ as graft
from helpers import refio as R

pytestmark = pytest.mark.gpu

LETTERS_DNA = np.frombuffer(b"ACGTN", np.uint8)
LETTERS_AA = np.frombuffer(R.AA_LETTERS.encode(), np.uint8)


@pytest.fixture(scope="module")
def pkg():
    p = graft.load_package()
    p.load_library()  # must exist: no fallback
    return p


@pytest.fixture(scope="module")
def synth(pkg):
    return importlib.import_module("taxator_tk_amd.synth")


class OneNodeTax:
    parent = np.zeros(1, np.uint32)
    left = np.array([1], np.uint32)
    right = np.array([2], np.uint32)
    depth = np.zeros(1, np.uint8)


def _pack(seqs):
    offsets = np.zeros(len(seqs) + 1, np.uint64)
    offsets[1:] = np.cumsum([len(s) for s in seqs])
    return np.concatenate(seqs) if seqs else np.zeros(0, np.uint8), offsets


def _check_dna(oracle, out, a_codes, b_codes):
    for k in range(len(out)):
        o = oracle.align_dna(a_codes[k], b_codes[k])
        got = out[k]
        assert (got["matches"], got["mismatches"], got["gaps"], got["length"]) == (o["matches"], o["mismatches"], o["gaps"], o["length"]), k
        assert np.float32(got["distance"]).tobytes() == np.float32(o["distance"]).tobytes(), k
        assert np.float32(got["similarity"]).tobytes() == np.float32(o["similarity"]).tobytes(), k


@pytest.mark.parametrize("short_kernel", ["word-per-lane", "pair-per-lane"])
def test_dna_pairs_golden(pkg, oracle, golden_dir, short_kernel, monkeypatch):
    """The reference's own SeqAn vectors through both one-strip kernels for launches of few pairs: K1w (a pattern word per lane,
    eight lanes per pair: the default below 16 384 pairs) and k_myers_short (a pair per lane)."""
    monkeypatch.setenv("RPA_SHORT_WAVE_MAX", "1000000" if short_kernel == "word-per-lane" else "0")
    vec = json.load(open(os.path.join(golden_dir, "pairs_dna.json")))
    m = pkg.RPAPredictionModel(OneNodeTax(), alphabet=pkg.DNA)
    seqs, pa, pb = [], [], []
    for v in vec:
        seqs += [np.frombuffer(v["a"].encode(), np.uint8), np.frombuffer(v["b"].encode(), np.uint8)]
        pa.append(len(seqs) - 2)
        pb.append(len(seqs) - 1)
    letters, offsets = _pack(seqs)
    out = m.align_pairs(letters, offsets, pa, pb)
    for v, got in zip(vec, out):
        assert (int(got["distance"]), got["matches"], got["mismatches"], got["gaps"], got["length"]) == (
            v["dist"], v["match"], v["mismatch"], v["gap"], v["length"]), (len(v["a"]), len(v["b"]))


@pytest.mark.parametrize("long_kernel", ["wave", "lane", None, "short-band", "short-pair-per-lane"])
def test_dna_pairs_random_all_bins(pkg, oracle, long_kernel, monkeypatch):
    """Every length bin of the nucleotide kernels against the oracle.  Patterns above 384 nt have two kernels (one pair
    per wavefront for few pairs, register strips per lane for many, chosen by a cost model): both are forced in turn;
    above 16384 nt only the strip kernel applies."""
    monkeypatch.delenv("RPA_LONG_KERNEL", raising=False)
    monkeypatch.delenv("RPA_SHORT_WAVE_MAX", raising=False)
    if long_kernel == "short-band": monkeypatch.setenv("RPA_BAND_SHORT_MIN", "1")  # banded one-strip kernel for launches of any size
    elif long_kernel == "short-pair-per-lane": monkeypatch.setenv("RPA_SHORT_WAVE_MAX", "0")  # (the default for these few pairs is K1w, a pattern word per lane)
    elif long_kernel: monkeypatch.setenv("RPA_LONG_KERNEL", long_kernel)
    rng = np.random.default_rng(42)
    m = pkg.RPAPredictionModel(OneNodeTax(), alphabet=pkg.DNA)
    a_codes, b_codes = [], []
    lens = list(range(1, 70)) + [95, 96, 97, 127, 128, 129, 159, 160, 161, 191, 192, 193, 223, 224, 225, 255, 256] * 3
    lens += [257, 300, 511, 512, 513, 1000, 2047, 2048, 2049, 3000, 4097, 6000, 8192, 10000, 12289, 16384]
    lens += [288, 289, 320, 321, 384, 385, 448, 449, 544, 545, 576, 640, 641, 672, 768, 769, 800, 896, 1024, 1025, 1056, 1281]
    lens += [16385, 20000, 33001]
    for la in lens:
        for mode in range(3):
            a = rng.integers(0, 5 if mode else 4, la).astype(np.uint8)
            if mode == 0:
                b = a.copy()
                mm = rng.random(la) < 0.1
                b[mm] = rng.integers(0, 4, int(mm.sum()))
                for _ in range(min(4, la // 8)):
                    b = np.delete(b, rng.integers(0, len(b))) if rng.random() < 0.5 else np.insert(b, rng.integers(0, len(b)), rng.integers(0, 4))
            elif mode == 1:
                b = rng.integers(0, 5, max(1, int(la * rng.uniform(0.5, 1.5)))).astype(np.uint8)
            else:
                b = a[::-1].copy()
            a_codes.append(a)
            b_codes.append(b.astype(np.uint8))
    seqs = [LETTERS_DNA[x] for pair in zip(a_codes, b_codes) for x in pair]
    letters, offsets = _pack(seqs)
    n = len(a_codes)
    out = m.align_pairs(letters, offsets, np.arange(n) * 2, np.arange(n) * 2 + 1)
    _check_dna(oracle, out, a_codes, b_codes)
    # argument order must not matter for DNA (getAlignmentDNA orders by length itself)
    out2 = m.align_pairs(letters, offsets, np.arange(n) * 2 + 1, np.arange(n) * 2)
    assert out.tobytes() == out2.tobytes()


@pytest.mark.parametrize("band", ["forced", "forced-retries-pair-per-lane", "four-words", "five-words", "off", "wide-ring-window"])
def test_dna_short_pairs_band(pkg, oracle, band, monkeypatch):
    """Patterns of 5..12 words (129..384 nt) go through a one-strip kernel that only updates the pattern words within two
    words of the diagonal; pairs whose distance lies outside that band (or whose lengths differ by 64 and more) are
    re-run by the full kernel.  Forced on for launches of any size / switched off: the same bytes either way, and both
    equal to the oracle -- from identical pairs to unrelated ones, with N's, and every length difference class.
    Patterns of 13..64 words (385..2048 nt) take the ring-window kernel (K1r: 6 / 10 words around the diagonal, only the window
    alive in registers) in front of the strip kernels: the same lists of pairs, around its limits of 160 / 288 edits."""
    monkeypatch.setenv("RPA_BAND_SHORT_MIN", "1")
    monkeypatch.delenv("RPA_SHORT_WAVE_MAX", raising=False)
    if band == "forced-retries-pair-per-lane": monkeypatch.setenv("RPA_SHORT_WAVE_MAX", "0")  # (a short retry list is K1w's by default)
    if band == "off": monkeypatch.setenv("RPA_BAND_SHORT", "0")
    if band == "five-words": monkeypatch.setenv("RPA_BAND_WORDS", "5")  # the in-phase window (k = 128 - |delta|)
    if band == "four-words": monkeypatch.setenv("RPA_BAND4_PHASES", "2")  # four words all the way, out of phase (96); default: a three-word middle (80)
    if band == "wide-ring-window": monkeypatch.setenv("RPA_BAND_WINDOW_WIDE", "1")  # K1r's wide instances (10 / 14 words: 288 / 416 edits) instead of 6 / 10
    rng = np.random.default_rng(12)
    m = pkg.RPAPredictionModel(OneNodeTax(), alphabet=pkg.DNA)
    a_codes, b_codes = [], []
    for la in [129, 150, 159, 160, 161, 180, 191, 192, 193, 200, 210, 223, 224, 225, 240, 255, 256,
               257, 270, 288, 289, 300, 319, 320, 321, 340, 352, 353, 370, 383, 384,   # (9 ... 12 words: one-strip classes since round 5)
               385, 400, 415, 416, 417, 450, 480, 511, 512, 513, 530, 544, 545, 600, 700, 767, 768, 769, 900, 1000, 1023, 1024,
               1025, 1300, 1600, 2047, 2048, 2049]:  # (13 ... 64 words: the ring window, K1r)
        for rate in [0.0, 0.02, 0.1, 0.25, 0.32, 0.4, 0.6, 1.0]:
            for extra in [0, 3, 31, 40, 63, 64, 65, 100] + ([79, 80, 81, 143, 144, 145, 200, 207, 208, 209] if la > 384 else []):  # (K1r: length differences around 80, 144 and 208)
                a = rng.integers(0, 5 if rate > 0.3 else 4, la).astype(np.uint8)
                r = rng.random(la)
                b = a.copy()
                sub = r < rate * 0.7
                b[sub] = rng.integers(0, 4, int(sub.sum()))
                b = b[~((r >= rate * 0.7) & (r < rate * 0.85))]
                ins = np.nonzero((r >= rate * 0.85) & (r < rate))[0]
                b = np.insert(b, np.minimum(ins, len(b)), rng.integers(0, 4, len(ins)).astype(np.uint8)) if len(b) else a[:1].copy()
                if extra:
                    tail = rng.integers(0, 4, extra).astype(np.uint8)
                    b = np.concatenate([tail, b]) if rng.random() < 0.5 else np.concatenate([b, tail])
                if rng.random() < 0.5: a, b = b, a
                a_codes.append(a.astype(np.uint8))
                b_codes.append(b.astype(np.uint8))
    seqs = [LETTERS_DNA[x] for pair in zip(a_codes, b_codes) for x in pair]
    letters, offsets = _pack(seqs)
    n = len(a_codes)
    out = m.align_pairs(letters, offsets, np.arange(n) * 2, np.arange(n) * 2 + 1)
    _check_dna(oracle, out, a_codes, b_codes)


@pytest.mark.parametrize("band", ["default", "tight", "off", "one-lane"])
def test_dna_long_pairs_band(pkg, oracle, band, monkeypatch):
    """The strip kernel only computes the cells inside Ukkonen's band around a probed distance estimate; a pair whose
    distance lies outside its band is re-run with the band wide open (one pair per wavefront up to 16384 nt, the strip
    kernel beyond).  Results must not depend on the band: default margin, a margin so tight that nearly every pair is
    retried, and no band at all, on pairs from identical to unrelated, with length differences, up to 20 kbp."""
    monkeypatch.setenv("RPA_LONG_KERNEL", "lane")
    if band == "tight": monkeypatch.setenv("RPA_BAND_MARGIN", "1")
    if band == "off": monkeypatch.setenv("RPA_BAND", "0")
    if band == "one-lane": monkeypatch.setenv("RPA_BAND_DUO", "0")  # the strip kernel with one lane per pair (default: two)
    rng = np.random.default_rng(11)
    m = pkg.RPAPredictionModel(OneNodeTax(), alphabet=pkg.DNA)
    a_codes, b_codes = [], []
    for la in [257, 300, 511, 777, 1024, 1500, 2049, 3000, 5000, 8192, 10000, 16384, 17000, 20000]:
        for rate in [0.0, 0.01, 0.05, 0.15, 0.3, 0.6]:
            a = rng.integers(0, 4, la).astype(np.uint8)
            r = rng.random(la)
            b = a.copy()
            sub = r < rate * 0.6
            b[sub] = rng.integers(0, 5, int(sub.sum()))
            keep = ~((r >= rate * 0.6) & (r < rate * 0.8))            # deletions
            ins = (r >= rate * 0.8) & (r < rate)                      # insertions (a random base behind the position)
            counts = keep.astype(np.int64) + ins.astype(np.int64)
            out = np.repeat(b, counts)
            second = np.cumsum(counts)[ins & keep] - 1
            out[second] = rng.integers(0, 4, len(second))
            if rate == 0.05:   # a divergent stretch in the middle of an otherwise close pair: the probe underestimates
                lo = len(out) // 2
                out[lo:lo + len(out) // 8] = rng.integers(0, 4, len(out[lo:lo + len(out) // 8]))
            if rate == 0.15:   # plus a length difference
                out = np.concatenate([out, rng.integers(0, 4, la // 10).astype(np.uint8)])
            if len(out) == 0: out = a[:1].copy()
            a_codes.append(a)
            b_codes.append(out.astype(np.uint8))
    seqs = [LETTERS_DNA[x] for pair in zip(a_codes, b_codes) for x in pair]
    letters, offsets = _pack(seqs)
    n = len(a_codes)
    out = m.align_pairs(letters, offsets, np.arange(n) * 2, np.arange(n) * 2 + 1)
    _check_dna(oracle, out, a_codes, b_codes)


def test_dna_pairs_many_200bp(pkg, oracle):
    """The benchmark shape: 200 bp x 200 bp, ~10 % divergence, checked against the oracle on a sample and through
    a size-independent property on all of them: d(a,b) == d(b,a) and |len diff| <= d <= max len."""
    rng = np.random.default_rng(7)
    m = pkg.RPAPredictionModel(OneNodeTax(), alphabet=pkg.DNA)
    n = 20000
    base = rng.integers(0, 4, (n, 200)).astype(np.uint8)
    mut = base.copy()
    mm = rng.random((n, 200)) < rng.uniform(0, 0.3, (n, 1))
    mut[mm] = rng.integers(0, 4, int(mm.sum()))
    cut = rng.integers(190, 201, n)
    seqs = []
    for k in range(n):
        seqs.append(LETTERS_DNA[base[k]])
        seqs.append(LETTERS_DNA[mut[k, :cut[k]]])
    letters, offsets = _pack(seqs)
    out = m.align_pairs(letters, offsets, np.arange(n) * 2, np.arange(n) * 2 + 1)
    rev = m.align_pairs(letters, offsets, np.arange(n) * 2 + 1, np.arange(n) * 2)
    assert out.tobytes() == rev.tobytes()
    d = out["distance"]
    assert ((d >= 200 - cut) & (d <= 200)).all()
    assert (out["matches"] + out["mismatches"] + out["gaps"] == out["length"]).all()
    idx = rng.choice(n, 600, replace=False)
    _check_dna(oracle, out[idx], [base[k] for k in idx], [mut[k, :cut[k]] for k in idx])


@pytest.mark.parametrize("lanes", ["lanes-per-pair", "one-lane"])
def test_protein_pairs_golden_and_random(pkg, oracle, golden_dir, lanes, monkeypatch):
    # default: packed pairs whose vertical sequence has <= 96 residues run with their strips in adjacent lanes (K3l);
    # RPA_PROTEIN_LANES=0 sends them all to the one-lane-per-pair kernel (K3p)
    if lanes == "one-lane":
        monkeypatch.setenv("RPA_PROTEIN_LANES", "0")
    vec = json.load(open(os.path.join(golden_dir, "pairs_protein.json")))
    m = pkg.RPAPredictionModel(OneNodeTax(), alphabet=pkg.PROTEIN)
    seqs = []
    for v in vec:
        seqs += [np.frombuffer(v["a"].encode(), np.uint8), np.frombuffer(v["b"].encode(), np.uint8)]
    letters, offsets = _pack(seqs)
    n = len(vec)
    out = m.align_pairs(letters, offsets, np.arange(n) * 2, np.arange(n) * 2 + 1)
    for v, got in zip(vec, out):
        assert (got["matches"], got["mismatches"], got["gaps"], got["length"]) == (v["match"], v["mismatch"], v["gap"], v["length"])
        d_ref = np.array([v["distance_bits"]], np.uint32).view(np.float32)[0]
        if v["selfscore"] >= v["mutual"] and d_ref >= 0:
            assert int(np.float32(got["distance"]).view(np.uint32)) == v["distance_bits"]
        else:
            assert np.isnan(got["distance"])  # a live assert of the reference build: flagged, raised only if consumed
        assert int(np.float32(got["similarity"]).view(np.uint32)) == v["similarity_bits"]
    rng = np.random.default_rng(3)
    a_codes, b_codes = [], []
    for la in [1, 2, 3, 31, 32, 33, 63, 64, 65, 66, 67, 68, 96, 97, 130, 200, 333]:
        for mode in range(3):
            a = rng.integers(0, 20, la).astype(np.uint8)
            if mode == 0:
                b = a.copy()
                mm = rng.random(la) < 0.3
                b[mm] = rng.integers(0, 20, int(mm.sum()))
                if la > 8:
                    b = np.delete(b, rng.integers(0, la, 3))
            elif mode == 1:
                b = rng.integers(0, 20, max(1, int(la * rng.uniform(0.6, 1.4)))).astype(np.uint8)
            else:
                b = a.copy()
            a_codes.append(a)
            b_codes.append(b.astype(np.uint8))
    # the packed-cell kernel takes pairs whose score and counter fields cannot overflow (312 x 312, 511 x 100, ...), the
    # two-register one the rest: both sides of the switch, the extreme scores the packed score field must hold (W/W =
    # +11 per residue, nothing in common = all gaps / mismatches), and lopsided pairs
    w_code = int(np.nonzero(LETTERS_AA == ord("W"))[0][0])
    c_code, g_code = (int(np.nonzero(LETTERS_AA == ord(ch))[0][0]) for ch in "CG")
    for la, lb in [(255, 255), (256, 256), (255, 256), (256, 255), (254, 255), (255, 1), (1, 255), (256, 1), (1, 256), (255, 24), (24, 255),
                   (255, 25), (72, 72), (48, 49), (312, 312), (313, 313), (313, 312), (511, 100), (100, 511), (512, 100), (511, 101), (340, 280)]:
        a_codes.append(np.full(la, w_code, np.uint8)); b_codes.append(np.full(lb, w_code, np.uint8))
        a_codes.append(np.full(la, w_code, np.uint8)); b_codes.append(np.full(lb, c_code, np.uint8))
        a_codes.append(np.full(la, g_code, np.uint8)); b_codes.append(np.full(lb, c_code, np.uint8))
        a_codes.append(rng.integers(0, 20, la).astype(np.uint8)); b_codes.append(rng.integers(0, 20, lb).astype(np.uint8))
        a = rng.integers(0, 20, la).astype(np.uint8)
        b = np.resize(a, lb).copy()
        mm = rng.random(lb) < 0.15
        b[mm] = rng.integers(0, 20, int(mm.sum()))
        a_codes.append(a); b_codes.append(b)
    # every class of the lanes-per-pair kernel (1 .. 13 strips of 24 rows) and its edges (columns go in groups of four;
    # lanes of one wavefront with different column counts), more pairs than one run of a wavefront holds
    for lb in [1, 2, 23, 24, 25, 47, 48, 49, 71, 72, 73, 95, 96, 97, 120, 121, 144, 145, 150, 168, 169, 192, 193, 216, 217, 240, 241, 250, 264,
               265, 288, 289, 311, 312]:
        for la in [1, 2, 3, 4, 5, 7, 8, 9, lb, lb + 1, 2 * lb + 3, 150]:
            a = rng.integers(0, 20, la).astype(np.uint8)
            b = np.resize(a, lb).copy()
            mm = rng.random(lb) < 0.25
            b[mm] = rng.integers(0, 20, int(mm.sum()))
            a_codes.append(a); b_codes.append(b)
    seqs = [LETTERS_AA[x] for pair in zip(a_codes, b_codes) for x in pair]
    letters, offsets = _pack(seqs)
    n = len(a_codes)
    out = m.align_pairs(letters, offsets, np.arange(n) * 2, np.arange(n) * 2 + 1)
    for k in range(n):
        rc, o = oracle.align_protein(a_codes[k], b_codes[k])
        got = out[k]
        assert (got["matches"], got["mismatches"], got["gaps"], got["length"]) == (o["matches"], o["mismatches"], o["gaps"], o["length"]), k
        if rc == 0:
            assert np.float32(got["distance"]).tobytes() == np.float32(o["distance"]).tobytes(), k
        assert np.float32(got["similarity"]).tobytes() == np.float32(o["similarity"]).tobytes(), k


def test_protein_pairs_beyond_the_packed_cell(pkg, oracle, monkeypatch):
    """Pairs the packed cell cannot hold (from 313 x 313 on) run in the placement path on K3p with the SCORE-AND-LENGTH cell:
    14 bits of score, 16 bits of diagonal steps, no match counter -- distance and similarity only need the score and the
    alignment's length.  RPA_PAIRS_WIDE=1 lets rpa_align_pairs take that kernel too (matches / mismatches then read
    0xFFFFFFFF): distance, similarity, gaps and length must be the oracle's bit for bit, at the field limits (W/W = +11
    per residue up to the largest score 14 bits hold, nothing in common = all gaps), lopsided pairs, and past the limit
    (K3 again)."""
    monkeypatch.setenv("RPA_PAIRS_WIDE", "1")
    rng = np.random.default_rng(8)
    w_code = int(np.nonzero(LETTERS_AA == ord("W"))[0][0])
    c_code, g_code = (int(np.nonzero(LETTERS_AA == ord(ch))[0][0]) for ch in "CG")
    a_codes, b_codes, wide = [], [], []
    for la, lb in [(313, 313), (312, 313), (511, 101), (512, 100), (400, 400), (511, 511), (512, 512), (800, 800), (1000, 999), (1257, 1257),
                   (1258, 1258), (1259, 1259), (1300, 1200), (3000, 300), (300, 3000), (5000, 100), (100, 5000), (16000, 20), (20, 16000), (700, 1)]:
        fits = la + lb + 11 * min(la, lb) + 32 <= 16383 and not (la <= 511 and lb <= 511 and la + lb + 11 * min(la, lb) + 32 <= 4095)
        for kind in range(5):
            if kind == 0: a, b = np.full(la, w_code, np.uint8), np.full(lb, w_code, np.uint8)
            elif kind == 1: a, b = np.full(la, w_code, np.uint8), np.full(lb, c_code, np.uint8)
            elif kind == 2: a, b = np.full(la, g_code, np.uint8), np.full(lb, c_code, np.uint8)
            elif kind == 3: a, b = rng.integers(0, 20, la).astype(np.uint8), rng.integers(0, 20, lb).astype(np.uint8)
            else:
                a = rng.integers(0, 20, la).astype(np.uint8)
                b = np.resize(a, lb).copy()
                mm = rng.random(lb) < 0.15
                b[mm] = rng.integers(0, 20, int(mm.sum()))
                if lb > 40:
                    b = np.delete(b, rng.integers(0, lb, 5))
                    b = np.insert(b, rng.integers(0, len(b), 5), rng.integers(0, 20, 5)).astype(np.uint8)
            a_codes.append(a); b_codes.append(b)
            wide.append(fits and len(b) == lb)
    m = pkg.RPAPredictionModel(OneNodeTax(), alphabet=pkg.PROTEIN)
    seqs = [LETTERS_AA[x] for pair in zip(a_codes, b_codes) for x in pair]
    letters, offsets = _pack(seqs)
    n = len(a_codes)
    out = m.align_pairs(letters, offsets, np.arange(n) * 2, np.arange(n) * 2 + 1)
    n_wide = 0
    for k in range(n):
        rc, o = oracle.align_protein(a_codes[k], b_codes[k])
        got = out[k]
        assert (got["gaps"], got["length"]) == (o["gaps"], o["length"]), (k, len(a_codes[k]), len(b_codes[k]))
        if got["matches"] == 0xFFFFFFFF:
            n_wide += 1
            assert got["mismatches"] == 0xFFFFFFFF
        else:
            assert (got["matches"], got["mismatches"]) == (o["matches"], o["mismatches"]), k
            assert not wide[k], (k, len(a_codes[k]), len(b_codes[k]))  # a pair the wide cell holds went elsewhere
        if rc == 0:
            assert np.float32(got["distance"]).tobytes() == np.float32(o["distance"]).tobytes(), k
        assert np.float32(got["similarity"]).tobytes() == np.float32(o["similarity"]).tobytes(), k
    assert n_wide >= sum(wide) > 40
    monkeypatch.setenv("RPA_PROTEIN_WIDE", "0")  # and with the wide cell off: the same bytes for distance / similarity
    m2 = pkg.RPAPredictionModel(OneNodeTax(), alphabet=pkg.PROTEIN)
    out2 = m2.align_pairs(letters, offsets, np.arange(n) * 2, np.arange(n) * 2 + 1)
    assert out2["distance"].tobytes() == out["distance"].tobytes() and out2["similarity"].tobytes() == out["similarity"].tobytes()
    assert (out2["length"] == out["length"]).all() and (out2["matches"] != 0xFFFFFFFF).all()


def _engine_vs_oracle(pkg, oracle, synth, ds, alphabet, speculation=None, exclude_factor=0.5, reeval_bandwidth=0.05):
    model = pkg.RPAPredictionModel(ds.tax, alphabet=alphabet, speculation=speculation, exclude_factor=exclude_factor,
                                   reeval_bandwidth=reeval_bandwidth)
    model.load_store(pkg.STORE_QUERY, ds.query_letters, ds.query_offsets)
    model.load_store(pkg.STORE_DB, ds.ref_letters, ds.ref_offsets)
    rows, stats = model.predict(ds.batch)
    tax = oracle.TaxArrays(ds.tax.parent, ds.tax.left, ds.tax.right, ds.tax.depth)
    q = oracle.StoreArrays(synth.encode(ds.query_letters, alphabet), ds.query_offsets)
    db = oracle.StoreArrays(synth.encode(ds.ref_letters, alphabet), ds.ref_offsets)
    rc, exp, est = oracle.predict_batch(tax, q, db, alphabet, ds.batch["set_offsets"], ds.batch["query_seq"],
                                        ds.batch["query_len"], synth.records_aos(ds.batch, oracle.RECORD_DTYPE),
                                        exclude_factor=exclude_factor, reeval_bandwidth=reeval_bandwidth)
    assert rc == 0
    for f in ("lower", "upper", "rtax", "support", "begin", "end", "flags"):
        bad = np.nonzero(rows[f] != exp[f])[0]
        assert len(bad) == 0, (f, bad[:5], rows[bad[:5]], exp[bad[:5]])
    assert rows["ival"].tobytes() == exp["ival"].tobytes()
    for f in ("n", "pass0", "pass1", "pass2"):
        assert (stats[f] == est[f]).all(), f
    return rows, stats


def test_predict_config1_plumbing(pkg, oracle, synth):
    """BASELINE config 1: 100 segments x 10 hits x 200 bp, b=2 mock taxonomy (128 nodes)."""
    ds = synth.make_dataset(n_sets=100, n_hits=10, seg_len=200, branching=2, genome_len=2000, seed=1)
    rows, stats = _engine_vs_oracle(pkg, oracle, synth, ds, 0)
    assert int(stats["pass0"].sum() + stats["pass1"].sum() + stats["pass2"].sum()) > 500


@pytest.mark.parametrize("seed", [2, 3])
def test_predict_dna_variety(pkg, oracle, synth, seed):
    ds = synth.make_dataset(n_sets=3000, n_hits=20, seg_len=200, branching=4, genome_len=3000, seed=seed, variety=True)
    rows, stats = _engine_vs_oracle(pkg, oracle, synth, ds, 0)
    assert (rows["flags"] & 1).any() and (rows["flags"] & 2).any() and (rows["flags"] & 4).any()


@pytest.mark.parametrize("speculation", [(0, 0), (1, 1), (3, 2), (7, 50)])
def test_speculation_chunks_never_change_results(pkg, oracle, synth, speculation):
    """The pass-1 / pass-2 anchor loops may be fed their alignments in chunks of any size (resumable loops): rows AND the
    demanded-alignment counters must stay those of the reference control flow."""
    ds = synth.make_dataset(n_sets=1500, n_hits=20, seg_len=150, branching=4, genome_len=3000, seed=21, variety=True)
    _engine_vs_oracle(pkg, oracle, synth, ds, 0, speculation=speculation)
    dp = synth.make_dataset(n_sets=300, n_hits=16, seg_len=60, branching=3, genome_len=1500, seed=22, alphabet=1,
                            level_rate=0.06, query_rate=0.08, variety=True)
    _engine_vs_oracle(pkg, oracle, synth, dp, 1, speculation=speculation)


@pytest.mark.parametrize("env", [
    {"RPA_SPEC1": "2", "RPA_SPEC2": "3", "RPA_SPEC_GROW": "1"},                 # paused loops double their chunk
    {"RPA_SPEC1": "3", "RPA_SPEC2": "2", "RPA_SPEC_ADAPTIVE": "1"},             # fixed start, per-round adaptation on
    {"RPA_SPARSE_DECIDE": "1", "RPA_SPARSE_EMIT": "1"},                         # every round after the first through the live list
    {"RPA_SPARSE_DECIDE": "1000000", "RPA_SPARSE_EMIT": "1000000"},             # never
    {"RPA_NO_COMBINED": "1", "RPA_SPARSE_DECIDE": "2", "RPA_SPARSE_EMIT": "50"},
    {"RPA_SPEC1": "2", "RPA_SPEC2": "3", "RPA_PASS1_FILTER": "0"},              # pass-1 chunks without the paused loop's thresholds
    {"RPA_SPEC1": "5", "RPA_SPEC2": "3", "RPA_PASS1_FILTER": "1"},              # ... with them, without the first-hit assumption
    {"RPA_PASS1_FILTER": "3"},                                                  # the first-hit assumption for short nucleotide segments too
], ids=["grow", "adaptive", "always-sparse", "never-sparse", "split-rounds", "unfiltered-chunks", "thresholds-only", "assume-always"])
def test_round_scheduling_knobs_never_change_results(pkg, oracle, synth, env, monkeypatch):
    """How the rounds are cut (chunk growth, when a round walks the live list instead of every row, pass 0 and pass 1 in
    one round or two) decides who computes what when -- never the rows or the demanded-alignment counters."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    ds = synth.make_dataset(n_sets=1200, n_hits=20, seg_len=150, branching=4, genome_len=3000, seed=23, variety=True)
    _engine_vs_oracle(pkg, oracle, synth, ds, 0)
    dp = synth.make_dataset(n_sets=300, n_hits=16, seg_len=60, branching=3, genome_len=1500, seed=24, alphabet=1,
                            level_rate=0.06, query_rate=0.08, variety=True)
    _engine_vs_oracle(pkg, oracle, synth, dp, 1)


def test_pass1_chunks_leave_out_what_the_loop_already_excludes(pkg, oracle, synth, monkeypatch):
    """A pass-1 loop that pauses asks for its next chunk without the hits that its score cut-off and similarity threshold
    at the pause already exclude (they only rise: level 1), and a chunk ends where the scores fall below the cut-off that
    its first hit would set (an assumption: level 2, the default): fewer alignments computed at every level, the same rows
    and demanded counters."""
    dp = synth.make_dataset(n_sets=2000, n_hits=20, seg_len=67, branching=4, genome_len=1500, seed=31, alphabet=1,
                            level_rate=0.06, query_rate=0.08)
    got = {}
    for flt, spec1 in (("0", "3"), ("1", "3"), ("2", "3"), ("2", "0")):
        monkeypatch.setenv("RPA_PASS1_FILTER", flt)
        monkeypatch.setenv("RPA_SPEC1", spec1); monkeypatch.setenv("RPA_SPEC2", "10")
        model = pkg.RPAPredictionModel(dp.tax, alphabet=pkg.PROTEIN)
        model.load_store(pkg.STORE_QUERY, dp.query_letters, dp.query_offsets)
        model.load_store(pkg.STORE_DB, dp.ref_letters, dp.ref_offsets)
        model.profile(enable=True, reset=True)
        rows, stats = model.predict(dp.batch)
        prof = model.profile()
        got[(flt, spec1)] = (prof["pairs"], prof["demanded"], rows.tobytes(), stats.tobytes())
    base = got[("0", "3")]
    assert all(g[1:] == base[1:] for g in got.values())
    assert got[("1", "3")][0] < 0.97 * base[0], (base[0], got[("1", "3")][0])
    assert got[("2", "3")][0] <= got[("1", "3")][0], (got[("1", "3")][0], got[("2", "3")][0])  # (three-hit chunks leave it little to cut)
    assert got[("2", "0")][0] < 1.15 * base[1], (got[("2", "0")][0], base[1])  # the whole loop in one request: < 1.15 computed per demanded
    monkeypatch.delenv("RPA_SPEC1"); monkeypatch.delenv("RPA_SPEC2"); monkeypatch.delenv("RPA_PASS1_FILTER")
    _engine_vs_oracle(pkg, oracle, synth, dp, 1)


@pytest.mark.parametrize("alphabet,step", [(1, 8.0), (1, 32.0), (0, 16.0)])
def test_pass1_chunk_end_on_tied_and_boundary_scores(pkg, oracle, synth, alphabet, step, monkeypatch):
    """The end of a pass-1 request is found by a binary search for the first hit whose score is below (float)(int)(X * the
    first hit's score): scores rounded to a coarse grid put runs of equal scores and hits exactly ON that cut-off into every
    record set (the cut-off test is >=: they stay in).  Level 3 = the assumption for nucleotide data too."""
    monkeypatch.setenv("RPA_PASS1_FILTER", "3")
    kw = dict(seg_len=67, genome_len=1500, level_rate=0.06, query_rate=0.08) if alphabet else dict(seg_len=150, genome_len=3000)
    ds = synth.make_dataset(n_sets=1500, n_hits=20, branching=4, seed=41 + int(step), alphabet=alphabet, **kw)
    sc = ds.batch["score"]
    ds.batch["score"] = (np.maximum(np.round(sc / step), 1.0) * step).astype(sc.dtype)
    rows, stats = _engine_vs_oracle(pkg, oracle, synth, ds, alphabet)
    assert int(stats["pass1"].sum()) > 1000


_DS_CACHE = {}


def _cached_dataset(synth, **kw):
    """Datasets that many parametrisations of one test share (never modified by the tests that use this)."""
    key = tuple(sorted(kw.items()))
    if key not in _DS_CACHE:
        _DS_CACHE[key] = synth.make_dataset(**kw)
    return _DS_CACHE[key]


X_VALUES = (0.0, 0.25, 0.5, 0.8, 1.0)   # taxator -x ("1 is fastest, 0 means using all alignments", extended/binning-last.bash:108)
T_VALUES = (0.0, 0.05, 0.1, 0.3)         # taxator -t


@pytest.mark.parametrize("mode", ["default", "assume-always", "chunks-1-1", "chunks-3-2"])
@pytest.mark.parametrize("t", T_VALUES)
@pytest.mark.parametrize("x", X_VALUES)
def test_heuristic_cutoff_and_reevaluation_band(pkg, oracle, synth, x, t, mode, monkeypatch):
    """`-x` (exclude_alignments_factor_) and `-t` (reeval_bandwidth) off their defaults.  X feeds an int truncation in pass 1
    ((int)(score * X), :657-659: X = 0 leaves the cut-off "unset" at every hit, so every later farther hit sets it again)
    and a float product in pass 2 (:752-757); t decides who joins the query group in pass 0 ((float)(1. - t) * best score,
    :334,:499: t = 0 keeps only ties with the best score, 0.3 a third of a set).  The pass-1 request rules (thresholds of a
    paused loop, the first-hit cut-off assumption) are built on exactly these values: every mode must give the oracle's
    rows, STATS counters and (below) trace for every pair of settings."""
    if mode == "assume-always": monkeypatch.setenv("RPA_PASS1_FILTER", "3")
    spec = {"chunks-1-1": (1, 1), "chunks-3-2": (3, 2)}.get(mode)
    ds = _cached_dataset(synth, n_sets=500, n_hits=20, seg_len=150, branching=4, genome_len=3000, seed=71, variety=True)
    rows, stats = _engine_vs_oracle(pkg, oracle, synth, ds, 0, speculation=spec, exclude_factor=x, reeval_bandwidth=t)
    assert int(stats["pass1"].sum()) > 400
    dp = _cached_dataset(synth, n_sets=200, n_hits=16, seg_len=60, branching=3, genome_len=1500, seed=72, alphabet=1,
                         level_rate=0.06, query_rate=0.08, variety=True)
    _engine_vs_oracle(pkg, oracle, synth, dp, 1, speculation=spec, exclude_factor=x, reeval_bandwidth=t)


@pytest.mark.parametrize("x,t", [(0.0, 0.3), (1.0, 0.0), (0.8, 0.1), (0.25, 0.05), (1.5, 0.5), (0.5, 1.0), (0.999, 0.999)])
def test_cutoff_settings_on_coarse_scores_and_big_sets(pkg, oracle, synth, x, t, monkeypatch):
    """The same on scores rounded to a coarse grid (runs of tied scores and hits exactly ON (float)(int)(X * score)), on
    record sets beyond the 64-hit bit sets, and with values outside the documented range (X > 1: the cut-off lies above the
    score that set it; t = 1: the pass-0 threshold is 0 and every hit is re-aligned) -- the reference does not range-check
    either option (taxator.cpp:286-287)."""
    monkeypatch.setenv("RPA_PASS1_FILTER", "3")
    ds = synth.make_dataset(n_sets=400, n_hits=20, seg_len=150, branching=4, genome_len=3000, seed=73)
    sc = ds.batch["score"]
    ds.batch["score"] = (np.maximum(np.round(sc / 16.0), 1.0) * 16.0).astype(sc.dtype)
    _engine_vs_oracle(pkg, oracle, synth, ds, 0, exclude_factor=x, reeval_bandwidth=t)
    monkeypatch.delenv("RPA_PASS1_FILTER")
    big = synth.make_dataset(n_sets=80, n_hits=70, seg_len=120, branching=3, genome_len=2000, seed=74, variety=True)
    _engine_vs_oracle(pkg, oracle, synth, big, 0, speculation=(5, 3), exclude_factor=x, reeval_bandwidth=t)
    dp = synth.make_dataset(n_sets=150, n_hits=20, seg_len=67, branching=4, genome_len=1500, seed=75, alphabet=1, level_rate=0.06, query_rate=0.08)
    sp = dp.batch["score"]
    dp.batch["score"] = (np.maximum(np.round(sp / 8.0), 1.0) * 8.0).astype(sp.dtype)
    _engine_vs_oracle(pkg, oracle, synth, dp, 1, exclude_factor=x, reeval_bandwidth=t)


def test_scores_of_any_sign_and_masked_records(pkg, oracle, synth):
    """What k_rank_records gathers per record set in one scan -- the rank of every record (score, identities, input order), the
    number of unmasked records, their query range, the best score (whose fraction is the pass-0 threshold, :499) -- on scores
    the parsers let through but no aligner writes: negative ones, zeros of either sign, runs of ties; a third of
    the records masked (a masked record must count for nothing: its score, range and position in the set)."""
    rng = np.random.default_rng(77)
    for k, (n_hits, alphabet) in enumerate(((20, 0), (7, 0), (70, 0), (20, 1))):
        kw = dict(alphabet=1, level_rate=0.06, query_rate=0.08) if alphabet else {}
        ds = synth.make_dataset(n_sets=300, n_hits=n_hits, seg_len=120 if alphabet == 0 else 67, branching=3, genome_len=2000 if alphabet == 0 else 1200,
                                seed=170 + k, variety=True, **kw)
        b = ds.batch
        sc = b["score"].astype(np.float32).copy()
        H = len(sc)
        kind = rng.integers(0, 8, H)
        sc[kind == 0] = -sc[kind == 0]
        sc[kind == 1] = 0.0
        sc[kind == 2] = -0.0
        sc[kind == 3] = np.round(sc[kind == 3] / 64.0) * 64.0
        # scores whose `int cut-off = X * score` (:657-659) leaves the int range: 3e9 (with X = 1 below), +-infinity (X = 0: inf * 0 is a
        # NaN) -- the reference's platform answers INT_MIN there (cvttss2si), and so do the oracle and the device (float_to_int_x86)
        sc[(kind == 4) & (rng.random(H) < 0.1)] = np.float32(3e9)
        sc[(kind == 5) & (rng.random(H) < 0.1)] = np.float32(np.inf)
        sc[(kind == 6) & (rng.random(H) < 0.05)] = np.float32(-np.inf)
        b["score"] = sc
        filt = b["filtered"].copy()
        filt[rng.random(H) < 0.3] = 1
        b["filtered"] = filt
        # (masked records with ranges and scores that would dominate the set if they counted)
        m = filt != 0
        b["qstart"] = np.where(m & (rng.random(H) < 0.5), 1, b["qstart"]).astype(np.uint32)
        b["score"] = np.where(m & (rng.random(H) < 0.5), np.float32(1e9), b["score"]).astype(np.float32)
        # (a set whose best unmasked score is negative has an empty best-reference group -- the score is below its own 95 % --
        # and the reference asserts: those sets get their scores' absolute values)
        so = b["set_offsets"].astype(np.int64)
        sc = b["score"].copy()
        for lo, hi in zip(so[:-1], so[1:]):
            act = b["filtered"][lo:hi] == 0
            if act.any() and sc[lo:hi][act].max() < 0:
                sc[lo:hi] = np.abs(sc[lo:hi])
        b["score"] = sc
        for x, t in ((0.5, 0.05), (0.0, 0.3), (1.0, 0.05)):
            _engine_vs_oracle(pkg, oracle, synth, ds, alphabet, exclude_factor=x, reeval_bandwidth=t)


@pytest.mark.parametrize("strips_band", ["1", "0"])
def test_ring_window_lists_start_over_in_a_rounds_second_launch(pkg, oracle, synth, strips_band, monkeypatch):
    """The ring-window kernel (K1r) keeps the pairs that left its band on a list of its own with a device-side count; a round with
    two launches (RPA_DEFER2=1) runs it twice, and the second run must start that list over (the soak found it appending behind the
    first run's entries: a result nobody wrote, the engine's ASSERT).  Segments of 1500 nt at ~30 % edits: every pair leaves the
    window, in both launches; with the strip kernel's own band on and off."""
    monkeypatch.setenv("RPA_DEFER2", "1")
    monkeypatch.setenv("RPA_BAND_SHORT_MIN", "1")  # (the banded one-strip launches, K1r among them, for launches of any size)
    monkeypatch.setenv("RPA_BAND", strips_band)
    for seg, rate in ((1500, 0.06), (700, 0.03), (450, 0.06)):
        ds = synth.make_dataset(n_sets=14, n_hits=30, seg_len=seg, branching=4, genome_len=4 * seg, seed=1075 + seg, level_rate=rate, query_rate=0.05, variety=True)
        _engine_vs_oracle(pkg, oracle, synth, ds, 0)
        _engine_vs_oracle(pkg, oracle, synth, ds, 0, speculation=(3, 7))


@pytest.mark.parametrize("mode", ["1", "0", "auto"])
def test_two_launches_per_pass2_round(pkg, oracle, synth, mode, monkeypatch):
    """Round 5: pass-2 hits that an earlier hit of their chunk may take out of the loop (by lifting unode_global over them, :771,
    :811-813) wait for the round's second launch; k_defer_filter replays the chunk with the first launch's results and lists the
    ones the loop can still reach (kDeferredMark, rpa_device.h).  Rows, STATS counters and the consumed-alignment trace must be the
    reference's whatever the setting (RPA_DEFER2 = 1: always, 0: never, auto: by pattern length), the chunk sizes (whole loops,
    chunks of 10 / 3 / 1 hits), the cut-off settings, the row layout and the alphabet -- and the setting must really change what is
    computed: fewer alignments with it than without."""
    monkeypatch.setenv("RPA_DEFER2", mode)
    computed = {}
    for alphabet, kw in ((0, dict(seg_len=150, genome_len=3000)), (1, dict(seg_len=60, genome_len=1500, alphabet=1, level_rate=0.06, query_rate=0.08)),
                         (0, dict(seg_len=700, genome_len=4000))):
        ds = synth.make_dataset(n_sets=1200 if kw["seg_len"] < 500 else 150, n_hits=24, branching=4, seed=520 + alphabet + kw["seg_len"], variety=True, **kw)
        for speculation, x, t in ((None, 0.5, 0.05), ((0, 0), 0.5, 0.05), ((0, 0), 0.2, 0.2), ((0, 10), 0.5, 0.05), ((5, 3), 0.5, 0.05), ((1, 1), 0.8, 0.0)):
            _engine_vs_oracle(pkg, oracle, synth, ds, alphabet, speculation=speculation, exclude_factor=x, reeval_bandwidth=t)
        # the trace of consumed alignments, and how many were computed
        model = pkg.RPAPredictionModel(ds.tax, alphabet=alphabet, speculation=(0, 0))
        model.load_store(pkg.STORE_QUERY, ds.query_letters, ds.query_offsets)
        model.load_store(pkg.STORE_DB, ds.ref_letters, ds.ref_offsets)
        model.profile(enable=True, reset=True)
        rows, stats = model.predict(ds.batch)
        prof = model.profile()
        assert prof["demanded"] == int(stats["pass0"].sum() + stats["pass1"].sum() + stats["pass2"].sum())
        computed[(alphabet, kw["seg_len"])] = prof["pairs"] / max(prof["demanded"], 1)
    if mode == "1":
        monkeypatch.setenv("RPA_DEFER2", "0")
        ds = synth.make_dataset(n_sets=1200, n_hits=24, seg_len=150, genome_len=3000, branching=4, seed=520 + 0 + 150, variety=True)
        model = pkg.RPAPredictionModel(ds.tax, alphabet=0, speculation=(0, 0))
        model.load_store(pkg.STORE_QUERY, ds.query_letters, ds.query_offsets)
        model.load_store(pkg.STORE_DB, ds.ref_letters, ds.ref_offsets)
        model.profile(enable=True, reset=True)
        model.predict(ds.batch)
        prof = model.profile()
        assert computed[(0, 150)] < prof["pairs"] / max(prof["demanded"], 1) - 0.02, (computed, prof)


def test_nan_scores_are_refused_where_the_reference_would_sort_them(pkg, oracle, synth):
    """A score that is not a number.  In a record set of ONE unmasked record, or on a masked record, it is never compared and the
    result is the reference's.  Among two or more unmasked records the reference's list::sort (:412) runs on a comparator that
    is no ordering (core/src/alignmentrecord.hh:89-93: every comparison with a NaN is false) and what it leaves is an accident of
    libstdc++'s merge tree: engine and oracle both refuse the batch, naming the record set, instead of inventing an order."""
    ds = synth.make_dataset(n_sets=200, n_hits=6, seg_len=100, branching=3, genome_len=1500, seed=411, variety=True)
    b = ds.batch
    so = b["set_offsets"].astype(np.int64)
    sc = b["score"].astype(np.float32).copy()
    filt = b["filtered"].copy()
    # harmless NaNs: on masked records, and on the only unmasked record of a set
    for s_ in range(0, 60, 3):
        lo, hi = so[s_], so[s_ + 1]
        if hi - lo >= 2:
            filt[lo + 1] = 1
            sc[lo + 1] = np.nan
    for s_ in range(60, 80, 4):
        lo, hi = so[s_], so[s_ + 1]
        if hi - lo >= 1:
            filt[lo:hi] = 1
            filt[lo] = 0
            sc[lo] = np.nan
    b["score"], b["filtered"] = sc, filt
    _engine_vs_oracle(pkg, oracle, synth, ds, 0)
    # ... and one in a set that is sorted
    victim = next(s_ for s_ in range(100, 200) if (filt[so[s_]:so[s_ + 1]] == 0).sum() >= 3)
    act = np.nonzero(filt[so[victim]:so[victim + 1]] == 0)[0]
    sc2 = sc.copy()
    sc2[so[victim] + act[1]] = np.nan
    b["score"] = sc2
    tax = oracle.TaxArrays(ds.tax.parent, ds.tax.left, ds.tax.right, ds.tax.depth)
    q = oracle.StoreArrays(synth.encode(ds.query_letters, 0), ds.query_offsets)
    db = oracle.StoreArrays(synth.encode(ds.ref_letters, 0), ds.ref_offsets)
    rc, _, _ = oracle.predict_batch(tax, q, db, 0, b["set_offsets"], b["query_seq"], b["query_len"], synth.records_aos(b, oracle.RECORD_DTYPE))
    assert rc == 5                                            # ORC_ERR_NAN_SCORE
    model = pkg.RPAPredictionModel(ds.tax, alphabet=pkg.DNA)
    model.load_store(pkg.STORE_QUERY, ds.query_letters, ds.query_offsets)
    model.load_store(pkg.STORE_DB, ds.ref_letters, ds.ref_offsets)
    with pytest.raises(pkg.RpaError) as ei:
        model.predict(b)
    assert ei.value.code == 1 and "not a number" in str(ei.value) and ("set / pair %d)" % victim) in str(ei.value)   # RPA_ERR_INVALID


def test_one_engine_across_batches_of_different_shapes(pkg, oracle, synth):
    """One engine, batches whose row shape and size change from call to call -- 20 hits per set (8-bit path labels, rows of two
    quads), 300 hits per set (16-bit labels, wider rows; a tree of 7^6 species), a handful of sets, many sets, and back: the working set is resized
    and re-tiled between the calls (as the `taxator` driver's blocks do) and every batch equals the oracle."""
    model = None
    for k, (n_sets, n_hits) in enumerate([(300, 20), (12, 300), (2000, 20), (3, 2), (40, 300), (300, 33)]):
        ds = synth.make_dataset(n_sets=n_sets, n_hits=n_hits, seg_len=60, branching=7, genome_len=400, seed=301 + k, variety=(k % 2 == 0))
        if model is None:
            model = pkg.RPAPredictionModel(ds.tax, alphabet=pkg.DNA)
        # (every data set brings its own sequences: the stores are reloaded, the engine and its working set stay)
        model.load_store(pkg.STORE_QUERY, ds.query_letters, ds.query_offsets)
        model.load_store(pkg.STORE_DB, ds.ref_letters, ds.ref_offsets)
        rows, stats = model.predict(ds.batch)
        tax = oracle.TaxArrays(ds.tax.parent, ds.tax.left, ds.tax.right, ds.tax.depth)
        q = oracle.StoreArrays(synth.encode(ds.query_letters, 0), ds.query_offsets)
        db = oracle.StoreArrays(synth.encode(ds.ref_letters, 0), ds.ref_offsets)
        rc, exp, est = oracle.predict_batch(tax, q, db, 0, ds.batch["set_offsets"], ds.batch["query_seq"], ds.batch["query_len"],
                                            synth.records_aos(ds.batch, oracle.RECORD_DTYPE))
        assert rc == 0
        assert rows.tobytes() == exp.tobytes() and stats.tobytes() == est.tobytes(), (k, n_sets, n_hits)


def test_band_widens_when_too_many_pairs_leave_it(pkg, oracle, synth, monkeypatch, capfd):
    """The one-strip band has a three-word middle (exact up to 80 - |delta| edits); a batch in which more than one banded
    pair in sixteen lies beyond that makes the engine use four words all the way (96 - |delta|) for the following batches.
    Divergent references: the first batch re-runs many pairs, the next ones fewer -- and all of them equal the oracle."""
    monkeypatch.setenv("RPA_TRACE_ROUNDS", "1")
    monkeypatch.setenv("RPA_BAND_SHORT_MIN", "1")  # (launches of any size go through the band: the pairs of a batch are spread over three length classes)
    ds = synth.make_dataset(n_sets=8000, n_hits=20, seg_len=200, branching=4, genome_len=3000, seed=51, level_rate=0.045, query_rate=0.06)
    model = pkg.RPAPredictionModel(ds.tax, alphabet=pkg.DNA)
    model.load_store(pkg.STORE_QUERY, ds.query_letters, ds.query_offsets)
    model.load_store(pkg.STORE_DB, ds.ref_letters, ds.ref_offsets)
    tax = oracle.TaxArrays(ds.tax.parent, ds.tax.left, ds.tax.right, ds.tax.depth)
    q = oracle.StoreArrays(synth.encode(ds.query_letters, 0), ds.query_offsets)
    db = oracle.StoreArrays(synth.encode(ds.ref_letters, 0), ds.ref_offsets)
    rc, exp, est = oracle.predict_batch(tax, q, db, 0, ds.batch["set_offsets"], ds.batch["query_seq"], ds.batch["query_len"],
                                        synth.records_aos(ds.batch, oracle.RECORD_DTYPE))
    assert rc == 0
    import re
    left = []
    for _ in range(3):
        capfd.readouterr()
        rows, stats = model.predict(ds.batch)
        assert rows.tobytes() == exp.tobytes() and stats.tobytes() == est.tobytes()
        m = re.search(r"batch done: (\d+) of (\d+) pairs left the one-strip band \((three|four)", capfd.readouterr().err)
        assert m, "no band summary in the round trace"
        left.append((int(m.group(1)), int(m.group(2)), m.group(3)))
    assert left[0][2] == "three" and left[0][0] * 16 > left[0][1], left       # the narrow band loses more than one pair in sixteen
    assert left[1][2] == "four" and left[2][2] == "four" and left[1][0] < left[0][0], left  # ... so the next batches run in the wide one


def test_ring_window_moves_to_its_wide_instance_and_then_to_the_strips(pkg, oracle, synth, monkeypatch, capfd):
    """K1r's feedback per length class: divergent 500 nt segments (~150 edits against the narrow window's 160) make the class move
    to its wide instance (10 words: 288) -- within the batch, from the totals every round's counters carry -- and stay there for
    the following batches, where hardly a pair leaves; 1500 nt segments at ~30 % edits (450) leave the wide one (14 words: 416)
    too and end on the strips.  Every batch equals the oracle whatever instance ran."""
    import re
    monkeypatch.setenv("RPA_TRACE_ROUNDS", "1")
    monkeypatch.setenv("RPA_BAND_SHORT_MIN", "1")
    for seg, n_sets, cls, expect in ((500, 400, "0", "wide window"), (1500, 60, "2", "strips")):
        ds = synth.make_dataset(n_sets=n_sets, n_hits=20, seg_len=seg, branching=4, genome_len=4 * seg, seed=77 + seg, level_rate=0.05, query_rate=0.06)
        model = pkg.RPAPredictionModel(ds.tax, alphabet=pkg.DNA)
        model.load_store(pkg.STORE_QUERY, ds.query_letters, ds.query_offsets)
        model.load_store(pkg.STORE_DB, ds.ref_letters, ds.ref_offsets)
        tax = oracle.TaxArrays(ds.tax.parent, ds.tax.left, ds.tax.right, ds.tax.depth)
        q = oracle.StoreArrays(synth.encode(ds.query_letters, 0), ds.query_offsets)
        db = oracle.StoreArrays(synth.encode(ds.ref_letters, 0), ds.ref_offsets)
        rc, exp, est = oracle.predict_batch(tax, q, db, 0, ds.batch["set_offsets"], ds.batch["query_seq"], ds.batch["query_len"],
                                            synth.records_aos(ds.batch, oracle.RECORD_DTYPE))
        assert rc == 0
        seen = []  # per batch: {length class: what it runs on next}
        for _ in range(4):
            capfd.readouterr()
            rows, stats = model.predict(ds.batch)
            assert rows.tobytes() == exp.tobytes() and stats.tobytes() == est.tobytes()
            err = capfd.readouterr().err
            seen.append({c: nxt for (_l, _n, c, nxt) in re.findall(r"batch done: (\d+) of (\d+) pairs left the ring window \(class (\d); next batch: ([a-z ]+)\)", err)})
        if expect == "wide window":  # the first batch already ends on the wide instance (the move happens between its rounds) and the class stays there
            assert all(b.get(cls) == "wide window" for b in seen), (seg, seen)
        else:  # narrow -> wide -> strips (an instance is judged on 1024 pairs of its own: with 1200 pairs per round that takes a batch or two), then no window at all
            said = [b.get(cls) for b in seen]
            assert "strips" in said and said[-1] is None and said.index("strips") < len(said) - 1 and all(x is None for x in said[said.index("strips") + 1:]), (seg, seen)


def test_predict_dna_ragged_sets_and_lengths(pkg, oracle, synth):
    """Record sets of different sizes (1..40 records) and segment lengths from 40 to 700 bp in one batch."""
    parts = [synth.make_dataset(n_sets=150, n_hits=h, seg_len=L, branching=3, genome_len=2500, seed=10 + k, variety=True)
             for k, (h, L) in enumerate([(3, 40), (7, 90), (20, 200), (40, 256), (12, 300), (9, 700)])]
    # all parts share nothing: run each through its own engine, plus one merged batch against a common taxonomy
    for ds in parts:
        _engine_vs_oracle(pkg, oracle, synth, ds, 0)


@pytest.mark.parametrize("n_hits", [33, 64, 65])
def test_predict_dna_group_bit_sets(pkg, oracle, synth, n_hits):
    """k_decide keeps the query group / outgroup of a record set of up to 64 hits as register bit sets (one or two words)
    and walks without stores; larger sets take the lists in memory.  Both sides of the switch, with loops that pause
    (small speculation chunks: the bit sets travel through SetState) and without."""
    ds = synth.make_dataset(n_sets=250, n_hits=n_hits, seg_len=120, branching=3, genome_len=2000, seed=60 + n_hits, variety=True)
    _engine_vs_oracle(pkg, oracle, synth, ds, 0)
    _engine_vs_oracle(pkg, oracle, synth, ds, 0, speculation=(5, 3))


@pytest.mark.parametrize("big_min", ["2", "7", "default", "0"])
def test_rank_with_a_workgroup_per_record_set(pkg, oracle, synth, big_min, monkeypatch):
    """Round 5: record sets from RPA_BIG_SET_MIN records on (192 by default, up to 1024) are ranked by k_rank_big_sets -- a workgroup per
    set, the set's columns staged in LDS -- instead of by k_rank_records' thread per record.  With the threshold at 2 EVERY record set
    of these batches takes the new kernel (sets of one record, masked records, perfect hits, ties, 8- / 16- / 32-bit labels, deep
    taxonomies, protein, windows of a resident store all included), at 7 a batch mixes both kernels tile by tile, 0 switches it off:
    the same rows and counters as the oracle every time."""
    if big_min != "default": monkeypatch.setenv("RPA_BIG_SET_MIN", big_min)
    ds = synth.make_dataset(n_sets=900, n_hits=20, seg_len=150, branching=4, genome_len=3000, seed=311, variety=True)
    _engine_vs_oracle(pkg, oracle, synth, ds, 0)
    _engine_vs_oracle(pkg, oracle, synth, ds, 0, speculation=(5, 3))
    ds = synth.make_ragged_dataset([(300, 3, 120), (200, 9, 100), (60, 40, 90), (30, 100, 80), (12, 300, 60), (3, 700, 60)], seed=31, branching=3)
    _engine_vs_oracle(pkg, oracle, synth, ds, 0)
    ds = synth.make_dataset(n_sets=300, n_hits=12, seg_len=67, branching=3, genome_len=1500, seed=312, alphabet=1, level_rate=0.06, query_rate=0.08, variety=True)
    _engine_vs_oracle(pkg, oracle, synth, ds, 1)
    for bits in ("16", "32"):  # wider taxon path labels than the batch needs: the other two packings of a row's label words
        monkeypatch.setenv("RPA_LABEL_BITS", bits)
        ds = synth.make_dataset(n_sets=200, n_hits=24, seg_len=100, branching=3, genome_len=2000, seed=313, variety=True)
        _engine_vs_oracle(pkg, oracle, synth, ds, 0)
    monkeypatch.delenv("RPA_LABEL_BITS")


@pytest.mark.parametrize("n_hits", [100, 300, 530])
def test_predict_dna_large_record_sets(pkg, oracle, synth, n_hits):
    """Record sets of hundreds of hits (k_decide keeps their outgroup as lists in memory, path labels of 8 and 16 bits), with
    loops that pause at several chunk sizes and without: the same rows and counters whatever the chunks."""
    ds = synth.make_dataset(n_sets=24, n_hits=n_hits, seg_len=70, branching=3, genome_len=1500, seed=700 + n_hits, variety=True)
    rows, stats = _engine_vs_oracle(pkg, oracle, synth, ds, 0)
    for spec in ((5, 3), (40, 70)):
        rows_p, stats_p = _engine_vs_oracle(pkg, oracle, synth, ds, 0, speculation=spec)
        assert rows_p.tobytes() == rows.tobytes() and stats_p.tobytes() == stats.tobytes()


def test_predict_dna_mixed_set_sizes_one_batch(pkg, oracle, synth):
    """One batch with record sets on every path of k_decide at once: <= 32 hits (packed LCA depths), <= 64 (register bit sets),
    beyond (lists in memory) -- as a real alignment stream mixes them (tools/ragged_bench.py)."""
    ds = synth.make_ragged_dataset([(200, 12, 150), (60, 40, 100), (40, 64, 90), (24, 65, 90), (16, 150, 70), (8, 400, 60), (3, 600, 60)],
                                   seed=5, branching=3)
    _engine_vs_oracle(pkg, oracle, synth, ds, 0)
    _engine_vs_oracle(pkg, oracle, synth, ds, 0, speculation=(6, 4))


@pytest.mark.parametrize("lanes", ["lanes-per-pair", "one-lane"])
def test_predict_protein(pkg, oracle, synth, lanes, monkeypatch):
    if lanes == "one-lane":
        monkeypatch.setenv("RPA_PROTEIN_LANES", "0")
    ds = synth.make_dataset(n_sets=400, n_hits=20, seg_len=67, branching=3, genome_len=1500, seed=4, alphabet=1,
                            level_rate=0.06, query_rate=0.08, variety=True)
    _engine_vs_oracle(pkg, oracle, synth, ds, 1)
    ds = synth.make_dataset(n_sets=60, n_hits=10, seg_len=200, branching=2, genome_len=1500, seed=5, alphabet=1,
                            level_rate=0.06, query_rate=0.08)
    _engine_vs_oracle(pkg, oracle, synth, ds, 1)
    # segments around the ~312 x 312 limit of the packed-cell kernel: one batch feeds both protein kernels
    ds = synth.make_dataset(n_sets=60, n_hits=10, seg_len=312, branching=2, genome_len=1500, seed=6, alphabet=1,
                            level_rate=0.06, query_rate=0.08)
    rows, stats = _engine_vs_oracle(pkg, oracle, synth, ds, 1)
    lens = np.abs(ds.batch["rstop"].astype(np.int64) - ds.batch["rstart"].astype(np.int64)) + 1
    assert (13 * lens + 32 <= 4095).any() and (13 * lens + 32 > 4095).any()
    # beyond it: 450 and 900 residues (the score-and-length cell up to 1258 x 1258, K3 behind it) -- and the same rows with it off
    # (round 5: with the lanes-per-pair kernel the strips of such a pair sit in 16 / 21 / 32 / 64 adjacent lanes -- 330, 450, 600, 900 aa)
    for seg, n_sets in ((330, 40), (450, 40), (600, 24), (900, 16), (1300, 8)):
        ds = synth.make_dataset(n_sets=n_sets, n_hits=8, seg_len=seg, branching=2, genome_len=2 * seg + 600, seed=7 + seg, alphabet=1,
                                level_rate=0.06, query_rate=0.08)
        rows, stats = _engine_vs_oracle(pkg, oracle, synth, ds, 1)
        monkeypatch.setenv("RPA_PROTEIN_WIDE", "0")
        rows0, stats0 = _engine_vs_oracle(pkg, oracle, synth, ds, 1)
        monkeypatch.delenv("RPA_PROTEIN_WIDE")
        assert rows.tobytes() == rows0.tobytes() and stats.tobytes() == stats0.tobytes()


def test_predict_long_contigs(pkg, oracle, synth):
    """BASELINE config 5 shape at a size the oracle finishes in seconds: 10 kbp x 10 kbp, 12 hits."""
    ds = synth.make_dataset(n_sets=6, n_hits=12, seg_len=10000, branching=2, genome_len=30000, seed=6)
    rows, stats = _engine_vs_oracle(pkg, oracle, synth, ds, 0)
    assert int(stats["pass1"].sum()) > 0


def _load_survey(folder, alphabet, tmp_path, **kw):
    ext = ".fna" if alphabet == 0 else ".faa"
    gz = os.path.join(folder, "ref" + ext + ".gz")
    if os.path.exists(gz):
        work = tmp_path / os.path.basename(folder)
        shutil.copytree(folder, work)
        with gzip.open(gz, "rb") as f, open(work / ("ref" + ext), "wb") as g:
            g.write(f.read())
        folder = str(work)
    return R.Dataset(folder, alphabet, **kw)


def _letters(fasta, alphabet):
    return (LETTERS_DNA if alphabet == 0 else LETTERS_AA)[fasta.residues]


@pytest.mark.parametrize("name,alphabet", [("survey_dna", 0), ("survey_protein", 1)])
def test_gff3_matches_reference_output(pkg, golden_dir, tmp_path, name, alphabet):
    """Through the engine, the GFF3 the reference code printed for the same input files (secondary evidence,
    tests/golden/make_survey_fixture.py)."""
    ds = _load_survey(os.path.join(golden_dir, name), alphabet, tmp_path)
    b = ds.batch()
    model = pkg.RPAPredictionModel(ds.tax, alphabet=alphabet)
    model.load_store(pkg.STORE_QUERY, _letters(ds.query, alphabet), ds.query.offsets)
    model.load_store(pkg.STORE_DB, _letters(ds.ref, alphabet), ds.ref.offsets)
    recs = b["recs"]
    batch = dict(set_offsets=b["offsets"], query_seq=b["qseq"], query_len=b["qlen"], filtered=recs["filtered"].astype(np.uint8))
    for f in ("qstart", "qstop", "rstart", "rstop", "score", "identities", "alnlen", "ref_seq", "node"):
        batch[f] = recs[f]
    rows, stats = model.predict(batch)
    assert ds.gff3(rows, b["qids"], b["qlen"]) == open(os.path.join(golden_dir, name, "out.gff3")).read()


def test_edge_sets_and_errors(pkg, oracle, golden_dir):
    folder = os.path.join(golden_dir, "survey_edge")
    ds = R.Dataset(folder, 0, query_name="query_edge.fna", aln_name="aln_edge.tsv")
    b = ds.batch()
    model = pkg.RPAPredictionModel(ds.tax, alphabet=pkg.DNA)
    model.load_store(pkg.STORE_QUERY, _letters(ds.query, 0), ds.query.offsets)
    model.load_store(pkg.STORE_DB, _letters(ds.ref, 0), ds.ref.offsets)
    recs = b["recs"]
    batch = dict(set_offsets=b["offsets"], query_seq=b["qseq"], query_len=b["qlen"], filtered=recs["filtered"].astype(np.uint8))
    for f in ("qstart", "qstop", "rstart", "rstop", "score", "identities", "alnlen", "ref_seq", "node"):
        batch[f] = recs[f].copy()
    rows, stats = model.predict(batch)
    tax = oracle.TaxArrays(ds.tax.parent, ds.tax.left, ds.tax.right, ds.tax.depth)
    q = oracle.StoreArrays(ds.query.residues, ds.query.offsets)
    db = oracle.StoreArrays(ds.ref.residues, ds.ref.offsets)
    rc, exp, est = oracle.predict_batch(tax, q, db, 0, b["offsets"], b["qseq"], b["qlen"], recs)
    assert rows.tobytes() == exp.tobytes() and stats.tobytes() == est.tobytes()
    exp_json = json.load(open(os.path.join(folder, "expected_edge.json")))
    assert [int(s) for s in stats["pass1"]] == [e["stats"][1] for e in exp_json]
    # empty batch, then a hit that starts beyond the end of its reference: SequenceRangeError, like the reference
    bad = dict(batch)
    bad["rstart"] = batch["rstart"].copy()
    bad["rstop"] = batch["rstop"].copy()
    k = int(b["offsets"][3])  # first record of e3 (3 active records -> alignments happen)
    bad["rstart"][k] = 10 ** 8
    bad["rstop"][k] = 10 ** 8 + 150
    with pytest.raises(pkg.RpaError) as ei:
        model.predict(bad)
    assert ei.value.code == 3
    # engine still usable afterwards
    rows2, _ = model.predict(batch)
    assert rows2.tobytes() == rows.tobytes()
    # protein engine: reverse reference coordinates are an assert in the reference build
    pm = pkg.RPAPredictionModel(ds.tax, alphabet=pkg.PROTEIN)
    pm.load_store(pkg.STORE_QUERY, _letters(ds.query, 0), ds.query.offsets)
    pm.load_store(pkg.STORE_DB, _letters(ds.ref, 0), ds.ref.offsets)
    with pytest.raises(pkg.RpaError) as ei:
        pm.predict(batch)
    assert ei.value.code == 5


def test_device_resident_batch_and_profile(pkg, oracle, synth):
    """The bench path: batch arrays already in HBM (torch tensors), results left on the device, live kernel timing."""
    import torch
    ds = synth.make_dataset(n_sets=2000, n_hits=20, seg_len=200, branching=4, genome_len=3000, seed=9)
    model = pkg.RPAPredictionModel(ds.tax, alphabet=pkg.DNA)
    model.load_store(pkg.STORE_QUERY, ds.query_letters, ds.query_offsets)
    model.load_store(pkg.STORE_DB, ds.ref_letters, ds.ref_offsets)
    host_rows, host_stats = model.predict(ds.batch)
    dev = {k: torch.from_numpy(np.ascontiguousarray(v).view(np.int32) if v.dtype == np.uint32 else np.ascontiguousarray(v)).cuda()
           for k, v in ds.batch.items()}
    out = torch.zeros(ds.n_sets * 8, dtype=torch.int32, device="cuda")
    st = torch.zeros(ds.n_sets * 4, dtype=torch.int32, device="cuda")
    model.profile(enable=True, reset=True)
    model.set_stream(torch.cuda.current_stream().cuda_stream)
    model.predict(dev, mem_kind=pkg.MEM_DEVICE, out=out, stats=st)
    torch.cuda.synchronize()
    assert out.cpu().numpy().tobytes() == host_rows.tobytes()
    assert st.cpu().numpy().tobytes() == host_stats.tobytes()
    prof = model.profile()
    demanded = int(host_stats["pass0"].sum() + host_stats["pass1"].sum() + host_stats["pass2"].sum())
    assert prof["demanded"] == demanded and prof["pairs"] >= demanded and prof["launches"] >= 3
    assert 0 < prof["align_ms"] <= prof["total_ms"]
    assert prof["residue_bytes"] > prof["pairs"] * 300 and prof["cells"] > prof["pairs"] * 150 * 150
    # launch timing alone (RPA_PROFILE_TIMING): the kernels carry no counters, the launches are timed as before, same rows
    model.profile(enable=pkg.PROFILE_TIMING, reset=True)
    model.predict(dev, mem_kind=pkg.MEM_DEVICE, out=out, stats=st)
    torch.cuda.synchronize()
    t_only = model.profile()
    assert out.cpu().numpy().tobytes() == host_rows.tobytes()
    assert t_only["launches"] == prof["launches"] and 0 < t_only["align_ms"] <= t_only["total_ms"] and t_only["rounds"] == prof["rounds"]
    assert t_only["pairs"] == 0 and t_only["cells"] == 0 and t_only["demanded"] == 0
    model.profile(enable=False)


def test_reverse_strand_at_store_edges_and_tiny_segments(pkg, oracle):
    """Segments that touch the first / last residues of the store, reverse strand included (the in-place reader's
    funnel shift runs off the front of the store there), 1..40 nt long, against the oracle."""
    rng = np.random.default_rng(17)
    ref = rng.integers(0, 5, 300).astype(np.uint8)
    qry = rng.integers(0, 4, 120).astype(np.uint8)

    class T:  # root + 3 levels, 4 leaves
        parent = np.array([0, 0, 1, 1, 2, 2, 3, 3], np.uint32)
        depth = np.array([0, 1, 2, 2, 3, 3, 3, 3], np.uint8)
        left = np.array([1, 2, 3, 9, 4, 6, 10, 12], np.uint32)
        right = np.array([16, 15, 8, 14, 5, 7, 11, 13], np.uint32)

    model = pkg.RPAPredictionModel(T(), alphabet=pkg.DNA, speculation=(2, 2))
    model.load_store(pkg.STORE_QUERY, LETTERS_DNA[qry], np.array([0, 120], np.uint64))
    model.load_store(pkg.STORE_DB, LETTERS_DNA[ref], np.array([0, 300], np.uint64))
    tax = oracle.TaxArrays(T.parent, T.left, T.right, T.depth)
    q = oracle.StoreArrays(qry, [0, 120])
    db = oracle.StoreArrays(ref, [0, 300])
    sets = []
    for qs, qe in [(1, 1), (1, 3), (1, 8), (1, 9), (1, 40), (100, 120), (113, 120), (120, 120)]:
        hits = []
        span = qe - qs
        for (a, b) in [(1, 1 + span), (1 + span, 1), (2, 2 + span), (2 + span, 2), (300 - span, 300), (300, 300 - span), (7, 7 + span), (9 + span, 9)]:
            hits.append((qs, qe, a, b, float(rng.integers(10, 50)), int(rng.integers(0, span + 1)), span + 1, 0, int(rng.integers(4, 8)), 0))
        sets.append(hits)
    recs = np.array([h for s in sets for h in s], oracle.RECORD_DTYPE)
    offs = np.cumsum([0] + [len(s) for s in sets]).astype(np.uint32)
    batch = dict(set_offsets=offs, query_seq=np.zeros(len(sets), np.uint32), query_len=np.full(len(sets), 120, np.uint32),
                 filtered=np.zeros(len(recs), np.uint8))
    for f in ("qstart", "qstop", "rstart", "rstop", "score", "identities", "alnlen", "ref_seq", "node"):
        batch[f] = recs[f].copy()
    rows, stats = model.predict(batch)
    rc, exp, est = oracle.predict_batch(tax, q, db, 0, offs, batch["query_seq"], batch["query_len"], recs)
    assert rc == 0
    assert rows.tobytes() == exp.tobytes() and stats.tobytes() == est.tobytes()
    assert int(est["pass1"].sum()) > 0


@pytest.mark.parametrize("shape", ["40 levels", "63 levels, uneven", "40 levels, sets of 300 records"])
def test_deep_taxonomies(pkg, oracle, synth, shape):
    """Taxonomies far deeper than the seven default ranks (taxator -d 0 keeps every node of the NCBI tree: ~40 levels; the
    reference walks parents and takes any depth, core/src/taxonomyinterface.cpp:52-77, core/src/taxontree.cpp:55-70,169-178).
    The same workloads on trees whose edges carry chains of extra nodes: taxon path labels of 40 and 63 levels (rows of 4
    and 6 quads), 16-bit labels on top (8 quads), BandFactor over up to 64 ranks -- every row and counter against the oracle."""
    if shape == "40 levels":
        ds = synth.make_dataset(n_sets=1500, n_hits=20, seg_len=150, branching=4, genome_len=2500, seed=41, variety=True)
        synth.stretch_dataset(ds, lambda i: 4 if ds.tax.depth[i] < 7 else 10)       # 6 x 5 + 11 = 41 levels
    elif shape == "63 levels, uneven":
        ds = synth.make_dataset(n_sets=800, n_hits=12, seg_len=120, branching=3, genome_len=2000, seed=42, variety=True)
        synth.stretch_dataset(ds, lambda i: (i * 7 + 3) % 10 if ds.tax.depth[i] > 1 else 0)
        assert int(ds.tax.depth.max()) <= 63
    else:
        ds = synth.make_dataset(n_sets=30, n_hits=300, seg_len=100, branching=4, genome_len=1200, seed=43, variety=True)
        synth.stretch_dataset(ds, lambda i: 4 if ds.tax.depth[i] < 7 else 10)
    assert int(ds.tax.depth.max()) >= 40
    rows, stats = _engine_vs_oracle(pkg, oracle, synth, ds, 0)
    assert len(set(ds.tax.depth[rows["lower"]])) > 4   # results at many different depths of the deep tree
    _engine_vs_oracle(pkg, oracle, synth, ds, 0, speculation=(3, 2))


def _random_big_tree(n, max_depth, rng):
    """A random taxonomy of n nodes and up to max_depth levels as (parent, left, right, depth): node i hangs under an earlier
    node, four of five times one of the previous batch (so the tree gains a level per batch), nested-set numbers by a walk."""
    parent = np.zeros(n, np.uint32)
    depth = np.zeros(n, np.uint8)
    lo, prev_lo = 1, 0
    while lo < n:
        hi = min(n, lo + 1000 if lo < 4000 else int(lo * 1.22))
        deep = rng.random(hi - lo) < 0.8
        cand = np.where(deep, rng.integers(prev_lo, lo, hi - lo), rng.integers(0, lo, hi - lo))
        cand[depth[cand] >= max_depth] = 0
        parent[lo:hi] = cand
        depth[lo:hi] = depth[cand] + 1
        prev_lo, lo = lo, hi
    order = np.argsort(parent[1:], kind="stable") + 1
    start = np.searchsorted(parent[order], np.arange(n + 1))
    left = np.zeros(n, np.uint32)
    right = np.zeros(n, np.uint32)
    nxt = start[:-1].copy()
    stack, ctr = [0], 1
    left[0] = 1
    while stack:
        v = stack[-1]
        if nxt[v] < start[v + 1]:
            c = int(order[nxt[v]])
            nxt[v] += 1
            ctr += 1
            left[c] = ctr
            stack.append(c)
        else:
            ctr += 1
            right[v] = ctr
            stack.pop()
    return parent, left, right, depth


def test_taxonomy_of_ncbi_shape(pkg, oracle, synth):
    """`taxator -d 0` keeps every node of the NCBI tree (~2.5 M nodes, ~40 levels).  Here a random tree of 500 k nodes and close to
    30 levels (tools/big_taxonomy.py runs the full size, 37 levels: profiles/r03_big_taxonomy.txt): rpa_create builds its preorder /
    ancestor tables from the parent pointers, an RPA batch whose taxa are random nodes of the big tree and a megan-lca batch
    over it equal the oracle, and the results sit at depths beyond the pruned tree's eight levels."""
    rng = np.random.default_rng(7)
    n = 500_000
    parent, left, right, depth = _random_big_tree(n, 37, rng)
    assert int(depth.max()) >= 25

    class T:
        pass
    tax = T()
    tax.parent, tax.left, tax.right, tax.depth = parent, left, right, depth
    model = pkg.RPAPredictionModel(tax, alphabet=pkg.DNA)
    ds = synth.make_dataset(n_sets=2000, n_hits=20, seg_len=150, branching=4, genome_len=2500, seed=5, variety=True)
    remap = rng.integers(1, n, len(ds.tax.parent)).astype(np.uint32)
    ds.batch["node"] = remap[ds.batch["node"]]
    model.load_store(pkg.STORE_QUERY, ds.query_letters, ds.query_offsets)
    model.load_store(pkg.STORE_DB, ds.ref_letters, ds.ref_offsets)
    rows, stats = model.predict(ds.batch)
    otax = oracle.TaxArrays(parent, left, right, depth)
    q = oracle.StoreArrays(synth.encode(ds.query_letters, 0), ds.query_offsets)
    db = oracle.StoreArrays(synth.encode(ds.ref_letters, 0), ds.ref_offsets)
    rc, exp, est = oracle.predict_batch(otax, q, db, 0, ds.batch["set_offsets"], ds.batch["query_seq"], ds.batch["query_len"],
                                        synth.records_aos(ds.batch, oracle.RECORD_DTYPE))
    assert rc == 0 and rows.tobytes() == exp.tobytes() and stats.tobytes() == est.tobytes()
    assert int(depth[rows["lower"]].max()) > 8
    import test_lca_models as TL
    offs, qlen, recs, evalue = TL.random_batch(rng, tax, 20000, 30)
    uncl = (rng.random(n) < 0.1).astype(np.uint8)
    rc, exp = oracle.lca_predict_batch(otax, oracle.LCA_MEGAN, offs, qlen, recs, evalue, uncl, 0.3, 0.0, 1000.0, 2, 1, True)
    batch = dict(set_offsets=offs, query_len=qlen, filtered=recs["filtered"].astype(np.uint8))
    for f in ("qstart", "qstop", "score", "node"):
        batch[f] = recs[f].copy()
    got = model.lca_predict(batch, pkg.LCA_MEGAN, evalue=evalue, node_unclassified=uncl, toppercent=0.3, minscore=0.0, maxevalue=1000.0,
                            minsupport=2, nbest=1, ignore_unclassified=True)
    assert rc == 0 and got.tobytes() == exp.tobytes()


@pytest.mark.parametrize("bits", ["16", "32"])
def test_wide_path_labels(pkg, oracle, synth, bits, monkeypatch):
    """RPA_LABEL_BITS forces the label width a batch with record sets of more than 256 / 65536 records would get (rows of 3
    and 4 quads on the default 8-level tree): same results."""
    monkeypatch.setenv("RPA_LABEL_BITS", bits)
    ds = synth.make_dataset(n_sets=2000, n_hits=20, seg_len=200, branching=4, genome_len=3000, seed=44, variety=True)
    _engine_vs_oracle(pkg, oracle, synth, ds, 0)


def _trace_vs_oracle(pkg, oracle, synth, ds, alphabet, x=0.5, t=0.05, min_checked=500):
    """The engine's trace of consumed alignments against the oracle's trace callback, per record set; returns what the
    caller needs to go on (model, rows, trace)."""
    model = pkg.RPAPredictionModel(ds.tax, alphabet=alphabet, exclude_factor=x, reeval_bandwidth=t)
    model.load_store(pkg.STORE_QUERY, ds.query_letters, ds.query_offsets)
    model.load_store(pkg.STORE_DB, ds.ref_letters, ds.ref_offsets)
    model.trace(4_000_000)
    rows, stats = model.predict(ds.batch)
    tr, dropped = model.read_trace()
    assert dropped == 0
    demanded = stats["pass0"].astype(np.int64) + stats["pass1"] + stats["pass2"]
    assert len(tr) == int(demanded.sum())
    tax = oracle.TaxArrays(ds.tax.parent, ds.tax.left, ds.tax.right, ds.tax.depth)
    q = oracle.StoreArrays(synth.encode(ds.query_letters, alphabet), ds.query_offsets)
    db = oracle.StoreArrays(synth.encode(ds.ref_letters, alphabet), ds.ref_offsets)
    recs = synth.records_aos(ds.batch, oracle.RECORD_DTYPE)
    first = np.concatenate([[0], np.cumsum(demanded)])
    b = ds.batch
    checked = 0
    for s_ in range(0, ds.n_sets, 3 if ds.n_sets > 200 else 1):
        lo, hi = int(b["set_offsets"][s_]), int(b["set_offsets"][s_ + 1])
        rc, out, st, want = oracle.predict(tax, q, db, alphabet, int(b["query_seq"][s_]), int(b["query_len"][s_]), recs[lo:hi], exclude_factor=x, reeval_bandwidth=t, want_trace=True)
        assert rc == 0
        got = tr[first[s_]:first[s_ + 1]]
        assert (got["set"] == s_).all() and list(got["seq"]) == list(range(len(got)))
        assert len(got) == len(want), (s_, len(got), len(want))
        for g, w_ in zip(got, want):
            assert (int(g["pass"]), int(g["i"]), int(g["j"])) == (w_[0], w_[1], w_[2]), (s_, g, w_)
            assert np.float32(g["dist"]).tobytes() == np.float32(w_[3]).tobytes() and np.float32(g["sim"]).tobytes() == np.float32(w_[4]).tobytes(), (s_, g, w_)
        checked += len(want)
    assert checked > min_checked
    return model, rows, tr


@pytest.mark.parametrize("alphabet,kw", [(0, dict(n_sets=700, n_hits=20, seg_len=200, branching=4, genome_len=3000, seed=61)),
                                         (0, dict(n_sets=60, n_hits=90, seg_len=120, branching=4, genome_len=1500, seed=62)),
                                         (1, dict(n_sets=500, n_hits=16, seg_len=70, branching=4, genome_len=1200, seed=63, level_rate=0.06, query_rate=0.08))])
@pytest.mark.parametrize("parts", [None, "3"])
def test_alignment_trace_equals_the_oracles(pkg, oracle, synth, alphabet, kw, parts, monkeypatch):
    """rpa_trace_begin / rpa_trace_read: the alignments the control flow CONSUMES, in the order the reference computes and
    logs them ("+ALN i <=> j dist=..; sim=..", taxonpredictionmodelsequence.hh:533,636,782,802) -- per record set the same
    (pass, i, j, distance, similarity) sequence as the oracle's trace callback (which tests/test_oracle_placement.py holds
    against the reference's own log, tests/golden/survey_*/expected.json), float bit patterns included, whatever the engine
    computed speculatively on the side; also when the batch is cut into parts."""
    if parts:
        monkeypatch.setenv("RPA_PARTS", parts)
    ds = synth.make_dataset(variety=True, alphabet=alphabet, **kw)
    model, rows, tr = _trace_vs_oracle(pkg, oracle, synth, ds, alphabet)
    # a capacity that is too small: what fits is a prefix-free sample, the rest is counted
    model.trace(100)
    model.predict(ds.batch)
    tr2, dropped2 = model.read_trace()
    assert len(tr2) == 100 and dropped2 == len(tr) - 100
    model.trace(0)
    rows3, _ = model.predict(ds.batch)
    assert rows3.tobytes() == rows.tobytes()


@pytest.mark.parametrize("x,t", [(0.0, 0.3), (1.0, 0.0), (0.8, 0.1), (0.25, 0.3)])
@pytest.mark.parametrize("alphabet", [0, 1])
def test_alignment_trace_under_cutoff_settings(pkg, oracle, synth, alphabet, x, t):
    """The consumed-alignment trace (= the reference's +ALN log lines) with `-x` / `-t` off their defaults: same alignments
    in the same order with the same float bit patterns as the oracle at the same settings."""
    kw = dict(n_sets=240, n_hits=16, seg_len=60, branching=3, genome_len=1500, seed=65, level_rate=0.06, query_rate=0.08) if alphabet else \
        dict(n_sets=400, n_hits=20, seg_len=150, branching=4, genome_len=3000, seed=63)
    ds = synth.make_dataset(variety=True, alphabet=alphabet, **kw)
    _trace_vs_oracle(pkg, oracle, synth, ds, alphabet, x=x, t=t, min_checked=60)  # (X = 1 cuts nearly every loop short)


def test_windowed_stores(pkg, oracle, synth):
    """Region-granular residency through the C ABI (rpa_store_reserve_windows / rpa_store_append_windows / rpa_store_reset):
    only the windows the batch reads are resident -- tile-aligned pieces of the reference sequences, appended in several
    calls, later ones swallowing earlier ones -- and the rows equal those of whole resident stores.  A segment outside the
    resident windows fails the batch; a window that overlaps a resident one partly is refused; after a reset the store
    is refilled from its start."""
    ds = synth.make_dataset(n_sets=600, n_hits=12, seg_len=150, branching=3, genome_len=3000, seed=51, variety=True)
    full = pkg.RPAPredictionModel(ds.tax, alphabet=0)
    full.load_store(pkg.STORE_QUERY, ds.query_letters, ds.query_offsets)
    full.load_store(pkg.STORE_DB, ds.ref_letters, ds.ref_offsets)
    want_rows, want_stats = full.predict(ds.batch)
    b = ds.batch
    roff = ds.ref_offsets.astype(np.int64)
    rlen = np.diff(roff)
    TILE = 256
    m = pkg.RPAPredictionModel(ds.tax, alphabet=0)
    m.load_store(pkg.STORE_QUERY, ds.query_letters, ds.query_offsets)
    m.reserve_windows(pkg.STORE_DB, rlen)
    # the tiles under every hit's range extended by its record set's query range
    set_of = np.repeat(np.arange(len(b["set_offsets"]) - 1), np.diff(b["set_offsets"].astype(np.int64)))
    span = np.array([int(b["qstop"][lo:hi].max()) - int(b["qstart"][lo:hi].min()) if hi > lo else 0
                     for lo, hi in zip(b["set_offsets"][:-1], b["set_offsets"][1:])], np.int64)
    lo = np.minimum(b["rstart"], b["rstop"]).astype(np.int64) - 1 - span[set_of]
    hi = np.maximum(b["rstart"], b["rstop"]).astype(np.int64) + span[set_of]
    seq = b["ref_seq"].astype(np.int64)
    ok = seq < len(rlen)
    lo, hi, seq = np.maximum(lo[ok], 0), np.minimum(hi[ok], rlen[seq[ok]]), seq[ok]
    ok = hi > lo
    lo, hi, seq = lo[ok], hi[ok], seq[ok]
    tiles = {}
    for s_, a, e in zip(seq, lo // TILE, (hi - 1) // TILE):
        tiles.setdefault(int(s_), set()).update(range(int(a), int(e) + 1))

    def runs(ts):
        ts = sorted(ts); out = []; a = p_ = ts[0]
        for t in ts[1:]:
            if t != p_ + 1: out.append((a, p_ + 1)); a = t
            p_ = t
        out.append((a, p_ + 1))
        return out

    def letters_of(s_, a, e):
        return ds.ref_letters[roff[s_] + a: roff[s_] + e]

    # first call: every second run of every sequence; second call: the whole hull of each sequence's tiles (covers and replaces them)
    first_call, second_call = [], []
    for s_, ts in tiles.items():
        rr = runs(ts)
        for (a, e) in rr[::2]:
            first_call.append((s_, a * TILE, min(e * TILE, int(rlen[s_]))))
        second_call.append((s_, rr[0][0] * TILE, min(rr[-1][1] * TILE, int(rlen[s_]))))
    m.append_windows(pkg.STORE_DB, [x[0] for x in first_call], [x[1] for x in first_call], [letters_of(*x) for x in first_call])
    info1 = m.store_info(pkg.STORE_DB)
    assert info1["windows"] == len(first_call) and info1["live_residues"] == sum(x[2] - x[1] for x in first_call)
    if any(len(runs(ts)) > 1 for ts in tiles.values()):
        with pytest.raises(pkg.RpaError) as ei:   # some segment lies in a run that is not resident yet
            m.predict(ds.batch)
        assert ei.value.code == 1 and "resident windows" in str(ei.value)
    s0, a0, e0 = first_call[0]
    if e0 - a0 > 2:
        with pytest.raises(pkg.RpaError) as ei:   # partial overlap with a resident window
            m.append_windows(pkg.STORE_DB, [s0], [a0 + 1], [letters_of(s0, a0 + 1, min(int(rlen[s0]), e0 + 1))] if e0 < rlen[s0] else [letters_of(s0, a0 + 1, e0)])
        assert ei.value.code == 1
    m.append_windows(pkg.STORE_DB, [x[0] for x in second_call], [x[1] for x in second_call], [letters_of(*x) for x in second_call])
    info2 = m.store_info(pkg.STORE_DB)
    assert info2["windows"] == len(second_call) and info2["live_residues"] == sum(x[2] - x[1] for x in second_call)
    assert info2["n_resident"] == len(tiles) and info2["residues"] >= info1["residues"] + info2["live_residues"]
    rows, stats = m.predict(ds.batch)
    assert rows.tobytes() == want_rows.tobytes() and stats.tobytes() == want_stats.tobytes()
    # asynchronous jobs read the same windows
    job = m.submit(ds.batch)
    rows, stats = job.wait()
    assert rows.tobytes() == want_rows.tobytes()
    # reset: nothing resident, the arena is refilled from its start
    m.reset_store(pkg.STORE_DB)
    info3 = m.store_info(pkg.STORE_DB)
    assert info3["windows"] == 0 and info3["residues"] == 0 and info3["n_resident"] == 0
    with pytest.raises(pkg.RpaError):
        m.predict(ds.batch)
    m.append_windows(pkg.STORE_DB, [x[0] for x in second_call], [x[1] for x in second_call], [letters_of(*x) for x in second_call])
    rows, stats = m.predict(ds.batch)
    assert rows.tobytes() == want_rows.tobytes() and stats.tobytes() == want_stats.tobytes()
    assert m.store_info(pkg.STORE_DB)["residues"] <= info2["residues"]


def test_large_record_sets(pkg, oracle, synth):
    """Record sets far larger than a wavefront (300 hits) and a batch mixing set sizes 2..300."""
    ds = synth.make_dataset(n_sets=40, n_hits=300, seg_len=120, branching=4, genome_len=1500, seed=31, variety=True)
    _engine_vs_oracle(pkg, oracle, synth, ds, 0)
    _engine_vs_oracle(pkg, oracle, synth, ds, 0, speculation=(16, 5))


def test_empty_and_degenerate_batches(pkg):
    class T:
        parent = np.zeros(1, np.uint32)
        left = np.array([1], np.uint32)
        right = np.array([2], np.uint32)
        depth = np.zeros(1, np.uint8)
    model = pkg.RPAPredictionModel(T(), alphabet=pkg.DNA)
    model.load_store(pkg.STORE_QUERY, LETTERS_DNA[np.zeros(10, np.uint8)], np.array([0, 10], np.uint64))
    model.load_store(pkg.STORE_DB, LETTERS_DNA[np.zeros(10, np.uint8)], np.array([0, 10], np.uint64))
    z32 = np.zeros(0, np.uint32)
    empty = dict(set_offsets=np.zeros(1, np.uint32), query_seq=z32, query_len=z32, qstart=z32, qstop=z32, rstart=z32, rstop=z32,
                 score=np.zeros(0, np.float32), identities=z32, alnlen=z32, ref_seq=z32, node=z32, filtered=np.zeros(0, np.uint8))
    rows, stats = model.predict(empty)
    assert len(rows) == 0
    # record sets without any record (set_offsets repeats): every one comes back unclassified with the stale-ival flag
    three = dict(empty)
    three["set_offsets"] = np.zeros(4, np.uint32)
    three["query_seq"] = np.zeros(3, np.uint32)
    three["query_len"] = np.array([10, 10, 10], np.uint32)
    rows, stats = model.predict(three)
    assert list(rows["flags"]) == [1, 1, 1] and list(rows["end"]) == [10, 10, 10] and list(stats["n"]) == [0, 0, 0]


def _subset_batch(ds, keep_counts):
    """The batch of `ds` with only the first keep_counts[s] records of record set s (records are independent)."""
    b = ds.batch
    so = b["set_offsets"].astype(np.int64)
    idx = np.concatenate([np.arange(so[s], so[s] + min(int(k), int(so[s + 1] - so[s]))) for s, k in enumerate(keep_counts)])
    out = {}
    for k, v in b.items():
        if k == "set_offsets":
            out[k] = np.concatenate([[0], np.cumsum([min(int(c), int(so[s + 1] - so[s])) for s, c in enumerate(keep_counts)])]).astype(np.uint32)
        elif k in ("query_seq", "query_len"):
            out[k] = v.copy()
        else:
            out[k] = v[idx].copy()
    return out


@pytest.mark.parametrize("layout", [None, "linear"])
def test_row_layouts_tiled_and_linear(pkg, oracle, synth, layout, monkeypatch):
    """The hit rows live either tiled (64 record sets side by side, every tile padded to its largest set) or linear.  A
    batch of evenly sized sets takes the tiled layout; one whose padding would dwarf the rows (a 300-hit set in every
    tile of 2..5-hit sets) falls back to linear by itself; RPA_ROW_LAYOUT=linear forces it.  Same rows either way."""
    if layout:
        monkeypatch.setenv("RPA_ROW_LAYOUT", layout)
    import copy
    ds = synth.make_dataset(n_sets=4096, n_hits=300, seg_len=48, branching=3, genome_len=1200, seed=41, variety=True)
    rng = np.random.default_rng(5)
    keep = rng.integers(2, 6, ds.n_sets)
    keep[::64] = 300                      # one big set per tile: 64 x 300 slots per tile against ~500 rows
    skew = copy.copy(ds)
    skew.batch = _subset_batch(ds, keep)
    _engine_vs_oracle(pkg, oracle, synth, skew, 0)
    even = copy.copy(ds)
    even.batch = _subset_batch(ds, np.full(ds.n_sets, 7))
    _engine_vs_oracle(pkg, oracle, synth, even, 0, speculation=(3, 2))


@pytest.mark.parametrize("parts", ["2", "5"])
def test_batch_cut_into_parts(pkg, oracle, synth, parts, monkeypatch):
    """RPA_PARTS=n: the record sets of a batch go through n worker engines (own streams, working sets and host threads,
    started one after the other so that one part's placement rounds run underneath another's alignment kernel).  Record
    sets are independent: rows and counters must be those of the whole batch, for host and for device-resident batches."""
    monkeypatch.setenv("RPA_PARTS", parts)
    ds = synth.make_dataset(n_sets=2500, n_hits=14, seg_len=120, branching=4, genome_len=3000, seed=51, variety=True)
    rows, stats = _engine_vs_oracle(pkg, oracle, synth, ds, 0)
    dp = synth.make_dataset(n_sets=700, n_hits=12, seg_len=67, branching=3, genome_len=1500, seed=52, alphabet=1,
                            level_rate=0.06, query_rate=0.08, variety=True)  # every worker has its own protein workspace / work counter
    _engine_vs_oracle(pkg, oracle, synth, dp, 1)
    import torch
    model = pkg.RPAPredictionModel(ds.tax, alphabet=0)
    model.load_store(pkg.STORE_QUERY, ds.query_letters, ds.query_offsets)
    model.load_store(pkg.STORE_DB, ds.ref_letters, ds.ref_offsets)
    dev = {k: torch.from_numpy(np.ascontiguousarray(v).view(np.int32) if v.dtype == np.uint32 else np.ascontiguousarray(v)).cuda()
           for k, v in ds.batch.items()}
    out = torch.zeros((ds.n_sets, 8), dtype=torch.int32, device="cuda")
    st = torch.zeros((ds.n_sets, 4), dtype=torch.int32, device="cuda")
    model.profile(enable=True, reset=True)
    model.predict(dev, mem_kind=pkg.MEM_DEVICE, out=out, stats=st)
    torch.cuda.synchronize()
    assert out.cpu().numpy().tobytes() == rows.view(np.int32).reshape(-1, 8).tobytes()
    assert st.cpu().numpy().tobytes() == stats.view(np.int32).reshape(-1, 4).tobytes()
    prof = model.profile()
    assert prof["demanded"] == int(stats["pass0"].sum() + stats["pass1"].sum() + stats["pass2"].sum())


def test_bad_batches_rejected(pkg, oracle, synth):
    """What must not reach a kernel through the C ABI: a taxonomy node index outside the taxonomy, set offsets that
    decrease or do not end at n_records (host and device batches alike: RPA_ERR_INVALID, no page fault); and the
    reference's order of events around the perfect-hit shortcut: the query range is fetched first (:415), so an unknown
    query sequence aborts the run even for a set the shortcut would have finished without looking at it."""
    import torch
    ds = synth.make_dataset(n_sets=92, n_hits=6, seg_len=120, branching=2, seed=9, variety=True)
    model = pkg.RPAPredictionModel(ds.tax, alphabet=pkg.DNA)
    model.load_store(pkg.STORE_QUERY, ds.query_letters, ds.query_offsets)
    model.load_store(pkg.STORE_DB, ds.ref_letters, ds.ref_offsets)
    rows, _ = model.predict(ds.batch)
    shortcut = np.nonzero(rows["flags"] & 2)[0]
    assert len(shortcut), "the variety generator plants perfect-hit sets"
    tax = oracle.TaxArrays(ds.tax.parent, ds.tax.left, ds.tax.right, ds.tax.depth)
    q = oracle.StoreArrays(synth.encode(ds.query_letters, 0), ds.query_offsets)
    db = oracle.StoreArrays(synth.encode(ds.ref_letters, 0), ds.ref_offsets)
    bad = {k: v.copy() for k, v in ds.batch.items()}
    bad["query_seq"][int(shortcut[0])] = 10 ** 6            # SequenceNotFound before the shortcut is even considered
    rc, _, _ = oracle.predict_batch(tax, q, db, 0, bad["set_offsets"], bad["query_seq"], bad["query_len"], synth.records_aos(bad, oracle.RECORD_DTYPE))
    assert rc == 2                                            # ORC_ERR_SEQ_ID
    with pytest.raises(pkg.RpaError) as ei:
        model.predict(bad)
    assert ei.value.code == 4                                 # RPA_ERR_SEQ_ID

    def as_dev(b):
        return {k: torch.from_numpy(np.ascontiguousarray(v).view(np.int32) if v.dtype == np.uint32 else np.ascontiguousarray(v)).cuda() for k, v in b.items()}

    for kind in ("node", "offsets_decrease", "offsets_end"):
        bad = {k: v.copy() for k, v in ds.batch.items()}
        if kind == "node": bad["node"][17] = len(ds.tax.parent) + 5
        elif kind == "offsets_decrease": bad["set_offsets"][40] = bad["set_offsets"][39] - 1
        else: bad["set_offsets"][-1] += 3
        for device in (False, True):
            with pytest.raises(pkg.RpaError) as ei:
                if device:
                    out = torch.zeros((ds.n_sets, 8), dtype=torch.int32, device="cuda")
                    model.predict(as_dev(bad), mem_kind=pkg.MEM_DEVICE, out=out)
                else:
                    model.predict(bad)
            assert ei.value.code == 1, (kind, device)
    rows2, _ = model.predict(ds.batch)                        # the engine is still usable
    assert rows2.tobytes() == rows.tobytes()


def test_async_jobs_equal_synchronous_calls(pkg, oracle, synth):
    """rpa_submit_batch / rpa_wait: two jobs in flight on two streams with their own working sets.  Different batches
    (sizes, seeds, with and without masked records) submitted back to back, waited for out of order: every job's rows and
    counters equal the synchronous call's; an invalid batch fails its own job only; a store append waits for the jobs."""
    ds = synth.make_dataset(n_sets=3000, n_hits=12, seg_len=160, branching=2, genome_len=3000, seed=21, variety=True)
    model = pkg.RPAPredictionModel(ds.tax, alphabet=pkg.DNA)
    model.load_store(pkg.STORE_QUERY, ds.query_letters, ds.query_offsets)
    model.load_store(pkg.STORE_DB, ds.ref_letters, ds.ref_offsets)
    sharding = importlib.import_module("taxator_tk_amd.sharding")
    cuts = [0, 1, 700, 701, 1500, 2999, 3000]
    parts = [sharding.slice_batch(ds.batch, a, b) for a, b in zip(cuts[:-1], cuts[1:])]
    want = [model.predict(p) for p in parts]
    jobs = [model.submit(p) for p in parts]
    for k in (1, 0, 3, 2, 5, 4):
        rows, st = jobs[k].wait()
        assert rows.tobytes() == want[k][0].tobytes() and st.tobytes() == want[k][1].tobytes(), k
    bad = {k: (v.copy() if v is not None else None) for k, v in parts[2].items()}
    bad["node"][5] = 10 ** 7
    j_ok, j_bad, j_ok2 = model.submit(parts[1]), model.submit(bad), model.submit(parts[3])
    with pytest.raises(pkg.RpaError) as ei:
        j_bad.wait()
    assert ei.value.code == 1
    assert j_ok.wait()[0].tobytes() == want[1][0].tobytes() and j_ok2.wait()[0].tobytes() == want[3][0].tobytes()
    full = model.submit(ds.batch)
    rows, _ = full.wait()
    tax = oracle.TaxArrays(ds.tax.parent, ds.tax.left, ds.tax.right, ds.tax.depth)
    q = oracle.StoreArrays(synth.encode(ds.query_letters, 0), ds.query_offsets)
    db = oracle.StoreArrays(synth.encode(ds.ref_letters, 0), ds.ref_offsets)
    rc, exp, _ = oracle.predict_batch(tax, q, db, 0, ds.batch["set_offsets"], ds.batch["query_seq"], ds.batch["query_len"], synth.records_aos(ds.batch, oracle.RECORD_DTYPE))
    assert rc == 0 and rows.tobytes() == exp.tobytes()

