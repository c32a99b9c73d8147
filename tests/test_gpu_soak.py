"""Differential soaks as part of the GPU suite (bounded, fixed seeds): tools/soak.py -- 100 seeded batches of random shape
(DNA 40-1500 nt, protein 20-400 aa, 2-40 hits, the edge cases of the variety generator, random speculation chunks,
live-list thresholds, split rounds, batch parts, and the exact bands forced on / tightened until every pair is retried /
switched off) through rpa_predict_batch against the CPU oracle, every row and counter; tools/soak_pairs.py -- random pairs
(nucleotide 1-18 000 nt through every Myers kernel, protein 1-800 aa through both protein kernels) through
rpa_align_pairs against the oracle's pair functions, counts and f32 bit patterns."""
import importlib.util
import os

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _tool(name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, "tools", name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_soak_record_sets():
    n, sets, aborts = _tool("soak").run(n_seeds=100, verbose=False)
    assert n == 100 and sets > 10_000


def test_soak_pairs():
    n, pairs = _tool("soak_pairs").run(n_batches=260, verbose=False)
    assert n == 260 and pairs >= 150_000
