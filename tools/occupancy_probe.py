"""tools/occupancy_probe.py -- does k_myers_short gain from more resident waves?  Pattern lengths 128/160/192/224/256 nt
(NW = 4..8 words: 20..40 KB of LDS per workgroup, i.e. 8/6/5/4/4 workgroups per CU) at the same instruction mix per word;
prints word-column updates per second (pairs x NW x text length / kernel time)."""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as graft
rpa = graft.load_package()
synth = importlib.import_module("taxator_tk_amd.synth")

tax = synth.MockTaxonomy(2)
model = rpa.RPAPredictionModel(tax, rpa.DNA)
model.profile(enable=True)
rng = np.random.default_rng(11)
acgt = np.frombuffer(b"ACGT", np.uint8)
n_pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
for L in (128, 160, 192, 224, 256):
    a = acgt[rng.integers(0, 4, (n_pairs, L))]
    b = a.copy()
    mut = rng.random((n_pairs, L)) < 0.15
    b[mut] = acgt[rng.integers(0, 4, int(mut.sum()))]
    letters = np.ascontiguousarray(np.stack([a, b], axis=1).reshape(-1))
    offs = (np.arange(2 * n_pairs + 1, dtype=np.uint64) * L)
    pa = np.arange(0, 2 * n_pairs, 2, dtype=np.uint32)
    model.align_pairs(letters, offs, pa, pa + 1)
    model.profile(reset=True)
    for _ in range(2):
        model.align_pairs(letters, offs, pa, pa + 1)
    p = model.profile()
    s = p["align_ms"] / 1e3
    nw = L // 32
    print("L=%d NW=%d LDS/block=%d KB: %.2f T word-columns/s, %.1f T cells/s (%.2f ms per launch)"
          % (L, nw, 5 * nw, p["pairs"] * nw * L / s / 1e12, p["cells"] / s / 1e12, p["align_ms"] / 2), flush=True)
