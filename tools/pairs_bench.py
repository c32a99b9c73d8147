"""tools/pairs_bench.py -- in-kernel throughput of the pair-level alignment kernels over a sweep of sequence lengths.

    python tools/pairs_bench.py [dna|protein] [len ...]

For each length L: random pairs (B = A with ~15 % edits -- PAIRS_MUT sets the substitution rate -- so lengths differ slightly), sized to ~2e11 cells (DNA) per
launch, run through rpa_align_pairs with the engine's event timing on.  Prints alignments/s and cell updates/s inside
the alignment kernels (rpa_profile.align_ms).  Measurement tool only; parity lives in tests/.
"""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def make_pairs(rng, alphabet, L, n_pairs):
    letters_set = np.frombuffer(b"ACGT" if alphabet == "dna" else b"ACDEFGHIKLMNPQRSTVWY", np.uint8)
    a = letters_set[rng.integers(0, len(letters_set), (n_pairs, L))]
    b = a.copy()
    mut = rng.random((n_pairs, L)) < float(os.environ.get("PAIRS_MUT", "0.12"))  # (substitutions; PAIRS_MUT=0.4 ~ 30 % edits)
    b[mut] = letters_set[rng.integers(0, len(letters_set), int(mut.sum()))]
    # ragged lengths: trim 0..3 % off the end of B
    lens_b = L - rng.integers(0, max(1, L // 32), n_pairs)
    seqs, offs = [], [0]
    for i in range(n_pairs):
        seqs.append(a[i]); offs.append(offs[-1] + L)
        seqs.append(b[i, :lens_b[i]]); offs.append(offs[-1] + int(lens_b[i]))
    letters = np.concatenate(seqs)
    pa = np.arange(0, 2 * n_pairs, 2, dtype=np.uint32)
    return letters, np.array(offs, np.uint64), pa, pa + 1


def sweep(rpa, synth, alphabet, lens, mut=None, calls=3):
    """[{length, pairs, alignments_per_s, cell_equivalents_per_s, ms_per_call}] for launches of random pairs of each length through
    rpa_align_pairs (in-kernel time); bench.py reports a short one as secondary.dna_pair_rates."""
    if mut is not None: os.environ["PAIRS_MUT"] = str(mut)
    tax = synth.MockTaxonomy(2)
    model = rpa.RPAPredictionModel(tax, rpa.DNA if alphabet == "dna" else rpa.PROTEIN)
    model.profile(enable=True)
    rng = np.random.default_rng(7)
    budget = 2e11 if alphabet == "dna" else 4e9
    out = []
    for L in lens:
        n_pairs = int(min(400000, max(256, budget / (L * L))))
        letters, offs, pa, pb = make_pairs(rng, alphabet, L, n_pairs)
        model.align_pairs(letters, offs, pa, pb)  # warm-up
        model.profile(reset=True)
        for _ in range(calls):
            model.align_pairs(letters, offs, pa, pb)
        p = model.profile()
        s = p["align_ms"] / 1e3
        out.append({"length": L, "pairs": n_pairs, "alignments_per_s": p["pairs"] / s, "cell_equivalents_per_s": p["cells"] / s, "ms_per_call": p["align_ms"] / calls})
    model.close()
    return out


def main():
    rpa = importlib.import_module("taxator-tk_amd")
    synth = importlib.import_module("taxator-tk_amd.synth")
    alphabet = sys.argv[1] if len(sys.argv) > 1 else "dna"
    lens = [int(x) for x in sys.argv[2:]] or ([100, 200, 256, 300, 400, 600, 1000, 1500, 2048, 3000, 5000, 10000, 16000] if alphabet == "dna" else [67, 200, 500])
    for r in sweep(rpa, synth, alphabet, lens):
        print(f"{alphabet} L={r['length']:6d} pairs/launch={r['pairs']:7d}  {r['alignments_per_s']:12.4g} alignments/s  {r['cell_equivalents_per_s'] / 1e12:7.2f} T cells/s"
              f"  ({r['ms_per_call']:.2f} ms per call)", flush=True)


if __name__ == "__main__":
    main()
