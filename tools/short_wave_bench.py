"""tools/short_wave_bench.py -- K1w (a pattern word per lane) against k_myers_short (a pair per lane) on launches of few pairs:
rpa_align_pairs on N pairs of ~200 nt, alignment-kernel time by the engine's events (align_ms), RPA_SHORT_WAVE_MAX = 0 / default."""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pairs_bench import make_pairs
synth = importlib.import_module("taxator-tk_amd.synth")
for knob in ("0", "1000000"):
    os.environ["RPA_SHORT_WAVE_MAX"] = knob
    rpa = importlib.import_module("taxator-tk_amd")
    tax = synth.MockTaxonomy(2)
    model = rpa.RPAPredictionModel(tax, rpa.DNA)
    rng = np.random.default_rng(7)
    for n in (8, 64, 512, 4096, 16384, 65536, 131072):
        letters, offs, pa, pb = make_pairs(rng, "dna", 200, n)
        model.align_pairs(letters, offs, pa, pb)
        model.profile(enable=True, reset=True)
        for _ in range(5):
            model.align_pairs(letters, offs, pa, pb)
        p = model.profile()
        print("RPA_SHORT_WAVE_MAX=%s: %6d pairs: %.1f us per launch (align_ms / launches), %d launches" % (knob, n, p["align_ms"] / max(1, p["launches"]) * 1e3, p["launches"]), flush=True)
    del model
