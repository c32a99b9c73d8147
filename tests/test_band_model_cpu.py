"""The exactness argument of the banded strip kernel (k_myers_band), checked on a bit-level model with small words
(tools/band_model.py) against plain Levenshtein: D' >= d always; d <= k = 2 min(h) + delta  <=>  the band test passes,
and then D' == d; the add carry into a strip equals the HN bit handed down (asserted on every column)."""
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_band_model():
    spec = importlib.util.spec_from_file_location("band_model", os.path.join(ROOT, "tools", "band_model.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert mod.check(trials=800, seed=3) == 800
    assert mod.check_window(trials=1500, seed=4) == 1500              # K1: five words, in phase with the text groups
    assert mod.check_window_phase(trials=2000, seed=6) == 2000       # K1: four words, half a group out of phase
    assert mod.check_window_phase3(trials=3000, seed=8) == 3000      # K1: ... with a three-word middle (the default)
    assert mod.check_window_ring(trials=3000, seed=10) == 3000       # K1r: 4 ... 10 words, only the window alive, any pattern length
