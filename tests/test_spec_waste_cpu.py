"""tools/spec_waste.py (CPU only): its Python replay of the reference's three passes must demand exactly the alignments the
oracle's trace lists (the tool asserts that per record set), and the request policies it plays over the replayed loops
must order as they do on the GPU: the plain four-hit chunks ask for more than the same chunks with the paused loop's
thresholds, and those for more than whole loops that end at the first hit's assumed score cut-off."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_replay_matches_the_trace_and_policies_order():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "spec_waste.py"), "c4", "40"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    ratio = {}
    for line in r.stdout.splitlines():
        m = re.match(r"\s+(\S+)\s+S1=(\d+) S2=(\d+)\s+computed/demanded ([0-9.]+)", line)
        if m:
            ratio[(m.group(1), int(m.group(2)), int(m.group(3)))] = float(m.group(4))
    assert ("plain", 4, 10) in ratio and ("filtered", 4, 10) in ratio and ("first-score", 0, 10) in ratio, r.stdout
    assert ratio[("plain", 4, 10)] > ratio[("filtered", 4, 10)] > ratio[("first-score", 0, 10)] >= 1.0, ratio
    assert ratio[("cautious", 0, 10)] < 1.05, ratio
