#!/usr/bin/env python3
"""tools/kt_inflight.py FILE [T0_FRACTION] -- the jobs-in-flight regime from a rocprofv3 --kernel-trace CSV (rpa:: kernels of a
pipelined bench run): over the steady-state window (the middle of the trace) the share of time with 0 / 1 / 2 / ... kernels
running, every kernel's summed duration per step against its solo duration, and who runs beside the dominant kernel."""
import csv, sys, collections
rows = list(csv.reader(open(sys.argv[1])))
hdr = rows[0]
ki = hdr.index("Kernel_Name"); si = hdr.index("Start_Timestamp"); ei = hdr.index("End_Timestamp")
ev = sorted((int(r[si]), int(r[ei]), r[ki].split("rpa::")[1].split("(")[0].split("<")[0]) for r in rows[1:] if len(r) > max(ki, si, ei) and "rpa::" in r[ki])
t_lo, t_hi = ev[0][0], max(e[1] for e in ev)
a = t_lo + (t_hi - t_lo) * 0.35; b = t_lo + (t_hi - t_lo) * 0.85
win = [(max(s, a), min(e, b), k) for s, e, k in ev if e > a and s < b]
n_rank = sum(1 for s, e, k in ev if k == "k_rank_records" and a <= s < b)
print("window %.2f ms, %d steps begin inside it -> %.3f ms per step" % ((b - a) / 1e6, n_rank, (b - a) / 1e6 / max(n_rank, 1)))
pts = []
for s, e, k in win:
    pts.append((s, 1, k)); pts.append((e, -1, k))
pts.sort()
hist = collections.Counter(); run = collections.Counter(); beside = collections.Counter(); cur = 0; last = a
for t, d, k in pts:
    hist[cur] += t - last
    if run["k_myers_short_band4"]:
        for kk, c in run.items():
            if c and kk != "k_myers_short_band4": beside[kk] += t - last
        if sum(run.values()) == run["k_myers_short_band4"]: beside["(only band4: %d)" % run["k_myers_short_band4"]] += t - last
    last = t; cur += d; run[k] += d
tot = sum(hist.values())
print("kernels running at once:", "  ".join("%d: %.1f %%" % (c, 100.0 * v / tot) for c, v in sorted(hist.items())))
dur = collections.Counter()
for s, e, k in win: dur[k] += e - s
print("summed duration per step (ms):", "  ".join("%s %.3f" % (k, v / 1e6 / max(n_rank, 1)) for k, v in dur.most_common()))
print("while a band4 kernel runs, beside it (share of the window):", "  ".join("%s %.1f %%" % (k, 100.0 * v / tot) for k, v in beside.most_common()))
