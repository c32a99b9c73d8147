#!/usr/bin/env python3
"""tools/kt_timeline.py FILE -- per-dispatch timeline of the LAST bench step from a rocprofv3 --kernel-trace CSV that was
filtered to the rpa:: kernels (header line + `grep rpa::` of the full kernel_trace.csv)."""
import csv, sys
rows = list(csv.reader(open(sys.argv[1])))
hdr = rows[0]
ki = hdr.index("Kernel_Name"); si = hdr.index("Start_Timestamp"); ei = hdr.index("End_Timestamp")
data = sorted((int(r[si]), int(r[ei]), r[ki].split("rpa::")[1].split("(")[0]) for r in rows[1:]
              if len(r) > max(ki, si, ei) and "rpa::" in r[ki] and "k_pack" not in r[ki])
starts = [i for i, d in enumerate(data) if d[2].startswith("k_rank_records")]
last = data[starts[-1]:]
t0 = last[0][0]; prev = t0; tot = {}
for s, e, k in last:
    print("%9.1f us  gap %6.1f  dur %8.1f  %s" % ((s - t0) / 1e3, (s - prev) / 1e3, (e - s) / 1e3, k))
    tot[k] = tot.get(k, 0) + (e - s) / 1e3; prev = e
print("step total %.3f ms" % ((last[-1][1] - t0) / 1e6))
for k, v in sorted(tot.items(), key=lambda x: -x[1]):
    print("  %-28s %8.1f us" % (k, v))
