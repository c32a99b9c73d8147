#!/usr/bin/env python3
"""tools/make_traffic_json.py TAG N_PREDICT_CALLS -- profiles/TAG_traffic.json from gpurun_out/TAG_pmc_{fetch,write}.txt
(the summaries tools/profile_round.sh / tools/pmc_traffic.sh write): HBM bytes of the alignment kernels per launch and per predict call,
2 x FETCH_SIZE + WRITE_SIZE as MI355X_MICROARCH.md prescribes for gfx950 (FETCH_SIZE counts 64 B per 128-B request)."""
import json, re, sys
tag, calls = sys.argv[1], int(sys.argv[2])
def read(path, counter):
    out, cur = {}, None
    for line in open(path):
        m = re.search(r"(rpa::.+?)\s+dispatches=(\d+)", line)  # (pmc_summary.py keeps the last 42 characters of a kernel's name: "void " may be cut)
        if m: cur = m.group(1); out[cur] = {"dispatches": int(m.group(2))}
        m = re.search(counter + r"\s+sum=(\S+)", line)
        if m and cur: out[cur]["sum_kb"] = float(m.group(1))
    return out
f = read("gpurun_out/%s_pmc_fetch.txt" % tag, "FETCH_SIZE")
w = read("gpurun_out/%s_pmc_write.txt" % tag, "WRITE_SIZE")
ker = sorted(k for k in f if "k_myers" in k or "k_nw_protein" in k)
fetch = sum(f[k]["sum_kb"] for k in ker); write = sum(w[k]["sum_kb"] for k in ker if k in w)
disp = sum(f[k]["dispatches"] for k in ker)
pk = [k for k in f if "k_pack_letters" in k]
# what k_pack_letters reads / writes ONCE per process in each configuration of bench.py (letters in: query + reference store, one
# byte per residue; packed words out: 4 bit per nucleotide, 8 bit per amino acid) -- the counters above are sums over the run
expected = ("config 4: 33.5 MB of query + 6.1 MB of reference letters in, the same 39.6 MB out (8 bit per residue)" if "_c4" in tag else
            "config 5: 125 MB of query + 164 MB of reference letters in, 144 MB of packed words out (4 bit per nucleotide)" if "_c5" in tag else
            "config 2: 200 MB of query + 16 MB of reference letters in, 108 MB of packed words out (4 bit per nucleotide)")
total = (2 * fetch + write) * 1024.0
json.dump({"kernels": ker, "dispatches": disp, "predict_calls": calls, "FETCH_SIZE_KB_sum": fetch, "WRITE_SIZE_KB_sum": write,
           "correction": "gfx950: FETCH_SIZE counts 64 B per 128-B request -> x2 (MI355X_MICROARCH.md, HBM); WRITE_SIZE as is. "
                         "Calibration in the same run: k_pack_letters FETCH_SIZE %.1f MB / WRITE_SIZE %.1f MB over %d predict call(s)' worth of store loads (%s)." %
                         (sum(f[k]["sum_kb"] for k in pk) / 1024, sum(w[k]["sum_kb"] for k in pk if k in w) / 1024, calls, expected),
           "traffic_bytes_per_launch": total / disp, "traffic_bytes_per_predict_call": total / calls,
           "command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (separate passes) --kernel-include-regex rpa:: -- python3 bench.py [--config cN] --steps 2 --warmup 1 --no-cpu-baseline"},
          open("profiles/%s_traffic.json" % tag, "w"), indent=1)
print(open("profiles/%s_traffic.json" % tag).read())
