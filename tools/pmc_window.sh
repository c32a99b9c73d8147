#!/bin/bash
# tools/pmc_window.sh -- the ring-window kernel (K1r) under counters: vector instructions and instruction fetches per dispatch of
# tools/pairs_bench.py dna 512 1024 (400 k / 190 k pairs, ~12 % edits), with the dispatch durations of the same pass -> gpurun_out/pmc_window.txt
R=$PWD; cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmcw
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_IFETCH SQ_WAVES --kernel-include-regex "k_myers_window" --output-format csv -d /tmp/pmcw -o pmc -- python3 $R/tools/pairs_bench.py dna 512 1024 > /dev/null 2>&1
python3 - /tmp/pmcw/pmc_counter_collection.csv /tmp/pmcw/pmc_kernel_trace.csv > $R/gpurun_out/pmc_window.txt <<'PY'
import csv, sys, collections
ctr = collections.defaultdict(dict)
for r in csv.DictReader(open(sys.argv[1])):
    ctr[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
    ctr[int(r["Dispatch_Id"])]["name"] = r["Kernel_Name"].split("(")[0][-24:]
dur = {}
for r in csv.DictReader(open(sys.argv[2])):
    dur[int(r["Dispatch_Id"])] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
for d in sorted(ctr):
    c = ctr[d]
    ns = dur.get(d, 0)
    if not ns: continue
    print("%-24s dispatch %4d  %8.1f us  VALU %.4g  LDS %.4g  SALU %.4g  IFETCH %.4g  waves %.4g  -> %.3f VALU wave-instr/ns/SIMD (1024 SIMDs), %.1f VALU per wave" % (
        c["name"], d, ns / 1e3, c.get("SQ_INSTS_VALU", 0), c.get("SQ_INSTS_LDS", 0), c.get("SQ_INSTS_SALU", 0), c.get("SQ_IFETCH", 0), c.get("SQ_WAVES", 0),
        c.get("SQ_INSTS_VALU", 0) / ns / 1024.0, c.get("SQ_INSTS_VALU", 0) / max(c.get("SQ_WAVES", 1), 1)))
PY
cat $R/gpurun_out/pmc_window.txt
