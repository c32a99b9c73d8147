#!/bin/bash
# tools/rank_parts.sh -- what the parts of k_rank_records cost: the kernel alone (RPA_DEBUG_RANK_ONLY) on config 2's batch, the
# full build against timing builds that leave one part out (-DRPA_STUB_RANK | LABELS | SEG | WRITEOUT; their rows are not usable)
# Build the timing libraries first (here, not on the GPU box): for V in RANK LABELS SEG WRITEOUT; do tools/build_variant.sh stub_$V -DRPA_STUB_$V; done
R=$PWD; cd /tmp && export TMPDIR=/tmp
for V in "" stub_RANK stub_LABELS stub_SEG stub_WRITEOUT; do
  rm -rf /tmp/rp$V
  LIB=$R/taxator-tk_amd/librpa_hip.so; [ -n "$V" ] && LIB=$R/build_ab/$V/librpa_hip.so
  RPA_LIB_PATH=$LIB rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp$V -o p -- python3 $R/tools/rank_parts.py > /dev/null 2>&1
  python3 - "$V" /tmp/rp$V/p_kernel_stats.csv <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[2])):
    if "k_rank_records" in r["Name"]:
        print("%-16s k_rank_records: %s calls, avg %.1f us" % (sys.argv[1] or "full", r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
