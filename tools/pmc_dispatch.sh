#!/bin/bash
# tools/pmc_dispatch.sh TAG "COUNTERS" KERNEL_REGEX [bench args] -- one --pmc pass of a synchronous bench run, PER DISPATCH of the kernels that match
# (tools/pmc_summary.py adds the dispatches of a kernel up; the rounds of one step differ by a factor of ten)  -> gpurun_out/TAG_pmc_dispatch.txt
TAG=$1; P=$2; K=$3; shift 3
R=$PWD; cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmcd_$TAG
rocprofv3 --kernel-trace --pmc $P --kernel-include-regex "$K" --output-format csv -d /tmp/pmcd_$TAG -o pmc -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --sync "$@" > /dev/null 2> $R/gpurun_out/${TAG}_pmc_dispatch.err
python3 - /tmp/pmcd_$TAG/pmc_counter_collection.csv > $R/gpurun_out/${TAG}_pmc_dispatch.txt <<'PY'
import csv, sys, collections
d = collections.OrderedDict()
for r in csv.DictReader(open(sys.argv[1])):
    key = (int(r["Dispatch_Id"]), r["Kernel_Name"].split("(")[0][-36:])
    d.setdefault(key, {})[r["Counter_Name"]] = float(r["Counter_Value"])
names = sorted({c for v in d.values() for c in v})
print("dispatch kernel " + " ".join(names))
for (i, k), v in sorted(d.items()):
    print(i, k, " ".join("%.4g" % v.get(c, 0) for c in names))
PY
