#!/bin/bash
# tools/kstats.sh TAG [bench args] -- rocprofv3 --kernel-trace --stats of a short bench run -> gpurun_out/TAG_kernel_stats.csv
TAG=$1; shift
R=$PWD; cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ks_$TAG
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_$TAG -o p -- python3 $R/bench.py --no-cpu-baseline --sync "$@" > $R/gpurun_out/${TAG}_bench_profiled.json 2> $R/gpurun_out/${TAG}_prof.err
cp /tmp/ks_$TAG/p_kernel_stats.csv $R/gpurun_out/${TAG}_kernel_stats.csv
grep "rpa::" $R/gpurun_out/${TAG}_kernel_stats.csv | cut -c1-60,200-
