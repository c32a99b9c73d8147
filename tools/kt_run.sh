#!/bin/bash
# tools/kt_run.sh [bench args] -- on the GPU box: kernel trace of two bench steps, rpa:: dispatches -> gpurun_out/kt_rpa.csv
R=$PWD; cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt
rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -o kt -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --sync "$@" > $R/gpurun_out/kt_bench.json 2>/dev/null
head -1 /tmp/kt/kt_kernel_trace.csv > $R/gpurun_out/kt_rpa.csv
grep "rpa::" /tmp/kt/kt_kernel_trace.csv >> $R/gpurun_out/kt_rpa.csv
wc -l $R/gpurun_out/kt_rpa.csv
