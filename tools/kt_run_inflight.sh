#!/bin/bash
# tools/kt_run_inflight.sh TAG [bench args] -- kernel trace of 12 pipelined bench steps (jobs in flight) -> gpurun_out/TAG_inflight.txt
TAG=$1; shift
R=$PWD; cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kti
rocprofv3 --kernel-trace --output-format csv -d /tmp/kti -o kt -- python3 $R/bench.py --steps 12 --warmup 2 --no-cpu-baseline --no-secondary "$@" > $R/gpurun_out/${TAG}_inflight_bench.json 2>/dev/null
head -1 /tmp/kti/kt_kernel_trace.csv > /tmp/kti/rpa.csv
grep "rpa::" /tmp/kti/kt_kernel_trace.csv >> /tmp/kti/rpa.csv
cd $R && python tools/kt_inflight.py /tmp/kti/rpa.csv > gpurun_out/${TAG}_inflight.txt; cat gpurun_out/${TAG}_inflight.txt
