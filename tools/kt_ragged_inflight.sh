#!/bin/bash
# tools/kt_ragged_inflight.sh [VARIANT] [IN_FLIGHT] -- on the GPU box: kernel trace of tools/ragged_bench.py with 16 steps in flight (and one
# synchronous step each for its other two legs), tools/kt_inflight.py over the middle of it -> gpurun_out/kt_ragged_inflight.txt
V=${1:-pruned}; J=${2:-4}
R=$PWD; cd /tmp && export TMPDIR=/tmp RAGGED_SYNC_STEPS=1
rm -rf /tmp/ktri
rocprofv3 --kernel-trace --output-format csv -d /tmp/ktri -o kt -- python3 $R/tools/ragged_bench.py --variants $V --steps 16 --in-flight $J --oracle-cells 0 > $R/gpurun_out/kt_ragged_inflight.json 2>/dev/null
head -1 /tmp/ktri/kt_kernel_trace.csv > /tmp/ktri/rpa.csv
grep "rpa::" /tmp/ktri/kt_kernel_trace.csv >> /tmp/ktri/rpa.csv
cd $R && python3 tools/kt_inflight.py /tmp/ktri/rpa.csv > gpurun_out/kt_ragged_inflight.txt; cat gpurun_out/kt_ragged_inflight.txt
