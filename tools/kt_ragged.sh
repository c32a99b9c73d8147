#!/bin/bash
# tools/kt_ragged.sh [VARIANT] -- on the GPU box: kernel trace of tools/ragged_bench.py (one variant, 2 synchronous steps), the last
# step's per-dispatch timeline -> gpurun_out/kt_ragged_timeline.txt
V=${1:-pruned}
R=$PWD; cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ktr
rocprofv3 --kernel-trace --output-format csv -d /tmp/ktr -o kt -- python3 $R/tools/ragged_bench.py --variants $V --steps 2 --in-flight 1 --oracle-cells 0 > /dev/null 2> $R/gpurun_out/kt_ragged_stderr.txt
head -1 /tmp/ktr/kt_kernel_trace.csv > $R/gpurun_out/kt_ragged.csv
grep "rpa::" /tmp/ktr/kt_kernel_trace.csv >> $R/gpurun_out/kt_ragged.csv
cd $R
python3 tools/kt_timeline.py gpurun_out/kt_ragged.csv > gpurun_out/kt_ragged_timeline.txt
tail -20 gpurun_out/kt_ragged_timeline.txt
