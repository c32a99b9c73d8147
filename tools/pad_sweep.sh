#!/bin/bash
# tools/pad_sweep.sh -- extra dynamic LDS of the one-strip band kernel when several batches are in flight (RPA_BAND4_LDS_PAD:
# 0 = six waves per SIMD, 20000 = four and room for the other jobs' placement kernels) x jobs in flight, config 2
for rep in 1 2; do
for N in 3 4; do
for PAD in 0 12000 20000 32000; do
RPA_JOB_SLOTS=$N RPA_BAND4_LDS_PAD=$PAD python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-secondary --in-flight $N 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('in flight $N pad $PAD', 'ms/step', round(d['ms_per_step'],3), 'G/s', round(d['value']/1e9,4))"
done; done; done
