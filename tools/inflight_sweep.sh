#!/bin/bash
# tools/inflight_sweep.sh [config] -- steps in flight on the engine's job slots (RPA_JOB_SLOTS = --in-flight = 1..6), device-resident batches
C=${1:-c2}
for i in 1 2; do
for N in 1 2 3 4 5 6; do
RPA_JOB_SLOTS=$N python bench.py --config $C --steps 12 --warmup 3 --no-cpu-baseline --in-flight $N 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$C in flight $N', 'ms/step', round(d['ms_per_step'],3), 'sync', round(d['sync_ms_per_step'],2), 'G/s', round(d['value']/1e9,4), 'hbm GB', d['hbm_in_use_gb'])"
done; done
