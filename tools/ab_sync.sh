#!/bin/bash
# tools/ab_sync.sh CONFIG LIB... -- like tools/ab.sh with the synchronous leg too (one call per step)
CFG=${1:-c2}; shift
LIBS=("$@")
for i in 1 2 3; do
for V in "${LIBS[@]}"; do
L=$PWD/taxator-tk_amd/librpa_hip${V}.so
A=$(RPA_LIB_PATH=$L python bench.py --config $CFG --steps 5 --warmup 2 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('in flight %.2f ms (align %.2f, computed/demanded %.3f, rounds %.0f)' % (d['ms_per_step'], d['align_ms_per_step'], d['computed_over_demanded'], d['rounds_per_step']))")
B=$(RPA_LIB_PATH=$L python bench.py --config $CFG --steps 5 --warmup 2 --no-cpu-baseline --no-secondary --sync 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('synchronous %.2f ms' % d['ms_per_step'])")
echo "$CFG ${V:-current}: $A; $B"
done; done
