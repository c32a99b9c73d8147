#!/bin/bash
# tools/ab.sh [config] LIB... -- same-device A/B of engine builds: bench.py (no cpu baseline) three times per build, interleaved.
# A LIB is a suffix of taxator-tk_amd/librpa_hip<suffix>.so ("" = the current build), e.g.  tools/ab.sh c2 "" _base
CFG=${1:-c2}; shift
LIBS=("$@"); [ ${#LIBS[@]} -eq 0 ] && LIBS=("" "_base")
for i in 1 2 3; do
for V in "${LIBS[@]}"; do
L=$PWD/taxator-tk_amd/librpa_hip${V}.so
RPA_LIB_PATH=$L python bench.py --config $CFG --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('${V:-current}', 'ms/step', round(d['ms_per_step'],2), 'align', round(d['align_ms_per_step'],2), 'G/s', round(d['value']/1e9,3))"
done; done
