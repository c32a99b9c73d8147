#!/bin/bash
# tools/ab_inflight.sh "NAME:LIB[:ENV=VAL,...]" ... -- same-device A/B of the headline with jobs in flight: every variant three
# times, interleaved (LIB: an A/B build under build_ab/ or "" = the current one; ENV: knobs for that variant)
for i in 1 2 3; do
for V in "$@"; do
  NAME=${V%%:*}; REST=${V#*:}; LIB=${REST%%:*}; ENVS=""; [ "$REST" != "$LIB" ] && ENVS=${REST#*:}
  (
    [ -n "$LIB" ] && export RPA_LIB_PATH=$PWD/$LIB
    for kv in ${ENVS//,/ }; do export "$kv"; done
    python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$NAME', 'in flight ms/step', round(d['ms_per_step'],3), 'sync', round(d['sync_ms_per_step'],3))"
  )
done; done
