#!/bin/bash
# tools/spec_sweep8.sh -- pass-1 chunks with the thresholds of the paused loop applied when the chunk is made (mode 5,
# RPA_PASS1_FILTER): chunk sizes for config 2 and config 4, three steps in flight, against the unfiltered requests.
run() { # cfg, label, env...
  local cfg=$1 label=$2; shift 2
  env "$@" python bench.py --config $cfg --steps 5 --warmup 2 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$cfg %-28s ms/step %6.2f  align %5.2f  computed/demanded %.3f  rounds %.1f  %.3f G/s' % ('$label', d['ms_per_step'], d['align_ms_per_step'], d['computed_over_demanded'], d['rounds_per_step'], d['value']/1e9))"
}
for rep in 1 2; do
run c4 "S1=4 unfiltered" RPA_PASS1_FILTER=0
run c4 "S1=4 filtered" RPA_PASS1_FILTER=1
run c4 "S1=3 filtered" RPA_PASS1_FILTER=1 RPA_SPEC1=3 RPA_SPEC2=10
run c4 "S1=5 filtered" RPA_PASS1_FILTER=1 RPA_SPEC1=5 RPA_SPEC2=10
run c4 "S1=6 filtered" RPA_PASS1_FILTER=1 RPA_SPEC1=6 RPA_SPEC2=10
run c2 "S1=0 (whole loop)" RPA_PASS1_FILTER=1
run c2 "S1=6 filtered" RPA_PASS1_FILTER=1 RPA_SPEC1=6 RPA_SPEC2=10
run c2 "S1=8 filtered" RPA_PASS1_FILTER=1 RPA_SPEC1=8 RPA_SPEC2=10
run c2 "S1=10 filtered" RPA_PASS1_FILTER=1 RPA_SPEC1=10 RPA_SPEC2=10
run c2 "S1=12 filtered" RPA_PASS1_FILTER=1 RPA_SPEC1=12 RPA_SPEC2=10
done
