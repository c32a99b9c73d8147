#!/bin/bash
# tools/spec_sweep9.sh -- pass-1 chunks that end where the scores fall below the cut-off their first hit would set
# (RPA_PASS1_FILTER=2) against the thresholds-at-the-pause requests (1) and the plain ones (0), with the pass-1 chunk sizes
# that go with them; three steps in flight.
run() { # cfg, label, env...
  local cfg=$1 label=$2; shift 2
  env "$@" python bench.py --config $cfg --steps 5 --warmup 2 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$cfg %-30s ms/step %6.2f  align %5.2f  computed/demanded %.3f  rounds %.1f  %.3f G/s' % ('$label', d['ms_per_step'], d['align_ms_per_step'], d['computed_over_demanded'], d['rounds_per_step'], d['value']/1e9))"
}
for rep in 1 2; do
run c4 "filter 1, S1=4 (before)" RPA_PASS1_FILTER=1
run c4 "filter 2, S1=4" RPA_PASS1_FILTER=2
run c4 "filter 2, whole loop" RPA_PASS1_FILTER=2 RPA_SPEC1=0 RPA_SPEC2=10 RPA_SPEC_ADAPTIVE=1
run c4 "filter 2, S1=8" RPA_PASS1_FILTER=2 RPA_SPEC1=8 RPA_SPEC2=10 RPA_SPEC_ADAPTIVE=1
run c4 "filter 2, whole loop, S2=6" RPA_PASS1_FILTER=2 RPA_SPEC1=0 RPA_SPEC2=6 RPA_SPEC_ADAPTIVE=1
run c2 "filter 1 (before)" RPA_PASS1_FILTER=1
run c2 "filter 2" RPA_PASS1_FILTER=2
done
