#!/bin/bash
# tools/spec_sweep10.sh -- pass 2: a short first chunk per anchor (RPA_SPEC2_FIRST), then the usual one; three steps in flight.
run() { local cfg=$1 label=$2; shift 2
  env "$@" python bench.py --config $cfg --steps 5 --warmup 2 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$cfg %-22s ms/step %6.2f  align %5.2f  computed/demanded %.3f  rounds %.1f  %.3f G/s' % ('$label', d['ms_per_step'], d['align_ms_per_step'], d['computed_over_demanded'], d['rounds_per_step'], d['value']/1e9))"
}
for rep in 1 2; do
for cfg in c2 c4; do
run $cfg "first chunk = 10"
run $cfg "first chunk = 2" RPA_SPEC2_FIRST=2
run $cfg "first chunk = 3" RPA_SPEC2_FIRST=3
run $cfg "first chunk = 4" RPA_SPEC2_FIRST=4
run $cfg "first chunk = 6" RPA_SPEC2_FIRST=6
done; done
