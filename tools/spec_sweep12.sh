#!/bin/bash
# tools/spec_sweep12.sh -- the pass-2 chunk again on the final engine (alignments are 14 % cheaper than when 10 was chosen)
run() { local cfg=$1 label=$2; shift 2
  env "$@" python bench.py --config $cfg --steps 8 --warmup 2 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$cfg %-14s ms/step %6.2f  align %5.2f  computed/demanded %.3f  rounds %.1f  %.3f G/s' % ('$label', d['ms_per_step'], d['align_ms_per_step'], d['computed_over_demanded'], d['rounds_per_step'], d['value']/1e9))"
}
for rep in 1 2; do
for S2 in 6 10 14 20 0; do
run c2 "S2=$S2" RPA_SPEC1=0 RPA_SPEC2=$S2 RPA_SPEC_ADAPTIVE=1
done
for S2 in 6 10 16; do
run c4 "S2=$S2" RPA_SPEC1=0 RPA_SPEC2=$S2 RPA_SPEC_ADAPTIVE=1
done
done
