run() { local cfg=$1 label=$2; shift 2
  env "$@" python bench.py --config $cfg --steps 5 --warmup 2 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$cfg %-30s ms/step %6.2f  align %5.2f  computed/demanded %.3f  rounds %.1f  %.3f G/s' % ('$label', d['ms_per_step'], d['align_ms_per_step'], d['computed_over_demanded'], d['rounds_per_step'], d['value']/1e9))"
}
for rep in 1 2 3; do
run c2 "filter 1" RPA_PASS1_FILTER=1
run c2 "filter 2 (one-probe early out)" RPA_PASS1_FILTER=2
run c4 "default (filter 2, whole loop)"
done
