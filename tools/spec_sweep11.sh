run() { local cfg=$1 label=$2; shift 2
  env "$@" python bench.py --config $cfg --steps 3 --warmup 1 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$cfg %-22s ms/step %7.2f  align %7.2f  computed/demanded %.3f  rounds %.1f  %.4f G/s' % ('$label', d['ms_per_step'], d['align_ms_per_step'], d['computed_over_demanded'], d['rounds_per_step'], d['value']/1e9))"
}
for rep in 1 2; do
run c5 "filter 1" RPA_PASS1_FILTER=1
run c5 "filter 2 (default)"
run c2 "filter 2 (default)"
run c2 "filter 3" RPA_PASS1_FILTER=3
done
