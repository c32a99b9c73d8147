#!/bin/bash
# tools/spec_sweep7.sh -- round 3 re-sweep of config 2's speculation chunks (pass-1 chunk, pass-2 chunk) on the label-path engine
for i in 1 2; do
for SP in "0 10" "0 7" "0 14" "0 20" "12 10" "0 0"; do
set -- $SP
RPA_SPEC_ADAPTIVE=1 RPA_SPEC1=$1 RPA_SPEC2=$2 python bench.py --config c2 --steps 8 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('c2 spec1/spec2=$SP', 'ms/step', round(d['ms_per_step'],2), 'sync', round(d['sync_ms_per_step'],2), 'align', round(d['align_ms_per_step'],2), 'comp/dem', round(d['computed_over_demanded'],3), 'rounds', d['rounds_per_step'], 'value %.4g'%d['value'])"
done; done
