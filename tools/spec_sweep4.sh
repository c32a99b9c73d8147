#!/bin/bash
# tools/spec_sweep4.sh -- default speculation chunk sizes for the protein path (config 4), per-round adaptation left on
for G in 0 1; do
for SP in "2 6" "2 10" "1 6" "3 6" "2 3"; do
set -- $SP
RPA_SPEC_GROW=$G RPA_SPEC_ADAPTIVE=1 RPA_SPEC1=$1 RPA_SPEC2=$2 python bench.py --config c4 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('c4 grow=$G spec=$SP', 'ms/step', round(d['ms_per_step'],2), 'align', round(d['align_ms_per_step'],2), 'comp/dem', round(d['computed_over_demanded'],3), 'rounds', d['rounds_per_step'], 'value %.3g'%d['value'])"
done; done
