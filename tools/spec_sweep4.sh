#!/bin/bash
# tools/spec_sweep4.sh -- default speculation chunk sizes for the protein path (config 4), per-round adaptation left on
for GS in "0 2 10" "0 3 10" "0 4 10" "0 5 10" "0 6 10" "0 3 6" "0 4 6" "1 2 10" "1 3 10"; do
set -- $GS
RPA_SPEC_GROW=$1 RPA_SPEC_ADAPTIVE=1 RPA_SPEC1=$2 RPA_SPEC2=$3 python bench.py --config c4 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('c4 grow/spec=$GS', 'ms/step', round(d['ms_per_step'],2), 'align', round(d['align_ms_per_step'],2), 'comp/dem', round(d['computed_over_demanded'],3), 'rounds', d['rounds_per_step'], 'value %.3g'%d['value'])"
done
