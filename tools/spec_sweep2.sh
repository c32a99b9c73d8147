#!/bin/bash
# tools/spec_sweep2.sh -- pass-1 / pass-2 chunk sizes on config 2 (per-round adaptation left on)
for GS in "0 0 10" "0 16 10" "0 14 10" "0 12 10" "0 10 10" "0 0 8" "0 0 14" "1 8 10"; do
set -- $GS
RPA_SPEC_GROW=$1 RPA_SPEC_ADAPTIVE=1 RPA_SPEC1=$2 RPA_SPEC2=$3 python bench.py --config c2 --steps 4 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('c2 grow/spec=$GS', 'ms/step', round(d['ms_per_step'],2), 'align', round(d['align_ms_per_step'],2), 'comp/dem', round(d['computed_over_demanded'],3), 'rounds', d['rounds_per_step'], 'value %.4g'%d['value'])"
done
