#!/bin/bash
# tools/spec_sweep6.sh -- round 3 re-sweep of the protein path's speculation chunks (config 4) on the label-path engine:
# pass-1 chunk, pass-2 chunk, growing chunks; per-round adaptation left on.  Three steps in flight (the headline mode).
for GS in "0 4 10" "0 2 10" "0 3 10" "0 6 10" "0 8 10" "1 2 10" "1 3 10" "1 4 10" "0 3 6" "0 4 20" "0 0 10" "1 1 10"; do
set -- $GS
RPA_SPEC_GROW=$1 RPA_SPEC_ADAPTIVE=1 RPA_SPEC1=$2 RPA_SPEC2=$3 python bench.py --config c4 --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('c4 grow/spec1/spec2=$GS', 'ms/step', round(d['ms_per_step'],2), 'sync', round(d['sync_ms_per_step'],2), 'align', round(d['align_ms_per_step'],2), 'comp/dem', round(d['computed_over_demanded'],3), 'rounds', d['rounds_per_step'], 'value %.3g'%d['value'])"
done
