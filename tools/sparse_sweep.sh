#!/bin/bash
# tools/sparse_sweep.sh CONFIG -- from which live fraction on a round goes through the live list (emit / decide)
C=${1:-c4}
for ED in "24 24" "12 24" "6 24" "3 24" "24 12" "24 6" "24 3" "6 6" "3 3" "2 2"; do
set -- $ED
RPA_SPARSE_EMIT=$1 RPA_SPARSE_DECIDE=$2 python bench.py --config $C --steps 4 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$C emit/decide 1/$1 1/$2', 'ms/step', round(d['ms_per_step'],3), 'align', round(d['align_ms_per_step'],2), 'rounds', d['rounds_per_step'])"
done
