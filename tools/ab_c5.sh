#!/bin/bash
# tools/ab_c5.sh LIB... -- config 5 (long contigs) on engine builds taxator-tk_amd/librpa_hip<suffix>.so, twice each, interleaved
for i in 1 2; do for V in "$@"; do
RPA_LIB_PATH=$PWD/taxator-tk_amd/librpa_hip$V.so python bench.py --config c5 --steps 3 --warmup 1 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('c5 ${V:-current}', 'ms/step', round(d['ms_per_step'],1), 'align', round(d['align_ms_per_step'],1))"
done; done
