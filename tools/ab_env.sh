#!/bin/bash
# tools/ab_env.sh CONFIG "ENV_A" "ENV_B" ... -- same-device A/B of engine settings: bench.py (no cpu baseline) three times each, interleaved
C=$1; shift
for i in 1 2 3; do
for E in "$@"; do
env $E python bench.py --config $C --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$C [$E]', 'ms/step', round(d['ms_per_step'],3), 'align', round(d['align_ms_per_step'],2), 'rounds', d['rounds_per_step'], 'G/s', round(d['value']/1e9,4))"
done; done
