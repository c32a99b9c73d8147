#!/bin/bash
# tools/ab_env2.sh CONFIG "ENV_A" "ENV_B" ... -- like ab_env.sh with the synchronous leg and the speculation ratio in the line:
# bench.py (no cpu baseline, no secondary) three times each, interleaved, 10 steps
C=$1; shift
for i in 1 2 3; do
for E in "$@"; do
env $E python bench.py --config $C --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$C [$E]', 'ms/step', round(d['ms_per_step'],3), 'sync', round(d['sync_unprofiled_ms_per_step'],3), 'align', round(d['align_ms_per_step'],3), 'rounds', d['rounds_per_step'], 'computed/demanded', round(d['computed_over_demanded'],4), 'launches/step', d['roofline']['launches']/10.0, 'G/s', round(d['value']/1e9,4))"
done; done
