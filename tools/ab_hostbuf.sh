#!/bin/bash
# tools/ab_hostbuf.sh -- the PCIe-inclusive path (bench.py --host-buffers, three jobs in flight) under different stream kinds:
# CU-masked streams (a hardware queue each; the default), the runtime's pooled queues, pooled queues with more hardware queues
for i in 1 2 3; do
for E in "RPA_CU_MASK_QUEUES=1" "GPU_MAX_HW_QUEUES=4" "GPU_MAX_HW_QUEUES=16" "GPU_MAX_HW_QUEUES=8"; do  # (r03_ab_hostbuf.txt was taken with the round-2 default: A=1 = CU-masked, RPA_POOLED_QUEUES=1 = plain streams)
env $E python bench.py --steps 12 --warmup 3 --no-cpu-baseline --host-buffers --in-flight 3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('hostbuf [$E]', 'ms/step', round(d['ms_per_step'],2), 'G/s', round(d['value']/1e9,3))"
done; done
