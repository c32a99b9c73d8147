#!/bin/bash
# tools/dna_hits.sh -- the placement path on config 2's segments (200 nt) with other numbers of hits per record set (20 M alignment
# records per step whatever the set size): steps in flight, the synchronous step, alignment ms.  On the GPU box, from the repo root.
for H in ${HITS:-2 5 10 20 40 64 65 100 200 400}; do
  S=$(python3 -c "print(max(2000, int(2e7 / $H)))")
  python bench.py --config c2 --hits $H --sets $S --steps 4 --warmup 1 --no-cpu-baseline --no-secondary 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('%4d hits x %8d sets: %7.2f ms/step in flight, %7.2f sync, align %6.2f, %7.1f M demanded realignments/s, %6.1f M record sets/s, computed/demanded %.3f, rounds %.1f' % ($H, $S, d['ms_per_step'], d['sync_unprofiled_ms_per_step'], d['align_ms_per_step'], d['value']/1e6, d['record_sets_per_s']/1e6, d['computed_over_demanded'], d['rounds_per_step']))"
done
