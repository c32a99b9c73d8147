#!/bin/bash
# tools/dna_lengths.sh -- the placement path on uniform nucleotide batches of other segment lengths than BASELINE's 200 nt and 10 kbp
# (config 2's shape: 20 hits per record set, the number of sets scaled to ~4e12 cell-equivalents per step): steps in flight,
# the synchronous step, alignment ms, cell-equivalents/s in-kernel.  On the GPU box, from the repo root.
for L in ${LENS:-100 150 200 256 300 384 450 512 700 1000 1500 2000 3000}; do
  S=$(python3 -c "print(max(2000, min(1000000, int(1e6 * 200 * 200 / ($L * $L)))))")
  python bench.py --config c2 --seg-len $L --sets $S --steps 4 --warmup 1 --no-cpu-baseline --no-secondary 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('%5d nt x %7d sets: %7.2f ms/step in flight, %7.2f sync, align %7.2f, %6.1f M demanded realignments/s, %6.1f T cell-equivalents/s in-kernel, computed/demanded %.3f, rounds %.1f' % ($L, $S, d['ms_per_step'], d['sync_unprofiled_ms_per_step'], d['align_ms_per_step'], d['value']/1e6, r['cell_updates_per_s_in_kernel']/1e12, d['computed_over_demanded'], d['rounds_per_step']))"
done
