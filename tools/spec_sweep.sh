#!/bin/bash
# tools/spec_sweep.sh SWEEP [REPS] -- the request-policy sweeps of rounds 1-5 as ONE script (they were thirteen near-identical files:
# spec_sweep.sh, spec_sweep2.sh, 4 ... 12, 9b): every line of a sweep is "CONFIG ENV..." and runs `python bench.py --config CONFIG`
# (10 steps, no CPU legs, no secondary) under that environment; the sweep is repeated REPS times (default 2), its lines interleaved,
# on the same device.  Prints ms per step in flight and synchronous, alignment ms, rounds, alignments computed per demanded one.
#   chunks   pass-1 / pass-2 chunk sizes (RPA_SPEC1 / RPA_SPEC2; 0 = the whole loop), configs 2 and 4          (round 1-3: sweeps 1, 2, 6, 7, 12)
#   grow     chunks that double when a loop pauses (RPA_SPEC_GROW), the per-round adaptation left on             (round 2: sweep 2)
#   filter   pass-1 requests without / with the paused loop's thresholds / ending at the first hit's score cut-off (round 3: sweeps 8, 9, 9b, 11)
#   first    an anchor's first pass-2 chunk shorter than the following ones (RPA_SPEC2_FIRST)                    (round 3: sweep 10)
#   defer    two launches per pass-2 round (RPA_DEFER2), alone and with whole pass-2 loops                       (round 5)
#   custom   lines from stdin
S=${1:-defer}; REPS=${2:-2}
case $S in
chunks) LINES=$'c2 RPA_SPEC1=0 RPA_SPEC2=6 RPA_SPEC_ADAPTIVE=1\nc2 RPA_SPEC1=0 RPA_SPEC2=10 RPA_SPEC_ADAPTIVE=1\nc2 RPA_SPEC1=0 RPA_SPEC2=14 RPA_SPEC_ADAPTIVE=1\nc2 RPA_SPEC1=0 RPA_SPEC2=20 RPA_SPEC_ADAPTIVE=1\nc2 RPA_SPEC1=0 RPA_SPEC2=0 RPA_SPEC_ADAPTIVE=1\nc2 RPA_SPEC1=8 RPA_SPEC2=10 RPA_SPEC_ADAPTIVE=1\nc4 RPA_SPEC1=0 RPA_SPEC2=6 RPA_SPEC_ADAPTIVE=1\nc4 RPA_SPEC1=0 RPA_SPEC2=10 RPA_SPEC_ADAPTIVE=1\nc4 RPA_SPEC1=0 RPA_SPEC2=16 RPA_SPEC_ADAPTIVE=1\nc4 RPA_SPEC1=4 RPA_SPEC2=10 RPA_SPEC_ADAPTIVE=1';;
grow) LINES=$'c2 RPA_SPEC_GROW=0 RPA_SPEC1=0 RPA_SPEC2=10 RPA_SPEC_ADAPTIVE=1\nc2 RPA_SPEC_GROW=1 RPA_SPEC1=8 RPA_SPEC2=10 RPA_SPEC_ADAPTIVE=1\nc2 RPA_SPEC_GROW=0 RPA_SPEC1=12 RPA_SPEC2=10 RPA_SPEC_ADAPTIVE=1\nc2 RPA_SPEC_GROW=0 RPA_SPEC1=16 RPA_SPEC2=10 RPA_SPEC_ADAPTIVE=1';;
filter) LINES=$'c4 RPA_PASS1_FILTER=0\nc4 RPA_PASS1_FILTER=1\nc4 RPA_PASS1_FILTER=2\nc4 RPA_PASS1_FILTER=2 RPA_SPEC1=8 RPA_SPEC2=10 RPA_SPEC_ADAPTIVE=1\nc2 RPA_PASS1_FILTER=1\nc2 RPA_PASS1_FILTER=2\nc2 RPA_PASS1_FILTER=3\nc5 RPA_PASS1_FILTER=1\nc5 RPA_PASS1_FILTER=2';;
first) LINES=$'c2 RPA_SPEC2_FIRST=0\nc2 RPA_SPEC2_FIRST=3\nc2 RPA_SPEC2_FIRST=6\nc4 RPA_SPEC2_FIRST=0\nc4 RPA_SPEC2_FIRST=4';;
defer) LINES=$'c2 RPA_DEFER2=0\nc2 RPA_DEFER2=1\nc2 RPA_DEFER2=1 RPA_SPEC2=0\nc2 RPA_DEFER2=0 RPA_SPEC2=0\nc4 RPA_DEFER2=0\nc4 RPA_DEFER2=1\nc5 RPA_DEFER2=0\nc5 RPA_DEFER2=1';;
custom) LINES=$(cat);;
*) echo "unknown sweep $S (chunks | grow | filter | first | defer | custom)"; exit 2;;
esac
for rep in $(seq $REPS); do
  while read -r C E; do
    [ -z "$C" ] && continue
    env $E python bench.py --config $C --steps 10 --warmup 2 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$C %-52s ms/step %7.3f  sync %7.3f  align %7.3f  rounds %.1f  computed/demanded %.4f  %.4f G/s' % ('$E', d['ms_per_step'], d['sync_unprofiled_ms_per_step'], d['align_ms_per_step'], d['rounds_per_step'], d['computed_over_demanded'], d['value']/1e9))"
  done <<< "$LINES"
done
