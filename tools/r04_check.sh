#!/bin/bash
# tools/r04_check.sh -- round 4: 300 aa protein segments with and without the score-and-length cell, then the default bench line
mkdir -p gpurun_out
for W in 1 0; do
  RPA_PROTEIN_WIDE=$W python bench.py --config c4 --sets 125000 --seg-len 300 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read())
print('300 aa, RPA_PROTEIN_WIDE=$W:', 'ms/step', round(d['ms_per_step'],2), 'M realignments/s', round(d['value']/1e6,1), 'sync ms', round(d['sync_ms_per_step'],2), 'cell-equivalents/s in kernel', '%.3g' % d['roofline']['cell_updates_per_s_in_kernel'], 'computed/demanded', round(d['computed_over_demanded'],3))"
done > gpurun_out/r04_protein_300aa.txt 2>&1
cat gpurun_out/r04_protein_300aa.txt
python bench.py > gpurun_out/r04_bench_a.json 2> gpurun_out/r04_bench_a.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r04_bench_a.json"))
print("headline G/s %.4f ms/step %.3f sync %.3f frac %.4f band/equiv %.3f computed/demanded %.4f rounds %.1f" % (d["value"] / 1e9, d["ms_per_step"], d["sync_ms_per_step"], d["roofline"]["frac"], d["roofline"]["band_cells_over_cell_equivalents"], d["computed_over_demanded"], d["rounds_per_step"]))
for k, v in d["secondary"].items():
    print(k, {a: b for a, b in v.items() if a in ("value", "ms_per_step", "sync_ms_per_step", "error", "wall_s", "record_sets_per_s", "pipeline_record_sets_per_s", "same_output_with_small_blocks", "stages_s", "ok")})
PY
tail -3 gpurun_out/r04_bench_a.err
