#!/bin/bash
# tools/quick.sh TAG [pytest -k expression] -- on the GPU box, after a kernel change: the parity file (rows, STATS, trace against the
# oracle), the per-dispatch timeline of a synchronous config-2 step and the headline line without its secondary legs
#   -> gpurun_out/TAG_{tests.log,timeline.txt,bench.json}
TAG=$1; K=${2:-}
mkdir -p gpurun_out
if [ -n "$K" ]; then python -m pytest tests/test_gpu_parity.py -x -q -k "$K" > gpurun_out/${TAG}_tests.log 2>&1
else python -m pytest tests/test_gpu_parity.py -x -q > gpurun_out/${TAG}_tests.log 2>&1; fi
rc=$?
tail -3 gpurun_out/${TAG}_tests.log
[ $rc -ne 0 ] && exit $rc
bash tools/kt_run.sh > /dev/null 2>&1 && python tools/kt_timeline.py gpurun_out/kt_rpa.csv > gpurun_out/${TAG}_timeline.txt
tail -12 gpurun_out/${TAG}_timeline.txt
python bench.py --no-cpu-baseline --no-secondary > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
python - <<PY
import json
d = json.load(open("gpurun_out/${TAG}_bench.json"))
print("headline G/s %.4f ms/step %.3f sync %.3f computed/demanded %.4f rounds %.1f" % (d["value"] / 1e9, d["ms_per_step"], d["sync_ms_per_step"], d["computed_over_demanded"], d["rounds_per_step"]))
PY
