#!/bin/bash
# tools/tl.sh TAG [LIB] -- timeline of a synchronous config-2 step only (LIB: an A/B build, RPA_LIB_PATH) -> gpurun_out/TAG_timeline.txt
TAG=$1
[ -n "$2" ] && export RPA_LIB_PATH=$PWD/$2
bash tools/kt_run.sh > /dev/null 2>&1 && python tools/kt_timeline.py gpurun_out/kt_rpa.csv > gpurun_out/${TAG}_timeline.txt
tail -11 gpurun_out/${TAG}_timeline.txt | head -5
