#!/bin/bash
# tools/tl.sh TAG [LIB] [bench args] -- timeline of a synchronous step only (LIB: an A/B build, RPA_LIB_PATH; "" = current) -> gpurun_out/TAG_timeline.txt
TAG=$1; LIB=$2; shift; shift
[ -n "$LIB" ] && export RPA_LIB_PATH=$PWD/$LIB
bash tools/kt_run.sh "$@" > /dev/null 2>&1 && python tools/kt_timeline.py gpurun_out/kt_rpa.csv > gpurun_out/${TAG}_timeline.txt
tail -11 gpurun_out/${TAG}_timeline.txt | head -5
