#!/bin/bash
# tools/pmc_c4.sh -- SQ counters of the protein alignment kernel on config 4 (one --pmc pass, our kernels only)
R=$PWD; cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_c4
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT --kernel-include-regex "rpa::k_nw" --output-format csv -d /tmp/pmc_c4 -o pmc -- python3 $R/bench.py --config c4 --steps 1 --warmup 1 --no-cpu-baseline --sync > /dev/null 2>&1
python3 $R/tools/pmc_summary.py /tmp/pmc_c4/pmc_counter_collection.csv > $R/gpurun_out/pmc_c4.txt
cat $R/gpurun_out/pmc_c4.txt
