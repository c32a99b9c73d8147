#!/bin/bash
# tools/pmc_pairs.sh -- PMC counters of the pair-level kernels over tools/occupancy_probe.py (GPU box, from the repo root)
R=$PWD; cd /tmp && export TMPDIR=/tmp
i=0
for P in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_IFETCH SQ_INST_CYCLES_VMEM_RD SQ_INSTS_SALU"; do
  i=$((i+1)); rm -rf /tmp/pmcp_$i
  rocprofv3 --kernel-trace --pmc $P --kernel-include-regex "k_myers_short" --output-format csv -d /tmp/pmcp_$i -o pmc -- python3 $R/tools/occupancy_probe.py 500000 > /dev/null 2>&1
  python3 $R/tools/pmc_summary.py /tmp/pmcp_$i/pmc_counter_collection.csv > $R/gpurun_out/pmc_pairs_$i.txt
done
cat $R/gpurun_out/pmc_pairs_1.txt $R/gpurun_out/pmc_pairs_2.txt
