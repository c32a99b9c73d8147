#!/bin/bash
# tools/bigstore.sh TAG [REPLICAS] -- config 2 against a reference store that is really in HBM:
#   gpurun_out/TAG_bigstore_bench_r1.json / _rN.json   bench lines with 1 and N replicas of the collection (N x 8 MB packed)
#   gpurun_out/TAG_bigstore_pmc_{fetch,tcc}_rN.txt     FETCH_SIZE and L2 hit / miss counters per kernel, both store sizes
TAG=$1; N=${2:-4000}
R=$PWD; cd /tmp && export TMPDIR=/tmp
for REP in 1 $N; do
  python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --store-replicas $REP > $R/gpurun_out/${TAG}_bigstore_bench_r$REP.json 2> $R/gpurun_out/${TAG}_bigstore_bench_r$REP.err
  for P in "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
    NAME=$(echo $P | cut -d' ' -f1 | tr A-Z a-z | cut -d_ -f1)
    rm -rf /tmp/pmc_${TAG}_${NAME}_$REP
    rocprofv3 --kernel-trace --pmc $P --kernel-include-regex "rpa::" --output-format csv -d /tmp/pmc_${TAG}_${NAME}_$REP -o pmc -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --sync --store-replicas $REP > /dev/null 2> $R/gpurun_out/${TAG}_bigstore_pmc_${NAME}_r$REP.err
    python3 $R/tools/pmc_summary.py /tmp/pmc_${TAG}_${NAME}_$REP/pmc_counter_collection.csv > $R/gpurun_out/${TAG}_bigstore_pmc_${NAME}_r$REP.txt
  done
  echo "replicas $REP done" >> $R/gpurun_out/${TAG}_bigstore_progress.txt
done
