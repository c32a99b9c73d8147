#!/bin/bash
# tools/pmc_traffic.sh TAG CONFIG -- the two HBM-traffic --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs, our kernels only)
# of bench.py on CONFIG -> gpurun_out/TAG_pmc_{fetch,write}.txt   (then: python3 tools/make_traffic_json.py TAG 3)
TAG=$1; CFG=$2
R=$PWD; cd /tmp && export TMPDIR=/tmp
for P in FETCH_SIZE WRITE_SIZE; do
  N=$(echo $P | tr A-Z a-z | cut -d_ -f1)
  rm -rf /tmp/pmc_${TAG}_$N
  rocprofv3 --kernel-trace --pmc $P --kernel-include-regex "rpa::" --output-format csv -d /tmp/pmc_${TAG}_$N -o pmc -- python3 $R/bench.py --config $CFG --steps 2 --warmup 1 --no-cpu-baseline --sync > /dev/null 2> $R/gpurun_out/${TAG}_pmc_$N.err
  python3 $R/tools/pmc_summary.py /tmp/pmc_${TAG}_$N/pmc_counter_collection.csv > $R/gpurun_out/${TAG}_pmc_$N.txt
  echo "$P done" >> $R/gpurun_out/${TAG}_pmc_progress.txt
done
