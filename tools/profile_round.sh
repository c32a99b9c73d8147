#!/bin/bash
# tools/profile_round.sh TAG [CONFIG] -- everything profiles/ needs for one round, on the GPU box (run from the repo root):
#   gpurun_out/TAG_bench.json            python bench.py (default = config 2, with cpu_baseline)
#   gpurun_out/TAG_kernel_stats.csv      rocprofv3 --kernel-trace --stats of the same command with --sync (no cpu baseline): the synchronous
#                                        calls behind the bench line's `roofline` (jobs in flight inflate each other's launch durations)
#   gpurun_out/TAG_pmc_{sq,fetch,write}.txt   three separate --pmc passes, our kernels only, summarised per kernel
TAG=${1:-rXX}
CFG=${2:-c2}
ROOT=$PWD
mkdir -p gpurun_out
python bench.py --config $CFG --steps 10 --warmup 2 > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -o p -- python3 $ROOT/bench.py --config $CFG --steps 5 --warmup 1 --no-cpu-baseline --sync > $ROOT/gpurun_out/${TAG}_bench_profiled.json 2>/dev/null
cp /tmp/prof_$TAG/p_kernel_stats.csv $ROOT/gpurun_out/${TAG}_kernel_stats.csv
i=0
for P in "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "FETCH_SIZE" "WRITE_SIZE"; do
  N=$(echo sq fetch write | cut -d" " -f$((i+1))); i=$((i+1))
  rocprofv3 --kernel-trace --pmc $P --kernel-include-regex "rpa::" --output-format csv -d /tmp/pmc_${TAG}_$N -o pmc -- python3 $ROOT/bench.py --config $CFG --steps 2 --warmup 1 --no-cpu-baseline --sync > /dev/null 2>&1
  python3 $ROOT/tools/pmc_summary.py /tmp/pmc_${TAG}_$N/pmc_counter_collection.csv > $ROOT/gpurun_out/${TAG}_pmc_$N.txt
done
cd $ROOT
cat gpurun_out/${TAG}_bench.json
