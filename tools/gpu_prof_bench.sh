#!/bin/bash
# kernel-trace summary of the bench command (one gpurun call).  usage: tools/gpu_prof_bench.sh <tag>
TAG=${1:-prof}; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp; R=$PWD
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof -o bench -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/prof_bench.json 2> $OUT/prof.err )
DB=$(find $OUT/prof -name "*.db" | head -1); python tools/prof_summary.py $DB > $OUT/bench_kernel_stats.txt; head -${2:-24} $OUT/bench_kernel_stats.txt | cut -c1-170
find $OUT -type f -size +5M -delete
