#!/bin/bash
# A/B of library builds on the B = 256 sampler call (+ optional kernel trace of the product build)
# usage: tools/gpu_libs_ab.sh <tag> <lib.so> [<lib.so> ...]      (names under mdt_policy_amd/csrc/)
TAG=${1:-libs}; shift; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp; R=$PWD
for rep in 1 2; do for lib in "$@"; do
  echo -n "$lib: "; MDT_HIP_LIB=$R/mdt_policy_amd/csrc/$lib timeout 200 python tools/latency.py 256 2>&1 | grep "B="
done; done | tee $OUT/ab.txt
if [ -n "$TRACE" ]; then
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof -o bench -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/prof_bench.json 2> $OUT/prof.err )
DB=$(find $OUT/prof -name "*.db" | head -1); python tools/prof_summary.py $DB > $OUT/bench_kernel_stats.txt; head -16 $OUT/bench_kernel_stats.txt | cut -c1-150
find $OUT -type f -size +20M -delete
fi
