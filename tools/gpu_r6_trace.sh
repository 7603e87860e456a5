#!/bin/bash
# kernel-trace summary of the training step at B = 1024 (train mode) under an environment: tools/gpu_r6_trace.sh <tag> "ENV=.. ENV=.."
TAG=${1:-r6tr}; ENVS=${2:-}; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp; R=$PWD
( cd /tmp && export $ENVS MDT_TRAIN_BENCH_MODES=train && timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/train -o train -- python $R/tools/train_bench.py 1024 > $OUT/train_run.txt 2> $OUT/train.err )
DB=$(find $OUT/train -name "*.db" | head -1); python tools/prof_summary.py $DB > $OUT/train_kernel_stats.txt; tail -1 $OUT/train_run.txt; head -60 $OUT/train_kernel_stats.txt | cut -c1-175
find $OUT -type f -size +20M -delete
