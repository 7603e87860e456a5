#!/bin/bash
# rocprofv3 kernel stats of the training step benchmark (eval-mode arithmetic), summary -> gpurun_out/<tag>/
# usage: tools/gpu_train_prof.sh <tag> [batch]
TAG=${1:-trainprof}; B=${2:-128}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
( cd /tmp && MDT_TRAIN_BENCH_MODES=eval timeout 420 rocprofv3 --kernel-trace --stats -d $OUT/prof -o train -- python $OLDPWD/tools/train_bench.py $B > $OUT/prof_run.txt 2> $OUT/prof.err ); echo "rocprof exit $?"
DB=$(find $OUT/prof -name "*.db" | head -1)
[ -n "$DB" ] && python tools/prof_summary.py $DB > $OUT/train_kernel_stats.txt && head -34 $OUT/train_kernel_stats.txt
find $OUT/prof -type f -size +20M -delete
