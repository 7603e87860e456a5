#!/bin/bash
# rocprofv3 kernel stats of the training step benchmark (eval-mode arithmetic), summary -> gpurun_out/<tag>/
TAG=${1:-trainprof}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 300 python tools/train_bench.py 128 > $OUT/train_bench.txt 2>&1; cat $OUT/train_bench.txt | tail -2
( cd /tmp && MDT_TRAIN_BENCH_MODES=eval timeout 420 rocprofv3 --kernel-trace --stats -d $OUT/prof -o train -- python $OLDPWD/tools/train_bench.py 128 > $OUT/prof_run.txt 2> $OUT/prof.err ); echo "rocprof exit $?"
DB=$(find $OUT/prof -name "*.db" | head -1)
[ -n "$DB" ] && python tools/prof_summary.py $DB > $OUT/train_kernel_stats.txt && head -40 $OUT/train_kernel_stats.txt
find $OUT/prof -type f -size +20M -delete
