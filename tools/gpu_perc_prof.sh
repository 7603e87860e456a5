#!/bin/bash
OUT=$PWD/gpurun_out/${1:-percprof}; mkdir -p $OUT; export TMPDIR=/tmp
python tests/perf_eager_baseline.py 2>&1 | tail -5 | tee $OUT/eager.txt
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof -o perc -- python $OLDPWD/tools/perceiver_train_bench.py 128 > $OUT/prof_run.txt 2> $OUT/prof.err ); echo "rocprof exit $?"
DB=$(find $OUT/prof -name "*.db" | head -1)
[ -n "$DB" ] && python tools/prof_summary.py $DB > $OUT/perc_kernel_stats.txt && head -24 $OUT/perc_kernel_stats.txt
find $OUT/prof -type f -size +20M -delete
