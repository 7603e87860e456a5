#!/bin/bash
# the masked-image head: its tests, its step time twice, its kernel stats
TAG=${1:-maechk}; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp; R=$PWD
timeout 900 python -m pytest tests/test_mae.py tests/test_c3_step.py -m gpu -x -q 2>&1 | tail -2
for rep in 1 2; do timeout 200 python tools/mae_bench.py 128 1024 2>&1 | grep "B=" | tr '\n' ' '; echo; done | tee $OUT/mae.txt
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/mae -o mae -- python $R/tools/mae_bench.py 1024 > $OUT/mae_run.txt 2> $OUT/mae.err )
DB=$(find $OUT/mae -name "*.db" | head -1); python tools/prof_summary.py $DB > $OUT/mae_kernel_stats.txt; head -14 $OUT/mae_kernel_stats.txt | cut -c1-150
find $OUT -name "*.db" -delete
