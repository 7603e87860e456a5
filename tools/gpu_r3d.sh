#!/bin/bash
# round 3: MGF head after the SwishGLU fusion: tests, timing, kernel stats
TAG=${1:-r3d}; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp; R=$PWD
timeout 900 python -m pytest tests/test_mae.py tests/test_c3_step.py -m gpu -q --timeout 600 -p no:cacheprovider -x > $OUT/pytest.log 2>&1; echo "tests exit $?" | tee -a $OUT/summary.txt; tail -8 $OUT/pytest.log
timeout 300 python tools/mae_bench.py 128 1024 2>&1 | grep "B=" | tee $OUT/mae_bench.txt
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/mae -o mae -- python $R/tools/mae_bench.py 1024 > $OUT/mae_run.txt 2> $OUT/mae.err )
DB=$(find $OUT/mae -name "*.db" | head -1); python tools/prof_summary.py $DB > $OUT/mae_kernel_stats.txt; head -24 $OUT/mae_kernel_stats.txt | cut -c1-150
find $OUT -type f -size +20M -delete
