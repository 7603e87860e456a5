#!/bin/bash
TAG=${1:-r3k}; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp; R=$PWD
timeout 2400 python -m pytest tests -m gpu -q -n 3 --timeout 900 -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "gpu tests exit $?" | tee -a $OUT/summary.txt; tail -3 $OUT/pytest_gpu.log
echo eager;  MDT_HIP_GRAPH=0 timeout 300 python tools/latency.py 1 2 4 8 16 2>&1 | grep B=
echo auto;  timeout 300 python tools/latency.py 1 8 2>&1 | grep B=
( cd /tmp && MDT_HIP_GRAPH=0 timeout 200 rocprofv3 --kernel-trace --stats -d $OUT/lat -o lat -- python $R/tools/latency.py 1 > $OUT/lat_run.txt 2> $OUT/lat.err )
DB=$(find $OUT/lat -name "*.db" | head -1); python tools/prof_summary.py $DB > $OUT/lat_kernel_stats.txt; head -12 $OUT/lat_kernel_stats.txt | cut -c1-150
find $OUT -type f -size +20M -delete
