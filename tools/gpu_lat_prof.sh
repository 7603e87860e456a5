#!/bin/bash
# B=1 latency: wall time vs sum of kernel durations (is the chain launch- or GPU-bound?)
OUT=$PWD/gpurun_out/${1:-latprof}; mkdir -p $OUT; export TMPDIR=/tmp
python tools/latency.py 1 2>&1 | tail -1 | tee $OUT/latency.txt
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof -o lat -- python $OLDPWD/tools/latency.py 1 > $OUT/prof_run.txt 2> $OUT/prof.err ); echo "rocprof exit $?"
DB=$(find $OUT/prof -name "*.db" | head -1)
[ -n "$DB" ] && python tools/prof_summary.py $DB > $OUT/lat_kernel_stats.txt && head -24 $OUT/lat_kernel_stats.txt
find $OUT/prof -type f -size +20M -delete
