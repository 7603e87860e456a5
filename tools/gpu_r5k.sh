#!/bin/bash
TAG=${1:-r5k}; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp; R=$PWD
for B in 256 1; do
  ( cd /tmp && timeout 200 rocprofv3 --kernel-trace -d $OUT/lat$B -o lat -- python $R/tools/latency.py $B > $OUT/lat${B}_run.txt 2> $OUT/lat$B.err )
  DB=$(find $OUT/lat$B -name "*.db" | head -1); python tools/prof_call.py $DB 260 > $OUT/call_B$B.txt; python tools/prof_summary.py $DB > $OUT/stats_B$B.txt; tail -1 $OUT/lat${B}_run.txt
done
find $OUT -name "*.db" -delete
