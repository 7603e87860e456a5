#!/bin/bash
# kernel stats of the masked-image head with the SwishGLU products on the tall body / on the 32-row tiles
TAG=${1:-r5c}; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp; R=$PWD
for V in 1 0; do
  ( cd /tmp && MDT_HIP_GLU_TALL=$V timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/mae$V -o mae -- python $R/tools/mae_bench.py 1024 > $OUT/mae${V}_run.txt 2> $OUT/mae$V.err )
  DB=$(find $OUT/mae$V -name "*.db" | head -1); python tools/prof_summary.py $DB > $OUT/mae${V}_kernel_stats.txt; head -16 $OUT/mae${V}_kernel_stats.txt | cut -c1-140
done
find $OUT -name "*.db" -delete
