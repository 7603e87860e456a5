#!/bin/bash
# kernel-trace timeline of the last sampler call at a rollout-sized batch (which kernels a latency-bound chain is made of)
# usage: tools/gpu_lowbatch_prof.sh <tag> <B> [<B> ...]
TAG=$1; shift; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp; R=$PWD
for B in "$@"; do
  ( cd /tmp && MDT_HIP_GRAPH=0 timeout 200 rocprofv3 --kernel-trace --stats -d $OUT/lat$B -o lat -- python $R/tools/latency.py $B > $OUT/lat${B}_run.txt 2> $OUT/lat$B.err )
  DB=$(find $OUT/lat$B -name "*.db" | head -1); python tools/prof_gaps.py $DB 400 > $OUT/lat${B}_gaps.txt; echo "== B=$B"; grep B= $OUT/lat${B}_run.txt; head -24 $OUT/lat${B}_gaps.txt | cut -c1-140
  find $OUT/lat$B -type f -size +5M -delete
done
