#!/bin/bash
# round 5, first GPU session: the touched tests, PMC passes of the training step / masked-image head, ordered traces of one call
TAG=${1:-r5a}; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp; R=$PWD
echo "== tests"; timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
echo "== train / mae PMC"; bash tools/gpu_train_pmc.sh $TAG/pmc
for B in 256 1; do
  ( cd /tmp && timeout 200 rocprofv3 --kernel-trace -d $OUT/lat$B -o lat -- python $R/tools/latency.py $B > $OUT/lat${B}_run.txt 2> $OUT/lat$B.err )
  DB=$(find $OUT/lat$B -name "*.db" | head -1); python tools/prof_call.py $DB 260 > $OUT/call_B$B.txt; tail -2 $OUT/lat${B}_run.txt
done
find $OUT -name "*.db" -size +30M -delete
