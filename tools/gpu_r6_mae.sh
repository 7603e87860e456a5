#!/bin/bash
# MAE head / C3 A/B: tools/gpu_r6_mae.sh <tag> "ENV=a|ENV=b" [reps]
TAG=${1:-r6mae}; SETS=${2:-"|"}; REPS=${3:-2}; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
IFS='|' read -ra ARR <<< "$SETS"
for r in $(seq 1 $REPS); do
  for S in "${ARR[@]}"; do
    echo "== [$S]" | tee -a $OUT/ab.txt
    ( export $S; timeout 200 python tools/mae_bench.py 1024 2>&1 | grep "B=" | tee -a $OUT/ab.txt )
  done
done
