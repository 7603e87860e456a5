#!/bin/bash
# round-6 A/B of the training step: usage: tools/gpu_r6_ab.sh <tag> "<pytest -k expr or empty>" "ENV1=a ENV2=b|ENV1=c|..." [reps]
TAG=${1:-r6ab}; K=${2:-}; SETS=${3:-"|"}; REPS=${4:-2}; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
if [ -n "$K" ]; then timeout 1500 python -m pytest tests -m gpu -x -q -k "$K" 2>&1 | tail -6 | tee $OUT/pytest.txt; fi
IFS='|' read -ra ARR <<< "$SETS"
for r in $(seq 1 $REPS); do
  for S in "${ARR[@]}"; do
    echo "== [$S]" | tee -a $OUT/ab.txt
    ( export $S MDT_TRAIN_BENCH_MODES=train; timeout 200 python tools/train_bench.py 1024 2>&1 | grep "B=" | tee -a $OUT/ab.txt )
  done
done
