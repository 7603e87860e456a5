#!/bin/bash
# A/B of the dW split-K knobs (slice target / depth) on the training step (eval-mode arithmetic) and the Perceiver
OUT=$PWD/gpurun_out/${1:-splitk}; mkdir -p $OUT
run() { echo "== $*" | tee -a $OUT/ab.txt; env "$@" MDT_TRAIN_BENCH_MODES=eval timeout 120 python tools/train_bench.py $B 2>&1 | tail -1 | tee -a $OUT/ab.txt; }
for B in 1024 128; do
run A=default
run MDT_HIP_SPLIT_TARGET=400 MDT_HIP_SPLIT_DEPTH=1024
run MDT_HIP_SPLIT_TARGET=1500
run MDT_HIP_SPLIT_TARGET=1500 MDT_HIP_SPLIT_DEPTH=256
run MDT_HIP_SPLIT_TARGET=3000 MDT_HIP_SPLIT_DEPTH=256
done
echo "== perceiver default" | tee -a $OUT/ab.txt; timeout 200 python tools/perceiver_train_bench.py 2>&1 | tail -2 | tee -a $OUT/ab.txt
echo "== perceiver old" | tee -a $OUT/ab.txt; MDT_HIP_SPLIT_TARGET=400 MDT_HIP_SPLIT_DEPTH=1024 timeout 200 python tools/perceiver_train_bench.py 2>&1 | tail -2 | tee -a $OUT/ab.txt
