#!/bin/bash
TAG=${1:-r3l}; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp; R=$PWD
timeout 2400 python -m pytest tests -m gpu -q -n 3 --timeout 900 -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "gpu tests exit $?" | tee -a $OUT/summary.txt; tail -3 $OUT/pytest_gpu.log
( timeout 200 python tools/train_bench.py 1024 ) 2>&1 | grep "B=" | tee $OUT/train_bench.txt
timeout 200 python tools/mae_bench.py 1024 2>&1 | grep "B=" | tee $OUT/mae_bench.txt
