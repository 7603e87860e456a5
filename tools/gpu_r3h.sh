#!/bin/bash
TAG=${1:-r3h}; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp; R=$PWD
timeout 600 python -m pytest tests/test_train_grads.py -m gpu -q --timeout 600 -p no:cacheprovider -k "torch_caches" > $OUT/pytest_mem.log 2>&1; echo "mem test exit $?" | tee -a $OUT/summary.txt; tail -6 $OUT/pytest_mem.log
timeout 2400 python -m pytest tests -m gpu -q -n 3 --timeout 900 -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "gpu tests exit $?" | tee -a $OUT/summary.txt; tail -3 $OUT/pytest_gpu.log
timeout 300 python tools/latency.py 1 2 4 8 2>&1 | grep B= | tee $OUT/sweep.txt
