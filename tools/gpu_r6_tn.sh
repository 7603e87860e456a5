#!/bin/bash
# round 6: the weight-gradient product as bf16 splits: parity, step and head A/B
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_train_ops.py -m gpu -q -k "weight_gradient or linear_backward" 2>&1 | tail -15 > gpurun_out/tn_tests.txt
for v in 0 1; do
  echo "== MDT_HIP_TN_SPLIT=$v" >> gpurun_out/tn_ab.txt
  MDT_HIP_TN_SPLIT=$v timeout 300 python tools/mae_bench.py 1024 2>&1 | tail -1 >> gpurun_out/tn_ab.txt
  MDT_HIP_TN_SPLIT=$v MDT_TRAIN_BENCH_OPT=fused MDT_TRAIN_BENCH_MODES=train timeout 300 python tools/train_bench.py 1024 2>&1 | tail -1 >> gpurun_out/tn_ab.txt
done
cat gpurun_out/tn_tests.txt gpurun_out/tn_ab.txt
