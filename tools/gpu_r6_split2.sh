#!/bin/bash
# round 6: K = 192 split form (masked-image head): parity, head A/B
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_mae.py -m gpu -q -k "split_three_ways or weight_stationary or swiglu" 2>&1 | tail -15 > gpurun_out/split2_tests.txt
for v in 0 1; do
  echo "== MDT_HIP_WS_SPLIT=$v" >> gpurun_out/split2_mae.txt
  MDT_HIP_WS_SPLIT=$v timeout 300 python tools/mae_bench.py 1024 2>&1 | tail -2 >> gpurun_out/split2_mae.txt
done
cat gpurun_out/split2_tests.txt gpurun_out/split2_mae.txt
