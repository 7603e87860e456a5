#!/bin/bash
# round 6: the three-way bf16 split of the weight-stationary body: parity, per-launch time, training step A/B
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_ops.py -q -k "split_three_ways or weight_stationary" 2>&1 | tail -15 > gpurun_out/split_tests.txt
timeout 300 python tools/ws_split_bench.py > gpurun_out/split_bench.txt 2>&1
for v in 0 1; do
  echo "== MDT_HIP_WS_SPLIT=$v" >> gpurun_out/split_train.txt
  MDT_HIP_WS_SPLIT=$v MDT_TRAIN_BENCH_OPT=fused MDT_TRAIN_BENCH_MODES=train timeout 300 python tools/train_bench.py 1024 2>&1 | tail -4 >> gpurun_out/split_train.txt
done
cat gpurun_out/split_tests.txt gpurun_out/split_bench.txt gpurun_out/split_train.txt
