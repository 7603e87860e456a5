#!/bin/bash
# all GPU tests, then the three headline timings (sampler latencies, denoiser training step, masked-image head)
TAG=${1:-chk}; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
timeout 300 python tools/latency.py 1 256 2>&1 | grep "B=" | tee $OUT/lat.txt
MDT_TRAIN_BENCH_MODES=train timeout 200 python tools/train_bench.py 1024 2>&1 | grep "B=" | tee $OUT/train.txt
timeout 200 python tools/mae_bench.py 1024 2>&1 | grep "B=" | tee $OUT/mae.txt
