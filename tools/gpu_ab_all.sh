#!/bin/bash
# alternating A/B of library variants on the three headline timings; usage: tools/gpu_ab_all.sh <tag> <name> [<name> ...]  ("base" = shipped)
TAG=$1; shift; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp; R=$PWD
[ -n "$SKIP_TESTS" ] || timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
for rep in $(seq ${REPS:-2}); do for name in "$@"; do
  lib=$R/mdt_policy_amd/csrc/libmdt_hip_$name.so; [ "$name" = base ] && lib=$R/mdt_policy_amd/csrc/libmdt_hip.so
  echo -n "$name: "; MDT_HIP_LIB=$lib timeout 300 python tools/latency.py 1 256 2>&1 | grep "B=" | sed 's/ms\/call pipelined.*$/ms/' | tr '\n' ' '
  MDT_HIP_LIB=$lib MDT_TRAIN_BENCH_MODES=train timeout 200 python tools/train_bench.py 1024 2>&1 | grep "B=" | sed 's/ms\/step.*$/ms train/' | tr '\n' ' '
  MDT_HIP_LIB=$lib timeout 200 python tools/mae_bench.py 1024 2>&1 | grep "B=" | sed 's/ms forward.*$/ms head/' | tr '\n' ' '; echo
done; done | tee $OUT/ab.txt
