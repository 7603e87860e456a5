#!/bin/bash
# k_mlp skew values / MFMA-loop priority (MDT_HIP_MLP_SKEW = k-steps | 256 for the priority), B = 256 sampler call
TAG=${1:-skew3}; shift; VALS=${@:-0 6 262}; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp; R=$PWD
for rep in 1 2; do for v in $VALS; do
  echo -n "skew $v: "; MDT_HIP_MLP_SKEW=$v timeout 200 python tools/latency.py 256 2>&1 | grep "B="
done; done | tee $OUT/ab.txt
