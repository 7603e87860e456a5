#!/bin/bash
# product vs an alternative library on everything that is benchmarked: sampler sweep, training step, MGF head
# usage: tools/gpu_all_ab.sh <tag> <alt lib under mdt_policy_amd/csrc/>
TAG=${1:-allab}; ALT=$2; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp; R=$PWD
for rep in 1 2; do for v in product alt; do
if [ $v = alt ]; then export MDT_HIP_LIB=$R/mdt_policy_amd/csrc/$ALT; else unset MDT_HIP_LIB; fi
echo "== $v"; timeout 300 python tools/latency.py 1 8 64 256 1024 2>&1 | grep B=
MDT_TRAIN_BENCH_MODES=train timeout 200 python tools/train_bench.py 1024 2>&1 | grep "B="
timeout 200 python tools/mae_bench.py 1024 2>&1 | grep "B="
done; done | tee $OUT/ab.txt
