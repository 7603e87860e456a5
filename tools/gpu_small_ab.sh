#!/bin/bash
# rollout-sized batches: product vs an alternative library.   usage: tools/gpu_small_ab.sh <tag> <alt lib under mdt_policy_amd/csrc/>
TAG=${1:-small}; ALT=$2; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp; R=$PWD
for rep in 1 2; do for v in product alt; do
if [ $v = alt ]; then export MDT_HIP_LIB=$R/mdt_policy_amd/csrc/$ALT; else unset MDT_HIP_LIB; fi
echo "== $v"; LAT_DEV_SIG=1 timeout 300 python tools/latency.py 1 2 8 16 2>&1 | grep B=
done; done | tee $OUT/ab.txt
