#!/bin/bash
# the full GPU test suite, then alternating sampler latencies of the shipped library against named variants
# usage: tools/gpu_ab_lat.sh <tag> "<batches>" <name> [<name> ...]   (libmdt_hip_<name>.so; "base" = the shipped library)
TAG=$1; BATCHES=$2; shift 2; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp; R=$PWD
[ -n "$SKIP_TESTS" ] || timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
for rep in $(seq ${REPS:-3}); do for name in "$@"; do
  lib=$R/mdt_policy_amd/csrc/libmdt_hip_$name.so; [ "$name" = base ] && lib=$R/mdt_policy_amd/csrc/libmdt_hip.so
  echo -n "$name: "; MDT_HIP_LIB=$lib timeout 300 python tools/latency.py $BATCHES 2>&1 | grep "B=" | sed 's/chunks\/s),.*synchronous/chunks\/s) sync/' | tr '\n' ' '; echo
done; done | tee $OUT/ab.txt
