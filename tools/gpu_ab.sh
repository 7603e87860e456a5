#!/bin/bash
# Generic A/B run of experimental builds of the library against the shipped one (one gpurun call):
#   python -m mdt_policy_amd.build -DFLAG --out=mdt_policy_amd/csrc/libmdt_hip_<name>.so     (here, on the CPU box)
#   gpurun -- tools/gpu_ab.sh <tag> "<batch sizes>" <name> [<name> ...]                          ("base" = the shipped library)
# Alternates the builds REPS (default 3) times; per build and round: the dominant kernel alone, then tools/latency.py.
TAG=$1; BATCHES=$2; shift 2; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp; R=$PWD
for rep in $(seq ${REPS:-3}); do for name in "$@"; do
  lib=$R/mdt_policy_amd/csrc/libmdt_hip_$name.so; [ "$name" = base ] && lib=$R/mdt_policy_amd/csrc/libmdt_hip.so
  echo -n "$name: "; MDT_HIP_LIB=$lib timeout 200 python -c "
import torch, bench
d = torch.device('cuda'); r = bench.time_dominant_kernel(d, 2560); print('k_mlp alone %.2f us' % r['alone']['avg_us'], end='   ')" 2>&1 | tail -1
  MDT_HIP_LIB=$lib timeout 300 python tools/latency.py $BATCHES 2>&1 | grep "B=" | tr '\n' ' '; echo
done; done | tee $OUT/ab.txt
