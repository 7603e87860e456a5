#!/bin/bash
# k_mlp wave skew: phase stamps of both waves of a SIMD (debug build) and the launch alone with deeper weight rings
TAG=${1:-skew2}; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp; R=$PWD
for v in 0 6; do MDT_HIP_MLP_SKEW=$v MDT_HIP_LIB=$R/mdt_policy_amd/csrc/libmdt_hip_dbg.so timeout 100 python tools/mlp_phases.py 2>&1 | grep -v amdgpu.ids; done | tee $OUT/phases.txt
for rep in 1 2; do for lib in libmdt_hip.so libmdt_hip_r4.so libmdt_hip_r5.so; do for v in 0 4 6; do
  echo -n "$lib skew $v: "; MDT_HIP_LIB=$R/mdt_policy_amd/csrc/$lib MDT_HIP_MLP_SKEW=$v timeout 200 python -c "
import torch, bench
d = torch.device('cuda'); r = bench.time_dominant_kernel(d, 2560); print('k_mlp %.2f us' % r['avg_us'])" 2>&1 | tail -1
done; done; done | tee $OUT/ab.txt
