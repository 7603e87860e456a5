#!/bin/bash
TAG=${1:-skew5}; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp; R=$PWD
for rep in 1 2; do for lib in libmdt_hip.so libmdt_hip_r4.so libmdt_hip_r5.so; do for v in 274 280; do
  echo -n "$lib skew $v: "; MDT_HIP_LIB=$R/mdt_policy_amd/csrc/$lib MDT_HIP_MLP_SKEW=$v timeout 200 python tools/latency.py 256 2>&1 | grep "B="
done; done; done | tee $OUT/ab.txt
for v in 277; do MDT_HIP_MLP_SKEW=$v MDT_HIP_LIB=$R/mdt_policy_amd/csrc/libmdt_hip_dbg.so timeout 100 python tools/mlp_phases.py 2>&1 | grep -v amdgpu.ids; done | tee $OUT/phases.txt
