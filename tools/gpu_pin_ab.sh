#!/bin/bash
# scheduling barriers in the GEMM k-steps (product) vs -DMDT_NO_SCHED_PIN (libmdt_hip_nopin.so), with and without the k_mlp wave skew;
# phase stamps of both waves of a SIMD from the -DMDT_DEBUG_TIMING build (libmdt_hip_dbg.so)
TAG=${1:-pin}; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp; R=$PWD
timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q -n 3 --timeout 300 -p no:cacheprovider > $OUT/pytest.log 2>&1; echo "tests exit $?"; tail -2 $OUT/pytest.log
for rep in 1 2; do for lib in libmdt_hip.so libmdt_hip_nopin.so; do for v in 0 6; do
  echo -n "$lib skew $v: "; MDT_HIP_LIB=$R/mdt_policy_amd/csrc/$lib MDT_HIP_MLP_SKEW=$v timeout 200 python -c "
import torch, bench
d = torch.device('cuda'); r = bench.time_dominant_kernel(d, 2560); print('k_mlp %.2f us' % r['avg_us'], end='   ')" 2>&1 | tail -1
  MDT_HIP_LIB=$R/mdt_policy_amd/csrc/$lib MDT_HIP_MLP_SKEW=$v timeout 200 python tools/latency.py 256 2>&1 | grep "B="
done; done; done | tee $OUT/ab.txt
for v in 0 6; do MDT_HIP_MLP_SKEW=$v MDT_HIP_LIB=$R/mdt_policy_amd/csrc/libmdt_hip_dbg.so timeout 100 python tools/mlp_phases.py 2>&1 | grep -v amdgpu.ids; done | tee $OUT/phases.txt
