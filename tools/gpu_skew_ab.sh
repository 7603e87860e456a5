#!/bin/bash
# k_mlp wave skew (MDT_HIP_MLP_SKEW = k-steps between the two waves of a SIMD; 0 = lockstep + barrier): parity, the launch
# alone, the B = 256 sampler call.   usage: tools/gpu_skew_ab.sh <tag> [skew values]
TAG=${1:-skew}; shift; VALS=${@:-0 6}; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
tools/micro/bin/simd_probe | tee $OUT/simd.txt
timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q -k "mlp or slab or fused" --timeout 300 -p no:cacheprovider > $OUT/pytest.log 2>&1; echo "tests exit $?"; tail -2 $OUT/pytest.log
for rep in 1 2; do for v in $VALS; do
  echo -n "skew $v: "; MDT_HIP_MLP_SKEW=$v timeout 200 python -c "
import torch, bench
d = torch.device('cuda'); r = bench.time_dominant_kernel(d, 2560); print('k_mlp %.2f us' % r['avg_us'], end='   ')" 2>&1 | tail -1
  MDT_HIP_MLP_SKEW=$v timeout 200 python tools/latency.py 256 2>&1 | grep "B=" ; done; done | tee $OUT/ab.txt
