#!/bin/bash
TAG=${1:-r5h}; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp; R=$PWD
for V in base nostoreu; do
  lib=$R/mdt_policy_amd/csrc/libmdt_hip_$V.so; [ "$V" = base ] && lib=$R/mdt_policy_amd/csrc/libmdt_hip.so
  ( cd /tmp && MDT_HIP_LIB=$lib MDT_HIP_WS_WAVES=8 timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/mae$V -o mae -- python $R/tools/mae_bench.py 1024 > $OUT/mae_run$V.txt 2> $OUT/mae$V.err )
  DB=$(find $OUT/mae$V -name "*.db" | head -1); python tools/prof_summary.py $DB > $OUT/mae_kernel_stats$V.txt; echo $V; grep "k_gemm_ws\|total kernel" $OUT/mae_kernel_stats$V.txt | cut -c1-150
done
find $OUT -name "*.db" -delete
