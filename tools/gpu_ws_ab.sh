#!/bin/bash
# weight-stationary body A/B on the masked-image head: tests, then alternating (12-wave shape | 8-wave shape | 32-row tiles), then kernel stats
TAG=${1:-wsab}; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp; R=$PWD
timeout 900 python -m pytest tests/test_mae.py tests/test_gpu_ops.py tests/test_c3_step.py -m gpu -x -q -k "swiglu or weight_stationary or c3 or mae" 2>&1 | tail -2
for rep in 1 2; do
  echo -n "ws12: "; timeout 200 python tools/mae_bench.py 1024 2>&1 | grep "B=" | tr '\n' ' '; echo
  echo -n "ws8 : "; MDT_HIP_WS_WAVES=8 timeout 200 python tools/mae_bench.py 1024 2>&1 | grep "B=" | tr '\n' ' '; echo
  echo -n "rows: "; MDT_HIP_WS=0 timeout 200 python tools/mae_bench.py 1024 2>&1 | grep "B=" | tr '\n' ' '; echo
done | tee $OUT/ab.txt
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/mae -o mae -- python $R/tools/mae_bench.py 1024 > $OUT/mae_run.txt 2> $OUT/mae.err )
DB=$(find $OUT/mae -name "*.db" | head -1); python tools/prof_summary.py $DB > $OUT/mae_kernel_stats.txt; grep "k_gemm_ws" $OUT/mae_kernel_stats.txt | cut -c1-150
find $OUT -name "*.db" -delete
