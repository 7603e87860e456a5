#!/bin/bash
# one sample per workgroup through attention + projection + cross-attention (k_attn_xattn): parity, then A/B against the two launches
TAG=${1:-r3q}; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp; R=$PWD
timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q -x --timeout 300 -p no:cacheprovider -k "one_sample_per_workgroup or projection_prologue or collapsed" > $OUT/pytest_op.log 2>&1; echo "op tests exit $?" | tee -a $OUT/summary.txt; tail -15 $OUT/pytest_op.log
for rep in 1 2; do
  echo "fused (default)"; timeout 300 python tools/latency.py 64 128 192 256 2>&1 | tee -a $OUT/lat_fused.txt
  echo "two launches"; MDT_HIP_ATTN_XATTN_MIN=0 timeout 300 python tools/latency.py 64 128 192 256 2>&1 | tee -a $OUT/lat_two.txt
done
echo "fused from 1 row on"; MDT_HIP_ATTN_XATTN_MIN=1 timeout 300 python tools/latency.py 16 32 64 128 2>&1 | tee -a $OUT/lat_fused_all.txt
echo "fused up to B = 1024"; MDT_HIP_ATTN_XATTN_MAX_B=1024 timeout 300 python tools/latency.py 512 1024 2>&1 | tee -a $OUT/lat_fused_big.txt
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof -o t -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_prof.json 2>$OUT/prof.err )
DB=$(find $OUT/prof -name "*.db" | head -1); python tools/prof_summary.py $DB > $OUT/bench_kernel_stats.txt; head -12 $OUT/bench_kernel_stats.txt | cut -c1-150
timeout 1500 python -m pytest tests -m gpu -q -n 3 --timeout 900 -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "gpu tests exit $?" | tee -a $OUT/summary.txt; tail -3 $OUT/pytest_gpu.log; grep "^FAILED" $OUT/pytest_gpu.log | head
