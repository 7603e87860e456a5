#!/bin/bash
TAG=${1:-attn_xattn_ab}; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp; R=$PWD
timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q -x --timeout 300 -p no:cacheprovider -k "one_sample_per_workgroup or projection_prologue or collapsed" > $OUT/pytest_op.log 2>&1; echo "op tests exit $?" | tee -a $OUT/summary.txt; tail -5 $OUT/pytest_op.log
MDT_HIP_LIB=$R/mdt_policy_amd/csrc/libmdt_hip_dbg.so ROWS_LANDED=1 timeout 100 python tools/attn_xattn_phases.py 256 2>&1 | grep -v amdgpu.ids | tee $OUT/phases.txt
for rep in 1 2; do
  echo "fused (default)"; timeout 300 python tools/latency.py 1 128 192 256 2>&1 | grep -v amdgpu.ids | tee -a $OUT/lat_fused.txt
done
echo "fused from 1 row on"; MDT_HIP_ATTN_XATTN_MIN=1 timeout 300 python tools/latency.py 16 32 64 128 2>&1 | grep -v amdgpu.ids | tee -a $OUT/lat_fused_all.txt
echo "two launches"; MDT_HIP_ATTN_XATTN_MIN=0 timeout 300 python tools/latency.py 16 32 64 128 256 2>&1 | grep -v amdgpu.ids | tee -a $OUT/lat_two.txt
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof -o t -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_prof.json 2>$OUT/prof.err )
DB=$(find $OUT/prof -name "*.db" | head -1); python tools/prof_summary.py $DB > $OUT/bench_kernel_stats.txt; head -7 $OUT/bench_kernel_stats.txt | cut -c1-150
