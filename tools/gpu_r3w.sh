#!/bin/bash
TAG=${1:-r3w}; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp; R=$PWD
timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q -x --timeout 300 -p no:cacheprovider -k "one_sample_per_workgroup or collapsed" > $OUT/pytest_op.log 2>&1; echo "op tests exit $?" | tee -a $OUT/summary.txt; tail -5 $OUT/pytest_op.log
for rep in 1 2; do timeout 300 python tools/latency.py 1 8 64 128 256 512 2>&1 | grep -v amdgpu.ids | tee -a $OUT/lat.txt; done
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof -o t -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_prof.json 2>$OUT/prof.err )
DB=$(find $OUT/prof -name "*.db" | head -1); python tools/prof_summary.py $DB > $OUT/bench_kernel_stats.txt; head -24 $OUT/bench_kernel_stats.txt | cut -c1-150
timeout 1500 python -m pytest tests -m gpu -q -n 3 --timeout 900 -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "gpu tests exit $?" | tee -a $OUT/summary.txt; tail -3 $OUT/pytest_gpu.log; grep "^FAILED" $OUT/pytest_gpu.log | head
