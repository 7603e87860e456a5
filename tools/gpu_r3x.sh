#!/bin/bash
TAG=${1:-r3x}; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp; R=$PWD
timeout 900 python -m pytest tests -m gpu -q -x --timeout 600 -p no:cacheprovider -k "g3_b256 or golden or persist" -n 3 > $OUT/pytest_sel.log 2>&1; echo "selected tests exit $?" | tee -a $OUT/summary.txt; tail -5 $OUT/pytest_sel.log
for rep in 1 2; do
echo "head in the next step's first GEMM"; timeout 300 python tools/latency.py 128 256 512 2>&1 | grep -v amdgpu.ids | tee -a $OUT/lat.txt
echo "head launch"; MDT_HIP_HEAD_FUSE=0 timeout 300 python tools/latency.py 128 256 512 2>&1 | grep -v amdgpu.ids | tee -a $OUT/lat_nohead.txt
done
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof -o t -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_prof.json 2>$OUT/prof.err )
DB=$(find $OUT/prof -name "*.db" | head -1); python tools/prof_summary.py $DB > $OUT/bench_kernel_stats.txt; head -12 $OUT/bench_kernel_stats.txt | cut -c1-150
