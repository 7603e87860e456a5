#!/bin/bash
# round 3: full GPU suite, latency sweep, bench (short), MGF head timing
TAG=${1:-r3e}; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp; R=$PWD
timeout 2400 python -m pytest tests -m gpu -q -n 3 --timeout 900 -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "gpu tests exit $?" | tee -a $OUT/summary.txt; tail -12 $OUT/pytest_gpu.log
timeout 300 python tools/latency.py 1 2 4 8 16 256 2>&1 | grep B= | tee $OUT/sweep.txt
for rep in 1 2; do timeout 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline > $OUT/bench_$rep.json 2>/dev/null; python -c "import json;d=json.load(open('$OUT/bench_$rep.json'));print(d['value'], d['median_ms'], d['p10_ms'], d['p90_ms'], d['roofline']['frac'], d['roofline']['dominant_kernel']['avg_us'])"; done
timeout 300 python tools/mae_bench.py 128 1024 2>&1 | grep "B=" | tee $OUT/mae_bench.txt
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof -o bench -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/prof_bench.json 2> $OUT/prof.err )
DB=$(find $OUT/prof -name "*.db" | head -1); python tools/prof_summary.py $DB > $OUT/bench_kernel_stats.txt; head -14 $OUT/bench_kernel_stats.txt | cut -c1-150
( cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats -d $OUT/lat -o lat -- python $R/tools/latency.py 1 > $OUT/lat_run.txt 2> $OUT/lat.err )
DB=$(find $OUT/lat -name "*.db" | head -1); python tools/prof_summary.py $DB > $OUT/lat_kernel_stats.txt; python tools/prof_gaps.py $DB 250 > $OUT/lat_gaps.txt; head -12 $OUT/lat_gaps.txt | cut -c1-150
find $OUT -type f -size +20M -delete
