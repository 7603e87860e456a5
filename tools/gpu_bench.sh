#!/bin/bash
# One gpurun call: bench (with stderr progress log), rocprofv3 kernel stats of the same command, optional extras.
# usage: tools/gpu_bench.sh <tag> [extra pytest -k expression]
TAG=${1:-bench}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
if [ -n "$2" ]; then
  timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_ops.py -m gpu -q -n 2 --timeout 300 -p no:cacheprovider -k "$2" > $OUT/pytest_sel.log 2>&1
  echo "pytest($2) exit $?" | tee -a $OUT/summary.txt; tail -4 $OUT/pytest_sel.log
fi
echo "== bench =="; timeout 420 python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; echo "bench exit $?" | tee -a $OUT/summary.txt
cat $OUT/bench.json; tail -8 $OUT/bench.err
echo "== rocprof =="
( cd /tmp && timeout 420 rocprofv3 --kernel-trace --stats -d $OUT/prof -o bench -- python $OLDPWD/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/prof_bench.json 2> $OUT/prof.err ); echo "rocprof exit $?" | tee -a $OUT/summary.txt
find $OUT/prof -name "*stats*" | head; 
STATS=$(find $OUT/prof -name "*kernel_stats.csv" | head -1)
[ -n "$STATS" ] && head -25 "$STATS"
# keep only the summaries (traces are large)
find $OUT/prof -name "*kernel_trace.csv" -size +20M -delete
if [ "$3" = "cpu" ]; then echo "== cpu threads =="; timeout 300 python tests/perf_cpu_threads.py 256 2>&1 | tee $OUT/cpu_threads.txt; fi
