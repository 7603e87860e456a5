#!/bin/bash
TAG=${1:-r3n}; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp; R=$PWD
timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -q --timeout 600 -p no:cacheprovider -k "attention" > $OUT/pytest_ops.log 2>&1; echo "ops exit $?"; tail -6 $OUT/pytest_ops.log
timeout 2400 python -m pytest tests -m gpu -q -n 3 --timeout 900 -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "gpu tests exit $?" | tee -a $OUT/summary.txt; tail -4 $OUT/pytest_gpu.log
for rep in 1 2 3; do for w in -1 0; do
timeout 200 python - <<PY
import json, subprocess, sys, os
os.environ["MDT_HIP_ATTN_WIDE_MIN"] = "1401" if $w == -1 else "0"
out = subprocess.run([sys.executable, "bench.py", "--steps", "30", "--warmup", "5", "--no-cpu-baseline"], capture_output=True, text=True, env=os.environ).stdout
d = json.loads([l for l in out.splitlines() if l.startswith("{")][-1])
print("wide" if $w == -1 else "two-launch", d["value"], d["median_ms"], d["p10_ms"], d["p90_ms"], d["roofline"]["frac"])
PY
done; done
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof -o bench -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/prof_bench.json 2> $OUT/prof.err )
DB=$(find $OUT/prof -name "*.db" | head -1); python tools/prof_summary.py $DB > $OUT/bench_kernel_stats.txt; head -10 $OUT/bench_kernel_stats.txt | cut -c1-150
find $OUT -type f -size +20M -delete
