#!/bin/bash
TAG=${1:-r3p}; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp; R=$PWD
timeout 2400 python -m pytest tests -m gpu -q -n 3 --timeout 900 -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "gpu tests exit $?" | tee -a $OUT/summary.txt; tail -3 $OUT/pytest_gpu.log; grep "^FAILED" $OUT/pytest_gpu.log | head
for rep in 1 2 3; do timeout 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline > $OUT/bench_$rep.json 2>/dev/null; python -c "import json;d=json.load(open('$OUT/bench_$rep.json'));print(d['value'], d['median_ms'], d['p10_ms'], d['p90_ms'], d['roofline']['frac'], d['roofline']['dominant_kernel']['avg_us'])"; done
MDT_HIP_LIB=$R/mdt_policy_amd/csrc/libmdt_hip_dbg.so timeout 100 python tools/mlp_phases.py | tail -9
