#!/bin/bash
TAG=${1:-r3i}; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp; R=$PWD
timeout 2400 python -m pytest tests -m gpu -q -n 3 --timeout 900 -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "gpu tests exit $?" | tee -a $OUT/summary.txt; tail -3 $OUT/pytest_gpu.log
for rep in 1 2 3; do for ov in 1 0; do
MDT_HIP_OVERLAP=$ov timeout 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline > $OUT/bench_${ov}_$rep.json 2>/dev/null; python -c "import json;d=json.load(open('$OUT/bench_${ov}_$rep.json'));print('overlap=$ov', d['value'], d['median_ms'], d['p10_ms'], d['p90_ms'], d['roofline']['frac'])"; done; done
echo overlap on;  MDT_HIP_GRAPH=0 timeout 300 python tools/latency.py 1 8 64 2>&1 | grep B=
echo overlap off; MDT_HIP_GRAPH=0 MDT_HIP_OVERLAP=0 timeout 300 python tools/latency.py 1 8 64 2>&1 | grep B=
echo graph auto;  timeout 300 python tools/latency.py 1 8 2>&1 | grep B=
