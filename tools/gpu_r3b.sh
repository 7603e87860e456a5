#!/bin/bash
# round 3: full GPU test suite + full bench line
TAG=${1:-r3b}; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp; R=$PWD
echo "== gpu tests"; timeout 2400 python -m pytest tests -m gpu -q -n 3 --timeout 900 -p no:cacheprovider --durations=15 > $OUT/pytest_gpu.log 2>&1; echo "gpu tests exit $?" | tee -a $OUT/summary.txt; tail -40 $OUT/pytest_gpu.log
echo "== bench"; timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench exit $?" | tee -a $OUT/summary.txt; tail -12 $OUT/bench.err; python -c "
import json;d=json.load(open('$OUT/bench.json'));print({k:d[k] for k in ('value','ms_per_step','median_ms','p10_ms','p90_ms')}, d['roofline']['frac'], d['roofline']['dominant_kernel']); print([ (l.get('threads'), l.get('value')) for l in d['cpu_baseline']['legs']]); print({k:(v.get('ms_per_step') or v.get('ms_per_chunk')) for k,v in d['other_configs'].items() if isinstance(v,dict)})"
