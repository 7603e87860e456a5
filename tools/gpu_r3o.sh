#!/bin/bash
# A/B: product build vs an alternative library (MDT_HIP_LIB): bench value / median / dominant kernel
TAG=${1:-r3o}; ALT=$2; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp; R=$PWD
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_parity.py -m gpu -q -n 3 --timeout 600 -p no:cacheprovider > $OUT/pytest.log 2>&1; echo "tests exit $?"; tail -2 $OUT/pytest.log
for rep in 1 2 3; do for v in product alt; do
if [ $v = alt ]; then export MDT_HIP_LIB=$R/$ALT; else unset MDT_HIP_LIB; fi
timeout 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline > $OUT/b_${v}_$rep.json 2>/dev/null; python -c "import json;d=json.load(open('$OUT/b_${v}_$rep.json'));print('$v', d['value'], d['median_ms'], d['p10_ms'], d['p90_ms'], d['roofline']['dominant_kernel']['avg_us'])"; done; done
