#!/bin/bash
# One gpurun call: GPU parity tests (kernel level + model level), smoke, a short bench.  Logs -> gpurun_out/.
# usage: tools/gpu_check.sh [tag]
TAG=${1:-run}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
rocminfo 2>/dev/null | grep -E "Marketing Name|gfx9|Compute Unit" | head -6 > $OUT/rocminfo.txt
python -c "import os; print('cpu_count', os.cpu_count())" >> $OUT/rocminfo.txt
grep -m1 "model name" /proc/cpuinfo >> $OUT/rocminfo.txt
echo "== ops ==";    timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -q -n 2 --timeout 300 -p no:cacheprovider > $OUT/pytest_ops.log 2>&1; echo "ops exit $?" | tee -a $OUT/summary.txt
tail -5 $OUT/pytest_ops.log
echo "== parity =="; timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q -n 2 --timeout 600 -p no:cacheprovider > $OUT/pytest_parity.log 2>&1; echo "parity exit $?" | tee -a $OUT/summary.txt
tail -5 $OUT/pytest_parity.log
echo "== smoke ==";  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke exit $?" | tee -a $OUT/summary.txt
tail -3 $OUT/smoke.log
echo "== bench ==";  timeout 600 python bench.py --steps 10 --warmup 3 > $OUT/bench.log 2>&1; echo "bench exit $?" | tee -a $OUT/summary.txt
tail -3 $OUT/bench.log
