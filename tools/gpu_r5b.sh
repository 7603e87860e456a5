#!/bin/bash
# round 5: SwishGLU products on the tall body -- tests, then the masked-image head with / without (alternating)
TAG=${1:-r5b}; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_mae.py tests/test_c3_step.py -m gpu -x -q 2>&1 | tail -4
for rep in 1 2 3; do
  echo -n "glu tall: "; timeout 200 python tools/mae_bench.py 128 1024 2>&1 | grep "B=" | tr '\n' ' '; echo
  echo -n "glu rows: "; MDT_HIP_GLU_TALL=0 timeout 200 python tools/mae_bench.py 128 1024 2>&1 | grep "B=" | tr '\n' ' '; echo
done | tee $OUT/ab.txt
