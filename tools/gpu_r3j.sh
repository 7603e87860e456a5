#!/bin/bash
TAG=${1:-r3j}; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp; R=$PWD
for rep in 1 2 3; do for v in product prio; do
if [ $v = prio ]; then export MDT_HIP_LIB=$R/mdt_policy_amd/csrc/libmdt_hip_prio.so; else unset MDT_HIP_LIB; fi
timeout 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline > $OUT/b_${v}_$rep.json 2>/dev/null; python -c "import json;d=json.load(open('$OUT/b_${v}_$rep.json'));print('$v', d['value'], d['median_ms'], d['roofline']['dominant_kernel']['avg_us'])"; done; done
