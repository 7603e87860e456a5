#!/bin/bash
# A/B of the weight-fragment ring depth (builds with -DMDT_RING_ADD=n in tools/micro/exp/): GEMM microbench + sampler
OUT=$PWD/gpurun_out/${1:-ring}; mkdir -p $OUT
for v in base 1 2 3; do
  if [ $v = base ]; then unset MDT_HIP_LIB; else export MDT_HIP_LIB=$PWD/tools/micro/exp/libmdt_ring$v.so; fi
  echo "=== ring +$v" | tee -a $OUT/ring.txt
  M=2560 timeout 120 python tools/gemm_micro.py 50 2>&1 | grep -v "^---" | tee -a $OUT/ring.txt
  M=10240 timeout 120 python tools/gemm_micro.py 30 2>&1 | grep "fc \|block total" | tee -a $OUT/ring.txt
  timeout 120 python tools/latency.py 256 2>&1 | tail -1 | tee -a $OUT/ring.txt
done
