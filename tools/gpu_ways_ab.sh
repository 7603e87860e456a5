#!/bin/bash
# A/B: two half-batch launch chains on two streams (MDT_HIP_WAYS=2) with 4-wave half-width GEMM tiles, so that every CU
# holds one workgroup of each chain and one chain's prologue / epilogue overlaps the other's MFMA loop.
out=gpurun_out/$1; mkdir -p $out
for cfg in "1 0 0" "2 0 0" "1 8 0" "2 8 0" "2 8 5" "2 9 5" "1 9 0" "2 9 0" "3 8 5" "4 8 5"; do
  set -- $cfg
  echo "== WAYS=$1 GEO_WIDE=$2 GEO_NARROW=$3" >> $out/ways.txt
  MDT_HIP_WAYS=$1 MDT_HIP_GEO_WIDE=$2 MDT_HIP_GEO_NARROW=$3 timeout 120 python tools/latency.py 256 512 2>&1 | grep "B=" >> $out/ways.txt
done
cat $out/ways.txt
