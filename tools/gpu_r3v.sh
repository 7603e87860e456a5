#!/bin/bash
TAG=${1:-r3v}; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp; R=$PWD
MDT_HIP_LIB=$R/mdt_policy_amd/csrc/libmdt_hip_dbg.so ROWS_LANDED=1 timeout 100 python tools/attn_xattn_phases.py 256 2>&1 | grep -v amdgpu.ids | tee $OUT/phases.txt
for rep in 1 2; do timeout 300 python tools/latency.py 1 2 4 8 16 32 64 128 192 256 512 1024 2>&1 | grep -v amdgpu.ids | tee -a $OUT/lat.txt; done
echo "fused from 1 row on"; MDT_HIP_ATTN_XATTN_MIN=1 timeout 300 python tools/latency.py 16 32 64 96 128 2>&1 | grep -v amdgpu.ids | tee -a $OUT/lat_fused_all.txt
echo "fused up to 1024"; MDT_HIP_ATTN_XATTN_MAX_B=1024 timeout 300 python tools/latency.py 512 1024 2>&1 | grep -v amdgpu.ids | tee -a $OUT/lat_fused_big.txt
timeout 1500 python -m pytest tests -m gpu -q -n 3 --timeout 900 -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "gpu tests exit $?" | tee -a $OUT/summary.txt; tail -3 $OUT/pytest_gpu.log; grep "^FAILED" $OUT/pytest_gpu.log | head
