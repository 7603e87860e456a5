#!/bin/bash
R=$PWD
for v in late late_rot; do
echo "== $v"; MDT_HIP_LIB=$R/mdt_policy_amd/csrc/libmdt_hip_${v}_dbg.so ROWS_LANDED=1 python tools/attn_xattn_phases.py 256 2>&1 | grep -v amdgpu.ids | head -9
done
for rep in 1 2; do
for v in late late_rot; do
echo "$v"; MDT_HIP_LIB=$R/mdt_policy_amd/csrc/libmdt_hip_$v.so python tools/latency.py 256 2>&1 | grep -v amdgpu.ids
done; done
