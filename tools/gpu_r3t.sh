#!/bin/bash
R=$PWD
echo "== requests of U / Wf at entry (shipped)"; MDT_HIP_LIB=$R/mdt_policy_amd/csrc/libmdt_hip_dbg.so ROWS_LANDED=1 python tools/attn_xattn_phases.py 256 2>&1 | grep -v amdgpu.ids
echo "== requests of U / Wf after the attention"; MDT_HIP_LIB=$R/mdt_policy_amd/csrc/libmdt_hip_late_dbg.so ROWS_LANDED=1 python tools/attn_xattn_phases.py 256 2>&1 | grep -v amdgpu.ids
for rep in 1 2; do
echo "shipped"; python tools/latency.py 256 2>&1 | grep -v amdgpu.ids
echo "late"; MDT_HIP_LIB=$R/mdt_policy_amd/csrc/libmdt_hip_late.so python tools/latency.py 256 2>&1 | grep -v amdgpu.ids
done
