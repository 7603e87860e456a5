#!/bin/bash
mkdir -p gpurun_out/attn_ab
{
MDT_HIP_LIB=$PWD/mdt_policy_amd/csrc/libmdt_hip_dbg.so python tools/mae_attn_phases.py
MDT_HIP_ATTN_FWD=1 MDT_HIP_LIB=$PWD/mdt_policy_amd/csrc/libmdt_hip_dbg.so python tools/mae_attn_phases.py
python - <<'PY'
import torch
from mdt_policy_amd import _lib
lib=_lib.load()
def run(B,H,hd,T,tag):
    D=H*hd
    qkv=torch.randn(B,T,3*D,device="cuda"); out=torch.empty(B,T,D,device="cuda")
    s=torch.cuda.current_stream().cuda_stream
    for _ in range(3): _lib.check(lib.mdt_op_attn_mid_fwd(qkv.data_ptr(),3*D,out.data_ptr(),D,B,H,hd,T,hd**-0.5,s))
    e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): _lib.check(lib.mdt_op_attn_mid_fwd(qkv.data_ptr(),3*D,out.data_ptr(),D,B,H,hd,T,hd**-0.5,s))
    e1.record(); torch.cuda.synchronize()
    print(f"{tag} B={B} H={H} hd={hd} T={T}: {e0.elapsed_time(e1)/20*1e3:.1f} us")
run(1024,8,24,102,"heads interleaved")
run(8192,1,24,102,"one head per sample (contiguous rows)")
run(1024,8,24,96,"T=96")
run(1024,8,24,112,"T=112")
run(1024,8,32,102,"hd=32")
PY
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/attn_ab/probe.txt
