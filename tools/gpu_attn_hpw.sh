#!/bin/bash
# heads per workgroup of the masked-image head's attention forward (MDT_HIP_ATTN_HPW): event times, then the residency picture
for hpw in 0 1 2 4 8; do
echo "== MDT_HIP_ATTN_HPW=$hpw"
MDT_HIP_ATTN_HPW=$hpw python - <<'PY' 2>&1 | grep -v amdgpu.ids
import torch
from mdt_policy_amd import _lib
lib=_lib.load()
for (B,H,hd,T) in [(1024,8,24,102),(256,8,24,102),(1024,8,24,51)]:
    D=H*hd
    qkv=torch.randn(B,T,3*D,device="cuda"); out=torch.empty(B,T,D,device="cuda")
    s=torch.cuda.current_stream().cuda_stream
    for _ in range(3): _lib.check(lib.mdt_op_attn_mid_fwd(qkv.data_ptr(),3*D,out.data_ptr(),D,B,H,hd,T,hd**-0.5,s))
    e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): _lib.check(lib.mdt_op_attn_mid_fwd(qkv.data_ptr(),3*D,out.data_ptr(),D,B,H,hd,T,hd**-0.5,s))
    e1.record(); torch.cuda.synchronize()
    q,k,v=qkv.view(B,T,3,H,hd).permute(2,0,3,1,4)
    ref=torch.nn.functional.scaled_dot_product_attention(q,k,v).transpose(1,2).reshape(B,T,D)
    print(f"B={B} H={H} hd={hd} T={T}: {e0.elapsed_time(e1)/20*1e3:.1f} us  max err {float((out-ref).abs().max()):.2e}")
PY
done
export MDT_HIP_LIB=$PWD/mdt_policy_amd/csrc/libmdt_hip_dbg.so
for hpw in 8 2; do MDT_HIP_ATTN_HPW=$hpw python tools/mae_attn_residency.py 2>&1 | grep -v amdgpu.ids; done
