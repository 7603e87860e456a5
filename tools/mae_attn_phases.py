"""Phase times of the masked-image head's attention kernels (workgroup (0, 0), shader-clock stamps of a
-DMDT_DEBUG_TIMING build).   usage: python -m mdt_policy_amd.build -DMDT_DEBUG_TIMING --out=/tmp/libmdt_dbg.so;
MDT_HIP_LIB=/tmp/libmdt_dbg.so python tools/mae_attn_phases.py [B H hd T]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from mdt_policy_amd import _lib

B, H, hd, T = [int(x) for x in (sys.argv[1:5] if len(sys.argv) >= 5 else (1024, 8, 24, 102))]
lib = _lib.load()
lib.mdt_mae_debug_ts.restype = C.c_int32
lib.mdt_mae_debug_ts.argtypes = [C.c_void_p]
D = H * hd
qkv = torch.randn(B, T, 3 * D, device="cuda")
out = torch.empty(B, T, D, device="cuda")
do = torch.randn(B, T, D, device="cuda")
dqkv = torch.empty_like(qkv)
s = torch.cuda.current_stream().cuda_stream
ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
for it in range(3):
    ev[0].record()
    _lib.check(lib.mdt_op_attn_mid_fwd(qkv.data_ptr(), 3 * D, out.data_ptr(), D, B, H, hd, T, hd ** -0.5, s))
    ev[1].record()
    _lib.check(lib.mdt_op_attn_mid_bwd(qkv.data_ptr(), 3 * D, out.data_ptr(), D, do.data_ptr(), D, dqkv.data_ptr(), 3 * D, B, H, hd, T, hd ** -0.5, s))
    ev[2].record()
torch.cuda.synchronize()
print(f"B={B} H={H} hd={hd} T={T}: forward {ev[0].elapsed_time(ev[1])*1e3:.1f} us, backward {ev[1].elapsed_time(ev[2])*1e3:.1f} us")
ts = (C.c_uint64 * 16)()
assert lib.mdt_mae_debug_ts(ts) == 0
t = list(ts)
names_f = ["load q k v", "row tiles of wave 0 (scores, softmax, P V)"]
t[2] = t[4]
names_b = ["load q k v dO, rowdot", "query tiles of wave 0", "barrier + partial dK / dV sums (LDS)", "write dK dV"]
print("forward, workgroup (0,0), shader clocks:")
for i, n in enumerate(names_f):
    print(f"  {n:44s} {t[i + 1] - t[i]:8d}")
print(f"  {'total':26s} {t[4] - t[0]:8d}")
print("backward:")
for i, n in enumerate(names_b):
    print(f"  {n:36s} {t[6 + i] - t[5 + i]:8d}")
print(f"  {'total':36s} {t[9] - t[5]:8d}")
