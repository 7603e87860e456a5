"""What each part of the masked-image head's attention forward costs: the kernel timed with parts switched off (results wrong;
-DMDT_DEBUG_TIMING build only).   usage: MDT_HIP_LIB=<debug .so> python tools/mae_attn_knobs.py [B H hd T]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from mdt_policy_amd import _lib

B, H, hd, T = [int(x) for x in (sys.argv[1:5] if len(sys.argv) >= 5 else (1024, 8, 24, 102))]
lib = _lib.load()
D = H * hd
qkv = torch.randn(B, T, 3 * D, device="cuda")
out = torch.empty(B, T, D, device="cuda")
s = torch.cuda.current_stream().cuda_stream
names = {0: "everything", 1: "no exp / subtraction", 2: "first head's fetch only", 4: "no stores", 8: "no P V product", 16: "no score product",
         24: "no MFMA at all", 7: "no exp, fetch, stores", 31: "nothing but commit + barriers + max"}
for knob, name in names.items():
    assert lib.mdt_mae_debug_knob(knob) == 0
    for _ in range(3):
        _lib.check(lib.mdt_op_attn_mid_fwd(qkv.data_ptr(), 3 * D, out.data_ptr(), D, B, H, hd, T, hd ** -0.5, s))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        _lib.check(lib.mdt_op_attn_mid_fwd(qkv.data_ptr(), 3 * D, out.data_ptr(), D, B, H, hd, T, hd ** -0.5, s))
    e1.record()
    torch.cuda.synchronize()
    print(f"knob {knob:2d} ({name:36s}): {e0.elapsed_time(e1) / 20 * 1e3:7.1f} us")
lib.mdt_mae_debug_knob(0)
