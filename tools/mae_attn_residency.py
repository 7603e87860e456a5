"""Residency picture of the masked-image head's attention forward: start / end of every workgroup on the constant 100 MHz
clock and where it ran (-DMDT_DEBUG_TIMING build).   usage: MDT_HIP_LIB=<debug .so> python tools/mae_attn_residency.py [B H hd T [bwd]]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from mdt_policy_amd import _lib

B, H, hd, T = [int(x) for x in (sys.argv[1:5] if len(sys.argv) >= 5 else (1024, 8, 24, 102))]
lib = _lib.load()
lib.mdt_mae_debug_wg.restype = C.c_int32
lib.mdt_mae_debug_wg.argtypes = [C.c_void_p, C.c_int32]
D = H * hd
qkv = torch.randn(B, T, 3 * D, device="cuda")
out = torch.empty(B, T, D, device="cuda")
s = torch.cuda.current_stream().cuda_stream
bwd = len(sys.argv) > 5 and sys.argv[5] == "bwd"
do = torch.randn(B, T, D, device="cuda")
dqkv = torch.empty_like(qkv)
for _ in range(3):
    _lib.check(lib.mdt_op_attn_mid_fwd(qkv.data_ptr(), 3 * D, out.data_ptr(), D, B, H, hd, T, hd ** -0.5, s))
if bwd:
    for _ in range(3):
        _lib.check(lib.mdt_op_attn_mid_bwd(qkv.data_ptr(), 3 * D, out.data_ptr(), D, do.data_ptr(), D, dqkv.data_ptr(), 3 * D, B, H, hd, T, hd ** -0.5, s))
torch.cuda.synchronize()
n = min(8192, B * H)
buf = np.zeros(3 * n, dtype=np.uint64)
assert lib.mdt_mae_debug_wg(buf.ctypes.data, n) == 0
w = buf.reshape(n, 3)
w = w[w[:, 1] > 0]
t0 = w[:, 0].min()
st, en = (w[:, 0] - t0).astype(np.int64), (w[:, 1] - t0).astype(np.int64)
xcc = (w[:, 2] >> np.uint64(32)) & np.uint64(0xf)
hw = w[:, 2] & np.uint64(0xffffffff)
cu = ((hw >> np.uint64(8)) & np.uint64(0xf)) | (((hw >> np.uint64(13)) & np.uint64(0x7)) << np.uint64(4)) | (xcc << np.uint64(8))
print(f"{'backward' if bwd else 'forward'} B={B} H={H} hd={hd} T={T}: {len(w)} workgroups stamped; span {en.max() / 100:.1f} us; lifetime mean {np.mean(en - st) / 100:.1f} us "
      f"(min {np.min(en - st) / 100:.1f}, max {np.max(en - st) / 100:.1f})")
print(f"distinct (xcc, se, cu) = {len(set(cu.tolist()))}; workgroups per XCC: {np.bincount(xcc.astype(np.int64)).tolist()}")
for q in (0.1, 0.25, 0.5, 0.75, 0.9):
    t = int(en.max() * q)
    print(f"  resident at {q:4.0%} of the span: {int(np.sum((st <= t) & (en > t)))}")
print("  start times (us), sorted, every 64th:", [round(x / 100, 1) for x in np.sort(st)[::64].tolist()])
