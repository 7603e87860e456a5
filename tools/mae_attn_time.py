"""Event times of the masked-image head's attention kernels (forward, backward) on the shipped library.
usage: python tools/mae_attn_time.py [B H hd T]..."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from mdt_policy_amd import _lib

lib = _lib.load()
args = [int(x) for x in sys.argv[1:]]
shapes = [tuple(args[i:i + 4]) for i in range(0, len(args), 4)] or [(1024, 8, 24, 102), (256, 8, 24, 102), (1024, 8, 24, 51), (1024, 6, 32, 102)]
for (B, H, hd, T) in shapes:
    D = H * hd
    qkv = torch.randn(B, T, 3 * D, device="cuda")
    out = torch.empty(B, T, D, device="cuda")
    do = torch.randn(B, T, D, device="cuda")
    dqkv = torch.empty_like(qkv)
    s = torch.cuda.current_stream().cuda_stream

    def fwd():
        _lib.check(lib.mdt_op_attn_mid_fwd(qkv.data_ptr(), 3 * D, out.data_ptr(), D, B, H, hd, T, hd ** -0.5, s))

    def bwd():
        _lib.check(lib.mdt_op_attn_mid_bwd(qkv.data_ptr(), 3 * D, out.data_ptr(), D, do.data_ptr(), D, dqkv.data_ptr(), 3 * D, B, H, hd, T, hd ** -0.5, s))

    t = []
    for f in (fwd, bwd):
        for _ in range(3):
            f()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            f()
        e1.record()
        torch.cuda.synchronize()
        t.append(e0.elapsed_time(e1) / 20 * 1e3)
    print(f"B={B} H={H} hd={hd} T={T}: forward {t[0]:.1f} us, backward {t[1]:.1f} us")
