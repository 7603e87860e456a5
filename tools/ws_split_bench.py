"""Weight-stationary K = 384 products of the B = 1024 training step: the fp32 MFMA body against its three-way bf16 split
(mdt_ws.h), per launch, back to back (HIP events).  python tools/ws_split_bench.py"""
import ctypes as C
import math
import sys

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from mdt_policy_amd import _lib as lib  # noqa: E402

L = lib.load()
K = 384
s = torch.cuda.current_stream().cuda_stream
for M, N, hooks in [(12288, 384, 0), (13312, 384, 0), (12288, 1152, 0), (12288, 1536, 1), (12288, 1536, 2), (13312, 1536, 1)]:
    g = torch.Generator().manual_seed(0)
    A = torch.randn(M, K, generator=g).cuda()
    W = torch.randn(N, K, generator=g) / math.sqrt(K)
    P = torch.empty(N * K, device="cuda")
    Wd = W.cuda().contiguous()
    lib.check(L.mdt_op_pack_weight(Wd.data_ptr(), N, K, P.data_ptr(), 0, N, s))
    out = torch.empty(M, N, device="cuda")
    aux = torch.randn(M, N, device="cuda")
    b = torch.randn(N, device="cuda")
    a = lib.GemmArgs()
    a.A, a.lda, a.Wp, a.out, a.ldo, a.M, a.N, a.K = A.data_ptr(), K, P.data_ptr(), out.data_ptr(), N, M, N, K
    a.bias = b.data_ptr()
    a.shift_off = a.scale_off = a.gate_off = -1
    a.rows_per_sample = a.gin = a.gout = 1
    if hooks:
        a.act, a.aux, a.aux_mode = lib.ACT["gelu"], aux.data_ptr(), hooks
    res = {}
    for split in (0, 1):
        L.mdt_op_set_ws_split(split)
        for _ in range(5):
            lib.check(L.mdt_op_gemm(C.byref(a), s))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            lib.check(L.mdt_op_gemm(C.byref(a), s))
        e1.record()
        torch.cuda.synchronize()
        res[split] = e0.elapsed_time(e1) / 50 * 1e3
    fl = 2.0 * M * N * K
    print(f"M={M} N={N} hooks={hooks}: fp32 {res[0]:.1f} us ({fl / res[0] / 1e6:.1f} TF)  split {res[1]:.1f} us ({fl / res[1] / 1e6:.1f} TF fp32-equivalent)")
L.mdt_op_set_ws_split(-1)
