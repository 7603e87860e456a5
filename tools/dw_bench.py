"""Weight-gradient products (k_gemm_tn + the sum of its slices) of the training shapes, through mdt_op_linear_bwd with dX = NULL.
usage: [MDT_HIP_TN_WIDE=0|1] python tools/dw_bench.py"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from mdt_policy_amd import _lib

lib = _lib.load()
s = torch.cuda.current_stream().cuda_stream
shapes = [(104448, 1536, 192), (104448, 192, 768), (104448, 576, 192), (104448, 192, 192), (104448, 192, 1536), (104448, 192, 576), (12288, 1536, 384), (12288, 384, 1536),
          (12288, 1152, 384), (12288, 384, 384), (4096, 1536, 384), (4096, 384, 384)]
for (M, N, K) in shapes:
    X = torch.randn(M, K, device="cuda")
    dY = torch.randn(M, N, device="cuda")
    dW = torch.empty(N, K, device="cuda")
    db = torch.empty(N, device="cuda")
    scratch = torch.empty(max(1, lib.mdt_op_linear_bwd_scratch(M, N, K)), device="cuda")
    a = _lib.LinearBwdArgs(X=X.data_ptr(), ldx=K, dY=dY.data_ptr(), ldy=N, Wt=None, dW=dW.data_ptr(), dbias=db.data_ptr(), dX=None, ldxo=K,
                           accumulate_dw=0, accumulate_dx=0, M=M, N=N, K=K, scratch=scratch.data_ptr())
    for _ in range(3):
        _lib.check(lib.mdt_op_linear_bwd(C.byref(a), s))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        _lib.check(lib.mdt_op_linear_bwd(C.byref(a), s))
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 10 * 1e3
    ref = dY[:4096].double().T @ X[:4096].double() if M <= 4096 else None
    err = "" if ref is None else f"  max err {float((dW.double() - ref).abs().max()):.2e}"
    print(f"M={M:6d} N={N:5d} K={K:5d}: {us:8.1f} us  {2.0 * M * N * K / us / 1e6:6.1f} TFLOP/s{err}")
