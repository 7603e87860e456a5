"""Plain-prologue GEMM shapes of the B = 1024 training step under every tiled geometry (which tile wins where).
usage: python tools/gemm_train_shapes.py"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mdt_policy_amd import _lib
lib = _lib.load()
dev = torch.device("cuda")
s = torch.cuda.current_stream().cuda_stream
g = torch.Generator().manual_seed(0)
def run(M, N, K, geos):
    W = (torch.randn(N, K, generator=g) * 0.05).to(dev)
    P = torch.zeros(N * K, device=dev)
    _lib.check(lib.mdt_op_pack_weight(W.data_ptr(), N, K, P.data_ptr(), 0, N, s))
    A = torch.randn(M, K, generator=g).to(dev); out = torch.empty(M, N, device=dev)
    a = _lib.GemmArgs(); a.A, a.lda, a.Wp, a.out, a.ldo, a.M, a.N, a.K = A.data_ptr(), K, P.data_ptr(), out.data_ptr(), N, M, N, K
    a.shift_off = a.scale_off = a.gate_off = -1; a.rows_per_sample = 1; a.gin = a.gout = 1
    res = []
    for geo in geos:
        lib.mdt_op_set_gemm_geometry(geo)
        for _ in range(5): _lib.check(lib.mdt_op_gemm(C.byref(a), s))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): _lib.check(lib.mdt_op_gemm(C.byref(a), s))
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 20
        res.append(f"g{geo} {us:6.1f}us {2*M*N*K/us/1e6:5.1f}TF")
    lib.mdt_op_set_gemm_geometry(0)
    print(f"M={M:6d} N={N:5d} K={K:5d}: " + " | ".join(res), flush=True)
geos = [int(x) for x in os.environ.get("GEOS", "0,1,2,3,5,7,8,9").split(",")]
shapes = [(10240, 384, 384), (10240, 1152, 384), (10240, 1536, 384), (10240, 384, 1536), (104448, 576, 192), (104448, 192, 768)] if os.environ.get("SHORT") else None
for (M, N, K) in shapes or [(10240, 384, 384), (10240, 1152, 384), (10240, 1536, 384), (10240, 384, 1536), (4096, 384, 384), (4096, 1152, 384), (4096, 1536, 384), (4096, 384, 1536),
                  (104448, 192, 192), (104448, 576, 192), (104448, 1536, 192), (104448, 192, 768), (104448, 192, 576), (104448, 192, 1536)]:
    run(M, N, K, geos)
