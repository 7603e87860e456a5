"""Time arbitrary GEMM shapes (M,N,K[,flags];...) across geometries.  flags: l = LayerNorm prologue, g = GELU epilogue, r = residual
epilogue (no flags field: residual, as the first version of this tool).  usage: GEOS=0,1,2,6 python tools/gemm_shapes.py 1024,384,32768 1024,1152,384,l"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mdt_policy_amd import _lib
lib = _lib.load(); dev = torch.device("cuda"); s = torch.cuda.current_stream().cuda_stream
geos = [int(x) for x in os.environ.get("GEOS", "0").split(",")]
for spec in sys.argv[1:]:
    f = spec.split(",")
    M, N, K = (int(v) for v in f[:3])
    flags = f[3] if len(f) > 3 else "r"
    lw = torch.ones(K, device=dev)
    A = torch.randn(M, K, device=dev); P = torch.randn(N * K, device=dev) * 0.01; out = torch.zeros(M, N, device=dev)
    a = _lib.GemmArgs(); a.A, a.lda, a.Wp, a.out, a.ldo, a.M, a.N, a.K = A.data_ptr(), K, P.data_ptr(), out.data_ptr(), N, M, N, K
    a.shift_off = a.scale_off = a.gate_off = -1; a.rows_per_sample = 1; a.gin = a.gout = 1; a.residual = int("r" in flags)
    if "l" in flags: a.ln, a.ln_w = 1, lw.data_ptr()
    if "g" in flags: a.act = 1
    for geo in geos:
        lib.mdt_op_set_gemm_geometry(geo)
        for _ in range(2): _lib.check(lib.mdt_op_gemm(C.byref(a), s))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); n = 10
        for _ in range(n): _lib.check(lib.mdt_op_gemm(C.byref(a), s))
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / n
        print(f"M={M} N={N} K={K} {flags:3s} geo={geo}: {us:9.1f} us  {2.0 * M * N * K / us / 1e6:7.1f} TFLOP/s", flush=True)
lib.mdt_op_set_gemm_geometry(0)
