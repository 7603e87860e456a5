"""Per-workgroup phase timing of the tall GEMM body (needs the -DMDT_DEBUG_TIMING build: MDT_HIP_LIB=.../libmdt_hip_dbg.so).
usage: GEOS=23 python tools/tall_phases.py [M N K]   (round 5 pruned geometries 10 / 12 / 16)"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from mdt_policy_amd import _lib
lib = _lib.load()
lib.mdt_debug_set_timing_buffer.argtypes = [C.c_void_p]
dev = torch.device("cuda"); s = torch.cuda.current_stream().cuda_stream
g = torch.Generator().manual_seed(0)
M, N, K = [int(x) for x in sys.argv[1:4]] if len(sys.argv) > 3 else (10240, 1536, 384)
W = (torch.randn(N, K, generator=g) * 0.05).to(dev); P = torch.zeros(N * K, device=dev)
_lib.check(lib.mdt_op_pack_weight(W.data_ptr(), N, K, P.data_ptr(), 0, N, s))
A = torch.randn(M, K, generator=g).to(dev); out = torch.empty(M, N, device=dev)
a = _lib.GemmArgs(); a.A, a.lda, a.Wp, a.out, a.ldo, a.M, a.N, a.K = A.data_ptr(), K, P.data_ptr(), out.data_ptr(), N, M, N, K
a.shift_off = a.scale_off = a.gate_off = -1; a.rows_per_sample = 1; a.gin = a.gout = 1
buf = torch.zeros(16384 * 8, dtype=torch.int64, device=dev)
for geo in [int(x) for x in os.environ.get("GEOS", "23").split(",")]:
    lib.mdt_op_set_gemm_geometry(geo)
    for _ in range(3): _lib.check(lib.mdt_op_gemm(C.byref(a), s))
    torch.cuda.synchronize(); buf.zero_(); torch.cuda.synchronize()
    assert lib.mdt_debug_set_timing_buffer(buf.data_ptr()) == 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); _lib.check(lib.mdt_op_gemm(C.byref(a), s)); e1.record(); torch.cuda.synchronize()
    lib.mdt_debug_set_timing_buffer(None)
    t = buf.cpu().numpy().reshape(-1, 8); t = t[t[:, 0] != 0]
    hw = t[:, 7]; xcc = (hw >> 32) & 0xF; cu = (hw >> 8) & 0xF; se = (hw >> 13) & 0x7; sh = (hw >> 12) & 1
    cuid = xcc * 1000 + se * 100 + sh * 16 + cu
    base = t[:, 0].min(); span = t[:, 4].max() - base
    # co-residency: for every workgroup, how many others on its CU overlap its lifetime at its midpoint
    mid = (t[:, 0] + t[:, 4]) // 2
    co = np.array([((cuid == cuid[i]) & (t[:, 0] <= mid[i]) & (t[:, 4] >= mid[i])).sum() for i in range(len(t))])
    print(f"== geometry {geo}  {M}x{N}x{K}: {len(t)} workgroups, event {e0.elapsed_time(e1)*1e3:.1f} us, span {span} clk ({span/(e0.elapsed_time(e1)*1e3):.0f} clk/us), "
          f"distinct CUs {len(set(cuid.tolist()))}, co-resident workgroups per CU (at mid-life) mean {co.mean():.2f} max {co.max()}")
    ph = {"entry -> first barrier passed": t[:, 1] - t[:, 0], "main loop (rest)": t[:, 3] - t[:, 1], "  of it at wait+barrier": t[:, 5],
          "epilogue": t[:, 4] - t[:, 3], "total": t[:, 4] - t[:, 0]}
    for k, v in ph.items():
        print(f"   {k:30s} mean {v.mean():9.0f}  p10 {np.percentile(v,10):9.0f}  p50 {np.percentile(v,50):9.0f}  p90 {np.percentile(v,90):9.0f}  max {v.max():9.0f}")
    st = t[:, 0] - base
    print(f"   start offset                   p10 {np.percentile(st,10):9.0f}  p50 {np.percentile(st,50):9.0f}  p90 {np.percentile(st,90):9.0f}  max {st.max():9.0f}")
lib.mdt_op_set_gemm_geometry(0)
