"""Per-workgroup phase timing of the GEMM kernel (needs the -DMDT_DEBUG_TIMING build: MDT_HIP_LIB=.../libmdt_hip_dbg.so)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from mdt_policy_amd import _lib
lib = _lib.load()
lib.mdt_debug_set_timing_buffer.argtypes = [C.c_void_p]
exec(open(os.path.join(os.path.dirname(__file__), "gemm_micro.py")).read().split("tot_ideal = tot = 0")[0].split("reps = int")[0].split("lib = _lib.load()")[1] if False else "")
dev = torch.device("cuda"); M, D = 2560, 384
g = torch.Generator().manual_seed(0); s = torch.cuda.current_stream().cuda_stream
def packed(N, K):
    W = (torch.randn(N, K, generator=g) * 0.05).to(dev); P = torch.zeros(N * K, device=dev)
    _lib.check(lib.mdt_op_pack_weight(W.data_ptr(), N, K, P.data_ptr(), 0, N, s)); return P
y = torch.randn(M, D, generator=g).to(dev); hid = torch.randn(M, 4 * D, generator=g).to(dev)
att = torch.randn(M, D, generator=g).to(dev); qkv = torch.empty(M, 3 * D, device=dev); qx = torch.empty(M, D, device=dev)
lw = torch.ones(D, device=dev); lb = torch.zeros(D, device=dev); mod = torch.randn(6 * D, generator=g).to(dev); bq = torch.zeros(3 * D, device=dev)
def args(A, lda, P, out, ldo, N, K, **kw):
    a = _lib.GemmArgs(); a.A, a.lda, a.Wp, a.out, a.ldo, a.M, a.N, a.K = A.data_ptr(), lda, P.data_ptr(), out.data_ptr(), ldo, M, N, K
    a.shift_off = a.scale_off = a.gate_off = -1; a.rows_per_sample = 10; a.gin = a.gout = 1
    for k, v in kw.items(): setattr(a, k, v)
    return a
shapes = {
 "qkv": args(y, D, packed(3 * D, D), qkv, 3 * D, 3 * D, D, ln=1, ln_w=lw.data_ptr(), mod=mod.data_ptr(), shift_off=0, scale_off=D, bias=bq.data_ptr()),
 "proj": args(att, D, packed(D, D), y, D, D, D, residual=1, mod=mod.data_ptr(), gate_off=2 * D),
 "xq": args(y, D, packed(D, D), qx, D, D, D, ln=1, ln_w=lw.data_ptr(), ln_b=lb.data_ptr(), bias=bq.data_ptr()),
 "fc": args(y, D, packed(4 * D, D), hid, 4 * D, 4 * D, D, ln=1, ln_w=lw.data_ptr(), mod=mod.data_ptr(), shift_off=3 * D, scale_off=4 * D, act=1),
 "proj2": args(hid, 4 * D, packed(D, 4 * D), y, D, D, 4 * D, residual=1, mod=mod.data_ptr(), gate_off=5 * D),
}
buf = torch.zeros(8192 * 8, dtype=torch.int64, device=dev)
for name, a in shapes.items():
    for _ in range(3): _lib.check(lib.mdt_op_gemm(C.byref(a), s))
    torch.cuda.synchronize(); buf.zero_(); torch.cuda.synchronize()
    assert lib.mdt_debug_set_timing_buffer(buf.data_ptr()) == 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); _lib.check(lib.mdt_op_gemm(C.byref(a), s)); e1.record(); torch.cuda.synchronize()
    lib.mdt_debug_set_timing_buffer(None)
    t = buf.cpu().numpy().reshape(-1, 8); t = t[t[:, 0] != 0]
    hw = t[:, 7]; xcc = (hw >> 32) & 0xF; cu = (hw >> 8) & 0xF; se = (hw >> 13) & 0x7; sh = (hw >> 12) & 1
    cuid = xcc * 1000 + se * 100 + sh * 16 + cu
    t0 = t[:, 0]; base = t0.min()
    ph = {"prologue(load+LN+lds)": t[:, 1] - t[:, 0], "barrier": t[:, 2] - t[:, 1], "mainloop": t[:, 3] - t[:, 2], "epilogue": t[:, 4] - t[:, 3], "total": t[:, 4] - t[:, 0]}
    print(f"== {name}: {len(t)} WGs, event {e0.elapsed_time(e1)*1e3:.1f} us, span {(t[:,4].max()-base)} clk, distinct CUs {len(set(cuid.tolist()))}, WGs/CU max {np.bincount(np.unique(cuid, return_inverse=True)[1]).max()}")
    for k, v in ph.items():
        print(f"   {k:24s} mean {v.mean():9.0f}  p10 {np.percentile(v,10):9.0f}  p50 {np.percentile(v,50):9.0f}  p90 {np.percentile(v,90):9.0f}  max {v.max():9.0f}")
    st = t0 - base
    print(f"   start offset             mean {st.mean():9.0f}  p50 {np.percentile(st,50):9.0f}  p90 {np.percentile(st,90):9.0f}  max {st.max():9.0f}")
    for x in range(8):
        sel = xcc == x
        if sel.any(): print(f"   xcc{x}: n={sel.sum():4d} start[{(t0[sel]-base).min():7d},{(t0[sel]-base).max():7d}] end max {(t[sel,4]-base).max():7d}", end="")
    print()
