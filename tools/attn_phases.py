"""Phase timing of k_attn / k_xattn_apply (debug build: MDT_HIP_LIB=.../libmdt_hip_dbg.so)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from mdt_policy_amd import _lib
lib = _lib.load(); lib.mdt_debug_set_timing_buffer.argtypes = [C.c_void_p]
dev = torch.device("cuda"); s = torch.cuda.current_stream().cuda_stream
B, H, hd, D, Ta, Te = 256, 8, 48, 384, 10, 4
g = torch.Generator().manual_seed(0)
qkv = torch.randn(B * Ta, 3 * D, generator=g).to(dev); out = torch.empty(B * Ta, D, device=dev)
a = _lib.AttnArgs(); a.q, a.ldq, a.k, a.v, a.ldkv = qkv.data_ptr(), 3 * D, qkv.data_ptr() + 4 * D, qkv.data_ptr() + 8 * D, 3 * D
a.out, a.ldo, a.B, a.H, a.hd, a.Tq, a.Tk, a.causal = out.data_ptr(), D, B, H, hd, Ta, Ta, 1
NP = 4 * H
U = torch.randn(B * NP * D, generator=g).to(dev) * 0.05; Wf = torch.randn(B * NP * D, generator=g).to(dev) * 0.05
c = torch.zeros(B * NP, device=dev); y = torch.randn(B * Ta, D, generator=g).to(dev); lw = torch.ones(D, device=dev)
x = _lib.XApplyArgs(); x.y, x.ln_w, x.U, x.Wf, x.c = y.data_ptr(), lw.data_ptr(), U.data_ptr(), Wf.data_ptr(), c.data_ptr()
x.B, x.H, x.D, x.Te, x.Ta = B, H, D, Te, Ta
buf = torch.zeros(4096 * 8, dtype=torch.int64, device=dev)
for name, fn in (("k_attn self 10x10", lambda: lib.mdt_op_attention(C.byref(a), s)), ("k_xattn_apply", lambda: lib.mdt_op_xattn_apply(C.byref(x), s))):
    for _ in range(3): _lib.check(fn())
    torch.cuda.synchronize(); buf.zero_(); torch.cuda.synchronize()
    assert lib.mdt_debug_set_timing_buffer(buf.data_ptr()) == 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); _lib.check(fn()); e1.record(); torch.cuda.synchronize(); lib.mdt_debug_set_timing_buffer(None)
    t = buf.cpu().numpy().reshape(-1, 8); t = t[t[:, 0] != 0]
    ph = {"load->lds": t[:, 1] - t[:, 0], "barrier": t[:, 2] - t[:, 1], "phase A": t[:, 3] - t[:, 2], "phase B+store": t[:, 4] - t[:, 3], "total": t[:, 4] - t[:, 0]}
    print(f"== {name}: {len(t)} WGs, event {e0.elapsed_time(e1)*1e3:.1f} us")
    for k, v in ph.items(): print(f"   {k:16s} mean {v.mean():8.0f} p10 {np.percentile(v,10):8.0f} p90 {np.percentile(v,90):8.0f}")
    e0.record()
    for _ in range(50): fn()
    e1.record(); torch.cuda.synchronize(); print(f"   avg over 50 back-to-back: {e0.elapsed_time(e1)*1e3/50:.2f} us")
