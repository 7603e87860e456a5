"""Per-workgroup phase timing of the fused MLP launch k_mlp (needs the -DMDT_DEBUG_TIMING build: MDT_HIP_LIB=.../libmdt_hip_dbg.so).
Stamps (mlp_tile): 0 entry, 1 rows staged, 2 barrier passed, 3 phase-1 MFMA loop done, 4 GELU -> LDS done, 5 phase-2 loop done, 6 stores issued."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from mdt_policy_amd import _lib
lib = _lib.load()
lib.mdt_debug_set_timing_buffer.argtypes = [C.c_void_p]
dev = torch.device("cuda"); M, D, N = 2560, 384, 1536
g = torch.Generator().manual_seed(0); s = torch.cuda.current_stream().cuda_stream
def packed(n, k):
    W = (torch.randn(n, k, generator=g) * 0.05).to(dev); P = torch.zeros(n * k, device=dev)
    _lib.check(lib.mdt_op_pack_weight(W.data_ptr(), n, k, P.data_ptr(), 0, n, s)); return P
x = torch.randn(M, D, generator=g).to(dev); lw = torch.ones(D, device=dev); mod = torch.randn(6 * D, generator=g).to(dev)
P1, P2 = packed(N, D), packed(D, N)
parts = torch.empty(3, M, D, device=dev)
f, p = _lib.GemmArgs(), _lib.GemmArgs()
f.A, f.lda, f.Wp, f.M, f.N, f.K = x.data_ptr(), D, P1.data_ptr(), M, N, D
p.A, p.lda, p.Wp, p.M, p.N, p.K, p.ldo = x.data_ptr(), D, P2.data_ptr(), M, D, N, D
f.ln, f.ln_w, f.act = 1, lw.data_ptr(), 1
f.mod = p.mod = mod.data_ptr()
f.shift_off, f.scale_off, f.gate_off, p.shift_off, p.scale_off, p.gate_off = 3 * D, 4 * D, -1, -1, -1, 5 * D
for a in (f, p):
    a.rows_per_sample, a.gin, a.gout, a.goff = 10, 1, 1, 0
n = C.c_int32(0)
def launch():
    _lib.check(lib.mdt_op_mlp(C.byref(f), C.byref(p), parts.data_ptr(), M * D, C.byref(n), s))
buf = torch.zeros(8192 * 8, dtype=torch.int64, device=dev)
for _ in range(5): launch()
torch.cuda.synchronize(); buf.zero_(); torch.cuda.synchronize()
assert lib.mdt_debug_set_timing_buffer(buf.data_ptr()) == 0
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); launch(); e1.record(); torch.cuda.synchronize()
lib.mdt_debug_set_timing_buffer(None)
tall = buf.cpu().numpy().reshape(-1, 8); G = 240
base = tall[:G, 0].min()
names = ["stage rows (load + LN + LDS)", "barrier", "phase-1 MFMA loop (+ skew wait)", "W2 / gate / residual requests + GELU -> LDS", "(barrier +) phase-2 MFMA loop", "final epilogue"]
print(f"k_mlp: {G} workgroups, event {e0.elapsed_time(e1) * 1e3:.1f} us, MDT_HIP_MLP_SKEW={os.environ.get('MDT_HIP_MLP_SKEW', 'default')}")
for half, who in ((0, "wave 0 (early wave of its SIMD)"), (1, "wave 4 (its SIMD partner)")):
    t = tall[half * G:(half + 1) * G]
    if not (t[:, 0] != 0).all(): continue
    print(f" {who}: first entry -> last stamp {t[:, 6].max() - base} clk")
    for i, nm in enumerate(names):
        v = t[:, i + 1] - t[:, i]
        print(f"   {nm:44s} mean {v.mean():9.0f}  p10 {np.percentile(v, 10):9.0f}  p50 {np.percentile(v, 50):9.0f}  p90 {np.percentile(v, 90):9.0f}  max {v.max():9.0f}")
    v = t[:, 6] - t[:, 0]
    print(f"   {'total inside the workgroup':44s} mean {v.mean():9.0f}  p10 {np.percentile(v, 10):9.0f}  p50 {np.percentile(v, 50):9.0f}  p90 {np.percentile(v, 90):9.0f}  max {v.max():9.0f}")
    v = t[:, 3] - tall[:G, 0]; print(f"   {'entry (wave 0) -> phase-1 loop done':44s} mean {v.mean():9.0f}")
    v = t[:, 4] - tall[:G, 0]; print(f"   {'entry (wave 0) -> hidden columns in LDS':44s} mean {v.mean():9.0f}")
    v = t[:, 5] - tall[:G, 0]; print(f"   {'entry (wave 0) -> phase-2 loop done':44s} mean {v.mean():9.0f}")
