"""Per-workgroup phase timing of k_xattn_gemm_smallm at rollout size (needs the -DMDT_DEBUG_TIMING build:
MDT_HIP_LIB=.../libmdt_hip_dbg.so).  Stamps, thread 0 of every workgroup: xattn_tile 0 entry (all requests issued behind it),
1 ln3 rows in LDS, 2 barrier, 3 score MFMAs + partial tiles in LDS, 4 softmax + combination + new rows in LDS; then, in the
rows behind the grid's, gemm_smallm_tile 0 entry, 1 row statistics, 2 barrier, 3 k-loop, 4 partial tiles met, 5 stored."""
import ctypes as C, math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from mdt_policy_amd import _lib
lib = _lib.load()
lib.mdt_debug_set_timing_buffer.argtypes = [C.c_void_p]
dev = torch.device("cuda"); B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
T, Te, H, hd = 10, 4, 8, 48
D = H * hd; NP = 4 * H; N = 4 * D
g = torch.Generator().manual_seed(0); s = torch.cuda.current_stream().cuda_stream
r = lambda *sh: torch.randn(*sh, generator=g).to(dev)
L = 4  # four blocks' worth of operands, walked in turn: every launch finds its weights where the sampler's chain finds them
Wd = [r(N, D) / math.sqrt(D) for _ in range(L)]; Pd = [torch.zeros(N * D, device=dev) for _ in range(L)]
for w, p in zip(Wd, Pd): _lib.check(lib.mdt_op_pack_weight(w.data_ptr(), N, D, p.data_ptr(), 0, N, s))
y, yo, out = r(B * T, D), torch.empty(B * T, D, device=dev), torch.empty(B * T, N, device=dev)
lw, lb, bias, mod, bo = torch.ones(D, device=dev), torch.zeros(D, device=dev), r(N) * 0.1, r(6 * D) * 0.5, r(D) * 0.1
U, Wf, c = r(L, B * NP * D) * 0.05, r(L, B * NP * D) * 0.05, r(L, B * NP) * 0.05
def args(l):
    x = _lib.XApplyArgs()
    x.y, x.ln_w, x.ln_b, x.U, x.Wf, x.c, x.bo = y.data_ptr(), lw.data_ptr(), lb.data_ptr(), U[l].data_ptr(), Wf[l].data_ptr(), c[l].data_ptr(), bo.data_ptr()
    x.y_out = yo.data_ptr(); x.B, x.H, x.D, x.Te, x.Ta = B, H, D, Te, T
    a = _lib.GemmArgs()
    a.A, a.lda, a.Wp, a.bias, a.out, a.ldo, a.M, a.N, a.K = y.data_ptr(), D, Pd[l].data_ptr(), bias.data_ptr(), out.data_ptr(), N, B * T, N, D
    a.ln, a.ln_w, a.ln_b, a.act = 1, lw.data_ptr(), lb.data_ptr(), 1
    a.gate_off = -1; a.mod, a.mod_stride, a.shift_off, a.scale_off = mod.data_ptr(), 0, 3 * D, 4 * D
    a.rows_per_sample, a.gin, a.gout = T, 1, 1
    return x, a
A = [args(l) for l in range(L)]
def launch(l): _lib.check(lib.mdt_op_xattn_gemm(C.byref(A[l][0]), C.byref(A[l][1]), s))
junk = torch.empty(64 << 20, dtype=torch.float32, device=dev)  # 256 MB: the operands come from the Infinity Cache / HBM as in the chain
buf = torch.zeros(8192 * 8, dtype=torch.int64, device=dev)
for _ in range(3):
    for l in range(L): launch(l)
nwg = (N // 16) * B
rows = []
for rep in range(5):
    junk.add_(1.0)
    for l in range(1, L): launch(l)
    torch.cuda.synchronize(); buf.zero_(); torch.cuda.synchronize()
    assert lib.mdt_debug_set_timing_buffer(buf.data_ptr()) == 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); launch(0); e1.record(); torch.cuda.synchronize()
    lib.mdt_debug_set_timing_buffer(None)
    t = buf.cpu().numpy().reshape(-1, 8)
    n = N // 16
    rows.append((np.concatenate([t[:n, :5], t[n:2 * n, :6]], axis=1), e0.elapsed_time(e1) * 1e3))
t, ev = rows[-1]
base = t[:, 0].min()
names = ["xattn: entry -> ln3 rows in LDS", "barrier", "score MFMAs -> partial tiles", "softmax, combination, rows -> LDS", "barrier -> GEMM tile entry",
         "GEMM: row statistics (weights requested)", "barrier", "k-loop (operand loads + MFMAs)", "partial tiles meet (barrier)", "epilogue + store"]
print(f"k_xattn_gemm_smallm B = {B}: {len(t)} workgroups (blockIdx.y = 0), event {ev:.1f} us (runs: {' '.join(f'{e:.1f}' for _, e in rows)}), "
      f"first entry -> last stamp {t[:, 10].max() - base} clk")
for i, nm in enumerate(names):
    v = t[:, i + 1] - t[:, i]
    print(f"   {nm:44s} mean {v.mean():8.0f}  p10 {np.percentile(v, 10):8.0f}  p50 {np.percentile(v, 50):8.0f}  p90 {np.percentile(v, 90):8.0f}  max {v.max():8.0f}")
st = t[:, 0] - base
print(f"   {'entry offset behind the first workgroup':44s} mean {st.mean():8.0f}  max {st.max():8.0f}")
