"""Per-workgroup phase timing of k_attn_xattn (needs the -DMDT_DEBUG_TIMING build: MDT_HIP_LIB=.../libmdt_hip_dbg.so).
Stamps (attn_xattn_tile): 0 entry, 1 attention done (q / k / v rows landed, scores, weighted values -> LDS), 2 barrier passed,
3 projection MFMA loop done, 4 epilogue -> LDS + barrier, 5 ln3 done (U / Wf requested), 6 scores / softmax / combination / stores."""
import ctypes as C, math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from mdt_policy_amd import _lib
lib = _lib.load()
lib.mdt_debug_set_timing_buffer.argtypes = [C.c_void_p]
dev = torch.device("cuda"); B, T, Te, H, hd = int(sys.argv[1]) if len(sys.argv) > 1 else 256, 10, 4, 8, 48
D = H * hd; NP = 4 * H
g = torch.Generator().manual_seed(0); s = torch.cuda.current_stream().cuda_stream
r = lambda *sh: torch.randn(*sh, generator=g).to(dev)
W = r(D, D) / math.sqrt(D); Pd = torch.zeros(D * D, device=dev)
_lib.check(lib.mdt_op_pack_weight(W.data_ptr(), D, D, Pd.data_ptr(), 0, D, s))
L = 4  # decoder blocks: each launch of a step has its own folded operands (cold, as in the sampler)
qkv, y, gate, bias, lw, lb = r(B * T, 3 * D), r(B * T, D), r(6 * D), r(D) * 0.1, torch.ones(D, device=dev), torch.zeros(D, device=dev)
U, Wf, c = r(L, B * NP * D) * 0.05, r(L, B * NP * D) * 0.05, r(L, B * NP) * 0.05
def args(l):
    a = _lib.GemmArgs()
    a.A, a.lda, a.Wp, a.bias, a.out, a.ldo, a.M, a.N, a.K = None, D, Pd.data_ptr(), bias.data_ptr(), y.data_ptr(), D, B * T, D, D
    a.shift_off = a.scale_off = -1; a.gate_off = 2 * D; a.mod, a.mod_stride = gate.data_ptr(), 0
    a.residual, a.rows_per_sample, a.gin, a.gout = 1, T, 1, 1
    x = _lib.XApplyArgs()
    x.y, x.ln_w, x.ln_b, x.U, x.Wf, x.c = y.data_ptr(), lw.data_ptr(), lb.data_ptr(), U[l].data_ptr(), Wf[l].data_ptr(), c[l].data_ptr()
    x.B, x.H, x.D, x.Te, x.Ta = B, H, D, Te, T
    return a, x
A = [args(l) for l in range(L)]
big = torch.empty(64 << 20, device=dev)  # 256 MB: pushes everything out of L2 / MALL between rounds
def launch(l):
    _lib.check(lib.mdt_op_attn_xattn(C.byref(A[l][0]), qkv.data_ptr(), 3 * D, C.byref(A[l][1]), hd, T, s))
buf = torch.zeros(8192 * 8, dtype=torch.int64, device=dev)
for _ in range(3):
    for l in range(L): launch(l)
torch.cuda.synchronize(); buf.zero_(); torch.cuda.synchronize()
assert lib.mdt_debug_set_timing_buffer(buf.data_ptr()) == 0
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for l in range(1, L): launch(l)
e0.record(); launch(0); e1.record(); torch.cuda.synchronize()
lib.mdt_debug_set_timing_buffer(None)
t = buf.cpu().numpy().reshape(-1, 8); t = t[t[:, 0] != 0]
base = t[:, 0].min()
names = ["q/k/v rows + attention -> LDS", "barrier", "projection MFMA loop", "epilogue -> LDS + barrier", "ln3 (+ U / Wf requests)", "scores, softmax, combination, stores"]
print(f"k_attn_xattn: {len(t)} workgroups, event {e0.elapsed_time(e1) * 1e3:.1f} us, first entry -> last stamp {t[:, 6].max() - base} clk")
for i, nm in enumerate(names):
    v = t[:, i + 1] - t[:, i]
    print(f"   {nm:44s} mean {v.mean():9.0f}  p10 {np.percentile(v, 10):9.0f}  p50 {np.percentile(v, 50):9.0f}  p90 {np.percentile(v, 90):9.0f}  max {v.max():9.0f}")
if os.environ.get("ROWS_LANDED"):
    v = t[:, 7] - t[:, 0]
    print(f"   {'   of which: entry -> q/k/v rows in LDS':44s} mean {v.mean():9.0f}  p10 {np.percentile(v, 10):9.0f}  p50 {np.percentile(v, 50):9.0f}  p90 {np.percentile(v, 90):9.0f}  max {v.max():9.0f}")
v = t[:, 6] - t[:, 0]
print(f"   {'total inside the workgroup':44s} mean {v.mean():9.0f}  p10 {np.percentile(v, 10):9.0f}  p50 {np.percentile(v, 50):9.0f}  p90 {np.percentile(v, 90):9.0f}  max {v.max():9.0f}")
st = t[:, 0] - base
print(f"   {'entry offset behind the first workgroup':44s} mean {st.mean():9.0f}  p50 {np.percentile(st, 50):9.0f}  p90 {np.percentile(st, 90):9.0f}  max {st.max():9.0f}")
