"""Launch the decoder's GEMM shapes in isolation (for rocprofv3 --pmc / timing A-B).  usage: gemm_micro.py [reps]"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mdt_policy_amd import _lib
lib = _lib.load()
dev = torch.device("cuda")
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
M, D = int(os.environ.get("M", "2560")), 384
g = torch.Generator().manual_seed(0)
s = torch.cuda.current_stream().cuda_stream
def packed(N, K):
    W = (torch.randn(N, K, generator=g) * 0.05).to(dev)
    P = torch.zeros(N * K, device=dev)
    _lib.check(lib.mdt_op_pack_weight(W.data_ptr(), N, K, P.data_ptr(), 0, N, s)); return P
y = torch.randn(M, D, generator=g).to(dev); hid = torch.randn(M, 4 * D, generator=g).to(dev)
att = torch.randn(M, D, generator=g).to(dev); qkv = torch.empty(M, 3 * D, device=dev); qx = torch.empty(M, D, device=dev)
lw = torch.ones(D, device=dev); lb = torch.zeros(D, device=dev); mod = torch.randn(6 * D, generator=g).to(dev)
bq = torch.zeros(3 * D, device=dev)
def args(A, lda, P, out, ldo, N, K, **kw):
    a = _lib.GemmArgs(); a.A, a.lda, a.Wp, a.out, a.ldo, a.M, a.N, a.K = A.data_ptr(), lda, P.data_ptr(), out.data_ptr(), ldo, M, N, K
    a.shift_off = a.scale_off = a.gate_off = -1; a.rows_per_sample = 10; a.gin = a.gout = 1
    for k, v in kw.items(): setattr(a, k, v)
    return a
shapes = {
 "qkv  LN+mod N=1152 K=384": args(y, D, packed(3 * D, D), qkv, 3 * D, 3 * D, D, ln=1, ln_w=lw.data_ptr(), mod=mod.data_ptr(), shift_off=0, scale_off=D, bias=bq.data_ptr()),
 "proj gated-res N=384 K=384": args(att, D, packed(D, D), y, D, D, D, residual=1, mod=mod.data_ptr(), gate_off=2 * D),
 "xq   LN N=384 K=384": args(y, D, packed(D, D), qx, D, D, D, ln=1, ln_w=lw.data_ptr(), ln_b=lb.data_ptr(), bias=bq.data_ptr()),
 "fc   LN+mod+GELU N=1536 K=384": args(y, D, packed(4 * D, D), hid, 4 * D, 4 * D, D, ln=1, ln_w=lw.data_ptr(), mod=mod.data_ptr(), shift_off=3 * D, scale_off=4 * D, act=1),
 "proj2 gated-res N=384 K=1536": args(hid, 4 * D, packed(D, 4 * D), y, D, D, 4 * D, residual=1, mod=mod.data_ptr(), gate_off=5 * D),
}
geos = [int(x) for x in os.environ.get("GEOS", "0").split(",")]
for a in shapes.values():  # clocks / first-use effects settle before the first timed geometry
    for _ in range(20): _lib.check(lib.mdt_op_gemm(C.byref(a), s))
torch.cuda.synchronize()
for geo in geos:
  lib.mdt_op_set_gemm_geometry(geo)
  print(f"--- geometry {geo}")
  tot_ideal = tot = 0
  for name, a in shapes.items():
      for _ in range(3): _lib.check(lib.mdt_op_gemm(C.byref(a), s))
      e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
      e0.record()
      for _ in range(reps): _lib.check(lib.mdt_op_gemm(C.byref(a), s))
      e1.record(); torch.cuda.synchronize()
      us = e0.elapsed_time(e1) * 1e3 / reps
      fl = 2.0 * M * a.N * a.K
      print(f"{name:34s} {us:8.2f} us  {fl / us / 1e6:7.2f} TFLOP/s  ideal {fl / 157.3e6:6.2f} us", flush=True)
      tot += us * (2 if name.startswith("proj ") else 1); tot_ideal += fl / 157.3e6 * (2 if name.startswith("proj ") else 1)
  print(f"block total (proj x2): {tot:.1f} us   ideal {tot_ideal:.1f} us   eff {tot_ideal / tot:.3f}")

lib.mdt_op_set_gemm_geometry(0)
