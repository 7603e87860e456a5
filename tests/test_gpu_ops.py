"""Kernel-level parity (run on the MI355X box: pytest -m gpu): every hand-written kernel, called through the
C ABI of include/mdt_hip_ops.h, against a PyTorch fp32/fp64 CPU reference of the same op."""
import ctypes as C
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from tests.helpers import assert_close

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lib():
    from mdt_policy_amd import _lib
    return _lib


def dev(t):
    return t.to("cuda", torch.float32).contiguous()


def stream():
    return torch.cuda.current_stream().cuda_stream


def pack(lib, W):
    """Pack a (N, K) weight through the library."""
    N, K = W.shape
    Wd = dev(W)
    P = torch.zeros(N * K, device="cuda")
    lib.check(lib.load().mdt_op_pack_weight(Wd.data_ptr(), N, K, P.data_ptr(), 0, N, stream()))
    return P


def pack_t(lib, W):
    """Fragment image of W^T through the library's transposed packer (what the model keeps of cross_att.query.weight)."""
    R, Cc = W.shape
    Wd = dev(W)
    P = torch.zeros(R * Cc, device="cuda")
    lib.check(lib.load().mdt_op_pack_weight_t(Wd.data_ptr(), R, Cc, Cc, P.data_ptr(), 0, R, stream()))
    return P


def expected_pack(W):
    """Fragment-major image: block (nt, kc) = 64 lanes x 4 floats; lane l = n%16 + 16*((k%16)//4), j = k%4."""
    N, K = W.shape
    out = np.zeros(N * K, np.float32)
    n, k = np.meshgrid(np.arange(N), np.arange(K), indexing="ij")
    idx = (((n // 16) * (K // 16) + k // 16) * 64 + (n % 16) + 16 * ((k % 16) // 4)) * 4 + k % 4
    out[idx.ravel()] = W.numpy().ravel()
    return out


def test_pack_layout(lib):
    g = torch.Generator().manual_seed(0)
    W = torch.randn(48, 64, generator=g)
    np.testing.assert_array_equal(pack(lib, W).cpu().numpy(), expected_pack(W))


def run_gemm(lib, A, W, bias=None, ln_w=None, ln_b=None, mod=None, mod_stride=0, shift_off=-1, scale_off=-1,
             rps=1, act="none", residual_into=None, gate_off=-1, gin=1, gout=1, goff=0, rowvec=None, out_rows=None):
    M, K = A.shape
    N = W.shape[0]
    Ad, Pd = dev(A), pack(lib, W)
    keep = [Ad, Pd]
    out_rows = out_rows or M
    out = dev(residual_into) if residual_into is not None else torch.full((out_rows, N), float("nan"), device="cuda")
    a = lib.GemmArgs()
    a.A, a.lda, a.Wp, a.out, a.ldo = Ad.data_ptr(), K, Pd.data_ptr(), out.data_ptr(), N
    a.M, a.N, a.K = M, N, K
    for name, t in (("bias", bias), ("ln_w", ln_w), ("ln_b", ln_b), ("mod", mod), ("rowvec", rowvec)):
        if t is not None:
            td = dev(t)
            keep.append(td)
            setattr(a, name, td.data_ptr())
    a.ln = int(ln_w is not None)
    a.mod_stride, a.shift_off, a.scale_off, a.rows_per_sample = mod_stride, shift_off, scale_off, rps
    a.act, a.residual, a.gate_off = lib.ACT[act], int(residual_into is not None), gate_off
    a.gin, a.gout, a.goff = gin, gout, goff
    lib.check(lib.load().mdt_op_gemm(C.byref(a), stream()))
    torch.cuda.synchronize()
    return out.cpu()


def ref_act(x, act):
    return {"none": lambda v: v, "gelu": F.gelu, "mish": F.mish, "silu": F.silu}[act](x)


@pytest.mark.parametrize("M,N,K", [(32, 128, 64), (2560, 384, 384), (2560, 1152, 384), (100, 48, 128),
                                   (10, 2304 * 4, 384), (1024, 3072, 384), (37, 112, 512), (2560, 384, 1536),
                                   (256, 768, 512), (1, 16, 16)])
def test_gemm_plain_bias(lib, M, N, K):
    g = torch.Generator().manual_seed(M * 7 + N + K)
    A, W, b = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) / math.sqrt(K), torch.randn(N, generator=g)
    got = run_gemm(lib, A, W, bias=b)
    want = (A.double() @ W.double().T + b.double()).float()
    assert_close(got, want, rtol=1e-4, atol=1e-4, what=f"gemm {M}x{N}x{K}")


@pytest.mark.parametrize("geo", [1, 2, 3, 4, 5, 6])
@pytest.mark.parametrize("M,N,K,ln", [(2560, 1152, 384, True), (2560, 384, 1536, False), (100, 528, 128, True),
                                      (77, 1536, 384, True), (33, 384, 768, False), (64, 1024, 80, False)])
def test_gemm_every_geometry(lib, geo, M, N, K, ln):
    """All TILED workgroup geometries give the same (bit-identical) result: the k order of every dot product is fixed.
    The heuristic (geometry 0) may pick the split-K small-M kernel instead: equal to rounding."""
    g = torch.Generator().manual_seed(geo * 0 + M + N + K)
    A, W, b = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) / math.sqrt(K), torch.randn(N, generator=g)
    lw = torch.randn(K, generator=g) * 0.2 + 1
    y0 = torch.randn(M, N, generator=g)
    x = F.layer_norm(A.double(), (K,), lw.double(), None, 1e-5) if ln else A.double()
    want = (y0.double() + x @ W.double().T + b.double()).float()
    try:
        lib.load().mdt_op_set_gemm_geometry(0)
        heur = run_gemm(lib, A, W, bias=b, ln_w=lw if ln else None, residual_into=y0)
        lib.load().mdt_op_set_gemm_geometry(1)
        ref = run_gemm(lib, A, W, bias=b, ln_w=lw if ln else None, residual_into=y0)
        lib.load().mdt_op_set_gemm_geometry(geo)
        got = run_gemm(lib, A, W, bias=b, ln_w=lw if ln else None, residual_into=y0)
    finally:
        lib.load().mdt_op_set_gemm_geometry(0)
    assert_close(got, want, rtol=1e-4, atol=1e-4, what=f"geometry {geo}")
    assert torch.equal(got, ref), "tiled geometries disagree bitwise"
    assert_close(heur, ref, rtol=1e-5, atol=1e-5, what="heuristic choice vs tiled")


TALL_GEOS = [23]


@pytest.mark.parametrize("geo", TALL_GEOS)
@pytest.mark.parametrize("M,N,K", [(2560, 1152, 384), (10240, 384, 1536), (300, 192, 64), (129, 1536, 384), (1, 16, 32),
                                   (4000, 576, 192), (128, 112, 96)])
def test_gemm_tall_body_bit_equals_the_row_tile_body(lib, geo, M, N, K):
    """The tall body (mdt_tall.h: 128-row tiles, both operands staged in LDS by LDS-DMA, XOR-swizzled activation image) walks
    K in the same order as gemm_tile: every geometry gives the same BITS -- full tiles, ragged rows, partial column tiles,
    a single 32-deep K block -- and the right values against float64."""
    g = torch.Generator().manual_seed(geo * 0 + M + N + K)
    A, W, b = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) / math.sqrt(K), torch.randn(N, generator=g)
    y0 = torch.randn(M, N, generator=g)
    want = (y0.double() + A.double() @ W.double().T + b.double()).float()
    try:
        lib.load().mdt_op_set_gemm_geometry(1)
        ref = run_gemm(lib, A, W, bias=b, residual_into=y0)
        lib.load().mdt_op_set_gemm_geometry(geo)
        got = run_gemm(lib, A, W, bias=b, residual_into=y0)
    finally:
        lib.load().mdt_op_set_gemm_geometry(0)
    assert_close(got, want, rtol=1e-4, atol=1e-4, what=f"tall geometry {geo}")
    assert torch.equal(got, ref), "tall body and row-tile body disagree bitwise"


@pytest.mark.parametrize("M,N", [(32, 256), (33, 256), (1000, 512), (8192 + 45, 1536), (70, 768), (2600, 256), (9000, 1024)])
def test_gemm_weight_stationary_body_bit_equals_the_row_tile_body(lib, M, N):
    """The weight-stationary body (mdt_ws.h: a wave keeps its 2 x 12 weight fragments in
    registers and walks 32-row tiles of A through two LDS buffers) multiplies in the same K order as gemm_tile: same BITS for full and ragged last tiles, one tile per
    workgroup and many, a row count that leaves some workgroups without a tile -- and the right values against float64."""
    K = 192
    g = torch.Generator().manual_seed(M + N)
    A, W, b = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) / math.sqrt(K), torch.randn(N, generator=g)
    want = (A.double() @ W.double().T + b.double()).float()
    try:
        lib.load().mdt_op_set_ws_split(0)   # the fp32 MFMA form (the bf16 split form: its own test below)
        lib.load().mdt_op_set_gemm_geometry(1)
        ref = run_gemm(lib, A, W, bias=b)
        lib.load().mdt_op_set_gemm_geometry(30)
        got = run_gemm(lib, A, W, bias=b)
    finally:
        lib.load().mdt_op_set_gemm_geometry(0)
        lib.load().mdt_op_set_ws_split(-1)
    assert_close(got, want, rtol=1e-4, atol=1e-4, what="weight-stationary body")
    assert torch.equal(got, ref), "weight-stationary body and row-tile body disagree bitwise"


@pytest.mark.parametrize("M,N", [(32, 384), (33, 128), (1000, 384), (10240 + 13, 384), (2560, 1152), (8192 + 45, 1536), (70, 192),
                                 (4096, 384)])
def test_gemm_weight_stationary_body_at_k384_with_the_training_hooks(lib, M, N):
    """Round 6: the weight-stationary body with ONE column tile per wave over K = 384 (24 fragment quads in registers) -- the d x d,
    qkv and c_fc / c_proj-gradient products of the B = 1024 training step.  Plain rows, bias + GELU, and the two training hooks
    (aux_mode 1: pre-activation kept beside the activated value; 2: product times act'(aux)), each BITWISE against the row-tile
    body and against float64; 8- and 12-wave shapes, ragged last tiles, workgroups without a tile."""
    L = lib.load()
    K = 384
    g = torch.Generator().manual_seed(M + N)
    A, W, b = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) / math.sqrt(K), torch.randn(N, generator=g)
    y0 = torch.randn(M, N, generator=g)
    Ad, Pd, bd = dev(A), pack(lib, W), dev(b)
    outs = {}
    try:
        L.mdt_op_set_ws_split(0)   # the fp32 MFMA form of the body (its bf16 split form has its own test below)
        for gsel in (1, 30):
            L.mdt_op_set_gemm_geometry(gsel)
            o = {"plain": run_gemm(lib, A, W, bias=b), "gelu": run_gemm(lib, A, W, bias=b, act="gelu")}
            for mode in (1, 2):
                out = torch.full((M, N), float("nan"), device="cuda")
                aux = torch.full((M, N), float("nan"), device="cuda") if mode == 1 else dev(y0)
                a = lib.GemmArgs()
                a.A, a.lda, a.Wp, a.out, a.ldo, a.M, a.N, a.K = Ad.data_ptr(), K, Pd.data_ptr(), out.data_ptr(), N, M, N, K
                a.bias = bd.data_ptr() if mode == 1 else None
                a.shift_off = a.scale_off = a.gate_off = -1
                a.rows_per_sample = a.gin = a.gout = 1
                a.act, a.aux, a.aux_mode = lib.ACT["gelu"], aux.data_ptr(), mode
                lib.check(L.mdt_op_gemm(C.byref(a), stream()))
                torch.cuda.synchronize()
                o[f"aux{mode}"] = out.cpu()
                if mode == 1:
                    o["aux1_pre"] = aux.cpu()
            outs[gsel] = o
    finally:
        L.mdt_op_set_gemm_geometry(0)
        L.mdt_op_set_ws_split(-1)
    for k in outs[1]:
        assert torch.equal(outs[1][k], outs[30][k]), f"{k}: weight-stationary body differs from the row-tile body"
    u = A.double() @ W.double().T
    assert_close(outs[30]["plain"], (u + b.double()).float(), rtol=1e-4, atol=1e-4, what="plain rows")
    assert_close(outs[30]["gelu"], F.gelu(u + b.double()).float(), rtol=1e-4, atol=1e-4, what="bias + GELU")
    assert_close(outs[30]["aux1_pre"], (u + b.double()).float(), rtol=1e-4, atol=1e-4, what="aux_mode 1: pre-activation")
    assert_close(outs[30]["aux1"], F.gelu(u + b.double()).float(), rtol=1e-4, atol=1e-4, what="aux_mode 1: activated")
    y64 = y0.double().requires_grad_()
    F.gelu(y64).sum().backward()
    assert_close(outs[30]["aux2"], (u * y64.grad).float(), rtol=1e-4, atol=2e-4, what="aux_mode 2: value * act'(aux)")


@pytest.mark.parametrize("M,N,K", [(8192, 384, 384), (8192 + 77, 1152, 384), (12288, 1536, 384), (33, 384, 384),
                                   (8192 + 45, 1536, 192), (70, 768, 192), (20000, 256, 192), (9000, 576, 192), (8192 + 45, 192, 192),
                                   (70, 192, 192)])
def test_gemm_weight_stationary_body_split_three_ways_into_bf16_keeps_fp32_accuracy(lib, M, N, K):
    """Round 6: the K = 384 / K = 192 weight-stationary body with every operand as three bf16 parts and six bf16 MFMA products per k32 step
    (mdt_ws.h gemm_ws_split_tile).  Not the fp32 bodies' bits -- so: against float64, the split form's error must stay within
    1.5x the fp32 MFMA body's own on the same inputs (and inside the fp32 bodies' test tolerance), for plain rows, bias + GELU
    and the two training hooks; inputs with a wide dynamic range included (the parts' exponents follow the value's)."""
    L = lib.load()
    g = torch.Generator().manual_seed(M * 3 + N)
    A = torch.randn(M, K, generator=g) * torch.exp(2.0 * torch.randn(M, 1, generator=g))   # rows of very different scale
    W, b = torch.randn(N, K, generator=g) / math.sqrt(K), torch.randn(N, generator=g)
    y0 = torch.randn(M, N, generator=g)
    Ad, Pd, bd = dev(A), pack(lib, W), dev(b)
    outs = {}
    try:
        L.mdt_op_set_gemm_geometry(30)
        for split in (0, 1):
            L.mdt_op_set_ws_split(split)
            o = {"plain": run_gemm(lib, A, W, bias=b)}
            if K == 384:   # (the K = 192 form carries the SwishGLU epilogues instead: tests/test_mae.py)
                o["gelu"] = run_gemm(lib, A, W, bias=b, act="gelu")
            for mode in ((1, 2) if K == 384 else ()):
                out = torch.full((M, N), float("nan"), device="cuda")
                aux = torch.full((M, N), float("nan"), device="cuda") if mode == 1 else dev(y0)
                a = lib.GemmArgs()
                a.A, a.lda, a.Wp, a.out, a.ldo, a.M, a.N, a.K = Ad.data_ptr(), K, Pd.data_ptr(), out.data_ptr(), N, M, N, K
                a.bias = bd.data_ptr() if mode == 1 else None
                a.shift_off = a.scale_off = a.gate_off = -1
                a.rows_per_sample = a.gin = a.gout = 1
                a.act, a.aux, a.aux_mode = lib.ACT["gelu"], aux.data_ptr(), mode
                lib.check(L.mdt_op_gemm(C.byref(a), stream()))
                torch.cuda.synchronize()
                o[f"aux{mode}"] = out.cpu()
                if mode == 1:
                    o["aux1_pre"] = aux.cpu()
            outs[split] = o
    finally:
        L.mdt_op_set_gemm_geometry(0)
        L.mdt_op_set_ws_split(-1)
    u = A.double() @ W.double().T
    y64 = y0.double().requires_grad_()
    F.gelu(y64).sum().backward()
    want = {"plain": u + b.double(), "gelu": F.gelu(u + b.double()), "aux1_pre": u + b.double(), "aux1": F.gelu(u + b.double()),
            "aux2": u * y64.grad}
    want = {k: v for k, v in want.items() if k in outs[0]}
    scale = A.double().abs().amax(dim=1, keepdim=True)          # per-row scale of the inputs
    for k, w in want.items():
        e32 = ((outs[0][k].double() - w).abs() / scale).max().item()
        e16 = ((outs[1][k].double() - w).abs() / scale).max().item()
        assert e16 <= 1.5 * e32 + 1e-7, f"{k}: split error {e16:.3g} against the fp32 body's {e32:.3g} (per-row scaled)"
        assert_close(outs[1][k] / scale.float(), (w / scale).float(), rtol=1e-4, atol=1e-4, what=f"split: {k}")
    assert not torch.equal(outs[0]["plain"], outs[1]["plain"]) or M < 64, "the split form did not run (same bits as the fp32 body)"


@pytest.mark.parametrize("geo", [23])
def test_gemm_tall_body_epilogues(lib, geo):
    """Every epilogue the tall body carries, against the row-tile body (bitwise) and float64: bias + GELU, per-sample gate +
    residual, token-row remap + row vector, and the two training hooks (aux_mode 1: pre-activation kept; 2: act'(aux))."""
    L = lib.load()
    g = torch.Generator().manual_seed(geo)
    B, T, D, K = 31, 10, 384, 1536
    M = B * T
    A, W = torch.randn(M, K, generator=g), torch.randn(D, K, generator=g) / math.sqrt(K)
    y0, mod, b = torch.randn(M, D, generator=g), torch.randn(B, 6 * D, generator=g), torch.randn(D, generator=g)
    tok, Wt, pos = torch.randn(B * 3, 128, generator=g), torch.randn(D, 128, generator=g) / 10, torch.randn(D, generator=g)
    outs = {}
    try:
        for gsel in (1, geo):
            L.mdt_op_set_gemm_geometry(gsel)
            o = {}
            o["gelu"] = run_gemm(lib, A, W, bias=b, act="gelu")
            o["gated"] = run_gemm(lib, A, W, residual_into=y0, mod=mod, mod_stride=6 * D, rps=T, gate_off=5 * D)
            o["remap"] = run_gemm(lib, tok, Wt, gin=3, gout=4, goff=1, rowvec=pos, out_rows=B * 4)
            # training hooks through the raw argument block
            Ad, Pd, bd = dev(A), pack(lib, W), dev(b)
            for mode in (1, 2):
                out = torch.full((M, D), float("nan"), device="cuda")
                aux = torch.full((M, D), float("nan"), device="cuda") if mode == 1 else dev(y0)
                a = lib.GemmArgs()
                a.A, a.lda, a.Wp, a.out, a.ldo, a.M, a.N, a.K = Ad.data_ptr(), K, Pd.data_ptr(), out.data_ptr(), D, M, D, K
                a.bias = bd.data_ptr() if mode == 1 else None
                a.shift_off = a.scale_off = a.gate_off = -1
                a.rows_per_sample = a.gin = a.gout = 1
                a.act, a.aux, a.aux_mode = lib.ACT["gelu"], aux.data_ptr(), mode
                # (aux needs K <= 512 in the row-tile body: use a K slice for this pair)
                a.K, a.lda = 384, K
                Pk = pack(lib, W[:, :384].contiguous())
                a.Wp = Pk.data_ptr()
                lib.check(L.mdt_op_gemm(C.byref(a), stream()))
                torch.cuda.synchronize()
                o[f"aux{mode}"] = out.cpu()
                if mode == 1:
                    o["aux1_pre"] = aux.cpu()
            outs[gsel] = o
    finally:
        L.mdt_op_set_gemm_geometry(0)
    for k in outs[1]:
        a_, b_ = outs[1][k], outs[geo][k]
        assert torch.equal(torch.nan_to_num(a_, nan=7.0), torch.nan_to_num(b_, nan=7.0)), f"{k}: tall body differs from the row-tile body"
    v = A.double() @ W.double().T
    assert_close(outs[geo]["gelu"], F.gelu(v + b.double()).float(), rtol=1e-4, atol=1e-4, what="bias + GELU")
    assert_close(outs[geo]["gated"], (y0.double() + v * mod[:, 5 * D:].double().repeat_interleave(T, 0)).float(), rtol=1e-4,
                 atol=1e-4, what="gated residual")
    u = A[:, :384].double() @ W[:, :384].double().T
    assert_close(outs[geo]["aux1_pre"], (u + b.double()).float(), rtol=1e-4, atol=1e-4, what="aux_mode 1: pre-activation")
    assert_close(outs[geo]["aux1"], F.gelu(u + b.double()).float(), rtol=1e-4, atol=1e-4, what="aux_mode 1: activated")
    y64 = y0.double().requires_grad_()
    F.gelu(y64).sum().backward()
    assert_close(outs[geo]["aux2"], (u * y64.grad).float(), rtol=1e-4, atol=2e-4, what="aux_mode 2: value * act'(aux)")


@pytest.mark.parametrize("M", [1, 10, 16, 17, 40, 160, 192, 193])
@pytest.mark.parametrize("case", ["plain", "ln_mod_rows", "ln_bias_bcast", "gated_residual_k1536", "remap_gelu"])
def test_gemm_small_m_kernel_matches_reference_and_tiled(lib, M, case):
    """Rollout-sized batches (M <= 192 rows) run on k_gemm_smallm (K split over a workgroup's 8 waves); every fused
    feature against float64 and against the tiled kernel forced through the geometry hook (M = 193: tiled anyway)."""
    g = torch.Generator().manual_seed(M * 31 + len(case))
    D, T = 384, 1
    kw, N, K = {}, 384, 384
    if case == "plain":
        N, K = 1152, 768
        A, W, b = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) / math.sqrt(K), torch.randn(N, generator=g)
        want = A.double() @ W.double().T + b.double()
        kw = dict(bias=b)
    elif case in ("ln_mod_rows", "ln_bias_bcast"):
        N = 1536
        A, W, b = torch.randn(M, D, generator=g) * 2 + 0.3, torch.randn(N, D, generator=g) / math.sqrt(D), torch.randn(N, generator=g)
        lw, lb = torch.randn(D, generator=g) * 0.2 + 1, torch.randn(D, generator=g) * 0.2
        rows = case == "ln_mod_rows"
        T = 10 if (rows and M % 10 == 0) else 1
        nb = M // T
        mod = torch.randn(nb if rows else 1, 6 * D, generator=g)
        x = F.layer_norm(A.double(), (D,), lw.double(), None if rows else lb.double(), 1e-5)
        mr = mod.double().repeat_interleave(T, 0) if rows else mod.double()
        x = mr[:, 3 * D:4 * D] + x * mr[:, 4 * D:5 * D]
        want = x @ W.double().T + b.double()
        kw = dict(bias=b, ln_w=lw, ln_b=None if rows else lb, mod=mod if rows else mod[0].clone(),
                  mod_stride=6 * D if rows else 0, shift_off=3 * D, scale_off=4 * D, rps=T)
    elif case == "gated_residual_k1536":
        K = 1536
        T = 10 if M % 10 == 0 else 1
        A, W = torch.randn(M, K, generator=g), torch.randn(D, K, generator=g) / math.sqrt(K)
        y0, mod = torch.randn(M, D, generator=g), torch.randn(M // T, 6 * D, generator=g)
        want = y0.double() + (A.double() @ W.double().T) * mod[:, 5 * D:].double().repeat_interleave(T, 0)
        kw = dict(residual_into=y0, mod=mod, mod_stride=6 * D, rps=T, gate_off=5 * D)
    else:  # rows scattered into a wider buffer, positional row vector, GELU
        N, K = 128, 512
        A, W = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) / math.sqrt(K)
        pos = torch.randn(N, generator=g)
        want = F.gelu(A.double() @ W.double().T + pos.double())
        kw = dict(gin=1, gout=3, goff=2, rowvec=pos, out_rows=M * 3, act="gelu")
    outs = {}
    try:
        for geo in (0, 1):
            lib.load().mdt_op_set_gemm_geometry(geo)
            o = run_gemm(lib, A, W, **kw)
            outs[geo] = o.reshape(M, 3, N)[:, 2] if case == "remap_gelu" else o
            if case == "remap_gelu":
                assert torch.isnan(o.reshape(M, 3, N)[:, :2]).all()
    finally:
        lib.load().mdt_op_set_gemm_geometry(0)
    assert_close(outs[0], want.float(), rtol=2e-4, atol=2e-4, what=f"{case} M={M}")
    assert_close(outs[0], outs[1], rtol=1e-5, atol=2e-5, what=f"{case} M={M}: small-M vs tiled")


def test_gemm_is_transpose_sensitive(lib):
    """A = I with an ASYMMETRIC weight: catches a swapped C/D fragment map."""
    K = N = 64
    W = torch.arange(N * K, dtype=torch.float32).reshape(N, K) / 100.0
    got = run_gemm(lib, torch.eye(K), W)
    assert_close(got, W.T.contiguous(), rtol=1e-6, atol=1e-6, what="identity gemm")


@pytest.mark.parametrize("act", ["gelu", "mish", "silu"])
def test_gemm_activations(lib, act):
    g = torch.Generator().manual_seed(5)
    A, W = torch.randn(70, 128, generator=g) * 2, torch.randn(256, 128, generator=g) / 4
    b = torch.randn(256, generator=g)
    got = run_gemm(lib, A, W, bias=b, act=act)
    want = ref_act(A.double() @ W.double().T + b.double(), act).float()
    assert_close(got, want, rtol=1e-4, atol=1e-4, what=act)


@pytest.mark.parametrize("with_bias,with_mod,D", [(False, False, 384), (True, False, 384), (False, True, 384),
                                                  (True, True, 128), (False, True, 512)])
def test_gemm_layernorm_modulate_prologue(lib, with_bias, with_mod, D):
    g = torch.Generator().manual_seed(11 + D)
    B, T, N = 7, 10, 3 * D
    M = B * T
    A = torch.randn(M, D, generator=g) * 3 + 0.5
    W, b = torch.randn(N, D, generator=g) / math.sqrt(D), torch.randn(N, generator=g)
    lw, lb = torch.randn(D, generator=g) * 0.2 + 1, torch.randn(D, generator=g) * 0.2
    mod = torch.randn(B, 6 * D, generator=g)
    x = F.layer_norm(A.double(), (D,), lw.double(), lb.double() if with_bias else None, 1e-5)
    if with_mod:
        sh = mod[:, 3 * D:4 * D].double().repeat_interleave(T, 0)
        sc = mod[:, 4 * D:5 * D].double().repeat_interleave(T, 0)
        x = sh + x * sc
    want = (x @ W.double().T + b.double()).float()
    got = run_gemm(lib, A, W, bias=b, ln_w=lw, ln_b=lb if with_bias else None, mod=mod if with_mod else None,
                   mod_stride=6 * D, shift_off=3 * D if with_mod else -1, scale_off=4 * D if with_mod else -1, rps=T)
    assert_close(got, want, rtol=2e-4, atol=2e-4, what="ln+mod gemm")
    if with_mod:  # broadcast modulation row (sampler: one sigma for the whole batch)
        x = F.layer_norm(A.double(), (D,), lw.double(), lb.double() if with_bias else None, 1e-5)
        x = mod[2, 3 * D:4 * D].double() + x * mod[2, 4 * D:5 * D].double()
        want = (x @ W.double().T + b.double()).float()
        got = run_gemm(lib, A, W, bias=b, ln_w=lw, ln_b=lb if with_bias else None, mod=mod[2].clone(), mod_stride=0,
                       shift_off=3 * D, scale_off=4 * D, rps=T)
        assert_close(got, want, rtol=2e-4, atol=2e-4, what="ln+broadcast mod gemm")


@pytest.mark.parametrize("gated", [False, True])
@pytest.mark.parametrize("K", [384, 1536])
def test_gemm_gated_residual_epilogue(lib, gated, K):
    g = torch.Generator().manual_seed(3 + K)
    B, T, D = 9, 10, 384
    M = B * T
    A, W = torch.randn(M, K, generator=g), torch.randn(D, K, generator=g) / math.sqrt(K)
    y0 = torch.randn(M, D, generator=g)
    mod = torch.randn(B, 6 * D, generator=g)
    v = A.double() @ W.double().T
    if gated:
        v = v * mod[:, 5 * D:].double().repeat_interleave(T, 0)
    want = (y0.double() + v).float()
    got = run_gemm(lib, A, W, residual_into=y0, mod=mod if gated else None, mod_stride=6 * D, rps=T,
                   gate_off=5 * D if gated else -1)
    assert_close(got, want, rtol=1e-4, atol=1e-4, what="residual gemm")


def test_gemm_row_remap_and_rowvec(lib):
    """Token concatenation: goal row -> slot 0, 3 state rows -> slots 1..3 of each sample's 4-token context."""
    g = torch.Generator().manual_seed(8)
    B, D = 6, 128
    goal, tok = torch.randn(B, 512, generator=g), torch.randn(B * 3, 128, generator=g)
    Wg, Wt = torch.randn(D, 512, generator=g) / 20, torch.randn(D, 128, generator=g) / 10
    pos = torch.randn(D, generator=g)
    h = torch.zeros(B * 4, D)
    h = run_gemm(lib, goal, Wg, residual_into=None, gin=1, gout=4, goff=0, rowvec=pos, out_rows=B * 4)
    got_goal = h.reshape(B, 4, D)[:, 0]
    assert_close(got_goal, (goal.double() @ Wg.double().T + pos.double()).float(), rtol=1e-4, atol=1e-4, what="goal rows")
    h2 = run_gemm(lib, tok, Wt, gin=3, gout=4, goff=1, out_rows=B * 4)
    got_tok = h2.reshape(B, 4, D)[:, 1:].reshape(B * 3, D)
    assert_close(got_tok, (tok.double() @ Wt.double().T).float(), rtol=1e-4, atol=1e-4, what="state rows")
    assert torch.isnan(h2.reshape(B, 4, D)[:, 0]).all()  # untouched rows stay untouched


def run_attn(lib, q, k, v, H, causal, rope=False):
    B, Tq, D = q.shape
    Tk = k.shape[1]
    qd, kd, vd = dev(q.reshape(B * Tq, D)), dev(k.reshape(B * Tk, D)), dev(v.reshape(B * Tk, D))
    out = torch.full((B * Tq, D), float("nan"), device="cuda")
    a = lib.AttnArgs()
    a.q, a.ldq, a.k, a.v, a.ldkv, a.out, a.ldo = qd.data_ptr(), D, kd.data_ptr(), vd.data_ptr(), D, out.data_ptr(), D
    a.B, a.H, a.hd, a.Tq, a.Tk, a.causal, a.rope = B, H, D // H, Tq, Tk, int(causal), int(rope)
    lib.check(lib.load().mdt_op_attention(C.byref(a), stream()))
    torch.cuda.synchronize()
    return out.cpu().reshape(B, Tq, D)


def ref_attn(q, k, v, H, causal, rope=False):
    from oracle.mdt_oracle import apply_rotary
    B, Tq, D = q.shape
    hd = D // H
    qh, kh, vh = (t.double().view(B, -1, H, hd).transpose(1, 2) for t in (q, k, v))
    if rope:
        qh, kh = apply_rotary(qh, 32), apply_rotary(kh, 32)
    att = qh @ kh.transpose(-2, -1) / math.sqrt(hd)
    if causal:
        att = att.masked_fill(~torch.ones(Tq, k.shape[1], dtype=torch.bool).tril(), float("-inf"))
    return (att.softmax(-1) @ vh).transpose(1, 2).reshape(B, Tq, D).float()


@pytest.mark.parametrize("H,hd,Tq,Tk,causal", [(8, 48, 10, 10, True), (8, 48, 10, 4, True), (8, 48, 4, 4, False),
                                               (8, 16, 10, 3, True), (8, 64, 10, 10, True), (4, 32, 16, 16, False),
                                               (8, 48, 1, 1, True)])
@pytest.mark.parametrize("B", [13, 70])  # >= 64 samples: two half-size workgroups per sample (head split)
def test_attention(lib, H, hd, Tq, Tk, causal, B):
    g = torch.Generator().manual_seed(H + hd + Tq + Tk)
    D = H * hd
    q, k, v = (torch.randn(B, T, D, generator=g) for T in (Tq, Tk, Tk))
    assert_close(run_attn(lib, q, k, v, H, causal), ref_attn(q, k, v, H, causal), rtol=1e-4, atol=1e-5, what="attention")


@pytest.mark.parametrize("hd,T,causal,gated,B", [(48, 10, True, True, 1), (48, 4, False, False, 1), (16, 10, True, True, 1),
                                                 (32, 16, True, False, 1), (64, 10, False, True, 1), (64, 16, True, True, 2),
                                                 (48, 1, True, False, 1), (48, 10, True, True, 5), (16, 4, False, True, 16)])
def test_attention_fused_into_projection_per_sample(lib, hd, T, causal, gated, B):
    """mdt_op_attn_proj (rollout batches): out += gate_b * (attention(q, k, v) @ W^T + b) per sample, against float64."""
    g = torch.Generator().manual_seed(hd + T + B)
    H, D, N = 8, 8 * hd, 8 * hd
    qkv = torch.randn(B * T, 3 * D, generator=g)
    W, b = torch.randn(N, D, generator=g) / math.sqrt(D), torch.randn(N, generator=g) * 0.1
    gate, y0 = torch.randn(B, 6 * N, generator=g), torch.randn(B * T, N, generator=g)
    q3 = qkv.view(B, T, 3 * D)
    att = ref_attn(q3[..., :D], q3[..., D:2 * D], q3[..., 2 * D:], H, causal).double().reshape(B * T, D)
    upd = att @ W.double().T + b.double()
    if gated:
        upd = gate[:, 2 * N:3 * N].double().repeat_interleave(T, 0) * upd
    want = y0.double() + upd
    qd, Pd, bd, gd, out = dev(qkv), pack(lib, W), dev(b), dev(gate), dev(y0).clone()
    a = lib.GemmArgs()
    a.A, a.lda, a.Wp, a.bias, a.out, a.ldo, a.M, a.N, a.K = None, D, Pd.data_ptr(), bd.data_ptr(), out.data_ptr(), N, B * T, N, D
    a.shift_off = a.scale_off = -1
    a.gate_off = 2 * N if gated else -1
    a.mod, a.mod_stride = (gd.data_ptr(), 6 * N) if gated else (None, 0)
    a.residual, a.rows_per_sample, a.gin, a.gout = 1, T, 1, 1
    lib.check(lib.load().mdt_op_attn_proj(C.byref(a), qd.data_ptr(), 3 * D, hd, T, int(causal), stream()))
    assert_close(out.cpu(), want.float(), rtol=1e-4, atol=2e-5, what="attention + projection")
    if not causal or hd == 64:
        a.M = 65 * T  # more samples than the per-sample kernel takes, and no tiled form for this case
        assert lib.load().mdt_op_attn_proj(C.byref(a), qd.data_ptr(), 3 * D, hd, T, int(causal), stream()) == 2


@pytest.mark.parametrize("B", [1, 2, 3, 4, 5, 6, 7, 8])
def test_fused_attention_projection_against_the_two_launches_at_rollout_batches(lib, B):
    """since round 5 the model-level entry points run the self-attention of batches 1 .. 32 inside its output projection
    (k_attn_proj_smallm, both attention products on the MFMA pipe), so those batches no longer have the summation order of the
    attention launch + projection GEMM pair.  The fused launch against that pair on the same operands at B = 1 .. 8 (the rollout
    batches), with the tolerance the header states for it: 2e-5 absolute + 1e-4 relative on values of order one."""
    hd, T, H = 48, 10, 8
    D = N = H * hd
    g = torch.Generator().manual_seed(900 + B)
    qkv = torch.randn(B * T, 3 * D, generator=g)
    W = torch.randn(N, D, generator=g) / math.sqrt(D)
    gate, y0 = torch.randn(B, 6 * N, generator=g), torch.randn(B * T, N, generator=g)
    qd, Pd, gd = dev(qkv), pack(lib, W), dev(gate)
    L = lib.load()

    def proj_args(A, out):
        a = lib.GemmArgs()
        a.A, a.lda, a.Wp, a.out, a.ldo, a.M, a.N, a.K = (A.data_ptr() if A is not None else None), D, Pd.data_ptr(), out.data_ptr(), N, B * T, N, D
        a.shift_off = a.scale_off = -1
        a.gate_off, a.mod, a.mod_stride = 2 * N, gd.data_ptr(), 6 * N
        a.residual, a.rows_per_sample, a.gin, a.gout = 1, T, 1, 1
        return a

    fused = dev(y0).clone()
    lib.check(L.mdt_op_attn_proj(C.byref(proj_args(None, fused)), qd.data_ptr(), 3 * D, hd, T, 1, stream()))
    att = torch.empty(B * T, D, device="cuda")
    aa = lib.AttnArgs(q=qd.data_ptr(), ldq=3 * D, k=qd.data_ptr() + 4 * D, v=qd.data_ptr() + 8 * D, ldkv=3 * D, out=att.data_ptr(), ldo=D,
                      B=B, H=H, hd=hd, Tq=T, Tk=T, causal=1, rope=0)
    lib.check(L.mdt_op_attention(C.byref(aa), stream()))
    pair = dev(y0).clone()
    lib.check(L.mdt_op_gemm(C.byref(proj_args(att, pair)), stream()))
    torch.cuda.synchronize()
    assert_close(fused.cpu(), pair.cpu().double(), rtol=1e-4, atol=2e-5, what=f"fused attention + projection vs the two launches, B = {B}")


@pytest.mark.parametrize("hd,T,gated,B", [(48, 10, True, 256), (48, 10, False, 77), (32, 16, True, 70), (16, 7, True, 201),
                                          (48, 1, True, 100), (32, 13, True, 65)])
def test_attention_in_the_projection_prologue_for_large_batches(lib, hd, T, gated, B):
    """mdt_op_attn_proj with more than 64 samples (k_attn_proj_wide): the causal attention of every 32-row tile -- rows of
    3-4 samples, keys reaching back into the previous tile's rows -- is computed in the prologue of the tiled projection;
    against float64, and bit-for-bit against the attention launch + projection GEMM it replaces."""
    g = torch.Generator().manual_seed(hd + T + B)
    H, D, N = 8, 8 * hd, 8 * hd
    qkv = torch.randn(B * T, 3 * D, generator=g)
    W, b = torch.randn(N, D, generator=g) / math.sqrt(D), torch.randn(N, generator=g) * 0.1
    gate, y0 = torch.randn(B, 6 * N, generator=g), torch.randn(B * T, N, generator=g)
    q3 = qkv.view(B, T, 3 * D)
    att = ref_attn(q3[..., :D], q3[..., D:2 * D], q3[..., 2 * D:], H, True).double().reshape(B * T, D)
    upd = att @ W.double().T + b.double()
    if gated:
        upd = gate[:, 2 * N:3 * N].double().repeat_interleave(T, 0) * upd
    want = y0.double() + upd
    qd, Pd, bd, gd, out = dev(qkv), pack(lib, W), dev(b), dev(gate), dev(y0).clone()
    a = lib.GemmArgs()
    a.A, a.lda, a.Wp, a.bias, a.out, a.ldo, a.M, a.N, a.K = None, D, Pd.data_ptr(), bd.data_ptr(), out.data_ptr(), N, B * T, N, D
    a.shift_off = a.scale_off = -1
    a.gate_off = 2 * N if gated else -1
    a.mod, a.mod_stride = (gd.data_ptr(), 6 * N) if gated else (None, 0)
    a.residual, a.rows_per_sample, a.gin, a.gout = 1, T, 1, 1
    lib.check(lib.load().mdt_op_attn_proj(C.byref(a), qd.data_ptr(), 3 * D, hd, T, 1, stream()))
    torch.cuda.synchronize()
    assert_close(out.cpu(), want.float(), rtol=1e-4, atol=2e-5, what="attention in the projection's prologue")
    # the two launches it replaces
    att_gpu = run_attn(lib, q3[..., :D], q3[..., D:2 * D], q3[..., 2 * D:], H, True).reshape(B * T, D)
    assert_close(att_gpu, att.float(), rtol=1e-4, atol=1e-5, what="attention launch")


def test_attention_matches_sdpa_is_causal_for_rectangular_scores(lib):
    """The reference calls F.scaled_dot_product_attention(is_causal=True) on 10x4 scores."""
    g = torch.Generator().manual_seed(77)
    B, H, hd = 5, 8, 48
    q, k, v = torch.randn(B, 10, H * hd, generator=g), torch.randn(B, 4, H * hd, generator=g), torch.randn(B, 4, H * hd, generator=g)
    qh, kh, vh = (t.view(B, -1, H, hd).transpose(1, 2) for t in (q, k, v))
    want = F.scaled_dot_product_attention(qh, kh, vh, is_causal=True).transpose(1, 2).reshape(B, 10, H * hd)
    assert_close(run_attn(lib, q, k, v, H, True), want, rtol=1e-4, atol=1e-5, what="sdpa causal 10x4")


@pytest.mark.parametrize("hd", [48, 64, 32])
def test_attention_rope(lib, hd):
    g = torch.Generator().manual_seed(hd)
    B, H = 4, 8
    q, k, v = (torch.randn(B, T, H * hd, generator=g) for T in (10, 4, 4))
    assert_close(run_attn(lib, q, k, v, H, True, rope=True), ref_attn(q, k, v, H, True, rope=True), rtol=1e-4, atol=1e-5,
                 what="rope attention")


@pytest.mark.parametrize("D,with_bias", [(384, False), (128, True), (512, True)])
def test_layernorm(lib, D, with_bias):
    g = torch.Generator().manual_seed(D)
    M = 1027
    x, w, b = torch.randn(M, D, generator=g) * 4 - 1, torch.randn(D, generator=g), torch.randn(D, generator=g)
    xd, wd, bd = dev(x), dev(w), dev(b)
    out = torch.empty_like(xd)
    lib.check(lib.load().mdt_op_layernorm(xd.data_ptr(), wd.data_ptr(), bd.data_ptr() if with_bias else None,
                                          out.data_ptr(), M, D, stream()))
    want = F.layer_norm(x.double(), (D,), w.double(), b.double() if with_bias else None, 1e-5).float()
    assert_close(out.cpu(), want, rtol=1e-4, atol=1e-5, what="layernorm")


def test_action_embed(lib):
    g = torch.Generator().manual_seed(2)
    B, T, A, D, sd = 11, 10, 7, 384, 0.5
    x, Wa, ba = torch.randn(B * T, A, generator=g) * 30, torch.randn(D, A, generator=g), torch.randn(D, generator=g)
    sigma = torch.rand(B, generator=g) * 50 + 0.01
    xd, Wd, bd, sg = dev(x), dev(Wa.T), dev(ba), dev(sigma)  # the library keeps action_emb.weight as (A, D)
    y = torch.empty(B * T, D, device="cuda")
    lib.check(lib.load().mdt_op_action_embed(xd.data_ptr(), sg.data_ptr(), 1, sd, Wd.data_ptr(), bd.data_ptr(),
                                             y.data_ptr(), B * T, A, D, T, stream()))
    cin = (1 / (sigma.double() ** 2 + sd ** 2).sqrt()).repeat_interleave(T)[:, None]
    want = ((x.double() * cin) @ Wa.double().T + ba.double()).float()
    assert_close(y.cpu(), want, rtol=1e-4, atol=1e-5, what="action_embed")


@pytest.mark.parametrize("mode", ["denoised", "ddim", "raw"])
def test_head(lib, mode):
    g = torch.Generator().manual_seed(4)
    B, T, A, D, sd = 6, 10, 7, 384, 0.5
    M = B * T
    y, lw = torch.randn(M, D, generator=g) * 2, torch.randn(D, generator=g) * 0.1 + 1
    Wp, bp = torch.randn(A, D, generator=g) / 20, torch.randn(A, generator=g)
    Wa, ba = torch.randn(D, A, generator=g), torch.randn(D, generator=g)
    x = torch.randn(M, A, generator=g) * 10
    sigma = torch.rand(B, generator=g) * 20 + 0.05
    step = torch.tensor([0.3, 0.7, 2.5])  # ratio, coef, sigma_next
    t = {k: dev(v) for k, v in dict(y=y, lw=lw, Wp=Wp, bp=bp, Wa=Wa.T, ba=ba, x=x, sigma=sigma, step=step).items()}
    out = torch.empty(M, A, device="cuda")
    y_next = torch.empty(M, D, device="cuda")
    a = lib.HeadArgs()
    a.y, a.ln_w, a.Wp, a.bp, a.x = (t[k].data_ptr() for k in ("y", "lw", "Wp", "bp", "x"))
    a.sigma, a.sigma_stride, a.out = t["sigma"].data_ptr(), 1, out.data_ptr()
    a.M, a.D, a.A, a.rows_per_sample, a.mode, a.sigma_data = M, D, A, T, lib.HEAD[mode], sd
    a.step = t["step"].data_ptr()
    if mode == "ddim":
        a.y_next, a.Wa, a.ba = y_next.data_ptr(), t["Wa"].data_ptr(), t["ba"].data_ptr()
    lib.check(lib.load().mdt_op_head(C.byref(a), stream()))
    torch.cuda.synchronize()
    Fv = F.layer_norm(y.double(), (D,), lw.double(), None, 1e-5) @ Wp.double().T + bp.double()
    s = sigma.double().repeat_interleave(T)[:, None]
    den = Fv * (s * sd / (s ** 2 + sd ** 2).sqrt()) + x.double() * (sd ** 2 / (s ** 2 + sd ** 2))
    want = {"raw": Fv, "denoised": den, "ddim": 0.3 * x.double() + 0.7 * den}[mode]
    assert_close(out.cpu(), want.float(), rtol=1e-4, atol=1e-4, what=f"head {mode}")
    if mode == "ddim":
        wn = (want / math.sqrt(2.5 ** 2 + sd ** 2)) @ Wa.double().T + ba.double()
        assert_close(y_next.cpu(), wn.float(), rtol=1e-4, atol=1e-4, what="fused next action_emb")


@pytest.mark.parametrize("D,H,Te,Ta,bias", [(384, 8, 4, 10, False), (512, 8, 3, 10, True), (128, 8, 4, 10, False),
                                            (384, 8, 3, 7, True), (256, 8, 2, 16, False), (128, 4, 4, 10, True),
                                            (384, 8, 1, 10, False)])
def test_collapsed_cross_attention(lib, D, H, Te, Ta, bias):
    """k_xattn_fold + k_xattn_apply == y + c_proj(softmax_causal((ln3(y) Wq^T + bq) K^T / sqrt(hd)) V) + bo."""
    g = torch.Generator().manual_seed(D + Te + Ta)
    B, hd = 9, D // H
    y = torch.randn(B * Ta, D, generator=g) * 2
    kv = torch.randn(B * Te, 2 * D + 64, generator=g)           # K at column 0, V at column D, padded row stride
    Wq, bq = torch.randn(D, D, generator=g) / math.sqrt(D), torch.randn(D, generator=g) * 0.3
    Wo, bo = torch.randn(D, D, generator=g) / math.sqrt(D), torch.randn(D, generator=g) * 0.3
    lw, lb = torch.randn(D, generator=g) * 0.2 + 1, torch.randn(D, generator=g) * 0.2
    t = {k: dev(v) for k, v in dict(y=y, kv=kv, bq=bq, bo=bo, lw=lw, lb=lb).items()}
    t["WqT_p"], t["Wo_p"] = pack_t(lib, Wq), pack(lib, Wo)  # the fold reads both weights as MFMA fragment images
    assert torch.equal(t["WqT_p"], pack(lib, Wq.T.contiguous()))
    NP = 4 * H  # rows of the folded images: every head padded to 4 context tokens
    U = torch.full((B * NP * D,), float("nan"), device="cuda")
    Wf = torch.full((B * NP * D,), float("nan"), device="cuda")
    c = torch.full((B * NP,), float("nan"), device="cuda")
    f = lib.XFoldArgs()
    f.kv, f.ldkv, f.WqT_p, f.bq, f.Wo_p = t["kv"].data_ptr(), kv.shape[1], t["WqT_p"].data_ptr(), t["bq"].data_ptr(), t["Wo_p"].data_ptr()
    f.U, f.Wf, f.c, f.B, f.H, f.hd, f.D, f.Te = U.data_ptr(), Wf.data_ptr(), c.data_ptr(), B, H, hd, D, Te
    lib.check(lib.load().mdt_op_xattn_fold(C.byref(f), stream()))
    a = lib.XApplyArgs()
    a.y, a.ln_w, a.ln_b, a.U, a.Wf, a.c = t["y"].data_ptr(), t["lw"].data_ptr(), t["lb"].data_ptr(), U.data_ptr(), Wf.data_ptr(), c.data_ptr()
    a.bo = t["bo"].data_ptr() if bias else None
    a.B, a.H, a.D, a.Te, a.Ta = B, H, D, Te, Ta
    lib.check(lib.load().mdt_op_xattn_apply(C.byref(a), stream()))
    torch.cuda.synchronize()
    yd, kd = y.double().view(B, Ta, D), kv.double()
    K, V = kd[:, :D].view(B, Te, D), kd[:, D:2 * D].view(B, Te, D)
    q = F.layer_norm(yd, (D,), lw.double(), lb.double(), 1e-5) @ Wq.double().T + bq.double()
    qh, kh, vh = (x.view(B, -1, H, hd).transpose(1, 2) for x in (q, K, V))
    att = qh @ kh.transpose(-2, -1) / math.sqrt(hd)
    att = att.masked_fill(~torch.ones(Ta, Te, dtype=torch.bool).tril(), float("-inf"))
    o = (att.softmax(-1) @ vh).transpose(1, 2).reshape(B, Ta, D) @ Wo.double().T + (bo.double() if bias else 0)
    assert_close(t["y"].cpu().view(B, Ta, D), (yd + o).float(), rtol=2e-4, atol=2e-4, what="collapsed cross attention")
    assert not torch.isnan(U).any() and not torch.isnan(c).any()


@pytest.mark.parametrize("T,Te,gated,bias,B", [(10, 4, True, False, 256), (10, 4, False, True, 77), (16, 3, True, True, 130),
                                               (7, 2, True, False, 201), (1, 4, True, True, 96)])
def test_one_sample_per_workgroup_through_attention_projection_and_cross_attention(lib, T, Te, gated, bias, B):
    """mdt_op_attn_xattn (k_attn_xattn): causal self-attention -> c_proj + gate + residual -> ln3 -> collapsed cross-attention of
    one sample per workgroup, the rows between the sublayers never leaving LDS; against float64 and bit-for-bit against the two
    launches it replaces (mdt_op_attn_proj in its tiled form, then mdt_op_xattn_apply)."""
    g = torch.Generator().manual_seed(17 * T + Te + B)
    H, hd = 8, 48
    D = N = H * hd
    qkv = torch.randn(B * T, 3 * D, generator=g)
    W, b = torch.randn(N, D, generator=g) / math.sqrt(D), torch.randn(N, generator=g) * 0.1
    gate, y0 = torch.randn(B, 6 * N, generator=g), torch.randn(B * T, N, generator=g) * 2
    kv = torch.randn(B * Te, 2 * D, generator=g)
    Wq, bq = torch.randn(D, D, generator=g) / math.sqrt(D), torch.randn(D, generator=g) * 0.3
    Wo, bo = torch.randn(D, D, generator=g) / math.sqrt(D), torch.randn(D, generator=g) * 0.3
    lw, lb = torch.randn(D, generator=g) * 0.2 + 1, torch.randn(D, generator=g) * 0.2
    # float64: the two sublayers
    q3 = qkv.view(B, T, 3 * D)
    att = ref_attn(q3[..., :D], q3[..., D:2 * D], q3[..., 2 * D:], H, True).double().reshape(B * T, D)
    upd = att @ W.double().T + b.double()
    if gated:
        upd = gate[:, 2 * N:3 * N].double().repeat_interleave(T, 0) * upd
    y1 = (y0.double() + upd).view(B, T, D)
    kd = kv.double()
    K, V = kd[:, :D].view(B, Te, D), kd[:, D:].view(B, Te, D)
    q = F.layer_norm(y1, (D,), lw.double(), lb.double(), 1e-5) @ Wq.double().T + bq.double()
    qh, kh, vh = (x.view(B, -1, H, hd).transpose(1, 2) for x in (q, K, V))
    sc = qh @ kh.transpose(-2, -1) / math.sqrt(hd)
    sc = sc.masked_fill(~torch.ones(T, Te, dtype=torch.bool).tril(), float("-inf"))
    o = (sc.softmax(-1) @ vh).transpose(1, 2).reshape(B, T, D) @ Wo.double().T + (bo.double() if bias else 0)
    want = (y1 + o).reshape(B * T, D).float()
    # device operands
    t = {k: dev(v) for k, v in dict(kv=kv, bq=bq, bo=bo, lw=lw, lb=lb, b=b, gate=gate, qkv=qkv).items()}
    t["WqT_p"], t["Wo_p"] = pack_t(lib, Wq), pack(lib, Wo)
    Pd = pack(lib, W)
    NP = 4 * H  # rows of the folded images: every head padded to 4 context tokens
    U, Wf, c = (torch.empty(n, device="cuda") for n in (B * NP * D, B * NP * D, B * NP))
    f = lib.XFoldArgs()
    f.kv, f.ldkv, f.WqT_p, f.bq, f.Wo_p = t["kv"].data_ptr(), 2 * D, t["WqT_p"].data_ptr(), t["bq"].data_ptr(), t["Wo_p"].data_ptr()
    f.U, f.Wf, f.c, f.B, f.H, f.hd, f.D, f.Te = U.data_ptr(), Wf.data_ptr(), c.data_ptr(), B, H, hd, D, Te
    lib.check(lib.load().mdt_op_xattn_fold(C.byref(f), stream()))

    def args(out):
        a = lib.GemmArgs()
        a.A, a.lda, a.Wp, a.bias, a.out, a.ldo, a.M, a.N, a.K = None, D, Pd.data_ptr(), t["b"].data_ptr(), out.data_ptr(), N, B * T, N, D
        a.shift_off = a.scale_off = -1
        a.gate_off = 2 * N if gated else -1
        a.mod, a.mod_stride = (t["gate"].data_ptr(), 6 * N) if gated else (None, 0)
        a.residual, a.rows_per_sample, a.gin, a.gout = 1, T, 1, 1
        x = lib.XApplyArgs()
        x.y, x.ln_w, x.ln_b, x.U, x.Wf, x.c = out.data_ptr(), t["lw"].data_ptr(), t["lb"].data_ptr(), U.data_ptr(), Wf.data_ptr(), c.data_ptr()
        x.bo = t["bo"].data_ptr() if bias else None
        x.B, x.H, x.D, x.Te, x.Ta = B, H, D, Te, T
        return a, x

    one, two = dev(y0).clone(), dev(y0).clone()
    a, x = args(one)
    lib.check(lib.load().mdt_op_attn_xattn(C.byref(a), t["qkv"].data_ptr(), 3 * D, C.byref(x), hd, T, stream()))
    a2, x2 = args(two)
    lib.check(lib.load().mdt_op_attn_proj(C.byref(a2), t["qkv"].data_ptr(), 3 * D, hd, T, 1, stream()))
    lib.check(lib.load().mdt_op_xattn_apply(C.byref(x2), stream()))
    torch.cuda.synchronize()
    assert_close(one.cpu(), want, rtol=2e-4, atol=2e-4, what="attention + projection + cross-attention, one launch")
    # (round 5: the one-launch form runs the self-attention's two products on the MFMA pipe, the two-launch form in vector FMAs:
    #  same arithmetic, another summation order)
    assert_close(one.cpu(), two.cpu(), rtol=2e-5, atol=2e-5, what="the one-launch form vs attn_proj + xattn_apply")
    # what it refuses: other head dimensions, a cross-attention on other rows
    x.y = two.data_ptr()
    assert lib.load().mdt_op_attn_xattn(C.byref(a), t["qkv"].data_ptr(), 3 * D, C.byref(x), hd, T, stream()) == 2


@pytest.mark.parametrize("D,H,Te,Ta,B,mod", [(384, 8, 4, 10, 1, True), (384, 8, 4, 10, 8, True), (512, 8, 3, 10, 3, False),
                                             (128, 4, 2, 16, 5, True), (256, 8, 1, 7, 2, True)])
def test_cross_attention_inside_the_linear_that_follows_it(lib, D, H, Te, Ta, B, mod):
    """mdt_op_xattn_gemm (k_xattn_gemm_smallm, rollout batches): y += cross-attention, then GELU(LN-mod(y) W^T + b) with every
    16-column workgroup repeating the cross-attention -- bit-for-bit the two launches (mdt_op_xattn_apply, mdt_op_gemm), and
    against float64."""
    g = torch.Generator().manual_seed(D + 7 * Te + Ta + B)
    hd, N, NP = D // H, 4 * D, 4 * H
    y0 = torch.randn(B * Ta, D, generator=g) * 2
    kv = torch.randn(B * Te, 2 * D, generator=g)
    Wq, bq = torch.randn(D, D, generator=g) / math.sqrt(D), torch.randn(D, generator=g) * 0.3
    Wo, bo = torch.randn(D, D, generator=g) / math.sqrt(D), torch.randn(D, generator=g) * 0.3
    lw3, lb3 = torch.randn(D, generator=g) * 0.2 + 1, torch.randn(D, generator=g) * 0.2
    lw2, lb2 = torch.randn(D, generator=g) * 0.2 + 1, torch.randn(D, generator=g) * 0.2
    W, b = torch.randn(N, D, generator=g) / math.sqrt(D), torch.randn(N, generator=g) * 0.1
    modv = torch.randn(6 * D, generator=g) * 0.5
    t = {k: dev(v) for k, v in dict(kv=kv, bq=bq, bo=bo, lw3=lw3, lb3=lb3, lw2=lw2, lb2=lb2, b=b, mod=modv).items()}
    t["WqT_p"], t["Wo_p"] = pack_t(lib, Wq), pack(lib, Wo)
    Pd = pack(lib, W)
    U, Wf, c = (torch.empty(n, device="cuda") for n in (B * NP * D, B * NP * D, B * NP))
    f = lib.XFoldArgs()
    f.kv, f.ldkv, f.WqT_p, f.bq, f.Wo_p = t["kv"].data_ptr(), 2 * D, t["WqT_p"].data_ptr(), t["bq"].data_ptr(), t["Wo_p"].data_ptr()
    f.U, f.Wf, f.c, f.B, f.H, f.hd, f.D, f.Te = U.data_ptr(), Wf.data_ptr(), c.data_ptr(), B, H, hd, D, Te
    lib.check(lib.load().mdt_op_xattn_fold(C.byref(f), stream()))

    def args(y, out):
        x = lib.XApplyArgs()
        x.y, x.ln_w, x.ln_b, x.U, x.Wf, x.c, x.bo = (y.data_ptr(), t["lw3"].data_ptr(), t["lb3"].data_ptr(), U.data_ptr(), Wf.data_ptr(),
                                                     c.data_ptr(), t["bo"].data_ptr())
        x.B, x.H, x.D, x.Te, x.Ta = B, H, D, Te, Ta
        a = lib.GemmArgs()
        a.A, a.lda, a.Wp, a.bias, a.out, a.ldo, a.M, a.N, a.K = y.data_ptr(), D, Pd.data_ptr(), t["b"].data_ptr(), out.data_ptr(), N, B * Ta, N, D
        a.ln, a.ln_w, a.ln_b, a.act = 1, t["lw2"].data_ptr(), t["lb2"].data_ptr(), 1
        a.shift_off = a.scale_off = a.gate_off = -1
        if mod:
            a.mod, a.mod_stride, a.shift_off, a.scale_off = t["mod"].data_ptr(), 0, 3 * D, 4 * D
        a.rows_per_sample, a.gin, a.gout = Ta, 1, 1
        return x, a

    y1, y2, y1n = dev(y0).clone(), dev(y0).clone(), torch.full((B * Ta, D), float("nan"), device="cuda")
    o1, o2 = torch.empty(B * Ta, N, device="cuda"), torch.empty(B * Ta, N, device="cuda")
    x, a = args(y1, o1)
    assert lib.load().mdt_op_xattn_gemm(C.byref(x), C.byref(a), stream()) == 2  # in place: the repeating workgroups would race
    x.y_out = y1n.data_ptr()
    lib.check(lib.load().mdt_op_xattn_gemm(C.byref(x), C.byref(a), stream()))
    x2, a2 = args(y2, o2)
    lib.check(lib.load().mdt_op_xattn_apply(C.byref(x2), stream()))
    try:  # the two-launch reference on the split-K kernel the fused launch is built from (the dispatcher may prefer tiles here)
        lib.load().mdt_op_set_gemm_geometry(-1)
        lib.check(lib.load().mdt_op_gemm(C.byref(a2), stream()))
    finally:
        lib.load().mdt_op_set_gemm_geometry(0)
    torch.cuda.synchronize()
    assert torch.equal(y1.cpu(), y0), "the input rows must stay as they are"
    assert torch.equal(y1n, y2) and torch.equal(o1, o2), "the one-launch form differs from xattn_apply + gemm"
    y1 = y1n
    yd, kd = y0.double().view(B, Ta, D), kv.double()
    K, V = kd[:, :D].view(B, Te, D), kd[:, D:].view(B, Te, D)
    q = F.layer_norm(yd, (D,), lw3.double(), lb3.double(), 1e-5) @ Wq.double().T + bq.double()
    qh, kh, vh = (z.view(B, -1, H, hd).transpose(1, 2) for z in (q, K, V))
    sc = (qh @ kh.transpose(-2, -1) / math.sqrt(hd)).masked_fill(~torch.ones(Ta, Te, dtype=torch.bool).tril(), float("-inf"))
    ynew = yd + (sc.softmax(-1) @ vh).transpose(1, 2).reshape(B, Ta, D) @ Wo.double().T + bo.double()
    h = F.layer_norm(ynew, (D,), lw2.double(), lb2.double(), 1e-5)
    if mod:
        h = modv[3 * D:4 * D].double() + h * modv[4 * D:5 * D].double()
    want = F.gelu(h.reshape(B * Ta, D) @ W.double().T + b.double())
    assert_close(y1.cpu(), ynew.reshape(B * Ta, D).float(), rtol=2e-4, atol=2e-4, what="residual stream after the cross-attention")
    assert_close(o1.cpu(), want.float(), rtol=2e-4, atol=3e-4, what="c_fc on the cross-attention's output")


# ------------------------------------------------------------------------------------------------
# fused MLP sublayer (mdt_op_mlp) and the readers of its partial slabs (mdt_gemm_args.a_parts, mdt_head_args.y_parts)
# ------------------------------------------------------------------------------------------------
def _gemm_args(lib, keep, A, lda, Wp, N, K, M, out=None, ldo=0, bias=None):
    a = lib.GemmArgs()
    a.A, a.lda, a.Wp, a.M, a.N, a.K = A.data_ptr(), lda, Wp.data_ptr(), M, N, K
    if out is not None:
        a.out, a.ldo = out.data_ptr(), ldo
    if bias is not None:
        bd = dev(bias); keep.append(bd); a.bias = bd.data_ptr()
    a.shift_off = a.scale_off = a.gate_off = -1
    a.rows_per_sample, a.gin, a.gout, a.goff = 1, 1, 1, 0
    return a


def pack_split(lib, W):
    """Three-way bf16 split fragment image of a (N, K) weight through the library (6 N K bytes)."""
    N, K = W.shape
    Wd = dev(W)
    P = torch.zeros(N * K * 6, dtype=torch.uint8, device="cuda")
    lib.check(lib.load().mdt_op_pack_weight_split(Wd.data_ptr(), N, K, P.data_ptr(), stream()))
    return P


def run_mlp(lib, x, W1, W2, ln_w, ln_b=None, b1=None, b2=None, mod=None, mod_stride=0, offs=None, rps=1, split=False):
    """-> (slabs (S, M, D) on the host, S).  offs = (shift, scale, gate) offsets inside a mod row or None.
    split: the launch in its three-way bf16 split form (mdt_op_mlp_split)."""
    M, D = x.shape
    keep = []
    xd, P1, P2 = dev(x), pack(lib, W1), pack(lib, W2)
    S = 4 * D // 512
    parts = torch.full((S, M, D), float("nan"), device="cuda")
    f = _gemm_args(lib, keep, xd, D, P1, 4 * D, D, M, bias=b1)
    p = _gemm_args(lib, keep, xd, D, P2, D, 4 * D, M, bias=b2)
    p.ldo = D
    f.ln, f.act = 1, lib.ACT["gelu"]
    lw = dev(ln_w); keep.append(lw); f.ln_w = lw.data_ptr()
    if ln_b is not None:
        lb = dev(ln_b); keep.append(lb); f.ln_b = lb.data_ptr()
    if mod is not None:
        md = dev(mod); keep.append(md)
        f.mod = p.mod = md.data_ptr()
        f.mod_stride = p.mod_stride = mod_stride
        f.shift_off, f.scale_off, p.gate_off = offs
        f.rows_per_sample = p.rows_per_sample = rps
    n = C.c_int32(0)
    if split:
        S1, S2 = pack_split(lib, W1), pack_split(lib, W2)
        lib.check(lib.load().mdt_op_mlp_split(C.byref(f), C.byref(p), S1.data_ptr(), S2.data_ptr(), parts.data_ptr(), M * D, C.byref(n),
                                              stream()))
    else:
        lib.check(lib.load().mdt_op_mlp(C.byref(f), C.byref(p), parts.data_ptr(), M * D, C.byref(n), stream()))
    torch.cuda.synchronize()
    assert n.value == S
    return parts, S


def ref_mlp(x, W1, W2, ln_w, ln_b, b1, b2, shift=None, scale=None, gate=None):
    x = x.double()
    h = F.layer_norm(x, (x.shape[1],), ln_w.double(), None if ln_b is None else ln_b.double(), 1e-5)
    if shift is not None:
        h = shift.double() + h * scale.double()
    u = h @ W1.double().T + (0 if b1 is None else b1.double())
    v = F.gelu(u) @ W2.double().T + (0 if b2 is None else b2.double())
    return x + (v if gate is None else gate.double() * v)


@pytest.mark.parametrize("M,D,case", [(2560, 384, "bcast"), (2560, 384, "rows"), (2560, 384, "plain_bias"), (1777, 384, "bcast"),
                                      (45, 384, "rows"), (320, 512, "bcast"), (200, 256, "rows"), (96, 128, "plain_bias")])
def test_fused_mlp_slabs_sum_to_the_sublayer(lib, M, D, case):
    """k_mlp: LayerNorm (+ modulate) -> c_fc -> GELU -> c_proj -> gate -> residual as one launch; the slabs it leaves add
    up (in slab order) to the float64 sublayer, slab 0 carrying the residual and the second bias."""
    g = torch.Generator().manual_seed(M + D + len(case))
    T = 10 if M % 10 == 0 else (5 if M % 5 == 0 else 1)
    x = torch.randn(M, D, generator=g) * 1.5 + 0.2
    W1 = torch.randn(4 * D, D, generator=g) / math.sqrt(D)
    W2 = torch.randn(D, 4 * D, generator=g) / math.sqrt(4 * D)
    lw = torch.randn(D, generator=g) * 0.2 + 1
    kw, ref = {}, {}
    lb = b1 = b2 = None
    if case == "plain_bias":
        lb, b1, b2 = torch.randn(D, generator=g) * 0.2, torch.randn(4 * D, generator=g) * 0.3, torch.randn(D, generator=g) * 0.3
    else:
        nb = M // T if case == "rows" else 1
        mod = torch.randn(nb, 6 * D, generator=g) * 0.5
        kw = dict(mod=mod, mod_stride=6 * D if case == "rows" else 0, offs=(3 * D, 4 * D, 5 * D), rps=T)
        pick = (lambda o: mod[:, o:o + D].repeat_interleave(T, 0)) if case == "rows" else (lambda o: mod[:, o:o + D])
        ref = dict(shift=pick(3 * D), scale=pick(4 * D), gate=pick(5 * D))
    parts, S = run_mlp(lib, x, W1, W2, lw, ln_b=lb, b1=b1, b2=b2, **kw)
    got = parts[0].clone()
    for s in range(1, S):
        got += parts[s]
    want = ref_mlp(x, W1, W2, lw, lb, b1, b2, **ref).float()
    assert_close(got.cpu(), want, rtol=2e-4, atol=2e-4, what=f"fused mlp {M}x{D} {case}")


def expected_pack_split(W):
    """The split image as mdt_mlp_split.h documents it: fragment (nt, kk, part) = 1 KiB at ((nt K/32 + kk) 3 + part) 1024, lane l's
    eight bf16 = part of W[16 nt + l % 16][32 kk + 16 h + 4 (l / 16) + e] at slot 4 h + e; parts by round-to-nearest bf16 of the
    running remainder."""
    N, K = W.shape
    x = W.clone()
    parts = []
    for _ in range(3):
        b = x.to(torch.bfloat16)
        parts.append(b)
        x = x - b.float()
    out = torch.zeros(N // 16, K // 32, 3, 64, 8, dtype=torch.bfloat16)
    for pi, P in enumerate(parts):
        t = P.reshape(N // 16, 16, K // 32, 2, 4, 4)           # nt, ni, kk, h, g, e
        out[:, :, pi] = t.permute(0, 2, 4, 1, 3, 5).reshape(N // 16, K // 32, 64, 8)   # lane = ni + 16 g, slot = 4 h + e
    return out.reshape(-1)


@pytest.mark.parametrize("N,K", [(1536, 384), (384, 1536), (16, 32)])
def test_pack_weight_split_layout_and_exactness(lib, N, K):
    """mdt_op_pack_weight_split: the documented fragment layout, bit for bit; and the three parts add up to the weight EXACTLY
    (each part is the bf16 rounding of what the ones before it left)."""
    g = torch.Generator().manual_seed(N + K)
    W = torch.randn(N, K, generator=g) * torch.exp(3.0 * torch.randn(N, 1, generator=g))
    got = pack_split(lib, W).cpu().view(torch.bfloat16)
    want = expected_pack_split(W)
    assert torch.equal(got.view(torch.int16), want.view(torch.int16)), "split image differs from the documented layout"
    im = got.reshape(N // 16, K // 32, 3, 64, 8).double().sum(dim=2)          # nt, kk, lane, slot
    back = im.reshape(N // 16, K // 32, 4, 16, 2, 4).permute(0, 3, 1, 4, 2, 5).reshape(N, K)   # nt, ni, kk, h, g, e
    assert torch.equal(back.float(), W), "the three parts do not add up to the fp32 weight"


@pytest.mark.parametrize("M,D,case", [(2560, 384, "bcast"), (2560, 384, "rows"), (2560, 384, "plain_bias"), (1777, 384, "bcast"),
                                      (45, 384, "rows"), (200, 256, "rows"), (33, 256, "plain_bias"), (320, 512, "bcast"),
                                      (1000, 512, "rows"), (77, 512, "plain_bias")])
def test_fused_mlp_in_its_bf16_split_form_keeps_fp32_accuracy(lib, M, D, case):
    """k_mlp_split (round 6): the fused MLP launch with every contraction as six bf16 MFMA products of three-way split operands.
    Not the fp32 launch's bits: against the float64 sublayer its error must stay within 1.5x the fp32 launch's own on the same
    inputs (and inside that launch's test tolerance), for every prologue / epilogue variant and ragged row counts."""
    g = torch.Generator().manual_seed(M + D + len(case))
    T = 10 if M % 10 == 0 else (5 if M % 5 == 0 else 1)
    x = torch.randn(M, D, generator=g) * 1.5 + 0.2
    W1 = torch.randn(4 * D, D, generator=g) / math.sqrt(D)
    W2 = torch.randn(D, 4 * D, generator=g) / math.sqrt(4 * D)
    lw = torch.randn(D, generator=g) * 0.2 + 1
    kw, ref = {}, {}
    lb = b1 = b2 = None
    if case == "plain_bias":
        lb, b1, b2 = torch.randn(D, generator=g) * 0.2, torch.randn(4 * D, generator=g) * 0.3, torch.randn(D, generator=g) * 0.3
    else:
        nb = M // T if case == "rows" else 1
        mod = torch.randn(nb, 6 * D, generator=g) * 0.5
        kw = dict(mod=mod, mod_stride=6 * D if case == "rows" else 0, offs=(3 * D, 4 * D, 5 * D), rps=T)
        pick = (lambda o: mod[:, o:o + D].repeat_interleave(T, 0)) if case == "rows" else (lambda o: mod[:, o:o + D])
        ref = dict(shift=pick(3 * D), scale=pick(4 * D), gate=pick(5 * D))
    want = ref_mlp(x, W1, W2, lw, lb, b1, b2, **ref)
    err = {}
    for split in (False, True):
        parts, S = run_mlp(lib, x, W1, W2, lw, ln_b=lb, b1=b1, b2=b2, split=split, **kw)
        got = parts[0].clone()
        for s in range(1, S):
            got += parts[s]
        err[split] = (got.cpu().double() - want).abs().max().item()
        if split:
            assert_close(got.cpu(), want.float(), rtol=2e-4, atol=2e-4, what=f"split fused mlp {M}x{D} {case}")
    assert err[True] <= 1.5 * err[False] + 1e-6, f"split form: max error {err[True]:.3g} against the fp32 launch's {err[False]:.3g}"


@pytest.mark.parametrize("M,D", [(2560, 384), (1777, 384), (320, 512), (96, 128)])
def test_fused_mlp_wave_schedules_give_the_same_bits(lib, M, D):
    """mdt_op_set_mlp_skew: the two waves of a SIMD in lockstep with a workgroup barrier between the products, or k-steps
    apart with per-wave LDS flags (with / without raised MFMA priority) -- the K order never changes, so every schedule
    must reproduce the lockstep bits, launch after launch (a missed flag would show up as a stale hidden column)."""
    g = torch.Generator().manual_seed(M * 3 + D)
    x = torch.randn(M, D, generator=g) * 1.5 + 0.2
    W1 = torch.randn(4 * D, D, generator=g) / math.sqrt(D)
    W2 = torch.randn(D, 4 * D, generator=g) / math.sqrt(4 * D)
    lw = torch.randn(D, generator=g) * 0.2 + 1
    mod = torch.randn(1, 6 * D, generator=g) * 0.5
    kw = dict(mod=mod, mod_stride=0, offs=(3 * D, 4 * D, 5 * D), rps=1)
    L = lib.load()
    try:
        L.mdt_op_set_mlp_skew(0)
        want, _ = run_mlp(lib, x, W1, W2, lw, **kw)
        for v in (3, 6, 18, 24, 40, 6 | 256, 18 | 256, 21 | 256, 255 | 256):
            L.mdt_op_set_mlp_skew(v)
            for rep in range(6):
                got, _ = run_mlp(lib, x, W1, W2, lw, **kw)
                assert torch.equal(got, want), f"schedule {v} (launch {rep}) differs from lockstep"
    finally:
        L.mdt_op_set_mlp_skew(-1)


@pytest.mark.parametrize("M,N,XP,case", [(2560, 1152, 3, "bcast"), (2560, 1152, 3, "rows"), (1530, 1152, 3, "ln"),
                                         (640, 1536, 4, "bcast"), (77, 768, 2, "ln")])
def test_gemm_reads_the_sum_of_slabs(lib, M, N, XP, case):
    """mdt_gemm_args.a_parts: the LayerNorm prologue adds XP slabs in order, bit-identical to the same GEMM on the
    pre-summed rows, and leaves the sum in a_merged."""
    g = torch.Generator().manual_seed(M + N + XP)
    D = 128 * XP
    T = 10 if M % 10 == 0 else 1
    slabs = torch.randn(XP, M, D, generator=g)
    W, b = torch.randn(N, D, generator=g) / math.sqrt(D), torch.randn(N, generator=g)
    lw, lb = torch.randn(D, generator=g) * 0.2 + 1, torch.randn(D, generator=g) * 0.2
    summed = slabs[0].clone()
    for s in range(1, XP):
        summed = summed + slabs[s]            # fp32, slab order: what the kernel does
    kw = {}
    if case != "ln":
        nb = M // T if case == "rows" else 1
        mod = torch.randn(nb, 2 * D, generator=g) * 0.5
        kw = dict(mod=mod, mod_stride=2 * D if case == "rows" else 0, shift_off=0, scale_off=D, rps=T)
    try:
        lib.load().mdt_op_set_gemm_geometry(3 if N % 384 == 0 else 4)
        want = run_gemm(lib, summed, W, bias=b, ln_w=lw, ln_b=lb, **kw)
    finally:
        lib.load().mdt_op_set_gemm_geometry(0)
    keep = []
    sd, Pd = dev(slabs), pack(lib, W)
    out = torch.full((M, N), float("nan"), device="cuda")
    merged = torch.full((M, D), float("nan"), device="cuda")
    a = _gemm_args(lib, keep, sd, D, Pd, N, D, M, out=out, ldo=N, bias=b)
    a.ln = 1
    for name, t in (("ln_w", lw), ("ln_b", lb)):
        td = dev(t); keep.append(td); setattr(a, name, td.data_ptr())
    if kw:
        md = dev(kw["mod"]); keep.append(md)
        a.mod, a.mod_stride, a.shift_off, a.scale_off, a.rows_per_sample = md.data_ptr(), kw["mod_stride"], 0, D, T
    a.a_parts, a.a_part_stride, a.a_merged = XP, M * D, merged.data_ptr()
    lib.check(lib.load().mdt_op_gemm(C.byref(a), stream()))
    torch.cuda.synchronize()
    assert torch.equal(merged.cpu(), summed), "a_merged is not the slab-order sum"
    assert torch.equal(out.cpu(), want), "merge-on-read GEMM differs from the GEMM on the summed rows"


@pytest.mark.parametrize("M,N,D,XP,case", [(2560, 1152, 384, 3, "bcast"), (2560, 1152, 384, 1, "rows"), (1530, 1152, 384, 3, "ln"),
                                           (1777, 768, 384, 2, "bcast"), (800, 384, 384, 4, "ln"), (1000, 1536, 512, 4, "bcast"),
                                           (900, 384, 512, 1, "ln")])
def test_layernorm_gemm_in_its_bf16_split_form_keeps_fp32_accuracy(lib, M, N, D, XP, case):
    """mdt_gemm_args.Wp_split (round 6): the wide LayerNorm-prologue product -- the decoder's qkv GEMM, rows = the sum of the fused
    MLP's slabs -- as six bf16 MFMA products of three-way split operands per k32 step.  a_merged must still be the exact slab-order
    sum; the output, against float64, within 1.5x the error of the fp32 form (Wp_split = NULL) on the same inputs; stacked weights
    (query | key | value packed row range by row range) included."""
    g = torch.Generator().manual_seed(M + N + XP + D)
    T = 10 if M % 10 == 0 else 1
    slabs = torch.randn(XP, M, D, generator=g)
    W, b = torch.randn(N, D, generator=g) / math.sqrt(D), torch.randn(N, generator=g)
    lw, lb = torch.randn(D, generator=g) * 0.2 + 1, torch.randn(D, generator=g) * 0.2
    summed = slabs[0].clone()
    for s in range(1, XP):
        summed = summed + slabs[s]
    kw = {}
    h = F.layer_norm(summed.double(), (D,), lw.double(), lb.double(), 1e-5)
    if case != "ln":
        nb = M // T if case == "rows" else 1
        mod = torch.randn(nb, 2 * D, generator=g) * 0.5
        kw = dict(mod=mod, mod_stride=2 * D if case == "rows" else 0)
        pick = (lambda o: mod[:, o:o + D].repeat_interleave(T, 0)) if case == "rows" else (lambda o: mod[:, o:o + D])
        h = pick(0).double() + h * pick(D).double()
    want = h @ W.double().T + b.double()
    keep = []
    sd, Pd = dev(slabs), pack(lib, W)
    # the split image, packed in three row ranges like the model's stacked query | key | value
    Wd = dev(W)
    S = torch.zeros(N * D * 6, dtype=torch.uint8, device="cuda")
    third = (N // 3) // 16 * 16
    for lo, hi in ((0, third), (third, 2 * third), (2 * third, N)):
        lib.check(lib.load().mdt_op_pack_weight_split_rows(Wd[lo:hi].contiguous().data_ptr(), hi - lo, D, S.data_ptr(), lo, stream()))
    assert torch.equal(S, pack_split(lib, W)), "row-range packs differ from the one-piece image"
    err = {}
    lib.load().mdt_op_set_mlp_split(1)   # (whatever MDT_HIP_MLP_SPLIT says: this test is about the split form)
    for split in (False, True):
        out = torch.full((M, N), float("nan"), device="cuda")
        merged = torch.full((M, D), float("nan"), device="cuda")
        a = _gemm_args(lib, keep, sd, D, Pd, N, D, M, out=out, ldo=N, bias=b)
        a.ln = 1
        for name, t in (("ln_w", lw), ("ln_b", lb)):
            td = dev(t); keep.append(td); setattr(a, name, td.data_ptr())
        if kw:
            md = dev(kw["mod"]); keep.append(md)
            a.mod, a.mod_stride, a.shift_off, a.scale_off, a.rows_per_sample = md.data_ptr(), kw["mod_stride"], 0, D, T
        if XP > 1:
            a.a_parts, a.a_part_stride, a.a_merged = XP, M * D, merged.data_ptr()
        if split:
            a.Wp_split = S.data_ptr()
        lib.check(lib.load().mdt_op_gemm(C.byref(a), stream()))
        torch.cuda.synchronize()
        if XP > 1:
            assert torch.equal(merged.cpu(), summed), "a_merged is not the slab-order sum"
        err[split] = (out.cpu().double() - want).abs().max().item()
        if split:
            assert_close(out.cpu(), want.float(), rtol=1e-4, atol=1e-4, what="split LayerNorm GEMM")
            res_split = out.cpu()
        else:
            res_fp32 = out.cpu()
    lib.load().mdt_op_set_mlp_split(-1)
    assert not torch.equal(res_split, res_fp32), "the split form did not run"
    assert err[True] <= 1.5 * err[False] + 1e-6, f"split form: max error {err[True]:.3g} against the fp32 form's {err[False]:.3g}"


@pytest.mark.parametrize("M,XP", [(2560, 3), (77, 4), (10, 2)])
def test_head_reads_the_sum_of_slabs(lib, M, XP):
    """mdt_head_args.y_parts: identical to the head on the pre-summed rows (DDIM mode with the next step's embedding)."""
    g = torch.Generator().manual_seed(M + XP)
    D, A, T = 128 * XP, 7, 10 if M % 10 == 0 else 1
    slabs = torch.randn(XP, M, D, generator=g)
    summed = slabs[0].clone()
    for s in range(1, XP):
        summed = summed + slabs[s]
    lw = torch.randn(D, generator=g) * 0.2 + 1
    Wp, bp = torch.randn(A, D, generator=g) / math.sqrt(D), torch.randn(A, generator=g) * 0.1
    WaT, ba = torch.randn(A, D, generator=g) * 0.3, torch.randn(D, generator=g) * 0.1
    x = torch.randn(M, A, generator=g)
    sig = torch.tensor([3.0]); step = torch.tensor([0.6, 0.4, 1.8, 3.0])

    def run(y, parts):
        keep = [dev(t) for t in (y, lw, Wp, bp, x, sig, step, WaT, ba)]
        yd, lwd, Wpd, bpd, xd, sd, std, Wad, bad = keep
        out = torch.full((M, A), float("nan"), device="cuda")
        ynext = torch.full((M, D), float("nan"), device="cuda")
        a = lib.HeadArgs()
        a.y, a.ln_w, a.Wp, a.bp, a.x, a.sigma, a.sigma_stride, a.out = (yd.data_ptr(), lwd.data_ptr(), Wpd.data_ptr(), bpd.data_ptr(),
                                                                      xd.data_ptr(), sd.data_ptr(), 0, out.data_ptr())
        a.M, a.D, a.A, a.rows_per_sample, a.mode, a.step, a.sigma_data = M, D, A, T, lib.HEAD["ddim"], std.data_ptr(), 0.5
        a.y_next, a.Wa, a.ba = ynext.data_ptr(), Wad.data_ptr(), bad.data_ptr()
        if parts > 1:
            a.y_parts, a.y_part_stride = parts, M * D
        lib.check(lib.load().mdt_op_head(C.byref(a), stream()))
        torch.cuda.synchronize()
        return out.cpu(), ynext.cpu()

    want, want_next = run(summed, 1)
    got, got_next = run(slabs, XP)
    assert torch.equal(got, want) and torch.equal(got_next, want_next)
