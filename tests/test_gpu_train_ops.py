"""Kernel-level parity of the TRAINING path (pytest -m gpu): every backward piece, called through the C ABI of
include/mdt_hip_train.h, against torch.autograd (float64 on the CPU) of the same op."""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from tests.helpers import assert_close
from tests.test_gpu_ops import dev, expected_pack, stream

pytestmark = pytest.mark.gpu

G_TOL = dict(rtol=1e-3, atol=2e-5)


@pytest.fixture(scope="module")
def lib():
    from mdt_policy_amd import _lib
    return _lib


def rnd(*shape, seed=0, scale=1.0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale


def test_pack_weight_t_layout(lib):
    W = rnd(48, 32, seed=1)                         # (rows, cols) -> image of W^T (N' = 32, K' = 64) at k offset 16
    P = torch.zeros(32 * 64, device="cuda")
    lib.check(lib.load().mdt_op_pack_weight_t(dev(W).data_ptr(), 48, 32, 32, P.data_ptr(), 16, 64, stream()))
    full = torch.zeros(32, 64)
    full[:, 16:64] = W.T
    np.testing.assert_array_equal(P.cpu().numpy(), expected_pack(full))


@pytest.mark.parametrize("D,rps,B,mod,bias", [(384, 10, 5, True, False), (384, 4, 3, False, True), (128, 10, 2, True, True),
                                             (512, 3, 2, True, False), (64, 1, 7, False, False), (384, 392, 3, False, True),
                                             (128, 77, 2, False, True)])
def test_layernorm_train_forward_and_backward(lib, D, rps, B, mod, bias):
    M = B * rps
    chunks = 8 if rps == 392 else (2 if rps == 77 else 0)  # long samples: several workgroups per sample
    x, w, b = rnd(M, D, seed=2), 1 + 0.1 * rnd(D, seed=3), (0.1 * rnd(D, seed=4) if bias else None)
    modt = rnd(B, 6 * D, seed=5) if mod else None
    sh, sc = 3 * D, 4 * D
    dh = rnd(M, D, seed=6)
    dx0 = rnd(M, D, seed=7)
    # reference (float64 autograd)
    x64 = x.double().requires_grad_()
    w64 = w.double().requires_grad_()
    b64 = b.double().requires_grad_() if bias else None
    m64 = modt.double().requires_grad_() if mod else None
    n = F.layer_norm(x64, (D,), w64, b64, 1e-5)
    if mod:
        rows = torch.arange(M) // rps
        h = m64[rows, sh:sh + D] + n * m64[rows, sc:sc + D]
    else:
        h = n
    h.backward(dh.double())
    # HIP forward
    L = lib.load()
    xd, wd, bd = dev(x), dev(w), (dev(b) if bias else None)
    md = dev(modt) if mod else None
    out = torch.empty(M, D, device="cuda")
    stats = torch.empty(M, 2, device="cuda")
    a = lib.LnTrainArgs(x=xd.data_ptr(), w=wd.data_ptr(), b=bd.data_ptr() if bias else None,
                        mod=md.data_ptr() if mod else None, mod_stride=6 * D, shift_off=sh, scale_off=sc,
                        rows_per_sample=rps, out=out.data_ptr(), stats=stats.data_ptr(), M=M, D=D)
    lib.check(L.mdt_op_ln_fwd_train(C.byref(a), stream()))
    assert_close(out.cpu(), h.detach(), what="ln forward")
    # HIP backward (accumulating into an existing dx, as the residual path does)
    dhd, dx = dev(dh), dev(dx0).clone()
    dmod = torch.zeros(B, 6 * D, device="cuda")
    pw, pb = torch.empty(B * max(chunks, 1), D, device="cuda"), torch.empty(B * max(chunks, 1), D, device="cuda")
    g = lib.LnBwdArgs(x=xd.data_ptr(), stats=stats.data_ptr(), w=wd.data_ptr(), b=bd.data_ptr() if bias else None,
                      mod=md.data_ptr() if mod else None, mod_stride=6 * D, shift_off=sh if mod else -1,
                      scale_off=sc if mod else -1, dh=dhd.data_ptr(), ld_dh=D, dx=dx.data_ptr(), accumulate=1,
                      d_mod=dmod.data_ptr() if mod else None, d_mod_stride=6 * D, pw=pw.data_ptr(), pb=pb.data_ptr(),
                      B=B, rows_per_sample=rps, D=D, row_chunks=chunks)
    lib.check(L.mdt_op_ln_bwd(C.byref(g), stream()))
    assert_close(dx.cpu() - dx0, x64.grad, what="dx", **G_TOL)
    dw = torch.zeros(D, device="cuda")
    lib.check(L.mdt_op_colsum(pw.data_ptr(), D, pw.shape[0], D, dw.data_ptr(), 0, stream()))
    assert_close(dw.cpu(), w64.grad, what="dw", **G_TOL)
    if bias:
        assert_close(pb.sum(0).cpu(), b64.grad, what="db", **G_TOL)
    if mod:
        assert_close(dmod.cpu(), m64.grad, what="d_mod", **G_TOL)


def _rope_tables():
    """(16, 16) cos / sin tables as mdt_create builds them, from the oracle's restatement."""
    from oracle.mdt_oracle import rotary_tables
    cos, sin = rotary_tables(16, torch.float64)                # (16, 32), pairs repeated
    return dev(cos[:, ::2].float().contiguous()), dev(sin[:, ::2].float().contiguous())


def _rot(t, rope):
    if not rope:
        return t
    from oracle.mdt_oracle import apply_rotary
    return apply_rotary(t, 32)


def test_layernorm_shift_only_conditioning_accumulates_its_gradient(lib):
    """NoiseBlock (transformer_blocks.py:335-341): h = ln(x) + c, the same c at several LayerNorms -> d_c adds up."""
    D, rps, B = 128, 10, 4
    M = B * rps
    L = lib.load()
    c = rnd(B, D, seed=61)
    c64 = c.double().requires_grad_()
    cd = dev(c)
    dc = torch.zeros(B, D, device="cuda")
    total = 0
    for i in range(2):
        x, w, dh = rnd(M, D, seed=62 + i), 1 + 0.1 * rnd(D, seed=64 + i), rnd(M, D, seed=66 + i)
        x64 = x.double().requires_grad_()
        h = F.layer_norm(x64, (D,), w.double(), None, 1e-5) + c64.repeat_interleave(rps, 0)
        total = total + (h * dh.double()).sum()
        xd, wd, dhd = dev(x), dev(w), dev(dh)
        out, stats = torch.empty(M, D, device="cuda"), torch.empty(M, 2, device="cuda")
        a = lib.LnTrainArgs(x=xd.data_ptr(), w=wd.data_ptr(), b=None, mod=cd.data_ptr(), mod_stride=D, shift_off=0,
                            scale_off=-1, rows_per_sample=rps, out=out.data_ptr(), stats=stats.data_ptr(), M=M, D=D)
        lib.check(L.mdt_op_ln_fwd_train(C.byref(a), stream()))
        assert_close(out.cpu(), h.detach(), what="ln + c forward")
        dx, pw = torch.zeros(M, D, device="cuda"), torch.empty(B, D, device="cuda")
        g = lib.LnBwdArgs(x=xd.data_ptr(), stats=stats.data_ptr(), w=wd.data_ptr(), b=None, mod=cd.data_ptr(), mod_stride=D,
                          shift_off=0, scale_off=-1, dh=dhd.data_ptr(), ld_dh=D, dx=dx.data_ptr(), accumulate=0,
                          d_mod=dc.data_ptr(), d_mod_stride=D, pw=pw.data_ptr(), pb=None, B=B, rows_per_sample=rps, D=D,
                          row_chunks=0, accumulate_dmod=1)
        lib.check(L.mdt_op_ln_bwd(C.byref(g), stream()))
        gx, = torch.autograd.grad((h * dh.double()).sum(), x64, retain_graph=True)
        assert_close(dx.cpu(), gx, what="dx", **G_TOL)
    total.backward()
    assert_close(dc.cpu(), c64.grad, what="d_c over two LayerNorms", **G_TOL)


@pytest.mark.parametrize("hd,H,Tq,Tk,causal,rope", [(48, 8, 10, 10, 1, 0), (48, 8, 10, 4, 1, 0), (48, 8, 4, 4, 0, 0),
                                                   (16, 8, 10, 5, 1, 0), (64, 2, 16, 16, 1, 0), (32, 4, 7, 3, 0, 0),
                                                   (48, 8, 10, 10, 1, 1), (48, 8, 10, 4, 1, 1), (32, 4, 7, 3, 0, 1),
                                                   (64, 2, 16, 16, 1, 1)])
def test_attention_backward(lib, hd, H, Tq, Tk, causal, rope):
    B, Dm = 3, H * hd
    rc, rs = _rope_tables()
    qkv = rnd(B * Tq, 3 * Dm, seed=11)            # self-attention layout when Tq == Tk; otherwise separate kv
    kv = rnd(B * Tk, 2 * Dm, seed=12)
    do = rnd(B * Tq, Dm, seed=13)
    q = qkv[:, :Dm].double().requires_grad_()
    k = kv[:, :Dm].double().requires_grad_()
    v = kv[:, Dm:].double().requires_grad_()
    split = lambda t, T: t.view(B, T, H, hd).transpose(1, 2)
    att = _rot(split(q, Tq), rope) @ _rot(split(k, Tk), rope).transpose(-1, -2) / hd ** 0.5
    if causal:
        att = att.masked_fill(~torch.ones(Tq, Tk, dtype=torch.bool).tril(), float("-inf"))
    y = (att.softmax(-1) @ split(v, Tk)).transpose(1, 2).reshape(B * Tq, Dm)
    y.backward(do.double())
    qd, kvd, dod = dev(qkv), dev(kv), dev(do)
    dq = torch.zeros(B * Tq, 3 * Dm, device="cuda")
    dkv0 = rnd(B * Tk, 2 * Dm, seed=14)
    dkv = dev(dkv0).clone()
    a = lib.AttnBwdArgs(q=qd.data_ptr(), ldq=3 * Dm, k=kvd.data_ptr(), v=kvd.data_ptr() + 4 * Dm, ldkv=2 * Dm,
                        d_out=dod.data_ptr(), ld_do=Dm, dq=dq.data_ptr(), ld_dq=3 * Dm, dk=dkv.data_ptr(),
                        dv=dkv.data_ptr() + 4 * Dm, ld_dkv=2 * Dm, accumulate_kv=1, B=B, H=H, hd=hd, Tq=Tq, Tk=Tk,
                        causal=causal, rope=rope, rope_cos=rc.data_ptr(), rope_sin=rs.data_ptr())
    lib.check(lib.load().mdt_op_attn_bwd(C.byref(a), stream()))
    assert_close(dq[:, :Dm].cpu(), q.grad, what="dq", **G_TOL)
    assert_close(dkv[:, :Dm].cpu() - dkv0[:, :Dm], k.grad, what="dk", **G_TOL)
    assert_close(dkv[:, Dm:].cpu() - dkv0[:, Dm:], v.grad, what="dv", **G_TOL)


@pytest.mark.parametrize("act,fn", [(1, F.gelu), (2, F.mish), (3, F.silu)])
def test_activation_forward_backward(lib, act, fn):
    u = torch.cat([rnd(5000, seed=21) * 3, torch.tensor([-30.0, -8.0, -1e-3, 0.0, 1e-3, 8.0, 25.0, 30.0])])
    dy = rnd(u.numel(), seed=22)
    u64 = u.double().requires_grad_()
    y = fn(u64)
    y.backward(dy.double())
    ud, dyd = dev(u), dev(dy)
    out, du = torch.empty_like(ud), torch.empty_like(ud)
    L = lib.load()
    lib.check(L.mdt_op_act_fwd(ud.data_ptr(), out.data_ptr(), u.numel(), act, stream()))
    lib.check(L.mdt_op_act_bwd(ud.data_ptr(), dyd.data_ptr(), du.data_ptr(), u.numel(), act, stream()))
    assert_close(out.cpu(), y.detach(), rtol=1e-4, atol=2e-6, what="act")
    assert_close(du.cpu(), u64.grad, rtol=1e-4, atol=5e-6, what="act grad")


def merge(lib, fn, x, a, gate, out, dgate, B, rps, D, p=0.0, site=0, seed=0, gate_col=0, ld=0):
    g = lib.MergeArgs(x=x.data_ptr(), a=a.data_ptr(), gate=(gate.data_ptr() + 4 * gate_col) if gate is not None else None,
                      gate_stride=ld, out=out.data_ptr(), dgate=(dgate.data_ptr() + 4 * gate_col) if dgate is not None else None,
                      dgate_stride=ld, B=B, rows_per_sample=rps, D=D, p=p, site=site, seed=seed)
    lib.check(getattr(lib.load(), fn)(C.byref(g), stream()))


def test_merge_gate_backward_and_colsum(lib):
    B, rps, D = 6, 10, 384
    x, a, mod = rnd(B * rps, D, seed=31), rnd(B * rps, D, seed=32), rnd(B, 6 * D, seed=33)
    xd, ad, md = dev(x), dev(a), dev(mod)
    out = torch.empty(B * rps, D, device="cuda")
    merge(lib, "mdt_op_merge_fwd", xd, ad, md, out, None, B, rps, D, gate_col=2 * D, ld=6 * D)
    gate = mod[:, 2 * D:3 * D].repeat_interleave(rps, 0)
    assert_close(out.cpu(), x + gate * a, what="x + g a")
    merge(lib, "mdt_op_merge_fwd", xd, ad, None, out, None, B, rps, D)
    assert_close(out.cpu(), x + a, what="x + a")
    dx = rnd(B * rps, D, seed=34)
    da = torch.empty(B * rps, D, device="cuda")
    dmod = torch.zeros(B, 6 * D, device="cuda")
    merge(lib, "mdt_op_merge_bwd", dev(dx), ad, md, da, dmod, B, rps, D, gate_col=2 * D, ld=6 * D)
    assert_close(da.cpu(), dx * gate, what="d_a", **G_TOL)
    assert_close(dmod[:, 2 * D:3 * D].cpu(), (dx * a).view(B, rps, D).sum(1).double(), what="d_gate", **G_TOL)
    assert dmod[:, :2 * D].abs().max() == 0 and dmod[:, 3 * D:].abs().max() == 0
    L = lib.load()
    for M, N in [(2560, 1536), (37, 7), (1, 100), (130, 64), (10240, 7), (2048, 384), (5000, 4096), (2049, 4100)]:
        X = rnd(M, N + 3, seed=M)
        out0 = rnd(N, seed=N)
        o = dev(out0).clone()
        lib.check(L.mdt_op_colsum(dev(X).data_ptr(), N + 3, M, N, o.data_ptr(), 1, stream()))
        assert_close(o.cpu() - out0, X[:, :N].double().sum(0), rtol=1e-4, atol=1e-4, what=f"colsum {M}x{N}")
    # few rows, very many columns (the slices of a weight gradient): the 16-byte kernel; odd strides keep the general one
    for M, N, pad in [(16, 589824, 0), (5, 20000, 4), (9, 16384, 0), (1, 65536, 8), (256, 16388, 0), (13, 16384, 3)]:
        X = rnd(M, N + pad, seed=M + N)
        out0 = rnd(N, seed=N + 1)
        for acc in (1, 0):
            o = dev(out0).clone()
            lib.check(L.mdt_op_colsum(dev(X).data_ptr(), N + pad, M, N, o.data_ptr(), acc, stream()))
            assert_close(o.cpu() - (out0 if acc else 0), X[:, :N].double().sum(0), rtol=1e-4, atol=1e-4, what=f"wide colsum {M}x{N}")


@pytest.mark.parametrize("B,rps,D", [(3, 10, 384), (2, 401, 384), (5, 7, 64), (2, 3, 1024), (4, 1, 192), (3, 5, 30), (2, 19, 96)])
def test_merge_vector_kernels_equal_the_scalar_ones(lib, B, rps, D):
    """The 16-byte kernels (every load of a thread in flight before its first store) against the one-column-per-thread kernels
    they replaced: the latter still take whatever is not 16-byte aligned, so a copy of the operands shifted by one float selects
    them.  Same mask, same arithmetic per element: forward and d_a bit for bit, d_gate to rounding (other summation order)."""
    p, seed, site = 0.1, 987654321, 3
    x, a, gate, dx = rnd(B * rps, D, seed=41), rnd(B * rps, D, seed=42), rnd(B, D, seed=43), rnd(B * rps, D, seed=44)

    def shifted(t):   # the same values at an address that is 4 (mod 16)
        buf = torch.empty(t.numel() + 4, device="cuda")
        v = buf[1:1 + t.numel()].view(t.shape)
        v.copy_(t)
        return v

    res = []
    for mk in (dev, lambda t: shifted(dev(t))):
        xd, ad, gd, dxd = mk(x), mk(a), mk(gate), mk(dx)
        out, da, dg = mk(torch.zeros_like(x)), mk(torch.zeros_like(x)), mk(torch.zeros_like(gate))
        merge(lib, "mdt_op_merge_fwd", xd, ad, gd, out, None, B, rps, D, p=p, site=site, seed=seed, ld=D)
        merge(lib, "mdt_op_merge_bwd", dxd, ad, gd, da, dg, B, rps, D, p=p, site=site, seed=seed, ld=D)
        res.append((out.clone(), da.clone(), dg.clone()))
    (o1, a1, g1), (o2, a2, g2) = res
    assert torch.equal(o1, o2) and torch.equal(a1, a2)
    # (a sum of rps products of unit-variance values, added in two different orders: fp32 rounding grows with sqrt(rps) -- 2e-5
    #  on entries near zero at rps = 401)
    assert_close(g1.cpu(), g2.cpu().double(), rtol=1e-5, atol=1e-5 * max(1.0, (rps / 10) ** 0.5) * 2, what="d_gate")
    keep = (o1 != dev(x)).float().mean().item()      # x + g * drop(a) differs from x exactly where the mask kept a
    assert abs(keep - (1 - p)) < 0.05
    mask = ((o1 - dev(x)) != 0).float() / (1 - p)
    assert_close(a1.cpu(), (dev(dx) * dev(gate).repeat_interleave(rps, 0) * mask).cpu().double(), what="d_a", **G_TOL)


def test_merge_dropout_mask_statistics_and_backward(lib):
    B, rps, D, p, seed = 64, 10, 384, 0.1, 123456789
    zeros, ones = torch.zeros(B * rps, D, device="cuda"), torch.ones(B * rps, D, device="cuda")
    m1, m1b, m2, m3 = (torch.empty(B * rps, D, device="cuda") for _ in range(4))
    merge(lib, "mdt_op_merge_fwd", zeros, ones, None, m1, None, B, rps, D, p=p, site=5, seed=seed)
    merge(lib, "mdt_op_merge_fwd", zeros, ones, None, m1b, None, B, rps, D, p=p, site=5, seed=seed)
    merge(lib, "mdt_op_merge_fwd", zeros, ones, None, m2, None, B, rps, D, p=p, site=6, seed=seed)
    merge(lib, "mdt_op_merge_fwd", zeros, ones, None, m3, None, B, rps, D, p=p, site=5, seed=seed + 1)
    assert torch.equal(m1, m1b)                                   # counter based: reproducible
    vals = torch.unique(m1).cpu()
    assert_close(vals, torch.tensor([0.0, 1 / (1 - p)]), rtol=1e-6, atol=0, what="mask values")
    n = m1.numel()
    keep = (m1 > 0).float().mean().item()
    assert abs(keep - (1 - p)) < 5 * (p * (1 - p) / n) ** 0.5      # binomial, 5 sigma
    for other in (m2, m3):                                         # other site / other seed: independent masks
        agree = ((m1 > 0) == (other > 0)).float().mean().item()
        assert abs(agree - ((1 - p) ** 2 + p ** 2)) < 0.01
    # columns / rows are not correlated in an obvious way
    assert abs((m1 > 0).float().mean(0).std().item() - (p * (1 - p) / (B * rps)) ** 0.5) < 0.01
    # forward / backward with that mask against autograd
    x, a, mod, dx = rnd(B * rps, D, seed=35), rnd(B * rps, D, seed=36), rnd(B, 6 * D, seed=37), rnd(B * rps, D, seed=38)
    out, da, dmod = torch.empty_like(m1), torch.empty_like(m1), torch.zeros(B, 6 * D, device="cuda")
    merge(lib, "mdt_op_merge_fwd", dev(x), dev(a), dev(mod), out, None, B, rps, D, p=p, site=5, seed=seed, gate_col=5 * D, ld=6 * D)
    merge(lib, "mdt_op_merge_bwd", dev(dx), dev(a), dev(mod), da, dmod, B, rps, D, p=p, site=5, seed=seed, gate_col=5 * D, ld=6 * D)
    mask = m1.cpu().double()
    a64, g64 = a.double().requires_grad_(), mod[:, 5 * D:].double().requires_grad_()
    y = x.double() + g64.repeat_interleave(rps, 0) * (a64 * mask)
    y.backward(dx.double())
    assert_close(out.cpu(), y.detach(), what="merge with dropout")
    assert_close(da.cpu(), a64.grad, what="d_a with dropout", **G_TOL)
    assert_close(dmod[:, 5 * D:].cpu(), g64.grad, what="d_gate with dropout", **G_TOL)


@pytest.mark.parametrize("B,rps,D,gated,mod,bias,p", [(5, 10, 384, True, True, False, 0.1), (3, 4, 384, False, False, False, 0.1),
                                                        (4, 10, 384, False, True, True, 0.05), (2, 10, 128, True, True, True, 0.0),
                                                        (3, 7, 30, True, False, False, 0.1)])
def test_fused_pairs_of_the_training_step_equal_their_two_launches(lib, B, rps, D, gated, mod, bias, p):
    """Round 6: every branch merge of the training forward shares a launch with the LayerNorm that reads its result
    (mdt_op_merge_ln_fwd), every LayerNorm backward with the merge backward behind it (mdt_op_ln_bwd_merge).  Same arithmetic
    per element as the two launches each replaces: bit for bit, except the gate gradient (its sum over a sample's rows is
    formed wave by wave instead of row-lane by row-lane).  D = 30 takes the two-launch route inside the fused entry points."""
    L = lib.load()
    M, seed, site = B * rps, 424242, 9
    x, a, w = dev(rnd(M, D, seed=51)), dev(rnd(M, D, seed=52)), dev(1 + 0.1 * rnd(D, seed=53))
    b = dev(0.1 * rnd(D, seed=54)) if bias else None
    modt = dev(rnd(B, 6 * D, seed=55))
    sh, sc, gc = 3 * D, 4 * D, 5 * D

    def margs(xin, out, dgate=None):
        return lib.MergeArgs(x=xin.data_ptr(), a=a.data_ptr(), gate=(modt.data_ptr() + 4 * gc) if gated else None, gate_stride=6 * D,
                             out=out.data_ptr(), dgate=(dgate.data_ptr() + 4 * gc) if (gated and dgate is not None) else None,
                             dgate_stride=6 * D, B=B, rows_per_sample=rps, D=D, p=p, site=site, seed=seed)

    def largs(xin, out, stats):
        return lib.LnTrainArgs(x=xin.data_ptr(), w=w.data_ptr(), b=b.data_ptr() if bias else None, mod=modt.data_ptr() if mod else None,
                               mod_stride=6 * D, shift_off=sh if mod else -1, scale_off=sc if mod else -1, rows_per_sample=rps,
                               out=out.data_ptr(), stats=stats.data_ptr(), M=M, D=D)

    # forward: merge -> LayerNorm
    x1a, ha, sta = torch.empty(M, D, device="cuda"), torch.empty(M, D, device="cuda"), torch.empty(M, 2, device="cuda")
    x1b, hb, stb = torch.empty_like(x1a), torch.empty_like(ha), torch.empty_like(sta)
    ga = margs(x, x1a)
    lib.check(L.mdt_op_merge_fwd(C.byref(ga), stream()))
    la = largs(x1a, ha, sta)
    lib.check(L.mdt_op_ln_fwd_train(C.byref(la), stream()))
    gb, lb = margs(x, x1b), largs(x, hb, stb)   # l.x is ignored by the fused call
    lib.check(L.mdt_op_merge_ln_fwd(C.byref(gb), C.byref(lb), stream()))
    assert torch.equal(x1a, x1b) and torch.equal(ha, hb) and torch.equal(sta, stb)
    # backward: LayerNorm backward (accumulating) -> merge backward on the gradient it leaves
    dh, dx0 = dev(rnd(M, D, seed=56)), dev(rnd(M, D, seed=57))
    res = []
    for fused in (False, True):
        dx = dx0.clone()
        dmod, da = torch.zeros(B, 6 * D, device="cuda"), torch.empty(M, D, device="cuda")
        pw, pb = torch.empty(B, D, device="cuda"), torch.empty(B, D, device="cuda")
        g = lib.LnBwdArgs(x=x1a.data_ptr(), stats=sta.data_ptr(), w=w.data_ptr(), b=b.data_ptr() if bias else None,
                          mod=modt.data_ptr() if mod else None, mod_stride=6 * D, shift_off=sh if mod else -1,
                          scale_off=sc if mod else -1, dh=dh.data_ptr(), ld_dh=D, dx=dx.data_ptr(), accumulate=1,
                          d_mod=dmod.data_ptr() if mod else None, d_mod_stride=6 * D, pw=pw.data_ptr(), pb=pb.data_ptr() if bias else None,
                          B=B, rows_per_sample=rps, D=D, row_chunks=0)
        mg = margs(dx, da, dmod)
        if fused:
            lib.check(L.mdt_op_ln_bwd_merge(C.byref(g), C.byref(mg), stream()))
        else:
            lib.check(L.mdt_op_ln_bwd(C.byref(g), stream()))
            lib.check(L.mdt_op_merge_bwd(C.byref(mg), stream()))
        res.append((dx, da, dmod, pw, pb if bias else None))
    (dx_a, da_a, dm_a, pw_a, pb_a), (dx_b, da_b, dm_b, pw_b, pb_b) = res
    assert torch.equal(dx_a, dx_b) and torch.equal(da_a, da_b) and torch.equal(pw_a, pw_b)
    if bias:
        assert torch.equal(pb_a, pb_b)
    if mod:
        assert torch.equal(dm_a[:, sh:gc], dm_b[:, sh:gc])          # d_shift, d_scale
    if gated:
        assert_close(dm_b[:, gc:].cpu(), dm_a[:, gc:].cpu().double(), rtol=1e-5, atol=1e-5, what="d_gate of the fused launch")
        assert dm_b[:, gc:].abs().max() > 0


@pytest.mark.parametrize("hd,H,Tq,Tk,causal,rope", [(48, 8, 10, 10, 1, 0), (48, 8, 10, 4, 1, 0), (16, 4, 4, 4, 0, 0),
                                                   (48, 8, 10, 10, 1, 1), (32, 4, 10, 4, 1, 1), (64, 2, 5, 16, 0, 1)])
def test_attention_dropout_forward_and_backward(lib, hd, H, Tq, Tk, causal, rope):
    B, Dm, p, seed, site = 5, H * hd, 0.3, 987654321, 11
    L = lib.load()
    rc, rs = _rope_tables()
    q, kv, do = rnd(B * Tq, Dm, seed=51), rnd(B * Tk, 2 * Dm, seed=52), rnd(B * Tq, Dm, seed=53)
    qd, dod = dev(q), dev(do)

    def fwd(kvt, pp):
        out = torch.empty(B * Tq, Dm, device="cuda")
        a = lib.AttnTrainArgs(q=qd.data_ptr(), ldq=Dm, k=kvt.data_ptr(), v=kvt.data_ptr() + 4 * Dm, ldkv=2 * Dm,
                              out=out.data_ptr(), ldo=Dm, B=B, H=H, hd=hd, Tq=Tq, Tk=Tk, causal=causal, p=pp, site=site,
                              seed=seed, rope=rope, rope_cos=rc.data_ptr(), rope_sin=rs.data_ptr())
        lib.check(L.mdt_op_attn_fwd_train(C.byref(a), stream()))
        return out.cpu()

    # probe the probabilities with one-hot values: out[i, h*hd + j] = P_used[i, j]   (Tk <= hd)
    probe = kv.clone()
    probe[:, Dm:] = torch.eye(Tk, hd).repeat(B, H)
    pick = lambda o: o.view(B, Tq, H, hd)[..., :Tk].permute(0, 2, 1, 3)        # (B, H, Tq, Tk)
    P0, Pd = pick(fwd(dev(probe), 0.0)), pick(fwd(dev(probe), p))
    visible = P0 > 0
    mask = torch.where(visible, Pd / P0.clamp_min(1e-30), torch.zeros_like(P0))
    vals = torch.unique(mask[visible].round(decimals=4))
    assert_close(vals, torch.tensor([0.0, round(1 / (1 - p), 4)]), rtol=1e-3, atol=1e-4, what="mask values")
    keep = (mask[visible] > 0).float().mean().item()
    nvis = int(visible.sum())
    assert abs(keep - (1 - p)) < 5 * (p * (1 - p) / nvis) ** 0.5
    # forward / backward with random values against autograd using that mask
    split = lambda t, T: t.view(B, T, H, hd).transpose(1, 2)
    q64 = q.double().requires_grad_()
    k64, v64 = kv[:, :Dm].double().requires_grad_(), kv[:, Dm:].double().requires_grad_()
    att = _rot(split(q64, Tq), rope) @ _rot(split(k64, Tk), rope).transpose(-1, -2) / hd ** 0.5
    if causal:
        att = att.masked_fill(~torch.ones(Tq, Tk, dtype=torch.bool).tril(), float("-inf"))
    y = ((att.softmax(-1) * mask.double()) @ split(v64, Tk)).transpose(1, 2).reshape(B * Tq, Dm)
    y.backward(do.double())
    assert_close(fwd(dev(kv), p), y.detach(), what="attention with dropout")
    kvd = dev(kv)
    dq, dkv = torch.zeros(B * Tq, Dm, device="cuda"), torch.zeros(B * Tk, 2 * Dm, device="cuda")
    g = lib.AttnBwdArgs(q=qd.data_ptr(), ldq=Dm, k=kvd.data_ptr(), v=kvd.data_ptr() + 4 * Dm, ldkv=2 * Dm,
                        d_out=dod.data_ptr(), ld_do=Dm, dq=dq.data_ptr(), ld_dq=Dm, dk=dkv.data_ptr(),
                        dv=dkv.data_ptr() + 4 * Dm, ld_dkv=2 * Dm, accumulate_kv=0, B=B, H=H, hd=hd, Tq=Tq, Tk=Tk,
                        causal=causal, p=p, site=site, seed=seed, rope=rope, rope_cos=rc.data_ptr(),
                        rope_sin=rs.data_ptr())
    lib.check(L.mdt_op_attn_bwd(C.byref(g), stream()))
    assert_close(dq.cpu(), q64.grad, what="dq", **G_TOL)
    assert_close(dkv[:, :Dm].cpu(), k64.grad, what="dk", **G_TOL)
    assert_close(dkv[:, Dm:].cpu(), v64.grad, what="dv", **G_TOL)


@pytest.mark.parametrize("M,N,K", [(8200, 384, 384), (8192, 1152, 384), (10240, 640, 128), (33000, 192, 768), (32768, 384, 384),
                                   (40000, 1536, 192), (33000, 576, 192), (32800, 256, 64)])
def test_weight_gradient_wide_tiles(lib, M, N, K):
    """The 128- and 192-column tiles of the weight-gradient product (8 / 12 waves per workgroup; from 8192 / 32768 reduction rows
    on, mdt_gemm_tn_tile) against float64 on the GPU, bias gradient included, ragged last slices."""
    L = lib.load()
    g = torch.Generator(device="cuda").manual_seed(M + N)
    X = torch.randn(M, K, device="cuda", generator=g)
    dY = torch.randn(M, N, device="cuda", generator=g)
    dW0 = torch.randn(N, K, device="cuda", generator=g)
    dW, db = dW0.clone(), torch.zeros(N, device="cuda")
    scratch = torch.empty(L.mdt_op_linear_bwd_scratch(M, N, K), device="cuda")
    a = lib.LinearBwdArgs(X=X.data_ptr(), ldx=K, dY=dY.data_ptr(), ldy=N, Wt=None, dW=dW.data_ptr(), dbias=db.data_ptr(), dX=None,
                          ldxo=K, accumulate_dw=1, accumulate_dx=0, M=M, N=N, K=K, scratch=scratch.data_ptr())
    lib.check(L.mdt_op_linear_bwd(C.byref(a), stream()))
    tol = dict(rtol=1e-3, atol=1e-4 * (M / 64) ** 0.5)
    assert_close((dW - dW0).cpu(), (dY.double().T @ X.double()).cpu(), what="dW", **tol)
    assert_close(db.cpu(), dY.double().sum(0).cpu(), what="dbias", **tol)


@pytest.mark.parametrize("M,N,K", [(8200, 384, 384), (12288, 1536, 384), (12288, 384, 1536), (33000, 768, 208), (40000, 1536, 192),
                                   (33000, 576, 192), (8192 + 31, 240, 80), (10000, 128, 128), (33000, 192, 768), (20000, 192, 1536),
                                   (9000, 192, 576), (9000, 192, 192), (8300, 384, 208)])
def test_weight_gradient_as_bf16_splits_keeps_fp32_accuracy(lib, M, N, K):
    """Round 6: from 8192 reduction rows on dW = dY^T X runs as three-way bf16 splits of both operands (k_gemm_tn_split: six bf16
    MFMA products per 32 rows, operands transposed into m-contiguous octets on their way into LDS).  Against float64: the error
    of dW and of the bias gradient within 2.5x the fp32 MFMA kernel's on the same inputs (rows of very different scale), ragged
    slices and tiles (N, K not multiples of the 128 x 128 / 128 x 192 / 192 x 128 tile), accumulation into dW."""
    L = lib.load()
    g = torch.Generator(device="cuda").manual_seed(M + N + K)
    X = torch.randn(M, K, device="cuda", generator=g) * torch.exp(1.5 * torch.randn(M, 1, device="cuda", generator=g))
    dY = torch.randn(M, N, device="cuda", generator=g)
    dW0 = torch.randn(N, K, device="cuda", generator=g)
    ref_W, ref_b = dY.double().T @ X.double(), dY.double().sum(0)
    err = {}
    try:
        for split in (0, 1):
            L.mdt_op_set_tn_split(split)
            dW, db = dW0.clone(), torch.zeros(N, device="cuda")
            scratch = torch.empty(L.mdt_op_linear_bwd_scratch(M, N, K), device="cuda")
            a = lib.LinearBwdArgs(X=X.data_ptr(), ldx=K, dY=dY.data_ptr(), ldy=N, Wt=None, dW=dW.data_ptr(), dbias=db.data_ptr(), dX=None,
                                  ldxo=K, accumulate_dw=1, accumulate_dx=0, M=M, N=N, K=K, scratch=scratch.data_ptr())
            lib.check(L.mdt_op_linear_bwd(C.byref(a), stream()))
            torch.cuda.synchronize()
            err[split] = (((dW - dW0).double() - ref_W).abs().max().item(), (db.double() - ref_b).abs().max().item(), (dW - dW0).clone())
            tol = dict(rtol=1e-3, atol=1e-4 * (M / 64) ** 0.5 * float(X.abs().max()))
            assert_close((dW - dW0).cpu(), ref_W.cpu(), what=f"dW (split {split})", **tol)
    finally:
        L.mdt_op_set_tn_split(-1)
    assert not torch.equal(err[0][2], err[1][2]), "the split kernel did not run (same bits as the fp32 kernel)"
    # (2.5x, not the 1.5x of the other split forms: the split kernel also reduces over fewer, deeper row slices -- one round of its
    #  one-per-CU workgroups -- and a deeper fp32 accumulation chain has the larger rounding error whatever the products are)
    assert err[1][0] <= 2.5 * err[0][0] + 1e-6, f"dW: split error {err[1][0]:.3g} against the fp32 kernel's {err[0][0]:.3g}"
    assert err[1][1] <= 2.5 * err[0][1] + 1e-4, f"dbias: split path {err[1][1]:.3g} against the fp32 kernel's {err[0][1]:.3g}"


@pytest.mark.parametrize("M,N,K", [(2560, 1536, 384), (1280, 384, 1536), (1024, 1152, 384), (250, 384, 384), (37, 768, 512),
                                   (3, 64, 128), (128, 9216, 384)])
def test_linear_backward_through_the_forward_gemm(lib, M, N, K):
    X, W, dY = rnd(M, K, seed=41), rnd(N, K, seed=42) / K ** 0.5, rnd(M, N, seed=43)
    L = lib.load()
    Xd, dYd = dev(X), dev(dY)
    Wt = torch.zeros(N * K, device="cuda")
    lib.check(L.mdt_op_pack_weight_t(dev(W).data_ptr(), N, K, K, Wt.data_ptr(), 0, N, stream()))
    Mp = (M + 15) // 16 * 16
    scratch = torch.empty(L.mdt_op_linear_bwd_scratch(M, N, K), device="cuda")
    dW0, dX0 = rnd(N, K, seed=44), rnd(M, K, seed=45)
    dW, dX = dev(dW0).clone(), dev(dX0).clone()
    db = torch.zeros(N, device="cuda")
    a = lib.LinearBwdArgs(X=Xd.data_ptr(), ldx=K, dY=dYd.data_ptr(), ldy=N, Wt=Wt.data_ptr(), dW=dW.data_ptr(),
                          dbias=db.data_ptr(), dX=dX.data_ptr(), ldxo=K, accumulate_dw=1, accumulate_dx=1, M=M, N=N, K=K,
                          scratch=scratch.data_ptr())
    lib.check(L.mdt_op_linear_bwd(C.byref(a), stream()))
    ref_dW = dY.double().T @ X.double()
    ref_dX = dY.double() @ W.double()
    tol = dict(rtol=1e-3, atol=1e-4 * max(1.0, (M / 64) ** 0.5))
    assert_close(dW.cpu() - dW0, ref_dW, what="dW", **tol)
    assert_close(dX.cpu() - dX0, ref_dX, what="dX", rtol=1e-3, atol=1e-4)
    assert_close(db.cpu(), dY.double().sum(0), what="dbias", rtol=1e-3, atol=1e-4 * max(1.0, (M / 64) ** 0.5))
