"""Masked generative foresight head (reference
mdt/models/img_generation/masked_transformer_decoder.py).  CPU: the oracle against the golden outputs AND gradients of the
reference's own forward / compute_loss (tests/golden/g15_mae_*.npz; the Voltron blocks inside are stood in for by the
oracle's restatement -- block internals parity-unpinned, see oracle/mae_oracle.py), the facade's parameter tree.
GPU: the HIP ops against float64 PyTorch, the facade against the goldens and against float64 autograd through the oracle."""
import ctypes as C

import os

import numpy as np
import pytest
import torch

from mdt_policy_amd import synthetic
from oracle import mae_oracle as O
from tests.helpers import assert_close, load_fixture

CASES = ["default", "tiny", "tiny_asym"]  # tiny_asym: symmetric_mask = False, the reference's other masking branch as written


def case(name):
    meta, fx = load_fixture(f"g15_mae_{name}.npz")
    kw = meta["kwargs"]
    shapes = [(k, tuple(s)) for k, s in meta["state_dict"] if k != "decoder_pe"]
    P = {k: torch.from_numpy(v) for k, v in synthetic.fill_state_dict(shapes, meta["weight_seed"], meta["profile"]).items()}
    P["decoder_pe"] = torch.from_numpy(fx["decoder_pe"])
    ctx = torch.from_numpy(synthetic.normal("ctx", (meta["B"], meta["Tc"], kw["context_dim"]), meta["ctx_seed"]))
    img = torch.from_numpy(synthetic.normal("img", (meta["B"], 2, 3, kw["resolution"], kw["resolution"]), meta["img_seed"]))
    shuffle = torch.argsort(torch.from_numpy(fx["restore"]), dim=1)
    return meta, fx, kw, P, ctx, img, shuffle


def summary(g):
    g = g.detach().double().cpu()
    return [float(g.norm()), float(g.sum())] + [float(v) for v in g.flatten()[:6]]


def check_summaries(got, want, what, rtol=2e-3):
    assert set(want) == set(got), (what, set(want) ^ set(got))
    for k, w in want.items():
        g, w = np.array(got[k]), np.array(w)
        tol = rtol * abs(w[0]) + 1e-6
        assert abs(g[0] - w[0]) <= tol, f"{what} {k}: norm {g[0]} vs {w[0]}"
        assert np.all(np.abs(g[1:] - w[1:]) <= tol), f"{what} {k}: {g[1:]} vs {w[1:]}"


@pytest.mark.parametrize("name", CASES)
def test_oracle_matches_the_reference_forward_loss_and_gradients(name):
    meta, fx, kw, P, ctx, img, shuffle = case(name)
    np.testing.assert_allclose(O.position_table(kw["decoder_embed_dim"], kw["resolution"] // kw["patch_size"]), fx["decoder_pe"][0], atol=1e-6)
    P64 = {k: v.double().requires_grad_(k != "decoder_pe") for k, v in P.items()}
    c64 = ctx.double().requires_grad_()
    rec, mask, restore, vis = O.forward(P64, kw, c64, img.double(), shuffle)
    loss = O.compute_loss(kw, img.double(), rec, mask)
    loss.backward()
    assert_close(rec.detach(), fx["rec"], what="reconstructions")
    assert np.array_equal(mask.numpy(), fx["mask"]) and np.array_equal(restore.numpy(), fx["restore"])
    assert_close(vis.detach(), fx["visible"], what="visible patches")
    assert abs(loss.item() - float(fx["loss"].reshape(-1)[0])) <= 1e-5 * abs(loss.item())
    assert_close(c64.grad, fx["d_ctx"], rtol=2e-3, atol=1e-8, what="d_context")
    check_summaries({k: summary(v.grad) for k, v in P64.items() if v.grad is not None}, meta["grads"], name)


@pytest.mark.parametrize("name", CASES)
def test_facade_parameter_tree_matches_the_reference(name):
    from mdt_policy_amd.models.img_generation.masked_transformer_decoder import MaskedTransformerImgDecoder
    meta, fx, kw, P, *_ = case(name)
    m = MaskedTransformerImgDecoder(**kw)
    assert [[k, list(v.shape)] for k, v in m.state_dict().items()] == meta["state_dict"]
    assert [k for k, _ in m.named_parameters()] == meta["named_parameters"]
    assert_close(m.decoder_pe.detach(), fx["decoder_pe"], atol=1e-6, what="decoder_pe (2-D sine-cosine table)")
    assert not m.decoder_pe.requires_grad
    assert torch.allclose(m.decoder_blocks[0].layer_scale_attn.gamma, torch.full((kw["decoder_embed_dim"],), 0.1))
    with pytest.raises(RuntimeError, match="ROCm GPU"):
        m(torch.zeros(1, 4, kw["context_dim"]), torch.zeros(1, 2, 3, kw["resolution"], kw["resolution"]))


# ---------------------------------------------------------------------------------------------------------------- GPU
def _lib():
    from mdt_policy_amd import _lib as L
    return L, L.load()


def _s():
    return torch.cuda.current_stream().cuda_stream


@pytest.mark.gpu
@pytest.mark.parametrize("B,H,hd,T", [(3, 8, 24, 102), (2, 4, 16, 36), (1, 2, 64, 64), (2, 2, 32, 100), (2, 3, 32, 17), (5, 8, 48, 7), (2, 2, 24, 1), (2, 2, 16, 128), (1, 3, 24, 113)])
def test_attention_mid_forward_and_backward(B, H, hd, T):
    L, lib = _lib()
    D = H * hd
    qkv = torch.from_numpy(synthetic.normal("qkv", (B, T, 3 * D), 161))
    do = torch.from_numpy(synthetic.normal("do", (B, T, D), 162))
    q64 = qkv.double().requires_grad_()
    q, k, v = (t.reshape(B, T, H, hd).transpose(1, 2) for t in q64.split(D, dim=-1))
    ref = ((q @ k.transpose(-1, -2) * hd ** -0.5).softmax(-1) @ v).transpose(1, 2).reshape(B, T, D)
    ref.backward(do.double())
    qd, dod = qkv.cuda(), do.cuda()
    out = torch.empty(B, T, D, device="cuda")
    L.check(lib.mdt_op_attn_mid_fwd(qd.data_ptr(), 3 * D, out.data_ptr(), D, B, H, hd, T, hd ** -0.5, _s()))
    assert_close(out.cpu(), ref.detach(), what="attention output")
    dq = torch.full((B, T, 3 * D), float("nan"), device="cuda")
    L.check(lib.mdt_op_attn_mid_bwd(qd.data_ptr(), 3 * D, out.data_ptr(), D, dod.data_ptr(), D, dq.data_ptr(), 3 * D, B, H, hd, T,
                                    hd ** -0.5, _s()))
    assert_close(dq.cpu(), q64.grad, rtol=1e-3, atol=1e-4, what="d_qkv")


@pytest.mark.gpu
@pytest.mark.parametrize("B,H,hd,T", [(256, 8, 24, 102), (1024, 8, 24, 102), (600, 6, 24, 70), (1100, 4, 32, 90), (520, 3, 16, 51), (64, 8, 24, 102)])
def test_attention_mid_forward_head_walk_regimes(B, H, hd, T):
    """Batch sizes at which one workgroup of the forward walks several heads of a sample (all of them from B * H / hpw >= 512
    workgroups on: hpw = 4, 8, 3, 2, 3, 1 here), every tile count 4 .. 7, against float64 attention on the GPU."""
    L, lib = _lib()
    D = H * hd
    g = torch.Generator(device="cuda").manual_seed(1234 + B)
    qkv = torch.randn(B, T, 3 * D, device="cuda", generator=g) * 1.5
    q, k, v = (t.reshape(B, T, H, hd).transpose(1, 2) for t in qkv.double().split(D, dim=-1))
    ref = ((q @ k.transpose(-1, -2) * hd ** -0.5).softmax(-1) @ v).transpose(1, 2).reshape(B, T, D)
    out = torch.full((B, T, D), float("nan"), device="cuda")
    L.check(lib.mdt_op_attn_mid_fwd(qkv.data_ptr(), 3 * D, out.data_ptr(), D, B, H, hd, T, hd ** -0.5, _s()))
    assert_close(out.cpu(), ref.cpu(), what="attention output")
    # strided views (the model passes the qkv projection's output and writes into a wider buffer)
    wide_in = torch.randn(B, T, 3 * D + 8, device="cuda", generator=g)
    wide_in[..., :3 * D] = qkv
    wide_out = torch.full((B, T, D + 4), float("nan"), device="cuda")
    L.check(lib.mdt_op_attn_mid_fwd(wide_in.data_ptr(), 3 * D + 8, wide_out.data_ptr(), D + 4, B, H, hd, T, hd ** -0.5, _s()))
    assert torch.equal(wide_out[..., :D], out)
    assert torch.isnan(wide_out[..., D:]).all()


@pytest.mark.gpu
def test_attention_mid_rejects_what_it_cannot_run():
    L, lib = _lib()
    x = torch.zeros(1, 129, 3 * 48, device="cuda")
    o = torch.zeros(1, 129, 48, device="cuda")
    assert lib.mdt_op_attn_mid_fwd(x.data_ptr(), 144, o.data_ptr(), 48, 1, 2, 24, 129, 1.0, _s()) != 0   # T > 128
    assert lib.mdt_op_attn_mid_fwd(x.data_ptr(), 144, o.data_ptr(), 48, 1, 4, 12, 64, 1.0, _s()) != 0    # head dim 12


@pytest.mark.gpu
@pytest.mark.parametrize("M,D", [(306, 192), (7, 64), (1030, 384)])
def test_rmsnorm_and_swiglu_ops(M, D):
    L, lib = _lib()
    x = torch.from_numpy(synthetic.normal("x", (M, D), 163))
    g = torch.from_numpy(synthetic.normal("g", (D,), 164, std=0.1, mean=1.0))
    dy = torch.from_numpy(synthetic.normal("dy", (M, D), 165))
    x64, g64 = x.double().requires_grad_(), g.double().requires_grad_()
    ref = O.rms_norm(x64, g64)
    ref.backward(dy.double())
    xd, gd, dyd = x.cuda(), g.cuda(), dy.cuda()
    out, dx, dg = torch.empty_like(xd), torch.empty_like(xd), torch.empty_like(gd)
    L.check(lib.mdt_op_rms_fwd(xd.data_ptr(), gd.data_ptr(), out.data_ptr(), M, D, 1e-8, _s()))
    scratch = torch.empty(lib.mdt_op_rms_bwd_scratch(M, D), device="cuda")
    L.check(lib.mdt_op_rms_bwd(xd.data_ptr(), gd.data_ptr(), dyd.data_ptr(), dx.data_ptr(), 0, dg.data_ptr(), 0, M, D, 1e-8,
                               scratch.data_ptr(), _s()))
    assert_close(out.cpu(), ref.detach(), what="rmsnorm")
    assert_close(dx.cpu(), x64.grad, what="rmsnorm dx")
    assert_close(dg.cpu(), g64.grad, rtol=1e-3, atol=1e-4 * max(1.0, (M / 64) ** 0.5), what="rmsnorm dg")
    # the norm at the head of a residual branch: dx = d_res + the above, bit for bit the separate sum
    d_res = torch.from_numpy(synthetic.normal("d_res", (M, D), 166)).cuda()
    dx2, dg2 = torch.full_like(xd, float("nan")), torch.empty_like(gd)
    L.check(lib.mdt_op_rms_bwd_res(xd.data_ptr(), gd.data_ptr(), dyd.data_ptr(), d_res.data_ptr(), dx2.data_ptr(), dg2.data_ptr(), 0, M,
                                   D, 1e-8, scratch.data_ptr(), _s()))
    assert torch.equal(dx2, dx + d_res) and torch.equal(dg2, dg)
    H = D // 2
    u64 = x.double().requires_grad_()
    a, b = u64.tensor_split(2, dim=-1)
    r2 = a * torch.nn.functional.silu(b)
    d2 = dy[:, :H]
    r2.backward(d2.double())
    o2, du = torch.empty(M, H, device="cuda"), torch.empty_like(xd)
    L.check(lib.mdt_op_swiglu_fwd(xd.data_ptr(), o2.data_ptr(), M, H, _s()))
    L.check(lib.mdt_op_swiglu_bwd(xd.data_ptr(), d2.contiguous().cuda().data_ptr(), du.data_ptr(), M, H, _s()))
    assert_close(o2.cpu(), r2.detach(), what="swiglu")
    assert_close(du.cpu(), u64.grad, what="swiglu du")


def _facade(kw, P):
    from mdt_policy_amd.models.img_generation.masked_transformer_decoder import MaskedTransformerImgDecoder
    m = MaskedTransformerImgDecoder(**kw)
    m.load_state_dict(P, strict=True)
    return m.cuda()


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_hip_head_matches_reference_goldens_and_float64_autograd(name):
    meta, fx, kw, P, ctx, img, shuffle = case(name)
    m = _facade(kw, P)
    c = ctx.cuda().requires_grad_()
    noise = torch.from_numpy(fx["restore"]).float().cuda()  # argsort(noise) == the fixture's shuffle
    rec, mask, restore, vis = m(c, img.cuda(), noise=noise)
    loss = m.compute_loss(img.cuda(), rec, mask, restore)
    loss.backward()
    torch.cuda.synchronize()
    assert_close(rec.detach().cpu(), fx["rec"], what="reconstructions vs reference")
    assert np.array_equal(mask.cpu().numpy(), fx["mask"]) and np.array_equal(restore.cpu().numpy(), fx["restore"])
    assert_close(vis.detach().cpu(), fx["visible"], what="visible patches vs reference")
    ref_loss = float(fx["loss"].reshape(-1)[0])
    assert abs(loss.item() - ref_loss) <= 1e-3 * abs(ref_loss)
    assert_close(c.grad.cpu(), fx["d_ctx"], rtol=2e-3, atol=2e-3 * float(np.abs(fx["d_ctx"]).max()), what="d_context vs reference")
    check_summaries({k: summary(p.grad) for k, p in m.named_parameters() if p.grad is not None}, meta["grads"], name + " vs reference")
    # full tensors against float64 autograd through the oracle
    P64 = {k: v.double().requires_grad_(k != "decoder_pe") for k, v in P.items()}
    r64, m64, _, _ = O.forward(P64, kw, ctx.double(), img.double(), shuffle)
    O.compute_loss(kw, img.double(), r64, m64).backward()
    for k, p in m.named_parameters():
        if not p.requires_grad:
            continue
        ref = P64[k].grad
        assert_close(p.grad.cpu(), ref, rtol=2e-3, atol=2e-3 * float(ref.abs().max()) + 1e-9, what=k)


@pytest.mark.gpu
def test_hip_head_training_batch_against_the_oracle_on_gpu_fp64():
    """A training-sized batch (B = 96: 9 792 decoder rows, the GEMMs' large-M geometries and split-K dW products) against
    float64 autograd through the oracle."""
    meta, fx, kw, P, _, _, _ = case("default")
    B = 96
    ctx = torch.from_numpy(synthetic.normal("ctx", (B, 4, kw["context_dim"]), 171))
    img = torch.from_numpy(synthetic.normal("img", (B, 2, 3, kw["resolution"], kw["resolution"]), 172))
    noise = torch.from_numpy(synthetic.uniform("mask_noise", (B, 49), 173))
    m = _facade(kw, P)
    c = ctx.cuda().requires_grad_()
    rec, mask, restore, _ = m(c, img.cuda(), noise=noise.cuda())
    loss = m.compute_loss(img.cuda(), rec, mask, restore)
    loss.backward()
    shuffle = torch.argsort(noise, dim=1)
    torch.set_num_threads(min(32, max(8, torch.get_num_threads())))
    P64 = {k: v.double().requires_grad_(k != "decoder_pe") for k, v in P.items()}
    c64 = ctx.double().requires_grad_()
    r64, m64, _, _ = O.forward(P64, kw, c64, img.double(), shuffle)
    l64 = O.compute_loss(kw, img.double(), r64, m64)
    l64.backward()
    assert abs(loss.item() - l64.item()) <= 1e-3 * abs(l64.item())
    assert_close(rec.detach().cpu(), r64.detach(), what="B=96 reconstructions")
    assert_close(c.grad.cpu(), c64.grad, rtol=2e-3, atol=2e-3 * float(c64.grad.abs().max()), what="B=96 d_context")
    for k, p in m.named_parameters():
        if p.requires_grad:
            ref = P64[k].grad
            assert_close(p.grad.cpu(), ref, rtol=2e-3, atol=2e-3 * float(ref.abs().max()) + 1e-9, what=f"B=96 {k}")


@pytest.mark.gpu
@pytest.mark.parametrize("M,D", [(306, 192), (5, 64), (4133, 384), (64, 1024)])
def test_scale_residual_op(M, D):
    """LayerScale on a residual branch, x + gamma * z, and its backward (dz, dgamma) against float64 torch."""
    from mdt_policy_amd.models.img_generation import _hip_ops as ops
    x = torch.from_numpy(synthetic.normal("sr_x", (M, D), 171)).cuda().requires_grad_()
    z = torch.from_numpy(synthetic.normal("sr_z", (M, D), 172)).cuda().requires_grad_()
    gamma = torch.from_numpy(synthetic.normal("sr_g", (D,), 173)).cuda().requires_grad_()
    w = torch.from_numpy(synthetic.normal("sr_w", (M, D), 174))
    out = ops.HipScaleResidual.apply(x, z, gamma)
    (out * w.cuda()).sum().backward()
    x64, z64, g64 = (t.detach().cpu().double().requires_grad_() for t in (x, z, gamma))
    ref = x64 + g64 * z64
    (ref * w.double()).sum().backward()
    assert_close(out.detach().cpu(), ref.detach(), what="x + gamma z")
    assert_close(x.grad.cpu(), x64.grad, what="dx")
    assert_close(z.grad.cpu(), z64.grad, what="dz")
    assert_close(gamma.grad.cpu(), g64.grad, rtol=1e-3, atol=1e-4 * float(g64.grad.abs().max()), what="dgamma")


@pytest.mark.gpu
@pytest.mark.parametrize("M,D", [(306, 192), (5, 64), (4133, 384), (130, 500)])
def test_scale_residual_with_the_next_norm(M, D):
    """x' = x + gamma z, h = RMSNorm(x') in one launch each way, against float64 autograd; with and without a gradient on the
    residual path; the the residual sum bit-equal to the stand-alone op."""
    L, lib = _lib()
    x = torch.from_numpy(synthetic.normal("x", (M, D), 171))
    z = torch.from_numpy(synthetic.normal("z", (M, D), 172))
    gamma = torch.from_numpy(synthetic.normal("gamma", (D,), 173, std=0.3))
    gn = torch.from_numpy(synthetic.normal("gn", (D,), 174, std=0.1, mean=1.0))
    d_h = torch.from_numpy(synthetic.normal("d_h", (M, D), 175))
    d_res = torch.from_numpy(synthetic.normal("d_res", (M, D), 176))
    xd, zd, gd, gnd, dhd, drd = (t.cuda() for t in (x, z, gamma, gn, d_h, d_res))
    xn, h = torch.empty_like(xd), torch.empty_like(xd)
    L.check(lib.mdt_op_scale_residual_rms_fwd(xd.data_ptr(), zd.data_ptr(), gd.data_ptr(), gnd.data_ptr(), xn.data_ptr(), h.data_ptr(),
                                              M, D, 1e-8, _s()))
    xn2, h2 = torch.empty_like(xd), torch.empty_like(xd)
    L.check(lib.mdt_op_scale_residual_fwd(xd.data_ptr(), zd.data_ptr(), gd.data_ptr(), xn2.data_ptr(), M, D, _s()))
    L.check(lib.mdt_op_rms_fwd(xn2.data_ptr(), gnd.data_ptr(), h2.data_ptr(), M, D, 1e-8, _s()))
    assert torch.equal(xn, xn2)
    assert_close(h.cpu(), h2.cpu().double(), rtol=1e-5, atol=1e-6, what="h against the stand-alone norm")
    for with_res in (True, False):
        v = [t.double().requires_grad_() for t in (x, z, gamma, gn)]
        xr = v[0] + v[2] * v[1]
        hr = O.rms_norm(xr, v[3])
        (hr * d_h.double()).sum().backward(retain_graph=with_res)
        if with_res:
            (xr * d_res.double()).sum().backward()
        dx, dz = torch.full_like(xd, float("nan")), torch.full_like(xd, float("nan"))
        dgam, dgn = torch.empty_like(gd), torch.empty_like(gnd)
        scratch = torch.empty(lib.mdt_op_scale_residual_rms_bwd_scratch(M, D), device="cuda")
        L.check(lib.mdt_op_scale_residual_rms_bwd(xn.data_ptr(), gnd.data_ptr(), dhd.data_ptr(), drd.data_ptr() if with_res else None,
                                                  zd.data_ptr(), gd.data_ptr(), dx.data_ptr(), dz.data_ptr(), dgam.data_ptr(),
                                                  dgn.data_ptr(), M, D, 1e-8, scratch.data_ptr(), _s()))
        tol = dict(rtol=1e-3, atol=1e-4 * max(1.0, (M / 64) ** 0.5))
        assert_close(dx.cpu(), v[0].grad, what="d_x")
        assert_close(dz.cpu(), v[1].grad, what="d_z")
        assert_close(dgam.cpu(), v[2].grad, what="d_gamma", **tol)
        assert_close(dgn.cpu(), v[3].grad, what="d_gnorm", **tol)
    for bad_d in (516, 190):
        assert lib.mdt_op_scale_residual_rms_fwd(xd.data_ptr(), zd.data_ptr(), gd.data_ptr(), gnd.data_ptr(), xn.data_ptr(), h.data_ptr(),
                                                 M, bad_d, 1e-8, _s()) != 0


@pytest.mark.gpu
@pytest.mark.parametrize("B,X,C_,R,P", [(3, 2, 3, 112, 16), (2, 1, 1, 32, 8), (1, 2, 3, 224, 16), (5, 2, 3, 48, 16), (2, 3, 4, 64, 4)])
def test_patch_loss_op(B, X, C_, R, P):
    """Masked per-patch MSE straight from the image planes (one workgroup per row of patches) against the patchified form in
    float64 (reference compute_loss :228-262: element order inside a patch is (row, column, channel)), value and gradient."""
    from mdt_policy_amd.models.img_generation import _hip_ops as ops
    g_ = R // P
    n, E = g_ * g_, P * P * C_
    gen = torch.Generator().manual_seed(7 * B + R)
    rec = torch.randn(B, X, n, E, generator=gen)
    imgs = torch.randn(B, X, C_, R, R, generator=gen)
    mask = (torch.rand(B, n, generator=gen) < 0.75).float()
    mask[0, 0] = 1.0
    patches = imgs.double().reshape(B, X, C_, g_, P, g_, P).permute(0, 1, 3, 5, 4, 6, 2).reshape(B, X, n, E)
    r64 = rec.double().requires_grad_()
    per_patch = ((r64 - patches) ** 2).mean(-1)                       # (B, X, n)
    ref = sum((per_patch[:, x] * mask.double()).sum() / mask.double().sum() for x in range(X)) / X
    ref.backward()
    rd = rec.cuda().requires_grad_()
    loss = ops.HipPatchMSE.apply(rd, imgs.cuda(), mask.cuda(), P)
    loss.backward()
    assert_close(loss.detach().cpu().reshape(()), ref.detach().reshape(()), rtol=1e-5, atol=1e-6, what="loss")
    assert_close(rd.grad.cpu(), r64.grad, rtol=1e-4, atol=1e-8, what="d_rec")
    assert (rd.grad.cpu()[mask[:, None, :, None].expand(B, X, n, E) == 0] == 0).all()


@pytest.mark.gpu
@pytest.mark.parametrize("M,K,H", [(306, 192, 768), (37, 64, 48), (4100, 192, 768), (64, 128, 128), (8192 + 77, 192, 768),
                                   (9000, 64, 96)])
def test_swiglu_on_the_gemm_epilogues(M, K, H):
    """SwishGLU riding on the GEMMs around it: forward on the project product (aux_mode 3: interleaved weight image, u and
    projected * silu(gate) from one launch), backward on mlp.1's input-gradient product (dx_act = SWIGLU), through the fused
    autograd function against float64 torch of Linear -> SwishGLU -> Linear."""
    from mdt_policy_amd.models.img_generation import _hip_ops as ops
    N1 = 80 if K == 64 else K
    x = torch.from_numpy(synthetic.normal("g_x", (M, K), 181)).cuda().requires_grad_()
    w0 = torch.from_numpy(synthetic.normal("g_w0", (2 * H, K), 182, std=K ** -0.5)).cuda().requires_grad_()
    b0 = torch.from_numpy(synthetic.normal("g_b0", (2 * H,), 183, std=0.3)).cuda().requires_grad_()
    w1 = torch.from_numpy(synthetic.normal("g_w1", (N1, H), 184, std=H ** -0.5)).cuda().requires_grad_()
    b1 = torch.from_numpy(synthetic.normal("g_b1", (N1,), 185, std=0.3)).cuda().requires_grad_()
    wy = torch.from_numpy(synthetic.normal("g_wy", (M, N1), 186))
    y = ops.HipSwiGLUMLP.apply(x, w0, b0, w1, b1, ops.PackedWeights())
    (y * wy.cuda()).sum().backward()
    torch.cuda.synchronize()
    r = [t.detach().cpu().double().requires_grad_() for t in (x, w0, b0, w1, b1)]
    u = r[0] @ r[1].T + r[2]
    p, g = u.tensor_split(2, dim=-1)
    ref = (p * torch.nn.functional.silu(g)) @ r[3].T + r[4]
    (ref * wy.double()).sum().backward()
    assert_close(y.detach().cpu(), ref.detach(), what="fused SwishGLU MLP output")
    for name, got, want in zip(("dx", "dW0", "db0", "dW1", "db1"), (x, w0, b0, w1, b1), r):
        sc = float(want.grad.abs().max())
        assert_close(got.grad.cpu(), want.grad, rtol=2e-3, atol=2e-3 * sc, what=name)
    if M >= 8192:
        # from 8192 rows on the K = 192 SwishGLU products run on the weight-stationary body (mdt_ws.h; the last 32-row tile here is
        # ragged).  Its fp32 MFMA form has the MFMA form and K order of the 32-row tiles: the geometry hook that forces those must
        # give the same bits.  Its default form splits every operand into three bf16 parts (six bf16 MFMA products per k32 step,
        # round 6): fp32 accuracy, other bits -- held to a few fp32 roundings of the fp32 form.
        from mdt_policy_amd import _lib
        lib = _lib.load()

        def run(geometry, split):
            t = [v.detach().clone().requires_grad_() for v in (x, w0, b0, w1, b1)]
            lib.mdt_op_set_gemm_geometry(geometry)
            lib.mdt_op_set_ws_split(split)
            try:
                yy = ops.HipSwiGLUMLP.apply(*t, ops.PackedWeights())
                (yy * wy.cuda()).sum().backward()
                torch.cuda.synchronize()
            finally:
                lib.mdt_op_set_gemm_geometry(0)
                lib.mdt_op_set_ws_split(-1)
            return yy.detach(), t[0].grad
        y_ws, dx_ws = run(0, 0)
        y_rt, dx_rt = run(9, 0)
        assert torch.equal(y_rt, y_ws), "weight-stationary SwishGLU forward differs from the 32-row tiles"
        assert torch.equal(dx_rt, dx_ws), "weight-stationary SwishGLU backward (dx) differs from the 32-row tiles"
        y_sp, dx_sp = run(0, 1)
        if K == 192:   # (other K: the products are not on the weight-stationary body at all)
            assert not torch.equal(y_sp, y_ws) and not torch.equal(dx_sp, dx_ws), "the bf16 split form did not run"
        if os.environ.get("MDT_HIP_WS_SPLIT", "1") != "0":
            assert torch.equal(y_sp, y.detach()) and torch.equal(dx_sp, x.grad), "the split form is not the default"
        for name, got, want in (("y", y_sp, y_ws), ("dx", dx_sp, dx_ws)):
            err, sc = float((got - want).abs().max()), float(want.abs().max())
            assert err <= 1e-5 * sc, f"{name}: split against fp32 products {err:.3g} at max {sc:.3g}"
