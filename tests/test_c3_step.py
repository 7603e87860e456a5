"""BASELINE configs[2] at its own size (B = 1024): the masked generative foresight head alone, and the COMBINED training step
exactly as bench.py's `train_step_c3_mdtv_B1024` leg runs it -- GCDenoiser.loss + gen_img(latent_encoder_emb, imgs) + ONE
FusedAdamW over both modules (mdtv_agent.py:258-269, 411-421) -- against float64 autograd through the two oracles (run on the
GPU's fp64 units; the oracles are the checkers, never the thing measured) and torch.optim.AdamW on the float64 copies.

B = 1024 reaches code the B <= 96 tests never do: 104 448 decoder rows of the head (other split-K slice counts, the 32 x 192
tiles chosen for training-sized row counts, k_gemm_tn's deep reductions), 10 240 action rows of the denoiser, and the
encoder gradient that arrives from BOTH losses through latent_encoder_emb.  Dropout is off (eval mode): the oracle has no
counterpart of the library's Philox streams; everything else is the bench's step."""
import pytest
import torch

from mdt_policy_amd import synthetic
from oracle import mae_oracle as MO
from oracle import mdt_oracle as O
from tests.helpers import assert_close, cfg_of, inputs_of, load_fixture, params_of

pytestmark = pytest.mark.gpu

B = 1024
GEN_KW = dict(resolution=112, patch_size=16, decoder_depth=6, decoder_embed_dim=192, decoder_n_heads=8, context_dim=384,
              mask_ratio=0.75)


def _head_params():
    meta, fx = load_fixture("g15_mae_default.npz")
    kw = meta["kwargs"]
    for k, v in GEN_KW.items():
        assert kw[k] == v, (k, kw[k], v)  # the fixture's head IS the bench's head
    shapes = [(k, tuple(s)) for k, s in meta["state_dict"] if k != "decoder_pe"]
    P = {k: torch.from_numpy(v) for k, v in synthetic.fill_state_dict(shapes, meta["weight_seed"], meta["profile"]).items()}
    P["decoder_pe"] = torch.from_numpy(fx["decoder_pe"])
    return kw, P


def _facade_head(kw, P):
    from mdt_policy_amd.models.img_generation.masked_transformer_decoder import MaskedTransformerImgDecoder
    m = MaskedTransformerImgDecoder(**kw)
    m.load_state_dict(P, strict=True)
    return m.cuda()


def _check_grads(named, ref_of, what):
    n = 0
    for k, p in named:
        if not p.requires_grad:
            continue
        ref = ref_of(k)
        if ref is None:
            assert p.grad is None, k
            continue
        scale = float(ref.abs().max())
        assert_close(p.grad.cpu(), ref.cpu(), rtol=2e-3, atol=2e-3 * scale + 1e-9, what=f"{what} {k}")
        n += 1
    return n


def test_mgf_head_forward_backward_at_the_c3_batch_b1024():
    """(i) reconstructions, loss, d_context and EVERY parameter gradient of the head at B = 1024."""
    kw, P = _head_params()
    ctx = torch.from_numpy(synthetic.normal("ctx", (B, 4, kw["context_dim"]), 181))
    img = torch.from_numpy(synthetic.normal("img", (B, 2, 3, kw["resolution"], kw["resolution"]), 182))
    noise = torch.from_numpy(synthetic.uniform("mask_noise", (B, 49), 183))
    m = _facade_head(kw, P)
    c = ctx.cuda().requires_grad_()
    rec, mask, restore, _ = m(c, img.cuda(), noise=noise.cuda())
    loss = m.compute_loss(img.cuda(), rec, mask, restore)
    loss.backward()
    torch.cuda.synchronize()
    dev = "cuda"
    P64 = {k: v.double().to(dev).requires_grad_(k != "decoder_pe") for k, v in P.items()}
    c64 = ctx.double().to(dev).requires_grad_()
    shuffle = torch.argsort(noise, dim=1).to(dev)
    r64, m64, _, _ = MO.forward(P64, kw, c64, img.double().to(dev), shuffle)
    l64 = MO.compute_loss(kw, img.double().to(dev), r64, m64)
    l64.backward()
    assert abs(loss.item() - l64.item()) <= 1e-3 * abs(l64.item())
    assert torch.equal(mask.cpu().double(), m64.cpu())
    assert_close(rec.detach().cpu(), r64.detach().cpu(), what="B=1024 reconstructions")
    assert_close(c.grad.cpu(), c64.grad.cpu(), rtol=2e-3, atol=2e-3 * float(c64.grad.abs().max()), what="B=1024 d_context")
    assert _check_grads(m.named_parameters(), lambda k: P64[k].grad, "B=1024 head") > 50


def test_combined_c3_step_follows_the_oracles_for_three_adamw_steps():
    """(ii) the bench's step_c3, three times: losses of every step, every gradient of the first step (both modules: the
    encoder receives d_context from both losses), and the weights after three FusedAdamW steps against torch.optim.AdamW
    on the float64 oracles."""
    from mdt_policy_amd.models.edm_diffusion.score_wrappers import GCDenoiser
    from mdt_policy_amd.optim import FusedAdamW
    meta, fx = load_fixture("g11_grads_mdtv_default.npz")
    cfg = cfg_of(meta)
    state, goal, _ = inputs_of(meta, batch=B)
    li = {k: torch.from_numpy(v) for k, v in synthetic.loss_inputs(B, cfg, meta["loss_seed"]).items()}
    kw, PH = _head_params()
    img = torch.from_numpy(synthetic.normal("img", (B, 2, 3, kw["resolution"], kw["resolution"]), 192))
    noises = [torch.from_numpy(synthetic.uniform("mask_noise", (B, 49), 193 + i)) for i in range(3)]
    lr, wd = 1e-4, 0.05

    model = GCDenoiser(cfg, 0.5)
    model.load_state_dict(params_of(meta))
    model = model.cuda().eval()
    gen = _facade_head(kw, PH)
    opt = FusedAdamW(list(model.parameters()) + list(gen.parameters()), lr=lr, weight_decay=wd)
    gstate = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in state.items()}
    ggoal, gimg = goal.cuda(), img.cuda()
    gl = {k: v.cuda() for k, v in li.items()}

    dev = "cuda"
    PD = {k: v.double().to(dev).requires_grad_(v.dtype.is_floating_point) for k, v in params_of(meta).items()}
    P64 = {k: v.double().to(dev).requires_grad_(k != "decoder_pe") for k, v in PH.items()}
    st64 = {k: (v.double().to(dev) if torch.is_tensor(v) else v) for k, v in state.items()}
    g64, i64 = goal.double().to(dev), img.double().to(dev)
    l64 = {k: v.double().to(dev) for k, v in li.items()}
    used = [k for k in PD if PD[k].requires_grad and "proprio_emb" not in k and "rotary" not in k and k != "inner_model.pos_emb"]
    opt64 = torch.optim.AdamW([PD[k] for k in used] + [v for v in P64.values() if v.requires_grad], lr=lr, weight_decay=wd)
    w0 = {k: p.detach().clone() for k, p in list(model.named_parameters()) + [("gen." + k, p) for k, p in gen.named_parameters()]}

    for step in range(3):
        opt.zero_grad(set_to_none=True)
        loss, _ = model.loss(gstate, gl["actions"], ggoal, gl["noise_train"], gl["sigma"])
        rec, mask, restore, _ = gen(model.inner_model.latent_encoder_emb, gimg, noise=noises[step].cuda())
        aux = gen.compute_loss(gimg, rec, mask, restore)
        (loss + aux).backward()

        opt64.zero_grad(set_to_none=True)
        lo, _ = O.loss(PD, cfg, st64, l64["actions"], g64, l64["noise_train"], l64["sigma"], arch=meta["arch"])
        c64 = O.encode(PD, cfg, st64, g64, meta["arch"], "forward", sigma=l64["sigma"])
        r64, m64, _, _ = MO.forward(P64, kw, c64, i64, torch.argsort(noises[step], dim=1).to(dev))
        ao = MO.compute_loss(kw, i64, r64, m64)
        (lo + ao).backward()

        assert abs(loss.item() - lo.item()) <= 2e-3 * abs(lo.item()), (step, "diffusion loss", loss.item(), lo.item())
        assert abs(aux.item() - ao.item()) <= 2e-3 * abs(ao.item()), (step, "masked-token loss", aux.item(), ao.item())
        if step == 0:
            n = _check_grads(model.named_parameters(), lambda k: PD[k].grad, "step 0 denoiser")
            n += _check_grads(gen.named_parameters(), lambda k: P64[k].grad, "step 0 head")
            assert n > 150
            g0 = {k: PD[k].grad.detach().clone() for k in used if PD[k].grad is not None}
            g0.update({"gen." + k: v.grad.detach().clone() for k, v in P64.items() if v.grad is not None})
        opt.step()
        opt64.step()
    torch.cuda.synchronize()
    # AdamW moves every weight by about lr per step whatever the gradient's size, so an element whose gradient is fp32
    # noise (key biases: exactly zero in theory, softmax is shift invariant; the k third of the head's fused qkv bias) may end
    # up to 2 * 3 * lr away.  Every entry stays inside that bound; the entries whose first gradient is clearly above the
    # gradient check's own tolerance (2 % of the tensor's largest) must move the same way.
    refs = {k: PD[k] for k in used}
    refs.update({"gen." + k: v for k, v in P64.items() if v.requires_grad})
    checked = 0
    for k, p in list(model.named_parameters()) + [("gen." + k, p) for k, p in gen.named_parameters()]:
        if not p.requires_grad or k not in refs:
            continue
        d_hip = (p.detach() - w0[k]).double().flatten()
        d_ref = (refs[k].detach() - w0[k].double()).flatten()
        if float(d_ref.norm()) == 0.0:
            assert float(d_hip.norm()) == 0.0, k
            continue
        assert float((d_hip - d_ref).abs().max()) <= 6.5 * lr, k
        g = g0[k].flatten().abs()
        sure = g >= 0.02 * g.max()
        if float(g.max()) < 1e-7 or int(sure.sum()) < 8:  # a tensor of noise-level gradients: nothing to compare beyond the bound
            continue
        a, b = d_hip[sure], d_ref[sure]
        cos = float(torch.dot(a, b) / (a.norm() * b.norm()))
        assert cos > 0.99, f"{k}: update direction cos = {cos:.5f} over {int(sure.sum())} entries"
        frac_close = float(((a - b).abs() <= 0.1 * 3 * lr).double().mean())
        assert frac_close > 0.95, f"{k}: only {frac_close:.3f} of the entries within 10 % of the three-step update"
        checked += 1
    assert checked > 120
