"""Perceiver resampler: oracle vs the reference's golden outputs on CPU, the HIP path
(facade -> ctypes -> mdt_resampler_* -> gfx950 kernels) vs both on the GPU.  Gate: rtol 1e-3 / atol 1e-4."""
import pytest
import torch

from oracle import perceiver_oracle as PO
from tests.helpers import assert_close, perceiver_case

CASES = ["default", "tiny_masked", "many_latents"]


@pytest.mark.parametrize("name", CASES)
def test_oracle_matches_the_reference_golden(name):
    meta, fx, P, x, mask = perceiver_case(name)
    out = PO.perceiver_resampler(P, x, meta["kwargs"]["heads"], mask)
    assert_close(out.numpy(), fx["out"], rtol=1e-4, atol=2e-5, what=name)


def _grad_summary(g):
    g = g.detach().double().cpu()
    return [float(g.norm()), float(g.sum())] + [float(v) for v in g.flatten()[:6]]


def _check_reference_grads(meta, fx, named_grads, dx, what):
    """Against the REFERENCE's own backward (summaries per parameter, head of d_x_f in full)."""
    import numpy as np
    from mdt_policy_amd import synthetic  # noqa: F401
    for k, w in meta["grads"].items():
        g, w = np.array(_grad_summary(named_grads[k])), np.array(w)
        tol = 2e-3 * abs(w[0]) + 1e-6
        assert np.all(np.abs(g - w) <= tol), f"{what} {k}: {g} vs {w}"
    assert_close(dx[:, :, :4, :], fx["d_x_head"], rtol=2e-3, atol=2e-3 * float(abs(fx["d_x_head"]).max()), what=what + " d_x_f")
    s = [float(dx.double().norm()), float(dx.double().sum())]
    assert abs(s[0] - meta["d_x_summary"][0]) <= 2e-3 * meta["d_x_summary"][0]


def _cotangent(meta, shape):
    from mdt_policy_amd import synthetic
    return torch.from_numpy(synthetic.normal("cotangent", tuple(shape), meta["cot_seed"]))


@pytest.mark.parametrize("name", CASES)
def test_oracle_autograd_matches_the_reference_gradients(name):
    meta, fx, P, x, mask = perceiver_case(name)
    P64 = {k: v.double().requires_grad_() for k, v in P.items()}
    x64 = x.double().requires_grad_()
    out = PO.perceiver_resampler(P64, x64, meta["kwargs"]["heads"], mask)
    (out * _cotangent(meta, out.shape).double()).sum().backward()
    _check_reference_grads(meta, fx, {k: v.grad for k, v in P64.items()}, x64.grad, name)


@pytest.mark.parametrize("name", CASES)
def test_facade_state_dict_matches_the_reference(name):
    from mdt_policy_amd.models.networks.transformers.perceiver_resampler import PerceiverResampler
    meta, _, _, _, _ = perceiver_case(name)
    m = PerceiverResampler(**meta["kwargs"])
    assert [[k, list(v.shape)] for k, v in m.state_dict().items()] == meta["state_dict"]
    assert all(p.requires_grad for p in m.parameters())
    assert not any(p.requires_grad for p in PerceiverResampler(**meta["kwargs"], trainable=False).parameters())


def test_facade_refuses_cpu():
    from mdt_policy_amd.models.networks.transformers.perceiver_resampler import PerceiverResampler
    m = PerceiverResampler(dim=64, depth=1, dim_head=16, heads=4, num_latents=2, num_time_embeds=1)
    with pytest.raises(RuntimeError, match="ROCm GPU"):
        m(torch.zeros(1, 1, 4, 64))
    with torch.no_grad(), pytest.raises(RuntimeError, match="ROCm GPU"):
        m(torch.zeros(1, 1, 4, 64))
    with pytest.raises(NotImplementedError):
        PerceiverResampler(dim=64, depth=1, activation="relu")


def _gpu_model(meta, P):
    from mdt_policy_amd.models.networks.transformers.perceiver_resampler import PerceiverResampler
    m = PerceiverResampler(**meta["kwargs"])
    m.load_state_dict(P, strict=True)
    return m.cuda().eval()


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_hip_matches_golden_and_oracle(name):
    meta, fx, P, x, mask = perceiver_case(name)
    m = _gpu_model(meta, P)
    with torch.no_grad():
        out = m(x.cuda(), None if mask is None else mask.cuda())
        assert_close(out.cpu(), fx["out"], what=f"{name} vs reference golden")
        assert_close(out.cpu(), PO.perceiver_resampler(P, x, meta["kwargs"]["heads"], mask), what=f"{name} vs oracle")
        # same call again (workspace reuse) and on a batch slice (batch independence)
        assert torch.equal(m(x.cuda(), None if mask is None else mask.cuda()), out)
        one = m(x[1:2].cuda(), None if mask is None else mask[1:2].cuda())
        assert_close(one.cpu(), fx["out"][1:2], what="single sample")


@pytest.mark.gpu
def test_hip_rollout_and_training_batch_shapes():
    """B = 1 (rollout) and a training-sized batch of the shipped configuration against the oracle; the large
    batch is checked on a few samples and through batch independence."""
    meta, _, P, _, _ = perceiver_case("default")
    m = _gpu_model(meta, P)
    g = torch.Generator().manual_seed(7)
    x = torch.randn(64, 1, 392, 384, generator=g)
    with torch.no_grad():
        big = m(x.cuda()).cpu()
        ref = PO.perceiver_resampler(P, x[[0, 31, 63]], 8)
        assert_close(big[[0, 31, 63]], ref, what="B=64")
        assert_close(m(x[5:6].cuda()).cpu(), big[5:6], rtol=1e-5, atol=1e-6, what="B=1 vs row of B=64")
    assert m.flops(1, 392) > 1.8e9


@pytest.mark.gpu
def test_hip_parameter_updates_and_errors():
    from mdt_policy_amd._lib import MDTHipError
    meta, fx, P, x, mask = perceiver_case("tiny_masked")
    m = _gpu_model(meta, P)
    with torch.no_grad():
        a = m(x.cuda(), mask.cuda())
        m.latents.mul_(2.0)  # in-place update must reach the packed arena
        b = m(x.cuda(), mask.cuda())
        assert not torch.allclose(a, b)
        P2 = dict(P, latents=P["latents"] * 2.0)
        assert_close(b.cpu(), PO.perceiver_resampler(P2, x, meta["kwargs"]["heads"], mask), what="after update")
        with pytest.raises(MDTHipError, match="frames"):
            m(torch.zeros(1, 5, 7, 64, device="cuda"))


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_hip_backward_matches_autograd_through_the_oracle(name):
    """Gradients of every parameter and of the media tokens against float64 autograd through the oracle (itself
    pinned on the reference's outputs above); a random cotangent stands in for the downstream loss."""
    meta, fx, P, x, mask = perceiver_case(name)
    m = _gpu_model(meta, P)
    xg = x.cuda().requires_grad_()
    out = m(xg, None if mask is None else mask.cuda())
    cot = _cotangent(meta, out.shape)
    (out * cot.cuda()).sum().backward()
    assert_close(out.detach().cpu(), fx["out"], what="forward under autograd")
    _check_reference_grads(meta, fx, {k: p.grad for k, p in m.named_parameters()}, xg.grad.cpu(), name + " (HIP)")
    P64 = {k: v.double().requires_grad_() for k, v in P.items()}
    x64 = x.double().requires_grad_()
    o64 = PO.perceiver_resampler(P64, x64, meta["kwargs"]["heads"], mask)
    (o64 * cot.double()).sum().backward()
    for k, p in m.named_parameters():
        ref = P64[k].grad
        assert_close(p.grad.cpu(), ref, rtol=2e-3, atol=2e-3 * float(ref.abs().max()) + 1e-7, what=k)
    assert_close(xg.grad.cpu(), x64.grad, rtol=2e-3, atol=2e-3 * float(x64.grad.abs().max()), what="d_x_f")
    # two forwards alive, one backward (the agent resamples per modality batch before its single backward)
    m.zero_grad()
    o1, o2 = m(x.cuda(), None if mask is None else mask.cuda()), m(x.cuda() * 0.5, None if mask is None else mask.cuda())
    (o1.sum() + o2.sum()).backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in m.parameters())


@pytest.mark.gpu
def test_hip_resampler_trains_end_to_end_with_the_denoiser():
    """media tokens -> resampler -> denoiser loss, one backward through both HIP modules; gradients reach the
    resampler's parameters through d(state_images) of the denoiser."""
    from mdt_policy_amd import configs, synthetic
    from mdt_policy_amd.models.edm_diffusion.score_wrappers import GCDenoiser
    from mdt_policy_amd.models.networks.transformers.perceiver_resampler import PerceiverResampler
    torch.manual_seed(0)
    cfg = configs.mdtv_default()
    den = GCDenoiser(cfg, 0.5).cuda().eval()
    res = PerceiverResampler(dim=384, depth=2, dim_head=64, heads=8, num_latents=3, num_time_embeds=1).cuda()
    B = 4
    media = torch.randn(B, 1, 40, 384, device="cuda")
    li = {k: torch.from_numpy(v).cuda() for k, v in synthetic.loss_inputs(B, cfg, 2).items()}
    goal = torch.randn(B, 1, 512, device="cuda")
    opt = torch.optim.AdamW(list(den.parameters()) + list(res.parameters()), lr=3e-4)
    losses = []
    for _ in range(8):
        opt.zero_grad()
        state = {"state_images": res(media), "modality": "lang"}
        loss, _ = den.loss(state, li["actions"], goal, li["noise_train"], li["sigma"])
        loss.backward()
        assert res.latents.grad is not None and res.latents.grad.abs().sum() > 0
        opt.step()
        losses.append(loss.item())
    assert losses[-1] < losses[0]


@pytest.mark.gpu
@pytest.mark.parametrize("B", [1, 3])
def test_composed_rollout_path_perceiver_then_sampler_against_the_oracles(B):
    """What MDTVAgent.forward runs per replan behind its frozen encoders, composed as the agent composes it
    (mdtv_agent.py:392-403 compute_voltron_embeddings, :688-719 forward, :523-550 denoise_actions): Voltron-shaped patch tokens of
    both cameras (B, 392, 384) -> unsqueeze(1) -> PerceiverResampler (6 layers, 3 latents) -> state_images (B, 3, 384), language
    goal (B, 1, 512), x_T = noise * sigma_max, sigmas on the device -> sample_ddim.  HIP (facade modules) against
    perceiver_oracle o mdt_oracle on the same seeded inputs; bench.py times the same composition (rollout_e2e_synthetic_B1)."""
    from mdt_policy_amd import configs, synthetic
    from mdt_policy_amd.models.edm_diffusion import gc_sampling as gs
    from mdt_policy_amd.models.edm_diffusion.score_wrappers import GCDenoiser
    from mdt_policy_amd.models.networks.transformers.perceiver_resampler import PerceiverResampler
    from oracle import mdt_oracle as O
    cfg = configs.mdtv_default()
    model = GCDenoiser(cfg, 0.5)
    shapes = [(k, tuple(v.shape)) for k, v in model.state_dict().items()]
    P = {k: torch.from_numpy(v) for k, v in synthetic.fill_state_dict(shapes, 5, "rich").items()}
    model.load_state_dict(P, strict=False)
    model = model.cuda().eval()
    perc = PerceiverResampler(dim=384, depth=6, dim_head=64, heads=8, num_latents=3, num_time_embeds=1)
    g = torch.Generator().manual_seed(77)
    with torch.no_grad():
        for n, p in perc.named_parameters():  # a spread of values, not the constructor's init (LayerNorm gains 1, biases 0)
            r = torch.randn(p.shape, generator=g)
            if n in ("latents", "time_pos_emb"):
                p.copy_(r)
            elif p.dim() == 1:
                p.copy_(1.0 + 0.1 * r if n.endswith("weight") else 0.1 * r)
            else:
                p.copy_(0.05 * r)
    PP = {k: v.detach().clone() for k, v in perc.state_dict().items()}
    perc = perc.cuda().eval()
    tokens = torch.randn(B, 2 * 196, 384, generator=g)
    goal = torch.randn(B, 512, generator=g)
    noise = torch.randn(B, 10, 7, generator=g)
    sig = gs.get_sigmas_exponential(10, 0.001, 80.0)
    # oracle composition
    st_o = PO.perceiver_resampler(PP, tokens.unsqueeze(1), 8)
    want = O.sample_ddim(P, cfg, {"state_images": st_o, "modality": "lang"}, noise * 80.0, goal.unsqueeze(1), sig)
    # HIP composition, as the agent's forward()
    with torch.no_grad():
        perceptual_emb = {"state_images": perc(tokens.cuda().unsqueeze(1))}
        perceptual_emb["modality"] = "lang"
        latent_goal = goal.cuda().unsqueeze(1)
        x = noise.cuda() * 80.0
        got = gs.sample_ddim(model, perceptual_emb, x, latent_goal, sig.cuda())
    assert_close(perceptual_emb["state_images"].cpu(), st_o, what="resampled state tokens")
    assert_close(got.cpu(), want, what="composed rollout actions")
