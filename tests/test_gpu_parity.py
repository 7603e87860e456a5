"""Model-level parity on the MI355X (pytest -m gpu): the HIP path, driven through the reference-shaped facade
(GCDenoiser / sample_* -> ctypes -> libmdt_hip.so), against (a) the golden outputs of the reference and
(b) the CPU oracle on the same seeded inputs.  Gate: rtol 1e-3 / atol 1e-4 (BASELINE.json north_star)."""
import numpy as np
import pytest
import torch

from oracle import mdt_oracle as O
from tests.helpers import assert_close, cfg_of, inputs_of, load_fixture, params_of

pytestmark = pytest.mark.gpu

_MODELS = {}


def build(meta, sigma_data=0.5):
    """Facade model on the GPU carrying the fixture's synthetic weights (cached per fixture config)."""
    from mdt_policy_amd.models.edm_diffusion.score_wrappers import GCDenoiser
    key = (meta["config"], str(meta.get("overrides")), meta["weight_seed"], meta["profile"])
    if key not in _MODELS:
        model = GCDenoiser(cfg_of(meta), sigma_data=sigma_data)
        missing = model.load_state_dict(params_of(meta), strict=True)
        assert not missing.missing_keys and not missing.unexpected_keys
        _MODELS[key] = model.cuda().eval()
    return _MODELS[key]


def gpu_inputs(meta):
    state, goal, noise = inputs_of(meta)
    state = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in state.items()}
    return state, goal.cuda(), noise.cuda()


def sampling():
    from mdt_policy_amd.models.edm_diffusion import gc_sampling
    return gc_sampling


@pytest.mark.parametrize("arch", ["mdtv", "mdt"])
def test_g1_tiny(arch):
    meta, fx = load_fixture(f"g1_tiny_{arch}.npz")
    model = build(meta)
    state, goal, noise = gpu_inputs(meta)
    gs = sampling()
    sig = gs.get_sigmas_exponential(meta["n_steps"], meta["sigma_min"], meta["sigma_max"])
    np.testing.assert_allclose(sig.numpy(), fx["sigmas"], rtol=1e-6)
    with torch.no_grad():
        out = gs.sample_ddim(model, state, noise * meta["sigma_max"], goal, sig, disable=True)
        assert_close(model.inner_model.latent_encoder_emb.cpu(), fx["ctx"], what="ctx")
        assert_close(out.cpu(), fx["actions"], what="actions (fused loop)")
        steps = []
        out2 = gs.sample_ddim(model, state, noise * meta["sigma_max"], goal, sig, disable=True,
                              callback=lambda d: steps.append(d["denoised"].cpu()))
    assert_close(torch.stack(steps), fx["denoised_steps"], what="per-step denoised")
    assert_close(out2.cpu(), fx["actions"], what="actions (python loop)")


@pytest.mark.parametrize("arch", ["mdtv", "mdt"])
@pytest.mark.parametrize("modality", ["lang", "vis"])
def test_g1_forward_context_only(arch, modality):
    meta, fx = load_fixture(f"g1_ctxonly_{arch}_{modality}.npz")
    model = build(meta)
    state, goal, noise = gpu_inputs(meta)
    with torch.no_grad():
        ctx = model.forward_context_only(state, noise, goal, torch.full((meta["B"],), 2.5, device="cuda"))
    assert_close(ctx.cpu(), fx["ctx"], what="ctx")


def test_g2_denoiser_forward_per_sample_sigma():
    meta, fx = load_fixture("g2_stages_mdtv.npz")
    model = build(meta)
    state, goal, noise = gpu_inputs(meta)
    sigma = torch.tensor(meta["sigma"], device="cuda")
    with torch.no_grad():
        out = model(state, noise * sigma[:, None, None], goal, sigma)
        assert_close(model.inner_model.latent_encoder_emb.cpu(), fx["ctx"], what="ctx")
        assert_close(out.cpu(), fx["denoised"], what="denoised")
        # raw score network (inner_model.forward): input already scaled by c_in, output un-preconditioned
        c_in = 1 / (sigma ** 2 + 0.25).sqrt()
        raw = model.inner_model(state, noise * sigma[:, None, None] * c_in[:, None, None], goal, sigma)
        assert_close(raw.cpu(), fx["action_pred"], what="action_pred")
        sig = sampling().get_sigmas_exponential(10, meta["sigma_min"], meta["sigma_max"])
        act = sampling().sample_ddim(model, state, noise * 80.0, goal, sig)
    assert_close(act.cpu(), fx["actions"], what="actions")


@pytest.mark.parametrize("tag", ["lang", "vis", "init"])
def test_g3_b256_full_size(tag):
    """BASELINE C2 at full size: B=256, d=384, 10 DDIM steps, vs the reference's own output."""
    meta, fx = load_fixture(f"g3_b256_{tag}.npz")
    model = build(meta)
    state, goal, noise = gpu_inputs(meta)
    with torch.no_grad():
        out = sampling().sample_ddim(model, state, noise * meta["sigma_max"], goal, torch.from_numpy(fx["sigmas"]))
    assert_close(out.cpu(), fx["actions"], what=f"B=256 actions ({tag})")


def test_b256_sampler_call_is_bit_reproducible_across_wave_schedules():
    """The full-size call 150 times over: the same bits every time, and the same bits whether the fused MLP launch runs its
    waves in lockstep (workgroup barrier) or skewed with LDS flags (a lost flag or a stale hidden column would differ)."""
    from mdt_policy_amd import _lib
    L = _lib.load()
    meta, fx = load_fixture("g3_b256_lang.npz")
    model = build(meta)
    state, goal, noise = gpu_inputs(meta)
    sig = torch.from_numpy(fx["sigmas"])
    gs = sampling()
    try:
        with torch.no_grad():
            L.mdt_op_set_mlp_skew(0)
            want = gs.sample_ddim(model, state, noise * meta["sigma_max"], goal, sig).clone()
            L.mdt_op_set_mlp_skew(-1)
            for i in range(150):
                got = gs.sample_ddim(model, state, noise * meta["sigma_max"], goal, sig)
                if i % 10 == 0 or i == 149:
                    assert torch.equal(got, want), f"call {i} differs from the lockstep result"
    finally:
        L.mdt_op_set_mlp_skew(-1)


def test_g3_eval_schedule_and_hoisting():
    meta, fx = load_fixture("g3_b8_smin1.npz")
    model = build(meta)
    state, goal, noise = gpu_inputs(meta)
    gs = sampling()
    sig = gs.get_sigmas_exponential(meta["n_steps"], meta["sigma_min"], meta["sigma_max"])
    with torch.no_grad():
        fused = gs.sample_ddim(model, state, noise * 80.0, goal, sig)
        # "as written": the encoder is re-run by every model(...) call, like the reference
        x = noise * 80.0
        s_in = x.new_ones([x.shape[0]])
        for i in range(len(sig) - 1):
            den = model(state, x, goal, sig[i].item() * s_in)
            t, tn = -sig[i].log(), -sig[i + 1].log()
            x = ((-tn).exp() / (-t).exp()).item() * x - (-(tn - t)).expm1().item() * den
    assert_close(fused.cpu(), fx["actions"], what="fused")
    assert_close(x.cpu(), fx["actions"], what="as-written loop")
    assert_close(x.cpu(), fused.cpu(), rtol=1e-5, atol=1e-5, what="hoisted vs as-written")


def test_g3_mdt_default():
    meta, fx = load_fixture("g3_b8_mdt.npz")
    model = build(meta)
    state, goal, noise = gpu_inputs(meta)
    with torch.no_grad():
        out = sampling().sample_ddim(model, state, noise * 80.0, goal, torch.from_numpy(fx["sigmas"]))
    assert_close(model.inner_model.latent_encoder_emb.cpu(), fx["ctx"], what="ctx")
    assert_close(out.cpu(), fx["actions"], what="actions")


def test_g4_loss_forward():
    from mdt_policy_amd import synthetic
    meta, fx = load_fixture("g4_loss.npz")
    model = build(meta)
    state, goal, _ = gpu_inputs(meta)
    li = {k: torch.from_numpy(v).cuda() for k, v in synthetic.loss_inputs(meta["B"], cfg_of(meta), meta["loss_seed"]).items()}
    with torch.no_grad():
        loss, mo = model.loss(state, li["actions"], goal, li["noise_train"], li["sigma"])
    assert_close(mo.cpu(), fx["model_output"], what="model_output")
    assert_close(loss.cpu(), fx["loss"].reshape(()), rtol=1e-3, atol=1e-6, what="loss")


def test_g6_rope():
    meta, fx = load_fixture("g6_rope.npz")
    model = build(meta)
    state, goal, noise = gpu_inputs(meta)
    with torch.no_grad():
        out = sampling().sample_ddim(model, state, noise * 80.0, goal, torch.from_numpy(fx["sigmas"]))
    assert_close(model.inner_model.latent_encoder_emb.cpu(), fx["ctx"], what="ctx")
    assert_close(out.cpu(), fx["actions"], what="actions")


@pytest.mark.parametrize("sched", ["exp", "karras"])
@pytest.mark.parametrize("name", ["ddim", "euler", "heun", "dpmpp_2m"])
def test_g7_other_samplers(name, sched):
    meta, fx = load_fixture("g7_samplers.npz")
    model = build(meta)
    state, goal, noise = gpu_inputs(meta)
    gs = sampling()
    sig = (gs.get_sigmas_exponential if sched == "exp" else gs.get_sigmas_karras)(10, 0.001, 80.0)
    with torch.no_grad():
        out = getattr(gs, "sample_" + name)(model, state, noise * 80.0, goal, sig)
    assert_close(out.cpu(), fx[f"{name}_{sched}"], what=f"{name}/{sched}")


def test_against_oracle_random_batch_sizes():
    """HIP vs oracle on fresh seeded inputs at ragged batch sizes (tile-boundary edge cases)."""
    meta, _ = load_fixture("g3_b256_lang.npz")
    model = build(meta)
    cfg, P = cfg_of(meta), params_of(meta)
    sig = O.get_sigmas_exponential(3, 0.01, 80.0)
    for B in (1, 3, 7, 33):
        m2 = dict(meta, B=B, input_seed=100 + B)
        state, goal, noise = inputs_of(m2)
        want = O.sample_ddim(P, cfg, state, noise * 80.0, goal, sig, hoist=True)
        with torch.no_grad():
            got = sampling().sample_ddim(model, {"state_images": state["state_images"].cuda(), "modality": "lang"},
                                         noise.cuda() * 80.0, goal.cuda(), sig)
        assert_close(got.cpu(), want, what=f"B={B}")


# one batch size per regime of the launch dispatcher (mdt_model.hip: run_self_attn / run_mlp / run_decoder_blocks and the
# geometry choice of mdt_launch_gemm): half-height 16 x 64 tiles (192 < M <= 1400 rows: B = 20 ... 140), the last batch before
# / the first batch on the one-workgroup-per-sample middle + the fused MLP launch (1401 rows: B = 140 | 141), the second round
# of per-sample workgroups (B = 257 ... 512), and the first batch beyond it (B = 513: k_attn + GEMM + k_xattn_apply again)
# (round 5: B = 9 ... 32 run the self-attention inside its projection launch like the rollout batches -- B = 12, 16, 20, 32; B = 33 is
#  the first batch back on k_attn + projection)
# (round 6: from 768 rows a launch the MLP sublayer runs fused and, like the qkv products, as three-way bf16 splits: the decoder from
#  B = 77 -- 76 is the last fp32 batch --, the encoder's four-token rows from B = 192)
REGIME_BATCHES = (12, 16, 20, 32, 33, 64, 76, 77, 128, 140, 141, 192, 257, 300, 512, 513)


@pytest.mark.parametrize("B", REGIME_BATCHES)
def test_every_dispatcher_regime_against_the_oracle(B):
    """3-step sample_ddim (gc_sampling.py:922-951) on fresh seeded inputs, HIP vs the CPU oracle, at one batch size per
    kernel-selection regime: the op-level tests pin each kernel, this pins their composition."""
    meta, _ = load_fixture("g3_b256_lang.npz")
    model = build(meta)
    cfg, P = cfg_of(meta), params_of(meta)
    sig = O.get_sigmas_exponential(3, 0.01, 80.0)
    m2 = dict(meta, B=B, input_seed=700 + B)
    state, goal, noise = inputs_of(m2)
    want = O.sample_ddim(P, cfg, state, noise * 80.0, goal, sig, hoist=True)
    with torch.no_grad():
        got = sampling().sample_ddim(model, {"state_images": state["state_images"].cuda(), "modality": "lang"},
                                     noise.cuda() * 80.0, goal.cuda(), sig)
    assert_close(got.cpu(), want, what=f"B={B} actions")


@pytest.mark.parametrize("B", (128, 300))
def test_denoiser_forward_with_per_sample_sigma_in_the_mid_batch_regimes(B):
    """GCDenoiser.forward (score_wrappers.py:65-80) with one sigma per sample (adaLN rows per sample: PRO_LN_MOD_ROWS, the
    per-sample gate of the epilogues) at batch sizes between the small-M and the B = 256 kernels, HIP vs the oracle."""
    meta, _ = load_fixture("g3_b256_lang.npz")
    model = build(meta)
    cfg, P = cfg_of(meta), params_of(meta)
    m2 = dict(meta, B=B, input_seed=900 + B)
    state, goal, noise = inputs_of(m2)
    g = torch.Generator().manual_seed(31 + B)
    sigma = torch.exp(torch.rand(B, generator=g) * 8.0 - 5.0)  # 0.0067 ... 20
    x = noise * sigma[:, None, None]
    want = O.denoise(P, cfg, state, x, goal, sigma)
    with torch.no_grad():
        got = model({"state_images": state["state_images"].cuda(), "modality": "lang"}, x.cuda(), goal.cuda(), sigma.cuda())
    assert_close(got.cpu(), want, what=f"B={B} denoised, per-sample sigma")


def test_full_size_properties():
    """Size-independent properties at B=256: determinism and batch independence (a sample's actions do not
    depend on what else is in the batch).  Bit-exact between batches served by the same GEMM kernels (the k-order of
    every dot product is fixed); where the dispatcher picks the split-K small-M kernel for one batch size and tiles for the
    other, equal to rounding."""
    meta, _ = load_fixture("g3_b256_lang.npz")
    model = build(meta)
    state, goal, noise = gpu_inputs(meta)
    sig = sampling().get_sigmas_exponential(10, 0.001, 80.0)
    with torch.no_grad():
        a = sampling().sample_ddim(model, state, noise * 80.0, goal, sig)
        b = sampling().sample_ddim(model, state, noise * 80.0, goal, sig)
        assert torch.equal(a, b), "non-deterministic"
        for sub, exact in ((slice(10, 250), True), (slice(100, 117), False)):  # 240 / 17 samples (the dispatcher's kernel choice per product is the same at 240 and 256)
            st = {"state_images": state["state_images"][sub].contiguous(), "modality": state["modality"]}
            c = sampling().sample_ddim(model, st, noise[sub] * 80.0, goal[sub], sig)
            if exact:
                assert torch.equal(a[sub], c), "batch dependence"
            else:
                assert_close(c.cpu(), a[sub].cpu(), rtol=1e-5, atol=2e-6, what="batch dependence (small-M kernel)")
    assert torch.isfinite(a).all()


def test_deepcopy_and_pickle_of_a_live_model():
    """EMA helpers and checkpointing code deep-copy / pickle whole modules: a module with a live library handle must
    copy as parameters only and build its own handle on first use."""
    import copy
    import pickle
    meta, fx = load_fixture("g1_tiny_mdtv.npz")
    model = build(meta)
    state, goal, noise = gpu_inputs(meta)
    sig = sampling().get_sigmas_exponential(meta["n_steps"], meta["sigma_min"], meta["sigma_max"])
    with torch.no_grad():
        a = sampling().sample_ddim(model, state, noise * meta["sigma_max"], goal, sig)
        twin = copy.deepcopy(model)
        again = pickle.loads(pickle.dumps(model))
        assert twin.inner_model._engines == {} and again.inner_model._engines == {}
        b = sampling().sample_ddim(twin, state, noise * meta["sigma_max"], goal, sig)
        c = sampling().sample_ddim(again, state, noise * meta["sigma_max"], goal, sig)
    assert torch.equal(a, b) and torch.equal(a, c)
    assert twin.inner_model.hip_engine() is not model.inner_model.hip_engine()


def test_parameter_updates_reach_the_kernels():
    meta, _ = load_fixture("g2_stages_mdtv.npz")
    from mdt_policy_amd.models.edm_diffusion.score_wrappers import GCDenoiser
    model = GCDenoiser(cfg_of(meta), 0.5)
    model.load_state_dict(params_of(meta))
    model = model.cuda().eval()
    state, goal, noise = gpu_inputs(meta)
    sigma = torch.full((4,), 1.5, device="cuda")
    with torch.no_grad():
        a = model(state, noise, goal, sigma)
        model.inner_model.action_pred.bias.add_(1.0)  # in-place update, like an optimizer / EMA copy
        b = model(state, noise, goal, sigma)
        new = {k: v * 1.01 for k, v in params_of(meta).items()}
        model.load_state_dict(new)
        c = model(state, noise, goal, sigma)
    c_out = (1.5 * 0.5) / (1.5 ** 2 + 0.25) ** 0.5
    assert_close((b - a).cpu(), torch.full_like(a, c_out).cpu(), rtol=1e-3, atol=1e-4, what="bias shift")
    assert not torch.allclose(c, a)


def test_fails_loudly_instead_of_falling_back():
    meta, _ = load_fixture("g1_tiny_mdtv.npz")
    from mdt_policy_amd.models.edm_diffusion.score_wrappers import GCDenoiser
    model = GCDenoiser(cfg_of(meta), 0.5)
    model.load_state_dict(params_of(meta))
    state, goal, noise = inputs_of(meta)
    with torch.no_grad(), pytest.raises(RuntimeError):
        model.eval()(state, noise, goal, torch.ones(1))  # CPU tensors / CPU model: no CPU path exists
    model = model.cuda()
    state = {"state_images": state["state_images"].cuda(), "modality": "lang"}
    with pytest.raises(NotImplementedError):
        model.eval()(state, noise.cuda(), goal.cuda(), torch.ones(1, device="cuda"))  # autograd not implemented
    with torch.no_grad(), pytest.raises(NotImplementedError):
        model.train()(state, noise.cuda(), goal.cuda(), torch.ones(1, device="cuda"))  # dropout not implemented
    with torch.no_grad(), pytest.raises(Exception):
        model.eval()(state, noise.cuda(), goal.cuda()[:, :, :100], torch.ones(1, device="cuda"))  # bad goal shape
    with torch.no_grad(), pytest.raises(ValueError, match="state_obs"):  # the proprioceptive token (tests/test_proprio.py): bad width
        model.eval()(dict(state, state_obs=torch.zeros(1, 1, 5, device="cuda")), noise.cuda(), goal.cuda(),
                     torch.ones(1, device="cuda"))


def test_rollout_batch_one_matches_the_oracle_at_the_default_size():
    """B = 1 (the reference's rollout batch) takes the small-batch kernels, incl. the self-attention fused into its output
    projection (8 heads of 48): the fused sampler loop against the oracle's sample_ddim."""
    from mdt_policy_amd import configs
    from mdt_policy_amd.models.edm_diffusion import gc_sampling as gs
    from mdt_policy_amd.models.edm_diffusion.score_wrappers import GCDenoiser
    from mdt_policy_amd import synthetic
    cfg = configs.mdtv_default()
    model = GCDenoiser(cfg, 0.5)
    shapes = [(k, tuple(v.shape)) for k, v in model.state_dict().items()]
    P = {k: torch.from_numpy(v) for k, v in synthetic.fill_state_dict(shapes, 5, "rich").items()}
    model.load_state_dict(P, strict=False)
    model = model.cuda().eval()
    inp = {k: torch.from_numpy(v) for k, v in synthetic.sampler_inputs(1, cfg, 6).items()}
    state = {"state_images": inp["state_images"], "modality": "lang"}
    sig = gs.get_sigmas_exponential(5, 0.001, 80.0)
    x_T = inp["noise"] * 80.0
    want = O.sample_ddim(P, cfg, state, x_T, inp["goal"], sig)
    with torch.no_grad():
        got = gs.sample_ddim(model, {"state_images": inp["state_images"].cuda(), "modality": "lang"}, x_T.cuda(),
                             inp["goal"].cuda(), sig)
    assert_close(got.cpu(), want, what="B = 1 actions")


def test_replayed_rollout_calls_leave_fresh_actions_and_context():
    """Rollout-sized calls switch to a HIP-graph replay from the third call with the same shapes on (gc_sampling._graph_wanted).
    The replay owns static output buffers; what the caller gets -- the actions AND inner_model.latent_encoder_emb, which the
    reference assigns afresh at every forward (mdtv_transformer.py:221; read at mdtv_agent.py:256,330,445) -- must be tensors
    the next call does not overwrite, bit-equal to the eager call on the same inputs."""
    from mdt_policy_amd import configs, synthetic
    from mdt_policy_amd.models.edm_diffusion import gc_sampling as gs
    from mdt_policy_amd.models.edm_diffusion.score_wrappers import GCDenoiser
    cfg = configs.mdtv_default()
    model = GCDenoiser(cfg, 0.5)
    shapes = [(k, tuple(v.shape)) for k, v in model.state_dict().items()]
    P = {k: torch.from_numpy(v) for k, v in synthetic.fill_state_dict(shapes, 5, "rich").items()}
    model.load_state_dict(P, strict=False)
    model = model.cuda().eval()
    sig = gs.get_sigmas_exponential(5, 0.001, 80.0).cuda()
    calls = []
    for seed in range(6):  # six different observations, one chunk each
        inp = {k: torch.from_numpy(v).cuda() for k, v in synthetic.sampler_inputs(1, cfg, 20 + seed).items()}
        calls.append(({"state_images": inp["state_images"], "modality": "lang"}, inp["noise"] * 80.0, inp["goal"]))
    eager = []
    with torch.no_grad():
        for st, x, g in calls:  # the eager native loop (the facade method never replays)
            a = model.sample_ddim(st, x, g, sig)
            eager.append((a.clone(), model.inner_model.latent_encoder_emb.clone()))
        got = []
        for st, x, g in calls:
            a = gs.sample_ddim(model, st, x, g, sig)
            got.append((a, model.inner_model.latent_encoder_emb))
    torch.cuda.synchronize()
    assert getattr(model, "_graphed_samplers", None), "the replay path was not taken"
    for i, ((a, c), (ea, ec)) in enumerate(zip(got, eager)):
        assert torch.equal(a, ea), f"call {i}: actions differ from the eager call (or were overwritten by a later call)"
        assert torch.equal(c, ec), f"call {i}: latent_encoder_emb differs from the eager call (or was overwritten by a later call)"
    ptrs = {t.data_ptr() for a, c in got for t in (a, c)}
    assert len(ptrs) == 2 * len(got), "two calls handed out the same buffer"
    inp0 = {k: torch.from_numpy(v) for k, v in synthetic.sampler_inputs(1, cfg, 25).items()}
    want = O.sample_ddim(P, cfg, {"state_images": inp0["state_images"], "modality": "lang"}, inp0["noise"] * 80.0, inp0["goal"], sig.cpu())
    assert_close(got[5][0].cpu(), want, what="replayed B = 1 actions")


def test_collapsed_and_explicit_cross_attention_paths_agree(monkeypatch):
    """MDT_HIP_XFOLD=0 keeps the q-GEMM / attention / c_proj-GEMM sequence; both must match the reference golden."""
    from mdt_policy_amd.models.edm_diffusion.score_wrappers import GCDenoiser
    meta, fx = load_fixture("g2_stages_mdtv.npz")
    state, goal, noise = gpu_inputs(meta)
    sig = sampling().get_sigmas_exponential(10, meta["sigma_min"], meta["sigma_max"])
    outs = {}
    for flag in ("1", "0"):
        monkeypatch.setenv("MDT_HIP_XFOLD", flag)
        model = GCDenoiser(cfg_of(meta), 0.5)
        model.load_state_dict(params_of(meta))
        model = model.cuda().eval()
        with torch.no_grad():
            outs[flag] = sampling().sample_ddim(model, state, noise * 80.0, goal, sig).cpu()
        assert_close(outs[flag], fx["actions"], what=f"actions (xfold={flag})")
    assert_close(outs["1"], outs["0"], rtol=1e-4, atol=1e-5, what="collapsed vs explicit")
    assert not torch.equal(outs["1"], outs["0"])  # different arithmetic: the two paths really are distinct


G8 = ["bias", "plain_goal", "two_tokens", "mdt_bias_nopos", "no_ada", "noise_block", "mdt_no_ada", "mlp_head", "mdt_mlp_head",
      "no_goal_cond", "mdt_no_goal_cond"]


@pytest.mark.parametrize("name", G8)
def test_g8_constructor_variants(name):
    meta, fx = load_fixture(f"g8_{name}.npz")
    from mdt_policy_amd.models.edm_diffusion.score_wrappers import GCDenoiser
    model = GCDenoiser(cfg_of(meta), 0.5)
    assert [[k, list(v.shape)] for k, v in model.state_dict().items()] == meta["state_dict"]
    model.load_state_dict(params_of(meta))
    model = model.cuda().eval()
    state, goal, noise = gpu_inputs(meta)
    gs = sampling()
    with torch.no_grad():
        steps = []
        out2 = gs.sample_ddim(model, state, noise * meta["sigma_max"], goal, torch.from_numpy(fx["sigmas"]),
                              callback=lambda d: steps.append(d["denoised"].cpu()))
        out = gs.sample_ddim(model, state, noise * meta["sigma_max"], goal, torch.from_numpy(fx["sigmas"]))
    assert_close(model.inner_model.latent_encoder_emb.cpu(), fx["ctx"], what="ctx")
    assert_close(torch.stack(steps), fx["denoised_steps"], what="denoised")
    assert_close(out.cpu(), fx["actions"], what="actions (fused loop)")
    assert_close(out2.cpu(), fx["actions"], what="actions (python loop)")


@pytest.mark.parametrize("name", ["no_ada", "noise_block", "mdt_no_ada", "mlp_head", "mdt_mlp_head", "no_goal_cond",
                                  "mdt_no_goal_cond"])
def test_conditioning_variants_entry_points_against_the_oracle(name):
    """use_ada_conditioning=False (sigma token in the context, plain decoder) and use_noise_encoder=True (NoiseBlock):
    forward with one sigma per sample, the split encoder/decoder calls, the loss, and another sampler, against the
    oracle (itself pinned on these variants by the g8 fixtures)."""
    meta, _ = load_fixture(f"g8_{name}.npz")
    model = build(meta)
    cfg, P, arch = cfg_of(meta), params_of(meta), meta["arch"]
    state, goal, noise = inputs_of(meta)
    gstate, ggoal, gnoise = gpu_inputs(meta)
    B = noise.shape[0]
    sigma = torch.tensor([0.07, 1.3, 40.0])[:B]
    im = model.inner_model
    with torch.no_grad():
        out = model(gstate, gnoise, ggoal, sigma.cuda())
        assert_close(out.cpu(), O.denoise(P, cfg, state, noise, goal, sigma, arch=arch), what="forward")
        assert_close(im.latent_encoder_emb.cpu(), O.encode(P, cfg, state, goal, arch, sigma=sigma), what="ctx")
        ctx = model.forward_context_only(gstate, gnoise, ggoal, sigma.cuda())
        assert_close(ctx.cpu(), O.forward_context_only(P, cfg, state, goal, arch, sigma=sigma), what="ctx only")
        if arch == "mdtv":  # MDT's forward_enc_only does not cache its context (mdt_transformer.py:257-281)
            raw = im.forward_dec_only(ctx, gnoise, sigma.cuda())
            assert_close(raw.cpu(), O.decode(P, cfg, O.encode(P, cfg, state, goal, arch, "forward_enc_only", sigma=sigma),
                                             noise, sigma), what="dec only")
        act = noise * 0.3
        eps = torch.randn(noise.shape, generator=torch.Generator().manual_seed(5))
        loss, _ = model.loss(gstate, act.cuda(), ggoal, eps.cuda(), sigma.cuda())
        ref_loss, _ = O.loss(P, cfg, state, act, goal, eps, sigma, arch=arch)
        assert_close(loss.cpu(), ref_loss, what="loss")
        gs = sampling()
        sig = gs.get_sigmas_karras(4, 0.01, 80.0)
        got = gs.sample_heun(model, gstate, gnoise * 80.0, ggoal, sig)
        assert_close(got.cpu(), O.sample_heun(P, cfg, state, noise * 80.0, goal, sig, arch=arch), what="heun")
        B5 = gs.sample_ddim(model, gstate, gnoise * 80.0, ggoal, gs.get_sigmas_exponential(5, 0.001, 80.0))
        ref5 = O.sample_ddim(P, cfg, state, noise * 80.0, goal, O.get_sigmas_exponential(5, 0.001, 80.0), arch=arch)
        assert_close(B5.cpu(), ref5, what="ddim 5 steps")


@pytest.mark.parametrize("name,kw,key", [
    ("lms", {}, "lms"), ("dpm_2", {}, "dpm_2"), ("dpmpp_2_with_lms", {}, "dpmpp_2_with_lms"), ("dpmpp_2s", {}, "dpmpp_2s"),
    ("euler_ancestral", dict(eta=0.), "euler_ancestral_eta0"), ("dpm_2_ancestral", dict(eta=0.), "dpm_2_ancestral_eta0"),
    ("dpmpp_2s_ancestral", dict(eta=0.), "dpmpp_2s_ancestral_eta0")])
def test_g7b_remaining_samplers(name, kw, key):
    """The rest of sample_loop's dispatch table with the HIP denoiser step (deterministic settings)."""
    meta, fx = load_fixture("g7b_samplers.npz")
    model = build(meta)
    state, goal, noise = gpu_inputs(meta)
    gs = sampling()
    with torch.no_grad():
        out = getattr(gs, "sample_" + name)(model, state, noise * 80.0, goal, gs.get_sigmas_exponential(10, 0.001, 80.0), **kw)
    assert_close(out.cpu(), fx[key], what=key)


def test_g7c_dpm_solver_fast_sde_and_adaptive():
    """DPM-Solver-fast and DPM-Solver++ SDE with the HIP denoiser step against the reference's outputs; the adaptive
    solver (no reference output exists: the reference's cannot run) against the same solver driven by the oracle."""
    from mdt_policy_amd import synthetic
    meta, fx = load_fixture("g7c_samplers.npz")
    model = build(meta)
    state, goal, noise = gpu_inputs(meta)
    gs = sampling()
    x0 = noise * 80.0
    fixed = torch.from_numpy(synthetic.normal("sde_noise", tuple(x0.shape), meta["noise_seed"])).cuda()
    ns = lambda s0, s1: fixed
    with torch.no_grad():
        for nfe in (9, 10, 11):
            out = gs.sample_dpm_fast(model, state, x0.clone(), goal, 0.001, 80.0, nfe, noise_sampler=ns)
            assert_close(out.cpu(), fx[f"dpm_fast_nfe{nfe}"], what=f"dpm_fast nfe={nfe}")
        sig = gs.get_sigmas_exponential(10, 0.001, 80.0)
        for eta, key in ((0., "dpmpp_sde_eta0"), (1., "dpmpp_sde_eta1_fixednoise")):
            out = gs.sample_dpmpp_sde(model, state, x0.clone(), goal, sig, eta=eta, noise_sampler=ns)
            assert_close(out.cpu(), fx[key], what=key)
        got, info = gs.sample_dpm_adaptive(model, state, x0.clone(), goal, 0.01, 80.0, return_info=True)
    cfg, P = cfg_of(meta), params_of(meta)
    cstate, cgoal, cnoise = inputs_of(meta)
    ctx = O.encode(P, cfg, cstate, cgoal)
    want, winfo = gs.sample_dpm_adaptive(lambda s, x, g, sg: O.denoise(P, cfg, s, x, g, sg, ctx=ctx), cstate, cnoise * 80.0,
                                         cgoal, 0.01, 80.0, return_info=True)
    assert info["nfe"] == winfo["nfe"] and info["n_accept"] == winfo["n_accept"]
    assert_close(got.cpu(), want, rtol=2e-3, atol=5e-4, what="dpm_adaptive (HIP step vs oracle step)")


def test_c4_total_batch_2048_matches_the_b256_golden_slice():
    """BASELINE config C4's total request (B = 2048) on one GPU: the first 256 chunks are the G3 inputs, the rest
    fresh seeds; their actions must equal the B=256 reference golden (batch independence at the maximum size)."""
    meta, fx = load_fixture("g3_b256_lang.npz")
    model = build(meta)
    state, goal, noise = gpu_inputs(meta)
    extra = dict(meta, B=2048 - 256, input_seed=777)
    s2, g2, n2 = gpu_inputs(extra)
    big_state = {"state_images": torch.cat([state["state_images"], s2["state_images"]]), "modality": "lang"}
    with torch.no_grad():
        out = sampling().sample_ddim(model, big_state, torch.cat([noise, n2]) * 80.0, torch.cat([goal, g2]),
                                     torch.from_numpy(fx["sigmas"]))
    assert out.shape == (2048, 10, 7) and torch.isfinite(out).all()
    assert_close(out[:256].cpu(), fx["actions"], what="first 256 of 2048")


def test_mark_dirty_picks_up_writes_the_version_counter_misses():
    """`p.data.mul_()` does not bump `p._version` (p.data carries its own counter): the packed arena would go stale
    silently.  `mark_dirty()` (and every train()/eval() switch, load_state_dict) forces the re-upload."""
    from mdt_policy_amd.models.edm_diffusion.score_wrappers import GCDenoiser
    meta, fx = load_fixture("g1_tiny_mdtv.npz")
    model = GCDenoiser(cfg_of(meta), sigma_data=0.5)
    model.load_state_dict(params_of(meta))
    model = model.cuda().eval()
    state, goal, noise = gpu_inputs(meta)
    sig = sampling().get_sigmas_exponential(meta["n_steps"], meta["sigma_min"], meta["sigma_max"])
    with torch.no_grad():
        a = sampling().sample_ddim(model, state, noise * meta["sigma_max"], goal, sig).clone()
        w = model.inner_model.action_pred.weight
        v0 = w._version
        w.data.mul_(0.5)
        assert w._version == v0  # the hazard: nothing the cache could have noticed
        model.inner_model.mark_dirty()
        b = sampling().sample_ddim(model, state, noise * meta["sigma_max"], goal, sig).clone()
        w.data.mul_(2.0)
        model.eval()  # a mode switch re-validates too
        c = sampling().sample_ddim(model, state, noise * meta["sigma_max"], goal, sig)
    assert_close(a.cpu(), fx["actions"], what="before")
    assert (a - b).abs().max().item() > 1e-3, "halved action_pred.weight must change the actions"
    assert torch.equal(a, c)


def test_unchanged_parameter_fast_path_still_sees_every_kind_of_change():
    """Round 5: after a call that uploaded nothing, `sync_params` checks the parameters with three sweeps over cached lists (same
    objects, same version counters, same storage pointers).  Every way a parameter can change must still leave that fast path: an
    in-place update (version counter), `p.data = ...` (storage pointer, counter untouched), a Parameter REPLACED by assignment
    (object identity), `mark_dirty()`."""
    from mdt_policy_amd.models.edm_diffusion.score_wrappers import GCDenoiser
    meta, fx = load_fixture("g1_tiny_mdtv.npz")
    model = GCDenoiser(cfg_of(meta), sigma_data=0.5)
    model.load_state_dict(params_of(meta))
    model = model.cuda().eval()
    state, goal, noise = gpu_inputs(meta)
    sig = sampling().get_sigmas_exponential(meta["n_steps"], meta["sigma_min"], meta["sigma_max"])
    run = lambda: sampling().sample_ddim(model, state, noise * meta["sigma_max"], goal, sig).clone()
    im = model.inner_model
    eng = im.hip_engine(0.5, state)
    with torch.no_grad():
        a = run(); a2 = run(); a3 = run()
        assert eng._fast is not None, "two unchanged calls must have armed the fast path"
        assert torch.equal(a, a2) and torch.equal(a, a3)
        im.action_pred.weight.mul_(0.5)                       # version counter
        b = run()
        assert eng._fast is not None
        im.action_pred.weight.data = im.action_pred.weight.data.clone() * 2.0   # new storage, same counter
        c = run()
        old = im.action_pred.weight
        im.action_pred.weight = torch.nn.Parameter(old.detach().clone() * 0.5)    # another Parameter object
        d = run()
        im.action_pred.weight.data.mul_(2.0)                  # invisible to every check ...
        im.mark_dirty()                                       # ... hence the escape hatch
        e = run()
    assert_close(a.cpu(), fx["actions"], what="before")
    assert (a - b).abs().max().item() > 1e-3 and torch.equal(b, d), "halved weights must change the actions, the same way both times"
    assert torch.equal(a, c) and torch.equal(a, e)


def test_cached_context_is_not_poisoned_by_an_interleaved_encode():
    """Inside ``cached_context`` a call with ANOTHER state / goal (classifier-free guidance's unconditional branch, a callback
    evaluating something else) re-encodes on the same handle; later calls with the original pair must not decode against
    that context (the engine's generation counter sends them through the full forward again)."""
    meta, fx = load_fixture("g3_b8_smin1.npz")
    model = build(meta)
    state, goal, noise = gpu_inputs(meta)
    other = {k: (v.flip(0).contiguous() if torch.is_tensor(v) else v) for k, v in state.items()}
    sig = torch.full((meta["B"],), 2.5, device="cuda")
    x = noise * 2.5
    with torch.no_grad():
        want = model(state, x, goal, sig).clone()
        with model.cached_context(state, goal):
            a = model(state, x, goal, sig).clone()
            model(other, x, goal.flip(0).contiguous(), sig)       # overwrites the handle's cached context
            b = model(state, x, goal, sig).clone()
            u = model(state, x, goal, sig, uncond=True).clone()    # zeroed goal: another context again
            c = model(state, x, goal, sig).clone()
    assert_close(a.cpu(), want.cpu(), rtol=1e-5, atol=1e-6, what="cached call")
    assert_close(b.cpu(), want.cpu(), rtol=1e-5, atol=1e-6, what="after an interleaved encode")
    assert_close(c.cpu(), want.cpu(), rtol=1e-5, atol=1e-6, what="after an unconditional call")
    assert (u - want).abs().max().item() > 1e-3


def test_goal_sequences_are_sliced_only_where_the_reference_slices():
    """preprocess_goals (mdtv_transformer.py:249) keeps goal[:, 0] only when the goal sequence is as long as the state
    sequence; another length must surface as an error, not be cut silently."""
    meta, fx = load_fixture("g3_b8_smin1.npz")
    model = build(meta)
    state, goal, noise = gpu_inputs(meta)
    sig = torch.full((meta["B"],), 2.5, device="cuda")
    with torch.no_grad():
        want = model(state, noise, goal, sig)
        n_tok = state["state_images"].shape[1]
        same = model(state, noise, goal.expand(-1, n_tok, -1).contiguous(), sig)   # as long as the states: first entry
        assert torch.equal(want, same)
        with pytest.raises(ValueError):
            model(state, noise, goal.expand(-1, n_tok + 2, -1).contiguous(), sig)


def test_batched_parameter_upload_matches_the_per_parameter_path():
    """mdt_load_params (every parameter in one launch; host-resident sources fall back to the staged per-parameter copy) must
    leave exactly the images mdt_load_param leaves: same sampled actions, bit for bit, from two handles loaded either way --
    with and without the W^T images of the training path, and with one source on the host."""
    import ctypes as C
    from mdt_policy_amd import _lib
    from mdt_policy_amd.models.edm_diffusion.score_wrappers import GCDenoiser
    meta, fx = load_fixture("g1_tiny_mdtv.npz")
    P = params_of(meta)
    state, goal, noise = gpu_inputs(meta)
    sig = sampling().get_sigmas_exponential(meta["n_steps"], meta["sigma_min"], meta["sigma_max"])
    lib = _lib.load()

    def actions(loader, train):
        model = GCDenoiser(cfg_of(meta), sigma_data=0.5)
        model.load_state_dict(P)
        model = model.cuda().eval()
        eng = model.inner_model.hip_engine(0.5, state)
        if train:
            eng.train_prepare()
        stream = torch.cuda.current_stream().cuda_stream
        named = [("inner_model." + n, p.detach()) for n, p in model.inner_model.named_parameters() if "inner_model." + n in eng.expected]
        keep = loader(eng, named, stream)
        eng._uploaded = {k: (p.data_ptr(), p._version) for k, p in
                         (("inner_model." + n, p) for n, p in model.inner_model.named_parameters()) if k in eng.expected}
        with torch.no_grad():
            out = sampling().sample_ddim(model, state, noise * meta["sigma_max"], goal, sig).clone()
        torch.cuda.synchronize()
        del keep
        return out

    def one_by_one(eng, named, stream):
        for k, t in named:
            _lib.check(lib.mdt_load_param(eng.handle, k.encode(), t.data_ptr(), t.numel(), stream))
        return named

    def batched(eng, named, stream):
        named = [(k, (t.cpu() if i == 3 else t)) for i, (k, t) in enumerate(named)]  # one host-resident source
        n = len(named)
        names = (C.c_char_p * n)(*[k.encode() for k, _ in named])
        srcs = (C.c_void_p * n)(*[t.data_ptr() for _, t in named])
        numels = (C.c_int64 * n)(*[t.numel() for _, t in named])
        _lib.check(lib.mdt_load_params(eng.handle, n, names, srcs, numels, stream))
        torch.cuda.synchronize()
        return named

    for train in (False, True):
        a, b = actions(one_by_one, train), actions(batched, train)
        assert torch.equal(a, b), f"train images = {train}"
        assert_close(a.cpu(), fx["actions"], what="actions")
    # an unknown name is refused, nothing is launched
    eng = GCDenoiser(cfg_of(meta), sigma_data=0.5).cuda().inner_model.hip_engine(0.5, state)
    t = torch.zeros(4, device="cuda")
    names = (C.c_char_p * 1)(b"inner_model.no_such.weight")
    srcs = (C.c_void_p * 1)(t.data_ptr())
    numels = (C.c_int64 * 1)(4)
    assert lib.mdt_load_params(eng.handle, 1, names, srcs, numels, torch.cuda.current_stream().cuda_stream) != 0


def test_split_sampler_launches_are_bit_reproducible_call_after_call():
    """Round 6: from 768 rows a launch the MLP sublayer and the qkv products run as three-way bf16 splits (k_mlp_split,
    k_gemm_ln_split).  No atomics, fixed summation orders: the same inputs must give the same BITS call after call, on the default
    stream and on a side stream, at a batch that uses them in the decoder only (B = 100) and one that uses them in the encoder too
    (B = 200)."""
    from mdt_policy_amd import configs, synthetic
    from mdt_policy_amd.models.edm_diffusion import gc_sampling as gs
    from mdt_policy_amd.models.edm_diffusion.score_wrappers import GCDenoiser
    cfg = configs.mdtv_default()
    torch.manual_seed(0)
    model = GCDenoiser(cfg, 0.5).cuda().eval()
    sig = gs.get_sigmas_exponential(4, 0.001, 80.0).cuda()
    side = torch.cuda.Stream()
    for B in (100, 200):
        inp = {k: torch.from_numpy(v).cuda() for k, v in synthetic.sampler_inputs(B, cfg, 11).items()}
        st = {"state_images": inp["state_images"], "modality": "lang"}
        ref = None
        with torch.no_grad():
            for i in range(12):
                if i % 3 == 2:
                    side.wait_stream(torch.cuda.current_stream())
                    with torch.cuda.stream(side):
                        a = model.sample_ddim(st, inp["noise"] * 80.0, inp["goal"], sig).clone()
                    torch.cuda.current_stream().wait_stream(side)
                else:
                    a = model.sample_ddim(st, inp["noise"] * 80.0, inp["goal"], sig).clone()
                torch.cuda.synchronize()
                if ref is None:
                    ref = a
                else:
                    assert torch.equal(a, ref), f"B = {B}: call {i} differs from the first one"


@pytest.mark.parametrize("over", [dict(embed_dim=256), dict(embed_dim=128, n_heads=4)])
def test_other_model_widths_at_a_split_batch_against_the_oracle(over):
    """d = 256: the fused MLP launch has a split instantiation (two hidden slices), the qkv products keep the fp32 launch (the split
    one is instantiated for d = 384); d = 128: one hidden slice, nothing fused, nothing split.  B = 100 (1000 decoder rows: above
    the split forms' row count), 3-step sample_ddim against the CPU oracle, split on and off."""
    from mdt_policy_amd import _lib, configs, synthetic
    from mdt_policy_amd.models.edm_diffusion import gc_sampling as gs
    from mdt_policy_amd.models.edm_diffusion.score_wrappers import GCDenoiser
    L = _lib.load()
    cfg = dict(configs.mdtv_default(), **over)
    torch.manual_seed(0)
    m = GCDenoiser(cfg, 0.5)
    shapes = [(k, tuple(v.shape)) for k, v in m.state_dict().items()]
    P = {k: torch.from_numpy(v) for k, v in synthetic.fill_state_dict(shapes, 5, "rich").items()}
    m.load_state_dict(P, strict=False)
    m = m.cuda().eval()
    sig = gs.get_sigmas_exponential(3, 0.01, 80.0)
    inp = {k: torch.from_numpy(v) for k, v in synthetic.sampler_inputs(100, cfg, 7).items()}
    st = {"state_images": inp["state_images"], "modality": "lang"}
    want = O.sample_ddim(P, cfg, st, inp["noise"] * 80.0, inp["goal"], sig, hoist=True)
    res = {}
    try:
        with torch.no_grad():
            for split in (1, 0):
                L.mdt_op_set_mlp_split(split)
                res[split] = m.sample_ddim({"state_images": inp["state_images"].cuda(), "modality": "lang"}, inp["noise"].cuda() * 80.0,
                                           inp["goal"].cuda(), sig.cuda()).cpu()
    finally:
        L.mdt_op_set_mlp_split(-1)
    for split in (1, 0):
        assert_close(res[split], want, what=f"d = {cfg['embed_dim']}, split {split}")
    assert torch.equal(res[0], res[1]) == (cfg["embed_dim"] == 128), "which widths run the split MLP launch changed"


def test_mdt_architecture_at_a_split_batch_against_the_oracle():
    """The MDT architecture (d = 512, six decoder blocks; g3_b8_mdt's weights and configuration) at B = 100: its fused MLP launch
    (four hidden slices, four column tiles per wave in the second product) and its qkv products (N = 1536: four 384-wide panels)
    run as three-way bf16 splits too.  3-step sample_ddim against the CPU oracle with the split forms on and off; the split must
    really have run."""
    from mdt_policy_amd import _lib
    L = _lib.load()
    meta, _ = load_fixture("g3_b8_mdt.npz")
    model = build(meta)
    cfg, P = cfg_of(meta), params_of(meta)
    sig = O.get_sigmas_exponential(3, 0.01, 80.0)
    state, goal, noise = inputs_of(dict(meta, input_seed=731), batch=100)
    want = O.sample_ddim(P, cfg, state, noise * 80.0, goal, sig, arch="mdt", hoist=True)
    gstate = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in state.items()}
    res = {}
    try:
        with torch.no_grad():
            for split in (1, 0):
                L.mdt_op_set_mlp_split(split)
                res[split] = sampling().sample_ddim(model, gstate, noise.cuda() * 80.0, goal.cuda(), sig).cpu()
    finally:
        L.mdt_op_set_mlp_split(-1)
    for split in (1, 0):
        assert_close(res[split], want, what=f"MDT, B = 100, split {split}")
    assert not torch.equal(res[0], res[1]), "the split launches did not run for the d = 512 model"
