#!/usr/bin/env python3
"""Generate the golden fixtures in tests/golden/ by running the REFERENCE on PyTorch-CPU.

Runs ONLY in the survey/build container, where /root/reference exists:

    python tests/golden/make_golden.py

It imports the reference's hot-path modules (mdt.models.edm_diffusion.*, mdt.models.networks.*) behind
four stub modules (hydra, omegaconf, torchsde, torchdiffeq are import-time-only dependencies that are not
installed here), fills the reference modules' state_dict from the build's own
deterministic generator (mdt_policy_amd/synthetic.py), feeds deterministic inputs and stores the
reference's OUTPUTS (and, for G2, per-stage intermediates) as .npz.  Weights and inputs are NOT stored:
tests regenerate them from the same generator.  Nothing of the reference's source is copied.
"""
import importlib
import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from mdt_policy_amd import configs, synthetic  # noqa: E402

REF = "/root/reference"


def install_stubs():
    def stub(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    def instantiate(cfg, *a, **kw):
        cfg = dict(cfg)
        tgt = cfg.pop("_target_")
        cfg.pop("_recursive_", None)
        mod, cls = tgt.rsplit(".", 1)
        return getattr(importlib.import_module(mod), cls)(*a, **cfg, **kw)

    hy = stub("hydra")
    hy.utils = stub("hydra.utils", instantiate=instantiate)
    stub("omegaconf", DictConfig=dict, OmegaConf=object)
    stub("torchsde")
    stub("torchdiffeq", odeint=None)
    # einops_exts.rearrange_many(tensors, pattern, **kw) = map(rearrange) (used by the Perceiver resampler only)
    import einops
    stub("einops_exts", rearrange_many=lambda ts, pattern, **kw: tuple(einops.rearrange(t, pattern, **kw) for t in ts))
    sys.path.insert(0, REF)


REF_TARGET = {
    "mdtv": "mdt.models.networks.mdtv_transformer.MDTVTransformer",
    "mdt": "mdt.models.networks.mdt_transformer.MDTTransformer",
}


def build_reference(cfg, arch, seed, profile, sigma_data=0.5):
    from mdt.models.edm_diffusion.score_wrappers import GCDenoiser

    cfg = configs.retarget(cfg, REF_TARGET[arch])
    if arch == "mdt":
        cfg.pop("n_obs_token", None)
    model = GCDenoiser(cfg, sigma_data=sigma_data).eval()
    sd = model.state_dict()
    new = synthetic.fill_state_dict([(k, tuple(v.shape)) for k, v in sd.items()], seed, profile)
    for k, v in new.items():
        sd[k] = torch.from_numpy(v)
    # without a modality encoder lang_emb IS goal_emb (one module under two names): keep one set of values
    if not cfg.get("use_modality_encoder", False):
        for k in list(sd):
            if k.startswith("inner_model.lang_emb"):
                sd[k] = sd[k.replace("inner_model.lang_emb", "inner_model.goal_emb")]
    model.load_state_dict(sd, strict=True)
    return model


def make_state(inp, arch, modality):
    if arch == "mdtv":
        return {"state_images": torch.from_numpy(inp["state_images"]), "modality": modality}
    return {"static": torch.from_numpy(inp["static"]), "gripper": torch.from_numpy(inp["gripper"]),
            "modality": modality}


def run_ddim(model, cfg, arch, B, n_steps, modality, in_seed, sigma_min=0.001, sigma_max=80.0, record=False):
    from mdt.models.edm_diffusion import gc_sampling

    inp = synthetic.sampler_inputs(B, cfg, in_seed, arch)
    state = make_state(inp, arch, modality)
    goal = torch.from_numpy(inp["goal"])
    sigmas = gc_sampling.get_sigmas_exponential(n_steps, sigma_min, sigma_max)
    x = torch.from_numpy(inp["noise"]) * sigma_max
    den = []
    cb = (lambda d: den.append(d["denoised"].clone())) if record else None
    out = gc_sampling.sample_ddim(model, state, x, goal, sigmas, disable=True, callback=cb)
    res = {"actions": out.numpy(), "sigmas": sigmas.numpy()}
    if record:
        res["denoised_steps"] = torch.stack(den).numpy()
        res["ctx"] = model.inner_model.latent_encoder_emb.detach().numpy()
    return res


def save(name, meta, **arrays):
    arrays = {k: np.ascontiguousarray(v) for k, v in arrays.items()}
    np.savez_compressed(os.path.join(HERE, name), meta=json.dumps(meta), **arrays)
    print(f"  wrote {name}: " + ", ".join(f"{k}{tuple(v.shape)}" for k, v in arrays.items()))


def g1():
    for arch, cfg in (("mdtv", configs.mdtv_tiny()), ("mdt", configs.mdt_tiny())):
        model = build_reference(cfg, arch, seed=11, profile="rich")
        res = run_ddim(model, cfg, arch, B=1, n_steps=4, modality="lang", in_seed=12, record=True)
        meta = dict(config=f"{arch}_tiny", arch=arch, B=1, n_steps=4, modality="lang", weight_seed=11,
                    profile="rich", input_seed=12, sigma_min=0.001, sigma_max=80.0)
        save(f"g1_tiny_{arch}.npz", meta, **res)
        # MDT forward_context_only honours the modality switch while forward() does not
        inp = synthetic.sampler_inputs(3, cfg, 13, arch)
        for modality in ("lang", "vis"):
            state = make_state(inp, arch, modality)
            x = torch.from_numpy(inp["noise"])
            ctx = model.forward_context_only(state, x, torch.from_numpy(inp["goal"]), torch.full((3,), 2.5))
            meta2 = dict(meta, B=3, input_seed=13, modality=modality)
            save(f"g1_ctxonly_{arch}_{modality}.npz", meta2, ctx=ctx.detach().numpy())


def g2():
    """Stage-by-stage intermediates of ONE GCDenoiser.forward (B=4, sigma per sample) + DDIM actions."""
    cfg, arch = configs.mdtv_default(), "mdtv"
    model = build_reference(cfg, arch, seed=21, profile="rich")
    inp = synthetic.sampler_inputs(4, cfg, 22, arch)
    state = make_state(inp, arch, "lang")
    goal = torch.from_numpy(inp["goal"])
    sigma = torch.tensor([80.0, 6.5, 0.53, 0.0035])
    x = torch.from_numpy(inp["noise"]) * sigma[:, None, None]
    cap = {}
    hooks = []

    def grab(key):
        def fn(_m, _i, o):
            cap[key] = (torch.cat(o, dim=-1) if isinstance(o, tuple) else o).detach().clone().numpy()
        return fn

    im = model.inner_model
    hooks.append(im.lang_emb.register_forward_hook(grab("goal_embed")))
    hooks.append(im.tok_emb.register_forward_hook(grab("state_embed")))
    for l, b in enumerate(im.encoder.blocks):
        hooks.append(b.register_forward_hook(grab(f"enc{l}")))
    hooks.append(im.encoder.register_forward_hook(grab("ctx")))
    hooks.append(im.sigma_emb.register_forward_hook(grab("sigma_emb")))
    hooks.append(im.action_emb.register_forward_hook(grab("action_emb")))
    for l, b in enumerate(im.decoder.blocks):
        hooks.append(b.adaLN_zero.register_forward_hook(grab(f"dec{l}.mod")))
        hooks.append(b.attn.register_forward_hook(grab(f"dec{l}.attn")))
        hooks.append(b.cross_att.register_forward_hook(grab(f"dec{l}.cross_att")))
        hooks.append(b.mlp.register_forward_hook(grab(f"dec{l}.mlp")))
        hooks.append(b.register_forward_hook(grab(f"dec{l}")))
    hooks.append(im.decoder.register_forward_hook(grab("decoder_ln")))
    hooks.append(im.action_pred.register_forward_hook(grab("action_pred")))
    with torch.no_grad():
        out = model(state, x, goal, sigma)
    for h in hooks:
        h.remove()
    res = run_ddim(model, cfg, arch, B=4, n_steps=10, modality="lang", in_seed=22)
    meta = dict(config="mdtv_default", arch=arch, B=4, n_steps=10, modality="lang", weight_seed=21, profile="rich",
                input_seed=22, sigma=[80.0, 6.5, 0.53, 0.0035], sigma_min=0.001, sigma_max=80.0)
    save("g2_stages_mdtv.npz", meta, denoised=out.numpy(), actions=res["actions"], sigmas=res["sigmas"], **cap)


def g3():
    cfg, arch = configs.mdtv_default(), "mdtv"
    for tag, profile, modality, wseed in (("lang", "rich", "lang", 31), ("vis", "rich", "vis", 31),
                                          ("init", "init", "lang", 0)):
        model = build_reference(cfg, arch, seed=wseed, profile=profile)
        res = run_ddim(model, cfg, arch, B=256, n_steps=10, modality=modality, in_seed=1)
        meta = dict(config="mdtv_default", arch=arch, B=256, n_steps=10, modality=modality, weight_seed=wseed,
                    profile=profile, input_seed=1, sigma_min=0.001, sigma_max=80.0)
        save(f"g3_b256_{tag}.npz", meta, actions=res["actions"], sigmas=res["sigmas"])
    # eval-time schedule of the released ABCD checkpoints (conf/mdt_evaluate.yaml: sigma_min 1.0), 5 steps
    model = build_reference(cfg, arch, seed=31, profile="rich")
    res = run_ddim(model, cfg, arch, B=8, n_steps=5, modality="lang", in_seed=2, sigma_min=1.0)
    meta = dict(config="mdtv_default", arch=arch, B=8, n_steps=5, modality="lang", weight_seed=31, profile="rich",
                input_seed=2, sigma_min=1.0, sigma_max=80.0)
    save("g3_b8_smin1.npz", meta, actions=res["actions"], sigmas=res["sigmas"])
    # MDT (ResNet-token) default architecture, d=512, 4+6 blocks
    cfg, arch = configs.mdt_default(), "mdt"
    model = build_reference(cfg, arch, seed=33, profile="rich")
    res = run_ddim(model, cfg, arch, B=8, n_steps=10, modality="vis", in_seed=3, record=True)
    meta = dict(config="mdt_default", arch=arch, B=8, n_steps=10, modality="vis", weight_seed=33, profile="rich",
                input_seed=3, sigma_min=0.001, sigma_max=80.0)
    save("g3_b8_mdt.npz", meta, actions=res["actions"], sigmas=res["sigmas"], ctx=res["ctx"])


def g4():
    cfg, arch = configs.mdtv_default(), "mdtv"
    model = build_reference(cfg, arch, seed=41, profile="rich")
    B = 32
    inp = synthetic.sampler_inputs(B, cfg, 42, arch)
    li = synthetic.loss_inputs(B, cfg, 43)
    state = make_state(inp, arch, "vis")
    with torch.no_grad():
        loss, mo = model.loss(state, torch.from_numpy(li["actions"]), torch.from_numpy(inp["goal"]),
                              torch.from_numpy(li["noise_train"]), torch.from_numpy(li["sigma"]))
    meta = dict(config="mdtv_default", arch=arch, B=B, modality="vis", weight_seed=41, profile="rich", input_seed=42,
                loss_seed=43)
    save("g4_loss.npz", meta, loss=np.array(loss.item(), np.float32), model_output=mo.numpy(), sigma=li["sigma"])


G11_CASES = {
    # name: (arch, config factory name, overrides, B)
    "mdtv_tiny": ("mdtv", "mdtv_tiny", {}, 6),
    "mdt_tiny": ("mdt", "mdt_tiny", {}, 5),
    "mdtv_bias_plain_goal": ("mdtv", "mdtv_tiny", dict(bias=True, use_mlp_goal=False, use_modality_encoder=False), 4),
    "mdtv_default": ("mdtv", "mdtv_default", {}, 8),
    # RoPE rotates 32 features: head_dim must be >= 32 (position_embeddings.py:66)
    "mdtv_rope": ("mdtv", "mdtv_tiny", dict(use_rot_embed=True, n_heads=4), 5),
    "mdt_rope": ("mdt", "mdt_tiny", dict(use_rot_embed=True, n_heads=2), 4),
    # conditioning variants of the decoder (SURVEY.md 8(f) items 1 x 3)
    "mdtv_noise_block": ("mdtv", "mdtv_tiny", dict(use_noise_encoder=True), 5),
    "mdtv_no_ada": ("mdtv", "mdtv_tiny", dict(use_ada_conditioning=False), 5),
    "mdt_no_ada": ("mdt", "mdt_tiny", dict(use_ada_conditioning=False, bias=True), 4),
    "mdtv_mlp_head": ("mdtv", "mdtv_tiny", dict(linear_output=False), 5),
    "mdtv_no_goal_cond": ("mdtv", "mdtv_tiny", dict(goal_conditioned=False, use_noise_encoder=True), 5),
    "mdt_no_goal_cond": ("mdt", "mdt_tiny", dict(goal_conditioned=False, use_ada_conditioning=False), 4),
}


def g11():
    """Training step of the reference: loss.backward() through GCDenoiser.loss in eval mode (no dropout) plus an
    extra scalar hung on latent_encoder_emb (the way the MGF / CLA auxiliary losses use it).  Stored per parameter:
    the gradient's l2 norm, sum and its first 6 entries; in full: the gradients of the encoder inputs."""
    for name, (arch, factory, ov, B) in G11_CASES.items():
        cfg = getattr(configs, factory)(**ov)
        model = build_reference(cfg, arch, seed=111, profile="rich")
        inp = synthetic.sampler_inputs(B, cfg, 112, arch)
        li = synthetic.loss_inputs(B, cfg, 113)
        state = make_state(inp, arch, "lang")
        leaves = {k: v.requires_grad_() for k, v in state.items() if torch.is_tensor(v)}
        goal = torch.from_numpy(inp["goal"]).requires_grad_()
        loss, _ = model.loss(state, torch.from_numpy(li["actions"]), goal, torch.from_numpy(li["noise_train"]),
                             torch.from_numpy(li["sigma"]))
        ctx = model.inner_model.latent_encoder_emb
        wctx = torch.from_numpy(synthetic.normal("ctx_weight", tuple(ctx.shape), 114))
        total = loss + 0.1 * (ctx * wctx).sum() / ctx.numel()
        total.backward()
        arrays, summ = {}, {}
        for k, p in model.named_parameters():
            if p.grad is None:
                summ[k] = None
                continue
            g = p.grad.detach().double()
            summ[k] = [float(g.norm()), float(g.sum())] + [float(v) for v in g.flatten()[:6]]
        for k, v in leaves.items():
            arrays["d_" + k] = v.grad.numpy()
        # MDTTransformer with goal_conditioned=False never reads the goal: autograd leaves its gradient None
        arrays["d_goal"] = goal.grad.numpy() if goal.grad is not None else np.zeros_like(inp["goal"])
        meta = dict(config=factory, overrides=ov, arch=arch, B=B, modality="lang", weight_seed=111, profile="rich",
                    input_seed=112, loss_seed=113, ctx_seed=114, grads=summ,
                    state_dict=[[k, list(v.shape)] for k, v in model.state_dict().items()])
        save(f"g11_grads_{name}.npz", meta, loss=np.array(loss.item(), np.float32), **arrays)


def g13():
    """Dropout statistics of the REFERENCE in train() mode (attn 0.3 / resid 0.1 / mlp 0.05): the random streams of
    two implementations cannot coincide, their distributions must.  400 seeded forward passes of GCDenoiser.loss on
    fixed weights / inputs: mean and std of the loss, mean and variance (over seeds) of the model output.
    Second fixture: MDT with only embed_pdrob (self.drop on goal / state / action embeddings) and goal_drop (mask_cond)."""
    _g13("g13_dropout_stats.npz", "mdtv_tiny", {}, "mdtv")
    _g13("g13_embed_goal_drop_stats.npz", "mdt_tiny",
         dict(embed_pdrob=0.2, goal_drop=0.15, attn_pdrop=0.0, resid_pdrop=0.0, mlp_pdrop=0.0), "mdt")


def _g13(fname, factory, ov, arch):
    cfg, B, n = getattr(configs, factory)(**ov), 6, 400
    model = build_reference(cfg, arch, seed=131, profile="rich").train()
    inp = synthetic.sampler_inputs(B, cfg, 132, arch)
    li = synthetic.loss_inputs(B, cfg, 133)
    state = make_state(inp, arch, "lang")
    args = (torch.from_numpy(li["actions"]), torch.from_numpy(inp["goal"]), torch.from_numpy(li["noise_train"]),
            torch.from_numpy(li["sigma"]))
    losses, outs = [], []
    with torch.no_grad():
        for i in range(n):
            torch.manual_seed(1000 + i)
            loss, mo = model.loss(state, *args)
            losses.append(loss.item()); outs.append(mo)
        model.eval()
        loss_eval, mo_eval = model.loss(state, *args)
    outs = torch.stack(outs).double()
    meta = dict(config=factory, overrides=ov, arch=arch, B=B, n=n, modality="lang", weight_seed=131, profile="rich", input_seed=132,
                loss_seed=133, loss_mean=float(np.mean(losses)), loss_std=float(np.std(losses, ddof=1)),
                loss_eval=float(loss_eval), out_var_mean=float(outs.var(0, unbiased=True).mean()),
                out_abs_dev_from_eval=float((outs.mean(0) - mo_eval.double()).abs().mean()))
    save(fname, meta, out_mean=outs.mean(0).float().numpy(), out_var=outs.var(0, unbiased=True).float().numpy())


def g5():
    from mdt.models.edm_diffusion import gc_sampling as gs

    arrays = {}
    for smin in (0.001, 1.0):
        for n in (1, 3, 5, 10, 20):
            s = gs.get_sigmas_exponential(n, smin, 80.0)
            arrays[f"exp_{smin}_{n}"] = s.numpy()
            ratios, coefs = [], []
            for i in range(n):
                t, tn = s[i].log().neg(), s[i + 1].log().neg()
                ratios.append((tn.neg().exp() / t.neg().exp()).item())
                coefs.append((-(-(tn - t)).expm1()).item())
            arrays[f"ddim_ratio_{smin}_{n}"] = np.array(ratios, np.float32)
            arrays[f"ddim_coef_{smin}_{n}"] = np.array(coefs, np.float32)
    arrays["karras_10"] = gs.get_sigmas_karras(10, 0.001, 80.0, 7.0).numpy()
    arrays["linear_10"] = gs.get_sigmas_linear(10, 0.001, 80.0).numpy()
    arrays["ve_10"] = gs.get_sigmas_ve(10, 0.001, 80.0).numpy()
    arrays["vp_10"] = gs.get_sigmas_vp(10).numpy()
    save("g5_schedules.npz", dict(what="noise schedules + DDIM scalar pairs"), **arrays)


def g6():
    cfg, arch = configs.mdtv_default(use_rot_embed=True), "mdtv"
    model = build_reference(cfg, arch, seed=61, profile="rich")
    res = run_ddim(model, cfg, arch, B=4, n_steps=10, modality="lang", in_seed=62, record=True)
    meta = dict(config="mdtv_default", overrides=dict(use_rot_embed=True), arch=arch, B=4, n_steps=10,
                modality="lang", weight_seed=61, profile="rich", input_seed=62, sigma_min=0.001, sigma_max=80.0)
    save("g6_rope.npz", meta, **res)


def g7():
    from mdt.models.edm_diffusion import gc_sampling as gs

    cfg, arch = configs.mdtv_default(), "mdtv"
    model = build_reference(cfg, arch, seed=71, profile="rich")
    inp = synthetic.sampler_inputs(4, cfg, 72, arch)
    state = make_state(inp, arch, "lang")
    goal = torch.from_numpy(inp["goal"])
    out = {}
    for sched, sigmas in (("exp", gs.get_sigmas_exponential(10, 0.001, 80.0)),
                          ("karras", gs.get_sigmas_karras(10, 0.001, 80.0, 7.0))):
        for name in ("euler", "heun", "dpmpp_2m", "ddim"):
            x = torch.from_numpy(inp["noise"]) * 80.0
            out[f"{name}_{sched}"] = getattr(gs, "sample_" + name)(model, state, x, goal, sigmas, disable=True).numpy()
    meta = dict(config="mdtv_default", arch=arch, B=4, n_steps=10, modality="lang", weight_seed=71, profile="rich",
                input_seed=72, sigma_min=0.001, sigma_max=80.0)
    save("g7_samplers.npz", meta, **out)
    # the rest of sample_loop's dispatch table (mdtv_agent.py:619-655) that runs in the reference; stochastic ones
    # once deterministic (eta = 0) and once with the CPU generator seeded right before the call
    out = {}
    sigmas = gs.get_sigmas_exponential(10, 0.001, 80.0)
    x0 = torch.from_numpy(inp["noise"]) * 80.0
    for name in ("lms", "dpm_2", "dpmpp_2_with_lms", "dpmpp_2s"):
        out[name] = getattr(gs, "sample_" + name)(model, state, x0.clone(), goal, sigmas, disable=True).numpy()
    for name in ("euler_ancestral", "dpm_2_ancestral", "dpmpp_2s_ancestral"):
        out[name + "_eta0"] = getattr(gs, "sample_" + name)(model, state, x0.clone(), goal, sigmas, disable=True, eta=0.).numpy()
        torch.manual_seed(1234)
        out[name + "_seed1234"] = getattr(gs, "sample_" + name)(model, state, x0.clone(), goal, sigmas, disable=True).numpy()
    torch.manual_seed(1234)
    out["euler_churn_seed1234"] = gs.sample_euler(model, state, x0.clone(), goal, sigmas, disable=True, s_churn=4.).numpy()
    save("g7b_samplers.npz", meta, **out)
    # DPM-Solver-fast / DPM-Solver++ SDE.  sample_dpm_fast only runs in the reference with an explicit noise sampler
    # (its default reads an undefined name, gc_sampling.py:600), sample_dpmpp_sde's default needs torchsde: both get a
    # deterministic stand-in sampler returning a fixed N(0,1) tensor.  sample_dpm_adaptive cannot run in the
    # reference at all (gc_sampling.py:633 reads noise_sampler before assignment): no golden for it.
    out = {}
    fixed = torch.from_numpy(synthetic.normal("sde_noise", tuple(x0.shape), 73))
    ns = lambda s0, s1: fixed
    solver = gs.DPMSolver(model)
    for nfe in (9, 10, 11):
        out[f"dpm_fast_nfe{nfe}"] = solver.dpm_solver_fast(state, x0.clone(), goal, solver.t(torch.tensor(80.0)),
                                                           solver.t(torch.tensor(0.001)), nfe, 0., 1., ns).detach().numpy()
    with torch.no_grad():
        out["dpmpp_sde_eta0"] = gs.sample_dpmpp_sde(model, state, x0.clone(), goal, sigmas, disable=True, eta=0.,
                                                    noise_sampler=ns).numpy()
        out["dpmpp_sde_eta1_fixednoise"] = gs.sample_dpmpp_sde(model, state, x0.clone(), goal, sigmas, disable=True, eta=1.,
                                                               noise_sampler=ns).numpy()
    out["iddpm_10"] = gs.get_iddpm_sigmas(10, 0.001, 80.0).numpy()
    out["iddpm_20_default"] = gs.get_iddpm_sigmas(20).numpy()
    save("g7c_samplers.npz", dict(meta, noise_seed=73), **out)


G8_VARIANTS = {
    # name: (arch, config overrides) -- constructor fields of the boundary that the default configs do not exercise
    "bias": ("mdtv", dict(bias=True)),
    "plain_goal": ("mdtv", dict(use_mlp_goal=False, use_modality_encoder=False)),
    "two_tokens": ("mdtv", dict(n_obs_token=2, action_seq_len=7, action_dim=5, n_heads=4)),
    "mdt_bias_nopos": ("mdt", dict(bias=True, use_abs_pos_emb=False, n_heads=4)),
    # conditioning variants of the decoder
    "no_ada": ("mdtv", dict(use_ada_conditioning=False)),
    "noise_block": ("mdtv", dict(use_noise_encoder=True)),
    "mdt_no_ada": ("mdt", dict(use_ada_conditioning=False)),
    # the MLP action head (linear_output=False: Linear(d, 100) -> GELU -> Linear(100, A))
    # goal_conditioned=False: MDT-V appends the goal token behind the state tokens, MDT (only runnable without adaLN
    # conditioning) has no goal token
    "no_goal_cond": ("mdtv", dict(goal_conditioned=False)),
    "mdt_no_goal_cond": ("mdt", dict(goal_conditioned=False, use_ada_conditioning=False)),
    "mlp_head": ("mdtv", dict(linear_output=False)),
    "mdt_mlp_head": ("mdt", dict(linear_output=False, bias=True)),
}


def g8():
    for name, (arch, ov) in G8_VARIANTS.items():
        cfg = (configs.mdtv_tiny if arch == "mdtv" else configs.mdt_tiny)(**ov)
        model = build_reference(cfg, arch, seed=81, profile="rich")
        res = run_ddim(model, cfg, arch, B=3, n_steps=3, modality="lang", in_seed=82, record=True)
        meta = dict(config=f"{arch}_tiny", overrides=ov, arch=arch, B=3, n_steps=3, modality="lang", weight_seed=81,
                    profile="rich", input_seed=82, sigma_min=0.001, sigma_max=80.0,
                    state_dict=[[k, list(v.shape)] for k, v in model.state_dict().items()])
        save(f"g8_{name}.npz", meta, **res)


G9_PERCEIVER = {
    # name: (constructor kwargs, (B, T, n), masked)
    "default": (dict(dim=384, depth=6, dim_head=64, heads=8, num_latents=3, num_time_embeds=1), (2, 1, 392), False),
    "tiny_masked": (dict(dim=64, depth=2, dim_head=16, heads=4, num_latents=5, num_time_embeds=4), (3, 3, 7), True),
    "many_latents": (dict(dim=128, depth=1, dim_head=32, heads=2, num_latents=16, num_time_embeds=2, ff_mult=2),
                     (2, 2, 150), False),
}


def g9():
    """Perceiver resampler (perceiver_resampler.py) on synthetic media tokens."""
    from mdt.models.networks.transformers.perceiver_resampler import PerceiverResampler

    for name, (kw, (B, T, n), masked) in G9_PERCEIVER.items():
        m = PerceiverResampler(**kw).eval()
        sd = m.state_dict()
        new = synthetic.fill_state_dict([(k, tuple(v.shape)) for k, v in sd.items()], 91, "rich")
        m.load_state_dict({k: torch.from_numpy(v) for k, v in new.items()}, strict=True)
        x = torch.from_numpy(synthetic.normal("media", (B, T, n, kw["dim"]), 92))
        mask = None
        if masked:
            mask = torch.tensor([[True, True, False], [True, False, True], [True, True, True]])[:B, :T]
        x.requires_grad_()
        out = m(x, mask)
        # training: gradients of every parameter and of the media tokens for a fixed random cotangent
        cot = torch.from_numpy(synthetic.normal("cotangent", tuple(out.shape), 93))
        (out * cot).sum().backward()
        grads = {}
        for k, p in m.named_parameters():
            g = p.grad.detach().double()
            grads[k] = [float(g.norm()), float(g.sum())] + [float(v) for v in g.flatten()[:6]]
        dx = x.grad.detach()
        meta = dict(kwargs=kw, B=B, T=T, n=n, masked=masked, weight_seed=91, input_seed=92, profile="rich",
                    mask=None if mask is None else mask.int().tolist(), cot_seed=93, grads=grads,
                    d_x_summary=[float(dx.double().norm()), float(dx.double().sum())],
                    state_dict=[[k, list(v.shape)] for k, v in sd.items()])
        save(f"g9_perceiver_{name}.npz", meta, out=out.detach().numpy(), d_x_head=dx[:, :, :4, :].numpy())


def g10():
    """String hashes of the harness (pyhash.fnv1_32 call sites) from the REFERENCE's fnv_32_buf, built by
    oracle/Makefile from pyhash-0.9.3/src/fnv/hash_32.c into oracle/_ref/.  str -> UTF-16 code units, as pyhash's
    Hash.h:241-268 feeds them."""
    from oracle import fnv_oracle as FO

    assert FO.reference_available(), "run `make -C oracle` first"
    strings = [str(i) for i in list(range(0, 40)) + [99, 100, 12345, 987654321]]
    strings += [str({"led": 0, "lightbulb": 1, "slider": "left", "drawer": "open"}.values()), "", "window", "\u00e9t\u00e9",
                "\u6f22\u5b57", "a" * 300]
    vec = [{"s": s, "seed": seed, "h": FO.reference_fnv_32_buf(FO.str_bytes(s), seed)}
           for s in strings for seed in (0, 1, 0x811C9DC5)]
    with open(os.path.join(HERE, "g10_fnv.json"), "w") as f:
        json.dump({"source": "fnv_32_buf of pyhash-0.9.3/src/fnv/hash_32.c (oracle/_ref/libfnv_ref.so), UTF-16 input",
                   "vectors": vec}, f, indent=0)
    print(f"  wrote g10_fnv.json: {len(vec)} vectors")


G14_CLA = {
    # name: (ClipStyleProjection kwargs, (B, N tokens))
    "map_tiny": (dict(clip_style="map", token_dim=128, clip_token_index=1, num_token=4), (5, 4)),
    "map_default": (dict(clip_style="map", token_dim=384, clip_token_index=1, num_token=4), (6, 4)),
    "map_state_only": (dict(clip_style="map_state_only", token_dim=128, clip_token_index=1, num_token=4), (4, 4)),
    "map_five_tokens": (dict(clip_style="map", token_dim=128, clip_token_index=1, num_token=5), (3, 5)),
    "mean_pooling": (dict(clip_style="mean_pooling", token_dim=128), (4, 4)),
    "mlp": (dict(clip_style="mlp", token_dim=128, num_token=4), (4, 4)),
    "single_token": (dict(clip_style="single_token", token_dim=128, clip_token_index=1), (4, 4)),
}


def install_agent_stubs():
    """mdtv_agent.py imports the training harness (Lightning, wandb, the Voltron encoder) at module level; none of it
    takes part in clip_auxiliary_loss, which is called UNBOUND below with an object that only carries logit_scale."""
    from unittest import mock

    class Auto(types.ModuleType):
        def __getattr__(self, k):
            if k.startswith("__"):
                raise AttributeError(k)
            return mock.MagicMock()

    def stub(name, **kw):
        m = Auto(name)
        m.__dict__.update(kw)
        m.__path__ = []
        sys.modules[name] = m

    stub("pytorch_lightning", LightningModule=torch.nn.Module, Callback=object)
    stub("pytorch_lightning.utilities", rank_zero_info=print, rank_zero_only=lambda f: f)
    stub("pytorch_lightning.utilities.exceptions", MisconfigurationException=Exception)
    stub("pytorch_lightning.utilities.types", STEP_OUTPUT=object)
    stub("wandb")
    stub("mdt.models.perceptual_encoders.voltron_encoder")


def g14():
    """Contrastive (CLA) head: ClipStyleProjection outputs + gradients for every clip style, and the agent's own
    clip_auxiliary_loss (values and gradients) in its three modes."""
    from mdt.models.networks.transformers.transformer_blocks import ClipStyleProjection

    for name, (kw, (B, N)) in G14_CLA.items():
        m = ClipStyleProjection(**kw)
        sd = m.state_dict()
        new = synthetic.fill_state_dict([(k, tuple(v.shape)) for k, v in sd.items()], 141, "rich")
        m.load_state_dict({k: torch.from_numpy(v) for k, v in new.items()}, strict=True)
        x = torch.from_numpy(synthetic.normal("ctx", (B, N, kw["token_dim"]), 142)).requires_grad_()
        out = m(x)
        cot = torch.from_numpy(synthetic.normal("cotangent", tuple(out.shape), 143))
        (out * cot).sum().backward()
        grads = {}
        for k, p in m.named_parameters():
            g = p.grad.detach().double()
            grads[k] = [float(g.norm()), float(g.sum())] + [float(v) for v in g.flatten()[:6]]
        meta = dict(kwargs=kw, B=B, N=N, weight_seed=141, input_seed=142, cot_seed=143, profile="rich", grads=grads,
                    state_dict=[[k, list(v.shape)] for k, v in sd.items()])
        save(f"g14_cla_{name}.npz", meta, out=out.detach().numpy(), d_x=x.grad.numpy())

    install_agent_stubs()
    from mdt.models.mdtv_agent import MDTVAgent

    class Holder:
        pass

    arrays, meta = {}, dict(input_seed=145, cases=[])
    for B, D, ls in ((7, 128, float(np.log(1 / 0.07))), (16, 384, 1.3), (1, 64, 0.0)):
        for mode in ("symmetric", "img_to_text", "text_to_img"):
            h = Holder()
            h.logit_scale = torch.tensor(ls, dtype=torch.float32, requires_grad=True)
            img = torch.from_numpy(synthetic.normal("img", (B, D), 145)).requires_grad_()
            lang = torch.from_numpy(synthetic.normal("lang", (B, D), 146) + 0.5 * synthetic.normal("img", (B, D), 145)).requires_grad_()
            loss = MDTVAgent.clip_auxiliary_loss(h, img, lang, mode=mode)
            loss.backward()
            key = f"B{B}_D{D}_{mode}"
            arrays[key + "_loss"] = np.array(loss.item(), np.float32)
            arrays[key + "_d_img"], arrays[key + "_d_lang"] = img.grad.numpy(), lang.grad.numpy()
            arrays[key + "_d_scale"] = np.array(h.logit_scale.grad.item(), np.float32)
            meta["cases"].append(dict(key=key, B=B, D=D, logit_scale=ls, mode=mode))
    save("g14_cla_infonce.npz", meta, **arrays)


G15_MAE = {
    # the shipped head (conf/model/img_gen/masked_transformer.yaml with gen_img_res = 112) and a small one
    "default": (dict(resolution=112, patch_size=16, decoder_depth=6, decoder_embed_dim=192, decoder_n_heads=8, context_dim=384,
                     mlp_ratio=4, in_channels=3, norm_pixel_loss=True, num_images=2, mask_ratio=0.75, symmetric_mask=True,
                     img_gen_frame_diff=3), 3, 4),
    "tiny": (dict(resolution=64, patch_size=16, decoder_depth=2, decoder_embed_dim=64, decoder_n_heads=4, context_dim=128,
                  mlp_ratio=4, in_channels=3, norm_pixel_loss=True, num_images=2, mask_ratio=0.5, symmetric_mask=True,
                  img_gen_frame_diff=3), 5, 4),
    # the non-default masking branch as the reference writes it (masked_transformer_decoder.py:158-170, 236-248, 256-258)
    "tiny_asym": (dict(resolution=64, patch_size=16, decoder_depth=2, decoder_embed_dim=64, decoder_n_heads=4, context_dim=128,
                       mlp_ratio=4, in_channels=3, norm_pixel_loss=True, num_images=2, mask_ratio=0.5, symmetric_mask=False,
                       img_gen_frame_diff=3), 5, 4),
}


def install_voltron_stand_in():
    """The decoder imports its transformer blocks from voltron-robotics, which is NOT vendored with the reference and not
    installed here (requirements.txt:20, unpinned).  For this golden run the missing classes are stood in for by the
    parameter tree of the published Voltron Block / RMSNorm evaluated with oracle/mae_oracle.py's restatement: what
    g15 pins is therefore the REFERENCE'S OWN code around the blocks (projection, patch embedding, masking, token assembly,
    position embeddings, loss); the block internals stay parity-unpinned."""
    from mdt_policy_amd.models.img_generation import masked_transformer_decoder as F
    from oracle import mae_oracle as O

    class Block(F.Block):
        def forward(self, x, mask=None):
            return O.block(dict(self.named_parameters()), "", x, self.n_heads)

    class RMSNorm(F.RMSNorm):
        def forward(self, x):
            return O.rms_norm(x, self.g)

    def stub(name, **kw):
        m = types.ModuleType(name)
        m.__dict__.update(kw)
        m.__path__ = []
        sys.modules[name] = m

    stub("voltron")
    stub("voltron.models")
    stub("voltron.models.util")
    stub("voltron.models.util.transformer", Block=Block, RMSNorm=RMSNorm, PatchEmbed=None, get_2D_position_embeddings=None)
    # import-time only (visualisation helpers of the same file)
    if "torchvision" not in sys.modules:
        stub("torchvision")
        stub("torchvision.transforms")
        stub("torchvision.transforms.functional", normalize=None, to_tensor=None, to_pil_image=None)
    try:
        import PIL  # noqa: F401
    except ImportError:
        stub("PIL", Image=None)


def g15():
    """Masked generative foresight head: the reference's MaskedTransformerImgDecoder.forward + compute_loss (+ gradients)
    around stood-in Voltron blocks (see install_voltron_stand_in)."""
    install_voltron_stand_in()
    from mdt.models.img_generation.masked_transformer_decoder import MaskedTransformerImgDecoder

    for name, (kw, B, Tc) in G15_MAE.items():
        m = MaskedTransformerImgDecoder(**kw)
        sd = m.state_dict()
        new = synthetic.fill_state_dict([(k, tuple(v.shape)) for k, v in sd.items() if k != "decoder_pe"], 151, "rich")
        m.load_state_dict({**{k: torch.from_numpy(v) for k, v in new.items()}, "decoder_pe": sd["decoder_pe"]}, strict=True)
        ctx = torch.from_numpy(synthetic.normal("ctx", (B, Tc, kw["context_dim"]), 152)).requires_grad_()
        img = torch.from_numpy(synthetic.normal("img", (B, 2, 3, kw["resolution"], kw["resolution"]), 153))
        torch.manual_seed(154)
        rec, mask, restore, visible = m(ctx, img)
        loss = m.compute_loss(img, rec, mask, restore)
        loss.backward()
        grads = {}
        for k, p in m.named_parameters():
            if p.grad is None:
                continue
            g = p.grad.detach().double()
            grads[k] = [float(g.norm()), float(g.sum())] + [float(v) for v in g.flatten()[:6]]
        meta = dict(kwargs=kw, B=B, Tc=Tc, weight_seed=151, ctx_seed=152, img_seed=153, profile="rich", grads=grads,
                    state_dict=[[k, list(v.shape)] for k, v in sd.items()], named_parameters=[k for k, _ in m.named_parameters()])
        save(f"g15_mae_{name}.npz", meta, rec=rec.detach().numpy(), mask=mask.numpy(), restore=restore.numpy(),
             visible=visible.detach().numpy(), loss=np.array(loss.item(), np.float32), d_ctx=ctx.grad.numpy(),
             decoder_pe=sd["decoder_pe"].numpy())


G16_PROPRIO = {
    # name: (config factory, overrides, B, ddim steps)
    "tiny": ("mdtv_tiny", {}, 5, 4),
    "default": ("mdtv_default", {}, 4, 10),
    "tiny_no_goal_cond": ("mdtv_tiny", dict(goal_conditioned=False), 3, 3),
    "tiny_no_ada": ("mdtv_tiny", dict(use_ada_conditioning=False), 3, 3),
}


def g16():
    """The proprioceptive token: state['state_obs'] (B, 1, proprio_dim) -> proprio_emb -> one more context token behind
    the state tokens (mdtv_transformer.py:260-266, 284-299).  Per case: one GCDenoiser.forward at per-sample sigmas, the
    DDIM actions + context, and a training step (loss.backward() + a scalar hung on latent_encoder_emb) with the gradient
    summaries of every parameter and the full input gradients (state_obs included)."""
    from mdt.models.edm_diffusion import gc_sampling
    for name, (factory, ov, B, n_steps) in G16_PROPRIO.items():
        cfg, arch = getattr(configs, factory)(**ov), "mdtv"
        model = build_reference(cfg, arch, seed=161, profile="rich")
        inp = synthetic.sampler_inputs(B, cfg, 162, arch)
        li = synthetic.loss_inputs(B, cfg, 163)
        obs = synthetic.normal("state_obs", (B, 1, cfg["proprio_dim"]), 164)
        state = make_state(inp, arch, "lang")
        state["state_obs"] = torch.from_numpy(obs)
        goal = torch.from_numpy(inp["goal"])
        arrays = {}
        with torch.no_grad():
            x = torch.from_numpy(li["actions"]) + torch.from_numpy(li["noise_train"]) * torch.from_numpy(li["sigma"])[:, None, None]
            arrays["denoised"] = model(state, x, goal, torch.from_numpy(li["sigma"])).numpy()
            arrays["ctx_forward"] = model.inner_model.latent_encoder_emb.numpy()
            sigmas = gc_sampling.get_sigmas_exponential(n_steps, 0.001, 80.0)
            arrays["actions"] = gc_sampling.sample_ddim(model, state, torch.from_numpy(inp["noise"]) * 80.0, goal, sigmas,
                                                        disable=True).numpy()
            arrays["sigmas"] = sigmas.numpy()
        leaves = {k: v.requires_grad_() for k, v in state.items() if torch.is_tensor(v)}
        goal = goal.clone().requires_grad_()
        loss, mo = model.loss(state, torch.from_numpy(li["actions"]), goal, torch.from_numpy(li["noise_train"]),
                              torch.from_numpy(li["sigma"]))
        ctx = model.inner_model.latent_encoder_emb
        wctx = torch.from_numpy(synthetic.normal("ctx_weight", tuple(ctx.shape), 165))
        (loss + 0.1 * (ctx * wctx).sum() / ctx.numel()).backward()
        summ = {}
        for k, p in model.named_parameters():
            if p.grad is None:
                summ[k] = None
                continue
            g = p.grad.detach().double()
            summ[k] = [float(g.norm()), float(g.sum())] + [float(v) for v in g.flatten()[:6]]
        for k, v in leaves.items():
            arrays["d_" + k] = v.grad.numpy()
        arrays["d_goal"] = goal.grad.numpy() if goal.grad is not None else np.zeros_like(inp["goal"])
        arrays["model_output"] = mo.detach().numpy()
        meta = dict(config=factory, overrides=ov, arch=arch, B=B, n_steps=n_steps, modality="lang", weight_seed=161,
                    profile="rich", input_seed=162, loss_seed=163, obs_seed=164, ctx_seed=165, grads=summ,
                    state_dict=[[k, list(v.shape)] for k, v in model.state_dict().items()])
        save(f"g16_proprio_{name}.npz", meta, loss=np.array(loss.item(), np.float32), **arrays)


G17_LL = {
    # name: (arch, config factory, overrides, B, sigma_min, sigma_max)
    "mdtv_tiny": ("mdtv", "mdtv_tiny", {}, 3, 0.001, 80.0),
    "mdt_tiny": ("mdt", "mdt_tiny", {}, 2, 0.01, 20.0),
    "mdtv_default": ("mdtv", "mdtv_default", {}, 2, 0.001, 80.0),
}


def scipy_odeint(fn, y0, t, atol, rtol, method):
    """Stand-in for torchdiffeq.odeint (not installed): scipy's Dormand-Prince RK45 over the flattened tuple state, so the
    REFERENCE's own log_likelihood (ode_fn, Hutchinson term, prior term) runs unchanged.  The integrator is therefore not
    the reference's: results agree with torchdiffeq's dopri5 to the requested tolerance, not bit for bit."""
    from scipy.integrate import solve_ivp
    assert method == "dopri5"
    shapes = [tuple(p.shape) for p in y0]
    sizes = [int(np.prod(sh)) for sh in shapes]

    def unpack(flat):
        out, o = [], 0
        for sh, n in zip(shapes, sizes):
            out.append(torch.from_numpy(flat[o:o + n].reshape(sh).astype(np.float32)))
            o += n
        return tuple(out)

    def rhs(tt, flat):
        out = fn(torch.tensor(tt, dtype=torch.float32), unpack(flat))
        return np.concatenate([o.detach().double().numpy().ravel() for o in out])

    flat0 = np.concatenate([p.detach().double().numpy().ravel() for p in y0])
    sol = solve_ivp(rhs, (float(t[0]), float(t[-1])), flat0, method="RK45", rtol=rtol, atol=atol)
    assert sol.success
    end = unpack(sol.y[:, -1])
    return tuple(torch.stack([a, b]) for a, b in zip(y0, end))


def g17():
    """log_likelihood of the REFERENCE (gc_sampling.py:468-490) with the integrator stand-in above.  Stored: the probe
    signs v the run drew (seeded CPU generator), the log-likelihoods, the number of model evaluations."""
    from mdt.models.edm_diffusion import gc_sampling
    gc_sampling.odeint = scipy_odeint
    for name, (arch, factory, ov, B, smin, smax) in G17_LL.items():
        cfg = getattr(configs, factory)(**ov)
        model = build_reference(cfg, arch, seed=171, profile="rich")
        inp = synthetic.sampler_inputs(B, cfg, 172, arch)
        li = synthetic.loss_inputs(B, cfg, 173)
        state = make_state(inp, arch, "lang")
        action = torch.from_numpy(li["actions"])
        torch.manual_seed(174)
        v = torch.randint_like(action, 2) * 2 - 1
        torch.manual_seed(174)
        ll, info = gc_sampling.log_likelihood(model, state, action, torch.from_numpy(inp["goal"]), smin, smax)
        meta = dict(config=factory, overrides=ov, arch=arch, B=B, modality="lang", weight_seed=171, profile="rich",
                    input_seed=172, loss_seed=173, probe_seed=174, sigma_min=smin, sigma_max=smax, fevals=info["fevals"],
                    state_dict=[[k, list(v_.shape)] for k, v_ in model.state_dict().items()])
        save(f"g17_loglik_{name}.npz", meta, ll=ll.numpy(), v=v.numpy())


def manifest():
    """state_dict names + shapes IN ORDER (the checkpoint / positional-EMA contract, evaluation/utils.py:98)."""
    out = {}
    for key, arch, cfg in (("mdtv_default", "mdtv", configs.mdtv_default()),
                           ("mdtv_rope", "mdtv", configs.mdtv_default(use_rot_embed=True)),
                           ("mdtv_tiny", "mdtv", configs.mdtv_tiny()),
                           ("mdt_default", "mdt", configs.mdt_default()),
                           ("mdt_tiny", "mdt", configs.mdt_tiny())):
        m = build_reference(cfg, arch, 0, "init")
        out[key] = {
            "state_dict": [[k, list(v.shape)] for k, v in m.state_dict().items()],
            "named_parameters": [k for k, _ in m.named_parameters()],
        }
    with open(os.path.join(HERE, "state_dict_manifest.json"), "w") as f:
        json.dump(out, f, indent=0)
    print("  wrote state_dict_manifest.json")


if __name__ == "__main__":
    assert os.path.isdir(REF), "this script needs the reference checkout at /root/reference"
    install_stubs()
    torch.manual_seed(0)
    which = sys.argv[1:] or ["g1", "g2", "g3", "g4", "g5", "g6", "g7", "g8", "g9", "g10", "g11", "g13", "g14", "g15", "g16", "g17", "manifest"]
    for w in which:
        print(w)
        globals()[w]()
