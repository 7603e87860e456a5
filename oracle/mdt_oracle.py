"""CPU ORACLE for the MDT action-denoising hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A clean-room PyTorch-CPU restatement of the reference algorithm, written as pure functions over a
``{state_dict name: tensor}`` mapping.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import this module; the product path
(``mdt_policy_amd`` -> C ABI -> HIP kernels) never does and fails loudly without its HIP library.

Pinning: the reference ships no tests or known-answer vectors for this path, so the
oracle is pinned against outputs of the reference itself, generated in the survey container by
``tests/golden/make_golden.py`` (imports /root/reference on PyTorch-CPU) and committed as
``tests/golden/*.npz``; ``tests/test_oracle_golden.py`` checks every fixture.

Each function cites the reference lines it restates (paths relative to /root/reference).
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
Params = Dict[str, Tensor]


# ----------------------------------------------------------------------------------------------
# noise schedules                                     mdt/models/edm_diffusion/gc_sampling.py:22-88
# ----------------------------------------------------------------------------------------------
def append_zero(x: Tensor) -> Tensor:
    """gc_sampling.py:22-23"""
    return torch.cat([x, x.new_zeros([1])])


def get_sigmas_exponential(n: int, sigma_min: float, sigma_max: float) -> Tensor:
    """gc_sampling.py:35-38  exp(linspace(ln smax, ln smin, n)) ++ [0]"""
    return append_zero(torch.linspace(math.log(sigma_max), math.log(sigma_min), n).exp())


def get_sigmas_karras(n: int, sigma_min: float, sigma_max: float, rho: float = 7.0) -> Tensor:
    """gc_sampling.py:26-32"""
    ramp = torch.linspace(0, 1, n)
    lo, hi = sigma_min ** (1 / rho), sigma_max ** (1 / rho)
    return append_zero((hi + ramp * (lo - hi)) ** rho)


def get_sigmas_linear(n: int, sigma_min: float, sigma_max: float) -> Tensor:
    """gc_sampling.py:41-44"""
    return append_zero(torch.linspace(sigma_max, sigma_min, n))


def get_sigmas_ve(n: int, sigma_min: float = 0.02, sigma_max: float = 100.0) -> Tensor:
    """gc_sampling.py:61-69 (note the reference's linspace(0, n+1, n) quirk)."""
    t = torch.linspace(0, n + 1, n)
    t = (sigma_max ** 2) * ((sigma_min ** 2 / sigma_max ** 2) ** (t / (n - 1)))
    return append_zero(torch.sqrt(t))


def get_sigmas_vp(n: int, beta_d: float = 19.9, beta_min: float = 0.1, eps_s: float = 1e-3) -> Tensor:
    """gc_sampling.py:84-88"""
    t = torch.linspace(1, eps_s, n)
    return append_zero(torch.sqrt(torch.exp(beta_d * t ** 2 / 2 + beta_min * t) - 1))


def ddim_coefficients(sigmas: Tensor):
    """Per-step scalar pairs of sample_ddim (gc_sampling.py:946-950):
    ratio_i = exp(-t_{i+1}) / exp(-t_i), coef_i = -expm1(-(t_{i+1} - t_i)), t = -ln(sigma)."""
    ratios, coefs = [], []
    for i in range(len(sigmas) - 1):
        t, t_next = sigmas[i].log().neg(), sigmas[i + 1].log().neg()
        h = t_next - t
        ratios.append(t_next.neg().exp() / t.neg().exp())
        coefs.append(-(-h).expm1())
    return torch.stack(ratios), torch.stack(coefs)


# ----------------------------------------------------------------------------------------------
# building blocks                       mdt/models/networks/transformers/transformer_blocks.py
# ----------------------------------------------------------------------------------------------
def _lin(P: Params, name: str, x: Tensor) -> Tensor:
    return F.linear(x, P[name + ".weight"], P.get(name + ".bias"))


def _ln(P: Params, name: str, x: Tensor) -> Tensor:
    """transformer_blocks.py:29-38 (bias optional) and nn.LayerNorm ln3 (:205); eps 1e-5."""
    w = P[name + ".weight"]
    return F.layer_norm(x, w.shape, w, P.get(name + ".bias"), 1e-5)


def rotary_tables(n_pos: int, dtype, rot_dim: int = 32, theta: float = 10000.0):
    """position_embeddings.py:83-106,188-200: freqs = theta^(-arange(0,dim,2)/dim), angle = pos*freq,
    each frequency repeated for the (even, odd) pair."""
    freqs = 1.0 / (theta ** (torch.arange(0, rot_dim, 2)[: rot_dim // 2].float() / rot_dim))
    ang = torch.arange(n_pos, dtype=torch.float32)[:, None] * freqs[None, :]
    ang = ang.repeat_interleave(2, dim=-1).to(dtype)
    return ang.cos(), ang.sin()


def apply_rotary(t: Tensor, rot_dim: int) -> Tensor:
    """position_embeddings.py:56-70,138-142: rotate the first rot_dim features of (B,H,T,hd) in
    interleaved pairs by the token's own position 0..T-1."""
    cos, sin = (v.to(t.device) for v in rotary_tables(t.shape[-2], t.dtype, rot_dim))
    tr, rest = t[..., :rot_dim], t[..., rot_dim:]
    x1, x2 = tr[..., 0::2], tr[..., 1::2]
    half = torch.stack((-x2, x1), dim=-1).flatten(-2)
    return torch.cat((tr * cos + half * sin, rest), dim=-1)


def attention(P: Params, pre: str, x: Tensor, ctx: Optional[Tensor], n_heads: int, causal: bool,
              use_rot: bool = False, trace: Optional[dict] = None) -> Tensor:
    """transformer_blocks.py:119-158.  q/k/v Linear with bias, heads split, softmax(q k^T / sqrt(hd) + mask),
    heads merged, c_proj (bias-less when cfg bias=False).  ``causal`` reproduces SDPA's is_causal=True,
    i.e. a TOP-LEFT aligned lower-triangular mask even when Tq != Tk (cross attention, :204 + :142)."""
    B, T, C = x.shape
    src = x if ctx is None else ctx
    hd = C // n_heads
    q = _lin(P, pre + ".query", x).view(B, T, n_heads, hd).transpose(1, 2)
    k = _lin(P, pre + ".key", src).view(B, -1, n_heads, hd).transpose(1, 2)
    v = _lin(P, pre + ".value", src).view(B, -1, n_heads, hd).transpose(1, 2)
    if use_rot:
        rot_dim = max(n_heads // 2, 32)  # transformer_blocks.py:108
        q, k = apply_rotary(q, rot_dim), apply_rotary(k, rot_dim)
    att = (q @ k.transpose(-2, -1)) * (1.0 / math.sqrt(hd))
    if causal:
        Tk = k.shape[-2]
        keep = torch.ones(T, Tk, dtype=torch.bool, device=att.device).tril()
        att = att.masked_fill(~keep, float("-inf"))
    y = att.softmax(dim=-1) @ v
    y = y.transpose(1, 2).reshape(B, T, C)
    return _lin(P, pre + ".c_proj", y)


def mlp(P: Params, pre: str, x: Tensor) -> Tensor:
    """transformer_blocks.py:161-180: c_fc -> exact-erf GELU -> c_proj."""
    return _lin(P, pre + ".c_proj", F.gelu(_lin(P, pre + ".c_fc", x)))


def block(P: Params, pre: str, x: Tensor, n_heads: int, use_rot: bool) -> Tensor:
    """Encoder Block.forward (transformer_blocks.py:209-214), non-causal, no cross attention."""
    x = x + attention(P, pre + ".attn", _ln(P, pre + ".ln_1", x), None, n_heads, False, use_rot)
    x = x + mlp(P, pre + ".mlp", _ln(P, pre + ".ln_2", x))
    return x


def conditioned_block(P: Params, pre: str, x: Tensor, c: Tensor, ctx: Tensor, n_heads: int, use_rot: bool,
                      trace: Optional[dict] = None) -> Tensor:
    """ConditionedBlock.forward (transformer_blocks.py:291-309) with AdaLNZero (:245-260) and
    modulate = shift + x*scale (:262, no '1 +')."""
    mod = F.linear(F.silu(c), P[pre + ".adaLN_zero.modulation.1.weight"], P[pre + ".adaLN_zero.modulation.1.bias"])
    sh1, sc1, g1, sh2, sc2, g2 = mod.chunk(6, dim=-1)
    a = attention(P, pre + ".attn", sh1 + _ln(P, pre + ".ln_1", x) * sc1, None, n_heads, True, use_rot)
    x = x + g1 * a
    xa = attention(P, pre + ".cross_att", _ln(P, pre + ".ln3", x), ctx, n_heads, True, use_rot)
    x = x + xa
    m = mlp(P, pre + ".mlp", sh2 + _ln(P, pre + ".ln_2", x) * sc2)
    if trace is not None:
        trace[pre + ".mod"] = mod
        trace[pre + ".attn"] = a
        trace[pre + ".cross_att"] = xa
        trace[pre + ".mlp"] = m
    x = x + g2 * m
    if trace is not None:
        trace[pre] = x
    return x


def noise_block(P: Params, pre: str, x: Tensor, c: Tensor, ctx: Tensor, n_heads: int, use_rot: bool) -> Tensor:
    """NoiseBlock.forward (transformer_blocks.py:335-341): the sigma embedding is ADDED to the normalised input of
    the self- and cross-attention; the MLP branch is unconditioned; no gates."""
    x = x + attention(P, pre + ".attn", _ln(P, pre + ".ln_1", x) + c, None, n_heads, True, use_rot)
    x = x + attention(P, pre + ".cross_att", _ln(P, pre + ".ln3", x) + c, ctx, n_heads, True, use_rot)
    x = x + mlp(P, pre + ".mlp", _ln(P, pre + ".ln_2", x))
    return x


def plain_decoder_block(P: Params, pre: str, x: Tensor, ctx: Tensor, n_heads: int, use_rot: bool) -> Tensor:
    """Block.forward with cross attention, causal (TransformerDecoder, transformer_blocks.py:209-214,460-506)."""
    x = x + attention(P, pre + ".attn", _ln(P, pre + ".ln_1", x), None, n_heads, True, use_rot)
    x = x + attention(P, pre + ".cross_att", _ln(P, pre + ".ln3", x), ctx, n_heads, True, use_rot)
    x = x + mlp(P, pre + ".mlp", _ln(P, pre + ".ln_2", x))
    return x


# ----------------------------------------------------------------------------------------------
# score networks        mdt/models/networks/mdtv_transformer.py, mdt/models/networks/mdt_transformer.py
# ----------------------------------------------------------------------------------------------
def _goal_embed(P: Params, cfg: dict, goal: Tensor, modality: str, honour_modality: bool) -> Tensor:
    """process_goal_embeddings (mdtv_transformer.py:268-273); lang_emb is goal_emb when
    use_modality_encoder is False (:99-100)."""
    use_lang = honour_modality and cfg.get("use_modality_encoder", False) and modality == "lang"
    name = "inner_model.lang_emb" if use_lang else "inner_model.goal_emb"
    if cfg.get("use_mlp_goal", False):
        return _lin(P, name + ".2", F.gelu(_lin(P, name + ".0", goal)))
    return _lin(P, name, goal)


def _prep_goal(cfg: dict, goal: Tensor, states_len: int) -> Tensor:
    """preprocess_goals, eval mode (mdtv_transformer.py:246-258)."""
    if goal.dim() == 2:
        goal = goal[:, None, :]
    if goal.shape[1] == states_len and cfg["goal_seq_len"] == 1:
        goal = goal[:, :1, :]
    if goal.shape[-1] == 2 * cfg["obs_dim"]:
        goal = goal[:, :, : cfg["obs_dim"]]
    return goal


def encode(P: Params, cfg: dict, state: dict, goal: Tensor, arch: str = "mdtv", entry: str = "forward",
           trace: Optional[dict] = None, sigma: Optional[Tensor] = None) -> Tensor:
    """MDT-V: forward_enc_only (mdtv_transformer.py:213-222).  MDT: enc_only_forward (mdt_transformer.py:211-229,
    always goal_emb) when entry == 'forward', forward_enc_only (:257-281, honours modality) otherwise.
    adaLN variants only (use_ada_conditioning=True): the encoder sees neither sigma nor the actions."""
    ada = cfg.get("use_ada_conditioning", False)
    assert ada or sigma is not None, "without adaLN conditioning the encoder needs sigma (its first token)"
    modality = state.get("modality", "vis")
    H, rot = cfg["n_heads"], cfg.get("use_rot_embed", False)
    if arch == "mdtv":
        tokens = state["state_images"]
        goal = _prep_goal(cfg, goal, tokens.shape[1])
        g = _goal_embed(P, cfg, goal, modality, True)
        s = _lin(P, "inner_model.tok_emb", tokens)
        # concatenate_inputs (:284-299): [goal, state] when goal_conditioned, else [state, drop(goal)]; a
        # proprioceptive 'state_obs' entry (process_state_embeddings :260-266) adds one token behind the state tokens
        # and then an un-conditioned model carries NO goal token at all (:291-294)
        parts = [g, s] if cfg.get("goal_conditioned", True) else [s]
        if "state_obs" in state:
            pe = _lin(P, "inner_model.proprio_emb.0", state["state_obs"].to(s.dtype))
            parts.append(_lin(P, "inner_model.proprio_emb.2", F.mish(pe)))
        elif not cfg.get("goal_conditioned", True):
            parts.append(g)
        h = torch.cat(parts, dim=1)
    else:
        goal = _prep_goal(cfg, goal, 1 if entry == "forward" else state["static"].shape[1])
        g = _goal_embed(P, cfg, goal, modality, entry != "forward")
        wdt = P["inner_model.tok_emb.weight"].dtype  # the reference casts with .float(); fp64 only for gradient checks
        st = _lin(P, "inner_model.tok_emb", state["static"].to(wdt))
        gr = _lin(P, "inner_model.incam_embed", state["gripper"].to(wdt))
        s = torch.stack((st, gr), dim=2).reshape(st.shape[0], 2, -1)  # mdt_transformer.py:300-307
        if cfg.get("use_abs_pos_emb", True):  # apply_position_embeddings (:309-315), t = 1
            pos = P["inner_model.pos_emb"]
            g = g + pos[:, : cfg["goal_seq_len"], :]
            s = s + pos[:, cfg["goal_seq_len"]: cfg["goal_seq_len"] + 1, :]
        # concatenate_inputs (mdt_transformer.py:326-334): without goal conditioning the goal never enters
        h = torch.cat([g, s], dim=1) if cfg.get("goal_conditioned", True) else s
    if not ada:  # concatenate_inputs: the sigma embedding is the FIRST encoder token (mdtv_transformer.py:296-297)
        h = torch.cat([sigma_embedding(P, cfg, sigma), h], dim=1)
    if trace is not None:
        trace["goal_embed"], trace["state_embed"] = g, s
    for l in range(cfg["n_enc_layers"]):
        h = block(P, f"inner_model.encoder.blocks.{l}", h, H, rot)
        if trace is not None:
            trace[f"inner_model.encoder.blocks.{l}"] = h
    ctx = _ln(P, "inner_model.encoder.ln", h)
    if trace is not None:
        trace["ctx"] = ctx
    return ctx


def sigma_embedding(P: Params, cfg: dict, sigma: Tensor) -> Tensor:
    """process_sigma_embeddings + SinusoidalPosEmb + sigma_emb MLP (mdtv_transformer.py:13-25,169-174,238-244)."""
    d = cfg["embed_dim"]
    half = d // 2
    s = (sigma.log() / 4)[:, None]
    f = torch.exp(torch.arange(half) * -(math.log(10000) / (half - 1))).to(sigma)  # dtype and device of sigma
    e = s * f[None, :]
    e = torch.cat((e.sin(), e.cos()), dim=-1)
    c = _lin(P, "inner_model.sigma_emb.3", F.mish(_lin(P, "inner_model.sigma_emb.1", e)))
    return c[:, None, :]


def decode(P: Params, cfg: dict, ctx: Tensor, x_in: Tensor, sigma: Tensor, trace: Optional[dict] = None) -> Tensor:
    """forward_dec_only (mdtv_transformer.py:224-236; identical in mdt_transformer.py:231-242):
    no positional embedding on the action tokens, TransformerFiLMDecoder (transformer_blocks.py:509-569)."""
    H, rot = cfg["n_heads"], cfg.get("use_rot_embed", False)
    c = sigma_embedding(P, cfg, sigma)
    y = _lin(P, "inner_model.action_emb", x_in)
    if trace is not None:
        trace["sigma_emb"], trace["action_emb"] = c, y
    for l in range(cfg["n_dec_layers"]):
        pre = f"inner_model.decoder.blocks.{l}"
        if not cfg.get("use_ada_conditioning", False):
            y = plain_decoder_block(P, pre, y, ctx, H, rot)
        elif cfg.get("use_noise_encoder", False):
            y = noise_block(P, pre, y, c, ctx, H, rot)
        else:
            y = conditioned_block(P, pre, y, c, ctx, H, rot, trace)
    y = _ln(P, "inner_model.decoder.ln", y)
    if cfg.get("linear_output", True):
        out = _lin(P, "inner_model.action_pred", y)
    else:
        out = _lin(P, "inner_model.action_pred.2", F.gelu(_lin(P, "inner_model.action_pred.0", y)))
    if trace is not None:
        trace["decoder.ln"], trace["action_pred"] = y, out
    return out


# ----------------------------------------------------------------------------------------------
# EDM preconditioner                               mdt/models/edm_diffusion/score_wrappers.py:31-97
# ----------------------------------------------------------------------------------------------
def get_scalings(sigma: Tensor, sigma_data: float):
    """score_wrappers.py:31-43"""
    c_skip = sigma_data ** 2 / (sigma ** 2 + sigma_data ** 2)
    c_out = sigma * sigma_data / (sigma ** 2 + sigma_data ** 2) ** 0.5
    c_in = 1 / (sigma ** 2 + sigma_data ** 2) ** 0.5
    return c_skip, c_out, c_in


def denoise(P: Params, cfg: dict, state: dict, x: Tensor, goal: Tensor, sigma: Tensor, sigma_data: float = 0.5,
            arch: str = "mdtv", ctx: Optional[Tensor] = None, trace: Optional[dict] = None) -> Tensor:
    """GCDenoiser.forward (score_wrappers.py:65-80); pass ``ctx`` to reuse a hoisted encoder output."""
    c_skip, c_out, c_in = [s[:, None, None] for s in get_scalings(sigma, sigma_data)]
    if ctx is None:
        ctx = encode(P, cfg, state, goal, arch, "forward", trace, sigma=sigma)
    out = decode(P, cfg, ctx, x * c_in, sigma, trace) * c_out + x * c_skip
    if trace is not None:
        trace["denoised"] = out
    return out


def loss(P: Params, cfg: dict, state: dict, action: Tensor, goal: Tensor, noise: Tensor, sigma: Tensor,
         sigma_data: float = 0.5, arch: str = "mdtv"):
    """GCDenoiser.loss (score_wrappers.py:45-63), eval mode (no dropout / goal masking)."""
    c_skip, c_out, c_in = [s[:, None, None] for s in get_scalings(sigma, sigma_data)]
    noised = action + noise * sigma[:, None, None]
    ctx = encode(P, cfg, state, goal, arch, "forward", sigma=sigma)
    model_output = decode(P, cfg, ctx, noised * c_in, sigma)
    target = (action - c_skip * noised) / c_out
    return (model_output - target).pow(2).flatten(1).mean(), model_output


def forward_context_only(P: Params, cfg: dict, state: dict, goal: Tensor, arch: str = "mdtv",
                         sigma: Optional[Tensor] = None) -> Tensor:
    """GCDenoiser.forward_context_only (score_wrappers.py:82-97) -> inner_model.forward_enc_only."""
    return encode(P, cfg, state, goal, arch, "forward_enc_only", sigma=sigma)


# ----------------------------------------------------------------------------------------------
# samplers                                         mdt/models/edm_diffusion/gc_sampling.py
# ----------------------------------------------------------------------------------------------
@torch.no_grad()
def _hoisted(P: Params, cfg: dict, state: dict, goal: Tensor, arch: str, hoist: bool) -> Optional[Tensor]:
    """The encoder output is step independent only when sigma conditions the decoder (use_ada_conditioning);
    otherwise sigma is the first context token and every model call re-runs the encoder, as the reference does."""
    if hoist and cfg.get("use_ada_conditioning", False):
        return encode(P, cfg, state, goal, arch)
    return None


def sample_ddim(P: Params, cfg: dict, state: dict, x: Tensor, goal: Tensor, sigmas: Tensor, sigma_data: float = 0.5,
                arch: str = "mdtv", hoist: bool = False, per_step: Optional[List[Tensor]] = None) -> Tensor:
    """sample_ddim (gc_sampling.py:922-951).  hoist=False re-runs the encoder every step exactly as the
    reference does; hoist=True evaluates it once (sigma-independent with adaLN conditioning)."""
    s_in = x.new_ones([x.shape[0]])
    ctx = _hoisted(P, cfg, state, goal, arch, hoist)
    for i in range(len(sigmas) - 1):
        den = denoise(P, cfg, state, x, goal, sigmas[i] * s_in, sigma_data, arch, ctx)
        if per_step is not None:
            per_step.append(den)
        t, t_next = sigmas[i].log().neg(), sigmas[i + 1].log().neg()
        h = t_next - t
        x = (t_next.neg().exp() / t.neg().exp()) * x - (-h).expm1() * den
    return x


@torch.no_grad()
def sample_euler(P: Params, cfg: dict, state: dict, x: Tensor, goal: Tensor, sigmas: Tensor, sigma_data: float = 0.5,
                 arch: str = "mdtv", hoist: bool = True) -> Tensor:
    """sample_euler with s_churn = 0 (gc_sampling.py:164-209): d = (x - den)/sigma; x += d * (s_next - s)."""
    s_in = x.new_ones([x.shape[0]])
    ctx = _hoisted(P, cfg, state, goal, arch, hoist)
    for i in range(len(sigmas) - 1):
        den = denoise(P, cfg, state, x, goal, sigmas[i] * s_in, sigma_data, arch, ctx)
        d = (x - den) / sigmas[i]
        x = x + d * (sigmas[i + 1] - sigmas[i])
    return x


@torch.no_grad()
def sample_heun(P: Params, cfg: dict, state: dict, x: Tensor, goal: Tensor, sigmas: Tensor, sigma_data: float = 0.5,
                arch: str = "mdtv", hoist: bool = True) -> Tensor:
    """sample_heun with s_churn = 0 (gc_sampling.py:256-312): Euler predictor + trapezoidal corrector,
    plain Euler on the last step (sigma_next == 0)."""
    s_in = x.new_ones([x.shape[0]])
    ctx = _hoisted(P, cfg, state, goal, arch, hoist)
    for i in range(len(sigmas) - 1):
        den = denoise(P, cfg, state, x, goal, sigmas[i] * s_in, sigma_data, arch, ctx)
        d = (x - den) / sigmas[i]
        dt = sigmas[i + 1] - sigmas[i]
        if sigmas[i + 1] == 0:
            x = x + d * dt
        else:
            x2 = x + d * dt
            den2 = denoise(P, cfg, state, x2, goal, sigmas[i + 1] * s_in, sigma_data, arch, ctx)
            d2 = (x2 - den2) / sigmas[i + 1]
            x = x + (d + d2) / 2 * dt
    return x


@torch.no_grad()
def sample_dpmpp_2m(P: Params, cfg: dict, state: dict, x: Tensor, goal: Tensor, sigmas: Tensor,
                    sigma_data: float = 0.5, arch: str = "mdtv", hoist: bool = True) -> Tensor:
    """sample_dpmpp_2m (gc_sampling.py:699-734): DPM-Solver++(2M) multistep."""
    s_in = x.new_ones([x.shape[0]])
    ctx = _hoisted(P, cfg, state, goal, arch, hoist)
    old = None
    for i in range(len(sigmas) - 1):
        den = denoise(P, cfg, state, x, goal, sigmas[i] * s_in, sigma_data, arch, ctx)
        t, t_next = sigmas[i].log().neg(), sigmas[i + 1].log().neg()
        h = t_next - t
        if old is None or sigmas[i + 1] == 0:
            x = (t_next.neg().exp() / t.neg().exp()) * x - (-h).expm1() * den
        else:
            h_last = t - sigmas[i - 1].log().neg()
            r = h_last / h
            den_d = (1 + 1 / (2 * r)) * den - (1 / (2 * r)) * old
            x = (t_next.neg().exp() / t.neg().exp()) * x - (-h).expm1() * den_d
        old = den
    return x


def to_dtype(P: Params, dtype) -> Params:
    return {k: v.to(dtype) for k, v in P.items()}
