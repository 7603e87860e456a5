"""EDM preconditioner facade (reference: mdt/models/edm_diffusion/score_wrappers.py:18-100).

``GCDenoiser(inner_model, sigma_data)`` keeps the reference's constructor (Hydra ``_target_`` with
``_recursive_: false``: ``inner_model`` arrives as a DictConfig and is instantiated here), methods and
state_dict (``inner_model.*``).  ``forward`` / ``loss`` / ``forward_context_only`` each are a single call into
libmdt_hip.so; the preconditioning arithmetic (c_skip, c_out, c_in) runs inside the action-embedding and
action-head kernels.
"""
from __future__ import annotations

import importlib
from contextlib import contextmanager

import torch
from torch import nn

from ..networks._engine import HipScoreNetwork


def _staged_backward() -> bool:
    """One autograd node per stage of the HIP backward (``_autograd.staged_diffusion_loss``) instead of one for all of it:
    MDT_HIP_BWD_STAGES=1 always, 0 never, unset: whenever a process group with more than one rank exists -- the case in which
    gradients that appear block by block let DistributedDataParallel reduce them behind the backward."""
    import os
    v = os.environ.get("MDT_HIP_BWD_STAGES", "auto")
    if v != "auto":
        return v not in ("", "0")
    try:
        import torch.distributed as dist
        return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
    except Exception:  # noqa: BLE001
        return False


def _instantiate(cfg):
    """hydra.utils.instantiate when Hydra is installed, else the same thing for a flat kwargs mapping."""
    if isinstance(cfg, nn.Module):
        return cfg
    try:
        import hydra  # the user's harness has it; the build/GPU images do not
        return hydra.utils.instantiate(cfg)
    except ImportError:
        kwargs = {k: v for k, v in dict(cfg).items() if k not in ("_target_", "_recursive_")}
        module, cls = dict(cfg)["_target_"].rsplit(".", 1)
        return getattr(importlib.import_module(module), cls)(**kwargs)


class GCDenoiser(nn.Module):
    """Karras et al. (2022) preconditioner around the MI355X-native score network."""

    def __init__(self, inner_model, sigma_data=1.):
        super().__init__()
        self.inner_model = _instantiate(inner_model)
        if not isinstance(self.inner_model, HipScoreNetwork):
            raise TypeError("mdt_policy_amd.GCDenoiser wraps mdt_policy_amd score networks "
                            f"(MDTVTransformer / MDTTransformer), got {type(self.inner_model).__name__}")
        self.sigma_data = sigma_data
        self.inner_model._sigma_data = float(sigma_data)
        self._ctx_key = None

    # -- reference API -------------------------------------------------------------------------
    def get_scalings(self, sigma):
        """c_skip, c_out, c_in (reference score_wrappers.py:31-43)."""
        sd = self.sigma_data
        c_skip = sd ** 2 / (sigma ** 2 + sd ** 2)
        c_out = sigma * sd / (sigma ** 2 + sd ** 2) ** 0.5
        c_in = 1 / (sigma ** 2 + sd ** 2) ** 0.5
        return c_skip, c_out, c_in

    def _engine(self, allow_grad: bool = False, state=None):
        self.inner_model._sigma_data = float(self.sigma_data)
        self.inner_model._guard_mode(allow_grad)
        return self.inner_model.hip_engine(float(self.sigma_data), state)

    def _wants_grad(self, *tensors) -> bool:
        return torch.is_grad_enabled() and (any(p.requires_grad for p in self.inner_model.parameters()) or
                                            any(torch.is_tensor(t) and t.requires_grad for t in tensors))

    def _train_inputs(self, eng, state, goal, honour_modality: bool, uncond: bool = False):
        """Encoder inputs as the contiguous fp32 tensors the C ABI takes, still attached to the autograd graph, and
        the parameters THIS forward reads: the others (proprio_emb, MDT-V's pos_emb, the other modality's goal
        embedder) stay out of the graph, so their .grad stays None and DistributedDataParallel's
        find_unused_parameters sees them exactly as it does on the reference modules."""
        im = self.inner_model
        prep = lambda t: t if (t.dtype == torch.float32 and t.is_contiguous() and t.data_ptr() % 16 == 0) \
            else t.float().contiguous().clone()
        if im._arch == "mdtv":
            # the proprioceptive input rides in the second token slot of the C ABI (include/mdt_hip.h: tokens2)
            tok, tok2 = prep(state["state_images"]), (prep(state["state_obs"]) if eng.proprio else None)
            B = tok.shape[0]
        else:
            tok, tok2 = prep(state["static"]), prep(state["gripper"])
            B = tok.shape[0]
        eng._tokens(state)  # shape validation
        g = prep(im._goals(goal, uncond))  # uncond: the goal is zeroed (preprocess_goals, mdtv_transformer.py:256-257)
        eng._goal(g, B)
        eng.train_prepare()
        unused = eng.unused_goal_embedder(state, honour_modality)
        named = [(n, p) for n, p in im.named_parameters()
                 if ("inner_model." + n) in eng._grad_layout and not (unused and n.startswith(unused))]
        return tok, tok2, g, B, [n for n, _ in named], [p for _, p in named]

    def forward(self, state, action, goal, sigma, **kwargs):
        """D(x; sigma) = F(x*c_in, sigma)*c_out + x*c_skip (reference score_wrappers.py:65-80)."""
        im = self.inner_model
        eng = self._engine(state=state)
        goal = im._goals(goal, bool(kwargs.get("uncond", False)))
        if self._ctx_key is not None and self._ctx_key == (id(state), id(goal), action.shape[0], eng.ctx_generation):
            return eng.denoise_cached(action, sigma, 0)  # inside cached_context(): encoder hoisted
        # (any other call that re-encoded on this handle since -- an uncond / other-state evaluation inside the block --
        #  bumped eng.ctx_generation: the key no longer matches and this call runs the full forward)
        out, ctx = eng.forward(state, action, goal, sigma)
        im.latent_encoder_emb = ctx
        return out

    def loss(self, state, action, goal, noise, sigma, **kwargs):
        """Denoising score-matching loss, forward value (reference score_wrappers.py:45-63)."""
        im = self.inner_model
        if self._wants_grad(goal, *[v for v in state.values() if torch.is_tensor(v)]):
            # training step: HIP forward with a tape + HIP backward behind torch.autograd
            from ._autograd import HipDiffusionLoss, staged_diffusion_loss
            eng = self._engine(allow_grad=True, state=state)
            tok, tok2, g, B, names, params = self._train_inputs(eng, state, goal, im._arch == "mdtv")
            a, nz = eng._in(action, (B, eng.Ta, eng.A)), eng._in(noise, (B, eng.Ta, eng.A))
            if _staged_backward():  # one autograd node per stage of the backward: gradients appear block by block (DDP overlap)
                eng.train_prepare()
                loss, model_output, ctx = staged_diffusion_loss(eng, state, tok, tok2, g, a, nz, eng._in(sigma, (B,)),
                                                                im.train_dropout(), names, params)
            else:
                loss, model_output, ctx = HipDiffusionLoss.apply(eng, state, tok, tok2, g, a, nz, eng._in(sigma, (B,)),
                                                                 im.train_dropout(), names, *params)
            im.latent_encoder_emb = ctx
            return loss, model_output
        loss, model_output, ctx = self._engine(state=state).loss_fwd(state, action, im._goals(goal, False), noise, sigma)
        im.latent_encoder_emb = ctx
        return loss, model_output

    def forward_context_only(self, state, action, goal, sigma, **kwargs):
        """Encoder tokens only (reference score_wrappers.py:82-97 -> inner_model.forward_enc_only)."""
        im = self.inner_model
        if self._wants_grad(goal, *[v for v in state.values() if torch.is_tensor(v)]):
            from ._autograd import HipContextOnly
            eng = self._engine(allow_grad=True, state=state)
            tok, tok2, g, B, names, params = self._train_inputs(eng, state, goal, True, bool(kwargs.get("uncond", False)))
            sg = eng._in(sigma, (B,)) if eng.sigma_in_context else None  # the sigma token leads the context
            ctx = HipContextOnly.apply(eng, state, tok, tok2, g, True, im.train_dropout(), sg, names, *params)
            if im._arch == "mdtv":
                im.latent_encoder_emb = ctx
            return ctx
        return im.forward_enc_only(state, action, goal, sigma, **kwargs)

    def get_params(self):
        return self.inner_model.parameters()

    # -- additions ---------------------------------------------------------------------------
    @contextmanager
    def cached_context(self, state, goal):
        """Evaluate the (sigma-independent) encoder + cross-attention K/V once and reuse them for every
        ``self(state, x, goal, sigma)`` inside the block -- what the Python samplers in gc_sampling use."""
        im = self.inner_model
        eng = self._engine(state=state)
        if eng.sigma_in_context:  # use_ada_conditioning=False: sigma is a context token, nothing can be hoisted
            yield None
            return
        g = im._goals(goal, False)
        ctx = eng.encode(state, g, honour_modality=im._arch == "mdtv")
        im.latent_encoder_emb = ctx
        B = ctx.shape[0]
        prev = self._ctx_key
        # _goals() returns its argument unchanged for the usual (B,1,G) goal, so id(goal) identifies it
        self._ctx_key = (id(state), id(g), B, eng.ctx_generation) if g is goal else None
        try:
            yield ctx
        finally:
            self._ctx_key = prev

    @torch.no_grad()
    def denoise_vjp(self, state, action, goal, sigma, v):
        """(D(action; sigma), (dD/daction)^T v) in eval-mode arithmetic: what gc_sampling.log_likelihood needs from
        autograd through ``self(state, action, goal, sigma)`` (reference gc_sampling.py:477-484), as one HIP forward
        and an input-gradient-only HIP backward."""
        im = self.inner_model
        return self._engine(state=state).denoise_vjp(state, action, im._goals(goal, False), sigma, v)

    @torch.no_grad()
    def sample_ddim(self, state, action, goal, sigmas):
        """Whole DDIM loop (reference gc_sampling.py:922-951) as one enqueue on the current stream."""
        im = self.inner_model
        out, ctx = self._engine(state=state).sample_ddim(state, action, im._goals(goal, False), sigmas)
        im.latent_encoder_emb = ctx
        return out
