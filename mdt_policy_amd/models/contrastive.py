"""Contrastive latent alignment (CLA) auxiliary loss on the denoiser's context tokens -- SURVEY.md 8(f) item 4.

Reference: MDTVAgent.compute_contrastive_loss (mdt/models/mdtv_agent.py:440-484), clip_extra_forward (:405-409),
clip_auxiliary_loss (:774-799); the same code in MDTAgent.  The pooled embeddings come from
``ClipStyleProjection`` (HIP MAPBlock, models/networks/transformers/map_pool.py) applied to
``model.inner_model.latent_encoder_emb`` (language goal, left by ``model.loss``) and to
``model.forward_context_only`` (vision goal) -- both differentiable HIP forwards, so the loss trains the goal
embedders, the encoder and the pooling head as in the reference.

The InfoNCE over the pooled (global batch, d) embeddings is one HIP enqueue (``mdt_op_infonce``: normalisation, the
B x B product on the fp32-MFMA GEMM, row / column log-sum-exp, loss, and all three gradients); with more than one
process the embeddings of all ranks are gathered with gradients over RCCL first (Lightning's
``all_gather(sync_grads=True)``, mdtv_agent.py:460-466).  No CPU path: CPU tensors raise.
"""
from __future__ import annotations

from typing import Optional

import ctypes as C

import torch
import torch.distributed as dist

from .. import _lib
from .edm_diffusion.utils import append_dims


class _InfoNCE(torch.autograd.Function):
    """Loss value and the three gradients come out of ONE enqueue; backward scales them by the incoming gradient."""

    @staticmethod
    def forward(ctx, img, lang, logit_scale, mode):
        lib = _lib.load()
        B, D = img.shape
        dev = img.device
        loss = torch.empty((), device=dev, dtype=torch.float32)
        need = any(ctx.needs_input_grad[:3])
        d_img = torch.empty_like(img) if need else None
        d_lang = torch.empty_like(lang) if need else None
        d_ls = torch.empty((), device=dev, dtype=torch.float32) if need else None
        scratch = torch.empty(int(lib.mdt_op_infonce_scratch(B, D)), device=dev, dtype=torch.float32)
        ptr = lambda t: None if t is None else t.data_ptr()
        a = _lib.InfoNCEArgs(image_features=img.data_ptr(), lang_features=lang.data_ptr(), logit_scale=logit_scale.data_ptr(),
                             batch=B, dim=D, mode=mode, loss=loss.data_ptr(), d_image=ptr(d_img), d_lang=ptr(d_lang),
                             d_logit_scale=ptr(d_ls), scratch=scratch.data_ptr())
        _lib.check(lib.mdt_op_infonce(C.byref(a), torch.cuda.current_stream(dev).cuda_stream))
        ctx.save_for_backward(d_img, d_lang, d_ls)
        return loss

    @staticmethod
    def backward(ctx, g):
        d_img, d_lang, d_ls = ctx.saved_tensors
        return g * d_img, g * d_lang, g * d_ls, None


def clip_auxiliary_loss(image_features: torch.Tensor, lang_features: torch.Tensor, logit_scale: torch.Tensor,
                        mode: str = "symmetric", lang_text=None) -> torch.Tensor:
    """InfoNCE between pooled vision-goal and language-goal contexts (reference mdtv_agent.py:774-799).
    ``logit_scale`` is the agent's log-temperature parameter (``log(1/0.07)`` initially, :140)."""
    if mode not in _lib.INFONCE_MODE:
        raise ValueError("Invalid mode. Expected one of: 'symmetric', 'img_to_text', 'text_to_img'.")
    if image_features.device.type != "cuda":
        raise RuntimeError("clip_auxiliary_loss runs only on a ROCm GPU (hand-written gfx950 kernels): there is no CPU "
                           "execution path")
    if image_features.shape != lang_features.shape or image_features.ndim != 2:
        raise ValueError(f"expected two (batch, dim) tensors, got {tuple(image_features.shape)} and {tuple(lang_features.shape)}")
    prep = lambda t: t if (t.dtype == torch.float32 and t.is_contiguous() and t.data_ptr() % 16 == 0) \
        else t.float().contiguous().clone()
    ls = logit_scale.to(device=image_features.device, dtype=torch.float32).reshape(())
    return _InfoNCE.apply(prep(image_features), prep(lang_features), ls, _lib.INFONCE_MODE[mode])


class _AllGatherWithGrad(torch.autograd.Function):
    """Lightning's ``all_gather(sync_grads=True)``: forward gathers every rank's block, backward sums the gradient of
    the gathered tensor over the ranks and hands each rank the slice that belongs to its own block."""

    @staticmethod
    def forward(ctx, x, group):
        ctx.group = group
        x = x.contiguous()
        world = dist.get_world_size(group)
        out = x.new_empty((world * x.shape[0],) + tuple(x.shape[1:]))  # the concatenated form every backend accepts
        dist.all_gather_into_tensor(out, x, group=group)
        return out.view((world,) + tuple(x.shape))

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous()
        dist.all_reduce(g, op=dist.ReduceOp.SUM, group=ctx.group)
        return g[dist.get_rank(ctx.group)], None


def all_gather_with_grad(x: torch.Tensor, group=None) -> torch.Tensor:
    """(B, d) on every rank -> (world, B, d), differentiable; every rank must hold the same B."""
    return _AllGatherWithGrad.apply(x, group)


def clip_extra_forward(model, perceptual_emb, latent_goal, actions, sigmas, noise):
    """Context of the OTHER goal modality for the same states (reference mdtv_agent.py:405-409)."""
    model.train()
    noised_input = actions + noise * append_dims(sigmas, actions.ndim)
    return model.forward_context_only(perceptual_emb, noised_input, latent_goal, sigmas)


def compute_contrastive_loss(model, clip_proj, logit_scale, perceptual_emb, image_latent_goal, actions, sigma, noise,
                             modality_scope: str = "lang", use_distributed_clip: bool = True,
                             clip_loss_type: str = "symmetric", lang_text=None, group=None) -> torch.Tensor:
    """MDTVAgent.compute_contrastive_loss (mdtv_agent.py:440-484): call right after ``model.loss(...)`` of a
    language-goal batch, whose context is ``model.inner_model.latent_encoder_emb``."""
    if "lang" not in modality_scope:
        return torch.tensor(0.0, device=actions.device)
    latent_language_embed = model.inner_model.latent_encoder_emb
    latent_vis_embed = clip_extra_forward(model, perceptual_emb, image_latent_goal, actions, sigma, noise)
    latent_language_embed = clip_proj(latent_language_embed)
    latent_vis_embed = clip_proj(latent_vis_embed)
    if use_distributed_clip and dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        latent_vis_embed = all_gather_with_grad(latent_vis_embed, group).flatten(0, 1)
        latent_language_embed = all_gather_with_grad(latent_language_embed, group).flatten(0, 1)
    return clip_auxiliary_loss(latent_vis_embed, latent_language_embed, logit_scale, mode=clip_loss_type,
                               lang_text=lang_text)
