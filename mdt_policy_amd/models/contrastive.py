"""Contrastive latent alignment (CLA) auxiliary loss on the denoiser's context tokens -- SURVEY.md 8(f) item 4.

Reference: MDTVAgent.compute_contrastive_loss (mdt/models/mdtv_agent.py:440-484), clip_extra_forward (:405-409),
clip_auxiliary_loss (:774-799); the same code in MDTAgent.  The pooled embeddings come from
``ClipStyleProjection`` (HIP MAPBlock, models/networks/transformers/map_pool.py) applied to
``model.inner_model.latent_encoder_emb`` (language goal, left by ``model.loss``) and to
``model.forward_context_only`` (vision goal) -- both differentiable HIP forwards, so the loss trains the goal
embedders, the encoder and the pooling head as in the reference.

The InfoNCE itself is a (global batch)^2 x d product and two cross-entropies over the pooled (B, d) embeddings: host
PyTorch (rocBLAS), like the reference; with more than one process the embeddings of all ranks are gathered with
gradients over RCCL first (Lightning's ``all_gather(sync_grads=True)``, mdtv_agent.py:460-466).
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.distributed as dist
import torch.nn.functional as F

from .edm_diffusion.utils import append_dims


def clip_auxiliary_loss(image_features: torch.Tensor, lang_features: torch.Tensor, logit_scale: torch.Tensor,
                        mode: str = "symmetric", lang_text=None) -> torch.Tensor:
    """InfoNCE between pooled vision-goal and language-goal contexts (reference mdtv_agent.py:774-799).
    ``logit_scale`` is the agent's log-temperature parameter (``log(1/0.07)`` initially, :140)."""
    image_features = F.normalize(image_features, dim=-1)
    lang_features = F.normalize(lang_features, dim=-1)
    scale = logit_scale.exp()
    similarity_matrix = scale * image_features @ lang_features.t()
    labels = torch.arange(similarity_matrix.shape[0], device=image_features.device)
    if mode == "symmetric":
        return (F.cross_entropy(similarity_matrix, labels) +
                F.cross_entropy(scale * lang_features @ image_features.t(), labels)) / 2
    if mode == "img_to_text":
        return F.cross_entropy(similarity_matrix, labels)
    if mode == "text_to_img":
        return F.cross_entropy(similarity_matrix.t(), labels)
    raise ValueError("Invalid mode. Expected one of: 'symmetric', 'img_to_text', 'text_to_img'.")


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
