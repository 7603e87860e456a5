"""Parameter containers with the reference's module tree (names AND registration order).

Reference: mdt/models/networks/transformers/transformer_blocks.py (LayerNorm :29, Attention :66, MLP :161,
Block :183, AdaLNZero :245, ConditionedBlock :266, NoiseBlock :311, TransformerEncoder :344, TransformerDecoder :460,
TransformerFiLMDecoder :509).
Checkpoints address parameters by state_dict key and the evaluation harness maps EMA weights POSITIONALLY onto
``named_parameters()`` (mdt/evaluation/utils.py:98), so both are part of the drop-in boundary.  These classes
own parameters only: the arithmetic of the whole tree runs in libmdt_hip.so (gfx950 kernels), driven by the
score-network facade that holds them, so none of them defines ``forward``.
"""
from __future__ import annotations

import torch
import torch.nn as nn


class _ParamOnly(nn.Module):
    def forward(self, *args, **kwargs):  # pragma: no cover - guard
        raise RuntimeError(
            f"{type(self).__name__} is a parameter container; its arithmetic runs inside the fused HIP path of the "
            "owning MDTVTransformer/MDTTransformer and cannot be called on its own")


class LayerNorm(_ParamOnly):
    """Bias-optional LayerNorm parameters (eps 1e-5 in the kernels)."""

    def __init__(self, ndim: int, bias: bool):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(ndim))
        self.bias = nn.Parameter(torch.zeros(ndim)) if bias else None


class RotaryFreqs(_ParamOnly):
    """Holds the (non-trainable) ``freqs`` entry the reference's RotaryEmbedding puts into the state_dict."""

    def __init__(self, dim: int, theta: float = 10000.0):
        super().__init__()
        freqs = 1.0 / (theta ** (torch.arange(0, dim, 2)[: dim // 2].float() / dim))
        self.freqs = nn.Parameter(freqs, requires_grad=False)


class Attention(_ParamOnly):
    def __init__(self, n_embd: int, n_head: int, attn_pdrop: float, resid_pdrop: float, block_size: int,
                 causal: bool = False, bias: bool = False, use_rot_embed: bool = False, rotary_xpos: bool = False):
        super().__init__()
        assert n_embd % n_head == 0
        if rotary_xpos:
            raise NotImplementedError("rotary_xpos=True is not supported by the HIP path")
        self.key = nn.Linear(n_embd, n_embd)
        self.query = nn.Linear(n_embd, n_embd)
        self.value = nn.Linear(n_embd, n_embd)
        self.c_proj = nn.Linear(n_embd, n_embd, bias=bias)
        self.n_head, self.n_embd, self.causal = n_head, n_embd, causal
        self.attn_pdrop, self.resid_pdrop = attn_pdrop, resid_pdrop
        self.use_rot_embed = use_rot_embed
        if use_rot_embed:
            self.rotary_pos_emb = RotaryFreqs(max(n_head // 2, 32))


class MLP(_ParamOnly):
    def __init__(self, n_embd: int, bias: bool, dropout: float = 0):
        super().__init__()
        self.c_fc = nn.Linear(n_embd, 4 * n_embd, bias=bias)
        self.c_proj = nn.Linear(4 * n_embd, n_embd, bias=bias)
        self.pdrop = dropout


class Block(_ParamOnly):
    def __init__(self, n_embd, n_heads, attn_pdrop, resid_pdrop, mlp_pdrop, block_size, causal,
                 use_cross_attention=False, use_rot_embed=False, rotary_xpos=False, bias=False):
        super().__init__()
        self.ln_1 = LayerNorm(n_embd, bias=bias)
        self.attn = Attention(n_embd, n_heads, attn_pdrop, resid_pdrop, block_size, causal, bias, use_rot_embed,
                              rotary_xpos)
        self.use_cross_attention = use_cross_attention
        if use_cross_attention:
            self.cross_att = Attention(n_embd, n_heads, attn_pdrop, resid_pdrop, block_size, causal, bias,
                                       use_rot_embed, rotary_xpos)
            self.ln3 = nn.LayerNorm(n_embd)
        self.ln_2 = LayerNorm(n_embd, bias=bias)
        self.mlp = MLP(n_embd, bias, mlp_pdrop)


class AdaLNZero(_ParamOnly):
    def __init__(self, hidden_size: int):
        super().__init__()
        self.modulation = nn.Sequential(nn.SiLU(), nn.Linear(hidden_size, 6 * hidden_size, bias=True))


class ConditionedBlock(Block):
    def __init__(self, n_embd, n_heads, attn_pdrop, resid_pdrop, mlp_pdrop, block_size, causal, film_cond_dim,
                 use_cross_attention=False, use_rot_embed=False, rotary_xpos=False, bias=False):
        super().__init__(n_embd, n_heads, attn_pdrop, resid_pdrop, mlp_pdrop, block_size, causal,
                         use_cross_attention=use_cross_attention, use_rot_embed=use_rot_embed,
                         rotary_xpos=rotary_xpos, bias=bias)
        self.adaLN_zero = AdaLNZero(film_cond_dim)


class NoiseBlock(Block):
    """Same parameters as Block; the sigma embedding is added to the normalised input of both attentions."""


class TransformerEncoder(_ParamOnly):
    def __init__(self, embed_dim, n_heads, attn_pdrop, resid_pdrop, n_layers, block_size, bias=False,
                 use_rot_embed=False, rotary_xpos=False, mlp_pdrop=0):
        super().__init__()
        self.blocks = nn.Sequential(*[
            Block(embed_dim, n_heads, attn_pdrop, resid_pdrop, mlp_pdrop, block_size, causal=False,
                  use_rot_embed=use_rot_embed, rotary_xpos=rotary_xpos, bias=bias) for _ in range(n_layers)])
        self.ln = LayerNorm(embed_dim, bias)


class TransformerFiLMDecoder(_ParamOnly):
    def __init__(self, embed_dim, n_heads, attn_pdrop, resid_pdrop, n_layers, block_size, film_cond_dim, bias=False,
                 use_rot_embed=False, rotary_xpos=False, mlp_pdrop=0, use_cross_attention=True,
                 use_noise_encoder=False, kwargs=None):
        super().__init__()
        if not use_cross_attention:
            raise NotImplementedError("the HIP decoder always cross-attends to the context")
        if use_noise_encoder:
            self.blocks = nn.Sequential(*[
                NoiseBlock(embed_dim, n_heads, attn_pdrop, resid_pdrop, mlp_pdrop, block_size, causal=True,
                           use_cross_attention=use_cross_attention, use_rot_embed=use_rot_embed,
                           rotary_xpos=rotary_xpos, bias=bias) for _ in range(n_layers)])
        else:
            self.blocks = nn.Sequential(*[
                ConditionedBlock(embed_dim, n_heads, attn_pdrop, resid_pdrop, mlp_pdrop, block_size, causal=True,
                                 use_cross_attention=use_cross_attention, use_rot_embed=use_rot_embed,
                                 rotary_xpos=rotary_xpos, bias=bias, film_cond_dim=film_cond_dim)
                for _ in range(n_layers)])
        self.ln = LayerNorm(embed_dim, bias)


class TransformerDecoder(_ParamOnly):
    """Plain cross-attending decoder of the use_ada_conditioning=False variant."""

    def __init__(self, embed_dim, n_heads, attn_pdrop, resid_pdrop, n_layers, block_size, bias=False,
                 use_rot_embed=False, rotary_xpos=False, mlp_pdrop=0, use_cross_attention=True):
        super().__init__()
        if not use_cross_attention:
            raise NotImplementedError("the HIP decoder always cross-attends to the context")
        self.blocks = nn.Sequential(*[
            Block(embed_dim, n_heads, attn_pdrop, resid_pdrop, mlp_pdrop, block_size, causal=True,
                  use_cross_attention=use_cross_attention, use_rot_embed=use_rot_embed, rotary_xpos=rotary_xpos,
                  bias=bias) for _ in range(n_layers)])
        self.ln = LayerNorm(embed_dim, bias)


# the contrastive head lives next to the blocks in the reference (transformer_blocks.py:42-62, :716-880)
from .map_pool import (ClipStyleProjection, MAPAttention, MAPBlock, MeanPooling, RMSNorm, SwishGLU)  # noqa: E402,F401
