"""MDT-V score network facade (reference: mdt/models/networks/mdtv_transformer.py:35-312).

Same constructor kwargs, same ``state_dict`` names/order, same methods
(``forward`` / ``forward_enc_only`` / ``forward_dec_only`` / ``latent_encoder_emb`` / ``get_params``); the
arithmetic runs in libmdt_hip.so.  Parameters the reference registers but never reads on this path
(``pos_emb``, ``proprio_emb``) are kept so checkpoints load unchanged.
"""
from __future__ import annotations

import logging

import torch
import torch.nn as nn

from ... import _lib
from ._engine import HipScoreNetwork
from .transformers.transformer_blocks import TransformerEncoder, TransformerDecoder, TransformerFiLMDecoder

logger = logging.getLogger(__name__)


def _goal_embedder(goal_dim: int, embed_dim: int, use_mlp_goal: bool) -> nn.Module:
    if use_mlp_goal:
        return nn.Sequential(nn.Linear(goal_dim, embed_dim * 2), nn.GELU(), nn.Linear(embed_dim * 2, embed_dim))
    return nn.Linear(goal_dim, embed_dim)


class _SinusoidalSlot(nn.Module):
    """Parameter-free placeholder for index 0 of ``sigma_emb`` (the reference's SinusoidalPosEmb) so that the
    Linear layers keep the state_dict indices 1 and 3."""


class MDTVTransformer(HipScoreNetwork):
    _arch = "mdtv"

    def __init__(self, obs_dim: int, goal_dim: int, device: str, n_obs_token: int, goal_conditioned: bool,
                 action_dim: int, proprio_dim: int, embed_dim: int, embed_pdrob: float, attn_pdrop: float,
                 resid_pdrop: float, mlp_pdrop: float, n_dec_layers: int, n_enc_layers: int, n_heads: int,
                 goal_seq_len: int, obs_seq_len: int, action_seq_len: int, goal_drop: float = 0.1, bias=False,
                 use_mlp_goal: bool = False, use_abs_pos_emb: bool = True, use_rot_embed: bool = False,
                 rotary_xpos: bool = False, linear_output: bool = True, use_ada_conditioning: bool = False,
                 use_noise_encoder: bool = False, use_modality_encoder: bool = False):
        super().__init__()
        self._init_common()
        self.linear_output = bool(linear_output)
        self.device = device
        self.goal_conditioned = goal_conditioned
        self.obs_dim, self.goal_dim, self.embed_dim = obs_dim, goal_dim, embed_dim
        self.n_obs_token = n_obs_token
        self.use_ada_conditioning = use_ada_conditioning
        self.use_noise_encoder = use_noise_encoder
        self.n_heads, self.n_enc_layers, self.n_dec_layers = n_heads, n_enc_layers, n_dec_layers
        self.bias_flag = bool(bias)
        self.use_mlp_goal = use_mlp_goal
        block_size = goal_seq_len + action_seq_len + obs_seq_len * n_obs_token + 2
        seq_size = goal_seq_len + obs_seq_len * n_obs_token + action_seq_len
        self.action_seq_len = action_seq_len
        self.use_modality_encoder = use_modality_encoder
        self._pdrops = (embed_pdrob, attn_pdrop, resid_pdrop, mlp_pdrop, goal_drop)

        # ---- registration order below IS the state_dict / named_parameters contract ----
        self.tok_emb = nn.Linear(obs_dim, embed_dim)
        self.goal_emb = _goal_embedder(goal_dim, embed_dim, use_mlp_goal)
        self.lang_emb = _goal_embedder(goal_dim, embed_dim, use_mlp_goal) if use_modality_encoder else self.goal_emb
        self.pos_emb = nn.Parameter(torch.zeros(1, seq_size, embed_dim))
        self.cond_mask_prob = goal_drop
        self.use_rot_embed = use_rot_embed
        self.use_abs_pos_emb = use_abs_pos_emb
        self.action_dim = action_dim
        self.encoder = TransformerEncoder(embed_dim=embed_dim, n_heads=n_heads, attn_pdrop=attn_pdrop,
                                          resid_pdrop=resid_pdrop, n_layers=n_enc_layers, block_size=block_size,
                                          bias=bias, use_rot_embed=use_rot_embed, rotary_xpos=rotary_xpos,
                                          mlp_pdrop=mlp_pdrop)
        if use_ada_conditioning:
            self.decoder = TransformerFiLMDecoder(embed_dim=embed_dim, n_heads=n_heads, attn_pdrop=attn_pdrop,
                                                  resid_pdrop=resid_pdrop, n_layers=n_dec_layers,
                                                  film_cond_dim=embed_dim, block_size=block_size, bias=bias,
                                                  use_rot_embed=use_rot_embed, rotary_xpos=rotary_xpos,
                                                  mlp_pdrop=mlp_pdrop, use_cross_attention=True,
                                                  use_noise_encoder=use_noise_encoder)
        else:  # sigma enters as the first encoder token instead (concatenate_inputs)
            self.decoder = TransformerDecoder(embed_dim=embed_dim, n_heads=n_heads, attn_pdrop=attn_pdrop,
                                              resid_pdrop=resid_pdrop, n_layers=n_dec_layers, block_size=block_size,
                                              bias=bias, use_rot_embed=use_rot_embed, rotary_xpos=rotary_xpos,
                                              mlp_pdrop=mlp_pdrop, use_cross_attention=True)
        self.proprio_emb = nn.Sequential(nn.Linear(proprio_dim, embed_dim * 2), nn.Mish(),
                                         nn.Linear(embed_dim * 2, embed_dim))
        self.block_size = block_size
        self.goal_seq_len = goal_seq_len
        self.obs_seq_len = obs_seq_len
        self.sigma_emb = nn.Sequential(_SinusoidalSlot(), nn.Linear(embed_dim, embed_dim * 2), nn.Mish(),
                                       nn.Linear(embed_dim * 2, embed_dim))
        self.action_emb = nn.Linear(action_dim, embed_dim)
        if linear_output:
            self.action_pred = nn.Linear(embed_dim, action_dim)
        else:  # reference mdtv_transformer.py:181-185 (parameter holder: the arithmetic runs in the HIP head)
            self.action_pred = nn.Sequential(nn.Linear(embed_dim, 100), nn.GELU(), nn.Linear(100, action_dim))
        self.apply(self._init_weights)

    def _init_weights(self, module):
        """Same initial distributions as the reference (mdtv_transformer.py:197-206)."""
        if isinstance(module, nn.Linear):
            torch.nn.init.normal_(module.weight, mean=0.0, std=0.02)
            if module.bias is not None:
                torch.nn.init.zeros_(module.bias)
        elif isinstance(module, nn.LayerNorm):
            torch.nn.init.zeros_(module.bias)
            torch.nn.init.ones_(module.weight)
        elif isinstance(module, MDTVTransformer):
            torch.nn.init.normal_(module.pos_emb, mean=0.0, std=0.02)

    def _hip_config(self, sigma_data: float, proprio: bool = False) -> _lib.MDTConfig:
        return _lib.MDTConfig(
            proprio_dim=int(self.proprio_emb[0].in_features), use_proprio=int(proprio),
            arch=_lib.ARCH["mdtv"], embed_dim=self.embed_dim, n_heads=self.n_heads, n_enc_layers=self.n_enc_layers,
            n_dec_layers=self.n_dec_layers, action_dim=self.action_dim, obs_dim=self.obs_dim, goal_dim=self.goal_dim,
            n_obs_token=self.n_obs_token, goal_seq_len=self.goal_seq_len, action_seq_len=self.action_seq_len,
            use_mlp_goal=int(self.use_mlp_goal), use_modality_encoder=int(self.use_modality_encoder),
            use_abs_pos_emb=int(self.use_abs_pos_emb), use_rot_embed=int(self.use_rot_embed),
            use_ada_conditioning=int(self.use_ada_conditioning), use_noise_encoder=int(self.use_noise_encoder),
            linear_output=int(self.linear_output), bias=int(self.bias_flag), sigma_data=float(sigma_data),
            no_goal_conditioning=int(not self.goal_conditioned))
