"""MDT (ResNet-token) score network facade (reference: mdt/models/networks/mdt_transformer.py:38-335).

Encoder input = [goal_emb(goal) + pos[0], tok_emb(static) + pos[1], incam_embed(gripper) + pos[1]]; the
decoder is the same adaLN decoder as MDT-V.  ``forward`` always embeds the goal with ``goal_emb``
(reference :215) while ``forward_enc_only`` honours ``states['modality']`` (:280-285).
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn as nn

from ... import _lib
from ._engine import HipScoreNetwork
from .mdtv_transformer import _goal_embedder, _SinusoidalSlot
from .transformers.transformer_blocks import TransformerEncoder, TransformerDecoder, TransformerFiLMDecoder


class MDTTransformer(HipScoreNetwork):
    _arch = "mdt"

    def __init__(self, obs_dim: int, goal_dim: int, device: str, goal_conditioned: bool, action_dim: int,
                 embed_dim: int, embed_pdrob: float, attn_pdrop: float, resid_pdrop: float, mlp_pdrop: float,
                 n_dec_layers: int, n_enc_layers: int, n_heads: int, goal_seq_len: int, obs_seq_len: int,
                 action_seq_len: int, proprio_dim: Optional[int] = None, goal_drop: float = 0.1, bias=False,
                 use_abs_pos_emb: bool = True, use_rot_embed: bool = False, rotary_xpos: bool = False,
                 linear_output: bool = True, use_ada_conditioning: bool = False,
                 use_noise_encoder: bool = False, latent_is_decoder: bool = False,
                 use_modality_encoder: bool = False, use_mlp_goal: bool = False):
        super().__init__()
        self._init_common()
        if not goal_conditioned and use_ada_conditioning:
            # the reference builds but cannot run this: concatenate_inputs cats the absent sigma token (:334)
            raise NotImplementedError("MDTTransformer with goal_conditioned=False needs use_ada_conditioning=False")
        self.linear_output = bool(linear_output)
        self.device = device
        self.goal_conditioned = goal_conditioned
        self.obs_dim, self.goal_dim, self.embed_dim = obs_dim, goal_dim, embed_dim
        self.use_ada_conditioning = use_ada_conditioning
        self.use_noise_encoder = use_noise_encoder
        self.proprio_dim = proprio_dim
        self.latent_is_decoder = latent_is_decoder
        self.n_heads, self.n_enc_layers, self.n_dec_layers = n_heads, n_enc_layers, n_dec_layers
        self.bias_flag = bool(bias)
        self.use_mlp_goal = use_mlp_goal
        block_size = goal_seq_len + action_seq_len + obs_seq_len + 1
        seq_size = goal_seq_len + action_seq_len
        self.action_seq_len = action_seq_len
        self.use_modality_encoder = use_modality_encoder
        self._pdrops = (embed_pdrob, attn_pdrop, resid_pdrop, mlp_pdrop, goal_drop)

        # ---- registration order below IS the state_dict / named_parameters contract ----
        self.tok_emb = nn.Linear(obs_dim, embed_dim)
        self.incam_embed = nn.Linear(obs_dim, embed_dim)
        self.pos_emb = nn.Parameter(torch.zeros(1, seq_size, embed_dim))
        self.cond_mask_prob = goal_drop
        self.use_rot_embed = use_rot_embed
        self.use_abs_pos_emb = use_abs_pos_emb
        self.action_dim = action_dim
        self.goal_emb = _goal_embedder(goal_dim, embed_dim, use_mlp_goal)
        self.lang_emb = _goal_embedder(goal_dim, embed_dim, use_mlp_goal) if use_modality_encoder else self.goal_emb
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
        self.block_size = block_size
        self.goal_seq_len = goal_seq_len
        self.obs_seq_len = obs_seq_len
        self.sigma_emb = nn.Sequential(_SinusoidalSlot(), nn.Linear(embed_dim, embed_dim * 2), nn.Mish(),
                                       nn.Linear(embed_dim * 2, embed_dim))
        self.action_emb = nn.Linear(action_dim, embed_dim)
        if linear_output:
            self.action_pred = nn.Linear(embed_dim, action_dim)
        else:  # reference mdt_transformer.py:173-177: hidden width embed_dim (parameter holder: the arithmetic runs in the HIP head)
            self.action_pred = nn.Sequential(nn.Linear(embed_dim, embed_dim), nn.GELU(), nn.Linear(embed_dim, action_dim))
        if proprio_dim is not None:
            self.proprio_emb = nn.Sequential(nn.Linear(proprio_dim, embed_dim * 2), nn.Mish(),
                                             nn.Linear(embed_dim * 2, embed_dim))
        self.apply(self._init_weights)

    def _init_weights(self, module):
        """Same initial distributions as the reference (mdt_transformer.py:196-205)."""
        if isinstance(module, nn.Linear):
            torch.nn.init.normal_(module.weight, mean=0.0, std=0.02)
            if module.bias is not None:
                torch.nn.init.zeros_(module.bias)
        elif isinstance(module, nn.LayerNorm):
            torch.nn.init.zeros_(module.bias)
            torch.nn.init.ones_(module.weight)
        elif isinstance(module, MDTTransformer):
            torch.nn.init.normal_(module.pos_emb, mean=0.0, std=0.02)

    # reference method names of the "as written" path (mdt_transformer.py:211-242)
    def enc_only_forward(self, states, actions, goals, sigma, uncond: Optional[bool] = False):
        self._guard_mode()
        ctx = self.hip_engine(state=states).encode(states, self._goals(goals, uncond), honour_modality=False)
        self.latent_encoder_emb = ctx
        return ctx

    def dec_only_forward(self, context, actions, sigma):
        return self.forward_dec_only(context, actions, sigma)

    def _hip_config(self, sigma_data: float, proprio: bool = False) -> _lib.MDTConfig:
        return _lib.MDTConfig(
            arch=_lib.ARCH["mdt"], embed_dim=self.embed_dim, n_heads=self.n_heads, n_enc_layers=self.n_enc_layers,
            n_dec_layers=self.n_dec_layers, action_dim=self.action_dim, obs_dim=self.obs_dim, goal_dim=self.goal_dim,
            n_obs_token=2, goal_seq_len=self.goal_seq_len, action_seq_len=self.action_seq_len,
            use_mlp_goal=int(self.use_mlp_goal), use_modality_encoder=int(self.use_modality_encoder),
            use_abs_pos_emb=int(self.use_abs_pos_emb), use_rot_embed=int(self.use_rot_embed),
            use_ada_conditioning=int(self.use_ada_conditioning), use_noise_encoder=int(self.use_noise_encoder),
            linear_output=int(self.linear_output), bias=int(self.bias_flag), sigma_data=float(sigma_data),
            no_goal_conditioning=int(not self.goal_conditioned))
