"""Canonical score-network configurations (constructor kwargs) for the MDT hot path.

Field names are the reference's Hydra constructor contract
(reference: conf/model/model/mdtv_transformer.yaml:6-35, conf/model/model/mdt_transformer.yaml:6-34,
conf/config_d.yaml:22-34).  They are plain dicts so that they can be fed to the facade classes in
``mdt_policy_amd.models`` *and*, inside the survey container only, to the reference classes when the
golden fixtures are generated (tests/golden/make_golden.py).
"""
from __future__ import annotations

import copy

_MDTV_TARGET = "mdt_policy_amd.models.networks.mdtv_transformer.MDTVTransformer"
_MDT_TARGET = "mdt_policy_amd.models.networks.mdt_transformer.MDTTransformer"


def mdtv_default(**overrides) -> dict:
    """MDT-V default: d=384, 4 encoder + 4 adaLN decoder blocks, 8 heads, Te=1+3, Ta=10 (BASELINE C2)."""
    cfg = dict(
        _target_=_MDTV_TARGET,
        action_dim=7,
        obs_dim=384,
        goal_dim=512,
        proprio_dim=8,
        goal_conditioned=True,
        embed_dim=384,
        n_dec_layers=4,
        n_enc_layers=4,
        n_obs_token=3,
        goal_seq_len=1,
        obs_seq_len=1,
        action_seq_len=10,
        embed_pdrob=0,
        goal_drop=0,
        attn_pdrop=0.3,
        resid_pdrop=0.1,
        mlp_pdrop=0.05,
        n_heads=8,
        device="cpu",
        linear_output=True,
        use_rot_embed=False,
        use_abs_pos_emb=True,
        bias=False,
        use_ada_conditioning=True,
        use_noise_encoder=False,
        use_modality_encoder=True,
        use_mlp_goal=True,
    )
    cfg.update(overrides)
    return cfg


def mdtv_tiny(**overrides) -> dict:
    """Tiny MDT-V used for plumbing tests (BASELINE C1 shape: d=128, 2 adaLN blocks, 8 heads -> hd 16)."""
    return mdtv_default(embed_dim=128, obs_dim=128, n_enc_layers=1, n_dec_layers=2, **overrides)


def mdt_default(**overrides) -> dict:
    """MDT (ResNet-token) default: d=512, 4 encoder + 6 decoder blocks (mdt_transformer.yaml)."""
    cfg = dict(
        _target_=_MDT_TARGET,
        action_dim=7,
        obs_dim=512,
        goal_dim=512,
        proprio_dim=8,
        goal_conditioned=True,
        embed_dim=512,
        n_dec_layers=6,
        n_enc_layers=4,
        goal_seq_len=1,
        obs_seq_len=1,
        action_seq_len=10,
        embed_pdrob=0,
        goal_drop=0,
        attn_pdrop=0.3,
        resid_pdrop=0.1,
        mlp_pdrop=0.05,
        n_heads=8,
        device="cpu",
        linear_output=True,
        use_rot_embed=False,
        use_abs_pos_emb=True,
        bias=False,
        use_ada_conditioning=True,
        use_noise_encoder=False,
        use_modality_encoder=True,
        use_mlp_goal=True,
    )
    cfg.update(overrides)
    return cfg


def mdt_tiny(**overrides) -> dict:
    """BASELINE C1 'MDT-D tiny': 2 DiT blocks, d=128, horizon 10."""
    return mdt_default(embed_dim=128, obs_dim=128, n_enc_layers=1, n_dec_layers=2, **overrides)


def retarget(cfg: dict, target: str) -> dict:
    """Copy of *cfg* with another ``_target_`` (used only by the golden generator to point at the reference)."""
    out = copy.deepcopy(cfg)
    out["_target_"] = target
    return out


NAMED = {
    "mdtv_default": mdtv_default,
    "mdtv_tiny": mdtv_tiny,
    "mdt_default": mdt_default,
    "mdt_tiny": mdt_tiny,
}
