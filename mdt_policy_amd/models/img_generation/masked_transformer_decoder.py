"""MaskedTransformerImgDecoder -- the masked generative foresight (MGF) head -- with the reference's constructor,
parameter tree and call signatures; its Linears, RMSNorm, SwishGLU and self-attention run on libmdt_hip.so
(include/mdt_mae.h, include/mdt_hip_ops.h), forward and backward.

Reference: mdt/models/img_generation/masked_transformer_decoder.py:68-283 (``MaskedTransformerImgDecoder``), built from
conf/model/img_gen/masked_transformer.yaml by MDTVAgent (mdt/models/mdtv_agent.py:99) and trained through
``compute_img_gen_loss`` (:411-421): ``pred, mask, restore, visible = gen_img(latent_encoder_emb, goal_imgs)``,
``loss = gen_img.compute_loss(goal_imgs, pred, mask, restore)``.  The transformer blocks are voltron-robotics'
``Block(do_rms_norm, do_swish_glu, do_layer_scale)`` -- un-vendored upstream, restated here from the published code
(parameter names ``pre_norm_attn.g, attn.qkv, attn.proj, layer_scale_attn.gamma, pre_norm_mlp.g, mlp.0.project, mlp.1,
layer_scale_mlp.gamma``): PARITY UNPINNED for the block internals, see oracle/mae_oracle.py.

What runs where: every Linear (context projection, patch embedding of the VISIBLE patches only, qkv / proj / SwishGLU
project / mlp.1 of the 6 blocks, patch prediction -- 98 % of the FLOPs) on the fp32-MFMA GEMM, RMSNorm / SwishGLU / the
102-token attention on their HIP kernels; the random mask, gathers, residual adds and the loss are PyTorch-ROCm glue, as in
the agent's own loss code.  No eager fallback: CPU tensors raise.
"""
from __future__ import annotations

from typing import Optional, Tuple

import numpy as np
import torch
from torch import nn

from . import _hip_ops as ops


def get_2D_position_embeddings(embed_dim: int, grid_size: int, cls_token: bool = False) -> np.ndarray:
    """2-D sine-cosine position table (reference :29-42; MAE repository layout): one half of the channels per grid axis."""
    def axis_table(dim, pos):
        omega = 1.0 / 10000 ** (np.arange(dim // 2, dtype=np.float32) / (dim / 2.0))
        out = np.einsum("m,d->md", pos.reshape(-1), omega)
        return np.concatenate([np.sin(out), np.cos(out)], axis=1)
    gh = gw = np.arange(grid_size, dtype=np.float32)
    grid = np.stack(np.meshgrid(gw, gh), axis=0).reshape(2, 1, grid_size, grid_size)
    table = np.concatenate([axis_table(embed_dim // 2, grid[0]), axis_table(embed_dim // 2, grid[1])], axis=1)
    return np.concatenate([np.zeros([1, embed_dim]), table], axis=0) if cls_token else table


class PatchEmbed(nn.Module):
    """Parameter container of the reference's PatchEmbed (:46-65): a Conv2d with kernel = stride = patch, i.e. a Linear
    over flattened (c, ph, pw) patches -- which is how the decoder applies it, to the visible patches only."""

    def __init__(self, resolution: int, patch_size: int, embed_dim: int, in_channels: int = 3, flatten: bool = True):
        super().__init__()
        self.resolution, self.patch_size = (resolution, resolution), (patch_size, patch_size)
        self.grid_size = (resolution // patch_size, resolution // patch_size)
        self.num_patches = self.grid_size[0] * self.grid_size[1]
        self.flatten = flatten
        self.proj = nn.Conv2d(in_channels, embed_dim, kernel_size=self.patch_size, stride=self.patch_size)


class RMSNorm(nn.Module):
    def __init__(self, dim: int, eps: float = 1e-8):
        super().__init__()
        self.scale, self.eps = dim ** -0.5, eps
        self.g = nn.Parameter(torch.ones(dim))


class SwishGLU(nn.Module):
    def __init__(self, in_dim: int, out_dim: int):
        super().__init__()
        self.act, self.project = nn.SiLU(), nn.Linear(in_dim, 2 * out_dim)


class LayerScale(nn.Module):
    def __init__(self, dim: int, init_values: float = 0.1):
        super().__init__()
        self.gamma = nn.Parameter(init_values * torch.ones(dim))


class Attention(nn.Module):
    def __init__(self, embed_dim: int, n_heads: int):
        super().__init__()
        assert embed_dim % n_heads == 0, "`embed_dim` must be divisible by `n_heads`!"
        self.n_heads, self.scale = n_heads, (embed_dim // n_heads) ** -0.5
        self.qkv, self.proj = nn.Linear(embed_dim, 3 * embed_dim, bias=True), nn.Linear(embed_dim, embed_dim)


class Block(nn.Module):
    """Parameter container of voltron's pre-norm transformer block (RMSNorm, SwishGLU MLP, LayerScale)."""

    def __init__(self, embed_dim: int, n_heads: int, mlp_ratio: float = 4.0, do_rms_norm: bool = True, do_swish_glu: bool = True,
                 do_layer_scale: bool = True):
        super().__init__()
        if not (do_rms_norm and do_swish_glu and do_layer_scale):
            raise NotImplementedError("the HIP block implements the decoder's configuration (RMSNorm + SwishGLU + LayerScale)")
        self.embed_dim, self.n_heads = embed_dim, n_heads
        self.pre_norm_attn = RMSNorm(embed_dim)
        self.attn = Attention(embed_dim, n_heads)
        self.layer_scale_attn = LayerScale(embed_dim)
        self.pre_norm_mlp = RMSNorm(embed_dim)
        hidden = int(mlp_ratio * embed_dim)
        self.mlp = nn.Sequential(SwishGLU(embed_dim, hidden), nn.Linear(hidden, embed_dim))
        self.layer_scale_mlp = LayerScale(embed_dim)


class MaskedTransformerImgDecoder(nn.Module):
    def __init__(self, resolution: int, patch_size: int, decoder_depth: int, decoder_embed_dim: int, decoder_n_heads: int,
                 context_dim: int, symmetric_mask: bool = True, num_images: int = 2, mlp_ratio: float = 4.0, in_channels: int = 3,
                 mask_ratio: float = 0.9, img_gen_frame_diff: int = 3, video_gen: bool = False, norm_pixel_loss: bool = True):
        super().__init__()
        self.img_gen_frame_diff, self.mask_ratio, self.patch_size, self.resolution = img_gen_frame_diff, mask_ratio, patch_size, resolution
        self.symmetric_mask = symmetric_mask
        self.num_patches = (resolution // patch_size) ** 2
        self.patch2embed = PatchEmbed(resolution, patch_size, decoder_embed_dim, in_channels=in_channels)
        self.in_channels, self.norm_pixel_loss, self.mlp_ratio = in_channels, norm_pixel_loss, mlp_ratio
        self.decoder_embed_dim, self.decoder_n_heads, self.decoder_depth = decoder_embed_dim, decoder_n_heads, decoder_depth
        self.encoder2decoder = nn.Linear(context_dim, decoder_embed_dim)
        self.num_images = num_images
        self.mask_token = nn.Parameter(torch.zeros(1, 1, decoder_embed_dim))
        self.ctx_dec_pe = nn.Parameter(torch.randn(1, 2, 1, decoder_embed_dim))
        self.decoder_pe = nn.Parameter(torch.zeros(1, self.num_patches, decoder_embed_dim), requires_grad=False)
        self.decoder_blocks = nn.ModuleList([Block(decoder_embed_dim, decoder_n_heads, mlp_ratio) for _ in range(decoder_depth)])
        self.decoder_norm = RMSNorm(decoder_embed_dim)
        self.decoder_patch_prediction = nn.Linear(decoder_embed_dim, (patch_size ** 2) * in_channels, bias=True)
        self.video_gen = video_gen
        self._packs = ops.PackedWeights()
        from ...utils import weight_cache
        weight_cache.track(self)
        self.initialize_weights()

    # -- initialisation (reference :176-203) ---------------------------------------------------------
    def initialize_weights(self) -> None:
        table = get_2D_position_embeddings(self.decoder_embed_dim, int(self.patch2embed.num_patches ** 0.5), cls_token=False)
        self.decoder_pe.data.copy_(torch.from_numpy(table).float().unsqueeze(0))
        w = self.patch2embed.proj.weight.data
        nn.init.xavier_uniform_(w.view([w.shape[0], -1]))
        nn.init.normal_(self.mask_token, std=0.02)
        self.apply(self.transformer_initializer)

    @staticmethod
    def transformer_initializer(m: nn.Module) -> None:
        if isinstance(m, nn.Linear):
            nn.init.xavier_uniform_(m.weight)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0.0)
        elif isinstance(m, nn.LayerNorm):
            nn.init.constant_(m.weight, 1.0)
            nn.init.constant_(m.bias, 0.0)

    def mark_dirty(self) -> None:
        """Forget the packed weight images (parameters written through ``.data`` / a foreign fused optimizer)."""
        self._packs.invalidate()

    def train(self, mode: bool = True):
        out = super().train(mode)
        self._packs.invalidate()
        return out

    def _apply(self, fn, *a, **kw):  # .to() / .cuda(): the parameters are re-created, the packed images belong to the old ones
        out = super()._apply(fn, *a, **kw)
        if getattr(self, "_packs", None) is not None:
            self._packs = ops.PackedWeights()
        return out

    def __getstate__(self):
        d = self.__dict__.copy()
        d["_packs"] = ops.PackedWeights()
        return d

    def __setstate__(self, state):  # the copy is a new module: register it with the optimizer hook (utils/weight_cache.py)
        super().__setstate__(state)
        from ...utils import weight_cache
        weight_cache.track(self)
        self.mark_dirty()

    # -- helpers -------------------------------------------------------------------------------------
    def _linear(self, x, lin) -> torch.Tensor:
        return ops.HipLinear.apply(x, lin.weight, lin.bias, self._packs)

    def patchify(self, imgs: torch.Tensor) -> torch.Tensor:
        """(B, ctx, C, R, R) -> (B, ctx, n_patches, ph*pw*C): the loss targets' layout (reference :206-213)."""
        B, X, Cn, R, _ = imgs.shape
        p = self.patch_size
        g = R // p
        return imgs.reshape(B, X, Cn, g, p, g, p).permute(0, 1, 3, 5, 4, 6, 2).reshape(B, X, g * g, p * p * Cn)

    def _conv_patches(self, imgs: torch.Tensor) -> torch.Tensor:
        """(B, ctx, C, R, R) -> (B, ctx, n_patches, C*ph*pw): a patch as the Conv2d kernel of ``patch2embed`` sees it."""
        B, X, Cn, R, _ = imgs.shape
        p = self.patch_size
        g = R // p
        return imgs.reshape(B, X, Cn, g, p, g, p).permute(0, 1, 3, 5, 2, 4, 6).reshape(B, X, g * g, Cn * p * p)

    def mask(self, ctx_patches: torch.Tensor, mask_ratio: Optional[float] = None):
        """The reference's method by its own signature (:126-171): random masking of (bsz, ctx, n_patches, d) patch embeddings
        -> (visible_patches, mask (1 = removed), restore_idxs); ``symmetric_mask``: the same patches for every frame.  The
        forward below does not come through here (it never embeds the patches it would drop); this is for callers of the
        reference API."""
        bsz, ctx_len, n_patches, d = ctx_patches.shape
        shuffle, m, restore, n_keep = self._mask_indices(n_patches, bsz, ctx_patches.device, mask_ratio, ctx_len=ctx_len)
        idx = shuffle[:, None, :n_keep, None] if self.symmetric_mask else shuffle[:, :, :n_keep, None]
        visible = torch.gather(ctx_patches, 2, idx.expand(bsz, ctx_len, n_keep, d))
        return visible, m, restore

    def _mask_indices(self, n_patches: int, bsz: int, device, mask_ratio: Optional[float] = None, noise: Optional[torch.Tensor] = None,
                      ctx_len: int = 2):
        """(shuffle_idxs, mask (1 = removed), restore_idxs, n_keep) as the reference's ``mask`` builds them (:124-171).
        symmetric_mask: per-sample shuffle of the patches, shared by the frames; ``noise`` (bsz, n_patches) replaces the draw.
        Otherwise the reference's other branch AS IT IS WRITTEN: (bsz, ctx, n_patches) noise argsorted over the CTX axis
        (:159-160), so shuffle / restore hold frame indices 0 .. ctx-1, the "visible" patches of a frame are its patches
        shuffle[:, c, :n_keep] (i.e. patch 0 or 1), and the mask -- constant along the axis it is gathered on (:167-169) --
        removes the patches n_keep .. n-1 of every frame; ``noise`` is (bsz, ctx, n_patches) then."""
        ratio = self.mask_ratio if mask_ratio is None else mask_ratio
        n_keep = int(n_patches * (1 - ratio))
        if not self.symmetric_mask:
            if noise is None:
                noise = torch.rand(bsz, ctx_len, n_patches, device=device)
            shuffle = torch.argsort(noise, dim=1)
            restore = torch.argsort(shuffle, dim=1)
            m = torch.ones(bsz, ctx_len, n_patches, device=device)
            m[:, :, :n_keep] = 0
            return shuffle, torch.gather(m, 1, restore), restore, n_keep
        if noise is None:
            noise = torch.rand(bsz, n_patches, device=device)
        shuffle = torch.argsort(noise, dim=1)
        restore = torch.argsort(shuffle, dim=1)
        m = torch.ones(bsz, n_patches, device=device)
        m.scatter_(1, shuffle[:, :n_keep], 0.0)
        return shuffle, m, restore, n_keep

    # -- reference API -------------------------------------------------------------------------------
    def forward(self, context, target_images, img_gen_frame_diff: int = 3, noise: Optional[torch.Tensor] = None):
        """-> (reconstructions (B, num_images, n_patches, p*p*C), mask, restore_idxs, visible_patches) (reference :215-283)."""
        if context.device.type != "cuda":
            raise RuntimeError("MaskedTransformerImgDecoder runs only on a ROCm GPU; there is no CPU execution path in mdt_policy_amd")
        d, X, n = self.decoder_embed_dim, self.num_images, self.num_patches
        B = context.shape[0]
        if X != 2 or target_images.shape[1] != 2:
            raise NotImplementedError("the decoder pairs frame 0 and frame K (num_images = 2), as the reference's forward does")
        if context.shape[1] + X * n > 128:
            raise NotImplementedError(f"{context.shape[1] + X * n} decoder tokens: the HIP attention covers up to 128 "
                                      "(the shipped 112 x 112 / 16 configuration has 102)")
        # every Linear's packed images (and, under autograd, the W^T images of the backward) in one launch
        self._packs.refresh([m.weight for m in self.modules() if isinstance(m, (nn.Linear, nn.Conv2d))],
                            need_t=torch.is_grad_enabled())
        emb_context = self._linear(context.float(), self.encoder2decoder)
        shuffle, m, restore, n_keep = self._mask_indices(n, B, context.device, self.mask_ratio, noise, ctx_len=X)
        # the patch of every visible slot: shared by the frames, or per frame (symmetric_mask = False)
        gidx = shuffle[:, None, :n_keep, None] if self.symmetric_mask else shuffle[:, :, :n_keep, None]
        # only the visible patches are embedded: the masked ones never reach the blocks (the reference embeds all 2 x 49
        # and gathers 2 x 12)
        pix = torch.gather(self._conv_patches(target_images.float()), 2, gidx.expand(B, X, n_keep, self.in_channels * self.patch_size ** 2))
        pe = self.decoder_pe.to(context.device)
        vis = ops.HipLinear.apply(pix, self.patch2embed.proj.weight, self.patch2embed.proj.bias, self._packs)
        vis = vis + torch.gather(pe.expand(B, n, d)[:, None].expand(B, X, n, d), 2, gidx.expand(B, X, n_keep, d))
        if self.symmetric_mask:
            tokens = self.mask_token.reshape(1, 1, 1, d).expand(B, X, n, d).scatter(2, gidx.expand(B, X, n_keep, d), vis)
        else:
            # the reference's un-shuffle with its (bsz, ctx, n_patches) restore_idxs (:243-248): position p of frame c takes
            # slot restore[b, c, p] of [visible | mask tokens]
            slots = torch.cat([vis, self.mask_token.reshape(1, 1, 1, d).expand(B, X, n - n_keep, d)], dim=2)
            tokens = torch.gather(slots, 2, restore[..., None].expand(B, X, n, d))
        tokens = tokens + pe[None] + self.ctx_dec_pe[:, :2]      # position embedding a second time, as the reference (:268-271)
        x = torch.cat([emb_context, tokens.reshape(B, X * n, d)], dim=1)
        # Per block (voltron Block): x = x + ls(attn(norm(x))); x = x + ls(mlp(norm(x))).  Every residual sum runs in one launch
        # with the norm that follows it (the next branch's, at the end the decoder's own), and that launch's backward also adds
        # the two gradients that reach the summed rows (HipScaleResidualNorm); the first norm hands out both uses of its input.
        blocks = list(self.decoder_blocks)
        if blocks:
            x, h = ops.HipRMSNormBranch.apply(x, blocks[0].pre_norm_attn.g)
            for i, blk in enumerate(blocks):
                att = ops.HipSelfAttention.apply(self._linear(h, blk.attn.qkv), blk.attn.n_heads, blk.attn.scale)
                x, h = ops.HipScaleResidualNorm.apply(x, self._linear(att, blk.attn.proj), blk.layer_scale_attn.gamma, blk.pre_norm_mlp.g)
                z = ops.HipSwiGLUMLP.apply(h, blk.mlp[0].project.weight, blk.mlp[0].project.bias, blk.mlp[1].weight, blk.mlp[1].bias,
                                           self._packs)
                g_next = blocks[i + 1].pre_norm_attn.g if i + 1 < len(blocks) else self.decoder_norm.g
                x, h = ops.HipScaleResidualNorm.apply(x, z, blk.layer_scale_mlp.gamma, g_next)
            x = h
        else:
            x = ops.HipRMSNorm.apply(x, self.decoder_norm.g)
        rec = self._linear(x[:, context.shape[1]:], self.decoder_patch_prediction).reshape(B, X, n, -1)
        return rec, m, restore, vis.reshape(B, X * n_keep, d)

    def compute_loss(self, imgs: torch.Tensor, ctx_reconstructions: torch.Tensor, mask: torch.Tensor, restore_idxs: torch.Tensor):
        """Mean squared error per patch of frame 0 and frame K over the REMOVED patches, averaged (reference :228-262)."""
        assert self.norm_pixel_loss, "`norm_pixel_loss` should always be true... false only for visualizations!"
        if ctx_reconstructions.device.type != "cuda":
            raise RuntimeError("MaskedTransformerImgDecoder runs only on a ROCm GPU; there is no CPU execution path in mdt_policy_amd")
        # = (sum_x (per_patch[:, x] * mask).sum() / mask.sum()) / 2 with per_patch = ((rec - patchify(imgs)) ** 2).mean(-1)
        if mask.dim() == 3:  # symmetric_mask = False: every frame against its own mask (:256-258)
            X = mask.shape[1]
            terms = [ops.HipPatchMSE.apply(ctx_reconstructions[:, x:x + 1].float(), imgs[:, x:x + 1].float(), mask[:, x].float(),
                                           self.patch_size) for x in range(X)]
            return sum(terms) / X
        return ops.HipPatchMSE.apply(ctx_reconstructions.float(), imgs.float(), mask.float(), self.patch_size)
