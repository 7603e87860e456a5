"""ClipStyleProjection and its MAPBlock pooling head with the reference's constructors, parameter trees and forward
signatures; MAPBlock's arithmetic (forward AND backward) runs in libmdt_hip.so (``mdt_map_pool_*`` of
include/mdt_map_pool.h, gfx950 kernels).

Reference: mdt/models/networks/transformers/transformer_blocks.py -- RMSNorm :43, SwishGLU :55, MAPAttention :718,
MAPBlock :746, ClipStyleProjection :833, MeanPooling :873.  The agent builds ``ClipStyleProjection('map', 384,
clip_token_index=1, num_token=...)`` (mdt/models/mdtv_agent.py:133-138) and pools ``latent_encoder_emb`` of the language
and of the vision goal with it for the contrastive loss (:440-484).

'map' / 'map_state_only' run the HIP MAPBlock.  The other clip styles are index / mean / one small Linear over a
(B, 4, d) tensor -- a slice, a mean and torch's own modules, kept as host PyTorch (none of them is the shipped
configuration).  There is no eager fallback for MAPBlock: CPU tensors or a missing library raise.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional

import torch
from torch import nn

from .... import _lib


class RMSNorm(nn.Module):
    """Parameter container (reference transformer_blocks.py:43-51); the arithmetic runs in the owning MAPBlock."""

    def __init__(self, dim: int, eps: float = 1e-8) -> None:
        super().__init__()
        self.scale, self.eps = dim ** -0.5, eps
        self.g = nn.Parameter(torch.ones(dim))


class SwishGLU(nn.Module):
    """Parameter container (reference transformer_blocks.py:55-62)."""

    def __init__(self, in_dim: int, out_dim: int) -> None:
        super().__init__()
        self.act, self.project = nn.SiLU(), nn.Linear(in_dim, 2 * out_dim)


class MAPAttention(nn.Module):
    """Parameter container (reference transformer_blocks.py:718-743)."""

    def __init__(self, embed_dim: int, n_heads: int) -> None:
        super().__init__()
        assert embed_dim % n_heads == 0, "`embed_dim` must be divisible by `n_heads`!"
        self.n_heads, self.scale = n_heads, (embed_dim // n_heads) ** -0.5
        self.q, self.kv = nn.Linear(embed_dim, embed_dim, bias=False), nn.Linear(embed_dim, 2 * embed_dim, bias=False)
        self.proj = nn.Linear(embed_dim, embed_dim)


class _MapTape:
    def __init__(self, mod, tape_id):
        self.mod, self.id = mod, tape_id

    def release(self):
        if self.id is not None and self.mod._handle is not None:
            try:
                _lib.load().mdt_map_pool_tape_release(self.mod._handle, self.id)
            except Exception:
                pass
        self.id = None

    __del__ = release


class _MapPoolFn(torch.autograd.Function):
    """MAPBlock.forward under autograd: HIP forward with a tape, HIP backward."""

    @staticmethod
    def forward(ctx, mod, x, names, *params):
        lib, stream = mod._engine(x.device, train=True)
        B, N, _ = x.shape
        out = torch.empty((B, mod.n_latents, mod.embed_dim), device=x.device, dtype=torch.float32)
        tape = C.c_int32(-1)
        _lib.check(lib.mdt_map_pool_forward_train(mod._handle, x.data_ptr(), B, N, out.data_ptr(), C.byref(tape), stream))
        ctx.mod, ctx.tape = mod, _MapTape(mod, int(tape.value))
        ctx.named = list(zip(names, params))
        ctx.x_shape, ctx.need_x = x.shape, x.requires_grad
        return out

    @staticmethod
    def backward(ctx, g_out):
        mod = ctx.mod
        if ctx.tape.id is None:
            raise RuntimeError("the HIP MAPBlock tape of this forward was already consumed (no retain_graph support)")
        lib = _lib.load()
        stream = torch.cuda.current_stream(g_out.device).cuda_stream
        g = g_out.detach().float().contiguous()
        grads = torch.zeros(mod._grad_numel, device=g.device, dtype=torch.float32)
        d_x = torch.empty(ctx.x_shape, device=g.device, dtype=torch.float32) if ctx.need_x else None
        _lib.check(lib.mdt_map_pool_backward(mod._handle, ctx.tape.id, g.data_ptr(), grads.data_ptr(),
                                             None if d_x is None else d_x.data_ptr(), stream))
        ctx.tape.release()
        out = []
        for name, p in ctx.named:
            off, n = mod._grad_layout[name]
            out.append(grads[off:off + n].view(p.shape) if p.requires_grad else None)
        return (None, d_x, None, *out)


class MAPBlock(nn.Module):
    """Multiheaded attention pooling block (reference transformer_blocks.py:746-791)."""

    def __init__(self, n_latents: int, embed_dim: int, n_heads: int, output_dim: None, mlp_ratio: float = 4.0,
                 do_rms_norm: bool = True, do_swish_glu: bool = True) -> None:
        super().__init__()
        if not (do_rms_norm and do_swish_glu):
            raise NotImplementedError("the HIP MAPBlock implements the reference's defaults (RMSNorm + SwishGLU)")
        self.n_latents, self.in_dim, self.n_heads = n_latents, embed_dim, 2 * n_heads
        self.embed_dim = output_dim if output_dim is not None else embed_dim
        self.projection = nn.Linear(embed_dim, self.embed_dim)
        self.latents = nn.Parameter(torch.zeros(self.n_latents, self.embed_dim))
        nn.init.normal_(self.latents, std=0.02)
        self.attn_norm = RMSNorm(self.embed_dim)
        self.attn = MAPAttention(self.embed_dim, n_heads=self.n_heads)
        self.mlp_norm = RMSNorm(self.embed_dim)
        hidden = int(mlp_ratio * self.embed_dim)
        self.mlp = nn.Sequential(SwishGLU(self.embed_dim, hidden), nn.Linear(hidden, self.embed_dim))
        self._cfg = dict(n_latents=n_latents, embed_dim=embed_dim, output_dim=self.embed_dim, n_heads=n_heads,
                         mlp_hidden=hidden)
        self._handle: Optional[C.c_void_p] = None
        self._handle_device = None
        self._uploaded: Dict[str, tuple] = {}
        from ....utils import weight_cache
        weight_cache.track(self)
        self._grad_layout = None

    # -- library handle --------------------------------------------------------------------------
    def __getstate__(self):  # copy.deepcopy / pickle: never the library handle
        d = self.__dict__.copy()
        d["_handle"], d["_handle_device"], d["_uploaded"], d["_grad_layout"] = None, None, {}, None
        return d

    def __setstate__(self, state):  # the copy is a new module: register it with the optimizer hook (utils/weight_cache.py)
        super().__setstate__(state)
        from ....utils import weight_cache
        weight_cache.track(self)
        self.mark_dirty()

    def _apply(self, fn, *a, **kw):  # .to()/.cuda(): parameters are re-created, drop the stale handle
        out = super()._apply(fn, *a, **kw)
        self._drop_handle()
        return out

    def _drop_handle(self):
        h = getattr(self, "_handle", None)
        if h is not None and h.value:
            try:
                _lib.load().mdt_map_pool_destroy(h)
            except Exception:
                pass
        self._handle, self._handle_device, self._uploaded, self._grad_layout = None, None, {}, None

    def __del__(self):
        try:
            self._drop_handle()
        except Exception:
            pass

    def mark_dirty(self) -> None:
        """Forget what was uploaded (weights written through ``.data`` / a foreign fused optimizer are not seen by the
        version counter): the next call re-uploads every parameter."""
        self._uploaded = {}

    def train(self, mode: bool = True):
        out = super().train(mode)
        self._uploaded = {}
        return out

    def _engine(self, device: torch.device, train: bool = False):
        if device.type != "cuda":
            raise RuntimeError("MAPBlock runs only on a ROCm GPU (hand-written gfx950 kernels); move the module and its "
                               "input with .to('cuda') -- there is no CPU execution path")
        lib = _lib.load()
        if self._handle is None or self._handle_device != device:
            self._drop_handle()
            cfg = _lib.MapPoolConfig(**self._cfg)
            h = C.c_void_p()
            from ....utils import torch_allocator
            torch_allocator.install()  # workspace / tapes / scratch live in torch's caching allocator
            with torch.cuda.device(device):
                _lib.check(lib.mdt_map_pool_create(C.byref(cfg), C.byref(h)))
            self._handle, self._handle_device = h, device
        if train and self._grad_layout is None:
            _lib.check(lib.mdt_map_pool_train_prepare(self._handle))
            self._uploaded = {}  # every weight is uploaded again so that its transposed image exists
            n = lib.mdt_map_pool_param_count(self._handle)
            self._grad_layout = {lib.mdt_map_pool_param_name(self._handle, i).decode():
                                 (int(lib.mdt_map_pool_grad_offset(self._handle, i)),
                                  int(lib.mdt_map_pool_param_numel(self._handle, i))) for i in range(n)}
            self._grad_numel = int(lib.mdt_map_pool_grad_numel(self._handle))
        stream = torch.cuda.current_stream(device).cuda_stream
        for name, p in self.named_parameters():
            tag = (p.data_ptr(), p._version)
            if self._uploaded.get(name) == tag:
                continue
            if p.device != device or p.dtype != torch.float32:
                raise RuntimeError(f"parameter {name} must be float32 on {device}, got {p.dtype} on {p.device}")
            src = p.detach().contiguous()
            _lib.check(lib.mdt_map_pool_load_param(self._handle, name.encode(), src.data_ptr(), src.numel(), stream))
            self._uploaded[name] = tag
        return lib, stream

    # -- reference API -----------------------------------------------------------------------------
    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """x (B, N, embed_dim) -> (B, n_latents, output_dim), squeezed on dim 1 (reference :787-791)."""
        assert x.ndim == 3 and x.shape[-1] == self.in_dim
        B, N, _ = x.shape
        if torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self.parameters())):
            if x.device.type != "cuda":
                self._engine(x.device)  # raises the no-CPU-path error
            xin = x if (x.dtype == torch.float32 and x.is_contiguous() and x.data_ptr() % 16 == 0) \
                else x.float().contiguous().clone()
            named = list(self.named_parameters())
            out = _MapPoolFn.apply(self, xin, [k for k, _ in named], *[p for _, p in named])
            return out.squeeze(dim=1)
        lib, stream = self._engine(x.device)
        xin = x.detach()
        if xin.dtype != torch.float32:
            xin = xin.float()
        if not xin.is_contiguous() or xin.data_ptr() % 16:
            xin = xin.contiguous().clone()
        out = torch.empty((B, self.n_latents, self.embed_dim), device=x.device, dtype=torch.float32)
        _lib.check(lib.mdt_map_pool_forward(self._handle, xin.data_ptr(), B, N, out.data_ptr(), stream))
        return out.squeeze(dim=1)


class MeanPooling(nn.Module):
    """Mean over the tokens (reference transformer_blocks.py:873-880)."""

    def __init__(self, token_dim):
        super().__init__()
        self.token_dim = token_dim

    def forward(self, x):
        return x.mean(dim=1).view(-1, self.token_dim)


# clip_style -> (pooling kind, drop the goal token first)
_CLIP_STYLES = {
    "map": ("map", False), "map_state_only": ("map", True),
    "mean_pooling": ("mean", False), "mean_pool_state_only": ("mean", True),
    "mlp": ("mlp", False), "single_token": ("token", False), "multihead": ("keep", False),
}


class ClipStyleProjection(nn.Module):
    """Pools the (B, tokens, d) context into the embedding the contrastive loss compares (reference
    transformer_blocks.py:833-870: same constructor, same ``latent_proj`` sub-module per style, same outputs)."""

    def __init__(self, clip_style, token_dim=384, clip_token_index=0, num_token=4):
        super().__init__()
        if clip_style not in _CLIP_STYLES:
            raise ValueError("Invalid clip_style. Expected 'map', 'mean_pooling', or 'single_token' or 'multihead'.")
        self.clip_style = clip_style
        self.clip_token_index = clip_token_index
        kind, self._state_only = _CLIP_STYLES[clip_style]
        self._kind = kind
        if kind == "map":
            self.latent_proj = MAPBlock(1, token_dim, 8, output_dim=token_dim)
        elif kind == "mean":
            self.latent_proj = MeanPooling(token_dim)
        elif kind == "mlp":
            self.latent_proj = nn.Sequential(nn.Linear(num_token * token_dim, token_dim), nn.LayerNorm(token_dim), nn.Tanh())
        else:  # one token / all tokens: nothing to learn
            self.latent_proj = nn.Identity()

    def forward(self, x):
        if self._kind == "token":
            x = x[:, self.clip_token_index, :]
        elif self._state_only:
            x = x[:, 1:]
        elif self._kind == "mlp":
            x = x.flatten(1)
        return self.latent_proj(x)
