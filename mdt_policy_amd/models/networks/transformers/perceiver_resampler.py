"""Perceiver resampler with the reference's constructor, parameter tree and forward signature; the arithmetic
runs in libmdt_hip.so (``mdt_resampler_*`` of include/mdt_resampler.h, gfx950 kernels).

Reference: mdt/models/networks/transformers/perceiver_resampler.py (PerceiverAttentionLayer :11-82,
PerceiverResampler :85-162); built by the agent at mdt/models/mdtv_agent.py:90-97 and called in
``compute_voltron_embeddings`` (:392-404) on the two cameras' Voltron patch tokens,
``(B, 1, 2*196, 384) -> (B, 3, 384)`` = the ``state_images`` tokens of the denoiser.

Differentiable: under autograd the forward keeps a tape in the library handle and the backward (HIP) returns the
gradient of every parameter and of the media tokens, so the module trains inside the agent like the reference's.
There is no eager fallback: CPU tensors or a missing library raise.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional

import torch
from torch import nn

from .... import _lib
from .utils import feed_forward_layer


class _ResamplerTape:
    def __init__(self, mod, tape_id):
        self.mod, self.id = mod, tape_id

    def release(self):
        if self.id is not None and self.mod._handle is not None:
            try:
                _lib.load().mdt_resampler_tape_release(self.mod._handle, self.id)
            except Exception:
                pass
        self.id = None

    __del__ = release


class _ResamplerFn(torch.autograd.Function):
    """PerceiverResampler.forward under autograd: HIP forward with a tape, HIP backward."""

    @staticmethod
    def forward(ctx, mod, x, mask_u8, names, *params):
        lib, stream = mod._engine(x.device, train=True)
        B, T, n, dim = x.shape
        out = torch.empty((B, mod.num_queries, dim), device=x.device, dtype=torch.float32)
        tape = C.c_int32(-1)
        _lib.check(lib.mdt_resampler_forward_train(mod._handle, x.data_ptr(), None if mask_u8 is None else mask_u8.data_ptr(),
                                                   B, T, n, out.data_ptr(), C.byref(tape), stream))
        ctx.mod, ctx.tape = mod, _ResamplerTape(mod, int(tape.value))
        ctx.named = list(zip(names, params))
        ctx.x_shape, ctx.need_x = x.shape, x.requires_grad
        return out

    @staticmethod
    def backward(ctx, g_out):
        mod = ctx.mod
        if ctx.tape.id is None:
            raise RuntimeError("the HIP resampler tape of this forward was already consumed (no retain_graph support)")
        lib = _lib.load()
        stream = torch.cuda.current_stream(g_out.device).cuda_stream
        g = g_out.detach().float().contiguous()
        grads = torch.zeros(mod._grad_numel, device=g.device, dtype=torch.float32)
        d_x = torch.empty(ctx.x_shape, device=g.device, dtype=torch.float32) if ctx.need_x else None
        _lib.check(lib.mdt_resampler_backward(mod._handle, ctx.tape.id, g.data_ptr(), grads.data_ptr(),
                                              None if d_x is None else d_x.data_ptr(), stream))
        ctx.tape.release()
        out = []
        for name, p in ctx.named:
            off, n = mod._grad_layout[name]
            out.append(grads[off:off + n].view(p.shape) if p.requires_grad else None)
        return (None, d_x, None, None, *out)


class PerceiverAttentionLayer(nn.Module):
    """Parameters of one latent cross-attention layer (reference perceiver_resampler.py:14-30)."""

    def __init__(self, dim: int, dim_head: int = 64, heads: int = 8):
        super().__init__()
        self.scale = dim_head ** -0.5
        self.heads = heads
        self.dim_head = dim_head
        inner_dim = dim_head * heads
        self.norm_media = nn.LayerNorm(dim)
        self.norm_latents = nn.LayerNorm(dim)
        self.to_q = nn.Linear(dim, inner_dim, bias=False)
        self.to_k = nn.Linear(dim, inner_dim, bias=False)
        self.to_v = nn.Linear(dim, inner_dim, bias=False)
        self.to_out = nn.Linear(inner_dim, dim, bias=False)

    def forward(self, features, latents):  # pragma: no cover - guard
        raise RuntimeError("PerceiverAttentionLayer is a parameter container; call the owning PerceiverResampler")


class PerceiverResampler(nn.Module):
    def __init__(self, dim: int, depth: int, dim_head: int = 64, heads: int = 8, num_latents: int = 64,
                 num_time_embeds: int = 4, ff_mult: int = 4, activation: str = "gelu", trainable: bool = True):
        super().__init__()
        from ....utils import weight_cache
        weight_cache.track(self)
        self.dim = dim
        self.num_queries = num_latents
        self._cfg = dict(dim=dim, depth=depth, dim_head=dim_head, heads=heads, num_latents=num_latents,
                         num_time_embeds=num_time_embeds, ff_mult=int(ff_mult), activation=0)
        if int(ff_mult) != ff_mult:
            raise NotImplementedError("ff_mult must be an integer")
        self.latents = nn.Parameter(torch.randn(num_latents, dim))
        self.time_pos_emb = nn.Parameter(torch.randn(num_time_embeds, 1, dim))
        self.layers = nn.ModuleList([])
        for _ in range(depth):
            self.layers.append(nn.ModuleList([PerceiverAttentionLayer(dim=dim, dim_head=dim_head, heads=heads),
                                              feed_forward_layer(dim=dim, mult=ff_mult, activation=activation)]))
        self.norm = nn.LayerNorm(dim)
        self._update_trainable_state(trainable)
        self._handle: Optional[C.c_void_p] = None
        self._handle_device = None
        self._uploaded: Dict[str, tuple] = {}

    def _update_trainable_state(self, trainable: bool = True):
        for param in self.parameters():
            param.requires_grad = trainable

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
                _lib.load().mdt_resampler_destroy(h)
            except Exception:
                pass
        self._handle, self._handle_device, self._uploaded = None, None, {}
        self._grad_layout = None

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
            raise RuntimeError("the Perceiver resampler runs only on a ROCm GPU (hand-written gfx950 kernels); move "
                               "the module and its input with .to('cuda') -- there is no CPU execution path")
        lib = _lib.load()
        if self._handle is None or self._handle_device != device:
            self._drop_handle()
            cfg = _lib.ResamplerConfig(**self._cfg)
            h = C.c_void_p()
            from ....utils import torch_allocator
            torch_allocator.install()  # workspace / tapes / scratch live in torch's caching allocator
            with torch.cuda.device(device):
                _lib.check(lib.mdt_resampler_create(C.byref(cfg), C.byref(h)))
            self._handle, self._handle_device = h, device
        if train and getattr(self, "_grad_layout", None) is None:
            _lib.check(lib.mdt_resampler_train_prepare(self._handle))
            self._uploaded = {}  # every weight is uploaded again so that its transposed image exists
            n = lib.mdt_resampler_param_count(self._handle)
            self._grad_layout = {lib.mdt_resampler_param_name(self._handle, i).decode():
                                 (int(lib.mdt_resampler_grad_offset(self._handle, i)),
                                  int(lib.mdt_resampler_param_numel(self._handle, i))) for i in range(n)}
            self._grad_numel = int(lib.mdt_resampler_grad_numel(self._handle))
        stream = torch.cuda.current_stream(device).cuda_stream
        for name, p in self.named_parameters():
            tag = (p.data_ptr(), p._version)
            if self._uploaded.get(name) == tag:
                continue
            if p.device != device or p.dtype != torch.float32:
                raise RuntimeError(f"parameter {name} must be float32 on {device}, got {p.dtype} on {p.device}")
            src = p.detach().contiguous()
            _lib.check(lib.mdt_resampler_load_param(self._handle, name.encode(), src.data_ptr(), src.numel(), stream))
            self._uploaded[name] = tag
        return lib, stream

    # -- reference API -----------------------------------------------------------------------------
    def forward(self, x_f: torch.Tensor, mask: Optional[torch.Tensor] = None) -> torch.Tensor:
        """x_f (batch, n_frames, n_features, dim), mask (batch, n_frames) bool -> (batch, num_latents, dim)
        (reference perceiver_resampler.py:124-162)."""
        assert x_f.ndim == 4
        batch_size, max_length, n_features, dim = x_f.shape
        assert dim == self.dim
        m = None
        if mask is not None:
            m = mask.to(device=x_f.device, dtype=torch.bool).reshape(batch_size, max_length).contiguous().view(torch.uint8)
        if torch.is_grad_enabled() and (x_f.requires_grad or any(p.requires_grad for p in self.parameters())):
            if x_f.device.type != "cuda":
                self._engine(x_f.device)  # raises the no-CPU-path error
            x = x_f if (x_f.dtype == torch.float32 and x_f.is_contiguous() and x_f.data_ptr() % 16 == 0) \
                else x_f.float().contiguous().clone()
            named = list(self.named_parameters())
            return _ResamplerFn.apply(self, x, m, [k for k, _ in named], *[p for _, p in named])
        lib, stream = self._engine(x_f.device)
        x = x_f.detach()
        if x.dtype != torch.float32:
            x = x.float()
        if not x.is_contiguous() or x.data_ptr() % 16:
            x = x.contiguous().clone()
        out = torch.empty((batch_size, self.num_queries, dim), device=x.device, dtype=torch.float32)
        _lib.check(lib.mdt_resampler_forward(self._handle, x.data_ptr(), None if m is None else m.data_ptr(),
                                             batch_size, max_length, n_features, out.data_ptr(), stream))
        return out

    def flops(self, n_frames: int, n_features: int) -> float:
        """Algorithmic FLOPs of one forward per sample (needs a live handle, i.e. a previous forward)."""
        if self._handle is None:
            raise RuntimeError("call forward() once first")
        return float(_lib.load().mdt_resampler_flops(self._handle, int(n_frames), int(n_features)))
