"""Bridge between the nn.Module facades and libmdt_hip.so.

``HipEngine`` owns one ``mdt_model`` handle (include/mdt_hip.h) for one score network on one GPU.  It keeps the
library's fragment-packed weight arena in sync with the module's ``nn.Parameter``s (re-uploading a parameter
when its storage pointer or in-place version counter changes, e.g. after ``load_state_dict`` / an EMA copy /
an optimizer step) and turns tensors into the raw device pointers + stream the C ABI takes.

``HipScoreNetwork`` is the shared base of ``MDTVTransformer`` / ``MDTTransformer`` (reference
mdt/models/networks/mdtv_transformer.py:35, mdt/models/networks/mdt_transformer.py:38).

There is no eager/CPU fallback here on purpose: a missing library, a CPU tensor, or an unsupported
configuration raises.
"""
from __future__ import annotations

import ctypes as C
import operator
from typing import Dict, Optional

import torch
import torch.nn as nn

from ... import _lib


import os

_RESYNC_EVERY = int(os.environ.get("MDT_HIP_PARAM_RESYNC", "0") or 0)


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


class HipEngine:
    def __init__(self, module: nn.Module, cfg: _lib.MDTConfig, device: torch.device):
        if device.type != "cuda":
            raise RuntimeError(
                "the MDT denoiser runs only on a ROCm GPU (hand-written gfx950 kernels); move the model with "
                ".to('cuda') -- there is no CPU execution path in mdt_policy_amd")
        self.lib = _lib.load()
        self.module = module
        self.device = device
        self.cfg = cfg
        handle = C.c_void_p()
        from ...utils import torch_allocator
        torch_allocator.install()  # workspace / tapes / scratch live in torch's caching allocator, not beside it
        with torch.cuda.device(device):
            _lib.check(self.lib.mdt_create(C.byref(cfg), C.byref(handle)))
        self.handle = handle
        n = self.lib.mdt_param_count(handle)
        self.expected = {self.lib.mdt_param_name(handle, i).decode(): self.lib.mdt_param_numel(handle, i)
                         for i in range(n)}
        self._uploaded: Dict[str, tuple] = {}
        self.ctx_generation = 0  # bumped by every call that rewrites the handle's cached context (encoder output, K|V, folds)
        self.sigma_in_context = not cfg.use_ada_conditioning  # sigma embedding is the first context token
        # goal_conditioned=False: MDTV keeps the goal token (behind the state tokens), MDT has none
        # (and MDTV gives it up when the proprioceptive token takes that place, mdtv_transformer.py:291-294)
        self.proprio = bool(cfg.use_proprio)
        self.has_goal_token = not (cfg.no_goal_conditioning and (cfg.arch == 1 or self.proprio))
        self.Te = int(self.sigma_in_context) + int(self.has_goal_token) + (cfg.n_obs_token if cfg.arch == 0 else 2) + \
            int(self.proprio)
        self.Ta, self.A, self.D = cfg.action_seq_len, cfg.action_dim, cfg.embed_dim

    def __del__(self):
        try:
            if getattr(self, "handle", None) is not None and self.handle.value:
                self.lib.mdt_destroy(self.handle)
                self.handle = C.c_void_p()
        except Exception:
            pass

    # ------------------------------------------------------------------------------------------
    def _stream(self) -> int:
        return torch.cuda.current_stream(self.device).cuda_stream

    def invalidate(self) -> None:
        """Forget what was uploaded: the next call re-uploads (and re-packs) every parameter.  For in-place weight
        updates the version counter does not see -- ``p.data.mul_()/copy_()``, apex / DeepSpeed multi-tensor kernels, any
        raw-pointer write other than this package's FusedAdamW / multi_tensor_ema (which bump the counter)."""
        self._uploaded.clear()
        self._fast = None

    def sync_params(self) -> None:
        """Upload every parameter whose storage or version changed since the last upload.  MDT_HIP_PARAM_RESYNC=N
        (debug safety net) additionally re-uploads everything every N-th call."""
        stream = self._stream()
        if _RESYNC_EVERY > 0:
            self._calls = getattr(self, "_calls", 0) + 1
            if self._calls % _RESYNC_EVERY == 0:
                self._uploaded.clear()
        # fast path (a rollout calls this ~1000 times a second with nothing changed): three C-level sweeps over the cached lists --
        # every parameter still the object its module holds, same version counters, same storage pointers -- instead of a Python
        # loop body per parameter (73 -> ~45 us per call of a 1.2 ms rollout step)
        fast = getattr(self, "_fast", None)
        if fast is not None and not self._dirty_fast():
            return
        seen = 0
        changed = []  # (key, source tensor, tag)
        # the parameters the library reads, resolved once per engine as (key, leaf name, owning module, parameter): walking the
        # module tree on every call was ~0.15 ms of host time.  A parameter REPLACED by assignment fails the identity check
        # below and the list is resolved again.
        plist = getattr(self, "_plist", None)
        if plist is None:
            plist = self._plist = []
            for n, p in self.module.named_parameters():
                if ("inner_model." + n) in self.expected:
                    owner = self.module.get_submodule(n.rsplit(".", 1)[0]) if "." in n else self.module
                    plist.append(("inner_model." + n, n.rsplit(".", 1)[-1], owner, p))
        uploaded = self._uploaded
        for key, leaf, owner, p in plist:
            if owner._parameters.get(leaf) is not p:
                self._plist = None
                return self.sync_params()
            seen += 1
            tag = (p.data_ptr(), p._version)
            name = key
            if uploaded.get(key) == tag:
                continue
            if p.device != self.device or p.dtype != torch.float32:
                raise RuntimeError(f"parameter {name} must be float32 on {self.device}, got {p.dtype} on {p.device}")
            src = p.detach()
            if not src.is_contiguous():
                src = src.contiguous()
            changed.append((key, src, tag))
        if changed:  # every changed parameter in ONE launch (a training step changes all of them)
            n = len(changed)
            names = (C.c_char_p * n)(*[k.encode() for k, _, _ in changed])
            srcs = (C.c_void_p * n)(*[t.data_ptr() for _, t, _ in changed])
            numels = (C.c_int64 * n)(*[t.numel() for _, t, _ in changed])
            _lib.check(self.lib.mdt_load_params(self.handle, n, names, srcs, numels, stream))
            for key, _, tag in changed:
                self._uploaded[key] = tag
            self._keep_src = [t for _, t, _ in changed]  # non-contiguous parameters were copied: keep those alive
        if seen != len(self.expected):
            have = {"inner_model." + n for n, _ in self.module.named_parameters()}
            raise RuntimeError(f"module lacks parameters the HIP path needs: {sorted(set(self.expected) - have)[:5]}")
        params = [p for _, _, _, p in plist]
        if all(self._uploaded.get(k) == (p.data_ptr(), p._version) for k, _, _, p in plist):
            self._fast = ([o._parameters for _, _, o, _ in plist], [l for _, l, _, _ in plist], params,
                          [p._version for p in params], [p.data_ptr() for p in params], plist)
        else:
            self._fast = None

    def _dirty_fast(self) -> bool:
        """True when the cached snapshot of sync_params no longer describes the module (or cannot tell)."""
        dicts, leafs, params, vers, ptrs, plist = self._fast
        if plist is not getattr(self, "_plist", None) or len(self._uploaded) < len(params):   # invalidate() / a re-resolved list
            self._fast = None
            return True
        try:
            same = all(map(operator.is_, [d[l] for d, l in zip(dicts, leafs)], params))
        except KeyError:
            same = False
        if same and [p._version for p in params] == vers and [p.data_ptr() for p in params] == ptrs:
            return False
        self._fast = None
        return True

    def _in(self, t: torch.Tensor, shape=None) -> torch.Tensor:
        if t.device != self.device:
            raise RuntimeError(f"input tensor on {t.device}, model on {self.device}")
        t = t.detach()
        if t.dtype != torch.float32:
            t = t.float()
        if shape is not None:
            t = t.reshape(shape)
        if not t.is_contiguous():
            t = t.contiguous()
        if t.data_ptr() % 16:
            t = t.clone()
        return t

    def _tokens(self, state: dict):
        if self.cfg.arch == 0:
            tok = self._in(state["state_images"])
            if tok.dim() != 3 or tok.shape[1] != self.cfg.n_obs_token or tok.shape[2] != self.cfg.obs_dim:
                raise ValueError(f"state_images must be (B,{self.cfg.n_obs_token},{self.cfg.obs_dim}), got {tuple(tok.shape)}")
            # 'state_obs' in states (mdtv_transformer.py:262) selects the context layout: HipScoreNetwork.hip_engine
            # hands out the handle built for this state dict's layout
            if ("state_obs" in state) != self.proprio:
                raise RuntimeError("state dict and library handle disagree about the proprioceptive token; obtain the "
                                   "engine with hip_engine(state=state)")
            obs = None
            if self.proprio:
                obs = self._in(state["state_obs"])
                if obs.numel() != tok.shape[0] * self.cfg.proprio_dim:
                    raise ValueError(f"state_obs must be (B,1,{self.cfg.proprio_dim}), got {tuple(obs.shape)}")
            return tok, obs, tok.shape[0]
        st, gr = self._in(state["static"]), self._in(state["gripper"])
        B = st.shape[0]
        if st.numel() != B * self.cfg.obs_dim or gr.numel() != B * self.cfg.obs_dim:
            raise ValueError("static / gripper must be (B,1,obs_dim)")
        return st, gr, B

    def _goal(self, goal: torch.Tensor, B: int) -> torch.Tensor:
        g = self._in(goal)
        if g.numel() != B * self.cfg.goal_dim:
            raise ValueError(f"goal must hold (B,1,{self.cfg.goal_dim}) values, got {tuple(goal.shape)}")
        return g

    def _modality(self, state: dict) -> int:
        return _lib.MODALITY["lang"] if state.get("modality", None) == "lang" else _lib.MODALITY["vis"]

    # ------------------------------------------------------------------------------------------
    def encode(self, state: dict, goal: torch.Tensor, honour_modality: bool,
               sigma: Optional[torch.Tensor] = None) -> torch.Tensor:
        """``sigma`` is only read by use_ada_conditioning=False models (it is their first context token)."""
        self.sync_params()
        tok, tok2, B = self._tokens(state)
        g = self._goal(goal, B)
        s = None
        if self.sigma_in_context:
            if sigma is None:
                raise ValueError("use_ada_conditioning=False: the encoder needs sigma (its first context token)")
            s = self._in(sigma.reshape(-1).expand(B) if sigma.numel() == 1 else sigma, (B,))
        ctx = torch.empty((B, self.Te, self.D), device=self.device, dtype=torch.float32)
        self.ctx_generation += 1
        _lib.call(self.lib.mdt_encode, self.handle, _ptr(tok), _ptr(tok2), _ptr(g), self._modality(state),
                                       int(honour_modality), _ptr(s), B, _ptr(ctx), self._stream())
        return ctx

    def denoise_cached(self, x: torch.Tensor, sigma: torch.Tensor, flags: int = 0) -> torch.Tensor:
        x_ = self._in(x)
        B = x_.shape[0]
        if sigma.numel() == 1 and B > 1:  # one noise level for the whole batch (every sampler): broadcast in-kernel
            s = self._in(sigma, (1,))
            flags |= _lib.SIGMA_SCALAR
        else:
            s = self._in(sigma, (B,))
        out = torch.empty((B, self.Ta, self.A), device=self.device, dtype=torch.float32)
        _lib.call(self.lib.mdt_denoise_cached, self.handle, _ptr(x_), _ptr(s), B, flags, _ptr(out), self._stream())
        return out

    def forward(self, state: dict, x: torch.Tensor, goal: torch.Tensor, sigma: torch.Tensor):
        self.sync_params()
        tok, tok2, B = self._tokens(state)
        if sigma.numel() == 1 and B > 1:
            sigma = sigma.reshape(1).expand(B)
        g, x_, s = self._goal(goal, B), self._in(x, (B, self.Ta, self.A)), self._in(sigma, (B,))
        out = torch.empty((B, self.Ta, self.A), device=self.device, dtype=torch.float32)
        ctx = torch.empty((B, self.Te, self.D), device=self.device, dtype=torch.float32)
        self.ctx_generation += 1
        _lib.call(self.lib.mdt_forward, self.handle, _ptr(tok), _ptr(tok2), _ptr(g), self._modality(state), _ptr(x_),
                                        _ptr(s), B, _ptr(out), _ptr(ctx), self._stream())
        return out, ctx

    def sample_ddim(self, state: dict, x_T: torch.Tensor, goal: torch.Tensor, sigmas):
        """Fused sampler call.  ``sigmas`` may live on the host (gc_sampling's default) or on the model's device --
        the agent builds its schedule there (mdtv_agent.py:660-667); a device schedule is consumed in place
        (mdt_sample_ddim_dev): no copy to the host, no synchronisation."""
        self.sync_params()
        tok, tok2, B = self._tokens(state)
        g, x_ = self._goal(goal, B), self._in(x_T, (B, self.Ta, self.A))
        out = torch.empty((B, self.Ta, self.A), device=self.device, dtype=torch.float32)
        ctx = torch.empty((B, self.Te, self.D), device=self.device, dtype=torch.float32)
        if torch.is_tensor(sigmas) and sigmas.device.type == "cuda":
            sig = self._in(sigmas.reshape(-1))
            n = sig.numel() - 1
            self._keep = sig  # the kernel that reads it is only enqueued: keep the (possibly converted) tensor alive
            self.ctx_generation += 1
            _lib.call(self.lib.mdt_sample_ddim_dev, self.handle, _ptr(tok), _ptr(tok2), _ptr(g), self._modality(state),
                                                    _ptr(x_), _ptr(sig), n, B, _ptr(out), _ptr(ctx), self._stream())
            return out, ctx
        sig = [float(v) for v in (sigmas.detach().tolist() if torch.is_tensor(sigmas) else sigmas)]
        n = len(sig) - 1
        arr = (C.c_float * len(sig))(*sig)
        self.ctx_generation += 1
        _lib.call(self.lib.mdt_sample_ddim, self.handle, _ptr(tok), _ptr(tok2), _ptr(g), self._modality(state),
                                            _ptr(x_), arr, n, B, _ptr(out), _ptr(ctx), self._stream())
        return out, ctx

    def loss_fwd(self, state: dict, action: torch.Tensor, goal: torch.Tensor, noise: torch.Tensor, sigma: torch.Tensor):
        self.sync_params()
        tok, tok2, B = self._tokens(state)
        g = self._goal(goal, B)
        a, nz = self._in(action, (B, self.Ta, self.A)), self._in(noise, (B, self.Ta, self.A))
        s = self._in(sigma, (B,))
        loss = torch.empty((), device=self.device, dtype=torch.float32)
        mo = torch.empty((B, self.Ta, self.A), device=self.device, dtype=torch.float32)
        ctx = torch.empty((B, self.Te, self.D), device=self.device, dtype=torch.float32)
        self.ctx_generation += 1
        _lib.call(self.lib.mdt_loss_fwd, self.handle, _ptr(tok), _ptr(tok2), _ptr(g), self._modality(state), _ptr(a),
                                         _ptr(nz), _ptr(s), B, _ptr(loss), _ptr(mo), _ptr(ctx), self._stream())
        return loss, mo, ctx

    def reserve(self, max_batch: int) -> None:
        _lib.call(self.lib.mdt_reserve, self.handle, int(max_batch))

    # -- training path (include/mdt_hip_train.h) ----------------------------------------------------
    def train_prepare(self) -> None:
        """Allocate the transposed weight images and learn the gradient layout (idempotent)."""
        if getattr(self, "_grad_layout", None) is not None:
            return
        with torch.cuda.device(self.device):
            _lib.call(self.lib.mdt_train_prepare, self.handle)
        self._uploaded.clear()  # every parameter is re-uploaded so that its transposed image exists
        n = self.lib.mdt_param_count(self.handle)
        self._grad_layout = {self.lib.mdt_param_name(self.handle, i).decode():
                             (int(self.lib.mdt_grad_offset(self.handle, i)), int(self.lib.mdt_param_numel(self.handle, i)))
                             for i in range(n)}
        self._grad_numel = int(self.lib.mdt_grad_numel(self.handle))
        # the stage of the staged backward (mdt_train_loss_bwd_stage) that completes each parameter's gradient
        self._n_stages = int(self.lib.mdt_train_loss_bwd_stages(self.handle))
        self._param_stage = {self.lib.mdt_param_name(self.handle, i).decode(): int(self.lib.mdt_train_param_stage(self.handle, i))
                             for i in range(n)}
        self.stages_enqueued = 0  # stages of staged backwards enqueued so far (tests read it from DDP communication hooks)

    def train_loss_fwd(self, state: dict, tok, tok2, goal, action, noise, sigma, drop=None):
        self.train_prepare()
        self.sync_params()
        B = tok.shape[0]
        loss = torch.empty((), device=self.device, dtype=torch.float32)
        mo = torch.empty((B, self.Ta, self.A), device=self.device, dtype=torch.float32)
        ctx = torch.empty((B, self.Te, self.D), device=self.device, dtype=torch.float32)
        tape = C.c_int32(-1)
        self.ctx_generation += 1
        _lib.call(self.lib.mdt_train_loss_fwd, self.handle, _ptr(tok), _ptr(tok2), _ptr(goal), self._modality(state),
                                               _ptr(action), _ptr(noise), _ptr(sigma), B,
                                               None if drop is None else C.byref(drop), _ptr(loss), _ptr(mo), _ptr(ctx),
                                               C.byref(tape), self._stream())
        return loss, mo, ctx, int(tape.value)

    def train_encode_fwd(self, state: dict, tok, tok2, goal, honour_modality: bool, drop=None, sigma=None):
        self.train_prepare()
        self.sync_params()
        B = tok.shape[0]
        ctx = torch.empty((B, self.Te, self.D), device=self.device, dtype=torch.float32)
        tape = C.c_int32(-1)
        if self.sigma_in_context and sigma is None:
            raise ValueError("use_ada_conditioning=False: sigma is a context token and must be given")
        self.ctx_generation += 1
        _lib.call(self.lib.mdt_train_encode_fwd, self.handle, _ptr(tok), _ptr(tok2), _ptr(goal), self._modality(state),
                                                 int(honour_modality), _ptr(sigma) if self.sigma_in_context else None, B,
                                                 None if drop is None else C.byref(drop), _ptr(ctx), C.byref(tape),
                                                 self._stream())
        return ctx, int(tape.value)

    def _input_grads(self, tok, tok2, goal, needs):
        d_tok = torch.empty_like(tok) if needs[0] else None
        d_tok2 = torch.empty_like(tok2) if (needs[1] and tok2 is not None) else None
        d_goal = torch.empty_like(goal) if needs[2] else None
        return d_tok, d_tok2, d_goal

    def train_loss_bwd(self, tape: int, g_loss, g_ctx, tok, tok2, goal, needs):
        grads = torch.zeros(self._grad_numel, device=self.device, dtype=torch.float32)
        d_tok, d_tok2, d_goal = self._input_grads(tok, tok2, goal, needs)
        gl = None if g_loss is None else self._in(g_loss, ())
        gc = None if g_ctx is None else self._in(g_ctx)
        _lib.call(self.lib.mdt_train_loss_bwd, self.handle, tape, _ptr(gl), _ptr(gc), _ptr(grads), _ptr(d_tok),
                                               _ptr(d_tok2), _ptr(d_goal), self._stream())
        return grads, d_tok, d_tok2, d_goal

    def train_loss_bwd_begin(self, g_loss, g_ctx, tok, tok2, goal, needs):
        """Buffers of one staged backward: (flat gradient buffer, input gradients, argument tuple of every stage call)."""
        grads = torch.zeros(self._grad_numel, device=self.device, dtype=torch.float32)
        d_tok, d_tok2, d_goal = self._input_grads(tok, tok2, goal, needs)
        gl = None if g_loss is None else self._in(g_loss, ())
        gc = None if g_ctx is None else self._in(g_ctx)
        return grads, (d_tok, d_tok2, d_goal), (gl, gc)

    def train_loss_bwd_stage(self, tape: int, stage: int, grads, dins, gs) -> None:
        """Enqueue stage `stage`; behind it (stream order) the gradients of the parameters of that stage are complete."""
        _lib.call(self.lib.mdt_train_loss_bwd_stage, self.handle, tape, int(stage), _ptr(gs[0]), _ptr(gs[1]), _ptr(grads),
                                                     _ptr(dins[0]), _ptr(dins[1]), _ptr(dins[2]), self._stream())
        self.stages_enqueued += 1

    def train_encode_bwd(self, tape: int, g_ctx, tok, tok2, goal, needs):
        grads = torch.zeros(self._grad_numel, device=self.device, dtype=torch.float32)
        d_tok, d_tok2, d_goal = self._input_grads(tok, tok2, goal, needs)
        gc = self._in(g_ctx)
        _lib.call(self.lib.mdt_train_encode_bwd, self.handle, tape, _ptr(gc), _ptr(grads), _ptr(d_tok), _ptr(d_tok2),
                                                 _ptr(d_goal), self._stream())
        return grads, d_tok, d_tok2, d_goal

    def denoise_vjp(self, state: dict, x: torch.Tensor, goal: torch.Tensor, sigma: torch.Tensor, v: torch.Tensor):
        """(D(x; sigma), (dD/dx)^T v): the eval-mode denoiser and its vector-Jacobian product w.r.t. the noisy actions
        (mdt_denoise_vjp: a tape-keeping forward plus an input-gradient-only backward)."""
        self.train_prepare()
        self.sync_params()
        tok, tok2, B = self._tokens(state)
        g = self._goal(goal, B)
        x_, v_ = self._in(x, (B, self.Ta, self.A)), self._in(v, (B, self.Ta, self.A))
        s = self._in(sigma.reshape(-1).expand(B) if sigma.numel() == 1 else sigma, (B,))
        den = torch.empty((B, self.Ta, self.A), device=self.device, dtype=torch.float32)
        vjp = torch.empty_like(den)
        self.ctx_generation += 1
        _lib.call(self.lib.mdt_denoise_vjp, self.handle, _ptr(tok), _ptr(tok2), _ptr(g), self._modality(state), _ptr(x_),
                                            _ptr(s), _ptr(v_), B, _ptr(den), _ptr(vjp), self._stream())
        return den, vjp

    def tape_release(self, tape: int) -> None:
        _lib.check(self.lib.mdt_tape_release(self.handle, tape))

    def unused_goal_embedder(self, state: dict, honour_modality: bool) -> Optional[str]:
        """Name prefix(es) of the goal embedder(s) this forward did NOT go through (None: all were used)."""
        if not self.has_goal_token:
            return ("goal_emb.", "lang_emb.")
        if not self.cfg.use_modality_encoder:
            return None
        lang = honour_modality and self._modality(state) == _lib.MODALITY["lang"]
        return "goal_emb." if lang else "lang_emb."

    def param_grads(self, grads: torch.Tensor, params, unused: Optional[str] = None):
        """Views of the flat gradient buffer, one per (name, parameter) of the module; None where this forward did
        not read the parameter (proprio_emb, the other modality's goal embedder), as autograd leaves it in the
        reference -- optimizers skip such parameters instead of decaying them."""
        out = []
        for name, p in params:
            ent = self._grad_layout.get("inner_model." + name)
            if ent is None or not p.requires_grad or (unused and name.startswith(unused)):
                out.append(None)
            else:
                out.append(grads[ent[0]:ent[0] + ent[1]].view(p.shape))
        return out

    def flops_per_chunk(self, n_steps: int) -> float:
        return float(self.lib.mdt_flops_per_chunk(self.handle, int(n_steps)))


class HipScoreNetwork(nn.Module):
    """Common facade logic of the two score networks; subclasses build the parameter tree in reference order."""

    _arch = "mdtv"

    def _init_common(self):
        from ...utils import weight_cache
        weight_cache.track(self)  # fused torch optimizers do not bump version counters: an optimizer hook marks us dirty
        self.latent_encoder_emb = None
        self._engines: Dict[tuple, HipEngine] = {}
        self._sigma_data = 1.0

    # -- configuration handed to the library ---------------------------------------------------
    def _hip_config(self, sigma_data: float) -> _lib.MDTConfig:
        raise NotImplementedError

    def hip_engine(self, sigma_data: Optional[float] = None, state: Optional[dict] = None) -> HipEngine:
        """The (lazily created) library handle.  sigma_data defaults to the value the owning GCDenoiser
        registered (``_sigma_data``); it only matters for the preconditioned entry points.  ``state``: the state dict of
        the call -- MDTVTransformer embeds a proprioceptive ``state['state_obs']`` into one more context token when the
        key is present (mdtv_transformer.py:260-266); the two context layouts live in two handles."""
        if sigma_data is None:
            sigma_data = getattr(self, "_sigma_data", 1.0)
        p = next(self.parameters())
        proprio = self._arch == "mdtv" and state is not None and "state_obs" in state
        key = (str(p.device), float(sigma_data), proprio)
        eng = self._engines.get(key)
        if eng is None:
            eng = HipEngine(self, self._hip_config(float(sigma_data), proprio), p.device)
            # a moved / re-scaled model drops the old arenas; the other context layout of the same model stays
            self._engines = {k: e for k, e in self._engines.items() if k[:2] == key[:2]}
            self._engines[key] = eng
        return eng

    def __getstate__(self):  # copy.deepcopy / pickle (EMA copies, checkpoints of whole modules): never the library handle
        d = self.__dict__.copy()
        d["_engines"] = {}
        d.pop("_ctx_engine", None)
        return d

    def __setstate__(self, state):  # the copy is a new module: the optimizer hook has to know it, and it owns no images yet
        super().__setstate__(state)
        from ...utils import weight_cache
        weight_cache.track(self)
        self.mark_dirty()

    def mark_dirty(self) -> None:
        """Public escape hatch of the weight cache (see HipEngine.invalidate): call after writing parameters through
        ``.data`` or a foreign fused optimizer."""
        for eng in getattr(self, "_engines", {}).values():
            eng.invalidate()

    def train(self, mode: bool = True):  # mode switches are rare and a natural point to re-validate the arena
        out = super().train(mode)
        self.mark_dirty()
        return out

    def _load_from_state_dict(self, *a, **kw):
        super()._load_from_state_dict(*a, **kw)
        self.mark_dirty()

    def _apply(self, fn, *a, **kw):  # .to()/.cuda()/.float(): parameters are re-created, drop stale handles
        out = super()._apply(fn, *a, **kw)
        self._engines = {}
        self._ctx_engine = None
        return out

    def train_dropout(self):
        """mdt_dropout of one train-mode forward (fresh seed from torch's CPU generator, so torch.manual_seed makes
        runs repeatable), or None in eval mode / without dropout."""
        embed_p, attn_p, resid_p, mlp_p, goal_p = (float(p) for p in self._pdrops)
        if not self.training or max(attn_p, resid_p, mlp_p, embed_p) <= 0:
            return None
        seed = int(torch.randint(1, 2 ** 62, (1,)).item())
        return _lib.Dropout(attn_p=attn_p, resid_p=resid_p, mlp_p=mlp_p, embed_p=embed_p, seed=seed)

    def _guard_mode(self, allow_grad: bool = False):
        embed_p, attn_p, resid_p, mlp_p, goal_p = (float(p) for p in self._pdrops)
        if self.training and max(attn_p, resid_p, mlp_p, embed_p) > 0 and not (allow_grad and torch.is_grad_enabled()):
            raise NotImplementedError(
                "train() mode with dropout > 0 outside a training step: only GCDenoiser.loss / forward_context_only "
                "under autograd apply dropout on the HIP path; call .eval() for inference")
        if not allow_grad and torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            # forward values are exact, but no autograd graph is recorded
            raise NotImplementedError(
                "autograd through the HIP denoiser is not implemented yet (SURVEY.md 8(f) item 1); wrap the call "
                "in torch.no_grad() for inference / loss evaluation")

    # -- reference API ---------------------------------------------------------------------------
    def _goals(self, goals: torch.Tensor, uncond: bool) -> torch.Tensor:
        """preprocess_goals (reference mdtv_transformer.py:246-258) incl. mask_cond (:302-310): in train() mode every
        goal ELEMENT is zeroed with probability goal_drop -- the reference's own torch.bernoulli draw on the goal's
        device, so a seeded run masks the same elements."""
        if goals.dim() == 2:
            goals = goals[:, None, :]
        # a goal SEQUENCE is reduced to its first entry only where the reference does it: when it is as long as the state
        # sequence (mdtv_transformer.py:249 with states_length = n_obs_token, mdt_transformer.py:213/262 with 1); any
        # other length reaches the shape check of the engine and raises, as the reference's Linear would
        states_length = self.n_obs_token if self._arch == "mdtv" else 1
        if goals.shape[1] == states_length and goals.shape[1] != 1 and self.goal_seq_len == 1:
            goals = goals[:, :1, :]
        goal_p = float(self._pdrops[4])
        if self.training and goal_p > 0.0:
            mask = torch.bernoulli(torch.ones(goals.shape, device=goals.device) * goal_p)
            goals = goals * (1.0 - mask)
        if uncond:
            goals = torch.zeros_like(goals)
        return goals

    def forward(self, states, actions, goals, sigma, uncond: Optional[bool] = False):
        """Raw score network F(states, actions, goals, sigma) (reference mdtv_transformer.py:208-211)."""
        self._guard_mode()
        eng = self.hip_engine(state=states)
        ctx = eng.encode(states, self._goals(goals, uncond), honour_modality=self._arch == "mdtv", sigma=sigma)
        self.latent_encoder_emb = ctx
        self._ctx_engine = eng
        return eng.denoise_cached(actions, sigma, _lib.RAW_OUTPUT | _lib.RAW_INPUT)

    def forward_enc_only(self, states, actions=None, goals=None, sigma=None, uncond: Optional[bool] = False):
        """Context tokens (reference mdtv_transformer.py:213-222 / mdt_transformer.py:257-281)."""
        self._guard_mode()
        eng = self.hip_engine(state=states)
        ctx = eng.encode(states, self._goals(goals, uncond), honour_modality=True, sigma=sigma)
        self._ctx_engine = eng
        if self._arch == "mdtv":  # MDTTransformer.forward_enc_only does not cache (mdt_transformer.py:257-281)
            self.latent_encoder_emb = ctx
        return ctx

    def forward_dec_only(self, context, actions, sigma):
        """reference mdtv_transformer.py:224-236; only valid on the context produced by the last encoder call."""
        if context is not self.latent_encoder_emb:
            raise NotImplementedError("forward_dec_only needs the context tensor returned by the immediately "
                                      "preceding forward_enc_only()/forward() of this module")
        self._guard_mode()
        eng = getattr(self, "_ctx_engine", None) or self.hip_engine()  # the handle that holds this context
        return eng.denoise_cached(actions, sigma, _lib.RAW_OUTPUT | _lib.RAW_INPUT)

    def get_params(self):
        return self.parameters()

    def get_block_size(self):
        return self.block_size
