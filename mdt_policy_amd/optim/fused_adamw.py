"""AdamW whose step() is ONE hand-written gfx950 launch per parameter group (``mdt_op_multi_adamw``).

Drop-in for the ``torch.optim.AdamW(optim_groups, lr=..., betas=...)`` the reference builds in
``MDTVAgent.configure_optimizers`` (mdt/models/mdtv_agent.py:164-199): same constructor arguments, same state_dict
layout (``step`` / ``exp_avg`` / ``exp_avg_sq`` per parameter), same arithmetic (decoupled weight decay, bias
correction).  fp32 CUDA(ROCm) parameters only; anything else raises -- there is no eager fallback here.
"""
from __future__ import annotations

import ctypes as C

import torch

from .. import _lib


class FusedAdamW(torch.optim.Optimizer):
    def __init__(self, params, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 1e-2):
        if lr < 0 or eps < 0 or not (0 <= betas[0] < 1) or not (0 <= betas[1] < 1) or weight_decay < 0:
            raise ValueError("invalid AdamW hyper-parameter")
        super().__init__(params, dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay))
        # host-side caches (not part of state_dict): the step count as a Python int and the addresses of the moment
        # buffers, so that a step makes two tensor calls per parameter instead of seven
        self._nsteps, self._mv = {}, {}

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self._nsteps, self._mv = {}, {}

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lib = _lib.load()
        for group in self.param_groups:
            by_step, steps = {}, []
            for p in group["params"]:
                if p.grad is None:
                    continue
                if p.device.type != "cuda" or p.dtype != torch.float32 or p.grad.dtype != torch.float32:
                    raise RuntimeError("FusedAdamW updates float32 parameters on a ROCm GPU only")
                if p.grad.is_sparse:
                    raise RuntimeError("FusedAdamW does not support sparse gradients")
                if not p.is_contiguous():
                    raise RuntimeError("FusedAdamW needs contiguous parameters")
                st = self.state[p]
                if not st:
                    st["step"] = torch.tensor(0.0)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                n = self._nsteps.get(p)
                mv = self._mv.get(p)
                if n is None or mv is None or mv[0] is not st["exp_avg"] or mv[1] is not st["exp_avg_sq"]:
                    n = int(st["step"].item())
                    self._mv[p] = (st["exp_avg"], st["exp_avg_sq"], st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(),
                                   p.numel())
                self._nsteps[p] = n + 1
                steps.append(st["step"])
                g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                by_step.setdefault(n + 1, []).append((p, g))
            if steps:
                torch._foreach_add_(steps, 1.0)  # the state_dict's per-parameter counters, one call for all of them
            for step, items in by_step.items():  # parameters that joined later carry their own step count
                tab = (_lib.OptTensor * len(items))()
                for i, (p, g) in enumerate(items):
                    _, _, m_ptr, v_ptr, numel = self._mv[p]
                    tab[i] = _lib.OptTensor(p=p.data_ptr(), g=g.data_ptr(), m=m_ptr, v=v_ptr, ema=None, numel=numel)
                dev = items[0][0].device
                with torch.cuda.device(dev):
                    _lib.check(lib.mdt_op_multi_adamw(tab, len(items), float(group["lr"]), float(group["betas"][0]),
                                                      float(group["betas"][1]), float(group["eps"]),
                                                      float(group["weight_decay"]), step,
                                                      torch.cuda.current_stream(dev).cuda_stream))
                # the library wrote the parameters behind autograd's back: bump their version counters (no kernel) so
                # that the HIP engines see the change and re-upload the weights before the next forward
                ps = [p for p, _ in items]
                try:
                    torch._C._autograd._unsafe_set_version_counter(ps, [p._version + 1 for p in ps])
                except (AttributeError, TypeError):  # other torch builds: an in-place no-op bumps it the public way
                    torch._foreach_add_(ps, 0.0)
        return loss
