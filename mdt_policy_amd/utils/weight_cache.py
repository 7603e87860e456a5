"""Keeping the library's packed weight images honest.

The facades re-upload a parameter when its storage pointer or autograd version counter changes.  torch's FUSED optimizers
(``torch.optim.AdamW(..., fused=True)``, fused SGD / Adam) update parameters without touching the version counter, so
after their ``step()`` nothing would look changed and the HIP path would keep computing with the old weights.  Every facade
module registers itself here; a process-wide optimizer post-step hook marks the registered modules that own one of the
stepped parameters dirty, and their next call re-uploads (one batched launch).

Writes that bypass both the counter and an optimizer (``p.data.copy_()``, apex / DeepSpeed kernels) still need an explicit
``module.mark_dirty()``; see INTEGRATION.md.
"""
from __future__ import annotations

import weakref

import torch

_tracked: "weakref.WeakSet" = weakref.WeakSet()
_hook = None


def _after_step(optimizer, args, kwargs):
    if not _tracked:
        return
    stepped = None
    for mod in list(_tracked):
        if stepped is None:
            stepped = {id(p) for group in optimizer.param_groups for p in group["params"]}
        if any(id(p) in stepped for p in mod.parameters()):
            mod.mark_dirty()


def track(module) -> None:
    """Register ``module`` (anything with ``parameters()`` and ``mark_dirty()``)."""
    global _hook
    _tracked.add(module)
    if _hook is None:
        from torch.optim.optimizer import register_optimizer_step_post_hook
        _hook = register_optimizer_step_post_hook(_after_step)
