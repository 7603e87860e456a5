"""The EMA callback's arithmetic on the MI355X (reference: mdt/callbacks/ema.py).

The reference's fast path is ``amp_C.multi_tensor_axpby`` from NVIDIA apex (ema.py:108-115) and it otherwise falls
back to a Python loop over every tensor (:117-126).  ``multi_tensor_ema`` is that fused update as ONE gfx950 launch
for the whole weight list; ``get_decay`` is the callback's power schedule (:84-91).  Plug-in: inside the reference's
``EMA.ema`` call ``multi_tensor_ema(self._ema_model_weights, list(pl_module.state_dict().values()),
self.get_decay(self._cur_step))``.
"""
from __future__ import annotations

from typing import Sequence

import torch

from .. import _lib


def get_decay(optimization_step: int, start_step: int = 0, inv_gamma: float = 1.0, power: float = 2 / 3,
              min_value: float = 0.0, max_value: float = 0.9999) -> float:
    """EMA.get_decay (reference ema.py:84-91): 1 - (1 + step/inv_gamma)^-power, clamped."""
    step = max(0, optimization_step - start_step - 1)
    value = 1 - (1 + step / inv_gamma) ** -power
    return max(min(value, max_value), min_value)


@torch.no_grad()
def multi_tensor_ema(ema_weights: Sequence[torch.Tensor], model_weights: Sequence[torch.Tensor], decay: float) -> None:
    """ema <- decay * ema + (1 - decay) * weight for every floating tensor; integer buffers are copied, as the
    reference's loop does (ema.py:120-121)."""
    if len(ema_weights) != len(model_weights):
        raise ValueError("ema_weights and model_weights differ in length")
    items = []
    for e, w in zip(ema_weights, model_weights):
        if not w.dtype.is_floating_point:
            e.copy_(w)
            continue
        if e.device.type != "cuda" or w.device != e.device or e.dtype != torch.float32 or w.dtype != torch.float32:
            raise RuntimeError("multi_tensor_ema updates float32 tensors on a ROCm GPU only")
        if not (e.is_contiguous() and w.is_contiguous()) or e.numel() != w.numel():
            raise RuntimeError("multi_tensor_ema needs contiguous tensors of equal size")
        items.append((e, w))
    if not items:
        return
    tab = (_lib.OptTensor * len(items))()
    for i, (e, w) in enumerate(items):
        tab[i] = _lib.OptTensor(p=w.data_ptr(), g=None, m=None, v=None, ema=e.data_ptr(), numel=w.numel())
    dev = items[0][0].device
    with torch.cuda.device(dev):
        _lib.check(_lib.load().mdt_op_multi_ema(tab, len(items), float(decay), torch.cuda.current_stream(dev).cuda_stream))
