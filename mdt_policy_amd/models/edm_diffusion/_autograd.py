"""torch.autograd bridges of the training path: the HIP forward keeps a tape in the library handle, the HIP
backward turns it into gradients of every parameter (views of one flat buffer, reference layout) and of the encoder
inputs, so ``loss.backward()``, any torch optimizer, EMA callbacks and DistributedDataParallel's gradient
all-reduce (RCCL) work on the facade exactly as on the reference modules.

Reference semantics: autograd through GCDenoiser.loss (score_wrappers.py:45-63) and
forward_context_only (:82-97), as driven by MDTVAgent.training_step (mdtv_agent.py:222-262).
"""
from __future__ import annotations

import torch


class _Tape:
    """Owns one library tape: released after the backward, or when the autograd node dies without one."""

    def __init__(self, eng, tape_id: int):
        self.eng, self.id = eng, tape_id

    def release(self):
        if self.id is not None:
            try:
                self.eng.tape_release(self.id)
            except Exception:
                pass  # handle already destroyed
            self.id = None

    __del__ = release


def _needs(*ts):
    return tuple(bool(t is not None and torch.is_tensor(t) and t.requires_grad) for t in ts)


class HipDiffusionLoss(torch.autograd.Function):
    """(loss, model_output, context) = GCDenoiser.loss(...); loss and context are differentiable."""

    @staticmethod
    def forward(ctx, eng, state, tok, tok2, goal, action, noise, sigma, drop, names, *params):
        loss, mo, cx, tape = eng.train_loss_fwd(state, tok, tok2, goal, action, noise, sigma, drop)
        ctx.eng, ctx.tape = eng, _Tape(eng, tape)
        ctx.named = list(zip(names, params))
        ctx.unused = eng.unused_goal_embedder(state, eng.cfg.arch == 0)  # MDT.forward always uses goal_emb
        ctx.inputs = (tok, tok2, goal)
        ctx.needs = _needs(tok, tok2, goal)
        ctx.mark_non_differentiable(mo)
        ctx.set_materialize_grads(False)
        return loss, mo, cx

    @staticmethod
    def backward(ctx, g_loss, _g_mo, g_ctx):
        eng = ctx.eng
        if ctx.tape.id is None:
            raise RuntimeError("the HIP training tape of this forward was already consumed (no retain_graph support)")
        tok, tok2, goal = ctx.inputs
        if g_loss is None:  # only the context was used downstream
            g_loss = torch.zeros((), device=eng.device)
        grads, d_tok, d_tok2, d_goal = eng.train_loss_bwd(ctx.tape.id, g_loss, g_ctx, tok, tok2, goal, ctx.needs)
        ctx.tape.release()
        return (None, None, d_tok, d_tok2, d_goal, None, None, None, None, None,
                *eng.param_grads(grads, ctx.named, ctx.unused))


class HipContextOnly(torch.autograd.Function):
    """context = GCDenoiser.forward_context_only(...), differentiable (CLA / MGF auxiliary losses hang on it)."""

    @staticmethod
    def forward(ctx, eng, state, tok, tok2, goal, honour, drop, sigma, names, *params):
        cx, tape = eng.train_encode_fwd(state, tok, tok2, goal, honour, drop, sigma)
        ctx.eng, ctx.tape = eng, _Tape(eng, tape)
        ctx.named = list(zip(names, params))
        ctx.unused = eng.unused_goal_embedder(state, honour)
        ctx.inputs = (tok, tok2, goal)
        ctx.needs = _needs(tok, tok2, goal)
        return cx

    @staticmethod
    def backward(ctx, g_ctx):
        eng = ctx.eng
        if ctx.tape.id is None:
            raise RuntimeError("the HIP training tape of this forward was already consumed (no retain_graph support)")
        tok, tok2, goal = ctx.inputs
        grads, d_tok, d_tok2, d_goal = eng.train_encode_bwd(ctx.tape.id, g_ctx, tok, tok2, goal, ctx.needs)
        ctx.tape.release()
        return (None, None, d_tok, d_tok2, d_goal, None, None, None, None, *eng.param_grads(grads, ctx.named, ctx.unused))
