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


# ------------------------------------------------------------------------------------------------------------------
# The same backward as a CHAIN of autograd nodes, one per stage of mdt_train_loss_bwd_stage (round 6).
#
# With one node, every parameter gradient appears at once when the whole backward has been enqueued: under
# DistributedDataParallel (reference mdt/training.py:74-79, Lightning's "ddp" strategy) no bucket could be reduced before the
# last kernel, and the 90 MB all-reduce sat fully exposed behind the backward.  Here node k runs stage k -- the last decoder
# block first, the token embeddings last -- and returns the gradients of exactly the parameters that stage completes, so their
# AccumulateGrad hooks fire (and DDP launches the bucket's reduction, ordered behind the stream as it stands at that moment)
# while the later stages have not even been enqueued.
#
# Graph: stage n-1's node is created first and owns the encoder inputs; each node hands a zero-dimensional `link` tensor to the
# next; the loss node (stage 0) consumes link_1.  The engine therefore runs stage 0, 1, ..., n-1 in this order.
# ------------------------------------------------------------------------------------------------------------------
class _StagedRun:
    """What the nodes of one forward share: the engine, the tape, and -- once stage 0 has run -- the buffers of the backward."""

    def __init__(self, eng):
        self.eng, self.tape, self.buf = eng, None, None
        self.stage_params = {}   # stage -> [(name, parameter)]
        self.unused = None

    def grads_of(self, stage):
        return self.eng.param_grads(self.buf[0], self.stage_params.get(stage, []), self.unused)

    def finish(self):
        if self.tape is not None:
            self.tape.release()
        self.buf = None


class HipLossStage(torch.autograd.Function):
    """Stage k >= 1 of the staged backward; the forward only threads the chain."""

    @staticmethod
    def forward(ctx, run, stage, last, link, tok, tok2, goal, *params):
        ctx.run, ctx.stage, ctx.last = run, stage, last
        ctx.set_materialize_grads(False)
        return torch.zeros((), device=run.eng.device)

    @staticmethod
    def backward(ctx, _g):
        run = ctx.run
        if run.buf is None:
            raise RuntimeError("stage of a HIP backward whose loss node has not run (or was already consumed)")
        eng = run.eng
        eng.train_loss_bwd_stage(run.tape.id, ctx.stage, *run.buf)
        grads = run.grads_of(ctx.stage)
        d_link = torch.zeros((), device=eng.device) if not ctx.last else None
        d_tok, d_tok2, d_goal = run.buf[1] if ctx.last else (None, None, None)
        if ctx.last:
            run.finish()
        return (None, None, None, d_link, d_tok, d_tok2, d_goal, *grads)


class HipDiffusionLossStaged(torch.autograd.Function):
    """The loss node of the chain: the whole forward, and stage 0 of the backward."""

    @staticmethod
    def forward(ctx, run, link, tok, tok2, goal, action, noise, sigma, drop, state, *params):
        eng = run.eng
        loss, mo, cx, tape = eng.train_loss_fwd(state, tok, tok2, goal, action, noise, sigma, drop)
        run.tape = _Tape(eng, tape)
        ctx.run = run
        ctx.inputs = (tok, tok2, goal)
        ctx.mark_non_differentiable(mo)
        ctx.set_materialize_grads(False)
        return loss, mo, cx

    @staticmethod
    def backward(ctx, g_loss, _g_mo, g_ctx):
        run = ctx.run
        eng = run.eng
        if run.tape is None or run.tape.id is None:
            raise RuntimeError("the HIP training tape of this forward was already consumed (no retain_graph support)")
        tok, tok2, goal = ctx.inputs
        if g_loss is None:  # only the context was used downstream
            g_loss = torch.zeros((), device=eng.device)
        run.buf = eng.train_loss_bwd_begin(g_loss, g_ctx, tok, tok2, goal, run.needs)
        eng.train_loss_bwd_stage(run.tape.id, 0, *run.buf)
        return (None, torch.zeros((), device=eng.device), None, None, None, None, None, None, None, None, *run.grads_of(0))


def staged_diffusion_loss(eng, state, tok, tok2, goal, action, noise, sigma, drop, names, params):
    """(loss, model_output, context) through the chain of stage nodes; same values and gradients as HipDiffusionLoss."""
    run = _StagedRun(eng)
    run.unused = eng.unused_goal_embedder(state, eng.cfg.arch == 0)  # MDT.forward always uses goal_emb
    run.needs = _needs(tok, tok2, goal)
    n = eng._n_stages
    for name, p in zip(names, params):
        run.stage_params.setdefault(eng._param_stage["inner_model." + name], []).append((name, p))
    # a leaf that requires grad heads the chain: every link then does, whatever the user froze, and no stage is skipped
    link = torch.zeros((), device=eng.device, requires_grad=True)
    for k in range(n - 1, 0, -1):  # the deepest stage first: it runs last in the backward
        last = k == n - 1
        ps = [p for _, p in run.stage_params.get(k, [])]
        link = HipLossStage.apply(run, k, last, link, tok if last else None, tok2 if last else None, goal if last else None, *ps)
    ps0 = [p for _, p in run.stage_params.get(0, [])]
    return HipDiffusionLossStaged.apply(run, link, tok.detach(), None if tok2 is None else tok2.detach(), goal.detach(), action, noise,
                                        sigma, drop, state, *ps0)


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
