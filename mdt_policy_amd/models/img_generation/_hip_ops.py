"""HIP-backed autograd functions of the masked-image decoder: every Linear on the fp32-MFMA GEMM (mdt_op_gemm forward,
mdt_op_linear_bwd backward), RMSNorm / SwishGLU row kernels and the mid-length self-attention (include/mdt_mae.h).
No eager fallback: CPU tensors or a missing library raise."""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional

import torch

from ... import _lib

RMS_EPS = 1e-8


def _stream(t: torch.Tensor) -> int:
    if t.device.type != "cuda":
        raise RuntimeError("the masked-image decoder runs only on a ROCm GPU (hand-written gfx950 kernels); there is no "
                           "CPU execution path in mdt_policy_amd")
    return torch.cuda.current_stream(t.device).cuda_stream


def _c(t: torch.Tensor) -> torch.Tensor:
    t = t.detach()
    if t.dtype != torch.float32:
        t = t.float()
    if not t.is_contiguous():
        t = t.contiguous()
    if t.data_ptr() % 16:
        t = t.clone()
    return t


class PackedWeights:
    """Fragment-packed images of an nn.Linear-shaped weight (N, K): the forward operand and the image of W^T for
    dX = dY W, refreshed when the parameter's storage or version counter changes (``invalidate()`` forces it)."""

    def __init__(self):
        self._cache: Dict[int, tuple] = {}

    def invalidate(self) -> None:
        self._cache.clear()

    def refresh(self, weights, need_t: bool) -> None:
        """Bring the images of every weight in ``weights`` (nn.Linear / Conv2d weight parameters) up to date with ONE launch
        (mdt_op_pack_many); ``get`` then finds them cached.  The image buffers are kept across refreshes."""
        stale = []
        for w in weights:
            ent = self._cache.get(id(w))
            tag = (w.data_ptr(), w._version, need_t)
            if ent is not None and ent[0][:2] == tag[:2] and (ent[0][2] or not need_t):
                continue
            w2d = w.reshape(w.shape[0], -1)
            N, K = w2d.shape
            if N % 16 or K % 16:
                raise ValueError(f"Linear ({N}, {K}): both dimensions must be multiples of 16 for the packed MFMA operand")
            src = _c(w2d)
            # image buffers are reused across refreshes -- but only on the parameter's CURRENT device (.to('cuda:1') moves the
            # parameter, not the images)
            reuse = ent is not None and ent[1].numel() == N * K and ent[1].device == src.device
            wp = ent[1] if reuse else torch.empty(N * K, device=src.device, dtype=torch.float32)
            wt = None
            if need_t:
                wt = ent[2] if reuse and ent[2] is not None else torch.empty(N * K, device=src.device, dtype=torch.float32)
            stale.append((w, src, N, K, wp, wt, tag))
        if not stale:
            return
        n = len(stale)
        lib = _lib.load()
        srcs = (C.c_void_p * n)(*[e[1].data_ptr() for e in stale])
        Ns = (C.c_int32 * n)(*[e[2] for e in stale])
        Ks = (C.c_int32 * n)(*[e[3] for e in stale])
        wps = (C.c_void_p * n)(*[e[4].data_ptr() for e in stale])
        wts = (C.c_void_p * n)(*[(e[5].data_ptr() if e[5] is not None else None) for e in stale])
        with torch.cuda.device(stale[0][1].device):  # the library keeps its tables per CURRENT device
            _lib.check(lib.mdt_op_pack_many(n, srcs, Ns, Ks, wps, wts, _stream(stale[0][1])))
        for w, src, N, K, wp, wt, tag in stale:
            self._cache[id(w)] = (tag, wp, wt)
        self._keep = [e[1] for e in stale]  # converted / re-laid-out sources stay alive until the launch has run

    def get_glu(self, w: torch.Tensor) -> torch.Tensor:
        """The tile-interleaved image of a SwishGLU project weight (2H, K) (mdt_op_pack_weight_glu), cached like the others."""
        tag = (w.data_ptr(), w._version)
        ent = self._cache.get(("glu", id(w)))
        if ent is not None and ent[0] == tag:
            return ent[1]
        src = _c(w)
        img = ent[1] if ent is not None and ent[1].device == src.device and ent[1].numel() == src.numel() else torch.empty(src.numel(), device=src.device, dtype=torch.float32)
        with torch.cuda.device(src.device):
            _lib.check(_lib.load().mdt_op_pack_weight_glu(src.data_ptr(), src.shape[0], src.shape[1], img.data_ptr(), _stream(src)))
        self._cache[("glu", id(w))] = (tag, img, None)
        self._keep_glu = src
        return img

    def get(self, w2d: torch.Tensor, key_param: torch.Tensor, need_t: bool):
        tag = (key_param.data_ptr(), key_param._version, need_t)
        ent = self._cache.get(id(key_param))
        if ent is not None and ent[0][:2] == tag[:2] and (ent[0][2] or not need_t):
            return ent[1], ent[2]
        lib = _lib.load()
        N, K = w2d.shape
        if N % 16 or K % 16:
            raise ValueError(f"Linear ({N}, {K}): both dimensions must be multiples of 16 for the packed MFMA operand")
        src = _c(w2d)
        s = _stream(src)
        wp = torch.empty(N * K, device=src.device, dtype=torch.float32)
        _lib.check(lib.mdt_op_pack_weight(src.data_ptr(), N, K, wp.data_ptr(), 0, N, s))
        wt = None
        if need_t:
            wt = torch.zeros(N * K, device=src.device, dtype=torch.float32)
            _lib.check(lib.mdt_op_pack_weight_t(src.data_ptr(), N, K, K, wt.data_ptr(), 0, N, s))
        self._cache[id(key_param)] = (tag, wp, wt)
        return wp, wt


# ---- weight gradients beside the backward chain (round 6) -----------------------------------------------------------------
# dW = dY^T X (and the bias gradient) of a Linear is a leaf of the backward: nothing reads it before the optimizer.  As in the
# denoiser's own backward (csrc/mdt_train.hip, MDT_HIP_DW_STREAM) it runs on a side stream behind "dY exists" while the chain
# -- the input-gradient products, the norm / attention backward -- goes on; ONE join, queued as an end-of-backward callback of
# the autograd engine, puts the side stream back in front of whatever follows loss.backward().  Only where nothing can look at
# the gradient earlier: the parameters carry no .grad yet (AccumulateGrad then adopts the tensor without a kernel) and no tensor
# hooks, and no multi-rank process group exists (DistributedDataParallel copies a gradient into its bucket the moment it appears).
_SIDE = {}


def _dw_beside(params) -> bool:
    import os
    if os.environ.get("MDT_HIP_MAE_DW_STREAM", "1") in ("", "0"):
        return False
    try:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            return False
    except Exception:  # noqa: BLE001
        pass
    return all(p is None or (p.grad is None and not p._backward_hooks) for p in params)


def _side_stream(dev: torch.device):
    """(side stream ordered behind the caller's stream as it stands, list that keeps operands alive until the join) -- or
    (None, None) when no end-of-backward callback can be queued (not inside the autograd engine)."""
    st = _SIDE.get(dev.index)
    if st is None:
        st = _SIDE[dev.index] = {"stream": torch.cuda.Stream(dev), "keep": []}
    side = st["stream"]

    def join():  # end of this backward pass, in the caller's thread and stream; one per Linear, all but the first find nothing to do
        if st["keep"]:  # (every side launch leaves its operands here: empty = an earlier callback of this pass has joined already)
            torch.cuda.current_stream(dev).wait_stream(side)
            del st["keep"][:]  # released BEHIND the join: whoever reuses their memory is ordered behind the side stream's reads

    try:
        torch.autograd.Variable._execution_engine.queue_callback(join)
    except Exception:  # noqa: BLE001 -- not inside a backward pass of the engine (or a torch without the hook): no deferral
        return None, None
    side.wait_stream(torch.cuda.current_stream(dev))  # the operands (dY, X) exist
    return side, st["keep"]


def _linear_bwd(lib, x2, dY, N, K, wt, need_x, need_w, need_b, dx_cols=None, act_u=None, act=0, beside=False):
    """mdt_op_linear_bwd on contiguous (M, K) x2 / (M, N) dY -> (dX, dW, db); dx_cols / act_u / act: the activation below
    rides on the input-gradient product (SwishGLU: dX has 2 K columns).  beside: dW / db on the side stream (above)."""
    M = x2.shape[0]
    dW = torch.empty((N, K), device=dY.device, dtype=torch.float32) if need_w else None
    db = torch.empty((N,), device=dY.device, dtype=torch.float32) if need_b else None
    xc = K if dx_cols is None else dx_cols
    dX = torch.empty((M, xc), device=dY.device, dtype=torch.float32) if need_x else None
    scratch = torch.empty(max(1, lib.mdt_op_linear_bwd_scratch(M, N, K)), device=dY.device, dtype=torch.float32)
    side = keep = None
    if beside and (need_w or need_b):
        side, keep = _side_stream(dY.device)
    if side is not None:
        w = _lib.LinearBwdArgs(X=x2.data_ptr(), ldx=K, dY=dY.data_ptr(), ldy=N, Wt=None, dW=None if dW is None else dW.data_ptr(),
                               dbias=None if db is None else db.data_ptr(), dX=None, ldxo=xc, accumulate_dw=0, accumulate_dx=0,
                               M=M, N=N, K=K, scratch=scratch.data_ptr(), dx_act_u=None, dx_act=0)
        _lib.check(lib.mdt_op_linear_bwd(C.byref(w), side.cuda_stream))
        # torch's allocator must not hand their memory on while the side stream uses it: held until the join (not record_stream --
        # blocks that come free at event-dependent moments made the caching allocator grow by fresh hipMallocs now and then: one
        # run in three of tools/mae_bench.py took 219 instead of 26 ms per step)
        # (NOT dW / db: a second reference would make AccumulateGrad clone the gradient -- on the chain's stream, before the side
        #  stream has written it -- instead of adopting the tensor; they stay alive as the parameters' .grad)
        keep.extend(t for t in (x2, dY, scratch) if t is not None)
        if dX is None:
            return dX, dW, db
        need_w = need_b = False
        scratch = torch.empty(max(1, lib.mdt_op_linear_bwd_scratch(M, N, K)), device=dY.device, dtype=torch.float32)
    a = _lib.LinearBwdArgs(X=x2.data_ptr(), ldx=K, dY=dY.data_ptr(), ldy=N, Wt=None if wt is None else wt.data_ptr(),
                           dW=dW.data_ptr() if need_w else None, dbias=db.data_ptr() if need_b else None,
                           dX=None if dX is None else dX.data_ptr(), ldxo=xc, accumulate_dw=0, accumulate_dx=0, M=M, N=N, K=K,
                           scratch=scratch.data_ptr(), dx_act_u=None if (act_u is None or dX is None) else act_u.data_ptr(), dx_act=act)
    _lib.check(lib.mdt_op_linear_bwd(C.byref(a), _stream(dY)))
    return dX, dW, db


class HipSwiGLUMLP(torch.autograd.Function):
    """The block's MLP, y = (projected * silu(gate)) W1^T + b1 with [projected | gate] = x W0^T + b0 (voltron
    ``nn.Sequential(SwishGLU(d, h), nn.Linear(h, d))``), with the SwishGLU riding on the GEMMs around it: forward on the
    EPILOGUE of the project product (mdt_gemm_args.aux_mode 3: the weight image interleaves the two halves tile by tile, one
    launch leaves u and projected * silu(gate)), backward on the epilogue of mlp.1's input-gradient product
    (mdt_linear_bwd_args.dx_act = SWIGLU: d_u straight from dY W1 and u).  Two elementwise passes over (rows, 2h) and
    (rows, h) tensors less each way than Linear -> SwishGLU -> Linear."""

    @staticmethod
    def forward(ctx, x, w0, b0, w1, b1, packs: PackedWeights):
        lib = _lib.load()
        H2, K = w0.shape
        H, N1 = H2 // 2, w1.shape[0]
        x2 = _c(x).reshape(-1, K)
        M = x2.shape[0]
        s = _stream(x2)
        need_t = torch.is_grad_enabled() and (x.requires_grad or w0.requires_grad)
        wg = packs.get_glu(w0)
        w1p, _ = packs.get(w1, w1, need_t)
        u = torch.empty((M, H2), device=x2.device, dtype=torch.float32)
        h = torch.empty((M, H), device=x2.device, dtype=torch.float32)
        a = _lib.GemmArgs()
        a.A, a.lda, a.Wp, a.out, a.ldo, a.M, a.N, a.K = x2.data_ptr(), K, wg.data_ptr(), h.data_ptr(), H, M, H2, K
        a.bias = None if b0 is None else _c(b0).data_ptr()
        a.shift_off, a.scale_off, a.gate_off, a.rows_per_sample, a.gin, a.gout, a.goff = -1, -1, -1, 1, 1, 1, 0
        a.aux, a.aux_mode = u.data_ptr(), 3
        _lib.check(lib.mdt_op_gemm(C.byref(a), s))
        y = torch.empty((M, N1), device=x2.device, dtype=torch.float32)
        g = _lib.GemmArgs()
        g.A, g.lda, g.Wp, g.out, g.ldo, g.M, g.N, g.K = h.data_ptr(), H, w1p.data_ptr(), y.data_ptr(), N1, M, N1, H
        g.bias = None if b1 is None else _c(b1).data_ptr()
        g.shift_off, g.scale_off, g.gate_off, g.rows_per_sample, g.gin, g.gout, g.goff = -1, -1, -1, 1, 1, 1, 0
        _lib.check(lib.mdt_op_gemm(C.byref(g), s))
        ctx.save_for_backward(x2, u, h, w0, b0, w1, b1)
        ctx.packs, ctx.xshape = packs, x.shape
        return y.reshape(*x.shape[:-1], N1)

    @staticmethod
    def backward(ctx, gy):
        lib = _lib.load()
        x2, u, h, w0, b0, w1, b1 = ctx.saved_tensors
        H2, K = w0.shape
        H, N1 = H2 // 2, w1.shape[0]
        M = x2.shape[0]
        dY = _c(gy).reshape(M, N1)
        nx, nw0, nb0, nw1, nb1 = (ctx.needs_input_grad[i] for i in range(5))
        need_u = nx or nw0 or (b0 is not None and nb0)
        w1t = ctx.packs.get(w1, w1, True)[1] if need_u else None
        # mlp.1: dW1 = dY^T h, db1, and d_u = swiglu'(u) (dY W1) in the same launch sequence
        beside = _dw_beside((w0, b0, w1, b1))
        du, dW1, db1 = _linear_bwd(lib, h, dY, N1, H, w1t, need_u, nw1, b1 is not None and nb1, dx_cols=H2, act_u=u, act=_lib.ACT["swiglu"],
                                   beside=beside)
        dX = dW0 = db0 = None
        if need_u:
            w0t = ctx.packs.get(w0, w0, True)[1] if nx else None
            dX, dW0, db0 = _linear_bwd(lib, x2, du, H2, K, w0t, nx, nw0, b0 is not None and nb0, beside=beside)
        return (None if dX is None else dX.reshape(ctx.xshape), dW0, db0, dW1, db1, None)


class HipLinear(torch.autograd.Function):
    """y = x W^T + b on the fused MFMA GEMM; x (..., K) -> (..., N).  ``w2d`` is the (N, K) view of ``weight`` (a Conv2d
    patch embedding is a Linear over flattened patches)."""

    @staticmethod
    def forward(ctx, x, weight, bias, packs: PackedWeights):
        lib = _lib.load()
        w2d = weight.reshape(weight.shape[0], -1)
        N, K = w2d.shape
        x2 = _c(x).reshape(-1, K)
        M = x2.shape[0]
        need_t = x.requires_grad
        wp, wt = packs.get(w2d, weight, need_t)
        out = torch.empty((M, N), device=x2.device, dtype=torch.float32)
        a = _lib.GemmArgs()
        a.A, a.lda, a.Wp, a.out, a.ldo, a.M, a.N, a.K = x2.data_ptr(), K, wp.data_ptr(), out.data_ptr(), N, M, N, K
        a.bias = None if bias is None else _c(bias).data_ptr()
        a.shift_off, a.scale_off, a.gate_off, a.rows_per_sample, a.gin, a.gout, a.goff = -1, -1, -1, 1, 1, 1, 0
        _lib.check(lib.mdt_op_gemm(C.byref(a), _stream(x2)))
        ctx.save_for_backward(x2, weight, bias)
        ctx.packs, ctx.xshape = packs, x.shape
        return out.reshape(*x.shape[:-1], N)

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        x2, weight, bias = ctx.saved_tensors
        w2d = weight.reshape(weight.shape[0], -1)
        N, K = w2d.shape
        M = x2.shape[0]
        dY = _c(g).reshape(M, N)
        need_x, need_w, need_b = ctx.needs_input_grad[0], ctx.needs_input_grad[1], bias is not None and ctx.needs_input_grad[2]
        wt = ctx.packs.get(w2d, weight, True)[1] if need_x else None
        dX, dW, db = _linear_bwd(lib, x2, dY, N, K, wt, need_x, need_w, need_b, beside=_dw_beside((weight, bias)))
        return (None if dX is None else dX.reshape(ctx.xshape), None if dW is None else dW.reshape(weight.shape), db, None)


class HipRMSNorm(torch.autograd.Function):
    """voltron RMSNorm: x / max(||x|| D^-1/2, eps) * g over the last dimension."""

    @staticmethod
    def forward(ctx, x, g):
        lib = _lib.load()
        D = x.shape[-1]
        x2 = _c(x).reshape(-1, D)
        gg = _c(g)
        out = torch.empty_like(x2)
        _lib.check(lib.mdt_op_rms_fwd(x2.data_ptr(), gg.data_ptr(), out.data_ptr(), x2.shape[0], D, RMS_EPS, _stream(x2)))
        ctx.save_for_backward(x2, gg)
        ctx.xshape = x.shape
        return out.reshape(x.shape)

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        x2, g = ctx.saved_tensors
        M, D = x2.shape
        d = _c(dy).reshape(M, D)
        dx = torch.empty_like(x2)
        dg = torch.empty_like(g)
        scratch = torch.empty(lib.mdt_op_rms_bwd_scratch(M, D), device=d.device, dtype=torch.float32)
        _lib.check(lib.mdt_op_rms_bwd(x2.data_ptr(), g.data_ptr(), d.data_ptr(), dx.data_ptr(), 0, dg.data_ptr(), 0, M, D, RMS_EPS,
                                      scratch.data_ptr(), _stream(d)))
        return dx.reshape(ctx.xshape), dg


class HipRMSNormBranch(torch.autograd.Function):
    """The norm at the head of a residual branch: returns (x, RMSNorm(x)) -- x for the residual sum, the normed rows for
    the branch -- so that ONE backward sees both gradients that reach x and the norm's backward kernel adds them on its way
    out (`mdt_op_rms_bwd_res`).  With `HipRMSNorm` autograd forms that sum itself: one elementwise launch over the
    (B, T, D) gradient per branch, twelve per step of the six-block decoder."""

    @staticmethod
    def forward(ctx, x, g):
        lib = _lib.load()
        D = x.shape[-1]
        x2 = _c(x).reshape(-1, D)
        gg = _c(g)
        out = torch.empty_like(x2)
        _lib.check(lib.mdt_op_rms_fwd(x2.data_ptr(), gg.data_ptr(), out.data_ptr(), x2.shape[0], D, RMS_EPS, _stream(x2)))
        ctx.save_for_backward(x2, gg)
        ctx.xshape = x.shape
        ctx.set_materialize_grads(False)  # an unused output arrives as None, not as a dense zero tensor: the branches below are live
        return x.view_as(x), out.reshape(x.shape)

    @staticmethod
    def backward(ctx, d_res, dy):
        lib = _lib.load()
        x2, g = ctx.saved_tensors
        M, D = x2.shape
        if dy is None:
            return d_res, None
        d = _c(dy).reshape(M, D)
        dx = torch.empty_like(x2)
        dg = torch.empty_like(g)
        scratch = torch.empty(lib.mdt_op_rms_bwd_scratch(M, D), device=d.device, dtype=torch.float32)
        if d_res is None:
            _lib.check(lib.mdt_op_rms_bwd(x2.data_ptr(), g.data_ptr(), d.data_ptr(), dx.data_ptr(), 0, dg.data_ptr(), 0, M, D, RMS_EPS,
                                          scratch.data_ptr(), _stream(d)))
        else:
            r = _c(d_res).reshape(M, D)
            _lib.check(lib.mdt_op_rms_bwd_res(x2.data_ptr(), g.data_ptr(), d.data_ptr(), r.data_ptr(), dx.data_ptr(), dg.data_ptr(), 0,
                                              M, D, RMS_EPS, scratch.data_ptr(), _stream(d)))
        return dx.reshape(ctx.xshape), dg


class HipSwishGLU(torch.autograd.Function):
    """u = [projected | gate] (..., 2H) -> projected * silu(gate)."""

    @staticmethod
    def forward(ctx, u):
        lib = _lib.load()
        H2 = u.shape[-1]
        u2 = _c(u).reshape(-1, H2)
        out = torch.empty((u2.shape[0], H2 // 2), device=u2.device, dtype=torch.float32)
        _lib.check(lib.mdt_op_swiglu_fwd(u2.data_ptr(), out.data_ptr(), u2.shape[0], H2 // 2, _stream(u2)))
        ctx.save_for_backward(u2)
        ctx.ushape = u.shape
        return out.reshape(*u.shape[:-1], H2 // 2)

    @staticmethod
    def backward(ctx, d_out):
        lib = _lib.load()
        (u2,) = ctx.saved_tensors
        M, H2 = u2.shape
        d = _c(d_out).reshape(M, H2 // 2)
        du = torch.empty_like(u2)
        _lib.check(lib.mdt_op_swiglu_bwd(u2.data_ptr(), d.data_ptr(), du.data_ptr(), M, H2 // 2, _stream(d)))
        return du.reshape(ctx.ushape)


class HipScaleResidual(torch.autograd.Function):
    """x + gamma * z (voltron LayerScale on a residual branch) in one pass; backward: dx = g, dz = gamma * g,
    dgamma = sum_rows g * z in one more."""

    @staticmethod
    def forward(ctx, x, z, gamma):
        lib = _lib.load()
        D = x.shape[-1]
        x2, z2, gm = _c(x).reshape(-1, D), _c(z).reshape(-1, D), _c(gamma)
        out = torch.empty_like(x2)
        _lib.check(lib.mdt_op_scale_residual_fwd(x2.data_ptr(), z2.data_ptr(), gm.data_ptr(), out.data_ptr(), x2.shape[0], D,
                                                 _stream(x2)))
        ctx.save_for_backward(z2, gm)
        ctx.xshape = x.shape
        return out.reshape(x.shape)

    @staticmethod
    def _bwd(g2, z2, gm):
        lib = _lib.load()
        M, D = z2.shape
        dz = torch.empty_like(z2)
        dgamma = torch.empty_like(gm)
        scratch = torch.empty(lib.mdt_op_scale_residual_bwd_scratch(M, D), device=g2.device, dtype=torch.float32)
        _lib.check(lib.mdt_op_scale_residual_bwd(g2.data_ptr(), z2.data_ptr(), gm.data_ptr(), dz.data_ptr(), dgamma.data_ptr(), M, D,
                                                 scratch.data_ptr(), _stream(g2)))
        return g2, dz, dgamma

    @staticmethod
    def backward(ctx, g):
        z2, gm = ctx.saved_tensors
        _, dz, dgamma = HipScaleResidual._bwd(_c(g).reshape(z2.shape), z2, gm)
        return g, dz.reshape(ctx.xshape), dgamma


class HipScaleResidualNorm(torch.autograd.Function):
    """x' = x + gamma * z and h = RMSNorm(x'; g_norm), the residual sum of one branch and the norm at the head of the next,
    in one pass each way (`mdt_op_scale_residual_rms_fwd / _bwd`).  Returns (x', h); the backward receives the gradient that
    reaches x' on the residual path (None when nothing else uses x': the decoder's last norm) and h's."""

    @staticmethod
    def forward(ctx, x, z, gamma, g_norm):
        lib = _lib.load()
        D = x.shape[-1]
        x2, z2, gm, gn = _c(x).reshape(-1, D), _c(z).reshape(-1, D), _c(gamma), _c(g_norm)
        xn, h = torch.empty_like(x2), torch.empty_like(x2)
        _lib.check(lib.mdt_op_scale_residual_rms_fwd(x2.data_ptr(), z2.data_ptr(), gm.data_ptr(), gn.data_ptr(), xn.data_ptr(),
                                                     h.data_ptr(), x2.shape[0], D, RMS_EPS, _stream(x2)))
        ctx.save_for_backward(xn, z2, gm, gn)
        ctx.xshape = x.shape
        ctx.set_materialize_grads(False)
        return xn.reshape(x.shape), h.reshape(x.shape)

    @staticmethod
    def backward(ctx, d_res, d_h):
        lib = _lib.load()
        xn, z2, gm, gn = ctx.saved_tensors
        M, D = xn.shape
        if d_h is None:   # only the residual sum was used
            if d_res is None:
                return None, None, None, None
            dx, dz, dgamma = HipScaleResidual._bwd(_c(d_res).reshape(M, D), z2, gm)
            return dx.reshape(ctx.xshape), dz.reshape(ctx.xshape), dgamma, None
        d = _c(d_h).reshape(M, D)
        r = None if d_res is None else _c(d_res).reshape(M, D)
        dx, dz = torch.empty_like(xn), torch.empty_like(xn)
        dgamma, dgn = torch.empty_like(gm), torch.empty_like(gn)
        scratch = torch.empty(lib.mdt_op_scale_residual_rms_bwd_scratch(M, D), device=d.device, dtype=torch.float32)
        _lib.check(lib.mdt_op_scale_residual_rms_bwd(xn.data_ptr(), gn.data_ptr(), d.data_ptr(), None if r is None else r.data_ptr(),
                                                     z2.data_ptr(), gm.data_ptr(), dx.data_ptr(), dz.data_ptr(), dgamma.data_ptr(),
                                                     dgn.data_ptr(), M, D, RMS_EPS, scratch.data_ptr(), _stream(d)))
        return dx.reshape(ctx.xshape), dz.reshape(ctx.xshape), dgamma, dgn


class HipPatchMSE(torch.autograd.Function):
    """compute_loss (reference :228-262): masked per-patch MSE of both frames straight from the images (no patchified copy),
    one pass forward and one backward (mdt_op_patch_mse_fwd / _bwd) instead of ~10 elementwise passes over 300 MB tensors."""

    @staticmethod
    def forward(ctx, rec, imgs, mask, patch: int):
        lib = _lib.load()
        B, X, n, E = rec.shape
        Cn, R = imgs.shape[2], imgs.shape[3]
        r, im, mk = _c(rec), _c(imgs), _c(mask)
        partial = torch.empty(B * X * n, device=r.device, dtype=torch.float32)
        out = torch.empty(2, device=r.device, dtype=torch.float32)  # loss, sum(mask)
        _lib.check(lib.mdt_op_patch_mse_fwd(r.data_ptr(), im.data_ptr(), mk.data_ptr(), partial.data_ptr(), out.data_ptr(),
                                            out.data_ptr() + 4, B, X, Cn, R, patch, _stream(r)))
        ctx.save_for_backward(r, im, mk, out)
        ctx.cfg = (B, X, Cn, R, patch)
        return out[0]

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        r, im, mk, out = ctx.saved_tensors
        B, X, Cn, R, patch = ctx.cfg
        gg = _c(g.reshape(1).float())
        d = torch.empty_like(r)
        _lib.check(lib.mdt_op_patch_mse_bwd(r.data_ptr(), im.data_ptr(), mk.data_ptr(), out.data_ptr() + 4, gg.data_ptr(), d.data_ptr(),
                                            B, X, Cn, R, patch, _stream(r)))
        return d, None, None, None


class HipSelfAttention(torch.autograd.Function):
    """qkv (B, T, 3 D) = q | k | v -> softmax(q k^T * scale) v (B, T, D), H heads, unmasked, T <= 128."""

    @staticmethod
    def forward(ctx, qkv, n_heads: int, scale: float):
        lib = _lib.load()
        B, T, D3 = qkv.shape
        D = D3 // 3
        q = _c(qkv)
        out = torch.empty((B, T, D), device=q.device, dtype=torch.float32)
        _lib.check(lib.mdt_op_attn_mid_fwd(q.data_ptr(), D3, out.data_ptr(), D, B, n_heads, D // n_heads, T, float(scale), _stream(q)))
        ctx.save_for_backward(q, out)  # the output is kept anyway (input of the projection that follows)
        ctx.cfg = (n_heads, float(scale))
        return out

    @staticmethod
    def backward(ctx, d_out):
        lib = _lib.load()
        q, out = ctx.saved_tensors
        n_heads, scale = ctx.cfg
        B, T, D3 = q.shape
        D = D3 // 3
        d = _c(d_out)
        dq = torch.empty_like(q)
        _lib.check(lib.mdt_op_attn_mid_bwd(q.data_ptr(), D3, out.data_ptr(), D, d.data_ptr(), D, dq.data_ptr(), D3, B, n_heads,
                                           D // n_heads, T, scale, _stream(d)))
        return dq, None, None
