"""Noise schedules and samplers (reference: mdt/models/edm_diffusion/gc_sampling.py).

``sample_ddim`` -- the sampler MDT ships with (conf/model/mdtv_agent.yaml:14) -- runs as ONE call into
libmdt_hip.so when ``model`` is this package's GCDenoiser and no Python hooks are requested: encoder and
cross-attention K/V once, adaLN vectors of all steps once, the DDIM update fused into the action-head kernel.
The other samplers keep the reference signatures and are host loops over ``model(state, x, goal, sigma)``
(the HIP denoiser step) with the sigma-independent encoder hoisted out of the loop.

Signatures follow the reference: ``sample_*(model, state, action, goal, sigmas, scaler=None, extra_args=None,
callback=None, disable=None, ...)``.
"""
from __future__ import annotations

import math
import os
from contextlib import nullcontext

import numpy as np
import torch

from . import utils
from .score_wrappers import GCDenoiser


# ------------------------------------------------------------------------------------------------
# schedules (reference gc_sampling.py:22-88); all return n+1 values ending in 0
# ------------------------------------------------------------------------------------------------
def append_zero(action):
    return torch.cat([action, action.new_zeros([1])])


def get_sigmas_karras(n, sigma_min, sigma_max, rho=7., device='cpu'):
    """Karras et al. (2022) schedule: linear ramp in sigma^(1/rho)."""
    ramp = torch.linspace(0, 1, n)
    lo, hi = sigma_min ** (1 / rho), sigma_max ** (1 / rho)
    return append_zero((hi + ramp * (lo - hi)) ** rho).to(device)


def get_sigmas_exponential(n, sigma_min, sigma_max, device='cpu'):
    """Geometric schedule: linear ramp in log sigma (the MDT default, mdtv_agent.yaml:18)."""
    return append_zero(torch.linspace(math.log(sigma_max), math.log(sigma_min), n, device=device).exp())


def get_sigmas_linear(n, sigma_min, sigma_max, device='cpu'):
    return append_zero(torch.linspace(sigma_max, sigma_min, n, device=device))


def cosine_beta_schedule(n, s=0.008, device='cpu'):
    """Cosine beta schedule, flipped and clipped (reference gc_sampling.py:47-58)."""
    steps = n + 1
    x = np.linspace(0, steps, steps)
    acp = np.cos(((x / steps) + s) / (1 + s) * np.pi * 0.5) ** 2
    acp = acp / acp[0]
    betas = np.clip(1 - (acp[1:] / acp[:-1]), a_min=0, a_max=0.999)
    return append_zero(torch.tensor(np.flip(betas).copy(), device=device, dtype=torch.float32))


def get_sigmas_ve(n, sigma_min=0.02, sigma_max=100, device='cpu'):
    """Variance-exploding schedule incl. the reference's linspace(0, n+1, n) parametrisation (:61-69)."""
    t = torch.linspace(0, n + 1, n, device=device)
    t = (sigma_max ** 2) * ((sigma_min ** 2 / sigma_max ** 2) ** (t / (n - 1)))
    return append_zero(torch.sqrt(t))


def get_iddpm_sigmas(n, sigma_min=0.02, sigma_max=100, M=1000, j_0=0, C_1=0.001, C_2=0.008, device='cpu'):
    """Improved-DDPM schedule (reference gc_sampling.py:71-81): u_{j-1} from the cosine alpha-bar recursion, the
    levels inside [sigma_min, sigma_max] sub-sampled at n evenly spaced (rounded) indices.  As in the reference the
    alpha-bar ratios are float32 (Python float x integer tensor promotes to the default dtype) and only the
    recursion itself runs in float64 -- the ratios near j = M sit within rounding of 1, so this matters at 1e-4."""
    j = torch.arange(0, M + 1, device=device)
    abar = (0.5 * np.pi * j / M / (C_2 + 1)).sin() ** 2              # float32
    ratio = (abar[:-1] / abar[1:]).clip(min=C_1)                     # ratio[j-1] = abar(j-1) / abar(j)
    u = torch.zeros(M + 1, dtype=torch.float64, device=device)
    for jj in range(M, j_0, -1):
        u[jj - 1] = ((u[jj] ** 2 + 1) / ratio[jj - 1] - 1).sqrt()
    kept = u[torch.logical_and(u >= sigma_min, u <= sigma_max)]
    idx = ((len(kept) - 1) / (n - 1) * torch.arange(n, dtype=torch.float64, device=device)).round().to(torch.int64)
    return append_zero(kept[idx]).to(torch.float32)


def get_sigmas_vp(n, beta_d=19.9, beta_min=0.1, eps_s=1e-3, device='cpu'):
    t = torch.linspace(1, eps_s, n, device=device)
    return append_zero(torch.sqrt(torch.exp(beta_d * t ** 2 / 2 + beta_min * t) - 1))


def to_d(action, sigma, denoised):
    """Karras ODE derivative dx/dsigma = (x - D(x; sigma)) / sigma."""
    if torch.is_tensor(sigma) and sigma.ndim > 0:
        return (action - denoised) / utils.append_dims(sigma.to(action.device), action.ndim)
    return (action - denoised) / _f(sigma)


def default_noise_sampler(x):
    return lambda sigma, sigma_next: torch.randn_like(x)


def get_ancestral_step(sigma_from, sigma_to, eta=1.):
    """Split a step into a deterministic part down to sigma_down and fresh noise sigma_up."""
    if not eta:
        return sigma_to, 0.
    sigma_up = min(sigma_to, eta * (sigma_to ** 2 * (sigma_from ** 2 - sigma_to ** 2) / sigma_from ** 2) ** 0.5)
    sigma_down = (sigma_to ** 2 - sigma_up ** 2) ** 0.5
    return sigma_down, sigma_up


# ------------------------------------------------------------------------------------------------
# samplers
# ------------------------------------------------------------------------------------------------
def _hoist(model, state, goal):
    """Encoder-once context for this package's denoiser; a no-op for any other callable."""
    if isinstance(model, GCDenoiser):
        return model.cached_context(state, goal)
    return nullcontext()


def _t(sigma):
    return sigma.log().neg()


def _sigma(t):
    return t.neg().exp()


def _host(sigmas):
    """Schedules are tiny: keep their scalar arithmetic on the host (fp32 0-dim CPU tensors, exactly the reference's
    expressions) so that no step of a sampler loop waits on the device for a scalar."""
    return sigmas.detach().to("cpu", torch.float32) if torch.is_tensor(sigmas) else torch.tensor(sigmas, dtype=torch.float32)


def _f(x):
    """0-dim fp32 host tensor (or number) -> Python float (exact), usable against tensors on any device."""
    return x.item() if torch.is_tensor(x) else float(x)


def _sig_in(sigma, action, model=None):
    """sigma * s_in of the reference.  This package's denoiser takes the single shared value (one adaLN row for the
    whole batch, broadcast inside the kernels); any other model gets the reference's (B,) vector."""
    n = 1 if isinstance(model, GCDenoiser) else action.shape[0]
    return torch.full((n,), _f(sigma), device=action.device, dtype=action.dtype)


# MDT_HIP_GRAPH: "1" = every fused DDIM call replays a HIP graph, "0" = never, unset = "auto": rollout-sized batches (B <= 8)
# from the third call with the same shapes on -- a rollout repeats one call for hundreds of steps (mdtv_agent.py:721-760) and
# its ~250 launches take the host longer to submit than the GPU to run (B = 1, host-synchronised: 1.56 ms eager, 1.43 ms
# replayed; tools/graph_probe.py), a one-off call is not worth a capture
_GRAPH_MODE = os.environ.get("MDT_HIP_GRAPH", "auto")
_GRAPH_SAMPLER = _GRAPH_MODE not in ("", "0", "auto")
_GRAPH_AUTO_MAX_BATCH, _GRAPH_AUTO_AFTER = 8, 2


def _graph_key(state, action, goal, sigmas):
    """Hashable for any `state` content: tensors by shape, everything else by type and repr (a list or dict value is legal)."""
    return (tuple(action.shape), tuple(goal.shape), len(sigmas),
            tuple(sorted((str(k), tuple(v.shape) if torch.is_tensor(v) else (type(v).__name__, repr(v)))
                         for k, v in state.items())))


def _graph_wanted(model, state, action, goal, sigmas) -> bool:
    if action.device.type != "cuda" or model.inner_model.training or torch.cuda.is_current_stream_capturing():
        return False
    if _GRAPH_SAMPLER:
        return True
    if _GRAPH_MODE != "auto" or action.shape[0] > _GRAPH_AUTO_MAX_BATCH:
        return False
    key = _graph_key(state, action, goal, sigmas)
    if key in model.__dict__.get("_graph_failed", ()):
        return False  # a capture of this call failed once: it stays eager
    seen = model.__dict__.setdefault("_graph_seen", {})
    seen[key] = seen.get(key, 0) + 1
    if len(seen) > 16:
        seen.clear()
    return seen.get(key, 0) > _GRAPH_AUTO_AFTER


def _graphed(model, state, action, goal, sigmas):
    """One GraphedDDIM per (shapes, modality, state keys) of a model, kept on the model (at most four)."""
    from .graphed import GraphedDDIM
    cache = model.__dict__.setdefault("_graphed_samplers", [])
    for gsamp in cache:
        if gsamp.matches(state, action, goal, sigmas):
            return gsamp(state, action, goal, sigmas)
    gsamp = GraphedDDIM(model, state, action, goal, sigmas if torch.is_tensor(sigmas) else torch.as_tensor(sigmas))
    cache.append(gsamp)
    del cache[:-4]
    return gsamp(state, action, goal, sigmas)


@torch.no_grad()
def sample_ddim(model, state, action, goal, sigmas, scaler=None, extra_args=None, callback=None, disable=None,
                eta=1.):
    """DPM-Solver-1 / DDIM (reference gc_sampling.py:922-951):
    x <- (sigma_{i+1}/sigma_i) x - expm1(-(t_{i+1} - t_i)) D(x; sigma_i),  t = -ln sigma.
    `scaler` is accepted and never read, exactly as in the reference (its DDIM has no `clip_output` call), so a harness run
    with `use_scaler` (mdtv_agent.py:606-614) keeps the fused native loop; only `callback` / `extra_args` need the step loop."""
    extra_args = {} if extra_args is None else extra_args
    if isinstance(model, GCDenoiser) and callback is None and not extra_args:
        if _graph_wanted(model, state, action, goal, sigmas):
            if _GRAPH_SAMPLER:
                return _graphed(model, state, action, goal, sigmas)  # the same launches, replayed as a HIP graph
            try:  # auto mode: a capture that fails (another thread allocating / synchronising while torch's global capture
                #   mode is on, ...) must not break a rollout the eager path would have served
                return _graphed(model, state, action, goal, sigmas)
            except Exception as exc:  # noqa: BLE001 -- whatever the capture raised, the eager launches below still work
                model.__dict__.setdefault("_graph_failed", set()).add(_graph_key(state, action, goal, sigmas))
                model.__dict__.pop("_graphed_samplers", None)
                import warnings
                warnings.warn(f"mdt_policy_amd: HIP-graph capture of sample_ddim failed ({exc!r}); this call shape stays eager")
                torch.cuda.synchronize()
        return model.sample_ddim(state, action, goal, sigmas)  # fused native loop
    sig = _host(sigmas)
    with _hoist(model, state, goal):
        for i in range(len(sig) - 1):
            denoised = model(state, action, goal, _sig_in(sig[i], action, model), **extra_args)
            if callback is not None:
                callback({'action': action, 'i': i, 'sigma': sig[i], 'sigma_hat': sig[i], 'denoised': denoised})
            t, t_next = _t(sig[i]), _t(sig[i + 1])
            h = t_next - t
            action = _f(_sigma(t_next) / _sigma(t)) * action - _f((-h).expm1()) * denoised
    return action


@torch.no_grad()
def sample_euler(model, state, action, goal, sigmas, scaler=None, extra_args=None, callback=None, disable=None,
                 s_churn=0., s_tmin=0., s_tmax=float('inf'), s_noise=1.):
    """Karras Algorithm 2 without the 2nd-order correction (reference gc_sampling.py:164-209)."""
    extra_args = {} if extra_args is None else extra_args
    sig = _host(sigmas)
    n = len(sig) - 1
    with _hoist(model, state, goal):
        for i in range(n):
            gamma = min(s_churn / n, 2 ** 0.5 - 1) if s_tmin <= sig[i] <= s_tmax else 0.
            eps = torch.randn_like(action) * s_noise  # drawn every step, like the reference (same generator stream)
            sigma_hat = sig[i] * (gamma + 1)
            if gamma > 0:
                action = action + eps * _f((sigma_hat ** 2 - sig[i] ** 2) ** 0.5)
            denoised = model(state, action, goal, _sig_in(sigma_hat, action, model), **extra_args)
            d = to_d(action, sigma_hat, denoised)
            if callback is not None:
                callback({'x': action, 'i': i, 'sigma': sig[i], 'sigma_hat': sigma_hat, 'denoised': denoised})
            action = action + d * _f(sig[i + 1] - sigma_hat)
            if scaler is not None:
                action = scaler.clip_output(action)
    return action


@torch.no_grad()
def sample_euler_ancestral(model, state, action, goal, sigmas, scaler=None, extra_args=None, callback=None,
                           disable=None, eta=1.):
    """Euler steps to sigma_down plus fresh noise sigma_up (reference gc_sampling.py:213-252)."""
    extra_args = {} if extra_args is None else extra_args
    sig = _host(sigmas)
    with _hoist(model, state, goal):
        for i in range(len(sig) - 1):
            denoised = model(state, action, goal, _sig_in(sig[i], action, model), **extra_args)
            sigma_down, sigma_up = get_ancestral_step(sig[i], sig[i + 1], eta=eta)
            if callback is not None:
                callback({'x': action, 'i': i, 'sigma': sig[i], 'sigma_hat': sig[i], 'denoised': denoised})
            d = to_d(action, sig[i], denoised)
            action = action + d * _f(sigma_down - sig[i])
            if sigma_down > 0:
                action = action + torch.randn_like(action) * _f(sigma_up)
            if scaler is not None:
                action = scaler.clip_output(action)
    return action


@torch.no_grad()
def sample_heun(model, state, action, goal, sigmas, scaler=None, extra_args=None, callback=None, disable=None,
                s_churn=0., s_tmin=0., s_tmax=float('inf'), s_noise=1.):
    """Karras Algorithm 2 with Heun's trapezoidal correction; plain Euler on the final step to sigma = 0
    (reference gc_sampling.py:256-312)."""
    extra_args = {} if extra_args is None else extra_args
    sig = _host(sigmas)
    n = len(sig) - 1
    with _hoist(model, state, goal):
        for i in range(n):
            gamma = min(s_churn / n, 2 ** 0.5 - 1) if s_tmin <= sig[i] <= s_tmax else 0.
            eps = torch.randn_like(action) * s_noise  # drawn every step, like the reference (same generator stream)
            sigma_hat = sig[i] * (gamma + 1)
            if gamma > 0:
                action = action + eps * _f((sigma_hat ** 2 - sig[i] ** 2) ** 0.5)
            denoised = model(state, action, goal, _sig_in(sigma_hat, action, model), **extra_args)
            d = to_d(action, sigma_hat, denoised)
            if callback is not None:
                callback({'x': action, 'i': i, 'sigma': sig[i], 'sigma_hat': sigma_hat, 'denoised': denoised})
            dt = _f(sig[i + 1] - sigma_hat)
            if sig[i + 1] == 0:
                action = action + d * dt
            else:
                action_2 = action + d * dt
                denoised_2 = model(state, action_2, goal, _sig_in(sig[i + 1], action, model), **extra_args)
                d_2 = to_d(action_2, sig[i + 1], denoised_2)
                action = action + (d + d_2) / 2 * dt
            if scaler is not None:
                action = scaler.clip_output(action)
    return action


@torch.no_grad()
def sample_dpmpp_2m(model, state, action, goal, sigmas, scaler=None, extra_args=None, callback=None, disable=None):
    """DPM-Solver++(2M) multistep (reference gc_sampling.py:699-734)."""
    extra_args = {} if extra_args is None else extra_args
    sig = _host(sigmas)
    old_denoised = None
    with _hoist(model, state, goal):
        for i in range(len(sig) - 1):
            denoised = model(state, action, goal, _sig_in(sig[i], action, model), **extra_args)
            if callback is not None:
                callback({'action': action, 'i': i, 'sigma': sig[i], 'sigma_hat': sig[i], 'denoised': denoised})
            t, t_next = _t(sig[i]), _t(sig[i + 1])
            h = t_next - t
            if old_denoised is None or sig[i + 1] == 0:
                denoised_d = denoised
            else:
                r = (t - _t(sig[i - 1])) / h
                denoised_d = _f(1 + 1 / (2 * r)) * denoised - _f(1 / (2 * r)) * old_denoised
            action = _f(_sigma(t_next) / _sigma(t)) * action - _f((-h).expm1()) * denoised_d
            old_denoised = denoised
    return action


@torch.no_grad()
def sample_dpmpp_2s(model, state, action, goal, sigmas, scaler=None, extra_args=None, callback=None, disable=None,
                    eta=1.):
    """DPM-Solver++(2S) single-step second order (reference gc_sampling.py:955-994)."""
    extra_args = {} if extra_args is None else extra_args
    sig = _host(sigmas)
    with _hoist(model, state, goal):
        for i in range(len(sig) - 1):
            denoised = model(state, action, goal, _sig_in(sig[i], action, model), **extra_args)
            if callback is not None:
                callback({'action': action, 'i': i, 'sigma': sig[i], 'sigma_hat': sig[i], 'denoised': denoised})
            if sig[i + 1] == 0:
                action = action + to_d(action, sig[i], denoised) * _f(sig[i + 1] - sig[i])
            else:
                t, t_next = _t(sig[i]), _t(sig[i + 1])
                h = t_next - t
                s = t + 0.5 * h
                x_2 = _f(_sigma(s) / _sigma(t)) * action - _f((-h * 0.5).expm1()) * denoised
                denoised_2 = model(state, x_2, goal, _sig_in(_sigma(s), action, model), **extra_args)
                action = _f(_sigma(t_next) / _sigma(t)) * action - _f((-h).expm1()) * denoised_2
            if scaler is not None:
                action = scaler.clip_output(action)
    return action


@torch.no_grad()
def sample_dpm_2(model, state, action, goal, sigmas, scaler=None, extra_args=None, callback=None, disable=None,
                 s_churn=0., s_tmin=0., s_tmax=float('inf'), s_noise=1.):
    """DPM-Solver-2 flavoured midpoint steps (reference gc_sampling.py:315-371): derivative at sigma_hat, a second
    evaluation at the log-midpoint sigma, full step with the midpoint derivative; Euler on the last step."""
    extra_args = {} if extra_args is None else extra_args
    sig = _host(sigmas)
    n = len(sig) - 1
    with _hoist(model, state, goal):
        for i in range(n):
            gamma = min(s_churn / n, 2 ** 0.5 - 1) if s_tmin <= sig[i] <= s_tmax else 0.
            eps = torch.randn_like(action) * s_noise  # drawn every step, like the reference (same generator stream)
            sigma_hat = sig[i] * (gamma + 1)
            if gamma > 0:
                action = action + eps * _f((sigma_hat ** 2 - sig[i] ** 2) ** 0.5)
            denoised = model(state, action, goal, _sig_in(sigma_hat, action, model), **extra_args)
            d = to_d(action, sigma_hat, denoised)
            if callback is not None:
                callback({'action': action, 'i': i, 'sigma': sig[i], 'sigma_hat': sigma_hat, 'denoised': denoised})
            if sig[i + 1] == 0:
                action = action + d * _f(sig[i + 1] - sigma_hat)
            else:
                sigma_mid = sigma_hat.log().lerp(sig[i + 1].log(), 0.5).exp()
                action_2 = action + d * _f(sigma_mid - sigma_hat)
                denoised_2 = model(state, action_2, goal, _sig_in(sigma_mid, action, model), **extra_args)
                d_2 = to_d(action_2, sigma_mid, denoised_2)
                action = action + d_2 * _f(sig[i + 1] - sigma_hat)
            if scaler is not None:
                action = scaler.clip_output(action)
    return action


@torch.no_grad()
def sample_dpm_2_ancestral(model, state, action, goal, sigmas, scaler=None, extra_args=None, callback=None,
                           disable=None, eta=1.):
    """Ancestral sampling with DPM-Solver-2 midpoint steps (reference gc_sampling.py:374-407)."""
    extra_args = {} if extra_args is None else extra_args
    sig = _host(sigmas)
    with _hoist(model, state, goal):
        for i in range(len(sig) - 1):
            denoised = model(state, action, goal, _sig_in(sig[i], action, model), **extra_args)
            sigma_down, sigma_up = get_ancestral_step(sig[i], sig[i + 1], eta=eta)
            if callback is not None:
                callback({'x': action, 'i': i, 'sigma': sig[i], 'sigma_hat': sig[i], 'denoised': denoised})
            d = to_d(action, sig[i], denoised)
            if sigma_down == 0:
                action = action + d * _f(sigma_down - sig[i])
            else:
                sigma_mid = sig[i].log().lerp(sigma_down.log(), 0.5).exp()
                action_2 = action + d * _f(sigma_mid - sig[i])
                denoised_2 = model(state, action_2, goal, _sig_in(sigma_mid, action, model), **extra_args)
                d_2 = to_d(action_2, sigma_mid, denoised_2)
                action = action + d_2 * _f(sigma_down - sig[i])
                action = action + torch.randn_like(action) * _f(sigma_up)
            if scaler is not None:
                action = scaler.clip_output(action)
    return action


def linear_multistep_coeff(order, t, i, j):
    """Integral over [t_i, t_{i+1}] of the j-th Lagrange basis polynomial through the last `order` nodes
    (reference gc_sampling.py:410-422; scipy quadrature, epsrel 1e-4)."""
    from scipy import integrate
    if order - 1 > i:
        raise ValueError(f'Order {order} too high for step {i}')

    def basis(tau):
        prod = 1.
        for k in range(order):
            if k != j:
                prod *= (tau - t[i - k]) / (t[i - j] - t[i - k])
        return prod
    return integrate.quad(basis, t[i], t[i + 1], epsrel=1e-4)[0]


@torch.no_grad()
def sample_lms(model, state, action, goal, sigmas, scaler=None, extra_args=None, callback=None, disable=None, order=4):
    """Linear multistep (Adams-Bashforth in sigma) sampler (reference gc_sampling.py:425-460)."""
    extra_args = {} if extra_args is None else extra_args
    sig = _host(sigmas)
    sig_np = sig.numpy()
    ds = []
    with _hoist(model, state, goal):
        for i in range(len(sig) - 1):
            denoised = model(state, action, goal, _sig_in(sig[i], action, model), **extra_args)
            ds.append(to_d(action, sig[i], denoised))
            if len(ds) > order:
                ds.pop(0)
            if callback is not None:
                callback({'x': action, 'i': i, 'sigma': sig[i], 'sigma_hat': sig[i], 'denoised': denoised})
            cur_order = min(i + 1, order)
            coeffs = [linear_multistep_coeff(cur_order, sig_np, i, j) for j in range(cur_order)]
            action = action + sum(coeff * d for coeff, d in zip(coeffs, reversed(ds)))
            if scaler is not None:
                action = scaler.clip_output(action)
    return action


@torch.no_grad()
def sample_dpmpp_2_with_lms(model, state, action, goal, sigmas, scaler=None, extra_args=None, callback=None,
                            disable=None):
    """The reference's second name for DPM-Solver++(2M) (gc_sampling.py:785-816 is line-for-line its :699-734)."""
    return sample_dpmpp_2m(model, state, action, goal, sigmas, scaler=scaler, extra_args=extra_args,
                           callback=callback, disable=disable)


@torch.no_grad()
def sample_dpmpp_2s_ancestral(model, state, action, goal, sigmas, scaler=None, extra_args=None, callback=None,
                              disable=None, eta=1., s_noise=1., noise_sampler=None):
    """Ancestral sampling with DPM-Solver++(2S) steps (reference gc_sampling.py:864-907)."""
    extra_args = {} if extra_args is None else extra_args
    noise_sampler = default_noise_sampler(action) if noise_sampler is None else noise_sampler
    sig = _host(sigmas)
    with _hoist(model, state, goal):
        for i in range(len(sig) - 1):
            denoised = model(state, action, goal, _sig_in(sig[i], action, model), **extra_args)
            sigma_down, sigma_up = get_ancestral_step(sig[i], sig[i + 1], eta=eta)
            if callback is not None:
                callback({'action': action, 'i': i, 'sigma': sig[i], 'sigma_hat': sig[i], 'denoised': denoised})
            if sigma_down == 0:
                action = action + to_d(action, sig[i], denoised) * _f(sigma_down - sig[i])
            else:
                t, t_next = _t(sig[i]), _t(sigma_down)
                h = t_next - t
                s = t + 0.5 * h
                x_2 = _f(_sigma(s) / _sigma(t)) * action - _f((-h * 0.5).expm1()) * denoised
                denoised_2 = model(state, x_2, goal, _sig_in(_sigma(s), action, model), **extra_args)
                action = _f(_sigma(t_next) / _sigma(t)) * action - _f((-h).expm1()) * denoised_2
            action = action + noise_sampler(sig[i], sig[i + 1]) * s_noise * _f(sigma_up)
            if scaler is not None:
                action = scaler.clip_output(action)
    return action


# ------------------------------------------------------------------------------------------------
# DPM-Solver (fixed-step "fast" and adaptive) and DPM-Solver++ SDE      (reference gc_sampling.py:495-690,737-790,834-870)
# ------------------------------------------------------------------------------------------------
class BrownianTreeNoiseSampler:
    """Noise sampler backed by torchsde.BrownianTree (reference gc_sampling.py:112-160), the default of
    ``sample_dpmpp_sde``.  torchsde is an optional dependency, exactly as in the reference: without it, pass your own
    ``noise_sampler(sigma, sigma_next)``."""

    def __init__(self, x, sigma_min, sigma_max, seed=None, transform=lambda x: x):
        try:
            import torchsde
        except ImportError as e:  # pragma: no cover - environment dependent
            raise ImportError("BrownianTreeNoiseSampler needs torchsde (as the reference does); pass noise_sampler=... "
                              "to sample_dpmpp_sde instead") from e
        self.transform = transform
        t0, t1 = self.transform(torch.as_tensor(sigma_min)), self.transform(torch.as_tensor(sigma_max))
        self.sign = 1 if t0 < t1 else -1
        lo, hi = (t0, t1) if t0 < t1 else (t1, t0)
        seeds = [torch.randint(0, 2 ** 63 - 1, []).item()] if seed is None else ([seed] if isinstance(seed, int) else list(seed))
        self.batched = not (seed is None or isinstance(seed, int))
        w0 = torch.zeros_like(x[0] if self.batched else x)
        self.trees = [torchsde.BrownianTree(lo, w0, hi, entropy=sd) for sd in seeds]

    def __call__(self, sigma, sigma_next):
        t0, t1 = self.transform(torch.as_tensor(sigma)), self.transform(torch.as_tensor(sigma_next))
        sign = 1 if t0 < t1 else -1
        lo, hi = (t0, t1) if t0 < t1 else (t1, t0)
        w = torch.stack([tree(lo, hi) for tree in self.trees]) * (self.sign * sign)
        return (w if self.batched else w[0]) / (t1 - t0).abs().sqrt()


# ------------------------------------------------------------------------------------------------
# DPM-Solver (arXiv:2206.00927) behind sample_dpm_fast / sample_dpm_adaptive (reference gc_sampling.py:495-697,
# 834-870).  One exponential-integrator routine covers the three orders: with lambda = t = -ln(sigma), h the step and
# r_1 < r_2 the interior nodes of the order,
#     x(t+h) = x - sigma(t+h) * [ expm1(h) * eps_0  +  c_k * (eps_k - eps_0) ]
# where eps_k is the noise prediction at node k (evaluated on the lower-order estimate at that node), and
#     order 1: no correction;   order 2: c_1 = expm1(h) / (2 r_1);   order 3: c_2 = (expm1(h)/h - 1) / r_2.
# ------------------------------------------------------------------------------------------------
_DPM_NODES = {1: (), 2: (1 / 2,), 3: (1 / 3, 2 / 3)}


class _EpsEvaluator:
    """eps(x, t) = (x - D(x; sigma(t))) / sigma(t) of a denoiser, counting its evaluations."""

    def __init__(self, model, state, goal, extra_args, on_eval=None):
        self.model, self.state, self.goal, self.kw = model, state, goal, (extra_args or {})
        self.evals, self.on_eval = 0, on_eval

    def __call__(self, x, t):
        sig = _sigma(t)
        self.evals += 1
        if self.on_eval is not None:
            self.on_eval()
        return (x - self.model(self.state, x, self.goal, _sig_in(sig, x, self.model), **self.kw)) / _f(sig)


def _dpm_stages(eps, x, t, t_next, nodes, eps0):
    """Noise predictions at the left end and at the interior `nodes` of [t, t_next] (each on the estimate the lower
    stages give there): the list [eps_0, eps_1, ...]."""
    h = t_next - t
    out = [eps0]
    for k, r in enumerate(nodes):
        s = t + r * h
        u = x - _f(_sigma(s) * (r * h).expm1()) * eps0
        if k == 1:  # second interior node: first-node correction scaled to this node
            r1 = nodes[0]
            u = u - _f(_sigma(s) * (r / r1) * ((r * h).expm1() / (r * h) - 1)) * (out[1] - eps0)
        out.append(eps(u, s))
    return out


def _dpm_combine(x, t, t_next, nodes, stages):
    """The order-(len(nodes)+1) update from the stage values of _dpm_stages."""
    h = t_next - t
    sn = _sigma(t_next)
    new = x - _f(sn * h.expm1()) * stages[0]
    if len(nodes) == 1:
        new = new - _f(sn / (2 * nodes[0]) * h.expm1()) * (stages[1] - stages[0])
    elif len(nodes) == 2:
        new = new - _f(sn / nodes[1] * (h.expm1() / h - 1)) * (stages[2] - stages[0])
    return new


def _ancestral_split(t, t_next, t_end, eta):
    """Deterministic end point and the noise level re-injected after it (eta > 0), in t = -ln(sigma)."""
    if not eta:
        return t_next, 0.
    sd, _ = get_ancestral_step(_sigma(t), _sigma(t_next), eta)
    t_det = torch.minimum(torch.as_tensor(t_end), _t(sd))
    return t_det, (_sigma(t_next) ** 2 - _sigma(t_det) ** 2) ** 0.5


class _StepControl:
    """Proportional-integral-derivative step-size control on the inverse error history (Soederlind's digital-filter
    form, as torchdiffeq / k-diffusion use it), evaluated in the log domain, with the arctan limiter."""

    def __init__(self, h, kp, ki, kd, order, safety, eps=1e-8):
        self.h = h
        self.weights = ((kp + ki + kd) / order, -(kp + 2 * kd) / order, kd / order)
        self.safety, self.eps = safety, eps
        self.history = None  # log inverse errors: [current, previous, the one before]

    def update(self, error):
        """Feed the scaled error of the step just tried; returns whether it is accepted and rescales ``h``."""
        cur = -math.log(float(error) + self.eps)
        self.history = [cur, cur, cur] if self.history is None else [cur] + self.history[1:]
        factor = math.exp(sum(w * e for w, e in zip(self.weights, self.history)))
        factor = 1 + math.atan(factor - 1)
        ok = factor >= self.safety
        if ok:
            self.history = [cur, cur, self.history[1]]  # shift: the accepted error becomes "previous"
        self.h *= factor
        return ok


def _check_sigma_range(sigma_min, sigma_max):
    if sigma_min <= 0 or sigma_max <= 0:
        raise ValueError('sigma_min and sigma_max must not be 0')


@torch.no_grad()
def sample_dpm_fast(model, state, action, goal, sigma_min, sigma_max, n, scaler=None, extra_args=None, callback=None,
                    disable=None, eta=0., s_noise=1., noise_sampler=None):
    """DPM-Solver-Fast, fixed step size, n model evaluations (reference gc_sampling.py:673-697 / 592-624): m =
    floor(n/3)+1 uniform steps in t, third order except for the tail (.., 2, 1 when 3 | n, else .., n mod 3).
    (The reference builds its default noise sampler from an undefined name, :600, so it only runs with an explicit one;
    the default here is the action-shaped Gaussian sampler it meant.)"""
    _check_sigma_range(sigma_min, sigma_max)
    t_start, t_end = _t(torch.tensor(float(sigma_max))), _t(torch.tensor(float(sigma_min)))
    if eta and not t_end > t_start:
        raise ValueError('eta must be 0 for reverse sampling')
    noise_sampler = default_noise_sampler(action) if noise_sampler is None else noise_sampler
    with _hoist(model, state, goal):
        return _dpm_fast_run(_EpsEvaluator(model, state, goal, extra_args), action, t_start, t_end, n, eta, s_noise,
                             noise_sampler, callback)


def _dpm_fast_run(eps, action, t_start, t_end, n, eta, s_noise, noise_sampler, callback):
    m = n // 3 + 1
    grid = torch.linspace(_f(t_start), _f(t_end), m + 1)
    orders = [3] * (m - 2) + [2, 1] if n % 3 == 0 else [3] * (m - 1) + [n % 3]
    for i, order in enumerate(orders):
        t, t_next = grid[i], grid[i + 1]
        t_det, s_up = _ancestral_split(t, t_next, t_end, eta)
        e0 = eps(action, t)
        if callback is not None:
            callback({'x': action, 'i': i, 't': t, 't_up': t, 'denoised': action - _f(_sigma(t)) * e0,
                      'sigma': _sigma(t), 'sigma_hat': _sigma(t)})
        nodes = _DPM_NODES[order]
        action = _dpm_combine(action, t, t_det, nodes, _dpm_stages(eps, action, t, t_det, nodes, e0))
        if _f(s_up) != 0:
            action = action + _f(s_up) * s_noise * noise_sampler(_sigma(t), _sigma(t_next))
    return action


@torch.no_grad()
def sample_dpm_adaptive(model, state, action, goal, sigma_min, sigma_max, extra_args=None, callback=None, disable=None,
                        order=3, rtol=0.05, atol=0.0078, h_init=0.05, pcoeff=0., icoeff=1., dcoeff=0., accept_safety=0.81,
                        eta=0., s_noise=1., return_info=False, noise_sampler=None):
    """DPM-Solver-12 / -23 with adaptive step size (reference gc_sampling.py:834-870 / 626-670): the embedded pair
    (order-1, order) shares its stage evaluations -- the order-2 estimate of the 23 pair uses r_1 = 1/3, the first node
    of the order-3 step -- and the scaled difference of the two drives the step-size control.
    (The reference reads `noise_sampler` before assigning it, :633: its adaptive solver cannot run.  Parity unpinned.)"""
    _check_sigma_range(sigma_min, sigma_max)
    if order not in (2, 3):
        raise ValueError('order should be 2 or 3')
    t_start = _t(torch.tensor(float(sigma_max))).to(torch.float32)
    t_end = _t(torch.tensor(float(sigma_min))).to(torch.float32)
    if eta and not bool(t_end > t_start):
        raise ValueError('eta must be 0 for reverse sampling')
    noise_sampler = default_noise_sampler(action) if noise_sampler is None else noise_sampler
    with _hoist(model, state, goal):
        action, info = _dpm_adaptive_run(_EpsEvaluator(model, state, goal, extra_args), action, t_start, t_end, order, rtol, atol,
                                         h_init, pcoeff, icoeff, dcoeff, accept_safety, eta, s_noise, noise_sampler, callback)
    return (action, info) if return_info else action


def _dpm_adaptive_run(eps, action, t_start, t_end, order, rtol, atol, h_init, pcoeff, icoeff, dcoeff, accept_safety, eta,
                      s_noise, noise_sampler, callback):
    forward = bool(t_end > t_start)
    ctl = _StepControl(abs(h_init) if forward else -abs(h_init), pcoeff, icoeff, dcoeff, 1.5 if eta else order, accept_safety)
    info = {'steps': 0, 'nfe': 0, 'n_accept': 0, 'n_reject': 0}
    nodes_hi = _DPM_NODES[order]
    nodes_lo = nodes_hi[:-1]  # order 2: () ; order 3: (1/3,) -- the same first node, so the pair shares eps_1
    s, prev = t_start, action
    while (s < t_end - 1e-5) if forward else (s > t_end + 1e-5):
        t = torch.minimum(t_end, s + ctl.h) if forward else torch.maximum(t_end, s + ctl.h)
        t_det, s_up = _ancestral_split(s, t, t_end, eta)
        stages = _dpm_stages(eps, action, s, t_det, nodes_hi, eps(action, s))
        low = _dpm_combine(action, s, t_det, nodes_lo, stages)
        high = _dpm_combine(action, s, t_det, nodes_hi, stages)
        denoised = action - _f(_sigma(s)) * stages[0]
        tol = torch.clamp(rtol * torch.maximum(low.abs(), prev.abs()), min=atol)
        error = torch.linalg.norm((low - high) / tol) / action.numel() ** 0.5
        if ctl.update(error):
            prev = low
            action = high if _f(s_up) == 0 else high + _f(s_up) * s_noise * noise_sampler(_sigma(s), _sigma(t))
            s = t
            info['n_accept'] += 1
        else:
            info['n_reject'] += 1
        info['steps'] += 1
        info['nfe'] = eps.evals
        if callback is not None:
            callback({'x': action, 'i': info['steps'] - 1, 't': s, 't_up': s, 'denoised': denoised, 'error': error,
                      'h': ctl.h, 'sigma': _sigma(s), 'sigma_hat': _sigma(s), **info})
    return action, info


class PIDStepSizeController(_StepControl):
    """The reference's controller by its own name and call form (gc_sampling.py:495-521) over _StepControl."""

    def __init__(self, h, pcoeff, icoeff, dcoeff, order=1, accept_safety=0.81, eps=1e-8):
        super().__init__(h, pcoeff, icoeff, dcoeff, order, accept_safety, eps)

    @staticmethod
    def limiter(action):
        return 1 + math.atan(action - 1)

    def propose_step(self, error):
        return self.update(error)


class DPMSolver(torch.nn.Module):
    """The reference's solver object by its own name and method set (gc_sampling.py:524-670), for code that builds
    ``DPMSolver(model, ...)`` directly; the work is done by the routines above.  ``eps_cache`` dictionaries carry the stage
    values under the reference's keys ('eps', 'eps_r1', 'eps_r2') and are honoured the same way (by key, whatever nodes
    produced them)."""
    _KEYS = ('eps', 'eps_r1', 'eps_r2')

    def __init__(self, model, extra_args=None, eps_callback=None, info_callback=None):
        super().__init__()
        self.model = model
        self.extra_args = {} if extra_args is None else extra_args
        self.eps_callback = eps_callback
        self.info_callback = info_callback

    def t(self, sigma):
        return -sigma.log()

    def sigma(self, t):
        return t.neg().exp()

    def _evaluator(self, state, goal):
        return _EpsEvaluator(self.model, state, goal, self.extra_args, self.eps_callback)

    def eps(self, eps_cache, key, state, action, goal, t, *args, **kwargs):
        if key in eps_cache:
            return eps_cache[key], eps_cache
        e = self._evaluator(state, goal)(action, t)
        return e, {key: e, **eps_cache}

    def _step(self, nodes, state, action, goal, t, t_next, eps_cache):
        cache = dict(eps_cache or {})
        ev = self._evaluator(state, goal)
        if 'eps' not in cache:
            cache['eps'] = ev(action, t)
        stage = [1]

        def staged(u, s):  # stage k of _dpm_stages <-> key k of the reference's cache
            key = self._KEYS[stage[0]]
            stage[0] += 1
            if key not in cache:
                cache[key] = ev(u, s)
            return cache[key]
        out = _dpm_combine(action, t, t_next, nodes, _dpm_stages(staged, action, t, t_next, nodes, cache['eps']))
        return out, cache

    def dpm_solver_1_step(self, state, action, goal, t, t_next, eps_cache=None):
        return self._step((), state, action, goal, t, t_next, eps_cache)

    def dpm_solver_2_step(self, state, action, goal, t, t_next, r1=1 / 2, eps_cache=None):
        return self._step((r1,), state, action, goal, t, t_next, eps_cache)

    def dpm_solver_3_step(self, state, action, goal, t, t_next, r1=1 / 3, r2=2 / 3, eps_cache=None):
        return self._step((r1, r2), state, action, goal, t, t_next, eps_cache)

    def dpm_solver_fast(self, state, action, goal, t_start, t_end, nfe, eta=0., s_noise=1., noise_sampler=None):
        t_start, t_end = torch.as_tensor(t_start), torch.as_tensor(t_end)
        if eta and not t_end > t_start:
            raise ValueError('eta must be 0 for reverse sampling')
        noise_sampler = default_noise_sampler(action) if noise_sampler is None else noise_sampler
        with _hoist(self.model, state, goal):
            return _dpm_fast_run(self._evaluator(state, goal), action, t_start, t_end, nfe, eta, s_noise, noise_sampler,
                                 self.info_callback)

    def dpm_solver_adaptive(self, state, action, goal, t_start, t_end, order=3, rtol=0.05, atol=0.0078, h_init=0.05, pcoeff=0.,
                            icoeff=1., dcoeff=0., accept_safety=0.81, eta=0., s_noise=1., noise_sampler=None):
        if order not in (2, 3):
            raise ValueError('order should be 2 or 3')
        t_start, t_end = torch.as_tensor(t_start, dtype=torch.float32), torch.as_tensor(t_end, dtype=torch.float32)
        if eta and not bool(t_end > t_start):
            raise ValueError('eta must be 0 for reverse sampling')
        noise_sampler = default_noise_sampler(action) if noise_sampler is None else noise_sampler
        with _hoist(self.model, state, goal):
            return _dpm_adaptive_run(self._evaluator(state, goal), action, t_start, t_end, order, rtol, atol, h_init, pcoeff,
                                     icoeff, dcoeff, accept_safety, eta, s_noise, noise_sampler, self.info_callback)


@torch.no_grad()
def sample_dpmpp_sde(model, state, action, goal, sigmas, extra_args=None, callback=None, disable=None, eta=1., s_noise=1.,
                     scaler=None, noise_sampler=None, r=1 / 2):
    """DPM-Solver++ (stochastic) (reference gc_sampling.py:737-790): a midpoint evaluation at s = t + r h, ancestral
    noise at both sub-steps from ``noise_sampler(sigma, sigma_next)`` (default: a torchsde Brownian tree)."""
    extra_args = {} if extra_args is None else extra_args
    sig = _host(sigmas)
    if noise_sampler is None:
        noise_sampler = BrownianTreeNoiseSampler(action, sig[sig > 0].min(), sig.max())
    fac = 1 / (2 * r)
    with _hoist(model, state, goal):
        for i in range(len(sig) - 1):
            denoised = model(state, action, goal, _sig_in(sig[i], action, model), **extra_args)
            if callback is not None:
                callback({'x': action, 'i': i, 'sigma': sig[i], 'sigma_hat': sig[i], 'denoised': denoised})
            if sig[i + 1] == 0:
                action = action + to_d(action, sig[i], denoised) * _f(sig[i + 1] - sig[i])  # Euler
                continue
            t, t_next = _t(sig[i]), _t(sig[i + 1])
            h = t_next - t
            s = t + h * r
            sd, su = get_ancestral_step(_sigma(t), _sigma(s), eta)
            s_ = _t(sd)
            x_2 = _f(_sigma(s_) / _sigma(t)) * action - _f((t - s_).expm1()) * denoised
            if _f(su) != 0:
                x_2 = x_2 + noise_sampler(_sigma(t), _sigma(s)) * (s_noise * _f(su))
            denoised_2 = model(state, x_2, goal, _sig_in(_sigma(s), action, model), **extra_args)
            sd, su = get_ancestral_step(_sigma(t), _sigma(t_next), eta)
            t_next_ = _t(sd)
            denoised_d = (1 - fac) * denoised + fac * denoised_2
            action = _f(_sigma(t_next_) / _sigma(t)) * action - _f((t - t_next_).expm1()) * denoised_d
            if _f(su) != 0:
                action = action + noise_sampler(_sigma(t), _sigma(t_next)) * (s_noise * _f(su))
            if scaler is not None:
                action = scaler.clip_output(action)
    return action


# ------------------------------------------------------------------------------------------------
# log-likelihood (reference gc_sampling.py:468-490)
# ------------------------------------------------------------------------------------------------
# Dormand-Prince 5(4) tableau: node positions, stage weights (row i feeds stage i+1), the 5th-order solution
# weights (= the last row: first-same-as-last) and the difference to the embedded 4th-order weights.
_DP_C = (0.0, 1 / 5, 3 / 10, 4 / 5, 8 / 9, 1.0, 1.0)
_DP_A = (
    (),
    (1 / 5,),
    (3 / 40, 9 / 40),
    (44 / 45, -56 / 15, 32 / 9),
    (19372 / 6561, -25360 / 2187, 64448 / 6561, -212 / 729),
    (9017 / 3168, -355 / 33, 46732 / 5247, 49 / 176, -5103 / 18656),
    (35 / 384, 0.0, 500 / 1113, 125 / 192, -2187 / 6784, 11 / 84),
)
_DP_E = (35 / 384 - 5179 / 57600, 0.0, 500 / 1113 - 7571 / 16695, 125 / 192 - 393 / 640, -2187 / 6784 + 92097 / 339200,
         11 / 84 - 187 / 2100, -1 / 40)


def _scaled_rms(parts, scale_parts):
    """Root mean square of err / scale over ALL entries of a tuple state (one number: the step is shared)."""
    num = sum(float(((p / s) ** 2).sum()) for p, s in zip(parts, scale_parts))
    return math.sqrt(num / sum(p.numel() for p in parts))


def _dopri5(fn, y0, t0, t1, rtol, atol, max_steps=10000):
    """Adaptive Dormand-Prince 5(4) integration of y' = fn(t, y) from t0 to t1 for a tuple of tensors ``y0``: embedded
    error estimate in a mixed absolute / relative norm, first-same-as-last stage reuse, step controller
    h <- h * clip(0.9 * err^(-1/5), 0.2, 10), Hairer's starting step; the last step is shortened to end on t1.  The role
    torchdiffeq.odeint(..., method='dopri5') plays in the reference (gc_sampling.py:486) -- that package is not a
    dependency here."""
    y = tuple(y0)
    t, direction = float(t0), (1.0 if t1 >= t0 else -1.0)
    k1 = fn(t, y)
    scale = tuple(atol + rtol * p.abs() for p in y)
    d0, d1 = _scaled_rms(y, scale), _scaled_rms(k1, scale)
    h = 1e-6 if (d0 < 1e-5 or d1 < 1e-5) else 0.01 * d0 / d1
    y_probe = tuple(p + direction * h * k for p, k in zip(y, k1))
    d2 = _scaled_rms(tuple(a - b for a, b in zip(fn(t + direction * h, y_probe), k1)), scale) / h
    h = min(100 * h, max(1e-6, 1e-3 * h) if max(d1, d2) <= 1e-15 else (0.01 / max(d1, d2)) ** 0.2)
    for _ in range(max_steps):
        if (t1 - t) * direction <= 0:
            return y
        h = min(h, abs(t1 - t))
        ks = [k1]
        for i in range(1, 7):
            yi = tuple(p + direction * h * sum(a * k[j] for a, k in zip(_DP_A[i], ks) if a != 0.0)
                       for j, p in enumerate(y))
            if i == 6:
                y_new = yi
            ks.append(fn(t + direction * _DP_C[i] * h, yi))
        err = tuple(h * sum(e * k[j] for e, k in zip(_DP_E, ks) if e != 0.0) for j in range(len(y)))
        scale = tuple(atol + rtol * torch.maximum(p.abs(), q.abs()) for p, q in zip(y, y_new))
        ratio = _scaled_rms(err, scale)
        if ratio <= 1.0:  # accept
            t = t1 if h >= abs(t1 - t) else t + direction * h
            y, k1 = y_new, ks[6]
        h *= 10.0 if ratio == 0.0 else min(10.0, max(0.2, 0.9 * ratio ** -0.2))
    raise RuntimeError("_dopri5: step budget exhausted before reaching the end of the interval")


def _probe_signs(action):
    """Rademacher probe of Hutchinson's trace estimator, drawn the way the reference draws it (gc_sampling.py:475)."""
    return torch.randint_like(action, 2) * 2 - 1


@torch.no_grad()
def log_likelihood(model, state, action, goal, sigma_min, sigma_max, extra_args=None, atol=1e-4, rtol=1e-4):
    """log p(action | state, goal) by integrating the probability-flow ODE dx/dsigma = (x - D(x; sigma)) / sigma from
    sigma_min to sigma_max together with its divergence, estimated with one Rademacher probe v as v^T (d f / d x) v
    (reference gc_sampling.py:468-490).  Returns (log-likelihood per sample, {'fevals': n}).

    With this package's GCDenoiser the denoiser value and the vector-Jacobian product (dD/dx)^T v come from one HIP
    forward + input-gradient-only backward (``mdt_denoise_vjp``); any other model is differentiated with
    torch.autograd, as in the reference."""
    extra_args = {} if extra_args is None else extra_args
    v = _probe_signs(action)
    fevals = 0
    hip = isinstance(model, GCDenoiser) and not extra_args

    def flow(sigma, y):
        nonlocal fevals
        x = y[0]
        fevals += 1
        if hip:
            sg = torch.full((x.shape[0],), sigma, device=x.device, dtype=x.dtype)
            denoised, jtv = model.denoise_vjp(state, x, goal, sg, v)
            grad = (v - jtv) / sigma  # d/dx of sum(((x - D(x)) / sigma) * v)
        else:
            with torch.enable_grad():
                x = x.detach().requires_grad_()
                denoised = model(state, x, goal, x.new_full((x.shape[0],), sigma), **extra_args)
                grad = torch.autograd.grad((to_d(x, sigma, denoised) * v).sum(), x)[0]
        return to_d(x, sigma, denoised).detach(), (v * grad).flatten(1).sum(1)

    latent, delta_ll = _dopri5(flow, (action, action.new_zeros([action.shape[0]])), float(sigma_min), float(sigma_max),
                               rtol, atol)
    ll_prior = torch.distributions.Normal(0, sigma_max).log_prob(latent).flatten(1).sum(1)
    return ll_prior + delta_ll, {'fevals': fevals}
