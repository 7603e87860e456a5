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


@torch.no_grad()
def sample_ddim(model, state, action, goal, sigmas, scaler=None, extra_args=None, callback=None, disable=None,
                eta=1.):
    """DPM-Solver-1 / DDIM (reference gc_sampling.py:922-951):
    x <- (sigma_{i+1}/sigma_i) x - expm1(-(t_{i+1} - t_i)) D(x; sigma_i),  t = -ln sigma."""
    extra_args = {} if extra_args is None else extra_args
    if isinstance(model, GCDenoiser) and scaler is None and callback is None and not extra_args:
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
