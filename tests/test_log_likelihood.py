"""log_likelihood (reference gc_sampling.py:468-490): probability-flow ODE + Hutchinson divergence estimate.

The reference integrates with torchdiffeq's dopri5, which is installed neither here nor on the GPU box; the fixtures
tests/golden/g17_loglik_*.npz are the reference's OWN log_likelihood function run with scipy's Dormand-Prince RK45 standing
in for that integrator (tests/golden/make_golden.py::scipy_odeint), so what they pin is the reference's ODE right-hand side,
divergence term and prior term; the integrators agree to the requested tolerance, not bit for bit ("parity unpinned" for
torchdiffeq's step sequence).  CPU: the integrator on known answers; the package's log_likelihood over the oracle denoiser
(torch.autograd) against the fixtures.  GPU: mdt_denoise_vjp against float64 autograd through the oracle; log_likelihood
through the HIP path against the fixtures."""
import math

import numpy as np
import pytest
import torch

from mdt_policy_amd import synthetic
from mdt_policy_amd.models.edm_diffusion import gc_sampling as gs
from oracle import mdt_oracle as O
from tests.helpers import assert_close, cfg_of, inputs_of, load_fixture, params_of

CASES = ["mdtv_tiny", "mdt_tiny", "mdtv_default"]


def case(name):
    meta, fx = load_fixture(f"g17_loglik_{name}.npz")
    cfg = cfg_of(meta)
    state, goal, _ = inputs_of(meta)
    li = {k: torch.from_numpy(v) for k, v in synthetic.loss_inputs(meta["B"], cfg, meta["loss_seed"]).items()}
    return meta, fx, cfg, state, goal, li["actions"]


def test_dopri5_known_answers():
    # tuple state, non-autonomous, both directions: y' = -2 y + sin t, z' = t^2
    def f(t, y):
        return (-2.0 * y[0] + math.sin(t), torch.full_like(y[1], t * t))

    def exact(t, y0, t0):
        part = lambda u: (2 * math.sin(u) - math.cos(u)) / 5
        return (y0 - part(t0)) * math.exp(-2 * (t - t0)) + part(t)

    y0, z0 = torch.tensor([1.0, -0.5, 3.0], dtype=torch.float64), torch.zeros(2, dtype=torch.float64)
    for t0, t1 in ((0.0, 4.0), (4.0, 0.5)):
        n = [0]

        def g(t, y):
            n[0] += 1
            return f(t, y)

        y, z = gs._dopri5(g, (y0, z0), t0, t1, 1e-7, 1e-9)
        assert_close(y, exact(t1, y0, t0), rtol=1e-5, atol=1e-7, what="linear ODE")
        assert_close(z, torch.full_like(z0, (t1 ** 3 - t0 ** 3) / 3), rtol=1e-6, atol=1e-8, what="quadrature")
        assert n[0] < 500
    # a looser tolerance takes fewer evaluations and is less exact, but inside its own tolerance class
    y, _ = gs._dopri5(f, (y0, z0), 0.0, 4.0, 1e-3, 1e-5)
    assert_close(y, exact(4.0, y0, 0.0), rtol=2e-2, atol=1e-3, what="loose")


def test_dopri5_against_scipy_rk45():
    from scipy.integrate import solve_ivp
    rhs = lambda t, y: np.array([y[1], (1 - y[0] ** 2) * y[1] - y[0]])  # van der Pol, mu = 1
    sol = solve_ivp(rhs, (0.0, 6.0), [2.0, 0.0], method="RK45", rtol=1e-6, atol=1e-8)
    y, = gs._dopri5(lambda t, y: (torch.stack([y[0][1], (1 - y[0][0] ** 2) * y[0][1] - y[0][0]]),),
                    (torch.tensor([2.0, 0.0], dtype=torch.float64),), 0.0, 6.0, 1e-6, 1e-8)
    assert_close(y, sol.y[:, -1], rtol=1e-4, atol=1e-5, what="van der Pol")


def test_log_likelihood_of_a_gaussian_is_exact():
    """Data ~ N(0, s^2 I): the ideal denoiser is linear, D(x; sigma) = x s^2 / (s^2 + sigma^2), Hutchinson's estimate of a
    multiple of the identity is exact for sign probes, and the flow's likelihood is the N(0, (s^2 + sigma_min^2) I) density."""
    s2 = 0.7 ** 2
    model = lambda state, x, goal, sigma: x * (s2 / (s2 + sigma ** 2)).reshape(-1, 1, 1)
    torch.manual_seed(3)
    x = torch.randn(4, 10, 7, dtype=torch.float64) * 0.7
    # (sigma_max large: the prior N(0, sigma_max^2) stands for the true N(0, s^2 + sigma_max^2), a relative variance error
    #  of s^2 / sigma_max^2)
    ll, info = gs.log_likelihood(model, {}, x, None, 0.02, 4000.0, atol=1e-7, rtol=1e-7)
    want = torch.distributions.Normal(0, math.sqrt(s2 + 0.02 ** 2)).log_prob(x).flatten(1).sum(1)
    assert_close(ll, want, rtol=1e-5, atol=1e-4, what="gaussian log-likelihood")
    assert info["fevals"] > 10


class OracleModel:
    """model(state, x, goal, sigma) over the oracle, differentiable w.r.t. x (what log_likelihood's autograd branch needs)."""

    def __init__(self, meta, cfg, dtype=torch.float32):
        self.P = {k: v.to(dtype) if v.dtype.is_floating_point else v for k, v in params_of(meta).items()}
        self.cfg, self.arch, self.dtype = cfg, meta["arch"], dtype

    def __call__(self, state, x, goal, sigma):
        st = {k: (v.to(self.dtype) if torch.is_tensor(v) else v) for k, v in state.items()}
        return O.denoise(self.P, self.cfg, st, x.to(self.dtype), goal.to(self.dtype), sigma.to(self.dtype), 0.5, self.arch)


@pytest.mark.parametrize("name", CASES[:2])
def test_log_likelihood_over_the_oracle_matches_the_reference(name, monkeypatch):
    meta, fx, cfg, state, goal, action = case(name)
    monkeypatch.setattr(gs, "_probe_signs", lambda a: torch.from_numpy(fx["v"]).to(a))
    ll, info = gs.log_likelihood(OracleModel(meta, cfg), state, action, goal, meta["sigma_min"], meta["sigma_max"])
    assert_close(ll, fx["ll"], rtol=2e-3, atol=0.05, what="log-likelihood")
    assert 0.3 * meta["fevals"] <= info["fevals"] <= 3 * meta["fevals"]


def test_probe_signs_are_drawn_like_the_reference():
    meta, fx, cfg, state, goal, action = case("mdtv_tiny")
    torch.manual_seed(meta["probe_seed"])
    assert np.array_equal(gs._probe_signs(action).numpy(), fx["v"])


# ----------------------------------------------------------------------------------------------------------------
def gpu_model(meta, cfg):
    from mdt_policy_amd.models.edm_diffusion.score_wrappers import GCDenoiser
    model = GCDenoiser(cfg, 0.5)
    model.load_state_dict(params_of(meta))
    return model.cuda().eval()


def to_cuda(state):
    return {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in state.items()}


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES + ["mdtv_mlp_head", "mdtv_noise_block", "mdtv_no_ada", "mdtv_rope"])
def test_hip_denoise_vjp_matches_float64_autograd(name):
    if name in CASES:
        meta, fx, cfg, state, goal, action = case(name)
    else:  # decoder variants: the training fixtures' configurations
        meta, _ = load_fixture(f"g11_grads_{name}.npz")
        cfg = cfg_of(meta)
        state, goal, _ = inputs_of(meta)
        action = torch.from_numpy(synthetic.loss_inputs(meta["B"], cfg, meta["loss_seed"])["actions"])
    B = action.shape[0]
    model = gpu_model(meta, cfg)
    v = torch.from_numpy(synthetic.normal("vjp_probe", tuple(action.shape), 5))
    ref = OracleModel(meta, cfg, torch.float64)
    for sig in (0.004, 0.6, 35.0):
        sigma = torch.full((B,), sig) * torch.linspace(0.8, 1.25, B)
        x = (action + sigma[:, None, None] * torch.from_numpy(synthetic.normal("vjp_noise", tuple(action.shape), 6)))
        den, jtv = model.denoise_vjp(to_cuda(state), x.cuda(), goal.cuda(), sigma.cuda(), v.cuda())
        x64 = x.double().requires_grad_()
        d64 = ref(state, x64, goal, sigma)
        j64, = torch.autograd.grad((d64 * v.double()).sum(), x64)
        assert_close(den.cpu(), d64.detach(), what=f"denoised sigma={sig}")
        assert_close(jtv.cpu(), j64, rtol=2e-3, atol=2e-3 * float(j64.abs().max()), what=f"vjp sigma={sig}")
        with torch.no_grad():  # and the plain forward agrees with the tape-keeping one
            assert_close(model(to_cuda(state), x.cuda(), goal.cuda(), sigma.cuda()).cpu(), den.cpu(), rtol=1e-4, atol=1e-5,
                         what="forward vs vjp forward")


@pytest.mark.gpu
def test_hip_denoise_vjp_with_the_proprio_token():
    from tests.test_proprio import case as pcase
    meta, fx, cfg, state, goal, noise, li = pcase("tiny")
    model = gpu_model(meta, cfg)
    x = li["actions"] + li["noise_train"] * li["sigma"][:, None, None]
    v = torch.from_numpy(synthetic.normal("vjp_probe", tuple(x.shape), 5))
    den, jtv = model.denoise_vjp(to_cuda(state), x.cuda(), goal.cuda(), li["sigma"].cuda(), v.cuda())
    P = {k: t.double() for k, t in params_of(meta).items()}
    st = {k: (t.double() if torch.is_tensor(t) else t) for k, t in state.items()}
    x64 = x.double().requires_grad_()
    d64 = O.denoise(P, cfg, st, x64, goal.double(), li["sigma"].double(), 0.5, "mdtv")
    j64, = torch.autograd.grad((d64 * v.double()).sum(), x64)
    assert_close(den.cpu(), fx["denoised"], what="denoised vs the reference")
    assert_close(jtv.cpu(), j64, rtol=2e-3, atol=2e-3 * float(j64.abs().max()), what="vjp")


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_hip_log_likelihood_matches_the_reference(name, monkeypatch):
    meta, fx, cfg, state, goal, action = case(name)
    model = gpu_model(meta, cfg)
    monkeypatch.setattr(gs, "_probe_signs", lambda a: torch.from_numpy(fx["v"]).to(a))
    ll, info = gs.log_likelihood(model, to_cuda(state), action.cuda(), goal.cuda(), meta["sigma_min"], meta["sigma_max"])
    assert_close(ll.cpu(), fx["ll"], rtol=2e-3, atol=0.05, what="log-likelihood")
    assert 0.3 * meta["fevals"] <= info["fevals"] <= 3 * meta["fevals"]
    # signs drawn on the device when nothing is patched in: +-1, right shape
    monkeypatch.undo()
    s = gs._probe_signs(action.cuda())
    assert s.shape == action.shape and bool(((s == 1) | (s == -1)).all())
